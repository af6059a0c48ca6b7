cd $GRAFT_REPO_ROOT
bash tools/profile_r03.sh lz4_block lzo mixed 2>&1 | grep -v amdgpu.ids
