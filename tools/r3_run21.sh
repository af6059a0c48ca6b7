cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 2400 python -m pytest tests/test_gpu_encode.py tests/test_gpu_containers.py -q -m gpu 2>&1 | tail -3
for q in 4 8 12 15; do echo "== enc lzss q$q"; bash tools/enc_kernels.sh lzss $q | grep dense; done
echo "== enc lz4 q8"; bash tools/enc_kernels.sh lz4_block 8 | grep dense
