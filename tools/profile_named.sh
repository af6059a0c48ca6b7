# usage (GPU box): bash tools/profile_named.sh fmt...  -- tools/gpu_profile.sh for each format; summaries in gpurun_out/r01_final_<fmt>.md
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  bash tools/gpu_profile.sh r01_final_$f $f > /dev/null 2>&1
  echo "== $f"; grep -E "alz_decode|corrected bytes|VALU busy" gpurun_out/r01_final_$f.md | cut -c1-200
done
