# usage (GPU box): bash tools/profile_r02.sh fmt...  -- tools/gpu_profile.sh for each format (kernel stats + separate PMC passes);
# summaries land in gpurun_out/r02_<fmt>.md (copy the ones to be judged into profiles/)
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  bash tools/gpu_profile.sh r02_$f $f > /dev/null 2>&1
  echo "== $f"; grep -E "alz_decode|corrected bytes|VALU busy|SQ_INSTS" gpurun_out/r02_$f.md | cut -c1-220
done
