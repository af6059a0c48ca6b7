# usage (GPU box): bash tools/r3_prs.sh  -- kernel ms of PRS (product build and variants under build/variants/prs_*.so), then counters of the product build
cd $GRAFT_REPO_ROOT
bash tools/ab.sh prs_be
ALZ_PRS2=0 bash tools/ab.sh prs_be
cp auroralib/compression_amd/libauroralz.so /tmp/keep.so
for v in build/variants/prs_*.so; do
  [ -f "$v" ] || continue
  cp $v auroralib/compression_amd/libauroralz.so
  echo "== $v"; bash tools/ab.sh prs_be
done
cp /tmp/keep.so auroralib/compression_amd/libauroralz.so
bash tools/gpu_profile.sh r03_prs_be prs_be > /dev/null 2>&1
tail -28 gpurun_out/r03_prs_be.md
bash tools/ab.sh lzo lz4_block snappy_raw mixed
