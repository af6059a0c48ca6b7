# usage (GPU box): bash tools/probe_text_experiment.sh  -- 1 024 windows of program text and prose (the repository's own files) at quality 4 / 8 / 12:
# the probe's choice (default build) against every stream on the one-position-per-lane kernels (threshold 0) and none (threshold 17)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for thr in 4 0 17; do
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="-DALZ_PROBE_THRESH16=$thr" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== ALZ_PROBE_THRESH16 $thr"
  ALZ_MID_DATA=text ALZ_MID_Q=4,8,12 ALZ_MID_N=1024 timeout 900 python tools/mid_batch_encode.py lzss yaz0 lz4_block 2>&1 | grep -v amdgpu
done
touch auroralib/compression_amd/csrc/alz_encode.hip; bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
