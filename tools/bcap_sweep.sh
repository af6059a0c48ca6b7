# usage (GPU box): bash tools/bcap_sweep.sh [qualities]  -- kernel B's compare cap against quality on 1 024 windows of Test.bmp per format; ms of
# kernels per call.  One build per cap (-DALZ_BCAP_FORCE=<cap> overrides choose_b_cap); the default build is restored at the end.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
patch -p1 -N -s < tools/variants/r04_encode_switches.patch || true   # (the compile-time switches this script turns live in a patch, not in the product sources; the GPU box works on a scratch copy)
for cap in 2040 256 96 48; do
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="-DALZ_BCAP_FORCE=$cap" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== cap $cap"
  ALZ_MID_Q=${1:-0,1,2,4,5,6,7,8,9,10,11,12,15} ALZ_MID_N=1024 timeout 1500 python tools/mid_batch_encode.py yaz0 lz11 prs_be lz4_block 2>&1 | grep -v amdgpu | awk '{printf "%s %s %s | ", $1, $2, $7} END {print ""}'
done
touch auroralib/compression_amd/csrc/alz_encode.hip; bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
