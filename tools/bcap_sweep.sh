# usage (GPU box): bash tools/bcap_sweep.sh  -- kernel B's compare cap (EncGeom.b_cap, here through the experiment hook ALZ_BCAP) against quality, on
# 1 024 windows of Test.bmp per format; ms of kernels per call
cd $GRAFT_REPO_ROOT
for cap in 0 48 256; do
  echo "== ALZ_BCAP $cap (0: 2040)"
  ALZ_BCAP=$cap ALZ_MID_Q=${1:-5,6,7,9,10,11,15} ALZ_MID_N=1024 timeout 1200 python tools/mid_batch_encode.py yaz0 lz11 prs_be lz4_block 2>&1 | grep -v amdgpu | awk '{printf "%s %s %s | ", $1, $2, $7} END {print ""}'
done
