# usage (GPU box): bash tools/profile_big_encode.sh  -- rocprofv3 kernel trace of ONE 1 000 KiB stream of Test.bmp through the whole-GPU
# encode path (csrc/alz_encode_big.h), every north-star body at quality 0 / 8 / 15: per-kernel microseconds of the fourth call of each case
# (tools/kernel_breakdown.py) -> gpurun_out/r04_big_encode.txt (copy to profiles/)
cd /tmp && export TMPDIR=/tmp ALZ_SINGLE_MODES=big
rm -rf /tmp/pbe
timeout 900 rocprofv3 --kernel-trace -d /tmp/pbe -o enc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/single_encode.py lzss lz10 lz11 yaz0 yay0 mio0 prs_be lz4_block lzo snappy_raw > /tmp/pbe.out 2>&1
f=$(find /tmp/pbe -name "*kernel_trace.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
{
  echo "# rocprofv3 --kernel-trace -- python3 tools/single_encode.py <ten bodies> (ALZ_SINGLE_MODES=big): kernel microseconds of the 4th call per case"
  echo "# cases in order: lzss lz10 lz11 yaz0 yay0 mio0 prs_be lz4_block lzo snappy_raw, each at quality 0, 8, 15 (case index = 3 * format + quality index)"
  python3 $GRAFT_REPO_ROOT/tools/kernel_breakdown.py $f
  echo "# the same calls, wall clock through alz_encode_batch on host buffers:"
  grep -v amdgpu /tmp/pbe.out | cut -c1-160
} > $GRAFT_REPO_ROOT/gpurun_out/r04_big_encode.txt
tail -5 $GRAFT_REPO_ROOT/gpurun_out/r04_big_encode.txt
