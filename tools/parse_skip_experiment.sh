# enc_parse_emit_kernel<FMT, true>: the look-ahead stages of windows the cursor has already jumped over left out -- for LZ11 / LZ40 by default, for every
# flag-bit format with -DALZ_PARSE_SKIP_ALL (docs/EXPERIMENTS.md 9.9).
cd "${GRAFT_REPO_ROOT:?}" || exit 1
patch -p1 -N -s < tools/variants/r04_encode_switches.patch || true   # (the compile-time switches this script turns live in a patch, not in the product sources; the GPU box works on a scratch copy)
run() {
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="$1" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== flags: $1"
  for f in lzss yaz0; do
  python bench.py --mode encode --format $f --quality 0 --steps 5 --warmup 1 --no-cpu-baseline --configs none --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 $f q0', d['ms_per_step'], 'ms')"
  done
  ALZ_MID_Q=0 ALZ_MID_N=1024,4096 timeout 600 python tools/mid_batch_encode.py yaz0 lz11 lz10 2>&1 | grep -v amdgpu
  ALZ_MID_DATA=text ALZ_MID_Q=0 ALZ_MID_N=256 timeout 600 python tools/mid_batch_encode.py yaz0 lz11 2>&1 | grep -v amdgpu
}
run ""
run "-DALZ_PARSE_SKIP_ALL"
run ""
run "-DALZ_PARSE_SKIP_ALL"
