# Kernel A at 15 bits + enc_narrow_kernel for the streams enc_words_kernel picks (the formats with windows above 8 KiB, hash wider than 15 bits, no
# min-length table; the formats with windows up to 8 KiB: every stream, -DALZ_NO_NARROW_WIN: not those) against kernel A at the finder's own hash
# width for all (-DALZ_NO_NARROW; -DALZ_NO_NARROW_MIN: not with the min-length table), and the threshold of the choice
# (-DALZ_NARROW_THRESH16=t: narrow below t / 16 distinct hashes per sampled position; 17 = always).  Results: docs/EXPERIMENTS.md 9.12.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
patch -p1 -N -s < tools/variants/r04_encode_switches.patch || true   # (the compile-time switches this script turns live in a patch, not in the product sources; the GPU box works on a scratch copy)
run() {
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="$1" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== flags: $1"
  for f in lz4_block snappy_raw; do
    python bench.py --mode encode --format $f --quality 8 --steps 3 --warmup 1 --no-cpu-baseline --configs none --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 $f q8', d['ms_per_step'], 'ms', d['config']['parity_ok'])"
  done
  ALZ_MID_Q=3,8,12 ALZ_MID_N=1024 timeout 900 python tools/mid_batch_encode.py lz4_block snappy_raw lzo 2>&1 | grep -v amdgpu
  ALZ_MID_DATA=text ALZ_MID_Q=8 ALZ_MID_N=256 timeout 600 python tools/mid_batch_encode.py lz4_block snappy_raw 2>&1 | grep -v amdgpu
}
for f in ${ALZ_NARROW_FLAGS:--DALZ_NO_NARROW none -DALZ_NARROW_THRESH16=4 -DALZ_NARROW_THRESH16=12 -DALZ_NARROW_THRESH16=17}; do
  if [ "$f" = none ]; then run ""; else run "$f"; fi
done
