# LZ4 blocks / raw Snappy / PRS: enc_roles_kernel + the emitter (and kernel B at quality 0) as separate kernels (-DALZ_SEQ_TWO_KERNELS) against
# the walk -- and at quality 0 the search -- inside the emitter (WinParse), and the bytes it compares per position at quality 0
# (-DALZ_SEQ_PARSE_CAP=.. for certain, -DALZ_SEQ_PARSE_CAP_HI=.. while fewer than -DALZ_SEQ_PARSE_MANY=.. lanes of a window are still equal).  Results: docs/EXPERIMENTS.md 9.9.  (-DALZ_SEQ_TWO_KERNELS and the kernels behind it exist up to commit 302184c; the
# default list below only sweeps the cap.)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
run() {
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="$1" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== flags: $1"
  for f in lz4_block prs_be; do for q in 0 8; do
    python bench.py --mode encode --format $f --quality $q --steps 3 --warmup 1 --no-cpu-baseline --configs none --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 $f q$q', d['ms_per_step'], 'ms')"
  done; done
  ALZ_MID_Q=0,8,12 ALZ_MID_N=1024 timeout 600 python tools/mid_batch_encode.py lz4_block snappy_raw prs_be 2>&1 | grep -v amdgpu
  ALZ_MID_DATA=text ALZ_MID_Q=0,8 ALZ_MID_N=256 timeout 600 python tools/mid_batch_encode.py lz4_block prs_be 2>&1 | grep -v amdgpu
}
for f in ${ALZ_SEQ_EXPERIMENT_FLAGS:-none -DALZ_SEQ_PARSE_CAP=32 -DALZ_SEQ_PARSE_CAP=64 -DALZ_SEQ_PARSE_CAP=128}; do
  if [ "$f" = none ]; then run ""; else run "$f"; fi
done
