# LZ4 blocks / raw Snappy: enc_roles_kernel + enc_emit_seq_kernel (and kernel B at quality 0) against enc_parse_seq_kernel, and the bytes the
# fused kernel compares per position at quality 0 (ALZ_SEQ_PARSE_CAP).  Results: docs/EXPERIMENTS.md 9.9.
cd $GRAFT_REPO_ROOT
run() {
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="$1" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== flags: $1"
  for q in 0 8; do
    python bench.py --mode encode --format lz4_block --quality $q --steps 3 --warmup 1 --no-cpu-baseline --configs none --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 lz4_block q$q', d['ms_per_step'], 'ms')"
  done
  ALZ_MID_Q=0,8,12 ALZ_MID_N=1024 timeout 600 python tools/mid_batch_encode.py lz4_block snappy_raw 2>&1 | grep -v amdgpu
  ALZ_MID_DATA=text ALZ_MID_Q=0,8 ALZ_MID_N=256 timeout 600 python tools/mid_batch_encode.py lz4_block 2>&1 | grep -v amdgpu
}
#run "-DALZ_SEQ_TWO_KERNELS"
run ""
run "-DALZ_SEQ_PARSE_CAP=32"
run "-DALZ_SEQ_PARSE_CAP=64"
run "-DALZ_SEQ_PARSE_CAP=128"
