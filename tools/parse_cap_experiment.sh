cd "${GRAFT_REPO_ROOT:?}" || exit 1
for cap in 2040 128 64 48 32; do
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="-DALZ_PARSE_CAP=$cap" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== ALZ_PARSE_CAP $cap"
  ALZ_MID_Q=0 ALZ_MID_N=4096 timeout 600 python tools/mid_batch_encode.py yaz0 lz11 2>&1 | grep -v amdgpu
  python bench.py --configs cfg5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for c in d['configs']:
    if c['name'] in ('cfg5_q0','cfg5_yaz0_q0'): print(c['name'], c['value'], c.get('kernel_ms'), c.get('parity_ok'))
"
done
