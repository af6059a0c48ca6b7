cd "${GRAFT_REPO_ROOT:?}" || exit 1
for lg in 2 3 4; do
  touch auroralib/compression_amd/csrc/alz_big.hip
  ALZ_EXTRA_FLAGS="-DBIG_LOG=${lg}u" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== BIG_LOG $lg"
  timeout 600 python bench.py --configs single --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
x=[c for c in d['configs'] if c['name'].startswith('single_') and not c['name'].startswith('single_compress')]
print('mean call %.4f ms, mean kernels %.4f ms, all ok %s' % (sum(c['ms_per_call'] for c in x)/len(x), sum(c['kernel_ms'] for c in x)/len(x), all(c['parity_ok'] for c in x)))
"
done
touch auroralib/compression_amd/csrc/alz_big.hip; bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
