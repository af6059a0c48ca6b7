cd $GRAFT_REPO_ROOT
echo "== bench default"; timeout 1200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','ms_per_step','n_gpus','steps','dtype')})
print('roofline',d['roofline']); print('cpu_baseline',{k:d['cpu_baseline'][k] for k in ('value','cores','kind','single_thread')})
print('pipelined',d['config'].get('pipelined'))
for c in d.get('configs') or []: print(c.get('name'), c.get('value'), c.get('unit'), c.get('ms_per_step') or c.get('kernel_ms'), c.get('parity_ok'), (c.get('roofline') or {}).get('frac'), (c.get('roofline') or {}).get('traffic'))
PY
