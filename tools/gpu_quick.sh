# usage (on the GPU box, via gpurun): bash tools/gpu_quick.sh [formats...]  -- parity tests, then one bench line per format
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for f in ${@:-yaz0}; do
  timeout 300 python bench.py --no-cpu-baseline --format $f --steps 10 2>&1 | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$f', d['value'], 'GiB/s kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'ok', d['config']['parity_ok'])
except Exception as e: print('$f FAILED', e)"
done
