# usage (GPU box): bash tools/cfg4_quick.sh -- the cfg4 shard (5 000 mixed streams) twice, value and ms per step
for i in 1 2; do python bench.py --configs cfg4 --no-cpu-baseline --no-extras --inflight 1 --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([(c['name'], c['value'], c['ms_per_step']) for c in d['configs']])"; done
