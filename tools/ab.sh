# usage (GPU box): bash tools/ab.sh [formats...]  -- per format: GiB/s one batch in flight / two in flight / kernel ms / parity
cd $GRAFT_REPO_ROOT
for f in ${@:-yaz0}; do
  echo -n "$f "
  python bench.py --no-cpu-baseline --no-extras --configs none --no-verify --steps 20 --format $f 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], (d['config']['pipelined'] or {}).get('value'), d['roofline']['kernel_ms'], d['config']['parity_ok'])"
done
