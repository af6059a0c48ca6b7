# usage (GPU box): bash tools/ab.sh [formats...]  -- bench lines (2 in flight / back to back / kernel ms) for the current build
cd $GRAFT_REPO_ROOT
for f in ${@:-yaz0}; do
  echo -n "$f "
  python bench.py --no-cpu-baseline --steps 20 --format $f 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['back_to_back']['value'], d['roofline']['kernel_ms'], d['config']['parity_ok'])"
done
