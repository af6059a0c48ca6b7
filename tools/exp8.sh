cd $GRAFT_REPO_ROOT
for nf in 2 3 4; do
  echo -n "inflight $nf: "
  python bench.py --no-cpu-baseline --no-verify --steps 30 --inflight $nf 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['back_to_back']['value'])"
done
