cd $GRAFT_REPO_ROOT
python3 -c "
from auroralib.compression_amd._lib import load
from auroralib.compression_amd.batch import Context
c = Context(0); l = load()
print('occupancy (WGs/CU):', [l.alz_debug_occupancy(f) for f in range(11)])"
for n in 6144 7168 8192 9216 10000 14336 16384; do
  echo -n "streams=$n "
  python bench.py --no-cpu-baseline --no-verify --steps 5 --streams $n $@ 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
done
