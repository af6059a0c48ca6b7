# usage (GPU box): bash tools/exp2.sh  -- builds variants given as extra compiler flags, prints bench for a few stream counts
cd $GRAFT_REPO_ROOT
i=0
while read -r flags; do
  i=$((i+1))
  rm -rf auroralib/compression_amd/csrc/_obj
  ALZ_EXTRA_FLAGS="$flags" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "VARIANT $i: $flags"
  python3 -c "
from auroralib.compression_amd._lib import load
from auroralib.compression_amd.batch import Context
c = Context(0); l = load()
print('occupancy:', [l.alz_debug_occupancy(f) for f in range(6)])" 2>/dev/null
  for n in 6144 7168 10000 16384; do
    echo -n "  streams=$n "
    python bench.py --no-cpu-baseline --no-verify --steps 5 --streams $n 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
  done
done <<'VARIANTS'
-DALZ_DUMMY=1
-DALZ_FAST_ATTR=__attribute__((amdgpu_waves_per_eu(7)))
-DALZ_FAST_ATTR=__attribute__((amdgpu_num_sgpr(80)))
VARIANTS
