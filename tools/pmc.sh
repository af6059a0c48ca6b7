# usage (GPU box): bash tools/pmc.sh TAG "COUNTER COUNTER ..." <bench.py args...>  -- one rocprofv3 --pmc pass with the given counters; mean per launch of
# every alz_* kernel with >= 1000 workgroups
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
TAG=$1; CTRS=$2; shift; shift
D=gpurun_out/pmc_$TAG; rm -rf $D; mkdir -p $D
rocprofv3 --pmc $CTRS --output-format csv -d $D -- python3 bench.py --no-cpu-baseline --no-verify --no-extras --inflight 1 "$@" > $D/log.txt 2>&1
python3 - $TAG <<'PY'
import csv,glob,sys,collections
tag=sys.argv[1]
per=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv'%tag, recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'alz_' in r['Kernel_Name'] and int(r.get('Grid_Size', '0') or 0) >= 64000:
            per[(r['Kernel_Name'].split('(')[0][-48:], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(per):
    print("PMC %s %s grid %s: " % (tag, k[0], k[1]) + "  ".join("%s=%.4g" % (c, sum(v)/len(v)) for c, v in sorted(per[k].items())))
PY
find $D -name "*.csv" -size +1M -delete
tail -3 $D/log.txt | cut -c1-300
