# usage (GPU box): bash tools/traffic.sh TAG <bench.py args...>  -- HBM traffic per launch of every alz_* kernel of the run with >= 1000 workgroups:
# two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
TAG=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  D=gpurun_out/traffic_${TAG}_$c; rm -rf $D; mkdir -p $D
  rocprofv3 --pmc $c --output-format csv -d $D -- python3 bench.py --no-cpu-baseline --no-verify --no-extras --inflight 1 "$@" > $D/log.txt 2>&1
done
python3 - $TAG <<'PY'
import csv,glob,sys,collections
tag=sys.argv[1]
acc={}
for c in ('FETCH_SIZE','WRITE_SIZE'):
    per=collections.defaultdict(list)
    for fn in glob.glob('gpurun_out/traffic_%s_%s/**/*counter_collection.csv'%(tag,c), recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'alz_' in r['Kernel_Name'] and r['Counter_Name']==c and int(r.get('Grid_Size', r.get('Grid_Size_X', '0')) or 0) >= 64000:
                per[(r['Kernel_Name'].split('(')[0], r.get('Grid_Size', r.get('Grid_Size_X')))].append(float(r['Counter_Value']))
    acc[c]=per
for k in sorted(acc['FETCH_SIZE']):
    f=acc['FETCH_SIZE'][k]; w=acc['WRITE_SIZE'].get(k,[0])
    fb=sum(f)/len(f)*1024*2; wb=sum(w)/len(w)*1024
    print("TRAFFIC %s %-60s grid %-9s fetch(x2) %.3f GB  write %.3f GB  total %.3f GB  (%d launches)" % (tag, k[0][-60:], k[1], fb/1e9, wb/1e9, (fb+wb)/1e9, len(f)))
PY
find gpurun_out/traffic_${TAG}_* -name "*.csv" -size +1M -delete
