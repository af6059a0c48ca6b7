# usage (GPU box): bash tools/traffic_cfgs.sh -- HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes) of the cfg2 / cfg3 / cfg4 workloads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
while read key fmt n kib; do
  for c in FETCH_SIZE WRITE_SIZE; do
    D=gpurun_out/traffic_${key}_$c; rm -rf $D; mkdir -p $D
    rocprofv3 --pmc $c --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras --configs none --inflight 1 --format $fmt --streams $n --stream-kib $kib > $D/log.txt 2>&1
  done
  python3 - $key $fmt $n $kib <<'PY'
import csv,glob,sys,collections
key,fmt,n,kib=sys.argv[1:5]
tot={}
for c in ('FETCH_SIZE','WRITE_SIZE'):
    per=collections.defaultdict(float); launches=collections.Counter()
    for fn in glob.glob('gpurun_out/traffic_%s_%s/**/*counter_collection.csv'%(key,c), recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'alz_decode' in r['Kernel_Name'] and r['Counter_Name']==c:
                per[r['Kernel_Name']]+=float(r['Counter_Value']); launches[r['Kernel_Name']]+=1
    # per launch of the batch = sum over the kernels of one step
    tot[c]=sum(v/launches[k] for k,v in per.items())
b=int(tot['FETCH_SIZE']*1024*2+tot['WRITE_SIZE']*1024)
print('"%s:%s:%s": %d,   # fetch raw KiB %.0f write KiB %.0f' % (fmt,n,kib,b,tot['FETCH_SIZE'],tot['WRITE_SIZE']))
PY
  find gpurun_out/traffic_${key}_* -name "*.csv" -size +1M -delete
done <<LIST
cfg2 yaz0 10000 64
cfg4 mixed 5000 256
cfg3 lz4_block 100000 256
LIST
