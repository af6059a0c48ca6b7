# HBM traffic of the encoder kernels (cfg5, Q0): separate FETCH_SIZE / WRITE_SIZE passes, per kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  D=gpurun_out/prof_enc_$c; rm -rf $D; mkdir -p $D
  rocprofv3 --pmc $c --output-format csv -d $D -- python3 tools/bench_encode.py --quality 0 --reps 1 > $D/log.txt 2>&1
  python3 - $D $c <<'PY'
import sys,glob,csv,collections
tot=collections.Counter()
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name')==sys.argv[2]:
            tot[r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')[:28]] += float(r['Counter_Value'])
for k,v in tot.most_common(8): print(sys.argv[2], k, '%.2f GB raw (KiB units x 1024)' % (v*1024/1e9))
PY
  find $D -name "*.csv" -size +1M -delete
done
