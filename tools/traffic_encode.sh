# usage (GPU box): bash tools/traffic_encode.sh -- HBM traffic of the compression kernels (cfg5, Q0 and Q8): FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
for q in 0 8 15; do
  for c in FETCH_SIZE WRITE_SIZE; do
    D=gpurun_out/traffic_enc_q${q}_$c; rm -rf $D; mkdir -p $D
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $D -- python3 tools/bench_encode.py --quality $q --reps 1 > $D/log.txt 2>&1
  done
  python3 - $q <<'PY'
import csv,glob,sys,collections
q=sys.argv[1]
tot={}; per={}
for c in ('FETCH_SIZE','WRITE_SIZE'):
    acc=collections.defaultdict(float)
    for fn in glob.glob('gpurun_out/traffic_enc_q%s_%s/**/*counter_collection.csv'%(q,c), recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'enc_' in r['Kernel_Name'] and r['Counter_Name']==c:
                acc[r['Kernel_Name'][r['Kernel_Name'].find('enc_'):][:22]]+=float(r['Counter_Value'])
    per[c]=acc; tot[c]=sum(acc.values())
b=int(tot['FETCH_SIZE']*1024*2+tot['WRITE_SIZE']*1024)
print('"lzss_encode_q%s:10000:256": %d,   # fetch raw KiB %.0f write KiB %.0f' % (q,b,tot['FETCH_SIZE'],tot['WRITE_SIZE']))
for k in per['FETCH_SIZE']: print('   ', k, 'fetch x2 %.2f GB  write %.2f GB' % (per['FETCH_SIZE'][k]*2048/1e9, per['WRITE_SIZE'].get(k,0)*1024/1e9))
PY
  find gpurun_out/traffic_enc_q${q}_* -name "*.csv" -size +1M -delete
done
