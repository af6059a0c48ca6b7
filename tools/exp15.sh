# encoder kernel A (prev links) against the number of streams: is it bound by per-stream latency (time steps with the
# number of 8192-wave rounds) or by HBM traffic (time proportional to the streams)?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for n in 2048 4096 8192 10000 16384; do
  D=gpurun_out/prof_enc_n$n; rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/bench_encode.py --quality 0 --reps 1 --streams $n > $D/log.txt 2>&1
  python3 - $D $n <<'PY'
import sys,glob,csv
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)
out=[]
for r in csv.DictReader(open(f[0])):
    if 'enc_' in r['Name']:
        out.append('%s %.1f' % (r['Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')[:22], float(r['TotalDurationNs'])/1e6))
print('streams', sys.argv[2], ' | '.join(out))
PY
  find $D -name "*.csv" -size +1M -delete
done
