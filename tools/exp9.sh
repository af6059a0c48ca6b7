# encoder kernel times (rocprofv3 --kernel-trace --stats) with the old fixed chunk (4096 streams) and the adaptive one
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for q in 0 8; do for ch in 4096 0; do
  D=gpurun_out/prof_enc_q${q}_ch${ch}; rm -rf $D; mkdir -p $D
  if [ $ch != 0 ]; then export ALZ_ENC_CHUNK=$ch; else unset ALZ_ENC_CHUNK; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/bench_encode.py --quality $q --reps 1 > $D/log.txt 2>&1
  echo "== q$q chunk $ch"; tail -1 $D/log.txt
  python3 - $D <<'PY'
import sys,glob,csv
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)
tot=0
for r in csv.DictReader(open(f[0])):
    if r['Name'].startswith('void (anonymous namespace)::enc_') or 'enc_' in r['Name']:
        print('   %-40s calls %s total_ms %.1f' % (r['Name'][:40], r['Calls'], float(r['TotalDurationNs'])/1e6)); tot+=float(r['TotalDurationNs'])/1e6
print('   encoder kernels total ms %.1f' % tot)
PY
done; done
