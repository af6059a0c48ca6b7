cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ch in 1024 2048 3072; do
  D=gpurun_out/prof_enc_q0_ch${ch}; rm -rf $D; mkdir -p $D
  export ALZ_ENC_CHUNK=$ch
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/bench_encode.py --quality 0 --reps 1 > $D/log.txt 2>&1
  echo "== q0 chunk $ch"
  python3 - $D <<'PY'
import sys,glob,csv
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)
tot=0
for r in csv.DictReader(open(f[0])):
    if 'enc_' in r['Name']:
        print('   %-40s calls %s total_ms %.1f' % (r['Name'][:40], r['Calls'], float(r['TotalDurationNs'])/1e6)); tot+=float(r['TotalDurationNs'])/1e6
print('   encoder kernels total ms %.1f' % tot)
PY
done
