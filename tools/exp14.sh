# encoder: prev() links from LDS (windows <= 4 KiB, ALZ_ENC_LDS_PREV=1) against the head tables in HBM (default)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "encode or wrapper or vectors" 2>&1 | tail -2
for q in 0 8 12; do for hbm in 0 1; do
  D=gpurun_out/prof_enc_q${q}_hbm${hbm}; rm -rf $D; mkdir -p $D
  if [ $hbm = 0 ]; then export ALZ_ENC_LDS_PREV=1; else unset ALZ_ENC_LDS_PREV; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/bench_encode.py --quality $q --reps 1 > $D/log.txt 2>&1
  echo "== q$q hbm_prev=$hbm"; tail -1 $D/log.txt | cut -c1-200
  python3 - $D <<'PY'
import sys,glob,csv
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)
tot=0
for r in csv.DictReader(open(f[0])):
    if 'enc_' in r['Name']:
        print('   %-44s calls %s total_ms %.1f' % (r['Name'][:44], r['Calls'], float(r['TotalDurationNs'])/1e6)); tot+=float(r['TotalDurationNs'])/1e6
print('   encoder kernels total ms %.1f' % tot)
PY
done; done
