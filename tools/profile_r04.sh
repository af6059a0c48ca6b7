# usage (GPU box): bash tools/profile_r04.sh [fmt...]  -- round-4 evidence: per format tools/gpu_profile.sh (kernel stats + separate PMC passes)
# -> gpurun_out/r04_<fmt>.md; then the named configurations (cfg2, cfg3 at 100 000 blocks, the cfg4 shard, the Test.bmp windows): kernel stats +
# FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/r04_cfg*.md and the traffic.json lines
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
for f in ${@:-yaz0 lz10 lz11 yay0 mio0 lzss prs_be lz4_block lzo snappy_raw mixed}; do
  bash tools/gpu_profile.sh r04_$f $f > /dev/null 2>&1
  echo "== $f"; grep -E "alz_decode|corrected bytes|VALU busy|SQ_INSTS" gpurun_out/r04_$f.md | cut -c1-200
done
while read key fmt n kib extra; do
  D=gpurun_out/prof_r04_$key; rm -rf $D; mkdir -p $D
  B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-extras --configs none --inflight 1 --format $fmt --streams $n --stream-kib $kib $extra"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- $B > $D/stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- $B > $D/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- $B > $D/write.log 2>&1
  python3 tools/profile_summary.py gpurun_out/r04_$key.md --stats $D/stats --fetch $D/fetch --write $D/write --cmd "rocprofv3 --kernel-trace --stats -- $B  (FETCH_SIZE / WRITE_SIZE: same command, separate --pmc passes)" > /dev/null
  python3 - $key $fmt $n $kib <<'PY'
import csv,glob,sys,collections
key,fmt,n,kib=sys.argv[1:5]
tot={}
for c in ('FETCH_SIZE','WRITE_SIZE'):
    per=collections.defaultdict(float); launches=collections.Counter()
    for fn in glob.glob('gpurun_out/prof_r04_%s/%s/**/*counter_collection.csv'%(key,'fetch' if c=='FETCH_SIZE' else 'write'), recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'alz_decode' in r['Kernel_Name'] and r['Counter_Name']==c:
                per[r['Kernel_Name']]+=float(r['Counter_Value']); launches[r['Kernel_Name']]+=1
    tot[c]=sum(v/launches[k] for k,v in per.items())
print('TRAFFIC "%s:%s:%s": %d,' % (fmt,n,kib,int(tot['FETCH_SIZE']*1024*2+tot['WRITE_SIZE']*1024)))
PY
  find $D -name "*.csv" -size +2M -delete
  echo "== $key"; grep -E "alz_decode" gpurun_out/r04_$key.md | cut -c1-160
done <<LIST
cfg2 yaz0 10000 64
cfg4_shard mixed 5000 256
cfg3_100000 lz4_block 100000 256
LIST
# the Test.bmp windows (bench.py --configs realistic: five formats), a 64-stream headline batch beside them
D=gpurun_out/prof_r04_realistic; rm -rf $D; mkdir -p $D
B="python3 bench.py --streams 64 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras --configs realistic --inflight 1"
rocprofv3 --kernel-trace --output-format csv -d $D/stats -- $B > $D/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- $B > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- $B > $D/write.log 2>&1
python3 tools/trace_by_grid.py gpurun_out/r04_realistic.md "realistic: the 256 KiB windows of Test.bmp, 10 000 streams per format (yaz0, lz10, lz11, prs_be, lz4_block)" 5000 --stats $D/stats --fetch $D/fetch --write $D/write --cmd "rocprofv3 --kernel-trace -- $B"
find $D -name "*.csv" -size +2M -delete
grep '^{' $D/stats.log | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
for c in d.get('configs') or []: print(c.get('name'), c.get('value'), c.get('ms_per_step'), c.get('roofline',{}).get('kernel_ms'), c.get('parity_ok'), c.get('error',''))"
