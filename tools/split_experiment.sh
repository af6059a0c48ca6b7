# usage (GPU box): bash tools/split_experiment.sh  -- the parse / copy SPLIT experiment for LZ4 blocks (alz_ctx_set_kernel_variant 4; docs/EXPERIMENTS.md 11.3) beside the product kernel:
# bench.py's 10 000 x 256 KiB LZ4 batch (verified against the oracle), kernel times (rocprofv3 --kernel-trace --stats), HBM traffic (FETCH_SIZE / WRITE_SIZE passes)
# The experiment's code is NOT in the product tree: tools/variants/r06_split.patch adds it (alz_ctx_set_kernel_variant 4: parse kernel + copy kernel with the product's 4 KiB ring and
# read-back; 5: the copy kernel with the whole 64 KiB window in LDS).  This script applies it to the GPU box's scratch copy, builds, measures and puts the tree back.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
git apply tools/variants/r06_split.patch 2>/dev/null || patch -p1 < tools/variants/r06_split.patch || exit 1
cp auroralib/compression_amd/libauroralz.so /tmp/libauroralz.keep
trap 'patch -R -p1 < tools/variants/r06_split.patch > /dev/null; cp /tmp/libauroralz.keep auroralib/compression_amd/libauroralz.so; touch auroralib/compression_amd/libauroralz.so' EXIT
bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1 || exit 1
for v in 0 4 5; do
  echo "== kernel variant $v"
  B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --configs none --inflight 1 --format lz4_block --kernel-variant $v"
  timeout 300 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench: value', d['value'], 'GiB/s  ms_per_step', d['ms_per_step'], ' parity', d['config'].get('parity_ok'), d['config'].get('verified_vs_oracle'))"
  D=gpurun_out/split_v$v; rm -rf $D; mkdir -p $D
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- $B --no-verify > $D/stats.log 2>&1
  find $D/stats -name "*kernel_stats.csv" | head -1 | xargs python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'alz_' in r['Name']: print('  ', r['Name'][:70], r['Calls'], 'avg ms %.4f' % (float(r['AverageNs'])/1e6))
"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $D/$c -- $B --no-verify > $D/$c.log 2>&1
  done
  python3 - $D <<'PY'
import csv,glob,sys,collections
tot={}
for c in ('FETCH_SIZE','WRITE_SIZE'):
    per=collections.defaultdict(float); n=collections.Counter()
    for fn in glob.glob(sys.argv[1]+'/'+c+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'alz_' in r['Kernel_Name'] and r['Counter_Name']==c: per[r['Kernel_Name'][:60]]+=float(r['Counter_Value']); n[r['Kernel_Name'][:60]]+=1
    for k in per: print('   %s %s per launch: %.3f GB (raw KiB counter x 1024%s)' % (c, k[:50], per[k]/n[k]*1024*(2 if c=='FETCH_SIZE' else 1)/1e9, ' x 2' if c=='FETCH_SIZE' else ''))
PY
  find $D -name "*.csv" -size +2M -delete
done
