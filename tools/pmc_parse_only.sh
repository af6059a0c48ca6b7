# usage (GPU box): bash tools/pmc_parse_only.sh fmt...  -- instruction counters of the token-queue kernels built parse-only (-DALZ_QEXP=3) and whole
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in parse whole; do
  rm -rf auroralib/compression_amd/csrc/_obj
  if [ $mode = parse ]; then ALZ_EXTRA_FLAGS="-DALZ_QEXP=3" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1; else bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1; fi
  for f in "$@"; do
    D=gpurun_out/pmc_po_${mode}_$f; rm -rf $D; mkdir -p $D
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras --configs none --inflight 1 --format $f > $D/log.txt 2>&1
    python3 - $D $mode $f <<'PY'
import csv,glob,sys,collections
d,mode,f=sys.argv[1:4]
acc=collections.defaultdict(list)
for fn in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'alz_decode' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(mode, f, {k:'%.3g'%(sum(v)/len(v)) for k,v in acc.items()})
PY
    find $D -name "*.csv" -size +1M -delete
  done
done
