# usage (GPU box): bash tools/pmc_realistic.sh -- instruction counters of the Yaz0 kernel on the synthetic batch and on the Test.bmp windows
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
D=gpurun_out/pmc_real; rm -rf $D; mkdir -p $D
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras --configs realistic --inflight 1 --format yaz0"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $D/p$i -- $B > $D/p$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_real/p*')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        rows=collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if 'fast_kernelILi3' not in r['Kernel_Name']: continue
            rows.setdefault(r['Dispatch_Id'],{})[r['Counter_Name']]=float(r['Counter_Value'])
        for k,v in rows.items(): print(k, {a:'%.4g'%b for a,b in v.items()})
PY
find $D -name "*.csv" -size +2M -delete
