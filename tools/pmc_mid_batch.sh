# usage (GPU box): bash tools/pmc_mid_batch.sh [format] [quality]  -- instruction / busy counters of the encoder kernels on 1 024 windows of Test.bmp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp ALZ_MID_N=1024 ALZ_MID_Q=${2:-8}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); D=gpurun_out/pmc_mid_$i; rm -rf $D; mkdir -p $D
  rocprofv3 --pmc $set --output-format csv -d $D -- python3 tools/mid_batch_encode.py ${1:-yaz0} > $D/log.txt 2>&1
  python3 - $D <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'enc_' in r['Kernel_Name']:
            k=r['Kernel_Name'][r['Kernel_Name'].find('enc_'):][:28]
            acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in acc.items(): print(k, {a:'%.3g'%(b/cnt[(k,a)]) for a,b in v.items()})
PY
  find $D -name "*.csv" -size +1M -delete
done
