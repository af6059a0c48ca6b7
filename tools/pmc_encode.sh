# usage (GPU box): bash tools/pmc_encode.sh [quality] -- instruction / busy counters of the encoder kernels (cfg5 shape)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
q=${1:-8}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); D=gpurun_out/pmc_enc_$i; rm -rf $D; mkdir -p $D
  rocprofv3 --pmc $set --output-format csv -d $D -- python3 tools/bench_encode.py --quality $q --reps 1 > $D/log.txt 2>&1
  python3 - $D <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k=''
        if 'enc_' in r['Kernel_Name']: acc[r['Kernel_Name'][r['Kernel_Name'].find('enc_'):][:24]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items(): print(k, {a:'%.3g'%b for a,b in v.items()})
PY
  find $D -name "*.csv" -size +1M -delete
done
