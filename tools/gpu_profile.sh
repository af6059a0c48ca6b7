# usage (on the GPU box): bash tools/gpu_profile.sh NAME FORMAT [extra bench args]
# kernel stats + separate PMC passes (never combined with other trace domains); summary lands in gpurun_out/NAME.md
cd "${GRAFT_REPO_ROOT:?}" || exit 1
NAME=$1; FMT=$2; shift 2
export TMPDIR=/tmp
D=gpurun_out/prof_$NAME
rm -rf $D; mkdir -p $D
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-extras --configs none --inflight 1 --format $FMT $@"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- $B > $D/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- $B > $D/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- $B > $D/write.log 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $D/pmc$i -- $B > $D/pmc$i.log 2>&1
done
python3 tools/profile_summary.py gpurun_out/$NAME.md --stats $D/stats --fetch $D/fetch --write $D/write --pmc $D/pmc1 $D/pmc2 $D/pmc3 $D/pmc4 --cmd "rocprofv3 --kernel-trace --stats -- $B  (PMC: same command, separate --pmc passes)" > /dev/null
find $D -name "*.csv" -size +2M -delete
tail -30 gpurun_out/$NAME.md
