# usage (GPU box): bash tools/profile_r06.sh [fmt...]  -- round-6 evidence: everything native is built FIRST (no compiler is started from a process that has touched the GPU), then per
# format tools/gpu_profile.sh (kernel stats + separate PMC passes, every rocprofv3 call under `timeout`) -> gpurun_out/r06_<fmt>.md, then the named configurations
# (cfg2, the cfg4 shard, cfg3 at 100 000 blocks): kernel stats + FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/r06_cfg*.md.  tools/update_counters.py (run on the checkout
# the counters came from) puts the figures into profiles/traffic.json / insts.json and stamps them with the kernel sources' hash.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1 || exit 1
for f in ${@:-yaz0 lz10 lz11 yay0 mio0 lzss prs_be lz4_block lzo snappy_raw mixed}; do
  timeout 2400 bash tools/gpu_profile.sh r06_$f $f > /dev/null 2>&1
  echo "== $f"; grep -E "alz_decode|corrected bytes|SQ_INSTS" gpurun_out/r06_$f.md | cut -c1-200
done
while read key fmt n kib extra; do
  D=gpurun_out/prof_r06_$key; rm -rf "$D"; mkdir -p "$D"
  B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-extras --configs none --inflight 1 --format $fmt --streams $n --stream-kib $kib $extra"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- $B > $D/stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- $B > $D/fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- $B > $D/write.log 2>&1
  python3 tools/profile_summary.py gpurun_out/r06_$key.md --stats $D/stats --fetch $D/fetch --write $D/write --cmd "rocprofv3 --kernel-trace --stats -- $B  (FETCH_SIZE / WRITE_SIZE: same command, separate --pmc passes)" > /dev/null
  find "$D" -name "*.csv" -size +2M -delete
  echo "== $key"; grep -E "alz_decode|corrected bytes" gpurun_out/r06_$key.md | cut -c1-200
done <<LIST
cfg2 yaz0 10000 64
cfg4_shard mixed 5000 256
cfg3_100000 lz4_block 100000 256
LIST
