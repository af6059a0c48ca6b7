# usage (GPU box): bash tools/profile_r04_encode.sh  -- rocprofv3 kernel stats of the batch encoder at the end of round 4: the synthetic cfg5
# batch (10 000 x 256 KiB, quality 0 / 8 / 15) and 1 024 windows of Test.bmp (tools/mid_batch_encode.py, quality 0 / 8) -> gpurun_out/r04_encode.md
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04_encode.md; mkdir -p $GRAFT_REPO_ROOT/gpurun_out
echo "# r04_encode -- the batch encoder, rocprofv3 --kernel-trace --stats (microseconds)" > $out
run() {   # title, command...
  local title="$1"; shift
  rm -rf /tmp/pe; "$@" > /tmp/pe.log 2>&1
  f=$(find /tmp/pe -name "*kernel_stats.csv" | head -1)
  echo -e "\n## $title\n\n\`$CMD\`\n\n| kernel | calls | average us | total ms | % |\n|---|---|---|---|---|" >> $out
  python3 - $f >> $out <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    if n.startswith("__amd") or float(r["Percentage"]) < 0.3: continue
    print("| `%s` | %s | %.1f | %.2f | %.1f |" % (n[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, float(r["Percentage"])))
PY
}
for q in 0 8 15; do
  CMD="python3 bench.py --mode encode --quality $q --steps 3 --warmup 1 --no-cpu-baseline"
  run "cfg5 (synthetic, LZSS) at quality $q" rocprofv3 --kernel-trace --stats -d /tmp/pe -o e --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --mode encode --quality $q --steps 3 --warmup 1 --no-cpu-baseline
done
for f in lz4_block prs_be; do
  CMD="python3 bench.py --mode encode --format $f --quality 0 --steps 3 --warmup 1 --no-cpu-baseline --configs none --no-extras"
  run "cfg5 (synthetic) as $f at quality 0" rocprofv3 --kernel-trace --stats -d /tmp/pe -o e --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --mode encode --format $f --quality 0 --steps 3 --warmup 1 --no-cpu-baseline --configs none --no-extras
done
export ALZ_MID_N=1024
for q in 0 8; do
  export ALZ_MID_Q=$q
  CMD="ALZ_MID_N=1024 ALZ_MID_Q=$q python3 tools/mid_batch_encode.py yaz0 lz4_block   (four calls per format)"
  run "1 024 windows of Test.bmp as Yaz0 and as LZ4 blocks at quality $q" rocprofv3 --kernel-trace --stats -d /tmp/pe -o e --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/mid_batch_encode.py yaz0 lz4_block
done
tail -30 $out
