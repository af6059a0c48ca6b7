# usage (GPU box): bash tools/enc_profile.sh  -- kernel stats of the compression path (cfg5) at Q0, Q8 and Q15 -> gpurun_out/r03_encode.md
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r03_encode.md
echo "# rocprofv3 summary: compression path (cfg5: LZSS(12,4,2), 10 000 x 256 KiB)" > $OUT
for q in 0 8 15; do
  D=gpurun_out/prof_enc_q$q; rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/bench_encode.py --quality $q --reps 1 > $D/log.txt 2>&1
  python3 tools/profile_summary.py $D/sum.md --stats $D --cmd "rocprofv3 --kernel-trace --stats -- python3 tools/bench_encode.py --quality $q --reps 1" > /dev/null
  echo >> $OUT; echo "## quality $q" >> $OUT; echo >> $OUT; echo '`'"$(grep '^{' $D/log.txt | tail -1)"'`' >> $OUT; echo >> $OUT
  grep -E "^command|^\| " $D/sum.md >> $OUT
  find $D -name "*.csv" -size +2M -delete
done
cat $OUT | cut -c1-220
