# timing experiment: LZ4 parse-only build (ALZ_QEXP=3) vs the full kernel, with and without sources beyond the LDS window
cd $GRAFT_REPO_ROOT
for q in 3 0; do
  rm -rf auroralib/compression_amd/csrc/_obj
  ALZ_EXTRA_FLAGS="-DALZ_QEXP=$q" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  for md in 0 3000; do
  echo -n "QEXP=$q maxdist=$md lz4 "
  ALZ_SYNTH_MAXDIST=$md python bench.py --no-cpu-baseline --no-verify --inflight 1 --steps 10 --format lz4_block 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
  done
done
