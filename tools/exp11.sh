# timing experiment: parse-only build (ALZ_QEXP=3: pipelined_rounds without the byte phase) for the round-parser formats
cd $GRAFT_REPO_ROOT
for q in 3 0; do
  rm -rf auroralib/compression_amd/csrc/_obj
  ALZ_EXTRA_FLAGS="-DALZ_QEXP=$q" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  for f in lzo snappy_raw lz4_block lzshrek cns; do
  echo -n "QEXP=$q $f "
  python bench.py --no-cpu-baseline --no-verify --inflight 1 --steps 10 --format $f 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
  done
done
