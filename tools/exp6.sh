# read-back batch depth (steps mapped before the copies start) for the queue formats
cd $GRAFT_REPO_ROOT
for nb in 12 16 8; do
  rm -rf auroralib/compression_amd/csrc/_obj
  ALZ_EXTRA_FLAGS="-DALZ_NB=$nb" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  for f in lz4_block snappy_raw lzo; do
  echo -n "NB=$nb $f "
  python bench.py --no-cpu-baseline --steps 10 --format $f 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['back_to_back']['value'], d['roofline']['kernel_ms'], d['config']['parity_ok'])"
  done
done
