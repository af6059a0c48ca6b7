# timing experiment: PRS parse-only build (ALZ_QEXP=2) vs the full kernel
cd $GRAFT_REPO_ROOT
for q in 2 0; do
  rm -rf auroralib/compression_amd/csrc/_obj
  ALZ_EXTRA_FLAGS="-DALZ_QEXP=$q" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo -n "QEXP=$q prs_be "
  python bench.py --no-cpu-baseline --no-verify --inflight 1 --steps 10 --format prs_be 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
done
