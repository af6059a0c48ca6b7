set -e
cd $GRAFT_REPO_ROOT
for e in 0 3 4 5 6 7; do
  rm -rf auroralib/compression_amd/csrc/_obj
  ALZ_EXTRA_FLAGS="-DALZ_EXP=$e" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "EXP=$e"
  python bench.py --no-cpu-baseline --no-verify --streams 16384 --steps 5 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
done
