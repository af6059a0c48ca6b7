# usage (GPU box): bash tools/exp.sh "0 1 2" [bench args]   -- rebuilds with -DALZ_EXP=n and prints value / back-to-back / kernel ms
cd $GRAFT_REPO_ROOT
EXPS=$1; shift
for e in $EXPS; do
  rm -rf auroralib/compression_amd/csrc/_obj
  ALZ_EXTRA_FLAGS="-DALZ_EXP=$e" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo -n "EXP=$e  "
  python bench.py --no-cpu-baseline --steps 20 $@ 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['back_to_back']['value'], d['roofline']['kernel_ms'], d['config']['parity_ok'])"
done
rm -rf auroralib/compression_amd/csrc/_obj
bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
