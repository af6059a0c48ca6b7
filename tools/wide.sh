# usage (GPU box): bash tools/wide.sh -- the flag-family kernels with groups starting in 128 instead of 64 input bytes per iteration
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_decode.py tests/test_kat.py -m gpu -x -q 2>&1 | tail -2
echo "wide (flat + walk):"; bash tools/ab.sh yaz0 lz11 lz40 lz10 lzss clz0
python bench.py --no-cpu-baseline --no-extras --configs realistic --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['configs'][0]; print('realistic', c['value'], c['roofline']['kernel_ms'], c['parity_ok'])"
for fl in "-DALZ_WIDE_WALK=0" "-DALZ_WIDE_WALK=0 -DALZ_WIDE_FLAT=0"; do
  rm -rf auroralib/compression_amd/csrc/_obj; ALZ_EXTRA_FLAGS="$fl" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "$fl:"; bash tools/ab.sh yaz0 lz11 lz10 lzss
  python bench.py --no-cpu-baseline --no-extras --configs realistic --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['configs'][0]; print('realistic', c['value'], c['roofline']['kernel_ms'], c['parity_ok'])"
done
