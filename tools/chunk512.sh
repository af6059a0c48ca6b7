# usage (GPU box): bash tools/chunk512.sh -- flag-family kernels with 512-byte input-cache chunks (31 instead of 25 waves per CU)
cd $GRAFT_REPO_ROOT
for fl in "-DALZ_FAST_CHUNK=512" "-DALZ_FAST_CHUNK=256"; do
  rm -rf auroralib/compression_amd/csrc/_obj; ALZ_EXTRA_FLAGS="$fl" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "$fl:"; bash tools/ab.sh yaz0 lz10
  python bench.py --no-cpu-baseline --no-extras --configs realistic --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['configs'][0]; print('realistic', c['value'], c['roofline']['kernel_ms'], c['parity_ok'])"
done
