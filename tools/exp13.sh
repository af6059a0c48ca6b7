# input-cache chunk of the 4 KiB-window lane-parallel kernels: 512 bytes (28 waves per CU) against 1 KiB (24)
cd $GRAFT_REPO_ROOT
for ch in 512 1024; do
  rm -rf auroralib/compression_amd/csrc/_obj
  ALZ_EXTRA_FLAGS="-DALZ_FAST_CHUNK=$ch" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  for f in yaz0 lz10 lz11 lzss clz0 lz40 lz02 lzhudson; do
  echo -n "CHUNK=$ch $f "
  python bench.py --no-cpu-baseline --steps 20 --format $f 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['back_to_back']['value'], d['roofline']['kernel_ms'], d['config']['parity_ok'])"
  done
done
