# LDS window of the 64 KiB-format queue kernels: 2 KiB (more waves per CU) / 4 KiB (default) / 8 KiB (fewer read-backs)
cd $GRAFT_REPO_ROOT
for lw in 2048 8192 4096; do
  rm -rf auroralib/compression_amd/csrc/_obj
  ALZ_EXTRA_FLAGS="-DALZ_QUEUE_LW=$lw" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  for f in lz4_block snappy_raw lzo; do
  echo -n "LW=$lw $f "
  python bench.py --no-cpu-baseline --steps 10 --format $f 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['back_to_back']['value'], d['roofline']['kernel_ms'], d['config']['parity_ok'])"
  done
done
