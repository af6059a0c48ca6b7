# usage (GPU box): bash tools/len_cap_experiment.sh  -- kernel B's compare cap (ALZ_LEN_CAP) against the real-data and the synthetic batch at quality 8
cd $GRAFT_REPO_ROOT
for cap in 2040 256 64; do
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="-DALZ_LEN_CAP=$cap" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== ALZ_LEN_CAP $cap"
  ALZ_MID_Q=8 ALZ_MID_N=1024 timeout 600 python tools/mid_batch_encode.py yaz0 lz11 lz4_block 2>&1 | grep -v amdgpu
  python bench.py --mode encode --quality 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 synthetic q8', d['value'], d['ms_per_step'])"
done
touch auroralib/compression_amd/csrc/alz_encode.hip; bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
