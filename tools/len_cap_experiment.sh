# usage (GPU box): bash tools/len_cap_experiment.sh [quality]  -- kernel B's compare cap (ALZ_LEN_CAP) against the real-data and the synthetic batch
cd $GRAFT_REPO_ROOT
q=${1:-8}
for cap in 2040 256 96 48; do
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="-DALZ_LEN_CAP=$cap" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== ALZ_LEN_CAP $cap"
  ALZ_MID_Q=$q ALZ_MID_N=2048 timeout 600 python tools/mid_batch_encode.py prs_be lz4_block lzo snappy_raw 2>&1 | grep -v amdgpu
  python bench.py --mode encode --format lz4_block --quality $q --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('synthetic lz4_block', d['value'], d['ms_per_step'])"
done
touch auroralib/compression_amd/csrc/alz_encode.hip; bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
