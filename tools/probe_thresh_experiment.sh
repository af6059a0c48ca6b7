# usage (GPU box): bash tools/probe_thresh_experiment.sh  -- the synthetic cfg5 batch with EVERY stream on the one-position-per-lane kernels
# (-DALZ_PROBE_THRESH16=0) against the default (the probe sends synthetic streams to the two-phase kernel)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for thr in 4 0; do
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="-DALZ_PROBE_THRESH16=$thr" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== ALZ_PROBE_THRESH16 $thr"
  for q in 4 8 12 15; do
    python bench.py --mode encode --quality $q --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 synthetic lzss q$q', d['value'], d['ms_per_step'])"
  done
  python bench.py --mode encode --format yaz0 --quality 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 synthetic yaz0 q8', d['value'], d['ms_per_step'])"
done
touch auroralib/compression_amd/csrc/alz_encode.hip; bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
