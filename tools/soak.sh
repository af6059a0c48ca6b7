# usage (GPU box): bash tools/soak.sh [first_seed] [count]  -- the malformed-input differential test under other seeds
cd $GRAFT_REPO_ROOT
s0=${1:-5000}; n=${2:-20}
for i in $(seq 0 $((n-1))); do
  seed=$((s0 + 97 * i))
  ALZ_FUZZ_SEED=$seed timeout 300 python -m pytest tests/test_gpu_decode.py -q -m gpu -k "fuzz" 2>&1 | grep -E "^FAILED|AssertionError|passed|failed" | sed "s/^/seed $seed: /" | head -8
done
