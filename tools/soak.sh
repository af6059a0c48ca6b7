# usage (GPU box): bash tools/soak.sh [first_seed] [count]  -- the malformed-input differential tests under other seeds: the host-buffer
# parity test (test_gpu_decode.py, fuzz) and the device-resident canary tests (test_gpu_canary.py: whole destination buffer compared,
# both kernel families, both wave shapes), the whole-GPU decode of ONE mutated stream (test_gpu_big_stream.py) and the whole-GPU ENCODE of
# random inputs (test_gpu_big_encode.py), the encoder's scan path forced over random inputs (test_gpu_scan_encode.py, round 6) and the work queue's randomly cut / mutated streams (test_gpu_chunks.py)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
s0=${1:-5000}; n=${2:-20}
for i in $(seq 0 $((n-1))); do
  seed=$((s0 + 97 * i))
  ALZ_FUZZ_SEED=$seed timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_canary.py tests/test_gpu_big_stream.py tests/test_gpu_big_encode.py tests/test_gpu_mid_encode.py tests/test_gpu_scan_encode.py tests/test_gpu_chunks.py -q -m gpu -k "fuzz or canary_lzss or early_or_late or real_data_and_garbage" 2>&1 | grep -E "^FAILED|AssertionError|passed|failed" | sed "s/^/seed $seed: /" | head -8
done
