set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_canary.py -m gpu -x -q -k "lz4 or snappy or LZ4 or SNAPPY or all_formats or encode" 2>&1 | tail -5
for q in 0 8; do timeout 300 python bench.py --mode encode --format lz4_block --quality $q --steps 3 --warmup 1 --no-cpu-baseline --configs none --no-extras 2>&1 | tail -1 | cut -c1-600; done
timeout 300 python bench.py --mode encode --format snappy_raw --quality 0 --steps 3 --warmup 1 --no-cpu-baseline --configs none --no-extras 2>&1 | tail -1 | cut -c1-400
ALZ_MID_Q=0,8 ALZ_MID_N=1024 timeout 300 python tools/mid_batch_encode.py lz4_block snappy_raw 2>&1 | tail -8
