cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_canary.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
ALZ_SEQ_EXPERIMENT_FLAGS="-DALZ_SEQ_TWO_KERNELS none" bash tools/seq_fused_experiment.sh
