cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 2400 python -m pytest tests/test_gpu_encode.py tests/test_gpu_containers.py tests/test_gpu_multi.py -q -m gpu 2>&1 | tail -8
for q in 0 8 15; do echo "== enc lzss q$q"; bash tools/enc_kernels.sh lzss $q; done
