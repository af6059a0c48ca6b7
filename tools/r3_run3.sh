cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_encode.py -q -m gpu -x 2>&1 | tail -3
for q in 8 15; do echo "== enc lzss q$q"; bash tools/enc_kernels.sh lzss $q; done
