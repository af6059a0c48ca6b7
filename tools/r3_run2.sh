cd $GRAFT_REPO_ROOT
echo "== ab"; bash tools/ab.sh lzo snappy_raw fastlz cns lz4_block
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_encode.py -q -m gpu -x 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_decode.py -q -m gpu -k "lzo or snappy or fastlz or cns" 2>&1 | tail -3
for q in 8 15; do echo "== enc lzss q$q"; bash tools/enc_kernels.sh lzss $q; done
echo "== enc prs q8"; bash tools/enc_kernels.sh prs_be 8
