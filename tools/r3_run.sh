cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -5
bash tools/ab.sh lzo lz4_block snappy_raw fastlz cns hig wflz refpack mixed
bash tools/cfg4_quick.sh
for q in 0 8; do bash tools/enc_kernels.sh lzss $q; done
