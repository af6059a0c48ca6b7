cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
echo "== ab"; bash tools/ab.sh lz4_block snappy_raw fastlz cns hig wflz refpack cnx2
for q in 0 8; do echo "== enc lzss q$q"; bash tools/enc_kernels.sh lzss $q | grep roles; done
