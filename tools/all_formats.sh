# usage (GPU box): bash tools/all_formats.sh  -- every body format, 10 000 x 256 KiB: GiB/s one batch in flight / two in flight / kernel ms / parity
cd $GRAFT_REPO_ROOT
bash tools/ab.sh yay0 yaz0 lz02 lz11 lz40 mio0 smsr00 lzss lz10 clz0 hig cns wflz wflz_be lzhudson lzshrek lz4_block fastlz cnx2 refpack blz snappy_raw lzo prs_be prs_le mixed
