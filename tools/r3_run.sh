# usage (GPU box): bash tools/r3_run.sh  -- round-3 work list: parity of what changed, then timings
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -8
bash tools/ab.sh yaz0 lz02 prs_be lz11
for q in 0 8; do bash tools/enc_kernels.sh lzss $q; done
python bench.py --mode encode --quality 8 --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-1500
