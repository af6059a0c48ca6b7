# usage (GPU box): bash tools/enc_split.sh "SPLIT ACHUNK" ...  -- cfg5 (LZSS compression, 10 000 x 256 KiB) kernel time against the
# wavefronts per stream of kernel A (ALZ_ENC_SPLIT) and the streams per pass of kernel A (ALZ_ENC_ACHUNK, 0 = all)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  sp=${cfg% *}; ac=${cfg#* }
  echo -n "split $sp achunk $ac: "
  ALZ_ENC_SPLIT=$sp ALZ_ENC_ACHUNK=$ac python bench.py --no-cpu-baseline --no-extras --no-verify --inflight 1 --steps 3 --configs cfg5 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print(' | '.join('%s %.1f ms %.1f GiB/s ok=%s' % (c['name'], c.get('kernel_ms',0), c.get('value',0), c.get('parity_ok')) for c in d['configs']))"
done
