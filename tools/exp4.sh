# usage: bash tools/exp4.sh  -- queue formats with the match distance capped below the LDS window (cost of the HBM read-back path)
cd $GRAFT_REPO_ROOT
for md in 0 3000 16000; do
for f in lz4_block snappy lzo; do
  ALZ_SYNTH_MAXDIST=$md timeout 300 python bench.py --no-cpu-baseline --format $f --steps 10 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('maxdist $md $f', d['value'], 'GiB/s kernel_ms', d['roofline']['kernel_ms'], 'b2b', d['config'].get('back_to_back'), 'ok', d['config']['parity_ok'])"
done; done
