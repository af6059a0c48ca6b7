# BASELINE.json configs[2] at full size (100 000 LZ4 blocks x 256 KiB) and one GPU's shard of configs[3] (5 000 mixed streams)
cd $GRAFT_REPO_ROOT
for args in "--format lz4_block --streams 100000 --steps 5 --warmup 1" "--format mixed --streams 5000 --steps 20" "--format yaz0 --stream-kib 64 --steps 20"; do
  echo -n "$args : "
  timeout 1500 python bench.py --no-cpu-baseline --no-verify $args 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['back_to_back']['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['parity_ok'])"
done
