# usage (GPU box): bash tools/quick_bench.sh "cfg list" [pytest -k expr]  -- decode parity tests (optionally filtered), then bench.py with the given configs; one line per entry
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_canary.py tests/test_kat.py -m gpu -q -x ${2:+-k "$2"} 2>&1 | tail -2
timeout 900 python bench.py --no-cpu-baseline --no-extras --configs "${1:-cfg2,cfg4,bodies,realistic}" --steps 20 > gpurun_out/qb.json 2>gpurun_out/qb.err
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/qb.json").read().splitlines() if l.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])   # (the full record; the last line is the small contract line)
print("yaz0 %.1f GiB/s  %.4f ms/step  kernel %.4f  pipelined %.1f  ok %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], (d["config"]["pipelined"] or {}).get("value", 0), d["config"]["parity_ok"]))
for c in d.get("configs") or []: print("%-22s %8.1f  %8s ms  ok %s %s" % (c.get("name"), c.get("value") or 0, c.get("ms_per_step", c.get("kernel_ms")), c.get("parity_ok"), c.get("error", "")))
PY
