"""The bench line the driver parses must stay small and LAST on stdout (round 4's 39 KB line left BENCH_r04.parsed null)."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench.py touches neither HIP nor torch)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def canned(world=1, with_configs=True):
    """A full record shaped like a real run's, with prose as long as the real one's (or longer)."""
    prose = "x" * 400
    roof = {"bound": "hbm", "achieved": 1007.49, "peak": 8000.0, "unit": "GB/s", "frac": 0.12594, "traffic": 3011755181, "traffic_source": prose,
            "kernel_ms": 2.9567, "algorithmic_bytes_per_launch": 2978822689, "issue_frac": 0.4321}
    full = {"metric": "decompressed GiB/s (whole job; 10k x 256KiB batch per GPU), one batch in flight", "value": 824.99, "unit": "GiB/s", "n_gpus": world,
            "steps": 20, "warmup": 3, "ms_per_step": 2.9593, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": prose, "format": "yaz0", "streams_this_rank": 10000, "streams_whole_job": 10000 * world, "stream_bytes": 262144,
                       "compressed_bytes_whole_job": 357382689 * world, "parallelism": prose, "batches_in_flight": 1,
                       "pipelined": {"value": 984.196, "unit": "GiB/s", "ms_per_step": 2.4806, "batches_in_flight": 2, "note": prose},
                       "parity_ok": True, "verified_vs_oracle": True},
            "roofline": roof,
            "cpu_baseline": {"value": 14.639, "unit": "GiB/s", "cores": 32, "kind": "port", "single_thread": 0.922, "threads_sweep_GiB_s": {str(t): 1.0 for t in range(64)},
                             "cpu_model": "AMD EPYC 9575F 64-Core Processor", "cgroup_cpu_quota_cores": 16.0, "sample": prose,
                             "single_stream_one_thread_GiB_s": {f: 0.7 for f in bench.SINGLE_FORMATS}},
            "ranks": [{"rank": r, "local_rank": r, "device": "AMD Radeon Graphics (gfx950:sramecc+:xnack-)", "pid": 100000 + r, "pci_domain_id": 0,
                       "pci_bus_id": 200 + r, "pci_device_id": 0, "uuid": "36343363-6338-3637-6665-646565383062"} for r in range(world)],
            "copy_bandwidth": {"value": 5928.8, "unit": "GB/s", "what": prose}, "end_to_end": {"value": 41.7, "unit": "GiB/s", "what": prose}}
    if with_configs:
        cfgs = []
        names = ["cfg2", "cfg3", "cfg4_shard"] + ["cfg5_%s" % x for x in ("q0", "q8", "q15", "yaz0_q0", "yaz0_q8", "lz4_block_q0", "lz4_block_q8")] + \
                ["body_" + f for f in bench.BODIES] + ["realistic_" + f for f in ("yaz0", "lz10", "lz11", "prs_be", "lz4_block")] + \
                ["realistic_compress_yaz0_q8", "realistic_compress_lz4_block_q8", "mid_batch_yaz0_q8", "extra_one", "extra_two"]
        for nme in names:
            cfgs.append({"name": nme, "workload": prose, "value": 123456.789, "unit": "GiB/s", "steps": 10, "ms_per_step": 12345.6789, "parity_ok": True,
                         "roofline": dict(roof), "cpu_port": {"value": 12.345, "unit": "GiB/s of raw input", "cores": 32, "kind": "port", "sample": prose}})
        for f in bench.SINGLE_FORMATS:
            for q in (0, 15):
                for d in ("single_", "single_compress_"):
                    cfgs.append({"name": "%s%s_q%d" % (d, f, q), "workload": prose, "value": 3.333, "parity_ok": True, "published_managed": {"where": prose}})
        full["configs"] = cfgs
    return full


def test_contract_line_is_small_and_complete():
    for world, wc in ((1, True), (1, False), (8, False), (8, True)):
        line = bench.contract_line(canned(world, wc))
        assert len(line) < bench.CONTRACT_LINE_MAX <= 4096, len(line)
        assert "\n" not in line
        o = json.loads(line)
        for k in REQUIRED:
            assert k in o, k
        assert o["roofline"]["frac"] == 0.12594 and o["roofline"]["traffic"] == 3011755181 and o["roofline"]["bound"] == "hbm"
        assert o["cpu_baseline"]["kind"] == "port" and o["cpu_baseline"]["cores"] == 32 and o["cpu_baseline"]["sample"]
        assert o["config"]["workload"] and "model" not in o["config"]
        assert len(o["ranks"]) == world and len({r["pci_bus_id"] for r in o["ranks"]}) == world
        if wc and world == 1:
            rows = o["configs_summary"]["rows"]
            assert rows["cfg3"][:3] == [123456.789, 2.9567, 0.12594] and rows["cfg5_q8"][-1] == 12.345
            assert o["configs_summary"]["single_stream_GiB_s"]["decode_q15"] == [3.333] * len(bench.SINGLE_FORMATS)
            assert o["configs_summary"]["not_ok"] == []


def test_failed_entries_are_named():
    full = canned()
    full["configs"][2]["parity_ok"] = False
    full["configs"].append({"name": "error", "error": "boom"})
    o = json.loads(bench.contract_line(full))
    assert o["configs_summary"]["not_ok"] == ["cfg4_shard", "error"]


def test_contract_line_is_the_last_stdout_line(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(canned())
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2
    assert lines[0].startswith(bench.DETAIL_PREFIX) and not lines[0].startswith("{")
    assert json.loads(lines[0][len(bench.DETAIL_PREFIX):])["configs"][0]["name"] == "cfg2"       # the full record survives, one line earlier
    assert lines[-1].startswith("{") and len(lines[-1]) < 4096
    assert json.loads(lines[-1])["value"] == 824.99
    assert json.load(open(tmp_path / "bench_detail.json"))["value"] == 824.99
    # a JSON-line consumer that scans from the end finds the contract line first; one that takes every line starting with "{" finds only it
    assert [l for l in lines if l.startswith("{")] == [lines[-1]]


def test_clean_launcher_protocol():
    """tests/_launcher.py (what the -m gpu session uses to start bench.py's ranks from a process that never touched the GPU)."""
    import subprocess
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_launcher.py")], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    try:
        for cmd, want_rc, want_out in (([sys.executable, "-c", "print('hello')"], 0, "hello\n"), ([sys.executable, "-c", "import sys; sys.exit(3)"], 3, ""),
                                       (["/nonexistent/program"], -1, "")):
            p.stdin.write(json.dumps({"cmd": cmd, "env": None, "cwd": ROOT, "timeout": 30}) + "\n"); p.stdin.flush()
            rep = json.loads(p.stdout.readline())
            assert rep["rc"] == want_rc and rep["stdout"] == want_out, rep
    finally:
        p.stdin.close()
        assert p.wait(timeout=10) == 0


def test_issue_object_prices_counts_against_the_measured_ceilings():
    """roofline.issue: committed instruction counts (profiles/insts.json) over the launch's CU cycles, per pipe."""
    r = bench.roofline(2978822689, 2.93, 3011897166, "yaz0:10000:256")
    iss = r["issue"]
    cyc = 2.93e-3 * 2.4e9 * 256
    c = json.load(open(os.path.join(ROOT, "profiles", "insts.json")))["yaz0:10000:256"]
    assert abs(iss["salu"] - c["salu"] / cyc / 0.95) < 2e-3
    assert abs(iss["valu"][0] - c["valu"] / cyc / 1.75) < 2e-3 and abs(iss["valu"][1] - c["valu"] / cyc / 0.95) < 2e-3
    assert abs(iss["lds"] - c["lds"] / cyc * 4.3) < 2e-3
    assert r["issue_frac"] == iss["issue_frac"] and 0.3 < r["issue_frac"] < 1.2
    assert iss["counts_from"].startswith("profiles/r0")
    assert "issue" not in bench.roofline(1e9, 1.0, None, "no_such_workload:1:1") and "issue" not in bench.roofline(1e9, 1.0)


def test_committed_counters_belong_to_the_kernels_at_head():
    """profiles/traffic.json and profiles/insts.json carry a hash of the kernel sources they were measured on (tools/kernel_hash.py, written by tools/update_counters.py on the
    checkout the counters came from): at HEAD it equals the tree's, so `roofline.counters_stale` of the bench line is false -- a kernel change without a re-profile turns this red."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("alz_kernel_hash", os.path.join(ROOT, "tools", "kernel_hash.py"))
    kh = importlib.util.module_from_spec(spec); spec.loader.exec_module(kh)
    cur = kh.current()
    for name in ("traffic.json", "insts.json"):
        assert kh.recorded(name).get("decode") == cur["decode"], (name, kh.recorded(name), cur)
    bench._STALE.clear()
    r = bench.roofline(2978822689, 2.5, bench.measured_traffic("yaz0", 10000, 256), "yaz0:10000:256")
    assert r["counters_stale"] is False and r["traffic"] is not None and "issue" in r
    line = json.loads(bench.contract_line({"metric": "m", "value": 1.0, "unit": "u", "roofline": r, "config": {"workload": "w"}}))
    assert line["roofline"]["counters_stale"] is False
