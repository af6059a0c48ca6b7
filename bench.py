#!/usr/bin/env python3
"""bench.py -- the hot path's headline measurement (BASELINE.json metric) and the named configurations.

A "step" is one pass of the batched decode over one synthetic batch whose compressed payload is already resident in
HBM; output stays in HBM.  Headline workload = the batch the metric is quoted on: 10 000 Yaz0 streams x 256 KiB per GPU
(`--stream-kib 64` gives BASELINE.json configs[1] exactly).  `value` is measured with ONE batch in flight -- K steps back to
back on one HIP stream, the execution the `roofline` object and the rocprofv3 summaries under profiles/ describe; the
figure with two batches in flight (the tail of one launch overlapping the head of the next) is reported next to it as
`config.pipelined`.

On one GPU the same JSON line also carries (rank 0, skipped with --configs none):
  configs       BASELINE.json configs[1..4] at their stated sizes (cfg2 Yaz0 10 000 x 64 KiB, cfg3 LZ4 100 000 x 256 KiB, one
                GPU's shard of cfg4 = 5 000 mixed LZ10/LZ11/Yaz0/PRS streams, cfg5 compression as LZSS at Q0 / Q8 / Q15 and as
                Yaz0 / LZ4 at Q0 / Q8), every other decode body of north_star on the headline's shape (body_<format>), ONE 1 000 KiB
                stream of Test.bmp as Yaz0 / Yay0 / MIO0 / LZ10 / LZ11 / LZSS / PRS / LZO / LZ4 (the reference's own benchmark shape: single_<format>_q<Q>), and the
                "realistic" data set of SURVEY.md 8d (the 256 KiB windows of the reference's Test.bmp, GPU-encoded), each
                with its own roofline object from HIP events on the launch stream
  copy_bandwidth  a measured device-to-device copy (second roofline denominator)
  end_to_end    the same batch through the host-buffer ABI (upload + decode + download, pinned staging)
  cpu_baseline  the C restatement of the managed path on the host cores: T = 1 and T = the cores this process may use

Multi-GPU: one process per GPU (`python -m torch.distributed.run ... bench.py --gpus N`; `python bench.py --gpus N` alone
starts exactly that as a child process).  --scaling weak (default): every rank decodes its own batch of --streams streams.
--scaling strong: ONE batch of --streams streams is partitioned over the ranks by alz_partition_batch (greedy LPT, the
library's host-side partitioner) -- `--scaling strong --format mixed --streams 40000` is BASELINE.json configs[3].  Either
way there is no data-path collective; value = bytes decoded by all ranks / max time, and every rank checks its first
1 024 streams byte for byte against the CPU restatement.

Rank 0 prints the full record as a `BENCH_DETAIL {...}` line (also written to bench_detail.json) and then, LAST, the small contract
line (< 4 KB: `contract_line`) -- the one JSON line the driver parses.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MIXED = ["lz10", "lz11", "yaz0", "prs_be"]
BODIES = ["lzss", "lz10", "lz11", "yay0", "mio0", "prs_be", "lz4_block", "lzo", "snappy_raw"]   # (yaz0 is the headline itself)


def measured_traffic(fmt, n, kib):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, gfx950 rule) for this
    exact workload, or None when it has not been profiled (profiles/traffic.json)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return t.get("%s:%d:%d" % (fmt, n, kib))
    except Exception:
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["decode", "encode"], default="decode",
                    help="decode: the headline metric.  encode: BASELINE.json configs[4] (compression, 1 -> 8 GPUs) -- every rank compresses its own "
                         "raw buffers, resident in HBM, through alz_encode_batch_device")
    ap.add_argument("--quality", type=int, default=0, help="--mode encode: CompressionSettings.Quality (0 / 15 are the reference's published levels, 8 its default)")
    ap.add_argument("--format", default=None, help="default: yaz0 (decode), lzss (encode)")
    ap.add_argument("--streams", type=int, default=10000, help="streams per GPU (weak scaling) or in the whole batch (strong scaling)")
    ap.add_argument("--stream-kib", type=int, default=256)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--configs", default="all", help="all | none | comma list of cfg2,cfg3,cfg4,cfg5,realistic (N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the byte comparison with the CPU restatement")
    ap.add_argument("--no-extras", action="store_true", help="skip copy bandwidth and the end-to-end (host buffer) run")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight of the secondary, pipelined figure (1 = do not measure it)")
    ap.add_argument("--kernel-variant", type=int, default=0, help="alz_ctx_set_kernel_variant: 0 the library chooses, 1 / 2 one / two wavefronts per stream where both shapes exist (tuning)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests of the N>1 path)")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1, help="smoke test: every rank uses this GPU (needs --dist-backend gloo)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the one-process-per-GPU job as a CHILD (never an exec: nothing
    here has touched HIP yet, and a process that has must not replace itself) and relay rank 0's line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    for l in p.stdout.splitlines():
        if l.startswith("BENCH_DETAIL "):
            print(l)
    if lines:
        print(lines[-1])
    sys.exit(p.returncode if p.returncode else (0 if lines else 4))


def gather_rank_identities(ctx, torch, dist, rank, local_rank, world):
    """Which GPU every rank ran on (device name, PCI bus id, uuid where the runtime reports one): a multi-GPU record must show N
    DISTINCT devices.  Gathered on every rank (collective), reported by rank 0."""
    me = {"rank": rank, "local_rank": local_rank, "device": ctx.info()["name"], "pid": os.getpid()}
    try:
        p = torch.cuda.get_device_properties(local_rank)
        for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"):
            if hasattr(p, k):
                me[k] = int(getattr(p, k))
        if hasattr(p, "uuid"):
            me["uuid"] = str(p.uuid)
    except Exception as e:
        me["props_error"] = repr(e)
    if dist is None:
        return [me]
    allr = [None] * world
    dist.all_gather_object(allr, me)
    return allr


def fmt_array(np, A, name, n, first=0):
    if name == "mixed":   # BASELINE.json configs[3]: LZ10/LZ11/Yaz0/PRS interleaved, per-format kernel dispatch
        return np.array([A.FORMAT_NAMES.index(MIXED[(first + i) % 4]) for i in range(n)], dtype=np.uint32)
    return np.full(n, A.FORMAT_NAMES.index(name), dtype=np.uint32)


TRAFFIC_SOURCE = "profiles/traffic.json (rocprofv3 --pmc passes of this exact workload, FETCH_SIZE x2 + WRITE_SIZE per launch, committed; not re-measured in this run)"


# The instruction-issue ceilings of one CU, measured (tools/ubench_issue.hip, profiles/r05_issue_ceiling.md): wave64 instructions per cycle per CU
ISSUE_CEILING = {"valu_fast": 1.75, "valu_slow": 0.95, "salu": 0.95, "lds_cycles_per_inst": 4.3, "clock_hz": 2.4e9, "cus": 256}     # (clock / CUs: an MI355X; main() takes them from the device it runs on)


def set_device_shape(torch, device_index):
    """CU count and shader clock of the device the bench runs on, for `roofline.issue` (the per-CU ceilings themselves were measured on an MI355X)."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        if getattr(p, "multi_processor_count", 0):
            ISSUE_CEILING["cus"] = int(p.multi_processor_count)
        khz = getattr(p, "clock_rate", 0)
        if khz:
            ISSUE_CEILING["clock_hz"] = float(khz) * 1e3
    except Exception:
        pass


def issue_object(key, kernel_ms):
    """`roofline.issue`: the launch's instruction counts (profiles/insts.json: committed rocprofv3 --pmc passes of this exact workload) priced against the
    MEASURED issue ceilings -- the share of each pipe's capacity the launch used.  valu: [all instructions in the 2-cycle class, all in the 4-cycle class];
    salu: the CU's one scalar pipe; lds: the CU's one LDS pipeline at the byte phase's average of 4.3 pipeline cycles per instruction (an estimate).
    `issue_frac` = the most loaded pipe, the vector ALU priced half fast, half slow."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "insts.json"))).get(key)
    except Exception:
        c = None
    if not c:
        return None
    cyc = kernel_ms * 1e-3 * ISSUE_CEILING["clock_hz"] * ISSUE_CEILING["cus"]
    v, sc, l = c["valu"] / cyc, c["salu"] / cyc, c["lds"] / cyc
    valu = [round(v / ISSUE_CEILING["valu_fast"], 3), round(v / ISSUE_CEILING["valu_slow"], 3)]
    salu = round(sc / ISSUE_CEILING["salu"], 3)
    lds = round(l * ISSUE_CEILING["lds_cycles_per_inst"], 3)
    mid = v / (2.0 / (1.0 / ISSUE_CEILING["valu_fast"] + 1.0 / ISSUE_CEILING["valu_slow"]))
    return {"valu": valu, "salu": salu, "lds": lds, "issue_frac": round(max(mid, salu, lds), 3), "counts_from": c.get("source")}


def _kernel_hash_tool():
    """tools/kernel_hash.py as a module (tools/ is not a package): the hash that ties profiles/traffic.json / insts.json to the kernel sources they were measured on."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("alz_kernel_hash", os.path.join(ROOT, "tools", "kernel_hash.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


_STALE = {}


def counters_stale(family):
    """True when the committed counter files were collected on OTHER kernel sources than this checkout's (`_kernel_hash` of profiles/traffic.json and insts.json against
    sha256 over csrc's kernel files, tools/kernel_hash.py): `traffic` / `issue` of the line are then figures of an earlier kernel.  family: "decode" | "encode"."""
    if family not in _STALE:
        try:
            kh = _kernel_hash_tool()
            _STALE[family] = bool(kh.stale("traffic.json", family) or (family == "decode" and kh.stale("insts.json", family)))
        except Exception:
            _STALE[family] = True
    return _STALE[family]


def roofline(algo_bytes, kernel_ms, traffic=None, issue_key=None, family="decode"):
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
         "traffic": traffic, "traffic_source": TRAFFIC_SOURCE if traffic is not None else None,
         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": int(algo_bytes)}
    iss = issue_object(issue_key, kernel_ms) if issue_key else None
    if iss:
        r["issue"] = iss
        r["issue_frac"] = iss["issue_frac"]
    if traffic is not None or iss:
        r["counters_stale"] = counters_stale(family)
    return r


CONTRACT_LINE_MAX = 4096    # the driver keeps ~8 KB of stdout tail: the round-4 line (39 KB) left BENCH_r04.parsed null
SINGLE_FORMATS = ["yaz0", "yay0", "mio0", "lz10", "lz11", "lzss", "prs_be", "lzo", "lz4_block", "snappy_raw"]
DETAIL_PREFIX = "BENCH_DETAIL "


def contract_line(full):
    """The ONE small JSON object the driver parses, built from the full record: the contract's keys, `roofline`, `cpu_baseline`, one
    short entry per rank and a fixed-size `configs_summary` (name -> [value, ms per launch, roofline.frac]); no prose, no nested blobs.
    Everything else stays in the detail record (printed as an EARLIER line and written to bench_detail.json)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full[k] for k in keep if k in full}
    c = full.get("config") or {}
    cfg = {k: c[k] for k in ("workload", "mode", "format", "quality", "streams_this_rank", "streams_whole_job", "stream_bytes", "compressed_bytes_whole_job",
                             "ratio", "batches_in_flight", "parity_ok", "verified_vs_oracle", "verified_roundtrip_and_vs_oracle") if k in c}
    cfg["workload"] = str(cfg.get("workload", ""))[:200]
    cfg["parallelism"] = str(c.get("parallelism", ""))[:96]
    if c.get("pipelined"):
        cfg["pipelined"] = {k: c["pipelined"][k] for k in ("value", "ms_per_step", "batches_in_flight") if k in c["pipelined"]}
    out["config"] = cfg
    r = full.get("roofline") or {}
    out["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_bytes_per_launch",
                                         "issue_frac", "counters_stale") if k in r}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "single_thread", "cpu_model", "cgroup_cpu_quota_cores") if k in cb}
        out["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    else:
        out["cpu_baseline"] = None
    out["ranks"] = [{k: x[k] for k in ("rank", "local_rank", "device", "pci_domain_id", "pci_bus_id", "pci_device_id", "uuid") if k in x} for x in (full.get("ranks") or [])][:16]
    for x in out["ranks"]:
        x["device"] = str(x.get("device", ""))[:48]
    for k in ("copy_bandwidth", "end_to_end"):
        if isinstance(full.get(k), dict) and "value" in full[k]:
            out[k] = {"value": full[k]["value"], "unit": full[k].get("unit")}
    configs = full.get("configs")
    if configs:
        summ, single, bad = {}, {}, []
        for e in configs:
            name = e.get("name", "?")
            if e.get("parity_ok") is False or name == "error":
                bad.append(name)
            if name.startswith("single_"):
                # single_<format>_q<Q> / single_compress_<format>_q<Q>: one row per (direction, quality) in SINGLE_FORMATS order
                comp = name.startswith("single_compress_")
                body = name[len("single_compress_" if comp else "single_"):]
                f, q = body.rsplit("_q", 1)
                row = single.setdefault(("compress_q" if comp else "decode_q") + q, [None] * len(SINGLE_FORMATS))
                if f in SINGLE_FORMATS:
                    row[SINGLE_FORMATS.index(f)] = e.get("value")
                continue
            rf = e.get("roofline") or {}
            ms = rf.get("kernel_ms", e.get("kernel_ms", e.get("ms_per_step")))
            row = [e.get("value"), ms, rf.get("frac")]
            if "issue_frac" in rf:
                row.append(rf["issue_frac"])
            if e.get("cpu_port"):
                row.append(e["cpu_port"].get("value"))
            summ[name] = row
        out["configs_summary"] = {"columns": ["value (GiB/s)", "kernel ms per launch", "roofline.frac", "(issue_frac)", "(cpu port GiB/s, T = cores)"],
                                  "rows": summ, "single_stream_GiB_s": {"formats": SINGLE_FORMATS, **single}, "not_ok": bad}
    out["detail"] = "bench_detail.json + the '%s' stdout line before this one" % DETAIL_PREFIX.strip()
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= CONTRACT_LINE_MAX:          # never let the line outgrow the driver's window again: drop the optional parts, largest first
        for k in ("configs_summary", "ranks", "end_to_end", "copy_bandwidth"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) < CONTRACT_LINE_MAX:
                break
    assert len(line) < CONTRACT_LINE_MAX, len(line)
    return line


def emit(full):
    """Rank 0's output: the full record first (one 'BENCH_DETAIL {...}' line, and bench_detail.json beside this file / under gpurun_out/
    when that exists), then -- LAST on stdout -- the small contract line."""
    detail = json.dumps(full)
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
        except OSError:
            pass
    sys.stdout.write(DETAIL_PREFIX + detail + "\n")
    sys.stdout.write(contract_line(full) + "\n")
    sys.stdout.flush()


class DeviceBatch:
    """A batch resident in HBM: compressed payload, output buffer, plan."""

    def __init__(self, ctx, batch, Plan):
        self.ctx, self.batch = ctx, batch
        self.d_src = ctx.malloc(batch.src.nbytes + 64)
        self.d_dst = ctx.malloc(batch.dst_bytes + 64)
        ctx.h2d(self.d_src, batch.src)
        ctx.memset(self.d_dst, 0, batch.dst_bytes)
        self.plan = Plan(ctx, batch.streams)

    def close(self):
        self.plan.close(); self.ctx.free(self.d_src); self.ctx.free(self.d_dst)


def run_steps(db, steps, warmup, sync):
    """W untimed launches, then EXACTLY K launches back to back on the context's stream between two `sync()`s.  Returns (wall seconds
    of the K steps, mean device milliseconds per step): the HIP events that give the second are recorded on the launch stream right
    around the same K launches (alz_plan_execute_timed), inside the wall-clock window -- one loop, two clocks."""
    for _ in range(warmup):
        db.plan.execute(db.d_src, db.d_dst)
    sync()
    t0 = time.perf_counter()
    kernel_ms = db.plan.execute_timed(db.d_src, db.d_dst, iters=steps)
    sync()
    return time.perf_counter() - t0, kernel_ms


def best_of_two(db, steps, sync):
    """The secondary configurations (not the headline, whose K steps are timed exactly once): two passes of K steps, the better wall
    clock reported with its own device time -- one pass in eight showed a host-side stall of tens of milliseconds between two launches
    (wall 14.5 ms per step around HIP events that measured 2.5) right after the previous configuration's buffers had been freed."""
    a = run_steps(db, steps, 2, sync)
    b = run_steps(db, steps, 0, sync)
    return a if a[0] <= b[0] else b


def decode_config(name, workload, ctx, batch, Plan, synth, np, steps, fmt_name, n, kib):
    """One named decode configuration on rank 0: K steps back to back + HIP-event kernel time + status / length check."""
    db = DeviceBatch(ctx, batch, Plan)
    try:
        dt, kernel_ms = best_of_two(db, steps, ctx.synchronize)
        res = synth.result_records(db.plan.results())
        recs = synth.stream_records(batch.streams)
        ok = bool((res["status"] == 0).all() and (res["dst_len"] == recs["decom_len"]).all())
        comp = int(recs["src_len"].astype(np.int64).sum()); dec = int(recs["decom_len"].astype(np.int64).sum())
        return {"name": name, "workload": workload, "value": round(dec * steps / dt / 2**30, 3), "unit": "GiB/s", "steps": steps,
                "ms_per_step": round(dt / steps * 1e3, 4), "timing": "best of two passes of %d steps" % steps, "parity_ok": ok,
                "roofline": roofline(comp + dec, kernel_ms, measured_traffic(fmt_name, n, kib), "%s:%d:%d" % (fmt_name, n, kib))}
    finally:
        db.close()


def main():
    args = parse_args()
    if args.format is None:
        args.format = "yaz0" if args.mode == "decode" else "lzss"
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)                       # (before anything touches HIP or imports torch)
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d` "
                         "(or `python bench.py --gpus %d` alone)\n" % (args.gpus, world, args.gpus, args.gpus, args.gpus))
        sys.exit(2)

    import numpy as np
    from auroralib.compression_amd import _abi as A
    from auroralib.compression_amd import synth
    from auroralib.compression_amd.batch import Context, Plan, partition_batch
    from auroralib.compression_amd.sharding import reduce_step_time, shard_seed

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    if args.all_ranks_on_device >= 0:
        local_rank = args.all_ranks_on_device
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)
    else:
        torch.cuda.set_device(local_rank)
    red_dev = "cuda" if (dist is not None and args.dist_backend == "nccl") else None
    set_device_shape(torch, local_rank)

    target = args.stream_kib * 1024
    if args.mode == "encode":
        run_encode_mode(args, np, A, synth, Context, Plan, shard_seed, reduce_step_time, rank, local_rank, world, dist, torch, red_dev)
        return
    if args.scaling == "strong" and world >= 1:
        # ONE batch of --streams streams: the library's partitioner decides which rank decodes which stream (it only looks at
        # format, decom_len and dst_cap, so the table needs no payload yet); a rank generates exactly its own streams
        ntot = args.streams
        table = (A.Stream * ntot)()
        tr = synth.stream_records(table)
        tr["dst_cap"], tr["decom_len"], tr["format"] = target, target, fmt_array(np, A, args.format, ntot)
        part, part_cost = partition_batch(table, world)
        mine = np.nonzero(part == rank)[0]
        n = len(mine)
        batch = synth.make_batch(tr["format"][mine], n, target, 0, seeds=(synth.seed_for(4) + mine).astype(np.uint64))
        parallelism = "ONE batch of %d streams, LPT-partitioned x%d (alz_partition_batch), no collective" % (ntot, world)
    else:
        n = args.streams
        # seed = 0xA17A0000 + 1000*config + stream index; ranks get disjoint stream indices
        batch = synth.make_batch(fmt_array(np, A, args.format, n), n, target, shard_seed(2, rank, n))
        parallelism = "stream-sharded x%d, every rank its own batch, no collective" % world
    recs = synth.stream_records(batch.streams)
    comp_bytes = int(recs["src_len"].astype(np.int64).sum())
    decomp_bytes = int(n) * target

    ctx = Context(local_rank)
    if args.kernel_variant:
        ctx.set_kernel_variant(args.kernel_variant)
    ranks = gather_rank_identities(ctx, torch, dist, rank, local_rank, world)
    main_db = DeviceBatch(ctx, batch, Plan)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    # ---- the timed region: EXACTLY K steps between barriers, one batch in flight
    dt_local, kernel_ms = run_steps(main_db, args.steps, args.warmup, barrier)
    dt = reduce_step_time(dt_local, dist, device=red_dev)

    # ---- secondary: the same steps with two batches in flight (two HIP streams, two plans, two output buffers)
    pipelined = None
    if args.inflight > 1:
        lanes = [main_db]
        for _ in range(1, args.inflight):
            lanes.append(DeviceBatch(Context(local_rank), batch, Plan))

        def barrier_all():
            barrier()
            for l in lanes[1:]:
                l.ctx.synchronize()
        for i in range(args.warmup):
            l = lanes[i % len(lanes)]; l.plan.execute(l.d_src, l.d_dst)
        barrier_all()
        t0 = time.perf_counter()
        for i in range(args.steps):
            l = lanes[i % len(lanes)]; l.plan.execute(l.d_src, l.d_dst)
        barrier_all()
        dt2 = reduce_step_time(time.perf_counter() - t0, dist, device=red_dev)
        for l in lanes[1:]:
            c2 = l.ctx; l.close(); c2.close()
        total2 = decomp_bytes * args.steps
        if dist is not None:
            tt = torch.tensor([float(total2)], dtype=torch.float64, device=red_dev or "cpu"); dist.all_reduce(tt); total2 = float(tt.item())
        pipelined = {"value": round(total2 / dt2 / 2**30, 3), "unit": "GiB/s", "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                     "batches_in_flight": args.inflight, "note": "caller-side overlap of consecutive batches on %d HIP streams / contexts" % args.inflight}

    # (kernel_ms: the dominant kernel's device time per launch, HIP events on the launch stream around the K timed steps themselves)

    # what was just measured decoded completely: every status OK, every length right (GPU results only)
    res = synth.result_records(main_db.plan.results())
    ok = bool((res["status"] == 0).all() and (res["dst_len"] == target).all())
    verified = None

    # ---- CPU restatement (oracle/): the reported baseline at N = 1, and the byte check of every rank's first streams.
    # The only place that touches oracle/ -- after the timed region, never inside it.
    cpu = None
    if not (args.no_verify and (args.no_cpu_baseline or world > 1)):
        import oracle_lib as O
        k = min(n, 1024)
        span = int(recs["dst_off"][k - 1]) + target
        o_dst = np.zeros(span + 64, dtype=np.uint8)
        o_res = (A.Result * k)()
        aff = len(os.sched_getaffinity(0))
        if not args.no_verify:
            O.lib.oracle_decode_batch(None, k, batch.src.ctypes.data, batch.streams, o_dst.ctypes.data, o_res, min(aff, 64))
            g = ctx.d2h(main_db.d_dst, span)
            verified = True
            for i in range(k):                            # stream by stream (gaps between streams are not output)
                a = int(recs["dst_off"][i])
                if not np.array_equal(g[a:a + target], o_dst[a:a + target]):
                    verified = False
                    break
            ok = ok and verified
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(O, A, np, batch, recs, n, target, args, aff)
    if dist is not None:                                # every rank's check counts
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=red_dev or "cpu"); dist.all_reduce(t, op=dist.ReduceOp.MIN); ok = bool(t.item() > 0.5)
        tb = torch.tensor([float(decomp_bytes), float(comp_bytes), float(n)], dtype=torch.float64, device=red_dev or "cpu"); dist.all_reduce(tb)
        job_dec, job_comp, job_n = float(tb[0].item()), float(tb[1].item()), int(tb[2].item())
    else:
        job_dec, job_comp, job_n = float(decomp_bytes), float(comp_bytes), n

    extras, configs = {}, None
    if rank == 0 and world == 1:
        if not args.no_extras:
            extras = run_extras(ctx, batch, recs, n, target, decomp_bytes, np, synth, A)
        if args.configs != "none":
            configs = run_configs(args, ctx, np, A, synth, Plan, Context)

    if rank == 0:
        value = job_dec * args.steps / dt / 2**30
        out = {
            "metric": "decompressed GiB/s (whole job; 10k x 256KiB batch per GPU), one batch in flight",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s decode, %d x %d KiB synthetic streams %s (SURVEY 8d token-level generator, seed 0xA17A0000+1000*cfg+i), device-resident"
                                   % (args.format, args.streams, args.stream_kib, "per GPU" if args.scaling == "weak" else "in ONE batch"),
                       "format": args.format, "streams_this_rank": n, "streams_whole_job": job_n, "stream_bytes": target,
                       "compressed_bytes_whole_job": int(job_comp), "parallelism": parallelism, "batches_in_flight": 1,
                       "pipelined": pipelined, "parity_ok": ok, "verified_vs_oracle": verified},
            "roofline": roofline(comp_bytes + decomp_bytes, kernel_ms, measured_traffic(args.format, n, args.stream_kib), "%s:%d:%d" % (args.format, n, args.stream_kib)),
            "cpu_baseline": cpu,
            "ranks": ranks,
        }
        out.update(extras)
        if configs is not None:
            out["configs"] = configs
        emit(out)
    main_db.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


def run_encode_mode(args, np, A, synth, Context, Plan, shard_seed, reduce_step_time, rank, local_rank, world, dist, torch, red_dev):
    """BASELINE.json configs[4] on N GPUs: every rank compresses its own raw buffers -- decoded synthetic LZSS streams, so they are
    compressible, produced and kept in HBM -- with alz_encode_batch_device; a step is one such call (all encode
    kernels of the batch, compressed streams left in HBM).  Buffers are independent (a fresh LzChainMatchFinder per call,
    LZSS.cs:135): no data-path collective.  --scaling strong: ONE batch of --streams buffers dealt out by index."""
    target = args.stream_kib * 1024
    fmt = A.FORMAT_NAMES.index(args.format)
    if args.scaling == "strong":
        mine = np.arange(rank, args.streams, world)
        n = len(mine)
        b = synth.make_batch(A.FMT_LZSS, n, target, 0, seeds=(synth.seed_for(5) + mine).astype(np.uint64))
        parallelism = "ONE batch of %d buffers dealt out x%d by index (equal sizes), no collective" % (args.streams, world)
    else:
        n = args.streams
        b = synth.make_batch(A.FMT_LZSS, n, target, shard_seed(5, rank, n))
        parallelism = "buffer-sharded x%d, every rank its own batch, no collective" % world
    ctx = Context(local_rank)
    ranks = gather_rank_identities(ctx, torch, dist, rank, local_rank, world)
    raw_db = DeviceBatch(ctx, b, Plan)                        # the raw buffers: decoded on the device, never leave it
    raw_db.plan.execute(raw_db.d_src, raw_db.d_dst); ctx.synchronize()
    rres = synth.result_records(raw_db.plan.results())
    recs = synth.stream_records(b.streams)
    cap = target + target // 4 + 64
    capal = (cap + 255) // 256 * 256
    streams = (A.Stream * n)()
    r2 = synth.stream_records(streams)
    r2["src_off"], r2["src_len"] = recs["dst_off"], target
    r2["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64(capal)
    r2["dst_cap"], r2["format"] = cap, fmt
    dst_bytes = n * capal + 64
    d_out = ctx.malloc(dst_bytes)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def step():
        return ctx.encode_batch_device(streams, raw_db.d_dst, b.dst_bytes, d_out, dst_bytes, quality=args.quality)
    for _ in range(max(1, args.warmup)):                      # (the first call at a quality grows the context's scratch: links, matches)
        eres, aux = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eres, aux = step()
    barrier()
    dt = reduce_step_time(time.perf_counter() - t0, dist, device=red_dev)
    kernel_ms = ctx.last_kernel_ms()
    er = synth.result_records(eres)
    comp = int(er["dst_len"].astype(np.int64).sum())
    ok = bool((rres["status"] == 0).all() and (er["status"] == 0).all())
    # what was written decodes back to the raw buffers (on the device), and the first buffers are the managed encoder's bytes
    verified = None
    if not args.no_verify:
        s3 = (A.Stream * n)()
        r3 = synth.stream_records(s3)
        r3["src_off"], r3["src_len"], r3["dst_off"], r3["dst_cap"], r3["decom_len"], r3["format"] = r2["dst_off"], er["dst_len"], recs["dst_off"], target, target, fmt
        ax = np.frombuffer(aux, dtype=np.uint32).reshape(n, 2)
        r3["aux0"], r3["aux1"] = ax[:, 0], ax[:, 1]
        d_back = ctx.malloc(b.dst_bytes + 64)
        pl = Plan(ctx, s3)
        pl.execute(d_out, d_back); ctx.synchronize()
        bres = synth.result_records(pl.results())
        k = min(n, 256)
        span = int(recs["dst_off"][k - 1]) + target
        g_raw, g_back = ctx.d2h(raw_db.d_dst, span), ctx.d2h(d_back, span)
        import oracle_lib as O
        back_ok = (bres["status"] == 0) & (bres["dst_len"] == target)
        quirk = None
        if fmt == A.FMT_LZO and not back_ok.all():
            # The managed LZO encoder can write two literal runs in a row (LZO.cs:167-188: a match shifted behind a short literal run and cut below
            # three bytes is dropped), which the managed decoder misreads (LZO.cs:70-95); parity keeps both.  For such a buffer the
            # check is: the compressed bytes are the restatement's, and the restatement's decoder reads them exactly as the GPU's did.
            bad = np.nonzero(~back_ok)[0]
            # Only a buffer that was CHECKED this way is waived; the others stay failures (at most 64 are checked: the quirk is rare).
            quirk = {"buffers_the_managed_decoder_misreads": int(len(bad)), "checked": int(min(len(bad), 64))}
            back_ok = back_ok.copy()
            for i in bad[:64]:
                raw_i = bytes(ctx.d2h(raw_db.d_dst, target, offset=int(recs["dst_off"][i])))
                got = bytes(ctx.d2h(d_out, int(er["dst_len"][i]), offset=int(r2["dst_off"][i])))
                want, _ = O.encode_stream(fmt, raw_i, quality=args.quality)
                odec, ores = O.decode_stream(fmt, got, decom_len=target, cap=target)
                gdec = bytes(ctx.d2h(d_back, int(bres["dst_len"][i]), offset=int(recs["dst_off"][i])))
                back_ok[i] = bool(got == want and int(ores.status) == int(bres["status"][i]) and int(ores.dst_len) == int(bres["dst_len"][i]) and odec == gdec)
        verified = bool(back_ok.all())
        for i in range(k):
            a = int(recs["dst_off"][i])
            if quirk is None or (bres["status"][i] == 0 and bres["dst_len"][i] == target):
                verified = verified and bool(np.array_equal(g_raw[a:a + target], g_back[a:a + target]))
        for i in range(min(n, 2)):
            a = int(recs["dst_off"][i])
            want, _ = O.encode_stream(fmt, bytes(g_raw[a:a + target]), quality=args.quality)
            got = ctx.d2h(d_out, int(er["dst_len"][i]), offset=int(r2["dst_off"][i]))
            verified = verified and bytes(got) == want
        pl.close(); ctx.free(d_back)
        ok = ok and verified
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_encode(ctx, raw_db, recs, r2, n, target, fmt, args, len(os.sched_getaffinity(0)), np, A)
    raw_bytes = float(n) * target
    if dist is not None:
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=red_dev or "cpu"); dist.all_reduce(t, op=dist.ReduceOp.MIN); ok = bool(t.item() > 0.5)
        tb = torch.tensor([raw_bytes, float(comp), float(n)], dtype=torch.float64, device=red_dev or "cpu"); dist.all_reduce(tb)
        job_raw, job_comp, job_n = float(tb[0].item()), float(tb[1].item()), int(tb[2].item())
    else:
        job_raw, job_comp, job_n = raw_bytes, float(comp), n
    if rank == 0:
        out = {
            "metric": "compressed raw GiB/s (whole job; BASELINE configs[4]: %s compression, 10k x 256KiB buffers per GPU)" % args.format,
            "value": round(job_raw * args.steps / dt / 2**30, 3), "unit": "GiB/s of raw input", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s compression at quality %d (LzChainMatchFinder + CompressHeaderless on the GPU, bit-identical), %d x %d KiB raw buffers %s "
                                   "(decoded synthetic LZSS streams, SURVEY 8d), device-resident" % (args.format, args.quality, args.streams, args.stream_kib,
                                                                                                      "per GPU" if args.scaling == "weak" else "in ONE batch"),
                       "mode": "encode", "format": args.format, "quality": args.quality, "streams_this_rank": n, "streams_whole_job": job_n, "stream_bytes": target,
                       "compressed_bytes_whole_job": int(job_comp), "ratio": round(job_comp / job_raw, 4), "parallelism": parallelism,
                       "parity_ok": ok, "verified_roundtrip_and_vs_oracle": verified, "lzo_reference_quirk": quirk if not args.no_verify else None},
            "roofline": roofline(raw_bytes + comp, kernel_ms, measured_traffic("%s_encode_q%d" % (args.format, args.quality), n, args.stream_kib), family="encode"),
            "cpu_baseline": cpu,
            "ranks": ranks,
        }
        emit(out)
    ctx.free(d_out); raw_db.close(); ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_quota():
    """CPU bandwidth limit of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: a process
    may be ALLOWED on every core (sched_getaffinity) and still be throttled to a few cores' worth of time."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_baseline(O, A, np, batch, recs, n, target, args, aff):
    """The C restatement of the managed ring + flush path (kind "port": the managed library itself cannot run, there is no
    .NET on either box) on a bounded sample of the measured batch: one thread, then the cores this process may use
    (sched_getaffinity, not os.cpu_count()), with the steps in between so that the scaling is visible."""
    def rate(nstreams, threads, budget):
        span = int(recs["dst_off"][nstreams - 1]) + target
        dst = np.ones(span + 64, dtype=np.uint8)            # pre-touched
        res = (A.Result * nstreams)()
        reps, t0 = 0, time.perf_counter()
        while reps < 1 or (time.perf_counter() - t0 < budget and reps < 50):
            O.lib.oracle_decode_batch(None, nstreams, batch.src.ctypes.data, batch.streams, dst.ctypes.data, res, threads)
            reps += 1
        return nstreams * target * reps / (time.perf_counter() - t0) / 2**30, reps
    sweep, reps_all = {}, 0
    threads = sorted(set([1] + [t for t in (8, 32, 64, 128) if t < aff] + [aff]))
    for t in threads:
        ns = min(n, max(256, 64 * t))
        v, r = rate(ns, t, 3.0 if t not in (1, aff) else 6.0)
        sweep[str(t)] = round(v, 3)
        if t == aff:
            reps_all = r
    best_t = max(sweep, key=lambda k: sweep[k])
    # ... and the reference's own benchmark shape on ONE core of this box: one 1 000 KiB stream of Test.bmp per body (what the
    # `single_<format>_q0` entries of `configs` decode through alz_decode)
    single, single_c = {}, {}
    try:
        bmp = O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0]
        raw1 = bytes(bmp[:1024000])
        for fname in ("yaz0", "yay0", "mio0", "lz10", "lz11", "lzss", "prs_be", "lzo", "lz4_block", "snappy_raw"):
            fmt = A.FORMAT_NAMES.index(fname)
            for q in (0, 15):                                     # Compress, the C port of LzChainMatchFinder + the format's writer, one core
                t0 = time.perf_counter(); comp, aux = O.encode_stream(fmt, raw1, quality=q); dt = time.perf_counter() - t0
                t0 = time.perf_counter(); comp, aux = O.encode_stream(fmt, raw1, quality=q); dt = min(dt, time.perf_counter() - t0)
                single_c["%s_q%d" % (fname, q)] = round(len(raw1) / dt / 2**30, 3)
            comp, aux = O.encode_stream(fmt, raw1, quality=0)
            sized = fname not in ("prs_be", "lzo", "lz4_block", "snappy_raw")
            reps, t0 = 0, time.perf_counter()
            while reps < 3 or time.perf_counter() - t0 < 0.25:
                out, r = O.decode_stream(fmt, comp, decom_len=len(raw1) if sized else 0, cap=len(raw1), aux0=aux.aux0, aux1=aux.aux1)
                reps += 1
            single[fname] = round(len(raw1) * reps / (time.perf_counter() - t0) / 2**30, 3)
    except Exception as e:
        single = {"error": repr(e)}
    # `value` = the best thread count of the sweep, `cores` = the threads that run used.  Round 1 reported T = os.cpu_count()
    # = 256 at 10 GiB/s = 0.04 GiB/s per thread against 0.96 for one thread: the sweep shows where the scaling stops (a cgroup
    # CPU quota -- reported below -- and / or one NUMA node's memory: the output buffer is first touched by one thread)
    return {"value": sweep[best_t], "unit": "GiB/s", "cores": int(best_t), "kind": "port",
            "single_thread": sweep["1"], "all_allowed_cores": {"threads": aff, "value": sweep[str(aff)]}, "threads_sweep_GiB_s": sweep,
            "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(), "sched_affinity": aff, "cgroup_cpu_quota_cores": cpu_quota(),
            "single_stream_one_thread_GiB_s": single, "single_stream_compress_one_thread_GiB_s": single_c,
            "sample": "first max(256, 64 T) of the %d x %d KiB %s streams per thread count T, %d passes at T = %d; C restatement of the managed "
                      "ring+flush path (oracle/alz_oracle.c), streams striped over threads" % (n, args.stream_kib, args.format, reps_all, aff)}


def cpu_baseline_encode(ctx, raw_db, recs, r2, n, target, fmt, args, aff, np, A):
    """Compression beside it: the C restatement of LzChainMatchFinder + CompressHeaderless (kind "port", oracle/alz_oracle.c) on the first
    buffers of the measured batch, one thread and the cores this process may use; a bounded sample (a few seconds per thread count)."""
    import ctypes as C
    import oracle_lib as O
    st = A.Settings(); st.quality = args.quality

    def rate(nb, threads, budget):
        span = int(recs["dst_off"][nb - 1]) + target
        raw = np.frombuffer(ctx.d2h(raw_db.d_dst, span + 64), dtype=np.uint8).copy()
        streams = (A.Stream * nb)()
        C.memmove(streams, r2[:nb].tobytes(), nb * C.sizeof(A.Stream))
        dst = np.ones(int(r2["dst_off"][nb - 1]) + int(r2["dst_cap"][nb - 1]) + 64, dtype=np.uint8)
        res = (A.Result * nb)(); aux = (A.EncodeAux * nb)()
        reps, t0 = 0, time.perf_counter()
        while reps < 1 or (time.perf_counter() - t0 < budget and reps < 20):
            O.lib.oracle_encode_batch(None, C.byref(st), nb, raw.ctypes.data, streams, dst.ctypes.data, res, aux, threads)
            reps += 1
        return nb * target * reps / (time.perf_counter() - t0) / 2**30, reps
    sweep, reps_all, nb_all = {}, 0, 0
    for t in sorted(set([1, aff])):
        nb = min(n, max(32, 8 * t))
        v, r = rate(nb, t, 6.0)
        sweep[str(t)] = round(v, 4)
        if t == aff:
            reps_all, nb_all = r, nb
    best_t = max(sweep, key=lambda k: sweep[k])
    return {"value": sweep[best_t], "unit": "GiB/s of raw input", "cores": int(best_t), "kind": "port", "single_thread": sweep["1"],
            "threads_sweep_GiB_s": sweep, "cpu_model": cpu_model(), "sched_affinity": aff, "cgroup_cpu_quota_cores": cpu_quota(),
            "sample": "first %d of the %d x %d KiB raw buffers at quality %d, %d passes at T = %d; C restatement of LzChainMatchFinder + "
                      "CompressHeaderless (oracle/alz_oracle.c), buffers striped over threads" % (nb_all, n, args.stream_kib, args.quality, reps_all, aff)}


def cpu_port_encode(np, A, raw, rec_table, nb, fmt, quality, budget=2.0):
    """`cpu_port` of a compression entry: the SAME buffers (the first `nb` of the entry's batch, host copy `raw`, descriptors `rec_table`)
    through oracle_encode_batch -- the C restatement of LzChainMatchFinder + CompressHeaderless -- on T = min(allowed cores, 32) threads,
    a bounded sample (`budget` seconds).  After the GPU measurement, never inside it."""
    import ctypes as C
    import oracle_lib as O
    st = A.Settings(); st.quality = quality
    threads = max(1, min(len(os.sched_getaffinity(0)), 32))
    streams = (A.Stream * nb)()
    r = np.frombuffer(streams, dtype=rec_table.dtype)
    r[:] = rec_table[:nb]
    base = int(r["src_off"].min())
    r["src_off"] -= np.uint64(base)
    cap = int(r["dst_cap"].max()); capal = (cap + 255) // 256 * 256
    r["dst_off"] = np.arange(nb, dtype=np.uint64) * np.uint64(capal)
    r["format"] = fmt
    dst = np.ones(nb * capal + 64, dtype=np.uint8)
    res = (A.Result * nb)(); aux = (A.EncodeAux * nb)()
    view = raw[base:]
    reps, t0 = 0, time.perf_counter()
    while reps < 1 or (time.perf_counter() - t0 < budget and reps < 20):
        O.lib.oracle_encode_batch(None, C.byref(st), nb, view.ctypes.data, streams, dst.ctypes.data, res, aux, threads)
        reps += 1
    dt = time.perf_counter() - t0
    nbytes = int(r["src_len"].astype(np.int64).sum())
    return {"value": round(nbytes * reps / dt / 2**30, 3), "unit": "GiB/s of raw input", "cores": threads, "kind": "port",
            "sample": "first %d buffers of this entry's batch, %d passes" % (nb, reps)}


def run_extras(ctx, batch, recs, n, target, decomp_bytes, np, synth, A):
    """Second roofline denominator and the PCIe-inclusive rate (never `value`)."""
    out = {}
    try:
        gbs = ctx.copy_bandwidth(1 << 30, 10)
        out["copy_bandwidth"] = {"value": round(gbs, 1), "unit": "GB/s", "what": "16 B/lane device-to-device copy kernel, 1 GiB, bytes read + written",
                                 "frac_of_spec_peak": round(gbs / HBM_PEAK_GBS, 4)}
    except Exception as e:                                   # a measurement aid must not take the bench line down
        out["copy_bandwidth"] = {"error": str(e)}
    try:
        g_dst, g_res = ctx.decode_batch(batch.streams, batch.src, batch.dst_bytes)       # grows the device staging buffers
        t0 = time.perf_counter()
        g_dst, g_res = ctx.decode_batch(batch.streams, batch.src, batch.dst_bytes, dst=g_dst)
        dt = time.perf_counter() - t0
        r = synth.result_records(g_res)
        t0 = time.perf_counter()
        ctx.decode_batch(batch.streams, batch.src, batch.dst_bytes)        # a NEW destination array: its pages are faulted in by the copy-out
        dt_fresh = time.perf_counter() - t0
        out["end_to_end"] = {"value": round(decomp_bytes / dt / 2**30, 3), "unit": "GiB/s", "seconds": round(dt, 4),
                             "fresh_destination_GiB_s": round(decomp_bytes / dt_fresh / 2**30, 3),
                             "what": "alz_decode_batch on host buffers: upload of the compressed batch + decode + download of %d x %d KiB, "
                                     "pageable caller buffers staged through two pinned 32 MiB buffers by the library's copy threads; "
                                     "`value` with a destination whose pages are resident, fresh_destination with a newly allocated one" % (n, target // 1024),
                             "ok": bool((r["status"] == 0).all())}
    except Exception as e:
        out["end_to_end"] = {"error": str(e)}
    return out


def run_configs(args, ctx, np, A, synth, Plan, Context):
    want = ["cfg2", "cfg3", "cfg4", "cfg5", "bodies", "realistic", "single"] if args.configs == "all" else args.configs.split(",")
    out = []
    steps = max(3, min(args.steps, 10))
    try:
        if "cfg2" in want:
            b = synth.make_batch(A.FMT_YAZ0, 10000, 65536, synth.seed_for(2))
            out.append(decode_config("cfg2", "BASELINE configs[1]: Yaz0 decode, 10 000 x 64 KiB synthetic streams, 1 GPU", ctx, b, Plan, synth, np, steps, "yaz0", 10000, 64))
        if "cfg3" in want:
            out.append(cfg3(ctx, np, A, synth, Plan))
        if "cfg4" in want:
            b = synth.make_batch(fmt_array(np, A, "mixed", 5000), 5000, 262144, synth.seed_for(4))
            out.append(decode_config("cfg4_shard", "BASELINE configs[3], one GPU's shard: 5 000 of the 40 000 mixed LZ10/LZ11/Yaz0/PRS streams x 256 KiB, "
                                     "per-format kernels forked onto side streams", ctx, b, Plan, synth, np, steps, "mixed", 5000, 256))
        if "cfg5" in want:
            out.extend(cfg5(ctx, np, A, synth, Plan, with_cpu=not args.no_cpu_baseline))
        if "bodies" in want:
            # every other decode body north_star names, on the metric's own shape (10 000 x 256 KiB synthetic streams, one GPU): the
            # per-format table of docs/EXPERIMENTS.md 4.4 in the driver-run line
            bsteps = max(3, min(args.steps, 5))
            for f in BODIES:
                b = synth.make_batch(A.FORMAT_NAMES.index(f), 10000, 262144, synth.seed_for(2))
                out.append(decode_config("body_" + f, "%s decode, 10 000 x 256 KiB synthetic streams, 1 GPU (the headline's shape, another body)" % f,
                                         ctx, b, Plan, synth, np, bsteps, f, 10000, 256))
                del b
        if "realistic" in want:
            # the reference benchmarks every algorithm on Test.bmp (Benchmarks/Benchmarks/TestAllAlgorithms.cs:26-69): the same data
            # per north-star format, as 256 KiB windows
            for f in ("yaz0", "lz10", "lz11", "prs_be", "lz4_block"):
                out.append(realistic(ctx, np, A, synth, Plan, steps, f))
            # ... and the other direction on that data: the encoder picks kernel B per stream (enc_probe_kernel: real data goes to the
            # one-position-per-lane kernel, the synthetic batches of cfg5 to the two-phase one)
            for f in ("yaz0", "lz4_block"):
                out.append(realistic_compress(ctx, np, A, synth, f, 8, with_cpu=not args.no_cpu_baseline))
            # ... and batches of FEW buffers (16 / 256 windows of 64 KiB): parse and emitter over segments, csrc/alz_encode_seg.h
            # (LZ4 blocks, LZO: round 6, the speculative walk per segment of csrc/alz_encode_seg_seq.h)
            for f, q, nb in (("yaz0", 8, 256), ("yaz0", 8, 16), ("yaz0", 0, 256), ("snappy_raw", 0, 256), ("lz4_block", 8, 256), ("lzo", 8, 256)):
                out.append(realistic_compress(ctx, np, A, synth, f, q, with_cpu=not args.no_cpu_baseline, n=nb, size=65536, prefix="mid_compress"))
        if "single" in want:
            out.extend(single_stream(ctx, np, A, synth, Plan))
    except Exception as e:                                   # report what ran; the headline line must still come out
        out.append({"name": "error", "error": repr(e)})
    return out


def cfg3(ctx, np, A, synth, Plan):
    """BASELINE configs[2] at its stated size: 100 000 independent LZ4 blocks x 256 KiB = 24.4 GiB of output per launch,
    generated and uploaded in ten parts of 10 000 (distinct seeds), decoded as ONE batch."""
    n, target, parts = 100000, 262144, 10
    per = n // parts
    bs = [None] * parts
    src_total = 0
    streams = (A.Stream * n)()
    rec = synth.stream_records(streams)
    dal = (target + 255) // 256 * 256
    d_dst = ctx.malloc(n * dal + 64)
    # two passes: sizes first (device buffer), then payload part by part
    srcs = []
    for p in range(parts):
        b = synth.make_batch(A.FMT_LZ4_BLOCK, per, target, synth.seed_for(3) + p * per)
        r = synth.stream_records(b.streams)
        sl = slice(p * per, (p + 1) * per)
        rec["src_off"][sl] = r["src_off"] + src_total
        rec["src_len"][sl], rec["dst_cap"][sl], rec["decom_len"][sl], rec["format"][sl] = r["src_len"], target, target, A.FMT_LZ4_BLOCK
        rec["dst_off"][sl] = (np.arange(per, dtype=np.uint64) + np.uint64(p * per)) * np.uint64(dal)
        srcs.append((src_total, b.src))
        src_total += (b.src.nbytes + 63) // 64 * 64
    d_src = ctx.malloc(src_total + 64)
    import ctypes as C
    for off, arr in srcs:
        ctx.lib.alz_memcpy_h2d(ctx.h, C.c_void_p(d_src.value + off), arr.ctypes.data_as(C.c_void_p), arr.nbytes)
    del srcs
    plan = Plan(ctx, streams)
    try:
        steps = 3
        plan.execute(d_src, d_dst); ctx.synchronize()
        t0 = time.perf_counter()
        kernel_ms = plan.execute_timed(d_src, d_dst, iters=steps)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        res = synth.result_records(plan.results())
        ok = bool((res["status"] == 0).all() and (res["dst_len"] == target).all())
        comp = int(rec["src_len"].astype(np.int64).sum())
        return {"name": "cfg3", "workload": "BASELINE configs[2]: LZ4 block decode, 100 000 x 256 KiB independent synthetic blocks, 1 GPU (24.4 GiB of output per launch)",
                "value": round(n * target * steps / dt / 2**30, 3), "unit": "GiB/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3), "parity_ok": ok,
                "roofline": roofline(comp + n * target, kernel_ms, measured_traffic("lz4_block", n, 256))}
    finally:
        plan.close(); ctx.free(d_src); ctx.free(d_dst)


def cfg5(ctx, np, A, synth, Plan, with_cpu=True):
    """BASELINE configs[4] on one GPU: compression of 10 000 x 256 KiB raw buffers (decoded synthetic LZSS streams, so they are
    compressible; produced on the device and kept there) through alz_encode_batch_device -- LZSS(12,4,2), the configuration's own
    format, at Q0 / Q8 / Q15 (the levels the reference publishes and its default), and Yaz0 and LZ4 blocks at Q0 / Q8.  The kernel time
    (every encode kernel of the call) comes from HIP events inside the call; what was written is decoded back on the device and the
    first 256 buffers are compared byte for byte with the input."""
    n, size = 10000, 262144
    b = synth.make_batch(A.FMT_LZSS, n, size, synth.seed_for(5))
    raw_db = DeviceBatch(ctx, b, Plan)
    out = []
    d_out = d_back = None
    try:
        raw_db.plan.execute(raw_db.d_src, raw_db.d_dst); ctx.synchronize()
        recs = synth.stream_records(b.streams)
        cap = size + size // 4 + 64
        capal = (cap + 255) // 256 * 256
        dst_bytes = n * capal + 64
        d_out = ctx.malloc(dst_bytes)
        d_back = ctx.malloc(b.dst_bytes + 64)
        k = 256
        span = int(recs["dst_off"][k - 1]) + size
        g_raw = ctx.d2h(raw_db.d_dst, span)
        for fname, q in (("lzss", 0), ("lzss", 8), ("lzss", 15), ("yaz0", 0), ("yaz0", 8), ("lz4_block", 0), ("lz4_block", 8)):
            fmt = A.FORMAT_NAMES.index(fname)
            streams = (A.Stream * n)()
            r2 = synth.stream_records(streams)
            r2["src_off"], r2["src_len"] = recs["dst_off"], size
            r2["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64(capal)
            r2["dst_cap"], r2["format"] = cap, fmt
            ctx.encode_batch_device(streams, raw_db.d_dst, b.dst_bytes, d_out, dst_bytes, quality=q)   # (first call at a quality: the context grows its device scratch)
            t0 = time.perf_counter()
            eres, aux = ctx.encode_batch_device(streams, raw_db.d_dst, b.dst_bytes, d_out, dst_bytes, quality=q)
            call_s = time.perf_counter() - t0
            kernel_ms = ctx.last_kernel_ms()
            er = synth.result_records(eres)
            comp = int(er["dst_len"].astype(np.int64).sum())
            # round trip on the GPU: what was written decodes back to the input
            s3 = (A.Stream * n)()
            r3 = synth.stream_records(s3)
            r3["src_off"], r3["src_len"], r3["dst_off"], r3["dst_cap"], r3["decom_len"], r3["format"] = r2["dst_off"], er["dst_len"], recs["dst_off"], size, size, fmt
            ctx.memset(d_back, 0, b.dst_bytes)
            pl = Plan(ctx, s3)
            try:
                pl.execute(d_out, d_back); ctx.synchronize()
                bres = synth.result_records(pl.results())
            finally:
                pl.close()
            g_back = ctx.d2h(d_back, span)
            ok = bool((er["status"] == 0).all() and (bres["status"] == 0).all() and (bres["dst_len"] == size).all())
            for i in range(k):
                a = int(recs["dst_off"][i])
                ok = ok and bool(np.array_equal(g_raw[a:a + size], g_back[a:a + size]))
            name = ("cfg5_q%d" % q) if fname == "lzss" else ("cfg5_%s_q%d" % (fname, q))
            cpu_port = None
            if with_cpu:
                try:
                    cpu_port = cpu_port_encode(np, A, g_raw, r2, k, fmt, q)
                except Exception as e:
                    cpu_port = {"error": repr(e)}
            out.append({"name": name, "cpu_port": cpu_port, "workload": "BASELINE configs[4]: %s compression, parallel hash-chain match-find + emit, 10 000 x 256 KiB, quality %d, 1 GPU, device-resident"
                        % ("LZSS(12,4,2)" if fname == "lzss" else fname, q),
                        "value": round(n * size / (kernel_ms * 1e-3) / 2**30, 3), "unit": "GiB/s of raw input (kernels)", "kernel_ms": round(kernel_ms, 3),
                        "call_ms": round(call_s * 1e3, 3), "ratio": round(comp / (n * size), 4), "parity_ok": ok,
                        "roofline": roofline(n * size + comp, kernel_ms, measured_traffic("%s_encode_q%d" % (fname, q), n, size // 1024), family="encode")})
    finally:
        if d_out is not None:
            ctx.free(d_out)
        if d_back is not None:
            ctx.free(d_back)
        raw_db.close()
        ctx.release_scratch()                                  # (the encoder's grow-only scratch: ~20 GB at Q8; the configurations behind this one start clean)
    return out


# Benchmarks.md (BenchmarkDotNet, Ryzen 7 3800X, .NET 8, one thread): MB/s of Decompress on the first 1 000 KiB of Test.bmp, Q0- / Q15-encoded input
PUBLISHED_SINGLE = {("yay0", 0): (470.82, "Benchmarks.md:78"), ("yay0", 15): (824.11, "Benchmarks.md:80"),
                    ("mio0", 0): (419.93, "Benchmarks.md:70"), ("mio0", 15): (419.39, "Benchmarks.md:72"),
                    ("yaz0", 0): (494.53, "Benchmarks.md:82"), ("yaz0", 15): (877.48, "Benchmarks.md:84"),
                    ("lz10", 0): (427.99, "Benchmarks.md:58"), ("lz10", 15): (429.68, "Benchmarks.md:60"),
                    ("lz11", 0): (560.85, "Benchmarks.md:62"), ("lz11", 15): (951.02, "Benchmarks.md:64"),
                    ("lzss", 0): (330.31, "Benchmarks.md:30"), ("lzss", 15): (359.94, "Benchmarks.md:32"),
                    ("prs_be", 0): (459.77, "Benchmarks.md:94"), ("prs_be", 15): (900.58, "Benchmarks.md:96"),
                    ("lzo", 0): (636.34, "Benchmarks.md:26"), ("lzo", 15): (1457.70, "Benchmarks.md:28"),
                    ("lz4_block", 0): (747.58, "Benchmarks.md:22 (LZ4Legacy: this block behind an 8-byte header)"),
                    ("lz4_block", 15): (2191.72, "Benchmarks.md:24 (LZ4Legacy: this block behind an 8-byte header)"),
                    ("snappy_raw", 0): (525.16, "Benchmarks.md:34"), ("snappy_raw", 15): (655.46, "Benchmarks.md:36")}
# the same table's Compress rows (MB/s of raw input), CompressionLevel 0 / 15
PUBLISHED_SINGLE_COMPRESS = {("yay0", 0): (229.43, "Benchmarks.md:77"), ("yay0", 15): (68.83, "Benchmarks.md:79"),
                             ("mio0", 0): (179.70, "Benchmarks.md:69"), ("mio0", 15): (111.97, "Benchmarks.md:71"),
                             ("yaz0", 0): (255.98, "Benchmarks.md:81"), ("yaz0", 15): (87.02, "Benchmarks.md:83"),
                             ("lz10", 0): (194.28, "Benchmarks.md:57"), ("lz10", 15): (112.11, "Benchmarks.md:59"),
                             ("lz11", 0): (246.42, "Benchmarks.md:61"), ("lz11", 15): (135.32, "Benchmarks.md:63"),
                             ("lzss", 0): (188.34, "Benchmarks.md:29"), ("lzss", 15): (109.92, "Benchmarks.md:31"),
                             ("prs_be", 0): (238.65, "Benchmarks.md:93"), ("prs_be", 15): (83.24, "Benchmarks.md:95"),
                             ("lzo", 0): (289.24, "Benchmarks.md:25"), ("lzo", 15): (89.93, "Benchmarks.md:27"),
                             ("lz4_block", 0): (260.67, "Benchmarks.md:21 (LZ4Legacy)"), ("lz4_block", 15): (103.50, "Benchmarks.md:23 (LZ4Legacy)"),
                             ("snappy_raw", 0): (286.26, "Benchmarks.md:33"), ("snappy_raw", 15): (109.70, "Benchmarks.md:35")}


def single_stream(ctx, np, A, synth, Plan):
    """The reference's OWN benchmark shape (Benchmarks/Benchmarks/TestAllAlgorithms.cs:37-69): ONE stream = the first 1 000 KiB of
    Test.bmp, compressed at quality 0 / 15, then Compress and Decompress timed -- for the formats whose single streams run on the whole GPU
    (LZSS / LZ10 / LZ11 / Yaz0 / Yay0 / MIO0 / PRS / LZO / LZ4 blocks / Snappy: csrc/alz_big.hip decodes, csrc/alz_encode_big.h encodes).  `value` = alz_decode on HOST buffers (upload + kernels + download + the call: what a format
    class's Decompress(Stream, Stream) costs), `device_GiB_s` = the kernels alone; beside them the managed figure the reference
    publishes for this exact input on its own machine (another CPU, no GPU: context, not a baseline measured here)."""
    from auroralib.compression_amd import formats as F
    lz = F.LZSS(A.LzProperties.from_bits(10, 6, 2))
    bmp = lz.Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read())
    raw = bytes(bmp[:1024000]); n = len(raw)
    out = []
    import json
    import xxhash
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_vectors.json")))["benchmark"]    # (made by tests/golden/make_vectors.py)
    rawsrc = np.frombuffer(raw + bytes(64), dtype=np.uint8)
    for fname in ("yaz0", "yay0", "mio0", "lz10", "lz11", "lzss", "prs_be", "lzo", "lz4_block", "snappy_raw"):
        fmt = A.FORMAT_NAMES.index(fname)
        for q in (0, 15):
            cap = n + n // 4 + 64
            es = (A.Stream * 1)(A.Stream(0, 0, n, cap, 0, 0, 0, fmt))
            before = ctx.big_stream()
            enc, eres, eaux = ctx.encode_batch(es, rawsrc, cap + 64, quality=q)
            enc_whole_gpu = ctx.big_stream() > before
            comp = bytes(enc[:eres[0].dst_len]); a0, a1 = eaux[0].aux0, eaux[0].aux1
            # ---- Compress: alz_encode_batch of ONE stream on host buffers (upload + kernels + download + the call: what a format class's
            # Compress(ReadOnlySpan<byte>, Stream) costs), checked against the committed vector of this exact input
            reps, t0, kms = 10, time.perf_counter(), 0.0
            for _ in range(reps):
                enc, eres, eaux = ctx.encode_batch(es, rawsrc, cap + 64, quality=q)
                kms += ctx.last_kernel_ms()
            cwall_ms = (time.perf_counter() - t0) / reps * 1e3
            glen, gdig, ga0, ga1 = golden["%s:%d:q%d" % (fname, n, q)]
            cok = (eres[0].status == 0 and eres[0].dst_len == glen and xxhash.xxh64(bytes(enc[:glen])).hexdigest() == gdig and
                   (eaux[0].aux0, eaux[0].aux1) == (ga0, ga1))
            cpub, cwhere = PUBLISHED_SINGLE_COMPRESS[(fname, q)]
            out.append({"name": "single_compress_%s_q%d" % (fname, q),
                        "workload": "ONE %s stream: Test.bmp[0:1 024 000] compressed at quality %d (ratio %.4f), the reference's benchmark shape; alz_encode_batch of one stream on host buffers"
                                    % (fname, q, len(comp) / n),
                        "value": round(n / (cwall_ms * 1e-3) / 2**30, 3), "unit": "GiB/s of raw input", "ms_per_call": round(cwall_ms, 4),
                        "kernel_ms": round(kms / reps, 4), "whole_gpu_path": enc_whole_gpu, "parity_ok": cok,
                        "published_managed": {"MB_per_s": cpub, "GiB_s": round(cpub * 1e6 / 2**30, 3), "where": cwhere,
                                              "hardware": "AMD Ryzen 7 3800X, .NET 8, one thread (BenchmarkDotNet); not measured here"}})
            decl = 0 if fname in ("lz4_block", "prs_be", "lzo", "snappy_raw") else n   # (an LZ4 block / a PRS, LZO or Snappy body carries no size the descriptor states: the destination's room bounds it)
            st = (A.Stream * 1)(A.Stream(0, 0, len(comp), n, decl, a0, a1, fmt))
            d_src, d_dst = ctx.malloc(len(comp) + 64), ctx.malloc(n + 64)
            try:
                ctx.h2d(d_src, np.frombuffer(comp + bytes(64), dtype=np.uint8))
                before = ctx.big_stream()
                p = Plan(ctx, st)
                p.execute(d_src, d_dst); ctx.synchronize()
                dev_ms = p.execute_timed(d_src, d_dst, iters=20)
                ok = p.results()[0].status == 0 and bytes(ctx.d2h(d_dst, n)) == raw
                p.close()
                whole_gpu = ctx.big_stream() > before
            finally:
                ctx.free(d_src); ctx.free(d_dst)
            got, r = ctx.decode(fmt, comp, decom_len=decl, cap=n, aux0=a0, aux1=a1)
            reps, t0 = 20, time.perf_counter()
            for _ in range(reps):
                got, r = ctx.decode(fmt, comp, decom_len=decl, cap=n, aux0=a0, aux1=a1)
            wall_ms = (time.perf_counter() - t0) / reps * 1e3
            ok = ok and r.status == 0 and got == raw
            pub, where = PUBLISHED_SINGLE[(fname, q)]
            out.append({"name": "single_%s_q%d" % (fname, q),
                        "workload": "ONE %s stream: Test.bmp[0:1 024 000] compressed at quality %d (ratio %.4f), the reference's benchmark shape; alz_decode on host buffers"
                                    % (fname, q, len(comp) / n),
                        "value": round(n / (wall_ms * 1e-3) / 2**30, 3), "unit": "GiB/s", "ms_per_call": round(wall_ms, 4),
                        "device_GiB_s": round(n / (dev_ms * 1e-3) / 2**30, 3), "kernel_ms": round(dev_ms, 4), "whole_gpu_path": whole_gpu, "parity_ok": ok,
                        "published_managed": {"MB_per_s": pub, "GiB_s": round(pub * 1e6 / 2**30, 3), "where": where,
                                              "hardware": "AMD Ryzen 7 3800X, .NET 8, one thread (BenchmarkDotNet); not measured here"}})
    return out


def realistic(ctx, np, A, synth, Plan, steps, fmt_name="yaz0"):
    """SURVEY.md 8d's second data set: the 256 KiB windows of the reference's Test.bmp at a stride of 4 KiB (193 windows),
    encoded at the default quality by the GPU encoder (bit-identical to the managed encoder, tests/test_gpu_encode.py),
    repeated to 10 000 streams.  Test.bmp itself is recovered from the reference's own fixture Test.lz on the GPU."""
    fmt = A.FORMAT_NAMES.index(fmt_name)
    from auroralib.compression_amd import formats as F
    lz = F.LZSS(A.LzProperties.from_bits(10, 6, 2))
    bmp = np.frombuffer(lz.Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read()), dtype=np.uint8)
    size, stride = 262144, 4096
    starts = list(range(0, len(bmp) - size + 1, stride))
    nw = len(starts)
    raw = np.concatenate([bmp[s:s + size] for s in starts])
    cap = size + size // 4 + 64
    st = (A.Stream * nw)()
    r = synth.stream_records(st)
    r["src_off"], r["src_len"] = np.arange(nw, dtype=np.uint64) * np.uint64(size), size
    r["dst_off"] = np.arange(nw, dtype=np.uint64) * np.uint64((cap + 255) // 256 * 256)
    r["dst_cap"], r["format"] = cap, fmt
    enc, eres, aux = ctx.encode_batch(st, raw, int(r["dst_off"][-1]) + cap + 64, quality=8)
    er = synth.result_records(eres)
    # the batch: 10 000 streams, stream i = window i mod nw (payload packed once per window)
    n = 10000
    offs = np.zeros(nw, dtype=np.uint64)
    al = (er["dst_len"].astype(np.uint64) + np.uint64(15)) & ~np.uint64(15)
    offs[1:] = np.cumsum(al)[:-1]
    src = np.zeros(int(offs[-1] + al[-1]) + 64, dtype=np.uint8)
    for w in range(nw):
        a = int(r["dst_off"][w]); src[int(offs[w]):int(offs[w]) + int(er["dst_len"][w])] = enc[a:a + int(er["dst_len"][w])]
    streams = (A.Stream * n)()
    s2 = synth.stream_records(streams)
    w = np.arange(n) % nw
    s2["src_off"], s2["src_len"] = offs[w], er["dst_len"][w]
    s2["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64(size)
    s2["dst_cap"], s2["decom_len"], s2["format"] = size, size, fmt

    class B:
        pass
    b = B(); b.src, b.streams, b.dst_bytes = src, streams, n * size
    db = DeviceBatch(ctx, b, Plan)
    try:
        dt, kernel_ms = best_of_two(db, steps, ctx.synchronize)
        res = synth.result_records(db.plan.results())
        ok = bool((res["status"] == 0).all() and (res["dst_len"] == size).all())
        g = ctx.d2h(db.d_dst, nw * size)
        ok = ok and bool(np.array_equal(g, raw))                                  # the first nw streams are the windows themselves
        comp = int(s2["src_len"].astype(np.int64).sum())
        return {"name": "realistic_" + fmt_name, "workload": "%s decode of the %d 256 KiB windows of Test.bmp (stride 4 KiB, GPU-encoded at Q8, ratio %.3f), repeated to 10 000 streams"
                % (fmt_name, nw, comp / (n * size)), "value": round(n * size * steps / dt / 2**30, 3), "unit": "GiB/s", "steps": steps,
                "ms_per_step": round(dt / steps * 1e3, 4), "parity_ok": ok,
                "input_residency": "the compressed input is %d distinct windows (%.1f MB) read by 10 000 streams: it is served by L2 / MALL, so HBM moves fewer bytes than "
                                   "the algorithmic count (`roofline.traffic` < `algorithmic_bytes_per_launch`); `frac` prices bytes DECODED against the HBM peak, "
                                   "not bytes HBM moved" % (nw, float(al.sum()) / 1e6),
                "roofline": roofline(comp + n * size, kernel_ms, measured_traffic("realistic_" + fmt_name, n, size // 1024))}
    finally:
        db.close()


def realistic_compress(ctx, np, A, synth, fmt_name, quality, with_cpu=True, n=10000, size=262144, prefix="realistic_compress"):
    """10 000 windows of 256 KiB of Test.bmp (evenly spaced starts: ~8 700 distinct windows, 2.4 GiB of raw input resident in HBM) through
    alz_encode_batch_device at the default quality: the batch encoder on real data, at the metric's own batch size.  Checked by decoding eight
    of the streams back (the bytes themselves are pinned by tests/test_gpu_encode.py); `cpu_port` = the first 256 of the same windows through
    the C restatement on min(cores, 32) threads.
    `mid_compress_*` (n = 16 / 256, size = 64 KiB): a batch of FEW buffers -- the chunks of one archive, a directory of files --, where one wavefront per
    buffer leaves the GPU idle: parse and emitter over segments (csrc/alz_encode_seg.h)."""
    fmt = A.FORMAT_NAMES.index(fmt_name)
    from auroralib.compression_amd import formats as F
    lz = F.LZSS(A.LzProperties.from_bits(10, 6, 2))
    bmp = np.frombuffer(lz.Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read()), dtype=np.uint8)
    starts = [(i * (len(bmp) - size)) // (n - 1) for i in range(n)]
    raw = np.empty(n * size + 64, dtype=np.uint8)
    for i, s0 in enumerate(starts):
        raw[i * size:(i + 1) * size] = bmp[s0:s0 + size]
    raw[n * size:] = 0
    cap = size + size // 4 + 64
    st = (A.Stream * n)()
    r = synth.stream_records(st)
    r["src_off"], r["src_len"] = np.arange(n, dtype=np.uint64) * np.uint64(size), size
    r["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64((cap + 255) // 256 * 256)
    r["dst_cap"], r["format"] = cap, fmt
    dst_bytes = int(r["dst_off"][-1]) + cap + 64
    d_src, d_dst = ctx.malloc(raw.nbytes + 64), ctx.malloc(dst_bytes)
    try:
        ctx.h2d(d_src, raw)
        ctx.encode_batch_device(st, d_src, raw.nbytes, d_dst, dst_bytes, quality=quality)
        ms = []
        for _ in range(3):
            res, aux = ctx.encode_batch_device(st, d_src, raw.nbytes, d_dst, dst_bytes, quality=quality)
            ms.append(ctx.last_kernel_ms())
        rr = synth.result_records(res)
        ok = bool((rr["status"] == 0).all())
        for i in range(0, n, max(1, n // 8)):
            comp = bytes(ctx.d2h(d_dst, int(rr["dst_len"][i]), offset=int(r["dst_off"][i])))
            sized = fmt_name not in ("lz4_block", "prs_be", "lzo", "snappy_raw")
            got, dr = ctx.decode(fmt, comp, decom_len=size if sized else 0, cap=size, aux0=aux[i].aux0, aux1=aux[i].aux1)
            ok = ok and dr.status == 0 and got == bytes(raw[i * size:(i + 1) * size])
    finally:
        ctx.free(d_src); ctx.free(d_dst)
        ctx.release_scratch()
    best = min(ms)
    comp_total = int(rr["dst_len"].astype(np.int64).sum())
    cpu_port = None
    if with_cpu:
        try:
            cpu_port = cpu_port_encode(np, A, raw, r, min(n, 256), fmt, quality)
        except Exception as e:
            cpu_port = {"error": repr(e)}
    return {"name": "%s_%s_q%d" % (prefix, fmt_name, quality) + ("" if n == 10000 else "_%dx%dk" % (n, size >> 10)),
            "workload": "%s compression of %d windows of %d KiB of Test.bmp at quality %d, device-resident (ratio %.3f)"
                        % (fmt_name, n, size >> 10, quality, comp_total / (n * size)),
            "value": round(n * size / (best * 1e-3) / 2**30, 3), "unit": "GiB/s of raw input (kernels)", "kernel_ms": round(best, 3), "parity_ok": ok,
            "roofline": roofline(n * size + comp_total, best), "cpu_port": cpu_port}


if __name__ == "__main__":
    main()
