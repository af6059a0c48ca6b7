#!/usr/bin/env python3
"""bench.py -- the hot path's headline measurement (BASELINE.json metric).

A "step" is one pass of the batched decode over one synthetic batch whose compressed payload is already resident in
HBM; output stays in HBM.  Default workload = the batch the metric is quoted on: 10 000 Yaz0 streams x 256 KiB per GPU
(`--stream-kib 64` gives BASELINE.json configs[1] exactly).  Multi-GPU: one process per GPU (torch.distributed.run),
every rank decodes its own batch (weak scaling, no data-path collective); value = bytes decoded by all ranks / max time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def measured_traffic(fmt, n, kib):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, gfx950 rule) for this
    exact workload, or None when it has not been profiled (profiles/traffic.json)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return t.get("%s:%d:%d" % (fmt, n, kib))
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--format", default="yaz0")
    ap.add_argument("--streams", type=int, default=10000)
    ap.add_argument("--stream-kib", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the byte comparison with the CPU baseline's output")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight per GPU: 2 (default) = double buffered on two HIP streams / two "
                    "output buffers, so the tail of one batch (few streams left, the chip half empty) overlaps the head of the next; "
                    "1 = the steps run back to back on one stream.  The back-to-back figure is always measured and reported too.")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests of the N>1 path)")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1, help="smoke test: every rank uses this GPU (needs --dist-backend gloo)")
    args = ap.parse_args()

    import numpy as np
    from auroralib.compression_amd import _abi as A
    from auroralib.compression_amd import synth
    from auroralib.compression_amd.batch import Context, Plan
    from auroralib.compression_amd.sharding import reduce_step_time, shard_seed, whole_job_value

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    target = args.stream_kib * 1024
    n = args.streams
    if args.format == "mixed":   # BASELINE.json configs[3]: LZ10/LZ11/Yaz0/PRS interleaved, per-format kernel dispatch
        fmt = np.array([[A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_PRS_BE][i % 4] for i in range(n)], dtype=np.uint32)
    else:
        fmt = A.FORMAT_NAMES.index(args.format)
    # seed = 0xA17A0000 + 1000*config + stream index; ranks get disjoint stream indices
    batch = synth.make_batch(fmt, n, target, shard_seed(2, rank, n))
    recs = synth.stream_records(batch.streams)
    comp_bytes = int(recs["src_len"].astype(np.int64).sum())
    decomp_bytes = int(n) * target

    ctx = Context(local_rank)
    d_src = ctx.malloc(batch.src.nbytes + 64)
    d_dst = ctx.malloc(batch.dst_bytes + 64)
    ctx.h2d(d_src, batch.src)
    ctx.memset(d_dst, 0, batch.dst_bytes)
    plan = Plan(ctx, batch.streams)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    lanes = [(ctx, plan, d_dst)]
    for _ in range(1, max(1, args.inflight)):          # extra pipelines: own HIP stream (context), own plan, own output buffer
        c2 = Context(local_rank)
        d2 = c2.malloc(batch.dst_bytes + 64)
        c2.memset(d2, 0, batch.dst_bytes)
        lanes.append((c2, Plan(c2, batch.streams), d2))

    def barrier_all():
        barrier()
        for c, _, _ in lanes[1:]:
            c.synchronize()

    def timed(nlanes):
        for i in range(args.warmup):
            c, pl, dd = lanes[i % nlanes]
            pl.execute(d_src, dd)
        barrier_all()
        t0 = time.perf_counter()
        for i in range(args.steps):
            c, pl, dd = lanes[i % nlanes]
            pl.execute(d_src, dd)
        barrier_all()
        return time.perf_counter() - t0

    dt_single = timed(1)                                # K steps back to back on one stream
    dt = timed(len(lanes)) if len(lanes) > 1 else dt_single
    dt_single = reduce_step_time(dt_single, dist, device="cuda" if (dist is not None and args.dist_backend == "nccl") else None)
    dt = reduce_step_time(dt, dist, device="cuda" if (dist is not None and args.dist_backend == "nccl") else None)

    # dominant kernel, HIP events on the launch stream (device time per launch)
    kernel_ms = plan.execute_timed(d_src, d_dst, iters=max(3, min(args.steps, 10)))

    # what was just measured decoded completely: every status OK, every length right (GPU results only)
    res = synth.result_records(plan.results())
    ok = bool((res["status"] == 0).all() and (res["dst_len"] == target).all())
    verified = None

    # cpu_baseline leg -- the only place that touches oracle/: the C restatement decodes the same batch on the host cores
    # (timed), and because it then holds the reference output anyway, the GPU's bytes are compared with it
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # (N = 1 only: a reported baseline, never part of the timed region)
        import oracle_lib as O
        cores = os.cpu_count() or 1
        o_dst = np.ones(batch.dst_bytes, dtype=np.uint8)  # pre-touched
        o_res = (A.Result * n)()
        reps, tcpu = 0, 0.0
        t1 = time.perf_counter()
        while reps < 1 or (tcpu < 10.0 and reps < 20):
            O.lib.oracle_decode_batch(None, n, batch.src.ctypes.data, batch.streams, o_dst.ctypes.data, o_res, cores)
            reps += 1
            tcpu = time.perf_counter() - t1
        cpu = {"value": round(decomp_bytes * reps / tcpu / 2**30, 3), "unit": "GiB/s", "cores": cores, "kind": "port",
               "sample": "full batch (%d x %d KiB %s) x %d passes, C restatement of the managed ring+flush path, %d host threads"
                         % (n, args.stream_kib, args.format, reps, cores)}
        if not args.no_verify:
            k = min(n, 1024)
            span = int(recs["dst_off"][k - 1]) + target
            g = ctx.d2h(d_dst, span)
            verified = True
            for i in range(k):                            # stream by stream (gaps between streams are not output)
                a = int(recs["dst_off"][i])
                if not np.array_equal(g[a:a + target], o_dst[a:a + target]):
                    verified = False
                    break
            ok = ok and verified

    if rank == 0:
        value = whole_job_value(decomp_bytes, world, args.steps, dt)
        algo_bytes = comp_bytes + decomp_bytes
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "decompressed GiB/s (whole job; 10k x 256KiB batch per GPU)",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s decode, %d x %d KiB synthetic streams per GPU (SURVEY 8d token-level generator, seed 0xA17A0000+2000+i), device-resident"
                                   % (args.format, n, args.stream_kib),
                       "format": args.format, "streams_per_gpu": n, "stream_bytes": target, "compressed_bytes_per_gpu": comp_bytes,
                       "parallelism": "stream-sharded x%d, no collective" % world, "batches_in_flight": len(lanes),
                       "back_to_back": {"value": round(whole_job_value(decomp_bytes, world, args.steps, dt_single), 3), "unit": "GiB/s",
                                        "ms_per_step": round(dt_single / args.steps * 1e3, 4), "batches_in_flight": 1},
                       "parity_ok": ok, "verified_vs_oracle": verified},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(args.format, n, args.stream_kib),
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": algo_bytes},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    for c, pl, dd in lanes[1:]:
        pl.close(); c.free(dd); c.close()
    plan.close()
    ctx.free(d_src)
    ctx.free(d_dst)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
