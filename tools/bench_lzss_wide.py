#!/usr/bin/env python3
"""LZSS with 14 / 16 window bits (LzProperties.cs:57-66): kernel time of the lane-parallel kernel with HBM read-back against the
exact kernel (round 1 ran these geometries on the exact kernel only).  GPU box: python tools/bench_lzss_wide.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from auroralib.compression_amd import _abi as A, synth  # noqa: E402
from auroralib.compression_amd.batch import Context, Plan  # noqa: E402

ctx = Context(0)
for bits, lb in ((12, 4), (14, 4), (16, 8)):
    lz = A.LzProperties.from_bits(bits, lb, 2)
    n, size = 10000, 262144
    b = synth.make_batch(A.FMT_LZSS, n, size, synth.seed_for(2), lz=lz)
    d_src, d_dst = ctx.malloc(b.src.nbytes + 64), ctx.malloc(b.dst_bytes + 64)
    ctx.h2d(d_src, b.src)
    plan = Plan(ctx, b.streams, lz=lz)
    out = []
    for exact in (0, 1):
        ctx.set_exact_kernels(exact)
        plan.execute(d_src, d_dst); ctx.synchronize()
        ms = plan.execute_timed(d_src, d_dst, iters=3 if exact else 10)
        r = synth.result_records(plan.results())
        out.append((ms, bool((r["status"] == 0).all() and (r["dst_len"] == size).all())))
    ctx.set_exact_kernels(0)
    print("LZSS(%d,%d,2): lane-parallel %.2f ms = %.0f GiB/s (ok %s); exact kernel %.2f ms = %.0f GiB/s (ok %s)"
          % (bits, lb, out[0][0], n * size / out[0][0] / 1e-3 / 2**30, out[0][1], out[1][0], n * size / out[1][0] / 1e-3 / 2**30, out[1][1]))
    plan.close(); ctx.free(d_src); ctx.free(d_dst)
