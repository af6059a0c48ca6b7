"""usage (GPU box): python tools/encode_overlap.py [format] [quality]
The synthetic encoder batch (10 000 x 256 KiB) as ONE call against the same buffers dealt out to 2 / 4 contexts (each its own HIP stream) that encode
at the same time from host threads: does kernel A of one slice (LDS-bound: one workgroup per CU) run beside the parse + emit kernel of another?
(An experiment for docs/EXPERIMENTS.md; the library does not slice.)"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from auroralib.compression_amd.batch import Context, Plan

fmt_name = sys.argv[1] if len(sys.argv) > 1 else "lzss"
q = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fmt = A.FORMAT_NAMES.index(fmt_name)
n, target = 10000, 262144
b = synth.make_batch(A.FMT_LZSS, n, target, synth.seed_for(5))
c0 = Context(0)
d_src = c0.malloc(b.src.nbytes + 64); d_raw = c0.malloc(b.dst_bytes + 64)
c0.h2d(d_src, b.src)
pl = Plan(c0, b.streams); pl.execute(d_src, d_raw); c0.synchronize()
recs = synth.stream_records(b.streams)
cap = target + target // 4 + 64; capal = (cap + 255) // 256 * 256
d_out = c0.malloc(n * capal + 64)

def records(idx):
    st = (A.Stream * len(idx))()
    r = synth.stream_records(st)
    r["src_off"], r["src_len"] = recs["dst_off"][idx], target
    r["dst_off"] = idx.astype(np.uint64) * np.uint64(capal)
    r["dst_cap"], r["format"] = cap, fmt
    return st

def run(slices, reps=5):
    ctxs = [Context(0) for _ in range(slices)]
    parts = [records(np.arange(k, n, slices)) for k in range(slices)]
    def work(k):
        ctxs[k].encode_batch_device(parts[k], d_raw, b.dst_bytes, d_out, n * capal + 64, quality=q)
    def once():
        th = [threading.Thread(target=work, args=(k,)) for k in range(slices)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return (time.perf_counter() - t0) * 1e3
    once(); once()
    ms = sorted(once() for _ in range(reps))
    for c in ctxs: c.close()
    return ms[0], ms[len(ms) // 2]

for s in (1, 2, 4, 8):
    best, med = run(s)
    print("%s q%d, %d slice(s) at once: wall %.2f ms best, %.2f median" % (fmt_name, q, s, best, med), flush=True)
