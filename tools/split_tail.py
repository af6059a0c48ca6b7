"""usage (GPU box): python tools/split_tail.py [fmt]  -- ONE batch of 10 000 x 256 KiB as two launches started together: the first N1 streams on
the one-wavefront kernel, the rest on the two-wavefront kernel (experiment: does the second shape fill the tail of the launch?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context, Plan

fmt = A.FORMAT_NAMES.index(sys.argv[1] if len(sys.argv) > 1 else "yaz0")
n, size = 10000, 262144
b = synth.make_batch(fmt, n, size, synth.seed_for(1))
ca, cb = Context(0), Context(0)
d_src = ca.malloc(b.src.nbytes + 64); d_dst = ca.malloc(b.dst_bytes + 64)
ca.h2d(d_src, b.src)

def sub(streams, lo, hi):
    k = hi - lo
    s = (A.Stream * k)()
    import ctypes as C
    C.memmove(s, C.byref(streams, lo * C.sizeof(A.Stream)), k * C.sizeof(A.Stream))
    return s

def timeit(fn, reps=20):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3

whole = Plan(ca, b.streams)
def one():
    whole.execute(d_src, d_dst); ca.synchronize()
print("one launch, library's choice: %.3f ms" % timeit(one), flush=True)
for n1 in (10000, 8192, 7168, 6144, 5120):
    for va, vb in ((1, 2), (1, 1)):
        if n1 == n and vb == 1: continue
        ca.set_kernel_variant(va); cb.set_kernel_variant(vb)
        pa = Plan(ca, sub(b.streams, 0, n1))
        pb = Plan(cb, sub(b.streams, n1, n)) if n1 < n else None
        def two():
            pa.execute(d_src, d_dst)
            if pb: pb.execute(d_src, d_dst)
            ca.synchronize()
            if pb: cb.synchronize()
        print("first %5d on shape %d, rest on shape %d: %.3f ms" % (n1, va, vb, timeit(two)), flush=True)
        pa.close()
        if pb: pb.close()
ca.set_kernel_variant(0)
res = synth.result_records(whole.results())
print("ok", bool((res["status"] == 0).all()))
