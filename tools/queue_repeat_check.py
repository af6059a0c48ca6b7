"""usage (GPU box): python tools/queue_repeat_check.py [FORMAT ...]  -- the work queue's repeat path (alz_plan_results: a bounded spin of alz_decode_fastq_kernel /
alz_decode_prs2q_kernel ran out -> the launch is repeated with one wavefront per stream).  Decodes 10 000-stream batches of the queue formats three times over, compares the whole
destination buffer with the oracle's and prints how often the repeat was taken.  With the product library: never.  With a library built with -DALZ_CHUNK_SPINS=0
(tools/build_variant.sh spin0 "-DALZ_CHUNK_SPINS=0" alz_kernels.hip) every wait that is not over at once runs out: the path is taken, the bytes must not change."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context, Plan

c = Context(0)
c.lib.alz_debug_chunk_repeats.restype = C.c_uint64; c.lib.alz_debug_chunk_repeats.argtypes = [C.c_void_p]
c.lib.alz_debug_plan_queue_items.argtypes = [C.c_void_p]
SHAPES = [(10000, 98304, 0), (8, 1 << 20, 3)]      # (streams, bytes each, kernel variant: 3 forces the queue on a batch too small to take it by itself -- there every chunk waits for the one before)
for name, (nstreams, each, variant) in [(f, s) for f in (sys.argv[1:] or ["yaz0", "lz10", "yay0", "prs_be"]) for s in SHAPES]:
    fmt = A.FORMAT_NAMES.index(name)
    c.set_kernel_variant(variant)
    b = synth.make_batch(fmt, nstreams, each, synth.seed_for(7))
    o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=8)
    d_src, d_dst = c.malloc(b.src.nbytes), c.malloc(b.dst_bytes)
    try:
        c.h2d(d_src, b.src)
        pl = Plan(c, b.streams)
        items = c.lib.alz_debug_plan_queue_items(pl.h)
        before, ok = c.lib.alz_debug_chunk_repeats(c.h), True
        for _ in range(3):
            c.memset(d_dst, 0, b.dst_bytes)
            pl.execute(d_src, d_dst)
            res = pl.results()
            g = c.d2h(d_dst, b.dst_bytes)
            ok = ok and bool(np.array_equal(g[:b.dst_bytes], o_dst[:b.dst_bytes])) and all((r.status, r.dst_len) == (q.status, q.dst_len) for r, q in zip(res, o_res))
        pl.close()
    finally:
        c.free(d_src); c.free(d_dst)
    print("%-8s %6d x %4d KiB, %d queue items: bytes and results identical to the oracle's %s, launches repeated without the queue %d" % (name, nstreams, each >> 10, items, ok, c.lib.alz_debug_chunk_repeats(c.h) - before), flush=True)
