"""usage (GPU box): python tools/bench_single.py  -- latency of ONE stream through the host-buffer ABI (alz_decode: what
`Decompress(Stream, Stream)` of the shim binds): upload + plan + kernel + download, per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context

ctx = Context(0)
for fmt in (A.FMT_YAZ0, A.FMT_LZ10, A.FMT_LZ4_BLOCK, A.FMT_PRS_BE):
    for size in (1024, 65536, 262144, 4 << 20):
        b = synth.make_batch(fmt, 1, size, 1234)
        r = synth.stream_records(b.streams)
        src = bytes(b.src[:int(r["src_len"][0])])
        a0, a1 = int(r["aux0"][0]), int(r["aux1"][0])
        out, res = ctx.decode(fmt, src, decom_len=size, aux0=a0, aux1=a1)
        assert res.status == 0 and len(out) == size
        n = 50 if size <= 262144 else 10
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.decode(fmt, src, decom_len=size, aux0=a0, aux1=a1)
        dt = (time.perf_counter() - t0) / n
        print("%-10s %8d B  %8.1f us per call  %7.3f GiB/s" % (A.FORMAT_NAMES[fmt], size, dt * 1e6, size / dt / 2**30), flush=True)
