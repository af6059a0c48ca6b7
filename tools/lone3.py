"""usage (GPU box): python tools/lone3.py  -- ONE stream alone on the GPU per flag-family format, one- and two-wavefront kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),"tests"))
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context
ctx = Context(0)
for fmt in (A.FMT_YAY0, A.FMT_MIO0, A.FMT_YAZ0):
    for size in (262144, 4 << 20):
        b = synth.make_batch(fmt, 1, size, 1234)
        r = synth.stream_records(b.streams)
        src = bytes(b.src[:int(r["src_len"][0])])
        a0, a1 = int(r["aux0"][0]), int(r["aux1"][0])
        out, res = ctx.decode(fmt, src, decom_len=size, aux0=a0, aux1=a1)
        assert res.status == 0 and len(out) == size
        n = 50
        t0 = time.perf_counter()
        for _ in range(n): ctx.decode(fmt, src, decom_len=size, aux0=a0, aux1=a1)
        dt = (time.perf_counter() - t0) / n
        print("%-10s %8d B  %8.1f us per call" % (A.FORMAT_NAMES[fmt], size, dt * 1e6), flush=True)
