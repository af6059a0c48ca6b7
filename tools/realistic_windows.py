"""usage (GPU box): python tools/realistic_windows.py [N]  -- kernel time of 10 000 copies of ONE Test.bmp window (Yaz0, Q8), for
every 16th window: how much the windows differ, i.e. how uneven the streams of the realistic batch are."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from auroralib.compression_amd import _abi as A, synth, formats as F
from auroralib.compression_amd.batch import Context, Plan

ctx = Context(0)
lz = F.LZSS(A.LzProperties.from_bits(10, 6, 2))
bmp = np.frombuffer(lz.Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read()), dtype=np.uint8)
size, stride = 262144, 4096
starts = list(range(0, len(bmp) - size + 1, stride))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for w in range(0, len(starts), 16):
    raw = bmp[starts[w]:starts[w] + size]
    cap = size + size // 4 + 64
    st = (A.Stream * 1)()
    r = synth.stream_records(st)
    r["src_off"], r["src_len"], r["dst_off"], r["dst_cap"], r["format"] = 0, size, 0, cap, A.FMT_YAZ0
    enc, eres, aux = ctx.encode_batch(st, raw, cap + 64, quality=8)
    clen = int(synth.result_records(eres)["dst_len"][0])
    src = np.zeros(clen + 64, dtype=np.uint8); src[:clen] = enc[:clen]
    streams = (A.Stream * n)()
    s2 = synth.stream_records(streams)
    s2["src_off"], s2["src_len"] = 0, clen
    s2["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64(size)
    s2["dst_cap"], s2["decom_len"], s2["format"] = size, size, A.FMT_YAZ0
    d_src = ctx.malloc(src.nbytes + 64); d_dst = ctx.malloc(n * size + 64)
    ctx.h2d(d_src, src)
    p = Plan(ctx, streams)
    p.execute(d_src, d_dst); ctx.synchronize()
    ms = p.execute_timed(d_src, d_dst, iters=5)
    res = synth.result_records(p.results())
    ok = bool((res["status"] == 0).all()) and bool(np.array_equal(ctx.d2h(d_dst, size), raw))
    print("window %3d  ratio %.3f  kernel %.3f ms  %.0f GiB/s  ok %s" % (w, clen / size, ms, n * size / ms / 2**30 * 1e3, ok), flush=True)
    p.close(); ctx.free(d_src); ctx.free(d_dst)
