"""usage (GPU box, with build/variants/jump_stats.so built from tools/variants/r04_jump_stats.patch):
    python tools/jump_stats.py [fmt ...]
Histogram of pointer-jumping rounds per 64-byte step of the byte phase (alz_emit_byte.h): the synthetic mix of the bench and, for Yaz0
/ LZ10 / LZ11 / PRS, three kinds of 256 KiB Test.bmp windows (photographic, mixed, flat).  The variant library returns the histogram in
place of the result fields, so everything goes through the device-resident plan path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from auroralib.compression_amd import _lib
_lib.SO_PATH = os.path.join(ROOT, "build", "variants", "jump_stats.so")
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context, Plan
import oracle_lib as O


def hist(ctx, streams, src, dst_bytes):
    d_src, d_dst = ctx.malloc(src.nbytes + 64), ctx.malloc(dst_bytes + 64)
    ctx.h2d(d_src, src)
    p = Plan(ctx, streams)
    p.execute(d_src, d_dst); ctx.synchronize()
    r = synth.result_records(p.results())
    p.close(); ctx.free(d_src); ctx.free(d_dst)
    lo = r["dst_len"].astype(np.uint64) | (r["src_used"].astype(np.uint64) << np.uint64(32))
    res = r["reserved"].astype(np.uint64)
    h = [int(((lo >> np.uint64(13 * k)) & np.uint64(0x1FFF)).sum()) for k in range(4)]
    h4 = ((lo >> np.uint64(52)) & np.uint64(0xFFF)) | ((res & np.uint64(1)) << np.uint64(12))
    h.append(int(h4.sum())); h.append(int(((res >> np.uint64(1)) & np.uint64(0x1FFF)).sum())); h.append(int((res >> np.uint64(14)).sum()))
    return np.array(h, dtype=np.float64)


def show(name, h):
    tot = h.sum()
    if tot == 0:
        print("%-28s (no byte-phase steps counted)" % name); return
    mean = (h * np.arange(7)).sum() / tot
    print("%-28s steps %9d  rounds/step %.2f  | " % (name, tot, mean) + "  ".join("%d:%4.1f%%" % (k, 100 * h[k] / tot) for k in range(7)), flush=True)


ctx = Context(0)
fmts = sys.argv[1:] or ["yaz0", "lz10", "lz11", "lzss", "yay0", "mio0"]
bmp = np.frombuffer(O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0], dtype=np.uint8)
for f in fmts:
    fmt = A.FORMAT_NAMES.index(f)
    b = synth.make_batch(fmt, 512, 262144, synth.seed_for(2))
    show(f + " synthetic", hist(ctx, b.streams, b.src, b.dst_bytes))
    for kind, w in (("photographic (window 0)", 0), ("mixed (window 40)", 40), ("flat (window 96)", 96), ("flat (window 176)", 176)):
        raw = bytes(bmp[w * 4096:w * 4096 + 262144])
        comp, aux = O.encode_stream(fmt, raw, quality=8)
        n = 64
        streams = (A.Stream * n)()
        for i in range(n):
            streams[i] = A.Stream(0, i * 262144, len(comp), 262144, 262144, aux.aux0, aux.aux1, fmt)
        src = np.frombuffer(comp + bytes(64), dtype=np.uint8)
        show("%s %s r=%.3f" % (f, kind, len(comp) / 262144), hist(ctx, streams, src, n * 262144))
