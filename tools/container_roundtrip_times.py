"""usage (GPU box): python tools/container_roundtrip_times.py  -- Compress / Decompress of the format classes on Test.bmp (1 MB) and sixteen copies of it, host buffers in and out, wall clock: a survey
for outliers (round 6: it found LZ4 frames and legacy files decoding at 150-290 MB/s)."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from auroralib.compression_amd import _abi as A, formats as F
bmp = F.LZSS(A.LzProperties.from_bits(10, 6, 2)).Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read())
def t(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t0) * 1e3)
    return r, min(ts)
for name, make, data in (("Snappy 16 MB", F.Snappy, bmp * 16), ("LZ4Legacy 16 MB", F.LZ4Legacy, bmp * 16), ("LZ4 1 MB (4 MiB block)", F.LZ4, bmp), ("Yaz0 1 MB", F.Yaz0, bmp), ("LZ10 1 MB", F.LZ10, bmp),
                         ("LZ11 1 MB", F.LZ11, bmp), ("LZO 1 MB", F.LZO, bmp), ("PRS 1 MB", F.PRS, bmp), ("LZ77 1 MB", F.LZ77, bmp), ("Yay0 1 MB", F.Yay0, bmp), ("MIO0 1 MB", F.MIO0, bmp),
                         ("LZSS 1 MB", F.LZSS, bmp), ("Yaz0 16 MB", F.Yaz0, bmp * 16), ("LZ4 16 MB (4 MiB blocks)", F.LZ4, bmp * 16)):
    try:
        f = make()
        comp, tc = t(lambda: f.Compress(data), 3)
        back, td = t(lambda: make().Decompress(comp), 5)
        print("%-28s compress %8.2f ms (%6.0f MB/s)  decompress %8.2f ms (%6.0f MB/s)  %s" % (name, tc, len(data) / tc / 1e3, td, len(data) / td / 1e3, "ok" if back == data else "MISMATCH"), flush=True)
    except Exception as e:
        print(name, "error", repr(e)[:100], flush=True)
