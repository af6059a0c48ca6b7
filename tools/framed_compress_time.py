"""usage (GPU box): python tools/framed_compress_time.py  -- what a FRAMED container's Compress costs end to end (host buffers in, container bytes out, wall clock): raw Snappy chunks of 64 KiB
behind the framing format (Snappy.cs:86: a new finder per chunk) are a batch of tens to hundreds of buffers -- the segmented encode of csrc/alz_encode_seg.h -- with the path on and off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from auroralib.compression_amd import _abi as A, formats as F
from auroralib.compression_amd._lib import load
lib = load()
bmp = F.LZSS(A.LzProperties.from_bits(10, 6, 2)).Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read())
for name, data in (("Test.bmp (1 MB, 16 chunks)", bmp), ("16 x Test.bmp (16 MB, 245 chunks)", bmp * 16)):
    for s, sname in ((F.CompressionSettings.Fastest, "Fastest"), (F.CompressionSettings.Balanced, "Balanced")):
        row = []
        for seg in (0, 0xFFFFFFFF):
            lib.alz_debug_seg_max_streams(F._context().h, seg)
            f = F.Snappy()
            out = f.Compress(data, s)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); out = f.Compress(data, s); ts.append((time.perf_counter() - t0) * 1e3)
            assert F.Snappy().Decompress(out) == data
            row.append((min(ts), len(out)))
        assert row[0][1] == row[1][1]
        print("Snappy framing, %-34s %-8s: %7.2f ms one wavefront per chunk -> %7.2f ms segments (%.0f -> %.0f MB/s of raw input; %d bytes either way)" %
              (name, sname, row[0][0], row[1][0], len(data) / row[0][0] / 1e3, len(data) / row[1][0] / 1e3, row[0][1]), flush=True)
