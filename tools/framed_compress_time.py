"""usage (GPU box): python tools/framed_compress_time.py  -- what a FRAMED container's Compress costs end to end (host buffers in, container bytes out, wall clock): raw Snappy chunks of 64 KiB
behind the framing format (Snappy.cs:86: a new finder per chunk) are a batch of tens to hundreds of buffers -- the segmented encode of csrc/alz_encode_seg.h -- with the path on and off.
Round 6: the LZ4 frame writer too (LZ4.Frame.cs:107-174: blocks of 64 KiB - 4 MiB, each compressed on its own), whose blocks take the speculative walk per segment of csrc/alz_encode_seg_seq.h."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from auroralib.compression_amd import _abi as A, formats as F
from auroralib.compression_amd._lib import load
lib = load()
bmp = F.LZSS(A.LzProperties.from_bits(10, 6, 2)).Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read())
import statistics
ctx = F._context()


def measure(make, data, s, check):
    """Path off / on in turns (the host side of a call -- page faults of fresh buffers, the allocator's state -- moves by milliseconds from call to call: eleven calls each, minimum and
    median of the wall clock, and the kernels' own time, which is what the path changes)."""
    rows = {0: ([], []), 0xFFFFFFFF: ([], [])}
    size = None
    for seg in (0, 0xFFFFFFFF):
        lib.alz_debug_seg_max_streams(ctx.h, seg)
        out = make().Compress(data, s)
        assert check(out) == data
        assert size is None or size == len(out)
        size = len(out)
    for _ in range(11):
        for seg in (0, 0xFFFFFFFF):
            lib.alz_debug_seg_max_streams(ctx.h, seg)
            f = make()
            t0 = time.perf_counter(); out = f.Compress(data, s); rows[seg][0].append((time.perf_counter() - t0) * 1e3); rows[seg][1].append(ctx.last_kernel_ms())
    lib.alz_debug_seg_max_streams(ctx.h, 0xFFFFFFFF)
    a, b = rows[0], rows[0xFFFFFFFF]
    return "wall min / median %6.2f / %6.2f -> %6.2f / %6.2f ms, kernels %5.2f -> %5.2f ms (%d bytes either way)" % (
        min(a[0]), statistics.median(a[0]), min(b[0]), statistics.median(b[0]), statistics.median(a[1]), statistics.median(b[1]), size)


print("one wavefront per block -> segments")
for name, data in (("Test.bmp (1 MB, 16 chunks)", bmp), ("16 x Test.bmp (16 MB, 245 chunks)", bmp * 16)):
    for s, sname in ((F.CompressionSettings.Fastest, "Fastest"), (F.CompressionSettings.Balanced, "Balanced")):
        print("Snappy framing, %-34s %-8s: %s" % (name, sname, measure(F.Snappy, data, s, lambda o: F.Snappy().Decompress(o))), flush=True)
for bs, bname in ((0x10000, "64 KiB blocks"), (0x40000, "256 KiB blocks")):
    for name, data in (("Test.bmp (1 MB)", bmp), ("16 x Test.bmp (16 MB)", bmp * 16)):
        for s, sname in ((F.CompressionSettings.Fastest, "Fastest"), (F.CompressionSettings.Balanced, "Balanced")):
            print("LZ4 frame, %-14s %-22s %-8s: %s" % (bname, name, sname, measure(lambda: F.LZ4(bs), data, s, lambda o: F.LZ4().Decompress(o))), flush=True)
