"""usage (GPU box): python tools/lz4_frame_decode_time.py  -- LZ4 frames as the reference's writer produces them (the block-independence flag cleared, every block compressed on its own:
LZ4.Frame.cs:184, LZ4.cs:205) through F.LZ4().Decompress: host buffer in, bytes out, wall clock.  16 MB in 64 / 256 KiB blocks, 67 MB in 4 MiB blocks."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from auroralib.compression_amd import _abi as A, formats as F
bmp = F.LZSS(A.LzProperties.from_bits(10, 6, 2)).Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read())
ctx = F._context()
for mult, bs, bname in ((16, 0x10000, "64 KiB blocks"), (16, 0x40000, "256 KiB blocks"), (64, 0x400000, "4 MiB blocks")):
    data = bmp * mult
    comp = F.LZ4(bs).Compress(data, F.CompressionSettings.Balanced)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); back = F.LZ4().Decompress(comp); ts.append((time.perf_counter() - t0) * 1e3)
    assert back == data
    print("LZ4 frame decode, %d MB, %s: wall min / median %.2f / %.2f ms (%.0f MB/s), kernels of the last call %.2f ms" % (len(data) // 1000000, bname, min(ts), statistics.median(ts), len(data) / min(ts) / 1e3, ctx.last_kernel_ms()), flush=True)
