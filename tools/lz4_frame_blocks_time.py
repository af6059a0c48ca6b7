"""usage (GPU box): python tools/lz4_frame_blocks_time.py  -- 64 copies of Test.bmp (67 MB) as ONE LZ4 frame of 1 MiB / 4 MiB blocks (4 MiB is the frame writer's default, LZ4.Frame.cs:107-174) at
Fastest / Balanced: kernel time of the call as the library chooses, with the segmented paths off and with the whole-GPU path off."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from auroralib.compression_amd import _abi as A, formats as F
from auroralib.compression_amd._lib import load
lib = load()
bmp = F.LZSS(A.LzProperties.from_bits(10, 6, 2)).Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read())
ctx = F._context()
data = bmp * 64
for bs, bname in ((0x100000, "1 MiB blocks"), (0x400000, "4 MiB blocks")):
    for s, sname in ((F.CompressionSettings.Fastest, "Fastest"), (F.CompressionSettings.Balanced, "Balanced")):
        row = []
        for mode in ("default", "segments off", "whole-GPU path off"):
            lib.alz_debug_seg_max_streams(ctx.h, 0 if mode == "segments off" else 0xFFFFFFFF)
            ctx.big_stream(0xFFFFFFFF if mode == "whole-GPU path off" else 24 << 10)
            f = F.LZ4(bs); out = f.Compress(data, s)
            ks = []
            for _ in range(3):
                out = f.Compress(data, s); ks.append(ctx.last_kernel_ms())
            row.append("%s %.2f ms" % (mode, statistics.median(ks)))
        assert F.LZ4().Decompress(out) == data
        print("LZ4 frame of 67 MB,", bname, sname, ": kernels", " | ".join(row), flush=True)
lib.alz_debug_seg_max_streams(ctx.h, 0xFFFFFFFF); ctx.big_stream(24 << 10)
