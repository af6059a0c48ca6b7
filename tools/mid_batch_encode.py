"""Encoder throughput against the number of streams in a batch (256 KiB windows of Test.bmp, device-resident): where the batch pipeline --
one workgroup (kernel A) and one wavefront (parse + emission) per stream -- leaves the GPU idle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from auroralib.compression_amd.batch import Context

bmp = np.frombuffer(O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0], dtype=np.uint8)
if os.environ.get("ALZ_MID_DATA") == "text":              # the repository's own sources and documents: program text and prose instead of a bitmap
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "**", "*.md"), recursive=True) + glob.glob(os.path.join(ROOT, "**", "*.hip"), recursive=True) +
                   glob.glob(os.path.join(ROOT, "**", "*.h"), recursive=True) + glob.glob(os.path.join(ROOT, "**", "*.py"), recursive=True) +
                   glob.glob(os.path.join(ROOT, "**", "*.cs"), recursive=True) + glob.glob(os.path.join(ROOT, "**", "*.cpp"), recursive=True))
    bmp = np.frombuffer(b"".join(open(f, "rb").read() for f in files if "gpurun_out" not in f), dtype=np.uint8)
    print("text corpus: %d files, %d bytes" % (len(files), len(bmp)))
size = int(os.environ.get("ALZ_MID_SIZE", str(262144)))
c = Context(0)
import ctypes as C
c.lib.alz_debug_seg_launches.restype = C.c_uint64; c.lib.alz_debug_seg_launches.argtypes = [C.c_void_p]
if os.environ.get("ALZ_MID_BIG") == "off": c.big_stream(0xFFFFFFFF)                                # (the whole-GPU path of ONE buffer at a time off)
if os.environ.get("ALZ_MID_SEG") is not None: c.lib.alz_debug_seg_max_streams(c.h, int(os.environ["ALZ_MID_SEG"]))      # (0: the segmented parse + emit off)
for fname in sys.argv[1:] or ["lzss", "yaz0"]:
    fmt = A.FORMAT_NAMES.index(fname)
    for q in [int(x) for x in os.environ.get("ALZ_MID_Q", "0,8").split(",")]:
        for n in [int(x) for x in os.environ.get("ALZ_MID_N", "1,8,32,33,64,128,256,512,1024,2048,4096").split(",")]:
            starts = [(i * 4096) % (len(bmp) - size) for i in range(n)]
            raw = np.concatenate([bmp[s:s + size] for s in starts])
            cap = size + size // 4 + 64
            st = (A.Stream * n)()
            r = synth.stream_records(st)
            r["src_off"], r["src_len"] = np.arange(n, dtype=np.uint64) * np.uint64(size), size
            r["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64((cap + 255) // 256 * 256)
            r["dst_cap"], r["format"] = cap, fmt
            dst_bytes = int(r["dst_off"][-1]) + cap + 64
            d_src, d_dst = c.malloc(raw.nbytes + 64), c.malloc(dst_bytes)
            try:
                c.h2d(d_src, raw)
                before = c.big_stream(); seg0 = c.lib.alz_debug_seg_launches(c.h)
                c.encode_batch_device(st, d_src, raw.nbytes, d_dst, dst_bytes, quality=q)
                ms, wall = [], []
                for _ in range(3):
                    t0 = time.perf_counter()
                    res, aux = c.encode_batch_device(st, d_src, raw.nbytes, d_dst, dst_bytes, quality=q); ms.append(c.last_kernel_ms())
                    wall.append((time.perf_counter() - t0) * 1e3)
                big = c.big_stream() - before; seg = c.lib.alz_debug_seg_launches(c.h) - seg0
            finally:
                c.free(d_src); c.free(d_dst)
            ok = all(x.status == 0 for x in res)
            print("%-6s q%d %5d x %d KiB: %8.3f ms kernels = %7.2f GiB/s  (whole-GPU path: %s, segments: %s, ok %s; the call: %.3f ms)" % (fname, q, n, size >> 10, min(ms), n * size / min(ms) / 2**30 * 1e3, big > 0, seg > 0, ok, min(wall)), flush=True)
