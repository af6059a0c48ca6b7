"""usage (GPU box): python tools/degenerate_encode_time.py  -- device-resident encode calls over DEGENERATE buffers (all zeros, runs of 20-60 thousand equal bytes, a 7-byte period) as LZ11, LZ4 blocks and LZO
at quality 8: 64 x 256 KiB, 256 x 64 KiB, 8 x 4 MiB (the whole-GPU path off).  What the speculative walk of csrc/alz_encode_seg_seq.h costs where every position of a buffer is capped in kernel B and the true
cursor jumps over whole segments (ALZ_SEG=0 in the environment: the segmented paths off)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import Context
c = Context(0); c.big_stream(0xFFFFFFFF)
if os.environ.get("ALZ_SEG") is not None: c.lib.alz_debug_seg_max_streams(c.h, int(os.environ["ALZ_SEG"]))
rng = np.random.default_rng(1)
for fmt in (A.FMT_LZ11, A.FMT_LZ4_BLOCK, A.FMT_LZO):
  for n, size in ((64, 262144), (256, 65536), (8, 4 << 20)):
    for kind in ("zeros", "runs", "period7"):
        if kind == "zeros": one = bytes(size)
        elif kind == "runs": one = b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(20000, 60000)) for _ in range(size // 20000 + 1))[:size]
        else: one = (bytes(rng.integers(0, 256, 7, dtype=np.uint8)) * (size // 7 + 1))[:size]
        raw = np.frombuffer(one * n + bytes(64), dtype=np.uint8)
        cap = size + size // 4 + 64; pitch = (cap + 255) // 256 * 256
        st = (A.Stream * n)()
        for i in range(n): st[i] = A.Stream(i * size, i * pitch, size, cap, 0, 0, 0, fmt)
        d_src, d_dst = c.malloc(raw.nbytes + 64), c.malloc(n * pitch + 64)
        c.h2d(d_src, raw)
        ms = []
        for q in (8,):
            for _ in range(3):
                res, aux = c.encode_batch_device(st, d_src, raw.nbytes, d_dst, n * pitch + 64, quality=q); ms.append(c.last_kernel_ms())
        print(A.FORMAT_NAMES[fmt], n, "x", size, kind, "q8: %.2f ms" % min(ms), "ok" if all(res[i].status == 0 for i in range(n)) else "FAIL", flush=True)
        c.free(d_src); c.free(d_dst)
