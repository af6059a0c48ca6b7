"""ONE small stream (4 KiB .. 128 KiB of Test.bmp) through alz_encode_batch: the whole-GPU path (threshold lowered to 4 KiB) against the batch
pipeline -- where the two cross."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import Context
bmp = O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0]
c = Context(0)
for fname in sys.argv[1:] or ["lzss", "yaz0", "lz4_block"]:
    fmt = A.FORMAT_NAMES.index(fname)
    for q in (0, 8):
        for n in (4096, 8192, 16384, 32768, 65536, 98304, 131072):
            raw = bytes(bmp[200000:200000 + n])
            st = (A.Stream * 1)(A.Stream(0, 0, n, n + n // 4 + 64, 0, 0, 0, fmt))
            src = np.frombuffer(raw + bytes(64), dtype=np.uint8)
            row = []
            for mode in ("big", "batch"):
                c.big_stream(4096 if mode == "big" else 0xFFFFFFFF)
                c.encode_batch(st, src, n + n // 4 + 128, quality=q)
                t0 = time.perf_counter()
                for _ in range(5): d, r, a = c.encode_batch(st, src, n + n // 4 + 128, quality=q)
                row.append(((time.perf_counter() - t0) / 5 * 1e3, c.last_kernel_ms(), bytes(d[:r[0].dst_len])))
            want = O.encode_stream(fmt, raw, quality=q)[0]
            print("%-9s q%d %7d B: whole-GPU %.3f ms (kernels %.3f) | batch %.3f ms (kernels %.3f) | same bytes %s %s" % (fname, q, n, row[0][0], row[0][1], row[1][0], row[1][1], row[0][2] == want, row[1][2] == want), flush=True)
c.big_stream(24 << 10)
