"""ONE small stream (8 KiB .. 256 KiB of Test.bmp, quality 8) through alz_decode: the whole-GPU path (threshold lowered) against the wavefront
kernels -- where the two cross."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import Context
bmp = O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0]
c = Context(0)
for fname in sys.argv[1:] or ["yaz0", "lz10", "yay0", "lz4_block", "prs_be"]:
    fmt = A.FORMAT_NAMES.index(fname)
    for n in (8192, 16384, 32768, 49152, 65536, 98304, 131072, 262144):
        raw = bytes(bmp[200000:200000 + n])
        comp, aux = O.encode_stream(fmt, raw, quality=8)
        sized = fname not in ("lz4_block", "prs_be", "prs_le", "lzo", "snappy_raw")
        row = []
        for mode in ("big", "wave"):
            c.big_stream(4096 if mode == "big" else 0xFFFFFFFF)
            before = c.big_stream()
            c.decode(fmt, comp, decom_len=n if sized else 0, cap=n, aux0=aux.aux0, aux1=aux.aux1)
            t0 = time.perf_counter()
            for _ in range(10): got, r = c.decode(fmt, comp, decom_len=n if sized else 0, cap=n, aux0=aux.aux0, aux1=aux.aux1)
            row.append(((time.perf_counter() - t0) / 10 * 1e3, got == raw and r.status == 0, c.big_stream() - before))
        print("%-9s %7d B (%6d compressed): whole-GPU %.3f ms (taken %d, ok %s) | wavefront kernels %.3f ms (ok %s)" % (fname, n, len(comp), row[0][0], row[0][2], row[0][1], row[1][0], row[1][1]), flush=True)
c.big_stream(24 << 10)
