"""usage (GPU box): python tools/big_stream.py  -- ONE stream of 256 KiB / 1 000 KiB / 4 MiB (the reference's benchmark: one 1 000 KiB stream of
Test.bmp, Benchmarks/Benchmarks/TestAllAlgorithms.cs:41-42) as Yay0 / MIO0 / Yaz0 / LZ10 / LZ11 / LZSS: device time and alz_decode wall time with the whole-GPU path
(csrc/alz_big.hip) and with the production kernel alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context, Plan

bmp = O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0]
c = Context(0)
for fmt in (A.FMT_YAY0, A.FMT_MIO0, A.FMT_YAZ0, A.FMT_LZ10, A.FMT_LZ11, A.FMT_LZSS, A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_LZO):
    for label, raw in (("Test.bmp[0:256 KiB]", bmp[:262144]), ("Test.bmp[0:1 000 KiB]", bmp[:1024000]), ("Test.bmp x 4 (4 MiB)", (bmp * 4)[:4 << 20])):
        comp, aux = O.encode_stream(fmt, raw, quality=8)
        n = len(raw)
        sized = fmt not in (A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZO)       # (those two carry no size in the descriptor: the destination's room bounds them)
        st = (A.Stream * 1)(A.Stream(0, 0, len(comp), n, n if sized else 0, aux.aux0, aux.aux1, fmt))
        src = np.frombuffer(comp + bytes(64), dtype=np.uint8)
        d_src, d_dst = c.malloc(src.nbytes), c.malloc(n + 64)
        c.h2d(d_src, src)
        out = {}
        for mode, thr in (("whole GPU", 64 << 10), ("one / two wavefronts", 0xFFFFFFFF)):
            c.big_stream(thr)
            p = Plan(c, st)
            p.execute(d_src, d_dst); c.synchronize()
            ms = p.execute_timed(d_src, d_dst, iters=10)
            ok = p.results()[0].status == 0 and bytes(c.d2h(d_dst, n)) == raw
            p.close()
            c.decode(fmt, comp, decom_len=n if sized else 0, cap=n, aux0=aux.aux0, aux1=aux.aux1)
            t0 = time.perf_counter()
            for _ in range(10):
                c.decode(fmt, comp, decom_len=n if sized else 0, cap=n, aux0=aux.aux0, aux1=aux.aux1)
            wall = (time.perf_counter() - t0) / 10 * 1e3
            out[mode] = (ms, wall, ok)
        print("%-5s %-22s ratio %.3f | " % (A.FORMAT_NAMES[fmt], label, len(comp) / n) + " | ".join(
            "%s: device %.3f ms = %.2f GiB/s, alz_decode %.3f ms = %.2f GiB/s, ok %s" % (m, v[0], n / v[0] / 2**30 * 1e3, v[1], n / v[1] / 2**30 * 1e3, v[2]) for m, v in out.items()), flush=True)
        c.free(d_src); c.free(d_dst)
