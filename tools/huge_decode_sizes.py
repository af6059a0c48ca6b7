"""ONE stream of 1 / 4 / 16 / 64 MiB (Test.bmp tiled) through the whole-GPU decode path, device-resident: kernel time against size."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import Context, Plan
bmp = O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0]
c = Context(0)
for mib in (1, 4, 16, 64):
    raw = bytes((bmp * (mib + 1))[:mib << 20]); n = len(raw)
    for fname in ("yaz0", "lz4_block"):
        fmt = A.FORMAT_NAMES.index(fname)
        st = (A.Stream * 1)(A.Stream(0, 0, n, n + n // 4 + 64, 0, 0, 0, fmt))
        d, r, a = c.encode_batch(st, np.frombuffer(raw + bytes(64), dtype=np.uint8), n + n // 4 + 128, quality=0)
        comp = bytes(d[:r[0].dst_len])
        sized = fname == "yaz0"
        ds = (A.Stream * 1)(A.Stream(0, 0, len(comp), n, n if sized else 0, 0, 0, fmt))
        d_src, d_dst = c.malloc(len(comp) + 64), c.malloc(n + 64)
        c.h2d(d_src, np.frombuffer(comp + bytes(64), dtype=np.uint8))
        p = Plan(c, ds); p.execute(d_src, d_dst); c.synchronize()
        ms = p.execute_timed(d_src, d_dst, iters=5)
        ok = p.results()[0].status == 0 and bytes(c.d2h(d_dst, n)) == raw
        p.close(); c.free(d_src); c.free(d_dst)
        print("%-9s %3d MiB: decode kernels %.3f ms = %.1f GiB/s (ok %s)" % (fname, mib, ms, n / ms / 2**30 * 1e3, ok), flush=True)
