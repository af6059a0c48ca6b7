"""ONE stream of 64 MiB (Test.bmp tiled, a little noise every 1 MiB) through the whole-GPU encode and decode paths: sizes far beyond the tests',
checked against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import Context
bmp = O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0]
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(1)
raw = bytearray((bmp * (mib + 1))[:mib << 20])
for k in range(mib):
    raw[(k << 20) + 777:(k << 20) + 777 + 3000] = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
raw = bytes(raw); n = len(raw)
c = Context(0)
for fname in sys.argv[2:] or ["yaz0", "lz4_block", "prs_be"]:
    fmt = A.FORMAT_NAMES.index(fname)
    for q in (0, 8):
        st = (A.Stream * 1)(A.Stream(0, 0, n, n + n // 4 + 64, 0, 0, 0, fmt))
        before = c.big_stream()
        t0 = time.perf_counter()
        d, r, a = c.encode_batch(st, np.frombuffer(raw + bytes(64), dtype=np.uint8), n + n // 4 + 128, quality=q)
        t1 = time.perf_counter()
        comp = bytes(d[:r[0].dst_len])
        want, waux = O.encode_stream(fmt, raw, quality=q)
        t2 = time.perf_counter()
        sized = fname not in ("lz4_block", "prs_be", "prs_le", "lzo", "snappy_raw")
        got, dr = c.decode(fmt, comp, decom_len=n if sized else 0, cap=n, aux0=a[0].aux0, aux1=a[0].aux1)
        t3 = time.perf_counter()
        print("%-9s q%d %d MiB: encode %.1f ms (kernels %.1f), C port %.0f ms, same bytes %s; decode %.1f ms, round trip %s; whole-GPU paths taken %d"
              % (fname, q, mib, (t1 - t0) * 1e3, c.last_kernel_ms(), (t2 - t1) * 1e3, comp == want, (t3 - t2) * 1e3, got == raw and dr.status == 0, c.big_stream() - before), flush=True)
