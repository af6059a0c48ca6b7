"""ONE stream of Test.bmp (1 000 KiB, the reference's benchmark input) through alz_encode_batch: the whole-GPU encode path against
the batch pipeline (the path switched off) and the C port of the managed encoder on one host core."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import Context

bmp = O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0]
if os.environ.get("ALZ_SINGLE_DATA") == "text":           # program text and prose: the repository's own sources and documents
    import glob
    files = sorted(f for pat in ("*.md", "*.hip", "*.h", "*.py", "*.cs", "*.cpp") for f in glob.glob(os.path.join(ROOT, "**", pat), recursive=True) if "gpurun_out" not in f)
    bmp = b"".join(open(f, "rb").read() for f in files)
raw = bytes(bmp[:1024000]); n = len(raw)
c = Context(0)
names = sys.argv[1:] or ["lzss", "lz10", "lz11", "yaz0", "yay0", "mio0"]
for fname in names:
    fmt = A.FORMAT_NAMES.index(fname)
    for q in [int(x) for x in os.environ.get("ALZ_SINGLE_Q", "0,8,15").split(",")]:
        st = (A.Stream * 1)(A.Stream(0, 0, n, n + n // 4 + 64, 0, 0, 0, fmt))
        src = np.frombuffer(raw + bytes(64), dtype=np.uint8)
        row = []
        for mode in os.environ.get("ALZ_SINGLE_MODES", "big,batch").split(","):
            c.big_stream(24 << 10 if mode == "big" else 0xFFFFFFFF)
            before = c.big_stream()
            c.encode_batch(st, src, n + n // 4 + 128, quality=q)
            t0 = time.perf_counter()
            for _ in range(3): d, r, a = c.encode_batch(st, src, n + n // 4 + 128, quality=q)
            wall = (time.perf_counter() - t0) / 3 * 1e3
            row.append((wall, c.last_kernel_ms(), bytes(d[:r[0].dst_len]), c.big_stream() - before, (a[0].aux0, a[0].aux1)))
        t0 = time.perf_counter(); want, waux = O.encode_stream(fmt, raw, quality=q); cpu = (time.perf_counter() - t0) * 1e3
        if len(row) < 2: row.append(row[0])
        print("%-6s q%-2d big: call %.2f ms (kernels %.2f, taken %d) = %.2f GiB/s | batch: call %.2f ms | C port one core %.2f ms = %.2f GiB/s | same bytes big %s batch %s aux %s"
              % (fname, q, row[0][0], row[0][1], row[0][3], n / row[0][0] / 2**30 * 1e3, row[1][0], cpu, n / cpu / 2**30 * 1e3,
                 row[0][2] == want, row[1][2] == want, row[0][4] == row[1][4]), flush=True)
