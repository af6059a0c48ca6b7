"""usage (GPU box): rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/big_stream_profile.py FORMAT QUALITY  -- one 1 000 KiB stream of
Test.bmp through the whole-GPU path, twenty times: which of its ~45 kernels take the time"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import Context, Plan
fmt = A.FORMAT_NAMES.index(sys.argv[1]); q = int(sys.argv[2])
bmp = O.container_decompress(A.C_LZSS, open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read(), lz=A.LzProperties.from_bits(10, 6, 2))[0]
raw = bmp[:1024000]; n = len(raw)
comp, aux = O.encode_stream(fmt, raw, quality=q)
sized = fmt not in (A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW)
c = Context(0)
st = (A.Stream * 1)(A.Stream(0, 0, len(comp), n, n if sized else 0, aux.aux0, aux.aux1, fmt))
d_src, d_dst = c.malloc(len(comp) + 64), c.malloc(n + 64)
c.h2d(d_src, np.frombuffer(comp + bytes(64), dtype=np.uint8))
p = Plan(c, st)
for _ in range(20):
    p.execute(d_src, d_dst)
c.synchronize()
print("status", p.results()[0].status, "ok", bytes(c.d2h(d_dst, n)) == raw, "ratio %.4f" % (len(comp) / n))
