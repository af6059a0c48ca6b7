"""usage (GPU box): python tools/ragged_batch_encode.py -- one 8 MB stream and 5 000 streams of 1 KB in one alz_encode_batch call: launch grids are sized by the longest stream"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A, synth
from gpu_common import ctx
rng = np.random.default_rng(9)
big = rng.integers(0, 6, 8_000_000, dtype=np.uint8)
small = [rng.integers(0, 6, 1000, dtype=np.uint8) for _ in range(5000)]
raws = [big] + small
n = len(raws)
streams = (A.Stream * n)()
so = do = 0; chunks = []
for i, r in enumerate(raws):
    cap = len(r) + len(r) // 4 + 64
    streams[i] = A.Stream(so, do, len(r), cap, 0, 0, 0, A.FMT_LZSS)
    pad = (-len(r)) % 16
    chunks.append(bytes(r) + bytes(pad)); so += len(r) + pad; do += (cap + 15) // 16 * 16
src = np.frombuffer(b"".join(chunks) + bytes(64), dtype=np.uint8).copy()
for q in (0, 8):
    ctx().encode_batch(streams, src, do + 64, quality=q)
    t = time.time(); dst, res, aux = ctx().encode_batch(streams, src, do + 64, quality=q); dt = time.time() - t
    ok = all(res[i].status == 0 for i in range(n))
    want, _ = O.encode_stream(A.FMT_LZSS, bytes(raws[7]), quality=q)
    got = bytes(dst[streams[7].dst_off:streams[7].dst_off + res[7].dst_len])
    print("q%d: one 8 MB stream + 5000 x 1 KB: %.3f s (kernels %.1f ms), ok %s, sample equal %s" % (q, dt, ctx().last_kernel_ms(), ok, got == want), flush=True)
