"""usage (GPU box): python tools/big_stream_encode.py -- ONE stream of tens of megabytes (bitmap, noise, zeros, a period of 777) through the encoder, against the CPU restatement"""
import os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A
from gpu_common import ctx
rng = np.random.default_rng(5)
lz = open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read()
bmp, st = O.container_decompress(A.C_LZSS, lz, lz=A.LzProperties.from_bits(10, 6, 2))
bmp = np.frombuffer(bmp, dtype=np.uint8)
big = np.concatenate([bmp, rng.integers(0, 256, 3_000_000, dtype=np.uint8), np.zeros(5_000_000, dtype=np.uint8), bmp[::-1], np.tile(rng.integers(0, 256, 777, dtype=np.uint8), 9000)] * 2)
print("stream of %.1f MB" % (len(big) / 1e6))
for fmt, q in ((A.FMT_LZSS, 0), (A.FMT_YAZ0, 8), (A.FMT_LZ4_BLOCK, 4), (A.FMT_LZSS, 12)):
    raw = bytes(big if q < 10 else big[:6_000_000])
    streams = (A.Stream * 1)()
    cap = len(raw) + len(raw) // 4 + 64
    streams[0] = A.Stream(0, 0, len(raw), cap, 0, 0, 0, fmt)
    src = np.frombuffer(raw + bytes(64), dtype=np.uint8).copy()
    t = time.time(); dst, res, aux = ctx().encode_batch(streams, src, cap + 64, quality=q); tg = time.time() - t
    t = time.time(); want, waux = O.encode_stream(fmt, raw, quality=q); tc = time.time() - t
    got = bytes(dst[:res[0].dst_len])
    print(A.FORMAT_NAMES[fmt], "q%d" % q, "status", res[0].status, "equal", got == want, "%d B" % len(want), "gpu %.2f s cpu %.2f s" % (tg, tc), flush=True)
