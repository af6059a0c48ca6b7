#!/usr/bin/env python3
"""usage (GPU box): [ALZ_SOAK_Q=0] python tools/soak_encode.py [first_seed] [count]      (ALZ_SOAK_Q: one quality for every batch instead of a random one)
Differential soak of the ENCODER: batches of raw buffers of many shapes (noise, runs, periods of 1..5000, few symbols, bitmap slices,
mixtures; 1 B .. 3 MB) at qualities 0..15 in several formats, GPU output against the CPU restatement byte for byte.  Aimed at kernel A's
queue / pass / drain logic (enc_prev_cu_kernel).  Not part of the test suite (minutes)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from auroralib.compression_amd import _abi as A  # noqa: E402
from gpu_common import ctx  # noqa: E402


def shapes(rng, bmp, size):
    kind = int(rng.integers(0, 8))
    if kind == 0:
        return bytes(rng.integers(0, 256, size, dtype=np.uint8))
    if kind == 1:
        return bytes([int(rng.integers(0, 256))]) * size
    if kind == 2:
        p = int(rng.integers(1, 5000)); unit = bytes(rng.integers(0, 256, p, dtype=np.uint8))
        return (unit * (size // p + 1))[:size]
    if kind == 3:
        return bytes(rng.integers(0, int(rng.integers(2, 6)), size, dtype=np.uint8))
    if kind == 4:
        off = int(rng.integers(0, max(1, len(bmp) - size - 1)))
        return bytes(bmp[off:off + size])
    if kind == 5:                                           # long runs between noise
        out = bytearray()
        while len(out) < size:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 40000)) + bytes(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8))
        return bytes(out[:size])
    if kind == 6:                                           # a few 4-grams all over the place (one hash class crowded)
        grams = [bytes(rng.integers(0, 256, 4, dtype=np.uint8)) for _ in range(int(rng.integers(1, 5)))]
        out = bytearray()
        while len(out) < size:
            out += grams[int(rng.integers(0, len(grams)))] + bytes(rng.integers(0, 256, int(rng.integers(0, 6)), dtype=np.uint8))
        return bytes(out[:size])
    a = shapes(rng, bmp, size // 2); b = shapes(rng, bmp, size - size // 2)
    return a + b


def main():
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 4242
    cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    lz = open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read()
    bmp, st = O.container_decompress(A.C_LZSS, lz, lz=A.LzProperties.from_bits(10, 6, 2))
    bmp = np.frombuffer(bmp, dtype=np.uint8)
    bad = 0
    for k in range(cnt):
        seed = s0 + 7919 * k
        rng = np.random.default_rng(seed)
        for fmt in (A.FMT_LZSS, A.FMT_LZ10, A.FMT_YAZ0, A.FMT_LZ4_BLOCK, A.FMT_LZ11, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZO, A.FMT_LZHUDSON, A.FMT_REFPACK):
            q = int(rng.integers(0, 16))
            if os.environ.get("ALZ_SOAK_Q"):
                q = int(os.environ["ALZ_SOAK_Q"])
            sizes = [int(rng.choice([1, 3, 4, 5, 63, 64, 65, 2047, 2048, 2049, 3071, 3072, 3073, 24575, 24576, 24577, 32768, 65536, 65537])) for _ in range(6)]
            sizes += [int(rng.integers(1, 400000)) for _ in range(8)] + [int(rng.integers(400000, 3000000))]
            if q >= 10:                                     # (chains of up to 1 024 candidates on the CPU side)
                sizes = [min(s, 150000) for s in sizes]
            raws = [shapes(rng, bmp, s) for s in sizes]
            n = len(raws)
            streams = (A.Stream * n)()
            so = do = 0
            chunks = []
            for i, r in enumerate(raws):
                cap = len(r) + len(r) // 4 + 64
                streams[i] = A.Stream(so, do, len(r), cap, 0, 0, 0, fmt)
                pad = (-len(r)) % 16
                chunks.append(bytes(r) + bytes(pad))
                so += len(r) + pad
                do += (cap + 15) // 16 * 16
            src = np.frombuffer(b"".join(chunks) + bytes(64), dtype=np.uint8).copy()
            dst, res, aux = ctx().encode_batch(streams, src, do + 64, quality=q)
            for i, r in enumerate(raws):
                try:
                    want, waux = O.encode_stream(fmt, r, quality=q)
                except ValueError:
                    continue
                got = bytes(dst[streams[i].dst_off:streams[i].dst_off + res[i].dst_len])
                if res[i].status != A.ST_OK or got != want:
                    bad += 1
                    print("MISMATCH seed %d %s q%d stream %d (%d B): status %d, gpu %d B, oracle %d B" % (seed, A.FORMAT_NAMES[fmt], q, i, len(r), res[i].status, len(got), len(want)), flush=True)
            print("seed %d %s q%d: %d streams, %.1f MB ok" % (seed, A.FORMAT_NAMES[fmt], q, n, sum(sizes) / 1e6), flush=True)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
