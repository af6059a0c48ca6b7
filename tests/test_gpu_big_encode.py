"""-m gpu: ONE big stream on the whole GPU, encoder side (csrc/alz_encode_big.h) against the oracle's restatement of LzChainMatchFinder +
FlagWriter + CompressHeaderless: a call of at most 32 streams, each of at least 8 KiB of one of the formats of the path, goes through segmented
prev(), list ranking for the parse and prefix-sum emission instead of one workgroup + one wavefront per stream.  The compressed bytes, the
section offsets and the statuses must be IDENTICAL to the oracle's -- and so to the batch pipeline's, which the same inputs go through with
the path switched off."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import Context
from cases import prose_like

pytestmark = pytest.mark.gpu
FMTS = [A.FMT_LZSS, A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_YAY0, A.FMT_MIO0, A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZO, A.FMT_LZ40, A.FMT_CLZ0, A.FMT_BLZ, A.FMT_LZHUDSON]
NORTH = [A.FMT_LZSS, A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_YAY0, A.FMT_MIO0, A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZO]
OFF = 0xFFFFFFFF
G = 4096


def _cap(n):
    return n + n // 4 + 64


def _encode(c, items, quality, caps=None, expect_big=None, what="", **kw):
    """items: (format, raw) -- ONE alz_encode_batch call; every stream against the oracle.  Returns the compressed streams."""
    n = len(items)
    streams = (A.Stream * n)()
    so = do = 0
    chunks = []
    for i, (fmt, r) in enumerate(items):
        cap = _cap(len(r)) if caps is None else caps[i]
        streams[i] = A.Stream(so, do, len(r), cap, 0, 0, 0, fmt)
        pad = (-len(r)) % 16
        chunks.append(bytes(r) + bytes(pad))
        so += len(r) + pad
        do += (cap + 15) // 16 * 16
    src = np.frombuffer(b"".join(chunks) + bytes(64), dtype=np.uint8).copy()
    before = c.big_stream()
    dst, res, aux = c.encode_batch(streams, src, do + 64, quality=quality, **kw)
    took = c.big_stream() - before
    if expect_big is not None:
        assert (took > 0) == expect_big, (what, took)
    out = []
    for i, (fmt, r) in enumerate(items):
        want, waux = O.encode_stream(fmt, r, quality=quality, **kw)
        tag = (what, A.FORMAT_NAMES[fmt], quality, i, len(r))
        if len(want) > streams[i].dst_cap:
            assert (res[i].status, res[i].dst_len) == (A.ST_OUTPUT_CAPACITY, 0), tag
            out.append(None)
            continue
        assert res[i].status == A.ST_OK, tag + (res[i].status,)
        got = bytes(dst[streams[i].dst_off:streams[i].dst_off + res[i].dst_len])
        if got != want:
            k = next((j for j in range(min(len(got), len(want))) if got[j] != want[j]), min(len(got), len(want)))
            raise AssertionError("%s: gpu %d B vs oracle %d B, first difference at %d" % (tag, len(got), len(want), k))
        assert res[i].src_used == len(r), tag
        assert (aux[i].aux0, aux[i].aux1) == (waux.aux0, waux.aux1), tag
        out.append(got)
    return out


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("quality", [0, 4, 8, 12, 15])
def test_whole_test_bmp_as_one_stream(fmt, quality, test_bmp):
    """The reference's benchmark shape: ONE ~1 MiB stream (Benchmarks/Benchmarks/TestAllAlgorithms.cs:41-42)."""
    with Context(0) as c:
        _encode(c, [(fmt, test_bmp)], quality, expect_big=True, what="bmp")


def _mixed(size, seed, test_bmp):
    """Text-like, binary and run-heavy pieces: bmp windows, low-entropy noise, zero runs, a repeated phrase."""
    rng = np.random.default_rng(seed)
    parts, total = [], 0
    while total < size:
        k = int(rng.integers(0, 5))
        ln = int(rng.integers(200, 20000))
        if k == 0:
            o = int(rng.integers(0, len(test_bmp) - ln)); p = test_bmp[o:o + ln]
        elif k == 1:
            p = bytes(rng.integers(0, 4, ln, dtype=np.uint8))
        elif k == 2:
            p = bytes(ln)
        elif k == 3:
            p = bytes(rng.integers(0, 256, ln // 8 + 1, dtype=np.uint8)) * 8
        else:
            p = bytes(rng.integers(0, 256, ln, dtype=np.uint8))
        parts.append(p); total += len(p)
    return b"".join(parts)[:size]


@pytest.mark.parametrize("fmt", NORTH)
def test_sizes_and_the_threshold(fmt, test_bmp):
    """Both sides of the 8 KiB threshold, the segment (16 384 positions) and tile (1 024 positions) boundaries, 4 MiB; the path is taken from
    the threshold on, never below it, never on a context with the path switched off -- and the bytes are the same either way."""
    with Context(0) as c:
        for size in (8191, 8192, 8193, 9215, 9216, 12345, 16383, 16384, 16385, 16384 + 4, 20000, 65536 + 3, 98304 + 16384 - 1, 7 * 16384, 7 * 16384 + 1,
                     7 * 16384 + 4, 131072 + 5, 262144, (1 << 22) + 3):
            raw = _mixed(size, size, test_bmp)
            q = 8 if size < (1 << 22) else 0
            a = _encode(c, [(fmt, raw)], q, expect_big=size >= 8192, what="size %d" % size)
            c.big_stream(OFF)
            b = _encode(c, [(fmt, raw)], q, expect_big=False, what="size %d, path off" % size)
            c.big_stream(24 << 10)
            assert a == b


@pytest.mark.parametrize("fmt", FMTS)
def test_degenerate_inputs(fmt):
    """Runs (every position of an LZ11 / LZ40 run has a match beyond kernel B's compare cap: the exact second search, one wavefront per
    position), two-byte periods, incompressible noise (the output is larger than the input), matches of 2 046 bytes and more at the cursor
    and at the lazy neighbour (tests/test_gpu_encode.py), a stream that ends inside a match / on a literal."""
    rng = np.random.default_rng(11)
    R = bytes(rng.integers(0, 256, 2600, dtype=np.uint8))
    J1, J2 = bytes(rng.integers(0, 256, 40, dtype=np.uint8)), bytes(rng.integers(0, 256, 300, dtype=np.uint8))
    lazy = b"q" + R[:2] + b"#" + J1 + R + J2 + b"q" + R + J1
    direct = J1 + R + J2 + R + R[:2100] + J1
    raws = [bytes(1 << 20), b"ab" * 100000, bytes(rng.integers(0, 256, 150000, dtype=np.uint8)), bytes(rng.integers(0, 3, 200000, dtype=np.uint8)),
            lazy + direct + bytes(2046) + b"z" + bytes(2047) + b"y" + bytes(2045) + b"x" + bytes(100000),
            (lazy + direct) * 12, bytes(120000) + b"abc", bytes(rng.integers(0, 256, 100000, dtype=np.uint8)) + bytes(30000),
            b"\xff" * 99000 + bytes(rng.integers(0, 256, 7, dtype=np.uint8))]
    with Context(0) as c:
        for q in (0, 8, 15):
            for i, r in enumerate(raws):
                _encode(c, [(fmt, r)], q, expect_big=True, what="degenerate %d" % i)


def test_settings_reach_the_path(test_bmp):
    """LZSS geometries, CompatibilityMode (no self-overlapping matches), VRAM mode (minimum distance 2), every quality."""
    raw = test_bmp[:300000]
    with Context(0) as c:
        for bits in [(10, 6, 2), (12, 4, 2), (8, 4, 2), (12, 4, 3), (16, 8, 3)]:
            lz = A.LzProperties.from_bits(*bits)
            if lz.max_distance > 0x8000:
                continue
            _encode(c, [(A.FMT_LZSS, raw)], 8, expect_big=True, what="lzss %r" % (bits,), lz=lz)
        _encode(c, [(A.FMT_LZSS, raw), (A.FMT_LZSS, b"ab" * 60000)], 8, expect_big=True, what="compat", strategy=1)
        for fmt in (A.FMT_LZ10, A.FMT_LZ11):
            for q in (8, 15):
                _encode(c, [(fmt, raw), (fmt, bytes(100000))], q, expect_big=True, what="vram", min_distance=2)
        for q in range(16):
            _encode(c, [(A.FMT_YAZ0, raw[:120000]), (A.FMT_LZ10, raw[100000:220000])], q, expect_big=True, what="quality")


def test_a_handful_of_streams_and_mixed_calls(test_bmp):
    """A handful of eligible streams take the path one after the other while that beats side by side (at most 32); a small one or a format
    outside the path sends the whole call through the batch pipeline -- with the same bytes."""
    with Context(0) as c:
        eight = [(NORTH[i % len(NORTH)], test_bmp[i * 50000:i * 50000 + 100000 + 1000 * i]) for i in range(8)]
        _encode(c, eight, 8, expect_big=True, what="eight")
        _encode(c, eight + [(A.FMT_YAZ0, test_bmp[:100000])], 8, expect_big=True, what="nine")
        # thirty of 100 KB each: side by side is the faster arrangement (the cost rule of encode_core); four of 1 MB among them change that
        thirty = [(NORTH[i % len(NORTH)], test_bmp[i * 1000:i * 1000 + 100000]) for i in range(30)]
        _encode(c, thirty, 4, expect_big=False, what="thirty small")                      # (30 x ~0.11 ms against ~2.1 ms side by side)
        _encode(c, thirty[:10] + [(A.FMT_LZ10, test_bmp[:1000000 + i]) for i in range(4)], 4, expect_big=True, what="ten small, four of 1 MB")
        _encode(c, [(A.FMT_LZ10, test_bmp[i:300000 + i]) for i in range(33)], 0, expect_big=False, what="thirty-three")
        _encode(c, eight[:3] + [(A.FMT_YAZ0, test_bmp[:5000])], 8, expect_big=False, what="one small")
        _encode(c, eight[:3] + [(A.FMT_REFPACK, test_bmp[:200000])], 8, expect_big=False, what="one RefPack (three property sets: not on the path)")


@pytest.mark.parametrize("fmt", NORTH)
def test_capacity_and_canary_device_resident(fmt, test_bmp):
    """alz_encode_batch_device with the source ending exactly at the end of its buffer (no slack behind it: the path copies the stream into
    its scratch) and the whole destination compared -- 0xA5 canary, guard regions: a destination of exactly the compressed size is filled
    and nothing else; one byte less is OUTPUT_CAPACITY with dst_len 0 and no byte outside the stream's range."""
    raw = test_bmp[4096:4096 + 300001]
    with Context(0) as c:
        for q in (0, 8):
            want, waux = O.encode_stream(fmt, raw, quality=q)
            for cap in (len(want), len(want) - 1, len(want) + 777, 100):
                d_src, d_dst = c.malloc(len(raw)), c.malloc(G + cap + G)
                try:
                    c.h2d(d_src, np.frombuffer(raw, dtype=np.uint8)); c.memset(d_dst, 0xA5, G + cap + G)
                    st = (A.Stream * 1)(A.Stream(0, G, len(raw), cap, 0, 0, 0, fmt))
                    before = c.big_stream()
                    res, aux = c.encode_batch_device(st, d_src, len(raw), d_dst, G + cap + G, quality=q)
                    assert c.big_stream() - before == 1
                    buf = c.d2h(d_dst, G + cap + G)
                finally:
                    c.free(d_src); c.free(d_dst)
                tag = (A.FORMAT_NAMES[fmt], q, cap - len(want))
                assert np.all(buf[:G] == 0xA5) and np.all(buf[G + cap:] == 0xA5), tag
                if cap >= len(want):
                    assert (res[0].status, res[0].dst_len, res[0].src_used) == (A.ST_OK, len(want), len(raw)), tag
                    assert bytes(buf[G:G + len(want)]) == want, tag
                    assert np.all(buf[G + len(want):G + cap] == 0xA5), tag
                    assert (aux[0].aux0, aux[0].aux1) == (waux.aux0, waux.aux1), tag
                else:
                    assert (res[0].status, res[0].dst_len) == (A.ST_OUTPUT_CAPACITY, 0), tag


def test_compress_through_the_format_classes(test_bmp):
    """ICompressionEncoder.Compress of the format-class mirror on the benchmark input == the oracle's container bytes."""
    from auroralib.compression_amd import formats as F
    for cls, cont in [(F.LZSS, A.C_LZSS), (F.LZ10, A.C_LZ10), (F.LZ11, A.C_LZ11), (F.Yaz0, A.C_YAZ0), (F.Yay0, A.C_YAY0), (F.MIO0, A.C_MIO0)]:
        for s in (F.CompressionSettings.Fastest, F.CompressionSettings.Balanced):
            f = cls()
            comp = f.Compress(test_bmp, s)
            assert comp == O.container_compress(cont, test_bmp, quality=s.Quality), (cls.__name__, s.Quality)


def test_the_benchmark_input_against_committed_vectors(test_bmp):
    """The reference's own benchmark input -- the first 1 000 KiB of Test.bmp at CompressionLevel 0 and 15 -- for the eleven north-star
    bodies against tests/golden/oracle_vectors.json["benchmark"]: no oracle at run time (lengths, XXH64 digests, section offsets), and every
    stream must have taken the whole-GPU path."""
    import json
    import os
    import xxhash
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))["benchmark"]
    assert len(vec) == 22
    with Context(0) as c:
        for key, (length, digest, a0, a1) in sorted(vec.items()):
            name, size, q = key.split(":")
            n = int(size)
            raw = np.frombuffer(test_bmp[:n] + bytes(64), dtype=np.uint8)
            st = (A.Stream * 1)(A.Stream(0, 0, n, _cap(n), 0, 0, 0, A.FORMAT_NAMES.index(name)))
            before = c.big_stream()
            dst, res, aux = c.encode_batch(st, raw, _cap(n) + 64, quality=int(q[1:]))
            assert c.big_stream() - before == 1, key
            assert (res[0].status, res[0].dst_len, res[0].src_used) == (A.ST_OK, length, n), key
            assert xxhash.xxh64(bytes(dst[:length])).hexdigest() == digest and (aux[0].aux0, aux[0].aux1) == (a0, a1), key


@pytest.mark.parametrize("fmt", NORTH)
def test_big_encode_fuzz(fmt, test_bmp):
    """Random inputs under ALZ_FUZZ_SEED (tools/soak.sh repeats this under other seeds): mixtures of bitmap windows, noise of two to 256
    symbols, runs, periodic pieces, at random lengths, qualities and destination sizes -- the bytes, the lengths and the statuses are the
    oracle's."""
    import os
    import random
    seed = int(os.environ.get("ALZ_FUZZ_SEED", "1234")) * 37 + fmt
    rng = random.Random(seed)
    with Context(0) as c:
        for k in range(6):
            size = rng.choice([8192 + rng.randrange(0, 3000), rng.randrange(8192, 98304), 98304 + rng.randrange(1, 70000), rng.randrange(100000, 400000), rng.randrange(100000, 1500000)])
            raw = _mixed(size, seed * 8 + k, test_bmp)
            if k % 3 == 2:                                         # long repeats: matches beyond kernel B's compare cap
                piece = raw[:rng.randrange(2100, 9000)]
                raw = (piece * (size // len(piece) + 1))[:size // 2] + raw[:size - size // 2]
            q = rng.choice([0, 1, 2, 3, 5, 8, 9, 10, 12, 15]) if size < 500000 else rng.choice([0, 3, 8])
            want_len = len(O.encode_stream(fmt, raw, quality=q)[0])
            cap = rng.choice([_cap(len(raw)), want_len, want_len + rng.randrange(1, 100), max(1, want_len - rng.randrange(1, 100))])
            _encode(c, [(fmt, raw)], q, caps=[cap], expect_big=True, what="fuzz %d (seed %d)" % (k, seed))


@pytest.mark.parametrize("fmt", NORTH)
def test_prose_like_input_both_paths(fmt):
    """Text instead of a bitmap, through the whole-GPU path (a call of three buffers) and through the batch pipeline (forty): the oracle's bytes."""
    with Context(0) as c:
        for q in (0, 5, 8, 12):
            three = [(fmt, prose_like(150000 + 7777 * i, 100 * q + i)) for i in range(3)]
            c.lib.alz_debug_seg_max_streams(c.h, 0)              # (three buffers of a format the segmented batch path takes would go side by side: csrc/alz_encode_seg.h)
            try:
                _encode(c, three, q, expect_big=True, what="text, three")
            finally:
                c.lib.alz_debug_seg_max_streams(c.h, 0xFFFFFFFF)
            forty = [(fmt, prose_like(9000 + 613 * i, 900 + 40 * q + i)) for i in range(40)]
            _encode(c, forty, q, expect_big=False, what="text, forty")
