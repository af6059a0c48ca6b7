"""-m gpu: encode batches of FEW buffers (tens to hundreds: a directory of files, the chunks of an archive) of the flag-bit formats --
csrc/alz_encode_seg.h: the roles walk alone per buffer, then tokens and flag bytes by prefix sums over SEGMENTS of the buffers, five small kernels
instead of one wavefront per buffer for parse + emit.  The compressed bytes, section offsets and statuses must be IDENTICAL to the oracle's
(LzChainMatchFinder + FlagWriter + CompressHeaderless restated) and to what the same call returns with the path switched off."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import Context
from test_gpu_big_encode import _encode, _mixed

pytestmark = pytest.mark.gpu
FAMILY = [A.FMT_LZSS, A.FMT_LZ10, A.FMT_YAZ0, A.FMT_YAY0, A.FMT_MIO0, A.FMT_CLZ0, A.FMT_BLZ, A.FMT_LZHUDSON, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE,
          A.FMT_LZ11, A.FMT_LZ40,        # (round 6: matches of up to 16 KiB -- on the speculative walk as well, in front of the token / flag emitters)
          A.FMT_LZ4_BLOCK, A.FMT_LZO]    # (round 6: every segment walked speculatively, the true walk strung together behind -- alz_encode_seg_seq.h)
OFF = 0xFFFFFFFF
G = 4096


def _seg(c):
    c.lib.alz_debug_seg_launches.restype = C.c_uint64
    c.lib.alz_debug_seg_launches.argtypes = [C.c_void_p]
    return int(c.lib.alz_debug_seg_launches(c.h))


def _both_ways(c, items, quality, what, **kw):
    """One call on the segmented path (asserted), one with it switched off (asserted): the same bytes, both equal to the oracle's."""
    c.big_stream(OFF)                                   # (a handful of buffers would otherwise go one by one through the whole-GPU path)
    try:
        before = _seg(c)
        a = _encode(c, items, quality, expect_big=False, what=what, **kw)
        assert _seg(c) > before, what
        c.lib.alz_debug_seg_max_streams(c.h, 0)
        try:
            before = _seg(c)
            b = _encode(c, items, quality, expect_big=False, what=what + ", path off", **kw)
            assert _seg(c) == before, what
        finally:
            c.lib.alz_debug_seg_max_streams(c.h, 0xFFFFFFFF)
    finally:
        c.big_stream(24 << 10)
    assert a == b, what
    return a


@pytest.mark.parametrize("fmt", FAMILY)
@pytest.mark.parametrize("quality", [0, 4, 8, 12])
def test_ragged_batch(fmt, quality, test_bmp):
    """Forty buffers of every kind of length -- empty, shorter than the finder's four bytes, one window of 64 positions more or less, the segment
    length (1 024 here) more or less, one buffer long enough for the path to be taken -- of bitmap rows, runs, noise and prose-like bytes."""
    sizes = [0, 1, 3, 4, 5, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 1024 + 63, 2047, 2048, 2049, 3071, 3072, 4095, 4096, 4097, 8191, 8192, 8193,
             10000, 12288, 12288 + 1, 16384 - 3, 20000, 33333, 40000, 50001, 65536, 70000, 9, 300, 5000, 70001]
    if fmt == A.FMT_LZ4_BLOCK:
        sizes = [s for s in sizes if s >= 5]                       # (a block ends in five literals: the reference cannot write a shorter one -- tests/test_gpu_encode.py has that case)
    items = [(fmt, _mixed(s, 1000 + s, test_bmp) if s else b"") for s in sizes]
    with Context(0) as c:
        _both_ways(c, items, quality, "ragged")


@pytest.mark.parametrize("fmt", [A.FMT_YAZ0, A.FMT_LZ10, A.FMT_YAY0, A.FMT_LZHUDSON, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_LZ11, A.FMT_LZ40, A.FMT_LZ4_BLOCK, A.FMT_LZO])
def test_segment_lengths(fmt, test_bmp):
    """The segment length follows from buffers x longest buffer (a launch aims at 8 192 segments): 300 x 64 KiB gives 2 432 positions, 700 x 24 KiB
    2 112; windows of Test.bmp 4 KiB apart."""
    with Context(0) as c:
        for n, size in ((300, 65536), (700, 24000), (64, 262144 + 77)):
            items = [(fmt, test_bmp[(i * 4096) % (len(test_bmp) - size):][:size - (i % 7)]) for i in range(n)]
            _both_ways(c, items, 0, "%d x %d" % (n, size))
        items = [(fmt, test_bmp[(i * 4096) % (len(test_bmp) - 65536):][:65536]) for i in range(256)]
        _both_ways(c, items, 8, "256 x 64 KiB, quality 8")


@pytest.mark.parametrize("fmt", FAMILY)
def test_degenerate_buffers(fmt):
    """Runs (one 273-byte match behind the other, no literal for whole segments; positions beyond kernel B's compare cap), two-byte periods, noise
    (nothing but literals: flag groups of eight literals, the output larger than the input), a buffer that ends inside a match / on a literal,
    literals only in the last segment."""
    rng = np.random.default_rng(5)
    raws = [bytes(100000), b"ab" * 40000, bytes(rng.integers(0, 256, 60000, dtype=np.uint8)), bytes(rng.integers(0, 3, 90000, dtype=np.uint8)),
            bytes(50000) + b"abc", bytes(rng.integers(0, 256, 30000, dtype=np.uint8)) + bytes(30000), b"\xff" * 33000 + bytes(rng.integers(0, 256, 7, dtype=np.uint8)),
            bytes(rng.integers(0, 256, 9, dtype=np.uint8)) * 5000, bytes(8192), bytes(8193), bytes(16384 + 2)]
    with Context(0) as c:
        for q in (0, 8, 15):
            _both_ways(c, [(fmt, r) for r in raws], q, "degenerate")


@pytest.mark.parametrize("fmt", [A.FMT_LZ4_BLOCK, A.FMT_LZO, A.FMT_LZ11, A.FMT_LZ40, A.FMT_SNAPPY_RAW])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_long_runs_across_segments(fmt, seed):
    """Runs of one to six thousand equal bytes (or of a short period) with a little noise between them: matches longer than a segment, longer than kernel B's
    compare cap and than the 2 046 bytes a match entry can hold, starting and ending anywhere relative to the segment boundaries -- the walk that enters a
    segment stands inside, on or just behind what the walk of the segment in front searched exactly."""
    rng = np.random.default_rng(900 + seed)
    raws = []
    for b in range(24):
        parts, total = [], 0
        while total < 30000 + 2000 * b:
            kind = int(rng.integers(0, 4))
            ln = int(rng.integers(1, 6000))
            if kind == 0: part = bytes([int(rng.integers(0, 256))]) * ln
            elif kind == 1: part = (bytes(rng.integers(0, 256, int(rng.integers(2, 9)), dtype=np.uint8)) * ln)[:ln]
            elif kind == 2: part = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
            else: part = b"".join(parts)[-ln:] if parts else b"x"          # (a copy of what lies right in front)
            parts.append(part); total += len(part)
        raws.append(b"".join(parts))
    with Context(0) as c:
        for q in (0, 6, 8, 11):
            _both_ways(c, [(fmt, r) for r in raws], q, "long runs, seed %d, quality %d" % (seed, q))


def test_settings_and_mixed_formats(test_bmp):
    """LZSS geometries, CompatibilityMode, VRAM mode; a call of several formats, some on the path (each with its own segments) and some not."""
    raw = test_bmp[:150000]
    with Context(0) as c:
        for bits in [(10, 6, 2), (12, 4, 2), (8, 4, 2), (12, 4, 3), (15, 4, 3), (16, 8, 3)]:      # (the last: 64 KiB distances do not fit the 16-bit links)
            lz = A.LzProperties.from_bits(*bits)
            _both_ways(c, [(A.FMT_LZSS, raw[i * 999:i * 999 + 20000 + i]) for i in range(40)], 8, "lzss %r" % (bits,), lz=lz)
        _both_ways(c, [(A.FMT_LZSS, raw), (A.FMT_LZSS, b"ab" * 60000)] * 20, 8, "compat", strategy=1)
        _both_ways(c, [(A.FMT_LZ10, raw), (A.FMT_LZ10, bytes(100000))] * 20, 8, "vram", min_distance=2)
        mixed = [([A.FMT_YAZ0, A.FMT_LZ11, A.FMT_LZ4_BLOCK, A.FMT_MIO0, A.FMT_LZO, A.FMT_SNAPPY_RAW][i % 6], test_bmp[i * 3000:i * 3000 + 30000 + 100 * i]) for i in range(60)]
        _both_ways(c, mixed, 8, "mixed formats")
        for q in range(16):
            _both_ways(c, [(A.FMT_YAZ0, raw[:40000]), (A.FMT_YAZ0, raw[100000:140000])] * 17, q, "quality %d" % q)


@pytest.mark.parametrize("fmt", [A.FMT_YAZ0, A.FMT_YAY0, A.FMT_LZ10, A.FMT_LZHUDSON, A.FMT_SNAPPY_RAW, A.FMT_PRS_LE, A.FMT_LZ4_BLOCK, A.FMT_LZO, A.FMT_LZ11])
def test_capacity_and_canary_device_resident(fmt, test_bmp):
    """alz_encode_batch_device with the whole destination compared (0xA5 canary, guard regions): destinations of exactly the compressed size are
    filled and nothing else; one byte less is OUTPUT_CAPACITY with dst_len 0 and no byte outside the buffer's own range."""
    n = 48
    raws = [test_bmp[4096 * i:4096 * i + 30001 + 17 * i] for i in range(n)]
    with Context(0) as c:
        c.big_stream(OFF)
        for q in (0, 8):
            wants = [O.encode_stream(fmt, r, quality=q) for r in raws]
            caps = [len(w[0]) - (1 if i % 3 == 1 else 0) + (777 if i % 3 == 2 else 0) for i, w in enumerate(wants)]
            caps[5] = 100
            src = np.frombuffer(b"".join(raws), dtype=np.uint8)
            st = (A.Stream * n)()
            so, do = 0, G
            for i in range(n):
                st[i] = A.Stream(so, do, len(raws[i]), caps[i], 0, 0, 0, fmt)
                so += len(raws[i]); do += caps[i] + G
            d_src, d_dst = c.malloc(len(src)), c.malloc(do)
            try:
                c.h2d(d_src, src); c.memset(d_dst, 0xA5, do)
                before = _seg(c)
                res, aux = c.encode_batch_device(st, d_src, len(src), d_dst, do, quality=q)
                assert _seg(c) > before
                buf = c.d2h(d_dst, do)
            finally:
                c.free(d_src); c.free(d_dst)
            assert np.all(buf[:G] == 0xA5)
            for i in range(n):
                want, waux = wants[i]
                o, tag = st[i].dst_off, (A.FORMAT_NAMES[fmt], q, i, caps[i] - len(want))
                assert np.all(buf[o + caps[i]:o + caps[i] + G] == 0xA5), tag
                if caps[i] >= len(want):
                    assert (res[i].status, res[i].dst_len, res[i].src_used) == (A.ST_OK, len(want), len(raws[i])), tag
                    assert bytes(buf[o:o + len(want)]) == want, tag
                    assert np.all(buf[o + len(want):o + caps[i]] == 0xA5), tag
                    assert (aux[i].aux0, aux[i].aux1) == (waux.aux0, waux.aux1), tag
                else:
                    assert (res[i].status, res[i].dst_len) == (A.ST_OUTPUT_CAPACITY, 0), tag


@pytest.mark.parametrize("fmt", [A.FMT_LZ4_BLOCK, A.FMT_LZ11])
def test_more_segments_than_the_fixup_keeps_in_lds(fmt, test_bmp):
    """A buffer of 9 MB is more than 1 024 segments of the longest kind (8 128 positions): the fix-up of the speculative walk holds the records of the first 1 024 in LDS and
    reads the others where they are; a short buffer beside it, so that the launch is a batch."""
    big = (test_bmp * 9)[:9 * 1000 * 1000 + 321]
    items = [(fmt, big), (fmt, test_bmp[:300000]), (fmt, bytes(200000) + test_bmp[5000:90000])]
    with Context(0) as c:
        _both_ways(c, items, 0, "9 MB")


def test_lzo_head_capacity_and_canary(test_bmp):
    """LZO's head kernel writes whole streams by itself -- noise (no match at all: one literal run, copied by the wavefront), buffers under 16 bytes, a buffer whose first match
    comes late -- and hands the others over after their first match: destinations of exactly the compressed size, one byte less and far too small, with the whole
    destination compared against a 0xA5 canary."""
    rng = np.random.default_rng(77)
    noise = lambda k: bytes(rng.integers(0, 256, k, dtype=np.uint8))
    raws = [noise(40000), noise(15), noise(16), noise(3), b"", noise(20000) + test_bmp[:30000], test_bmp[:50000], noise(2) + bytes(30000), noise(5) + b"abcabcabc" * 3000,
            noise(70) + noise(70)[:60] * 400, bytes(9000), test_bmp[100000:160000]] * 3
    n = len(raws)
    with Context(0) as c:
        c.big_stream(OFF)
        for q in (0, 8):
            wants = [O.encode_stream(A.FMT_LZO, r, quality=q) for r in raws]
            caps = [max(0, len(w[0]) - (1 if i // 12 == 1 else 0)) if i // 12 < 2 else min(len(w[0]) // 2, 37) for i, w in enumerate(wants)]
            caps[6] = len(wants[6][0]) + 555                                   # (one long buffer that fits, so that the path is taken whatever else fails)
            src = np.frombuffer(b"".join(raws) + bytes(64), dtype=np.uint8)
            st = (A.Stream * n)()
            so, do = 0, G
            for i in range(n):
                st[i] = A.Stream(so, do, len(raws[i]), caps[i], 0, 0, 0, A.FMT_LZO)
                so += len(raws[i]); do += caps[i] + G
            d_src, d_dst = c.malloc(len(src)), c.malloc(do)
            try:
                c.h2d(d_src, src); c.memset(d_dst, 0xA5, do)
                before = _seg(c)
                res, aux = c.encode_batch_device(st, d_src, len(src), d_dst, do, quality=q)
                assert _seg(c) > before
                buf = c.d2h(d_dst, do)
            finally:
                c.free(d_src); c.free(d_dst)
            assert np.all(buf[:G] == 0xA5)
            for i in range(n):
                want = wants[i][0]
                o, tag = st[i].dst_off, (q, i, len(raws[i]), caps[i] - len(want))
                assert np.all(buf[o + caps[i]:o + caps[i] + G] == 0xA5), tag
                if caps[i] >= len(want):
                    assert (res[i].status, res[i].dst_len, res[i].src_used) == (A.ST_OK, len(want), len(raws[i])), tag
                    assert bytes(buf[o:o + len(want)]) == want, tag
                    assert np.all(buf[o + len(want):o + caps[i]] == 0xA5), tag
                else:
                    assert (res[i].status, res[i].dst_len) == (A.ST_OUTPUT_CAPACITY, 0), tag


@pytest.mark.parametrize("fmt", FAMILY)
def test_fuzz_path_on_against_path_off(fmt, test_bmp):
    """Random batches under ALZ_FUZZ_SEED (tools/soak.sh repeats this under other seeds): 2-70 buffers of 0-200 KB of bitmap windows, noise, runs and
    periodic bytes, any quality, some destinations too small -- the segmented path and one wavefront per buffer must return the same bytes, lengths and
    statuses (the latter is pinned against the oracle by tests/test_gpu_encode.py), and decode back to the input."""
    import os
    seed = int(os.environ.get("ALZ_FUZZ_SEED", "1234")) * 41 + fmt
    rng = np.random.default_rng(seed)
    with Context(0) as c:
        c.big_stream(OFF)
        for trial in range(6):
            n = int(rng.integers(2, 70))
            q = int(rng.integers(0, 16))
            sizes = [int(2 ** rng.uniform(0, 17.6)) - 1 for _ in range(n)]
            sizes[int(rng.integers(0, n))] = int(rng.integers(8192, 200000))                 # (long enough for the path to be taken)
            raws = [_mixed(s, seed * 1000 + trial * 100 + i, test_bmp) if s else b"" for i, s in enumerate(sizes)]
            caps = [len(r) + len(r) // 4 + 64 if rng.integers(0, 10) else int(rng.integers(0, max(2, len(r) // 2))) for r in raws]
            streams = (A.Stream * n)()
            so = do = 0
            for i, r in enumerate(raws):
                streams[i] = A.Stream(so, do, len(r), caps[i], 0, 0, 0, fmt)
                so += (len(r) + 15) // 16 * 16; do += (caps[i] + 15) // 16 * 16
            src = np.zeros(so + 64, dtype=np.uint8)
            for i, r in enumerate(raws):
                src[streams[i].src_off:streams[i].src_off + len(r)] = np.frombuffer(r, dtype=np.uint8)
            got = []
            for off in (False, True):
                if off:
                    c.lib.alz_debug_seg_max_streams(c.h, 0)
                try:
                    before = _seg(c)
                    dst, res, aux = c.encode_batch(streams, src, do + 64, quality=q)
                    assert (_seg(c) > before) == (not off), (seed, trial)
                finally:
                    c.lib.alz_debug_seg_max_streams(c.h, 0xFFFFFFFF)
                got.append([(res[i].status, res[i].dst_len, res[i].src_used, aux[i].aux0, aux[i].aux1,
                             bytes(dst[streams[i].dst_off:streams[i].dst_off + res[i].dst_len])) for i in range(n)])
            for i in range(n):
                assert got[0][i] == got[1][i], (seed, trial, i, A.FORMAT_NAMES[fmt], q, len(raws[i]), caps[i], got[0][i][:5], got[1][i][:5])
            i = int(np.argmax([len(r) for r in raws]))
            if got[0][i][0] == A.ST_OK and fmt in (A.FMT_LZSS, A.FMT_LZ10, A.FMT_LZ11, A.FMT_LZ40, A.FMT_YAZ0, A.FMT_YAY0, A.FMT_MIO0, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZ4_BLOCK):
                # (not LZO: the reference's writer drops a match that starts within the first three bytes and then writes two literal runs in a row, which no LZO1X decoder -- its
                # own included -- reads back; the GPU writes what it writes, INTEGRATION.md 4 -- seed 86 of the soak met such an input: both paths and the oracle agree on the bytes)
                sized = fmt not in (A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZ4_BLOCK)
                back, dr = c.decode(fmt, got[0][i][5], decom_len=len(raws[i]) if sized else 0, cap=len(raws[i]), aux0=got[0][i][3], aux1=got[0][i][4])
                assert dr.status == 0 and back == raws[i], (seed, trial, i)


@pytest.mark.parametrize("fmt", [A.FMT_YAZ0, A.FMT_PRS_BE, A.FMT_SNAPPY_RAW, A.FMT_MIO0, A.FMT_LZ11, A.FMT_LZ40, A.FMT_LZ4_BLOCK, A.FMT_LZO])
def test_a_few_large_buffers(fmt, test_bmp):
    """Four buffers of 1-5 MB (the whole-GPU path off): more than a thousand segments per buffer, kernel A over segments of its own length, exit tables
    chained across hundreds of boundaries."""
    big = (test_bmp * 5)[:5 * 1000 * 1000 + 123]
    items = [(fmt, big), (fmt, _mixed(3000000, 77, test_bmp)), (fmt, bytes(1500000) + test_bmp[:200000]), (fmt, test_bmp[:1000001])]
    with Context(0) as c:
        _both_ways(c, items, 0, "large")
        _both_ways(c, items[1:], 8, "large, quality 8")
