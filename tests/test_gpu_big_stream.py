"""-m gpu: ONE big stream on the whole GPU (csrc/alz_big.hip, alz_ctx_big_stream) against the oracle: Yay0 / MIO0 (three sections: the
cursors of a token are prefix sums) and LZSS / LZ10 / LZ11 / Yaz0 (one interleaved stream: the group starts come from list ranking over
the input bytes).

A batch of one stream of >= 24 KiB goes through that path and pointer jumping over its output bytes instead of through one or two
wavefronts.  Valid streams must come out bit-exact with status / dst_len / src_used of the oracle; every malformed
stream (truncated, overshooting, undershooting, short destination, cursors outside the input) must come out exactly as the exact kernel
behind the path decodes it -- the path may only ever DECLINE such a stream.  The device-resident cases compare the whole destination buffer
(0xA5 canary, guard regions), as tests/test_gpu_canary.py does."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from auroralib.compression_amd.batch import Context, Plan

pytestmark = pytest.mark.gpu
FMTS = [A.FMT_YAY0, A.FMT_MIO0, A.FMT_LZSS, A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZO]
ELEM = (A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZO)       # no size in the descriptor: the room in the destination bounds the launch
THREE = (A.FMT_YAY0, A.FMT_MIO0)
OFF = 0xFFFFFFFF


def _one(c, fmt, comp, decom_len, aux0, aux1, cap=None, expect_big=None, what="", lz=None):
    """The stream through alz_decode (host buffers) and through a device-resident plan with a canary; both against the oracle."""
    cap = decom_len if cap is None else cap
    if fmt in ELEM:
        decom_len, aux0, aux1 = 0, 0, 0
    want, wr = O.decode_stream(fmt, comp, decom_len=decom_len, cap=cap, aux0=aux0, aux1=aux1, lz=lz)
    before = c.big_stream()
    got, r = c.decode(fmt, comp, decom_len=decom_len, cap=cap, aux0=aux0, aux1=aux1, lz=lz)
    took = c.big_stream() - before
    assert (r.status, r.dst_len) == (wr.status, wr.dst_len), (what, r.status, r.dst_len, wr.status, wr.dst_len)
    if wr.status != A.ST_OUTPUT_CAPACITY:
        assert r.src_used == wr.src_used, (what, r.src_used, wr.src_used)
    assert got == want, what
    if expect_big is not None:
        assert (took > 0) == expect_big, (what, took)
    # device-resident, whole buffer
    G = 4096
    src = np.frombuffer(bytes(comp) + bytes(64), dtype=np.uint8)
    st = (A.Stream * 1)(A.Stream(0, 0, len(comp), cap, decom_len, aux0, aux1, fmt))
    d_src, d_dst = c.malloc(src.nbytes), c.malloc(G + cap + G)
    try:
        c.h2d(d_src, src); c.memset(d_dst, 0xA5, G + cap + G)
        p = Plan(c, st, lz)
        p.execute(d_src, C.c_void_p(d_dst.value + G))
        pr = p.results()[0]
        buf = c.d2h(d_dst, G + cap + G)
        p.close()
    finally:
        c.free(d_src); c.free(d_dst)
    assert (pr.status, pr.dst_len) == (wr.status, wr.dst_len), what
    exp = np.full(G + cap + G, 0xA5, dtype=np.uint8)
    exp[G:G + len(want)] = np.frombuffer(want, dtype=np.uint8)
    assert np.array_equal(buf, exp), (what, int((buf != exp).sum()), int(np.flatnonzero(buf != exp)[0]) - G)


@pytest.mark.parametrize("fmt", FMTS)
def test_whole_test_bmp_as_one_stream(fmt, test_bmp):
    """The reference's benchmark shape: one ~1 MiB stream (Benchmarks/Benchmarks/TestAllAlgorithms.cs:41-42), Q0 and the default Q8."""
    with Context(0) as c:
        for q in (0, 8):
            comp, aux = O.encode_stream(fmt, test_bmp, quality=q)
            _one(c, fmt, comp, len(test_bmp), aux.aux0, aux.aux1, expect_big=True, what="bmp q%d" % q)


@pytest.mark.parametrize("fmt", FMTS)
def test_synthetic_sizes_and_the_threshold(fmt):
    """Sizes on both sides of the 24 KiB threshold, tile boundaries (1 024 tokens), 4 MiB; the path must be taken from the threshold on
    and never below it, nor on contexts with forced kernels."""
    with Context(0) as c:
        for size in (24575, 24576, 30000, 65536 + 1, 98304, 100000, 131072 + 5, 262144, 1 << 20, (1 << 22) + 3):
            b = synth.make_batch(fmt, 1, size, synth.seed_for(30 + fmt, size))
            s = b.streams[0]
            comp = bytes(b.src[s.src_off:s.src_off + s.src_len])
            # (the formats whose descriptor states no size are taken by their INPUT: at least 8 KiB of it)
            _one(c, fmt, comp, size, s.aux0, s.aux1, expect_big=size >= 24576 and (fmt not in ELEM or len(comp) >= 8192), what="synthetic %d" % size)
            if fmt in ELEM:                                    # a destination with room to spare (the usual case for a body without a size)
                _one(c, fmt, comp, size, 0, 0, cap=size + 70000, expect_big=len(comp) >= 8192, what="synthetic %d, roomy" % size)
        b = synth.make_batch(fmt, 1, 300000, 7)
        s = b.streams[0]
        comp = bytes(b.src[s.src_off:s.src_off + s.src_len])
        c.big_stream(OFF)
        _one(c, fmt, comp, 300000, s.aux0, s.aux1, expect_big=False, what="switched off")
        c.big_stream(64 << 10)
        c.set_kernel_variant(1)
        _one(c, fmt, comp, 300000, s.aux0, s.aux1, expect_big=False, what="forced variant")
        c.set_kernel_variant(0)
        c.set_exact_kernels(1)
        _one(c, fmt, comp, 300000, s.aux0, s.aux1, expect_big=False, what="exact kernels")


@pytest.mark.parametrize("fmt", FMTS)
def test_degenerate_data(fmt):
    """Runs (every byte a copy at distance 1: the deepest pointer chains), short periods, all literals, zeros."""
    rng = np.random.default_rng(3)
    raws = [bytes(400000), b"\xAB" * 300001, b"abc" * 100000, bytes(rng.integers(0, 256, 150000, dtype=np.uint8)),
            bytes(rng.integers(0, 256, 5000, dtype=np.uint8)) * 60, b"".join(bytes([i & 255]) * (1 + i % 300) for i in range(2000))]
    with Context(0) as c:
        for k, raw in enumerate(raws):
            for q in (0, 8):
                comp, aux = O.encode_stream(fmt, raw, quality=q)
                # (LZ4 / Snappy: an input below 8 KiB is not worth the launches -- runs compress to a few hundred bytes)
                _one(c, fmt, comp, len(raw), aux.aux0, aux.aux1, expect_big=(len(comp) >= 8192 if fmt in ELEM else True), what="degenerate %d q%d" % (k, q))


@pytest.mark.parametrize("fmt", FMTS)
def test_malformed_streams_fall_through_to_the_exact_kernel(fmt, test_bmp):
    """The path declines; what comes out is the production kernel's answer = the oracle's."""
    import random
    rng = random.Random(5 + fmt)
    raw = test_bmp[:300000]
    comp, aux = O.encode_stream(fmt, raw, quality=8)
    n = len(raw)
    with Context(0) as c:
        cases = [(comp[:len(comp) // 2], n, n, aux.aux0, aux.aux1, "truncated input"),
                 (comp[:len(comp) - 1], n, n, aux.aux0, aux.aux1, "one byte short"),
                 (comp, n - 1000, n, aux.aux0, aux.aux1, "declared size too small (overshoot)"),
                 (comp, n - 1, n - 1, aux.aux0, aux.aux1, "declared size one too small"),
                 (comp, n + 5000, n + 5000, aux.aux0, aux.aux1, "declared size too large"),
                 (comp, n, n - 4097, aux.aux0, aux.aux1, "short destination"),
                 (comp + bytes(5000), n, n, aux.aux0, aux.aux1, "trailing zeros behind the stream"),
                 (comp + comp[:7000], n, n, aux.aux0, aux.aux1, "trailing data behind the stream")]
        if fmt in THREE:
            cases += [(comp, n, n, aux.aux0 + 2, aux.aux1, "token cursor off by one token"),
                      (comp, n, n, aux.aux0, aux.aux1 + 1, "literal cursor off by one"),
                      (comp, n, n, len(comp) + 7, aux.aux1, "token section outside the input"),
                      (comp, n, n, aux.aux0, len(comp), "literal section at the end of the input"),
                      (comp, n, n, 0, 0, "all three sections on top of each other")]
        else:
            cases += [(comp[1:], n, n, 0, 0, "first byte missing (every group misparsed)"),
                      (comp[:1000] + comp[1001:], n, n, 0, 0, "one byte dropped in the middle")]
        for _ in range(12):
            b = bytearray(comp)
            for _ in range(rng.randrange(1, 5)):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            cases.append((bytes(b), n, n, aux.aux0, aux.aux1, "bit flips"))
        cases.append((bytes(rng.randrange(256) for _ in range(120000)), n, n, 20000 if fmt in THREE else 0, 60000 if fmt in THREE else 0, "noise"))
        for src, decl, cap, a0, a1, what in cases:
            _one(c, fmt, src, decl, a0, a1, cap=cap, what=what)


def test_lzss_geometries_on_the_path(test_bmp):
    """LZSS with other windows than the default 4 KiB: the path has no LDS ring, every window up to 64 KiB is the same code
    (LzProperties.cs:57-66; the KAT geometry (10,6,2) included)."""
    with Context(0) as c:
        for bits in [(10, 6, 2), (12, 4, 2), (14, 4, 2), (16, 8, 2)]:
            lz = A.LzProperties.from_bits(*bits)
            for raw, q in ((test_bmp[:500000], 8), (bytes(200000), 0)):
                comp, aux = O.encode_stream(A.FMT_LZSS, raw, quality=q, lz=lz)
                _one(c, A.FMT_LZSS, comp, len(raw), 0, 0, expect_big=True, what="lzss%r q%d" % (bits, q), lz=lz)


def test_the_format_classes_reach_it(test_bmp):
    """Decompress of the format classes of the host mirror (= ICompressionDecoder.Decompress on one file) takes the path."""
    from auroralib.compression_amd import formats as F
    for cls, cont in ((F.Yay0, A.C_YAY0), (F.MIO0, A.C_MIO0), (F.Yaz0, A.C_YAZ0), (F.LZ10, A.C_LZ10), (F.LZ11, A.C_LZ11), (F.LZSS, A.C_LZSS)):
        comp = O.container_compress(cont, test_bmp, quality=8)
        before = F._context().big_stream()
        assert cls().Decompress(comp) == test_bmp
        assert F._context().big_stream() > before, cls.__name__


def test_a_handful_of_big_streams_in_one_batch(test_bmp):
    """A small batch of streams that are ALL big takes the path stream by stream when that is faster than running them side by side on
    wavefronts of their own (mixed formats, one of them damaged: that one falls through to its exact kernel, the others do not -- and
    alz_ctx_big_stream counts the streams the path ACCEPTED, on the device: five of the six); more
    streams than pay off, or one small stream among them, run on the production kernels."""
    from gpu_common import _check, pack_streams
    fmts = [A.FMT_YAZ0, A.FMT_LZ10, A.FMT_MIO0, A.FMT_LZ11, A.FMT_YAY0, A.FMT_LZSS]
    items = []
    for i, f in enumerate(fmts):
        raw = test_bmp[i * 50000:i * 50000 + 200000 + 1000 * i]
        comp, aux = O.encode_stream(f, raw, quality=[0, 8][i & 1])
        if i == 3:
            comp = comp[:len(comp) // 2]                       # truncated
        items.append(dict(fmt=f, src=comp, decom_len=len(raw), aux0=aux.aux0, aux1=aux.aux1))
    with Context(0) as c:
        for extra, expect in (([], len(fmts) - 1), ([dict(fmt=A.FMT_YAZ0, src=O.encode_stream(A.FMT_YAZ0, test_bmp[:1000], quality=4)[0], decom_len=1000)], 0),
                              ([items[0]] * 30, 0)):
            its = items + extra
            streams, src, dst_bytes = pack_streams(its)
            o_dst, o_res = O.decode_batch(streams, src, dst_bytes, nthreads=4)
            before = c.big_stream()
            g_dst, g_res = c.decode_batch(streams, src, dst_bytes)
            assert c.big_stream() - before == expect, (len(its), c.big_stream() - before)
            _check(streams, g_dst, g_res, o_dst, o_res, "handful of %d" % len(its))


def test_lz4_long_extensions_and_a_wall_of_ff():
    """LZ4 length extensions: matches and runs of hundreds of KiB (chains of 0xFF bytes) stay on the path up to its bound and fall through
    beyond it; an input that is nothing but 0xFF (every byte position speculates an extension chain up to the bound) finishes promptly."""
    rng = np.random.default_rng(9)
    noise = bytes(rng.integers(0, 256, 40000, dtype=np.uint8))
    with Context(0) as c:
        for raw, big in ((noise + bytes(300000) + noise, True), (noise + bytes(900000) + noise, False), (noise[:9000] + bytes(rng.integers(0, 256, 300000, dtype=np.uint8)), True)):
            comp, _ = O.encode_stream(A.FMT_LZ4_BLOCK, raw, quality=8)
            _one(c, A.FMT_LZ4_BLOCK, comp, len(raw), 0, 0, what="lz4 long %d" % len(raw))
        wall = b"\xFF" * 200000
        _one(c, A.FMT_LZ4_BLOCK, wall, 0, 0, 0, cap=1 << 20, what="wall of 0xFF")
        _one(c, A.FMT_SNAPPY_RAW, b"\xFF" * 100000, 0, 0, 0, cap=1 << 20, what="snappy wall of 0xFF")


@pytest.mark.parametrize("fmt", FMTS)
def test_big_stream_fuzz(fmt, test_bmp):
    """Mutated big streams (bit flips, spliced noise, cuts, wrong sizes) under ALZ_FUZZ_SEED: whatever the whole-GPU path makes of them --
    finishing them or handing them to the exact kernel -- status, lengths, src_used, every output byte and every byte AROUND the output
    must be the oracle's (tools/soak.sh repeats this under other seeds)."""
    import os
    import random
    rng = random.Random(int(os.environ.get("ALZ_FUZZ_SEED", "1234")) * 31 + fmt)
    raw = test_bmp[30000:30000 + 180000]
    comp, aux = O.encode_stream(fmt, raw, quality=4)
    n = len(raw)
    with Context(0) as c:
        for k in range(16):
            b = bytearray(comp)
            kind = k % 4
            if kind == 0:
                for _ in range(rng.randrange(1, 4)):
                    b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            elif kind == 1:
                cut = rng.randrange(len(b))
                b[cut:cut] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 2000)))
            elif kind == 2:
                b = b[:rng.randrange(len(b) // 2, len(b))]
            else:
                i = rng.randrange(len(b) - 64)
                b[i:i + rng.randrange(1, 64)] = bytes(rng.choice([0, 0xFF]) for _ in range(1))[:1] * 1
            decl = rng.choice([n, n, n - 1, n + 1, n - rng.randrange(1, 5000)])
            cap = rng.choice([decl, decl + 100, n + 4096])
            if cap < 98304:
                cap = 98304
            _one(c, fmt, bytes(b), decl, aux.aux0, aux.aux1, cap=max(cap, decl) if fmt not in ELEM else cap, what="fuzz %d kind %d" % (k, kind))


@pytest.mark.parametrize("fmt", FMTS)
def test_prose_like_streams(fmt):
    """Text instead of a bitmap or the synthetic mix: short matches at many distances, three times the tokens per output byte -- one stream on the
    whole GPU and on its wavefronts, against the oracle."""
    from cases import prose_like
    with Context(0) as c:
        for size, q in ((40000, 0), (300000, 8), (1 << 20, 12)):
            raw = prose_like(size, size + q)
            comp, aux = O.encode_stream(fmt, raw, quality=q)
            _one(c, fmt, comp, size, aux.aux0, aux.aux1, expect_big=fmt not in ELEM or len(comp) >= 8192, what="prose %d" % size)
            c.big_stream(OFF)
            _one(c, fmt, comp, size, aux.aux0, aux.aux1, expect_big=False, what="prose %d, wavefront kernels" % size)
            c.big_stream(24 << 10)
        if fmt == A.FMT_LZO:
            # the reference's encoder writes two literal runs in a row here (tests/test_oracle_golden.py): its decoder reads the second as a
            # match from far in front of the stream -- whatever comes out, both paths must produce the oracle's bytes, status and lengths
            raw = b"     " + prose_like(200000, 5)
            comp, aux = O.encode_stream(fmt, raw, quality=0)
            assert O.decode_stream(fmt, comp, decom_len=0, cap=len(raw))[0] != raw
            _one(c, fmt, comp, len(raw), 0, 0, what="two literal runs")
            c.big_stream(OFF)
            _one(c, fmt, comp, len(raw), 0, 0, expect_big=False, what="two literal runs, wavefront kernels")
            c.big_stream(24 << 10)
