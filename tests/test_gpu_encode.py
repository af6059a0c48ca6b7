"""-m gpu: the GPU encoder (alz_encode_batch) against the oracle's restatement of LzChainMatchFinder + FlagWriter +
CompressHeaderless: compressed bytes must be IDENTICAL, for every format and quality class, and decode back."""
import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from gpu_common import ctx

pytestmark = pytest.mark.gpu
ALL = list(range(A.FMT_COUNT))


def _encode_and_compare(fmt, raws, quality, **kw):
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
    dst, res, aux = ctx().encode_batch(streams, src, do + 64, quality=quality, **kw)
    for i, r in enumerate(raws):
        try:
            want, waux = O.encode_stream(fmt, r, quality=quality, **kw)
        except ValueError:
            assert res[i].status != A.ST_OK, (A.FORMAT_NAMES[fmt], i, "oracle refuses, gpu accepted")
            continue
        assert res[i].status == A.ST_OK, (A.FORMAT_NAMES[fmt], quality, i, len(r), res[i].status)
        got = bytes(dst[streams[i].dst_off:streams[i].dst_off + res[i].dst_len])
        if got != want:
            k = next((j for j in range(min(len(got), len(want))) if got[j] != want[j]), min(len(got), len(want)))
            raise AssertionError("%s q%d stream %d (%d B): gpu %d B vs oracle %d B, first difference at %d" % (
                A.FORMAT_NAMES[fmt], quality, i, len(r), len(got), len(want), k))
        assert (aux[i].aux0, aux[i].aux1) == (waux.aux0, waux.aux1)


@pytest.mark.parametrize("fmt", ALL)
@pytest.mark.parametrize("quality", [0, 4, 8, 12, 15])
def test_encode_bit_identical_bmp(fmt, quality, test_bmp):
    raws = [test_bmp[:10], test_bmp[:10240], test_bmp[100000:100000 + 65536], test_bmp[500000:500000 + 30000], test_bmp[4096:4096 + 262144]]
    if quality >= 12:
        raws = raws[:4]
    _encode_and_compare(fmt, raws, quality)


@pytest.mark.parametrize("fmt", ALL)
def test_encode_edge_inputs(fmt):
    rng = np.random.default_rng(5)
    raws = [b"", b"a", b"ab", b"abc", b"abcd", b"abcde", bytes(5), bytes(15), bytes(16), bytes(17), bytes(0x100), bytes(5000), bytes(70000),
            b"ab" * 3000, b"abc" * 1000 + b"x", bytes(rng.integers(0, 256, 3000, dtype=np.uint8)),
            bytes(rng.integers(0, 4, 20000, dtype=np.uint8)), (b"0123456789" * 30 + bytes(rng.integers(0, 256, 50, dtype=np.uint8))) * 20]
    for q in (0, 8, 15):
        _encode_and_compare(fmt, raws, q)


def _token_soup(rng, size):
    """Short literal runs between short and long repeats at short and long distances: token starts on every lane of a 64-position window,
    literal runs of 0..5 (LZO counts 0-3 of them in the token in front), matches that end in the next window or many windows on."""
    out = bytearray(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tobytes())
    while len(out) < size:
        k = int(rng.integers(0, 10))
        if k < 3 or len(out) < 8:
            out += rng.integers(0, 256, int(rng.integers(1, 6)), dtype=np.uint8).tobytes()
        else:
            d = int(rng.integers(1, min(len(out), 70 if k < 7 else 3000) + 1))
            ln = int(rng.integers(2, 12)) if k < 8 else int(rng.integers(12, 400))
            for _ in range(ln):
                out.append(out[-d])
    return bytes(out[:size])


@pytest.mark.parametrize("fmt", [A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW, A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZO, A.FMT_LZ11, A.FMT_LZ40])
@pytest.mark.parametrize("quality", [0, 3, 8])
def test_encode_window_edges(fmt, quality):
    """The kernels that walk and emit 64 positions at a time (WinParse; at quality 0 with the search inside): buffers of every length around
    one, two and three windows, and token soup that puts starts, carried starts and match ends on every lane."""
    rng = np.random.default_rng(1000 * fmt + quality)
    sizes = list(range(1, 24)) + list(range(56, 72)) + list(range(120, 136)) + list(range(184, 200)) + [255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097]
    raws = [_token_soup(rng, n) for n in sizes] + [_token_soup(rng, int(rng.integers(300, 20000))) for _ in range(40)]
    raws += [bytes([7]) * n for n in (63, 64, 65, 128, 129, 5000)] + [(b"abcdefg" * 1000)[:n] for n in (64, 65, 127, 4099)]
    _encode_and_compare(fmt, raws, quality)


@pytest.mark.parametrize("fmt", [A.FMT_LZ10, A.FMT_LZ11])
def test_encode_vram_mode(fmt, test_bmp):
    _encode_and_compare(fmt, [test_bmp[:20000], bytes(300)], 8, min_distance=2)
    _encode_and_compare(fmt, [test_bmp[:20000], bytes(300)], 15, min_distance=2)


def test_encode_compatibility_mode(test_bmp):
    _encode_and_compare(A.FMT_LZSS, [test_bmp[:20000], bytes(300), b"ab" * 500], 8, strategy=1)


def test_encode_lzss_geometries(test_bmp):
    for bits in [(10, 6, 2), (12, 4, 2), (8, 4, 2)]:
        _encode_and_compare(A.FMT_LZSS, [test_bmp[:30000], bytes(1000)], 8, lz=A.LzProperties.from_bits(*bits))


def test_encode_synthetic_decoded_batch():
    """cfg5-style: raw buffers obtained by decoding synthetic LZSS streams, compressed as LZSS(12,4,2) at Q0 and Q8."""
    b = synth.make_batch(A.FMT_LZSS, 24, 65536, synth.seed_for(5))
    dst, res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=4)
    recs = synth.stream_records(b.streams)
    raws = [bytes(dst[int(recs["dst_off"][i]):int(recs["dst_off"][i]) + 65536]) for i in range(b.n)]
    for q in (0, 8):
        _encode_and_compare(A.FMT_LZSS, raws, q)


def test_encode_capacity_too_small(test_bmp):
    raw = test_bmp[:10240]
    streams = (A.Stream * 1)(A.Stream(0, 0, len(raw), 100, 0, 0, 0, A.FMT_LZ10))
    dst, res, aux = ctx().encode_batch(streams, np.frombuffer(raw + bytes(64), dtype=np.uint8), 4096, quality=8)
    assert res[0].status == A.ST_OUTPUT_CAPACITY


def test_container_compress_roundtrip(test_bmp):
    """ICompressionEncoder.Compress through the format-class mirror == the oracle's container bytes, and decodes back."""
    from auroralib.compression_amd import formats as F
    raw = test_bmp[:10240]
    for cls, cont in [(F.LZSS, A.C_LZSS), (F.LZ10, A.C_LZ10), (F.LZ11, A.C_LZ11), (F.Yaz0, A.C_YAZ0), (F.Yay0, A.C_YAY0), (F.MIO0, A.C_MIO0), (F.PRS, A.C_PRS), (F.LZO, A.C_LZO)]:
        for s in (F.CompressionSettings.Fastest, F.CompressionSettings.Balanced, F.CompressionSettings.Maximum):
            f = cls()
            comp = f.Compress(raw, s)
            assert comp == O.container_compress(cont, raw, quality=s.Quality), (cls.__name__, s.Quality)
            assert f.Decompress(comp, capacity=len(raw) + 300) == raw


def test_encode_matches_committed_vectors(test_bmp):
    """The GPU encoder against tests/golden/oracle_vectors.json -- no oracle involved at run time: every body of the
    round-trip matrix as ONE encode batch per quality, lengths and XXH64 digests compared with the committed ones."""
    import json
    import os
    import xxhash
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))
    by_q = {}
    for key, v in vec["bodies"].items():
        name, size, q = key.split(":")
        by_q.setdefault(int(q[1:]), []).append((A.FORMAT_NAMES.index(name), int(size), v, key))
    for q, items in by_q.items():
        n = len(items)
        raw = np.frombuffer(test_bmp[:max(s for _, s, _, _ in items)] + bytes(64), dtype=np.uint8).copy()
        streams = (A.Stream * n)()
        off = 0
        for i, (fmt, size, _, _) in enumerate(items):
            cap = size + size // 4 + 64
            streams[i] = A.Stream(0, off, size, cap, 0, 0, 0, fmt)
            off += (cap + 255) // 256 * 256
        dst, res, aux = ctx().encode_batch(streams, raw, off + 64, quality=q)
        auxv = np.frombuffer(aux, dtype=np.uint32).reshape(n, 2)
        for i, (fmt, size, (length, digest, a0, a1), key) in enumerate(items):
            assert res[i].status == 0 and res[i].dst_len == length, key
            body = bytes(dst[int(streams[i].dst_off):int(streams[i].dst_off) + length])
            assert xxhash.xxh64(body).hexdigest() == digest and (int(auxv[i, 0]), int(auxv[i, 1])) == (a0, a1), key


def test_scratch_is_reused_and_can_be_released(test_bmp):
    """The context keeps the encoder's device scratch between calls (alz_host.cpp: enc_buf) -- a second, smaller or larger call
    on dirty buffers and a call after alz_ctx_release_scratch give the same bytes as the first."""
    big = [test_bmp[4096:4096 + 200000], test_bmp[:70000]]
    small = [test_bmp[1000:1000 + 5000]]
    _encode_and_compare(A.FMT_YAY0, big, 8)            # (allocates the section buffers too)
    _encode_and_compare(A.FMT_LZSS, small, 12)         # smaller batch, larger hash table, min table: dirty slots
    _encode_and_compare(A.FMT_LZ4_BLOCK, big, 0)
    ctx().release_scratch()
    _encode_and_compare(A.FMT_LZSS, big, 8)
    b = synth.make_batch(A.FMT_YAZ0, 8, 50000, 99)
    g_dst, g_res = ctx().decode_batch(b.streams, b.src, b.dst_bytes)   # decode staging was released as well
    o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes)
    assert np.array_equal(g_dst[:b.dst_bytes], o_dst[:b.dst_bytes])


@pytest.mark.parametrize("quality,mwb", [(5, 14), (8, 17), (12, 16), (8, 13), (4, 17), (15, 15)])
def test_fastlz_level2(quality, mwb, test_bmp):
    """FastLZ.CompressHeaderless picks level 2 per source (>= 64 KiB, Quality > 4, MaxWindowBits > 13  FastLZ.cs:169-175): two
    property sets in the finder, long distances, 255-chains of length bytes.  One batch mixes sources on both sides of the 64 KiB
    line; bytes identical to the oracle's, and what was written decodes back."""
    raws = [test_bmp[:200000], test_bmp[300000:300000 + 65536], test_bmp[1000:1000 + 65535], test_bmp[:5000],
            bytes(70000), (b"0123456789abcdef" * 40 + bytes(np.random.default_rng(3).integers(0, 256, 100, dtype=np.uint8))) * 150]
    _encode_and_compare(A.FMT_FASTLZ, raws, quality, max_window_bits=mwb)
    if mwb > 13 and quality > 4:                               # the level-2 streams decode (level 1 with an enlarged finder window does not: the reference's own footgun)
        big = [r for r in raws if len(r) >= 0x10000]
        items = []
        for r in big:
            comp, aux = O.encode_stream(A.FMT_FASTLZ, r, quality=quality, max_window_bits=mwb)
            assert comp[0] >> 5 == 1
            items.append(dict(fmt=A.FMT_FASTLZ, src=comp, decom_len=len(r)))
        from gpu_common import compare_batch, pack_streams
        streams, src, dst_bytes = pack_streams(items)
        gr, g_dst = compare_batch(streams, src, dst_bytes, what="fastlz level 2")
        assert (gr["status"] == 0).all()
        recs = synth.stream_records(streams)
        for i, r in enumerate(big):
            a = int(recs["dst_off"][i])
            assert bytes(g_dst[a:a + len(r)]) == bytes(r)


def test_max_window_bits_within_and_beyond_the_format_window(test_bmp):
    """CompressionSettings.MaxWindowBits only widens the managed finder (LzChainMatchFinder.cs:69-73): within the format's own window it
    changes nothing -- the GPU encoder takes it and writes the oracle's bytes (which restates the max() rule) --, beyond it the managed
    finder returns distances the format cannot store and the call is refused (FastLZ aside: test_fastlz_level2)."""
    raws = [test_bmp[:30000], test_bmp[100000:100000 + 70000], bytes(3000)]
    for fmt, ok_bits, bad_bits in ((A.FMT_LZ10, (8, 12), (13, 14)), (A.FMT_YAZ0, (10, 12), (13,)), (A.FMT_LZ4_BLOCK, (12, 15), (16, 17)),   # (LZ4: 1 << 16 > 0xFFFF)
                                   (A.FMT_PRS_BE, (12,), (13,)), (A.FMT_LZO, (15,), (16,)), (A.FMT_SNAPPY_RAW, (15,), (16,)), (A.FMT_LZSS, (11, 12), (13,))):
        for b in ok_bits:
            for q in (0, 8):
                _encode_and_compare(fmt, raws, q, max_window_bits=b)
        for b in bad_bits:
            with pytest.raises(Exception):
                _encode_and_compare(fmt, raws[:1], 8, max_window_bits=b)


@pytest.mark.parametrize("quality", [0, 1, 3, 6, 8, 10, 12, 15])
def test_prev_links_through_the_lds_table(quality, test_bmp):
    """Kernel A with the head table in LDS (enc_prev_cu_kernel): 1, 2, 4, 8 and 16 passes (hashBits 15..19), streams whose positions pile
    up in one hash class (runs, short and long periods: the queue overflow paths), stream ends on and around every chunk boundary.
    Quality >= 10 adds the two passes of the min-length table (its own hash, its own links); 15 has 32 + 2 passes."""
    rng = np.random.default_rng(77 + quality)
    noise = bytes(rng.integers(0, 256, 70000, dtype=np.uint8))
    raws = [bytes(300000), b"\x01\x02\x03" * 50000, bytes(range(7)) * 9000, bytes(range(23)) * 5000, noise[:100] * 700,
            noise, noise[:24576 + 3], noise[:24575 + 3], noise[:2048 + 3], noise[:2047 + 3], noise[:2049 + 3], noise[:12288 + 4], noise[:3], noise[:4], noise[:5], noise[:67],
            bytes(40000) + noise[:30000] + bytes(range(5)) * 6000 + noise[:500] * 40, test_bmp[:200000],
            bytes(rng.integers(0, 3, 120000, dtype=np.uint8)), (noise[:300] + bytes(900)) * 100]
    if quality >= 10:                                                   # (the CPU restatement walks chains of up to 1 024 candidates there)
        raws = [r[:60000] for r in raws]
    _encode_and_compare(A.FMT_LZSS, raws, quality)
    _encode_and_compare(A.FMT_LZ4_BLOCK, raws[3:12], quality)          # (searches stop five bytes before the end: LZ4.cs:208)


@pytest.mark.parametrize("quality", [0, 8, 15])
def test_matches_beyond_the_match_array_entry(quality):
    """A match-array entry holds lengths below 2 046; longer ones (kernel B caps its compare at 2 040 bytes and marks the position, the
    parse recomputes it exactly) are written as an escape with the length in the NEXT entry -- for the match the parse takes, which may
    be the one at the cursor (R repeated right behind junk) or the lazy neighbour's (a 3-byte match `q R0 R1` at the cursor, all of R
    one byte on).  Formats whose lengths reach that far, windows that reach back to the first copy; zeros for the self-overlapping kind."""
    rng = np.random.default_rng(5)
    R = bytes(rng.integers(0, 256, 2600, dtype=np.uint8))
    J1, J2 = bytes(rng.integers(0, 256, 40, dtype=np.uint8)), bytes(rng.integers(0, 256, 300, dtype=np.uint8))
    lazy = b"q" + R[:2] + b"#" + J1 + R + J2 + b"q" + R + J1
    direct = J1 + R + J2 + R + R[:2100] + J1
    both = lazy + direct + bytes(2046) + b"z" + bytes(2047) + b"y" + bytes(2045) + b"x" + bytes(70000)
    for fmt in (A.FMT_LZ11, A.FMT_LZ40, A.FMT_LZ4_BLOCK, A.FMT_LZO, A.FMT_SNAPPY_RAW, A.FMT_HIG, A.FMT_REFPACK, A.FMT_WFLZ, A.FMT_YAZ0, A.FMT_PRS_BE, A.FMT_LZ02):
        _encode_and_compare(fmt, [lazy, direct, both, bytes(2046 + 1), bytes(2047 + 1), bytes(4100)], quality)
    _encode_and_compare(A.FMT_FASTLZ, [lazy + bytes(66000), direct + bytes(66000), both], quality, max_window_bits=17)


def test_window_bits_beyond_the_entry_are_refused(test_bmp):
    """A distance has 21 bits in the match array: FastLZ with MaxWindowBits above 20 goes back to the caller's own encoder."""
    _encode_and_compare(A.FMT_FASTLZ, [test_bmp[:70000]], 8, max_window_bits=20)
    with pytest.raises(Exception):
        _encode_and_compare(A.FMT_FASTLZ, [test_bmp[:70000]], 8, max_window_bits=21)
