"""Pins the CPU oracle against the reference's own fixtures.

 (1) CompressionTest/Test.lz -> XXH64 11520079745250749767   CompressionAlgorithmTest.cs:31-48
 (2) round-trip matrix on Test.bmp prefixes                   CompressionAlgorithmTest.cs:81-130
 (3) DataRecognitionTest: 0x100 zero bytes @Fastest          CompressionAlgorithmTest.cs:60-80
 (4) published compression ratios on Test.bmp[0:1024000]      Benchmarks.md (encoder restatement)
"""
import os

import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_xxh64_known_answers():
    # standard XXH64 vectors (xxHash reference implementation)
    assert O.xxh64(b"") == 0xEF46DB3751D8E999
    assert O.xxh64(b"a") == 0xD24EC4F1A98C6E5B
    assert O.xxh64(b"abc") == 0x44BC2CF5AD770999
    assert O.xxh32(b"") == 0x02CC5D05
    assert O.xxh32(b"abc") == 0x32D153FF
    assert O.crc32c(b"123456789") == 0xE3069283


def test_lzss_static_decoding_kat():
    """LzssStaticDecodingTest: the reference's only absolute known-answer test."""
    data = open(os.path.join(GOLD, "Test.lz"), "rb").read()
    lz = A.LzProperties.from_bits(10, 6, 2)          # new LZSS(new LzProperties((byte)10, 6, 2))
    size = O.container_decompressed_size(A.C_LZSS, data, lz=lz)
    assert size == 1048726
    out, st = O.container_decompress(A.C_LZSS, data, cap=size, lz=lz)
    assert st == A.ST_OK and len(out) == size
    assert O.xxh64(out) == 11520079745250749767
    # the flat out[q]=out[q-d] model (what the GPU kernels implement) gives the same bytes
    body = data[16:]
    flat, r = O.decode_stream(A.FMT_LZSS, body, decom_len=size, lz=lz, flat=True)
    assert r.status == A.ST_OK and flat == out and r.src_used == 285913


HEADERLESS = [A.FMT_LZSS, A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_YAY0, A.FMT_MIO0, A.FMT_PRS_BE, A.FMT_PRS_LE,
              A.FMT_LZ4_BLOCK, A.FMT_LZO, A.FMT_SNAPPY_RAW, A.FMT_LZ40, A.FMT_LZHUDSON, A.FMT_SMSR00]


def _roundtrip(fmt, raw, quality, **kw):
    comp, aux = O.encode_stream(fmt, raw, quality=quality, **kw)
    cap = len(raw)
    out, r = O.decode_stream(fmt, comp, decom_len=len(raw), cap=cap, aux0=aux.aux0, aux1=aux.aux1)
    assert r.status == A.ST_OK, (A.FORMAT_NAMES[fmt], quality, r.status)
    assert out == raw
    out2, r2 = O.decode_stream(fmt, comp, decom_len=len(raw), cap=cap, aux0=aux.aux0, aux1=aux.aux1, flat=True)
    assert r2.status == A.ST_OK and out2 == raw and r2.src_used == r.src_used
    if fmt not in (A.FMT_YAY0, A.FMT_MIO0):
        assert r.src_used == len(comp)
    return comp


@pytest.mark.parametrize("fmt", HEADERLESS)
def test_roundtrip_10kb_balanced(fmt, test_bmp):
    _roundtrip(fmt, test_bmp[:10240], 8)


@pytest.mark.parametrize("fmt", HEADERLESS)
def test_roundtrip_10kb_maximum(fmt, test_bmp):
    _roundtrip(fmt, test_bmp[:10240], 15)


@pytest.mark.parametrize("fmt", HEADERLESS)
def test_roundtrip_1mb_fastest(fmt, test_bmp):
    _roundtrip(fmt, test_bmp[:1024 * 1024], 0)


@pytest.mark.parametrize("fmt", HEADERLESS)
def test_roundtrip_10b(fmt, test_bmp):
    # CompressionLevel.Fastest -> CompressionSettings.Fast (quality 4), CompressionSettings.cs:54-63
    _roundtrip(fmt, test_bmp[:10], 4)


@pytest.mark.parametrize("fmt", [A.FMT_LZ10, A.FMT_LZ11])
def test_roundtrip_vram_mode(fmt, test_bmp):
    comp = _roundtrip(fmt, test_bmp[:10240], 8, min_distance=2)
    assert len(comp) > 0


CONTAINERS = [A.C_LZSS, A.C_LZ10, A.C_LZ11, A.C_YAZ0, A.C_YAY0, A.C_MIO0, A.C_PRS, A.C_LZO]


@pytest.mark.parametrize("container", CONTAINERS)
@pytest.mark.parametrize("size,quality", [(10240, 8), (10240, 15), (10, 4)])
def test_container_roundtrip(container, size, quality, test_bmp):
    raw = test_bmp[:size]
    comp = O.container_compress(container, raw, quality=quality)
    out, st = O.container_decompress(container, comp, cap=len(raw))
    assert st == A.ST_OK and out == raw


@pytest.mark.parametrize("container", [A.C_LZSS, A.C_LZ10, A.C_LZ11, A.C_YAZ0, A.C_YAY0, A.C_MIO0])
def test_data_recognition_zero_block(container):
    """DataRecognitionTest: 0x100 zero bytes at Fastest; GetDecompressedSize == 0x100."""
    raw = bytes(0x100)
    comp = O.container_compress(container, raw, quality=0)
    assert O.container_decompressed_size(container, comp) == 0x100
    out, st = O.container_decompress(container, comp)
    assert st == A.ST_OK and out == raw


# Benchmarks.md ratios (Ratio % = compressed/raw*100, whole container) on Test.bmp[0:1,024,000].
# The Q0 byte counts are the survey's restatement results.
RATIO_PINS_Q0 = {A.C_LZ10: (261953, 25.58), A.C_LZSS: (261898, 25.58), A.C_YAZ0: (183160, 17.89),
                 A.C_YAY0: (183160, 17.89), A.C_LZ11: (179455, 17.52), A.C_MIO0: (None, 25.58), A.C_PRS: (None, 16.18),
                 A.C_LZO: (None, 15.74), A.C_FASTLZ: (None, 16.20), A.C_CNX2: (None, 26.34), A.C_BLZ: (None, 33.74), A.C_CLZ0: (None, 25.58), A.C_CNS: (None, 26.74), A.C_LZ02: (None, 19.56), A.C_REFPACK: (None, 16.99), A.C_WFLZ: (None, 19.89), A.C_LZSHREK: (None, 20.81)}
# (FastLZ at Q15: level 1 gives 14.11 % against 13.99 % published -- the published run predates the MaxWindowBits > 13
# condition of FastLZ.cs:169-175 or set it, i.e. it wrote level 2: pinned separately below.)
RATIO_PINS_Q15 = {A.C_LZ10: 22.84, A.C_LZSS: 22.84, A.C_YAZ0: 15.01, A.C_YAY0: 15.01, A.C_LZ11: 14.28, A.C_MIO0: 22.84,
                  A.C_PRS: 13.83, A.C_LZO: 11.29, A.C_CNX2: 24.80, A.C_BLZ: 22.86, A.C_CLZ0: 22.84, A.C_CNS: 26.69, A.C_REFPACK: 11.46, A.C_WFLZ: 14.03, A.C_LZSHREK: 20.09}
# (LZ02 at Q15: 16.47 % here against 16.57 % published, 0.10 below -- just outside the band of the others; Q0 is pinned.)


@pytest.mark.parametrize("container", sorted(RATIO_PINS_Q0))
def test_published_ratio_q0(container, test_bmp):
    raw = test_bmp[:1024000]
    comp = O.container_compress(container, raw, quality=0)
    nbytes, pct = RATIO_PINS_Q0[container]
    assert round(len(comp) / len(raw) * 100, 2) == pytest.approx(pct, abs=0.011), len(comp)
    if nbytes is not None:
        assert abs(len(comp) - nbytes) <= 16  # header bytes are counted differently per container


@pytest.mark.parametrize("container", sorted(RATIO_PINS_Q15))
def test_published_ratio_q15(container, test_bmp):
    raw = test_bmp[:1024000]
    comp = O.container_compress(container, raw, quality=15)
    # Soft pin: Q0 reproduces Benchmarks.md to two decimals for every format, Q15 lands 0.00-0.08
    # percentage points BELOW the published ratio for the 4 KiB-window formats (LZO: exact).  Benchmarks.md
    # carries no commit id; the current LzChainMatchFinder source is what the oracle restates.
    assert round(len(comp) / len(raw) * 100, 2) == pytest.approx(RATIO_PINS_Q15[container], abs=0.09), len(comp)


def test_published_ratio_q15_fastlz_level2(test_bmp):
    """Benchmarks.md:19: FastLZ at Q15 = 13.99 %.  Level 1 (what MaxWindowBits = 0 selects today) gives 14.11 %; level 2 -- two
    property sets in the finder, FastLZ.cs:23-27 -- gives 13.96 %, inside the 0.00-0.08 band of the other Q15 figures, and its
    first byte carries the level tag."""
    raw = test_bmp[:1024000]
    comp, _ = O.encode_stream(A.FMT_FASTLZ, raw, quality=15, max_window_bits=14)
    assert comp[0] >> 5 == 1
    assert round(len(comp) / len(raw) * 100, 2) == pytest.approx(13.99, abs=0.09), len(comp)
    out, st = O.container_decompress(A.C_FASTLZ, comp, cap=len(raw))
    assert st == A.ST_OK and out == raw
    lvl1, _ = O.encode_stream(A.FMT_FASTLZ, raw, quality=15)
    assert lvl1[0] >> 5 == 0 and len(lvl1) > len(comp)


def test_committed_oracle_vectors(test_bmp):
    """tests/golden/oracle_vectors.json (made by tests/golden/make_vectors.py): the encoders' output for the round-trip
    matrix is frozen -- any drift of the restatement shows up here, not as a silent change of what 'parity' means."""
    import json
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))
    for key, (length, digest, a0, a1) in list(vec["bodies"].items()) + list(vec["benchmark"].items()):   # (benchmark: the reference's own benchmark input, 1 000 KiB)
        name, size, q = key.split(":")
        comp, aux = O.encode_stream(A.FORMAT_NAMES.index(name), test_bmp[:int(size)], quality=int(q[1:]))
        assert (len(comp), "%016x" % O.xxh64(comp), aux.aux0, aux.aux1) == (length, digest, a0, a1), key
    for key, (length, digest) in vec["containers"].items():
        name, size, q = key.split(":")
        comp = O.container_compress(getattr(A, "C_" + name), test_bmp[:int(size)], quality=int(q[1:]))
        assert (len(comp), "%016x" % O.xxh64(comp)) == (length, digest), key


def test_lzo_two_literal_runs_in_a_row_is_the_reference():
    """A quirk of the reference that parity keeps: LZO.CompressHeaderless (Formats/Common/LZO.cs:167-188) moves a match that starts within the
    first three bytes to offset 4 and shortens it; when fewer than three bytes of it are left the match is dropped -- and the literals behind it
    go out as a SECOND literal-run instruction right behind the first.  In LZO1X a byte below 16 behind a literal run is a match, not a run
    (LZO.cs:70-95: the run sets plain = 4, and flag code 0 with plain > 3 is "D = 2049-3072, L = 3"), so the reference's own decoder does not read this stream back.  Found on program text (five spaces, then code: a match at
    offset 1, distance 1, length 4); the bytes below follow from the C# by hand: run of 4 = 0x01 + 4 literals, run of 12 = 0x09 + 12 literals,
    the end token 0x11 0x00 0x00."""
    raw = b"     const u32 x"
    comp, _ = O.encode_stream(A.FMT_LZO, raw, quality=0)
    assert comp == bytes([0x01]) + raw[:4] + bytes([0x09]) + raw[4:] + bytes([0x11, 0x00, 0x00])
    out, r = O.decode_stream(A.FMT_LZO, comp, decom_len=0, cap=64)
    assert out != raw                                              # (what the managed decoder makes of it: not the input)
