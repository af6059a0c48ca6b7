"""-m gpu: the format-class mirror (header code in csrc/alz_container.cpp, bodies on the GPU) against the oracle's
container layer -- the reference's round-trip matrix read the way CompressionAlgorithmTest.cs reads."""
import os

import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import formats as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [(F.LZSS, A.C_LZSS), (F.LZ10, A.C_LZ10), (F.LZ11, A.C_LZ11), (F.Yaz0, A.C_YAZ0), (F.Yay0, A.C_YAY0), (F.MIO0, A.C_MIO0),
         (F.PRS, A.C_PRS), (F.LZO, A.C_LZO)]


def test_lzss_static_decoding():
    """LzssStaticDecodingTest (CompressionAlgorithmTest.cs:30-48) through the format-class mirror."""
    data = open(os.path.join(GOLD, "Test.lz"), "rb").read()
    lz = F.LZSS(A.LzProperties.from_bits(10, 6, 2))
    size = lz.GetDecompressedSize(data)
    out = lz.Decompress(data, capacity=size)
    assert len(out) == size == 1048726 and O.xxh64(out) == 11520079745250749767
    assert lz.last_src_used == len(data)


@pytest.mark.parametrize("cls,container", CASES)
@pytest.mark.parametrize("size,quality", [(10, 4), (10240, 8), (10240, 15), (1024 * 1024, 0)])
def test_decoding_match(cls, container, size, quality, test_bmp):
    """EncodingAndDecodingMatchTest_*: oracle-compressed container -> GPU Decompress == original."""
    raw = test_bmp[:size]
    comp = O.container_compress(container, raw, quality=quality)
    f = cls()
    out = f.Decompress(comp, capacity=None if f.provides_size else size + 64)
    assert O.xxh64(out) == O.xxh64(raw)


@pytest.mark.parametrize("cls,container", [(F.Yaz0, A.C_YAZ0), (F.PRS, A.C_PRS)])
def test_endianness_retry(cls, container, test_bmp):
    """Yaz0.cs:67-78 / PRS.cs:47-56: a stream written in the other byte order still decodes (catch -> retry)."""
    raw = test_bmp[:20000]
    comp = O.container_compress(container, raw, quality=8, big_endian=False)
    f = cls()                                   # FormatByteOrder = Big by default
    out = f.Decompress(comp, capacity=len(raw) + 300)
    assert out == raw


def test_exceptions(test_bmp):
    raw = test_bmp[:5000]
    comp = O.container_compress(A.C_LZ10, raw, quality=8)
    with pytest.raises(F.EndOfStreamException):
        F.LZ10().Decompress(comp[:len(comp) // 2])
    with pytest.raises(F.InvalidIdentifierException):
        F.LZ11().Decompress(comp)
    bad = bytes([0x10]) + (4000).to_bytes(3, "little") + comp[4:]     # declared size smaller than the stream decodes to
    with pytest.raises(F.DecompressedSizeException):
        F.LZ10().Decompress(bad)


WRAPPERS = [(F.GCLZ, A.C_GCLZ), (F.CXLZ, A.C_CXLZ), (F.LZ_3DS, A.C_LZ_3DS), (F.COMP, A.C_COMP), (F.Yaz1, A.C_YAZ1), (F.AKLZ, A.C_AKLZ),
            (F.LZ01, A.C_LZ01), (F.LZSega, A.C_LZSEGA), (F.Level5LZSS, A.C_LEVEL5LZSS), (F.LZOn, A.C_LZON), (F.LZ77, A.C_LZ77), (F.Level5, A.C_LEVEL5),
            (F.MDB4, A.C_MDB4), (F.FCMP, A.C_FCMP), (F.IECP, A.C_IECP), (F.GCZ, A.C_GCZ), (F.ECD, A.C_ECD), (F.SDPC, A.C_SDPC),
            (F.LZ40, A.C_LZ40), (F.LZ60, A.C_LZ60), (F.LZHudson, A.C_LZHUDSON), (F.SMSR00, A.C_SMSR00), (F.LZ00, A.C_LZ00), (F.FastLZ, A.C_FASTLZ), (F.CNX2, A.C_CNX2), (F.BLZ, A.C_BLZ), (F.CLZ0, A.C_CLZ0), (F.CNS, A.C_CNS), (F.LZ02, A.C_LZ02), (F.RefPack, A.C_REFPACK), (F.LZShrek, A.C_LZSHREK), (F.HIG, A.C_HIG)]


@pytest.mark.parametrize("cls,container", WRAPPERS)
def test_wrapper_formats_roundtrip(cls, container, test_bmp):
    """Header-only wrappers over the GPU bodies: Compress == oracle bytes, Decompress == original."""
    for size, q in ((10, 4), (10240, 8), (70000, 0), (10240, 15)):
        raw = test_bmp[:size]
        f = cls()
        kw = dict(key=0x5EED0000 + size) if cls is F.LZ00 else {}      # LZ00: keystream seed (the managed default is the clock)
        comp = f.Compress(raw, F.CompressionSettings(q), **kw)
        assert comp == O.container_compress(container, raw, quality=q, **kw), (cls.__name__, size, q)
        assert f.Decompress(comp, capacity=len(raw) + 300) == raw


def test_wflz_both_byte_orders(test_bmp):
    """WFLZ: FormatByteOrder selects the body variant; Compress == oracle bytes, Decompress == original, '!=' size rule."""
    for big in (False, True):
        for size, q in ((10, 4), (10240, 8), (70000, 0), (200000, 15)):
            raw = test_bmp[:size]
            f = F.WFLZ()
            f.FormatByteOrder = "Big" if big else "Little"
            comp = f.Compress(raw, F.CompressionSettings(q))
            assert comp == O.container_compress(A.C_WFLZ, raw, quality=q, big_endian=big), (big, size, q)
            assert f.Decompress(comp, capacity=len(raw) + 300) == raw
        bad = bytearray(comp)
        bad[8:12] = (len(raw) + 1).to_bytes(4, "big" if big else "little")
        with pytest.raises(F.DecompressedSizeException):
            f.Decompress(bytes(bad), capacity=len(raw) + 300)


def test_lz77_chunklz10_is_one_gpu_batch(test_bmp):
    """ChunkLZ10: independent 4 KiB LZ10 chunks, encoded and decoded as one batch."""
    raw = test_bmp[:60000]           # the u16 end offsets cap a ChunkLZ10 file at 64 KiB of compressed chunks (LZ77.cs:92-95)
    f = F.LZ77()
    f.Type = F.LZ77.ChunkLZ10
    comp = f.Compress(raw, F.CompressionSettings.Balanced)
    assert comp == O.container_compress(A.C_LZ77, raw, quality=8, variant=A.LZ77_CHUNKLZ10)
    assert f.Decompress(comp, capacity=len(raw)) == raw
    f2 = F.LZ77(); f2.Type = F.LZ77.LZ11
    c2 = f2.Compress(raw[:30000])
    assert c2 == O.container_compress(A.C_LZ77, raw[:30000], quality=8, variant=A.LZ77_LZ11) and f2.Decompress(c2) == raw[:30000]
    from auroralib.compression_amd._lib import AlzError
    with pytest.raises(AlzError):      # "chunks too large to process"
        f.Compress(test_bmp[:400000], F.CompressionSettings.Fastest)


# ---------------------------------------------------------------------------------------------- SURVEY.md 8f rank 2
import struct  # noqa: E402

import framing_cases as FC  # noqa: E402

FRAMED = [(F.LZ4, A.C_LZ4_FRAME, {}), (F.LZ4, A.C_LZ4_FRAME, {"chunk_size": 0x10000}), (F.LZ4Legacy, A.C_LZ4_LEGACY, {}), (F.Snappy, A.C_SNAPPY, {})]


@pytest.mark.parametrize("cls,container,kw", FRAMED)
def test_framing_roundtrip(cls, container, kw, test_bmp):
    """LZ4 frame / legacy and Snappy framing: GPU Compress == the oracle's bytes (blocks / chunks as one GPU batch),
    GPU Decompress == original (EncodingAndDecodingMatchTest incl. _LZ4Frame, CompressionAlgorithmTest.cs:81-139)."""
    for size, q in ((10, 4), (10240, 8), (10240, 15), (300000, 0), (1024 * 1024, 0)):
        raw = test_bmp[:size]
        f = cls(kw["chunk_size"]) if kw else cls()
        comp = f.Compress(raw, F.CompressionSettings(q))
        assert comp == O.container_compress(container, raw, quality=q, **kw), (cls.__name__, size, q)
        assert f.Decompress(comp) == raw
        assert f.last_src_used == len(comp)
    assert cls().Decompress(O.container_compress(container, bytes(0x100), quality=0, **kw)) == bytes(0x100)   # DataRecognitionTest


def test_snappy_stored_and_skippable_chunks():
    raw = os.urandom(70000) + bytes(70000) + os.urandom(100)
    f = F.Snappy()
    comp = f.Compress(raw)
    assert comp == O.container_compress(A.C_SNAPPY, raw, quality=8) and comp[10] == 1
    assert f.Decompress(comp) == raw
    padded = comp[:10] + bytes([0xFE, 3, 0, 0, 1, 2, 3]) + comp[10:]
    assert f.Decompress(padded) == raw
    with pytest.raises(F.InvalidIdentifierException):
        f.Decompress(comp[:10] + bytes([0x02, 0, 0, 0]) + comp[10:])
    with pytest.raises(BufferError):
        f.Decompress(comp, capacity=100000)
    with pytest.raises(F.EndOfStreamException):
        f.Decompress(comp[:len(comp) - 50])


def test_lz4_linked_frames():
    """One window per frame (LZ4.Frame.cs:120): blocks run in order, each with the frame's earlier output as history
    (alz_stream.aux0); results equal the oracle's single ring window and the byte-wise model."""
    blocks, expect = FC.lz4_linked_blocks(7, 5, 30000)
    f = F.LZ4()
    for flg in (0x40, 0x40 | 4 | 16, 0x40 | 8, 0x40 | 4 | 8 | 16):
        frame = FC.lz4_frame(blocks, O.xxh32, flg=flg, bd=0x40, content=expect)
        assert O.container_decompress(A.C_LZ4_FRAME, frame, cap=len(expect) + 16) == (expect, A.ST_OK)
        assert f.Decompress(frame) == expect
    blocks2, expect2 = FC.lz4_linked_blocks(11, 3, 200000)                 # far references: history beyond the LDS window
    assert f.Decompress(FC.lz4_frame(blocks2, O.xxh32, bd=0x50)) == expect2
    bad = bytearray(FC.lz4_frame(blocks, O.xxh32, flg=0x40 | 4, bd=0x40, content=expect)); bad[-1] ^= 1
    with pytest.raises(F.InvalidDataException):
        f.Decompress(bytes(bad))
    bad = bytearray(FC.lz4_frame(blocks, O.xxh32, flg=0x40 | 16, bd=0x40, content=expect)); bad[20] ^= 1
    with pytest.raises(F.InvalidDataException):
        f.Decompress(bytes(bad))
    with pytest.raises(F.DecompressedSizeException):
        f.Decompress(FC.lz4_frame(blocks, O.xxh32, flg=0x40 | 8, bd=0x40, content=expect, content_size=len(expect) + 1))
    with pytest.raises(F.EndOfStreamException):
        f.Decompress(FC.lz4_frame(blocks, O.xxh32)[:-30])
    # frames concatenate, skippable frames are skipped, anything else ends the file (LZ4.cs:50-93)
    b2, e2 = FC.lz4_linked_blocks(9, 1, 5000)
    leg = FC.lz4_legacy([b2[0], b2[0]])
    cat = FC.lz4_frame(blocks[:1], O.xxh32) + struct.pack("<II", 0x184D2A53, 5) + b"hello" + leg + b"\x01\x02\x03\x04junk"
    e1 = FC.lz4_linked_blocks(7, 1, 30000)[1]
    assert f.Decompress(cat) == e1 + e2 + e2 == O.container_decompress(A.C_LZ4_FRAME, cat, cap=1 << 20)[0]
    assert F.LZ4Legacy().Decompress(leg) == e2 + e2


def test_lz4_frame_flagged_independent_whose_blocks_still_reach_back():
    """The managed reader uses ONE LzWindows for all blocks of a frame whatever the block-independence flag says
    (LZ4.Frame.cs:120), so such a frame decodes there: the library walks the blocks' sequences on the host and
    decodes the frame in order when any match points in front of its block."""
    blocks, expect = FC.lz4_linked_blocks(21, 4, 20000)
    frame = FC.lz4_frame(blocks, O.xxh32, flg=0x40 | 32 | 4, bd=0x40, content=expect)
    assert O.container_decompress(A.C_LZ4_FRAME, frame, cap=len(expect) + 16) == (expect, A.ST_OK)
    assert F.LZ4().Decompress(frame) == expect
    # without a content checksum nothing else would have caught it
    assert F.LZ4().Decompress(FC.lz4_frame(blocks, O.xxh32, flg=0x40 | 32, bd=0x40)) == expect


def test_lz4_independent_blocks_are_one_batch(test_bmp):
    """Frames with the block-independence flag and legacy files: all blocks in one GPU batch at nominal offsets; a block
    that does not fill its slot (any but the last) falls back to in-order decoding."""
    raw = test_bmp[:0x70000] + os.urandom(0x10000) + test_bmp[:100000]     # the random 64 KiB block is stored
    frame = bytearray(O.container_compress(A.C_LZ4_FRAME, raw, quality=0, chunk_size=0x10000))
    assert any(struct.unpack("<I", frame[i:i + 4])[0] == 0x80010000 for i in range(7, len(frame) - 4))
    frame[4] |= 32                                                        # block independence (the header checksum is not verified)
    assert F.LZ4().Decompress(bytes(frame)) == raw
    # short blocks in the middle: nominal offsets are wrong from the second block on
    pieces = [test_bmp[:0x10000], test_bmp[0x10000:0x10000 + 5000], test_bmp[0x20000:0x30000]]
    blocks = [O.encode_stream(A.FMT_LZ4_BLOCK, p, quality=4)[0] for p in pieces]
    assert F.LZ4().Decompress(FC.lz4_frame(blocks, O.xxh32, flg=0x60, bd=0x40)) == b"".join(pieces)
    assert F.LZ4Legacy().Decompress(FC.lz4_legacy(blocks)) == b"".join(pieces)
    # the same frames WITHOUT the flag -- what the reference's own writer produces (LZ4.Frame.cs:184 clears it; every block has a finder of its own, LZ4.cs:205): no block
    # reaches back, so they are one batch as well (round 6: 16 MB in 64 KiB blocks took 112 ms one launch after the other, 3.3 as a batch); short blocks in the middle
    # fall back to in-order decoding, and a linked frame whose blocks DO reach back still decodes in order (test_lz4_linked_frames)
    plain = O.container_compress(A.C_LZ4_FRAME, raw, quality=0, chunk_size=0x10000)
    assert plain[4] & 32 == 0 and F.LZ4().Decompress(plain) == raw
    assert F.LZ4().Decompress(FC.lz4_frame(blocks, O.xxh32, flg=0x40, bd=0x40)) == b"".join(pieces)
    many = (test_bmp * 3)[:40 * 0x10000 + 1234]
    frame40 = O.container_compress(A.C_LZ4_FRAME, many, quality=8, chunk_size=0x10000)
    assert frame40[4] & 32 == 0 and F.LZ4().Decompress(frame40) == many == O.container_decompress(A.C_LZ4_FRAME, frame40, cap=len(many) + 16)[0]
    # legacy at its real block size (8 MiB): two blocks, nominal offsets hold
    big = (test_bmp[:4096] * 2200)[:0x800000 + 300000]
    comp = O.container_compress(A.C_LZ4_LEGACY, big, quality=0)
    assert struct.unpack("<I", comp[4:8])[0] + 8 < len(comp) - 5            # a second block follows
    assert F.LZ4Legacy().Decompress(comp) == big
