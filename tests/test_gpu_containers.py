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
            (F.LZ01, A.C_LZ01), (F.LZSega, A.C_LZSEGA), (F.Level5LZSS, A.C_LEVEL5LZSS), (F.LZOn, A.C_LZON), (F.LZ77, A.C_LZ77), (F.Level5, A.C_LEVEL5)]


@pytest.mark.parametrize("cls,container", WRAPPERS)
def test_wrapper_formats_roundtrip(cls, container, test_bmp):
    """Header-only wrappers over the GPU bodies: Compress == oracle bytes, Decompress == original."""
    for size, q in ((10, 4), (10240, 8), (70000, 0), (10240, 15)):
        raw = test_bmp[:size]
        f = cls()
        comp = f.Compress(raw, F.CompressionSettings(q))
        assert comp == O.container_compress(container, raw, quality=q), (cls.__name__, size, q)
        assert f.Decompress(comp, capacity=len(raw) + 300) == raw


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
