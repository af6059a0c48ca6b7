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
