"""CPU: the header half of the format classes (csrc/alz_container.cpp) -- IsMatch / GetDecompressedSize -- against the
oracle's container layer.  These entry points are pure host code, so they run without a GPU."""
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import formats as F

CASES = [(F.LZSS, A.C_LZSS), (F.LZ10, A.C_LZ10), (F.LZ11, A.C_LZ11), (F.Yaz0, A.C_YAZ0), (F.Yay0, A.C_YAY0), (F.MIO0, A.C_MIO0)]


@pytest.mark.parametrize("cls,container", CASES)
def test_decompressed_size_and_is_match(cls, container, test_bmp):
    for raw, q in ((bytes(0x100), 0), (test_bmp[:10240], 8), (test_bmp[:100], 4)):
        comp = O.container_compress(container, raw, quality=q)
        f = cls()
        assert f.GetDecompressedSize(comp) == len(raw) == O.container_decompressed_size(container, comp)
        assert f.IsMatch(comp)
        with pytest.raises(F.InvalidIdentifierException):
            f.GetDecompressedSize(b"\x00\x01\x02\x03\x04\x05\x06\x07\x08")


def test_is_match_rejects_other_formats(test_bmp):
    raw = test_bmp[:4096]
    blobs = {c: O.container_compress(c, raw, quality=4) for c in (A.C_LZSS, A.C_LZ10, A.C_LZ11, A.C_YAZ0, A.C_YAY0, A.C_MIO0, A.C_PRS)}
    assert not F.Yaz0().IsMatch(blobs[A.C_YAY0]) and not F.Yay0().IsMatch(blobs[A.C_MIO0])
    assert not F.LZ10().IsMatch(blobs[A.C_LZ11]) and not F.LZ11().IsMatch(blobs[A.C_LZ10])
    assert F.PRS().IsMatch(blobs[A.C_PRS])            # PRS.GetByteOrder heuristic (PRS.cs:161-218)
    assert not F.LZSS().IsMatch(raw)


def test_large_nintendo_header():
    # size > 0xFFFFFF uses the 8-byte header form (LZ10.cs:69-77)
    hdr = bytes([0x10, 0, 0, 0]) + (0x1234567).to_bytes(4, "little") + bytes(16)
    assert F.LZ10().GetDecompressedSize(hdr) == 0x1234567


WRAPPERS = [(F.GCLZ, A.C_GCLZ), (F.CXLZ, A.C_CXLZ), (F.LZ_3DS, A.C_LZ_3DS), (F.COMP, A.C_COMP), (F.Yaz1, A.C_YAZ1), (F.AKLZ, A.C_AKLZ),
            (F.LZ01, A.C_LZ01), (F.LZSega, A.C_LZSEGA), (F.Level5LZSS, A.C_LEVEL5LZSS), (F.LZOn, A.C_LZON), (F.LZ77, A.C_LZ77), (F.Level5, A.C_LEVEL5)]


@pytest.mark.parametrize("cls,container", WRAPPERS)
def test_wrapper_headers_host_vs_oracle(cls, container, test_bmp):
    """Header-only wrapper formats: oracle round trip, product GetDecompressedSize / IsMatch."""
    for raw, q in ((test_bmp[:10240], 8), (test_bmp[:100], 4), (bytes(0x100), 0)):
        comp = O.container_compress(container, raw, quality=q)
        out, st = O.container_decompress(container, comp, cap=len(raw) + 300)
        assert st == A.ST_OK and out == raw
        f = cls()
        assert f.GetDecompressedSize(comp) == len(raw) == O.container_decompressed_size(container, comp)
        if container != A.C_LEVEL5:          # Level5.IsMatch leans on file extensions / zlib probing: not mirrored
            assert f.IsMatch(comp)


def test_lz77_chunk_mode_oracle(test_bmp):
    """LZ77 ChunkLZ10: table of u16 chunk end offsets + one LZ10 file per 4 KiB chunk (LZ77.cs:75-98, 131-150)."""
    raw = test_bmp[:50000]
    comp = O.container_compress(A.C_LZ77, raw, quality=8, variant=A.LZ77_CHUNKLZ10)
    assert comp[:4] == b"LZ77" and comp[4] == 0xF7 and int.from_bytes(comp[5:8], "little") == len(raw)
    nseg = (len(raw) + 0xFFF) // 0x1000
    ends = [int.from_bytes(comp[8 + 2 * i:10 + 2 * i], "little") for i in range(nseg)]
    assert ends == sorted(ends) and ends[-1] + 8 + 2 * nseg == len(comp)
    out, st = O.container_decompress(A.C_LZ77, comp, cap=len(raw))
    assert st == A.ST_OK and out == raw
    small = O.container_compress(A.C_LZ77, raw[:1000], quality=8, variant=A.LZ77_CHUNKLZ10)   # ChunkSize >= length: plain LZ10 file
    assert small[4] == 0x10
