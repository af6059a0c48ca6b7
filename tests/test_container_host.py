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
