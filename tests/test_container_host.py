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
            (F.LZ01, A.C_LZ01), (F.LZSega, A.C_LZSEGA), (F.Level5LZSS, A.C_LEVEL5LZSS), (F.LZOn, A.C_LZON), (F.LZ77, A.C_LZ77), (F.Level5, A.C_LEVEL5),
            (F.MDB4, A.C_MDB4), (F.FCMP, A.C_FCMP), (F.IECP, A.C_IECP), (F.GCZ, A.C_GCZ), (F.ECD, A.C_ECD), (F.SDPC, A.C_SDPC),
            (F.LZ40, A.C_LZ40), (F.LZ60, A.C_LZ60), (F.LZHudson, A.C_LZHUDSON), (F.SMSR00, A.C_SMSR00), (F.LZ00, A.C_LZ00), (F.CNX2, A.C_CNX2), (F.BLZ, A.C_BLZ), (F.CLZ0, A.C_CLZ0), (F.CNS, A.C_CNS), (F.LZ02, A.C_LZ02), (F.RefPack, A.C_REFPACK), (F.LZShrek, A.C_LZSHREK), (F.HIG, A.C_HIG)]


@pytest.mark.parametrize("cls,container", WRAPPERS)
def test_wrapper_headers_host_vs_oracle(cls, container, test_bmp):
    """Header-only wrapper formats: oracle round trip, product GetDecompressedSize / IsMatch."""
    for raw, q in ((test_bmp[:10240], 8), (test_bmp[:100], 4), (bytes(0x100), 0)):
        comp = O.container_compress(container, raw, quality=q)
        out, st = O.container_decompress(container, comp, cap=len(raw) + 300)
        # (HIG: an initial literal block of exactly 2 bytes is written as count byte 0, which its decoder reads as "u16 count
        #  follows" (HIG.cs:235 against :137-138): data that starts with a run does not round-trip in the managed code either)
        assert (st == A.ST_OK and out == raw) or (container == A.C_HIG and raw[:3] == bytes(3))
        f = cls()
        assert f.GetDecompressedSize(comp) == len(raw) == O.container_decompressed_size(container, comp)
        if container not in (A.C_LEVEL5, A.C_GCZ):   # Level5 / GCZ lean on file extensions (and zlib probing): not mirrored
            assert f.IsMatch(comp)


def test_fastlz_oracle_and_validate(test_bmp):
    """FastLZ (Formats/Common/FastLZ.cs): headerless, level in the top bits of the first byte; IsMatch = Validate (:246-291),
    which -- as written -- accepts level-1 streams only; no IProvidesDecompressedSize."""
    import ctypes as C
    for raw, q in ((bytes(0x100), 0), (test_bmp[:10240], 8), (test_bmp[:70000], 15), (test_bmp[:7], 4)):
        comp = O.container_compress(A.C_FASTLZ, raw, quality=q)
        assert comp[0] < 0x20                                                       # level 1: first control byte is a literal run
        out, st = O.container_decompress(A.C_FASTLZ, comp, cap=len(raw) + 16)
        assert st == A.ST_OK and out == raw
        assert F.FastLZ().IsMatch(comp) == bool(O.lib.oracle_fastlz_validate(comp, len(comp)))
        if len(raw) >= 0x100:
            assert F.FastLZ().IsMatch(comp)                                         # DataRecognitionTest (CompressionAlgorithmTest.cs:60-80)
    with pytest.raises(NotImplementedError):
        F.FastLZ().GetDecompressedSize(b"\x00\x41")
    # hand-made level 2 stream: literal run "abcde" (first byte carries the level), match d=5 l=3+6+255+4 via chained length
    # bytes, then a far match through the 16-bit offset extension (distance 0x1FFF + 1 + 0 reaches before the start: zeros)
    l2 = bytes([0x20 | 4]) + b"abcde" + bytes([0xE0, 255, 4, 4]) + bytes([0x3F, 0xFF, 0x00, 0x00])
    out, st = O.container_decompress(A.C_FASTLZ, l2, cap=4096)
    assert st == A.ST_OK and len(out) == 5 + (3 + 6 + 255 + 4) + 3
    assert out[:5] == b"abcde" and out[5:5 + 268] == (b"abcde" * 60)[:268] and out[-3:] == bytes(3)
    assert not F.FastLZ().IsMatch(l2)                                               # (sic) Validate rejects level 2
    rng = __import__("random").Random(5)
    for _ in range(300):                                                            # Validate on noise: product == oracle
        blob = bytes(rng.choice([0, 1, 3, 0x1F, 0x20, 0x41, 0xE0, 0xFF, rng.randrange(256)]) for _ in range(rng.randrange(0, 40)))
        assert F.FastLZ().IsMatch(blob) == bool(O.lib.oracle_fastlz_validate(blob, len(blob))), blob.hex()
    for bad in (b"", bytes([0x40, 1, 2])):                                          # empty: IndexOutOfRange; level 3: InvalidDataException
        out, st = O.container_decompress(A.C_FASTLZ, bad, cap=64)
        assert st == (A.ST_INPUT_TRUNCATED if not bad else A.ST_BAD_TOKEN)


def test_wflz_oracle_both_byte_orders(test_bmp):
    """WFLZ (WayForward/WFLZ.cs): "WFLZ" + compressed size + size in FormatByteOrder (default little), body of 4-byte blocks."""
    for big in (False, True):
        for raw, q in ((test_bmp[:10240], 8), (test_bmp[:100], 4), (bytes(0x100), 0), (b"", 8)):
            comp = O.container_compress(A.C_WFLZ, raw, quality=q, big_endian=big)
            order = "big" if big else "little"
            assert comp[:4] == b"WFLZ" and int.from_bytes(comp[4:8], order) == len(comp) - 12 and int.from_bytes(comp[8:12], order) == len(raw)
            assert comp[-4:] == bytes(4)                                            # the end block
            out, st = O.container_decompress(A.C_WFLZ, comp, cap=len(raw) + 300, big_endian=big)
            assert st == A.ST_OK and out == raw
            f = F.WFLZ()
            assert f.FormatByteOrder == "Little"
            f.FormatByteOrder = "Big" if big else "Little"
            assert f.GetDecompressedSize(comp) == len(raw) == O.container_decompressed_size(A.C_WFLZ, comp, big_endian=big)
            assert f.IsMatch(comp) == (len(comp) > 0x10)
        # a wrong size field: DecompressedSizeException in both directions ('!=', WFLZ.cs:82-85)
        comp = bytearray(O.container_compress(A.C_WFLZ, test_bmp[:5000], quality=8, big_endian=big))
        for wrong in (4999, 5001):
            comp[8:12] = wrong.to_bytes(4, "big" if big else "little")
            out, st = O.container_decompress(A.C_WFLZ, bytes(comp), cap=6000, big_endian=big)
            assert st == A.ST_OUTPUT_SIZE_MISMATCH and len(out) == 5000


def test_lz00_keystream_oracle(test_bmp):
    """LZ00 (Sega/LZ00.cs): 64-byte header (magic, csize, name[32] at 16, size at 48, key at 52) + an LZSS body XORed with the
    keystream of the key -- key * 1103515245 + 12345 per byte, byte ^= (((key >> 16) & 0x7FFF) * 255) >> 15 (:128-141)."""
    raw = test_bmp[:20000]
    plain = O.container_compress(A.C_LZSEGA, raw, quality=8)[8:]                 # the same LZSS body without a keystream
    for key in (0, 1, 0x5F3759DF, 0xFFFFFFFF):
        comp = O.container_compress(A.C_LZ00, raw, quality=8, key=key, name=b"data.bin")
        assert comp[:4] == b"LZ00" and int.from_bytes(comp[4:8], "little") == len(comp) and comp[8:16] == bytes(8)
        assert comp[16:48] == b"data.bin".ljust(32, b"\0") and int.from_bytes(comp[48:52], "little") == len(raw)
        assert int.from_bytes(comp[52:56], "little") == key and comp[56:64] == bytes(8) and len(comp) == 64 + len(plain)
        k, ks = key, bytearray()
        for _ in range(len(plain)):
            k = (k * 1103515245 + 12345) & 0xFFFFFFFF
            ks.append(((((k >> 16) & 0x7FFF) * 255) >> 15) & 0xFF)
        assert bytes(a ^ b for a, b in zip(comp[64:], ks)) == plain
        out, st = O.container_decompress(A.C_LZ00, comp, cap=len(raw))
        assert st == A.ST_OK and out == raw
    assert O.container_compress(A.C_LZ00, raw[:10], quality=0)[16:24] == b"Temp.dat"      # default Name  LZ00.cs:30
    # GenerateNextKey as written (shift/add chain, :130-136) against the folded multiplier, on a few keys
    for key in (0, 1, 12345, 0xDEADBEEF):
        M = 0xFFFFFFFF
        x = (((key << 1) + key) << 5) - key
        x = (((x << 5) + key) << 7) - key
        x &= M
        x = ((x << 6) - x) & M
        x = ((x << 4) - x) & M
        assert (((x << 2) - x) + 12345) & M == (key * 1103515245 + 12345) & M


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


# ---------------------------------------------------------------------------------------------- SURVEY.md 8f rank 2
import struct  # noqa: E402

import framing_cases as FC  # noqa: E402


def test_checksum_known_answers():
    """XXH32 / CRC-32C published test vectors, and the frame-descriptor checksum bytes every `lz4` file starts with."""
    assert O.xxh32(b"") == 0x02CC5D05 and O.xxh32(b"abc") == 0x32D153FF
    assert O.crc32c(b"123456789") == 0xE3069283
    assert (O.xxh32(bytes([0x64, 0x40])) >> 8) & 0xFF == 0xA7      # 04 22 4D 18 64 40 A7: the lz4 CLI's default header
    assert (O.xxh32(bytes([0x40, 0x70])) >> 8) & 0xFF == 0xDF


@pytest.mark.parametrize("container,kw", [(A.C_LZ4_FRAME, {}), (A.C_LZ4_FRAME, {"chunk_size": 0x10000}), (A.C_LZ4_LEGACY, {}), (A.C_SNAPPY, {})])
def test_framing_oracle_roundtrip(container, kw, test_bmp):
    """EncodingAndDecodingMatchTest for LZ4 (frame default), LZ4Legacy and Snappy (CompressionAlgorithmTest.cs:81-139)."""
    for raw, q in ((test_bmp[:10], 4), (test_bmp[:10240], 8), (test_bmp[:10240], 15), (test_bmp[:300000], 0), (bytes(0x100), 0)):
        comp = O.container_compress(container, raw, quality=q, **kw)
        out, st = O.container_decompress(container, comp, cap=len(raw) + 64)
        assert st == A.ST_OK and out == raw
        cls = {A.C_LZ4_FRAME: F.LZ4, A.C_LZ4_LEGACY: F.LZ4Legacy, A.C_SNAPPY: F.Snappy}[container]
        if len(comp) > 0x11:
            assert cls().IsMatch(comp)                                   # DataRecognitionTest (:60-80)
    comp = O.container_compress(container, test_bmp[:200000], quality=0, **kw)
    if container == A.C_LZ4_FRAME:
        bd = 0x40 if kw else 0x70
        assert comp[:7] == bytes([0x04, 0x22, 0x4D, 0x18, 0x40, bd, (O.xxh32(bytes([0x40, bd])) >> 8) & 0xFF])   # Flags &= IsVersion1
        assert comp[-4:] == bytes(4)                                      # EndMark, no content checksum
    elif container == A.C_LZ4_LEGACY:
        assert comp[:4] == bytes([0x02, 0x21, 0x4C, 0x18]) and comp[-1] == 0xFF
        assert struct.unpack("<I", comp[4:8])[0] == len(comp) - 9         # one block (< 8 MiB)
    else:
        assert comp[:10] == bytes([0xff, 6, 0, 0]) + b"sNaPpY" and comp[10] == 0
        n0 = int.from_bytes(comp[11:14], "little")
        assert struct.unpack("<I", comp[14:18])[0] == ((lambda c: (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF)(O.crc32c(test_bmp[:0x10000])))
        assert comp[10 + 4 + n0] in (0, 1)                                # next chunk header follows the declared length


def test_snappy_stored_chunk_and_skippable():
    import os
    raw = os.urandom(70000)                                               # incompressible: stored chunks (Snappy.cs:89-96)
    comp = O.container_compress(A.C_SNAPPY, raw, quality=8)
    assert comp[10] == 1 and int.from_bytes(comp[11:14], "little") == 0x10000 + 4
    out, st = O.container_decompress(A.C_SNAPPY, comp, cap=len(raw))
    assert st == A.ST_OK and out == raw
    assert O.container_decompressed_size(A.C_SNAPPY, comp) == len(raw)
    padded = comp[:10] + bytes([0xFE, 3, 0, 0, 1, 2, 3]) + comp[10:]      # skippable chunk 0x80..0xFE
    out, st = O.container_decompress(A.C_SNAPPY, padded, cap=len(raw))
    assert st == A.ST_OK and out == raw
    with pytest.raises(ValueError):                                       # reserved unskippable chunk 0x02..0x7F
        O.container_decompress(A.C_SNAPPY, comp[:10] + bytes([0x02, 0, 0, 0]) + comp[10:], cap=len(raw))


def test_lz4_linked_frame_oracle():
    """Blocks of a frame share one window (LZ4.Frame.cs:120): matches reach into earlier blocks."""
    blocks, expect = FC.lz4_linked_blocks(7, 5, 30000)
    for flg in (0x40, 0x40 | 4 | 16, 0x40 | 8, 0x40 | 4 | 8 | 16):
        frame = FC.lz4_frame(blocks, O.xxh32, flg=flg, bd=0x40, content=expect)
        out, st = O.container_decompress(A.C_LZ4_FRAME, frame, cap=len(expect) + 16)
        assert st == A.ST_OK and out == expect
    bad = bytearray(FC.lz4_frame(blocks, O.xxh32, flg=0x40 | 4, bd=0x40, content=expect)); bad[-1] ^= 1
    with pytest.raises(ValueError) as e:
        O.container_decompress(A.C_LZ4_FRAME, bytes(bad), cap=len(expect) + 16)
    assert e.value.rc == A.E_CHECKSUM
    wrong = FC.lz4_frame(blocks, O.xxh32, flg=0x40 | 8, bd=0x40, content=expect, content_size=len(expect) + 1)
    out, st = O.container_decompress(A.C_LZ4_FRAME, wrong, cap=len(expect) + 16)
    assert st == A.ST_OUTPUT_SIZE_MISMATCH                                 # LZ4.Frame.cs:152-155
    # legacy: a fresh window per block (LZ4.cs:164) -> references across blocks read zeros (E2), not the earlier block
    b2, e2 = FC.lz4_linked_blocks(9, 1, 5000)
    leg = FC.lz4_legacy([b2[0], b2[0]])
    out, st = O.container_decompress(A.C_LZ4_LEGACY, leg, cap=2 * len(e2))
    assert st == A.ST_OK and out == e2 + e2
    # frames concatenate; a skippable frame in between is skipped; trailing junk stops the loop (LZ4.cs:50-93)
    f1 = FC.lz4_frame(blocks[:1], O.xxh32)
    skip = struct.pack("<II", 0x184D2A53, 5) + b"hello"
    cat = f1 + skip + leg + b"\x01\x02\x03\x04junk"
    e1 = FC.lz4_linked_blocks(7, 1, 30000)[1]
    out, st = O.container_decompress(A.C_LZ4_FRAME, cat, cap=len(e1) + 2 * len(e2) + 16)
    assert st == A.ST_OK and out == e1 + e2 + e2


def test_lz4_capacity_hint_of_the_wrapper(test_bmp):
    """formats._lz4_capacity_hint: blocks x the frame's block maximum (a legacy file: x 8 MiB) -- an upper bound the wrapper decodes into before it falls back to growing a
    buffer; never below the true size for files the library's own writer (= the reference's) produces, None for what it does not understand."""
    raw = test_bmp[:300000]
    for bs, bmax in ((0x10000, 0x10000), (0x40000, 0x40000), (0, 0x400000)):
        frame = O.container_compress(A.C_LZ4_FRAME, raw, quality=0, chunk_size=bs)
        nblocks = (len(raw) + bmax - 1) // bmax
        assert F._lz4_capacity_hint(frame) == nblocks * bmax + 64 >= len(raw)
        assert F._lz4_capacity_hint(frame + frame) == 2 * nblocks * bmax + 64           # frames concatenate (LZ4.cs:50-93)
    legacy = O.container_compress(A.C_LZ4_LEGACY, raw, quality=0)
    assert F._lz4_capacity_hint(legacy) == 0x800000 + 64
    assert legacy[-1] == 0xFF                                                            # the EOF flag: what follows it is not read (LZ4.cs:96-111)
    assert F._lz4_capacity_hint(legacy + O.container_compress(A.C_LZ4_FRAME, raw, quality=0)) == 0x800000 + 64
    assert F._lz4_capacity_hint(legacy[:-1] + O.container_compress(A.C_LZ4_FRAME, raw, quality=0)) is None   # another file right behind the last block: the growing loop's
    assert F._lz4_capacity_hint(b"") is None and F._lz4_capacity_hint(b"\x00" * 64) is None and F._lz4_capacity_hint(raw[:100]) is None
    blocks, expect = FC.lz4_linked_blocks(7, 5, 30000)
    assert F._lz4_capacity_hint(FC.lz4_frame(blocks, O.xxh32, flg=0x40 | 4 | 8 | 16, bd=0x40, content=expect)) == 5 * 0x10000 + 64
