"""Python mirror of the reference's format classes (ICompressionAlgorithm surface) over the C ABI container layer.

Same member names and argument meaning as the managed classes so the parity tests read like the reference's own
(CompressionTest/CompressionAlgorithmTest.cs).  All work happens in libauroralz.so: header code in
csrc/alz_container.cpp, bodies on the GPU.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import AlzError, check, load


class DecompressedSizeException(Exception):
    """src/AuroraLib.Compression/Exceptions/DecompressedSizeException.cs"""


class EndOfStreamException(EOFError):
    pass


class InvalidIdentifierException(ValueError):
    pass


class InvalidDataException(ValueError):
    """System.IO.InvalidDataException (LZ4 frame checksum mismatch, LZ4.Frame.cs:27)."""


class CompressionSettings:
    """src/AuroraLib.Compression/CompressionSettings.cs:11-84"""

    def __init__(self, quality=8, max_window_bits=0, strategy=0):
        if not 0 <= quality <= 15:
            raise ValueError("quality")
        self.Quality, self.MaxWindowBits, self.Strategy = quality, max_window_bits, strategy


CompressionSettings.Fastest = CompressionSettings(0)
CompressionSettings.Fast = CompressionSettings(4)
CompressionSettings.Balanced = CompressionSettings(8)
CompressionSettings.High = CompressionSettings(12)
CompressionSettings.Maximum = CompressionSettings(15)

_ctx = None


def _context():
    global _ctx
    if _ctx is None:
        from .batch import Context
        _ctx = Context(0)
    return _ctx


class _Format:
    container = None
    provides_size = True

    def __init__(self):
        self.FormatByteOrder = "Big"      # IEndianDependentFormat.FormatByteOrder default (Yaz0.cs:30, PRS.cs:24)
        self.MemoryAlignment = 0          # Yaz0.MemoryAlignment
        self.lz = None
        self.Type = 0                     # LZ77.Type / Level5.Type (0 = class default)
        self.ChunkSize = 0                # LZ77.ChunkSize (0 = 0x1000)
        self.Key = 0                      # LZ00: keystream seed of the next Compress
        self.Name = ""                    # LZ00.Name ("" = "Temp.dat")

    def _opt(self):
        o = A.ContainerOptions()
        o.big_endian = 1 if self.FormatByteOrder == "Big" else 0
        o.memory_alignment = self.MemoryAlignment
        o.variant, o.chunk_size = self.Type, self.ChunkSize
        o.key = self.Key & 0xFFFFFFFF
        nm = self.Name.encode("latin-1")[:32]
        for i, b in enumerate(nm):
            o.name[i] = b
        if self.lz is not None:
            o.lz = self.lz
        return o

    def IsMatch(self, data):
        data = bytes(data)
        return bool(load().alz_container_is_match(self.container, data, len(data)))

    def _capacity_hint(self, data):
        """An upper bound on the decompressed size that the container's framing gives away without decoding (None: none)."""
        return None

    def GetDecompressedSize(self, data):
        if not self.provides_size:
            raise NotImplementedError("%s does not implement IProvidesDecompressedSize" % type(self).__name__)
        data = bytes(data)
        size, o = C.c_uint32(), self._opt()
        rc = load().alz_container_decompressed_size(self.container, C.byref(o), data, len(data), C.byref(size))
        if rc == A.E_FORMAT:
            raise InvalidIdentifierException()
        check(rc)
        return size.value

    def Decompress(self, data, capacity=None):
        """ICompressionDecoder.Decompress: returns the decompressed bytes; raises the reference's exception types.
        Formats without a size field grow the destination until it fits (a managed Stream grows by itself)."""
        data = bytes(data)
        if capacity is None and not self.provides_size:
            hint = self._capacity_hint(data)
            if hint is not None:
                try:
                    return self.Decompress(data, hint)
                except BufferError:
                    pass                                   # (a frame whose blocks decode to more than their nominal size: grow as for any other)
            cap = max(len(data) * 8, 1 << 16)
            while True:
                try:
                    return self.Decompress(data, cap)
                except BufferError:
                    if cap >= 1 << 31:
                        raise
                    cap *= 4
        if capacity is None and getattr(self, "size_either_order", False):
            # Yaz0.Decompress retries with the size field byte-swapped (Yaz0.cs:66-78) into a stream that grows by itself: the
            # capacity is the reading in FormatByteOrder unless that one is absurd, the other reading when it was not enough
            n = self.GetDecompressedSize(data)
            m = int.from_bytes(n.to_bytes(4, "big"), "little")
            first = n if n <= (256 << 20) else m
            try:
                return self.Decompress(data, first + 273)
            except BufferError:
                if max(n, m) <= first or max(n, m) > (1 << 31):
                    raise
                return self.Decompress(data, max(n, m) + 273)
        if capacity is None:
            capacity = self.GetDecompressedSize(data) + 273
        o = self._opt()
        dst_arr = np.empty(max(capacity, 1), dtype=np.uint8)       # (untouched memory: create_string_buffer writes the whole capacity first -- 68 MB of zeros for a 67 MB frame)
        dst = dst_arr.ctypes.data_as(C.c_void_p)
        dl, su, st = C.c_size_t(), C.c_size_t(), C.c_int32()
        lib = load()
        lib.alz_container_decompress.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = lib.alz_container_decompress(_context().h, self.container, C.byref(o), data, len(data), dst, capacity, C.byref(dl), C.byref(su), C.byref(st))
        if rc == A.E_FORMAT:
            raise InvalidIdentifierException()
        if rc == A.E_CHECKSUM:
            raise InvalidDataException("Checksum mismatch")
        if rc == A.E_STREAM:
            if st.value == A.ST_INPUT_TRUNCATED:
                raise EndOfStreamException()
            if st.value == A.ST_OUTPUT_SIZE_MISMATCH:
                raise DecompressedSizeException(dl.value)
            if st.value == A.ST_OUTPUT_CAPACITY:
                raise BufferError("destination too small")    # NotSupportedException of a fixed-size stream
            raise ValueError("bad token")
        check(rc)
        self.last_src_used = su.value
        return dst_arr[:dl.value].tobytes()

    def Compress(self, data, settings=None):
        data = bytes(data)
        s = settings or CompressionSettings.Balanced
        st = A.Settings(s.Quality, s.MaxWindowBits, s.Strategy, 0)
        o = self._opt()
        lib = load()
        lib.alz_container_compress_bound.restype = C.c_size_t
        lib.alz_container_compress_bound.argtypes = [C.c_uint32, C.c_size_t]
        cap = lib.alz_container_compress_bound(self.container, len(data))
        dst_arr = np.empty(max(cap, 1), dtype=np.uint8)
        dst = dst_arr.ctypes.data_as(C.c_void_p)
        dl = C.c_size_t()
        lib.alz_container_compress.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        check(lib.alz_container_compress(_context().h, self.container, C.byref(o), C.byref(st), data, len(data), dst, cap, C.byref(dl)))
        return dst_arr[:dl.value].tobytes()


class LZSS(_Format):
    """src/AuroraLib.Compression/Formats/Common/LZSS.cs"""
    container = A.C_LZSS

    def __init__(self, lz=None):
        super().__init__()
        self.lz = lz


class LZ10(_Format):
    container = A.C_LZ10


class LZ11(_Format):
    container = A.C_LZ11


class Yaz0(_Format):
    size_either_order = True
    container = A.C_YAZ0


class Yay0(_Format):
    container = A.C_YAY0


class MIO0(_Format):
    container = A.C_MIO0


class PRS(_Format):
    container, provides_size = A.C_PRS, False


class LZO(_Format):
    container, provides_size = A.C_LZO, False


# header-only wrappers over the same GPU bodies
class GCLZ(_Format):
    container = A.C_GCLZ


class CXLZ(_Format):
    container = A.C_CXLZ


class LZ_3DS(_Format):
    container = A.C_LZ_3DS


class COMP(_Format):
    container = A.C_COMP


class Yaz1(_Format):
    size_either_order = True
    container = A.C_YAZ1


class AKLZ(_Format):
    container = A.C_AKLZ


class LZ01(_Format):
    container = A.C_LZ01


class LZSega(_Format):
    container = A.C_LZSEGA


class HIG(_Format):
    """src/AuroraLib.Compression-Extended/Specialized/HIG.cs -- High Impact Games WAD LZ; writes version 6 with the default path."""
    container = A.C_HIG


class LZShrek(_Format):
    """src/AuroraLib.Compression-Extended/Activision/LZShrek.cs -- groups of literals + up to eight matches."""
    container = A.C_LZSHREK


class WFLZ(_Format):
    """src/AuroraLib.Compression-Extended/WayForward/WFLZ.cs -- 4-byte blocks + literals; FormatByteOrder defaults to little."""
    container = A.C_WFLZ

    def __init__(self):
        super().__init__()
        self.FormatByteOrder = "Little"           # WFLZ.cs:31


class RefPack(_Format):
    """src/AuroraLib.Compression-Extended/EA/RefPack.cs -- EA's RefPack / QFS; reads header versions 1-3, writes version 2."""
    container = A.C_REFPACK


class LZ02(_Format):
    """src/AuroraLib.Compression-Extended/Camelot/LZ02.cs -- flag-byte LZ that ends at a terminator token."""
    container = A.C_LZ02


class CNS(_Format):
    """src/AuroraLib.Compression-Extended/Specialized/CNS.cs -- byte-oriented runs / matches, 256-byte window."""
    container = A.C_CNS


class CLZ0(_Format):
    """src/AuroraLib.Compression-Extended/Marvelous/CLZ0.cs -- LZSS family, LSB-first flags with 1 = match."""
    container = A.C_CLZ0


class BLZ(_Format):
    """src/AuroraLib.Compression.Nintendo/Nintendo/BLZ.cs -- backwards LZ: code section stored back to front + footer."""
    container = A.C_BLZ


class CNX2(_Format):
    """src/AuroraLib.Compression.Sega/Sega/CNX2.cs -- 2-bit codes (skip / literal / match / literal run), 2 KiB window."""
    container = A.C_CNX2


class FastLZ(_Format):
    """src/AuroraLib.Compression/Formats/Common/FastLZ.cs -- headerless FastLZ stream (levels 1 / 2 decode, level 1 encode)."""
    container = A.C_FASTLZ
    provides_size = False


class LZ00(_Format):
    """src/AuroraLib.Compression.Sega/Sega/LZ00.cs -- LZSS body behind the StreamTransformer keystream.  Compress(data,
    settings, key=None): the managed parameterless overload seeds the keystream with the Unix time (LZ00.cs:64-68)."""
    container = A.C_LZ00

    def Compress(self, data, settings=None, key=None):
        import time
        self.Key = int(time.time()) if key is None else key
        return super().Compress(data, settings)

    def Decompress(self, data, capacity=None):
        out = super().Decompress(data, capacity)
        self.Name = bytes(data[16:48]).split(b"\0")[0].decode("latin-1")     # Name = source.ReadString(32)  LZ00.cs:50
        return out


class Level5LZSS(_Format):
    container = A.C_LEVEL5LZSS


class LZOn(_Format):
    container = A.C_LZON


class LZ4(_Format):
    """src/AuroraLib.Compression/Formats/Common/LZ4.cs + LZ4.Frame.cs: frame (default), legacy and skippable frames.
    BlockSize: 0x10000 / 0x40000 / 0x100000 / 0x400000 (default).  As in the reference, Compress writes a descriptor with
    only the version flag (`Flags &= IsVersion1`, LZ4.Frame.cs:184)."""
    container = A.C_LZ4_FRAME
    provides_size = False
    Block64KB, Block256KB, Block1MB, Block4MB = 0x10000, 0x40000, 0x100000, 0x400000

    def __init__(self, BlockSize=0):
        super().__init__()
        self.ChunkSize = BlockSize

    def _capacity_hint(self, data):
        return _lz4_capacity_hint(data)


def _lz4_capacity_hint(data):
        """blocks x the frame's maximum block size (LZ4.Frame.cs:107-174: FLG, BD, optional content size / dictionary id, header checksum; then u32 sizes; a legacy file:
        blocks of at most 8 MiB, LZ4.cs:96-111) -- a 67 MB frame made the growing loop decode twice, the second time into 256 MB.  None: not a file this walk understands."""
        total, pos, n = 0, 0, len(data)
        if n >= 8 and int.from_bytes(data[:4], "little") == 0x184C2102:
            pos = 4
            while pos + 4 <= n:
                bs = int.from_bytes(data[pos:pos + 4], "little")
                if bs in (0x184D2204, 0x184C2102) or (bs & 0xFFFFFFF0) == 0x184D2A50:
                    return None                            # (another file behind this one: the growing loop's)
                if pos + 4 + bs > n:
                    break
                total += 0x800000; pos += 4 + bs
                if pos < n and data[pos] == 0xFF:
                    break
            return total + 64 if total else None
        while pos + 7 <= n and int.from_bytes(data[pos:pos + 4], "little") == 0x184D2204:
            flg, bd = data[pos + 4], data[pos + 5]
            bmax = {4: 0x10000, 5: 0x40000, 6: 0x100000, 7: 0x400000}.get((bd >> 4) & 7)
            if bmax is None:
                return None
            pos += 7 + (8 if flg & 8 else 0) + (4 if flg & 1 else 0)
            while pos + 4 <= n:
                bsz = int.from_bytes(data[pos:pos + 4], "little"); pos += 4
                if bsz == 0:
                    break
                total += bmax
                pos += (bsz & 0x7FFFFFFF) + (4 if flg & 16 else 0)
            pos += 4 if flg & 4 else 0
        return total + 64 if total else None


class LZ4Legacy(_Format):
    """src/AuroraLib.Compression/Formats/Common/LZ4Legacy.cs"""
    container = A.C_LZ4_LEGACY
    provides_size = False

    def _capacity_hint(self, data):
        return _lz4_capacity_hint(data)


class Snappy(_Format):
    """src/AuroraLib.Compression/Formats/Common/Snappy.cs (framing format, 64 KiB chunks)."""
    container = A.C_SNAPPY
    provides_size = False


class LZ40(_Format):
    """src/AuroraLib.Compression.Nintendo/Nintendo/LZ40.cs -- LZ11-like tokens, little endian, negated flag bytes."""
    container = A.C_LZ40


class LZ60(_Format):
    """src/AuroraLib.Compression.Nintendo/Nintendo/LZ60.cs -- identifier 0x60 over LZ40's body."""
    container = A.C_LZ60


class LZHudson(_Format):
    """src/AuroraLib.Compression.Nintendo/HudsonSoft/LZHudson.cs -- Yay0's grammar, one stream, 32-bit flag words."""
    container = A.C_LZHUDSON


class SMSR00(_Format):
    """src/AuroraLib.Compression.Nintendo/Nintendo/SMSR00.cs -- 16-bit masks + MIO0 tokens | literals."""
    container = A.C_SMSR00


class MDB4(_Format):
    """src/AuroraLib.Compression-Extended/Specialized/MDB4.cs"""
    container = A.C_MDB4


class FCMP(_Format):
    container = A.C_FCMP


class IECP(_Format):
    container = A.C_IECP


class GCZ(_Format):
    """Konami/GCZ.cs -- recognised by its file extension only, so IsMatch(data) is always False here."""
    container = A.C_GCZ


class ECD(_Format):
    """Specialized/ECD.cs -- 4 plain bytes + LZSS(10,6,2); stored when Quality == 0 or compression does not pay."""
    container = A.C_ECD


class SDPC(_Format):
    container = A.C_SDPC


class LZ77(_Format):
    """src/AuroraLib.Compression.Nintendo/Nintendo/LZ77.cs -- Type: LZ10 (default) / LZ11 / ChunkLZ10."""
    container = A.C_LZ77
    LZ10, LZ11, ChunkLZ10 = A.LZ77_LZ10, A.LZ77_LZ11, A.LZ77_CHUNKLZ10


class Level5(_Format):
    container = A.C_LEVEL5
    OnlySave, LZ10 = A.LEVEL5_ONLYSAVE, A.LEVEL5_LZ10


ALL_FORMATS = [LZSS, LZ10, LZ11, Yaz0, Yay0, MIO0, PRS, LZO, LZ4, LZ4Legacy, Snappy, GCLZ, CXLZ, LZ_3DS, COMP, Yaz1, AKLZ, LZ01, LZSega, Level5LZSS, LZOn, MDB4, FCMP, IECP, GCZ, ECD, SDPC, LZ40, LZ60, LZHudson, SMSR00, LZ00, FastLZ, CNX2, BLZ, CLZ0, CNS, LZ02, RefPack, WFLZ, LZShrek, HIG, LZ77, Level5]
__all__ = [c.__name__ for c in ALL_FORMATS] + ["CompressionSettings", "DecompressedSizeException", "EndOfStreamException", "InvalidIdentifierException", "InvalidDataException", "AlzError"]
