"""Python mirror of the reference's format classes (ICompressionAlgorithm surface) over the C ABI container layer.

Same member names and argument meaning as the managed classes so the parity tests read like the reference's own
(CompressionTest/CompressionAlgorithmTest.cs).  All work happens in libauroralz.so: header code in
csrc/alz_container.cpp, bodies on the GPU.  No CPU fallback.
"""
import ctypes as C

from . import _abi as A
from ._lib import AlzError, check, load


class DecompressedSizeException(Exception):
    """src/AuroraLib.Compression/Exceptions/DecompressedSizeException.cs"""


class EndOfStreamException(EOFError):
    pass


class InvalidIdentifierException(ValueError):
    pass


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

    def _opt(self):
        o = A.ContainerOptions()
        o.big_endian = 1 if self.FormatByteOrder == "Big" else 0
        o.memory_alignment = self.MemoryAlignment
        if self.lz is not None:
            o.lz = self.lz
        return o

    def IsMatch(self, data):
        data = bytes(data)
        return bool(load().alz_container_is_match(self.container, data, len(data)))

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
        """ICompressionDecoder.Decompress: returns the decompressed bytes; raises the reference's exception types."""
        data = bytes(data)
        if capacity is None:
            capacity = self.GetDecompressedSize(data) + 273 if self.provides_size else max(len(data) * 40, 1 << 16)
        o = self._opt()
        dst = C.create_string_buffer(max(capacity, 1))
        dl, su, st = C.c_size_t(), C.c_size_t(), C.c_int32()
        lib = load()
        lib.alz_container_decompress.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = lib.alz_container_decompress(_context().h, self.container, C.byref(o), data, len(data), dst, capacity, C.byref(dl), C.byref(su), C.byref(st))
        if rc == A.E_FORMAT:
            raise InvalidIdentifierException()
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
        return dst.raw[:dl.value]

    def Compress(self, data, settings=None):
        data = bytes(data)
        s = settings or CompressionSettings.Balanced
        st = A.Settings(s.Quality, s.MaxWindowBits, s.Strategy, 0)
        o = self._opt()
        lib = load()
        lib.alz_container_compress_bound.restype = C.c_size_t
        lib.alz_container_compress_bound.argtypes = [C.c_uint32, C.c_size_t]
        cap = lib.alz_container_compress_bound(self.container, len(data))
        dst = C.create_string_buffer(cap)
        dl = C.c_size_t()
        lib.alz_container_compress.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        check(lib.alz_container_compress(_context().h, self.container, C.byref(o), C.byref(st), data, len(data), dst, cap, C.byref(dl)))
        return dst.raw[:dl.value]


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
    container = A.C_YAZ0


class Yay0(_Format):
    container = A.C_YAY0


class MIO0(_Format):
    container = A.C_MIO0


class PRS(_Format):
    container, provides_size = A.C_PRS, False


class LZO(_Format):
    container, provides_size = A.C_LZO, False


ALL_FORMATS = [LZSS, LZ10, LZ11, Yaz0, Yay0, MIO0, PRS, LZO]
__all__ = [c.__name__ for c in ALL_FORMATS] + ["CompressionSettings", "DecompressedSizeException", "EndOfStreamException", "InvalidIdentifierException", "AlzError"]
