"""ctypes mirror of include/auroralz.h (POD structs, enums).  No logic here."""
import ctypes as C

ABI_VERSION = 2

# alz_format
FMT_LZSS, FMT_LZ10, FMT_LZ11, FMT_YAZ0, FMT_YAY0, FMT_MIO0, FMT_PRS_BE, FMT_PRS_LE, FMT_LZ4_BLOCK, FMT_LZO, FMT_SNAPPY_RAW, FMT_LZ40, FMT_LZHUDSON, FMT_SMSR00, FMT_FASTLZ, FMT_CNX2, FMT_BLZ, FMT_CLZ0, FMT_CNS, FMT_LZ02, FMT_REFPACK, FMT_WFLZ, FMT_WFLZ_BE, FMT_LZSHREK, FMT_HIG = range(25)
FMT_COUNT = 25
FORMAT_NAMES = ["lzss", "lz10", "lz11", "yaz0", "yay0", "mio0", "prs_be", "prs_le", "lz4_block", "lzo", "snappy_raw", "lz40", "lzhudson", "smsr00", "fastlz", "cnx2", "blz", "clz0", "cns", "lz02", "refpack", "wflz", "wflz_be", "lzshrek", "hig"]

# alz_status
ST_OK, ST_INPUT_TRUNCATED, ST_OUTPUT_SIZE_MISMATCH, ST_OUTPUT_CAPACITY, ST_BAD_TOKEN = range(5)

# API errors
E_INVALID, E_NO_DEVICE, E_HIP, E_NOMEM, E_UNSUPPORTED, E_FORMAT, E_STREAM, E_CHECKSUM = -1, -2, -3, -4, -5, -6, -7, -8

# alz_container
C_LZSS, C_LZ10, C_LZ11, C_YAZ0, C_YAY0, C_MIO0, C_PRS, C_LZ4_LEGACY, C_LZO, C_SNAPPY = range(10)
C_GCLZ, C_CXLZ, C_LZ_3DS, C_COMP, C_YAZ1, C_AKLZ, C_LZ01, C_LZSEGA, C_LEVEL5LZSS, C_LZON, C_LZ77, C_LEVEL5 = range(10, 22)
C_LZ4_FRAME = 22
C_MDB4, C_FCMP, C_IECP, C_GCZ, C_ECD, C_SDPC, C_LZ40, C_LZ60, C_LZHUDSON, C_SMSR00 = range(23, 33)
C_LZ00 = 33
C_FASTLZ = 34
C_CNX2 = 35
C_BLZ = 36
C_CLZ0 = 37
C_CNS = 38
C_LZ02 = 39
C_REFPACK = 40
C_WFLZ = 41
C_LZSHREK = 42
C_HIG = 43
C_COUNT = 44
LZ77_LZ10, LZ77_LZ11, LZ77_CHUNKLZ10 = 0x10, 0x11, 0xF7
LEVEL5_ONLYSAVE, LEVEL5_LZ10 = 0, 1


class LzProperties(C.Structure):
    """alz_lz_properties == LzProperties (src/AuroraLib.Compression/LzProperties.cs:9-97)."""
    _fields_ = [("window_bits", C.c_uint8), ("length_bits", C.c_uint8), ("min_length", C.c_uint8), ("reserved0", C.c_uint8),
                ("windows_start", C.c_uint32), ("max_distance", C.c_uint32), ("reserved1", C.c_uint32)]

    @classmethod
    def from_bits(cls, distance_bits, length_bits, threshold=2):
        """LzProperties(byte distanceBits, byte lengthBits, byte threshold) -- LzProperties.cs:57-66."""
        md = 1 << distance_bits
        return cls(distance_bits, length_bits, threshold + 1, 0, md - (1 << length_bits) - threshold, md, 0)


class Stream(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("src_len", C.c_uint32), ("dst_cap", C.c_uint32),
                ("decom_len", C.c_uint32), ("aux0", C.c_uint32), ("aux1", C.c_uint32), ("format", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("dst_len", C.c_uint32), ("src_used", C.c_uint32), ("status", C.c_int32), ("reserved", C.c_uint32)]


class Settings(C.Structure):
    """alz_settings == CompressionSettings (src/AuroraLib.Compression/CompressionSettings.cs:11-84)."""
    _fields_ = [("quality", C.c_int32), ("max_window_bits", C.c_int32), ("strategy", C.c_int32), ("min_distance", C.c_int32)]


class EncodeAux(C.Structure):
    _fields_ = [("aux0", C.c_uint32), ("aux1", C.c_uint32)]


class ContainerOptions(C.Structure):
    _fields_ = [("big_endian", C.c_uint32), ("memory_alignment", C.c_uint32), ("lz", LzProperties), ("variant", C.c_uint32), ("chunk_size", C.c_uint32),
                ("key", C.c_uint32), ("name", C.c_uint8 * 32)]


assert C.sizeof(Stream) == 40 and C.sizeof(Result) == 16 and C.sizeof(LzProperties) == 16
