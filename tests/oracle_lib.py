"""ctypes access to the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from auroralib.compression_amd import _abi as A

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.environ.get("ALZ_ORACLE_SO") or os.path.join(_ROOT, "oracle", "liboracle.so")   # (ALZ_ORACLE_SO: the sanitizer build, tests/test_sanitizers_cpu.py)


def _load():
    src = os.path.join(_ROOT, "oracle", "alz_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        from auroralib.compression_amd import _lib as _libmod        # (a process that has initialised the GPU must not start a compiler: __graft_entry__.build() first)
        _libmod.refuse_build_after_gpu(_SO)
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), os.path.basename(_SO)], stdout=subprocess.DEVNULL)
    lib = C.CDLL(_SO)
    lib.oracle_xxh64.restype = C.c_uint64
    lib.oracle_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    lib.oracle_xxh32.restype = C.c_uint32
    lib.oracle_xxh32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
    lib.oracle_crc32c.restype = C.c_uint32
    lib.oracle_crc32c.argtypes = [C.c_void_p, C.c_size_t]
    lib.oracle_decode_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_decode_stream.argtypes = [C.c_void_p] * 5
    lib.oracle_decode_stream_flat.argtypes = [C.c_void_p] * 5
    lib.oracle_encode_stream.restype = C.c_int64
    lib.oracle_encode_stream.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.oracle_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_container_decompress.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.oracle_container_compress.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.oracle_container_decompressed_size.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


lib = _load()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if isinstance(a, np.ndarray) else a


def xxh64(data, seed=0):
    b = bytes(data)
    return lib.oracle_xxh64(b, len(b), seed)


def xxh32(data, seed=0):
    b = bytes(data)
    return lib.oracle_xxh32(b, len(b), seed)


def crc32c(data):
    b = bytes(data)
    return lib.oracle_crc32c(b, len(b))


def _props_ptr(lz):
    return C.byref(lz) if lz is not None else None


def decode_stream(fmt, src, decom_len=0, cap=None, aux0=0, aux1=0, lz=None, flat=False):
    """Headerless decode of one stream. Returns (bytes, Result)."""
    src = bytes(src)
    if cap is None:
        cap = decom_len
    s = A.Stream(0, 0, len(src), cap, decom_len, aux0, aux1, fmt)
    dst = C.create_string_buffer(max(cap, 1))
    r = A.Result()
    fn = lib.oracle_decode_stream_flat if flat else lib.oracle_decode_stream
    fn(_props_ptr(lz), C.byref(s), src, dst, C.byref(r))
    return dst.raw[:r.dst_len], r


def decode_batch(streams, src, dst_bytes, lz=None, nthreads=1):
    """streams: ctypes array of A.Stream; src: np.uint8 array. Returns (dst np.uint8, results array)."""
    n = len(streams)
    dst = np.zeros(dst_bytes, dtype=np.uint8)
    res = (A.Result * n)()
    lib.oracle_decode_batch(_props_ptr(lz), n, _ptr(src), streams, _ptr(dst), res, nthreads)
    return dst, res


def encode_stream(fmt, data, quality=8, lz=None, strategy=0, min_distance=0, max_window_bits=0, cap=None):
    """Headerless encode. Returns (bytes, EncodeAux)."""
    data = bytes(data)
    if cap is None:
        cap = len(data) * 2 + 1024
    st = A.Settings(quality, max_window_bits, strategy, min_distance)
    aux = A.EncodeAux()
    dst = C.create_string_buffer(cap)
    n = lib.oracle_encode_stream(fmt, _props_ptr(lz), C.byref(st), data, len(data), dst, cap, C.byref(aux))
    if n < 0:
        raise ValueError("oracle_encode_stream failed: %d" % n)
    return dst.raw[:n], aux


def _opt(big_endian=True, lz=None, memory_alignment=0, variant=0, chunk_size=0, key=0, name=b""):
    o = A.ContainerOptions()
    o.key = key
    for i, b in enumerate(bytes(name)[:32]):
        o.name[i] = b
    o.big_endian = 1 if big_endian else 0
    o.memory_alignment = memory_alignment
    o.variant, o.chunk_size = variant, chunk_size
    if lz is not None:
        o.lz = lz
    return o


def container_decompressed_size(container, data, big_endian=True, lz=None):
    o = _opt(big_endian, lz)
    size = C.c_uint32()
    rc = lib.oracle_container_decompressed_size(container, C.byref(o), bytes(data), len(data), C.byref(size))
    if rc != 0:
        raise ValueError("bad header: %d" % rc)
    return size.value


def container_decompress(container, data, cap=None, big_endian=True, lz=None):
    data = bytes(data)
    o = _opt(big_endian, lz)
    if cap is None:
        try:
            cap = container_decompressed_size(container, data, big_endian, lz)
        except ValueError:
            cap = 1 << 24
    dst = C.create_string_buffer(max(cap, 1))
    dl, su, st = C.c_size_t(), C.c_size_t(), C.c_int32()
    rc = lib.oracle_container_decompress(container, C.byref(o), data, len(data), dst, cap, C.byref(dl), C.byref(su), C.byref(st))
    if rc not in (0, A.E_STREAM):
        e = ValueError("container_decompress rc=%d" % rc)
        e.rc = rc
        raise e
    return dst.raw[:dl.value], st.value


def container_compress(container, data, quality=8, big_endian=True, lz=None, strategy=0, min_distance=0, variant=0, chunk_size=0, key=0, name=b""):
    data = bytes(data)
    o = _opt(big_endian, lz, variant=variant, chunk_size=chunk_size, key=key, name=name)
    st = A.Settings(quality, 0, strategy, min_distance)
    cap = len(data) * 2 + 1024
    dst = C.create_string_buffer(cap)
    dl = C.c_size_t()
    rc = lib.oracle_container_compress(container, C.byref(o), C.byref(st), data, len(data), dst, cap, C.byref(dl))
    if rc != 0:
        raise ValueError("container_compress rc=%d" % rc)
    return dst.raw[:dl.value]
