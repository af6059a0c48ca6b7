"""Batch producers of the reference's CLI, restated as callers of the batched GPU path:
ScanDecompressCommand (scan a file for embedded streams) and BruteForceCommand (try every raw decoder on one buffer).
Thin ctypes wrappers over alz_container_scan / alz_brute_force (include/auroralz.h)."""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import check, load
from .formats import _context


class ScanHit(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("dst_off", C.c_uint64), ("dst_len", C.c_uint32), ("container", C.c_uint32)]


def scan(data, formats, big_endian=True, lz=None, max_hits=1 << 16, dst_cap=None):
    """ScanDecompressCommand.Execute (src/AuroraLib.Compression.CLI/Commands/ScanDecompressCommand.cs:12-104).
    `formats`: format classes (or alz_container ids) in identification order.
    Returns [(start, end, container, decompressed bytes)] in file order."""
    data = bytes(data)
    ids = [f if isinstance(f, int) else f.container for f in formats]
    arr = (C.c_uint32 * len(ids))(*ids)
    o = A.ContainerOptions()
    o.big_endian = 1 if big_endian else 0
    if lz is not None:
        o.lz = lz
    cap = dst_cap if dst_cap is not None else max(len(data) * 16, 1 << 20)
    lib = load()
    lib.alz_container_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    while True:
        dst_arr = np.empty(cap, dtype=np.uint8)              # (untouched memory; a ctypes buffer's .raw copies the WHOLE capacity on every access -- once per hit, below)
        dst = dst_arr.ctypes.data_as(C.c_void_p)
        hits = (ScanHit * max_hits)()
        n, used = C.c_uint32(), C.c_size_t()
        rc = lib.alz_container_scan(_context().h, arr, len(ids), C.byref(o), data, len(data), dst, cap, hits, max_hits, C.byref(n), C.byref(used))
        if rc == A.E_NOMEM and dst_cap is None and cap < (1 << 33) and n.value < max_hits:
            cap *= 4
            continue
        check(rc)
        return [(h.start, h.end, h.container, dst_arr[h.dst_off:h.dst_off + h.dst_len].tobytes()) for h in hits[:n.value]]


def brute_force(raw, expected_size):
    """BruteForceCommand.Execute (src/AuroraLib.Compression.CLI/Commands/BruteForceCommand.cs:24-94) for the raw decoders
    on the GPU path.  Returns {decoder name: (success, status, output bytes)}; success == "successfully unpacked"."""
    raw = bytes(raw)
    lib = load()
    nd = 19
    slot = max(expected_size, 1)
    dst_arr = np.empty(slot * nd, dtype=np.uint8)
    dst = dst_arr.ctypes.data_as(C.c_void_p)
    res = (A.Result * nd)()
    lib.alz_brute_force.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.alz_brute_decoder_name.restype = C.c_char_p
    check(lib.alz_brute_force(_context().h, raw, len(raw), expected_size, dst, slot, res))
    out = {}
    for i in range(nd):
        name = lib.alz_brute_decoder_name(i).decode()
        ok = res[i].status == A.ST_OK and res[i].dst_len == expected_size
        out[name] = (ok, res[i].status, dst_arr[i * slot:i * slot + res[i].dst_len].tobytes())
    return out
