"""Seeded synthetic compressed-stream batches -- bench/test input only.

Wraps csrc/synth/alz_synth.c (plain host C, built by __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libalzsynth.so")
_SRC = os.path.join(_HERE, "csrc", "synth", "alz_synth.c")
_lib = None


def build_synth(force=False):
    inc = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include")
    if force or not os.path.exists(_SO) or (os.path.exists(_SRC) and os.path.getmtime(_SO) < os.path.getmtime(_SRC)):
        from . import _lib as _libmod
        _libmod.refuse_build_after_gpu(_SO)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-I", inc, "-o", _SO, _SRC, "-lpthread", "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build_synth()
        _lib = C.CDLL(_SO)
        _lib.alz_synth_stream.restype = C.c_int64
        _lib.alz_synth_stream.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.alz_synth_batch_seeds.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return _lib


def seed_for(config, index=0):
    """seed = 0xA17A0000 + 1000*config + stream_index."""
    return 0xA17A0000 + 1000 * config + index


class Batch:
    """A packed batch: `src` (np.uint8), ctypes array of alz_stream, and per-stream target sizes."""

    def __init__(self, src, streams, targets, formats, dst_bytes):
        self.src, self.streams, self.targets, self.formats, self.dst_bytes = src, streams, targets, formats, dst_bytes
        self.n = len(streams)

    @property
    def compressed_bytes(self):
        return int(sum(s.src_len for s in self.streams)) if self.n < 4096 else int(np.frombuffer(self.streams, dtype=np.uint32).reshape(self.n, 10)[:, 4].sum())

    @property
    def decompressed_bytes(self):
        return int(np.asarray(self.targets, dtype=np.uint64).sum())


def make_batch(formats, n, target, base_seed, lz=None, nthreads=None, dst_align=256, dst_slack=0, seeds=None):
    """Generate n streams decoding to `target` bytes each (int or array).  `formats` is one alz_format or an
    array of n.  Streams are packed 16-byte aligned in `src`; outputs are laid out `dst_align`-aligned.
    Stream i is seeded with base_seed + i, or with seeds[i] (a rank's share of ONE batch: the global stream indices)."""
    lib = _load()
    nthreads = nthreads or len(os.sched_getaffinity(0)) or 1
    sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
    sdp = None if sd is None else sd.ctypes.data_as(C.c_void_p)
    fm = np.full(n, formats, dtype=np.uint32) if np.isscalar(formats) else np.ascontiguousarray(formats, dtype=np.uint32)
    tg = np.full(n, target, dtype=np.uint32) if np.isscalar(target) else np.ascontiguousarray(target, dtype=np.uint32)
    sizes = np.zeros(n, dtype=np.uint32)
    aux = (A.EncodeAux * n)()
    lzp = C.byref(lz) if lz is not None else None
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.alz_synth_batch_seeds(vp(fm), 0, lzp, base_seed, sdp, n, vp(tg), 0, None, None, vp(sizes), aux, nthreads)
    offs = np.zeros(n, dtype=np.uint64)
    al = (sizes.astype(np.uint64) + 15) & ~np.uint64(15)
    offs[1:] = np.cumsum(al)[:-1]
    total = int(offs[-1] + al[-1]) if n else 0
    src = np.zeros(total + 64, dtype=np.uint8)
    lib.alz_synth_batch_seeds(vp(fm), 0, lzp, base_seed, sdp, n, vp(tg), 0, vp(src), vp(offs), vp(sizes), aux, nthreads)
    caps = tg.astype(np.uint64) + np.uint64(dst_slack)
    dal = (caps + np.uint64(dst_align - 1)) & ~np.uint64(dst_align - 1)
    doffs = np.zeros(n, dtype=np.uint64)
    doffs[1:] = np.cumsum(dal)[:-1]
    dst_bytes = int(doffs[-1] + dal[-1]) if n else 0
    streams = (A.Stream * n)()
    rec = np.frombuffer(streams, dtype=np.dtype([("src_off", "<u8"), ("dst_off", "<u8"), ("src_len", "<u4"), ("dst_cap", "<u4"),
                                                 ("decom_len", "<u4"), ("aux0", "<u4"), ("aux1", "<u4"), ("format", "<u4")]))
    auxv = np.frombuffer(aux, dtype=np.uint32).reshape(n, 2)
    rec["src_off"], rec["dst_off"], rec["src_len"] = offs, doffs, sizes
    rec["dst_cap"], rec["decom_len"] = caps.astype(np.uint32), tg
    rec["aux0"], rec["aux1"], rec["format"] = auxv[:, 0], auxv[:, 1], fm
    return Batch(src, streams, tg, fm, dst_bytes)


def stream_records(streams):
    """numpy structured view over a ctypes alz_stream array."""
    return np.frombuffer(streams, dtype=np.dtype([("src_off", "<u8"), ("dst_off", "<u8"), ("src_len", "<u4"), ("dst_cap", "<u4"),
                                                  ("decom_len", "<u4"), ("aux0", "<u4"), ("aux1", "<u4"), ("format", "<u4")]))


def result_records(results):
    return np.frombuffer(results, dtype=np.dtype([("dst_len", "<u4"), ("src_used", "<u4"), ("status", "<i4"), ("reserved", "<u4")]))
