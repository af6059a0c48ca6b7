"""Thin object layer over the batch half of the C ABI (context, plans, device buffers)."""
import ctypes as C

import numpy as np

from . import _abi as A
from . import _lib as _libmod
from ._lib import check, load


def _vp(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return a


def device_count():
    """alz_device_count: HIP devices visible to this process (0 without a GPU)."""
    _libmod.GPU_TOUCHED = True
    return int(load().alz_device_count())


class Context:
    """alz_ctx: one HIP device + one HIP stream."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        _libmod.GPU_TOUCHED = True
        check(self.lib.alz_create(device, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.lib.alz_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def info(self):
        name = C.create_string_buffer(256)
        cu, mem = C.c_int(), C.c_uint64()
        check(self.lib.alz_device_info(self.h, name, 256, C.byref(cu), C.byref(mem)))
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}

    def set_exact_kernels(self, on):
        """Kernel family of this context: the exact one-token-at-a-time kernels instead of the lane-parallel ones."""
        check(self.lib.alz_ctx_set_exact_kernels(self.h, 1 if on else 0))

    def set_kernel_variant(self, variant):
        """alz_ctx_set_kernel_variant: 0 the library chooses, 1 / 2 one / two wavefronts per stream where both shapes exist."""
        check(self.lib.alz_ctx_set_kernel_variant(self.h, variant))

    def big_stream(self, min_bytes=0):
        """alz_ctx_big_stream: threshold of the whole-GPU paths for ONE stream of the north-star bodies -- decode from `min_bytes` of output on
        (default 24 KiB), encode from min(min_bytes, 8 KiB) of input on; 0 keeps the current value, 0xFFFFFFFF switches both paths off.
        Returns how many streams have taken either path on this context."""
        n = C.c_uint64()
        check(self.lib.alz_ctx_big_stream(self.h, min_bytes, C.byref(n)))
        return n.value

    def release_scratch(self):
        """alz_ctx_release_scratch: return the grow-only staging / encoder scratch of the host-buffer calls to the device."""
        check(self.lib.alz_ctx_release_scratch(self.h))

    def copy_bandwidth(self, nbytes=1 << 30, iters=10):
        """Measured device-to-device copy bandwidth in GB/s (bytes read + written): the second roofline denominator."""
        v = C.c_double()
        check(self.lib.alz_measure_copy_bandwidth(self.h, nbytes, iters, C.byref(v)))
        return v.value

    def last_kernel_ms(self):
        v = C.c_float()
        check(self.lib.alz_last_kernel_ms(self.h, C.byref(v)))
        return v.value

    # ---- host-buffer decode (upload, decode on GPU, download)
    def decode_batch(self, streams, src, dst_bytes, lz=None, dst=None):
        """alz_decode_batch on host buffers.  `dst`: a caller-owned uint8 array of >= dst_bytes to decode into (default: a new one)."""
        n = len(streams)
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if dst is None:
            dst = np.zeros(max(dst_bytes, 1), dtype=np.uint8)
        elif dst.dtype != np.uint8 or not dst.flags.c_contiguous or dst.nbytes < dst_bytes:
            raise ValueError("dst must be a contiguous uint8 array of at least dst_bytes")
        res = (A.Result * n)()
        check(self.lib.alz_decode_batch(self.h, C.byref(lz) if lz is not None else None, n, _vp(src), src.nbytes, streams, _vp(dst), dst_bytes, res))
        return dst, res

    def decode(self, fmt, src, decom_len=0, cap=None, aux0=0, aux1=0, lz=None):
        src = bytes(src)
        cap = decom_len if cap is None else cap
        dst = np.empty(max(cap, 1), dtype=np.uint8)          # (untouched memory: a ctypes buffer would be written full of zeros first)
        r = A.Result()
        check(self.lib.alz_decode(self.h, fmt, C.byref(lz) if lz is not None else None, src, len(src), decom_len, aux0, aux1, _vp(dst), cap, C.byref(r)))
        return dst[:r.dst_len].tobytes(), r

    # ---- host-buffer encode
    def encode_batch(self, streams, src, dst_bytes, quality=8, lz=None, strategy=0, min_distance=0, max_window_bits=0):
        """alz_encode_batch: streams describe RAW inputs (src_*) and compressed-output capacity (dst_*)."""
        n = len(streams)
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.zeros(max(dst_bytes, 1), dtype=np.uint8)
        res = (A.Result * n)()
        aux = (A.EncodeAux * n)()
        st = A.Settings(quality, max_window_bits, strategy, min_distance)
        check(self.lib.alz_encode_batch(self.h, C.byref(lz) if lz is not None else None, C.byref(st), n, _vp(src), src.nbytes, streams,
                                        _vp(dst), dst_bytes, res, aux))
        return dst, res, aux

    def encode_batch_device(self, streams, d_src, src_bytes, d_dst, dst_bytes, quality=8, lz=None, strategy=0, min_distance=0, max_window_bits=0):
        """alz_encode_batch_device: raw buffers already in HBM at d_src, compressed streams left in HBM at d_dst (offsets of
        `streams` are relative to the two device pointers).  Returns (results, aux); last_kernel_ms() is the device time."""
        n = len(streams)
        res = (A.Result * n)()
        aux = (A.EncodeAux * n)()
        st = A.Settings(quality, max_window_bits, strategy, min_distance)
        check(self.lib.alz_encode_batch_device(self.h, C.byref(lz) if lz is not None else None, C.byref(st), n, d_src, src_bytes, streams,
                                               d_dst, dst_bytes, res, aux))
        return res, aux

    # ---- device memory
    def malloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.alz_device_malloc(self.h, nbytes, C.byref(p)))
        return p

    def free(self, p):
        check(self.lib.alz_device_free(self.h, p))

    def h2d(self, d, arr):
        arr = np.ascontiguousarray(arr)
        check(self.lib.alz_memcpy_h2d(self.h, d, _vp(arr), arr.nbytes))

    def d2h(self, d, nbytes, offset=0):
        out = np.empty(nbytes, dtype=np.uint8)
        check(self.lib.alz_memcpy_d2h(self.h, _vp(out), C.c_void_p(d.value + offset), nbytes))
        return out

    def memset(self, d, value, nbytes):
        check(self.lib.alz_memset_d(self.h, d, value, nbytes))

    def synchronize(self):
        check(self.lib.alz_synchronize(self.h))


class Plan:
    """alz_plan: descriptor table resident in HBM, grouped per format."""

    def __init__(self, ctx, streams, lz=None):
        self.ctx, self.n = ctx, len(streams)
        h = C.c_void_p()
        check(ctx.lib.alz_plan_create(ctx.h, C.byref(lz) if lz is not None else None, self.n, streams, C.byref(h)))
        self.h = h

    def execute(self, d_src, d_dst, hip_stream=None):
        check(self.ctx.lib.alz_plan_execute(self.ctx.h, self.h, d_src, d_dst, hip_stream))

    def execute_timed(self, d_src, d_dst, iters=1):
        ms = C.c_float()
        check(self.ctx.lib.alz_plan_execute_timed(self.ctx.h, self.h, d_src, d_dst, iters, C.byref(ms)))
        return ms.value

    def results(self):
        res = (A.Result * self.n)()
        check(self.ctx.lib.alz_plan_results(self.ctx.h, self.h, res))
        return res

    def close(self):
        if self.h:
            self.ctx.lib.alz_plan_destroy(self.ctx.h, self.h)
            self.h = None


class MultiPlan:
    """alz_multi_plan: ONE device-resident batch over several contexts (one per GPU) -- the batch partitioned by the library (or by `part_of`),
    one plan per context, no host staging, no collective.  A stream's offsets are relative to the device buffers of ITS context."""

    def __init__(self, ctxs, streams, lz=None, part_of=None):
        self.ctxs, self.n = list(ctxs), len(streams)
        self.lib = self.ctxs[0].lib
        arr = (C.c_void_p * len(self.ctxs))(*[c.h for c in self.ctxs])
        part = np.zeros(max(self.n, 1), dtype=np.uint32)
        given = None
        if part_of is not None:
            given = np.ascontiguousarray(part_of, dtype=np.uint32)
        h = C.c_void_p()
        check(self.lib.alz_plan_create_multi(arr, len(self.ctxs), C.byref(lz) if lz is not None else None, self.n, streams,
                                             _vp(given) if given is not None else None, C.byref(h), _vp(part)))
        self.h, self.part_of = h, part[:self.n]

    def execute(self, d_srcs, d_dsts):
        k = len(self.ctxs)
        a = (C.c_void_p * k)(*[p.value if hasattr(p, "value") else p for p in d_srcs])
        b = (C.c_void_p * k)(*[p.value if hasattr(p, "value") else p for p in d_dsts])
        check(self.lib.alz_plan_execute_multi(self.h, a, b))

    def results(self):
        res = (A.Result * self.n)()
        check(self.lib.alz_plan_results_multi(self.h, res))
        return res

    def close(self):
        if self.h:
            self.lib.alz_plan_destroy_multi(self.h)
            self.h = None


def partition_batch(streams, n_parts):
    """alz_partition_batch: greedy LPT partition of a batch (host code, no GPU).  Returns (part_of np.uint32[n], cost np.uint64[n_parts])."""
    lib = load()
    n = len(streams)
    part = np.zeros(max(n, 1), dtype=np.uint32)
    cost = np.zeros(n_parts, dtype=np.uint64)
    check(lib.alz_partition_batch(n, streams, n_parts, _vp(part), _vp(cost)))
    return part[:n], cost


def decode_batch_multi(ctxs, streams, src, dst_bytes, lz=None):
    """alz_decode_batch_multi: ONE batch over several contexts (one per GPU; host threads inside the library)."""
    lib = load()
    n = len(streams)
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros(max(dst_bytes, 1), dtype=np.uint8)
    res = (A.Result * n)()
    part = np.zeros(max(n, 1), dtype=np.uint32)
    hs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    check(lib.alz_decode_batch_multi(hs, len(ctxs), C.byref(lz) if lz is not None else None, n, _vp(src), src.nbytes, streams,
                                     _vp(dst), dst_bytes, res, _vp(part)))
    return dst, res, part[:n]


def encode_batch_multi(ctxs, streams, src, dst_bytes, quality=8, lz=None, strategy=0, min_distance=0, max_window_bits=0):
    """alz_encode_batch_multi: ONE batch of raw buffers over several contexts (one per GPU; host threads inside the library)."""
    lib = load()
    n = len(streams)
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros(max(dst_bytes, 1), dtype=np.uint8)
    res = (A.Result * n)()
    aux = (A.EncodeAux * n)()
    part = np.zeros(max(n, 1), dtype=np.uint32)
    st = A.Settings(quality, max_window_bits, strategy, min_distance)
    hs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    check(lib.alz_encode_batch_multi(hs, len(ctxs), C.byref(lz) if lz is not None else None, C.byref(st), n, _vp(src), src.nbytes, streams,
                                     _vp(dst), dst_bytes, res, aux, _vp(part)))
    return dst, res, aux, part[:n]
