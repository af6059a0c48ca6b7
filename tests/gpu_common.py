"""Helpers shared by the -m gpu parity tests: run a batch through the C ABI and through the oracle, compare bit-exactly."""
import numpy as np

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        from auroralib.compression_amd.batch import Context
        _ctx = Context(0)
    return _ctx


def compare_batch(streams, src, dst_bytes, lz=None, what=""):
    """Decode on the GPU (host-buffer ABI) and with the oracle; every result field and every output byte must match."""
    n = len(streams)
    o_dst, o_res = O.decode_batch(streams, src, dst_bytes, lz=lz, nthreads=8)
    # both GPU kernel families: the lane-parallel kernels (default dispatch) and the exact serial kernels
    for serial in (1, 0):
        ctx().set_exact_kernels(serial)
        try:
            g_dst, g_res = ctx().decode_batch(streams, src, dst_bytes, lz=lz)
        finally:
            ctx().set_exact_kernels(0)
        gr, g_dst = _check(streams, g_dst, g_res, o_dst, o_res, what + (" [serial kernels]" if serial else " [fast kernels]"))
    return gr, g_dst


def _check(streams, g_dst, g_res, o_dst, o_res, what):
    n = len(streams)
    gr, orr, sr = synth.result_records(g_res), synth.result_records(o_res), synth.stream_records(streams)
    bad = np.nonzero((gr["status"] != orr["status"]) | (gr["dst_len"] != orr["dst_len"]))[0]
    assert bad.size == 0, "%s: stream %d: gpu(status=%d,len=%d) oracle(status=%d,len=%d)" % (
        what, bad[0], gr["status"][bad[0]], gr["dst_len"][bad[0]], orr["status"][bad[0]], orr["dst_len"][bad[0]])
    ok = orr["status"] != A.ST_OUTPUT_CAPACITY                 # src_used is defined for every other status (include/auroralz.h)
    badu = np.nonzero(ok & (gr["src_used"] != orr["src_used"]))[0]
    assert badu.size == 0, "%s: stream %d src_used gpu=%d oracle=%d" % (what, badu[0] if badu.size else -1, gr["src_used"][badu[0]], orr["src_used"][badu[0]])
    for i in range(n):
        a, ln = int(sr["dst_off"][i]), int(orr["dst_len"][i])
        if not np.array_equal(g_dst[a:a + ln], o_dst[a:a + ln]):
            d = np.nonzero(g_dst[a:a + ln] != o_dst[a:a + ln])[0]
            raise AssertionError("%s: stream %d differs at byte %d of %d (gpu=%d oracle=%d), %d bytes differ" % (
                what, i, d[0], ln, g_dst[a + d[0]], o_dst[a + d[0]], d.size))
    return gr, g_dst


def pack_streams(items, dst_align=16, dst_slack=0):
    """items: list of dicts(fmt, src(bytes), decom_len, cap(optional), aux0, aux1). Returns (streams, src array, dst_bytes)."""
    n = len(items)
    streams = (A.Stream * n)()
    chunks, so, do = [], 0, 0
    for i, it in enumerate(items):
        b = bytes(it["src"])
        cap = it.get("cap", it.get("decom_len", 0))
        streams[i] = A.Stream(so, do, len(b), cap, it.get("decom_len", 0), it.get("aux0", 0), it.get("aux1", 0), it["fmt"])
        pad = (-len(b)) % 16 + it.get("src_misalign", 0)
        chunks.append(b + bytes(pad))
        so += len(b) + pad
        do += (cap + dst_slack + dst_align - 1) // dst_align * dst_align + it.get("dst_misalign", 0)
    src = np.frombuffer(b"".join(chunks) + bytes(64), dtype=np.uint8).copy()
    return streams, src, do + 64


def hip_streams(k):
    """k HIP streams of the CALLER's (hipStreamCreateWithFlags, non-blocking) as ctypes void pointers, plus a function that destroys them: what a
    host program hands alz_plan_execute as `hip_stream`.  The runtime is the one libauroralz.so is bound to (already loaded by its soname)."""
    import ctypes as C
    ctx()                                                    # (the library, and with it the runtime, is loaded)
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    out = []
    for _ in range(k):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        out.append(s)

    def sync():
        for s in out:
            assert hip.hipStreamSynchronize(s) == 0

    def destroy():
        for s in out:
            hip.hipStreamDestroy(s)
    return out, sync, destroy
