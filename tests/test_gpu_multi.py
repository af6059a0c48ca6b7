"""alz_decode_batch_multi: ONE batch split over several contexts (BASELINE.json configs[3]).  The driver's GPU
box has one device, so N contexts on device 0 stand in for N devices: the partitioning, the per-share packing, the host
threads and the result scatter are exactly what N GPUs run."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from auroralib.compression_amd._lib import AlzError
from auroralib.compression_amd.batch import Context, decode_batch_multi
from gpu_common import _check, pack_streams

pytestmark = pytest.mark.gpu


def _mixed(n, size, seed):
    fm = np.array([[A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_PRS_BE][i % 4] for i in range(n)], dtype=np.uint32)
    return synth.make_batch(fm, n, size, seed)


@pytest.mark.parametrize("nctx", [1, 2, 3, 8])
def test_mixed_batch_over_n_contexts_is_bit_exact(nctx):
    b = _mixed(257, 20000, synth.seed_for(4))
    o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=8)
    ctxs = [Context(0) for _ in range(nctx)]
    try:
        g_dst, g_res, part = decode_batch_multi(ctxs, b.streams, b.src, b.dst_bytes)
    finally:
        for c in ctxs:
            c.close()
    _check(b.streams, g_dst, g_res, o_dst, o_res, "multi x%d" % nctx)
    assert set(part.tolist()) == set(range(nctx))


def test_cfg4_shape_scaled_down_every_stream_hashes_like_the_oracle():
    """BASELINE.json configs[3]: mixed LZ10/LZ11/Yaz0/PRS batch sharded across 8 devices, per-format kernel dispatch,
    bit-exact check -- 4 000 x 64 KiB here (the driver's box has one GPU: eight contexts share it)."""
    b = _mixed(4000, 65536, synth.seed_for(4))
    o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=8)
    ctxs = [Context(0) for _ in range(8)]
    try:
        g_dst, g_res, part = decode_batch_multi(ctxs, b.streams, b.src, b.dst_bytes)
    finally:
        for c in ctxs:
            c.close()
    _check(b.streams, g_dst, g_res, o_dst, o_res, "cfg4")
    assert np.bincount(part, minlength=8).min() > 300


def test_ragged_and_failing_streams_keep_their_results():
    items = []
    good = synth.make_batch(A.FMT_YAZ0, 6, 3000, 99)
    recs = synth.stream_records(good.streams)
    for i in range(6):
        body = bytes(good.src[int(recs["src_off"][i]):int(recs["src_off"][i]) + int(recs["src_len"][i])])
        items.append(dict(fmt=A.FMT_YAZ0, src=body, decom_len=3000, cap=3000))
        items.append(dict(fmt=A.FMT_YAZ0, src=body[:len(body) // 2], decom_len=3000, cap=3000))       # truncated
        items.append(dict(fmt=A.FMT_YAZ0, src=body, decom_len=3000, cap=1000))                        # capacity
    items.append(dict(fmt=A.FMT_LZ10, src=b"", decom_len=0, cap=0))                                     # empty
    streams, src, dst_bytes = pack_streams(items)
    o_dst, o_res = O.decode_batch(streams, src, dst_bytes, nthreads=2)
    ctxs = [Context(0) for _ in range(3)]
    try:
        g_dst, g_res, _ = decode_batch_multi(ctxs, streams, src, dst_bytes)
    finally:
        for c in ctxs:
            c.close()
    _check(streams, g_dst, g_res, o_dst, o_res, "ragged multi")


def test_bad_arguments_are_refused():
    with Context(0) as c:
        b = _mixed(4, 1000, 5)
        with pytest.raises(AlzError):
            decode_batch_multi([c, c], b.streams, b.src, b.dst_bytes)              # a context is single-threaded
        # offsets near 2^64 must not wrap past the range check
        st = (A.Stream * 1)(A.Stream(0xFFFFFFFFFFFFFFF0, 0, 0x20, 16, 16, 0, 0, A.FMT_LZ10))
        with pytest.raises(AlzError):
            c.decode_batch(st, np.zeros(64, dtype=np.uint8), 64)
        st = (A.Stream * 1)(A.Stream(0, 0xFFFFFFFFFFFFFFF0, 4, 0x20, 16, 0, 0, A.FMT_LZ10))
        with pytest.raises(AlzError):
            c.decode_batch(st, np.zeros(64, dtype=np.uint8), 64)
        with pytest.raises(AlzError):
            decode_batch_multi([c], st, np.zeros(64, dtype=np.uint8), 64)
