"""alz_decode_batch_multi: ONE batch split over several contexts (BASELINE.json configs[3]).  Contexts are dealt
round-robin over the devices alz_device_count() reports: on a box with several GPUs context i runs on device i mod N (distinct
devices, hipSetDevice per host thread); on the one-GPU box the driver's tests run on, N contexts on device 0 stand in for N
devices -- the partitioning, the per-share packing, the host threads and the result scatter are exactly what N GPUs run."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from auroralib.compression_amd._lib import AlzError
from auroralib.compression_amd.batch import Context, decode_batch_multi, device_count
from gpu_common import _check, pack_streams

pytestmark = pytest.mark.gpu


def _contexts(n):
    """n contexts over the devices of this box: context i on device i mod (number of devices)."""
    nd = max(1, device_count())
    return [Context(i % nd) for i in range(n)]


def _mixed(n, size, seed):
    fm = np.array([[A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_PRS_BE][i % 4] for i in range(n)], dtype=np.uint32)
    return synth.make_batch(fm, n, size, seed)


@pytest.mark.parametrize("nctx", [1, 2, 3, 8])
def test_mixed_batch_over_n_contexts_is_bit_exact(nctx):
    b = _mixed(257, 20000, synth.seed_for(4))
    o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=8)
    ctxs = _contexts(nctx)
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
    ctxs = _contexts(8)
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
    ctxs = _contexts(3)
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


# ---------------------------------------------------------------------------------------------------------------------
# The compression configuration on N devices (BASELINE.json configs[4]: "LZSS compression ... 1 -> 8 GPUs scaling"): the same batch
# through alz_encode_batch_multi over N contexts and through alz_encode_batch on one must write the same bytes -- which
# tests/test_gpu_encode.py in turn compares with the oracle's restatement of the managed encoder.
def _raw_batch(n, fmts, seed):
    rng = np.random.default_rng(seed)
    b = synth.make_batch(A.FMT_LZSS, n, 40000, synth.seed_for(5))
    raw, res = Context(0).decode_batch(b.streams, b.src, b.dst_bytes)
    recs = synth.stream_records(b.streams)
    streams = (A.Stream * n)()
    do = 0
    for i in range(n):
        ln = int(rng.integers(1, 40000)) if i % 5 else 40000          # ragged: the partitioner has something to balance
        cap = ln + ln // 4 + 64
        streams[i] = A.Stream(int(recs["dst_off"][i]), do, ln, cap, 0, 0, 0, fmts[i % len(fmts)])
        do += (cap + 255) // 256 * 256
    return streams, raw, do + 64


@pytest.mark.parametrize("nctx", [1, 2, 3, 8])
@pytest.mark.parametrize("quality", [0, 8])
def test_encode_batch_over_n_contexts_writes_the_bytes_of_one(nctx, quality):
    from auroralib.compression_amd.batch import encode_batch_multi
    streams, raw, dst_bytes = _raw_batch(203, [A.FMT_LZSS, A.FMT_LZ10, A.FMT_YAZ0, A.FMT_YAY0, A.FMT_PRS_BE, A.FMT_LZ4_BLOCK], 11)
    one = Context(0)
    ctxs = _contexts(nctx)
    try:
        d1, r1, a1 = one.encode_batch(streams, raw, dst_bytes, quality=quality)
        dn, rn, an, part = encode_batch_multi(ctxs, streams, raw, dst_bytes, quality=quality)
    finally:
        one.close()
        for c in ctxs:
            c.close()
    assert set(part.tolist()) == set(range(nctx))
    for i in range(len(streams)):
        assert (rn[i].status, rn[i].dst_len) == (r1[i].status, r1[i].dst_len) and r1[i].status == A.ST_OK, i
        assert (an[i].aux0, an[i].aux1) == (a1[i].aux0, a1[i].aux1), i
        a = streams[i].dst_off
        assert np.array_equal(dn[a:a + rn[i].dst_len], d1[a:a + r1[i].dst_len]), i
    # and one of them against the oracle's encoder, so that "the same" is "the managed encoder's"
    k = 7
    want, _ = O.encode_stream(streams[k].format, bytes(raw[streams[k].src_off:streams[k].src_off + streams[k].src_len]), quality=quality)
    assert bytes(dn[streams[k].dst_off:streams[k].dst_off + rn[k].dst_len]) == want


def test_device_resident_encode_leaves_the_same_streams_in_hbm():
    """alz_encode_batch_device: raw buffers already in HBM, compressed streams left in HBM -- the entry bench.py times for the
    compression configuration.  Same bytes as the host-buffer call."""
    streams, raw, dst_bytes = _raw_batch(64, [A.FMT_LZSS], 3)
    c = Context(0)
    try:
        d1, r1, a1 = c.encode_batch(streams, raw, dst_bytes, quality=8)
        d_src, d_dst = c.malloc(raw.nbytes + 64), c.malloc(dst_bytes + 64)
        c.h2d(d_src, raw)
        c.memset(d_dst, 0, dst_bytes)
        r2, a2 = c.encode_batch_device(streams, d_src, raw.nbytes, d_dst, dst_bytes, quality=8)
        assert c.last_kernel_ms() > 0
        out = c.d2h(d_dst, dst_bytes)
        c.free(d_src); c.free(d_dst)
    finally:
        c.close()
    for i in range(len(streams)):
        assert (r2[i].status, r2[i].dst_len) == (r1[i].status, r1[i].dst_len), i
        a = streams[i].dst_off
        assert np.array_equal(out[a:a + r2[i].dst_len], d1[a:a + r1[i].dst_len]), i


# ---------------------------------------------------------------------------------------------------------------------
# Every device of the box, not only device 0 (round 3's verdict: hipSetDevice(c->device) with device != 0 had never executed).
def _all_devices():
    return list(range(max(1, device_count())))


@pytest.mark.parametrize("dev", range(8))
def test_context_on_device_n_decodes_and_encodes_like_the_oracle(dev):
    """One context on device `dev`: host-buffer decode (mixed batch), device-resident plan, host-buffer encode -- against the oracle.
    Devices the box does not have are SKIPPED (never replaced by device 0)."""
    if dev >= device_count():
        pytest.skip("this box has %d HIP device(s)" % device_count())
    from auroralib.compression_amd.batch import Plan
    b = _mixed(131, 30000, synth.seed_for(4, dev))
    o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=8)
    with Context(dev) as c:
        assert "gfx950" in c.info()["name"]
        g_dst, g_res = c.decode_batch(b.streams, b.src, b.dst_bytes)
        _check(b.streams, g_dst, g_res, o_dst, o_res, "device %d host buffers" % dev)
        d_src, d_dst = c.malloc(b.src.nbytes), c.malloc(b.dst_bytes)
        try:
            c.h2d(d_src, b.src); c.memset(d_dst, 0, b.dst_bytes)
            pl = Plan(c, b.streams)
            pl.execute(d_src, d_dst)
            res = pl.results(); pl.close()
            _check(b.streams, c.d2h(d_dst, b.dst_bytes), res, o_dst, o_res, "device %d plan" % dev)
        finally:
            c.free(d_src); c.free(d_dst)
        raw = bytes(o_dst[:30000])
        st = (A.Stream * 1)(A.Stream(0, 0, len(raw), len(raw) * 2, 0, 0, 0, A.FMT_YAZ0))
        e_dst, e_res, _ = c.encode_batch(st, np.frombuffer(raw + bytes(64), dtype=np.uint8), len(raw) * 2 + 64, quality=8)
        assert e_res[0].status == A.ST_OK and bytes(e_dst[:e_res[0].dst_len]) == O.encode_stream(A.FMT_YAZ0, raw, quality=8)[0]


def test_multi_uses_distinct_devices_when_the_box_has_them():
    """With >= 2 GPUs: one context per device, a batch decoded and a batch encoded over all of them, every share on its own device."""
    nd = device_count()
    if nd < 2:
        pytest.skip("needs >= 2 HIP devices (this box has %d)" % nd)
    from auroralib.compression_amd.batch import encode_batch_multi
    ctxs = [Context(d) for d in range(nd)]
    try:
        b = _mixed(64 * nd + 3, 65536, synth.seed_for(4, 77))
        o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=8)
        g_dst, g_res, part = decode_batch_multi(ctxs, b.streams, b.src, b.dst_bytes)
        _check(b.streams, g_dst, g_res, o_dst, o_res, "multi over %d devices" % nd)
        assert set(part.tolist()) == set(range(nd))
        streams, raw, dst_bytes = _raw_batch(40 * nd, [A.FMT_LZSS, A.FMT_YAZ0, A.FMT_LZ4_BLOCK], 13)
        dn, rn, an, part = encode_batch_multi(ctxs, streams, raw, dst_bytes, quality=8)
        assert set(part.tolist()) == set(range(nd))
        for i in range(0, len(streams), 7):
            want, _ = O.encode_stream(streams[i].format, bytes(raw[streams[i].src_off:streams[i].src_off + streams[i].src_len]), quality=8)
            assert rn[i].status == A.ST_OK and bytes(dn[streams[i].dst_off:streams[i].dst_off + rn[i].dst_len]) == want, i
    finally:
        for c in ctxs:
            c.close()


# ---------------------------------------------------------------------------------------------------------------------
# The device-resident path over several contexts (alz_plan_create_multi / _execute_multi / _results_multi): no host staging.
@pytest.mark.parametrize("given_partition", [False, True])
def test_multi_plan_device_resident_matches_the_oracle(given_partition):
    """ONE mixed batch, partitioned over 2 (or device_count()) contexts -- on distinct devices where the box has them, otherwise two contexts
    of device 0 --, every context's share decoded from ITS device buffers, results gathered in batch order: bit-exact against the oracle."""
    from auroralib.compression_amd.batch import MultiPlan, partition_batch
    nd = device_count()
    k = nd if nd >= 2 else 2
    ctxs = [Context(q if nd >= 2 else 0) for q in range(k)]
    try:
        b = _mixed(97 * k + 5, 40000, synth.seed_for(4, 91))
        o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=8)
        part_of = None
        if given_partition:
            part_of = (np.arange(len(b.streams)) * 7 % k).astype(np.uint32)
        mp = MultiPlan(ctxs, b.streams, part_of=part_of)
        if given_partition:
            assert np.array_equal(mp.part_of, part_of)
        else:
            assert np.array_equal(mp.part_of, partition_batch(b.streams, k)[0]) and set(mp.part_of.tolist()) == set(range(k))
        # every device gets the batch's layout with ONLY its own share's payload in it (the rest stays 0xEE): a stream decoded on the wrong
        # device, or from the wrong buffer, cannot come out right
        d_src, d_dst = [], []
        recs = synth.stream_records(b.streams)
        for q, c in enumerate(ctxs):
            mine = np.nonzero(mp.part_of == q)[0]
            host = np.full(b.src.nbytes, 0xEE, dtype=np.uint8)
            for i in mine:
                a, n = int(recs["src_off"][i]), int(recs["src_len"][i])
                host[a:a + n] = b.src[a:a + n]
            s, d = c.malloc(b.src.nbytes), c.malloc(b.dst_bytes)
            c.h2d(s, host); c.memset(d, 0, b.dst_bytes)
            d_src.append(s); d_dst.append(d)
        try:
            mp.execute(d_src, d_dst)
            res = mp.results()
            out = np.zeros(b.dst_bytes, dtype=np.uint8)
            for q, c in enumerate(ctxs):
                got = c.d2h(d_dst[q], b.dst_bytes)
                for i in np.nonzero(mp.part_of == q)[0]:
                    a, n = int(recs["dst_off"][i]), int(res[i].dst_len)
                    out[a:a + n] = got[a:a + n]
                # ... and nothing of another context's share was written here
                for i in np.nonzero(mp.part_of != q)[0][:50]:
                    a, n = int(recs["dst_off"][i]), int(recs["dst_cap"][i])
                    assert not got[a:a + n].any(), (q, i)
            _check(b.streams, out, res, o_dst, o_res, "multi plan over %d contexts" % k)
        finally:
            mp.close()
            for q, c in enumerate(ctxs):
                c.free(d_src[q]); c.free(d_dst[q])
    finally:
        for c in ctxs:
            c.close()


def test_multi_plan_rejects_bad_arguments():
    from auroralib.compression_amd.batch import MultiPlan
    from auroralib.compression_amd._lib import AlzError
    c = Context(0)
    try:
        b = _mixed(8, 4096, synth.seed_for(4, 5))
        with pytest.raises(AlzError):
            MultiPlan([c, c], b.streams)                                         # a context listed twice
        with pytest.raises(AlzError):
            MultiPlan([c], b.streams, part_of=np.full(len(b.streams), 3, dtype=np.uint32))   # a part that does not exist
    finally:
        c.close()
