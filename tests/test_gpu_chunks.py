"""-m gpu: the flag-byte family as a WORK QUEUE of chunks (40 KiB of output each; PRS 80 KiB) (alz_decode_fastq_kernel; include/auroralz.h, alz_ctx_set_kernel_variant 3).
A plan created in variant 3 decodes through the queue whatever its size: every case below goes through tests/test_gpu_canary.py's checker --
status / dst_len / src_used against the oracle, the WHOLE canary-filled destination buffer byte for byte -- with streams long enough to be cut:
chunk limits at multiples of the chunk size, streams that end in their first / a middle / their last chunk (errors, truncation, capacity), matches longer
than a chunk (the state is passed through), three-cursor formats, every LZSS window up to 4 KiB, uneven batches (a workgroup waits for another's
chunk), more items than the GPU holds wavefronts (the automatic choice of variant 0; the full-size batches of test_gpu_fullsize.py take it in their device-resident
leg, `_resident_plan_and_compare` -- their host-buffer leg never does: alz_decode_batch plans without a queue), and ONE queue plan executed on two streams of the
caller's into two pairs of buffers (its executes are ordered by an event: the queue heads, flags and slots are the plan's)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from gpu_common import ctx, pack_streams
from test_gpu_canary import canary_decode

pytestmark = pytest.mark.gpu
QUEUE = [A.FMT_LZSS, A.FMT_LZ10, A.FMT_LZ11, A.FMT_LZ40, A.FMT_CLZ0, A.FMT_YAZ0, A.FMT_YAY0, A.FMT_MIO0, A.FMT_PRS_BE, A.FMT_PRS_LE]   # (PRS: the two-wavefront kernel, 80 KiB chunks)
SEED = int(os.environ.get("ALZ_FUZZ_SEED", "1234"))


@pytest.mark.parametrize("fmt", QUEUE)
def test_queue_synthetic_sizes_around_the_chunk_limits(fmt):
    # (chunks of 40 KiB -- PRS 80 KiB --; the powers of two of the first build's 32 / 64 KiB chunks stay in the list)
    sizes = np.array([1, 4095, 32767, 32768, 32769, 40959, 40960, 40961, 65535, 65536, 65537, 81919, 81920, 81921, 100000, 122880, 131071, 131072, 131073, 163839, 163840, 163841, 196608, 200001, 245760, 245761,
                      262144, 300001, 327680, 524288, 1 << 20, 70001, 3, 262143] * 2, dtype=np.uint32)
    b = synth.make_batch(fmt, len(sizes), sizes, synth.seed_for(70 + fmt), dst_align=1)
    canary_decode(b.streams, b.src, b.dst_bytes, what="queue sizes " + A.FORMAT_NAMES[fmt], queue=True)


@pytest.mark.parametrize("fmt", QUEUE)
def test_queue_streams_that_end_early_or_late(fmt):
    """Valid 300 000-byte streams with the input cut at many places, the capacity below the size, the declared size above / below the truth:
    the stream ends with an error status in whichever chunk that falls into, and its later chunks must do nothing."""
    rng = np.random.default_rng(SEED + fmt)
    b = synth.make_batch(fmt, 40, 300000, synth.seed_for(80 + fmt))
    rec = synth.stream_records(b.streams)
    items = []
    for i in range(40):
        comp = bytes(b.src[int(rec["src_off"][i]):int(rec["src_off"][i]) + int(rec["src_len"][i])])
        a0, a1 = int(rec["aux0"][i]), int(rec["aux1"][i])
        kind = i % 5
        if kind == 0:
            items.append(dict(fmt=fmt, src=comp, decom_len=300000, aux0=a0, aux1=a1))
        elif kind == 1:                                         # truncated input (anywhere: first, middle, last chunk)
            items.append(dict(fmt=fmt, src=comp[:int(rng.integers(1, len(comp)))], decom_len=300000, aux0=a0, aux1=a1))
        elif kind == 2:                                         # capacity below the size (E5)
            items.append(dict(fmt=fmt, src=comp, decom_len=300000, cap=int(rng.integers(1, 300000)), aux0=a0, aux1=a1))
        elif kind == 3:                                         # declared size below the truth (the last match may overshoot: E4)
            items.append(dict(fmt=fmt, src=comp, decom_len=int(rng.integers(1, 300000)), cap=300000, aux0=a0, aux1=a1))
        else:                                                   # declared size above the truth: the input runs out
            items.append(dict(fmt=fmt, src=comp, decom_len=400000, cap=400000, aux0=a0, aux1=a1))
    streams, src, dst_bytes = pack_streams(items, dst_slack=32)
    canary_decode(streams, src, dst_bytes, what="queue early/late " + A.FORMAT_NAMES[fmt], queue=True)


@pytest.mark.parametrize("fmt", QUEUE)
def test_queue_real_data_and_garbage(fmt, test_bmp):
    """1 000 KiB of Test.bmp at three qualities (runs at distance 4, long matches), a megabyte of zeros (LZ11 / LZ40: matches longer than a chunk --
    whole chunks are passed through), and mutated copies (a flipped byte per 4 KiB: bad tokens and wrong sizes somewhere in the middle)."""
    rng = np.random.default_rng(SEED * 3 + fmt)
    items = []
    raw = bytes(test_bmp[:1024000])
    for q in (0, 8, 15):
        comp, aux = O.encode_stream(fmt, raw, quality=q)
        items.append(dict(fmt=fmt, src=comp, decom_len=len(raw), aux0=aux.aux0, aux1=aux.aux1))
        m = bytearray(comp)
        for k in range(0, len(m), 4096):
            m[k + int(rng.integers(0, min(4096, len(m) - k)))] ^= 1 << int(rng.integers(0, 8))
        items.append(dict(fmt=fmt, src=bytes(m), decom_len=len(raw), aux0=aux.aux0, aux1=aux.aux1))
    zeros = bytes(1 << 20)
    comp, aux = O.encode_stream(fmt, zeros, quality=15)
    items.append(dict(fmt=fmt, src=comp, decom_len=len(zeros), aux0=aux.aux0, aux1=aux.aux1))
    items.append(dict(fmt=fmt, src=comp, decom_len=len(zeros), cap=700001, aux0=aux.aux0, aux1=aux.aux1))
    streams, src, dst_bytes = pack_streams(items, dst_slack=32)
    canary_decode(streams, src, dst_bytes, what="queue real " + A.FORMAT_NAMES[fmt], queue=True)


@pytest.mark.parametrize("bits", [(8, 4, 2), (10, 6, 2), (11, 5, 3), (12, 4, 2)])
def test_queue_lzss_windows(bits):
    lz = A.LzProperties.from_bits(*bits)
    b = synth.make_batch(A.FMT_LZSS, 24, np.array([200000, 65536, 131073, 70000] * 6, dtype=np.uint32), synth.seed_for(90 + bits[0]), lz=lz)
    canary_decode(b.streams, b.src, b.dst_bytes, lz=lz, what="queue lzss %s" % (bits,), queue=True)


def test_queue_uneven_batch_more_items_than_wavefronts():
    """One batch of 3 000 short and 300 long Yaz0 / LZ10 / MIO0 streams: ~25 000 items (one workgroup each) on ~6 400 wavefront places, chunks of one stream
    decoded on different CUs one after the other, workgroups waiting for each other's hand-overs."""
    fm = np.array([(A.FMT_YAZ0, A.FMT_LZ10, A.FMT_MIO0, A.FMT_PRS_BE)[i % 4] for i in range(3300)], dtype=np.uint32)
    sizes = np.array([1 << 20 if i % 11 == 0 else 70000 + 517 * (i % 97) for i in range(3300)], dtype=np.uint32)
    b = synth.make_batch(fm, len(fm), sizes, synth.seed_for(99))
    canary_decode(b.streams, b.src, b.dst_bytes, what="queue uneven", queue=True)


def test_queue_is_what_a_big_plan_runs_by_itself_and_small_plans_do_not():
    """Variant 0: a plan of more streams than the GPU holds wavefronts takes the queue, a small one does not (alz_debug_plan_queue_items);
    both decode like the oracle."""
    import ctypes as C
    from auroralib.compression_amd.batch import Plan
    from gpu_common import _check
    c = ctx()
    c.lib.alz_debug_plan_queue_items.argtypes = [C.c_void_p]
    ch = c.lib.alz_debug_chunk_bytes()
    for n, size, queued in ((9000, 140000, True), (4500, 140000, True), (500, 140000, False), (9000, ch - 5, False)):
        b = synth.make_batch(A.FMT_YAZ0, n, size, synth.seed_for(98))
        o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=8)
        d_src, d_dst = c.malloc(b.src.nbytes), c.malloc(b.dst_bytes)
        try:
            c.h2d(d_src, b.src); c.memset(d_dst, 0, b.dst_bytes)
            pl = Plan(c, b.streams)
            items = c.lib.alz_debug_plan_queue_items(pl.h)
            assert (items > 0) == queued, (n, size, items)
            if queued:
                assert items == n * ((size + ch - 1) // ch)
            pl.execute(d_src, d_dst)
            pl.execute(d_src, d_dst)                                   # (a second launch of the same plan: every polled word is zeroed again)
            res = pl.results(); pl.close()
            _check(b.streams, c.d2h(d_dst, b.dst_bytes), res, o_dst, o_res, "%d streams of %d bytes, variant 0" % (n, size))
        finally:
            c.free(d_src); c.free(d_dst)


@pytest.mark.parametrize("fmt,n,size", [(A.FMT_YAZ0, 8000, 100000), (A.FMT_MIO0, 7000, 90000), (A.FMT_PRS_BE, 3400, 170000), (A.FMT_YAZ0, 64, 300000)])
def test_one_queue_plan_on_two_caller_streams_and_two_buffer_pairs(fmt, n, size):
    """include/auroralz.h: "One plan may be executed again while an earlier execute is still in flight, also on another stream and into other buffers."  A queue plan owns
    mutable device state (queue head, hand-over flags, slots with LDS windows and cursors), so alz_plan_execute orders its executes by an event.  Here ONE plan is executed
    four times back to back, alternating between two streams of the caller's and between two (source, destination) pairs whose CONTENTS differ: if the executes overlapped,
    an item of one would pick up the other's hand-over window and cursors.  Both destinations must come out byte-identical to the oracle's decode of their own source
    (whole canary-filled buffers), and the plan's result table must be the last execute's.  The two sources share one descriptor table: every stream sits in a slot of
    fixed pitch with src_len = the pitch (these decoders stop at the declared size; src_used says what each consumed)."""
    import ctypes as C
    from auroralib.compression_amd.batch import Plan
    from gpu_common import hip_streams
    c = ctx()
    c.lib.alz_debug_plan_queue_items.argtypes = [C.c_void_p]
    c.lib.alz_debug_chunk_repeats.restype = C.c_uint64; c.lib.alz_debug_chunk_repeats.argtypes = [C.c_void_p]
    pair = [synth.make_batch(fmt, n, size, synth.seed_for(60 + k, fmt)) for k in range(2)]
    recs = [synth.stream_records(b.streams) for b in pair]
    three = fmt in (A.FMT_YAY0, A.FMT_MIO0)
    pitch = (int(max(r["src_len"].max() for r in recs)) + 15) // 16 * 16
    dpitch = (size + 255) // 256 * 256
    streams = (A.Stream * n)()
    t = synth.stream_records(streams)
    t["src_off"], t["src_len"] = np.arange(n, dtype=np.uint64) * np.uint64(pitch), pitch
    t["dst_off"], t["dst_cap"], t["decom_len"], t["format"] = np.arange(n, dtype=np.uint64) * np.uint64(dpitch), size, size, fmt
    srcs = []
    for b, r in zip(pair, recs):
        s = np.zeros(n * pitch + 64, dtype=np.uint8)
        for i in range(n):
            a, ln = int(r["src_off"][i]), int(r["src_len"][i])
            s[i * pitch:i * pitch + ln] = b.src[a:a + ln]
        srcs.append(s)
    if three:
        # (the section offsets live in the descriptor: both sources must agree on them, or the table cannot be shared)
        if not (np.array_equal(recs[0]["aux0"], recs[1]["aux0"]) and np.array_equal(recs[0]["aux1"], recs[1]["aux1"])):
            # same token structure is not guaranteed across seeds: decode the SAME streams from two buffers, the second with its bytes behind every stream's end replaced
            srcs[1] = srcs[0].copy()
            for i in range(n):
                ln = int(recs[0]["src_len"][i])
                srcs[1][i * pitch + ln:(i + 1) * pitch] = 0x5A
        t["aux0"], t["aux1"] = recs[0]["aux0"], recs[0]["aux1"]
    dst_bytes = n * dpitch + 64
    GUARD, CANARY = 4096, 0xA5
    total = GUARD + dst_bytes + GUARD
    want = []
    for s in srcs:
        o_dst, o_res = O.decode_batch(streams, s, dst_bytes, nthreads=8)
        orr = synth.result_records(o_res).copy()
        assert (orr["status"] == 0).all() and (orr["dst_len"] == size).all()
        w = np.full(total, CANARY, dtype=np.uint8)
        for i in range(n):
            w[GUARD + i * dpitch:GUARD + i * dpitch + size] = o_dst[i * dpitch:i * dpitch + size]
        want.append((w, orr))
    if not three:
        assert not np.array_equal(want[0][0], want[1][0])
    hs, sync, destroy = hip_streams(2)
    d_src = [c.malloc(s.nbytes) for s in srcs]
    d_dst = [c.malloc(total) for _ in srcs]
    plan = None
    try:
        for d, s in zip(d_src, srcs):
            c.h2d(d, s)
        plan = Plan(c, streams)
        forced = c.lib.alz_debug_plan_queue_items(plan.h) == 0
        if forced:                                                # (a batch that does not take the queue by itself -- small, or PRS below what the GPU holds -- is made to: there EVERY chunk waits for the one before)
            plan.close()
            c.set_kernel_variant(3)
            try:
                plan = Plan(c, streams)
            finally:
                c.set_kernel_variant(0)
        assert forced == (n < 3500)
        assert c.lib.alz_debug_plan_queue_items(plan.h) > n
        before = c.lib.alz_debug_chunk_repeats(c.h)
        try:
            for d in d_dst:
                c.memset(d, CANARY, total)
            c.synchronize()
            use = [None, hs[1]] if three else hs                       # (MIO0: the context's own stream and one of the caller's -- the own stream's event is recorded only when another stream needs it)
            for k in (0, 1, 0, 1):                                    # back to back, no synchronisation in between
                plan.execute(d_src[k], C.c_void_p(d_dst[k].value + GUARD), use[k])
            gr = synth.result_records(plan.results()).copy()          # (waits for the LAST execute, on hs[1] ...)
            sync()                                                    # (... and the caller waits for its own streams)
        finally:
            c.set_kernel_variant(0)
        if not os.environ.get("ALZ_EXPECT_SPIN_TIMEOUTS"):        # (tools/variants.sh ... spin0: a library whose every real wait "runs out" -- the repair path must give the same bytes)
            assert c.lib.alz_debug_chunk_repeats(c.h) == before
        for f in ("status", "dst_len", "src_used"):
            assert np.array_equal(gr[f], want[1][1][f]), f
        for k in (0, 1):
            g = c.d2h(d_dst[k], total)
            if not np.array_equal(g, want[k][0]):
                at = int(np.flatnonzero(g != want[k][0])[0]) - GUARD
                raise AssertionError("buffer pair %d differs at destination offset %d (stream %d, byte %d): %d bytes differ" % (k, at, at // dpitch, at % dpitch, int((g != want[k][0]).sum())))
    finally:
        if plan is not None:
            plan.close()
        for d in d_src + d_dst:
            c.free(d)
        destroy()
