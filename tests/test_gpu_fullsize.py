"""-m gpu: BASELINE.json's configurations at their FULL single-GPU sizes.  The oracle decodes the same batch on all host
cores (a fraction of a second per GiB), so the decode cases are still byte-for-byte; the compression case (cfg5; configurations numbered from 1 as in DESIGN.md), whose
oracle would take minutes at this size, is checked through size-independent properties: encode -> decode round trip of
the whole batch on the GPU, every status OK, and bit-identity with the oracle on a sample of the streams."""
import os

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from gpu_common import ctx

pytestmark = pytest.mark.gpu
CORES = os.cpu_count() or 8


def _decode_and_compare(fmt, n, size, seed, queue_formats=None):
    """Host-buffer entry point AND (queue_formats is not None) the device-resident plan -- the path bench.py times -- against the oracle.
    queue_formats: the formats expected to run as a work queue of chunks in the resident plan (alz_decode_fastq_kernel; () = none must)."""
    b = synth.make_batch(fmt, n, size, seed)
    o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=CORES)
    g_dst, g_res = ctx().decode_batch(b.streams, b.src, b.dst_bytes)
    gr, orr = synth.result_records(g_res), synth.result_records(o_res)
    for f in ("status", "dst_len", "src_used"):
        assert np.array_equal(gr[f], orr[f]), f
    assert (gr["status"] == 0).all() and (gr["dst_len"] == size).all()
    assert np.array_equal(g_dst[:b.dst_bytes], o_dst[:b.dst_bytes])
    # checksum of checksums: one number that pins the whole batch (and the generator) in the log of a failing run
    recs = synth.stream_records(b.streams)
    step = max(1, n // 64)
    sums = [O.xxh64(bytes(g_dst[int(recs["dst_off"][i]):int(recs["dst_off"][i]) + size])) for i in range(0, n, step)]
    assert O.xxh64(np.array(sums, dtype=np.uint64).tobytes()) == O.xxh64(np.array(
        [O.xxh64(bytes(o_dst[int(recs["dst_off"][i]):int(recs["dst_off"][i]) + size])) for i in range(0, n, step)], dtype=np.uint64).tobytes())
    if queue_formats is not None:
        _resident_plan_and_compare(b, o_dst, orr, size, queue_formats)
    return b, g_dst


def _resident_plan_and_compare(b, o_dst, orr, size, queue_formats):
    """The kernel bench.py TIMES, under test at the full size: alz_plan_create on the device-resident batch (which, unlike the host-buffer entry point above, may plan
    a format as a work queue of (stream, chunk) items), the number of items the plan holds against what the cut by alz_debug_chunk_bytes() predicts, two executes into a
    destination filled with 0xA5 between guard regions, and then EVERY byte of the buffer -- inside the streams the oracle's, outside them still 0xA5 -- plus status,
    length and src_used of every stream."""
    import ctypes as C
    from auroralib.compression_amd.batch import Plan
    c = ctx()
    c.lib.alz_debug_plan_queue_items.argtypes = [C.c_void_p]
    c.lib.alz_debug_chunk_repeats.restype = C.c_uint64; c.lib.alz_debug_chunk_repeats.argtypes = [C.c_void_p]
    recs = synth.stream_records(b.streams)
    GUARD, CANARY = 8192, 0xA5
    total = GUARD + b.dst_bytes + GUARD
    want_items = 0
    for f in queue_formats:
        ch = 81920 if f in (A.FMT_PRS_BE, A.FMT_PRS_LE) else c.lib.alz_debug_chunk_bytes()
        want_items += int(((recs["decom_len"][recs["format"] == f].astype(np.int64) + ch - 1) // ch).sum())
    d_src, d_dst = c.malloc(b.src.nbytes + 64), c.malloc(total)
    plan = None
    try:
        c.h2d(d_src, b.src)
        plan = Plan(c, b.streams)
        assert c.lib.alz_debug_plan_queue_items(plan.h) == want_items, (c.lib.alz_debug_plan_queue_items(plan.h), want_items)
        repeats = c.lib.alz_debug_chunk_repeats(c.h)
        c.memset(d_dst, CANARY, total)
        plan.execute(d_src, C.c_void_p(d_dst.value + GUARD))
        plan.execute(d_src, C.c_void_p(d_dst.value + GUARD))     # (again, behind the first: every queue head and flag is set up anew)
        gr = synth.result_records(plan.results())
        assert c.lib.alz_debug_chunk_repeats(c.h) == repeats       # (no bounded spin ran out: the bytes below are the queue kernel's, not the repair's)
        for f in ("status", "dst_len", "src_used"):
            assert np.array_equal(gr[f], orr[f]), f
        # the whole buffer, in pieces of 256 MiB: the oracle's bytes inside [dst_off, dst_off + dst_len) of every stream, the canary everywhere else
        offs, lens = recs["dst_off"].astype(np.int64), orr["dst_len"].astype(np.int64)
        order = np.argsort(offs, kind="stable")
        piece = 256 << 20
        for a in range(0, total, piece):
            e = min(total, a + piece)
            g = c.d2h(d_dst, e - a, offset=a)
            want = np.full(e - a, CANARY, dtype=np.uint8)
            for i in order[np.searchsorted(offs[order] + lens[order], a - GUARD, side="right"):np.searchsorted(offs[order], e - GUARD, side="left")]:
                lo, hi = max(int(offs[i]), a - GUARD), min(int(offs[i] + lens[i]), e - GUARD)      # (destination offsets)
                if lo < hi:
                    want[lo + GUARD - a:hi + GUARD - a] = o_dst[lo:hi]
            if not np.array_equal(g, want):
                at = int(np.flatnonzero(g != want)[0]) + a - GUARD
                raise AssertionError("resident plan: device buffer differs at offset %d of the destination (gpu 0x%02X, expected 0x%02X; %d bytes differ in this piece)"
                                     % (at, int(g[at + GUARD - a]), int(want[at + GUARD - a]), int((g != want).sum())))
    finally:
        if plan is not None:
            plan.close()
        c.free(d_src); c.free(d_dst)


def test_cfg2_yaz0_10000_x_64k():
    _decode_and_compare(A.FMT_YAZ0, 10000, 65536, synth.seed_for(2, 7), queue_formats=(A.FMT_YAZ0,))


def test_metric_config_yaz0_10000_x_256k():
    """The batch bench.py times by default (same generator seed)."""
    _decode_and_compare(A.FMT_YAZ0, 10000, 262144, synth.seed_for(2), queue_formats=(A.FMT_YAZ0,))


def test_cfg3_lz4_blocks_256k():
    """cfg3's stream shape; 10 000 of its 100 000 blocks (the full count is 24 GiB of output: tools/cfg34.sh times it)."""
    _decode_and_compare(A.FMT_LZ4_BLOCK, 10000, 262144, synth.seed_for(3))


def test_cfg3_lz4_100000_blocks_full_count():
    """BASELINE configs[2] at its stated count: 100 000 independent LZ4 blocks x 256 KiB decoded as ONE device-resident batch
    (24.4 GiB of output), generated part by part like bench.py's cfg3 entry.  Every status / length / src_used is compared with
    the oracle's, and every part is downloaded and compared byte for byte (the oracle decodes a 2.4 GiB part in seconds on the
    host cores)."""
    import ctypes as C
    from auroralib.compression_amd.batch import Plan
    c = ctx()
    n, size, parts = 100000, 262144, 10
    per = n // parts
    streams = (A.Stream * n)()
    rec = synth.stream_records(streams)
    batches, offs, src_total = [], [], 0
    for p in range(parts):
        b = synth.make_batch(A.FMT_LZ4_BLOCK, per, size, synth.seed_for(3) + p * per)
        r = synth.stream_records(b.streams)
        sl = slice(p * per, (p + 1) * per)
        rec["src_off"][sl] = r["src_off"] + src_total
        rec["src_len"][sl], rec["dst_cap"][sl], rec["decom_len"][sl], rec["format"][sl] = r["src_len"], size, size, A.FMT_LZ4_BLOCK
        rec["dst_off"][sl] = (np.arange(per, dtype=np.uint64) + np.uint64(p * per)) * np.uint64(size)
        batches.append(b); offs.append(src_total)
        src_total += (b.src.nbytes + 63) // 64 * 64
    d_src, d_dst = c.malloc(src_total + 64), c.malloc(n * size + 64)
    plan = None
    try:
        for b, off in zip(batches, offs):
            c.h2d(C.c_void_p(d_src.value + off), b.src)
        plan = Plan(c, streams)
        plan.execute(d_src, d_dst)
        c.synchronize()
        gr = synth.result_records(plan.results())
        assert (gr["status"] == 0).all() and (gr["dst_len"] == size).all()
        sums = []
        for p, b in enumerate(batches):
            o_dst, o_res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=min(CORES, 32))
            orr = synth.result_records(o_res)
            sl = slice(p * per, (p + 1) * per)
            for f in ("status", "dst_len", "src_used"):
                assert np.array_equal(gr[f][sl], orr[f]), (p, f)
            r = synth.stream_records(b.streams)
            assert (r["dst_off"] == np.arange(per, dtype=np.uint64) * np.uint64(size)).all()      # the generator packs its outputs
            g = c.d2h(d_dst, per * size, offset=p * per * size)
            assert np.array_equal(g, o_dst[:per * size]), p
            sums.append(O.xxh64(g[::64].tobytes()))
        assert len(set(sums)) == parts                            # (ten different parts)
    finally:
        if plan is not None:
            plan.close()
        c.free(d_src); c.free(d_dst)


def test_cfg4_mixed_shard():
    """What each of 8 GPUs gets of cfg4's 40 000 mixed streams: 5 000, formats interleaved, per-format kernel dispatch."""
    n = 5000
    fm = np.array([[A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_PRS_BE][i % 4] for i in range(n)], dtype=np.uint32)
    _decode_and_compare(fm, n, 262144, synth.seed_for(4), queue_formats=())        # (1 250 streams per format: everything resident at once, no queue -- DESIGN 4.2)


@pytest.mark.parametrize("quality", [0, 8])
def test_cfg5_lzss_compression_10000_x_256k(quality):
    n, size = 10000, 262144
    b = synth.make_batch(A.FMT_LZSS, n, size, synth.seed_for(5))
    raw, res = ctx().decode_batch(b.streams, b.src, b.dst_bytes)
    recs = synth.stream_records(b.streams)
    cap = size + size // 4 + 64
    streams = (A.Stream * n)()
    r2 = synth.stream_records(streams)
    r2["src_off"], r2["src_len"] = recs["dst_off"], size
    r2["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64((cap + 255) // 256 * 256)
    r2["dst_cap"], r2["format"] = cap, A.FMT_LZSS
    dst_bytes = int(r2["dst_off"][-1]) + cap + 64
    comp, eres, aux = ctx().encode_batch(streams, raw, dst_bytes, quality=quality)
    er = synth.result_records(eres)
    assert (er["status"] == 0).all()
    assert (er["dst_len"] < size).all()                       # every synthetic buffer compresses
    # round trip of the whole batch on the GPU
    s3 = (A.Stream * n)()
    r3 = synth.stream_records(s3)
    r3["src_off"], r3["src_len"], r3["dst_off"], r3["dst_cap"], r3["decom_len"], r3["format"] = r2["dst_off"], er["dst_len"], recs["dst_off"], size, size, A.FMT_LZSS
    back, dres = ctx().decode_batch(s3, comp, b.dst_bytes)
    dr = synth.result_records(dres)
    assert (dr["status"] == 0).all() and np.array_equal(dr["src_used"], er["dst_len"])
    assert np.array_equal(back[:b.dst_bytes], raw[:b.dst_bytes])
    # bit-identity with the oracle's encoder on a sample
    for i in range(0, n, n // 16):
        a = int(recs["dst_off"][i])
        want, _ = O.encode_stream(A.FMT_LZSS, bytes(raw[a:a + size]), quality=quality)
        o = int(r2["dst_off"][i])
        assert bytes(comp[o:o + int(er["dst_len"][i])]) == want, i
