"""-m gpu: device-side memory safety made observable.

The host-buffer ABI downloads only [dst_off, dst_off + dst_len) of every stream, so a kernel that wrote past dst_len or past
dst_cap in HBM would pass every parity test of test_gpu_decode.py unchanged.  Here the batches that carry malformed input
(fuzz / truncation / capacity / hand-crafted edge tokens / unaligned buffers) run through the DEVICE-RESIDENT plan path into
a destination that was filled with 0xA5 first and has guard pages on both sides; the whole buffer comes back and

  * every byte outside [dst_off, dst_off + dst_len) of every stream -- the slack between streams, the guards in front of the
    first and behind the last stream, the part of a slot a failed stream did not produce -- must still be 0xA5,
  * every byte inside must equal the oracle's (which also catches a source "before the stream start" (E2) that was read from
    the neighbour's 0xA5 instead of the zero-filled window),
  * status and dst_len must equal the oracle's, and src_used for every status include/auroralz.h defines it for (all but
    OUTPUT_CAPACITY),

for all 25 formats, both kernel families (alz_ctx_set_exact_kernels) and both wave shapes (alz_ctx_set_kernel_variant 1 / 2).
Why it matters: the scan / brute-force producers decode garbage at every offset by design
(CLI/Commands/ScanDecompressCommand.cs:36-37,72-73, BruteForceCommand.cs:88-125).  The encoder's dst_cap is covered the same
way through alz_encode_batch_device."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from auroralib.compression_amd.batch import Plan
from gpu_common import ctx, pack_streams

pytestmark = pytest.mark.gpu
ALL = list(range(A.FMT_COUNT))
CANARY = 0xA5
GUARD = 8192
SEED = int(os.environ.get("ALZ_FUZZ_SEED", "1234"))
# alz_result.src_used is the reference's source.Position after the call; it is compared wherever include/auroralz.h defines it
SRC_USED_DEFINED = (A.ST_OK, A.ST_OUTPUT_SIZE_MISMATCH, A.ST_INPUT_TRUNCATED, A.ST_BAD_TOKEN)


def _first(mask):
    return int(np.flatnonzero(mask)[0])


def canary_decode(streams, src, dst_bytes, lz=None, what="", variants=(0, 1, 2), src_exact=False, queue=False):
    """Runs the batch on every kernel family / wave shape into a canary-filled device buffer and checks the whole buffer."""
    n = len(streams)
    o_dst, o_res = O.decode_batch(streams, src, dst_bytes, lz=lz, nthreads=8)
    orr, sr = synth.result_records(o_res), synth.stream_records(streams)
    expect = np.full(GUARD + dst_bytes + GUARD, CANARY, dtype=np.uint8)
    for i in range(n):
        a, ln = int(sr["dst_off"][i]), int(orr["dst_len"][i])
        assert ln <= int(sr["dst_cap"][i])
        expect[GUARD + a:GUARD + a + ln] = o_dst[a:a + ln]
    c = ctx()
    total = GUARD + dst_bytes + GUARD
    src = np.ascontiguousarray(src, dtype=np.uint8)
    d_src = c.malloc(max(src.nbytes, 16))
    d_dst = c.malloc(total)
    try:
        c.h2d(d_src, src)
        if queue:                                        # (a plan CREATED in variant 3 decodes through the work queue of chunks, whatever its size)
            c.set_kernel_variant(3)
        try:
            plan = Plan(c, streams, lz)
        finally:
            c.set_kernel_variant(0)
        try:
            for serial in ((0,) if queue else (1, 0)):
                for variant in ((3,) if queue else (variants if not serial else (0,))):
                    tag = "%s [%s kernels, variant %d]" % (what, "serial" if serial else "fast", variant)
                    c.set_exact_kernels(serial)
                    c.set_kernel_variant(variant)
                    try:
                        c.memset(d_dst, CANARY, total)
                        plan.execute(d_src, C.c_void_p(d_dst.value + GUARD))
                        gr = synth.result_records(plan.results()).copy()
                        buf = c.d2h(d_dst, total)
                    finally:
                        c.set_exact_kernels(0)
                        c.set_kernel_variant(0)
                    defined = np.isin(orr["status"], SRC_USED_DEFINED)
                    if os.environ.get("ALZ_CANARY_REPORT"):            # (survey aid: where src_used differs in the statuses that leave it unspecified)
                        with open(os.environ["ALZ_CANARY_REPORT"], "a") as fh:
                            for stv in range(5):
                                sel = (orr["status"] == stv) & (gr["status"] == stv)
                                if sel.any():
                                    fh.write("%s status %d: %d streams, src_used differs in %d\n" % (tag, stv, int(sel.sum()), int((sel & (gr["src_used"] != orr["src_used"])).sum())))
                    for f in ("status", "dst_len", "src_used"):
                        bad = (gr[f] != orr[f]) & (defined if f == "src_used" else True)
                        assert not bad.any(), "%s: stream %d (%s, src_len %d, decom_len %d, cap %d) %s gpu=%d oracle=%d [status gpu=%d oracle=%d]" % (
                            tag, _first(bad), A.FORMAT_NAMES[int(sr["format"][_first(bad)])], sr["src_len"][_first(bad)], sr["decom_len"][_first(bad)],
                            sr["dst_cap"][_first(bad)], f, gr[f][_first(bad)], orr[f][_first(bad)], gr["status"][_first(bad)], orr["status"][_first(bad)])
                    diff = buf != expect
                    if diff.any():
                        at = _first(diff) - GUARD
                        owner = [i for i in range(n) if int(sr["dst_off"][i]) <= at < int(sr["dst_off"][i]) + max(int(sr["dst_cap"][i]), 1)]
                        inside = [i for i in owner if at < int(sr["dst_off"][i]) + int(orr["dst_len"][i])]
                        raise AssertionError("%s: device buffer differs at offset %d (%d bytes differ): gpu=0x%02X expected=0x%02X; %s" % (
                            tag, at, int(diff.sum()), int(buf[at + GUARD]), int(expect[at + GUARD]),
                            ("inside the output of stream %d" % inside[0]) if inside else
                            ("inside the slot of stream %d BEHIND its dst_len %d (status %d)" % (owner[0], orr["dst_len"][owner[0]], orr["status"][owner[0]])) if owner else
                            "outside every stream's slot"))
        finally:
            plan.close()
    finally:
        c.free(d_src)
        c.free(d_dst)


@pytest.mark.parametrize("fmt", ALL)
def test_canary_fuzz(fmt, test_bmp):
    from cases import fuzz_items
    streams, src, dst_bytes = pack_streams(fuzz_items(fmt, test_bmp, seed=SEED), dst_slack=32)
    canary_decode(streams, src, dst_bytes, what="fuzz " + A.FORMAT_NAMES[fmt])


@pytest.mark.parametrize("fmt", ALL)
def test_canary_truncated(fmt, test_bmp):
    from cases import truncated_items
    streams, src, dst_bytes = pack_streams(truncated_items(fmt, test_bmp))
    canary_decode(streams, src, dst_bytes, what="trunc " + A.FORMAT_NAMES[fmt])


@pytest.mark.parametrize("fmt", ALL)
def test_canary_capacity(fmt, test_bmp):
    from cases import capacity_items
    streams, src, dst_bytes = pack_streams(capacity_items(fmt, test_bmp), dst_slack=32)
    canary_decode(streams, src, dst_bytes, what="cap " + A.FORMAT_NAMES[fmt])


def test_canary_handcrafted_and_unaligned():
    from cases import handcrafted_items, unaligned_items
    streams, src, dst_bytes = pack_streams(handcrafted_items(), dst_slack=16)
    canary_decode(streams, src, dst_bytes, what="handcrafted")
    streams, src, dst_bytes = pack_streams(unaligned_items(), dst_slack=8)
    canary_decode(streams, src, dst_bytes, what="unaligned")


@pytest.mark.parametrize("fmt", ALL)
def test_canary_valid_streams_every_alignment(fmt):
    """Valid synthetic streams of ragged sizes packed with NO slack at every destination residue mod 16: the ragged head and tail
    granules of the 16 B/lane write-back must not touch the neighbour's bytes."""
    sizes = np.array([1, 2, 3, 15, 16, 17, 31, 33, 63, 64, 65, 100, 255, 257, 1000, 1023, 1025, 4095, 4097, 5000, 8191, 8193, 10000, 70001], dtype=np.uint32)
    b = synth.make_batch(fmt, len(sizes), sizes, synth.seed_for(60 + fmt), dst_align=1)
    canary_decode(b.streams, b.src, b.dst_bytes, what="ragged " + A.FORMAT_NAMES[fmt])


def test_canary_lzss_geometries(test_bmp):
    from cases import fuzz_items
    for bits in [(8, 4, 2), (10, 6, 2), (14, 4, 2), (16, 8, 2)]:
        lz = A.LzProperties.from_bits(*bits)
        streams, src, dst_bytes = pack_streams(fuzz_items(A.FMT_LZSS, test_bmp, seed=SEED + bits[0], count=48), dst_slack=32)
        canary_decode(streams, src, dst_bytes, lz=lz, what="fuzz lzss%r" % (bits,))


# ------------------------------------------------------------------------------------------------ encoder
def _oracle_encode(fmt, raw, quality, **kw):
    try:
        return O.encode_stream(fmt, raw, quality=quality, **kw)[0]
    except ValueError:
        return None


def canary_encode(fmt, raws, caps, quality, what="", exact_src=False, **kw):
    """alz_encode_batch_device into a canary-filled buffer: nothing outside [dst_off, dst_off + dst_len) may change; streams the
    oracle's encoder fits into `cap` must come out byte-identical, the others must fail with OUTPUT_CAPACITY and dst_len 0."""
    n = len(raws)
    streams = (A.Stream * n)()
    so = do = 0
    chunks = []
    for i, r in enumerate(raws):
        streams[i] = A.Stream(so, do, len(r), caps[i], 0, 0, 0, fmt)
        pad = 0 if exact_src else (-len(r)) % 16
        chunks.append(bytes(r) + bytes(pad))
        so += len(r) + pad
        do += caps[i] + (i * 7) % 23                      # ragged slots: every alignment, little or no slack
    src = np.frombuffer(b"".join(chunks) + (b"" if exact_src else bytes(64)), dtype=np.uint8).copy()
    dst_bytes = do
    total = GUARD + dst_bytes + GUARD
    c = ctx()
    d_src = c.malloc(max(src.nbytes, 16))
    d_dst = c.malloc(total)
    try:
        if src.nbytes:
            c.h2d(d_src, src)
        c.memset(d_dst, CANARY, total)
        res, aux = c.encode_batch_device(streams, d_src, src.nbytes, C.c_void_p(d_dst.value + GUARD), dst_bytes, quality=quality, **kw)
        buf = c.d2h(d_dst, total)
    finally:
        c.free(d_src)
        c.free(d_dst)
    expect = np.full(total, CANARY, dtype=np.uint8)
    for i, r in enumerate(raws):
        want = _oracle_encode(fmt, r, quality, **kw)
        a = GUARD + int(streams[i].dst_off)
        if want is None:                                   # the managed encoder refuses this input (e.g. LZ4 below 5 bytes): not OK, nothing usable written
            assert res[i].status != A.ST_OK, (what, i, len(r), res[i].status)
            expect[a:a + caps[i]] = buf[a:a + caps[i]]
        elif len(want) <= caps[i]:
            assert res[i].status == A.ST_OK and res[i].dst_len == len(want), (what, i, len(r), caps[i], res[i].status, res[i].dst_len, len(want))
            expect[a:a + len(want)] = np.frombuffer(want, dtype=np.uint8)
        else:
            assert res[i].status == A.ST_OUTPUT_CAPACITY and res[i].dst_len == 0, (what, i, len(r), caps[i], res[i].status, res[i].dst_len, len(want))
            expect[a:a + caps[i]] = buf[a:a + caps[i]]   # a failed stream may have written any part of ITS slot, nothing else
    diff = buf != expect
    if diff.any():
        at = _first(diff) - GUARD
        raise AssertionError("%s: encoder output buffer differs at offset %d (%d bytes): gpu=0x%02X expected=0x%02X; slots %r" % (
            what, at, int(diff.sum()), int(buf[at + GUARD]), int(expect[at + GUARD]), [(int(s.dst_off), int(s.dst_cap)) for s in streams]))


@pytest.mark.parametrize("fmt", ALL)
@pytest.mark.parametrize("quality", [0, 8])
def test_canary_encode_capacity(fmt, quality, test_bmp):
    """dst_cap exactly the compressed size, one byte less, a third of it, zero, and generous -- in ragged, slack-free slots."""
    rng = np.random.default_rng(11 + fmt)
    raws = [test_bmp[:10240], test_bmp[100000:100000 + 30000], bytes(5000), bytes(rng.integers(0, 256, 3000, dtype=np.uint8)), b"abc", b"",
            test_bmp[4096:4096 + 70000]]
    sizes = [len(_oracle_encode(fmt, r, quality) or b"") for r in raws]
    for k, pick in enumerate((lambda s: s, lambda s: max(s - 1, 0), lambda s: s // 3, lambda s: 0, lambda s: s + 100)):
        canary_encode(fmt, raws, [pick(s) for s in sizes], quality, what="%s q%d caps#%d" % (A.FORMAT_NAMES[fmt], quality, k))


@pytest.mark.parametrize("fmt", [A.FMT_LZSS, A.FMT_YAZ0, A.FMT_LZ4_BLOCK, A.FMT_PRS_BE, A.FMT_LZO])
def test_encode_source_allocated_exactly(fmt, test_bmp):
    """The raw buffers end exactly at the end of the device allocation (no slack behind src_bytes): the encoder's look-ahead loads must
    stay inside it (loadL / kernel B read up to 28 bytes behind a stream).  An out-of-bounds read shows up as a memory fault
    or as a difference from the oracle; sizes around a 2 MiB allocation granule put the end of the last stream on a page boundary."""
    raws = [test_bmp[:65536], test_bmp[65536:65536 + 2 * 1024 * 1024 - 65536]]
    for q in (0, 8):
        canary_encode(fmt, raws, [len(r) + len(r) // 4 + 64 for r in raws], q, what="exact src %s q%d" % (A.FORMAT_NAMES[fmt], q), exact_src=True)


def test_encode_tiny_streams_after_a_dirty_scratch(test_bmp):
    """at quality 0 the fused search of enc_parse_emit_kernel read the link slot of position 0 for streams shorter
    than 4 bytes -- a slot kernel A never wrote for them -- and dereferenced data - link.  First a batch that leaves non-zero links in
    the grow-only scratch, then 1-3 byte streams at src_off 0."""
    rng = np.random.default_rng(3)
    big = [test_bmp[:200000], bytes(rng.integers(0, 4, 100000, dtype=np.uint8))]
    for fmt in (A.FMT_LZSS, A.FMT_LZ10, A.FMT_YAZ0, A.FMT_LZ4_BLOCK):
        for q in (0, 8):
            canary_encode(fmt, big, [len(r) + len(r) // 4 + 64 for r in big], q, what="dirty %s q%d" % (A.FORMAT_NAMES[fmt], q))
            for tiny in ([b"a"], [b"ab"], [b"abc"], [b"abc", b"z", b"", b"xy", b"abcd"]):
                canary_encode(fmt, tiny, [16] * len(tiny), q, what="tiny %s q%d %r" % (A.FORMAT_NAMES[fmt], q, tiny), exact_src=True)
