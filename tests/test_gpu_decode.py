"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Edge cases follow the reference's tests and the frozen definitions E1-E6:
empty and ragged inputs, truncated inputs, overshoot of the declared size, capacity clipping,
self-overlapping matches, sources before the stream start."""
import os

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from gpu_common import compare_batch, ctx, pack_streams

pytestmark = pytest.mark.gpu
ALL = list(range(A.FMT_COUNT))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_device_is_mi355x():
    info = ctx().info()
    assert "gfx950" in info["name"], info
    assert info["cu_count"] == 256


def test_kat_test_lz_on_gpu():
    """LzssStaticDecodingTest through the GPU: XXH64 == 11520079745250749767 (CompressionAlgorithmTest.cs:31-48)."""
    data = open(os.path.join(GOLD, "Test.lz"), "rb").read()
    lz = A.LzProperties.from_bits(10, 6, 2)
    out, r = ctx().decode(A.FMT_LZSS, data[16:], decom_len=1048726, lz=lz)
    assert r.status == A.ST_OK and r.dst_len == 1048726 and r.src_used == 285913
    assert O.xxh64(out) == 11520079745250749767


@pytest.mark.parametrize("fmt", ALL)
def test_synthetic_small_sizes(fmt):
    sizes = np.array([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 18, 19, 31, 63, 64, 65, 100, 255, 256, 257, 1000, 1023, 1024, 1025,
                      4095, 4096, 4097, 5000, 8191, 8192, 8193, 10000], dtype=np.uint32)
    b = synth.make_batch(fmt, len(sizes), sizes, synth.seed_for(90 + fmt), dst_align=16)
    compare_batch(b.streams, b.src, b.dst_bytes, what=A.FORMAT_NAMES[fmt])


@pytest.mark.parametrize("fmt", ALL)
def test_synthetic_64k_and_256k(fmt):
    for target, n in ((65536, 96), (262144, 24)):
        b = synth.make_batch(fmt, n, target, synth.seed_for(20 + fmt, target))
        gr, _ = compare_batch(b.streams, b.src, b.dst_bytes, what="%s %d" % (A.FORMAT_NAMES[fmt], target))
        assert (gr["status"] == 0).all() and (gr["dst_len"] == target).all()


def test_mixed_format_batch():
    """cfg4-style: LZ10/LZ11/Yaz0/PRS interleaved in one batch, per-format kernel dispatch."""
    n = 256
    fm = np.array([[A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_PRS_BE][i % 4] for i in range(n)], dtype=np.uint32)
    b = synth.make_batch(fm, n, 65536, synth.seed_for(4))
    gr, _ = compare_batch(b.streams, b.src, b.dst_bytes, what="mixed")
    assert (gr["status"] == 0).all()


def test_lzss_geometries():
    for bits in [(8, 4, 2), (10, 6, 2), (12, 4, 2), (13, 5, 2), (14, 4, 2), (16, 8, 2)]:
        lz = A.LzProperties.from_bits(*bits)
        b = synth.make_batch(A.FMT_LZSS, 16, np.array([300, 5000, 70000, 9] * 4, dtype=np.uint32), synth.seed_for(70, bits[0]), lz=lz)
        compare_batch(b.streams, b.src, b.dst_bytes, lz=lz, what="lzss%r" % (bits,))


@pytest.mark.parametrize("bits", [(14, 4, 2), (15, 6, 2), (16, 8, 2)])
def test_lzss_windows_beyond_the_lds_ring(bits, test_bmp):
    """14..16 window bits (LzProperties.cs:57-66): the lane-parallel kernel keeps 4 KiB of the window in LDS and reads older
    sources back from the stream's own output (chunked byte phase) -- full-length synthetic streams, whose distances reach the
    whole window, and oracle-encoded Test.bmp at two qualities; truncation and capacity cuts included."""
    lz = A.LzProperties.from_bits(*bits)
    b = synth.make_batch(A.FMT_LZSS, 48, np.array([262144, 70000, 200000, 65537] * 12, dtype=np.uint32), synth.seed_for(71, bits[0]), lz=lz)
    compare_batch(b.streams, b.src, b.dst_bytes, lz=lz, what="wide lzss%r" % (bits,))
    items = []
    for off, size, q in [(0, 300000, 8), (50000, 262144, 0), (7, 100000, 15)]:
        raw = test_bmp[off:off + size]
        body, _ = O.encode_stream(A.FMT_LZSS, raw, quality=q, lz=lz)
        items.append(dict(fmt=A.FMT_LZSS, src=body, decom_len=size))
        items.append(dict(fmt=A.FMT_LZSS, src=body[:len(body) * 2 // 3], decom_len=size))            # truncated
        items.append(dict(fmt=A.FMT_LZSS, src=body, decom_len=size, cap=size // 2))                  # capacity
        items.append(dict(fmt=A.FMT_LZSS, src=body, decom_len=size - 1000, cap=size))                # declared size too small
    streams, src, dst_bytes = pack_streams(items)
    compare_batch(streams, src, dst_bytes, lz=lz, what="wide lzss bmp%r" % (bits,))


@pytest.mark.parametrize("fmt", ALL)
def test_real_data_roundtrip(fmt, test_bmp):
    """Oracle-encoded windows of Test.bmp (the reference's round-trip corpus) decode bit-exactly on the GPU."""
    items = []
    for k, (off, size, q) in enumerate([(0, 10, 4), (0, 10240, 8), (0, 10240, 15), (4096, 65536, 0), (100000, 262144, 8), (500000, 70000, 12)]):
        raw = test_bmp[off:off + size]
        comp, aux = O.encode_stream(fmt, raw, quality=q)
        items.append(dict(fmt=fmt, src=comp, decom_len=len(raw), aux0=aux.aux0, aux1=aux.aux1))
    streams, src, dst_bytes = pack_streams(items)
    gr, g_dst = compare_batch(streams, src, dst_bytes, what="real " + A.FORMAT_NAMES[fmt])
    assert (gr["status"] == 0).all()
    recs = synth.stream_records(streams)
    for k, (off, size, q) in enumerate([(0, 10, 4), (0, 10240, 8), (0, 10240, 15), (4096, 65536, 0), (100000, 262144, 8), (500000, 70000, 12)]):
        a = int(recs["dst_off"][k])
        assert bytes(g_dst[a:a + size]) == test_bmp[off:off + size]


@pytest.mark.parametrize("fmt", ALL)
def test_run_heavy_roundtrip(fmt):
    """Degenerate inputs the lane-parallel parsers see least of in Test.bmp: one repeated byte (every token a maximum-length
    match with its extension byte), short periods, runs broken by single literals."""
    import random
    rng = random.Random(77 + fmt)
    raws = [bytes(200000), b"\xAB" * 70001, b"abc" * 30000, bytes(rng.randrange(256) for _ in range(5000)) * 30,
            b"".join(bytes([rng.randrange(256)]) * rng.choice([1, 2, 3, 17, 18, 19, 272, 273, 274, 300, 5000]) for _ in range(400)),
            b"".join((bytes([rng.randrange(256)]) * rng.randrange(1, 40)) for _ in range(6000))]
    if fmt == A.FMT_LZSHREK:                 # its literal count is a u16 + 286: the managed encoder wraps beyond 65 821 literals in a
        del raws[3]                          # row (LZShrek.cs:165 `(ushort)(plain - 286)`) -- 150 000 bytes of period 5 000 do not round-trip
    items = []
    for k, raw in enumerate(raws):
        comp, aux = O.encode_stream(fmt, raw, quality=[0, 8, 15][k % 3])
        items.append(dict(fmt=fmt, src=comp, decom_len=len(raw), aux0=aux.aux0, aux1=aux.aux1))
    streams, src, dst_bytes = pack_streams(items)
    gr, g_dst = compare_batch(streams, src, dst_bytes, what="runs " + A.FORMAT_NAMES[fmt])
    if fmt == A.FMT_HIG:                      # a 2-byte initial literal block is not decodable (HIG.cs:235 against :137-138): data that
        return                                # starts with a run does not round-trip in the managed code; GPU == oracle was checked above
    assert (gr["status"] == 0).all()
    recs = synth.stream_records(streams)
    for k, raw in enumerate(raws):
        a = int(recs["dst_off"][k])
        assert bytes(g_dst[a:a + len(raw)]) == raw


@pytest.mark.parametrize("fmt", ALL)
def test_truncated_inputs(fmt, test_bmp):
    """EndOfStreamException paths: every prefix length class of a valid stream."""
    from cases import truncated_items
    items = truncated_items(fmt, test_bmp)
    streams, src, dst_bytes = pack_streams(items)
    compare_batch(streams, src, dst_bytes, what="trunc " + A.FORMAT_NAMES[fmt])


@pytest.mark.parametrize("fmt", ALL)
def test_capacity_and_size_mismatch(fmt, test_bmp):
    """E4/E5: declared size smaller than the stream decodes to (overshoot), destination smaller than the output."""
    from cases import capacity_items
    items = capacity_items(fmt, test_bmp)
    streams, src, dst_bytes = pack_streams(items, dst_slack=32)
    compare_batch(streams, src, dst_bytes, what="cap " + A.FORMAT_NAMES[fmt])


def test_handcrafted_edge_tokens():
    """E1 distance==0 / ==W, E2 source before stream start, long self-overlapping runs, Yaz0 length byte at EOF."""
    from cases import handcrafted_items
    items = handcrafted_items()
    streams, src, dst_bytes = pack_streams(items, dst_slack=16)
    compare_batch(streams, src, dst_bytes, what="handcrafted")


def test_unaligned_buffers():
    """src/dst offsets at every residue mod 16: the 16 B granule logic of InCache/OutWin."""
    from cases import unaligned_items
    items = unaligned_items()
    streams, src, dst_bytes = pack_streams(items, dst_slack=8)
    compare_batch(streams, src, dst_bytes, what="unaligned")


def test_sparse_outputs_are_packed_on_the_device():
    """Host-buffer API with outputs far apart (slots much larger than what the streams produce, every alignment, some
    streams failing): the produced bytes come back through the device-side pack + one copy (alz_host.cpp, download_packed),
    and the caller's bytes between them are untouched."""
    import ctypes as C
    from auroralib.compression_amd.batch import _vp, check
    n, stride = 48, 3 << 20
    b = synth.make_batch(A.FMT_YAZ0, n, np.array([1 + 977 * i for i in range(n)], dtype=np.uint32), synth.seed_for(78))
    recs = synth.stream_records(b.streams)
    recs["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64(stride) + (np.arange(n, dtype=np.uint64) * np.uint64(7)) % np.uint64(23)
    recs["dst_cap"] = recs["decom_len"]
    recs["src_len"][5] //= 2                                   # a truncated stream: partial output, not OK
    recs["decom_len"][9] += 3                                  # declared size never reached
    dst_bytes = n * stride
    o_dst, o_res = O.decode_batch(b.streams, b.src, dst_bytes)
    c = ctx()
    g_dst = np.full(dst_bytes, 0xA5, dtype=np.uint8)
    g_res = (A.Result * n)()
    check(c.lib.alz_decode_batch(c.h, None, n, _vp(b.src), b.src.nbytes, b.streams, _vp(g_dst), dst_bytes, g_res))
    gr, orr = synth.result_records(g_res), synth.result_records(o_res)
    assert (gr["status"] == orr["status"]).all() and (gr["dst_len"] == orr["dst_len"]).all() and gr["status"][5] != 0
    keep = np.ones(dst_bytes, dtype=bool)
    for i in range(n):
        a, ln = int(recs["dst_off"][i]), int(gr["dst_len"][i])
        assert bytes(g_dst[a:a + ln]) == bytes(o_dst[a:a + ln]), i
        keep[a:a + ln] = False
    assert (g_dst[keep] == 0xA5).all()


def test_prs_terminator_inside_the_bulk_path(test_bmp):
    """PRS ends at its zero word (PRS.cs:78-79) wherever that is: with >= 1100 bytes of trailing data the terminator is met by
    the lane-assisted parser, not by the tail parser; src_used must stop right behind it."""
    import os
    items = []
    for fmt in (A.FMT_PRS_BE, A.FMT_PRS_LE):
        for size, q in ((30000, 8), (5000, 0), (200, 4)):
            raw = test_bmp[3000:3000 + size]
            comp, _ = O.encode_stream(fmt, raw, quality=q)
            for tail in (os.urandom(3000), bytes(2000), comp):
                items.append(dict(fmt=fmt, src=comp + tail, decom_len=0, cap=size + 64))
    streams, src, dst_bytes = pack_streams(items)
    gr, _ = compare_batch(streams, src, dst_bytes, what="prs trailing data")
    assert (gr["status"] == 0).all()


@pytest.mark.parametrize("fmt", ALL)
def test_fuzz_garbage_and_mutations(fmt, test_bmp):
    """Malformed input: random bytes, valid streams with bit flips / splices, wrong declared sizes.  Every result (status,
    lengths, bytes below dst_len) must equal the oracle's; nothing may hang or write past dst_cap."""
    from cases import fuzz_items
    items = fuzz_items(fmt, test_bmp, seed=int(os.environ.get("ALZ_FUZZ_SEED", "1234")))     # (ALZ_FUZZ_SEED: soak runs with other seeds)
    streams, src, dst_bytes = pack_streams(items, dst_slack=32)
    compare_batch(streams, src, dst_bytes, what="fuzz " + A.FORMAT_NAMES[fmt])


def _agrees_with_the_oracle(c, fmts, damaged_too):
    import random
    sizes = np.array([1, 7, 300, 5000, 70000, 262144, 100001, 64], dtype=np.uint32)
    rng = random.Random(99)
    for fmt in fmts:
        b = synth.make_batch(fmt, len(sizes), sizes, 4242)
        for damaged in ((False, True) if damaged_too else (False,)):
            src = b.src.copy()
            if damaged:
                for _ in range(40):
                    src[rng.randrange(len(src))] ^= 1 << rng.randrange(8)
            o_dst, o_res = O.decode_batch(b.streams, src, b.dst_bytes)
            g_dst, g_res = c.decode_batch(b.streams, src, b.dst_bytes)
            gr, orr = synth.result_records(g_res), synth.result_records(o_res)
            assert all(np.array_equal(gr[f], orr[f]) for f in ("status", "dst_len", "src_used")), (fmt, damaged)
            assert np.array_equal(g_dst[:b.dst_bytes], o_dst[:b.dst_bytes]), (fmt, damaged)


def test_prs_one_wavefront_kernel_still_agrees():
    """PRS runs on two wavefronts per stream by default (alz_decode_prs2_kernel); alz_ctx_set_kernel_variant(1) selects the
    one-wavefront queue kernel, whose loop the second wavefront also falls back to."""
    from auroralib.compression_amd.batch import Context
    with Context(0) as c:
        c.set_kernel_variant(1)
        _agrees_with_the_oracle(c, (A.FMT_PRS_BE, A.FMT_PRS_LE), True)


@pytest.mark.parametrize("variant", [1, 2])
def test_lz4_lzo_snappy_agree_on_one_and_two_wavefronts(variant):
    """LZ4 / LZO / Snappy run on two wavefronts per stream in launches that cannot fill the GPU (alz_decode_queue2_kernel: one parses,
    one executes, literal runs read from the input in global memory) and on one otherwise; alz_ctx_set_kernel_variant forces either."""
    from auroralib.compression_amd.batch import Context
    with Context(0) as c:
        c.set_kernel_variant(variant)
        _agrees_with_the_oracle(c, (A.FMT_LZ4_BLOCK, A.FMT_LZO, A.FMT_SNAPPY_RAW), True)


@pytest.mark.parametrize("variant", [1, 2])
def test_flag_formats_agree_on_one_and_two_wavefronts(variant):
    """The flag-byte formats pick one or two wavefronts per stream by the size of the batch (alz_ctx_set_kernel_variant: 1 / 2 force
    either shape); both kernels must give the oracle's bytes, lengths, consumed input and status for valid and noisy streams."""
    from auroralib.compression_amd.batch import Context
    with Context(0) as c:
        c.set_kernel_variant(variant)
        _agrees_with_the_oracle(c, (A.FMT_LZSS, A.FMT_LZ10, A.FMT_LZ11, A.FMT_LZ40, A.FMT_CLZ0, A.FMT_YAZ0, A.FMT_YAY0, A.FMT_MIO0), True)


def test_cfg1_whole_test_bmp_lz10_at_the_default_quality(test_bmp):
    """BASELINE.json configs[0]: "LZ10 decompress of Test.bmp via the ICompressionAlgorithm path" -- the whole 1 048 726-byte file,
    compressed at the default quality 8 (the managed encoder's bytes: the oracle's), through the LZ10 class of the host mirror:
    IsMatch, GetDecompressedSize, Decompress -- on both kernel families -- and the GPU encoder writes the same file."""
    from auroralib.compression_amd import formats as F
    comp = O.container_compress(A.C_LZ10, test_bmp, quality=8)
    lz10 = F.LZ10()
    assert lz10.IsMatch(comp) and lz10.GetDecompressedSize(comp) == len(test_bmp) == 1048726
    for serial in (1, 0):
        F._context().set_exact_kernels(serial)                    # (the format classes keep a context of their own)
        try:
            assert lz10.Decompress(comp) == test_bmp
        finally:
            F._context().set_exact_kernels(0)
    assert lz10.Compress(test_bmp, F.CompressionSettings(quality=8)) == comp
