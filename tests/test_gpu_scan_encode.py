"""-m gpu: the encoder's SCAN path (csrc/alz_encode.hip: enc_scan_select_kernel, benc_wave_scan_search, enc_parse_emit_kernel<FMT, false, true>; round 6) against the oracle's
restatement of LzChainMatchFinder + CompressHeaderless.  Streams whose parse visits few positions go without kernels A and B: every position the cursor stands on is
searched exactly by scanning the window behind it for the positions with its hash.  The bytes must be IDENTICAL to the managed encoder's whichever way a stream goes:

  * forced (alz_debug_scan_mode 1: every eligible stream scans -- whatever its data): the ten flag-bit formats, LZ4 blocks and raw Snappy x qualities 2..9, Test.bmp pieces of every kind, runs,
    noise, token soup on every lane of a 64-position window, buffer lengths around one / two / three windows, LZSS geometries, CompatibilityMode, a minimum distance,
    a destination one byte short (whole-buffer canary through alz_encode_batch_device);
  * chosen by the probe (mode 0) on a batch that mixes flat, mixed and photographic windows: some streams must go each way (alz_debug_scan_streams), all bytes the oracle's;
  * off (mode 2): the same call gives the same bytes.

The whole-GPU path of a few big buffers and the segmented path of a small batch are switched off for these calls (they would take the batch first)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from gpu_common import ctx
from test_gpu_encode import _token_soup

pytestmark = pytest.mark.gpu
FAM = [A.FMT_LZSS, A.FMT_LZ10, A.FMT_LZ11, A.FMT_LZ40, A.FMT_YAZ0, A.FMT_YAY0, A.FMT_MIO0, A.FMT_CLZ0, A.FMT_BLZ, A.FMT_LZHUDSON,
       A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW]          # (the last two: enc_scan_seq_kernel -- sequences instead of flag bits, windows of 64 / 32 KiB)
OFF = 0xFFFFFFFF


def _lib(c):
    c.lib.alz_debug_scan_streams.restype = C.c_uint64
    c.lib.alz_debug_scan_streams.argtypes = [C.c_void_p]
    c.lib.alz_debug_scan_mode.argtypes = [C.c_void_p, C.c_int]
    c.lib.alz_debug_seg_max_streams.argtypes = [C.c_void_p, C.c_uint32]
    return c.lib


class _regular_batch_pipeline:
    """The batch pipeline proper for whatever the number of buffers: no whole-GPU path, no segmented path; the scan path in `mode`."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        c = ctx()
        lib = _lib(c)
        c.big_stream(OFF)
        lib.alz_debug_seg_max_streams(c.h, 0)
        lib.alz_debug_scan_mode(c.h, self.mode)
        return c

    def __exit__(self, *a):
        c = ctx()
        c.lib.alz_debug_scan_mode(c.h, 0)
        c.lib.alz_debug_seg_max_streams(c.h, OFF)
        c.big_stream(24 << 10)


def _encode(c, fmt, raws, quality, **kw):
    n = len(raws)
    streams = (A.Stream * n)()
    so = do = 0
    chunks = []
    for i, r in enumerate(raws):
        cap = len(r) + len(r) // 4 + 64
        streams[i] = A.Stream(so, do, len(r), cap, 0, 0, 0, fmt)
        pad = (-len(r)) % 16
        chunks.append(bytes(r) + bytes(pad))
        so += len(r) + pad
        do += (cap + 15) // 16 * 16
    src = np.frombuffer(b"".join(chunks) + bytes(64), dtype=np.uint8).copy()
    dst, res, aux = c.encode_batch(streams, src, do + 64, quality=quality, **kw)
    return [(res[i].status, bytes(dst[streams[i].dst_off:streams[i].dst_off + res[i].dst_len]), aux[i].aux0, aux[i].aux1) for i in range(n)]


def _check(fmt, raws, quality, mode=1, expect_taken=None, **kw):
    with _regular_batch_pipeline(mode) as c:
        before = c.lib.alz_debug_scan_streams(c.h)
        got = _encode(c, fmt, raws, quality, **kw)
        taken = c.lib.alz_debug_scan_streams(c.h) - before
    for i, r in enumerate(raws):
        st, g, a0, a1 = got[i]
        try:
            want, waux = O.encode_stream(fmt, r, quality=quality, **kw)
        except ValueError:                                         # (LZ4 blocks of fewer than five bytes: the managed encoder throws)
            assert st != A.ST_OK, (A.FORMAT_NAMES[fmt], i, "oracle refuses, gpu accepted")
            continue
        assert st == A.ST_OK, (A.FORMAT_NAMES[fmt], quality, i, len(r), st)
        if g != want:
            k = next((j for j in range(min(len(g), len(want))) if g[j] != want[j]), min(len(g), len(want)))
            raise AssertionError("%s q%d mode %d stream %d (%d B): gpu %d B vs oracle %d B, first difference at %d" % (A.FORMAT_NAMES[fmt], quality, mode, i, len(r), len(g), len(want), k))
        assert (a0, a1) == (waux.aux0, waux.aux1)
    if expect_taken is not None:
        assert expect_taken(taken), (A.FORMAT_NAMES[fmt], quality, mode, taken, len(raws))
    return taken


@pytest.mark.parametrize("fmt", FAM)
@pytest.mark.parametrize("quality", [2, 5, 8, 9])
def test_forced_scan_bit_identical_bmp(fmt, quality, test_bmp):
    raws = [test_bmp[:10], test_bmp[:10240], test_bmp[100000:100000 + 65536], test_bmp[500000:500000 + 30000], test_bmp[4096:4096 + 262144], test_bmp[96 * 4096:96 * 4096 + 262144],
            test_bmp[720000:720000 + 100001]]
    _check(fmt, raws, quality, expect_taken=lambda t: t == len(raws))


@pytest.mark.parametrize("fmt", FAM)
def test_forced_scan_edge_inputs_and_window_edges(fmt):
    rng = np.random.default_rng(50 + fmt)
    raws = [b"", b"a", b"ab", b"abc", b"abcd", b"abcde", bytes(5), bytes(15), bytes(16), bytes(17), bytes(0x100), bytes(5000), bytes(70000),
            b"ab" * 3000, b"abc" * 1000 + b"x", bytes(rng.integers(0, 256, 3000, dtype=np.uint8)),
            bytes(rng.integers(0, 4, 20000, dtype=np.uint8)), (b"0123456789" * 30 + bytes(rng.integers(0, 256, 50, dtype=np.uint8))) * 20]
    sizes = list(range(1, 24)) + list(range(56, 72)) + list(range(120, 136)) + list(range(184, 200)) + [255, 256, 257, 1023, 1024, 1025, 1026, 1027, 2047, 2048, 2049, 4095, 4096, 4097, 4098, 4099, 4100, 5119, 5120, 5121, 8191, 8192, 8193]
    raws += [_token_soup(rng, n) for n in sizes] + [_token_soup(rng, int(rng.integers(300, 20000))) for _ in range(40)]
    raws += [bytes([7]) * n for n in (63, 64, 65, 128, 129, 5000)] + [(b"abcdefg" * 1000)[:n] for n in (64, 65, 127, 4099)]
    for q in (3, 8):
        _check(fmt, raws, q, expect_taken=lambda t: t == len(raws))


def test_forced_scan_settings(test_bmp):
    """A minimum distance (VRAM mode), CompatibilityMode (no self-overlap), LZSS geometries with 256-byte to 4 KiB windows."""
    for fmt in (A.FMT_LZ10, A.FMT_LZ11):
        _check(fmt, [test_bmp[:20000], bytes(300), test_bmp[400000:430000]], 8, min_distance=2)
    _check(A.FMT_LZSS, [test_bmp[:20000], bytes(300), b"ab" * 500, test_bmp[400000:440000]], 8, strategy=1)
    for bits in [(10, 6, 2), (12, 4, 2), (8, 4, 2), (11, 5, 3)]:
        _check(A.FMT_LZSS, [test_bmp[:30000], bytes(1000), test_bmp[390000:460000]], 8, lz=A.LzProperties.from_bits(*bits))


def test_probe_chooses_per_stream_and_off_is_off(test_bmp):
    """Mode 0 on 2 100 windows of 32 KiB from all over Test.bmp (photographic at the top, flat further down; the path is only taken from 2 048 buffers of a format on: a scan stream costs its
    own latency, which only a launch that keeps the GPU busy hides): the probe sends some streams each way; mode 2 none; a small batch none; bytes the oracle's."""
    raws = [test_bmp[(i * 457) % (len(test_bmp) - 32768):][:32768] for i in range(2100)]
    for fmt, q in ((A.FMT_YAZ0, 8), (A.FMT_LZ11, 5), (A.FMT_LZ4_BLOCK, 8)):
        _check(fmt, raws, q, mode=0, expect_taken=lambda t: 0 < t < len(raws))
    _check(A.FMT_YAZ0, raws, 8, mode=2, expect_taken=lambda t: t == 0)
    _check(A.FMT_YAZ0, raws[:1000], 8, mode=0, expect_taken=lambda t: t == 0)
    # (formats whose matches end at 18 bytes are not sent that way unless forced: kernel B is fast on them)
    _check(A.FMT_LZ10, raws, 8, mode=0, expect_taken=lambda t: t == 0)
    # qualities / formats the path does not cover go the regular way even when forced
    small = [test_bmp[i * 20000:i * 20000 + 65536] for i in range(6)]
    _check(A.FMT_YAZ0, small, 12, mode=1, expect_taken=lambda t: t == 0)
    _check(A.FMT_YAZ0, small, 0, mode=1, expect_taken=lambda t: t == 0)
    _check(A.FMT_LZ4_BLOCK, small, 11, mode=1, expect_taken=lambda t: t == 0)       # (maxChain 64: more candidates than the lanes measure at once)
    _check(A.FMT_LZ4_BLOCK, small + [bytes(70000) + b"x" + bytes(30000)], 10, mode=1, expect_taken=lambda t: t == 7)   # (maxChain 32: the most the path takes)


def test_forced_scan_capacity_one_byte_short_with_canary(test_bmp):
    """alz_encode_batch_device into a canary-filled destination: a stream whose capacity is one byte short fails with OUTPUT_CAPACITY and nothing outside any stream's capacity is written."""
    from auroralib.compression_amd import synth
    raws = [test_bmp[96 * 4096:96 * 4096 + 65536], test_bmp[:30000], test_bmp[300000:300000 + 50000]]
    with _regular_batch_pipeline(1) as c:
        for fmt in (A.FMT_YAZ0, A.FMT_YAY0, A.FMT_LZ10):
            want = [O.encode_stream(fmt, r, quality=8)[0] for r in raws]
            n = len(raws)
            streams = (A.Stream * n)()
            so = do = 0
            chunks = []
            for i, r in enumerate(raws):
                cap = len(want[i]) - (1 if i == 1 else 0)
                streams[i] = A.Stream(so, do, len(r), cap, 0, 0, 0, fmt)
                pad = (-len(r)) % 16
                chunks.append(bytes(r) + bytes(pad)); so += len(r) + pad
                do += (cap + 15) // 16 * 16 + 32
            src = np.frombuffer(b"".join(chunks) + bytes(64), dtype=np.uint8).copy()
            total = do + 64
            d_src, d_dst = c.malloc(src.nbytes + 64), c.malloc(total)
            try:
                c.h2d(d_src, src); c.memset(d_dst, 0xA5, total)
                res, aux = c.encode_batch_device(streams, d_src, src.nbytes, d_dst, total, quality=8)
                buf = c.d2h(d_dst, total)
            finally:
                c.free(d_src); c.free(d_dst)
            expect = np.full(total, 0xA5, dtype=np.uint8)
            for i in range(n):
                if i == 1:
                    assert res[i].status == A.ST_OUTPUT_CAPACITY
                    buf[streams[i].dst_off:streams[i].dst_off + streams[i].dst_cap] = 0xA5      # (what a failed stream leaves inside its capacity is unspecified)
                else:
                    assert res[i].status == A.ST_OK and res[i].dst_len == len(want[i])
                    expect[streams[i].dst_off:streams[i].dst_off + len(want[i])] = np.frombuffer(want[i], dtype=np.uint8)
            assert np.array_equal(buf, expect), A.FORMAT_NAMES[fmt]


@pytest.mark.parametrize("fmt", FAM)
def test_forced_scan_fuzz(fmt, test_bmp):
    """Random inputs under ALZ_FUZZ_SEED (tools/soak.sh repeats this under other seeds): token soup, slices of Test.bmp with noise spliced in, runs with a defect, at a random quality 2..9
    (LZ4 blocks / raw Snappy: up to 10), forced through the scan path -- bytes, lengths and section offsets the oracle's."""
    import os
    seed = int(os.environ.get("ALZ_FUZZ_SEED", "1234")) * 43 + fmt
    rng = np.random.default_rng(seed)
    raws = []
    for _ in range(24):
        kind = int(rng.integers(0, 4))
        n = int(rng.integers(1, 90000))
        if kind == 0:
            raws.append(_token_soup(rng, n))
        elif kind == 1:
            a = int(rng.integers(0, len(test_bmp) - n))
            b = bytearray(test_bmp[a:a + n])
            for _k in range(int(rng.integers(0, 6))):
                p = int(rng.integers(0, n)); m = int(rng.integers(1, 200))
                b[p:p + m] = rng.integers(0, 256, len(b[p:p + m]), dtype=np.uint8).tobytes()
            raws.append(bytes(b))
        elif kind == 2:
            b = bytearray([int(rng.integers(0, 256))] * n)
            for _k in range(int(rng.integers(0, 4))):
                b[int(rng.integers(0, n))] ^= 0x55
            raws.append(bytes(b))
        else:
            unit = rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8).tobytes()
            raws.append((unit * (n // len(unit) + 1))[:n])
    q = int(rng.integers(2, 11 if fmt in (A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW) else 10))
    _check(fmt, raws, q, expect_taken=lambda t: t == len(raws))
