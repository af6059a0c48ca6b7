"""CPU: the ring+flush restatement of LzWindows and the flat out[q]=out[q-d] model (what the kernels implement)
agree on every format, on synthetic batches, real data and the hand-crafted edge tokens."""
import numpy as np
import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from cases import handcrafted_items


def _both(fmt, src, **kw):
    a, ra = O.decode_stream(fmt, src, **kw)
    b, rb = O.decode_stream(fmt, src, flat=True, **kw)
    assert (ra.status, ra.dst_len) == (rb.status, rb.dst_len), (A.FORMAT_NAMES[fmt], ra.status, rb.status, ra.dst_len, rb.dst_len)
    assert a == b
    if ra.status in (A.ST_OK, A.ST_OUTPUT_SIZE_MISMATCH):
        assert ra.src_used == rb.src_used
    return a, ra


def test_handcrafted_ring_vs_flat():
    for it in handcrafted_items():
        _both(it["fmt"], it["src"], decom_len=it.get("decom_len", 0), cap=it.get("cap", it.get("decom_len", 0)),
              aux0=it.get("aux0", 0), aux1=it.get("aux1", 0))


def test_handcrafted_expected_values():
    # LZ10 RLE: 'A' x 127
    out, r = _both(A.FMT_LZ10, bytes([0b01111111, 0x41] + [0xF0, 0x00] * 7), decom_len=127)
    assert r.status == A.ST_OK and out == b"A" * 127
    # E2: source before the stream start reads zeros
    out, r = _both(A.FMT_LZ10, bytes([0b01000000, 0x42, 0xFF, 0xFF, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48]), decom_len=25)
    assert r.status == A.ST_OK and out == b"B" + bytes(18) + b"CDEFGH"
    # Yaz0 length byte at EOF -> 17 (Stream.ReadByte() == -1)
    out, r = _both(A.FMT_YAZ0, bytes([0b10000000, 0x55, 0x00, 0x00]), decom_len=18)
    assert r.status == A.ST_OK and out == b"U" * 18
    out, r = _both(A.FMT_YAZ0, bytes([0b10000000, 0x55, 0x00, 0x00]), decom_len=40)
    assert r.status == A.ST_INPUT_TRUNCATED and r.dst_len == 18
    # Snappy copy-4 beyond the window is refused
    out, r = _both(A.FMT_SNAPPY_RAW, bytes([20, 0x00, 0x61, 0x0F, 0x00, 0x00, 0x02, 0x00]), cap=64)
    assert r.status == A.ST_BAD_TOKEN
    # LZO first-byte literal run + end marker
    out, r = _both(A.FMT_LZO, bytes([17 + 5, 1, 2, 3, 4, 5, 0x11, 0, 0]), cap=64)
    assert r.status == A.ST_OK and out == bytes([1, 2, 3, 4, 5])
    # overshoot: LZ10 declared 100 but the last match crosses it -> SIZE_MISMATCH, bytes written up to cap
    out, r = _both(A.FMT_LZ10, bytes([0b01111111, 0x41] + [0xF0, 0x00] * 7), decom_len=100, cap=127)
    assert r.status == A.ST_OUTPUT_SIZE_MISMATCH and r.dst_len == 109
    out, r = _both(A.FMT_LZ10, bytes([0b01111111, 0x41] + [0xF0, 0x00] * 7), decom_len=100, cap=100)
    assert r.status == A.ST_OUTPUT_SIZE_MISMATCH and r.dst_len == 100
    out, r = _both(A.FMT_LZ10, bytes([0b01111111, 0x41] + [0xF0, 0x00] * 7), decom_len=100, cap=50)
    assert r.status == A.ST_OUTPUT_CAPACITY and r.dst_len == 50


@pytest.mark.parametrize("fmt", range(A.FMT_COUNT))
def test_synthetic_ring_vs_flat(fmt):
    sizes = np.array([1, 2, 5, 17, 100, 1000, 4095, 4096, 4097, 8193, 20000, 65536, 70000, 140000], dtype=np.uint32)
    b = synth.make_batch(fmt, len(sizes), sizes, synth.seed_for(50 + fmt))
    recs = synth.stream_records(b.streams)
    for i in range(b.n):
        s = bytes(b.src[int(recs["src_off"][i]):int(recs["src_off"][i]) + int(recs["src_len"][i])])
        out, r = _both(fmt, s, decom_len=int(sizes[i]), aux0=int(recs["aux0"][i]), aux1=int(recs["aux1"][i]))
        assert r.status == A.ST_OK and r.dst_len == sizes[i]


@pytest.mark.parametrize("fmt", range(A.FMT_COUNT))
def test_truncation_and_capacity_ring_vs_flat(fmt, test_bmp):
    raw = test_bmp[1000:4000]
    comp, aux = O.encode_stream(fmt, raw, quality=8)
    for cut in range(0, len(comp), max(1, len(comp) // 40)):
        _both(fmt, comp[:cut], decom_len=len(raw), aux0=aux.aux0, aux1=aux.aux1)
    for decl, cap in [(3000, 3000), (2990, 2990), (2990, 3010), (1000, 1001), (3000, 500), (3000, 0), (0, 0), (3500, 3500)]:
        _both(fmt, comp, decom_len=decl, cap=cap, aux0=aux.aux0, aux1=aux.aux1)
