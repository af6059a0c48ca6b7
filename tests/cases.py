"""Hand-crafted edge-case streams shared by the CPU (oracle ring vs flat) and GPU parity tests."""
import numpy as np

from auroralib.compression_amd import _abi as A


def handcrafted_items():
    """E1 distance==0 / ==W, E2 source before stream start, long self-overlapping runs, Yaz0 length byte at EOF."""

    items = []
    # LZ10: literal 'A', then match d=1 len=18 x many (RLE), then d=4096 (before start -> zeros)
    body = bytes([0b01111111, 0x41] + [0xF0, 0x00] * 7)
    items.append(dict(fmt=A.FMT_LZ10, src=body, decom_len=1 + 18 * 7))
    body = bytes([0b01000000, 0x42, 0xFF, 0xFF, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48])   # d=4096,len=18 at pos 1: reads before start
    items.append(dict(fmt=A.FMT_LZ10, src=body, decom_len=1 + 18 + 6))
    # Yaz0: 3-byte token whose length byte is missing (EOF -> 17)
    items.append(dict(fmt=A.FMT_YAZ0, src=bytes([0b10000000, 0x55, 0x00, 0x00]), decom_len=18))
    items.append(dict(fmt=A.FMT_YAZ0, src=bytes([0b10000000, 0x55, 0x00, 0x00]), decom_len=40))
    # LZ11: 4-byte token, len 273+ with d=1
    items.append(dict(fmt=A.FMT_LZ11, src=bytes([0b01000000, 0x61, 0x10, 0x3E, 0x70, 0x00]), decom_len=1 + 273 + 999))
    # LZ4: distance 0 (E1 -> d = 65536 -> zeros), then literals
    items.append(dict(fmt=A.FMT_LZ4_BLOCK, src=bytes([0x14, 0x61, 0x00, 0x00, 0x50, 1, 2, 3, 4, 5]), decom_len=0, cap=64))
    # LZ4: long literal + long match via extension bytes
    lit = bytes(range(256)) * 2
    items.append(dict(fmt=A.FMT_LZ4_BLOCK, src=bytes([0xFF, 255, 512 - 15 - 255]) + lit + bytes([0x01, 0x00, 255, 255, 10, 0x50, 9, 9, 9, 9, 9]), decom_len=0, cap=4096))
    # PRS BE: literal, long match with v>>3 == 0 (d = 8192 -> zeros), terminator
    items.append(dict(fmt=A.FMT_PRS_BE, src=bytes([0b10101000 | 0b00000000, 0x31, 0x00, 0x03, 0x00, 0x00]), decom_len=0, cap=64))
    # Snappy: copy-4 with distance > 65536 -> BAD_TOKEN ; copy with distance 0
    items.append(dict(fmt=A.FMT_SNAPPY_RAW, src=bytes([20, 0x00, 0x61, 0x0F, 0x00, 0x00, 0x02, 0x00]), decom_len=0, cap=64))
    items.append(dict(fmt=A.FMT_SNAPPY_RAW, src=bytes([20, 0x00, 0x61, 0x0E, 0x00, 0x00, 0x00, 0x61, 0x62]), decom_len=0, cap=64))
    # LZO: first byte > 17 literal run then end marker; and empty input
    items.append(dict(fmt=A.FMT_LZO, src=bytes([17 + 5, 1, 2, 3, 4, 5, 0x11, 0, 0]), decom_len=0, cap=64))
    items.append(dict(fmt=A.FMT_LZO, src=b"", decom_len=0, cap=64))
    # PRS: the destination fills up (E5) in front of the place where the input ends -- the capacity error comes first
    # in stream order, although a queueing parser only notices it when the queue is executed
    for f in (A.FMT_PRS_BE, A.FMT_PRS_LE):
        for cap in (0, 1, 4, 8, 9):
            items.append(dict(fmt=f, src=bytes([0xFF]) + b"ABCDEFGH" + bytes([0xFF]) + b"IJK", decom_len=0, cap=cap))
    for f in (A.FMT_LZSS, A.FMT_LZ10, A.FMT_YAZ0, A.FMT_PRS_BE, A.FMT_LZ4_BLOCK, A.FMT_SNAPPY_RAW, A.FMT_MIO0, A.FMT_YAY0):
        items.append(dict(fmt=f, src=b"", decom_len=0, cap=0))
        items.append(dict(fmt=f, src=b"", decom_len=10, cap=10))
    return items


# ---- builders of the malformed / ragged batches (shared by the host-buffer parity tests of test_gpu_decode.py and the
# device-resident canary tests of test_gpu_canary.py; they need the oracle's encoder, so they import it lazily)
def truncated_items(fmt, bmp):
    """EndOfStreamException paths: every prefix length class of a valid stream."""
    import oracle_lib as O
    raw = bmp[1000:1000 + 3000]
    comp, aux = O.encode_stream(fmt, raw, quality=8)
    cuts = sorted(set([0, 1, 2, 3, 4, 5, 8, 9, 17, len(comp) // 3, len(comp) // 2, len(comp) - 3, len(comp) - 2, len(comp) - 1]))
    return [dict(fmt=fmt, src=comp[:c], decom_len=len(raw), cap=len(raw), aux0=aux.aux0, aux1=aux.aux1) for c in cuts if c >= 0]


def capacity_items(fmt, bmp):
    """E4/E5: declared size smaller than the stream decodes to (overshoot), destination smaller than the output."""
    import oracle_lib as O
    raw = bmp[2000:2000 + 20000]
    comp, aux = O.encode_stream(fmt, raw, quality=8)
    items = []
    for decl, cap in [(20000, 20000), (19990, 19990), (19990, 20010), (10000, 10000), (10000, 10001), (20000, 5000), (20000, 0),
                      (1, 1), (0, 0), (20000, 19999), (25000, 25000)]:
        items.append(dict(fmt=fmt, src=comp, decom_len=decl, cap=cap, aux0=aux.aux0, aux1=aux.aux1))
    return items


def fuzz_items(fmt, bmp, seed=1234, count=96):
    """Malformed input: random bytes, valid streams with bit flips / splices, wrong declared sizes."""
    import random
    import oracle_lib as O
    rng = random.Random(seed + fmt)
    items = []
    raw = bmp[7000:7000 + 30000]
    comp, aux = O.encode_stream(fmt, raw, quality=4)
    for k in range(count):
        kind = k % 4
        if kind == 0:                                     # pure noise, lengths around every threshold of the bulk parsers
            n = rng.choice([0, 1, 2, 7, 63, 64, 129, 1000, 1099, 1100, 1101, 1500, 4000, 9000])
            src = bytes(rng.randrange(256) for _ in range(n))
        elif kind == 1:                                   # bit flips in a valid stream
            b = bytearray(comp)
            for _ in range(rng.randrange(1, 6)):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            src = bytes(b)
        elif kind == 2:                                   # noise spliced into a valid stream
            cut = rng.randrange(len(comp))
            src = comp[:cut] + bytes(rng.randrange(256) for _ in range(rng.randrange(1, 3000))) + comp[cut:]
        else:                                             # biased noise (many zeros / 0xFF: long runs, terminators, extensions)
            n = rng.randrange(1100, 6000)
            src = bytes(rng.choice([0, 0, 0xFF, 0x0F, 0xF0, rng.randrange(256)]) for _ in range(n))
        decl = rng.choice([len(raw), len(raw), 100, 70000, 0])
        cap = rng.choice([decl, decl + 300, max(decl, 1) // 2, 70000])
        three = fmt in (A.FMT_YAY0, A.FMT_MIO0, A.FMT_SMSR00)   # aux = section offsets there (LZ4: aux0 would be frame history)
        items.append(dict(fmt=fmt, src=src, decom_len=decl, cap=cap, aux0=aux.aux0 if kind in (1, 2) else (rng.randrange(0, 3000) if three else 0),
                          aux1=aux.aux1 if kind in (1, 2) else (rng.randrange(0, 3000) if three else 0)))
    return items


def unaligned_items():
    """src/dst offsets at every residue mod 16: the 16 B granule logic of InCache/OutWin."""
    import oracle_lib as O
    from auroralib.compression_amd import synth
    items = []
    b = synth.make_batch(A.FMT_YAZ0, 16, 5000, synth.seed_for(77))
    recs = synth.stream_records(b.streams)
    for i in range(16):
        s = bytes(b.src[int(recs["src_off"][i]):int(recs["src_off"][i]) + int(recs["src_len"][i])])
        items.append(dict(fmt=A.FMT_YAZ0, src=s, decom_len=5000, src_misalign=1, dst_misalign=1))
        items.append(dict(fmt=A.FMT_LZ4_BLOCK, src=O.encode_stream(A.FMT_LZ4_BLOCK, s + s, quality=4)[0], decom_len=0, cap=len(s) * 2, src_misalign=3, dst_misalign=5))
    return items


def prose_like(size, seed):
    """Deterministic prose-like bytes: words of a small vocabulary with a skewed distribution, punctuation, indentation, line breaks --
    short matches at many distances, frequent lazy-parse decisions (what a bitmap does not have)."""
    rng = np.random.default_rng(seed)
    vocab = [bytes(rng.integers(97, 123, int(rng.integers(1, 10)), dtype=np.uint8)) for _ in range(600)]
    w = 1.0 / np.arange(1, len(vocab) + 1) ** 1.1
    w /= w.sum()
    out, total = [], 0
    while total < size:
        k = int(rng.integers(3, 14))
        words = [vocab[i] for i in rng.choice(len(vocab), k, p=w)]
        line = b" " * (4 * int(rng.integers(0, 4))) + b" ".join(words) + (b";" if rng.random() < 0.3 else b".") + b"\n"
        out.append(line); total += len(line)
    return b"".join(out)[:size]
