"""Hand-built LZ4 frames / legacy files / Snappy framed streams for the container tests."""
import random
import struct


def lz4_len_ext(n):
    out = bytearray()
    n -= 15
    while n >= 255:
        out.append(255)
        n -= 255
    out.append(n)
    return bytes(out)


def lz4_linked_blocks(seed, nblocks, block_bytes, max_dist=65535):
    """Random LZ4 blocks whose matches reach back across block boundaries (one window per frame, LZ4.Frame.cs:120).
    Returns (list of compressed blocks, expected output), the output computed by a plain byte-wise model."""
    rng = random.Random(seed)
    out = bytearray()
    blocks = []
    for _ in range(nblocks):
        blk = bytearray()
        start = len(out)
        while len(out) - start < block_bytes:
            lit = bytes(rng.randrange(256) for _ in range(min(int(rng.expovariate(1 / 6.0)), 300)))
            if not out and not lit:
                lit = b"x"
            mlen = 4 + min(int(rng.expovariate(1 / 10.0)), 600)
            dist = rng.randint(1, min(len(out) + len(lit), max_dist))
            if rng.random() < 0.1:
                dist = min(dist, rng.randint(1, 4))
            tok = (min(len(lit), 15) << 4) | min(mlen - 4, 15)
            blk.append(tok)
            if len(lit) >= 15:
                blk += lz4_len_ext(len(lit))
            blk += lit
            out += lit
            blk += struct.pack("<H", dist)
            if mlen - 4 >= 15:
                blk += lz4_len_ext(mlen - 4)
            for _ in range(mlen):
                out.append(out[-dist])
        lit = bytes(rng.randrange(256) for _ in range(5 + rng.randrange(20)))     # last sequence: literals only
        blk.append(min(len(lit), 15) << 4)
        if len(lit) >= 15:
            blk += lz4_len_ext(len(lit))
        blk += lit
        out += lit
        blocks.append(bytes(blk))
    return blocks, bytes(out)


def lz4_frame(blocks, xxh32, flg=0x40, bd=0x40, content=None, raw_flags=None, content_size=None):
    """Frame around ready-made blocks.  flg bits: 4 content checksum, 8 content size, 16 block checksum, 32 independent."""
    desc = bytearray([flg, bd])
    if flg & 8:
        desc += struct.pack("<Q", content_size if content_size is not None else len(content))
    hc = (xxh32(bytes(desc)) >> 8) & 0xFF
    f = bytearray(struct.pack("<I", 0x184D2204)) + desc + bytes([hc])
    for i, b in enumerate(blocks):
        raw = bool(raw_flags and raw_flags[i])
        f += struct.pack("<I", len(b) | (0x80000000 if raw else 0)) + b
        if flg & 16:
            f += struct.pack("<I", xxh32(b))
    f += struct.pack("<I", 0)
    if flg & 4:
        f += struct.pack("<I", xxh32(content))
    return bytes(f)


def lz4_legacy(blocks, eof_flag=True):
    f = bytearray(struct.pack("<I", 0x184C2102))
    for b in blocks:
        f += struct.pack("<I", len(b)) + b
    if eof_flag:
        f.append(0xFF)
    return bytes(f)
