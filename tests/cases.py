"""Hand-crafted edge-case streams shared by the CPU (oracle ring vs flat) and GPU parity tests."""
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
