"""-m gpu: the reference's two batch producers -- ScanDecompressCommand and BruteForceCommand -- as callers of the
batched GPU path, checked against a literal replay of the managed loops through the oracle."""
import os
import random

import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import formats as F
from auroralib.compression_amd import scan as S

pytestmark = pytest.mark.gpu


def replay_scan(data, classes):
    """ScanDecompressCommand.Execute (ScanDecompressCommand.cs:51-104) literally: byte-wise walk, Identify = first IsMatch,
    decode through the oracle, keep when no exception and more than 0x10 bytes came out."""
    found, i = [], 0
    while i < len(data):
        adv = i + 1
        for cls in classes:
            if cls().IsMatch(data[i:]):                      # IsMatch is host code (tested in test_container_host.py)
                try:
                    size = O.container_decompressed_size(cls.container, data[i:])
                    if size <= 256 << 20:
                        o = A.ContainerOptions(); o.big_endian = 1
                        import ctypes as C
                        dst = C.create_string_buffer(max(size, 1)); dl, su, st = C.c_size_t(), C.c_size_t(), C.c_int32()
                        rc = O.lib.oracle_container_decompress(cls.container, C.byref(o), data[i:], len(data) - i, dst, size, C.byref(dl), C.byref(su), C.byref(st))
                        if rc == 0 and dl.value > 0x10:
                            found.append((i, i + su.value, cls.container, dst.raw[:dl.value]))
                            adv = i + su.value
                except ValueError:
                    pass
                break
        i = adv
    return found


def build_image(test_bmp, seed):
    """A 'ROM image': junk, real streams of several formats at odd offsets, and decoy magics that do not decode."""
    rng = random.Random(seed)
    parts, truth = [], []
    pos = 0

    def junk(n):
        return bytes(rng.randrange(256) for _ in range(n))

    specs = [(A.C_YAZ0, 30000, 8), (A.C_LZ10, 5000, 4), (A.C_LZ11, 70000, 0), (A.C_YAY0, 12000, 8), (A.C_MIO0, 9000, 8), (A.C_LZSS, 20000, 8),
             (A.C_GCLZ, 4000, 8), (A.C_COMP, 6000, 8), (A.C_YAZ0, 100, 4), (A.C_LZ10, 40, 8), (A.C_YAZ1, 8000, 8), (A.C_YAZ0, 16, 8)]
    for k, (c, size, q) in enumerate(specs):
        j = junk(rng.randrange(1, 700))
        parts.append(j); pos += len(j)
        off = rng.randrange(0, len(test_bmp) - size)
        raw = test_bmp[off:off + size]
        comp = O.container_compress(c, raw, quality=q)
        truth.append((pos, c, raw))
        parts.append(comp); pos += len(comp)
        if k % 3 == 0:                                        # decoys: a magic followed by garbage, a truncated real stream
            d = b"Yaz0" + junk(60) + comp[:len(comp) // 2] + b"MIO0" + junk(40) + bytes([0x10, 0x40, 0, 0]) + junk(30)
            parts.append(d); pos += len(d)
    parts.append(junk(300))
    return b"".join(parts), truth


def test_scan_finds_embedded_streams(test_bmp):
    image, truth = build_image(test_bmp, 1)
    classes = [F.Yaz0, F.Yay0, F.MIO0, F.LZSS, F.GCLZ, F.COMP, F.Yaz1, F.LZ10, F.LZ11]
    hits = S.scan(image, classes)
    assert hits == replay_scan(image, classes)
    got = {(h[0], h[2]): h[3] for h in hits}
    exact = 0
    for pos, c, raw in truth:
        if len(raw) > 0x10:
            if got.get((pos, c)) == raw:
                exact += 1
            else:   # swallowed by a false positive starting in the junk before it, or not identified at all: LZ10.Validate walks on
                    # into the junk behind a stream with fewer than four matches (LZ10.cs:139-175)
                cls = [k for k in classes if k.container == c][0]
                assert any(h[0] < pos < h[1] for h in hits) or not cls().IsMatch(image[pos:]), (pos, c)
        else:
            assert (pos, c) not in got                        # destination.Length <= 0x10: not kept (ScanDecompressCommand.cs:85)
    assert exact >= 9
    # one format only (Execute(sourceFile, destinationFolder, format), :12-46)
    only = S.scan(image, [F.Yaz0])
    assert only == replay_scan(image, [F.Yaz0]) and all(h[2] == A.C_YAZ0 for h in only) and len(only) >= 2


def test_scan_rank4_containers(test_bmp):
    """The magic-carrying containers added with SURVEY 8f rank 4 (and SMSR00) as scan targets."""
    rng = random.Random(7)
    parts, truth, pos = [], [], 0
    for c, size, q in [(A.C_CNX2, 20000, 8), (A.C_CLZ0, 9000, 4), (A.C_CNS, 30000, 8), (A.C_HIG, 15000, 8), (A.C_SMSR00, 12000, 8), (A.C_CNX2, 70000, 0)]:
        j = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 500)))
        parts.append(j); pos += len(j)
        off = rng.randrange(0, len(test_bmp) - size)
        raw = test_bmp[off:off + size]
        comp = O.container_compress(c, raw, quality=q)
        truth.append((pos, c, raw)); parts.append(comp); pos += len(comp)
    image = b"".join(parts) + bytes(rng.randrange(256) for _ in range(200))
    classes = [F.CNX2, F.CLZ0, F.CNS, F.HIG, F.SMSR00]
    hits = S.scan(image, classes)
    assert hits == replay_scan(image, classes)
    got = {(h[0], h[2]): h[3] for h in hits}
    assert sum(1 for pos, c, raw in truth if got.get((pos, c)) == raw) == len(truth)


def test_scan_nested_and_adjacent(test_bmp):
    """A stream whose payload contains another stream's bytes: the walk continues behind the outer one (:98)."""
    inner = O.container_compress(A.C_LZ10, test_bmp[:3000], quality=8)
    outer_raw = os.urandom(200) + inner + os.urandom(200)     # incompressible: the inner file survives verbatim as literals
    outer = O.container_compress(A.C_YAZ0, outer_raw, quality=8)
    image = os.urandom(50) + outer + inner + outer
    classes = [F.Yaz0, F.LZ10]
    hits = S.scan(image, classes)
    assert hits == replay_scan(image, classes)
    assert [h[2] for h in hits if h[0] >= 50][:3] == [A.C_YAZ0, A.C_LZ10, A.C_YAZ0]


def test_brute_force(test_bmp):
    """BruteForceCommand: every raw decoder against a fixed destination; exactly the right one 'successfully unpacks'."""
    raw = test_bmp[5000:5000 + 40000]
    lz1062 = A.LzProperties.from_bits(10, 6, 2)
    cases = [("LZ10", A.FMT_LZ10, None), ("LZ11", A.FMT_LZ11, None), ("Yaz0", A.FMT_YAZ0, None), ("PRS big", A.FMT_PRS_BE, None),
             ("PRS Little", A.FMT_PRS_LE, None), ("LZO", A.FMT_LZO, None), ("LZ4", A.FMT_LZ4_BLOCK, None),
             ("LZSS (12, 4, 2)", A.FMT_LZSS, None), ("LZSS (10, 6, 2)", A.FMT_LZSS, lz1062), ("LZ40", A.FMT_LZ40, None), ("LZHudson", A.FMT_LZHUDSON, None),
             ("RefPack", A.FMT_REFPACK, None), ("LZ02", A.FMT_LZ02, None), ("CLZ0", A.FMT_CLZ0, None), ("CNS", A.FMT_CNS, None), ("LZShrek", A.FMT_LZSHREK, None)]
    names = ["LZO", "LZ4", "LZSS (12, 4, 2)", "LZSS (12, 4, 3)", "LZSS (10, 6, 2)", "LZSS (10, 6, 3)", "LZSS0", "PRS big", "PRS Little", "LZ10", "LZ11", "Yaz0", "LZ40", "LZHudson", "RefPack", "LZ02", "CLZ0", "CNS", "LZShrek"]
    fmts = {"LZO": (A.FMT_LZO, None), "LZ4": (A.FMT_LZ4_BLOCK, None), "LZSS (12, 4, 2)": (A.FMT_LZSS, A.LzProperties.from_bits(12, 4, 2)),
            "LZSS (12, 4, 3)": (A.FMT_LZSS, A.LzProperties.from_bits(12, 4, 3)), "LZSS (10, 6, 2)": (A.FMT_LZSS, lz1062),
            "LZSS (10, 6, 3)": (A.FMT_LZSS, A.LzProperties.from_bits(10, 6, 3)), "LZSS0": (A.FMT_LZSS, A.LzProperties.from_bits(12, 4, 2)),
            "PRS big": (A.FMT_PRS_BE, None), "PRS Little": (A.FMT_PRS_LE, None), "LZ10": (A.FMT_LZ10, None), "LZ11": (A.FMT_LZ11, None), "Yaz0": (A.FMT_YAZ0, None),
            "LZ40": (A.FMT_LZ40, None), "LZHudson": (A.FMT_LZHUDSON, None), "RefPack": (A.FMT_REFPACK, None), "LZ02": (A.FMT_LZ02, None),
            "CLZ0": (A.FMT_CLZ0, None), "CNS": (A.FMT_CNS, None), "LZShrek": (A.FMT_LZSHREK, None)}
    for name, fmt, lz in cases:
        comp, _ = O.encode_stream(fmt, raw, quality=8, lz=lz)
        res = S.brute_force(comp, len(raw))
        assert list(res.keys()) == names
        assert res[name][0] and res[name][2] == raw, name
        if name == "LZSS (12, 4, 2)":
            assert res["LZSS0"][0]                              # the same geometry under another name (LZSS.cs:34)
        # every decoder's verdict and output equal the oracle's on the same fixed destination
        for n2, (f2, lz2) in fmts.items():
            st = (A.Stream * 1)(A.Stream(0, 0, len(comp), len(raw), len(raw), 0, 0, f2))
            import numpy as np
            o_dst, o_res = O.decode_batch(st, np.frombuffer(comp + bytes(64), dtype=np.uint8), len(raw) + 64, lz=lz2)
            ok = o_res[0].status == A.ST_OK and o_res[0].dst_len == len(raw)
            assert res[n2][0] == ok and res[n2][1] == o_res[0].status, (name, n2)
            assert res[n2][2] == bytes(o_dst[:o_res[0].dst_len]), (name, n2)


def test_scan_retries_a_failed_yaz0_candidate_with_the_size_byte_swapped(test_bmp):
    """Yaz0.Decompress catches the failure of the first attempt and decodes again with the size field read in the other byte
    order (Yaz0.cs:66-78); the scan does the same for a candidate that fails.  Size 0x00020000 reads as 0x200 big-endian, and
    the token that crosses byte 0x200 of this stream is a long match: the first attempt overshoots its declared size."""
    raw = test_bmp[5000:5300] + bytes(1000) + test_bmp[7000:7000 + 0x20000 - 1300]
    assert len(raw) == 0x20000
    body = O.encode_stream(A.FMT_YAZ0, raw, quality=8)[0]
    first, r = O.decode_stream(A.FMT_YAZ0, body, decom_len=0x200, cap=0x200 + 300)
    assert r.status == A.ST_OUTPUT_SIZE_MISMATCH                      # the big-endian reading fails ...
    le_file = b"Yaz0" + (0x20000).to_bytes(4, "little") + bytes(8) + body
    junk = bytes(range(7, 200))
    hits = S.scan(junk + le_file + junk, [F.Yaz0])
    assert len(hits) == 1 and hits[0][0] == len(junk) and hits[0][3] == raw and hits[0][1] == len(junk) + len(le_file)   # ... the retry decodes
    assert F.Yaz0().Decompress(le_file) == raw                        # (Decompress itself always retried)


def test_scan_keeps_a_yaz0_candidate_whose_first_size_reading_is_absurd(test_bmp):
    """A little-endian size with a non-zero low byte (0x00012345) reads big-endian as 0x45230100 -- more than any stream the scan
    takes: the managed first attempt runs out of input and the retry reads the field reversed (Yaz0.cs:66-78); the scan decodes such
    a candidate with the reversed size at once instead of dropping it."""
    raw = test_bmp[9000:9000 + 0x12345]
    body = O.encode_stream(A.FMT_YAZ0, raw, quality=8)[0]
    le_file = b"Yaz0" + (0x12345).to_bytes(4, "little") + bytes(8) + body
    junk = bytes(range(3, 120))
    hits = S.scan(junk + le_file + junk, [F.Yaz0])
    assert len(hits) == 1 and hits[0][0] == len(junk) and hits[0][3] == raw and hits[0][1] == len(junk) + len(le_file)
    assert F.Yaz0().Decompress(le_file) == raw
