#!/usr/bin/env python3
"""make_kats.py -- hand-assembled known-answer tests for the eleven north-star decode bodies.

This script does NOT call the oracle, the library or any decoder.  Every case is written down as a list of operations
    ("lit", bytes)      literal bytes
    ("copy", d, n)      n bytes from distance d (out[q] = out[q - d], LzWindows.BackCopy  IO/LzWindows.cs:72-100)
and two independent things are derived from that list:
  * the expected OUTPUT: the operations applied to a plain Python bytearray (expand());
  * the compressed STREAM: the operations written in the token encoding the cited C# decoder reads -- one small writer per
    format below, each line next to the decoder statement it is the inverse of (paths under /root/reference/src).
A decoder that disagrees with the C# field layout therefore fails these vectors even when its own encoder agrees with it
(the round-trip matrix cannot see that).  The vectors cover every token form of every body: each length class and its
extension bytes at their boundaries, distance extremes, self-overlapping copies, partial last flag groups, the lazy flag
fetch of PRS in both bit orders, the LZO `plain` states (and its "> 17" first byte quirk) and the Yaz0 length-byte-at-EOF rule.

Output: tests/golden/kat_<format>.json (committed).  tests/test_kat.py checks the oracle against them (CPU suite) and the
HIP path through the C ABI (-m gpu); tests/test_kat.py also re-runs this script and compares, so the files cannot drift.
"""
import base64
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))


def expand(ops):
    out = bytearray()
    for op in ops:
        if op[0] == "skip":                                       # (CNX2: bytes the decoder jumps over)
            continue
        if op[0] == "lit":
            out += bytes(op[1])
        else:
            d, n = op[1], op[2]
            assert 1 <= d <= len(out), (d, len(out))
            for _ in range(n):
                out.append(out[-d])
    return bytes(out)


def tokens(ops):
    """literal runs split into single literals (the flag-byte formats have one flag bit per literal byte)"""
    for op in ops:
        if op[0] == "lit":
            for b in bytes(op[1]):
                yield ("lit", b)
        else:
            yield op


# ------------------------------------------------------------------------------------------------ flag-byte family
class Flags8:
    """8-bit flag byte written in FRONT of the payload of its (up to) 8 tokens: the decoder fetches a flag byte when it has no
    bit left (FlagReader.Readbit  IO/FlagReader.cs:52-64) and the payload of a token right after its bit."""

    def __init__(self, msb_first):
        self.msb, self.out, self.pos, self.n = msb_first, bytearray(), None, 0

    def bit(self, v):
        if self.n == 0:
            self.pos = len(self.out); self.out.append(0); self.n = 8
        shift = self.n - 1 if self.msb else 8 - self.n          # FlagReader.cs:60
        self.out[self.pos] |= (1 if v else 0) << shift
        self.n -= 1


def enc_lzss(ops, wbits=12, lbits=4, thr=2):
    """LZSS.DecompressHeaderless  AuroraLib.Compression/Formats/Common/LZSS.cs:91-130; LzProperties bit ctor LzProperties.cs:57-66"""
    W, minlen = 1 << wbits, thr + 1
    ws = W - (1 << lbits) - thr                                   # WindowsStart (LzProperties.cs:63)
    f, pos = Flags8(msb_first=False), 0                           # :95 flags LSB first
    for t in tokens(ops):
        if t[0] == "lit":
            f.bit(1); f.out.append(t[1]); pos += 1                # :106-108
        else:
            _, d, n = t
            assert minlen <= n <= minlen + (1 << lbits) - 1 and 1 <= d <= W
            ring = (pos - d) % W                                  # address OffsetCopy must resolve to  LzWindows.cs:108-115
            off = (ring + ws) % W                                 # :117 off = (MaxDistance + off - WindowsStart) & (MaxDistance - 1)
            f.bit(0); f.out.append(off & 0xFF); f.out.append(((off >> 8) << lbits) | (n - minlen))   # :115-116
            pos += n
    return bytes(f.out)


def enc_lz10(ops):
    """LZ10.DecompressHeaderless  AuroraLib.Compression.Nintendo/Nintendo/LZ10.cs:82-111"""
    f = Flags8(msb_first=True)                                    # :88
    for t in tokens(ops):
        if t[0] == "lit":
            f.bit(0); f.out.append(t[1])                          # :102
        else:
            _, d, n = t
            assert 3 <= n <= 18 and 1 <= d <= 4096
            f.bit(1); f.out.append(((n - 3) << 4) | ((d - 1) >> 8)); f.out.append((d - 1) & 0xFF)   # :94-98
    return bytes(f.out)


def enc_lz11(ops):
    """LZ11.DecompressHeaderless  AuroraLib.Compression.Nintendo/Nintendo/LZ11.cs:83-133"""
    f = Flags8(msb_first=True)
    for t in tokens(ops):
        if t[0] == "lit":
            f.bit(0); f.out.append(t[1])
        else:
            _, d, n = t
            assert 1 <= d <= 4096
            f.bit(1)
            if n <= 16:                                           # :113-118  n = (b1 >> 4) + 1  (3..16: nibbles 0 and 1 are taken)
                assert n >= 3
                f.out += bytes([((n - 1) << 4) | ((d - 1) >> 8), (d - 1) & 0xFF])
            elif n <= 272:                                        # :98-104  n = ((b1 & 0xF) << 4 | b2 >> 4) + 17
                v = n - 17
                f.out += bytes([v >> 4, ((v & 0xF) << 4) | ((d - 1) >> 8), (d - 1) & 0xFF])
            else:                                                 # :105-112  n = ((b1 & 0xF) << 12 | b2 << 4 | b3 >> 4) + 273
                v = n - 273
                assert v < (1 << 16)
                f.out += bytes([0x10 | (v >> 12), (v >> 4) & 0xFF, ((v & 0xF) << 4) | ((d - 1) >> 8), (d - 1) & 0xFF])
    return bytes(f.out)


def yay0_token(d, n):
    """Yay0.DecompressHeaderless  AuroraLib.Compression.Nintendo/Nintendo/Yay0.cs:124-134: (two token bytes, length byte or None)"""
    assert 1 <= d <= 4096 and 3 <= n <= 0x111
    if n <= 17:
        return bytes([((n - 2) << 4) | ((d - 1) >> 8), (d - 1) & 0xFF]), None     # :133 n = (b1 >> 4) + 2
    return bytes([(d - 1) >> 8, (d - 1) & 0xFF]), n - 0x12                         # :130-131 n = ReadByte() + 0x12


def enc_yaz0(ops, drop_last_length_byte=False):
    """Yaz0 = Yay0.DecompressHeaderless with all three cursors on one stream  Yaz0.cs:91-92, Yay0.cs:110-144"""
    f = Flags8(msb_first=True)
    for t in tokens(ops):
        if t[0] == "lit":
            f.bit(1); f.out.append(t[1])                          # Yay0.cs:118-121: bit 1 = literal
        else:
            tok, lb = yay0_token(t[1], t[2])
            f.bit(0); f.out += tok
            if lb is not None:
                f.out.append(lb)
    out = bytes(f.out)
    return out[:-1] if drop_last_length_byte else out


def enc_3cursor(ops, mio0):
    """Yay0 (Yay0.cs:99-144) / MIO0 (MIO0.cs:105-149): flags | tokens | literals (+ Yay0's length bytes).  Returns
    (stream, aux0 = token section offset, aux1 = literal section offset), all relative to the first flag byte."""
    flags, toks, lits = Flags8(msb_first=True), bytearray(), bytearray()
    for t in tokens(ops):
        if t[0] == "lit":
            flags.bit(1); lits.append(t[1])
        else:
            _, d, n = t
            flags.bit(0)
            if mio0:
                assert 3 <= n <= 18 and 1 <= d <= 4096
                toks += bytes([((n - 3) << 4) | ((d - 1) >> 8), (d - 1) & 0xFF])   # MIO0.cs:129-136
            else:
                tok, lb = yay0_token(d, n)
                toks += tok
                if lb is not None:
                    lits.append(lb)                               # Yay0.cs:130: the length byte comes from the literal stream
    fl = bytes(flags.out)
    fl += bytes((-len(fl)) % 4)                                   # the writers pad the flag section to 4 bytes (Yay0.cs:70-77)
    return fl + bytes(toks) + bytes(lits), len(fl), len(fl) + len(toks)


def enc_clz0(ops):
    """CLZ0.DecompressHeaderless  AuroraLib.Compression-Extended/Marvelous/CLZ0.cs:66-100: flags LSB first (FlagReader(source, Endian.Little) :70),
    bit 1 = match; two bytes: low 8 bits of the window delta, then delta[11:8] << 4 | (length - 3); distance = 0x1000 - delta."""
    f = Flags8(msb_first=False)
    for t in tokens(ops):
        if t[0] == "lit":
            f.bit(0); f.out.append(t[1])                          # :89-92
        else:
            _, d, n = t
            assert 3 <= n <= 18 and 1 <= d <= 4096
            delta = 0x1000 - d                                    # :82
            f.bit(1); f.out += bytes([delta & 0xFF, ((delta >> 8) << 4) | (n - 3)])   # :78-83
    return bytes(f.out)


def enc_lz02(ops, terminate=True):
    """LZ02.DecompressHeaderless  AuroraLib.Compression-Extended/Camelot/LZ02.cs:84-121: flags MSB first, bit 1 = match: DDDDLLLL DDDDDDDD,
    length = nibble + 1 (2..16); nibble 0: distance 0 is the END of the stream, otherwise a third byte holds length - 17 (17..272).  A
    distance field of 0 under a non-zero nibble reaches LzWindows.BackCopy as 0, i.e. one whole window (4096)."""
    f = Flags8(msb_first=True)
    for t in tokens(ops):
        if t[0] == "lit":
            f.bit(0); f.out.append(t[1])                          # :114-117
        else:
            _, d, n = t
            f.bit(1)
            if 2 <= n <= 16:
                assert 1 <= d <= 4096
                dd = d & 0xFFF                                    # 4096 -> 0
                f.out += bytes([((dd >> 8) << 4) | (n - 1), dd & 0xFF])            # :92-96
            else:
                assert 17 <= n <= 272 and 1 <= d <= 4095
                f.out += bytes([(d >> 8) << 4, d & 0xFF, n - 17])                  # :98-109
    if terminate:
        f.bit(1); f.out += bytes([0, 0])                          # :100-107
    return bytes(f.out)


def enc_lz40(ops):
    """LZ40.DecompressHeaderless  AuroraLib.Compression.Nintendo/Nintendo/LZ40.cs:84-132: the flag byte is stored NEGATED (flag = (byte)-ReadByte()
    :97), consumed from bit 7 down, bit 1 = match: a little-endian u16 distance << 4 | length (2..15); length field 0: + one byte, length - 16
    (16..271); length field 1: + a little-endian u16, length - 272 (272..65 807).  Distance field 0 = one whole window."""
    f = Flags8(msb_first=True)
    flagpos = []
    for t in tokens(ops):
        if f.n == 0:
            flagpos.append(len(f.out))
        if t[0] == "lit":
            f.bit(0); f.out.append(t[1])                          # :123-126
        else:
            _, d, n = t
            assert 1 <= d <= 4096
            dd = (d & 0xFFF) << 4
            f.bit(1)
            if 2 <= n <= 15:
                f.out += bytes([(dd | n) & 0xFF, (dd | n) >> 8])                   # :103-105
            elif n <= 271:
                f.out += bytes([dd & 0xFF, dd >> 8, n - 16])                       # :108-112
            else:
                v = n - 272
                assert v < (1 << 16)
                f.out += bytes([(dd | 1) & 0xFF, (dd | 1) >> 8, v & 0xFF, v >> 8])  # :113-117
    out = bytearray(f.out)
    for q in flagpos:
        out[q] = (-out[q]) & 0xFF
    return bytes(out)


class FlagsWide:
    """FlagReader with a flag word of `nbytes` bytes, big endian, consumed from the top bit down (FlagReader(source, Endian.Big, n, Endian.Big)):
    the word sits where the decoder is when it needs a bit and has none left, i.e. in front of the payload of its first token."""

    def __init__(self, nbytes):
        self.w, self.out, self.pos, self.n = nbytes, bytearray(), None, 0

    def bit(self, v):
        if self.n == 0:
            self.pos = len(self.out); self.out += bytes(self.w); self.n = 8 * self.w
        k = self.n - 1                                            # bit index in the word, MSB first
        if v:
            self.out[self.pos + (self.w - 1 - k // 8)] |= 1 << (k % 8)
        self.n -= 1


def enc_lzhudson(ops):
    """LZHudson.DecompressHeaderless  AuroraLib.Compression.Nintendo/HudsonSoft/LZHudson.cs:52-53: Yay0's tokens (Yay0.cs:110-144) behind 32-bit
    big-endian flag words, all three cursors on one stream."""
    f = FlagsWide(4)
    for t in tokens(ops):
        if t[0] == "lit":
            f.bit(1); f.out.append(t[1])
        else:
            tok, lb = yay0_token(t[1], t[2])
            f.bit(0); f.out += tok
            if lb is not None:
                f.out.append(lb)
    return bytes(f.out)


def enc_smsr00(ops):
    """SMSR00.DecompressHeaderless  AuroraLib.Compression.Nintendo/Nintendo/SMSR00.cs:85-131: a code section of big-endian u16 words -- 16-bit
    masks consumed from bit 15 down (1 = literal) and, interleaved in consumption order, one word per match: (length - 3) << 12 |
    (distance - 1) -- followed by the literal section.  Returns (stream, aux0 = length of the code section)."""
    codes, lits, mpos, nbits = bytearray(), bytearray(), None, 0
    for t in tokens(ops):
        if nbits == 0:
            mpos = len(codes); codes += bytes(2); nbits = 16       # :101-105
        k = nbits - 1
        if t[0] == "lit":
            codes[mpos + (1 - k // 8)] |= 1 << (k % 8)            # :107-110
            lits.append(t[1])
        else:
            _, d, n = t
            assert 3 <= n <= 18 and 1 <= d <= 4096
            v = ((n - 3) << 12) | (d - 1)                         # :113-117
            codes += bytes([v >> 8, v & 0xFF])
        nbits -= 1
    return bytes(codes) + bytes(lits), len(codes)


def enc_fastlz(ops, level):
    """FastLZ.DecompressHeaderless  AuroraLib.Compression/Formats/Common/FastLZ.cs:63-160: control bytes; below 32: a run of ctrl + 1 literals
    (the FIRST control byte carries level - 1 in its top three bits, :66, :74); otherwise a match: (ctrl >> 5) - 1 = length - 3 (6 = extended),
    13-bit distance - 1.  Level 1: one extension byte.  Level 2: extension bytes chained while they are 255 (:118-126), and a distance field
    of 0x1FFF announces two more big-endian bytes holding distance - 1 - 0x1FFF (:131-137)."""
    out, first = bytearray(), True
    for op in ops:
        if op[0] == "lit":
            lit = bytes(op[1])
            for i in range(0, len(lit), 32):
                run = lit[i:i + 32]
                out.append((len(run) - 1) | (((level - 1) << 5) if first else 0)); first = False
                out += run
        else:
            assert not first                                       # a stream starts with literals
            _, d, n = op
            length, dist = n - 3, d - 1
            assert length >= 0
            sd = min(dist, 0x1FFF) if level == 2 else dist
            assert sd <= 0x1FFF and (level == 2 or length <= 6 + 255)
            out.append(((min(length, 6) + 1) << 5) | (sd >> 8))
            if length >= 6:
                rest = length - 6
                while level == 2 and rest >= 255:
                    out.append(255); rest -= 255
                out.append(rest)
            out.append(sd & 0xFF)
            if level == 2 and dist >= 0x1FFF:
                ext = dist - 0x1FFF
                assert ext <= 0xFFFF
                out += bytes([ext >> 8, ext & 0xFF])
    return bytes(out)


def enc_cns(ops):
    """CNS.DecompressHeaderless  AuroraLib.Compression-Extended/Specialized/CNS.cs:66-98: a length byte; bit 7 clear: that many literals follow
    (0..127); bit 7 set: (byte & 0x7F) + 3 bytes from distance next byte + 1 (1..256)."""
    out = bytearray()
    for op in ops:
        if op[0] == "lit":
            lit = bytes(op[1])
            for i in range(0, len(lit), 127):
                run = lit[i:i + 127]
                out.append(len(run)); out += run                  # :79-82
        else:
            _, d, n = op
            assert 3 <= n <= 130 and 1 <= d <= 256
            out += bytes([0x80 | (n - 3), d - 1])                 # :83-90
    return bytes(out)


def enc_wflz(ops, big):
    """WFLZ.DecompressHeaderless  AuroraLib.Compression-Extended/WayForward/WFLZ.cs:82-112: blocks of u16 distance (byte order of the file), length
    byte (+ 4 = 5..259; 0 = no match), literal count byte, then the literals; a block of zero length and zero literals ends the stream."""
    out, i, ops = bytearray(), 0, list(ops)
    def block(d, n, lit):
        out.extend([(d >> 8) & 0xFF, d & 0xFF] if big else [d & 0xFF, (d >> 8) & 0xFF])
        out.append(n - 4 if n else 0); out.append(len(lit)); out.extend(lit)
    while i < len(ops):
        d = n = 0
        if ops[i][0] == "copy":
            _, d, n = ops[i]; i += 1
            assert 5 <= n <= 259 and 1 <= d <= 0xFFFF
        lit = b""
        if i < len(ops) and ops[i][0] == "lit":
            lit = bytes(ops[i][1]); i += 1
        first = lit[:255]
        block(d, n, first)                                        # :95-107
        for k in range(255, len(lit), 255):
            block(0, 0, lit[k:k + 255])                           # literals only: length 0, count != 0
    block(0, 0, b"")                                              # :100-103
    return bytes(out)


def enc_cnx2(ops):
    """CNX2.DecompressHeaderless  AuroraLib.Compression.Sega/Sega/CNX2.cs:89-127: 2-bit codes, read LSB first from 8-bit flags (FlagReader
    (source, Endian.Little) + ReadInt(2): the first bit read is bit 0 of the code).  1: one literal.  3: a count byte and that many
    literals.  2: a big-endian u16, distance - 1 in its top 11 bits, length - 4 in its low 5.  0: a count byte, that many bytes are
    SKIPPED, and the rest of the flag byte is dropped (flag.Reset())."""
    f = Flags8(msb_first=False)
    def code(c):
        f.bit(c & 1); f.bit(c >> 1)                               # FlagReader.ReadInt :75-87
    for op in ops:
        if op[0] == "lit":
            lit = bytes(op[1])
            if len(op) > 2 and op[2] == "run":                    # :117-120
                assert len(lit) <= 255
                code(3); f.out.append(len(lit)); f.out += lit
            else:
                for b in lit:
                    code(1); f.out.append(b)                      # :104-106
        elif op[0] == "skip":                                     # :97-101
            code(0); f.out.append(len(op[1])); f.out += bytes(op[1]); f.n = 0
        else:
            _, d, n = op[:3]
            assert 4 <= n <= 35 and 1 <= d <= 2048
            v = ((d - 1) << 5) | (n - 4)                          # :109-114
            code(2); f.out += bytes([v >> 8, v & 0xFF])
    return bytes(f.out)


def enc_refpack(ops):
    """RefPack.DecompressHeaderless  AuroraLib.Compression-Extended/EA/RefPack.cs:190-245: every command first copies 0-3 literals (P) and then a
    match.  0DDLLLPP D: length 3-10, distance 1-1024.  10LLLLLL PPDDDDDD D: 4-67, 1-16 384.  110DLLPP D D L: 5-1 028, 1-131 072.
    111PPPPP: 4 x (P + 1) literals (4-112), no match; 111111PP: P literals and the END.  A copy op may name its form ("short", "medium",
    "long"); literal runs in front of a match are written as 4-112 blocks with the last 0-3 bytes riding on the match command."""
    out, pending = bytearray(), b""
    def flush_blocks():
        nonlocal pending
        while len(pending) >= 4:
            n = min(len(pending) // 4 * 4, 112)
            out.append(0xE0 | (n // 4 - 1)); out.extend(pending[:n]); pending = pending[n:]     # :229-232, :240
    for op in ops:
        if op[0] == "lit":
            pending += bytes(op[1])
            continue
        _, d, n = op[:3]
        form = op[3] if len(op) > 3 else ("short" if 3 <= n <= 10 and d <= 1024 else "medium" if 4 <= n <= 67 and d <= 16384 else "long")
        flush_blocks()
        p = len(pending); assert p <= 3
        if form == "short":
            assert 3 <= n <= 10 and 1 <= d <= 1024
            out += bytes([(((d - 1) >> 8) << 5) | ((n - 3) << 2) | p, (d - 1) & 0xFF])            # :200-207
        elif form == "medium":
            assert 4 <= n <= 67 and 1 <= d <= 16384
            out += bytes([0x80 | (n - 4), (p << 6) | ((d - 1) >> 8), (d - 1) & 0xFF])             # :208-217
        else:
            assert 5 <= n <= 1028 and 1 <= d <= 131072
            out += bytes([0xC0 | (((d - 1) >> 16) << 4) | (((n - 5) >> 8) << 2) | p, ((d - 1) >> 8) & 0xFF, (d - 1) & 0xFF, (n - 5) & 0xFF])   # :218-228
        out.extend(pending); pending = b""
    flush_blocks()
    out.append(0xFC | len(pending)); out.extend(pending)                                           # :233-239
    return bytes(out)


def enc_lzshrek(ops):
    """LZShrek.DecompressHeaderless  AuroraLib.Compression-Extended/Activision/LZShrek.cs:72-119, ReadDistance :176-192: groups of a header byte --
    (matches in the group - 1) in its low 3 bits, the literal count in its top 5 (0-29; 30: + one byte; 31: 286 + a little-endian u16) --
    the literals, then 1-8 match commands: length 1-7 in the low 3 bits (0: the next byte + 7, and a next byte of 0 ENDS the stream),
    distance - 1 in the top 5 bits with the same escapes."""
    def dist_field(v):                                            # ReadDistance: 0-29 inline, 30-285 one byte, 286-65 821 two
        if v < 30:
            return v, b""
        if v < 286:
            return 30, bytes([v - 30])
        assert v <= 286 + 0xFFFF
        return 31, bytes([(v - 286) & 0xFF, (v - 286) >> 8])
    def match_cmd(d, n):
        f, ext = dist_field(d - 1)
        if 1 <= n <= 7:
            return bytes([(f << 3) | n]) + ext                    # :96-97, :113
        assert 8 <= n <= 262
        return bytes([f << 3, n - 7]) + ext                       # :99-110: the length byte comes BEFORE the distance bytes
    out, i, ops = bytearray(), 0, list(ops)
    while i < len(ops):
        lit = b""
        if ops[i][0] == "lit":
            lit = bytes(ops[i][1]); i += 1
        ms = []
        while i < len(ops) and ops[i][0] == "copy" and len(ms) < 8:
            ms.append(ops[i]); i += 1
        last = i >= len(ops)
        cmds = [match_cmd(m[1], m[2]) for m in ms]
        if last and len(cmds) < 8:
            cmds.append(bytes([0, 0])); last = False              # the end marker rides as one more command of this group  :101-108
            done = True
        else:
            done = False
        assert cmds
        f, ext = dist_field(len(lit))
        out += bytes([(f << 3) | (len(cmds) - 1)]) + ext + lit + b"".join(cmds)                    # :84-92
        if i >= len(ops) and not done:
            out += bytes([0, 0, 0])                               # a group of its own for the end marker: no literals, one command = END
            break
    return bytes(out)


def enc_hig(ops):
    """HIG.DecompressHeaderless  AuroraLib.Compression-Extended/Specialized/HIG.cs:91-164: a raw block, then (match, raw block) pairs until the size
    is reached.  Raw block in front: a byte n - 2 (3-257 literals) or 0 + a little-endian u16.  Match, by its first byte's top 3 bits:
    0-5: LLLDDDPP D, length 4-9, 11-bit distance; 6: 110LLLLL DDDDDDPP D, length 4-35, 14 bits; 7: 111DLLLL [ext] DDDDDDPP D, length 4-18
    from the nibble, nibble 0 = next byte + 18 (19-273), that byte 0 = a big-endian u16; 15 bits.  PP after a match: 1 / 2 literals,
    3 none, 0 another counted raw block.  The distance field is the distance itself (no + 1)."""
    ops = list(ops)
    out = bytearray()
    def raw(lit):                                                 # :100-103, :146-150
        n = len(lit)
        if 3 <= n <= 257:
            out.append(n - 2)
        else:
            out.append(0); out.extend([n & 0xFF, n >> 8])
        out.extend(lit)
    assert ops[0][0] == "lit"
    raw(bytes(ops[0][1])); i = 1
    while i < len(ops):
        _, d, n = ops[i][:3]
        form = ops[i][3] if len(ops[i]) > 3 else None
        i += 1
        lit = b""
        if i < len(ops) and ops[i][0] == "lit":
            lit = bytes(ops[i][1]); i += 1
        pp = 3 if len(lit) == 0 else (len(lit) if len(lit) <= 2 else 0)
        assert 1 <= d <= 0x7FFF
        if form is None:
            form = "A" if 4 <= n <= 9 and d <= 2047 else "B" if 4 <= n <= 35 and d <= 16383 else "C"
        if form == "A":                                           # :111-116
            assert 4 <= n <= 9 and d <= 2047
            out += bytes([((n - 4) << 5) | ((d >> 8) << 2) | pp, d & 0xFF])
        elif form == "B":                                         # :119-124, :141-145
            assert 4 <= n <= 35 and d <= 16383
            out += bytes([0xC0 | (n - 4), ((d >> 8) << 2) | pp, d & 0xFF])
        else:                                                     # :125-145
            hi = (d >> 14) << 4
            if 4 <= n <= 18 and form != "C16" and form != "C8":
                out.append(0xE0 | hi | (n - 3))
            elif 19 <= n <= 273 and form != "C16":
                out += bytes([0xE0 | hi, n - 18])
            else:
                assert n <= 0xFFFF
                out += bytes([0xE0 | hi, 0, n >> 8, n & 0xFF])
            out += bytes([(((d >> 8) & 0x3F) << 2) | pp, d & 0xFF])
        if pp == 0:
            raw(lit)
        else:
            out.extend(lit)
    return bytes(out)


# ------------------------------------------------------------------------------------------------ PRS
class LazyFlags:
    """FlagReader over the SAME stream as the data (PRS.cs:62): a flag byte sits wherever the decoder happens to be when it
    needs a bit and has none left -- possibly between the bits of one token.  Bit order = byte order (PRS.cs:62)."""

    def __init__(self, big):
        self.big, self.out, self.pos, self.n = big, bytearray(), None, 0

    def bit(self, v):
        if self.n == 0:
            self.pos = len(self.out); self.out.append(0); self.n = 8
        shift = self.n - 1 if self.big else 8 - self.n
        self.out[self.pos] |= (1 if v else 0) << shift
        self.n -= 1


def enc_prs(ops, big, terminate=True):
    """PRS.DecompressHeaderless  AuroraLib.Compression.Sega/Sega/PRS.cs:59-102"""
    f = LazyFlags(big)
    u16 = (lambda v: bytes([v >> 8, v & 0xFF])) if big else (lambda v: bytes([v & 0xFF, v >> 8]))
    for t in tokens(ops):
        if t[0] == "lit":
            f.bit(1); f.out.append(t[1])                          # :66-69
        else:
            d, n = t[1], t[2]
            if d <= 0x100 and 2 <= n <= 5 and not (len(t) > 3 and t[3] in ("long", "ext")):
                f.bit(0); f.bit(0)                                # :93-97 short: ReadInt(2, reverse) = first bit is the high one
                f.bit((n - 2) >> 1); f.bit((n - 2) & 1)
                f.out.append((0x100 - d) & 0xFF)
            else:
                assert 1 <= d <= 0x1FFF and 1 <= n <= 0x100
                f.bit(0); f.bit(1)                                # :73-91 long
                if 3 <= n <= 9 and not (len(t) > 3 and t[3] == "ext"):
                    f.out += u16(((0x2000 - d) << 3) | (n - 2))   # :82-90 n = (v & 7) + 2
                else:
                    f.out += u16((0x2000 - d) << 3); f.out.append(n - 1)   # :84-86 n = ReadUInt8() + 1
    if terminate:
        f.bit(0); f.bit(1); f.out += b"\x00\x00"                  # :75-80
    return bytes(f.out)


# ------------------------------------------------------------------------------------------------ LZ4 / LZO / Snappy
def lz4_ext(v):
    """ReadExtension  LZ4.cs:241-252: 15 in the nibble, then bytes of 255 and one byte < 255"""
    if v < 15:
        return b""
    v -= 15
    out = bytearray()
    while v >= 255:
        out.append(255); v -= 255
    out.append(v)
    return bytes(out)


def enc_lz4(ops):
    """LZ4.DecompressBlockHeaderless  AuroraLib.Compression/Formats/Common/LZ4.cs:176-200: sequences of (literals, match); the
    block ends with a literals-only sequence (:190)."""
    out, i = bytearray(), 0
    ops = list(ops)
    while i < len(ops):
        lit = b""
        if ops[i][0] == "lit":
            lit = bytes(ops[i][1]); i += 1
        if i < len(ops):
            _, d, n = ops[i]; i += 1
            assert n >= 4 and 1 <= d <= 0xFFFF
            out.append((min(len(lit), 15) << 4) | min(n - 4, 15))
            out += lz4_ext(len(lit)) + lit + bytes([d & 0xFF, d >> 8]) + lz4_ext(n - 4)
            if i == len(ops):                                     # a block cannot end behind a match: empty last sequence
                out.append(0)
        else:
            out.append(min(len(lit), 15) << 4); out += lz4_ext(len(lit)) + lit
    return bytes(out)


def lzo_ext(v):
    """ReadExtendedInt  LZO.cs:252-262: zero bytes count 255 each, then one non-zero byte"""
    out = bytearray()
    while v > 255:
        out.append(0); v -= 255
    assert v >= 1
    out.append(v)
    return bytes(out)


def enc_lzo(ops, first_run_quirk=False):
    """LZO.DecompressHeaderless  AuroraLib.Compression/Formats/Common/LZO.cs:49-139.  ops: ("lit", bytes) runs and
    ("copy", d, n, form) with form in M1a (2 bytes, after 1-3 literals), M1b (3 bytes, after a run), M2, M2b, M3, M4.  The
    0-3 literals that FOLLOW a match ride in the low bits of its last header byte (`plain = flag & 3` :131)."""
    out, ops = bytearray(), list(ops)
    i = 0
    if first_run_quirk:                                           # :59-64 first byte > 17: that many - 17 literals, plain stays 0
        lit = bytes(ops[0][1]); assert 1 <= len(lit) <= 238
        out.append(17 + len(lit)); out += lit; i = 1
    while i < len(ops):
        op = ops[i]
        if op[0] == "lit":                                        # state plain == 0: a literal run of 4.. bytes  :75-85
            lit = bytes(op[1]); assert len(lit) >= 4
            if len(lit) <= 18:
                out.append(len(lit) - 3)
            else:
                out.append(0); out += lzo_ext(len(lit) - 18)
            out += lit; i += 1
            continue
        _, d, n, form = op
        trail = b""
        if i + 1 < len(ops) and ops[i + 1][0] == "lit" and len(ops[i + 1][1]) <= 3 and not (len(ops[i + 1]) > 2 and ops[i + 1][2] == "run"):
            trail = bytes(ops[i + 1][1]); i += 1
        p = len(trail)
        if form == "M1a":                                         # :86-91 after 1-3 literals: length 2, distance 1..1024
            assert n == 2 and 1 <= d <= 1024
            out += bytes([(((d - 1) & 3) << 2) | p, (d - 1) >> 2])
        elif form == "M1b":                                       # :92-97 after a literal run: length 3, distance 2049..3072
            assert n == 3 and 2049 <= d <= 3072
            out += bytes([(((d - 2049) & 3) << 2) | p, (d - 2049) >> 2])
        elif form == "M2":                                        # :122-127 01LD DDPP: length 3-4, distance 1..2048
            assert n in (3, 4) and 1 <= d <= 2048
            out += bytes([0x40 | ((n - 3) << 5) | (((d - 1) & 7) << 2) | p, (d - 1) >> 3])
        elif form == "M2b":                                       # :128-133 1LLD DDPP: length 5-8
            assert 5 <= n <= 8 and 1 <= d <= 2048
            out += bytes([0x80 | ((n - 5) << 5) | (((d - 1) & 7) << 2) | p, (d - 1) >> 3])
        elif form == "M3":                                        # :111-120 001L LLLL: length 3-33 or 33 + ext, distance 1..16384
            assert n >= 3 and 1 <= d <= 16384
            if n <= 33:
                out.append(0x20 | (n - 2))
            else:
                out.append(0x20); out += lzo_ext(n - 33)
            out += bytes([(((d - 1) & 0x3F) << 2) | p, (d - 1) >> 6])
        else:                                                     # M4 :98-110 0001 HLLL: length 3-9 or 9 + ext, distance 16385..49151
            assert form == "M4" and n >= 3 and 16385 <= d <= 49151
            h = (d - 16384) >> 14
            if n <= 9:
                out.append(0x10 | (h << 3) | (n - 2))
            else:
                out.append(0x10 | (h << 3)); out += lzo_ext(n - 9)
            low = (d - 16384) & 0x3FFF
            out += bytes([((low & 0x3F) << 2) | p, low >> 6])
        out += trail
        i += 1
    out += bytes([0x11, 0x00, 0x00])                              # :107-109 end marker: code 1, distance 16384
    return bytes(out)


def enc_snappy(ops):
    """Snappy.DecompressHeaderless  AuroraLib.Compression/Formats/Common/Snappy.cs:205-250, varint size :109-122"""
    total = len(expand(ops))
    out, v = bytearray(), total
    while True:
        b = v & 0x7F; v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            break
    for op in ops:
        if op[0] == "lit":
            lit = bytes(op[1]); n = len(lit)
            if n <= 60:
                out.append((n - 1) << 2)                          # :221-233 tag >> 2 < 60: that + 1 literals
            elif n <= 256:
                out += bytes([60 << 2, n - 1])                    # 60: one length byte
            else:
                out += bytes([61 << 2, (n - 1) & 0xFF, (n - 1) >> 8])   # 61: two length bytes, little endian
            out += lit
        else:
            _, d, n = op[:3]
            form = op[3] if len(op) > 3 else None
            if form != "copy2" and 4 <= n <= 11 and d < 2048:
                out += bytes([1 | ((n - 4) << 2) | ((d >> 8) << 5), d & 0xFF])   # :235-239
            else:
                assert 1 <= n <= 64 and d <= 0xFFFF
                out += bytes([2 | ((n - 1) << 2), d & 0xFF, d >> 8])             # :240-243
    return bytes(out)


# ------------------------------------------------------------------------------------------------ the cases
def pat(n, seed):
    """n distinct-looking bytes (no generator state: a formula)"""
    return bytes((seed * 37 + i * 29 + (i * i) // 7) & 0xFF for i in range(n))


def case(name, fmt, src, ops, cite, lz=None, aux0=0, aux1=0, decom_len=None, src_used=None, note=""):
    exp = expand(ops)
    # expected bytes: zlib + base64 (the long cases are repetitive by construction: 60 KiB of output in 200 bytes of JSON)
    c = {"name": name, "format": fmt, "src": src.hex(), "expect_len": len(exp), "expect_zlib_b64": base64.b64encode(zlib.compress(exp, 9)).decode(),
         "decom_len": len(exp) if decom_len is None else decom_len,
         "aux0": aux0, "aux1": aux1, "cite": cite}
    if lz:
        c["lz"] = lz
    if src_used is not None:
        c["src_used"] = src_used
    if note:
        c["note"] = note
    return c


def build():
    K = {}
    A, B, C3 = pat(9, 1), pat(20, 2), pat(5, 3)

    # ---- LZSS (12, 4, 2): lengths 3..18, distances 1..4096
    ops = [("lit", A), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x00\xff"), ("copy", 3, 7), ("copy", 30, 18), ("lit", b"Z")]
    K.setdefault("lzss", []).append(case("default geometry: min / max length, distance 1 (RLE) and 3, partial last flag byte", "lzss", enc_lzss(ops), ops, "LZSS.cs:91-130"))
    big = [("lit", pat(60, 7))] + [("copy", 60, 18)] * 226 + [("lit", b"\x01\x02\x03\x04")] + [("copy", 4096, 18), ("copy", 4095, 3), ("copy", 2048, 5)]
    K["lzss"].append(case("distance 4096 = the whole window (ring address == write position), 4095, across the 0xFEE ring origin", "lzss", enc_lzss(big), big, "LZSS.cs:115-119, LzWindows.cs:108-115"))
    ops = [("lit", B), ("copy", 20, 66), ("copy", 1, 3), ("lit", b"q"), ("copy", 64, 40)]
    K["lzss"].append(case("KAT geometry LzProperties((byte)10, 6, 2): W 1024, lengths 3..66, WindowsStart 958", "lzss", enc_lzss(ops, 10, 6, 2), ops,
                          "CompressionTest/CompressionAlgorithmTest.cs:31-48, LzProperties.cs:57-66", lz={"window_bits": 10, "length_bits": 6, "min_length": 3, "windows_start": 958, "max_distance": 1024}))
    # ---- LZ10
    ops = [("lit", A), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x10\x20"), ("copy", 2, 9), ("copy", 33, 17)]
    K.setdefault("lz10", []).append(case("lengths 3 / 9 / 17 / 18, distances 1, 2 and beyond one flag group", "lz10", enc_lz10(ops), ops, "LZ10.cs:82-111"))
    big = [("lit", pat(64, 9))] + [("copy", 64, 18)] * 224 + [("copy", 4096, 18), ("copy", 4095, 3), ("lit", b"end")]
    K["lz10"].append(case("distance 4096 (12 bits all set) and 4095", "lz10", enc_lz10(big), big, "LZ10.cs:94-98"))
    ops = [("lit", b"\xAA")] + [("copy", 1, 3)] * 8 + [("lit", b"\xBB")]
    K["lz10"].append(case("a flag byte of eight matches (0xFF) followed by a one-literal group (0x00)", "lz10", enc_lz10(ops), ops, "LZ10.cs:88-102"))
    # ---- LZ11
    ops = [("lit", A), ("copy", 9, 3), ("copy", 4, 16), ("copy", 1, 17), ("copy", 7, 272), ("lit", b"\x7f"), ("copy", 100, 273), ("copy", 300, 1000)]
    K.setdefault("lz11", []).append(case("2-byte form 3 / 16, 3-byte form 17 / 272, 4-byte form 273 / 1000", "lz11", enc_lz11(ops), ops, "LZ11.cs:83-133"))
    ops = [("lit", pat(16, 4)), ("copy", 16, 65808), ("lit", b"!")]
    K["lz11"].append(case("the longest token: 4-byte form with all length bits set (65 808 bytes)", "lz11", enc_lz11(ops), ops, "LZ11.cs:105-112"))
    big = [("lit", pat(50, 5)), ("copy", 50, 4046), ("copy", 4096, 273), ("copy", 4095, 17), ("copy", 4096, 3)]
    K["lz11"].append(case("distance 4096 in each of the three forms' distance fields", "lz11", enc_lz11(big), big, "LZ11.cs:98-118"))
    # ---- Yaz0
    ops = [("lit", A), ("copy", 9, 3), ("copy", 5, 17), ("copy", 1, 18), ("copy", 2, 273), ("lit", b"\x00"), ("copy", 21, 100)]
    K.setdefault("yaz0", []).append(case("2-byte form 3 / 17, 3-byte form 18 (length byte 0) / 273 (0xFF) / 100", "yaz0", enc_yaz0(ops), ops, "Yay0.cs:110-144, Yaz0.cs:91-92"))
    big = [("lit", pat(40, 6)), ("copy", 40, 273)] + [("copy", 313, 273)] * 14 + [("copy", 4096, 39), ("copy", 4095, 3), ("lit", b"ok")]
    K["yaz0"].append(case("distance 4096 and 4095", "yaz0", enc_yaz0(big), big, "Yay0.cs:127"))
    ops = [("lit", C3), ("copy", 5, 17)]
    K["yaz0"].append(case("ReadByte() == -1 at the end of the input: a 3-byte token whose length byte is missing copies 0x12 - 1 = 17 bytes",
                          "yaz0", enc_yaz0([("lit", C3), ("copy", 5, 18)], drop_last_length_byte=True), ops, "Yay0.cs:130-131"))
    # ---- Yay0 / MIO0
    ops = [("lit", A), ("copy", 9, 3), ("copy", 5, 17), ("copy", 1, 18), ("lit", b"\x42\x43"), ("copy", 2, 273), ("copy", 30, 60), ("lit", b"\x99")]
    s, a0, a1 = enc_3cursor(ops, mio0=False)
    K.setdefault("yay0", []).append(case("three sections; the length bytes of the 3-byte form interleave with the literals", "yay0", s, ops, "Yay0.cs:99-144", aux0=a0, aux1=a1, src_used=len(s)))
    big = [("lit", pat(33, 8)), ("copy", 33, 273)] + [("copy", 306, 273)] * 14 + [("copy", 4096, 19), ("copy", 4095, 4)]
    s, a0, a1 = enc_3cursor(big, mio0=False)
    K["yay0"].append(case("distance 4096 / 4095; two flag bytes padded to a 4-byte flag section", "yay0", s, big, "Yay0.cs:127", aux0=a0, aux1=a1, src_used=len(s)))
    ops = [("lit", b"\x01")] + [("copy", 1, 3)] * 31 + [("lit", b"\x02")]
    s, a0, a1 = enc_3cursor(ops, mio0=False)
    K["yay0"].append(case("33 tokens: five flag bytes, the last one with a single bit used", "yay0", s, ops, "Yay0.cs:113-121", aux0=a0, aux1=a1, src_used=len(s)))
    ops = [("lit", A), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x42"), ("copy", 2, 10), ("copy", 12, 18)]
    s, a0, a1 = enc_3cursor(ops, mio0=True)
    K.setdefault("mio0", []).append(case("lengths 3 / 10 / 18", "mio0", s, ops, "MIO0.cs:105-149", aux0=a0, aux1=a1, src_used=len(s)))
    big = [("lit", pat(64, 10))] + [("copy", 64, 18)] * 224 + [("copy", 4096, 18), ("copy", 4095, 3)]
    s, a0, a1 = enc_3cursor(big, mio0=True)
    K["mio0"].append(case("distance 4096 / 4095", "mio0", s, big, "MIO0.cs:129-136", aux0=a0, aux1=a1, src_used=len(s)))
    ops = [("lit", b"\xEE")] + [("copy", 1, 3)] * 8
    s, a0, a1 = enc_3cursor(ops, mio0=True)
    K["mio0"].append(case("nine tokens: a full flag byte and one bit of the next", "mio0", s, ops, "MIO0.cs:117-123", aux0=a0, aux1=a1, src_used=len(s)))
    # ---- PRS, both bit orders
    for big_e, fmt in ((True, "prs_be"), (False, "prs_le")):
        ops = [("lit", A), ("copy", 9, 2), ("copy", 1, 5), ("copy", 2, 256), ("copy", 13, 3), ("lit", b"\x31\x32\x33\x34"), ("copy", 4, 3, "long"), ("copy", 270, 9), ("copy", 7, 1, "ext"), ("copy", 11, 6, "ext")]
        K.setdefault(fmt, []).append(case("short matches 2 / 5 / 3, long matches 3 (forced long form) / 9, length byte forms 256 and 1; flag bytes land inside tokens",
                                          fmt, enc_prs(ops, big_e), ops, "PRS.cs:59-102, FlagReader.cs:52-100", decom_len=0))
        far = [("lit", pat(70, 11))] + [("copy", 70, 256)] * 32 + [("copy", 0x1FFF, 9), ("copy", 0x100, 5), ("copy", 0x101, 3), ("lit", b"\xfe")]
        K[fmt].append(case("distance 0x1FFF (13 bits), 0x100 (short form, offset byte 0) and 0x101 (first long-only distance)", fmt, enc_prs(far, big_e), far,
                           "PRS.cs:82-97", decom_len=0))
        ops = [("lit", b"ab"), ("copy", 2, 4)] * 5 + [("lit", b"\x00")]
        K[fmt].append(case("the 2-bit length of a short match split across two flag bytes", fmt, enc_prs(ops, big_e), ops, "PRS.cs:95, FlagReader.cs:75-100", decom_len=0))
    # ---- LZ4 block
    ops = [("lit", A), ("copy", 9, 4), ("copy", 1, 18), ("lit", pat(15, 12)), ("copy", 3, 19), ("lit", pat(270, 13)), ("copy", 100, 274), ("copy", 300, 530), ("lit", b"tail!")]
    K.setdefault("lz4_block", []).append(case("literal counts 9 / 0 / 15 (extension byte 0) / 270 (255 + 0); match lengths 4 / 18 / 19 (ext 0) / 274 (255 + 0) / 530 (255, 255, 1)",
                                              "lz4_block", enc_lz4(ops), ops, "LZ4.cs:176-200, :241-252", decom_len=0))
    far = [("lit", pat(80, 14))] + [("copy", 80, 1000)] * 66 + [("copy", 0xFFFF, 40), ("copy", 0x8000, 4), ("lit", b"12345")]
    K["lz4_block"].append(case("distance 0xFFFF and 0x8000 (u16 little endian)", "lz4_block", enc_lz4(far), far, "LZ4.cs:195", decom_len=0))
    ops = [("lit", b"x"), ("copy", 1, 4)]
    K["lz4_block"].append(case("a block that ends right behind a match: the input is exhausted at the next token byte", "lz4_block", enc_lz4(ops)[:-1], ops, "LZ4.cs:180, :190", decom_len=0))
    # ---- LZO
    ops = [("lit", pat(3100, 16)), ("copy", 2049, 3, "M1b"), ("lit", b"\x01"), ("copy", 1, 2, "M1a"), ("lit", b"\x02\x03"), ("copy", 1024, 2, "M1a"), ("lit", b"\x04\x05\x06"),
           ("copy", 5, 3, "M2"), ("copy", 2048, 4, "M2"), ("lit", pat(18, 17), "run"), ("copy", 3072, 3, "M1b"), ("copy", 9, 8, "M2b"), ("copy", 2048, 5, "M2b"),
           ("copy", 1, 33, "M3"), ("copy", 3000, 34, "M3"), ("copy", 100, 600, "M3"), ("lit", pat(19, 18), "run"), ("copy", 7, 3, "M3")]
    K.setdefault("lzo", []).append(case("every opcode class below 16 KiB: literal runs 3100 (18 + 255 x 12 + 22) / 18 / 19, M1 after 1-3 literals (length 2) and after a run (length 3, "
                                        "distance 2049 / 3072), M2 3 / 4, M2b 5 / 8, M3 3 / 33 / 34 / 600, trailing literals 0-3 in the low bits", "lzo", enc_lzo(ops), ops, "LZO.cs:49-139, :252-262", decom_len=0))
    far = [("lit", pat(200, 19))] + [("copy", 200, 2000, "M3")] * 25 + [("copy", 16384, 34, "M3"), ("copy", 16385, 3, "M4"), ("copy", 49151, 9, "M4"), ("lit", b"\x09"), ("copy", 32768, 10, "M4"), ("copy", 20000, 700, "M4")]
    K["lzo"].append(case("M3 at its largest distance 16384; M4: distances 16385 / 32768 / 49151 (the H bit), lengths 3 / 9 / 10 / 700", "lzo", enc_lzo(far), far, "LZO.cs:98-110", decom_len=0))
    ops = [("lit", pat(5, 20)), ("lit", pat(4, 21)), ("copy", 3, 4, "M2")]
    K["lzo"].append(case("first byte > 17: an initial run of (byte - 17) literals that leaves plain at 0, so the next opcode < 16 is a literal RUN (canonical LZO1X would read a match)",
                         "lzo", enc_lzo(ops, first_run_quirk=True), [("lit", pat(5, 20) + pat(4, 21)), ("copy", 3, 4)], "LZO.cs:51, :59-64, :75-85", decom_len=0))
    # ---- Snappy
    ops = [("lit", A), ("copy", 9, 4), ("copy", 1, 11), ("lit", pat(60, 22)), ("copy", 2047, 5), ("lit", pat(61, 23)), ("copy", 70, 4, "copy2"), ("copy", 3, 64), ("copy", 5, 1, "copy2"), ("lit", pat(300, 24)), ("copy", 400, 12)]
    ops[4] = ("copy", 60, 5)
    K.setdefault("snappy_raw", []).append(case("literal tags 9 / 60 (largest inline) / 61 (one length byte) / 300 (two length bytes); copy-1 lengths 4 / 11; copy-2 lengths 4 / 64 / 1 / 12",
                                               "snappy_raw", enc_snappy(ops), ops, "Snappy.cs:205-250, :109-122", decom_len=0))
    far = [("lit", pat(90, 25))] + [("copy", 90, 64)] * 1030 + [("copy", 0xFFFF, 64), ("copy", 2047, 11), ("copy", 2048, 11), ("lit", b"z")]
    K["snappy_raw"].append(case("distance 0xFFFF (copy-2), 2047 (largest copy-1) and 2048; a three-byte varint size", "snappy_raw", enc_snappy(far), far, "Snappy.cs:235-243", decom_len=0))
    ops = [("lit", b"\x00")]
    K["snappy_raw"].append(case("the shortest stream: size 1, one literal", "snappy_raw", enc_snappy(ops), ops, "Snappy.cs:221-233", decom_len=0))
    # ---- the other flag-byte formats of the lane-parallel kernel family: CLZ0, LZ02, LZ40, LZHudson, SMSR00
    ops = [("lit", A), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x10\x20"), ("copy", 2, 9), ("copy", 33, 17)]
    K.setdefault("clz0", []).append(case("lengths 3 / 9 / 17 / 18, distances 1, 2 and 33; flags LSB first", "clz0", enc_clz0(ops), ops, "CLZ0.cs:66-100"))
    big = [("lit", pat(64, 9))] + [("copy", 64, 18)] * 224 + [("copy", 4096, 18), ("copy", 4095, 3), ("copy", 256, 4), ("copy", 255, 5), ("lit", b"end")]
    K["clz0"].append(case("distance 4096 (delta 0), 4095 (delta 1), 256 / 255 (the delta's low byte wraps)", "clz0", enc_clz0(big), big, "CLZ0.cs:78-83"))
    ops = [("lit", A), ("copy", 9, 2), ("copy", 1, 16), ("copy", 4, 17), ("copy", 7, 272), ("lit", b"\x7f"), ("copy", 255, 100), ("copy", 256, 3)]
    K.setdefault("lz02", []).append(case("2-byte form 2 / 16 / 3, 3-byte form 17 / 272 / 100, the terminator token", "lz02", enc_lz02(ops), ops, "LZ02.cs:84-121"))
    big = [("lit", pat(50, 5)), ("copy", 50, 272)] + [("copy", 322, 272)] * 14 + [("copy", 4095, 17), ("copy", 4096, 16), ("copy", 4096, 2), ("lit", b"ok")]
    K["lz02"].append(case("distance 4095 in the 3-byte form; a distance field of 0 under a non-zero length nibble = one whole window", "lz02", enc_lz02(big), big, "LZ02.cs:92-109, LzWindows.cs:72-100"))
    ops = [("lit", A), ("copy", 9, 3), ("copy", 1, 15), ("copy", 4, 16), ("copy", 7, 271), ("lit", b"\x7f"), ("copy", 100, 272), ("copy", 300, 1000), ("copy", 2, 2)]
    K.setdefault("lz40", []).append(case("2-byte form 3 / 15 / 2, 3-byte form 16 / 271, 4-byte form 272 / 1000; negated flag bytes", "lz40", enc_lz40(ops), ops, "LZ40.cs:84-132"))
    big = [("lit", pat(50, 5)), ("copy", 50, 4046), ("copy", 4096, 271), ("copy", 4095, 15), ("copy", 4096, 3), ("copy", 4096, 300)]
    K["lz40"].append(case("a distance field of 0 (= 4096) in each of the three forms, and 4095", "lz40", enc_lz40(big), big, "LZ40.cs:103-119"))
    ops = [("lit", A), ("copy", 9, 3), ("copy", 5, 17), ("copy", 1, 18), ("copy", 2, 273), ("lit", b"\x00"), ("copy", 21, 100)] + [("lit", pat(3, 70 + i)) if i % 3 else ("copy", 7, 4) for i in range(30)]
    K.setdefault("lzhudson", []).append(case("Yaz0's token forms behind two 32-bit big-endian flag words (the second one partial)", "lzhudson", enc_lzhudson(ops), ops, "LZHudson.cs:52-53, Yay0.cs:110-144"))
    big = [("lit", pat(40, 6)), ("copy", 40, 273)] + [("copy", 313, 273)] * 14 + [("copy", 4096, 39), ("copy", 4095, 3), ("lit", b"ok")]
    K["lzhudson"].append(case("distance 4096 and 4095", "lzhudson", enc_lzhudson(big), big, "Yay0.cs:127"))
    ops = [("lit", A), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x42"), ("copy", 2, 10), ("copy", 12, 18)] + [("lit", pat(2, 80 + i)) if i % 2 else ("copy", 5, 6) for i in range(20)]
    st, a0 = enc_smsr00(ops)
    K.setdefault("smsr00", []).append(case("masks of 16 tokens and the match words between them, literals in their own section (three masks, the last one partial)", "smsr00", st, ops, "SMSR00.cs:85-131", aux0=a0))
    big = [("lit", pat(64, 9))] + [("copy", 64, 18)] * 224 + [("copy", 4096, 18), ("copy", 4095, 3), ("lit", b"end")]
    st, a0 = enc_smsr00(big)
    K["smsr00"].append(case("distance 4096 (12 bits all set) and 4095", "smsr00", st, big, "SMSR00.cs:113-117", aux0=a0))
    # ---- FastLZ, both levels (level 2 is what the round-2 encoder work writes)
    ops = [("lit", pat(1, 90)), ("copy", 1, 3), ("lit", pat(32, 91)), ("copy", 33, 8), ("lit", pat(33, 92)), ("copy", 2, 264), ("copy", 256, 9), ("copy", 100, 100)]
    K.setdefault("fastlz", []).append(case("level 1: literal runs of 1 / 32 / 33 (two control bytes), match lengths 3 / 8 / 9 (extension byte 0) / 264 (255) / 100", "fastlz", enc_fastlz(ops, 1), ops, "FastLZ.cs:71-105", decom_len=0))
    far = [("lit", pat(64, 93))] + [("copy", 64, 264)] * 31 + [("copy", 8192, 5), ("copy", 8191, 3), ("lit", b"z")]
    K["fastlz"].append(case("level 1: distance 8192 (all 13 bits set) and 8191", "fastlz", enc_fastlz(far, 1), far, "FastLZ.cs:84-91", decom_len=0))
    ops = [("lit", pat(20, 94)), ("copy", 20, 3), ("copy", 1, 9), ("copy", 7, 263), ("copy", 7, 264), ("copy", 5, 265), ("copy", 3, 1000), ("lit", pat(40, 95)), ("copy", 40, 519)]
    K["fastlz"].append(case("level 2 (tag in the first control byte): length extension 0 / 254 / 255 + 0 / 255 + 1 / 255 x 3 + 226 / 255 x 2 + 0", "fastlz", enc_fastlz(ops, 2), ops, "FastLZ.cs:107-160", decom_len=0))
    far = [("lit", pat(64, 96))] + [("copy", 64, 1000)] * 74 + [("copy", 8191, 4), ("copy", 8192, 5), ("copy", 8193, 6), ("copy", 73727, 3), ("copy", 73727, 700), ("lit", b"!")]
    K["fastlz"].append(case("level 2: distance 8191 (short form), 8192 (field 0x1FFF + extension 0), 8193, and 73 727 = 0x1FFF + 0xFFFF + 1, the largest", "fastlz", enc_fastlz(far, 2), far, "FastLZ.cs:128-139", decom_len=0))
    # ---- CNS and WFLZ (token-queue kernels, one-byte / four-byte headers)
    ops = [("lit", pat(3, 100)), ("copy", 3, 3), ("copy", 1, 130), ("lit", pat(127, 101)), ("copy", 127, 64), ("lit", pat(128, 102)), ("copy", 256, 5), ("copy", 255, 4), ("lit", b"")]
    K.setdefault("cns", []).append(case("literal runs 3 / 127 / 128 (two headers) / 0, match lengths 3 / 130 / 64, distances 1 / 3 / 255 / 256 (one byte + 1)", "cns", enc_cns(ops), ops, "CNS.cs:66-98"))
    ops = [("lit", b"a"), ("copy", 1, 129), ("lit", b"b")]
    K["cns"].append(case("the shortest shapes: one literal, an overlapping match, one literal", "cns", enc_cns(ops), ops, "CNS.cs:79-90"))
    for big in (False, True):
        name = "wflz_be" if big else "wflz"
        ops = [("lit", pat(9, 110)), ("copy", 9, 5), ("copy", 1, 259), ("lit", pat(255, 111)), ("copy", 300, 100), ("lit", pat(256, 112)), ("copy", 2, 6)]
        K.setdefault(name, []).append(case("match lengths 5 / 259 / 100 / 6, literal counts 9 / 0 / 255 / 256 (a literal-only block follows) / 0, the end block", name, enc_wflz(ops, big), ops, "WFLZ.cs:82-112", decom_len=0))
        far = [("lit", pat(200, 113))] + [("copy", 200, 259)] * 253 + [("copy", 0xFFFF, 5), ("copy", 0x0100, 7), ("copy", 0x00FF, 9), ("lit", b"!")]
        K[name].append(case("distance 0xFFFF, 0x0100 and 0x00FF (the two distance bytes in the file's byte order)", name, enc_wflz(far, big), far, "WFLZ.cs:90-99", decom_len=0))
    # ---- CNX2 (2-bit codes)
    ops = [("lit", pat(5, 120)), ("copy", 5, 4), ("copy", 1, 35), ("lit", pat(255, 121), "run"), ("copy", 200, 20), ("lit", b"x", "run"), ("skip", b"\xde\xad\xbe"),
           ("lit", pat(3, 122)), ("copy", 2, 5), ("lit", b"", "run"), ("lit", b"z")]
    K.setdefault("cnx2", []).append(case("codes 1 / 2 / 3 / 0: single literals, runs of 255 / 1 / 0, match lengths 4 / 35 / 20 / 5, a 3-byte skip that also drops the rest of its flag byte",
                                         "cnx2", enc_cnx2(ops), ops, "CNX2.cs:89-127"))
    far = [("lit", pat(200, 123), "run")] + [("copy", 200, 35)] * 60 + [("copy", 2048, 4), ("copy", 2047, 35), ("copy", 33, 4), ("copy", 32, 4), ("lit", b"!")]
    K["cnx2"].append(case("distance 2048 (all 11 bits), 2047, 33 / 32 (the bit that crosses the pair's byte boundary)", "cnx2", enc_cnx2(far), far, "CNX2.cs:109-114"))
    # ---- RefPack and LZShrek
    ops = [("lit", pat(2, 130)), ("copy", 2, 3), ("copy", 1, 10), ("lit", pat(3, 131)), ("copy", 5, 4, "medium"), ("copy", 9, 67), ("lit", pat(4, 132)), ("copy", 11, 5, "long"),
           ("lit", pat(113, 133)), ("copy", 100, 1028), ("lit", pat(7, 134)), ("copy", 1024, 3), ("lit", b"end")]
    K.setdefault("refpack", []).append(case("short 3 / 10, medium 4 / 67, long 5 / 1 028; 0-3 literals on a command, blocks of 4 / 112, an end command with 3 literals", "refpack", enc_refpack(ops), ops, "RefPack.cs:190-245"))
    far = [("lit", pat(64, 135))] + [("copy", 64, 1028)] * 130 + [("copy", 1025, 4, "medium"), ("copy", 16384, 4), ("copy", 16385, 5), ("copy", 65536, 6, "long"), ("copy", 131072, 7), ("lit", b"")]
    K["refpack"].append(case("distance 1 025 (medium), 16 384, 16 385 (long), 65 536 and 131 072 (the 17th distance bit in the prefix); an end command without literals", "refpack", enc_refpack(far), far, "RefPack.cs:208-228"))
    ops = [("lit", pat(29, 140)), ("copy", 29, 1), ("copy", 1, 7), ("copy", 30, 8), ("lit", pat(30, 141)), ("copy", 31, 262), ("lit", pat(285, 142)), ("copy", 286, 100), ("lit", pat(286, 143)),
           ("copy", 287, 2), ("copy", 3, 3), ("copy", 3, 3), ("copy", 3, 3), ("copy", 3, 3), ("copy", 3, 3), ("copy", 3, 3), ("copy", 3, 3), ("copy", 2, 5), ("lit", b"z")]
    K.setdefault("lzshrek", []).append(case("literal counts 29 / 30 / 285 / 286 (the escapes), match lengths 1 / 7 / 8 / 262, distances 29-31 / 286 / 287, a full group of eight commands", "lzshrek", enc_lzshrek(ops), ops, "LZShrek.cs:72-119, :176-192"))
    far = [("lit", pat(100, 144))] + [("copy", 100, 262)] * 16 + [("copy", 4096, 9), ("copy", 4095, 3)]
    K["lzshrek"].append(case("distance 4096 and 4095 (two-byte escape); the stream ends on a match, the end marker is a group of its own", "lzshrek", enc_lzshrek(far), far, "LZShrek.cs:113, :183-188"))
    # ---- HIG
    ops = [("lit", pat(5, 150)), ("copy", 5, 4), ("lit", b"a"), ("copy", 1, 9), ("lit", b"bc"), ("copy", 3, 10), ("copy", 7, 35), ("lit", pat(3, 151)), ("copy", 20, 18, "C"), ("lit", pat(257, 152)),
           ("copy", 256, 19), ("lit", pat(258, 153)), ("copy", 100, 273), ("copy", 2, 274), ("copy", 9, 4, "C"), ("copy", 9, 1000)]
    K.setdefault("hig", []).append(case("forms A 4 / 9, B 10 / 35, C nibble 18 / 4, C byte 19 / 273, C u16 274 / 1 000; raw blocks after a match: 1, 2, none, 3 / 257 (one byte), 258 (u16)", "hig", enc_hig(ops), ops, "HIG.cs:91-164"))
    far = [("lit", b"")] if False else [("lit", pat(300, 154))] + [("copy", 300, 1000)] * 33 + [("copy", 2047, 4), ("copy", 2048, 5), ("copy", 16383, 6), ("copy", 16384, 7), ("copy", 32767, 8), ("lit", b"!")]
    K["hig"].append(case("distance 2 047 (form A), 2 048 and 16 383 (B), 16 384 and 32 767 (C: the 15th bit in the first byte); a u16 raw block in front", "hig", enc_hig(far), far, "HIG.cs:114, :123, :128, :142-145"))
    return K


def xxh32(data, seed=0):
    """xxHash32 as published (Yann Collet's specification), written out here so that the vectors depend on nothing of ours."""
    P1, P2, P3, P4, P5 = 2654435761, 2246822519, 3266489917, 668265263, 374761393
    M = 0xFFFFFFFF
    rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & M
    n, i = len(data), 0
    if n >= 16:
        v = [(seed + P1 + P2) & M, (seed + P2) & M, seed & M, (seed - P1) & M]
        while i + 16 <= n:
            for k in range(4):
                w = int.from_bytes(data[i + 4 * k:i + 4 * k + 4], "little")
                v[k] = (rotl((v[k] + w * P2) & M, 13) * P1) & M
            i += 16
        h = (rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18)) & M
    else:
        h = (seed + P5) & M
    h = (h + n) & M
    while i + 4 <= n:
        h = (rotl((h + int.from_bytes(data[i:i + 4], "little") * P3) & M, 17) * P4) & M
        i += 4
    while i < n:
        h = (rotl((h + data[i] * P5) & M, 11) * P1) & M
        i += 1
    h ^= h >> 15; h = (h * P2) & M
    h ^= h >> 13; h = (h * P3) & M
    h ^= h >> 16
    return h


def crc32c(data):
    """CRC-32C (Castagnoli), bit by bit: reflected polynomial 0x82F63B78, initial value and final XOR 0xFFFFFFFF."""
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def be32(v):
    return bytes([(v >> 24) & 0xFF, (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])


def le32(v):
    return bytes([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF, (v >> 24) & 0xFF])


def build_containers():
    """Container files assembled by hand from the header code of the format classes (not from the library's or the oracle's
    header layer, which share a mould): magic / size fields / section pointers in front of a body from the writers above."""
    out = []

    def cont(name, container, blob, ops, cite, big_endian=1, note=""):
        exp = expand(ops)
        c = {"name": name, "container": container, "file": blob.hex(), "expect_len": len(exp),
             "expect_zlib_b64": base64.b64encode(zlib.compress(exp, 9)).decode(), "big_endian": big_endian, "cite": cite}
        if note:
            c["note"] = note
        out.append(c)

    ops = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x10\x20"), ("copy", 2, 9), ("copy", 33, 17)]
    n = len(expand(ops))
    cont("0x10 + u24 LE size", "LZ10", bytes([0x10, n & 0xFF, (n >> 8) & 0xFF, (n >> 16) & 0xFF]) + enc_lz10(ops), ops, "LZ10.cs:47-57, :67-80")
    cont("0x10 + zero u24 + u32 LE size (the form for sizes above 0xFFFFFF, legal for any size)", "LZ10", bytes([0x10, 0, 0, 0]) + le32(n) + enc_lz10(ops), ops, "LZ10.cs:52-54")
    ops11 = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 4, 16), ("copy", 1, 17), ("copy", 7, 272), ("lit", b"\x7f"), ("copy", 100, 273)]
    n = len(expand(ops11))
    cont("0x11 + u24 LE size", "LZ11", bytes([0x11, n & 0xFF, (n >> 8) & 0xFF, (n >> 16) & 0xFF]) + enc_lz11(ops11), ops11, "LZ11.cs:43-53")
    opsy = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 5, 17), ("copy", 1, 18), ("copy", 2, 273), ("lit", b"\x00"), ("copy", 21, 100)]
    n = len(expand(opsy))
    cont("\"Yaz0\" + BE size + BE alignment + 0", "YAZ0", b"Yaz0" + be32(n) + be32(0x20) + be32(0) + enc_yaz0(opsy), opsy, "Yaz0.cs:58-64, :84-87")
    cont("little-endian size field: the first attempt reads a size of 0x%08X, fails, and the byte-swapped retry succeeds" % int.from_bytes(le32(n), "big"), "YAZ0",
         b"Yaz0" + le32(n) + le32(0) + le32(0) + enc_yaz0(opsy), opsy, "Yaz0.cs:66-78")
    for magic, cname, mio in ((b"Yay0", "YAY0", False), (b"MIO0", "MIO0", True)):
        o = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x42"), ("copy", 2, 10), ("copy", 12, 18)] if mio else \
            [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 5, 17), ("copy", 1, 18), ("lit", b"\x42\x43"), ("copy", 2, 273), ("copy", 30, 60), ("lit", b"\x99")]
        body, a0, a1 = enc_3cursor(o, mio0=mio)
        n = len(expand(o))
        cont("\"%s\" + BE size + token pointer + literal pointer (both from the start of the file), flags at 0x10" % magic.decode(), cname,
             magic + be32(n) + be32(0x10 + a0) + be32(0x10 + a1) + body, o, "Yay0.cs:50-60, :70-77" if not mio else "MIO0.cs:51-61, :72-79")
    opss = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x00\xff"), ("copy", 3, 7), ("copy", 30, 18), ("lit", b"Z")]
    body = enc_lzss(opss)
    cont("\"LZSS\" + BE size + BE compressed size + BE 0", "LZSS", b"LZSS" + be32(len(expand(opss))) + be32(len(body)) + be32(0) + body, opss, "LZSS.cs:53-88")

    # ---- wrappers that only put a magic in front of another container (cls: the format class of the Python mirror that reads it)
    def cont2(name, container, cls, blob, ops, cite, provides_size=True, **kw):
        cont(name, container, blob, ops, cite, **kw)
        out[-1]["cls"] = cls
        out[-1]["provides_size"] = provides_size

    ops = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x10\x20"), ("copy", 2, 9), ("copy", 33, 17)]
    n = len(expand(ops))
    lz10file = bytes([0x10, n & 0xFF, (n >> 8) & 0xFF, (n >> 16) & 0xFF]) + enc_lz10(ops)
    cont2("\"GCLZ\" in front of an LZ10 file", "GCLZ", "GCLZ", b"GCLZ" + lz10file, ops, "GCLZ.cs:33-55")
    cont2("\"CXLZ\" in front of an LZ10 file", "CXLZ", "CXLZ", b"CXLZ" + lz10file, ops, "CXLZ.cs:34-56")
    n = len(expand(ops11))
    cont2("\"COMP\" in front of an LZ11 file", "COMP", "COMP", b"COMP" + bytes([0x11, n & 0xFF, (n >> 8) & 0xFF, (n >> 16) & 0xFF]) + enc_lz11(ops11), ops11, "COMP.cs:34-56")
    n = len(expand(opsy))
    cont2("\"Yaz1\": Yaz0's header under another magic", "YAZ1", "Yaz1", b"Yaz1" + be32(n) + be32(0) + be32(0) + enc_yaz0(opsy), opsy, "Yaz1.cs:19-33, Yaz0.cs:58-64")
    opso = [("lit", pat(20, 30)), ("copy", 20, 8, "M2b"), ("copy", 3, 4, "M2"), ("lit", b"q"), ("copy", 21, 6, "M2b"), ("lit", pat(5, 31))]
    body = enc_lzo(opso)
    plain = [(o[0],) + tuple(o[1:3]) for o in opso]
    cont2("\"LZOn\" 00 2F F1 71 + BE size + BE compressed size + an LZO stream with its end marker", "LZON", "LZOn",
          b"LZOn\x00\x2f\xf1\x71" + be32(len(expand(plain))) + be32(len(body)) + body, plain, "LZOn.cs:41-60")

    # ---- the LZSS-bodied wrappers (LZSS.DefaultProperties and Lzss0Properties are the same geometry: LZSS.cs:33-34)
    body = enc_lzss(opss); n = len(expand(opss))
    cont2("\"AKLZ~?Qd=\\xCC\\xCC\\xCD\" (12 bytes) + BE size + LZSS body", "AKLZ", "AKLZ", b"AKLZ\x7e\x3f\x51\x64\x3d\xcc\xcc\xcd" + be32(n) + body, opss, "AKLZ.cs:14, :35-40")
    cont2("\"LZ01\" + LE file length + LE size + LE 0 + LZSS body", "LZ01", "LZ01", b"LZ01" + le32(16 + len(body)) + le32(n) + le32(0) + body, opss, "LZ01.cs:37-52")
    cont2("no magic: LE compressed size (the body's) + LE size + LZSS body (whose first flag byte has bit 0 set: IsMatch)", "LZSEGA", "LZSega", le32(len(body)) + le32(n) + body, opss, "LZSega.cs:21-33, :41-46")
    cont2("\"SSZL\" + LE 0 + LE compressed size + LE size + LZSS body", "LEVEL5LZSS", "Level5LZSS", b"SSZL" + le32(0) + le32(len(body)) + le32(n) + body, opss, "Level5LZSS.cs:23-24, :34-49")
    cont2("\"MDB4\" + LE size + 1 + LE size + LE compressed size + 16 zero bytes + LZSS body", "MDB4", "MDB4", b"MDB4" + le32(n + 1) + le32(n) + le32(len(body)) + bytes(16) + body, opss, "MDB4.cs:33-50, :57-64")
    cont2("\"FCMP\" + LE size + LE 305397760 + LZSS body", "FCMP", "FCMP", b"FCMP" + le32(n) + le32(305397760) + body, opss, "FCMP.cs:35-41, :47-48")
    cont2("\"IECP\" + LE size + LZSS body", "IECP", "IECP", b"IECP" + le32(n) + body, opss, "IECP.cs:34-39")

    cont2("no magic: LE size + LZSS body; recognised by the file extension only, so IsMatch on bytes alone is false", "GCZ", "GCZ", le32(n) + body, opss, "GCZ.cs:29-41")
    out[-1]["is_match"] = False
    opse = [("lit", pat(11, 60)), ("copy", 11, 3), ("copy", 1, 66), ("lit", b"\x01\x02"), ("copy", 70, 40), ("copy", 2, 5)]
    plain4, bodye = pat(4, 61), enc_lzss(opse, wbits=10, lbits=6, thr=2)
    cont2("\"ECD\" + 1 (compressed) + BE plain size + BE compressed size + BE size + 4 plain bytes + LZSS(0x400, 0x42, 3, 0x3BE) body", "ECD", "ECD",
          b"ECD\x01" + be32(4) + be32(4 + len(bodye)) + be32(4 + len(expand(opse))) + plain4 + bodye, [("lit", plain4)] + opse, "ECD.cs:18, :44-80")
    bodyo = enc_lzo(opso)
    cont2("\"SDPC\" + LE size + an LZO stream", "SDPC", "SDPC", b"SDPC" + le32(len(expand(plain))) + bodyo, plain, "SDPC.cs:31-32, :44-57")

    # ---- the headers of the other flag-byte formats
    ops40 = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 1, 15), ("copy", 4, 16), ("copy", 7, 271), ("lit", b"\x7f"), ("copy", 100, 272), ("copy", 300, 1000), ("copy", 2, 2)]
    n = len(expand(ops40))
    cont2("0x40 + u24 LE size + LZ40 body", "LZ40", "LZ40", bytes([0x40, n & 0xFF, (n >> 8) & 0xFF, n >> 16]) + enc_lz40(ops40), ops40, "LZ40.cs:44-61")
    cont2("0x60 + zero u24 + u32 LE size + LZ40 body", "LZ60", "LZ60", bytes([0x60, 0, 0, 0]) + le32(n) + enc_lz40(ops40), ops40, "LZ60.cs:30-32, LZ40.cs:49-51")
    opsh = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 5, 17), ("copy", 1, 18), ("copy", 2, 273), ("lit", b"\x00"), ("copy", 21, 100)] + [("lit", pat(3, 70 + i)) if i % 3 else ("copy", 7, 4) for i in range(30)]
    cont2("BE size + Yaz0's tokens behind 32-bit flag words", "LZHUDSON", "LZHudson", be32(len(expand(opsh))) + enc_lzhudson(opsh), opsh, "LZHudson.cs:31-46")
    opsm = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x42"), ("copy", 2, 10), ("copy", 12, 18)] + [("lit", pat(2, 80 + i)) if i % 2 else ("copy", 5, 6) for i in range(20)]
    st, a0 = enc_smsr00(opsm)
    cont2("\"SMSR00\" + 2 bytes + BE size + BE pointer to the literal section (from the start of the file); codes at 0x10", "SMSR00", "SMSR00",
          b"SMSR00\x00\x00" + be32(len(expand(opsm))) + be32(0x10 + a0) + st, opsm, "SMSR00.cs:36-37, :68-75")
    opsc = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x10\x20"), ("copy", 2, 9), ("copy", 33, 17)]
    n = len(expand(opsc))
    cont2("\"CLZ\\0\" + BE size + BE 0 + BE size + CLZ0 body", "CLZ0", "CLZ0", b"CLZ\x00" + be32(n) + be32(0) + be32(n) + enc_clz0(opsc), opsc, "CLZ0.cs:33-34, :54-64")
    opsn = [("lit", pat(3, 100)), ("copy", 3, 3), ("copy", 1, 130), ("lit", pat(127, 101)), ("copy", 127, 64), ("lit", pat(128, 102)), ("copy", 256, 5)]
    cont2("\"@CNS\" + a four-character extension + LE size + 4 zero bytes + CNS body", "CNS", "CNS", b"@CNSbin\x00" + le32(len(expand(opsn))) + bytes(4) + enc_cns(opsn), opsn, "CNS.cs:38-39, :54-64")
    ops02 = [("lit", pat(9, 1)), ("copy", 9, 2), ("copy", 1, 16), ("copy", 4, 17), ("copy", 7, 272), ("lit", b"\x7f"), ("copy", 255, 100), ("copy", 256, 3)]
    n = len(expand(ops02))
    cont2("type byte 1 + u24 BE size + LZ02 body with its terminator", "LZ02", "LZ02", bytes([1, n >> 16, (n >> 8) & 0xFF, n & 0xFF]) + enc_lz02(ops02), ops02, "LZ02.cs:52-65")

    # ---- headers of the token-queue formats
    opsx = [("lit", pat(5, 120)), ("copy", 5, 4), ("copy", 1, 35), ("lit", pat(255, 121), "run"), ("copy", 200, 20), ("lit", b"x", "run"), ("lit", pat(3, 122)), ("copy", 2, 5), ("lit", b"z")]
    bodyx = enc_cnx2(opsx)
    cont2("\"CNX\\x02\" + a four-character extension + BE compressed size + BE size + CNX2 body", "CNX2", "CNX2", b"CNX\x02bin\x00" + be32(len(bodyx)) + be32(len(expand(opsx))) + bodyx, opsx, "CNX2.cs:38-39, :54-70")
    opsr = [("lit", pat(2, 130)), ("copy", 2, 3), ("copy", 1, 10), ("lit", pat(3, 131)), ("copy", 5, 4, "medium"), ("copy", 9, 67), ("lit", pat(4, 132)), ("copy", 11, 5, "long"), ("lit", pat(113, 133)), ("copy", 100, 1028), ("lit", b"end")]
    bodyr = enc_refpack(opsr); n = len(expand(opsr))
    cont2("version 1: flag 0x10 + 0xFB + u24 BE size + RefPack body", "REFPACK", "RefPack", bytes([0x10, 0xFB, n >> 16, (n >> 8) & 0xFF, n & 0xFF]) + bodyr, opsr, "RefPack.cs:45-54, :77-102")
    cont2("version 2: LE size of the rest of the file, then the version-1 header", "REFPACK", "RefPack", le32(5 + len(bodyr)) + bytes([0x10, 0xFB, n >> 16, (n >> 8) & 0xFF, n & 0xFF]) + bodyr, opsr, "RefPack.cs:83-90, :107-124")
    opsw = [("lit", pat(9, 110)), ("copy", 9, 5), ("copy", 1, 259), ("lit", pat(255, 111)), ("copy", 300, 100), ("lit", pat(256, 112)), ("copy", 2, 6)]
    bodyw = enc_wflz(opsw, False)
    cont2("\"WFLZ\" + LE compressed size + LE size + WFLZ blocks (little endian, the default byte order)", "WFLZ", "WFLZ", b"WFLZ" + le32(len(bodyw)) + le32(len(expand(opsw))) + bodyw, opsw, "WFLZ.cs:42-43, :57-86", big_endian=0)
    opsk = [("lit", pat(29, 140)), ("copy", 29, 1), ("copy", 1, 7), ("copy", 30, 8), ("lit", pat(30, 141)), ("copy", 31, 262), ("lit", b"z")]
    bodyk = enc_lzshrek(opsk)
    cont2("LE 0x10 (body offset) + LE size + LE body length + LE 0, body at 0x10", "LZSHREK", "LZShrek", le32(0x10) + le32(len(expand(opsk))) + le32(len(bodyk)) + le32(0) + bodyk, opsk, "LZShrek.cs:30-31, :43-60")
    opsg = [("lit", pat(5, 150)), ("copy", 5, 4), ("lit", b"a"), ("copy", 1, 9), ("lit", b"bc"), ("copy", 3, 10), ("copy", 7, 35), ("lit", pat(3, 151)), ("copy", 20, 18, "C"), ("copy", 2, 274)]
    hdr = b"HIG!" + le32(0x40) + bytes(4 * 12) + le32(1) + le32(len(expand(opsg)))
    assert len(hdr) == 0x40
    cont2("64-byte header: \"HIG!\", body offset (0x40) in word 1, version in word 14, size in word 15", "HIG", "HIG", hdr + enc_hig(opsg), opsg, "HIG.cs:46-47, :58-90")

    # ---- LZ10 / LZ11 / LZSS behind further wrappers
    opsl = [("lit", pat(9, 1)), ("copy", 9, 3), ("copy", 1, 18), ("lit", b"\x10\x20"), ("copy", 2, 9), ("copy", 33, 17)]
    nl = len(expand(opsl))
    lz10body = enc_lz10(opsl)
    lz10f = bytes([0x10, nl & 0xFF, (nl >> 8) & 0xFF, nl >> 16]) + lz10body
    cont2("\"3DS-LZ\\r\\n\" in front of an LZ10 file", "LZ_3DS", "LZ_3DS", b"3DS-LZ\r\n" + lz10f, opsl, "3DS-LZ.cs:24-39")
    cont2("LE u32: size << 3 | 1 (LZ10), then the LZ10 body", "LEVEL5", "Level5", le32((nl << 3) | 1) + lz10body, opsl, "Level5.cs:62-98, :151-159")
    out[-1]["skip_is_match"] = True                               # (IsMatch runs LZ10.Validate over the body: not asserted here)
    cont2("\"LZ77\" + type 0x10 + u24 LE size + LZ10 body", "LZ77", "LZ77", b"LZ77" + bytes([0x10, nl & 0xFF, (nl >> 8) & 0xFF, nl >> 16]) + lz10body, opsl, "LZ77.cs:112-124")
    n11 = len(expand(ops11))
    cont2("\"LZ77\" + type 0x11 + u24 LE size + LZ11 body", "LZ77", "LZ77", b"LZ77" + bytes([0x11, n11 & 0xFF, (n11 >> 8) & 0xFF, n11 >> 16]) + enc_lz11(ops11), ops11, "LZ77.cs:125-127")
    ch2 = [("lit", pat(7, 170)), ("copy", 7, 18), ("copy", 3, 3), ("lit", b"!")]
    n2 = len(expand(ch2))
    f2 = bytes([0x10, n2 & 0xFF, (n2 >> 8) & 0xFF, n2 >> 16]) + enc_lz10(ch2)
    ntot = nl + n2
    ends = [len(lz10f), len(lz10f) + len(f2)]
    table = b"".join(bytes([e & 0xFF, e >> 8]) for e in ends)
    cont2("\"LZ77\" + type 0xF7 (ChunkLZ10) + u24 LE total size + u16 LE chunk end offsets (relative to the end of the table, the last one reaching the end of the file) + one LZ10 FILE per chunk",
          "LZ77", "LZ77", b"LZ77" + bytes([0xF7, ntot & 0xFF, (ntot >> 8) & 0xFF, ntot >> 16]) + table + lz10f + f2, opsl + [("lit", expand(ch2))], "LZ77.cs:137-160")
    # LZ00: the LZSS body is XORed with a keystream, one step of the generator per byte read  LZ00.cs:128-141
    def lz00_crypt(data, key):
        M, outb = 0xFFFFFFFF, bytearray()
        for v in data:
            x = ((key << 1) + key) & M                             # (((((((Key << 1) + Key) << 5) - Key) << 5) + Key) << 7) - Key, in 32 bits
            x = ((x << 5) - key) & M
            x = ((x << 5) + key) & M
            x = ((x << 7) - key) & M
            x = ((x << 6) - x) & M
            x = ((x << 4) - x) & M
            key = (((x << 2) - x) + 12345) & M
            t = (key >> 16) & 0x7FFF
            outb.append((v ^ ((((t << 8) - t) & M) >> 15)) & 0xFF)
        return bytes(outb)
    body = enc_lzss(opss); n = len(expand(opss)); key = 0x1234ABCD
    cont2("\"LZ00\" + LE file length + 8 bytes + a 32-byte name + LE size + LE key + 8 bytes, then the LZSS body under the keystream", "LZ00", "LZ00",
          b"LZ00" + le32(0x40 + len(body)) + bytes(8) + b"kat.bin".ljust(32, b"\x00") + le32(n) + le32(key) + bytes(8) + lz00_crypt(body, key), opss, "LZ00.cs:37-38, :62-82, :128-141")

    # ---- BLZ: the file is read from its END (footer, then the code backwards), and the output is written from its end as well
    def enc_blz_file(ops):
        """BLZ.Decompress / DecompressHeaderless  AuroraLib.Compression.Nintendo/Nintendo/BLZ.cs:60-135: in the order the decoder consumes them, the
        code bytes are LZ10-like -- flags MSB first (:104-108), bit 1 = two bytes, high then low: (length - 3) << 12 | (distance - 3) (:116-119) --
        describing the output from its LAST byte to its first; the file holds them reversed, followed by u24 LE compressed size, u8 footer +
        padding size (8), i32 LE (decompressed - compressed size)."""
        f = Flags8(msb_first=True)
        for t in tokens(ops):
            if t[0] == "lit":
                f.bit(0); f.out.append(t[1])
            else:
                _, d, n = t
                assert 3 <= n <= 18 and 3 <= d <= 4098
                v = ((n - 3) << 12) | (d - 3)
                f.bit(1); f.out += bytes([v >> 8, v & 0xFF])
        code = bytes(f.out)[::-1]
        csize, dsize = len(code) + 8, len(expand(ops))
        return code + bytes([csize & 0xFF, (csize >> 8) & 0xFF, csize >> 16, 8]) + ((dsize - csize) & 0xFFFFFFFF).to_bytes(4, "little")
    opsb = [("lit", pat(9, 160)), ("copy", 9, 3), ("copy", 3, 18), ("lit", b"\x10\x20"), ("copy", 4, 9), ("copy", 33, 17)] + [("copy", 20, 18)] * 230 + [("copy", 4098, 5), ("copy", 4097, 3), ("lit", b"#")]
    cont2("read backwards: lengths 3 / 9 / 17 / 18, distances 3 (the smallest), 4 097 and 4 098 (all 12 bits + 3); footer without padding", "BLZ", "BLZ",
          enc_blz_file(opsb), opsb, "BLZ.cs:60-135")
    exp = expand(opsb)[::-1]                                       # what the ops describe is the output back to front
    out[-1]["expect_len"] = len(exp); out[-1]["expect_zlib_b64"] = base64.b64encode(zlib.compress(exp, 9)).decode()

    # ---- LZ4: legacy frames and LZ4 frames
    b1 = [("lit", pat(12, 40)), ("copy", 12, 8), ("copy", 1, 30), ("lit", pat(5, 41))]
    b2 = [("lit", pat(7, 42)), ("copy", 3, 20), ("lit", pat(6, 43))]
    blk1, blk2 = enc_lz4(b1), enc_lz4(b2)
    cont2("legacy frame: magic 0x184C2102 + (u32 LE block size + block) x 2, ended by the end of the file; every block starts a new window", "LZ4_LEGACY", "LZ4Legacy",
          le32(0x184C2102) + le32(len(blk1)) + blk1 + le32(len(blk2)) + blk2, b1 + b2, "LZ4.cs:50-58, :96-111, :162-175", provides_size=False)
    raw2 = pat(23, 44)
    desc = bytes([0x40 | 0x20, 0x40])                                  # version 01, independent blocks; 64 KiB blocks
    frame = le32(0x184D2204) + desc + bytes([(xxh32(desc) >> 8) & 0xFF]) + le32(len(blk1)) + blk1 + le32(0x80000000 | len(raw2)) + raw2 + le32(0)
    cont2("LZ4 frame: descriptor FLG / BD / header checksum byte, a compressed block, an UNCOMPRESSED block (size with bit 31), EndMark", "LZ4_FRAME", "LZ4",
          frame, b1 + [("lit", raw2)], "LZ4.Frame.cs:107-150, LZ4.FrameDescriptor.cs:18-44", provides_size=False)
    # linked blocks (no independence flag): the second block copies from the first block's output; content size, block and content checksums
    b3 = [("lit", pat(4, 45)), ("copy", 30, 12), ("copy", 5, 4)]      # distance 30 reaches into block 1's output
    blk3 = enc_lz4(b3)
    content = expand(b1 + b3)
    desc = bytes([0x40 | 0x10 | 0x08 | 0x04, 0x50]) + len(content).to_bytes(8, "little")
    frame = le32(0x184D2204) + desc + bytes([(xxh32(desc) >> 8) & 0xFF]) + le32(len(blk1)) + blk1 + le32(xxh32(blk1)) + le32(len(blk3)) + blk3 + le32(xxh32(blk3)) + le32(0) + le32(xxh32(content))
    cont2("LZ4 frame with linked blocks (one window for the frame), ContentSize, block checksums and a content checksum (xxHash32, seed 0)", "LZ4_FRAME", "LZ4",
          frame, b1 + b3, "LZ4.Frame.cs:120, :136-138, :152-173", provides_size=False)
    skip = le32(0x184D2A53) + le32(5) + b"SKIP!"
    cont2("a skippable frame (magic 0x184D2A50..5F + u32 size) between two frames of one file", "LZ4_FRAME", "LZ4",
          le32(0x184C2102) + le32(len(blk2)) + blk2 + skip + le32(0x184D2204) + bytes([0x60, 0x40]) + bytes([(xxh32(bytes([0x60, 0x40])) >> 8) & 0xFF]) + le32(len(blk1)) + blk1 + le32(0),
          b2 + b1, "LZ4.cs:52-93", provides_size=False)

    # ---- Snappy framing
    s1 = [("lit", pat(9, 50)), ("copy", 9, 4), ("copy", 1, 11), ("lit", pat(3, 51))]
    sb = enc_snappy(s1)
    rawc = pat(17, 52)
    ident = b"\xff\x06\x00\x00sNaPpY"
    def u24(v): return bytes([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF])
    def mask(c): return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    blob = ident + b"\x00" + u24(4 + len(sb)) + le32(mask(crc32c(expand(s1)))) + sb + b"\xfe" + u24(3) + b"pad" + b"\x01" + u24(4 + len(rawc)) + le32(mask(crc32c(rawc))) + rawc + b"\x80" + u24(2) + b"zz"
    cont2("stream identifier + a compressed chunk (masked CRC-32C, raw Snappy body) + a padding chunk (0xFE) + an uncompressed chunk + a skippable chunk (0x80)", "SNAPPY", "Snappy",
          blob, s1 + [("lit", rawc)], "Snappy.cs:39-68", provides_size=False)
    return out


def main():
    with open(os.path.join(HERE, "kat_containers.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_kats.py (headers assembled by hand from the format classes; no decoder, no header layer involved)",
                   "cases": build_containers()}, f, indent=1)
        f.write("\n")
    K = build()
    for fmt, cases in sorted(K.items()):
        with open(os.path.join(HERE, "kat_%s.json" % fmt), "w") as f:
            json.dump({"format": fmt, "generator": "tests/golden/make_kats.py (hand-assembled from the cited C# lines; no decoder involved)", "cases": cases}, f, indent=1)
            f.write("\n")
    print("wrote %d files, %d cases" % (len(K), sum(len(v) for v in K.values())))


if __name__ == "__main__":
    main()
