// alz_decode_serial.h -- exact, wave-uniform token parsers for every format on the hot path.
//
// These are the "reference semantics" decoders of the GPU build: one token at a
// time, parse state in SGPRs, the byte work (match copy / literal runs) spread
// over the 64 lanes by OutWin.  They implement every frozen edge definition
// (E1-E6) and are what the lane-parallel fast paths fall back to
// at stream tails.  Each function cites the reference body it replaces
// (paths relative to /root/reference/src).
#pragma once
#include "alz_device.h"
#include "auroralz.h"

struct DecState {
    u32 p;              // input offset (wave-uniform)
    u32 bits;           // FlagReader.BitsLeft
    u32 flag;           // FlagReader.CurrentFlag
    bool eof, ovf, bad, done;
    u64 attempted_end;  // E5 bookkeeping
};

__device__ __forceinline__ void dec_state_init(DecState& s) {
    s.p = 0; s.bits = 0; s.flag = 0; s.eof = false; s.ovf = false; s.bad = false; s.done = false; s.attempted_end = 0;
}

// E5: clip a token of `len` bytes against dst_cap
template <class OW>
__device__ __forceinline__ u32 clip_token(const OW& out, DecState& s, u64 len) {
    u64 end = (u64)out.produced + len;
    if (end > (u64)out.cap) { s.ovf = true; s.attempted_end = end; return out.cap - out.produced; }
    return (u32)len;
}

// The parsers below are written once against a SINK.  DirectSink executes every token immediately on the window
// (exact serial semantics).  QueueSink (alz_decode_fast.h) only records tokens in lane registers and executes 64 of
// them at a time through the lane-parallel byte phase.  Every sink operation returns false once decoding must stop
// (capacity reached, E5); the parser then returns and the kernel resolves the status.
template <class OW>
struct DirectSink {
    OW& out; DecState& s;
    __device__ __forceinline__ DirectSink(OW& o, DecState& st) : out(o), s(st) {}
    __device__ __forceinline__ u32 produced() const { return out.produced; }
    __device__ __forceinline__ void ensure(InCache& in, u32 p, u32 need) { in.ensure(p, need); }
    __device__ __forceinline__ bool lit(u32 b) { if (clip_token(out, s, 1) < 1) return false; out.put_byte(b); return true; }
    __device__ __forceinline__ bool match(u32 dist, u64 len, u32 W) { u32 cl = clip_token(out, s, len); out.back_copy(dist, cl, W); return !s.ovf; }
    __device__ __forceinline__ bool run(InCache& in, u32 p, u64 len) { u32 cl = clip_token(out, s, len); out.copy_from(in, p, cl); return !s.ovf; }
    __device__ __forceinline__ void flush() {}
};

// ---------------------------------------------------------------------------------------------
// LZSS.DecompressHeaderless  Formats/Common/LZSS.cs:91-130  (flags LSB-first, bit 1 = literal)
template <class SK>
__device__ void dec_lzss_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 size,
                                u32 length_bits, u32 min_length, u32 windows_start, u32 max_distance, u32 W) {
    const u32 f = (1u << length_bits) - 1u, n = max_distance - 1u;
    while (sk.produced() < size) {
        sk.ensure(in, s.p, 8);
        if (s.bits == 0) { if (s.p >= src_len) { s.eof = true; return; } s.flag = in.peek1(s.p); s.p++; s.bits = 8; }
        u32 bit = (s.flag >> (8 - s.bits)) & 1u; s.bits--;
        if (bit) {
            if (s.p >= src_len) { s.eof = true; return; }
            u32 b = in.peek1(s.p); s.p++;
            if (!sk.lit(b)) return;
        } else {
            if (s.p + 2 > src_len) { s.eof = true; return; }
            u32 w = in.peek4(s.p); s.p += 2;
            u32 b1 = w & 0xFF, b2 = (w >> 8) & 0xFF;
            u32 offset = ((b2 >> length_bits) << 8) | b1;
            u32 length = (b2 & f) + min_length;
            offset = (max_distance + offset - windows_start) & n;
            u32 pos = sk.produced() & (W - 1);                // LzWindows.OffsetCopy  IO/LzWindows.cs:108-115
            u32 distance = pos >= offset ? pos - offset : pos - offset + W;
            if (!sk.match(distance, length, W)) return;
        }
    }
}

// LZ10.DecompressHeaderless  Nintendo/LZ10.cs:82-111 ; LZ11.DecompressHeaderless  Nintendo/LZ11.cs:83-133
template <class SK, bool LZ11>
__device__ void dec_lz1x_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 size) {
    while (sk.produced() < size) {
        sk.ensure(in, s.p, 8);
        if (s.bits == 0) { if (s.p >= src_len) { s.eof = true; return; } s.flag = in.peek1(s.p); s.p++; s.bits = 8; }
        u32 bit = (s.flag >> (s.bits - 1)) & 1u; s.bits--;
        if (bit) {
            if (s.p + 2 > src_len) { s.eof = true; return; }
            u32 w = in.peek4(s.p);
            u32 b1 = w & 0xFF, b2 = (w >> 8) & 0xFF, b3 = (w >> 16) & 0xFF, b4 = w >> 24;
            u32 distance, length;
            if (LZ11 && (b1 >> 4) == 0) {
                if (s.p + 3 > src_len) { s.eof = true; return; }
                distance = (((b2 & 0xF) << 8) | b3) + 1; length = (((b1 & 0xF) << 4) | (b2 >> 4)) + 17; s.p += 3;
            } else if (LZ11 && (b1 >> 4) == 1) {
                if (s.p + 4 > src_len) { s.eof = true; return; }
                distance = (((b3 & 0xF) << 8) | b4) + 1; length = (((b1 & 0xF) << 12) | (b2 << 4) | (b3 >> 4)) + 273; s.p += 4;
            } else {
                distance = (((b1 & 0xF) << 8) | b2) + 1; length = (b1 >> 4) + (LZ11 ? 1 : 3); s.p += 2;
            }
            if (!sk.match(distance, length, 4096)) return;
        } else {
            if (s.p >= src_len) { s.eof = true; return; }
            u32 b = in.peek1(s.p); s.p++;
            if (!sk.lit(b)) return;
        }
    }
}

// CLZ0.DecompressHeaderless  Marvelous/CLZ0.cs:64-97: flags LSB first, 1 = match; delta = b1 | (b2 >> 4) << 8, distance = 0x1000 - delta,
// length = (b2 & 15) + 3.  (ReadByte() == -1 inside a match is refused as truncated input, see the oracle.)
template <class SK>
__device__ void dec_clz0_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 size) {
    while (sk.produced() < size) {
        sk.ensure(in, s.p, 8);
        if (s.bits == 0) { if (s.p >= src_len) { s.eof = true; return; } s.flag = in.peek1(s.p); s.p++; s.bits = 8; }
        const u32 bit = (s.flag >> (8 - s.bits)) & 1u; s.bits--;
        if (bit) {
            if (s.p + 2 > src_len) { s.eof = true; s.p = src_len; return; }
            const u32 w = in.peek4(s.p); s.p += 2;
            const u32 b1 = w & 0xFF, b2 = (w >> 8) & 0xFF;
            if (!sk.match(0x1000u - (b1 | ((b2 >> 4) << 8)), (b2 & 0xFu) + 3u, 4096)) return;
        } else {
            if (s.p >= src_len) { s.eof = true; return; }
            const u32 b = in.peek1(s.p); s.p++;
            if (!sk.lit(b)) return;
        }
    }
}

// LZ40.DecompressHeaderless  Nintendo/LZ40.cs:80-132 (also LZ60's body).  s.flag holds the flag byte already negated
// (:92), bits MSB first, 1 = match; tokens are u16 LE distance << 4 | length nibble, nibble 0 / 1 = one / two more
// length bytes; distance 0 is what the encoder writes for 4096 (E1).
template <class SK>
__device__ void dec_lz40_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 size) {
    while (sk.produced() < size) {
        sk.ensure(in, s.p, 8);
        if (s.bits == 0) { if (s.p >= src_len) { s.eof = true; return; } s.flag = (0u - in.peek1(s.p)) & 0xFFu; s.p++; s.bits = 8; }
        u32 bit = (s.flag >> (s.bits - 1)) & 1u; s.bits--;
        if (bit) {
            if (s.p + 2 > src_len) { s.eof = true; return; }
            u32 w = in.peek4(s.p);
            u32 v = w & 0xFFFFu, b3 = (w >> 16) & 0xFF, b4 = w >> 24;
            u32 length = v & 0xFu, distance = v >> 4;
            if (length == 0) { if (s.p + 3 > src_len) { s.eof = true; return; } length = b3 + 16u; s.p += 3; }
            else if (length == 1) { if (s.p + 4 > src_len) { s.eof = true; return; } length = (b3 | (b4 << 8)) + 272u; s.p += 4; }
            else s.p += 2;
            if (!sk.match(distance, length, 4096)) return;
        } else {
            if (s.p >= src_len) { s.eof = true; return; }
            u32 b = in.peek1(s.p); s.p++;
            if (!sk.lit(b)) return;
        }
    }
}

// Yaz0: Yay0.DecompressHeaderless with all three cursors on one stream  Nintendo/Yay0.cs:110-144, Yaz0.cs:91-92
template <class SK>
__device__ void dec_yaz0_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 size) {
    while (sk.produced() < size) {
        sk.ensure(in, s.p, 8);
        if (s.bits == 0) { if (s.p >= src_len) { s.eof = true; return; } s.flag = in.peek1(s.p); s.p++; s.bits = 8; }
        u32 bit = (s.flag >> (s.bits - 1)) & 1u; s.bits--;
        if (bit) {
            if (s.p >= src_len) { s.eof = true; return; }
            u32 b = in.peek1(s.p); s.p++;
            if (!sk.lit(b)) return;
        } else {
            if (s.p + 2 > src_len) { s.eof = true; return; }
            u32 w = in.peek4(s.p); s.p += 2;
            u32 b1 = w & 0xFF, b2 = (w >> 8) & 0xFF, b3 = (w >> 16) & 0xFF;
            u32 distance = (((b1 & 0x0F) << 8) | b2) + 1;
            u32 length = b1 >> 4;
            if (length == 0) {                                   // ReadByte(): -1 at EOF => 17   Yay0.cs:130-131
                if (s.p < src_len) { length = b3 + 0x12; s.p++; } else length = 17;
            } else length += 2;
            if (!sk.match(distance, length, 4096)) return;
        }
    }
}

// Yay0 (three cursors)  Nintendo/Yay0.cs:99-144 ; MIO0  Nintendo/MIO0.cs:105-149
// fc: flags from 0, cc: tokens from aux0, uc: literals from aux1; each cursor bounded by its slice length.
template <class SK, bool MIO0>
__device__ u32 dec_3cursor_serial(InCache& fin, InCache& cin, InCache& uin, SK& sk, DecState& s, u32 src_len, u32 size,
                                  u32 fptr0, u32 cptr0, u32 uptr0) {    // returns the input bytes used (by value: a
                                                                        // reference to a caller's local puts it in scratch)
    u32 fp = fptr0, cp = cptr0, up = uptr0;
    while (sk.produced() < size) {
        if (s.bits == 0) {
            if (fp >= src_len) { s.eof = true; break; }
            sk.ensure(fin, fp, 1); s.flag = fin.peek1(fp); fp++; s.bits = 8;
        }
        u32 bit = (s.flag >> (s.bits - 1)) & 1u; s.bits--;
        if (bit) {
            if (up >= src_len) { s.eof = true; break; }
            sk.ensure(uin, up, 1); u32 b = uin.peek1(up); up++;
            if (!sk.lit(b)) break;
        } else {
            if (cp + 2 > src_len) { s.eof = true; if (MIO0 && cp < src_len) cp++; break; }
            sk.ensure(cin, cp, 4); u32 w = cin.peek4(cp); cp += 2;
            u32 b1 = w & 0xFF, b2 = (w >> 8) & 0xFF;
            u32 distance = (((b1 & 0x0F) << 8) | b2) + 1;
            u32 length;
            if (MIO0) length = (b1 >> 4) + 3;
            else {
                length = b1 >> 4;
                if (length == 0) {
                    if (up < src_len) { sk.ensure(uin, up, 1); length = uin.peek1(up) + 0x12; up++; } else length = 17;
                } else length += 2;
            }
            if (!sk.match(distance, length, 4096)) break;
        }
    }
    return cp > up ? cp : up;
}

// LZHudson.DecompressHeaderless  HudsonSoft/LZHudson.cs:53: Yay0.DecompressHeaderless with all three cursors on one stream
// (like Yaz0) and FlagReader(source, Endian.Big, 4, Endian.Big): 32-bit big-endian flag words, MSB first
template <class SK>
__device__ void dec_lzhudson_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 size) {
    while (sk.produced() < size) {
        sk.ensure(in, s.p, 8);
        if (s.bits == 0) {
            if (s.p + 4 > src_len) { s.eof = true; s.p = src_len; return; }      // ReadInt32 past the end: EndOfStreamException
            s.flag = __builtin_bswap32(in.peek4(s.p)); s.p += 4; s.bits = 32;
        }
        u32 bit = (s.flag >> (s.bits - 1)) & 1u; s.bits--;
        if (bit) {
            if (s.p >= src_len) { s.eof = true; return; }
            u32 b = in.peek1(s.p); s.p++;
            if (!sk.lit(b)) return;
        } else {
            if (s.p + 2 > src_len) { s.eof = true; return; }
            u32 w = in.peek4(s.p); s.p += 2;
            u32 b1 = w & 0xFF, b2 = (w >> 8) & 0xFF, b3 = (w >> 16) & 0xFF;
            u32 distance = (((b1 & 0x0F) << 8) | b2) + 1, length = b1 >> 4;
            if (length == 0) { if (s.p < src_len) { length = b3 + 0x12; s.p++; } else length = 17; }   // ReadByte() == -1 -> 17  Yay0.cs:130-131
            else length += 2;
            if (!sk.match(distance, length, 4096)) return;
        }
    }
}

// SMSR00.DecompressHeaderless  Nintendo/SMSR00.cs:85-131: cin walks the code section [0, codes_len) -- 16-bit big-endian
// masks (MSB first, 1 = literal) each followed by the match words of its 16 tokens --, uin the literal section behind it
template <class SK>
__device__ void dec_smsr00_serial(InCache& cin, InCache& uin, SK& sk, DecState& s, u32 src_len, u32 size, u32 codes_len, u32& used,
                                  u32 cp0 = 0, u32 up0 = 0xFFFFFFFFu) {
    u32 cp = cp0, up = up0 == 0xFFFFFFFFu ? codes_len : up0;        // (resumed behind the lane-parallel loop at a mask boundary)
    while (sk.produced() < size) {
        if (s.bits == 0) {
            if (cp + 2 > codes_len) { s.eof = true; break; }                      // codes[codePointer++]: IndexOutOfRangeException
            sk.ensure(cin, cp, 4); const u32 w = cin.peek4(cp); cp += 2;
            s.flag = ((w & 0xFF) << 8) | ((w >> 8) & 0xFF); s.bits = 16;
        }
        u32 bit = (s.flag >> (s.bits - 1)) & 1u; s.bits--;
        if (bit) {
            if (up >= src_len) { s.eof = true; break; }
            sk.ensure(uin, up, 1); u32 b = uin.peek1(up); up++;
            if (!sk.lit(b)) break;
        } else {
            if (cp + 2 > codes_len) { s.eof = true; break; }
            sk.ensure(cin, cp, 4); const u32 w = cin.peek4(cp); cp += 2;
            const u32 data = ((w & 0xFF) << 8) | ((w >> 8) & 0xFF);
            if (!sk.match((data & 0x0FFF) + 1, (data >> 12) + 3, 4096)) break;
        }
    }
    used = up;                                                                      // source.Position: behind the literals read
}

// PRS.DecompressHeaderless(Stream, Stream, Endian)  Sega/PRS.cs:59-102
template <class SK, bool BIG>
__device__ void dec_prs_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 max_tokens = 0xFFFFFFFFu) {
#define PRS_READBIT(dst)                                                                       \
    do {                                                                                       \
        if (s.bits == 0) { if (s.p >= src_len) { s.eof = true; return; } s.flag = in.peek1(s.p); s.p++; s.bits = 8; } \
        dst = BIG ? (s.flag >> (s.bits - 1)) & 1u : (s.flag >> (8 - s.bits)) & 1u; s.bits--;   \
    } while (0)
    while (s.p < src_len) {
        if (max_tokens-- == 0) return;
        sk.ensure(in, s.p, 16);
        u32 bit; PRS_READBIT(bit);
        if (bit) {
            if (s.p >= src_len) { s.eof = true; return; }
            u32 b = in.peek1(s.p); s.p++;
            if (!sk.lit(b)) return;
        } else {
            u32 distance, length, bit2; PRS_READBIT(bit2);
            if (bit2) {
                if (s.p + 2 > src_len) { s.eof = true; return; }
                u32 w = in.peek4(s.p); s.p += 2;
                u32 x0 = w & 0xFF, x1 = (w >> 8) & 0xFF;
                u32 v = BIG ? ((x0 << 8) | x1) : ((x1 << 8) | x0);
                if (v == 0) { s.done = true; return; }
                length = v & 7; distance = 0x2000 - (v >> 3);
                if (length == 0) { if (s.p >= src_len) { s.eof = true; return; } length = ((w >> 16) & 0xFF) + 1; s.p++; }
                else length += 2;
            } else {
                u32 h, l; PRS_READBIT(h); PRS_READBIT(l);
                length = ((h << 1) | l) + 2;
                if (s.p >= src_len) { s.eof = true; return; }
                distance = 0x100 - in.peek1(s.p); s.p++;
            }
            if (!sk.match(distance, length, 8192)) return;
        }
    }
    s.eof = true;   // EndOfStreamException  PRS.cs:101
#undef PRS_READBIT
}

// LZ4.DecompressBlockHeaderless  Formats/Common/LZ4.cs:176-200
template <class SK>
__device__ void dec_lz4_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 max_tokens = 0xFFFFFFFFu) {
    while (s.p < src_len) {
        if (max_tokens-- == 0) return;
        sk.ensure(in, s.p, 8);
        u32 token = in.peek1(s.p); s.p++;
        u64 plain = token >> 4;
        if (plain == 0xF) {
            u32 b;
            do { if (s.p >= src_len) { s.eof = true; return; } sk.ensure(in, s.p, 1); b = in.peek1(s.p); s.p++; plain += b; } while (b == 255);
        }
        if (plain > (u64)(src_len - s.p)) { s.eof = true; return; }
        if (!sk.run(in, s.p, plain)) return;
        s.p += (u32)plain;
        if (s.p >= src_len) break;
        u64 mlen = token & 0xF;
        if (s.p + 2 > src_len) { s.eof = true; return; }
        sk.ensure(in, s.p, 4);
        u32 w = in.peek4(s.p); s.p += 2;
        u32 dist = w & 0xFFFF;
        if (mlen == 0xF) {
            u32 b;
            do { if (s.p >= src_len) { s.eof = true; return; } sk.ensure(in, s.p, 1); b = in.peek1(s.p); s.p++; mlen += b; } while (b == 255);
        }
        if (!sk.match(dist, mlen + 4, 65536)) return;
    }
}

// BLZ.DecompressHeaderless  Nintendo/BLZ.cs:97-135 in stream order (the managed code walks both spans from their ends: src is
// the code section reversed, the output comes out reversed; the container layer does both reversals).  L = length of the
// destination span.  Flags MSB first, 1 = match (big-endian u16: length - 3 in the high nibble, distance - 3 below).  A match
// is cut silently where the span ends (:121); a literal there, a read behind the input, or a match source beyond the span
// are IndexOutOfRangeExceptions.  Decoding ends when the input is used up.
// `until`: return at the first flag-byte boundary where that much output exists (the lane-parallel loop takes over).
template <class SK>
__device__ __forceinline__ void dec_blz_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 L, u32 until = 0xFFFFFFFFu) {
    while (s.p < src_len) {
        if (s.bits == 0 && sk.produced() >= until) return;
        sk.ensure(in, s.p, 8);
        if (s.bits == 0) { s.flag = in.peek1(s.p); s.p++; s.bits = 8; }
        const u32 bit = (s.flag >> (s.bits - 1)) & 1u; s.bits--;
        if (!bit) {
            if (sk.produced() >= L) { sk.flush(); if (!s.ovf) { s.ovf = true; s.attempted_end = (u64)sk.produced() + 1u; } return; }   // destination[--dst]
            if (s.p >= src_len) { s.eof = true; return; }                            // source[--src]
            if (!sk.lit(in.peek1(s.p))) return;
            s.p++;
        } else {
            if (s.p + 2u > src_len) { s.eof = true; s.p = src_len; return; }
            const u32 w = in.peek4(s.p); s.p += 2;
            const u32 inf = ((w & 0xFFu) << 8) | ((w >> 8) & 0xFFu);
            const u32 dist = (inf & 0x0FFFu) + 3u; u32 len = (inf >> 12) + 3u;
            const u32 room = L - sk.produced();
            if (len > room) len = room;
            if (len && dist > sk.produced()) { s.bad = true; return; }               // destination[dst - 1 + distance]
            if (len && !sk.match(dist, len, 8192)) return;
        }
    }
}

// HIG.DecompressHeaderless  Specialized/HIG.cs:126-212: an initial literal block (count byte + 2, or 0 + u16 LE count), then while the output
// is short of the declared size: a match -- LLLD DDPP D (length 4-9), 110L LLLL DDDD DDPP D (4-35), 111D LLLL [L | 0 LL] DDDD DDPP D
// (3-65 535) -- followed by the literals its PP field announces (3: none, 1, 2, 0: a counted block).  32 KiB window.
template <class SK>
__device__ __forceinline__ bool hig_raw(InCache& in, SK& sk, DecState& s, u32 src_len) {
    if (s.p >= src_len) { s.eof = true; return false; }                              // ReadByte() == -1: a count of 1 that ReadExactly fails on
    sk.ensure(in, s.p, 4);
    u32 plain = in.peek1(s.p) + 2u; s.p++;
    if (plain == 2u) {
        if (s.p + 2u > src_len) { s.eof = true; s.p = src_len; return false; }
        plain = in.peek4(s.p) & 0xFFFFu; s.p += 2;
    }
    if (plain > src_len - s.p) { s.eof = true; return false; }                       // LzWindows.CopyFrom -> ReadExactly throws
    if (!sk.run(in, s.p, plain)) return false;
    s.p += plain;
    return true;
}
// Resumable at match boundaries: s.bits = 1 once the initial literal block is out.
template <class SK>
__device__ __forceinline__ void dec_hig_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 size, u32 max_tokens = 0xFFFFFFFFu) {
    if (s.bits == 0) { if (!hig_raw(in, sk, s, src_len)) return; s.bits = 1; }
    while (sk.produced() < size) {
        if (max_tokens-- == 0) return;
        if (s.p >= src_len) { s.eof = true; return; }
        sk.ensure(in, s.p, 8);
        const u32 b = in.peek1(s.p); s.p++;
        u32 length = b >> 5, distance, plain;
        if (length < 6u) { length += 4u; distance = (b & 0x1Cu) << 6; plain = b & 3u; }
        else {
            if (length == 6u) { length = (b & 0x1Fu) + 4u; distance = 0; }
            else {
                length = (b & 0xFu) + 3u; distance = (b & 0x10u) << 10;
                if (length == 3u) {
                    if (s.p >= src_len) { s.eof = true; return; }
                    length = in.peek1(s.p) + 18u; s.p++;
                    if (length == 18u) {
                        if (s.p + 2u > src_len) { s.eof = true; s.p = src_len; return; }
                        sk.ensure(in, s.p, 4);
                        const u32 w = in.peek4(s.p); s.p += 2;
                        length = ((w & 0xFFu) << 8) | ((w >> 8) & 0xFFu);
                    }
                }
            }
            if (s.p >= src_len) { s.eof = true; return; }
            sk.ensure(in, s.p, 4);
            const u32 b2 = in.peek1(s.p); s.p++;
            distance |= (b2 & 0xFCu) << 6; plain = b2 & 3u;
        }
        if (s.p >= src_len) { s.eof = true; return; }
        distance |= in.peek1(s.p); s.p++;
        if (length && !sk.match(distance, length, 32768)) return;                    // (E1: distance 0 = the window size)
        if (plain == 0u) { if (!hig_raw(in, sk, s, src_len)) return; }
        else if (plain < 3u) {                                                       // WriteByte(ReadUInt8()) once or twice: read, then write
            for (u32 i = 0; i < plain; i++) {
                if (s.p >= src_len) { s.eof = true; return; }
                const u32 at = s.p; s.p++;
                if (!sk.run(in, at, 1)) return;
            }
        }
    }
}

// LZShrek.DecompressHeaderless  Activision/LZShrek.cs:73-119 (span based).  A group is a flag (literal count field << 3 | matches - 1),
// the literals, then 1..8 matches; count / distance fields: 0-29 in the flag, 30 = 30 + next byte, 31 = 286 + next u16 LE
// (ReadDistance :176-190); a match length 1..7 sits in its flag, 0 = a length byte follows (0 there: the end, s.done).
// Resumable between tokens: s.bits = matches still to come in the current group.  A distance beyond the 4 KiB window is a
// bad token (see the oracle).
template <class SK>
__device__ __forceinline__ bool shrek_field(InCache& in, SK& sk, DecState& s, u32 src_len, u32 flag, u32& v) {
    v = flag >> 3;
    if (v == 0x1Eu) { if (s.p >= src_len) { s.eof = true; return false; } sk.ensure(in, s.p, 1); v += in.peek1(s.p); s.p++; }
    else if (v == 0x1Fu) {
        if (s.p + 2u > src_len) { s.eof = true; s.p = src_len; return false; }
        sk.ensure(in, s.p, 4); v = 286u + (in.peek4(s.p) & 0xFFFFu); s.p += 2;
    }
    return true;
}
template <class SK>
__device__ __forceinline__ void dec_lzshrek_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 max_tokens = 0xFFFFFFFFu) {
    for (;;) {
        if (max_tokens-- == 0) return;
        if (s.bits == 0) {                                                           // group start
            if (s.p >= src_len) { s.eof = true; return; }                            // while (sourcePointer < source.Length) ... throw  :80, :118
            sk.ensure(in, s.p, 8);
            const u32 flag = in.peek1(s.p); s.p++;
            u32 unc;
            if (!shrek_field(in, sk, s, src_len, flag, unc)) return;
            s.bits = (flag & 7u) + 1u;
            if (unc != 0u) {
                if (unc > src_len - s.p) { s.eof = true; return; }                   // Slice throws  :88
                if (!sk.run(in, s.p, unc)) return;
                s.p += unc;
                continue;                                                            // (the run counts as one token)
            }
        }
        if (s.p >= src_len) { s.eof = true; return; }
        sk.ensure(in, s.p, 8);
        const u32 flag = in.peek1(s.p); s.p++;
        u32 length = flag & 7u;
        if (length == 0u) {
            if (s.p >= src_len) { s.eof = true; return; }
            length = in.peek1(s.p); s.p++;
            if (length == 0u) { s.done = true; return; }                             // end  :100-107
            length += 7u;
        }
        u32 d;
        if (!shrek_field(in, sk, s, src_len, flag, d)) return;
        if (d + 1u > 0x1000u) { s.bad = true; return; }
        s.bits--;
        if (!sk.match(d + 1u, length, 4096)) return;
    }
}

// WFLZ.DecompressHeaderless  WayForward/WFLZ.cs:130-159: blocks of (u16 distance, length - 4 or 0, literal count), each followed by
// its literals; the match comes first; 0 / 0 / 0 ends the stream (s.done).  Resumable at block boundaries.
template <class SK, bool BIG>
__device__ __forceinline__ void dec_wflz_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 max_tokens = 0xFFFFFFFFu) {
    for (;;) {
        if (max_tokens-- == 0) return;
        if (s.p + 4u > src_len || s.p + 4u < s.p) { s.eof = true; return; }           // Slice / indexer throw  :138-140
        sk.ensure(in, s.p, 8);
        const u32 w = in.peek4(s.p); s.p += 4;
        const u32 dist = BIG ? (((w & 0xFFu) << 8) | ((w >> 8) & 0xFFu)) : (w & 0xFFFFu);
        const u32 length = (w >> 16) & 0xFFu, plain = w >> 24;
        if (length != 0u) { if (!sk.match(dist, (u64)length + 4u, 65536)) return; }
        else if (plain == 0u) { s.done = true; return; }
        if (plain != 0u) {
            if (plain > src_len - s.p) { s.eof = true; return; }                     // Slice throws  :155
            if (!sk.run(in, s.p, plain)) return;
            s.p += plain;
        }
    }
}

// RefPack.DecompressHeaderless  EA/RefPack.cs:177-245: prefix byte selects the form -- 0DDLLLPP D, 10LLLLLL PPDDDDDD D,
// 110DLLPP D D L (0-3 literals, then a match), 111PPPPP (4..112 literals), 111111PP (0-3 literals, the end: s.done).
// Resumable at element boundaries.  Runs to the end token; the declared size is only compared there.
template <class SK>
__device__ __forceinline__ void dec_refpack_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 max_tokens = 0xFFFFFFFFu) {
    for (;;) {
        if (max_tokens-- == 0) return;
        if (s.p >= src_len) { s.eof = true; return; }                                // while (source.Position < source.Length) ... throw
        sk.ensure(in, s.p, 8);
        const u32 w = in.peek4(s.p);
        const u32 prefix = w & 0xFF, d0 = (w >> 8) & 0xFF, d1 = (w >> 16) & 0xFF, d2 = w >> 24;
        u32 plain, length = 0, distance = 0, hdr;
        if ((prefix & 0x80u) == 0u) { hdr = 2; plain = prefix & 3u; length = ((prefix & 0x1Cu) >> 2) + 3u; distance = (((prefix & 0x60u) << 3) | d0) + 1u; }
        else if ((prefix & 0x40u) == 0u) { hdr = 3; plain = d0 >> 6; length = (prefix & 0x3Fu) + 4u; distance = (((d0 & 0x3Fu) << 8) | d1) + 1u; }
        else if ((prefix & 0x20u) == 0u) { hdr = 4; plain = prefix & 3u; length = (((prefix & 0x0Cu) << 6) | d2) + 5u; distance = (((((prefix & 0x10u) << 4) | d0) << 8) | d1) + 1u; }
        else { hdr = 1; plain = (prefix & 0x1Fu) * 4u + 4u; }
        if (hdr > src_len - s.p) { s.eof = true; s.p = src_len; return; }             // ReadUInt8 throws
        s.p += hdr;
        const bool end = hdr == 1u && plain > 0x70u;
        if (end) plain = prefix & 3u;
        if (plain > src_len - s.p) { s.eof = true; return; }                         // LzWindows.CopyFrom -> ReadExactly throws
        if (!sk.run(in, s.p, plain)) return;
        s.p += plain;
        if (end) { s.done = true; return; }
        if (length && !sk.match(distance, length, 131072)) return;
    }
}

// LZ02.DecompressHeaderless  Camelot/LZ02.cs:77-115: flags MSB first, 1 = match: b1 b2 = DDDDLLLL DDDDDDDD, length = nibble + 1;
// nibble 0: distance 0 is the terminator (s.done), otherwise a third byte holds length - 17.  Runs until the terminator --
// the declared size is only compared there -- or until the input ends (EndOfStreamException).
template <class SK>
__device__ void dec_lz02_serial(InCache& in, SK& sk, DecState& s, u32 src_len) {
    for (;;) {
        if (s.p >= src_len) { s.eof = true; return; }                                // while (source.Position < source.Length) ... throw  :83, :114
        sk.ensure(in, s.p, 8);
        if (s.bits == 0) { s.flag = in.peek1(s.p); s.p++; s.bits = 8; }
        const u32 bit = (s.flag >> (s.bits - 1)) & 1u; s.bits--;
        if (bit) {
            if (s.p + 2 > src_len) { s.eof = true; s.p = src_len; return; }
            const u32 w = in.peek4(s.p); s.p += 2;
            const u32 b1 = w & 0xFF, b2 = (w >> 8) & 0xFF;
            const u32 distance = ((b1 & 0xF0u) << 4) | b2; u32 length = (b1 & 0xFu) + 1u;
            if (length == 1u) {
                if (distance == 0u) { s.done = true; return; }
                if (s.p >= src_len) { s.eof = true; return; }
                length = ((w >> 16) & 0xFF) + 17u; s.p++;
            }
            if (!sk.match(distance, length, 4096)) return;
        } else {
            if (s.p >= src_len) { s.eof = true; return; }
            const u32 b = in.peek1(s.p); s.p++;
            if (!sk.lit(b)) return;
        }
    }
}

// CNS.DecompressHeaderless  Specialized/CNS.cs:77-108: control byte c < 0x80 = c literals, else a match of (c & 0x7F) + 3 bytes at
// distance next byte + 1 (256-byte window).  Resumable at element boundaries.
template <class SK>
__device__ __forceinline__ void dec_cns_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 size, u32 max_tokens = 0xFFFFFFFFu) {
    while (sk.produced() < size) {
        if (max_tokens-- == 0) return;
        sk.ensure(in, s.p, 8);
        if (s.p >= src_len) { s.eof = true; return; }
        const u32 c = in.peek1(s.p); s.p++;
        if ((c & 0x80u) == 0u) {
            if (c > src_len - s.p) { s.eof = true; return; }                         // LzWindows.CopyFrom -> ReadExactly throws
            if (!sk.run(in, s.p, c)) return;
            s.p += c;
        } else {
            if (s.p >= src_len) { s.eof = true; return; }
            const u32 d = in.peek1(s.p); s.p++;
            if (!sk.match(d + 1u, (u64)(c & 0x7Fu) + 3u, 256)) return;
        }
    }
}

// CNX2.DecompressHeaderless  Sega/CNX2.cs:83-139: two flag bits per token, first bit = bit 0 (FlagReader Endian.Little,
// ReadInt(2)  FlagReader.cs:75-87).  0: skip n bytes and drop the rest of the flag byte; 1: literal; 2: match (big-endian
// u16: distance - 1 in the high 11 bits, length - 4 in the low 5); 3: n literals.  Resumable at token boundaries
// (s.flag / s.bits hold the flag byte, shifted as bits are consumed).
template <class SK>
__device__ __forceinline__ void dec_cnx2_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32 size, u32 max_tokens = 0xFFFFFFFFu) {
    while (sk.produced() < size) {
        if (max_tokens-- == 0) return;
        sk.ensure(in, s.p, 8);
        if (s.bits == 0) { if (s.p >= src_len) { s.eof = true; return; } s.flag = in.peek1(s.p); s.p++; s.bits = 8; }
        const u32 code = s.flag & 3u; s.flag >>= 2; s.bits -= 2;
        if (code == 0u) {
            if (s.p >= src_len) { s.eof = true; return; }
            const u32 n = in.peek1(s.p); s.p += 1u + n;                              // source.Position += length (may pass the end)  :98
            s.bits = 0;                                                              // flag.Reset()
        } else if (code == 1u) {
            if (s.p >= src_len) { s.eof = true; return; }
            if (!sk.run(in, s.p, 1)) return;
            s.p++;
        } else if (code == 2u) {
            if (s.p >= src_len || s.p + 2u > src_len) { s.eof = true; return; }
            const u32 w = in.peek4(s.p); s.p += 2;
            const u32 pair = ((w & 0xFFu) << 8) | ((w >> 8) & 0xFFu);
            if (!sk.match((pair >> 5) + 1u, (u64)(pair & 0x1Fu) + 4u, 2048)) return;
        } else {
            if (s.p >= src_len) { s.eof = true; return; }
            const u32 n = in.peek1(s.p); s.p++;
            if (n > src_len - s.p) { s.eof = true; return; }                         // LzWindows.CopyFrom -> ReadExactly throws
            if (!sk.run(in, s.p, n)) return;
            s.p += n;
        }
    }
}

// FastLZ.DecompressHeaderless  Formats/Common/FastLZ.cs:54-160 (levels 1 and 2).  Resumable at token boundaries: `fz`
// carries the level (top bits of the stream's first byte) and s.p always points at the next control byte; s.done is set
// when the input ends behind a token (the only way a FastLZ stream ends, :96).
struct FastlzState { u32 level; bool started; };
__device__ __forceinline__ void fastlz_state_init(FastlzState& fz) { fz.level = 0; fz.started = false; }
__device__ __forceinline__ u32 fastlz_window(const FastlzState& fz) { return fz.level == 2u ? 131072u : 8192u; }   // _lz1 / _lz2[1] WindowsBits  :22-27

template <class SK>
__device__ __forceinline__ void dec_fastlz_serial(InCache& in, SK& sk, DecState& s, u32 src_len, FastlzState& fz, u32 max_tokens = 0xFFFFFFFFu) {
    if (!fz.started) {
        if (src_len == 0) { s.eof = true; return; }                                  // source[0]: IndexOutOfRangeException  :57
        sk.ensure(in, 0, 1);
        fz.level = (in.peek1(0) >> 5) + 1u;
        if (fz.level != 1u && fz.level != 2u) { s.bad = true; return; }             // InvalidDataException  :60
    }
    for (;;) {
        if (max_tokens-- == 0) return;
        sk.ensure(in, s.p, 8);
        u32 ctrl = in.peek1(s.p); s.p++;
        if (!fz.started) { ctrl &= 31u; fz.started = true; }                         // :67 / :109
        if (ctrl >= 32u) {
            u64 len = (ctrl >> 5) - 1u; u32 ofs = (ctrl & 31u) << 8;
            if (len == 6u) {
                if (fz.level == 1u) { if (s.p >= src_len) { s.eof = true; return; } len += in.peek1(s.p); s.p++; }
                else { u32 b; do { if (s.p >= src_len) { s.eof = true; return; } sk.ensure(in, s.p, 1); b = in.peek1(s.p); s.p++; len += b; } while (b == 255u); }
            }
            if (s.p >= src_len) { s.eof = true; return; }
            sk.ensure(in, s.p, 4);
            ofs |= in.peek1(s.p); s.p++;
            if (fz.level == 2u && ofs == 0x1FFFu) {                                  // large offset extension  :138-143
                if (s.p + 2u > src_len) { s.eof = true; return; }
                const u32 w = in.peek4(s.p); s.p += 2;
                ofs = (((w & 0xFFu) << 8) | ((w >> 8) & 0xFFu)) + 0x1FFFu;
            }
            if (!sk.match(ofs + 1u, len + 3u, fastlz_window(fz))) return;
        } else {
            ctrl++;
            if (ctrl > src_len - s.p) { s.eof = true; return; }                      // Slice throws  :91
            if (!sk.run(in, s.p, ctrl)) return;
            s.p += ctrl;
        }
        if (s.p >= src_len) { s.done = true; return; }                               // :96
    }
}

// LZO.DecompressHeaderless  Formats/Common/LZO.cs:49-139.  Resumable at instruction boundaries: `ls` carries the one
// piece of decoder state that crosses instructions (the literal count of the previous instruction, LZO.cs:55) and s.p
// always points at the next instruction's first byte.
struct LzoState { u32 plain; bool started; };
__device__ __forceinline__ void lzo_state_init(LzoState& ls) { ls.plain = 0; ls.started = false; }

template <class SK>
__device__ __forceinline__ void dec_lzo_serial(InCache& in, SK& sk, DecState& s, u32 src_len, LzoState& ls, u32 max_instr = 0xFFFFFFFFu) {
    // residency is established once per instruction (16 bytes cover every fixed-size header); only the unbounded
    // length-extension loop asks again -- one ensure() per byte made this parser 50 KB of code under QueueSink
#define LZO_BYTE(dst) do { if (s.p >= src_len) { s.eof = true; return; } dst = in.peek1(s.p); s.p++; } while (0)
#define LZO_EXT(dst) do { u32 b_, acc_ = 0; for (;;) { sk.ensure(in, s.p, 1); LZO_BYTE(b_); if (b_ != 0) break; acc_ += 255; } sk.ensure(in, s.p, 16); dst = acc_ + b_; } while (0)
    u32 flag, length, distance;
    if (!ls.started) {                                    // the first byte may announce an initial literal run  LZO.cs:58-66
        ls.started = true;
        sk.ensure(in, s.p, 16);
        LZO_BYTE(flag);
        if (flag > 17) {
            length = flag - 17;
            if (length > src_len - s.p) { s.eof = true; return; }
            if (!sk.run(in, s.p, length)) return;
            s.p += length;
        } else s.p--;                                     // an ordinary instruction: read again below
    }
    for (;;) {
        if (max_instr-- == 0) return;
        sk.ensure(in, s.p, 16);
        LZO_BYTE(flag);                                   // ReadByte() == -1 -> EndOfStreamException  LZO.cs:136-137
        u32 plain = ls.plain;
        u32 flagcode = flag >> 4;
        bool literal_op = false;
        if (flagcode == 0) {
            if (plain == 0) {
                length = 3 + flag;
                if (length == 3) { u32 e; LZO_EXT(e); length = 18 + e; }
                ls.plain = 4;
                if (length > src_len - s.p) { s.eof = true; return; }
                if (!sk.run(in, s.p, length)) return;
                s.p += length;
                literal_op = true;
            } else {
                u32 d; LZO_BYTE(d);
                if (plain <= 3) { distance = (d << 2) + (flag >> 2) + 1; length = 2; }
                else { distance = (d << 2) + (flag >> 2) + (2048 + 1); length = 3; }
            }
        } else if (flagcode <= 3) {
            const bool m4 = flagcode == 1;
            const u32 lm = m4 ? 0x7u : 0x1fu;
            length = 2 + (flag & lm);
            if (length == 2) { u32 e; LZO_EXT(e); length = (m4 ? 9u : 33u) + e; }
            const u32 base = m4 ? 16384u + ((flag & 0x8) << 11) : 0u;
            LZO_BYTE(flag);
            u32 hi; LZO_BYTE(hi);
            if (m4) { distance = base | (hi << 6) | (flag >> 2); if (distance == 16384) { s.done = true; return; } }
            else distance = ((hi << 6) | (flag >> 2)) + 1;
        } else {
            u32 d; LZO_BYTE(d);
            if (flagcode <= 7) { length = 3 + ((flag >> 5) & 0x1); distance = (d << 3) + ((flag >> 2) & 0x7) + 1; }
            else { length = 5 + ((flag >> 5) & 0x3); distance = (d << 3) + ((flag & 0x1c) >> 2) + 1; }
        }
        if (!literal_op) {
            plain = flag & 0x3;
            ls.plain = plain;
            if (!sk.match(distance, length, 65536)) return;
            if (plain > src_len - s.p) { s.eof = true; return; }
            if (!sk.run(in, s.p, plain)) return;
            s.p += plain;
        }
    }
#undef LZO_BYTE
#undef LZO_EXT
}

// Snappy.DecompressHeaderless  Formats/Common/Snappy.cs:205-250 (+ varint :109-122)
template <class SK>
__device__ void dec_snappy_serial(InCache& in, SK& sk, DecState& s, u32 src_len, u32& size, bool& have_size, u32 max_tokens = 0xFFFFFFFFu) {
#define SN_BYTE(dst) do { if (s.p >= src_len) { s.eof = true; return; } sk.ensure(in, s.p, 1); dst = in.peek1(s.p); s.p++; } while (0)
    if (!have_size) {                                                     // ReadDecompressedSize  Snappy.cs:109-122
        u32 shift = 0, b = 0x80; size = 0;
        while (b & 0x80) { SN_BYTE(b); size |= (b & 0x7F) << (shift & 31); shift += 7; }
        have_size = true;
    }
    while (sk.produced() < size) {
        if (max_tokens-- == 0) return;
        u32 tag; SN_BYTE(tag);
        u32 type = tag & 3, length = tag >> 2, distance;
        if (type == 0) {
            if (length >= 60) {
                u32 lenBytes = length - 59; length = 0;
                for (u32 i = 0; i < lenBytes; i++) { u32 x; SN_BYTE(x); length |= x << (8 * i); }
            }
            u32 run = length + 1;
            if (run > src_len - s.p) { s.eof = true; return; }
            if (!sk.run(in, s.p, run)) return;
            s.p += run;
            continue;
        } else if (type == 1) {
            length = (length & 0x7) + 3;
            u32 x; SN_BYTE(x); distance = ((tag >> 5) << 8) | x;
        } else if (type == 2) {
            if (s.p + 2 > src_len) { s.eof = true; return; }
            sk.ensure(in, s.p, 4); distance = in.peek4(s.p) & 0xFFFF; s.p += 2;
        } else {
            if (src_len - s.p < 4) { s.eof = true; return; }
            sk.ensure(in, s.p, 4); distance = in.peek4(s.p); s.p += 4;
            if (distance > 65536u) { s.bad = true; return; }    // E3
        }
        if (!sk.match(distance, (u64)length + 1, 65536)) return;
    }
#undef SN_BYTE
}

// status resolution (precedence: truncated > bad token > E5 capacity/size > overshoot), see DESIGN.md "Status rules"
__device__ __forceinline__ int resolve_status(const DecState& s, bool has_size, u32 produced, u32 size, u32 cap) {
    if (s.eof) return ALZ_ST_INPUT_TRUNCATED;
    if (s.bad) return ALZ_ST_BAD_TOKEN;
    if (s.ovf) return (has_size && s.attempted_end > (u64)size && cap >= size) ? ALZ_ST_OUTPUT_SIZE_MISMATCH : ALZ_ST_OUTPUT_CAPACITY;
    if (has_size && produced > size) return ALZ_ST_OUTPUT_SIZE_MISMATCH;
    return ALZ_ST_OK;
}
