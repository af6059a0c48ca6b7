/*
 * alz_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See alz_oracle.h for the rules.  Plain C11, no dependencies.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against the
 * reference's own fixtures (Test.lz KAT XXH64 11520079745250749767, round-trip
 * matrix, published ratios).  Paths below are relative to /root/reference/src.
 *
 * Frozen definitions of reference-undefined behaviour:
 *  E1  distance == 0            -> behaves as distance == W (self copy of the ring slot)
 *  E2  source before stream start -> 0x00 (the rented ring is treated as zero-filled)
 *  E3  distance > W             -> only reachable by Snappy copy-4; refused with ALZ_ST_BAD_TOKEN
 *  E4  last match overshoots the declared size -> bytes are written (up to dst_cap), then SIZE_MISMATCH
 *  E5  decoding stops at the first token whose output would exceed dst_cap: the
 *      bytes below dst_cap are written, status is SIZE_MISMATCH if the format has
 *      a declared size, the token crosses it and dst_cap >= decom_len, else CAPACITY.
 *  E6  a read past the end of the input ends decoding with INPUT_TRUNCATED at
 *      that read (a literal run that does not fit the input is not copied).
 */
#include "alz_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ============================================================ hashes */

#define P64_1 11400714785074694791ULL
#define P64_2 14029467366897019727ULL
#define P64_3 1609587929392839161ULL
#define P64_4 9650029242287828579ULL
#define P64_5 2870177450012600261ULL

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint64_t rd64le(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32le(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

static inline uint64_t xxh64_round(uint64_t acc, uint64_t in) {
    acc += in * P64_2; acc = rotl64(acc, 31); acc *= P64_1; return acc;
}
static inline uint64_t xxh64_merge(uint64_t acc, uint64_t v) {
    v = xxh64_round(0, v); acc ^= v; acc = acc * P64_1 + P64_4; return acc;
}

uint64_t oracle_xxh64(const void* data, size_t len, uint64_t seed) {
    const uint8_t* p = (const uint8_t*)data; const uint8_t* end = p + len; uint64_t h;
    if (len >= 32) {
        const uint8_t* lim = end - 32;
        uint64_t v1 = seed + P64_1 + P64_2, v2 = seed + P64_2, v3 = seed, v4 = seed - P64_1;
        do {
            v1 = xxh64_round(v1, rd64le(p)); p += 8; v2 = xxh64_round(v2, rd64le(p)); p += 8;
            v3 = xxh64_round(v3, rd64le(p)); p += 8; v4 = xxh64_round(v4, rd64le(p)); p += 8;
        } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxh64_merge(h, v1); h = xxh64_merge(h, v2); h = xxh64_merge(h, v3); h = xxh64_merge(h, v4);
    } else {
        h = seed + P64_5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xxh64_round(0, rd64le(p)); h = rotl64(h, 27) * P64_1 + P64_4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32le(p) * P64_1; h = rotl64(h, 23) * P64_2 + P64_3; p += 4; }
    while (p < end) { h ^= (*p) * P64_5; h = rotl64(h, 11) * P64_1; p++; }
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
    return h;
}

#define P32_1 2654435761U
#define P32_2 2246822519U
#define P32_3 3266489917U
#define P32_4 668265263U
#define P32_5 374761393U

uint32_t oracle_xxh32(const void* data, size_t len, uint32_t seed) {
    const uint8_t* p = (const uint8_t*)data; const uint8_t* end = p + len; uint32_t h;
    if (len >= 16) {
        const uint8_t* lim = end - 16;
        uint32_t v1 = seed + P32_1 + P32_2, v2 = seed + P32_2, v3 = seed, v4 = seed - P32_1;
        do {
            v1 = rotl32(v1 + rd32le(p) * P32_2, 13) * P32_1; p += 4;
            v2 = rotl32(v2 + rd32le(p) * P32_2, 13) * P32_1; p += 4;
            v3 = rotl32(v3 + rd32le(p) * P32_2, 13) * P32_1; p += 4;
            v4 = rotl32(v4 + rd32le(p) * P32_2, 13) * P32_1; p += 4;
        } while (p <= lim);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + P32_5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + rd32le(p) * P32_3, 17) * P32_4; p += 4; }
    while (p < end) { h = rotl32(h + (*p) * P32_5, 11) * P32_1; p++; }
    h ^= h >> 15; h *= P32_2; h ^= h >> 13; h *= P32_3; h ^= h >> 16;
    return h;
}

/* CRC-32C (Castagnoli), table driven; Formats/../CRC32c.cs is the reference's own. */
uint32_t oracle_crc32c(const void* data, size_t len) {
    static uint32_t table[256]; static int init = 0;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78U : c >> 1;
            table[i] = c;
        }
        init = 1;
    }
    uint32_t crc = 0xFFFFFFFFU; const uint8_t* p = (const uint8_t*)data;
    for (size_t i = 0; i < len; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc ^ 0xFFFFFFFFU;
}

/* ============================================================ geometry */

/* LzProperties(byte distanceBits, byte lengthBits, byte threshold)  LzProperties.cs:57-66 */
void oracle_lz_properties_bits(uint8_t distance_bits, uint8_t length_bits, uint8_t threshold, alz_lz_properties* out) {
    memset(out, 0, sizeof(*out));
    out->window_bits = distance_bits;
    out->length_bits = length_bits;
    out->min_length = (uint8_t)(threshold + 1);
    out->max_distance = 1u << distance_bits;
    out->windows_start = out->max_distance - (1u << length_bits) - threshold;
}

static alz_lz_properties lzss_effective(const alz_lz_properties* p) {
    alz_lz_properties d;
    if (!p || p->window_bits == 0) oracle_lz_properties_bits(12, 4, 2, &d); /* LZSS.DefaultProperties LZSS.cs:33 */
    else { d = *p; if (d.max_distance == 0) d.max_distance = 1u << d.window_bits; }
    return d;
}

/* ============================================================ source cursor */

typedef struct { const uint8_t* p; uint32_t len; uint32_t pos; int eof; } cur_t;

/* Stream.ReadUInt8(): throws EndOfStreamException at EOF (AuroraLib.Core) */
static inline int cur_u8(cur_t* c) { if (c->pos >= c->len) { c->eof = 1; return 0; } return c->p[c->pos++]; }
/* Stream.ReadByte(): -1 at EOF */
static inline int cur_byte(cur_t* c) { if (c->pos >= c->len) return -1; return c->p[c->pos++]; }

/* ============================================================ LzWindows (IO/LzWindows.cs:15-280) */

typedef struct {
    int flat;           /* 0: ring + flush exactly as LzWindows; 1: flat out[q]=out[q-d] model */
    uint8_t* ring; uint32_t W, mask, pos;
    uint8_t* dst; uint64_t cap;
    uint64_t flushed;   /* destination.Position (ring mode) / produced (flat mode) */
    int overflow; uint64_t attempted_end;
} win_t;

static inline uint64_t win_produced(const win_t* w) { return w->flat ? w->flushed : w->flushed + w->pos; }

/* E5: clip a token of `len` bytes against dst_cap */
static inline uint32_t win_clip(win_t* w, uint64_t len) {
    uint64_t pr = win_produced(w);
    if (pr + len > w->cap) { w->overflow = 1; w->attempted_end = pr + len; return (uint32_t)(w->cap - pr); }
    return (uint32_t)len;
}

/* FlushToDestination / _Destination.Write  LzWindows.cs:219,247 */
static inline void ring_out(win_t* w, const uint8_t* buf, uint32_t len) {
    memcpy(w->dst + w->flushed, buf, len); /* never beyond cap: tokens are clipped first */
    w->flushed += len;
}

/* InternWrite  LzWindows.cs:192-227 */
static void ring_intern_write(win_t* w, const uint8_t* src, uint32_t len) {
    if (!len) return;
    uint32_t pos = w->pos, W = w->W;
    if (W > pos + len) {
        memmove(w->ring + pos, src, len);
        w->pos += len;
    } else {
        uint32_t left = W - pos, remaining = len - left;
        if (left > 0) memmove(w->ring + pos, src, left);
        ring_out(w, w->ring, W);
        if (remaining != 0) memmove(w->ring, src + len - remaining, remaining);
        w->pos = remaining;
    }
}

/* BackCopy  LzWindows.cs:72-100 (len already clipped by the caller) */
static void win_back_copy(win_t* w, uint32_t distance, uint32_t length) {
    if (w->flat) {
        uint32_t d = distance ? distance : w->W; /* E1 */
        uint64_t q = w->flushed;
        for (uint32_t i = 0; i < length; i++, q++) w->dst[q] = (q >= d) ? w->dst[q - d] : 0; /* E2 */
        w->flushed = q;
        return;
    }
    int64_t len = length;
    while (len > 0) {
        uint32_t chunk = (uint32_t)len;
        uint32_t srcPos = (w->pos - distance) & w->mask;
        if (distance < (uint32_t)len && distance != 0) chunk = distance;
        if (srcPos + chunk > w->W) chunk = w->W - srcPos;
        ring_intern_write(w, w->ring + srcPos, chunk);
        len -= chunk;
    }
}

/* OffsetCopy  LzWindows.cs:108-115 */
static void win_offset_copy(win_t* w, uint32_t offset, uint32_t length) {
    uint32_t pos = w->flat ? (uint32_t)(w->flushed & w->mask) : w->pos;
    uint32_t distance = pos >= offset ? pos - offset : pos - offset + w->W;
    win_back_copy(w, distance, length);
}

/* WriteByte  LzWindows.cs:232-237 */
static inline void win_write_byte(win_t* w, uint8_t v) {
    if (w->flat) { w->dst[w->flushed++] = v; return; }
    w->ring[w->pos] = v;
    w->pos = (w->pos + 1) & w->mask;
    if (w->pos == 0) ring_out(w, w->ring, w->W);
}

/* Write(span)  LzWindows.cs:168-190 */
static void win_write(win_t* w, const uint8_t* buf, uint32_t len) {
    if (w->flat) { memcpy(w->dst + w->flushed, buf, len); w->flushed += len; return; }
    uint32_t W = w->W;
    if (len < W) { ring_intern_write(w, buf, len); return; }
    uint32_t off = 0;
    while (len - off >= W) { ring_intern_write(w, buf + off, W); off += W; }
    if (off < len) ring_intern_write(w, buf + off, len - off);
}

/* CopyFrom(Stream, length)  LzWindows.cs:124-135; caller guarantees the input holds `length` bytes (E6) */
static void win_copy_from(win_t* w, cur_t* c, uint32_t length) {
    if (w->flat) { memcpy(w->dst + w->flushed, c->p + c->pos, length); w->flushed += length; c->pos += length; return; }
    while (length != 0) {
        uint32_t l = w->W - w->pos; if (length < l) l = length;
        memcpy(w->ring + w->pos, c->p + c->pos, l); c->pos += l;
        w->pos = (w->pos + l) & w->mask;
        length -= l;
        if (w->pos == 0) ring_out(w, w->ring, w->W);
    }
}

/* Dispose  LzWindows.cs:269-278 */
static void win_dispose(win_t* w) {
    if (!w->flat && w->pos != 0) { ring_out(w, w->ring, w->pos); w->pos = 0; }
}

/* ============================================================ FlagReader (IO/FlagReader.cs:12-103), 8-bit flags */

typedef struct { cur_t* c; int bits_left; int cur; int msb_first; int nbytes; /* 0/1: byte flags; 2 / 4: big-endian flag words */ } flag_t;

/* Readbit  FlagReader.cs:53-65 */
static inline int flag_readbit(flag_t* f) {
    const int width = f->nbytes > 1 ? 8 * f->nbytes : 8;
    if (f->bits_left == 0) {                                                             /* ReadNextFlag  FlagReader.cs:40-47 */
        if (f->nbytes > 1) { uint32_t v = 0; for (int i = 0; i < f->nbytes; i++) v = (v << 8) | (uint32_t)cur_u8(f->c); f->cur = (int)v; }
        else f->cur = cur_u8(f->c);
        f->bits_left = width;
    }
    int shift = f->msb_first ? f->bits_left - 1 : width - f->bits_left;
    f->bits_left--;
    return (int)(((uint32_t)f->cur >> shift) & 1u);
}

/* ============================================================ decoders */

typedef struct { int has_size; int bad_token; uint32_t src_used; } dec_info;

/* LZSS.DecompressHeaderless  Formats/Common/LZSS.cs:91-130 */
static void dec_lzss(const alz_lz_properties* lz, cur_t* c, win_t* w, uint32_t size) {
    flag_t flag = { c, 0, 0, 0, 1 };                     /* FlagReader(source, Endian.Little)  :95 */
    uint32_t f = (1u << lz->length_bits) - 1;            /* GetLengthBitsFlag */
    uint32_t n = lz->max_distance - 1;                   /* GetWindowsFlag */
    while (win_produced(w) < size) {                     /* :104 */
        int bit = flag_readbit(&flag); if (c->eof) return;
        if (bit) {
            int b = cur_u8(c); if (c->eof) return;
            if (win_clip(w, 1) < 1) return;
            win_write_byte(w, (uint8_t)b);               /* :107 */
        } else {
            int b1 = cur_u8(c); if (c->eof) return;
            int b2 = cur_u8(c); if (c->eof) return;
            uint32_t offset = ((uint32_t)(b2 >> lz->length_bits) << 8) | (uint32_t)b1;   /* :115 */
            uint32_t length = ((uint32_t)b2 & f) + lz->min_length;                       /* :116 */
            offset = (lz->max_distance + offset - lz->windows_start) & n;                /* :117 */
            uint32_t cl = win_clip(w, length);
            win_offset_copy(w, offset, cl);                                              /* :119 */
            if (w->overflow) return;
        }
    }
}

/* LZ10.DecompressHeaderless  Nintendo/LZ10.cs:82-111 */
static void dec_lz10(cur_t* c, win_t* w, uint32_t size) {
    flag_t flag = { c, 0, 0, 1, 1 };                        /* FlagReader(source, Endian.Big) :88 */
    while (win_produced(w) < size) {
        int bit = flag_readbit(&flag); if (c->eof) return;
        if (bit) {
            int b1 = cur_u8(c); if (c->eof) return;
            int b2 = cur_u8(c); if (c->eof) return;
            uint32_t distance = (uint32_t)(((b1 & 0xf) << 8) | b2) + 1;                  /* :96 */
            uint32_t length = (uint32_t)(b1 >> 4) + 3;                                   /* :97 */
            uint32_t cl = win_clip(w, length);
            win_back_copy(w, distance, cl);
            if (w->overflow) return;
        } else {
            int b = cur_u8(c); if (c->eof) return;
            if (win_clip(w, 1) < 1) return;
            win_write_byte(w, (uint8_t)b);                                               /* :102 */
        }
    }
}

/* LZ11.DecompressHeaderless  Nintendo/LZ11.cs:83-133 */
static void dec_lz11(cur_t* c, win_t* w, uint32_t size) {
    flag_t flag = { c, 0, 0, 1, 1 };
    while (win_produced(w) < size) {
        int bit = flag_readbit(&flag); if (c->eof) return;
        if (bit) {
            uint32_t distance, length;
            int b1 = cur_u8(c); if (c->eof) return;
            int b2 = cur_u8(c); if (c->eof) return;
            if ((b1 >> 4) == 0) {                                                         /* :98-104 */
                int b3 = cur_u8(c); if (c->eof) return;
                distance = (uint32_t)(((b2 & 0xf) << 8) | b3) + 1;
                length = (uint32_t)(((b1 & 0xf) << 4) | (b2 >> 4)) + 17;
            } else if ((b1 >> 4) == 1) {                                                  /* :105-112 */
                int b3 = cur_u8(c); if (c->eof) return;
                int b4 = cur_u8(c); if (c->eof) return;
                distance = (uint32_t)(((b3 & 0xf) << 8) | b4) + 1;
                length = (uint32_t)(((b1 & 0xf) << 12) | (b2 << 4) | (b3 >> 4)) + 273;
            } else {                                                                      /* :113-118 */
                distance = (uint32_t)(((b1 & 0xf) << 8) | b2) + 1;
                length = (uint32_t)(b1 >> 4) + 1;
            }
            uint32_t cl = win_clip(w, length);
            win_back_copy(w, distance, cl);
            if (w->overflow) return;
        } else {
            int b = cur_u8(c); if (c->eof) return;
            if (win_clip(w, 1) < 1) return;
            win_write_byte(w, (uint8_t)b);
        }
    }
}

/* LZ40.DecompressHeaderless  Nintendo/LZ40.cs:80-132: the flag byte is stored negated (:92), bits MSB first, 1 = match;
   tokens are u16 LE distance << 4 | length nibble, nibble 0 / 1 = one / two more length bytes. */
static void dec_lz40(cur_t* c, win_t* w, uint32_t size) {
    int flag = 0, flagbits = 0;
    while (win_produced(w) < size) {
        if (flagbits == 0) {
            int b = cur_byte(c);                                                          /* ReadByte(): -1 at EOF, (byte)-(-1) = 1 */
            flag = (-b) & 0xFF; flagbits = 8;
        }
        if (flag & 0x80) {
            int x0 = cur_u8(c); if (c->eof) return;
            int x1 = cur_u8(c); if (c->eof) return;
            uint32_t v = (uint32_t)x0 | ((uint32_t)x1 << 8), length = v & 0xF, distance = v >> 4;
            if (length == 0) { int e = cur_u8(c); if (c->eof) return; length = (uint32_t)e + 16; }            /* :101-105 */
            else if (length == 1) { int e0 = cur_u8(c); if (c->eof) return; int e1 = cur_u8(c); if (c->eof) return; length = ((uint32_t)e0 | ((uint32_t)e1 << 8)) + 272; }
            uint32_t cl = win_clip(w, length);
            win_back_copy(w, distance, cl);
            if (w->overflow) return;
        } else {
            int b = cur_u8(c); if (c->eof) return;
            if (win_clip(w, 1) < 1) return;
            win_write_byte(w, (uint8_t)b);
        }
        flag <<= 1; flagbits--;
    }
}

/* Yay0.DecompressHeaderless(FlagReader, compressed, uncompressed, dest, len)  Nintendo/Yay0.cs:110-144.
 * Yaz0 passes the same stream for all three cursors (Yaz0.cs:91-92). */
static void dec_yay0_w(cur_t* fc, cur_t* cc, cur_t* uc, win_t* w, uint32_t size, int flag_bytes) {
    flag_t flag = { fc, 0, 0, 1, flag_bytes };
    while (win_produced(w) < size) {
        int bit = flag_readbit(&flag); if (fc->eof) return;
        if (bit) {
            int b = cur_u8(uc); if (uc->eof) return;                                     /* :120 */
            if (win_clip(w, 1) < 1) return;
            win_write_byte(w, (uint8_t)b);
        } else {
            int b1 = cur_u8(cc); if (cc->eof) return;
            int b2 = cur_u8(cc); if (cc->eof) return;
            uint32_t distance = (uint32_t)(((b1 & 0x0F) << 8) | b2) + 1;                 /* :127 */
            int length = b1 >> 4;
            if (length == 0) length = cur_byte(uc) + 0x12;   /* ReadByte(): -1 at EOF => 17   :130-131 */
            else length += 2;                                                            /* :133 */
            uint32_t cl = win_clip(w, (uint32_t)length);
            win_back_copy(w, distance, cl);
            if (w->overflow) return;
        }
    }
}

static void dec_yay0(cur_t* fc, cur_t* cc, cur_t* uc, win_t* w, uint32_t size) { dec_yay0_w(fc, cc, uc, w, size, 1); }

/* SMSR00.DecompressHeaderless(Stream uncompressed, ReadOnlySpan<ushort> codes, ...)  Nintendo/SMSR00.cs:85-131:
   codes = 16-bit big-endian words: a mask (MSB first, 1 = literal) followed by the match words of its 16 tokens */
static void dec_smsr00(cur_t* codes, cur_t* lit, win_t* w, uint32_t size) {
    uint32_t mask = 0; int bits = 0;
    while (win_produced(w) < size) {
        if (bits == 0) {
            if (codes->pos + 2 > codes->len) { codes->eof = 1; return; }                 /* codes[codePointer++]: IndexOutOfRange */
            mask = ((uint32_t)codes->p[codes->pos] << 8) | codes->p[codes->pos + 1]; codes->pos += 2; bits = 16;
        }
        if (mask & 0x8000u) {
            int b = cur_u8(lit); if (lit->eof) return;
            if (win_clip(w, 1) < 1) return;
            win_write_byte(w, (uint8_t)b);
        } else {
            if (codes->pos + 2 > codes->len) { codes->eof = 1; return; }
            uint32_t data = ((uint32_t)codes->p[codes->pos] << 8) | codes->p[codes->pos + 1]; codes->pos += 2;
            uint32_t cl = win_clip(w, (data >> 12) + 3);
            win_back_copy(w, (data & 0x0FFF) + 1, cl);
            if (w->overflow) return;
        }
        mask = (mask << 1) & 0xFFFFu; bits--;
    }
}

/* MIO0.DecompressHeaderless(ReadOnlySpan<byte>, ...)  Nintendo/MIO0.cs:105-149 */
static void dec_mio0(cur_t* c, win_t* w, uint32_t size, uint32_t cptr, uint32_t uptr, uint32_t* used) {
    uint32_t fptr = 0; int maskBits = 0, mask = 0;
    while (win_produced(w) < size) {
        if (maskBits == 0) {
            if (fptr >= c->len) { c->eof = 1; break; }
            mask = c->p[fptr++]; maskBits = 8;                                           /* :117-121 */
        }
        if ((mask & 0x80) == 0x80) {
            if (uptr >= c->len) { c->eof = 1; break; }
            uint8_t b = c->p[uptr++];
            if (win_clip(w, 1) < 1) break;
            win_write_byte(w, b);                                                        /* :125 */
        } else {
            if (cptr >= c->len) { c->eof = 1; break; }
            int b1 = c->p[cptr++];
            if (cptr >= c->len) { c->eof = 1; break; }
            int b2 = c->p[cptr++];
            uint32_t distance = (uint32_t)(((b1 & 0x0F) << 8) | b2) + 1;                 /* :133 */
            uint32_t length = (uint32_t)(b1 >> 4) + 3;                                   /* :134 */
            uint32_t cl = win_clip(w, length);
            win_back_copy(w, distance, cl);
            if (w->overflow) break;
        }
        mask <<= 1; maskBits--;                                                          /* :140-141 */
    }
    *used = cptr > uptr ? cptr : uptr;                                                   /* :148 */
}

/* PRS.DecompressHeaderless(Stream, Stream, Endian)  Sega/PRS.cs:59-102.  Returns 1 when the terminator was read. */
static int dec_prs(cur_t* c, win_t* w, int big) {
    flag_t flag = { c, 0, 0, big, 1 };                      /* FlagReader(source, order): bit order = byte order :62 */
    while (c->pos < c->len) {                            /* :64 */
        int bit = flag_readbit(&flag); if (c->eof) return 0;
        if (bit) {
            int b = cur_u8(c); if (c->eof) return 0;
            if (win_clip(w, 1) < 1) return 0;
            win_write_byte(w, (uint8_t)b);                                               /* :68 */
        } else {
            int distance, length;
            int bit2 = flag_readbit(&flag); if (c->eof) return 0;
            if (bit2) {
                int x0 = cur_u8(c); if (c->eof) return 0;
                int x1 = cur_u8(c); if (c->eof) return 0;
                distance = big ? ((x0 << 8) | x1) : ((x1 << 8) | x0);                    /* ReadUInt16(order) :75 */
                if (distance == 0) return 1;                                             /* :77-80 */
                length = distance & 7;
                distance = 0x2000 - (distance >> 3);                                     /* :83 */
                if (length == 0) { int e = cur_u8(c); if (c->eof) return 0; length = e + 1; } /* :86 */
                else length += 2;                                                        /* :90 */
            } else {
                int v = 0;                                                               /* ReadInt(2, true) FlagReader.cs:88-98 */
                for (int i = 0; i < 2; i++) { v <<= 1; int b = flag_readbit(&flag); if (c->eof) return 0; if (b) v |= 1; }
                length = v + 2;                                                          /* :95 */
                int e = cur_u8(c); if (c->eof) return 0;
                distance = 0x100 - e;                                                    /* :96 */
            }
            uint32_t cl = win_clip(w, (uint32_t)length);
            win_back_copy(w, (uint32_t)distance, cl);
            if (w->overflow) return 0;
        }
    }
    c->eof = 1;                                          /* throw new EndOfStreamException() :101 */
    return 0;
}

/* LZ4.DecompressBlockHeaderless(ReadOnlySpan<byte>, LzWindows)  Formats/Common/LZ4.cs:176-200 */
static void dec_lz4(cur_t* c, win_t* w) {
    const uint8_t* s = c->p; uint32_t n = c->len; uint32_t sp = 0;
    while (sp < n) {
        uint32_t token = s[sp++];
        uint64_t plain = token >> 4;
        if (plain == 0xF) {                              /* ReadExtension :241-252 */
            uint32_t b;
            do { if (sp >= n) { c->eof = 1; c->pos = sp; return; } b = s[sp++]; plain += b; } while (b == 255);
        }
        if (plain > (uint64_t)(n - sp)) { c->eof = 1; c->pos = sp; return; }            /* Slice throws :187 */
        uint32_t cl = win_clip(w, plain);
        win_write(w, s + sp, cl);
        if (w->overflow) { c->pos = sp; return; }
        sp += (uint32_t)plain;
        if (sp >= n) break;                                                              /* :190 */
        uint64_t mlen = token & 0xF;
        if (sp + 2 > n) { c->eof = 1; c->pos = sp; return; }
        uint32_t dist = (uint32_t)s[sp] | ((uint32_t)s[sp + 1] << 8); sp += 2;          /* :195 */
        if (mlen == 0xF) {
            uint32_t b;
            do { if (sp >= n) { c->eof = 1; c->pos = sp; return; } b = s[sp++]; mlen += b; } while (b == 255);
        }
        cl = win_clip(w, mlen + 4);
        win_back_copy(w, dist, cl);                                                      /* :198 */
        if (w->overflow) { c->pos = sp; return; }
    }
    c->pos = sp;
}

/* BLZ.DecompressHeaderless  Nintendo/BLZ.cs:97-135, restated in stream order: both spans are walked from their ends, so
 * here `c` is the code section reversed and the window receives the output reversed (the container layer reverses both).
 * L = length of the destination span.  A match is cut silently where the span ends (:121 `dst > 0`); a literal there, a
 * read behind the input, or a match source beyond the span end are IndexOutOfRangeExceptions; decoding stops when the
 * input is used up and the span must then be full (:131). */
static void dec_blz(cur_t* c, win_t* w, uint32_t size, dec_info* info, int* short_out) {
    const uint8_t* s = c->p; uint32_t n = c->len, sp = 0; uint32_t flags = 0, mask = 0;
    uint64_t L = size < w->cap ? size : w->cap;
    while (sp < n) {                                                                     /* :104 */
        if ((mask >>= 1) == 0) { flags = s[sp++]; mask = 0x80; }
        if ((flags & mask) == 0) {
            if (win_produced(w) >= L) { w->overflow = 1; w->attempted_end = win_produced(w) + 1; c->pos = sp; return; }   /* destination[--dst] */
            if (sp >= n) { c->eof = 1; c->pos = sp; return; }                            /* source[--src] */
            win_write_byte(w, s[sp++]);
        } else {
            if (sp + 2 > n) { c->eof = 1; c->pos = n; return; }
            uint32_t inf = ((uint32_t)s[sp] << 8) | s[sp + 1]; sp += 2;
            uint32_t distance = (inf & 0x0FFF) + 3, length = ((inf >> 12) & 0xF) + 3;
            uint64_t room = L - win_produced(w);
            if (length > room) length = (uint32_t)room;
            if (length && distance > win_produced(w)) { info->bad_token = 1; c->pos = sp; return; }   /* destination[dst - 1 + distance] */
            win_back_copy(w, distance, length);
        }
    }
    c->pos = sp;
    if (win_produced(w) != size) *short_out = 1;                                         /* :131 DecompressedSizeException */
}

/* HIG literal block: count byte + 2, or 0 followed by a u16 LE count (HIG.cs:136-139, :189-193); Stream.ReadByte() == -1 gives a count
 * of 1 that ReadExactly then fails on: truncated either way.  Returns 0 when decoding has to stop. */
static int hig_raw(cur_t* c, win_t* w) {
    int b = cur_u8(c); if (c->eof) return 0;
    uint32_t plain = (uint32_t)b + 2;
    if (plain == 2) {
        int lo = cur_u8(c); if (c->eof) return 0;
        int hi = cur_u8(c); if (c->eof) return 0;
        plain = (uint32_t)lo | ((uint32_t)hi << 8);
    }
    if (c->pos > c->len || plain > c->len - c->pos) { c->eof = 1; return 0; }            /* LzWindows.CopyFrom -> ReadExactly */
    uint32_t cl = win_clip(w, plain);
    win_write(w, c->p + c->pos, cl);
    if (w->overflow) return 0;
    c->pos += plain;
    return 1;
}
/* HIG.DecompressHeaderless  Specialized/HIG.cs:126-212 */
static void dec_hig(cur_t* c, win_t* w, uint32_t size) {
    if (!hig_raw(c, w)) return;                                                          /* initial RAW block  :136-139 */
    while (win_produced(w) < size) {                                                     /* :141 */
        int b = cur_u8(c); if (c->eof) return;
        uint32_t length = (uint32_t)b >> 5, distance, plain;
        if (length < 6) {                                                                /* LLLD DDPP  DDDD DDDD */
            length += 4; distance = ((uint32_t)b & 0x1C) << 6; plain = (uint32_t)b & 3;
        } else {
            if (length == 6) { length = ((uint32_t)b & 0x1F) + 4; distance = 0; }        /* 110L LLLL  DDDD DDPP  DDDD DDDD */
            else {                                                                       /* 111D LLLL ... */
                length = ((uint32_t)b & 0xF) + 3; distance = ((uint32_t)b & 0x10) << 10;
                if (length == 3) {
                    int l = cur_u8(c); if (c->eof) return;
                    length = (uint32_t)l + 18;
                    if (length == 18) { int h = cur_u8(c); if (c->eof) return; int lo = cur_u8(c); if (c->eof) return; length = ((uint32_t)h << 8) | (uint32_t)lo; }
                }
            }
            int b2 = cur_u8(c); if (c->eof) return;
            distance |= ((uint32_t)b2 & 0xFC) << 6; plain = (uint32_t)b2 & 3;
        }
        int last = cur_u8(c); if (c->eof) return;
        distance |= (uint32_t)last;
        uint32_t cl = win_clip(w, length);
        win_back_copy(w, distance, cl);                                                  /* :184 */
        if (w->overflow) return;
        if (plain == 0) { if (!hig_raw(c, w)) return; }                                  /* :187-204 */
        else if (plain < 3) {
            for (uint32_t i = 0; i < plain; i++) {
                int v = cur_u8(c); if (c->eof) return;
                if (win_clip(w, 1) < 1) return;
                win_write_byte(w, (uint8_t)v);
            }
        }
    }
}

/* LZShrek.ReadDistance  Activision/LZShrek.cs:176-190: 0-29 in the flag, 30 = 30 + next byte, 31 = 286 + next u16 LE.  -1: input ended. */
static int64_t shrek_field(const uint8_t* s, uint32_t n, uint32_t* sp, uint32_t flag) {
    uint32_t v = flag >> 3;
    if (v == 0x1E) { if (*sp >= n) return -1; v += s[(*sp)++]; }
    else if (v == 0x1F) { if (*sp + 2 > n) { *sp = n; return -1; } v = 286u + s[*sp] + ((uint32_t)s[*sp + 1] << 8); *sp += 2; }
    return v;
}
/* LZShrek.DecompressHeaderless  Activision/LZShrek.cs:73-119 (span based).  Returns 1 at the end marker.  A distance beyond the
 * 4 KiB window is encodable (up to 65 822) but the managed decoder wraps it around its ring: refused as a bad token. */
static int dec_lzshrek(cur_t* c, win_t* w, dec_info* info) {
    const uint8_t* s = c->p; uint32_t n = c->len, sp = 0;
    while (sp < n) {                                                                     /* :80 */
        uint32_t flag = s[sp++];
        uint32_t compressed = (flag & 7) + 1;
        int64_t unc = shrek_field(s, n, &sp, flag);
        if (unc < 0) { c->eof = 1; c->pos = sp; return 0; }
        if (unc != 0) {
            if ((uint64_t)unc > n - sp) { c->eof = 1; c->pos = sp; return 0; }           /* Slice throws  :88 */
            uint32_t cl = win_clip(w, (uint32_t)unc);
            win_write(w, s + sp, cl);
            if (w->overflow) { c->pos = sp; return 0; }
            sp += (uint32_t)unc;
        }
        for (uint32_t i = 0; i < compressed; i++) {
            if (sp >= n) { c->eof = 1; c->pos = sp; return 0; }
            flag = s[sp++];
            uint32_t length = flag & 7;
            if (length == 0) {
                if (sp >= n) { c->eof = 1; c->pos = sp; return 0; }
                length = s[sp++];
                if (length == 0) { c->pos = sp; return 1; }                              /* end  :100-107 */
                length += 7;
            }
            int64_t d = shrek_field(s, n, &sp, flag);
            if (d < 0) { c->eof = 1; c->pos = sp; return 0; }
            if (d + 1 > 0x1000) { info->bad_token = 1; c->pos = sp; return 0; }
            uint32_t cl = win_clip(w, length);
            win_back_copy(w, (uint32_t)d + 1, cl);                                       /* :113-115 */
            if (w->overflow) { c->pos = sp; return 0; }
        }
    }
    c->eof = 1; c->pos = sp;                                                             /* :118 */
    return 0;
}

/* WFLZ.DecompressHeaderless  WayForward/WFLZ.cs:130-159 (span based: reads past the end are exceptions -> INPUT_TRUNCATED) */
static void dec_wflz(cur_t* c, win_t* w, int big) {
    const uint8_t* s = c->p; uint32_t n = c->len, sp = 0;
    for (;;) {
        if (sp + 4 > n) { c->eof = 1; c->pos = sp; return; }                             /* Slice / indexer throw  :138-140 */
        uint32_t dist = big ? (((uint32_t)s[sp] << 8) | s[sp + 1]) : ((uint32_t)s[sp] | ((uint32_t)s[sp + 1] << 8));
        uint32_t length = s[sp + 2], plain = s[sp + 3];
        sp += 4;
        if (length != 0) {
            uint32_t cl = win_clip(w, length + 4);
            win_back_copy(w, dist, cl);                                                  /* :146 */
            if (w->overflow) { c->pos = sp; return; }
        } else if (plain == 0) { c->pos = sp; return; }                                  /* :148-151 */
        if (plain != 0) {
            if (plain > n - sp) { c->eof = 1; c->pos = sp; return; }                     /* Slice throws  :155 */
            uint32_t cl = win_clip(w, plain);
            win_write(w, s + sp, cl);
            if (w->overflow) { c->pos = sp; return; }
            sp += plain;
        }
    }
}

/* RefPack.DecompressHeaderless  EA/RefPack.cs:177-245.  Returns 1 at the end token (0xFC-0xFF). */
static int dec_refpack(cur_t* c, win_t* w) {
    while (c->pos < c->len) {                                                            /* :183 */
        uint32_t plain, length = 0, distance = 0;
        int prefix = cur_u8(c); if (c->eof) return 0;
        if ((prefix & 0x80) == 0) {                                                      /* 0DDLLLPP DDDDDDDD */
            int d0 = cur_u8(c); if (c->eof) return 0;
            plain = (uint32_t)prefix & 3; length = (((uint32_t)prefix & 0x1C) >> 2) + 3;
            distance = ((((uint32_t)prefix & 0x60) << 3) | (uint32_t)d0) + 1;
        } else if ((prefix & 0x40) == 0) {                                               /* 10LLLLLL PPDDDDDD DDDDDDDD */
            int d0 = cur_u8(c); if (c->eof) return 0;
            int d1 = cur_u8(c); if (c->eof) return 0;
            plain = (uint32_t)d0 >> 6; length = ((uint32_t)prefix & 0x3F) + 4;
            distance = ((((uint32_t)d0 & 0x3F) << 8) | (uint32_t)d1) + 1;
        } else if ((prefix & 0x20) == 0) {                                               /* 110DLLPP DDDDDDDD DDDDDDDD LLLLLLLL */
            int d0 = cur_u8(c); if (c->eof) return 0;
            int d1 = cur_u8(c); if (c->eof) return 0;
            int d2 = cur_u8(c); if (c->eof) return 0;
            plain = (uint32_t)prefix & 3; length = ((((uint32_t)prefix & 0x0C) << 6) | (uint32_t)d2) + 5;
            distance = ((((((uint32_t)prefix & 0x10) << 4) | (uint32_t)d0) << 8) | (uint32_t)d1) + 1;
        } else {                                                                         /* 111PPPPP */
            plain = ((uint32_t)prefix & 0x1F) * 4 + 4;
            if (plain > 0x70) {                                                          /* 111111PP: the end  :222-232 */
                plain = (uint32_t)prefix & 3;
                if (plain > c->len - c->pos) { c->eof = 1; return 0; }
                uint32_t cl = win_clip(w, plain);
                win_write(w, c->p + c->pos, cl);
                if (w->overflow) return 0;
                c->pos += plain;
                return 1;
            }
        }
        if (plain > c->len - c->pos) { c->eof = 1; return 0; }                           /* LzWindows.CopyFrom -> ReadExactly */
        uint32_t cl = win_clip(w, plain);
        win_write(w, c->p + c->pos, cl);
        if (w->overflow) return 0;
        c->pos += plain;
        cl = win_clip(w, length);
        win_back_copy(w, distance, cl);                                                  /* :236 (length 0 behind a literal run) */
        if (w->overflow) return 0;
    }
    c->eof = 1;                                                                          /* :238 */
    return 0;
}

/* LZ02.DecompressHeaderless  Camelot/LZ02.cs:77-115: runs until the terminator token (distance 0, length nibble 0), where only
 * MORE output than declared is an error (:97-100); input that ends first is EndOfStreamException (:114).  Returns 1 at the
 * terminator. */
static int dec_lz02(cur_t* c, win_t* w) {
    flag_t flag = { c, 0, 0, 1, 1 };
    while (c->pos < c->len) {                                                            /* :83 */
        int bit = flag_readbit(&flag); if (c->eof) return 0;
        if (bit) {
            int b1 = cur_u8(c); if (c->eof) return 0;
            int b2 = cur_u8(c); if (c->eof) return 0;
            uint32_t distance = ((uint32_t)(b1 & 0xF0) << 4) | (uint32_t)b2, length = ((uint32_t)b1 & 0xF) + 1;
            if (length == 1) {
                if (distance == 0) return 1;                                             /* :95-101 */
                int b3 = cur_u8(c); if (c->eof) return 0;
                length = (uint32_t)b3 + 17;
            }
            uint32_t cl = win_clip(w, length);
            win_back_copy(w, distance, cl);                                              /* :105 */
            if (w->overflow) return 0;
        } else {
            int b = cur_u8(c); if (c->eof) return 0;
            if (win_clip(w, 1) < 1) return 0;
            win_write_byte(w, (uint8_t)b);
        }
    }
    c->eof = 1;                                                                          /* :114 */
    return 0;
}

/* CNS.DecompressHeaderless  Specialized/CNS.cs:77-108 */
static void dec_cns(cur_t* c, win_t* w, uint32_t size) {
    while (win_produced(w) < size) {                                                     /* :83 */
        int length = cur_u8(c); if (c->eof) return;
        if ((length & 0x80) == 0) {                                                      /* LzWindows.CopyFrom -> ReadExactly */
            if ((uint32_t)length > c->len - c->pos) { c->eof = 1; return; }
            uint32_t cl = win_clip(w, (uint32_t)length);
            win_write(w, c->p + c->pos, cl);
            if (w->overflow) return;
            c->pos += (uint32_t)length;
        } else {
            int d = cur_u8(c); if (c->eof) return;
            uint32_t cl = win_clip(w, ((uint32_t)length & 0x7F) + 3);
            win_back_copy(w, (uint32_t)d + 1, cl);                                       /* :95-98 */
            if (w->overflow) return;
        }
    }
}

/* CLZ0.DecompressHeaderless  Marvelous/CLZ0.cs:64-97: FlagReader(source, Endian.Little), 1 = match.  (The two match
 * bytes are read with Stream.ReadByte(), which returns -1 at the end instead of throwing: the managed code then copies 18
 * bytes from a garbage distance before its next flag read throws EndOfStreamException -- refused here before the copy.) */
static void dec_clz0(cur_t* c, win_t* w, uint32_t size) {
    flag_t flag = { c, 0, 0, 0, 1 };
    while (win_produced(w) < size) {
        int bit = flag_readbit(&flag); if (c->eof) return;
        if (bit) {
            int b1 = cur_u8(c); if (c->eof) return;
            int b2 = cur_u8(c); if (c->eof) return;
            uint32_t delta = (uint32_t)b1 | ((uint32_t)(b2 >> 4) << 8);                  /* :76 */
            uint32_t cl = win_clip(w, ((uint32_t)b2 & 0x0F) + 3);
            win_back_copy(w, 0x1000 - delta, cl);                                        /* :77-80 */
            if (w->overflow) return;
        } else {
            int b = cur_u8(c); if (c->eof) return;
            if (win_clip(w, 1) < 1) return;
            win_write_byte(w, (uint8_t)b);
        }
    }
}

/* CNX2.DecompressHeaderless  Sega/CNX2.cs:83-139: FlagReader(source, Endian.Little), ReadInt(2) = two bits, first one is
 * bit 0 (FlagReader.cs:75-87); code 0 skips `n` bytes and drops the rest of the flag byte (Reset, :102). */
static void dec_cnx2(cur_t* c, win_t* w, uint32_t size) {
    flag_t flag = { c, 0, 0, 0, 1 };
    while (win_produced(w) < size) {                                                     /* :90 */
        int b0 = flag_readbit(&flag); if (c->eof) return;
        int b1 = flag_readbit(&flag); if (c->eof) return;
        switch (b0 | (b1 << 1)) {
        case 0: {                                                                        /* :96-100 */
            int n = cur_u8(c); if (c->eof) return;
            c->pos += (uint32_t)n;                                                       /* source.Position += length (may pass the end) */
            flag.bits_left = 0;
            break;
        }
        case 1: {                                                                        /* :103-105 */
            int b = cur_u8(c); if (c->eof) return;
            if (win_clip(w, 1) < 1) return;
            win_write_byte(w, (uint8_t)b);
            break;
        }
        case 2: {                                                                        /* :108-114 */
            int h = cur_u8(c); if (c->eof) return;
            int l = cur_u8(c); if (c->eof) return;
            uint32_t pair = ((uint32_t)h << 8) | (uint32_t)l;
            uint32_t cl = win_clip(w, (pair & 0x1F) + 4);
            win_back_copy(w, (pair >> 5) + 1, cl);
            if (w->overflow) return;
            break;
        }
        default: {                                                                       /* :117-120  LzWindows.CopyFrom -> ReadExactly */
            int n = cur_u8(c); if (c->eof) return;
            if (c->pos > c->len || (uint32_t)n > c->len - c->pos) { c->eof = 1; return; }
            uint32_t cl = win_clip(w, (uint32_t)n);
            win_write(w, c->p + c->pos, cl);
            if (w->overflow) return;
            c->pos += (uint32_t)n;
            break;
        }
        }
    }
}

/* FastLZ.DecompressHeaderless  Formats/Common/FastLZ.cs:54-160 (span based: reads past the end are
 * IndexOutOfRange / ArgumentOutOfRange exceptions -> INPUT_TRUNCATED; an unknown level is InvalidDataException). */
static void dec_fastlz(cur_t* c, win_t* w, dec_info* info) {
    const uint8_t* s = c->p; uint32_t n = c->len; uint32_t sp = 0;
    if (n == 0) { c->eof = 1; return; }                                                  /* source[0]  :57 */
    int level = (s[0] >> 5) + 1;
    if (level != 1 && level != 2) { info->bad_token = 1; return; }                       /* :60 */
    uint32_t ctrl = s[sp++] & 31u;                                                       /* :67 / :109 */
    for (;;) {
        if (ctrl >= 32) {
            uint64_t len = (ctrl >> 5) - 1; uint32_t ofs = (ctrl & 31u) << 8;
            if (len == 6) {
                if (level == 1) { if (sp >= n) { c->eof = 1; c->pos = sp; return; } len += s[sp++]; }          /* :79-80 */
                else { uint32_t b; do { if (sp >= n) { c->eof = 1; c->pos = sp; return; } b = s[sp++]; len += b; } while (b == 255); }   /* :124-132 */
            }
            if (sp >= n) { c->eof = 1; c->pos = sp; return; }
            ofs |= s[sp++];
            if (level == 2 && ofs == 0x1FFF) {                                           /* :138-143 */
                if (sp + 2 > n) { c->eof = 1; c->pos = sp; return; }
                ofs = ((uint32_t)s[sp] << 8) | s[sp + 1]; sp += 2; ofs += 0x1FFF;
            }
            uint32_t cl = win_clip(w, len + 3);
            win_back_copy(w, ofs + 1, cl);                                               /* :84 / :145 */
            if (w->overflow) { c->pos = sp; return; }
        } else {
            ctrl++;
            if (ctrl > n - sp) { c->eof = 1; c->pos = sp; return; }                      /* Slice throws  :91 */
            uint32_t cl = win_clip(w, ctrl);
            win_write(w, s + sp, cl);
            if (w->overflow) { c->pos = sp; return; }
            sp += ctrl;
        }
        if (sp >= n) break;                                                              /* :96 */
        ctrl = s[sp++];
    }
    c->pos = sp;
}

/* LZO.ReadExtendedInt  Formats/Common/LZO.cs:252-262 */
static uint32_t lzo_ext(cur_t* c) {
    int b; uint32_t length = 0;
    while ((b = cur_byte(c)) == 0) length += 255;
    if (b == -1) { c->eof = 1; return 0; }
    return length + (uint32_t)b;
}

/* LZO.DecompressHeaderless  Formats/Common/LZO.cs:49-139.  Returns 1 when the end marker was read. */
static int dec_lzo(cur_t* c, win_t* w) {
    int flag; uint32_t length, distance, plain = 0;
    flag = cur_byte(c); if (flag < 0) { c->eof = 1; return 0; }                          /* :56 (E6) */
    if (flag > 17) {                                                                     /* :59-64 */
        length = (uint32_t)flag - 17;
        if (length > c->len - c->pos) { c->eof = 1; return 0; }
        uint32_t cl = win_clip(w, length);
        win_copy_from(w, c, cl);
        if (w->overflow) return 0;
        flag = cur_byte(c); if (flag < 0) { c->eof = 1; return 0; }
    }
    do {
        int flagcode = flag >> 4;
        if (flagcode == 0) {
            if (plain == 0) {                                                            /* :72-82 */
                length = 3 + (uint32_t)flag;
                if (length == 3) { length = 18 + lzo_ext(c); if (c->eof) return 0; }
                plain = 4;
                if (length > c->len - c->pos) { c->eof = 1; return 0; }
                uint32_t cl = win_clip(w, length);
                win_copy_from(w, c, cl);
                if (w->overflow) return 0;
                continue;
            } else if (plain <= 3) {                                                     /* :83-88 */
                int d = cur_byte(c); if (d < 0) { c->eof = 1; return 0; }
                distance = ((uint32_t)d << 2) + ((uint32_t)flag >> 2) + 1;
                length = 2;
            } else {                                                                     /* :89-94 */
                int d = cur_byte(c); if (d < 0) { c->eof = 1; return 0; }
                distance = ((uint32_t)d << 2) + ((uint32_t)flag >> 2) + (2048 + 1);
                length = 3;
            }
        } else if (flagcode == 1) {                                                      /* :96-109 */
            length = 2 + ((uint32_t)flag & 0x7);
            if (length == 2) { length = 9 + lzo_ext(c); if (c->eof) return 0; }
            distance = 16384 + (((uint32_t)flag & 0x8) << 11);
            flag = cur_byte(c); if (flag < 0) { c->eof = 1; return 0; }
            int hi = cur_byte(c); if (hi < 0) { c->eof = 1; return 0; }
            distance |= ((uint32_t)hi << 6) | ((uint32_t)flag >> 2);
            if (distance == 16384) return 1;
        } else if (flagcode <= 3) {                                                      /* :110-119 */
            length = 2 + ((uint32_t)flag & 0x1f);
            if (length == 2) { length = 33 + lzo_ext(c); if (c->eof) return 0; }
            flag = cur_byte(c); if (flag < 0) { c->eof = 1; return 0; }
            int hi = cur_byte(c); if (hi < 0) { c->eof = 1; return 0; }
            distance = (((uint32_t)hi << 6) | ((uint32_t)flag >> 2)) + 1;
        } else if (flagcode <= 7) {                                                      /* :120-125 */
            length = 3 + (((uint32_t)flag >> 5) & 0x1);
            int d = cur_byte(c); if (d < 0) { c->eof = 1; return 0; }
            distance = ((uint32_t)d << 3) + (((uint32_t)flag >> 2) & 0x7) + 1;
        } else {                                                                         /* :126-131 */
            length = 5 + (((uint32_t)flag >> 5) & 0x3);
            int d = cur_byte(c); if (d < 0) { c->eof = 1; return 0; }
            distance = ((uint32_t)d << 3) + (((uint32_t)flag & 0x1c) >> 2) + 1;
        }
        plain = (uint32_t)flag & 0x3;                                                    /* :132 */
        uint32_t cl = win_clip(w, length);
        win_back_copy(w, distance, cl);                                                  /* :133 */
        if (w->overflow) return 0;
        if (plain > c->len - c->pos) { c->eof = 1; return 0; }
        cl = win_clip(w, plain);
        win_copy_from(w, c, cl);                                                         /* :134 */
        if (w->overflow) return 0;
    } while ((flag = cur_byte(c)) != -1);                                                /* :136 */
    c->eof = 1;                                                                          /* :137 */
    return 0;
}

/* Snappy.ReadDecompressedSize  Formats/Common/Snappy.cs:109-122 */
static uint32_t snappy_varint(cur_t* c) {
    uint32_t result = 0; int shift = 0; int b = -1;
    while ((b & 0x80) != 0) {
        b = cur_u8(c); if (c->eof) return 0;
        result |= (uint32_t)(b & 0x7F) << (shift & 31);  /* C# masks the shift count */
        shift += 7;
    }
    return result;
}

/* Snappy.DecompressHeaderless  Formats/Common/Snappy.cs:205-250 */
static void dec_snappy(cur_t* c, win_t* w, dec_info* info) {
    uint32_t size = snappy_varint(c); if (c->eof) return;
    while (win_produced(w) < size) {
        int tag = cur_byte(c); if (tag < 0) { c->eof = 1; return; }
        int type = tag & 3; uint32_t length = (uint32_t)tag >> 2; uint32_t distance;
        if (type == 0) {                                                                 /* :221-234 */
            if (length >= 60) {
                int lenBytes = (int)length - 59; length = 0;
                for (int i = 0; i < lenBytes; i++) { int b = cur_byte(c); if (b < 0) { c->eof = 1; return; } length |= (uint32_t)b << (8 * i); }
            }
            uint32_t run = length + 1;                     /* CopyFrom(source, length + 1) */
            if (run > c->len - c->pos) { c->eof = 1; return; }
            uint32_t cl = win_clip(w, run);
            win_copy_from(w, c, cl);
            if (w->overflow) return;
            continue;
        } else if (type == 1) {                                                          /* :235-239 */
            length = (length & 0x7) + 3;
            int b = cur_byte(c); if (b < 0) { c->eof = 1; return; }
            distance = (((uint32_t)tag >> 5) << 8) | (uint32_t)b;
        } else if (type == 2) {                                                          /* :240-243 */
            int b0 = cur_u8(c); if (c->eof) return;
            int b1 = cur_u8(c); if (c->eof) return;
            distance = (uint32_t)b0 | ((uint32_t)b1 << 8);
        } else {                                                                         /* :244-247 */
            if (c->len - c->pos < 4) { c->eof = 1; return; }
            distance = rd32le(c->p + c->pos); c->pos += 4;
            if (distance > w->W) { info->bad_token = 1; return; }                        /* E3 */
        }
        uint32_t cl = win_clip(w, length + 1);
        win_back_copy(w, distance, cl);                                                  /* :248 */
        if (w->overflow) return;
    }
}

static int fmt_window_bits(uint32_t format, const alz_lz_properties* lz) {
    switch (format) {
    case ALZ_FMT_LZSS: return lz->window_bits;
    case ALZ_FMT_LZ10: case ALZ_FMT_LZ11: case ALZ_FMT_LZ40: case ALZ_FMT_YAZ0: case ALZ_FMT_YAY0: case ALZ_FMT_MIO0:
    case ALZ_FMT_LZHUDSON: case ALZ_FMT_SMSR00: case ALZ_FMT_CLZ0: case ALZ_FMT_LZ02: case ALZ_FMT_LZSHREK: return 12; /* LZ10.cs:25 ... CLZ0.cs:24 */
    case ALZ_FMT_PRS_BE: case ALZ_FMT_PRS_LE: return 13;   /* PRS.cs:21 ceil(log2 0x1FFF) */
    case ALZ_FMT_CNX2: return 11;                          /* CNX2.cs:25 ceil(log2 0x800) */
    case ALZ_FMT_CNS: return 8;                            /* CNS.cs:24 ceil(log2 0x100) */
    case ALZ_FMT_REFPACK: return 17;                       /* RefPack.cs:31 ceil(log2 0x20000) */
    case ALZ_FMT_HIG: return 15;                           /* HIG.cs:28 ceil(log2 0x7FFF) */
    case ALZ_FMT_BLZ: return 13;                           /* flat spans in the managed code; distances reach 0xFFF + 3 */
    case ALZ_FMT_LZ4_BLOCK: case ALZ_FMT_LZO: case ALZ_FMT_SNAPPY_RAW: case ALZ_FMT_WFLZ: case ALZ_FMT_WFLZ_BE: return 16; /* LZ4.cs:29, LZO.cs:24, Snappy.cs:213 */
    default: return 12;
    }
}

static void decode_one(const alz_lz_properties* props, const alz_stream* s, const uint8_t* src_base, uint8_t* dst_base,
                       alz_result* r, int flat) {
    alz_lz_properties lz = lzss_effective(props);
    cur_t c = { src_base + s->src_off, s->src_len, 0, 0 };
    win_t w; memset(&w, 0, sizeof(w));
    int wb = fmt_window_bits(s->format, &lz);
    if (s->format == ALZ_FMT_FASTLZ) wb = (s->src_len && (c.p[0] >> 5) == 1) ? 17 : 13;   /* _lz1 / _lz2[1] WindowsBits  FastLZ.cs:22-27 */
    w.flat = flat; w.W = 1u << wb; w.mask = w.W - 1;
    w.dst = dst_base + s->dst_off; w.cap = s->dst_cap;
    /* LZ4 block continuing the window of earlier blocks of its frame (alz_stream.aux0 = history bytes in front of
       dst_off, LZ4.Frame.cs:120): decoded in the flat model with the origin moved back by the history */
    uint32_t hist = s->format == ALZ_FMT_LZ4_BLOCK ? s->aux0 : 0;
    if (hist) { flat = 1; w.flat = 1; w.dst -= hist; w.cap += hist; w.flushed = hist; }
    if (!flat) w.ring = (uint8_t*)calloc(w.W, 1); /* E2: zero-filled */
    dec_info info = { 0, 0, 0 };
    int terminated = 1; uint32_t used = 0; int used_set = 0; int blz_short = 0;
    uint32_t size = s->decom_len;

    switch (s->format) {
    case ALZ_FMT_LZSS: info.has_size = 1; dec_lzss(&lz, &c, &w, size); break;
    case ALZ_FMT_LZ10: info.has_size = 1; dec_lz10(&c, &w, size); break;
    case ALZ_FMT_LZ11: info.has_size = 1; dec_lz11(&c, &w, size); break;
    case ALZ_FMT_LZ40: info.has_size = 1; dec_lz40(&c, &w, size); break;
    case ALZ_FMT_LZHUDSON: info.has_size = 1; dec_yay0_w(&c, &c, &c, &w, size, 4); break;   /* FlagReader(source, Endian.Big, 4, Endian.Big)  LZHudson.cs:53 */
    case ALZ_FMT_SMSR00: {
        info.has_size = 1;
        if (s->aux0 > s->src_len) { c.eof = 1; break; }                                     /* ReadExactly(buffer, 0, codesLength) throws */
        cur_t codes = { c.p, s->aux0, 0, 0 };
        cur_t lit = { c.p + s->aux0, s->src_len - s->aux0, 0, 0 };
        dec_smsr00(&codes, &lit, &w, size);
        if (codes.eof || lit.eof) c.eof = 1;
        used = s->aux0 + lit.pos; used_set = 1;                                             /* source.Position: behind the codes + literals read */
        break;
    }
    case ALZ_FMT_YAZ0: info.has_size = 1; dec_yay0(&c, &c, &c, &w, size); break;
    case ALZ_FMT_YAY0: {
        info.has_size = 1;
        if (s->aux0 > s->src_len || s->aux1 > s->src_len) { c.eof = 1; break; }        /* Slice throws  Yay0.cs:102-103 */
        cur_t cc = { c.p + s->aux0, s->src_len - s->aux0, 0, 0 };
        cur_t uc = { c.p + s->aux1, s->src_len - s->aux1, 0, 0 };
        dec_yay0(&c, &cc, &uc, &w, size);
        if (cc.eof || uc.eof) c.eof = 1;
        uint32_t a = s->aux0 + cc.pos, b = s->aux1 + uc.pos;                             /* Yay0.cs:107 */
        used = a > b ? a : b; used_set = 1;
        break;
    }
    case ALZ_FMT_MIO0: info.has_size = 1; dec_mio0(&c, &w, size, s->aux0, s->aux1, &used); used_set = 1; break;
    case ALZ_FMT_PRS_BE: terminated = dec_prs(&c, &w, 1); break;
    case ALZ_FMT_PRS_LE: terminated = dec_prs(&c, &w, 0); break;
    case ALZ_FMT_LZ4_BLOCK: dec_lz4(&c, &w); break;
    case ALZ_FMT_LZO: terminated = dec_lzo(&c, &w); break;
    case ALZ_FMT_SNAPPY_RAW: dec_snappy(&c, &w, &info); break;
    case ALZ_FMT_FASTLZ: dec_fastlz(&c, &w, &info); break;
    case ALZ_FMT_CNX2: info.has_size = 1; dec_cnx2(&c, &w, size); break;
    case ALZ_FMT_BLZ: info.has_size = 1; dec_blz(&c, &w, size, &info, &blz_short); break;
    case ALZ_FMT_CLZ0: info.has_size = 1; dec_clz0(&c, &w, size); break;
    case ALZ_FMT_CNS: info.has_size = 1; dec_cns(&c, &w, size); break;
    case ALZ_FMT_LZ02: info.has_size = 1; terminated = dec_lz02(&c, &w); break;
    case ALZ_FMT_REFPACK: info.has_size = 1; terminated = dec_refpack(&c, &w); break;
    case ALZ_FMT_WFLZ: dec_wflz(&c, &w, 0); break;
    case ALZ_FMT_LZSHREK: info.has_size = 1; terminated = dec_lzshrek(&c, &w, &info); break;
    case ALZ_FMT_HIG: info.has_size = 1; dec_hig(&c, &w, size); break;
    case ALZ_FMT_WFLZ_BE: dec_wflz(&c, &w, 1); break;
    default: info.bad_token = 1; break;
    }
    (void)terminated;
    win_dispose(&w);
    if (!flat) free(w.ring);

    uint64_t produced = win_produced(&w) - hist;
    r->dst_len = (uint32_t)produced;
    r->src_used = used_set ? used : c.pos;
    r->reserved = 0;
    if (c.eof) { r->status = ALZ_ST_INPUT_TRUNCATED; r->src_used = s->src_len; }   /* the reader ran into the end of the input: Position is there (auroralz.h) */
    else if (info.bad_token) r->status = ALZ_ST_BAD_TOKEN;
    else if (w.overflow)
        r->status = (info.has_size && w.attempted_end > size && w.cap >= size) ? ALZ_ST_OUTPUT_SIZE_MISMATCH : ALZ_ST_OUTPUT_CAPACITY;
    else if (info.has_size && (produced > size || blz_short)) r->status = ALZ_ST_OUTPUT_SIZE_MISMATCH; /* LZ10.cs:107 '>' ; LZSS.cs:126 '!=' (same: loop ran to >= size); BLZ.cs:131 */
    else r->status = ALZ_ST_OK;
}

void oracle_decode_stream(const alz_lz_properties* props, const alz_stream* s, const uint8_t* src_base, uint8_t* dst_base, alz_result* r) {
    decode_one(props, s, src_base, dst_base, r, 0);
}
void oracle_decode_stream_flat(const alz_lz_properties* props, const alz_stream* s, const uint8_t* src_base, uint8_t* dst_base, alz_result* r) {
    decode_one(props, s, src_base, dst_base, r, 1);
}

/* ---- batch over host threads (streams striped) ---- */
typedef struct {
    const alz_lz_properties* props; const alz_settings* settings; uint32_t n; const uint8_t* src_base; const alz_stream* streams;
    uint8_t* dst_base; alz_result* results; alz_encode_aux* aux; int tid, nthreads; int encode;
} job_t;

static void* batch_worker(void* arg) {
    job_t* j = (job_t*)arg;
    for (uint32_t i = (uint32_t)j->tid; i < j->n; i += (uint32_t)j->nthreads) {
        const alz_stream* s = &j->streams[i];
        if (!j->encode) {
            decode_one(j->props, s, j->src_base, j->dst_base, &j->results[i], 0);
        } else {
            alz_encode_aux a = { 0, 0 };
            int64_t sz = oracle_encode_stream(s->format, j->props, j->settings, j->src_base + s->src_off, s->src_len,
                                              j->dst_base + s->dst_off, s->dst_cap, &a);
            j->results[i].dst_len = sz < 0 ? 0 : (uint32_t)sz;
            j->results[i].src_used = s->src_len;
            j->results[i].status = sz == -1 ? ALZ_ST_OUTPUT_CAPACITY : sz < 0 ? ALZ_ST_BAD_TOKEN : ALZ_ST_OK;
            j->results[i].reserved = 0;
            if (j->aux) j->aux[i] = a;
        }
    }
    return NULL;
}

static int run_batch(job_t* proto) {
    int nt = proto->nthreads < 1 ? 1 : proto->nthreads;
    if (nt > 256) nt = 256;
    if (nt == 1) { proto->tid = 0; proto->nthreads = 1; batch_worker(proto); return 0; }
    pthread_t th[256]; job_t jobs[256];
    for (int t = 0; t < nt; t++) { jobs[t] = *proto; jobs[t].tid = t; jobs[t].nthreads = nt; pthread_create(&th[t], NULL, batch_worker, &jobs[t]); }
    for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
    return 0;
}

int oracle_decode_batch(const alz_lz_properties* props, uint32_t n, const uint8_t* src_base, const alz_stream* streams,
                        uint8_t* dst_base, alz_result* results, int nthreads) {
    job_t j = { props, NULL, n, src_base, streams, dst_base, results, NULL, 0, nthreads, 0 };
    return run_batch(&j);
}

int oracle_encode_batch(const alz_lz_properties* props, const alz_settings* settings, uint32_t n, const uint8_t* src_base,
                        const alz_stream* streams, uint8_t* dst_base, alz_result* results, alz_encode_aux* aux, int nthreads) {
    job_t j = { props, settings, n, src_base, streams, dst_base, results, aux, 0, nthreads, 1 };
    return run_batch(&j);
}

/* ============================================================ encoder side */

typedef struct { uint8_t* p; size_t len, cap; int fail; int owned; } buf_t;

static void buf_put(buf_t* b, const void* data, size_t n) {
    if (b->fail) return;
    if (b->len + n > b->cap) {
        if (!b->owned) { b->fail = 1; return; }
        size_t nc = b->cap ? b->cap * 2 : 256; while (nc < b->len + n) nc *= 2;
        b->p = (uint8_t*)realloc(b->p, nc); b->cap = nc;
    }
    memcpy(b->p + b->len, data, n); b->len += n;
}
static inline void buf_u8(buf_t* b, uint32_t v) { uint8_t x = (uint8_t)v; buf_put(b, &x, 1); }
static inline void buf_u16be(buf_t* b, uint32_t v) { uint8_t x[2] = { (uint8_t)(v >> 8), (uint8_t)v }; buf_put(b, x, 2); }
static inline void buf_u16le(buf_t* b, uint32_t v) { uint8_t x[2] = { (uint8_t)v, (uint8_t)(v >> 8) }; buf_put(b, x, 2); }
static inline void buf_u32be(buf_t* b, uint32_t v) { uint8_t x[4] = { (uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v }; buf_put(b, x, 4); }
static inline void buf_u32le(buf_t* b, uint32_t v) { uint8_t x[4] = { (uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24) }; buf_put(b, x, 4); }
static buf_t buf_owned(void) { buf_t b = { NULL, 0, 0, 0, 1 }; return b; }
static void buf_free(buf_t* b) { if (b->owned) free(b->p); b->p = NULL; }

/* FlagWriter  IO/FlagWriter.cs:13-147 (8-bit flags) */
typedef struct { buf_t* base; buf_t buffer; int bits_left; uint32_t cur; int msb_first; int negate; int nbytes; } fw_t;

static void fw_init(fw_t* f, buf_t* base, int msb_first) { f->base = base; f->buffer = buf_owned(); f->bits_left = 8; f->cur = 0; f->msb_first = msb_first; f->negate = 0; f->nbytes = 1; }
static void fw_set_width(fw_t* f, int nbytes) { f->nbytes = nbytes; f->bits_left = 8 * nbytes; }   /* FlagWriter(dest, Endian.Big, flagSize, Endian.Big) */
/* Flush  FlagWriter.cs:111-127 */
static void fw_flush(fw_t* f) {
    if (f->bits_left != 8 * f->nbytes) {
        if (f->nbytes == 1) buf_u8(f->base, f->negate ? (0u - f->cur) & 0xFFu : f->cur);   /* LZ40: i => WriteByte((byte)-i) */
        else for (int i = f->nbytes - 1; i >= 0; i--) buf_u8(f->base, (f->cur >> (8 * i)) & 0xFFu);
        f->bits_left = 8 * f->nbytes; f->cur = 0;
    }
    if (f->buffer.len != 0) { buf_put(f->base, f->buffer.p, f->buffer.len); f->buffer.len = 0; }
}
/* WriteBit  FlagWriter.cs:70-80 */
static void fw_bit(fw_t* f, int bit) {
    if (bit) { int shift = f->msb_first ? f->bits_left - 1 : 8 * f->nbytes - f->bits_left; f->cur |= 1u << shift; }
    f->bits_left--;
    if (f->bits_left == 0) fw_flush(f);
}
/* FlushIfNecessary  FlagWriter.cs:132-139 */
static void fw_flush_if_necessary(fw_t* f) {
    if (f->bits_left == 8 * f->nbytes && f->buffer.len != 0) { buf_put(f->base, f->buffer.p, f->buffer.len); f->buffer.len = 0; }
}
static void fw_dispose(fw_t* f) { fw_flush(f); buf_free(&f->buffer); }

/* LzChainMatchFinder  MatchFinder/LzChainMatchFinder.cs:13-372 */
typedef struct {
    int minLen, maxLen, minDist, maxDist;
    int chainMask, hashBits, hashMask, maxChain; uint32_t minMask;
    int lazy, noSelfOverlap;
    int* head; int* chain; int* minTable;
    int position;
    int nprops;                                  /* > 1: the LzProperties[] form of the constructor (:42-69), ScoreMatch walks them (:301-321) */
    struct { int maxLen, minLen, maxDist, minDist; } props[3];
} mf_t;

typedef struct { int offset, distance, length; } lzmatch_t;
typedef struct { int windowBits, maxLen, minLen, maxDist, minDist; } fmt_props;

static int isqrt_floor(int v) { int r = 0; while ((r + 1) * (r + 1) <= v) r++; return r; }

/* GetMaxChain  LzChainMatchFinder.cs:111-119 */
static int mf_max_chain(int q) {
    if (q < 6) return q + 1;
    if (q >= 11) return 1 << (q - 5);
    int b = 1 << (q >> 1);
    return b | (b >> (q & 1));
}

/* Reset  LzChainMatchFinder.cs:125-132 */
static void mf_reset(mf_t* m) {
    m->position = 0;
    for (int i = 0; i <= m->hashMask; i++) m->head[i] = -1;
    if (m->maxChain != 1) for (int i = 0; i <= m->chainMask; i++) m->chain[i] = -1;
    if (m->minTable) for (int i = 0; i < 65536; i++) m->minTable[i] = -1;
}

/* ctor  LzChainMatchFinder.cs:42-109 */
static void mf_init_multi(mf_t* m, const fmt_props* p, int np, const alz_settings* st) {
    int q = st ? st->quality : 8;
    memset(m, 0, sizeof(*m));
    /* global constraints: the loosest of each over all property sets (:55-69) */
    m->minLen = p[0].minLen; m->maxLen = p[0].maxLen; m->minDist = p[0].minDist; m->maxDist = p[0].maxDist;
    int windowsBits = p[0].windowBits;
    m->nprops = np;
    for (int i = 0; i < np; i++) {
        if (m->minLen > p[i].minLen) m->minLen = p[i].minLen;
        if (m->maxLen < p[i].maxLen) m->maxLen = p[i].maxLen;
        if (m->minDist > p[i].minDist) m->minDist = p[i].minDist;
        if (m->maxDist < p[i].maxDist) m->maxDist = p[i].maxDist;
        if (windowsBits < p[i].windowBits) windowsBits = p[i].windowBits;
        m->props[i].maxLen = p[i].maxLen; m->props[i].minLen = p[i].minLen; m->props[i].maxDist = p[i].maxDist; m->props[i].minDist = p[i].minDist;
    }
    int maxWindowBits = st ? st->max_window_bits : 0;
    if (maxWindowBits != 0) {
        if (windowsBits < maxWindowBits) windowsBits = maxWindowBits;
        if (m->maxDist < (1 << maxWindowBits)) m->maxDist = 1 << maxWindowBits;
    }
    m->lazy = 3 + q / 3;
    m->noSelfOverlap = st ? (st->strategy & 1) : 0;
    m->hashBits = 15 + isqrt_floor(2 * q);
    m->hashMask = (1 << m->hashBits) - 1;
    m->head = (int*)malloc(sizeof(int) << m->hashBits);
    m->maxChain = mf_max_chain(q);
    if (m->maxChain == 1) { m->chain = NULL; m->chainMask = 0; }
    else {
        int cb = 17 + isqrt_floor(2 * q); if (cb > windowsBits) cb = windowsBits;
        m->chain = (int*)malloc(sizeof(int) << cb); m->chainMask = (1 << cb) - 1;
    }
    if (q >= 10 && m->minLen < 4) { m->minMask = 0xFFFFFFFFu >> ((4 - m->minLen) * 8); m->minTable = (int*)malloc(sizeof(int) * 65536); }
    mf_reset(m);
}
static void mf_init(mf_t* m, const fmt_props* p, const alz_settings* st) { mf_init_multi(m, p, 1, st); }
static void mf_free(mf_t* m) { free(m->head); free(m->chain); free(m->minTable); }

/* ComputeHash  LzChainMatchFinder.cs:288-299 */
static inline void mf_hash(const mf_t* m, const uint8_t* d, int* h4, int* hm) {
    uint32_t v = rd32le(d); uint32_t mn = v & m->minMask;
    v *= 2654435761u; mn *= 2654435761u;
    *h4 = (int)(v >> (32 - m->hashBits)) & m->hashMask;
    *hm = (int)((mn >> 16) & 0xFFFF);
}
static inline int mf_next(const mf_t* m, int pos) { return m->chain ? m->chain[pos & m->chainMask] : -1; } /* GetNext :285, NoChainTable :41 */
/* Insert  :134-144 */
static inline void mf_insert(mf_t* m, int pos, int h4, int hm) {
    if (m->chainMask != 0) m->chain[pos & m->chainMask] = m->head[h4];
    m->head[h4] = pos;
    if (m->minTable) m->minTable[hm] = pos;
}
/* GetMatchLength  :338-357 */
static inline int mf_match_len(const uint8_t* a, const uint8_t* b, int max) {
    int len = 0;
    while (len + 8 <= max) {
        uint64_t diff = rd64le(a + len) ^ rd64le(b + len);
        if (diff != 0) return len + (__builtin_ctzll(diff) >> 3);
        len += 8;
    }
    while (len < max && a[len] == b[len]) len++;
    return len;
}
/* ScoreMatch  :301-321 */
static inline int mf_score(const mf_t* m, int* len, int dist) {
    if (m->noSelfOverlap && *len > dist) *len = dist;
    if (m->nprops <= 1) return *len - m->minLen;
    for (int i = 0; i < m->nprops; i++) {                    /* the first property set that admits the match */
        if (dist <= m->props[i].maxDist && *len >= m->props[i].minLen && dist >= m->props[i].minDist) {
            if (*len > m->props[i].maxLen) *len = m->props[i].maxLen;
            return *len - m->props[i].minLen;
        }
    }
    *len = 0;
    return -1;
}

/* MatchSearch  :214-246 (ChainMatches :248-282 inlined) */
static void mf_search(mf_t* m, const uint8_t* data, int dataLength, int pos, int attempts, int* bestDistance, int* bestLength) {
    const uint8_t* dp = data + pos; int h4, hm;
    mf_hash(m, dp, &h4, &hm);
    int cur = m->head[h4];
    int bestPossible = dataLength - pos; if (bestPossible > m->maxLen) bestPossible = m->maxLen;
    *bestDistance = 0; *bestLength = 0; int bestScore = -1;
    while (cur != -1 && attempts-- > 0) {
        int distance = pos - cur;
        if (distance > m->maxDist) break;
        if (distance < m->minDist) { cur = mf_next(m, cur); continue; }
        int len = mf_match_len(dp, data + cur, bestPossible);
        int score = mf_score(m, &len, distance);
        if (score > bestScore) {
            bestScore = score; *bestLength = len; *bestDistance = distance;
            if (*bestLength == bestPossible) break;
        }
        cur = mf_next(m, cur);
    }
    if (*bestLength == 0 && m->minTable) {                                               /* :226-243 */
        cur = m->minTable[hm];
        if (cur != -1) {
            int distance = pos - cur;
            if (distance < m->minDist) distance = m->minDist;
            /* oracle guard: the reference would read before the span when pos < distance (undefined) */
            if (distance <= m->maxDist && pos - distance >= 0) {
                *bestLength = mf_match_len(dp, data + pos - distance, bestPossible);
                (void)mf_score(m, bestLength, distance);
                *bestDistance = distance;
            }
        }
    }
    mf_insert(m, pos, h4, hm);
}

/* FindNextBestMatch  :157-212 */
static lzmatch_t mf_find(mf_t* m, const uint8_t* data, int length) {
    int limit = length - 4;
    while (m->position <= limit) {
        int bestDistance, bestLength;
        mf_search(m, data, length, m->position, m->maxChain, &bestDistance, &bestLength);
        if (bestLength < m->minLen) { m->position++; continue; }
        int skip = 0;
        if (bestLength <= m->lazy && m->position + 1 <= limit) {
            int nextPos = m->position + 1, nd, nl;
            mf_search(m, data, length, nextPos, m->maxChain, &nd, &nl);
            if (nl > bestLength) { bestLength = nl; bestDistance = nd; m->position = nextPos; }
            else skip++;
        }
        lzmatch_t match = { m->position, bestDistance, bestLength };
        int end = m->position + bestLength;
        m->position++;
        m->position += skip;
        while (m->position < end && m->position <= limit) {
            int h4, hm; mf_hash(m, data + m->position, &h4, &hm);
            mf_insert(m, m->position, h4, hm); m->position++;
        }
        return match;
    }
    m->position = length;
    lzmatch_t end = { length, 0, 0 };
    return end;
}

/* per-format LzProperties */
static fmt_props props_for(uint32_t format, const alz_lz_properties* lzp, const alz_settings* st) {
    fmt_props p = { 12, 18, 3, 0x1000, 1 };
    switch (format) {
    case ALZ_FMT_LZSS: {
        alz_lz_properties lz = lzss_effective(lzp);
        p.windowBits = lz.window_bits; p.minLen = lz.min_length;
        p.maxLen = (1 << lz.length_bits) + lz.min_length - 1; p.maxDist = (int)lz.max_distance; p.minDist = 1; break;
    }
    case ALZ_FMT_LZ10: p = (fmt_props){ 12, 18, 3, 0x1000, 1 }; break;                   /* LZ10.cs:25 */
    case ALZ_FMT_LZ11: p = (fmt_props){ 12, 0x4000, 3, 0x1000, 1 }; break;               /* LZ11.cs:25 */
    case ALZ_FMT_LZ40: p = (fmt_props){ 12, 0x4000, 3, 0x1000, 1 }; break;               /* LZ40.cs:25 */
    case ALZ_FMT_LZHUDSON: p = (fmt_props){ 12, 0xff + 18, 3, 0x1000, 1 }; break;         /* LZHudson.cs:20 */
    case ALZ_FMT_SMSR00: p = (fmt_props){ 12, 18, 3, 0x1000, 1 }; break;                 /* SMSR00.cs:25 */
    case ALZ_FMT_YAZ0: case ALZ_FMT_YAY0: p = (fmt_props){ 12, 0xff + 0x12, 3, 0x1000, 1 }; break; /* Yay0.cs:27 */
    case ALZ_FMT_MIO0: p = (fmt_props){ 12, 18, 3, 0x1000, 1 }; break;                   /* MIO0.cs:28 */
    case ALZ_FMT_PRS_BE: case ALZ_FMT_PRS_LE: p = (fmt_props){ 13, 0x100, 2, 0x1FFF, 1 }; break; /* PRS.cs:21 */
    case ALZ_FMT_LZ4_BLOCK: p = (fmt_props){ 16, 0x7FFFFFFF, 4, 0xFFFF, 1 }; break;      /* LZ4.cs:29 */
    case ALZ_FMT_LZO: p = (fmt_props){ 16, 0x7FFFFFFF, 3, 0xBFFF, 1 }; break;            /* LZO.cs:24 */
    case ALZ_FMT_SNAPPY_RAW: p = (fmt_props){ 15, 64, 4, 0x8000, 1 }; break;             /* Snappy.cs:28 */
    case ALZ_FMT_FASTLZ: p = (fmt_props){ 13, 255 + 3 + 6, 3, 0x2000, 1 }; break;        /* level 1  FastLZ.cs:22 */
    case ALZ_FMT_CNX2: p = (fmt_props){ 11, 0x1F + 4, 4, 0x800, 1 }; break;               /* CNX2.cs:25 */
    case ALZ_FMT_BLZ: p = (fmt_props){ 12, 18, 3, 0x1000, 3 }; break;                     /* BLZ.cs:24 (minDistance 3) */
    case ALZ_FMT_CLZ0: p = (fmt_props){ 12, 18, 3, 0x1000, 1 }; break;                    /* CLZ0.cs:24 */
    case ALZ_FMT_CNS: p = (fmt_props){ 8, 130, 3, 0x100, 1 }; break;                      /* CNS.cs:24 */
    case ALZ_FMT_LZ02: p = (fmt_props){ 12, 272, 3, 0xFFF, 1 }; break;                    /* LZ02.cs:23 */
    case ALZ_FMT_WFLZ: case ALZ_FMT_WFLZ_BE: p = (fmt_props){ 16, 255, 5, 0xFFFF, 1 }; break;   /* WFLZ.cs:20 */
    case ALZ_FMT_LZSHREK: p = (fmt_props){ 12, 262, 3, 0x1000, 1 }; break;                /* LZShrek.cs:20 */
    case ALZ_FMT_HIG: p = (fmt_props){ 15, 0xFFFF, 4, 0x7FFF, 1 }; break;                 /* HIG.cs:28 */
    default: break;
    }
    if (st && st->min_distance > 0) p.minDist = st->min_distance;                        /* _lzVram LZ10.cs:30 */
    return p;
}

/* LZSS.CompressHeaderless  LZSS.cs:132-160 */
static void enc_lzss(const alz_lz_properties* lzp, const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    alz_lz_properties lz = lzss_effective(lzp);
    fmt_props p = props_for(ALZ_FMT_LZSS, lzp, st);
    mf_t m; mf_init(&m, &p, st); fw_t flag; fw_init(&flag, out, 0);
    int sp = 0; uint32_t nmask = lz.max_distance - 1, f = (1u << lz.length_bits) - 1;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(&flag.buffer, src[sp++]); fw_bit(&flag, 1); }
        if (match.length == 0) break;
        uint32_t offset = (lz.windows_start + (uint32_t)sp - (uint32_t)match.distance) & nmask;
        uint32_t v = (offset & 0xFF) | ((offset & 0xFF00) << lz.length_bits) | ((((uint32_t)match.length - lz.min_length) & f) << 8);
        buf_u16le(&flag.buffer, v & 0xFFFF);
        fw_bit(&flag, 0);
        sp += match.length;
    }
    fw_dispose(&flag); mf_free(&m);
}

/* LZ10.CompressHeaderless  LZ10.cs:113-137 */
static void enc_lz10(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_LZ10, NULL, st);
    mf_t m; mf_init(&m, &p, st); fw_t flag; fw_init(&flag, out, 1);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(&flag.buffer, src[sp++]); fw_bit(&flag, 0); }
        if (match.length == 0) break;
        buf_u16be(&flag.buffer, (uint32_t)(((match.length - 3) << 12) | ((match.distance - 1) & 0xFFF)) & 0xFFFF);
        sp += match.length;
        fw_bit(&flag, 1);
    }
    fw_dispose(&flag); mf_free(&m);
}

/* LZ11.CompressHeaderless  LZ11.cs:135-171 */
static void enc_lz11(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_LZ11, NULL, st);
    mf_t m; mf_init(&m, &p, st); fw_t flag; fw_init(&flag, out, 1);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(&flag.buffer, src[sp++]); fw_bit(&flag, 0); }
        if (match.length == 0) break;
        if (match.length <= 16) {
            buf_u16be(&flag.buffer, (uint32_t)(((match.length - 1) << 12) | ((match.distance - 1) & 0xFFF)) & 0xFFFF);
        } else if (match.length <= 272) {
            buf_u8(&flag.buffer, (uint32_t)(((match.length - 17) & 0xFF) >> 4));
            buf_u16be(&flag.buffer, (uint32_t)(((match.length - 17) << 12) | ((match.distance - 1) & 0xFFF)) & 0xFFFF);
        } else {
            buf_u32be(&flag.buffer, 0x10000000u | ((uint32_t)((match.length - 273) & 0xFFFF) << 12) | (uint32_t)((match.distance - 1) & 0xFFF));
        }
        sp += match.length;
        fw_bit(&flag, 1);
    }
    fw_dispose(&flag); mf_free(&m);
}

/* LZ40.CompressHeaderless  LZ40.cs:134-176 */
static void enc_lz40(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_LZ40, NULL, st);
    mf_t m; mf_init(&m, &p, st); fw_t flag; fw_init(&flag, out, 1); flag.negate = 1;
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(&flag.buffer, src[sp++]); fw_bit(&flag, 0); }
        if (match.length == 0) break;
        if (match.length < 16) buf_u16le(&flag.buffer, (uint32_t)((match.distance << 4) | match.length) & 0xFFFF);
        else if (match.length < 272) { buf_u16le(&flag.buffer, (uint32_t)(match.distance << 4) & 0xFFFF); buf_u8(&flag.buffer, (uint32_t)(match.length - 16)); }
        else { buf_u16le(&flag.buffer, (uint32_t)((match.distance << 4) | 1) & 0xFFFF); buf_u16le(&flag.buffer, (uint32_t)(match.length - 272) & 0xFFFF); }
        sp += match.length;
        fw_bit(&flag, 1);
    }
    fw_dispose(&flag); mf_free(&m);
}

/* Yay0.CompressHeaderless(source, compressedData, uncompressedData, FlagWriter, settings)  Yay0.cs:152-184.
 * Yaz0 passes flag.Buffer for both data streams (Yaz0.cs:94-98). */
static void enc_yay0_core(const alz_settings* st, const uint8_t* src, int n, buf_t* comp, buf_t* uncomp, fw_t* flag, uint32_t format) {
    fmt_props p = props_for(format, NULL, st);
    mf_t m; mf_init(&m, &p, st);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(uncomp, src[sp++]); fw_bit(flag, 1); }
        if (match.length == 0) break;
        if (match.length < 18) {
            buf_u16be(comp, (uint32_t)((match.distance - 1) | ((match.length - 2) << 12)) & 0xFFFF);
        } else {
            buf_u16be(comp, (uint32_t)((match.distance - 1) & 0xFFF));
            buf_u8(uncomp, (uint32_t)(match.length - 0x12));
        }
        sp += match.length;
        fw_bit(flag, 0);
    }
    mf_free(&m);
}

/* MIO0.CompressHeaderless  MIO0.cs:159-184 */
static void enc_mio0_core(const alz_settings* st, const uint8_t* src, int n, buf_t* comp, buf_t* uncomp, fw_t* flag) {
    fmt_props p = props_for(ALZ_FMT_MIO0, NULL, st);
    mf_t m; mf_init(&m, &p, st);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(uncomp, src[sp++]); fw_bit(flag, 1); }
        if (match.length == 0) break;
        buf_u16be(comp, (uint32_t)((match.distance - 1) | ((match.length - 3) << 12)) & 0xFFFF);
        sp += match.length;
        fw_bit(flag, 0);
    }
    mf_free(&m);
}

/* PRS.CompressHeaderless  PRS.cs:104-159 */
static void enc_prs(const alz_settings* st, const uint8_t* src, int n, buf_t* out, int big) {
    fmt_props p = props_for(ALZ_FMT_PRS_BE, NULL, st);
    mf_t m; mf_init(&m, &p, st); fw_t flag; fw_init(&flag, out, big);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(&flag.buffer, src[sp++]); fw_bit(&flag, 1); }
        if (match.length == 0) break;
        if (match.length == 2 && match.distance > 0x100) continue;                       /* :124-125 */
        sp += match.length;
        int distance = -match.distance; int length = match.length;
        fw_bit(&flag, 0);
        if (distance >= -0x100 && length <= 5) {
            fw_bit(&flag, 0);
            fw_bit(&flag, ((length - 2) >> 1) & 1); fw_bit(&flag, (length - 2) & 1);       /* WriteInt(length-2, 2, true) */
            buf_u8(&flag.buffer, (uint32_t)distance & 0xFF);
            fw_flush_if_necessary(&flag);
        } else {
            if (length > 9) {
                uint32_t v = ((uint32_t)distance << 3) & 0xFFFF;          /* (distance is negative: C#'s (ushort)(distance << 3) on the two's-complement bits) */
                if (big) buf_u16be(&flag.buffer, v); else buf_u16le(&flag.buffer, v);
                buf_u8(&flag.buffer, (uint32_t)(length - 1));
            } else {
                uint32_t v = (((uint32_t)distance << 3) | (uint32_t)(length - 2)) & 0xFFFF;
                if (big) buf_u16be(&flag.buffer, v); else buf_u16le(&flag.buffer, v);
            }
            fw_bit(&flag, 1);
        }
    }
    fw_bit(&flag, 0);
    buf_u8(&flag.buffer, 0); buf_u8(&flag.buffer, 0);
    fw_bit(&flag, 1);
    fw_dispose(&flag); mf_free(&m);
}

/* LZ4.WriteExtension  LZ4.cs:254-268 */
static void lz4_write_ext(buf_t* out, int length) {
    length -= 0xF;
    if (length >= 0) {
        int b;
        do { b = length < 0xFF ? length : 0xFF; buf_u8(out, (uint32_t)b); length -= b; } while (b == 0xFF);
    }
}

/* LZ4.CompressBlockHeaderless  LZ4.cs:202-238; returns -2 where the reference throws (n < 5) */
static int enc_lz4(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    if (n < 5) return -2;                                /* source.Slice(0, Length - 5) throws */
    fmt_props p = props_for(ALZ_FMT_LZ4_BLOCK, NULL, st);
    mf_t m; mf_init(&m, &p, st);
    int sp = 0, plain, token; int en = n - 5;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, en);
        plain = match.offset - sp;
        token = (plain > 0xF ? 0xF : plain) << 4;
        if (match.length != 0) token |= (match.length - 4 > 0xF ? 0xF : match.length - 4);
        else { plain = n - sp; token = (plain > 0xF ? 0xF : plain) << 4; }
        buf_u8(out, (uint32_t)token);
        lz4_write_ext(out, plain);
        buf_put(out, src + sp, (size_t)plain);
        sp += plain;
        if (sp >= n) break;
        buf_u16le(out, (uint32_t)match.distance & 0xFFFF);
        lz4_write_ext(out, match.length - 4);
        sp += match.length;
    }
    mf_free(&m);
    return 0;
}

/* LZO.WriteExtendedInt  LZO.cs:263-271 */
static void lzo_write_ext(buf_t* out, int v) { while (v > 255) { buf_u8(out, 0); v -= 255; } buf_u8(out, (uint32_t)v); }

/* LZO.CompressHeaderless  LZO.cs:141-250; returns -2 where the reference would throw */
static int enc_lzo(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    if (n < 0x10) {                                                                      /* :143-152 */
        buf_u8(out, (uint32_t)(17 + n)); buf_put(out, src, (size_t)n);
        buf_u8(out, 0x11); buf_u8(out, 0); buf_u8(out, 0);
        return 0;
    }
    fmt_props p = props_for(ALZ_FMT_LZO, NULL, st);
    mf_t m; mf_init(&m, &p, st);
    int sp = 0, rc = 0;
    lzmatch_t match = mf_find(&m, src, n), next = mf_find(&m, src, n);
    while (sp != n) {
        int plain = match.offset - sp;
        if (plain != 0) {
            if (plain < 4) {                                                             /* :167-172 */
                int dif = 4 - plain;
                match.offset += dif; match.length -= dif;
                plain = 4;
            }
            if (plain > 18) { buf_u8(out, 0); lzo_write_ext(out, plain - 18); }
            else buf_u8(out, (uint32_t)(plain - 3));
            if (sp + plain > n) { rc = -2; break; }      /* Slice would throw */
            buf_put(out, src + sp, (size_t)plain);
            sp += plain;
        }
        if (match.length >= 3) {                                                         /* _lz.MinLength :187 */
            sp += match.length;
            plain = next.offset - sp;
            if (plain > 3) plain = 0;
            if (plain < 0) { rc = -2; break; }
            if (match.length <= 8 && match.distance <= 2048) {
                uint32_t flag = (uint32_t)(plain | (((match.distance - 1) & 0x7) << 2)) & 0xFF;
                if (match.length <= 4) buf_u8(out, flag | 0x40 | (uint32_t)((match.length - 3) << 5));
                else buf_u8(out, flag | 0x80 | (uint32_t)((match.length - 5) << 5));
                buf_u8(out, (uint32_t)((match.distance - 1) >> 3));
            } else if (match.distance <= 16384) {
                if (match.length > 33) { buf_u8(out, 0x20); lzo_write_ext(out, match.length - 33); }
                else buf_u8(out, 0x20 | (uint32_t)(match.length - 2));
                buf_u8(out, (uint32_t)(plain | ((match.distance - 1) << 2)));
                buf_u8(out, (uint32_t)((match.distance - 1) >> 6));
            } else {
                const int hFlag = 0x4000;
                int distance = match.distance - hFlag;
                uint32_t flag = (uint32_t)(0x10 | ((distance & hFlag) >> 11)) & 0xFF;
                if (match.length > 9) { buf_u8(out, flag); lzo_write_ext(out, match.length - 9); }
                else buf_u8(out, flag | (uint32_t)(match.length - 2));
                buf_u8(out, (uint32_t)(plain | (distance << 2)));
                buf_u8(out, (uint32_t)(distance >> 6));
            }
            if (sp + plain > n) { rc = -2; break; }
            buf_put(out, src + sp, (size_t)plain);
            sp += plain;
        }
        match = next;
        next = mf_find(&m, src, n);
    }
    buf_u8(out, 0x11); buf_u8(out, 0); buf_u8(out, 0);
    mf_free(&m);
    return rc;
}

/* Snappy.CompressHeaderless  Snappy.cs:124-203 */
static void enc_snappy(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_SNAPPY_RAW, NULL, st);
    mf_t m; mf_init(&m, &p, st);
    int v = n;
    while (v >= 0x80) { buf_u8(out, (uint32_t)(v | 0x80)); v >>= 7; }
    buf_u8(out, (uint32_t)v);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        if (plain > 0) {
            if (plain <= 60) buf_u8(out, (uint32_t)((plain - 1) << 2));
            else {
                int len = plain - 1;
                if (len <= 0xFF) { buf_u8(out, 60 << 2); buf_u8(out, (uint32_t)len); }
                else if (len <= 0xFFFF) { buf_u8(out, 61 << 2); buf_u16le(out, (uint32_t)len); }
                else if (len <= 0xFFFFFF) { buf_u8(out, 62 << 2); buf_u8(out, (uint32_t)len); buf_u8(out, (uint32_t)len >> 8); buf_u8(out, (uint32_t)len >> 16); }
                else { buf_u8(out, 63 << 2); buf_u32le(out, (uint32_t)len); }
            }
            buf_put(out, src + sp, (size_t)plain);
            sp += plain;
        }
        if (match.length == 0) break;
        sp += match.length;
        if (match.distance < 2048 && match.length >= 4 && match.length <= 11) {
            buf_u8(out, (uint32_t)(1 | ((match.length - 4) << 2) | ((match.distance >> 8) << 5)));
            buf_u8(out, (uint32_t)match.distance);
        } else {
            buf_u8(out, (uint32_t)(2 | ((match.length - 1) << 2)));
            buf_u16le(out, (uint32_t)match.distance & 0xFFFF);
        }
    }
    mf_free(&m);
}

/* BLZ.CompressHeaderless  Nintendo/BLZ.cs:137-216 in stream order (src = the raw data reversed, :148-150): the managed code
 * fills its buffer from the end, flag byte first, so read backwards it is a flag byte (MSB first, 1 = match) followed by
 * its tokens; a flag byte exists only if a token follows it (:180-197). */
static void enc_blz(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_BLZ, NULL, st);
    mf_t m; mf_init(&m, &p, st); fw_t flag; fw_init(&flag, out, 1);
    int sp = 0;
    while (sp < n) {                                                                     /* :152 */
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(&flag.buffer, src[sp++]); fw_bit(&flag, 0); }
        if (match.length == 0) break;
        buf_u16be(&flag.buffer, (uint32_t)(((match.length - 3) << 12) | ((match.distance - 3) & 0xFFF)) & 0xFFFF);
        sp += match.length;
        fw_bit(&flag, 1);
    }
    fw_dispose(&flag); mf_free(&m);
}

/* HIG literal block  HIG.cs:234-243, :303-313: count - 2 in a byte, or 0 + the count as (ushort) */
static void hig_put_raw(buf_t* out, int plain) {
    if (plain <= 255 + 2) buf_u8(out, (uint32_t)(plain - 2) & 0xFF);
    else { buf_u8(out, 0); buf_u16le(out, (uint32_t)plain & 0xFFFF); }
}
/* HIG.CompressHeaderless  Specialized/HIG.cs:214-323 */
static void enc_hig(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_HIG, NULL, st);
    mf_t m; mf_init(&m, &p, st);
    lzmatch_t next, match = mf_find(&m, src, n);
    int plain = match.offset;
    if (plain < 2) {                                                                     /* the initial block holds at least 2 bytes  :223-232 */
        match.length -= plain + 1; match.offset = 2;
        plain = 2;
        if (match.length < p.minLen) { match = mf_find(&m, src, n); plain = match.offset; }
    }
    hig_put_raw(out, plain);
    buf_put(out, src, (size_t)plain);                                                    /* (inputs below 2 bytes give a count byte of 0xFE / 0xFF: not decodable, as in the managed code) */
    int sp = plain;
    while (match.length != 0) {
        next = mf_find(&m, src, n);
        plain = next.offset - (match.offset + match.length);
        int b = plain == 0 ? 3 : (plain == 1 ? 1 : (plain == 2 ? 2 : 0));
        if (match.distance <= 0x7FF && match.length <= 5 + 4) {
            b |= ((match.length - 4) << 5) | ((match.distance >> 6) & 0x1C);
            buf_u8(out, (uint32_t)b & 0xFF);
        } else {
            if (match.distance <= 0x3FFF && match.length <= 31 + 4) buf_u8(out, (uint32_t)(0xC0 | (match.length - 4)));
            else {
                int length = match.length <= 15 + 3 ? match.length - 3 : 0;
                buf_u8(out, (uint32_t)(0xE0 | ((match.distance >> 10) & 0x10) | length));
                if (length == 0) {
                    if (match.length <= 255 + 18) buf_u8(out, (uint32_t)(match.length - 18) & 0xFF);
                    else { buf_u8(out, 0); buf_u16be(out, (uint32_t)match.length & 0xFFFF); }
                }
            }
            b |= (match.distance >> 6) & 0xFC;
            buf_u8(out, (uint32_t)b & 0xFF);
        }
        buf_u8(out, (uint32_t)match.distance & 0xFF);
        sp += match.length;
        if (plain != 0) {
            if (plain > 2) hig_put_raw(out, plain);
            buf_put(out, src + sp, (size_t)plain);
            sp += plain;
        }
        match = next;
    }
    mf_free(&m);
}

/* LZShrek.CompressHeaderless  Activision/LZShrek.cs:121-174: a group = flag (literal count field << 3 | matches - 1), the
 * literals, then the matches that follow each other without a gap (at most 8) */
static void enc_lzshrek(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_LZSHREK, NULL, st);
    mf_t m; mf_init(&m, &p, st);
    int sp = 0;
    uint8_t buffer[8 * 4]; int blen;
    lzmatch_t match = mf_find(&m, src, n);
    while (sp != n) {                                                                    /* :129 */
        int plain = match.offset - sp, compressedLength = 0;
        const uint8_t* unc = src + sp;
        sp += plain;
        blen = 0;
        while (match.length != 0 && compressedLength < 8 && match.offset == sp) {        /* :136 */
            int lengthflag = match.length > 7 ? 0 : match.length;
            int distanceflag = match.distance > 30 ? (match.distance > 286 ? 0x1F : 0x1E) : match.distance - 1;
            buffer[blen++] = (uint8_t)((distanceflag << 3) | lengthflag);
            if (lengthflag == 0) buffer[blen++] = (uint8_t)(match.length - 7);
            if (distanceflag == 0x1E) buffer[blen++] = (uint8_t)(match.distance - 31);
            else if (distanceflag == 0x1F) { uint32_t v = (uint32_t)(match.distance - 287) & 0xFFFF; buffer[blen++] = (uint8_t)v; buffer[blen++] = (uint8_t)(v >> 8); }
            sp += match.length;
            match = mf_find(&m, src, n);
            if (match.length == 0 || compressedLength >= 7 || match.offset != sp) break;  /* :155-157 */
            compressedLength++;
        }
        int uflag = plain > 29 ? (plain > 285 ? 0x1F : 0x1E) : plain;
        buf_u8(out, (uint32_t)((uflag << 3) | compressedLength) & 0xFF);
        if (uflag == 0x1E) buf_u8(out, (uint32_t)(plain - 30) & 0xFF);
        else if (uflag == 0x1F) buf_u16le(out, (uint32_t)(plain - 286) & 0xFFFF);
        buf_put(out, unc, (size_t)plain);
        buf_put(out, buffer, (size_t)blen);
    }
    buf_u32le(out, 0);                                                                   /* destination.Write(0)  :173 */
    mf_free(&m);
}

/* WFLZ.CompressHeaderless  WayForward/WFLZ.cs:161-196 */
static void enc_wflz(const alz_settings* st, const uint8_t* src, int n, buf_t* out, int big) {
    fmt_props p = props_for(ALZ_FMT_WFLZ, NULL, st);
    mf_t m; mf_init(&m, &p, st);
    int sp = 0;
    lzmatch_t match = { 0, 0, 0 }, next = mf_find(&m, src, n);
    int plain = next.offset;
    for (;;) {
        uint32_t bp = (uint32_t)(plain < 255 ? plain : 255), d = (uint32_t)match.distance & 0xFFFF;
        if (big) buf_u16be(out, d); else buf_u16le(out, d);
        buf_u8(out, match.length == 0 ? 0u : (uint32_t)(match.length - 4));
        buf_u8(out, bp);
        plain -= (int)bp;
        sp += match.length;
        buf_put(out, src + sp, bp);
        sp += (int)bp;
        if (plain == 0) {
            if (sp == n) break;
            match = next;
            next = mf_find(&m, src, n);
            plain = next.offset - (match.offset + match.length);
        } else { match.offset = 0; match.distance = 0; match.length = 0; }
    }
    buf_u32le(out, 0);                                                                   /* end block */
    mf_free(&m);
}

/* RefPack.CompressHeaderless  EA/RefPack.cs:247-303: three property sets (long / medium / short form), the finder scores a
 * candidate with the first one that admits it (LzChainMatchFinder.cs:301-321) */
static void enc_refpack(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    const fmt_props props[3] = { { 17, 1028, 5, 0x20000, 1 }, { 14, 67, 4, 0x4000, 1 }, { 10, 10, 3, 0x400, 1 } };   /* RefPack.cs:29-34 */
    mf_t m; mf_init_multi(&m, props, 3, st);
    int sp = 0, plain = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        plain = match.offset - sp;
        while (plain > 3) {                                                              /* :257-265 */
            int copyflag = (plain > 0x70 ? 0x70 : plain) / 4 - 1;
            buf_u8(out, (uint32_t)(0xE0 | copyflag));
            copyflag = copyflag * 4 + 4;
            buf_put(out, src + sp, (size_t)copyflag);
            sp += copyflag; plain -= copyflag;
        }
        if (match.length == 0) break;
        int d1 = match.distance - 1;
        if (match.length <= 10 && match.distance <= 0x400) {
            buf_u8(out, (uint32_t)(plain | ((d1 & 0x300) >> 3) | ((match.length - 3) << 2)));
            buf_u8(out, (uint32_t)d1 & 0xFF);
        } else if (match.length >= 4 && match.length <= 67 && match.distance <= 0x4000) {
            buf_u8(out, (uint32_t)(0x80 | (match.length - 4)));
            buf_u8(out, (uint32_t)((d1 >> 8) | (plain << 6)) & 0xFF);
            buf_u8(out, (uint32_t)d1 & 0xFF);
        } else {
            buf_u8(out, (uint32_t)(0xC0 | ((d1 >> 16) << 4) | (((match.length - 5) >> 8) << 2) | plain) & 0xFF);
            buf_u8(out, (uint32_t)(d1 >> 8) & 0xFF);
            buf_u8(out, (uint32_t)d1 & 0xFF);
            buf_u8(out, (uint32_t)(match.length - 5) & 0xFF);
        }
        buf_put(out, src + sp, (size_t)plain);
        sp += plain + match.length;
        plain = 0;
    }
    buf_u8(out, (uint32_t)(0xFC | plain));                                               /* :300-301 */
    buf_put(out, src + sp, (size_t)plain);
    mf_free(&m);
}

/* LZ02.CompressHeaderless  Camelot/LZ02.cs:117-151 */
static void enc_lz02(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_LZ02, NULL, st);
    mf_t m; mf_init(&m, &p, st); fw_t flag; fw_init(&flag, out, 1);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(&flag.buffer, src[sp++]); fw_bit(&flag, 0); }
        if (match.length == 0) break;
        int length = match.length > 16 ? 0 : match.length - 1;
        buf_u8(&flag.buffer, (uint32_t)(((match.distance >> 8) << 4) | length) & 0xFF);
        buf_u8(&flag.buffer, (uint32_t)match.distance & 0xFF);
        if (length == 0) buf_u8(&flag.buffer, (uint32_t)(match.length - 17));
        sp += match.length;
        fw_bit(&flag, 1);
    }
    buf_u8(&flag.buffer, 0); buf_u8(&flag.buffer, 0); fw_bit(&flag, 1);                  /* terminator  :147-149 */
    fw_dispose(&flag); mf_free(&m);
}

/* CNS.CompressHeaderless  Specialized/CNS.cs:111-141 (its FlagWriter never receives a bit: nothing written by it) */
static void enc_cns(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_CNS, NULL, st);
    mf_t m; mf_init(&m, &p, st);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) {
            int length = plain < 127 ? plain : 127;
            buf_u8(out, (uint32_t)length); buf_put(out, src + sp, (size_t)length);
            sp += length; plain -= length;
        }
        if (match.length == 0) break;
        buf_u8(out, (uint32_t)(0x80 | (match.length - 3)));
        buf_u8(out, (uint32_t)(match.distance - 1));
        sp += match.length;
    }
    mf_free(&m);
}

/* CLZ0.CompressHeaderless  Marvelous/CLZ0.cs:99-130 */
static void enc_clz0(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_CLZ0, NULL, st);
    mf_t m; mf_init(&m, &p, st); fw_t flag; fw_init(&flag, out, 0);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) { plain--; buf_u8(&flag.buffer, src[sp++]); fw_bit(&flag, 0); }
        if (match.length == 0) break;
        int delta = 0x1000 - match.distance;
        buf_u8(&flag.buffer, (uint32_t)delta & 0xFF);
        buf_u8(&flag.buffer, (uint32_t)((match.length - 3) | ((delta >> 8) << 4)) & 0xFF);
        sp += match.length;
        fw_bit(&flag, 1);
    }
    fw_dispose(&flag); mf_free(&m);
}

/* CNX2.CompressHeaderless  Sega/CNX2.cs:140-172: FlagWriter(destination, Endian.Little), WriteInt(v, 2) = bit 0 first */
static void enc_cnx2(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    fmt_props p = props_for(ALZ_FMT_CNX2, NULL, st);
    mf_t m; mf_init(&m, &p, st); fw_t flag; fw_init(&flag, out, 0);
    int sp = 0;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain != 0) {
            int length = plain < 255 ? plain : 255;
            if (length == 1) { buf_u8(&flag.buffer, src[sp]); fw_bit(&flag, 1); fw_bit(&flag, 0); }
            else { buf_u8(&flag.buffer, (uint32_t)length); buf_put(&flag.buffer, src + sp, (size_t)length); fw_bit(&flag, 1); fw_bit(&flag, 1); }
            sp += length; plain -= length;
        }
        if (match.length == 0) break;
        buf_u16be(&flag.buffer, (uint32_t)((((match.distance - 1) & 0x7FF) << 5) | ((match.length - 4) & 0x1F)));
        sp += match.length;
        fw_bit(&flag, 0); fw_bit(&flag, 1);
    }
    fw_dispose(&flag); mf_free(&m);
}

/* FastLZ.CompressHeaderless  Formats/Common/FastLZ.cs:162-245.  Level 2 is chosen for sources >= 64 KiB at Quality > 4 when the
 * caller sets MaxWindowBits > 13 (:169-175); its finder takes two property sets -- short matches (distance <= 0x1FFF, min 3) and
 * long ones (<= 0x11FFF, min 5), :23-27 -- and ScoreMatch admits a candidate with the first set that fits it. */
static int enc_fastlz(const alz_settings* st, const uint8_t* src, int n, buf_t* out) {
    const int level2 = n >= 0x10000 && st->quality > 4 && st->max_window_bits > 13;
    const fmt_props lz1 = { 13, 255 + 3 + 6, 3, 0x2000, 1 };                              /* :22 */
    const fmt_props lz2[2] = { { 13, 0x7FFFFFFF, 3, 0x1FFF, 1 }, { 17, 0x7FFFFFFF, 5, 0x11FFF, 1 } };   /* WindowsBits = ceil(log2(windowsSize))  LzProperties.cs:58 */
    mf_t m;
    if (level2) mf_init_multi(&m, lz2, 2, st);
    else {
        fmt_props p = lz1;
        if (st->min_distance > 0) p.minDist = st->min_distance;
        mf_init_multi(&m, &p, 1, st);                                                    /* (one set: _lzProperties stays null, :52-53) */
    }
    int sp = 0, level2First = level2;
    for (;;) {
        lzmatch_t match = mf_find(&m, src, n);
        int plain = match.offset - sp;
        while (plain > 0) {                                                              /* :181-198 */
            int chunk = plain < 32 ? plain : 32;
            uint32_t ctrl = (uint32_t)(chunk - 1);
            if (level2First) { ctrl |= 1u << 5; level2First = 0; }                        /* the level tag rides on the first literal run */
            buf_u8(out, ctrl);
            buf_put(out, src + sp, (size_t)chunk);
            sp += chunk; plain -= chunk;
        }
        if (match.length == 0) break;
        int length = match.length - 3, distance = match.distance - 1;                    /* :204-206 */
        int shortDistance = level2 ? (distance < 0x1FFF ? distance : 0x1FFF) : distance;
        buf_u8(out, (uint32_t)(((((length < 6 ? length : 6) + 1) << 5) | (shortDistance >> 8)) & 0xFF));
        if (length >= 6) {                                                               /* :214-225 */
            length -= 6;
            while (level2 && length >= 255) { buf_u8(out, 255); length -= 255; }
            buf_u8(out, (uint32_t)length & 0xFF);
        }
        buf_u8(out, (uint32_t)shortDistance & 0xFF);
        if (level2 && distance >= 0x1FFF) {                                              /* :229-235 */
            distance -= 0x1FFF;
            buf_u8(out, (uint32_t)(distance >> 8) & 0xFF);
            buf_u8(out, (uint32_t)distance & 0xFF);
        }
        sp += match.length;
    }
    mf_free(&m);
    return 0;
}

int64_t oracle_encode_stream(uint32_t format, const alz_lz_properties* props, const alz_settings* settings,
                             const uint8_t* src, size_t n, uint8_t* dst, size_t cap, alz_encode_aux* aux) {
    buf_t out = { dst, 0, cap, 0, 0 };
    alz_settings dflt = { 8, 0, 0, 0 };
    const alz_settings* st = settings ? settings : &dflt;
    int rc = 0;
    if (aux) { aux->aux0 = 0; aux->aux1 = 0; }
    switch (format) {
    case ALZ_FMT_LZSS: enc_lzss(props, st, src, (int)n, &out); break;
    case ALZ_FMT_LZ10: enc_lz10(st, src, (int)n, &out); break;
    case ALZ_FMT_LZ11: enc_lz11(st, src, (int)n, &out); break;
    case ALZ_FMT_LZ40: enc_lz40(st, src, (int)n, &out); break;
    case ALZ_FMT_YAZ0: {                                                                 /* Yaz0.cs:94-98 */
        fw_t flag; fw_init(&flag, &out, 1);
        enc_yay0_core(st, src, (int)n, &flag.buffer, &flag.buffer, &flag, ALZ_FMT_YAZ0);
        fw_dispose(&flag); break;
    }
    case ALZ_FMT_LZHUDSON: {                                                             /* LZHudson.cs:55-59 */
        fw_t flag; fw_init(&flag, &out, 1); fw_set_width(&flag, 4);
        enc_yay0_core(st, src, (int)n, &flag.buffer, &flag.buffer, &flag, ALZ_FMT_LZHUDSON);
        fw_dispose(&flag); break;
    }
    case ALZ_FMT_SMSR00: {                                                               /* SMSR00.cs:133-137, :52-66 */
        buf_t codes = buf_owned(), uncomp = buf_owned();
        fw_t flag; fw_init(&flag, &codes, 1); fw_set_width(&flag, 2);
        enc_mio0_core(st, src, (int)n, &flag.buffer, &uncomp, &flag);
        fw_dispose(&flag);
        if (aux) { aux->aux0 = (uint32_t)codes.len; aux->aux1 = 0; }
        buf_put(&out, codes.p, codes.len); buf_put(&out, uncomp.p, uncomp.len);
        buf_free(&codes); buf_free(&uncomp);
        break;
    }
    case ALZ_FMT_YAY0: case ALZ_FMT_MIO0: {                                              /* Yay0.cs:62-77 / MIO0.cs:64-79 */
        buf_t flags = buf_owned(), comp = buf_owned(), uncomp = buf_owned();
        fw_t flag; fw_init(&flag, &flags, 1);
        if (format == ALZ_FMT_YAY0) enc_yay0_core(st, src, (int)n, &comp, &uncomp, &flag, ALZ_FMT_YAY0);
        else enc_mio0_core(st, src, (int)n, &comp, &uncomp, &flag);
        fw_dispose(&flag);
        if (aux) { aux->aux0 = (uint32_t)flags.len; aux->aux1 = (uint32_t)(flags.len + comp.len); }
        buf_put(&out, flags.p, flags.len); buf_put(&out, comp.p, comp.len); buf_put(&out, uncomp.p, uncomp.len);
        buf_free(&flags); buf_free(&comp); buf_free(&uncomp);
        break;
    }
    case ALZ_FMT_PRS_BE: enc_prs(st, src, (int)n, &out, 1); break;
    case ALZ_FMT_PRS_LE: enc_prs(st, src, (int)n, &out, 0); break;
    case ALZ_FMT_LZ4_BLOCK: rc = enc_lz4(st, src, (int)n, &out); break;
    case ALZ_FMT_LZO: rc = enc_lzo(st, src, (int)n, &out); break;
    case ALZ_FMT_SNAPPY_RAW: enc_snappy(st, src, (int)n, &out); break;
    case ALZ_FMT_FASTLZ: rc = enc_fastlz(st, src, (int)n, &out); break;
    case ALZ_FMT_CNX2: enc_cnx2(st, src, (int)n, &out); break;
    case ALZ_FMT_BLZ: enc_blz(st, src, (int)n, &out); break;
    case ALZ_FMT_CLZ0: enc_clz0(st, src, (int)n, &out); break;
    case ALZ_FMT_CNS: enc_cns(st, src, (int)n, &out); break;
    case ALZ_FMT_LZ02: enc_lz02(st, src, (int)n, &out); break;
    case ALZ_FMT_REFPACK: enc_refpack(st, src, (int)n, &out); break;
    case ALZ_FMT_WFLZ: enc_wflz(st, src, (int)n, &out, 0); break;
    case ALZ_FMT_LZSHREK: enc_lzshrek(st, src, (int)n, &out); break;
    case ALZ_FMT_HIG: enc_hig(st, src, (int)n, &out); break;
    case ALZ_FMT_WFLZ_BE: enc_wflz(st, src, (int)n, &out, 1); break;
    default: return -2;
    }
    if (out.fail) return -1;
    if (rc < 0) return rc;
    return (int64_t)out.len;
}

/* ============================================================ containers */

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint32_t rd32(const uint8_t* p, int big) { return big ? be32(p) : rd32le(p); }
static uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00) | ((v << 8) & 0xFF0000) | (v << 24); }
static void wr32(uint8_t* p, uint32_t v, int big) {
    if (big) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
    else { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
}

static const uint8_t AKLZ_MAGIC[12] = { 'A', 'K', 'L', 'Z', '~', '?', 'Q', 'd', '=', 0xCC, 0xCC, 0xCD };   /* Sega/AKLZ.cs:16 */
static const uint8_t LZON_MAGIC[8] = { 'L', 'Z', 'O', 'n', 0x00, 0x2F, 0xF1, 0x71 };                           /* Nintendo/LZOn.cs:17 */

/* header of LZ10 / LZ11: id byte, u24 LE size, 0 => u32 LE (LZ10.cs:47-57).  Returns header length or -1. */
static int nin_header(const uint8_t* src, size_t len, uint8_t id, uint32_t* size) {
    if (len < 4 || src[0] != id) return -1;
    uint32_t s = (uint32_t)src[1] | ((uint32_t)src[2] << 8) | ((uint32_t)src[3] << 16);
    if (s != 0) { *size = s; return 4; }
    if (len < 8) return -1;
    *size = rd32le(src + 4); return 8;
}

static const uint8_t SNAPPY_ID[10] = { 0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59 };

/* RefPack.InternalReadHeader  EA/RefPack.cs:77-102: flags + 0xFB + size [+ compressed size], optionally behind a u32 compressed
 * size (version 2).  Returns the header length, or an ALZ_E_* code. */
static int refpack_header(const uint8_t* src, size_t len, uint32_t* size) {
    size_t pos = 0;
    if (len < 2) return ALZ_E_FORMAT;
    if (src[1] != 0xFB) {                                            /* not version 1 / 3: a pre-header must follow  :81-90 */
        if (len < 6 || src[4] != 0x10 || src[5] != 0xFB) return ALZ_E_FORMAT;
        pos = 4;
    }
    uint8_t flag = src[pos]; pos += 2;
    if (!(flag & 0x10)) return ALZ_E_UNSUPPORTED;                    /* NotSupportedException("No supported Flag")  :92-93 */
    int wide = (flag & 0x80) != 0, n = wide ? 4 : 3;
    if (len < pos + (size_t)n) return ALZ_E_FORMAT;
    *size = wide ? be32(src + pos) : (((uint32_t)src[pos] << 16) | ((uint32_t)src[pos + 1] << 8) | src[pos + 2]);
    pos += (size_t)n;
    if (flag & 1) pos += (size_t)n;                                  /* StoresCompressedSize */
    return (int)pos;
}

/* FastLZ.Validate  Formats/Common/FastLZ.cs:246-291 (IsMatch: Position + 4 < Length && Validate) */
static int fastlz_validate(const uint8_t* s, size_t n) {
    size_t pos = 0;
    int ctrl = pos < n ? s[pos++] : -1;
    int level = (ctrl >> 5) + 1;
    if (level != 0 && level != 1) return 0;                         /* (sic: only a first byte below 0x20 passes) */
    int i = 3; long buffer = 0;
    while (ctrl != -1) {
        if (ctrl >= 32) {
            int length = (ctrl >> 5) - 1, distance = (ctrl & 31) << 8;
            if (length == 6) length += pos < n ? s[pos++] : -1;
            ctrl = pos < n ? s[pos++] : -1;
            distance |= ctrl;
            if (ctrl == -1 || length < 0) return 0;
            if (distance + 1 > buffer) return 0;
            if (i-- == 0) return 1;
            buffer += length + 3;
        } else {
            ctrl++;
            buffer += ctrl;
            if (pos + (size_t)ctrl > n) return 0;
            pos += (size_t)ctrl;
        }
        if (pos >= n) return i < 3;
        ctrl = s[pos++];
    }
    return 0;
}
int oracle_fastlz_validate(const uint8_t* s, size_t n) { return n > 4 && fastlz_validate(s, n); }

/* LZ00.StreamTransformer  Sega/LZ00.cs:128-141: every byte that passes through it is XORed with the next element of a
 * keystream; GenerateNextKey's shift/add chain multiplies by 3, 95, 3041, 389247, then 63, 15, 3 = 1103515245 and adds
 * 12345.  The WRITE side (WriteByte / Write(span), :150, :177-199) transforms the body bytes strictly in order, so byte i
 * of the body carries keystream element i + 1; decoding is the inverse.  (The modern-.NET Read(Span) override drops a
 * byte of the base stream per call, :156-162: that only matters if LZSS reads spans through the transformer, which is
 * decided inside the un-vendored AuroraLib.Core -- the decoder here is pinned by the encoder, i.e. by round trip.) */
static void lz00_keystream(uint8_t* p, size_t n, uint32_t key) {
    for (size_t i = 0; i < n; i++) {
        key = key * 1103515245u + 12345u;
        const uint32_t t = (key >> 16) & 0x7FFFu;
        p[i] ^= (uint8_t)(((t << 8) - t) >> 15);
    }
}

int oracle_container_decompressed_size(uint32_t container, const alz_container_options* opt, const uint8_t* src, size_t len, uint32_t* size_out) {
    int big = opt ? (int)opt->big_endian : 1;
    switch (container) {
    case ALZ_C_LZSS: if (len < 8 || memcmp(src, "LZSS", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 4); return 0;           /* LZSS.cs:45-50 */
    case ALZ_C_LZ10: return nin_header(src, len, 0x10, size_out) < 0 ? ALZ_E_FORMAT : 0;
    case ALZ_C_LZ11: return nin_header(src, len, 0x11, size_out) < 0 ? ALZ_E_FORMAT : 0;
    case ALZ_C_LZ40: return nin_header(src, len, 0x40, size_out) < 0 ? ALZ_E_FORMAT : 0;          /* LZ40.cs:40-52 */
    case ALZ_C_LZHUDSON: if (len < 4) return ALZ_E_FORMAT; *size_out = be32(src); return 0;                                        /* LZHudson.cs:30-31 */
    case ALZ_C_SMSR00: if (len < 12 || memcmp(src, "SMSR00", 6)) return ALZ_E_FORMAT; *size_out = be32(src + 8); return 0;          /* SMSR00.cs:33-39 */
    case ALZ_C_LZ60: return nin_header(src, len, 0x60, size_out) < 0 ? ALZ_E_FORMAT : 0;          /* LZ60.cs:29-41 */
    case ALZ_C_LZ00: if (len < 52 || memcmp(src, "LZ00", 4)) return ALZ_E_FORMAT; *size_out = rd32le(src + 48); return 0;      /* Sega/LZ00.cs:31-37 */
    case ALZ_C_CNX2: if (len < 16 || memcmp(src, "CNX\x02", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 12); return 0;    /* Sega/CNX2.cs:36-42 */
    case ALZ_C_CLZ0: if (len < 16 || memcmp(src, "CLZ\0", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 12); return 0;      /* Marvelous/CLZ0.cs:33-39 */
    case ALZ_C_CNS: if (len < 12 || memcmp(src, "@CNS", 4)) return ALZ_E_FORMAT; *size_out = rd32le(src + 8); return 0;         /* Specialized/CNS.cs:36-42 */
    case ALZ_C_LZ02: if (len < 4 || (src[0] != 1 && src[0] != 2)) return ALZ_E_FORMAT; *size_out = ((uint32_t)src[1] << 16) | ((uint32_t)src[2] << 8) | src[3]; return 0;   /* Camelot/LZ02.cs:49-58 */
    case ALZ_C_REFPACK: { int h = refpack_header(src, len, size_out); return h < 0 ? h : 0; }                                  /* EA/RefPack.cs:56-62 */
    case ALZ_C_LZSHREK: if (len < 8) return ALZ_E_FORMAT; *size_out = rd32le(src + 4); return 0;                                 /* Activision/LZShrek.cs:28-33 */
    case ALZ_C_HIG: if (len < 0x40 || memcmp(src, "HIG!", 4)) return ALZ_E_FORMAT; *size_out = rd32le(src + 0x3C); return 0;       /* Specialized/HIG.cs:39-45 */
    case ALZ_C_WFLZ: if (len < 12 || memcmp(src, "WFLZ", 4)) return ALZ_E_FORMAT; *size_out = (opt && opt->big_endian) ? be32(src + 8) : rd32le(src + 8); return 0;   /* WFLZ.cs:41-48 (FormatByteOrder defaults to little) */
    case ALZ_C_BLZ: {                                                                                                       /* Nintendo/BLZ.cs:32-41 */
        if (len < 8 || src[len - 5] < 8) return ALZ_E_FORMAT;
        uint32_t csz = (uint32_t)src[len - 8] | ((uint32_t)src[len - 7] << 8) | ((uint32_t)src[len - 6] << 16);
        *size_out = rd32le(src + len - 4) + csz; return 0;
    }
    case ALZ_C_YAZ0: if (len < 8 || memcmp(src, "Yaz0", 4)) return ALZ_E_FORMAT; *size_out = rd32(src + 4, big); return 0;      /* Yaz0.cs:50-55 */
    case ALZ_C_YAY0: if (len < 8 || memcmp(src, "Yay0", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 4); return 0;           /* Yay0.cs:41-47 reads Endian.Big */
    case ALZ_C_MIO0: if (len < 8 || memcmp(src, "MIO0", 4)) return ALZ_E_FORMAT; *size_out = rd32(src + 4, big); return 0;      /* MIO0.cs:41-48 */
    /* header-only wrappers */
    case ALZ_C_GCLZ: case ALZ_C_CXLZ: case ALZ_C_COMP: {
        const char* m = container == ALZ_C_GCLZ ? "GCLZ" : container == ALZ_C_CXLZ ? "CXLZ" : "COMP";
        if (len < 4 || memcmp(src, m, 4)) return ALZ_E_FORMAT;
        return oracle_container_decompressed_size(container == ALZ_C_COMP ? ALZ_C_LZ11 : ALZ_C_LZ10, opt, src + 4, len - 4, size_out);
    }
    case ALZ_C_LZ_3DS: if (len < 8 || memcmp(src, "3DS-LZ\r\n", 8)) return ALZ_E_FORMAT; return oracle_container_decompressed_size(ALZ_C_LZ10, opt, src + 8, len - 8, size_out);
    case ALZ_C_YAZ1: if (len < 8 || memcmp(src, "Yaz1", 4)) return ALZ_E_FORMAT; *size_out = rd32(src + 4, big); return 0;
    case ALZ_C_AKLZ: if (len < 16 || memcmp(src, AKLZ_MAGIC, 12)) return ALZ_E_FORMAT; *size_out = be32(src + 12); return 0;     /* Sega/AKLZ.cs:33-38 */
    case ALZ_C_LZ01: if (len < 12 || memcmp(src, "LZ01", 4)) return ALZ_E_FORMAT; *size_out = rd32le(src + 8); return 0;          /* Sega/LZ01.cs:37-43 */
    case ALZ_C_LZSEGA: if (len < 8) return ALZ_E_FORMAT; *size_out = rd32le(src + 4); return 0;                                  /* Sega/LZSega.cs:41-46 */
    case ALZ_C_LEVEL5LZSS: if (len < 16 || memcmp(src, "SSZL", 4)) return ALZ_E_FORMAT; *size_out = rd32le(src + 12); return 0;  /* Level5/Level5LZSS.cs:33-39 */
    case ALZ_C_LZON: if (len < 12 || memcmp(src, LZON_MAGIC, 8)) return ALZ_E_FORMAT; *size_out = be32(src + 8); return 0;       /* Nintendo/LZOn.cs:33-38 */
    case ALZ_C_LZ77: {                                                                                                           /* Nintendo/LZ77.cs:45-54 */
        if (len < 8 || memcmp(src, "LZ77", 4)) return ALZ_E_FORMAT;
        uint32_t sz = (uint32_t)src[5] | ((uint32_t)src[6] << 8) | ((uint32_t)src[7] << 16);
        if (sz == 0) { if (len < 12) return ALZ_E_FORMAT; sz = rd32le(src + 8); }
        *size_out = sz; return 0;
    }
    case ALZ_C_LEVEL5: if (len < 5) return ALZ_E_FORMAT; *size_out = src[4] == 0x78 ? rd32le(src) : rd32le(src) >> 3; return 0;  /* Level5/Level5.cs:55-60 */
    case ALZ_C_MDB4: if (len < 12 || memcmp(src, "MDB4", 4)) return ALZ_E_FORMAT; *size_out = rd32le(src + 8); return 0;            /* Specialized/MDB4.cs:25-31 */
    case ALZ_C_FCMP: case ALZ_C_IECP: case ALZ_C_SDPC:                                                                             /* Marvelous/FCMP.cs:28-33 */
        if (len < 8 || memcmp(src, container == ALZ_C_FCMP ? "FCMP" : container == ALZ_C_IECP ? "IECP" : "SDPC", 4)) return ALZ_E_FORMAT;
        *size_out = rd32le(src + 4); return 0;
    case ALZ_C_GCZ: if (len < 4) return ALZ_E_FORMAT; *size_out = rd32le(src); return 0;                                            /* Konami/GCZ.cs:30 */
    case ALZ_C_ECD:                                                                                                                /* Specialized/ECD.cs:34-43 */
        if (len < 16 || memcmp(src, "ECD", 3)) return ALZ_E_FORMAT;
        *size_out = (uint64_t)be32(src + 8) + 0x10 > len ? 0u : be32(src + 12); return 0;
    case ALZ_C_LZ4_FRAME:   /* not an IProvidesDecompressedSize in the reference; offered where the descriptor carries ContentSize */
        if (len < 15 || rd32le(src) != 0x184D2204u || !(src[4] & 8)) return ALZ_E_UNSUPPORTED;
        if (rd32le(src + 10) != 0) return ALZ_E_UNSUPPORTED;
        *size_out = rd32le(src + 6); return 0;
    case ALZ_C_SNAPPY: {    /* same: the sum of the chunks' declared sizes */
        if (len < 10 || memcmp(src, SNAPPY_ID, 10)) return ALZ_E_FORMAT;
        size_t pos = 10; uint64_t total = 0;
        while (pos + 4 <= len) {
            uint32_t type = src[pos], cl = (uint32_t)src[pos + 1] | ((uint32_t)src[pos + 2] << 8) | ((uint32_t)src[pos + 3] << 16); pos += 4;
            if (type == 0) { cur_t c = { src + pos + 4, pos + 4 <= len ? (uint32_t)(len - pos - 4) : 0, 0, 0 }; total += snappy_varint(&c); }
            else if (type == 1) total += cl >= 4 ? cl - 4 : 0;
            pos += cl;
        }
        *size_out = (uint32_t)total; return 0;
    }
    default: return ALZ_E_UNSUPPORTED;
    }
}

static void run_stream(uint32_t format, const alz_lz_properties* lz, const uint8_t* body, uint32_t body_len, uint32_t size,
                       uint32_t aux0, uint32_t aux1, uint8_t* dst, size_t dst_cap, alz_result* r) {
    alz_stream s; memset(&s, 0, sizeof(s));
    s.src_off = 0; s.src_len = body_len; s.dst_off = 0; s.dst_cap = dst_cap > 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)dst_cap;
    s.decom_len = size; s.aux0 = aux0; s.aux1 = aux1; s.format = format;
    decode_one(lz, &s, body, dst, r, 0);
}

/* ============================================================ LZ4 frame / legacy and Snappy framing */

static int lz4_magic_defined(uint32_t v) {                     /* Enum.IsDefined(typeof(FrameTypes), v)  LZ4.Frame.cs:50-70 */
    return v == 0x184C2102u || v == 0x184D2204u || (v >= 0x184D2A50u && v <= 0x184D2A5Fu);
}

/* LZ4.Decompress  Formats/Common/LZ4.cs:50-93, ReadLZ4L :96-111, DecompressLZ4FrameHeader  LZ4.Frame.cs:107-174.
   Checksums are verified as with LZ4.HashAlgorithm = XXH32 (what the CLI and the test-suite install). */
static int lz4_file_decompress(const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* dst_len, size_t* src_used, int32_t* status) {
    size_t pos = 0, out = 0; int32_t st = ALZ_ST_OK; int rc = 0;
    while (pos < len && st == ALZ_ST_OK && rc == 0) {
        if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
        uint32_t magic = rd32le(src + pos); pos += 4;
    again:
        if (magic == 0x184C2102u) {                                                       /* legacy: independent blocks */
            if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            uint32_t bs = rd32le(src + pos); pos += 4;
            int next = 0;
            for (;;) {
                if (bs > len - pos) { st = ALZ_ST_INPUT_TRUNCATED; break; }
                alz_result r;
                run_stream(ALZ_FMT_LZ4_BLOCK, NULL, src + pos, bs, 0, 0, 0, dst + out, out < cap ? cap - out : 0, &r);   /* fresh LzWindows per block  LZ4.cs:164 */
                out += r.dst_len; pos += bs;
                if (r.status != ALZ_ST_OK) { st = r.status; break; }
                if (pos >= len) break;                                                    /* ReadByte() == -1 */
                if (src[pos] == 0xFF) { pos++; goto done; }                               /* (sbyte)0xFF == -1: the EOF flag; Decompress returns */
                if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
                bs = rd32le(src + pos); pos += 4;
                if (lz4_magic_defined(bs)) { next = 1; break; }
            }
            if (st != ALZ_ST_OK) break;
            if (next) { magic = bs; goto again; }                                         /* "Mixed LZ4 Legacy and LZ4 Frame" */
            goto done;                                                                    /* blockSize == 0: EOF */
        } else if (magic == 0x184D2204u) {
            size_t frame_start = out;
            if (pos + 2 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            uint32_t flg = src[pos], bd = src[pos + 1]; pos += 2;
            uint32_t bmax;
            switch ((bd & 0x70) >> 4) { case 4: bmax = 0x10000; break; case 5: bmax = 0x40000; break; case 6: bmax = 0x100000; break; case 7: bmax = 0x400000; break; default: return ALZ_E_FORMAT; }
            uint64_t content = 0;
            if (flg & 8) { if (pos + 8 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; } content = (uint64_t)rd32le(src + pos) | ((uint64_t)rd32le(src + pos + 4) << 32); pos += 8; }
            if (flg & 1) { if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; } pos += 4; }
            if (pos + 1 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            pos += 1;                                                                     /* HeaderChecksum: read, not verified */
            if (flg & 1) return ALZ_E_UNSUPPORTED;                                        /* external dictionaries  LZ4.Frame.cs:113-114 */
            /* one window for all blocks of the frame (LZ4.Frame.cs:120): ring mode, exactly LzWindows */
            win_t w; memset(&w, 0, sizeof(w));
            w.W = 65536; w.mask = 65535; w.dst = dst + out; w.cap = out < cap ? cap - out : 0;
            w.ring = (uint8_t*)calloc(w.W, 1);
            for (;;) {
                if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
                uint32_t bsz = rd32le(src + pos); pos += 4;
                if (bsz == 0) break;                                                      /* EndMark */
                int raw = (bsz & 0x80000000u) != 0; uint32_t n = bsz & 0x7FFFFFFFu;
                if (n > bmax) { rc = ALZ_E_FORMAT; break; }                               /* buffer.AsSpan(0, n) on a BlockMaxSize buffer */
                if (n > len - pos) { st = ALZ_ST_INPUT_TRUNCATED; break; }
                const uint8_t* blk = src + pos; pos += n;
                if (flg & 16) {                                                           /* block checksum */
                    if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
                    if (rd32le(src + pos) != oracle_xxh32(blk, n, 0)) { rc = ALZ_E_CHECKSUM; break; }
                    pos += 4;
                }
                if (raw) { uint32_t cl = win_clip(&w, n); win_write(&w, blk, cl); }
                else { cur_t c = { blk, n, 0, 0 }; dec_lz4(&c, &w); if (c.eof) { st = ALZ_ST_INPUT_TRUNCATED; } }
                if (w.overflow) { st = ALZ_ST_OUTPUT_CAPACITY; }
                if (st != ALZ_ST_OK) break;
            }
            win_dispose(&w);
            out += (size_t)win_produced(&w);
            free(w.ring);
            if (rc || st != ALZ_ST_OK) break;
            if ((flg & 8) && (uint64_t)(out - frame_start) != content) { st = ALZ_ST_OUTPUT_SIZE_MISMATCH; break; }   /* LZ4.Frame.cs:152-155 */
            if (flg & 4) {                                                                /* content checksum */
                if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
                if (rd32le(src + pos) != oracle_xxh32(dst + frame_start, out - frame_start, 0)) { rc = ALZ_E_CHECKSUM; break; }
                pos += 4;
            }
        } else if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {                         /* skippable */
            if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            uint32_t n = rd32le(src + pos); pos += 4;
            pos = (uint64_t)pos + n > len ? len : pos + n;                                /* Position += n (may pass the end: loop ends) */
        } else { pos -= 4; break; }                                                       /* not a frame: stop in front of it */
    }
done:
    if (dst_len) *dst_len = out;
    if (src_used) *src_used = pos;
    if (status) *status = st;
    if (rc) return rc;
    return st == ALZ_ST_OK ? 0 : ALZ_E_STREAM;
}

/* LZ4.Compress  LZ4.cs:113-160 (legacy) / CompressLZ4FrameHeader  LZ4.Frame.cs:176-229.
   `Flags &= FrameDescriptorFlags.IsVersion1` (LZ4.Frame.cs:184) leaves only the version bit, so the frame written never
   carries a content size or checksums, whatever LZ4.Flags held. */
static int lz4_file_compress(int legacy, uint32_t block_size, const alz_settings* st, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* dst_len) {
    size_t o = 0;
    if (cap < 16) return ALZ_E_NOMEM;
    if (legacy) { wr32(dst, 0x184C2102u, 0); o = 4; block_size = 0x800000; }             /* (int)BlockMaxSizes.Block4MB * 2 */
    else {
        uint8_t bdb;
        switch (block_size) { case 0: block_size = 0x400000; bdb = 0x70; break; case 0x10000: bdb = 0x40; break; case 0x40000: bdb = 0x50; break;
                              case 0x100000: bdb = 0x60; break; case 0x400000: bdb = 0x70; break; default: return ALZ_E_INVALID; }
        wr32(dst, 0x184D2204u, 0); dst[4] = 0x40; dst[5] = bdb; dst[6] = (uint8_t)((oracle_xxh32(dst + 4, 2, 0) >> 8) & 0xFF); o = 7;
    }
    size_t sp = 0;
    while (sp != n) {
        size_t bl = n - sp < block_size ? n - sp : block_size;
        if (bl < 5) return ALZ_E_INVALID;                                                 /* source.Slice(0, Length - 5) throws  LZ4.cs:208 */
        if (o + 4 > cap) return ALZ_E_NOMEM;
        int64_t c = oracle_encode_stream(ALZ_FMT_LZ4_BLOCK, NULL, st, src + sp, bl, dst + o + 4, cap - o - 4, NULL);
        if (c == -1) {                                                                    /* does not fit: it can still be a stored block */
            if (legacy || o + 4 + bl > cap) return ALZ_E_NOMEM;
            c = (int64_t)block_size;
        } else if (c < 0) return ALZ_E_INVALID;
        if (!legacy && c >= (int64_t)block_size) {                                        /* buffer.Position >= (int)BlockSize: stored */
            wr32(dst + o, (uint32_t)bl | 0x80000000u, 0); memcpy(dst + o + 4, src + sp, bl); o += 4 + bl;
        } else { wr32(dst + o, (uint32_t)c, 0); o += 4 + (size_t)c; }
        sp += bl;
    }
    if (legacy) { if (o + 1 > cap) return ALZ_E_NOMEM; dst[o++] = 0xFF; }                 /* EOF flag */
    else { if (o + 4 > cap) return ALZ_E_NOMEM; wr32(dst + o, 0, 0); o += 4; }            /* EndMark */
    if (dst_len) *dst_len = o;
    return 0;
}

static uint32_t snappy_crc_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }   /* Snappy.cs:252 */

/* Snappy.Decompress  Formats/Common/Snappy.cs:39-69: chunks are decoded from where the previous one stopped; CRCs are skipped */
static int snappy_file_decompress(const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* dst_len, size_t* src_used, int32_t* status) {
    if (len < 10 || memcmp(src, SNAPPY_ID, 10)) return ALZ_E_FORMAT;
    size_t pos = 10, out = 0; int32_t st = ALZ_ST_OK;
    while (pos < len) {
        if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
        uint32_t type = src[pos], cl = (uint32_t)src[pos + 1] | ((uint32_t)src[pos + 2] << 8) | ((uint32_t)src[pos + 3] << 16); pos += 4;
        if (type == 0) {
            if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            pos += 4;
            alz_result r;
            run_stream(ALZ_FMT_SNAPPY_RAW, NULL, src + pos, (uint32_t)(len - pos), 0, 0, 0, dst + out, out < cap ? cap - out : 0, &r);
            out += r.dst_len; pos += r.src_used;
            if (r.status != ALZ_ST_OK) { st = r.status; break; }
        } else if (type == 1) {
            if (pos + 4 > len || cl < 4) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            pos += 4;
            uint32_t n = cl - 4; if (n > len - pos) n = (uint32_t)(len - pos);            /* SubStream.CopyTo copies what is there */
            if (out + n > cap) { st = ALZ_ST_OUTPUT_CAPACITY; break; }
            memcpy(dst + out, src + pos, n); out += n; pos += n;
        } else {
            if (type >= 0x02 && type <= 0x7F) return ALZ_E_FORMAT;                        /* reserved unskippable chunk */
            pos = (uint64_t)pos + cl > len ? len : pos + cl;
        }
    }
    if (dst_len) *dst_len = out;
    if (src_used) *src_used = pos;
    if (status) *status = st;
    return st == ALZ_ST_OK ? 0 : ALZ_E_STREAM;
}

/* Snappy.Compress  Formats/Common/Snappy.cs:71-107: 64 KiB chunks, match finder reset per chunk */
static int snappy_file_compress(const alz_settings* st, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* dst_len) {
    if (cap < 10) return ALZ_E_NOMEM;
    memcpy(dst, SNAPPY_ID, 10);
    size_t o = 10, pos = 0;
    uint8_t* tmp = (uint8_t*)malloc(0x10000 + 0x10000 / 6 + 64);
    while (pos < n) {
        size_t cs = n - pos < 0x10000 ? n - pos : 0x10000;
        int64_t c = oracle_encode_stream(ALZ_FMT_SNAPPY_RAW, NULL, st, src + pos, cs, tmp, 0x10000 + 0x10000 / 6 + 64, NULL);
        if (c < 0) { free(tmp); return ALZ_E_INVALID; }
        uint32_t crc = snappy_crc_mask(oracle_crc32c(src + pos, cs));
        int stored = (size_t)c >= cs;
        size_t body = stored ? cs : (size_t)c;
        if (o + 8 + body > cap) { free(tmp); return ALZ_E_NOMEM; }
        dst[o] = stored ? 1 : 0;
        dst[o + 1] = (uint8_t)(body + 4); dst[o + 2] = (uint8_t)((body + 4) >> 8); dst[o + 3] = (uint8_t)((body + 4) >> 16);
        wr32(dst + o + 4, crc, 0);
        memcpy(dst + o + 8, stored ? src + pos : tmp, body);
        o += 8 + body; pos += cs;
    }
    free(tmp);
    if (dst_len) *dst_len = o;
    return 0;
}


/* PRS.ValidateByteOrder  Sega/PRS.cs:171-218 */
static int prs_validate(const uint8_t* src, size_t len, int big) {
    cur_t c = { src, (uint32_t)len, 0, 0 }; flag_t flag = { &c, 0, 0, big, 1 };
    int i = 3; uint64_t buffer = 0;
    while (c.pos < c.len) {
        int bit = flag_readbit(&flag); if (c.eof) return 0;
        if (bit) { c.pos++; buffer++; continue; }
        uint32_t distance, length;
        int bit2 = flag_readbit(&flag); if (c.eof) return 0;
        if (bit2) {
            int x0 = cur_u8(&c), x1 = cur_u8(&c); if (c.eof) return 0;
            uint32_t v = big ? (uint32_t)((x0 << 8) | x1) : (uint32_t)((x1 << 8) | x0);
            if (v == 0) return 1;
            length = v & 7; distance = 0x2000 - (v >> 3);
            if (length == 0) { int e = cur_u8(&c); if (c.eof) return 0; length = (uint32_t)e + 1; } else length += 2;
        } else {
            int h = flag_readbit(&flag); if (c.eof) return 0;
            int l = flag_readbit(&flag); if (c.eof) return 0;
            length = (uint32_t)((h << 1) | l) + 2;
            int e = cur_u8(&c); if (c.eof) return 0;
            distance = 0x100 - (uint32_t)e;
        }
        if (distance > buffer) return 0;
        if (i == 0) return 1;
        i--; buffer += length;
    }
    return 0;
}
/* PRS.GetByteOrder  Sega/PRS.cs:161-169: 1 little, 2 big, 0 none */
static int prs_byte_order(const uint8_t* src, size_t len) {
    if (len == 0) return 0;
    uint8_t f = src[0];
    if (f > 12 && (f & 1) == 1 && prs_validate(src, len, 0)) return 1;
    if ((f & 128) == 128 && prs_validate(src, len, 1)) return 2;
    return 0;
}

int oracle_container_decompress(uint32_t container, const alz_container_options* opt, const uint8_t* src, size_t len,
                                uint8_t* dst, size_t dst_cap, size_t* dst_len, size_t* src_used, int32_t* status) {
    int big = opt ? (int)opt->big_endian : 1;
    const alz_lz_properties* lz = opt ? &opt->lz : NULL;
    alz_result r; memset(&r, 0, sizeof(r));
    uint32_t size = 0; size_t hdr = 0;
    switch (container) {
    case ALZ_C_LZSS:                                                                     /* LZSS.cs:53-69 */
        if (len < 16 || memcmp(src, "LZSS", 4)) return ALZ_E_FORMAT;
        size = be32(src + 4); hdr = 16;
        run_stream(ALZ_FMT_LZSS, lz, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZ10: case ALZ_C_LZ11: {                                                  /* LZ10.cs:60-64 */
        int h = nin_header(src, len, container == ALZ_C_LZ10 ? 0x10 : 0x11, &size);
        if (h < 0) return ALZ_E_FORMAT;
        hdr = (size_t)h;
        run_stream(container == ALZ_C_LZ10 ? ALZ_FMT_LZ10 : ALZ_FMT_LZ11, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_LZHUDSON:                                                                 /* LZHudson.cs:33-37 */
        if (len < 4) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = be32(src); hdr = 4;
        run_stream(ALZ_FMT_LZHUDSON, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_SMSR00: {                                                                 /* SMSR00.cs:41-48 */
        if (len < 6 || memcmp(src, "SMSR00", 6)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = be32(src + 8); hdr = 16;
        uint32_t up = be32(src + 12);                                                    /* uncompressedDataPointer - source.Position */
        run_stream(ALZ_FMT_SMSR00, NULL, src + hdr, (uint32_t)(len - hdr), size, up - 16, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_HIG: {                                                                    /* Specialized/HIG.cs:47-80 */
        if (len < 4 || memcmp(src, "HIG!", 4)) return ALZ_E_FORMAT;
        if (len < 0x40) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        uint32_t start = rd32le(src + 4), ver = rd32le(src + 0x38);
        size = rd32le(src + 0x3C);
        if (ver == 5 || ver == 6) start = 0xC0;                                          /* extension header 0x40-0xC0: compressed size + path */
        if (start > len) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        hdr = start;
        run_stream(ALZ_FMT_HIG, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_LZSHREK: {                                                                /* Activision/LZShrek.cs:35-53 */
        if (len < 12) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        uint32_t offset = rd32le(src), csz = rd32le(src + 8);
        size = rd32le(src + 4);
        if (offset > len || csz > len - offset) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }   /* Seek + ReadExactly */
        run_stream(ALZ_FMT_LZSHREK, NULL, src + offset, csz, size, 0, 0, dst, dst_cap, &r);
        r.src_used = csz; hdr = offset;
        break;
    }
    case ALZ_C_WFLZ: {                                                                   /* WayForward/WFLZ.cs:50-86 */
        if (len < 4 || memcmp(src, "WFLZ", 4)) return ALZ_E_FORMAT;
        if (len < 12) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        const int wbig = opt && opt->big_endian;
        uint32_t csz = wbig ? be32(src + 4) : rd32le(src + 4);
        size = wbig ? be32(src + 8) : rd32le(src + 8); hdr = 12;
        if (csz > len - hdr) csz = (uint32_t)(len - hdr);                                /* Stream.Read returns what is there; the body then runs off its end */
        run_stream(wbig ? ALZ_FMT_WFLZ_BE : ALZ_FMT_WFLZ, NULL, src + hdr, csz, 0, 0, 0, dst, dst_cap, &r);
        if (r.status == ALZ_ST_OK && r.dst_len != size) r.status = ALZ_ST_OUTPUT_SIZE_MISMATCH;   /* '!='  :82-85 */
        r.src_used = csz;                                                                /* the whole compressed span is read up front */
        break;
    }
    case ALZ_C_REFPACK: {                                                                /* EA/RefPack.cs:64-75 */
        int h = refpack_header(src, len, &size);
        if (h < 0) return h;
        hdr = (size_t)h;
        if (len < hdr) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        run_stream(ALZ_FMT_REFPACK, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_LZ02:                                                                     /* Camelot/LZ02.cs:60-64 */
        if (len < 1 || (src[0] != 1 && src[0] != 2)) return ALZ_E_FORMAT;
        if (len < 4) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = ((uint32_t)src[1] << 16) | ((uint32_t)src[2] << 8) | src[3]; hdr = 4;
        run_stream(ALZ_FMT_LZ02, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_CNS:                                                                      /* Specialized/CNS.cs:44-55 */
        if (len < 4 || memcmp(src, "@CNS", 4)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = rd32le(src + 8); hdr = 16;
        run_stream(ALZ_FMT_CNS, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_CLZ0:                                                                     /* Marvelous/CLZ0.cs:41-51 */
        if (len < 4 || memcmp(src, "CLZ\0", 4)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = be32(src + 12); hdr = 16;
        run_stream(ALZ_FMT_CLZ0, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_BLZ: {                                                                    /* Nintendo/BLZ.cs:43-69 */
        if (len < 8) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        uint32_t csz = (uint32_t)src[len - 8] | ((uint32_t)src[len - 7] << 8) | ((uint32_t)src[len - 6] << 16);
        uint32_t hp = src[len - 5];
        if (hp < 8) return ALZ_E_FORMAT;                                                 /* "Invalid BLZ header." */
        size = rd32le(src + len - 4) + csz;
        if (csz > len || hp > csz) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }         /* Position before the stream start / negative codeSize */
        uint32_t code = csz - hp;
        if ((uint64_t)size > dst_cap) { r.status = ALZ_ST_OUTPUT_CAPACITY; break; }
        uint8_t* rev = (uint8_t*)malloc(code + 1); uint8_t* tmp = (uint8_t*)malloc((size_t)size + 1);
        if (!rev || !tmp) { free(rev); free(tmp); return ALZ_E_NOMEM; }
        const uint8_t* cs = src + (len - csz);
        for (uint32_t i = 0; i < code; i++) rev[i] = cs[code - 1 - i];
        run_stream(ALZ_FMT_BLZ, NULL, rev, code, size, 0, 0, tmp, size, &r);
        if (r.status == ALZ_ST_OK) for (uint32_t i = 0; i < size; i++) dst[i] = tmp[size - 1 - i];
        else r.dst_len = 0;                                                              /* the managed code writes nothing when the body throws (:62-64) */
        free(rev); free(tmp);
        r.src_used = (uint32_t)len; hdr = 0;
        break;
    }
    case ALZ_C_CNX2:                                                                     /* Sega/CNX2.cs:45-62 */
        if (len < 4 || memcmp(src, "CNX\x02", 4)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = be32(src + 12); hdr = 16;
        run_stream(ALZ_FMT_CNX2, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_FASTLZ:                                                                   /* FastLZ.cs:40-52: the whole rest of the stream is the body */
        run_stream(ALZ_FMT_FASTLZ, NULL, src, (uint32_t)len, 0, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZ00: {                                                                   /* Sega/LZ00.cs:40-60 */
        if (len < 4 || memcmp(src, "LZ00", 4)) return ALZ_E_FORMAT;
        if (len < 64) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = rd32le(src + 48); hdr = 64;
        uint8_t* plain = (uint8_t*)malloc(len - hdr + 1);
        if (!plain) return ALZ_E_NOMEM;
        memcpy(plain, src + hdr, len - hdr);
        lz00_keystream(plain, len - hdr, rd32le(src + 52));
        run_stream(ALZ_FMT_LZSS, NULL, plain, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        free(plain);
        break;
    }
    case ALZ_C_LZ40: case ALZ_C_LZ60: {                                                  /* LZ40.cs:54-61, LZ60.cs:43-47 */
        int h = nin_header(src, len, container == ALZ_C_LZ40 ? 0x40 : 0x60, &size);
        if (h < 0) return ALZ_E_FORMAT;
        hdr = (size_t)h;
        run_stream(ALZ_FMT_LZ40, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_YAZ0:                                                                     /* Yaz0.cs:58-79 */
        if (len < 16 || memcmp(src, "Yaz0", 4)) return ALZ_E_FORMAT;
        size = rd32(src + 4, big); hdr = 16;
        run_stream(ALZ_FMT_YAZ0, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        if (r.status != ALZ_ST_OK)                                                       /* catch (Exception): try other order */
            run_stream(ALZ_FMT_YAZ0, NULL, src + hdr, (uint32_t)(len - hdr), bswap32(size), 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_YAY0: case ALZ_C_MIO0: {                                                  /* Yay0.cs:50-60 / MIO0.cs:51-61 */
        if (len < 16 || memcmp(src, container == ALZ_C_YAY0 ? "Yay0" : "MIO0", 4)) return ALZ_E_FORMAT;
        /* DetectByteOrder<uint>(3) lives in the unvendored AuroraLib.Core: parity unpinned; the caller's FormatByteOrder is used */
        size = rd32(src + 4, big); uint32_t cp = rd32(src + 8, big), up = rd32(src + 12, big); hdr = 16;
        run_stream(container == ALZ_C_YAY0 ? ALZ_FMT_YAY0 : ALZ_FMT_MIO0, NULL, src + hdr, (uint32_t)(len - hdr), size, cp - 0x10, up - 0x10, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_PRS: {                                                                    /* PRS.cs:42-57: detected order first, then the other */
        int first_big = prs_byte_order(src, len) == 2;   /* GetByteOrder(source) == Endian.Big ? Big : Little */
        (void)big;
        run_stream(first_big ? ALZ_FMT_PRS_BE : ALZ_FMT_PRS_LE, NULL, src, (uint32_t)len, 0, 0, 0, dst, dst_cap, &r);
        if (r.status != ALZ_ST_OK) run_stream(first_big ? ALZ_FMT_PRS_LE : ALZ_FMT_PRS_BE, NULL, src, (uint32_t)len, 0, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_LZO: run_stream(ALZ_FMT_LZO, NULL, src, (uint32_t)len, 0, 0, 0, dst, dst_cap, &r); break;
    case ALZ_C_LZ4_LEGACY: case ALZ_C_LZ4_FRAME:                                          /* LZ4Legacy.Decompress -> LZ4.Decompress */
        return lz4_file_decompress(src, len, dst, dst_cap, dst_len, src_used, status);
    case ALZ_C_MDB4:                                                                     /* Specialized/MDB4.cs:33-50 */
        if (len < 4 || memcmp(src, "MDB4", 4)) return ALZ_E_FORMAT;
        if (len < 32) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = rd32le(src + 8); hdr = 32;
        run_stream(ALZ_FMT_LZSS, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_FCMP: case ALZ_C_IECP: case ALZ_C_GCZ: {                                  /* FCMP.cs:36-41, IECP.cs:35-39, GCZ.cs:32-36 */
        size_t ml = container == ALZ_C_GCZ ? 0 : 4;
        hdr = ml + (container == ALZ_C_FCMP ? 8 : 4);
        if (len < ml || (ml && memcmp(src, container == ALZ_C_FCMP ? "FCMP" : "IECP", 4))) return ALZ_E_FORMAT;
        if (len < hdr) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = rd32le(src + ml);
        run_stream(ALZ_FMT_LZSS, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_SDPC:                                                                     /* Specialized/SDPC.cs:34-47 */
        if (len < 4 || memcmp(src, "SDPC", 4)) return ALZ_E_FORMAT;
        if (len < 8) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = rd32le(src + 4); hdr = 8;
        run_stream(ALZ_FMT_LZO, NULL, src + hdr, (uint32_t)(len - hdr), 0, 0, 0, dst, dst_cap, &r);
        if (r.status == ALZ_ST_OK && r.dst_len > size) r.status = ALZ_ST_OUTPUT_SIZE_MISMATCH;
        break;
    case ALZ_C_ECD: {                                                                    /* Specialized/ECD.cs:45-71 */
        if (len < 3 || memcmp(src, "ECD", 3)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        int compressed = src[3] == 1;
        uint32_t plain = be32(src + 4); size = be32(src + 12); hdr = 16;
        if (!compressed) {
            size_t n = len - hdr;
            if (n > dst_cap) { memcpy(dst, src + hdr, dst_cap); r.dst_len = (uint32_t)dst_cap; r.status = ALZ_ST_OUTPUT_CAPACITY; break; }
            memcpy(dst, src + hdr, n); r.dst_len = (uint32_t)n; r.src_used = (uint32_t)n; r.status = ALZ_ST_OK;
            break;
        }
        if ((uint64_t)plain > len - hdr) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        if (plain > dst_cap) { r.status = ALZ_ST_OUTPUT_CAPACITY; break; }
        memcpy(dst, src + hdr, plain);
        alz_lz_properties e; oracle_lz_properties_bits(10, 6, 2, &e);                    /* LzProperties(0x400, 0x42, 3, 0x3BE)  ECD.cs:15 */
        run_stream(ALZ_FMT_LZSS, &e, src + hdr + plain, (uint32_t)(len - hdr - plain), size - plain, 0, 0, dst + plain, dst_cap - plain, &r);
        r.dst_len += plain; r.src_used += plain;
        break;
    }
    case ALZ_C_SNAPPY: return snappy_file_decompress(src, len, dst, dst_cap, dst_len, src_used, status);
    case ALZ_C_GCLZ: case ALZ_C_CXLZ: case ALZ_C_LZ_3DS: case ALZ_C_COMP: {                /* magic + inner file */
        const char* m = container == ALZ_C_GCLZ ? "GCLZ" : container == ALZ_C_CXLZ ? "CXLZ" : container == ALZ_C_COMP ? "COMP" : "3DS-LZ\r\n";
        size_t ml = container == ALZ_C_LZ_3DS ? 8 : 4, used = 0;
        if (len < ml || memcmp(src, m, ml)) return ALZ_E_FORMAT;
        int rc = oracle_container_decompress(container == ALZ_C_COMP ? ALZ_C_LZ11 : ALZ_C_LZ10, opt, src + ml, len - ml, dst, dst_cap, dst_len, &used, status);
        if (src_used) *src_used = ml + used;
        return rc;
    }
    case ALZ_C_YAZ1:
        if (len < 16 || memcmp(src, "Yaz1", 4)) return ALZ_E_FORMAT;
        size = rd32(src + 4, big); hdr = 16;
        run_stream(ALZ_FMT_YAZ0, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        if (r.status != ALZ_ST_OK) run_stream(ALZ_FMT_YAZ0, NULL, src + hdr, (uint32_t)(len - hdr), bswap32(size), 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_AKLZ:                                                                       /* Sega/AKLZ.cs:41-46 */
        if (len < 16 || memcmp(src, AKLZ_MAGIC, 12)) return ALZ_E_FORMAT;
        size = be32(src + 12); hdr = 16;
        run_stream(ALZ_FMT_LZSS, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZ01:                                                                       /* Sega/LZ01.cs:47-62 */
        if (len < 16 || memcmp(src, "LZ01", 4)) return ALZ_E_FORMAT;
        size = rd32le(src + 8); hdr = 16;
        run_stream(ALZ_FMT_LZSS, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZSEGA:                                                                     /* Sega/LZSega.cs:49-54 */
        if (len < 8) return ALZ_E_FORMAT;
        size = rd32le(src + 4); hdr = 8;
        run_stream(ALZ_FMT_LZSS, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LEVEL5LZSS:                                                                 /* Level5/Level5LZSS.cs:42-59 */
        if (len < 16 || memcmp(src, "SSZL", 4)) return ALZ_E_FORMAT;
        size = rd32le(src + 12); hdr = 16;
        run_stream(ALZ_FMT_LZSS, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZON:                                                                       /* Nintendo/LZOn.cs:41-60 */
        if (len < 16 || memcmp(src, LZON_MAGIC, 8)) return ALZ_E_FORMAT;
        size = be32(src + 8); hdr = 16;
        run_stream(ALZ_FMT_LZO, NULL, src + hdr, (uint32_t)(len - hdr), 0, 0, 0, dst, dst_cap, &r);
        if (r.status == ALZ_ST_OK && r.dst_len != size) r.status = ALZ_ST_OUTPUT_SIZE_MISMATCH;
        break;
    case ALZ_C_LEVEL5: {                                                                   /* Level5/Level5.cs:62-110 */
        if (len < 4) return ALZ_E_FORMAT;
        uint32_t ts = rd32le(src); hdr = 4;
        if (len > 4 && src[4] == 0x78) return ALZ_E_UNSUPPORTED;
        size = ts >> 3;
        if ((ts & 7) == ALZ_LEVEL5_ONLYSAVE) {
            if (len - hdr < size) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
            if (dst_cap < size) { r.status = ALZ_ST_OUTPUT_CAPACITY; break; }
            memcpy(dst, src + hdr, size); r.dst_len = size; r.src_used = size; r.status = ALZ_ST_OK;
        } else if ((ts & 7) == ALZ_LEVEL5_LZ10) run_stream(ALZ_FMT_LZ10, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
        else return ALZ_E_UNSUPPORTED;
        break;
    }
    case ALZ_C_LZ77: {                                                                     /* Nintendo/LZ77.cs:105-153 */
        if (len < 8 || memcmp(src, "LZ77", 4)) return ALZ_E_FORMAT;
        uint32_t type = src[4];
        size = (uint32_t)src[5] | ((uint32_t)src[6] << 8) | ((uint32_t)src[7] << 16); hdr = 8;
        if (size == 0) { if (len < 12) return ALZ_E_FORMAT; size = rd32le(src + 8); hdr = 12; }
        if (type == ALZ_LZ77_LZ10 || type == ALZ_LZ77_LZ11) {
            run_stream(type == ALZ_LZ77_LZ10 ? ALZ_FMT_LZ10 : ALZ_FMT_LZ11, NULL, src + hdr, (uint32_t)(len - hdr), size, 0, 0, dst, dst_cap, &r);
            break;
        }
        if (type != ALZ_LZ77_CHUNKLZ10) return ALZ_E_UNSUPPORTED;
        /* ChunkLZ10: u16 end offsets until last + position == length; one LZ10 file per chunk, decoded in order */
        size_t pos = hdr, nseg = 0, ends_cap = 65536; uint32_t* ends = (uint32_t*)malloc(ends_cap * sizeof(uint32_t));
        int trunc = 0;
        for (;;) {
            if (pos + 2 > len || nseg >= ends_cap) { trunc = 1; break; }
            ends[nseg++] = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8); pos += 2;
            if (ends[nseg - 1] + pos == len) break;
        }
        if (trunc) { free(ends); r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size_t header_end = pos; uint64_t out_off = 0;
        r.status = ALZ_ST_OK; r.dst_len = 0;
        for (size_t i = 0; i < nseg; i++) {
            size_t a = header_end + (i ? ends[i - 1] : 0); uint32_t csz = 0;
            int h = a < len ? nin_header(src + a, len - a, 0x10, &csz) : -1;
            if (h < 0) { free(ends); return ALZ_E_FORMAT; }
            alz_result cr;
            run_stream(ALZ_FMT_LZ10, NULL, src + a + h, (uint32_t)(len - a - h), csz, 0, 0, dst + out_off, out_off < dst_cap ? dst_cap - out_off : 0, &cr);
            r.dst_len = (uint32_t)(out_off + cr.dst_len);
            if (cr.status != ALZ_ST_OK) { r.status = cr.status; break; }
            out_off += csz;
        }
        if (r.status == ALZ_ST_OK && r.dst_len > size) r.status = ALZ_ST_OUTPUT_SIZE_MISMATCH;
        r.src_used = (uint32_t)(header_end + ends[nseg - 1] - hdr);
        free(ends);
        break;
    }
    default: return ALZ_E_UNSUPPORTED;
    }
    if (dst_len) *dst_len = r.dst_len;
    if (src_used) *src_used = hdr + r.src_used;
    if (status) *status = r.status;
    return r.status == ALZ_ST_OK ? 0 : ALZ_E_STREAM;
}

int oracle_container_compress(uint32_t container, const alz_container_options* opt, const alz_settings* settings,
                              const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* dst_len) {
    int big = opt ? (int)opt->big_endian : 1;
    const alz_lz_properties* lz = opt ? &opt->lz : NULL;
    size_t hdr = 0; int64_t body; alz_encode_aux aux;
    alz_settings st = settings ? *settings : (alz_settings){ 8, 0, 0, 0 };
    switch (container) {
    case ALZ_C_ECD: {                                                                      /* Specialized/ECD.cs:73-109 */
        int compressed = st.quality != 0 && n > 0x10;
        uint32_t plain = compressed ? 4u : 0u;
        if (cap < 16 + (compressed ? plain : n)) return ALZ_E_NOMEM;
        memcpy(dst, "ECD", 3);
        if (compressed) {
            alz_lz_properties e; oracle_lz_properties_bits(10, 6, 2, &e);
            int64_t b = oracle_encode_stream(ALZ_FMT_LZSS, &e, &st, src + plain, n - plain, dst + 16 + plain, cap - 16 - plain, NULL);
            if (b < -1) return ALZ_E_INVALID;
            if (b >= 0 && (uint64_t)plain + (uint64_t)b < n) {
                dst[3] = 1; wr32(dst + 4, plain, 1); wr32(dst + 8, plain + (uint32_t)b, 1); wr32(dst + 12, (uint32_t)n, 1);
                memcpy(dst + 16, src, plain);
                if (dst_len) *dst_len = 16 + plain + (size_t)b;
                return 0;
            }
            if (cap < 16 + n) return ALZ_E_NOMEM;
        }
        dst[3] = 0; wr32(dst + 4, 0, 1); wr32(dst + 8, (uint32_t)n, 1); wr32(dst + 12, (uint32_t)n, 1);
        memcpy(dst + 16, src, n);
        if (dst_len) *dst_len = 16 + n;
        return 0;
    }
    case ALZ_C_MDB4: case ALZ_C_FCMP: case ALZ_C_IECP: case ALZ_C_GCZ: case ALZ_C_SDPC: {
        hdr = container == ALZ_C_MDB4 ? 32 : container == ALZ_C_FCMP ? 12 : container == ALZ_C_GCZ ? 4 : 8;
        if (cap < hdr) return ALZ_E_NOMEM;
        body = oracle_encode_stream(container == ALZ_C_SDPC ? ALZ_FMT_LZO : ALZ_FMT_LZSS, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return body == -1 ? ALZ_E_NOMEM : ALZ_E_INVALID;
        switch (container) {
        case ALZ_C_MDB4: memcpy(dst, "MDB4", 4); wr32(dst + 4, (uint32_t)n + 1, 0); wr32(dst + 8, (uint32_t)n, 0); wr32(dst + 12, 16 + (uint32_t)body, 0); memset(dst + 16, 0, 16); break;
        case ALZ_C_FCMP: memcpy(dst, "FCMP", 4); wr32(dst + 4, (uint32_t)n, 0); wr32(dst + 8, 305397760u, 0); break;
        case ALZ_C_IECP: memcpy(dst, "IECP", 4); wr32(dst + 4, (uint32_t)n, 0); break;
        case ALZ_C_GCZ: wr32(dst, (uint32_t)n, 0); break;
        default: memcpy(dst, "SDPC", 4); wr32(dst + 4, (uint32_t)n, 0); break;
        }
        if (dst_len) *dst_len = hdr + (size_t)body;
        return 0;
    }
    case ALZ_C_LZ4_LEGACY: return lz4_file_compress(1, 0, &st, src, n, dst, cap, dst_len);
    case ALZ_C_LZ4_FRAME: return lz4_file_compress(0, opt ? opt->chunk_size : 0, &st, src, n, dst, cap, dst_len);
    case ALZ_C_SNAPPY: return snappy_file_compress(&st, src, n, dst, cap, dst_len);
    case ALZ_C_GCLZ: case ALZ_C_CXLZ: case ALZ_C_LZ_3DS: case ALZ_C_COMP: {                /* e.g. Nintendo/GCLZ.cs:40-44 */
        const char* m = container == ALZ_C_GCLZ ? "GCLZ" : container == ALZ_C_CXLZ ? "CXLZ" : container == ALZ_C_COMP ? "COMP" : "3DS-LZ\r\n";
        size_t ml = container == ALZ_C_LZ_3DS ? 8 : 4, inner = 0;
        if (cap < ml) return ALZ_E_NOMEM;
        memcpy(dst, m, ml);
        int rc = oracle_container_compress(container == ALZ_C_COMP ? ALZ_C_LZ11 : ALZ_C_LZ10, opt, settings, src, n, dst + ml, cap - ml, &inner);
        if (dst_len) *dst_len = ml + inner;
        return rc;
    }
    case ALZ_C_YAZ1:
        if (cap < 16) return ALZ_E_NOMEM;
        memcpy(dst, "Yaz1", 4); wr32(dst + 4, (uint32_t)n, big); wr32(dst + 8, opt ? opt->memory_alignment : 0, big); wr32(dst + 12, 0, 0); hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_YAZ0, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        break;
    case ALZ_C_AKLZ:                                                                       /* Sega/AKLZ.cs:50-55 */
        if (cap < 16) return ALZ_E_NOMEM;
        memcpy(dst, AKLZ_MAGIC, 12); wr32(dst + 12, (uint32_t)n, 1); hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_LZSS, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        break;
    case ALZ_C_LZ01:                                                                       /* Sega/LZ01.cs:65-82 */
        if (cap < 16) return ALZ_E_NOMEM;
        hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_LZSS, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        memcpy(dst, "LZ01", 4); wr32(dst + 4, (uint32_t)(hdr + body), 0); wr32(dst + 8, (uint32_t)n, 0); wr32(dst + 12, 0, 0);
        break;
    case ALZ_C_LZSEGA:                                                                     /* Sega/LZSega.cs:57-67 */
        if (cap < 8) return ALZ_E_NOMEM;
        hdr = 8;
        body = oracle_encode_stream(ALZ_FMT_LZSS, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        wr32(dst, (uint32_t)body, 0); wr32(dst + 4, (uint32_t)n, 0);
        break;
    case ALZ_C_LEVEL5LZSS:                                                                 /* Level5/Level5LZSS.cs:62-72 */
        if (cap < 16) return ALZ_E_NOMEM;
        hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_LZSS, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        memcpy(dst, "SSZL", 4); wr32(dst + 4, 0, 0); wr32(dst + 8, (uint32_t)body, 0); wr32(dst + 12, (uint32_t)n, 0);
        break;
    case ALZ_C_LZON:                                                                       /* Nintendo/LZOn.cs:63-79 */
        if (cap < 16) return ALZ_E_NOMEM;
        hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_LZO, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return body == -1 ? ALZ_E_NOMEM : ALZ_E_INVALID;
        memcpy(dst, LZON_MAGIC, 8); wr32(dst + 8, (uint32_t)n, 1); wr32(dst + 12, (uint32_t)body, 1);
        break;
    case ALZ_C_LEVEL5: {                                                                   /* Level5/Level5.cs:112-146 */
        uint32_t type = opt && opt->variant ? opt->variant : ALZ_LEVEL5_LZ10;
        if (st.quality == 0) type = ALZ_LEVEL5_ONLYSAVE;
        if (cap < 4) return ALZ_E_NOMEM;
        wr32(dst, type | ((uint32_t)n << 3), 0); hdr = 4;
        if (type == ALZ_LEVEL5_ONLYSAVE) { if (cap < 4 + n) return ALZ_E_NOMEM; memcpy(dst + 4, src, n); body = (int64_t)n; break; }
        if (type != ALZ_LEVEL5_LZ10) return ALZ_E_UNSUPPORTED;
        if (st.min_distance == 0) st.min_distance = 2;
        body = oracle_encode_stream(ALZ_FMT_LZ10, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        break;
    }
    case ALZ_C_LZ77: {                                                                     /* Nintendo/LZ77.cs:56-102 */
        uint32_t type = opt && opt->variant ? opt->variant : ALZ_LZ77_LZ10;
        size_t chunk = opt && opt->chunk_size ? opt->chunk_size : 0x1000;
        if (cap < 8) return ALZ_E_NOMEM;
        memcpy(dst, "LZ77", 4);
        if (type == ALZ_LZ77_LZ10 || type == ALZ_LZ77_LZ11 || (type == ALZ_LZ77_CHUNKLZ10 && chunk >= n)) {
            size_t inner = 0;
            int rc = oracle_container_compress(type == ALZ_LZ77_LZ11 ? ALZ_C_LZ11 : ALZ_C_LZ10, opt, settings, src, n, dst + 4, cap - 4, &inner);
            if (dst_len) *dst_len = 4 + inner;
            return rc;
        }
        if (type != ALZ_LZ77_CHUNKLZ10) return ALZ_E_UNSUPPORTED;
        if (n > 0xFFFFFF) return ALZ_E_INVALID;
        size_t segs = (n + chunk - 1) / chunk, header_end = 8 + 2 * segs, pos = header_end;
        if (cap < header_end) return ALZ_E_NOMEM;
        wr32(dst + 4, type | ((uint32_t)n << 8), 0);
        for (size_t i = 0; i < segs; i++) {
            size_t a = i * chunk, sz = n - a < chunk ? n - a : chunk, inner = 0;
            int rc = oracle_container_compress(ALZ_C_LZ10, opt, settings, src + a, sz, dst + pos, cap - pos, &inner);
            if (rc) return rc;
            pos += inner;
            size_t endoff = pos - header_end;
            if (endoff > 0xFFFF) return ALZ_E_INVALID;
            dst[8 + 2 * i] = (uint8_t)endoff; dst[8 + 2 * i + 1] = (uint8_t)(endoff >> 8);
        }
        if (dst_len) *dst_len = pos;
        return 0;
    }
    case ALZ_C_LZSS:                                                                     /* LZSS.cs:72-88 */
        if (cap < 16) return ALZ_E_NOMEM;
        memcpy(dst, "LZSS", 4); wr32(dst + 4, (uint32_t)n, 1); wr32(dst + 8, 0, 1); wr32(dst + 12, 0, 1); hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_LZSS, lz, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        wr32(dst + 8, (uint32_t)body, 1);
        break;
    case ALZ_C_LZ10: case ALZ_C_LZ11: {                                                  /* LZ10.cs:67-80 */
        uint8_t id = container == ALZ_C_LZ10 ? 0x10 : 0x11;
        if (cap < 8) return ALZ_E_NOMEM;
        if (n <= 0xFFFFFF) { wr32(dst, id | ((uint32_t)n << 8), 0); hdr = 4; }
        else { wr32(dst, id, 0); wr32(dst + 4, (uint32_t)n, 0); hdr = 8; }
        if (container == ALZ_C_LZ10 && st.min_distance == 0) st.min_distance = 2;        /* GbaVramCompatibilityMode default true LZ10.cs:33 */
        body = oracle_encode_stream(container == ALZ_C_LZ10 ? ALZ_FMT_LZ10 : ALZ_FMT_LZ11, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        break;
    }
    case ALZ_C_LZHUDSON:                                                                 /* LZHudson.cs:39-43 */
        if (cap < 4) return ALZ_E_NOMEM;
        wr32(dst, (uint32_t)n, 1); hdr = 4;
        body = oracle_encode_stream(ALZ_FMT_LZHUDSON, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        break;
    case ALZ_C_SMSR00:                                                                   /* SMSR00.cs:50-66 */
        if (cap < 16) return ALZ_E_NOMEM;
        hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_SMSR00, NULL, &st, src, n, dst + hdr, cap - hdr, &aux);
        if (body < 0) return ALZ_E_NOMEM;
        memcpy(dst, "SMSR00", 6); dst[6] = 0; dst[7] = 0; wr32(dst + 8, (uint32_t)n, 1); wr32(dst + 12, 16 + aux.aux0, 1);
        break;
    case ALZ_C_HIG:                                                                      /* Specialized/HIG.cs:82-124: Version 6, the default path string */
        if (cap < 0xC0) return ALZ_E_NOMEM;
        hdr = 0xC0;
        body = oracle_encode_stream(ALZ_FMT_HIG, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        memset(dst, 0, 0xC0); memcpy(dst, "HIG!", 4); wr32(dst + 0x38, 6, 0); wr32(dst + 0x3C, (uint32_t)n, 0);
        wr32(dst + 0x40, (uint32_t)body, 0); memcpy(dst + 0x44, "C:\\HIG\\PROJECTS\\test.mb.wad.conf", 32);
        break;
    case ALZ_C_LZSHREK:                                                                  /* Activision/LZShrek.cs:60-71 */
        if (cap < 16) return ALZ_E_NOMEM;
        hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_LZSHREK, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        wr32(dst, 0x10, 0); wr32(dst + 4, (uint32_t)n, 0); wr32(dst + 8, (uint32_t)body, 0); wr32(dst + 12, 0, 0);
        break;
    case ALZ_C_WFLZ: {                                                                   /* WayForward/WFLZ.cs:89-105 */
        const int wbig = opt && opt->big_endian;
        if (cap < 12) return ALZ_E_NOMEM;
        hdr = 12;
        body = oracle_encode_stream(wbig ? ALZ_FMT_WFLZ_BE : ALZ_FMT_WFLZ, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        memcpy(dst, "WFLZ", 4); wr32(dst + 4, (uint32_t)body, wbig); wr32(dst + 8, (uint32_t)n, wbig);
        break;
    }
    case ALZ_C_REFPACK:                                                                  /* EA/RefPack.cs:105-125: Options = Default | UsePreHeader -> version 2 */
        if (n >= 0xFFFFFF) return ALZ_E_UNSUPPORTED;                                     /* "RefPack Version 2 does not support files over 16MB." */
        if (cap < 9) return ALZ_E_NOMEM;
        hdr = 9;
        body = oracle_encode_stream(ALZ_FMT_REFPACK, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        wr32(dst, (uint32_t)(hdr + body - 4), 0); dst[4] = 0x10; dst[5] = 0xFB; dst[6] = (uint8_t)(n >> 16); dst[7] = (uint8_t)(n >> 8); dst[8] = (uint8_t)n;
        break;
    case ALZ_C_LZ02:                                                                     /* Camelot/LZ02.cs:66-75 (no extension data: DataType.Default) */
        if (cap < 4) return ALZ_E_NOMEM;
        hdr = 4;
        body = oracle_encode_stream(ALZ_FMT_LZ02, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        dst[0] = 1; dst[1] = (uint8_t)(n >> 16); dst[2] = (uint8_t)(n >> 8); dst[3] = (uint8_t)n;
        break;
    case ALZ_C_CNS:                                                                      /* Specialized/CNS.cs:57-75 */
        if (n < 4) return ALZ_E_INVALID;                                                 /* source[3]: IndexOutOfRangeException */
        if (cap < 16) return ALZ_E_NOMEM;
        hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_CNS, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        memcpy(dst, "@CNS", 4);
        memcpy(dst + 4, (src[0] == 0x00 && src[1] == 0x20 && src[2] == 0xAF && src[3] == 0x30) ? "TPL\0" : "PAK\0", 4);
        wr32(dst + 8, (uint32_t)n, 0); wr32(dst + 12, 0, 0);
        break;
    case ALZ_C_CLZ0:                                                                     /* Marvelous/CLZ0.cs:53-62 */
        if (cap < 16) return ALZ_E_NOMEM;
        hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_CLZ0, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        memcpy(dst, "CLZ\0", 4); wr32(dst + 4, (uint32_t)n, 1); wr32(dst + 8, 0, 1); wr32(dst + 12, (uint32_t)n, 1);
        break;
    case ALZ_C_BLZ: {                                                                    /* Nintendo/BLZ.cs:71-95 */
        uint8_t* rev = (uint8_t*)malloc(n + 1); uint8_t* tmp = (uint8_t*)malloc(n + n / 4 + 64);
        if (!rev || !tmp) { free(rev); free(tmp); return ALZ_E_NOMEM; }
        for (size_t i = 0; i < n; i++) rev[i] = src[n - 1 - i];
        body = oracle_encode_stream(ALZ_FMT_BLZ, NULL, &st, rev, n, tmp, n + n / 4 + 64, NULL);
        free(rev);
        if (body < 0) { free(tmp); return ALZ_E_NOMEM; }
        uint32_t total = (uint32_t)body + 8; uint32_t pad = (16 - (total % 16)) % 16; total += pad;
        if (cap < total) { free(tmp); return ALZ_E_NOMEM; }
        for (int64_t i = 0; i < body; i++) dst[i] = tmp[body - 1 - i];                   /* the buffer was filled from its end: stored back to front */
        free(tmp);
        memset(dst + body, 0xFF, pad);
        uint8_t* f = dst + body + pad;
        f[0] = (uint8_t)total; f[1] = (uint8_t)(total >> 8); f[2] = (uint8_t)(total >> 16); f[3] = (uint8_t)(8 + pad);
        wr32(f + 4, (uint32_t)((int64_t)n - (int64_t)total), 0);
        hdr = 0; body = total;
        break;
    }
    case ALZ_C_CNX2:                                                                     /* Sega/CNX2.cs:64-81: Extension "DEC" padded with 0x10 */
        if (cap < 16) return ALZ_E_NOMEM;
        hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_CNX2, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        memcpy(dst, "CNX\x02" "DEC\x10", 8); wr32(dst + 8, (uint32_t)body, 1); wr32(dst + 12, (uint32_t)n, 1);
        break;
    case ALZ_C_FASTLZ: {                                                                 /* FastLZ.cs:162-163 */
        hdr = 0;
        body = oracle_encode_stream(ALZ_FMT_FASTLZ, NULL, &st, src, n, dst, cap, NULL);
        if (body == ALZ_E_UNSUPPORTED) return ALZ_E_UNSUPPORTED;
        if (body < 0) return ALZ_E_NOMEM;
        break;
    }
    case ALZ_C_LZ00: {                                                                   /* Sega/LZ00.cs:71-96 */
        if (cap < 64) return ALZ_E_NOMEM;
        hdr = 64;
        body = oracle_encode_stream(ALZ_FMT_LZSS, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        const uint32_t key = opt ? opt->key : 0;
        lz00_keystream(dst + hdr, (size_t)body, key);
        memset(dst, 0, 64); memcpy(dst, "LZ00", 4); wr32(dst + 4, (uint32_t)(hdr + body), 0);
        int named = 0; if (opt) for (int i = 0; i < 32; i++) named |= opt->name[i];
        if (named) memcpy(dst + 16, opt->name, 32); else memcpy(dst + 16, "Temp.dat", 8);
        wr32(dst + 48, (uint32_t)n, 0); wr32(dst + 52, key, 0);
        break;
    }
    case ALZ_C_LZ40: case ALZ_C_LZ60: {                                                  /* LZ40.cs:64-77, LZ60.cs:49-61 */
        uint8_t id = container == ALZ_C_LZ40 ? 0x40 : 0x60;
        if (cap < 8) return ALZ_E_NOMEM;
        if (n <= 0xFFFFFF) { wr32(dst, id | ((uint32_t)n << 8), 0); hdr = 4; }
        else { wr32(dst, id, 0); wr32(dst + 4, (uint32_t)n, 0); hdr = 8; }
        body = oracle_encode_stream(ALZ_FMT_LZ40, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);   /* GbaVramCompatibilityMode = false */
        if (body < 0) return ALZ_E_NOMEM;
        break;
    }
    case ALZ_C_YAZ0:                                                                     /* Yaz0.cs:82-89 */
        if (cap < 16) return ALZ_E_NOMEM;
        memcpy(dst, "Yaz0", 4); wr32(dst + 4, (uint32_t)n, big); wr32(dst + 8, opt ? opt->memory_alignment : 0, big); wr32(dst + 12, 0, 0); hdr = 16;
        body = oracle_encode_stream(ALZ_FMT_YAZ0, NULL, &st, src, n, dst + hdr, cap - hdr, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        break;
    case ALZ_C_YAY0: case ALZ_C_MIO0:                                                    /* Yay0.cs:62-77 */
        if (cap < 16) return ALZ_E_NOMEM;
        memcpy(dst, container == ALZ_C_YAY0 ? "Yay0" : "MIO0", 4); hdr = 16;
        body = oracle_encode_stream(container == ALZ_C_YAY0 ? ALZ_FMT_YAY0 : ALZ_FMT_MIO0, NULL, &st, src, n, dst + hdr, cap - hdr, &aux);
        if (body < 0) return ALZ_E_NOMEM;
        wr32(dst + 4, (uint32_t)n, big); wr32(dst + 8, 0x10 + aux.aux0, big); wr32(dst + 12, 0x10 + aux.aux1, big);
        break;
    case ALZ_C_PRS:
        body = oracle_encode_stream(big ? ALZ_FMT_PRS_BE : ALZ_FMT_PRS_LE, NULL, &st, src, n, dst, cap, NULL);
        if (body < 0) return ALZ_E_NOMEM;
        break;
    case ALZ_C_LZO:
        body = oracle_encode_stream(ALZ_FMT_LZO, NULL, &st, src, n, dst, cap, NULL);
        if (body < 0) return body == -1 ? ALZ_E_NOMEM : ALZ_E_INVALID;
        break;
    default: return ALZ_E_UNSUPPORTED;
    }
    if (dst_len) *dst_len = hdr + (size_t)body;
    return 0;
}
