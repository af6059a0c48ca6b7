/*
 * alz_synth.c -- seeded token-level generators of synthetic compressed streams
 * (SURVEY.md section 8d "Concrete synthetic inputs").  Host-side C, no GPU.
 *
 * Streams are generated directly in compressed form: the generator only tracks
 * how many bytes the stream will decode to, so it runs in O(compressed size).
 * PRNG: splitmix64, seed = base_seed + stream_index.
 *
 * Token mix for the flag-byte formats (LZSS/LZ10/LZ11/Yaz0/Yay0/MIO0/PRS):
 *   literal w.p. 0.5 (uniform byte); match length uniform over [min, min(max,18)]
 *   w.p. 0.9 else uniform over the format's long range; distance uniform in
 *   [1, min(produced, W)], forced to 1 w.p. 0.05 (RLE), < length w.p. 0.1
 *   (self-overlap); the last token is clipped so the stream ends exactly at the
 *   target size.
 * LZ4/LZO/Snappy: literal run geometric(mean 6), match length min+geometric(mean 10),
 *   distance uniform in [1, min(produced, Dmax)].
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "auroralz.h"

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t* r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint32_t rng_range(rng_t* r, uint32_t lo, uint32_t hi) { /* inclusive */
    return lo + (uint32_t)(rng_next(r) % (uint64_t)(hi - lo + 1));
}
static inline double rng_unit(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint32_t rng_geometric(rng_t* r, double mean) { /* >= 0, mean `mean` */
    double u = rng_unit(r); if (u <= 0) u = 1e-300;
    double p = 1.0 / (mean + 1.0);
    return (uint32_t)floor(log(u) / log(1.0 - p));
}

/* output sink that can also just count */
typedef struct { uint8_t* p; size_t len, cap; int fail; } out_t;
static inline void o_put(out_t* o, const void* d, size_t n) {
    if (o->p) { if (o->len + n > o->cap) { o->fail = 1; } else memcpy(o->p + o->len, d, n); }
    o->len += n;
}
static inline void o_u8(out_t* o, uint32_t v) { uint8_t x = (uint8_t)v; o_put(o, &x, 1); }

/* flag byte accumulator: flag byte is emitted before the payload of its 8 tokens */
typedef struct { out_t* base; uint8_t payload[32 * 4 + 16]; int plen; int bits_left; uint32_t cur; int msb; int neg; int nbytes; } fw_t;
static void fw_init(fw_t* f, out_t* base, int msb) { f->base = base; f->plen = 0; f->bits_left = 8; f->cur = 0; f->msb = msb; f->neg = 0; f->nbytes = 1; }
static void fw_width(fw_t* f, int nbytes) { f->nbytes = nbytes; f->bits_left = 8 * nbytes; }   /* big-endian flag words (LZHudson 4, SMSR00 2) */
static void fw_flush(fw_t* f) {
    if (f->bits_left != 8 * f->nbytes) {
        if (f->nbytes == 1) o_u8(f->base, f->neg ? (0u - f->cur) & 0xFFu : f->cur);   /* LZ40 stores the flag byte negated */
        else for (int i = f->nbytes - 1; i >= 0; i--) o_u8(f->base, (f->cur >> (8 * i)) & 0xFFu);
        f->bits_left = 8 * f->nbytes; f->cur = 0;
    }
    if (f->plen) { o_put(f->base, f->payload, (size_t)f->plen); f->plen = 0; }
}
static void fw_bit(fw_t* f, int bit) {
    if (bit) { int sh = f->msb ? f->bits_left - 1 : 8 * f->nbytes - f->bits_left; f->cur |= 1u << sh; }
    if (--f->bits_left == 0) fw_flush(f);
}
static inline void fw_pay(fw_t* f, uint32_t v) { f->payload[f->plen++] = (uint8_t)v; }
static void fw_flush_if_necessary(fw_t* f) { if (f->bits_left == 8 * f->nbytes && f->plen) { o_put(f->base, f->payload, (size_t)f->plen); f->plen = 0; } }

typedef struct { uint32_t len, dist; } tok_t;

/* draw a match for a stream that has produced `produced` bytes and may still produce `rem` */
static tok_t draw_match(rng_t* r, uint32_t produced, uint32_t rem, uint32_t minlen, uint32_t shortmax,
                        uint32_t longlo, uint32_t longhi, uint32_t maxdist) {
    tok_t t;
    if (longhi && rng_unit(r) >= 0.9) t.len = rng_range(r, longlo, longhi);
    else t.len = rng_range(r, minlen, shortmax);
    if (t.len > rem) t.len = rem;
    uint32_t maxd = produced < maxdist ? produced : maxdist;
    double u = rng_unit(r);
    if (u < 0.05) t.dist = 1;
    else if (u < 0.15 && t.len > 1) { uint32_t hi = t.len - 1 < maxd ? t.len - 1 : maxd; t.dist = rng_range(r, 1, hi); }
    else t.dist = rng_range(r, 1, maxd);
    return t;
}

static alz_lz_properties lzss_eff(const alz_lz_properties* p) {
    alz_lz_properties d;
    if (!p || p->window_bits == 0) { memset(&d, 0, sizeof(d)); d.window_bits = 12; d.length_bits = 4; d.min_length = 3; d.max_distance = 4096; d.windows_start = 0xFEE; }
    else { d = *p; if (!d.max_distance) d.max_distance = 1u << d.window_bits; }
    return d;
}

/* ---- flag-byte family ---- */
static void gen_flagfmt(uint32_t format, const alz_lz_properties* props, rng_t* r, uint32_t target, out_t* out, alz_encode_aux* aux) {
    alz_lz_properties lz = lzss_eff(props);
    uint32_t minlen = 3, shortmax = 18, longlo = 0, longhi = 0, W = 4096;
    int msb = 1;
    switch (format) {
    case ALZ_FMT_LZSS: msb = 0; minlen = lz.min_length; shortmax = (1u << lz.length_bits) + lz.min_length - 1; if (shortmax > 18 && lz.length_bits <= 4) shortmax = 18;
        if (lz.length_bits > 4) { longlo = 19; longhi = (1u << lz.length_bits) + lz.min_length - 1; shortmax = 18; }
        W = 1u << lz.window_bits; break;
    case ALZ_FMT_LZ10: case ALZ_FMT_MIO0: break;
    case ALZ_FMT_CLZ0: msb = 0; break;
    case ALZ_FMT_LZ02: shortmax = 16; longlo = 17; longhi = 272; W = 4095; break;
    case ALZ_FMT_LZ11: longlo = 17; longhi = 272; break;
    case ALZ_FMT_LZ40: shortmax = 15; longlo = 16; longhi = 271; break;
    case ALZ_FMT_YAZ0: case ALZ_FMT_YAY0: case ALZ_FMT_LZHUDSON: longlo = 18; longhi = 273; break;
    case ALZ_FMT_SMSR00: break;
    default: break;
    }
    int three = (format == ALZ_FMT_YAY0 || format == ALZ_FMT_MIO0);
    int smsr = (format == ALZ_FMT_SMSR00);     /* codes (masks + match words through the flag writer) | literals */
    /* three-section formats buffer tokens/literals separately */
    out_t flags = { NULL, 0, 0, 0 }, comp = { NULL, 0, 0, 0 }, unc = { NULL, 0, 0, 0 };
    uint8_t *fb = NULL, *cb = NULL, *ub = NULL;
    if ((three || smsr) && out->p) {
        size_t cap = (size_t)target + (target >> 2) + 64;
        const size_t fcap = smsr ? cap : cap / 8 + 64;     /* SMSR00: the flag writer's stream is the whole code section */
        fb = (uint8_t*)malloc(fcap); cb = (uint8_t*)malloc(cap); ub = (uint8_t*)malloc(cap);
        flags.p = fb; flags.cap = fcap; comp.p = cb; comp.cap = cap; unc.p = ub; unc.cap = cap;
    }
    fw_t fw; fw_init(&fw, (three || smsr) ? &flags : out, msb);
    fw.neg = format == ALZ_FMT_LZ40;
    if (format == ALZ_FMT_LZHUDSON) fw_width(&fw, 4);
    if (smsr) fw_width(&fw, 2);
    uint32_t produced = 0;
    while (produced < target) {
        uint32_t rem = target - produced;
        int lit = produced == 0 || rem < minlen || rng_unit(r) < 0.5;
        if (lit) {
            uint32_t b = (uint32_t)(rng_next(r) & 0xFF);
            switch (format) {
            case ALZ_FMT_LZSS: fw_pay(&fw, b); fw_bit(&fw, 1); break;
            case ALZ_FMT_LZ10: case ALZ_FMT_LZ11: case ALZ_FMT_LZ40: case ALZ_FMT_CLZ0: case ALZ_FMT_LZ02: fw_pay(&fw, b); fw_bit(&fw, 0); break;
            case ALZ_FMT_YAZ0: case ALZ_FMT_LZHUDSON: fw_pay(&fw, b); fw_bit(&fw, 1); break;
            default: o_u8(&unc, b); fw_bit(&fw, 1); break; /* YAY0 / MIO0 / SMSR00 */
            }
            produced++;
            continue;
        }
        tok_t t = draw_match(r, produced, rem, minlen, shortmax, longlo, longhi, W);
        if ((format == ALZ_FMT_LZ11 || format == ALZ_FMT_LZ40) && rng_unit(r) < 0.002 && rem > 273) { t.len = rng_range(r, 273, rem < 2000 ? rem : 2000); }
        uint32_t d1 = t.dist - 1;
        switch (format) {
        case ALZ_FMT_LZSS: {
            uint32_t off = (lz.windows_start + produced - t.dist) & (W - 1);
            fw_pay(&fw, off & 0xFF);
            fw_pay(&fw, ((off >> 8) << lz.length_bits) | ((t.len - lz.min_length) & ((1u << lz.length_bits) - 1)));
            fw_bit(&fw, 0); break;
        }
        case ALZ_FMT_LZ10: fw_pay(&fw, ((t.len - 3) << 4) | (d1 >> 8)); fw_pay(&fw, d1 & 0xFF); fw_bit(&fw, 1); break;
        case ALZ_FMT_LZ02:                                     /* DDDDLLLL DDDDDDDD [+ length - 17]  LZ02.cs:136-141 */
            if (t.len <= 16) { fw_pay(&fw, ((t.dist >> 8) << 4) | (t.len - 1)); fw_pay(&fw, t.dist & 0xFF); }
            else { fw_pay(&fw, (t.dist >> 8) << 4); fw_pay(&fw, t.dist & 0xFF); fw_pay(&fw, t.len - 17); }
            fw_bit(&fw, 1); break;
        case ALZ_FMT_CLZ0: { uint32_t delta = 0x1000 - t.dist; fw_pay(&fw, delta & 0xFF); fw_pay(&fw, (t.len - 3) | ((delta >> 8) << 4)); fw_bit(&fw, 1); break; }   /* CLZ0.cs:121-124 */
        case ALZ_FMT_LZ11:
            if (t.len <= 16) { fw_pay(&fw, ((t.len - 1) << 4) | (d1 >> 8)); fw_pay(&fw, d1 & 0xFF); }
            else if (t.len <= 272) { uint32_t l = t.len - 17; fw_pay(&fw, l >> 4); fw_pay(&fw, ((l & 0xF) << 4) | (d1 >> 8)); fw_pay(&fw, d1 & 0xFF); }
            else { uint32_t l = t.len - 273; fw_pay(&fw, 0x10 | (l >> 12)); fw_pay(&fw, (l >> 4) & 0xFF); fw_pay(&fw, ((l & 0xF) << 4) | (d1 >> 8)); fw_pay(&fw, d1 & 0xFF); }
            fw_bit(&fw, 1); break;
        case ALZ_FMT_LZ40: {                                   /* u16 LE distance << 4 | length; distance 4096 wraps to 0 (LZ40.cs:158-171) */
            uint32_t dv = (t.dist << 4) & 0xFFFF;
            if (t.len < 16) { uint32_t v = dv | t.len; fw_pay(&fw, v & 0xFF); fw_pay(&fw, v >> 8); }
            else if (t.len < 272) { fw_pay(&fw, dv & 0xFF); fw_pay(&fw, dv >> 8); fw_pay(&fw, t.len - 16); }
            else { uint32_t v = dv | 1, l = t.len - 272; fw_pay(&fw, v & 0xFF); fw_pay(&fw, v >> 8); fw_pay(&fw, l & 0xFF); fw_pay(&fw, l >> 8); }
            fw_bit(&fw, 1); break;
        }
        case ALZ_FMT_SMSR00: fw_pay(&fw, ((t.len - 3) << 4) | (d1 >> 8)); fw_pay(&fw, d1 & 0xFF); fw_bit(&fw, 0); break;
        case ALZ_FMT_YAZ0: case ALZ_FMT_LZHUDSON:
            if (t.len < 18) { fw_pay(&fw, ((t.len - 2) << 4) | (d1 >> 8)); fw_pay(&fw, d1 & 0xFF); }
            else { fw_pay(&fw, d1 >> 8); fw_pay(&fw, d1 & 0xFF); fw_pay(&fw, t.len - 0x12); }
            fw_bit(&fw, 0); break;
        case ALZ_FMT_YAY0:
            if (t.len < 18) { o_u8(&comp, ((t.len - 2) << 4) | (d1 >> 8)); o_u8(&comp, d1 & 0xFF); }
            else { o_u8(&comp, d1 >> 8); o_u8(&comp, d1 & 0xFF); o_u8(&unc, t.len - 0x12); }
            fw_bit(&fw, 0); break;
        default: /* MIO0 */
            o_u8(&comp, ((t.len - 3) << 4) | (d1 >> 8)); o_u8(&comp, d1 & 0xFF); fw_bit(&fw, 0); break;
        }
        produced += t.len;
    }
    if (format == ALZ_FMT_LZ02) { fw_pay(&fw, 0); fw_pay(&fw, 0); fw_bit(&fw, 1); }   /* terminator */
    fw_flush(&fw);
    if (smsr) {
        if (aux) { aux->aux0 = (uint32_t)flags.len; aux->aux1 = 0; }
        if (out->p) { o_put(out, fb, flags.len); o_put(out, ub, unc.len); } else out->len += flags.len + unc.len;
        if (flags.fail || unc.fail) out->fail = 1;
        free(fb); free(cb); free(ub);
    } else if (three) {
        if (aux) { aux->aux0 = (uint32_t)flags.len; aux->aux1 = (uint32_t)(flags.len + comp.len); }
        if (out->p) { o_put(out, fb, flags.len); o_put(out, cb, comp.len); o_put(out, ub, unc.len); }
        else out->len += flags.len + comp.len + unc.len;
        if (flags.fail || comp.fail || unc.fail) out->fail = 1;
        free(fb); free(cb); free(ub);
    }
}

/* ---- PRS (Sega/PRS.cs:104-159 emission rules) ---- */
static void gen_prs(rng_t* r, uint32_t target, out_t* out, int big) {
    fw_t fw; fw_init(&fw, out, big);
    uint32_t produced = 0;
    while (produced < target) {
        uint32_t rem = target - produced;
        int lit = produced == 0 || rem < 2 || rng_unit(r) < 0.5;
        if (lit) { fw_pay(&fw, (uint32_t)(rng_next(r) & 0xFF)); fw_bit(&fw, 1); produced++; continue; }
        tok_t t = draw_match(r, produced, rem, 2, 18, 10, 256, 0x1FFF);
        if (t.len == 2 && t.dist > 0x100) t.dist = rng_range(r, 1, produced < 0x100 ? produced : 0x100);
        fw_bit(&fw, 0);
        if (t.dist <= 0x100 && t.len <= 5) {
            fw_bit(&fw, 0);
            fw_bit(&fw, ((t.len - 2) >> 1) & 1); fw_bit(&fw, (t.len - 2) & 1);
            fw_pay(&fw, (0x100 - t.dist) & 0xFF);
            fw_flush_if_necessary(&fw);
        } else {
            uint32_t v = ((0x2000 - t.dist) << 3) & 0xFFFF;
            if (t.len > 9) { if (big) { fw_pay(&fw, v >> 8); fw_pay(&fw, v & 0xFF); } else { fw_pay(&fw, v & 0xFF); fw_pay(&fw, v >> 8); } fw_pay(&fw, t.len - 1); }
            else { v |= (t.len - 2); if (big) { fw_pay(&fw, v >> 8); fw_pay(&fw, v & 0xFF); } else { fw_pay(&fw, v & 0xFF); fw_pay(&fw, v >> 8); } }
            fw_bit(&fw, 1);
        }
        produced += t.len;
    }
    fw_bit(&fw, 0); fw_pay(&fw, 0); fw_pay(&fw, 0); fw_bit(&fw, 1);
    fw_flush(&fw);
}

static void put_rand(out_t* o, rng_t* r, uint32_t n) {
    while (n >= 8) { uint64_t v = rng_next(r); o_put(o, &v, 8); n -= 8; }
    if (n) { uint64_t v = rng_next(r); o_put(o, &v, n); }
}

/* experiment knob (environment ALZ_SYNTH_MAXDIST, read by alz_synth_batch): caps the match distance of the LZ4 / LZO / Snappy
 * generators below the format maximum, to separate the cost of sources older than the LDS window from the rest */
static uint32_t g_seq_maxdist = 0;
static tok_t draw_match_seq(rng_t* r, uint32_t produced, uint32_t rem, uint32_t minlen, uint32_t maxlen, uint32_t maxdist) {
    tok_t t;
    if (rng_unit(r) < 0.1) t.len = rng_range(r, 19, 300); else t.len = minlen + rng_geometric(r, 10.0);
    if (t.len > maxlen) t.len = maxlen;
    if (t.len > rem) t.len = rem;
    if (g_seq_maxdist && g_seq_maxdist < maxdist) maxdist = g_seq_maxdist;
    uint32_t maxd = produced < maxdist ? produced : maxdist;
    double u = rng_unit(r);
    if (u < 0.05) t.dist = 1;
    else if (u < 0.15 && t.len > 1) { uint32_t hi = t.len - 1 < maxd ? t.len - 1 : maxd; t.dist = rng_range(r, 1, hi); }
    else t.dist = rng_range(r, 1, maxd);
    return t;
}

/* ---- LZ4 block ---- */
static void lz4_ext(out_t* o, uint32_t len) { if (len >= 15) { len -= 15; while (len >= 255) { o_u8(o, 255); len -= 255; } o_u8(o, len); } }
static void gen_lz4(rng_t* r, uint32_t target, out_t* out) {
    uint32_t produced = 0;
    for (;;) {
        uint32_t rem = target - produced;
        uint32_t lit = rng_geometric(r, 6.0); if (produced == 0 && lit == 0) lit = 1;
        /* final sequence: literals only, at least 5 bytes where the block allows it */
        if (lit + 4 + 5 > rem) {
            lit = rem;
            o_u8(out, (lit > 15 ? 15 : lit) << 4); lz4_ext(out, lit); put_rand(out, r, lit);
            return;
        }
        tok_t t = draw_match_seq(r, produced + lit, rem - lit - 5, 4, 0x7FFFFFFF, 65535);
        if (t.len < 4) t.len = 4;
        o_u8(out, ((lit > 15 ? 15 : lit) << 4) | (t.len - 4 > 15 ? 15 : t.len - 4));
        lz4_ext(out, lit); put_rand(out, r, lit);
        o_u8(out, t.dist & 0xFF); o_u8(out, t.dist >> 8);
        lz4_ext(out, t.len - 4);
        produced += lit + t.len;
    }
}

/* ---- Snappy raw ---- */
static void gen_snappy(rng_t* r, uint32_t target, out_t* out) {
    uint32_t v = target; while (v >= 0x80) { o_u8(out, v | 0x80); v >>= 7; } o_u8(out, v);
    uint32_t produced = 0;
    while (produced < target) {
        uint32_t rem = target - produced;
        uint32_t lit = 1 + rng_geometric(r, 5.0); if (lit > rem) lit = rem;
        if (rng_unit(r) < 0.01) { lit = rng_range(r, 61, 400); if (lit > rem) lit = rem; }
        if (lit <= 60) o_u8(out, (lit - 1) << 2);
        else if (lit - 1 <= 0xFF) { o_u8(out, 60 << 2); o_u8(out, lit - 1); }
        else { o_u8(out, 61 << 2); o_u8(out, (lit - 1) & 0xFF); o_u8(out, (lit - 1) >> 8); }
        put_rand(out, r, lit); produced += lit; rem -= lit;
        if (rem < 4) { if (rem) { o_u8(out, (rem - 1) << 2); put_rand(out, r, rem); produced += rem; } continue; }
        tok_t t = draw_match_seq(r, produced, rem, 4, 64, 65535);
        if (t.len < 4) t.len = 4;
        if (t.dist < 2048 && t.len <= 11) { o_u8(out, 1 | ((t.len - 4) << 2) | ((t.dist >> 8) << 5)); o_u8(out, t.dist & 0xFF); }
        else { o_u8(out, 2 | ((t.len - 1) << 2)); o_u8(out, t.dist & 0xFF); o_u8(out, t.dist >> 8); }
        produced += t.len;
    }
}

/* ---- FastLZ (Formats/Common/FastLZ.cs:63-160): level 1 (8 KiB window, one length byte) or level 2 (length chains,
 * 16-bit offset extension); the level is drawn from the stream's seed ---- */
static void gen_fastlz(rng_t* r, uint32_t target, out_t* out) {
    const int level2 = rng_unit(r) < 0.5;
    uint32_t produced = 0; int first = 1;
    if (target == 0) { o_u8(out, level2 ? 0x20 : 0x00); o_u8(out, 0x55); return; }   /* (a stream cannot be empty: one literal, clipped by capacity) */
    while (produced < target) {
        uint32_t rem = target - produced;
        uint32_t lit = rng_geometric(r, 6.0); if (first && lit == 0) lit = 1;
        if (lit > rem) lit = rem;
        while (lit) {                                                  /* runs of 1..32 */
            uint32_t chunk = lit < 32 ? lit : 32;
            o_u8(out, (chunk - 1) | (first && level2 ? 0x20 : 0)); first = 0;
            put_rand(out, r, chunk); produced += chunk; lit -= chunk;
        }
        rem = target - produced;
        if (rem < 3) { if (rem) { o_u8(out, rem - 1); put_rand(out, r, rem); produced += rem; } continue; }
        tok_t t = draw_match_seq(r, produced, rem, 3, level2 ? 0x7FFFFFFF : 264, level2 ? 0x11FFF : 0x2000);
        if (t.len < 3) t.len = 3;
        if (level2 && rng_unit(r) < 0.02 && rem > 600) t.len = rng_range(r, 264, rem < 3000 ? rem : 3000);   /* chained length bytes */
        uint32_t length = t.len - 3, distance = t.dist - 1;
        uint32_t sd = level2 && distance > 0x1FFF ? 0x1FFF : distance;
        o_u8(out, (((length < 6 ? length : 6) + 1) << 5) | (sd >> 8));
        if (length >= 6) {
            length -= 6;
            while (level2 && length >= 255) { o_u8(out, 255); length -= 255; }
            o_u8(out, length);
        }
        o_u8(out, sd & 0xFF);
        if (level2 && distance >= 0x1FFF) { distance -= 0x1FFF; o_u8(out, distance >> 8); o_u8(out, distance & 0xFF); }
        produced += t.len;
    }
}

/* ---- CNX2 (Sega/CNX2.cs:83-139): 2-bit codes, four per flag byte (LSB first); code 0 (skip n bytes, drop the rest of the
 * flag byte) is sprinkled in although the managed encoder never writes it ---- */
static void gen_cnx2(rng_t* r, uint32_t target, out_t* out) {
    uint32_t produced = 0;
    uint8_t pay[4 * 260]; uint32_t plen = 0, flag = 0, ncodes = 0;
    while (produced < target) {
        uint32_t rem = target - produced, code;
        double u = rng_unit(r);
        if (u < 0.01) code = 0; else if (u < 0.40) code = 1; else if (u < 0.50 || rem < 4) code = 3; else code = 2;
        if (code == 2 && produced == 0) code = 1;
        flag |= code << (2 * ncodes); ncodes++;
        if (code == 0) {                                              /* skip: the decoder jumps over n junk bytes and resets the flag reader */
            uint32_t n = rng_range(r, 0, 40);
            pay[plen++] = (uint8_t)n; for (uint32_t i = 0; i < n; i++) pay[plen++] = (uint8_t)rng_next(r);
            o_u8(out, flag); o_put(out, pay, plen); plen = 0; flag = 0; ncodes = 0;
            continue;
        } else if (code == 1) { pay[plen++] = (uint8_t)rng_next(r); produced += 1; }
        else if (code == 3) {
            uint32_t n = rng_unit(r) < 0.1 ? rng_range(r, 0, 255) : 2 + rng_geometric(r, 6.0);
            if (n > 255) n = 255; if (n > rem) n = rem;
            pay[plen++] = (uint8_t)n; for (uint32_t i = 0; i < n; i++) pay[plen++] = (uint8_t)rng_next(r);
            produced += n;
        } else {
            tok_t t = draw_match(r, produced, rem, 4, 35, 0, 0, 2048);
            if (t.len < 4) t.len = 4;                                  /* (rem >= 4 here) */
            uint32_t v = ((t.dist - 1) << 5) | (t.len - 4);
            pay[plen++] = (uint8_t)(v >> 8); pay[plen++] = (uint8_t)v;
            produced += t.len;
        }
        if (ncodes == 4) { o_u8(out, flag); o_put(out, pay, plen); plen = 0; flag = 0; ncodes = 0; }
    }
    if (ncodes) { o_u8(out, flag); o_put(out, pay, plen); }
}

/* ---- BLZ in stream order (Nintendo/BLZ.cs:97-135 read back to front): flag bytes MSB first, 1 = match, big-endian u16
 * (length - 3) << 12 | (distance - 3); the stream ends exactly where the output is full ---- */
static void gen_blz(rng_t* r, uint32_t target, out_t* out) {
    uint32_t produced = 0;
    uint8_t pay[16]; uint32_t plen = 0, flag = 0, nbits = 0;
    while (produced < target) {
        uint32_t rem = target - produced;
        int match = produced >= 3 && rem >= 3 && rng_unit(r) < 0.5;
        if (match) {
            tok_t t = draw_match(r, produced, rem, 3, 18, 0, 0, 4098);
            if (t.dist < 3) t.dist = 3;
            if (t.len < 3) t.len = 3;
            uint32_t v = ((t.len - 3) << 12) | (t.dist - 3);
            pay[plen++] = (uint8_t)(v >> 8); pay[plen++] = (uint8_t)v;
            flag |= 0x80u >> nbits; produced += t.len;
        } else { pay[plen++] = (uint8_t)rng_next(r); produced += 1; }
        if (++nbits == 8) { o_u8(out, flag); o_put(out, pay, plen); plen = 0; flag = 0; nbits = 0; }
    }
    if (nbits) { o_u8(out, flag); o_put(out, pay, plen); }
}

/* ---- RefPack (EA/RefPack.cs:177-245): 0-3 literals + a match in one of three forms, literal runs of 4..112 (multiples of
 * four), and the end token with the last 0-3 literals ---- */
static void gen_refpack(rng_t* r, uint32_t target, out_t* out) {
    uint32_t produced = 0;
    for (;;) {
        uint32_t rem = target - produced;
        uint32_t lit = rng_geometric(r, 4.0); if (produced == 0 && lit == 0) lit = 1;
        if (lit + 3 > rem) {                                           /* finish: runs of multiples of four, then the end token */
            lit = rem;
            while (lit > 3) { uint32_t c = (lit > 0x70 ? 0x70 : lit) / 4; o_u8(out, 0xE0 | (c - 1)); put_rand(out, r, 4 * c); lit -= 4 * c; }
            o_u8(out, 0xFC | lit); put_rand(out, r, lit);
            return;
        }
        while (lit > 3) { uint32_t c = (lit > 0x70 ? 0x70 : lit) / 4; o_u8(out, 0xE0 | (c - 1)); put_rand(out, r, 4 * c); produced += 4 * c; lit -= 4 * c; }
        rem = target - produced;
        tok_t t = draw_match_seq(r, produced + lit, rem - lit, 3, 1028, 0x20000);
        if (t.len < 3) t.len = 3;
        if (t.len == 3 && t.dist > 0x400) t.dist = rng_range(r, 1, 0x400);      /* only the short form holds length 3 ... */
        if (t.len == 4 && t.dist > 0x4000) t.dist = rng_range(r, 1, 0x4000);    /* ... and only short / medium length 4 */
        const uint32_t d1 = t.dist - 1;
        const int can_s = t.len <= 10 && t.dist <= 0x400, can_m = t.len >= 4 && t.len <= 67 && t.dist <= 0x4000, can_l = t.len >= 5;
        const double u = rng_unit(r);                                            /* mostly the shortest form, sometimes a longer one */
        if (can_s && (u < 0.8 || (!can_m && !can_l))) { o_u8(out, lit | ((d1 & 0x300) >> 3) | ((t.len - 3) << 2)); o_u8(out, d1 & 0xFF); }
        else if (can_m && (u < 0.95 || !can_l)) { o_u8(out, 0x80 | (t.len - 4)); o_u8(out, (d1 >> 8) | (lit << 6)); o_u8(out, d1 & 0xFF); }
        else { o_u8(out, 0xC0 | ((d1 >> 16) << 4) | (((t.len - 5) >> 8) << 2) | lit); o_u8(out, (d1 >> 8) & 0xFF); o_u8(out, d1 & 0xFF); o_u8(out, (t.len - 5) & 0xFF); }
        put_rand(out, r, lit);
        produced += lit + t.len;
    }
}

/* ---- HIG (Specialized/HIG.cs:126-212): initial literal block, then matches in three forms, each followed by 0 / 1 / 2 / counted literals ---- */
static void hig_raw(out_t* out, rng_t* r, uint32_t n) {                /* count byte n - 2 (1..255), or 0 + the count as u16 LE */
    if (n >= 3 && n <= 257) o_u8(out, n - 2);
    else { o_u8(out, 0); o_u8(out, n & 0xFF); o_u8(out, n >> 8); }
    put_rand(out, r, n);
}
static void gen_hig(rng_t* r, uint32_t target, out_t* out) {
    uint32_t first = 2 + rng_geometric(r, 6.0);
    if (first > target || target - first < 4) first = target;          /* a match needs 4 bytes: never leave 1..3 */
    hig_raw(out, r, first);
    uint32_t produced = first;
    while (produced < target) {
        uint32_t rem = target - produced;                              /* >= 4 */
        tok_t t = draw_match_seq(r, produced, rem, 4, 0xFFFF, 0x7FFF);
        if (t.len < 4) t.len = 4;
        if (rng_unit(r) < 0.01 && rem > 400) t.len = rng_range(r, 274, rem < 5000 ? rem : 5000);
        uint32_t after = rem - t.len;
        uint32_t plain = rng_unit(r) < 0.3 ? 0 : rng_geometric(r, 3.0);
        if (rng_unit(r) < 0.01) plain = rng_range(r, 250, 700);
        if (plain > after) plain = after;
        if (after - plain < 4) plain = after;                          /* the rest as literals: no room for another match */
        const uint32_t pf = plain == 0 ? 3 : (plain == 1 ? 1 : (plain == 2 ? 2 : 0));
        const int f1 = t.dist <= 0x7FF && t.len <= 9, f2 = t.dist <= 0x3FFF && t.len <= 35;
        const double u = rng_unit(r);
        if (f1 && u < 0.8) o_u8(out, pf | ((t.len - 4) << 5) | ((t.dist >> 6) & 0x1C));
        else {
            if (f2 && u < 0.95) o_u8(out, 0xC0 | (t.len - 4));
            else {
                const uint32_t l = t.len <= 18 ? t.len - 3 : 0;
                o_u8(out, 0xE0 | ((t.dist >> 10) & 0x10) | l);
                if (l == 0) {
                    if (t.len <= 273) o_u8(out, t.len - 18);
                    else { o_u8(out, 0); o_u8(out, t.len >> 8); o_u8(out, t.len & 0xFF); }
                }
            }
            o_u8(out, pf | ((t.dist >> 6) & 0xFC));
        }
        o_u8(out, t.dist & 0xFF);
        if (plain > 2) hig_raw(out, r, plain); else put_rand(out, r, plain);
        produced += t.len + plain;
    }
}

/* ---- LZShrek (Activision/LZShrek.cs:73-119): groups of flag (literal count field << 3 | matches - 1) + literals + 1..8 matches;
 * a match = flag (distance field << 3 | length 1..7, 0 = length byte follows) [+ length - 7] [+ distance extension] ---- */
static void shrek_field(out_t* o, uint32_t v, uint32_t low) {        /* v: 0..65821 */
    if (v > 285) { o_u8(o, (0x1F << 3) | low); }
    else if (v > 29) { o_u8(o, (0x1E << 3) | low); }
    else o_u8(o, (v << 3) | low);
}
static void gen_lzshrek(rng_t* r, uint32_t target, out_t* out) {
    uint32_t produced = 0;
    while (produced < target) {
        uint32_t rem = target - produced;
        uint32_t lit = rng_unit(r) < 0.02 ? rng_range(r, 30, 700) : rng_geometric(r, 6.0); if (produced == 0 && lit == 0) lit = 1;
        if (lit > rem) lit = rem;
        rem -= lit;
        /* the matches of the group: drawn first, because the flag in front of the literals counts them */
        tok_t ms[8]; uint32_t nm = 0, pos = produced + lit;
        uint32_t want = 1 + (uint32_t)(rng_next(r) & 7);
        while (nm < want && rem >= 3) {
            tok_t t = draw_match(r, pos, rem, 3, 18, 19, 262, 4096);
            if (t.len < 1) t.len = 1;
            ms[nm++] = t; pos += t.len; rem -= t.len;
        }
        if (nm == 0) {                                                 /* no room for a match: the remaining bytes as literals, then the end */
            lit += rem;
            shrek_field(out, lit, 0);
            if (lit > 285) { o_u8(out, (lit - 286) & 0xFF); o_u8(out, (lit - 286) >> 8); } else if (lit > 29) o_u8(out, lit - 30);
            put_rand(out, r, lit);
            produced += lit;
            break;
        }
        shrek_field(out, lit, nm - 1);
        if (lit > 285) { o_u8(out, (lit - 286) & 0xFF); o_u8(out, (lit - 286) >> 8); } else if (lit > 29) o_u8(out, lit - 30);
        put_rand(out, r, lit);
        for (uint32_t i = 0; i < nm; i++) {
            uint32_t d = ms[i].dist - 1, len = ms[i].len;
            shrek_field(out, d, len > 7 ? 0 : len);
            if (len > 7) o_u8(out, len - 7);
            if (d > 285) { o_u8(out, (d - 286) & 0xFF); o_u8(out, (d - 286) >> 8); } else if (d > 29) o_u8(out, d - 30);
        }
        produced = pos;
    }
    o_u8(out, 0); o_u8(out, 0); o_u8(out, 0); o_u8(out, 0);            /* end: a group whose first match has length byte 0 */
}

/* ---- WFLZ (WayForward/WFLZ.cs:130-159): blocks of (u16 distance, length - 4, literal count) + literals; 0/0/0 ends ---- */
static void gen_wflz(rng_t* r, uint32_t target, out_t* out, int big) {
    uint32_t produced = 0;
    while (produced < target) {
        uint32_t rem = target - produced;
        tok_t t = { 0, 0 };
        if (produced > 0 && rem >= 5 && rng_unit(r) < 0.85) { t = draw_match_seq(r, produced, rem, 5, 255, 65535); if (t.len < 5) t.len = 5; }
        rem -= t.len;
        uint32_t lit = rng_unit(r) < 0.03 ? rng_range(r, 0, 255) : rng_geometric(r, 5.0);
        if (lit > 255) lit = 255; if (lit > rem) lit = rem;
        if (t.len == 0 && lit == 0) lit = 1;                             /* (0/0/0 would be the end block) */
        if (big) { o_u8(out, t.dist >> 8); o_u8(out, t.dist & 0xFF); } else { o_u8(out, t.dist & 0xFF); o_u8(out, t.dist >> 8); }
        o_u8(out, t.len ? t.len - 4 : 0); o_u8(out, lit);
        put_rand(out, r, lit);
        produced += t.len + lit;
    }
    o_u8(out, 0); o_u8(out, 0); o_u8(out, 0); o_u8(out, 0);
}

/* ---- CNS (Specialized/CNS.cs:77-108): control byte < 0x80 = literal run, else match (c & 0x7F) + 3 at distance byte + 1 ---- */
static void gen_cns(rng_t* r, uint32_t target, out_t* out) {
    uint32_t produced = 0;
    while (produced < target) {
        uint32_t rem = target - produced;
        if (produced == 0 || rem < 3 || rng_unit(r) < 0.4) {
            uint32_t n = rng_unit(r) < 0.05 ? rng_range(r, 0, 127) : 1 + rng_geometric(r, 5.0);
            if (n > 127) n = 127; if (n > rem) n = rem;
            o_u8(out, n); put_rand(out, r, n); produced += n;
        } else {
            tok_t t = draw_match(r, produced, rem, 3, 18, 19, 130, 256);
            if (t.len < 3) t.len = 3;
            o_u8(out, 0x80 | (t.len - 3)); o_u8(out, t.dist - 1); produced += t.len;
        }
    }
}

/* ---- LZO (opcode forms of Formats/Common/LZO.cs:141-250) ---- */
static void lzo_ext(out_t* o, uint32_t v) { while (v > 255) { o_u8(o, 0); v -= 255; } o_u8(o, v); }
static void gen_lzo(rng_t* r, uint32_t target, out_t* out) {
    uint32_t produced = 0;
    if (target < 16) { /* LZO.cs:143-152 */
        o_u8(out, 17 + target); put_rand(out, r, target); o_u8(out, 0x11); o_u8(out, 0); o_u8(out, 0); return;
    }
    /* initial literal run >= 4 via the "plain copy" opcode (plain == 0 state) */
    uint32_t pending = 4 + rng_geometric(r, 4.0); if (pending > target) pending = target;
    for (;;) {
        /* emit a literal run of `pending` (>= 4) bytes */
        if (pending) {
            if (pending > 18) { o_u8(out, 0); lzo_ext(out, pending - 18); } else o_u8(out, pending - 3);
            put_rand(out, r, pending); produced += pending;
        }
        /* matches, each followed by 0..3 trailing literals, until we need a long run again */
        for (;;) {
            uint32_t rem = target - produced;
            if (rem < 3) {
                /* finish: trailing bytes must ride on a literal opcode of >= 4; rewind is impossible, so emit a
                   minimal match chain: rem in {0,1,2}.  rem==0: done.  else encode them as trailing literals of a
                   length-3 match drawn earlier -- handled by always keeping rem >= 3 + tail below. */
                goto done;
            }
            tok_t t = draw_match_seq(r, produced, rem, 3, 0x7FFFFFFF, 0xBFFF);
            if (t.len < 3) t.len = 3;
            uint32_t after = rem - t.len;
            /* choose trailing literal count so that the remainder never becomes 1..2 bytes of orphan literals */
            uint32_t tail = rng_range(r, 0, 3);
            uint32_t longrun = 0;
            if (rng_unit(r) < 0.35) { tail = 0; longrun = 4 + rng_geometric(r, 4.0); }
            if (tail > after) tail = after;
            after -= tail;
            if (longrun) { if (longrun > after) { if (after >= 4) longrun = after; else { longrun = 0; } } after -= longrun; }
            if (after > 0 && after < 3) { /* absorb the orphan bytes */
                if (longrun) { longrun += after; }
                else if (tail + after <= 3) { tail += after; }
                else { t.len += after; }
                after = 0;
            }
            /* encode match */
            if (t.len <= 8 && t.dist <= 2048) {
                uint32_t flag = tail | (((t.dist - 1) & 7) << 2);
                if (t.len <= 4) o_u8(out, flag | 0x40 | ((t.len - 3) << 5)); else o_u8(out, flag | 0x80 | ((t.len - 5) << 5));
                o_u8(out, (t.dist - 1) >> 3);
            } else if (t.dist <= 16384) {
                if (t.len > 33) { o_u8(out, 0x20); lzo_ext(out, t.len - 33); } else o_u8(out, 0x20 | (t.len - 2));
                o_u8(out, (tail | ((t.dist - 1) << 2)) & 0xFF); o_u8(out, (t.dist - 1) >> 6);
            } else {
                uint32_t d = t.dist - 0x4000; uint32_t flag = 0x10 | ((d & 0x4000) >> 11);
                if (t.len > 9) { o_u8(out, flag); lzo_ext(out, t.len - 9); } else o_u8(out, flag | (t.len - 2));
                o_u8(out, (tail | (d << 2)) & 0xFF); o_u8(out, (d >> 6) & 0xFF);
            }
            put_rand(out, r, tail);
            produced += t.len + tail;
            if (longrun) { pending = longrun; break; }
            if (produced >= target) goto done;
        }
        if (produced >= target) break;
    }
done:
    o_u8(out, 0x11); o_u8(out, 0); o_u8(out, 0);
}

/* one stream; dst may be NULL to only measure.  Returns compressed size (or -1 on overflow). */
int64_t alz_synth_stream(uint32_t format, const alz_lz_properties* props, uint64_t seed, uint32_t target,
                         uint8_t* dst, size_t cap, alz_encode_aux* aux) {
    rng_t r = { seed };
    out_t out = { dst, 0, cap, 0 };
    if (aux) { aux->aux0 = 0; aux->aux1 = 0; }
    switch (format) {
    case ALZ_FMT_CLZ0: case ALZ_FMT_LZ02:
    case ALZ_FMT_LZSS: case ALZ_FMT_LZ10: case ALZ_FMT_LZ11: case ALZ_FMT_LZ40: case ALZ_FMT_YAZ0: case ALZ_FMT_YAY0: case ALZ_FMT_MIO0:
    case ALZ_FMT_LZHUDSON: case ALZ_FMT_SMSR00:
        gen_flagfmt(format, props, &r, target, &out, aux); break;
    case ALZ_FMT_PRS_BE: gen_prs(&r, target, &out, 1); break;
    case ALZ_FMT_PRS_LE: gen_prs(&r, target, &out, 0); break;
    case ALZ_FMT_LZ4_BLOCK: gen_lz4(&r, target, &out); break;
    case ALZ_FMT_FASTLZ: gen_fastlz(&r, target, &out); break;
    case ALZ_FMT_CNX2: gen_cnx2(&r, target, &out); break;
    case ALZ_FMT_BLZ: gen_blz(&r, target, &out); break;
    case ALZ_FMT_CNS: gen_cns(&r, target, &out); break;
    case ALZ_FMT_REFPACK: gen_refpack(&r, target, &out); break;
    case ALZ_FMT_WFLZ: gen_wflz(&r, target, &out, 0); break;
    case ALZ_FMT_LZSHREK: gen_lzshrek(&r, target, &out); break;
    case ALZ_FMT_HIG: gen_hig(&r, target, &out); break;
    case ALZ_FMT_WFLZ_BE: gen_wflz(&r, target, &out, 1); break;
    case ALZ_FMT_LZO: gen_lzo(&r, target, &out); break;
    case ALZ_FMT_SNAPPY_RAW: gen_snappy(&r, target, &out); break;
    default: return -2;
    }
    return out.fail ? -1 : (int64_t)out.len;
}

typedef struct {
    const uint32_t* formats; uint32_t format; const alz_lz_properties* props; uint64_t base_seed; uint32_t n;
    const uint32_t* targets; uint32_t target; uint8_t* dst; const uint64_t* offs; uint32_t* sizes; alz_encode_aux* aux;
    int tid, nt;
    const uint64_t* seeds;      /* optional: seed of stream i (else base_seed + i) */
} sjob_t;

static void* synth_worker(void* arg) {
    sjob_t* j = (sjob_t*)arg;
    for (uint32_t i = (uint32_t)j->tid; i < j->n; i += (uint32_t)j->nt) {
        uint32_t fmt = j->formats ? j->formats[i] : j->format;
        uint32_t tgt = j->targets ? j->targets[i] : j->target;
        alz_encode_aux a;
        const uint64_t seed = j->seeds ? j->seeds[i] : j->base_seed + i;
        if (!j->dst) j->sizes[i] = (uint32_t)alz_synth_stream(fmt, j->props, seed, tgt, NULL, 0, &a);
        else (void)alz_synth_stream(fmt, j->props, seed, tgt, j->dst + j->offs[i], j->sizes[i], &a);
        if (j->aux) j->aux[i] = a;
    }
    return NULL;
}

/* Two-pass batch: call with dst == NULL to get sizes[], lay the streams out, call again with dst/offs. */
int alz_synth_batch_seeds(const uint32_t* formats, uint32_t format, const alz_lz_properties* props, uint64_t base_seed, const uint64_t* seeds, uint32_t n,
                    const uint32_t* targets, uint32_t target, uint8_t* dst, const uint64_t* offs, uint32_t* sizes,
                    alz_encode_aux* aux, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    { const char* e = getenv("ALZ_SYNTH_MAXDIST"); g_seq_maxdist = e ? (uint32_t)strtoul(e, NULL, 0) : 0; }
    pthread_t th[256]; sjob_t jobs[256];
    for (int t = 0; t < nthreads; t++) {
        sjob_t j = { formats, format, props, base_seed, n, targets, target, dst, offs, sizes, aux, t, nthreads, seeds };
        jobs[t] = j;
        if (nthreads == 1) synth_worker(&jobs[0]); else pthread_create(&th[t], NULL, synth_worker, &jobs[t]);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    return 0;
}

int alz_synth_batch(const uint32_t* formats, uint32_t format, const alz_lz_properties* props, uint64_t base_seed, uint32_t n,
                    const uint32_t* targets, uint32_t target, uint8_t* dst, const uint64_t* offs, uint32_t* sizes,
                    alz_encode_aux* aux, int nthreads) {
    return alz_synth_batch_seeds(formats, format, props, base_seed, NULL, n, targets, target, dst, offs, sizes, aux, nthreads);
}
