// alz_container.cpp -- host side of the reference's format classes, restated above the C ABI.
//
// In the reference every format class does two things: (1) parse/emit its container header in managed code and
// (2) run the headerless LZ body.  (2) is the hot path and runs on the GPU (alz_decode / alz_encode_batch); this file
// is (1): header parsing, IsMatch heuristics, GetDecompressedSize and the endianness-retry logic, with the same
// argument meaning and error behaviour as the managed classes (cited per function, paths relative to
// /root/reference/src).  A C# shim would keep (1) in managed code and P/Invoke only alz_decode/alz_encode_batch
// (INTEGRATION.md); hosts without the managed library use these entry points instead.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "auroralz.h"

namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t le32(const uint8_t* p) { return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]; }
inline uint32_t rd32(const uint8_t* p, bool big) { return big ? be32(p) : le32(p); }
inline uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }
inline void wr32(uint8_t* p, uint32_t v, bool big) {
    if (big) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
    else { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
}

// LZ10/LZ11.InternalGetDecompressedSize  Nintendo/LZ10.cs:47-57: id byte, u24 LE size, 0 => u32 LE
int nin_header(const uint8_t* src, size_t len, uint8_t id, uint32_t* size) {
    if (len < 4 || src[0] != id) return -1;
    uint32_t s = (uint32_t)src[1] | ((uint32_t)src[2] << 8) | ((uint32_t)src[3] << 16);
    if (s != 0) { *size = s; return 4; }
    if (len < 8) return -1;
    *size = le32(src + 4);
    return 8;
}

// LZ10.Validate / LZ11.Validate  Nintendo/LZ10.cs:139-175, LZ11.cs:173-223: walk the first tokens, every match must
// point inside what was produced; accept after 4 matches or when the whole stream adds up to the declared size.
bool nin_validate(const uint8_t* src, size_t len, bool lz11) {
    uint32_t size; int h = nin_header(src, len, lz11 ? 0x11 : 0x10, &size);
    if (h < 0 || size == 0) return false;
    size_t p = (size_t)h; int i = 3; uint64_t buffer = 0; int bits = 0; uint32_t flag = 0;
    while (p < len) {
        if (bits == 0) { flag = src[p++]; bits = 8; if (p > len) return false; }
        bool bit = (flag >> (bits - 1)) & 1; bits--;
        if (bit) {
            if (p + 2 > len) return false;
            uint32_t b1 = src[p], b2 = src[p + 1], distance, length;
            if (lz11 && (b1 >> 4) == 0) { if (p + 3 > len) return false; distance = (((b2 & 0xf) << 8) | src[p + 2]) + 1; length = (((b1 & 0xf) << 4) | (b2 >> 4)) + 17; p += 3; }
            else if (lz11 && (b1 >> 4) == 1) { if (p + 4 > len) return false; distance = (((src[p + 2] & 0xf) << 8) | src[p + 3]) + 1; length = (((b1 & 0xf) << 12) | (b2 << 4) | (src[p + 2] >> 4)) + 273; p += 4; }
            else { distance = (((b1 & 0xf) << 8) | b2) + 1; length = (b1 >> 4) + (lz11 ? 1 : 3); p += 2; }
            if (distance > buffer) return false;
            if (i == 0) return true;
            i--; buffer += length;
        } else { p++; buffer++; }
    }
    return buffer == size;
}

// PRS.ValidateByteOrder  Sega/PRS.cs:171-218
bool prs_validate(const uint8_t* src, size_t len, bool big) {
    size_t p = 0; int i = 3; uint64_t buffer = 0; int bits = 0; uint32_t flag = 0;
    auto readbit = [&](bool& ok) -> int {
        if (bits == 0) { if (p >= len) { ok = false; return 0; } flag = src[p++]; bits = 8; }
        int sh = big ? bits - 1 : 8 - bits; bits--; return (flag >> sh) & 1;
    };
    bool ok = true;
    while (p < len) {
        int b = readbit(ok); if (!ok) return false;
        if (b) { p++; buffer++; continue; }
        uint32_t distance, length;
        int b2 = readbit(ok); if (!ok) return false;
        if (b2) {
            if (p + 2 > len) return false;
            uint32_t v = big ? ((uint32_t)src[p] << 8 | src[p + 1]) : ((uint32_t)src[p + 1] << 8 | src[p]); p += 2;
            if (v == 0) return true;
            length = v & 7; distance = 0x2000 - (v >> 3);
            if (length == 0) { if (p >= len) return false; length = (uint32_t)src[p++] + 1; } else length += 2;
        } else {
            int h = readbit(ok); if (!ok) return false; int l = readbit(ok); if (!ok) return false;
            length = (uint32_t)((h << 1) | l) + 2;
            if (p >= len) return false;
            distance = 0x100 - src[p++];
        }
        if (distance > buffer) return false;
        if (i == 0) return true;
        i--; buffer += length;
    }
    return false;
}

// PRS.GetByteOrder  Sega/PRS.cs:161-169: 1 little, 2 big, 0 none
int prs_byte_order(const uint8_t* src, size_t len) {
    if (len == 0) return 0;
    uint8_t flag = src[0];
    if (flag > 12 && (flag & 0x1) == 1 && prs_validate(src, len, false)) return 1;
    if ((flag & 128) == 128 && prs_validate(src, len, true)) return 2;
    return 0;
}

const uint8_t kAklzMagic[12] = { 'A', 'K', 'L', 'Z', '~', '?', 'Q', 'd', '=', 0xCC, 0xCC, 0xCD };   // "AKLZ~?Qd=\xCC\xCC\xCD"  AKLZ.cs:16
const uint8_t kLzonMagic[8] = { 'L', 'Z', 'O', 'n', 0x00, 0x2F, 0xF1, 0x71 };                           // LZOn.cs:17

uint32_t clamp32(size_t v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; }

int run_body(alz_ctx* ctx, uint32_t fmt, const alz_lz_properties* lz, const uint8_t* body, size_t body_len, uint32_t size,
             uint32_t aux0, uint32_t aux1, uint8_t* dst, size_t cap, alz_result* r) {
    return alz_decode(ctx, fmt, lz, body, clamp32(body_len), size, aux0, aux1, dst, clamp32(cap), r);
}

// ---------------------------------------------------------------------------------------------- checksums (host side)
// XXH32 (the LZ4 frame format's checksum; the reference takes it as LZ4.HashAlgorithm, LZ4.Frame.cs:17-18)
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t xxh32(const uint8_t* p, size_t len, uint32_t seed) {
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const uint8_t* end = p + len; uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = rotl32(v1 + le32(p) * P2, 13) * P1; v2 = rotl32(v2 + le32(p + 4) * P2, 13) * P1;
            v3 = rotl32(v3 + le32(p + 8) * P2, 13) * P1; v4 = rotl32(v4 + le32(p + 12) * P2, 13) * P1; p += 16;
        } while (p + 16 <= end);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else h = seed + P5;
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + le32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl32(h + (*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}
// CRC-32C (Castagnoli, reflected 0x82F63B78); Snappy.cs:87 masks it with CRCMask (:252).  The host's crc32 instruction where it has one (SSE4.2: three
// independent streams of 8 bytes per step would be faster still; one is ~16 x the table walk already -- 16 MB of Snappy chunks 32 -> 2 ms, which was most of
// what a framed Compress cost), a byte-wise table otherwise.
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(const uint8_t* p, size_t len) {
    uint64_t c = 0xFFFFFFFFu;
    while (len && ((uintptr_t)p & 7u)) { c = __builtin_ia32_crc32qi((uint32_t)c, *p++); len--; }
    for (; len >= 8; len -= 8, p += 8) { uint64_t v; memcpy(&v, p, 8); c = __builtin_ia32_crc32di(c, v); }
    while (len--) c = __builtin_ia32_crc32qi((uint32_t)c, *p++);
    return (uint32_t)c ^ 0xFFFFFFFFu;
}
#endif
uint32_t crc32c(const uint8_t* p, size_t len) {
#if defined(__x86_64__)
    static const bool hw = __builtin_cpu_supports("sse4.2");
    if (hw) return crc32c_hw(p, len);
#endif
    static uint32_t table[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u))); table[i] = c; } init = true; }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < len; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
inline uint32_t snappy_crc_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

const uint8_t kSnappyId[10] = { 0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59 };                // Snappy.cs:17

inline bool lz4_magic_defined(uint32_t v) { return v == 0x184C2102u || v == 0x184D2204u || (v >= 0x184D2A50u && v <= 0x184D2A5Fu); }   // LZ4.Frame.cs:50-70

struct DevBuf {   // device allocation released on every exit path
    alz_ctx* c; void* p;
    explicit DevBuf(alz_ctx* ctx) : c(ctx), p(nullptr) {}
    ~DevBuf() { if (p) alz_device_free(c, p); }
};

// Does any match of this LZ4 block point in front of the block's own output?  (A walk over the sequences: input only.)
static bool lz4_block_reaches_back(const uint8_t* b, uint32_t n) {
    uint64_t produced = 0; uint32_t p = 0;
    while (p < n) {
        const uint32_t tok = b[p++];
        uint64_t lit = tok >> 4;
        if (lit == 15) { uint32_t x; do { if (p >= n) return false; x = b[p++]; lit += x; } while (x == 255); }
        if (lit > n - p) return false;                                  // truncated: the decoder reports it
        p += (uint32_t)lit; produced += lit;
        if (p >= n) break;
        if (p + 2 > n) return false;
        const uint32_t dist = b[p] | (b[p + 1] << 8); p += 2;
        uint64_t ml = tok & 15;
        if (ml == 15) { uint32_t x; do { if (p >= n) return false; x = b[p++]; ml += x; } while (x == 255); }
        if (dist > produced) return true;
        produced += ml + 4;
    }
    return false;
}

struct Lz4Block { size_t off; uint32_t len; bool raw; };

// A block's destination as the whole-GPU decode path of ONE stream wants it (alz_big_eligible: no more than 32 x the input + 64 KiB -- its launches are sized by the room in the
// destination): the frame's block maximum is 4 MiB whatever the block holds, and a 1 MB file in one block had 33 x its 126 KB of input to decode into -- 3.7 ms on two wavefronts
// instead of 0.28.  A block that does not fit the tighter room (it compressed better than 32 : 1) reports OUTPUT_CAPACITY and is decoded again with all of it.
inline uint32_t lz4_tight_cap(uint32_t cap, uint32_t src_len) {
    const uint64_t t = 32ull * src_len + 65536ull;
    return t < cap ? (uint32_t)t : cap;
}
// (worth the whole GPU by itself: plan_create's own test, restated for the split below -- a wrong guess costs time, not bytes)
inline bool lz4_long_block(const alz_stream& s) { return s.src_len >= 8192u && s.dst_cap >= (24u << 10); }

// LZ4.Decompress  Formats/Common/LZ4.cs:50-93 (+ ReadLZ4L :96-111, DecompressLZ4FrameHeader  LZ4.Frame.cs:107-174).
// The file and the output stay in HBM for the whole call.  Blocks that cannot reference each other (legacy frames:
// a fresh LzWindows per block, LZ4.cs:164; frames with the block-independence flag) go to the GPU as ONE batch at
// nominal offsets; blocks of a linked frame share one window (LZ4.Frame.cs:120) and run in order, each with the
// frame's earlier output as history (alz_stream.aux0).  Checksums are verified as with LZ4.HashAlgorithm = XXH32.
int lz4_file_decompress(alz_ctx* ctx, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* dst_len, size_t* src_used, int32_t* status) {
    DevBuf d_src(ctx), d_dst(ctx);
    int rc;
    if ((rc = alz_device_malloc(ctx, len + 64, &d_src.p)) != ALZ_OK) return rc;
    if ((rc = alz_device_malloc(ctx, cap + 64, &d_dst.p)) != ALZ_OK) return rc;
    if (len && (rc = alz_memcpy_h2d(ctx, d_src.p, src, len)) != ALZ_OK) return rc;
    size_t pos = 0, out = 0; int32_t st = ALZ_ST_OK;

    // runs `n` streams and returns their results
    auto run = [&](std::vector<alz_stream>& ss, std::vector<alz_result>& rs) -> int {
        alz_plan* pl = nullptr; rs.resize(ss.size());
        int e = alz_plan_create(ctx, nullptr, (uint32_t)ss.size(), ss.data(), &pl);
        if (e != ALZ_OK) return e;
        e = alz_plan_execute(ctx, pl, d_src.p, d_dst.p, nullptr);
        if (e == ALZ_OK) e = alz_plan_results(ctx, pl, rs.data());
        alz_plan_destroy(ctx, pl);
        return e;
    };
    // blocks in order, each seeing `out - origin` bytes of history when `linked`
    auto run_sequential = [&](const std::vector<Lz4Block>& bl, size_t first, size_t origin, bool linked) -> int {
        for (size_t i = first; i < bl.size() && st == ALZ_ST_OK; i++) {
            if (bl[i].raw) {
                if (out + bl[i].len > cap) { st = ALZ_ST_OUTPUT_CAPACITY; break; }
                int e = alz_memcpy_h2d(ctx, (uint8_t*)d_dst.p + out, src + bl[i].off, bl[i].len); if (e != ALZ_OK) return e;
                out += bl[i].len; continue;
            }
            std::vector<alz_stream> ss(1); std::vector<alz_result> rs;
            memset(&ss[0], 0, sizeof(alz_stream));
            const size_t hist = linked ? out - origin : 0;
            ss[0].src_off = bl[i].off; ss[0].src_len = bl[i].len; ss[0].dst_off = out; ss[0].dst_cap = clamp32(out < cap ? cap - out : 0);
            if ((uint64_t)ss[0].dst_cap + hist > 0xFFFFFF00ull) return ALZ_E_UNSUPPORTED;
            ss[0].aux0 = (uint32_t)hist; ss[0].format = ALZ_FMT_LZ4_BLOCK;
            const uint32_t full = ss[0].dst_cap;
            if (!linked) ss[0].dst_cap = lz4_tight_cap(full, ss[0].src_len);
            int e = run(ss, rs); if (e != ALZ_OK) return e;
            if (rs[0].status == ALZ_ST_OUTPUT_CAPACITY && ss[0].dst_cap < full) { ss[0].dst_cap = full; e = run(ss, rs); if (e != ALZ_OK) return e; }   // (a block that compressed more than 32 : 1)
            out += rs[0].dst_len;
            if (rs[0].status != ALZ_ST_OK) st = rs[0].status;
        }
        return ALZ_OK;
    };
    // independent blocks: one batch, block k at origin + k * nominal; falls back to in-order decoding from the first
    // block that did not fill its nominal slot
    auto run_independent = [&](const std::vector<Lz4Block>& bl, uint32_t nominal) -> int {
        if (bl.size() <= 1) return run_sequential(bl, 0, out, false);
        const size_t origin = out;
        std::vector<alz_stream> ss; std::vector<size_t> idx; std::vector<bool> tightened;
        for (size_t i = 0; i < bl.size(); i++) {
            const size_t o = origin + i * (size_t)nominal;
            if (bl[i].raw) continue;
            alz_stream s; memset(&s, 0, sizeof(s));
            s.src_off = bl[i].off; s.src_len = bl[i].len; s.dst_off = o < cap ? o : cap;
            s.dst_cap = clamp32(o < cap ? (cap - o < nominal ? cap - o : nominal) : 0); s.format = ALZ_FMT_LZ4_BLOCK;
            tightened.push_back(lz4_tight_cap(s.dst_cap, s.src_len) < s.dst_cap);
            s.dst_cap = lz4_tight_cap(s.dst_cap, s.src_len);
            ss.push_back(s); idx.push_back(i);
        }
        // A plan takes its streams one after the other on the whole GPU only when ALL of them are worth it (plan_create), and the last block of a file is usually a
        // short one: 16 MB in 4 MiB blocks -- four of them and 2 KB -- decoded every block on wavefronts of its own, 14 ms for the 4 MiB ones instead of 0.5 each.
        // The long blocks and the short ones go out as two plans.
        std::vector<alz_result> rs(ss.size());
        {
            std::vector<alz_stream> sa, sb; std::vector<size_t> ia, ib;
            for (size_t k = 0; k < ss.size(); k++) { if (lz4_long_block(ss[k])) { sa.push_back(ss[k]); ia.push_back(k); } else { sb.push_back(ss[k]); ib.push_back(k); } }
            if (!sa.empty() && !sb.empty() && sa.size() <= 32) {
                std::vector<alz_result> ra, rb;
                int e = run(sa, ra); if (e != ALZ_OK) return e;
                e = run(sb, rb); if (e != ALZ_OK) return e;
                for (size_t k = 0; k < ia.size(); k++) rs[ia[k]] = ra[k];
                for (size_t k = 0; k < ib.size(); k++) rs[ib[k]] = rb[k];
            } else if (!ss.empty()) { int e = run(ss, rs); if (e != ALZ_OK) return e; }
        }
        size_t k = 0;
        for (size_t i = 0; i < bl.size(); i++) {
            const size_t o = origin + i * (size_t)nominal;
            uint32_t produced; int32_t bst = ALZ_ST_OK;
            if (bl[i].raw) {
                produced = bl[i].len;
                if (out != o || out + produced > cap) return run_sequential(bl, i, origin, false);
                int e = alz_memcpy_h2d(ctx, (uint8_t*)d_dst.p + out, src + bl[i].off, produced); if (e != ALZ_OK) return e;
            } else {
                produced = rs[k].dst_len; bst = rs[k].status;
                const bool tight = tightened[k]; k++;
                if (out != o) return run_sequential(bl, i, origin, false);
                if (bst == ALZ_ST_OUTPUT_CAPACITY && tight) return run_sequential(bl, i, origin, false);             // it compressed better than 32 : 1: again, with all the room there is
            }
            if (bst == ALZ_ST_OUTPUT_CAPACITY && o + nominal <= cap) return run_sequential(bl, i, origin, false);   // the block is larger than nominal
            out = o + produced;
            if (bst != ALZ_ST_OK) { st = bst; return ALZ_OK; }
        }
        return ALZ_OK;
    };

    while (pos < len && st == ALZ_ST_OK) {
        if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
        uint32_t magic = le32(src + pos); pos += 4;
    again:
        if (magic == 0x184C2102u) {                                                      // legacy  LZ4.cs:96-111
            if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            uint32_t bs = le32(src + pos); pos += 4;
            std::vector<Lz4Block> bl; bool next = false, eof_flag = false, trunc = false;
            for (;;) {
                if (bs > len - pos) { trunc = true; break; }
                bl.push_back(Lz4Block{ pos, bs, false }); pos += bs;
                if (pos >= len) break;                                                   // ReadByte() == -1
                if (src[pos] == 0xFF) { pos++; eof_flag = true; break; }                 // (sbyte)0xFF == -1: the EOF flag
                if (pos + 4 > len) { trunc = true; break; }
                bs = le32(src + pos); pos += 4;
                if (lz4_magic_defined(bs)) { next = true; break; }
            }
            if ((rc = run_independent(bl, 0x800000u)) != ALZ_OK) return rc;
            if (st == ALZ_ST_OK && trunc) st = ALZ_ST_INPUT_TRUNCATED;
            if (st != ALZ_ST_OK) break;
            if (next) { magic = bs; goto again; }
            (void)eof_flag;
            break;                                                                       // blockSize == 0: Decompress returns
        } else if (magic == 0x184D2204u) {                                               // frame  LZ4.Frame.cs:107-174
            const size_t frame_start = out;
            if (pos + 2 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            const uint32_t flg = src[pos], bd = src[pos + 1]; pos += 2;
            uint32_t bmax;
            switch ((bd & 0x70) >> 4) { case 4: bmax = 0x10000; break; case 5: bmax = 0x40000; break; case 6: bmax = 0x100000; break; case 7: bmax = 0x400000; break; default: return ALZ_E_FORMAT; }
            uint64_t content = 0;
            if (flg & 8) { if (pos + 8 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; } content = (uint64_t)le32(src + pos) | ((uint64_t)le32(src + pos + 4) << 32); pos += 8; }
            if (flg & 1) { if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; } pos += 4; }
            if (pos + 1 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            pos += 1;                                                                    // HeaderChecksum: read, not verified
            if (flg & 1) return ALZ_E_UNSUPPORTED;                                       // external dictionaries  LZ4.Frame.cs:113-114
            std::vector<Lz4Block> bl; bool trunc = false;
            for (;;) {
                if (pos + 4 > len) { trunc = true; break; }
                const uint32_t bsz = le32(src + pos); pos += 4;
                if (bsz == 0) break;                                                     // EndMark
                const bool raw = (bsz & 0x80000000u) != 0; const uint32_t n = bsz & 0x7FFFFFFFu;
                if (n > bmax) return ALZ_E_FORMAT;
                if (n > len - pos) { trunc = true; break; }
                const size_t boff = pos; pos += n;
                if (flg & 16) {                                                          // block checksum over the stored bytes
                    if (pos + 4 > len) { trunc = true; break; }
                    if (le32(src + pos) != xxh32(src + boff, n, 0)) return ALZ_E_CHECKSUM;
                    pos += 4;
                }
                bl.push_back(Lz4Block{ boff, n, raw });
            }
            // The managed reader keeps ONE LzWindows for all blocks of a frame whatever the independence flag says
            // (LZ4.Frame.cs:120), so a frame that is flagged independent but whose blocks still reach into earlier output decodes
            // there.  Blocks go out as one batch only when a walk over their sequences (host, input only) shows that none does -- whatever the flag says the other
            // way round, too: the reference's own writer clears it (LZ4.Frame.cs:184) and compresses every block with a finder of its own (LZ4.cs:205), so the frames it
            // writes -- 16 MB in 64 KiB blocks: 256 blocks -- were decoded one launch after the other, each with the output so far as its history: 112 ms.
            bool indep = true;
            for (size_t i = 1; indep && i < bl.size(); i++) if (!bl[i].raw && lz4_block_reaches_back(src + bl[i].off, bl[i].len)) indep = false;
            if (indep) rc = run_independent(bl, bmax); else rc = run_sequential(bl, 0, frame_start, true);
            if (rc != ALZ_OK) return rc;
            if (st == ALZ_ST_OK && trunc) st = ALZ_ST_INPUT_TRUNCATED;
            if (st != ALZ_ST_OK) break;
            if ((flg & 8) && (uint64_t)(out - frame_start) != content) { st = ALZ_ST_OUTPUT_SIZE_MISMATCH; break; }   // LZ4.Frame.cs:152-155
            if (flg & 4) {                                                               // content checksum
                if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
                if (out > frame_start && (rc = alz_memcpy_d2h(ctx, dst + frame_start, (uint8_t*)d_dst.p + frame_start, out - frame_start)) != ALZ_OK) return rc;
                if (le32(src + pos) != xxh32(dst + frame_start, out - frame_start, 0)) return ALZ_E_CHECKSUM;
                pos += 4;
            }
        } else if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {                        // skippable
            if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            const uint32_t n = le32(src + pos); pos += 4;
            pos = (uint64_t)pos + n > len ? len : pos + n;
        } else { pos -= 4; break; }                                                      // not a frame: stop in front of it
    }
    if (out && (rc = alz_memcpy_d2h(ctx, dst, d_dst.p, out)) != ALZ_OK) return rc;
    if (dst_len) *dst_len = out;
    if (src_used) *src_used = pos;
    if (status) *status = st;
    return st == ALZ_ST_OK ? ALZ_OK : ALZ_E_STREAM;
}

// Compresses `n` bytes as independent blocks of `block` bytes in ONE GPU batch; block i's compressed bytes land at
// tmp + i * slot.  (LZ4: a new LzChainMatchFinder per block, LZ4.cs:205; Snappy: matchFinder.Reset() per chunk, :86)
// (the slots of the compressed blocks: memory nobody has touched -- a std::vector would write nb * slot zeros first, 21 MB for 16 MB of 64 KiB blocks, and only the compressed
// bytes are ever written and read)
struct BlockSlots {
    uint8_t* p = nullptr; size_t n = 0;
    BlockSlots() {}
    BlockSlots(const BlockSlots&) = delete; BlockSlots& operator=(const BlockSlots&) = delete;
    ~BlockSlots() { free(p); }
    bool resize(size_t bytes) { free(p); p = (uint8_t*)malloc(bytes ? bytes : 1); n = p ? bytes : 0; return p != nullptr; }
    uint8_t* data() const { return p; }
    size_t size() const { return n; }
};
int encode_blocks(alz_ctx* ctx, uint32_t fmt, const alz_settings* st, const uint8_t* src, size_t n, size_t block, BlockSlots& tmp, size_t& slot,
                  std::vector<alz_result>& rs) {
    const size_t nb = (n + block - 1) / block;
    slot = (block + block / 4 + 64 + 255) & ~(size_t)255;
    if (!tmp.resize(nb * slot + 64)) return ALZ_E_NOMEM;
    std::vector<alz_stream> ss(nb); rs.resize(nb);
    for (size_t i = 0; i < nb; i++) {
        memset(&ss[i], 0, sizeof(alz_stream));
        ss[i].src_off = i * block; ss[i].src_len = (uint32_t)(n - i * block < block ? n - i * block : block);
        ss[i].dst_off = i * slot; ss[i].dst_cap = (uint32_t)slot; ss[i].format = fmt;
    }
    return nb ? alz_encode_batch(ctx, nullptr, st, (uint32_t)nb, src, n, ss.data(), tmp.data(), tmp.size(), rs.data(), nullptr) : ALZ_OK;
}

// LZ4.Compress  LZ4.cs:113-160 (legacy) / CompressLZ4FrameHeader  LZ4.Frame.cs:176-229.  `Flags &= IsVersion1`
// (LZ4.Frame.cs:184) leaves only the version bit: the frame written never carries a content size or checksums.
int lz4_file_compress(alz_ctx* ctx, bool legacy, uint32_t block_size, const alz_settings* st, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* dst_len) {
    size_t o = 0;
    if (cap < 16) return ALZ_E_NOMEM;
    if (legacy) { wr32(dst, 0x184C2102u, false); o = 4; block_size = 0x800000u; }       // (int)BlockMaxSizes.Block4MB * 2
    else {
        uint8_t bdb;
        switch (block_size) { case 0: block_size = 0x400000; bdb = 0x70; break; case 0x10000: bdb = 0x40; break; case 0x40000: bdb = 0x50; break;
                              case 0x100000: bdb = 0x60; break; case 0x400000: bdb = 0x70; break; default: return ALZ_E_INVALID; }
        wr32(dst, 0x184D2204u, false); dst[4] = 0x40; dst[5] = bdb; dst[6] = (uint8_t)((xxh32(dst + 4, 2, 0) >> 8) & 0xFF); o = 7;
    }
    if (n && n % block_size != 0 && n % block_size < 5) return ALZ_E_INVALID;            // source.Slice(0, Length - 5) throws  LZ4.cs:208
    BlockSlots tmp; std::vector<alz_result> rs; size_t slot = 0;
    int rc = encode_blocks(ctx, ALZ_FMT_LZ4_BLOCK, st, src, n, block_size, tmp, slot, rs);
    if (rc != ALZ_OK) return rc;
    for (size_t i = 0; i < rs.size(); i++) {
        const size_t bl = n - i * block_size < block_size ? n - i * block_size : block_size;
        if (rs[i].status != ALZ_ST_OK) return rs[i].status == ALZ_ST_OUTPUT_CAPACITY ? ALZ_E_NOMEM : ALZ_E_INVALID;
        if (!legacy && rs[i].dst_len >= block_size) {                                    // buffer.Position >= (int)BlockSize: stored
            if (o + 4 + bl > cap) return ALZ_E_NOMEM;
            wr32(dst + o, (uint32_t)bl | 0x80000000u, false); memcpy(dst + o + 4, src + i * block_size, bl); o += 4 + bl;
        } else {
            if (o + 4 + rs[i].dst_len > cap) return ALZ_E_NOMEM;
            wr32(dst + o, rs[i].dst_len, false); memcpy(dst + o + 4, tmp.data() + i * slot, rs[i].dst_len); o += 4 + rs[i].dst_len;
        }
    }
    if (legacy) { if (o + 1 > cap) return ALZ_E_NOMEM; dst[o++] = 0xFF; }                // EOF flag
    else { if (o + 4 > cap) return ALZ_E_NOMEM; wr32(dst + o, 0, false); o += 4; }       // EndMark
    if (dst_len) *dst_len = o;
    return ALZ_OK;
}

inline uint32_t snappy_varint(const uint8_t* p, size_t len, size_t* used) {            // Snappy.ReadDecompressedSize  Snappy.cs:109-122
    uint32_t v = 0; int shift = 0; size_t i = 0; int b = 0x80;
    while ((b & 0x80) && i < len) { b = p[i++]; if (shift < 32) v |= (uint32_t)(b & 0x7F) << shift; shift += 7; }
    if (used) *used = i;
    return v;
}

// Snappy.Decompress  Formats/Common/Snappy.cs:39-69.  Every compressed chunk declares its size, so all chunks of a file
// decode as ONE GPU batch; CRCs are skipped as in the reference.  The managed decoder continues wherever a chunk's body
// stopped; here a chunk whose body does not end at its declared length is refused (ALZ_E_FORMAT).
int snappy_file_decompress(alz_ctx* ctx, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* dst_len, size_t* src_used, int32_t* status) {
    if (len < 10 || memcmp(src, kSnappyId, 10)) return ALZ_E_FORMAT;
    size_t pos = 10; uint64_t out = 0; int32_t st = ALZ_ST_OK;
    std::vector<alz_stream> ss; std::vector<uint32_t> clen;
    struct Raw { size_t off; uint32_t n; uint64_t out; }; std::vector<Raw> raws;
    while (pos < len) {
        if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
        const uint32_t type = src[pos], cl = (uint32_t)src[pos + 1] | ((uint32_t)src[pos + 2] << 8) | ((uint32_t)src[pos + 3] << 16); pos += 4;
        if (type == 0) {
            if (pos + 4 > len) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            const size_t body = pos + 4;
            const uint32_t size = snappy_varint(src + body, len - body, nullptr);
            alz_stream s; memset(&s, 0, sizeof(s));
            s.src_off = body; s.src_len = clamp32(len - body); s.dst_off = out < cap ? out : cap;
            s.dst_cap = clamp32(out < cap ? (cap - out < size ? cap - out : size) : 0); s.format = ALZ_FMT_SNAPPY_RAW;
            ss.push_back(s); clen.push_back(cl);
            out += size; pos = (uint64_t)pos + cl > len ? len : pos + cl;
        } else if (type == 1) {
            if (pos + 4 > len || cl < 4) { st = ALZ_ST_INPUT_TRUNCATED; break; }
            uint32_t n = cl - 4; if (n > len - pos - 4) n = (uint32_t)(len - pos - 4);    // SubStream.CopyTo copies what is there
            raws.push_back(Raw{ pos + 4, n, out }); out += n; pos += 4 + n;
        } else {
            if (type >= 0x02 && type <= 0x7F) return ALZ_E_FORMAT;                       // reserved unskippable chunk  Snappy.cs:61-62
            pos = (uint64_t)pos + cl > len ? len : pos + cl;
        }
    }
    std::vector<alz_result> rs(ss.size());
    if (!ss.empty()) { int rc = alz_decode_batch(ctx, nullptr, (uint32_t)ss.size(), src, len, ss.data(), dst, cap, rs.data()); if (rc != ALZ_OK) return rc; }
    // first failing chunk in file order decides status and length
    size_t produced = (size_t)(out < cap ? out : cap); int32_t fst = ALZ_ST_OK; uint64_t fail_at = ~0ull;
    for (size_t i = 0; i < ss.size(); i++) {
        const uint32_t size = snappy_varint(src + ss[i].src_off, len - ss[i].src_off, nullptr);
        int32_t cs = rs[i].status;
        if (cs == ALZ_ST_OK && rs[i].dst_len < size) cs = ALZ_ST_OUTPUT_CAPACITY;        // the declared size did not fit dst
        if (cs == ALZ_ST_OK && (uint64_t)rs[i].src_used + 4 != clen[i]) return ALZ_E_FORMAT;
        if (cs != ALZ_ST_OK) { fst = cs; fail_at = ss[i].dst_off; produced = (size_t)(ss[i].dst_off + rs[i].dst_len); break; }
    }
    for (const Raw& r : raws) {
        if (r.out >= fail_at) break;
        if (r.out + r.n > cap) { if (fst == ALZ_ST_OK || r.out < fail_at) { fst = ALZ_ST_OUTPUT_CAPACITY; produced = (size_t)r.out; } break; }
        memcpy(dst + r.out, src + r.off, r.n);
    }
    if (fst == ALZ_ST_OK && st != ALZ_ST_OK) fst = st;
    if (dst_len) *dst_len = produced;
    if (src_used) *src_used = pos;
    if (status) *status = fst;
    return fst == ALZ_ST_OK ? ALZ_OK : ALZ_E_STREAM;
}

// Snappy.Compress  Formats/Common/Snappy.cs:71-107
int snappy_file_compress(alz_ctx* ctx, const alz_settings* st, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* dst_len) {
    if (cap < 10) return ALZ_E_NOMEM;
    memcpy(dst, kSnappyId, 10);
    size_t o = 10;
    BlockSlots tmp; std::vector<alz_result> rs; size_t slot = 0;
    int rc = encode_blocks(ctx, ALZ_FMT_SNAPPY_RAW, st, src, n, 0x10000, tmp, slot, rs);
    if (rc != ALZ_OK) return rc;
    for (size_t i = 0; i < rs.size(); i++) {
        const size_t cs = n - i * 0x10000 < 0x10000 ? n - i * 0x10000 : 0x10000;
        if (rs[i].status != ALZ_ST_OK) return ALZ_E_INVALID;
        const uint32_t crc = snappy_crc_mask(crc32c(src + i * 0x10000, cs));
        const bool stored = rs[i].dst_len >= cs;                                         // buffer.Length >= chunkSize
        const size_t body = stored ? cs : rs[i].dst_len;
        if (o + 8 + body > cap) return ALZ_E_NOMEM;
        dst[o] = stored ? 1 : 0;
        dst[o + 1] = (uint8_t)(body + 4); dst[o + 2] = (uint8_t)((body + 4) >> 8); dst[o + 3] = (uint8_t)((body + 4) >> 16);
        wr32(dst + o + 4, crc, false);
        memcpy(dst + o + 8, stored ? src + i * 0x10000 : tmp.data() + i * slot, body);
        o += 8 + body;
    }
    if (dst_len) *dst_len = o;
    return ALZ_OK;
}

}  // namespace

// RefPack.InternalReadHeader  RefPack.cs:77-102: flags + 0xFB + size [+ compressed size], optionally behind a u32 compressed size
// (version 2).  Returns the header length, or an ALZ_E_* code.
static int refpack_header(const uint8_t* src, size_t len, uint32_t* size) {
    size_t pos = 0;
    if (len < 2) return ALZ_E_FORMAT;
    if (src[1] != 0xFB) {                                            // not version 1 / 3: a pre-header must follow  :81-90
        if (len < 6 || src[4] != 0x10 || src[5] != 0xFB) return ALZ_E_FORMAT;
        pos = 4;
    }
    const uint8_t flag = src[pos]; pos += 2;
    if (!(flag & 0x10)) return ALZ_E_UNSUPPORTED;                    // NotSupportedException("No supported Flag")  :92-93
    const bool wide = (flag & 0x80) != 0; const size_t n = wide ? 4 : 3;
    if (len < pos + n) return ALZ_E_FORMAT;
    *size = wide ? be32(src + pos) : (be32(src + pos - 1) & 0xFFFFFFu);
    pos += n;
    if (flag & 1) pos += n;                                          // StoresCompressedSize
    return (int)pos;
}

// FastLZ.Validate  FastLZ.cs:246-291 (sic: only streams whose first byte is below 0x20, i.e. level 1, pass)
static bool fastlz_validate(const uint8_t* s, size_t n) {
    size_t pos = 0;
    int ctrl = pos < n ? s[pos++] : -1;
    const int level = (ctrl >> 5) + 1;
    if (level != 0 && level != 1) return false;
    int i = 3; long buffer = 0;
    while (ctrl != -1) {
        if (ctrl >= 32) {
            int length = (ctrl >> 5) - 1, distance = (ctrl & 31) << 8;
            if (length == 6) length += pos < n ? s[pos++] : -1;
            ctrl = pos < n ? s[pos++] : -1;
            distance |= ctrl;
            if (ctrl == -1 || length < 0) return false;             // end of stream
            if (distance + 1 > buffer) return false;                // reaches before the start of the window
            if (i-- == 0) return true;
            buffer += length + 3;
        } else {
            ctrl++;
            buffer += ctrl;
            if (pos + (size_t)ctrl > n) return false;
            pos += (size_t)ctrl;
        }
        if (pos >= n) return i < 3;
        ctrl = s[pos++];
    }
    return false;
}

// LZ00.StreamTransformer  LZ00.cs:128-141: body byte i is XORed with element i + 1 of the keystream of `key`.
// GenerateNextKey's shift/add chain is key * 1103515245 + 12345 (3 -> 95 -> 3041 -> 389247, then x 63 x 15 x 3).  A header-level
// transform like the frame checksums: it runs on the host, on the compressed bytes, before / after the body kernels.
static void lz00_keystream(uint8_t* p, size_t n, uint32_t key) {
    for (size_t i = 0; i < n; i++) {
        key = key * 1103515245u + 12345u;
        const uint32_t t = (key >> 16) & 0x7FFFu;
        p[i] ^= (uint8_t)(((t << 8) - t) >> 15);
    }
}

extern "C" {

// IProvidesDecompressedSize.GetDecompressedSize  Interfaces/IProvidesDecompressedSize.cs:20
int alz_container_decompressed_size(uint32_t container, const alz_container_options* opt, const uint8_t* src, size_t len, uint32_t* size_out) {
    if (!src || !size_out) return ALZ_E_INVALID;
    const bool big = opt ? opt->big_endian != 0 : true;
    switch (container) {
    case ALZ_C_LZSS: if (len < 8 || memcmp(src, "LZSS", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 4); return ALZ_OK;        // LZSS.cs:45-50
    case ALZ_C_LZ10: return nin_header(src, len, 0x10, size_out) < 0 ? ALZ_E_FORMAT : ALZ_OK;                                  // LZ10.cs:44-57
    case ALZ_C_LZ11: return nin_header(src, len, 0x11, size_out) < 0 ? ALZ_E_FORMAT : ALZ_OK;                                  // LZ11.cs:40-53
    case ALZ_C_LZ40: return nin_header(src, len, 0x40, size_out) < 0 ? ALZ_E_FORMAT : ALZ_OK;                                  // LZ40.cs:40-52
    case ALZ_C_LZHUDSON: if (len < 4) return ALZ_E_FORMAT; *size_out = be32(src); return ALZ_OK;                                        // LZHudson.cs:30-31
    case ALZ_C_LZ00: if (len < 52 || memcmp(src, "LZ00", 4)) return ALZ_E_FORMAT; *size_out = le32(src + 48); return ALZ_OK;               // LZ00.cs:31-37
    case ALZ_C_CNX2: if (len < 16 || memcmp(src, "CNX\x02", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 12); return ALZ_OK;             // CNX2.cs:36-42
    case ALZ_C_CLZ0: if (len < 16 || memcmp(src, "CLZ\0", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 12); return ALZ_OK;               // CLZ0.cs:33-39
    case ALZ_C_CNS: if (len < 12 || memcmp(src, "@CNS", 4)) return ALZ_E_FORMAT; *size_out = le32(src + 8); return ALZ_OK;                  // CNS.cs:36-42
    case ALZ_C_LZ02: if (len < 4 || (src[0] != 1 && src[0] != 2)) return ALZ_E_FORMAT; *size_out = be32(src) & 0xFFFFFFu; return ALZ_OK;   // LZ02.cs:49-58
    case ALZ_C_REFPACK: { const int h = refpack_header(src, len, size_out); return h < 0 ? h : ALZ_OK; }                                     // RefPack.cs:56-62
    case ALZ_C_LZSHREK: if (len < 8) return ALZ_E_FORMAT; *size_out = le32(src + 4); return ALZ_OK;                                           // LZShrek.cs:28-33
    case ALZ_C_HIG: if (len < 0x40 || memcmp(src, "HIG!", 4)) return ALZ_E_FORMAT; *size_out = le32(src + 0x3C); return ALZ_OK;              // HIG.cs:39-45
    case ALZ_C_WFLZ: if (len < 12 || memcmp(src, "WFLZ", 4)) return ALZ_E_FORMAT; *size_out = rd32(src + 8, opt && opt->big_endian); return ALZ_OK;   // WFLZ.cs:41-48 (FormatByteOrder defaults to little)
    case ALZ_C_BLZ: {                                                                                                                      // BLZ.cs:32-41
        if (len < 8 || src[len - 5] < 8) return ALZ_E_FORMAT;                                                                              // "Invalid BLZ header."
        *size_out = le32(src + len - 4) + (le32(src + len - 8) & 0xFFFFFFu); return ALZ_OK;
    }
    case ALZ_C_SMSR00: if (len < 12 || memcmp(src, "SMSR00", 6)) return ALZ_E_FORMAT; *size_out = be32(src + 8); return ALZ_OK;          // SMSR00.cs:33-39
    case ALZ_C_LZ60: return nin_header(src, len, 0x60, size_out) < 0 ? ALZ_E_FORMAT : ALZ_OK;                                  // LZ60.cs:29-41
    case ALZ_C_YAZ0: if (len < 8 || memcmp(src, "Yaz0", 4)) return ALZ_E_FORMAT; *size_out = rd32(src + 4, big); return ALZ_OK;   // Yaz0.cs:50-55
    case ALZ_C_YAY0: if (len < 8 || memcmp(src, "Yay0", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 4); return ALZ_OK;        // Yay0.cs:41-47 (always Endian.Big)
    case ALZ_C_MIO0: if (len < 8 || memcmp(src, "MIO0", 4)) return ALZ_E_FORMAT; *size_out = rd32(src + 4, big); return ALZ_OK;   // MIO0.cs:41-48
    case ALZ_C_GCLZ: if (len < 4 || memcmp(src, "GCLZ", 4)) return ALZ_E_FORMAT; return alz_container_decompressed_size(ALZ_C_LZ10, opt, src + 4, len - 4, size_out);   // GCLZ.cs:32-37
    case ALZ_C_CXLZ: if (len < 4 || memcmp(src, "CXLZ", 4)) return ALZ_E_FORMAT; return alz_container_decompressed_size(ALZ_C_LZ10, opt, src + 4, len - 4, size_out);
    case ALZ_C_LZ_3DS: if (len < 8 || memcmp(src, "3DS-LZ\r\n", 8)) return ALZ_E_FORMAT; return alz_container_decompressed_size(ALZ_C_LZ10, opt, src + 8, len - 8, size_out);
    case ALZ_C_COMP: if (len < 4 || memcmp(src, "COMP", 4)) return ALZ_E_FORMAT; return alz_container_decompressed_size(ALZ_C_LZ11, opt, src + 4, len - 4, size_out);
    case ALZ_C_YAZ1: if (len < 8 || memcmp(src, "Yaz1", 4)) return ALZ_E_FORMAT; *size_out = rd32(src + 4, big); return ALZ_OK;
    case ALZ_C_AKLZ: if (len < 16 || memcmp(src, kAklzMagic, 12)) return ALZ_E_FORMAT; *size_out = be32(src + 12); return ALZ_OK;          // AKLZ.cs:33-38
    case ALZ_C_LZ01: if (len < 12 || memcmp(src, "LZ01", 4)) return ALZ_E_FORMAT; *size_out = le32(src + 8); return ALZ_OK;                // LZ01.cs:37-43
    case ALZ_C_LZSEGA: if (len < 8) return ALZ_E_FORMAT; *size_out = le32(src + 4); return ALZ_OK;                                        // LZSega.cs:41-46
    case ALZ_C_LEVEL5LZSS: if (len < 16 || memcmp(src, "SSZL", 4)) return ALZ_E_FORMAT; *size_out = le32(src + 12); return ALZ_OK;        // Level5LZSS.cs:33-39
    case ALZ_C_LZON: if (len < 12 || memcmp(src, kLzonMagic, 8)) return ALZ_E_FORMAT; *size_out = be32(src + 8); return ALZ_OK;            // LZOn.cs:33-38
    case ALZ_C_MDB4: if (len < 12 || memcmp(src, "MDB4", 4)) return ALZ_E_FORMAT; *size_out = le32(src + 8); return ALZ_OK;               // MDB4.cs:25-31
    case ALZ_C_FCMP: case ALZ_C_IECP: case ALZ_C_SDPC:                                                                                     // FCMP.cs:28-33
        if (len < 8 || memcmp(src, container == ALZ_C_FCMP ? "FCMP" : container == ALZ_C_IECP ? "IECP" : "SDPC", 4)) return ALZ_E_FORMAT;
        *size_out = le32(src + 4); return ALZ_OK;
    case ALZ_C_GCZ: if (len < 4) return ALZ_E_FORMAT; *size_out = le32(src); return ALZ_OK;                                                // GCZ.cs:30
    case ALZ_C_ECD:                                                                                                                        // ECD.cs:34-43
        if (len < 16 || memcmp(src, "ECD", 3)) return ALZ_E_FORMAT;
        *size_out = (uint64_t)be32(src + 8) + 0x10 > len ? 0u : be32(src + 12); return ALZ_OK;
    case ALZ_C_LZ4_FRAME:   // not an IProvidesDecompressedSize in the reference; offered where the descriptor carries ContentSize
        if (len < 15 || le32(src) != 0x184D2204u || !(src[4] & 8) || le32(src + 10) != 0) return ALZ_E_UNSUPPORTED;
        *size_out = le32(src + 6); return ALZ_OK;
    case ALZ_C_SNAPPY: {    // same: the sum of the chunks' declared sizes
        if (len < 10 || memcmp(src, kSnappyId, 10)) return ALZ_E_FORMAT;
        size_t pos = 10; uint64_t total = 0;
        while (pos + 4 <= len) {
            const uint32_t type = src[pos], cl = (uint32_t)src[pos + 1] | ((uint32_t)src[pos + 2] << 8) | ((uint32_t)src[pos + 3] << 16); pos += 4;
            if (type == 0) total += pos + 4 <= len ? snappy_varint(src + pos + 4, len - pos - 4, nullptr) : 0;
            else if (type == 1) total += cl >= 4 ? cl - 4 : 0;
            pos += cl;
        }
        *size_out = (uint32_t)total; return ALZ_OK;
    }
    case ALZ_C_LZ77: {                                                                                                                     // LZ77.cs:45-54
        if (len < 8 || memcmp(src, "LZ77", 4)) return ALZ_E_FORMAT;
        uint32_t s = (uint32_t)src[5] | ((uint32_t)src[6] << 8) | ((uint32_t)src[7] << 16);
        if (s == 0) { if (len < 12) return ALZ_E_FORMAT; s = le32(src + 8); }
        *size_out = s; return ALZ_OK;
    }
    case ALZ_C_LEVEL5: if (len < 5) return ALZ_E_FORMAT; *size_out = src[4] == 0x78 ? le32(src) : le32(src) >> 3; return ALZ_OK;          // Level5.cs:55-60
    default: return ALZ_E_UNSUPPORTED;   // PRS / LZO / LZ4 / Snappy do not implement IProvidesDecompressedSize
    }
}

// IFormatInfoProvider.IsMatch
int alz_container_is_match(uint32_t container, const uint8_t* src, size_t len) {
    if (!src) return 0;
    switch (container) {
    case ALZ_C_LZSS: return len > 0x10 && !memcmp(src, "LZSS", 4);                        // LZSS.cs:41-42
    case ALZ_C_LZ10: return len > 0x8 && nin_validate(src, len, false);                     // LZ10.cs:40-41
    case ALZ_C_LZ11: return len > 0x8 && nin_validate(src, len, true);                      // LZ11.cs:35-36
    case ALZ_C_YAZ0: return len > 0x10 && !memcmp(src, "Yaz0", 4);                        // Yaz0.cs:46-47
    case ALZ_C_YAY0: return len > 0x10 && !memcmp(src, "Yay0", 4);
    case ALZ_C_MIO0: return len > 0x10 && !memcmp(src, "MIO0", 4);
    case ALZ_C_PRS: return len > 0x4 && prs_byte_order(src, len) != 0;                      // PRS.cs:33-34
    case ALZ_C_LZO: { if (len == 0) return 0; int f = src[0]; return (f > 11 && f < 0x20) || f < 0x10; }   // LZO.cs:33-39 (no extension given)
    case ALZ_C_LZ4_LEGACY: return len > 0x10 && le32(src) == 0x184C2102u;                   // LZ4Legacy.cs:28-29
    case ALZ_C_LZ4_FRAME: return len > 0x10 && lz4_magic_defined(le32(src));                // LZ4.cs:38-39
    case ALZ_C_SNAPPY: return len > 0x10 && !memcmp(src, kSnappyId, 10);                    // Snappy.cs:36-37
    case ALZ_C_GCLZ: return len > 0x8 && !memcmp(src, "GCLZ", 4) && alz_container_is_match(ALZ_C_LZ10, src + 4, len - 4);
    case ALZ_C_CXLZ: return len > 0x8 && !memcmp(src, "CXLZ", 4) && alz_container_is_match(ALZ_C_LZ10, src + 4, len - 4);
    case ALZ_C_LZ_3DS: return len > 0x10 && !memcmp(src, "3DS-LZ\r\n", 8);
    case ALZ_C_COMP: return len > 0x8 && !memcmp(src, "COMP", 4) && alz_container_is_match(ALZ_C_LZ11, src + 4, len - 4);
    case ALZ_C_YAZ1: return len > 0x10 && !memcmp(src, "Yaz1", 4);
    case ALZ_C_AKLZ: return len > 0x10 && !memcmp(src, kAklzMagic, 12);
    case ALZ_C_LZ01: return len > 0x10 && !memcmp(src, "LZ01", 4);
    case ALZ_C_LZSEGA: { if (len < 0x12) return 0; const uint32_t cs = le32(src), ds = le32(src + 4); return (cs == len - 8 || cs == len) && ds != 0 && (src[8] & 1) == 1; }   // LZSega.cs:27-38
    case ALZ_C_LEVEL5LZSS: return len > 0x10 && !memcmp(src, "SSZL", 4) && le32(src + 4) == 0;
    case ALZ_C_LZON: return len > 0x10 && !memcmp(src, kLzonMagic, 8);
    case ALZ_C_LZ40: case ALZ_C_LZ60:                                                       // "no distinct header, recognition is inaccurate"  LZ40.cs:36-38
        return len > 0x8 && src[0] == (container == ALZ_C_LZ40 ? 0x40 : 0x60) && ((src[1] | src[2] | src[3]) != 0 || le32(src + 4) != 0);
    case ALZ_C_LZHUDSON: return len > 0x8 && le32(src) != 0;                                // (+ the file extension when one is given)  LZHudson.cs:27-28
    case ALZ_C_SMSR00: return len > 0x10 && !memcmp(src, "SMSR00", 6);                      // SMSR00.cs:30-31
    case ALZ_C_LZ00: return len > 0x40 && !memcmp(src, "LZ00", 4);                           // LZ00.cs:36-37
    case ALZ_C_FASTLZ: return len > 0x4 && fastlz_validate(src, len);                        // FastLZ.cs:34-35
    case ALZ_C_CNX2: return len > 0x10 && !memcmp(src, "CNX\x02", 4);                        // CNX2.cs:33-34
    case ALZ_C_CLZ0: return len > 0x10 && !memcmp(src, "CLZ\0", 4);                          // CLZ0.cs:30-31
    case ALZ_C_CNS: return len > 0x10 && !memcmp(src, "@CNS", 4);                            // CNS.cs:33-34
    case ALZ_C_WFLZ: return len > 0x10 && !memcmp(src, "WFLZ", 4);                           // WFLZ.cs:38-39
    case ALZ_C_HIG: return len > 0x8 && !memcmp(src, "HIG!", 4);                             // HIG.cs:36-37
    case ALZ_C_LZSHREK: return len > 0x10 && le32(src) == 0x10 && le32(src + 4) != 0 && le32(src + 8) == len - 0x10 && le32(src + 12) == 0;   // LZShrek.cs:25-26
    case ALZ_C_REFPACK:                                                                      // RefPack.cs:40-53 (versions 1 / 3, or version 2 behind its pre-header)
        return len > 0x8 && (((src[0] & 0x2E) == 0 && (src[0] & 0x10) && src[1] == 0xFB && (be32(src + 1) & 0xFFFFFFu) != 0) ||
                             (le32(src) != 0 && src[4] == 0x10 && src[5] == 0xFB && (be32(src + 5) & 0xFFFFFFu) != 0));
    case ALZ_C_LZ02: return len > 0x8 && (src[0] == 1 || src[0] == 2) && (be32(src) & 0xFFFFFFu) != 0 && (src[4] & 0x80) == 0;   // LZ02.cs:28-35 (no extension given)
    case ALZ_C_BLZ: return len >= 8 && (le32(src + len - 8) & 0xFFFFFFu) == len && src[len - 5] >= 8;   // BLZ.cs:28-30 (the footer spans the whole stream)
    case ALZ_C_MDB4: return len > 0x10 && !memcmp(src, "MDB4", 4);
    case ALZ_C_FCMP: return len > 0x10 && !memcmp(src, "FCMP", 4);
    case ALZ_C_IECP: return len > 0x10 && !memcmp(src, "IECP", 4);
    case ALZ_C_GCZ: return 0;                                                               // extension ".gcz" required (GCZ.cs:23-28): no file name here
    case ALZ_C_ECD: return len > 0x10 && !memcmp(src, "ECD", 3) && (uint64_t)be32(src + 8) + 0x10 <= len && be32(src + 12) != 0;   // ECD.cs:29-30
    case ALZ_C_SDPC: return len > 0x10 && !memcmp(src, "SDPC", 4) && le32(src + 4) != 0;
    case ALZ_C_LZ77: return len > 0x8 && !memcmp(src, "LZ77", 4) && (src[4] == 0x10 || src[4] == 0x11 || src[4] == 0x24 || src[4] == 0x28 || src[4] == 0x30 || src[4] == 0xF7);
    default: return 0;
    }
}

// ICompressionDecoder.Decompress(Stream source, Stream destination)  Interfaces/ICompressionDecoder.cs:24
int alz_container_decompress(alz_ctx* ctx, uint32_t container, const alz_container_options* opt, const uint8_t* src, size_t len,
                             uint8_t* dst, size_t dst_cap, size_t* dst_len, size_t* src_used, int32_t* status) {
    if (!ctx || !src || (!dst && dst_cap)) return ALZ_E_INVALID;
    const bool big = opt ? opt->big_endian != 0 : true;
    const alz_lz_properties* lz = opt ? &opt->lz : nullptr;
    alz_result r; memset(&r, 0, sizeof(r));
    uint32_t size = 0; size_t hdr = 0; int rc = ALZ_OK;
    switch (container) {
    case ALZ_C_LZSS:                                                                        // LZSS.cs:53-69
        if (len < 16 || memcmp(src, "LZSS", 4)) return ALZ_E_FORMAT;
        size = be32(src + 4); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_LZSS, lz, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZ10: case ALZ_C_LZ11: {                                                     // LZ10.cs:60-64, LZ11.cs:56-60
        int h = nin_header(src, len, container == ALZ_C_LZ10 ? 0x10 : 0x11, &size);
        if (h < 0) return ALZ_E_FORMAT;
        hdr = (size_t)h;
        rc = run_body(ctx, container == ALZ_C_LZ10 ? ALZ_FMT_LZ10 : ALZ_FMT_LZ11, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_YAZ0:                                                                        // Yaz0.cs:58-79
        if (len < 16 || memcmp(src, "Yaz0", 4)) return ALZ_E_FORMAT;
        size = rd32(src + 4, big); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_YAZ0, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        if (rc == ALZ_OK && r.status != ALZ_ST_OK)                                          // catch (Exception): try the other byte order
            rc = run_body(ctx, ALZ_FMT_YAZ0, nullptr, src + hdr, len - hdr, bswap32(size), 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_YAY0: case ALZ_C_MIO0: {                                                     // Yay0.cs:50-60, MIO0.cs:51-61
        if (len < 16 || memcmp(src, container == ALZ_C_YAY0 ? "Yay0" : "MIO0", 4)) return ALZ_E_FORMAT;
        // Stream.DetectByteOrder<uint>(3) lives in the unvendored AuroraLib.Core (parity unpinned): the caller's
        // FormatByteOrder decides here.
        size = rd32(src + 4, big); const uint32_t cp = rd32(src + 8, big), up = rd32(src + 12, big); hdr = 16;
        rc = run_body(ctx, container == ALZ_C_YAY0 ? ALZ_FMT_YAY0 : ALZ_FMT_MIO0, nullptr, src + hdr, len - hdr, size, cp - 0x10, up - 0x10, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_PRS: {                                                                       // PRS.cs:42-57
        const bool first_big = prs_byte_order(src, len) == 2;                               // GetByteOrder(source) == Endian.Big ? Big : Little
        rc = run_body(ctx, first_big ? ALZ_FMT_PRS_BE : ALZ_FMT_PRS_LE, nullptr, src, len, 0, 0, 0, dst, dst_cap, &r);
        if (rc == ALZ_OK && r.status != ALZ_ST_OK)
            rc = run_body(ctx, first_big ? ALZ_FMT_PRS_LE : ALZ_FMT_PRS_BE, nullptr, src, len, 0, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_LZO:                                                                         // LZO.cs:42-43
        rc = run_body(ctx, ALZ_FMT_LZO, nullptr, src, len, 0, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZ4_LEGACY: case ALZ_C_LZ4_FRAME:                                            // LZ4Legacy.Decompress -> LZ4.Decompress
        return lz4_file_decompress(ctx, src, len, dst, dst_cap, dst_len, src_used, status);
    case ALZ_C_SNAPPY:
        return snappy_file_decompress(ctx, src, len, dst, dst_cap, dst_len, src_used, status);
    case ALZ_C_GCLZ: case ALZ_C_CXLZ: case ALZ_C_LZ_3DS: case ALZ_C_COMP: {                 // magic + inner file  GCLZ.cs:47-51
        const char* magic = container == ALZ_C_GCLZ ? "GCLZ" : container == ALZ_C_CXLZ ? "CXLZ" : container == ALZ_C_COMP ? "COMP" : "3DS-LZ\r\n";
        const size_t ml = container == ALZ_C_LZ_3DS ? 8 : 4;
        if (len < ml || memcmp(src, magic, ml)) return ALZ_E_FORMAT;
        size_t used = 0;
        rc = alz_container_decompress(ctx, container == ALZ_C_COMP ? ALZ_C_LZ11 : ALZ_C_LZ10, opt, src + ml, len - ml, dst, dst_cap, dst_len, &used, status);
        if (src_used) *src_used = ml + used;
        return rc;
    }
    case ALZ_C_YAZ1:                                                                        // Yaz1.cs: Yaz0 with another magic
        if (len < 16 || memcmp(src, "Yaz1", 4)) return ALZ_E_FORMAT;
        size = rd32(src + 4, big); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_YAZ0, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        if (rc == ALZ_OK && r.status != ALZ_ST_OK) rc = run_body(ctx, ALZ_FMT_YAZ0, nullptr, src + hdr, len - hdr, bswap32(size), 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_AKLZ:                                                                        // AKLZ.cs:41-46
        if (len < 16 || memcmp(src, kAklzMagic, 12)) return ALZ_E_FORMAT;
        size = be32(src + 12); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_LZSS, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZ01:                                                                        // LZ01.cs:47-62
        if (len < 16 || memcmp(src, "LZ01", 4)) return ALZ_E_FORMAT;
        size = le32(src + 8); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_LZSS, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZSEGA:                                                                      // LZSega.cs:49-54
        if (len < 8) return ALZ_E_FORMAT;
        size = le32(src + 4); hdr = 8;
        rc = run_body(ctx, ALZ_FMT_LZSS, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LEVEL5LZSS:                                                                  // Level5LZSS.cs:42-59
        if (len < 16 || memcmp(src, "SSZL", 4)) return ALZ_E_FORMAT;
        size = le32(src + 12); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_LZSS, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZON:                                                                        // LZOn.cs:41-60
        if (len < 16 || memcmp(src, kLzonMagic, 8)) return ALZ_E_FORMAT;
        size = be32(src + 8); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_LZO, nullptr, src + hdr, len - hdr, 0, 0, 0, dst, dst_cap, &r);
        if (rc == ALZ_OK && r.status == ALZ_ST_OK && r.dst_len != size) r.status = ALZ_ST_OUTPUT_SIZE_MISMATCH;   // DecompressedSizeException.ThrowIfMismatch
        break;
    case ALZ_C_LZHUDSON:                                                                    // LZHudson.cs:33-37
        if (len < 4) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = be32(src); hdr = 4;
        rc = run_body(ctx, ALZ_FMT_LZHUDSON, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_SMSR00:                                                                      // SMSR00.cs:41-48
        if (len < 6 || memcmp(src, "SMSR00", 6)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = be32(src + 8); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_SMSR00, nullptr, src + hdr, len - hdr, size, be32(src + 12) - 16u, 0, dst, dst_cap, &r);   // uncompressedDataPointer - source.Position
        break;
    case ALZ_C_HIG: {                                                                       // HIG.cs:47-80
        if (len < 4 || memcmp(src, "HIG!", 4)) return ALZ_E_FORMAT;
        if (len < 0x40) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        uint32_t start = le32(src + 4); const uint32_t ver = le32(src + 0x38);
        size = le32(src + 0x3C);
        if (ver == 5 || ver == 6) start = 0xC0;                                             // extension header 0x40-0xC0: compressed size + path
        if (start > len) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        hdr = start;
        rc = run_body(ctx, ALZ_FMT_HIG, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_LZSHREK: {                                                                   // LZShrek.cs:35-53
        if (len < 12) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        const uint32_t offset = le32(src), csz = le32(src + 8);
        size = le32(src + 4);
        if (offset > len || csz > len - offset) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }   // Seek + ReadExactly
        rc = run_body(ctx, ALZ_FMT_LZSHREK, nullptr, src + offset, csz, size, 0, 0, dst, dst_cap, &r);
        r.src_used = csz; hdr = offset;
        break;
    }
    case ALZ_C_WFLZ: {                                                                      // WFLZ.cs:50-86
        if (len < 4 || memcmp(src, "WFLZ", 4)) return ALZ_E_FORMAT;
        if (len < 12) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        const bool wbig = opt && opt->big_endian;                                           // FormatByteOrder: little unless the caller says otherwise
        uint32_t csz = rd32(src + 4, wbig);
        size = rd32(src + 8, wbig); hdr = 12;
        if (csz > len - hdr) csz = clamp32(len - hdr);                                      // Stream.Read returns what is there; the body then runs off its end
        rc = run_body(ctx, wbig ? ALZ_FMT_WFLZ_BE : ALZ_FMT_WFLZ, nullptr, src + hdr, csz, 0, 0, 0, dst, dst_cap, &r);
        if (rc == ALZ_OK && r.status == ALZ_ST_OK && r.dst_len != size) r.status = ALZ_ST_OUTPUT_SIZE_MISMATCH;   // '!='  :82-85
        r.src_used = csz;                                                                   // the whole compressed span is read up front
        break;
    }
    case ALZ_C_REFPACK: {                                                                   // RefPack.cs:64-75
        const int h = refpack_header(src, len, &size);
        if (h < 0) return h;
        hdr = (size_t)h;
        if (len < hdr) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        rc = run_body(ctx, ALZ_FMT_REFPACK, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_LZ02:                                                                        // LZ02.cs:60-64
        if (len < 1 || (src[0] != 1 && src[0] != 2)) return ALZ_E_FORMAT;
        if (len < 4) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = be32(src) & 0xFFFFFFu; hdr = 4;
        rc = run_body(ctx, ALZ_FMT_LZ02, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_CNS:                                                                         // CNS.cs:44-55
        if (len < 4 || memcmp(src, "@CNS", 4)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = le32(src + 8); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_CNS, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_CLZ0:                                                                        // CLZ0.cs:41-51
        if (len < 4 || memcmp(src, "CLZ\0", 4)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = be32(src + 12); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_CLZ0, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_BLZ: {                                                                       // BLZ.cs:43-69
        if (len < 8) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        const uint32_t csz = le32(src + len - 8) & 0xFFFFFFu, hp = src[len - 5];
        if (hp < 8) return ALZ_E_FORMAT;                                                    // "Invalid BLZ header."
        size = le32(src + len - 4) + csz;
        if (csz > len || hp > csz) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        const uint32_t code = csz - hp;
        if ((uint64_t)size > dst_cap) { r.status = ALZ_ST_OUTPUT_CAPACITY; break; }
        // the managed code walks both spans from their ends (BLZ.cs:99-101): the body kernel sees the code section
        // reversed and leaves the output reversed -- two host-side passes around the match-copy work, like the checksums
        std::vector<uint8_t> rev(src + (len - csz), src + (len - csz) + code);
        std::reverse(rev.begin(), rev.end());
        rc = run_body(ctx, ALZ_FMT_BLZ, nullptr, rev.data(), code, size, 0, 0, dst, size, &r);
        if (rc == ALZ_OK && r.status == ALZ_ST_OK) std::reverse(dst, dst + size);
        else r.dst_len = 0;                                                                 // nothing reaches the destination when the body throws (:62-64)
        r.src_used = clamp32(len); hdr = 0;
        break;
    }
    case ALZ_C_CNX2:                                                                        // CNX2.cs:45-62
        if (len < 4 || memcmp(src, "CNX\x02", 4)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = be32(src + 12); hdr = 16;
        rc = run_body(ctx, ALZ_FMT_CNX2, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_FASTLZ:                                                                      // FastLZ.cs:40-52: the rest of the stream is the body
        rc = run_body(ctx, ALZ_FMT_FASTLZ, nullptr, src, len, 0, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_LZ00: {                                                                      // LZ00.cs:40-60
        if (len < 4 || memcmp(src, "LZ00", 4)) return ALZ_E_FORMAT;
        if (len < 64) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = le32(src + 48); hdr = 64;
        std::vector<uint8_t> plain(src + hdr, src + len);                                   // the body behind the StreamTransformer
        lz00_keystream(plain.data(), plain.size(), le32(src + 52));
        rc = run_body(ctx, ALZ_FMT_LZSS, nullptr, plain.data(), plain.size(), size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_LZ40: case ALZ_C_LZ60: {                                                     // LZ40.cs:54-61, LZ60.cs:43-47
        int h = nin_header(src, len, container == ALZ_C_LZ40 ? 0x40 : 0x60, &size);
        if (h < 0) return ALZ_E_FORMAT;
        hdr = (size_t)h;
        rc = run_body(ctx, ALZ_FMT_LZ40, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_MDB4:                                                                        // MDB4.cs:33-50
        if (len < 4 || memcmp(src, "MDB4", 4)) return ALZ_E_FORMAT;
        if (len < 32) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = le32(src + 8); hdr = 32;
        rc = run_body(ctx, ALZ_FMT_LZSS, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    case ALZ_C_FCMP: case ALZ_C_IECP: case ALZ_C_GCZ: {                                     // FCMP.cs:36-41, IECP.cs:35-39, GCZ.cs:32-36
        const char* magic = container == ALZ_C_FCMP ? "FCMP" : "IECP";
        const size_t ml = container == ALZ_C_GCZ ? 0 : 4;
        hdr = ml + (container == ALZ_C_FCMP ? 8 : 4);
        if (len < ml || (ml && memcmp(src, magic, 4))) return ALZ_E_FORMAT;
        if (len < hdr) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = le32(src + ml);
        rc = run_body(ctx, ALZ_FMT_LZSS, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        break;
    }
    case ALZ_C_SDPC:                                                                        // SDPC.cs:34-47
        if (len < 4 || memcmp(src, "SDPC", 4)) return ALZ_E_FORMAT;
        if (len < 8) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        size = le32(src + 4); hdr = 8;
        rc = run_body(ctx, ALZ_FMT_LZO, nullptr, src + hdr, len - hdr, 0, 0, 0, dst, dst_cap, &r);
        if (rc == ALZ_OK && r.status == ALZ_ST_OK && r.dst_len > size) r.status = ALZ_ST_OUTPUT_SIZE_MISMATCH;   // '>' only  SDPC.cs:43
        break;
    case ALZ_C_ECD: {                                                                       // ECD.cs:45-71
        if (len < 3 || memcmp(src, "ECD", 3)) return ALZ_E_FORMAT;
        if (len < 16) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
        const bool compressed = src[3] == 1;
        const uint32_t plain = be32(src + 4); size = be32(src + 12); hdr = 16;
        if (!compressed) {                                                                  // source.CopyTo(destination)
            const size_t n = len - hdr;
            if (n > dst_cap) { memcpy(dst, src + hdr, dst_cap); r.dst_len = clamp32(dst_cap); r.status = ALZ_ST_OUTPUT_CAPACITY; break; }
            memcpy(dst, src + hdr, n); r.dst_len = clamp32(n); r.src_used = clamp32(n); r.status = ALZ_ST_OK;
            break;
        }
        if ((uint64_t)plain > len - hdr) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }      // ReadByte() == -1 would write 0xFF bytes: refused
        if (plain > dst_cap) { r.status = ALZ_ST_OUTPUT_CAPACITY; break; }
        memcpy(dst, src + hdr, plain);
        alz_lz_properties e; memset(&e, 0, sizeof(e));                                      // LzProperties(0x400, 0x42, 3, 0x3BE) == (10, 6, 2)  ECD.cs:15
        e.window_bits = 10; e.length_bits = 6; e.min_length = 3; e.windows_start = 0x3BE; e.max_distance = 0x400;
        rc = run_body(ctx, ALZ_FMT_LZSS, &e, src + hdr + plain, len - hdr - plain, size - plain, 0, 0, dst + plain, dst_cap - plain, &r);
        r.dst_len += plain; r.src_used += plain;
        break;
    }
    case ALZ_C_LEVEL5: {                                                                    // Level5.cs:62-110
        if (len < 4) return ALZ_E_FORMAT;
        const uint32_t ts = le32(src); hdr = 4;
        if (len > 4 && src[4] == 0x78) return ALZ_E_UNSUPPORTED;                            // zlib payload: BCL code, out of scope
        size = ts >> 3;
        if ((ts & 7) == ALZ_LEVEL5_ONLYSAVE) {
            if (len - hdr < size) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
            if (dst_cap < size) { r.status = ALZ_ST_OUTPUT_CAPACITY; break; }
            memcpy(dst, src + hdr, size); r.dst_len = size; r.src_used = size; r.status = ALZ_ST_OK;
        } else if ((ts & 7) == ALZ_LEVEL5_LZ10) rc = run_body(ctx, ALZ_FMT_LZ10, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
        else return ALZ_E_UNSUPPORTED;                                                      // RLE / Huffman: not LZ
        break;
    }
    case ALZ_C_LZ77: {                                                                      // LZ77.cs:105-153
        if (len < 8 || memcmp(src, "LZ77", 4)) return ALZ_E_FORMAT;
        const uint32_t type = src[4];
        size = (uint32_t)src[5] | ((uint32_t)src[6] << 8) | ((uint32_t)src[7] << 16); hdr = 8;
        if (size == 0) { if (len < 12) return ALZ_E_FORMAT; size = le32(src + 8); hdr = 12; }
        if (type == ALZ_LZ77_LZ10 || type == ALZ_LZ77_LZ11) {
            rc = run_body(ctx, type == ALZ_LZ77_LZ10 ? ALZ_FMT_LZ10 : ALZ_FMT_LZ11, nullptr, src + hdr, len - hdr, size, 0, 0, dst, dst_cap, &r);
            break;
        }
        if (type != ALZ_LZ77_CHUNKLZ10) return ALZ_E_UNSUPPORTED;                            // RLE30 / HUF20: not LZ
        // ChunkLZ10: u16 end offsets until (last + position == length), then one LZ10 FILE per chunk.  The chunks are
        // independent streams: they go to the GPU as ONE batch.
        std::vector<uint32_t> ends; size_t pos = hdr;
        for (;;) {
            if (pos + 2 > len) { r.status = ALZ_ST_INPUT_TRUNCATED; break; }
            ends.push_back((uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8)); pos += 2;
            if (ends.back() + pos == len) break;
        }
        if (r.status == ALZ_ST_INPUT_TRUNCATED) break;
        const size_t header_end = pos;
        std::vector<alz_stream> st(ends.size()); std::vector<alz_result> rs(ends.size());
        uint64_t out_off = 0; bool bad = false;
        for (size_t i = 0; i < ends.size() && !bad; i++) {
            const size_t a = header_end + (i ? ends[i - 1] : 0);
            uint32_t csz = 0; const int h = a < len ? nin_header(src + a, len - a, 0x10, &csz) : -1;
            if (h < 0) { bad = true; break; }
            memset(&st[i], 0, sizeof(alz_stream));
            st[i].src_off = a + (size_t)h; st[i].src_len = clamp32(len - a - (size_t)h); st[i].dst_off = out_off;
            st[i].dst_cap = clamp32(out_off < dst_cap ? dst_cap - out_off : 0); st[i].decom_len = csz; st[i].format = ALZ_FMT_LZ10;
            out_off += csz;
        }
        if (bad) return ALZ_E_FORMAT;
        rc = alz_decode_batch(ctx, nullptr, (uint32_t)st.size(), src, len, st.data(), dst, dst_cap, rs.data());
        if (rc != ALZ_OK) return rc;
        r.status = ALZ_ST_OK; r.dst_len = 0;
        for (size_t i = 0; i < rs.size(); i++) {
            if (rs[i].status != ALZ_ST_OK) { r.status = rs[i].status; r.dst_len = (uint32_t)(st[i].dst_off + rs[i].dst_len); break; }
            r.dst_len = (uint32_t)(st[i].dst_off + rs[i].dst_len);
        }
        if (r.status == ALZ_ST_OK && r.dst_len > size) r.status = ALZ_ST_OUTPUT_SIZE_MISMATCH;   // LZ77.cs:146-149
        r.src_used = (uint32_t)(header_end + ends.back() - hdr);
        break;
    }
    default: return ALZ_E_UNSUPPORTED;
    }
    if (rc != ALZ_OK) return rc;
    if (dst_len) *dst_len = r.dst_len;
    if (src_used) *src_used = hdr + r.src_used;
    if (status) *status = r.status;
    return r.status == ALZ_ST_OK ? ALZ_OK : ALZ_E_STREAM;
}

size_t alz_container_compress_bound(uint32_t container, size_t n) {
    if (container == ALZ_C_SNAPPY) return 10 + n + (n / 0x10000 + 1) * 8 + 64;      // stored chunks bound the size
    if (container == ALZ_C_LZ4_FRAME || container == ALZ_C_LZ4_LEGACY) return n + n / 200 + (n / 0x10000 + 1) * 8 + 64;
    return n + n / 4 + 256;  // (HIG header: 0xC0) flag-byte formats: <= 9/8 n + header; LZ4/LZO/Snappy literal-run overhead n/255
}

// ICompressionEncoder.Compress(ReadOnlySpan<byte>, Stream, CompressionSettings)  Interfaces/ICompressionEncoder.cs:19
int alz_container_compress(alz_ctx* ctx, uint32_t container, const alz_container_options* opt, const alz_settings* settings,
                           const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* dst_len) {
    if (!ctx || (!src && n) || !dst) return ALZ_E_INVALID;
    const bool big = opt ? opt->big_endian != 0 : true;
    const alz_lz_properties* lz = opt ? &opt->lz : nullptr;
    alz_settings st; if (settings) st = *settings; else { st.quality = 8; st.max_window_bits = 0; st.strategy = 0; st.min_distance = 0; }
    // ---- wrappers that prepend a magic to another container / have their own small header
    switch (container) {
    case ALZ_C_ECD: {                                                                       // ECD.cs:73-109
        const bool compressed = st.quality != 0 && n > 0x10;
        const uint32_t plain = compressed ? 4u : 0u;                                        // ECD.PlainSize default
        if (cap < 16 + (compressed ? plain : n)) return ALZ_E_NOMEM;
        memcpy(dst, "ECD", 3);
        if (compressed) {
            alz_lz_properties e; memset(&e, 0, sizeof(e));
            e.window_bits = 10; e.length_bits = 6; e.min_length = 3; e.windows_start = 0x3BE; e.max_distance = 0x400;
            alz_stream s1; memset(&s1, 0, sizeof(s1));
            s1.src_len = clamp32(n - plain); s1.dst_cap = clamp32(cap - 16 - plain); s1.format = ALZ_FMT_LZSS;
            alz_result r1;
            int rc1 = alz_encode_batch(ctx, &e, &st, 1, src + plain, n - plain, &s1, dst + 16 + plain, cap - 16 - plain, &r1, nullptr);
            if (rc1 != ALZ_OK) return rc1;
            if (r1.status == ALZ_ST_OK && (uint64_t)plain + r1.dst_len < n) {
                dst[3] = 1; wr32(dst + 4, plain, true); wr32(dst + 8, plain + r1.dst_len, true); wr32(dst + 12, (uint32_t)n, true);
                memcpy(dst + 16, src, plain);
                if (dst_len) *dst_len = 16 + plain + r1.dst_len;
                return ALZ_OK;
            }
            if (r1.status != ALZ_ST_OK && r1.status != ALZ_ST_OUTPUT_CAPACITY) return ALZ_E_INVALID;
            if (cap < 16 + n) return ALZ_E_NOMEM;                                           // compression was ineffective: stored  ECD.cs:99-104
        }
        dst[3] = 0; wr32(dst + 4, 0, true); wr32(dst + 8, (uint32_t)n, true); wr32(dst + 12, (uint32_t)n, true);
        memcpy(dst + 16, src, n);
        if (dst_len) *dst_len = 16 + n;
        return ALZ_OK;
    }
    case ALZ_C_LZ4_LEGACY: return lz4_file_compress(ctx, true, 0, &st, src, n, dst, cap, dst_len);
    case ALZ_C_LZ4_FRAME: return lz4_file_compress(ctx, false, opt ? opt->chunk_size : 0, &st, src, n, dst, cap, dst_len);
    case ALZ_C_SNAPPY: return snappy_file_compress(ctx, &st, src, n, dst, cap, dst_len);
    case ALZ_C_GCLZ: case ALZ_C_CXLZ: case ALZ_C_LZ_3DS: case ALZ_C_COMP: {                 // GCLZ.cs:40-44
        const char* magic = container == ALZ_C_GCLZ ? "GCLZ" : container == ALZ_C_CXLZ ? "CXLZ" : container == ALZ_C_COMP ? "COMP" : "3DS-LZ\r\n";
        const size_t ml = container == ALZ_C_LZ_3DS ? 8 : 4;
        if (cap < ml) return ALZ_E_NOMEM;
        memcpy(dst, magic, ml);
        size_t inner = 0;
        int rc2 = alz_container_compress(ctx, container == ALZ_C_COMP ? ALZ_C_LZ11 : ALZ_C_LZ10, opt, settings, src, n, dst + ml, cap - ml, &inner);
        if (dst_len) *dst_len = ml + inner;
        return rc2;
    }
    case ALZ_C_LZ77: {                                                                      // LZ77.cs:56-102
        const uint32_t type = opt && opt->variant ? opt->variant : ALZ_LZ77_LZ10;
        const size_t chunk = opt && opt->chunk_size ? opt->chunk_size : 0x1000;
        if (cap < 8) return ALZ_E_NOMEM;
        memcpy(dst, "LZ77", 4);
        if (type == ALZ_LZ77_LZ10 || type == ALZ_LZ77_LZ11 || (type == ALZ_LZ77_CHUNKLZ10 && chunk >= n)) {
            size_t inner = 0;
            int rc2 = alz_container_compress(ctx, type == ALZ_LZ77_LZ11 ? ALZ_C_LZ11 : ALZ_C_LZ10, opt, settings, src, n, dst + 4, cap - 4, &inner);
            if (dst_len) *dst_len = 4 + inner;
            return rc2;
        }
        if (type != ALZ_LZ77_CHUNKLZ10) return ALZ_E_UNSUPPORTED;
        if (n > 0xFFFFFF) return ALZ_E_INVALID;
        // ChunkLZ10: every chunk is an independent LZ10 file -> ONE encode batch on the GPU
        const size_t segs = (n + chunk - 1) / chunk;
        const size_t header_end = 8 + 2 * segs;
        if (cap < header_end) return ALZ_E_NOMEM;
        wr32(dst + 4, (uint32_t)type | ((uint32_t)n << 8), false);
        std::vector<alz_stream> stv(segs); std::vector<alz_result> rs(segs);
        const size_t slot = chunk + chunk / 4 + 64;
        BlockSlots tmp; if (!tmp.resize(segs * slot)) return ALZ_E_NOMEM;
        alz_settings st2; if (settings) st2 = *settings; else { st2.quality = 8; st2.max_window_bits = 0; st2.strategy = 0; }
        st2.min_distance = 2;                                                                // new LZ10(): GbaVramCompatibilityMode = true
        for (size_t i = 0; i < segs; i++) {
            memset(&stv[i], 0, sizeof(alz_stream));
            stv[i].src_off = i * chunk; stv[i].src_len = (uint32_t)((n - i * chunk) < chunk ? (n - i * chunk) : chunk);
            stv[i].dst_off = i * slot; stv[i].dst_cap = (uint32_t)slot; stv[i].format = ALZ_FMT_LZ10;
        }
        int rc2 = alz_encode_batch(ctx, nullptr, &st2, (uint32_t)segs, src, n, stv.data(), tmp.data(), tmp.size(), rs.data(), nullptr);
        if (rc2 != ALZ_OK) return rc2;
        size_t pos = header_end;
        for (size_t i = 0; i < segs; i++) {
            if (rs[i].status != ALZ_ST_OK) return ALZ_E_INVALID;
            if (pos + 4 + rs[i].dst_len > cap) return ALZ_E_NOMEM;
            wr32(dst + pos, 0x10u | (stv[i].src_len << 8), false);                           // LZ10 header of the chunk (LZ10.cs:69-71)
            memcpy(dst + pos + 4, tmp.data() + i * slot, rs[i].dst_len);
            pos += 4 + rs[i].dst_len;
            const size_t endoff = pos - header_end;
            if (endoff > 0xFFFF) return ALZ_E_INVALID;                                        // "chunks too large to process"  LZ77.cs:92-95
            dst[8 + 2 * i] = (uint8_t)endoff; dst[8 + 2 * i + 1] = (uint8_t)(endoff >> 8);
        }
        if (dst_len) *dst_len = pos;
        return ALZ_OK;
    }
    case ALZ_C_BLZ: {                                                                       // BLZ.cs:71-95, :137-150
        std::vector<uint8_t> rev(src, src + n);                                             // "Our LZMatcher only works in one direction."
        std::reverse(rev.begin(), rev.end());
        std::vector<uint8_t> body(n + n / 4 + 64);
        alz_stream s; memset(&s, 0, sizeof(s));
        s.src_len = clamp32(n); s.dst_cap = clamp32(body.size()); s.format = ALZ_FMT_BLZ;
        alz_result r; alz_encode_aux aux;
        int rc = alz_encode_batch(ctx, nullptr, &st, 1, rev.data(), n, &s, body.data(), body.size(), &r, &aux);
        if (rc != ALZ_OK) return rc;
        if (r.status != ALZ_ST_OK) return r.status == ALZ_ST_OUTPUT_CAPACITY ? ALZ_E_NOMEM : ALZ_E_INVALID;
        uint32_t total = r.dst_len + 8u; const uint32_t pad = (16u - (total % 16u)) % 16u; total += pad;
        if (cap < total) return ALZ_E_NOMEM;
        std::reverse_copy(body.begin(), body.begin() + r.dst_len, dst);                     // the managed buffer is filled from its end: stored back to front
        memset(dst + r.dst_len, 0xFF, pad);
        uint8_t* f = dst + r.dst_len + pad;
        f[0] = (uint8_t)total; f[1] = (uint8_t)(total >> 8); f[2] = (uint8_t)(total >> 16); f[3] = (uint8_t)(8u + pad);
        wr32(f + 4, (uint32_t)((int64_t)n - (int64_t)total), false);
        if (dst_len) *dst_len = total;
        return ALZ_OK;
    }
    case ALZ_C_LEVEL5: {                                                                    // Level5.cs:112-146
        uint32_t type = opt && opt->variant ? opt->variant : ALZ_LEVEL5_LZ10;
        if (st.quality == 0) type = ALZ_LEVEL5_ONLYSAVE;
        if (cap < 4) return ALZ_E_NOMEM;
        wr32(dst, type | ((uint32_t)n << 3), false);
        if (type == ALZ_LEVEL5_ONLYSAVE) { if (cap < 4 + n) return ALZ_E_NOMEM; memcpy(dst + 4, src, n); if (dst_len) *dst_len = 4 + n; return ALZ_OK; }
        if (type != ALZ_LEVEL5_LZ10) return ALZ_E_UNSUPPORTED;
        break;
    }
    default: break;
    }
    size_t hdr = 0; uint32_t fmt;
    switch (container) {
    case ALZ_C_YAZ1: fmt = ALZ_FMT_YAZ0; hdr = 16; break;
    case ALZ_C_AKLZ: fmt = ALZ_FMT_LZSS; hdr = 16; lz = nullptr; break;
    case ALZ_C_LZ01: fmt = ALZ_FMT_LZSS; hdr = 16; lz = nullptr; break;
    case ALZ_C_LZSEGA: fmt = ALZ_FMT_LZSS; hdr = 8; lz = nullptr; break;
    case ALZ_C_LZ00: fmt = ALZ_FMT_LZSS; hdr = 64; lz = nullptr; break;
    case ALZ_C_LEVEL5LZSS: fmt = ALZ_FMT_LZSS; hdr = 16; lz = nullptr; break;
    case ALZ_C_LZON: fmt = ALZ_FMT_LZO; hdr = 16; break;
    case ALZ_C_MDB4: fmt = ALZ_FMT_LZSS; hdr = 32; lz = nullptr; break;
    case ALZ_C_FCMP: fmt = ALZ_FMT_LZSS; hdr = 12; lz = nullptr; break;
    case ALZ_C_IECP: fmt = ALZ_FMT_LZSS; hdr = 8; lz = nullptr; break;
    case ALZ_C_GCZ: fmt = ALZ_FMT_LZSS; hdr = 4; lz = nullptr; break;
    case ALZ_C_SDPC: fmt = ALZ_FMT_LZO; hdr = 8; break;
    case ALZ_C_LEVEL5: fmt = ALZ_FMT_LZ10; hdr = 4; if (st.min_distance == 0) st.min_distance = 2; break;   // LZ10.CompressHeaderless default gbaVramCompatibilityMode = true
    case ALZ_C_LZSS: fmt = ALZ_FMT_LZSS; hdr = 16; break;
    case ALZ_C_LZ10: fmt = ALZ_FMT_LZ10; hdr = n <= 0xFFFFFF ? 4 : 8; if (st.min_distance == 0) st.min_distance = 2; break;   // GbaVramCompatibilityMode = true  LZ10.cs:33
    case ALZ_C_LZ11: fmt = ALZ_FMT_LZ11; hdr = n <= 0xFFFFFF ? 4 : 8; break;
    case ALZ_C_LZ40: case ALZ_C_LZ60: fmt = ALZ_FMT_LZ40; hdr = n <= 0xFFFFFF ? 4 : 8; break;   // GbaVramCompatibilityMode = false  LZ40.cs:29
    case ALZ_C_LZHUDSON: fmt = ALZ_FMT_LZHUDSON; hdr = 4; break;
    case ALZ_C_SMSR00: fmt = ALZ_FMT_SMSR00; hdr = 16; break;
    case ALZ_C_YAZ0: fmt = ALZ_FMT_YAZ0; hdr = 16; break;
    case ALZ_C_YAY0: fmt = ALZ_FMT_YAY0; hdr = 16; break;
    case ALZ_C_MIO0: fmt = ALZ_FMT_MIO0; hdr = 16; break;
    case ALZ_C_PRS: fmt = big ? ALZ_FMT_PRS_BE : ALZ_FMT_PRS_LE; break;
    case ALZ_C_LZO: fmt = ALZ_FMT_LZO; break;
    case ALZ_C_CNX2: fmt = ALZ_FMT_CNX2; hdr = 16; break;
    case ALZ_C_CLZ0: fmt = ALZ_FMT_CLZ0; hdr = 16; break;
    case ALZ_C_LZ02: fmt = ALZ_FMT_LZ02; hdr = 4; break;
    case ALZ_C_WFLZ: fmt = (opt && opt->big_endian) ? ALZ_FMT_WFLZ_BE : ALZ_FMT_WFLZ; hdr = 12; break;
    case ALZ_C_LZSHREK: fmt = ALZ_FMT_LZSHREK; hdr = 16; break;
    case ALZ_C_HIG: fmt = ALZ_FMT_HIG; hdr = 0xC0; break;                                      // Version 6 (the class default): 0x40 header + 0x80 extension
    case ALZ_C_REFPACK: if (n >= 0xFFFFFF) return ALZ_E_UNSUPPORTED; fmt = ALZ_FMT_REFPACK; hdr = 9; break;   // "RefPack Version 2 does not support files over 16MB."  RefPack.cs:110-111
    case ALZ_C_CNS: if (n < 4) return ALZ_E_INVALID; fmt = ALZ_FMT_CNS; hdr = 16; break;          // source[3]: IndexOutOfRangeException  CNS.cs:61
    case ALZ_C_FASTLZ: fmt = ALZ_FMT_FASTLZ; break;                                             // FastLZ.cs:162-163 (level 1: MaxWindowBits stays 0)
    default: return ALZ_E_UNSUPPORTED;
    }
    if (cap < hdr) return ALZ_E_NOMEM;
    alz_stream s; memset(&s, 0, sizeof(s));
    s.src_off = 0; s.src_len = clamp32(n); s.dst_off = 0; s.dst_cap = clamp32(cap - hdr); s.format = fmt;
    alz_result r; alz_encode_aux aux;
    int rc = alz_encode_batch(ctx, lz, &st, 1, src, n, &s, dst + hdr, cap - hdr, &r, &aux);
    if (rc != ALZ_OK) return rc;
    if (r.status == ALZ_ST_OUTPUT_CAPACITY) return ALZ_E_NOMEM;
    if (r.status != ALZ_ST_OK) return ALZ_E_INVALID;
    switch (container) {
    case ALZ_C_LZSS: memcpy(dst, "LZSS", 4); wr32(dst + 4, (uint32_t)n, true); wr32(dst + 8, r.dst_len, true); wr32(dst + 12, 0, true); break;   // LZSS.cs:72-88
    case ALZ_C_LZ10: case ALZ_C_LZ11: case ALZ_C_LZ40: case ALZ_C_LZ60: {                                                                      // LZ10.cs:67-80, LZ40.cs:64-77
        const uint8_t id = container == ALZ_C_LZ10 ? 0x10 : container == ALZ_C_LZ11 ? 0x11 : container == ALZ_C_LZ40 ? 0x40 : 0x60;
        if (n <= 0xFFFFFF) wr32(dst, id | ((uint32_t)n << 8), false); else { wr32(dst, id, false); wr32(dst + 4, (uint32_t)n, false); }
        break;
    }
    case ALZ_C_YAZ1: memcpy(dst, "Yaz1", 4); wr32(dst + 4, (uint32_t)n, big); wr32(dst + 8, opt ? opt->memory_alignment : 0, big); wr32(dst + 12, 0, false); break;
    case ALZ_C_AKLZ: memcpy(dst, kAklzMagic, 12); wr32(dst + 12, (uint32_t)n, true); break;                                                        // AKLZ.cs:50-55
    case ALZ_C_LZ01: memcpy(dst, "LZ01", 4); wr32(dst + 4, (uint32_t)(hdr + r.dst_len), false); wr32(dst + 8, (uint32_t)n, false); wr32(dst + 12, 0, false); break;   // LZ01.cs:65-82
    case ALZ_C_LZSEGA: wr32(dst, r.dst_len, false); wr32(dst + 4, (uint32_t)n, false); break;                                                      // LZSega.cs:57-67
    case ALZ_C_LZ00: {                                                                                                                         // LZ00.cs:71-96
        const uint32_t key = opt ? opt->key : 0;
        lz00_keystream(dst + hdr, r.dst_len, key);                                          // LZSS.CompressHeaderless writes through the StreamTransformer
        memset(dst, 0, 64); memcpy(dst, "LZ00", 4); wr32(dst + 4, (uint32_t)(hdr + r.dst_len), false);
        bool named = false; if (opt) for (int i = 0; i < 32; i++) named |= opt->name[i] != 0;
        if (named) memcpy(dst + 16, opt->name, 32); else memcpy(dst + 16, "Temp.dat", 8);
        wr32(dst + 48, (uint32_t)n, false); wr32(dst + 52, key, false);
        break;
    }
    case ALZ_C_REFPACK:                                                                                                                            // RefPack.cs:105-125: Options = Default | UsePreHeader -> version 2
        wr32(dst, (uint32_t)(hdr + r.dst_len - 4), false); dst[4] = 0x10; dst[5] = 0xFB; dst[6] = (uint8_t)(n >> 16); dst[7] = (uint8_t)(n >> 8); dst[8] = (uint8_t)n; break;
    case ALZ_C_HIG:                                                                                                                                // HIG.cs:82-124
        memset(dst, 0, 0xC0); memcpy(dst, "HIG!", 4); wr32(dst + 0x38, 6, false); wr32(dst + 0x3C, (uint32_t)n, false);
        wr32(dst + 0x40, r.dst_len, false); memcpy(dst + 0x44, "C:\\HIG\\PROJECTS\\test.mb.wad.conf", 32); break;
    case ALZ_C_LZSHREK: wr32(dst, 0x10, false); wr32(dst + 4, (uint32_t)n, false); wr32(dst + 8, r.dst_len, false); wr32(dst + 12, 0, false); break;   // LZShrek.cs:60-71
    case ALZ_C_WFLZ: { const bool wbig = opt && opt->big_endian; memcpy(dst, "WFLZ", 4); wr32(dst + 4, r.dst_len, wbig); wr32(dst + 8, (uint32_t)n, wbig); break; }   // WFLZ.cs:89-105
    case ALZ_C_LZ02: dst[0] = 1; dst[1] = (uint8_t)(n >> 16); dst[2] = (uint8_t)(n >> 8); dst[3] = (uint8_t)n; break;                                  // LZ02.cs:66-75 (DataType.Default: no extension data)
    case ALZ_C_CNS:                                                                                                                                // CNS.cs:57-75
        memcpy(dst, "@CNS", 4); memcpy(dst + 4, (src[0] == 0x00 && src[1] == 0x20 && src[2] == 0xAF && src[3] == 0x30) ? "TPL\0" : "PAK\0", 4);
        wr32(dst + 8, (uint32_t)n, false); wr32(dst + 12, 0, false); break;
    case ALZ_C_CLZ0: memcpy(dst, "CLZ\0", 4); wr32(dst + 4, (uint32_t)n, true); wr32(dst + 8, 0, true); wr32(dst + 12, (uint32_t)n, true); break;           // CLZ0.cs:53-62
    case ALZ_C_CNX2: memcpy(dst, "CNX\x02" "DEC\x10", 8); wr32(dst + 8, r.dst_len, true); wr32(dst + 12, (uint32_t)n, true); break;               // CNX2.cs:64-81 (Extension "DEC", padded with 0x10)
    case ALZ_C_LEVEL5LZSS: memcpy(dst, "SSZL", 4); wr32(dst + 4, 0, false); wr32(dst + 8, r.dst_len, false); wr32(dst + 12, (uint32_t)n, false); break;   // Level5LZSS.cs:62-72
    case ALZ_C_LZON: memcpy(dst, kLzonMagic, 8); wr32(dst + 8, (uint32_t)n, true); wr32(dst + 12, r.dst_len, true); break;                          // LZOn.cs:63-79
    case ALZ_C_LZHUDSON: wr32(dst, (uint32_t)n, true); break;                                                                                   // LZHudson.cs:39-43
    case ALZ_C_SMSR00: memcpy(dst, "SMSR00", 6); dst[6] = 0; dst[7] = 0; wr32(dst + 8, (uint32_t)n, true); wr32(dst + 12, 16 + aux.aux0, true); break;   // SMSR00.cs:58-63
    case ALZ_C_MDB4: memcpy(dst, "MDB4", 4); wr32(dst + 4, (uint32_t)n + 1, false); wr32(dst + 8, (uint32_t)n, false); wr32(dst + 12, 16 + r.dst_len, false); memset(dst + 16, 0, 16); break;   // MDB4.cs:52-72
    case ALZ_C_FCMP: memcpy(dst, "FCMP", 4); wr32(dst + 4, (uint32_t)n, false); wr32(dst + 8, 305397760u, false); break;                          // FCMP.cs:43-50
    case ALZ_C_IECP: memcpy(dst, "IECP", 4); wr32(dst + 4, (uint32_t)n, false); break;                                                          // IECP.cs:41-46
    case ALZ_C_GCZ: wr32(dst, (uint32_t)n, false); break;                                                                                       // GCZ.cs:38-42
    case ALZ_C_SDPC: memcpy(dst, "SDPC", 4); wr32(dst + 4, (uint32_t)n, false); break;                                                          // SDPC.cs:49-54
    case ALZ_C_LEVEL5: break;                                                                                                                       // header written above
    case ALZ_C_YAZ0: memcpy(dst, "Yaz0", 4); wr32(dst + 4, (uint32_t)n, big); wr32(dst + 8, opt ? opt->memory_alignment : 0, big); wr32(dst + 12, 0, false); break;   // Yaz0.cs:82-89
    case ALZ_C_YAY0: case ALZ_C_MIO0:                                                                                                          // Yay0.cs:62-77
        memcpy(dst, container == ALZ_C_YAY0 ? "Yay0" : "MIO0", 4); wr32(dst + 4, (uint32_t)n, big); wr32(dst + 8, 0x10 + aux.aux0, big); wr32(dst + 12, 0x10 + aux.aux1, big); break;
    default: break;
    }
    if (dst_len) *dst_len = hdr + r.dst_len;
    return ALZ_OK;
}

}  // extern "C"

namespace {

// ---------------------------------------------------------------------------------------------- batch producers

// header of a single-body container at p -> stream descriptor (offsets relative to p); false: not decodable from here
static bool describe_stream(uint32_t container, bool big, const uint8_t* p, size_t n, alz_stream* s, size_t* hdr) {
    memset(s, 0, sizeof(*s));
    uint32_t size = 0;
    switch (container) {
    case ALZ_C_LZSS: if (n < 16) return false; size = be32(p + 4); *hdr = 16; s->format = ALZ_FMT_LZSS; break;
    case ALZ_C_LZ10: case ALZ_C_LZ11: { const int h = nin_header(p, n, container == ALZ_C_LZ10 ? 0x10 : 0x11, &size); if (h < 0) return false; *hdr = (size_t)h; s->format = container == ALZ_C_LZ10 ? ALZ_FMT_LZ10 : ALZ_FMT_LZ11; break; }
    case ALZ_C_LZ40: case ALZ_C_LZ60: { const int h = nin_header(p, n, container == ALZ_C_LZ40 ? 0x40 : 0x60, &size); if (h < 0) return false; *hdr = (size_t)h; s->format = ALZ_FMT_LZ40; break; }
    case ALZ_C_YAZ0: case ALZ_C_YAZ1: if (n < 16) return false; size = rd32(p + 4, big); *hdr = 16; s->format = ALZ_FMT_YAZ0; break;
    case ALZ_C_YAY0: case ALZ_C_MIO0:
        if (n < 16) return false;
        size = rd32(p + 4, big); s->aux0 = rd32(p + 8, big) - 0x10; s->aux1 = rd32(p + 12, big) - 0x10; *hdr = 16;
        s->format = container == ALZ_C_YAY0 ? ALZ_FMT_YAY0 : ALZ_FMT_MIO0; break;
    case ALZ_C_GCLZ: case ALZ_C_CXLZ: case ALZ_C_LZ_3DS: case ALZ_C_COMP: {
        const size_t ml = container == ALZ_C_LZ_3DS ? 8 : 4; size_t ih = 0;
        if (n < ml || !describe_stream(container == ALZ_C_COMP ? ALZ_C_LZ11 : ALZ_C_LZ10, big, p + ml, n - ml, s, &ih)) return false;
        *hdr = ml + ih; return true;
    }
    case ALZ_C_AKLZ: if (n < 16) return false; size = be32(p + 12); *hdr = 16; s->format = ALZ_FMT_LZSS; break;
    case ALZ_C_LZ01: if (n < 16) return false; size = le32(p + 8); *hdr = 16; s->format = ALZ_FMT_LZSS; break;
    case ALZ_C_LZSEGA: if (n < 8) return false; size = le32(p + 4); *hdr = 8; s->format = ALZ_FMT_LZSS; break;
    case ALZ_C_LEVEL5LZSS: if (n < 16) return false; size = le32(p + 12); *hdr = 16; s->format = ALZ_FMT_LZSS; break;
    case ALZ_C_MDB4: if (n < 32) return false; size = le32(p + 8); *hdr = 32; s->format = ALZ_FMT_LZSS; break;
    case ALZ_C_FCMP: if (n < 12) return false; size = le32(p + 4); *hdr = 12; s->format = ALZ_FMT_LZSS; break;
    case ALZ_C_IECP: if (n < 8) return false; size = le32(p + 4); *hdr = 8; s->format = ALZ_FMT_LZSS; break;
    case ALZ_C_CNX2: if (n < 16) return false; size = be32(p + 12); *hdr = 16; s->format = ALZ_FMT_CNX2; break;
    case ALZ_C_CLZ0: if (n < 16) return false; size = be32(p + 12); *hdr = 16; s->format = ALZ_FMT_CLZ0; break;
    case ALZ_C_CNS: if (n < 16) return false; size = le32(p + 8); *hdr = 16; s->format = ALZ_FMT_CNS; break;
    case ALZ_C_SMSR00: if (n < 16) return false; size = be32(p + 8); s->aux0 = be32(p + 12) - 16u; *hdr = 16; s->format = ALZ_FMT_SMSR00; break;
    case ALZ_C_HIG: {
        if (n < 0x40) return false;
        const uint32_t ver = le32(p + 0x38); const size_t start = (ver == 5 || ver == 6) ? 0xC0 : le32(p + 4);
        if (start > n) return false;
        size = le32(p + 0x3C); *hdr = start; s->format = ALZ_FMT_HIG; break;
    }
    default: return false;
    }
    s->decom_len = size;
    return true;
}

static const struct { const char* name; uint32_t format; uint8_t wb, lb, th; } kBrute[ALZ_BRUTE_DECODERS] = {   // BruteForceCommand.cs:96-131, the decoders on the path
    { "LZO", ALZ_FMT_LZO, 0, 0, 0 }, { "LZ4", ALZ_FMT_LZ4_BLOCK, 0, 0, 0 },
    { "LZSS (12, 4, 2)", ALZ_FMT_LZSS, 12, 4, 2 }, { "LZSS (12, 4, 3)", ALZ_FMT_LZSS, 12, 4, 3 },
    { "LZSS (10, 6, 2)", ALZ_FMT_LZSS, 10, 6, 2 }, { "LZSS (10, 6, 3)", ALZ_FMT_LZSS, 10, 6, 3 },
    { "LZSS0", ALZ_FMT_LZSS, 12, 4, 2 },                                   // LzProperties(0x1000, 0xF + 3, 3, 0xFEE) == (12, 4, 2)  LZSS.cs:34
    { "PRS big", ALZ_FMT_PRS_BE, 0, 0, 0 }, { "PRS Little", ALZ_FMT_PRS_LE, 0, 0, 0 },
    { "LZ10", ALZ_FMT_LZ10, 0, 0, 0 }, { "LZ11", ALZ_FMT_LZ11, 0, 0, 0 }, { "Yaz0", ALZ_FMT_YAZ0, 0, 0, 0 },
    { "LZ40", ALZ_FMT_LZ40, 0, 0, 0 }, { "LZHudson", ALZ_FMT_LZHUDSON, 0, 0, 0 },
    { "RefPack", ALZ_FMT_REFPACK, 0, 0, 0 }, { "LZ02", ALZ_FMT_LZ02, 0, 0, 0 }, { "CLZ0", ALZ_FMT_CLZ0, 0, 0, 0 }, { "CNS", ALZ_FMT_CNS, 0, 0, 0 },
    { "LZShrek", ALZ_FMT_LZSHREK, 0, 0, 0 },
    // (the command's "BLZ" entry decodes into MemoryStream.GetBuffer() and never advances Position, :121 -- it cannot
    //  report success, so it is not offered here)
};

}  // namespace

extern "C" {

const char* alz_brute_decoder_name(uint32_t i) { return i < ALZ_BRUTE_DECODERS ? kBrute[i].name : nullptr; }

int alz_brute_force(alz_ctx* ctx, const uint8_t* raw, size_t raw_len, uint32_t expected_size, uint8_t* dst, size_t slot, alz_result* results) {
    if (!ctx || (!raw && raw_len) || !dst || !results || slot < expected_size || raw_len > 0xFFFFFFFFull) return ALZ_E_INVALID;
    DevBuf d_src(ctx), d_dst(ctx);
    int rc;
    const size_t dslot = (slot + 255) & ~(size_t)255;
    if ((rc = alz_device_malloc(ctx, raw_len + 64, &d_src.p)) != ALZ_OK) return rc;
    if ((rc = alz_device_malloc(ctx, dslot * ALZ_BRUTE_DECODERS + 64, &d_dst.p)) != ALZ_OK) return rc;
    if (raw_len && (rc = alz_memcpy_h2d(ctx, d_src.p, raw, raw_len)) != ALZ_OK) return rc;
    bool done[ALZ_BRUTE_DECODERS] = { false };
    for (uint32_t g = 0; g < ALZ_BRUTE_DECODERS; g++) {                     // one batch per LZSS geometry, one for all fixed formats
        if (done[g]) continue;
        std::vector<alz_stream> ss; std::vector<uint32_t> who;
        for (uint32_t i = g; i < ALZ_BRUTE_DECODERS; i++) {
            if (done[i]) continue;
            const bool same = kBrute[g].format == ALZ_FMT_LZSS ? (kBrute[i].format == ALZ_FMT_LZSS && kBrute[i].wb == kBrute[g].wb && kBrute[i].lb == kBrute[g].lb && kBrute[i].th == kBrute[g].th)
                                                               : kBrute[i].format != ALZ_FMT_LZSS;
            if (!same) continue;
            alz_stream s; memset(&s, 0, sizeof(s));
            s.src_len = (uint32_t)raw_len; s.dst_off = (uint64_t)i * dslot; s.dst_cap = expected_size; s.decom_len = expected_size; s.format = kBrute[i].format;
            ss.push_back(s); who.push_back(i); done[i] = true;
        }
        alz_lz_properties lz; memset(&lz, 0, sizeof(lz));
        if (kBrute[g].format == ALZ_FMT_LZSS) {                            // LzProperties(byte, byte, byte)  LzProperties.cs:57-66
            lz.window_bits = kBrute[g].wb; lz.length_bits = kBrute[g].lb; lz.min_length = (uint8_t)(kBrute[g].th + 1);
            lz.max_distance = 1u << kBrute[g].wb; lz.windows_start = lz.max_distance - (1u << kBrute[g].lb) - kBrute[g].th;
        }
        alz_plan* pl = nullptr; std::vector<alz_result> rs(ss.size());
        if ((rc = alz_plan_create(ctx, kBrute[g].format == ALZ_FMT_LZSS ? &lz : nullptr, (uint32_t)ss.size(), ss.data(), &pl)) != ALZ_OK) return rc;
        rc = alz_plan_execute(ctx, pl, d_src.p, d_dst.p, nullptr);
        if (rc == ALZ_OK) rc = alz_plan_results(ctx, pl, rs.data());
        alz_plan_destroy(ctx, pl);
        if (rc != ALZ_OK) return rc;
        for (size_t k = 0; k < who.size(); k++) {
            results[who[k]] = rs[k];
            if (rs[k].dst_len && (rc = alz_memcpy_d2h(ctx, dst + (size_t)who[k] * slot, (uint8_t*)d_dst.p + (size_t)who[k] * dslot, rs[k].dst_len)) != ALZ_OK) return rc;
        }
    }
    return ALZ_OK;
}

int alz_container_scan(alz_ctx* ctx, const uint32_t* containers, uint32_t nc, const alz_container_options* opt, const uint8_t* src, size_t len,
                       uint8_t* dst, size_t dst_cap, alz_scan_hit* hits, uint32_t max_hits, uint32_t* nhits, size_t* dst_used) {
    if (!ctx || !containers || !nc || (!src && len) || !nhits || (max_hits && !hits)) return ALZ_E_INVALID;
    const bool big = opt ? opt->big_endian != 0 : true;
    *nhits = 0; if (dst_used) *dst_used = 0;
    const alz_lz_properties* lzp = nullptr;                                 // LZSS geometry: only the LZSS container itself may be non-default
    bool wrapper_lzss = false;
    for (uint32_t k = 0; k < nc; k++) {
        if (containers[k] >= ALZ_C_COUNT) return ALZ_E_INVALID;
        switch (containers[k]) {
        case ALZ_C_LZSS: if (opt && opt->lz.window_bits) lzp = &opt->lz; break;
        case ALZ_C_AKLZ: case ALZ_C_LZ01: case ALZ_C_LZSEGA: case ALZ_C_LEVEL5LZSS: case ALZ_C_MDB4: case ALZ_C_FCMP: case ALZ_C_IECP: wrapper_lzss = true; break;
        case ALZ_C_LZ10: case ALZ_C_LZ11: case ALZ_C_YAZ0: case ALZ_C_YAZ1: case ALZ_C_YAY0: case ALZ_C_MIO0:
        case ALZ_C_GCLZ: case ALZ_C_CXLZ: case ALZ_C_LZ_3DS: case ALZ_C_COMP: case ALZ_C_LZ40: case ALZ_C_LZ60:
        case ALZ_C_CNX2: case ALZ_C_CLZ0: case ALZ_C_CNS: case ALZ_C_SMSR00: case ALZ_C_HIG: break;
        default: return ALZ_E_UNSUPPORTED;
        }
    }
    if (lzp && wrapper_lzss) return ALZ_E_UNSUPPORTED;                      // one LZSS geometry per batch
    // 1. candidates: every offset some container identifies
    struct Cand { size_t off; uint32_t container; alz_stream st; size_t hdr; bool swapped; };
    std::vector<Cand> cands;
    const uint32_t kMaxStream = 256u << 20;
    for (size_t i = 0; i < len; i++) {
        for (uint32_t k = 0; k < nc; k++) {
            if (!alz_container_is_match(containers[k], src + i, len - i)) continue;
            Cand c; c.off = i; c.container = containers[k]; c.swapped = false;
            if (describe_stream(containers[k], big, src + i, len - i, &c.st, &c.hdr) && c.hdr <= len - i) {
                // Yaz0 / Yaz1: a size that only makes sense in the other byte order (Yaz0.cs:66-78 -- the first attempt runs out of input,
                // the second one reads the field reversed) is decoded with the reversed size at once; nothing is retried behind it
                if ((c.container == ALZ_C_YAZ0 || c.container == ALZ_C_YAZ1) && c.st.decom_len > kMaxStream && __builtin_bswap32(c.st.decom_len) <= kMaxStream) {
                    c.st.decom_len = __builtin_bswap32(c.st.decom_len); c.swapped = true;
                }
                if (c.st.decom_len <= kMaxStream) cands.push_back(c);
            }
            break;                                                          // Identify(): the first match decides
        }
    }
    if (cands.empty()) return ALZ_OK;
    DevBuf d_src(ctx), d_dst(ctx);
    int rc;
    if ((rc = alz_device_malloc(ctx, len + 64, &d_src.p)) != ALZ_OK) return rc;
    if ((rc = alz_memcpy_h2d(ctx, d_src.p, src, len)) != ALZ_OK) return rc;
    const uint64_t kRound = 1ull << 30;
    size_t d_cap = 0, next_free = 0, used = 0;
    size_t ci = 0;
    while (ci < cands.size()) {
        // 2. one round: as many candidates as fit kRound bytes of output
        std::vector<alz_stream> ss; std::vector<size_t> who; uint64_t arena = 0;
        while (ci < cands.size()) {
            Cand& c = cands[ci];
            if (c.off < next_free) { ci++; continue; }                      // already inside an accepted stream
            const uint64_t need = ((uint64_t)c.st.decom_len + 255) & ~255ull;
            if (!ss.empty() && arena + need > kRound) break;
            alz_stream s = c.st;
            s.src_off = c.off + c.hdr; s.src_len = clamp32(len - c.off - c.hdr); s.dst_off = arena; s.dst_cap = s.decom_len;
            ss.push_back(s); who.push_back(ci); arena += need; ci++;
        }
        if (ss.empty()) break;
        if (arena + 64 > d_cap) { if (d_dst.p) { alz_device_free(ctx, d_dst.p); d_dst.p = nullptr; } d_cap = (size_t)arena + 64; if ((rc = alz_device_malloc(ctx, d_cap, &d_dst.p)) != ALZ_OK) return rc; }
        alz_plan* pl = nullptr; std::vector<alz_result> rs(ss.size());
        if ((rc = alz_plan_create(ctx, lzp, (uint32_t)ss.size(), ss.data(), &pl)) != ALZ_OK) return rc;
        rc = alz_plan_execute(ctx, pl, d_src.p, d_dst.p, nullptr);
        if (rc == ALZ_OK) rc = alz_plan_results(ctx, pl, rs.data());
        alz_plan_destroy(ctx, pl);
        if (rc != ALZ_OK) return rc;
        // 3. replay the walk over this round's results
        for (size_t k = 0; k < ss.size(); k++) {
            const Cand& c = cands[who[k]];
            if (c.off < next_free) continue;
            const uint8_t* d_out = (uint8_t*)d_dst.p + ss[k].dst_off;
            DevBuf d_retry(ctx);
            if (rs[k].status != ALZ_ST_OK && !c.swapped && (c.container == ALZ_C_YAZ0 || c.container == ALZ_C_YAZ1)) {
                // Yaz0.Decompress catches the failure and decodes again with the size field read in the other byte order
                // (Yaz0.cs:66-78): one more single-stream decode for this candidate
                alz_stream s2 = ss[k];
                s2.decom_len = __builtin_bswap32(s2.decom_len); s2.dst_cap = s2.decom_len; s2.dst_off = 0;
                if (s2.decom_len <= kMaxStream) {
                    if ((rc = alz_device_malloc(ctx, (size_t)s2.decom_len + 64, &d_retry.p)) != ALZ_OK) return rc;
                    alz_plan* p2 = nullptr;
                    if ((rc = alz_plan_create(ctx, lzp, 1, &s2, &p2)) != ALZ_OK) return rc;
                    rc = alz_plan_execute(ctx, p2, d_src.p, d_retry.p, nullptr);
                    if (rc == ALZ_OK) rc = alz_plan_results(ctx, p2, &rs[k]);
                    alz_plan_destroy(ctx, p2);
                    if (rc != ALZ_OK) return rc;
                    d_out = (const uint8_t*)d_retry.p;
                }
            }
            if (rs[k].status != ALZ_ST_OK || rs[k].dst_len <= 0x10) continue;   // exception, or destination.Length <= 0x10
            if (*nhits >= max_hits || used + rs[k].dst_len > dst_cap) { if (dst_used) *dst_used = used; return ALZ_E_NOMEM; }
            if ((rc = alz_memcpy_d2h(ctx, dst + used, d_out, rs[k].dst_len)) != ALZ_OK) return rc;
            alz_scan_hit& h = hits[(*nhits)++];
            h.start = c.off; h.end = c.off + c.hdr + rs[k].src_used; h.dst_off = used; h.dst_len = rs[k].dst_len; h.container = c.container;
            used += rs[k].dst_len;
            next_free = (size_t)h.end;
        }
    }
    if (dst_used) *dst_used = used;
    return ALZ_OK;
}

}  // extern "C"
