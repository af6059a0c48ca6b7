// alz_container.cpp -- host side of the reference's format classes, restated above the C ABI.
//
// In the reference every format class does two things: (1) parse/emit its container header in managed code and
// (2) run the headerless LZ body.  (2) is the hot path and runs on the GPU (alz_decode / alz_encode_batch); this file
// is (1): header parsing, IsMatch heuristics, GetDecompressedSize and the endianness-retry logic, with the same
// argument meaning and error behaviour as the managed classes (cited per function, paths relative to
// /root/reference/src).  A C# shim would keep (1) in managed code and P/Invoke only alz_decode/alz_encode_batch
// (INTEGRATION.md); hosts without the managed library use these entry points instead.
#include <cstdint>
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

uint32_t clamp32(size_t v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; }

int run_body(alz_ctx* ctx, uint32_t fmt, const alz_lz_properties* lz, const uint8_t* body, size_t body_len, uint32_t size,
             uint32_t aux0, uint32_t aux1, uint8_t* dst, size_t cap, alz_result* r) {
    return alz_decode(ctx, fmt, lz, body, clamp32(body_len), size, aux0, aux1, dst, clamp32(cap), r);
}

}  // namespace

extern "C" {

// IProvidesDecompressedSize.GetDecompressedSize  Interfaces/IProvidesDecompressedSize.cs:20
int alz_container_decompressed_size(uint32_t container, const alz_container_options* opt, const uint8_t* src, size_t len, uint32_t* size_out) {
    if (!src || !size_out) return ALZ_E_INVALID;
    const bool big = opt ? opt->big_endian != 0 : true;
    switch (container) {
    case ALZ_C_LZSS: if (len < 8 || memcmp(src, "LZSS", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 4); return ALZ_OK;        // LZSS.cs:45-50
    case ALZ_C_LZ10: return nin_header(src, len, 0x10, size_out) < 0 ? ALZ_E_FORMAT : ALZ_OK;                                  // LZ10.cs:44-57
    case ALZ_C_LZ11: return nin_header(src, len, 0x11, size_out) < 0 ? ALZ_E_FORMAT : ALZ_OK;                                  // LZ11.cs:40-53
    case ALZ_C_YAZ0: if (len < 8 || memcmp(src, "Yaz0", 4)) return ALZ_E_FORMAT; *size_out = rd32(src + 4, big); return ALZ_OK;   // Yaz0.cs:50-55
    case ALZ_C_YAY0: if (len < 8 || memcmp(src, "Yay0", 4)) return ALZ_E_FORMAT; *size_out = be32(src + 4); return ALZ_OK;        // Yay0.cs:41-47 (always Endian.Big)
    case ALZ_C_MIO0: if (len < 8 || memcmp(src, "MIO0", 4)) return ALZ_E_FORMAT; *size_out = rd32(src + 4, big); return ALZ_OK;   // MIO0.cs:41-48
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
    case ALZ_C_SNAPPY: { static const uint8_t id[10] = { 0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59 }; return len > 0x10 && !memcmp(src, id, 10); }  // Snappy.cs:36-37
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
    default: return ALZ_E_UNSUPPORTED;
    }
    if (rc != ALZ_OK) return rc;
    if (dst_len) *dst_len = r.dst_len;
    if (src_used) *src_used = hdr + r.src_used;
    if (status) *status = r.status;
    return r.status == ALZ_ST_OK ? ALZ_OK : ALZ_E_STREAM;
}

size_t alz_container_compress_bound(uint32_t container, size_t n) {
    (void)container;
    return n + n / 4 + 64;   // flag-byte formats: <= 9/8 n + header; LZ4/LZO/Snappy literal-run overhead n/255
}

// ICompressionEncoder.Compress(ReadOnlySpan<byte>, Stream, CompressionSettings)  Interfaces/ICompressionEncoder.cs:19
int alz_container_compress(alz_ctx* ctx, uint32_t container, const alz_container_options* opt, const alz_settings* settings,
                           const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* dst_len) {
    if (!ctx || (!src && n) || !dst) return ALZ_E_INVALID;
    const bool big = opt ? opt->big_endian != 0 : true;
    const alz_lz_properties* lz = opt ? &opt->lz : nullptr;
    alz_settings st; if (settings) st = *settings; else { st.quality = 8; st.max_window_bits = 0; st.strategy = 0; st.min_distance = 0; }
    size_t hdr = 0; uint32_t fmt;
    switch (container) {
    case ALZ_C_LZSS: fmt = ALZ_FMT_LZSS; hdr = 16; break;
    case ALZ_C_LZ10: fmt = ALZ_FMT_LZ10; hdr = n <= 0xFFFFFF ? 4 : 8; if (st.min_distance == 0) st.min_distance = 2; break;   // GbaVramCompatibilityMode = true  LZ10.cs:33
    case ALZ_C_LZ11: fmt = ALZ_FMT_LZ11; hdr = n <= 0xFFFFFF ? 4 : 8; break;
    case ALZ_C_YAZ0: fmt = ALZ_FMT_YAZ0; hdr = 16; break;
    case ALZ_C_YAY0: fmt = ALZ_FMT_YAY0; hdr = 16; break;
    case ALZ_C_MIO0: fmt = ALZ_FMT_MIO0; hdr = 16; break;
    case ALZ_C_PRS: fmt = big ? ALZ_FMT_PRS_BE : ALZ_FMT_PRS_LE; break;
    case ALZ_C_LZO: fmt = ALZ_FMT_LZO; break;
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
    case ALZ_C_LZ10: case ALZ_C_LZ11: {                                                                                                        // LZ10.cs:67-80
        const uint8_t id = container == ALZ_C_LZ10 ? 0x10 : 0x11;
        if (n <= 0xFFFFFF) wr32(dst, id | ((uint32_t)n << 8), false); else { wr32(dst, id, false); wr32(dst + 4, (uint32_t)n, false); }
        break;
    }
    case ALZ_C_YAZ0: memcpy(dst, "Yaz0", 4); wr32(dst + 4, (uint32_t)n, big); wr32(dst + 8, opt ? opt->memory_alignment : 0, big); wr32(dst + 12, 0, false); break;   // Yaz0.cs:82-89
    case ALZ_C_YAY0: case ALZ_C_MIO0:                                                                                                          // Yay0.cs:62-77
        memcpy(dst, container == ALZ_C_YAY0 ? "Yay0" : "MIO0", 4); wr32(dst + 4, (uint32_t)n, big); wr32(dst + 8, 0x10 + aux.aux0, big); wr32(dst + 12, 0x10 + aux.aux1, big); break;
    default: break;
    }
    if (dst_len) *dst_len = hdr + r.dst_len;
    return ALZ_OK;
}

}  // extern "C"
