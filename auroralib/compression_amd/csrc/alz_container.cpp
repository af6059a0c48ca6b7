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

const uint8_t kAklzMagic[12] = { 'A', 'K', 'L', 'Z', '~', '?', 'Q', 'd', '=', 0xCC, 0xCC, 0xCD };   // "AKLZ~?Qd=\xCC\xCC\xCD"  AKLZ.cs:16
const uint8_t kLzonMagic[8] = { 'L', 'Z', 'O', 'n', 0x00, 0x2F, 0xF1, 0x71 };                           // LZOn.cs:17

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
    case ALZ_C_SNAPPY: { static const uint8_t id[10] = { 0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59 }; return len > 0x10 && !memcmp(src, id, 10); }  // Snappy.cs:36-37
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
    // ---- wrappers that prepend a magic to another container / have their own small header
    switch (container) {
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
        std::vector<uint8_t> tmp(segs * slot);
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
    case ALZ_C_LEVEL5LZSS: fmt = ALZ_FMT_LZSS; hdr = 16; lz = nullptr; break;
    case ALZ_C_LZON: fmt = ALZ_FMT_LZO; hdr = 16; break;
    case ALZ_C_LEVEL5: fmt = ALZ_FMT_LZ10; hdr = 4; if (st.min_distance == 0) st.min_distance = 2; break;   // LZ10.CompressHeaderless default gbaVramCompatibilityMode = true
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
    case ALZ_C_YAZ1: memcpy(dst, "Yaz1", 4); wr32(dst + 4, (uint32_t)n, big); wr32(dst + 8, opt ? opt->memory_alignment : 0, big); wr32(dst + 12, 0, false); break;
    case ALZ_C_AKLZ: memcpy(dst, kAklzMagic, 12); wr32(dst + 12, (uint32_t)n, true); break;                                                        // AKLZ.cs:50-55
    case ALZ_C_LZ01: memcpy(dst, "LZ01", 4); wr32(dst + 4, (uint32_t)(hdr + r.dst_len), false); wr32(dst + 8, (uint32_t)n, false); wr32(dst + 12, 0, false); break;   // LZ01.cs:65-82
    case ALZ_C_LZSEGA: wr32(dst, r.dst_len, false); wr32(dst + 4, (uint32_t)n, false); break;                                                      // LZSega.cs:57-67
    case ALZ_C_LEVEL5LZSS: memcpy(dst, "SSZL", 4); wr32(dst + 4, 0, false); wr32(dst + 8, r.dst_len, false); wr32(dst + 12, (uint32_t)n, false); break;   // Level5LZSS.cs:62-72
    case ALZ_C_LZON: memcpy(dst, kLzonMagic, 8); wr32(dst + 8, (uint32_t)n, true); wr32(dst + 12, r.dst_len, true); break;                          // LZOn.cs:63-79
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
