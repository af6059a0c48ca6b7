// alz_device.h -- per-wavefront building blocks of the gfx950 LZ decode kernels.
//
// Execution model: ONE wavefront (64 lanes) decodes ONE stream.  All control
// flow is wave-uniform: parse state lives in SGPRs (readfirstlane), the 64 lanes
// cooperate on the byte work (match copy, literal runs, HBM writeback).
//
//   InCache  : sliding 2 KiB window of the compressed input in LDS, refilled by
//              coalesced 16 B/lane global loads one 1 KiB chunk ahead (prefetch
//              kept in registers so the HBM latency overlaps the parse).
//   OutWin   : the LZ sliding window in LDS == the reference's LzWindows ring
//              (IO/LzWindows.cs:15-280) without its second copy: the ring is the
//              staging buffer and is written back to HBM in 16 B/lane coalesced
//              stores as soon as a flush block completes.
//
// LDS operations of one wavefront execute in issue order, so lanes exchange data
// through LDS without s_barrier; wave_sync() only stops the compiler from moving
// LDS accesses across the exchange point.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define ALZ_WAVE 64
#define ALZ_WIN_SLACK 32u   /* bytes behind an LDS ring that mirror its head (chunked byte phase) */

__device__ __forceinline__ u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ---------------------------------------------------------------- input cache
// Coordinates: a = p + ishift, ishift = (address of src) & 15, so `a` is congruent to the
// global address mod 16 and every 16 B granule is aligned in both LDS and HBM.
struct InCache {
    const u8* gbase;  // src - ishift  (16 B aligned)
    u8* lds;          // 2 * ch bytes, 16 B aligned
    u32 lo, hi;       // valid a-range [lo, hi)
    u32 cb;           // a-coordinate of lds[0]; multiple of ch
    u32 ch;           // chunk size: 1024 (16 B per lane and load), 512 (8 B: the queue kernels) or 256 (4 B per lane: 544 B of LDS instead of 2 080 --
                      // what lets the flag-family kernels keep 32 waves per CU)
    uint4 pf;         // this lane's part of the chunk at cb + 2 ch (prefetched; .x only when ch == 256)
    int lane;

    __device__ __forceinline__ uint4 load_chunk(u32 ca) const {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ch == 1024u) {
            u32 ga = ca + 16u * (u32)lane;
            if (ga + 16u > lo && ga < hi) v = *reinterpret_cast<const uint4*>(gbase + ga);
        } else if (ch == 512u) {
            u32 ga = ca + 8u * (u32)lane;
            if (ga + 8u > lo && ga < hi) { const uint2 t = *reinterpret_cast<const uint2*>(gbase + ga); v.x = t.x; v.y = t.y; }
        } else {
            u32 ga = ca + 4u * (u32)lane;
            if (ga + 4u > lo && ga < hi) v.x = *reinterpret_cast<const u32*>(gbase + ga);
        }
        return v;
    }
    __device__ __forceinline__ void store_chunk(u32 half, uint4 v) {
        if (ch == 1024u) *reinterpret_cast<uint4*>(lds + half * 1024u + 16 * lane) = v;
        else if (ch == 512u) *reinterpret_cast<uint2*>(lds + half * 512u + 8 * lane) = make_uint2(v.x, v.y);
        else *reinterpret_cast<u32*>(lds + half * 256u + 4 * lane) = v.x;
    }
    __device__ __forceinline__ void init(const u8* src, u32 len, u8* lds_, int lane_, u32 chunk = 1024u) {
        u32 ishift = (u32)(reinterpret_cast<uintptr_t>(src) & 15u);
        gbase = src - ishift; lds = lds_; lo = ishift; hi = ishift + len; lane = lane_; cb = 0; ch = chunk;
        uint4 c0 = load_chunk(0), c1 = load_chunk(ch);
        pf = load_chunk(2u * ch);
        store_chunk(0, c0); store_chunk(1, c1);
        wave_sync();
    }
    // init + seek(p) without the loads of offset 0 (a wavefront that takes a stream over in the middle: alz_decode_fastq_kernel)
    __device__ __forceinline__ void init_at(const u8* src, u32 len, u8* lds_, int lane_, u32 chunk, u32 p) {
        u32 ishift = (u32)(reinterpret_cast<uintptr_t>(src) & 15u);
        gbase = src - ishift; lds = lds_; lo = ishift; hi = ishift + len; lane = lane_; cb = 0; ch = chunk;
        seek(p);
    }
    // reposition so that lds[0] is the chunk containing input offset p (used by seeks)
    __device__ __forceinline__ void seek(u32 p) {
        u32 a = p + lo;
        cb = a & ~(ch - 1u);
        uint4 c0 = load_chunk(cb), c1 = load_chunk(cb + ch);
        pf = load_chunk(cb + 2u * ch);
        wave_sync();
        store_chunk(0, c0); store_chunk(1, c1);
        wave_sync();
    }
    __device__ __forceinline__ void advance() {
        wave_sync();
        if (ch == 1024u) {
            uint4 up = *reinterpret_cast<const uint4*>(lds + 1024 + 16 * lane);
            *reinterpret_cast<uint4*>(lds + 16 * lane) = up;
        } else if (ch == 512u) {
            uint2 up = *reinterpret_cast<const uint2*>(lds + 512 + 8 * lane);
            *reinterpret_cast<uint2*>(lds + 8 * lane) = up;
        } else {
            u32 up = *reinterpret_cast<const u32*>(lds + 256 + 4 * lane);
            *reinterpret_cast<u32*>(lds + 4 * lane) = up;
        }
        store_chunk(1, pf);
        cb += ch;
        pf = load_chunk(cb + 2u * ch);
        wave_sync();
    }
    // make [p, p+need) resident (need <= ch); p is wave-uniform and only moves forward
    __device__ __forceinline__ void ensure(u32 p, u32 need) {
        u32 a = p + lo;
        if (a - cb >= 3u * ch) { seek(p); return; }   // far jump (never on the sequential path)
        while (a + need > cb + 2u * ch) advance();
    }
    __device__ __forceinline__ u32 idx(u32 p) const { return p + lo - cb; }
    // per-lane byte at input offset p (p may differ per lane); caller guarantees residency
    __device__ __forceinline__ u32 byte_at(u32 p) const { return lds[idx(p)]; }
    // wave-uniform little-endian dword at p (bytes past the input end are unspecified)
    __device__ __forceinline__ u32 peek4(u32 p) const {
        u32 i = idx(p);
        const u32* w = reinterpret_cast<const u32*>(lds + (i & ~3u));
        u32 lo32 = w[0], hi32 = w[1];   // lds has 8 guard bytes behind it
        u32 v = __builtin_amdgcn_alignbyte(hi32, lo32, i & 3u);
        return uni(v);
    }
    __device__ __forceinline__ u32 peek1(u32 p) const { return uni((u32)lds[idx(p)]); }
};

// ---------------------------------------------------------------- output window
// LW  = bytes of window kept in LDS (power of two).  For the 4/8 KiB formats LW == W,
//       for the 64 KiB formats LW < W and sources older than LW are read back from
//       the stream's own output in HBM (L2-served loads).
// Coordinates: ao = q + oshift, oshift = (address of dst) & 15.
template <bool GLOBAL_FALLBACK>
struct OutWin {
    static constexpr bool FB = GLOBAL_FALLBACK;
    static constexpr bool PUBLISH = false;   // (see WalkOut in alz_emit_byte.h)
    u8* dst;         // global output of this stream
    u8* win;         // LDS, LW bytes, 16 B aligned
    u32 lw_mask;     // LW - 1
    u32 fl;          // flush block (power of two, <= LW/4, >= 16)
    u32 oshift;
    u32 cap;         // dst_cap
    u32 produced;    // bytes decoded so far (wave-uniform)
    u32 flushed;     // bytes already stored to HBM (wave-uniform)
    int lane;
    // The chunked byte phase (alz_emit_chunk.h) moves dword-aligned pieces of up to 24 bytes, which may run past the end of
    // the ring: kernels that use it allocate ALZ_WIN_SLACK bytes behind the ring that mirror its first bytes.  The byte-wise
    // writers below only touch the ring proper and mark the mirror stale; the chunked phase refreshes it before it reads.
    bool slack_dirty;
    u32 mtag;        // tag of the byte phase's current mark block (alz_emit_byte.h; wave-uniform, 0 = none yet)

    __device__ __forceinline__ void init(u8* dst_, u32 cap_, u8* win_, u32 lw, int lane_, u32 slack = 0) {
        dst = dst_; cap = cap_; win = win_; lw_mask = lw - 1; lane = lane_;
        fl = lw >= 4096 ? 1024u : (lw >> 2);
        oshift = (u32)(reinterpret_cast<uintptr_t>(dst_) & 15u);
        produced = 0; flushed = 0; slack_dirty = false; mtag = 0;
        // E2: the reference's rented ring is treated as zero-filled
        for (u32 i = 16u * (u32)lane; i < lw + slack; i += 16u * ALZ_WAVE) *reinterpret_cast<uint4*>(win + i) = make_uint4(0, 0, 0, 0);
        wave_sync();
    }
    __device__ __forceinline__ u32 slot(u32 q) const { return (q + oshift) & lw_mask; }

    // History (LZ4 frames with linked blocks, LZ4.Frame.cs:120: one LzWindows for all blocks of a frame): the caller
    // passed dst - hist as `dst`; the first `hist` bytes are output of earlier blocks and count as already produced.
    // The most recent LW of them are loaded into the ring, older ones are reached through the HBM read-back path.
    __device__ void preload(u32 hist) {
        const u32 lw = lw_mask + 1;
        const u32 n = hist < lw ? hist : lw;
        for (u32 i = (u32)lane; i < n; i += ALZ_WAVE) { const u32 pos = hist - n + i; win[slot(pos)] = dst[pos]; }
        wave_sync();
        produced = hist; flushed = hist; slack_dirty = true;
    }

    // store [flushed, limit) to HBM; 16 B granules aligned in LDS and HBM, ragged ends bytewise
    __device__ void flush_to(u32 limit) {
        wave_sync();
        u32 a0 = flushed + oshift, a1 = limit + oshift;
        u8* gb = dst - oshift;
        for (u32 g = (a0 & ~15u) + 16u * (u32)lane; g < a1; g += 16u * ALZ_WAVE) {
            if (g >= a0 && g + 16u <= a1) {
                uint4 v = *reinterpret_cast<const uint4*>(win + (g & lw_mask));
                *reinterpret_cast<uint4*>(gb + g) = v;
            } else {
                u32 b0 = g < a0 ? a0 : g, b1 = g + 16u < a1 ? g + 16u : a1;
                for (u32 b = b0; b < b1; b++) gb[b] = win[b & lw_mask];
            }
        }
        flushed = limit;
        if (GLOBAL_FALLBACK) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // stores reach L2 before read-back
    }
    // flush every complete block; keeps produced - flushed < fl
    __device__ __forceinline__ void flush_blocks() {
        u32 lim = ((produced + oshift) & ~(fl - 1));
        if (lim > oshift && lim - oshift > flushed) flush_to(lim - oshift);
    }
    __device__ __forceinline__ void finish() { if (produced > flushed) flush_to(produced); }

    // one literal byte (wave-uniform value)
    __device__ __forceinline__ void put_byte(u32 b) {
        if (lane == 0) win[slot(produced)] = (u8)b;
        produced += 1; slack_dirty = true;
        if (((produced + oshift) & (fl - 1)) == 0) flush_blocks();
    }

    // BackCopy(distance, length)  IO/LzWindows.cs:72-100 as out[q] = out[q-d]; `dw` = W of the format (E1).
    // len is already clipped against cap.  Period doubling: once `done` bytes exist, the next
    // min(rem, done + d) bytes are a copy shifted by P = done + d (a multiple of d), all sources < produced.
    __device__ void back_copy(u32 d, u32 len, u32 dw) {
        if (d == 0) d = dw;                                   // E1
        slack_dirty = true;
        const u32 lw = lw_mask + 1;
        u32 done = 0, P = d;                                  // P: multiple of d, P <= done + d
        while (done < len) {
            u32 span = len - done; if (span > P) span = P;
            // pieces of <= fl bytes so that unflushed data is never overwritten
            u32 off = 0;
            while (off < span) {
                u32 n = span - off; if (n > fl) n = fl;
                u32 c = produced;
                for (u32 j = (u32)lane; j < n; j += ALZ_WAVE) {
                    u32 q = c + j;
                    u32 v = 0;                                       // E2: before the stream start
                    if (P <= q) {
                        u32 sp = q - P;
                        // slots of [c-lw, c+n-lw) are overwritten during this piece: those sources are
                        // already flushed (n + fl <= lw) and come back from HBM through L2
                        if (GLOBAL_FALLBACK && sp + lw < c + n) v = (u32)__hip_atomic_load(dst + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else v = win[slot(sp)];
                    }
                    win[slot(q)] = (u8)v;
                }
                wave_sync();
                produced = c + n; off += n;
                flush_blocks();
            }
            done += span;
            // period doubling: after a full span the copy shift may double (still a multiple of d, <= done + d);
            // without the HBM fallback the shift must stay inside the LDS window
            if (span == P && (GLOBAL_FALLBACK || 2 * P <= lw)) P *= 2;
        }
    }

    // literal run that is already resident in LDS (a queued token of the lane-parallel parsers: `src` points into the input cache)
    __device__ __forceinline__ void copy_lds(const u8* src, u32 len) {
        u32 off = 0;
        slack_dirty = true;
        while (off < len) {
            u32 n = len - off; if (n > fl) n = fl;
            u32 c = produced;
            for (u32 j = (u32)lane; j < n; j += ALZ_WAVE) win[slot(c + j)] = src[off + j];
            wave_sync();
            produced = c + n; off += n;
            flush_blocks();
        }
    }

    // literal run straight from the input cache (LzWindows.CopyFrom / Write); len clipped, input residency handled here
    __device__ void copy_from(InCache& in, u32 p, u32 len) {
        u32 off = 0;
        slack_dirty = true;
        while (off < len) {
            u32 n = len - off; if (n > fl) n = fl; if (n > in.ch) n = in.ch;
            in.ensure(p + off, n);
            u32 c = produced;
            for (u32 j = (u32)lane; j < n; j += ALZ_WAVE) win[slot(c + j)] = (u8)in.byte_at(p + off + j);
            wave_sync();
            produced = c + n; off += n;
            flush_blocks();
        }
    }
};
