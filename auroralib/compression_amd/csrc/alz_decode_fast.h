// alz_decode_fast.h -- lane-parallel decode of the flag-byte LZSS family (LZSS / LZ10 / LZ11 / Yaz0 interleaved,
// Yay0 / MIO0 three-cursor).  One wavefront per stream; each loop iteration handles up to 8 flag groups = 64 tokens:
//
//   front end (interleaved formats)
//     1. 128 input bytes from the LDS input cache, two per lane.
//     2. wave ballots give the per-byte "3-byte / 4-byte token starts here" masks (the only data a token's SIZE
//        depends on besides its flag bit).
//     3. every lane speculatively walks the 8 tokens of "a group that starts at my byte" (pure VALU, no memory),
//        producing that group's size and the 8 token offsets.
//     4. the real group chain (<= 8 hops through v_readlane) picks the lanes whose speculation was real.
//     5. lane 8j+k becomes token k of group j: fetches its offset from the speculating lane (ds_bpermute), reads its
//        2-4 token bytes and decodes (length, distance | literal).
//   front end (three-cursor formats): token/literal cursors are prefix popcounts of the flag bits (mbcnt).
//   back end (shared)
//     6. wave prefix sum of lengths -> output offsets; size / capacity rules (E4, E5) become prefix cuts.
//     7. byte phase: 64 output bytes per step, one per lane; a lane finds its token through a 64-entry LDS mark
//        array + ballot/mbcnt, reads its source byte from the LDS window, resolves sources that are produced inside
//        the same 64-byte step by pointer jumping over ds_bpermute, writes the window; 1 KiB blocks are written back
//        to HBM coalesced as they complete.
//
// The serial decoders (alz_decode_serial.h) remain the exact reference: the fast loop only runs while at least 128
// input bytes remain (so no token can be truncated) and hands the stream tail to them.
#pragma once
#include "alz_decode_serial.h"

__device__ __forceinline__ u32 wave_bperm(u32 src_lane, u32 v) { return (u32)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
__device__ __forceinline__ u32 wave_readlane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u32 mbcnt64(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }

// inclusive prefix sum over the 64 lanes: DPP row shifts inside the 16-lane rows, then row broadcasts (gfx9 DPP)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_add(u32 v) {
    return v + (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ u32 wave_incl_scan(u32 v, int lane) {
    (void)lane;
    v = dpp_add<0x111, 0xF>(v);   // row_shr:1
    v = dpp_add<0x112, 0xF>(v);   // row_shr:2
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

// descriptor of a token for the byte phase: bit31 literal, bits 17..24 literal value, bits 0..16 distance
#define ALZ_DESC_LIT(b) (0x80000000u | ((b) << 17))
#define ALZ_DESC_MATCH(d) (d)
#define ALZ_DESC_DIST(x) ((x) & 0x1FFFFu)

struct FastGeom {           // LZSS geometry (other formats ignore it)
    u32 length_bits, min_length, windows_start, max_distance, W;
};

// One step of the byte phase: 64 consecutive output bytes, one per lane.  The kernel is bound by VALU issue (one wave
// instruction per 4 cycles per SIMD), so the step is written for the fewest vector instructions:
//   * token lanes mark the lane where their output ENDS inside the step; the token of byte L is then
//     (#tokens ended before the step) + (#marks below L): one mbcnt pair, fused with the x4 of the bpermute address;
//   * a match descriptor IS its distance (literal descriptors have bit 31 set), so the source slot is
//     (slot - descriptor) & mask and "source inside this very step" is the unsigned test descriptor <= lane;
//   * pointer jumping only runs in steps where that test fires for some lane;
//   * EARLY (first W bytes of a stream: sources may lie before the stream start, E2) and !FULL (last, partial step)
//     are separate instantiations so the steady state does not pay for them.
template <class OW, u32 OMASK, bool EARLY, bool FULL>
__device__ __forceinline__ void byte_step(OW& out, u8* segmark, int lane, u32 desc, u32& relm, u32& qs, u32& tbase4, u32 nseg) {
    const u32 omask = OMASK ? OMASK : out.lw_mask;
    u8* const win = out.win;
    segmark[relm < 64u ? relm : 64u + (u32)lane] = 1;
    wave_sync();
    const u32 mk = segmark[lane];
    segmark[lane] = 0;
    const u64 M = __ballot(mk != 0);
    const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(M >> 32), __builtin_amdgcn_mbcnt_lo((u32)M, 0u));
    const u32 dsc = (u32)__builtin_amdgcn_ds_bpermute((int)((cnt << 2) + tbase4), (int)desc);   // match: the distance; literal: bit31 | value << 17
    tbase4 += 4u * (u32)__popcll(M);
    u32 wv = win[(qs - dsc) & omask];                        // source byte (garbage for literals, never used)
    if (EARLY) { if (dsc > qs - out.oshift) wv = 0; }        // E2: before the stream start
    u32 val = ((int)dsc < 0) ? ((dsc >> 17) & 0xFFu) : wv;
    const bool instep = FULL ? (dsc <= (u32)lane) : (dsc <= (u32)lane && (u32)lane < nseg);   // source produced inside this very step
    if (__ballot(instep)) {
        // pointer jumping (at most 6 rounds), one packed ds_bpermute per round: value | source lane << 8, 0x40 = resolved
        u32 st = val | ((instep ? ((u32)lane - dsc) : 0x40u) << 8);
        do {
            const u32 f = wave_bperm(st >> 8, st);
            if (st < 0x4000u) st = (f >= 0x4000u) ? f : ((st & 0xFFu) | (f & 0xFF00u));
        } while (__ballot(st < 0x4000u));
        val = st & 0xFFu;
    }
    if (FULL) win[qs & omask] = (u8)val;
    else if ((u32)lane < nseg) win[qs & omask] = (u8)val;
    wave_sync();
    qs += 64u; relm -= 64u;
}

// Shared back end.  Per-lane token: valid, len (>=1), desc, tend = input offset just past the token (relative to the
// iteration's base).  For LZSS the descriptor holds the ring OFFSET and is turned into a distance here, once the
// token's output position is known (LzWindows.OffsetCopy  IO/LzWindows.cs:108-115).
// Returns true when the stream is finished (declared size reached, or capacity hit).
template <class OW, bool LZSS, u32 OMASK>
__device__ __forceinline__ bool fast_emit(OW& out, DecState& s, u32 size, bool valid, u32 len, u32 desc, u32 tend, u8* segmark,
                                          int lane, u32& last_tend, u32 W) {
    u32 end = wave_incl_scan(valid ? len : 0u, lane);
    u32 off = end - len;
    const u32 O = out.produced;
    const u32 left = size - O;                               // > 0 (caller guarantees produced < size)
    bool keep = valid && off < left;                         // the token exists in the stream (prefix of lanes)
    u64 km = __ballot(keep);
    u32 nk = (u32)__popcll(km);
    u32 nvalid = (u32)__popcll(__ballot(valid));
    u32 Tend = wave_readlane(end, nk - 1);
    bool fin = nk < nvalid || Tend >= left;
    u32 lastk = nk - 1;
    u32 T = Tend;
    const u32 room = out.cap - O;
    if (Tend > room) {                                       // E5: first token whose output would exceed dst_cap
        u64 om = __ballot(keep && end > room);
        lastk = (u32)__builtin_ctzll(om);
        s.ovf = true; s.attempted_end = (u64)O + wave_readlane(end, lastk);
        T = room; fin = true;
    }
    last_tend = wave_readlane(tend, lastk);
    if (LZSS) {
        if (!(desc >> 31)) {
            u32 offset = ALZ_DESC_DIST(desc);
            u32 pos = (O + off) & (W - 1);
            u32 d = (pos - offset) & (W - 1);
            if (d == 0) d = W;                               // E1
            desc = ALZ_DESC_MATCH(d);
        }
    }
    // ---- byte phase (byte_step below): 64 output bytes per step, one per lane
    u32 tbase4 = 0;                                          // 4 x (tokens that ended before the current step)
    u32 relm = keep ? end - 1u : 0xFFFFFF00u;                // my token's LAST byte relative to the current step (huge: none)
    u32 qs = O + (u32)lane + out.oshift;                     // slot coordinate of this lane's byte in the current step
    u32 X = 0;
    // steps that may still point before the stream start (E2) -- only inside the first W bytes of a stream
    while (X + 64u <= T && O + X < W) { byte_step<OW, OMASK, true, true>(out, segmark, lane, desc, relm, qs, tbase4, 64u); X += 64u; out.produced = O + X; if (out.produced - out.flushed >= out.fl) out.flush_blocks(); }
    while (X + 64u <= T) { byte_step<OW, OMASK, false, true>(out, segmark, lane, desc, relm, qs, tbase4, 64u); X += 64u; out.produced = O + X; if (out.produced - out.flushed >= out.fl) out.flush_blocks(); }
    if (X < T) { byte_step<OW, OMASK, true, false>(out, segmark, lane, desc, relm, qs, tbase4, T - X); out.produced = O + T; if (out.produced - out.flushed >= out.fl) out.flush_blocks(); }
    return fin;
}

template <int FMT> struct FamTraits;
template <> struct FamTraits<ALZ_FMT_LZSS> { static constexpr bool MSB = false, LIT1 = true,  H3 = false, H4 = false; };
template <> struct FamTraits<ALZ_FMT_LZ10> { static constexpr bool MSB = true,  LIT1 = false, H3 = false, H4 = false; };
template <> struct FamTraits<ALZ_FMT_LZ11> { static constexpr bool MSB = true,  LIT1 = false, H3 = true,  H4 = true; };
template <> struct FamTraits<ALZ_FMT_YAZ0> { static constexpr bool MSB = true,  LIT1 = true,  H3 = true,  H4 = false; };

// bits [i, i+32) of the 128-bit mask hi:lo, i = 0..63
__device__ __forceinline__ u32 mask_window(u64 lo, u64 hi, int i) {
    u64 w = (lo >> i) | ((hi << (63 - i)) << 1);
    return (u32)w;
}

// Interleaved formats.  Precondition: s.bits == 0 (group boundary), s.p + 128 <= src_len, out.produced < size.
template <int FMT, class OW>
__device__ __forceinline__ bool fast_iter_interleaved(InCache& in, OW& out, DecState& s, u32 size, u8* segmark, int lane, const FastGeom& gm) {
    typedef FamTraits<FMT> TR;
    const u32 p = s.p;
    in.ensure(p, 128);
    const u32 x0 = in.byte_at(p + (u32)lane), x1 = in.byte_at(p + 64u + (u32)lane);
    u32 l3 = 0, l4 = 0;
    if (TR::H3) { u64 lo = __ballot((x0 >> 4) == 0), hi = __ballot((x1 >> 4) == 0); l3 = mask_window(lo, hi, lane); }
    if (TR::H4) { u64 lo = __ballot((x0 >> 4) == 1), hi = __ballot((x1 >> 4) == 1); l4 = mask_window(lo, hi, lane); }
    // speculative walk of "the group that starts at byte p + lane"
    const u32 mbits = TR::LIT1 ? (~x0 & 0xFFu) : x0;          // bit set = match token
    u32 r = 1, rp0 = 0, rp1 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 m = (mbits >> (TR::MSB ? 7 - k : k)) & 1u;
        u32 extra = m;
        if (TR::H3) extra += m & (l3 >> r);
        if (TR::H4) extra += (m & (l4 >> r)) << 1;
        if (k < 4) rp0 |= (r << (6 * k)) | (m << (24 + k)); else rp1 |= (r << (6 * (k - 4))) | (m << (24 + k - 4));
        r += 1u + extra;
    }
    const u32 gsize = r;
    // real group chain
    u32 g = 0, ng = 0, gstart = 0;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        if (g < 64u) {
            if ((lane >> 3) == it) gstart = g;
            g += wave_readlane(gsize, g);
            ng = (u32)it + 1u;
        }
    }
    // lane 8j+k = token k of group j
    const u32 k = (u32)lane & 7u;
    const bool valid = ((u32)lane >> 3) < ng;
    const u32 w0 = wave_bperm(gstart, rp0), w1 = wave_bperm(gstart, rp1);
    const u32 w = k < 4 ? w0 : w1;
    const u32 kk = k & 3u;
    const u32 to = gstart + ((w >> (6 * kk)) & 63u);
    const u32 m = (w >> (24 + kk)) & 1u;
    const u32 ti = in.idx(p + to);
    const u32 b1 = in.lds[ti], b2 = in.lds[ti + 1];
    u32 len = 1, desc = ALZ_DESC_LIT(b1), tend = to + 1;
    if (m) {
        if (FMT == ALZ_FMT_LZSS) {
            u32 offset = ((b2 >> gm.length_bits) << 8) | b1;
            len = (b2 & ((1u << gm.length_bits) - 1u)) + gm.min_length;
            offset = (gm.max_distance + offset - gm.windows_start) & (gm.max_distance - 1u);
            desc = ALZ_DESC_MATCH(offset); tend = to + 2;
        } else if (FMT == ALZ_FMT_LZ10) {
            desc = ALZ_DESC_MATCH((((b1 & 0xFu) << 8) | b2) + 1u); len = (b1 >> 4) + 3u; tend = to + 2;
        } else if (FMT == ALZ_FMT_LZ11) {
            const u32 b3 = in.lds[ti + 2], b4 = in.lds[ti + 3];
            const u32 nib = b1 >> 4;
            if (nib == 0) { desc = ALZ_DESC_MATCH((((b2 & 0xFu) << 8) | b3) + 1u); len = (((b1 & 0xFu) << 4) | (b2 >> 4)) + 17u; tend = to + 3; }
            else if (nib == 1) { desc = ALZ_DESC_MATCH((((b3 & 0xFu) << 8) | b4) + 1u); len = (((b1 & 0xFu) << 12) | (b2 << 4) | (b3 >> 4)) + 273u; tend = to + 4; }
            else { desc = ALZ_DESC_MATCH((((b1 & 0xFu) << 8) | b2) + 1u); len = nib + 1u; tend = to + 2; }
        } else {  // YAZ0
            const u32 b3 = in.lds[ti + 2];
            const u32 nib = b1 >> 4;
            desc = ALZ_DESC_MATCH((((b1 & 0xFu) << 8) | b2) + 1u);
            if (nib == 0) { len = b3 + 0x12u; tend = to + 3; } else { len = nib + 2u; tend = to + 2; }
        }
    }
    u32 last_tend;
    const bool fin = fast_emit<OW, FMT == ALZ_FMT_LZSS, (FMT == ALZ_FMT_LZSS ? 0u : 4095u)>(out, s, size, valid, len, desc, tend, segmark, lane, last_tend, gm.W);
    s.p = p + (fin ? last_tend : g);
    return fin;
}

// Three-cursor formats (Yay0 / MIO0).  Precondition: s.bits == 0, fp + 8 <= src_len, cp + 128 <= src_len, up + 64 <= src_len.
template <bool MIO0, class OW>
__device__ __forceinline__ bool fast_iter_3cursor(InCache& fin_, InCache& cin, InCache& uin, OW& out, DecState& s, u32 size, u8* segmark,
                                                  int lane, u32& fp, u32& cp, u32& up) {
    fin_.ensure(fp, 8); cin.ensure(cp, 128); uin.ensure(up, 64);
    const u32 fb = fin_.byte_at(fp + ((u32)lane >> 3));
    const bool lit = (fb >> (7 - (lane & 7))) & 1u;                // MSB-first, bit 1 = literal  Yay0.cs:118, MIO0.cs:123
    const u64 lm = __ballot(lit);
    const u32 midx = mbcnt64(~lm);                                  // matches before me
    u32 b1 = 0, b2 = 0;
    if (!lit) { const u32 ci = cin.idx(cp + 2u * midx); b1 = cin.lds[ci]; b2 = cin.lds[ci + 1]; }
    bool usesu = lit;
    if (!MIO0) usesu = lit || (b1 >> 4) == 0;                       // Yay0: 3-byte token takes its length from the literal stream
    const u64 um = __ballot(usesu);
    const u32 uidx = mbcnt64(um);
    u32 ub = 0;
    if (usesu) ub = uin.byte_at(up + uidx);
    u32 len = 1, desc = ALZ_DESC_LIT(ub);
    if (!lit) {
        desc = ALZ_DESC_MATCH((((b1 & 0xFu) << 8) | b2) + 1u);
        if (MIO0) len = (b1 >> 4) + 3u;
        else len = (b1 >> 4) ? (b1 >> 4) + 2u : ub + 0x12u;
    }
    // cursors after this token, packed so one readlane recovers both (c: 8 bits is enough for <=128, u: <=64)
    const u32 tend = ((2u * (midx + (lit ? 0u : 1u))) << 8) | (uidx + (usesu ? 1u : 0u));
    u32 last;
    const bool fin = fast_emit<OW, false, 4095u>(out, s, size, true, len, desc, tend, segmark, lane, last, 4096);
    if (fin) { cp += last >> 8; up += last & 0xFFu; }
    else { cp += 2u * (u32)__popcll(~lm); up += (u32)__popcll(um); fp += 8; }
    return fin;
}
