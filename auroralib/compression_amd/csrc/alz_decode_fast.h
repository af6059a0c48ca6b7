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
// The serial decoders (alz_decode_serial.h) remain the exact reference: the fast loop runs to the last complete token of
// the input and hands the stream tail -- and every error path -- to them.
//
// (Also here: the lane-parallel iterations of SMSR00 and LZHudson; BLZ runs LZ10's iteration with its own distance bias.)
//
// Second half of the file: the token queue (QueueSink) that lets the byte phase execute tokens of ANY grammar, and the
// lane-assisted parsers that fill it for the grammars whose token boundaries can only be found by walking the stream:
// LZ4 / Snappy / LZO / FastLZ (per-byte speculation "the element that would start here" + a scalar v_readlane walk that
// hands element j to lane j), CNX2 (the same with whole flag-byte groups as elements) and PRS (per-byte token
// interpretations + a scalar walk over a VGPR-resident window with the flag register in an SGPR).  pipelined_rounds
// executes the rounds of the first five, overlapping the HBM read-backs of one round with the parse of the next.
#pragma once
#ifndef ALZ_DESCTAB_ALL
#define ALZ_DESCTAB_ALL 0
#endif
// The descriptor table (EmitCfg::DESCTAB) with 512-byte input-cache chunks for the single-cursor formats it pays for: behind the work queue of chunks Yaz0 2.58 -> 2.55 ms
// (two batches in flight 980 -> 996 GiB/s), LZ11 2.636 -> 2.620; LZ10 and LZSS lose a few tenths of a percent and keep 1 024-byte chunks without the table (EXP 10.3).
#ifndef ALZ_TAB512
#define ALZ_TAB512 1
#endif
#include "alz_emit_byte.h"
#include "alz_prs_table.h"
template <int FMT> struct alz_tab512 { static constexpr bool value = ALZ_DESCTAB_ALL != 0 || (ALZ_TAB512 != 0 && (FMT == ALZ_FMT_YAZ0 || FMT == ALZ_FMT_LZ11)); };

#ifndef ALZ_QRUN
#define ALZ_QRUN 200u   /* longest literal run / element a lane-parallel round takes: window (256) + element stay inside one 512-byte cache chunk */
#endif
#ifndef ALZ_WIDE_FLAT
#define ALZ_WIDE_FLAT 1   /* groups start in the first 128 (not 64) input bytes of an iteration: formats without a group walk (LZSS, LZ10, CLZ0, BLZ) */
#endif
#ifndef ALZ_WIDE_WALK
#define ALZ_WIDE_WALK 1   /* ... and formats with one (LZ11, Yaz0, LZ40, LZ02): a second speculative walk per lane */
#endif
#ifndef ALZ_NB
#define ALZ_NB 8    /* steps whose HBM read-backs are issued together (two-pass byte phase of the 64 KiB formats) */
#endif

// v_writelane_b32: write a wave-uniform value into one lane of a VGPR (no clang builtin in ROCm 7.2).  gfx9 allows one
// SGPR on the constant bus, so the lane select travels in M0 (what LLVM's own lowering of llvm.amdgcn.writelane does).
__device__ __forceinline__ u32 wave_writelane(u32 old, u32 val, u32 lane) {
    u32 tmp;   // the value goes through an SGPR (v_writelane takes no literal constants)
    asm volatile("s_mov_b32 %1, %2\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(old), "=&s"(tmp) : "s"(val), "s"(lane) : "m0");
    return old;
}

struct FastGeom {           // LZSS geometry (other formats ignore it)
    u32 length_bits, min_length, windows_start, max_distance, W;
};

template <int FMT> struct FamTraits;
// MSB: flag bits MSB first; LIT1: flag bit 1 = literal; H3 / H4: a match whose size nibble is 0 / 1 has 3 / 4 bytes; NIBLO: that
// nibble is the low one of the first token byte (LZ40) instead of the high one; NEG: the flag byte is stored negated (LZ40)
template <> struct FamTraits<ALZ_FMT_LZSS> { static constexpr bool MSB = false, LIT1 = true,  H3 = false, H4 = false, NIBLO = false, NEG = false; };
template <> struct FamTraits<ALZ_FMT_LZ10> { static constexpr bool MSB = true,  LIT1 = false, H3 = false, H4 = false, NIBLO = false, NEG = false; };
template <> struct FamTraits<ALZ_FMT_LZ02> { static constexpr bool MSB = true,  LIT1 = false, H3 = true,  H4 = false, NIBLO = true,  NEG = false; };   // (the 2-byte terminator also has nibble 0: nothing behind it counts)
template <> struct FamTraits<ALZ_FMT_CLZ0> { static constexpr bool MSB = false, LIT1 = false, H3 = false, H4 = false, NIBLO = false, NEG = false; };
template <> struct FamTraits<ALZ_FMT_BLZ>  { static constexpr bool MSB = true,  LIT1 = false, H3 = false, H4 = false, NIBLO = false, NEG = false; };   // LZ10's grammar in stream order, distance - 3
template <> struct FamTraits<ALZ_FMT_LZ11> { static constexpr bool MSB = true,  LIT1 = false, H3 = true,  H4 = true,  NIBLO = false, NEG = false; };
template <> struct FamTraits<ALZ_FMT_YAZ0> { static constexpr bool MSB = true,  LIT1 = true,  H3 = true,  H4 = false, NIBLO = false, NEG = false; };
template <> struct FamTraits<ALZ_FMT_LZ40> { static constexpr bool MSB = true,  LIT1 = false, H3 = true,  H4 = true,  NIBLO = true,  NEG = true; };

// bits [i, i+32) of the 128-bit mask hi:lo, i = 0..63
__device__ __forceinline__ u32 mask_window(u64 lo, u64 hi, int i) {
    u64 w = (lo >> i) | ((hi << (63 - i)) << 1);
    return (u32)w;
}

// Interleaved formats.  Precondition: s.bits == 0 (group boundary), s.p + 128 <= src_len, out.produced < size.
template <int FMT, class OW>
__device__ __forceinline__ bool fast_iter_interleaved(InCache& in, OW& out, DecState& s, u32 size, u32 src_len, bool& to_serial, u8* segmark, int lane, const FastGeom& gm) {
    typedef FamTraits<FMT> TR;
    constexpr bool WALK = TR::H3 || TR::H4;
    // WIDE: groups may start anywhere in the first 128 input bytes instead of the first 64 (every lane speculates for two
    // bytes).  With ~13 input bytes per group only five of the eight group slots were filled per iteration; the fixed cost of
    // an iteration (chain, token decode, prologue) is now spread over all eight.
    constexpr bool WIDE = WALK ? (ALZ_WIDE_WALK != 0) : (ALZ_WIDE_FLAT != 0);
    const u32 p = s.p;
    in.ensure(p, WIDE ? 224 : 128);
    const u32 x0 = in.byte_at(p + (u32)lane), x1 = in.byte_at(p + 64u + (u32)lane);
    u32 l3 = 0, l4 = 0, l3b = 0, l4b = 0;
    if (WALK) {
        const u32 n0 = TR::NIBLO ? (x0 & 0xFu) : (x0 >> 4), n1 = TR::NIBLO ? (x1 & 0xFu) : (x1 >> 4);
        u32 n2 = 0;
        if (WIDE) { const u32 x2 = in.byte_at(p + 128u + (u32)lane); n2 = TR::NIBLO ? (x2 & 0xFu) : (x2 >> 4); }
        if (TR::H3) { const u64 lo = __ballot(n0 == 0), hi = __ballot(n1 == 0); l3 = mask_window(lo, hi, lane); if (WIDE) l3b = mask_window(hi, __ballot(n2 == 0), lane); }
        if (TR::H4) { const u64 lo = __ballot(n0 == 1), hi = __ballot(n1 == 1); l4 = mask_window(lo, hi, lane); if (WIDE) l4b = mask_window(hi, __ballot(n2 == 1), lane); }
    }
    // speculative walk of "the group that starts at byte xb": group size + what a token lane needs to find its
    // offset.  Formats whose token size depends only on the flag bit (LZSS, LZ10) need no walk at all: the offset of token
    // k is 1 + k + popcount(match bits before k).  Otherwise `info` packs (token size - 1) as 8 nibbles and a token lane
    // sums the nibbles below its own with one v_dot8_u32_u4.
    auto spec = [&](u32 xb, u32 w3, u32 w4, u32& gsize, u32& info) {
        const u32 mbits = TR::NEG ? ((0u - xb) & 0xFFu) : (TR::LIT1 ? (~xb & 0xFFu) : xb);   // bit set = match token
        if (!WALK) { gsize = 9u + (u32)__popc(mbits); info = mbits; }
        else {
            // (five vector instructions per token for the formats with one odd size -- bit-field extracts of the flag bit and of the "3-byte
            // match starts at the cursor" bit, extra = m * z + m as one v_mad_u32_u24, the cursor as one v_add3, the nibble as one v_lshl_or --
            // where `m & (w3 >> r)` compiled to six + the packing of the nibbles afterwards)
            u32 r = 1; info = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 m = __builtin_amdgcn_ubfe(mbits, (u32)(TR::MSB ? 7 - k : k), 1u);
                u32 extra;
                if (TR::H3 && !TR::H4) { const u32 z = __builtin_amdgcn_ubfe(w3, r, 1u); asm("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(extra) : "v"(m), "v"(z)); }   // (left to itself the compiler writes (z + 1) & -m: one instruction more)
                else {                                              // two odd sizes (LZ11, LZ40): z = 0 / 1 / 2 extra bytes of a match that starts at the cursor
                    const u32 z = __builtin_amdgcn_ubfe(w3, r, 1u) + 2u * __builtin_amdgcn_ubfe(w4, r, 1u);
                    asm("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(extra) : "v"(m), "v"(z));
                }
                if (k == 0) info = extra;                          // (one v_lshl_or per token: left to itself the compiler shifts every nibble and ORs them in a tree, 11 instead of 7)
                else asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(info) : "v"(extra), "n"(4 * k));
                r += 1u + extra;
            }
            gsize = r;
        }
    };
    u32 gsize, info, gsize1 = 0, info1 = 0;
    spec(x0, l3, l4, gsize, info);
    if (WIDE) spec(x1, l3b, l4b, gsize1, info1);
    // real group chain (round 4 measured two other forms: the hops written with selects instead of branches -- 14 vector instructions fewer per
    // iteration, 3.17 against 2.97 ms per launch: ten dependent scalar instructions per hop instead of five, and the tail of a launch runs at
    // the latency of exactly this chain --, and the starts packed into a scalar pair with one bit-field extract per lane afterwards: 3.09 ms,
    // while two batches in flight gained 1 %)
    u32 g = 0, ng = 0, gstart = 0;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        if (g < (WIDE ? 128u : 64u)) {
            if ((lane >> 3) == it) gstart = g;
            if (WIDE) g += g < 64u ? wave_readlane(gsize, g) : wave_readlane(gsize1, g - 64u);
            else g += wave_readlane(gsize, g);
            ng = (u32)it + 1u;
        }
    }
    // lane 8j+k = token k of group j
    const u32 k = (u32)lane & 7u;
    u32 inf = wave_bperm(gstart & 63u, info);
    if (WIDE) { const u32 inf1 = wave_bperm(gstart & 63u, info1); inf = gstart < 64u ? inf : inf1; }
    u32 m, to;
    if (!TR::H3 && !TR::H4) {
        m = (inf >> (TR::MSB ? 7u - k : k)) & 1u;
        const u32 before = TR::MSB ? (u32)__popc(inf >> (8u - k)) : (u32)__popc(inf & ((1u << k) - 1u));
        to = gstart + 1u + k + before;
    } else {
        m = ((inf >> (4u * k)) & 0xFu) != 0u;
        to = gstart + 1u + k + (u32)__builtin_amdgcn_udot8(inf & ((1u << (4u * k)) - 1u), 0x11111111u, 0u, false);
    }
    const u32 ti = in.idx(p + to);
    const u32 b1 = in.lds[ti], b2 = in.lds[ti + 1];
    u32 len = 1, desc = ALZ_DESC_LIT(b1), tend = to + 1;
    bool term = false;                                           // LZ02: this token is the terminator
    if (m) {
        if (FMT == ALZ_FMT_LZSS) {
            u32 offset = ((b2 >> gm.length_bits) << 8) | b1;
            len = (b2 & ((1u << gm.length_bits) - 1u)) + gm.min_length;
            offset = (gm.max_distance + offset - gm.windows_start) & (gm.max_distance - 1u);
            desc = ALZ_DESC_MATCH(offset); tend = to + 2;
        } else if (FMT == ALZ_FMT_LZ10) {
            desc = ALZ_DESC_MATCH((((b1 & 0xFu) << 8) | b2) + 1u); len = (b1 >> 4) + 3u; tend = to + 2;
        } else if (FMT == ALZ_FMT_LZ02) {                    // LZ02.cs:88-103
            const u32 b3 = in.lds[ti + 2];
            const u32 nib = b1 & 0xFu, d = ((b1 & 0xF0u) << 4) | b2;
            desc = ALZ_DESC_MATCH(d ? d : 4096u);                // E1
            if (nib == 0) { len = b3 + 17u; tend = to + 3; term = d == 0u; } else { len = nib + 1u; tend = to + 2; }
        } else if (FMT == ALZ_FMT_CLZ0) {
            desc = ALZ_DESC_MATCH(0x1000u - (b1 | ((b2 >> 4) << 8))); len = (b2 & 0xFu) + 3u; tend = to + 2;   // CLZ0.cs:76-78
        } else if (FMT == ALZ_FMT_BLZ) {
            desc = ALZ_DESC_MATCH((((b1 & 0xFu) << 8) | b2) + 3u); len = (b1 >> 4) + 3u; tend = to + 2;   // BLZ.cs:117-118
        } else if (FMT == ALZ_FMT_LZ11) {
            const u32 b3 = in.lds[ti + 2], b4 = in.lds[ti + 3];
            const u32 nib = b1 >> 4;
            if (nib == 0) { desc = ALZ_DESC_MATCH((((b2 & 0xFu) << 8) | b3) + 1u); len = (((b1 & 0xFu) << 4) | (b2 >> 4)) + 17u; tend = to + 3; }
            else if (nib == 1) { desc = ALZ_DESC_MATCH((((b3 & 0xFu) << 8) | b4) + 1u); len = (((b1 & 0xFu) << 12) | (b2 << 4) | (b3 >> 4)) + 273u; tend = to + 4; }
            else { desc = ALZ_DESC_MATCH((((b1 & 0xFu) << 8) | b2) + 1u); len = nib + 1u; tend = to + 2; }
        } else if (FMT == ALZ_FMT_LZ40) {
            const u32 b3 = in.lds[ti + 2], b4 = in.lds[ti + 3];
            const u32 nib = b1 & 0xFu, d = (b1 >> 4) | (b2 << 4);
            desc = ALZ_DESC_MATCH(d ? d : 4096u);                // E1: 0 is what 4096 wraps to
            if (nib == 0) { len = b3 + 16u; tend = to + 3; } else if (nib == 1) { len = (b3 | (b4 << 8)) + 272u; tend = to + 4; } else { len = nib; tend = to + 2; }
        } else {  // YAZ0
            const u32 b3 = in.lds[ti + 2];
            const u32 nib = b1 >> 4;
            desc = ALZ_DESC_MATCH((((b1 & 0xFu) << 8) | b2) + 1u);
            if (nib == 0) { len = b3 + 0x12u; tend = to + 3; } else { len = nib + 2u; tend = to + 2; }
        }
    }
    // Near the end of the input the window holds bytes past src_len: only tokens that lie completely inside the input are
    // real (token offsets grow with the lane, so they form a prefix).  The first token that does not is left to the exact
    // parser together with the flag-reader state it needs (E6, and Yay0.cs:130-131's length-byte-at-EOF rule).
    const u32 inlim = src_len - p;
    const u64 ingm = lanes_below(8u * ng);                        // lanes of the groups found (a prefix: scalar)
    const u64 fitm = wave_ballot(tend <= inlim);
    u64 vm = ingm & fitm;                                         // the lanes that hold a real token
    bool cut = (ingm & ~fitm) != 0;
    if (FMT == ALZ_FMT_LZ02) {                                    // the terminator and everything behind it: exact parser (same hand-over)
        const u64 tm = vm & wave_ballot(term);
        if (tm) { vm &= lanes_below((u32)__builtin_ctzll(tm)); cut = true; }
    }
    if (cut && vm == 0) { to_serial = true; return false; }
    u32 last_tend;
    const bool fin = fast_emit<OW, EmitCfg<((FMT == ALZ_FMT_LZSS || FMT == ALZ_FMT_BLZ) ? 0u : 4095u), FMT == ALZ_FMT_LZSS, false, OW::FB, false, false, alz_tab512<FMT>::value>>(out, s, size, vm, len, desc, tend, segmark, nullptr, lane, last_tend, gm.W);
    if (fin) {
        s.p = p + last_tend;
        if (FMT == ALZ_FMT_LZ02) {                                // not the end of an LZ02 stream: the exact parser goes on to the terminator
            const u32 lk = (u32)__builtin_ctzll((vm & wave_ballot(tend == last_tend)) | (1ull << 63));
            s.bits = 7u - (lk & 7u);
            s.flag = in.peek1(p + wave_readlane(gstart, lk));
        }
        return true;
    }
    if (!cut) { s.p = p + g; return false; }
    // stopped inside a group: hand (position, remaining flag bits, flag byte) to the serial parser
    const u32 lk = (u32)__popcll(vm) - 1u;
    s.p = p + wave_readlane(tend, lk);
    s.bits = 7u - (lk & 7u);
    s.flag = in.peek1(p + wave_readlane(gstart, lk));
    if (TR::NEG) s.flag = (0u - s.flag) & 0xFFu;
    to_serial = true;
    return false;
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
    const bool fin = fast_emit<OW, EmitCfg<4095u, false, false, false, false, false, true>>(out, s, size, ~0ull, len, desc, tend, segmark, nullptr, lane, last, 4096);
    // (selects, not an if / else of "+=" through the references: the compiler sinks those stores into ONE store through a
    // pointer phi of &up / &fp before inlining, and the cursors then live in scratch memory for the whole kernel)
    cp += fin ? (last >> 8) : 2u * (u32)__popcll(~lm);
    up += fin ? (last & 0xFFu) : (u32)__popcll(um);
    fp += fin ? 0u : 8u;
    return fin;
}

// LZHudson (HudsonSoft/LZHudson.cs:53): Yaz0's tokens behind 32-bit big-endian flag words -- four flag bytes, then the 32
// tokens they describe.  One flag word per iteration (lanes 8r..8r+7 are the tokens of its r-th flag byte): the start of run
// r + 1 is known once run r has been walked, but the walk itself -- "8 tokens with flag byte f start at my byte: how long are
// they?" -- runs on all lanes at once for the (wave-uniform) flag byte of the round, so the chain is four rounds of one
// speculative walk + one v_readlane.  A run that starts beyond byte 63 of the window (only possible behind > 20 three-byte
// matches) waits for the next iteration, which resumes inside the flag word (s.flag / s.bits, the serial parser's state).
// Precondition: s.bits in {0, 8, 16, 24, 32}, s.p + 128 <= src_len, out.produced < size.
template <class OW>
__device__ __forceinline__ bool fast_iter_lzhudson(InCache& in, OW& out, DecState& s, u32 size, u8* segmark, int lane) {
    const u32 p = s.p;
    in.ensure(p, 128);
    const u32 x0 = in.byte_at(p + (u32)lane), x1 = in.byte_at(p + 64u + (u32)lane);
    const u64 zlo = __ballot((x0 >> 4) == 0), zhi = __ballot((x1 >> 4) == 0);
    const u32 l3 = mask_window(zlo, zhi, lane);                  // bit r: the byte r behind mine has a zero high nibble (3-byte match)
    u32 F = uni(s.flag), nb = uni(s.bits), T = 0;                 // T: start of the next run, relative to p
    if (nb == 0) {
        F = (wave_readlane(x0, 0u) << 24) | (wave_readlane(x0, 1u) << 16) | (wave_readlane(x0, 2u) << 8) | wave_readlane(x0, 3u);
        nb = 32; T = 4;
    }
    const u32 nruns = nb >> 3;
    u32 tstart = 0, infosel = 0, done = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if ((u32)r < nruns && T <= 63u && done == (u32)r) {        // (wave-uniform)
            const u32 mbits = ~(F >> (nb - 8u * (u32)(r + 1))) & 0xFFu;   // flag byte of run r, MSB first, 1 = literal; set = match
            u32 rr = 0, info = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 m = __builtin_amdgcn_ubfe(mbits, (u32)(7 - k), 1u);
                u32 extra;                                        // m * z + m (see fast_iter_interleaved)
                { const u32 z = __builtin_amdgcn_ubfe(l3, rr, 1u); asm("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(extra) : "v"(m), "v"(z)); }
                info |= extra << (4 * k);
                rr += 1u + extra;
            }
            const u32 inf = wave_readlane(info, T);
            if (((u32)lane >> 3) == (u32)r) { tstart = T; infosel = inf; }
            T += wave_readlane(rr, T);
            done = (u32)r + 1u;
        }
    }
    const u32 k = (u32)lane & 7u;
    const u32 m = ((infosel >> (4u * k)) & 0xFu) != 0u;
    const u32 to = tstart + k + (u32)__builtin_amdgcn_udot8(infosel & ((1u << (4u * k)) - 1u), 0x11111111u, 0u, false);
    const u32 ti = in.idx(p + to);
    const u32 b1 = in.lds[ti], b2 = in.lds[ti + 1], b3 = in.lds[ti + 2];
    u32 len = 1, desc = ALZ_DESC_LIT(b1), tend = to + 1;
    if (m) {
        const u32 nib = b1 >> 4;
        desc = ALZ_DESC_MATCH((((b1 & 0xFu) << 8) | b2) + 1u);
        if (nib == 0) { len = b3 + 0x12u; tend = to + 3; } else { len = nib + 2u; tend = to + 2; }
    }
    u32 last_tend;
    const bool fin = fast_emit<OW, EmitCfg<4095u, false, false, false>>(out, s, size, lanes_below(8u * done), len, desc, tend, segmark, nullptr, lane, last_tend, 4096);
    s.p = fin ? p + last_tend : p + T;
    s.flag = F; s.bits = nb - 8u * done;
    return fin;
}

// SMSR00 (Nintendo/SMSR00.cs:85-131): 16-bit big-endian masks and the match words of their 16 tokens share one code
// stream, literals have their own.  A group's size depends on its mask alone (1 + number of 0 bits), so the chain of the
// four groups of an iteration is three v_readlane hops over "1 + popcount of the word at my lane"; lane 16 j + k is token k
// of group j, its match word sits (zero bits before k) words behind the mask, its literal at the prefix count of 1 bits.
// Precondition: s.bits == 0, cp + 136 <= codes_len (4 masks + 64 match words), up + 64 <= src_len.
template <class OW>
__device__ __forceinline__ bool fast_iter_smsr00(InCache& cin, InCache& uin, OW& out, DecState& s, u32 size, u8* segmark, int lane, u32& cp, u32& up) {
    cin.ensure(cp, 136); uin.ensure(up, 64);
    const u32 wi = cin.idx(cp + 2u * (u32)lane);
    const u32 w = ((u32)cin.lds[wi] << 8) | cin.lds[wi + 1];          // the code word at my lane, big endian
    const u32 gsize = 1u + (u32)__popc(~w & 0xFFFFu);                 // if a group started here
    const u32 g0 = 0u, g1 = g0 + wave_readlane(gsize, g0), g2 = g1 + wave_readlane(gsize, g1), g3 = g2 + wave_readlane(gsize, g2);
    const u32 gend = g3 + wave_readlane(gsize, g3);
    const u32 j = (u32)lane >> 4, k = (u32)lane & 15u;
    const u32 gstart = j == 0u ? g0 : (j == 1u ? g1 : (j == 2u ? g2 : g3));
    const u32 m0 = wave_readlane(w, g0), m1 = wave_readlane(w, g1), m2 = wave_readlane(w, g2), m3 = wave_readlane(w, g3);
    const u32 mask = j == 0u ? m0 : (j == 1u ? m1 : (j == 2u ? m2 : m3));
    const bool lit = (mask >> (15u - k)) & 1u;                        // MSB first, 1 = literal
    const u32 mbefore = (u32)__popc((~mask & 0xFFFFu) >> (16u - k));  // match tokens of my group before me (k = 0: shift by 16 -> 0)
    const u64 lm = wave_ballot(lit);
    const u32 uidx = mbcnt64(lm);
    u32 len = 1, desc;
    if (lit) desc = ALZ_DESC_LIT(uin.byte_at(up + uidx));
    else {
        const u32 ci = cin.idx(cp + 2u * (gstart + 1u + mbefore));
        const u32 data = ((u32)cin.lds[ci] << 8) | cin.lds[ci + 1];
        desc = ALZ_DESC_MATCH((data & 0x0FFFu) + 1u); len = (data >> 12) + 3u;
    }
    // cursors after this token, packed so one readlane recovers both: code words (<= 68) << 8 | literals (<= 64)
    const u32 tend = ((gstart + 1u + mbefore + (lit ? 0u : 1u)) << 8) | (uidx + (lit ? 1u : 0u));
    u32 last;
    const bool fin = fast_emit<OW, EmitCfg<4095u, false, false, false, false, false, true>>(out, s, size, ~0ull, len, desc, tend, segmark, nullptr, lane, last, 4096);
    if (fin) { cp += 2u * (last >> 8); up += last & 0xFFu; }
    else { cp += 2u * gend; up += (u32)__popcll(lm); }
    return fin;
}


struct EmitRet { u32 produced, flushed, ovf, att_lo, att_hi, mtag; };

// A queued token is ONE 32-bit word per lane: [31:18] length (1..16383), [17] literal flag, [16:0] match distance /
// literal byte / low bits of the literal-run cache offset.
#define ALZ_TOK_MATCH(len, dist) (((len) << 18) | (dist))
#define ALZ_TOK_LIT(len, lo) (((len) << 18) | 0x20000u | (lo))
#define ALZ_TOK_MAXLEN 16383u

// out-of-line execution of one token queue (arguments travel in VGPRs under the device calling convention; the
// wave-uniform ones are re-scalarised on entry)
template <class OW, class CFG>
__device__ __attribute__((noinline)) EmitRet queue_emit_call(u8* dst, u8* win, u32 lw_mask, u32 fl, u32 oshift, u32 cap, u32 produced, u32 flushed,
                                                             int lane, u8* segmark, const u8* inlds, u32 W, u32 nt, u32 qtok, u32 mtag, u32 dirty = 1u) {
    OW out;
    out.dst = reinterpret_cast<u8*>(((u64)uni((u32)((u64)dst >> 32)) << 32) | uni((u32)(u64)dst));
    out.win = win; out.lw_mask = uni(lw_mask); out.fl = uni(fl); out.oshift = uni(oshift); out.cap = uni(cap);
    out.produced = uni(produced); out.flushed = uni(flushed); out.lane = lane;
    out.slack_dirty = uni(dirty) != 0u;                      // (the exact parsers' byte-wise writers may have run in between)
    out.mtag = uni(mtag);                                    // (the byte phase's mark tag lives across calls)
    DecState s; dec_state_init(s);
    const u32 len = qtok >> 18, lo = qtok & 0x1FFFFu;
    u32 desc = lo;
    if (qtok & 0x20000u) desc = CFG::LITRUN ? (0x80000000u | lo) : ALZ_DESC_LIT(lo & 0xFFu);
    u32 last;
    fast_emit<OW, CFG>(out, s, 0xFFFFFFFFu, lanes_below(uni(nt)), len, desc, 0u, segmark, inlds, lane, last, uni(W));
    EmitRet r; r.produced = out.produced; r.flushed = out.flushed; r.ovf = s.ovf ? 1u : 0u;
    r.att_lo = (u32)s.attempted_end; r.att_hi = (u32)(s.attempted_end >> 32); r.mtag = out.mtag;
    return r;
}

// out-of-line, token-by-token execution of a queue on the window (chunked configurations, see QueueSink::flush): small enough
// to live in the caller-saved registers of the calling convention
template <class OW>
__device__ __attribute__((noinline)) EmitRet queue_serial_call(u8* dst, u8* win, u32 lw_mask, u32 fl, u32 oshift, u32 cap, u32 produced, u32 flushed,
                                                               int lane, const u8* inlds, u32 W, u32 nt, u32 qtok) {
    OW out;
    out.dst = reinterpret_cast<u8*>(((u64)uni((u32)((u64)dst >> 32)) << 32) | uni((u32)(u64)dst));
    out.win = win; out.lw_mask = uni(lw_mask); out.fl = uni(fl); out.oshift = uni(oshift); out.cap = uni(cap);
    out.produced = uni(produced); out.flushed = uni(flushed); out.lane = lane; out.slack_dirty = true;
    const u32 n = uni(nt), w = uni(W);
    for (u32 j = 0; j < n; j++) {
        const u32 t = wave_readlane(qtok, j), len = t >> 18, lo = t & 0x1FFFFu;
        if (t & 0x20000u) out.copy_lds(inlds + lo, len); else out.back_copy(lo, len, w);
    }
    EmitRet r; r.produced = out.produced; r.flushed = out.flushed; r.ovf = 0; r.att_lo = 0; r.att_hi = 0; r.mtag = 0;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// QueueSink: the serial parsers of alz_decode_serial.h (PRS, LZ4, LZO, Snappy -- grammars whose token boundaries can
// only be found sequentially) run on the scalar unit and merely RECORD tokens, one per lane (v_writelane); every 64
// tokens -- or when the input cache has to move, or a literal run is too long to stay resident -- the queue is executed
// by the lane-parallel byte phase.  Parsing needs the input only, never the output, so it is decoupled from the copy.
template <class OW, class CFG>
struct QueueSink {
    typedef OW OWT; typedef CFG CFGT;
    OW& out; DecState& s; u8* segmark; const u8* inlds; int lane; u32 W;
    u32 qtok;                 // per-lane token register
    u32 nt, qbytes;           // tokens queued, bytes they will produce (wave-uniform)
    __device__ __forceinline__ QueueSink(OW& o, DecState& st, u8* sm, const u8* il, int ln, u32 w)
        : out(o), s(st), segmark(sm), inlds(il), lane(ln), W(w), qtok(0), nt(0), qbytes(0) {}
    __device__ __forceinline__ u32 produced() const { return out.produced + qbytes; }
    __device__ __forceinline__ void flush() {
        if (nt == 0) return;
        if constexpr (EmitUsesChunks<CFG>::value) {
            // Chunked configurations: the sink only serves the exact parser -- stream tails and the odd token the lane-parallel
            // rounds decline, one or two tokens at a time -- so its queue is executed token by token on the window (E5 was
            // checked when the tokens were queued).  The chunked phase itself is inlined ONCE, in pipelined_rounds: called out
            // of line it needed 98 registers (4 waves per SIMD) where the inlined copy needs 80.
            const EmitRet r = queue_serial_call<OW>(out.dst, out.win, out.lw_mask, out.fl, out.oshift, out.cap, out.produced, out.flushed, lane, inlds, W, nt, qtok);
            out.produced = uni(r.produced); out.flushed = uni(r.flushed); out.slack_dirty = true;
            nt = 0; qbytes = 0;
            return;
        }
        // ONE out-of-line copy of the byte phase per kernel: the sink operations are inlined at every token site of the
        // parsers, and inlining the byte phase there as well made 100+ KB kernels that thrash the instruction cache
        const EmitRet r = queue_emit_call<OW, CFG>(out.dst, out.win, out.lw_mask, out.fl, out.oshift, out.cap, out.produced, out.flushed,
                                                   lane, segmark, inlds, W, nt, qtok, out.mtag);
        out.produced = uni(r.produced); out.flushed = uni(r.flushed); out.mtag = uni(r.mtag);
        if (uni(r.ovf)) { s.ovf = true; s.attempted_end = ((u64)uni(r.att_hi) << 32) | uni(r.att_lo); }
        nt = 0; qbytes = 0;
    }
    __device__ __forceinline__ void ensure(InCache& in, u32 p, u32 need) {
        if (p + in.lo + need > in.cb + 2u * in.ch || p + in.lo < in.cb) { flush(); in.ensure(p, need); }   // queued literal runs point into the cache
    }
    // record one packed token (len <= ALZ_TOK_MAXLEN); the caller keeps nt < 64
    __device__ __forceinline__ void push_word(u32 word, u32 len) {
        qtok = wave_writelane(qtok, uni(word), uni(nt));
        nt = uni(nt + 1u); qbytes = uni(qbytes + len);   // uni(): tells the compiler the queue state is wave-uniform
    }
    __device__ __forceinline__ bool push(u32 word, u32 len) {
        push_word(word, len);
        if (nt == 64u || qbytes >= 0x40000000u) flush();
        return !s.ovf;
    }
    __device__ __forceinline__ bool lit(u32 b) {
        if (produced() >= out.cap) {                                             // E5 at its exact place in the stream (the parser must not run on)
            flush(); if (s.ovf) return false;
            (void)clip_token(out, s, 1u); return false;
        }
        return push(ALZ_TOK_LIT(1u, b & 0xFFu), 1u);
    }
    __device__ __forceinline__ bool match(u32 dist, u64 len, u32 w) {
        if (len == 0) return true;
        if (len > ALZ_TOK_MAXLEN || dist > 0x1FFFFu || (u64)produced() + len > (u64)out.cap) {   // rare: long token / distance beyond the token word (RefPack's 131 072) / exact E5 handling on the serial path
            flush(); if (s.ovf) return false;
            u32 cl = clip_token(out, s, len); out.back_copy(dist, cl, w); return !s.ovf;
        }
        return push(ALZ_TOK_MATCH((u32)len, dist ? dist : w), (u32)len);         // E1
    }
    __device__ __forceinline__ bool run(InCache& in, u32 p, u64 len) {
        if (len == 0) return true;
        if (len > in.ch || (u64)produced() + len > (u64)out.cap) {
            flush(); if (s.ovf) return false;
            u32 cl = clip_token(out, s, len); out.copy_from(in, p, cl); return !s.ovf;
        }
        ensure(in, p, (u32)len);
        return push(ALZ_TOK_LIT((u32)len, in.idx(p)), (u32)len);   // the run is addressed by its input-cache index (< 2048)
    }
};

// Walk of a chain of variable-size elements over a 256-byte window (4 x 64): nx[w] holds, per lane, the size
// of "the element that would start at byte 64 w + lane" (ALZ_NX_BAD: unusual element, stop in front of it).  Lane j of
// `spos` receives the start offset of the j-th element (v_writelane takes one SGPR + M0, so the element counter lives in
// M0); returns the offset of the first element not taken and the count.  The CU's single scalar unit is what bounds the
// parsers, so the loop is cut to four scalar instructions per element: an unusual element has a size that leaves every
// window (found and undone afterwards), and the element limit is checked once per window -- a window is entered while
// fewer than `enter_below` elements have been taken (the caller knows how many one window can add).
// What one walk costs (round 2, every walk executed twice, 10 000 x 256 KiB, kernel ms): LZ4 5.48 -> 6.16, Snappy 7.15 -> 8.74,
// FastLZ 5.90 -> 6.43 -- 12 / 21 / 9 % of the kernel; a two- or four-element hop (nx2 = nx + nx[. + nx] through LDS) would trade
// ~100 scalar instructions per round for ~20 LDS ones and was not built.
#define ALZ_NX_BAD 0x1000u
// `fill3` (round 3): the sizes of the FOURTH window, worked out only when the walk gets there -- with one-token elements of ~5 bytes the
// first three windows already hold more than 32 elements and the fourth is never entered: a quarter of the speculation was for nothing.
// (The first three stay eager: their LDS reads are in flight together.)
struct NoFill3 { __device__ __forceinline__ void operator()(u32&) const {} };
template <class F3 = NoFill3>
__device__ __forceinline__ void lane_walk_pos(u32 (&nx)[4], u32 enter_below, u32& spos_out, u32& sp_out, u32& n_out, F3 fill3 = F3()) {
    // (round 3: inside a window the offset runs 64 (w + 1) below zero, so that the carry of "offset += size" IS "left the window" and
    // the loop needs no compare -- five instructions per element; v_readlane takes the low six bits, which are the offset's.  What
    // the lanes received is put right afterwards from the element counts at the window ends.)
    u32 spos = 0, sp = 0, cnt = 0, c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (cnt < enter_below && sp < 64u * (u32)(w + 1)) {
            if (w == 3) fill3(nx[3]);
            u32 n, t = sp - 64u * (u32)(w + 1);
            asm volatile(
                "s_mov_b32 m0, %[cnt]\n\t"
                "s_nop 1\n"
                "1:\n\t"
                "v_readlane_b32 %[n], %[nx], %[t]\n\t"
                "v_writelane_b32 %[spos], %[t], m0\n\t"
                "s_add_u32 m0, m0, 1\n\t"
                "s_add_u32 %[t], %[t], %[n]\n\t"
                "s_cbranch_scc0 1b\n\t"
                "s_mov_b32 %[cnt], m0\n\t"
                : [n] "=&s"(n), [t] "+s"(t), [spos] "+v"(spos), [cnt] "+s"(cnt)
                : [nx] "v"(nx[w])
                : "scc", "m0");
            sp = t + 64u * (u32)(w + 1);
        }
        if (w == 0) c0 = cnt; else if (w == 1) c1 = cnt; else if (w == 2) c2 = cnt;
    }
    {
        const u32 l = (u32)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        spos += 64u + (l >= c0 ? 64u : 0u) + (l >= c1 ? 64u : 0u) + (l >= c2 ? 64u : 0u);
    }
    if (sp >= ALZ_NX_BAD) { cnt -= 1u; sp = wave_readlane(spos, cnt); }     // the last element counted was an unusual one
    spos_out = spos; sp_out = sp; n_out = cnt;
}

// Lane-parallel LZ4 parse.  Where a sequence starts can only be found by walking the chain of sequences, but what the
// walk needs -- the size of "the sequence that would start at this byte" -- depends on that byte and at most two length
// bytes, so every lane computes it for its own byte of a 256-byte window (4 x 64) and the walk itself is one v_readlane
// per sequence on the scalar unit.  The lanes that turned out to start a sequence build its two tokens (literal run,
// match) and compact them into the queue through a 64-dword LDS staging array; the byte phase executes them.
// Sequences with a second length-extension byte (run >= 270 / match >= 274 bytes) stop the walk and are left to the
// exact parser.  Preconditions: queue empty, one cache chunk + 76 input bytes ahead of s.p, cache covers [p, p + chunk).
// Returns false when nothing was parsed (first sequence unusual, or the batch would not fit the output capacity).
// One round: tokens of up to 32 sequences starting at input offset p -> qt (one per lane, nt of them), the bytes they
// produce and the input bytes they cover.  Touches the input cache and `stage` only.
__device__ __forceinline__ bool lz4_parse_round(InCache& in, u32 p, u32* stage, int lane, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    // 1. speculation: only the SIZE of "the sequence that would start at my byte" (the fields of the real sequences are
    //    decoded once, after the walk, by one lane per sequence)
    // (three loops, not one: written as one, the compiler kept the windows in program order -- token and extension byte of window 0, wait,
    // its match-extension byte, wait, then window 1 ... -- eight dependent LDS round trips per round where two suffice)
    u32 nx[4], bq[4], eq[4], opq[4], emq[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;        // cache index of "my" byte
        bq[w] = in.lds[pos];
        eq[w] = in.lds[pos + 1];
    }
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;
        const u32 L0 = bq[w] >> 4;
        const u32 lx = L0 == 15u ? eq[w] + 1u : 0u;           // literal-length extension byte + its value
        opq[w] = pos + 1u + L0 + lx;                          // offset bytes
        emq[w] = in.lds[(opq[w] + 2u) & 2047u];               // match length extension (if any); masked: garbage lanes may point anywhere
    }
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;
        const u32 b = bq[w], e1 = eq[w], em = emq[w], op = opq[w];
        const u32 L0 = b >> 4, M0 = b & 15u;
        const u32 lx = L0 == 15u ? e1 + 1u : 0u;
        const bool bad = (L0 == 15u && e1 == 255u) || (M0 == 15u && em == 255u) || L0 + lx > ALZ_QRUN;   // (second extension bytes; a run beyond the resident cache chunk)
        nx[w] = bad ? ALZ_NX_BAD : (op + 2u + (M0 == 15u ? 1u : 0u)) - pos;
    }
    // 2. the walk; lane j receives the start offset of the j-th sequence (a window holds <= 22 sequences: lanes suffice)
    u32 spos, sp, nseq;
    lane_walk_pos(nx, 33u, spos, sp, nseq);
    if (nseq == 0u) return false;
    // 3. one lane per sequence: fields, then (literal run?, match) into the queue order (the sequences whose tokens fit its 64 slots)
    bool st = (u32)lane < nseq;
    const u32 pos = i0 + spos;
    const u32 b = in.lds[pos], e1 = in.lds[pos + 1];
    const u32 L0 = b >> 4, M0 = b & 15u;
    const u32 L = L0 + (L0 == 15u ? e1 : 0u);
    const u32 lp = pos + 1u + (L0 == 15u ? 1u : 0u);
    const u32 op = lp + L;
    const u32 d0 = in.lds[op & 2047u], d1 = in.lds[(op + 1u) & 2047u], em = in.lds[(op + 2u) & 2047u];
    const u32 M = M0 + 4u + (M0 == 15u ? em : 0u);
    const u32 dist = d0 | (d1 << 8);
    // (masks of a prefix of lanes come from the scalar unit; a ballot of `prefix && x` would cost two vector instructions more than the ballot of x)
    const u64 hasl = wave_ballot(L != 0u);
    u64 litm = lanes_below(nseq) & hasl;
    const u32 rank = (u32)lane + mbcnt64(litm);
    {
        const u32 keep = (u32)__popcll(lanes_below(nseq) & wave_ballot(rank + (L ? 2u : 1u) <= 64u));
        if (keep < nseq) { nseq = keep; sp = wave_readlane(spos, keep); st = (u32)lane < keep; litm = lanes_below(keep) & hasl; }
    }
    if (st) {
        u32 r = rank;
        if (L) { stage[r] = ALZ_TOK_LIT(L, lp); r++; }
        stage[r] = ALZ_TOK_MATCH(M, dist ? dist : 65536u);          // E1
    }
    const u32 base = nseq + (u32)__popcll(litm);
    wave_sync();
    const u32 qt = (u32)lane < base ? stage[lane] : 0u;
    wave_sync();
    qt_out = qt; nt_out = base; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63);
    return true;
}

// Rounds of a lane-parallel parser executed through the byte phase, software-pipelined: the read-backs of round k (sources
// older than the LDS window, 1-2 us from HBM) are issued by emit_begin, round k + 1 is parsed while they are in flight,
// then emit_finish copies round k.  Parsing needs the input cache only -- which must not move while literal runs of round
// k still point into it, so the pipeline drains whenever the cache has to slide.  `parse(p, qt, nt, total, adv)` yields the
// tokens of one round starting at input offset p (false: nothing parsed) and `commit()` accepts it (parser state).
// A round is only taken while its output stays below `maxout` (capacity / declared size: those rules stay with the exact
// parser).  Preconditions: queue empty, one cache chunk + 76 input bytes ahead of s.p, cache covers [s.p, s.p + chunk).
template <class OW, class CFG, class PARSE>
__device__ __forceinline__ bool pipelined_rounds(InCache& in, OW& out, DecState& s, u32 src_len, u8* segmark, const u8* inlds, int lane,
                                                 u32 W, u32 maxout, PARSE& parse) {
    u32 qt, nt, total, adv;
    if (!parse(s.p, qt, nt, total, adv)) return false;
    if (total > maxout - out.produced) return false;
    for (;;) {
        parse.commit();
        s.p += adv;
        u32 last;
        const u32 len = qt >> 18, lo = qt & 0x1FFFFu;
        const u32 desc = (qt & 0x20000u) ? (0x80000000u | lo) : lo;
        bool more = false;
        u32 qt2 = 0, nt2 = 0, total2 = 0, adv2 = 0;
        const u32 p = s.p;
        const bool ahead = (u64)p + in.ch + 76u <= src_len && !(p + in.lo + in.ch > in.cb + 2u * in.ch);   // next round: input ahead, cache already covers it
        {   // (round 1 parsed round k + 1 between the issue and the use of round k's HBM read-backs; the chunked phase reads a far
            // source as one 20-byte load per chunk right where it needs it, and keeping a round's token state alive across the
            // parse cost more registers -- a wave per SIMD -- than the overlap gained)
            (void)fast_emit<OW, CFG>(out, s, 0xFFFFFFFFu, lanes_below(nt), len, desc, 0u, segmark, inlds, lane, last, W);
            if (ahead && !s.ovf) {
                more = parse(p, qt2, nt2, total2, adv2);
                if (more && total2 > maxout - out.produced) more = false;
            }
        }
        if (!more) break;
        qt = qt2; nt = nt2; total = total2; adv = adv2;
    }
    return true;
}

struct Lz4Rounds {
    InCache& in; u32* stage; int lane;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) { return lz4_parse_round(in, p, stage, lane, qt, nt, total, adv); }
    __device__ __forceinline__ void commit() {}
};

// Lane-parallel Snappy parse (Snappy.cs:205-250): an element's size depends on its tag byte and, for literals of 61+
// bytes, on one or two length bytes -- the same per-byte speculation + scalar walk as LZ4, one token per element.
// Elements the walk does not take (literals above 200 bytes, copies with 4-byte offsets) are left to the exact parser.
__device__ __forceinline__ bool snappy_parse_round(InCache& in, u32 p, int lane, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    // 1. speculation: the size of "the element that would start at my byte"
    auto size_of = [&](u32 b, u32 e1, u32 e2) -> u32 {
        const u32 type = b & 3u, hi = b >> 2;
        u32 n = type + 1u;                                              // copies with a 1- / 2-byte offset: 2 / 3 bytes
        if (type == 0u) {
            const u32 len = hi < 60u ? hi + 1u : (hi == 60u ? e1 + 1u : (e1 | (e2 << 8)) + 1u);
            const u32 hdr = hi < 60u ? 1u : hi - 58u;
            n = (hi > 61u || len > ALZ_QRUN) ? ALZ_NX_BAD : hdr + len;      // the run has to stay inside the resident input cache
        }
        if (type == 3u) n = ALZ_NX_BAD;                                 // 4-byte offsets (E3 check included) stay with the exact parser
        return n;
    };
    auto size_at = [&](int w) -> u32 { const u32 pos = i0 + 64u * (u32)w + (u32)lane; return size_of(in.lds[pos], in.lds[pos + 1], in.lds[pos + 2]); };
    u32 nx[4], sb[3][3];
#pragma unroll
    for (int w = 0; w < 3; w++) { const u32 pos = i0 + 64u * (u32)w + (u32)lane; sb[w][0] = in.lds[pos]; sb[w][1] = in.lds[pos + 1]; sb[w][2] = in.lds[pos + 2]; }   // (all three windows' bytes in flight together)
#pragma unroll
    for (int w = 0; w < 3; w++) nx[w] = size_of(sb[w][0], sb[w][1], sb[w][2]);
    nx[3] = 0;
    // 2. the walk: one token per element, lane j = j-th element (a window holds <= 32 elements)
    u32 spos, sp, nel;
    lane_walk_pos(nx, 33u, spos, sp, nel, [&](u32& n3) { n3 = size_at(3); });
    if (nel == 0u) return false;
    // 3. the elements' tokens
    const u32 pos = i0 + spos;
    const u32 b = in.lds[pos], e1 = in.lds[pos + 1], e2 = in.lds[pos + 2];
    const u32 type = b & 3u, hi = b >> 2;
    u32 t;
    if (type == 0u) {
        const u32 len = hi < 60u ? hi + 1u : (hi == 60u ? e1 + 1u : (e1 | (e2 << 8)) + 1u);
        const u32 hdr = hi < 60u ? 1u : hi - 58u;
        t = ALZ_TOK_LIT(len & 0x3FFu, (pos + hdr) & 2047u);
    } else if (type == 1u) {
        const u32 d = ((b >> 5) << 8) | e1;
        t = ALZ_TOK_MATCH((hi & 7u) + 4u, d ? d : 65536u);
    } else {
        const u32 d = e1 | (e2 << 8);
        t = ALZ_TOK_MATCH(hi + 1u, d ? d : 65536u);
    }
    const u32 qt = (u32)lane < nel ? t : 0u;
    qt_out = qt; nt_out = nel; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63);
    return true;
}
struct SnappyRounds {
    InCache& in; int lane;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) { return snappy_parse_round(in, p, lane, qt, nt, total, adv); }
    __device__ __forceinline__ void commit() {}
};

// Lane-parallel CNX2 parse (CNX2.cs:83-139).  The unit of speculation is the GROUP -- a flag byte and the up to four
// tokens its 2-bit codes announce --: every lane works out how long "the group that would start at my byte" is (four
// dependent steps, because a literal run's length byte decides where the next token starts), the scalar walk hops from
// group to group, and lane 4 g + k then decodes token k of the g-th real group (codes 0 -- skip, which also drops the rest
// of its flag byte -- and empty runs yield no token, so the tokens are compacted through `stage`).  Groups longer than 700
// bytes are left to the exact parser.  Precondition: at a flag-byte boundary (s.bits == 0).  `total` is reported one too
// high so that the round in which the output reaches the declared size is refused: what follows that point in the group
// must not be consumed (the exact parser stops there).
__device__ __forceinline__ void cnx2_step(const InCache& in, u32 base, u32 f, int k, u32& off, bool& live, u32& code, u32& b) {
    code = (f >> (2 * k)) & 3u;
    b = in.lds[(base + off) & 2047u];
    const u32 sz = code == 1u ? 1u : (code == 2u ? 2u : 1u + b);
    if (!live) code = 0xFFu;                                         // behind a skip: not a token
    off += live ? sz : 0u;
    live = live && code != 0u;
}
__device__ __forceinline__ bool cnx2_parse_round(InCache& in, u32 p, u32* stage, int lane, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    u32 nx[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;
        const u32 f = in.lds[pos];
        u32 off = 1, code, b; bool live = true;
#pragma unroll
        for (int k = 0; k < 4; k++) cnx2_step(in, pos, f, k, off, live, code, b);
        nx[w] = off > 700u ? ALZ_NX_BAD : off;                           // (CNX2 keeps 1 KiB cache chunks: a group can be four 256-byte runs)
    }
    u32 spos, sp, ng;
    lane_walk_pos(nx, 16u, spos, sp, ng);                              // a group has >= 2 bytes: <= 32 per window
    if (ng > 16u) { ng = 16u; sp = wave_readlane(spos, 16u); }         // 16 groups x 4 tokens fill the queue
    if (ng == 0u) return false;
    const u32 g = (u32)lane >> 2;
    const u32 gs = i0 + wave_bperm(g, spos);
    const u32 f = in.lds[gs & 2047u];
    u32 off = 1, code = 0, b = 0; bool live = true;
    u32 toff = 1, tcode = 0xFFu, tb = 0;                               // my token: offset of its first byte, code, that byte
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 o = off;
        cnx2_step(in, gs, f, k, off, live, code, b);
        if (((u32)lane & 3u) == (u32)k) { toff = o; tcode = code; tb = b; }
    }
    const u32 tp = gs + toff;
    u32 t = 0;
    if (tcode == 1u) t = ALZ_TOK_LIT(1u, tp & 2047u);
    else if (tcode == 2u) { const u32 pair = (tb << 8) | in.lds[(tp + 1u) & 2047u]; t = ALZ_TOK_MATCH((pair & 0x1Fu) + 4u, (pair >> 5) + 1u); }
    else if (tcode == 3u && tb != 0u) t = ALZ_TOK_LIT(tb, (tp + 1u) & 2047u);
    const bool valid = g < ng && t != 0u;
    const u64 vm = __ballot(valid);
    const u32 nv = (u32)__popcll(vm);
    if (nv == 0u) return false;
    if (valid) stage[mbcnt64(vm)] = t;
    wave_sync();
    const u32 qt = (u32)lane < nv ? stage[lane] : 0u;
    wave_sync();
    qt_out = qt; nt_out = nv; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63) + 1u;
    return true;
}
struct Cnx2Rounds {
    InCache& in; u32* stage; int lane;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) { return cnx2_parse_round(in, p, stage, lane, qt, nt, total, adv); }
    __device__ __forceinline__ void commit() {}
};

// Lane-parallel LZShrek parse (LZShrek.cs:73-119).  What an element is depends on one piece of state -- how many matches the
// current group still owes: none = a group header (flag + its literals), otherwise a match -- so, as for LZO, every lane
// interprets its byte both ways and the scalar walk carries the state: pk = [9:0] size as a header (0x3FF: not taken),
// [13:10] the group's match count, [26:14] size as a match (0x1000: not taken -- the end marker, a distance beyond the
// window).  Lane j of `spos` receives (start offset | state in front of it << 10) of the j-th element.
__device__ __forceinline__ void shrek_walk_pos(const u32 (&pk)[4], u32& spos_out, u32& sp_out, u32& n_out, u32& state_io) {
    u32 spos = 0, sp = 0, cnt = 0, state = uni(state_io), n = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (cnt < 64u && n < 0x3FFu && sp < 64u * (u32)(w + 1)) {
            u32 v, t;
            asm volatile(
                "s_mov_b32 m0, %[cnt]\n\t"
                "s_nop 1\n"
                "1:\n\t"
                "v_readlane_b32 %[v], %[pk], %[sp]\n\t"
                "s_lshl_b32 %[t], %[state], 10\n\t"
                "s_or_b32 %[t], %[t], %[sp]\n\t"
                "s_cmp_eq_u32 %[state], 0\n\t"
                "v_writelane_b32 %[spos], %[t], m0\n\t"
                "s_cbranch_scc0 2f\n\t"
                "s_and_b32 %[n], %[v], 0x3ff\n\t"                  // a group header: its size, and the matches it announces
                "s_bfe_u32 %[state], %[v], 0x4000a\n\t"
                "s_branch 3f\n"
                "2:\n\t"
                "s_bfe_u32 %[n], %[v], 0xd000e\n\t"                 // a match
                "s_sub_u32 %[state], %[state], 1\n"
                "3:\n\t"
                "s_add_u32 m0, m0, 1\n\t"
                "s_add_u32 %[sp], %[sp], %[n]\n\t"
                "s_cmp_ge_u32 m0, 64\n\t"
                "s_cbranch_scc1 4f\n\t"
                "s_cmp_lt_u32 %[sp], %[lim]\n\t"
                "s_cbranch_scc1 1b\n"
                "4:\n\t"
                "s_mov_b32 %[cnt], m0\n\t"
                : [v] "=&s"(v), [n] "+s"(n), [t] "=&s"(t), [sp] "+s"(sp), [spos] "+v"(spos), [cnt] "+s"(cnt), [state] "+s"(state)
                : [pk] "v"(pk[w]), [lim] "s"(64u * (u32)(w + 1))
                : "scc", "m0");
        }
    }
    if (n >= 0x3FFu) { cnt -= 1u; const u32 v = wave_readlane(spos, cnt); sp = v & 0x3FFu; state = v >> 10; }   // undo the element not taken
    spos_out = spos; sp_out = sp; n_out = cnt; state_io = state;
}
__device__ __forceinline__ u32 shrek_count(u32 v, u32 x, u32 y) { return v < 30u ? v : (v == 30u ? 30u + x : 286u + x + (y << 8)); }
__device__ __forceinline__ bool lzshrek_parse_round(InCache& in, u32 p, u32* stage, int lane, u32& state, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    u32 pk[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;
        const u32 b = in.lds[pos], e1 = in.lds[pos + 1], e2 = in.lds[pos + 2], e3 = in.lds[pos + 3];
        const u32 v = b >> 3, ext = v == 30u ? 1u : (v == 31u ? 2u : 0u);
        const u32 lits = shrek_count(v, e1, e2);
        const u32 sizeH = lits > ALZ_QRUN ? 0x3FFu : 1u + ext + lits;
        const u32 lenb = (b & 7u) == 0u ? 1u : 0u;
        const u32 x = lenb ? e2 : e1, y = lenb ? e3 : e2;
        const u32 dist = shrek_count(v, x, y) + 1u;
        const bool badm = (lenb && e1 == 0u) || dist > 0x1000u;
        const u32 sizeM = badm ? 0x1000u : 1u + lenb + ext;
        pk[w] = sizeH | (((b & 7u) + 1u) << 10) | (sizeM << 14);
    }
    u32 spos, sp, nel;
    shrek_walk_pos(pk, spos, sp, nel, state);
    if (nel == 0u) return false;
    const bool st = (u32)lane < nel;
    const u32 pos = i0 + (spos & 0x3FFu);
    const bool header = (spos >> 10) == 0u;
    const u32 b = in.lds[pos & 2047u], e1 = in.lds[(pos + 1u) & 2047u], e2 = in.lds[(pos + 2u) & 2047u], e3 = in.lds[(pos + 3u) & 2047u];
    const u32 v = b >> 3, ext = v == 30u ? 1u : (v == 31u ? 2u : 0u);
    u32 t;
    if (header) { const u32 lits = shrek_count(v, e1, e2); t = lits ? ALZ_TOK_LIT(lits, (pos + 1u + ext) & 2047u) : 0u; }
    else {
        const u32 lenb = (b & 7u) == 0u ? 1u : 0u;
        const u32 len = lenb ? e1 + 7u : (b & 7u);
        t = ALZ_TOK_MATCH(len, shrek_count(v, lenb ? e2 : e1, lenb ? e3 : e2) + 1u);
    }
    const bool valid = st && t != 0u;
    const u64 vm = __ballot(valid);
    const u32 nv = (u32)__popcll(vm);
    if (nv == 0u) return false;                                          // (only empty group headers: the exact parser steps over them)
    if (valid) stage[mbcnt64(vm)] = t;
    wave_sync();
    const u32 qt = (u32)lane < nv ? stage[lane] : 0u;
    wave_sync();
    qt_out = qt; nt_out = nv; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63);
    return true;
}
struct LzshrekRounds {
    InCache& in; u32* stage; int lane; u32 state, pending;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) {
        pending = state;
        return lzshrek_parse_round(in, p, stage, lane, pending, qt, nt, total, adv);
    }
    __device__ __forceinline__ void commit() { state = pending; }
};

// Lane-parallel HIG parse (HIG.cs:141-206): an element is a match (2-6 header bytes in three forms) and the literals its PP field
// announces (none, 1, 2, or a counted block); it yields a match token and / or a literal-run token, in that order.  Counted
// blocks above 200 literals and lengths beyond the token word are left to the exact parser, as is the initial literal block.
// `total` is reported one too high (the loop stops at the declared size: nothing behind that point may be consumed).
__device__ __forceinline__ void hig_element(const InCache& in, u32 pos, u32& hdr, u32& length, u32& distance, u32& rawp, u32& cnt) {
    const u32 b = in.lds[pos & 2047u], e1 = in.lds[(pos + 1u) & 2047u];
    u32 pp, L = b >> 5;
    if (L < 6u) { hdr = 2; length = L + 4u; distance = (b & 0x1Cu) << 6; pp = b & 3u; }
    else {
        u32 b2p;
        if (L == 6u) { length = (b & 0x1Fu) + 4u; distance = 0; b2p = pos + 1u; }
        else {
            length = (b & 0xFu) + 3u; distance = (b & 0x10u) << 10; b2p = pos + 1u;
            if (length == 3u) {
                length = e1 + 18u; b2p = pos + 2u;
                if (length == 18u) { length = ((u32)in.lds[(pos + 2u) & 2047u] << 8) | in.lds[(pos + 3u) & 2047u]; b2p = pos + 4u; }
            }
        }
        const u32 b2 = in.lds[b2p & 2047u];
        distance |= (b2 & 0xFCu) << 6; pp = b2 & 3u;
        hdr = b2p - pos + 2u;
    }
    distance |= in.lds[(pos + hdr - 1u) & 2047u];
    rawp = pos + hdr; cnt = pp == 3u ? 0u : pp;
    if (pp == 0u) {
        const u32 c = in.lds[rawp & 2047u];
        if (c != 0u) { cnt = c + 2u; rawp += 1u; }
        else { cnt = (u32)in.lds[(rawp + 1u) & 2047u] | ((u32)in.lds[(rawp + 2u) & 2047u] << 8); rawp += 3u; }
    }
}
__device__ __forceinline__ bool hig_parse_round(InCache& in, u32 p, u32* stage, int lane, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    u32 nx[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;
        u32 hdr, length, distance, rawp, cnt;
        hig_element(in, pos, hdr, length, distance, rawp, cnt);
        nx[w] = (cnt > ALZ_QRUN || length > ALZ_TOK_MAXLEN) ? ALZ_NX_BAD : (rawp - pos) + cnt;
    }
    u32 spos, sp, nel;
    lane_walk_pos(nx, 33u, spos, sp, nel);                               // an element has >= 2 bytes: <= 32 per window
    if (nel == 0u) return false;
    bool st = (u32)lane < nel; u64 stm = lanes_below(nel);
    u32 hdr, length, distance, rawp, cnt;
    hig_element(in, i0 + spos, hdr, length, distance, rawp, cnt);
    u64 mm = (stm & wave_ballot(length != 0u)), litm = (stm & wave_ballot(cnt != 0u));
    const u32 rank = mbcnt64(mm) + mbcnt64(litm);
    {   // the elements whose tokens fit the 64 slots of the queue
        const u32 keep = (u32)__popcll((stm & wave_ballot(rank + (length ? 1u : 0u) + (cnt ? 1u : 0u) <= 64u)));
        if (keep < nel) { nel = keep; sp = wave_readlane(spos, keep); st = (u32)lane < keep; stm = lanes_below(keep); mm = (stm & wave_ballot(length != 0u)); litm = (stm & wave_ballot(cnt != 0u)); }
    }
    if (st) {
        u32 r = rank;
        if (length) { stage[r] = ALZ_TOK_MATCH(length, distance ? distance : 32768u); r++; }   // E1
        if (cnt) stage[r] = ALZ_TOK_LIT(cnt, rawp & 2047u);
    }
    const u32 base = (u32)__popcll(mm) + (u32)__popcll(litm);
    wave_sync();
    const u32 qt = (u32)lane < base ? stage[lane] : 0u;
    wave_sync();
    if (base == 0u) return false;
    qt_out = qt; nt_out = base; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63) + 1u;
    return true;
}
struct HigRounds {
    InCache& in; u32* stage; int lane;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) { return hig_parse_round(in, p, stage, lane, qt, nt, total, adv); }
    __device__ __forceinline__ void commit() {}
};

// Lane-parallel WFLZ parse (WFLZ.cs:130-159): an element is a 4-byte block + its literals and yields a match token and / or a
// literal-run token -- in that order.  The end block is left to the exact parser.
template <bool BIG>
__device__ __forceinline__ bool wflz_parse_round(InCache& in, u32 p, u32* stage, int lane, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    u32 nx[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;
        const u32 length = in.lds[pos + 2], plain = in.lds[pos + 3];
        nx[w] = ((length | plain) == 0u || plain > ALZ_QRUN) ? ALZ_NX_BAD : 4u + plain;
    }
    u32 spos, sp, nel;
    lane_walk_pos(nx, 33u, spos, sp, nel);                               // a block has >= 4 bytes: <= 16 per window
    if (nel == 0u) return false;
    bool st = (u32)lane < nel; u64 stm = lanes_below(nel);
    const u32 pos = i0 + spos;
    const u32 b0 = in.lds[pos], b1 = in.lds[pos + 1], length = in.lds[pos + 2], plain = in.lds[pos + 3];
    const u32 dist = BIG ? ((b0 << 8) | b1) : (b0 | (b1 << 8));
    u64 mm = (stm & wave_ballot(length != 0u)), litm = (stm & wave_ballot(plain != 0u));
    const u32 rank = mbcnt64(mm) + mbcnt64(litm);
    {   // the blocks whose tokens fit the 64 slots of the queue
        const u32 keep = (u32)__popcll((stm & wave_ballot(rank + (length ? 1u : 0u) + (plain ? 1u : 0u) <= 64u)));
        if (keep < nel) { nel = keep; sp = wave_readlane(spos, keep); st = (u32)lane < keep; stm = lanes_below(keep); mm = (stm & wave_ballot(length != 0u)); litm = (stm & wave_ballot(plain != 0u)); }
    }
    if (st) {
        u32 r = rank;
        if (length) { stage[r] = ALZ_TOK_MATCH(length + 4u, dist ? dist : 65536u); r++; }      // E1
        if (plain) stage[r] = ALZ_TOK_LIT(plain, (pos + 4u) & 2047u);
    }
    const u32 base = (u32)__popcll(mm) + (u32)__popcll(litm);
    wave_sync();
    const u32 qt = (u32)lane < base ? stage[lane] : 0u;
    wave_sync();
    qt_out = qt; nt_out = base; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63);
    return true;
}
template <bool BIG>
struct WflzRounds {
    InCache& in; u32* stage; int lane;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) { return wflz_parse_round<BIG>(in, p, stage, lane, qt, nt, total, adv); }
    __device__ __forceinline__ void commit() {}
};

// Lane-parallel RefPack parse (RefPack.cs:177-245): an element is a prefix byte, 0-3 data bytes and its literals (0-3 in front
// of a match, 4..112 alone).  Like LZ4, an element yields a literal-run token and / or a match token, compacted through
// `stage`.  The end token and a distance of 131 072 (beyond the token word) are left to the exact parser.
__device__ __forceinline__ bool refpack_parse_round(InCache& in, u32 p, u32* stage, int lane, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    u32 nx[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;
        const u32 b = in.lds[pos], e1 = in.lds[pos + 1];
        u32 n;
        if (b < 0x80u) n = 2u + (b & 3u);
        else if (b < 0xC0u) n = 3u + (e1 >> 6);
        else if (b < 0xE0u) n = 4u + (b & 3u);
        else n = b >= 0xFCu ? ALZ_NX_BAD : 5u + (b & 0x1Fu) * 4u;
        nx[w] = n;
    }
    u32 spos, sp, nel;
    lane_walk_pos(nx, 33u, spos, sp, nel);                               // an element has >= 2 bytes: <= 32 per window
    if (nel == 0u) return false;
    const u64 stm = lanes_below(nel);
    const u32 pos = i0 + spos;
    const u32 b = in.lds[pos], d0 = in.lds[pos + 1], d1 = in.lds[(pos + 2u) & 2047u], d2 = in.lds[(pos + 3u) & 2047u];
    u32 plain, length = 0, distance = 1, hdr;
    if (b < 0x80u) { hdr = 2; plain = b & 3u; length = ((b & 0x1Cu) >> 2) + 3u; distance = (((b & 0x60u) << 3) | d0) + 1u; }
    else if (b < 0xC0u) { hdr = 3; plain = d0 >> 6; length = (b & 0x3Fu) + 4u; distance = (((d0 & 0x3Fu) << 8) | d1) + 1u; }
    else if (b < 0xE0u) { hdr = 4; plain = b & 3u; length = (((b & 0x0Cu) << 6) | d2) + 5u; distance = (((((b & 0x10u) << 4) | d0) << 8) | d1) + 1u; }
    else { hdr = 1; plain = (b & 0x1Fu) * 4u + 4u; }
    if ((stm & wave_ballot(distance > 0x1FFFFu))) {                           // cut the round in front of the first such element
        const u32 first = (u32)__builtin_ctzll((stm & wave_ballot(distance > 0x1FFFFu)));
        if (first == 0u) return false;
        nel = first; sp = wave_readlane(spos, first);
    }
    bool st2 = (u32)lane < nel; u64 st2m = lanes_below(nel);
    u64 litm = (st2m & wave_ballot(plain != 0u)), mm = (st2m & wave_ballot(length != 0u));
    const u32 rank = mbcnt64(litm) + mbcnt64(mm);
    {   // the elements whose tokens fit the 64 slots of the queue
        const u32 keep = (u32)__popcll((st2m & wave_ballot(rank + (plain ? 1u : 0u) + (length ? 1u : 0u) <= 64u)));
        if (keep < nel) { nel = keep; sp = wave_readlane(spos, keep); st2 = (u32)lane < keep; st2m = lanes_below(keep); litm = (st2m & wave_ballot(plain != 0u)); mm = (st2m & wave_ballot(length != 0u)); }
    }
    if (st2) {
        u32 r = rank;
        if (plain) { stage[r] = ALZ_TOK_LIT(plain, (pos + hdr) & 2047u); r++; }
        if (length) stage[r] = ALZ_TOK_MATCH(length, distance);
    }
    const u32 base = (u32)__popcll(litm) + (u32)__popcll(mm);
    wave_sync();
    const u32 qt = (u32)lane < base ? stage[lane] : 0u;
    wave_sync();
    if (base == 0u) return false;
    qt_out = qt; nt_out = base; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63);
    return true;
}
struct RefpackRounds {
    InCache& in; u32* stage; int lane;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) { return refpack_parse_round(in, p, stage, lane, qt, nt, total, adv); }
    __device__ __forceinline__ void commit() {}
};

// Lane-parallel CNS parse (CNS.cs:77-108): an element is a control byte + its literals, or a control byte + a distance byte.
// One token per element, lane j = element j; empty literal runs are left to the exact parser.  `total` is reported one too
// high for the same reason as in cnx2_parse_round (nothing behind the declared size may be consumed).
__device__ __forceinline__ bool cns_parse_round(InCache& in, u32 p, int lane, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    u32 nx[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 b = in.lds[i0 + 64u * (u32)w + (u32)lane];
        nx[w] = b >= 0x80u ? 2u : (b == 0u ? ALZ_NX_BAD : b + 1u);
    }
    u32 spos, sp, nel;
    lane_walk_pos(nx, 33u, spos, sp, nel);                               // elements have >= 2 bytes: <= 32 per window
    if (nel == 0u) return false;
    const u32 pos = i0 + spos;
    const u32 b = in.lds[pos], e1 = in.lds[pos + 1];
    const u32 t = b >= 0x80u ? ALZ_TOK_MATCH((b & 0x7Fu) + 3u, e1 + 1u) : ALZ_TOK_LIT(b, (pos + 1u) & 2047u);
    const u32 qt = (u32)lane < nel ? t : 0u;
    qt_out = qt; nt_out = nel; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63) + 1u;
    return true;
}
struct CnsRounds {
    InCache& in; int lane;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) { return cns_parse_round(in, p, lane, qt, nt, total, adv); }
    __device__ __forceinline__ void commit() {}
};

// Lane-parallel FastLZ parse (FastLZ.cs:63-160): an element is a control byte plus 1..32 literals, or a match of 2-3 bytes
// (level 1) / 2-5 bytes (level 2: a second length byte of 255 chains on -> exact parser; offset 0x1FFF is followed by a
// 16-bit big-endian extension).  One token per element, lane j = element j.  Never sees the stream's first byte.
__device__ __forceinline__ bool fastlz_parse_round(InCache& in, u32 p, int lane, u32 level, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    const bool l2 = level == 2u;
    u32 nx[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;
        const u32 b = in.lds[pos], e1 = in.lds[pos + 1], e2 = in.lds[pos + 2];
        u32 n = b + 2u;                                                  // literal run: control byte + (b + 1) literals
        if (b >= 32u) {
            const bool ext = (b >> 5) == 7u;
            const u32 low = ext ? e2 : e1;
            n = ext ? 3u : 2u;
            if (l2 && (b & 31u) == 31u && low == 255u) n += 2u;
            if (l2 && ext && e1 == 255u) n = ALZ_NX_BAD;
        }
        nx[w] = n;
    }
    u32 spos, sp, nel;
    lane_walk_pos(nx, 33u, spos, sp, nel);                               // elements have >= 2 bytes: <= 32 per window
    if (nel == 0u) return false;
    const u32 pos = i0 + spos;
    const u32 b = in.lds[pos], e1 = in.lds[pos + 1], e2 = in.lds[pos + 2];
    u32 t = ALZ_TOK_LIT(b + 1u, (pos + 1u) & 2047u);
    if (b >= 32u) {
        const bool ext = (b >> 5) == 7u;
        const u32 lowp = pos + (ext ? 2u : 1u);
        u32 ofs = ((b & 31u) << 8) | (ext ? e2 : e1);
        if (l2 && ofs == 0x1FFFu) ofs = (((u32)in.lds[(lowp + 1u) & 2047u] << 8) | in.lds[(lowp + 2u) & 2047u]) + 0x1FFFu;
        t = ALZ_TOK_MATCH((b >> 5) + 2u + (ext ? e1 : 0u), ofs + 1u);
    }
    const u32 qt = (u32)lane < nel ? t : 0u;
    qt_out = qt; nt_out = nel; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63);
    return true;
}
struct FastlzRounds {
    InCache& in; int lane; u32 level;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) { return fastlz_parse_round(in, p, lane, level, qt, nt, total, adv); }
    __device__ __forceinline__ void commit() {}
};

// ---------------------------------------------------------------------------------------------------------------
// Lane-parallel LZO1X parse (LZO.cs:49-139).  What an instruction is depends on its first byte and, for the opcodes
// below 16, on ONE piece of state -- whether the previous instruction copied 0, 1-3 or 4+ literals (LZO.cs:55) -- so every
// lane works out "the instruction that would start at my byte" for both cases (state A: no pending literals -> a
// literal run; state B/C: a 2-byte match of length 2 / 3) and the scalar walk carries the state from instruction to
// instruction.  Each instruction yields a match token plus, when its low two bits say so, a run of 1-3 trailing
// literals.  Length extensions of more than one byte, the end marker and anything else unusual stop the walk.

// walk over a 256-byte window: `pk` per lane = two halves of 16 bits, the low one for an instruction entered in state A, the high one for
// states B / C: [8:0] size (511: not an instruction the walk takes), [10:9] the state it leaves behind, [15:11] the SHIFT that selects the
// half of the NEXT instruction (16 if that state is B or C, else 0).  state: 0 = A, 1 = B (1-3 literals pending), 2 = C (a literal run
// came before).  Lane j of `spos` receives (start offset | 256 if entered in B / C) of the j-th instruction; the exact state (B or C) a
// lane takes from what the lane before it leaves behind (lzo_parse_round).  Round 3: nine scalar instructions per instruction where
// there were thirteen -- the walk is two thirds of this kernel's scalar instructions --: what the loop carries is the shift, not the
// state; the record is one shift-and-add; the add's carry is the window-exit test (the offset runs 64 (w + 1) below zero inside window
// w: lane_walk_pos).  Same loop discipline as lane_walk_pos (an instruction has >= 2 bytes: <= 32 per window).
template <class F3>
__device__ __forceinline__ void lzo_walk_pos(u32 (&pk)[4], u32& spos_out, u32& sp_out, u32& n_out, u32 state_in, F3 fill3) {
    u32 spos = 0, sp = 0, cnt = 0, sh = uni(state_in) ? 16u : 0u, n = 0, c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (cnt < 33u && sp < 64u * (u32)(w + 1)) {          // (a window of 64 bytes adds at most 32 instructions: lane j <= 63)
            if (w == 3) fill3(pk[3]);
            u32 v, rec, t = sp - 64u * (u32)(w + 1);
            asm volatile(
                "s_mov_b32 m0, %[cnt]\n\t"
                "s_nop 1\n"
                "1:\n\t"
                "v_readlane_b32 %[v], %[pk], %[t]\n\t"
                "s_lshl4_add_u32 %[rec], %[sh], %[t]\n\t"         // offset | 256 when entered in state B / C
                "s_lshr_b32 %[v], %[v], %[sh]\n\t"
                "v_writelane_b32 %[spos], %[rec], m0\n\t"
                "s_and_b32 %[n], %[v], 0x1ff\n\t"
                "s_bfe_u32 %[sh], %[v], 0x5000b\n\t"            // bits [15:11]: the shift for the instruction behind this one
                "s_add_u32 m0, m0, 1\n\t"
                "s_add_u32 %[t], %[t], %[n]\n\t"
                "s_cbranch_scc0 1b\n\t"
                "s_mov_b32 %[cnt], m0\n\t"
                : [v] "=&s"(v), [n] "+s"(n), [sh] "+s"(sh), [rec] "=&s"(rec), [t] "+s"(t), [spos] "+v"(spos), [cnt] "+s"(cnt)
                : [pk] "v"(pk[w])
                : "scc", "m0");
            sp = t + 64u * (u32)(w + 1);
        }
        if (w == 0) c0 = cnt; else if (w == 1) c1 = cnt; else if (w == 2) c2 = cnt;
    }
    {
        const u32 l = (u32)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        spos += 64u + (l >= c0 ? 64u : 0u) + (l >= c1 ? 64u : 0u) + (l >= c2 ? 64u : 0u);
    }
    if (n == 511u) { cnt -= 1u; sp = wave_readlane(spos, cnt) & 0xFFu; }   // the last instruction counted was one the walk does not take: resume in front of it
    spos_out = spos; sp_out = sp; n_out = cnt;
}

// "the instruction that would start at cache index pos": sizes / next states for the walk (TOK = false) or the tokens of
// the instruction entered in `state` (TOK = true: first token, and the trailing-literal token or 0)
template <bool TOK>
__device__ __forceinline__ u32 lzo_interpret_bytes(u32 pos, u32 f, u32 e1, u32 e2, u32 e3, u32 state, u32& second);
template <bool TOK>
__device__ __forceinline__ u32 lzo_interpret(const InCache& in, u32 pos, u32 state, u32& second) {
    const u32 f = in.lds[pos], e1 = in.lds[pos + 1], e2 = in.lds[pos + 2], e3 = in.lds[pos + 3];
    return lzo_interpret_bytes<TOK>(pos, f, e1, e2, e3, state, second);
}
template <bool TOK>
__device__ __forceinline__ u32 lzo_interpret_bytes(u32 pos, u32 f, u32 e1, u32 e2, u32 e3, u32 state, u32& second) {
    second = 0;
    if (f >= 16u) {                                              // opcodes that do not depend on the state
        u32 len, dist, t, size;
        bool bad = false;
        if (f < 64u) {                                           // M4 (16..31) / M3 (32..63): length [+ 1 extension byte] + 2 bytes
            const u32 lm = f < 32u ? 7u : 31u;
            const bool ext = (f & lm) == 0u;
            len = ext ? (f < 32u ? 9u : 33u) + e1 : 2u + (f & lm);
            bad = ext && e1 == 0u;                               // a second extension byte: exact parser
            const u32 b0 = ext ? e2 : e1, b1 = ext ? e3 : e2;
            if (f < 32u) { dist = (16384u + ((f & 8u) << 11)) | (b1 << 6) | (b0 >> 2); bad = bad || dist == 16384u; }   // end marker
            else dist = ((b1 << 6) | (b0 >> 2)) + 1u;
            t = b0 & 3u; size = (ext ? 4u : 3u) + t;
        } else if (f < 128u) { len = 3u + ((f >> 5) & 1u); dist = (e1 << 3) + ((f >> 2) & 7u) + 1u; t = f & 3u; size = 2u + t; }
        else { len = 5u + ((f >> 5) & 3u); dist = (e1 << 3) + ((f & 0x1cu) >> 2) + 1u; t = f & 3u; size = 2u + t; }
        if (TOK) { if (t) second = ALZ_TOK_LIT(t, (pos + size - t) & 2047u); return ALZ_TOK_MATCH(len, dist); }
        const u32 sz = bad ? 511u : size, nx = t ? 1u : 0u;
        const u32 half = sz | (nx << 9) | (nx << 15);            // (the shift behind it: 16 x (state != A))
        return half | (half << 16);
    }
    const bool ext = f == 0u;                                    // state A: literal run  LZO.cs:75-85
    const u32 len = ext ? 18u + e1 : 3u + f;
    const u32 t = f & 3u;                                        // states B / C: 2-byte match of length 2 / 3  LZO.cs:86-97
    if (TOK) {
        if (state == 0u) return ALZ_TOK_LIT(len, (pos + (ext ? 2u : 1u)) & 2047u);
        if (t) second = ALZ_TOK_LIT(t, (pos + 2u) & 2047u);
        return state == 1u ? ALZ_TOK_MATCH(2u, (e1 << 2) + (f >> 2) + 1u) : ALZ_TOK_MATCH(3u, (e1 << 2) + (f >> 2) + 2049u);
    }
    const u32 sizeA = ((ext && e1 == 0u) || len > ALZ_QRUN) ? 511u : (ext ? 2u : 1u) + len;   // (a run beyond the resident cache chunk: exact parser)
    return (sizeA | (2u << 9) | (16u << 11)) | (((2u + t) | ((t ? 1u : 0u) << 9) | ((t ? 16u : 0u) << 11)) << 16);
}

// Preconditions: queue empty, one cache chunk + 76 input bytes ahead of s.p (an instruction boundary), cache covers [p, p + chunk).
// `state` (0 = A, 1 = B, 2 = C) is the walk's state in front of the round on entry, behind it on return.
__device__ __forceinline__ bool lzo_parse_round(InCache& in, u32 p, u32* stage, int lane, u32& state, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out) {
    const u32 i0 = in.idx(p);
    u32 pk[4], dummy, fb[3][4];
    // (the bytes of the three windows first, then the arithmetic: as one loop the compiler kept the windows in program order, three
    // dependent LDS round trips where one suffices)
#pragma unroll
    for (int w = 0; w < 3; w++) {
        const u32 pos = i0 + 64u * (u32)w + (u32)lane;
        fb[w][0] = in.lds[pos]; fb[w][1] = in.lds[pos + 1]; fb[w][2] = in.lds[pos + 2]; fb[w][3] = in.lds[pos + 3];
    }
#pragma unroll
    for (int w = 0; w < 3; w++) pk[w] = lzo_interpret_bytes<false>(i0 + 64u * (u32)w + (u32)lane, fb[w][0], fb[w][1], fb[w][2], fb[w][3], 0u, dummy);
    pk[3] = 0;                                                   // (worked out when the walk gets there: usually it does not)
    u32 spos, sp, ninstr;
    lzo_walk_pos(pk, spos, sp, ninstr, state, [&](u32& p3) { u32 d2; p3 = lzo_interpret<false>(in, i0 + 192u + (u32)lane, 0u, d2); });
    if (ninstr == 0u) return false;
    // the exact state every instruction is entered in: what the instruction in front of it leaves behind (the first: the round's)
    const u32 mypos = i0 + (spos & 0xFFu), mynz = (spos >> 8) & 1u;
    const u32 mf = in.lds[mypos], me1 = in.lds[mypos + 1], me2 = in.lds[mypos + 2], me3 = in.lds[mypos + 3];
    const u32 mypk = lzo_interpret_bytes<false>(mypos, mf, me1, me2, me3, 0u, dummy);
    const u32 leaves = (mypk >> (mynz ? 25u : 9u)) & 3u;
    const u32 before = wave_bperm(((u32)lane - 1u) & 63u, leaves);
    const u32 mystate = lane == 0 ? uni(state) : before;
    // one lane per instruction, in the state the walk entered it: match / run token + trailing-literal token.  The walk takes as
    // many instructions as its windows hold (up to 64; round 2 stopped at 32 whatever they yield, ~39 tokens); the round keeps those
    // whose tokens fit the 64 slots of the queue and resumes in front of the first one that does not.
    u32 tl;
    const u32 first = lzo_interpret_bytes<true>(mypos, mf, me1, me2, me3, mystate, tl);
    bool st = (u32)lane < ninstr; u64 stm = lanes_below(ninstr);
    bool second = st && tl != 0u;
    u64 sm = __ballot(second);
    const u32 rank = (u32)lane + mbcnt64(sm);
    const u32 keep = (u32)__popcll((stm & wave_ballot(rank + (second ? 2u : 1u) <= 64u)));
    if (keep < ninstr) {
        ninstr = keep;
        sp = wave_readlane(spos, keep) & 0xFFu;
        st = (u32)lane < keep; stm = lanes_below(keep); second = second && st; sm = __ballot(second);
    }
    state = wave_readlane(leaves, ninstr - 1u);                  // (ninstr >= 1: an instruction has at most two tokens)
    if (st) { stage[rank] = first; if (second) stage[rank + 1u] = tl; }
    const u32 base = ninstr + (u32)__popcll(sm);
    wave_sync();
    const u32 qt = (u32)lane < base ? stage[lane] : 0u;
    wave_sync();
    qt_out = qt; nt_out = base; adv_out = sp;
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63);
    return true;
}
struct LzoRounds {
    InCache& in; u32* stage; int lane; u32 state, pending;
    __device__ __forceinline__ bool operator()(u32 p, u32& qt, u32& nt, u32& total, u32& adv) {
        pending = state;
        return lzo_parse_round(in, p, stage, lane, pending, qt, nt, total, adv);
    }
    __device__ __forceinline__ void commit() { state = pending; }
};

// ---------------------------------------------------------------------------------------------------------------
// PRS: control bits and data bytes interleave, and a flag byte is fetched at the moment a bit is needed -- possibly in the
// middle of a token -- so where tokens start depends on everything before them.  What CAN be done for all bytes at once
// is what a token would BE if it started at a given byte: every lane interprets "its" byte as a literal, a short match
// and a long match, and two ballots tell the walk the only data it needs (is the 16-bit word here zero = terminator,
// are its low 3 bits zero = a third byte follows).  The walk itself then touches no memory: it runs on the scalar unit
// over a 64-byte window held in one VGPR (flag bytes come in through v_readlane), consumes control bits from a
// sentinel-terminated shift register and records token positions as bit masks per token type -- ~10 scalar instructions
// for a literal, ~19 for a long match, against the 60-100 the compiler generated for the equivalent C++ loop.

// FlagReader state <-> shift register: remaining bits in consumption order, LSB first, with a sentinel 1 above them
template <bool BIG>
__device__ __forceinline__ u32 prs_to_norm(u32 bits, u32 flag) {
    if (bits == 0) return 1u;
    const u32 r = BIG ? (__builtin_bitreverse32(flag & ((1u << bits) - 1u)) >> (32u - bits)) : ((flag & 0xFFu) >> (8u - bits));
    return r | (1u << bits);
}
template <bool BIG>
__device__ __forceinline__ void prs_from_norm(u32 fl, u32& bits, u32& flag) {
    const u32 nb = 31u - (u32)__builtin_clz(fl);
    const u32 r = fl & ((1u << nb) - 1u);
    bits = nb;
    if (nb == 0) { flag = 0; return; }
    flag = BIG ? (__builtin_bitreverse32(r) >> (32u - nb)) : (r << (8u - nb));
}

// The walk advances by a GROUP -- a flag byte and the data bytes of the tokens whose last control bit lies in it, about five tokens
// (alz_prs_table.h) -- instead of by a token: per group one v_readlane (the flag byte), one s_load_dword (the table entry of (entry
// state, flag byte), from the scalar cache), and per long match in the group (at most four) one s_bitcmp1_b64 + s_addc_u32: is the
// bit of "the low three bits of the word here are zero" set where that match's data starts -- the only data a group's size depends
// on.  ~30 scalar instructions per group against ~65 for its tokens one by one (round 2), and none of them touches a token: what the
// walk leaves behind is, per group, (position of its flag byte | entry state << 11) in lane j of `gw`.
//
// One window: the 64 input bytes at cache index i0 + base (lane = byte); groups that START at window positions <= 49 (a group has at
// most 14 bytes: all of it lies inside the window).  pos / stw: window position and state << 11 in front of the next group.
#ifndef ALZ_PRS_NWIN
#define ALZ_PRS_NWIN 2      /* windows per round: ~26 tokens each on the synthetic mix */
#endif
static __device__ const AlzPrsTable g_alz_prs_table = alz_make_prs_table();

template <bool BIG>
__device__ __forceinline__ void prs_walk_groups(const InCache& in, u32 i0, int lane, u32 base, u32& pos_out, u32& stw_io, u32& gw_io, u32& ng_io) {
    const u32 b0 = in.lds[i0 + base + (u32)lane];
    u32 lowb = b0;                                                 // the byte that holds the low bits of "the word that starts here"  PRS.cs:75-77
    if (BIG) lowb = in.lds[i0 + base + (u32)lane + 1u];
    const u32 xf8 = (BIG ? (__builtin_bitreverse32(b0) >> 24) : b0) << 3;   // flag byte in consumption order, as a table offset (8-byte entries)
    const u64 extm = __ballot((lowb & 7u) == 0u);                  // a long match here has its length in a third byte  PRS.cs:85-90
    const u32* tab = g_alz_prs_table.w;
    u32 pos = 0, stw = stw_io, gw = gw_io, cnt = ng_io;
    u32 off, ent, w, a;
    asm volatile(
        "s_mov_b32 m0, %[cnt]\n\t"
        "s_nop 0\n"
        "Lpg_top_%=:\n\t"
        "s_cmp_gt_u32 %[pos], 49\n\t"
        "s_cbranch_scc1 Lpg_end_%=\n\t"
        "v_readlane_b32 %[off], %[xf8], %[pos]\n\t"
        "s_add_u32 %[w], %[pos], %[base]\n\t"
        "s_or_b32 %[off], %[off], %[stw]\n\t"
        "s_load_dword %[ent], %[tab], %[off]\n\t"
        "s_or_b32 %[w], %[w], %[stw]\n\t"
        "v_writelane_b32 %[gw], %[w], m0\n\t"
        "s_add_u32 m0, m0, 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_and_b32 %[stw], %[ent], 0x3800\n\t"
        "s_bitcmp0_b32 %[ent], 4\n\t"
        "s_cbranch_scc1 Lpg_done_%=\n\t"
        "s_bfe_u32 %[a], %[ent], 0x40010\n\t"                     // long match 0: where its data starts
        "s_add_u32 %[a], %[a], %[pos]\n\t"
        "s_bitcmp1_b64 %[extm], %[a]\n\t"
        "s_addc_u32 %[pos], %[pos], 0\n\t"                        // a third byte: everything behind it moves by one
        "s_bitcmp0_b32 %[ent], 5\n\t"
        "s_cbranch_scc1 Lpg_done_%=\n\t"
        "s_bfe_u32 %[a], %[ent], 0x40014\n\t"
        "s_add_u32 %[a], %[a], %[pos]\n\t"
        "s_bitcmp1_b64 %[extm], %[a]\n\t"
        "s_addc_u32 %[pos], %[pos], 0\n\t"
        "s_bitcmp0_b32 %[ent], 6\n\t"
        "s_cbranch_scc1 Lpg_done_%=\n\t"
        "s_bfe_u32 %[a], %[ent], 0x40018\n\t"
        "s_add_u32 %[a], %[a], %[pos]\n\t"
        "s_bitcmp1_b64 %[extm], %[a]\n\t"
        "s_addc_u32 %[pos], %[pos], 0\n\t"
        "s_bitcmp0_b32 %[ent], 10\n\t"
        "s_cbranch_scc1 Lpg_done_%=\n\t"
        "s_bfe_u32 %[a], %[ent], 0x4001c\n\t"
        "s_add_u32 %[a], %[a], %[pos]\n\t"
        "s_bitcmp1_b64 %[extm], %[a]\n\t"
        "s_addc_u32 %[pos], %[pos], 0\n"
        "Lpg_done_%=:\n\t"
        "s_and_b32 %[a], %[ent], 15\n\t"
        "s_add_u32 %[pos], %[pos], %[a]\n\t"
        "s_branch Lpg_top_%=\n"
        "Lpg_end_%=:\n\t"
        "s_mov_b32 %[cnt], m0\n\t"
        : [pos] "+&s"(pos), [stw] "+&s"(stw), [gw] "+&v"(gw), [cnt] "+&s"(cnt), [off] "=&s"(off), [ent] "=&s"(ent), [w] "=&s"(w), [a] "=&s"(a)
        : [xf8] "v"(xf8), [extm] "s"(extm), [tab] "s"(tab), [base] "s"(base)
        : "scc", "m0", "memory");
    pos_out = pos; stw_io = stw; gw_io = gw; ng_io = cnt;
}

// The parse of one round on its own (no window, no output): the tokens of the groups of ALZ_PRS_NWIN walk windows starting at input
// offset p -> qt (one per lane, nt of them), the bytes they produce, the input they cover, the flag register behind them, "the
// terminator was read".  `fl` is the normalised flag register (prs_to_norm); a round starts and ends between two groups, where the
// register holds nothing but the unfinished token's bits ("", "0", "00", "00h"): any other register is declined (the exact parser
// takes tokens until one ends a flag byte).  Touches the input cache and `stage` only.  Preconditions: one cache chunk + 76 input
// bytes ahead of p, cache covers [p, p + chunk).  Returns false (state untouched) when nothing could be parsed.
//
// Behind the walk lane r becomes token r of the round: the groups' token counts (table) give every group its first rank, a mark per
// group and one mbcnt give every token its group, and the token finds its data the way the walk found the group's size -- the
// third bytes of the long matches in front of it, one LDS byte each.
template <bool BIG>
__device__ __forceinline__ bool prs_parse_round(InCache& in, u32 p, u32 fl_in, u32* stage, int lane, u32& qt_out, u32& nt_out, u32& total_out, u32& adv_out, u32& fl_out, u32& term_out) {
    u32 st;
    if (fl_in == 1u) st = 0u; else if (fl_in == 2u) st = 1u; else if (fl_in == 4u) st = 2u; else if (fl_in == 8u) st = 3u; else if (fl_in == 12u) st = 4u;
    else return false;
    const u32 i0 = in.idx(p);
    u32 stw = st << 11, gw = 0, ng = 0, base = 0;
#pragma unroll
    for (int w = 0; w < ALZ_PRS_NWIN; w++) {
        u32 pos;
        prs_walk_groups<BIG>(in, i0, lane, base, pos, stw, gw, ng);
        base += pos;
    }
    gw = wave_writelane(gw, base | stw, ng);                        // what follows the last group: where the round ends if nothing is cut
    // ---- group lanes: tokens per group, first rank of every group
    const bool isg = (u32)lane < ng;
    const u32 gpos = isg ? (gw & 0xFFu) : 0u, gst = isg ? ((gw >> 11) & 7u) : 0u;
    const u32 fb = in.lds[i0 + gpos];
    const u32 fn = BIG ? (__builtin_bitreverse32(fb) >> 24) : fb;
    const uint2 T = *reinterpret_cast<const uint2*>(g_alz_prs_table.w + 2u * (gst * 256u + fn));
    const u32 ntok = isg ? ((T.x >> 7) & 7u) + 1u : 0u;
    const u32 rin = wave_incl_scan(ntok, lane);
    const u32 ngk = (u32)__popcll(__ballot(isg && rin <= 64u));     // the groups that fit the 64 token lanes (a prefix; >= 1)
    u32 nt = wave_readlane(rin, ngk - 1u);
    stage[lane] = 0u;
    wave_sync();
    if ((u32)lane < ngk) stage[rin - 1u] = 1u;                     // the last token of every group
    wave_sync();
    const u32 g = mbcnt64(__ballot(stage[lane] != 0u));            // groups that end in front of me = my group
    const u32 gw2 = gw | ((rin - ntok) << 16);
    const u32 gv = wave_bperm(g, gw2), tx = wave_bperm(g, T.x), ty = wave_bperm(g, T.y);
    // ---- token lanes
    const u32 tpos = gv & 0xFFu, k = ((u32)lane - (gv >> 16)) & 7u;
    const u32 code = (ty >> (3u * k)) & 7u;
    const u32 nlb = (u32)__popc((ty >> 24) & ((1u << k) - 1u));    // long matches of my group in front of me
    u32 E = 0;                                                     // third bytes in front of me
    const u32 gb = i0 + tpos + (BIG ? 1u : 0u);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (__ballot((u32)j < nlb)) {
            const u32 b = in.lds[gb + ((tx >> (16 + 4 * j)) & 15u) + E];
            E += ((u32)j < nlb && (b & 7u) == 0u) ? 1u : 0u;
        }
    }
    const u32 dend = tpos + 1u + k + nlb + E;                      // my data: k - nlb one-byte tokens and nlb long matches in front of it
    const u32 dp = i0 + dend;
    const u32 x0 = in.lds[dp], x1 = in.lds[dp + 1u], x2 = in.lds[dp + 2u];
    const u32 v = BIG ? ((x0 << 8) | x1) : ((x1 << 8) | x0);       // long match word  PRS.cs:75-77
    u32 tok = ALZ_TOK_LIT(1u, x0);                                 // PRS.cs:66-97
    if (code & 4u) tok = ALZ_TOK_MATCH(2u + (code & 3u), 0x100u - x0);
    else if (code == 1u) tok = ALZ_TOK_MATCH((v & 7u) ? (v & 7u) + 2u : x2 + 1u, 0x2000u - (v >> 3));
    u32 adv, stx, term = 0;
    const u64 tm = __ballot((u32)lane < nt && code == 1u && v == 0u);   // PRS.cs:78-79: the zero word ends the stream
    if (tm) {
        const u32 t = (u32)__builtin_ctzll(tm);
        nt = t; term = 1u; adv = wave_readlane(dend, t) + 2u; stx = 0u;
    } else { const u32 e = wave_readlane(gw, ngk); adv = e & 0xFFu; stx = (e >> 11) & 7u; }
    if (nt == 0u && !term) return false;
    const u32 qt = (u32)lane < nt ? tok : 0u;
    qt_out = qt; nt_out = nt; adv_out = adv; term_out = term;
    fl_out = stx == 0u ? 1u : (stx == 1u ? 2u : (stx == 2u ? 4u : (stx == 3u ? 8u : 12u)));
    total_out = wave_readlane(wave_incl_scan(qt >> 18, lane), 63);
    return true;
}

template <class SK, bool BIG>
__device__ __forceinline__ bool prs_lane_parse(InCache& in, SK& sk, DecState& s, u32* stage, int lane, u32& fl_io) {
    const u32 p = s.p;
    u32 qt, nt, total, pos, fl, term;
    if (!prs_parse_round<BIG>(in, p, fl_io, stage, lane, qt, nt, total, pos, fl, term)) return false;
    if (total > sk.out.cap - sk.out.produced) return false;   // the capacity rule (E5) stays with the exact parser
    sk.qtok = qt; sk.nt = nt; sk.qbytes = total;
    s.p = p + pos; fl_io = fl;
    if (term) s.done = true;                                   // PRS.cs:78-79: the zero word ends the stream
    {   // executed in line (the out-of-line queue_emit_call stays with the exact parser's token sites): a round is only ~30
        // tokens, the call's argument traffic was a tenth of it
        sk.nt = 0; sk.qbytes = 0;
        const u32 len = qt >> 18, lo = qt & 0x1FFFFu;
        const u32 desc = (qt & 0x20000u) ? ALZ_DESC_LIT(lo & 0xFFu) : lo;
        u32 last;
        (void)fast_emit<typename SK::OWT, typename SK::CFGT>(sk.out, s, 0xFFFFFFFFu, lanes_below(nt), len, desc, 0u, sk.segmark, sk.inlds, lane, last, sk.W);
    }
    return true;
}
