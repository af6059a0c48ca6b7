// alz_emit_byte.h -- the byte-per-lane back end (round 1): 64 output bytes per step, one per lane.  It stays the back end of
// every configuration whose whole window lives in LDS: on token-dense streams (the synthetic mix: ~7.5 output bytes per
// token, every sixth match reaching into the 500 bytes in front of it) the chunked phase of alz_emit_chunk.h needs 3-4
// re-read passes per step and loses (Yaz0 4.7 ms against 3.3 ms per 10 000 x 256 KiB; docs/EXPERIMENTS.md 4.4), while this phase
// resolves sources inside its 64-byte step in registers (pointer jumping over ds_bpermute).
#pragma once
#include "alz_emit_chunk.h"

// One step of the byte phase: 64 consecutive output bytes, one per lane.  The kernel is balanced between vector issue (2.2 cycles per wave
// instruction and SIMD for add / and / or / shift-right, 4.2 for the rest), the CU's one scalar pipe and its one LDS pipeline
// (profiles/r05_issue_ceiling.md), so the step is written for the fewest instructions of all three kinds:
//   * token lanes mark the lane where their output ENDS inside the step; the token of byte L is then
//     (#tokens ended before the step) + (#marks below L): one mbcnt pair, fused with the x4 of the bpermute address;
//   * a match descriptor IS its distance (literal descriptors have bit 31 set), so the source slot is
//     (slot - descriptor) & mask and "source inside this very step" is the unsigned test descriptor <= lane;
//   * pointer jumping only runs in steps where that test fires for some lane;
//   * EARLY (first W bytes of a stream: sources may lie before the stream start, E2) and !FULL (last, partial step)
//     are separate instantiations so the steady state does not pay for them.
// Pointer jumping for the lanes whose source byte is produced inside the step itself (at most 6 rounds, one ds_bpermute per round).
// State word: an unresolved lane holds 0x40000000 | (source lane << 2) -- which IS its ds_bpermute address: the instruction takes
// (address / 4) mod 64 --, a resolved lane holds its value (a window byte, or a literal descriptor: bit 31 set), so "unresolved" is
// the signed test word >= 0x40000000.  An unresolved lane simply takes over its source's word: that is either the value (resolved) or
// the source's source (jump).  The predicate lives in a scalar register pair across the rounds: the compare that ends a round is the
// select mask of the next -- two vector instructions per round (select, compare).
__device__ __forceinline__ u32 pointer_jump(bool instep, u32 srclane, u32 val) {
    u32 st = instep ? ((srclane << 2) | 0x40000000u) : val;
    bool un = instep;
    do {
        const u32 f = (u32)__builtin_amdgcn_ds_bpermute((int)st, (int)st);
        if (un) st = f;
        un = (int)st >= 0x40000000;
    } while (__ballot(un));
    return st;
}

// ---- byte -> token.  A token lane marks the output byte where its token ENDS; a byte's token is then (#tokens ended before the step) +
// (#marks below the byte): one ballot + one mbcnt pair.  Two ways to keep the marks (CFG::STEPMARKS):
//   * per step (rounds 1-4): every step writes its marks (one ds_write_b8 per lane, most of them into a dump slot), reads them and clears
//     them again -- three of the step's six LDS instructions, no scalar ones;
//   * per block under a tag (round 5): the 128 mark bytes cover a BLOCK of two steps; the mapper writes the marks of a block once (only the
//     lanes whose token ends inside it), a step reads its 64 bytes and compares them with the block's tag, nothing is ever cleared -- a stale
//     mark has an older tag.  Tags run 1..255; when they wrap the array is zeroed.  Steps are counted from the start of the batch (m): even
//     steps open a block.  1.5 LDS instructions per step fewer, ~2.5 scalar ones more (tag, exec mask around the store).
// The LDS pipeline is what a step costs most on (tools/variants/r05_sensitivity.patch: one more ds_write_b8 per step +4.5 %, one more
// ds_bpermute +8.3 %, two more vector instructions +0.7-2.7 %, four more scalar ones +2.5 %; profiles/r05_issue_ceiling.md): tags win on
// the flag-byte family (Yaz0 2.95 -> 2.90 ms per 10 000 x 256 KiB, LZ10 4.14 -> 4.02, 985 -> 1 017 GiB/s with two batches in flight; the
// Test.bmp windows 3.62 -> 3.49 / 4.29 -> 4.02) and lose on PRS (5.45 -> 5.74), whose kernel is bound by the CU's one scalar unit: PRS keeps
// the per-step marks.  (Mapping a whole block per trip -- both mark halves read, both descriptor fetches in flight together, so that the
// mapping chain is paid once per 128 bytes -- was measured on top of the tags: 2.96 against 2.90 ms, PRS 5.80.  Not kept.)
struct Mapper {
    u32 relm;        // my token's LAST byte relative to the next block to be marked (huge: no token / already behind)
    u32 tbase4;      // 4 x (tokens that ended before the next step to be mapped)
    u32 m;           // steps mapped so far in this batch (wave-uniform)
};
// The descriptor of token idx4 / 4.  DESCTAB: the descriptors of a batch's tokens live in a 64-entry LDS table behind the marks for the length of
// the batch -- one ds_read_b32 (2.7 LDS pipeline cycles against the 6.2 of the ds_bpermute_b32 that fetches it from the token lane's register;
// one ds_write_b32 per batch fills the table; Yay0 2.67 -> 2.60 ms per 10 000 x 256 KiB).
template <bool DESCTAB>
__device__ __forceinline__ u32 desc_of(const u8* segmark, u32 idx4, u32 desc) {
    if (DESCTAB) return *reinterpret_cast<const u32*>(segmark + 128u + idx4);
    return (u32)__builtin_amdgcn_ds_bpermute((int)idx4, (int)desc);
}
template <class OW>
__device__ __forceinline__ void mark_block(OW& out, u8* segmark, int lane, Mapper& mp) {
    u32 tag = out.mtag + 1u;
    if (tag > 255u) {                                         // (once per 32 KiB of output)
        segmark[lane] = 0; segmark[64 + lane] = 0; tag = 1u;
        wave_sync();
    }
    out.mtag = tag;
    if (mp.relm < 128u) segmark[mp.relm] = (u8)tag;
    mp.relm -= 128u;
}
// the marks of the next step as a lane mask.  PAR: 0 = the step opens a block, 1 = second step of its block, -1 = decided at run time (steps
// outside the pipelined loop)
template <class OW, int PAR>
__device__ __forceinline__ u64 step_marks(OW& out, u8* segmark, int lane, Mapper& mp) {
    u32 half;
    if (PAR == 0 || (PAR < 0 && (mp.m & 1u) == 0u)) { mark_block(out, segmark, lane, mp); wave_sync(); }
    if (PAR >= 0) half = 64u * (u32)PAR; else half = 64u * (mp.m & 1u);
    const u32 mk = segmark[half + (u32)lane];
    return __ballot(mk == out.mtag);
}
template <class OW, bool DESCTAB, int PAR>
__device__ __forceinline__ u32 map_one(OW& out, u8* segmark, int lane, u32 desc, Mapper& mp) {
    const u64 M = step_marks<OW, PAR>(out, segmark, lane, mp);
    const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(M >> 32), __builtin_amdgcn_mbcnt_lo((u32)M, 0u));
    const u32 dsc = desc_of<DESCTAB>(segmark, (cnt << 2) + mp.tbase4, desc);   // match: the distance; literal: bit31 | ...
    mp.tbase4 += 4u * (u32)__popcll(M);
    mp.m += 1u;
    return dsc;
}

// The copy of one step: `dsc` = the descriptors of its 64 bytes.  EARLY (first W bytes of a stream: sources may lie before the stream start,
// E2) and !FULL (last, partial step) are separate instantiations so the steady state does not pay for them.
template <class OW, class CFG, bool EARLY, bool FULL>
__device__ __forceinline__ void copy_one(OW& out, const u8* inlds, int lane, u32 dsc, u32& qs, u32 nseg) {
    const u32 omask = CFG::OMASK ? CFG::OMASK : out.lw_mask;
    u8* const win = out.win;
    u32 wv = win[(qs - dsc) & omask];                        // source byte (garbage for literals, never used)
    if (EARLY) { if (dsc > qs - out.oshift) wv = 0; }        // E2: before the stream start
    u32 val;                                                 // only the low byte is ever stored
    if (CFG::LITRUN) { const u32 lv = inlds[(qs + dsc) & 2047u]; val = ((int)dsc < 0) ? lv : wv; }
    else val = ((int)dsc < 0) ? dsc : wv;
    const bool instep = FULL ? (dsc <= (u32)lane) : (dsc <= (u32)lane && (u32)lane < nseg);   // source produced inside this very step
    if (__ballot(instep)) val = pointer_jump(instep, (u32)lane - dsc, val);
    if (FULL) win[qs & omask] = (u8)val;
    else if ((u32)lane < nseg) win[qs & omask] = (u8)val;
    wave_sync();
    qs += 64u;
}

// (Round 4 also measured the mark write as `if (my token's step == the step being mapped) mark` with a scalar step counter -- one compare
// instead of subtract + clamp, but a saved / restored exec mask around the store: 12 instead of 13 vector instructions per step and 3.03
// against 2.97 ms per launch, PRS 5.63 against 5.50.  Not kept.)
// (Round 4 measured the pipeline one stage deeper -- copy step k, fetch the descriptors of step k + 1 through addresses found a trip earlier,
// map step k + 2, so that a trip waits for one LDS round trip instead of two dependent ones: Yaz0 3.02 against 2.97 ms per 10 000 x 256 KiB,
// 959 against 979 GiB/s with two batches in flight; the 64 KiB streams of cfg2 gained 1.5 %.  Not kept.)
// Software-pipelined steady-state step: copies step k with the descriptors found one step earlier and maps step k+1.
// The two halves are independent, so their LDS round trips overlap: the dependent chain per step shrinks from
// (mark read -> bpermute -> window read -> window write) to max(mark read -> bpermute, window read -> window write).
// `dsc` holds the descriptors of the current step on entry, `dsc_next` those of the next step on return; the mapper is one step ahead of qs.
template <class OW, class CFG, int PAR>   // PAR: 0 / 1 (the mapper's position in its block is known at compile time in the steady state)
__device__ __forceinline__ void fused_step(OW& out, u8* segmark, const u8* inlds, int lane, u32 desc, Mapper& mp, u32& qs, const u32 dsc, u32& dsc_next) {
    const u32 omask = CFG::OMASK ? CFG::OMASK : out.lw_mask;
    u8* const win = out.win;
    if (PAR == 0) { mark_block(out, segmark, lane, mp); wave_sync(); }
    const u32 wv = win[(qs - dsc) & omask];                  // copy side: source byte of step k (issued BEFORE the mark read: the order matters, 2.90 against 2.93 ms)
    u32 lv = 0;
    if (CFG::LITRUN) lv = inlds[(qs + dsc) & 2047u];
    const u32 mk = segmark[64u * (u32)PAR + (u32)lane];      // map side: marks of step k+1
    const u64 M = __ballot(mk == out.mtag);
    const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(M >> 32), __builtin_amdgcn_mbcnt_lo((u32)M, 0u));
    const u32 dscn = desc_of<CFG::DESCTAB>(segmark, (cnt << 2) + mp.tbase4, desc);
    mp.tbase4 += 4u * (u32)__popcll(M);
    mp.m += 1u;
    u32 val;
    if (CFG::LITRUN) val = ((int)dsc < 0) ? lv : wv;
    else val = ((int)dsc < 0) ? dsc : wv;
    const bool instep = dsc <= (u32)lane;
    if (__ballot(instep)) val = pointer_jump(instep, (u32)lane - dsc, val);
    win[qs & omask] = (u8)val;
    wave_sync();
    qs += 64u; dsc_next = dscn;
}

// Execution of one batch of tokens (after the shared emit_prologue of alz_emit_chunk.h), 64 output bytes per step.
// `segmark`: 128 bytes of LDS (zeroed by the kernel; only this phase ever writes them), `inlds`: the LDS input cache literal runs point into.
// (Round 3 tried the byte -> token mapping without the mark array -- every token sets one bit of a bitmap of the batch's output once per
// 1 024 bytes, a step's 64 bits are two v_readlane away from a register copy of it, no LDS traffic per step: 7 % SLOWER on every byte-phase
// format (Yaz0 3.50 against 3.27 ms, LZ10 4.49 / 4.23, PRS 5.92 / 5.57).  The mark chain runs beside the copy chain and is not what a step waits for.)
template <class OW, class CFG>
__device__ __forceinline__ void byte_emit_steps(OW& out, u8* segmark, const u8* inlds, int lane, const EmitState& e) {
    u32 desc = e.desc;
    if (CFG::LITRUN) {                                        // literal run: (input-cache index of the run) - (slot coordinate of its first byte)
        if (desc >> 31) desc = 0x80000000u | ((desc - (e.O + e.off + out.oshift)) & 2047u);
    }
    const u32 O = e.O, T = e.T, W = e.W;
    u32 X = 0;
    u32 qs = O + (u32)lane + out.oshift;                     // slot coordinate of this lane's byte in the current step
    Mapper mp;
    mp.tbase4 = 0; mp.m = 0;
    mp.relm = e.kept ? e.off + e.clen - 1u : 0xFFFFFF00u;
    if (CFG::DESCTAB) reinterpret_cast<u32*>(segmark + 128u)[lane] = desc;    // (ordered before the first read by the wave_sync behind the first block's marks)
    // unpipelined steps: the first W bytes of a stream (E2 test), and one more if that leaves the mapper in the middle of a block
    while (X + 64u <= T && (O + X < W || (mp.m & 1u))) {
        const u32 d0 = map_one<OW, CFG::DESCTAB, -1>(out, segmark, lane, desc, mp);
        copy_one<OW, CFG, true, true>(out, inlds, lane, d0, qs, 64u);
        X += 64u; out.produced = O + X; if (out.produced - out.flushed >= out.fl) out.flush_blocks();
    }
    u32 dA = 0, dB = 0; u32 ahead = 0;                        // steps mapped and not yet copied (their descriptors: dA, then dB)
    if (X + 128u <= T) {                                      // the steady state: two steps per trip (descriptor registers ping-pong, no copy)
        dA = map_one<OW, CFG::DESCTAB, 0>(out, segmark, lane, desc, mp);
        ahead = 1;
        do {
            fused_step<OW, CFG, 1>(out, segmark, inlds, lane, desc, mp, qs, dA, dB);
            fused_step<OW, CFG, 0>(out, segmark, inlds, lane, desc, mp, qs, dB, dA);
            X += 128u; out.produced = O + X;
            if (out.produced - out.flushed >= out.fl) out.flush_blocks();
        } while (X + 128u <= T);
    }
    if (X + 64u <= T) {                                       // one more whole step (fusing it with the mapping of a partial step behind it: Yaz0 2.94 against 2.90 ms)
        if (!ahead) dA = map_one<OW, CFG::DESCTAB, -1>(out, segmark, lane, desc, mp);
        copy_one<OW, CFG, true, true>(out, inlds, lane, dA, qs, 64u); ahead = 0;
        X += 64u; out.produced = O + X; if (out.produced - out.flushed >= out.fl) out.flush_blocks();
    }
    if (X < T) {
        if (!ahead) dA = map_one<OW, CFG::DESCTAB, -1>(out, segmark, lane, desc, mp);
        copy_one<OW, CFG, true, false>(out, inlds, lane, dA, qs, T - X);
        out.produced = O + T; if (out.produced - out.flushed >= out.fl) out.flush_blocks();
    }
}

// ---- the per-step marks of rounds 1-4, verbatim (CFG::STEPMARKS: the PRS kernels).  Every step writes its marks (one ds_write_b8 per lane, most
// of them into a dump slot), reads them and clears them again; no scalar instructions.
template <class OW, class CFG, bool EARLY, bool FULL>
__device__ __forceinline__ void byte_step_sm(OW& out, u8* segmark, const u8* inlds, int lane, u32 desc, u32& relm, u32& qs, u32& tbase4, u32 nseg) {
    const u32 omask = CFG::OMASK ? CFG::OMASK : out.lw_mask;
    u8* const win = out.win;
    { const u32 dump = 64u + (u32)lane; segmark[relm < dump ? relm : dump] = 1; }   // slots 64..127 are never read
    wave_sync();
    const u32 mk = segmark[lane];
    segmark[lane] = 0;
    const u64 M = __ballot(mk != 0);
    const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(M >> 32), __builtin_amdgcn_mbcnt_lo((u32)M, 0u));
    const u32 dsc = (u32)__builtin_amdgcn_ds_bpermute((int)((cnt << 2) + tbase4), (int)desc);   // match: the distance; literal: bit31 | ...
    tbase4 += 4u * (u32)__popcll(M);
    u32 wv = win[(qs - dsc) & omask];                        // source byte (garbage for literals, never used)
    if (EARLY) { if (dsc > qs - out.oshift) wv = 0; }        // E2: before the stream start
    u32 val;                                                 // only the low byte is ever stored
    if (CFG::LITRUN) { const u32 lv = inlds[(qs + dsc) & 2047u]; val = ((int)dsc < 0) ? lv : wv; }
    else val = ((int)dsc < 0) ? dsc : wv;
    const bool instep = FULL ? (dsc <= (u32)lane) : (dsc <= (u32)lane && (u32)lane < nseg);   // source produced inside this very step
    if (__ballot(instep)) val = pointer_jump(instep, (u32)lane - dsc, val);
    if (FULL) win[qs & omask] = (u8)val;
    else if ((u32)lane < nseg) win[qs & omask] = (u8)val;
    wave_sync();
    qs += 64u; relm -= 64u;
}

// Two-pass form of the step for configurations with HBM read-back (see fast_emit): map_step_sm finds the descriptor of
// this lane's byte, copy_step_sm moves the byte.
__device__ __forceinline__ u32 map_step_sm(u8* segmark, int lane, u32 desc, u32& relm, u32& tbase4) {
    { const u32 dump = 64u + (u32)lane; segmark[relm < dump ? relm : dump] = 1; }
    wave_sync();
    const u32 mk = segmark[lane];
    segmark[lane] = 0;
    const u64 M = __ballot(mk != 0);
    const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(M >> 32), __builtin_amdgcn_mbcnt_lo((u32)M, 0u));
    const u32 dsc = (u32)__builtin_amdgcn_ds_bpermute((int)((cnt << 2) + tbase4), (int)desc);
    tbase4 += 4u * (u32)__popcll(M);
    relm -= 64u;
    return dsc;
}

template <class OW, class CFG>
__device__ __forceinline__ void copy_step_sm(OW& out, const u8* inlds, int lane, u32 dsc, u32 far, u32& qs, u32 nseg, bool early) {
    const u32 omask = CFG::OMASK ? CFG::OMASK : out.lw_mask;
    u8* const win = out.win;
    u32 wv = win[(qs - dsc) & omask];
        if (early) { if (dsc > qs - out.oshift) wv = 0; }        // E2: before the stream start
    u32 val;
    if (CFG::LITRUN) { const u32 lv = inlds[(qs + dsc) & 2047u]; val = ((int)dsc < 0) ? lv : wv; }
    else val = ((int)dsc < 0) ? dsc : wv;
    const bool instep = dsc <= (u32)lane && (u32)lane < nseg;
    if (__ballot(instep)) val = pointer_jump(instep, (u32)lane - dsc, val);
    if ((u32)lane < nseg) win[qs & omask] = (u8)val;
    wave_sync();
    qs += 64u;
}

// (Round 4 also measured the mark write as `if (my token's step == the step being mapped) mark` with a scalar step counter -- one compare
// instead of subtract + clamp, but a saved / restored exec mask around the store: 12 instead of 13 vector instructions per step and 3.03
// against 2.97 ms per launch, PRS 5.63 against 5.50.  Not kept.)
// (Round 4 measured the pipeline one stage deeper -- copy step k, fetch the descriptors of step k + 1 through addresses found a trip earlier,
// map step k + 2, so that a trip waits for one LDS round trip instead of two dependent ones: Yaz0 3.02 against 2.97 ms per 10 000 x 256 KiB,
// 959 against 979 GiB/s with two batches in flight; the 64 KiB streams of cfg2 gained 1.5 %.  Not kept.)
// Software-pipelined steady-state step: copies step k with the descriptors found one step earlier and maps step k+1.
// The two halves are independent, so their LDS round trips overlap: the dependent chain per step shrinks from
// (mark read -> bpermute -> window read -> window write) to max(mark read -> bpermute, window read -> window write).
// `dsc` holds the descriptors of the current step on entry and of the next step on return; relm / tbase4 belong to the
// mapping side (one step ahead of qs).
template <class OW, class CFG>
__device__ __forceinline__ void fused_step_sm(OW& out, u8* segmark, const u8* inlds, int lane, u32 desc, u32& relm, u32& qs, u32& tbase4, const u32 dsc, u32& dsc_next) {
    const u32 omask = CFG::OMASK ? CFG::OMASK : out.lw_mask;
    u8* const win = out.win;
    { const u32 dump = 64u + (u32)lane; segmark[relm < dump ? relm : dump] = 1; }
    wave_sync();
    const u32 wv = win[(qs - dsc) & omask];                  // copy side: source byte of step k
    u32 lv = 0;
    if (CFG::LITRUN) lv = inlds[(qs + dsc) & 2047u];
    const u32 mk = segmark[lane];                            // map side: marks of step k+1
    segmark[lane] = 0;
    const u64 M = __ballot(mk != 0);
    const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(M >> 32), __builtin_amdgcn_mbcnt_lo((u32)M, 0u));
    const u32 dscn = (u32)__builtin_amdgcn_ds_bpermute((int)((cnt << 2) + tbase4), (int)desc);
    tbase4 += 4u * (u32)__popcll(M);
    u32 val;
    if (CFG::LITRUN) val = ((int)dsc < 0) ? lv : wv;
    else val = ((int)dsc < 0) ? dsc : wv;
    const bool instep = dsc <= (u32)lane;
    if (__ballot(instep)) val = pointer_jump(instep, (u32)lane - dsc, val);
    win[qs & omask] = (u8)val;
    wave_sync();
    qs += 64u; relm -= 64u; dsc_next = dscn;
}

// Execution of one batch of tokens (after the shared emit_prologue of alz_emit_chunk.h), 64 output bytes per step.
// `segmark`: 128 bytes of LDS (zeroed by the kernel), `inlds`: the LDS input cache literal runs point into.
// (Round 3 tried the byte -> token mapping without the mark array -- every token sets one bit of a bitmap of the batch's output once per
// 1 024 bytes, a step's 64 bits are two v_readlane away from a register copy of it, no LDS traffic per step: 7 % SLOWER on every byte-phase
// format (Yaz0 3.50 against 3.27 ms, LZ10 4.49 / 4.23, PRS 5.92 / 5.57).  The mark chain runs beside the copy chain and is not what a step waits for.)
template <class OW, class CFG>
__device__ __forceinline__ void byte_emit_steps_sm(OW& out, u8* segmark, const u8* inlds, int lane, const EmitState& e) {
    u32 desc = e.desc;
    if (CFG::LITRUN) {                                        // literal run: (input-cache index of the run) - (slot coordinate of its first byte)
        if (desc >> 31) desc = 0x80000000u | ((desc - (e.O + e.off + out.oshift)) & 2047u);
    }
    const u32 O = e.O, T = e.T, W = e.W;
    u32 X = 0;
    u32 qs = O + (u32)lane + out.oshift;                     // slot coordinate of this lane's byte in the current step
    u32 tbase4 = 0;                                          // tbase4: 4 x (tokens that ended before the current step)
    u32 relm = e.kept ? e.off + e.clen - 1u : 0xFFFFFF00u;   // my token's LAST byte relative to the current step (huge: none)
    while (X + 64u <= T && O + X < W) { byte_step_sm<OW, CFG, true, true>(out, segmark, inlds, lane, desc, relm, qs, tbase4, 64u); X += 64u; out.produced = O + X; if (out.produced - out.flushed >= out.fl) out.flush_blocks(); }
    u32 nleft = (T - X) >> 6;
    u32 dsc = 0; bool have = false;
    if (nleft) {
        dsc = map_step_sm(segmark, lane, desc, relm, tbase4); have = true;
        do {
            const u32 pos = O + X + out.oshift;
            u32 nb = (out.fl - (pos & (out.fl - 1u)) + 63u) >> 6;
            if (nb > nleft) nb = nleft;
            u32 k = nb, dsc2;
            for (; k >= 2u; k -= 2u) {
                fused_step_sm<OW, CFG>(out, segmark, inlds, lane, desc, relm, qs, tbase4, dsc, dsc2);
                fused_step_sm<OW, CFG>(out, segmark, inlds, lane, desc, relm, qs, tbase4, dsc2, dsc);
            }
            if (k) { fused_step_sm<OW, CFG>(out, segmark, inlds, lane, desc, relm, qs, tbase4, dsc, dsc2); dsc = dsc2; }
            X += 64u * nb; nleft -= nb; out.produced = O + X;
            if (out.produced - out.flushed >= out.fl) out.flush_blocks();
        } while (nleft);
    }
    if (X < T) {
        if (!have) dsc = map_step_sm(segmark, lane, desc, relm, tbase4);
        copy_step_sm<OW, CFG>(out, inlds, lane, dsc, 0u, qs, T - X, true);
        out.produced = O + T; if (out.produced - out.flushed >= out.fl) out.flush_blocks();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The back end a configuration uses: the chunked phase where sources come back from HBM (the 64 KiB formats: one 20-byte
// read per chunk instead of one byte per lane), the byte phase where the whole window lives in LDS.
template <class CFG> struct EmitUsesChunks { static constexpr bool value = CFG::FALLBACK; };

template <class OW, class CFG>
__device__ __forceinline__ void emit_begin(OW& out, DecState& s, u32 size, u64 vm, u32 len, u32 desc, u32 tend, u8* scratch, int lane, u32& last_tend, u32 W, EmitState& e) {
    (void)scratch;
    emit_prologue<OW, CFG>(out, s, size, vm, len, desc, tend, lane, last_tend, W, e);
}
template <class OW, class CFG>
__device__ __forceinline__ void emit_finish(OW& out, u8* scratch, const u8* inlds, int lane, EmitState& e) {
    if constexpr (EmitUsesChunks<CFG>::value) emit_steps<OW, CFG>(out, scratch, inlds, lane, e);
    else if constexpr (CFG::STEPMARKS) byte_emit_steps_sm<OW, CFG>(out, scratch, inlds, lane, e);
    else byte_emit_steps<OW, CFG>(out, scratch, inlds, lane, e);
}

// The "window" of a wavefront that only PARSES (two wavefronts per stream, alz_decode_fast2_kernel): it knows how far the output has
// got and how much room there is -- all the token prologue needs -- and hands every batch of tokens, prologue done, to the
// executing wavefront through a two-slot LDS mailbox: per lane (length after the cuts | 0 = not executed, offset, descriptor), per
// batch (kind, output position, bytes, window).  One workgroup barrier per batch.
#define ALZ_MBOX_WORDS (3u * 64u + 16u)
struct WalkOut {
    static constexpr bool FB = false;
    static constexpr bool PUBLISH = true;
    u32 produced, cap;
    u32* mbox; u32 k;
    __device__ __forceinline__ void publish(const EmitState& e, int lane) {
        u32* slot = mbox + (k & 1u) * ALZ_MBOX_WORDS;
        slot[lane] = e.kept ? e.clen : 0u; slot[64 + lane] = e.off; slot[128 + lane] = e.desc;
        if (lane == 0) { slot[192] = 1u; slot[193] = e.O; slot[194] = e.T; slot[195] = e.W; }
        __syncthreads();
        k++;
    }
};

// Returns true when the stream is finished (declared size reached, or capacity hit).
template <class OW, class CFG>
__device__ __forceinline__ bool fast_emit(OW& out, DecState& s, u32 size, u64 vm, u32 len, u32 desc, u32 tend, u8* scratch,
                                          const u8* inlds, int lane, u32& last_tend, u32 W) {
    EmitState e;
    emit_prologue<OW, CFG>(out, s, size, vm, len, desc, tend, lane, last_tend, W, e);
    if constexpr (OW::PUBLISH) {                               // a parsing wavefront: the batch goes to the executing one
        out.publish(e, lane);
        out.produced = e.O + e.T;
        return e.fin;
    } else {
    emit_finish<OW, CFG>(out, scratch, inlds, lane, e);
    return e.fin;
    }
}
