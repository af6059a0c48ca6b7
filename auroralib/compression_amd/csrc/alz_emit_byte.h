// alz_emit_byte.h -- the byte-per-lane back end (round 1): 64 output bytes per step, one per lane.  It stays the back end of
// every configuration whose whole window lives in LDS: on token-dense streams (the synthetic mix: ~7.5 output bytes per
// token, every sixth match reaching into the 500 bytes in front of it) the chunked phase of alz_emit_chunk.h needs 3-4
// re-read passes per step and loses (Yaz0 4.7 ms against 3.3 ms per 10 000 x 256 KiB; docs/EXPERIMENTS.md 4.4), while this phase
// resolves sources inside its 64-byte step in registers (pointer jumping over ds_bpermute).
#pragma once
#include "alz_emit_chunk.h"

// One step of the byte phase: 64 consecutive output bytes, one per lane.  The kernel is bound by VALU issue (one wave
// instruction per 4 cycles per SIMD), so the step is written for the fewest vector instructions:
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

template <class OW, class CFG, bool EARLY, bool FULL>
__device__ __forceinline__ void byte_step(OW& out, u8* segmark, const u8* inlds, int lane, u32 desc, u32& relm, u32& qs, u32& tbase4, u32 nseg) {
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

// Two-pass form of the step for configurations with HBM read-back (see fast_emit): map_step finds the descriptor of
// this lane's byte, copy_step moves the byte.
__device__ __forceinline__ u32 map_step(u8* segmark, int lane, u32 desc, u32& relm, u32& tbase4) {
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
__device__ __forceinline__ void copy_step(OW& out, const u8* inlds, int lane, u32 dsc, u32 far, u32& qs, u32 nseg, bool early) {
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
__device__ __forceinline__ void fused_step(OW& out, u8* segmark, const u8* inlds, int lane, u32 desc, u32& relm, u32& qs, u32& tbase4, const u32 dsc, u32& dsc_next) {
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
__device__ __forceinline__ void byte_emit_steps(OW& out, u8* segmark, const u8* inlds, int lane, const EmitState& e) {
    u32 desc = e.desc;
    if (CFG::LITRUN) {                                        // literal run: (input-cache index of the run) - (slot coordinate of its first byte)
        if (desc >> 31) desc = 0x80000000u | ((desc - (e.O + e.off + out.oshift)) & 2047u);
    }
    const u32 O = e.O, T = e.T, W = e.W;
    u32 X = 0;
    u32 qs = O + (u32)lane + out.oshift;                     // slot coordinate of this lane's byte in the current step
    u32 tbase4 = 0;                                          // tbase4: 4 x (tokens that ended before the current step)
    u32 relm = e.kept ? e.off + e.clen - 1u : 0xFFFFFF00u;   // my token's LAST byte relative to the current step (huge: none)
    while (X + 64u <= T && O + X < W) { byte_step<OW, CFG, true, true>(out, segmark, inlds, lane, desc, relm, qs, tbase4, 64u); X += 64u; out.produced = O + X; if (out.produced - out.flushed >= out.fl) out.flush_blocks(); }
    u32 nleft = (T - X) >> 6;
    u32 dsc = 0; bool have = false;
    if (nleft) {
        dsc = map_step(segmark, lane, desc, relm, tbase4); have = true;
        do {
            const u32 pos = O + X + out.oshift;
            u32 nb = (out.fl - (pos & (out.fl - 1u)) + 63u) >> 6;
            if (nb > nleft) nb = nleft;
            u32 k = nb, dsc2;
            for (; k >= 2u; k -= 2u) {
                fused_step<OW, CFG>(out, segmark, inlds, lane, desc, relm, qs, tbase4, dsc, dsc2);
                fused_step<OW, CFG>(out, segmark, inlds, lane, desc, relm, qs, tbase4, dsc2, dsc);
            }
            if (k) { fused_step<OW, CFG>(out, segmark, inlds, lane, desc, relm, qs, tbase4, dsc, dsc2); dsc = dsc2; }
            X += 64u * nb; nleft -= nb; out.produced = O + X;
            if (out.produced - out.flushed >= out.fl) out.flush_blocks();
        } while (nleft);
    }
    if (X < T) {
        if (!have) dsc = map_step(segmark, lane, desc, relm, tbase4);
        copy_step<OW, CFG>(out, inlds, lane, dsc, 0u, qs, T - X, true);
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
