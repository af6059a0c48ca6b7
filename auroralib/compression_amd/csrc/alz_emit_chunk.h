// alz_emit_chunk.h -- the shared back end of every lane-parallel decoder: executes up to 64 parsed tokens per call.
//
// This is the reference's LzWindows.BackCopy / WriteByte / CopyFrom (IO/LzWindows.cs:72-100, :232-237, :124-135) for a whole
// batch of tokens at once.  Round 1 moved ONE output byte per lane per step (64 bytes per step, ~0.6 vector instructions
// per output byte); this version moves one CHUNK -- up to 16 consecutive output bytes of one token -- per lane per step:
//
//   * a match / literal-run token of n bytes is cut into ceil(n / 16) chunks; a step executes 64 chunks (up to 1 KiB);
//     single literals are not chunks: their token lanes store the byte straight into the window (ds_write_b8);
//   * chunk -> token: token lanes mark their last chunk (64-entry LDS mark array), ballot + mbcnt give every chunk lane
//     the rank of its token among the chunked tokens, one ds_read_b64 fetches the token from a table the token lanes
//     filled by rank;
//   * a chunk is moved in the DESTINATION's dword alignment: gfx950 executes byte-unaligned LDS accesses one lane per
//     cycle (tools/ubench_lds.hip: 56 cycles per wave instruction against 10-18 for dword-aligned ones), so the lane reads
//     24 source bytes at a dword-aligned address (ds_read_b128 + ds_read_b64), funnels them with five v_alignbyte into the
//     five dwords the chunk touches in the window, and stores head and tail dword under a byte mask (ds_mskor_b32) and
//     the dwords between them whole;
//   * sources that are produced in the same step: all reads of a step precede its writes, and lanes whose source range
//     reaches into the step's own output re-read and re-write until nothing changes (the dependency graph is a DAG --
//     sources lie strictly earlier --, so the fixed point is unique; typical depth 1).  Self-overlapping matches
//     (distance < length) do not chain through this: distances 1, 2 and 4 are replicated in registers, for every other
//     distance chunk c reads at s - d + (16 c mod d), i.e. from the pattern in front of the token (+ its first chunk);
//   * windows longer than the LDS ring (64 KiB formats): sources older than the ring come back from the stream's own
//     output in HBM as ONE 20-byte read per chunk (round 1: one byte per lane);
//   * tokens longer than 1 KiB run alone, 64 chunks per step, with sources folded to lie before the step.
//
// Frozen edge semantics: E1 (distance 0) is resolved by the parsers, E2 (source before the stream start reads
// 0x00) by the zero-filled ring (HBM sources: explicit), E4 / E5 (declared size / capacity) by the token prologue.
#pragma once
#include "alz_decode_serial.h"

#define ALZ_EMIT_SCRATCH 640u     /* LDS scratch of the chunked phase: 128 B of marks + 64 x 8 B token table */
#define ALZ_BYTE_SCRATCH 384u     /* LDS scratch of the byte phase with CFG::DESCTAB: 128 B of marks + 64 x 4 B descriptor table (128 B of marks without) */
#define ALZ_LONGTOK 1024u         /* tokens above this run alone (their fields would not fit the table entry) */

__device__ __forceinline__ u64 wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
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

// descriptor of a token for the byte phase: bit31 = literal (bits 0..7 the byte; LITRUN configurations: bits 0..16 the
// input-cache index of a literal RUN), else bits 0..16 the distance
#define ALZ_DESC_LIT(b) (0x80000000u | (b))
#define ALZ_DESC_MATCH(d) (d)
#define ALZ_DESC_DIST(x) ((x) & 0x1FFFFu)

// compile-time configuration of the byte phase
//   OMASK    ring mask when it is a compile-time constant (0: use out.lw_mask)
//   LZSS     descriptors hold ring offsets that become distances once the output position is known
//   LITRUN   bit-31 descriptors are literal RUNS copied from the LDS input cache (LZ4 / LZO / Snappy ...), otherwise ONE
//            literal byte (flag-byte formats, PRS)
//   FALLBACK the LDS ring is shorter than the format's window (64 KiB formats keep 4 KiB): older sources are read back
//            from the stream's own output in HBM (flush_to() made them visible)
//   RUNGLOBAL literal runs are read from the stream's INPUT in global memory (`inlds` is then src + the offset that turns a run's
//            cache index into an input offset), like a far source: the executing wavefront of a two-wavefront kernel has no
//            input cache of its own while the parsing wavefront slides its cache as it pleases
//   STEPMARKS the byte phase writes, reads and clears its marks every step (rounds 1-4) instead of once per block of two steps under a tag
//            (round 5): more LDS instructions, no scalar ones -- the PRS kernels are bound by the CU's scalar unit (alz_emit_byte.h)
//   DESCTAB  the byte phase keeps the batch's descriptors in a 64-entry LDS table behind its marks (the kernel allocates ALZ_BYTE_SCRATCH) and a
//            byte fetches its token's with a ds_read_b32 instead of a ds_bpermute_b32 from the token lane's register -- for the kernels that
//            have the 256 bytes to spare: the three-cursor formats (small input caches).  The single-cursor kernels sit 96 bytes below the
//            allocation step at which a CU holds 24 instead of 25 of their waves (+8 % time; with 512-byte cache chunks the table is worth ~1 %)
template <u32 OMASK_, bool LZSS_, bool LITRUN_, bool FALLBACK_, bool RUNGLOBAL_ = false, bool STEPMARKS_ = false, bool DESCTAB_ = false>
struct EmitCfg { static constexpr u32 OMASK = OMASK_; static constexpr bool LZSS = LZSS_, LITRUN = LITRUN_, FALLBACK = FALLBACK_, RUNGLOBAL = RUNGLOBAL_, STEPMARKS = STEPMARKS_, DESCTAB = DESCTAB_; };

typedef u32 alz_v4 __attribute__((ext_vector_type(4), aligned(4)));   // 16 / 8 bytes at a dword-aligned address
typedef u32 alz_v2 __attribute__((ext_vector_type(2), aligned(4)));
typedef __attribute__((address_space(3))) u8 lds_u8;
__device__ __forceinline__ u32 lds_addr(const u8* p) { return (u32)(uintptr_t)(const lds_u8*)p; }
// MEM[a] = (MEM[a] & ~mask) | data : a dword store under a byte mask (data already masked)
__device__ __forceinline__ void lds_mskor(u32 a, u32 mask, u32 data) { asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"(a), "v"(mask), "v"(data) : "memory"); }

// a mod d for a < 2^24, d >= 1 (v_rcp_f32 is within 1 ulp: the quotient can only come out one too small)
__device__ __forceinline__ u32 umod24(u32 a, u32 d) {
    const u32 qf = (u32)((float)a * __builtin_amdgcn_rcpf((float)d));
    u32 r = a - qf * d;
    if (r >= d) r -= d;
    return r;
}

// The state of one batch of tokens between its phases (kept as a struct so that callers can split the work: emit_begin =
// token prologue, emit_finish = the steps; a caller may parse the next batch in between).
struct EmitState {
    bool kept, fin;      // this lane's token is executed; the stream is finished after this batch
    u32 off, clen;       // output offset of the token inside the batch, its length after the size / capacity cuts
    u32 desc;            // distance | literal byte | cache index of a run
    u32 O, T, W;         // output position of the batch, bytes it produces, the format's window (E2 of the byte phase)
};

// mask of the lanes below n (n <= 64), on the scalar unit
__device__ __forceinline__ u64 lanes_below(u32 n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }

// Token prologue.  Per-lane token: len (>= 1), desc, tend = input offset just past the token; `vm` = the mask of the lanes that hold a
// token (a scalar pair: every caller knows it as one -- a prefix of the lanes, or the ballot of one compare -- and a predicate that IS a
// lane mask costs nothing to use, while the ballot of a compound predicate costs two vector instructions to materialise).  For LZSS the
// descriptor holds the ring OFFSET and becomes a distance here, once the token's output position is known
// (LzWindows.OffsetCopy  IO/LzWindows.cs:108-115).  The size / capacity rules (E4, E5) are prefix cuts.
template <class OW, class CFG>
__device__ __forceinline__ void emit_prologue(OW& out, DecState& s, u32 size, u64 vm, u32 len, u32 desc, u32 tend, int lane, u32& last_tend, u32 W, EmitState& e) {
    const bool valid = __builtin_amdgcn_inverse_ballot_w64(vm);
    const u32 end = wave_incl_scan(valid ? len : 0u, lane);
    const u32 off = end - len;
    const u32 O = out.produced;
    const u32 left = size - O;                               // > 0 (caller guarantees produced < size)
    const u64 km = vm & wave_ballot(off < left);             // the token exists in the stream (prefix of lanes)
    const bool keep = __builtin_amdgcn_inverse_ballot_w64(km);
    const u32 nk = (u32)__popcll(km);
    const u32 Tend = wave_readlane(end, nk - 1);
    bool fin = km != vm || Tend >= left;                     // (km is a subset of vm: "fewer kept than valid" as one scalar compare of the masks; `nk < nvalid` became a 64-bit VECTOR compare)
    u32 lastk = nk - 1;
    u32 T = Tend;
    const u32 room = out.cap - O;
    if (Tend > room) {                                       // E5: first token whose output would exceed dst_cap
        const u64 om = km & wave_ballot(end > room);
        lastk = (u32)__builtin_ctzll(om);
        s.ovf = true; s.attempted_end = (u64)O + wave_readlane(end, lastk);
        T = room; fin = true;
    }
    last_tend = wave_readlane(tend, lastk);
    if (CFG::LZSS) {
        if (!(desc >> 31)) {
            const u32 offset = ALZ_DESC_DIST(desc);
            const u32 pos = (O + off) & (W - 1);
            u32 d = (pos - offset) & (W - 1);
            if (d == 0) d = W;                               // E1
            desc = ALZ_DESC_MATCH(d);
        }
    }
    e.kept = keep && off < T;
    e.off = off; e.clen = (T - off < len) ? T - off : len;
    e.desc = desc; e.O = O; e.T = T; e.W = W; e.fin = fin;
}

// The five window dwords of a chunk, stored under their byte masks (ds_mskor_b32; a zero mask leaves the dword alone).
__device__ __forceinline__ void chunk_store(bool wr, u32 base, u32 M0, u32 M1, u32 M2, u32 M3, u32 M4, u32 E0, u32 E1, u32 E2, u32 E3, u32 E4) {
    if (wr) {
        lds_mskor(base, M0, E0 & M0); lds_mskor(base + 4u, M1, E1 & M1); lds_mskor(base + 8u, M2, E2 & M2);
        lds_mskor(base + 12u, M3, E3 & M3); lds_mskor(base + 16u, M4, E4 & M4);
    }
}
// A chunk that crosses the end of the ring, or lies in its first bytes, is stored a second time: only the dwords on the
// other side of the seam (the mirror behind the ring, see OutWin::slack_dirty).  a0 = offset of dword 0 in the ring.
__device__ __forceinline__ void chunk_mirror(bool wr, u32 wbase, u32 a0, u32 LW, u32 jt, u32 M0, u32 M1, u32 M2, u32 M3, u32 M4, u32 E0, u32 E1, u32 E2, u32 E3, u32 E4) {
    const bool cross = wr && a0 + 4u * jt >= LW, head = wr && a0 < ALZ_WIN_SLACK;
    if (wave_ballot(cross || head)) {
        const u32 a1 = cross ? a0 - LW : a0 + LW, lo = cross ? 0u : LW;          // wanted: lo <= a1 + 4 j < lo + slack
#define ALZ_MM(j, M) ((a1 + 4u * (j) - lo) < ALZ_WIN_SLACK ? (M) : 0u)
        chunk_store(cross || head, wbase + a1, ALZ_MM(0u, M0), ALZ_MM(1u, M1), ALZ_MM(2u, M2), ALZ_MM(3u, M3), ALZ_MM(4u, M4), E0, E1, E2, E3, E4);
#undef ALZ_MM
    }
}
// byte mask of window dword j from the 20-bit byte map of the chunk (bit x = window byte x belongs to the chunk)
__device__ __forceinline__ u32 chunk_mask(u32 bmap, u32 j) {
    const u32 nib = (bmap >> (4u * j)) & 0xFu;
    const u32 t = (nib * 0x204081u) & 0x01010101u;          // bit k of the nibble -> bit 0 of byte k (a 24-bit multiply: full rate)
    return (t << 8) - t;                                    // x 0xFF without v_mul_lo_u32 (a quarter-rate instruction)
}
// 24 source bytes at a dword-aligned LDS address, funnelled to the five window dwords (t = byte offset of the first one)
__device__ __forceinline__ void chunk_read(const lds_u8* sbase, u32 t, u32& N0, u32& N1, u32& N2, u32& N3, u32& N4) {
    const alz_v4 ra = *reinterpret_cast<const __attribute__((address_space(3))) alz_v4*>(sbase);
    const alz_v2 rb = *reinterpret_cast<const __attribute__((address_space(3))) alz_v2*>(sbase + 16);
    N0 = __builtin_amdgcn_alignbyte(ra.y, ra.x, t); N1 = __builtin_amdgcn_alignbyte(ra.z, ra.y, t);
    N2 = __builtin_amdgcn_alignbyte(ra.w, ra.z, t); N3 = __builtin_amdgcn_alignbyte(rb.x, ra.w, t);
    N4 = __builtin_amdgcn_alignbyte(rb.y, rb.x, t);
}
// dword-periodic patterns (distance 1, 2, 4): pattern P0.. = window bytes b..; window byte x is P[(x - b) mod d]
__device__ __forceinline__ void chunk_rep(u32 d, u32 b, u32& N0, u32& N1, u32& N2, u32& N3, u32& N4) {
    const u32 p = __builtin_amdgcn_alignbyte(N1, N0, b);
    const u32 u = d == 1u ? (p & 0xFFu) * 0x01010101u : (d == 2u ? (p & 0xFFFFu) * 0x00010001u : p);
    const u32 w = __builtin_amdgcn_alignbyte(u, u, (4u - b) & 3u);
    N0 = w; N1 = w; N2 = w; N3 = w; N4 = w;
}

// One step of the byte phase: every active lane moves one chunk -- n <= 16 bytes at output position q, bytes i0.. of its
// token -- and the token lanes with `lit` store their literal byte.  `run`: the source is the LDS input cache at index d
// (+ i0), otherwise the window at distance d.  Xs = output position up to which everything is final (the step's own
// output starts there).  LATER: step >= 1 of a token that runs alone (sources are folded to lie before Xs).
template <class OW, class CFG, bool LATER>
__device__ __forceinline__ void chunk_copy(OW& out, const u8* inlds, bool act, u32 q, u32 n, u32 i0, bool run, u32 d, bool dchunks, u32 Xs,
                                           bool lit, u32 litq, u32 litb) {
    const u32 omask = CFG::OMASK ? CFG::OMASK : out.lw_mask;
    const u32 LW = omask + 1u;
    u8* const win = out.win;
    const u32 wbase = lds_addr(win);
    const u32 qa = q + out.oshift;
    const u32 b = qa & 3u;                                  // the chunk starts at byte b of its first window dword
    const u32 a0 = qa & omask & ~3u;
    const bool isrun = CFG::LITRUN && run;
    const bool grun = CFG::RUNGLOBAL && act && isrun;        // a literal run that comes from global memory
    const bool m = act && !isrun;
    // ---- where the bytes come from
    const bool ovl = m && d < i0 + n;                       // the plain source range q - d .. would reach into the token itself
    const bool rep = ovl && (d == 1u || d == 2u || d == 4u);   // dword-periodic patterns: replicated in registers
    u32 sp = q - d;                                         // source POSITION of chunk byte 0 (may be "negative": E2)
    if (wave_ballot(ovl)) {
        if (LATER) {                                        // smallest multiple of d that puts the whole source in front of Xs
            const u32 x = q + n - Xs, dd = ovl ? d : 1u;
            const u32 a = x + dd - 1u;
            const u32 D = a - umod24(a, dd);
            if (ovl) sp = q - D;
        } else if (wave_ballot(ovl && i0 != 0u && !rep && !dchunks)) {  // the pattern in front of the token (+ its first chunk)
            const u32 r = umod24(i0, ovl ? d : 1u);                 // (dchunks: the token's chunks are d bytes, i0 is a multiple of d)
            if (ovl) sp = q - i0 - d + ((rep || dchunks) ? 0u : r);
        } else if (ovl) sp = q - i0 - d;
    }
    bool dep = m && (int)(sp + (rep ? d : n) - Xs) > 0;     // the source reaches into this step's own output
    if (LATER) dep = dep && !ovl;
    bool far = false;
    if (CFG::FALLBACK) far = m && q - sp > LW - 1536u;      // older than the ring keeps intact: already flushed to HBM (never dep)
    // source window: the 20 bytes that land in the chunk's five window dwords start at source byte -b
    u32 sa;                                                 // LDS byte address of that window (ring / input cache)
    if (isrun) sa = CFG::RUNGLOBAL ? wbase : lds_addr(inlds) + d + i0 - b;
    else sa = wbase + ((sp + out.oshift - b) & omask);
    // (t = sa & 3 and the aligned address are re-derived from `sa` at every read, the last dword touched at every mirror check:
    // three registers fewer across the passes -- these kernels sit at the 80-register step of 6 waves per SIMD)
#define ALZ_SBASE ((const lds_u8*)(uintptr_t)(sa & ~3u))
#define ALZ_ST (sa & 3u)
#define ALZ_JT ((b + n - 1u) >> 2)
    // ---- byte masks of the five window dwords
    const u32 bmap = ((1u << n) - 1u) << b;
    const u32 M0 = chunk_mask(bmap, 0u), M1 = chunk_mask(bmap, 1u), M2 = chunk_mask(bmap, 2u), M3 = chunk_mask(bmap, 3u), M4 = chunk_mask(bmap, 4u);
    // ---- pass 1: every chunk
    u32 E0 = 0, E1 = 0, E2 = 0, E3 = 0, E4 = 0;
    if (CFG::FALLBACK) {
        if (wave_ballot(far || grun)) {
            const u32 fp = sp - b;                          // position of the window's first byte
            const bool slow = wave_ballot(far && (int)fp < 0) != 0ull;   // E2 / the very start of the stream: byte-wise, zero in front of position 0
            if (slow) {
                if (far) {
#pragma unroll 1
                    for (u32 j = 0; j < 5u; j++) {          // (E0..E4 as a shift register: no indexed array, no scratch)
                        u32 w = 0;
#pragma unroll
                        for (u32 k = 0; k < 4u; k++) { const u32 pj = fp + 4u * j + k; if ((int)pj >= 0 && pj < out.flushed) w |= (u32)__hip_atomic_load(out.dst + pj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << (8u * k); }
                        E0 = E1; E1 = E2; E2 = E3; E3 = E4; E4 = w;
                    }
                }
            }
            const bool gl = (far && !slow) || grun;          // one 20-byte read: a far source from the stream's output, a literal run from its input
            if (gl) {
                const u8* gp = grun ? inlds + d + i0 - b : out.dst + fp;
                uint4 g; u32 g4;
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dword %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(g), "=&v"(g4) : "v"(gp) : "memory");
                E0 = g.x; E1 = g.y; E2 = g.z; E3 = g.w; E4 = g4;
            }
        }
    }
    if (act && !far && !grun) chunk_read(ALZ_SBASE, ALZ_ST, E0, E1, E2, E3, E4);
    if (wave_ballot(rep)) { if (rep) chunk_rep(d, b, E0, E1, E2, E3, E4); }
    // single literals: stored after the step's reads (their slots may still hold the bytes a distance == W match wants)
    if (wave_ballot(lit)) {
        const u32 sl = (litq + out.oshift) & omask;
        if (lit) win[sl] = (u8)litb;
        if (wave_ballot(lit && sl < ALZ_WIN_SLACK)) { if (lit && sl < ALZ_WIN_SLACK) win[sl + LW] = (u8)litb; }
    }
    chunk_store(act, wbase + a0, M0, M1, M2, M3, M4, E0, E1, E2, E3, E4);
    chunk_mirror(act, wbase, a0, LW, ALZ_JT, M0, M1, M2, M3, M4, E0, E1, E2, E3, E4);
    // ---- chunks whose source reaches into this step's own output: read again until nothing changes.  Chunks ascend with the
    // lane, so when every such source ends in front of the FIRST dependent chunk it was written by pass 1 and one more pass is
    // final -- the common case (a match right behind the token it copies): no third read to find that nothing changed.
    const u64 depm = wave_ballot(dep);
    bool shallow = false;
    if (depm) {
        const u32 dmin = wave_readlane(q, (u32)__builtin_ctzll(depm));
        shallow = !wave_ballot(dep && (int)(sp + (rep ? d : n) - dmin) > 0);
    }
    bool again = depm != 0;
    while (again) {
        u32 N0 = E0, N1 = E1, N2 = E2, N3 = E3, N4 = E4;
        if (dep) chunk_read(ALZ_SBASE, ALZ_ST, N0, N1, N2, N3, N4);
        if (wave_ballot(rep && dep)) { if (rep && dep) chunk_rep(d, b, N0, N1, N2, N3, N4); }
        const bool wr = dep && (N0 != E0 || N1 != E1 || N2 != E2 || N3 != E3 || N4 != E4);
        if (!wave_ballot(wr)) break;
        again = !shallow;
        E0 = N0; E1 = N1; E2 = N2; E3 = N3; E4 = N4;
        chunk_store(wr, wbase + a0, M0, M1, M2, M3, M4, E0, E1, E2, E3, E4);
        chunk_mirror(wr, wbase, a0, LW, ALZ_JT, M0, M1, M2, M3, M4, E0, E1, E2, E3, E4);
    }
#undef ALZ_SBASE
#undef ALZ_ST
#undef ALZ_JT
}

// Execution of one batch of tokens (after emit_prologue).  `scratch`: ALZ_EMIT_SCRATCH bytes of LDS (marks zeroed by the
// kernel), `inlds`: the LDS input cache literal runs point into (LITRUN configurations).
template <class OW, class CFG>
__device__ __forceinline__ void emit_steps(OW& out, u8* scratch, const u8* inlds, int lane, const EmitState& e) {
    const u32 omask = CFG::OMASK ? CFG::OMASK : out.lw_mask;
    u8* const segmark = scratch;
    uint2* const table = reinterpret_cast<uint2*>(scratch + 128);
    if (out.slack_dirty) {                                   // byte-wise writers ran since the last batch: refresh the mirror of the ring's head
        if (lane < (int)(ALZ_WIN_SLACK / 4u)) *reinterpret_cast<u32*>(out.win + omask + 1u + 4u * (u32)lane) = *reinterpret_cast<const u32*>(out.win + 4u * (u32)lane);
        out.slack_dirty = false;
        wave_sync();
    }
    const u32 O = e.O;
    const bool islit = e.kept && (e.desc >> 31) && !CFG::LITRUN;
    const bool ism = e.kept && !islit;                       // chunked tokens: matches and literal runs
    const bool isrun = CFG::LITRUN && (e.desc >> 31);
    const u32 dfield = ALZ_DESC_DIST(e.desc);
    u64 longm = wave_ballot(ism && e.clen > ALZ_LONGTOK);
    u32 seg0 = 0;
    u32 Xs = O;                                              // everything in front of this position is final
    for (;;) {
        const u32 seg1 = longm ? (u32)__builtin_ctzll(longm) : 64u;
        // ---- tokens seg0 .. seg1 - 1: mapped steps of 64 chunks
        const bool inseg = (u32)lane >= seg0 && (u32)lane < seg1;
        // chunk size: 16 bytes -- except for a self-overlapping match whose distance d < 16 is not 1, 2 or 4 (those are
        // replicated in registers): its chunks are d bytes, so that every one of them is a plain copy of the d pattern bytes
        // in front of the token (no chunk ever reads what another chunk of its token writes)
        const bool tsm = ism && inseg && !isrun && dfield < 16u && dfield < e.clen && dfield != 1u && dfield != 2u && dfield != 4u;
        u32 cs = 16u, nch = (ism && inseg) ? (e.clen + 15u) >> 4 : 0u;
        if (wave_ballot(tsm)) {
            const u32 dd = tsm ? dfield : 1u;
            const u32 a = e.clen + dd - 1u;
            if (tsm) { cs = dd; nch = (u32)((float)(a - umod24(a, dd)) * __builtin_amdgcn_rcpf((float)dd) + 0.5f); }   // ceil(len / d): an exact multiple of d over d
        }
        const bool lits = islit && inseg;
        const u32 segO = O + (seg0 < 64u ? wave_readlane(e.off, seg0) : 0u);   // output position of the segment (table offsets are relative to it)
        if (wave_ballot(nch != 0u || lits)) {
            const u32 cend = wave_incl_scan(nch, lane);
            const u32 cstart = cend - nch;
            const u32 total = wave_readlane(cend, 63);
            const u64 cm = wave_ballot(nch != 0u);
            if (nch) table[mbcnt64(cm)] = make_uint2(dfield | (isrun ? 0x20000u : 0u) | ((e.clen - 1u) << 18) | ((cs - 1u) << 28), (e.off - (segO - O)) | (cstart << 16));
            u32 relm = nch ? cend - 1u : 0xFFFFFF00u;        // my token's LAST chunk relative to the current step (huge: none)
            u32 rbase = 0;
            const u32 nsteps = (total >> 6) + 1u;
            for (u32 k = 0; k < nsteps; k++) {
                const u32 cid0 = k << 6;
                const u32 nact = total - cid0 < 64u ? total - cid0 : 64u;
                bool act = false; u32 q = 0, n = 1, i0 = 0, d = 1; bool run = false, dch = false;
                if (nact) {
                    { const u32 dump = 64u + (u32)lane; segmark[relm < dump ? relm : dump] = 1; }   // slots 64..127 are never read
                    wave_sync();
                    const u32 mk = segmark[lane];
                    segmark[lane] = 0;
                    const u64 M = wave_ballot(mk != 0);
                    const uint2 tk = table[(rbase + mbcnt64(M)) & 63u];
                    rbase += (u32)__popcll(M);
                    act = (u32)lane < nact;
                    const u32 tlen = ((tk.x >> 18) & 0x3FFu) + 1u, toff = tk.y & 0xFFFFu, tcs = tk.y >> 16, csz = (tk.x >> 28) + 1u;
                    i0 = (cid0 + (u32)lane - tcs) * csz;
                    if (act) { n = tlen - i0 < csz ? tlen - i0 : csz; d = tk.x & 0x1FFFFu; run = (tk.x & 0x20000u) != 0u; dch = csz != 16u; }
                    else i0 = 0;
                    q = segO + toff + i0;
                }
                chunk_copy<OW, CFG, false>(out, inlds, act, q, n, i0, run, d, dch, Xs, lits && (cstart >> 6) == k, O + e.off, e.desc & 0xFFu);
                relm -= 64u;
                // everything in front of the first chunk of the next step (or, behind the last step, of the next segment) is final
                Xs = nact == 64u ? wave_readlane(q + n, 63) : (seg1 < 64u ? O + wave_readlane(e.off, seg1) : O + e.T);
                out.produced = Xs;
                if (out.produced - out.flushed >= out.fl) out.flush_blocks();
            }
        }
        if (!longm) break;
        // ---- token seg1 runs alone: 64 chunks per step
        {
            const u32 tlen = wave_readlane(e.clen, seg1), toff = wave_readlane(e.off, seg1), td = wave_readlane(dfield, seg1);
            const bool trun = CFG::LITRUN && (wave_readlane(e.desc, seg1) >> 31);
            const u32 s0 = O + toff;
            Xs = s0; out.produced = Xs;
            if (out.produced - out.flushed >= out.fl) out.flush_blocks();
            for (u32 base = 0; base < tlen; base += 1024u) {
                const u32 i0 = base + 16u * (u32)lane;
                const bool act = i0 < tlen;
                const u32 n = act ? (tlen - i0 < 16u ? tlen - i0 : 16u) : 1u;
                if (base == 0) chunk_copy<OW, CFG, false>(out, inlds, act, s0 + i0, n, act ? i0 : 0u, trun, td, false, Xs, false, 0u, 0u);
                else chunk_copy<OW, CFG, true>(out, inlds, act, s0 + i0, n, act ? i0 : 0u, trun, td, false, Xs, false, 0u, 0u);
                Xs = s0 + (tlen - base < 1024u ? tlen : base + 1024u);
                out.produced = Xs;
                if (out.produced - out.flushed >= out.fl) out.flush_blocks();
            }
        }
        longm &= longm - 1u;
        seg0 = seg1 + 1u;
    }
    out.produced = O + e.T;
    if (out.produced - out.flushed >= out.fl) out.flush_blocks();
}

