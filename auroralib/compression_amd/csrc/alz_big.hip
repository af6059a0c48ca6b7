// alz_big.hip -- ONE big stream on the whole GPU: the three-section formats (Yay0, MIO0) here, the interleaved flag-byte formats (LZSS, LZ10,
// LZ11, Yaz0) further down.
//
// The production kernels give a stream one wavefront (or two): a lone 256 KiB stream takes 0.8 ms, a lone 1 MiB stream 3 ms -- below the
// 0.40-0.79 GiB/s the managed decoders reach on one CPU core (the reference's own benchmark is ONE 1 000 KiB stream,
// Benchmarks/Benchmarks/TestAllAlgorithms.cs:41-42).  Yay0 (Nintendo/Yay0.cs:110-144) and MIO0 (Nintendo/MIO0.cs:105-149) keep their
// flag bits, match tokens and literals in three separate sections, so nothing about a token depends on the tokens before it except three
// COUNTS: how many match tokens came before (where its two bytes are), how many bytes of the literal section are used up (where its
// literal / Yay0's length byte is), and how many bytes of output exist (where it writes).  Counts are prefix sums:
//
//   tiles of 1 024 tokens (one wavefront each)            single workgroup
//   K1  matches per tile (popcount of the flag bytes)     S   exclusive scan over the tiles
//   K3  literal-section bytes per tile (Yay0 only)        S
//   K5  output bytes per tile                             S
//   K7  every token knows its three cursors: (output offset, length, descriptor, source.Position behind it) per token
//   B   one thread per OUTPUT byte finds its token by binary search over the offsets and writes its entry: either the byte (a literal) or
//       the POSITION the byte is copied from -- for a self-overlapping match (distance < length) the position in front of the token that
//       holds the same pattern byte (start - d + (j mod d)), so a byte's source always lies in front of its token and a source "before
//       the stream start" (E2) is the literal 0x00
//   J   pointer jumping over the output bytes: an unresolved byte takes over its source's entry -- the byte (done) or the source's source
//       (jump); after r rounds an entry spans >= 2^r hops, the depth of a chain is at most the number of tokens: ceil(log2 tokens) + 1
//       rounds, launched as that many kernels that return at once when the round before changed nothing
//   W   the bytes go to the destination; the stream's result is written
//
// Only a stream that is VALID gets its result from here -- every read inside the input, the output ending exactly at the declared size,
// room for it in the destination.  Anything else (truncated, overshooting or undershooting streams, a short destination) opens a gate
// and the exact production kernel, enqueued behind this path, decodes the stream again with the reference's error semantics; for a valid
// stream it finds the gate closed and returns.  So the statuses, src_used and partial outputs of malformed streams are the production
// kernels', and parity for them is theirs (tests/test_gpu_big_stream.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "alz_internal.h"
#include "alz_prs_table.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

#define BIG_TILE 1024u          /* tokens per tile (one wavefront, 16 rounds of 64 tokens) */
#ifndef BIG_LOG
#define BIG_LOG 2u              /* rounds of list ranking / pointer jumping per launch (big_rank_round, big_jump) */
#endif
#define BIG_HOPS (1u << BIG_LOG)
#define BIG_LIT 0x80000000u

struct BigArgs {
    const u8* src; u8* dst;
    u32 src_len, size, aux0, aux1;
    u32 ntok;                   // tokens looked at: min(size, 8 x flag bytes the input can hold) -- an upper bound of the tokens of the stream
    u32 ntiles;
};

// ctl words
enum { C_BAD = 0, C_END = 1, C_USED = 2, C_TOTAL = 3, C_NG = 4, C_NT = 5, C_SIZE = 6 /* the output size where only the device knows it (LZ4, Snappy, PRS) */, C_TERM = 7 /* PRS: the first terminator token */, C_FLAGS = 8 /* .. C_FLAGS + rounds: "round r changed something" */ };
#define BIG_OOB 0x40000000u     /* a token that does not lie inside the input (descriptor bit) */
#define BIG_RUN 0x20000000u     /* a literal RUN: the low 29 bits are its position in the input (LZ4, Snappy) */

__device__ __forceinline__ u32 big_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 big_mbcnt(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 big_dpp_add(u32 v) { return v + (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ u32 big_incl_scan(u32 v) {
    v = big_dpp_add<0x111, 0xF>(v); v = big_dpp_add<0x112, 0xF>(v); v = big_dpp_add<0x114, 0xF>(v); v = big_dpp_add<0x118, 0xF>(v);
    v = big_dpp_add<0x142, 0xA>(v); v = big_dpp_add<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ u32 big_total(u32 incl) { return (u32)__builtin_amdgcn_readlane((int)incl, 63); }

// One round of 64 tokens, token index t0 + lane.  Cursors on entry: matches / literal-section bytes before token t0.
struct BigTok { bool lit, oob; u32 len, dist, byte, mafter, uafter; };   // mafter / uafter: the two cursors BEHIND this token
template <bool MIO0, bool NEED_U>
__device__ __forceinline__ BigTok big_round(const BigArgs& a, u32 t0, u32 mbase, u32 ubase, u32& mcount, u32& ucount) {
    const u32 lane = big_lane();
    const u32 t = t0 + lane;
    const u32 fi = t >> 3;
    BigTok k; k.oob = fi >= a.src_len;
    const u32 fb = k.oob ? 0xFFu : a.src[fi];
    k.lit = (fb >> (7u - (t & 7u))) & 1u;                       // MSB first, 1 = literal  Yay0.cs:118, MIO0.cs:123
    const u64 lm = __builtin_amdgcn_ballot_w64(k.lit);
    const u32 midx = mbase + big_mbcnt(~lm);
    mcount = (u32)__popcll(~lm);
    u32 b1 = 0, b2 = 0;
    if (!k.lit) {
        const u64 cp = (u64)a.aux0 + 2ull * midx;
        if (cp + 2u > a.src_len) k.oob = true; else { b1 = a.src[cp]; b2 = a.src[cp + 1]; }
    }
    k.dist = (((b1 & 0xFu) << 8) | b2) + 1u;
    k.mafter = midx + (k.lit ? 0u : 1u);
    k.byte = 0; k.len = 1;
    if (MIO0) {
        const u32 uidx = t - midx;                              // literals before me = tokens before me - matches before me
        ucount = 64u - mcount;
        k.uafter = uidx + (k.lit ? 1u : 0u);
        if (k.lit) { const u64 up = (u64)a.aux1 + uidx; if (up >= a.src_len) k.oob = true; else k.byte = a.src[up]; }
        else k.len = (b1 >> 4) + 3u;
    } else {
        const bool usesu = k.lit || (b1 >> 4) == 0u;            // a 3-byte match takes its length from the literal section  Yay0.cs:130-131
        const u64 um = __builtin_amdgcn_ballot_w64(usesu);
        ucount = (u32)__popcll(um);
        const u32 uidx = ubase + big_mbcnt(um);
        k.uafter = uidx + (usesu ? 1u : 0u);
        if (NEED_U) {
            u32 ub = 0;
            if (usesu) { const u64 up = (u64)a.aux1 + uidx; if (up >= a.src_len) k.oob = true; else ub = a.src[up]; }   // (the managed ReadByte() == -1 rule is the exact kernel's)
            if (k.lit) k.byte = ub; else k.len = (b1 >> 4) ? (b1 >> 4) + 2u : ub + 0x12u;
        }
    }
    return k;
}

// K1: match tokens per tile
__global__ __launch_bounds__(64) void big_count_matches(BigArgs a, u32* __restrict__ tile_m) {
    const u32 tile = blockIdx.x, lane = big_lane();
    u32 cnt = 0;
    // 128 flag bytes per tile, two per lane
    for (u32 j = 0; j < 2u; j++) {
        const u32 fi = tile * (BIG_TILE / 8u) + j * 64u + lane;
        const u32 t = fi * 8u;
        if (t < a.ntok) {
            const u32 fb = fi < a.src_len ? a.src[fi] : 0xFFu;
            u32 nb = a.ntok - t; if (nb > 8u) nb = 8u;
            const u32 m = (~fb & 0xFFu) >> (8u - nb);            // the match bits of the tokens that exist (MSB first)
            cnt += (u32)__popc(m);
        }
    }
    const u32 tot = big_total(big_incl_scan(cnt));
    if (lane == 0) tile_m[tile] = tot;
}

// S: exclusive scan over the tiles, one workgroup; total (u64 clamped to 2^32 - 1) behind the last element
__global__ __launch_bounds__(1024) void big_scan(const u32* __restrict__ in, u32* __restrict__ out, u32 n, u32* __restrict__ total_out) {
    __shared__ u64 part[1024];
    const u32 tid = threadIdx.x;
    const u32 per = (n + 1023u) / 1024u;
    const u32 b = tid * per, e = b + per < n ? b + per : n;
    u64 s = 0;
    for (u32 i = b; i < e; i++) s += in[i];
    part[tid] = s;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {
        const u64 v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u64 run = tid ? part[tid - 1] : 0;
    for (u32 i = b; i < e; i++) { out[i] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)run; run += in[i]; }
    if (tid == 1023u && total_out) { const u64 t = part[1023]; *total_out = t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)t; }
}

// K3 / K5: per tile, the sum of a per-token quantity: WHAT 0 = literal-section bytes (Yay0), 1 = output bytes
template <bool MIO0, int WHAT>
__global__ __launch_bounds__(64) void big_tile_sum(BigArgs a, const u32* __restrict__ tile_mb, const u32* __restrict__ tile_ub, u32* __restrict__ tile_out) {
    const u32 tile = blockIdx.x, lane = big_lane();
    u32 mbase = tile_mb[tile], ubase = (!MIO0 && WHAT == 1) ? tile_ub[tile] : 0u;
    u32 acc = 0;
    for (u32 r = 0; r < BIG_TILE / 64u; r++) {
        const u32 t0 = tile * BIG_TILE + r * 64u;
        if (t0 >= a.ntok) break;
        u32 mc, uc;
        const BigTok k = big_round<MIO0, WHAT == 1>(a, t0, mbase, ubase, mc, uc);
        const bool exists = t0 + lane < a.ntok;
        if (WHAT == 0) acc += uc;                                // (wave-uniform; tokens behind ntok only exist in the last tile, whose count feeds no base)
        else acc += exists ? k.len : 0u;
        mbase += mc; ubase += uc;
    }
    if (WHAT == 0) { if (lane == 0) tile_out[tile] = acc; }
    else { const u32 tot = big_total(big_incl_scan(acc)); if (lane == 0) tile_out[tile] = tot; }
}

// K7: every token's (output offset, length, descriptor, source.Position behind it) -- the per-byte entries are written by big_emit_bytes
// (below: one thread per OUTPUT byte, binary search for its token), which the interleaved formats share
template <bool MIO0>
__global__ __launch_bounds__(64) void big_emit(BigArgs a, const u32* __restrict__ tile_mb, const u32* __restrict__ tile_ub, const u32* __restrict__ tile_ob,
                                               u32* __restrict__ toff, u32* __restrict__ tlen, u32* __restrict__ tdesc, u32* __restrict__ tend) {
    const u32 tile = blockIdx.x, lane = big_lane();
    u32 mbase = tile_mb[tile], ubase = MIO0 ? 0u : tile_ub[tile];
    u64 obase = tile_ob[tile];
    for (u32 r = 0; r < BIG_TILE / 64u; r++) {
        const u32 t0 = tile * BIG_TILE + r * 64u;
        if (t0 >= a.ntok) break;
        const u32 t = t0 + lane;
        if (obase >= a.size) { if (t < a.ntok) toff[t] = 0xFFFFFFFFu; continue; }   // behind the end of the stream: only the (monotone) offsets matter
        u32 mc, uc;
        const BigTok k = big_round<MIO0, true>(a, t0, mbase, ubase, mc, uc);
        const bool exists = t < a.ntok;
        const u32 end = big_incl_scan(exists ? k.len : 0u);
        if (exists) {
            const u64 off = obase + end - k.len;
            toff[t] = off > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)off;
            tlen[t] = k.len;
            tdesc[t] = k.oob ? (BIG_LIT | BIG_OOB) : (k.lit ? (BIG_LIT | k.byte) : k.dist);
            const u64 cu = (u64)a.aux0 + 2ull * k.mafter, uu = (u64)a.aux1 + k.uafter;   // source.Position = the further of the two cursors  Yay0.cs:107, MIO0.cs:148
            const u64 mx = cu > uu ? cu : uu;
            tend[t] = mx > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)mx;
        }
        obase += big_total(end); mbase += mc; ubase += uc;
    }
}

// J: one round of pointer jumping; flags[r] says whether round r left anything unresolved (flags[-1] of round 0 is preset to 1)
__global__ __launch_bounds__(256) void big_jump(u32* __restrict__ val, u32 n, const u32* __restrict__ dev_n, const u32* __restrict__ flag_prev, u32* __restrict__ flag_cur) {
    if (__builtin_nontemporal_load(flag_prev) == 0u) return;
    const u32 q = blockIdx.x * 256u + threadIdx.x;
    if (dev_n) { const u32 m = *dev_n; if (m < n) n = m; }       // (entries exist only below the size the device found)
    if (q >= n) return;
    const u32 v = val[q];
    if (v & BIG_LIT) return;
    // (BIG_HOPS - 1 hops per launch: every entry read spans at least as many hops as all entries did when the launch began, so a launch multiplies
    // the span by BIG_HOPS -- BIG_LOG rounds' worth -- whatever the other threads have or have not written yet)
    u32 w = v;
#pragma unroll
    for (u32 k = 1; k < BIG_HOPS; k++) if (!(w & BIG_LIT)) w = __hip_atomic_load(val + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    val[q] = w;
    if (!(w & BIG_LIT)) *flag_cur = 1u;
}

// what a launch sequence needs initialised, in one launch: the control words (zero; flag 0 of the pointer jumping = 1; C_NT / C_TERM as given) and
// the marks of the list ranking (zero; node 0 = 1)
__global__ __launch_bounds__(256) void big_init(u32* __restrict__ ctl, u32 nt, u32 term, u8* __restrict__ mark, u32 nmark) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < C_FLAGS + 40u) ctl[i] = i == (u32)C_FLAGS ? 1u : i == (u32)C_NT ? nt : i == (u32)C_TERM ? term : 0u;
    if (mark && i < nmark) mark[i] = i == 0u ? 1 : 0;
}

// J0: the same inside tiles of 1 024 output bytes, in LDS -- ten rounds; what is left unresolved points in front of its tile, and a chain of such
// entries crosses at least one tile boundary per hop: ceil(log2 tiles) + 1 rounds of big_jump instead of ceil(log2 tokens) + 1 (11 instead of 19
// launches for a 1 000 KiB stream -- a launch is ~4 us whatever it does)
__global__ __launch_bounds__(1024) void big_jump_tile(u32* __restrict__ val, u32 n, const u32* __restrict__ dev_n) {
    __shared__ u32 V[1024];
    if (dev_n) { const u32 m = *dev_n; if (m < n) n = m; }
    const u32 ts = blockIdx.x * 1024u, tid = threadIdx.x, q = ts + tid;
    if (ts >= n) return;
    V[tid] = q < n ? val[q] : BIG_LIT;
    __syncthreads();
    for (u32 r = 0; r < 10u; r++) {
        const u32 v = V[tid];
        const u32 w = (!(v & BIG_LIT) && v >= ts) ? V[v - ts] : v;   // (a source lies in front of its byte)
        __syncthreads();
        V[tid] = w;
        __syncthreads();
    }
    if (q < n) val[q] = V[tid];
}

// W: bytes out, result, gate
// (DEVSIZE: the output size is ctl[C_SIZE], at most `a.size` = the room in the destination)
template <bool DEVSIZE>
__global__ __launch_bounds__(256) void big_write(BigArgs a, const u32* __restrict__ val, const u32* __restrict__ ctl, alz_result* __restrict__ result, u32* __restrict__ gate, u32* __restrict__ acc) {
    const u32 size = DEVSIZE ? ctl[C_SIZE] : a.size;
    const bool ok = ctl[C_BAD] == 0u && ctl[C_END] == 1u && ctl[C_TOTAL] >= size && size <= a.size && (!DEVSIZE || size != 0u);
    const u32 q = blockIdx.x * 256u + threadIdx.x;
    if (q == 0) {
        if (ok) { alz_result r; r.dst_len = size; r.src_used = ctl[C_USED]; r.status = ALZ_ST_OK; r.reserved = 0; *result = r; *gate = 0u; atomicAdd(acc, 1u); }   // (acc: the context's count of streams this path has ACCEPTED, alz_ctx_big_stream)
        else *gate = 1u;
    }
    if (!ok || q >= size) return;
    a.dst[q] = (u8)val[q];
}

// ===============================================================================================================
// The interleaved flag-byte formats (LZSS, LZ10, LZ11, Yaz0): flag bytes, literals and match tokens share ONE byte stream, so where a
// group of eight tokens starts is only known once the group before it has been sized -- a linked list through the input.  But what a
// group that started at byte p WOULD occupy is a pure function of the bytes behind p (the speculation the wavefront kernels run per lane,
// alz_decode_fast.h), so:
//
//   P1  next[p] = p + size of "the group that starts at p", for EVERY input byte p
//   P2  list ranking: the real group starts are the nodes reachable from 0.  Round k marks jump[p] for every marked p and squares the
//       jump table (jump' = jump o jump, double-buffered: a round must see jumps of exactly 2^k hops, or nodes are skipped); after
//       ceil(log2 groups) rounds every real start is marked (and the garbage "groups" behind the end of the stream, which the cut at
//       the declared size never reaches)
//   P3  prefix sums over the marks number the groups; P4 one thread per group decodes its eight tokens into (length, descriptor, end)
//   P5  prefix sums over the lengths place the tokens in the output; P6 one thread per OUTPUT BYTE finds its token by binary search and
//       writes its entry (the byte, or the position it copies from); then the pointer jumping and the write-out of the three-cursor path.
struct BigGeom { u32 length_bits, min_length, windows_start, max_distance, W; };
template <int FMT> struct BigFam;
template <> struct BigFam<ALZ_FMT_LZSS> { static constexpr bool MSB = false, LIT1 = true,  H3 = false, H4 = false; };   // LZSS.cs:95-119
template <> struct BigFam<ALZ_FMT_LZ10> { static constexpr bool MSB = true,  LIT1 = false, H3 = false, H4 = false; };   // LZ10.cs:88-102
template <> struct BigFam<ALZ_FMT_LZ11> { static constexpr bool MSB = true,  LIT1 = false, H3 = true,  H4 = true;  };   // LZ11.cs:88-118
template <> struct BigFam<ALZ_FMT_YAZ0> { static constexpr bool MSB = true,  LIT1 = true,  H3 = true,  H4 = false; };   // Yay0.cs:118-133 on one cursor

template <int FMT>
__device__ __forceinline__ u32 big_match_bits(u32 fb) { return BigFam<FMT>::LIT1 ? (~fb & 0xFFu) : fb; }
template <int FMT>
__device__ __forceinline__ bool big_is_match(u32 mbits, u32 k) { return (mbits >> (BigFam<FMT>::MSB ? 7u - k : k)) & 1u; }
template <int FMT>
__device__ __forceinline__ u32 big_token_size(const u8* src, u32 src_len, u32 pos, bool m) {
    typedef BigFam<FMT> TR;
    if (!m) return 1u;
    if (!TR::H3 && !TR::H4) return 2u;
    const u32 nib = pos < src_len ? (u32)src[pos] >> 4 : 0xFu;
    if (TR::H3 && nib == 0u) return 3u;
    if (TR::H4 && nib == 1u) return 4u;
    return 2u;
}

// P1
template <int FMT>
__global__ __launch_bounds__(256) void big_group_sizes(const u8* __restrict__ src, u32 src_len, u32* __restrict__ next) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p > src_len) return;
    if (p == src_len) { next[p] = src_len; return; }             // the end node points at itself
    const u32 mbits = big_match_bits<FMT>(src[p]);
    u32 r = 1;
    if (!BigFam<FMT>::H3 && !BigFam<FMT>::H4) r = 9u + (u32)__popc(mbits);
    else for (u32 k = 0; k < 8u; k++) r += big_token_size<FMT>(src, src_len, p + r, big_is_match<FMT>(mbits, k));
    const u64 n = (u64)p + r;
    next[p] = n > src_len ? src_len : (u32)n;
}

// P2: one round of list ranking
// (BIG_LOG rounds per launch: the jump table to the power BIG_HOPS = 2^BIG_LOG, a marked node marks the nodes on the way -- BIG_HOPS dependent reads of
// the old table instead of BIG_LOG launches of two; a launch costs ~4 us whatever it does)
__global__ __launch_bounds__(256) void big_rank_round(const u32* __restrict__ jump_a, u32* __restrict__ jump_b, u8* __restrict__ mark, u32 nodes) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= nodes) return;
    u32 j = jump_a[p];
    const bool m = mark[p] != 0;
#pragma unroll
    for (u32 k = 1; k < BIG_HOPS; k++) { if (m) mark[j] = 1; j = jump_a[j]; }
    jump_b[p] = j;
}

// P2 on two levels (round 4, from the encoder's whole-GPU path, alz_encode_big.h): a round over all nodes is a launch (~4 us whatever it
// does), and ceil(log2 groups) + 1 = 15-18 of them were a third of a 1 000 KiB stream's time.  Jumps only go forward, so a tile of 1 024 nodes
// can be ranked in LDS: ten rounds of pointer jumping give every node its EXIT -- the first node of its chain behind the tile (big_tile_exit);
// the chain of exits from node 0 has at most one node per tile: ceil(log2 tiles) + 1 rounds over all nodes mark where the real chain ENTERS
// each tile; eleven rounds of the same ranking on the one-hop jumps in LDS mark the real nodes inside (big_tile_mark).
#define BIG_RTILE 1024u
__global__ __launch_bounds__(1024) void big_tile_exit(const u32* __restrict__ next1, u32* __restrict__ exitj, u32 nodes) {
    __shared__ u32 J[BIG_RTILE];
    const u32 ts = blockIdx.x * BIG_RTILE, te = ts + BIG_RTILE, tid = threadIdx.x, p = ts + tid;
    J[tid] = p < nodes ? next1[p] : 0xFFFFFFFFu;                  // (a thread behind the last node: never inside a tile)
    __syncthreads();
    for (u32 r = 0; r < 10u; r++) {
        const u32 j = J[tid];
        const u32 j2 = (j >= ts && j < te) ? J[j - ts] : j;       // (an end node points at itself)
        __syncthreads();
        J[tid] = j2;
        __syncthreads();
    }
    if (p < nodes) exitj[p] = J[tid];
}
__global__ __launch_bounds__(1024) void big_tile_mark(const u32* __restrict__ next1, u8* __restrict__ mark, u32 nodes) {
    __shared__ u32 JA[BIG_RTILE], JB[BIG_RTILE], M[BIG_RTILE];
    const u32 ts = blockIdx.x * BIG_RTILE, te = ts + BIG_RTILE, tid = threadIdx.x, p = ts + tid;
    JA[tid] = p < nodes ? next1[p] : 0xFFFFFFFFu;
    M[tid] = p < nodes ? mark[p] : 0u;
    __syncthreads();
    u32* ja = JA; u32* jb = JB;
    for (u32 r = 0; r < 11u; r++) {
        const u32 j = ja[tid];
        const bool inside = j >= ts && j < te;
        if (inside && M[tid]) M[j - ts] = 1u;
        jb[tid] = inside ? ja[j - ts] : j;
        __syncthreads();
        u32* t = ja; ja = jb; jb = t;
    }
    if (p < nodes) mark[p] = (u8)M[tid];
}
// the three launches of a list ranking: next1 = the one-hop jumps (kept), jump_a / jump_b = work tables, mark[0] = 1 set by the caller
static void big_rank(hipStream_t stream, const u32* next1, u32* jump_a, u32* jump_b, u8* mark, u32 nodes);

// P3: marked positions per tile of 1 024 input bytes; after the scan, the group's number -> its position
__global__ __launch_bounds__(64) void big_mark_count(const u8* __restrict__ mark, u32 src_len, u32* __restrict__ tile_c) {
    const u32 tile = blockIdx.x, lane = big_lane();
    u32 cnt = 0;
    for (u32 j = 0; j < 16u; j++) { const u32 p = tile * 1024u + j * 64u + lane; if (p < src_len && mark[p]) cnt++; }
    const u32 tot = big_total(big_incl_scan(cnt));
    if (lane == 0) tile_c[tile] = tot;
}
__global__ __launch_bounds__(64) void big_mark_scatter(const u8* __restrict__ mark, u32 src_len, const u32* __restrict__ tile_b, u32* __restrict__ gpos) {
    const u32 tile = blockIdx.x, lane = big_lane();
    u32 base = tile_b[tile];
    for (u32 j = 0; j < 16u; j++) {
        const u32 p = tile * 1024u + j * 64u + lane;
        const bool m = p < src_len && mark[p];
        const u64 mm = __builtin_amdgcn_ballot_w64(m);
        if (m) gpos[base + big_mbcnt(mm)] = p;
        base += (u32)__popcll(mm);
    }
}

// P4: the eight tokens of every group
template <int FMT>
__global__ __launch_bounds__(256) void big_group_tokens(const u8* __restrict__ src, u32 src_len, BigGeom gm, const u32* __restrict__ gpos, u32* __restrict__ ctl,
                                                        u32* __restrict__ tlen, u32* __restrict__ tdesc, u32* __restrict__ tend) {
    const u32 ng = ctl[C_NG];
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g == 0) ctl[C_NT] = 8u * ng;
    if (g >= ng) return;
    const u32 p = gpos[g];
    const u32 mbits = big_match_bits<FMT>(src[p]);
    u64 pos = (u64)p + 1u;
    for (u32 k = 0; k < 8u; k++) {
        const bool m = big_is_match<FMT>(mbits, k);
        const u32 sz = big_token_size<FMT>(src, src_len, pos > src_len ? src_len : (u32)pos, m);
        u32 len = 1, desc;
        if (pos + sz > src_len) desc = BIG_LIT | BIG_OOB;          // the token does not lie inside the input (garbage behind the end, or a truncated stream)
        else {
            const u8* t = src + pos;
            const u32 b1 = t[0];
            if (!m) desc = BIG_LIT | b1;
            else {
                const u32 b2 = t[1];
                if (FMT == ALZ_FMT_LZSS) {                       // LZSS.cs:115-119: a ring offset; it becomes a distance where the token's output position is known
                    u32 offset = ((b2 >> gm.length_bits) << 8) | b1;
                    len = (b2 & ((1u << gm.length_bits) - 1u)) + gm.min_length;
                    desc = (gm.max_distance + offset - gm.windows_start) & (gm.max_distance - 1u);
                } else if (FMT == ALZ_FMT_LZ10) { desc = (((b1 & 0xFu) << 8) | b2) + 1u; len = (b1 >> 4) + 3u; }
                else if (FMT == ALZ_FMT_LZ11) {                  // LZ11.cs:98-118
                    const u32 nib = b1 >> 4;
                    if (nib == 0u) { const u32 b3 = t[2]; desc = (((b2 & 0xFu) << 8) | b3) + 1u; len = (((b1 & 0xFu) << 4) | (b2 >> 4)) + 17u; }
                    else if (nib == 1u) { const u32 b3 = t[2], b4 = t[3]; desc = (((b3 & 0xFu) << 8) | b4) + 1u; len = (((b1 & 0xFu) << 12) | (b2 << 4) | (b3 >> 4)) + 273u; }
                    else { desc = (((b1 & 0xFu) << 8) | b2) + 1u; len = nib + 1u; }
                } else {                                         // Yaz0: Yay0.cs:127-133
                    const u32 nib = b1 >> 4;
                    desc = (((b1 & 0xFu) << 8) | b2) + 1u;
                    len = nib ? nib + 2u : (u32)t[2] + 0x12u;
                }
            }
        }
        const u32 i = 8u * g + k;
        tlen[i] = len; tdesc[i] = desc; tend[i] = pos + sz > src_len ? src_len : (u32)(pos + sz);
        pos += sz;
    }
}

// P5: token lengths per tile of 1 024 tokens; after the scan, every token's output offset (saturating: the garbage behind the end of the
// stream may add up to anything, the binary search of P6 only needs the offsets to be monotone)
__global__ __launch_bounds__(64) void big_len_count(const u32* __restrict__ tlen, const u32* __restrict__ ctl, u32* __restrict__ tile_l) {
    const u32 tile = blockIdx.x, lane = big_lane(), nt = ctl[C_NT];
    u32 acc = 0;
    for (u32 j = 0; j < 16u; j++) { const u32 i = tile * 1024u + lane * 16u + j; if (i < nt) acc += tlen[i]; }   // (<= 16 x 65 808 per lane)
    const u32 tot = big_total(big_incl_scan(acc));
    if (lane == 0) tile_l[tile] = tile * 1024u < nt ? tot : 0u;
}
__global__ __launch_bounds__(64) void big_len_offsets(const u32* __restrict__ tlen, const u32* __restrict__ ctl, const u32* __restrict__ tile_b, u32* __restrict__ toff) {
    const u32 tile = blockIdx.x, lane = big_lane(), nt = ctl[C_NT];
    if (tile * 1024u >= nt) return;
    u32 mine[16], acc = 0;
    for (u32 j = 0; j < 16u; j++) { const u32 i = tile * 1024u + lane * 16u + j; mine[j] = i < nt ? tlen[i] : 0u; acc += mine[j]; }
    const u32 incl = big_incl_scan(acc);
    u64 run = (u64)tile_b[tile] + (incl - acc);
    for (u32 j = 0; j < 16u; j++) {
        const u32 i = tile * 1024u + lane * 16u + j;
        if (i < nt) toff[i] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)run;
        run += mine[j];
    }
}

// P6: the entry of every output byte
template <bool LZSS, bool DEVSIZE = false>
__global__ __launch_bounds__(256) void big_emit_bytes(u32 size, BigGeom gm, const u32* __restrict__ toff, const u32* __restrict__ tlen, const u32* __restrict__ tdesc,
                                                      const u32* __restrict__ tend, u32* __restrict__ val, u32* __restrict__ ctl, const u8* __restrict__ src = nullptr, bool used_is_set = false) {
    const u32 q = blockIdx.x * 256u + threadIdx.x;
    if (DEVSIZE) { const u32 m = ctl[C_SIZE]; if (m > size) { if (q == 0) ctl[C_BAD] = 1u; return; } size = m; }   // (more output than the destination holds: E5, the exact kernel's)
    if (q >= size) return;
    const u32 nt = ctl[C_NT];
    if (nt == 0u) { ctl[C_BAD] = 1u; return; }
    u32 lo = 0, hi = nt;                                         // the last token whose offset is <= q
    while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (toff[mid] <= q) lo = mid; else hi = mid; }
    const u32 t = lo, off = toff[t], len = tlen[t], desc = tdesc[t];
    const u32 j = q - off;
    if (j >= len || (desc & BIG_OOB)) { ctl[C_BAD] = 1u; val[q] = BIG_LIT; return; }   // the tokens end in front of the declared size / a token of the stream reads past the input
    if (q == size - 1u) {
        if ((u64)off + len == size) { if (!used_is_set) ctl[C_USED] = tend[t]; ctl[C_END] = 1u; }   // the stream ends with this token: source.Position is just behind it (PRS: behind its terminator)
        else ctl[C_BAD] = 1u;                                    // the last match overshoots the declared size (E4): the exact kernel's case
    }
    if (desc & BIG_LIT) { val[q] = (desc & BIG_RUN) ? (BIG_LIT | src[(desc & 0x1FFFFFFFu) + j]) : desc; return; }
    u32 d = desc;
    if (LZSS) { d = ((off & (gm.W - 1u)) - desc) & (gm.W - 1u); if (d == 0u) d = gm.W; }   // LzWindows.OffsetCopy  IO/LzWindows.cs:108-115; E1
    const u32 r = j % d;                                         // byte j copies start - d + (j mod d): the pattern in front of the token
    val[q] = off + r >= d ? off + r - d : BIG_LIT;               // (in front of the stream start: E2 reads 0x00)
}

// ===============================================================================================================
// Element streams: LZ4 blocks (Formats/Common/LZ4.cs:176-200) and raw Snappy (Formats/Common/Snappy.cs:205-250).  One byte stream of
// variable-size elements; the size of "the element that would start at byte p" depends on the bytes behind p alone, so the same list
// ranking finds the real element starts.  An LZ4 sequence yields two tokens (literal run, match), a Snappy element one; a literal run's
// descriptor is its position in the input.  Neither body carries a size the host could read: LZ4 decodes its whole input, Snappy states
// its size in a varint at the front -- the device finds it (ctl[C_SIZE]), the launch is sized by the room in the destination.
// Elements with more than BIG_EXT length-extension bytes (runs / matches beyond ~512 KiB) and Snappy's 4-byte-offset copies beyond the
// window (E3) are left to the exact kernel (the gate): the speculation runs for EVERY input byte and must be bounded (an input of n bytes
// 0xFF costs n x BIG_EXT loop trips: 2 G for a megabyte, a few hundred microseconds of the whole GPU).
#define BIG_EXT 2048u

struct BigElem { u32 size; bool bad; u32 run_pos, run_len, mlen, dist; bool has_match; };

// ReadExtension (LZ4.cs:241-252): bytes are added until one is not 0xFF -- eight at a time (a chain of a thousand 0xFF bytes, one dependent
// load each, was 0.4 ms of latency for ONE lane: the Q15 encoding of Test.bmp has such matches).  False: the chain runs past the input
// or past BIG_EXT bytes.
__device__ __forceinline__ bool big_lz4_ext(const u8* src, u32 n, u32& sp, u32& len) {
    for (u32 k = 0; k < BIG_EXT; k += 8u) {
        if (sp + 8u <= n) {
            u64 v; __builtin_memcpy(&v, src + sp, 8);
            const u64 z = ~v;                                     // a byte that is not 0xFF leaves a non-zero byte here
            if (z == 0ull) { len += 8u * 255u; sp += 8u; continue; }
            const u32 i = (u32)__builtin_ctzll(z) >> 3;           // the first such byte
            len += 255u * i + (u32)((v >> (8u * i)) & 0xFFu); sp += i + 1u;
            return true;
        }
        for (u32 i = 0; i < 8u; i++) { if (sp >= n) return false; const u32 b = src[sp++]; len += b; if (b != 255u) return true; }
    }
    return false;
}
// LZ4 sequence at p (p < n)
__device__ __forceinline__ BigElem big_lz4_elem(const u8* src, u32 n, u32 p) {
    BigElem e; e.bad = false; e.has_match = false; e.run_len = 0; e.mlen = 0; e.dist = 0; e.run_pos = 0;
    u32 sp = p;
    const u32 tok = src[sp++];
    u32 L = tok >> 4;
    if (L == 15u && !big_lz4_ext(src, n, sp, L)) { e.bad = true; e.size = n - p; return e; }
    if (L > n - sp) { e.bad = true; e.size = n - p; return e; }   // Slice throws  LZ4.cs:187
    e.run_pos = sp; e.run_len = L;
    sp += L;
    if (sp >= n) { e.size = sp - p; return e; }                   // the last sequence has literals only  :190
    if (sp + 2u > n) { e.bad = true; e.size = n - p; return e; }
    e.dist = (u32)src[sp] | ((u32)src[sp + 1] << 8); sp += 2;     // :195
    u32 M = tok & 15u;
    if (M == 15u && !big_lz4_ext(src, n, sp, M)) { e.bad = true; e.size = n - p; return e; }
    e.mlen = M + 4u; e.has_match = true;                          // :198
    if (e.dist == 0u) e.dist = 65536u;                            // E1
    e.size = sp - p;
    return e;
}
// Snappy element at p; p == 0 is the varint in front (size = its length, run_len = the declared size)
__device__ __forceinline__ BigElem big_snappy_elem(const u8* src, u32 n, u32 p) {
    BigElem e; e.bad = false; e.has_match = false; e.run_len = 0; e.mlen = 0; e.dist = 0; e.run_pos = 0;
    if (p == 0u) {                                                // ReadDecompressedSize  Snappy.cs:109-122
        u32 result = 0, sh = 0, i = 0, b;
        do { if (i >= n || i >= 5u) { e.bad = true; e.size = n; return e; } b = src[i++]; result |= (b & 0x7Fu) << (sh & 31u); sh += 7u; } while (b & 0x80u);
        e.size = i; e.run_len = result;
        return e;
    }
    const u32 tag = src[p], type = tag & 3u, hi = tag >> 2;
    if (type == 0u) {
        u32 hdr = 1, len = hi + 1u;
        if (hi >= 60u) {
            const u32 nb = hi - 59u;
            if (p + 1u + nb > n) { e.bad = true; e.size = n - p; return e; }
            u32 v = 0; for (u32 i = 0; i < nb; i++) v |= (u32)src[p + 1u + i] << (8u * i);
            if (v >= 0x1FFFFFFFu) { e.bad = true; e.size = n - p; return e; }
            len = v + 1u; hdr = 1u + nb;
        }
        if (len > n - p - hdr) { e.bad = true; e.size = n - p; return e; }
        e.run_pos = p + hdr; e.run_len = len; e.size = hdr + len;
        return e;
    }
    const u32 need = type == 1u ? 2u : (type == 2u ? 3u : 5u);
    if (p + need > n) { e.bad = true; e.size = n - p; return e; }
    e.has_match = true;
    if (type == 1u) { e.mlen = (hi & 7u) + 4u; e.dist = ((tag >> 5) << 8) | src[p + 1]; }
    else if (type == 2u) { e.mlen = hi + 1u; e.dist = (u32)src[p + 1] | ((u32)src[p + 2] << 8); }
    else { e.mlen = hi + 1u; e.dist = (u32)src[p + 1] | ((u32)src[p + 2] << 8) | ((u32)src[p + 3] << 16) | ((u32)src[p + 4] << 24); if (e.dist > 65536u) e.bad = true; }   // E3: BAD_TOKEN in the exact kernel
    if (e.dist == 0u) e.dist = 65536u;                            // E1
    e.size = need;
    return e;
}
template <bool LZ4>
__global__ __launch_bounds__(256) void big_elem_sizes(const u8* __restrict__ src, u32 src_len, u32* __restrict__ next) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p > src_len) return;
    if (p == src_len) { next[p] = src_len; return; }
    const BigElem e = LZ4 ? big_lz4_elem(src, src_len, p) : big_snappy_elem(src, src_len, p);
    const u64 nx = (u64)p + (e.size ? e.size : 1u);
    next[p] = nx > src_len ? src_len : (u32)nx;
}
// the tokens of every real element: LZ4 two (run, match), Snappy one
template <bool LZ4>
__global__ __launch_bounds__(256) void big_elem_tokens(const u8* __restrict__ src, u32 src_len, const u32* __restrict__ gpos, u32* __restrict__ ctl,
                                                       u32* __restrict__ tlen, u32* __restrict__ tdesc, u32* __restrict__ tend) {
    const u32 ng = ctl[C_NG];
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    constexpr u32 TPE = LZ4 ? 2u : 1u;
    if (g == 0) ctl[C_NT] = TPE * ng;
    if (g >= ng) return;
    const u32 p = gpos[g];
    const BigElem e = LZ4 ? big_lz4_elem(src, src_len, p) : big_snappy_elem(src, src_len, p);
    const u64 endp = (u64)p + e.size;
    const u32 te = endp > src_len ? src_len : (u32)endp;
    if (LZ4) {
        if (e.bad) { tlen[2u * g] = 1u; tdesc[2u * g] = BIG_LIT | BIG_OOB; tlen[2u * g + 1u] = 0u; tdesc[2u * g + 1u] = BIG_LIT; }
        else {
            tlen[2u * g] = e.run_len; tdesc[2u * g] = BIG_LIT | BIG_RUN | e.run_pos;
            tlen[2u * g + 1u] = e.has_match ? e.mlen : 0u; tdesc[2u * g + 1u] = e.has_match ? e.dist : BIG_LIT;
        }
        tend[2u * g] = te; tend[2u * g + 1u] = te;
    } else {
        if (p == 0u) { if (e.bad) ctl[C_BAD] = 1u; ctl[C_SIZE] = e.run_len; tlen[g] = 0u; tdesc[g] = BIG_LIT; }   // the varint: no output, the declared size
        else if (e.bad) { tlen[g] = 1u; tdesc[g] = BIG_LIT | BIG_OOB; }
        else if (e.has_match) { tlen[g] = e.mlen; tdesc[g] = e.dist; }
        else { tlen[g] = e.run_len; tdesc[g] = BIG_LIT | BIG_RUN | e.run_pos; }
        tend[g] = te;
    }
}
// LZ4: the output size is what the tokens add up to (ctl[C_TOTAL], saturated) -- and the stream is over with its input
__global__ void big_lz4_size(u32* __restrict__ ctl, u32 src_len) { ctl[C_SIZE] = ctl[C_TOTAL]; ctl[C_USED] = src_len; }

// ===============================================================================================================
// PRS (Sega/PRS.cs:59-102): control bits and data bytes interleave, and a flag byte is fetched when a bit is needed -- but every flag byte
// is followed by the data of exactly the tokens whose LAST control bit lies in it (a group), and which tokens those are depends on the flag
// byte and on how much of a token the flag byte before it left unfinished: five entry states (csrc/alz_prs_table.h, the table the wavefront
// kernels walk with).  So the list to rank has one node per (input byte, entry state): next(p, s) = (p + size of the group, exit state),
// where the only DATA the size depends on is, per long match of the group, whether the low three bits of its word are zero (a third byte
// follows).  The stream has no size: it ends at the first long match whose word is zero (:78-79); the tokens in front of it add up to
// the output size.
__device__ const AlzPrsTable big_prs_table = alz_make_prs_table();
__device__ __forceinline__ u32 big_rev8(u32 b) { return __builtin_bitreverse32(b) >> 24; }

struct BigPrsGroup { u32 size, exit_state, ntok; bool oob; u32 w0, w1; };
// the group whose flag byte is at p, entered in state st; thirds = bit k set: long match k has a third byte
template <bool BIG>
__device__ __forceinline__ BigPrsGroup big_prs_group(const u8* src, u32 n, u32 p, u32 st, u32& thirds) {
    BigPrsGroup g; g.oob = false; thirds = 0;
    const u32 f = BIG ? big_rev8(src[p]) : (u32)src[p];
    g.w0 = big_prs_table.w[2u * (st * 256u + f)]; g.w1 = big_prs_table.w[2u * (st * 256u + f) + 1u];
    g.exit_state = (g.w0 >> 11) & 7u; g.ntok = ((g.w0 >> 7) & 7u) + 1u;
    u32 size = g.w0 & 15u;                                        // flag byte + data bytes without the third bytes
    const u32 nlong = ((g.w0 >> 4) & 1u) + ((g.w0 >> 5) & 1u) + ((g.w0 >> 6) & 1u) + ((g.w0 >> 10) & 1u);
    u32 extra = 0;
    for (u32 k = 0; k < nlong; k++) {
        const u64 at = (u64)p + ((g.w0 >> (16u + 4u * k)) & 15u) + extra;   // first data byte of long match k
        if (at + 2u > n) { g.oob = true; break; }
        const u32 lowbyte = BIG ? src[at + 1u] : src[at];        // the byte that holds the low three bits of the word
        if ((lowbyte & 7u) == 0u) {
            const u32 other = BIG ? src[at] : src[at + 1u];
            if ((lowbyte | other) != 0u) { thirds |= 1u << k; extra++; }   // (the zero word is the terminator: no third byte)
        }
    }
    g.size = size + extra;
    return g;
}
template <bool BIG>
__global__ __launch_bounds__(256) void big_prs_sizes(const u8* __restrict__ src, u32 src_len, u32* __restrict__ next) {
    const u32 id = blockIdx.x * 256u + threadIdx.x;               // node = 5 p + state
    const u32 p = id / ALZ_PRS_STATES, st = id - p * ALZ_PRS_STATES;
    if (p > src_len) return;
    if (p == src_len) { next[id] = id; return; }
    u32 thirds;
    const BigPrsGroup g = big_prs_group<BIG>(src, src_len, p, st, thirds);
    const u64 np = (u64)p + g.size;
    next[id] = (np >= src_len || g.oob) ? src_len * ALZ_PRS_STATES : (u32)np * ALZ_PRS_STATES + g.exit_state;
}
// up to eight tokens per group (unused slots: length 0)
template <bool BIG>
__global__ __launch_bounds__(256) void big_prs_tokens(const u8* __restrict__ src, u32 src_len, const u32* __restrict__ gpos, u32* __restrict__ ctl,
                                                      u32* __restrict__ tlen, u32* __restrict__ tdesc, u32* __restrict__ tend) {
    const u32 ng = ctl[C_NG];
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g == 0) ctl[C_NT] = 8u * ng;
    if (g >= ng) return;
    const u32 id = gpos[g], p = id / ALZ_PRS_STATES, st = id - p * ALZ_PRS_STATES;
    u32 thirds;
    const BigPrsGroup gr = big_prs_group<BIG>(src, src_len, p, st, thirds);
    u64 pos = (u64)p + 1u;
    u32 nl = 0;
    for (u32 k = 0; k < 8u; k++) {
        const u32 i = 8u * g + k;
        u32 len = 0, desc = BIG_LIT;
        if (k < gr.ntok) {
            const u32 code = (gr.w1 >> (3u * k)) & 7u;
            if (code == 0u) {                                     // literal  PRS.cs:66-70
                if (pos + 1u > src_len) { len = 1; desc = BIG_LIT | BIG_OOB; } else { len = 1; desc = BIG_LIT | src[pos]; }
                pos += 1;
            } else if (code == 1u) {                              // long match  :73-90
                const bool third = (thirds >> nl) & 1u; nl++;
                const u32 need = third ? 3u : 2u;
                if (pos + need > src_len) { len = 1; desc = BIG_LIT | BIG_OOB; }
                else {
                    const u32 x0 = src[pos], x1 = src[pos + 1];
                    const u32 v = BIG ? ((x0 << 8) | x1) : ((x1 << 8) | x0);
                    if (v == 0u) { len = 0; desc = BIG_LIT; atomicMin(ctl + C_TERM, i); }   // the terminator  :78-79
                    else { desc = 0x2000u - (v >> 3); len = (v & 7u) ? (v & 7u) + 2u : (u32)src[pos + 2] + 1u; }
                }
                pos += need;
            } else {                                              // short match  :91-96
                if (pos + 1u > src_len) { len = 1; desc = BIG_LIT | BIG_OOB; } else { len = (code & 3u) + 2u; desc = 0x100u - src[pos]; }
                pos += 1;
            }
        }
        tlen[i] = len; tdesc[i] = desc; tend[i] = pos > src_len ? src_len : (u32)pos;
    }
}
// the output ends in front of the first terminator; source.Position is just behind it
__global__ void big_prs_size(u32* __restrict__ ctl, const u32* __restrict__ toff, const u32* __restrict__ tend) {
    const u32 t = ctl[C_TERM];
    if (t == 0xFFFFFFFFu || t >= ctl[C_NT]) { ctl[C_BAD] = 1u; ctl[C_SIZE] = 0u; return; }   // no terminator in the input: EndOfStreamException  :101
    ctl[C_SIZE] = toff[t]; ctl[C_USED] = tend[t];
}

// ===============================================================================================================
// LZO (Formats/Common/LZO.cs:49-139): what an instruction byte means depends on its first byte and, for the opcodes below 16, on how many
// literals the instruction before it copied ("plain": 0, 1-3, or 4 after a literal-run instruction) -- three entry states, one list node
// per (input byte, state).  An instruction yields up to two tokens (a match or a literal run, then 0-3 trailing literals); the stream
// ends at the end marker (a code-1 instruction with distance 16 384, :107-109), which plays the part of PRS's terminator.  The first
// byte of the stream is special (:59-64: above 17 it is a literal run, and the instruction behind it is entered with plain = 0 -- the
// managed decoder's quirk, kept).
#define BIG_LZO_STATES 3u
struct BigLzo { u32 size, next_state; bool bad, term; u32 len0, desc0, len1, desc1; };
// ReadExtendedInt (LZO.cs:252-262): zero bytes add 255 each until a non-zero byte -- eight at a time, as big_lz4_ext
__device__ __forceinline__ bool big_lzo_ext(const u8* src, u32 n, u32& sp, u32& len) {
    for (u32 k = 0; k < BIG_EXT; k += 8u) {
        if (sp + 8u <= n) {
            u64 v; __builtin_memcpy(&v, src + sp, 8);
            if (v == 0ull) { len += 8u * 255u; sp += 8u; continue; }
            const u32 i = (u32)__builtin_ctzll(v) >> 3;
            len += 255u * i + (u32)((v >> (8u * i)) & 0xFFu); sp += i + 1u;
            return true;
        }
        for (u32 i = 0; i < 8u; i++) { if (sp >= n) return false; const u32 b = src[sp++]; if (b) { len += b; return true; } len += 255u; }
    }
    return false;
}
__device__ __forceinline__ BigLzo big_lzo_instr(const u8* src, u32 n, u32 p, u32 st) {
    BigLzo r; r.bad = false; r.term = false; r.len0 = r.len1 = 0; r.desc0 = r.desc1 = BIG_LIT; r.next_state = 0;
    u32 sp = p;
    u32 flag = src[sp++];
    if (p == 0u && flag > 17u) {                                  // :59-64
        const u32 len = flag - 17u;
        if (len > n - sp) { r.bad = true; r.size = n - p; return r; }
        r.len0 = len; r.desc0 = BIG_LIT | BIG_RUN | sp; r.size = 1u + len; r.next_state = 0;
        return r;
    }
    const u32 code = flag >> 4;
    u32 length, distance;
    if (code == 0u) {
        if (st == 0u) {                                           // a literal run  :72-82
            length = 3u + flag;
            if (length == 3u) { length = 18u; if (!big_lzo_ext(src, n, sp, length)) { r.bad = true; r.size = n - p; return r; } }
            if (length > n - sp) { r.bad = true; r.size = n - p; return r; }
            r.len0 = length; r.desc0 = BIG_LIT | BIG_RUN | sp; r.size = sp - p + length; r.next_state = 2;   // plain = 4
            return r;
        }
        if (sp >= n) { r.bad = true; r.size = n - p; return r; }
        const u32 d = src[sp++];
        if (st == 1u) { distance = (d << 2) + (flag >> 2) + 1u; length = 2u; }              // :83-88
        else { distance = (d << 2) + (flag >> 2) + 2049u; length = 3u; }                    // :89-94
    } else if (code == 1u) {                                      // :96-109
        length = 2u + (flag & 7u);
        if (length == 2u) { length = 9u; if (!big_lzo_ext(src, n, sp, length)) { r.bad = true; r.size = n - p; return r; } }
        distance = 16384u + ((flag & 8u) << 11);
        if (sp + 2u > n) { r.bad = true; r.size = n - p; return r; }
        flag = src[sp++]; const u32 hi = src[sp++];
        distance |= (hi << 6) | (flag >> 2);
        if (distance == 16384u) { r.term = true; r.size = sp - p; return r; }
    } else if (code <= 3u) {                                      // :110-119
        length = 2u + (flag & 0x1Fu);
        if (length == 2u) { length = 33u; if (!big_lzo_ext(src, n, sp, length)) { r.bad = true; r.size = n - p; return r; } }
        if (sp + 2u > n) { r.bad = true; r.size = n - p; return r; }
        flag = src[sp++]; const u32 hi = src[sp++];
        distance = ((hi << 6) | (flag >> 2)) + 1u;
    } else if (code <= 7u) {                                      // :120-125
        length = 3u + ((flag >> 5) & 1u);
        if (sp >= n) { r.bad = true; r.size = n - p; return r; }
        const u32 d = src[sp++];
        distance = (d << 3) + ((flag >> 2) & 7u) + 1u;
    } else {                                                      // :126-131
        length = 5u + ((flag >> 5) & 3u);
        if (sp >= n) { r.bad = true; r.size = n - p; return r; }
        const u32 d = src[sp++];
        distance = (d << 3) + ((flag & 0x1Cu) >> 2) + 1u;
    }
    const u32 plain = flag & 3u;                                  // :132
    if (plain > n - sp) { r.bad = true; r.size = n - p; return r; }
    r.len0 = length; r.desc0 = distance;
    r.len1 = plain; r.desc1 = BIG_LIT | BIG_RUN | sp;
    r.size = sp - p + plain; r.next_state = plain ? 1u : 0u;
    return r;
}
__global__ __launch_bounds__(256) void big_lzo_sizes(const u8* __restrict__ src, u32 src_len, u32* __restrict__ next) {
    const u32 id = blockIdx.x * 256u + threadIdx.x;               // node = 3 p + state
    const u32 p = id / BIG_LZO_STATES, st = id - p * BIG_LZO_STATES;
    if (p > src_len) return;
    if (p == src_len) { next[id] = id; return; }
    const BigLzo r = big_lzo_instr(src, src_len, p, st);
    const u64 np = (u64)p + (r.size ? r.size : 1u);
    next[id] = (np >= src_len || r.bad || r.term) ? src_len * BIG_LZO_STATES : (u32)np * BIG_LZO_STATES + r.next_state;
}
__global__ __launch_bounds__(256) void big_lzo_tokens(const u8* __restrict__ src, u32 src_len, const u32* __restrict__ gpos, u32* __restrict__ ctl,
                                                      u32* __restrict__ tlen, u32* __restrict__ tdesc, u32* __restrict__ tend) {
    const u32 ng = ctl[C_NG];
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g == 0) ctl[C_NT] = 2u * ng;
    if (g >= ng) return;
    const u32 id = gpos[g], p = id / BIG_LZO_STATES, st = id - p * BIG_LZO_STATES;
    const BigLzo r = big_lzo_instr(src, src_len, p, st);
    const u64 endp = (u64)p + r.size;
    const u32 te = endp > src_len ? src_len : (u32)endp;
    if (r.bad) { tlen[2u * g] = 1u; tdesc[2u * g] = BIG_LIT | BIG_OOB; tlen[2u * g + 1u] = 0u; tdesc[2u * g + 1u] = BIG_LIT; }
    else {
        if (r.term) atomicMin(ctl + C_TERM, 2u * g);
        tlen[2u * g] = r.len0; tdesc[2u * g] = r.desc0; tlen[2u * g + 1u] = r.len1; tdesc[2u * g + 1u] = r.desc1;
    }
    tend[2u * g] = te; tend[2u * g + 1u] = te;
}

// ---------------------------------------------------------------------------------------------------------------- host side
static u32 big_ntok(const alz_stream& st) {
    const u64 by_flags = 8ull * st.src_len;
    return (u32)(by_flags < st.decom_len ? by_flags : st.decom_len);
}
static u32 big_rounds(u32 n) { u32 r = 1; while ((1ull << r) < n) r++; return r + 1u; }
static void big_rank(hipStream_t stream, const u32* next1, u32* jump_a, u32* jump_b, u8* mark, u32 nodes) {
    const u32 rtiles = (nodes + BIG_RTILE - 1u) / BIG_RTILE, nbn = (nodes + 255u) / 256u;
    hipLaunchKernelGGL(big_tile_exit, dim3(rtiles), dim3(1024), 0, stream, next1, jump_a, nodes);
    const u32 rr = (big_rounds(rtiles + 1u) + BIG_LOG - 1u) / BIG_LOG;   // (a launch is BIG_LOG rounds)
    for (u32 r = 0; r < rr; r++) { hipLaunchKernelGGL(big_rank_round, dim3(nbn), dim3(256), 0, stream, jump_a, jump_b, mark, nodes); u32* t = jump_a; jump_a = jump_b; jump_b = t; }
    hipLaunchKernelGGL(big_tile_mark, dim3(rtiles), dim3(1024), 0, stream, next1, mark, nodes);
}
static bool big_three(int fmt) { return fmt == ALZ_FMT_YAY0 || fmt == ALZ_FMT_MIO0; }
static bool big_inter(int fmt) { return fmt == ALZ_FMT_LZSS || fmt == ALZ_FMT_LZ10 || fmt == ALZ_FMT_LZ11 || fmt == ALZ_FMT_YAZ0; }
static bool big_prs(int fmt) { return fmt == ALZ_FMT_PRS_BE || fmt == ALZ_FMT_PRS_LE; }
static bool big_elem(int fmt) { return fmt == ALZ_FMT_LZ4_BLOCK || fmt == ALZ_FMT_SNAPPY_RAW || fmt == ALZ_FMT_LZO || big_prs(fmt); }
static size_t big_al(size_t x) { return (x + 255) & ~(size_t)255; }

// layout of the scratch of the interleaved path
struct InterLayout {
    u32 nodes, max_ng, max_nt, mtiles, ttiles;
    size_t val, jump_a, jump_b, next1, mark, tile_c, tile_cb, gpos, tlen, tdesc, tend, toff, tile_l, tile_lb, ctl, total;
    InterLayout(const alz_stream& st, int fmt = ALZ_FMT_YAZ0) {
        nodes = st.src_len + 1u;
        if (big_prs(fmt)) { nodes = (st.src_len + 1u) * ALZ_PRS_STATES; max_ng = st.src_len / 2u + 2u; max_nt = 8u * max_ng; }   // (a group has a flag byte and at least one data byte)
        else if (fmt == ALZ_FMT_LZO) { nodes = (st.src_len + 1u) * BIG_LZO_STATES; max_ng = st.src_len / 2u + 2u; max_nt = 2u * max_ng; }   // (an instruction has at least two bytes, all but a run of one)
        else if (fmt == ALZ_FMT_LZ4_BLOCK) { max_ng = st.src_len / 2u + 2u; max_nt = 2u * max_ng; }        // (a sequence has at least a token and -- all but the last -- an offset)
        else if (fmt == ALZ_FMT_SNAPPY_RAW) { max_ng = st.src_len / 2u + 2u; max_nt = max_ng; }         // (an element has at least two bytes)
        else { max_ng = st.src_len / 9u + 2u; max_nt = 8u * max_ng; }
        const u32 out_bound = big_elem(fmt) ? st.dst_cap : st.decom_len;
        mtiles = (nodes + 1023u) / 1024u; ttiles = (max_nt + 1023u) / 1024u;
        size_t o = 0;
        val = o; o += big_al((size_t)out_bound * 4);
        jump_a = o; o += big_al((size_t)nodes * 4); jump_b = o; o += big_al((size_t)nodes * 4); next1 = o; o += big_al((size_t)nodes * 4);
        mark = o; o += big_al((size_t)nodes + 64);
        tile_c = o; o += big_al((size_t)(mtiles + 64) * 4); tile_cb = o; o += big_al((size_t)(mtiles + 64) * 4);
        gpos = o; o += big_al((size_t)max_ng * 4);
        tlen = o; o += big_al((size_t)max_nt * 4); tdesc = o; o += big_al((size_t)max_nt * 4);
        tend = o; o += big_al((size_t)max_nt * 4); toff = o; o += big_al((size_t)max_nt * 4);
        tile_l = o; o += big_al((size_t)(ttiles + 64) * 4); tile_lb = o; o += big_al((size_t)(ttiles + 64) * 4);
        ctl = o; o += big_al((C_FLAGS + 40) * 4);
        total = o;
    }
};

bool alz_big_eligible(int fmt, const alz_stream* st, const alz_lz_properties* lz, uint32_t min_bytes) {
    if (!big_three(fmt) && !big_inter(fmt) && !big_elem(fmt)) return false;
    if (big_elem(fmt)) {                                          // no size in the descriptor: what the destination holds, and an input worth the launches
        if (min_bytes == 0xFFFFFFFFu || st->dst_cap < min_bytes || st->dst_cap > 0x40000000u || st->src_len < 8192u || st->src_len > 0x10000000u) return false;
        if ((uint64_t)st->dst_cap > 32ull * st->src_len + 65536ull) return false;   // the launches are sized by the room in the destination: not for a destination far beyond what the input can plausibly yield
        if (fmt == ALZ_FMT_LZ4_BLOCK && st->aux0 != 0u) return false;   // a block of a linked frame continues the window of its predecessors
        return true;
    }
    if (st->decom_len < min_bytes || st->decom_len > 0x40000000u) return false;
    if (st->dst_cap < st->decom_len || st->src_len == 0 || st->src_len > 0x40000000u) return false;
    if (big_three(fmt) && (st->aux0 > st->src_len || st->aux1 > st->src_len)) return false;
    if (fmt == ALZ_FMT_LZSS && (lz->window_bits < 8 || lz->window_bits > 16 || lz->length_bits < 1 || lz->length_bits > 8 || lz->max_distance != (1u << lz->window_bits))) return false;
    return true;
}

size_t alz_big_scratch_bytes(int fmt, const alz_stream* st) {
    if (big_inter(fmt) || big_elem(fmt)) return InterLayout(*st, fmt).total + 256;
    const u32 ntok = big_ntok(*st);
    const size_t ntiles = (ntok + BIG_TILE - 1) / BIG_TILE;
    return big_al((size_t)st->decom_len * 4) + 6 * big_al((ntiles + 64) * 4) + 4 * big_al((size_t)ntok * 4) + big_al((C_FLAGS + 40) * 4) + 256;
}

template <int FMT>
static hipError_t launch_inter(hipStream_t stream, const u8* src, u8* dst, const alz_stream* st, const BigGeom& gm, alz_result* d_result, u8* base, u32* d_gate, u32* d_acc) {
    const InterLayout L(*st);
    u32* val = (u32*)(base + L.val); u32* jump_a = (u32*)(base + L.jump_a); u32* jump_b = (u32*)(base + L.jump_b); u8* mark = base + L.mark;
    u32* tile_c = (u32*)(base + L.tile_c); u32* tile_cb = (u32*)(base + L.tile_cb); u32* gpos = (u32*)(base + L.gpos);
    u32* tlen = (u32*)(base + L.tlen); u32* tdesc = (u32*)(base + L.tdesc); u32* tend = (u32*)(base + L.tend); u32* toff = (u32*)(base + L.toff);
    u32* tile_l = (u32*)(base + L.tile_l); u32* tile_lb = (u32*)(base + L.tile_lb); u32* ctl = (u32*)(base + L.ctl);
    hipLaunchKernelGGL(big_init, dim3((L.nodes + 64u + 255u) / 256u), dim3(256), 0, stream, ctl, 0u, 0u, mark, L.nodes + 64u);   // (the first group starts at byte 0)
    const u32 nbn = (L.nodes + 255u) / 256u;
    u32* next1 = (u32*)(base + L.next1);
    hipLaunchKernelGGL((big_group_sizes<FMT>), dim3(nbn), dim3(256), 0, stream, src, st->src_len, next1);
    big_rank(stream, next1, jump_a, jump_b, mark, L.nodes);
    hipLaunchKernelGGL(big_mark_count, dim3(L.mtiles), dim3(64), 0, stream, mark, st->src_len, tile_c);
    hipLaunchKernelGGL(big_scan, dim3(1), dim3(1024), 0, stream, tile_c, tile_cb, L.mtiles, ctl + C_NG);
    hipLaunchKernelGGL(big_mark_scatter, dim3(L.mtiles), dim3(64), 0, stream, mark, st->src_len, tile_cb, gpos);
    hipLaunchKernelGGL((big_group_tokens<FMT>), dim3((L.max_ng + 255u) / 256u), dim3(256), 0, stream, src, st->src_len, gm, gpos, ctl, tlen, tdesc, tend);
    hipLaunchKernelGGL(big_len_count, dim3(L.ttiles), dim3(64), 0, stream, tlen, ctl, tile_l);
    hipLaunchKernelGGL(big_scan, dim3(1), dim3(1024), 0, stream, tile_l, tile_lb, L.ttiles, ctl + C_TOTAL);
    hipLaunchKernelGGL(big_len_offsets, dim3(L.ttiles), dim3(64), 0, stream, tlen, ctl, tile_lb, toff);
    const u32 nb = (st->decom_len + 255u) / 256u;
    hipLaunchKernelGGL((big_emit_bytes<FMT == ALZ_FMT_LZSS>), dim3(nb), dim3(256), 0, stream, st->decom_len, gm, toff, tlen, tdesc, tend, val, ctl);
    hipLaunchKernelGGL(big_jump_tile, dim3((st->decom_len + 1023u) / 1024u), dim3(1024), 0, stream, val, st->decom_len, (const u32*)nullptr);
    const u32 rounds = (big_rounds((st->decom_len + 1023u) / 1024u + 1u) + BIG_LOG - 1u) / BIG_LOG;   // (a launch of big_jump is BIG_LOG rounds)
    for (u32 r = 0; r < rounds; r++)
        hipLaunchKernelGGL(big_jump, dim3(nb), dim3(256), 0, stream, val, st->decom_len, (const u32*)nullptr, ctl + C_FLAGS + r, ctl + C_FLAGS + r + 1);
    BigArgs a; a.src = src; a.dst = dst; a.src_len = st->src_len; a.size = st->decom_len; a.aux0 = a.aux1 = 0; a.ntok = 0; a.ntiles = 0;
    hipLaunchKernelGGL((big_write<false>), dim3(nb), dim3(256), 0, stream, a, val, ctl, d_result, d_gate, d_acc);
    return hipGetLastError();
}

template <bool LZ4>
static hipError_t launch_elem(hipStream_t stream, const u8* src, u8* dst, const alz_stream* st, alz_result* d_result, u8* base, u32* d_gate, u32* d_acc) {
    const InterLayout L(*st, LZ4 ? ALZ_FMT_LZ4_BLOCK : ALZ_FMT_SNAPPY_RAW);
    u32* val = (u32*)(base + L.val); u32* jump_a = (u32*)(base + L.jump_a); u32* jump_b = (u32*)(base + L.jump_b); u8* mark = base + L.mark;
    u32* tile_c = (u32*)(base + L.tile_c); u32* tile_cb = (u32*)(base + L.tile_cb); u32* gpos = (u32*)(base + L.gpos);
    u32* tlen = (u32*)(base + L.tlen); u32* tdesc = (u32*)(base + L.tdesc); u32* tend = (u32*)(base + L.tend); u32* toff = (u32*)(base + L.toff);
    u32* tile_l = (u32*)(base + L.tile_l); u32* tile_lb = (u32*)(base + L.tile_lb); u32* ctl = (u32*)(base + L.ctl);
    hipLaunchKernelGGL(big_init, dim3((L.nodes + 64u + 255u) / 256u), dim3(256), 0, stream, ctl, 0u, 0u, mark, L.nodes + 64u);   // (the first group starts at byte 0)
    const u32 nbn = (L.nodes + 255u) / 256u;
    u32* next1 = (u32*)(base + L.next1);
    hipLaunchKernelGGL((big_elem_sizes<LZ4>), dim3(nbn), dim3(256), 0, stream, src, st->src_len, next1);
    big_rank(stream, next1, jump_a, jump_b, mark, L.nodes);
    hipLaunchKernelGGL(big_mark_count, dim3(L.mtiles), dim3(64), 0, stream, mark, st->src_len, tile_c);
    hipLaunchKernelGGL(big_scan, dim3(1), dim3(1024), 0, stream, tile_c, tile_cb, L.mtiles, ctl + C_NG);
    hipLaunchKernelGGL(big_mark_scatter, dim3(L.mtiles), dim3(64), 0, stream, mark, st->src_len, tile_cb, gpos);
    hipLaunchKernelGGL((big_elem_tokens<LZ4>), dim3((L.max_ng + 255u) / 256u), dim3(256), 0, stream, src, st->src_len, gpos, ctl, tlen, tdesc, tend);
    hipLaunchKernelGGL(big_len_count, dim3(L.ttiles), dim3(64), 0, stream, tlen, ctl, tile_l);
    hipLaunchKernelGGL(big_scan, dim3(1), dim3(1024), 0, stream, tile_l, tile_lb, L.ttiles, ctl + C_TOTAL);
    hipLaunchKernelGGL(big_len_offsets, dim3(L.ttiles), dim3(64), 0, stream, tlen, ctl, tile_lb, toff);
    if (LZ4) hipLaunchKernelGGL(big_lz4_size, dim3(1), dim3(1), 0, stream, ctl, st->src_len);
    const u32 nb = (st->dst_cap + 255u) / 256u;
    BigGeom gm; gm.length_bits = gm.min_length = gm.windows_start = gm.max_distance = gm.W = 0;
    // (LZ4 decodes its WHOLE input -- sequences that add no output may follow the last output byte, e.g. a lone zero token --, so
    // source.Position is the end of the input and not the end of the last token with output: found by tools/soak.sh, seed 9488)
    hipLaunchKernelGGL((big_emit_bytes<false, true>), dim3(nb), dim3(256), 0, stream, st->dst_cap, gm, toff, tlen, tdesc, tend, val, ctl, src, LZ4);
    hipLaunchKernelGGL(big_jump_tile, dim3((st->dst_cap + 1023u) / 1024u), dim3(1024), 0, stream, val, st->dst_cap, (const u32*)(ctl + C_SIZE));
    const u32 rounds = (big_rounds((st->dst_cap + 1023u) / 1024u + 1u) + BIG_LOG - 1u) / BIG_LOG;   // (a launch of big_jump is BIG_LOG rounds)
    for (u32 r = 0; r < rounds; r++)
        hipLaunchKernelGGL(big_jump, dim3(nb), dim3(256), 0, stream, val, st->dst_cap, (const u32*)(ctl + C_SIZE), ctl + C_FLAGS + r, ctl + C_FLAGS + r + 1);
    BigArgs a; a.src = src; a.dst = dst; a.src_len = st->src_len; a.size = st->dst_cap; a.aux0 = a.aux1 = 0; a.ntok = 0; a.ntiles = 0;
    hipLaunchKernelGGL((big_write<true>), dim3(nb), dim3(256), 0, stream, a, val, ctl, d_result, d_gate, d_acc);
    return hipGetLastError();
}

// KIND 0 / 1: PRS little / big endian; 2: LZO -- the formats whose stream ends at a terminator token
template <int KIND>
static hipError_t launch_term(hipStream_t stream, const u8* src, u8* dst, const alz_stream* st, alz_result* d_result, u8* base, u32* d_gate, u32* d_acc) {
    constexpr bool BIG = KIND == 1;
    constexpr u32 NST = KIND == 2 ? BIG_LZO_STATES : ALZ_PRS_STATES;
    const InterLayout L(*st, KIND == 2 ? ALZ_FMT_LZO : (BIG ? ALZ_FMT_PRS_BE : ALZ_FMT_PRS_LE));
    u32* val = (u32*)(base + L.val); u32* jump_a = (u32*)(base + L.jump_a); u32* jump_b = (u32*)(base + L.jump_b); u8* mark = base + L.mark;
    u32* tile_c = (u32*)(base + L.tile_c); u32* tile_cb = (u32*)(base + L.tile_cb); u32* gpos = (u32*)(base + L.gpos);
    u32* tlen = (u32*)(base + L.tlen); u32* tdesc = (u32*)(base + L.tdesc); u32* tend = (u32*)(base + L.tend); u32* toff = (u32*)(base + L.toff);
    u32* tile_l = (u32*)(base + L.tile_l); u32* tile_lb = (u32*)(base + L.tile_lb); u32* ctl = (u32*)(base + L.ctl);
    hipLaunchKernelGGL(big_init, dim3((L.nodes + 64u + 255u) / 256u), dim3(256), 0, stream, ctl, 0u, 0xFFFFFFFFu, mark, L.nodes + 64u);   // (node (byte 0, nothing pending) starts the chain)
    const u32 nbn = (L.nodes + 255u) / 256u;
    const u32 real_nodes = L.nodes - NST;                         // the nodes of the bytes that exist (the end node is the first behind them)
    u32* next1 = (u32*)(base + L.next1);
    if (KIND == 2) hipLaunchKernelGGL(big_lzo_sizes, dim3(nbn), dim3(256), 0, stream, src, st->src_len, next1);
    else hipLaunchKernelGGL((big_prs_sizes<BIG>), dim3(nbn), dim3(256), 0, stream, src, st->src_len, next1);
    big_rank(stream, next1, jump_a, jump_b, mark, L.nodes);
    hipLaunchKernelGGL(big_mark_count, dim3(L.mtiles), dim3(64), 0, stream, mark, real_nodes, tile_c);
    hipLaunchKernelGGL(big_scan, dim3(1), dim3(1024), 0, stream, tile_c, tile_cb, L.mtiles, ctl + C_NG);
    hipLaunchKernelGGL(big_mark_scatter, dim3(L.mtiles), dim3(64), 0, stream, mark, real_nodes, tile_cb, gpos);
    if (KIND == 2) hipLaunchKernelGGL(big_lzo_tokens, dim3((L.max_ng + 255u) / 256u), dim3(256), 0, stream, src, st->src_len, gpos, ctl, tlen, tdesc, tend);
    else hipLaunchKernelGGL((big_prs_tokens<BIG>), dim3((L.max_ng + 255u) / 256u), dim3(256), 0, stream, src, st->src_len, gpos, ctl, tlen, tdesc, tend);
    hipLaunchKernelGGL(big_len_count, dim3(L.ttiles), dim3(64), 0, stream, tlen, ctl, tile_l);
    hipLaunchKernelGGL(big_scan, dim3(1), dim3(1024), 0, stream, tile_l, tile_lb, L.ttiles, ctl + C_TOTAL);
    hipLaunchKernelGGL(big_len_offsets, dim3(L.ttiles), dim3(64), 0, stream, tlen, ctl, tile_lb, toff);
    hipLaunchKernelGGL(big_prs_size, dim3(1), dim3(1), 0, stream, ctl, toff, tend);
    const u32 nb = (st->dst_cap + 255u) / 256u;
    BigGeom gm; gm.length_bits = gm.min_length = gm.windows_start = gm.max_distance = gm.W = 0;
    hipLaunchKernelGGL((big_emit_bytes<false, true>), dim3(nb), dim3(256), 0, stream, st->dst_cap, gm, toff, tlen, tdesc, tend, val, ctl, src, true);
    hipLaunchKernelGGL(big_jump_tile, dim3((st->dst_cap + 1023u) / 1024u), dim3(1024), 0, stream, val, st->dst_cap, (const u32*)(ctl + C_SIZE));
    const u32 rounds = (big_rounds((st->dst_cap + 1023u) / 1024u + 1u) + BIG_LOG - 1u) / BIG_LOG;   // (a launch of big_jump is BIG_LOG rounds)
    for (u32 r = 0; r < rounds; r++)
        hipLaunchKernelGGL(big_jump, dim3(nb), dim3(256), 0, stream, val, st->dst_cap, (const u32*)(ctl + C_SIZE), ctl + C_FLAGS + r, ctl + C_FLAGS + r + 1);
    BigArgs a; a.src = src; a.dst = dst; a.src_len = st->src_len; a.size = st->dst_cap; a.aux0 = a.aux1 = 0; a.ntok = 0; a.ntiles = 0;
    hipLaunchKernelGGL((big_write<true>), dim3(nb), dim3(256), 0, stream, a, val, ctl, d_result, d_gate, d_acc);
    return hipGetLastError();
}

hipError_t alz_launch_big(int fmt, hipStream_t stream, const void* d_src_base, void* d_dst_base, const alz_stream* st, const alz_lz_properties* lz,
                          alz_result* d_result, void* d_scratch, uint32_t* d_gate, uint32_t* d_acc) {
    const u8* src = (const u8*)d_src_base + st->src_off; u8* dst = (u8*)d_dst_base + st->dst_off;
    if (fmt == ALZ_FMT_PRS_BE) return launch_term<1>(stream, src, dst, st, d_result, (u8*)d_scratch, d_gate, d_acc);
    if (fmt == ALZ_FMT_PRS_LE) return launch_term<0>(stream, src, dst, st, d_result, (u8*)d_scratch, d_gate, d_acc);
    if (fmt == ALZ_FMT_LZO) return launch_term<2>(stream, src, dst, st, d_result, (u8*)d_scratch, d_gate, d_acc);
    if (fmt == ALZ_FMT_LZ4_BLOCK) return launch_elem<true>(stream, src, dst, st, d_result, (u8*)d_scratch, d_gate, d_acc);
    if (fmt == ALZ_FMT_SNAPPY_RAW) return launch_elem<false>(stream, src, dst, st, d_result, (u8*)d_scratch, d_gate, d_acc);
    if (big_inter(fmt)) {
        BigGeom gm; gm.length_bits = lz->length_bits; gm.min_length = lz->min_length; gm.windows_start = lz->windows_start;
        gm.max_distance = lz->max_distance; gm.W = 1u << lz->window_bits;
        switch (fmt) {
        case ALZ_FMT_LZSS: return launch_inter<ALZ_FMT_LZSS>(stream, src, dst, st, gm, d_result, (u8*)d_scratch, d_gate, d_acc);
        case ALZ_FMT_LZ10: return launch_inter<ALZ_FMT_LZ10>(stream, src, dst, st, gm, d_result, (u8*)d_scratch, d_gate, d_acc);
        case ALZ_FMT_LZ11: return launch_inter<ALZ_FMT_LZ11>(stream, src, dst, st, gm, d_result, (u8*)d_scratch, d_gate, d_acc);
        default: return launch_inter<ALZ_FMT_YAZ0>(stream, src, dst, st, gm, d_result, (u8*)d_scratch, d_gate, d_acc);
        }
    }
    BigArgs a;
    a.src = src; a.dst = dst;
    a.src_len = st->src_len; a.size = st->decom_len; a.aux0 = st->aux0; a.aux1 = st->aux1;
    a.ntok = big_ntok(*st);
    a.ntiles = (a.ntok + BIG_TILE - 1) / BIG_TILE;
    const size_t tl = big_al(((size_t)a.ntiles + 64) * 4), ta = big_al((size_t)a.ntok * 4);
    u8* p = (u8*)d_scratch;
    u32* val = (u32*)p; p += big_al((size_t)a.size * 4);
    u32* tile_m = (u32*)p; p += tl; u32* tile_mb = (u32*)p; p += tl;
    u32* tile_u = (u32*)p; p += tl; u32* tile_ub = (u32*)p; p += tl;
    u32* tile_l = (u32*)p; p += tl; u32* tile_ob = (u32*)p; p += tl;
    u32* toff = (u32*)p; p += ta; u32* tlen = (u32*)p; p += ta; u32* tdesc = (u32*)p; p += ta; u32* tend = (u32*)p; p += ta;
    u32* ctl = (u32*)p;
    const u32 rounds = (big_rounds((a.size + 1023u) / 1024u + 1u) + BIG_LOG - 1u) / BIG_LOG;   // (a launch of big_jump is BIG_LOG rounds)      // (behind big_jump_tile: see there)
    hipLaunchKernelGGL(big_init, dim3(1), dim3(256), 0, stream, ctl, a.ntok, 0u, (u8*)nullptr, 0u);   // (round 0 of the jumping always runs: its "previous flag" holds 1)
    const bool mio0 = fmt == ALZ_FMT_MIO0;
    hipLaunchKernelGGL(big_count_matches, dim3(a.ntiles), dim3(64), 0, stream, a, tile_m);
    hipLaunchKernelGGL(big_scan, dim3(1), dim3(1024), 0, stream, tile_m, tile_mb, a.ntiles, (u32*)nullptr);
    if (!mio0) {
        hipLaunchKernelGGL((big_tile_sum<false, 0>), dim3(a.ntiles), dim3(64), 0, stream, a, tile_mb, tile_ub, tile_u);
        hipLaunchKernelGGL(big_scan, dim3(1), dim3(1024), 0, stream, tile_u, tile_ub, a.ntiles, (u32*)nullptr);
        hipLaunchKernelGGL((big_tile_sum<false, 1>), dim3(a.ntiles), dim3(64), 0, stream, a, tile_mb, tile_ub, tile_l);
    } else {
        hipLaunchKernelGGL((big_tile_sum<true, 1>), dim3(a.ntiles), dim3(64), 0, stream, a, tile_mb, tile_ub, tile_l);
    }
    hipLaunchKernelGGL(big_scan, dim3(1), dim3(1024), 0, stream, tile_l, tile_ob, a.ntiles, ctl + C_TOTAL);
    if (mio0) hipLaunchKernelGGL((big_emit<true>), dim3(a.ntiles), dim3(64), 0, stream, a, tile_mb, tile_ub, tile_ob, toff, tlen, tdesc, tend);
    else hipLaunchKernelGGL((big_emit<false>), dim3(a.ntiles), dim3(64), 0, stream, a, tile_mb, tile_ub, tile_ob, toff, tlen, tdesc, tend);
    const u32 nb = (a.size + 255u) / 256u;
    BigGeom gm; gm.length_bits = gm.min_length = gm.windows_start = gm.max_distance = gm.W = 0;
    hipLaunchKernelGGL((big_emit_bytes<false>), dim3(nb), dim3(256), 0, stream, a.size, gm, toff, tlen, tdesc, tend, val, ctl);
    hipLaunchKernelGGL(big_jump_tile, dim3((a.size + 1023u) / 1024u), dim3(1024), 0, stream, val, a.size, (const u32*)nullptr);
    for (u32 r = 0; r < rounds; r++)
        hipLaunchKernelGGL(big_jump, dim3(nb), dim3(256), 0, stream, val, a.size, (const u32*)nullptr, ctl + C_FLAGS + r, ctl + C_FLAGS + r + 1);
    hipLaunchKernelGGL((big_write<false>), dim3(nb), dim3(256), 0, stream, a, val, ctl, d_result, d_gate, d_acc);
    return hipGetLastError();
}
