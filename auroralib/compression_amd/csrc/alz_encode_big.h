// alz_encode_big.h -- ONE big stream on the whole GPU, encoder side (included at the end of alz_encode.hip: same translation unit, same
// EncGeom / match array / MatchSearch).
//
// The batch encoder gives a stream one workgroup for prev() (kernel A), and one wavefront for the parse and the emission: a lone
// 1 000 KiB stream -- what the reference's own benchmark compresses, Benchmarks/Benchmarks/TestAllAlgorithms.cs:41-42 -- takes 20-120 ms,
// several times what the managed encoder needs on one CPU core; here 0.13-0.4 ms of kernels at quality 0, 0.3-6 at quality 15.  Nothing in the pipeline is sequential by nature:
//
//   A'  prev(): the stream is cut into segments of S positions, and kernel A runs on "virtual streams" [j S - W, (j + 1) S) -- W >= maxDistance
//       positions of warm-up in front of every segment.  A link found inside a virtual stream is the stream's link; one that would reach in
//       front of it is longer than maxDistance, which ends a chain walk exactly as no link does (LzChainMatchFinder.cs:259-260).  The links
//       of the warm-up positions are thrown away (the segment before has them right): benc_gather copies the rest into the arrays kernel B
//       reads, min-length-table links moved from segment to stream positions.
//   B   MatchSearch for every position: the kernels with one position per lane (enc_match_kernel; enc_match_dyn_kernel where matches are
//       short), on as many workgroups as the stream has blocks; a candidate that cannot win is not measured.  (Not the two-phase kernel the
//       synthetic batches use from maxChain 3 on: it walks every chain to its end before it compares, and on a lone stream of real data --
//       Test.bmp: runs, repeated rows -- the walk that ends at the first candidate of full length is the faster one: 2.7 against 6.1 ms for
//       Yaz0 at quality 15, 0.50 against 1.52 for LZSS.)
//   C   the greedy / lazy parse (FindNextBestMatch :157-212) is a walk "cursor += jump[cursor]", and jump[p] -- what the parse does IF its
//       cursor is at p: no match, a match here, or a literal and the better match at p + 1 -- is a pure function of match[p] and match[p + 1].
//       A linked list through the positions: the cursors are the nodes reachable from 0, found by list ranking (mark + square the jump
//       table, as alz_big.hip does for the decoders' group starts) -- on two levels: inside tiles of 1 024 nodes in LDS, and over the tiles' exits.
//   D   with the cursors known every position knows whether a token starts on it and which; prefix sums over the positions give every
//       token its number (-> flag group and bit), its payload offset and -- Yay0 / MIO0 -- its place in the literal section; the payload
//       bytes go straight to the destination, the flag bits through a byte per token, one thread per flag group gathers them.
//
// A position whose candidate ran into kernel B's compare cap (2 040 bytes) is searched again exactly, by a whole wavefront -- what the batch parse
// does on one lane when its cursor meets one.  Not every such position (inside a run or a repeated row of Test.bmp every position is one, and
// each would measure the whole rest of it: quadratic; 380 ms for Test.bmp as an LZ4 block at quality 15), only those somebody's cursor
// could land on: every position with a known match asks for the place its cursor would go to, a searched one asks for its own target, and
// so on until nobody asks (benc_request0 / benc_exact_g1 / benc_exact_rest).  What the true parse visits has been asked for by the position it
// came from.  Output: bit-identical to the batch pipeline's, which is bit-identical to the oracle's.  (The path can decline a stream --
// ctl[BC_BAD], the host then runs the batch pipeline --; nothing does at present.)
#pragma once

namespace {

#define BENC_TILE 1024u
enum { BC_BAD = 0, BC_CAPN = 1, BC_TAIL = 2, BC_T = 3, BC_P = 4, BC_U = 5, BC_F0N = 8, BC_F1N = 9, BC_WORDS = 16 };

struct BencArgs {
    const u8* data;          // the stream
    u32 N;                   // its length
    int n, limit;            // the finder's bound (N - tail) and the last position it searches (n - 4)
    u32 nodes;               // limit + 2: the positions 0..limit and the end node limit + 1
    u32 S, W, K, stride;     // segment length, warm-up, segments, ints per segment in the segment arrays
};

// the virtual streams of A' (entries 0..K-1) and the real one (entry K) with their index list and array offsets
__global__ __launch_bounds__(256) void benc_setup(alz_stream real, BencArgs a, alz_stream* __restrict__ vs, u32* __restrict__ vindex, u64* __restrict__ vpos,
                                                  u32* __restrict__ ctl, alz_result* __restrict__ result, alz_encode_aux* __restrict__ aux) {
    const u32 j = blockIdx.x * 256u + threadIdx.x;
    if (j < (u32)BC_WORDS) ctl[j] = j == (u32)BC_TAIL ? (u32)(a.limit + 1) : 0u;      // (the control words: zero, the literals behind the parse start at limit + 1 unless a match says otherwise)
    if (j == 0u) { alz_result r; r.dst_len = 0xFFFFFFFFu; r.src_used = 0xFFFFFFFFu; r.status = -1; r.reserved = 0xFFFFFFFFu; *result = r; aux->aux0 = 0; aux->aux1 = 0; }
    if (j > a.K) return;
    alz_stream s = real;
    u64 po = 0;
    if (j < a.K) {
        const u32 first = j * a.S, start = first >= a.W ? first - a.W : 0u;
        u32 end = first + a.S; if (end > (u32)a.limit + 1u) end = (u32)a.limit + 1u;
        s.src_off = real.src_off + start;
        s.src_len = end - start + 3u;            // the last hashed position is end - 1: four bytes
        po = (u64)j * a.stride;
    }
    vs[j] = s; vindex[j] = j; vpos[j] = po;
}

// the links of the segments -> the arrays of the stream
template <bool L16>
__global__ __launch_bounds__(256) void benc_gather(BencArgs a, const int* __restrict__ seg4, const int* __restrict__ segm, int* __restrict__ fin4, int* __restrict__ finm,
                                                   u8* __restrict__ mark) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if ((int)p > a.limit) return;
    mark[p] = p == 0u ? 1 : 0;                                     // the ranking's marks: the cursor starts at position 0
    if ((int)p == a.limit) mark[p + 1u] = 0;                       // (the end node)
    const u32 j = p / a.S, first = j * a.S, start = first >= a.W ? first - a.W : 0u, local = p - start;
    const size_t off = (size_t)j * a.stride;
    if (L16) reinterpret_cast<unsigned short*>(fin4)[p] = reinterpret_cast<const unsigned short*>(seg4 + off)[local];
    else { const int v = seg4[off + local]; fin4[p] = v < 0 ? -1 : v + (int)start; }
    if (segm) { const int v = segm[off + local]; finm[p] = v < 0 ? -1 : v + (int)start; }
}

// C0: the match array -> (length, distance) per position.  A position whose candidate ran into kernel B's compare cap has no length yet
// (BENC_UNKNOWN) and state 1 ("capped, nobody has asked for it").
#define BENC_UNKNOWN 0xFFFFFFFFu
__global__ __launch_bounds__(256) void benc_unpack(BencArgs a, const mentry* __restrict__ match, u32* __restrict__ ml, u32* __restrict__ md,
                                                   u32* __restrict__ stt, u32* __restrict__ ctl) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if ((int)p > a.limit) return;
    const uint2 u = m_unpack(__builtin_nontemporal_load(match + p));
    const bool capped = u.y == ALZ_CAPPED;
    const u64 cm = __ballot(capped);
    if (cm && benc_mbcnt(cm) == 0u && capped) atomicAdd(ctl + BC_CAPN, (u32)__popcll(cm));
    ml[p] = capped ? BENC_UNKNOWN : u.y; md[p] = capped ? 0u : u.x;
    if (stt) stt[p] = capped ? 1u : 0u;
}

// the rule of FindNextBestMatch (:157-212) for a cursor at p whose match has l0 bytes and whose neighbour's has l1: what it takes
// (0 a literal, 1 the match here, 2 a literal and the neighbour's match) and where the cursor goes
__device__ __forceinline__ u32 benc_rule(const EncGeom& g, int limit, int pi, int l0, int l1, u32& s) {
    int jump = 1; s = 0;
    if (l0 >= g.min_len) {
        const bool lazyc = l0 <= g.lazy && pi + 1 <= limit;
        if (lazyc && l1 > l0) { s = 2; const int e = pi + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; jump = (pi + 2 > stop ? pi + 2 : stop) - pi; }
        else { s = 1; const int skip = lazyc ? 1 : 0; const int e = pi + l0; const int stop = e < limit + 1 ? e : limit + 1; jump = (pi + 1 + skip > stop ? pi + 1 + skip : stop) - pi; }
    }
    return (u32)(pi + jump);
}
__device__ __forceinline__ bool benc_lazy_needs(const EncGeom& g, int limit, int pi, int l0) { return l0 >= g.min_len && l0 <= g.lazy && pi + 1 <= limit; }

// "somebody's cursor may land on x": a capped position is searched exactly once somebody asks
// (`stride`: the format's longest match where that is a few KiB -- LZ11, LZ40: 16 Ki --, else 0.  Inside a long repeat every cursor takes a match
// of exactly that length and lands on the next capped position: a chain of requests, one generation each -- 3.7 of the 7.4 ms of Test.bmp as
// LZ11 at quality 12.  So a request also asks for the positions one, two, ... longest matches on while they are capped: wrong guesses cost a
// search nobody needed, right ones bring the whole chain into the first, parallel generation.)
__device__ __forceinline__ void benc_request(const BencArgs& a, u32 x, u32* stt, u32* front, u32* tail, u32 stride) {
    if ((int)x > a.limit) return;
    if (__hip_atomic_load(stt + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) return;      // (nearly every target: not a capped position -- a load, not an atomic)
    if (atomicCAS(stt + x, 1u, 2u) != 1u) return;
    front[atomicAdd(tail, 1u)] = x;
    if (stride == 0u) return;
    for (u32 k = 0, y = x + stride; k < 64u && (int)y <= a.limit; k++, y += stride) {
        if (atomicCAS(stt + y, 1u, 2u) != 1u) break;
        front[atomicAdd(tail, 1u)] = y;
    }
}

// C1a: who may land on a capped position?  Every position whose own match is known asks for the place its cursor would go to (or for its
// neighbour, if the lazy rule needs the neighbour's length first); position 0 for itself.
__global__ __launch_bounds__(256) void benc_request0(BencArgs a, EncGeom g, const u32* __restrict__ ml, u32* __restrict__ stt, u32* __restrict__ front, u32* __restrict__ ctl) {
    if (ctl[BC_CAPN] == 0u) return;
    const u32 stride = g.max_len <= 65536 ? (u32)g.max_len : 0u;
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if ((int)p > a.limit) return;
    const u32 l0 = ml[p];
    if (l0 == BENC_UNKNOWN) { if (p == 0u) benc_request(a, 0u, stt, front, ctl + BC_F0N, stride); return; }
    u32 l1 = (int)p + 1 <= a.limit ? ml[p + 1u] : 0u;
    if (l1 == BENC_UNKNOWN) {
        if (benc_lazy_needs(g, a.limit, (int)p, (int)l0)) { benc_request(a, p + 1u, stt, front, ctl + BC_F0N, stride); return; }   // (its own target follows when the neighbour is known)
        l1 = 0;
    }
    u32 s;
    benc_request(a, benc_rule(g, a.limit, (int)p, (int)l0, (int)l1, s), stt, front, ctl + BC_F0N, stride);
}

// One requested position, by one wavefront: its match, exactly; then where ITS cursor would go -- another request -- and, if the position in
// front of it was waiting for this length (lazy rule), that one's target too.
template <bool MINT>
__device__ __forceinline__ void benc_resolve(const BencArgs& a, const EncGeom& g, const int* p4, const int* pm, u32 c, u32* ml, u32* md, u32* stt, u32* front, u32* tail) {
    const bool l0lane = benc_lane() == 0u;
    const u32 stride = g.max_len <= 65536 ? (u32)g.max_len : 0u;
    auto known = [&](u32 q, int& d, int& l) -> bool {
        const u32 v = __hip_atomic_load(ml + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (v == BENC_UNKNOWN) return false;
        l = (int)v; d = (int)__hip_atomic_load(md + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    };
    auto settle = [&](u32 q, int& d, int& l) {                        // the exact match of q (searched here unless somebody has already)
        if (known(q, d, l)) return;
        benc_wave_search<MINT>(a.data, a.n, g, p4, pm, (int)q, d, l);
        if (l0lane) {
            __hip_atomic_store(md + q, (u32)d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ml + q, (u32)l, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the position in front of q, if its own match is known and short enough for the lazy rule, was waiting for this length
        if (q >= 1u) {
            int dq, lq;
            if (stt[q - 1u] == 0u && known(q - 1u, dq, lq) && benc_lazy_needs(g, a.limit, (int)q - 1, lq)) {
                u32 s; const u32 x = benc_rule(g, a.limit, (int)q - 1, lq, l, s);
                if (l0lane) benc_request(a, x, stt, front, tail, stride);
            }
        }
    };
    int d0, l0, d1 = 0, l1 = 0;
    settle(c, d0, l0);
    if (benc_lazy_needs(g, a.limit, (int)c, l0)) settle(c + 1u, d1, l1);
    u32 s; const u32 x = benc_rule(g, a.limit, (int)c, l0, l1, s);
    if (l0lane) benc_request(a, x, stt, front, tail, stride);
}

// C1b: the first generation of requests, all wavefronts of the GPU; what they ask for goes onto a second list
template <bool MINT>
__global__ __launch_bounds__(64) void benc_exact_g1(BencArgs a, EncGeom g, const int* __restrict__ p4, const int* __restrict__ pm, u32* __restrict__ ml, u32* __restrict__ md,
                                                    u32* __restrict__ stt, const u32* __restrict__ front0, u32* __restrict__ front1, u32* __restrict__ ctl) {
    const u32 count = ctl[BC_F0N];
    for (u32 i = blockIdx.x; i < count; i += gridDim.x) benc_resolve<MINT>(a, g, p4, pm, front0[i], ml, md, stt, front1, ctl + BC_F1N);
}
// C1c: the generations behind it -- a cursor that lands on a capped position from a capped position: few, and one after the other by
// nature -- by ONE workgroup, level by level, until nobody asks any more
template <bool MINT>
__global__ __launch_bounds__(1024) void benc_exact_rest(BencArgs a, EncGeom g, const int* __restrict__ p4, const int* __restrict__ pm, u32* __restrict__ ml, u32* __restrict__ md,
                                                        u32* __restrict__ stt, u32* __restrict__ front1, u32* __restrict__ ctl) {
    const u32 w = threadIdx.x >> 6;
    u32 f0 = 0;
    for (;;) {
        __syncthreads();
        const u32 f1 = __hip_atomic_load(ctl + BC_F1N, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (f1 == f0) break;
        for (u32 i = f0 + w; i < f1; i += 16u) benc_resolve<MINT>(a, g, p4, pm, front1[i], ml, md, stt, front1, ctl + BC_F1N);
        __threadfence();
        f0 = f1;
    }
}

// C2: what FindNextBestMatch does if its cursor is at p (enc_roles_kernel's rule, one position per thread) -- and the first level of the
// list ranking.  As 21 rounds over all nodes the ranking was 105 of the ~220 us a 1 000 KiB stream takes at quality 0 (a round is a launch:
// ~5 us whatever it does).  A tile of 1 024 nodes fits LDS: ten rounds of pointer jumping there give every node its EXIT -- the first node of
// its chain behind the tile; the chain of exits from node 0 has at most one node per tile, so the rounds over all nodes are
// ceil(log2 tiles) + 1 = 11; they mark where the parse ENTERS each tile, and benc_tile_mark marks the cursors inside from there.
// FROM_MATCH (formats kernel B cannot cap): straight from the match array, benc_unpack's job done here.
#define BENC_RTILE 1024u
template <bool FROM_MATCH>
__global__ __launch_bounds__(1024) void benc_tile_exit(BencArgs a, EncGeom g, const mentry* __restrict__ match, u32* __restrict__ ml, u32* __restrict__ md,
                                                       u32* __restrict__ next1, u32* __restrict__ exitj, u8* __restrict__ sr) {
    __shared__ u32 J[BENC_RTILE];
    const u32 ts = blockIdx.x * BENC_RTILE, te = ts + BENC_RTILE, tid = threadIdx.x, p = ts + tid;
    const int limit = a.limit;
    u32 x = 0xFFFFFFFFu;                                           // (a thread behind the last node: never inside a tile)
    if (p < a.nodes) {
        if ((int)p > limit) x = p;                                 // the end node points at itself
        else {
            u32 l0, l1 = 0;
            if (FROM_MATCH) {
                const uint2 u = m_unpack(__builtin_nontemporal_load(match + p));
                l0 = u.y; ml[p] = u.y; md[p] = u.x;
                if ((int)p + 1 <= limit) l1 = m_unpack(match[p + 1u]).y;
            } else {
                // (a capped position nobody asked for cannot be reached from position 0: what it says does not matter, as long as it says something)
                l0 = ml[p]; if (l0 == BENC_UNKNOWN) l0 = 0;
                if ((int)p + 1 <= limit) { l1 = ml[p + 1u]; if (l1 == BENC_UNKNOWN) l1 = 0; }
            }
            u32 s; x = benc_rule(g, limit, (int)p, (int)l0, (int)l1, s);
            sr[p] = (u8)s;
        }
        next1[p] = x;
    }
    J[tid] = x;
    __syncthreads();
    for (u32 r = 0; r < 10u; r++) {
        const u32 j = J[tid];
        const u32 j2 = j < te ? J[j - ts] : j;                     // (j >= ts: a jump goes forward, the end node to itself)
        __syncthreads();
        J[tid] = j2;
        __syncthreads();
    }
    if (p < a.nodes) exitj[p] = J[tid];
}

// one round of list ranking: every marked node marks where its jump lands, every jump is squared (double-buffered: a round must see
// jumps of exactly 2^k hops)
__global__ __launch_bounds__(256) void benc_rank_round(const u32* __restrict__ jump_a, u32* __restrict__ jump_b, u8* __restrict__ mark, u32 nodes) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= nodes) return;
    // (TWO rounds per launch: the table to the fourth power, a marked node marks the three nodes on the way -- alz_big.hip)
    const u32 j1 = jump_a[p], j2 = jump_a[j1], j3 = jump_a[j2];
    if (mark[p]) { mark[j1] = 1; mark[j2] = 1; mark[j3] = 1; }
    jump_b[p] = jump_a[j3];
}

// the last level: the parse enters the tile at the node the rounds marked (at most one; node 0 in the first tile); eleven rounds of the same
// ranking on the one-hop jumps in LDS mark the cursors inside.  The last cursor of all says where the literals behind the parse start (the
// end of its match, at least limit + 1).
__global__ __launch_bounds__(1024) void benc_tile_mark(BencArgs a, const u32* __restrict__ next1, const u8* __restrict__ sr, const u32* __restrict__ ml,
                                                       u8* __restrict__ mark, u32* __restrict__ ctl) {
    __shared__ u32 JA[BENC_RTILE], JB[BENC_RTILE], M[BENC_RTILE];
    const u32 ts = blockIdx.x * BENC_RTILE, te = ts + BENC_RTILE, tid = threadIdx.x, p = ts + tid;
    JA[tid] = p < a.nodes ? next1[p] : 0xFFFFFFFFu;
    M[tid] = p < a.nodes ? mark[p] : 0u;
    __syncthreads();
    u32* ja = JA; u32* jb = JB;
    for (u32 r = 0; r < 11u; r++) {
        const u32 j = ja[tid];
        const bool inside = j < te;
        if (inside && M[tid]) M[j - ts] = 1u;
        jb[tid] = inside ? ja[j - ts] : j;
        __syncthreads();
        u32* t = ja; ja = jb; jb = t;
    }
    if (p >= a.nodes) return;
    const u32 m = M[tid];
    mark[p] = (u8)m;
    if (m && (int)p <= a.limit) {
        const u32 s = sr[p];
        const u32 e = s == 1u ? p + ml[p] : s == 2u ? p + 1u + ml[p + 1u] : p + 1u;
        if (e >= (u32)a.limit + 1u) ctl[BC_TAIL] = e;
    }
}

template <int FMT> struct FlagFmt {
    static constexpr bool THREE = (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0);
    static constexpr bool LIT_BIT = (FMT == ALZ_FMT_LZSS || FMT == ALZ_FMT_YAZ0 || FMT == ALZ_FMT_LZHUDSON || THREE);
    static constexpr bool MSB = (FMT != ALZ_FMT_LZSS && FMT != ALZ_FMT_CLZ0);
    static constexpr u32 FBITS = FMT == ALZ_FMT_LZHUDSON ? 32u : 8u, FB = FBITS / 8u;
};

// The token that starts at position p, if any: 0 none, 1 a literal, 2 a match (its length in `len`)
struct BencTok { u32 kind, len, dist; };
__device__ __forceinline__ BencTok benc_token(const BencArgs& a, u32 p, u32 tail, const u8* mark, const u8* sr, const u32* ml, const u32* md) {
    BencTok t; t.kind = 0; t.len = 0; t.dist = 0;
    if (p >= a.N) return t;
    const int limit = a.limit;
    const bool cur = (int)p <= limit && mark[p];
    const u32 s = cur ? sr[p] : 0u;
    const bool after = p >= 1u && (int)(p - 1u) <= limit && mark[p - 1u] && sr[p - 1u] == 2u;   // the cursor in front took a literal and MY match
    if ((cur && s == 1u) || after) { t.kind = 2; t.len = ml[p]; t.dist = md[p]; }
    else if (cur || p >= tail) t.kind = 1;
    return t;
}
template <int FMT>
__device__ __forceinline__ void benc_sizes(const BencTok& t, u32& psize, u32& usize) {
    typedef FlagFmt<FMT> F;
    psize = 0; usize = 0;
    if (t.kind == 1u) { if (F::THREE) usize = 1; else psize = 1; }
    else if (t.kind == 2u) {
        const u32 len = t.len;
        if (FMT == ALZ_FMT_LZ40) psize = len < 16u ? 2u : len < 272u ? 3u : 4u;
        else if (FMT == ALZ_FMT_LZ11) psize = len <= 16u ? 2u : len <= 272u ? 3u : 4u;
        else if (FMT == ALZ_FMT_YAZ0 || FMT == ALZ_FMT_LZHUDSON) psize = len < 18u ? 2u : 3u;
        else if (FMT == ALZ_FMT_YAY0) { psize = 2; usize = len < 18u ? 0u : 1u; }
        else psize = 2;
    }
}

// D1: tokens, payload bytes and literal-section bytes per tile of 1 024 positions
template <int FMT>
__global__ __launch_bounds__(64) void benc_count(BencArgs a, const u8* __restrict__ mark, const u8* __restrict__ sr, const u32* __restrict__ ml,
                                                 const u32* __restrict__ md, const u32* __restrict__ ctl, u32* __restrict__ tile_t, u32* __restrict__ tile_p,
                                                 u32* __restrict__ tile_u) {
    const u32 tile = blockIdx.x, lane = benc_lane(), tail = ctl[BC_TAIL];
    u32 ct = 0, cp = 0, cu = 0;
    for (u32 r = 0; r < BENC_TILE / 64u; r++) {
        const u32 p = tile * BENC_TILE + r * 64u + lane;
        const BencTok t = benc_token(a, p, tail, mark, sr, ml, md);
        u32 ps, us; benc_sizes<FMT>(t, ps, us);
        ct += t.kind ? 1u : 0u; cp += ps; cu += us;
    }
    ct = benc_last(scan_add(ct)); cp = benc_last(scan_add(cp)); cu = benc_last(scan_add(cu));
    if (lane == 0) { tile_t[tile] = ct; tile_p[tile] = cp; tile_u[tile] = cu; }
}

// exclusive scans over the tiles: workgroup k scans array k (in[k * pitch ..] -> out[k * pitch ..]), its total to ctl[BC_T + k]
__global__ __launch_bounds__(1024) void benc_scan3(const u32* __restrict__ in, u32* __restrict__ out, u32 n, u32 pitch, u32* __restrict__ ctl) {
    __shared__ u32 part[1024];
    const u32 tid = threadIdx.x, k = blockIdx.x;
    in += (size_t)k * pitch; out += (size_t)k * pitch;
    const u32 per = (n + 1023u) / 1024u;
    const u32 b = tid * per, e = b + per < n ? b + per : n;
    u32 s = 0;
    for (u32 i = b; i < e; i++) s += in[i];
    part[tid] = s;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {
        const u32 v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 run = tid ? part[tid - 1] : 0;
    for (u32 i = b; i < e; i++) { out[i] = run; run += in[i]; }
    if (tid == 1023u) ctl[BC_T + k] = part[1023];
}

// D2: every token to its place.  The payload goes straight to the destination, the flag bit into a byte per token, and the first token of a
// flag group notes where the group's flag byte(s) go (FlagWriter order: flag, then the payload of its tokens  IO/FlagWriter.cs:70-80,111-127)
template <int FMT>
__global__ __launch_bounds__(64) void benc_place(BencArgs a, EncGeom g, const u8* __restrict__ mark, const u8* __restrict__ sr, const u32* __restrict__ ml,
                                                 const u32* __restrict__ md, const u32* __restrict__ ctl, const u32* __restrict__ base_t, const u32* __restrict__ base_p,
                                                 const u32* __restrict__ base_u, u8* __restrict__ dst, u32 cap, u8* __restrict__ tokbit, u32* __restrict__ gofs) {
    typedef FlagFmt<FMT> F;
    const u32 tile = blockIdx.x, lane = benc_lane(), tail = ctl[BC_TAIL];
    const u32 T = ctl[BC_T], P = ctl[BC_P], U = ctl[BC_U];
    const u32 nflags = F::FB * ((T + F::FBITS - 1u) / F::FBITS);
    if (ctl[BC_BAD] || (u64)nflags + P + U > cap) return;        // (declined, or no room: benc_result says so)
    u32 tb = base_t[tile], pb = base_p[tile], ub = base_u[tile];
    for (u32 r = 0; r < BENC_TILE / 64u; r++) {
        const u32 p = tile * BENC_TILE + r * 64u + lane;
        if (tile * BENC_TILE + r * 64u >= a.N) break;
        const BencTok t = benc_token(a, p, tail, mark, sr, ml, md);
        u32 psize, usize; benc_sizes<FMT>(t, psize, usize);
        const bool tok = t.kind != 0u;
        const u64 tm = __ballot(tok);
        const u32 ti = tb + benc_mbcnt(tm);
        const u32 pincl = scan_add(psize), poff = pb + pincl - psize;
        const u32 uincl = F::THREE ? scan_add(usize) : 0u, uoff = ub + uincl - usize;
        if (tok) {
            const u32 group = ti / F::FBITS;
            const bool lit = t.kind == 1u;
            tokbit[ti] = (u8)((lit ? F::LIT_BIT : !F::LIT_BIT) ? 1u : 0u);
            if (ti % F::FBITS == 0u) gofs[group] = F::THREE ? F::FB * group : poff + F::FB * group;
            u32 b0 = 0, b1 = 0, b2 = 0, b3 = 0, ps = 0;
            if (lit) b0 = a.data[p];
            else flag_payload<FMT>(g, p, make_uint2(t.dist, t.len), b0, b1, b2, b3, ps);
            if (!F::THREE) {
                u8* o = dst + poff + F::FB * (group + 1u);
                o[0] = (u8)b0; if (psize > 1u) o[1] = (u8)b1; if (psize > 2u) o[2] = (u8)b2; if (psize > 3u) o[3] = (u8)b3;
            } else if (lit) dst[nflags + P + uoff] = (u8)b0;
            else { dst[nflags + poff] = (u8)b0; dst[nflags + poff + 1u] = (u8)b1; if (usize) dst[nflags + P + uoff] = (u8)b2; }
        }
        tb += (u32)__popcll(tm); pb += benc_last(pincl); if (F::THREE) ub += benc_last(uincl);
    }
}

// D3: one thread per flag group; the stream's result
template <int FMT>
__global__ __launch_bounds__(256) void benc_flags(BencArgs a, const u32* __restrict__ ctl, const u8* __restrict__ tokbit, const u32* __restrict__ gofs,
                                                  u8* __restrict__ dst, u32 cap, alz_result* __restrict__ result, alz_encode_aux* __restrict__ aux) {
    typedef FlagFmt<FMT> F;
    const u32 T = ctl[BC_T], P = ctl[BC_P], U = ctl[BC_U];
    const u32 ngroups = (T + F::FBITS - 1u) / F::FBITS, nflags = F::FB * ngroups;
    const u64 total = (u64)nflags + P + U;
    const bool declined = ctl[BC_BAD] != 0u, room = total <= cap;
    const u32 gi = blockIdx.x * 256u + threadIdx.x;
    if (gi == 0u && !declined) {
        alz_result r; r.dst_len = room ? (u32)total : 0u; r.src_used = a.N; r.status = room ? ALZ_ST_OK : ALZ_ST_OUTPUT_CAPACITY; r.reserved = 0;
        *result = r;
        if (aux) { aux->aux0 = F::THREE && room ? nflags : 0u; aux->aux1 = F::THREE && room ? nflags + P : 0u; }
    }
    if (declined || !room || gi >= ngroups) return;
    u32 acc = 0;
    for (u32 k = 0; k < F::FBITS; k++) {
        const u32 ti = gi * F::FBITS + k;
        if (ti < T && tokbit[ti]) acc |= 1u << (F::MSB ? F::FBITS - 1u - k : k);
    }
    u8* o = dst + gofs[gi];
    if (F::FB == 1u) o[0] = (u8)(FMT == ALZ_FMT_LZ40 ? 0u - acc : acc);
    else { o[0] = (u8)(acc >> 24); o[1] = (u8)(acc >> 16); o[2] = (u8)(acc >> 8); o[3] = (u8)acc; }
}

static size_t benc_al(size_t x) { return (x + 255) & ~(size_t)255; }
static u32 benc_rounds(u32 n) { u32 r = 1; while ((1ull << r) < n) r++; return r + 1u; }

struct BencLayout {
    BencArgs a;
    u32 tiles;
    u32* ctl_dev = nullptr;      // the stream's control words (the caller's: they travel to the host with the result)
    size_t vs, vindex, vpos, seg4, segm, fin4, finm, match, ml, md, jump_a, jump_b, next1, front1, mark, sr, tile_in, tile_out, tokbit, bitv, gofs, ctl, total;
    BencLayout(const alz_stream& st, const EncGeom& g, int tail) {
        a.data = nullptr; a.N = st.src_len; a.n = (int)st.src_len - tail; a.limit = a.n - 4;
        a.nodes = (u32)a.limit + 2u;
        a.W = ((u32)g.max_dist + 63u) & ~63u;
        // (a segment costs kernel A its S + W positions; while all segments of the stream are resident at once -- one workgroup per CU --
        // shorter ones finish sooner: a 1 000 KiB Yaz0 stream at quality 8 0.29 -> 0.25 ms of kernels, at quality 15 1.22 -> 0.94)
        {   // as short as keeps the segments at about one per CU (240 of 256), between 4 Ki and 16 Ki positions
            u32 S = ((st.src_len / 240u) + 2047u) & ~2047u;
            if (S < 4096u) S = 4096u;
            if (S > 16384u || a.W > 8192u) S = 16384u;
            a.S = S;
        }
        a.K = ((u32)a.limit + a.S) / a.S;
        a.stride = a.S + a.W + 64u;
        tiles = (a.N + BENC_TILE - 1u) / BENC_TILE + 1u;         // (+ 1: PRS has a token behind the data)
        const size_t np = (size_t)a.N + 64;
        size_t o = 0;
        vs = o; o += benc_al((a.K + 1) * sizeof(alz_stream)); vindex = o; o += benc_al((a.K + 1) * 4); vpos = o; o += benc_al((a.K + 1) * 8);
        seg4 = o; o += benc_al((size_t)a.K * a.stride * 4 + 256);
        segm = o; if (g.use_min_table) o += benc_al((size_t)a.K * a.stride * 4 + 256);
        fin4 = o; o += benc_al(np * 4 + 256);
        finm = o; if (g.use_min_table) o += benc_al(np * 4);
        match = o; o += benc_al(np * 4);
        ml = o; o += benc_al(np * 4); md = o; o += benc_al(np * 4);
        jump_a = o; o += benc_al(np * 4); jump_b = o; o += benc_al(np * 4);
        next1 = o; o += benc_al(np * 4);
        front1 = o; if (g.max_len > g.b_cap) o += benc_al(np * 4);
        mark = o; o += benc_al(np); sr = o; o += benc_al(np);
        tile_in = o; o += benc_al((size_t)3 * (tiles + 64) * 4); tile_out = o; o += benc_al((size_t)3 * (tiles + 64) * 4);
        tokbit = o; o += benc_al(np); bitv = o; o += benc_al(2 * np + 64); gofs = o; o += benc_al((np / 4 + 64) * 4);
        ctl = o; o += benc_al(BC_WORDS * 4);
        total = o;
    }
};

static bool benc_format(int fmt) {
    return fmt == ALZ_FMT_LZSS || fmt == ALZ_FMT_LZ10 || fmt == ALZ_FMT_LZ11 || fmt == ALZ_FMT_LZ40 || fmt == ALZ_FMT_YAZ0 || fmt == ALZ_FMT_YAY0 ||
           fmt == ALZ_FMT_MIO0 || fmt == ALZ_FMT_CLZ0 || fmt == ALZ_FMT_BLZ || fmt == ALZ_FMT_LZHUDSON || fmt == ALZ_FMT_LZ4_BLOCK || fmt == ALZ_FMT_SNAPPY_RAW ||
           fmt == ALZ_FMT_PRS_BE || fmt == ALZ_FMT_PRS_LE || fmt == ALZ_FMT_LZO;
}

template <int FMT>
static void benc_emit(hipStream_t stream, const BencLayout& L, const BencArgs& a, const EncGeom& g, u8* base, u8* dst, u32 cap, alz_result* d_result, alz_encode_aux* d_aux) {
    const u8* mark = base + L.mark; const u8* sr = base + L.sr; const u32* ml = (const u32*)(base + L.ml); const u32* md = (const u32*)(base + L.md);
    u32* ctl = L.ctl_dev; u32* tin = (u32*)(base + L.tile_in); u32* tout = (u32*)(base + L.tile_out);
    const u32 pitch = L.tiles + 64u;
    hipLaunchKernelGGL((benc_count<FMT>), dim3(L.tiles), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, tin, tin + pitch, tin + 2 * pitch);
    hipLaunchKernelGGL(benc_scan3, dim3(3), dim3(1024), 0, stream, tin, tout, L.tiles, pitch, ctl);
    hipLaunchKernelGGL((benc_place<FMT>), dim3(L.tiles), dim3(64), 0, stream, a, g, mark, sr, ml, md, ctl, tout, tout + pitch, tout + 2 * pitch, dst, cap,
                       base + L.tokbit, (u32*)(base + L.gofs));
    const u32 maxgroups = (a.N + FlagFmt<FMT>::FBITS - 1u) / FlagFmt<FMT>::FBITS + 1u;
    hipLaunchKernelGGL((benc_flags<FMT>), dim3((maxgroups + 255u) / 256u), dim3(256), 0, stream, a, ctl, base + L.tokbit, (const u32*)(base + L.gofs), dst, cap, d_result, d_aux);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// LZ4 blocks and raw Snappy (enc_parse_seq_kernel's formats): a sequence = the literals since the last match + the match, so a match start
// needs where the match in front of it ended -- a prefix MAX over the positions -- before its size is known, and the sizes' prefix SUM
// before it can be written.  Three passes over tiles of 1 024 positions with the same body: 0 the tile's highest match end, 1 the tile's
// bytes, 2 the bytes themselves; a scan over the tiles between them.  The literals behind the last match (LZ4: always a sequence) go out
// through a kernel of their own, one byte per thread.
enum { BC_COVER = 6, BC_SEQ = 7 };
__global__ __launch_bounds__(1024) void benc_scanmax(const u32* __restrict__ in, u32* __restrict__ out, u32 n, u32* __restrict__ total_out) {
    __shared__ u32 part[1024];
    const u32 tid = threadIdx.x;
    const u32 per = (n + 1023u) / 1024u;
    const u32 b = tid * per, e = b + per < n ? b + per : n;
    u32 s = 0;
    for (u32 i = b; i < e; i++) s = in[i] > s ? in[i] : s;
    part[tid] = s;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {
        const u32 v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        if (v > part[tid]) part[tid] = v;
        __syncthreads();
    }
    u32 run = tid ? part[tid - 1] : 0;
    for (u32 i = b; i < e; i++) { out[i] = run; run = in[i] > run ? in[i] : run; }
    if (tid == 1023u) *total_out = part[1023];
}

__device__ __forceinline__ u32 benc_snappy_varint(u32 n) { return n < 0x80u ? 1u : n < 0x4000u ? 2u : n < 0x200000u ? 3u : n < 0x10000000u ? 4u : 5u; }

template <int FMT>
__device__ __forceinline__ u64 benc_seq_total(const BencArgs& a, const u32* ctl) {
    typedef SeqFmt<FMT> F;
    constexpr bool LZ4 = FMT == ALZ_FMT_LZ4_BLOCK;
    const u32 plain = a.N - ctl[BC_COVER];
    return (u64)(LZ4 ? 0u : benc_snappy_varint(a.N)) + ctl[BC_SEQ] + ((LZ4 || plain) ? F::lit_hdr(plain) : 0u) + plain;
}

template <int FMT, int PASS>
__global__ __launch_bounds__(64) void benc_seq(BencArgs a, const u8* __restrict__ mark, const u8* __restrict__ sr, const u32* __restrict__ ml,
                                               const u32* __restrict__ md, const u32* __restrict__ ctl, const u32* __restrict__ cover_base,
                                               const u32* __restrict__ off_base, u32* __restrict__ tile_out, u8* __restrict__ dst, u32 cap) {
    typedef SeqFmt<FMT> F;
    constexpr bool LZ4 = FMT == ALZ_FMT_LZ4_BLOCK;
    const u32 tile = blockIdx.x, lane = benc_lane();
    if (PASS == 2 && (ctl[BC_BAD] || benc_seq_total<FMT>(a, ctl) > cap)) return;
    u32 cover = PASS >= 1 ? cover_base[tile] : 0u;
    u32 obase = PASS == 2 ? off_base[tile] + (LZ4 ? 0u : benc_snappy_varint(a.N)) : 0u;
    for (u32 r = 0; r < BENC_TILE / 64u; r++) {
        const u32 P = tile * BENC_TILE + r * 64u, p = P + lane;
        if (P >= a.N) break;
        const BencTok t = benc_token(a, p, 0xFFFFFFFFu, mark, sr, ml, md);
        const bool start = t.kind == 2u;
        if (__ballot(start) == 0ull) continue;                                // (no match starts here: the literals wait for the next one)
        const u32 M = start ? t.len : 0u, D = t.dist;
        const u32 mend = start ? p + M : 0u;
        const u32 pmax = scan_max(mend);
        if (PASS >= 1) {
            u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)pmax, 0x138, 0xF, 0xF, false);   // wave_shr:1 -> max over lanes below
            if (before < cover) before = cover;
            const u32 L = start ? p - before : 0u;
            const u32 lh = start ? F::lit_hdr(L) : 0u;
            const u32 esz = start ? lh + L + F::match_size(D, M) : 0u;
            const u32 incl = scan_add(esz);
            if (PASS == 2) {
                const u32 off = obase + incl - esz;
                if (start) {
                    F::put_lit_hdr(dst + off, L, M, false);
                    F::put_match(dst + off + lh + L, D, M);
                }
                // the literals, as in enc_parse_seq_kernel: every literal position whose sequence starts in this group of 64 stores its own
                // byte; what the first start owns of earlier groups the wavefront copies
                const u64 sm = __ballot(start);
                const u64 above = (lane < 63u ? sm >> (lane + 1u) : 0ull);
                const int s = above ? (int)lane + 1 + (int)__builtin_ctzll(above) : (int)lane;
                const u32 sbef = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)before);
                const u32 sbase = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)(off + lh - before));
                if (above && !start && p >= sbef && p < a.N) dst[sbase + p] = a.data[p];
                const int f0 = (int)__builtin_ctzll(sm);
                const u32 fbef = (u32)__builtin_amdgcn_readlane((int)before, f0);
                if (fbef < P) wave_copy(dst + (u32)__builtin_amdgcn_readlane((int)(off + lh), f0), a.data + fbef, P - fbef, (int)lane);
            }
            obase += benc_last(incl);
        }
        const u32 wmax = benc_last(pmax);
        if (wmax > cover) cover = wmax;
    }
    if (PASS == 0 && lane == 0) tile_out[tile] = cover;
    if (PASS == 1 && lane == 0) tile_out[tile] = obase;
}

// the literals behind the last match, and the stream's result
template <int FMT>
__global__ __launch_bounds__(256) void benc_seq_end(BencArgs a, const u32* __restrict__ ctl, u8* __restrict__ dst, u32 cap, alz_result* __restrict__ result,
                                                    alz_encode_aux* __restrict__ aux) {
    typedef SeqFmt<FMT> F;
    constexpr bool LZ4 = FMT == ALZ_FMT_LZ4_BLOCK;
    if (ctl[BC_BAD]) return;
    const u64 total = benc_seq_total<FMT>(a, ctl);
    const bool room = total <= cap;
    const u32 cover = ctl[BC_COVER], plain = a.N - cover, k = LZ4 ? 0u : benc_snappy_varint(a.N);
    const u32 lh = (LZ4 || plain) ? F::lit_hdr(plain) : 0u, obase = k + ctl[BC_SEQ];
    const u32 q = blockIdx.x * 256u + threadIdx.x;
    if (q == 0u) {
        alz_result r; r.dst_len = room ? (u32)total : 0u; r.src_used = a.N; r.status = room ? ALZ_ST_OK : ALZ_ST_OUTPUT_CAPACITY; r.reserved = 0;
        *result = r;
        if (aux) { aux->aux0 = 0; aux->aux1 = 0; }
        if (room) {
            if (!LZ4) { u32 v = a.N, i = 0; while (v >= 0x80u) { dst[i++] = (u8)((v | 0x80u) & 0xFFu); v >>= 7; } dst[i] = (u8)v; }   // the decompressed length as a varint  Snappy.cs:126-135
            if (LZ4 || plain) F::put_lit_hdr(dst + obase, plain, 4u, true);
        }
    }
    if (room && q < plain) dst[obase + lh + q] = a.data[cover + q];
}

template <int FMT>
static void benc_emit_seq(hipStream_t stream, const BencLayout& L, const BencArgs& a, u8* base, u8* dst, u32 cap, alz_result* d_result, alz_encode_aux* d_aux) {
    const u8* mark = base + L.mark; const u8* sr = base + L.sr; const u32* ml = (const u32*)(base + L.ml); const u32* md = (const u32*)(base + L.md);
    u32* ctl = L.ctl_dev; u32* tin = (u32*)(base + L.tile_in); u32* tout = (u32*)(base + L.tile_out);
    const u32 pitch = L.tiles + 64u;
    hipLaunchKernelGGL((benc_seq<FMT, 0>), dim3(L.tiles), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, (const u32*)nullptr, (const u32*)nullptr, tin, dst, cap);
    hipLaunchKernelGGL(benc_scanmax, dim3(1), dim3(1024), 0, stream, tin, tout, L.tiles, ctl + BC_COVER);
    hipLaunchKernelGGL((benc_seq<FMT, 1>), dim3(L.tiles), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, tout, (const u32*)nullptr, tin + pitch, dst, cap);
    hipLaunchKernelGGL(benc_scan3, dim3(1), dim3(1024), 0, stream, tin + pitch, tout + pitch, L.tiles, pitch, ctl + (BC_SEQ - BC_T));
    hipLaunchKernelGGL((benc_seq<FMT, 2>), dim3(L.tiles), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, tout, tout + pitch, (u32*)nullptr, dst, cap);
    hipLaunchKernelGGL((benc_seq_end<FMT>), dim3((a.N + 255u) / 256u), dim3(256), 0, stream, a, ctl, dst, cap, d_result, d_aux);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// PRS (enc_emit_prs_kernel's rules, PRS.cs:104-159): tokens of one, two or four flag bits.  With B0 = the flag bits in front of a token and
// pidx = the payload bytes in front of it -- two prefix sums over the positions -- everything has its place: the payload behind
// floor(Bp / 8) + 1 flag bytes (Bp: the bits written when the payload is handed over), the offset byte of a short match whose four bits
// just completed a flag byte in FRONT of the next flag byte; flag byte k in front of the first payload that waits for it -- the smallest
// (kp + pidx + special) among the tokens with kp = k: an atomic minimum.  The flag bits go through a byte per bit; one thread per flag
// byte gathers them.  A match of length 2 further than 0x100 back is written as two literals; the end token (bit 0, two zero bytes, bit 1)
// is the token of the position behind the data.
enum { BC_BITS = 6, BC_PAY = 7 };
struct BencPrs { u32 kind, len, dist, nbits, psize; bool shortm; };   // kind: 0 none, 1 literal, 2 match, 3 the end token
__device__ __forceinline__ BencPrs benc_prs_token(const BencArgs& a, u32 p, u32 tail, const u8* mark, const u8* sr, const u32* ml, const u32* md) {
    BencPrs r; r.kind = 0; r.len = 0; r.dist = 0; r.nbits = 0; r.psize = 0; r.shortm = false;
    if (p > a.N) return r;
    if (p == a.N) { r.kind = 3; r.nbits = 2; r.psize = 2; return r; }
    const BencTok t = benc_token(a, p, tail, mark, sr, ml, md);
    auto dropped = [](const BencTok& k) { return k.kind == 2u && k.len == 2u && k.dist > 0x100u; };   // PRS.cs: not worth a long match -- literals
    if (t.kind == 2u && !dropped(t)) {
        r.kind = 2; r.len = t.len; r.dist = t.dist;
        r.shortm = t.dist <= 0x100u && t.len <= 5u;
        r.nbits = r.shortm ? 4u : 2u; r.psize = r.shortm ? 1u : (t.len > 9u ? 3u : 2u);
        return r;
    }
    bool lit = t.kind != 0u;                                       // a literal, or the first byte of a dropped match
    if (!lit && p >= 1u) lit = dropped(benc_token(a, p - 1u, tail, mark, sr, ml, md));   // its second byte
    if (lit) { r.kind = 1; r.nbits = 1; r.psize = 1; }
    return r;
}

template <bool BIG, int PASS>
__global__ __launch_bounds__(64) void benc_prs(BencArgs a, const u8* __restrict__ mark, const u8* __restrict__ sr, const u32* __restrict__ ml, const u32* __restrict__ md,
                                               const u32* __restrict__ ctl, const u32* __restrict__ base_b, const u32* __restrict__ base_p, u32* __restrict__ tile_b,
                                               u32* __restrict__ tile_p, u8* __restrict__ dst, u32 cap, u8* __restrict__ bitv, u32* __restrict__ gofs) {
    const u32 tile = blockIdx.x, lane = benc_lane(), tail = ctl[BC_TAIL];
    if (PASS == 1) { const u64 total = (u64)((ctl[BC_BITS] + 7u) >> 3) + ctl[BC_PAY]; if (ctl[BC_BAD] || total > cap) return; }
    u32 bb = PASS == 1 ? base_b[tile] : 0u, pb = PASS == 1 ? base_p[tile] : 0u;
    for (u32 r = 0; r < BENC_TILE / 64u; r++) {
        const u32 P = tile * BENC_TILE + r * 64u, p = P + lane;
        if (P > a.N) break;
        const BencPrs t = benc_prs_token(a, p, tail, mark, sr, ml, md);
        const u32 bincl = scan_add(t.nbits), pincl = scan_add(t.psize);
        if (PASS == 1 && t.kind) {
            const bool lit = t.kind == 1u, endtok = t.kind == 3u;
            const u32 B0 = bb + bincl - t.nbits, pidx = pb + pincl - t.psize;
            const u32 Bp = B0 + (lit ? 0u : t.shortm ? 4u : 1u);
            const bool special = t.shortm && (Bp & 7u) == 0u;
            const u32 kp = Bp >> 3;
            const u32 out = kp + 1u - (special ? 1u : 0u) + pidx;
            atomicMin(gofs + kp, kp + pidx + (special ? 1u : 0u));
            const u32 l2 = t.len - 2u;
            const u32 pattern = lit ? 1u : t.shortm ? ((((l2 >> 1) & 1u) << 2) | ((l2 & 1u) << 3)) : 2u;   // bit i of `pattern` = my i-th flag bit
            for (u32 i = 0; i < t.nbits; i++) bitv[B0 + i] = (u8)((pattern >> i) & 1u);
            if (lit) dst[out] = a.data[p];
            else if (t.shortm) dst[out] = (u8)((0u - t.dist) & 0xFFu);
            else if (endtok) { dst[out] = 0; dst[out + 1] = 0; }
            else {
                u32 v = ((0u - t.dist) << 3) & 0xFFFFu;
                if (t.len <= 9u) v |= t.len - 2u;
                if (BIG) { dst[out] = (u8)(v >> 8); dst[out + 1] = (u8)(v & 0xFFu); } else { dst[out] = (u8)(v & 0xFFu); dst[out + 1] = (u8)(v >> 8); }
                if (t.len > 9u) dst[out + 2] = (u8)(t.len - 1u);
            }
        }
        bb += benc_last(bincl); pb += benc_last(pincl);
    }
    if (PASS == 0 && lane == 0) { tile_b[tile] = bb; tile_p[tile] = pb; }
}

template <bool BIG>
__global__ __launch_bounds__(256) void benc_prs_flags(BencArgs a, const u32* __restrict__ ctl, const u8* __restrict__ bitv, const u32* __restrict__ gofs,
                                                      u8* __restrict__ dst, u32 cap, alz_result* __restrict__ result, alz_encode_aux* __restrict__ aux) {
    const u32 bits = ctl[BC_BITS], nflags = (bits + 7u) >> 3;
    const u64 total = (u64)nflags + ctl[BC_PAY];
    const bool declined = ctl[BC_BAD] != 0u, room = total <= cap;
    const u32 k = blockIdx.x * 256u + threadIdx.x;
    if (k == 0u && !declined) {
        alz_result r; r.dst_len = room ? (u32)total : 0u; r.src_used = a.N; r.status = room ? ALZ_ST_OK : ALZ_ST_OUTPUT_CAPACITY; r.reserved = 0;
        *result = r;
        if (aux) { aux->aux0 = 0; aux->aux1 = 0; }
    }
    if (declined || !room || k >= nflags) return;
    u32 acc = 0;
    for (u32 i = 0; i < 8u; i++) { const u32 b = 8u * k + i; if (b < bits && bitv[b]) acc |= 1u << (BIG ? 7u - i : i); }   // (Dispose(): a partial flag byte goes out with its unused bits zero)
    dst[gofs[k]] = (u8)acc;
}

template <bool BIG>
static hipError_t benc_emit_prs(hipStream_t stream, const BencLayout& L, const BencArgs& a, u8* base, u8* dst, u32 cap, alz_result* d_result, alz_encode_aux* d_aux) {
    const u8* mark = base + L.mark; const u8* sr = base + L.sr; const u32* ml = (const u32*)(base + L.ml); const u32* md = (const u32*)(base + L.md);
    u32* ctl = L.ctl_dev; u32* tin = (u32*)(base + L.tile_in); u32* tout = (u32*)(base + L.tile_out);
    u8* bitv = base + L.bitv; u32* gofs = (u32*)(base + L.gofs);
    const u32 pitch = L.tiles + 64u, tiles1 = a.N / BENC_TILE + 1u;            // (the tile of the position behind the data: the end token)
    hipError_t e = hipMemsetAsync(gofs, 0xFF, ((size_t)a.N / 4 + 64) * 4, stream);   // (flag bytes: at most (2 N + 2) / 8 + 1)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((benc_prs<BIG, 0>), dim3(tiles1), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, (const u32*)nullptr, (const u32*)nullptr, tin, tin + pitch, dst, cap, bitv, gofs);
    hipLaunchKernelGGL(benc_scan3, dim3(2), dim3(1024), 0, stream, tin, tout, tiles1, pitch, ctl + (BC_BITS - BC_T));
    hipLaunchKernelGGL((benc_prs<BIG, 1>), dim3(tiles1), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, tout, tout + pitch, (u32*)nullptr, (u32*)nullptr, dst, cap, bitv, gofs);
    hipLaunchKernelGGL((benc_prs_flags<BIG>), dim3((a.N / 4u + 2u + 255u) / 256u), dim3(256), 0, stream, a, ctl, bitv, gofs, dst, cap, d_result, d_aux);
    return hipSuccess;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// LZO1X (enc_parse_lzo_kernel's rules, LZO.cs:141-250).  The writer is sequential only at the head of a stream, until the first match is out:
// one wavefront walks it the reference's way (benc_lzo_head).  From there every match start is one unit -- a literal run of >= 4 in front
// of it, its token, the 0-3 literals behind it -- whose size follows from its own numbers, the end of the match in front of it (prefix
// max) and whether a start or the end of the data lies within three bytes behind it: the three passes of the sequence formats.
enum { BC_LZ_SP0 = 10, BC_LZ_OB = 11, BC_LZ_MO = 12, BC_LZ_STATUS = 13, BC_LZ_DONE = 14, BC_LZ_FAIL = 15 };
__device__ __forceinline__ void benc_wave_copy16(u8* d, const u8* s, u32 len, u32 lane) {
    u32 i = 0;
    for (; i + 1024u <= len; i += 1024u) { u64 v[2]; __builtin_memcpy(v, s + i + 16u * lane, 16); __builtin_memcpy(d + i + 16u * lane, v, 16); }
    for (u32 j = i + lane; j < len; j += 64u) d[j] = s[j];
}
__global__ __launch_bounds__(64) void benc_lzo_head(BencArgs a, const u8* __restrict__ mark, const u8* __restrict__ sr, const u32* __restrict__ ml_, const u32* __restrict__ md_,
                                                    u32* __restrict__ ctl, u8* __restrict__ dst, u32 cap) {
    const u32 lane = benc_lane(), n = a.N;
    if (ctl[BC_BAD]) return;
    auto is_start = [&](u32 q) { return benc_token(a, q, 0xFFFFFFFFu, mark, sr, ml_, md_).kind == 2u; };
    auto next_start = [&](u32 from) -> u32 {                       // the next start at or behind `from` (n: none)
        for (u32 P = from & ~63u; (int)P <= a.limit; P += 64u) {
            const u32 q = P + lane;
            const u64 m = __ballot(q >= from && is_start(q));
            if (m) return P + (u32)__builtin_ctzll(m);
        }
        return n;
    };
    u32 sp = 0, olen = 0; bool fail = false; int status = ALZ_ST_OK;
    auto put = [&](u32 b) { if (olen < cap) { if (lane == 0) dst[olen] = (u8)b; } else fail = true; olen++; };
    auto copy = [&](u32 from, u32 len) {
        const u32 room = olen < cap ? cap - olen : 0u, k = len < room ? len : room;
        benc_wave_copy16(dst + olen, a.data + from, k, lane);
        if (k < len) fail = true;
        olen += len;
    };
    u32 mo = next_start(0), ml = 0, md = 0;
    if (mo < n) { ml = ml_[mo]; md = md_[mo]; }
    u32 mbit = mo;
    bool clean = false;
    while (sp != n && !clean) {
        u32 plain = mo - sp;
        if (plain != 0u) {
            if (plain < 4u) { const u32 dif = 4u - plain; mo += dif; ml = ml > dif ? ml - dif : 0u; plain = 4u; }
            if (plain > 18u) { put(0); u32 v = plain - 18u; while (v > 255u) { put(0); v -= 255u; } put(v); } else put(plain - 3u);
            if (sp + plain > n) { status = ALZ_ST_BAD_TOKEN; break; }
            copy(sp, plain); sp += plain;
        }
        u32 no = mbit < n ? next_start(mbit + 1u) : n, nl = 0, nd = 0;
        if (no < n) { nl = ml_[no]; nd = md_[no]; }
        if (ml >= 3u) {
            sp += ml;
            u32 emb = no - sp;
            if (no < sp) { status = ALZ_ST_BAD_TOKEN; break; }
            if (emb > 3u) emb = 0;
            u8 tok[8 + 4];                                          // (a token is short except for its extension bytes)
            if (lzo_match_size(md, ml) <= 8u) { const u32 k = lzo_put_match(tok, md, ml, emb); for (u32 i = 0; i < k; i++) put(tok[i]); }
            else if (md <= 16384u) { put(0x20); u32 v = ml - 33u; while (v > 255u) { put(0); v -= 255u; } put(v); put((emb | ((md - 1u) << 2)) & 0xFFu); put(((md - 1u) >> 6) & 0xFFu); }
            else { const u32 d2 = md - 0x4000u, flag = (0x10u | ((d2 & 0x4000u) >> 11)) & 0xFFu; put(flag); u32 v = ml - 9u; while (v > 255u) { put(0); v -= 255u; } put(v); put((emb | (d2 << 2)) & 0xFFu); put((d2 >> 6) & 0xFFu); }
            if (sp + emb > n) { status = ALZ_ST_BAD_TOKEN; break; }
            copy(sp, emb); sp += emb;
            clean = true;
        }
        mo = no; mbit = no; ml = nl; md = nd;
    }
    const bool done = status != ALZ_ST_OK || sp == n;
    if (done && status == ALZ_ST_OK) { put(0x11); put(0); put(0); }
    if (lane == 0) { ctl[BC_LZ_SP0] = sp; ctl[BC_LZ_OB] = olen; ctl[BC_LZ_MO] = mo; ctl[BC_LZ_STATUS] = (u32)status; ctl[BC_LZ_DONE] = done ? 1u : 0u; ctl[BC_LZ_FAIL] = fail ? 1u : 0u; }
}

// the literals behind the last match (0-3 went out with it), the end token: what the whole stream takes
__device__ __forceinline__ u64 benc_lzo_total(const BencArgs& a, const u32* ctl, u32& rest, u32& lsz, u32& obase) {
    const u32 cover = ctl[BC_COVER] > ctl[BC_LZ_SP0] ? ctl[BC_COVER] : ctl[BC_LZ_SP0];
    rest = a.N - cover; if (rest <= 3u) rest = 0u;
    lsz = rest ? lzo_lit_size(rest) : 0u;
    obase = ctl[BC_LZ_OB] + ctl[BC_SEQ];
    return (u64)obase + lsz + rest + 3u;
}

template <int PASS>
__global__ __launch_bounds__(64) void benc_lzo(BencArgs a, const u8* __restrict__ mark, const u8* __restrict__ sr, const u32* __restrict__ ml, const u32* __restrict__ md,
                                               const u32* __restrict__ ctl, const u32* __restrict__ cover_base, const u32* __restrict__ off_base, u32* __restrict__ tile_out,
                                               u8* __restrict__ dst, u32 cap) {
    const u32 tile = blockIdx.x, lane = benc_lane(), n = a.N;
    const u32 mo = ctl[BC_LZ_MO], sp0 = ctl[BC_LZ_SP0];
    if (ctl[BC_BAD] || ctl[BC_LZ_DONE]) { if (PASS < 2 && lane == 0) tile_out[tile] = 0u; return; }
    if (PASS == 2) { u32 r_, l_, o_; if (ctl[BC_LZ_FAIL] || benc_lzo_total(a, ctl, r_, l_, o_) > cap) return; }
    u32 cover = PASS >= 1 ? cover_base[tile] : 0u;
    if (cover < sp0) cover = sp0;
    u32 obase = PASS == 2 ? ctl[BC_LZ_OB] + off_base[tile] : 0u;
    auto is_start = [&](u32 q) { return benc_token(a, q, 0xFFFFFFFFu, mark, sr, ml, md).kind == 2u; };
    for (u32 r = 0; r < BENC_TILE / 64u; r++) {
        const u32 P = tile * BENC_TILE + r * 64u, p = P + lane;
        if (P >= n) break;
        const BencTok t = benc_token(a, p, 0xFFFFFFFFu, mark, sr, ml, md);
        const bool start = t.kind == 2u && p >= mo;
        if (__ballot(start) == 0ull) continue;
        const u32 M = start ? t.len : 0u, D = t.dist;
        const u32 mend = start ? p + M : 0u;
        const u32 pmax = scan_max(mend);
        if (PASS >= 1) {
            u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)pmax, 0x138, 0xF, 0xF, false);
            if (before < cover) before = cover;
            const u32 Lb = start ? p - before : 0u;
            u32 emb = 0;
            if (start) {
                for (u32 kk = 0; kk < 4u; kk++) {
                    const u32 q = mend + kk;
                    const bool hit = q >= n || is_start(q);
                    if (hit) { emb = q >= n ? n - mend : kk; break; }
                }
            }
            const u32 lsz = Lb >= 4u ? lzo_lit_size(Lb) : 0u, lcp = Lb >= 4u ? Lb : 0u;
            const u32 esz = start ? lsz + lcp + lzo_match_size(D, M) + emb : 0u;
            const u32 incl = scan_add(esz);
            if (PASS == 2) {
                const u32 off = obase + incl - esz;
                if (start) {
                    u32 q = off;
                    if (Lb >= 4u) { q += lzo_put_lit(dst + q, Lb); if (Lb <= ALZ_LZO_LANE_LIT) for (u32 i = 0; i < Lb; i++) dst[q + i] = a.data[before + i]; q += Lb; }
                    q += lzo_put_match(dst + q, D, M, emb);
                    for (u32 i = 0; i < emb; i++) dst[q + i] = a.data[mend + i];
                }
                u64 longs = __ballot(start && Lb > ALZ_LZO_LANE_LIT);
                while (longs) {
                    const int l0 = (int)__builtin_ctzll(longs);
                    const u32 so = (u32)__builtin_amdgcn_readlane((int)before, l0), len = (u32)__builtin_amdgcn_readlane((int)Lb, l0);
                    const u32 dq = (u32)__builtin_amdgcn_readlane((int)(off + lsz), l0);
                    benc_wave_copy16(dst + dq, a.data + so, len, lane);
                    longs &= longs - 1ull;
                }
            }
            obase += benc_last(incl);
        }
        const u32 wmax = benc_last(pmax);
        if (wmax > cover) cover = wmax;
    }
    if (PASS == 0 && lane == 0) tile_out[tile] = cover > sp0 ? cover : 0u;
    if (PASS == 1 && lane == 0) tile_out[tile] = obase;
}

__global__ __launch_bounds__(256) void benc_lzo_end(BencArgs a, const u32* __restrict__ ctl, u8* __restrict__ dst, u32 cap, alz_result* __restrict__ result,
                                                    alz_encode_aux* __restrict__ aux) {
    if (ctl[BC_BAD]) return;
    const u32 q = blockIdx.x * 256u + threadIdx.x;
    const int status = (int)ctl[BC_LZ_STATUS];
    alz_result r; r.src_used = a.N; r.reserved = 0;
    if (ctl[BC_LZ_DONE]) {                                         // the head was the whole stream (or refused it)
        const bool ok = status == ALZ_ST_OK && !ctl[BC_LZ_FAIL];
        r.dst_len = ok ? ctl[BC_LZ_OB] : 0u; r.status = status != ALZ_ST_OK ? status : (ok ? ALZ_ST_OK : ALZ_ST_OUTPUT_CAPACITY);
        if (q == 0u) { *result = r; if (aux) { aux->aux0 = 0; aux->aux1 = 0; } }
        return;
    }
    u32 rest, lsz, obase;
    const u64 total = benc_lzo_total(a, ctl, rest, lsz, obase);
    const bool room = !ctl[BC_LZ_FAIL] && total <= cap;
    if (q == 0u) {
        r.dst_len = room ? (u32)total : 0u; r.status = room ? ALZ_ST_OK : ALZ_ST_OUTPUT_CAPACITY;
        *result = r; if (aux) { aux->aux0 = 0; aux->aux1 = 0; }
        if (room) {
            if (rest) (void)lzo_put_lit(dst + obase, rest);
            dst[total - 3u] = 0x11; dst[total - 2u] = 0; dst[total - 1u] = 0;
        }
    }
    if (room && q < rest) dst[obase + lsz + q] = a.data[a.N - rest + q];
}

static void benc_emit_lzo(hipStream_t stream, const BencLayout& L, const BencArgs& a, u8* base, u8* dst, u32 cap, alz_result* d_result, alz_encode_aux* d_aux) {
    const u8* mark = base + L.mark; const u8* sr = base + L.sr; const u32* ml = (const u32*)(base + L.ml); const u32* md = (const u32*)(base + L.md);
    u32* ctl = L.ctl_dev; u32* tin = (u32*)(base + L.tile_in); u32* tout = (u32*)(base + L.tile_out);
    const u32 pitch = L.tiles + 64u;
    hipLaunchKernelGGL(benc_lzo_head, dim3(1), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, dst, cap);
    hipLaunchKernelGGL((benc_lzo<0>), dim3(L.tiles), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, (const u32*)nullptr, (const u32*)nullptr, tin, dst, cap);
    hipLaunchKernelGGL(benc_scanmax, dim3(1), dim3(1024), 0, stream, tin, tout, L.tiles, ctl + BC_COVER);
    hipLaunchKernelGGL((benc_lzo<1>), dim3(L.tiles), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, tout, (const u32*)nullptr, tin + pitch, dst, cap);
    hipLaunchKernelGGL(benc_scan3, dim3(1), dim3(1024), 0, stream, tin + pitch, tout + pitch, L.tiles, pitch, ctl + (BC_SEQ - BC_T));
    hipLaunchKernelGGL((benc_lzo<2>), dim3(L.tiles), dim3(64), 0, stream, a, mark, sr, ml, md, ctl, tout, tout + pitch, (u32*)nullptr, dst, cap);
    hipLaunchKernelGGL(benc_lzo_end, dim3((a.N + 255u) / 256u), dim3(256), 0, stream, a, ctl, dst, cap, d_result, d_aux);
}

}  // namespace

bool alz_encode_big_eligible(int fmt, const void* geom, const alz_stream* st, uint32_t min_bytes) {
    EncGeom g; memcpy(&g, geom, sizeof(g));
    if (!benc_format(fmt) || min_bytes == 0xFFFFFFFFu) return false;
    if (st->src_len < min_bytes || st->src_len < 4096u || st->src_len > 0x20000000u) return false;
    return g.nprops <= 1 && g.link16 && g.hash_bits >= 15 && g.hash_bits <= 20;
}

size_t alz_encode_big_scratch_bytes(int fmt, const void* geom, const alz_stream* st) {
    EncGeom g; memcpy(&g, geom, sizeof(g));
    g.b_cap = ALZ_LEN_CAP;                                        // (not choose_b_cap's: here EVERY capped position somebody may land on is searched, generation by generation -- Yaz0 at quality 0 0.17 -> 1.5 ms with the cap at 48, quality 12 1.1 -> 500)
    return BencLayout(*st, g, fmt == ALZ_FMT_LZ4_BLOCK ? 5 : 0).total + 256;
}

// Enqueues the whole-GPU encode of ONE stream.  d_result / d_aux: the stream's slots; d_ctl: 16 control words of the stream (the caller
// copies them to the host with the result: word 0 != 0 says the path gave the stream up -- nothing does at present -- and the caller runs
// the batch pipeline); d_scratch: alz_encode_big_scratch_bytes().  Everything the path needs initialised it initialises itself.
hipError_t alz_launch_encode_big(int fmt, hipStream_t stream, const void* d_src_base, void* d_dst_base, const alz_stream* st, alz_result* d_result,
                                 alz_encode_aux* d_aux, void* d_scratch, uint32_t* d_ctl, const void* geom) {
    EncGeom g; memcpy(&g, geom, sizeof(g));
    g.b_cap = ALZ_LEN_CAP;                                        // (not choose_b_cap's: here EVERY capped position somebody may land on is searched, generation by generation -- Yaz0 at quality 0 0.17 -> 1.5 ms with the cap at 48, quality 12 1.1 -> 500)
    const int tail = fmt == ALZ_FMT_LZ4_BLOCK ? 5 : 0;
    BencLayout L(*st, g, tail);
    L.ctl_dev = d_ctl;
    u8* base = (u8*)d_scratch;
    BencArgs a = L.a; a.data = (const u8*)d_src_base + st->src_off;
    alz_stream* vs = (alz_stream*)(base + L.vs); u32* vindex = (u32*)(base + L.vindex); u64* vpos = (u64*)(base + L.vpos);
    int* seg4 = (int*)(base + L.seg4); int* segm = g.use_min_table ? (int*)(base + L.segm) : nullptr;
    int* fin4 = (int*)(base + L.fin4); int* finm = g.use_min_table ? (int*)(base + L.finm) : nullptr;
    mentry* match = (mentry*)(base + L.match);
    u32* ml = (u32*)(base + L.ml); u32* md = (u32*)(base + L.md); u32* jump_a = (u32*)(base + L.jump_a); u32* jump_b = (u32*)(base + L.jump_b);
    u8* mark = base + L.mark; u8* sr = base + L.sr; u32* ctl = d_ctl;
    hipError_t e;
    hipLaunchKernelGGL(benc_setup, dim3((a.K + 256u) / 256u), dim3(256), 0, stream, *st, a, vs, vindex, vpos, ctl, d_result, d_aux);
    // A': kernel A on the segments, then the links to where kernel B reads them
    e = launch_prev(stream, (const u8*)d_src_base, vs, vindex, a.K, seg4, segm, vpos, g, 0, true);        // (a workgroup per segment AND pass: 1.24 -> 0.3 ms for an LZ4 block at quality 8)
    if (e != hipSuccess) return e;
    const u32 nbp = ((u32)a.limit + 256u) / 256u;
    if (g.link16) hipLaunchKernelGGL((benc_gather<true>), dim3(nbp), dim3(256), 0, stream, a, seg4, segm, fin4, finm, mark);
    else hipLaunchKernelGGL((benc_gather<false>), dim3(nbp), dim3(256), 0, stream, a, seg4, segm, fin4, finm, mark);
    // B: on the real stream (entry K)
    launch_match(stream, (const u8*)d_src_base, vs, vindex + a.K, 1u, st->src_len, fin4, finm, match, vpos, g, tail, 4096u,
                 false);                                           // (the two-phase kernel here too: tools/variants/r04_encode_switches.patch, -DALZ_BENC_DENSE)
    // C: the parse
    const u32 nbn = (a.nodes + 255u) / 256u;
    const bool caps = g.max_len > g.b_cap;                        // (only then can kernel B have capped anything)
    u32* stt = caps ? jump_a : nullptr;                           // (the jump tables are free until the ranking; the token-bit area until the emission)
    u32* front0 = jump_b; u32* front1 = (u32*)(base + L.front1);
    if (caps) {
        hipLaunchKernelGGL(benc_unpack, dim3(nbp), dim3(256), 0, stream, a, match, ml, md, stt, ctl);
        hipLaunchKernelGGL(benc_request0, dim3(nbp), dim3(256), 0, stream, a, g, ml, stt, front0, ctl);
        if (g.use_min_table) {
            hipLaunchKernelGGL((benc_exact_g1<true>), dim3(4096), dim3(64), 0, stream, a, g, fin4, finm, ml, md, stt, front0, front1, ctl);
            hipLaunchKernelGGL((benc_exact_rest<true>), dim3(1), dim3(1024), 0, stream, a, g, fin4, finm, ml, md, stt, front1, ctl);
        } else {
            hipLaunchKernelGGL((benc_exact_g1<false>), dim3(4096), dim3(64), 0, stream, a, g, fin4, finm, ml, md, stt, front0, front1, ctl);
            hipLaunchKernelGGL((benc_exact_rest<false>), dim3(1), dim3(1024), 0, stream, a, g, fin4, finm, ml, md, stt, front1, ctl);
        }
    }
    const u32 rtiles = (a.nodes + BENC_RTILE - 1u) / BENC_RTILE;
    u32* next1 = (u32*)(base + L.next1);
    if (caps) hipLaunchKernelGGL((benc_tile_exit<false>), dim3(rtiles), dim3(1024), 0, stream, a, g, match, ml, md, next1, jump_a, sr);
    else hipLaunchKernelGGL((benc_tile_exit<true>), dim3(rtiles), dim3(1024), 0, stream, a, g, match, ml, md, next1, jump_a, sr);
    const u32 rr = (benc_rounds(rtiles + 1u) + 1u) / 2u;           // (a launch is two rounds)
    for (u32 r = 0; r < rr; r++) { hipLaunchKernelGGL(benc_rank_round, dim3(nbn), dim3(256), 0, stream, jump_a, jump_b, mark, a.nodes); u32* t = jump_a; jump_a = jump_b; jump_b = t; }
    hipLaunchKernelGGL(benc_tile_mark, dim3(rtiles), dim3(1024), 0, stream, a, next1, sr, ml, mark, ctl);
    // D: the tokens
    u8* dst = (u8*)d_dst_base + st->dst_off;
    switch (fmt) {
    case ALZ_FMT_LZSS: benc_emit<ALZ_FMT_LZSS>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_LZ10: benc_emit<ALZ_FMT_LZ10>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_LZ11: benc_emit<ALZ_FMT_LZ11>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_LZ40: benc_emit<ALZ_FMT_LZ40>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_YAZ0: benc_emit<ALZ_FMT_YAZ0>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_YAY0: benc_emit<ALZ_FMT_YAY0>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_MIO0: benc_emit<ALZ_FMT_MIO0>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_CLZ0: benc_emit<ALZ_FMT_CLZ0>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_BLZ: benc_emit<ALZ_FMT_BLZ>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_LZHUDSON: benc_emit<ALZ_FMT_LZHUDSON>(stream, L, a, g, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_LZ4_BLOCK: benc_emit_seq<ALZ_FMT_LZ4_BLOCK>(stream, L, a, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_SNAPPY_RAW: benc_emit_seq<ALZ_FMT_SNAPPY_RAW>(stream, L, a, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_LZO: benc_emit_lzo(stream, L, a, base, dst, st->dst_cap, d_result, d_aux); break;
    case ALZ_FMT_PRS_BE: e = benc_emit_prs<true>(stream, L, a, base, dst, st->dst_cap, d_result, d_aux); if (e != hipSuccess) return e; break;
    case ALZ_FMT_PRS_LE: e = benc_emit_prs<false>(stream, L, a, base, dst, st->dst_cap, d_result, d_aux); if (e != hipSuccess) return e; break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
