// alz_encode.hip -- batched GPU encoder, bit-identical to the reference's greedy/lazy hash-chain encoder.
//
// Replaces MatchFinder/LzChainMatchFinder.cs:13-372 + IO/FlagWriter.cs:13-147 + the CompressHeaderless bodies
// (LZSS.cs:132-160, LZ10.cs:113-137, LZ11.cs:135-171, Yay0.cs:152-184, MIO0.cs:159-184, PRS.cs:104-159, LZ4.cs:202-238,
// LZO.cs:141-250, Snappy.cs:130-203).
//
// The reference finder looks strictly sequential (its hash tables mutate per position), but it has a property that
// makes it data-parallel: EVERY position below the cursor has been inserted exactly once, in increasing order, by the
// time a position is searched (searched positions are inserted by MatchSearch :245, skipped ones by the fill loop
// :199-203).  Hence
//     head[h] at the moment position p is searched == the largest q < p with hash(q) == hash(p)   =: prev(p)
//     chain[q & mask]                                == prev(q)   (slots are only reused beyond maxDistance, where the
//                                                                  chain walk has already stopped :259-260)
// so MatchSearch(p) is a PURE function of the data, and the parse (FindNextBestMatch :157-212) only consumes it.
// Three stages:
//   A  enc_prev_cu_kernel       prev(p) for the 4-byte hash (and the min-length hash when quality >= 10): one workgroup of 16
//                               wavefronts per stream, the head table in LDS.  Matches that reach back at most 8 KiB (every
//                               format but LZ4 / LZO / Snappy / FastLZ / RefPack ...): ONE pass whatever the hash width, through a
//                               14-bit table and a ring of tags and skip links; otherwise 2^(hashBits - 15) passes.  (The earlier
//                               forms -- head tables in HBM, a counting sort, hash-partitioned wavefronts, one wavefront per
//                               stream on LDS -- are in the history of this file and in docs/EXPERIMENTS.md 4.5 with their numbers.)
//   B  enc_match_kernel         one lane per position: the chain walk of MatchSearch/ChainMatches (:214-282) over prev(),
//                               embarrassingly parallel; writes (distance, length) per position.  From maxChain 3 on
//                               enc_match_dense_kernel: the chains walked first into an LDS list, the pairs compared 64 at a time.
//   C  the greedy/lazy parse as a walk over 64-position windows and the emission of a window's tokens by prefix sums, in ONE kernel from
//                               the same registers (one wavefront per stream): enc_parse_emit_kernel (the flag-bit formats: LZSS, LZ10, LZ11,
//                               Yaz0, Yay0, MIO0 ...), enc_parse_seq_kernel (LZ4, Snappy), enc_emit_prs_kernel and enc_parse_lzo_kernel,
//                               the last three on struct WinParse.  At quality 0 (one candidate per position) kernel B is inside them too.
//      enc_roles_kernel         the walk alone: a bit per token start, in front of enc_emit_kernel: emission on one lane per stream, for the
//                               formats that have no parallel emit yet.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "alz_device.h"
#include "alz_internal.h"

namespace {

struct EncGeom {           // per-format LzProperties + finder parameters
    int min_len, max_len, min_dist, max_dist;
    int max_chain, lazy, hash_bits, use_min_table, no_self_overlap;
    u32 min_mask;
    // LZSS
    u32 length_bits, lz_min_length, windows_start, lz_max_distance;
    // the LzProperties[] form of the finder (RefPack): ScoreMatch takes the first set that admits a candidate
    int nprops, p_max_len[3], p_min_len[3], p_max_dist[3], p_min_dist[3];
    int variant;           // FastLZ: 1 = level 2 (token format + the two property sets)
    int link16;            // the 4-byte-hash links are 16-bit DISTANCES (0: none or out of reach): every finder whose maxDistance fits
    int b_cap;             // kernel B stops comparing a candidate here and marks the position (the parse searches it exactly if its cursor ever stands on it)
};

// prev() of a position from the link array kernel A wrote: a position (or -1), or -- L16, maxDistance <= 65 535 -- a 16-bit distance,
// 0 for "none".  A link longer than maxDistance ends a chain walk exactly as no link does (ChainMatches :259-260), so the distances
// lose nothing, and the array is half the bytes for kernel A to write and kernel B to read (round 3).
template <bool L16>
__device__ __forceinline__ int link_at(const int* p4, int pos) {
    if (L16) { const u32 d = reinterpret_cast<const unsigned short*>(p4)[pos]; return d ? pos - (int)d : -1; }
    return p4[pos];
}

// (global memory takes unaligned dword / qword loads: one global_load_dword instead of four byte loads and three shifts)
__device__ __forceinline__ u32 load32(const u8* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ u64 load64(const u8* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }

__device__ __forceinline__ u32 scan_add(u32 v);
__device__ __forceinline__ u32 scan_max(u32 v);


// ---------------------------------------------------------------------------------------------- kernel A
// Kernel A with the head table in LDS: ONE workgroup of 16 wavefronts per stream, one stream per CU.  The table of 2^15 entries is
// 128 KB of the CU's 160; a finder with more hash bits takes 2^(hashBits - 15) passes over the stream, pass k owning the hashes whose
// top bits are k (entries of different hashes never meet, so the passes are independent).  Inside a pass the table is cut into 16
// classes (the top four bits of the 15-bit index), class c belonging to wavefront c alone: the positions of a chunk of 4 096 are
// hashed by all wavefronts together (256 each), ranked per class with ballots, and written in position order into 16 LDS queues;
// every owner then takes its queue 64 entries at a time -- old head, in-step duplicates (found with the table itself: write,
// read back, losers mark, winners see the mark), new head, prev() out to HBM -- with no barrier inside: an owner's table words,
// queue and step state are private to it, and the LDS executes one wavefront's operations in order.  Two workgroup barriers per
// chunk.  A class that would overflow its queue (runs of equal bytes: one hash, one class) makes the chunk go through one
// wavefront's 256 positions at a time.  prev() comes out exactly as from the table in HBM.
#ifndef ALZ_CU_FILL4
#define ALZ_CU_FILL4 2u       /* several passes: quarters of `stage` a slice fills on average */
#endif
// U: groups of 64 entries a wavefront brings to a chunk -- 2 for one pass (a chunk of 2 048 positions), 3 for several (`stage` holds 192
// entries per wavefront, a slice fills half of that on average: the synthetic streams overfilled a 128-entry stage at 3/4 in a third of
// their chunks)
// WIN (formats whose matches reach back at most 8 KiB: a finder with more than 15 hash bits would take 2^(hashBits - 15) passes): ONE
// pass whatever the hash width.  The table is indexed by the low 14 hash bits only, so an exchange hands a position the previous one
// of its table word -- a chain through ALL hashes that share the word -- and the position finds its prev() by walking that chain to
// the first entry whose remaining hash bits (the tag) equal its own.  Kernel B never follows a link beyond maxDistance (:259-260), so
// the walk ends there with "none", and only the last 16 Ki positions have to be remembered: a ring of one tag byte (written where the
// position is hashed) and one 16-bit link (written by the owner of the word) per position, in one dword -- 64 KB beside a 64 KB table.  Inside
// 8 KiB a 14-bit word is shared by half a random position on average, so the walk is short; what the passes cost -- every pass hashes
// the whole stream again, 67 of 138 ms at quality 8 -- is gone.  Entries wait for their step at most DRW positions, so that no ring
// slot is overwritten (by the position 16 Ki further on) while a walk may still read it.
template <int U, bool WIN>
__global__ __launch_bounds__(1024) void enc_prev_cu_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams,
                                                           const u32* __restrict__ index_list, u32 count, int* __restrict__ prev4,
                                                           int* __restrict__ prevm, const u64* __restrict__ pos_off, EncGeom g, int tail_skip) {
    constexpr u32 ALZ_CU_QCAP = U == 2 ? 352u : 288u;          // (LDS: the table, the queues, the staging rows -- 160 KB)
    constexpr u32 TB = WIN ? 14u : 15u;           // table bits
    constexpr u32 RING = 16384u, RM = RING - 1u;  // WIN: positions remembered
    __shared__ int T[(1 << TB) + 64];
    __shared__ u32 Q[16][ALZ_CU_QCAP];
    __shared__ u32 stage[16][WIN ? 1 : U * 64];   // several passes: a wavefront's entries of this pass, gathered from its slice
    // WIN: per position, ONE dword -- byte 0 the hash bits above the table index (the tag), the upper half the distance to the previous
    // entry of the same table word that has ANOTHER tag (0: none).  (As two arrays the walk below read the tag, waited, and only then
    // read the link of the lanes that needed it: the compiler sinks the second read into the branch -- two LDS round trips per hop.)
    __shared__ u32 ring[WIN ? RING : 1u];
    u8* const ringb = reinterpret_cast<u8*>(ring);
    unsigned short* const ringh = reinterpret_cast<unsigned short*>(ring);
    __shared__ u32 cnts[16][16];                  // [wavefront][class]: entries of the current chunk
    __shared__ u32 qpub[32];                      // [class]: ring index behind the queue's last entry; [16 + class]: entries waiting
    __shared__ u32 spill[3];                      // a slice held more entries of this pass than `stage` takes (one flag per call, three in rotation)
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)(threadIdx.x & 63u);
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 sid = index_list[bid];
    if (sid == 0xFFFFFFFFu) return;               // (a list written on the device, enc_words_kernel: unused slots)
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const int n = (int)st.src_len - tail_skip;
    const int limit = n - 4;
    int* const p4a = prev4 + pos_off[sid];
    int* const pma = g.use_min_table ? prevm + pos_off[sid] : nullptr;
    // (volatile, and in the LDS address space -- a generic volatile pointer turns into flat_load / flat_store sc0 sc1 with a full
    // s_waitcnt vmcnt(0) each: the steps below write a word and read it back to see the other lanes' writes)
    typedef __attribute__((address_space(3))) volatile int lds_vint;
    lds_vint* Tv = (lds_vint*)T;
    const u32 hb = (u32)g.hash_bits, hmask = (1u << hb) - 1u;
    // passes: 2^(hashBits - 15) for the 4-byte hash, and for finders with the min-length table (quality >= 10: keyed by another hash,
    // 16 bits, links into their own array) two more
    const u32 npass4 = WIN ? 1u : 1u << (hb - 15u), npassm = g.use_min_table ? (WIN ? 1u : 2u) : 0u;
    const u64 lanes_below = (1ull << lane) - 1ull;
    constexpr int CH = 1024 * U;
    // several passes: a wavefront looks at SB groups of 64 positions per chunk and keeps what belongs to the pass -- 3/4 of what
    // `stage` takes on average
    static_assert(U * ALZ_CU_FILL4 <= 6, "a chunk of several passes must stay below 32 Ki positions");
    if (threadIdx.x < 3u) spill[threadIdx.x] = 0;
    // (gridDim.y > 1: a workgroup per (stream, pass) -- the passes are independent; what the whole-GPU path of ONE stream launches, whose
    // "streams" are a few dozen overlapping segments: alz_encode_big.h)
    const u32 pfirst = gridDim.y > 1u ? blockIdx.y : 0u, plast = gridDim.y > 1u ? pfirst + 1u : npass4 + npassm;
    if (pfirst >= npass4 + npassm) return;
    for (u32 pass_all = pfirst; pass_all < plast; pass_all++) {
        const bool mt = pass_all >= npass4;       // a pass of the min-length table
        const u32 pass = mt ? pass_all - npass4 : pass_all;
        const u32 npass = WIN ? 1u : (mt ? 2u : npass4);       // passes of this pass's table
        int* const p4 = mt ? pma : p4a;           // where this pass's links go
        const u32 vmask = mt ? g.min_mask : 0xFFFFFFFFu, hshift = mt ? 16u : 32u - hb, hmask2 = mt ? 0xFFFFu : hmask;
        int SB = npass == 1u ? U : (int)(npass * (u32)U * ALZ_CU_FILL4 / 4u);
        if (SB > 24) SB = 24;                     // (a slice stays below 32 Ki positions: 32 passes fill their staging rows a quarter)
        const int CHM = 1024 * SB;
        for (u32 i = threadIdx.x; i < (1u << TB); i += 1024u) T[i] = -1;
        u32 qhead = 0, qn = 0;                    // the queue of class w (this wavefront's)
        if (lane == 0) { qpub[w] = 0; qpub[16 + w] = 0; }
        __syncthreads();
        int since = 0;                            // positions since the queues were last emptied (entries keep 16 bits of theirs)

        // One group of 64 positions at `pos`: the entry of my position, or none.
        // (`keepm`: the lanes that keep their entry, as a mask built from the ballots of the single compares -- the ballot of their AND goes
        // through a 0 / 1 register and a compare, twice 24 times per slice in scan() below)
        auto entry_of = [&](int pos, u32 v, bool direct, u32& e, int hi = 0x7FFFFFFF, u64* keepm = nullptr) -> bool {
            const bool act = pos <= limit && pos < hi;             // (`hi`: a multiple of 64 -- whole groups in or out)
            const u32 h = (((v & vmask) * 2654435761u) >> hshift) & hmask2;      // ComputeHash :288-299 / the min-length table's :226-243 (one multiply either way)
            bool keep = act && (WIN || (h >> 15) == pass);
            if (keepm) *keepm = __ballot(pos <= limit) & (WIN ? ~0ull : __ballot((h >> 15) == pass));
            if (WIN) { if (act) ringb[4u * ((u32)pos & RM)] = (u8)(h >> TB); }
            u32 wonly = 0;
            if (direct) {
                // Runs (one byte, one pixel repeated) give every position the hash of a neighbour, all of them in one class.  A position
                // whose hash also belongs to one of the four lanes below it (same row of 16) has its prev() right there; if one of the
                // four lanes above has it too, nobody ever reads what it would write into the table: it stays out of the queue.
                // Without such a lane above it only writes (flag bit 16), without one below it is an ordinary entry.
                const u32 hk = keep ? h + 1u : 0x80000000u | (u32)lane;      // (never 0 = what a shift brings in from outside the row, never equal to a real hash or to a neighbour's)
                const u64 m1 = __ballot(hk == (u32)__builtin_amdgcn_update_dpp(0, (int)hk, 0x111, 0xF, 0xF, true));
                const u64 m2 = __ballot(hk == (u32)__builtin_amdgcn_update_dpp(0, (int)hk, 0x112, 0xF, 0xF, true));
                const u64 m3 = __ballot(hk == (u32)__builtin_amdgcn_update_dpp(0, (int)hk, 0x113, 0xF, 0xF, true));
                const u64 m4 = __ballot(hk == (u32)__builtin_amdgcn_update_dpp(0, (int)hk, 0x114, 0xF, 0xF, true));
                const u64 hasp = m1 | m2 | m3 | m4;
                if (hasp) {
                    const u64 hass = (m1 >> 1) | (m2 >> 2) | (m3 >> 3) | (m4 >> 4);
                    const u64 me = 1ull << lane;
                    if (hasp & me) {
                        const int d = (m1 & me) ? 1 : (m2 & me) ? 2 : (m3 & me) ? 3 : 4;
                        if (!mt && g.link16) reinterpret_cast<unsigned short*>(p4)[pos] = (unsigned short)d; else p4[pos] = pos - d;
                        if (hass & me) keep = false; else wonly = 1u << 16;
                    }
                }
            }
            e = ((u32)pos & 0xFFFFu) | wonly | (WIN ? (h & 0x3FFFu) << 18 : (h & 0x7FFFu) << 17);
            return keep;
        };

        // my class: full steps of 64 (everything when `all`), then tell the others where my queue stands.  (Tried: at most 2-3 steps per
        // chunk, so that a burst in one class spreads over the chunks behind it, and a catch-up round before the 16-round fallback:
        // no gain at 1, 4 or 16 passes -- what the owners wait for is not a burst.)
        auto steps = [&](bool all, int cend1) {
            while (qn >= 64u || (all && qn)) {
                const u32 nstep = qn < 64u ? qn : 64u;
                const bool actl = (u32)lane < nstep;
                u32 slot = qhead + (u32)lane; if (slot >= ALZ_CU_QCAP) slot -= ALZ_CU_QCAP;
                const u32 e = Q[w][slot];
                const int pos = cend1 - (int)(((u32)cend1 - e) & 0xFFFFu);
                const u32 idx = actl ? e >> (WIN ? 18 : 17) : (1u << TB) + (u32)lane;        // (idle lanes: a private word behind the table, no exec masks below)
                // ONE exchange per lane: my position in, the word's previous content out.  Lanes of one step that share a word are
                // served one after the other; served in lane order (= position order) each of them gets exactly its prev() -- the
                // head from before the step for the first, the lane before for the others -- and the last one leaves the new head.
                // Any other order hands some lane a position BEHIND its own (a chain of increasing positions is the lane order),
                // so `got > pos` anywhere in the step proves it; only then are those groups ordered by hand: the old head is the
                // smallest value the group got back (it lies before every position of the step), the rest follows from the lanes.
                const int got = __hip_atomic_exchange(&T[idx], pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                int prev = got;
                const u64 bad = __ballot(actl && got > pos);
                if (bad) {
                    u64 mygrp = 0; int ghead = 0;
                    u64 todo = bad;
                    while (todo) {
                        const int l0 = (int)__builtin_ctzll(todo);
                        const u32 iv = (u32)__builtin_amdgcn_readlane((int)idx, l0);
                        const u64 grp = __ballot(idx == iv);
                        int mn = 0x7FFFFFFF;
                        for (u64 g2 = grp; g2; g2 &= g2 - 1ull) { const int r = __builtin_amdgcn_readlane(got, (int)__builtin_ctzll(g2)); mn = r < mn ? r : mn; }
                        if (idx == iv) { mygrp = grp; ghead = mn; }
                        todo &= ~grp;
                    }
                    const u64 below = mygrp & lanes_below;
                    const int from = below ? 63 - (int)__builtin_clzll(below) : lane;
                    const int pp = __builtin_amdgcn_ds_bpermute(from << 2, pos);       // position of the next lower lane of my group
                    if (mygrp) {
                        prev = below ? pp : ghead;
                        if ((mygrp >> lane) <= 1ull) Tv[idx] = pos;                      // the highest lane of a group leaves the new head
                    }
                }
                if (WIN) {
                    // `prev` = the previous position of my table word, whatever its hash.  The ring keeps per position its tag and ONE link:
                    // the previous entry of its word with ANOTHER tag (what lies between has the position's own tag, so a walker that
                    // stands on the position with a different tag may skip it: a run of one 4-byte pattern -- dozens of entries in one
                    // word -- costs a walker one hop, not dozens).  My own link follows from prev's tag and link; then I walk: prev has
                    // my tag (two positions in three), or I hop from tag run to tag run until one has it or maxDistance is behind me.
                    // Lanes of this step that share a word see each other's links: reads of a round precede its writes (one
                    // wavefront's LDS operations execute in order), and any link written so far is a valid, if shorter, skip.
                    const u32 me = (u32)pos & RM;
                    const u32 mytag = ring[me] & 0xFFu;
                    const int p0 = __builtin_amdgcn_readfirstlane(pos);        // candidates >= p0 are lanes of this step
                    const bool has = actl && prev >= 0 && (u32)(pos - prev) <= 0xFFFFu;
                    const u32 cs = (u32)(has ? prev : pos) & RM;
                    const u32 e0 = ring[cs], t0 = e0 & 0xFFu, l0 = e0 >> 16;
                    const bool same = has && t0 == mytag, instep = has && prev >= p0;
                    u32 mylink = 0;
                    if (has) {
                        if (!same || instep) mylink = (u32)(pos - prev);      // (same tag inside the step: provisional, refined below)
                        else { const u32 far = (u32)(pos - prev) + l0; mylink = (l0 != 0u && far <= 0xFFFFu) ? far : 0u; }
                    }
                    if (actl) ringh[2u * me + 1u] = (unsigned short)mylink;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    bool chase = same && instep;                               // a chain of my tag inside the step: pointer jumping over it
                    while (__ballot(chase)) {
                        const int x = pos - (int)mylink;
                        const u32 lx = ring[(u32)(chase ? x : pos) & RM] >> 16;
                        const int y = x - (int)lx;
                        const u32 ty = ring[(u32)((chase && lx != 0u) ? y : pos) & RM] & 0xFFu;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                        if (chase) {
                            const u32 far = mylink + lx;
                            const bool ok = lx != 0u && far <= 0xFFFFu;
                            mylink = ok ? far : 0u;
                            chase = ok && y >= p0 && ty == mytag;
                            ringh[2u * me + 1u] = (unsigned short)mylink;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    }
                    const bool need = actl && !(e & 0x10000u);
                    const bool inwin = has && pos - prev <= g.max_dist;
                    int res = (same && inwin) ? prev : -1;
                    int cand = prev;
                    bool go = need && inwin && !same;
                    if (go && !instep) {                                       // prev's link is final unless prev is a lane of this step
                        if (l0 == 0u) go = false; else { cand -= (int)l0; go = pos - cand <= g.max_dist; }
                    }
                    while (__ballot(go)) {
                        const u32 cs2 = (u32)(go ? cand : pos) & RM;
                        const u32 e2 = ring[cs2], t = e2 & 0xFFu, dl = e2 >> 16;
                        if (go) {
                            if (t == mytag) { res = cand; go = false; }
                            else if (dl == 0u) go = false;
                            else { cand -= (int)dl; go = pos - cand <= g.max_dist; }
                        }
                    }
                    prev = res;
                }
                if (actl && !(e & 0x10000u)) {
                    if (!mt && g.link16) { const u32 dd = prev < 0 ? 0u : (u32)(pos - prev); reinterpret_cast<unsigned short*>(p4)[pos] = (unsigned short)(dd > 0xFFFFu ? 0u : dd); }
                    else p4[pos] = prev;
                }
                qhead += nstep; if (qhead >= ALZ_CU_QCAP) qhead -= ALZ_CU_QCAP;
                qn -= nstep;
            }
            if (lane == 0) { u32 t = qhead + qn; if (t >= ALZ_CU_QCAP) t -= ALZ_CU_QCAP; qpub[w] = t; qpub[16 + w] = qn; }
        };
        // The steps of a chunk wait until the next chunk's entries are being found: half of the wavefronts (two of the four on every
        // SIMD) take them first and hash afterwards, the other half hash and rank first -- the steps are chains of LDS round trips, the
        // hashing and ranking is vector work, and both lie between the same two barriers.
        bool pend = false, pend_all = false; int pend_cend1 = 0;
        const bool late = ((w >> 2) & 1u) != 0u;
        auto run_pending = [&]() { if (pend) { steps(pend_all, pend_cend1); pend = false; } };
        // The rest of a chunk, from the (at most U x 64) entries a wavefront found in it: ranks per class, the queues, the steps.
        // `par` >= 0: give up (false) behind the first barrier if some wavefront could not hold its entries.
        auto finish = [&](const u32 (&ent)[U], const bool (&keep)[U], int cend, int clen, int par) -> bool {
            u32 cls[U], rank[U], prior[U];
            if (lane < 16) cnts[w][lane] = 0;     // my row: entries of class c among mine (the last lane of a class keeps it up to date)
#pragma unroll
            for (int u = 0; u < U; u++) {
                cls[u] = keep[u] ? ent[u] >> 28 : 16u;
                // the lanes of a class = AND over the four bit planes of the class number (plane or its complement): every lane
                // forms the mask of its own class, without a loop over classes; its rank is the count of lower lanes in it
                const u64 valid = __ballot(keep[u]);
                if (valid == 0ull) { rank[u] = 0; prior[u] = 0; continue; }     // (several passes: the third group of a slice is empty more often than not)
                u64 pl[4];
#pragma unroll
                for (int b = 0; b < 4; b++) pl[b] = __ballot((cls[u] >> b) & 1u);
                u64 mine = valid;
#pragma unroll
                for (int b = 0; b < 4; b++) { const u64 s1 = (u64)0 - (u64)((cls[u] >> b) & 1u); mine &= ~(pl[b] ^ s1); }
                rank[u] = __builtin_amdgcn_mbcnt_hi((u32)(mine >> 32), __builtin_amdgcn_mbcnt_lo((u32)mine, 0u));
                prior[u] = u == 0 ? 0u : cnts[w][cls[u] & 15u];            // what my earlier groups brought to my class
                if (keep[u] && (mine >> lane) <= 1ull) cnts[w][cls[u]] = prior[u] + rank[u] + 1u;
            }
            run_pending();                        // (the late half; the early half has none left here)
            __syncthreads();
            if (par >= 0 && spill[par]) return false;
            // ---- where my entries go: lanes 0..15 hold, for class = lane, the entries of the wavefronts before mine and of all
            u32 before = 0, tot = 0, tail_c = 0, wait_c = 0;
            if (lane < 16) {
#pragma unroll
                for (u32 ww = 0; ww < 16; ww++) { const u32 x = cnts[ww][lane]; tot += x; if (ww < w) before += x; }
                tail_c = qpub[lane]; wait_c = qpub[16 + lane];
            }
            const bool last = cend > limit;
            // (nothing waits longer than 32 Ki positions + a chunk: 16 bits tell where it was.  WIN: no longer than the rings allow --
            // a slot is reused 16 Ki positions on, walks reach back maxDistance, steps run one chunk behind the hashing)
            const int drw = WIN ? ((int)(RING / (u32)CH) - (g.max_dist + CH - 1) / CH - 1) * CH : 32768;
            const bool drain = last || since + clen >= drw;
            since = drain ? 0 : since + clen;
            const int cend1 = cend - 1;
            const bool narrow = __ballot(lane < 16 && wait_c + tot > ALZ_CU_QCAP) != 0ull;   // (the same answer in every wavefront)
            const u32 rounds = narrow ? 16u : 1u;
            for (u32 r = 0; r < rounds; r++) {
                if (narrow) {
                    if (r) { __syncthreads(); if (lane < 16) { tail_c = qpub[lane]; } }
                    before = 0; tot = lane < 16 ? cnts[r][lane] : 0u;
                }
                if (!narrow || r == w) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const u32 b = (u32)__builtin_amdgcn_ds_bpermute((int)((cls[u] & 15u) << 2), (int)(tail_c + before)) + prior[u];
                        if (cls[u] < 16u) {
                            u32 slot = b + rank[u];
                            while (slot >= ALZ_CU_QCAP) slot -= ALZ_CU_QCAP;
                            Q[cls[u]][slot] = ent[u];
                        }
                    }
                }
                __syncthreads();
                // ---- my class: full steps of 64 (everything when the queue has to drain)
                qn += (u32)__builtin_amdgcn_readlane((int)tot, (int)w);
                if (narrow) steps(drain && r == 15u, cend1);
                else if (npass != 1u) steps(drain, cend1);             // (several passes: waiting gains nothing, measured)
                else { pend = true; pend_all = drain; pend_cend1 = cend1; }
            }
            return true;
        };
        if (WIN || npass == 1u) {
            u32 vnext[U];                         // the dwords of the next chunk's positions, loaded one chunk ahead
#pragma unroll
            for (int u = 0; u < U; u++) { const int pos = (int)((w * U + (u32)u) * 64u) + lane; vnext[u] = load32(data + (pos < limit ? pos : (limit > 0 ? limit : 0))); }
            for (int cbase = 0; cbase <= limit; cbase += CH) {
                if (!late) run_pending();
                u32 ent[U]; bool keep[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int pos = cbase + (int)((w * U + (u32)u) * 64u) + lane;
                    const u32 v = vnext[u];
                    vnext[u] = load32(data + (pos + CH < limit ? pos + CH : limit));
                    keep[u] = entry_of(pos, v, true, ent[u]);
                }
                (void)finish(ent, keep, cbase + CH, CH, -1);
            }
        } else if constexpr (!WIN) {
            // Several passes.  A piece of work is `sbn` groups of 64 positions per wavefront starting at `from`: every wavefront keeps what
            // its slice holds of this pass (through `stage`, in position order, at most U x 64 entries).  If some slice held more, nothing
            // is done and the piece is cut in two; a half that fails again goes position by position (1 024 U at a time, every entry
            // straight from its position: a hash all over the chunk).  One loop, one call of finish(): the pieces wait on a small stack.
            auto scan = [&](int from, auto sbc) -> u32 {
                constexpr int SBN = decltype(sbc)::value;
                const int p0 = from + (int)(w * (u32)SBN * 64u) + lane;
                u32 vv[SBN];                      // the whole slice in flight at once; no branch around a load (behind one, each waits for the one before)
#pragma unroll
                for (int sb = 0; sb < SBN; sb++) { const int q = p0 + sb * 64; vv[sb] = load32(data + (q < limit ? q : limit)); }
                u32 fill = 0;
#pragma unroll
                for (int sb = 0; sb < SBN; sb++) {
                    u32 e; u64 m; const bool kp = entry_of(p0 + sb * 64, vv[sb], false, e, 0x7FFFFFFF, &m);
                    const u32 at = fill + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                    if (kp && at < (u32)(U * 64)) stage[w][at] = e;
                    fill += (u32)__popcll(m);
                }
                return fill;
            };
            u32 par = 0;
            int nxt = 0;                          // the next chunk starts here
            int pf[3], ps[3], pl[3], np = 0;      // pieces waiting: from, groups of 64 per wavefront, level (0 chunk, 1 half, 2 position by position)
            for (;;) {
                if (np == 0) {
                    if (nxt > limit) break;
                    pf[0] = nxt; ps[0] = SB; pl[0] = 0; np = 1; nxt += CHM;
                }
                np--;
                const int from = np == 0 ? pf[0] : np == 1 ? pf[1] : pf[2];
                const int sbn = np == 0 ? ps[0] : np == 1 ? ps[1] : ps[2];
                const int lvl = np == 0 ? pl[0] : np == 1 ? pl[1] : pl[2];
                if (from > limit) continue;
                if (!late) run_pending();
                u32 ent[U]; bool keep[U];
                int cend, flag = -1;
                if (lvl < 2) {
                    par = par == 2u ? 0u : par + 1u;
                    if (threadIdx.x == 0) spill[par == 2u ? 0u : par + 1u] = 0;      // (the flag of the call after this one; last read two calls ago)
                    u32 fill = 0;
                    switch (sbn) {
                    case 24: fill = scan(from, std::integral_constant<int, 24>{}); break;
                    case 12: fill = scan(from, std::integral_constant<int, 12>{}); break;
                    case 6: fill = scan(from, std::integral_constant<int, 6>{}); break;
                    case 3: fill = scan(from, std::integral_constant<int, 3>{}); break;
                    case 2: fill = scan(from, std::integral_constant<int, 2>{}); break;
                    default: fill = scan(from, std::integral_constant<int, 1>{}); break;
                    }
                    const bool over = fill > (u32)(U * 64);
                    if (over && lane == 0) spill[par] = 1;
#pragma unroll
                    for (int u = 0; u < U; u++) { keep[u] = !over && 64u * (u32)u + (u32)lane < fill; ent[u] = stage[w][64 * u + lane]; }
                    cend = from + sbn * 1024; flag = (int)par;
                } else {
                    const int to = from + sbn * 1024;
                    cend = from + CH < to ? from + CH : to;                  // (`to` need not be a multiple of 1 024 U away)
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int pos = from + (int)((w * U + (u32)u) * 64u) + lane;
                        keep[u] = entry_of(pos, load32(data + (pos < limit ? pos : limit)), true, ent[u], cend);
                    }
                    if (cend < to) { pf[np] = cend; ps[np] = (to - cend) / 1024; pl[np] = 2; np++; }
                }
                if (!finish(ent, keep, cend, cend - from, flag)) {
                    const int sba = sbn / 2, sbb = sbn - sba;
                    const int nl = sba == 0 ? 2 : lvl + 1;
                    // (the second half first: the stack gives the first half back first)
                    if (sba) { pf[np] = from + sba * 1024; ps[np] = sbb; pl[np] = nl; np++; pf[np] = from; ps[np] = sba; pl[np] = nl; np++; }
                    else { pf[np] = from; ps[np] = sbn; pl[np] = 2; np++; }
                }
            }
        }
        run_pending();
        __syncthreads();
    }
}

#ifndef ALZ_LEN_CAP
#define ALZ_LEN_CAP 2040
#endif
// The search inside the parse + emit kernel (quality 0) compares the 32 bytes it has prefetched and no more: a position whose candidate still
// matches there is "capped", and searched exactly (by the whole wavefront, benc_wave_search) only if the cursor ever stands on it.  With the
// cap at 2 040 every position inside a long match measured that match to its end -- in the runs and repeated rows of real data nearly all of
// them, for the few the parse visits: 4 096 windows of Test.bmp as Yaz0 at quality 0 51 -> 19.6 ms, as LZ11 77 -> 24.9; the synthetic batches
// (matches of at most 18 bytes) 42.5 ms either way.
#ifndef ALZ_PARSE_CAP
#define ALZ_PARSE_CAP 32
#endif
#ifndef ALZ_PARSE_CAP_HI
#define ALZ_PARSE_CAP_HI 128     /* ... and on up to this many while fewer than ALZ_PARSE_MANY lanes of the window are still equal (WinParse::matchof) */
#endif
#ifndef ALZ_PARSE_MANY
#define ALZ_PARSE_MANY 16
#endif
#define ALZ_CAPPED 0xFFFFFFFFu

// The match array kernel B hands to the parse and the emitters: ONE 32-bit entry per position (round 3; two words before) -- distance in
// the low 21 bits, length in the 11 above.  Kernel B compares at most ALZ_LEN_CAP = 2 040 bytes, so its lengths fit; the two length codes
// above them mean: ALZ_M_CAP -- kernel B ran into its cap here (the parse recomputes the position exactly if it ever visits it) --, and
// ALZ_M_LONG -- an exact length of 2 046 or more, written by the parse for a match it TOOK: the length itself is the next entry (that
// position lies inside the match: neither the parse nor an emitter ever looks at it as a position).  Finder geometries whose
// maxDistance does not fit 21 bits (FastLZ with MaxWindowBits above 20) are refused by alz_encode_geom.
typedef u32 mentry;
#define ALZ_M_DBITS 21u
#define ALZ_M_DMASK 0x1FFFFFu
#define ALZ_M_LONG 0x7FEu
#define ALZ_M_CAP 0x7FFu
__device__ __forceinline__ mentry m_pack(u32 d, u32 l) { return (l << ALZ_M_DBITS) | d; }
__device__ __forceinline__ uint2 m_unpack(mentry e) { const u32 l = e >> ALZ_M_DBITS; return make_uint2(e & ALZ_M_DMASK, l == ALZ_M_CAP ? ALZ_CAPPED : l); }
// GetMatchLength  LzChainMatchFinder.cs:338-357
__device__ __forceinline__ int match_len(const u8* a, const u8* b, int max) {
    int len = 0;
    while (len + 8 <= max) {
        const u64 x = load64(a + len) ^ load64(b + len);
        if (x) return len + (__builtin_ctzll(x) >> 3);
        len += 8;
    }
    if (len + 4 <= max) {
        const u32 x = load32(a + len) ^ load32(b + len);
        if (x) return len + (__builtin_ctz(x) >> 3);
        len += 4;
    }
    while (len < max && a[len] == b[len]) len++;
    return len;
}

// ScoreMatch  LzChainMatchFinder.cs:301-321
__device__ __forceinline__ int score_match(const EncGeom& g, int& len, int dist) {
    if (g.no_self_overlap && len > dist) len = dist;
    if (g.nprops <= 1) return len - g.min_len;
    for (int i = 0; i < g.nprops; i++) {
        if (dist <= g.p_max_dist[i] && len >= g.p_min_len[i] && dist >= g.p_min_dist[i]) {
            if (len > g.p_max_len[i]) len = g.p_max_len[i];
            return len - g.p_min_len[i];
        }
    }
    len = 0;
    return -1;
}

// GetMatchLength behind the first sixteen bytes, for the wavefront: eight bytes per trip for every lane that is still equal, no byte
// loop at the end (a trip reads up to seven bytes past the length that counts: inside the staging slack, clamped by the caller).
// As a per-lane loop with a byte tail, every trip of kernel B's loop paid the dependent byte loads of its slowest lane.
__device__ __forceinline__ int wave_match_tail(const u8* a, const u8* b, int max, bool go) {
    int l = 16;
    // (sixteen bytes per side and trip, both loads of a side in flight together: the loop runs for the whole wavefront as long as its longest
    // lane -- up to 32 trips of eight bytes for a Yaz0 match, 253 for a capped one -- and a trip is a memory round trip.  A trip may read up
    // to fifteen bytes past the length that counts.)
    while (__ballot(go)) {
        const int o = go ? l : 0;
        u64 va[2], vb[2];                                               // (ONE load per side: a scattered load costs the L1 a lookup per lane whatever its width, and the lookups are what bounds kernel B on real data)
        __builtin_memcpy(va, a + o, 16); __builtin_memcpy(vb, b + o, 16);
        const u64 z0 = va[0] ^ vb[0], z1 = va[1] ^ vb[1];
        if (go) {
            if (z0) { l += (int)(__builtin_ctzll(z0) >> 3); go = false; }
            else if (z1) { l += 8 + (int)(__builtin_ctzll(z1) >> 3); go = false; }
            else { l += 16; if (l >= max) go = false; }
        }
    }
    return l;
}

// MatchSearch :214-246 with ChainMatches :248-282 as a pure function of (data, prev); returns false when CAP > 0 and a
// candidate still matched after CAP bytes
template <bool MINT>
__device__ __forceinline__ bool match_search(const u8* data, int n, int pos, const int* p4, const int* pm, const EncGeom& g, int cap,
                                             int& best_d, int& best_l) {
    const u8* dp = data + pos;
    auto lk = [&](int q) { return g.link16 ? link_at<true>(p4, q) : link_at<false>(p4, q); };
    int cur = lk(pos);
    int best_possible = n - pos; if (best_possible > g.max_len) best_possible = g.max_len;
    const int cmp_max = (cap > 0 && best_possible > cap) ? cap : best_possible;
    best_d = 0; best_l = 0; int best_score = -1;
    int attempts = g.max_chain;
    while (cur != -1 && attempts-- > 0) {
        const int dist = pos - cur;
        if (dist > g.max_dist) break;
        if (dist < g.min_dist) { cur = lk(cur); continue; }
        int len = match_len(dp, data + cur, cmp_max);
        if (len == cmp_max && cmp_max < best_possible) return false;
        const int score = score_match(g, len, dist);
        if (score > best_score) { best_score = score; best_l = len; best_d = dist; if (best_l == best_possible) break; }
        cur = lk(cur);
    }
    if (MINT && best_l == 0) {                                          // small-match fallback :226-243
        const int c2 = pm[pos];
        if (c2 != -1) {
            int dist = pos - c2;
            if (dist < g.min_dist) dist = g.min_dist;
            if (dist <= g.max_dist && pos - dist >= 0) {
                int len = match_len(dp, data + pos - dist, cmp_max);
                if (len == cmp_max && cmp_max < best_possible) return false;
                (void)score_match(g, len, dist);
                best_l = len; best_d = dist;
            }
        }
    }
    return true;
}

// ---- the exact search by a whole wavefront (wave-uniform control flow): what the parse kernels run when their cursor meets a position
// kernel B (or the search inside the parse) had capped, and what the whole-GPU path of ONE stream runs for requested positions (alz_encode_big.h)
__device__ __forceinline__ u32 benc_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 benc_mbcnt(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
__device__ __forceinline__ u32 benc_last(u32 incl) { return (u32)__builtin_amdgcn_readlane((int)incl, 63); }

// GetMatchLength (LzChainMatchFinder.cs:338-357) by the whole wavefront: 4 KiB per trip -- four loads of sixteen bytes per lane and side in
// flight (a trip may read up to 63 bytes behind `max`: inside the slack behind every source buffer, never counted)
__device__ __forceinline__ int benc_wave_match_len(const u8* a, const u8* b, int max) {
    const int lane = (int)benc_lane();
    for (int base = 0; base < max; base += 4096) {
        u64 x[4][2];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int off = base + 1024 * k + 16 * lane;
            u64 va[2] = {0, 0}, vb[2] = {0, 0};
            if (off < max) { __builtin_memcpy(va, a + off, 16); __builtin_memcpy(vb, b + off, 16); }
            x[k][0] = va[0] ^ vb[0]; x[k][1] = va[1] ^ vb[1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u64 mm = __ballot((x[k][0] | x[k][1]) != 0ull);
            if (mm) {
                const int l0 = (int)__builtin_ctzll(mm);
                const u64 lo = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(x[k][0] >> 32), l0) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)x[k][0], l0);
                const u64 hi = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(x[k][1] >> 32), l0) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)x[k][1], l0);
                const int len = base + 1024 * k + 16 * l0 + (lo ? (int)(__builtin_ctzll(lo) >> 3) : 8 + (int)(__builtin_ctzll(hi) >> 3));
                return len < max ? len : max;
            }
        }
    }
    return max;
}

// MatchSearch (:214-246, ChainMatches :248-282) exactly, by the whole wavefront (wave-uniform control flow)
template <bool MINT>
__device__ __forceinline__ void benc_wave_search(const u8* data, int n, const EncGeom& g, const int* p4, const int* pm, int pos, int& best_d, int& best_l) {
    const u8* dp = data + pos;
    auto lk = [&](int q) { return g.link16 ? link_at<true>(p4, q) : link_at<false>(p4, q); };
    int cur = lk(pos);
    int best_possible = n - pos; if (best_possible > g.max_len) best_possible = g.max_len;
    best_d = 0; best_l = 0; int best_score = -1;
    int attempts = g.max_chain;
    while (cur != -1 && attempts-- > 0) {
        const int dist = pos - cur;
        if (dist > g.max_dist) break;
        if (dist < g.min_dist) { cur = lk(cur); continue; }
        // (a candidate wins only with a strictly higher score, i.e. a longer match -- ScoreMatch :301-321 with one property set is the length,
        // cut to the distance in CompatibilityMode --: one whose byte at offset best_l differs cannot be longer than best_l, and is not measured.
        // In the repeated rows of Test.bmp every candidate of a chain matches up to the same place, tens of KiB on: 1.4 -> 0.3 ms for an LZ4 block at Q8)
        if (g.nprops <= 1 && best_l > 0 && dp[best_l] != data[cur + best_l]) { cur = lk(cur); continue; }
        int len = benc_wave_match_len(dp, data + cur, best_possible);
        const int score = score_match(g, len, dist);
        if (score > best_score) { best_score = score; best_l = len; best_d = dist; if (best_l == best_possible) break; }
        cur = lk(cur);
    }
    if (MINT && best_l == 0) {                                          // small-match fallback :226-243
        const int c2 = pm[pos];
        if (c2 != -1) {
            int dist = pos - c2;
            if (dist < g.min_dist) dist = g.min_dist;
            if (dist <= g.max_dist && pos - dist >= 0) {
                int len = benc_wave_match_len(dp, data + pos - dist, best_possible);
                (void)score_match(g, len, dist);
                best_l = len; best_d = dist;
            }
        }
    }
}

// The exact matches FindNextBestMatch needs at a cursor q one of whose two entries is capped (:157-212): q's own -- searched only if ITS entry is the capped one (c0) -- and the lazy
// neighbour's, which the finder looks at only behind a match no longer than the lazy threshold (:180-190).  A capped entry is a match of the compare cap or more, far above that
// threshold: until round 6 both positions were searched whatever the first one found, and on flat data -- where the cursor lands on capped positions all the time, each search tens
// of KiB of compares -- half of the exact searches were for a neighbour nobody asked about.  e0 / e1: the two entries as the caller holds them (wave-uniform); s0 / s1: searched here.
template <bool MINT>
__device__ __forceinline__ void benc_capped_cursor(const u8* data, int n, const EncGeom& g, const int* p4, const int* pm, int q, int limit, uint2 e0, bool c0, uint2 e1, bool c1,
                                                   int& d0, int& l0, int& d1, int& l1, bool& s0, bool& s1) {
    s0 = c0;
    if (s0) benc_wave_search<MINT>(data, n, g, p4, pm, q, d0, l0); else { d0 = (int)e0.x; l0 = (int)e0.y; }
    d1 = 0; l1 = 0; s1 = false;
    if (l0 >= g.min_len && l0 <= g.lazy && q + 1 <= limit) {
        s1 = c1;
        if (s1) benc_wave_search<MINT>(data, n, g, p4, pm, q + 1, d1, l1); else { d1 = (int)e1.x; l1 = (int)e1.y; }
    }
}

// MatchSearch (:214-246, ChainMatches :248-282) exactly, by the whole wavefront, WITHOUT links (round 6): prev() chains are the positions below `pos` with pos's hash, nearest first,
// and the walk ends at the first one further back than maxDistance (:259-260) -- so the candidates are exactly the positions of [pos - maxDistance, pos) whose four bytes hash
// like pos's (ComputeHash :288-299: the top hashBits bits of one product), in descending order, at most maxChain of them.  The wavefront finds them by SCANNING the window
// behind the cursor, a block of 1 024 positions at a time, nearest block first (16 positions per lane: 20 bytes, one multiply per position), and measures the candidates of a
// block before it looks at the next: in a run or a stretch of repeated rows the nearest candidate already reaches the longest possible match and the walk ends in the first
// block.  No kernel A, no kernel B, no match array -- for the streams whose parse visits few positions (enc_scan_select_kernel): kernel B searches EVERY position, the
// managed parse only the ones its cursor stands on (FindNextBestMatch :157-212), and on the flat windows of Test.bmp that is one position in a hundred.
// One property set, no min-length table (quality < 10).
// A block's candidates go into a list (`cl`: 64 halfwords of LDS), nearest first, as many as there are attempts left, and ALL of them are measured at once: four lanes per
// candidate (two from 17 candidates on), 32 bytes per lane and round, a candidate drops out of the rounds at its first mismatch (a quad minimum over DPP).  ChainMatches
// keeps the FIRST candidate of the best score (a later one must be strictly better, :271-279) and ends at one that reaches the longest possible match -- nothing behind it could be
// strictly better --, so the result is the maximum over the list of (score, nearest first): one wave maximum.  (Measured one after the other, nearest first, with a byte test
// in front of each -- the managed order -- a flat window of Test.bmp cost 1 400 instructions and 37 loads per search, 8 us: its pixels repeat at distance 4, every search meets
// maxChain candidates, and in a gradient each of them is a little longer than the one before.)
// The listed candidates -- cl[j] = the distance of candidate j, nearest first, nc <= 32 of them -- measured at once and ChainMatches' choice among them folded into the
// best so far: GetMatchLength (:338-357) in rounds of 32 bytes per lane (a round is a memory round trip: 128 bytes per candidate with four lanes each), a candidate drops out
// at its first mismatch; the first candidate of the best score wins and must beat the best of the blocks before (strictly: :271-279).  True: that match is the longest possible.
__device__ __forceinline__ bool benc_scan_measure(const u8* data, const EncGeom& g, int pos, int best_possible, const unsigned short* cl, u32 nc, int lane, int& best_score, int& best_d, int& best_l) {
    const u8* dp = data + pos;
    const u32 gsh = nc <= 16u ? 2u : nc <= 32u ? 1u : 0u;             // lanes per candidate: 4 / 2 / 1
    const u32 j = (u32)lane >> gsh, t = (u32)lane & ((1u << gsh) - 1u);
    const int dist = j < nc ? (int)cl[j] : 0;
    const int c = pos - dist;
    const bool valid = j < nc && dist >= g.min_dist;                  // closer than minDistance: skipped, the attempt is spent  :262-266
    int len = best_possible;
    bool go = valid;
    const int stride = 32 << gsh;
    for (int off = 0; off < best_possible && __ballot(go); off += stride) {
        const int o = off + 32 * (int)t;
        const bool ld = go && o < best_possible;
        u64 va[4], vb[4];
        __builtin_memcpy(va, dp + (ld ? o : 0), 32); __builtin_memcpy(vb, data + c + (ld ? o : 0), 32);   // (up to 31 bytes behind best_possible: the slack behind every source buffer)
        const u64 x0 = va[0] ^ vb[0], x1 = va[1] ^ vb[1], x2 = va[2] ^ vb[2], x3 = va[3] ^ vb[3];
        // where my 32 bytes end the match (0x7FFFFFFF: they do not); lanes behind best_possible end it where they start
        u32 key = !ld ? (u32)o : x0 ? (u32)o + (u32)(__builtin_ctzll(x0) >> 3) : x1 ? (u32)o + 8u + (u32)(__builtin_ctzll(x1) >> 3)
                               : x2 ? (u32)o + 16u + (u32)(__builtin_ctzll(x2) >> 3) : x3 ? (u32)o + 24u + (u32)(__builtin_ctzll(x3) >> 3) : 0x7FFFFFFFu;
        if (gsh >= 1u) { const u32 k2 = (u32)__builtin_amdgcn_update_dpp((int)key, (int)key, 0xB1, 0xF, 0xF, false); key = k2 < key ? k2 : key; }   // quad_perm [1,0,3,2]
        if (gsh >= 2u) { const u32 k2 = (u32)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x4E, 0xF, 0xF, false); key = k2 < key ? k2 : key; }   // quad_perm [2,3,0,1]
        if (go && key != 0x7FFFFFFFu) { len = (int)key < best_possible ? (int)key : best_possible; go = false; }
    }
    // ChainMatches' choice: the first candidate of the best score (ScoreMatch :301-321, one property set: the length, cut to the distance in CompatibilityMode)
    int l2 = len;
    if (g.no_self_overlap && l2 > dist) l2 = dist;
    const int score = l2 - g.min_len;
    const u32 skey = (valid && t == 0u && score >= 0) ? (((u32)score + 1u) << 8) | (255u - j) : 0u;
    const u32 smax = (u32)__builtin_amdgcn_readlane((int)scan_max(skey), 63);
    if (smax != 0u) {
        const int sc = (int)(smax >> 8) - 1;
        if (sc > best_score) {
            const int jb = 255 - (int)(smax & 255u);
            best_score = sc;
            best_l = __builtin_amdgcn_readlane(l2, jb << gsh);
            best_d = __builtin_amdgcn_readlane(dist, jb << gsh);
            if (best_l == best_possible) return true;
        }
    }
    return false;
}

// `p4` (LZ4 blocks, raw Snappy: windows of 64 / 32 KiB): kernel A's links.  The nearest ALZ_SCAN_NEAR blocks are scanned -- in a run or a stretch of repeated rows that is where the candidates
// are --, and what lies further back is reached by FOLLOWING the chain from the farthest candidate seen (from the position itself where the scan found none): a position with few candidates
// would otherwise look at all 64 blocks of its window.
// Returns what the search cost beside its fixed part: the blocks it looked at + the links it followed (the probe's measure).
#ifndef ALZ_SCAN_NEAR
#define ALZ_SCAN_NEAR 2   /* (10 000 windows of Test.bmp at quality 8, ms per call, LZ4 blocks / raw Snappy: 1 -> 144.5 / 126.8, 2 -> 144.4 / 121.5, 4 -> 153.0 / 126.0, 8 -> 160.9 / 131.7; every block: 263.7 / 162.0) */
#endif
__device__ __forceinline__ int benc_wave_scan_search(const u8* data, int n, const EncGeom& g, int pos, int& best_d, int& best_l, unsigned short* cl, const int* p4 = nullptr) {
    const int lane = (int)benc_lane();
    const u8* dp = data + pos;
    const u32 sh = 32u - (u32)g.hash_bits;
    const u32 own = load32(dp) * 2654435761u;
    int best_possible = n - pos; if (best_possible > g.max_len) best_possible = g.max_len;
    best_d = 0; best_l = 0; int best_score = -1;
    int attempts = g.max_chain;
    const int lo = pos - g.max_dist > 0 ? pos - g.max_dist : 0;          // candidates: [lo, pos)
    int blocks = 0;
    int far = pos;                                                       // the farthest candidate seen (the chain goes on from it)
    int top = pos;
    for (; top > lo && attempts > 0 && (p4 == nullptr || blocks < ALZ_SCAN_NEAR); top -= 1024) {
        blocks++;
        // my sixteen positions of the block [top - 1024, top): [base, base + 16), read from b0 = max(base, 0) on (nothing is read in front of the stream)
        const int base = top - 1024 + 16 * lane;
        const int b0 = base > 0 ? base : 0;
        u32 m16 = 0;
        if (base + 16 > lo) {
            u32 w[5];
            __builtin_memcpy(w, data + b0, 20);                           // (up to pos + 2: inside the stream, pos <= n - 4)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 v = (u32)((((u64)w[(k >> 2) + 1] << 32) | w[k >> 2]) >> (8 * (k & 3)));
                if (((v * 2654435761u) ^ own) >> sh == 0u) m16 |= 1u << k;
            }
            // valid: absolute position a = b0 + k with lo <= a < base + 16
            const int hi = base + 16 - b0;                                // (16, fewer where the block starts in front of the stream)
            if (hi < 16) m16 &= (1u << (hi > 0 ? hi : 0)) - 1u;
            if (b0 < lo) { const int cut = lo - b0; m16 = cut >= 16 ? 0u : (m16 >> cut) << cut; }
        }
        if (__ballot(m16 != 0u) == 0ull) continue;
        // ---- the block's candidates, nearest first: rank = candidates in the lanes above mine + those above the bit in my own word
        const u32 mine = (u32)__popc(m16);
        const u32 incl = scan_add(mine);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u32 above = total - incl;
        u32 nc = total < (u32)attempts ? total : (u32)attempts;           // while (cur != -1 && attempts-- > 0)  :255 -- every listed candidate spends an attempt
        if (nc > 32u) nc = 32u;                                           // (never: the path is taken up to maxChain 32 -- quality 10 -- only, alz_launch_encode)
        {
            u32 mm = m16, r = above;
            while (mm && r < nc) { const int k = 31 - (int)__builtin_clz(mm); mm &= ~(1u << k); cl[r++] = (unsigned short)(pos - (b0 + k)); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        attempts -= (int)nc;
        far = pos - (int)cl[nc - 1u];
        const bool full = benc_scan_measure(data, g, pos, best_possible, cl, nc, lane, best_score, best_d, best_l);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (full) return blocks;
    }
    if (p4 != nullptr && attempts > 0 && top > lo) {
        // ---- behind the scanned blocks: the chain itself (ChainMatches :248-282 with 16-bit links), as many candidates as there are attempts left, then measured like a block's
        u32 nc = 0; int hops = 0;
        int cur = link_at<true>(p4, far);
        while (cur != -1 && attempts > 0) {
            attempts--; hops++;
            const int dist = pos - cur;
            if (dist > g.max_dist) break;                                  // :259-260
            if (lane == 0) cl[nc] = (unsigned short)dist;
            nc++;
            cur = link_at<true>(p4, cur);
        }
        blocks += hops;
        if (nc) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            (void)benc_scan_measure(data, g, pos, best_possible, cl, nc, lane, best_score, best_d, best_l);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
    }
    return blocks;
}

__device__ __forceinline__ u32 benc_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// The same function as kernel B runs it (the exact recomputations inside the roles / emit kernels keep the plain form above: the
// serial emit kernels, one lane per wavefront, ran 12 % slower with this body inlined).  MatchSearch :214-246 with ChainMatches
// :248-282; returns false when CAP > 0 and a
// candidate still matched after CAP bytes
// WAVEPOS: the wavefront's lanes hold 64 CONSECUTIVE positions (enc_match_kernel).  In a run or a stretch of repeated rows their candidates lie at ONE distance d, and
// 64 lanes each comparing up to 273 bytes compare the same 337 bytes 64 times over (68 loads per position -- the kernel is bound by the L1's lookups).  Then the mismatches of
// [P, P + 512) against [P - d, ...) are found once, eight bytes per lane, and every lane reads its length off the mismatch bits: the first one at or behind its own position.
template <bool MINT, bool L16, bool PRUNE = false, bool WAVEPOS = false>
__device__ __forceinline__ bool match_search_b(const u8* data, int n, int pos, const int* p4, const int* pm, const EncGeom& g, int cap,
                                             int& best_d, int& best_l) {
    const u8* dp = data + pos;
    int cur;                                                            // (each position's first link is read once, by this lane: past the caches)
    if (L16) { const u32 d = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(p4) + pos); cur = d ? pos - (int)d : -1; }
    else cur = __builtin_nontemporal_load(p4 + pos);
    int best_possible = n - pos; if (best_possible > g.max_len) best_possible = g.max_len;
    const int cmp_max = (cap > 0 && best_possible > cap) ? cap : best_possible;
    best_d = 0; best_l = 0; int best_score = -1;
    // The candidate loop is written for the wave, not for the lane: a fixed trip count (maxChain) with a per-lane `act` flag, one
    // wave-uniform exit test, and straight-line, predicated code in between -- the first eight bytes of a candidate are one load
    // against the position's own first qword (read once) and a count of trailing zeros; only a candidate that survives it enters the
    // compare loop.  As two nested divergent loops with early exits the kernel was bound by the CU's scalar unit (79 % busy: 87
    // scalar instructions of exec-mask bookkeeping per candidate), not by its loads (L1 serves 90 % of them).  Reading eight bytes
    // at a candidate or at the position may run up to four bytes past the stream: inside the staging buffer's slack, never compared
    // (lengths are clamped to cmp_max).
    const u64 head = load64(dp), head2 = load64(dp + 8);
    u64 own_bl = 0;                                                     // PRUNE: the position's eight bytes up to offset best_l (best_l >= 16)
    bool act = cur != -1, capped = false;
    const int chain = g.max_chain;
    for (int it = 0; it < chain; it++) {
        if (!__ballot(act)) break;
        const int c = act ? cur : pos;
        const int dist = pos - c;
        const bool within = act && dist <= g.max_dist;                  // beyond maxDistance the walk ends  :259-260
        // (a candidate out of reach is not touched: the previous position with this hash usually lies further back than a 4 KiB window
        //  reaches, anywhere in the stream -- an 8-byte read that misses every cache and is thrown away)
        const int cl = within ? c : pos;
        const int nxt = (within && it + 1 < chain) ? link_at<L16>(p4, cl) : -1;   // (the last candidate's link is never followed: at maxChain 1 that is every one)
        bool ok = within && dist >= g.min_dist;                         // closer than minDistance: skipped, the walk goes on  :262-266
        if (PRUNE && g.nprops <= 1) {
            // A candidate wins only with a LONGER match than the best so far (ScoreMatch :301-321 with one property set is the length, cut to
            // the distance in CompatibilityMode): one whose eight bytes up to offset best_l differ from the position's cannot, and is not
            // looked at any further -- one load instead of its sixteen bytes, the compare loop and what follows.  The kernel is bound by the
            // L1's lookups (a scattered load costs one per lane), and in the runs and repeated rows of real data nearly every candidate
            // behind the first good one ends here.
            const bool chk = ok && best_l >= 16;
            if (__ballot(chk)) {
                const u64 pb = load64(data + (chk ? cl + best_l - 7 : pos));
                if (chk && pb != own_bl) ok = false;
            }
        }
        int len = 0;
        bool shared = false;                                            // my candidate was measured with the wavefront's (below)
        if (WAVEPOS) {
            // (all 64 lanes here; classes of eight or more lanes with one distance, one class after the other while there are such; nobody compares more than
            // the 512 - 63 bytes the bits reach)
            u64 todo = __ballot(ok && cmp_max <= 448);
            const int lane = (int)benc_lane_id();
            if (__ballot(true) == ~0ull)
            while (__popcll(todo) >= 8) {
                const int d0 = __builtin_amdgcn_readlane(dist, (int)__builtin_ctzll(todo));
                const bool mine = ((todo >> lane) & 1ull) && dist == d0;
                const u64 same = __ballot(mine);
                if (__popcll(same) < 8 || pos - lane < d0) break;       // (P - d inside the stream: no group of eight bytes straddles its start)
                const int x0 = pos - lane + 8 * lane;                    // my eight bytes of [P, P + 512), P = the position of lane 0
                u64 xr = ~0ull;
                if (x0 < n) xr = load64(data + x0) ^ load64(data + x0 - d0);     // (up to seven bytes behind n: the buffer's slack; never counted -- cmp_max ends at n)
                u32 lo = (u32)xr, hi = (u32)(xr >> 32);
                lo |= lo >> 4; lo |= lo >> 2; lo |= lo >> 1; lo &= 0x01010101u;
                hi |= hi >> 4; hi |= hi >> 2; hi |= hi >> 1; hi &= 0x01010101u;
                const u32 bm = (((lo * 0x01020408u) >> 24) & 0xFu) | ((((hi * 0x01020408u) >> 24) & 0xFu) << 4);   // bit b: my byte b differs
                const u64 nz = __ballot(bm != 0u);
                const int g0 = lane >> 3, off = lane & 7;
                const u32 b0 = (u32)__builtin_amdgcn_ds_bpermute(g0 << 2, (int)bm) >> off;
                const u64 rest = g0 < 63 ? nz >> (g0 + 1) : 0ull;
                const int g1 = rest ? g0 + 1 + (int)__builtin_ctzll(rest) : 63;
                const u32 b1 = (u32)__builtin_amdgcn_ds_bpermute(g1 << 2, (int)bm);
                if (mine) { len = b0 ? (int)__builtin_ctz(b0) : rest ? 8 * g1 + (int)__builtin_ctz(b1) - lane : 512 - lane; shared = true; }
                todo &= ~same;
            }
        }
        const bool okc = ok && !shared;                                 // the others: each lane its own candidate
        if (__ballot(okc)) {
        const int cl2 = okc ? cl : pos;                                 // (a candidate that is out: its bytes are not fetched)
        u64 cv[2]; __builtin_memcpy(cv, data + cl2, 16);                // (the candidate's sixteen bytes as ONE load: the second eight cost no lookup of their own)
        const u64 x = head ^ cv[0];
        int len1 = x ? (int)(__builtin_ctzll(x) >> 3) : 8;
        const bool more = okc && x == 0ull && cmp_max > 8;
        if (__ballot(more)) {
            // the second eight bytes the same way (most formats' matches end inside them); the compare loop only behind sixteen
            const u64 y = head2 ^ cv[1];
            if (more) len1 = 8 + (y ? (int)(__builtin_ctzll(y) >> 3) : 8);
            const bool more2 = more && y == 0ull && cmp_max > 16;
            // (the compare loop runs for the whole wavefront as long as its longest lane: with the test above a 1 000 KiB Yaz0 stream at
            // quality 8 went 0.58 -> 0.29 ms, an LZ4 block at quality 15 15 -> 5.8)
            if (__ballot(more2)) { const int l3 = wave_match_tail(dp, data + (more2 ? c : 0), cmp_max, more2); if (more2) len1 = l3; }
        }
        if (!shared) len = len1;
        }
        if (len > cmp_max) len = cmp_max;
        bool stop = !within;
        if (g.nprops <= 1) {
            // (selects, not branches: nested ifs on per-lane conditions cost ~30 scalar instructions of exec-mask bookkeeping per
            // candidate, and the CU's one scalar unit is what bounds this kernel)
            const bool hitcap = ok && len == cmp_max && cmp_max < best_possible;
            int l2 = len;
            if (g.no_self_overlap && l2 > dist) l2 = dist;                   // ScoreMatch  :301-321, one property set
            const int score = l2 - g.min_len;
            const bool better = ok && !hitcap && score > best_score;
            best_score = better ? score : best_score; best_l = better ? l2 : best_l; best_d = better ? dist : best_d;
            if (PRUNE) { const bool rl = better && best_l >= 16 && best_l < best_possible; if (__ballot(rl)) { const u64 v = load64(dp + (rl ? best_l - 7 : 0)); if (rl) own_bl = v; } }
            capped = capped || hitcap;
            stop = stop || hitcap || (better && l2 == best_possible);
        } else if (ok) {
            if (len == cmp_max && cmp_max < best_possible) { capped = true; stop = true; }
            else {
                const int score = score_match(g, len, dist);
                if (score > best_score) { best_score = score; best_l = len; best_d = dist; if (best_l == best_possible) stop = true; }
            }
        }
        cur = nxt;
        act = act && !stop && cur != -1;
    }
    if (capped) return false;
    if (MINT && best_l == 0) {                                          // small-match fallback :226-243
        const int c2 = pm[pos];
        if (c2 != -1) {
            int dist = pos - c2;
            if (dist < g.min_dist) dist = g.min_dist;
            if (dist <= g.max_dist && pos - dist >= 0) {
                int len = match_len(dp, data + pos - dist, cmp_max);
                if (len == cmp_max && cmp_max < best_possible) return false;
                (void)score_match(g, len, dist);
                best_l = len; best_d = dist;
            }
        }
    }
    return true;
}

// Kernel B for finders that look at several candidates per position (maxChain >= 3), in two phases per wavefront and block of 64
// positions.  With one position per lane for the whole walk a wavefront goes on until its last lane is done: at Q8 14.7 of 64 lanes
// have a candidate in an average trip (28.8 at Q4).  Here the chains are walked first -- links only: p4[.], the distance rules, the
// attempt count -- and every candidate that ChainMatches would compare goes into an LDS list as (position, step, candidate); then the
// list is worked off 64 pairs at a time, every lane comparing.  MatchSearch keeps the FIRST candidate of the best score (a later one
// must be strictly better), and a candidate that reaches the longest possible match ends the walk -- nothing behind it could be
// strictly better -- so the result is the maximum over all listed pairs of (score, earliest step): one 64-bit LDS atomic max per pair,
// key = score + 1 | 4095 - step | length | distance.  A pair that runs into kernel B's length cap marks its position (ALZ_CAPPED, as
// before: the parse recomputes it exactly if it ever visits it); that can only happen where no candidate can reach the longest
// possible match, so the order of the two events in the sequential walk does not matter.
// DYN (maxChain >= 8, blocks of 256 positions): in the first phase a lane whose walk has ended takes the next position of the block --
// 59 -> 54 ms at Q8; at Q4 (five candidates at most) the fixed assignment in blocks of 64 is the faster one (36.8 against 38.0 ms).
#define ALZ_DENSE_LIST 256
template <bool MINT, bool DYN, int ALZ_DENSE_POS, bool L16>
__global__ __launch_bounds__(64) void enc_match_dense_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams,
                                                             const u32* __restrict__ index_list, const int* __restrict__ prev4,
                                                             const int* __restrict__ prevm, mentry* __restrict__ match,
                                                             const u64* __restrict__ pos_off, EncGeom g, int tail_skip, u32 xlog,
                                                             const u32* __restrict__ sel = nullptr) {
    __shared__ u32 lpos[ALZ_DENSE_LIST];          // position inside the block | step << 8
    __shared__ int lcand[ALZ_DENSE_LIST];
    __shared__ unsigned long long best[ALZ_DENSE_POS];
    __shared__ u32 capf[ALZ_DENSE_POS];
    const u32 sid = index_list[blockIdx.y];
    if (sid == 0xFFFFFFFFu) return;               // (a list written on the device, enc_scan_select_kernel: the stream goes the other way)
    const u32 selv = sel ? sel[sid] : 0u;         // (read together with the descriptor: as a test of its own in front of it, ten million workgroups paid one more round trip each)
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const int n = (int)st.src_len - tail_skip;
    const int limit = n - 4;
    const int* p4 = prev4 + pos_off[sid];
    const int* pm = MINT ? prevm + pos_off[sid] : nullptr;
    mentry* m = match + pos_off[sid];
    if (selv & 1u) return;                        // (enc_probe_kernel gave this stream to the one-position-per-lane kernel)
    const int lane = (int)threadIdx.x;
    const int chain = g.max_chain;
  // (the grid holds at most 4 096 workgroups per stream: a stream longer than 4 096 blocks -- and a batch whose longest stream is far
  //  longer than the others -- goes round; up to that length a workgroup has one block, which is the faster arrangement)
  // Which block: workgroups go to the eight XCDs in turn (blockIdx.x mod 8), so with block = blockIdx.x every block of 256 positions
  // pulled its own window of links and bytes (12 KB for 4 096 positions back) into another L2: 99 GB of fetches per 2.6 GB of input at
  // quality 8.  XCD k now takes RUNS of 2^xlog consecutive blocks, the runs dealt out in turn (gridDim.x is a multiple of 8 runs;
  // xlog 0 is the old order).  Measured (quality 8, 256 positions per block): runs of 4 blocks 52.3 ms and 40 GB, of 16 53.8 / 15, of 32
  // 58.4 / 12, an eighth of the stream per XCD 60.8 / 9 -- against 52.4 ms / 99 GB; with 64 positions per block (quality 4) the longest
  // runs are also the fastest (35.6 -> 34.9 ms).
  const u32 xk = blockIdx.x & 7u, xi = blockIdx.x >> 3;
  const long long blk0 = (long long)(((((xi >> xlog) << 3) + xk) << xlog) + (xi & ((1u << xlog) - 1u)));
  for (long long base64 = blk0 * ALZ_DENSE_POS; base64 <= (long long)limit; base64 += (long long)gridDim.x * ALZ_DENSE_POS) {
    const int base = (int)base64;
#pragma unroll
    for (int r = 0; r < ALZ_DENSE_POS / 64; r++) { best[64 * r + lane] = 0ull; capf[64 * r + lane] = 0u; }
    u32 ln = 0;                                   // pairs in the list (wave-uniform)

    // the pairs of the list, 64 at a time
    auto work_off = [&]() {
        for (u32 i0 = 0; i0 < ln; i0 += 64u) {
            const bool on = i0 + (u32)lane < ln;
            const u32 lp = on ? lpos[i0 + lane] : 0u;
            const int c = on ? lcand[i0 + lane] : 0;
            const u32 pl = lp & 0xFFu, step = lp >> 8;
            const int pos = base + (int)pl;
            const u8* dp = data + pos;
            const int dist = pos - c;
            int best_possible = n - pos; if (best_possible > g.max_len) best_possible = g.max_len;
            const int cmp_max = best_possible > g.b_cap ? g.b_cap : best_possible;
            // sixteen bytes of either side as ONE load each (round 3: eight, and eight more whenever some pair of the 64 had matched them --
            // nearly always --: a second round trip per batch, and a scattered load costs the L1 one lookup per lane whatever its width;
            // may run a few bytes past the stream: inside the staging slack, never compared)
            u64 hv[2], cv[2];
            __builtin_memcpy(hv, dp, 16); __builtin_memcpy(cv, data + c, 16);
            const u64 x = hv[0] ^ cv[0], y = hv[1] ^ cv[1];
            int len = x ? (int)(__builtin_ctzll(x) >> 3) : (y ? 8 + (int)(__builtin_ctzll(y) >> 3) : 16);
            const bool more2 = on && len == 16 && cmp_max > 16;
            if (__ballot(more2)) { const int l3 = wave_match_tail(dp, data + (more2 ? c : 0), cmp_max, more2); if (more2) len = l3; }
            if (len > cmp_max) len = cmp_max;
            if (on) {
                if (len == cmp_max && cmp_max < best_possible) capf[pl] = 1u;
                else {
                    const int score = score_match(g, len, dist);               // (may shorten len: CompatibilityMode, property sets)
                    if (score >= 0) {
                        const unsigned long long key = ((unsigned long long)(u32)(score + 1) << 52) | ((unsigned long long)(4095u - step) << 40) |
                                                       ((unsigned long long)(u32)len << 28) | (unsigned long long)(u32)dist;
                        (void)__hip_atomic_fetch_max(&best[pl], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                }
            }
        }
        ln = 0;
    };

    // ---- 1. the chains, links only
    if constexpr (DYN) {
        // A lane whose walk has ended takes the next position of the block.  The state a lane carries from trip to trip is integers only:
        // the link it loaded in the trip before -- raw, i.e. a position or a 16-bit distance from `cbase` -- becomes its candidate at the
        // top of the next trip, so the loads of a trip (the next link of the walking lanes, the first link of the lanes that took a new
        // position) are in flight together and waited for once; "active" is "has a candidate", never a flag of its own (round 3: as
        // per-lane booleans carried around the loop, `act` and `fresh` went through 0 / 1 registers and compares on every trip, and the
        // conversion of a 16-bit link sat right behind its load).
        const int npos = limit - base + 1 < ALZ_DENSE_POS ? limit - base + 1 : ALZ_DENSE_POS;
        constexpr int NONE = L16 ? 0 : -1;
        constexpr u32 IDLE = 0x7FFFFFFFu;                                   // the distance of a lane without a candidate
        const u32 range = (u32)(g.max_dist - g.min_dist);
        // (two registers for the two loads of a trip: one register, written under two different lane masks, made the second wait for the
        // first.  What the lane state is kept as is the candidate's DISTANCE -- with 16-bit links the next one is this one plus the link --,
        // and every lane condition below is ONE compare on it: a ballot of an AND of compares goes through a 0 / 1 register.)
        int nextp = 0, pl = 0, pos = 0, it = 0, rawr = NONE, rawf = NONE; u32 dbase = 0;
        for (;;) {
            const int raw = L16 ? (rawr | rawf) : (rawr & rawf);            // (at most one of them holds a link; waits for the loads of the trip before)
            const u32 dist = L16 ? (raw != 0 ? dbase + (u32)raw : IDLE) : (raw != -1 ? (u32)(pos - raw) : IDLE);
            rawr = NONE; rawf = NONE;
            it++;
            const u64 actm = __ballot(dist != IDLE);
            u64 refm = 0;
            if (~actm && nextp < npos) {
                const u64 idle = ~actm;
                const int np = nextp + (int)__builtin_amdgcn_mbcnt_hi((u32)(idle >> 32), __builtin_amdgcn_mbcnt_lo((u32)idle, 0u));
                nextp += (int)__popcll(idle);
                refm = __ballot(np < npos) & idle;
                if (dist == IDLE && np < npos) {
                    pl = np; pos = base + np; it = -1; dbase = 0;             // (its first candidate arrives a trip later, as step 0)
                    if (L16) rawr = (int)reinterpret_cast<const unsigned short*>(p4)[pos]; else rawr = p4[pos];
                }
            }
            if (!(actm | refm)) break;
            const int c = pos - (int)dist;                                      // (meaningless on a lane without a candidate; never used there)
            const bool within = dist <= (u32)g.max_dist;                        // beyond maxDistance the walk ends  :259-260
            const bool ok = dist - (u32)g.min_dist <= range;                    // closer than minDistance: skipped, the walk goes on  :262-266
            if (within && it + 1 < chain) {                                     // the link behind this candidate: next trip's candidate
                if (L16) rawf = (int)reinterpret_cast<const unsigned short*>(p4)[c]; else rawf = p4[c];   // (the last candidate's link is never followed, nor that of one out of reach)
                dbase = dist;
            }
            const u64 om = __ballot(ok);
            if (om) {
                const u32 k = (u32)__popcll(om);
                if (ln + k > ALZ_DENSE_LIST) work_off();
                if (ok) {
                    const u32 at = ln + __builtin_amdgcn_mbcnt_hi((u32)(om >> 32), __builtin_amdgcn_mbcnt_lo((u32)om, 0u));
                    lpos[at] = (u32)pl | ((u32)it << 8);
                    lcand[at] = c;
                }
                ln += k;
            }
        }
    } else {
#pragma unroll 1
    for (int r = 0; r < ALZ_DENSE_POS / 64; r++) {
        // (the same integer-only lane state as above: the link loaded in a trip becomes the candidate at the top of the next one, the
        // candidate is kept as its distance, every lane condition is one compare)
        constexpr int NONE = L16 ? 0 : -1;
        constexpr u32 IDLE = 0x7FFFFFFFu;
        const u32 range = (u32)(g.max_dist - g.min_dist);
        const int pos = base + 64 * r + lane;
        int raw = NONE; u32 dbase = 0;
        if (pos <= limit) { if (L16) raw = (int)reinterpret_cast<const unsigned short*>(p4)[pos]; else raw = p4[pos]; }
        for (int it = 0; it < chain; it++) {
            const u32 dist = L16 ? (raw != 0 ? dbase + (u32)raw : IDLE) : (raw != -1 ? (u32)(pos - raw) : IDLE);
            if (!__ballot(dist != IDLE)) break;
            const int c = pos - (int)dist;                                      // (meaningless on a lane without a candidate; never used there)
            const bool within = dist <= (u32)g.max_dist;                        // beyond maxDistance the walk ends  :259-260
            const bool ok = dist - (u32)g.min_dist <= range;                    // closer than minDistance: skipped, the walk goes on  :262-266
            raw = NONE;
            if (within && it + 1 < chain) {                                     // (the last candidate's link is never followed, nor that of a candidate out of reach)
                if (L16) raw = (int)reinterpret_cast<const unsigned short*>(p4)[c]; else raw = p4[c];
                dbase = dist;
            }
            const u64 om = __ballot(ok);
            if (om) {
                const u32 k = (u32)__popcll(om);
                if (ln + k > ALZ_DENSE_LIST) work_off();
                if (ok) {
                    const u32 at = ln + __builtin_amdgcn_mbcnt_hi((u32)(om >> 32), __builtin_amdgcn_mbcnt_lo((u32)om, 0u));
                    lpos[at] = (u32)(64 * r + lane) | ((u32)it << 8);
                    lcand[at] = c;
                }
                ln += k;
            }
        }
    }
    }
    work_off();

    // ---- 2. every position's match
#pragma unroll 1
    for (int r = 0; r < ALZ_DENSE_POS / 64; r++) {
        const int pl = 64 * r + lane, pos = base + pl;
        if (pos > limit) continue;
        const unsigned long long key = best[pl];
        bool capped = capf[pl] != 0u;
        int best_l = (int)((key >> 28) & 0xFFFu), best_d = (int)(key & 0xFFFFFFFu);
        if (MINT && !capped && best_l == 0) {                                  // small-match fallback :226-243
            const int c2 = pm[pos];
            if (c2 != -1) {
                int best_possible = n - pos; if (best_possible > g.max_len) best_possible = g.max_len;
                const int cmp_max = best_possible > g.b_cap ? g.b_cap : best_possible;
                int dist = pos - c2;
                if (dist < g.min_dist) dist = g.min_dist;
                if (dist <= g.max_dist && pos - dist >= 0) {
                    int len = match_len(data + pos, data + pos - dist, cmp_max);
                    if (len == cmp_max && cmp_max < best_possible) capped = true;
                    else { (void)score_match(g, len, dist); best_l = len; best_d = dist; }
                }
            }
        }
        __builtin_nontemporal_store(capped ? 0xFFFFFFFFu : m_pack((u32)best_d, (u32)best_l), m + pos);
    }
  }
}

// (Tried in round 3: kernel B with the stream's window in LDS -- a workgroup of eight wavefronts copies the bytes and the links, as 16-bit
// distances, of [base - 4096, base + 8192) into LDS and runs both phases of the dense kernel on LDS alone, candidate bytes from three aligned
// dwords and v_alignbyte.  Memory traffic falls from 223 GB to about 40, but the kernel is not bound by it: 78 ms against 54 at quality 8,
// 701 against 112 at quality 15.  Counters: the LDS pipe is 5 % busy, VALU + SALU issue 86 % at quality 8 -- the walk is bound by the
// instructions of its trips, and the LDS form has more of them (address arithmetic, three reads and two alignbytes per eight bytes) at
// half the wavefronts per CU (68 KB of LDS per workgroup); at quality 15 a workgroup waits for the one wavefront whose block holds a
// 1024-step chain.  docs/EXPERIMENTS.md 8.)
template <bool MINT, bool L16, bool PRUNE = false>
__global__ __launch_bounds__(256) void enc_match_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams,
                                                        const u32* __restrict__ index_list, const int* __restrict__ prev4,
                                                        const int* __restrict__ prevm, mentry* __restrict__ match,
                                                        const u64* __restrict__ pos_off, EncGeom g, int tail_skip, const u32* __restrict__ list = nullptr) {
  // (`list`: the streams enc_probe_kernel gave to this kernel -- their number, then their ids; the grid's y is then smaller than the batch
  // and goes round: a batch of which the probe gives this kernel nothing costs a few thousand empty workgroups, not count x 32)
  const u32 ny = list ? list[0] : gridDim.y;
  for (u32 y = blockIdx.y; y < ny; y += gridDim.y) {
    const u32 sid = list ? list[1u + y] : index_list[y];
    if (sid == 0xFFFFFFFFu) continue;             // (enc_scan_select_kernel's list: the stream goes the other way)
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const int n = (int)st.src_len - tail_skip;
    const int limit = n - 4;
    const int* p4 = prev4 + pos_off[sid];
    const int* pm = MINT ? prevm + pos_off[sid] : nullptr;
    mentry* m = match + pos_off[sid];
    // A workgroup takes ONE contiguous range of the stream (round 3).  With the positions of a workgroup 32 Ki apart -- 256 here, 256 there --
    // every group of 256 fetched its own 4 KiB of history into its XCD's L2: 30 GB of fetches for 2.6 GB of input, and this is the one
    // kernel of the path that waits for memory (0.57 instructions per cycle and CU).
    const int span = (((limit + 1 + (int)gridDim.x - 1) / (int)gridDim.x) + 255) & ~255;
    const int first = (int)blockIdx.x * span;
    const int last = first + span - 1 < limit ? first + span - 1 : limit;
    for (int pos = first + (int)threadIdx.x; pos <= last; pos += 256) {
        int bd, bl;
        const bool okm = match_search_b<MINT, L16, PRUNE, true>(data, n, pos, p4, pm, g, g.b_cap, bd, bl);       // (256 threads, positions in thread order: a wavefront's are consecutive)
        __builtin_nontemporal_store(okm ? m_pack((u32)bd, (u32)bl) : 0xFFFFFFFFu, m + pos);   // (written once, read by the next kernel: past the caches)
    }
  }
}

// The same search with the lanes kept busy: a lane whose walk has ended stores its result and takes the next position of its wavefront's
// range, instead of waiting until the longest walk of its 64 positions is over.  On real data the walks are of very different lengths -- in a
// run the first candidate has the full length and ends it, next to it a position goes through all 16 (quality 8); where it is used and what it
// gains: launch_match.  One property set, 16-bit links, PRUNE as above; maxChain >= 3 (below that there is nothing to wait for).
template <bool MINT>
__global__ __launch_bounds__(256) void enc_match_dyn_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams,
                                                            const u32* __restrict__ index_list, const int* __restrict__ prev4,
                                                            const int* __restrict__ prevm, mentry* __restrict__ match,
                                                            const u64* __restrict__ pos_off, EncGeom g, int tail_skip, const u32* __restrict__ list = nullptr) {
  const u32 ny = list ? list[0] : gridDim.y;
  for (u32 y = blockIdx.y; y < ny; y += gridDim.y) {
    const u32 sid = list ? list[1u + y] : index_list[y];
    if (sid == 0xFFFFFFFFu) continue;             // (enc_scan_select_kernel's list: the stream goes the other way)
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const int n = (int)st.src_len - tail_skip;
    const int limit = n - 4;
    const int* p4 = prev4 + pos_off[sid];
    const unsigned short* p16 = reinterpret_cast<const unsigned short*>(p4);
    const int* pm = MINT ? prevm + pos_off[sid] : nullptr;
    mentry* m = match + pos_off[sid];
    const int span = (((limit + 1 + (int)gridDim.x - 1) / (int)gridDim.x) + 255) & ~255;
    const int first = (int)blockIdx.x * span;
    const int last = first + span - 1 < limit ? first + span - 1 : limit;
    const int wspan = span >> 2, w = (int)(threadIdx.x >> 6);
    int nextp = first + w * wspan;                                       // (wave-uniform) the next position nobody has taken
    const int wlast = nextp + wspan - 1 < last ? nextp + wspan - 1 : last;
    const int chain = g.max_chain, cap = g.b_cap;
    // lane state: the position (-1: none), its walk
    int pos = -1, cur = -1, it = 0, best_l = 0, best_d = 0, best_score = -1, best_possible = 0, cmp_max = 0;
    bool capped = false; u64 head = 0, head2 = 0, own_bl = 0;
    for (;;) {
        const u64 idle = __ballot(pos < 0);
        if (idle && nextp <= wlast) {
            const int np = nextp + (int)__builtin_amdgcn_mbcnt_hi((u32)(idle >> 32), __builtin_amdgcn_mbcnt_lo((u32)idle, 0u));
            nextp += (int)__popcll(idle);
            if (pos < 0 && np <= wlast) {
                pos = np;
                const u32 d = __builtin_nontemporal_load(p16 + pos); cur = d ? pos - (int)d : -1;
                best_possible = n - pos; if (best_possible > g.max_len) best_possible = g.max_len;
                cmp_max = best_possible > cap ? cap : best_possible;
                best_l = 0; best_d = 0; best_score = -1; capped = false; it = 0;
                head = load64(data + pos); head2 = load64(data + pos + 8);
            }
        }
        const bool have = pos >= 0;
        if (!__ballot(have)) break;
        const bool act = have && cur != -1;
        const u8* dp = data + (have ? pos : 0);
        // ---- one candidate (match_search_b's loop body, one property set)
        const int c = act ? cur : pos;
        const int dist = pos - c;
        const bool within = act && dist <= g.max_dist;
        const int cl = within ? c : (have ? pos : 0);
        const int nxt = (within && it + 1 < chain) ? link_at<true>(p4, cl) : -1;
        bool ok = within && dist >= g.min_dist;
        {   // (a candidate whose eight bytes up to offset best_l differ cannot win: match_search_b)
            const bool chk = ok && best_l >= 16;
            if (__ballot(chk)) {
                const u64 pb = load64(data + (chk ? cl + best_l - 7 : 0));
                if (chk && pb != own_bl) ok = false;
            }
        }
        const int cl2 = ok ? cl : (have ? pos : 0);
        u64 cv[2]; __builtin_memcpy(cv, data + cl2, 16);
        const u64 x = head ^ cv[0];
        int len = x ? (int)(__builtin_ctzll(x) >> 3) : 8;
        const bool more = ok && x == 0ull && cmp_max > 8;
        if (__ballot(more)) {
            const u64 yv = head2 ^ cv[1];
            if (more) len = 8 + (yv ? (int)(__builtin_ctzll(yv) >> 3) : 8);
            const bool more2 = more && yv == 0ull && cmp_max > 16;
            if (__ballot(more2)) { const int l3 = wave_match_tail(dp, data + (more2 ? c : 0), cmp_max, more2); if (more2) len = l3; }
        }
        if (len > cmp_max) len = cmp_max;
        const bool hitcap = ok && len == cmp_max && cmp_max < best_possible;
        int l2 = len;
        if (g.no_self_overlap && l2 > dist) l2 = dist;
        const int score = l2 - g.min_len;
        const bool better = ok && !hitcap && score > best_score;
        best_score = better ? score : best_score; best_l = better ? l2 : best_l; best_d = better ? dist : best_d;
        { const bool rl = better && best_l >= 16 && best_l < best_possible; if (__ballot(rl)) { const u64 v = load64(dp + (rl ? best_l - 7 : 0)); if (rl) own_bl = v; } }
        capped = capped || hitcap;
        const bool stop = !within || hitcap || (better && l2 == best_possible);
        cur = nxt; it++;
        const bool goon = act && !stop && cur != -1;
        if (have && !goon) {                                             // this position is done
            bool okm = !capped;
            if (MINT && okm && best_l == 0) {                            // small-match fallback :226-243
                const int c2 = pm[pos];
                if (c2 != -1) {
                    int dd = pos - c2;
                    if (dd < g.min_dist) dd = g.min_dist;
                    if (dd <= g.max_dist && pos - dd >= 0) {
                        int ln = match_len(dp, data + pos - dd, cmp_max);
                        if (ln == cmp_max && cmp_max < best_possible) okm = false;
                        else { (void)score_match(g, ln, dd); best_l = ln; best_d = dd; }
                    }
                }
            }
            __builtin_nontemporal_store(okm ? m_pack((u32)best_d, (u32)best_l) : 0xFFFFFFFFu, m + pos);
            pos = -1;
        }
    }
  }
}

// (Tried in round 3 for maxChain 1 (quality 0), where this kernel issues only 0.57 instructions per cycle and CU: four positions per thread
// with the loads of every stage in flight together -- links and the positions' own sixteen bytes, then the candidates' sixteen bytes as one
// load each, then the arithmetic: 16.7-17.1 ms against 16.8.  Neither the chain of dependent round trips nor the L1's lookups (0.77 per
// cycle) is the bound: without the link loads (10.5 GB) the kernel takes 12.7 ms, without the scattered candidate loads 15.3 -- it moves
// 18 GB in and 21 GB out (8 bytes of match per position) at ~2.4 TB/s, three streams per workgroup against a copy kernel's two at 5.8.
// What would pay is fewer bytes per position in the arrays the kernels hand to each other: docs/EXPERIMENTS.md 8.  Tried again at the end of the
// round with the arrays at 2 + 4 bytes per position (19.5 GB per launch, 1.4 TB/s, 0.75 instructions per cycle): 13.8 ms either way;
// one store in 64: 13.1, no scattered candidate loads: 12.5 -- no single stream of accesses is the bound.)
// ---------------------------------------------------------------------------------------------- kernel C
struct Out {                 // bounded byte sink of one stream
    u8* p; u32 len, cap; bool fail;
    __device__ __forceinline__ void put(u32 b) { if (len < cap) p[len] = (u8)b; else fail = true; len++; }
    __device__ __forceinline__ void put16be(u32 v) { put(v >> 8); put(v & 0xFF); }
    __device__ __forceinline__ void put16le(u32 v) { put(v & 0xFF); put(v >> 8); }
    __device__ __forceinline__ void copy(const u8* s, u32 n) {       // a literal run: eight bytes at a time where it fits (sources carry 64 bytes of slack)
        if (len <= cap && n <= cap - len) {
            u32 i = 0;
            for (; i + 8u <= n; i += 8u) { const u64 v = load64(s + i); __builtin_memcpy(p + len + i, &v, 8); }
            for (; i < n; i++) p[len + i] = s[i];
            len += n;
        } else for (u32 i = 0; i < n; i++) put(s[i]);
    }
};

// FlagWriter  IO/FlagWriter.cs:13-147: the flag byte (or big-endian flag word) goes out before the payload of its tokens.  The managed
// writer buffers the payload until the flag is complete; here the flag's place is reserved when its first bit arrives, the payload that
// follows goes straight to the output and the flag is filled in when it is complete -- the same bytes in the same order, without a
// payload array per lane in scratch memory (a memory round trip per byte for the one lane that works: PRS 323 ms of emit).  Only payload
// that arrives while NO flag is open has to wait (at most the bytes of one token): whether a flag goes in front of it is decided by what
// comes next -- a bit (yes), flush_if_necessary() or the end (no).
struct FlagW {
    Out* base; int bits_left, width; u32 cur, slot; bool msb, neg, open; unsigned long long pend; int npend;
    // nbytes 1: byte flags (optionally stored negated, LZ40); 2 / 4: big-endian flag words (SMSR00 / LZHudson)
    __device__ void init(Out* b, bool m, bool negate = false, int nbytes = 1) { base = b; width = 8 * nbytes; bits_left = width; cur = 0; slot = 0; msb = m; neg = negate; open = false; pend = 0; npend = 0; }
    __device__ void put_pending() { for (int i = 0; i < npend; i++) base->put((u32)(pend >> (8 * i)) & 0xFFu); npend = 0; pend = 0; }
    __device__ void close_flag() {
        if (width == 8) { if (slot < base->cap) base->p[slot] = (u8)(neg ? (0u - cur) & 0xFFu : cur); }     // LZ40: i => WriteByte((byte)-i)
        else { u32 k = 0; for (int i = width - 8; i >= 0; i -= 8, k++) if (slot + k < base->cap) base->p[slot + k] = (u8)((cur >> i) & 0xFFu); }
        bits_left = width; cur = 0; open = false;
    }
    __device__ void flush() { if (open) close_flag(); put_pending(); }
    __device__ void bit(int b) {
        if (!open) { slot = base->len; for (int i = 0; i < width; i += 8) base->put(0); open = true; put_pending(); }
        if (b) cur |= 1u << (msb ? bits_left - 1 : width - bits_left);
        if (--bits_left == 0) close_flag();
    }
    __device__ void pay(u32 v) { if (open) base->put(v); else { pend |= (unsigned long long)(v & 0xFFu) << (8 * npend); npend++; } }
    __device__ void flush_if_necessary() { if (!open) put_pending(); }
};

struct Match { int offset, distance, length; };

// FindNextBestMatch  LzChainMatchFinder.cs:157-212 for the serial emitters: the parse is the start mask of the roles walk (one bit per
// match it takes, set at the match's first byte), so the next match is the next set bit, and only ITS entry of the match array is read.
struct Finder {
    const mentry* m; int n, limit, position;
    const u64* mask = nullptr; int widx = -1; u64 wbits = 0, wnext = 0;
    __device__ Match next() {
        const int nwords = limit >= 0 ? (limit >> 6) + 1 : 0;
        for (;;) {
            if (wbits == 0ull) {
                widx++;
                if (widx >= nwords) break;
                wbits = widx == 0 ? mask[0] : wnext;
                if (widx + 1 < nwords) wnext = mask[widx + 1];       // (the word after this one is on its way while this one is used)
                continue;
            }
            const int b = (int)__builtin_ctzll(wbits);
            wbits &= wbits - 1ull;
            const int p = widx * 64 + b;
            uint2 r = m_unpack(m[p]);                                // (exact: the roles walk has recomputed what kernel B had capped)
            if (r.y == ALZ_M_LONG) r.y = m[p + 1];
            Match out = { p, (int)r.x, (int)r.y };
            return out;
        }
        position = n;
        Match e = { n, 0, 0 };
        return e;
    }
};

__device__ void lz4_ext(Out& o, int length) {                // LZ4.WriteExtension  LZ4.cs:254-268
    length -= 0xF;
    if (length >= 0) { int b; do { b = length < 0xFF ? length : 0xFF; o.put((u32)b); length -= b; } while (b == 0xFF); }
}
__device__ void lzo_ext(Out& o, int v) { while (v > 255) { o.put(0); v -= 255; } o.put((u32)v); }   // LZO.WriteExtendedInt

template <int FMT>
__global__ __launch_bounds__(64) void enc_emit_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base,
                                                      const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                      u32 count, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                      u8* __restrict__ side, alz_result* __restrict__ results,
                                                      alz_encode_aux* __restrict__ aux, EncGeom g, u32 lone, const u64* __restrict__ startmask) {
    // lone: ONE stream per wavefront, lane 0 works.  Sixty-four streams per wavefront executed the union of 64 divergent token
    // paths with their flag-writer state in scratch memory: PRS 1 900 ms per 10 000 x 256 KiB against the 280 ms of the four
    // kernels in front of it; a lone lane per wavefront is latency bound instead, and 10 000 wavefronts hide each other's latency.
    const u32 i = lone ? blockIdx.x : blockIdx.x * 64 + threadIdx.x;
    if (i >= count || (lone && threadIdx.x != 0)) return;
    const u32 sid = index_list[i];
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const int n = (int)st.src_len;
    Out out = { dst_base + st.dst_off, 0, st.dst_cap, false };
    Finder mf; mf.m = match + pos_off[sid]; mf.position = 0;
    mf.n = (FMT == ALZ_FMT_LZ4_BLOCK) ? n - 5 : n; mf.limit = mf.n - 4;
    mf.mask = startmask + (pos_off[sid] >> 6);
    int status = ALZ_ST_OK; u32 a0 = 0, a1 = 0;
    int sp = 0;

    if constexpr (FMT == ALZ_FMT_LZSS) {                                  // LZSS.cs:132-160
        FlagW fw; fw.init(&out, false);
        const u32 nmask = g.lz_max_distance - 1, f = (1u << g.length_bits) - 1;
        for (;;) {
            Match mt = mf.next();
            for (int plain = mt.offset - sp; plain != 0; plain--) { fw.pay(src[sp++]); fw.bit(1); }
            if (mt.length == 0) break;
            const u32 offset = (g.windows_start + (u32)sp - (u32)mt.distance) & nmask;
            const u32 v = (offset & 0xFF) | ((offset & 0xFF00) << g.length_bits) | ((((u32)mt.length - g.lz_min_length) & f) << 8);
            fw.pay(v & 0xFF); fw.pay((v >> 8) & 0xFF); fw.bit(0);
            sp += mt.length;
        }
        fw.flush();
    } else if constexpr (FMT == ALZ_FMT_LZ10 || FMT == ALZ_FMT_LZ11) {      // LZ10.cs:113-137, LZ11.cs:135-171
        FlagW fw; fw.init(&out, true);
        for (;;) {
            Match mt = mf.next();
            for (int plain = mt.offset - sp; plain != 0; plain--) { fw.pay(src[sp++]); fw.bit(0); }
            if (mt.length == 0) break;
            const u32 d1 = (u32)(mt.distance - 1) & 0xFFF;
            if (FMT == ALZ_FMT_LZ10) { const u32 v = (((u32)mt.length - 3) << 12 | d1) & 0xFFFF; fw.pay(v >> 8); fw.pay(v & 0xFF); }
            else if (mt.length <= 16) { const u32 v = (((u32)mt.length - 1) << 12 | d1) & 0xFFFF; fw.pay(v >> 8); fw.pay(v & 0xFF); }
            else if (mt.length <= 272) { fw.pay((((u32)mt.length - 17) & 0xFF) >> 4); const u32 v = (((u32)mt.length - 17) << 12 | d1) & 0xFFFF; fw.pay(v >> 8); fw.pay(v & 0xFF); }
            else { const u32 v = 0x10000000u | ((((u32)mt.length - 273) & 0xFFFF) << 12) | d1; fw.pay(v >> 24); fw.pay((v >> 16) & 0xFF); fw.pay((v >> 8) & 0xFF); fw.pay(v & 0xFF); }
            sp += mt.length;
            fw.bit(1);
        }
        fw.flush();
    } else if constexpr (FMT == ALZ_FMT_LZ40) {                             // LZ40.cs:134-176
        FlagW fw; fw.init(&out, true, true);
        for (;;) {
            Match mt = mf.next();
            for (int plain = mt.offset - sp; plain != 0; plain--) { fw.pay(src[sp++]); fw.bit(0); }
            if (mt.length == 0) break;
            const u32 dv = ((u32)mt.distance << 4) & 0xFFFFu;
            if (mt.length < 16) { const u32 v = dv | (u32)mt.length; fw.pay(v & 0xFF); fw.pay(v >> 8); }
            else if (mt.length < 272) { fw.pay(dv & 0xFF); fw.pay(dv >> 8); fw.pay((u32)mt.length - 16); }
            else { const u32 v = dv | 1u, l = ((u32)mt.length - 272u) & 0xFFFFu; fw.pay(v & 0xFF); fw.pay(v >> 8); fw.pay(l & 0xFF); fw.pay(l >> 8); }
            sp += mt.length;
            fw.bit(1);
        }
        fw.flush();
    } else if constexpr (FMT == ALZ_FMT_YAZ0) {                             // Yaz0.cs:94-98 over Yay0.cs:152-184
        FlagW fw; fw.init(&out, true);
        for (;;) {
            Match mt = mf.next();
            for (int plain = mt.offset - sp; plain != 0; plain--) { fw.pay(src[sp++]); fw.bit(1); }
            if (mt.length == 0) break;
            if (mt.length < 18) { const u32 v = ((u32)(mt.distance - 1) | (((u32)mt.length - 2) << 12)) & 0xFFFF; fw.pay(v >> 8); fw.pay(v & 0xFF); }
            else { const u32 v = (u32)(mt.distance - 1) & 0xFFF; fw.pay(v >> 8); fw.pay(v & 0xFF); fw.pay((u32)mt.length - 0x12); }
            sp += mt.length;
            fw.bit(0);
        }
        fw.flush();
    } else if constexpr (FMT == ALZ_FMT_LZHUDSON) {                         // LZHudson.cs:55-59: Yay0's tokens, one stream, 32-bit BE flag words
        FlagW fw; fw.init(&out, true, false, 4);
        for (;;) {
            Match mt = mf.next();
            for (int plain = mt.offset - sp; plain != 0; plain--) { fw.pay(src[sp++]); fw.bit(1); }
            if (mt.length == 0) break;
            if (mt.length < 18) { const u32 v = ((u32)(mt.distance - 1) | (((u32)mt.length - 2) << 12)) & 0xFFFF; fw.pay(v >> 8); fw.pay(v & 0xFF); }
            else { const u32 v = (u32)(mt.distance - 1) & 0xFFF; fw.pay(v >> 8); fw.pay(v & 0xFF); fw.pay((u32)mt.length - 0x12); }
            sp += mt.length;
            fw.bit(0);
        }
        fw.flush();
    } else if constexpr (FMT == ALZ_FMT_SMSR00) {                           // SMSR00.cs:133-137 over MIO0.cs:159-184: codes | literals
        u8* sb = side + 2 * pos_off[sid];
        Out unc = { sb, 0, (u32)n + 16, false };
        FlagW fw; fw.init(&out, true, false, 2);
        for (;;) {
            Match mt = mf.next();
            for (int plain = mt.offset - sp; plain != 0; plain--) { unc.put(src[sp++]); fw.bit(1); }
            if (mt.length == 0) break;
            const u32 v = ((u32)(mt.distance - 1) | (((u32)mt.length - 3) << 12)) & 0xFFFF; fw.pay(v >> 8); fw.pay(v & 0xFF);
            sp += mt.length;
            fw.bit(0);
        }
        fw.flush();
        a0 = out.len; a1 = 0;
        out.copy(unc.p, unc.len);
    } else if constexpr (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0) {      // Yay0.cs:62-77,152-184 / MIO0.cs:64-79,159-184
        // three sections: flags go straight to dst, tokens and literals to side buffers, concatenated afterwards
        u8* sb = side + 2 * pos_off[sid];
        Out comp = { sb, 0, (u32)n + 16, false };
        Out unc = { sb + n + 16, 0, (u32)n + 16, false };
        FlagW fw; fw.init(&out, true);
        for (;;) {
            Match mt = mf.next();
            for (int plain = mt.offset - sp; plain != 0; plain--) { unc.put(src[sp++]); fw.bit(1); }
            if (mt.length == 0) break;
            if (FMT == ALZ_FMT_MIO0) comp.put16be(((u32)(mt.distance - 1) | (((u32)mt.length - 3) << 12)) & 0xFFFF);
            else if (mt.length < 18) comp.put16be(((u32)(mt.distance - 1) | (((u32)mt.length - 2) << 12)) & 0xFFFF);
            else { comp.put16be((u32)(mt.distance - 1) & 0xFFF); unc.put((u32)mt.length - 0x12); }
            sp += mt.length;
            fw.bit(0);
        }
        fw.flush();
        a0 = out.len; a1 = out.len + comp.len;
        out.copy(comp.p, comp.len); out.copy(unc.p, unc.len);
    } else if constexpr (FMT == ALZ_FMT_PRS_BE || FMT == ALZ_FMT_PRS_LE) {  // PRS.cs:104-159
        constexpr bool big = FMT == ALZ_FMT_PRS_BE;
        FlagW fw; fw.init(&out, big);
        for (;;) {
            Match mt = mf.next();
            for (int plain = mt.offset - sp; plain != 0; plain--) { fw.pay(src[sp++]); fw.bit(1); }
            if (mt.length == 0) break;
            if (mt.length == 2 && mt.distance > 0x100) continue;
            sp += mt.length;
            const int distance = -mt.distance, length = mt.length;
            fw.bit(0);
            if (distance >= -0x100 && length <= 5) {
                fw.bit(0); fw.bit(((length - 2) >> 1) & 1); fw.bit((length - 2) & 1);
                fw.pay((u32)distance & 0xFF);
                fw.flush_if_necessary();
            } else {
                u32 v = (u32)(distance << 3) & 0xFFFF;
                if (length <= 9) v |= (u32)(length - 2);
                if (big) { fw.pay(v >> 8); fw.pay(v & 0xFF); } else { fw.pay(v & 0xFF); fw.pay(v >> 8); }
                if (length > 9) fw.pay((u32)(length - 1));
                fw.bit(1);
            }
        }
        fw.bit(0); fw.pay(0); fw.pay(0); fw.bit(1);
        fw.flush();
    } else if constexpr (FMT == ALZ_FMT_LZ4_BLOCK) {                        // LZ4.cs:202-238
        if (n < 5) status = ALZ_ST_BAD_TOKEN;                               // source.Slice(0, Length - 5) throws
        else for (;;) {
            Match mt = mf.next();
            int plain = mt.offset - sp;
            int token = (plain > 0xF ? 0xF : plain) << 4;
            if (mt.length != 0) token |= (mt.length - 4 > 0xF ? 0xF : mt.length - 4);
            else { plain = n - sp; token = (plain > 0xF ? 0xF : plain) << 4; }
            out.put((u32)token);
            lz4_ext(out, plain);
            out.copy(src + sp, (u32)plain);
            sp += plain;
            if (sp >= n) break;
            out.put16le((u32)mt.distance & 0xFFFF);
            lz4_ext(out, mt.length - 4);
            sp += mt.length;
        }
    } else if constexpr (FMT == ALZ_FMT_LZO) {                              // LZO.cs:141-250
        if (n < 0x10) { out.put((u32)(17 + n)); out.copy(src, (u32)n); out.put(0x11); out.put(0); out.put(0); }
        else {
            Match mt = mf.next(), nx = mf.next();
            while (sp != n) {
                int plain = mt.offset - sp;
                if (plain != 0) {
                    if (plain < 4) { const int dif = 4 - plain; mt.offset += dif; mt.length -= dif; plain = 4; }
                    if (plain > 18) { out.put(0); lzo_ext(out, plain - 18); } else out.put((u32)(plain - 3));
                    if (sp + plain > n) { status = ALZ_ST_BAD_TOKEN; break; }
                    out.copy(src + sp, (u32)plain); sp += plain;
                }
                if (mt.length >= 3) {
                    sp += mt.length;
                    plain = nx.offset - sp;
                    if (plain > 3) plain = 0;
                    if (plain < 0) { status = ALZ_ST_BAD_TOKEN; break; }
                    if (mt.length <= 8 && mt.distance <= 2048) {
                        const u32 flag = (u32)(plain | (((mt.distance - 1) & 0x7) << 2)) & 0xFF;
                        if (mt.length <= 4) out.put(flag | 0x40 | (u32)((mt.length - 3) << 5)); else out.put(flag | 0x80 | (u32)((mt.length - 5) << 5));
                        out.put((u32)((mt.distance - 1) >> 3));
                    } else if (mt.distance <= 16384) {
                        if (mt.length > 33) { out.put(0x20); lzo_ext(out, mt.length - 33); } else out.put(0x20 | (u32)(mt.length - 2));
                        out.put((u32)(plain | ((mt.distance - 1) << 2)) & 0xFF); out.put((u32)((mt.distance - 1) >> 6) & 0xFF);
                    } else {
                        const int distance = mt.distance - 0x4000;
                        const u32 flag = (u32)(0x10 | ((distance & 0x4000) >> 11)) & 0xFF;
                        if (mt.length > 9) { out.put(flag); lzo_ext(out, mt.length - 9); } else out.put(flag | (u32)(mt.length - 2));
                        out.put((u32)(plain | (distance << 2)) & 0xFF); out.put((u32)(distance >> 6) & 0xFF);
                    }
                    if (sp + plain > n) { status = ALZ_ST_BAD_TOKEN; break; }
                    out.copy(src + sp, (u32)plain); sp += plain;
                }
                mt = nx; nx = mf.next();
            }
            out.put(0x11); out.put(0); out.put(0);
        }
    } else if constexpr (FMT == ALZ_FMT_HIG) {                              // HIG.cs:214-323
        auto put_raw = [&](int plain) {                                     // count - 2 in a byte, or 0 + the count as (ushort)
            if (plain <= 255 + 2) out.put((u32)(plain - 2) & 0xFF); else { out.put(0); out.put16le((u32)plain & 0xFFFF); }
        };
        Match nx, mt = mf.next();
        int plain = mt.offset;
        if (plain < 2) {                                                    // the initial block holds at least 2 bytes  :223-232
            mt.length -= plain + 1; mt.offset = 2;
            plain = 2;
            if (mt.length < g.min_len) { mt = mf.next(); plain = mt.offset; }
        }
        put_raw(plain);
        out.copy(src, (u32)plain);
        sp = plain;
        while (mt.length != 0) {
            nx = mf.next();
            plain = nx.offset - (mt.offset + mt.length);
            int b = plain == 0 ? 3 : (plain == 1 ? 1 : (plain == 2 ? 2 : 0));
            if (mt.distance <= 0x7FF && mt.length <= 5 + 4) {
                b |= ((mt.length - 4) << 5) | ((mt.distance >> 6) & 0x1C);
                out.put((u32)b & 0xFF);
            } else {
                if (mt.distance <= 0x3FFF && mt.length <= 31 + 4) out.put((u32)(0xC0 | (mt.length - 4)));
                else {
                    const int length = mt.length <= 15 + 3 ? mt.length - 3 : 0;
                    out.put((u32)(0xE0 | ((mt.distance >> 10) & 0x10) | length));
                    if (length == 0) {
                        if (mt.length <= 255 + 18) out.put((u32)(mt.length - 18) & 0xFF);
                        else { out.put(0); out.put16be((u32)mt.length & 0xFFFF); }
                    }
                }
                b |= (mt.distance >> 6) & 0xFC;
                out.put((u32)b & 0xFF);
            }
            out.put((u32)mt.distance & 0xFF);
            sp += mt.length;
            if (plain != 0) {
                if (plain > 2) put_raw(plain);
                out.copy(src + sp, (u32)plain);
                sp += plain;
            }
            mt = nx;
        }
    } else if constexpr (FMT == ALZ_FMT_LZSHREK) {                          // LZShrek.cs:121-174
        u8 buffer[32]; int blen;
        Match mt = mf.next();
        while (sp != n) {
            const int plain = mt.offset - sp; int clen = 0;
            const u8* unc = src + sp;
            sp += plain;
            blen = 0;
            while (mt.length != 0 && clen < 8 && mt.offset == sp) {
                const int lf = mt.length > 7 ? 0 : mt.length;
                const int df = mt.distance > 30 ? (mt.distance > 286 ? 0x1F : 0x1E) : mt.distance - 1;
                buffer[blen++] = (u8)((df << 3) | lf);
                if (lf == 0) buffer[blen++] = (u8)(mt.length - 7);
                if (df == 0x1E) buffer[blen++] = (u8)(mt.distance - 31);
                else if (df == 0x1F) { const u32 v = (u32)(mt.distance - 287) & 0xFFFFu; buffer[blen++] = (u8)v; buffer[blen++] = (u8)(v >> 8); }
                sp += mt.length;
                mt = mf.next();
                if (mt.length == 0 || clen >= 7 || mt.offset != sp) break;
                clen++;
            }
            const int uf = plain > 29 ? (plain > 285 ? 0x1F : 0x1E) : plain;
            out.put((u32)((uf << 3) | clen) & 0xFF);
            if (uf == 0x1E) out.put((u32)(plain - 30) & 0xFF);
            else if (uf == 0x1F) out.put16le((u32)(plain - 286) & 0xFFFF);
            out.copy(unc, (u32)plain);
            for (int i = 0; i < blen; i++) out.put(buffer[i]);
        }
        out.put(0); out.put(0); out.put(0); out.put(0);
    } else if constexpr (FMT == ALZ_FMT_WFLZ || FMT == ALZ_FMT_WFLZ_BE) {   // WFLZ.cs:161-196
        Match mt = { 0, 0, 0 }, nx = mf.next();
        int plain = nx.offset;
        for (;;) {
            const u32 bp = (u32)(plain < 255 ? plain : 255), d = (u32)mt.distance & 0xFFFFu;
            if (FMT == ALZ_FMT_WFLZ_BE) out.put16be(d); else out.put16le(d);
            out.put(mt.length == 0 ? 0u : (u32)(mt.length - 4)); out.put(bp);
            plain -= (int)bp;
            sp += mt.length;
            out.copy(src + sp, bp); sp += (int)bp;
            if (plain == 0) {
                if (sp == n) break;
                mt = nx; nx = mf.next();
                plain = nx.offset - (mt.offset + mt.length);
            } else { mt.offset = 0; mt.distance = 0; mt.length = 0; }
        }
        out.put(0); out.put(0); out.put(0); out.put(0);                     // end block
    } else if constexpr (FMT == ALZ_FMT_REFPACK) {                          // RefPack.cs:247-303
        int plain = 0;
        for (;;) {
            Match mt = mf.next();
            plain = mt.offset - sp;
            while (plain > 3) {                                             // runs of 4..112 literals, multiples of four
                int c = (plain > 0x70 ? 0x70 : plain) / 4 - 1;
                out.put((u32)(0xE0 | c));
                c = c * 4 + 4;
                out.copy(src + sp, (u32)c); sp += c; plain -= c;
            }
            if (mt.length == 0) break;
            const int d1 = mt.distance - 1;
            if (mt.length <= 10 && mt.distance <= 0x400) { out.put((u32)(plain | ((d1 & 0x300) >> 3) | ((mt.length - 3) << 2))); out.put((u32)d1 & 0xFF); }
            else if (mt.length >= 4 && mt.length <= 67 && mt.distance <= 0x4000) { out.put((u32)(0x80 | (mt.length - 4))); out.put((u32)((d1 >> 8) | (plain << 6)) & 0xFF); out.put((u32)d1 & 0xFF); }
            else { out.put((u32)(0xC0 | ((d1 >> 16) << 4) | (((mt.length - 5) >> 8) << 2) | plain) & 0xFF); out.put((u32)(d1 >> 8) & 0xFF); out.put((u32)d1 & 0xFF); out.put((u32)(mt.length - 5) & 0xFF); }
            out.copy(src + sp, (u32)plain);
            sp += plain + mt.length;
            plain = 0;
        }
        out.put((u32)(0xFC | plain)); out.copy(src + sp, (u32)plain);     // the end token carries the last 0-3 literals
    } else if constexpr (FMT == ALZ_FMT_LZ02) {                             // LZ02.cs:117-151
        FlagW fw; fw.init(&out, true);
        for (;;) {
            Match mt = mf.next();
            for (int plain = mt.offset - sp; plain != 0; plain--) { fw.pay(src[sp++]); fw.bit(0); }
            if (mt.length == 0) break;
            const u32 length = mt.length > 16 ? 0u : (u32)mt.length - 1u;
            fw.pay((((u32)mt.distance >> 8) << 4) | length); fw.pay((u32)mt.distance & 0xFF);
            if (length == 0) fw.pay((u32)mt.length - 17u);
            sp += mt.length;
            fw.bit(1);
        }
        fw.pay(0); fw.pay(0); fw.bit(1);                                    // terminator
        fw.flush();
    } else if constexpr (FMT == ALZ_FMT_CNS) {                              // CNS.cs:111-141
        for (;;) {
            Match mt = mf.next();
            int plain = mt.offset - sp;
            while (plain != 0) {
                const int length = plain < 127 ? plain : 127;
                out.put((u32)length); out.copy(src + sp, (u32)length);
                sp += length; plain -= length;
            }
            if (mt.length == 0) break;
            out.put((u32)(0x80 | (mt.length - 3))); out.put((u32)(mt.distance - 1));
            sp += mt.length;
        }
    } else if constexpr (FMT == ALZ_FMT_CNX2) {                             // CNX2.cs:140-172
        // FlagWriter order without its buffer (a flag byte's payload can be four 256-byte runs): the flag byte's slot is
        // reserved when its first code is written and patched when the fourth is (or at the end): the same bytes
        u32 flagpos = 0, cur = 0, ncodes = 0;
        auto code = [&](u32 v) {
            cur |= v << (2u * ncodes);
            if (++ncodes == 4u) { if (flagpos < out.cap) out.p[flagpos] = (u8)cur; cur = 0; ncodes = 0; }
        };
        auto reserve = [&]() { if (ncodes == 0u) { flagpos = out.len; out.put(0); } };
        for (;;) {
            Match mt = mf.next();
            int plain = mt.offset - sp;
            while (plain != 0) {
                const int length = plain < 255 ? plain : 255;
                reserve();
                if (length == 1) { out.put(src[sp]); code(1); }
                else { out.put((u32)length); out.copy(src + sp, (u32)length); code(3); }
                sp += length; plain -= length;
            }
            if (mt.length == 0) break;
            reserve();
            out.put16be((u32)((((mt.distance - 1) & 0x7FF) << 5) | ((mt.length - 4) & 0x1F)));
            sp += mt.length;
            code(2);
        }
        if (ncodes && flagpos < out.cap) out.p[flagpos] = (u8)cur;
    } else if constexpr (FMT == ALZ_FMT_FASTLZ) {                           // FastLZ.cs:162-245, levels 1 and 2 (g.variant)
        const bool level2 = g.variant == 1;
        bool first = level2;                                                // the level tag rides on the first literal run
        for (;;) {
            Match mt = mf.next();
            int plain = mt.offset - sp;
            while (plain > 0) {                                             // literal runs of 1..32
                const int chunk = plain < 32 ? plain : 32;
                out.put((u32)(chunk - 1) | (first ? 0x20u : 0u)); first = false;
                out.copy(src + sp, (u32)chunk); sp += chunk; plain -= chunk;
            }
            if (mt.length == 0) break;
            int length = mt.length - 3, distance = mt.distance - 1;
            const int sd = level2 ? (distance < 0x1FFF ? distance : 0x1FFF) : distance;
            out.put((u32)((((length < 6 ? length : 6) + 1) << 5) | (sd >> 8)) & 0xFFu);
            if (length >= 6) {
                length -= 6;
                while (level2 && length >= 255) { out.put(255); length -= 255; }
                out.put((u32)length & 0xFFu);
            }
            out.put((u32)sd & 0xFF);
            if (level2 && distance >= 0x1FFF) { distance -= 0x1FFF; out.put((u32)(distance >> 8) & 0xFFu); out.put((u32)distance & 0xFFu); }
            sp += mt.length;
        }
    } else {                                                                // Snappy.cs:124-203
        int v = n; while (v >= 0x80) { out.put((u32)(v | 0x80) & 0xFF); v >>= 7; } out.put((u32)v);
        for (;;) {
            Match mt = mf.next();
            const int plain = mt.offset - sp;
            if (plain > 0) {
                if (plain <= 60) out.put((u32)((plain - 1) << 2));
                else {
                    const int len = plain - 1;
                    if (len <= 0xFF) { out.put(60 << 2); out.put((u32)len); }
                    else if (len <= 0xFFFF) { out.put(61 << 2); out.put16le((u32)len); }
                    else if (len <= 0xFFFFFF) { out.put(62 << 2); out.put((u32)len & 0xFF); out.put(((u32)len >> 8) & 0xFF); out.put(((u32)len >> 16) & 0xFF); }
                    else { out.put(63 << 2); out.put16le((u32)len & 0xFFFF); out.put16le((u32)len >> 16); }
                }
                out.copy(src + sp, (u32)plain); sp += plain;
            }
            if (mt.length == 0) break;
            sp += mt.length;
            if (mt.distance < 2048 && mt.length >= 4 && mt.length <= 11) { out.put((u32)(1 | ((mt.length - 4) << 2) | ((mt.distance >> 8) << 5))); out.put((u32)mt.distance & 0xFF); }
            else { out.put((u32)(2 | ((mt.length - 1) << 2)) & 0xFF); out.put16le((u32)mt.distance & 0xFFFF); }
        }
    }
    if (out.fail && status == ALZ_ST_OK) status = ALZ_ST_OUTPUT_CAPACITY;
    alz_result r; r.dst_len = out.fail ? 0u : out.len; r.src_used = st.src_len; r.status = status; r.reserved = 0;
    results[sid] = r;
    if (aux) { aux[sid].aux0 = a0; aux[sid].aux1 = a1; }
}

// ---------------------------------------------------------------------------------------------- kernels C1 + C2
// Lane-parallel emission for the flag-byte formats (LZSS / LZ10 / LZ11 / Yaz0 / Yay0 / MIO0).
//
// C1 (enc_roles_kernel, one wavefront per stream): the greedy/lazy parse.  For 64 consecutive positions at a time every
//    lane evaluates "what FindNextBestMatch does if its cursor is HERE" (no token / match here / literal + match at the
//    next position, and where the cursor goes), then the real cursor hops through the window with v_readlane.  Output:
//    one bit per position, "a match token starts here" (the match itself is match[p]).
// C2 (the second half of enc_parse_emit_kernel, one wavefront per stream): with the match starts known everything else is prefix sums over
//    positions: covered positions (prefix max of match ends), literals, token index (-> flag group and bit), payload
//    offsets.  Token lanes store their payload bytes; flag bytes are accumulated in LDS and stored when their group
//    completes.  FlagWriter order (IO/FlagWriter.cs:70-80,111-127): flag byte, then the payload of its 8 tokens.

__device__ __forceinline__ u32 scan_add(u32 v) {            // inclusive wave prefix sum (gfx9 DPP)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
__device__ __forceinline__ u32 scan_max(u32 v) {            // inclusive wave prefix max
    u32 t;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); v = v > t ? v : t;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); v = v > t ? v : t;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); v = v > t ? v : t;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); v = v > t ? v : t;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); v = v > t ? v : t;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); v = v > t ? v : t;
    return v;
}

// SEGS (alz_encode_seg.h): the grid's x is a segment of the stream; the walk runs from that segment's synchronisation point (a position every
// walk lands on, enc_sync_kernel) to the next segment's -- several wavefronts per stream, each exact.  Mask words are OR-ed into memory: the window
// that holds a synchronisation point is written from both sides (the same bits where they overlap).
template <bool SEGS>
__global__ __launch_bounds__(64) void enc_roles_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams,
                                                       const u32* __restrict__ index_list, u32 count, mentry* __restrict__ match,
                                                       const u64* __restrict__ pos_off, const int* __restrict__ prev4,
                                                       const int* __restrict__ prevm, u64* __restrict__ startmask, EncGeom g, int tail_skip,
                                                       const u32* __restrict__ sync = nullptr, u32 kpitch = 0) {
    // The match entries come through a tile in LDS: 1 024 positions (+ 64: the neighbour of a window's last lane) per fill, the next tile on its way
    // in registers meanwhile.  With one window loaded ahead (round 3) every window of 64 positions waited for a load from HBM -- ~1.1 us, whatever the
    // walk itself took: 1.17 ms per 64 KiB (profiles/r05_mid_batch_encode.md); a tile pays that once per sixteen windows.
    constexpr int TILE = 1024, TLEN = TILE + 64, TPL = TLEN / 64;
    __shared__ u8 hopmark[64];
    __shared__ mentry tile[TLEN];
    const u32 bid = SEGS ? blockIdx.y : blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)threadIdx.x;
    hopmark[lane] = 0;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const int n = (int)st.src_len - tail_skip;            // LZ4 searches source[0 : n-5]  (LZ4.cs:208)
    const int limit = n - 4;
    mentry* m = match + pos_off[sid];
    const int* p4 = prev4 + pos_off[sid];
    const int* pm = g.use_min_table ? prevm + pos_off[sid] : nullptr;
    u64* mask = startmask + (pos_off[sid] >> 6);
    int cur = 0;                    // cursor of FindNextBestMatch (absolute position)
    int wend = 0x7FFFFFFF;          // SEGS: where the next wavefront takes over
    if (SEGS) {
        const u32* sy = sync + (size_t)bid * kpitch;
        const u32 k = blockIdx.x;
        const u32 s0 = sy[k];
        if (s0 == 0xFFFFFFFFu) return;                                  // (no synchronisation point in front of this segment: the walk before goes on through it)
        cur = (int)s0;
        for (u32 j0 = k + 1u; j0 < kpitch; j0 += 64u) {
            const u32 j = j0 + (u32)lane;
            const u32 v = j < kpitch ? sy[j] : 0xFFFFFFFFu;
            const u64 bal = __ballot(v != 0xFFFFFFFFu);
            if (bal) { wend = __builtin_amdgcn_readlane((int)v, (int)__builtin_ctzll(bal)); break; }
        }
    }
    auto put_mask = [&](u32 w, u64 bits) {                              // (lane 0)
        if (SEGS) atomicOr(reinterpret_cast<unsigned long long*>(mask + w), (unsigned long long)bits); else mask[w] = bits;
    };
    int tbase = -TILE - TILE;       // first position of the tile in LDS (none yet)
    int pbase = -1;                 // first position of the tile in `pf` (-1: none)
    mentry pf[TPL];
    // (positions above `limit` were never searched: no match -- their entries are whatever an earlier batch left there)
    auto fetch = [&](int base) {
#pragma unroll
        for (int j = 0; j < TPL; j++) { const int q = base + 64 * j + lane; pf[j] = q <= limit ? __builtin_nontemporal_load(m + q) : 0u; }
        pbase = base;
    };
    auto install = [&]() {
#pragma unroll
        for (int j = 0; j < TPL; j++) tile[64 * j + lane] = pf[j];
        tbase = pbase; pbase = -1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    };
    u32 carryw = 0xFFFFFFFFu;       // window whose lane 0 starts a token found by the window in front of it (none: ~0)
    bool lone = false;              // SEGS: the next cursor is taken on its own (below)
    while (cur <= limit && cur < wend) {
        const int P = cur & ~63;    // window that holds the cursor (windows the cursor jumps over keep their zero mask)
        const int p = P + lane;
        const bool carry_in = carryw == ((u32)P >> 6);
        if (!carry_in && carryw != 0xFFFFFFFFu && lane == 0) put_mask(carryw, 1ull);  // (that window is jumped over: its only bit)
        carryw = 0xFFFFFFFFu;
        if (P < tbase || P >= tbase + TILE) {                                          // (the cursor only moves forward)
            const int want = P & ~(TILE - 1);
            if (pbase != want) fetch(want);
            install();
            if (want + TILE <= limit) fetch(want + TILE);
        }
        if (SEGS && lone) {
            // ONE cursor, not a window: in a run or a stretch of repeated rows a jump leaves its window (273 bytes: four windows on), and the window
            // machinery below -- 64 jumps, the hop loop, the marks, three ballots: ~2 000 cycles for a lone wavefront -- would run once per token.  The
            // cursor's two entries come from the tile at a uniform address.  Back to windows at the first short jump or capped entry.
            const uint2 ca = m_unpack(tile[cur - tbase]), cb = m_unpack(tile[cur + 1 - tbase]);
            if (ca.y != ALZ_CAPPED && cb.y != ALZ_CAPPED) {
                int cj = 1, csr = 0;
                if ((int)ca.y >= g.min_len) {
                    const int l0 = (int)ca.y, l1 = (int)cb.y;
                    const bool lazyc = l0 <= g.lazy && cur + 1 <= limit;
                    if (lazyc && l1 > l0) { csr = 2; const int e = cur + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; cj = (cur + 2 > stop ? cur + 2 : stop) - cur; }
                    else { csr = 1; const int skip = lazyc ? 1 : 0; const int e = cur + l0; const int stop = e < limit + 1 ? e : limit + 1; cj = (cur + 1 + skip > stop ? cur + 1 + skip : stop) - cur; }
                }
                cj = __builtin_amdgcn_readfirstlane(cj); csr = __builtin_amdgcn_readfirstlane(csr);
                if (carry_in && lane == 0) put_mask((u32)P >> 6, 1ull);
                if (csr && lane == 0) { const u32 bp = (u32)cur + (csr == 2 ? 1u : 0u); put_mask(bp >> 6, 1ull << (bp & 63u)); }
                cur += cj;
                if (cj < 64) lone = false;
                continue;
            }
            lone = false;
        }
        // match[p] and match[p+1]
        // (an ALZ_M_LONG entry -- or the raw length behind one -- is never looked at here: both lie at or inside a match this walk has taken)
        const uint2 a = m_unpack(tile[P - tbase + lane]);
        const uint2 b = m_unpack(tile[P - tbase + lane + 1]);
        const bool capped = a.y == ALZ_CAPPED || b.y == ALZ_CAPPED;
        int jump = 1, startrel = 0;   // startrel: 0 no token here, 1 match starts here, 2 literal here + match at p + 1
        if (!capped && p <= limit && (int)a.y >= g.min_len) {
            const int l0 = (int)a.y, l1 = (int)b.y;
            const bool lazyc = l0 <= g.lazy && p + 1 <= limit;
            if (lazyc && l1 > l0) { startrel = 2; const int e = p + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; jump = (p + 2 > stop ? p + 2 : stop) - p; }
            else { startrel = 1; const int skip = lazyc ? 1 : 0; const int e = p + l0; const int stop = e < limit + 1 ? e : limit + 1; jump = (p + 1 + skip > stop ? p + 1 + skip : stop) - p; }
        }
        u64 bits = carry_in ? 1ull : 0ull;
        int rel = cur - P;
        // The walk over the window: a chain of "cursor += jump[cursor]" hops, ~18 per window.  Written in C++ each hop is three
        // v_readlane and ~30 scalar instructions, and with 32 waves per CU the kernel was bound by the CU's one scalar unit
        // (24.6 ms per 10 000 x 256 KiB).  Without capped candidates in the window the hop is five instructions: the visited lanes
        // are collected as a bit mask, and the token starts follow from two ballots.
        if (__ballot(capped) == 0) {
            u64 M = 0; u32 r = (u32)rel, j;
            const u32 lim = (u32)(limit + 1 - P) < 64u ? (u32)(limit + 1 - P) : 64u;     // (r < lim on entry: cur <= limit)
            if (lim == 64u) {
                // a full window: the cursor runs 64 below zero, so that the add's carry IS "left the window" -- four instructions per hop
                // (s_bitset1 and v_readlane take the low six bits of it, which are the cursor's) -- and a hop is TWO tokens: jump2 = my
                // jump + the jump of where it lands; the lanes in between are where the marked lanes' own jumps land (enc_parse_emit_kernel)
                const u32 tgt = (u32)lane + (u32)jump;
                const int jn = __builtin_amdgcn_ds_bpermute((int)((tgt < 64u ? tgt : (u32)lane) << 2), jump);
                const int jump2 = tgt < 64u ? jump + jn : jump;
                r -= 64u;
                asm volatile(
                    "s_nop 3\n"
                    "1:\n\t"
                    "s_bitset1_b64 %[M], %[r]\n\t"
                    "v_readlane_b32 %[j], %[jump], %[r]\n\t"
                    "s_add_u32 %[r], %[r], %[j]\n\t"
                    "s_cbranch_scc0 1b\n\t"
                    : [M] "+s"(M), [r] "+s"(r), [j] "=&s"(j)
                    : [jump] "v"(jump2)
                    : "scc");
                r += 64u;
                if (((M >> lane) & 1ull) && tgt < 64u) hopmark[tgt] = 1;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                const u32 hm = hopmark[lane];
                if (hm) hopmark[lane] = 0;
                M |= __ballot(hm != 0u);
            } else
            asm volatile(
                "s_nop 3\n"
                "1:\n\t"
                "s_bitset1_b64 %[M], %[r]\n\t"
                "v_readlane_b32 %[j], %[jump], %[r]\n\t"
                "s_add_u32 %[r], %[r], %[j]\n\t"
                "s_cmp_lt_u32 %[r], %[lim]\n\t"
                "s_cbranch_scc1 1b\n\t"
                : [M] "+s"(M), [r] "+s"(r), [j] "=&s"(j)
                : [jump] "v"(jump), [lim] "s"(lim)
                : "scc");
            const u64 s1 = __ballot(startrel == 1) & M, s2 = __ballot(startrel == 2) & M;
            bits |= s1 | (s2 << 1);
            if (s2 >> 63) carryw = ((u32)P >> 6) + 1u;                         // start in lane 0 of the next window
            rel = (int)r;
            if (SEGS && __popcll(M) <= 2) lone = true;                         // (a window of one or two tokens: the next cursor on its own)
        }
        else while (rel < 64 && P + rel <= limit) {
            int j, sr;
            if (__builtin_amdgcn_readlane((int)capped, rel)) {
                // kernel B capped a candidate here: redo MatchSearch exactly for this cursor and its lazy neighbour
                const int q = P + rel;
                int d0, l0, d1 = 0, l1 = 0; bool s0, s1;
                // (by the whole wavefront, 4 KiB per trip: as a plain loop on every lane a match of a few hundred bytes took ~10 us per cursor)
                const uint2 e0 = m_unpack(tile[q - tbase]), e1 = m_unpack(tile[q + 1 - tbase]);
                if (g.use_min_table) benc_capped_cursor<true>(data, n, g, p4, pm, q, limit, e0, e0.y == ALZ_CAPPED, e1, e1.y == ALZ_CAPPED, d0, l0, d1, l1, s0, s1);
                else benc_capped_cursor<false>(data, n, g, p4, pm, q, limit, e0, e0.y == ALZ_CAPPED, e1, e1.y == ALZ_CAPPED, d0, l0, d1, l1, s0, s1);
                j = 1; sr = 0;
                if (l0 >= g.min_len) {
                    const bool lazyc = l0 <= g.lazy && q + 1 <= limit;
                    if (lazyc && l1 > l0) { sr = 2; const int e = q + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; j = (q + 2 > stop ? q + 2 : stop) - q; }
                    else { sr = 1; const int skip = lazyc ? 1 : 0; const int e = q + l0; const int stop = e < limit + 1 ? e : limit + 1; j = (q + 1 + skip > stop ? q + 1 + skip : stop) - q; }
                }
                // The exact results go back into the array: the match that is TAKEN in full (the emitters read it; 2 046 bytes or more: the
                // length in the next entry, a position inside the match), the other one as far as an entry holds it -- q's is never looked
                // at again, q + 1's is if the walk goes there next, and stays "capped" (recomputed then) when it is too long for an entry.
                // (the tile in LDS gets the lazy neighbour's entry too: the walk may stand on it next.  The tile on its way in `pf` may still hold
                // kernel B's entry for it -- exact or "capped", in which case it is searched again: the same result)
                // (only what was searched here: an entry that was exact stays -- and the neighbour is searched only behind a match of the lazy threshold or less, never a long one)
                if (lane == 0) {
                    if (s0) {
                        if (sr == 1 && l0 >= (int)ALZ_M_LONG) { m[q] = m_pack((u32)d0, ALZ_M_LONG); m[q + 1] = (u32)l0; }
                        else m[q] = m_pack((u32)d0, l0 < (int)ALZ_M_LONG ? (u32)l0 : ALZ_M_LONG - 1u);
                    }
                    if (s1) {
                        if (l1 < (int)ALZ_M_LONG) { m[q + 1] = m_pack((u32)d1, (u32)l1); tile[q + 1 - tbase] = m_pack((u32)d1, (u32)l1); }
                        else if (sr == 2) { m[q + 1] = m_pack((u32)d1, ALZ_M_LONG); m[q + 2] = (u32)l1; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            } else {
                j = __builtin_amdgcn_readlane(jump, rel);
                sr = __builtin_amdgcn_readlane(startrel, rel);
            }
            if (sr == 1) bits |= 1ull << rel;
            else if (sr == 2) { if (rel + 1 < 64) bits |= 1ull << (rel + 1); else carryw = ((u32)P >> 6) + 1u; }   // start in lane 0 of the next window
            rel += j;
        }
        if (bits && lane == 0) put_mask((u32)P >> 6, bits);   // (one wavefront per stream: a plain store -- the mask array is zeroed before the launch, a window
                                                              //  is visited once, and the one bit another window contributes travels in `carryw`: no load per window)
        cur = P + rel;
    }
    if (carryw != 0xFFFFFFFFu && lane == 0) put_mask(carryw, 1ull);
}

// Synchronisation points of the parse (alz_encode_seg.h): a position s that NO jump from a position in front of it crosses -- max over q < s of
// q + jump(q) <= s -- is a cursor of every walk, wherever it started: steps are at least 1 and none goes over s.  Per stream and segment boundary
// S = k * seglen: the last such position in (S - seglen, S], or none (a run, a stretch of repeated rows: one match behind the other for whole segments --
// the walk in front then carries on through this segment).  One wavefront per boundary: jump(q) for every position of the segment in front and of the
// longest jump's worth of positions before it, exactly as the walk computes it; a prefix maximum.  A position kernel B has capped has no jump here: nothing behind
// it counts as a synchronisation point for this boundary.
__global__ __launch_bounds__(64) void enc_sync_kernel(const alz_stream* __restrict__ streams, const u32* __restrict__ index_list, const mentry* __restrict__ match,
                                                      const u64* __restrict__ pos_off, u32* __restrict__ sync, u32 kpitch, u32 seglen, EncGeom g) {
    const u32 k = blockIdx.x, bid = blockIdx.y;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const int n = (int)streams[sid].src_len;
    const int limit = n - 4;
    u32* out = sync + (size_t)bid * kpitch + k;
    const int S = (int)(k * seglen);
    if (k == 0u) { if (lane == 0) *out = 0u; return; }
    if (S > limit) { if (lane == 0) *out = 0xFFFFFFFFu; return; }
    const mentry* m = match + pos_off[sid];
    const int hist = (g.max_len + 2 + 63) & ~63;
    const int lo = S - (int)seglen;
    const int x0 = lo - hist > 0 ? lo - hist : 0;
    u32 pmx = 0, best = 0xFFFFFFFFu;
    auto ldm = [&](int q) { return q <= limit ? m_unpack(m[q]) : make_uint2(0, 0); };
    for (int P = x0; P < S; P += 64) {
        const int p = P + lane;
        const uint2 a = ldm(p), b = ldm(p + 1);
        int jump = 1;
        if (a.y == ALZ_CAPPED || b.y == ALZ_CAPPED) jump = 0x40000000;
        else if ((int)a.y >= g.min_len) {                               // (p < S <= limit: searched)
            const int l0 = (int)a.y, l1 = (int)b.y;
            const bool lazyc = l0 <= g.lazy && p + 1 <= limit;
            if (lazyc && l1 > l0) { const int e = p + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; jump = (p + 2 > stop ? p + 2 : stop) - p; }
            else { const int skip = lazyc ? 1 : 0; const int e = p + l0; const int stop = e < limit + 1 ? e : limit + 1; jump = (p + 1 + skip > stop ? p + 1 + skip : stop) - p; }
        }
        const u32 incl = scan_max((u32)(p + jump));
        u32 excl = (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138, 0xF, 0xF, false);     // wave_shr:1 -> max over the lanes below
        if (excl < pmx) excl = pmx;
        const u64 bal = __ballot(p > lo && excl <= (u32)p);
        if (bal) best = (u32)P + 63u - (u32)__builtin_clzll(bal);
        const u32 wm = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        if (wm > pmx) pmx = wm;
    }
    if (pmx <= (u32)S) best = (u32)S;
    if (lane == 0) *out = best;
}

// The payload bytes of a match token of the flag-bit formats: `mt` = (distance, length) of the match that starts at position p.
// (Shared by the batch kernel below and the whole-GPU path of ONE stream, alz_encode_big.h.)
template <int FMT>
__device__ __forceinline__ void flag_payload(const EncGeom& g, u32 p, uint2 mt, u32& b0, u32& b1, u32& b2, u32& b3, u32& psize) {
    const u32 d1 = (mt.x - 1u) & 0xFFFu, len = mt.y;
    if (FMT == ALZ_FMT_LZSS) {
        const u32 offset = (g.windows_start + p - mt.x) & (g.lz_max_distance - 1u);
        const u32 v = (offset & 0xFFu) | ((offset & 0xFF00u) << g.length_bits) | (((len - g.lz_min_length) & ((1u << g.length_bits) - 1u)) << 8);
        b0 = v & 0xFF; b1 = (v >> 8) & 0xFF; psize = 2;
    } else if (FMT == ALZ_FMT_CLZ0) {                          // CLZ0.cs:121-124: delta = 0x1000 - distance
        const u32 delta = 0x1000u - mt.x; b0 = delta & 0xFF; b1 = ((len - 3u) | ((delta >> 8) << 4)) & 0xFF; psize = 2;
    } else if (FMT == ALZ_FMT_BLZ) {                           // BLZ.cs:172: distance - 3
        const u32 v = (((len - 3u) << 12) | ((mt.x - 3u) & 0xFFFu)) & 0xFFFFu; b0 = v >> 8; b1 = v & 0xFF; psize = 2;
    } else if (FMT == ALZ_FMT_LZ10 || FMT == ALZ_FMT_MIO0) {
        const u32 v = (((len - 3u) << 12) | d1) & 0xFFFFu; b0 = v >> 8; b1 = v & 0xFF; psize = 2;
    } else if (FMT == ALZ_FMT_LZ40) {                          // u16 LE distance << 4 | length (+ 1 or 2 length bytes)
        const u32 dv = (mt.x << 4) & 0xFFFFu;
        if (len < 16) { const u32 v = dv | len; b0 = v & 0xFF; b1 = v >> 8; psize = 2; }
        else if (len < 272) { b0 = dv & 0xFF; b1 = dv >> 8; b2 = len - 16u; psize = 3; }
        else { const u32 v = dv | 1u, l = (len - 272u) & 0xFFFFu; b0 = v & 0xFF; b1 = v >> 8; b2 = l & 0xFF; b3 = l >> 8; psize = 4; }
    } else if (FMT == ALZ_FMT_LZ11) {
        if (len <= 16) { const u32 v = (((len - 1u) << 12) | d1) & 0xFFFFu; b0 = v >> 8; b1 = v & 0xFF; psize = 2; }
        else if (len <= 272) { b0 = ((len - 17u) & 0xFFu) >> 4; const u32 v = (((len - 17u) << 12) | d1) & 0xFFFFu; b1 = v >> 8; b2 = v & 0xFF; psize = 3; }
        else { const u32 v = 0x10000000u | (((len - 273u) & 0xFFFFu) << 12) | d1; b0 = v >> 24; b1 = (v >> 16) & 0xFF; b2 = (v >> 8) & 0xFF; b3 = v & 0xFF; psize = 4; }
    } else {   // YAZ0 / YAY0
        if (len < 18) { const u32 v = (d1 | ((len - 2u) << 12)) & 0xFFFFu; b0 = v >> 8; b1 = v & 0xFF; psize = 2; }
        else { b0 = d1 >> 8; b1 = d1 & 0xFF; b2 = len - 0x12u; psize = 3; }
    }
}

// The parse and the emitter of the flag-bit formats in ONE kernel (round 3): enc_roles_kernel's walk over a window of 64 positions, then
// the emitter's tokens of the same window, from the same registers.  As two kernels the parse wrote a start mask and the exact
// matches it had recomputed, and the emitter read mask and match array again: 10.5 GB of the 53 the pipeline moved at quality 0, and
// ~30 of the ~280 instructions the two spend per window.  The walk visits windows in order (one that lies inside a match has nothing
// to parse, the emitter still passes it: it is covered); what the window in front found for THIS window's first position travels in
// `carry`; a position kernel B had capped is recomputed exactly when the cursor stands on it and patched into the window's registers.
// SEARCH (finders that look at ONE candidate per position -- maxChain 1: quality 0 --, one property set, no min-length table, 16-bit
// links): kernel B is in here as well.  What a position's match is does not depend on the parse, so the loads are issued ahead of it:
// while window w is parsed and emitted, the matches of window w + 1 are worked out from bytes that arrived during the window before
// (the lazy rule of the last position of w needs the first of w + 1), the candidates' bytes of window w + 2 are in flight, and so
// are the links and the positions' own bytes of window w + 3: no match array at all (10.5 GB written, 13 GB read at quality 0),
// and the search runs at the parse's instruction rate instead of waiting for its own loads (0.75 instructions per cycle as a kernel).
// 32 bytes per side are compared from the prefetched registers; a longer match finishes in memory.
template <int FMT, bool SEARCH>
__global__ __launch_bounds__(64) void enc_parse_emit_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base,
                                                            const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                            u32 count, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                            const int* __restrict__ prev4, const int* __restrict__ prevm, u8* __restrict__ side,
                                                            alz_result* __restrict__ results, alz_encode_aux* __restrict__ aux, EncGeom g) {
    constexpr bool THREE = (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0);
    constexpr bool LIT_BIT = (FMT == ALZ_FMT_LZSS || FMT == ALZ_FMT_YAZ0 || FMT == ALZ_FMT_LZHUDSON || THREE);   // flag bit of a literal token
    constexpr bool MSB = (FMT != ALZ_FMT_LZSS && FMT != ALZ_FMT_CLZ0);
    constexpr u32 FBITS = FMT == ALZ_FMT_LZHUDSON ? 32u : 8u, FB = FBITS / 8u;    // LZHudson: Yay0's tokens behind 32-bit big-endian flag words  LZHudson.cs:55-59
    __shared__ u32 flagacc[16];
    __shared__ u32 gofs[16];
    __shared__ u8 hopmark[64];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)threadIdx.x;
    hopmark[lane] = 0;
    const u32 sid = index_list[bid];
    if (sid == 0xFFFFFFFFu) return;               // (a list written on the device, enc_scan_select_kernel: the stream goes the other way)
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u8* data = src;
    const u32 n = st.src_len;
    const int limit = (int)n - 4;                                        // FindNextBestMatch searches up to length - 4  :159
    u8* dst = dst_base + st.dst_off;
    const u32 cap = st.dst_cap;
    const mentry* m = match + pos_off[sid];
    const int* p4 = prev4 + pos_off[sid];
    const int* pm = g.use_min_table ? prevm + pos_off[sid] : nullptr;
    u8* compb = THREE ? side + 2 * pos_off[sid] : nullptr;            // token section (Yay0/MIO0)
    u8* uncb = THREE ? side + 2 * pos_off[sid] + n + 16 : nullptr;    // literal section
    if (lane < 16) { flagacc[lane] = 0; gofs[lane] = 0; }
    __syncthreads();
    u32 tok_base = 0;       // tokens emitted before the window
    u32 pay_base = 0;       // payload bytes before the window (THREE: token-section bytes)
    u32 unc_base = 0;       // THREE: literal-section bytes before the window
    u32 cover = 0;          // end of the last match seen so far
    bool fail = false;
    int cur = 0;            // cursor of FindNextBestMatch (absolute position)
    bool carry = false;     // the window in front found a token that starts at this window's first position
    // (match entry and source byte of a window are loaded while the window before it is worked on; positions above `limit` were never
    // searched: no match)
    auto ldm = [&](u32 q) { return (int)q <= limit ? m_unpack(__builtin_nontemporal_load(m + q)) : make_uint2(0, 0); };
    // ---- SEARCH: the stages of the look-ahead.  L: link + 32 own bytes of a window; C: the 32 bytes of its candidates.
    const unsigned short* lk16 = reinterpret_cast<const unsigned short*>(p4);
    const u32 srange = (u32)(g.max_dist - g.min_dist);
    u32 lkA = 0, lkB = 0; u64 ownA[4] = {0, 0, 0, 0}, ownB[4] = {0, 0, 0, 0}, cndB[4] = {0, 0, 0, 0};
    auto clampq = [&](u32 q) { return (int)q <= limit ? q : (u32)(limit > 0 ? limit : 0); };             // (positions behind the last searched one: loads stay inside, results unused)
    // (a position behind the last searched one has NO link: kernel A never wrote its slot -- the scratch is grow-only and never zeroed, so the
    // slot holds whatever an earlier batch left there, and loadC would dereference data - that; a stream shorter than 4 bytes has no
    // searched position at all.  The own bytes may run up to 28 bytes past the stream: the slack every source buffer has, auroralz.h)
    auto loadL = [&](u32 q, u32& lkv, u64 (&own)[4]) { const u32 qq = clampq(q); lkv = (int)q <= limit ? (u32)__builtin_nontemporal_load(lk16 + qq) : 0u; __builtin_memcpy(own, data + qq, 32); };
    auto loadC = [&](u32 q, u32 lkv, u64 (&cnd)[4]) { const u32 qq = clampq(q); const bool ok = lkv - (u32)g.min_dist <= srange; __builtin_memcpy(cnd, data + qq - (ok ? lkv : 0u), 32); };   // (a candidate out of reach is not touched)
    auto matchof = [&](u32 q, u32 lkv, const u64 (&own)[4], const u64 (&cnd)[4]) -> uint2 {
        if ((int)q > limit) return make_uint2(0, 0);
        const bool ok = lkv - (u32)g.min_dist <= srange;                // a candidate (0: none), within maxDistance, not closer than minDistance (the walk would go on -- and at maxChain 1 it is over)  :259-266
        int best_possible = (int)n - (int)q; if (best_possible > g.max_len) best_possible = g.max_len;
        // (ALZ_PARSE_CAP bytes for certain; LZ11 / LZ40 -- matches of up to 16 KiB -- up to ALZ_PARSE_CAP_HI while only a few lanes of the window are still
        // equal, as WinParse::matchof: program text 10.3 -> 8.7 ms, Test.bmp 11.3 -> 11.5.  Yaz0 with its 273 bytes: text 8.8 -> 7.8, but Test.bmp 10.5 -> 11.3
        // whatever the number of lanes -- not there.)
        constexpr int CAPHI = (FMT == ALZ_FMT_LZ11 || FMT == ALZ_FMT_LZ40) ? ALZ_PARSE_CAP_HI : ALZ_PARSE_CAP;
        int cmp_max = best_possible > CAPHI ? CAPHI : best_possible;
        const u64 x0 = own[0] ^ cnd[0], x1 = own[1] ^ cnd[1], x2 = own[2] ^ cnd[2], x3 = own[3] ^ cnd[3];
        int len = x0 ? (int)(__builtin_ctzll(x0) >> 3) : x1 ? 8 + (int)(__builtin_ctzll(x1) >> 3) : x2 ? 16 + (int)(__builtin_ctzll(x2) >> 3) : x3 ? 24 + (int)(__builtin_ctzll(x3) >> 3) : 32;
        bool go = ok && len == 32 && cmp_max > 32;
        if (__ballot(go)) {                                              // GetMatchLength behind the prefetched bytes (wave_match_tail's loop)
            const u8* pa = data + q; const u8* pb = data + q - (go ? lkv : 0u);
            int l = 32;
            const bool went = go;
            while (__ballot(go)) {
                if (CAPHI > ALZ_PARSE_CAP && __popcll(__ballot(go && l >= ALZ_PARSE_CAP)) >= ALZ_PARSE_MANY) { if (go && l >= ALZ_PARSE_CAP) { cmp_max = l; go = false; } continue; }
                const u64 z = load64(pa + (go ? l : 0)) ^ load64(pb + (go ? l : 0));
                if (go) { if (z) { l += (int)(__builtin_ctzll(z) >> 3); go = false; } else { l += 8; if (l >= cmp_max) go = false; } }
            }
            if (went) len = l;
        }
        if (len > cmp_max) len = cmp_max;
        const bool hitcap = ok && len == cmp_max && cmp_max < best_possible;
        int l2 = len;
        if (g.no_self_overlap && l2 > (int)lkv) l2 = (int)lkv;          // ScoreMatch  :301-321, one property set
        const bool take = ok && !hitcap && l2 >= g.min_len;
        return hitcap ? make_uint2(ALZ_CAPPED, ALZ_CAPPED) : make_uint2(take ? lkv : 0u, take ? (u32)l2 : 0u);
    };
    uint2 a_n;
    if (SEARCH) {
        loadL((u32)lane, lkB, ownB); loadL(64u + (u32)lane, lkA, ownA);
        loadC((u32)lane, lkB, cndB);
        a_n = matchof((u32)lane, lkB, ownB, cndB);                      // window 0
        lkB = lkA; ownB[0] = ownA[0]; ownB[1] = ownA[1]; ownB[2] = ownA[2]; ownB[3] = ownA[3];
        loadC(64u + (u32)lane, lkB, cndB);                              // window 1's candidates
        loadL(128u + (u32)lane, lkA, ownA);                             // window 2's links and own bytes
    } else a_n = ldm((u32)lane);
    u32 sb_n = (u32)lane < n ? src[lane] : 0u;
    for (u32 P = 0; P < n; P += 64) {
        const u32 p = P + (u32)lane;
        uint2 a = a_n;
        const u32 sb = sb_n;
        if (SEARCH) {
            // (LZ11 / LZ40 -- matches of up to 16 KiB --: a window the cursor has already jumped over is never looked at, its stage is left out
            // as in enc_parse_seq_kernel: 1 024 windows of Test.bmp 11.65 -> 11.23 ms, 4 096 25.0 -> 23.7.  With Yaz0's 273 bytes -3 % on the
            // bitmap and +2 % on text; the 18 bytes of LZ10 never skip a window: +1.5 %)
            constexpr bool SKIPW = FMT == ALZ_FMT_LZ11 || FMT == ALZ_FMT_LZ40;     // (for every format: tools/variants/r04_encode_switches.patch, -DALZ_PARSE_SKIP_ALL)
            if (!SKIPW || cur < (int)P + 128) a_n = matchof(p + 64u, lkB, ownB, cndB);    // window w + 1 (bytes that arrived during the window before)
            lkB = lkA; ownB[0] = ownA[0]; ownB[1] = ownA[1]; ownB[2] = ownA[2]; ownB[3] = ownA[3];
            if (!SKIPW || cur < (int)P + 192) loadC(p + 128u, lkB, cndB);                 // window w + 2's candidates
            if (!SKIPW || cur < (int)P + 256) loadL(p + 192u, lkA, ownA);                 // window w + 3's links and own bytes
        } else a_n = ldm(p + 64u);
        if (p + 64 < n) sb_n = src[p + 64];
        u64 sm = carry ? 1ull : 0ull;
        carry = false;
        if (cur < (int)P + 64 && cur <= limit) {
            // ---- the parse of this window (the body of enc_roles_kernel)
            uint2 b;
            b.x = (u32)__builtin_amdgcn_ds_bpermute((lane + 1) << 2, (int)a.x); b.y = (u32)__builtin_amdgcn_ds_bpermute((lane + 1) << 2, (int)a.y);
            {   // lane 63's neighbour is the first position of the next window
                const u32 n0x = (u32)__builtin_amdgcn_readlane((int)a_n.x, 0), n0y = (u32)__builtin_amdgcn_readlane((int)a_n.y, 0);
                if (lane == 63) b = make_uint2(n0x, n0y);
            }
            const bool capped = a.y == ALZ_CAPPED || b.y == ALZ_CAPPED;
            int jump = 1, startrel = 0;   // startrel: 0 no token here, 1 match starts here, 2 literal here + match at p + 1
            if (!capped && (int)p <= limit && (int)a.y >= g.min_len) {
                const int l0 = (int)a.y, l1 = (int)b.y, pi = (int)p;
                const bool lazyc = l0 <= g.lazy && pi + 1 <= limit;
                if (lazyc && l1 > l0) { startrel = 2; const int e = pi + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; jump = (pi + 2 > stop ? pi + 2 : stop) - pi; }
                else { startrel = 1; const int skip = lazyc ? 1 : 0; const int e = pi + l0; const int stop = e < limit + 1 ? e : limit + 1; jump = (pi + 1 + skip > stop ? pi + 1 + skip : stop) - pi; }
            }
            int rel = cur - (int)P;
            if (__ballot(capped) == 0) {
                u64 M = 0; u32 r = (u32)rel, j;
                const u32 lim = (u32)(limit + 1 - (int)P) < 64u ? (u32)(limit + 1 - (int)P) : 64u;     // (r < lim on entry: cur <= limit)
                if (lim == 64u) {
                    // Two tokens per hop: jump2 = my jump + the jump of where it lands (one ds_bpermute), so the scalar loop -- four
                    // instructions per hop, ~18 tokens per window: the largest single item of this kernel -- runs half as often.  It
                    // marks every second visited lane; the ones in between are where the marked lanes' own jumps land: scattered
                    // through 64 bytes of LDS.  (The add's carry is "left the window": enc_roles_kernel.)
                    const u32 tgt = (u32)lane + (u32)jump;                     // where my jump lands (>= 64: outside)
                    const int jn = __builtin_amdgcn_ds_bpermute((int)((tgt < 64u ? tgt : (u32)lane) << 2), jump);
                    const int jump2 = tgt < 64u ? jump + jn : jump;
                    r -= 64u;
                    asm volatile(
                        "s_nop 3\n"
                        "1:\n\t"
                        "s_bitset1_b64 %[M], %[r]\n\t"
                        "v_readlane_b32 %[j], %[jump], %[r]\n\t"
                        "s_add_u32 %[r], %[r], %[j]\n\t"
                        "s_cbranch_scc0 1b\n\t"
                        : [M] "+s"(M), [r] "+s"(r), [j] "=&s"(j)
                        : [jump] "v"(jump2)
                        : "scc");
                    r += 64u;
                    if (((M >> lane) & 1ull) && tgt < 64u) hopmark[tgt] = 1;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const u32 hm = hopmark[lane];
                    if (hm) hopmark[lane] = 0;
                    M |= __ballot(hm != 0u);
                } else
                asm volatile(
                    "s_nop 3\n"
                    "1:\n\t"
                    "s_bitset1_b64 %[M], %[r]\n\t"
                    "v_readlane_b32 %[j], %[jump], %[r]\n\t"
                    "s_add_u32 %[r], %[r], %[j]\n\t"
                    "s_cmp_lt_u32 %[r], %[lim]\n\t"
                    "s_cbranch_scc1 1b\n\t"
                    : [M] "+s"(M), [r] "+s"(r), [j] "=&s"(j)
                    : [jump] "v"(jump), [lim] "s"(lim)
                    : "scc");
                const u64 s1 = __ballot(startrel == 1) & M, s2 = __ballot(startrel == 2) & M;
                sm |= s1 | (s2 << 1);
                if (s2 >> 63) carry = true;                                    // start in lane 0 of the next window
                rel = (int)r;
            }
            else while (rel < 64 && (int)P + rel <= limit) {
                int j, sr;
                if (__builtin_amdgcn_readlane((int)capped, rel)) {
                    // kernel B capped a candidate here: redo MatchSearch exactly for this cursor and its lazy neighbour, and put both
                    // into the registers the emitter takes its matches from (the neighbour may be the next window's first position)
                    const int q = (int)P + rel;
                    int d0, l0, d1 = 0, l1 = 0; bool s0, s1;
                    const uint2 e0 = make_uint2((u32)__builtin_amdgcn_readlane((int)a.x, rel), (u32)__builtin_amdgcn_readlane((int)a.y, rel));
                    const uint2 e1 = rel + 1 < 64 ? make_uint2((u32)__builtin_amdgcn_readlane((int)a.x, (rel + 1) & 63), (u32)__builtin_amdgcn_readlane((int)a.y, (rel + 1) & 63))
                                                  : make_uint2((u32)__builtin_amdgcn_readlane((int)a_n.x, 0), (u32)__builtin_amdgcn_readlane((int)a_n.y, 0));
                    if (g.use_min_table) benc_capped_cursor<true>(data, (int)n, g, p4, pm, q, limit, e0, e0.y == ALZ_CAPPED, e1, e1.y == ALZ_CAPPED, d0, l0, d1, l1, s0, s1);
                    else benc_capped_cursor<false>(data, (int)n, g, p4, pm, q, limit, e0, e0.y == ALZ_CAPPED, e1, e1.y == ALZ_CAPPED, d0, l0, d1, l1, s0, s1);
                    if (s0 && lane == rel) a = make_uint2((u32)d0, (u32)l0);
                    if (s1) {
                        if (rel + 1 < 64) { if (lane == rel + 1) a = make_uint2((u32)d1, (u32)l1); }
                        else if (lane == 0) a_n = make_uint2((u32)d1, (u32)l1);
                    }
                    j = 1; sr = 0;
                    if (l0 >= g.min_len) {
                        const bool lazyc = l0 <= g.lazy && q + 1 <= limit;
                        if (lazyc && l1 > l0) { sr = 2; const int e = q + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; j = (q + 2 > stop ? q + 2 : stop) - q; }
                        else { sr = 1; const int skip = lazyc ? 1 : 0; const int e = q + l0; const int stop = e < limit + 1 ? e : limit + 1; j = (q + 1 + skip > stop ? q + 1 + skip : stop) - q; }
                    }
                } else {
                    j = __builtin_amdgcn_readlane(jump, rel);
                    sr = __builtin_amdgcn_readlane(startrel, rel);
                }
                if (sr == 1) sm |= 1ull << rel;
                else if (sr == 2) { if (rel + 1 < 64) sm |= 1ull << (rel + 1); else carry = true; }   // start in lane 0 of the next window
                rel += j;
            }
            cur = (int)P + rel;
        }
        // ---- the tokens of this window: prefix sums over the start mask give every token its flag group and byte offset
        const bool start = ((sm >> lane) & 1ull) && p < n;
        uint2 mt = make_uint2(0, 0);
        if (start) mt = a;
        const u32 mend = start ? p + mt.y : 0u;
        const u32 pmax = scan_max(mend);                               // inclusive
        u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)pmax, 0x138, 0xF, 0xF, false);   // wave_shr:1 -> max over lanes below
        if (before < cover) before = cover;
        const bool lit = !start && p < n && p >= before;
        const bool tok = start || lit;
        const u64 tm = __ballot(tok);
        const u32 ti = tok_base + __builtin_amdgcn_mbcnt_hi((u32)(tm >> 32), __builtin_amdgcn_mbcnt_lo((u32)tm, 0u));
        // payload of my token
        u32 b0 = 0, b1 = 0, b2 = 0, b3 = 0, psize = 0, usize = 0;
        if (lit) { b0 = sb; psize = 1; }
        else if (start) flag_payload<FMT>(g, p, mt, b0, b1, b2, b3, psize);
        if (THREE) {   // literals (and Yay0's long-length byte) live in their own section
            if (lit) { usize = 1; psize = 0; }
            else if (start && FMT == ALZ_FMT_YAY0 && psize == 3) { usize = 1; psize = 2; }
        }
        const u32 pincl = scan_add(psize);
        const u32 poff = pay_base + pincl - psize;
        const u32 uincl = THREE ? scan_add(usize) : 0u;
        const u32 uoff = unc_base + uincl - usize;
        const u32 group = ti / FBITS, bitpos = ti % FBITS;
        // flag bytes: the first token of a group fixes the flag byte's position, the last one stores it
        const u32 flag_off = THREE ? group : poff + FB * group;        // interleaved: flag g sits right before the payload of its first token
        if (tok && bitpos == 0) { gofs[group & 15u] = flag_off; flagacc[group & 15u] = 0; }
        __syncthreads();
        if (tok) {
            const u32 bitv = (lit ? LIT_BIT : !LIT_BIT) ? 1u : 0u;
            if (bitv) atomicOr(&flagacc[group & 15u], 1u << (MSB ? FBITS - 1u - bitpos : bitpos));
        }
        __syncthreads();
        if (tok) {
            if (bitpos == FBITS - 1u) {
                const u32 fo = gofs[group & 15u], acc = flagacc[group & 15u];
                if (fo + FB <= cap) { if (FB == 1u) dst[fo] = (u8)(FMT == ALZ_FMT_LZ40 ? 0u - acc : acc); else { dst[fo] = (u8)(acc >> 24); dst[fo + 1] = (u8)(acc >> 16); dst[fo + 2] = (u8)(acc >> 8); dst[fo + 3] = (u8)acc; } }
                else fail = true;
            }
            if (!THREE) {
                const u32 o = poff + FB * (group + 1u);
                if (o + psize <= cap) { dst[o] = (u8)b0; if (psize > 1) dst[o + 1] = (u8)b1; if (psize > 2) dst[o + 2] = (u8)b2; if (psize > 3) dst[o + 3] = (u8)b3; }
                else fail = true;
            } else {
                if (lit) uncb[uoff] = (u8)b0;
                else { compb[poff] = (u8)b0; compb[poff + 1] = (u8)b1; if (usize) uncb[uoff] = (u8)b2; }
            }
        }
        __syncthreads();
        tok_base += (u32)__popcll(tm);
        pay_base += (u32)__builtin_amdgcn_readlane((int)pincl, 63);
        if (THREE) unc_base += (u32)__builtin_amdgcn_readlane((int)uincl, 63);
        const u32 wmax = (u32)__builtin_amdgcn_readlane((int)pmax, 63);
        if (wmax > cover) cover = wmax;
    }
    // Dispose(): a partial flag byte is written with its unused bits zero (FlagWriter.cs:141-145)
    const u32 nflags = FB * ((tok_base + FBITS - 1u) / FBITS);          // (bytes)
    if ((tok_base % FBITS) != 0 && lane == 0) {
        const u32 gi = tok_base / FBITS; const u32 fo = gofs[gi & 15u], acc = flagacc[gi & 15u];
        if (fo + FB <= cap) { if (FB == 1u) dst[fo] = (u8)(FMT == ALZ_FMT_LZ40 ? 0u - acc : acc); else { dst[fo] = (u8)(acc >> 24); dst[fo + 1] = (u8)(acc >> 16); dst[fo + 2] = (u8)(acc >> 8); dst[fo + 3] = (u8)acc; } }
        else fail = true;
    }
    u32 total;
    if (!THREE) total = pay_base + nflags;
    else {
        total = nflags + pay_base + unc_base;
        if (total <= cap) {
            for (u32 i = (u32)lane; i < pay_base; i += 64) dst[nflags + i] = compb[i];
            for (u32 i = (u32)lane; i < unc_base; i += 64) dst[nflags + pay_base + i] = uncb[i];
        } else fail = true;
        if (lane == 0 && aux) { aux[sid].aux0 = nflags; aux[sid].aux1 = nflags + pay_base; }
    }
    if (total > cap) fail = true;
    const bool anyfail = __ballot(fail) != 0;
    if (lane == 0) {
        alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
        results[sid] = r;
        if (!THREE && aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
    }
}

// The streams enc_scan_select_kernel picked (round 6): parse + emission WITHOUT links or a match array, one wavefront per stream.  The walk is FindNextBestMatch itself
// (:157-212): the position the cursor stands on -- and its neighbour where the lazy rule looks at it (:175-190) -- is searched exactly by the whole wavefront
// (benc_wave_scan_search), and what it finds goes into a list of tokens, one per lane: a match (position, distance, length) or a literal (a position the cursor stepped
// over, or one in front of a lazily taken match, or what is left behind the last match).  64 tokens are emitted at once, with enc_parse_emit_kernel's arithmetic: flag group
// and bit from the token's number, payload offsets by prefix sums, a flag byte in front of its group's first payload.  (The first form of this path walked
// enc_parse_emit_kernel's 64-POSITION windows with every position "capped": one emission pass per token where tokens lie a hundred bytes apart -- 2 000 flat windows 18.5 ms.)
template <int FMT>
__global__ __launch_bounds__(64) void enc_scan_emit_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                           const u32* __restrict__ index_list, u32 count, const u64* __restrict__ pos_off, u8* __restrict__ side,
                                                           alz_result* __restrict__ results, alz_encode_aux* __restrict__ aux, EncGeom g) {
    constexpr bool THREE = (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0);
    constexpr bool LIT_BIT = (FMT == ALZ_FMT_LZSS || FMT == ALZ_FMT_YAZ0 || FMT == ALZ_FMT_LZHUDSON || THREE);   // flag bit of a literal token
    constexpr bool MSB = (FMT != ALZ_FMT_LZSS && FMT != ALZ_FMT_CLZ0);
    constexpr u32 FBITS = FMT == ALZ_FMT_LZHUDSON ? 32u : 8u, FB = FBITS / 8u;
    __shared__ u32 flagacc[16];
    __shared__ u32 gofs[16];
    __shared__ unsigned short candl[64];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const u32 sid = index_list[bid];
    if (sid == 0xFFFFFFFFu) return;               // (the stream went the other way)
    const int lane = (int)threadIdx.x;
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const u32 n = st.src_len;
    const int limit = (int)n - 4;                                        // FindNextBestMatch searches up to length - 4  :159
    u8* dst = dst_base + st.dst_off;
    const u32 cap = st.dst_cap;
    u8* compb = THREE ? side + 2 * pos_off[sid] : nullptr;            // token section (Yay0 / MIO0)
    u8* uncb = THREE ? side + 2 * pos_off[sid] + n + 16 : nullptr;    // literal section
    if (lane < 16) { flagacc[lane] = 0; gofs[lane] = 0; }
    __syncthreads();
    u32 tok_base = 0, pay_base = 0, unc_base = 0;
    bool fail = false;
    int cur = 0;                                                         // the cursor (Position)
    u32 cover = 0;                                                       // end of the last match
    u32 tail = 0;                                                        // behind the walk: the next position of what is left (literals)
    bool walk = limit >= 0;
    for (;;) {
        // ---- up to 64 tokens: lane k holds token k -- (position, distance, length), length 0 = a literal
        u32 tp = 0; uint2 tm2 = make_uint2(0, 0);
        int k = 0;
        while (walk && k <= 62) {
            if (cur > limit) { walk = false; tail = (u32)cur > cover ? (u32)cur : cover; break; }
            int d0, l0, d1 = 0, l1 = 0;
            benc_wave_scan_search(data, (int)n, g, cur, d0, l0, candl);
            if (l0 < g.min_len) { if (lane == k) { tp = (u32)cur; tm2 = make_uint2(0, 0); } k++; cur++; continue; }      // :166-170
            const bool lazyc = l0 <= g.lazy && cur + 1 <= limit;
            if (lazyc) benc_wave_scan_search(data, (int)n, g, cur + 1, d1, l1, candl);
            int mp = cur, md = d0, ml = l0, skip = lazyc ? 1 : 0;
            if (lazyc && l1 > l0) { if (lane == k) { tp = (u32)cur; tm2 = make_uint2(0, 0); } k++; mp = cur + 1; md = d1; ml = l1; skip = 0; }   // the byte in front becomes a literal  :181-186
            if (lane == k) { tp = (u32)mp; tm2 = make_uint2((u32)md, (u32)ml); }
            k++;
            const int e = mp + ml, stop = e < limit + 1 ? e : limit + 1;  // :195-203
            cur = mp + 1 + skip > stop ? mp + 1 + skip : stop;
            cover = (u32)e;
        }
        if (!walk) {
            // what is left behind the last match: literals (the last three bytes are never searched; a stream shorter than four bytes is all of this)
            while (k < 64 && tail < n) { if (lane == k) { tp = tail; tm2 = make_uint2(0, 0); } k++; tail++; }
        }
        if (k == 0) break;
        // ---- the tokens of this batch (the arithmetic of enc_parse_emit_kernel: prefix sums over the tokens)
        const bool tok = lane < k;
        const bool start = tok && tm2.y != 0u, lit = tok && tm2.y == 0u;
        const u32 ti = tok_base + (u32)lane;
        u32 b0 = 0, b1 = 0, b2 = 0, b3 = 0, psize = 0, usize = 0;
        if (lit) { b0 = data[tp]; psize = 1; }
        else if (start) flag_payload<FMT>(g, tp, tm2, b0, b1, b2, b3, psize);
        if (THREE) {   // literals (and Yay0's long-length byte) live in their own section
            if (lit) { usize = 1; psize = 0; }
            else if (start && FMT == ALZ_FMT_YAY0 && psize == 3) { usize = 1; psize = 2; }
        }
        const u32 pincl = scan_add(psize);
        const u32 poff = pay_base + pincl - psize;
        const u32 uincl = THREE ? scan_add(usize) : 0u;
        const u32 uoff = unc_base + uincl - usize;
        const u32 group = ti / FBITS, bitpos = ti % FBITS;
        const u32 flag_off = THREE ? group : poff + FB * group;        // interleaved: flag g sits right before the payload of its first token
        if (tok && bitpos == 0) { gofs[group & 15u] = flag_off; flagacc[group & 15u] = 0; }
        __syncthreads();
        if (tok) {
            const u32 bitv = (lit ? LIT_BIT : !LIT_BIT) ? 1u : 0u;
            if (bitv) atomicOr(&flagacc[group & 15u], 1u << (MSB ? FBITS - 1u - bitpos : bitpos));
        }
        __syncthreads();
        if (tok) {
            if (bitpos == FBITS - 1u) {
                const u32 fo = gofs[group & 15u], acc = flagacc[group & 15u];
                if (fo + FB <= cap) { if (FB == 1u) dst[fo] = (u8)(FMT == ALZ_FMT_LZ40 ? 0u - acc : acc); else { dst[fo] = (u8)(acc >> 24); dst[fo + 1] = (u8)(acc >> 16); dst[fo + 2] = (u8)(acc >> 8); dst[fo + 3] = (u8)acc; } }
                else fail = true;
            }
            if (!THREE) {
                const u32 o = poff + FB * (group + 1u);
                if (o + psize <= cap) { dst[o] = (u8)b0; if (psize > 1) dst[o + 1] = (u8)b1; if (psize > 2) dst[o + 2] = (u8)b2; if (psize > 3) dst[o + 3] = (u8)b3; }
                else fail = true;
            } else {
                if (lit) uncb[uoff] = (u8)b0;
                else { compb[poff] = (u8)b0; compb[poff + 1] = (u8)b1; if (usize) uncb[uoff] = (u8)b2; }
            }
        }
        __syncthreads();
        tok_base += (u32)k;
        pay_base += (u32)__builtin_amdgcn_readlane((int)pincl, 63);
        if (THREE) unc_base += (u32)__builtin_amdgcn_readlane((int)uincl, 63);
        if (!walk && tail >= n) break;
    }
    // Dispose(): a partial flag byte is written with its unused bits zero (FlagWriter.cs:141-145)
    const u32 nflags = FB * ((tok_base + FBITS - 1u) / FBITS);          // (bytes)
    if ((tok_base % FBITS) != 0 && lane == 0) {
        const u32 gi = tok_base / FBITS; const u32 fo = gofs[gi & 15u], acc = flagacc[gi & 15u];
        if (fo + FB <= cap) { if (FB == 1u) dst[fo] = (u8)(FMT == ALZ_FMT_LZ40 ? 0u - acc : acc); else { dst[fo] = (u8)(acc >> 24); dst[fo + 1] = (u8)(acc >> 16); dst[fo + 2] = (u8)(acc >> 8); dst[fo + 3] = (u8)acc; } }
        else fail = true;
    }
    u32 total;
    if (!THREE) total = pay_base + nflags;
    else {
        total = nflags + pay_base + unc_base;
        if (total <= cap) {
            for (u32 i = (u32)lane; i < pay_base; i += 64) dst[nflags + i] = compb[i];
            for (u32 i = (u32)lane; i < unc_base; i += 64) dst[nflags + pay_base + i] = uncb[i];
        } else fail = true;
        if (lane == 0 && aux) { aux[sid].aux0 = nflags; aux[sid].aux1 = nflags + pay_base; }
    }
    if (total > cap) fail = true;
    const bool anyfail = __ballot(fail) != 0;
    if (lane == 0) {
        alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
        results[sid] = r;
        if (!THREE && aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
    }
}

// Where a segment has no synchronisation point in front of it, the cursor that enters it still follows from the one that entered the segment before:
// exit(e) = the first cursor at or behind the next boundary B on the chain from e, for every e a jump can land on behind the boundary in front (the
// first `hist` positions of the segment).  One wavefront per segment sweeps it right to left: exit(p) = p + jump(p) if that is >= B, else exit(p + jump(p)) --
// from a ring in LDS when the target lies in a window already done, by pointer doubling among the 64 lanes when it lies in the same window.  If the next
// boundary HAS a synchronisation point s, the sweep stops at the window that holds s and hands exit(s) over directly: no chain through this segment.
// enc_compose_kernel then strings the entries together (one table lookup per boundary without a synchronisation point), and every wavefront of
// enc_roles_kernel<true> walks exactly one segment: from the cursor that enters it to the cursor that enters the next.
#define ALZ_ENTRY_NONE 0xFFFFFFFFu
#define ALZ_ENTRY_TABLE 0xFFFFFFFEu
__global__ __launch_bounds__(64) void enc_exit_kernel(const alz_stream* __restrict__ streams, const u32* __restrict__ index_list, const mentry* __restrict__ match,
                                                      const u64* __restrict__ pos_off, const u32* __restrict__ sync, u32* __restrict__ direct, u32* __restrict__ ftab,
                                                      u32 kpitch, u32 seglen, u32 hist, EncGeom g) {
    __shared__ u32 ring[4096];
    const u32 k = blockIdx.x, bid = blockIdx.y;
    const int lane = (int)threadIdx.x;
    if (k + 1u >= kpitch) return;
    const u32 sid = index_list[bid];
    const int n = (int)streams[sid].src_len;
    const int limit = n - 4;
    const size_t o = (size_t)bid * kpitch + k + 1u;
    const int S = (int)(k * seglen), B = S + (int)seglen;
    if (B > limit) { if (lane == 0) direct[o] = ALZ_ENTRY_NONE; return; }
    const u32 s = sync[o];
    if (s == (u32)B) { if (lane == 0) direct[o] = (u32)B; return; }
    const mentry* m = match + pos_off[sid];
    const int lo = s != ALZ_ENTRY_NONE ? (int)(s & ~63u) : S;
    auto ldm = [&](int q) { return q <= limit ? m_unpack(m[q]) : make_uint2(0, 0); };
    bool bad = false;
    u32 dval = ALZ_ENTRY_NONE;
    uint2 a_n = ldm(B - 64 + lane), b_n = ldm(B - 64 + lane + 1);
    for (int P = B - 64; P >= lo; P -= 64) {
        const int p = P + lane;
        const uint2 a = a_n, b = b_n;
        if (P - 64 >= lo) { a_n = ldm(p - 64); b_n = ldm(p - 63); }
        if (__ballot(a.y == ALZ_CAPPED || b.y == ALZ_CAPPED)) { bad = true; break; }
        int jump = 1;
        if ((int)a.y >= g.min_len) {                                    // (p < B <= limit: searched)
            const int l0 = (int)a.y, l1 = (int)b.y;
            const bool lazyc = l0 <= g.lazy && p + 1 <= limit;
            if (lazyc && l1 > l0) { const int e = p + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; jump = (p + 2 > stop ? p + 2 : stop) - p; }
            else { const int skip = lazyc ? 1 : 0; const int e = p + l0; const int stop = e < limit + 1 ? e : limit + 1; jump = (p + 1 + skip > stop ? p + 1 + skip : stop) - p; }
        }
        const int t = p + jump;
        u32 val = (u32)t; int res = 1, tl = lane;
        if (t < B) { if (t >= P + 64) val = ring[t & 4095]; else { res = 0; tl = t - P; } }
        while (__ballot(res == 0)) {
            const u32 tv = (u32)__builtin_amdgcn_ds_bpermute(tl << 2, (int)val);
            const int tr = __builtin_amdgcn_ds_bpermute(tl << 2, res), tt = __builtin_amdgcn_ds_bpermute(tl << 2, tl);
            if (res == 0) { if (tr) { val = tv; res = 1; } else tl = tt; }
        }
        ring[p & 4095] = val;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (s != ALZ_ENTRY_NONE && P == lo) dval = (u32)__builtin_amdgcn_readlane((int)val, (int)(s - (u32)P));
    }
    if (s != ALZ_ENTRY_NONE) { if (lane == 0) direct[o] = bad ? ALZ_ENTRY_NONE : dval; return; }
    if (lane == 0) direct[o] = bad ? ALZ_ENTRY_NONE : ALZ_ENTRY_TABLE;
    if (!bad) {
        u32* tab = ftab + ((size_t)bid * kpitch + k) * hist;
        for (u32 i = (u32)lane; i < hist; i += 64) tab[i] = ring[((u32)S + i) & 4095u];      // (hist <= seglen: all of them swept)
    }
}

// the cursor that enters every segment: 0; exit(synchronisation point) where the boundary has one; else the table of the segment in front at the cursor that entered IT
__global__ __launch_bounds__(64) void enc_compose_kernel(const u32* __restrict__ direct, const u32* __restrict__ ftab, u32* __restrict__ entry, u32 kpitch, u32 seglen, u32 hist) {
    __shared__ u32 d[8192];
    const u32 bid = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const size_t base = (size_t)bid * kpitch;
    for (u32 k = (u32)lane; k < kpitch; k += 64) d[k] = k ? direct[base + k] : 0u;
    __syncthreads();
    if (lane != 0) return;
    u32 c = 0; bool known = true;
    entry[base] = 0u;
    for (u32 k = 1; k < kpitch; k++) {
        const u32 v = d[k];
        u32 e = ALZ_ENTRY_NONE;
        if (v == ALZ_ENTRY_TABLE) {
            if (known) { const u32 i = c - (k - 1u) * seglen; e = i < hist ? ftab[(base + k - 1u) * hist + i] : ALZ_ENTRY_NONE; }
            if (e == ALZ_ENTRY_NONE) known = false; else c = e;
        } else if (v != ALZ_ENTRY_NONE) { e = v; c = v; known = true; }
        else known = false;                                             // (behind the last searched position, or a capped entry in the segment: the walk in front carries on)
        entry[base + k] = e;
    }
}

#include "alz_encode_seg.h"

int isqrt_floor(int v) { int r = 0; while ((r + 1) * (r + 1) <= v) r++; return r; }

}  // namespace

// quality -> finder parameters  LzChainMatchFinder.cs:108-119 ; per-format LzProperties
bool alz_encode_geometry(int fmt, const alz_lz_properties* lz, const alz_settings* st, void* out_geom, int* window_bits, int variant) {
    EncGeom g; memset(&g, 0, sizeof(g));
    int wb = 12;
    switch (fmt) {
    case ALZ_FMT_LZSS:
        wb = lz->window_bits; g.min_len = lz->min_length; g.max_len = (1 << lz->length_bits) + lz->min_length - 1; g.max_dist = (int)lz->max_distance; break;
    case ALZ_FMT_LZ10: case ALZ_FMT_MIO0: case ALZ_FMT_SMSR00: case ALZ_FMT_CLZ0: g.min_len = 3; g.max_len = 18; g.max_dist = 0x1000; break;
    case ALZ_FMT_LZ11: case ALZ_FMT_LZ40: g.min_len = 3; g.max_len = 0x4000; g.max_dist = 0x1000; break;
    case ALZ_FMT_YAZ0: case ALZ_FMT_YAY0: case ALZ_FMT_LZHUDSON: g.min_len = 3; g.max_len = 0xff + 0x12; g.max_dist = 0x1000; break;
    case ALZ_FMT_PRS_BE: case ALZ_FMT_PRS_LE: wb = 13; g.min_len = 2; g.max_len = 0x100; g.max_dist = 0x1FFF; break;
    case ALZ_FMT_LZ4_BLOCK: wb = 16; g.min_len = 4; g.max_len = 0x7FFFFFFF; g.max_dist = 0xFFFF; break;
    case ALZ_FMT_LZO: wb = 16; g.min_len = 3; g.max_len = 0x7FFFFFFF; g.max_dist = 0xBFFF; break;
    case ALZ_FMT_SNAPPY_RAW: wb = 15; g.min_len = 4; g.max_len = 64; g.max_dist = 0x8000; break;
    case ALZ_FMT_FASTLZ:
        if (variant == 1) {                                                                                 // level 2: two sets  FastLZ.cs:23-27 (WindowsBits = ceil(log2(0x11FFF)) = 17)
            wb = 17; g.min_len = 3; g.max_len = 0x7FFFFFFF; g.max_dist = 0x11FFF; g.nprops = 2; g.variant = 1;
            g.p_max_dist[0] = 0x1FFF;  g.p_max_len[0] = 0x7FFFFFFF; g.p_min_len[0] = 3;
            g.p_max_dist[1] = 0x11FFF; g.p_max_len[1] = 0x7FFFFFFF; g.p_min_len[1] = 5;
            g.p_min_dist[0] = g.p_min_dist[1] = 1;
        } else { wb = 13; g.min_len = 3; g.max_len = 255 + 3 + 6; g.max_dist = 0x2000; }                    // level 1  FastLZ.cs:22
        break;
    case ALZ_FMT_CNX2: wb = 11; g.min_len = 4; g.max_len = 0x1F + 4; g.max_dist = 0x800; break;             // CNX2.cs:25
    case ALZ_FMT_CNS: wb = 8; g.min_len = 3; g.max_len = 130; g.max_dist = 0x100; break;                    // CNS.cs:24
    case ALZ_FMT_LZ02: g.min_len = 3; g.max_len = 272; g.max_dist = 0xFFF; break;                         // LZ02.cs:23
    case ALZ_FMT_LZSHREK: g.min_len = 3; g.max_len = 262; g.max_dist = 0x1000; break;                     // LZShrek.cs:20
    case ALZ_FMT_HIG: wb = 15; g.min_len = 4; g.max_len = 0xFFFF; g.max_dist = 0x7FFF; break;               // HIG.cs:28
    case ALZ_FMT_WFLZ: case ALZ_FMT_WFLZ_BE: wb = 16; g.min_len = 5; g.max_len = 255; g.max_dist = 0xFFFF; break;   // WFLZ.cs:20
    case ALZ_FMT_REFPACK:                                                                                   // RefPack.cs:29-34: three sets; the globals are the loosest of each (LzChainMatchFinder.cs:55-69)
        wb = 17; g.min_len = 3; g.max_len = 1028; g.max_dist = 0x20000; g.nprops = 3;
        g.p_max_dist[0] = 0x20000; g.p_max_len[0] = 1028; g.p_min_len[0] = 5;
        g.p_max_dist[1] = 0x4000;  g.p_max_len[1] = 67;   g.p_min_len[1] = 4;
        g.p_max_dist[2] = 0x400;   g.p_max_len[2] = 10;   g.p_min_len[2] = 3;
        g.p_min_dist[0] = g.p_min_dist[1] = g.p_min_dist[2] = 1;
        break;
    case ALZ_FMT_BLZ: g.min_len = 3; g.max_len = 18; g.max_dist = 0x1000; break;                          // BLZ.cs:24 (+ minDistance 3 below)
    default: return false;
    }
    if (st->max_window_bits != 0) {                                  // LzChainMatchFinder.cs:70-74 (the host only lets FastLZ through)
        if (wb < st->max_window_bits) wb = st->max_window_bits;
        if (g.max_dist < (1 << st->max_window_bits)) g.max_dist = 1 << st->max_window_bits;
    }
    g.min_dist = st->min_distance > 0 ? st->min_distance : (fmt == ALZ_FMT_BLZ ? 3 : 1);
    const int q = st->quality;
    g.max_chain = q < 6 ? q + 1 : q >= 11 ? 1 << (q - 5) : ((1 << (q >> 1)) | ((1 << (q >> 1)) >> (q & 1)));
    g.lazy = 3 + q / 3;
    g.hash_bits = 15 + isqrt_floor(2 * q);
    g.no_self_overlap = st->strategy & 1;
    g.use_min_table = (q >= 10 && g.min_len < 4) ? 1 : 0;
    g.min_mask = g.use_min_table ? (0xFFFFFFFFu >> ((4 - g.min_len) * 8)) : 0u;
    g.length_bits = lz->length_bits; g.lz_min_length = lz->min_length; g.windows_start = lz->windows_start; g.lz_max_distance = lz->max_distance;
    // the chain table of the reference has 1 << min(17 + floor(sqrt(2q)), windowsBits) slots; the prev() identity needs
    // it to cover maxDistance, which holds for every format geometry of the reference
    const int chain_bits = (17 + isqrt_floor(2 * q)) < wb ? (17 + isqrt_floor(2 * q)) : wb;
    if (g.max_chain != 1 && (1 << chain_bits) < g.max_dist) return false;
    // (16-bit links: 67 -> 57 GB of HBM traffic per 10 000 x 256 KiB at quality 0, 217 -> 143 at quality 8, times within 1 %; at quality 15 --
    // chains of up to 1 024 links -- they cost kernel B 5 % while the conversion sat right behind the load, and gain 1 % (104.9 -> 103.6 ms)
    // since the dense kernel converts a link at the top of the NEXT trip)
    g.link16 = g.max_dist <= 0xFFFF ? 1 : 0;
    g.b_cap = ALZ_LEN_CAP;
    if (g.max_dist > (int)ALZ_M_DMASK) return false;                  // a distance has 21 bits in the match array (FastLZ with MaxWindowBits above 20: the caller's own encoder)
    memcpy(out_geom, &g, sizeof(g));
    if (window_bits) *window_bits = wb;
    return true;
}

size_t alz_encode_geom_size(void) { return sizeof(EncGeom); }
int alz_encode_geom_hash_bits(const void* geom) { return ((const EncGeom*)geom)->hash_bits; }
int alz_encode_geom_min_table(const void* geom) { return ((const EncGeom*)geom)->use_min_table; }
int alz_encode_geom_max_dist(const void* geom) { return ((const EncGeom*)geom)->max_dist; }

// LZ4 blocks (LZ4.cs:202-238) and raw Snappy (Snappy.cs:124-203) from the start mask of the roles walk: every match start is one
// sequence -- LZ4: token, literal-length bytes, the literals since the match before it, offset, match-length bytes; Snappy: a literal
// element (tag, 0-4 length bytes, the literals) if there are any, then a copy element of two or three bytes -- whose size follows from
// the two lengths and the distance, so a prefix sum over the window places them all; the literals of the sequences (and the
// literal-only end) are copied by the whole wavefront.  The serial form (one lane per stream, a byte at a time) took 161 ms (LZ4) per
// 10 000 x 256 KiB.
__device__ __forceinline__ u32 lz4_extn(u32 v) { return v >= 15u ? 1u + (v - 15u) / 255u : 0u; }    // bytes of LZ4.WriteExtension  LZ4.cs:254-268
__device__ __forceinline__ void wave_copy(u8* d, const u8* s, u32 len, int lane) {
    u32 i = 0;
    for (; i + 256u <= len; i += 256u) { const u32 v = load32(s + i + 4u * (u32)lane); __builtin_memcpy(d + i + 4u * (u32)lane, &v, 4); }
    for (u32 j = i + (u32)lane; j < len; j += 64u) d[j] = s[j];
}
template <int FMT> struct SeqFmt;
template <> struct SeqFmt<ALZ_FMT_LZ4_BLOCK> {
    static __device__ __forceinline__ u32 lit_hdr(u32 L) { return 1u + lz4_extn(L); }                         // the token and the literal-length bytes
    static __device__ __forceinline__ u32 match_size(u32, u32 M) { return 2u + lz4_extn(M - 4u); }
    static __device__ __forceinline__ void put_lit_hdr(u8* q, u32 L, u32 M, bool last) {
        *q++ = (u8)(((L > 15u ? 15u : L) << 4) | (last ? 0u : (M - 4u > 15u ? 15u : M - 4u)));
        if (L >= 15u) { u32 v = L - 15u; while (v >= 255u) { *q++ = 255; v -= 255u; } *q++ = (u8)v; }
    }
    static __device__ __forceinline__ void put_match(u8* q, u32 D, u32 M) {
        *q++ = (u8)(D & 0xFFu); *q++ = (u8)((D >> 8) & 0xFFu);
        if (M - 4u >= 15u) { u32 v = M - 4u - 15u; while (v >= 255u) { *q++ = 255; v -= 255u; } *q++ = (u8)v; }
    }
};
template <> struct SeqFmt<ALZ_FMT_SNAPPY_RAW> {
    static __device__ __forceinline__ u32 lit_hdr(u32 L) {                                                     // Snappy.cs:160-186
        if (L == 0u) return 0u;
        const u32 len = L - 1u;
        return L <= 60u ? 1u : len <= 0xFFu ? 2u : len <= 0xFFFFu ? 3u : len <= 0xFFFFFFu ? 4u : 5u;
    }
    static __device__ __forceinline__ u32 match_size(u32 D, u32 M) { return (D < 2048u && M >= 4u && M <= 11u) ? 2u : 3u; }    // :188-201
    static __device__ __forceinline__ void put_lit_hdr(u8* q, u32 L, u32, bool) {
        if (L == 0u) return;
        const u32 len = L - 1u;
        if (L <= 60u) *q = (u8)(len << 2);
        else if (len <= 0xFFu) { q[0] = 60u << 2; q[1] = (u8)len; }
        else if (len <= 0xFFFFu) { q[0] = 61u << 2; q[1] = (u8)(len & 0xFFu); q[2] = (u8)(len >> 8); }
        else if (len <= 0xFFFFFFu) { q[0] = 62u << 2; q[1] = (u8)(len & 0xFFu); q[2] = (u8)((len >> 8) & 0xFFu); q[3] = (u8)(len >> 16); }
        else { q[0] = 63u << 2; q[1] = (u8)(len & 0xFFu); q[2] = (u8)((len >> 8) & 0xFFu); q[3] = (u8)((len >> 16) & 0xFFu); q[4] = (u8)(len >> 24); }
    }
    static __device__ __forceinline__ void put_match(u8* q, u32 D, u32 M) {
        if (D < 2048u && M >= 4u && M <= 11u) { q[0] = (u8)(1u | ((M - 4u) << 2) | ((D >> 8) << 5)); q[1] = (u8)(D & 0xFFu); }
        else { q[0] = (u8)((2u | ((M - 1u) << 2)) & 0xFFu); q[1] = (u8)(D & 0xFFu); q[2] = (u8)((D >> 8) & 0xFFu); }
    }
};
__device__ __forceinline__ u32 lzo_extn(u32 v) { return 1u + (v - 1u) / 255u; }                       // bytes of LZO.WriteExtendedInt(v), v >= 1
__device__ __forceinline__ u32 lzo_put_ext(u8* q, u32 v) { u32 k = 0; while (v > 255u) { q[k++] = 0; v -= 255u; } q[k++] = (u8)v; return k; }
__device__ __forceinline__ u32 lzo_lit_size(u32 L) { return L > 18u ? 1u + lzo_extn(L - 18u) : 1u; }  // the run's length token (L >= 4)
__device__ __forceinline__ u32 lzo_put_lit(u8* q, u32 L) { if (L > 18u) { q[0] = 0; return 1u + lzo_put_ext(q + 1, L - 18u); } q[0] = (u8)(L - 3u); return 1u; }
__device__ __forceinline__ u32 lzo_match_size(u32 D, u32 M) {
    if (M <= 8u && D <= 2048u) return 2u;
    if (D <= 16384u) return (M > 33u ? 1u + lzo_extn(M - 33u) : 1u) + 2u;
    return (M > 9u ? 1u + lzo_extn(M - 9u) : 1u) + 2u;
}
__device__ __forceinline__ u32 lzo_put_match(u8* q, u32 D, u32 M, u32 emb) {                          // emb: the 0-3 literals that follow, in the token's low bits
    if (M <= 8u && D <= 2048u) {
        const u32 flag = (emb | (((D - 1u) & 7u) << 2)) & 0xFFu;
        q[0] = (u8)(M <= 4u ? (flag | 0x40u | ((M - 3u) << 5)) : (flag | 0x80u | ((M - 5u) << 5)));
        q[1] = (u8)((D - 1u) >> 3);
        return 2u;
    }
    u32 k;
    if (D <= 16384u) {
        if (M > 33u) { q[0] = 0x20; k = 1u + lzo_put_ext(q + 1, M - 33u); } else { q[0] = (u8)(0x20u | (M - 2u)); k = 1u; }
        q[k] = (u8)((emb | ((D - 1u) << 2)) & 0xFFu); q[k + 1] = (u8)(((D - 1u) >> 6) & 0xFFu);
        return k + 2u;
    }
    const u32 d2 = D - 0x4000u, flag = (0x10u | ((d2 & 0x4000u) >> 11)) & 0xFFu;
    if (M > 9u) { q[0] = (u8)flag; k = 1u + lzo_put_ext(q + 1, M - 9u); } else { q[0] = (u8)(flag | (M - 2u)); k = 1u; }
    q[k] = (u8)((emb | (d2 << 2)) & 0xFFu); q[k + 1] = (u8)((d2 >> 6) & 0xFFu);
    return k + 2u;
}
// LZO over segments (alz_encode_seg_seq.h, round 6): a unit -- the literal run in front of a match (four and more: a run with its own length token; 0-3: bare, counted in the
// token of the match in front) and the match's token -- as a sequence of enc_seq_seg_kernel; `emb`: the count of the 0-3 literals BEHIND the match, in its token's low bits
template <> struct SeqFmt<ALZ_FMT_LZO> {
    static __device__ __forceinline__ u32 lit_hdr(u32 L) { return L >= 4u ? lzo_lit_size(L) : 0u; }
    static __device__ __forceinline__ u32 match_size(u32 D, u32 M) { return lzo_match_size(D, M); }
    static __device__ __forceinline__ void put_lit_hdr(u8* q, u32 L, u32, bool) { if (L >= 4u) (void)lzo_put_lit(q, L); }
    static __device__ __forceinline__ void put_match(u8* q, u32 D, u32 M, u32 emb = 0u) { (void)lzo_put_match(q, D, M, emb); }
};
#ifndef ALZ_SEQ_PARSE_CAP
#define ALZ_SEQ_PARSE_CAP 32   /* bytes the search inside WinParse compares per position for certain (what the look-ahead holds in registers) ... */
#endif
#ifndef ALZ_SEQ_PARSE_CAP_HI
#define ALZ_SEQ_PARSE_CAP_HI 128  /* ... and up to this many while fewer than ALZ_SEQ_PARSE_MANY lanes of the window are still equal */
#endif
#ifndef ALZ_SEQ_PARSE_MANY
#define ALZ_SEQ_PARSE_MANY 16     /* (Test.bmp / text as LZ4 blocks at quality 0, ms: a fixed cap of 32: 10.6 / 9.3, 48: 11.9 / 8.1, 128: 16.7 / 7.4; 32 .. 128 with 16 lanes: 11.4 / 7.7, with 8: 12.0 / 7.8, 32: 14.6 / 7.5) */
#endif
#ifndef ALZ_SEQ_LANE_LIT
#define ALZ_SEQ_LANE_LIT 4u      /* literal runs up to this long are copied by their own lane, longer ones by the wavefront (4: 15.8 ms, 16: 17.8) */
#endif
// The walk of enc_roles_kernel one window of 64 positions at a time, for kernels that write a window's tokens right behind its parse
// (enc_parse_seq_kernel, enc_emit_prs_kernel, enc_parse_lzo_kernel): no start mask in memory, and the exact matches the walk recomputes for capped
// positions stay in registers.  SEARCH (one candidate per position -- quality 0 --, one property set, 16-bit links): kernel B is in here
// too, as in enc_parse_emit_kernel<FMT, true>: the matches of window w + 1 are worked out while window w is parsed, the candidates'
// bytes of window w + 2 and the links and own bytes of window w + 3 are in flight -- no match array at all.  CAP bytes are compared per
// position; a longer match is measured exactly when the cursor stands on it.
template <bool SEARCH, int CAP>
struct WinParse {
    const EncGeom& g;
    const u8* data; int ns, limit, lane;                // ns: the bytes the finder is given; limit = ns - 4: the last searched position  :159
    const mentry* m; const int* p4; const int* pm;
    u8* hopmark;                                        // 64 bytes of LDS
    const unsigned short* lk16; u32 srange;
    int cur;                                            // cursor of FindNextBestMatch (absolute position)
    bool carry;                                         // the window in front found a token that starts at this window's first position
    uint2 a_n;                                          // the matches of the next window
    // ---- SEARCH: the stages of the look-ahead.  L: link + 32 own bytes of a window; C: the 32 bytes of its candidates.
    u32 lkA, lkB; u64 ownA[4], ownB[4], cndB[4];
    __device__ __forceinline__ WinParse(const EncGeom& g_, const u8* data_, int ns_, int lane_, const mentry* m_, const int* p4_, const int* pm_, u8* hopmark_)
        : g(g_), data(data_), ns(ns_), limit(ns_ - 4), lane(lane_), m(m_), p4(p4_), pm(pm_), hopmark(hopmark_),
          lk16(reinterpret_cast<const unsigned short*>(p4_)), srange((u32)(g_.max_dist - g_.min_dist)), cur(0), carry(false), lkA(0), lkB(0),
          ownA{0, 0, 0, 0}, ownB{0, 0, 0, 0}, cndB{0, 0, 0, 0} {
        if (SEARCH) {
            loadL((u32)lane, lkB, ownB); loadL(64u + (u32)lane, lkA, ownA);
            loadC((u32)lane, lkB, cndB);
            a_n = matchof((u32)lane, lkB, ownB, cndB);                      // window 0
            lkB = lkA; ownB[0] = ownA[0]; ownB[1] = ownA[1]; ownB[2] = ownA[2]; ownB[3] = ownA[3];
            loadC(64u + (u32)lane, lkB, cndB);                              // window 1's candidates
            loadL(128u + (u32)lane, lkA, ownA);                             // window 2's links and own bytes
        } else a_n = ldm((u32)lane);
    }
    __device__ __forceinline__ uint2 ldm(u32 q) { return (int)q <= limit ? m_unpack(__builtin_nontemporal_load(m + q)) : make_uint2(0, 0); }
    __device__ __forceinline__ u32 clampq(u32 q) { return (int)q <= limit ? q : (u32)(limit > 0 ? limit : 0); }             // (positions behind the last searched one: loads stay inside, results unused)
    __device__ __forceinline__ void loadL(u32 q, u32& lkv, u64 (&own)[4]) { const u32 qq = clampq(q); lkv = (int)q <= limit ? (u32)__builtin_nontemporal_load(lk16 + qq) : 0u; __builtin_memcpy(own, data + qq, 32); }
    __device__ __forceinline__ void loadC(u32 q, u32 lkv, u64 (&cnd)[4]) { const u32 qq = clampq(q); const bool ok = lkv - (u32)g.min_dist <= srange; __builtin_memcpy(cnd, data + qq - (ok ? lkv : 0u), 32); }   // (a candidate out of reach is not touched)
    __device__ __forceinline__ uint2 matchof(u32 q, u32 lkv, const u64 (&own)[4], const u64 (&cnd)[4]) {
        if ((int)q > limit) return make_uint2(0, 0);
        const bool ok = lkv - (u32)g.min_dist <= srange;                // a candidate (0: none), within maxDistance, not closer than minDistance  :259-266
        int best_possible = ns - (int)q; if (best_possible > g.max_len) best_possible = g.max_len;
        // CAP bytes are compared for certain, up to CAPHI while only a few lanes of the window are still equal: in a run or a repeated row every
        // lane is, each trip of the loop below is a memory round trip for the whole wavefront, and the cursor jumps over most of those positions
        // anyway (Test.bmp wants a small cap); in text a long match is one lane's, and every position that stops at the cap costs a search by
        // the whole wavefront when the cursor stands on it (text wants a large one).
        constexpr int CAPHI = CAP >= 64 ? CAP : ALZ_SEQ_PARSE_CAP_HI;
        int cmp_max = best_possible > CAPHI ? CAPHI : best_possible;
        const u64 x0 = own[0] ^ cnd[0], x1 = own[1] ^ cnd[1], x2 = own[2] ^ cnd[2], x3 = own[3] ^ cnd[3];
        int len = x0 ? (int)(__builtin_ctzll(x0) >> 3) : x1 ? 8 + (int)(__builtin_ctzll(x1) >> 3) : x2 ? 16 + (int)(__builtin_ctzll(x2) >> 3) : x3 ? 24 + (int)(__builtin_ctzll(x3) >> 3) : 32;
        bool go = ok && len == 32 && cmp_max > 32;
        if (__ballot(go)) {                                              // GetMatchLength behind the prefetched bytes
            const u8* pa = data + q; const u8* pb = data + q - (go ? lkv : 0u);
            int l = 32;
            const bool went = go;
            while (__ballot(go)) {
                // (before the trip: with CAP = 32 a window of a run takes none at all; l is wave-uniform among the lanes still equal, eight bytes per trip from 32 on)
                if (CAPHI > CAP && __popcll(__ballot(go && l >= CAP)) >= ALZ_SEQ_PARSE_MANY) { if (go && l >= CAP) { cmp_max = l; go = false; } continue; }
                const u64 z = load64(pa + (go ? l : 0)) ^ load64(pb + (go ? l : 0));
                if (go) { if (z) { l += (int)(__builtin_ctzll(z) >> 3); go = false; } else { l += 8; if (l >= cmp_max) go = false; } }
            }
            if (went) len = l;
        }
        if (len > cmp_max) len = cmp_max;
        const bool hitcap = ok && len == cmp_max && cmp_max < best_possible;
        int l2 = len;
        if (g.no_self_overlap && l2 > (int)lkv) l2 = (int)lkv;          // ScoreMatch  :301-321, one property set
        const bool take = ok && !hitcap && l2 >= g.min_len;
        return hitcap ? make_uint2(ALZ_CAPPED, ALZ_CAPPED) : make_uint2(take ? lkv : 0u, take ? (u32)l2 : 0u);
    }
    // The window at P (windows are taken in order, none left out): `a` = (distance, length) of the match at each position -- exact where a
    // token starts --, `sm` = the positions where the parse starts a match token.
    __device__ __forceinline__ void window(u32 P, uint2& a, u64& sm) {
        const u32 p = P + (u32)lane;
        a = a_n;
        // (a window the cursor has already jumped over -- LZ4 matches have no longest length -- is never looked at: its stage is left out.  The
        // cursor only moves forward, so a window that IS parsed had all three of its stages)
        if (SEARCH) {
            if (cur < (int)P + 128) a_n = matchof(p + 64u, lkB, ownB, cndB);     // window w + 1 (bytes that arrived during the window before)
            lkB = lkA; ownB[0] = ownA[0]; ownB[1] = ownA[1]; ownB[2] = ownA[2]; ownB[3] = ownA[3];
            if (cur < (int)P + 192) loadC(p + 128u, lkB, cndB);                  // window w + 2's candidates
            if (cur < (int)P + 256) loadL(p + 192u, lkA, ownA);                  // window w + 3's links and own bytes
        } else if (cur < (int)P + 128) a_n = ldm(p + 64u);
        sm = carry ? 1ull : 0ull;
        carry = false;
        if (cur < (int)P + 64 && cur <= limit) {
            // ---- the parse of this window (the body of enc_roles_kernel)
            uint2 b;
            b.x = (u32)__builtin_amdgcn_ds_bpermute((lane + 1) << 2, (int)a.x); b.y = (u32)__builtin_amdgcn_ds_bpermute((lane + 1) << 2, (int)a.y);
            {   // lane 63's neighbour is the first position of the next window
                const u32 n0x = (u32)__builtin_amdgcn_readlane((int)a_n.x, 0), n0y = (u32)__builtin_amdgcn_readlane((int)a_n.y, 0);
                if (lane == 63) b = make_uint2(n0x, n0y);
            }
            const bool capped = a.y == ALZ_CAPPED || b.y == ALZ_CAPPED;
            int jump = 1, startrel = 0;   // startrel: 0 no token here, 1 match starts here, 2 literal here + match at p + 1
            if (!capped && (int)p <= limit && (int)a.y >= g.min_len) {
                const int l0 = (int)a.y, l1 = (int)b.y, pi = (int)p;
                const bool lazyc = l0 <= g.lazy && pi + 1 <= limit;
                if (lazyc && l1 > l0) { startrel = 2; const int e = pi + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; jump = (pi + 2 > stop ? pi + 2 : stop) - pi; }
                else { startrel = 1; const int skip = lazyc ? 1 : 0; const int e = pi + l0; const int stop = e < limit + 1 ? e : limit + 1; jump = (pi + 1 + skip > stop ? pi + 1 + skip : stop) - pi; }
            }
            int rel = cur - (int)P;
            if (__ballot(capped) == 0) {
                u64 M = 0; u32 r = (u32)rel, j;
                const u32 lim = (u32)(limit + 1 - (int)P) < 64u ? (u32)(limit + 1 - (int)P) : 64u;     // (r < lim on entry: cur <= limit)
                if (lim == 64u) {
                    // (two tokens per hop, the cursor 64 below zero: enc_parse_emit_kernel)
                    const u32 tgt = (u32)lane + (u32)jump;                     // where my jump lands (>= 64: outside)
                    const int jn = __builtin_amdgcn_ds_bpermute((int)((tgt < 64u ? tgt : (u32)lane) << 2), jump);
                    const int jump2 = tgt < 64u ? jump + jn : jump;
                    r -= 64u;
                    asm volatile(
                        "s_nop 3\n"
                        "1:\n\t"
                        "s_bitset1_b64 %[M], %[r]\n\t"
                        "v_readlane_b32 %[j], %[jump], %[r]\n\t"
                        "s_add_u32 %[r], %[r], %[j]\n\t"
                        "s_cbranch_scc0 1b\n\t"
                        : [M] "+s"(M), [r] "+s"(r), [j] "=&s"(j)
                        : [jump] "v"(jump2)
                        : "scc");
                    r += 64u;
                    if (((M >> lane) & 1ull) && tgt < 64u) hopmark[tgt] = 1;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const u32 hm = hopmark[lane];
                    if (hm) hopmark[lane] = 0;
                    M |= __ballot(hm != 0u);
                } else
                asm volatile(
                    "s_nop 3\n"
                    "1:\n\t"
                    "s_bitset1_b64 %[M], %[r]\n\t"
                    "v_readlane_b32 %[j], %[jump], %[r]\n\t"
                    "s_add_u32 %[r], %[r], %[j]\n\t"
                    "s_cmp_lt_u32 %[r], %[lim]\n\t"
                    "s_cbranch_scc1 1b\n\t"
                    : [M] "+s"(M), [r] "+s"(r), [j] "=&s"(j)
                    : [jump] "v"(jump), [lim] "s"(lim)
                    : "scc");
                const u64 s1 = __ballot(startrel == 1) & M, s2 = __ballot(startrel == 2) & M;
                sm |= s1 | (s2 << 1);
                if (s2 >> 63) carry = true;                                    // start in lane 0 of the next window
                rel = (int)r;
            }
            else while (rel < 64 && (int)P + rel <= limit) {
                int j, sr;
                if (__builtin_amdgcn_readlane((int)capped, rel)) {
                    // a capped candidate here: redo MatchSearch exactly for this cursor and its lazy neighbour, and put both into the
                    // registers the sequences take their matches from (the neighbour may be the next window's first position)
                    const int q = (int)P + rel;
                    int d0, l0, d1 = 0, l1 = 0; bool s0, s1;
                    const uint2 e0 = make_uint2((u32)__builtin_amdgcn_readlane((int)a.x, rel), (u32)__builtin_amdgcn_readlane((int)a.y, rel));
                    const uint2 e1 = rel + 1 < 64 ? make_uint2((u32)__builtin_amdgcn_readlane((int)a.x, (rel + 1) & 63), (u32)__builtin_amdgcn_readlane((int)a.y, (rel + 1) & 63))
                                                  : make_uint2((u32)__builtin_amdgcn_readlane((int)a_n.x, 0), (u32)__builtin_amdgcn_readlane((int)a_n.y, 0));
                    if (g.use_min_table) benc_capped_cursor<true>(data, ns, g, p4, pm, q, limit, e0, e0.y == ALZ_CAPPED, e1, e1.y == ALZ_CAPPED, d0, l0, d1, l1, s0, s1);
                    else benc_capped_cursor<false>(data, ns, g, p4, pm, q, limit, e0, e0.y == ALZ_CAPPED, e1, e1.y == ALZ_CAPPED, d0, l0, d1, l1, s0, s1);
                    if (s0 && lane == rel) a = make_uint2((u32)d0, (u32)l0);
                    if (s1) {
                        if (rel + 1 < 64) { if (lane == rel + 1) a = make_uint2((u32)d1, (u32)l1); }
                        else if (lane == 0) a_n = make_uint2((u32)d1, (u32)l1);
                    }
                    j = 1; sr = 0;
                    if (l0 >= g.min_len) {
                        const bool lazyc = l0 <= g.lazy && q + 1 <= limit;
                        if (lazyc && l1 > l0) { sr = 2; const int e = q + 1 + l1; const int stop = e < limit + 1 ? e : limit + 1; j = (q + 2 > stop ? q + 2 : stop) - q; }
                        else { sr = 1; const int skip = lazyc ? 1 : 0; const int e = q + l0; const int stop = e < limit + 1 ? e : limit + 1; j = (q + 1 + skip > stop ? q + 1 + skip : stop) - q; }
                    }
                } else {
                    j = __builtin_amdgcn_readlane(jump, rel);
                    sr = __builtin_amdgcn_readlane(startrel, rel);
                }
                if (sr == 1) sm |= 1ull << rel;
                else if (sr == 2) { if (rel + 1 < 64) sm |= 1ull << (rel + 1); else carry = true; }   // start in lane 0 of the next window
                rel += j;
            }
            cur = (int)P + rel;
        }
    }
};

// LZ4 blocks and raw Snappy in ONE kernel behind kernel A (round 4): the walk over a window of 64 positions (WinParse), then the sequences
// that start in it, from the same registers; at quality 0 the search as well.
template <int FMT, bool SEARCH>
__global__ __launch_bounds__(64) void enc_parse_seq_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base,
                                                           const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                           u32 count, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                           const int* __restrict__ prev4, const int* __restrict__ prevm,
                                                           alz_result* __restrict__ results, alz_encode_aux* __restrict__ aux, EncGeom g) {
    typedef SeqFmt<FMT> F;
    constexpr bool LZ4 = FMT == ALZ_FMT_LZ4_BLOCK;
    constexpr int SEQ_CAP = LZ4 ? ALZ_SEQ_PARSE_CAP : 64;                    // (Snappy's longest copy: never capped)
    __shared__ u8 hopmark[64];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)threadIdx.x;
    hopmark[lane] = 0;
    const u32 sid = index_list[bid];
    if (sid == 0xFFFFFFFFu) return;               // (a list written on the device, enc_scan_select_kernel: the stream goes the other way)
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u8* data = src;
    const u32 n = st.src_len;
    u8* dst = dst_base + st.dst_off;
    const u32 cap = st.dst_cap;
    if (lane == 0 && aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
    if (LZ4 && n < 5u) {                                                      // source.Slice(0, Length - 5) throws
        if (lane == 0) { alz_result r; r.dst_len = 0; r.src_used = n; r.status = ALZ_ST_BAD_TOKEN; r.reserved = 0; results[sid] = r; }
        return;
    }
    const int ns = (int)n - (LZ4 ? 5 : 0);                                    // what the finder is given: LZ4 searches source[0 : n-5]  (LZ4.cs:208)
    const mentry* m = match + pos_off[sid];
    const int* p4 = prev4 + pos_off[sid];
    const int* pm = g.use_min_table ? prevm + pos_off[sid] : nullptr;
    u32 cover = 0;          // end of the last match = first literal not yet written
    u32 obase = 0;          // bytes written before the window
    bool fail = false;
    if (!LZ4) {                                                               // Snappy: the decompressed length as a varint  :126-135
        const u32 k = n < 0x80u ? 1u : n < 0x4000u ? 2u : n < 0x200000u ? 3u : n < 0x10000000u ? 4u : 5u;
        if (k <= cap) { if (lane == 0) { u32 v = n, q = 0; while (v >= 0x80u) { dst[q++] = (u8)((v | 0x80u) & 0xFFu); v >>= 7; } dst[q] = (u8)v; } } else fail = true;
        obase = k;
    }
    WinParse<SEARCH, SEQ_CAP> ps(g, data, ns, lane, m, p4, pm, hopmark);
    for (u32 P = 0; P < n; P += 64) {
        const u32 p = P + (u32)lane;
        uint2 a; u64 sm;
        ps.window(P, a, sm);
        if (sm == 0ull) continue;                                             // (no match starts here: the literals wait for the next one)
        // ---- the sequences that start in this window
        const bool start = ((sm >> lane) & 1ull) != 0ull;
        const u32 M = start ? a.y : 0u, D = a.x;
        const u32 mend = start ? p + M : 0u;
        const u32 pmax = scan_max(mend);                                       // inclusive
        u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)pmax, 0x138, 0xF, 0xF, false);   // wave_shr:1 -> max over lanes below
        if (before < cover) before = cover;
        const u32 L = start ? p - before : 0u;
        const u32 lh = start ? F::lit_hdr(L) : 0u;
        const u32 esz = start ? lh + L + F::match_size(D, M) : 0u;
        const u32 incl = scan_add(esz);
        const u32 off = obase + incl - esz;
        const bool fits = start && off + esz <= cap;
        if (start && !fits) fail = true;
        if (fits) {
            F::put_lit_hdr(dst + off, L, M, false);
            F::put_match(dst + off + lh + L, D, M);
        }
        // The literals: every literal POSITION of this window whose sequence starts in this window stores its own byte -- one pass, whatever
        // the number of runs (as a loop over the runs, each copied by the wavefront, this was the longest kernel of the LZ4 batch at quality 0:
        // 18.3 of 59 ms -- a dozen runs of a dozen bytes per window).  A lane finds the next start at or behind it with the start mask and
        // takes that lane's numbers; only the FIRST start of a window can own literals of earlier windows: those the wavefront copies.
        {
            const u64 above = (lane < 63 ? sm >> (lane + 1) : 0ull);
            const int s = above ? lane + 1 + (int)__builtin_ctzll(above) : lane;          // the next start behind me (my own lane: none)
            const u32 sbef = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)before);
            const u32 sbase = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)(off + lh - before));
            const u32 sfit = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)(fits ? 1u : 0u));
            if (above && !start && sfit && p >= sbef && p < n) dst[sbase + p] = src[p];
            const int f0 = (int)__builtin_ctzll(sm);                                       // the first start of the window
            const u32 fbef = (u32)__builtin_amdgcn_readlane((int)before, f0);
            if (fbef < P && __builtin_amdgcn_readlane((int)(fits ? 1u : 0u), f0)) {
                const u32 dq = (u32)__builtin_amdgcn_readlane((int)(off + lh), f0);
                wave_copy(dst + dq, src + fbef, P - fbef, lane);
            }
        }
        obase += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u32 wmax = (u32)__builtin_amdgcn_readlane((int)pmax, 63);
        if (wmax > cover) cover = wmax;
    }
    // the end: the remaining literals (LZ4: at least five, always a sequence; Snappy: an element only if there are any)
    const u32 plain = n - cover, lh = (LZ4 || plain) ? F::lit_hdr(plain) : 0u;
    const u32 total = obase + lh + plain;
    if (total > cap) fail = true;
    const bool anyfail = __ballot(fail) != 0ull;
    if (!anyfail && (LZ4 || plain)) {
        if (lane == 0) F::put_lit_hdr(dst + obase, plain, 4u, true);
        wave_copy(dst + obase + lh, src + cover, plain, lane);
    }
    if (lane == 0) {
        alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
        results[sid] = r;
    }
}

// LZ4 blocks and raw Snappy for the streams enc_scan_select_kernel picked (round 6): enc_scan_emit_kernel's walk -- every position the cursor stands on searched exactly by the
// whole wavefront, no links, no match array --, the matches into a list, one per lane, and 64 sequences emitted at once with enc_parse_seq_kernel's arithmetic: a sequence is the
// literals since the end of the match before it, then the match; sizes by prefix sums.  Literal runs are copied by the wavefront, one after the other (few and short where this
// path is taken).
template <int FMT>
__global__ __launch_bounds__(64) void enc_scan_seq_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                          const u32* __restrict__ index_list, u32 count, const int* __restrict__ prev4, const u64* __restrict__ pos_off,
                                                          alz_result* __restrict__ results, alz_encode_aux* __restrict__ aux, EncGeom g) {
    typedef SeqFmt<FMT> F;
    constexpr bool LZ4 = FMT == ALZ_FMT_LZ4_BLOCK;
    __shared__ unsigned short candl[64];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const u32 sid = index_list[bid];
    if (sid == 0xFFFFFFFFu) return;               // (the stream went the other way)
    const int lane = (int)threadIdx.x;
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u32 n = st.src_len;
    u8* dst = dst_base + st.dst_off;
    const u32 cap = st.dst_cap;
    if (lane == 0 && aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
    if (LZ4 && n < 5u) {                                                      // source.Slice(0, Length - 5) throws
        if (lane == 0) { alz_result r; r.dst_len = 0; r.src_used = n; r.status = ALZ_ST_BAD_TOKEN; r.reserved = 0; results[sid] = r; }
        return;
    }
    const int ns = (int)n - (LZ4 ? 5 : 0);                                    // what the finder is given: LZ4 searches source[0 : n-5]  (LZ4.cs:208)
    const int limit = ns - 4;
    const int* p4 = prev4 + pos_off[sid];                                     // kernel A's links: what lies behind the nearest blocks is reached through them
    u32 cover = 0, obase = 0;
    bool fail = false;
    if (!LZ4) {                                                               // Snappy: the decompressed length as a varint  :126-135
        const u32 k = n < 0x80u ? 1u : n < 0x4000u ? 2u : n < 0x200000u ? 3u : n < 0x10000000u ? 4u : 5u;
        if (k <= cap) { if (lane == 0) { u32 v = n, q = 0; while (v >= 0x80u) { dst[q++] = (u8)((v | 0x80u) & 0xFFu); v >>= 7; } dst[q] = (u8)v; } } else fail = true;
        obase = k;
    }
    int cur = 0;
    bool walk = limit >= 0;
    while (walk) {
        // ---- up to 64 matches: lane k holds match k (position, distance, length)
        u32 tp = 0, D = 0, M = 0;
        int k = 0;
        while (k < 64) {
            if (cur > limit) { walk = false; break; }
            int d0, l0, d1 = 0, l1 = 0;
            (void)benc_wave_scan_search(src, ns, g, cur, d0, l0, candl, p4);
            if (l0 < g.min_len) { cur++; continue; }                            // :166-170
            const bool lazyc = l0 <= g.lazy && cur + 1 <= limit;
            if (lazyc) (void)benc_wave_scan_search(src, ns, g, cur + 1, d1, l1, candl, p4);
            int mp = cur, md = d0, ml = l0, skip = lazyc ? 1 : 0;
            if (lazyc && l1 > l0) { mp = cur + 1; md = d1; ml = l1; skip = 0; }   // :181-186
            if (lane == k) { tp = (u32)mp; D = (u32)md; M = (u32)ml; }
            k++;
            const int e = mp + ml, stop = e < limit + 1 ? e : limit + 1;         // :195-203
            cur = mp + 1 + skip > stop ? mp + 1 + skip : stop;
        }
        if (k == 0) break;
        // ---- the sequences of this batch
        const bool start = lane < k;
        const u32 mend = start ? tp + M : 0u;
        u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)mend, 0x138, 0xF, 0xF, false);     // wave_shr:1 -> the end of the match before mine
        if (lane == 0) before = cover;
        const u32 L = start ? tp - before : 0u;
        const u32 lh = start ? F::lit_hdr(L) : 0u;
        const u32 esz = start ? lh + L + F::match_size(D, M) : 0u;
        const u32 incl = scan_add(esz);
        const u32 off = obase + incl - esz;
        const bool fits = start && off + esz <= cap;
        if (start && !fits) fail = true;
        if (fits) {
            F::put_lit_hdr(dst + off, L, M, false);
            F::put_match(dst + off + lh + L, D, M);
        }
        u64 runs = __ballot(fits && L != 0u);
        while (runs) {
            const int j = (int)__builtin_ctzll(runs);
            runs &= runs - 1ull;
            wave_copy(dst + (u32)__builtin_amdgcn_readlane((int)(off + lh), j), src + (u32)__builtin_amdgcn_readlane((int)before, j), (u32)__builtin_amdgcn_readlane((int)L, j), lane);
        }
        obase += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        cover = (u32)__builtin_amdgcn_readlane((int)mend, k - 1);
    }
    // the end: the remaining literals (LZ4: at least five, always a sequence; Snappy: an element only if there are any)
    const u32 plain = n - cover, lh = (LZ4 || plain) ? F::lit_hdr(plain) : 0u;
    const u32 total = obase + lh + plain;
    if (total > cap) fail = true;
    const bool anyfail = __ballot(fail) != 0ull;
    if (!anyfail && (LZ4 || plain)) {
        if (lane == 0) F::put_lit_hdr(dst + obase, plain, 4u, true);
        wave_copy(dst + obase + lh, src + cover, plain, lane);
    }
    if (lane == 0) {
        alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
        results[sid] = r;
    }
}

#include "alz_encode_seg_seq.h"

// PRS (PRS.cs:104-159) from the start mask: tokens of one, two or four flag bits.  With B = the flag bits written before a token's
// payload is handed to the flag writer (a literal's before its bit, a short match's behind its four bits, a long match's between its two)
// the payload lands behind floor(B / 8) + 1 flag bytes -- the byte its bits belong to is in place before it -- except for the offset byte
// of a short match whose four bits just completed a flag byte: the writer is told to let it out at once (flush_if_necessary), in front
// of the next flag byte.  The flag byte k stands in front of the first payload with B >= 8 k (behind it, if that is such an offset byte).
// So two prefix sums (bits, payload bytes) place everything; flag bytes collect their bits in LDS and are stored by the token that owns
// their last bit, as in enc_parse_emit_kernel.  A match of length 2 further than 0x100 back is not written as a match (its bytes go out as
// literals; the parse has moved on behind it either way).
// The walk is in here (WinParse); SEARCH: the one-candidate search of quality 0 too.
template <bool BIG, bool SEARCH>
__global__ __launch_bounds__(64) void enc_emit_prs_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base,
                                                          const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                          u32 count, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                          alz_result* __restrict__ results,
                                                          alz_encode_aux* __restrict__ aux, const int* __restrict__ prev4,
                                                          const int* __restrict__ prevm, EncGeom g) {
    __shared__ u32 flagacc[64];
    __shared__ u32 gofs[64];
    __shared__ u8 hopmark[64];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u32 n = st.src_len;
    u8* dst = dst_base + st.dst_off;
    const u32 cap = st.dst_cap;
    const mentry* m = match + pos_off[sid];
    flagacc[lane] = 0; gofs[lane] = 0; hopmark[lane] = 0;
    __syncthreads();
    u32 bit_base = 0;       // flag bits before the window
    u32 pay_base = 0;       // payload bytes before the window
    u32 cover = 0;          // end of the last match written as a match
    u32 lastk = 0xFFFFFFFFu;    // flag byte of the last payload before the window (none yet)
    bool fail = false;
    u32 sb_n = (u32)lane < n ? src[lane] : 0u;
    WinParse<SEARCH, ALZ_SEQ_PARSE_CAP> ps(g, src, (int)n, lane, m, prev4 + pos_off[sid], g.use_min_table ? prevm + pos_off[sid] : nullptr, hopmark);
    // one more trip behind the data for the end token (bit 0, two zero bytes, bit 1) on lane 0
    for (u32 P = 0; P < n + 64u; P += 64) {
        const bool tail = P >= n;
        if (tail && P > ((n + 63u) & ~63u)) break;                             // (exactly one trip behind the last window)
        const u32 p = P + (u32)lane;
        u64 sm = 0ull;
        const u32 sb = sb_n;
        uint2 mt_all = make_uint2(0, 0);
        if (!tail) ps.window(P, mt_all, sm);
        if (!tail && p + 64 < n) sb_n = src[p + 64];
        bool start = !tail && ((sm >> lane) & 1ull) && p < n;
        uint2 mt = make_uint2(0, 0);
        if (start) mt = mt_all;
        if (start && mt.y == 2u && mt.x > 0x100u) start = false;               // PRS.cs: not worth a long match -- literals
        const u32 mend = start ? p + mt.y : 0u;
        const u32 pmax = scan_max(mend);                                       // inclusive
        u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)pmax, 0x138, 0xF, 0xF, false);   // wave_shr:1 -> max over lanes below
        if (before < cover) before = cover;
        const bool lit = !tail && !start && p < n && p >= before;
        const bool endtok = tail && lane == 0;
        const bool shortm = start && mt.x <= 0x100u && mt.y <= 5u;
        const bool longm = (start && !shortm) || endtok;
        const bool tok = lit || start || endtok;
        const u32 nbits = lit ? 1u : shortm ? 4u : longm ? 2u : 0u;
        const u32 psize = lit ? 1u : shortm ? 1u : endtok ? 2u : longm ? (mt.y > 9u ? 3u : 2u) : 0u;
        const u32 bincl = scan_add(nbits), pincl = scan_add(psize);
        const u32 B0 = bit_base + bincl - nbits;                               // my first bit
        const u32 pidx = pay_base + pincl - psize;                             // my first payload byte among all payload bytes
        const u32 Bp = B0 + (lit ? 0u : shortm ? 4u : 1u);                     // bits written when my payload is handed over
        const bool special = shortm && (Bp & 7u) == 0u;
        const u32 kp = Bp >> 3;                                                // the flag byte my payload waits for (special: the one it follows)
        const u32 out = kp + 1u - (special ? 1u : 0u) + pidx;                  // where my payload goes
        // the flag byte kp stands in front of the first payload with B >= 8 kp: that is me if the payload before mine had a smaller one
        // (kp never falls and rises by at most one from token to token: two payloads are at most five bits apart)
        const u32 kinc = scan_max(tok ? kp + 1u : 0u);                          // the latest kp + 1 up to and including my lane
        const u32 kexc = (u32)__builtin_amdgcn_update_dpp(0, (int)kinc, 0x138, 0xF, 0xF, false);
        const u32 prevk = kexc ? kexc - 1u : lastk;
        const bool opener = tok && (prevk == 0xFFFFFFFFu || prevk < kp);
        if (opener) { gofs[kp & 63u] = kp + pidx + (special ? 1u : 0u); }
        __syncthreads();
        // my bits into their flag bytes (a token may straddle two)
        if (tok) {
            // bit values in order: literal 1; short 0,0,b1,b0; long 0 .. 1
            const u32 l2 = mt.y - 2u;
            const u32 pattern = lit ? 1u : shortm ? ((((l2 >> 1) & 1u) << 2) | ((l2 & 1u) << 3)) : 2u;   // bit i of `pattern` = my i-th flag bit
#pragma unroll
            for (u32 i = 0; i < 4u; i++) {
                if (i < nbits && ((pattern >> i) & 1u)) {
                    const u32 b = B0 + i;
                    atomicOr(&flagacc[(b >> 3) & 63u], 1u << (BIG ? 7u - (b & 7u) : (b & 7u)));
                }
            }
        }
        __syncthreads();
        if (tok) {
            // the flag bytes whose last bit is mine are complete: store them
#pragma unroll
            for (u32 i = 0; i < 4u; i++) {
                const u32 b = B0 + i;
                if (i < nbits && (b & 7u) == 7u) {
                    const u32 k = b >> 3, fo = gofs[k & 63u];
                    if (fo < cap) dst[fo] = (u8)flagacc[k & 63u]; else fail = true;
                    flagacc[k & 63u] = 0;
                }
            }
            // payload
            if (out + psize <= cap) {
                if (lit) dst[out] = (u8)sb;
                else if (shortm) dst[out] = (u8)((0u - mt.x) & 0xFFu);
                else if (endtok) { dst[out] = 0; dst[out + 1] = 0; }
                else {
                    u32 v = ((0u - mt.x) << 3) & 0xFFFFu;
                    if (mt.y <= 9u) v |= mt.y - 2u;
                    if (BIG) { dst[out] = (u8)(v >> 8); dst[out + 1] = (u8)(v & 0xFFu); } else { dst[out] = (u8)(v & 0xFFu); dst[out + 1] = (u8)(v >> 8); }
                    if (mt.y > 9u) dst[out + 2] = (u8)(mt.y - 1u);
                }
            } else fail = true;
        }
        __syncthreads();
        bit_base += (u32)__builtin_amdgcn_readlane((int)bincl, 63);
        pay_base += (u32)__builtin_amdgcn_readlane((int)pincl, 63);
        {   const u32 last = (u32)__builtin_amdgcn_readlane((int)kinc, 63); if (last) lastk = last - 1u; }
        const u32 wmax = (u32)__builtin_amdgcn_readlane((int)pmax, 63);
        if (wmax > cover) cover = wmax;
    }
    // Dispose(): a partial flag byte goes out with its unused bits zero
    const u32 nflags = (bit_base + 7u) >> 3;
    if ((bit_base & 7u) != 0u && lane == 0) { const u32 k = bit_base >> 3, fo = gofs[k & 63u]; if (fo < cap) dst[fo] = (u8)flagacc[k & 63u]; else fail = true; }
    const u32 total = nflags + pay_base;
    if (total > cap) fail = true;
    const bool anyfail = __ballot(fail) != 0ull;
    if (lane == 0) {
        alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
        results[sid] = r;
        if (aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
    }
}

// LZO1X (LZO.cs:141-250), the walk (WinParse) and the writer in one kernel.  The writer is sequential only at the head of a stream: a match
// that 1-3 literals precede is cut at its front so that four go out, and a match cut below three bytes is not written -- which can leave
// 1-3 literals in front of the next one.  Once a match HAS been written the state is clean for good: what follows it is 0-3 literals,
// which ride in its token and go out right behind it, or four and more, a literal run of its own.  So lane 0 walks the head the
// reference's way until the first match is out, and from there every match start is one unit -- the 0-3 literals in front of it, or a
// literal run of >= 4, then its token in one of three forms -- whose size follows from its own numbers: a prefix sum places the units, the
// wavefront copies the long runs.  The count of the 0-3 literals behind a match sits in ITS token: that byte is written by the next unit.
// (lzo_extn .. lzo_put_match: in front of SeqFmt<ALZ_FMT_LZO>, above)
#ifndef ALZ_LZO_LANE_LIT
#define ALZ_LZO_LANE_LIT 4u
#endif
// the same without the byte that holds the count of the literals behind the match: `epos` = where it goes, `ebase` = its other bits
__device__ __forceinline__ u32 lzo_put_match_def(u8* q, u32 D, u32 M, u32& epos, u32& ebase) {
    if (M <= 8u && D <= 2048u) {
        const u32 flag = (((D - 1u) & 7u) << 2) & 0xFFu;
        ebase = M <= 4u ? (flag | 0x40u | ((M - 3u) << 5)) : (flag | 0x80u | ((M - 5u) << 5)); epos = 0;
        q[1] = (u8)((D - 1u) >> 3);
        return 2u;
    }
    u32 k;
    if (D <= 16384u) {
        if (M > 33u) { q[0] = 0x20; k = 1u + lzo_put_ext(q + 1, M - 33u); } else { q[0] = (u8)(0x20u | (M - 2u)); k = 1u; }
        ebase = ((D - 1u) << 2) & 0xFFu; epos = k; q[k + 1] = (u8)(((D - 1u) >> 6) & 0xFFu);
        return k + 2u;
    }
    const u32 d2 = D - 0x4000u, flag = (0x10u | ((d2 & 0x4000u) >> 11)) & 0xFFu;
    if (M > 9u) { q[0] = (u8)flag; k = 1u + lzo_put_ext(q + 1, M - 9u); } else { q[0] = (u8)(flag | (M - 2u)); k = 1u; }
    ebase = (d2 << 2) & 0xFFu; epos = k; q[k + 1] = (u8)((d2 >> 6) & 0xFFu);
    return k + 2u;
}
template <bool SEARCH>
__global__ __launch_bounds__(64) void enc_parse_lzo_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base,
                                                          const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                          u32 count, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                          alz_result* __restrict__ results, alz_encode_aux* __restrict__ aux,
                                                          const int* __restrict__ prev4, const int* __restrict__ prevm, EncGeom g) {
    __shared__ u8 hopmark[64];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)threadIdx.x;
    hopmark[lane] = 0;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u32 n = st.src_len;
    u8* dst = dst_base + st.dst_off;
    const u32 cap = st.dst_cap;
    const mentry* m = match + pos_off[sid];
    auto finish = [&](u32 total, bool fail, int status) {
        if (lane == 0) {
            alz_result r; r.dst_len = (fail || status != ALZ_ST_OK) ? 0u : total; r.src_used = n;
            r.status = status != ALZ_ST_OK ? status : (fail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK); r.reserved = 0;
            results[sid] = r;
            if (aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
        }
    };
    // ---- the head, the reference's way, on lane 0 (everything here is wave-uniform: the other lanes follow along and do not store)
    u32 sp = 0, olen = 0; bool fail = false; int status = ALZ_ST_OK;
    auto put = [&](u32 b) { if (olen < cap) { if (lane == 0) dst[olen] = (u8)b; } else fail = true; olen++; };
    auto copy = [&](u32 from, u32 len) { for (u32 i = 0; i < len; i++) put(src[from + i]); };
    if (n < 0x10u) {
        put(17u + n); copy(0, n); put(0x11); put(0); put(0);
        finish(olen, fail, status);
        return;
    }
    // The walk, window by window (WinParse): `wa` / `wsm` = the matches and the start bits of the window at wP.  The head asks for one start
    // after the other (the next one always behind the last), which moves the window forward; the units behind it take the windows in order.
    WinParse<SEARCH, ALZ_SEQ_PARSE_CAP> ps(g, src, (int)n, lane, m, prev4 + pos_off[sid], g.use_min_table ? prevm + pos_off[sid] : nullptr, hopmark);
    u32 wP = 0; uint2 wa; u64 wsm;
    ps.window(0, wa, wsm);
    // the next start at or behind `from` (n: none), and its match
    auto next_start = [&](u32 from, u32& d, u32& l) -> u32 {
        for (;;) {
            u64 w = wsm;
            if (from > wP) w = from - wP < 64u ? w & (~0ull << (from - wP)) : 0ull;
            if (w) {
                const int b = (int)__builtin_ctzll(w);
                d = (u32)__builtin_amdgcn_readlane((int)wa.x, b); l = (u32)__builtin_amdgcn_readlane((int)wa.y, b);
                return wP + (u32)b;
            }
            if (wP + 64u >= n) { wsm = 0ull; d = 0; l = 0; return n; }
            wP += 64u;
            ps.window(wP, wa, wsm);
        }
    };
    u32 ml = 0, md = 0;                                                       // mt: offset, length, distance
    u32 mo = next_start(0, md, ml);
    u32 mbit = mo;                                                            // mt's bit in the mask (mo itself may be moved below)
    bool clean = false;
    while (sp != n && !clean) {
        u32 plain = mo - sp;
        if (plain != 0u) {
            if (plain < 4u) { const u32 dif = 4u - plain; mo += dif; ml = ml > dif ? ml - dif : 0u; plain = 4u; }
            if (plain > 18u) { put(0); u32 v = plain - 18u; while (v > 255u) { put(0); v -= 255u; } put(v); } else put(plain - 3u);
            if (sp + plain > n) { status = ALZ_ST_BAD_TOKEN; break; }
            copy(sp, plain); sp += plain;
        }
        // the finder's next match: the next bit of the mask
        u32 nl = 0, nd = 0;
        const u32 no = mbit < n ? next_start(mbit + 1u, nd, nl) : n;
        if (ml >= 3u) {
            sp += ml;
            u32 emb = no - sp;
            if (no < sp) { status = ALZ_ST_BAD_TOKEN; break; }
            if (emb > 3u) emb = 0;
            {   // (tokens are short except for their extension bytes: written through put() byte by byte)
                if (ml <= 8u && md <= 2048u) {
                    const u32 flag = (emb | (((md - 1u) & 7u) << 2)) & 0xFFu;
                    put(ml <= 4u ? (flag | 0x40u | ((ml - 3u) << 5)) : (flag | 0x80u | ((ml - 5u) << 5))); put((md - 1u) >> 3);
                } else if (md <= 16384u) {
                    if (ml > 33u) { put(0x20); u32 v = ml - 33u; while (v > 255u) { put(0); v -= 255u; } put(v); } else put(0x20u | (ml - 2u));
                    put((emb | ((md - 1u) << 2)) & 0xFFu); put(((md - 1u) >> 6) & 0xFFu);
                } else {
                    const u32 d2 = md - 0x4000u, flag = (0x10u | ((d2 & 0x4000u) >> 11)) & 0xFFu;
                    if (ml > 9u) { put(flag); u32 v = ml - 9u; while (v > 255u) { put(0); v -= 255u; } put(v); } else put(flag | (ml - 2u));
                    put((emb | (d2 << 2)) & 0xFFu); put((d2 >> 6) & 0xFFu);
                }
            }
            if (sp + emb > n) { status = ALZ_ST_BAD_TOKEN; break; }
            copy(sp, emb); sp += emb;
            clean = true;                                                      // from here on: 0 or >= 4 literals in front of every match
        }
        mo = no; mbit = no; ml = nl; md = nd;
    }
    if (status != ALZ_ST_OK || sp == n) {
        if (status == ALZ_ST_OK) { put(0x11); put(0); put(0); }
        finish(olen, fail, status);
        return;
    }
    // ---- the rest: one unit per match start at or behind sp
    const u32 sp0 = sp;
    u32 cover = sp0, obase = olen;
    // A token carries the count of the 0-3 literals BEHIND it (its low two bits): that is known when the next start is, so the byte that holds
    // them is written by the NEXT unit (or at the end of the data) -- `pend`: where it goes and its other bits -- and those literals are the
    // first bytes of that unit.
    u32 pend_addr = 0xFFFFFFFFu, pend_val = 0;
    bool held = true;                                                          // the window the head stopped in is in wa / wsm
    for (u32 P = wP; P < n; P += 64) {
        const u32 p = P + (u32)lane;
        uint2 mt_all; u64 sm;
        if (held) { mt_all = wa; sm = wsm; held = false; } else ps.window(P, mt_all, sm);
        const bool start = ((sm >> lane) & 1ull) && p >= mo && p < n;          // (mo: the first match not yet written)
        const u64 stm = __ballot(start);
        if (stm == 0ull) continue;
        uint2 mt = make_uint2(0, 0);
        if (start) mt = mt_all;
        const u32 M = mt.y, D = mt.x;
        const u32 mend = start ? p + M : 0u;
        const u32 pmax = scan_max(mend);
        u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)pmax, 0x138, 0xF, 0xF, false);
        if (before < cover) before = cover;
        const u32 Lb = start ? p - before : 0u;                                // literals since the match before (0-3: that one's token counts them)
        const u32 lsz = Lb >= 4u ? lzo_lit_size(Lb) : 0u;
        const u32 esz = start ? lsz + Lb + lzo_match_size(D, M) : 0u;
        const u32 incl = scan_add(esz);
        const u32 off = obase + incl - esz;
        const bool fits = start && off + esz <= cap;
        if (start && !fits) fail = true;
        u32 eaddr = 0xFFFFFFFFu, ebase = 0;
        if (fits) {
            u32 q = off;
            if (Lb >= 4u) { q += lzo_put_lit(dst + q, Lb); q += Lb; }
            else { for (u32 i = 0; i < Lb; i++) dst[q + i] = src[before + i]; q += Lb; }
            u32 epos;
            (void)lzo_put_match_def(dst + q, D, M, epos, ebase);
            eaddr = q + epos;
        }
        {   // the deferred byte of the token in front of mine (the start below me in this window, or `pend`) with my literal count
            const u64 below = stm & ((1ull << lane) - 1ull);
            const int pl = below ? 63 - (int)__builtin_clzll(below) : lane;
            u32 paddr = (u32)__builtin_amdgcn_ds_bpermute(pl << 2, (int)eaddr), pval = (u32)__builtin_amdgcn_ds_bpermute(pl << 2, (int)ebase);
            if (!below) { paddr = pend_addr; pval = pend_val; }
            if (start && paddr < cap) dst[paddr] = (u8)(pval | (Lb <= 3u ? Lb : 0u));
            const int l0 = 63 - (int)__builtin_clzll(stm);
            pend_addr = (u32)__builtin_amdgcn_readlane((int)eaddr, l0); pend_val = (u32)__builtin_amdgcn_readlane((int)ebase, l0);
        }
        {   // the literal runs (Lb >= 4), as in enc_parse_seq_kernel: every literal position whose unit starts in this window stores its own byte;
            // what the first start owns of earlier windows the wavefront copies
            const u64 above = (lane < 63 ? stm >> (lane + 1) : 0ull);
            const int s = above ? lane + 1 + (int)__builtin_ctzll(above) : lane;
            const u32 sbef = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)before);
            const u32 sbase = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)(off + lsz - before));
            const u32 srun = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)((fits && Lb >= 4u) ? 1u : 0u));
            if (above && !start && srun && p >= sbef && p < n) dst[sbase + p] = src[p];
            const int f0 = (int)__builtin_ctzll(stm);
            const u32 fbef = (u32)__builtin_amdgcn_readlane((int)before, f0);
            if (fbef < P && __builtin_amdgcn_readlane((int)((fits && Lb >= 4u) ? 1u : 0u), f0))
                wave_copy(dst + (u32)__builtin_amdgcn_readlane((int)(off + lsz), f0), src + fbef, P - fbef, lane);
        }
        obase += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u32 wmax = (u32)__builtin_amdgcn_readlane((int)pmax, 63);
        if (wmax > cover) cover = wmax;
    }
    // ---- behind the last match: 0-3 literals went out with it; four and more are a run of their own; then the end token
    u32 rest = n - cover;
    if (rest <= 3u) {                                                          // (they ride in the last token; there is one: the head ends behind a match with >= 4 bytes or a start to go)
        if (pend_addr < cap && lane == 0) dst[pend_addr] = (u8)(pend_val | rest);
        if (obase + rest <= cap) { if ((u32)lane < rest) dst[obase + (u32)lane] = src[cover + (u32)lane]; } else fail = true;
        obase += rest; rest = 0u;
    } else if (pend_addr < cap && lane == 0) dst[pend_addr] = (u8)pend_val;
    const u32 lsz = rest ? lzo_lit_size(rest) : 0u;
    const u32 total = obase + lsz + rest + 3u;
    if (total > cap) fail = true;
    const bool anyfail = __ballot(fail) != 0ull;
    if (!anyfail) {
        if (rest) { if (lane == 0) (void)lzo_put_lit(dst + obase, rest); wave_copy(dst + obase + lsz, src + cover, rest, lane); }
        if (lane == 0) { dst[total - 3u] = 0x11; dst[total - 2u] = 0; dst[total - 1u] = 0; }
    }
    finish(total, anyfail, ALZ_ST_OK);
}

template <int FMT>
static void launch_emit(hipStream_t s, u32 count, const u8* src, u8* dst, const alz_stream* streams, const u32* index, const mentry* match,
                        const u64* pos_off, const int* prev4, const int* prevm, u8* side, alz_result* results, alz_encode_aux* aux, const EncGeom& g,
                        u64* mask = nullptr) {
    // one stream per wavefront (lane 0 parses and emits; 64 streams per wavefront were the union of 64 divergent token paths), the
    // parse from the roles walk's start mask
    const int tail = FMT == ALZ_FMT_LZ4_BLOCK ? 5 : 0;
    if (mask) hipLaunchKernelGGL((enc_roles_kernel<false>), dim3(count), dim3(64), 0, s, src, streams, index, count, (mentry*)match, pos_off, prev4, prevm, mask, g, tail, (const u32*)nullptr, 0u);
    hipLaunchKernelGGL((enc_emit_kernel<FMT>), dim3(count), dim3(64), 0, s, src, dst, streams, index, count, match, pos_off, side, results, aux, g, 1u,
                       (const u64*)mask);
}

// one candidate per position and a format whose parse and emit are one kernel: the search is in that kernel too (no kernel B, no match array)
static bool searches_in_the_parse(int fmt, const EncGeom& g) {
    const bool par = fmt == ALZ_FMT_LZSS || fmt == ALZ_FMT_LZ10 || fmt == ALZ_FMT_LZ11 || fmt == ALZ_FMT_LZ40 || fmt == ALZ_FMT_YAZ0 || fmt == ALZ_FMT_YAY0 ||
                     fmt == ALZ_FMT_MIO0 || fmt == ALZ_FMT_CLZ0 || fmt == ALZ_FMT_BLZ || fmt == ALZ_FMT_LZHUDSON ||
                     fmt == ALZ_FMT_LZ4_BLOCK || fmt == ALZ_FMT_SNAPPY_RAW ||                        // (enc_parse_seq_kernel)
                     fmt == ALZ_FMT_PRS_BE || fmt == ALZ_FMT_PRS_LE ||                               // (enc_emit_prs_kernel)
                     fmt == ALZ_FMT_LZO;                                                             // (enc_parse_lzo_kernel)
    return par && g.max_chain == 1 && g.nprops <= 1 && !g.use_min_table && g.link16;
}

// 1: the format's emitter runs behind enc_roles_kernel and reads its start mask (the formats without a parallel emitter); the others walk inside their emitter
int alz_encode_format_needs_mask(int fmt) {
    switch (fmt) {
    case ALZ_FMT_SMSR00: case ALZ_FMT_FASTLZ: case ALZ_FMT_HIG: case ALZ_FMT_LZSHREK: case ALZ_FMT_WFLZ: case ALZ_FMT_WFLZ_BE: case ALZ_FMT_REFPACK:
    case ALZ_FMT_LZ02: case ALZ_FMT_CNS: case ALZ_FMT_CNX2: return 1;
    default: return 0;
    }
}

int alz_encode_geom_needs_match(int fmt, const void* geom) { EncGeom g; memcpy(&g, geom, sizeof(g)); return searches_in_the_parse(fmt, g) ? 0 : 1; }

template <int FMT>
static void launch_emit_par(hipStream_t s, u32 count, const u8* src, u8* dst, const alz_stream* streams, const u32* index, mentry* match,
                            const u64* pos_off, const int* prev4, const int* prevm, u64* mask, u8* side, alz_result* results, alz_encode_aux* aux, const EncGeom& g) {
    (void)mask;
    if (searches_in_the_parse(FMT, g)) hipLaunchKernelGGL((enc_parse_emit_kernel<FMT, true>), dim3(count), dim3(64), 0, s, src, dst, streams, index, count, match, pos_off, prev4, prevm, side, results, aux, g);
    else hipLaunchKernelGGL((enc_parse_emit_kernel<FMT, false>), dim3(count), dim3(64), 0, s, src, dst, streams, index, count, match, pos_off, prev4, prevm, side, results, aux, g);
}

// one pass over the stream whatever the hash width: formats whose matches reach back at most 8 KiB (enc_prev_cu_kernel<2, true>)
static bool uses_win_prev(const EncGeom& g) {
    return g.max_dist <= 8192 && (g.hash_bits > 15 || g.use_min_table);
}
static bool narrows_links(const EncGeom& g) {
    // (the switches of tools/narrow_experiment.sh live in tools/variants/r04_encode_switches.patch)
    return g.link16 && g.nprops <= 1 && g.hash_bits > 15;
}
int alz_encode_geom_narrows(const void* geom) { EncGeom g; memcpy(&g, geom, sizeof(g)); return narrows_links(g) ? 1 : 0; }

// Which kernel B for a stream, from maxChain 3 on?  The two-phase kernel (chains first, the pairs 64 at a time) keeps its lanes busy where
// candidates are many and short -- the synthetic batches: 94 against 138 ms per 10 000 x 256 KiB at quality 8 --, the one-position-per-lane
// kernel ends a walk at the first candidate of full length and does not measure one that cannot win -- real data, runs and repeated rows:
// 1 024 windows of Test.bmp as Yaz0 at quality 8 38 against 109 ms, as LZ4 blocks 59 against 293.  The probe looks at up to 1 024 positions
// of the stream: how often do the sixteen bytes at the position equal the sixteen at its first candidate?  (The 256 KiB windows of Test.bmp: 74 % on
// average -- the photographic ones 0.3 % --, the synthetic streams 0.8-4.3 %; the line is drawn at 25 %.  The probe, the second launch and the
// test in the two-phase kernel cost a synthetic batch 0.7 ms of 94.)
#ifndef ALZ_PROBE_THRESH16
#define ALZ_PROBE_THRESH16 4u      /* sixteenths of the sampled positions */
#endif
template <bool L16>
__global__ __launch_bounds__(64) void enc_probe_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                       const int* __restrict__ prev4, const u64* __restrict__ pos_off, EncGeom g, int tail_skip,
                                                       u32* __restrict__ sel, u32* __restrict__ list, u32 thresh16) {
    const u32 sid = index_list[blockIdx.x];
    if (sid == 0xFFFFFFFFu) return;               // (enc_scan_select_kernel's list: the stream goes the other way)
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const int n = (int)st.src_len - tail_skip, limit = n - 4;
    const int* p4 = prev4 + pos_off[sid];
    const int lane = (int)threadIdx.x;
    if (limit < 64) { if (lane == 0) sel[sid] = 0u; return; }     // (a stream of a few bytes: whichever)
    const int step = (limit + 1) / 1024 > 0 ? (limit + 1) / 1024 : 1;
    u32 hit = 0, tot = 0;
    for (int k = 0; k < 16; k++) {
        const int pos = (k * 64 + lane) * step;
        const bool in = pos <= limit;
        int d = 0;
        if (in) { if (L16) d = (int)reinterpret_cast<const unsigned short*>(p4)[pos]; else { const int c = p4[pos]; d = c < 0 ? 0 : pos - c; } }
        const bool cand = in && d >= g.min_dist && d <= g.max_dist && d > 0;
        bool eq = false;
        if (cand) { u64 a[2], b[2]; __builtin_memcpy(a, data + pos, 16); __builtin_memcpy(b, data + pos - d, 16); eq = a[0] == b[0] && a[1] == b[1]; }
        hit += (u32)__popcll(__ballot(eq)); tot += (u32)__popcll(__ballot(in));
    }
    if (lane == 0) {
        const u32 mine = hit * 16u >= tot * thresh16 ? 1u : 0u;
        sel[sid] = mine | ((tot ? hit * 1000u / tot : 0u) << 8);
        if (mine) list[1u + atomicAdd(list, 1u)] = sid;
    }
}

// Which streams go WITHOUT kernels A and B (round 6): the ones whose parse visits few positions.  Kernel B searches every position of a stream -- 46 CU-cycles per position at quality 8
// whatever the data (a wavefront walks on until the last of its 64 chains ends): 2 000 copies of a flat 256 KiB window of Test.bmp took it 38-52 ms, as long as a mixed window --, the
// managed parse searches only where its cursor stands (FindNextBestMatch :157-212): ~2 600 of the 262 144 positions of such a window.  But the scan walk is ONE wavefront per stream
// and serial: ~3-6 us per search, so what it costs is the stream's own latency (a mixed window with its ~50 000 searches: 141 ms against 26 us of kernel B's throughput), hidden behind the
// other streams' kernels A / B / parse on a second HIP stream.  The probe walks the real greedy / lazy parse (scan searches) from eight places of the stream, up to 16 searches and 2 KiB
// each, and adds up searches PER KiB place by place (a stream that is flat here and photographic there is as slow as its photographic part): up to ALZ_SCAN_MAX_PER_KIB on average the
// stream takes enc_scan_emit_kernel<FMT> (LZ4 blocks, raw Snappy: enc_scan_seq_kernel<FMT>) -- the choice decides time only, the bytes are MatchSearch's either way.  Writes two lists over the launch's streams, each with
// 0xFFFFFFFF where the stream went the other way.
#ifndef ALZ_SCAN_MAX_PER_KIB
#define ALZ_SCAN_MAX_PER_KIB 72     /* searches per KiB, a search that looks at ONE block of 1 024 positions counting 1 (4 + blocks quarters: the formats with 32 / 64 KiB windows look at up to 32 / 64) */
                                    /* 10 000 windows of 256 KiB of Test.bmp as Yaz0 at quality 8, ms per call: 40 -> 117.2, 56 -> 111.0, 72 -> 109.6 (262 without the path; counted in plain searches: 20 -> 126.9, 40 -> 109.5, 80 -> 112.1, 120 -> 159.6) */
#endif
#ifndef ALZ_SCAN_MIN_STREAMS
#define ALZ_SCAN_MIN_STREAMS 2048u  /* a scan stream costs its own LATENCY (a flat 256 KiB window ~30 ms, a 64 KiB one ~4), hidden only behind a launch whose other streams keep the GPU busy that
                                       long.  Windows of 256 KiB of Test.bmp as LZ4 blocks at quality 8 / 5, ms per call without -> with the path: 128 buffers 11.5 / 8.1 -> 23.9 / 27.7, 512: 22.2 / 13.8 ->
                                       33.8 / 31.2, 1 024: 36.4 / 21.8 -> 38.7 / 35.8, 10 000: 294 / 158 -> 144 / 109 (tools/scan_on_off.py with N=...); 64 KiB windows cross at ~256 buffers.  Below this
                                       many buffers of a format in a call: not taken. */
#endif
#ifndef ALZ_SCAN_MIN_LEN
#define ALZ_SCAN_MIN_LEN 16384     /* shorter buffers: whichever (the regular way) */
#endif
__global__ __launch_bounds__(64) void enc_scan_select_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams, const u32* __restrict__ index_list, u32 count,
                                                             EncGeom g, int tail_skip, int force, u32* __restrict__ idx_regular, u32* __restrict__ idx_scan, u32* __restrict__ taken,
                                                             const int* __restrict__ prev4, const u64* __restrict__ pos_off) {
    __shared__ unsigned short candl[64];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const u32 sid = index_list[bid];
    bool take = force != 0;
    if (!take && sid != 0xFFFFFFFFu) {
        const alz_stream st = streams[sid];
        const u8* data = src_base + st.src_off;
        const int n = (int)st.src_len - tail_skip, limit = n - 4;           // (what the finder is given: LZ4 searches source[0 : n - 5], LZ4.cs:208)
        const int* p4 = prev4 ? prev4 + pos_off[sid] : nullptr;             // (the formats with 32 / 64 KiB windows: the search follows kernel A's links behind the nearest blocks)
        if (n >= ALZ_SCAN_MIN_LEN) {
            int per_kib = 0;                                          // searches per KiB, summed over the eight places
            for (int k = 0; k < 8; k++) {
                const int start = (int)(((long long)limit * (2 * k + 1)) >> 4);
                const int end = start + 2048 < limit + 1 ? start + 2048 : limit + 1;
                int cur = start, cnt = 0, cost = 0;
                while (cur < end && cnt < 16) {
                    int d0, l0, d1, l1;
                    cost += 3 + benc_wave_scan_search(data, n, g, cur, d0, l0, candl, p4); cnt++;
                    if (l0 < g.min_len) { cur++; continue; }
                    if (l0 <= g.lazy && cur + 1 <= limit) { cost += 3 + benc_wave_scan_search(data, n, g, cur + 1, d1, l1, candl, p4); cnt++; cur += l1 > l0 ? 1 + l1 : l0; }
                    else cur += l0;
                }
                per_kib += (cost << 8) / (cur > start ? cur - start : 1);   // (quarters of a one-block search, per KiB)
                if (per_kib > 8 * ALZ_SCAN_MAX_PER_KIB) break;             // (already over: the synthetic batches leave after two places)
            }
            take = per_kib <= 8 * ALZ_SCAN_MAX_PER_KIB;
        }
    }
    if (threadIdx.x == 0) {
        const bool s2 = take && sid != 0xFFFFFFFFu;
        idx_regular[bid] = s2 ? 0xFFFFFFFFu : sid;
        idx_scan[bid] = s2 ? sid : 0xFFFFFFFFu;
        if (s2 && taken) atomicAdd(taken, 1u);                           // (the context's count of streams that went this way: alz_debug_scan_streams)
    }
}

// kernel A: the head table in LDS (hashBits = 15 + floor(sqrt(2 Q)) = 15..20, LzChainMatchFinder.cs:108-119) -- one pass with the
// tag / link rings where matches reach back at most 8 KiB, otherwise 2^(hashBits - 15) passes (+ 2 for the min-length table)
// Which way for a stream (see below): the share of DISTINCT 15-bit hashes among 4 x 1 024 consecutive positions.  Flat and repetitive data (few
// distinct words: the 15-bit chain's first entry is nearly always the one) narrows fast and makes kernel A's passes slow (crowded hash classes):
// 1 024 windows of Test.bmp as LZ4 blocks at quality 8 48.6 -> 37.3 ms; on data whose words are nearly all distinct (the synthetic batch,
// program text) a narrowing walk runs through two or three candidates per position and loses against the passes (145 -> 158 ms, 13.6 -> 14.5).
// Two lists of stream ids (cursor in front, unused slots stay 0xFFFFFFFF): [0] narrow, [pitch] kernel A at the finder's own width.
#ifndef ALZ_NARROW_RANGE
#define ALZ_NARROW_RANGE 4096         /* positions per workgroup of enc_narrow_lds_kernel at 4 KiB windows */
#endif
#ifndef ALZ_NARROW_SPLIT_MIN
#define ALZ_NARROW_SPLIT_MIN 2048u    /* streams of a launch from which the choice is per stream (eight rounds of kernel A's workgroups); below: the majority's way for all */
#endif
#ifndef ALZ_NARROW_MIN_THRESH16
#define ALZ_NARROW_MIN_THRESH16 11u  /* windows up to 8 KiB with the min-length table: narrow (behind 1 + 2 passes at 15 bits) from 11 / 16 distinct hashes on, the others through the one-pass kernel with tags.
                                        (Shares measured on the host: Test.bmp windows 0.01-0.24, a tenth of them -- photographs -- 0.67-0.92; program text 0.44-0.70; the synthetic batch 0.68-0.72.) */
#endif
#ifndef ALZ_NARROW_WGS
#define ALZ_NARROW_WGS 4096u      /* 64 KiB windows: workgroups of enc_narrow_kernel per launch, at least (each fetches the 64 KiB behind its range again) */
#endif
#ifndef ALZ_NARROW_THRESH16
#define ALZ_NARROW_THRESH16 4u      /* narrow below a quarter: Test.bmp 39.5 ms (8: 39.2, every stream: 37.3), program text 13.75 (8: 14.95, kernel A's passes: 13.6), the synthetic batch 147 (12: 161) */
#endif
__global__ __launch_bounds__(256) void enc_words_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams,
                                                        const u32* __restrict__ index_list, int tail_skip, u32* __restrict__ lists, u32 pitch, u32 thresh16, bool invert) {
    __shared__ u32 bm[1024];
    __shared__ u32 cnt[2];
    const u32 sid = index_list[blockIdx.x];
    if (sid == 0xFFFFFFFFu) return;               // (enc_scan_select_kernel's list: the stream goes the other way)
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const int limit = (int)st.src_len - tail_skip - 4;
    if (threadIdx.x < 2u) cnt[threadIdx.x] = 0;
    u32 fresh = 0, tot = 0;
    for (int k = 0; k < 4; k++) {
        for (u32 i = threadIdx.x; i < 1024u; i += 256u) bm[i] = 0;
        __syncthreads();
        const long long base = ((long long)limit + 1) * (2 * k + 1) / 8;
        for (u32 i = threadIdx.x; i < 1024u; i += 256u) {
            const long long p = base + i;
            if (p <= (long long)limit) {
                const u32 h = (load32(data + p) * 2654435761u) >> 17;
                const u32 bit = 1u << (h & 31u);
                if (!(atomicOr(&bm[h >> 5], bit) & bit)) fresh++;
                tot++;
            }
        }
        __syncthreads();
    }
    atomicAdd(&cnt[0], fresh); atomicAdd(&cnt[1], tot);
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool few = cnt[0] * 16u < cnt[1] * thresh16;
        const bool narrow = cnt[1] >= 256u && (invert ? !few : few);
        u32* l = lists + (narrow ? 0u : pitch);
        l[1u + atomicAdd(l, 1u)] = sid;
    }
}

// A launch of few streams does not split: each of the two forms of kernel A would run its own (partly empty) round of workgroups one after the
// other -- 256 windows of program text as Yaz0 at quality 12 23.5 -> 26.4 ms.  All of them go the way of the majority.
__global__ __launch_bounds__(256) void enc_words_merge_kernel(u32* __restrict__ lists, u32 pitch) {
    const u32 na = lists[0], nb = lists[pitch];
    if (na == 0u || nb == 0u) return;
    u32* to = na >= nb ? lists : lists + pitch;
    u32* from = na >= nb ? lists + pitch : lists;
    const u32 nt = na >= nb ? na : nb, nf = na >= nb ? nb : na;
    for (u32 i = threadIdx.x; i < nf; i += 256u) { to[1u + nt + i] = from[1u + i]; }
    __syncthreads();
    for (u32 i = threadIdx.x; i < nf; i += 256u) from[1u + i] = 0xFFFFFFFFu;
    if (threadIdx.x == 0) { to[0] = nt + nf; from[0] = 0u; }
}

// Kernel A at 15 bits for every hash width (round 4).  The finder's hash is the top hashBits bits of ONE product (ComputeHash :288-299), so the
// positions with my hashBits-bit hash are a subsequence of the positions with my 15-bit hash, in the same order: prev() at hashBits bits is the
// first position on the 15-bit chain whose product agrees in the top hashBits bits.  Kernel A pays 2^(hashBits - 15) passes over a stream for
// the wide hash where the head table does not fit the LDS (the formats with windows above 8 KiB: 64.5 against 14.7 ms per 10 000 x 256 KiB as
// LZ4 blocks at quality 8); one pass at 15 bits and this kernel -- a position per thread walks its 15-bit chain to the first agreeing
// position, usually the first or second -- cost less.  A link further than maxDistance back is stored as none: that ends a walk either way
// (LzChainMatchFinder.cs:259-260).  16-bit links, one property set, no min-length table (its links live in the array this kernel writes).
// Per stream, where it pays: enc_words_kernel above.
__global__ __launch_bounds__(256) void enc_narrow_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams,
                                                         const u32* __restrict__ index_list, const int* __restrict__ prev15,
                                                         int* __restrict__ prevw, const u64* __restrict__ pos_off, EncGeom g, int tail_skip) {
    const u32 sid = index_list[blockIdx.y];
    if (sid == 0xFFFFFFFFu) return;               // (a list written on the device, enc_words_kernel: unused slots)
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const int limit = (int)st.src_len - tail_skip - 4;
    const unsigned short* l15 = reinterpret_cast<const unsigned short*>(prev15 + pos_off[sid]);
    unsigned short* lw = reinterpret_cast<unsigned short*>(prevw + pos_off[sid]);
    const u32 sh = 32u - (u32)g.hash_bits;
    // (a workgroup takes ONE contiguous range of the stream: the window behind it stays in its XCD's L2 -- with the workgroups of a stream interleaved,
    // every XCD pulled every window of every stream: 50 GB of fetches per 2.6 GB of input for the synthetic LZSS batch at quality 8)
    const long long per = (((long long)limit + 1 + gridDim.x - 1) / gridDim.x + 255) & ~255ll;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < (long long)limit + 1 ? lo + per : (long long)limit + 1;
    for (long long p64 = lo + threadIdx.x; p64 < hi; p64 += 256) {
        const int pos = (int)p64;
        const u32 own = load32(data + pos) * 2654435761u;
        u32 link = l15[pos], d = 0, res = 0;
        while (link != 0u) {
            d += link;
            if (d > (u32)g.max_dist) break;
            const int c = pos - (int)d;
            if (((load32(data + c) * 2654435761u) ^ own) >> sh == 0u) { res = d; break; }
            link = l15[c];
        }
        lw[pos] = (unsigned short)res;
    }
}

// The same for windows up to 8 KiB with the range and the window behind it in LDS: LOOK bytes and links of look-back + LOOK of the range per
// workgroup (24 KB at 4 KiB windows with ranges of 4 KiB: six workgroups per CU), every hop of a walk two LDS reads instead of two scattered loads.
template <int LOOK, int RANGE>
__global__ __launch_bounds__(256) void enc_narrow_lds_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams,
                                                             const u32* __restrict__ index_list, const int* __restrict__ prev15,
                                                             int* __restrict__ prevw, const u64* __restrict__ pos_off, EncGeom g, int tail_skip) {
    __shared__ __attribute__((aligned(16))) u8 sd[LOOK + RANGE + 16];
    __shared__ __attribute__((aligned(16))) unsigned short sl[LOOK + RANGE];
    const u32 sid = index_list[blockIdx.y];
    if (sid == 0xFFFFFFFFu) return;               // (a list written on the device, enc_words_kernel: unused slots)
    const alz_stream st = streams[sid];
    const u8* data = src_base + st.src_off;
    const int limit = (int)st.src_len - tail_skip - 4;
    const unsigned short* l15 = reinterpret_cast<const unsigned short*>(prev15 + pos_off[sid]);
    unsigned short* lw = reinterpret_cast<unsigned short*>(prevw + pos_off[sid]);
    const u32 sh = 32u - (u32)g.hash_bits;
    for (long long lo64 = (long long)blockIdx.x * RANGE; lo64 <= (long long)limit; lo64 += (long long)gridDim.x * RANGE) {
        const int lo = (int)lo64, lb = lo >= LOOK ? lo - LOOK : 0;
        const int hi = lo + RANGE <= limit + 1 ? lo + RANGE : limit + 1;       // positions [lo, hi)
        const int nb = hi + 3 - lb;                                             // bytes [lb, hi + 3): the last position's word
        __syncthreads();                                                        // (the round before has finished with the arrays)
        for (int i = 4 * (int)threadIdx.x; i < nb; i += 1024) { const u32 v = load32(data + lb + i); __builtin_memcpy(sd + i, &v, 4); }   // (up to three bytes past hi + 3: inside the stream or its slack)
        for (int i = 2 * (int)threadIdx.x; i < hi - lb; i += 512) { const u32 v = *reinterpret_cast<const u32*>(l15 + lb + i); *reinterpret_cast<u32*>(sl + i) = v; }
        __syncthreads();
        for (int pos = lo + (int)threadIdx.x; pos < hi; pos += 256) {
            u32 ow; __builtin_memcpy(&ow, sd + (pos - lb), 4);
            const u32 own = ow * 2654435761u;
            u32 link = sl[pos - lb], d = 0, res = 0;
            while (link != 0u) {
                d += link;
                if (d > (u32)g.max_dist) break;
                const int c = pos - (int)d - lb;                                // (>= 0: LOOK >= maxDistance)
                u32 w; __builtin_memcpy(&w, sd + c, 4);
                if (((w * 2654435761u) ^ own) >> sh == 0u) { res = d; break; }
                link = sl[c];
            }
            lw[pos] = (unsigned short)res;
        }
    }
}

// kernel A over segments (alz_encode_seg.h): `aseg` = the scratch behind the launch's segment records, 0 = one workgroup per buffer
struct AsegPlan { void* mem; u32 SA, ka, W, stride; };
static hipError_t launch_prev(hipStream_t stream, const u8* src, const alz_stream* d_streams, const uint32_t* d_index, uint32_t count,
                              int* d_prev4, int* d_prevm, const uint64_t* d_pos_off, const EncGeom& g, int tail, bool split_passes, bool no_win);
static hipError_t launch_prev_aseg(hipStream_t stream, const u8* src, const alz_stream* d_streams, const uint32_t* d_index, uint32_t count, uint32_t max_len,
                                   int* d_prev4, const uint64_t* d_pos_off, const EncGeom& g15, const AsegPlan& a, int tail) {
    const size_t V = (size_t)count * a.ka;
    u8* base = (u8*)a.mem;
    alz_stream* vs = (alz_stream*)base; base += V * sizeof(alz_stream);
    u64* vpos = (u64*)base; base += V * sizeof(u64);
    u32* vindex = (u32*)base; base += ((V * sizeof(u32)) + 63u) & ~(size_t)63u;
    int* seg4 = (int*)base;
    hipLaunchKernelGGL(enc_aseg_setup_kernel, dim3((u32)((V + 255u) / 256u)), dim3(256), 0, stream, d_streams, d_index, count, vs, vindex, vpos, a.ka, a.SA, a.W, a.stride, tail);
    const hipError_t e = launch_prev(stream, src, vs, vindex, (u32)V, seg4, nullptr, vpos, g15, 0, false, true);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(enc_aseg_gather_kernel, dim3((max_len + 255u) / 256u, count), dim3(256), 0, stream, d_streams, d_index, (const int*)seg4, d_prev4, d_pos_off, a.ka, a.SA, a.W, a.stride, tail);
    return hipSuccess;
}
static hipError_t launch_prev(hipStream_t stream, const u8* src, const alz_stream* d_streams, const uint32_t* d_index, uint32_t count,
                              int* d_prev4, int* d_prevm, const uint64_t* d_pos_off, const EncGeom& g, int tail, bool split_passes = false, bool no_win = false) {
    if (g.hash_bits < 15 || g.hash_bits > 20) return hipErrorInvalidValue;
    if (uses_win_prev(g) && !no_win) hipLaunchKernelGGL((enc_prev_cu_kernel<2, true>), dim3(count), dim3(1024), 0, stream, src, d_streams, d_index, count, d_prev4, d_prevm, d_pos_off, g, tail);
    else if (g.hash_bits == 15 && !g.use_min_table) hipLaunchKernelGGL((enc_prev_cu_kernel<2, false>), dim3(count), dim3(1024), 0, stream, src, d_streams, d_index, count, d_prev4, d_prevm, d_pos_off, g, tail);
    else {
        const u32 passes = (1u << (g.hash_bits - 15)) + (g.use_min_table ? 2u : 0u);       // (as the kernel counts them)
        // (a workgroup per (stream, pass) where a workgroup per stream leaves CUs idle: 16 x 64 KiB as raw Snappy at quality 8 -- 16 passes -- 383 -> 32 us of the
        // call's 560; from 256 streams on the two arrangements are the same work on the same CUs)
        const bool split = split_passes || count < 256u;
        hipLaunchKernelGGL((enc_prev_cu_kernel<3, false>), dim3(count, split ? passes : 1u), dim3(1024), 0, stream, src, d_streams, d_index, count, d_prev4, d_prevm, d_pos_off, g, tail);
    }
    return hipSuccess;
}

// Where kernel B stops comparing (EncGeom.b_cap).  A candidate that reaches the cap ends the walk of its position and marks it; the parse
// searches a marked position exactly -- the whole chain, by the whole wavefront -- only if its cursor ever stands on it.  So a low cap moves
// work from "every position inside a long match" to "the positions the parse visits": on real data (tools/bcap_sweep.sh: 1 024 - 2 048
// windows of Test.bmp per format, ms per call at caps 2 040 / 256 / 48) it pays where a walk is short (quality 0-4, one to five candidates: LZ4
// blocks Q0 43 / 31 / 23, Yaz0 Q1 34 / 34 / 25) and where it is very long (quality 11-15, 64 to 1 024 candidates, which the cap spares every
// position of a long match: LZ11 Q12 686 / 401 / 164, LZ4 Q15 908 / 486 / 394); in between (quality 5-10, 6 to 32 candidates) the exact search of
// a visited position costs more than the cap saves and 256 is the better line (LZ11 Q8 92 / 76 / 100).  Formats whose longest match is below 96
// bytes (Snappy: 64) lose with any cap (Q0 20 -> 30 ms at 48): none.  The synthetic batches (matches of at most 18 bytes) never reach a cap.
static int choose_b_cap(const EncGeom& g) {
    if (g.max_len < 96) return ALZ_LEN_CAP;
    const int cap = (g.max_chain <= 5 || g.max_chain >= 64) ? 48 : 256;
    return g.max_len > cap ? cap : ALZ_LEN_CAP;
}

// kernel B over `count` streams; `wg_cap`: workgroups per stream of the one-position-per-lane form (32 in a batch; a lone stream takes
// as many as it has blocks of 256 positions)
static void launch_match(hipStream_t stream, const u8* src, const alz_stream* d_streams, const uint32_t* d_index, uint32_t count, uint32_t max_len,
                         const int* d_prev4, const int* d_prevm, void* d_match, const uint64_t* d_pos_off, const EncGeom& g_in, int tail, u32 wg_cap, bool dense_ok = true, u32* d_sel = nullptr, u32 sel_pitch = 0) {
    const EncGeom& g = g_in;
    u32 bx = (max_len + 255) / 256; if (bx == 0) bx = 1; if (bx > 4096) bx = 4096;
    if (bx > wg_cap) bx = wg_cap;
    // (workgroups per stream, each with one contiguous range: 32 -- 8 Ki positions of a 256 KiB stream, 4 KiB of history in front of them fetched
    // again -- move 9.1 GB at quality 0, 128 move 12.7, both in 13.6 ms; one position per thread, ten million workgroups per launch: 18.6 ms)
    if (dense_ok && g.max_chain >= 3 && g.max_chain <= 4095 && g.max_dist < (1 << 28)) {        // (from maxChain 3 on: the chains first, the pairs 64 at a time)
        const bool dyn = g.max_chain >= 8;
        u32 bd = dyn ? (max_len + 255u) / 256u : (max_len + 63u) / 64u; if (bd == 0) bd = 1; if (bd > 4096u) bd = 4096u;
#ifndef ALZ_DENSE_XLOG
#define ALZ_DENSE_XLOG 3u    /* runs of eight blocks per XCD (round 6, cfg5 at quality 8, kernel B ms / fetch x 2 GB: runs of 4 45.1 / 40.5, of 8 45.3 / 23.2, of 16 47.7 / 15.4, of 64 53.2 / 9.7: profiles/r06_cfg5_traffic.md) */
#endif
        u32 xlog = !dyn ? 7u : g.max_chain <= 128 ? ALZ_DENSE_XLOG : 0u;        //           // runs of consecutive blocks per XCD (enc_match_dense_kernel; the longest chains lose with them: 104.9 -> 113.7 ms at quality 15, while quality 12 gains 85.0 -> 83.6)
        while (xlog && (8u << xlog) > bd) xlog--;
        if (xlog) bd = (bd + (8u << xlog) - 1u) / (8u << xlog) * (8u << xlog);
        // (per stream: the two-phase kernel or the one-position-per-lane one -- enc_probe_kernel; both are launched, each leaves the other's streams alone)
        const u32* sel = nullptr;
        if (d_sel && g.link16 && g.nprops <= 1) {
            const u32 thr = ALZ_PROBE_THRESH16;
            u32* list = d_sel + sel_pitch;                          // [0]: how many streams, then their ids
            (void)hipMemsetAsync(list, 0, 4, stream);
            hipLaunchKernelGGL((enc_probe_kernel<true>), dim3(count), dim3(64), 0, stream, src, d_streams, d_index, d_prev4, d_pos_off, g, tail, d_sel, list, thr);
            sel = d_sel;
            const u32 gy = count < 512u ? count : 512u;
            // (lanes that take the next position when their walk ends -- enc_match_dyn_kernel -- where walks are long or compares short; with
            // long matches AND walks of 6-32 candidates the lanes' compare loops fall out of step and every one of them runs for the whole
            // wavefront: 1 024 windows at quality 8 as Yaz0 40 -> 47 ms, as LZ4 blocks 56 -> 63, but as LZSS 17.1 -> 14.1, and at quality 12
            // Yaz0 139 -> 125, LZ11 ~160 -> 92, LZSS 40 -> 26)
            const bool dynk = g.max_len <= 32 || g.max_chain >= 64;
            if (dynk) {
                if (g.use_min_table) hipLaunchKernelGGL((enc_match_dyn_kernel<true>), dim3(bx, gy), dim3(256), 0, stream, src, d_streams, d_index, d_prev4, d_prevm, (mentry*)d_match, d_pos_off, g, tail, list);
                else hipLaunchKernelGGL((enc_match_dyn_kernel<false>), dim3(bx, gy), dim3(256), 0, stream, src, d_streams, d_index, d_prev4, d_prevm, (mentry*)d_match, d_pos_off, g, tail, list);
            }
            else if (g.use_min_table) hipLaunchKernelGGL((enc_match_kernel<true, true, true>), dim3(bx, gy), dim3(256), 0, stream, src, d_streams, d_index, d_prev4, d_prevm, (mentry*)d_match, d_pos_off, g, tail, list);
            else hipLaunchKernelGGL((enc_match_kernel<false, true, true>), dim3(bx, gy), dim3(256), 0, stream, src, d_streams, d_index, d_prev4, d_prevm, (mentry*)d_match, d_pos_off, g, tail, list);
        }
#define ALZ_LB(K, grid, block) hipLaunchKernelGGL(K, grid, block, 0, stream, src, d_streams, d_index, d_prev4, d_prevm, (mentry*)d_match, d_pos_off, g, tail, xlog, sel)
        const dim3 gd(bd, count);
        if (dyn) {
            if (g.use_min_table) { if (g.link16) ALZ_LB((enc_match_dense_kernel<true, true, 256, true>), gd, dim3(64)); else ALZ_LB((enc_match_dense_kernel<true, true, 256, false>), gd, dim3(64)); }
            else { if (g.link16) ALZ_LB((enc_match_dense_kernel<false, true, 256, true>), gd, dim3(64)); else ALZ_LB((enc_match_dense_kernel<false, true, 256, false>), gd, dim3(64)); }
        } else {
            if (g.use_min_table) { if (g.link16) ALZ_LB((enc_match_dense_kernel<true, false, 64, true>), gd, dim3(64)); else ALZ_LB((enc_match_dense_kernel<true, false, 64, false>), gd, dim3(64)); }
            else { if (g.link16) ALZ_LB((enc_match_dense_kernel<false, false, 64, true>), gd, dim3(64)); else ALZ_LB((enc_match_dense_kernel<false, false, 64, false>), gd, dim3(64)); }
        }
#undef ALZ_LB
        return;
    }
#define ALZ_LB(K, grid, block) hipLaunchKernelGGL(K, grid, block, 0, stream, src, d_streams, d_index, d_prev4, d_prevm, (mentry*)d_match, d_pos_off, g, tail)
    if (!dense_ok && g.max_chain >= 3 && g.max_len <= 32 && g.link16 && g.nprops <= 1) {   // (the whole-GPU path of ONE stream; short compares only: see above)
        if (g.use_min_table) ALZ_LB((enc_match_dyn_kernel<true>), dim3(bx, count), dim3(256)); else ALZ_LB((enc_match_dyn_kernel<false>), dim3(bx, count), dim3(256));
        return;
    }
    // (PRUNE in match_search_b: a candidate that cannot win is not measured -- with 16-bit links, i.e. every finder but RefPack's and FastLZ level 2's)
    // (not with one candidate per position -- quality 0 --: nothing to prune there, and the test costs the synthetic LZ4 batch 5 %: 58.2 -> 61.1 ms)
    const bool prune = g.link16 && g.max_chain > 1;
    if (g.use_min_table) { if (prune) ALZ_LB((enc_match_kernel<true, true, true>), dim3(bx, count), dim3(256)); else if (g.link16) ALZ_LB((enc_match_kernel<true, true>), dim3(bx, count), dim3(256)); else ALZ_LB((enc_match_kernel<true, false>), dim3(bx, count), dim3(256)); }
    else { if (prune) ALZ_LB((enc_match_kernel<false, true, true>), dim3(bx, count), dim3(256)); else if (g.link16) ALZ_LB((enc_match_kernel<false, true>), dim3(bx, count), dim3(256)); else ALZ_LB((enc_match_kernel<false, false>), dim3(bx, count), dim3(256)); }
#undef ALZ_LB
}

hipError_t alz_launch_encode(int fmt, hipStream_t stream, const void* d_src, void* d_dst, const alz_stream* d_streams, const uint32_t* d_index,
                             uint32_t count, uint32_t max_len, int* d_prev4, int* d_prevm, int* d_narrow, void* d_match,
                             const uint64_t* d_pos_off, void* d_side, void* d_mask, alz_result* d_results, alz_encode_aux* d_aux, const void* geom,
                             uint32_t* d_sel, uint32_t sel_pitch, void* d_seg, uint32_t seg_len, uint32_t seg_kmax, int scan_mode, uint32_t* d_scan_taken, const alz_encode_side* side_q) {
    if (count == 0) return hipSuccess;
    EncGeom g; memcpy(&g, geom, sizeof(g));
    // (the segmented path of a small batch, alz_encode_seg.h: no cap -- its longest match is at most 2 040 bytes, kernel B has the GPU to itself, and
    // every capped position the roles walk stands on costs that ONE wavefront two exact searches: 16 x 64 KiB of Test.bmp as Yaz0 at quality 8 1.17 ms of walk)
    // (LZ4 blocks and LZO keep the cap: their segments are walked all at once -- alz_encode_seg_seq.h, enc_spec_walk_kernel --, so the exact searches of capped cursors run side by side,
    // while no cap means every position of a flat stretch compared over 2 040 bytes: 256 x 64 KiB of Test.bmp at quality 8, kernel B 1.70 ms of the call's 3.37)
    g.b_cap = (d_seg != nullptr && seg_len != 0u && !seg_spec_format(fmt)) ? ALZ_LEN_CAP : choose_b_cap(g);
#ifndef ALZ_SPEC_BCAP_SHORT
#define ALZ_SPEC_BCAP_SHORT 48
#endif
    if (d_seg != nullptr && seg_len != 0u && seg_spec_format(fmt) && g.max_chain <= 5 && g.max_len > ALZ_SPEC_BCAP_SHORT) g.b_cap = ALZ_SPEC_BCAP_SHORT;
    // (a lower cap for them, -DALZ_SPEC_BCAP=48 / 96 / 128 against choose_b_cap's 256 at quality 8, 256 x 64 KiB of Test.bmp, ms per call: windows 4 KiB apart 2.10 / 2.04 / 2.07 against 2.23,
    // windows spread over the whole file -- flat stretches, where the true cursor lands on capped positions and its searches stay serial per buffer -- 6.37 / 3.67 / 3.78 against 2.69: not taken)
    const u8* src = (const u8*)d_src; u8* dst = (u8*)d_dst;
    const int tail = fmt == ALZ_FMT_LZ4_BLOCK ? 5 : 0;
    // ---- the streams whose parse visits few positions go without kernels A and B (enc_scan_select_kernel; scan_mode 0: the probe decides, 1: every stream, 2: none).
    // The flag-bit formats of enc_parse_emit_kernel with windows up to 8 KiB, one property set, no min-length table (quality 2-9), a full batch (not the segmented path).
    const u32* d_index_scan = nullptr;
    bool scan_joined = true; const alz_encode_side* scan_side = nullptr;
    const bool seqf = fmt == ALZ_FMT_LZ4_BLOCK || fmt == ALZ_FMT_SNAPPY_RAW;          // (enc_scan_seq_kernel; windows of 64 / 32 KiB: the nearest blocks scanned, kernel A's links behind them)
    const bool scan_fam = seqf || fmt == ALZ_FMT_LZSS || fmt == ALZ_FMT_LZ10 || fmt == ALZ_FMT_LZ11 || fmt == ALZ_FMT_LZ40 || fmt == ALZ_FMT_YAZ0 || fmt == ALZ_FMT_YAY0 || fmt == ALZ_FMT_MIO0 ||
                          fmt == ALZ_FMT_CLZ0 || fmt == ALZ_FMT_BLZ || fmt == ALZ_FMT_LZHUDSON;
    const bool scan_ok = scan_mode != 2 && scan_fam && d_sel != nullptr && !(d_seg != nullptr && seg_len != 0u) && g.nprops <= 1 && !g.use_min_table && g.max_chain >= 3 &&
                         g.max_chain <= 32 /* (a block's candidates are measured at once, two lanes each at least) */ && (seqf || g.max_dist <= 8192) && g.link16 && !searches_in_the_parse(fmt, g) &&
                         (scan_mode == 1 || (g.max_len >= 64 && count >= ALZ_SCAN_MIN_STREAMS));  /* (matches of at most 18 bytes -- LZ10, MIO0, the default LZSS -- keep kernel B's compares short and every stream above the probe's
                                                                  line: 10 000 windows of Test.bmp at quality 8 as LZ10 86.7 ms without the path, 95.0 with it; LZSS 86.5 / 106.1.  Forced: the parity tests.) */
    // (`links`: kernel A's, for the formats whose search follows them behind the nearest blocks -- their select + scan kernels are launched behind kernel A, which then runs for every stream)
    auto launch_scan = [&](const int* links) -> hipError_t {
        u32* idx_regular = d_sel + 2u * (size_t)sel_pitch + 64u;      // (behind the probe's and the narrowing's lists; sel_pitch words each)
        u32* idx_scan = idx_regular + sel_pitch;
        hipLaunchKernelGGL(enc_scan_select_kernel, dim3(count), dim3(64), 0, stream, src, d_streams, d_index, count, g, tail, scan_mode == 1 ? 1 : 0, idx_regular, idx_scan, d_scan_taken, links, d_pos_off);
        d_index = idx_regular; d_index_scan = idx_scan;
        // the scan streams' ONE kernel: on the side stream where there is one (a wavefront per stream walking serially -- latency, not throughput -- beside the other streams'
        // kernels A / B / parse, which fill the GPU), joined at the end of this launch
        hipStream_t sq = stream;
        if (side_q && side_q->s && side_q->fork && side_q->join && hipEventRecord(side_q->fork, stream) == hipSuccess && hipStreamWaitEvent(side_q->s, side_q->fork, 0) == hipSuccess) sq = side_q->s;
        u8* side = (u8*)d_side;
#define ALZ_SCANK(F) case F: hipLaunchKernelGGL((enc_scan_emit_kernel<F>), dim3(count), dim3(64), 0, sq, src, dst, d_streams, d_index_scan, count, d_pos_off, side, d_results, d_aux, g); break;
        switch (fmt) {
        ALZ_SCANK(ALZ_FMT_LZSS) ALZ_SCANK(ALZ_FMT_LZ10) ALZ_SCANK(ALZ_FMT_LZ11) ALZ_SCANK(ALZ_FMT_LZ40) ALZ_SCANK(ALZ_FMT_YAZ0) ALZ_SCANK(ALZ_FMT_YAY0) ALZ_SCANK(ALZ_FMT_MIO0)
        ALZ_SCANK(ALZ_FMT_CLZ0) ALZ_SCANK(ALZ_FMT_BLZ) ALZ_SCANK(ALZ_FMT_LZHUDSON)
        case ALZ_FMT_LZ4_BLOCK: hipLaunchKernelGGL((enc_scan_seq_kernel<ALZ_FMT_LZ4_BLOCK>), dim3(count), dim3(64), 0, sq, src, dst, d_streams, d_index_scan, count, links, d_pos_off, d_results, d_aux, g); break;
        case ALZ_FMT_SNAPPY_RAW: hipLaunchKernelGGL((enc_scan_seq_kernel<ALZ_FMT_SNAPPY_RAW>), dim3(count), dim3(64), 0, sq, src, dst, d_streams, d_index_scan, count, links, d_pos_off, d_results, d_aux, g); break;
        default: break;
        }
#undef ALZ_SCANK
        if (sq != stream) { if (hipEventRecord(side_q->join, sq) != hipSuccess) return hipGetLastError(); scan_joined = false; scan_side = side_q; }
        return hipSuccess;
    };
    if (scan_ok && !seqf) { const hipError_t es = launch_scan(nullptr); if (es != hipSuccess) return es; }
    AsegPlan aseg = { nullptr, 0, 0, 0, 0 };                                   // (kernel A over segments: a launch on the segmented path with at most 128 buffers)
    if (d_seg != nullptr && seg_len != 0u) {
        size_t ab = 0; const u32 hist = seg_rec_hist(fmt, g, seg_len);      // (as alz_encode_segmented sized the records)
        if (alz_encode_aseg(geom, count, max_len, &aseg.SA, &aseg.ka, &aseg.W, &aseg.stride, &ab))
            aseg.mem = (u8*)d_seg + ((alz_encode_seg_bytes(count, seg_kmax, hist) + 255u) & ~(size_t)255u);
    }
    if (narrows_links(g) && d_narrow != nullptr && d_sel != nullptr) {
        // Kernel A at 15 bits and the links of the finder's own hash width narrowed from them (enc_narrow_kernel), into an array of their own: what
        // follows reads its links there (the min-length table's links, where there are any, are kernel A's own either way).  Which streams:
        //   windows up to 8 KiB, no min-length table (quality 1-9): all -- a 15-bit chain meets a collision once in eight positions, the walk is short
        //     on any data, and range + window fit the LDS: synthetic LZSS batch at quality 8 94.2 -> 84.5 ms, Test.bmp windows -2 %;
        //   windows up to 8 KiB with the min-length table (quality >= 10): those of MANY distinct words.  Kernel A's alternative there is its one-pass
        //     form with tags, which does both tables at once and suits repetitive data (Test.bmp as Yaz0 at quality 12 121.8 ms against 126.9 behind
        //     1 + 2 passes at 15 bits), data of many distinct words does not (synthetic LZSS batch at quality 15 188.5 -> 175.3 ms);
        //   64 KiB windows: those of FEW distinct words (enc_words_kernel); the others through kernel A's 2^(hashBits - 15) passes.
        EncGeom g15 = g; g15.hash_bits = 15;
        // (workgroups per stream, each with one contiguous range -- and the window behind it fetched again: ranges of 8 KiB for windows up to 8 KiB -- 16 KiB
        // move the same 15 GB: what is fetched are the lines of the scattered candidate words --; for 64 KiB windows as few as still fill the GPU)
        u32 bx = (max_len + 255u) / 256u; if (bx == 0u) bx = 1u;
#ifndef ALZ_NARROW_BX_MAX
#define ALZ_NARROW_BX_MAX 256u
#endif
        // (... and for a handful of long buffers more than 32 each: 16 x 4 MiB as LZ4 blocks at quality 8 were 512 workgroups, two wavefronts per SIMD for a kernel of dependent loads)
        if (!uses_win_prev(g)) { const u32 want = (ALZ_NARROW_WGS + count - 1u) / count; if (bx > want) bx = want; if (bx > ALZ_NARROW_BX_MAX) bx = ALZ_NARROW_BX_MAX; }
        else if (bx > 32u) bx = 32u;
        auto narrow = [&](const u32* list) {
            if (g.max_dist <= 8192) {                                               // (range and window in LDS)
                const u32 look = g.max_dist <= 4096 ? 4096u : 8192u, range = look == 4096u ? ALZ_NARROW_RANGE : 8192u;
                u32 gx = (max_len + range - 1u) / range; if (gx == 0u) gx = 1u; if (gx > 4096u) gx = 4096u;
                if (look == 4096u) hipLaunchKernelGGL((enc_narrow_lds_kernel<4096, ALZ_NARROW_RANGE>), dim3(gx, count), dim3(256), 0, stream, src, d_streams, list, d_prev4, d_narrow, d_pos_off, g, tail);
                else hipLaunchKernelGGL((enc_narrow_lds_kernel<8192, 8192>), dim3(gx, count), dim3(256), 0, stream, src, d_streams, list, d_prev4, d_narrow, d_pos_off, g, tail);
                return;
            }
            hipLaunchKernelGGL(enc_narrow_kernel, dim3(bx, count), dim3(256), 0, stream, src, d_streams, list, d_prev4, d_narrow, d_pos_off, g, tail);
        };
        if (uses_win_prev(g) && !g.use_min_table) {
            const hipError_t e15 = aseg.mem ? launch_prev_aseg(stream, src, d_streams, d_index, count, max_len, d_prev4, d_pos_off, g15, aseg, tail)
                                            : launch_prev(stream, src, d_streams, d_index, count, d_prev4, d_prevm, d_pos_off, g15, tail);
            if (e15 != hipSuccess) return e15;
            narrow(d_index);
        } else {
            const bool winm = uses_win_prev(g);                                                        // (small windows with the min-length table: the choice the other way round)
            u32* l_narrow = d_sel; u32* l_wide = d_sel + sel_pitch + 2u;                               // (count + 1 words each; the probe of kernel B takes the array over afterwards)
            (void)hipMemsetAsync(d_sel, 0xFF, ((size_t)sel_pitch + 2u + count + 1u) * sizeof(u32), stream);
            (void)hipMemsetAsync(l_narrow, 0, 4, stream); (void)hipMemsetAsync(l_wide, 0, 4, stream);
            hipLaunchKernelGGL(enc_words_kernel, dim3(count), dim3(256), 0, stream, src, d_streams, d_index, tail, d_sel, sel_pitch + 2u,
                               winm ? ALZ_NARROW_MIN_THRESH16 : ALZ_NARROW_THRESH16, winm);
            if (count < ALZ_NARROW_SPLIT_MIN) hipLaunchKernelGGL(enc_words_merge_kernel, dim3(1), dim3(256), 0, stream, d_sel, sel_pitch + 2u);
            const hipError_t e15 = (aseg.mem && !winm) ? launch_prev_aseg(stream, src, d_streams, l_narrow + 1, count, max_len, d_prev4, d_pos_off, g15, aseg, tail)
                                                       : launch_prev(stream, src, d_streams, l_narrow + 1, count, d_prev4, d_prevm, d_pos_off, g15, tail, false, true);
            if (e15 != hipSuccess) return e15;
            narrow(l_narrow + 1);
            const hipError_t ew = launch_prev(stream, src, d_streams, l_wide + 1, count, d_narrow, d_prevm, d_pos_off, g, tail);
            if (ew != hipSuccess) return ew;
        }
        d_prev4 = d_narrow;
    } else {
        const hipError_t ea = (aseg.mem && g.hash_bits == 15) ? launch_prev_aseg(stream, src, d_streams, d_index, count, max_len, d_prev4, d_pos_off, g, aseg, tail)
                                                              : launch_prev(stream, src, d_streams, d_index, count, d_prev4, d_prevm, d_pos_off, g, tail);
        if (ea != hipSuccess) return ea;
    }
    if (scan_ok && seqf) { const hipError_t es = launch_scan(d_prev4); if (es != hipSuccess) return es; }
    const bool segmented = d_seg != nullptr && seg_len != 0u;                   // (a batch of few buffers: alz_encode_seg.h -- always behind kernel B)
    const u32 wgc = !segmented ? 32u : count < 128u ? 128u : 64u;           // (workgroups per buffer in kernel B: 64 buffers of 64 KiB at quality 8 0.92 -> 0.79 ms with 128, 256 buffers 2.15 -> 2.08 with 64)
    if (segmented || !searches_in_the_parse(fmt, g)) launch_match(stream, src, d_streams, d_index, count, max_len, d_prev4, d_prevm, d_match, d_pos_off, g, tail, wgc, true, d_sel, sel_pitch);
#define ALZ_SEG(F) if (segmented) { launch_emit_seg_long<F>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, d_seg, seg_len, seg_kmax, d_results, d_aux, g); break; }
#define ALZ_SEGL(F) if (segmented) { launch_emit_seg_long<F>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, d_seg, seg_len, seg_kmax, d_results, d_aux, g); break; }
    const mentry* m = (const mentry*)d_match; u8* side = (u8*)d_side;
    switch (fmt) {
    case ALZ_FMT_LZSS: ALZ_SEG(ALZ_FMT_LZSS) launch_emit_par<ALZ_FMT_LZSS>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_LZ10: ALZ_SEG(ALZ_FMT_LZ10) launch_emit_par<ALZ_FMT_LZ10>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_LZ11: ALZ_SEGL(ALZ_FMT_LZ11) launch_emit_par<ALZ_FMT_LZ11>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_LZ40: ALZ_SEGL(ALZ_FMT_LZ40) launch_emit_par<ALZ_FMT_LZ40>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_YAZ0: ALZ_SEG(ALZ_FMT_YAZ0) launch_emit_par<ALZ_FMT_YAZ0>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_YAY0: ALZ_SEG(ALZ_FMT_YAY0) launch_emit_par<ALZ_FMT_YAY0>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_MIO0: ALZ_SEG(ALZ_FMT_MIO0) launch_emit_par<ALZ_FMT_MIO0>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_CLZ0: ALZ_SEG(ALZ_FMT_CLZ0) launch_emit_par<ALZ_FMT_CLZ0>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_BLZ: ALZ_SEG(ALZ_FMT_BLZ) launch_emit_par<ALZ_FMT_BLZ>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_LZHUDSON: ALZ_SEG(ALZ_FMT_LZHUDSON) launch_emit_par<ALZ_FMT_LZHUDSON>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, side, d_results, d_aux, g); break;
    case ALZ_FMT_SMSR00: launch_emit<ALZ_FMT_SMSR00>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    case ALZ_FMT_PRS_BE: {
        if (segmented) { launch_emit_seg_prs<true>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, d_seg, seg_len, seg_kmax, d_results, d_aux, g); break; }
        if (searches_in_the_parse(fmt, g)) hipLaunchKernelGGL((enc_emit_prs_kernel<true, true>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_results, d_aux, d_prev4, d_prevm, g);
        else hipLaunchKernelGGL((enc_emit_prs_kernel<true, false>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_results, d_aux, d_prev4, d_prevm, g);
        break; }
    case ALZ_FMT_PRS_LE: {
        if (segmented) { launch_emit_seg_prs<false>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, d_seg, seg_len, seg_kmax, d_results, d_aux, g); break; }
        if (searches_in_the_parse(fmt, g)) hipLaunchKernelGGL((enc_emit_prs_kernel<false, true>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_results, d_aux, d_prev4, d_prevm, g);
        else hipLaunchKernelGGL((enc_emit_prs_kernel<false, false>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_results, d_aux, d_prev4, d_prevm, g);
        break; }
    case ALZ_FMT_LZ4_BLOCK: {
        if (segmented) { launch_emit_seg_spec<ALZ_FMT_LZ4_BLOCK>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, d_seg, seg_len, seg_kmax, d_results, d_aux, g); break; }
        if (searches_in_the_parse(fmt, g)) hipLaunchKernelGGL((enc_parse_seq_kernel<ALZ_FMT_LZ4_BLOCK, true>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_prev4, d_prevm, d_results, d_aux, g);
        else hipLaunchKernelGGL((enc_parse_seq_kernel<ALZ_FMT_LZ4_BLOCK, false>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_prev4, d_prevm, d_results, d_aux, g);
        break; }
    case ALZ_FMT_LZO: {
        if (segmented) { launch_emit_seg_spec<ALZ_FMT_LZO>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, d_seg, seg_len, seg_kmax, d_results, d_aux, g); break; }
        if (searches_in_the_parse(fmt, g)) hipLaunchKernelGGL((enc_parse_lzo_kernel<true>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_results, d_aux, d_prev4, d_prevm, g);
        else hipLaunchKernelGGL((enc_parse_lzo_kernel<false>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_results, d_aux, d_prev4, d_prevm, g);
        break; }
    case ALZ_FMT_SNAPPY_RAW: {
        if (segmented) { launch_emit_seg_seq<ALZ_FMT_SNAPPY_RAW>(stream, count, src, dst, d_streams, d_index, (mentry*)d_match, d_pos_off, d_prev4, d_prevm, (u64*)d_mask, d_seg, seg_len, seg_kmax, d_results, d_aux, g); break; }
        if (searches_in_the_parse(fmt, g)) hipLaunchKernelGGL((enc_parse_seq_kernel<ALZ_FMT_SNAPPY_RAW, true>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_prev4, d_prevm, d_results, d_aux, g);
        else hipLaunchKernelGGL((enc_parse_seq_kernel<ALZ_FMT_SNAPPY_RAW, false>), dim3(count), dim3(64), 0, stream, src, dst, d_streams, d_index, count, m, d_pos_off, d_prev4, d_prevm, d_results, d_aux, g);
        break; }
    case ALZ_FMT_FASTLZ: launch_emit<ALZ_FMT_FASTLZ>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    case ALZ_FMT_HIG: launch_emit<ALZ_FMT_HIG>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    case ALZ_FMT_LZSHREK: launch_emit<ALZ_FMT_LZSHREK>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    case ALZ_FMT_WFLZ: launch_emit<ALZ_FMT_WFLZ>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    case ALZ_FMT_WFLZ_BE: launch_emit<ALZ_FMT_WFLZ_BE>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    case ALZ_FMT_REFPACK: launch_emit<ALZ_FMT_REFPACK>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    case ALZ_FMT_LZ02: launch_emit<ALZ_FMT_LZ02>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    case ALZ_FMT_CNS: launch_emit<ALZ_FMT_CNS>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    case ALZ_FMT_CNX2: launch_emit<ALZ_FMT_CNX2>(stream, count, src, dst, d_streams, d_index, m, d_pos_off, d_prev4, d_prevm, side, d_results, d_aux, g, (u64*)d_mask); break;
    default: return hipErrorInvalidValue;
    }
#undef ALZ_SEG
    if (!scan_joined && hipStreamWaitEvent(stream, scan_side->join, 0) != hipSuccess) return hipGetLastError();      // (the scan streams' kernel)
    return hipGetLastError();
}

#include "alz_encode_big.h"
