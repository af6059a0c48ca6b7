// alz_encode_seg_seq.h -- raw Snappy and PRS for batches of FEW buffers (a framed Snappy stream is chunks of 64 KiB): alz_encode_seg.h's arrangement with the
// emitter of enc_parse_seq_kernel.  Included by alz_encode.hip behind that kernel.
//
// A sequence is a match start with the literals since the match before it (Snappy.cs:124-203); a segment owns the sequences whose match STARTS in it.
// Only a segment's FIRST sequence depends on what lies in front of the segment -- its literals begin at the end of the last match before it, however far
// back that is --, so the count pass leaves that one out and hands its start and match to the prefix kernel, which knows every segment's last match end.
// LZ4 blocks and LZO, whose matches have no longest length, have no synchronisation points (one needs every jump that could cross it, and kernel B only measures
// matches up to its compare cap): their walk is the speculative one further down (round 6), their emitter this one.
//   sync + walk as for the flag-bit formats
//   C  enc_seq_seg_kernel<FMT, false>   per segment: bytes of its sequences but the first; the first one's start, distance and length; the end of its last match
//   P  enc_seq_prefix_kernel<FMT>       per buffer: the end of the last match in front of each segment, the first sequences' sizes, the byte offsets
//   E  enc_seq_seg_kernel<FMT, true>    the sequences, with enc_parse_seq_kernel's arithmetic (literals of earlier windows and segments go with the first start behind them)
//   F  enc_seq_finish_kernel<FMT>       the length varint, the literals behind the last match, the result
// LZ4 blocks and LZO (round 6), in place of sync + walk, in front of C:
//   W  enc_spec_walk_kernel<FMT>        per segment: the walk from the segment's first position, as if a cursor stood there -- start mask, cursor mask, exit
//   X  enc_spec_fix_kernel<FMT>         per buffer: the true cursor through every segment until it stands on one of W's cursors; the start mask made the true walk's
//   L  enc_spec_long_kernel<FMT>        per segment: the raw lengths of the true walk's matches of 2 046 bytes and more, behind their entries
//   H  enc_lzo_head_kernel              LZO, per buffer: the head of the stream the reference's way, until a match has been written (LZO.cs:141-250)

template <int FMT, bool EMIT>
__global__ __launch_bounds__(64) void enc_seq_seg_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                         const u32* __restrict__ index_list, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                         const u64* __restrict__ startmask, SegRec* __restrict__ seg, const u32* __restrict__ stot, u32 kpitch, u32 seglen, EncGeom g) {
    typedef SeqFmt<FMT> F;
    const u32 k = blockIdx.x, bid = blockIdx.y;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const u32 n = st.src_len;
    const u32 S = k * seglen;
    if (S >= n) return;
    const u32 E = S + seglen < n ? S + seglen : n;
    const int limit = (int)n - 4;
    const u8* src = src_base + st.src_off;
    u8* dst = dst_base + st.dst_off;
    const u32 cap = st.dst_cap;
    const mentry* m = match + pos_off[sid];
    const u64* mask = startmask + (pos_off[sid] >> 6);
    SegRec* rec = seg + (size_t)bid * kpitch + k;
    u32 cover = EMIT ? rec->tok : 0u;      // end of the last match so far (EMIT: of the segments in front, from the prefix kernel)
    u32 obase = EMIT ? rec->pay : 0u;      // bytes of the sequences before the window
    u32 first_p = 0xFFFFFFFFu, first_d = 0, first_m = 0;
    bool fail = false;
    // LZO: the head of the stream went out the reference's way (enc_lzo_head_kernel); the units are the starts from `mo` on -- or none: the head ended the stream or ran out of room
    u32 mo = 0;
    if (FMT == ALZ_FMT_LZO) {
        mo = stot[4 * (size_t)bid + 2];
        if (stot[4 * (size_t)bid + 3] != 0u) mo = 0xFFFFFFFFu;
    }
    auto ldm = [&](u32 q) { return (int)q <= limit ? m_unpack(m[q]) : make_uint2(0, 0); };
    u64 sm_n = mask[S >> 6];
    uint2 a_n = ldm(S + (u32)lane);
    for (u32 P = S; P < E; P += 64) {
        const u32 p = P + (u32)lane;
        u64 sm = sm_n; const uint2 a = a_n;
        if (P + 64 < E) { sm_n = mask[(P + 64) >> 6]; a_n = ldm(p + 64u); }
        if (FMT == ALZ_FMT_LZO && mo > P) sm = mo - P < 64u ? sm & (~0ull << (mo - P)) : 0ull;
        if (sm == 0ull) continue;
        // ---- the sequences that start in this window (enc_parse_seq_kernel)
        const bool start = ((sm >> lane) & 1ull) != 0ull;
        u32 M = start ? a.y : 0u; const u32 D = a.x;
        if (FMT == ALZ_FMT_LZ4_BLOCK || FMT == ALZ_FMT_LZO) { if (start && M == ALZ_M_LONG) M = m[p + 1]; }      // (a match of 2 046 bytes or more: its length is the next entry)
        const u32 mend = start ? p + M : 0u;
        const u32 pmax = scan_max(mend);
        u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)pmax, 0x138, 0xF, 0xF, false);   // wave_shr:1 -> max over lanes below
        if (before < cover) before = cover;
        const u32 L = start ? p - before : 0u;
        const u32 lh = start ? F::lit_hdr(L) : 0u;
        u32 esz = start ? lh + L + F::match_size(D, M) : 0u;
        const int f0 = (int)__builtin_ctzll(sm);                                           // the first start of the window
        if (!EMIT && first_p == 0xFFFFFFFFu) {                                             // the segment's first sequence: sized by the prefix kernel
            first_p = P + (u32)f0; first_d = (u32)__builtin_amdgcn_readlane((int)D, f0); first_m = (u32)__builtin_amdgcn_readlane((int)M, f0);
            if (lane == f0) esz = 0u;
        }
        const u32 incl = scan_add(esz);
        if (EMIT) {
            const u32 off = obase + incl - esz;
            const bool fits = start && off + esz <= cap;
            if (start && !fits) fail = true;
            if (fits) {
                F::put_lit_hdr(dst + off, L, M, false);
                if constexpr (FMT == ALZ_FMT_LZO) {
                    // the literals behind my match, if there are fewer than four, are counted in my token: the next start lies right behind it (0-3 on), or the data ends there
                    u32 emb = 0u;
                    if (mend + 3u >= n) emb = n - mend <= 3u ? n - mend : 0u;
                    else {
                        u64 w = mask[mend >> 6] >> (mend & 63u);
                        if ((mend & 63u) > 60u) w |= mask[(mend >> 6) + 1u] << (64u - (mend & 63u));
                        w &= 0xFull;
                        emb = w ? (u32)__builtin_ctzll(w) : 0u;
                    }
                    F::put_match(dst + off + lh + L, D, M, emb);
                } else F::put_match(dst + off + lh + L, D, M);
            }
            const u64 above = (lane < 63 ? sm >> (lane + 1) : 0ull);
            const int s = above ? lane + 1 + (int)__builtin_ctzll(above) : lane;          // the next start behind me (my own lane: none)
            const u32 sbef = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)before);
            const u32 sbase = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)(off + lh - before));
            const u32 sfit = (u32)__builtin_amdgcn_ds_bpermute(s << 2, (int)(fits ? 1u : 0u));
            if (above && !start && sfit && p >= sbef && p < n) dst[sbase + p] = src[p];
            const u32 fbef = (u32)__builtin_amdgcn_readlane((int)before, f0);
            if (fbef < P && __builtin_amdgcn_readlane((int)(fits ? 1u : 0u), f0)) {        // literals of earlier windows (and segments)
                const u32 dq = (u32)__builtin_amdgcn_readlane((int)(off + lh), f0);
                wave_copy(dst + dq, src + fbef, P - fbef, lane);
            }
        }
        obase += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u32 wmax = (u32)__builtin_amdgcn_readlane((int)pmax, 63);
        if (wmax > cover) cover = wmax;
    }
    if (!EMIT) {
        if (lane == 0) { SegRec r; r.tok = cover; r.pay = obase; r.unc = first_p; r.head = first_d; r.tailbits = first_m; r.tailofs = 0; r.fail = 0; r.pad = 0; *rec = r; }
        return;
    }
    const bool anyfail = __ballot(fail) != 0ull;
    if (lane == 0) rec->fail = anyfail ? 1u : 0u;
}

// bytes of the decompressed length in front of raw Snappy  Snappy.cs:126-135
__device__ __forceinline__ u32 snappy_varint_size(u32 n) { return n < 0x80u ? 1u : n < 0x4000u ? 2u : n < 0x200000u ? 3u : n < 0x10000000u ? 4u : 5u; }

template <int FMT>
__global__ __launch_bounds__(64) void enc_seq_prefix_kernel(const alz_stream* __restrict__ streams, const u32* __restrict__ index_list, SegRec* __restrict__ seg,
                                                            u32* __restrict__ stot, u32 kpitch, u32 seglen) {
    typedef SeqFmt<FMT> F;
    const u32 bid = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const u32 n = streams[index_list[bid]].src_len;
    const u32 K = (n + seglen - 1u) / seglen;
    SegRec* rec = seg + (size_t)bid * kpitch;
    u32 ccover = 0, cbytes = FMT == ALZ_FMT_SNAPPY_RAW ? snappy_varint_size(n) : 0u;
    if (FMT == ALZ_FMT_LZO) {                                                  // behind the head (enc_lzo_head_kernel): its bytes, the position it reached
        if (stot[4 * (size_t)bid + 3] != 0u) return;
        cbytes = stot[4 * (size_t)bid]; ccover = stot[4 * (size_t)bid + 1];
    }
    for (u32 k0 = 0; k0 < K; k0 += 64) {
        const u32 k = k0 + (u32)lane;
        u32 cov = 0, bytes = 0, fp = 0xFFFFFFFFu, fd = 0, fm = 0;
        if (k < K) { cov = rec[k].tok; bytes = rec[k].pay; fp = rec[k].unc; fd = rec[k].head; fm = rec[k].tailbits; }
        const u32 cincl = scan_max(cov);
        u32 cin = (u32)__builtin_amdgcn_update_dpp(0, (int)cincl, 0x138, 0xF, 0xF, false);    // the segments in front, this round
        if (cin < ccover) cin = ccover;
        if (fp != 0xFFFFFFFFu) { const u32 L = fp - cin; bytes += F::lit_hdr(L) + L + F::match_size(fd, fm); }
        const u32 bincl = scan_add(bytes);
        if (k < K) { rec[k].tok = cin; rec[k].pay = cbytes + bincl - bytes; }
        const u32 cm = (u32)__builtin_amdgcn_readlane((int)cincl, 63);
        if (cm > ccover) ccover = cm;
        cbytes += (u32)__builtin_amdgcn_readlane((int)bincl, 63);
    }
    if (lane == 0) { stot[4 * (size_t)bid] = cbytes; stot[4 * (size_t)bid + 1] = ccover; if (FMT != ALZ_FMT_LZO) { stot[4 * (size_t)bid + 2] = 0; stot[4 * (size_t)bid + 3] = 0; } }
}

template <int FMT>
__global__ __launch_bounds__(64) void enc_seq_finish_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                            const u32* __restrict__ index_list, const SegRec* __restrict__ seg, const u32* __restrict__ stot,
                                                            u32 kpitch, u32 seglen, alz_result* __restrict__ results, alz_encode_aux* __restrict__ aux) {
    typedef SeqFmt<FMT> F;
    const u32 bid = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const u32 n = st.src_len, cap = st.dst_cap;
    const u8* src = src_base + st.src_off;
    u8* dst = dst_base + st.dst_off;
    if (FMT == ALZ_FMT_LZ4_BLOCK && n < 5u) {                                 // source.Slice(0, Length - 5) throws  (LZ4.cs:208)
        if (lane == 0) { alz_result r; r.dst_len = 0; r.src_used = n; r.status = ALZ_ST_BAD_TOKEN; r.reserved = 0; results[sid] = r; if (aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; } }
        return;
    }
    const u32 K = (n + seglen - 1u) / seglen;
    const SegRec* rec = seg + (size_t)bid * kpitch;
    bool fail = false;
    if (FMT == ALZ_FMT_LZO) {
        const u32 hs = stot[4 * (size_t)bid + 3];
        if (hs == 1u) return;                                                  // the head wrote the whole stream (or refused it) and its result
        if (hs == 2u) {                                                        // the head ran out of room
            if (lane == 0) { alz_result r; r.dst_len = 0; r.src_used = n; r.status = ALZ_ST_OUTPUT_CAPACITY; r.reserved = 0; results[sid] = r; if (aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; } }
            return;
        }
    }
    for (u32 k = (u32)lane; k < K; k += 64) if (rec[k].fail) fail = true;
    const u32 obase = stot[4 * (size_t)bid], cover = stot[4 * (size_t)bid + 1];
    if constexpr (FMT == ALZ_FMT_LZO) {
        // behind the last match: 0-3 literals are counted in its token (enc_seq_seg_kernel's emb) and follow it bare; four and more are a run of their own; then the end token
        const u32 rest = n - cover;
        const u32 lsz = rest >= 4u ? lzo_lit_size(rest) : 0u;
        const u32 total = obase + lsz + rest + 3u;
        if (total > cap) fail = true;
        const bool anyfail = __ballot(fail) != 0ull;
        if (!anyfail) {
            if (rest >= 4u && lane == 0) (void)lzo_put_lit(dst + obase, rest);
            wave_copy(dst + obase + lsz, src + cover, rest, lane);
            if (lane == 0) { dst[total - 3u] = 0x11; dst[total - 2u] = 0; dst[total - 1u] = 0; }
        }
        if (lane == 0) {
            alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
            results[sid] = r;
            if (aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
        }
        return;
    }
    if (FMT == ALZ_FMT_SNAPPY_RAW) {
        const u32 kv = snappy_varint_size(n);
        if (kv <= cap) { if (lane == 0) { u32 v = n, q = 0; while (v >= 0x80u) { dst[q++] = (u8)((v | 0x80u) & 0xFFu); v >>= 7; } dst[q] = (u8)v; } } else fail = true;
    }
    // the end: the remaining literals (LZ4: at least five, always a sequence; Snappy: an element only if there are any)
    const bool lastseq = FMT == ALZ_FMT_LZ4_BLOCK || n - cover != 0u;
    const u32 plain = n - cover, lh = lastseq ? F::lit_hdr(plain) : 0u;
    const u32 total = obase + lh + plain;
    if (total > cap) fail = true;
    const bool anyfail = __ballot(fail) != 0ull;
    if (!anyfail && lastseq) {
        if (lane == 0) F::put_lit_hdr(dst + obase, plain, 4u, true);
        wave_copy(dst + obase + lh, src + cover, plain, lane);
    }
    if (lane == 0) {
        alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
        results[sid] = r;
        if (aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
    }
}

// ---------------------------------------------------------------------------------------------- LZ4 blocks over segments (round 6)
// An LZ4 match has no longest length, so no bounded look-back proves a synchronisation point, and kernel B's capped entries poison a buffer-wide one (alz_encode_seg.h).  The walk
// is made parallel another way: every segment is walked SPECULATIVELY from its first position (enc_spec_walk_kernel, one wavefront per segment, over kernel B's match entries -- a
// capped one searched exactly where a cursor stands on it, as everywhere), and one wavefront per buffer then strings the true walk together (enc_spec_fix_kernel): the cursor
// that really enters a segment walks on until it stands on a cursor of the speculative walk -- from there on the two are the same walk, because what FindNextBestMatch does at
// a cursor depends on that cursor alone (:157-212; MatchSearch is a pure function of the data) -- which on real data is a few tokens.  What the speculative walk found in front
// of that point is dropped, what the true cursor found is put in its place.  A token is owned by the segment of its CURSOR: one that a lazy step puts on the first position of
// the next segment is kept in the segment's record and entered by the fix-up when it is part of the true walk.  A speculative walk may write the exact match of a capped position
// into the array (any walk would write the same), but never the raw length of a 2 046-byte-or-longer match into the NEXT entry -- the true walk might stand there: those
// lengths are entered last (enc_spec_long_kernel), for the starts of the true walk only.  Then alz_encode_seg_seq.h's emitter over the start mask.
struct SpecRec { u32 exit, carried, pad0, pad1; };      // the first cursor behind the segment; carried: a lazy step left a token on the next segment's first position
#define ALZ_SPEC_MAXW 128u                              /* mask words of a segment (8 192 positions) */

// The walk from `cur` to the end of the segment [S, End) (or to the first cursor that is set in `stop`: the speculative walk's cursor mask), window by window: start bits into
// smw, cursor bits into cmw (both LDS, relative to S).  Returns the cursor it ended on; met: it ended on a cursor of `stop`; carried: its last step left a token on position End.
template <bool MINT>
__device__ __forceinline__ int spec_walk(const u8* data, int ns, int limit, const EncGeom& g, mentry* m, const int* p4, const int* pm, int S, int End, int cur, const u64* stop,
                                         u64* smw, u64* cmw, int lane, bool& met, bool& carried, bool has0 = false, u64 stop0 = 0ull) {
    // (has0 / stop0: the caller holds the first word of `stop` already -- the fix-up, which has every segment's in LDS)
    met = false; carried = false;
    auto ldm = [&](int q) { return q <= limit ? m_unpack(m[q]) : make_uint2(0, 0); };
    while (cur <= limit && cur < End) {
        const int P = cur & ~63, p = P + lane;
        const uint2 a = ldm(p), b = ldm(p + 1);
        // (capped by kernel B, or searched exactly by some walk that found 2 046 bytes or more -- the entry holds the code, not the length: searched again)
        const bool capped = a.y >= ALZ_M_LONG || b.y >= ALZ_M_LONG;
        int jump = 1, startrel = 0;   // startrel: 0 no token here, 1 match starts here, 2 literal here + match at p + 1
        if (!capped && p <= limit && (int)a.y >= g.min_len) {
            const int l0 = (int)a.y, l1 = (int)b.y;
            const bool lazyc = l0 <= g.lazy && p + 1 <= limit;
            if (lazyc && l1 > l0) { startrel = 2; const int e = p + 1 + l1; const int stop2 = e < limit + 1 ? e : limit + 1; jump = (p + 2 > stop2 ? p + 2 : stop2) - p; }
            else { startrel = 1; const int skip = lazyc ? 1 : 0; const int e = p + l0; const int stop2 = e < limit + 1 ? e : limit + 1; jump = (p + 1 + skip > stop2 ? p + 1 + skip : stop2) - p; }
        }
        const u64 stopw = !stop ? 0ull : (has0 && P == S) ? stop0 : stop[(u32)(P - S) >> 6];
        u64 sb = 0, cb = 0; bool nextbit = false;
        int rel = cur - P;
        if (__ballot(capped) == 0ull) {
            // no capped entry in the window: the hops as five scalar instructions each, the visited lanes collected as a bit mask and the token starts read off two
            // ballots (enc_roles_kernel's loop -- written in C++ a hop is ~40 instructions of the CU's one scalar unit, which seven wavefronts per SIMD share: the walk
            // kernel of 256 x 64 KiB at quality 0 took 0.23 ms of the call's 0.64)
            u64 M = 0; u32 r = (u32)rel, j;
            u32 lim = (u32)(limit + 1 - P) < 64u ? (u32)(limit + 1 - P) : 64u;     // (r < lim on entry: cur <= limit and cur < End)
            if ((u32)(End - P) < lim) lim = (u32)(End - P);
            lim = (u32)__builtin_amdgcn_readfirstlane((int)lim); r = (u32)__builtin_amdgcn_readfirstlane((int)r);      // (wave-uniform by construction: into scalar registers)
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
            if (M & stopw) {                                                       // the first cursor the speculative walk stood on as well: the walks are one from here
                const u32 b0 = (u32)__builtin_ctzll(M & stopw);
                met = true; M &= (1ull << b0) - 1ull; r = b0;
            }
            const u64 s1 = __ballot(startrel == 1) & M, s2 = __ballot(startrel == 2) & M;
            sb = s1 | (s2 << 1); nextbit = (s2 >> 63) != 0ull; cb = M;
            rel = (int)r;
        }
        else while (rel < 64 && P + rel <= limit && P + rel < End) {
            if ((stopw >> rel) & 1ull) { met = true; break; }
            cb |= 1ull << rel;
            int j, sr;
            if (__builtin_amdgcn_readlane((int)capped, rel)) {
                const int q = P + rel;
                int d0, l0, d1 = 0, l1 = 0; bool s0, s1;
                const uint2 e0 = make_uint2((u32)__builtin_amdgcn_readlane((int)a.x, rel), (u32)__builtin_amdgcn_readlane((int)a.y, rel));
                const uint2 e1 = make_uint2((u32)__builtin_amdgcn_readlane((int)b.x, rel), (u32)__builtin_amdgcn_readlane((int)b.y, rel));
                benc_capped_cursor<MINT>(data, ns, g, p4, pm, q, limit, e0, e0.y >= ALZ_M_LONG, e1, e1.y >= ALZ_M_LONG, d0, l0, d1, l1, s0, s1);
                j = 1; sr = 0;
                if (l0 >= g.min_len) {
                    const bool lazyc = l0 <= g.lazy && q + 1 <= limit;
                    if (lazyc && l1 > l0) { sr = 2; const int e = q + 1 + l1; const int stop2 = e < limit + 1 ? e : limit + 1; j = (q + 2 > stop2 ? q + 2 : stop2) - q; }
                    else { sr = 1; const int skip = lazyc ? 1 : 0; const int e = q + l0; const int stop2 = e < limit + 1 ? e : limit + 1; j = (q + 1 + skip > stop2 ? q + 1 + skip : stop2) - q; }
                }
                // the exact matches go into the array -- what MatchSearch returns for q and q + 1, whoever asks -- a length of 2 046 or more as its code only
                if (lane == 0) {
                    if (s0) m[q] = m_pack((u32)d0, l0 < (int)ALZ_M_LONG ? (u32)l0 : ALZ_M_LONG);
                    if (s1) m[q + 1] = m_pack((u32)d1, l1 < (int)ALZ_M_LONG ? (u32)l1 : ALZ_M_LONG);
                }
            } else { j = __builtin_amdgcn_readlane(jump, rel); sr = __builtin_amdgcn_readlane(startrel, rel); }
            if (sr == 1) sb |= 1ull << rel;
            else if (sr == 2) { if (rel + 1 < 64) sb |= 1ull << (rel + 1); else nextbit = true; }
            rel += j;
        }
        // (a token on position End -- the first of the next segment -- is the caller's to enter; one inside the segment goes into the mask)
        if (P + 64 >= End) { if (nextbit) carried = true; }
        else if (nextbit && lane == 0) smw[((u32)(P - S) >> 6) + 1u] |= 1ull;
        if (lane == 0) { smw[(u32)(P - S) >> 6] |= sb; cmw[(u32)(P - S) >> 6] |= cb; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        cur = P + rel;
        if (met) break;
    }
    return cur;
}

template <int FMT>
__global__ __launch_bounds__(64) void enc_spec_walk_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                           mentry* __restrict__ match, const u64* __restrict__ pos_off, const int* __restrict__ prev4, const int* __restrict__ prevm,
                                                           u64* __restrict__ startmask, u32* __restrict__ spec, u32 kpitch, u32 seglen, u32 recw, EncGeom g) {
    __shared__ u64 smw[ALZ_SPEC_MAXW], cmw[ALZ_SPEC_MAXW];
    const u32 k = blockIdx.x, bid = blockIdx.y;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const int n = (int)st.src_len, ns = n - (FMT == ALZ_FMT_LZ4_BLOCK ? 5 : 0), limit = ns - 4;
    const int S = (int)(k * seglen), End = S + (int)seglen;
    if (S >= n) return;
    const u32 nw = seglen >> 6;
    u32* rec = spec + ((size_t)bid * kpitch + k) * recw;
    u64* cm_out = reinterpret_cast<u64*>(rec + 4);
    u64* mask = startmask + (pos_off[sid] >> 6) + ((u32)S >> 6);
    for (u32 w = (u32)lane; w < nw; w += 64) { smw[w] = 0ull; cmw[w] = 0ull; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    const u8* data = src_base + st.src_off;
    mentry* m = match + pos_off[sid];
    const int* p4 = prev4 + pos_off[sid];
    const int* pm = g.use_min_table ? prevm + pos_off[sid] : nullptr;
    bool met, carried;
    const int x = g.use_min_table ? spec_walk<true>(data, ns, limit, g, m, p4, pm, S, End, S, nullptr, smw, cmw, lane, met, carried)
                                  : spec_walk<false>(data, ns, limit, g, m, p4, pm, S, End, S, nullptr, smw, cmw, lane, met, carried);
    for (u32 w = (u32)lane; w < nw; w += 64) { if ((u32)S + 64u * w < (u32)n) mask[w] = smw[w]; cm_out[w] = cmw[w]; }
    if (lane == 0) { rec[0] = (u32)x; rec[1] = carried ? 1u : 0u; rec[2] = 0; rec[3] = 0; }
}

template <int FMT>
__global__ __launch_bounds__(64) void enc_spec_fix_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                          mentry* __restrict__ match, const u64* __restrict__ pos_off, const int* __restrict__ prev4, const int* __restrict__ prevm,
                                                          u64* __restrict__ startmask, const u32* __restrict__ spec, u32 kpitch, u32 seglen, u32 recw, EncGeom g) {
    __shared__ u64 nbw[ALZ_SPEC_MAXW], ncw[ALZ_SPEC_MAXW];
    // Every step of this loop is latency -- one wavefront, one segment after the other --, so what a step needs from the speculative walk's records is in LDS before the first one:
    // exit, carried and the first words of the cursor mask and of the start mask (the true cursor nearly always enters a segment in its first window) of up to ALZ_SPEC_PRE segments.
    // A step was ~450 instructions on this one wavefront, ~2 us -- 45 steps for a 64 KiB buffer, a quarter of a call of 16 buffers.  (The match entries of the windows the segments
    // will most likely be entered in, fetched 96 segments at a time into LDS as well: measured, nothing -- the steps that walk are bound by their instructions, not by that load.)
    constexpr u32 ALZ_SPEC_PRE = 1024u;
    __shared__ u32 exL[ALZ_SPEC_PRE], caL[ALZ_SPEC_PRE];
    __shared__ u64 c0L[ALZ_SPEC_PRE], m0L[ALZ_SPEC_PRE];                           // (m0L: the first word of the segment's start mask, as the speculative walk left it)
    const u32 bid = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const int n = (int)st.src_len, ns = n - (FMT == ALZ_FMT_LZ4_BLOCK ? 5 : 0), limit = ns - 4;
    if (limit < 0) return;
    const u32 nw = seglen >> 6;
    const u32 K = ((u32)n + seglen - 1u) / seglen;
    u64* mask0 = startmask + (pos_off[sid] >> 6);
    for (u32 k = (u32)lane; k < K && k < ALZ_SPEC_PRE; k += 64) {
        const u32* r = spec + ((size_t)bid * kpitch + k) * recw;
        if ((int)(k * seglen) <= limit) { exL[k] = r[0]; caL[k] = r[1]; c0L[k] = *reinterpret_cast<const u64*>(r + 4); m0L[k] = mask0[(k * seglen) >> 6]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    const u8* data = src_base + st.src_off;
    mentry* m = match + pos_off[sid];
    const int* p4 = prev4 + pos_off[sid];
    const int* pm = g.use_min_table ? prevm + pos_off[sid] : nullptr;
    int e = 0;                                                                   // the cursor that enters the segment
    bool carry_in = false;                                                       // the walk in front left a token on this segment's first position
    for (u32 k = 0; k < K; k++) {
        const int S = (int)(k * seglen), End = S + (int)seglen;
        if (S > limit) break;
        // Three steps in four, the cursor that enters the segment is one the speculative walk stood on as well, in the segment's first window: nothing to walk -- the speculative
        // walk's starts below it go, a token the segment in front put on the first position comes in (a store, nothing read: ~30 instructions instead of ~450 on this one wavefront)
        if (k < ALZ_SPEC_PRE && e >= S && e - S < 64 && ((c0L[k] >> (u32)(e - S)) & 1ull)) {
            if ((e > S || carry_in) && lane == 0) mask0[(u32)S >> 6] = (m0L[k] & (~0ull << (u32)(e - S))) | (carry_in ? 1ull : 0ull);
            carry_in = caL[k] != 0u;
            e = (int)exL[k];
            continue;
        }
        const u32* rec = spec + ((size_t)bid * kpitch + k) * recw;
        const u64* cm = reinterpret_cast<const u64*>(rec + 4);
        u64* mask = mask0 + ((u32)S >> 6);
        int x = e, c = End;                                                      // c: what the speculative walk found below it is not part of the true walk
        bool carried = false;
        for (u32 w = (u32)lane; w < nw; w += 64) { nbw[w] = 0ull; ncw[w] = 0ull; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (e < End) {
            bool met;
            const bool pre = k < ALZ_SPEC_PRE;
            const u64 stop0 = pre ? c0L[k] : 0ull;
            const int cur = g.use_min_table ? spec_walk<true>(data, ns, limit, g, m, p4, pm, S, End, e, cm, nbw, ncw, lane, met, carried, pre, stop0)
                                            : spec_walk<false>(data, ns, limit, g, m, p4, pm, S, End, e, cm, nbw, ncw, lane, met, carried, pre, stop0);
            if (met) { c = cur; x = (int)(pre ? exL[k] : uni(rec[0])); carried = (pre ? caL[k] : uni(rec[1])) != 0u; }
            else { c = End; x = cur; }
        }
        // the segment's start mask: the speculative walk's starts from c on, the true cursor's in front of it, the token carried in from the segment before.  Only the word
        // c lies in is read; the words in front of it are the true cursor's alone, the words behind it stay as they are (but for a token this walk put on c itself or carried in)
        for (u32 w = (u32)lane; w < nw; w += 64) {
            if ((u32)S + 64u * w >= (u32)n) continue;
            const int lo = S + 64 * (int)w;
            const u64 mine = nbw[w] | ((w == 0u && carry_in) ? 1ull : 0ull);
            if (c >= lo + 64) mask[w] = mine;
            else if (c > lo) mask[w] = (mask[w] & (~0ull << (u32)(c - lo))) | mine;
            else if (mine) mask[w] |= mine;
        }
        carry_in = carried;
        e = x;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}

// the raw lengths of the true walk's matches of 2 046 bytes or more, behind their entries (the emitters read them there): MatchSearch once more for each of them
template <int FMT>
__global__ __launch_bounds__(64) void enc_spec_long_kernel(const u8* __restrict__ src_base, const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                           mentry* __restrict__ match, const u64* __restrict__ pos_off, const int* __restrict__ prev4, const int* __restrict__ prevm,
                                                           const u64* __restrict__ startmask, u32 seglen, EncGeom g) {
    const u32 k = blockIdx.x, bid = blockIdx.y;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const int n = (int)st.src_len, ns = n - (FMT == ALZ_FMT_LZ4_BLOCK ? 5 : 0), limit = ns - 4;
    const int S = (int)(k * seglen);
    if (S > limit) return;
    const int End = S + (int)seglen < limit + 1 ? S + (int)seglen : limit + 1;
    const u8* data = src_base + st.src_off;
    mentry* m = match + pos_off[sid];
    const int* p4 = prev4 + pos_off[sid];
    const int* pm = g.use_min_table ? prevm + pos_off[sid] : nullptr;
    const u64* mask = startmask + (pos_off[sid] >> 6);
    for (int P = S; P < End; P += 64) {
        const u64 sm = mask[(u32)P >> 6];
        if (sm == 0ull) continue;
        const int p = P + lane;
        const bool lg = ((sm >> lane) & 1ull) && p <= limit && (m[p] >> ALZ_M_DBITS) == ALZ_M_LONG;
        u64 todo = __ballot(lg);
        while (todo) {
            const int q = P + (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            int d0, l0;
            if (g.use_min_table) benc_wave_search<true>(data, ns, g, p4, pm, q, d0, l0); else benc_wave_search<false>(data, ns, g, p4, pm, q, d0, l0);
            if (lane == 0) m[q + 1] = (u32)l0;
        }
    }
}

// LZO: the head of a stream, the reference's way (LZO.cs:141-250, as enc_parse_lzo_kernel's head -- the starts from the mask instead of a walk): a match that 1-3 literals precede
// is cut at its front so that four go out, one cut below three bytes is not written; once a match HAS been written what follows every match is 0-3 literals or a run of four and
// more, and the rest of the stream is units (SeqFmt<ALZ_FMT_LZO>).  stot[4 bid ..]: the head's bytes, the position it reached, the first start that is a unit, and
// 0 units follow / 1 the head ended the stream and wrote its result / 2 it ran out of room.
__global__ __launch_bounds__(64) void enc_lzo_head_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                          const u32* __restrict__ index_list, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                          const u64* __restrict__ startmask, u32* __restrict__ stot, alz_result* __restrict__ results, alz_encode_aux* __restrict__ aux) {
    const u32 bid = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u32 n = st.src_len;
    u8* dst = dst_base + st.dst_off;
    const u32 cap = st.dst_cap;
    const mentry* m = match + pos_off[sid];
    const u64* mask = startmask + (pos_off[sid] >> 6);
    u32* tot = stot + 4 * (size_t)bid;
    auto finish = [&](u32 total, bool fail, int status) {
        if (lane == 0) {
            alz_result r; r.dst_len = (fail || status != ALZ_ST_OK) ? 0u : total; r.src_used = n;
            r.status = status != ALZ_ST_OK ? status : (fail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK); r.reserved = 0;
            results[sid] = r;
            if (aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
            tot[0] = 0; tot[1] = 0; tot[2] = 0; tot[3] = 1u;
        }
    };
    u32 sp = 0, olen = 0; bool fail = false; int status = ALZ_ST_OK;
    auto put = [&](u32 b) { if (olen < cap) { if (lane == 0) dst[olen] = (u8)b; } else fail = true; olen++; };
    auto copy = [&](u32 from, u32 len) { for (u32 i = 0; i < len; i++) put(src[from + i]); };
    if (n < 0x10u) {
        put(17u + n); copy(0, n); put(0x11); put(0); put(0);
        finish(olen, fail, status);
        return;
    }
    const u32 nwords = (n + 63u) >> 6;
    // the next start at or behind `from` (n: none), and its match: 64 mask words per round
    auto next_start = [&](u32 from, u32& d, u32& l) -> u32 {
        const u32 w0 = from >> 6;
        for (u32 wb = w0; wb < nwords; wb += 64u) {
            const u32 wi = wb + (u32)lane;
            u64 w = wi < nwords ? mask[wi] : 0ull;
            if (wi == w0) w &= ~0ull << (from & 63u);
            const u64 nz = __ballot(w != 0ull);
            if (nz) {
                const int l0 = (int)__builtin_ctzll(nz);
                const u64 ww = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(w >> 32), l0) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)w, l0);
                const u32 pos = ((wb + (u32)l0) << 6) + (u32)__builtin_ctzll(ww);
                const mentry e = m[pos];
                d = e & ALZ_M_DMASK; l = e >> ALZ_M_DBITS;
                if (l == ALZ_M_LONG) l = m[pos + 1u];
                return pos;
            }
        }
        d = 0; l = 0; return n;
    };
    u32 ml = 0, md = 0;
    u32 mo = next_start(0, md, ml);
    u32 mbit = mo;                                                            // the match's bit in the mask (mo itself may be moved below)
    bool clean = false;
    while (sp != n && !clean) {
        u32 plain = mo - sp;
        if (plain != 0u) {
            if (plain < 4u) { const u32 dif = 4u - plain; mo += dif; ml = ml > dif ? ml - dif : 0u; plain = 4u; }
            if (plain > 18u) { put(0); u32 v = plain - 18u; while (v > 255u) { put(0); v -= 255u; } put(v); } else put(plain - 3u);
            if (sp + plain > n) { status = ALZ_ST_BAD_TOKEN; break; }
            if (plain > 64u && !fail && olen + plain <= cap) { wave_copy(dst + olen, src + sp, plain, lane); olen += plain; } else copy(sp, plain);
            sp += plain;
        }
        u32 nl = 0, nd = 0;
        const u32 no = mbit < n ? next_start(mbit + 1u, nd, nl) : n;
        if (ml >= 3u) {
            sp += ml;
            u32 emb = no - sp;
            if (no < sp) { status = ALZ_ST_BAD_TOKEN; break; }
            if (emb > 3u) emb = 0;
            if (lzo_match_size(md, ml) <= 8u) { u8 tok[8]; const u32 kk = lzo_put_match(tok, md, ml, emb); for (u32 i = 0; i < kk; i++) put(tok[i]); }
            else if (md <= 16384u) {
                put(0x20); u32 v = ml - 33u; while (v > 255u) { put(0); v -= 255u; } put(v);
                put((emb | ((md - 1u) << 2)) & 0xFFu); put(((md - 1u) >> 6) & 0xFFu);
            } else {
                const u32 d2 = md - 0x4000u, flag = (0x10u | ((d2 & 0x4000u) >> 11)) & 0xFFu;
                put(flag); u32 v = ml - 9u; while (v > 255u) { put(0); v -= 255u; } put(v);
                put((emb | (d2 << 2)) & 0xFFu); put((d2 >> 6) & 0xFFu);
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
    if (lane == 0) { tot[0] = olen; tot[1] = sp; tot[2] = mo; tot[3] = fail ? 2u : 0u; }
}

template <int FMT>
static void launch_emit_seg_spec(hipStream_t s, u32 count, const u8* src, u8* dst, const alz_stream* streams, const u32* index, mentry* match, const u64* pos_off,
                                 const int* prev4, const int* prevm, u64* mask, void* d_seg, u32 seglen, u32 kmax, alz_result* results, alz_encode_aux* aux, const EncGeom& g) {
    SegRec* seg = (SegRec*)d_seg;
    u32* stot = (u32*)((u8*)d_seg + (size_t)count * kmax * sizeof(SegRec));
    u32* spec = stot + 4 * (size_t)count;
    const u32 recw = 4u + (seglen >> 5);                                         // a record: SpecRec + the segment's cursor mask
    hipLaunchKernelGGL((enc_spec_walk_kernel<FMT>), dim3(kmax, count), dim3(64), 0, s, src, streams, index, match, pos_off, prev4, prevm, mask, spec, kmax, seglen, recw, g);
    hipLaunchKernelGGL((enc_spec_fix_kernel<FMT>), dim3(count), dim3(64), 0, s, src, streams, index, match, pos_off, prev4, prevm, mask, (const u32*)spec, kmax, seglen, recw, g);
    hipLaunchKernelGGL((enc_spec_long_kernel<FMT>), dim3(kmax, count), dim3(64), 0, s, src, streams, index, match, pos_off, prev4, prevm, (const u64*)mask, seglen, g);
    if (FMT == ALZ_FMT_LZO) hipLaunchKernelGGL(enc_lzo_head_kernel, dim3(count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, stot, results, aux);
    hipLaunchKernelGGL((enc_seq_seg_kernel<FMT, false>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, (const u32*)stot, kmax, seglen, g);
    hipLaunchKernelGGL((enc_seq_prefix_kernel<FMT>), dim3(count), dim3(64), 0, s, streams, index, seg, stot, kmax, seglen);
    hipLaunchKernelGGL((enc_seq_seg_kernel<FMT, true>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, (const u32*)stot, kmax, seglen, g);
    hipLaunchKernelGGL((enc_seq_finish_kernel<FMT>), dim3(count), dim3(64), 0, s, src, dst, streams, index, (const SegRec*)seg, (const u32*)stot, kmax, seglen, results, aux);
}

// LZ11 / LZ40 (flag-bit formats whose matches reach 16 KiB and more): the speculative walk in front of alz_encode_seg.h's token / flag emitters -- or, with -DALZ_SPEC_LONG11=0, that
// file's synchronisation points with their 16 KiB look-back
template <int FMT>
static void launch_emit_seg_long(hipStream_t s, u32 count, const u8* src, u8* dst, const alz_stream* streams, const u32* index, mentry* match, const u64* pos_off,
                                 const int* prev4, const int* prevm, u64* mask, void* d_seg, u32 seglen, u32 kmax, alz_result* results, alz_encode_aux* aux, const EncGeom& g) {
    if (!seg_spec_format(FMT)) { launch_emit_seg<FMT>(s, count, src, dst, streams, index, match, pos_off, prev4, prevm, mask, d_seg, seglen, kmax, results, aux, g); return; }
    SegRec* seg = (SegRec*)d_seg;
    u32* stot = (u32*)((u8*)d_seg + (size_t)count * kmax * sizeof(SegRec));
    u32* spec = stot + 4 * (size_t)count;
    const u32 recw = 4u + (seglen >> 5);
    hipLaunchKernelGGL((enc_spec_walk_kernel<FMT>), dim3(kmax, count), dim3(64), 0, s, src, streams, index, match, pos_off, prev4, prevm, mask, spec, kmax, seglen, recw, g);
    hipLaunchKernelGGL((enc_spec_fix_kernel<FMT>), dim3(count), dim3(64), 0, s, src, streams, index, match, pos_off, prev4, prevm, mask, (const u32*)spec, kmax, seglen, recw, g);
    hipLaunchKernelGGL((enc_spec_long_kernel<FMT>), dim3(kmax, count), dim3(64), 0, s, src, streams, index, match, pos_off, prev4, prevm, (const u64*)mask, seglen, g);
    hipLaunchKernelGGL((enc_seg_kernel<FMT, false>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, (const u32*)stot, kmax, seglen, g);
    hipLaunchKernelGGL(enc_seg_prefix_kernel, dim3(count), dim3(64), 0, s, streams, index, seg, stot, kmax, seglen);
    hipLaunchKernelGGL((enc_seg_kernel<FMT, true>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, (const u32*)stot, kmax, seglen, g);
    hipLaunchKernelGGL((enc_seg_flags_kernel<FMT>), dim3(count), dim3(64), 0, s, dst, streams, index, (const SegRec*)seg, (const u32*)stot, kmax, seglen, results, aux);
}

template <int FMT>
static void launch_emit_seg_seq(hipStream_t s, u32 count, const u8* src, u8* dst, const alz_stream* streams, const u32* index, mentry* match, const u64* pos_off,
                                const int* prev4, const int* prevm, u64* mask, void* d_seg, u32 seglen, u32 kmax, alz_result* results, alz_encode_aux* aux, const EncGeom& g) {
    SegRec* seg = (SegRec*)d_seg;
    u32* stot = (u32*)((u8*)d_seg + (size_t)count * kmax * sizeof(SegRec));
    u32* sync = stot + 4 * (size_t)count;
    launch_seg_walk(s, count, src, streams, index, match, pos_off, prev4, prevm, mask, sync, seglen, kmax, g);
    hipLaunchKernelGGL((enc_seq_seg_kernel<FMT, false>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, (const u32*)stot, kmax, seglen, g);
    hipLaunchKernelGGL((enc_seq_prefix_kernel<FMT>), dim3(count), dim3(64), 0, s, streams, index, seg, stot, kmax, seglen);
    hipLaunchKernelGGL((enc_seq_seg_kernel<FMT, true>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, (const u32*)stot, kmax, seglen, g);
    hipLaunchKernelGGL((enc_seq_finish_kernel<FMT>), dim3(count), dim3(64), 0, s, src, dst, streams, index, (const SegRec*)seg, (const u32*)stot, kmax, seglen, results, aux);
}

// ---------------------------------------------------------------------------------------------- PRS over segments
// enc_emit_prs_kernel's arithmetic per segment.  A token's place follows from the flag bits and payload bytes in front of it and from `lastk`, the flag byte
// the payload before it waited for: three numbers per segment boundary (+ the end of the last match written as a match, at most 256 bytes back).  A flag
// byte is "opened" by the first payload that waits for it (which fixes its address) and stored by the token that owns its last bit; one that is open across
// a segment boundary -- some of its bits on either side, or opened by a short match whose four bits ended exactly in front of it -- is put together by
// enc_prs_flags_kernel from the address the opening side recorded and the bits of both.
//   C  enc_prs_seg_kernel<BIG, false>: flag bits, payload bytes, the last token's B (relative);  P  enc_prs_prefix_kernel;  E  enc_prs_seg_kernel<BIG, true>;
//   F  enc_prs_flags_kernel<BIG>.  The end token (bit 0, two zero bytes, bit 1; PRS.cs:150-157) belongs to the last segment.
template <bool BIG, bool EMIT>
__global__ __launch_bounds__(64) void enc_prs_seg_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                         const u32* __restrict__ index_list, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                         const u64* __restrict__ startmask, SegRec* __restrict__ seg, u32 kpitch, u32 seglen, EncGeom g) {
    __shared__ u32 flagacc[64];
    __shared__ u32 gofs[64];
    const u32 k = blockIdx.x, bid = blockIdx.y;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const u32 n = st.src_len;
    const u32 S = k * seglen;
    if (S >= n && !(k == 0u && n == 0u)) return;                        // (an empty buffer still has its end token)
    const u32 E = S + seglen < n ? S + seglen : n;
    const bool last_seg = E == n;
    const int limit = (int)n - 4;
    const u8* src = src_base + st.src_off;
    u8* dst = dst_base + st.dst_off;
    const u32 cap = st.dst_cap;
    const mentry* m = match + pos_off[sid];
    const u64* mask = startmask + (pos_off[sid] >> 6);
    SegRec* rec = seg + (size_t)bid * kpitch + k;
    flagacc[lane] = 0; gofs[lane] = 0;
    __syncthreads();
    // the end of the last match WRITTEN AS A MATCH that starts in front of the segment (a match of two bytes further than 0x100 back goes out as literals)
    u32 cover = 0;
    if (k) {
        const int w = (int)(S >> 6) - 1 - lane;
        u32 endv = 0;
        if (w >= 0 && (u32)lane * 64u < (u32)g.max_len + 64u) {
            u64 mw = mask[w];
            while (mw) {
                const u32 hb = 63u - (u32)__builtin_clzll(mw);
                const u32 q = (u32)w * 64u + hb;
                const uint2 e = m_unpack(m[q]);
                if (!(e.y == 2u && e.x > 0x100u)) { endv = q + e.y; break; }
                mw &= ~(1ull << hb);
            }
        }
        cover = (u32)__builtin_amdgcn_readlane((int)scan_max(endv), 63);
    }
    u32 bit_base = EMIT ? rec->tok : 0u, pay_base = EMIT ? rec->pay : 0u;
    u32 lastk = EMIT ? rec->unc : 0xFFFFFFFFu;
    const u32 k0 = bit_base >> 3;
    const bool foreign = EMIT && ((bit_base & 7u) != 0u || lastk == k0);   // flag byte k0 was opened in front of this segment
    u32 last_bp = 0xFFFFFFFFu;                                              // !EMIT: B of the segment's last token, relative to the segment's first bit
    bool fail = false;
    auto ldm = [&](u32 q) { return (int)q <= limit ? m_unpack(m[q]) : make_uint2(0, 0); };
    u64 sm_n = n ? mask[S >> 6] : 0ull;
    uint2 a_n = ldm(S + (u32)lane);
    u32 sb_n = S + (u32)lane < n ? src[S + (u32)lane] : 0u;
    const u32 trips_end = last_seg ? ((n + 63u) & ~63u) + 64u : E;          // (the last segment: one more trip behind the data for the end token)
    for (u32 P = S; P < trips_end; P += 64) {
        const bool tail = P >= n;
        const u32 p = P + (u32)lane;
        const u64 sm = tail ? 0ull : sm_n; const uint2 a = a_n; const u32 sb = sb_n;
        if (!tail && P + 64 < E) { sm_n = mask[(P + 64) >> 6]; a_n = ldm(p + 64u); sb_n = p + 64u < n ? src[p + 64u] : 0u; }
        // ---- enc_emit_prs_kernel
        bool start = !tail && ((sm >> lane) & 1ull) && p < n;
        uint2 mt = make_uint2(0, 0);
        if (start) mt = a;
        if (start && mt.y == 2u && mt.x > 0x100u) start = false;               // PRS.cs: not worth a long match -- literals
        const u32 mend = start ? p + mt.y : 0u;
        const u32 pmax = scan_max(mend);
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
        if (!EMIT) {
            const u64 tm = __ballot(tok);
            if (tm) last_bp = (u32)__builtin_amdgcn_readlane((int)Bp, 63 - (int)__builtin_clzll(tm));
        } else {
            const bool special = shortm && (Bp & 7u) == 0u;
            const u32 kp = Bp >> 3;                                            // the flag byte my payload waits for (special: the one it follows)
            const u32 out = kp + 1u - (special ? 1u : 0u) + pidx;              // where my payload goes
            const u32 kinc = scan_max(tok ? kp + 1u : 0u);                      // the latest kp + 1 up to and including my lane
            const u32 kexc = (u32)__builtin_amdgcn_update_dpp(0, (int)kinc, 0x138, 0xF, 0xF, false);
            const u32 prevk = kexc ? kexc - 1u : lastk;
            const bool opener = tok && (prevk == 0xFFFFFFFFu || prevk < kp);
            if (opener) { gofs[kp & 63u] = kp + pidx + (special ? 1u : 0u); }
            __syncthreads();
            if (tok) {
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
#pragma unroll
                for (u32 i = 0; i < 4u; i++) {
                    const u32 b = B0 + i;
                    if (i < nbits && (b & 7u) == 7u) {                          // the flag byte whose last bit is mine is complete
                        const u32 kk = b >> 3;
                        if (foreign && kk == k0) rec->head = flagacc[kk & 63u];     // (opened in front of this segment: enc_prs_flags_kernel has its address)
                        else { const u32 fo = gofs[kk & 63u]; if (fo < cap) dst[fo] = (u8)flagacc[kk & 63u]; else fail = true; }
                        flagacc[kk & 63u] = 0;
                    }
                }
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
            { const u32 last = (u32)__builtin_amdgcn_readlane((int)kinc, 63); if (last) lastk = last - 1u; }
        }
        bit_base += (u32)__builtin_amdgcn_readlane((int)bincl, 63);
        pay_base += (u32)__builtin_amdgcn_readlane((int)pincl, 63);
        const u32 wmax = (u32)__builtin_amdgcn_readlane((int)pmax, 63);
        if (wmax > cover) cover = wmax;
    }
    if (!EMIT) {
        if (lane == 0) { SegRec r; r.tok = bit_base; r.pay = pay_base; r.unc = last_bp; r.head = 0; r.tailbits = 0; r.tailofs = 0; r.fail = 0; r.pad = 0; *rec = r; }
        return;
    }
    const bool anyfail = __ballot(fail) != 0ull;
    if (lane == 0) {
        const u32 kt = bit_base >> 3;                                          // (bit_base: one behind the segment's last bit)
        const bool open = (bit_base & 7u) != 0u || lastk == kt;                 // flag byte kt is open behind this segment
        if (foreign && kt == k0) rec->head = flagacc[k0 & 63u];                 // still the byte it began in
        else if (open) { rec->tailbits = flagacc[kt & 63u]; rec->tailofs = gofs[kt & 63u]; }
        rec->fail = anyfail ? 1u : 0u;
    }
}

// P: flag bits and payload bytes in front of every segment, and `lastk` there: the flag byte the last payload in front of it waited for
__global__ __launch_bounds__(64) void enc_prs_prefix_kernel(const alz_stream* __restrict__ streams, const u32* __restrict__ index_list, SegRec* __restrict__ seg,
                                                            u32* __restrict__ stot, u32 kpitch, u32 seglen) {
    const u32 bid = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const u32 n = streams[index_list[bid]].src_len;
    const u32 K = n ? (n + seglen - 1u) / seglen : 1u;
    SegRec* rec = seg + (size_t)bid * kpitch;
    u32 cb = 0, cp = 0, clast = 0xFFFFFFFFu;
    for (u32 k0 = 0; k0 < K; k0 += 64) {
        const u32 k = k0 + (u32)lane;
        u32 b = 0, p = 0, lb = 0xFFFFFFFFu;
        if (k < K) { b = rec[k].tok; p = rec[k].pay; lb = rec[k].unc; }
        const u32 bi = scan_add(b), pi = scan_add(p);
        const u32 bbase = cb + bi - b;
        const u32 lk_out = lb != 0xFFFFFFFFu ? ((bbase + lb) >> 3) + 1u : 0u;          // (+ 1: 0 = the segment has no token)
        // the last segment in front of me that has a token: lk_out never falls along the stream, so a prefix maximum finds it
        const u32 linc = scan_max(lk_out);
        const u32 lexc = (u32)__builtin_amdgcn_update_dpp(0, (int)linc, 0x138, 0xF, 0xF, false);
        const u32 lin = lexc ? lexc - 1u : clast;
        if (k < K) { rec[k].tok = bbase; rec[k].pay = cp + pi - p; rec[k].unc = lin; }
        cb += (u32)__builtin_amdgcn_readlane((int)bi, 63); cp += (u32)__builtin_amdgcn_readlane((int)pi, 63);
        const u32 lm = (u32)__builtin_amdgcn_readlane((int)linc, 63);
        if (lm) clast = lm - 1u;
    }
    if (lane == 0) { stot[4 * (size_t)bid] = cb; stot[4 * (size_t)bid + 1] = cp; stot[4 * (size_t)bid + 2] = clast; stot[4 * (size_t)bid + 3] = 0; }
}

template <bool BIG>
__global__ __launch_bounds__(64) void enc_prs_flags_kernel(u8* __restrict__ dst_base, const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                           const SegRec* __restrict__ seg, const u32* __restrict__ stot, u32 kpitch, u32 seglen,
                                                           alz_result* __restrict__ results, alz_encode_aux* __restrict__ aux) {
    const u32 bid = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const u32 n = st.src_len, cap = st.dst_cap;
    u8* dst = dst_base + st.dst_off;
    const u32 K = n ? (n + seglen - 1u) / seglen : 1u;
    const SegRec* rec = seg + (size_t)bid * kpitch;
    const u32 bit_total = stot[4 * (size_t)bid], pay_total = stot[4 * (size_t)bid + 1], last_total = stot[4 * (size_t)bid + 2];
    bool fail = false;
    for (u32 k = (u32)lane; k < K; k += 64) {
        if (rec[k].fail) fail = true;
        const u32 bb = rec[k].tok, be = k + 1u < K ? rec[k + 1u].tok : bit_total;
        const u32 lk_in = rec[k].unc, lk_out = k + 1u < K ? rec[k + 1u].unc : last_total;
        const u32 k0 = bb >> 3, kt = be >> 3;
        const bool foreign = (bb & 7u) != 0u || lk_in == k0;
        const bool open = (be & 7u) != 0u || lk_out == kt;
        if (open && !(foreign && kt == k0)) {                                   // this segment opened flag byte kt and did not finish it
            u32 acc = rec[k].tailbits;
            const u32 fo = rec[k].tailofs;
            for (u32 j = k + 1u; j < K && (rec[j].tok >> 3) == kt; j++) acc |= rec[j].head;     // (every such segment found kt open in front of it)
            if (fo < cap) dst[fo] = (u8)acc; else fail = true;
        }
    }
    const u32 total = ((bit_total + 7u) >> 3) + pay_total;
    const bool anyfail = __ballot(fail) != 0ull || total > cap;
    if (lane == 0) {
        alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
        results[sid] = r;
        if (aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
    }
}

template <bool BIG>
static void launch_emit_seg_prs(hipStream_t s, u32 count, const u8* src, u8* dst, const alz_stream* streams, const u32* index, mentry* match, const u64* pos_off,
                                const int* prev4, const int* prevm, u64* mask, void* d_seg, u32 seglen, u32 kmax, alz_result* results, alz_encode_aux* aux, const EncGeom& g) {
    SegRec* seg = (SegRec*)d_seg;
    u32* stot = (u32*)((u8*)d_seg + (size_t)count * kmax * sizeof(SegRec));
    u32* sync = stot + 4 * (size_t)count;
    launch_seg_walk(s, count, src, streams, index, match, pos_off, prev4, prevm, mask, sync, seglen, kmax, g);
    hipLaunchKernelGGL((enc_prs_seg_kernel<BIG, false>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, kmax, seglen, g);
    hipLaunchKernelGGL(enc_prs_prefix_kernel, dim3(count), dim3(64), 0, s, streams, index, seg, stot, kmax, seglen);
    hipLaunchKernelGGL((enc_prs_seg_kernel<BIG, true>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, kmax, seglen, g);
    hipLaunchKernelGGL((enc_prs_flags_kernel<BIG>), dim3(count), dim3(64), 0, s, dst, streams, index, (const SegRec*)seg, (const u32*)stot, kmax, seglen, results, aux);
}
