// alz_encode_seg_seq.h -- raw Snappy and PRS for batches of FEW buffers (a framed Snappy stream is chunks of 64 KiB): alz_encode_seg.h's arrangement with the
// emitter of enc_parse_seq_kernel.  Included by alz_encode.hip behind that kernel.
//
// A sequence is a match start with the literals since the match before it (Snappy.cs:124-203); a segment owns the sequences whose match STARTS in it.
// Only a segment's FIRST sequence depends on what lies in front of the segment -- its literals begin at the end of the last match before it, however far
// back that is --, so the count pass leaves that one out and hands its start and match to the prefix kernel, which knows every segment's last match end.
// LZ4 blocks and LZO, whose matches have no longest length, are not on this path: a synchronisation point needs every jump that could cross it, and
// kernel B only measures matches up to its compare cap.
//   sync + walk as for the flag-bit formats
//   C  enc_seq_seg_kernel<FMT, false>   per segment: bytes of its sequences but the first; the first one's start, distance and length; the end of its last match
//   P  enc_seq_prefix_kernel<FMT>       per buffer: the end of the last match in front of each segment, the first sequences' sizes, the byte offsets
//   E  enc_seq_seg_kernel<FMT, true>    the sequences, with enc_parse_seq_kernel's arithmetic (literals of earlier windows and segments go with the first start behind them)
//   F  enc_seq_finish_kernel<FMT>       the length varint, the literals behind the last match, the result

template <int FMT, bool EMIT>
__global__ __launch_bounds__(64) void enc_seq_seg_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                         const u32* __restrict__ index_list, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                         const u64* __restrict__ startmask, SegRec* __restrict__ seg, u32 kpitch, u32 seglen, EncGeom g) {
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
    auto ldm = [&](u32 q) { return (int)q <= limit ? m_unpack(m[q]) : make_uint2(0, 0); };
    u64 sm_n = mask[S >> 6];
    uint2 a_n = ldm(S + (u32)lane);
    for (u32 P = S; P < E; P += 64) {
        const u32 p = P + (u32)lane;
        const u64 sm = sm_n; const uint2 a = a_n;
        if (P + 64 < E) { sm_n = mask[(P + 64) >> 6]; a_n = ldm(p + 64u); }
        if (sm == 0ull) continue;
        // ---- the sequences that start in this window (enc_parse_seq_kernel)
        const bool start = ((sm >> lane) & 1ull) != 0ull;
        const u32 M = start ? a.y : 0u, D = a.x;
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
                F::put_match(dst + off + lh + L, D, M);
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
    if (lane == 0) { stot[4 * (size_t)bid] = cbytes; stot[4 * (size_t)bid + 1] = ccover; stot[4 * (size_t)bid + 2] = 0; stot[4 * (size_t)bid + 3] = 0; }
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
    const u32 K = (n + seglen - 1u) / seglen;
    const SegRec* rec = seg + (size_t)bid * kpitch;
    bool fail = false;
    for (u32 k = (u32)lane; k < K; k += 64) if (rec[k].fail) fail = true;
    const u32 obase = stot[4 * (size_t)bid], cover = stot[4 * (size_t)bid + 1];
    if (FMT == ALZ_FMT_SNAPPY_RAW) {
        const u32 kv = snappy_varint_size(n);
        if (kv <= cap) { if (lane == 0) { u32 v = n, q = 0; while (v >= 0x80u) { dst[q++] = (u8)((v | 0x80u) & 0xFFu); v >>= 7; } dst[q] = (u8)v; } } else fail = true;
    }
    // the end: the remaining literals (Snappy: an element only if there are any)
    const u32 plain = n - cover, lh = plain ? F::lit_hdr(plain) : 0u;
    const u32 total = obase + lh + plain;
    if (total > cap) fail = true;
    const bool anyfail = __ballot(fail) != 0ull;
    if (!anyfail && plain) {
        if (lane == 0) F::put_lit_hdr(dst + obase, plain, 4u, true);
        wave_copy(dst + obase + lh, src + cover, plain, lane);
    }
    if (lane == 0) {
        alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
        results[sid] = r;
        if (aux) { aux[sid].aux0 = 0; aux[sid].aux1 = 0; }
    }
}

template <int FMT>
static void launch_emit_seg_seq(hipStream_t s, u32 count, const u8* src, u8* dst, const alz_stream* streams, const u32* index, mentry* match, const u64* pos_off,
                                const int* prev4, const int* prevm, u64* mask, void* d_seg, u32 seglen, u32 kmax, alz_result* results, alz_encode_aux* aux, const EncGeom& g) {
    SegRec* seg = (SegRec*)d_seg;
    u32* stot = (u32*)((u8*)d_seg + (size_t)count * kmax * sizeof(SegRec));
    u32* sync = stot + 4 * (size_t)count;
    launch_seg_walk(s, count, src, streams, index, match, pos_off, prev4, prevm, mask, sync, seglen, kmax, g);
    hipLaunchKernelGGL((enc_seq_seg_kernel<FMT, false>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, kmax, seglen, g);
    hipLaunchKernelGGL((enc_seq_prefix_kernel<FMT>), dim3(count), dim3(64), 0, s, streams, index, seg, stot, kmax, seglen);
    hipLaunchKernelGGL((enc_seq_seg_kernel<FMT, true>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, kmax, seglen, g);
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
