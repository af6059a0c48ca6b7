// alz_encode_seg_seq.h -- raw Snappy for batches of FEW buffers (a framed Snappy stream is chunks of 64 KiB): alz_encode_seg.h's arrangement with the
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
    hipLaunchKernelGGL(enc_sync_kernel, dim3(kmax, count), dim3(64), 0, s, streams, index, (const mentry*)match, pos_off, sync, kmax, seglen, g);
    hipLaunchKernelGGL((enc_roles_kernel<true>), dim3(kmax, count), dim3(64), 0, s, src, streams, index, count, match, pos_off, prev4, prevm, mask, g, 0, (const u32*)sync, kmax);
    hipLaunchKernelGGL((enc_seq_seg_kernel<FMT, false>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, kmax, seglen, g);
    hipLaunchKernelGGL((enc_seq_prefix_kernel<FMT>), dim3(count), dim3(64), 0, s, streams, index, seg, stot, kmax, seglen);
    hipLaunchKernelGGL((enc_seq_seg_kernel<FMT, true>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, kmax, seglen, g);
    hipLaunchKernelGGL((enc_seq_finish_kernel<FMT>), dim3(count), dim3(64), 0, s, src, dst, streams, index, (const SegRec*)seg, (const u32*)stot, kmax, seglen, results, aux);
}
