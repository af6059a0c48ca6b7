// alz_encode_seg.h -- the flag-bit emitter for batches of FEW buffers (tens to hundreds of 64 KiB - 1 MiB: a directory of files, the chunks of
// one archive), included by alz_encode.hip.
//
// enc_parse_emit_kernel gives a buffer ONE wavefront for the parse and its tokens: ~1.7 us per window of 64 positions whatever else the GPU is
// doing, 1.7 ms per 64 KiB, while 256 buffers leave 24 of a CU's 25 places empty (tools/mid_batch_encode.py: 16-256 buffers of 64 KiB as Yaz0 at
// quality 8 took 1.8-3.4 ms).  Only the WALK of the parse is serial (cursor += jump[cursor]); everything the emitter does with the token starts is
// prefix sums over positions -- and the walk itself has points every walk passes.  So, for such batches (alz_encode_segmented):
//   S  enc_sync_kernel        per segment boundary the last SYNCHRONISATION POINT in front of it: a position no jump crosses, which every walk lands on
//   X  enc_exit_kernel        per segment exit(e), the first cursor behind the next boundary on the chain from e, by a right-to-left sweep (down to the boundary's
//      enc_compose_kernel     synchronisation point, or over the whole segment into a table); the cursor that enters every segment, strung together per buffer
//   R  enc_roles_kernel<true> the walk, one wavefront per segment: a bit per position "a match token starts here" (kernel B runs without a compare cap on this
//                             path: no position is left for the walk to search)
//   C  enc_seg_kernel<false>  one wavefront per SEGMENT of a buffer (1-8 Ki positions, so that the launch has a few thousand): the tokens, payload
//                             bytes and literal-section bytes that start in it.  What it needs from the left is the end of the last match that
//                             starts before the segment: at most maxLength back in the start mask
//   P  enc_seg_prefix_kernel  exclusive prefix sums over a buffer's segments (one wavefront per buffer)
//   E  enc_seg_kernel<true>   the same windows again, now with their offsets: payload bytes, and every flag byte whose group of eight tokens
//                             begins AND ends in the segment
//   F  enc_seg_flags_kernel   the flag bytes of groups that straddle segments (the bits each side found, OR-ed), the partial last one
//                             (FlagWriter.Dispose  IO/FlagWriter.cs:141-145), the results
// Same bytes as enc_parse_emit_kernel by construction: the same per-window arithmetic (LZSS.cs:132-160, LZ10.cs:113-137, Yaz0 / Yay0 / MIO0 ...
// through flag_payload), the same order.  Formats whose longest match fits 63 windows (not LZ11 / LZ40: 16 KiB); raw Snappy and PRS: alz_encode_seg_seq.h.

// The longest jump the walk's tables have to hold: a match entry kernel B wrote is shorter than its compare cap (ALZ_LEN_CAP, what the segmented path runs it with) or CAPPED, and a
// capped entry makes its stretch of the buffer the serial walker's -- so for LZ11 / LZ40, whose matches reach 16 KiB (round 6), the exit tables and the ring are those of a 2 040-byte
// format, and only enc_sync_kernel looks back the full 16 KiB (a capped position that far in front may jump over the boundary).
static inline u32 seg_table_hist(const EncGeom& g) { const u32 jl = g.max_len > ALZ_LEN_CAP ? (u32)ALZ_LEN_CAP : (u32)g.max_len; return (jl + 2u + 63u) & ~63u; }

struct SegRec { u32 tok, pay, unc, head, tailbits, tailofs, fail, pad; };     // counts (C) -> exclusive prefix (P); the flag bits of straddling groups (E)

template <int FMT, bool EMIT>
__global__ __launch_bounds__(64) void enc_seg_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                     const u32* __restrict__ index_list, const mentry* __restrict__ match, const u64* __restrict__ pos_off,
                                                     const u64* __restrict__ startmask, SegRec* __restrict__ seg, const u32* __restrict__ stot,
                                                     u32 kpitch, u32 seglen, EncGeom g) {
    constexpr bool THREE = (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0);
    constexpr bool LIT_BIT = (FMT == ALZ_FMT_LZSS || FMT == ALZ_FMT_YAZ0 || FMT == ALZ_FMT_LZHUDSON || THREE);
    constexpr bool MSB = (FMT != ALZ_FMT_LZSS && FMT != ALZ_FMT_CLZ0);
    constexpr u32 FBITS = FMT == ALZ_FMT_LZHUDSON ? 32u : 8u, FB = FBITS / 8u;
    __shared__ u32 flagacc[16];
    __shared__ u32 gofs[16];
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
    if (lane < 16) { flagacc[lane] = 0; gofs[lane] = 0; }
    __syncthreads();
    // the end of the last match that starts in front of the segment: a match further back than maxLength cannot reach it
    u32 cover = 0;
    if (k) {
        // (64 windows per round: one round for the formats up to 2 040 bytes, five for LZ11 / LZ40; a taken match of 2 046 bytes or more has its length in the NEXT entry)
        for (u32 r0 = 0; r0 * 64u < (u32)g.max_len + 64u; r0 += 64u) {
            const int w = (int)(S >> 6) - 1 - (int)r0 - lane;
            u32 endv = 0;
            if (w >= 0 && (r0 + (u32)lane) * 64u < (u32)g.max_len + 64u) {
                const u64 mw = mask[w];
                if (mw) { const u32 q = (u32)w * 64u + 63u - (u32)__builtin_clzll(mw); u32 ml = m_unpack(m[q]).y; if (ml == ALZ_M_LONG) ml = m[q + 1]; endv = q + ml; }
            }
            const u32 cv = (u32)__builtin_amdgcn_readlane((int)scan_max(endv), 63);
            if (cv > cover) cover = cv;
        }
    }
    u32 tok_base = 0, pay_base = 0, unc_base = 0, nflags = 0, pay_total = 0;
    if (EMIT) {
        tok_base = rec->tok; pay_base = rec->pay; unc_base = rec->unc;
        if (THREE) { nflags = FB * ((stot[4 * (size_t)bid] + FBITS - 1u) / FBITS); pay_total = stot[4 * (size_t)bid + 1]; }
    }
    const u32 g0 = tok_base / FBITS;
    const bool straddle = (tok_base % FBITS) != 0u;          // the first tokens complete a group that began in the segment before
    bool fail = false;
    auto ldm = [&](u32 q) { return (int)q <= limit ? m_unpack(m[q]) : make_uint2(0, 0); };
    // (mask word, match entries and source bytes of a window are loaded while the window before it is worked on)
    u64 sm_n = mask[S >> 6];
    uint2 a_n = ldm(S + (u32)lane);
    u32 sb_n = S + (u32)lane < n ? src[S + (u32)lane] : 0u;
    for (u32 P = S; P < E; P += 64) {
        const u32 p = P + (u32)lane;
        const u64 sm = sm_n; const uint2 a = a_n; const u32 sb = sb_n;
        if (P + 64 < E) { sm_n = mask[(P + 64) >> 6]; a_n = ldm(p + 64u); sb_n = p + 64u < n ? src[p + 64u] : 0u; }
        // ---- as enc_parse_emit_kernel: prefix sums over the start mask give every token its flag group and byte offset
        const bool start = ((sm >> lane) & 1ull) && p < n;
        uint2 mt = make_uint2(0, 0);
        if (start) mt = a;
        if (FMT == ALZ_FMT_LZ11 || FMT == ALZ_FMT_LZ40) { if (start && mt.y == ALZ_M_LONG) mt.y = m[p + 1]; }     // (a taken match of 2 046 bytes or more: the walk left its length in the next entry)
        const u32 mend = start ? p + mt.y : 0u;
        const u32 pmax = scan_max(mend);
        u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)pmax, 0x138, 0xF, 0xF, false);   // wave_shr:1 -> max over lanes below
        if (before < cover) before = cover;
        const bool lit = !start && p < n && p >= before;
        const bool tok = start || lit;
        const u64 tm = __ballot(tok);
        const u32 ti = tok_base + __builtin_amdgcn_mbcnt_hi((u32)(tm >> 32), __builtin_amdgcn_mbcnt_lo((u32)tm, 0u));
        u32 b0 = 0, b1 = 0, b2 = 0, b3 = 0, psize = 0, usize = 0;
        if (lit) { b0 = sb; psize = 1; }
        else if (start) flag_payload<FMT>(g, p, mt, b0, b1, b2, b3, psize);
        if (THREE) {
            if (lit) { usize = 1; psize = 0; }
            else if (start && FMT == ALZ_FMT_YAY0 && psize == 3) { usize = 1; psize = 2; }
        }
        const u32 pincl = scan_add(psize);
        const u32 poff = pay_base + pincl - psize;
        const u32 uincl = THREE ? scan_add(usize) : 0u;
        const u32 uoff = unc_base + uincl - usize;
        if (EMIT) {
            const u32 group = ti / FBITS, bitpos = ti % FBITS;
            const u32 flag_off = THREE ? group : poff + FB * group;
            if (tok && bitpos == 0) { gofs[group & 15u] = flag_off; flagacc[group & 15u] = 0; }
            __syncthreads();
            if (tok) {
                const u32 bitv = (lit ? LIT_BIT : !LIT_BIT) ? 1u : 0u;
                if (bitv) atomicOr(&flagacc[group & 15u], 1u << (MSB ? FBITS - 1u - bitpos : bitpos));
            }
            __syncthreads();
            if (tok) {
                if (bitpos == FBITS - 1u) {
                    const u32 acc = flagacc[group & 15u];
                    if (straddle && group == g0) rec->head = acc;             // (its flag byte lies in the segment before: enc_seg_flags_kernel)
                    else {
                        const u32 fo = gofs[group & 15u];
                        if (fo + FB <= cap) { if (FB == 1u) dst[fo] = (u8)(FMT == ALZ_FMT_LZ40 ? 0u - acc : acc); else { dst[fo] = (u8)(acc >> 24); dst[fo + 1] = (u8)(acc >> 16); dst[fo + 2] = (u8)(acc >> 8); dst[fo + 3] = (u8)acc; } }
                        else fail = true;
                    }
                }
                if (!THREE) {
                    const u32 o = poff + FB * (group + 1u);
                    if (o + psize <= cap) { dst[o] = (u8)b0; if (psize > 1) dst[o + 1] = (u8)b1; if (psize > 2) dst[o + 2] = (u8)b2; if (psize > 3) dst[o + 3] = (u8)b3; }
                    else fail = true;
                } else {
                    // (the sections straight into place: their offsets are known)
                    const u32 oc = nflags + poff, ou = nflags + pay_total + uoff;
                    if (lit) { if (ou < cap) dst[ou] = (u8)b0; else fail = true; }
                    else {
                        if (oc + 2u <= cap) { dst[oc] = (u8)b0; dst[oc + 1] = (u8)b1; } else fail = true;
                        if (usize) { if (ou < cap) dst[ou] = (u8)b2; else fail = true; }
                    }
                }
            }
            __syncthreads();
        }
        tok_base += (u32)__popcll(tm);
        pay_base += (u32)__builtin_amdgcn_readlane((int)pincl, 63);
        if (THREE) unc_base += (u32)__builtin_amdgcn_readlane((int)uincl, 63);
        const u32 wmax = (u32)__builtin_amdgcn_readlane((int)pmax, 63);
        if (wmax > cover) cover = wmax;
    }
    if (!EMIT) {
        if (lane == 0) { SegRec r; r.tok = tok_base; r.pay = pay_base; r.unc = unc_base; r.head = 0; r.tailbits = 0; r.tailofs = 0; r.fail = 0; r.pad = 0; *rec = r; }
        return;
    }
    const bool anyfail = __ballot(fail) != 0ull;
    if (lane == 0) {
        const u32 gt = tok_base / FBITS;                                       // (tok_base: one behind the segment's last token)
        if (straddle && gt == g0) rec->head = flagacc[g0 & 15u];               // still inside the group it began in
        else if ((tok_base % FBITS) != 0u) { rec->tailbits = flagacc[gt & 15u]; rec->tailofs = gofs[gt & 15u]; }
        rec->fail = anyfail ? 1u : 0u;
    }
}

// P: a buffer's segment counts -> exclusive prefix sums, in place; the totals to stot[4 * buffer ...]
__global__ __launch_bounds__(64) void enc_seg_prefix_kernel(const alz_stream* __restrict__ streams, const u32* __restrict__ index_list, SegRec* __restrict__ seg,
                                                            u32* __restrict__ stot, u32 kpitch, u32 seglen) {
    const u32 bid = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const u32 n = streams[index_list[bid]].src_len;
    const u32 K = (n + seglen - 1u) / seglen;
    SegRec* rec = seg + (size_t)bid * kpitch;
    u32 ct = 0, cp = 0, cu = 0;
    for (u32 k0 = 0; k0 < K; k0 += 64) {
        const u32 k = k0 + (u32)lane;
        u32 t = 0, p = 0, u = 0;
        if (k < K) { t = rec[k].tok; p = rec[k].pay; u = rec[k].unc; }
        const u32 ti = scan_add(t), pi = scan_add(p), ui = scan_add(u);
        if (k < K) { rec[k].tok = ct + ti - t; rec[k].pay = cp + pi - p; rec[k].unc = cu + ui - u; }
        ct += (u32)__builtin_amdgcn_readlane((int)ti, 63); cp += (u32)__builtin_amdgcn_readlane((int)pi, 63); cu += (u32)__builtin_amdgcn_readlane((int)ui, 63);
    }
    if (lane == 0) { stot[4 * (size_t)bid] = ct; stot[4 * (size_t)bid + 1] = cp; stot[4 * (size_t)bid + 2] = cu; stot[4 * (size_t)bid + 3] = 0; }
}

// F: the flag bytes of the groups that do not begin and end inside one segment, and the buffer's result
template <int FMT>
__global__ __launch_bounds__(64) void enc_seg_flags_kernel(u8* __restrict__ dst_base, const alz_stream* __restrict__ streams, const u32* __restrict__ index_list,
                                                           const SegRec* __restrict__ seg, const u32* __restrict__ stot, u32 kpitch, u32 seglen,
                                                           alz_result* __restrict__ results, alz_encode_aux* __restrict__ aux) {
    constexpr bool THREE = (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0);
    constexpr u32 FBITS = FMT == ALZ_FMT_LZHUDSON ? 32u : 8u, FB = FBITS / 8u;
    const u32 bid = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list[bid];
    const alz_stream st = streams[sid];
    const u32 n = st.src_len, cap = st.dst_cap;
    u8* dst = dst_base + st.dst_off;
    const u32 K = (n + seglen - 1u) / seglen;
    const SegRec* rec = seg + (size_t)bid * kpitch;
    const u32 tok_total = stot[4 * (size_t)bid], pay_total = stot[4 * (size_t)bid + 1], unc_total = stot[4 * (size_t)bid + 2];
    bool fail = false;
    for (u32 k = (u32)lane; k < K; k += 64) {
        const u32 tb = rec[k].tok, te = k + 1u < K ? rec[k + 1u].tok : tok_total;
        const u32 g0 = tb / FBITS, gt = te / FBITS;
        const bool straddle = (tb % FBITS) != 0u;
        if (rec[k].fail) fail = true;
        if ((te % FBITS) != 0u && !(straddle && gt == g0)) {                    // a group begins in this segment and does not end in it
            u32 acc = rec[k].tailbits;
            const u32 fo = rec[k].tailofs;
            for (u32 j = k + 1u; j < K && rec[j].tok / FBITS == gt; j++) acc |= rec[j].head;
            if (fo + FB <= cap) { if (FB == 1u) dst[fo] = (u8)(FMT == ALZ_FMT_LZ40 ? 0u - acc : acc); else { dst[fo] = (u8)(acc >> 24); dst[fo + 1] = (u8)(acc >> 16); dst[fo + 2] = (u8)(acc >> 8); dst[fo + 3] = (u8)acc; } }
            else fail = true;
        }
    }
    const u32 nflags = FB * ((tok_total + FBITS - 1u) / FBITS);
    const u32 total = nflags + pay_total + (THREE ? unc_total : 0u);
    const bool anyfail = __ballot(fail) != 0ull || total > cap;
    if (lane == 0) {
        alz_result r; r.dst_len = anyfail ? 0u : total; r.src_used = n; r.status = anyfail ? ALZ_ST_OUTPUT_CAPACITY : ALZ_ST_OK; r.reserved = 0;
        results[sid] = r;
        if (aux) { aux[sid].aux0 = THREE ? nflags : 0u; aux[sid].aux1 = THREE ? nflags + pay_total : 0u; }
    }
}

// Which launches go this way, and with what segments.  One wavefront per buffer takes len / 64 x ~1.7 us whatever the batch; this path is bound by
// kernel B's throughput (it needs the match array even at quality 0, and runs B without a compare cap).  Where the two cross, in buffers of one format per
// call (tools/mid_batch_encode.py on 64 KiB and 256 KiB windows of Test.bmp and of program text, profiles/r05_mid_batch_encode_sweep2.txt): quality 0 -- the other
// side searches inside its parse, no match array -- between 1 024 (Yaz0 2.53 -> 2.06 ms, LZ10 2.05 -> 1.84, Snappy 2.88 -> 1.88) and 2 048 (3.43 -> 4.13);
// quality 1-10 around 2 048 (Yaz0 still wins: 12.7 -> 11.8 at quality 8; LZ10 loses: 5.3 -> 5.7); quality 11-15 -- chains of 64 and more candidates, where
// the cap saves kernel B most -- between 512 (23.4 -> 17.7) and 1 024 (29.2 -> 34.1).
#ifndef ALZ_SEG_MIN_LEN
#define ALZ_SEG_MIN_LEN 8192u
#endif
#ifndef ALZ_SEG_WAVES
#define ALZ_SEG_WAVES 8192u      /* segments a launch aims at (4 096: 5-20 % slower from 256 buffers on -- 256 x 64 KiB as Yaz0 at quality 0 0.59 -> 0.67 ms --; 16 384: within 2 % either way) */
#endif
}  // namespace

// max_streams: 0xFFFFFFFF: the rule above; 0: the path is off; anything else: that many buffers instead of the rule (a context's debug override, alz_debug_seg_max_streams:
// tests, tools/mid_batch_encode.py) -- never more than one launch of encode_core takes (65 535 buffers: the scratch is laid out for the launch's own count)
// the words a segment's record holds behind its three fixed ones (alz_encode_seg_bytes): the exit table of the longest jump -- or, for the formats of the speculative walk
// (LZ4 blocks, LZO: alz_encode_seg_seq.h), SpecRec and the segment's cursor mask.  ONE function for the host's sizing and the launch's layout.
#ifndef ALZ_SPEC_LONG11
#define ALZ_SPEC_LONG11 1     /* LZ11 / LZ40 (matches of up to 16 KiB and more) on the speculative walk as well -- 0: synchronisation points with a 16 KiB look-back, up to 256 buffers.  64 KiB windows of
                                 Test.bmp as LZ11, ms per call, 0 / 1: quality 8 16 buffers 0.44 / 0.33, 64 1.06 / 0.70, 256 2.77 / 1.82, 512 4.46 (one wavefront per buffer) / 3.13, 1 024 6.92 / 5.93; quality 0 at 256
                                 1.56 / 0.61; quality 12 at 64 / 256: 6.1 / 2.0, 22.9 / 3.8 (the stretches with capped entries were walked serially) */
#endif
// the formats whose segments are walked speculatively (alz_encode_seg_seq.h): no longest match that a look-back could be bounded by
#ifndef ALZ_SPEC_FLAG
#define ALZ_SPEC_FLAG 0       /* the flag-bit formats with matches of at most 273 bytes on it too: measured, not taken -- 16 / 256 / 1 024 x 64 KiB as Yaz0 at quality 8 0.19 / 1.74 / 5.96 ms with the synchronisation
                                 points against 0.32 / 1.81 / 5.83 (every segment is a step of its buffer's serial fix-up: 64 steps of ~2 us for a 64 KiB buffer, which the synchronisation points do not have),
                                 quality 12 at 64 buffers 2.66 against 9.65 */
#endif
static inline bool seg_spec_format(int fmt) {
    if (ALZ_SPEC_FLAG && (fmt == ALZ_FMT_LZSS || fmt == ALZ_FMT_LZ10 || fmt == ALZ_FMT_YAZ0 || fmt == ALZ_FMT_YAY0 || fmt == ALZ_FMT_MIO0 || fmt == ALZ_FMT_CLZ0 || fmt == ALZ_FMT_BLZ || fmt == ALZ_FMT_LZHUDSON)) return true;
    return fmt == ALZ_FMT_LZ4_BLOCK || fmt == ALZ_FMT_LZO || (ALZ_SPEC_LONG11 && (fmt == ALZ_FMT_LZ11 || fmt == ALZ_FMT_LZ40));
}
static inline u32 seg_rec_hist(int fmt, const EncGeom& g, u32 seg_len) {
    return seg_spec_format(fmt) ? 1u + (seg_len >> 5) : seg_table_hist(g);
}
int alz_encode_seg_spec_format(int fmt) { return seg_spec_format(fmt) ? 1 : 0; }
int alz_encode_segmented(int fmt, const void* geom, uint32_t count, uint32_t max_len, uint32_t max_streams, uint32_t* seg_len, uint32_t* kmax, uint32_t* hist_out) {
    EncGeom g; memcpy(&g, geom, sizeof(g));
    // LZ4 blocks, LZO (round 6): no synchronisation points -- every segment walked speculatively, the true walk strung together behind (alz_encode_seg_seq.h: enc_spec_walk_kernel)
    const bool spec4 = seg_spec_format(fmt);
    const bool long11 = !spec4 && (fmt == ALZ_FMT_LZ11 || fmt == ALZ_FMT_LZ40);                                                                                          // (matches of up to 16 KiB: round 6, seg_table_hist)
    const bool fam = fmt == ALZ_FMT_LZSS || fmt == ALZ_FMT_LZ10 || fmt == ALZ_FMT_YAZ0 || fmt == ALZ_FMT_YAY0 || fmt == ALZ_FMT_MIO0 || fmt == ALZ_FMT_CLZ0 || long11 || spec4 ||
                     fmt == ALZ_FMT_BLZ || fmt == ALZ_FMT_LZHUDSON || fmt == ALZ_FMT_SNAPPY_RAW || fmt == ALZ_FMT_PRS_BE || fmt == ALZ_FMT_PRS_LE;        // (raw Snappy, PRS: alz_encode_seg_seq.h)
    const u32 rule = g.max_chain == 1 ? 1280u : g.max_chain < 64 ? 1536u : 512u;
    // (LZ11 / LZ40: enc_sync_kernel looks back 16 KiB per boundary and the stretches with capped entries stay serial -- 64 KiB windows of Test.bmp, ms per call, one wavefront per
    // buffer -> segments: 16 buffers 1.83 / 2.13 -> 0.42 / 0.44 at quality 0 / 8, 64: 1.84 / 2.57 -> 0.75 / 1.07, 256: 1.92 / 3.16 -> 1.57 / 2.78, 1 024: 2.71 / 5.19 -> 4.26 / 8.67)
    // (LZ4 blocks, 64 KiB windows of Test.bmp 4 KiB apart, ms per call without / with the path -- quality 8: 16 buffers 1.74 / 0.42, 64 2.62 / 1.06, 256 3.43 / 2.22, 512 4.95 / 3.94,
    // 1 024 8.04 / 7.59; quality 0 at 256: 1.90 / 0.82; quality 12: 19.8 / 5.2, quality 15: 66.5 / 20.4; 1 500 x 16 KiB at quality 8: 2.71 / 2.93 -- up to 1 024 buffers then)
    // (... since the walk's hops are scalar loops and the segments are cut finer: 1 024 / 1 536 / 2 048 x 64 KiB without / with the path, quality 0 2.16 / 2.42 / 2.74 against 1.69 / 2.53 / 3.23,
    // quality 8 7.53 / 10.35 / 8.94 -- from 2 048 buffers on the scan path -- against 6.62 / 9.61 / 13.30, quality 12 15.8 / -- / 24.4 against 11.0 / -- / 20.7; 1 500 x 16 KiB at quality 8 2.63 against 2.41:
    // the common line, and 2 048 buffers for the long chains)
    const u32 rule2 = (long11 && rule > 256u) ? 256u : (spec4 && g.max_chain >= 64) ? 2048u : rule;
    const u32 most = max_streams == 0xFFFFFFFFu ? rule2 : (max_streams < 65535u ? max_streams : 65535u);
    if (!fam || (g.max_len > 2040 && !long11 && !spec4) || g.nprops > 1 || count == 0 || count > most || max_len < ALZ_SEG_MIN_LEN) return 0;
#ifndef ALZ_SPEC_WAVES
#define ALZ_SPEC_WAVES 12288u    /* the speculative walk's segments (its kernel holds 7 wavefronts per SIMD: 7 168 places, so 8 192 segments are two rounds of unequal length) -- 256 x 64 KiB of Test.bmp as LZ4
                                    blocks, ms per call at 4 096 / 6 144 / 7 168 / 8 192 / 12 288 / 16 384: quality 0 0.95 / 0.70 / 0.71 / 0.74 / 0.65 / 0.70, quality 8 2.28 / 2.10 / 2.11 / 2.21 / 2.05 / 2.12; windows
                                    spread over the whole file at quality 8, 8 192 / 12 288 / 16 384: LZ4 2.67 / 2.23 / 2.43, LZO 2.72 / 2.26 / 2.44; 1 024 buffers 7.73 / 7.16 / 7.67.  But every segment is one step of the
                                    serial fix-up of its buffer: 64 x 256 KiB 3.10 / 3.36 / 3.41 -- the finer cut only while a buffer has at most 64 segments */
#endif
    u32 waves = ALZ_SEG_WAVES;
    if (spec4) {
        uint64_t w2 = ((uint64_t)count * max_len + ALZ_SPEC_WAVES - 1u) / ALZ_SPEC_WAVES;
        if (w2 < 1024u) w2 = 1024u;
        if ((max_len + w2 - 1u) / w2 <= 64u) waves = ALZ_SPEC_WAVES;
    }
    uint64_t want = ((uint64_t)count * max_len + waves - 1u) / waves;
    if (want < 1024u) want = 1024u;
    u32 sl = (u32)((want + 63u) & ~(uint64_t)63u);
    u32 hist = seg_table_hist(g);                                          // the longest jump, in whole windows: what a segment's exit table covers (enc_exit_kernel)
    if (spec4) {
        // (a segment's masks live in LDS -- at most 8 192 positions --, and its record is SpecRec + the cursor mask: `hist` words behind three -- alz_encode_seg_bytes)
#ifndef ALZ_SPEC_SLMIN_K
#define ALZ_SPEC_SLMIN_K 32u     /* a segment is at least sqrt(K x longest buffer) positions: a walk costs ~1 us a window, the serial fix-up ~2 us a segment of the buffer, and while the launch does not fill the GPU both are
                                    latency -- 16 x 256 KiB as LZ4 blocks at quality 0, ms per call at K = 0 / 32 / 64 / 128: 0.90 / 0.74 / 0.76 / 0.81; 16 x 64 KiB 0.32 / 0.32 / 0.33 / 0.36; from 256 buffers on nothing moves */
#endif
        if (ALZ_SPEC_SLMIN_K) {
            u32 lo = 64u; while ((uint64_t)lo * lo < (uint64_t)ALZ_SPEC_SLMIN_K * max_len) lo += 64u;
            if (sl < lo) sl = lo;
        }
        if (sl > 8192u) sl = 8192u;
#ifndef ALZ_SPEC_ODD
#define ALZ_SPEC_ODD 1
#endif
        // An ODD number of windows per segment.  Every speculative walk starts on a segment boundary, and a boundary that falls on the same column of every row of a bitmap (2 048 bytes a
        // row, 2 048 positions a segment: 16 x 1 MiB of Test.bmp) finds the same thing there row after row -- in one stretch of 124 KB a match that swallows the whole segment, which the
        // true cursor, two bytes further on, does not have: 238 segments in a row without one speculative cursor to meet, all of them walked by the serial fix-up (8.4 ms instead of 3.4).
        if (ALZ_SPEC_ODD && ((sl >> 6) & 1u) == 0u) sl = sl >= 8192u ? sl - 64u : sl + 64u;
        if ((max_len + sl - 1u) / sl > 8192u) return 0;
        hist = seg_rec_hist(fmt, g, sl);
    } else {
    if (sl < hist) sl = hist;
    while ((max_len + sl - 1u) / sl > 8192u) sl += 64u;                     // (enc_compose_kernel holds a buffer's boundaries in LDS; never reached: a launch aims at 8 192 segments in all)
    }
    if (hist_out) *hist_out = hist;
    if (seg_len) *seg_len = sl;
    if (kmax) *kmax = (max_len + sl - 1u) / sl;
    return 1;
}
size_t alz_encode_seg_bytes(uint32_t count, uint32_t kmax, uint32_t hist) { return (size_t)count * kmax * (sizeof(SegRec) + (3u + hist) * sizeof(u32)) + (size_t)count * 16u + 64u; }   // records, totals; synchronisation points, direct exits, entries, exit tables
namespace {

// ---- kernel A for a batch that does not even fill the CUs with a workgroup per buffer (at most 128 buffers): over overlapping SEGMENTS of the buffers, as
// alz_encode_big.h does it for ONE buffer.  Segment j of a buffer is hashed from W = maxDistance (whole windows) in front of j * SA on: a link that would
// reach in front of the warm-up is longer than maxDistance, which ends a chain walk exactly as no link does (LzChainMatchFinder.cs:259-260; the narrowing
// of 15-bit chains stops there too).  The segments are "virtual streams" of enc_prev_cu_kernel (descriptors written on the device); the 16-bit links --
// distances, so they need no rebasing -- are gathered into the buffer's own array.  One workgroup per buffer took 1.55 us per KiB of the LONGEST buffer.
__global__ __launch_bounds__(256) void enc_aseg_setup_kernel(const alz_stream* __restrict__ streams, const u32* __restrict__ index_list, u32 count, alz_stream* __restrict__ vs,
                                                             u32* __restrict__ vindex, u64* __restrict__ vpos, u32 ka, u32 SA, u32 W, u32 stride, int tail_skip) {
    const u32 t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count * ka) return;
    const u32 bid = t / ka, j = t % ka;
    const u32 sid0 = index_list[bid];
    if (sid0 == 0xFFFFFFFFu) { vindex[t] = 0xFFFFFFFFu; return; }           // (a list written on the device -- enc_words_kernel: the streams whose links are narrowed -- with unused slots)
    alz_stream s = streams[sid0];
    const int limit = (int)s.src_len - tail_skip - 4;                      // (tail_skip: the bytes an LZ4 block keeps back -- the last segment ends in front of them)
    const u32 first = j * SA;
    if ((int)first > limit) { vindex[t] = 0xFFFFFFFFu; return; }          // (no such segment: kernel A leaves the slot alone)
    const u32 start = first >= W ? first - W : 0u;
    u32 end = first + SA; if (end > (u32)limit + 1u) end = (u32)limit + 1u;
    s.src_off += start;
    s.src_len = end - start + 3u;                                          // the last hashed position is end - 1: four bytes
    vs[t] = s; vindex[t] = t; vpos[t] = (u64)t * stride;
}
__global__ __launch_bounds__(256) void enc_aseg_gather_kernel(const alz_stream* __restrict__ streams, const u32* __restrict__ index_list, const int* __restrict__ seg4,
                                                              int* __restrict__ fin4, const u64* __restrict__ pos_off, u32 ka, u32 SA, u32 W, u32 stride, int tail_skip) {
    const u32 p = blockIdx.x * 256u + threadIdx.x, bid = blockIdx.y;
    const u32 sid = index_list[bid];
    if (sid == 0xFFFFFFFFu) return;
    const int limit = (int)streams[sid].src_len - tail_skip - 4;
    if ((int)p > limit) return;
    const u32 j = p / SA, first = j * SA, start = first >= W ? first - W : 0u, local = p - start;
    reinterpret_cast<unsigned short*>(fin4 + pos_off[sid])[p] = reinterpret_cast<const unsigned short*>(seg4 + (size_t)(bid * ka + j) * stride)[local];
}
#ifndef ALZ_ASEG_MAX_STREAMS
#define ALZ_ASEG_MAX_STREAMS 128u
#endif
}  // namespace
// 1: kernel A of such a launch runs over segments; *bytes: its scratch (behind alz_encode_seg_bytes in the same allocation)
int alz_encode_aseg(const void* geom, uint32_t count, uint32_t max_len, uint32_t* sa_out, uint32_t* ka_out, uint32_t* w_out, uint32_t* stride_out, size_t* bytes) {
    EncGeom g; memcpy(&g, geom, sizeof(g));
    if (bytes) *bytes = 0;
    const u32 W = ((u32)g.max_dist + 63u) & ~63u;
    // (16-bit links, no min-length table -- quality < 10 --, and a kernel A of ONE pass at 15 bits: the finder's own 15 bits, or narrowed afterwards in windows up to 8 KiB)
    // (round 6: the 64 KiB windows too -- LZ4 blocks, LZO, raw Snappy above quality 0 --, whose launch narrows the streams of ONE of enc_words_kernel's two lists: unused slots of the list stay out)
    const bool plain = g.hash_bits == 15, narrowed = g.link16 && g.nprops <= 1 && g.hash_bits > 15;
    if (!g.link16 || g.use_min_table || !(plain || narrowed) || count == 0 || count > ALZ_ASEG_MAX_STREAMS || max_len < 4u * W) return 0;
    uint64_t sa = ((uint64_t)count * max_len + 511u) / 512u;              // ~512 workgroups: two per CU
    if (sa < 2u * W) sa = 2u * W;
    const u32 SA = (u32)((sa + 63u) & ~(uint64_t)63u), ka = (max_len + SA - 1u) / SA, stride = (SA + W) / 2u + 16u;
    const uint64_t b = (uint64_t)count * ka * ((uint64_t)stride * 4u + sizeof(alz_stream) + 4u + 8u) + 256u;
    if (b > (1ull << 30)) return 0;
    if (sa_out) *sa_out = SA; if (ka_out) *ka_out = ka; if (w_out) *w_out = W; if (stride_out) *stride_out = stride;
    if (bytes) *bytes = (size_t)b;
    return 1;
}
namespace {

// the parse of a launch on this path: synchronisation points, exits, the cursor that enters every segment, the walk -- the start mask is complete behind it
static void launch_seg_walk(hipStream_t s, u32 count, const u8* src, const alz_stream* streams, const u32* index, mentry* match, const u64* pos_off,
                            const int* prev4, const int* prevm, u64* mask, u32* sync, u32 seglen, u32 kmax, const EncGeom& g) {
    const u32 hist = seg_table_hist(g);
    u32* direct = sync + (size_t)count * kmax;
    u32* entry = direct + (size_t)count * kmax;
    u32* ftab = entry + (size_t)count * kmax;
    hipLaunchKernelGGL(enc_sync_kernel, dim3(kmax, count), dim3(64), 0, s, streams, index, (const mentry*)match, pos_off, sync, kmax, seglen, g);
    hipLaunchKernelGGL(enc_exit_kernel, dim3(kmax, count), dim3(64), 0, s, streams, index, (const mentry*)match, pos_off, (const u32*)sync, direct, ftab, kmax, seglen, hist, g);
    hipLaunchKernelGGL(enc_compose_kernel, dim3(count), dim3(64), 0, s, (const u32*)direct, (const u32*)ftab, entry, kmax, seglen, hist);
    hipLaunchKernelGGL((enc_roles_kernel<true>), dim3(kmax, count), dim3(64), 0, s, src, streams, index, count, match, pos_off, prev4, prevm, mask, g, 0, (const u32*)entry, kmax);
}

template <int FMT>
static void launch_emit_seg(hipStream_t s, u32 count, const u8* src, u8* dst, const alz_stream* streams, const u32* index, mentry* match, const u64* pos_off,
                            const int* prev4, const int* prevm, u64* mask, void* d_seg, u32 seglen, u32 kmax, alz_result* results, alz_encode_aux* aux, const EncGeom& g) {
    SegRec* seg = (SegRec*)d_seg;
    u32* stot = (u32*)((u8*)d_seg + (size_t)count * kmax * sizeof(SegRec));
    u32* sync = stot + 4 * (size_t)count;
    launch_seg_walk(s, count, src, streams, index, match, pos_off, prev4, prevm, mask, sync, seglen, kmax, g);
    hipLaunchKernelGGL((enc_seg_kernel<FMT, false>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, (const u32*)stot, kmax, seglen, g);
    hipLaunchKernelGGL(enc_seg_prefix_kernel, dim3(count), dim3(64), 0, s, streams, index, seg, stot, kmax, seglen);
    hipLaunchKernelGGL((enc_seg_kernel<FMT, true>), dim3(kmax, count), dim3(64), 0, s, src, dst, streams, index, (const mentry*)match, pos_off, (const u64*)mask, seg, (const u32*)stot, kmax, seglen, g);
    hipLaunchKernelGGL((enc_seg_flags_kernel<FMT>), dim3(count), dim3(64), 0, s, dst, streams, index, (const SegRec*)seg, (const u32*)stot, kmax, seglen, results, aux);
}
