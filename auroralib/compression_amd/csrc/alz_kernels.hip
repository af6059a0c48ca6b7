// alz_kernels.hip -- gfx950 decode kernels + launch wrappers (one launch per format present in a plan).
//
// Grid mapping: one 64-thread workgroup (= one wavefront) per stream.  A batch of
// 10 000 streams therefore launches 10 000 workgroups >> 256 CUs; consecutive
// workgroups land on different XCDs round-robin and each stream's input, LDS
// window and output are private to its wave, so there is no inter-workgroup
// traffic and no L2 sharing to arrange.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "alz_decode_fast.h"
#include "alz_internal.h"

#define ALZ_INCACHE_BYTES (2048 + 32)
#define ALZ_INCACHE_SMALL (512 + 32)     /* 256-byte chunks: lane-parallel flag-family kernels */
#ifndef ALZ_FAST_ATTR
#define ALZ_FAST_ATTR
#endif
#ifndef ALZ_WPB
#define ALZ_WPB 1                        /* waves (= streams) per workgroup of the lane-parallel kernels */
#endif
#ifndef ALZ_PRS_LW
#define ALZ_PRS_LW 8192
#endif
#ifndef ALZ_QUEUE_LW
#define ALZ_QUEUE_LW 4096
#endif

template <bool FB>
__device__ __forceinline__ void write_result(alz_result* r, int lane, const OutWin<FB>& out, u32 src_used, int status, u32 src_len, u32 hist = 0) {
    // src_used = source.Position after the call.  INPUT_TRUNCATED: the reader ran into the end of the input -- Position is there (where
    // exactly a multi-byte read gave up is not something a caller can use; include/auroralz.h fixes it as src_len)
    if (status == ALZ_ST_INPUT_TRUNCATED) src_used = src_len;
    if (lane == 0) { r->dst_len = out.produced - hist; r->src_used = src_used; r->status = status; r->reserved = 0; }
}

// ------------------------------------------------------------------------------------------------
// Generic exact kernel: FMT selects the parser at compile time.
template <int FMT, bool FB>
__global__ __launch_bounds__(64) void alz_decode_serial_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base,
                                                               const alz_stream* __restrict__ streams,
                                                               const u32* __restrict__ index_list, u32 count,
                                                               alz_result* __restrict__ results, alz_lz_properties lz, u32 lw) {
    extern __shared__ uint4 smem[];
    u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list ? index_list[bid] : bid;
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    u8* dst = dst_base + st.dst_off;
    const u32 src_len = uni(st.src_len), size = uni(st.decom_len);
    u32 cap = uni(st.dst_cap);
    // LZ4 block with history (alz_stream.aux0): the stream continues the window of earlier blocks of its frame
    const u32 hist = (FMT == ALZ_FMT_LZ4_BLOCK) ? uni(st.aux0) : 0u;
    dst -= hist; cap += hist;

    u8* lds = reinterpret_cast<u8*>(smem);
    OutWin<FB> out; out.init(dst, cap, lds, lw, lane);
    if (hist) out.preload(hist);
    u8* inc_lds = lds + lw;
    InCache in; in.init(src, src_len, inc_lds, lane);
    DecState s; dec_state_init(s);
    bool has_size = false; u32 used = 0; bool used_set = false;
    typedef DirectSink<OutWin<FB>> SK;
    SK sk(out, s);

    if constexpr (FMT == ALZ_FMT_LZSS) {
        has_size = true;
        u32 W = 1u << lz.window_bits;
        dec_lzss_serial(in, sk, s, src_len, size, lz.length_bits, lz.min_length, lz.windows_start, lz.max_distance, W);
    } else if constexpr (FMT == ALZ_FMT_LZ10) {
        has_size = true; dec_lz1x_serial<SK, false>(in, sk, s, src_len, size);
    } else if constexpr (FMT == ALZ_FMT_LZ11) {
        has_size = true; dec_lz1x_serial<SK, true>(in, sk, s, src_len, size);
    } else if constexpr (FMT == ALZ_FMT_CLZ0) {
        has_size = true; dec_clz0_serial(in, sk, s, src_len, size);
    } else if constexpr (FMT == ALZ_FMT_LZ40) {
        has_size = true; dec_lz40_serial(in, sk, s, src_len, size);
    } else if constexpr (FMT == ALZ_FMT_LZHUDSON) {
        has_size = true; dec_lzhudson_serial(in, sk, s, src_len, size);
    } else if constexpr (FMT == ALZ_FMT_SMSR00) {
        has_size = true;
        const u32 a0 = uni(st.aux0);
        if (a0 > src_len) s.eof = true;                                   // ReadExactly(buffer, 0, codesLength) throws  SMSR00.cs:76
        else {
            InCache uin;
            uin.init(src, src_len, inc_lds + ALZ_INCACHE_BYTES, lane); uin.seek(a0 < src_len ? a0 : 0);
            dec_smsr00_serial(in, uin, sk, s, src_len, size, a0, used);
            used_set = true;
        }
    } else if constexpr (FMT == ALZ_FMT_YAZ0) {
        has_size = true; dec_yaz0_serial(in, sk, s, src_len, size);
    } else if constexpr (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0) {
        has_size = true;
        const u32 a0 = uni(st.aux0), a1 = uni(st.aux1);
        if (FMT == ALZ_FMT_YAY0 && (a0 > src_len || a1 > src_len)) s.eof = true;   // Slice() throws  Yay0.cs:102-103
        else {
            InCache cin, uin;
            cin.init(src, src_len, inc_lds + ALZ_INCACHE_BYTES, lane); cin.seek(a0 < src_len ? a0 : 0);
            uin.init(src, src_len, inc_lds + 2 * ALZ_INCACHE_BYTES, lane); uin.seek(a1 < src_len ? a1 : 0);
            used = dec_3cursor_serial<SK, FMT == ALZ_FMT_MIO0>(in, cin, uin, sk, s, src_len, size, 0, a0, a1);
            used_set = true;
        }
    } else if constexpr (FMT == ALZ_FMT_PRS_BE) {
        dec_prs_serial<SK, true>(in, sk, s, src_len);
    } else if constexpr (FMT == ALZ_FMT_PRS_LE) {
        dec_prs_serial<SK, false>(in, sk, s, src_len);
    } else if constexpr (FMT == ALZ_FMT_LZ4_BLOCK) {
        dec_lz4_serial(in, sk, s, src_len);
    } else if constexpr (FMT == ALZ_FMT_LZO) {
        LzoState ls; lzo_state_init(ls);
        dec_lzo_serial(in, sk, s, src_len, ls);
    } else if constexpr (FMT == ALZ_FMT_SNAPPY_RAW) {
        u32 sz = 0; bool have = false;
        dec_snappy_serial(in, sk, s, src_len, sz, have);
    } else if constexpr (FMT == ALZ_FMT_FASTLZ) {
        FastlzState fz; fastlz_state_init(fz);
        dec_fastlz_serial(in, sk, s, src_len, fz);
    } else if constexpr (FMT == ALZ_FMT_CNX2) {
        has_size = true; dec_cnx2_serial(in, sk, s, src_len, size);
    } else if constexpr (FMT == ALZ_FMT_HIG) {
        has_size = true; dec_hig_serial(in, sk, s, src_len, size);
    } else if constexpr (FMT == ALZ_FMT_LZSHREK) {
        has_size = true; dec_lzshrek_serial(in, sk, s, src_len);
    } else if constexpr (FMT == ALZ_FMT_WFLZ || FMT == ALZ_FMT_WFLZ_BE) {
        dec_wflz_serial<SK, FMT == ALZ_FMT_WFLZ_BE>(in, sk, s, src_len);
    } else if constexpr (FMT == ALZ_FMT_REFPACK) {
        has_size = true; dec_refpack_serial(in, sk, s, src_len);
    } else if constexpr (FMT == ALZ_FMT_LZ02) {
        has_size = true; dec_lz02_serial(in, sk, s, src_len);
    } else if constexpr (FMT == ALZ_FMT_CNS) {
        has_size = true; dec_cns_serial(in, sk, s, src_len, size);
    } else if constexpr (FMT == ALZ_FMT_BLZ) {
        has_size = true; dec_blz_serial(in, sk, s, src_len, size < cap ? size : cap);
    }
    out.finish();
    int status = resolve_status(s, has_size, out.produced, size, cap);
    if (FMT == ALZ_FMT_BLZ && status == ALZ_ST_OK && out.produced != size) status = ALZ_ST_OUTPUT_SIZE_MISMATCH;   // the span must be full  BLZ.cs:131
    write_result(&results[sid], lane, out, used_set ? used : s.p, status, src_len, hist);
}

// ------------------------------------------------------------------------------------------------
// Lane-parallel kernel of the flag-byte family (alz_decode_fast.h); the exact serial parser finishes the tail.
// FBK: the format's window is longer than the LDS ring (LZSS with 14..16 window bits): sources older than the ring come back
// from the stream's own output in HBM, through the chunked byte phase, as in the 64 KiB queue kernels
template <int FMT, int LWMAX = 4096, bool FBK = false>
__global__ __launch_bounds__(64 * ALZ_WPB) ALZ_FAST_ATTR void alz_decode_fast_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base,
                                                             const alz_stream* __restrict__ streams,
                                                             const u32* __restrict__ index_list, u32 count,
                                                             alz_result* __restrict__ results, alz_lz_properties lz, u32 lw, const u32* __restrict__ gate) {
    if (gate != nullptr && __builtin_nontemporal_load(gate) == 0u) return;   // (alz_launch_decode_gated: nothing to do unless the word is set)
    constexpr bool SMSR = (FMT == ALZ_FMT_SMSR00);                    // code stream + literal stream
    constexpr bool THREE = (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0 || SMSR);   // (several input cursors: small caches)
    constexpr int NC = SMSR ? 2 : (THREE ? 3 : 1);
    // static LDS (absolute addresses fold into the DS instructions' offset fields): marks | input caches | window
    // (the waves of a workgroup never interact: several streams share a workgroup only because a CU holds more waves than
    // single-wave workgroups)
    // input caches: 1 KiB chunks where there is one (16 B/lane loads), 256 B chunks for the three-cursor formats -- with three
    // big caches a CU would hold 15 waves instead of 24 (measured: Yay0 585 -> 758, MIO0 379 -> 481 GiB/s)
#ifndef ALZ_FAST_CHUNK
#define ALZ_FAST_CHUNK 1024
#endif
#ifndef ALZ_DESCTAB_ALL
#define ALZ_DESCTAB_ALL 0       // (experiment: the descriptor table for the single-cursor kernels too)
#endif
    constexpr u32 CHUNK = THREE ? 256u : ((LWMAX > 4096 || alz_tab512<FMT>::value) ? 512u : (u32)ALZ_FAST_CHUNK);      // (8 KiB windows: 17 instead of 15 waves per CU)
    constexpr u32 CACHE = THREE ? ALZ_INCACHE_SMALL : 2u * CHUNK + 32u;
    constexpr u32 FSCR = FBK ? ALZ_EMIT_SCRATCH : ((THREE || alz_tab512<FMT>::value) ? ALZ_BYTE_SCRATCH : 128u), FSLACK = FBK ? ALZ_WIN_SLACK : 0u;   // (chunked byte phase: token table + ring mirror)
    __shared__ __attribute__((aligned(16))) u8 lds_all[ALZ_WPB][FSCR + NC * CACHE + LWMAX + FSLACK];
    const u32 wid = ALZ_WPB == 1 ? 0u : (u32)threadIdx.x >> 6;   // (constant 0: LDS addresses stay immediates)
    u8* const lds = lds_all[wid];
    u32 bid = blockIdx.x * ALZ_WPB + wid;
    if (bid >= count) return;
    const int lane = (int)(threadIdx.x & 63u);
    const u32 sid = index_list ? index_list[bid] : bid;
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    u8* dst = dst_base + st.dst_off;
    const u32 src_len = uni(st.src_len), cap = uni(st.dst_cap), size = uni(st.decom_len);
    u8* segmark = lds;
    u8* inc_lds = lds + FSCR;
    typedef OutWin<FBK> OWF;
    OWF out; out.init(dst, cap, lds + FSCR + NC * CACHE, FBK ? (u32)LWMAX : lw, lane, FSLACK);
    segmark[lane] = 0; segmark[64 + lane] = 0;
    InCache in; in.init(src, src_len, inc_lds, lane, CHUNK);
    DecState s; dec_state_init(s);
    u32 used = 0; bool used_set = false;
    bool fin = false;

    if constexpr (FMT == ALZ_FMT_BLZ) {
        // exact parser while a match could still point beyond the span (an error here, not zeros) and for the end of the
        // stream (matches cut at the span end, the span-must-be-full rule); LZ10's lane-parallel loop in between, kept away
        // from the span end by more than one iteration can produce (64 tokens x 18 bytes)
        typedef DirectSink<OWF> SK; SK sk(out, s);
        const u32 L = size < cap ? size : cap;
        dec_blz_serial(in, sk, s, src_len, L, 4098u);
        if (!s.eof && !s.bad && !s.ovf) {
            FastGeom gm; gm.length_bits = 4; gm.min_length = 3; gm.windows_start = 0; gm.max_distance = 4096; gm.W = 4096;
            bool to_serial = false;
            while (!to_serial && (u64)out.produced + 1152u < L && s.p < src_len) (void)fast_iter_interleaved<FMT>(in, out, s, L, src_len, to_serial, segmark, lane, gm);
            dec_blz_serial(in, sk, s, src_len, L);
        }
    } else if constexpr (FMT == ALZ_FMT_LZ02) {
        // no declared size inside the loop: the stream runs to its terminator (the size is compared there); only the
        // capacity bounds the lane-parallel iterations, and an exactly full destination still has to see the terminator
        FastGeom gm; gm.length_bits = 4; gm.min_length = 3; gm.windows_start = 0; gm.max_distance = 4096; gm.W = 4096;
        bool to_serial = false;
        while (!s.ovf && !to_serial && out.produced < cap && s.p < src_len) (void)fast_iter_interleaved<FMT>(in, out, s, cap, src_len, to_serial, segmark, lane, gm);
        if (!s.ovf) { typedef DirectSink<OWF> SK; SK sk(out, s); dec_lz02_serial(in, sk, s, src_len); }
    } else if constexpr (FMT == ALZ_FMT_LZHUDSON) {
        while (!fin && out.produced < size && (u64)s.p + 128u <= src_len) fin = fast_iter_lzhudson(in, out, s, size, segmark, lane);
        if (!fin) { typedef DirectSink<OWF> SK; SK sk(out, s); dec_lzhudson_serial(in, sk, s, src_len, size); }
    } else if constexpr (!THREE) {
        FastGeom gm; gm.length_bits = lz.length_bits; gm.min_length = lz.min_length; gm.windows_start = lz.windows_start;
        gm.max_distance = lz.max_distance; gm.W = 1u << lz.window_bits;
        bool to_serial = false;   // the fast loop runs to the last complete token of the input; the exact parser finishes
        while (!fin && !to_serial && out.produced < size && s.p < src_len) fin = fast_iter_interleaved<FMT>(in, out, s, size, src_len, to_serial, segmark, lane, gm);
        if (!fin) {
            typedef DirectSink<OWF> SK;
            SK sk(out, s);
            if constexpr (FMT == ALZ_FMT_LZSS) dec_lzss_serial(in, sk, s, src_len, size, lz.length_bits, lz.min_length, lz.windows_start, lz.max_distance, gm.W);
            else if constexpr (FMT == ALZ_FMT_LZ10) dec_lz1x_serial<SK, false>(in, sk, s, src_len, size);
            else if constexpr (FMT == ALZ_FMT_LZ11) dec_lz1x_serial<SK, true>(in, sk, s, src_len, size);
            else if constexpr (FMT == ALZ_FMT_LZ40) dec_lz40_serial(in, sk, s, src_len, size);
            else if constexpr (FMT == ALZ_FMT_CLZ0) dec_clz0_serial(in, sk, s, src_len, size);
            else dec_yaz0_serial(in, sk, s, src_len, size);
        }
    } else if constexpr (SMSR) {
        const u32 a0 = uni(st.aux0);                                       // length of the code section
        if (a0 > src_len) s.eof = true;                                    // ReadExactly(buffer, 0, codesLength) throws  SMSR00.cs:76
        else {
            InCache uin;
            uin.init(src, src_len, inc_lds + CACHE, lane, CHUNK); uin.seek(a0 < src_len ? a0 : 0);
            u32 cp = 0, up = a0;
            while (!fin && out.produced < size && (u64)cp + 136u <= a0 && (u64)up + 64u <= src_len)
                fin = fast_iter_smsr00(in, uin, out, s, size, segmark, lane, cp, up);
            used = up;
            if (!fin) { typedef DirectSink<OWF> SK; SK sk(out, s); dec_smsr00_serial(in, uin, sk, s, src_len, size, a0, used, cp, up); }
            used_set = true;
        }
    } else {
        const u32 a0 = uni(st.aux0), a1 = uni(st.aux1);
        if (FMT == ALZ_FMT_YAY0 && (a0 > src_len || a1 > src_len)) s.eof = true;   // Slice() throws  Yay0.cs:102-103
        else {
            InCache cin, uin;
            cin.init(src, src_len, inc_lds + CACHE, lane, CHUNK); cin.seek(a0 < src_len ? a0 : 0);
            uin.init(src, src_len, inc_lds + 2 * CACHE, lane, CHUNK); uin.seek(a1 < src_len ? a1 : 0);
            u32 fp = 0, cp = a0, up = a1;
            while (!fin && out.produced < size && fp + 8u <= src_len && (u64)cp + 128u <= src_len && (u64)up + 64u <= src_len)
                fin = fast_iter_3cursor<FMT == ALZ_FMT_MIO0>(in, cin, uin, out, s, size, segmark, lane, fp, cp, up);
            used = cp > up ? cp : up;
            if (!fin) { typedef DirectSink<OWF> SK; SK sk(out, s); used = dec_3cursor_serial<SK, FMT == ALZ_FMT_MIO0>(in, cin, uin, sk, s, src_len, size, fp, cp, up); }
            used_set = true;
        }
    }
    out.finish();
    int status = resolve_status(s, true, out.produced, size, cap);
    if (FMT == ALZ_FMT_BLZ && status == ALZ_ST_OK && out.produced != size) status = ALZ_ST_OUTPUT_SIZE_MISMATCH;   // the span must be full  BLZ.cs:131
    write_result(&results[sid], lane, out, used_set ? used : s.p, status, src_len);
}

// ------------------------------------------------------------------------------------------------
// The same decode as a WORK QUEUE of chunks (round 5; tickets, epochs and the in-order repair: round 6).  With one wavefront per stream a batch of more streams than the GPU holds
// wavefronts ends in a partly filled round: 10 000 streams on 6 400 places = one full round and 3 600 streams at 14 waves per CU, where a
// wave is bound by its own issue rate -- 833 GiB/s for one batch against 1 004 with a second batch filling the tail.  Here a stream is
// decoded in CHUNKS of ALZ_CHUNK_BYTES of output and the launch is a queue of (stream, chunk) items -- eight sub-queues, chunk-major inside each --, one workgroup
// per item, each drawing its item as a ticket when it starts (queue_ticket): every place stays busy until the queue is empty, and what is left at the end is one chunk's
// latency, not one stream's.  A chunk ends at the first iteration boundary at or behind its limit (the lane-parallel loop consumes whole
// flag groups: the state between two iterations is the input offset -- three of them for Yay0 / MIO0 -- and the output position); the wave
// that decoded it flushes its output, hands the LDS window and that state to whoever holds the stream's next chunk -- through a slot of its own
// in global memory, written with write-through (sc1) stores, drained, then ONE flag carrying the launch's epoch (MI355X_MICROARCH.md, inter-workgroup visibility; the
// slot is written once per launch, so no L2 can hold an older copy of it) -- and ends.  The workgroup of the next chunk
// polls that flag (relaxed, one lane, bounded), acquires once, loads the window.  Tickets of a sub-queue go out in order to workgroups that have started, so the chunk an item waits
// for is in the hands of a running workgroup that waits, if at all, for a still earlier item.  The stream's last chunk runs the exact parser over the tail and writes the result; a stream that ends early (an error, a
// terminator, E5) marks its remaining boundaries "ended" as their items come up.  A bounded spin that runs out (a fault: never seen) sets the plan's sticky word `tmo`; the gated
// launch the host enqueues behind this kernel then decodes the format's streams again with one wavefront per stream, in stream order (alz_plan_execute).
#define ALZ_CHUNK_BYTES ALZ_CHUNK_OUT     /* (alz_internal.h: the host cuts the streams by the same number) */
#ifndef ALZ_CHUNK_SPINS
#define ALZ_CHUNK_SPINS (1u << 20)    /* x (sleep + one load): about a second.  (-DALZ_CHUNK_SPINS=0: every wait that is not over at once runs out -- the test of the repeat path, tools/variants/README.md) */
#endif
#define ALZ_CHUNK_FLAG_WORDS 32u       /* a flag has a 128-byte line of its own: pollers of one boundary never touch the line another boundary's flag, the queue head or the timeout word lives in */
typedef __attribute__((address_space(1))) u32 alz_gu32;
typedef __attribute__((address_space(1))) unsigned long long alz_gu64;
__device__ __forceinline__ void chunk_store32(void* p, u32 v) { __hip_atomic_store(reinterpret_cast<u32*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 chunk_load32(const void* p) { return __hip_atomic_load(reinterpret_cast<const u32*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The workgroup's item.  The queue is ALZ_QUEUE_SHARDS sub-queues (a stream's chunks all in one of them, chunk-major inside it), each with its own head word on its own
// 128-byte line: ONE head word serves ~88 returning device-scope atomics per microsecond (MI355X_MICROARCH.md, "dequeue"), so the ~6 400 workgroups a launch starts with
// spent the first ~70 us queueing for their tickets (measured: the Yaz0 launch 2.509 -> 2.574 ms with one head), eight heads serve them in ~9.  A workgroup prefers the
// sub-queue of the XCD it runs on (HW_REG_XCC_ID: workgroups are dealt round-robin over the XCDs, so the eight heads are drawn from at the same rate) and goes round the
// others when that one is exhausted -- preference and placement are for speed only.  What correctness rests on: inside a sub-queue tickets are handed out in order to
// workgroups that HAVE started, and the chunk an item waits for is an earlier item of the SAME sub-queue; so the waiting item with the smallest ticket of its sub-queue
// always waits for a running (or finished) workgroup, which waits, if at all, for a still smaller ticket: no cycle, whatever the dispatch order.
// Every workgroup draws at most one ticket per sub-queue and the launch has exactly as many workgroups as items, so every item is taken.
// Nothing of the queue is zeroed between launches (round 6: a memset in front of every launch was a third serialised operation per execute): the launch carries the
// plan's EPOCH -- a hand-over flag counts only when it holds this launch's epoch (queue_flag_* below) --, and the heads exist twice: launch e draws from set e & 1, and
// the workgroup that draws ticket 0 of a sub-queue zeroes that sub-queue's head in the OTHER set, which the launch before left behind and the next one will draw from
// (launches of one plan never overlap: alz_plan_execute orders them by an event).
__device__ __forceinline__ u32 queue_ticket(u32* ctl, u32 epoch, const alz_queue_bounds& qb, int lane) {
    const u32 xcc = (u32)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & (ALZ_QUEUE_SHARDS - 1u);    // hwreg(HW_REG_XCC_ID, 0, 4)
    u32* const head = ctl + (epoch & 1u) * (ALZ_CHUNK_CTL_WORDS / 2u);
    u32* const other = ctl + ((epoch & 1u) ^ 1u) * (ALZ_CHUNK_CTL_WORDS / 2u);
    u32 item = 0xFFFFFFFFu;
#pragma unroll
    for (u32 j = 0; j < ALZ_QUEUE_SHARDS; j++) {
        if (item == 0xFFFFFFFFu) {                               // (wave-uniform: `item` comes out of readfirstlane)
            const u32 q = (xcc + j) & (ALZ_QUEUE_SHARDS - 1u);
            const u32 lo = qb.off[q], hi = qb.off[q + 1u];
            if (lo < hi) {
                u32 t = 0;
                if (lane == 0) t = __hip_atomic_fetch_add(head + 32u * q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t = uni(t);
                if (t == 0u && lane == 0) chunk_store32(other + 32u * q, 0u);
                if (t < hi - lo) item = lo + t;
            }
        }
    }
    return item;
}
// A hand-over flag: (epoch << 2) | state, state 1 handed over, 2 the stream has ended, 3 a wait ran out.  A word of another epoch (0: never written) is "not yet".
__device__ __forceinline__ u32 queue_flag_wait(const u32* flag, u32 epoch, int lane) {
    u32 f = 0;
    if (lane == 0) {
        u32 spins = 0;
        for (;;) {
            const u32 w = chunk_load32(flag);
            if ((w >> 2) == epoch) { f = w & 3u; break; }
            if (++spins > ALZ_CHUNK_SPINS) break;
            __builtin_amdgcn_s_sleep(16);
        }
    }
    return uni(f);
}

// One item of the queue: set up from the hand-over slot (or from nothing), decode to the chunk's limit, flush, hand over or finish.
// Returns the flag the item leaves in its slot: 1 handed over (window + cursors stored write-through into `out_slot`, NOT yet drained),
// 2 the stream ended here (result written), 0 the stream's last chunk (result written).
template <int FMT>
__device__ __forceinline__ u32 fastq_item(const u8* src_base, u8* dst_base, const alz_stream* streams, alz_result* results, const alz_lz_properties& lz,
                                          u8* lds, u32 lw, u32 sid, u32 c, u32 last, const u8* in_slot, u8* out_slot, u32 p0, u32 p1, u32 p2, u32 start) {
    constexpr bool THREE = (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0);
    constexpr int NC = THREE ? 3 : 1;
    constexpr u32 CHUNK = THREE ? 256u : (alz_tab512<FMT>::value ? 512u : (u32)ALZ_FAST_CHUNK);
    constexpr u32 CACHE = THREE ? ALZ_INCACHE_SMALL : 2u * CHUNK + 32u;
    constexpr u32 FSCR = (THREE || alz_tab512<FMT>::value) ? ALZ_BYTE_SCRATCH : 128u;
    const int lane = lane_id();
    u8* segmark = lds;
    u8* inc_lds = lds + FSCR;
    u8* const win = lds + FSCR + NC * CACHE;
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    u8* dst = dst_base + st.dst_off;
    const u32 src_len = uni(st.src_len), cap = uni(st.dst_cap), size = uni(st.decom_len);
    // ---- set up: window (zeros, or what the chunk before left), marks, input caches at the hand-over offsets
    typedef OutWin<false> OWF;
    OWF out;
    if (c == 0u) out.init(dst, cap, win, lw, lane, 0u);
    else {
        out.dst = dst; out.cap = cap; out.win = win; out.lw_mask = lw - 1u; out.lane = lane;
        out.fl = lw >= 4096u ? 1024u : (lw >> 2);
        out.oshift = (u32)(reinterpret_cast<uintptr_t>(dst) & 15u);
        out.produced = start; out.flushed = start; out.slack_dirty = false; out.mtag = 0;
        for (u32 i = 16u * (u32)lane; i < lw; i += 1024u) *reinterpret_cast<uint4*>(win + i) = *reinterpret_cast<const uint4*>(in_slot + 32 + i);
        wave_sync();
    }
    segmark[lane] = 0; segmark[64 + lane] = 0;
    InCache in; in.init_at(src, src_len, inc_lds, lane, CHUNK, c > 0u ? p0 : 0u);
    DecState s; dec_state_init(s);
    const u32 limit = last ? 0xFFFFFFFFu : (c + 1u) * ALZ_CHUNK_BYTES;     // (the last chunk runs to the end of the stream)
    u32 used = 0; bool used_set = false;
    bool fin = false, handoff = false;
    if constexpr (!THREE) {
        if (c > 0u) s.p = p0;
        FastGeom gm; gm.length_bits = lz.length_bits; gm.min_length = lz.min_length; gm.windows_start = lz.windows_start;
        gm.max_distance = lz.max_distance; gm.W = 1u << lz.window_bits;
        bool to_serial = false;
        while (!fin && !to_serial && out.produced < size && s.p < src_len && out.produced < limit) fin = fast_iter_interleaved<FMT>(in, out, s, size, src_len, to_serial, segmark, lane, gm);
        handoff = !fin && !to_serial && out.produced < size && s.p < src_len;       // (then the loop stopped at the chunk's limit)
        if (!handoff && !fin) {
            typedef DirectSink<OWF> SK;
            SK sk(out, s);
            if constexpr (FMT == ALZ_FMT_LZSS) dec_lzss_serial(in, sk, s, src_len, size, lz.length_bits, lz.min_length, lz.windows_start, lz.max_distance, gm.W);
            else if constexpr (FMT == ALZ_FMT_LZ10) dec_lz1x_serial<SK, false>(in, sk, s, src_len, size);
            else if constexpr (FMT == ALZ_FMT_LZ11) dec_lz1x_serial<SK, true>(in, sk, s, src_len, size);
            else if constexpr (FMT == ALZ_FMT_LZ40) dec_lz40_serial(in, sk, s, src_len, size);
            else if constexpr (FMT == ALZ_FMT_CLZ0) dec_clz0_serial(in, sk, s, src_len, size);
            else dec_yaz0_serial(in, sk, s, src_len, size);
        }
        p0 = s.p;
    } else {
        const u32 a0 = uni(st.aux0), a1 = uni(st.aux1);
        if (FMT == ALZ_FMT_YAY0 && (a0 > src_len || a1 > src_len)) s.eof = true;   // Slice() throws  Yay0.cs:102-103
        else {
            InCache cin, uin;
            u32 fp = 0, cp = a0, up = a1;
            if (c > 0u) { fp = p0; cp = p1; up = p2; }
            cin.init_at(src, src_len, inc_lds + CACHE, lane, CHUNK, cp < src_len ? cp : 0);
            uin.init_at(src, src_len, inc_lds + 2 * CACHE, lane, CHUNK, up < src_len ? up : 0);
            while (!fin && out.produced < size && fp + 8u <= src_len && (u64)cp + 128u <= src_len && (u64)up + 64u <= src_len && out.produced < limit)
                fin = fast_iter_3cursor<FMT == ALZ_FMT_MIO0>(in, cin, uin, out, s, size, segmark, lane, fp, cp, up);
            handoff = !fin && out.produced < size && fp + 8u <= src_len && (u64)cp + 128u <= src_len && (u64)up + 64u <= src_len;
            used = cp > up ? cp : up;
            if (!handoff && !fin) { typedef DirectSink<OWF> SK; SK sk(out, s); used = dec_3cursor_serial<SK, FMT == ALZ_FMT_MIO0>(in, cin, uin, sk, s, src_len, size, fp, cp, up); }
            used_set = true;
            p0 = fp; p1 = cp; p2 = up;
        }
    }
    out.finish();
    if (handoff) {
        // ---- hand the stream on: window + cursors into this chunk's slot (write-through; the caller drains and signals)
        wave_sync();
        for (u32 i = 16u * (u32)lane; i < lw; i += 1024u) {
            const uint4 v = *reinterpret_cast<const uint4*>(win + i);
            __hip_atomic_store(reinterpret_cast<u64*>(out_slot + 32 + i), ((u64)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<u64*>(out_slot + 40 + i), ((u64)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) { chunk_store32(out_slot, p0); chunk_store32(out_slot + 4, p1); chunk_store32(out_slot + 8, p2); chunk_store32(out_slot + 12, out.produced); }
        return 1u;
    }
    const int status = resolve_status(s, true, out.produced, size, cap);
    write_result(&results[sid], lane, out, used_set ? used : s.p, status, src_len);
    return last ? 0u : 2u;                                    // the stream ended here: its later chunks have nothing to do
}

template <int FMT>
__global__ __launch_bounds__(64) ALZ_FAST_ATTR void alz_decode_fastq_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base,
                                                             const alz_stream* __restrict__ streams, const alz_chunk_item* __restrict__ items, alz_queue_bounds qb,
                                                             alz_result* __restrict__ results, alz_lz_properties lz, u32 lw,
                                                             u32* __restrict__ head, u32* __restrict__ flags, u8* __restrict__ slots, u32* __restrict__ tmo, u32 epoch) {
    constexpr bool THREE = (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0);
    constexpr int NC = THREE ? 3 : 1;
    constexpr u32 LWMAX = 4096u;
    constexpr u32 CHUNK = THREE ? 256u : (alz_tab512<FMT>::value ? 512u : (u32)ALZ_FAST_CHUNK);
    constexpr u32 CACHE = THREE ? ALZ_INCACHE_SMALL : 2u * CHUNK + 32u;
    constexpr u32 FSCR = (THREE || alz_tab512<FMT>::value) ? ALZ_BYTE_SCRATCH : 128u;
    __shared__ __attribute__((aligned(16))) u8 lds[FSCR + NC * CACHE + LWMAX];
    const int lane = (int)(threadIdx.x & 63u);
    const u32 slot_bytes = 32u + lw;                         // a boundary's slot: 8 state words, then the window
    // ONE item per workgroup, taken as a TICKET when the workgroup starts (round 6; round 5 used item = blockIdx.x and relied on workgroups being dispatched in
    // index order, which HIP does not promise): queue_ticket draws the next number of a sub-queue, once per sub-queue at most, no loop around the decode.
    const u32 item = queue_ticket(head, epoch, qb, lane);
    if (item == 0xFFFFFFFFu) return;
    const alz_chunk_item it = items[item];
    const u32 sid = uni(it.sid), c = uni(it.chunk), oslot = uni(it.slot), last = uni(it.last);
    u32 p0 = 0, p1 = 0, p2 = 0, start = 0;
    u32 flagv = 0;                                        // what this item leaves in its slot's flag: 1 handed over, 2 the stream has ended, 3 a wait ran out
    bool run = true;
    const u8* in_slot = slots + (size_t)(oslot ? oslot - 1u : 0u) * slot_bytes;
    u8* out_slot = slots + (size_t)oslot * slot_bytes;
    if (c > 0u) {
        // ---- the chunk before this one: wait for its flag (ONE lane polls ONE word, relaxed), acquire once
        const u32 f = queue_flag_wait(flags + (size_t)(oslot - 1u) * ALZ_CHUNK_FLAG_WORDS, epoch, lane);
        if (f == 0u || f == 3u) {                         // never seen: the host repeats the launch the other way; whoever waits for THIS chunk does not wait long
            if (lane == 0) atomicOr(tmo, 1u);
            flagv = 3u; run = false;
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (f == 2u) { flagv = 2u; run = false; }     // the stream ended in an earlier chunk
            else {
                p0 = uni(chunk_load32(in_slot)); p1 = uni(chunk_load32(in_slot + 4)); p2 = uni(chunk_load32(in_slot + 8)); start = uni(chunk_load32(in_slot + 12));
                if (!last && start >= (c + 1u) * ALZ_CHUNK_BYTES) {
                    // one iteration of an earlier chunk went past this whole chunk (a long match): pass the state on as it is
                    for (u32 i = 16u * (u32)lane; i < lw; i += 1024u) {
                        const uint4 v = *reinterpret_cast<const uint4*>(in_slot + 32 + i);
                        __hip_atomic_store(reinterpret_cast<u64*>(out_slot + 32 + i), ((u64)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(reinterpret_cast<u64*>(out_slot + 40 + i), ((u64)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane == 0) { chunk_store32(out_slot, p0); chunk_store32(out_slot + 4, p1); chunk_store32(out_slot + 8, p2); chunk_store32(out_slot + 12, start); }
                    flagv = 1u; run = false;
                }
            }
        }
    }
    if (run) flagv = fastq_item<FMT>(src_base, dst_base, streams, results, lz, lds, lw, sid, c, last, in_slot, out_slot, p0, p1, p2, start);
    // ---- the item's ONE flag: every storing lane drains its write-through stores first, then one lane signals
    if (flagv) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) chunk_store32(flags + (size_t)oslot * ALZ_CHUNK_FLAG_WORDS, (epoch << 2) | flagv);
    }
}

// ------------------------------------------------------------------------------------------------
// Token-queue kernel (PRS, LZ4, LZO, Snappy, FastLZ, CNX2): lane-assisted / scalar parse into a 64-token queue, lane-parallel
// execution (pipelined_rounds for the bulk, QueueSink behind the exact parsers).
template <int FMT>
#ifndef ALZ_QUEUE_WAVES
#define ALZ_QUEUE_WAVES 6
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FMT == ALZ_FMT_LZO ? 5 : ALZ_QUEUE_WAVES, 8))) void alz_decode_queue_kernel(   /* (LZO: 81 registers; forcing 80 spills one) */const u8* __restrict__ src_base, u8* __restrict__ dst_base,
                                                              const alz_stream* __restrict__ streams,
                                                              const u32* __restrict__ index_list, u32 count,
                                                              alz_result* __restrict__ results) {
    constexpr bool PRS = (FMT == ALZ_FMT_PRS_BE || FMT == ALZ_FMT_PRS_LE);
    // These kernels are bound by the latency of one wave's scalar parse, so waves per CU matter more than LDS hits:
    // only the most recent 4 KiB of the window stay in LDS (5.4 KB per wave -> 28 waves per CU instead of 15), older
    // sources are read back from the stream's own output in HBM, batched per token queue.
    constexpr bool CNS = (FMT == ALZ_FMT_CNS);
    constexpr bool SHREK = (FMT == ALZ_FMT_LZSHREK);        // 4 KiB window, literal runs: the same configuration
    constexpr bool CNX = (FMT == ALZ_FMT_CNX2 || CNS || SHREK);      // 2 KiB / 256 B windows: all of it in LDS, literal runs from the input cache
    constexpr bool PRSFB = (ALZ_PRS_LW < 8192);             // PRS with a window smaller than its 8 KiB: read-back like the 64 KiB formats
    constexpr u32 LW = PRS ? (u32)ALZ_PRS_LW : ALZ_QUEUE_LW;   // PRS: its whole 8 KiB window (half of its matches would otherwise go to HBM)
    // static LDS: byte-phase scratch (marks + token table) | token staging (256) | input cache | window + mirror
    // input cache of two 512-byte chunks (CNX2: 1 KiB chunks): a round looks at most 256 + ALZ_QRUN bytes ahead, and 1 KiB less LDS
    // per wave is four more waves per CU for the 4 KiB-window formats
    constexpr u32 QCH = (FMT == ALZ_FMT_CNX2) ? 1024u : 512u, QCACHE = 2u * QCH + 32u, QAHEAD = QCH + 76u;
    constexpr bool FB = CNX ? false : (!PRS || PRSFB);
    constexpr u32 SCR = FB ? ALZ_EMIT_SCRATCH : 128u, SLACK = FB ? ALZ_WIN_SLACK : 0u;     // (chunked byte phase: token table + ring mirror)
    __shared__ __attribute__((aligned(16))) u8 lds[SCR + 256 + QCACHE + LW + SLACK];
    u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)threadIdx.x;
    const u32 sid = index_list ? index_list[bid] : bid;
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    u8* dst = dst_base + st.dst_off;
    const u32 src_len = uni(st.src_len);
    u32 cap = uni(st.dst_cap);
    const u32 hist = (FMT == ALZ_FMT_LZ4_BLOCK) ? uni(st.aux0) : 0u;      // LZ4 frames with linked blocks (see OutWin::preload)
    dst -= hist; cap += hist;
    u8* segmark = lds;
    u32* stage = reinterpret_cast<u32*>(lds + SCR);
    u8* inc_lds = lds + SCR + 256;
    typedef OutWin<FB> OW;
    OW out; out.init(dst, cap, lds + SCR + 256 + QCACHE, LW, lane, SLACK);
    if (hist) out.preload(hist);
    segmark[lane] = 0; segmark[64 + lane] = 0;
    InCache in; in.init(src, src_len, inc_lds, lane, QCH);
    DecState s; dec_state_init(s);
    typedef EmitCfg<LW - 1u, false, !PRS, FB, false, PRS> CFG;   // (PRS: bound by the scalar unit -- per-step marks, alz_emit_byte.h)
    typedef QueueSink<OW, CFG> SK;
    SK sk(out, s, segmark, inc_lds, lane, PRS ? 8192u : ((FMT == ALZ_FMT_FASTLZ || FMT == ALZ_FMT_REFPACK) ? 131072u : FMT == ALZ_FMT_HIG ? 32768u : (CNS ? 256u : (SHREK ? 4096u : (CNX ? 2048u : 65536u)))));   // (window of the E2 rule: FastLZ level 2 reaches 0x11FFF back)
    if constexpr (PRS) {
        // bulk of the stream: lane-assisted parse (prs_lane_parse) while enough input bytes remain; every token it
        // declines, and the tail of the stream, goes through the exact parser one token at a time
        constexpr bool BIG = (FMT == ALZ_FMT_PRS_BE);
        u32 fl = 1u;                                             // normalised flag register (no bits pending)
#ifndef ALZ_PRS_PRIO
#define ALZ_PRS_PRIO 2
#endif
        // A PRS stream is one long chain of dependent scalar instructions (2.5 ms alone for 256 KiB, three times a Yaz0 stream): in a
        // mixed batch its waves decide when the launch ends, so they take precedence over the other formats' waves on their CU.
        if (ALZ_PRS_PRIO) __builtin_amdgcn_s_setprio(ALZ_PRS_PRIO);
        for (;;) {
            if (s.p + QAHEAD <= src_len && !s.done) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                if (prs_lane_parse<SK, BIG>(in, sk, s, stage, lane, fl)) { if (s.ovf || s.done) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            prs_from_norm<BIG>(fl, s.bits, s.flag);
            dec_prs_serial<SK, BIG>(in, sk, s, src_len, tail ? 0xFFFFFFFFu : 1u);
            fl = prs_to_norm<BIG>(s.bits, s.flag);
            if (tail || s.eof || s.ovf || s.bad || s.done) break;
        }
    } else if constexpr (FMT == ALZ_FMT_LZ4_BLOCK) {
        // bulk of the stream: lane-parallel sequence parse while enough input bytes remain, exact parser otherwise
        for (;;) {
            if (s.p + QAHEAD <= src_len) {
                sk.ensure(in, s.p, QCH);                        // cache covers [p, p + 1024); flushes the queue if it has to move
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                Lz4Rounds rounds{in, stage, lane};
                if (pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, 65536u, cap, rounds)) { if (s.ovf) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_lz4_serial(in, sk, s, src_len, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || s.done) break;
        }
    }
    else if constexpr (FMT == ALZ_FMT_LZO) {
        LzoState ls; lzo_state_init(ls);
        for (;;) {
            if (ls.started && s.p + QAHEAD <= src_len) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                LzoRounds rounds{in, stage, lane, ls.plain == 0u ? 0u : (ls.plain <= 3u ? 1u : 2u), 0u};
                if (pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, 65536u, cap, rounds)) {
                    ls.plain = rounds.state == 0u ? 0u : (rounds.state == 1u ? 1u : 4u);
                    if (s.ovf) break;
                    continue;
                }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_lzo_serial(in, sk, s, src_len, ls, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || s.done) break;
        }
    }
    else if constexpr (FMT == ALZ_FMT_HIG) {
        const u32 size = uni(st.decom_len);
        for (;;) {
            if (s.bits != 0 && s.p + QAHEAD <= src_len && sk.produced() < size) {     // (behind the initial literal block)
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                HigRounds rounds{in, stage, lane};
                if (out.produced < cap && pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, 32768u, size < cap ? size : cap, rounds)) { if (s.ovf) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_hig_serial(in, sk, s, src_len, size, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || sk.produced() >= size) break;
        }
    }
    else if constexpr (SHREK) {
        for (;;) {
            if (s.p + QAHEAD <= src_len) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                LzshrekRounds rounds{in, stage, lane, s.bits, 0u};           // s.bits: matches the current group still owes (the exact parser's state)
                if (out.produced < cap && pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, 4096u, cap, rounds)) {
                    s.bits = rounds.state;
                    if (s.ovf) break;
                    continue;
                }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_lzshrek_serial(in, sk, s, src_len, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || s.done) break;
        }
    }
    else if constexpr (CNS) {
        const u32 size = uni(st.decom_len);
        for (;;) {
            if (s.p + QAHEAD <= src_len && sk.produced() < size) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                CnsRounds rounds{in, lane};
                if (out.produced < cap && pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, 256u, size < cap ? size : cap, rounds)) { if (s.ovf) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_cns_serial(in, sk, s, src_len, size, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || sk.produced() >= size) break;
        }
    }
    else if constexpr (CNX) {
        // bulk: lane-parallel rounds of whole groups while enough input bytes remain; the exact parser for a group the
        // rounds decline, to get back to a flag-byte boundary, and for the end of the stream
        const u32 size = uni(st.decom_len);
        for (;;) {
            if (s.bits == 0 && s.p + QAHEAD <= src_len && sk.produced() < size) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                Cnx2Rounds rounds{in, stage, lane};
                if (out.produced < cap && pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, 2048u, size < cap ? size : cap, rounds)) { if (s.ovf) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_cnx2_serial(in, sk, s, src_len, size, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || sk.produced() >= size) break;
        }
    }
    else if constexpr (FMT == ALZ_FMT_WFLZ || FMT == ALZ_FMT_WFLZ_BE) {
        constexpr bool BIG = (FMT == ALZ_FMT_WFLZ_BE);
        for (;;) {
            if (s.p + QAHEAD <= src_len) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                WflzRounds<BIG> rounds{in, stage, lane};
                if (out.produced < cap && pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, 65536u, cap, rounds)) { if (s.ovf) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_wflz_serial<SK, BIG>(in, sk, s, src_len, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || s.done) break;
        }
    }
    else if constexpr (FMT == ALZ_FMT_REFPACK) {
        for (;;) {
            if (s.p + QAHEAD <= src_len) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                RefpackRounds rounds{in, stage, lane};
                if (out.produced < cap && pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, 131072u, cap, rounds)) { if (s.ovf) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_refpack_serial(in, sk, s, src_len, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || s.done) break;
        }
    }
    else if constexpr (FMT == ALZ_FMT_FASTLZ) {
        FastlzState fz; fastlz_state_init(fz);
        for (;;) {
            if (fz.started && s.p + QAHEAD <= src_len) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                FastlzRounds rounds{in, lane, fz.level};
                if (pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, fastlz_window(fz), cap, rounds)) { if (s.ovf) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_fastlz_serial(in, sk, s, src_len, fz, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || s.done) break;
        }
    }
    else {
        // Snappy: varint size, then elements until the output reaches it; lane-parallel parse for the bulk
        u32 size = 0; bool have = false;
        for (;;) {
            if (have && s.p + QAHEAD <= src_len) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                if (out.produced >= size) break;
                SnappyRounds rounds{in, lane};
                if (out.produced < cap && pipelined_rounds<OW, CFG>(in, out, s, src_len, segmark, inc_lds, lane, 65536u, size < cap ? size : cap, rounds)) { if (s.ovf) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            dec_snappy_serial(in, sk, s, src_len, size, have, tail ? 0xFFFFFFFFu : 1u);
            if (tail || s.eof || s.ovf || s.bad || sk.produced() >= size) break;
        }
    }
    sk.flush();                                    // tokens parsed before an error/terminator are part of the output
    out.finish();
    constexpr bool SIZED = CNX || FMT == ALZ_FMT_REFPACK || FMT == ALZ_FMT_HIG;       // (RefPack: only more output than declared is an error, checked at the end token)
    write_result(&results[sid], lane, out, s.p, resolve_status(s, SIZED, out.produced, SIZED ? uni(st.decom_len) : 0u, cap), src_len, hist);
}

// ------------------------------------------------------------------------------------------------
// The single-cursor flag-byte formats with TWO wavefronts per stream, for launches that cannot fill the GPU with one (a CU holds
// ~24 streams either way: below ~6 000 streams the time of a launch is the time of ONE stream -- 1.1 ms per 256 KiB of Yaz0, front
// end and byte phase taking turns on a single wavefront).  Wavefront 0 runs the front end and the token prologue of
// fast_iter_interleaved unchanged, on a WalkOut instead of a window: every batch of tokens goes through an LDS mailbox to
// wavefront 1, which owns the window and runs the byte phase (byte_emit_steps); one workgroup barrier per batch.  When the fast loop
// ends (stream finished, or its tail left to the exact parser) the walker hands over its parser state and leaves; wavefront 1 finishes
// alone exactly as the one-wavefront kernel does.
template <int FMT>
__global__ __launch_bounds__(128) void alz_decode_fast2_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                               const u32* __restrict__ index_list, u32 count, alz_result* __restrict__ results,
                                                               alz_lz_properties lz, u32 lw, const u32* __restrict__ gate) {
    if (gate != nullptr && __builtin_nontemporal_load(gate) == 0u) return;   // (alz_launch_decode_gated)
    constexpr u32 CHUNK = 1024u, CACHE = 2u * CHUNK + 32u, FSCR = 128u, LWMAX = 4096u;
    __shared__ __attribute__((aligned(16))) u8 lds[FSCR + CACHE + LWMAX + CACHE + 2u * ALZ_MBOX_WORDS * 4u];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)(threadIdx.x & 63u);
    const u32 sid = index_list ? index_list[bid] : bid;
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u32 src_len = uni(st.src_len), cap = uni(st.dst_cap), size = uni(st.decom_len);
    u32* mbox = reinterpret_cast<u32*>(lds + FSCR + CACHE + LWMAX + CACHE);
    FastGeom gm; gm.length_bits = lz.length_bits; gm.min_length = lz.min_length; gm.windows_start = lz.windows_start;
    gm.max_distance = lz.max_distance; gm.W = 1u << lz.window_bits;
    if (threadIdx.x < 64u) {
        // ---- the parsing wavefront
        InCache in; in.init(src, src_len, lds + FSCR + CACHE + LWMAX, lane, CHUNK);
        DecState s; dec_state_init(s);
        WalkOut out; out.produced = 0; out.cap = cap; out.mbox = mbox; out.k = 0;
        bool fin = false, to_serial = false;
        while (!fin && !to_serial && out.produced < size && s.p < src_len) fin = fast_iter_interleaved<FMT>(in, out, s, size, src_len, to_serial, nullptr, lane, gm);
        u32* slot = mbox + (out.k & 1u) * ALZ_MBOX_WORDS;
        if (lane == 0) {
            slot[192] = 2u; slot[193] = s.p; slot[194] = s.bits; slot[195] = s.flag; slot[196] = fin ? 1u : 0u; slot[197] = s.ovf ? 1u : 0u;
            slot[198] = (u32)s.attempted_end; slot[199] = (u32)(s.attempted_end >> 32);
        }
        __syncthreads();
        return;
    }
    // ---- the executing wavefront
    u8* dst = dst_base + st.dst_off;
    u8* segmark = lds;
    u8* inc_lds = lds + FSCR;
    typedef OutWin<false> OWF;
    OWF out; out.init(dst, cap, lds + FSCR + CACHE, lw, lane, 0u);
    segmark[lane] = 0; segmark[64 + lane] = 0;
    DecState s; dec_state_init(s);
    typedef EmitCfg<(FMT == ALZ_FMT_LZSS ? 0u : 4095u), FMT == ALZ_FMT_LZSS, false, false> CFG;
    bool fin = false;
    for (u32 k = 0;; k++) {
        __syncthreads();
        const u32* slot = mbox + (k & 1u) * ALZ_MBOX_WORDS;
        const u32 kind = uni(slot[192]);
        if (kind == 2u) {
            s.p = uni(slot[193]); s.bits = uni(slot[194]); s.flag = uni(slot[195]); fin = uni(slot[196]) != 0u;
            s.ovf = uni(slot[197]) != 0u; s.attempted_end = ((u64)uni(slot[199]) << 32) | uni(slot[198]);
            break;
        }
        EmitState e;
        e.clen = slot[lane]; e.kept = e.clen != 0u; e.off = slot[64 + lane]; e.desc = slot[128 + lane];
        e.O = uni(slot[193]); e.T = uni(slot[194]); e.W = uni(slot[195]); e.fin = false;
        byte_emit_steps<OWF, CFG>(out, segmark, nullptr, lane, e);
    }
    if (!fin) {
        InCache in; in.init(src, src_len, inc_lds, lane, CHUNK);
        typedef DirectSink<OWF> SK;
        SK sk(out, s);
        if constexpr (FMT == ALZ_FMT_LZSS) dec_lzss_serial(in, sk, s, src_len, size, lz.length_bits, lz.min_length, lz.windows_start, lz.max_distance, gm.W);
        else if constexpr (FMT == ALZ_FMT_LZ10) dec_lz1x_serial<SK, false>(in, sk, s, src_len, size);
        else if constexpr (FMT == ALZ_FMT_LZ11) dec_lz1x_serial<SK, true>(in, sk, s, src_len, size);
        else if constexpr (FMT == ALZ_FMT_LZ40) dec_lz40_serial(in, sk, s, src_len, size);
        else if constexpr (FMT == ALZ_FMT_CLZ0) dec_clz0_serial(in, sk, s, src_len, size);
        else dec_yaz0_serial(in, sk, s, src_len, size);
    }
    out.finish();
    write_result(&results[sid], lane, out, s.p, resolve_status(s, true, out.produced, size, cap), src_len);
}

// The same for the three-cursor formats (Yay0 / MIO0): the walker owns three small input caches (flags, tokens, literals) and hands over
// its three cursors at the end.
template <int FMT>
__global__ __launch_bounds__(128) void alz_decode_fast2c_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                                const u32* __restrict__ index_list, u32 count, alz_result* __restrict__ results, u32 lw, const u32* __restrict__ gate) {
    if (gate != nullptr && __builtin_nontemporal_load(gate) == 0u) return;   // (alz_launch_decode_gated)
    constexpr u32 CHUNK = 256u, CACHE = ALZ_INCACHE_SMALL, FSCR = ALZ_BYTE_SCRATCH, LWMAX = 4096u;
    __shared__ __attribute__((aligned(16))) u8 lds[FSCR + 3u * CACHE + LWMAX + 3u * CACHE + 2u * ALZ_MBOX_WORDS * 4u];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)(threadIdx.x & 63u);
    const u32 sid = index_list ? index_list[bid] : bid;
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u32 src_len = uni(st.src_len), cap = uni(st.dst_cap), size = uni(st.decom_len);
    const u32 a0 = uni(st.aux0), a1 = uni(st.aux1);
    u32* mbox = reinterpret_cast<u32*>(lds + FSCR + 3u * CACHE + LWMAX + 3u * CACHE);
    const bool bad = FMT == ALZ_FMT_YAY0 && (a0 > src_len || a1 > src_len);        // Slice() throws  Yay0.cs:102-103
    if (threadIdx.x < 64u) {
        // ---- the parsing wavefront
        u8* wl = lds + FSCR + 3u * CACHE + LWMAX;
        DecState s; dec_state_init(s);
        WalkOut out; out.produced = 0; out.cap = cap; out.mbox = mbox; out.k = 0;
        bool fin = false;
        u32 fp = 0, cp = a0, up = a1;
        if (!bad) {
            InCache in, cin, uin;
            in.init(src, src_len, wl, lane, CHUNK);
            cin.init(src, src_len, wl + CACHE, lane, CHUNK); cin.seek(a0 < src_len ? a0 : 0);
            uin.init(src, src_len, wl + 2 * CACHE, lane, CHUNK); uin.seek(a1 < src_len ? a1 : 0);
            while (!fin && out.produced < size && fp + 8u <= src_len && (u64)cp + 128u <= src_len && (u64)up + 64u <= src_len)
                fin = fast_iter_3cursor<FMT == ALZ_FMT_MIO0>(in, cin, uin, out, s, size, nullptr, lane, fp, cp, up);
        }
        u32* slot = mbox + (out.k & 1u) * ALZ_MBOX_WORDS;
        if (lane == 0) {
            slot[192] = 2u; slot[193] = fp; slot[194] = cp; slot[195] = up; slot[196] = fin ? 1u : 0u; slot[197] = s.ovf ? 1u : 0u;
            slot[198] = (u32)s.attempted_end; slot[199] = (u32)(s.attempted_end >> 32);
        }
        __syncthreads();
        return;
    }
    // ---- the executing wavefront
    u8* dst = dst_base + st.dst_off;
    u8* segmark = lds;
    u8* inc_lds = lds + FSCR;
    typedef OutWin<false> OWF;
    OWF out; out.init(dst, cap, lds + FSCR + 3u * CACHE, lw, lane, 0u);
    segmark[lane] = 0; segmark[64 + lane] = 0;
    DecState s; dec_state_init(s);
    typedef EmitCfg<4095u, false, false, false, false, false, true> CFG;   // (FSCR = ALZ_BYTE_SCRATCH: the descriptor table)
    bool fin = false;
    u32 fp = 0, cp = a0, up = a1;
    for (u32 k = 0;; k++) {
        __syncthreads();
        const u32* slot = mbox + (k & 1u) * ALZ_MBOX_WORDS;
        if (uni(slot[192]) == 2u) {
            fp = uni(slot[193]); cp = uni(slot[194]); up = uni(slot[195]); fin = uni(slot[196]) != 0u;
            s.ovf = uni(slot[197]) != 0u; s.attempted_end = ((u64)uni(slot[199]) << 32) | uni(slot[198]);
            break;
        }
        EmitState e;
        e.clen = slot[lane]; e.kept = e.clen != 0u; e.off = slot[64 + lane]; e.desc = slot[128 + lane];
        e.O = uni(slot[193]); e.T = uni(slot[194]); e.W = uni(slot[195]); e.fin = false;
        byte_emit_steps<OWF, CFG>(out, segmark, nullptr, lane, e);
    }
    u32 used = cp > up ? cp : up;
    if (bad) s.eof = true;
    else if (!fin) {
        InCache in, cin, uin;
        in.init(src, src_len, inc_lds, lane, CHUNK); in.seek(fp < src_len ? fp : 0);
        cin.init(src, src_len, inc_lds + CACHE, lane, CHUNK); cin.seek(cp < src_len ? cp : 0);
        uin.init(src, src_len, inc_lds + 2 * CACHE, lane, CHUNK); uin.seek(up < src_len ? up : 0);
        typedef DirectSink<OWF> SK; SK sk(out, s);
        used = dec_3cursor_serial<SK, FMT == ALZ_FMT_MIO0>(in, cin, uin, sk, s, src_len, size, fp, cp, up);
    }
    out.finish();
    write_result(&results[sid], lane, out, bad ? s.p : used, resolve_status(s, true, out.produced, size, cap), src_len);
}

// ------------------------------------------------------------------------------------------------
// PRS with TWO wavefronts per stream: wavefront 0 walks (prs_parse_round: input cache + scalar walk, no window), wavefront 1
// executes (the byte phase on the window); a round's tokens cross in a two-slot LDS mailbox, one workgroup barrier per round.  A PRS
// stream alone is a chain of ~46 000 dependent token steps: ~60 cycles of scalar walk and ~70 of byte phase per token, one after
// the other on a single wavefront (2.5 ms per 256 KiB) -- which is what a small or mixed batch waits for (cfg4's shard: 3.3 ms for
// 1 250 PRS streams of 2.5-2.9 ms).  Here round k + 1 is parsed while round k is executed.  The walker stops at the first round
// it cannot take (input tail, a round that would not fit the capacity, nothing parsed): wavefront 1 then carries on alone from
// that position with the one-wavefront loop of alz_decode_queue_kernel (exact parser included).
template <int FMT>
__global__ __launch_bounds__(128) void alz_decode_prs2_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                              const u32* __restrict__ index_list, u32 count, alz_result* __restrict__ results, const u32* __restrict__ gate) {
    if (gate != nullptr && __builtin_nontemporal_load(gate) == 0u) return;   // (alz_launch_decode_gated)
    constexpr bool BIG = (FMT == ALZ_FMT_PRS_BE);
    constexpr u32 LW = 8192u, QCH = 512u, QCACHE = 2u * QCH + 32u, QAHEAD = QCH + 76u, SCR = 128u;
    constexpr u32 MB = 64u + 8u;                              // mailbox slot: 64 tokens + {nt | stop, total, position behind, flag register, terminator}
    __shared__ __attribute__((aligned(16))) u8 lds[SCR + 256 + QCACHE + LW + 256 + QCACHE + 2u * MB * 4u];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)(threadIdx.x & 63u);
    const bool walker = threadIdx.x < 64u;
    const u32 sid = index_list ? index_list[bid] : bid;
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u32 src_len = uni(st.src_len), cap = uni(st.dst_cap);
    u32* mbox = reinterpret_cast<u32*>(lds + SCR + 256 + QCACHE + LW + 256 + QCACHE);
    if (walker) {
        u32* stage = reinterpret_cast<u32*>(lds + SCR + 256 + QCACHE + LW);
        InCache in; in.init(src, src_len, lds + SCR + 256 + QCACHE + LW + 256, lane, QCH);
        u32 p = 0, fl = 1u, produced = 0;
        for (u32 k = 0;; k++) {
            u32 qt = 0, nt = 0, total = 0, adv = 0, fl2 = fl, term = 0;
            bool ok = (u64)p + QAHEAD <= src_len;
            if (ok) { in.ensure(p, QCH); ok = prs_parse_round<BIG>(in, p, fl, stage, lane, qt, nt, total, adv, fl2, term); }
            if (ok && total > cap - produced) ok = false;         // the capacity rule (E5) stays with the exact parser
            u32* slot = mbox + (k & 1u) * MB;
            slot[lane] = qt;
            if (lane == 0) { slot[64] = ok ? nt : 0xFFFFFFFFu; slot[65] = total; slot[66] = ok ? p + adv : p; slot[67] = ok ? fl2 : fl; slot[68] = term; }
            __syncthreads();                                       // round k is in the mailbox (and round k - 1 has been executed)
            if (!ok || term) return;
            p += adv; fl = fl2; produced += total;
        }
    }
    // ---- the executing wavefront
    u8* dst = dst_base + st.dst_off;
    u8* segmark = lds;
    u32* stage = reinterpret_cast<u32*>(lds + SCR);
    u8* inc_lds = lds + SCR + 256;
    typedef OutWin<false> OW;
    OW out; out.init(dst, cap, lds + SCR + 256 + QCACHE, LW, lane, 0u);
    segmark[lane] = 0; segmark[64 + lane] = 0;
    DecState s; dec_state_init(s);
    typedef EmitCfg<LW - 1u, false, false, false, false, true> CFG;
    typedef QueueSink<OW, CFG> SK;
    __builtin_amdgcn_s_setprio(ALZ_PRS_PRIO);
    u32 fl = 1u;
    for (u32 k = 0;; k++) {
        __syncthreads();
        const u32* slot = mbox + (k & 1u) * MB;
        const u32 qt = slot[lane], nt = uni(slot[64]), p2 = uni(slot[66]);
        fl = uni(slot[67]);
        const u32 term = uni(slot[68]);
        s.p = p2;
        if (nt == 0xFFFFFFFFu) break;                              // the walker stopped in front of this round
        const u32 len = qt >> 18, lo = qt & 0x1FFFFu;
        const u32 desc = (qt & 0x20000u) ? ALZ_DESC_LIT(lo & 0xFFu) : lo;
        u32 last;
        (void)fast_emit<OW, CFG>(out, s, 0xFFFFFFFFu, lanes_below(nt), len, desc, 0u, segmark, inc_lds, lane, last, 8192u);
        if (term) { s.done = true; break; }                        // PRS.cs:78-79: the zero word ends the stream
    }
    SK sk(out, s, segmark, inc_lds, lane, 8192u);
    if (!s.done && !s.ovf) {
        // alone from here: the loop of alz_decode_queue_kernel
        InCache in; in.init(src, src_len, inc_lds, lane, QCH);
        for (;;) {
            if (s.p + QAHEAD <= src_len && !s.done) {
                sk.ensure(in, s.p, QCH);
                if (sk.nt) { sk.flush(); if (s.ovf) break; }
                if (prs_lane_parse<SK, BIG>(in, sk, s, stage, lane, fl)) { if (s.ovf || s.done) break; continue; }
            }
            const bool tail = s.p + QAHEAD > src_len;
            prs_from_norm<BIG>(fl, s.bits, s.flag);
            dec_prs_serial<SK, BIG>(in, sk, s, src_len, tail ? 0xFFFFFFFFu : 1u);
            fl = prs_to_norm<BIG>(s.bits, s.flag);
            if (tail || s.eof || s.ovf || s.bad || s.done) break;
        }
    }
    sk.flush();
    out.finish();
    write_result(&results[sid], lane, out, s.p, resolve_status(s, false, out.produced, 0u, cap), src_len, 0u);
}

// ------------------------------------------------------------------------------------------------
// LZ4 / LZO / Snappy with TWO wavefronts per stream (round 3), as PRS: wavefront 0 parses rounds (input cache + element walk, no
// window), wavefront 1 executes them (chunked byte phase on the window, far sources from HBM); a round's tokens cross in a two-slot
// LDS mailbox, one workgroup barrier per round.  Parse and copy were about equal halves of the one-wavefront kernel, taking turns;
// and its rounds drained whenever the input cache had to slide, because a round's literal-run tokens point into the cache.  Here the
// executing wavefront reads a literal run from the stream's INPUT in global memory -- one 20-byte read per chunk, the very read it
// does for a far source (the bytes are in this CU's L1: the parsing wavefront fetched them a moment ago) -- so the parser's cache is
// the parser's alone.  What the rounds do not take (an element with a second length-extension byte, the stream's head and tail, the
// capacity rule) goes to the exact parser on wavefront 1, which then tells the parser where and in which state to go on.
template <int FMT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(FMT == ALZ_FMT_LZO ? 5 : ALZ_QUEUE_WAVES, 8)))
void alz_decode_queue2_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                              const u32* __restrict__ index_list, u32 count, alz_result* __restrict__ results, const u32* __restrict__ gate) {
    if (gate != nullptr && __builtin_nontemporal_load(gate) == 0u) return;   // (alz_launch_decode_gated)
    constexpr bool LZ4 = FMT == ALZ_FMT_LZ4_BLOCK, LZO = FMT == ALZ_FMT_LZO, SNAPPY = FMT == ALZ_FMT_SNAPPY_RAW;
    static_assert(LZ4 || LZO || SNAPPY, "formats with lane-parallel rounds and literal runs");
    constexpr u32 LW = ALZ_QUEUE_LW, QCH = 512u, QCACHE = 2u * QCH + 32u, QAHEAD = QCH + 76u, SCR = ALZ_EMIT_SCRATCH, SLACK = ALZ_WIN_SLACK;
    constexpr u32 MB = 64u + 8u;                              // mailbox slot: 64 tokens + {nt | stop, total, round start, its cache index, parser state}
    // executing wavefront: byte-phase scratch | staging | input cache (exact parser only) | window + mirror;  parsing wavefront: staging | input cache;
    // mailbox (two slots) | control words (go / done, position, parser state, bytes produced, output limit)
    __shared__ __attribute__((aligned(16))) u8 lds[SCR + 256 + QCACHE + LW + SLACK + 256 + QCACHE + 2u * MB * 4u + 64u];
    const u32 bid = blockIdx.x;
    if (bid >= count) return;
    const int lane = (int)(threadIdx.x & 63u);
    const bool walker = threadIdx.x < 64u;
    const u32 sid = index_list ? index_list[bid] : bid;
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u32 src_len = uni(st.src_len);
    u32 cap = uni(st.dst_cap);
    const u32 hist = LZ4 ? uni(st.aux0) : 0u;                 // LZ4 frames with linked blocks (see OutWin::preload)
    cap += hist;
    u8* const wl = lds + SCR + 256 + QCACHE + LW + SLACK;
    u32* const mbox = reinterpret_cast<u32*>(wl + 256 + QCACHE);
    u32* const ctl = mbox + 2u * MB;
    if (walker) {
        u32* stage = reinterpret_cast<u32*>(wl);
        InCache in; in.init(src, src_len, wl + 256, lane, QCH);
        __syncthreads();                                       // the executing wavefront has read the stream's head
        if (!uni(ctl[0])) return;
        u32 p = uni(ctl[1]), state = uni(ctl[2]), produced = uni(ctl[3]);
        const u32 maxout = uni(ctl[4]);
        for (u32 k = 0;; k++) {
            u32 qt = 0, nt = 0, total = 0, adv = 0, st2 = state;
            bool ok = (u64)p + QAHEAD <= src_len && produced < maxout;
            if (ok) {
                in.ensure(p, QCH);
                if constexpr (LZ4) { Lz4Rounds r{in, stage, lane}; ok = r(p, qt, nt, total, adv); }
                else if constexpr (LZO) { LzoRounds r{in, stage, lane, state, 0u}; ok = r(p, qt, nt, total, adv); st2 = r.pending; }
                else { SnappyRounds r{in, lane}; ok = r(p, qt, nt, total, adv); }
            }
            if (ok && total > maxout - produced) ok = false;   // the size / capacity rules stay with the exact parser
            u32* slot = mbox + (k & 1u) * MB;
            slot[lane] = qt;
            if (lane == 0) { slot[64] = ok ? nt : 0xFFFFFFFFu; slot[65] = total; slot[66] = p; slot[67] = ok ? in.idx(p) : 0u; slot[68] = state; }
            __syncthreads();                                   // message k is in the mailbox (and round k - 1 has been executed)
            if (!ok) {
                __syncthreads();                               // the exact parser has taken what the rounds did not
                if (!uni(ctl[0])) return;
                p = uni(ctl[1]); state = uni(ctl[2]); produced = uni(ctl[3]);
                continue;
            }
            p += adv; state = st2; produced += total;
        }
    }
    // ---- the executing wavefront
    u8* dst = dst_base + st.dst_off - hist;
    u8* segmark = lds;
    u8* inc_lds = lds + SCR + 256;
    typedef OutWin<true> OW;
    OW out; out.init(dst, cap, lds + SCR + 256 + QCACHE, LW, lane, SLACK);
    if (hist) out.preload(hist);
    segmark[lane] = 0; segmark[64 + lane] = 0;
    InCache in; in.init(src, src_len, inc_lds, lane, QCH);
    DecState s; dec_state_init(s);
    typedef EmitCfg<LW - 1u, false, true, true> CFG;          // the exact parser's tokens: literal runs from this wavefront's input cache
    typedef EmitCfg<LW - 1u, false, true, true, true> CFG2;   // the parser's rounds: literal runs from the stream's input in global memory
    typedef QueueSink<OW, CFG> SK;
    SK sk(out, s, segmark, inc_lds, lane, 65536u);
    LzoState ls; lzo_state_init(ls);
    u32 size = 0; bool have = false;
    // what the rounds do not take: one element through the exact parser (everything that is left once the input tail is reached);
    // true: the stream is finished
    auto serial_step = [&]() -> bool {
        const bool tail = (u64)s.p + QAHEAD > src_len;
        const u32 n = tail ? 0xFFFFFFFFu : 1u;
        if constexpr (LZ4) dec_lz4_serial(in, sk, s, src_len, n);
        else if constexpr (LZO) dec_lzo_serial(in, sk, s, src_len, ls, n);
        else dec_snappy_serial(in, sk, s, src_len, size, have, n);
        sk.flush();
        return tail || s.eof || s.ovf || s.bad || s.done || (SNAPPY && sk.produced() >= size);
    };
    // the head: until rounds may start (LZO: behind the first byte's rule; Snappy: behind the varint size; all: at least four bytes
    // in, so that the 20-byte read of a literal run never starts in front of the stream)
    bool fin = false;
    while (!fin && !(s.p >= 4u && (!LZO || ls.started) && (!SNAPPY || have))) fin = serial_step();
    if (lane == 0) {
        ctl[0] = fin ? 0u : 1u; ctl[1] = s.p; ctl[2] = LZO ? (ls.plain == 0u ? 0u : (ls.plain <= 3u ? 1u : 2u)) : 0u; ctl[3] = out.produced;
        ctl[4] = SNAPPY ? (size < cap ? size : cap) : cap;
    }
    __syncthreads();
    if (!fin) for (u32 k = 0;; k++) {
        __syncthreads();                                       // message k is in the mailbox
        const u32* slot = mbox + (k & 1u) * MB;
        const u32 qt = slot[lane], nt = uni(slot[64]);
        if (nt == 0xFFFFFFFFu) {                               // the parser stopped in front of an element the rounds do not take
            s.p = uni(slot[66]);
            if (LZO) { const u32 ws = uni(slot[68]); ls.plain = ws == 0u ? 0u : (ws == 1u ? 1u : 4u); }
            fin = serial_step();
            if (lane == 0) { ctl[0] = fin ? 0u : 1u; ctl[1] = s.p; ctl[2] = LZO ? (ls.plain == 0u ? 0u : (ls.plain <= 3u ? 1u : 2u)) : 0u; ctl[3] = out.produced; }
            __syncthreads();
            if (fin) break;
            continue;
        }
        const u8* runbase = src + uni(slot[66]) - uni(slot[67]);     // + a literal run's cache index = where the run lies in the input
        const u32 len = qt >> 18, lo = qt & 0x1FFFFu;
        const u32 desc = (qt & 0x20000u) ? (0x80000000u | lo) : lo;
        u32 last;
        (void)fast_emit<OW, CFG2>(out, s, 0xFFFFFFFFu, lanes_below(nt), len, desc, 0u, segmark, runbase, lane, last, 65536u);
    }
    sk.flush();
    out.finish();
    write_result(&results[sid], lane, out, s.p, resolve_status(s, false, out.produced, 0u, cap), src_len, hist);
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (host)
// PRS as a work queue of chunks (the two-wavefront kernel above, one (stream, chunk) item per workgroup; the hand-over of alz_decode_fastq_kernel
// with an 8 KiB window and the parser's state -- input offset, flag register -- in the slot).  The PARSING wavefront decides where a chunk ends: the
// round that takes the output to the chunk's limit carries a mark, the executing wavefront hands over behind it.  A round the lane parser declines
// sends the executing wavefront on alone, to the end of the stream, as in the kernel above.
template <int FMT>
__global__ __launch_bounds__(128) void alz_decode_prs2q_kernel(const u8* __restrict__ src_base, u8* __restrict__ dst_base, const alz_stream* __restrict__ streams,
                                                               const alz_chunk_item* __restrict__ items, alz_queue_bounds qb, alz_result* __restrict__ results,
                                                               u32* __restrict__ head, u32* __restrict__ flags, u8* __restrict__ slots, u32* __restrict__ tmo, u32 epoch) {
    constexpr bool BIG = (FMT == ALZ_FMT_PRS_BE);
    constexpr u32 LW = 8192u, QCH = 512u, QCACHE = 2u * QCH + 32u, QAHEAD = QCH + 76u, SCR = 128u;
    constexpr u32 MB = 64u + 8u;                              // mailbox slot: 64 tokens + {nt | stop, total, position behind, flag register, terminator, chunk end}
    constexpr u32 CHB = ALZ_CHUNK_OUT_PRS;
    __shared__ __attribute__((aligned(16))) u8 lds[SCR + 256 + QCACHE + LW + 256 + QCACHE + 2u * MB * 4u + 32u];
    const int lane = (int)(threadIdx.x & 63u);
    const bool walker = threadIdx.x < 64u;
    // the workgroup's item: a ticket drawn by the first wavefront (queue_ticket, above alz_decode_fastq_kernel), shared with the other one through LDS
    u32* const ticket = reinterpret_cast<u32*>(lds + SCR + 256 + QCACHE + LW + 256 + QCACHE) + 2u * MB + 4u;
    if (threadIdx.x < 64u) { const u32 t = queue_ticket(head, epoch, qb, lane); if (lane == 0) *ticket = t; }
    __syncthreads();
    const u32 item = uni(*ticket);
    if (item == 0xFFFFFFFFu) return;
    const alz_chunk_item it = items[item];
    const u32 sid = uni(it.sid), c = uni(it.chunk), oslot = uni(it.slot), last = uni(it.last);
    const alz_stream st = streams[sid];
    const u8* src = src_base + st.src_off;
    const u32 src_len = uni(st.src_len), cap = uni(st.dst_cap);
    u32* mbox = reinterpret_cast<u32*>(lds + SCR + 256 + QCACHE + LW + 256 + QCACHE);
    u32* share = mbox + 2u * MB;                              // {go, p, fl, start}: what the chunk before left, from the executing wavefront to the parsing one
    constexpr u32 slot_bytes = 32u + LW;
    const u32 limit = last ? 0xFFFFFFFFu : (c + 1u) * CHB;
    if (walker) {
        __syncthreads();                                       // [S0] the state is in `share`
        if (share[0] == 0u) return;
        u32* stage = reinterpret_cast<u32*>(lds + SCR + 256 + QCACHE + LW);
        u32 p = share[1], fl = share[2], produced = share[3];
        InCache in; in.init_at(src, src_len, lds + SCR + 256 + QCACHE + LW + 256, lane, QCH, p);
        for (u32 k = 0;; k++) {
            u32 qt = 0, nt = 0, total = 0, adv = 0, fl2 = fl, term = 0;
            bool ok = (u64)p + QAHEAD <= src_len;
            if (ok) { in.ensure(p, QCH); ok = prs_parse_round<BIG>(in, p, fl, stage, lane, qt, nt, total, adv, fl2, term); }
            if (ok && total > cap - produced) ok = false;         // the capacity rule (E5) stays with the exact parser
            const bool cend = ok && !term && produced + total >= limit;   // this round takes the output to the chunk's limit: the stream is handed over behind it
            u32* slot = mbox + (k & 1u) * MB;
            slot[lane] = qt;
            if (lane == 0) { slot[64] = ok ? nt : 0xFFFFFFFFu; slot[65] = total; slot[66] = ok ? p + adv : p; slot[67] = ok ? fl2 : fl; slot[68] = term; slot[69] = cend ? 1u : 0u; }
            __syncthreads();                                       // round k is in the mailbox (and round k - 1 has been executed)
            if (!ok || term || cend) return;
            p += adv; fl = fl2; produced += total;
        }
    }
    // ---- the executing wavefront: waits for the chunk before, owns the window, hands over
    u8* dst = dst_base + st.dst_off;
    u8* segmark = lds;
    u32* stage = reinterpret_cast<u32*>(lds + SCR);
    u8* inc_lds = lds + SCR + 256;
    u8* const win = lds + SCR + 256 + QCACHE;
    const u8* in_slot = slots + (size_t)(oslot ? oslot - 1u : 0u) * slot_bytes;
    u8* out_slot = slots + (size_t)oslot * slot_bytes;
    u32 flagv = 0; bool run = true;
    u32 p0 = 0, fl = 1u, start = 0;
    if (c > 0u) {
        const u32 f = queue_flag_wait(flags + (size_t)(oslot - 1u) * ALZ_CHUNK_FLAG_WORDS, epoch, lane);
        if (f == 0u || f == 3u) { if (lane == 0) atomicOr(tmo, 1u); flagv = 3u; run = false; }
        else {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (f == 2u) { flagv = 2u; run = false; }
            else { p0 = uni(chunk_load32(in_slot)); fl = uni(chunk_load32(in_slot + 4)); start = uni(chunk_load32(in_slot + 12)); }
        }
    }
    if (lane == 0) { share[0] = run ? 1u : 0u; share[1] = p0; share[2] = fl; share[3] = start; }
    __syncthreads();                                           // [S0]
    if (run) {
        typedef OutWin<false> OW;
        OW out;
        if (c == 0u) out.init(dst, cap, win, LW, lane, 0u);
        else {
            out.dst = dst; out.cap = cap; out.win = win; out.lw_mask = LW - 1u; out.lane = lane; out.fl = 1024u;
            out.oshift = (u32)(reinterpret_cast<uintptr_t>(dst) & 15u);
            out.produced = start; out.flushed = start; out.slack_dirty = false; out.mtag = 0;
            for (u32 i = 16u * (u32)lane; i < LW; i += 1024u) *reinterpret_cast<uint4*>(win + i) = *reinterpret_cast<const uint4*>(in_slot + 32 + i);
            wave_sync();
        }
        segmark[lane] = 0; segmark[64 + lane] = 0;
        DecState s; dec_state_init(s);
        s.p = p0;
        typedef EmitCfg<LW - 1u, false, false, false, false, true> CFG;
        typedef QueueSink<OW, CFG> SK;
        __builtin_amdgcn_s_setprio(ALZ_PRS_PRIO);
        bool handoff = false;
        for (u32 k = 0;; k++) {
            __syncthreads();
            const u32* slot = mbox + (k & 1u) * MB;
            const u32 qt = slot[lane], nt = uni(slot[64]), p2 = uni(slot[66]);
            fl = uni(slot[67]);
            const u32 term = uni(slot[68]), cend = uni(slot[69]);
            s.p = p2;
            if (nt == 0xFFFFFFFFu) break;                              // the walker stopped in front of this round
            const u32 len = qt >> 18, lo = qt & 0x1FFFFu;
            const u32 desc = (qt & 0x20000u) ? ALZ_DESC_LIT(lo & 0xFFu) : lo;
            u32 lastt;
            (void)fast_emit<OW, CFG>(out, s, 0xFFFFFFFFu, lanes_below(nt), len, desc, 0u, segmark, inc_lds, lane, lastt, 8192u);
            if (term) { s.done = true; break; }                        // PRS.cs:78-79: the zero word ends the stream
            if (cend) { handoff = !s.ovf; break; }
        }
        if (handoff) {
            out.finish();
            wave_sync();
            for (u32 i = 16u * (u32)lane; i < LW; i += 1024u) {
                const uint4 v = *reinterpret_cast<const uint4*>(win + i);
                __hip_atomic_store(reinterpret_cast<u64*>(out_slot + 32 + i), ((u64)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<u64*>(out_slot + 40 + i), ((u64)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) { chunk_store32(out_slot, s.p); chunk_store32(out_slot + 4, fl); chunk_store32(out_slot + 8, 0u); chunk_store32(out_slot + 12, out.produced); }
            flagv = 1u;
        } else {
            SK sk(out, s, segmark, inc_lds, lane, 8192u);
            if (!s.done && !s.ovf) {
                // alone from here, to the end of the stream: the loop of alz_decode_queue_kernel
                InCache in; in.init_at(src, src_len, inc_lds, lane, QCH, s.p);
                for (;;) {
                    if (s.p + QAHEAD <= src_len && !s.done) {
                        sk.ensure(in, s.p, QCH);
                        if (sk.nt) { sk.flush(); if (s.ovf) break; }
                        if (prs_lane_parse<SK, BIG>(in, sk, s, stage, lane, fl)) { if (s.ovf || s.done) break; continue; }
                    }
                    const bool tail = s.p + QAHEAD > src_len;
                    prs_from_norm<BIG>(fl, s.bits, s.flag);
                    dec_prs_serial<SK, BIG>(in, sk, s, src_len, tail ? 0xFFFFFFFFu : 1u);
                    fl = prs_to_norm<BIG>(s.bits, s.flag);
                    if (tail || s.eof || s.ovf || s.bad || s.done) break;
                }
            }
            sk.flush();
            out.finish();
            write_result(&results[sid], lane, out, s.p, resolve_status(s, false, out.produced, 0u, cap), src_len, 0u);
            if (!last) flagv = 2u;
        }
    }
    if (flagv) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) chunk_store32(flags + (size_t)oslot * ALZ_CHUNK_FLAG_WORDS, (epoch << 2) | flagv);
    }
}

static thread_local u32 t_batch_total = 0;     // streams of the whole batch the current launch belongs to (alz_launch_decode)
static thread_local int t_variant = 0;         // alz_ctx_set_kernel_variant: 0 automatic, 1 one wavefront per stream, 2 two where such a kernel exists

// PRS: two wavefronts per stream (alz_decode_prs2_kernel) unless the context asks for one
static bool prs_two_waves(u32 count) { (void)count; return t_variant != 1; }
// the flag-byte formats: two wavefronts per stream for launches that cannot fill the GPU with one -- below this many streams in the
// whole batch a launch is bound by the time of ONE stream (a mixed batch fills the GPU with all its formats together)
// Up to this many streams in a launch two wavefronts share a stream: 12 per CU (3 072 on the 256 CUs of an MI355X, where the crossings were measured), from the current device's CU count
uint32_t alz_two_wave_max(void) {
    static uint32_t cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 3072u;
    if (!cached[dev]) { int cus = 0; cached[dev] = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? 12u * (uint32_t)cus : 3072u; }
    return cached[dev];
}
static bool fast_two_waves() { return t_variant == 2 || (t_variant == 0 && t_batch_total <= alz_two_wave_max()); }

template <int FMT, bool FB>
static hipError_t launch_serial(hipStream_t stream, const u8* src, u8* dst, const alz_stream* streams, const u32* index, u32 count,
                                alz_result* results, const alz_lz_properties& lz, u32 lw, int ncaches) {
    size_t lds = lw + (size_t)ncaches * ALZ_INCACHE_BYTES;
    hipLaunchKernelGGL((alz_decode_serial_kernel<FMT, FB>), dim3(count), dim3(64), lds, stream, src, dst, streams, index, count, results, lz, lw);
    return hipGetLastError();
}

template <int FMT, int LWMAX = 4096, bool FBK = false>
static hipError_t launch_fast(hipStream_t stream, const u8* src, u8* dst, const alz_stream* streams, const u32* index, u32 count,
                              alz_result* results, const alz_lz_properties& lz, u32 lw, int ncaches, const u32* gate = nullptr) {
    (void)ncaches;
    // launches that cannot fill the GPU with one wavefront per stream: two per stream (alz_decode_fast2_kernel)
    if constexpr (LWMAX == 4096 && !FBK && (FMT == ALZ_FMT_LZSS || FMT == ALZ_FMT_LZ10 || FMT == ALZ_FMT_LZ11 || FMT == ALZ_FMT_LZ40 || FMT == ALZ_FMT_CLZ0 || FMT == ALZ_FMT_YAZ0)) {
        if (fast_two_waves()) {
            hipLaunchKernelGGL((alz_decode_fast2_kernel<FMT>), dim3(count), dim3(128), 0, stream, src, dst, streams, index, count, results, lz, lw, gate);
            return hipGetLastError();
        }
    }
    if constexpr (LWMAX == 4096 && !FBK && (FMT == ALZ_FMT_YAY0 || FMT == ALZ_FMT_MIO0)) {
        if (fast_two_waves()) {
            hipLaunchKernelGGL((alz_decode_fast2c_kernel<FMT>), dim3(count), dim3(128), 0, stream, src, dst, streams, index, count, results, lw, gate);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((alz_decode_fast_kernel<FMT, LWMAX, FBK>), dim3((count + ALZ_WPB - 1) / ALZ_WPB), dim3(64 * ALZ_WPB), 0, stream, src, dst, streams, index, count, results, lz, lw, gate);
    return hipGetLastError();
}

// LZ4 / LZO / Snappy: two wavefronts per stream (alz_decode_queue2_kernel) for launches that cannot fill the GPU -- there a launch takes
// as long as one stream, and parse and copy overlapping cut that by a fifth (256 KiB alone: LZ4 1.70 -> 1.37 ms).  A full GPU is
// bound by throughput, and there the pair loses: the copying wavefront is the longer half and the parsing one waits for it (10 000
// streams: LZ4 7.2 against 5.5 ms, LZO 9.2 against 8.2, Snappy 8.2 against 7.2).
#ifndef ALZ_QUEUE2_MAX
#define ALZ_QUEUE2_MAX 3072u   /* 2 500 streams: LZ4 2.15 against 2.44 ms, LZO 2.97 / 3.48, Snappy 2.45 / 3.17; 4 000: 3.59 / 2.74, 4.46 / 4.02, 3.97 / 3.54 */
#endif
static bool queue_two_waves() { return t_variant == 2 || (t_variant == 0 && t_batch_total <= (ALZ_QUEUE2_MAX == 3072u ? alz_two_wave_max() : ALZ_QUEUE2_MAX)); }
template <int FMT>
static hipError_t launch_queue2(hipStream_t stream, const u8* src, u8* dst, const alz_stream* streams, const u32* index, u32 count, alz_result* results, const u32* gate = nullptr) {
    hipLaunchKernelGGL((alz_decode_queue2_kernel<FMT>), dim3(count), dim3(128), 0, stream, src, dst, streams, index, count, results, gate);
    return hipGetLastError();
}

template <int FMT>
static hipError_t launch_queue(hipStream_t stream, const u8* src, u8* dst, const alz_stream* streams, const u32* index, u32 count, alz_result* results) {
    hipLaunchKernelGGL((alz_decode_queue_kernel<FMT>), dim3(count), dim3(64), 0, stream, src, dst, streams, index, count, results);
    return hipGetLastError();
}

// test / tuning hook: resident workgroups (= waves) per CU of the production kernel of `fmt`
int alz_kernel_occupancy(int fmt) {
    int n = 0; hipError_t e = hipErrorInvalidValue;
    switch (fmt) {
    case ALZ_FMT_LZSS: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_LZSS>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_LZ10: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_LZ10>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_LZ11: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_LZ11>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_LZ40: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_LZ40>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_YAZ0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_YAZ0>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_YAY0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_YAY0>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_MIO0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_MIO0>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_PRS_BE: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_PRS_BE>, 64, 0); break;
    case ALZ_FMT_PRS_LE: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_PRS_LE>, 64, 0); break;
    case ALZ_FMT_LZ4_BLOCK: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_LZ4_BLOCK>, 64, 0); break;
    case ALZ_FMT_LZO: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_LZO>, 64, 0); break;
    case ALZ_FMT_SNAPPY_RAW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_SNAPPY_RAW>, 64, 0); break;
    case ALZ_FMT_FASTLZ: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_FASTLZ>, 64, 0); break;
    case ALZ_FMT_CNX2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_CNX2>, 64, 0); break;
    case ALZ_FMT_CNS: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_CNS>, 64, 0); break;
    case ALZ_FMT_REFPACK: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_REFPACK>, 64, 0); break;
    case ALZ_FMT_WFLZ: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_WFLZ>, 64, 0); break;
    case ALZ_FMT_LZSHREK: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_LZSHREK>, 64, 0); break;
    case ALZ_FMT_HIG: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_HIG>, 64, 0); break;
    case ALZ_FMT_WFLZ_BE: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_queue_kernel<ALZ_FMT_WFLZ_BE>, 64, 0); break;
    case ALZ_FMT_BLZ: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_BLZ, 8192>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_CLZ0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_CLZ0>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_LZ02: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_LZ02>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_LZHUDSON: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_LZHUDSON>, 64 * ALZ_WPB, 0); break;
    case ALZ_FMT_SMSR00: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_fast_kernel<ALZ_FMT_SMSR00>, 64 * ALZ_WPB, 0); break;
    default: break;
    }
    return e == hipSuccess ? n : -1;
}

hipError_t alz_launch_decode_gated(int fmt, hipStream_t stream, const void* src, void* dst, const alz_stream* streams, const u32* index,
                                   u32 count, alz_result* results, const alz_lz_properties* lzp, const u32* gate) {
    t_batch_total = count; t_variant = 0;
    const u8* s = (const u8*)src; u8* d = (u8*)dst;
    const alz_lz_properties lz = *lzp;
    switch (fmt) {
    case ALZ_FMT_YAY0: return launch_fast<ALZ_FMT_YAY0>(stream, s, d, streams, index, count, results, lz, 4096, 3, gate);
    case ALZ_FMT_MIO0: return launch_fast<ALZ_FMT_MIO0>(stream, s, d, streams, index, count, results, lz, 4096, 3, gate);
    case ALZ_FMT_LZ10: return launch_fast<ALZ_FMT_LZ10>(stream, s, d, streams, index, count, results, lz, 4096, 1, gate);
    case ALZ_FMT_LZ11: return launch_fast<ALZ_FMT_LZ11>(stream, s, d, streams, index, count, results, lz, 4096, 1, gate);
    case ALZ_FMT_LZ40: return launch_fast<ALZ_FMT_LZ40>(stream, s, d, streams, index, count, results, lz, 4096, 1, gate);
    case ALZ_FMT_CLZ0: return launch_fast<ALZ_FMT_CLZ0>(stream, s, d, streams, index, count, results, lz, 4096, 1, gate);
    case ALZ_FMT_YAZ0: return launch_fast<ALZ_FMT_YAZ0>(stream, s, d, streams, index, count, results, lz, 4096, 1, gate);
    case ALZ_FMT_PRS_BE: hipLaunchKernelGGL((alz_decode_prs2_kernel<ALZ_FMT_PRS_BE>), dim3(count), dim3(128), 0, stream, s, d, streams, index, count, results, gate); return hipGetLastError();
    case ALZ_FMT_PRS_LE: hipLaunchKernelGGL((alz_decode_prs2_kernel<ALZ_FMT_PRS_LE>), dim3(count), dim3(128), 0, stream, s, d, streams, index, count, results, gate); return hipGetLastError();
    case ALZ_FMT_LZ4_BLOCK: return launch_queue2<ALZ_FMT_LZ4_BLOCK>(stream, s, d, streams, index, count, results, gate);   // (a lone stream: the two-wavefront shape)
    case ALZ_FMT_SNAPPY_RAW: return launch_queue2<ALZ_FMT_SNAPPY_RAW>(stream, s, d, streams, index, count, results, gate);
    case ALZ_FMT_LZO: return launch_queue2<ALZ_FMT_LZO>(stream, s, d, streams, index, count, results, gate);
    case ALZ_FMT_LZSS: {                                      // (the same choice of window as alz_launch_decode)
        const u32 W = 1u << lz.window_bits;
        if (W <= 4096 && lz.max_distance == W) return launch_fast<ALZ_FMT_LZSS, 4096>(stream, s, d, streams, index, count, results, lz, W, 1, gate);
        if (W <= 8192 && lz.max_distance == W) return launch_fast<ALZ_FMT_LZSS, 8192>(stream, s, d, streams, index, count, results, lz, W, 1, gate);
        if (W <= 65536 && lz.max_distance == W) return launch_fast<ALZ_FMT_LZSS, 4096, true>(stream, s, d, streams, index, count, results, lz, W, 1, gate);
        return hipErrorInvalidValue;
    }
    default: return hipErrorInvalidValue;
    }
}

int alz_chunk_places_per_cu(int fmt) {
    int n = 0;
    if (fmt == ALZ_FMT_PRS_BE) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_prs2_kernel<ALZ_FMT_PRS_BE>, 128, 0) != hipSuccess) n = 0; return n; }
    if (fmt == ALZ_FMT_PRS_LE) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, alz_decode_prs2_kernel<ALZ_FMT_PRS_LE>, 128, 0) != hipSuccess) n = 0; return n; }
    return alz_kernel_occupancy(fmt);
}
bool alz_chunk_format(int fmt, const alz_lz_properties* lz, uint32_t* lw_out) {
    u32 lw = 4096;
    switch (fmt) {
    case ALZ_FMT_LZSS: { const u32 W = 1u << lz->window_bits; if (W > 4096u || W < 256u || lz->max_distance != W) return false; lw = W; break; }
    case ALZ_FMT_LZ10: case ALZ_FMT_LZ11: case ALZ_FMT_LZ40: case ALZ_FMT_CLZ0: case ALZ_FMT_YAZ0: case ALZ_FMT_YAY0: case ALZ_FMT_MIO0: break;
    case ALZ_FMT_PRS_BE: case ALZ_FMT_PRS_LE: lw = 8192; break;
    default: return false;
    }
    if (lw_out) *lw_out = lw;
    return true;
}
template <int FMT>
static hipError_t launch_fastq(hipStream_t stream, const u8* s, u8* d, const alz_stream* streams, const alz_chunk_item* items, u32 n_items, const alz_queue_bounds& qb, alz_result* results,
                               const alz_lz_properties& lz, u32 lw, u32* ctl, u32* flags, u8* slots, u32* tmo, u32 epoch) {
    const u32 grid = n_items;                              // one workgroup per (stream, chunk) item, in queue order
    hipLaunchKernelGGL((alz_decode_fastq_kernel<FMT>), dim3(grid), dim3(64), 0, stream, s, d, streams, items, qb, results, lz, lw, ctl, flags, slots, tmo, epoch);
    return hipGetLastError();
}
hipError_t alz_launch_decode_chunked(int fmt, hipStream_t stream, const void* src, void* dst, const alz_stream* streams, const alz_chunk_item* items,
                                     u32 n_items, const alz_queue_bounds* bounds, alz_result* results, const alz_lz_properties* lzp, u32* ctl, u32* flags, u8* slots, u32* tmo, u32 epoch) {
    if (n_items == 0) return hipSuccess;
    u32 lw = 0;
    if (!alz_chunk_format(fmt, lzp, &lw)) return hipErrorInvalidValue;
    const u8* s = (const u8*)src; u8* d = (u8*)dst;
    const alz_lz_properties lz = *lzp;
    const alz_queue_bounds qb = *bounds;
    if (qb.off[0] != 0u || qb.off[ALZ_QUEUE_SHARDS] != n_items || epoch == 0u || epoch > 0x3FFFFFFFu) return hipErrorInvalidValue;
    switch (fmt) {
    case ALZ_FMT_LZSS: return launch_fastq<ALZ_FMT_LZSS>(stream, s, d, streams, items, n_items, qb, results, lz, lw, ctl, flags, slots, tmo, epoch);
    case ALZ_FMT_LZ10: return launch_fastq<ALZ_FMT_LZ10>(stream, s, d, streams, items, n_items, qb, results, lz, lw, ctl, flags, slots, tmo, epoch);
    case ALZ_FMT_LZ11: return launch_fastq<ALZ_FMT_LZ11>(stream, s, d, streams, items, n_items, qb, results, lz, lw, ctl, flags, slots, tmo, epoch);
    case ALZ_FMT_LZ40: return launch_fastq<ALZ_FMT_LZ40>(stream, s, d, streams, items, n_items, qb, results, lz, lw, ctl, flags, slots, tmo, epoch);
    case ALZ_FMT_CLZ0: return launch_fastq<ALZ_FMT_CLZ0>(stream, s, d, streams, items, n_items, qb, results, lz, lw, ctl, flags, slots, tmo, epoch);
    case ALZ_FMT_YAZ0: return launch_fastq<ALZ_FMT_YAZ0>(stream, s, d, streams, items, n_items, qb, results, lz, lw, ctl, flags, slots, tmo, epoch);
    case ALZ_FMT_YAY0: return launch_fastq<ALZ_FMT_YAY0>(stream, s, d, streams, items, n_items, qb, results, lz, lw, ctl, flags, slots, tmo, epoch);
    case ALZ_FMT_MIO0: return launch_fastq<ALZ_FMT_MIO0>(stream, s, d, streams, items, n_items, qb, results, lz, lw, ctl, flags, slots, tmo, epoch);
    case ALZ_FMT_PRS_BE: hipLaunchKernelGGL((alz_decode_prs2q_kernel<ALZ_FMT_PRS_BE>), dim3(n_items), dim3(128), 0, stream, s, d, streams, items, qb, results, ctl, flags, slots, tmo, epoch); return hipGetLastError();
    case ALZ_FMT_PRS_LE: hipLaunchKernelGGL((alz_decode_prs2q_kernel<ALZ_FMT_PRS_LE>), dim3(n_items), dim3(128), 0, stream, s, d, streams, items, qb, results, ctl, flags, slots, tmo, epoch); return hipGetLastError();
    default: return hipErrorInvalidValue;
    }
}

hipError_t alz_launch_decode(int fmt, hipStream_t stream, const void* src, void* dst, const alz_stream* streams, const u32* index,
                             u32 count, alz_result* results, const alz_lz_properties* lzp, bool exact, u32 batch_total, int variant) {
    if (count == 0) return hipSuccess;
    t_batch_total = batch_total > count ? batch_total : count;
    t_variant = variant;
    const u8* s = (const u8*)src; u8* d = (u8*)dst;
    alz_lz_properties lz = *lzp;
    if (!exact) {
        switch (fmt) {   // lane-parallel kernels
        case ALZ_FMT_LZSS: {
            u32 W = 1u << lz.window_bits;
            // LDS window sized for the geometry: 4 KiB windows (the LZSS default and every wrapper) keep 24 waves per CU
            if (W <= 4096 && lz.max_distance == W) return launch_fast<ALZ_FMT_LZSS, 4096>(stream, s, d, streams, index, count, results, lz, W, 1);
            if (W <= 8192 && lz.max_distance == W) return launch_fast<ALZ_FMT_LZSS, 8192>(stream, s, d, streams, index, count, results, lz, W, 1);
            if (W <= 65536 && lz.max_distance == W) return launch_fast<ALZ_FMT_LZSS, 4096, true>(stream, s, d, streams, index, count, results, lz, W, 1);   // 14..16 window bits (LzProperties.cs:57-66)
            break;
        }
        case ALZ_FMT_LZ10: return launch_fast<ALZ_FMT_LZ10>(stream, s, d, streams, index, count, results, lz, 4096, 1);
        case ALZ_FMT_LZ11: return launch_fast<ALZ_FMT_LZ11>(stream, s, d, streams, index, count, results, lz, 4096, 1);
        case ALZ_FMT_LZ40: return launch_fast<ALZ_FMT_LZ40>(stream, s, d, streams, index, count, results, lz, 4096, 1);
        case ALZ_FMT_YAZ0: return launch_fast<ALZ_FMT_YAZ0>(stream, s, d, streams, index, count, results, lz, 4096, 1);
        case ALZ_FMT_YAY0: return launch_fast<ALZ_FMT_YAY0>(stream, s, d, streams, index, count, results, lz, 4096, 3);
        case ALZ_FMT_MIO0: return launch_fast<ALZ_FMT_MIO0>(stream, s, d, streams, index, count, results, lz, 4096, 3);
        case ALZ_FMT_SMSR00: return launch_fast<ALZ_FMT_SMSR00>(stream, s, d, streams, index, count, results, lz, 4096, 2);
        case ALZ_FMT_LZHUDSON: return launch_fast<ALZ_FMT_LZHUDSON>(stream, s, d, streams, index, count, results, lz, 4096, 1);
        case ALZ_FMT_PRS_BE: if (prs_two_waves(count)) { hipLaunchKernelGGL((alz_decode_prs2_kernel<ALZ_FMT_PRS_BE>), dim3(count), dim3(128), 0, stream, s, d, streams, index, count, results, (const u32*)nullptr); return hipGetLastError(); }
                             return launch_queue<ALZ_FMT_PRS_BE>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_PRS_LE: if (prs_two_waves(count)) { hipLaunchKernelGGL((alz_decode_prs2_kernel<ALZ_FMT_PRS_LE>), dim3(count), dim3(128), 0, stream, s, d, streams, index, count, results, (const u32*)nullptr); return hipGetLastError(); }
                             return launch_queue<ALZ_FMT_PRS_LE>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_LZ4_BLOCK: if (queue_two_waves()) return launch_queue2<ALZ_FMT_LZ4_BLOCK>(stream, s, d, streams, index, count, results);
                                return launch_queue<ALZ_FMT_LZ4_BLOCK>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_LZO: if (queue_two_waves()) return launch_queue2<ALZ_FMT_LZO>(stream, s, d, streams, index, count, results);
                          return launch_queue<ALZ_FMT_LZO>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_SNAPPY_RAW: if (queue_two_waves()) return launch_queue2<ALZ_FMT_SNAPPY_RAW>(stream, s, d, streams, index, count, results);
                                 return launch_queue<ALZ_FMT_SNAPPY_RAW>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_FASTLZ: return launch_queue<ALZ_FMT_FASTLZ>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_CNX2: return launch_queue<ALZ_FMT_CNX2>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_CNS: return launch_queue<ALZ_FMT_CNS>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_REFPACK: return launch_queue<ALZ_FMT_REFPACK>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_WFLZ: return launch_queue<ALZ_FMT_WFLZ>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_LZSHREK: return launch_queue<ALZ_FMT_LZSHREK>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_HIG: return launch_queue<ALZ_FMT_HIG>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_WFLZ_BE: return launch_queue<ALZ_FMT_WFLZ_BE>(stream, s, d, streams, index, count, results);
        case ALZ_FMT_BLZ: return launch_fast<ALZ_FMT_BLZ, 8192>(stream, s, d, streams, index, count, results, lz, 8192, 1);
        case ALZ_FMT_CLZ0: return launch_fast<ALZ_FMT_CLZ0>(stream, s, d, streams, index, count, results, lz, 4096, 1);
        case ALZ_FMT_LZ02: return launch_fast<ALZ_FMT_LZ02>(stream, s, d, streams, index, count, results, lz, 4096, 1);
        default: break;
        }
    }
    switch (fmt) {       // exact serial kernels
    case ALZ_FMT_LZSS: {
        u32 W = 1u << lz.window_bits;
        if (W <= 8192) return launch_serial<ALZ_FMT_LZSS, false>(stream, s, d, streams, index, count, results, lz, W, 1);
        return launch_serial<ALZ_FMT_LZSS, true>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    }
    case ALZ_FMT_LZ10: return launch_serial<ALZ_FMT_LZ10, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    case ALZ_FMT_LZ11: return launch_serial<ALZ_FMT_LZ11, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    case ALZ_FMT_LZ40: return launch_serial<ALZ_FMT_LZ40, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    case ALZ_FMT_LZHUDSON: return launch_serial<ALZ_FMT_LZHUDSON, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    case ALZ_FMT_SMSR00: return launch_serial<ALZ_FMT_SMSR00, false>(stream, s, d, streams, index, count, results, lz, 4096, 2);
    case ALZ_FMT_YAZ0: return launch_serial<ALZ_FMT_YAZ0, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    case ALZ_FMT_YAY0: return launch_serial<ALZ_FMT_YAY0, false>(stream, s, d, streams, index, count, results, lz, 4096, 3);
    case ALZ_FMT_MIO0: return launch_serial<ALZ_FMT_MIO0, false>(stream, s, d, streams, index, count, results, lz, 4096, 3);
    case ALZ_FMT_PRS_BE: return launch_serial<ALZ_FMT_PRS_BE, false>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_PRS_LE: return launch_serial<ALZ_FMT_PRS_LE, false>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_LZ4_BLOCK: return launch_serial<ALZ_FMT_LZ4_BLOCK, true>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_LZO: return launch_serial<ALZ_FMT_LZO, true>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_SNAPPY_RAW: return launch_serial<ALZ_FMT_SNAPPY_RAW, true>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_FASTLZ: return launch_serial<ALZ_FMT_FASTLZ, true>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_CNX2: return launch_serial<ALZ_FMT_CNX2, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    case ALZ_FMT_CNS: return launch_serial<ALZ_FMT_CNS, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    case ALZ_FMT_REFPACK: return launch_serial<ALZ_FMT_REFPACK, true>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_WFLZ: return launch_serial<ALZ_FMT_WFLZ, true>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_LZSHREK: return launch_serial<ALZ_FMT_LZSHREK, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    case ALZ_FMT_HIG: return launch_serial<ALZ_FMT_HIG, true>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_WFLZ_BE: return launch_serial<ALZ_FMT_WFLZ_BE, true>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_BLZ: return launch_serial<ALZ_FMT_BLZ, false>(stream, s, d, streams, index, count, results, lz, 8192, 1);
    case ALZ_FMT_CLZ0: return launch_serial<ALZ_FMT_CLZ0, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    case ALZ_FMT_LZ02: return launch_serial<ALZ_FMT_LZ02, false>(stream, s, d, streams, index, count, results, lz, 4096, 1);
    default: return hipErrorInvalidValue;
    }
}
