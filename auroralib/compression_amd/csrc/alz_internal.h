// alz_internal.h -- declarations shared by the kernel TU and the host TU (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "auroralz.h"

// enqueue the decode kernel of one format over `count` streams (index list selects them; NULL = 0..count-1)
// `exact`: the exact one-token-at-a-time kernels (alz_ctx_set_exact_kernels) instead of the lane-parallel ones
hipError_t alz_launch_decode(int fmt, hipStream_t stream, const void* d_src, void* d_dst, const alz_stream* d_streams,
                             const uint32_t* d_index, uint32_t count, alz_result* d_results, const alz_lz_properties* lz, bool exact, uint32_t batch_total = 0, int variant = 0)   /* batch_total: streams of ALL formats of the batch this launch belongs to (0 = count); variant: alz_ctx_set_kernel_variant */;
int alz_kernel_occupancy(int fmt);

// ---- the flag-byte family as a work queue of (stream, chunk) items (alz_decode_fastq_kernel, alz_kernels.hip)
struct alz_chunk_item { uint32_t sid, chunk, slot, last; };   // stream, its chunk, the hand-over slot this chunk WRITES (it reads slot - 1), 1 = the stream's last chunk
#define ALZ_CHUNK_FLAG_STRIDE 32u                             /* words between two flags (= ALZ_CHUNK_FLAG_WORDS of the kernel) */
#define ALZ_QUEUE_SHARDS 8u                                   /* a format's queue is ALZ_QUEUE_SHARDS sub-queues, each with a head word on a 128-byte line of its own: one per XCD (MI355X_MICROARCH.md, "dequeue": ONE head word
                                                                 saturates at ~88 dequeues per microsecond -- the 6 400 workgroups a launch starts with would queue up for 73 us in front of it) */
#define ALZ_CHUNK_CTL_WORDS 512u                              /* control words in front of the flags: two sets of sub-queue heads, set e & 1 of launch (epoch) e at word 256 (e & 1) + 32 q (q < 8) */
struct alz_queue_bounds { uint32_t off[ALZ_QUEUE_SHARDS + 1]; };   // items [off[q], off[q + 1]) of the item list are sub-queue q
#ifndef ALZ_CHUNK_OUT
#define ALZ_CHUNK_OUT 40960u                                  /* output bytes per chunk.  NOT a power of two: streams of 64 KiB, 256 KiB, 1 MiB then end in a SHORT last chunk, and the last chunks are what a launch
                                                                 drains at its end (10 000 x 256 KiB as Yaz0, ms per launch: 24 KiB 2.69, 32 KiB 2.55, 36 KiB 2.55, 40 KiB 2.49-2.51, 42 KiB 2.55, 48 KiB 2.56, 51 KiB 2.57 -- a last chunk of 16 KiB beats one of 4 KiB too; every queue format gains 1.5-3 % over 32 KiB, cfg2's 64 KiB streams 2 %;
                                                                 before the descriptor table: 16 KiB 2.62, 24 KiB 2.57, 32 KiB 2.58, 64 KiB 2.62, 128 KiB 2.76; one wavefront per stream 2.93) */
#endif
bool alz_chunk_format(int fmt, const alz_lz_properties* lz, uint32_t* lw_out);   // does the format have the work-queue kernel, and with which LDS window
#ifndef ALZ_CHUNK_OUT_PRS
#define ALZ_CHUNK_OUT_PRS 81920u                              /* PRS hands over an 8 KiB window: larger chunks (10 000 x 256 KiB, ms per launch: 48 KiB 5.44, 64 KiB 5.36, 80 KiB 5.31) */
#endif
static inline uint32_t alz_chunk_bytes(int fmt) { return (fmt == ALZ_FMT_PRS_BE || fmt == ALZ_FMT_PRS_LE) ? ALZ_CHUNK_OUT_PRS : ALZ_CHUNK_OUT; }
uint32_t alz_two_wave_max(void);                              // launches of at most this many streams give a stream two wavefronts (12 per CU of the current device)
int alz_chunk_places_per_cu(int fmt);                        // streams one CU holds of the format's one-workgroup-per-stream kernel (what the work queue is weighed against)
// d_ctl: ALZ_CHUNK_CTL_WORDS words (two sets of sub-queue heads) and d_flags (ALZ_CHUNK_FLAG_STRIDE words per slot: a line of its own), zeroed ONCE when the plan is made: a launch carries
// an `epoch` (1 .. 2^30 - 1, a new one per launch of the plan, launches of a plan never overlap) -- a flag counts only when it holds that epoch, launch e draws its tickets from head set e & 1
// and zeroes the other set on the way; d_slots: n_slots x (32 + lw) bytes; d_tmo: ONE sticky word, set when a bounded spin runs out -- the caller enqueues
// alz_launch_decode_gated(..., d_tmo) behind this launch, so that such a launch is repaired in stream order, and never clears the word.  The item list is ALZ_QUEUE_SHARDS
// sub-queues one behind the other (`bounds`), each in chunk-major order over ITS streams: a stream's chunks all lie in one sub-queue.
hipError_t alz_launch_decode_chunked(int fmt, hipStream_t stream, const void* d_src, void* d_dst, const alz_stream* d_streams, const alz_chunk_item* d_items,
                                     uint32_t n_items, const alz_queue_bounds* bounds, alz_result* d_results, const alz_lz_properties* lz, uint32_t* d_ctl, uint32_t* d_flags, uint8_t* d_slots, uint32_t* d_tmo, uint32_t epoch);
// the same launch gated by a device word: the kernels return at once while *d_gate == 0 (alz_big.hip: the production -- lane-parallel -- decode, with the reference's error semantics, behind the
// whole-GPU path of ONE big stream, needed only when that path declined the stream; and behind a work-queue launch, gated by its sticky timeout word).  The formats of those two paths only.
hipError_t alz_launch_decode_gated(int fmt, hipStream_t stream, const void* d_src, void* d_dst, const alz_stream* d_streams, const uint32_t* d_index,
                                   uint32_t count, alz_result* d_results, const alz_lz_properties* lz, const uint32_t* d_gate);

// ---- one big stream on the whole GPU (alz_big.hip): Yay0 / MIO0 (three sections) and LZSS / LZ10 / LZ11 / Yaz0 (one interleaved stream)
bool alz_big_eligible(int fmt, const alz_stream* st, const alz_lz_properties* lz, uint32_t min_bytes);
size_t alz_big_scratch_bytes(int fmt, const alz_stream* st);
hipError_t alz_launch_big(int fmt, hipStream_t stream, const void* d_src_base, void* d_dst_base, const alz_stream* st, const alz_lz_properties* lz,
                          alz_result* d_result, void* d_scratch, uint32_t* d_gate, uint32_t* d_accepted /* += 1 when the path takes the stream (gate stays closed) */);   // resident waves per CU of the production decode kernel (tuning aid)

// ---- encoder (alz_encode.hip)
bool alz_encode_geometry(int fmt, const alz_lz_properties* lz, const alz_settings* st, void* out_geom, int* window_bits, int variant = 0);   // variant 1: FastLZ level 2
size_t alz_encode_geom_size(void);
int alz_encode_geom_hash_bits(const void* geom);
int alz_encode_geom_min_table(const void* geom);
int alz_encode_geom_max_dist(const void* geom);
int alz_encode_geom_narrows(const void* geom);                  // 1: kernel A links at 15 bits and enc_narrow_kernel writes the finder's own links into d_narrow (which must exist)
int alz_encode_format_needs_mask(int fmt);                      // 1: the format's emitter reads the start mask of enc_roles_kernel (zeroed before the launch)
int alz_encode_geom_needs_match(int fmt, const void* geom);   // 0: the search runs inside the parse + emit kernel (no kernel B, no match array)
// a batch of few buffers of a flag-bit format: parse and emitter as five small kernels over segments (alz_encode_seg.h); 1: taken -- then the launch
// needs the match array, the zeroed start mask and alz_encode_seg_bytes(count, *kmax, *hist) of scratch
int alz_encode_segmented(int fmt, const void* geom, uint32_t count, uint32_t max_len, uint32_t max_streams /* ~0: the rule */, uint32_t* seg_len, uint32_t* kmax, uint32_t* hist);
size_t alz_encode_seg_bytes(uint32_t count, uint32_t kmax, uint32_t hist);
int alz_encode_aseg(const void* geom, uint32_t count, uint32_t max_len, uint32_t* sa, uint32_t* ka, uint32_t* w, uint32_t* stride, size_t* bytes);   // kernel A over segments for at most 128 buffers (alz_encode_seg.h)
struct alz_encode_side { hipStream_t s; hipEvent_t fork, join; };   // a second HIP stream + two events of the caller's: where alz_launch_encode puts the scan streams' kernel, beside the rest
hipError_t alz_launch_encode(int fmt, hipStream_t stream, const void* d_src, void* d_dst, const alz_stream* d_streams, const uint32_t* d_index,
                             uint32_t count, uint32_t max_len, int* d_prev4, int* d_prevm, int* d_narrow, void* d_match,
                             const uint64_t* d_pos_off, void* d_side, void* d_mask, alz_result* d_results, alz_encode_aux* d_aux, const void* geom,
                             uint32_t* d_sel /* which kernel B per stream (enc_probe_kernel): sel_pitch words indexed by stream, then 1 + count words of list; NULL: the two-phase kernel from maxChain 3 on */,
                             uint32_t sel_pitch, void* d_seg = nullptr /* alz_encode_segmented: its scratch, */, uint32_t seg_len = 0, uint32_t seg_kmax = 0 /* segment length and segments per buffer */,
                             int scan_mode = 0 /* the streams that go without kernels A and B (enc_scan_select_kernel): 0 its probe decides, 1 every eligible stream, 2 none; d_sel then has 4 sel_pitch + 128 words */,
                             uint32_t* d_scan_taken = nullptr /* += the streams that went that way */, const alz_encode_side* side_q = nullptr);
// ---- ONE big stream on the whole GPU, encoder side (alz_encode_big.h): the flag-bit formats
bool alz_encode_big_eligible(int fmt, const void* geom, const alz_stream* st, uint32_t min_bytes);
size_t alz_encode_big_scratch_bytes(int fmt, const void* geom, const alz_stream* st);
hipError_t alz_launch_encode_big(int fmt, hipStream_t stream, const void* d_src_base, void* d_dst_base, const alz_stream* st, alz_result* d_result,
                                 alz_encode_aux* d_aux, void* d_scratch, uint32_t* d_ctl /* 16 words */, const void* geom);
