/*
 * auroralz.h -- C ABI of the MI355X-native batched LZ codec ("auroralz").
 *
 * This is the drop-in boundary for the LZ match-copy hot path of
 * Venomalia/AuroraLib.Compression.  The reference is pure managed C# and has
 * no FFI of its own; every entry point below replaces one managed interface
 * of the reference (cited as file:line under /root/reference) and is what a
 * C# `[DllImport("auroralz")]` shim binds.
 *
 * Rules of the ABI: extern "C", plain pointers and sizes, POD structs with
 * fixed-width fields, no ownership crosses the boundary (the caller owns all
 * src/dst buffers; the library owns its device scratch inside alz_ctx /
 * alz_plan).  Functions return 0 on success or a negative ALZ_E_* code for API
 * level failures (bad argument, HIP failure, no device).  Per-stream decode
 * outcomes are reported in alz_result.status (ALZ_ST_*), mirroring the
 * reference's exceptions.
 *
 * There is NO CPU fallback behind this ABI: every decode/encode runs on the
 * GPU through the hand-written gfx950 kernels.  Without a device alz_create()
 * fails with ALZ_E_NO_DEVICE.
 */
#ifndef AURORALZ_H
#define AURORALZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALZ_ABI_VERSION 2

/* ---- formats: the headerless bodies on the hot path ---- */
typedef enum alz_format {
    ALZ_FMT_LZSS       = 0,  /* LZSS.DecompressHeaderless    src/AuroraLib.Compression/Formats/Common/LZSS.cs:91-130   */
    ALZ_FMT_LZ10       = 1,  /* LZ10.DecompressHeaderless    src/AuroraLib.Compression.Nintendo/Nintendo/LZ10.cs:82-111 */
    ALZ_FMT_LZ11       = 2,  /* LZ11.DecompressHeaderless    src/AuroraLib.Compression.Nintendo/Nintendo/LZ11.cs:83-133 */
    ALZ_FMT_YAZ0       = 3,  /* Yaz0.DecompressHeaderless    src/AuroraLib.Compression.Nintendo/Nintendo/Yaz0.cs:91-92 (Yay0 body, one cursor) */
    ALZ_FMT_YAY0       = 4,  /* Yay0.DecompressHeaderless    src/AuroraLib.Compression.Nintendo/Nintendo/Yay0.cs:99-144 (three cursors) */
    ALZ_FMT_MIO0       = 5,  /* MIO0.DecompressHeaderless    src/AuroraLib.Compression.Nintendo/Nintendo/MIO0.cs:105-149 */
    ALZ_FMT_PRS_BE     = 6,  /* PRS.DecompressHeaderless(.., Endian.Big)    src/AuroraLib.Compression.Sega/Sega/PRS.cs:59-102 */
    ALZ_FMT_PRS_LE     = 7,  /* PRS.DecompressHeaderless(.., Endian.Little) same body, LSB-first flags + LE u16 */
    ALZ_FMT_LZ4_BLOCK  = 8,  /* LZ4.DecompressBlockHeaderless src/AuroraLib.Compression/Formats/Common/LZ4.cs:176-200 */
    ALZ_FMT_LZO        = 9,  /* LZO.DecompressHeaderless     src/AuroraLib.Compression/Formats/Common/LZO.cs:49-139 */
    ALZ_FMT_SNAPPY_RAW = 10, /* Snappy.DecompressHeaderless  src/AuroraLib.Compression/Formats/Common/Snappy.cs:205-250 */
    ALZ_FMT_LZ40       = 11, /* LZ40.DecompressHeaderless    src/AuroraLib.Compression.Nintendo/Nintendo/LZ40.cs:80-132 (also the body of LZ60) */
    ALZ_FMT_LZHUDSON   = 12, /* LZHudson.DecompressHeaderless: the Yay0 grammar on ONE stream with 32-bit big-endian flag words
                                src/AuroraLib.Compression.Nintendo/HudsonSoft/LZHudson.cs:53 */
    ALZ_FMT_SMSR00     = 13, /* SMSR00.DecompressHeaderless: u16 BE codes (16-bit masks + MIO0 tokens) | literals; aux0 = length of the
                                code section   src/AuroraLib.Compression.Nintendo/Nintendo/SMSR00.cs:70-131 */
    ALZ_FMT_FASTLZ     = 14, /* FastLZ.DecompressHeaderless (levels 1 and 2; the level is the top 3 bits of the first byte)
                                src/AuroraLib.Compression/Formats/Common/FastLZ.cs:54-160.  SURVEY.md 8f rank 4. */
    ALZ_FMT_CNX2       = 15, /* CNX2.DecompressHeaderless: 2-bit codes (skip / literal / match / literal run), 2 KiB window
                                src/AuroraLib.Compression.Sega/Sega/CNX2.cs:83-139.  SURVEY.md 8f rank 4. */
    ALZ_FMT_BLZ        = 16, /* BLZ.DecompressHeaderless in stream order: the managed code walks both spans from their ends
                                (src/AuroraLib.Compression.Nintendo/Nintendo/BLZ.cs:97-135), so `src` is the code section REVERSED and
                                `dst` receives the output REVERSED (alz_container_* does both reversals); decom_len = the length of
                                the destination span.  SURVEY.md 8f rank 4. */
    ALZ_FMT_CLZ0       = 17, /* CLZ0.DecompressHeaderless: LZSS family, flags LSB first with 1 = match, distance = 0x1000 - delta
                                src/AuroraLib.Compression-Extended/Marvelous/CLZ0.cs:64-97.  SURVEY.md 8f rank 4. */
    ALZ_FMT_CNS        = 18, /* CNS.DecompressHeaderless: control byte < 0x80 = that many literals, else a match of (c & 0x7F) + 3 bytes
                                at distance next byte + 1; 256-byte window
                                src/AuroraLib.Compression-Extended/Specialized/CNS.cs:77-108.  SURVEY.md 8f rank 4. */
    ALZ_FMT_LZ02       = 19, /* LZ02.DecompressHeaderless: flags MSB first, 1 = match (DDDDLLLL DDDDDDDD [+ length byte]), ends at the
                                terminator token, not at the declared size
                                src/AuroraLib.Compression-Extended/Camelot/LZ02.cs:77-115.  SURVEY.md 8f rank 4. */
    ALZ_FMT_REFPACK    = 20, /* RefPack.DecompressHeaderless: prefix byte + 1-3 data bytes = 0-3 literals + a match (three forms), 0xE0-0xFB =
                                4-112 literals, 0xFC-0xFF = 0-3 literals and the end; 128 KiB window; the declared size is only
                                compared at the end   src/AuroraLib.Compression-Extended/EA/RefPack.cs:177-245.  SURVEY.md 8f rank 4. */
    ALZ_FMT_WFLZ       = 21, /* WFLZ.DecompressHeaderless(.., Endian.Little): 4-byte blocks (u16 distance, length - 4, literal count) each
                                followed by its literals; block 0/0/0 ends the stream
                                src/AuroraLib.Compression-Extended/WayForward/WFLZ.cs:130-159.  SURVEY.md 8f rank 4. */
    ALZ_FMT_WFLZ_BE    = 22, /* WFLZ.DecompressHeaderless(.., Endian.Big): the same body with big-endian distance words */
    ALZ_FMT_LZSHREK    = 23, /* LZShrek.DecompressHeaderless: groups of (flag: literal count | match count - 1) + literals + 1..8 matches with
                                variable-length length / distance fields; ends at a zero length byte; 4 KiB window (a distance beyond it --
                                encodable, but the managed decoder wraps it around its ring -- is BAD_TOKEN)
                                src/AuroraLib.Compression-Extended/Activision/LZShrek.cs:73-119.  SURVEY.md 8f rank 4. */
    ALZ_FMT_HIG        = 24, /* HIG.DecompressHeaderless: an initial literal block, then matches (three forms, lengths to 65 535, 32 KiB
                                window) each followed by 0 / 1 / 2 / counted literals
                                src/AuroraLib.Compression-Extended/Specialized/HIG.cs:126-212.  SURVEY.md 8f rank 4. */
    ALZ_FMT_COUNT      = 25
} alz_format;

/* ---- per-stream status: the reference's exception types ---- */
typedef enum alz_status {
    ALZ_ST_OK                   = 0,
    ALZ_ST_INPUT_TRUNCATED      = 1, /* EndOfStreamException / IndexOutOfRangeException */
    ALZ_ST_OUTPUT_SIZE_MISMATCH = 2, /* DecompressedSizeException (LZ10.cs:107-110 '>', LZSS.cs:126-129 '!=') */
    ALZ_ST_OUTPUT_CAPACITY      = 3, /* NotSupportedException of a fixed-size destination */
    ALZ_ST_BAD_TOKEN            = 4  /* reference-undefined input the library refuses: see E3 below */
} alz_status;

/*
 * E3 -- a match distance beyond the window W of its format (encodable by Snappy's 4-byte-offset copy against its 64 KiB
 * LzWindows, Snappy.cs:244-247 / :213, and by LZShrek's distances up to 65 822 against its 4 KiB one) -- is REFUSED with
 * ALZ_ST_BAD_TOKEN; every other body cannot encode one.  A decision against SURVEY.md 8a-1, which froze E3 as "d mod W in
 * Release": reading LzWindows.BackCopy (IO/LzWindows.cs:72-100) and InternWrite (:192-227) shows that the masked source
 * position (`srcPos = (_Position - distance) & mask`) is what "d mod W" describes only while the masked distance is at least the
 * chunk of the pass (`chunk = min(length, distance, W - srcPos)`: for distance > W nothing keeps it below the masked distance).
 * Otherwise BackCopy hands Unsafe.CopyBlockUnaligned (cpblk) a source span that OVERLAPS its destination inside the ring -- a
 * case ECMA-335 leaves unspecified (forward copy on one runtime, memmove on another): the managed Release output for such a token
 * is not a function of the stream, so there is nothing to be bit-exact with.  Debug builds assert (:75).  The oracle and both kernel
 * families agree on the refusal (tests/cases.py, tests/golden/kat_*.json); valid streams never contain such a token.
 * E1 (distance 0 = W), E2 (sources before the stream start read 0x00), E4 / E5 (overshoot of the declared size / of dst_cap) are
 * as SURVEY.md 8a-1 froze them.
 */

/* ---- API-level error codes ---- */
#define ALZ_OK            0
#define ALZ_E_INVALID    -1  /* bad argument */
#define ALZ_E_NO_DEVICE  -2  /* no HIP device / runtime */
#define ALZ_E_HIP        -3  /* a HIP call failed; see alz_last_error() */
#define ALZ_E_NOMEM      -4
#define ALZ_E_UNSUPPORTED -5
#define ALZ_E_FORMAT     -6  /* container header invalid (InvalidIdentifierException) */
#define ALZ_E_STREAM     -7  /* single-stream helper: per-stream status != OK (status is returned separately) */
#define ALZ_E_CHECKSUM   -8  /* LZ4 frame block / content checksum mismatch (InvalidDataException, LZ4.Frame.cs:22-28) */

/*
 * LzProperties of the generic LZSS body (src/AuroraLib.Compression/LzProperties.cs:9-97).
 * Only consulted for ALZ_FMT_LZSS streams; all other formats have fixed geometry.
 * Defaults (all zero) mean LZSS.DefaultProperties = LzProperties((byte)12, 4, 2)
 * (LZSS.cs:33): window_bits 12, length_bits 4, min_length 3, windows_start 0xFEE.
 */
typedef struct alz_lz_properties {
    uint8_t  window_bits;    /* LzProperties.WindowsBits  (7..16 supported on the GPU path) */
    uint8_t  length_bits;    /* LzProperties.LengthBits   */
    uint8_t  min_length;     /* LzProperties.MinLength    */
    uint8_t  reserved0;
    uint32_t windows_start;  /* LzProperties.WindowsStart */
    uint32_t max_distance;   /* LzProperties.MaxDistance  (== 1 << window_bits for the bit-based ctor) */
    uint32_t reserved1;
} alz_lz_properties;

/*
 * One stream of a batch.  Offsets are relative to the src_base / dst_base
 * passed to the batch call, so one descriptor table serves host and device
 * resident buffers alike.
 *
 * decom_len : the `decomLength` argument of the reference's DecompressHeaderless
 *             (LZSS/LZ10/LZ11/LZ40/YAZ0/YAY0/MIO0/LZHUDSON/SMSR00).  Ignored by PRS/LZ4/LZO (no size
 *             field, terminated by token / end of input) and by SNAPPY_RAW
 *             (varint inside the body).
 * aux0/aux1 : YAY0/MIO0: compressedDataPointer / uncompressedDataPointer
 *             relative to the first flag byte (Yay0.cs:60, MIO0.cs:61).
 *             SMSR00: aux0 = bytes of the code section (the literal section follows it).
 *             LZ4_BLOCK: aux0 = history, the number of bytes in front of dst_off
 *             (<= dst_off) that are earlier output of the same LZ4 frame and may
 *             be referenced by matches -- one LzWindows serves all blocks of a
 *             frame (LZ4.Frame.cs:120).  0 = a fresh window (LZ4.cs:164).
 * dst_cap   : bytes the library may write at dst_off.  Never exceeded.
 */
typedef struct alz_stream {
    uint64_t src_off;
    uint64_t dst_off;
    uint32_t src_len;
    uint32_t dst_cap;
    uint32_t decom_len;
    uint32_t aux0;
    uint32_t aux1;
    uint32_t format;     /* alz_format */
} alz_stream;

typedef struct alz_result {
    uint32_t dst_len;    /* bytes written at dst_off (<= dst_cap) */
    uint32_t src_used;   /* source bytes consumed; the reference leaves source.Position there
                            (Yay0.cs:89-90, MIO0.cs:92-93, LZSS.cs:68).  OK / OUTPUT_SIZE_MISMATCH / BAD_TOKEN: just behind the
                            last token read.  INPUT_TRUNCATED: src_len (the reader ran into the end of the input).
                            OUTPUT_CAPACITY: unspecified (the managed code throws from inside LzWindows' write to the
                            caller's fixed-size stream; nothing reads Position after that). */
    int32_t  status;     /* alz_status */
    uint32_t reserved;
} alz_result;

/* CompressionSettings (src/AuroraLib.Compression/CompressionSettings.cs:11-84) */
typedef struct alz_settings {
    int32_t quality;          /* 0..15; presets Fastest 0, Fast 4, Balanced 8 (default), High 12, Maximum 15 */
    int32_t max_window_bits;  /* 0 = auto.  The managed finder only ever WIDENS its window with it (windowsBits = max(format,
                               * MaxWindowBits), maxDistance = max(format, 1 << MaxWindowBits); MatchFinder/LzChainMatchFinder.cs:
                               * 69-73), so a value within the format's own window is accepted and changes nothing.  A larger one
                               * is taken for FastLZ (> 13 selects level 2 for sources >= 64 KiB at quality > 4, Formats/Common/
                               * FastLZ.cs:169-175) up to 20 -- the finder on the device keeps a distance in 21 bits -- and refused
                               * (ALZ_E_UNSUPPORTED) otherwise: the managed finder would return distances the format cannot store */
    int32_t strategy;         /* 0 Default, 1 CompatibilityMode (no self-overlapping matches) */
    int32_t min_distance;     /* 0 = format default; 2 = LZ10/LZ11 GbaVramCompatibilityMode (LZ10.cs:25-33) */
} alz_settings;

typedef struct alz_ctx  alz_ctx;   /* one HIP device + one HIP stream; single-threaded */
typedef struct alz_plan alz_plan;  /* a prepared batch: descriptors resident in HBM, grouped per format */

/* ---------------------------------------------------------------- context */
int         alz_abi_version(void);
int         alz_device_count(void);
int         alz_create(int device, alz_ctx** out);
void        alz_destroy(alz_ctx* ctx);
const char* alz_last_error(void);          /* thread-local text of the last failure */
int         alz_device_info(alz_ctx* ctx, char* name, size_t name_cap, int* cu_count, uint64_t* hbm_bytes);
/* Kernel family of a context.  0 (default): the lane-parallel production kernels.  1: the exact kernels -- one token at a
 * time, the statement-for-statement GPU restatement of the managed bodies; every lane-parallel kernel hands its stream
 * tails and error paths to them.  A verification mode (the parity tests run every case through both families). */
int         alz_ctx_set_exact_kernels(alz_ctx* ctx, int on);
/* Several production kernels exist in two shapes: one wavefront per stream (a full GPU: most streams resident) and two --
 * one parses, one copies -- for launches that cannot fill the GPU anyway (a lone stream, a small batch).  0 (default): the
 * library chooses by the size of the batch; 1 / 2: always the one- / two-wavefront shape where both exist.  Results are
 * identical; a tuning and verification hook (the library reads no environment variable for kernel selection).
 * A third shape exists for the flag-byte family (LZSS with windows up to 4 KiB, LZ10, LZ11, LZ40, CLZ0, Yaz0, Yay0, MIO0) and PRS on device-resident
 * plans: the batch as a WORK QUEUE of (stream, chunk) items -- 40 KiB of output per chunk, PRS 80 KiB --, ONE workgroup per item, each drawing its item as a ticket from the queue head
 * when it starts -- taken by itself (variant 0) when a plan has more streams of such a format than 0.4 of what the GPU holds of that format's wavefronts (and more than fit the two-wavefront
 * shape: half the GPU's places; PRS: more than the GPU holds), so that the launch does not end in a partly filled round; 3: plans created in this
 * mode use it whatever their size (the parity tests).  Same results: a chunk ends between two iterations of the lane-parallel loop and hands
 * the LDS window and the cursors on.  An item waits for the chunk before its own; tickets are drawn in start order, so that chunk is always in the hands of a workgroup that is
 * already running (progress does not depend on the order in which the hardware dispatches workgroups).  The wait is still a BOUNDED spin; if one ever ran out (a fault: never
 * seen) a sticky word of the plan is set, and the gated launch that alz_plan_execute enqueues behind every queue launch decodes that format's streams again with one wavefront
 * per stream -- on the same stream, in order: work the caller chained behind the execute sees whole output either way.  alz_plan_results then moves the plan off the queue. */
int         alz_ctx_set_kernel_variant(alz_ctx* ctx, int variant);
/* ONE big stream: a batch of one -- or of a few, as long as one after the other on the whole GPU beats side by side on wavefronts of their
 * own -- LZSS / LZ10 / LZ11 / Yaz0 / Yay0 / MIO0 / PRS / LZO / LZ4-block / raw-Snappy streams (all eleven
 * north-star bodies) of at least `min_bytes` of output each (default 24 KiB -- where the two cross: ONE stream of 32 KiB takes 0.18 ms here and 0.30 on its wavefront, of 256 KiB 0.20 and 1.3; PRS / LZO / LZ4 / Snappy, which carry no size in the
 * descriptor: of dst_cap, with at least 8 KiB of input) is decoded
 * stream by stream by the whole GPU instead of by one or two wavefronts per stream (csrc/alz_big.hip).  Yay0 / MIO0 keep flags, match tokens
 * and literals in three sections (Yay0.cs:99-108, MIO0.cs:105-116): every token's cursors are prefix sums.  The other four interleave them
 * in one byte stream, but what a group of eight tokens (LZ4: a sequence, Snappy: an element, PRS: the tokens behind one flag byte, per
 * entry state of its control-bit automaton; LZO: an instruction, per class of the literal count before it) that started at byte p WOULD occupy is a function of the bytes behind p alone, so the real starts are found by list ranking over the input bytes.  Either way the copies are resolved by pointer jumping over the
 * output bytes.  So the single-stream call a format class's Decompress(Stream, Stream) makes (Interfaces/ICompressionDecoder.cs:24; the
 * reference's own benchmark is ONE 1 000 KiB stream, Benchmarks/Benchmarks/TestAllAlgorithms.cs:41-42) does not fall behind the managed
 * decoder.  Results are identical: a stream that path cannot finish -- any malformed one -- is decoded by the exact kernel behind it.
 * The ENCODER has the same switch (csrc/alz_encode_big.h): an alz_encode_batch / _device call of at most 32 buffers, each of at least
 * 8 KiB (or `min_bytes`, if that is less: the batch pipeline gives a buffer one workgroup and one wavefront, and loses from 8 KiB on) of one of these formats (and LZ40 / CLZ0 / BLZ / LZHudson; distances within 16 bits), is compressed buffer by buffer on the whole
 * GPU -- prev() on overlapping segments, the greedy / lazy parse (FindNextBestMatch, MatchFinder/LzChainMatchFinder.cs:157-212) as list ranking
 * over "where would a cursor at p go", the emission by prefix sums -- instead of by one workgroup and one wavefront per buffer: what
 * a format class's Compress(ReadOnlySpan<byte>, Stream) makes of ONE buffer (Interfaces/ICompressionEncoder.cs; the reference's benchmark
 * compresses ONE 1 000 KiB buffer, TestAllAlgorithms.cs:44-69).  The bytes are the same either way.
 * Between the two -- more buffers than that, fewer than fill the GPU with a wavefront each (1 280 / 1 536 / 512 of a format per call at quality 0 / 1-10 / 11-15), the longest
 * of at least 8 KiB: the chunks of one archive, a directory of files -- the flag-bit formats with a bounded match length (LZSS, LZ10, Yaz0, Yay0, MIO0, CLZ0, BLZ, LZHudson),
 * raw Snappy and PRS parse from one synchronisation point of the walk to the next (a position no jump crosses) with a wavefront each and emit per segment
 * (csrc/alz_encode_seg.h).  No switch: the bytes are the same, and a context in exact or forced-variant mode never goes that way.
 * min_bytes: 0 keeps the current threshold, 0xFFFFFFFF switches both paths off; launches_out (may be NULL) receives how many streams the two paths have TAKEN on this
 * context: the decode side counts on the device, where the path decides -- a stream it declines (any malformed one) is decoded by the kernel behind the gate and NOT counted --,
 * and asking waits for everything enqueued on the device so far.  Contexts in exact or forced-variant mode (alz_ctx_set_exact_kernels / alz_ctx_set_kernel_variant) never take them. */
int         alz_ctx_big_stream(alz_ctx* ctx, uint32_t min_bytes, uint64_t* launches_out);
/* The host-buffer entry points keep their device staging buffers and the encoder's scratch (per input byte: a 16- or 32-bit link
 * in a 32-bit slot -- above quality 0 a second one, the links of the finder's wider hash narrowed from 15-bit ones --, a 32-bit
 * match entry -- not at quality 0 for the formats whose search runs inside their emit kernel --, two bytes of section buffers for
 * Yay0 / MIO0 / SMSR00, one bit of start mask for the formats without a parallel emitter; the finder's head tables live in LDS) in
 * the context and only ever grow them, so that a caller
 * working through batch after batch does not pay a device allocation per call.  This returns all of it to the device (the
 * managed side has no counterpart: ArrayPool<int>.Shared keeps LzChainMatchFinder's tables the same way,
 * MatchFinder/LzChainMatchFinder.cs:85-104, :323-334). */
int         alz_ctx_release_scratch(alz_ctx* ctx);

/* ----------------------------------------------- decode: host buffers in/out
 * Replaces the loop a managed caller writes around the static
 * `DecompressHeaderless(Stream source, Stream destination, uint decomLength)`
 * bodies -- the uniform raw-decoder delegate of
 * src/AuroraLib.Compression.CLI/Commands/BruteForceCommand.cs:88-133.
 * src_base/dst_base are HOST pointers; the call uploads, decodes on the GPU,
 * downloads and returns.  `props` may be NULL (LZSS defaults). */
int alz_decode_batch(alz_ctx* ctx, const alz_lz_properties* props, uint32_t n,
                     const uint8_t* src_base, size_t src_bytes,
                     const alz_stream* streams,
                     uint8_t* dst_base, size_t dst_bytes,
                     alz_result* results);

/* The same call over several contexts = several GPUs of one node: the batch is partitioned by
 * alz_partition_batch, each share is packed, uploaded, decoded and downloaded by its own host thread on its own context
 * (one context per device; ctxs[] must not repeat a context), no collective -- streams are independent (a fresh LzWindows
 * per Decompress call, src/AuroraLib.Compression.Nintendo/Nintendo/LZ10.cs:86).  part_of_out (may be NULL) receives the
 * index into ctxs[] that decoded stream i.  Results are identical to alz_decode_batch on any one of the contexts. */
int alz_decode_batch_multi(alz_ctx* const* ctxs, uint32_t n_ctx, const alz_lz_properties* props, uint32_t n,
                           const uint8_t* src_base, size_t src_bytes,
                           const alz_stream* streams,
                           uint8_t* dst_base, size_t dst_bytes,
                           alz_result* results, uint32_t* part_of_out);
/* Greedy LPT partition of a batch over n_parts devices, balanced by decompressed bytes x a per-format cost (the host-side
 * partitioning of SURVEY.md 8e).  part_of[i] = part of stream i; part_cost[n_parts] (may be NULL) = the load of each part.
 * Pure host code: one process per GPU (bench.py --scaling strong) uses it to pick its share of ONE batch. */
int alz_partition_batch(uint32_t n, const alz_stream* streams, uint32_t n_parts, uint32_t* part_of, uint64_t* part_cost);

/* Single stream: backs ICompressionDecoder.Decompress(Stream, Stream) of one format class
 * (src/AuroraLib.Compression/Interfaces/ICompressionDecoder.cs:24) after the managed shim parsed the header. */
int alz_decode(alz_ctx* ctx, uint32_t format, const alz_lz_properties* props,
               const uint8_t* src, uint32_t src_len, uint32_t decom_len, uint32_t aux0, uint32_t aux1,
               uint8_t* dst, uint32_t dst_cap, alz_result* result);

/* -------------------------------------------- decode: device-resident batches
 * The measured path: payload already in HBM, output left in HBM.  The kernels never WRITE outside a stream's
 * [dst_off, dst_off + dst_len) (tests/test_gpu_canary.py); they READ the input in aligned 16-byte granules and, for the 64 KiB
 * formats, up to 24 bytes beyond a source position of the stream's own output, so both device buffers need 64 readable bytes
 * behind the end of their last stream -- alz_device_malloc adds that slack to every allocation by itself.
 * alz_plan_create uploads the descriptor table and groups it per format (one
 * kernel launch per format present).  alz_plan_execute only enqueues kernels
 * on `hip_stream` (a hipStream_t, or NULL for the context's own stream) and
 * does not synchronise.  d_src_base / d_dst_base are DEVICE pointers.
 * One plan may be executed again while an earlier execute is still in flight, also on another stream and into other buffers.  Two kinds of plan own MUTABLE state on the
 * device -- a plan of big streams (the whole-GPU path, alz_ctx_big_stream: its scratch) and a plan that runs a format as a work queue of chunks (alz_ctx_set_kernel_variant:
 * queue heads, hand-over flags and slots) -- so their executes are ordered one behind the other by an event, whichever streams they are enqueued on (they do not
 * overlap; a caller who wants two batches in flight creates two plans); any other plan only reads its tables.  Every plan has ONE result table: alz_plan_results
 * returns the results of the LAST execute and waits for it, whichever stream it was enqueued on. */
int  alz_plan_create(alz_ctx* ctx, const alz_lz_properties* props, uint32_t n,
                     const alz_stream* streams, alz_plan** out);
int  alz_plan_execute(alz_ctx* ctx, alz_plan* plan, const void* d_src_base, void* d_dst_base, void* hip_stream);
/* Runs `iters` executions bracketed by HIP events on the launch stream and returns the
 * mean milliseconds per execution (kernel time as seen by the device). Synchronises. */
int  alz_plan_execute_timed(alz_ctx* ctx, alz_plan* plan, const void* d_src_base, void* d_dst_base,
                            int iters, float* mean_ms);
int  alz_plan_results(alz_ctx* ctx, alz_plan* plan, alz_result* results); /* synchronises, copies n results */
void alz_plan_destroy(alz_ctx* ctx, alz_plan* plan);

/* The device-resident path over several contexts = several GPUs of one node, without host staging: what
 * alz_decode_batch_multi does for host buffers, for a caller whose compressed payload is already in HBM (and whose output stays there).
 * alz_plan_create_multi partitions the batch (part_of[i] = index into ctxs[] that decodes stream i; NULL: alz_partition_batch decides,
 * part_of_out -- may be NULL -- receives the choice) and creates one plan per context over that context's streams.  A stream's
 * src_off / dst_off are relative to the device pointers of ITS context: d_src_bases[q] / d_dst_bases[q] in alz_plan_execute_multi, which
 * only enqueues every context's kernels on that context's own stream (from the calling thread: launches are asynchronous, no host
 * threads, no collective -- streams are independent, a fresh LzWindows per Decompress call, Nintendo/LZ10.cs:86) and does not
 * synchronise.  alz_plan_results_multi waits for all of them and returns the n results in batch order.  ctxs[] must not repeat a
 * context; two contexts may share a device. */
typedef struct alz_multi_plan alz_multi_plan;
int  alz_plan_create_multi(alz_ctx* const* ctxs, uint32_t n_ctx, const alz_lz_properties* props, uint32_t n,
                           const alz_stream* streams, const uint32_t* part_of, alz_multi_plan** out, uint32_t* part_of_out);
int  alz_plan_execute_multi(alz_multi_plan* plan, const void* const* d_src_bases, void* const* d_dst_bases);
int  alz_plan_results_multi(alz_multi_plan* plan, alz_result* results);
void alz_plan_destroy_multi(alz_multi_plan* plan);

/* ------------------------------------------------------------- encode
 * Replaces the static `CompressHeaderless(ReadOnlySpan<byte>, Stream, CompressionSettings)`
 * bodies (e.g. LZSS.cs:132-160, LZ10.cs:113-137) + LzChainMatchFinder
 * (src/AuroraLib.Compression/MatchFinder/LzChainMatchFinder.cs:157-212).
 * For encode, alz_stream.src_* describe the RAW input and dst_* the compressed
 * output capacity; results[i].dst_len is the compressed size.  For YAY0/MIO0
 * the three sections are written flags|tokens|literals and aux0/aux1 of the
 * result are returned in alz_encode_aux. */
typedef struct alz_encode_aux { uint32_t aux0; uint32_t aux1; } alz_encode_aux;
int alz_encode_batch(alz_ctx* ctx, const alz_lz_properties* props, const alz_settings* settings, uint32_t n,
                     const uint8_t* src_base, size_t src_bytes,
                     const alz_stream* streams,
                     uint8_t* dst_base, size_t dst_bytes,
                     alz_result* results, alz_encode_aux* aux /* may be NULL */);

/* The same with the raw buffers already in HBM and the compressed streams left there: alz_stream.src_off / dst_off are
 * relative to the two DEVICE pointers.  What a caller that produces its input on the device uses, and what bench.py times
 * (kernels, no PCIe); alz_last_kernel_ms() reports the device time of the call.  Nothing outside [d_src_base, d_src_base +
 * src_bytes) is read and nothing outside a stream's [dst_off, dst_off + dst_cap) is written: the finder's look-ahead loads run up
 * to 32 bytes past a stream's end, so a stream that ends inside the last 64 bytes of the source buffer is searched in a scratch copy. */
int alz_encode_batch_device(alz_ctx* ctx, const alz_lz_properties* props, const alz_settings* settings, uint32_t n,
                            const void* d_src_base, size_t src_bytes,
                            const alz_stream* streams,
                            void* d_dst_base, size_t dst_bytes,
                            alz_result* results, alz_encode_aux* aux /* may be NULL */);
/* alz_encode_batch over several contexts (one per GPU), as alz_decode_batch_multi: every CompressHeaderless call builds its own
 * LzChainMatchFinder (src/AuroraLib.Compression/Formats/Common/LZSS.cs:135), so the buffers of a batch are independent; the
 * library deals them out by raw size (longest first onto the least loaded context), every context receives and returns only
 * its share, one host thread per context, no collective.  part_of_out (may be NULL): the context each stream ran on.  The
 * output is byte-identical to alz_encode_batch on one context. */
int alz_encode_batch_multi(alz_ctx* const* ctxs, uint32_t n_ctx, const alz_lz_properties* props, const alz_settings* settings, uint32_t n,
                           const uint8_t* src_base, size_t src_bytes, const alz_stream* streams,
                           uint8_t* dst_base, size_t dst_bytes, alz_result* results, alz_encode_aux* aux /* may be NULL */,
                           uint32_t* part_of_out /* may be NULL */);

/* ------------------------------------------------ device memory helpers
 * For hosts without their own HIP allocator (the C# shim, the test harness). */
int alz_device_malloc(alz_ctx* ctx, size_t bytes, void** d_ptr);
int alz_device_free(alz_ctx* ctx, void* d_ptr);
int alz_memcpy_h2d(alz_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int alz_memcpy_d2h(alz_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int alz_memset_d(alz_ctx* ctx, void* d_dst, int value, size_t bytes);
int alz_synchronize(alz_ctx* ctx);
/* Measurement helper ("achievable copy bandwidth as a second denominator"): a 16 B/lane device-to-device copy
 * kernel over `bytes`, `iters` times; *gb_per_s = (bytes read + bytes written) / time. */
int alz_measure_copy_bandwidth(alz_ctx* ctx, size_t bytes, int iters, double* gb_per_s);
/* Device time (HIP events on the launch stream) of the kernels of the last alz_plan_execute_timed (mean per execution) or
 * alz_encode_batch (hash-table resets + all encode kernels of the call) on this context. */
int alz_last_kernel_ms(alz_ctx* ctx, float* ms);

/* ------------------------------------------------ container layer (host side)
 * The managed part of the reference's format classes restated above the
 * ABI: header parse/emit + endianness retry, body on the GPU.  One function
 * per ICompressionAlgorithm member.  `container` values are alz_container. */
typedef enum alz_container {
    ALZ_C_LZSS   = 0,  /* "LZSS"+BE size+BE csize+0      LZSS.cs:53-88   */
    ALZ_C_LZ10   = 1,  /* 0x10 + u24 LE size             LZ10.cs:47-80   */
    ALZ_C_LZ11   = 2,  /* 0x11 + u24 LE size             LZ11.cs:43-81   */
    ALZ_C_YAZ0   = 3,  /* "Yaz0"+size+align+0            Yaz0.cs:58-88   */
    ALZ_C_YAY0   = 4,  /* "Yay0"+size+tokOff+litOff      Yay0.cs:50-77   */
    ALZ_C_MIO0   = 5,  /* "MIO0"+size+tokOff+litOff      MIO0.cs:51-79   */
    ALZ_C_PRS    = 6,  /* headerless                     PRS.cs:38-57    */
    ALZ_C_LZ4_LEGACY = 7, /* LZ4Legacy: 0x184C2102 + u32-sized independent 8 MiB blocks + 0xFF   LZ4Legacy.cs, LZ4.cs:96-111,120-135 */
    ALZ_C_LZO    = 8,  /* headerless                     LZO.cs:42-47    */
    ALZ_C_SNAPPY = 9,  /* framed "sNaPpY": 64 KiB chunks (one GPU batch) + masked CRC-32C   Snappy.cs:39-107 */
    /* header-only wrappers over the same bodies */
    ALZ_C_GCLZ   = 10, /* "GCLZ" + LZ10 file             src/AuroraLib.Compression.Nintendo/Nintendo/GCLZ.cs        */
    ALZ_C_CXLZ   = 11, /* "CXLZ" + LZ10 file             src/AuroraLib.Compression.Nintendo/Sega/CXLZ.cs            */
    ALZ_C_LZ_3DS = 12, /* "3DS-LZ\r\n" + LZ10 file       src/AuroraLib.Compression.Nintendo/Nintendo/3DS-LZ.cs      */
    ALZ_C_COMP   = 13, /* "COMP" + LZ11 file             src/AuroraLib.Compression.Nintendo/Sega/COMP.cs            */
    ALZ_C_YAZ1   = 14, /* Yaz0 with magic "Yaz1"         src/AuroraLib.Compression.Nintendo/Nintendo/Yaz1.cs        */
    ALZ_C_AKLZ   = 15, /* 12-byte magic + BE size + LZSS src/AuroraLib.Compression.Sega/Sega/AKLZ.cs:41-56          */
    ALZ_C_LZ01   = 16, /* "LZ01"+csize+size+0 + LZSS     src/AuroraLib.Compression.Sega/Sega/LZ01.cs:47-83          */
    ALZ_C_LZSEGA = 17, /* csize+size + LZSS              src/AuroraLib.Compression.Sega/Sega/LZSega.cs:49-68        */
    ALZ_C_LEVEL5LZSS = 18, /* "SSZL"+0+csize+size + LZSS src/AuroraLib.Compression.Nintendo/Level5/Level5LZSS.cs:42-72 */
    ALZ_C_LZON   = 19, /* "LZOn"+002FF171+BE size+csize + LZO  src/AuroraLib.Compression.Nintendo/Nintendo/LZOn.cs:41-79 */
    ALZ_C_LZ77   = 20, /* "LZ77"+type: LZ10 / LZ11 / ChunkLZ10 (independent 4 KiB chunks = one GPU batch)  Nintendo/LZ77.cs:56-153 */
    ALZ_C_LEVEL5 = 21, /* u32 type|size<<3: OnlySave / LZ10   src/AuroraLib.Compression.Nintendo/Level5/Level5.cs:62-146 */
    ALZ_C_LZ4_FRAME = 22, /* LZ4: frame 0x184D2204 (descriptor, linked or independent blocks, xxHash32 block / content
                             checksums), legacy and skippable frames, concatenated   LZ4.cs:50-93, LZ4.Frame.cs:107-215.
                             Decoding: a frame none of whose blocks reaches in front of itself (a host walk over the sequences)
                             is ONE GPU batch whatever its block-independence flag says -- the reference's own writer clears
                             the flag and compresses every block on its own --; any other frame decodes block after block */
    /* more header-only wrappers (the reference's .Extended assembly) */
    ALZ_C_MDB4   = 23, /* "MDB4"+(n+1)+n+csize+16 zero bytes + LZSS   src/AuroraLib.Compression-Extended/Specialized/MDB4.cs:33-72 */
    ALZ_C_FCMP   = 24, /* "FCMP"+n+0x12340000 + LZSS               src/AuroraLib.Compression-Extended/Marvelous/FCMP.cs:36-50    */
    ALZ_C_IECP   = 25, /* "IECP"+n + LZSS                          src/AuroraLib.Compression-Extended/Marvelous/IECP.cs:35-46    */
    ALZ_C_GCZ    = 26, /* n + LZSS (recognised by file extension only: IsMatch is always 0 here)   Konami/GCZ.cs:23-42          */
    ALZ_C_ECD    = 27, /* "ECD"+flag+BE plain/csize/size; 4 plain bytes + LZSS(10,6,2), or stored   Specialized/ECD.cs:45-109   */
    ALZ_C_SDPC   = 28, /* "SDPC"+n + LZO                           src/AuroraLib.Compression-Extended/Specialized/SDPC.cs:34-54 */
    ALZ_C_LZ40   = 29, /* 0x40 + u24 LE size + LZ40 body (negated MSB-first flag bytes, LE tokens)   Nintendo/LZ40.cs:40-77 */
    ALZ_C_LZ60   = 30, /* 0x60 + u24 LE size + the same body                                         Nintendo/LZ60.cs:29-58 */
    ALZ_C_LZHUDSON = 31, /* BE size + LZHudson body                 src/AuroraLib.Compression.Nintendo/HudsonSoft/LZHudson.cs:33-51 */
    ALZ_C_SMSR00 = 32, /* "SMSR00"+u16 0+BE size+BE literal pointer + codes | literals   Nintendo/SMSR00.cs:41-66 */
    ALZ_C_LZ00   = 33, /* "LZ00"+csize+8x0+name[32]+size+key+8x0, then an LZSS body XORed with the keystream of `key`
                          src/AuroraLib.Compression.Sega/Sega/LZ00.cs:40-96, :128-141 (see alz_container_options.key) */
    ALZ_C_FASTLZ = 34, /* headerless FastLZ stream, levels 1 / 2 (IsMatch = FastLZ.Validate; written at level 1)
                          src/AuroraLib.Compression/Formats/Common/FastLZ.cs:29-52, :246-291 */
    ALZ_C_BLZ    = 36, /* code section (stored back to front) + 0xFF padding + u24 LE total size + header size + i32 LE size delta
                          src/AuroraLib.Compression.Nintendo/Nintendo/BLZ.cs:28-95 */
    ALZ_C_CLZ0   = 37, /* "CLZ\0" + BE size + BE 0 + BE size + CLZ0 body   src/AuroraLib.Compression-Extended/Marvelous/CLZ0.cs:41-62 */
    ALZ_C_CNS    = 38, /* "@CNS" + extension[4] + LE size + 0 + CNS body   src/AuroraLib.Compression-Extended/Specialized/CNS.cs:44-75 */
    ALZ_C_LZ02   = 39, /* type byte (1 / 2) + u24 BE size + LZ02 body [+ extension data]   src/AuroraLib.Compression-Extended/Camelot/LZ02.cs:60-75 */
    ALZ_C_REFPACK = 40, /* [u32 LE compressed size] + flags + 0xFB + BE u24/u32 size [+ compressed size] + RefPack body; written with
                          the pre-header (version 2)   src/AuroraLib.Compression-Extended/EA/RefPack.cs:38-175 */
    ALZ_C_WFLZ   = 41, /* "WFLZ" + compressed size + size (FormatByteOrder, default little) + WFLZ body
                          src/AuroraLib.Compression-Extended/WayForward/WFLZ.cs:36-105 */
    ALZ_C_LZSHREK = 42, /* u32 LE 0x10 + size + compressed size + 0 + LZShrek body   src/AuroraLib.Compression-Extended/Activision/LZShrek.cs:22-71 */
    ALZ_C_HIG    = 43, /* "HIG!" + 15 ints (data offset, ..., version, size) [+ compressed size + path[0x7C] for versions 5 / 6] + HIG body
                          src/AuroraLib.Compression-Extended/Specialized/HIG.cs:47-124 */
    ALZ_C_CNX2   = 35, /* "CNX\x02" + extension[4] + BE csize + BE size + CNX2 body   src/AuroraLib.Compression.Sega/Sega/CNX2.cs:45-81 */
    ALZ_C_COUNT  = 44
} alz_container;

/* alz_container_options.variant for ALZ_C_LZ77 (LZ77.CompressionType, LZ77.cs:156-164) and ALZ_C_LEVEL5 (Level5.cs:151-159) */
#define ALZ_LZ77_LZ10      0x10u
#define ALZ_LZ77_LZ11      0x11u
#define ALZ_LZ77_CHUNKLZ10 0xF7u
#define ALZ_LEVEL5_ONLYSAVE 0u
#define ALZ_LEVEL5_LZ10     1u

typedef struct alz_container_options {
    uint32_t big_endian;          /* IEndianDependentFormat.FormatByteOrder: 1 = Endian.Big (default for Yaz0/Yay0/MIO0/PRS) */
    uint32_t memory_alignment;    /* Yaz0.MemoryAlignment (Yaz0.cs:39) */
    alz_lz_properties lz;         /* LZSS geometry */
    uint32_t variant;             /* LZ77.Type / Level5.Type when compressing; 0 = the class default (LZ10) */
    uint32_t chunk_size;          /* LZ77.ChunkSize (default 0x1000); ALZ_C_LZ4_FRAME: LZ4.BlockSize, one of 0x10000 /
                                     0x40000 / 0x100000 / 0x400000 (0 = the class default Block4MB, LZ4.cs:33) */
    uint32_t key;                 /* ALZ_C_LZ00 when compressing: the keystream seed written to the header (LZ00.Compress(..., uint key, ...)
                                     LZ00.cs:71; the parameterless overload passes the Unix time) */
    uint8_t  name[32];            /* ALZ_C_LZ00 when compressing: LZ00.Name, zero padded (all zero = the class default "Temp.dat", LZ00.cs:30) */
} alz_container_options;

/* IProvidesDecompressedSize.GetDecompressedSize (Interfaces/IProvidesDecompressedSize.cs:20) */
int alz_container_decompressed_size(uint32_t container, const alz_container_options* opt,
                                    const uint8_t* src, size_t src_len, uint32_t* size_out);
/* IFormatInfoProvider.IsMatch (e.g. LZ10.cs:36-41): 1 match, 0 no match */
int alz_container_is_match(uint32_t container, const uint8_t* src, size_t src_len);
/* ICompressionDecoder.Decompress(Stream, Stream).  On ALZ_E_STREAM, *status holds the alz_status. */
int alz_container_decompress(alz_ctx* ctx, uint32_t container, const alz_container_options* opt,
                             const uint8_t* src, size_t src_len,
                             uint8_t* dst, size_t dst_cap, size_t* dst_len, size_t* src_used, int32_t* status);
/* ICompressionEncoder.Compress(ReadOnlySpan<byte>, Stream, CompressionSettings) */
int alz_container_compress(alz_ctx* ctx, uint32_t container, const alz_container_options* opt,
                           const alz_settings* settings,
                           const uint8_t* src, size_t src_len,
                           uint8_t* dst, size_t dst_cap, size_t* dst_len);
/* worst-case compressed size for dst_cap sizing */
size_t alz_container_compress_bound(uint32_t container, size_t src_len);

/* ------------------------------------------------ batch producers
 * The two places in the reference that issue many independent decodes, restated as callers of the batched path.
 *
 * alz_container_scan: ScanDecompressCommand (src/AuroraLib.Compression.CLI/Commands/ScanDecompressCommand.cs:12-104).
 * Walks `src` byte by byte; at every offset the first container of `containers` whose IsMatch accepts identifies the
 * format (FormatService.Formats.Identify, :62); the stream is decoded and kept when it decodes without an exception
 * and yields more than 0x10 bytes (:85), the walk then continues behind it (:98), otherwise at the next byte (:100).
 * Here every candidate offset is decoded in ONE GPU batch (rounds of <= 1 GiB of output) and the walk is replayed over
 * the results.  Supported: the containers with a size header and one body (LZSS, LZ10, LZ11, YAZ0, YAY0, MIO0, GCLZ,
 * CXLZ, LZ_3DS, COMP, YAZ1, AKLZ, LZ01, LZSEGA, LEVEL5LZSS, MDB4, FCMP, IECP, LZ40, LZ60, CNX2, CLZ0, CNS, SMSR00, HIG).  A Yaz0 /
 * Yaz1 candidate that fails is decoded once more with its size field byte-swapped, as Yaz0.Decompress does (Yaz0.cs:66-78).
 * Outputs of the accepted streams are packed into dst in file order; ALZ_E_NOMEM when dst or hits is too small
 * (nhits / dst_used then describe what fitted). */
typedef struct alz_scan_hit {
    uint64_t start;      /* offset of the stream in src */
    uint64_t end;        /* source.Position after Decompress */
    uint64_t dst_off;    /* its output inside dst */
    uint32_t dst_len;
    uint32_t container;  /* alz_container that identified it */
} alz_scan_hit;
int alz_container_scan(alz_ctx* ctx, const uint32_t* containers, uint32_t n_containers, const alz_container_options* opt,
                       const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap,
                       alz_scan_hit* hits, uint32_t max_hits, uint32_t* nhits, size_t* dst_used);

/* alz_brute_force: BruteForceCommand (src/AuroraLib.Compression.CLI/Commands/BruteForceCommand.cs:24-133): one raw
 * buffer, every raw decoder of the path tried against a fixed destination of `expected_size` bytes -- one GPU batch per
 * LZSS geometry.  Decoder i writes to dst + i * slot (slot >= expected_size); it "successfully unpacked the file" (:42)
 * when results[i].status == ALZ_ST_OK and results[i].dst_len == expected_size. */
#define ALZ_BRUTE_DECODERS 19
const char* alz_brute_decoder_name(uint32_t i);   /* the names of GetRawDecodersList (:96-131), e.g. "LZSS (10, 6, 2)" */
int alz_brute_force(alz_ctx* ctx, const uint8_t* raw, size_t raw_len, uint32_t expected_size,
                    uint8_t* dst, size_t slot, alz_result* results /* [ALZ_BRUTE_DECODERS] */);

#ifdef __cplusplus
}
#endif
#endif /* AURORALZ_H */
