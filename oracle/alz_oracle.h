/*
 * alz_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the LZ match-copy hot path of
 * Venomalia/AuroraLib.Compression.  It exists only to check the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under auroralib/ links or calls it.
 *
 * The reference is managed C#; no .NET toolchain exists in the build image, so
 * the reference itself cannot be compiled here (no oracle/_ref).  Parity is
 * pinned by the reference's own fixtures instead (tests/test_oracle_golden.py):
 *   - CompressionTest/Test.lz decoded with LZSS(LzProperties((byte)10,6,2))
 *     must hash to XXH64 11520079745250749767 (CompressionAlgorithmTest.cs:31-48);
 *   - the round-trip matrix on Test.bmp prefixes (CompressionAlgorithmTest.cs:81-130);
 *   - the published compression ratios of Benchmarks.md (encoder restatement).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).
 */
#ifndef ALZ_ORACLE_H
#define ALZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "auroralz.h" /* shared POD types only: alz_stream, alz_result, alz_lz_properties, alz_settings, enums */

#ifdef __cplusplus
extern "C" {
#endif

/* XXH64 (standard algorithm; the reference's tests use HashDepot 2.0.3). */
uint64_t oracle_xxh64(const void* data, size_t len, uint64_t seed);
uint32_t oracle_xxh32(const void* data, size_t len, uint32_t seed);
uint32_t oracle_crc32c(const void* data, size_t len);

/* Headerless decode of one stream through the restated LzWindows ring. */
void oracle_decode_stream(const alz_lz_properties* props, const alz_stream* s,
                          const uint8_t* src_base, uint8_t* dst_base, alz_result* r);

/* Same decode through a flat "out[q] = out[q-d]" model (no ring).  Used by the
 * tests to cross-check the ring restatement against the semantics the GPU
 * kernels implement. */
void oracle_decode_stream_flat(const alz_lz_properties* props, const alz_stream* s,
                               const uint8_t* src_base, uint8_t* dst_base, alz_result* r);

/* Batch decode, streams striped over `nthreads` host threads (>=1). */
int oracle_decode_batch(const alz_lz_properties* props, uint32_t n,
                        const uint8_t* src_base, const alz_stream* streams,
                        uint8_t* dst_base, alz_result* results, int nthreads);

/* Headerless encode of one buffer. Returns compressed size or a negative error
 * (-1 capacity, -2 input the reference would throw on).  For YAY0/MIO0 the
 * output is flags|tokens|literals and aux receives the two section offsets. */
int64_t oracle_encode_stream(uint32_t format, const alz_lz_properties* props, const alz_settings* settings,
                             const uint8_t* src, size_t n, uint8_t* dst, size_t cap, alz_encode_aux* aux);

int oracle_encode_batch(const alz_lz_properties* props, const alz_settings* settings, uint32_t n,
                        const uint8_t* src_base, const alz_stream* streams,
                        uint8_t* dst_base, alz_result* results, alz_encode_aux* aux, int nthreads);

/* Container layer (header parse/emit + retry), mirrors the alz_container_* ABI. */
int oracle_container_decompressed_size(uint32_t container, const alz_container_options* opt,
                                       const uint8_t* src, size_t src_len, uint32_t* size_out);
int oracle_container_decompress(uint32_t container, const alz_container_options* opt,
                                const uint8_t* src, size_t src_len,
                                uint8_t* dst, size_t dst_cap, size_t* dst_len, size_t* src_used, int32_t* status);
int oracle_container_compress(uint32_t container, const alz_container_options* opt, const alz_settings* settings,
                              const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* dst_len);

/* Geometry helper: fills LzProperties per the bit-based ctor (LzProperties.cs:57-66). */
void oracle_lz_properties_bits(uint8_t distance_bits, uint8_t length_bits, uint8_t threshold, alz_lz_properties* out);

#ifdef __cplusplus
}
#endif
#endif
