// alz_internal.h -- declarations shared by the kernel TU and the host TU (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "auroralz.h"

// enqueue the decode kernel of one format over `count` streams (index list selects them; NULL = 0..count-1)
hipError_t alz_launch_decode(int fmt, hipStream_t stream, const void* d_src, void* d_dst, const alz_stream* d_streams,
                             const uint32_t* d_index, uint32_t count, alz_result* d_results, const alz_lz_properties* lz);

// debug/test switch: route every format through the exact serial kernels (still GPU; used by the parity tests to
// cover both kernel families)
void alz_set_force_serial(int on);
