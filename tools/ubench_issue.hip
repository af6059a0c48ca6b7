// ubench_issue.hip -- the gfx950 instruction-ISSUE ceiling the decode / encode kernels are priced against.
// For 1, 2, 4, 6, 8 wavefronts per SIMD: cycles per wave64 instruction of independent and dependent chains of the instruction classes the
// kernels are made of (32-bit integer VALU, VOP3 bit-field, v_cndmask, DPP moves, ds_bpermute, LDS byte reads, v_readlane + scalar chains,
// v_cmp -> SGPR pair, plain SALU), SALU beside VALU in one wave and in neighbouring waves.
// Every wave stamps s_memtime around its loop; a row reports the median cycles per instruction of ONE wave and, from it, wave64
// instructions per cycle per SIMD and per CU (4 SIMDs).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_issue tools/ubench_issue.hip ; run on the GPU box (no arguments).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
typedef unsigned int u32; typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { OP_VADD = 0, OP_VADD_DEP, OP_VBFE, OP_VCNDMASK, OP_VMAD24, OP_VLSHLADD, OP_VPERM, OP_DPP, OP_DPP_DEP, OP_BPERMUTE, OP_BPERMUTE_DEP, OP_LDS_U8, OP_LDS_B32,
       OP_LDS_U8_DEP, OP_READLANE_DEP, OP_READLANE, OP_VCMP, OP_MBCNT, OP_SADD, OP_SADD_DEP, OP_SBFE, OP_MIX_SV, OP_SPLIT_SV, OP_WALK, OP_VADD16, OP_READFIRST, OP_SPLIT_V_ONLY, OP_VMOV, OP_VAND, OP_VLSHL, OP_VLSHR_V, OP_VMINU, OP_VMUL24, OP_VCND_VCC, OP_VCND_SGPR, OP_VADD_E64, OP_VADD_SGPR, OP_VADD_LIT, OP_VANDOR, OP_VADD3, OP_VALIGN, OP_VBCNT, OP_VMULLO, OP_VCMP_VCC, OP_VBFE_V, OP_SDWA, OP_DPP_ADD, OP_DSWRITE8, OP_DSREAD128, OP_VCND_E64_VCC, OP_VSUB, OP_VOR, OP_VXOR, OP_VMAXU, OP_VLSHL_V, OP_VLSHR_C, OP_VAND_C, OP_VADD_C, OP_VLSHLOR, OP_VLSHL16, OP_VADDC, OP_VADDCO, OP_SWIZZLE, OP_VADD_2SRC, OP_COUNT };
static const char* NAMES[OP_COUNT] = {
    "v_add_u32, 8 independent chains", "v_add_u32, ONE dependent chain", "v_bfe_u32 (VOP3), 8 chains", "v_cndmask_b32 (vcc), 8 chains", "v_mad_u32_u24, 8 chains",
    "v_lshl_add_u32, 8 chains", "v_perm_b32, 8 chains", "v_mov_b32 DPP row_shr:1, 8 chains", "v_add_u32 DPP row_shr:1, ONE dependent chain",
    "ds_bpermute_b32, 8 independent (one wait per 8)", "ds_bpermute_b32, dependent (wait each)", "ds_read_u8, 8 independent random (one wait per 8)",
    "ds_read_b32, 8 independent (one wait per 8)", "ds_read_u8 dependent (pointer chase)", "v_readlane_b32 -> s index -> v_readlane (dependent walk)",
    "v_readlane_b32, constant lanes, 8 independent", "v_cmp_lt_u32 -> SGPR pair, 8 independent", "v_mbcnt_lo + v_mbcnt_hi pairs", "s_add_u32, 8 independent chains",
    "s_add_u32, ONE dependent chain", "s_bfe_u32, 8 chains", "v_add_u32 + s_add_u32 interleaved 1:1 in one wave (per PAIR)",
    "VALU-only waves beside SALU-only waves (even / odd wave of a workgroup; per instruction of either)",
    "lane walk: v_readlane + v_writelane m0 + s_add m0 + s_add + s_and, dependent (per ELEMENT: 5 instr)", "v_add_u32, 16 independent chains", "v_readfirstlane_b32, 8 independent",
    "(control for the row above) the SAME kernel, odd waves idle: the VALU-only half alone",
    "v_mov_b32 (VOP1), 8 chains",
    "v_and_b32 (VOP2), 8 chains",
    "v_lshlrev_b32 (VOP2), 8 chains",
    "v_lshrrev_b32 by a VGPR (VOP2), 8 chains",
    "v_min_u32 (VOP2), 8 chains",
    "v_mul_u32_u24 (VOP2), 8 chains",
    "v_cndmask_b32 ..., vcc (VOP2; vcc not rewritten), 8 chains",
    "v_cndmask_b32_e64 ..., s[n:n+1] (VOP3), 8 chains",
    "v_add_u32_e64 (VOP3 encoding of a 2-operand op), 8 chains",
    "v_add_u32 v, s, v (VOP2 with an SGPR source), 8 chains",
    "v_add_u32 v, 0x12345, v (VOP2 with a 32-bit literal), 8 chains",
    "v_and_or_b32 (VOP3, 3 sources), 8 chains",
    "v_add3_u32 (VOP3, 3 sources), 8 chains",
    "v_alignbit_b32 (VOP3), 8 chains",
    "v_bcnt_u32_b32 (VOP3), 8 chains",
    "v_mul_lo_u32 (VOP3), 8 chains",
    "v_cmp_lt_u32 vcc, ... (VOPC), 8 independent",
    "v_bfe_u32 with VGPR offset (VOP3), 8 chains",
    "v_and_b32 SDWA src0_sel:BYTE_1 (VOP2 SDWA), 8 chains",
    "v_add_u32 DPP quad_perm (VOP2 DPP), 8 chains, source another chain",
    "ds_write_b8, 8 independent (one wait per 8)",
    "ds_read_b128 aligned lane-contiguous (per instruction)",
    "v_cndmask_b32_e64 ..., vcc (VOP3 encoding, mask in vcc), 8 chains",
    "v_sub_u32 (VOP2), 8 chains",
    "v_or_b32 (VOP2), 8 chains",
    "v_xor_b32 (VOP2), 8 chains",
    "v_max_u32 (VOP2), 8 chains",
    "v_lshlrev_b32 by a VGPR (VOP2), 8 chains",
    "v_lshrrev_b32 by an inline constant (VOP2), 8 chains",
    "v_and_b32 with an inline constant (VOP2), 8 chains",
    "v_add_u32 with an inline constant (VOP2), 8 chains",
    "v_lshl_or_b32 (VOP3), 8 chains",
    "v_lshlrev_b16 (VOP2), 8 chains",
    "v_addc_co_u32 ..., vcc, vcc (VOP2), 8 chains",
    "v_add_co_u32 ..., vcc (VOP2), 8 chains",
    "ds_swizzle_b32 (one wait per 8), 8 independent",
    "v_add_u32 v[j], v[j+3], v[k] (VOP2, two different VGPR sources per instruction), 8 chains" };

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
constexpr int UNROLL = 8;      // groups of 8 instructions per loop body: 64 instructions of the class per iteration

template <int OP>
__global__ __launch_bounds__(256) void k_issue(u64* cyc, u32* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ u32 lds[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 1024; i += 256) lds[i] = (i * 2654435761u) >> 22;            // values 0..1023 (pointer chase stays inside)
    if (dyn[0] == 77 && sink == nullptr) lds[0] = 1;                                     // (keeps the dynamic LDS allocation alive)
    __syncthreads();
    u32 a[16];
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = lane * 17u + j * 3u + blockIdx.x;
    u32 k = lane | 1u, acc = 0;
    u32 s[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = __builtin_amdgcn_readfirstlane(blockIdx.x + j);
    u32 sidx = __builtin_amdgcn_readfirstlane(blockIdx.x & 63);
    const u64 mask64 = 0x5555333300FF0F0Full ^ (u64)sidx;
    const u32 ldsaddr = (u32)lane * 4u;
    const bool role = (wave & 1) != 0;
    u64 t0, t1;
    asm volatile("s_mov_b64 vcc, %0" :: "s"(mask64) : "vcc");           // (a defined mask for the rows that read vcc)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (OP == OP_VADD) {
#define X(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(k));
                R8(X)
#undef X
            } else if (OP == OP_VADD16) {
#define X(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(k)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j + 8]) : "v"(k));
                R8(X)
#undef X
            } else if (OP == OP_VADD_DEP) {
#define X(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(k));
                R8(X)
#undef X
            } else if (OP == OP_VBFE) {
#define X(j) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a[j]));
                R8(X)
#undef X
            } else if (OP == OP_VCNDMASK) {
#define X(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(k) : "vcc");
                R8(X)
#undef X
            } else if (OP == OP_VMAD24) {
#define X(j) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[j]) : "v"(k));
                R8(X)
#undef X
            } else if (OP == OP_VLSHLADD) {
#define X(j) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[j]) : "v"(k));
                R8(X)
#undef X
            } else if (OP == OP_VPERM) {
#define X(j) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(k));
                R8(X)
#undef X
            } else if (OP == OP_DPP) {
#define X(j) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(a[(j + 4) & 7]));
                R8(X)
#undef X
            } else if (OP == OP_DPP_DEP) {
#define X(j) asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
                R8(X)
#undef X
            } else if (OP == OP_BPERMUTE) {
                u32 r[8];
#define X(j) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r[j]) : "v"(a[j] << 2), "v"(k));
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define X(j) acc ^= r[j];
                R8(X)
#undef X
            } else if (OP == OP_BPERMUTE_DEP) {
#define X(j) asm volatile("ds_bpermute_b32 %0, %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[0]) : "v"(k));
                R8(X)
#undef X
            } else if (OP == OP_LDS_U8 || OP == OP_LDS_B32) {
                u32 r[8];
#define X(j) { const u32 ad = (OP == OP_LDS_U8) ? ((a[j] + acc) & 4095u) : (((a[j] + acc) & 1023u) << 2); \
               if (OP == OP_LDS_U8) asm volatile("ds_read_u8 %0, %1" : "=v"(r[j]) : "v"(ad)); else asm volatile("ds_read_b32 %0, %1" : "=v"(r[j]) : "v"(ad)); }
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define X(j) acc += r[j];
                R8(X)
#undef X
            } else if (OP == OP_LDS_U8_DEP) {
#define X(j) asm volatile("ds_read_u8 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[0]));
                R8(X)
#undef X
            } else if (OP == OP_READLANE_DEP) {
#define X(j) sidx = __builtin_amdgcn_readlane(a[1], sidx) & 63u;
                R8(X)
#undef X
            } else if (OP == OP_READLANE) {
#define X(j) asm volatile("v_readlane_b32 %0, %1, " #j : "=s"(s[j]) : "v"(a[j]));
                R8(X)
#undef X
            } else if (OP == OP_READFIRST) {
#define X(j) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s[j]) : "v"(a[j]));
                R8(X)
#undef X
            } else if (OP == OP_VCMP) {
                u64 m[8];
#define X(j) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m[j]) : "v"(a[j]), "v"(k));
                R8(X)
#undef X
#define X(j) s[j] ^= (u32)m[j];
                if (u == UNROLL - 1) { R8(X) }
#undef X
            } else if (OP == OP_MBCNT) {
#define X(j) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %2, %0" : "=&v"(a[j]) : "s"(s[j]), "s"(s[(j + 1) & 7]));
                R8(X)
#undef X
            } else if (OP == OP_SADD) {
#define X(j) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[j]) : "s"(sidx) : "scc");
                R8(X)
#undef X
            } else if (OP == OP_SADD_DEP) {
#define X(j) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[0]) : "s"(sidx) : "scc");
                R8(X)
#undef X
            } else if (OP == OP_SBFE) {
#define X(j) asm volatile("s_bfe_u32 %0, %0, 0x1f0001" : "+s"(s[j]) :: "scc");
                R8(X)
#undef X
            } else if (OP == OP_MIX_SV) {
#define X(j) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(a[j]), "+s"(s[j]) : "v"(k), "s"(sidx) : "scc");
                R8(X)
#undef X
            } else if (OP == OP_SPLIT_SV || OP == OP_SPLIT_V_ONLY) {
                // (the role is wave-uniform and loop-invariant: `role` below selects one of two straight-line bodies through scalar branches)
                if (role) {
                    if (OP == OP_SPLIT_SV) {
#define X(j) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[j]) : "s"(sidx) : "scc");
                        R8(X)
#undef X
                    }
                } else {
#define X(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(k));
                    R8(X)
#undef X
                }
            } else if (OP == OP_WALK) {
                // the loop of lane_walk_pos (csrc/alz_decode_fast.h:508-517) itself: readlane the element's size, record its position, count, advance
#define X(j) { u32 sz; asm volatile("v_readlane_b32 %[n], %[nx], %[t]\n\tv_writelane_b32 %[spos], %[t], m0\n\ts_add_u32 m0, m0, 1\n\ts_add_u32 %[t], %[t], %[n]\n\ts_and_b32 %[t], %[t], 63" \
               : [n] "=&s"(sz), [t] "+s"(sidx), [spos] "+v"(a[2]) : [nx] "v"(a[1]) : "scc", "m0"); }
                R8(X)
#undef X
            } else if (OP == OP_VMOV) {
#define X(j) asm volatile("v_mov_b32 %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VAND) {
#define X(j) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VLSHL) {
#define X(j) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VLSHR_V) {
#define X(j) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VMINU) {
#define X(j) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VMUL24) {
#define X(j) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VCND_VCC) {
#define X(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VCND_SGPR) {
#define X(j) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VADD_E64) {
#define X(j) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VADD_SGPR) {
#define X(j) asm volatile("v_add_u32 %0, %3, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VADD_LIT) {
#define X(j) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VANDOR) {
#define X(j) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VADD3) {
#define X(j) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VALIGN) {
#define X(j) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VBCNT) {
#define X(j) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VMULLO) {
#define X(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VCMP_VCC) {
#define X(j) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j) : "vcc");
                R8(X)
#undef X
            } else if (OP == OP_VBFE_V) {
#define X(j) asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_SDWA) {
#define X(j) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_DPP_ADD) {
#define X(j) asm volatile("v_add_u32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_DSWRITE8) {
#define X(j) asm volatile("ds_write_b8 %5, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VCND_E64_VCC) {
#define X(j) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VSUB) {
#define X(j) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VOR) {
#define X(j) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VXOR) {
#define X(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VMAXU) {
#define X(j) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VLSHL_V) {
#define X(j) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VLSHR_C) {
#define X(j) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VAND_C) {
#define X(j) asm volatile("v_and_b32 %0, 63, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VADD_C) {
#define X(j) asm volatile("v_add_u32 %0, 1, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VLSHLOR) {
#define X(j) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VLSHL16) {
#define X(j) asm volatile("v_lshlrev_b16 %0, 1, %0" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VADDC) {
#define X(j) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VADDCO) {
#define X(j) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_VADD_2SRC) {
#define X(j) asm volatile("v_add_u32 %0, %4, %1" : "+v"(a[j]) : "v"(k), "s"(mask64), "s"(sidx), "v"(a[(j + 3) & 7]), "v"(ldsaddr + 64u * j));
                R8(X)
#undef X
            } else if (OP == OP_SWIZZLE) {
                u32 r[8];
#define X(j) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(BITMASK_PERM,\"01pip\")" : "=v"(r[j]) : "v"(a[j]));
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define X(j) acc ^= r[j];
                R8(X)
#undef X
            } else if (OP == OP_DSREAD128) {
                uint4 r[8];
#define X(j) asm volatile("ds_read_b128 %0, %1 offset:" #j "*1024" : "=v"(r[j]) : "v"((u32)lane * 16u));
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define X(j) acc ^= r[j].x ^ r[j].w;
                R8(X)
#undef X
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; j++) acc ^= a[j];
#pragma unroll
    for (int j = 0; j < 8; j++) acc ^= s[j];
    acc ^= sidx;
    if (acc == 0x12345u) sink[0] = acc;
    if (lane == 0) {
        u32 hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        u64* o = cyc + (size_t)(blockIdx.x * 4 + wave) * 4;
        o[0] = t1 - t0; o[1] = t0; o[2] = t1; o[3] = ((u64)xcc << 32) | hw;
    }
}

struct Row { double cpi, wall_ipc_cu, overlap; int max_per_simd, min_per_simd; };
template <int OP>
static Row run(int waves_per_simd, int iters, u64* d_cyc, u32* d_sink, std::vector<u64>& h, int ncu) {
    // one workgroup = 4 wavefronts = one per SIMD; `waves_per_simd` workgroups per CU, pinned by the dynamic LDS they ask for
    const int grid = ncu * waves_per_simd;
    const size_t lds_dyn = (size_t)(160 * 1024 / waves_per_simd - 4096 - 256) & ~(size_t)255;
    CK(hipFuncSetAttribute((const void*)k_issue<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dyn));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(256), lds_dyn, 0, d_cyc, d_sink, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(256), lds_dyn, 0, d_cyc, d_sink, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    const size_t nw = (size_t)grid * 4;
    CK(hipMemcpy(h.data(), d_cyc, sizeof(u64) * nw * 4, hipMemcpyDeviceToHost));
    std::vector<u64> v(nw);
    u64 tmin = ~0ull, tmax = 0;
    std::map<u64, int> per_simd;
    for (size_t i = 0; i < nw; i++) {
        v[i] = h[4 * i]; tmin = std::min(tmin, h[4 * i + 1]); tmax = std::max(tmax, h[4 * i + 2]);
        const u32 hw = (u32)h[4 * i + 3], xcc = (u32)(h[4 * i + 3] >> 32) & 15u;
        // HW_ID: simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
        per_simd[((u64)xcc << 20) | (hw & 0xFF30u)]++;
    }
    std::sort(v.begin(), v.end());
    const double med = (double)v[v.size() / 2];
    const double n = (double)iters * UNROLL * 8 * (OP == OP_VADD16 ? 2 : 1) * (OP == OP_SPLIT_V_ONLY ? 0.5 : 1.0);     // asm statements per wave (a pair for mbcnt / mix, an element for the walk)
    Row r;
    r.cpi = med / n;
    r.wall_ipc_cu = n * (double)nw / ((double)ms * 1e-3 * 2.4e9) / ncu;         // statements per 2.4 GHz cycle per CU, from the HIP-event time of the launch
    r.overlap = (double)ms * 1e-3 * 2.4e9 / med;                                  // launch time / one wave's own time: ~1 = all waves ran side by side; k = they ran in k shifts
    (void)tmin; (void)tmax;
    r.max_per_simd = 0; r.min_per_simd = 1 << 30;
    for (auto& kv : per_simd) { r.max_per_simd = std::max(r.max_per_simd, kv.second); r.min_per_simd = std::min(r.min_per_simd, kv.second); }
    if ((int)per_simd.size() != ncu * 4) r.min_per_simd = 0;                      // some SIMD got no wave at all
    return r;
}

static bool g_only_set = false; static bool g_only[OP_COUNT]; static int g_w8 = 0;
int main(int argc, char** argv) {
    // usage: ubench_issue [--w8] [op index ...]   (--w8: only 8 waves per SIMD; indices: only those rows -- for rocprofv3 --pmc calibration passes)
    for (int i = 1; i < argc; i++) { if (!strcmp(argv[i], "--w8")) { g_w8 = 1; continue; } const int o = atoi(argv[i]); if (o >= 0 && o < OP_COUNT) { g_only[o] = true; g_only_set = true; } }
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, ncu, p.clockRate);
    u64* d_cyc; u32* d_sink;
    CK(hipMalloc(&d_cyc, sizeof(u64) * ncu * 8 * 4 * 4)); CK(hipMalloc(&d_sink, 64));
    std::vector<u64> h((size_t)ncu * 8 * 4 * 4);
    const int ws[5] = {1, 2, 4, 6, 8};
    const int iters = 4000;
    printf("per cell: cycles per wave64 instruction as ONE wave sees them (s_memtime, median over waves) | wave64 instructions per cycle per CU from the\n"
           "HIP-event time of the whole launch at 2.4 GHz | shifts = launch time / one wave's own time (1 = the waves ran side by side; k = the SIMD served the oldest waves first and the rest later) |\n"
           "waves per SIMD as HW_ID / XCC_ID report them (min-max over the SIMDs)\n");
#define ROW(OP) if (!g_only_set || g_only[OP]) { printf("[%d] %s\n", (int)OP, NAMES[OP]); for (int w : ws) { if (g_w8 && w != 8) continue; Row r = run<OP>(w, iters, d_cyc, d_sink, h, ncu); \
                  printf("    %d waves/SIMD asked: %7.2f cyc/instr/wave  -> %5.3f instr/cyc/SIMD (s_memtime) | %5.3f instr/cyc/CU (wall) | shifts %4.2f | placed %d-%d per SIMD\n", \
                         w, r.cpi, w / r.cpi, r.wall_ipc_cu, r.overlap, r.min_per_simd, r.max_per_simd); } fflush(stdout); }
    ROW(OP_VADD) ROW(OP_VADD16) ROW(OP_VADD_DEP) ROW(OP_VBFE) ROW(OP_VMAD24) ROW(OP_VLSHLADD) ROW(OP_VPERM) ROW(OP_DPP) ROW(OP_DPP_DEP)
    ROW(OP_BPERMUTE) ROW(OP_BPERMUTE_DEP) ROW(OP_LDS_U8) ROW(OP_LDS_B32) ROW(OP_LDS_U8_DEP) ROW(OP_READLANE) ROW(OP_READFIRST) ROW(OP_READLANE_DEP) ROW(OP_VCMP) ROW(OP_MBCNT)
    ROW(OP_SADD) ROW(OP_SADD_DEP) ROW(OP_SBFE) ROW(OP_MIX_SV) ROW(OP_SPLIT_SV) ROW(OP_SPLIT_V_ONLY) ROW(OP_WALK)
    ROW(OP_VMOV) ROW(OP_VAND) ROW(OP_VLSHL) ROW(OP_VLSHR_V) ROW(OP_VMINU) ROW(OP_VMUL24) ROW(OP_VCND_VCC) ROW(OP_VCND_SGPR) ROW(OP_VADD_E64) ROW(OP_VADD_SGPR) ROW(OP_VADD_LIT) ROW(OP_VANDOR) ROW(OP_VADD3) ROW(OP_VALIGN) ROW(OP_VBCNT) ROW(OP_VMULLO) ROW(OP_VCMP_VCC) ROW(OP_VBFE_V) ROW(OP_SDWA) ROW(OP_DPP_ADD) ROW(OP_DSWRITE8) ROW(OP_DSREAD128)
    ROW(OP_VCND_E64_VCC) ROW(OP_VSUB) ROW(OP_VOR) ROW(OP_VXOR) ROW(OP_VMAXU) ROW(OP_VLSHL_V) ROW(OP_VLSHR_C) ROW(OP_VAND_C) ROW(OP_VADD_C) ROW(OP_VLSHLOR) ROW(OP_VLSHL16) ROW(OP_VADDC) ROW(OP_VADDCO) ROW(OP_SWIZZLE) ROW(OP_VADD_2SRC)
    return 0;
}
