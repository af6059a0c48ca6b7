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
#include <vector>
typedef unsigned int u32; typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { OP_VADD = 0, OP_VADD_DEP, OP_VBFE, OP_VCNDMASK, OP_VMAD24, OP_VLSHLADD, OP_VPERM, OP_DPP, OP_DPP_DEP, OP_BPERMUTE, OP_BPERMUTE_DEP, OP_LDS_U8, OP_LDS_B32,
       OP_LDS_U8_DEP, OP_READLANE_DEP, OP_READLANE, OP_VCMP, OP_MBCNT, OP_SADD, OP_SADD_DEP, OP_SBFE, OP_MIX_SV, OP_SPLIT_SV, OP_WALK, OP_VADD16, OP_READFIRST, OP_COUNT };
static const char* NAMES[OP_COUNT] = {
    "v_add_u32, 8 independent chains", "v_add_u32, ONE dependent chain", "v_bfe_u32 (VOP3), 8 chains", "v_cndmask_b32 (vcc), 8 chains", "v_mad_u32_u24, 8 chains",
    "v_lshl_add_u32, 8 chains", "v_perm_b32, 8 chains", "v_mov_b32 DPP row_shr:1, 8 chains", "v_add_u32 DPP row_shr:1, ONE dependent chain",
    "ds_bpermute_b32, 8 independent (one wait per 8)", "ds_bpermute_b32, dependent (wait each)", "ds_read_u8, 8 independent random (one wait per 8)",
    "ds_read_b32, 8 independent (one wait per 8)", "ds_read_u8 dependent (pointer chase)", "v_readlane_b32 -> s index -> v_readlane (dependent walk)",
    "v_readlane_b32, constant lanes, 8 independent", "v_cmp_lt_u32 -> SGPR pair, 8 independent", "v_mbcnt_lo + v_mbcnt_hi pairs", "s_add_u32, 8 independent chains",
    "s_add_u32, ONE dependent chain", "s_bfe_u32, 8 chains", "v_add_u32 + s_add_u32 interleaved 1:1 in one wave (per PAIR)",
    "VALU-only waves beside SALU-only waves (even / odd wave of a workgroup; per instruction of either)",
    "lane walk: v_readlane + v_writelane m0 + s_add m0 + s_add + s_and, dependent (per ELEMENT: 5 instr)", "v_add_u32, 16 independent chains", "v_readfirstlane_b32, 8 independent" };

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
    u64 t0, t1;
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
            } else if (OP == OP_SPLIT_SV) {
                if (wave & 1) {
#define X(j) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[j]) : "s"(sidx) : "scc");
                    R8(X)
#undef X
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
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int OP>
static void run(int waves_per_simd, int iters, u64* d_cyc, u32* d_sink, std::vector<u64>& h, int ncu, double* out_cpi) {
    // one workgroup = 4 wavefronts = one per SIMD; `waves_per_simd` workgroups per CU, pinned by the dynamic LDS they ask for
    const int grid = ncu * waves_per_simd;
    const size_t lds_dyn = (size_t)(160 * 1024 / waves_per_simd - 4096 - 256) & ~(size_t)255;
    CK(hipFuncSetAttribute((const void*)k_issue<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dyn));
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(256), lds_dyn, 0, d_cyc, d_sink, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_cyc, sizeof(u64) * grid * 4, hipMemcpyDeviceToHost));
    std::vector<u64> v(h.begin(), h.begin() + grid * 4);
    if (OP == OP_SPLIT_SV) {}   // (both kinds of waves in one median: they run side by side for the same number of instructions)
    std::sort(v.begin(), v.end());
    const double med = (double)v[v.size() / 2];
    const double per = (OP == OP_MBCNT || OP == OP_MIX_SV) ? 1.0 : 1.0;     // rows are per asm statement (a pair for mbcnt / mix, an element for the walk)
    const double n = (double)iters * UNROLL * 8 * (OP == OP_VADD16 ? 2 : 1) * per;
    *out_cpi = med / n;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, ncu, p.clockRate);
    u64* d_cyc; u32* d_sink;
    CK(hipMalloc(&d_cyc, sizeof(u64) * ncu * 8 * 4)); CK(hipMalloc(&d_sink, 64));
    std::vector<u64> h(ncu * 8 * 4);
    const int ws[5] = {1, 2, 4, 6, 8};
    const int iters = 2000;
    printf("cycles per wave64 instruction as ONE wave sees them (median over waves) | instructions per cycle per SIMD = waves / that | per CU = x4\n");
    printf("%-92s", "instruction class \\ waves per SIMD");
    for (int w : ws) printf("        %d          ", w);
    printf("\n");
#define ROW(OP) { printf("%-92s", NAMES[OP]); for (int w : ws) { double cpi; run<OP>(w, iters, d_cyc, d_sink, h, ncu, &cpi); \
                  printf(" %6.2f (%4.2f/SIMD)", cpi, w / cpi); } printf("\n"); fflush(stdout); }
    ROW(OP_VADD) ROW(OP_VADD16) ROW(OP_VADD_DEP) ROW(OP_VBFE) ROW(OP_VCNDMASK) ROW(OP_VMAD24) ROW(OP_VLSHLADD) ROW(OP_VPERM) ROW(OP_DPP) ROW(OP_DPP_DEP)
    ROW(OP_BPERMUTE) ROW(OP_BPERMUTE_DEP) ROW(OP_LDS_U8) ROW(OP_LDS_B32) ROW(OP_LDS_U8_DEP) ROW(OP_READLANE) ROW(OP_READFIRST) ROW(OP_READLANE_DEP) ROW(OP_VCMP) ROW(OP_MBCNT)
    ROW(OP_SADD) ROW(OP_SADD_DEP) ROW(OP_SBFE) ROW(OP_MIX_SV) ROW(OP_SPLIT_SV) ROW(OP_WALK)
    return 0;
}
