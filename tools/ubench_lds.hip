// ubench_lds.hip -- gfx950 LDS behaviour the chunked byte phase relies on: unaligned ds_read_b128 / ds_write_b128 /
// b64 / b32 / b16 (correct at every byte alignment?) and what they cost next to aligned and byte-wise accesses.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_lds tools/ubench_lds.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32; typedef unsigned char u8; typedef unsigned short u16; typedef unsigned long long u64;
struct __attribute__((packed)) P2 { u16 v; };
struct __attribute__((packed)) P4 { u32 v; };
struct __attribute__((packed)) P8 { uint2 v; };
struct __attribute__((packed)) P16 { uint4 v; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u8 pat(u32 i) { return (u8)(i * 7u + 3u + (i >> 8)); }

__global__ void k_correct(u32* errs) {
    __shared__ __attribute__((aligned(16))) u8 lds[8192 + 64];
    const int lane = threadIdx.x;
    u32 bad = 0;
    for (u32 al = 0; al < 16; al++) {
        for (u32 i = lane; i < 8192 + 64; i += 64) lds[i] = pat(i);
        __syncthreads();
        const u32 a = 48u * lane + al + 5u * (lane & 3);           // every alignment, lanes spread
        const uint4 v = reinterpret_cast<const P16*>(lds + a)->v;
        const uint2 w = reinterpret_cast<const P8*>(lds + a + 1)->v;
        const u32 x = reinterpret_cast<const P4*>(lds + a + 2)->v;
        const u32 h = reinterpret_cast<const P2*>(lds + a + 3)->v;
        u32 vv[4] = {v.x, v.y, v.z, v.w};
        for (int j = 0; j < 16; j++) if (((vv[j >> 2] >> (8 * (j & 3))) & 0xFF) != pat(a + j)) bad |= 1;
        u32 ww[2] = {w.x, w.y};
        for (int j = 0; j < 8; j++) if (((ww[j >> 2] >> (8 * (j & 3))) & 0xFF) != pat(a + 1 + j)) bad |= 2;
        for (int j = 0; j < 4; j++) if (((x >> (8 * j)) & 0xFF) != pat(a + 2 + j)) bad |= 4;
        for (int j = 0; j < 2; j++) if (((h >> (8 * j)) & 0xFF) != pat(a + 3 + j)) bad |= 8;
        __syncthreads();
        // unaligned stores: 16 + 8 + 4 + 2 bytes back to back at a + 4096
        const u32 b = 4096u + a;
        reinterpret_cast<P16*>(lds + b)->v = make_uint4(0x03020100u + lane, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu);
        reinterpret_cast<P8*>(lds + b + 16)->v = make_uint2(0x13121110u, 0x17161514u);
        reinterpret_cast<P4*>(lds + b + 24)->v = 0x1b1a1918u;
        reinterpret_cast<P2*>(lds + b + 28)->v = (u16)0x1d1cu;
        __syncthreads();
        for (int j = 0; j < 30; j++) { const u32 e = (j == 0) ? ((u32)lane & 0xFF) : (u32)j; if (lds[b + j] != (u8)e && !(j == 0 && lds[b] == (u8)lane)) bad |= 16; }
        if (lds[b + 30] != pat(b + 30) || (a > 0 && lane == 0 && lds[b - 1] != pat(b - 1))) bad |= 32;   // neighbours untouched
        __syncthreads();
    }
    atomicOr(errs, bad);
}

// MODE 0: aligned b128 reads, lane-contiguous; 1: unaligned b128, lane-contiguous + 3; 2: unaligned b128 at random addresses;
// 3: aligned b128 random; 4: 16 x u8 reads at random addresses (the 1-byte-per-lane shape x16); 5: unaligned b128 write random;
// 6: aligned b128 write random; 7: unaligned b64 write random; 8: unaligned b32 read random; 9: u8 write random
template <int MODE>
__global__ __launch_bounds__(64) void k_rate(u32* out, int iters) {
    __shared__ __attribute__((aligned(16))) u8 lds[4096 + 64];
    const int lane = threadIdx.x;
    for (u32 i = lane; i < 4096 + 64; i += 64) lds[i] = pat(i);
    __syncthreads();
    u32 acc = 0, r = lane * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            r = r * 1664525u + 1013904223u;
            u32 a;
            if (MODE == 0) a = ((r >> 20) & 0xC00u) + 16u * lane;
            else if (MODE == 1) a = ((r >> 20) & 0xC00u) + 16u * lane + 3u;
            else if (MODE == 3 || MODE == 6) a = (r >> 20) & 0xFF0u;
            else a = (r >> 20) & 0xFFFu;
            if (MODE <= 3) { const uint4 v = reinterpret_cast<const P16*>(lds + a)->v; acc ^= v.x + v.y + v.z + v.w; }
            else if (MODE == 4) { u32 s = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) s += lds[(a + 97u * j) & 0xFFFu]; acc ^= s; }
            else if (MODE == 5 || MODE == 6) reinterpret_cast<P16*>(lds + a)->v = make_uint4(r, acc, r, acc);
            else if (MODE == 7) reinterpret_cast<P8*>(lds + a)->v = make_uint2(r, acc);
            else if (MODE == 8) acc ^= reinterpret_cast<const P4*>(lds + a)->v;
            else if (MODE == 9) lds[a] = (u8)r;
            else if (MODE == 10) { const uint4 v = reinterpret_cast<const P16*>(lds + (a & ~3u))->v; acc ^= v.x + v.y + v.z + v.w; }      // b128 read, dword aligned
            else if (MODE == 11) { const uint4 v = reinterpret_cast<const P16*>(lds + (a & ~7u))->v; acc ^= v.x + v.y + v.z + v.w; }      // b128 read, 8-byte aligned
            else if (MODE == 12) { if (lane < 8) { const uint4 v = reinterpret_cast<const P16*>(lds + a)->v; acc ^= v.x + v.y + v.z + v.w; } }   // unaligned, 8 lanes
            else if (MODE == 13) { if (lane < 1) { const uint4 v = reinterpret_cast<const P16*>(lds + a)->v; acc ^= v.x + v.y + v.z + v.w; } }   // unaligned, 1 lane
            else if (MODE == 14) { const u32 aa = (lane == 5) ? a : (a & ~15u); const uint4 v = reinterpret_cast<const P16*>(lds + aa)->v; acc ^= v.x + v.y + v.z + v.w; }   // one lane unaligned
            else if (MODE == 15) { const u32 aa = a & ~3u; asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"(aa), "v"(0xFF00u), "v"(r) : "memory"); }
            else if (MODE == 16) { *reinterpret_cast<u32*>(lds + (a & ~3u)) = r; }
            else if (MODE == 17) { *reinterpret_cast<u16*>(lds + (a & ~1u)) = (u16)r; }
            else if (MODE == 18) { const uint2 v = reinterpret_cast<const P8*>(lds + (a & ~3u))->v; acc ^= v.x + v.y; }                  // b64 read dword aligned
            else if (MODE == 19) { acc ^= *reinterpret_cast<const u32*>(lds + (a & ~3u)); }                                            // b32 read aligned random
            else if (MODE == 20) { reinterpret_cast<P16*>(lds + (a & ~3u))->v = make_uint4(r, acc, r, acc); }                           // b128 write dword aligned
            else if (MODE == 21) { reinterpret_cast<P8*>(lds + (a & ~3u))->v = make_uint2(r, acc); }                                    // b64 write dword aligned
            else if (MODE == 22) { acc ^= lds[a]; }                                                                                    // u8 read random
        }
    }
    if (MODE >= 5 && MODE != 8) acc ^= lds[lane * 4] + lds[lane * 4 + 1];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, u32* d_out, double bytes_per_op) {
    const int blocks = 256 * 16, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * iters * 8;                 // wave-instructions (MODE 4: groups of 16)
    // 256 CUs, assume 2.1 GHz under load
    printf("%-34s %8.3f ms  %7.2f ns/wave-op/CU  ~%6.1f cyc@2.1GHz  %8.1f GB/s chip\n", name, ms, ms * 1e6 / (ops / 256.0), ms * 1e6 / (ops / 256.0) * 2.1,
           ops * 64 * bytes_per_op / (ms * 1e-3) / 1e9);
}

int main() {
    u32* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipLaunchKernelGGL(k_correct, dim3(1), dim3(64), 0, 0, d);
    u32 errs = 0; CK(hipMemcpy(&errs, d, 4, hipMemcpyDeviceToHost));
    printf("unaligned LDS correctness mask (0 = all good; 1 b128 read, 2 b64 read, 4 b32 read, 8 b16 read, 16 stores, 32 neighbours): %u\n", errs);
    run<0>("b128 read aligned contiguous", d, 16);
    run<1>("b128 read unaligned contiguous+3", d, 16);
    run<2>("b128 read unaligned random", d, 16);
    run<3>("b128 read aligned random", d, 16);
    run<4>("16 x u8 read random", d, 16);
    run<5>("b128 write unaligned random", d, 16);
    run<6>("b128 write aligned random", d, 16);
    run<7>("b64 write unaligned random", d, 8);
    run<8>("b32 read unaligned random", d, 4);
    run<9>("u8 write random", d, 1);
    run<10>("b128 read dword-aligned random", d, 16);
    run<11>("b128 read 8B-aligned random", d, 16);
    run<12>("b128 read unaligned, 8 lanes", d, 16);
    run<13>("b128 read unaligned, 1 lane", d, 16);
    run<14>("b128 read, ONE lane unaligned", d, 16);
    run<15>("ds_mskor_b32 aligned random", d, 4);
    run<16>("b32 write aligned random", d, 4);
    run<17>("b16 write aligned random", d, 2);
    run<18>("b64 read dword-aligned random", d, 8);
    run<19>("b32 read aligned random", d, 4);
    run<20>("b128 write dword-aligned random", d, 16);
    run<21>("b64 write dword-aligned random", d, 8);
    run<22>("u8 read random", d, 1);
    return 0;
}
