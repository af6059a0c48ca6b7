// ubench_graph.hip -- what a sequence of ~35 small dependent kernels costs per call on this runtime: launched one by one on a stream, replayed
// as a captured hipGraph, or as ONE persistent kernel with a device-scope barrier between its phases (the three shapes VERDICT r04 item 8 names for
// the whole-GPU single-stream paths).  Each phase touches 1 MiB (what a round of pointer jumping over a 1 000 KiB stream does).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_graph tools/ubench_graph.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void phase(u32* __restrict__ a, const u32* __restrict__ b, u32 n, u32 k) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) a[i] = b[(i * 2654435761u + k) % n] + k;
}
// persistent form: `nph` phases in one launch, grid <= resident workgroups, sense-reversing barrier on two device words, bounded spin
__global__ __launch_bounds__(256) void phases_persistent(u32* __restrict__ a, u32* __restrict__ b, u32 n, u32 nph, u32* __restrict__ bar, u32* __restrict__ fail) {
    const u32 nblk = gridDim.x;
    for (u32 k = 0; k < nph; k++) {
        u32* dst = (k & 1u) ? b : a; const u32* src = (k & 1u) ? a : b;
        for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += nblk * 256u) dst[i] = src[(i * 2654435761u + k) % n] + k;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const u32 target = (k + 1u) * nblk;
            atomicAdd(bar, 1u);
            u32 spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) { if (++spins > 20000000u) { *fail = 1u; break; } __builtin_amdgcn_s_sleep(1); }
        }
        __syncthreads();
    }
}

int main() {
    const u32 n = 1u << 18, nk = 35;
    u32 *a, *b, *bar, *fail;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&bar, 64)); CK(hipMalloc(&fail, 64));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 1, n * 4)); CK(hipMemset(fail, 0, 64));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto seq = [&]() { for (u32 k = 0; k < nk; k++) hipLaunchKernelGGL(phase, dim3(n / 256), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, n, k); };
    auto timeit = [&](const char* name, auto fn, int reps) {
        for (int i = 0; i < 5; i++) { fn(); CK(hipStreamSynchronize(s)); }
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) { fn(); CK(hipStreamSynchronize(s)); }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        // the same back to back without a sync in between: what the device needs when the host is not the limit
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) fn();
        const double us_enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        CK(hipStreamSynchronize(s));
        const double us_b2b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("%-58s %8.1f us per call + sync | %8.1f us to enqueue | %8.1f us back to back\n", name, us, us_enq, us_b2b);
    };
    timeit("35 kernels launched one by one", seq, 200);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); seq(); CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    timeit("the same 35 kernels as ONE captured hipGraph", [&]() { CK(hipGraphLaunch(ge, s)); }, 200);
    int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, phases_persistent, 256, 0));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const u32 grid = (u32)p.multiProcessorCount * (u32)(nb > 4 ? 4 : nb);        // (a margin below the API's figure: MI355X_MICROARCH.md, grid barriers)
    timeit("ONE persistent kernel, 35 phases, device-scope barrier", [&]() { CK(hipMemsetAsync(bar, 0, 4, s)); hipLaunchKernelGGL(phases_persistent, dim3(grid), dim3(256), 0, s, a, b, n, nk, bar, fail); }, 200);
    u32 f = 0; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    printf("persistent grid %u workgroups (API: %d per CU), spin bound hit: %u\n", grid, nb, f);
    return 0;
}
