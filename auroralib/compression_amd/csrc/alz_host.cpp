// alz_host.cpp -- context / plan management and the batch entry points of include/auroralz.h.
// Compiled by hipcc as host code; everything that computes runs in alz_kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "alz_internal.h"

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                                  \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(ALZ_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

struct alz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // grow-only staging for the host-buffer entry points
    void* d_src = nullptr; size_t d_src_cap = 0;
    void* d_dst = nullptr; size_t d_dst_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // fork/join resources for per-format kernels of a mixed batch (they are independent: run them concurrently)
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[4] = {nullptr, nullptr, nullptr, nullptr};
};

struct alz_plan {
    uint32_t n = 0;
    alz_lz_properties lz{};
    alz_stream* d_streams = nullptr;
    alz_result* d_results = nullptr;
    uint32_t* d_index = nullptr;            // concatenated per-format index lists
    uint32_t fmt_off[ALZ_FMT_COUNT] = {0};
    uint32_t fmt_cnt[ALZ_FMT_COUNT] = {0};
};

static alz_lz_properties effective_lz(const alz_lz_properties* p) {
    alz_lz_properties lz;
    if (!p || p->window_bits == 0) {           // LZSS.DefaultProperties = LzProperties((byte)12, 4, 2)  LZSS.cs:33
        memset(&lz, 0, sizeof(lz)); lz.window_bits = 12; lz.length_bits = 4; lz.min_length = 3; lz.max_distance = 4096; lz.windows_start = 0xFEE;
    } else { lz = *p; if (lz.max_distance == 0) lz.max_distance = 1u << lz.window_bits; }
    return lz;
}

extern "C" {

int alz_abi_version(void) { return ALZ_ABI_VERSION; }
/* not in the public header: test hook that selects the exact serial kernels for every format (still the GPU path) */
void alz_debug_force_serial(int on) { alz_set_force_serial(on); }
/* not in the public header: resident waves per CU of the production kernel of `format` (tuning aid) */
int alz_debug_occupancy(int format) { return alz_kernel_occupancy(format); }
const char* alz_last_error(void) { return g_err; }

int alz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int alz_create(int device, alz_ctx** out) {
    if (!out) return fail(ALZ_E_INVALID, "alz_create: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(ALZ_E_NO_DEVICE, "no HIP device: the auroralz ABI has no CPU fallback");
    if (device < 0 || device >= n) return fail(ALZ_E_INVALID, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    alz_ctx* c = new (std::nothrow) alz_ctx();
    if (!c) return fail(ALZ_E_NOMEM, "out of host memory");
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
    for (int i = 0; i < 4 && e == hipSuccess; i++) { e = hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking); if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming); }
    if (e != hipSuccess) { delete c; return fail(ALZ_E_HIP, "context creation failed: %s", hipGetErrorString(e)); }
    *out = c;
    return ALZ_OK;
}

void alz_destroy(alz_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_src) (void)hipFree(c->d_src);
    if (c->d_dst) (void)hipFree(c->d_dst);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->fork) (void)hipEventDestroy(c->fork);
    for (int i = 0; i < 4; i++) { if (c->join[i]) (void)hipEventDestroy(c->join[i]); if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int alz_device_info(alz_ctx* c, char* name, size_t name_cap, int* cu_count, uint64_t* hbm_bytes) {
    if (!c) return fail(ALZ_E_INVALID, "ctx is NULL");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, c->device));
    if (name && name_cap) { snprintf(name, name_cap, "%s (%s)", p.name, p.gcnArchName); }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    return ALZ_OK;
}

// ---------------------------------------------------------------- device memory helpers
int alz_device_malloc(alz_ctx* c, size_t bytes, void** d_ptr) {
    if (!c || !d_ptr) return fail(ALZ_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 16));
    return ALZ_OK;
}
int alz_device_free(alz_ctx* c, void* d_ptr) {
    if (!c) return fail(ALZ_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipFree(d_ptr));
    return ALZ_OK;
}
int alz_memcpy_h2d(alz_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    if (!c) return fail(ALZ_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ALZ_OK;
}
int alz_memcpy_d2h(alz_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    if (!c) return fail(ALZ_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ALZ_OK;
}
int alz_memset_d(alz_ctx* c, void* d_dst, int value, size_t bytes) {
    if (!c) return fail(ALZ_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(d_dst, value, bytes, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ALZ_OK;
}
int alz_synchronize(alz_ctx* c) {
    if (!c) return fail(ALZ_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ALZ_OK;
}

// ---------------------------------------------------------------- plans
void alz_plan_destroy(alz_ctx* c, alz_plan* p) {
    if (!p) return;
    if (c) (void)hipSetDevice(c->device);
    if (p->d_streams) (void)hipFree(p->d_streams);
    if (p->d_results) (void)hipFree(p->d_results);
    if (p->d_index) (void)hipFree(p->d_index);
    delete p;
}

int alz_plan_create(alz_ctx* c, const alz_lz_properties* props, uint32_t n, const alz_stream* streams, alz_plan** out) {
    if (!c || !out || (n && !streams)) return fail(ALZ_E_INVALID, "alz_plan_create: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    alz_lz_properties lz = effective_lz(props);
    std::vector<uint32_t> cnt(ALZ_FMT_COUNT, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (streams[i].format >= ALZ_FMT_COUNT) return fail(ALZ_E_INVALID, "stream %u: unknown format %u", i, streams[i].format);
        if (streams[i].format == ALZ_FMT_LZ4_BLOCK && streams[i].aux0) {   // history in front of dst_off
            if (streams[i].aux0 > streams[i].dst_off) return fail(ALZ_E_INVALID, "stream %u: history %u exceeds dst_off", i, streams[i].aux0);
            if ((uint64_t)streams[i].aux0 + streams[i].dst_cap > 0xFFFFFF00ull) return fail(ALZ_E_UNSUPPORTED, "stream %u: history + dst_cap exceed 4 GiB", i);
        }
        cnt[streams[i].format]++;
    }
    if (cnt[ALZ_FMT_LZSS] && (lz.window_bits < 8 || lz.window_bits > 16 || lz.length_bits < 1 || lz.length_bits > 8))
        return fail(ALZ_E_UNSUPPORTED, "LZSS geometry outside the GPU path (window_bits 8..16, length_bits 1..8)");
    alz_plan* p = new (std::nothrow) alz_plan();
    if (!p) return fail(ALZ_E_NOMEM, "out of host memory");
    p->n = n; p->lz = lz;
    std::vector<uint32_t> index(n ? n : 1);
    uint32_t off = 0;
    for (int f = 0; f < ALZ_FMT_COUNT; f++) { p->fmt_off[f] = off; p->fmt_cnt[f] = cnt[f]; off += cnt[f]; }
    std::vector<uint32_t> fill(ALZ_FMT_COUNT, 0);
    for (uint32_t i = 0; i < n; i++) { uint32_t f = streams[i].format; index[p->fmt_off[f] + fill[f]++] = i; }
    size_t nn = n ? n : 1;
    hipError_t e = hipMalloc((void**)&p->d_streams, nn * sizeof(alz_stream));
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_results, nn * sizeof(alz_result));
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_index, nn * sizeof(uint32_t));
    if (e == hipSuccess && n) e = hipMemcpyAsync(p->d_streams, streams, n * sizeof(alz_stream), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(p->d_index, index.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_results, 0xFF, nn * sizeof(alz_result), c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { alz_plan_destroy(c, p); return fail(ALZ_E_HIP, "plan upload failed: %s", hipGetErrorString(e)); }
    *out = p;
    return ALZ_OK;
}

int alz_plan_execute(alz_ctx* c, alz_plan* p, const void* d_src_base, void* d_dst_base, void* hip_stream) {
    if (!c || !p) return fail(ALZ_E_INVALID, "alz_plan_execute: bad argument");
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    int nfmt = 0;
    for (int f = 0; f < ALZ_FMT_COUNT; f++) nfmt += p->fmt_cnt[f] ? 1 : 0;
    if (nfmt <= 1) {
        for (int f = 0; f < ALZ_FMT_COUNT; f++) {
            if (!p->fmt_cnt[f]) continue;
            hipError_t e = alz_launch_decode(f, s, d_src_base, d_dst_base, p->d_streams, p->d_index + p->fmt_off[f], p->fmt_cnt[f], p->d_results, &p->lz);
            if (e != hipSuccess) return fail(ALZ_E_HIP, "kernel launch (format %d) failed: %s", f, hipGetErrorString(e));
        }
        return ALZ_OK;
    }
    // mixed batch: one kernel per format, forked onto side streams so that they share the GPU (each format alone may
    // have far fewer streams than the device has wave slots), joined back into the caller's stream
    HIP_TRY(hipEventRecord(c->fork, s));
    int k = 0; bool used[4] = {false, false, false, false};
    for (int f = 0; f < ALZ_FMT_COUNT; f++) {
        if (!p->fmt_cnt[f]) continue;
        hipStream_t a = c->aux[k & 3];
        if (!used[k & 3]) { HIP_TRY(hipStreamWaitEvent(a, c->fork, 0)); used[k & 3] = true; }
        hipError_t e = alz_launch_decode(f, a, d_src_base, d_dst_base, p->d_streams, p->d_index + p->fmt_off[f], p->fmt_cnt[f], p->d_results, &p->lz);
        if (e != hipSuccess) return fail(ALZ_E_HIP, "kernel launch (format %d) failed: %s", f, hipGetErrorString(e));
        k++;
    }
    for (int i = 0; i < 4; i++) if (used[i]) { HIP_TRY(hipEventRecord(c->join[i], c->aux[i])); HIP_TRY(hipStreamWaitEvent(s, c->join[i], 0)); }
    return ALZ_OK;
}

int alz_plan_execute_timed(alz_ctx* c, alz_plan* p, const void* d_src_base, void* d_dst_base, int iters, float* mean_ms) {
    if (!c || !p || iters < 1 || !mean_ms) return fail(ALZ_E_INVALID, "alz_plan_execute_timed: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < iters; i++) { int rc = alz_plan_execute(c, p, d_src_base, d_dst_base, nullptr); if (rc) return rc; }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *mean_ms = ms / (float)iters;
    return ALZ_OK;
}

int alz_plan_results(alz_ctx* c, alz_plan* p, alz_result* results) {
    if (!c || !p || (p->n && !results)) return fail(ALZ_E_INVALID, "alz_plan_results: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (p->n) HIP_TRY(hipMemcpy(results, p->d_results, p->n * sizeof(alz_result), hipMemcpyDeviceToHost));
    return ALZ_OK;
}

// ---------------------------------------------------------------- host-buffer entry points
static int grow(alz_ctx* c, void** buf, size_t* cap, size_t need) {
    if (*cap >= need) return ALZ_OK;
    if (*buf) { HIP_TRY(hipFree(*buf)); *buf = nullptr; *cap = 0; }
    size_t want = need + need / 4 + 4096;
    HIP_TRY(hipMalloc(buf, want));
    *cap = want;
    return ALZ_OK;
}

// Sparse outputs (compressed streams in worst-case-sized slots, batches with failed streams): one workgroup per stream packs
// the produced bytes into a dense device buffer, so that ONE copy crosses PCIe instead of one per stream.  `to` is congruent
// to `from` modulo 16, which keeps the 16-byte body of the copy aligned on both sides.
struct pack_item { uint64_t from, to; uint32_t len, pad; };
__global__ void __launch_bounds__(256) alz_pack_outputs_kernel(const uint8_t* __restrict__ base, uint8_t* __restrict__ pack, const pack_item* __restrict__ items) {
    const pack_item it = items[blockIdx.x];
    const uint8_t* s = base + it.from;
    uint8_t* d = pack + it.to;
    uint32_t head = (16u - (uint32_t)(it.from & 15)) & 15u;
    if (head > it.len) head = it.len;
    for (uint32_t k = threadIdx.x; k < head; k += blockDim.x) d[k] = s[k];
    const uint32_t body = (it.len - head) >> 4;
    const uint4* s4 = (const uint4*)(s + head);
    uint4* d4 = (uint4*)(d + head);
    for (uint32_t k = threadIdx.x; k < body; k += blockDim.x) d4[k] = s4[k];
    for (uint32_t k = head + (body << 4) + threadIdx.x; k < it.len; k += blockDim.x) d[k] = s[k];
}

// false = could not pack (no device memory for the dense copy): the caller falls back to one copy per stream
static bool download_packed(alz_ctx* c, uint32_t n, const alz_stream* streams, const alz_result* results, uint8_t* dst_base, bool only_ok) {
    std::vector<pack_item> items;
    uint64_t cur = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (!results[i].dst_len || (only_ok && results[i].status != ALZ_ST_OK)) continue;
        const uint64_t to = cur + (streams[i].dst_off & 15);
        items.push_back(pack_item{streams[i].dst_off, to, results[i].dst_len, i});
        cur = (to + results[i].dst_len + 15) & ~15ull;
    }
    void *d_items = nullptr, *d_pack = nullptr;
    if (hipMalloc(&d_items, items.size() * sizeof(pack_item)) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMalloc(&d_pack, cur) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_items); return false; }
    std::vector<uint8_t> bounce(cur);
    hipError_t e = hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(pack_item), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        alz_pack_outputs_kernel<<<dim3((uint32_t)items.size()), dim3(256), 0, c->stream>>>((const uint8_t*)c->d_dst, (uint8_t*)d_pack, (const pack_item*)d_items);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(bounce.data(), d_pack, cur, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_items); (void)hipFree(d_pack);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    for (const pack_item& it : items) memcpy(dst_base + it.from, bounce.data() + it.to, it.len);
    return true;
}

// Download per-stream outputs: one bulk copy when the produced ranges are reasonably dense, else packed on the device first.
static int download_outputs(alz_ctx* c, uint32_t n, const alz_stream* streams, const alz_result* results, uint8_t* dst_base, bool only_ok) {
    uint64_t lo = ~0ull, hi = 0, sum = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (!results[i].dst_len || (only_ok && results[i].status != ALZ_ST_OK)) continue;
        const uint64_t a = streams[i].dst_off, b = a + results[i].dst_len;
        if (a < lo) lo = a; if (b > hi) hi = b; sum += results[i].dst_len;
    }
    if (sum == 0) return ALZ_OK;
    if (hi - lo <= 2 * sum + (64ull << 20)) {
        // the caller's bytes between streams are preserved: stage through a bounce buffer and copy each stream out
        std::vector<uint8_t> bounce(hi - lo);
        HIP_TRY(hipMemcpyAsync(bounce.data(), (const uint8_t*)c->d_dst + lo, hi - lo, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (uint32_t i = 0; i < n; i++) {
            if (!results[i].dst_len || (only_ok && results[i].status != ALZ_ST_OK)) continue;
            memcpy(dst_base + streams[i].dst_off, bounce.data() + (streams[i].dst_off - lo), results[i].dst_len);
        }
        return ALZ_OK;
    }
    if (n >= 16 && download_packed(c, n, streams, results, dst_base, only_ok)) return ALZ_OK;
    for (uint32_t i = 0; i < n; i++) {
        if (!results[i].dst_len || (only_ok && results[i].status != ALZ_ST_OK)) continue;
        HIP_TRY(hipMemcpyAsync(dst_base + streams[i].dst_off, (const uint8_t*)c->d_dst + streams[i].dst_off, results[i].dst_len, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ALZ_OK;
}

int alz_decode_batch(alz_ctx* c, const alz_lz_properties* props, uint32_t n, const uint8_t* src_base, size_t src_bytes,
                     const alz_stream* streams, uint8_t* dst_base, size_t dst_bytes, alz_result* results) {
    if (!c || (n && (!streams || !results))) return fail(ALZ_E_INVALID, "alz_decode_batch: bad argument");
    for (uint32_t i = 0; i < n; i++) {
        if (streams[i].src_off + streams[i].src_len > src_bytes) return fail(ALZ_E_INVALID, "stream %u: source range exceeds src_bytes", i);
        if (streams[i].dst_off + streams[i].dst_cap > dst_bytes) return fail(ALZ_E_INVALID, "stream %u: destination range exceeds dst_bytes", i);
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = grow(c, &c->d_src, &c->d_src_cap, src_bytes + 64))) return rc;
    if ((rc = grow(c, &c->d_dst, &c->d_dst_cap, dst_bytes + 64))) return rc;
    if (src_bytes) HIP_TRY(hipMemcpyAsync(c->d_src, src_base, src_bytes, hipMemcpyHostToDevice, c->stream));
    for (uint32_t i = 0; i < n; i++)   // LZ4 blocks that continue a frame's window: their history has to be in HBM too
        if (streams[i].format == ALZ_FMT_LZ4_BLOCK && streams[i].aux0 && streams[i].aux0 <= streams[i].dst_off)
            HIP_TRY(hipMemcpyAsync((uint8_t*)c->d_dst + streams[i].dst_off - streams[i].aux0, dst_base + streams[i].dst_off - streams[i].aux0,
                                   streams[i].aux0, hipMemcpyHostToDevice, c->stream));
    alz_plan* p = nullptr;
    if ((rc = alz_plan_create(c, props, n, streams, &p))) return rc;
    rc = alz_plan_execute(c, p, c->d_src, c->d_dst, nullptr);
    if (!rc) rc = alz_plan_results(c, p, results);
    if (!rc) rc = download_outputs(c, n, streams, results, dst_base, false);   // copy back only what each stream produced
    alz_plan_destroy(c, p);
    return rc;
}

int alz_decode(alz_ctx* c, uint32_t format, const alz_lz_properties* props, const uint8_t* src, uint32_t src_len, uint32_t decom_len,
               uint32_t aux0, uint32_t aux1, uint8_t* dst, uint32_t dst_cap, alz_result* result) {
    if (!result) return fail(ALZ_E_INVALID, "alz_decode: result is NULL");
    alz_stream s; memset(&s, 0, sizeof(s));
    s.src_len = src_len; s.dst_cap = dst_cap; s.decom_len = decom_len; s.aux0 = aux0; s.aux1 = aux1; s.format = format;
    return alz_decode_batch(c, props, 1, src, src_len, &s, dst, dst_cap, result);
}

// Device buffers of one encode call, freed on every exit path
struct EncScratch {
    std::vector<void*> bufs;
    ~EncScratch() { for (void* p : bufs) if (p) (void)hipFree(p); }
    hipError_t alloc(void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) bufs.push_back(*p); return e; }
};

int alz_encode_batch(alz_ctx* c, const alz_lz_properties* props, const alz_settings* settings, uint32_t n, const uint8_t* src_base,
                     size_t src_bytes, const alz_stream* streams, uint8_t* dst_base, size_t dst_bytes, alz_result* results, alz_encode_aux* aux) {
    if (!c || (n && (!streams || !results || !dst_base))) return fail(ALZ_E_INVALID, "alz_encode_batch: bad argument");
    if (n == 0) return ALZ_OK;
    alz_settings st; if (settings) st = *settings; else { st.quality = 8; st.max_window_bits = 0; st.strategy = 0; st.min_distance = 0; }
    if (st.quality < 0 || st.quality > 15) return fail(ALZ_E_INVALID, "quality %d outside 0..15 (CompressionSettings.cs:38-50)", st.quality);
    if (st.max_window_bits != 0) return fail(ALZ_E_UNSUPPORTED, "CompressionSettings.MaxWindowBits != 0 is not supported by the GPU encoder");
    alz_lz_properties lz = effective_lz(props);
    std::vector<uint32_t> cnt(ALZ_FMT_COUNT, 0);
    std::vector<uint64_t> pos_off(n);
    uint64_t total = 0; uint32_t max_len = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (streams[i].format >= ALZ_FMT_COUNT) return fail(ALZ_E_INVALID, "stream %u: unknown format %u", i, streams[i].format);
        if (streams[i].src_off + streams[i].src_len > src_bytes) return fail(ALZ_E_INVALID, "stream %u: source range exceeds src_bytes", i);
        if (streams[i].dst_off + streams[i].dst_cap > dst_bytes) return fail(ALZ_E_INVALID, "stream %u: destination range exceeds dst_bytes", i);
        if (streams[i].src_len > 0x7FFFFF00u) return fail(ALZ_E_UNSUPPORTED, "stream %u: inputs above 2 GiB are not supported", i);
        cnt[streams[i].format]++;
        pos_off[i] = total; total += ((uint64_t)streams[i].src_len + 16 + 63) & ~63ull;   // 64-aligned: one start-mask word per 64 positions
        if (streams[i].src_len > max_len) max_len = streams[i].src_len;
    }
    if (cnt[ALZ_FMT_LZSS] && (lz.window_bits < 8 || lz.window_bits > 16 || lz.length_bits < 1 || lz.length_bits > 8 || lz.max_distance != (1u << lz.window_bits)))
        return fail(ALZ_E_UNSUPPORTED, "LZSS geometry outside the GPU path");
    std::vector<unsigned char> geom(ALZ_FMT_COUNT * alz_encode_geom_size());
    int hash_bits = 0; bool any_min = false;
    for (int f = 0; f < ALZ_FMT_COUNT; f++) {
        if (!cnt[f]) continue;
        void* g = geom.data() + f * alz_encode_geom_size();
        if (!alz_encode_geometry(f, &lz, &st, g, nullptr)) return fail(ALZ_E_UNSUPPORTED, "format %d: geometry not supported by the GPU encoder", f);
        hash_bits = alz_encode_geom_hash_bits(g);
        any_min = any_min || alz_encode_geom_min_table(g);
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = grow(c, &c->d_src, &c->d_src_cap, src_bytes + 64))) return rc;
    if ((rc = grow(c, &c->d_dst, &c->d_dst_cap, dst_bytes + 64))) return rc;
    // streams are processed in chunks so that the per-stream head tables (4 B << hash_bits each) stay bounded: 16 GiB of
    // tables per chunk (Q0: every stream at once, Q8: 8 192, Q15: 4 096) -- kernel A is bound by the latency of its
    // head-table round trips, so a chunk should at least fill the device's 8 192 wave slots
    uint32_t CH = 4096;
    { const uint64_t per = (uint64_t)sizeof(int) << hash_bits; const uint64_t fit = (16ull << 30) / per; if (fit > CH) CH = fit > 0x100000ull ? 0x100000u : (uint32_t)fit; }
    if (const char* e = getenv("ALZ_ENC_CHUNK")) { const long v = atol(e); if (v >= 64) CH = (uint32_t)v; }   // tuning knob
    EncScratch sc;
    alz_stream* d_streams = nullptr; alz_result* d_results = nullptr; alz_encode_aux* d_aux = nullptr; uint32_t* d_index = nullptr;
    uint64_t* d_pos = nullptr; int *d_head4 = nullptr, *d_headm = nullptr, *d_prev4 = nullptr, *d_prevm = nullptr; void *d_match = nullptr, *d_side = nullptr, *d_mask = nullptr;
    const uint32_t chn = n < CH ? n : CH;
    hipError_t e = sc.alloc((void**)&d_streams, (size_t)n * sizeof(alz_stream));
    if (e == hipSuccess) e = sc.alloc((void**)&d_results, (size_t)n * sizeof(alz_result));
    if (e == hipSuccess) e = sc.alloc((void**)&d_aux, (size_t)n * sizeof(alz_encode_aux));
    if (e == hipSuccess) e = sc.alloc((void**)&d_index, (size_t)n * sizeof(uint32_t));
    if (e == hipSuccess) e = sc.alloc((void**)&d_pos, (size_t)n * sizeof(uint64_t));
    if (e == hipSuccess) e = sc.alloc((void**)&d_head4, ((size_t)chn << hash_bits) * sizeof(int));
    if (e == hipSuccess && any_min) e = sc.alloc((void**)&d_headm, ((size_t)chn << 16) * sizeof(int));
    if (e == hipSuccess) e = sc.alloc((void**)&d_prev4, (size_t)total * sizeof(int));
    if (e == hipSuccess && any_min) e = sc.alloc((void**)&d_prevm, (size_t)total * sizeof(int));
    if (e == hipSuccess) e = sc.alloc(&d_match, (size_t)total * 8);
    if (e == hipSuccess && (cnt[ALZ_FMT_YAY0] || cnt[ALZ_FMT_MIO0] || cnt[ALZ_FMT_SMSR00])) e = sc.alloc(&d_side, (size_t)total * 2 + 64);   // section buffers
    if (e == hipSuccess) e = sc.alloc(&d_mask, (size_t)total / 8 + 64);
    if (e != hipSuccess) return fail(ALZ_E_NOMEM, "encoder scratch allocation failed: %s", hipGetErrorString(e));
    std::vector<uint32_t> index(n), foff(ALZ_FMT_COUNT, 0), fill(ALZ_FMT_COUNT, 0);
    { uint32_t off = 0; for (int f = 0; f < ALZ_FMT_COUNT; f++) { foff[f] = off; off += cnt[f]; } }
    for (uint32_t i = 0; i < n; i++) { uint32_t f = streams[i].format; index[foff[f] + fill[f]++] = i; }
    if (src_bytes) HIP_TRY(hipMemcpyAsync(c->d_src, src_base, src_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_streams, streams, (size_t)n * sizeof(alz_stream), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_index, index.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_pos, pos_off.data(), (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_results, 0xFF, (size_t)n * sizeof(alz_result), c->stream));
    HIP_TRY(hipMemsetAsync(d_aux, 0, (size_t)n * sizeof(alz_encode_aux), c->stream));
    HIP_TRY(hipMemsetAsync(d_mask, 0, (size_t)total / 8 + 64, c->stream));
    for (int f = 0; f < ALZ_FMT_COUNT; f++) {
        const void* g = geom.data() + f * alz_encode_geom_size();
        for (uint32_t done = 0; done < cnt[f]; done += CH) {
            const uint32_t k = cnt[f] - done < CH ? cnt[f] - done : CH;
            HIP_TRY(hipMemsetAsync(d_head4, 0xFF, ((size_t)k << alz_encode_geom_hash_bits(g)) * sizeof(int), c->stream));   // Reset(): tables = -1  :125-132
            if (alz_encode_geom_min_table(g)) HIP_TRY(hipMemsetAsync(d_headm, 0xFF, ((size_t)k << 16) * sizeof(int), c->stream));
            e = alz_launch_encode(f, c->stream, c->d_src, c->d_dst, d_streams, d_index + foff[f] + done, k, max_len, d_head4, d_headm, d_prev4, d_prevm,
                                  d_match, d_pos, d_side, d_mask, d_results, d_aux, g);
            if (e != hipSuccess) return fail(ALZ_E_HIP, "encode launch (format %d) failed: %s", f, hipGetErrorString(e));
        }
    }
    HIP_TRY(hipMemcpyAsync(results, d_results, (size_t)n * sizeof(alz_result), hipMemcpyDeviceToHost, c->stream));
    std::vector<alz_encode_aux> haux(n);
    HIP_TRY(hipMemcpyAsync(haux.data(), d_aux, (size_t)n * sizeof(alz_encode_aux), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n; i++) if (aux) aux[i] = haux[i];
    if ((rc = download_outputs(c, n, streams, results, dst_base, true))) return rc;
    return ALZ_OK;
}

}  // extern "C"
