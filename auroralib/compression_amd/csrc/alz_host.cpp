// alz_host.cpp -- context / plan management and the batch entry points of include/auroralz.h.
// Compiled by hipcc as host code; everything that computes runs in alz_kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <functional>
#include <thread>
#include <vector>
#include <sched.h>

#include "alz_internal.h"

static thread_local char g_err[512] = "";

// ALZ_TIMING=1: phase times of the host-buffer entry points on stderr (a tuning aid)
#include <chrono>
struct phase_timer {
    bool on; const char* what; std::chrono::steady_clock::time_point t;
    explicit phase_timer(const char* w) : on(getenv("ALZ_TIMING") != nullptr), what(w), t(std::chrono::steady_clock::now()) {}
    void mark(const char* phase) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[alz timing] %s: %s %.3f ms\n", what, phase, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// Host copies between the caller's (pageable) buffers and the pinned staging buffers run on a few threads: one core moves
// ~10 GB/s, a PCIe 5 x16 link ~55 GB/s, so a single memcpy loop left the link idle three quarters of the time.  One pool per
// context (the multi-GPU entry points drive one context per device from their own threads).  ALZ_COPY_THREADS overrides the
// thread count (1 = copy inline).
struct copy_job { uint8_t* dst; const uint8_t* src; size_t len; };
class copy_pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, done_cv;
    const copy_job* jobs = nullptr; size_t njobs = 0;
    std::atomic<size_t> next{0};
    size_t pending = 0; uint64_t gen = 0; bool stop = false;
    void work() { for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < njobs;) memcpy(jobs[i].dst, jobs[i].src, jobs[i].len); }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; }
            work();
            { std::lock_guard<std::mutex> l(m); if (--pending == 0) done_cv.notify_one(); }
        }
    }
public:
    static int default_threads() {
        if (const char* e = getenv("ALZ_COPY_THREADS")) { const int v = atoi(e); return v < 1 ? 1 : (v > 64 ? 64 : v); }
        cpu_set_t set; int n = 1;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
        n /= 2;
        return n < 1 ? 1 : (n > 16 ? 16 : n);
    }
    explicit copy_pool(int nthreads) { for (int i = 1; i < nthreads; i++) th.emplace_back([this] { loop(); }); }
    ~copy_pool() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        for (std::thread& t : th) t.join();
    }
    // runs every job; returns when all are done (the calling thread takes its share)
    void run(const std::vector<copy_job>& v) {
        size_t bytes = 0;
        for (const copy_job& j : v) bytes += j.len;
        if (th.empty() || bytes < (1u << 20)) { for (const copy_job& j : v) memcpy(j.dst, j.src, j.len); return; }
        { std::lock_guard<std::mutex> l(m); jobs = v.data(); njobs = v.size(); next.store(0); pending = th.size(); gen++; }
        cv.notify_all();
        work();
        std::unique_lock<std::mutex> l(m);
        done_cv.wait(l, [&] { return pending == 0; });
    }
};
// one copy, cut into pieces the pool can spread
static void add_copy(std::vector<copy_job>& v, uint8_t* dst, const uint8_t* src, size_t len) {
    const size_t piece = 1u << 20;
    for (size_t o = 0; o < len; o += piece) v.push_back(copy_job{dst + o, src + o, len - o < piece ? len - o : piece});
}

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
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};    // side streams of a mixed launch (with the launch stream: four lanes = the runtime's four hardware queues)
    hipEvent_t fork = nullptr, join[4] = {nullptr, nullptr, nullptr, nullptr};
    float last_kernel_ms = 0.f;                // device time of the kernels of the last timed / encode call (HIP events on the launch stream)
    bool exact = false;                        // alz_ctx_set_exact_kernels: the exact one-token-at-a-time kernels instead of the lane-parallel ones
    int variant = 0;                           // alz_ctx_set_kernel_variant
    uint64_t big_enc_launches = 0;             // streams the whole-GPU ENCODE path has taken (alz_encode_big.h)
    uint64_t seg_enc_launches = 0;             // launches of the segmented parse + emit (alz_encode_seg.h)
    int scan_mode = 0;                         // alz_debug_scan_mode: the encoder's scan path (enc_scan_select_kernel) -- 0 its probe decides, 1 every eligible stream, 2 off
    uint32_t seg_max_streams = 0xFFFFFFFFu;    // alz_debug_seg_max_streams: ~0 the rule, 0 the path off, else that many buffers instead of the rule -- of THIS context
    uint64_t chunk_repeats = 0;                // executes alz_plan_results repeated without the work queue (a bounded spin ran out)
    uint32_t big_min = 24u << 10;              // (24 KiB: tools/single_decode_sizes.py -- 0.18 ms either way at 16 KiB, 0.18 against 0.30 at 32) a lone stream of at least this many output bytes goes to the whole-GPU path (alz_big.hip)
    uint64_t big_launches = 0;                 // how often that path was enqueued
    uint32_t* d_big_accepted = nullptr;        // device words: [0] streams that path has ACCEPTED -- decoded itself, gate left closed (big_write bumps it; alz_ctx_big_stream reports it); [1] streams the encoder's scan path has taken
    hipEvent_t big_evt = nullptr; bool big_evt_set = false;   // behind the last whole-GPU decode that used d_bigbuf (plans that borrow it run one after the other)
    void* d_bigbuf = nullptr; size_t d_bigbuf_cap = 0;   // its scratch for the plans of the host-buffer entry points (grow-only)
    // two pinned staging buffers: host-buffer calls move the caller's (pageable) bytes through them, so that the memcpy of
    // one piece overlaps the PCIe transfer of the other
    void* pin[2] = {nullptr, nullptr}; size_t pin_cap = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_busy[2] = {false, false};                     // pin_ev[k] was recorded behind a DMA on pin[k] and nobody has waited for it yet
    // grow-only scratch of download_packed (item table + dense copy)
    void* d_items = nullptr; size_t d_items_cap = 0;
    void* d_pack = nullptr; size_t d_pack_cap = 0;
    void* d_plan = nullptr; size_t d_plan_cap = 0;   // plan arrays of the host-buffer entry points (no hipMalloc / hipFree per call)
    // encoder scratch (prev links, narrowed links, matches, masks ...: ~45 GB for 10 000 x 256 KiB at quality 8), one grow-only slot
    // per purpose: allocating and freeing it per call cost 1-2 s, four times the kernels.  alz_ctx_release_scratch() returns it.
    void* enc_buf[15] = {nullptr}; size_t enc_cap[15] = {0};
    copy_pool* pool = nullptr;                 // created with the pinned buffers
    std::vector<copy_job> jobs;                // (scratch of the staging loops)
    void copy(uint8_t* dst, const uint8_t* src, size_t len) { jobs.clear(); add_copy(jobs, dst, src, len); pool->run(jobs); }
};

static const size_t kPinBytes = 32u << 20;
#define ALZ_BIG_MAX_STREAMS 32u             /* a batch of at most this many streams, all of them big, may take the whole-GPU path stream by stream (plan_create weighs it) */

struct alz_plan {
    uint32_t n = 0;
    alz_lz_properties lz{};
    alz_stream* d_streams = nullptr;
    alz_result* d_results = nullptr;
    uint32_t* d_index = nullptr;            // concatenated per-format index lists
    uint32_t fmt_off[ALZ_FMT_COUNT] = {0};
    uint32_t fmt_cnt[ALZ_FMT_COUNT] = {0};
    bool borrowed = false;                  // the three device arrays live in the context's plan scratch (host-buffer entry points)
    std::vector<uint32_t> index_host;       // such a plan keeps its index list: its upload is not waited for (plan_create)
    // ONE big Yay0 / MIO0 stream: decoded by the whole GPU (alz_big.hip), the production kernel behind it only if that path declines
    // (or a few of them, one after the other: n streams through that path take n x ~0.1-0.3 ms, on wavefronts of their own they take as
    // long as the largest of them alone, 1-5 ms per MiB; plan_create weighs the two)
    bool big = false, big_borrowed = false; std::vector<alz_stream> big_streams; std::vector<uint32_t> big_pos; void* d_big = nullptr; uint32_t* d_gate = nullptr;
    // A big plan owns MUTABLE device state (val / jump / ctl / gate in d_big): two executes in flight on different streams would race on it.
    // Every execute waits for the event the one before recorded (the context's event when the scratch is the context's: several plans share it).
    hipEvent_t big_evt = nullptr; bool big_evt_set = false;
    // the stream of the last execute + an event behind it: alz_plan_results orders its copy behind the caller's own stream too
    hipEvent_t done_evt = nullptr; bool done_evt_set = false;
    // The flag-byte family as a work queue of (stream, chunk) items (alz_decode_fastq_kernel): for a format whose streams are more than the GPU holds
    // wavefronts, so that the launch does not end in a partly filled round.  One allocation: [64 control words | a 128-byte line per slot's flag | items | slots].
    struct chunk_plan { uint32_t n_items = 0, n_slots = 0, lw = 0; void* d_mem = nullptr; uint32_t* d_ctl = nullptr; uint32_t* d_flags = nullptr;
                        alz_chunk_item* d_items = nullptr; uint8_t* d_slots = nullptr; size_t zero_bytes = 0; alz_queue_bounds bounds{}; };
    chunk_plan chunk[ALZ_FMT_COUNT];
    // A queue plan owns MUTABLE device state too (queue heads, flags, hand-over slots): its executes are ordered one behind the other by `q_evt`, on whatever streams
    // the caller enqueues them (round 6; big plans: `big_evt`).  `d_tmo`: one STICKY word per format outside the region an execute zeroes -- a bounded spin that ran out
    // sets it, the gated launch enqueued behind every queue launch then decodes that format again with one wavefront per stream IN STREAM ORDER (so every execute is
    // whole when its stream gets past it, whichever buffers it ran on), nothing ever clears it, and alz_plan_results, seeing it, moves the plan off the queue for good.
    uint32_t* d_tmo = nullptr;
    hipEvent_t q_evt = nullptr; bool q_evt_set = false;     // q_evt_set: the event stands behind the last execute on a stream of the CALLER's
    bool q_ctx_pending = false;              // an execute went onto the context's own stream since: ordered by that stream itself, its event is recorded only when another stream needs it
    uint32_t epoch = 0;                      // of the last queue launch (1 .. 2^30 - 1; flags and heads are zeroed once, at creation, and again when the counter wraps)
    bool no_chunks = false;                  // a bounded spin of the queue kernel ran out once: this plan stays with the one-wavefront-per-stream kernels
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
int alz_ctx_big_stream(alz_ctx* c, uint32_t min_bytes, uint64_t* launches_out) {
    if (!c) return fail(ALZ_E_INVALID, "alz_ctx_big_stream: ctx is NULL");
    if (min_bytes) c->big_min = min_bytes;
    if (launches_out) {
        // streams the two whole-GPU paths have TAKEN: the decode side counts on the device, where the path decides (a stream it declines goes to the kernel behind the gate
        // and is not counted); everything enqueued so far on any stream is waited for, so the number is exact when the caller asks
        uint32_t acc = 0;
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipDeviceSynchronize());
        if (c->d_big_accepted) HIP_TRY(hipMemcpy(&acc, c->d_big_accepted, sizeof(acc), hipMemcpyDeviceToHost));
        *launches_out = (uint64_t)acc + c->big_enc_launches;
    }
    return ALZ_OK;
}
int alz_ctx_set_kernel_variant(alz_ctx* c, int variant) {
    if (!c || variant < 0 || variant > 3) return fail(ALZ_E_INVALID, "alz_ctx_set_kernel_variant: bad argument");
    c->variant = variant;
    return ALZ_OK;
}
int alz_ctx_set_exact_kernels(alz_ctx* c, int on) {
    if (!c) return fail(ALZ_E_INVALID, "ctx is NULL");
    c->exact = on != 0;
    return ALZ_OK;
}
/* not in the public header: resident waves per CU of the production kernel of `format` (tuning aid) */
int alz_debug_occupancy(int format) { return alz_kernel_occupancy(format); }
/* not in the public header: the largest batch (buffers of one format) whose parse + emit runs over segments (alz_encode_seg.h; 0: never), and how often a context has gone that way */
void alz_debug_seg_max_streams(alz_ctx* c, uint32_t v) { if (c) c->seg_max_streams = v; }
/* not in the public header: the encoder's scan path (streams whose parse visits few positions go without kernels A and B): 0 the probe decides, 1 every eligible stream, 2 off */
void alz_debug_scan_mode(alz_ctx* c, int mode) { if (c && mode >= 0 && mode <= 2) c->scan_mode = mode; }
/* ... and how many streams have gone that way on this context (counted on the device, where the probe decides; waits for the device) */
uint64_t alz_debug_scan_streams(alz_ctx* c) {
    uint32_t v = 0;
    if (!c || !c->d_big_accepted || hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, c->d_big_accepted + 1, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return v;
}
uint64_t alz_debug_seg_launches(const alz_ctx* c) { return c ? c->seg_enc_launches : 0; }
/* not in the public header: how often alz_plan_results has repeated a launch without the work queue on this context (expected: never) */
uint64_t alz_debug_chunk_repeats(const alz_ctx* c) { return c ? c->chunk_repeats : 0; }
/* not in the public header: output bytes per chunk of the work-queue kernels */
int alz_debug_chunk_bytes(void) { return (int)ALZ_CHUNK_OUT; }
/* not in the public header: the (stream, chunk) items of a plan's work queues (0: the plan decodes with one wavefront per stream) */
/* not in the public header: the control words of a format's work queue (256 (e & 1) + 32 q, q < 8: the head of sub-queue q in the set launch e drew from = tickets drawn; word 1: the plan's sticky
   timeout word for the format, which lives elsewhere and is shown here) and, behind them from word ALZ_CHUNK_CTL_WORDS = 512 on, the flag of every hand-over slot (one per 32 words), reduced to the
   state the LAST launch left */
int alz_debug_plan_queue_ctl(alz_plan* p, int fmt, uint32_t* out, uint32_t nwords) {
    if (!p || fmt < 0 || fmt >= ALZ_FMT_COUNT || !p->chunk[fmt].d_ctl) return -1;
    const uint32_t have = ALZ_CHUNK_CTL_WORDS + p->chunk[fmt].n_slots * ALZ_CHUNK_FLAG_STRIDE;
    if (nwords > have) nwords = have;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, p->chunk[fmt].d_ctl, 4 * (size_t)nwords, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (uint32_t k = ALZ_CHUNK_CTL_WORDS; k < nwords; k += ALZ_CHUNK_FLAG_STRIDE) out[k] = (out[k] >> 2) == p->epoch ? (out[k] & 3u) : 0u;   // (flags of the LAST launch: 0 not set, 1 handed over, 2 ended, 3 timed out)
    if (nwords > 1u && p->d_tmo && hipMemcpy(out + 1, p->d_tmo + fmt, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)nwords;
}
int alz_debug_plan_queue_items(alz_plan* p) { int n = 0; if (p) for (int f = 0; f < ALZ_FMT_COUNT; f++) n += (int)p->chunk[f].n_items; return n; }
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
    for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_big_accepted, 256);
    if (e == hipSuccess) e = hipMemset(c->d_big_accepted, 0, 256);
    if (e != hipSuccess) { alz_destroy(c); return fail(ALZ_E_HIP, "context creation failed: %s", hipGetErrorString(e)); }   // (what was created so far goes with it)
    *out = c;
    return ALZ_OK;
}

static void release_scratch(alz_ctx* c);
void alz_destroy(alz_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    release_scratch(c);
    for (int i = 0; i < 2; i++) { if (c->pin[i]) (void)hipHostFree(c->pin[i]); if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]); }
    if (c->big_evt) (void)hipEventDestroy(c->big_evt);
    if (c->d_big_accepted) (void)hipFree(c->d_big_accepted);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->fork) (void)hipEventDestroy(c->fork);
    for (int i = 0; i < 4; i++) if (c->join[i]) (void)hipEventDestroy(c->join[i]);
    for (int i = 0; i < 3; i++) if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c->pool;
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
    HIP_TRY(hipMalloc(d_ptr, bytes + 64));              // (+ the read slack of the device-resident entry points, auroralz.h)
    return ALZ_OK;
}
int alz_device_free(alz_ctx* c, void* d_ptr) {
    if (!c) return fail(ALZ_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipFree(d_ptr));
    return ALZ_OK;
}
static int staged_h2d(alz_ctx* c, void* d_dst, const uint8_t* h_src, size_t bytes);
static int staged_d2h(alz_ctx* c, uint8_t* h_dst, const void* d_src, size_t bytes);
int alz_memcpy_h2d(alz_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    if (!c) return fail(ALZ_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = staged_h2d(c, d_dst, (const uint8_t*)h_src, bytes)) return rc;      // (pageable memory: through the pinned buffers + copy threads)
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ALZ_OK;
}
int alz_memcpy_d2h(alz_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    if (!c) return fail(ALZ_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = staged_d2h(c, (uint8_t*)h_dst, d_src, bytes)) return rc;
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
    if (!p->borrowed) {
        if (p->d_streams) (void)hipFree(p->d_streams);
        if (p->d_results) (void)hipFree(p->d_results);
        if (p->d_index) (void)hipFree(p->d_index);
    }
    if (p->d_big && !p->big_borrowed) (void)hipFree(p->d_big);
    if (p->big_evt) (void)hipEventDestroy(p->big_evt);
    if (p->q_evt) (void)hipEventDestroy(p->q_evt);
    if (p->d_tmo) (void)hipFree(p->d_tmo);
    if (p->done_evt) (void)hipEventDestroy(p->done_evt);
    for (int f = 0; f < ALZ_FMT_COUNT; f++) if (p->chunk[f].d_mem) (void)hipFree(p->chunk[f].d_mem);
    delete p;
}

static int grow(alz_ctx* c, void** buf, size_t* cap, size_t need);
// `scratch`: the plan's device arrays come out of the context's grow-only plan scratch (one plan at a time: the host-buffer
// entry points, which create, run and drop a plan inside one call -- a hipMalloc / hipFree trio per call cost more than the
// kernel of a small batch)
static int plan_create(alz_ctx* c, const alz_lz_properties* props, uint32_t n, const alz_stream* streams, alz_plan** out, bool scratch) {
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
    // Longest first inside a format's launch: workgroups are dispatched in index order and a CU holds ~24 of them, so the
    // costly streams start at once and the cheap ones fill the tail.  A stream's cost is its tokens (~ compressed bytes)
    // plus its output bytes: the 256 KiB windows of Test.bmp take 2.8 ms (flat, ratio 0.02) to 9.2 ms (photographic, ratio
    // 0.59) per 10 000; in input order that batch ran 5.8 ms per launch (tools/realistic_windows.py).
    auto cost = [&](uint32_t i) { return 3ull * streams[i].src_len + (streams[i].decom_len ? streams[i].decom_len : streams[i].dst_cap); };
    for (int f = 0; f < ALZ_FMT_COUNT; f++)
        if (cnt[f] > 1) std::stable_sort(index.begin() + p->fmt_off[f], index.begin() + p->fmt_off[f] + cnt[f], [&](uint32_t a, uint32_t b) { return cost(a) > cost(b); });
    size_t nn = n ? n : 1;
    hipError_t e = hipSuccess;
    if (scratch) {
        const size_t a = (nn * sizeof(alz_stream) + 255) & ~(size_t)255, b = (nn * sizeof(alz_result) + 255) & ~(size_t)255;
        if (int rc = grow(c, &c->d_plan, &c->d_plan_cap, a + b + nn * sizeof(uint32_t))) { delete p; return rc; }
        p->borrowed = true;
        p->d_streams = (alz_stream*)c->d_plan; p->d_results = (alz_result*)((uint8_t*)c->d_plan + a); p->d_index = (uint32_t*)((uint8_t*)c->d_plan + a + b);
    } else {
        e = hipMalloc((void**)&p->d_streams, nn * sizeof(alz_stream));
        if (e == hipSuccess) e = hipMalloc((void**)&p->d_results, nn * sizeof(alz_result));
        if (e == hipSuccess) e = hipMalloc((void**)&p->d_index, nn * sizeof(uint32_t));
    }
    if (e == hipSuccess && n) e = hipMemcpyAsync(p->d_streams, streams, n * sizeof(alz_stream), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(p->d_index, index.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_results, 0xFF, nn * sizeof(alz_result), c->stream);
    // (a plan of the library's own -- one host-buffer call: created, executed, read, destroyed -- does not wait here: the caller's stream
    // table outlives the call, the index list moves into the plan, and the kernels queue up behind the uploads; the one stream of a
    // format class's Decompress call pays ~30 us for every synchronisation)
    if (scratch) p->index_host = std::move(index);
    else if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipStreamSynchronize(c->stream); alz_plan_destroy(c, p); return fail(ALZ_E_HIP, "plan upload failed: %s", hipGetErrorString(e)); }
    bool all_big = n >= 1 && n <= ALZ_BIG_MAX_STREAMS && !c->exact && c->variant == 0;
    for (uint32_t i = 0; all_big && i < n; i++) all_big = alz_big_eligible((int)streams[i].format, &streams[i], &lz, c->big_min);
    if (all_big && n > 1) {
        // several big streams: one after the other through the whole-GPU path (measured: ~0.08 ms of launches + 0.1 ms per MiB of output
        // each) against all of them side by side on wavefronts of their own, which takes as long as the LARGEST of them alone (~4.4 ms per
        // MiB): 16 streams of 1 MiB 2.9 against 4.4 ms, 32 streams of 256 KiB 3.4 against 1.1
        double t_big = 0, t_lone = 0;
        for (uint32_t i = 0; i < n; i++) {
            const double mib = (double)(streams[i].decom_len ? streams[i].decom_len : streams[i].dst_cap) / 1048576.0;
            t_big += 0.08 + 0.1 * mib;
            if (4.4 * mib > t_lone) t_lone = 4.4 * mib;
        }
        all_big = t_big < t_lone;
    }
    if (all_big) {
        // (the scratch -- 4 bytes per output byte of the largest stream, they run one after the other -- belongs to the plan; when it cannot
        // be had the production kernels decode the streams)
        size_t need = 0;
        for (uint32_t i = 0; i < n; i++) { const size_t b = alz_big_scratch_bytes((int)streams[i].format, &streams[i]) + 64; if (b > need) need = b; }
        if (scratch) { if (grow(c, &c->d_bigbuf, &c->d_bigbuf_cap, need) == ALZ_OK) { p->d_big = c->d_bigbuf; p->big_borrowed = true; } }
        else if (hipMalloc(&p->d_big, need) != hipSuccess) { p->d_big = nullptr; (void)hipGetLastError(); }
        if (p->d_big) {
            p->big = true; p->big_streams.assign(streams, streams + n); p->d_gate = (uint32_t*)((uint8_t*)p->d_big + need - 64);
            p->big_pos.resize(n);
            const std::vector<uint32_t>& ix = scratch ? p->index_host : index;
            for (uint32_t k = 0; k < n; k++) p->big_pos[ix[k]] = k;      // where stream i sits in the (per-format, cost-ordered) index list
        }
    }
    // ---- the work queue of chunks (device-resident plans): a format whose streams are more than the GPU holds of its wavefronts, and
    // long enough to be cut.  Failing to get the memory only means the one-wavefront-per-stream launch.
    const bool force_queue = c->variant == 3;     // (plans created while the context is in variant 3 use the queue whatever the number of streams: the parity tests)
    if (!scratch && !c->exact && (force_queue || (!p->big && c->variant == 0))) {
        hipDeviceProp_t pr;
        const bool have_pr = hipGetDeviceProperties(&pr, c->device) == hipSuccess;
        for (int f = 0; f < ALZ_FMT_COUNT && have_pr; f++) {
            uint32_t lw = 0;
            if (!cnt[f] || !alz_chunk_format(f, &lz, &lw)) continue;
            const int occ = alz_chunk_places_per_cu(f);
            const uint32_t chb = alz_chunk_bytes(f);
            // (from 0.6 of the GPU's wavefront places on: 10 000 x 256 KiB as Yaz0, ms per launch one wavefront per stream / queue -- 3 000 streams 1.12 (two wavefronts each) / 1.33,
            // 4 000 1.44 / 1.40, 5 000 1.60 / 1.49, 6 400 2.43 / 1.69, 8 000 2.09 (queue); below that everything is resident at once and a stream's own latency decides)
            // -- counted per FORMAT: in a mixed batch the formats' kernels run side by side, and queues whose items wait for each other take the places the others need (the cfg4 shard, 5 000 mixed
            // streams: 2.54 ms with one wavefront per stream, 3.20 with its three flag-byte formats as queues)
            // PRS (two wavefronts per stream, bound by the scalar pipe: a stream in a half-empty GPU is fast anyway) only above what the GPU holds: 2 000 / 3 000 / 4 000 / 6 000 / 10 000 streams
            // 1.39 / 1.69 / 2.72 / 3.43 / 5.50 ms with a workgroup per stream, 1.59 / 1.80 / 2.34 / 3.32 / 5.36 as a queue (3 072 places)
            // (with 40 KiB chunks and the descriptor table -- 29 places per CU for Yaz0 / LZ11 -- the queue wins earlier: 3 600 / 4 200 streams 1.44 / 1.58 ms one wavefront per stream, 1.37 / 1.42 as a
            // queue: from 0.4 of the places on, and never where two wavefronts share a stream -- up to 3 072 streams: 3 000 streams 1.14 that way, 1.31 as a queue)
            const uint64_t tenths = (f == ALZ_FMT_PRS_BE || f == ALZ_FMT_PRS_LE) ? 10ull : 4ull;
            if (!force_queue && (occ < 1 || cnt[f] <= alz_two_wave_max() || 10ull * cnt[f] <= tenths * (uint64_t)occ * (uint64_t)pr.multiProcessorCount)) continue;
            // items in chunk-major order over the format's cost-ordered list; a stream's slots are consecutive
            std::vector<uint32_t> nch(cnt[f]), base(cnt[f]);
            uint32_t slots = 0, maxch = 0; uint64_t items64 = 0;
            for (uint32_t k = 0; k < cnt[f]; k++) {
                const alz_stream& st = streams[index[p->fmt_off[f] + k]];
                const uint32_t bound = st.decom_len ? (st.decom_len < st.dst_cap ? st.decom_len : st.dst_cap) : st.dst_cap;
                uint32_t m = (bound + chb - 1u) / chb; if (m == 0) m = 1;
                nch[k] = m; base[k] = slots; slots += m; items64 += m; if (m > maxch) maxch = m;
            }
            if (maxch < 2 || items64 > 0x3FFFFFFFull) continue;
            // ALZ_QUEUE_SHARDS sub-queues, one behind the other: stream k of the cost-ordered list goes to sub-queue k mod 8 (every sub-queue gets its share of the costly
            // streams, and the sub-queues come out equally long: workgroups are dealt round-robin over the XCDs and each prefers its XCD's sub-queue), chunk-major inside
            std::vector<alz_chunk_item> items; items.reserve((size_t)items64);
            alz_queue_bounds qb{};
            for (uint32_t q = 0; q < ALZ_QUEUE_SHARDS; q++) {
                qb.off[q] = (uint32_t)items.size();
                for (uint32_t ch = 0; ch < maxch; ch++)
                    for (uint32_t k = q; k < cnt[f]; k += ALZ_QUEUE_SHARDS)
                        if (ch < nch[k]) items.push_back(alz_chunk_item{index[p->fmt_off[f] + k], ch, base[k] + ch, ch + 1u == nch[k] ? 1u : 0u});
            }
            qb.off[ALZ_QUEUE_SHARDS] = (uint32_t)items.size();
            alz_plan::chunk_plan& cp = p->chunk[f];
            const size_t zero_bytes = (4 * (size_t)ALZ_CHUNK_CTL_WORDS + (size_t)slots * 4 * ALZ_CHUNK_FLAG_STRIDE + 255) & ~(size_t)255, items_bytes = (items.size() * sizeof(alz_chunk_item) + 255) & ~(size_t)255;
            const size_t slot_bytes = (size_t)slots * (32 + lw);
            if (hipMalloc(&cp.d_mem, zero_bytes + items_bytes + slot_bytes + 256) != hipSuccess) { cp.d_mem = nullptr; (void)hipGetLastError(); continue; }
            cp.d_ctl = (uint32_t*)cp.d_mem; cp.d_flags = cp.d_ctl + ALZ_CHUNK_CTL_WORDS; cp.zero_bytes = zero_bytes; cp.bounds = qb;
            cp.d_items = (alz_chunk_item*)((uint8_t*)cp.d_mem + zero_bytes); cp.d_slots = (uint8_t*)cp.d_mem + zero_bytes + items_bytes;
            if (hipMemset(cp.d_mem, 0, zero_bytes) != hipSuccess || hipMemcpy(cp.d_items, items.data(), items.size() * sizeof(alz_chunk_item), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(cp.d_mem); cp = alz_plan::chunk_plan(); (void)hipGetLastError(); continue; }
            cp.n_items = (uint32_t)items.size(); cp.n_slots = slots; cp.lw = lw;
            if (!p->d_tmo) {                                   // the sticky timeout words, one per format (zeroed ONCE, here)
                if (hipMalloc((void**)&p->d_tmo, 256) != hipSuccess || hipMemset(p->d_tmo, 0, 256) != hipSuccess) {
                    if (p->d_tmo) (void)hipFree(p->d_tmo);
                    p->d_tmo = nullptr; (void)hipFree(cp.d_mem); cp = alz_plan::chunk_plan(); (void)hipGetLastError();
                }
            }
        }
    }
    *out = p;
    return ALZ_OK;
}
int alz_plan_create(alz_ctx* c, const alz_lz_properties* props, uint32_t n, const alz_stream* streams, alz_plan** out) {
    return plan_create(c, props, n, streams, out, false);
}

static uint32_t format_weight(uint32_t fmt);
// Time of ONE 256 KiB stream alone on the GPU, in units of 10 us (measured: docs/EXPERIMENTS.md 4.4 / 8): what orders the kernels of a mixed launch
static uint32_t format_latency(uint32_t fmt) {
    switch (fmt) {
    case ALZ_FMT_YAY0: return 91;
    case ALZ_FMT_YAZ0: case ALZ_FMT_LZ11: case ALZ_FMT_LZ40: case ALZ_FMT_LZ02: case ALZ_FMT_PRS_BE: case ALZ_FMT_PRS_LE: return 109;
    case ALZ_FMT_MIO0: case ALZ_FMT_SMSR00: return 125;
    case ALZ_FMT_LZSS: case ALZ_FMT_LZ10: case ALZ_FMT_CLZ0: case ALZ_FMT_CNS: case ALZ_FMT_LZHUDSON: case ALZ_FMT_LZSHREK: return 154;
    case ALZ_FMT_LZ4_BLOCK: case ALZ_FMT_FASTLZ: case ALZ_FMT_WFLZ: case ALZ_FMT_WFLZ_BE: case ALZ_FMT_HIG: case ALZ_FMT_CNX2: return 170;
    case ALZ_FMT_BLZ: case ALZ_FMT_REFPACK: case ALZ_FMT_SNAPPY_RAW: return 200;
    default: return 240;                                     // LZO
    }
}
// An execute on a stream of the CALLER's leaves an event behind it, so that alz_plan_results (which copies on the context's own,
// non-blocking stream) is ordered behind those kernels; executes on the context's stream are ordered by the stream itself.
static int plan_mark_done(alz_plan* p, hipStream_t s, bool foreign) {
    if (!foreign) { p->done_evt_set = false; return ALZ_OK; }
    if (!p->done_evt) HIP_TRY(hipEventCreateWithFlags(&p->done_evt, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->done_evt, s));
    p->done_evt_set = true;
    return ALZ_OK;
}
int alz_plan_execute(alz_ctx* c, alz_plan* p, const void* d_src_base, void* d_dst_base, void* hip_stream) {
    if (!c || !p) return fail(ALZ_E_INVALID, "alz_plan_execute: bad argument");
    HIP_TRY(hipSetDevice(c->device));                 // (a host thread may hold contexts of several devices)
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (p->big && !c->exact && c->variant == 0) {
        hipEvent_t* ev = p->big_borrowed ? &c->big_evt : &p->big_evt;
        bool* ev_set = p->big_borrowed ? &c->big_evt_set : &p->big_evt_set;
        if (!*ev) HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
        if (*ev_set) HIP_TRY(hipStreamWaitEvent(s, *ev, 0));           // (a no-op on the stream that recorded it; orders any other stream behind the last run)
        for (uint32_t i = 0; i < p->n; i++) {
            const int f = (int)p->big_streams[i].format;
            hipError_t e = alz_launch_big(f, s, d_src_base, d_dst_base, &p->big_streams[i], &p->lz, p->d_results + i, p->d_big, p->d_gate, c->d_big_accepted);
            if (e == hipSuccess) e = alz_launch_decode_gated(f, s, d_src_base, d_dst_base, p->d_streams, p->d_index + p->big_pos[i], 1, p->d_results, &p->lz, p->d_gate);
            if (e != hipSuccess) return fail(ALZ_E_HIP, "big-stream launch (format %d) failed: %s", f, hipGetErrorString(e));
            c->big_launches++;
        }
        HIP_TRY(hipEventRecord(*ev, s)); *ev_set = true;
        return plan_mark_done(p, s, s != c->stream);
    }
    int nfmt = 0;
    for (int f = 0; f < ALZ_FMT_COUNT; f++) nfmt += p->fmt_cnt[f] ? 1 : 0;
    // The work queues of this plan are in use by this execute: it starts behind the execute before it (the queue heads, flags and slots are the plan's, one set),
    // wherever that one was enqueued, and leaves the event for the next.  Plans without a queue only read their tables.
    const bool queued = p->d_tmo && !c->exact && (c->variant == 0 || c->variant == 3) && !p->no_chunks;
    const bool foreign = s != c->stream;
    if (queued) {
        // (executes on the context's own stream -- the usual case, and bench.py's -- are ordered by the stream and cost no event: a record per execute is a
        // barrier packet between the kernels of a 0.7 ms launch.  A stream of the caller's may be gone by the next call, so its event is recorded at once.)
        if (!p->q_evt) HIP_TRY(hipEventCreateWithFlags(&p->q_evt, hipEventDisableTiming));
        if (foreign && p->q_ctx_pending) { HIP_TRY(hipEventRecord(p->q_evt, c->stream)); p->q_evt_set = true; p->q_ctx_pending = false; }
        if (p->q_evt_set) { HIP_TRY(hipStreamWaitEvent(s, p->q_evt, 0)); if (!foreign) p->q_evt_set = false; }
        // this launch's epoch: what makes a hand-over flag count, and which of the two sets of queue heads is drawn from -- nothing is zeroed per launch.  When the
        // 30-bit counter comes round (a billion executes) everything is zeroed once, as at creation.
        if (p->epoch >= 0x3FFFFFFFu) {
            for (int f = 0; f < ALZ_FMT_COUNT; f++) if (p->chunk[f].n_items) HIP_TRY(hipMemsetAsync(p->chunk[f].d_ctl, 0, p->chunk[f].zero_bytes, s));
            p->epoch = 0;
        }
        p->epoch++;
    }
    // a format's launch: the work queue of chunks where the plan has one (and the context has not been switched to other kernels since), else one wavefront per stream
    auto launch_format = [&](int f, hipStream_t on) -> hipError_t {
        alz_plan::chunk_plan& cp = p->chunk[f];
        if (cp.n_items && queued) {
            hipError_t e = alz_launch_decode_chunked(f, on, d_src_base, d_dst_base, p->d_streams, cp.d_items, cp.n_items, &cp.bounds, p->d_results, &p->lz, cp.d_ctl, cp.d_flags, cp.d_slots, p->d_tmo + f, p->epoch);
            // ... and behind it the same streams with one wavefront per stream, gated by that word: workgroups that return at once (always, so far), or the whole repair in stream order
            if (e == hipSuccess) e = alz_launch_decode_gated(f, on, d_src_base, d_dst_base, p->d_streams, p->d_index + p->fmt_off[f], p->fmt_cnt[f], p->d_results, &p->lz, p->d_tmo + f);
            return e;
        }
        return alz_launch_decode(f, on, d_src_base, d_dst_base, p->d_streams, p->d_index + p->fmt_off[f], p->fmt_cnt[f], p->d_results, &p->lz, c->exact, p->n, c->variant);
    };
    if (nfmt <= 1) {
        for (int f = 0; f < ALZ_FMT_COUNT; f++) {
            if (!p->fmt_cnt[f]) continue;
            hipError_t e = launch_format(f, s);
            if (e != hipSuccess) return fail(ALZ_E_HIP, "kernel launch (format %d) failed: %s", f, hipGetErrorString(e));
        }
        if (queued) { if (foreign) { HIP_TRY(hipEventRecord(p->q_evt, s)); p->q_evt_set = true; } else p->q_ctx_pending = true; }
        return plan_mark_done(p, s, s != c->stream);
    }
    // mixed batch: one kernel per format, forked onto side streams so that they share the GPU (each format alone may
    // have far fewer streams than the device has wave slots), joined back into the caller's stream -- also when a launch
    // fails half way (the side streams that already started are joined, then the error is reported)
    // The format whose streams take longest ALONE goes first: a mixed launch is resident all at once and ends with its slowest
    // stream, and a stream's own time hardly depends on how many others of its format there are.
    // Four lanes: the caller's stream itself and three side streams.  Round 2 forked onto FOUR side streams; with the caller's
    // stream that made five, the HIP runtime maps streams onto four hardware queues, and rocprofv3 showed the fourth kernel of
    // the cfg4 shard waiting in the queue of the third until that had finished (3.09 ms for kernels of 2.2 / 1.8 / 1.8 / 1.3 ms).
    int order[ALZ_FMT_COUNT];
    for (int f = 0; f < ALZ_FMT_COUNT; f++) order[f] = f;
    std::stable_sort(order, order + ALZ_FMT_COUNT, [](int a, int b) { return format_latency((uint32_t)a) > format_latency((uint32_t)b); });
    HIP_TRY(hipEventRecord(c->fork, s));
    // Formats beyond the fourth queue up behind an earlier kernel: each goes to the lane with the least work queued so far (longest
    // first onto the least loaded lane; a kernel's share of the launch ~ its streams x the format's cost per byte) -- dealt round-robin,
    // the fifth format waited behind the slowest kernel of all while the faster lanes drained.
    int rc = ALZ_OK; bool used[4] = {false, false, false, false};
    uint64_t load[4] = {0, 0, 0, 0};
    for (int oi = 0; oi < ALZ_FMT_COUNT && rc == ALZ_OK; oi++) {
        const int f = order[oi];
        if (!p->fmt_cnt[f]) continue;
        int lane = 0;                                       // lane 0: the caller's stream; 1..3: side streams
        for (int l = 1; l < 4; l++) if (load[l] < load[lane]) lane = l;
        load[lane] += (uint64_t)p->fmt_cnt[f] * format_weight((uint32_t)f) + 1u;
        hipStream_t a = lane == 0 ? s : c->aux[lane - 1];
        if (lane != 0 && !used[lane]) {
            hipError_t w = hipStreamWaitEvent(a, c->fork, 0);
            if (w != hipSuccess) { rc = fail(ALZ_E_HIP, "hipStreamWaitEvent failed: %s", hipGetErrorString(w)); break; }
            used[lane] = true;
        }
        hipError_t e = launch_format(f, a);
        if (e != hipSuccess) rc = fail(ALZ_E_HIP, "kernel launch (format %d) failed: %s", f, hipGetErrorString(e));
    }
    for (int i = 1; i < 4; i++) if (used[i]) {
        hipError_t e = hipEventRecord(c->join[i], c->aux[i - 1]);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, c->join[i], 0);
        if (e != hipSuccess && rc == ALZ_OK) rc = fail(ALZ_E_HIP, "joining the side streams failed: %s", hipGetErrorString(e));
    }
    if (queued) {
        if (!foreign) p->q_ctx_pending = true;
        else { hipError_t e = hipEventRecord(p->q_evt, s); if (e == hipSuccess) p->q_evt_set = true; else if (rc == ALZ_OK) rc = fail(ALZ_E_HIP, "hipEventRecord failed: %s", hipGetErrorString(e)); }
    }
    if (rc == ALZ_OK) rc = plan_mark_done(p, s, s != c->stream);
    return rc;
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
    c->last_kernel_ms = *mean_ms;
    return ALZ_OK;
}

int alz_plan_results(alz_ctx* c, alz_plan* p, alz_result* results) {
    if (!c || !p || (p->n && !results)) return fail(ALZ_E_INVALID, "alz_plan_results: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (p->done_evt_set) HIP_TRY(hipStreamWaitEvent(c->stream, p->done_evt, 0));    // (the last execute ran on a stream of the caller's)
    // the work queue's bounded spins: if one ever ran out (never seen), the gated launch behind that execute has already decoded the format again with one wavefront per
    // stream -- in stream order, so the results and every execute's bytes are whole -- and the word stays set: the plan leaves the queue for good
    uint32_t tmo[ALZ_FMT_COUNT] = {0};
    const bool look = p->d_tmo && !p->no_chunks;
    if (look) HIP_TRY(hipMemcpyAsync(tmo, p->d_tmo, sizeof(tmo), hipMemcpyDeviceToHost, c->stream));
    if (p->n) HIP_TRY(hipMemcpyAsync(results, p->d_results, p->n * sizeof(alz_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (look) for (int f = 0; f < ALZ_FMT_COUNT; f++) if (tmo[f]) { p->no_chunks = true; c->chunk_repeats++; break; }
    return ALZ_OK;
}

// ---------------------------------------------------------------- host-buffer entry points
// off + len <= total without wrapping (offsets are caller-supplied 64-bit values)
static inline bool range_ok(uint64_t off, uint64_t len, uint64_t total) { return off <= total && len <= total - off; }

static int ensure_pinned(alz_ctx* c) {
    if (c->pin_cap) return ALZ_OK;
    for (int i = 0; i < 2; i++) {
        if (!c->pin[i]) HIP_TRY(hipHostMalloc(&c->pin[i], kPinBytes, hipHostMallocDefault));
        if (!c->pin_ev[i]) HIP_TRY(hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
    }
    if (!c->pool) c->pool = new (std::nothrow) copy_pool(copy_pool::default_threads());
    if (!c->pool) return fail(ALZ_E_NOMEM, "out of memory");
    c->pin_cap = kPinBytes;                                  // (set last: everything above exists from here on)
    return ALZ_OK;
}
// A pinned buffer is only touched by the host once the DMA recorded on it has completed -- also across calls: the flag lives in
// the context, so a second staging call that follows at once (or one that follows a call that returned early with an error)
// waits for the first call's DMA instead of overwriting its source.
static int pin_wait(alz_ctx* c, int k) {
    if (c->pin_busy[k]) { HIP_TRY(hipEventSynchronize(c->pin_ev[k])); c->pin_busy[k] = false; }
    return ALZ_OK;
}
static int pin_mark(alz_ctx* c, int k) {
    HIP_TRY(hipEventRecord(c->pin_ev[k], c->stream)); c->pin_busy[k] = true;
    return ALZ_OK;
}
// host (pageable) -> device through the two pinned buffers: the memcpy of piece k + 1 overlaps the DMA of piece k
static int staged_h2d(alz_ctx* c, void* d_dst, const uint8_t* h_src, size_t bytes) {
    if (!bytes) return ALZ_OK;
    if (bytes < (1u << 20) || ensure_pinned(c) != ALZ_OK) {
        HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
        return ALZ_OK;
    }
    size_t done = 0; int k = 0, rc;
    while (done < bytes) {
        const size_t n = bytes - done < c->pin_cap ? bytes - done : c->pin_cap;
        if ((rc = pin_wait(c, k))) return rc;
        c->copy((uint8_t*)c->pin[k], h_src + done, n);
        HIP_TRY(hipMemcpyAsync((uint8_t*)d_dst + done, c->pin[k], n, hipMemcpyHostToDevice, c->stream));
        if ((rc = pin_mark(c, k))) return rc;
        done += n; k ^= 1;
    }
    return ALZ_OK;
}
// device -> host (pageable), the same way: the DMA of piece k + 1 runs while piece k is copied out of its pinned buffer
static int staged_d2h(alz_ctx* c, uint8_t* h_dst, const void* d_src, size_t bytes) {
    if (!bytes) return ALZ_OK;
    if (bytes < (1u << 20) || ensure_pinned(c) != ALZ_OK) {
        HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return ALZ_OK;
    }
    size_t issued = 0, copied = 0; int k = 0, rc;
    size_t len[2] = {0, 0};
    {   const size_t n = bytes < c->pin_cap ? bytes : c->pin_cap;
        HIP_TRY(hipMemcpyAsync(c->pin[0], d_src, n, hipMemcpyDeviceToHost, c->stream));   // (stream order keeps it behind an earlier DMA out of pin[0])
        if ((rc = pin_mark(c, 0))) return rc;
        len[0] = n; issued = n; }
    while (copied < bytes) {
        if (issued < bytes) {
            const size_t n = bytes - issued < c->pin_cap ? bytes - issued : c->pin_cap;
            HIP_TRY(hipMemcpyAsync(c->pin[k ^ 1], (const uint8_t*)d_src + issued, n, hipMemcpyDeviceToHost, c->stream));
            if ((rc = pin_mark(c, k ^ 1))) return rc;
            len[k ^ 1] = n; issued += n;
        }
        if ((rc = pin_wait(c, k))) return rc;
        c->copy(h_dst + copied, (const uint8_t*)c->pin[k], len[k]);
        copied += len[k]; k ^= 1;
    }
    return ALZ_OK;
}

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

// one produced range: `len` bytes at device offset `dev` (relative to ctx->d_dst) that belong at `host`
struct out_seg { uint64_t dev; uint8_t* host; uint32_t len; };

// false = could not pack (no device memory for the dense copy): the caller falls back to one copy per stream
// Segments sorted by `dev` (offsets relative to d_base), pinned buffers present: windows of the device range go through the two
// pinned buffers -- the DMA of window w + 1 overlaps the copy-out of window w, which the context's copy threads spread.  Only
// the segments' own bytes are written on the host side.
static int download_windows(alz_ctx* c, const void* d_base, const std::vector<out_seg>& segs) {
    uint64_t lo = segs.front().dev, hi = 0;
    for (const out_seg& g : segs) if (g.dev + g.len > hi) hi = g.dev + g.len;
    const uint64_t W = c->pin_cap;
    const uint64_t nwin = (hi - lo + W - 1) / W;
    size_t first = 0;                                        // first segment that may still overlap the current window
    auto issue = [&](uint64_t w, int k) -> hipError_t {
        const uint64_t a = lo + w * W, n = hi - a < W ? hi - a : W;
        hipError_t e = hipMemcpyAsync(c->pin[k], (const uint8_t*)d_base + a, n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->pin_ev[k], c->stream);
        if (e == hipSuccess) c->pin_busy[k] = true;
        return e;
    };
    HIP_TRY(issue(0, 0));
    for (uint64_t w = 0; w < nwin; w++) {
        const int k = (int)(w & 1);
        if (w + 1 < nwin) HIP_TRY(issue(w + 1, k ^ 1));
        { const int rcw = pin_wait(c, k); if (rcw) return rcw; }
        const uint64_t a = lo + w * W, b = a + W;
        while (first < segs.size() && segs[first].dev + segs[first].len <= a) first++;
        c->jobs.clear();
        for (size_t i = first; i < segs.size() && segs[i].dev < b; i++) {
            const uint64_t s0 = segs[i].dev > a ? segs[i].dev : a, s1 = segs[i].dev + segs[i].len < b ? segs[i].dev + segs[i].len : b;
            if (s1 > s0) add_copy(c->jobs, segs[i].host + (s0 - segs[i].dev), (const uint8_t*)c->pin[k] + (s0 - a), s1 - s0);
        }
        c->pool->run(c->jobs);
    }
    return ALZ_OK;
}

// false = could not pack (no device memory for the dense copy, no pinned memory): the caller falls back to one copy per stream
static bool download_packed(alz_ctx* c, const std::vector<out_seg>& segs) {
    if (ensure_pinned(c) != ALZ_OK) { (void)hipGetLastError(); return false; }
    std::vector<pack_item> items;
    std::vector<out_seg> packed;                             // the same segments at their offsets in the dense copy
    items.reserve(segs.size()); packed.reserve(segs.size());
    uint64_t cur = 0;
    for (const out_seg& g : segs) {
        const uint64_t to = cur + (g.dev & 15);
        items.push_back(pack_item{g.dev, to, g.len, 0});
        packed.push_back(out_seg{to, g.host, g.len});
        cur = (to + g.len + 15) & ~15ull;
    }
    // (scratch of the context, grown on demand: no hipMalloc / hipFree per call)
    if (grow(c, &c->d_items, &c->d_items_cap, items.size() * sizeof(pack_item)) != ALZ_OK) { (void)hipGetLastError(); return false; }
    if (grow(c, &c->d_pack, &c->d_pack_cap, cur + 16) != ALZ_OK) { (void)hipGetLastError(); return false; }
    hipError_t e = hipMemcpyAsync(c->d_items, items.data(), items.size() * sizeof(pack_item), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        alz_pack_outputs_kernel<<<dim3((uint32_t)items.size()), dim3(256), 0, c->stream>>>((const uint8_t*)c->d_dst, (uint8_t*)c->d_pack, (const pack_item*)c->d_items);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    // (items is read by the asynchronous upload above: the first window's event wait below orders the two)
    if (download_windows(c, c->d_pack, packed) != ALZ_OK) { (void)hipGetLastError(); return false; }
    return true;
}

// Download the produced ranges: through the pinned staging buffers window by window when they are reasonably dense (the
// caller's bytes between streams are never touched), else packed on the device first.
static int download_segs(alz_ctx* c, std::vector<out_seg>& segs) {
    if (segs.empty()) return ALZ_OK;
    std::sort(segs.begin(), segs.end(), [](const out_seg& a, const out_seg& b) { return a.dev < b.dev; });
    uint64_t lo = segs.front().dev, hi = 0, sum = 0;
    for (const out_seg& g : segs) { if (g.dev + g.len > hi) hi = g.dev + g.len; sum += g.len; }
    if (hi - lo > 2 * sum + (64ull << 20)) {
        if (segs.size() >= 16 && download_packed(c, segs)) return ALZ_OK;
        for (const out_seg& g : segs) HIP_TRY(hipMemcpyAsync(g.host, (const uint8_t*)c->d_dst + g.dev, g.len, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return ALZ_OK;
    }
    if (ensure_pinned(c) != ALZ_OK) {                        // no pinned memory: one bounce buffer
        std::vector<uint8_t> bounce(hi - lo);
        HIP_TRY(hipMemcpyAsync(bounce.data(), (const uint8_t*)c->d_dst + lo, hi - lo, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (const out_seg& g : segs) memcpy(g.host, bounce.data() + (g.dev - lo), g.len);
        return ALZ_OK;
    }
    return download_windows(c, c->d_dst, segs);
}

static int download_outputs(alz_ctx* c, uint32_t n, const alz_stream* streams, const alz_result* results, uint8_t* dst_base, bool only_ok) {
    std::vector<out_seg> segs;
    segs.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (!results[i].dst_len || (only_ok && results[i].status != ALZ_ST_OK)) continue;
        segs.push_back(out_seg{streams[i].dst_off, dst_base + streams[i].dst_off, results[i].dst_len});
    }
    return download_segs(c, segs);
}

int alz_decode_batch(alz_ctx* c, const alz_lz_properties* props, uint32_t n, const uint8_t* src_base, size_t src_bytes,
                     const alz_stream* streams, uint8_t* dst_base, size_t dst_bytes, alz_result* results) {
    if (!c || (n && (!streams || !results))) return fail(ALZ_E_INVALID, "alz_decode_batch: bad argument");
    for (uint32_t i = 0; i < n; i++) {
        if (!range_ok(streams[i].src_off, streams[i].src_len, src_bytes)) return fail(ALZ_E_INVALID, "stream %u: source range exceeds src_bytes", i);
        if (!range_ok(streams[i].dst_off, streams[i].dst_cap, dst_bytes)) return fail(ALZ_E_INVALID, "stream %u: destination range exceeds dst_bytes", i);
        if (streams[i].format == ALZ_FMT_LZ4_BLOCK && streams[i].aux0 > streams[i].dst_off)   // (as alz_decode_batch_multi: the history lies in front of the block's output)
            return fail(ALZ_E_INVALID, "stream %u: LZ4 history (aux0) reaches in front of the destination buffer", i);
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = grow(c, &c->d_src, &c->d_src_cap, src_bytes + 64))) return rc;
    if ((rc = grow(c, &c->d_dst, &c->d_dst_cap, dst_bytes + 64))) return rc;
    if ((rc = staged_h2d(c, c->d_src, src_base, src_bytes))) return rc;
    for (uint32_t i = 0; i < n; i++)   // LZ4 blocks that continue a frame's window: their history has to be in HBM too
        if (streams[i].format == ALZ_FMT_LZ4_BLOCK && streams[i].aux0 && streams[i].aux0 <= streams[i].dst_off)
            HIP_TRY(hipMemcpyAsync((uint8_t*)c->d_dst + streams[i].dst_off - streams[i].aux0, dst_base + streams[i].dst_off - streams[i].aux0,
                                   streams[i].aux0, hipMemcpyHostToDevice, c->stream));
    alz_plan* p = nullptr;
    if ((rc = plan_create(c, props, n, streams, &p, true))) return rc;
    rc = alz_plan_execute(c, p, c->d_src, c->d_dst, nullptr);
    // ONE stream that states its size (what a format class's Decompress(Stream, Stream) hands over): the output travels with the result --
    // one wait instead of two (0.254 against 0.261 ms per call, A/B in one run) -- and what the stream really produced is copied out of the staging buffer
    const uint32_t spec = (n == 1 && streams[0].decom_len) ? (streams[0].decom_len < streams[0].dst_cap ? streams[0].decom_len : streams[0].dst_cap) : 0u;
    if (!rc && spec && spec <= kPinBytes && ensure_pinned(c) == ALZ_OK) {
        if ((rc = pin_wait(c, 0)) == ALZ_OK) {
            hipError_t e = hipMemcpyAsync(results, p->d_results, sizeof(alz_result), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(c->pin[0], (const uint8_t*)c->d_dst + streams[0].dst_off, spec, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) rc = fail(ALZ_E_HIP, "download failed: %s", hipGetErrorString(e));
        }
        if (!rc) {
            if (results[0].dst_len <= spec) { if (results[0].dst_len) memcpy(dst_base + streams[0].dst_off, c->pin[0], results[0].dst_len); }
            else rc = download_outputs(c, n, streams, results, dst_base, false);         // (an overshoot of the declared size: E4)
        }
        if (rc) (void)hipStreamSynchronize(c->stream);       // (an error path: the plan's pageable upload sources may still be queued)
        alz_plan_destroy(c, p);
        return rc;
    }
    if (!rc) rc = alz_plan_results(c, p, results);
    if (!rc) rc = download_outputs(c, n, streams, results, dst_base, false);   // copy back only what each stream produced
    if (rc) (void)hipStreamSynchronize(c->stream);
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
    alz_ctx* c; int slot = 0;
    explicit EncScratch(alz_ctx* ctx) : c(ctx) {}
    // slots are handed out in call order; `want` = false skips one (its buffer, if any, stays for a later call)
    hipError_t alloc(void** p, size_t bytes, bool want = true) {
        const int k = slot++;
        if (!want) { *p = nullptr; return hipSuccess; }
        if (bytes < 16) bytes = 16;
        if (c->enc_cap[k] < bytes) {
            if (c->enc_buf[k]) { (void)hipFree(c->enc_buf[k]); c->enc_buf[k] = nullptr; c->enc_cap[k] = 0; }
            hipError_t e = hipMalloc(&c->enc_buf[k], bytes);
            if (e != hipSuccess) { c->enc_buf[k] = nullptr; return e; }
            c->enc_cap[k] = bytes;
        }
        *p = c->enc_buf[k];
        return hipSuccess;
    }
};
// (alz_encode_seg.h) 1: a small batch of this format is walked speculatively per segment (alz_encode_seg_seq.h), not from synchronisation points
extern "C++" int alz_encode_seg_spec_format(int fmt);
static void release_scratch(alz_ctx* c) {
    for (int k = 0; k < 15; k++) { if (c->enc_buf[k]) (void)hipFree(c->enc_buf[k]); c->enc_buf[k] = nullptr; c->enc_cap[k] = 0; }
    void** bufs[] = {&c->d_src, &c->d_dst, &c->d_items, &c->d_pack, &c->d_plan, &c->d_bigbuf};
    size_t* caps[] = {&c->d_src_cap, &c->d_dst_cap, &c->d_items_cap, &c->d_pack_cap, &c->d_plan_cap, &c->d_bigbuf_cap};
    for (int i = 0; i < 6; i++) { if (*bufs[i]) (void)hipFree(*bufs[i]); *bufs[i] = nullptr; *caps[i] = 0; }
}
int alz_ctx_release_scratch(alz_ctx* c) {
    if (!c) return fail(ALZ_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    release_scratch(c);
    return ALZ_OK;
}

// FastLZ.CompressHeaderless picks level 2 per source  FastLZ.cs:169-175
static inline bool fastlz_level2(const alz_settings& st, uint32_t src_len) { return src_len >= 0x10000u && st.quality > 4 && st.max_window_bits > 13; }

// The encode of one batch with everything on the device: `streams` (a host table) holds offsets relative to d_src / d_dst, which
// the caller has filled / will read.  `upload(d_src, d_dst)` runs once the arguments are validated and the scratch exists -- the
// host-buffer entry points grow their device buffers and stage their input there (a refused batch never touches the device);
// the device-resident entry point just hands its two pointers over.
// `src_has_slack`: the source buffer has >= 64 readable bytes behind src_bytes (the library's own staging buffer).  A caller's device
// buffer need not: the finder's look-ahead loads read up to 32 bytes past a stream's end, so the streams that end inside the last 64
// bytes of the caller's buffer are copied into scratch with room behind them and searched there (their descriptors -- the device copy
// only -- point at the copy; offsets are differences of device addresses).
static int encode_core(alz_ctx* c, const alz_lz_properties* props, const alz_settings* settings, uint32_t n, size_t src_bytes, const alz_stream* streams,
                       size_t dst_bytes, alz_result* results, alz_encode_aux* aux, const std::function<int(const void*&, void*&)>& upload,
                       bool src_has_slack = true, bool no_big = false) {
    const void* d_src_base = nullptr; void* d_dst_base = nullptr;
    alz_settings st; if (settings) st = *settings; else { st.quality = 8; st.max_window_bits = 0; st.strategy = 0; st.min_distance = 0; }
    if (st.quality < 0 || st.quality > 15) return fail(ALZ_E_INVALID, "quality %d outside 0..15 (CompressionSettings.cs:38-50)", st.quality);
    if (st.max_window_bits < 0 || st.max_window_bits > 24) return fail(ALZ_E_INVALID, "max_window_bits %d outside 0..24", st.max_window_bits);
    alz_lz_properties lz = effective_lz(props);
    phase_timer tm("alz_encode_batch");
    std::vector<uint32_t> cnt(ALZ_FMT_COUNT, 0);
    std::vector<uint64_t> pos_off(n);
    uint64_t total = 0; uint32_t max_len = 0, n_fastlz2 = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (streams[i].format >= ALZ_FMT_COUNT) return fail(ALZ_E_INVALID, "stream %u: unknown format %u", i, streams[i].format);
        if (!range_ok(streams[i].src_off, streams[i].src_len, src_bytes)) return fail(ALZ_E_INVALID, "stream %u: source range exceeds src_bytes", i);
        if (!range_ok(streams[i].dst_off, streams[i].dst_cap, dst_bytes)) return fail(ALZ_E_INVALID, "stream %u: destination range exceeds dst_bytes", i);
        if (streams[i].src_len > 0x7FFFFF00u) return fail(ALZ_E_UNSUPPORTED, "stream %u: inputs above 2 GiB are not supported", i);
        cnt[streams[i].format]++;
        if (streams[i].format == ALZ_FMT_FASTLZ && fastlz_level2(st, streams[i].src_len)) n_fastlz2++;
        pos_off[i] = total; total += ((uint64_t)streams[i].src_len + 16 + 63) & ~63ull;   // 64-aligned: one start-mask word per 64 positions
        if (streams[i].src_len > max_len) max_len = streams[i].src_len;
    }
    if (cnt[ALZ_FMT_LZSS] && (lz.window_bits < 8 || lz.window_bits > 16 || lz.length_bits < 1 || lz.length_bits > 8 || lz.max_distance != (1u << lz.window_bits)))
        return fail(ALZ_E_UNSUPPORTED, "LZSS geometry outside the GPU path");
    std::vector<unsigned char> geom((ALZ_FMT_COUNT + 1) * alz_encode_geom_size());   // last slot: FastLZ level 2
    bool any_min = false, any_match = false, any_mask = false, any_narrow = false;
    std::vector<uint32_t> seg_len(ALZ_FMT_COUNT + 1, 0), seg_kmax(ALZ_FMT_COUNT + 1, 0); size_t seg_bytes = 0;
    for (int f = 0; f <= ALZ_FMT_COUNT; f++) {
        const bool lvl2 = f == ALZ_FMT_COUNT;
        if (lvl2 ? !n_fastlz2 : !(cnt[f] - (f == ALZ_FMT_FASTLZ ? n_fastlz2 : 0u))) continue;
        void* g = geom.data() + f * alz_encode_geom_size();
        alz_settings sf = st;
        if (st.max_window_bits != 0 && f != ALZ_FMT_FASTLZ && !lvl2) {
            // CompressionSettings.MaxWindowBits can only WIDEN the finder (windowsBits = max(.., maxWindowBits), maxDistance = max(.., 1 <<
            // maxWindowBits)  MatchFinder/LzChainMatchFinder.cs:69-73): a value within the format's own window changes nothing, and the
            // batch is encoded as with 0.  A larger one lets the managed finder return distances the format cannot store (FastLZ, whose
            // level 2 is selected by it, aside): the caller's own encoder.
            sf.max_window_bits = 0;
            int wb0 = 0;
            if (!alz_encode_geometry(f, &lz, &sf, g, &wb0, 0)) return fail(ALZ_E_UNSUPPORTED, "format %d: geometry not supported by the GPU encoder", f);
            if (st.max_window_bits > wb0 || (1ll << st.max_window_bits) > (long long)alz_encode_geom_max_dist(g))
                return fail(ALZ_E_UNSUPPORTED, "format %d: CompressionSettings.MaxWindowBits %d is beyond the format's window (%d bits, distances up to %d): "
                            "the managed finder would return distances the format cannot store", f, st.max_window_bits, wb0, alz_encode_geom_max_dist(g));
        }
        if (!alz_encode_geometry(lvl2 ? ALZ_FMT_FASTLZ : f, &lz, &sf, g, nullptr, lvl2 ? 1 : 0))
            return fail(ALZ_E_UNSUPPORTED, "format %d: geometry not supported by the GPU encoder", lvl2 ? ALZ_FMT_FASTLZ : f);
        any_min = any_min || alz_encode_geom_min_table(g);
        any_narrow = any_narrow || alz_encode_geom_narrows(g);
        any_match = any_match || alz_encode_geom_needs_match(lvl2 ? ALZ_FMT_FASTLZ : f, g);
        any_mask = any_mask || alz_encode_format_needs_mask(lvl2 ? ALZ_FMT_FASTLZ : f);
        // a batch of few buffers of a flag-bit format: parse and emitter over segments (alz_encode_seg.h) -- behind kernel B and the roles walk
        uint32_t seg_hist = 0;
        if (!lvl2 && !c->exact && c->variant == 0 && alz_encode_segmented(f, g, cnt[f], max_len, c->seg_max_streams, &seg_len[f], &seg_kmax[f], &seg_hist)) {
            any_match = any_mask = true;
            size_t ab = 0; (void)alz_encode_aseg(g, cnt[f], max_len, nullptr, nullptr, nullptr, nullptr, &ab);
            const size_t b = ((alz_encode_seg_bytes(cnt[f], seg_kmax[f], seg_hist) + 255) & ~(size_t)255) + ab; if (b > seg_bytes) seg_bytes = b;
        }
    }
    // ---- a handful of big streams: each of them on the whole GPU (alz_encode_big.h).  A stream the path declines (too many positions that
    // need an exact second search) sends the whole call through the batch pipeline below.
    // As in the decoder (plan_create): one after the other on the whole GPU while that beats side by side with a workgroup + a wavefront
    // each -- measured at quality 0-8 on Test.bmp (tools/mid_batch_encode.py): ~0.10 ms + 0.10 ms per MiB on the whole GPU, ~22 ms per MiB of the LONGEST buffer side by side.
    if (!no_big && n <= ALZ_BIG_MAX_STREAMS && c->big_min != 0xFFFFFFFFu && !c->exact && c->variant == 0) {
        // (from 8 KiB on -- not the decoder's 24 KiB: ONE buffer of 8 KiB takes 0.08 ms of kernels this way and 0.22 through the batch pipeline, of
        // 64 KiB 0.10 against 1.6, tools/single_encode_sizes.py; a threshold the caller has set below that is the caller's)
        const uint32_t enc_min = c->big_min < 8192u ? c->big_min : 8192u;
        bool all = true; size_t sb = 0;
        double t_big = 0, t_side = 0;
        const bool q0 = st.quality == 0;                                      // (no narrowing behind kernel A, one candidate per position in kernel B: 8 x 1 MB 0.99 one by one / 0.68)
        for (uint32_t i = 0; all && i < n; i++) {
            const void* g = geom.data() + streams[i].format * alz_encode_geom_size();
            all = streams[i].format != ALZ_FMT_FASTLZ && alz_encode_big_eligible((int)streams[i].format, g, &streams[i], enc_min);
            if (all) { const size_t b = alz_encode_big_scratch_bytes((int)streams[i].format, g, &streams[i]); if (b > sb) sb = b; }
            const double mib = streams[i].src_len / 1048576.0;
            // (round 6, 2-32 x 1 MiB of Test.bmp one by one, ms per MiB: LZ4 blocks / LZO -- 64 KiB windows -- 0.32 at quality 0 and 0.80 at quality 8, LZ11 / LZ40 0.21 / 0.37)
            const bool wide = alz_encode_geom_max_dist(g) > 8192, l11 = streams[i].format == ALZ_FMT_LZ11 || streams[i].format == ALZ_FMT_LZ40;
            const double rate = wide ? (q0 ? 0.22 : 0.70) : (l11 && !q0) ? 0.27 : 0.10;
            t_big += 0.10 + rate * mib; if (22.0 * mib > t_side) t_side = 22.0 * mib;
        }
        // (side by side is no longer one wavefront per buffer where the segmented parse + emit takes the launch, alz_encode_seg.h: what is left of the longest
        // buffer's serial time is kernel A -- over segments too, for these few buffers: ~0.9 ms per MiB; one workgroup per buffer where that does not apply: 2.2 --
        // + ~0.14 ms of small kernels per format + ~0.08 ms per MiB of the whole call.  tools/mid_batch_encode.py with ALZ_MID_BIG=on / off,
        // profiles/r05_mid_big_vs_seg.txt: 4 x 256 KiB at quality 8 0.47 ms one by one against 0.42, 8 x 1 MB 1.66 / 1.73, 16 x 1 MB 3.33 / 2.18)
        {
            bool seg_all = true, aseg_all = true, spec_any = false, wide_any = false; int nf = 0; double mib_all = 0;
            for (int f = 0; f < ALZ_FMT_COUNT; f++) if (cnt[f]) {
                nf++; seg_all = seg_all && seg_len[f] != 0; spec_any = spec_any || alz_encode_seg_spec_format(f);
                wide_any = wide_any || alz_encode_geom_max_dist(geom.data() + f * alz_encode_geom_size()) > 8192;
                aseg_all = aseg_all && alz_encode_aseg(geom.data() + f * alz_encode_geom_size(), cnt[f], max_len, nullptr, nullptr, nullptr, nullptr, nullptr);
            }
            for (uint32_t i = 0; i < n; i++) mib_all += streams[i].src_len / 1048576.0;
            // (the formats of the speculative walk -- alz_encode_seg_seq.h -- have one serial step per segment of the longest buffer behind that, and the 64 KiB windows above quality 0 their
            // words / narrowing passes: ~0.4 / 2.0 ms per MiB; 2 / 4 / 8 / 16 x 1 MiB at the end of round 6, ms, one by one | side by side: LZ4 blocks at quality 0 0.65 / 1.28 / 2.53 / 5.09 |
            // 0.99 / 1.03 / 1.11 / 1.60, at quality 8 1.61 / 3.19 / 6.39 / 12.8 | 2.93 / 3.06 / 3.43 / 4.63; LZ11 at quality 0 0.42 / 0.84 / 1.64 / 3.27 | 0.72 / 0.79 / 0.90 / 1.42, at quality 8
            // 0.76 / 1.47 / 2.92 / 5.89 | 1.65 / 1.84 / 2.07 / 2.95)
            const double t_spec = spec_any ? ((wide_any && !q0) ? 2.0 : 0.4) : 0.0;
            if (seg_all) { const double t_seg = 0.14 * nf + ((aseg_all ? (q0 ? 0.45 : 0.9) : 2.2) + t_spec) * (max_len / 1048576.0) + (q0 ? 0.03 : 0.08) * mib_all; if (t_seg < t_side) t_side = t_seg; }
        }
        all = all && t_big < t_side;
        if (all) {
            HIP_TRY(hipSetDevice(c->device));
            EncScratch sc(c);
            void *d_tail = nullptr, *d_big = nullptr, *skip = nullptr;
            size_t tail_bytes = 0;
            if (!src_has_slack)
                for (uint32_t i = 0; i < n; i++)
                    if (streams[i].src_off + streams[i].src_len + 64 > src_bytes) tail_bytes += ((size_t)streams[i].src_len + 64 + 63) & ~(size_t)63;
            // what travels back: per stream its result, its section offsets and the path's 16 control words -- one block, one copy
            const size_t out_one = sizeof(alz_result) + sizeof(alz_encode_aux) + 16 * sizeof(uint32_t), sb_al = (sb + 255) & ~(size_t)255;
            hipError_t e = hipSuccess;
            for (int k = 0; k < 10 && e == hipSuccess; k++) e = sc.alloc(&skip, 0, false);   // (the batch pipeline's slots, in its order: the buffers are shared)
            if (e == hipSuccess) e = sc.alloc(&d_tail, tail_bytes, tail_bytes != 0);
            if (e == hipSuccess) e = sc.alloc(&d_big, sb_al + (size_t)n * out_one + 64);
            if (e != hipSuccess) return fail(ALZ_E_NOMEM, "encoder scratch allocation failed: %s", hipGetErrorString(e));
            uint8_t* d_out = (uint8_t*)d_big + sb_al;
            alz_result* d_results = (alz_result*)d_out;
            alz_encode_aux* d_aux = (alz_encode_aux*)(d_out + (size_t)n * sizeof(alz_result));
            uint32_t* d_ctl = (uint32_t*)(d_out + (size_t)n * (sizeof(alz_result) + sizeof(alz_encode_aux)));
            int rc;
            if ((rc = upload(d_src_base, d_dst_base))) return rc;
            tm.mark("upload");
            HIP_TRY(hipEventRecord(c->ev0, c->stream));
            size_t toff = 0;
            for (uint32_t i = 0; i < n; i++) {
                alz_stream st1 = streams[i];
                if (!src_has_slack && st1.src_off + st1.src_len + 64 > src_bytes) {      // its look-ahead would leave the caller's buffer
                    uint8_t* to = (uint8_t*)d_tail + toff;
                    HIP_TRY(hipMemcpyAsync(to, (const uint8_t*)d_src_base + st1.src_off, st1.src_len, hipMemcpyDeviceToDevice, c->stream));
                    st1.src_off = (uint64_t)(uintptr_t)to - (uint64_t)(uintptr_t)d_src_base;
                    toff += ((size_t)st1.src_len + 64 + 63) & ~(size_t)63;
                }
                e = alz_launch_encode_big((int)st1.format, c->stream, d_src_base, d_dst_base, &st1, d_results + i, d_aux + i, d_big, d_ctl + 16 * (size_t)i,
                                          geom.data() + st1.format * alz_encode_geom_size());
                if (e != hipSuccess) return fail(ALZ_E_HIP, "big-stream encode launch (format %u) failed: %s", st1.format, hipGetErrorString(e));
            }
            HIP_TRY(hipEventRecord(c->ev1, c->stream));
            std::vector<uint8_t> hout((size_t)n * out_one);
            HIP_TRY(hipMemcpyAsync(hout.data(), d_out, hout.size(), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            { float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_kernel_ms = ms; }
            tm.mark("kernels (big streams)");
            const alz_encode_aux* haux = (const alz_encode_aux*)(hout.data() + (size_t)n * sizeof(alz_result));
            const uint32_t* hctl = (const uint32_t*)(hout.data() + (size_t)n * (sizeof(alz_result) + sizeof(alz_encode_aux)));
            bool any = false;
            for (uint32_t i = 0; i < n; i++) any = any || hctl[16 * (size_t)i] != 0u;
            if (!any) {
                c->big_enc_launches += n;
                memcpy(results, hout.data(), (size_t)n * sizeof(alz_result));
                for (uint32_t i = 0; i < n; i++) if (aux) aux[i] = haux[i];
                return ALZ_OK;
            }
            return encode_core(c, props, settings, n, src_bytes, streams, dst_bytes, results, aux, upload, src_has_slack, true);
        }
    }
    // streams whose look-ahead would leave the caller's source buffer (see above)
    std::vector<uint32_t> tail_ix; size_t tail_bytes = 0;
    if (!src_has_slack)
        for (uint32_t i = 0; i < n; i++)
            if (streams[i].src_off + streams[i].src_len + 64 > src_bytes) { tail_ix.push_back(i); tail_bytes += ((size_t)streams[i].src_len + 64 + 63) & ~(size_t)63; }
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    // (kernel A keeps its hash table in LDS: no per-stream tables in HBM, nothing that bounds the streams of a launch but the grid:
    // a launch carries the stream in gridDim.y; 70 000 went through on this runtime, the cap is caution)
    const uint32_t CH = 65535u;
    EncScratch sc(c);
    alz_stream* d_streams = nullptr; alz_result* d_results = nullptr; alz_encode_aux* d_aux = nullptr; uint32_t* d_index = nullptr;
    uint64_t* d_pos = nullptr; int *d_prev4 = nullptr, *d_prevm = nullptr; void *d_match = nullptr, *d_side = nullptr, *d_mask = nullptr;
    hipError_t e = sc.alloc((void**)&d_streams, (size_t)n * sizeof(alz_stream));
    if (e == hipSuccess) e = sc.alloc((void**)&d_results, (size_t)n * sizeof(alz_result));
    if (e == hipSuccess) e = sc.alloc((void**)&d_aux, (size_t)n * sizeof(alz_encode_aux));
    if (e == hipSuccess) e = sc.alloc((void**)&d_index, (size_t)n * sizeof(uint32_t));
    if (e == hipSuccess) e = sc.alloc((void**)&d_pos, (size_t)n * sizeof(uint64_t));
    if (e == hipSuccess) e = sc.alloc((void**)&d_prev4, (size_t)total * sizeof(int) + 256);   // (+ slack: the look-ahead of the fused parse kernel reads a link of an empty last stream)
    if (e == hipSuccess) e = sc.alloc((void**)&d_prevm, (size_t)total * sizeof(int) + 256, any_min);   // the min-length table's links
    if (e == hipSuccess) e = sc.alloc(&d_match, (size_t)total * 4 + 64, any_match);   // one 32-bit entry per position (alz_encode.hip: mentry); not when every launch searches inside its parse + emit kernel
    if (e == hipSuccess) e = sc.alloc(&d_side, (size_t)total * 2 + 64, cnt[ALZ_FMT_YAY0] || cnt[ALZ_FMT_MIO0] || cnt[ALZ_FMT_SMSR00]);   // section buffers
    if (e == hipSuccess) e = sc.alloc(&d_mask, (size_t)total / 8 + 64, any_mask);          // a bit per position: the start mask of enc_roles_kernel, for the formats whose emitter is a kernel of its own
    void* d_tail = nullptr; uint32_t* d_sel = nullptr;
    if (e == hipSuccess) e = sc.alloc(&d_tail, tail_bytes, !tail_ix.empty());
    { void* skip_big = nullptr; if (e == hipSuccess) e = sc.alloc(&skip_big, 0, false); }        // (slot 11: the whole-GPU path's scratch)
    if (e == hipSuccess) e = sc.alloc((void**)&d_sel, ((size_t)4 * n + 128) * sizeof(uint32_t), any_match);     // which kernel B per stream (enc_probe_kernel); behind it the two lists of enc_scan_select_kernel
    int* d_narrow = nullptr;
    if (e == hipSuccess) e = sc.alloc((void**)&d_narrow, (size_t)total * sizeof(int) + 256, any_narrow);   // slot 13: the links of the finder's own hash width, narrowed from 15-bit ones (enc_narrow_kernel)
    void* d_seg = nullptr;
    if (e == hipSuccess) e = sc.alloc(&d_seg, seg_bytes, seg_bytes != 0);                                  // slot 14: segment records of alz_encode_seg.h
    if (e != hipSuccess) return fail(ALZ_E_NOMEM, "encoder scratch allocation failed: %s", hipGetErrorString(e));
    tm.mark("validate + allocate");
    std::vector<uint32_t> index(n), foff(ALZ_FMT_COUNT, 0), fill(ALZ_FMT_COUNT, 0);
    { uint32_t off = 0; for (int f = 0; f < ALZ_FMT_COUNT; f++) { foff[f] = off; off += cnt[f]; } }
    for (uint32_t i = 0; i < n; i++) { uint32_t f = streams[i].format; index[foff[f] + fill[f]++] = i; }
    if (n_fastlz2)                                           // FastLZ: the level-1 streams first, then the level-2 ones (own geometry, own launch)
        std::stable_partition(index.begin() + foff[ALZ_FMT_FASTLZ], index.begin() + foff[ALZ_FMT_FASTLZ] + cnt[ALZ_FMT_FASTLZ],
                              [&](uint32_t i) { return !fastlz_level2(st, streams[i].src_len); });
    if ((rc = upload(d_src_base, d_dst_base))) return rc;
    tm.mark("upload");
    std::vector<alz_stream> moved;                               // (stays alive until the synchronise below: the copy is asynchronous)
    if (!tail_ix.empty()) {
        moved.assign(streams, streams + n);
        size_t off = 0;
        for (uint32_t i : tail_ix) {
            uint8_t* to = (uint8_t*)d_tail + off;
            if (streams[i].src_len) HIP_TRY(hipMemcpyAsync(to, (const uint8_t*)d_src_base + streams[i].src_off, streams[i].src_len, hipMemcpyDeviceToDevice, c->stream));
            moved[i].src_off = (uint64_t)(uintptr_t)to - (uint64_t)(uintptr_t)d_src_base;
            off += ((size_t)streams[i].src_len + 64 + 63) & ~(size_t)63;
        }
    }
    HIP_TRY(hipMemcpyAsync(d_streams, moved.empty() ? streams : moved.data(), (size_t)n * sizeof(alz_stream), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_index, index.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_pos, pos_off.data(), (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_results, 0xFF, (size_t)n * sizeof(alz_result), c->stream));
    HIP_TRY(hipMemsetAsync(d_aux, 0, (size_t)n * sizeof(alz_encode_aux), c->stream));
    if (any_mask) HIP_TRY(hipMemsetAsync(d_mask, 0, (size_t)total / 8 + 64, c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    const alz_encode_side side_q = { c->aux[0], c->fork, c->join[1] };      // (the scan streams' kernel runs beside the others: alz_launch_encode)
    for (int f = 0; f <= ALZ_FMT_COUNT; f++) {
        const bool lvl2 = f == ALZ_FMT_COUNT;
        const int fmt = lvl2 ? ALZ_FMT_FASTLZ : f;
        const uint32_t first = foff[fmt] + (lvl2 ? cnt[fmt] - n_fastlz2 : 0u);
        const uint32_t count = lvl2 ? n_fastlz2 : cnt[f] - (f == ALZ_FMT_FASTLZ ? n_fastlz2 : 0u);
        const void* g = geom.data() + f * alz_encode_geom_size();
        for (uint32_t done = 0; done < count; done += CH) {
            const uint32_t k = count - done < CH ? count - done : CH;
            e = alz_launch_encode(fmt, c->stream, d_src_base, d_dst_base, d_streams, d_index + first + done, k, max_len, d_prev4, d_prevm, d_narrow,
                                  d_match, d_pos, d_side, d_mask, d_results, d_aux, g, d_sel, n, seg_len[f] ? d_seg : nullptr, seg_len[f], seg_kmax[f], (c->exact || c->variant != 0) ? 2 : c->scan_mode, c->d_big_accepted ? c->d_big_accepted + 1 : nullptr, &side_q);
            if (e != hipSuccess) return fail(ALZ_E_HIP, "encode launch (format %d) failed: %s", fmt, hipGetErrorString(e));
            if (seg_len[f]) c->seg_enc_launches++;
        }
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipMemcpyAsync(results, d_results, (size_t)n * sizeof(alz_result), hipMemcpyDeviceToHost, c->stream));
    std::vector<alz_encode_aux> haux(n);
    HIP_TRY(hipMemcpyAsync(haux.data(), d_aux, (size_t)n * sizeof(alz_encode_aux), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    { float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_kernel_ms = ms; }   // table resets + the encode kernels
    tm.mark("kernels");
    for (uint32_t i = 0; i < n; i++) if (aux) aux[i] = haux[i];
    return ALZ_OK;
}

int alz_encode_batch(alz_ctx* c, const alz_lz_properties* props, const alz_settings* settings, uint32_t n, const uint8_t* src_base,
                     size_t src_bytes, const alz_stream* streams, uint8_t* dst_base, size_t dst_bytes, alz_result* results, alz_encode_aux* aux) {
    if (!c || (n && (!streams || !results || !dst_base))) return fail(ALZ_E_INVALID, "alz_encode_batch: bad argument");
    if (n == 0) return ALZ_OK;
    int rc = encode_core(c, props, settings, n, src_bytes, streams, dst_bytes, results, aux, [&](const void*& ds, void*& dd) {
        int r;
        if ((r = grow(c, &c->d_src, &c->d_src_cap, src_bytes + 64))) return r;
        if ((r = grow(c, &c->d_dst, &c->d_dst_cap, dst_bytes + 64))) return r;
        ds = c->d_src; dd = c->d_dst;
        return staged_h2d(c, c->d_src, src_base, src_bytes);
    });
    if (rc) return rc;
    return download_outputs(c, n, streams, results, dst_base, true);
}

// The same with the raw buffers already in HBM and the compressed streams left there (alz_stream offsets are relative to the two
// device pointers): what a caller that produces its input on the device uses, and what a benchmark times -- kernels, no PCIe.
int alz_encode_batch_device(alz_ctx* c, const alz_lz_properties* props, const alz_settings* settings, uint32_t n, const void* d_src_base,
                            size_t src_bytes, const alz_stream* streams, void* d_dst_base, size_t dst_bytes, alz_result* results, alz_encode_aux* aux) {
    if (!c || (n && (!streams || !results || !d_src_base || !d_dst_base))) return fail(ALZ_E_INVALID, "alz_encode_batch_device: bad argument");
    if (n == 0) return ALZ_OK;
    return encode_core(c, props, settings, n, src_bytes, streams, dst_bytes, results, aux,
                       [&](const void*& ds, void*& dd) { ds = d_src_base; dd = d_dst_base; return (int)ALZ_OK; }, false);
}

// ---------------------------------------------------------------- multi-GPU: one batch over several contexts
// Decode cost per output byte of a format relative to the fastest one (x64; from the measured per-format rates, docs/EXPERIMENTS.md 4.4)
static uint32_t format_weight(uint32_t fmt) {
    switch (fmt) {
    case ALZ_FMT_YAY0: return 55;
    case ALZ_FMT_YAZ0: case ALZ_FMT_LZ02: case ALZ_FMT_LZ11: case ALZ_FMT_LZ40: return 64;
    case ALZ_FMT_MIO0: case ALZ_FMT_SMSR00: return 78;
    case ALZ_FMT_LZSS: case ALZ_FMT_LZ10: case ALZ_FMT_CLZ0: case ALZ_FMT_CNS: case ALZ_FMT_LZHUDSON: case ALZ_FMT_LZSHREK: return 83;
    case ALZ_FMT_CNX2: case ALZ_FMT_HIG: case ALZ_FMT_PRS_BE: case ALZ_FMT_PRS_LE: case ALZ_FMT_LZ4_BLOCK: case ALZ_FMT_FASTLZ: case ALZ_FMT_WFLZ: case ALZ_FMT_WFLZ_BE: return 109;
    case ALZ_FMT_BLZ: case ALZ_FMT_REFPACK: case ALZ_FMT_SNAPPY_RAW: return 140;
    default: return 171;                                     // LZO
    }
}

int alz_partition_batch(uint32_t n, const alz_stream* streams, uint32_t n_parts, uint32_t* part_of, uint64_t* part_cost) {
    if ((n && (!streams || !part_of)) || n_parts == 0) return fail(ALZ_E_INVALID, "alz_partition_batch: bad argument");
    std::vector<uint64_t> cost(n), load(n_parts, 0);
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t f = streams[i].format < ALZ_FMT_COUNT ? streams[i].format : 0;
        const uint64_t out = streams[i].decom_len && streams[i].decom_len < streams[i].dst_cap ? streams[i].decom_len : streams[i].dst_cap;
        cost[i] = (out + 4096) * format_weight(f);           // (+ a fixed share: a wave per stream, header / tail handling)
        order[i] = i;
    }
    // greedy LPT: longest job first onto the least loaded part (equal jobs, as in the metric's batches, are dealt round-robin)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    for (uint32_t k = 0; k < n; k++) {
        uint32_t best = 0;
        for (uint32_t q = 1; q < n_parts; q++) if (load[q] < load[best]) best = q;
        part_of[order[k]] = best; load[best] += cost[order[k]];
    }
    if (part_cost) for (uint32_t q = 0; q < n_parts; q++) part_cost[q] = load[q];
    return ALZ_OK;
}

struct in_seg { uint64_t dev; const uint8_t* host; uint32_t len; };
// packed device range <- scattered host ranges (segments sorted by device offset), window by window through the pinned buffers
static int upload_segs(alz_ctx* c, void* d_base, const std::vector<in_seg>& segs, uint64_t total) {
    if (segs.empty() || total == 0) return ALZ_OK;
    if (ensure_pinned(c) != ALZ_OK) {
        for (const in_seg& g : segs) if (g.len) HIP_TRY(hipMemcpyAsync((uint8_t*)d_base + g.dev, g.host, g.len, hipMemcpyHostToDevice, c->stream));
        return ALZ_OK;
    }
    const uint64_t W = c->pin_cap;
    size_t first = 0; int rcp;
    for (uint64_t a = 0, w = 0; a < total; a += W, w++) {
        const int k = (int)(w & 1);
        const uint64_t b = a + W < total ? a + W : total;
        if ((rcp = pin_wait(c, k))) return rcp;
        while (first < segs.size() && segs[first].dev + segs[first].len <= a) first++;
        c->jobs.clear();
        for (size_t i = first; i < segs.size() && segs[i].dev < b; i++) {
            const uint64_t s0 = segs[i].dev > a ? segs[i].dev : a, s1 = segs[i].dev + segs[i].len < b ? segs[i].dev + segs[i].len : b;
            if (s1 > s0) add_copy(c->jobs, (uint8_t*)c->pin[k] + (s0 - a), segs[i].host + (s0 - segs[i].dev), s1 - s0);
        }
        c->pool->run(c->jobs);
        HIP_TRY(hipMemcpyAsync((uint8_t*)d_base + a, c->pin[k], b - a, hipMemcpyHostToDevice, c->stream));
        if ((rcp = pin_mark(c, k))) return rcp;
    }
    return ALZ_OK;
}

// One context's share of a batch: its streams are packed into device buffers of their own (16-byte aligned inputs, 256-byte
// aligned outputs), so every device receives and returns only the bytes of its share.
static int decode_share(alz_ctx* c, const alz_lz_properties* props, const std::vector<uint32_t>& idx, const uint8_t* src_base,
                        const alz_stream* streams, uint8_t* dst_base, alz_result* results) {
    const uint32_t m = (uint32_t)idx.size();
    if (m == 0) return ALZ_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<alz_stream> ds(m);
    std::vector<in_seg> ins; ins.reserve(m);
    uint64_t so = 0, dof = 0;
    for (uint32_t j = 0; j < m; j++) {
        const alz_stream& st = streams[idx[j]];
        const uint64_t hist = (st.format == ALZ_FMT_LZ4_BLOCK) ? st.aux0 : 0;
        ds[j] = st;
        ds[j].src_off = so; ins.push_back(in_seg{so, src_base + st.src_off, st.src_len});
        so += ((uint64_t)st.src_len + 15) & ~15ull;
        dof = (dof + hist + 255) & ~255ull;
        ds[j].dst_off = dof;
        dof = (dof + st.dst_cap + 255) & ~255ull;
    }
    int rc;
    if ((rc = grow(c, &c->d_src, &c->d_src_cap, so + 64))) return rc;
    if ((rc = grow(c, &c->d_dst, &c->d_dst_cap, dof + 64))) return rc;
    if ((rc = upload_segs(c, c->d_src, ins, so))) return rc;
    for (uint32_t j = 0; j < m; j++) {                       // LZ4 blocks that continue a frame's window: their history travels too
        const alz_stream& st = streams[idx[j]];
        if (st.format == ALZ_FMT_LZ4_BLOCK && st.aux0)
            HIP_TRY(hipMemcpyAsync((uint8_t*)c->d_dst + ds[j].dst_off - st.aux0, dst_base + st.dst_off - st.aux0, st.aux0, hipMemcpyHostToDevice, c->stream));
    }
    alz_plan* p = nullptr;
    if ((rc = plan_create(c, props, m, ds.data(), &p, true))) return rc;
    std::vector<alz_result> rs(m);
    rc = alz_plan_execute(c, p, c->d_src, c->d_dst, nullptr);
    if (!rc) rc = alz_plan_results(c, p, rs.data());
    if (rc) (void)hipStreamSynchronize(c->stream);           // (the stream table `ds` and the plan's index list are upload sources)
    alz_plan_destroy(c, p);
    if (rc) return rc;
    std::vector<out_seg> outs; outs.reserve(m);
    for (uint32_t j = 0; j < m; j++) {
        results[idx[j]] = rs[j];
        if (rs[j].dst_len) outs.push_back(out_seg{ds[j].dst_off, dst_base + streams[idx[j]].dst_off, rs[j].dst_len});
    }
    return download_segs(c, outs);
}

int alz_decode_batch_multi(alz_ctx* const* ctxs, uint32_t n_ctx, const alz_lz_properties* props, uint32_t n,
                           const uint8_t* src_base, size_t src_bytes, const alz_stream* streams,
                           uint8_t* dst_base, size_t dst_bytes, alz_result* results, uint32_t* part_of_out) {
    if (!ctxs || n_ctx == 0 || n_ctx > 64 || (n && (!streams || !results))) return fail(ALZ_E_INVALID, "alz_decode_batch_multi: bad argument");
    for (uint32_t q = 0; q < n_ctx; q++) {
        if (!ctxs[q]) return fail(ALZ_E_INVALID, "alz_decode_batch_multi: context %u is NULL", q);
        for (uint32_t r = 0; r < q; r++) if (ctxs[r] == ctxs[q]) return fail(ALZ_E_INVALID, "alz_decode_batch_multi: context %u is listed twice (a context is single-threaded)", q);
    }
    for (uint32_t i = 0; i < n; i++) {
        if (streams[i].format >= ALZ_FMT_COUNT) return fail(ALZ_E_INVALID, "stream %u: unknown format %u", i, streams[i].format);
        if (!range_ok(streams[i].src_off, streams[i].src_len, src_bytes)) return fail(ALZ_E_INVALID, "stream %u: source range exceeds src_bytes", i);
        if (!range_ok(streams[i].dst_off, streams[i].dst_cap, dst_bytes)) return fail(ALZ_E_INVALID, "stream %u: destination range exceeds dst_bytes", i);
        if (streams[i].format == ALZ_FMT_LZ4_BLOCK && streams[i].aux0 > streams[i].dst_off) return fail(ALZ_E_INVALID, "stream %u: history %u exceeds dst_off", i, streams[i].aux0);
    }
    std::vector<uint32_t> part(n ? n : 1);
    int rc = alz_partition_batch(n, streams, n_ctx, part.data(), nullptr);
    if (rc) return rc;
    if (part_of_out) for (uint32_t i = 0; i < n; i++) part_of_out[i] = part[i];
    std::vector<std::vector<uint32_t>> idx(n_ctx);
    for (uint32_t i = 0; i < n; i++) idx[part[i]].push_back(i);        // (ascending inside a share: source order)
    // one host thread per context (a context is single-threaded; distinct contexts run concurrently), no collective: the
    // shares are independent (a fresh LzWindows per stream, LZ10.cs:86)
    std::vector<int> rcs(n_ctx, ALZ_OK);
    std::vector<std::string> errs(n_ctx);
    auto work = [&](uint32_t q) {
        rcs[q] = decode_share(ctxs[q], props, idx[q], src_base, streams, dst_base, results);
        if (rcs[q]) errs[q] = g_err;
    };
    if (n_ctx == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (uint32_t q = 1; q < n_ctx; q++) th.emplace_back(work, q);
        work(0);
        for (std::thread& t : th) t.join();
    }
    for (uint32_t q = 0; q < n_ctx; q++) if (rcs[q]) return fail(rcs[q], "context %u (device %d): %s", q, ctxs[q]->device, errs[q].c_str());
    return ALZ_OK;
}

// ---------------------------------------------------------------- device-resident batches over several contexts
struct alz_multi_plan {
    std::vector<alz_ctx*> ctxs;
    std::vector<alz_plan*> plans;                  // one per context (NULL: no stream of the batch went there)
    std::vector<std::vector<uint32_t>> idx;        // the batch indices of every context's streams, ascending
    uint32_t n = 0;
};
void alz_plan_destroy_multi(alz_multi_plan* mp) {
    if (!mp) return;
    for (size_t q = 0; q < mp->plans.size(); q++) if (mp->plans[q]) alz_plan_destroy(mp->ctxs[q], mp->plans[q]);
    delete mp;
}
int alz_plan_create_multi(alz_ctx* const* ctxs, uint32_t n_ctx, const alz_lz_properties* props, uint32_t n, const alz_stream* streams,
                          const uint32_t* part_of, alz_multi_plan** out, uint32_t* part_of_out) {
    if (!ctxs || n_ctx == 0 || n_ctx > 64 || !out || (n && !streams)) return fail(ALZ_E_INVALID, "alz_plan_create_multi: bad argument");
    for (uint32_t q = 0; q < n_ctx; q++) {
        if (!ctxs[q]) return fail(ALZ_E_INVALID, "alz_plan_create_multi: context %u is NULL", q);
        for (uint32_t r = 0; r < q; r++) if (ctxs[r] == ctxs[q]) return fail(ALZ_E_INVALID, "alz_plan_create_multi: context %u is listed twice (a context is single-threaded)", q);
    }
    std::vector<uint32_t> part(n ? n : 1);
    if (part_of) { for (uint32_t i = 0; i < n; i++) { if (part_of[i] >= n_ctx) return fail(ALZ_E_INVALID, "alz_plan_create_multi: part_of[%u] = %u, %u contexts", i, part_of[i], n_ctx); part[i] = part_of[i]; } }
    else if (int rc = alz_partition_batch(n, streams, n_ctx, part.data(), nullptr)) return rc;
    if (part_of_out) for (uint32_t i = 0; i < n; i++) part_of_out[i] = part[i];
    alz_multi_plan* mp = new (std::nothrow) alz_multi_plan();
    if (!mp) return fail(ALZ_E_NOMEM, "out of memory");
    mp->n = n; mp->ctxs.assign(ctxs, ctxs + n_ctx); mp->plans.assign(n_ctx, nullptr); mp->idx.resize(n_ctx);
    for (uint32_t i = 0; i < n; i++) mp->idx[part[i]].push_back(i);
    std::vector<alz_stream> share;
    for (uint32_t q = 0; q < n_ctx; q++) {
        if (mp->idx[q].empty()) continue;
        share.clear();
        for (uint32_t i : mp->idx[q]) share.push_back(streams[i]);
        if (int rc = alz_plan_create(ctxs[q], props, (uint32_t)share.size(), share.data(), &mp->plans[q])) { alz_plan_destroy_multi(mp); return rc; }
    }
    *out = mp;
    return ALZ_OK;
}
int alz_plan_execute_multi(alz_multi_plan* mp, const void* const* d_src_bases, void* const* d_dst_bases) {
    if (!mp || !d_src_bases || !d_dst_bases) return fail(ALZ_E_INVALID, "alz_plan_execute_multi: bad argument");
    for (size_t q = 0; q < mp->plans.size(); q++) {
        if (!mp->plans[q]) continue;
        if (int rc = alz_plan_execute(mp->ctxs[q], mp->plans[q], d_src_bases[q], d_dst_bases[q], nullptr)) return rc;   // (asynchronous: every device works while the next is being fed)
    }
    return ALZ_OK;
}
int alz_plan_results_multi(alz_multi_plan* mp, alz_result* results) {
    if (!mp || (mp->n && !results)) return fail(ALZ_E_INVALID, "alz_plan_results_multi: bad argument");
    std::vector<alz_result> rs;
    for (size_t q = 0; q < mp->plans.size(); q++) {
        if (!mp->plans[q]) continue;
        rs.resize(mp->idx[q].size());
        if (int rc = alz_plan_results(mp->ctxs[q], mp->plans[q], rs.data())) return rc;
        for (size_t j = 0; j < rs.size(); j++) results[mp->idx[q][j]] = rs[j];
    }
    return ALZ_OK;
}

// One context's share of an encode batch: its raw buffers are packed into a device buffer of their own, the compressed streams come
// back from worst-case-sized slots.
static int encode_share(alz_ctx* c, const alz_lz_properties* props, const alz_settings* settings, const std::vector<uint32_t>& idx,
                        const uint8_t* src_base, const alz_stream* streams, uint8_t* dst_base, alz_result* results, alz_encode_aux* aux) {
    const uint32_t m = (uint32_t)idx.size();
    if (m == 0) return ALZ_OK;
    std::vector<alz_stream> ds(m);
    std::vector<in_seg> ins; ins.reserve(m);
    uint64_t so = 0, dof = 0;
    for (uint32_t j = 0; j < m; j++) {
        const alz_stream& st = streams[idx[j]];
        ds[j] = st;
        ds[j].src_off = so; ins.push_back(in_seg{so, src_base + st.src_off, st.src_len});
        so += ((uint64_t)st.src_len + 16 + 15) & ~15ull;    // (the match finder reads up to 16 bytes past a buffer: never another stream's)
        ds[j].dst_off = dof;
        dof = (dof + st.dst_cap + 255) & ~255ull;
    }
    std::vector<alz_result> rs(m);
    std::vector<alz_encode_aux> ax(m);
    int rc = encode_core(c, props, settings, m, (size_t)so, ds.data(), (size_t)dof, rs.data(), ax.data(), [&](const void*& dsrc, void*& ddst) {
        int r;
        if ((r = grow(c, &c->d_src, &c->d_src_cap, so + 64))) return r;
        if ((r = grow(c, &c->d_dst, &c->d_dst_cap, dof + 64))) return r;
        dsrc = c->d_src; ddst = c->d_dst;
        return upload_segs(c, c->d_src, ins, so);
    });
    if (rc) return rc;
    std::vector<out_seg> outs; outs.reserve(m);
    for (uint32_t j = 0; j < m; j++) {
        results[idx[j]] = rs[j];
        if (aux) aux[idx[j]] = ax[j];
        if (rs[j].dst_len && rs[j].status == ALZ_ST_OK) outs.push_back(out_seg{ds[j].dst_off, dst_base + streams[idx[j]].dst_off, rs[j].dst_len});
    }
    return download_segs(c, outs);
}

// alz_encode_batch over several contexts (one per GPU): streams are independent -- a fresh LzChainMatchFinder per CompressHeaderless
// call (LZSS.cs:135) -- so the batch is dealt out by raw size (LPT) and every share runs on its own device, no collective.
int alz_encode_batch_multi(alz_ctx* const* ctxs, uint32_t n_ctx, const alz_lz_properties* props, const alz_settings* settings, uint32_t n,
                           const uint8_t* src_base, size_t src_bytes, const alz_stream* streams,
                           uint8_t* dst_base, size_t dst_bytes, alz_result* results, alz_encode_aux* aux, uint32_t* part_of_out) {
    if (!ctxs || n_ctx == 0 || n_ctx > 64 || (n && (!streams || !results || !dst_base))) return fail(ALZ_E_INVALID, "alz_encode_batch_multi: bad argument");
    for (uint32_t q = 0; q < n_ctx; q++) {
        if (!ctxs[q]) return fail(ALZ_E_INVALID, "alz_encode_batch_multi: context %u is NULL", q);
        for (uint32_t r = 0; r < q; r++) if (ctxs[r] == ctxs[q]) return fail(ALZ_E_INVALID, "alz_encode_batch_multi: context %u is listed twice (a context is single-threaded)", q);
    }
    for (uint32_t i = 0; i < n; i++) {
        if (streams[i].format >= ALZ_FMT_COUNT) return fail(ALZ_E_INVALID, "stream %u: unknown format %u", i, streams[i].format);
        if (!range_ok(streams[i].src_off, streams[i].src_len, src_bytes)) return fail(ALZ_E_INVALID, "stream %u: source range exceeds src_bytes", i);
        if (!range_ok(streams[i].dst_off, streams[i].dst_cap, dst_bytes)) return fail(ALZ_E_INVALID, "stream %u: destination range exceeds dst_bytes", i);
    }
    // greedy LPT over the raw sizes (the cost of an encode is its positions), longest first onto the least loaded context
    std::vector<uint32_t> part(n ? n : 1), order(n);
    std::vector<uint64_t> load(n_ctx, 0);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return streams[a].src_len > streams[b].src_len; });
    for (uint32_t k = 0; k < n; k++) {
        uint32_t best = 0;
        for (uint32_t q = 1; q < n_ctx; q++) if (load[q] < load[best]) best = q;
        part[order[k]] = best; load[best] += (uint64_t)streams[order[k]].src_len + 4096;
    }
    if (part_of_out) for (uint32_t i = 0; i < n; i++) part_of_out[i] = part[i];
    std::vector<std::vector<uint32_t>> idx(n_ctx);
    for (uint32_t i = 0; i < n; i++) idx[part[i]].push_back(i);
    std::vector<int> rcs(n_ctx, ALZ_OK);
    std::vector<std::string> errs(n_ctx);
    auto work = [&](uint32_t q) {
        rcs[q] = encode_share(ctxs[q], props, settings, idx[q], src_base, streams, dst_base, results, aux);
        if (rcs[q]) errs[q] = g_err;
    };
    if (n_ctx == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (uint32_t q = 1; q < n_ctx; q++) th.emplace_back(work, q);
        work(0);
        for (std::thread& t : th) t.join();
    }
    for (uint32_t q = 0; q < n_ctx; q++) if (rcs[q]) return fail(rcs[q], "context %u (device %d): %s", q, ctxs[q]->device, errs[q].c_str());
    return ALZ_OK;
}

// ---------------------------------------------------------------- measurement helper: device copy bandwidth
typedef unsigned int alz_u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) alz_copy_kernel(const alz_u4* __restrict__ a, alz_u4* __restrict__ b, size_t n16) {
    // four 16-byte loads in flight per lane, then four stores (a block moves 16 KiB per trip)
    const size_t stride = (size_t)gridDim.x * 1024;
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + 768 < n16; i += stride) {
        const alz_u4 v0 = __builtin_nontemporal_load(a + i), v1 = __builtin_nontemporal_load(a + i + 256), v2 = __builtin_nontemporal_load(a + i + 512), v3 = __builtin_nontemporal_load(a + i + 768);
        __builtin_nontemporal_store(v0, b + i); __builtin_nontemporal_store(v1, b + i + 256); __builtin_nontemporal_store(v2, b + i + 512); __builtin_nontemporal_store(v3, b + i + 768);
    }
    for (; i < n16; i += 256) b[i] = a[i];
}
int alz_last_kernel_ms(alz_ctx* c, float* ms) {
    if (!c || !ms) return fail(ALZ_E_INVALID, "bad argument");
    *ms = c->last_kernel_ms;
    return ALZ_OK;
}
int alz_measure_copy_bandwidth(alz_ctx* c, size_t bytes, int iters, double* gb_per_s) {
    if (!c || !gb_per_s || iters < 1 || bytes < (1u << 20)) return fail(ALZ_E_INVALID, "alz_measure_copy_bandwidth: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    void *a = nullptr, *b = nullptr;
    HIP_TRY(hipMalloc(&a, bytes));
    hipError_t e = hipMalloc(&b, bytes);
    if (e != hipSuccess) { (void)hipFree(a); return fail(ALZ_E_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    const size_t n16 = bytes / 16;
    (void)hipMemsetAsync(a, 1, bytes, c->stream);
    alz_copy_kernel<<<dim3(256 * 8), dim3(256), 0, c->stream>>>((const alz_u4*)a, (alz_u4*)b, n16);      // warm-up
    (void)hipEventRecord(c->ev0, c->stream);
    for (int i = 0; i < iters; i++) alz_copy_kernel<<<dim3(256 * 8), dim3(256), 0, c->stream>>>((const alz_u4*)a, (alz_u4*)b, n16);
    (void)hipEventRecord(c->ev1, c->stream);
    e = hipEventSynchronize(c->ev1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
    (void)hipFree(a); (void)hipFree(b);
    if (e != hipSuccess) return fail(ALZ_E_HIP, "copy kernel failed: %s", hipGetErrorString(e));
    *gb_per_s = 2.0 * (double)(n16 * 16) * iters / (ms * 1e-3) / 1e9;   // bytes read + bytes written
    return ALZ_OK;
}

}  // extern "C"
