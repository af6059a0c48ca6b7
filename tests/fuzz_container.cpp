// fuzz_container.cpp -- the HOST-side header parsers of csrc/alz_container.cpp (alz_container_is_match, alz_container_decompressed_size:
// what a caller runs on untrusted bytes before anything reaches the GPU) under AddressSanitizer + UndefinedBehaviorSanitizer.
// Built by `make -C oracle fuzz_container` from alz_container.cpp itself with g++ (no HIP: the GPU entry points the file calls are
// link-time stand-ins that refuse; the parsers never reach them).  Every input lives in an exact-size heap buffer, so one byte read past
// the end is a report.  Inputs: every committed container KAT file (tests/golden/kat_containers.json, passed as hex on stdin by the test),
// every prefix of it, and seeded mutations (byte flips, size-field extremes, truncations) -- for all ALZ_C_COUNT containers, both byte
// orders.  Test infrastructure only.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "auroralz.h"

extern "C" {   // stand-ins for the GPU side of the ABI (never called by the parsers)
int alz_decode(alz_ctx*, uint32_t, const alz_lz_properties*, const uint8_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t, alz_result*) { return ALZ_E_NO_DEVICE; }
int alz_decode_batch(alz_ctx*, const alz_lz_properties*, uint32_t, const uint8_t*, size_t, const alz_stream*, uint8_t*, size_t, alz_result*) { return ALZ_E_NO_DEVICE; }
int alz_encode_batch(alz_ctx*, const alz_lz_properties*, const alz_settings*, uint32_t, const uint8_t*, size_t, const alz_stream*, uint8_t*, size_t, alz_result*, alz_encode_aux*) { return ALZ_E_NO_DEVICE; }
int alz_device_malloc(alz_ctx*, size_t, void**) { return ALZ_E_NO_DEVICE; }
int alz_device_free(alz_ctx*, void*) { return ALZ_E_NO_DEVICE; }
int alz_plan_create(alz_ctx*, const alz_lz_properties*, uint32_t, const alz_stream*, alz_plan**) { return ALZ_E_NO_DEVICE; }
int alz_plan_execute(alz_ctx*, alz_plan*, const void*, void*, void*) { return ALZ_E_NO_DEVICE; }
int alz_plan_results(alz_ctx*, alz_plan*, alz_result*) { return ALZ_E_NO_DEVICE; }
void alz_plan_destroy(alz_ctx*, alz_plan*) {}
int alz_memcpy_h2d(alz_ctx*, void*, const void*, size_t) { return ALZ_E_NO_DEVICE; }
int alz_memcpy_d2h(alz_ctx*, void*, const void*, size_t) { return ALZ_E_NO_DEVICE; }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

static uint64_t calls = 0;
static void probe(const uint8_t* data, size_t len) {
    uint8_t* buf = (uint8_t*)malloc(len ? len : 1);          // exact size: ASan sees the first byte past the end
    if (len) memcpy(buf, data, len);
    for (uint32_t c = 0; c < ALZ_C_COUNT; c++) {
        (void)alz_container_is_match(c, buf, len); calls++;
        for (uint32_t be = 0; be < 2; be++) {
            alz_container_options opt; memset(&opt, 0, sizeof(opt)); opt.big_endian = be;
            uint32_t size = 0;
            (void)alz_container_decompressed_size(c, &opt, buf, len, &size); calls++;
        }
        uint32_t size = 0;
        (void)alz_container_decompressed_size(c, nullptr, buf, len, &size); calls++;
    }
    free(buf);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    std::vector<std::vector<uint8_t>> seeds;
    char line[1 << 16];
    while (fgets(line, sizeof(line), stdin)) {               // one hex string per line
        std::vector<uint8_t> v;
        for (size_t i = 0; line[i] && line[i + 1] && line[i] != '\n'; i += 2) { unsigned b; if (sscanf(line + i, "%2x", &b) != 1) break; v.push_back((uint8_t)b); }
        if (!v.empty()) seeds.push_back(v);
    }
    if (seeds.empty()) { seeds.push_back(std::vector<uint8_t>(64, 0)); }
    probe(nullptr, 0);
    for (auto& s : seeds) {
        for (size_t n = 0; n <= s.size() && n <= 96; n++) probe(s.data(), n);          // every short prefix: the header parsers live in the first bytes
        probe(s.data(), s.size());
        for (int r = 0; r < rounds; r++) {
            std::vector<uint8_t> m = s;
            const uint32_t kind = rnd() % 5u;
            const size_t head = m.size() < 64 ? m.size() : 64;
            if (kind == 0 && head) m[rnd() % head] ^= (uint8_t)(1u << (rnd() % 8u));
            else if (kind == 1 && head >= 4) { const size_t at = rnd() % (head - 3); const uint32_t ext[4] = {0u, 0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u}; memcpy(&m[at], &ext[rnd() % 4u], 4); }
            else if (kind == 2) m.resize(rnd() % (m.size() + 1));
            else if (kind == 3 && head) { for (int k = 0; k < 4; k++) m[rnd() % head] = (uint8_t)rnd(); }
            else { const size_t extra = rnd() % 32u; for (size_t k = 0; k < extra; k++) m.push_back((uint8_t)rnd()); }
            probe(m.data(), m.size());
        }
    }
    printf("fuzz_container: %zu seeds, %llu parser calls, no report\n", seeds.size(), (unsigned long long)calls);
    return 0;
}
