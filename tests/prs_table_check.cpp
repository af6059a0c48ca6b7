// prs_table_check.cpp -- host check of csrc/alz_prs_table.h (built and run by tests/test_prs_table_cpu.py).
//
// Parses random PRS streams twice: token by token, the way Sega/PRS.cs:59-102 reads them (flag bytes fetched when a bit is
// needed), and group by group through the table the GPU walk uses (scalar side: word 0, the chain of "third byte follows" tests;
// token side: word 1) -- and compares every token (kind, position of its data, length field) and every flag-byte position.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "alz_prs_table.h"

struct Tok { int kind; uint32_t pos; uint32_t len; };   // kind 0 literal, 1 long, 2 short; pos = offset of the first data byte

static uint64_t rs = 0x1234567ull;
static uint32_t rnd() { rs = rs * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rs >> 33); }

static bool ext_at(const std::vector<uint8_t>& d, uint32_t p, bool big) { return ((big ? d[p + 1] : d[p]) & 7u) == 0u; }

int main() {
    static const AlzPrsTable T = alz_make_prs_table();
    for (int big = 0; big < 2; big++) for (int it = 0; it < 20000; it++) {
        const uint32_t n = 64 + rnd() % 400;
        std::vector<uint8_t> d(n + 32);
        const uint32_t mode = rnd() % 4;
        for (auto& b : d) { b = (uint8_t)rnd(); if (mode == 1 && (rnd() & 1)) b &= 0xF8; if (mode == 2) b |= (rnd() & 1) ? 0xAA : 0x55; if (mode == 3 && (rnd() % 3) == 0) b = 0; }
        // ---- serial reference parse up to position n (never stops at a zero word: the walk does not either)
        std::vector<Tok> ref; std::vector<uint32_t> refflags;
        uint32_t p = 0, bits = 0, flag = 0;
        auto rb = [&]() { if (!bits) { refflags.push_back(p); flag = d[p++]; bits = 8; } uint32_t b = big ? (flag >> (bits - 1)) & 1u : (flag >> (8 - bits)) & 1u; bits--; return b; };
        while (p < n) {
            if (rb()) { ref.push_back({0, p, 1}); p += 1; }
            else if (rb()) { const bool e = ext_at(d, p, big != 0); ref.push_back({1, p, e ? 3u : 2u}); p += e ? 3 : 2; }
            else { const uint32_t h = rb(), l = rb(); ref.push_back({2, p, 2 + 2 * h + l}); p += 1; }
        }
        // ---- group parse
        std::vector<Tok> got; std::vector<uint32_t> gotflags;
        uint32_t pos = 0, st = 0;
        while (pos < n) {
            gotflags.push_back(pos);
            uint32_t f = d[pos];
            if (big) { uint32_t r = 0; for (int i = 0; i < 8; i++) r |= ((f >> i) & 1u) << (7 - i); f = r; }
            const uint32_t w0 = T.w[2 * (st * 256 + f)], w1 = T.w[2 * (st * 256 + f) + 1];
            // scalar side: size of the group
            uint32_t B = pos; const uint32_t nl = ((w0 >> 4) & 1u) + ((w0 >> 5) & 1u) + ((w0 >> 6) & 1u) + ((w0 >> 10) & 1u);
            for (uint32_t k = 0; k < nl; k++) B += ext_at(d, B + ((w0 >> (16 + 4 * k)) & 15u), big != 0) ? 1u : 0u;
            // token side
            const uint32_t ntok = ((w0 >> 7) & 7u) + 1u;
            for (uint32_t k = 0; k < ntok; k++) {
                const uint32_t code = (w1 >> (3 * k)) & 7u, lm = w1 >> 24;
                const uint32_t nlb = (uint32_t)__builtin_popcount(lm & ((1u << k) - 1u));
                uint32_t E = 0;
                for (uint32_t j = 0; j < nlb; j++) E += ext_at(d, pos + ((w0 >> (16 + 4 * j)) & 15u) + E, big != 0) ? 1u : 0u;
                const uint32_t dp = pos + 1 + (k - nlb) + 2 * nlb + E;
                if (code == 0) got.push_back({0, dp, 1});
                else if (code == 1) got.push_back({1, dp, ext_at(d, dp, big != 0) ? 3u : 2u});
                else got.push_back({2, dp, 2 + (code & 3u)});
            }
            pos = B + (w0 & 15u);
            st = (w0 >> 11) & 7u;
        }
        // the group parse may have completed a few more tokens than the serial one (it stops at a group boundary); compare the common prefix
        size_t m = ref.size() < got.size() ? ref.size() : got.size();
        if (got.size() < ref.size() && ref.size() - got.size() > 0) {
            // the serial parse can only be ahead by tokens of a group that started before n: never, both stop at >= n
        }
        for (size_t i = 0; i < m; i++) if (ref[i].kind != got[i].kind || ref[i].pos != got[i].pos || ref[i].len != got[i].len) {
            printf("MISMATCH big=%d it=%d token %zu: ref (%d,%u,%u) got (%d,%u,%u)\n", big, it, i, ref[i].kind, ref[i].pos, ref[i].len, got[i].kind, got[i].pos, got[i].len);
            return 1;
        }
        size_t mf = refflags.size() < gotflags.size() ? refflags.size() : gotflags.size();
        for (size_t i = 0; i < mf; i++) if (refflags[i] != gotflags[i]) { printf("FLAG MISMATCH big=%d it=%d flag %zu: %u vs %u\n", big, it, i, refflags[i], gotflags[i]); return 1; }
        if (m + 12 < ref.size() || mf + 2 < refflags.size()) { printf("SHORT big=%d it=%d: %zu / %zu tokens, %zu / %zu flags\n", big, it, got.size(), ref.size(), gotflags.size(), refflags.size()); return 1; }
    }
    printf("ok\n");
    return 0;
}
