// alz_prs_table.h -- the control-bit automaton of PRS (Sega/PRS.cs:59-102) as a table, one entry per (entry state, flag byte).
//
// PRS reads its control bits from flag bytes that are fetched when a bit is needed -- possibly in the middle of a token -- but a
// token reads ALL its control bits before its first data byte.  In the byte stream every flag byte is therefore followed by the data
// bytes of exactly the tokens whose LAST control bit lies in it: a GROUP.  Which tokens those are depends only on the flag byte and
// on how much of a token the previous flag byte left unfinished (the entry state):
//
//     state 0  nothing pending            state 1  "0" read (a match; long or short not yet known)
//     state 2  "00" read (a short match)  state 3 / 4  "00h" read, h = 0 / 1 (the short match's first length bit)
//
// so the walk over a stream can advance by a whole flag byte -- about five tokens -- instead of by a token: the only data a group's
// SIZE depends on is, per long match in it (at most four), whether the low three bits of its 16-bit word are zero (a third byte
// follows, PRS.cs:85-90).  Bits are numbered in consumption order (bit 0 first): the callers normalise a flag byte of the
// big-endian variant by reversing it.
//
// Entry = two 32-bit words.
//   word 0 (the scalar walk):  [3:0]   1 + data bytes of the group without the third bytes (the distance to the next flag byte)
//                              [4], [5], [6], [10]  the group has at least 1, 2, 3, 4 long matches (one s_bitcmp0 each, no field to extract)
//                              [9:7]   tokens in the group - 1
//                              [13:11] exit state
//                              [16+4k+3 : 16+4k]  long match k: offset of its first data byte from the flag byte, without the
//                                                 third bytes of the long matches in front of it
//   word 1 (the token lanes):  [3k+2 : 3k]  token k: 0 literal, 1 long match, 4 + 2h + l short match of length 2 + 2h + l
//                              [31:24]      bit k: token k is a long match
#pragma once
#include <stdint.h>

#define ALZ_PRS_STATES 5u

struct AlzPrsTable { uint32_t w[ALZ_PRS_STATES * 256u * 2u]; };

constexpr AlzPrsTable alz_make_prs_table() {
    AlzPrsTable t{};
    for (uint32_t st = 0; st < ALZ_PRS_STATES; st++) {
        for (uint32_t f = 0; f < 256u; f++) {
            uint32_t code[12] = {0};
            uint32_t ntok = 0, i = 0, exit_state = 0;
            // the token the previous flag byte left unfinished
            if (st == 1u) {
                if (f & 1u) { code[ntok++] = 1u; i = 1; }
                else { code[ntok++] = 4u + 2u * ((f >> 1) & 1u) + ((f >> 2) & 1u); i = 3; }
            } else if (st == 2u) { code[ntok++] = 4u + 2u * (f & 1u) + ((f >> 1) & 1u); i = 2; }
            else if (st >= 3u) { code[ntok++] = 4u + 2u * (st - 3u) + (f & 1u); i = 1; }
            while (i < 8u) {
                if ((f >> i) & 1u) { code[ntok++] = 0u; i += 1; continue; }                 // literal  PRS.cs:66-70
                if (i + 1u >= 8u) { exit_state = 1; break; }
                if ((f >> (i + 1u)) & 1u) { code[ntok++] = 1u; i += 2; continue; }            // long match  PRS.cs:73-90
                if (i + 2u >= 8u) { exit_state = 2; break; }
                if (i + 3u >= 8u) { exit_state = 3u + ((f >> (i + 2u)) & 1u); break; }
                code[ntok++] = 4u + 2u * ((f >> (i + 2u)) & 1u) + ((f >> (i + 3u)) & 1u);  // short match  PRS.cs:91-96
                i += 4;
            }
            uint32_t w0 = 0, w1 = 0, nlong = 0, off = 1;
            for (uint32_t k = 0; k < ntok; k++) {
                w1 |= code[k] << (3u * k);
                if (code[k] == 1u) { w1 |= 1u << (24u + k); w0 |= off << (16u + 4u * nlong); nlong++; off += 2; }
                else off += 1;
            }
            const uint32_t atleast = (nlong >= 1u ? 0x10u : 0u) | (nlong >= 2u ? 0x20u : 0u) | (nlong >= 3u ? 0x40u : 0u) | (nlong >= 4u ? 0x400u : 0u);
            w0 |= off | atleast | ((ntok - 1u) << 7) | (exit_state << 11);
            t.w[2u * (st * 256u + f)] = w0;
            t.w[2u * (st * 256u + f) + 1u] = w1;
        }
    }
    return t;
}

// every group completes at least one token and at most eight, its long matches start at offsets below 16, its size fits four bits
constexpr bool alz_check_prs_table() {
    const AlzPrsTable t = alz_make_prs_table();
    for (uint32_t e = 0; e < ALZ_PRS_STATES * 256u; e++) {
        const uint32_t w0 = t.w[2u * e];
        const uint32_t nlong = ((w0 >> 4) & 1u) + ((w0 >> 5) & 1u) + ((w0 >> 6) & 1u) + ((w0 >> 10) & 1u), size = w0 & 15u;
        if (nlong > 4u || size < 2u || size > 13u || ((w0 >> 11) & 7u) >= ALZ_PRS_STATES) return false;
        for (uint32_t k = 0; k < nlong; k++) if (((w0 >> (16u + 4u * k)) & 15u) + 2u > size) return false;
    }
    return true;
}
static_assert(alz_check_prs_table(), "PRS group table");
