// metasnv_amd/csrc/crc32.cpp -- CRC-32 (IEEE 802.3, reflected polynomial 0xEDB88320: the one in the gzip / BGZF trailer) of a byte range.
//
// Every BGZF block the host inflates is checked against its trailer (round 3; htslib does the same inside bgzf_read, qaCompute.cpp:441),
// which made the CRC a fifth of the inflate threads' time with the system zlib's byte-table crc32 (~0.9 GB/s on the build host).  Two
// forms, picked once at run time:
//   * carry-less multiplication (crc32_pclmul.cpp, compiled with -mpclmul -msse4.1): four 16-byte lanes folded per 64 bytes, then one
//     lane per 16 bytes, then Barrett reduction -- the published folding scheme (Gopal et al., "Fast CRC Computation for Generic
//     Polynomials Using PCLMULQDQ Instruction", Intel 2009) with the constants of this polynomial;
//   * eight table lookups per eight bytes (slicing) where the instruction is missing, and for what the folding leaves over.
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "msnv_internal.h"

namespace msnv {

uint32_t crc32_pclmul_fold(uint32_t state, const uint8_t *p, size_t n);      // crc32_pclmul.cpp: n >= 64, a multiple of 16; state = ~crc in, ~crc out

namespace {

uint32_t g_tab[8][256];
bool g_have_clmul = false;

struct Init {
    Init() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            g_tab[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) g_tab[t][i] = (g_tab[t - 1][i] >> 8) ^ g_tab[0][g_tab[t - 1][i] & 0xffu];
#if defined(__x86_64__) || defined(__i386__)
        __builtin_cpu_init();
        g_have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
#endif
        if (const char *e = getenv("MSNV_CRC")) if (e[0] == 't') g_have_clmul = false;      // MSNV_CRC=table: the other form (tests)
    }
} g_init;

// state in / out: the running register (= ~crc)
inline uint32_t sliced(uint32_t s, const uint8_t *p, size_t n) {
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= s;                                                   // (little-endian hosts only: the build targets x86-64)
        s = g_tab[7][v & 0xffu] ^ g_tab[6][(v >> 8) & 0xffu] ^ g_tab[5][(v >> 16) & 0xffu] ^ g_tab[4][(v >> 24) & 0xffu] ^
            g_tab[3][(v >> 32) & 0xffu] ^ g_tab[2][(v >> 40) & 0xffu] ^ g_tab[1][(v >> 48) & 0xffu] ^ g_tab[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) s = (s >> 8) ^ g_tab[0][(s ^ *p++) & 0xffu];
    return s;
}

}  // namespace

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
    uint32_t s = ~crc;
    if (g_have_clmul && n >= 64) {
        const size_t m = n & ~(size_t)15;
        s = crc32_pclmul_fold(s, p, m);
        p += m; n -= m;
    }
    return ~sliced(s, p, n);
}

uint32_t bgzf_crc32(const uint8_t *data, uint32_t n) { return crc32_update(0u, data, n); }

}  // namespace msnv
