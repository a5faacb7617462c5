// metasnv_amd/csrc/crc32_pclmul.cpp -- the folding half of crc32.cpp, compiled with -mpclmul -msse4.1 (Makefile) and only called when
// the CPU has the instruction.  Reflected CRC-32, polynomial 0x1DB710641 (bit-reflected 0x104C11DB7 with the x^32 term).
// Constants (x^n mod P in the bit-reflected domain, as tabulated for this polynomial in the Intel paper and used by every
// PCLMULQDQ implementation of gzip's CRC):
//   fold by 4 x 128 bits:  k1 = x^(4*128+32) mod P = 0x154442bd4,  k2 = x^(4*128-32) mod P = 0x1c6e41596
//   fold by 1 x 128 bits:  k3 = x^(128+32)   mod P = 0x1751997d0,  k4 = x^(128-32)   mod P = 0x0ccaa009e
//   128 -> 64 bits:        k5 = x^64 mod P = 0x163cd6124
//   Barrett:               P = 0x1db710641,  mu = floor(x^64 / P) = 0x1f7011641
#include <cstddef>
#include <cstdint>
#include <immintrin.h>

namespace msnv {

uint32_t crc32_pclmul_fold(uint32_t state, const uint8_t *p, size_t n) {
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    const __m128i lo32 = _mm_setr_epi32(-1, 0, -1, 0);
    auto ld = [](const uint8_t *q) { return _mm_loadu_si128(reinterpret_cast<const __m128i *>(q)); };
    // fold `a` forward by the distance the constants in `k` stand for and add the next 16 bytes
    auto fold = [](const __m128i a, const __m128i k, const __m128i next) {
        return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(a, k, 0x00), _mm_clmulepi64_si128(a, k, 0x11)), next);
    };
    __m128i x1 = _mm_xor_si128(ld(p), _mm_cvtsi32_si128((int)state)), x2 = ld(p + 16), x3 = ld(p + 32), x4 = ld(p + 48);
    p += 64; n -= 64;
    while (n >= 64) {                                               // four independent lanes, 64 bytes a step
        x1 = fold(x1, k1k2, ld(p)); x2 = fold(x2, k1k2, ld(p + 16)); x3 = fold(x3, k1k2, ld(p + 32)); x4 = fold(x4, k1k2, ld(p + 48));
        p += 64; n -= 64;
    }
    x1 = fold(x1, k3k4, x2); x1 = fold(x1, k3k4, x3); x1 = fold(x1, k3k4, x4);      // four lanes into one
    while (n >= 16) { x1 = fold(x1, k3k4, ld(p)); p += 16; n -= 16; }
    // 128 -> 64 bits
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, lo32), k5, 0x00), t);
    // Barrett reduction 64 -> 32 bits
    t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, lo32), poly, 0x10), lo32);
    x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(t, poly, 0x00));
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

}  // namespace msnv
