// tbk_crc.cpp — CRC-32 (the gzip polynomial) by carry-less multiplication.
//
// Every byte a gzip member holds is summed once when it is read and once when it is written, and
// zlib 1.2.11's table-driven crc32() does about 1 GB/s a core: a third of the time of the bin
// writer's encoder and half of the marker-resolving pass of the several-thread inflater.  This is
// the folding scheme of Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ"
// (Intel, 2009), with the constants for the bit-reflected polynomial 0xEDB88320: four 128-bit lanes
// folded 512 bits at a time, then 128 bits at a time, then 128 -> 64 -> 32 bits and a Barrett
// reduction.  Same values as zlib's crc32() (tests/test_host_native_io.py compares them on ragged
// lengths and alignments, and tbk_crc32 itself checks a known answer once and falls back to zlib if
// the CPU or the answer is not what it expects).
#include <immintrin.h>
#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace {

#define TBK_CLMUL_TARGET __attribute__((target("pclmul,sse4.1")))
TBK_CLMUL_TARGET inline __m128i fold_step(__m128i x, __m128i k, __m128i next) {
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), next);
}
TBK_CLMUL_TARGET inline __m128i load16(const uint8_t *p) { return _mm_loadu_si128((const __m128i *)p); }

// raw register in, raw register out (no pre/post inversion); len >= 64 and a multiple of 16
TBK_CLMUL_TARGET uint32_t fold(uint32_t crc, const uint8_t *buf, size_t len) {
    const __m128i R2R1 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll);    // x^(512+64), x^512 mod P, reflected
    const __m128i R4R3 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);    // x^(128+64), x^128
    const __m128i R5 = _mm_set_epi64x(0, 0x0000000163cd6124ll);                         // x^64
    const __m128i MASK32 = _mm_set_epi64x(0, 0xFFFFFFFFll);
    const __m128i RU = _mm_set_epi64x(0x00000001F7011641ll, 0x00000001DB710641ll);      // mu, P
    __m128i x1 = _mm_xor_si128(load16(buf), _mm_cvtsi32_si128((int)crc)), x2 = load16(buf + 16), x3 = load16(buf + 32), x4 = load16(buf + 48);
    buf += 64; len -= 64;
    while (len >= 64) {
        x1 = fold_step(x1, R2R1, load16(buf));
        x2 = fold_step(x2, R2R1, load16(buf + 16));
        x3 = fold_step(x3, R2R1, load16(buf + 32));
        x4 = fold_step(x4, R2R1, load16(buf + 48));
        buf += 64; len -= 64;
    }
    x1 = fold_step(x1, R4R3, x2);
    x1 = fold_step(x1, R4R3, x3);
    x1 = fold_step(x1, R4R3, x4);
    while (len >= 16) {
        x1 = fold_step(x1, R4R3, load16(buf));
        buf += 16; len -= 16;
    }
    // 128 -> 64 bits (this also appends the 32 zero bits of the CRC's definition)
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), _mm_clmulepi64_si128(R4R3, x1, 0x01));
    // 64 -> 32 bits
    __m128i x2b = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, MASK32), R5, 0x00), x2b);
    // Barrett reduction
    __m128i t = _mm_clmulepi64_si128(_mm_and_si128(x1, MASK32), RU, 0x10);
    t = _mm_clmulepi64_si128(_mm_and_si128(t, MASK32), RU, 0x00);
    return (uint32_t)_mm_extract_epi32(_mm_xor_si128(x1, t), 1);
}

bool usable() {
    const char *e = getenv("TBK_CRC");
    if (e && strcmp(e, "zlib") == 0) return false;
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
    uint8_t probe[208];
    for (size_t i = 0; i < sizeof probe; i++) probe[i] = (uint8_t)(i * 37 + 11);
    const uint32_t want = (uint32_t)crc32(crc32(0L, Z_NULL, 0), probe, (uInt)sizeof probe);
    return ~fold(~0u, probe, sizeof probe) == want;
}

}  // namespace

// zlib's crc32(crc, p, n) for any n (size_t), the bulk of it by carry-less multiplication
uint32_t tbk_crc32(uint32_t crc, const uint8_t *p, size_t n) {
    static const bool fast = usable();
    if (fast && n >= 64) {
        const size_t bulk = n & ~(size_t)15;
        crc = ~fold(~crc, p, bulk);
        p += bulk; n -= bulk;
    }
    while (n) {
        const size_t m = n < ((size_t)1 << 30) ? n : ((size_t)1 << 30);
        crc = (uint32_t)crc32(crc, p, (uInt)m);
        p += m; n -= m;
    }
    return crc;
}

extern "C" uint32_t tbk_crc32_c(uint32_t crc, const uint8_t *p, size_t n) { return tbk_crc32(crc, p, n); }
