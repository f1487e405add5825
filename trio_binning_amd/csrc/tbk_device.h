// tbk_device.h — device helpers shared by the probe kernels (tbk_kernels.hip) and the counting
// kernels (tbk_count_kernels.hip): the geometry of a wave pass and the packing of the read stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int TBK_WPL = 32;                 // windows per lane per pass
constexpr int TBK_PASS = 64 * TBK_WPL;      // window starts per wave pass (2048)
constexpr int TBK_CHUNKS = 130;             // 128 chunks of 16 bases + 2 halo chunks

// Pack 16 ASCII bases (4 little-endian words) into 2-bit codes and a not-ACGT mask.
__device__ __forceinline__ void pack4(uint32_t w, uint32_t &code8, uint32_t &bad4) {
    // code = ((c >> 1) ^ (c >> 2)) & 3 : A(0x41)->0 C(0x43)->1 G(0x47)->2 T(0x54)->3
    uint32_t c = ((w >> 1) ^ (w >> 2)) & 0x03030303u;
    // the byte each code stands for: 0x41 + 2*lo + 6*hi + 11*(lo&hi)
    const uint32_t lo = c & 0x01010101u, hi = (c >> 1) & 0x01010101u;
    const uint32_t expect = 0x41414141u + 2u * lo + 6u * hi + 11u * (lo & hi);
    const uint32_t diff = w ^ expect;
    uint32_t nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;  // 0x80 per bad byte
    nz >>= 7;
    bad4 = (nz | (nz >> 7) | (nz >> 14) | (nz >> 21)) & 0xFu;
    c |= c >> 6;
    code8 = (c | (c >> 12)) & 0xFFu;
}

__device__ __forceinline__ uint64_t pack16(uint4 v) {
    uint32_t c0, c1, c2, c3, b0, b1, b2, b3;
    pack4(v.x, c0, b0); pack4(v.y, c1, b1); pack4(v.z, c2, b2); pack4(v.w, c3, b3);
    const uint32_t code = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    const uint32_t bad = b0 | (b1 << 4) | (b2 << 8) | (b3 << 12);
    return (uint64_t)code | ((uint64_t)bad << 32);
}

// Load the 16-byte chunk that starts at stream position pos; bytes at or past `total`
// read as 0 (not ACGT).  The stream base is 16-byte aligned (hipMalloc) and pos is a
// multiple of 16.
__device__ __forceinline__ uint64_t load_chunk(const uint8_t *bases, uint64_t pos, uint64_t total) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (pos + 16 <= total) {
        v = *reinterpret_cast<const uint4 *>(bases + pos);
    } else if (pos < total) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; pos + i < total; i++) w[i >> 2] |= (uint32_t)bases[pos + i] << (8 * (i & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return pack16(v);
}

// The same chunk from the packed transfer format: its code word and its dense 16-bit mask; chunks
// past the end of the stream read as sixteen not-ACGT bases (the last, partial chunk carries the
// mask bits of the positions past the end already - tbk_pack.cpp).
__device__ __forceinline__ uint64_t load_packed_chunk(const uint32_t *codes, const uint16_t *bad16, uint64_t chunk, uint64_t n_chunks) {
    if (chunk >= n_chunks) return 0xFFFFull << 32;
    return (uint64_t)codes[chunk] | ((uint64_t)bad16[chunk] << 32);
}

// reverse the order of the sixteen 2-bit groups of a word
__device__ __forceinline__ uint32_t rev_pairs(uint32_t x) {
    x = __brev(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

