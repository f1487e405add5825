// tbk_common.h — layout and arithmetic shared by host and device code of libtbk_hip.so.
//
// Table layout in HBM (replicated per GPU).
//   The classifier holds the two k-mer lists as two open-addressing tables of uint64 slots
//   that share one bucket index and are interleaved line by line: bucket i is one 128-byte
//   aligned line = [8 hapA slots | 8 hapB slots].  Measured on MI355X (profiles/,
//   DESIGN.md §4): random line reads saturate at ~48 G lines/s whether the line is 64 or
//   128 bytes, so fetching both tables' buckets as ONE 128-byte line halves the cost of a
//   window compared with two independent 64-byte lines.  hapA and hapB stay separate tables
//   (separate slots, separate probes, hapA priority applied afterwards).
//   A slot holds a packed k-mer (base i at bits 2i..2i+1, A=0 C=1 G=2 T=3 — the reference's
//   encoding, c/kmers.c:50-72) or TBK_EMPTY.  A key lives in the first bucket, walking from
//   its home bucket, whose half had a free slot when it was inserted (linear probing at
//   line granularity), so a lookup stops at the first half-line that still has a free slot.
//   A standalone list (tbk_table) is just its packed keys in HBM; a single-table form of
//   the same layout (64-byte lines, 8 slots) is built on demand for tbk_table_contains.
//   The reference's layout (8-byte slots + a parallel "full" byte array at load 0.75,
//   c/kmers.c:12-38,160-180) is not observable; only membership is.
//
// TBK_EMPTY = all ones is never a canonical lookup key: for k < 32 every key is < 4^k, and
// for k = 32 all-ones is T x32 whose reverse complement A x32 = 0 is smaller.  A list line
// that encodes to all-ones (only "T"x32) is dropped at insert; it could never be matched
// (the reference stores list lines verbatim, c/kmers.c:113, and looks up min(fwd, rc),
// c/kmers.c:255).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define TBK_HD __host__ __device__ __forceinline__
#else
#define TBK_HD inline
#endif

#define TBK_EMPTY 0xFFFFFFFFFFFFFFFFull
#define TBK_SLOTS_PER_BUCKET 8
#define TBK_BUCKET_BYTES 64

// 64 -> 32-bit mixer for bucket selection.  Not the reference's hash_function
// (c/kmers.c:98-103): hash values are not observable, so this one is chosen to be cheap on
// the VALU (three 32-bit multiplies) with good high bits for the multiply-shift range
// reduction below.
TBK_HD uint32_t tbk_mix32(uint64_t key) {
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t h = lo * 0x9E3779B1u ^ (hi + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0xC2B2AE3Du;
    h ^= h >> 13;
    return h;
}

// bucket = floor(h * n_buckets / 2^32): uniform over [0, n_buckets) without a division.
TBK_HD uint32_t tbk_reduce(uint32_t h, uint32_t n_buckets) {
    return (uint32_t)(((uint64_t)h * (uint64_t)n_buckets) >> 32);
}

TBK_HD uint32_t tbk_home_bucket(uint64_t key, uint32_t n_buckets) {
    return tbk_reduce(tbk_mix32(key), n_buckets);
}

// Device view of a table: bucket b's slots for this list start at
// slots[b * stride + half] (stride 8, half 0 for a standalone table; stride 16 and half 0 / 8
// for the hapA / hapB halves of a paired table).
struct TbkTableView {
    const uint64_t *slots;
    uint32_t n_buckets;
    uint32_t stride;  // slots per bucket line (8 or 16)
    uint32_t half;    // first slot of this list inside the line (0 or 8)
};

// The paired (hapA | hapB) table the probe kernel reads: bucket b = 16 slots = 128 bytes.
struct TbkPairView {
    const uint64_t *slots;  // n_buckets * 16
    uint32_t n_buckets;
};

// ---- synthetic key sequence (bench inputs) ---------------------------------
// Key i of `seed` is a k-mer whose k-2 middle bases are a bijective scramble of i over
// 2(k-2) bits (distinct i -> distinct k-mer) and whose end bases (b0, b_{k-1}) satisfy
// b0 + b_{k-1} < 3, which makes the packed integer strictly smaller than its reverse
// complement's (the top base pair decides: b_{k-1} < 3 - b0), i.e. the k-mer is canonical.
// Requires 3 <= k <= 32 and i < 4^(k-2).
TBK_HD uint64_t tbk_synth_key(uint64_t seed, uint64_t i, int k) {
    const int mb = 2 * (k - 2);
    const uint64_t mm = mb >= 64 ? ~0ull : ((1ull << mb) - 1ull);
    uint64_t x = (i + seed * 0x9E3779B97F4A7C15ull) & mm;
    const int s1 = mb / 2 + 1 > 63 ? 63 : mb / 2 + 1;
    x = (x * 0xD1342543DE82EF95ull) & mm;
    x ^= x >> s1;
    x = (x * 0xAF251AF3B0F025B5ull) & mm;
    x ^= x >> s1;
    x = (x * 0x9E3779B97F4A7C15ull) & mm;
    x ^= x >> s1;
    // ends: 6 admissible (b0, b_{k-1}) pairs, picked by a hash of i
    uint64_t e = (i ^ seed) * 0xC6A4A7935BD1E995ull;
    e ^= e >> 29;
    const uint32_t pick = (uint32_t)((e >> 11) % 6u);
    // pairs: (0,0) (0,1) (0,2) (1,0) (1,1) (2,0)
    const uint32_t b0 = pick < 3 ? 0u : (pick < 5 ? 1u : 2u);
    const uint32_t bl = pick < 3 ? pick : (pick < 5 ? pick - 3u : 0u);
    return (uint64_t)b0 | (x << 2) | ((uint64_t)bl << (2 * (k - 1)));
}

// Reverse complement of a packed k-mer (used by generators and host utilities).
TBK_HD uint64_t tbk_revcomp_packed(uint64_t x, int k) {
    uint64_t y = ~x;
    // reverse the order of the 32 two-bit groups
    y = ((y >> 2) & 0x3333333333333333ull) | ((y & 0x3333333333333333ull) << 2);
    y = ((y >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((y & 0x0F0F0F0F0F0F0F0Full) << 4);
    y = ((y >> 8) & 0x00FF00FF00FF00FFull) | ((y & 0x00FF00FF00FF00FFull) << 8);
    y = ((y >> 16) & 0x0000FFFF0000FFFFull) | ((y & 0x0000FFFF0000FFFFull) << 16);
    y = (y >> 32) | (y << 32);
    return y >> (64 - 2 * k);
}

// splitmix64 step, the counter-based PRNG of the read generator.
TBK_HD uint64_t tbk_splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
