// tbk_kernels.hip — the MI355X (gfx950, wave64) kernels of the classify-by-kmers path.
//
//   tbk_probe_kernel   replaces the per-window loop of count_kmers_in_read
//                      (c/kmers.c:270-299) and both kmer_in_hash_set probes
//                      (c/kmers.c:245-268) for a whole batch of reads.
//   tbk_insert_kernel  replaces add_to_hash (c/kmers.c:112-122).
//   tbk_contains_kernel  raw-key membership (tests).
//
// Integer/hash work: no MFMA.  The bound is HBM random-line throughput.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tbk_common.h"

// =======================================================================================
// insert
// =======================================================================================
// One key per thread.  Scan this list's 8 slots of the home bucket; claim the first free
// slot with a 64-bit CAS; a bucket half without a free slot sends the key to the next
// bucket.  Duplicates are detected (the reference stores them twice, c/kmers.c:112-122;
// membership is the same).  `stride`/`half` select a standalone table (8, 0) or the hapA /
// hapB half of a paired table (16, 0 / 8).
__global__ void __launch_bounds__(256)
tbk_insert_kernel(uint64_t *__restrict__ slots, uint32_t n_buckets, uint32_t stride, uint32_t half,
                  const uint64_t *__restrict__ keys, uint64_t n,
                  unsigned long long *__restrict__ n_distinct, int *__restrict__ failed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (; i < n; i += step) {
        const uint64_t key = keys[i];
        if (key == TBK_EMPTY) continue;
        uint32_t b = tbk_home_bucket(key, n_buckets);
        bool done = false;
        for (uint32_t walked = 0; walked < n_buckets && !done; walked++) {
            unsigned long long *line = (unsigned long long *)(slots + (uint64_t)b * stride + half);
            for (int s = 0; s < TBK_SLOTS_PER_BUCKET && !done; s++) {
                unsigned long long cur = __hip_atomic_load(&line[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == key) { done = true; break; }
                if (cur == TBK_EMPTY) {
                    unsigned long long old = atomicCAS(&line[s], (unsigned long long)TBK_EMPTY, (unsigned long long)key);
                    if (old == TBK_EMPTY) { mine++; done = true; }
                    else if (old == key) { done = true; }
                    // else: somebody else's key took the slot; keep scanning
                }
            }
            if (!done) { b++; if (b == n_buckets) b = 0; }
        }
        if (!done) atomicExch(failed, 1);
    }
    if (mine) atomicAdd(n_distinct, mine);
}

// =======================================================================================
// contains (raw keys; one thread per key, whole-bucket scan) — test utility, not the hot path
// =======================================================================================
__global__ void __launch_bounds__(256)
tbk_contains_kernel(TbkTableView t, const uint64_t *__restrict__ keys, uint64_t n,
                    uint8_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    uint8_t found = 0;
    if (key != TBK_EMPTY) {
        uint32_t b = tbk_home_bucket(key, t.n_buckets);
        for (uint32_t walked = 0; walked < t.n_buckets; walked++) {
            const uint64_t *line = t.slots + (uint64_t)b * t.stride + t.half;
            bool has_free = false;
            for (int s = 0; s < TBK_SLOTS_PER_BUCKET; s++) {
                uint64_t cur = line[s];
                if (cur == key) found = 1;
                if (cur == TBK_EMPTY) has_free = true;
            }
            if (found || has_free) break;
            b++; if (b == t.n_buckets) b = 0;
        }
    }
    out[i] = found;
}

// =======================================================================================
// probe
// =======================================================================================
// Work decomposition.  The batch is one byte stream of `total` bases; window starts are
// cut into PASSes of 1024 consecutive positions.  One wave owns one pass: lane l owns the
// 16 window starts P0+16l .. P0+16l+15 and needs bases P0+16l .. P0+16l+15+k-1 <= 47
// bases = three 16-base chunks.  Each lane loads one 16-byte chunk of the read stream
// (coalesced: 1 KiB per wave-instruction), packs it to 32 bits of 2-bit codes + a 16-bit
// "not ACGT" mask, and stages it in LDS; lanes 0/1 also stage the two halo chunks.  Each
// lane then reads its three chunks back (conflict-free b64 reads) and holds a 96-bit
// forward stream and the 96-bit reverse-complement stream in registers, from which each
// window's forward and reverse-complement k-mer are bit slices (no per-base loop).
//
// Probing is quad-cooperative: a bucket is one 128-byte line [8 hapA slots | 8 hapB slots]
// read by the 4 lanes of a quad, each lane taking 16 bytes (2 slots) of the hapA half and
// 16 bytes of the hapB half, so one wave-instruction touches 16 whole lines with 4 adjacent
// lanes per line (coalesced bucket-line loads) and a window costs ONE random line for both
// probes.  In sub-step (j, s) quad q probes window j of its lane s: key and bucket index are
// broadcast inside the quad with DPP quad_perm moves, each lane compares its slots, and the
// v_cmp results are the wave ballots: a key is stored at most once per table, so
// popcount(ballot) is the number of windows that hit.  hapA has priority over hapB
// (c/kmers.c:291-294): both halves are fetched concurrently and the hapB ballot is masked
// by the quad-expanded hapA ballot.
//
// Per-read attribution.  A pass that lies inside one read (the usual case for long
// reads) accumulates its two counts in scalar registers and issues one atomicAdd pair.
// A pass that touches several reads attributes each hit to the read that contains the
// window start (per-quad read index, broadcast alongside the key).

constexpr int TBK_WPL = 16;                 // windows per lane per pass
constexpr int TBK_PASS = 64 * TBK_WPL;      // window starts per wave pass
constexpr int TBK_WAVES_PER_BLOCK = 4;
constexpr int TBK_CHUNKS = 66;              // 64 chunks + 2 halo chunks of 16 bases

struct ProbeArgs {
    const uint8_t *bases;
    const uint64_t *offsets;  // n_reads + 1
    uint64_t n_reads;
    uint64_t total;           // offsets[n_reads]
    uint64_t n_passes;
    TbkPairView t;            // hapA | hapB interleaved
    int k;
    int32_t *counts;          // [n_reads][2], zeroed by the caller
};

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

// reverse the order of the sixteen 2-bit groups of a word
__device__ __forceinline__ uint32_t rev_pairs(uint32_t x) {
    x = __brev(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

// largest r in [0, n_reads] with offsets[r] <= pos (pos <= total, offsets[n_reads] = total)
__device__ __forceinline__ uint64_t find_read(const uint64_t *offsets, uint64_t n_reads, uint64_t pos) {
    uint64_t lo = 0, hi = n_reads + 1;  // offsets[lo] <= pos < offsets[hi] (virtually +inf)
    while (hi - lo > 1) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

template <int S>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    // DPP quad_perm:[S,S,S,S]
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, S * 0x55, 0xF, 0xF, true);
}

// per-quad OR of a lane mask, result at each quad's lane 0 bit
__device__ __forceinline__ uint64_t quad_any(uint64_t m) {
    return (m | (m >> 1) | (m >> 2) | (m >> 3)) & 0x1111111111111111ull;
}

// Continue a lookup past buckets whose half (hapA: half = 0, hapB: half = 8) had no free
// slot.  `pending` has a bit at lane 0 of every quad that must keep walking; returns the
// quads (lane-0 bits) that found the key.  Rare: a half is full with probability < 1% at
// the load factors the library builds.
__device__ __noinline__ uint64_t probe_walk(const TbkPairView t, uint32_t half, uint64_t key, uint32_t bucket,
                                            uint64_t pending, uint32_t sub) {
    uint64_t found = 0;
    const uint64_t my_quad_bit = 1ull << (__lane_id() & ~3u);
    uint32_t guard = 0;
    while (pending && guard++ < t.n_buckets) {
        const bool act = (pending & my_quad_bit) != 0;
        bucket = bucket + 1 == t.n_buckets ? 0 : bucket + 1;
        ulonglong2 v = make_ulonglong2(0, 0);
        if (act) v = *reinterpret_cast<const ulonglong2 *>(t.slots + (uint64_t)bucket * 16 + half + sub * 2);
        const uint64_t hit = quad_any(__ballot(act && (v.x == key || v.y == key)));
        const uint64_t fre = quad_any(__ballot(act && (v.x == TBK_EMPTY || v.y == TBK_EMPTY)));
        found |= hit;
        pending &= ~(hit | fre);
    }
    return found;
}

template <bool MULTI>
__device__ __forceinline__ void probe_pass(const ProbeArgs &p, const uint64_t e0, const uint64_t e1,
                                           const uint64_t e2, const uint64_t P0, const uint64_t r_first,
                                           const uint32_t lane) {
    const int k = p.k;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t sub = lane & 3u;
    // forward stream S = bases 0..47 of this lane, reverse-complement stream R = rc(S)
    uint32_t s0 = (uint32_t)e0, s1 = (uint32_t)e1, s2 = (uint32_t)e2;
    uint32_t t0 = rev_pairs(~s2), t1 = rev_pairs(~s1), t2 = rev_pairs(~s0);
    // window j's rc k-mer = bits [96-2j-2k, 96-2j) of R; after j left shifts by 2 it sits at
    // [96-2k, 96): keep R shifted so the slice position is constant.
    const uint64_t bad48 = (e0 >> 32) | ((e1 >> 32) << 16) | ((e2 >> 32) << 32);
    const uint64_t badk = k == 32 ? 0xFFFFFFFFull : ((1ull << k) - 1ull);
    const int rsh = 96 - 2 * k;  // 32..94

    // read bookkeeping
    const uint64_t p_lane = P0 + (uint64_t)lane * TBK_WPL;
    uint64_t rid = r_first;
    uint64_t rend;
    if (MULTI) {
        // this lane's first window start may be in a later read than the pass start
        uint64_t pl = p_lane < p.total ? p_lane : p.total;
        rid = find_read(p.offsets, p.n_reads, pl);
        rend = rid < p.n_reads ? p.offsets[rid + 1] : p.total;
    } else {
        rend = r_first < p.n_reads ? p.offsets[r_first + 1] : p.total;
    }
    uint32_t acc_a = 0, acc_b = 0;  // wave-uniform in the single-read case

#pragma unroll 2
    for (int j = 0; j < TBK_WPL; j++) {
        // ---- this lane's window j ---------------------------------------------------
        const uint64_t fwd = ((uint64_t)s0 | ((uint64_t)s1 << 32)) & kmask;
        // bits [rsh, rsh+2k) of (t0,t1,t2)
        uint64_t rc;
        {
            const unsigned __int128 R = (unsigned __int128)t0 | ((unsigned __int128)t1 << 32) |
                                        ((unsigned __int128)t2 << 64);
            rc = (uint64_t)(R >> rsh) & kmask;
        }
        const uint64_t key = fwd < rc ? fwd : rc;
        const uint64_t pw = p_lane + (uint64_t)j;
        if (MULTI) {
            while (rid < p.n_reads && pw >= rend) { rid++; rend = rid < p.n_reads ? p.offsets[rid + 1] : p.total; }
        }
        const bool ok = ((bad48 >> j) & badk) == 0 && pw + (uint64_t)k <= rend && rid < p.n_reads;
        const uint32_t my_bk = ok ? tbk_home_bucket(key, p.t.n_buckets) : 0u;
        const uint32_t my_klo = (uint32_t)key, my_khi = (uint32_t)(key >> 32);
        const uint32_t my_ok = ok ? 1u : 0u;
        const uint32_t my_rid = (uint32_t)rid;

        // advance the streams to window j+1: S >>= 2, R <<= 2
        s0 = (s0 >> 2) | (s1 << 30); s1 = (s1 >> 2) | (s2 << 30); s2 >>= 2;
        t2 = (t2 << 2) | (t1 >> 30); t1 = (t1 << 2) | (t0 >> 30); t0 <<= 2;

        // ---- four quad sub-steps: fetch both lines of each of the quad's 4 windows -----
        uint32_t klo[4], khi[4], bk[4], okq[4], ridq[4];
        ulonglong2 va[4], vb[4];
#define TBK_BCAST(S)                                                        \
        klo[S] = quad_bcast<S>(my_klo); khi[S] = quad_bcast<S>(my_khi);     \
        bk[S] = quad_bcast<S>(my_bk);   okq[S] = quad_bcast<S>(my_ok);      \
        if (MULTI) ridq[S] = quad_bcast<S>(my_rid);
        TBK_BCAST(0) TBK_BCAST(1) TBK_BCAST(2) TBK_BCAST(3)
#undef TBK_BCAST
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint64_t *line = p.t.slots + (uint64_t)bk[s] * 16 + sub * 2;
            va[s] = *reinterpret_cast<const ulonglong2 *>(line);
            vb[s] = *reinterpret_cast<const ulonglong2 *>(line + 8);
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint64_t kk = (uint64_t)klo[s] | ((uint64_t)khi[s] << 32);
            const bool okl = okq[s] != 0;
            const uint64_t okm = quad_any(__ballot(okl));
            uint64_t hit_a = quad_any(__ballot(va[s].x == kk || va[s].y == kk)) & okm;
            const uint64_t fre_a = quad_any(__ballot(va[s].x == TBK_EMPTY || va[s].y == TBK_EMPTY));
            uint64_t hit_b = quad_any(__ballot(vb[s].x == kk || vb[s].y == kk)) & okm;
            const uint64_t fre_b = quad_any(__ballot(vb[s].x == TBK_EMPTY || vb[s].y == TBK_EMPTY));
            const uint64_t more_a = okm & ~hit_a & ~fre_a;
            if (more_a) hit_a |= probe_walk(p.t, 0, kk, bk[s], more_a, sub);
            const uint64_t more_b = okm & ~hit_a & ~hit_b & ~fre_b;
            if (more_b) hit_b |= probe_walk(p.t, 8, kk, bk[s], more_b, sub);
            hit_b &= ~hit_a;  // hapA wins (c/kmers.c:291-294)
            if (!MULTI) {
                acc_a += (uint32_t)__popcll(hit_a);
                acc_b += (uint32_t)__popcll(hit_b);
            } else if (hit_a | hit_b) {
                const uint64_t me = 1ull << lane;  // quad lane 0 carries the quad's bit
                if (hit_a & me) atomicAdd(&p.counts[2 * (uint64_t)ridq[s]], 1);
                if (hit_b & me) atomicAdd(&p.counts[2 * (uint64_t)ridq[s] + 1], 1);
            }
        }
    }
    if (!MULTI) {
        if (lane == 0) {
            if (acc_a) atomicAdd(&p.counts[2 * r_first], (int)acc_a);
            if (acc_b) atomicAdd(&p.counts[2 * r_first + 1], (int)acc_b);
        }
    }
}

__global__ void __launch_bounds__(64 * TBK_WAVES_PER_BLOCK)
tbk_probe_kernel(const ProbeArgs p) {
    __shared__ uint64_t stage[TBK_WAVES_PER_BLOCK][TBK_CHUNKS + 2];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t passes_per_iter = (uint64_t)gridDim.x * TBK_WAVES_PER_BLOCK;
    const uint64_t n_iter = (p.n_passes + passes_per_iter - 1) / passes_per_iter;

    for (uint64_t it = 0; it < n_iter; it++) {
        const uint64_t pass = it * passes_per_iter + (uint64_t)blockIdx.x * TBK_WAVES_PER_BLOCK + wave;
        const bool live = pass < p.n_passes;
        const uint64_t P0 = pass * TBK_PASS;
        if (live) {
            stage[wave][lane] = load_chunk(p.bases, P0 + (uint64_t)lane * 16, p.total);
            if (lane < 2) stage[wave][64 + lane] = load_chunk(p.bases, P0 + (uint64_t)(64 + lane) * 16, p.total);
        }
        __syncthreads();
        if (live) {
            const uint64_t e0 = stage[wave][lane], e1 = stage[wave][lane + 1], e2 = stage[wave][lane + 2];
            // which read(s) does this pass touch?  (wave-uniform)
            const uint64_t r_first = find_read(p.offsets, p.n_reads, P0);
            const uint64_t last_pos = (P0 + TBK_PASS - 1 < p.total ? P0 + TBK_PASS - 1 : p.total - 1);
            const uint64_t r_end = r_first < p.n_reads ? p.offsets[r_first + 1] : p.total;
            if (last_pos < r_end) probe_pass<false>(p, e0, e1, e2, P0, r_first, lane);
            else probe_pass<true>(p, e0, e1, e2, P0, r_first, lane);
        }
        __syncthreads();
    }
}

// =======================================================================================
// launchers (called from tbk_host.cpp)
// =======================================================================================
extern "C" hipError_t tbk_launch_insert(uint64_t *slots, uint32_t n_buckets, uint32_t stride, uint32_t half,
                                        const uint64_t *d_keys, uint64_t n, unsigned long long *d_distinct,
                                        int *d_failed, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(tbk_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slots, n_buckets, stride,
                       half, d_keys, n, d_distinct, d_failed);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_contains(TbkTableView t, const uint64_t *d_keys, uint64_t n,
                                          uint8_t *d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tbk_contains_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t,
                       d_keys, n, d_out);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_probe(const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                                       uint64_t total, TbkPairView t, int k, int32_t *d_counts, int max_blocks,
                                       hipStream_t stream) {
    if (total == 0 || n_reads == 0) return hipSuccess;
    ProbeArgs p;
    p.bases = d_bases; p.offsets = d_offsets; p.n_reads = n_reads; p.total = total;
    p.n_passes = (total + TBK_PASS - 1) / TBK_PASS;
    p.t = t; p.k = k; p.counts = d_counts;
    uint64_t blocks = (p.n_passes + TBK_WAVES_PER_BLOCK - 1) / TBK_WAVES_PER_BLOCK;
    if (max_blocks > 0 && blocks > (uint64_t)max_blocks) blocks = (uint64_t)max_blocks;
    hipLaunchKernelGGL(tbk_probe_kernel, dim3((unsigned)blocks), dim3(64 * TBK_WAVES_PER_BLOCK), 0, stream, p);
    return hipGetLastError();
}
