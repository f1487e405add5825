// tbk_count_kernels.hip — the MI355X kernels of the find-unique-kmers step: k-mer counting
// into a table in HBM, its histogram, the A-minus-B selection.  Host side: tbk_count.cpp.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "tbk_common.h"
#include "tbk_device.h"

// =======================================================================================
// k-mer counting (the find-unique-kmers step)
// =======================================================================================
// The reference shells out to KMC (find_unique_kmers.py:62-233): count canonical k-mers of a read
// set, keep those seen at least twice (kmc's default -ci2), cap counters at 255 (-cs255), take a
// histogram, subtract the other parent's database and dump the k-mers whose counter lies between
// two cut-offs.  Here the database is a table in HBM: 64-byte lines of 8 keys with a parallel
// array of 32-bit counters, bucket chosen like the classifier's (minimizer of the k-mer, so the
// consecutive windows of a read update the same line while it sits in L2), probe sequence
// tbk_next_bucket.

// Copy reads that lie back to back into a stream where every read is followed by one 'N', upper-
// casing on the way (KMC counts lower-case bases like upper-case ones): a window can then never
// span two reads and validity is the not-ACGT mask alone.  One wave per read.
__global__ void __launch_bounds__(256)
tbk_separate_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, uint64_t n_reads,
                    uint8_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < n_reads; r += waves) {
        const uint64_t lo = offsets[r], hi = offsets[r + 1];
        uint8_t *dst = out + lo + r;
        for (uint64_t i = lo + lane; i < hi; i += 64) dst[i - lo] = bases[i] & 0xDFu;
        if (lane == 0) dst[hi - lo] = 'N';
    }
}

// bucket b = one 128-byte line: 8 keys (TBK_EMPTY = free), then their 8 32-bit counters, then 32
// spare bytes - keys and counters of a bucket arrive with one HBM line and the increments of a run
// of windows land in a line that already sits in L2
constexpr int TBK_COUNT_LINE = 16;  // uint64 words per bucket line
struct TbkCountView {
    uint64_t *lines;    // n_buckets * 16 words
    uint32_t n_buckets;
    TbkMz mz;
    __device__ __forceinline__ unsigned long long *keys(uint32_t b) const { return (unsigned long long *)(lines + (uint64_t)b * TBK_COUNT_LINE); }
    __device__ __forceinline__ uint32_t *counts(uint32_t b) const { return (uint32_t *)(lines + (uint64_t)b * TBK_COUNT_LINE + TBK_SLOTS_PER_BUCKET); }
};

// A probe sequence longer than this means the table is as good as full (the host grows the table
// long before: tbk_count.cpp); giving up keeps a mis-sized table from turning into an endless walk.
constexpr uint32_t TBK_COUNT_MAX_WALK = 1u << 12;

// find or claim the key's slot along its probe sequence, from bucket b on, and add `n` occurrences;
// `claimed` counts the slots newly taken
__device__ __forceinline__ bool count_from(const TbkCountView &t, uint64_t key, uint32_t b, bool at_home, uint32_t n, uint32_t &claimed) {
    for (uint32_t walked = 0; walked <= t.n_buckets && walked < TBK_COUNT_MAX_WALK; walked++) {
        unsigned long long *line = t.keys(b);
        for (int s = 0; s < TBK_SLOTS_PER_BUCKET; s++) {
            unsigned long long cur = __hip_atomic_load(&line[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == TBK_EMPTY) {
                cur = atomicCAS(&line[s], (unsigned long long)TBK_EMPTY, (unsigned long long)key);
                if (cur == TBK_EMPTY) { cur = key; claimed++; }
            }
            if (cur == key) { atomicAdd(&t.counts(b)[s], n); return true; }
        }
        b = tbk_next_bucket(key, t.mz, t.n_buckets, b, at_home && walked == 0);
    }
    return false;
}

// One wave per pass of 2048 window starts of the separated stream, staged and rolled like the probe
// kernel's; every clean window counts its canonical k-mer.  A lane keeps the 8 keys of the bucket
// of its previous window in registers: consecutive windows mostly share their minimizer, so the
// line is fetched once per run and a window costs its compares and one fire-and-forget atomic add.
// The copy may be stale - other lanes insert meanwhile - but only in one direction: a slot seen
// occupied never changes, and a slot seen free is claimed with a compare-and-swap that returns what
// is really there.
template <int W, bool M64>
__global__ void __launch_bounds__(64)
tbk_count_kernel(const uint8_t *__restrict__ bases, uint64_t total, uint64_t first_pass, uint64_t n_passes, int k, TbkCountView t,
                 int *__restrict__ failed, unsigned long long *__restrict__ used) {
    __shared__ uint64_t stage[TBK_CHUNKS + 2];
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t badk = k == 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
    using win_t = typename std::conditional<M64, uint64_t, uint32_t>::type;
    const int m = t.mz.m, o = t.mz.o;
    const uint64_t mmask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
    auto mmer_order = [&](uint64_t fwd64, uint64_t rc64, uint32_t fsh, uint32_t bsh) -> win_t {
        if (M64) {
            const uint64_t x = (fwd64 >> fsh) & mmask, y = (rc64 >> bsh) & mmask;
            return (win_t)tbk_mmer_hash64(x < y ? x : y);
        }
        const uint32_t x = (uint32_t)(fwd64 >> fsh) & (uint32_t)mmask, y = (uint32_t)(rc64 >> bsh) & (uint32_t)mmask;
        return (win_t)tbk_mmer_hash(x < y ? x : y);
    };
    for (uint64_t pass = first_pass + blockIdx.x; pass < first_pass + n_passes; pass += gridDim.x) {
        const uint64_t P0 = pass * TBK_PASS;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        stage[lane] = load_chunk(bases, P0 + (uint64_t)lane * 16, total);
        stage[64 + lane] = load_chunk(bases, P0 + (uint64_t)(64 + lane) * 16, total);
        if (lane < 2) stage[128 + lane] = load_chunk(bases, P0 + (uint64_t)(128 + lane) * 16, total);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint64_t e0 = stage[2 * lane], e1 = stage[2 * lane + 1], e2 = stage[2 * lane + 2], e3 = stage[2 * lane + 3];
        uint32_t s0 = (uint32_t)e0, s1 = (uint32_t)e1, s2 = (uint32_t)e2, s3 = (uint32_t)e3;
        const unsigned __int128 R128 = (unsigned __int128)rev_pairs(~s3) | ((unsigned __int128)rev_pairs(~s2) << 32) |
                                       ((unsigned __int128)rev_pairs(~s1) << 64) | ((unsigned __int128)rev_pairs(~s0) << 96);
        const unsigned __int128 Rs = R128 >> (64 - 2 * k);
        uint32_t t0 = (uint32_t)Rs, t1 = (uint32_t)(Rs >> 32), t2 = (uint32_t)(Rs >> 64), t3 = (uint32_t)(Rs >> 96);
        uint32_t bad_lo = (uint32_t)(e0 >> 32) | ((uint32_t)(e1 >> 32) << 16);
        uint32_t bad_hi = (uint32_t)(e2 >> 32) | ((uint32_t)(e3 >> 32) << 16);
        // minimizer state: the hashes of the span's W m-mers (see probe_pass)
        win_t win[W > 0 ? W : 1];
        uint32_t fsh_new = 0, bsh_new = 0;
        if (W > 0) {
            const uint64_t fs = ((uint64_t)s1 << 32) | s0, bs = ((uint64_t)t3 << 32) | t2;
            win[0] = (win_t)~0ull;
#pragma unroll
            for (int i = 0; i + 1 < W; i++) win[i + 1] = mmer_order(fs, bs, (uint32_t)(2 * (o + i)), (uint32_t)(2 * (o + W - 1 - i)));
            fsh_new = (uint32_t)(2 * (o + W - 1));
            bsh_new = (uint32_t)(2 * o);
        }
        uint64_t held[TBK_SLOTS_PER_BUCKET];  // keys of bucket held_bk as last seen
#pragma unroll
        for (int s = 0; s < TBK_SLOTS_PER_BUCKET; s++) held[s] = 0;
        uint32_t held_bk = 0xFFFFFFFFu;
        uint32_t claimed = 0;  // slots this lane took for new k-mers
        uint32_t adds = 0;     // 64-bit atomic adds this lane issued (bench.py prices the kernel against the chip's rate for THOSE)
        bool full = false;
        // The adds of consecutive windows are merged where they can be: a bucket's eight 32-bit counters
        // are four 64-bit words, the k-mers of a run of windows were inserted one after the other - so
        // they mostly sit in neighbouring slots - and one 64-bit add of (1 | 1 << 32) counts both
        // halves of a word (a counter would have to pass 2^32 to carry into its neighbour: the host keeps
        // every counter below that, tbk_count_clamp_kernel; readers cap at 255).  The lane keeps the adds to the four words of ONE bucket pending and sends them when
        // a window counts in another bucket: the kernel runs at the rate the chip executes atomic adds,
        // so fewer adds is the lever.
        uint32_t pend_bk = 0xFFFFFFFFu;
        unsigned long long pend[4] = {0, 0, 0, 0};
        auto flush = [&]() {
            if (pend_bk == 0xFFFFFFFFu) return;
            unsigned long long *words = reinterpret_cast<unsigned long long *>(t.counts(pend_bk));
#pragma unroll
            for (int w = 0; w < 4; w++)
                if (pend[w]) { atomicAdd(&words[w], pend[w]); pend[w] = 0; adds++; }
        };
#pragma unroll 2
        for (int j = 0; j < TBK_WPL; j++) {
            const uint64_t fwd = ((uint64_t)s0 | ((uint64_t)s1 << 32)) & kmask;
            const uint64_t rc = ((uint64_t)t2 | ((uint64_t)t3 << 32)) & kmask;
            const uint64_t key = fwd < rc ? fwd : rc;
            const bool ok = (bad_lo & badk) == 0 && P0 + (uint64_t)lane * TBK_WPL + (uint64_t)j + (uint64_t)k <= total;
            uint32_t hsel;
            if (W > 0) {
#pragma unroll
                for (int i = 0; i + 1 < W; i++) win[i] = win[i + 1];
                win[W - 1] = mmer_order(((uint64_t)s1 << 32) | s0, ((uint64_t)t3 << 32) | t2, fsh_new, bsh_new);
                win_t best = win[0];
#pragma unroll
                for (int i = 1; i < W; i++) best = win[i] < best ? win[i] : best;
                hsel = M64 ? (uint32_t)best : tbk_scramble((uint32_t)best);
            } else {
                hsel = tbk_mix32(key);
            }
            if (ok) {
                // home bucket first, then along the probe sequence; every bucket visited is fetched with
                // four 16-byte loads in flight at once and scanned in registers
                uint32_t b = tbk_reduce(hsel, t.n_buckets);
                bool done = false;
                for (uint32_t walked = 0; !done && walked < TBK_COUNT_MAX_WALK; walked++) {
                    unsigned long long *line = t.keys(b);
                    if (b != held_bk) {
                        const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(line);
                        const ulonglong2 v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
                        held[0] = v0.x; held[1] = v0.y; held[2] = v1.x; held[3] = v1.y;
                        held[4] = v2.x; held[5] = v2.y; held[6] = v3.x; held[7] = v3.y;
                        held_bk = b;
                    }
#pragma unroll
                    for (int s = 0; s < TBK_SLOTS_PER_BUCKET; s++) {
                        if (done) continue;
                        if (held[s] == TBK_EMPTY) {
                            const unsigned long long old = atomicCAS(&line[s], (unsigned long long)TBK_EMPTY, (unsigned long long)key);
                            held[s] = old == TBK_EMPTY ? key : old;
                            claimed += old == TBK_EMPTY ? 1u : 0u;
                        }
                        if (held[s] == key) {
                            if (b != pend_bk) { flush(); pend_bk = b; }
                            pend[s >> 1] += 1ull << (32 * (s & 1));
                            done = true;
                        }
                    }
                    if (!done) b = tbk_next_bucket(key, t.mz, t.n_buckets, b, walked == 0);
                }
                if (!done) full = true;
            }
            s0 = (s0 >> 2) | (s1 << 30); s1 = (s1 >> 2) | (s2 << 30); s2 = (s2 >> 2) | (s3 << 30); s3 >>= 2;
            t3 = (t3 << 2) | (t2 >> 30); t2 = (t2 << 2) | (t1 >> 30); t1 = (t1 << 2) | (t0 >> 30); t0 <<= 2;
            bad_lo = (bad_lo >> 1) | (bad_hi << 31); bad_hi >>= 1;
        }
        flush();
        if (full) atomicExch(failed, 1);
        // slots taken by this wave: one atomic per pass
        uint32_t sum = claimed;
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0 && sum) atomicAdd(used, (unsigned long long)sum);
        uint32_t asum = adds;
        for (int d = 32; d > 0; d >>= 1) asum += __shfl_xor(asum, d);
        if (lane == 0 && asum) atomicAdd(used + 1 + (pass & 1023u), (unsigned long long)asum);  // (1024 tallies: every pass adding to ONE word cost 3 ms per launch - the chip's rate for atomics on one address)
    }
}

// Counters are 32 bits wide and neighbours share a 64-bit word that the counting kernel adds to in one piece: a counter
// that passed 2^32 would carry into its neighbour.  Readers cap at 255, so a counter may stop anywhere above that: the host
// runs this pass before the window starts added since the last one could take any counter from 2^31 to 2^32 - every
// counter above 2^31 is set back to 2^31 (saturation, not a carry).
__global__ void __launch_bounds__(256)
tbk_count_clamp_kernel(TbkCountView t) {
    const uint64_t n = (uint64_t)t.n_buckets * TBK_SLOTS_PER_BUCKET;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t *c = &t.counts((uint32_t)(i / TBK_SLOTS_PER_BUCKET))[i % TBK_SLOTS_PER_BUCKET];
        if (*c > 0x80000000u) *c = 0x80000000u;
    }
}

extern "C" hipError_t tbk_launch_count_clamp(uint64_t *d_lines, uint32_t n_buckets, TbkMz mz, hipStream_t stream) {
    hipLaunchKernelGGL(tbk_count_clamp_kernel, dim3(4096), dim3(256), 0, stream, TbkCountView{d_lines, n_buckets, mz});
    return hipGetLastError();
}

// Move every (key, counter) of an old table into a new, larger one (the host grows the table when
// the next batch could fill it).
__global__ void __launch_bounds__(256)
tbk_count_rehash_kernel(TbkCountView from, TbkCountView to, int *__restrict__ failed) {
    const uint64_t n_slots = (uint64_t)from.n_buckets * TBK_SLOTS_PER_BUCKET;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += step) {
        const uint32_t b = (uint32_t)(i >> 3), s = (uint32_t)(i & 7);
        const uint64_t key = from.keys(b)[s];
        if (key == TBK_EMPTY) continue;
        uint32_t claimed = 0;
        if (!count_from(to, key, tbk_bucket_of(key, to.mz, to.n_buckets), true, from.counts(b)[s], claimed)) atomicExch(failed, 1);
    }
}

// hist[c] = k-mers whose counter, capped at 255, equals c (c = 1..255); hist[0] = occupied slots
__global__ void __launch_bounds__(256)
tbk_count_histogram_kernel(TbkCountView t, unsigned long long *__restrict__ hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t n_slots = (uint64_t)t.n_buckets * TBK_SLOTS_PER_BUCKET;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += step) {
        const uint32_t b = (uint32_t)(i >> 3), s = (uint32_t)(i & 7);
        if (t.keys(b)[s] == TBK_EMPTY) continue;
        const uint32_t raw = t.counts(b)[s], c = raw < 255u ? raw : 255u;
        atomicAdd(&h[c], 1u);
        atomicAdd(&h[0], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// counter of `key` in a counting table (0 if absent).  Keys are never removed and take the first
// free slot along their probe sequence, so a line with a free slot ends the search.
__device__ __forceinline__ uint32_t count_lookup(const TbkCountView &t, uint64_t key) {
    uint32_t b = tbk_bucket_of(key, t.mz, t.n_buckets);
    for (uint32_t walked = 0; walked <= t.n_buckets && walked < TBK_COUNT_MAX_WALK; walked++) {
        const unsigned long long *line = t.keys(b);
        for (int s = 0; s < TBK_SLOTS_PER_BUCKET; s++) {
            const uint64_t cur = line[s];
            if (cur == key) return t.counts(b)[s];
            if (cur == TBK_EMPTY) return 0;
        }
        b = tbk_next_bucket(key, t.mz, t.n_buckets, b, walked == 0);
    }
    return 0;
}

// kmc_tools simple A B kmers_subtract + kmc_dump -ci -cx: the k-mers of database A (counter >= 2)
// that database B does not hold (its counter < 2) and whose counter, capped at 255, lies in
// [ci, cx].  Each is appended as its lexicographic rank (base 0 in the top bits), ready to sort.
__global__ void __launch_bounds__(256)
tbk_count_unique_kernel(TbkCountView a, TbkCountView b, int k, uint32_t ci, uint32_t cx, uint64_t *__restrict__ out,
                        uint64_t capacity, unsigned long long *__restrict__ n_out) {
    const uint64_t n_slots = (uint64_t)a.n_buckets * TBK_SLOTS_PER_BUCKET;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    // whole waves iterate together: the append below is a wave operation
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x - lane; i0 < n_slots; i0 += step) {
        const uint64_t i = i0 + lane;
        bool emit = false;
        uint64_t key = 0;
        if (i < n_slots) {
            key = a.keys((uint32_t)(i >> 3))[i & 7];
            if (key != TBK_EMPTY) {
                const uint32_t raw = a.counts((uint32_t)(i >> 3))[i & 7], c = raw < 255u ? raw : 255u;
                emit = raw >= 2u && c >= ci && c <= cx && count_lookup(b, key) < 2u;
            }
        }
        const uint64_t mask = __builtin_amdgcn_ballot_w64(emit);
        if (mask) {
            unsigned long long base = 0;
            const int leader = __builtin_ctzll(mask);
            if ((int)lane == leader) base = atomicAdd(n_out, (unsigned long long)__popcll(mask));
            base = __shfl(base, leader);
            if (emit) {
                const uint64_t at = base + (uint64_t)__popcll(mask & ((1ull << lane) - 1ull));
                if (at < capacity) {
                    const uint64_t lex = ((uint64_t)rev_pairs((uint32_t)key) << 32) | (uint64_t)rev_pairs((uint32_t)(key >> 32));
                    out[at] = lex >> (64 - 2 * k);
                }
            }
        }
    }
}

// =======================================================================================
// launchers (called from tbk_count.cpp)
// =======================================================================================
extern "C" hipError_t tbk_launch_separate(const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint8_t *d_out,
                                          hipStream_t stream) {
    if (!n_reads) return hipSuccess;
    uint64_t blocks = (n_reads + 3) / 4;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(tbk_separate_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_bases, d_offsets, n_reads, d_out);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_count_rehash(uint64_t *from_lines, uint32_t from_buckets, TbkMz from_mz, uint64_t *to_lines,
                                              uint32_t to_buckets, TbkMz to_mz, int *d_failed, hipStream_t stream) {
    hipLaunchKernelGGL(tbk_count_rehash_kernel, dim3(8192), dim3(256), 0, stream, TbkCountView{from_lines, from_buckets, from_mz},
                       TbkCountView{to_lines, to_buckets, to_mz}, d_failed);
    return hipGetLastError();
}

// passes [first_pass, first_pass + n_passes) of the separated stream (tbk_probe_passes(total) in all)
extern "C" hipError_t tbk_launch_count(const uint8_t *d_sep, uint64_t total, uint64_t first_pass, uint64_t n_passes, int k,
                                       uint64_t *d_lines, uint32_t n_buckets, TbkMz mz, int *d_failed, unsigned long long *d_used,
                                       hipStream_t stream) {
    if (total < (uint64_t)k || !n_passes) return hipSuccess;
    const uint64_t blocks = n_passes < (1u << 20) ? n_passes : (1u << 20);
    const TbkCountView view{d_lines, n_buckets, mz};
    const dim3 grid((unsigned)blocks), block(64);
    const bool m64 = mz.m > 16;
#define TBK_COUNT_LAUNCH(N) case N: if (m64) hipLaunchKernelGGL((tbk_count_kernel<N, true>), grid, block, 0, stream, d_sep, total, first_pass, n_passes, k, view, d_failed, d_used); \
                                    else hipLaunchKernelGGL((tbk_count_kernel<N, false>), grid, block, 0, stream, d_sep, total, first_pass, n_passes, k, view, d_failed, d_used); break;
    switch (mz.t > 0 ? -1 : mz.w) {
        case 0: hipLaunchKernelGGL((tbk_count_kernel<0, false>), grid, block, 0, stream, d_sep, total, first_pass, n_passes, k, view, d_failed, d_used); break;
        TBK_COUNT_LAUNCH(1) TBK_COUNT_LAUNCH(2) TBK_COUNT_LAUNCH(3) TBK_COUNT_LAUNCH(4)
        TBK_COUNT_LAUNCH(5) TBK_COUNT_LAUNCH(6) TBK_COUNT_LAUNCH(7) TBK_COUNT_LAUNCH(8)
        default: return hipErrorInvalidValue;
    }
#undef TBK_COUNT_LAUNCH
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_count_histogram(uint64_t *d_lines, uint32_t n_buckets, TbkMz mz, unsigned long long *d_hist,
                                                 hipStream_t stream) {
    hipLaunchKernelGGL(tbk_count_histogram_kernel, dim3(4096), dim3(256), 0, stream, TbkCountView{d_lines, n_buckets, mz}, d_hist);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_count_unique(uint64_t *a_lines, uint32_t a_buckets, TbkMz a_mz, uint64_t *b_lines, uint32_t b_buckets,
                                              TbkMz b_mz, int k, uint32_t ci, uint32_t cx, uint64_t *d_out, uint64_t capacity,
                                              unsigned long long *d_n, hipStream_t stream) {
    hipLaunchKernelGGL(tbk_count_unique_kernel, dim3(8192), dim3(256), 0, stream, TbkCountView{a_lines, a_buckets, a_mz},
                       TbkCountView{b_lines, b_buckets, b_mz}, k, ci, cx, d_out, capacity, d_n);
    return hipGetLastError();
}
