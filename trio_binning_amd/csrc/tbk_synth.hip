// tbk_synth.hip — synthetic workload generators and roofline-calibration kernels.
//
// These produce BASELINE.json's bench inputs on the GPU: deterministic
// distinct canonical k-mer lists and random reads with planted list k-mers, so that the
// timed region of bench.py starts with its inputs already resident in HBM.  They are
// workload generators, not part of the reference's path; the tests copy what they generate
// back to the host and check the probe kernel's counts on it against the CPU checker.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tbk_common.h"

// ---------------------------------------------------------------------------------------
// keys
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
tbk_synth_keys_kernel(uint64_t seed, uint64_t first, uint64_t n, int k, uint64_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = tbk_synth_key(seed, first + i, k);
}

// ---------------------------------------------------------------------------------------
// reads: background
// ---------------------------------------------------------------------------------------
// One thread writes 16 bases (one 16-byte store): 32 random bits from a counter-based
// generator keyed on (seed, absolute chunk index), so any sub-range can be regenerated.
__global__ void __launch_bounds__(256)
tbk_synth_background_kernel(uint64_t seed, uint64_t first_chunk, uint64_t n_chunks, uint4 *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_chunks; i += stride) {
        const uint64_t r = tbk_splitmix(seed ^ ((first_chunk + i) * 0xD6E8FEB86659FD93ull));
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t code = (uint32_t)(r >> (2 * (4 * q + b))) & 3u;
                // "ACGT"[code]
                const uint32_t ch = (0x54474341u >> (8 * code)) & 0xFFu;
                word |= ch << (8 * b);
            }
            w[q] = word;
        }
        out[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__global__ void __launch_bounds__(256)
tbk_synth_offsets_kernel(uint64_t n_reads, uint32_t read_len, uint64_t *__restrict__ offsets) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_reads) offsets[i] = i * (uint64_t)read_len;
}

// ---------------------------------------------------------------------------------------
// reads: planting
// ---------------------------------------------------------------------------------------
// One thread per (read, plant slot).  The read is cut into n_slots equal slots; plant j
// goes at a random offset inside slot j (never crossing into the next slot), so plants
// never overlap.  Read origin: A / B / none with p = .45 / .45 / .10.  Origin reads carry
// `major` k-mers of their own list in the first `major` slots and `minor` of the other;
// origin-less reads carry `minor` of each.  Each planted k-mer is written forward or
// reverse-complemented with p = .5.
__global__ void __launch_bounds__(256)
tbk_synth_plant_kernel(uint64_t read_seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                       uint64_t key_seed, uint64_t n_a, uint64_t n_b, int k, int major, int minor,
                       uint8_t *__restrict__ bases) {
    const int n_slots = major + minor;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_reads * (uint64_t)n_slots) return;
    const uint64_t r = t / n_slots;
    const int j = (int)(t % n_slots);
    const uint32_t slot_len = read_len / n_slots;
    if (slot_len < (uint32_t)k) return;
    const uint64_t rr = tbk_splitmix(read_seed ^ ((first_read + r) * 0xA0761D6478BD642Full));
    const uint32_t u = (uint32_t)(rr % 100u);
    const int origin = u < 45 ? 0 : (u < 90 ? 1 : 2);  // 0=A 1=B 2=none
    int which;  // list of this plant
    if (origin == 2) {
        if (j >= 2 * minor) return;
        which = j < minor ? 0 : 1;
    } else {
        which = j < major ? origin : 1 - origin;
    }
    const uint64_t pr = tbk_splitmix(rr ^ ((uint64_t)(j + 1) * 0xE7037ED1A0B428DBull));
    const uint64_t n_list = which == 0 ? n_a : n_b;
    if (n_list == 0) return;
    const uint64_t idx = (which == 0 ? 0 : n_a) + (pr >> 8) % n_list;
    uint64_t key = tbk_synth_key(key_seed, idx, k);
    if (pr & 1) key = tbk_revcomp_packed(key, k);
    const uint32_t off = (uint32_t)((pr >> 40) % (slot_len - (uint32_t)k + 1u));
    uint8_t *dst = bases + r * (uint64_t)read_len + (uint64_t)j * slot_len + off;
    for (int i = 0; i < k; i++) {
        const uint32_t code = (uint32_t)(key >> (2 * i)) & 3u;
        dst[i] = (uint8_t)((0x54474341u >> (8 * code)) & 0xFFu);
    }
}

// ---------------------------------------------------------------------------------------
// haplotype-shaped lists and reads (tbk_hap_bases)
// ---------------------------------------------------------------------------------------
// One thread scans SEG consecutive k-mer start positions of the genome with rolling k-mers of
// both haplotypes and emits, for every window that covers a position where they differ, hapA's
// canonical k-mer to list A and hapB's to list B (same index: the lists have equal length).
// Appends are wave-aggregated (one atomic per wave-iteration).  Order is not deterministic, the
// sets are.
constexpr int TBK_HAP_SEG = 64;
__global__ void __launch_bounds__(256)
tbk_synth_hap_keys_kernel(uint64_t seed, uint64_t genome_len, uint32_t snp24, int k, uint64_t *__restrict__ out_a,
                          uint64_t *__restrict__ out_b, uint64_t capacity, unsigned long long *__restrict__ n_out) {
    const uint64_t n_windows = genome_len - (uint64_t)k + 1;
    const uint64_t n_seg = (n_windows + TBK_HAP_SEG - 1) / TBK_HAP_SEG;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t lane = threadIdx.x & 63u;
    // whole waves iterate together (the append is a wave operation): segment index per lane
    for (uint64_t seg0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x - lane); seg0 < n_seg; seg0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t seg = seg0 + lane;
        const uint64_t q0 = seg * TBK_HAP_SEG;
        uint64_t fa = 0, ra = 0, fb = 0, rb = 0;
        uint32_t since_diff = 0x7FFFFFFFu;  // positions since the haplotypes last differed
        for (int i = 0; i < TBK_HAP_SEG + k - 1; i++) {
            const uint64_t p = q0 + (uint64_t)i;
            uint32_t a = 0, b = 0;
            const bool in = seg < n_seg && p < genome_len;
            if (in) tbk_hap_bases(seed, p, snp24, a, b);
            // base i of a k-mer at bits 2i (LSB first): the window's newest base enters at the top
            fa = (fa >> 2) | ((uint64_t)a << (2 * (k - 1)));
            fb = (fb >> 2) | ((uint64_t)b << (2 * (k - 1)));
            ra = ((ra << 2) | (uint64_t)(3u - a)) & kmask;
            rb = ((rb << 2) | (uint64_t)(3u - b)) & kmask;
            since_diff = (in && a != b) ? 0u : (since_diff < 0x7FFFFFFFu ? since_diff + 1u : since_diff);
            const bool emit = in && i >= k - 1 && since_diff < (uint32_t)k;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(emit);
            if (mask) {
                unsigned long long base = 0;
                if (lane == (uint32_t)__builtin_ctzll(mask)) base = atomicAdd(n_out, (unsigned long long)__popcll(mask));
                base = __shfl(base, __builtin_ctzll(mask));
                if (emit) {
                    const uint64_t at = base + (uint64_t)__popcll(mask & ((1ull << lane) - 1ull));
                    if (at < capacity) {
                        out_a[at] = fa < ra ? fa : ra;
                        out_b[at] = fb < rb ? fb : rb;
                    }
                }
            }
        }
    }
}

// Reads drawn from the haplotypes: read r comes from haplotype r & 1, from a hashed start
// position, on a hashed strand, with substitution errors at err24 / 2^24 per base.  One thread
// writes 16 bases.
__global__ void __launch_bounds__(256)
tbk_synth_hap_reads_kernel(uint64_t seed, uint64_t genome_len, uint32_t snp24, uint64_t read_seed, uint64_t first_read,
                           uint64_t n_reads, uint32_t read_len, uint32_t err24, uint8_t *__restrict__ bases) {
    const uint64_t total = n_reads * (uint64_t)read_len;
    const uint64_t n_chunks = (total + 15) / 16;
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; c < n_chunks; c += stride) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16; j++) {
            const uint64_t pos = c * 16 + (uint64_t)j;
            uint32_t code = 0;
            if (pos < total) {
                const uint64_t r = pos / read_len, i = pos % read_len;
                const uint64_t rr = tbk_splitmix(read_seed ^ ((first_read + r) * 0xA0761D6478BD642Full));
                const uint64_t start = (rr >> 1) % (genome_len - (uint64_t)read_len + 1);
                const bool rev = rr & 1ull;
                const uint64_t p = rev ? start + (uint64_t)read_len - 1 - i : start + i;
                uint32_t a, b;
                tbk_hap_bases(seed, p, snp24, a, b);
                code = ((first_read + r) & 1ull) ? b : a;
                if (rev) code = 3u - code;
                const uint64_t e = tbk_splitmix(read_seed ^ 0x5851F42D4C957F2Dull ^ ((first_read * read_len + pos) * 0xD6E8FEB86659FD93ull));
                if (((uint32_t)e & 0xFFFFFFu) < err24) code = (code + 1u + (uint32_t)((e >> 32) % 3u)) & 3u;
            }
            w[j >> 2] |= ((0x54474341u >> (8 * code)) & 0xFFu) << (8 * (j & 3));
        }
        reinterpret_cast<uint4 *>(bases)[c] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------------------
// reads of any lengths (BASELINE configs[4]: log-normal lengths, N50 ~ 100 kb)
// ---------------------------------------------------------------------------------------
// The read whose bases hold stream position pos: largest r with offsets[r] <= pos (offsets[n_reads] = total > pos).
__device__ __forceinline__ uint64_t tbk_synth_find_read(const uint64_t *offsets, uint64_t n_reads, uint64_t pos) {
    uint64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// Planting into reads of any lengths: one thread per read walks the read's slots of `slot_len` bases (15 000 / 33: the
// density of tbk_synth_plant_kernel's 30 + 3 plants per 15 kb read); slot j of an origin read carries a k-mer of the
// read's own list, every 11th one of the other list; an origin-less read plants in every 8th slot, lists alternating.
__global__ void __launch_bounds__(256)
tbk_synth_plant_ragged_kernel(uint64_t read_seed, uint64_t first_read, uint64_t n_reads, const uint64_t *__restrict__ offsets,
                              uint64_t key_seed, uint64_t n_a, uint64_t n_b, int k, uint32_t slot_len, uint8_t *__restrict__ bases) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads || slot_len < (uint32_t)k) return;
    const uint64_t start = offsets[r], len = offsets[r + 1] - start;
    const uint64_t rr = tbk_splitmix(read_seed ^ ((first_read + r) * 0xA0761D6478BD642Full));
    const uint32_t u = (uint32_t)(rr % 100u);
    const int origin = u < 45 ? 0 : (u < 90 ? 1 : 2);  // 0=A 1=B 2=none
    for (uint64_t j = 0; (j + 1) * slot_len <= len; j++) {
        int which;
        if (origin == 2) {
            if (j % 8u) continue;
            which = (int)((j / 8u) & 1u);
        } else {
            which = j % 11u == 10u ? 1 - origin : origin;
        }
        const uint64_t n_list = which == 0 ? n_a : n_b;
        if (n_list == 0) continue;
        const uint64_t pr = tbk_splitmix(rr ^ ((j + 1) * 0xE7037ED1A0B428DBull));
        uint64_t key = tbk_synth_key(key_seed, (which == 0 ? 0 : n_a) + (pr >> 8) % n_list, k);
        if (pr & 1) key = tbk_revcomp_packed(key, k);
        uint8_t *dst = bases + start + j * slot_len + (uint32_t)((pr >> 40) % (slot_len - (uint32_t)k + 1u));
        for (int i = 0; i < k; i++) dst[i] = (uint8_t)((0x54474341u >> (8 * ((uint32_t)(key >> (2 * i)) & 3u))) & 0xFFu);
    }
}

// Haplotype reads of any lengths: read r comes from haplotype (first_read + r) & 1, a hashed start and strand (as
// tbk_synth_hap_reads_kernel).  One thread writes 16 bases; it finds the read of its first base by bisection and steps on
// from there.
__global__ void __launch_bounds__(256)
tbk_synth_hap_reads_ragged_kernel(uint64_t seed, uint64_t genome_len, uint32_t snp24, uint64_t read_seed, uint64_t first_read,
                                  uint64_t n_reads, const uint64_t *__restrict__ offsets, uint32_t err24, uint8_t *__restrict__ bases) {
    const uint64_t total = offsets[n_reads];
    const uint64_t n_chunks = (total + 15) / 16;
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; c < n_chunks; c += stride) {
        uint32_t w[4] = {0, 0, 0, 0};
        uint64_t r = tbk_synth_find_read(offsets, n_reads, c * 16 < total ? c * 16 : total - 1);
        uint64_t r_start = offsets[r], r_end = offsets[r + 1];
        for (int j = 0; j < 16; j++) {
            const uint64_t pos = c * 16 + (uint64_t)j;
            uint32_t code = 0;
            if (pos < total) {
                while (pos >= r_end) { r++; r_start = r_end; r_end = offsets[r + 1]; }  // (empty reads are stepped over)
                const uint64_t len = r_end - r_start, i = pos - r_start;
                const uint64_t rr = tbk_splitmix(read_seed ^ ((first_read + r) * 0xA0761D6478BD642Full));
                const uint64_t start = (rr >> 1) % (genome_len - len + 1);
                const bool rev = rr & 1ull;
                const uint64_t p = rev ? start + len - 1 - i : start + i;
                uint32_t a, b;
                tbk_hap_bases(seed, p, snp24, a, b);
                code = ((first_read + r) & 1ull) ? b : a;
                if (rev) code = 3u - code;
                const uint64_t e = tbk_splitmix(read_seed ^ 0x5851F42D4C957F2Dull ^ (((first_read + r) * 0x9E3779B97F4A7C15ull + i) * 0xD6E8FEB86659FD93ull));
                if (((uint32_t)e & 0xFFFFFFu) < err24) code = (code + 1u + (uint32_t)((e >> 32) % 3u)) & 3u;
            }
            w[j >> 2] |= ((0x54474341u >> (8 * code)) & 0xFFu) << (8 * (j & 3));
        }
        reinterpret_cast<uint4 *>(bases)[c] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------------------
// full-membership sweep: a list's own keys laid out as reads (tests/test_gpu_scale.py, bench.py --sweep)
// ---------------------------------------------------------------------------------------
// Key i of the list becomes k bases of a read - as it stands when i is even, reverse-complemented when odd (a lookup
// takes min(fwd, rc), c/kmers.c:251-255: both spellings must find the key).  per_read = 1: every key is a read of k bases
// (one window: the multi-read passes); per_read = P > 1: P keys to a read with an 'N' between neighbours (windows across
// two keys hold the N and score nothing: a read counts exactly its member keys - the single-read passes).
__global__ void __launch_bounds__(256)
tbk_synth_keys_as_reads_kernel(const uint64_t *__restrict__ keys, uint64_t first, uint64_t n, int k, uint32_t per_read, uint8_t *__restrict__ bases,
                               uint64_t *__restrict__ offsets, uint64_t n_reads) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t read_len = (uint64_t)per_read * (uint64_t)k + per_read - 1;
    if (i <= n_reads) offsets[i] = i < n_reads ? i * read_len : (n_reads - 1) * read_len + (n - (n_reads - 1) * per_read) * ((uint64_t)k + 1) - 1;
    if (i >= n) return;
    uint64_t key = keys[i];
    if ((first + i) & 1ull) key = tbk_revcomp_packed(key, k);
    const uint64_t r = i / per_read, j = i % per_read;
    uint8_t *dst = bases + r * read_len + j * ((uint64_t)k + 1);
    for (int b = 0; b < k; b++) dst[b] = (uint8_t)((0x54474341u >> (8 * ((uint32_t)(key >> (2 * b)) & 3u))) & 0xFFu);
    if (j + 1 < per_read && i + 1 < n) dst[k] = (uint8_t)'N';
}

// Near misses: key i with ONE base substituted (position and base hashed from seed and i), canonicalised - a non-member
// that shares its m-mer, position and most flank bits with a member: what a compressed slot could confuse it with.
__global__ void __launch_bounds__(256)
tbk_synth_mutate_keys_kernel(const uint64_t *__restrict__ keys, uint64_t first, uint64_t n, int k, uint64_t seed, uint64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = tbk_splitmix(seed ^ ((first + i) * 0xD6E8FEB86659FD93ull));
    const uint32_t p = (uint32_t)(h % (uint64_t)k);
    const uint64_t delta = 1ull + (h >> 32) % 3ull;
    uint64_t key = keys[i];
    const uint64_t base = (key >> (2 * p)) & 3ull;
    key = (key & ~(3ull << (2 * p))) | (((base + delta) & 3ull) << (2 * p));
    const uint64_t rc = tbk_revcomp_packed(key, k);
    out[i] = rc < key ? rc : key;
}

// what a read spelling key i must count: 1 = (1, 0) when hapA's list holds its canonical form, else 2 = (0, 1) when
// hapB's does, else 0 (in_a / in_b: tbk_table_contains_device of the canonical keys)
__global__ void __launch_bounds__(256)
tbk_sweep_expect_kernel(const uint8_t *__restrict__ in_a, const uint8_t *__restrict__ in_b, uint64_t n, uint8_t *__restrict__ expect) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) expect[i] = in_a[i] ? 1 : in_b[i] ? 2 : 0;
}

__global__ void __launch_bounds__(256)
tbk_canonical_keys_kernel(const uint64_t *__restrict__ keys, uint64_t n, int k, uint64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i], rc = tbk_revcomp_packed(key, k);
    out[i] = rc < key ? rc : key;
}

// counts[r] against what read r must count: its n_k keys (per_read, fewer in the last read) times (1, 0) / (0, 1) / (0, 0)
// for expect = 1 / 2 / 0, or per key from d_expect (a read of per_read keys must count how many of ITS keys say 1 and 2).  out: [0] sum of hapA counts, [1] of hapB counts,
// [2] reads that differ, [3] the first of them (first + r; all ones: none).
__global__ void __launch_bounds__(256)
tbk_counts_check_kernel(const int32_t *__restrict__ counts, uint64_t first, uint64_t n_reads, uint32_t per_read, uint64_t n_keys, int expect,
                        const uint8_t *__restrict__ d_expect, unsigned long long *__restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long a = 0, b = 0, bad = 0;
    if (r < n_reads) {
        const uint64_t nk = r + 1 < n_reads ? per_read : n_keys - r * per_read;
        a = (unsigned long long)counts[2 * r]; b = (unsigned long long)counts[2 * r + 1];
        unsigned long long want_a = expect == 1 ? nk : 0, want_b = expect == 2 ? nk : 0;
        if (d_expect) {  // per key: the read's keys are d_expect[r * per_read .. + nk) - 1 counts for hapA, 2 for hapB
            want_a = want_b = 0;
            const uint8_t *e = d_expect + r * per_read;
            for (uint64_t i = 0; i < nk; i++) { want_a += e[i] == 1; want_b += e[i] == 2; }
        }
        bad = (a != want_a || b != want_b) ? 1 : 0;
        if (bad) atomicMin(&out[3], (unsigned long long)(first + r));
    }
    // wave sums, one atomic per wave and counter
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); bad += __shfl_down(bad, off); }
    if ((threadIdx.x & 63u) == 0) {
        if (a) atomicAdd(&out[0], a);
        if (b) atomicAdd(&out[1], b);
        if (bad) atomicAdd(&out[2], bad);
    }
}

// ---------------------------------------------------------------------------------------
// calibration: random line gather and streaming read
// ---------------------------------------------------------------------------------------
// LPL lanes share one line (16 B per lane when LPL > 1; the whole line per lane when
// LPL == 1, as LINE/16 consecutive 16-byte loads).  Each lane group draws `iters` x INF
// independent random line indices; INF loads are in flight per lane before any is used.
template <int LINE, int LPL, int INF>
__global__ void __launch_bounds__(256)
tbk_calib_gather_kernel(const uint4 *__restrict__ buf, uint64_t n_lines_buf, uint32_t iters, uint64_t seed,
                        uint32_t *__restrict__ sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t group = gid / LPL;
    const uint32_t sub = (uint32_t)(gid % LPL);
    constexpr int PER_LINE = LINE / 16;           // 16-byte pieces per line
    constexpr int PIECES = PER_LINE / LPL;        // pieces each lane loads per line (LPL = 4 on 128-byte lines: the probe kernel's shape, two 64-byte halves)
    uint32_t acc = 0;
    uint64_t state = seed ^ (group * 0x9E3779B97F4A7C15ull);
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[INF][PIECES];
#pragma unroll
        for (int f = 0; f < INF; f++) {
            state = tbk_splitmix(state);
            const uint64_t line = (uint64_t)(((unsigned __int128)state * n_lines_buf) >> 64);
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) {
                const uint64_t piece = (uint64_t)sub + (uint64_t)pc * LPL;
                v[f][pc] = buf[line * PER_LINE + piece];
            }
        }
#pragma unroll
        for (int f = 0; f < INF; f++)
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) acc += v[f][pc].x ^ v[f][pc].y ^ v[f][pc].z ^ v[f][pc].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;  // keep the loads alive
}

__global__ void __launch_bounds__(256)
tbk_calib_stream_kernel(const uint4 *__restrict__ buf, uint64_t n_vec, uint32_t *__restrict__ sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + 3 * stride < n_vec; i += 4 * stride) {
        const uint4 a = buf[i], b = buf[i + stride], c = buf[i + 2 * stride], d = buf[i + 3 * stride];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n_vec; i += stride) { const uint4 a = buf[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// The gather in the probe kernels' own shape (round 5: the yardstick must not sit below what it measures): one-wave blocks -
// so that eight waves per SIMD fit - two lanes per line, 16 bytes each of the line's first 32 (the entry kernels' front), INF
// independent lines per pair in flight before any is used, cheap index arithmetic (one multiply per line).
template <int INF>
__global__ void __launch_bounds__(64, 8)
tbk_calib_gather_pairs_kernel(const uint4 *__restrict__ buf, uint64_t n_lines_buf, uint32_t iters, uint64_t seed, uint32_t *__restrict__ sink) {
    const uint64_t pair = ((uint64_t)blockIdx.x * 64 + threadIdx.x) >> 1;
    const uint32_t sub = threadIdx.x & 1u;
    uint64_t state = tbk_splitmix(seed ^ (pair * 0x9E3779B97F4A7C15ull));
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[INF];
#pragma unroll
        for (int f = 0; f < INF; f++) {
            state = state * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull;  // (an LCG step: the index is its high half times the line count)
            const uint64_t line = (uint64_t)(((unsigned __int128)(state ^ (state >> 29)) * n_lines_buf) >> 64);
            v[f] = buf[line * 8 + sub];
        }
#pragma unroll
        for (int f = 0; f < INF; f++) acc += v[f].x ^ v[f].y ^ v[f].z ^ v[f].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// Streaming read, tuned: U x 16 bytes per thread in flight, non-temporal (nothing is read twice), a grid of `blocks` that walks
// the buffer in strides.
template <int U>
__global__ void __launch_bounds__(256)
tbk_calib_stream_nt_kernel(const uint4 *__restrict__ buf, uint64_t n_vec, uint32_t *__restrict__ sink) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 *src = reinterpret_cast<const u4 *>(buf);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + (U - 1) * stride < n_vec; i += (uint64_t)U * stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(src + i + (uint64_t)u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n_vec; i += stride) { const u4 a = src[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// Fire-and-forget 32-bit atomic adds.  `run` consecutive adds of a lane go to consecutive words of
// one random 128-byte line (run = 1: every add to its own random line), which is the counting
// kernel's pattern: the windows of a read that share a minimizer update counters of one line.
__global__ void __launch_bounds__(256)
tbk_calib_atomics_kernel(uint32_t *__restrict__ buf, uint64_t n_lines, uint32_t iters, uint32_t run, uint64_t seed) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t state = seed ^ (gid * 0x9E3779B97F4A7C15ull);
    for (uint32_t it = 0; it < iters; it += run) {
        state = tbk_splitmix(state);
        uint32_t *line = buf + (uint64_t)(((unsigned __int128)state * n_lines) >> 64) * 32;
        for (uint32_t r = 0; r < run && it + r < iters; r++) atomicAdd(&line[(r + (uint32_t)(state >> 60)) & 31u], 1u);
    }
}

// the counting kernel's own adds: 64-bit (two 32-bit counters at once), `run` of them on neighbouring words of one line's
// counter block (words 8..11 of the 16-word line: tbk_count_kernels.hip)
__global__ void __launch_bounds__(256)
tbk_calib_atomics64_kernel(unsigned long long *__restrict__ buf, uint64_t n_lines, uint32_t iters, uint32_t run, uint64_t seed) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t state = seed ^ (gid * 0x9E3779B97F4A7C15ull);
    for (uint32_t it = 0; it < iters; it += run) {
        state = tbk_splitmix(state);
        unsigned long long *line = buf + (uint64_t)(((unsigned __int128)state * n_lines) >> 64) * 16 + 8;
        for (uint32_t r = 0; r < run && it + r < iters; r++) atomicAdd(&line[(r + (uint32_t)(state >> 62)) & 3u], 0x100000001ull);
    }
}

__global__ void __launch_bounds__(256)
tbk_fill_kernel(uint4 *__restrict__ buf, uint64_t n_vec, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_vec; i += stride) {
        const uint64_t r = tbk_splitmix(seed + i);
        buf[i] = make_uint4((uint32_t)r, (uint32_t)(r >> 32), (uint32_t)i, (uint32_t)(i >> 32));
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static inline unsigned grid_for(uint64_t n, unsigned cap = 1u << 16) {
    uint64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

extern "C" hipError_t tbk_launch_synth_keys(uint64_t seed, uint64_t first, uint64_t n, int k, uint64_t *d_out,
                                            hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(tbk_synth_keys_kernel, dim3(grid_for(n)), dim3(256), 0, s, seed, first, n, k, d_out);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_synth_reads(uint64_t read_seed, uint64_t first_read, uint64_t n_reads,
                                             uint32_t read_len, uint64_t key_seed, uint64_t n_a, uint64_t n_b,
                                             int k, int major, int minor, uint8_t *d_bases, uint64_t *d_offsets,
                                             hipStream_t s) {
    if (!n_reads) return hipSuccess;
    const uint64_t total = n_reads * (uint64_t)read_len;
    const uint64_t n_chunks = (total + 15) / 16;  // buffer is allocated to a multiple of 16
    const uint64_t first_chunk = first_read * (((uint64_t)read_len + 15) / 16);
    hipLaunchKernelGGL(tbk_synth_background_kernel, dim3(grid_for(n_chunks)), dim3(256), 0, s,
                       read_seed, first_chunk, n_chunks, (uint4 *)d_bases);
    hipLaunchKernelGGL(tbk_synth_offsets_kernel, dim3((unsigned)((n_reads + 256) / 256)), dim3(256), 0, s,
                       n_reads, read_len, d_offsets);
    const uint64_t n_plant = n_reads * (uint64_t)(major + minor);
    if (n_plant)
        hipLaunchKernelGGL(tbk_synth_plant_kernel, dim3((unsigned)((n_plant + 255) / 256)), dim3(256), 0, s,
                           read_seed, first_read, n_reads, read_len, key_seed, n_a, n_b, k, major, minor,
                           d_bases);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_synth_hap_keys(uint64_t seed, uint64_t genome_len, uint32_t snp24, int k, uint64_t *d_a,
                                                uint64_t *d_b, uint64_t capacity, unsigned long long *d_n, hipStream_t s) {
    const uint64_t n_seg = (genome_len - (uint64_t)k + 1 + TBK_HAP_SEG - 1) / TBK_HAP_SEG;
    hipLaunchKernelGGL(tbk_synth_hap_keys_kernel, dim3(grid_for(n_seg)), dim3(256), 0, s, seed, genome_len, snp24, k, d_a, d_b,
                       capacity, d_n);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_synth_hap_reads(uint64_t seed, uint64_t genome_len, uint32_t snp24, uint64_t read_seed,
                                                 uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint32_t err24,
                                                 uint8_t *d_bases, uint64_t *d_offsets, hipStream_t s) {
    if (!n_reads) return hipSuccess;
    const uint64_t n_chunks = (n_reads * (uint64_t)read_len + 15) / 16;
    hipLaunchKernelGGL(tbk_synth_hap_reads_kernel, dim3(grid_for(n_chunks)), dim3(256), 0, s, seed, genome_len, snp24, read_seed,
                       first_read, n_reads, read_len, err24, d_bases);
    hipLaunchKernelGGL(tbk_synth_offsets_kernel, dim3((unsigned)((n_reads + 256) / 256)), dim3(256), 0, s,
                       n_reads, read_len, d_offsets);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_synth_reads_ragged(uint64_t read_seed, uint64_t first_read, uint64_t n_reads, const uint64_t *d_offsets, uint64_t total,
                                                    uint64_t key_seed, uint64_t n_a, uint64_t n_b, int k, uint32_t slot_len, uint8_t *d_bases, hipStream_t s) {
    if (!n_reads || !total) return hipSuccess;
    const uint64_t n_chunks = (total + 15) / 16;
    // (the background is a function of the absolute chunk index: batches of one run start at distinct indices)
    hipLaunchKernelGGL(tbk_synth_background_kernel, dim3(grid_for(n_chunks)), dim3(256), 0, s, read_seed, first_read * 1024, n_chunks, (uint4 *)d_bases);
    hipLaunchKernelGGL(tbk_synth_plant_ragged_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, s, read_seed, first_read, n_reads, d_offsets,
                       key_seed, n_a, n_b, k, slot_len, d_bases);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_synth_hap_reads_ragged(uint64_t seed, uint64_t genome_len, uint32_t snp24, uint64_t read_seed, uint64_t first_read,
                                                        uint64_t n_reads, const uint64_t *d_offsets, uint64_t total, uint32_t err24, uint8_t *d_bases, hipStream_t s) {
    if (!n_reads || !total) return hipSuccess;
    hipLaunchKernelGGL(tbk_synth_hap_reads_ragged_kernel, dim3(grid_for((total + 15) / 16)), dim3(256), 0, s, seed, genome_len, snp24, read_seed, first_read,
                       n_reads, d_offsets, err24, d_bases);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_keys_as_reads(const uint64_t *d_keys, uint64_t first, uint64_t n, int k, uint32_t per_read, uint8_t *d_bases, uint64_t *d_offsets,
                                               uint64_t n_reads, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(tbk_synth_keys_as_reads_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, d_keys, first, n, k, per_read, d_bases, d_offsets, n_reads);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_mutate_keys(const uint64_t *d_keys, uint64_t first, uint64_t n, int k, uint64_t seed, uint64_t *d_out, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(tbk_synth_mutate_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_keys, first, n, k, seed, d_out);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_canonical_keys(const uint64_t *d_keys, uint64_t n, int k, uint64_t *d_out, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(tbk_canonical_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_keys, n, k, d_out);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_sweep_expect(const uint8_t *d_in_a, const uint8_t *d_in_b, uint64_t n, uint8_t *d_expect, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(tbk_sweep_expect_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in_a, d_in_b, n, d_expect);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_counts_check(const int32_t *d_counts, uint64_t first, uint64_t n_reads, uint32_t per_read, uint64_t n_keys, int expect,
                                              const uint8_t *d_expect, unsigned long long *d_out, hipStream_t s) {
    if (!n_reads) return hipSuccess;
    hipLaunchKernelGGL(tbk_counts_check_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, s, d_counts, first, n_reads, per_read, n_keys, expect, d_expect, d_out);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_atomics(void *d_buf, uint64_t bytes, uint32_t iters, uint32_t run, uint64_t seed, unsigned blocks,
                                         hipStream_t s) {
    hipLaunchKernelGGL(tbk_calib_atomics_kernel, dim3(blocks), dim3(256), 0, s, (uint32_t *)d_buf, bytes / 128, iters, run, seed);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_atomics64(void *d_buf, uint64_t bytes, uint32_t iters, uint32_t run, uint64_t seed, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL(tbk_calib_atomics64_kernel, dim3(blocks), dim3(256), 0, s, (unsigned long long *)d_buf, bytes / 128, iters, run, seed);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_fill(void *d_buf, uint64_t bytes, uint64_t seed, hipStream_t s) {
    const uint64_t n_vec = bytes / 16;
    hipLaunchKernelGGL(tbk_fill_kernel, dim3(grid_for(n_vec, 8192)), dim3(256), 0, s, (uint4 *)d_buf, n_vec, seed);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_stream(const void *d_buf, uint64_t bytes, uint32_t *d_sink, hipStream_t s) {
    const uint64_t n_vec = bytes / 16;
    hipLaunchKernelGGL(tbk_calib_stream_kernel, dim3(256 * 16), dim3(256), 0, s, (const uint4 *)d_buf, n_vec, d_sink);
    return hipGetLastError();
}

// lines_done: what one launch gathers.  waves_per_simd blocks of one wave per SIMD of 256 CUs, each pair `iters` x inf lines.
extern "C" hipError_t tbk_launch_gather_pairs(const void *d_buf, uint64_t bytes, int inf, int waves_per_simd, uint64_t n_lines, uint64_t seed, uint32_t *d_sink,
                                              uint64_t *lines_done, hipStream_t s) {
    const unsigned blocks = 256u * 4u * (unsigned)(waves_per_simd < 1 ? 1 : waves_per_simd > 8 ? 8 : waves_per_simd) * 4u;  // four rounds of resident waves
    const uint64_t pairs = (uint64_t)blocks * 32;
    uint64_t iters = n_lines / (pairs * (uint64_t)inf);
    if (iters < 1) iters = 1;
    *lines_done = iters * pairs * (uint64_t)inf;
    const uint64_t n_lines_buf = bytes / 128;
#define TBK_GP(F) if (inf == F) { hipLaunchKernelGGL((tbk_calib_gather_pairs_kernel<F>), dim3(blocks), dim3(64), 0, s, (const uint4 *)d_buf, n_lines_buf, (uint32_t)iters, seed, d_sink); return hipGetLastError(); }
    TBK_GP(1) TBK_GP(2) TBK_GP(3) TBK_GP(4) TBK_GP(6) TBK_GP(8)
#undef TBK_GP
    return hipErrorInvalidValue;
}

extern "C" hipError_t tbk_launch_stream_nt(const void *d_buf, uint64_t bytes, int unroll, unsigned blocks, uint32_t *d_sink, hipStream_t s) {
    const uint64_t n_vec = bytes / 16;
#define TBK_SN(U) if (unroll == U) { hipLaunchKernelGGL((tbk_calib_stream_nt_kernel<U>), dim3(blocks), dim3(256), 0, s, (const uint4 *)d_buf, n_vec, d_sink); return hipGetLastError(); }
    TBK_SN(1) TBK_SN(2) TBK_SN(4) TBK_SN(8)
#undef TBK_SN
    return hipErrorInvalidValue;
}

template <int LINE, int LPL, int INF>
static hipError_t launch_gather_t(const void *d_buf, uint64_t bytes, uint64_t n_lines, uint64_t seed,
                                  uint32_t *d_sink, hipStream_t s) {
    const uint64_t n_lines_buf = bytes / LINE;
    // groups = lines in flight at once; each group does iters*INF lines
    const unsigned blocks = 256 * 8;  // 8 blocks of 256 threads per CU
    const uint64_t groups = (uint64_t)blocks * 256 / LPL;
    uint64_t iters = n_lines / (groups * INF);
    if (iters < 1) iters = 1;
    hipLaunchKernelGGL((tbk_calib_gather_kernel<LINE, LPL, INF>), dim3(blocks), dim3(256), 0, s,
                       (const uint4 *)d_buf, n_lines_buf, (uint32_t)iters, seed, d_sink);
    return hipGetLastError();
}

// returns the number of lines one launch actually gathers through *lines_done
extern "C" hipError_t tbk_launch_gather(const void *d_buf, uint64_t bytes, int line, int lpl, int inf,
                                        uint64_t n_lines, uint64_t seed, uint32_t *d_sink, uint64_t *lines_done,
                                        hipStream_t s) {
    const uint64_t groups = (uint64_t)256 * 8 * 256 / (uint64_t)lpl;
    uint64_t iters = n_lines / (groups * (uint64_t)inf);
    if (iters < 1) iters = 1;
    *lines_done = iters * groups * (uint64_t)inf;
#define TBK_G(L, P, F) \
    if (line == L && lpl == P && inf == F) return launch_gather_t<L, P, F>(d_buf, bytes, n_lines, seed, d_sink, s);
#define TBK_GF(L, P) TBK_G(L, P, 1) TBK_G(L, P, 2) TBK_G(L, P, 4) TBK_G(L, P, 8)
    TBK_GF(64, 1) TBK_GF(64, 4) TBK_GF(128, 1) TBK_GF(128, 8) TBK_GF(128, 4) TBK_GF(32, 2)  // (32, 2): two lanes x 16 bytes of a line, the pair-cooperative probe of the entry layout
#undef TBK_GF
#undef TBK_G
    return hipErrorInvalidValue;
}
