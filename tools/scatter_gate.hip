// scatter_gate.hip: the gate of "partition-then-probe" (VERDICT r3 item 5).  The idea: make table access sequential by
// radix-partitioning a batch's super-k-mer records (~0.9 G records of 16 bytes per 3.9-Gbase batch) by bucket range in two
// LDS-staged levels, then probe each partition against an L2-sized slice of the table.  The gate: does one level of that
// scatter sustain >= 3 TB/s of effective traffic (16 B read + 16 B written per record)?  Two levels at 3 TB/s are
// 2 x 0.9e9 x 32 B / 3e12 = 19 ms per batch - as long as the whole probe of the entry layout takes today - so anything below
// ends the idea before a line of the probe side is written.
// One level here: 256 partitions by 8 bits of the record's key.  A block takes a tile of 4096 records into registers,
// histograms the digits in LDS, orders the tile by digit in LDS (64 KB), claims room in every partition with one atomic add per
// digit and tile, and writes each digit's run of the tile contiguously (on average 16 records = 256 bytes per run).
// Build + run: hipcc -O3 --offload-arch=gfx950 tools/scatter_gate.hip -o /tmp/scatter_gate && /tmp/scatter_gate [records]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

constexpr int TILE = 4096, THREADS = 256, PER = TILE / THREADS, PARTS = 256;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; return x ^ (x >> 16); }

__global__ void __launch_bounds__(THREADS) fill(uint4 *rec, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * THREADS)
        rec[i] = make_uint4(mix((uint32_t)i), mix((uint32_t)(i >> 3) + 77u), (uint32_t)i, (uint32_t)(i >> 32));
}

template <int SHIFT>
__global__ void __launch_bounds__(THREADS) scatter(const uint4 *__restrict__ in, uint64_t n, uint4 *__restrict__ out, unsigned long long *__restrict__ cursor,
                                                   uint64_t part_cap) {
    __shared__ uint4 tile[TILE];
    __shared__ uint32_t hist[PARTS], start[PARTS], base_lo[PARTS], base_hi[PARTS];
    const uint32_t t = threadIdx.x;
    for (uint64_t t0 = (uint64_t)blockIdx.x * TILE; t0 < n; t0 += (uint64_t)gridDim.x * TILE) {
        hist[t] = 0;
        __syncthreads();
        uint4 r[PER];
        uint32_t d[PER], rank[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint64_t at = t0 + (uint64_t)i * THREADS + t;
            r[i] = at < n ? in[at] : make_uint4(0, 0, 0, 0);
            d[i] = at < n ? (r[i].x >> SHIFT) & (PARTS - 1) : 0xFFFFFFFFu;
            rank[i] = d[i] != 0xFFFFFFFFu ? atomicAdd(&hist[d[i]], 1u) : 0;
        }
        __syncthreads();
        // exclusive prefix of the 256 digit counts (one wave does it with shuffles; the tile is small)
        if (t < 64) {
            uint32_t v[4], s = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { v[j] = hist[4 * t + j]; s += v[j]; }
            uint32_t inc = s;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off, 64); if ((int)t >= off) inc += o; }
            uint32_t ex = inc - s;
#pragma unroll
            for (int j = 0; j < 4; j++) { start[4 * t + j] = ex; ex += v[j]; }
        }
        __syncthreads();
        // room in the partitions: one atomic per digit and tile
        {
            const uint32_t c = hist[t];
            const unsigned long long b = c ? atomicAdd(&cursor[t], (unsigned long long)c) : 0ull;
            base_lo[t] = (uint32_t)b; base_hi[t] = (uint32_t)(b >> 32);
        }
#pragma unroll
        for (int i = 0; i < PER; i++)
            if (d[i] != 0xFFFFFFFFu) tile[start[d[i]] + rank[i]] = r[i];
        __syncthreads();
        // the ordered tile goes out: consecutive threads write consecutive records of a digit's run
        const uint32_t n_tile = (uint32_t)(n - t0 < TILE ? n - t0 : TILE);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint32_t at = (uint32_t)i * THREADS + t;
            if (at < n_tile) {
                const uint4 v = tile[at];
                const uint32_t dg = (v.x >> SHIFT) & (PARTS - 1);
                const uint64_t b = ((uint64_t)base_hi[dg] << 32) | base_lo[dg];
                out[(uint64_t)dg * part_cap + b + (at - start[dg])] = v;
            }
        }
        __syncthreads();
    }
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 900000000ull;
    const uint64_t part_cap = n / PARTS + n / PARTS / 8 + 65536;  // (uniform digits: a partition stays within 1.125 x its share)
    uint4 *in = nullptr, *out = nullptr;
    unsigned long long *cursor = nullptr;
    if (hipMalloc(&in, n * 16) != hipSuccess || hipMalloc(&out, part_cap * PARTS * 16) != hipSuccess || hipMalloc(&cursor, PARTS * 8) != hipSuccess) { printf("allocation failed\n"); return 1; }
    hipLaunchKernelGGL(fill, dim3(4096), dim3(THREADS), 0, 0, in, n);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks_per_cu : {2, 4, 8}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            (void)hipMemset(cursor, 0, PARTS * 8);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(scatter<8>, dim3(256 * blocks_per_cu), dim3(THREADS), 0, 0, in, n, out, cursor, part_cap);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); return 1; }
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("one level, %llu records of 16 B into 256 partitions, %d blocks per CU: %.2f ms = %.2f TB/s effective (read + write), %.1f G records/s\n",
               (unsigned long long)n, blocks_per_cu, best, (double)n * 32 / (best * 1e-3) / 1e12, (double)n / (best * 1e-3) / 1e9);
    }
    // check: every record arrived in its partition exactly once (counts only)
    unsigned long long h[PARTS];
    (void)hipMemcpy(h, cursor, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long tot = 0, mx = 0;
    for (int i = 0; i < PARTS; i++) { tot += h[i]; mx = h[i] > mx ? h[i] : mx; }
    printf("records scattered %llu of %llu, largest partition %llu (capacity %llu)\n", tot, (unsigned long long)n, mx, (unsigned long long)part_cap);
    return tot == n && mx <= part_cap ? 0 : 2;
}
