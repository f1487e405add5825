// Does the memory type of the table change what a random 32-byte request costs?  The same gather (one-wave blocks, two lanes x
// 16 bytes of a 128-byte line, INF lines in flight per pair) over 33 GB allocated with hipMalloc, as fine-grained and as
// uncached device memory (hipExtMallocWithFlags): G lines per second.  Under rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum ... the
// request sizes the L2 sends to the fabric for each.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/uncached_gather tools/uncached_gather.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int INF, int BYTES>
__global__ __launch_bounds__(64, 8) void gather(const uint4 *__restrict__ buf, uint64_t n_lines, uint32_t iters, uint64_t seed, uint32_t *sink) {
    const uint32_t lane = threadIdx.x, pair = lane >> 1, half = lane & 1;
    uint64_t s = seed ^ ((uint64_t)blockIdx.x * 32 + pair) * 0x9E3779B97F4A7C15ull;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[INF];
#pragma unroll
        for (int j = 0; j < INF; j++) {
            s = mix(s + j + 1);
            const uint64_t line = (uint64_t)(((unsigned __int128)s * n_lines) >> 64);
            v[j] = buf[line * 8 + half * (BYTES / 32)];   // BYTES = 32: the line's first 32 bytes; 64: 16 bytes of each of its first two 32-byte sectors
        }
#pragma unroll
        for (int j = 0; j < INF; j++) acc ^= v[j].x ^ v[j].w;
    }
    if (acc == 0x12345u) *sink = acc;
}

int main(int argc, char **argv) {
    const uint64_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : 33424860800ull;
    const uint64_t n_lines = bytes / 128;
    const uint32_t blocks = 256 * 4 * 8 * 16, iters = 64;
    uint32_t *sink; CK(hipMalloc((void **)&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"hipMalloc", "fine-grained", "uncached"};
    for (int kind = 0; kind < 3; kind++) {
        void *buf = nullptr;
        hipError_t e = kind == 0 ? hipMalloc(&buf, bytes) : hipExtMallocWithFlags(&buf, bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
        if (e != hipSuccess) { printf("%s: allocation failed: %s\n", names[kind], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipMemsetAsync(buf, 1, bytes, nullptr));
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 3; rep++) {
            float ms;
            const double lines = (double)blocks * 32 * iters * 4;
            gather<4, 32><<<blocks, 64>>>((const uint4 *)buf, n_lines, iters / 4, 77 + rep, sink);   // warm-up
            CK(hipEventRecord(e0, nullptr));
            gather<4, 32><<<blocks, 64>>>((const uint4 *)buf, n_lines, iters, 99 + rep, sink);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-13s 32 B of a line   %.3f ms  %.2f G lines/s\n", names[kind], ms, lines / ms / 1e6);
            CK(hipEventRecord(e0, nullptr));
            gather<4, 64><<<blocks, 64>>>((const uint4 *)buf, n_lines, iters, 199 + rep, sink);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-13s 2 x 16 B in two sectors   %.3f ms  %.2f G lines/s\n", names[kind], ms, lines / ms / 1e6);
        }
        CK(hipFree(buf));
    }
    return 0;
}
