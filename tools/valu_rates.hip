// valu_rates.hip: issue cost of the compare instructions the probe kernel lives on, relative to v_xor_b32.
// Every SIMD runs one wave that issues 16 independent instructions per loop trip; time / (trips * 16) against
// the same loop of v_xor_b32 says how many passes the instruction takes on the 16-lane SIMD.
// Build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP>
__global__ void __launch_bounds__(64) rate_kernel(uint64_t *out, int trips, uint64_t seed) {
    uint64_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
    uint32_t a32 = (uint32_t)a, b32 = (uint32_t)b, acc = 0;
    uint64_t m = 0;
    for (int i = 0; i < trips; i++) {
        if (OP == 0) { REP16(asm volatile("v_xor_b32 %0, %1, %0" : "+v"(acc) : "v"(a32));) }
        if (OP == 1) { REP16(asm volatile("v_cmp_eq_u32 %0, %1, %2" : "=s"(m) : "v"(a32), "v"(b32));) }
        if (OP == 2) { REP16(asm volatile("v_cmp_eq_u64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));) }
        if (OP == 3) { REP16(asm volatile("v_cmp_gt_u64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));) }
        if (OP == 4) { REP16(asm volatile("v_min_u32 %0, %1, %0" : "+v"(acc) : "v"(a32));) }
        if (OP == 5) { REP16(asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(acc) : "v"(a32) : );) }
        if (OP == 6) { REP16(asm volatile("v_alignbit_b32 %0, %1, %0, 2" : "+v"(acc) : "v"(a32));) }
        if (OP == 7) { REP16(asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(a));) }
        if (OP == 8) { REP16(asm volatile("s_or_b64 %0, %0, %1" : "+s"(m) : "s"(seed));) }
    }
    if (acc == 0x12345 || m == 0x1234567 || a == 77) out[0] = acc + m + a;
}

template <int OP>
static double run(const char *name, uint64_t *d_out, double base) {
    const int trips = 20000, blocks = 256 * 4;  // one wave per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(64), 0, 0, d_out, 100, 12345ull);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(64), 0, 0, d_out, trips, 12345ull);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_per = ms * 1e6 / ((double)trips * 16);
    printf("%-16s %7.3f ms  %6.3f ns per instruction per wave%s", name, ms, ns_per, base > 0 ? "" : "\n");
    if (base > 0) printf("  = %.2f x v_xor_b32\n", ns_per / base);
    return ns_per;
}

int main() {
    uint64_t *d_out;
    hipMalloc(&d_out, 64);
    const double base = run<0>("v_xor_b32", d_out, 0);
    run<1>("v_cmp_eq_u32", d_out, base);
    run<2>("v_cmp_eq_u64", d_out, base);
    run<3>("v_cmp_gt_u64", d_out, base);
    run<4>("v_min_u32", d_out, base);
    run<5>("v_cndmask_b32", d_out, base);
    run<6>("v_alignbit_b32", d_out, base);
    run<7>("v_lshrrev_b64", d_out, base);
    run<8>("s_or_b64", d_out, base);
    // the same with five waves per SIMD (what the probe kernel runs at): does the rate per SIMD change?
    return 0;
}
