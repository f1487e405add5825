// valu_rates.hip: issue THROUGHPUT of the instructions the probe kernel lives on, relative to v_xor_b32.
// Eight waves per SIMD (one-wave blocks, the chip filled), every wave issuing 16 instructions per loop trip over four
// independent accumulators, so that neither a dependency nor a lone wave's issue gap is what gets measured:
// time / (trips * 16 * waves per SIMD) = the SIMD's time per instruction.  (Round 3's version ran one wave per SIMD on
// one dependent chain and so measured latency: every full-rate instruction read "1.0".)
// Build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(X0, X1, X2, X3) X0 X1 X2 X3 X0 X1 X2 X3 X0 X1 X2 X3 X0 X1 X2 X3

enum {
    OP_XOR, OP_MUL_LO, OP_MUL_HI, OP_MUL_U24, OP_MAD_U24, OP_LSHR64, OP_LSHL_ADD64, OP_CMP_EQ64, OP_CMP_EQ32, OP_CMP_GT64, OP_DPP, OP_XOR_DPP,
    OP_BFI, OP_MIN3, OP_BITOP3, OP_BFE, OP_ALIGNBIT, OP_CNDMASK, OP_BCNT, OP_AND_OR, OP_MAD64, OP_PERM, OP_FFBH, OP_XOR_SDWA, OP_ADD3,
    OP_LSHL_OR, OP_S_OR64, OP_S_BCNT, OP_S_AND32, OP_MIX_VS, OP_CNDMASK_SGPR, OP_CMP_CNDMASK, OP_CMP_ONLY, OP_MIX_XOR_CND, OP_SUBB, OP_COUNT
};

template <int OP>
__global__ void __launch_bounds__(64) rate_kernel(uint64_t *out, int trips, uint64_t seed) {
    uint64_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c64 = seed * 5 + threadIdx.x, d64 = seed * 7 + threadIdx.x;
    uint32_t a32 = (uint32_t)a, b32 = (uint32_t)b;
    uint32_t r0 = 1, r1 = 2, r2 = 3, r3 = 4;
    uint64_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#define V1(INS) REP4(asm volatile(INS : "+v"(r0) : "v"(a32), "v"(b32));, asm volatile(INS : "+v"(r1) : "v"(a32), "v"(b32));, \
                     asm volatile(INS : "+v"(r2) : "v"(a32), "v"(b32));, asm volatile(INS : "+v"(r3) : "v"(a32), "v"(b32));)
#define C64(INS) REP4(asm volatile(INS : "=s"(m0) : "v"(a), "v"(b));, asm volatile(INS : "=s"(m1) : "v"(c64), "v"(b));, \
                      asm volatile(INS : "=s"(m2) : "v"(a), "v"(d64));, asm volatile(INS : "=s"(m3) : "v"(c64), "v"(d64));)
#define C32(INS) REP4(asm volatile(INS : "=s"(m0) : "v"(a32), "v"(b32));, asm volatile(INS : "=s"(m1) : "v"(r0), "v"(b32));, \
                      asm volatile(INS : "=s"(m2) : "v"(a32), "v"(r1));, asm volatile(INS : "=s"(m3) : "v"(r2), "v"(r3));)
#define W64(INS) REP4(asm volatile(INS : "+v"(a) : "v"(b32));, asm volatile(INS : "+v"(b) : "v"(b32));, asm volatile(INS : "+v"(c64) : "v"(b32));, \
                      asm volatile(INS : "+v"(d64) : "v"(b32));)
#define S64(INS) REP4(asm volatile(INS : "+s"(m0) : "s"(seed));, asm volatile(INS : "+s"(m1) : "s"(seed));, asm volatile(INS : "+s"(m2) : "s"(seed));, \
                      asm volatile(INS : "+s"(m3) : "s"(seed));)
    for (int i = 0; i < trips; i++) {
        if (OP == OP_XOR) { V1("v_xor_b32 %0, %1, %0") }
        if (OP == OP_MUL_LO) { V1("v_mul_lo_u32 %0, %1, %0") }
        if (OP == OP_MUL_HI) { V1("v_mul_hi_u32 %0, %1, %0") }
        if (OP == OP_MUL_U24) { V1("v_mul_u32_u24 %0, %1, %0") }
        if (OP == OP_MAD_U24) { V1("v_mad_u32_u24 %0, %1, %0, %2") }
        if (OP == OP_LSHR64) { W64("v_lshrrev_b64 %0, %1, %0") }
        if (OP == OP_LSHL_ADD64) { W64("v_lshl_add_u64 %0, %0, 3, %0") }
        if (OP == OP_CMP_EQ64) { C64("v_cmp_eq_u64 %0, %1, %2") }
        if (OP == OP_CMP_GT64) { C64("v_cmp_gt_u64 %0, %1, %2") }
        if (OP == OP_CMP_EQ32) { C32("v_cmp_eq_u32 %0, %1, %2") }
        if (OP == OP_DPP) { V1("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf") }
        if (OP == OP_XOR_DPP) { V1("v_xor_b32_dpp %0, %1, %0 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf") }
        if (OP == OP_BFI) { V1("v_bfi_b32 %0, %1, %0, %2") }
        if (OP == OP_MIN3) { V1("v_min3_u32 %0, %1, %0, %2") }
        if (OP == OP_BITOP3) { V1("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x6e") }
        if (OP == OP_BFE) { V1("v_bfe_u32 %0, %0, %1, 10") }
        if (OP == OP_ALIGNBIT) { V1("v_alignbit_b32 %0, %1, %0, 2") }
        if (OP == OP_CNDMASK) { V1("v_cndmask_b32 %0, %1, %0, vcc") }
        if (OP == OP_CNDMASK_SGPR) {  // the mask in a scalar pair that nothing writes (the e64 form)
            REP4(asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r0) : "v"(a32), "s"(seed));, asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r1) : "v"(a32), "s"(seed));,
                 asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r2) : "v"(a32), "s"(seed));, asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r3) : "v"(a32), "s"(seed));)
        }
        if (OP == OP_CMP_CNDMASK) {  // a select as the compiler writes it: compare into vcc, then v_cndmask (two instructions per accumulator)
            REP4(asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r0) : "v"(a32) : "vcc");, asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r1) : "v"(a32) : "vcc");,
                 asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r2) : "v"(a32) : "vcc");, asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r3) : "v"(a32) : "vcc");)
        }
        if (OP == OP_CMP_ONLY) {
            REP4(asm volatile("v_cmp_lt_u32 vcc, %1, %0" : "+v"(r0) : "v"(a32) : "vcc");, asm volatile("v_cmp_lt_u32 vcc, %1, %0" : "+v"(r1) : "v"(a32) : "vcc");,
                 asm volatile("v_cmp_lt_u32 vcc, %1, %0" : "+v"(r2) : "v"(a32) : "vcc");, asm volatile("v_cmp_lt_u32 vcc, %1, %0" : "+v"(r3) : "v"(a32) : "vcc");)
        }
        if (OP == OP_MIX_XOR_CND) {  // three v_xor and one v_cndmask per accumulator round: does the slow one hide among others?
            REP4(asm volatile("v_xor_b32 %0, %1, %0\n v_xor_b32 %0, %2, %0\n v_xor_b32 %0, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r0) : "v"(a32), "v"(b32));,
                 asm volatile("v_xor_b32 %0, %1, %0\n v_xor_b32 %0, %2, %0\n v_xor_b32 %0, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r1) : "v"(a32), "v"(b32));,
                 asm volatile("v_xor_b32 %0, %1, %0\n v_xor_b32 %0, %2, %0\n v_xor_b32 %0, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r2) : "v"(a32), "v"(b32));,
                 asm volatile("v_xor_b32 %0, %1, %0\n v_xor_b32 %0, %2, %0\n v_xor_b32 %0, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r3) : "v"(a32), "v"(b32));)
        }
        if (OP == OP_SUBB) { V1("v_subb_co_u32 %0, vcc, %1, %0, vcc") }
        if (OP == OP_BCNT) { V1("v_bcnt_u32_b32 %0, %1, %0") }
        if (OP == OP_AND_OR) { V1("v_and_or_b32 %0, %1, %0, %2") }
        if (OP == OP_MAD64) {  // (writes a carry: vcc is declared clobbered - undeclared, the loop around it never ended)
            REP4(asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a) : "v"(b32) : "vcc");, asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(b) : "v"(b32) : "vcc");,
                 asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(c64) : "v"(b32) : "vcc");, asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(d64) : "v"(b32) : "vcc");)
        }
        if (OP == OP_PERM) { V1("v_perm_b32 %0, %1, %0, %2") }
        if (OP == OP_FFBH) { V1("v_ffbh_u32 %0, %0") }
        if (OP == OP_XOR_SDWA) { V1("v_xor_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") }
        if (OP == OP_ADD3) { V1("v_add3_u32 %0, %1, %0, %2") }
        if (OP == OP_LSHL_OR) { V1("v_lshl_or_b32 %0, %0, 2, %1") }
        if (OP == OP_S_OR64) { S64("s_or_b64 %0, %0, %1") }
        if (OP == OP_S_BCNT) { REP4(asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(s0) : "s"(seed));, asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(s1) : "s"(seed));,
                                    asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(s2) : "s"(seed));, asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(s3) : "s"(seed));) }
        if (OP == OP_S_AND32) { REP4(asm volatile("s_and_b32 %0, %0, %1" : "+s"(s0) : "s"((uint32_t)seed));, asm volatile("s_and_b32 %0, %0, %1" : "+s"(s1) : "s"((uint32_t)seed));,
                                     asm volatile("s_and_b32 %0, %0, %1" : "+s"(s2) : "s"((uint32_t)seed));, asm volatile("s_and_b32 %0, %0, %1" : "+s"(s3) : "s"((uint32_t)seed));) }
        if (OP == OP_MIX_VS) {  // 16 vector + 16 scalar instructions, interleaved: do the two pipes overlap across waves?
            REP4(asm volatile("v_xor_b32 %0, %2, %0\n s_or_b64 %1, %1, %3" : "+v"(r0), "+s"(m0) : "v"(a32), "s"(seed));,
                 asm volatile("v_xor_b32 %0, %2, %0\n s_or_b64 %1, %1, %3" : "+v"(r1), "+s"(m1) : "v"(a32), "s"(seed));,
                 asm volatile("v_xor_b32 %0, %2, %0\n s_or_b64 %1, %1, %3" : "+v"(r2), "+s"(m2) : "v"(a32), "s"(seed));,
                 asm volatile("v_xor_b32 %0, %2, %0\n s_or_b64 %1, %1, %3" : "+v"(r3), "+s"(m3) : "v"(a32), "s"(seed));)
        }
    }
    if (r0 + r1 + r2 + r3 == 0x12345 || (m0 ^ m1 ^ m2 ^ m3) == 0x1234567 || (a ^ b ^ c64 ^ d64) == 77 || s0 + s1 + s2 + s3 == 99) out[0] = r0 + m0 + a + s0;
}

static int g_waves = 8;

template <int OP>
static double run(const char *name, uint64_t *d_out, double base) {
    const int trips = 4000, blocks = 256 * 4 * g_waves;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(64), 0, 0, d_out, 100, 12345ull);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(64), 0, 0, d_out, trips, 12345ull);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_per = ms * 1e6 / ((double)trips * 16 * g_waves);
    printf("%-18s %7.3f ms  %6.3f ns per instruction per SIMD", name, ms, ns_per);
    if (base > 0) printf("  = %.2f x v_xor_b32", ns_per / base);
    printf("\n");
    return ns_per;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc > 1) g_waves = atoi(argv[1]);
    uint64_t *d_out;
    hipMalloc(&d_out, 64);
    printf("waves per SIMD: %d\n", g_waves);
    const double base = run<OP_XOR>("v_xor_b32", d_out, 0);
    if (argc > 2) {  // (`valu_rates 8 select`: what a select costs - round 5)
        run<OP_MIN3>("v_min3_u32", d_out, base);
        run<OP_BFI>("v_bfi_b32", d_out, base);
        run<OP_CMP_EQ32>("v_cmp_eq_u32", d_out, base);
        run<OP_CNDMASK>("v_cndmask_b32 vcc", d_out, base);
        run<OP_CNDMASK_SGPR>("v_cndmask e64 sgpr", d_out, base);
        run<OP_CMP_ONLY>("v_cmp_lt_u32 vcc", d_out, base);
        run<OP_CMP_CNDMASK>("cmp+cndmask (x2 instr)", d_out, base);
        run<OP_MIX_XOR_CND>("3 xor + 1 cndmask (x4)", d_out, base);
        run<OP_SUBB>("v_subb_co_u32", d_out, base);
        return 0;
    }
    run<OP_MUL_LO>("v_mul_lo_u32", d_out, base);
    run<OP_MUL_HI>("v_mul_hi_u32", d_out, base);
    run<OP_MUL_U24>("v_mul_u32_u24", d_out, base);
    run<OP_MAD_U24>("v_mad_u32_u24", d_out, base);
    run<OP_MAD64>("v_mad_u64_u32", d_out, base);
    run<OP_LSHR64>("v_lshrrev_b64", d_out, base);
    run<OP_LSHL_ADD64>("v_lshl_add_u64", d_out, base);
    run<OP_CMP_EQ32>("v_cmp_eq_u32", d_out, base);
    run<OP_CMP_EQ64>("v_cmp_eq_u64", d_out, base);
    run<OP_CMP_GT64>("v_cmp_gt_u64", d_out, base);
    run<OP_DPP>("v_mov_b32_dpp", d_out, base);
    run<OP_XOR_DPP>("v_xor_b32_dpp", d_out, base);
    run<OP_XOR_SDWA>("v_xor_b32_sdwa", d_out, base);
    run<OP_BFI>("v_bfi_b32", d_out, base);
    run<OP_MIN3>("v_min3_u32", d_out, base);
    run<OP_BITOP3>("v_bitop3_b32", d_out, base);
    run<OP_BFE>("v_bfe_u32", d_out, base);
    run<OP_ALIGNBIT>("v_alignbit_b32", d_out, base);
    run<OP_CNDMASK>("v_cndmask_b32", d_out, base);
    run<OP_CNDMASK_SGPR>("v_cndmask e64 sgpr", d_out, base);
    run<OP_CMP_ONLY>("v_cmp_lt_u32 vcc", d_out, base);
    run<OP_CMP_CNDMASK>("cmp+cndmask (x2 instr)", d_out, base);
    run<OP_MIX_XOR_CND>("3 xor + 1 cndmask (x4)", d_out, base);
    run<OP_SUBB>("v_subb_co_u32", d_out, base);
    run<OP_BCNT>("v_bcnt_u32_b32", d_out, base);
    run<OP_AND_OR>("v_and_or_b32", d_out, base);
    run<OP_ADD3>("v_add3_u32", d_out, base);
    run<OP_LSHL_OR>("v_lshl_or_b32", d_out, base);
    run<OP_PERM>("v_perm_b32", d_out, base);
    run<OP_FFBH>("v_ffbh_u32", d_out, base);
    run<OP_S_OR64>("s_or_b64", d_out, base);
    run<OP_S_BCNT>("s_bcnt1_i32_b64", d_out, base);
    run<OP_S_AND32>("s_and_b32", d_out, base);
    run<OP_MIX_VS>("v_xor + s_or pair", d_out, base);
    return 0;
}
