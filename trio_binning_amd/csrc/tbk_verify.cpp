// tbk_verify.cpp — workload generators and checkers that sit beside the hot path, not on it:
//   * reads of any lengths (BASELINE configs[4]: "50x ONT ultra-long (N50 100 kb)": log-normal lengths);
//   * the full-membership sweep: a list's own keys laid out as reads and pushed through the ordinary probe
//     (tbk_classify_device), tallied on the device.  The reference stores every list line (c/kmers.c:112-122) and
//     finds every stored canonical key (c/kmers.c:245-268); the tables here hold compressed and merged forms of the
//     keys, so "every key answers, and nothing else does" is checked key by key at BASELINE's table sizes.
// Uses the library's own C-ABI only (include/tbk.h) plus the generator kernels of tbk_synth.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/tbk.h"
#include "tbk_common.h"

extern "C" void tbk_set_error_(int, const char *msg);
extern "C" hipError_t tbk_launch_synth_reads_ragged(uint64_t, uint64_t, uint64_t, const uint64_t *, uint64_t, uint64_t, uint64_t, uint64_t, int, uint32_t, uint8_t *, hipStream_t);
extern "C" hipError_t tbk_launch_synth_hap_reads_ragged(uint64_t, uint64_t, uint32_t, uint64_t, uint64_t, uint64_t, const uint64_t *, uint64_t, uint32_t, uint8_t *, hipStream_t);
extern "C" hipError_t tbk_launch_keys_as_reads(const uint64_t *, uint64_t, uint64_t, int, uint32_t, uint8_t *, uint64_t *, uint64_t, hipStream_t);
extern "C" hipError_t tbk_launch_mutate_keys(const uint64_t *, uint64_t, uint64_t, int, uint64_t, uint64_t *, hipStream_t);
extern "C" hipError_t tbk_launch_canonical_keys(const uint64_t *, uint64_t, int, uint64_t *, hipStream_t);
extern "C" hipError_t tbk_launch_sweep_expect(const uint8_t *, const uint8_t *, uint64_t, uint8_t *, hipStream_t);
extern "C" hipError_t tbk_launch_counts_check(const int32_t *, uint64_t, uint64_t, uint32_t, uint64_t, int, const uint8_t *, unsigned long long *, hipStream_t);

static int vfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tbk_set_error_(code, buf);
    return code;
}

#define V_TRY(expr)                                                                                        \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return vfail(e_ == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static int on_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return vfail(TBK_ERR_NO_DEVICE, "no HIP device visible; libtbk_hip has no CPU fallback"); }
    if (device < 0 || device >= n) return vfail(TBK_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    V_TRY(hipSetDevice(device));
    return TBK_OK;
}

// ---- read lengths ---------------------------------------------------------------------------------------------------
// Log-normal lengths whose N50 (the length L with half of all BASES in reads of at least L) is n50: for ln L ~ N(mu,
// sigma^2) the base-weighted median is exp(mu + sigma^2), so mu = ln n50 - sigma^2.  sigma = 0.9: median 44 kb, mean 67 kb,
// 2.7e-4 of the reads beyond 1 Mb.  A share `short_fraction` of the reads is debris instead: lengths log-uniform between
// min_len and 5 kb (reads shorter than k included when min_len is).  Counter-based: read i's length depends on (seed, i)
// alone.  Host code, double precision.
extern "C" int tbk_synth_lognormal_lengths(uint64_t seed, uint64_t first_read, uint64_t n_reads, double n50, double sigma, double short_fraction,
                                           uint32_t min_len, uint32_t max_len, uint64_t *offsets) {
    if (!offsets) return vfail(TBK_ERR_INVALID, "offsets is NULL");
    if (!(n50 >= 1) || !(sigma >= 0) || !(short_fraction >= 0 && short_fraction <= 1) || min_len < 1 || max_len < min_len)
        return vfail(TBK_ERR_INVALID, "bad length distribution (n50 %g, sigma %g, short fraction %g, lengths %u..%u)", n50, sigma, short_fraction, min_len, max_len);
    const double mu = std::log(n50) - sigma * sigma;
    const double lo = std::log((double)min_len), hi = std::log(std::max<double>(min_len, std::min<double>(5000.0, max_len)));
    offsets[0] = 0;
    for (uint64_t i = 0; i < n_reads; i++) {
        const uint64_t h1 = tbk_splitmix(seed ^ ((first_read + i) * 0xD1342543DE82EF95ull)), h2 = tbk_splitmix(h1), h3 = tbk_splitmix(h2);
        const double u1 = ((double)(h1 >> 11) + 0.5) / 9007199254740992.0, u2 = ((double)(h2 >> 11) + 0.5) / 9007199254740992.0;
        const double u3 = ((double)(h3 >> 11) + 0.5) / 9007199254740992.0;
        double len;
        if (u3 < short_fraction) len = std::exp(lo + (hi - lo) * u1);
        else len = std::exp(mu + sigma * std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));  // Box-Muller
        uint64_t l = (uint64_t)std::llround(len);
        l = std::max<uint64_t>(min_len, std::min<uint64_t>(max_len, l));
        offsets[i + 1] = offsets[i] + l;
    }
    return TBK_OK;
}

extern "C" int tbk_synth_reads_ragged_device(int device, uint64_t read_seed, uint64_t first_read, uint64_t n_reads, const void *d_offsets, uint64_t total_bases,
                                             uint64_t key_seed, uint64_t n_a, uint64_t n_b, int k, uint32_t slot_len, void *d_bases) {
    if (k < 3 || k > 32 || (n_reads && (!d_offsets || !d_bases))) return vfail(TBK_ERR_INVALID, "bad synth parameters");
    if (((uintptr_t)d_bases & 15) != 0) return vfail(TBK_ERR_INVALID, "d_bases must be 16-byte aligned");
    int rc = on_device(device);
    if (rc) return rc;
    V_TRY(tbk_launch_synth_reads_ragged(read_seed, first_read, n_reads, (const uint64_t *)d_offsets, total_bases, key_seed, n_a, n_b, k, slot_len, (uint8_t *)d_bases, nullptr));
    V_TRY(hipDeviceSynchronize());
    return TBK_OK;
}

extern "C" int tbk_synth_hap_reads_ragged_device(int device, uint64_t seed, uint64_t genome_len, uint32_t snp_per_2p24, uint64_t read_seed, uint64_t first_read,
                                                 uint64_t n_reads, const void *d_offsets, uint64_t total_bases, uint64_t longest_read, uint32_t err_per_2p24, void *d_bases) {
    if (n_reads && (!d_offsets || !d_bases)) return vfail(TBK_ERR_INVALID, "NULL argument");
    if (genome_len < longest_read || !genome_len) return vfail(TBK_ERR_INVALID, "the genome (%llu) is shorter than the longest read (%llu)", (unsigned long long)genome_len, (unsigned long long)longest_read);
    if (((uintptr_t)d_bases & 15) != 0) return vfail(TBK_ERR_INVALID, "d_bases must be 16-byte aligned");
    int rc = on_device(device);
    if (rc) return rc;
    V_TRY(tbk_launch_synth_hap_reads_ragged(seed, genome_len, snp_per_2p24, read_seed, first_read, n_reads, (const uint64_t *)d_offsets, total_bases, err_per_2p24, (uint8_t *)d_bases, nullptr));
    V_TRY(hipDeviceSynchronize());
    return TBK_OK;
}

// ---- the sweep ------------------------------------------------------------------------------------------------------
extern "C" int tbk_synth_mutate_keys_device(int device, const void *d_keys, uint64_t first, uint64_t n, int k, uint64_t seed, void *d_out) {
    if (k < 1 || k > 32 || (n && (!d_keys || !d_out))) return vfail(TBK_ERR_INVALID, "bad arguments");
    int rc = on_device(device);
    if (rc) return rc;
    V_TRY(tbk_launch_mutate_keys((const uint64_t *)d_keys, first, n, k, seed, (uint64_t *)d_out, nullptr));
    V_TRY(hipDeviceSynchronize());
    return TBK_OK;
}

extern "C" int tbk_sweep_expectation_device(tbk_table *a, tbk_table *b, const void *d_keys, uint64_t n, void *d_expect) {
    if (!a || !b || (n && (!d_keys || !d_expect))) return vfail(TBK_ERR_INVALID, "NULL argument");
    if (tbk_table_k(a) != tbk_table_k(b) || tbk_table_device(a) != tbk_table_device(b)) return vfail(TBK_ERR_INVALID, "the lists differ in k or device");
    int rc = on_device(tbk_table_device(a));
    if (rc || !n) return rc;
    uint64_t *d_canon = nullptr;
    uint8_t *d_in = nullptr;
    V_TRY(hipMalloc((void **)&d_canon, n * 8));
    hipError_t e = hipMalloc((void **)&d_in, 2 * n);
    if (e == hipSuccess) e = tbk_launch_canonical_keys((const uint64_t *)d_keys, n, tbk_table_k(a), d_canon, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) {
        rc = tbk_table_contains_device(a, d_canon, n, d_in);
        if (!rc) rc = tbk_table_contains_device(b, d_canon, n, d_in + n);
        if (!rc) {
            e = tbk_launch_sweep_expect(d_in, d_in + n, n, (uint8_t *)d_expect, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
    }
    (void)hipFree(d_canon);
    if (d_in) (void)hipFree(d_in);
    if (rc) return rc;
    if (e != hipSuccess) return vfail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "tbk_sweep_expectation_device: %s", hipGetErrorString(e));
    return TBK_OK;
}

extern "C" int tbk_classifier_sweep_keys(tbk_classifier *c, const void *d_keys, uint64_t n, int k, uint32_t keys_per_read, int expect, const void *d_expect,
                                         uint64_t chunk_keys, uint64_t out[4]) {
    if (!c || !out || (n && !d_keys)) return vfail(TBK_ERR_INVALID, "NULL argument");
    if (keys_per_read < 1 || expect < 0 || expect > 2 || k < 1 || k > 32) return vfail(TBK_ERR_INVALID, "bad sweep parameters");
    out[0] = out[1] = out[2] = 0; out[3] = ~0ull;
    int rc = on_device(tbk_classifier_device(c));
    if (rc || !n) return rc;
    if (!chunk_keys) chunk_keys = (uint64_t)1 << 25;
    chunk_keys = std::max<uint64_t>(keys_per_read, chunk_keys / keys_per_read * keys_per_read);  // (whole reads per chunk)
    chunk_keys = std::min(chunk_keys, (n + keys_per_read - 1) / keys_per_read * keys_per_read);
    const uint64_t reads_cap = chunk_keys / keys_per_read, bases_cap = chunk_keys * ((uint64_t)k + 1);
    uint8_t *d_bases = nullptr;
    uint64_t *d_offsets = nullptr;
    int32_t *d_counts = nullptr;
    unsigned long long *d_out = nullptr, h_out[4] = {0, 0, 0, ~0ull};
    hipError_t e = hipMalloc((void **)&d_bases, bases_cap + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&d_offsets, (reads_cap + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_counts, reads_cap * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof h_out);
    if (e == hipSuccess) e = hipMemcpy(d_out, h_out, sizeof h_out, hipMemcpyHostToDevice);
    for (uint64_t first = 0; first < n && e == hipSuccess && !rc; first += chunk_keys) {
        const uint64_t nk = std::min(chunk_keys, n - first), nr = (nk + keys_per_read - 1) / keys_per_read;
        const uint64_t total = (nr - 1) * ((uint64_t)keys_per_read * k + keys_per_read - 1) + (nk - (nr - 1) * keys_per_read) * ((uint64_t)k + 1) - 1;
        e = tbk_launch_keys_as_reads((const uint64_t *)d_keys + first, first, nk, k, keys_per_read, d_bases, d_offsets, nr, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();  // (the generator runs on the null stream, the probe on the classifier's)
        if (e != hipSuccess) break;
        rc = tbk_classify_device(c, d_bases, d_offsets, nr, total, d_counts);
        if (!rc) rc = tbk_classifier_sync(c);
        if (rc) break;
        e = tbk_launch_counts_check(d_counts, first / keys_per_read, nr, keys_per_read, nk, expect, d_expect ? (const uint8_t *)d_expect + first : nullptr, d_out, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (e == hipSuccess && !rc) e = hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    if (d_bases) (void)hipFree(d_bases);
    if (d_offsets) (void)hipFree(d_offsets);
    if (d_counts) (void)hipFree(d_counts);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return vfail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "tbk_classifier_sweep_keys: %s", hipGetErrorString(e));
    for (int i = 0; i < 4; i++) out[i] = h_out[i];
    return TBK_OK;
}

// Every line of both lists through the finished table, against the lists' standalone tables (verbatim keys): what a build can
// be checked by in the field - a concurrent build (CAS on slots, the wide entries' per-piece lock) leaves no other trace of a
// lost or misfiled key.  Chunks of 2^24 keys; out = {lines checked, keys counted for hapA, for hapB, lines that differ, the first
// such line (hapA's lines first, then hapB's; all ones: none)}.  Under a second at 2 x 3e8 keys.
// Two halves, so that a replica on ANOTHER device (which has the lists' keys but not their standalone tables) is checked too:
// the expectations - one byte per list line, hapA's lines first - are written once on the lists' device ...
extern "C" int tbk_verify_expectations_(tbk_table *a, tbk_table *b, void *d_expect) {
    if (!a || !b || !d_expect) return vfail(TBK_ERR_INVALID, "NULL argument");
    if (tbk_table_k(a) != tbk_table_k(b) || tbk_table_device(a) != tbk_table_device(b)) return vfail(TBK_ERR_INVALID, "the lists differ in k or device");
    const uint64_t chunk = (uint64_t)1 << 24;
    uint64_t base = 0;
    int rc = TBK_OK;
    for (tbk_table *t : {a, b}) {
        const uint64_t n = tbk_table_num_kmers(t);
        const uint64_t *d_keys = (const uint64_t *)tbk_table_device_keys(t);
        for (uint64_t first = 0; first < n && !rc; first += chunk)
            rc = tbk_sweep_expectation_device(a, b, d_keys + first, std::min(chunk, n - first), (uint8_t *)d_expect + base + first);
        base += n;
    }
    return rc;
}

// ... and the classifier answers for the keys where IT lives (d_keys_a / _b and d_expect on the classifier's device).
extern "C" int tbk_classifier_verify_expect_(tbk_classifier *c, const void *d_keys_a, uint64_t na, const void *d_keys_b, uint64_t nb, int k, const void *d_expect,
                                             uint64_t out[5]) {
    if (!c || !out || !d_expect) return vfail(TBK_ERR_INVALID, "NULL argument");
    out[0] = out[1] = out[2] = out[3] = 0; out[4] = ~0ull;
    const uint64_t chunk = (uint64_t)1 << 24;
    uint64_t base = 0;
    int rc = TBK_OK;
    for (int list = 0; list < 2 && !rc; list++) {
        const uint64_t n = list ? nb : na;
        const uint64_t *d_keys = (const uint64_t *)(list ? d_keys_b : d_keys_a);
        uint64_t r[4];
        if (n) rc = tbk_classifier_sweep_keys(c, d_keys, n, k, 1, 0, (const uint8_t *)d_expect + base, chunk, r);
        if (rc || !n) { base += n; continue; }
        out[0] += n; out[1] += r[0]; out[2] += r[1]; out[3] += r[2];
        if (out[4] == ~0ull && r[3] != ~0ull) out[4] = base + r[3];
        base += n;
    }
    return rc;
}

extern "C" int tbk_classifier_verify(tbk_classifier *c, tbk_table *a, tbk_table *b, uint64_t out[5]) {
    if (!c || !a || !b || !out) return vfail(TBK_ERR_INVALID, "NULL argument");
    const int k = tbk_table_k(a);
    if (k != tbk_table_k(b) || tbk_table_device(a) != tbk_classifier_device(c) || tbk_table_device(b) != tbk_classifier_device(c))
        return vfail(TBK_ERR_INVALID, "the lists and the classifier do not belong together");
    int rc = on_device(tbk_classifier_device(c));
    if (rc) return rc;
    out[0] = out[1] = out[2] = out[3] = 0; out[4] = ~0ull;
    const uint64_t na = tbk_table_num_kmers(a), nb = tbk_table_num_kmers(b);
    uint8_t *d_expect = nullptr;
    V_TRY(hipMalloc((void **)&d_expect, na + nb + 8));
    rc = tbk_verify_expectations_(a, b, d_expect);
    if (!rc) rc = tbk_classifier_verify_expect_(c, tbk_table_device_keys(a), na, tbk_table_device_keys(b), nb, k, d_expect, out);
    (void)hipFree(d_expect);
    return rc;
}
