// tbk_count.cpp — host side of the k-mer counter behind find-unique-kmers.
//
// The reference's find_unique_kmers.py is subprocess glue around KMC 3 (kmc, kmc_tools, kmc_dump;
// find_unique_kmers.py:62-233).  KMC is not part of the reference checkout, so what is restated
// here is its published behaviour at the call sites' settings:
//   kmc -k<k> -t<n> @files db tmp     canonical k-mers of all reads, both strands; k-mers holding
//                                     a symbol outside ACGT are skipped; defaults -ci2 (k-mers seen
//                                     once are not stored), -cs255 (counters saturate at 255)
//   kmc_tools transform db histogram  one row per counter value: value <tab> number of k-mers
//   kmc_tools simple A B kmers_subtract   k-mers of A that B does not hold
//   kmc_dump -ci<a> -cx<b> db out     k-mers with a <= counter <= b, in lexicographic order
// Parity with KMC itself is unpinned (no binary, no golden output in the reference); the tests
// check this code against a CPU restatement of the list above.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/tbk.h"
#include "tbk_common.h"

extern "C" void tbk_set_error_(int code, const char *msg);
extern "C" hipError_t tbk_launch_separate(const uint8_t *, const uint64_t *, uint64_t, uint8_t *, hipStream_t);
extern "C" hipError_t tbk_launch_count(const uint8_t *, uint64_t, uint64_t, uint64_t, int, uint64_t *, uint32_t, TbkMz, int *, unsigned long long *,
                                       hipStream_t);
extern "C" uint64_t tbk_probe_passes(uint64_t total);
extern "C" hipError_t tbk_launch_count_clamp(uint64_t *, uint32_t, TbkMz, hipStream_t);
extern "C" hipError_t tbk_launch_count_rehash(uint64_t *, uint32_t, TbkMz, uint64_t *, uint32_t, TbkMz, int *, hipStream_t);
extern "C" hipError_t tbk_launch_count_histogram(uint64_t *, uint32_t, TbkMz, unsigned long long *, hipStream_t);
extern "C" hipError_t tbk_launch_count_unique(uint64_t *, uint32_t, TbkMz, uint64_t *, uint32_t, TbkMz, int, uint32_t, uint32_t,
                                              uint64_t *, uint64_t, unsigned long long *, hipStream_t);
extern "C" hipError_t tbk_launch_sort_u64(const uint64_t *, uint64_t *, uint64_t, int, hipStream_t);

static int cfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tbk_set_error_(code, buf);
    return code;
}
#define CHIP(expr)                                                                                                      \
    do {                                                                                                                \
        hipError_t e_ = (expr);                                                                                         \
        if (e_ != hipSuccess)                                                                                           \
            return cfail(e_ == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct tbk_counter {
    int device = 0, k = 0;
    uint64_t *d_lines = nullptr;  // n_buckets lines of 128 B: 8 keys | 8 x 32-bit counters | 32 spare bytes
    uint32_t n_buckets = 0;
    TbkMz mz{0, 0, 0, 0};
    int *d_failed = nullptr;
    unsigned long long *d_used = nullptr;  // [0] slots taken so far (distinct k-mers met), [1 .. 1024] tallies of the 64-bit atomic adds issued: kept by the kernels
    uint64_t since_clamp = 0;              // window starts counted since the counters were last held below 2^31 (tbk_count_clamp_kernel)
    uint64_t used = 0;
    double load = 0.6;
    // staging of one batch: reads back to back, their offsets, and the separated upper-cased copy
    uint8_t *d_raw = nullptr, *d_sep = nullptr;
    uint64_t *d_off = nullptr;
    size_t cap_raw = 0, cap_sep = 0, cap_reads = 0;
    uint64_t bases_added = 0, reads_added = 0;
    // HIP-event timing of the counting kernel (every launch; read by tbk_counter_kernel_timing)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t timed_launches = 0, timed_windows = 0;
    double timed_ms = 0.0;
};

static int counter_device(const tbk_counter *c) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return cfail(TBK_ERR_NO_DEVICE, "no HIP device visible; libtbk_hip has no CPU fallback");
    CHIP(hipSetDevice(c->device));
    return TBK_OK;
}

// lines for `capacity` distinct k-mers at the counter's target load; keys = all ones (free), counters = 0
static int alloc_lines(int k, uint64_t capacity, double load, uint64_t **d_lines, uint32_t *n_buckets, TbkMz *mz) {
    uint64_t nb = (uint64_t)((double)capacity / (TBK_SLOTS_PER_BUCKET * load)) + 16;
    if (nb > 0x7FFFFFF0ull) return cfail(TBK_ERR_NOMEM, "%llu distinct k-mers are more than one table holds", (unsigned long long)capacity);
    const size_t bytes = (size_t)nb * 128;
    hipError_t e = hipMalloc((void **)d_lines, bytes);
    if (e == hipSuccess) e = hipMemset2D(*d_lines, 128, 0xFF, 64, nb);
    if (e == hipSuccess) e = hipMemset2D((uint8_t *)*d_lines + 64, 128, 0, 64, nb);
    if (e != hipSuccess) {
        if (*d_lines) (void)hipFree(*d_lines);
        *d_lines = nullptr;
        (void)hipGetLastError();
        return cfail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "counting table for %llu k-mers (%zu bytes): %s",
                     (unsigned long long)capacity, bytes, hipGetErrorString(e));
    }
    *n_buckets = (uint32_t)nb;
    const char *ew = getenv("TBK_COUNT_W"), *em = getenv("TBK_COUNT_M");
    *mz = tbk_mz_params(k, ew ? atoi(ew) : 6, capacity, em ? atoi(em) : 0, 0);
    return TBK_OK;
}

extern "C" int tbk_counter_create(int k, uint64_t capacity_kmers, int device, tbk_counter **out) {
    if (!out) return cfail(TBK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (k < 1 || k > 32) return cfail(TBK_ERR_INVALID, "k = %d out of range (1..32)", k);
    if (!capacity_kmers) return cfail(TBK_ERR_INVALID, "capacity is 0");
    tbk_counter tmp;
    tmp.device = device;
    int rc = counter_device(&tmp);
    if (rc) return rc;
    // 8-slot lines at load <= 0.6.  A counting table meets every distinct k-mer of the reads,
    // sequencing errors included; the capacity is the caller's estimate, and the table is rebuilt
    // twice as large whenever the next batch could fill it.
    tbk_counter *c = new tbk_counter();
    c->device = device; c->k = k;
    const char *ev = getenv("TBK_COUNT_LOAD");
    c->load = ev ? atof(ev) : 0.6;
    if (c->load < 0.05) c->load = 0.05;
    if (c->load > 0.9) c->load = 0.9;
    rc = alloc_lines(k, capacity_kmers, c->load, &c->d_lines, &c->n_buckets, &c->mz);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMalloc((void **)&c->d_failed, sizeof(int));
    if (!rc && e == hipSuccess) e = hipMalloc((void **)&c->d_used, 1025 * sizeof(unsigned long long));
    if (!rc && e == hipSuccess) e = hipMemset(c->d_failed, 0, sizeof(int));
    if (!rc && e == hipSuccess) e = hipMemset(c->d_used, 0, 1025 * sizeof(unsigned long long));
    if (!rc && e != hipSuccess) rc = cfail(TBK_ERR_HIP, "tbk_counter_create: %s", hipGetErrorString(e));
    if (rc) { tbk_counter_destroy(c); return rc; }
    *out = c;
    return TBK_OK;
}

// Rebuild the table for `capacity` distinct k-mers and move every (key, counter) over.
static int counter_grow(tbk_counter *c, uint64_t capacity) {
    uint64_t *d_new = nullptr;
    uint32_t nb = 0;
    TbkMz mz{0, 0, 0, 0};
    int rc = alloc_lines(c->k, capacity, c->load, &d_new, &nb, &mz);
    if (rc) return rc;
    hipError_t e = tbk_launch_count_rehash(c->d_lines, c->n_buckets, c->mz, d_new, nb, mz, c->d_failed, nullptr);
    int failed = 0;
    if (e == hipSuccess) e = hipMemcpy(&failed, c->d_failed, sizeof failed, hipMemcpyDeviceToHost);
    if (e != hipSuccess || failed) {
        (void)hipFree(d_new);
        return cfail(TBK_ERR_HIP, "counting table rebuild failed: %s", e != hipSuccess ? hipGetErrorString(e) : "new table full");
    }
    (void)hipFree(c->d_lines);
    c->d_lines = d_new; c->n_buckets = nb; c->mz = mz;
    return TBK_OK;
}

extern "C" void tbk_counter_destroy(tbk_counter *c) {
    if (!c) return;
    if (hipSetDevice(c->device) == hipSuccess) {
        (void)hipDeviceSynchronize();
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        for (void *p : {(void *)c->d_lines, (void *)c->d_failed, (void *)c->d_used, (void *)c->d_raw, (void *)c->d_sep, (void *)c->d_off})
            if (p) (void)hipFree(p);
    }
    delete c;
}

static int counter_run(tbk_counter *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t total) {
    const size_t need_sep = (size_t)total + n_reads + 64;
    if (need_sep > c->cap_sep) {
        if (c->d_sep) CHIP(hipFree(c->d_sep));
        c->d_sep = nullptr; c->cap_sep = 0;
        CHIP(hipMalloc((void **)&c->d_sep, need_sep + need_sep / 8));
        c->cap_sep = need_sep + need_sep / 8;
    }
    CHIP(tbk_launch_separate(d_bases, d_offsets, n_reads, c->d_sep, nullptr));
    // The stream is counted in pieces of a quarter of the table's slots (at least 64 M window
    // starts).  Every window of a piece could be a k-mer never seen before, so before a piece that
    // could fill the table the table is rebuilt twice as large - a piece can then never run out of
    // room half way, and the table grows once its load passes 0.6.
    const uint64_t sep_total = total + n_reads, passes = tbk_probe_passes(sep_total);
    for (uint64_t p0 = 0; p0 < passes;) {
        uint64_t slots = (uint64_t)c->n_buckets * TBK_SLOTS_PER_BUCKET;
        const uint64_t piece = std::max<uint64_t>(32768, slots / 4 / 2048);
        const uint64_t np = std::min(piece, passes - p0), windows = np * 2048;
        if ((double)(c->used + windows) > 0.85 * (double)slots) {
            uint64_t want = (uint64_t)((double)slots * c->load) * 2;  // twice the present capacity
            while ((double)(c->used + windows) > 0.85 * ((double)want / c->load)) want *= 2;
            const int rc = counter_grow(c, want);
            if (rc) return rc;
            slots = (uint64_t)c->n_buckets * TBK_SLOTS_PER_BUCKET;
        }
        // no counter may pass 2^32 (it would carry into the neighbour it shares a 64-bit word with): one grows by at most a
        // piece's window starts, so before 2^31 of them have been added since the last clamp, every counter goes back to <= 2^31
        if (c->since_clamp + windows >= ((uint64_t)1 << 31)) {
            CHIP(tbk_launch_count_clamp(c->d_lines, c->n_buckets, c->mz, nullptr));
            c->since_clamp = 0;
        }
        c->since_clamp += windows;
        if (!c->ev0) { CHIP(hipEventCreate(&c->ev0)); CHIP(hipEventCreate(&c->ev1)); }
        CHIP(hipEventRecord(c->ev0, nullptr));
        CHIP(tbk_launch_count(c->d_sep, sep_total, p0, np, c->k, c->d_lines, c->n_buckets, c->mz, c->d_failed, c->d_used, nullptr));
        CHIP(hipEventRecord(c->ev1, nullptr));
        int failed = 0;
        unsigned long long used = 0;
        CHIP(hipMemcpy(&failed, c->d_failed, sizeof failed, hipMemcpyDeviceToHost));
        {
            float ms = 0;
            CHIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
            c->timed_ms += ms; c->timed_launches++; c->timed_windows += std::min<uint64_t>(windows, sep_total - p0 * 2048);
        }
        CHIP(hipMemcpy(&used, c->d_used, sizeof used, hipMemcpyDeviceToHost));
        c->used = used;
        if (failed) return cfail(TBK_ERR_NOMEM, "counting table is full (%llu slots, %llu taken)", (unsigned long long)slots, used);
        p0 += np;
    }
    c->bases_added += total;
    c->reads_added += n_reads;
    return TBK_OK;
}

extern "C" int tbk_check_offsets_(const uint64_t *offsets, uint64_t n_reads);

extern "C" int tbk_counter_add_batch(tbk_counter *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads) {
    if (!c || (n_reads && (!bases || !offsets))) return cfail(TBK_ERR_INVALID, "NULL argument");
    if (!n_reads) return TBK_OK;
    int rc = tbk_check_offsets_(offsets, n_reads);  // same rule as tbk_stream_submit
    if (rc) return rc;
    rc = counter_device(c);
    if (rc) return rc;
    const uint64_t total = offsets[n_reads];
    if (total + 16 > c->cap_raw) {
        if (c->d_raw) CHIP(hipFree(c->d_raw));
        c->d_raw = nullptr; c->cap_raw = 0;
        CHIP(hipMalloc((void **)&c->d_raw, total + total / 8 + 64));
        c->cap_raw = total + total / 8 + 64;
    }
    if (n_reads + 1 > c->cap_reads) {
        if (c->d_off) CHIP(hipFree(c->d_off));
        c->d_off = nullptr; c->cap_reads = 0;
        CHIP(hipMalloc((void **)&c->d_off, (n_reads + n_reads / 8 + 2) * sizeof(uint64_t)));
        c->cap_reads = n_reads + n_reads / 8 + 2;
    }
    CHIP(hipMemcpy(c->d_raw, bases, total, hipMemcpyHostToDevice));
    CHIP(hipMemcpy(c->d_off, offsets, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    return counter_run(c, c->d_raw, c->d_off, n_reads, total);
}

extern "C" int tbk_counter_add_device(tbk_counter *c, const void *d_bases, const void *d_offsets, uint64_t n_reads, uint64_t total_bases) {
    if (!c || (n_reads && (!d_bases || !d_offsets))) return cfail(TBK_ERR_INVALID, "NULL argument");
    if (!n_reads) return TBK_OK;
    int rc = counter_device(c);
    if (rc) return rc;
    return counter_run(c, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_reads, total_bases);
}

extern "C" int tbk_counter_kernel_timing(tbk_counter *c, uint64_t *launches, uint64_t *window_starts, double *total_ms, int reset) {
    if (!c) return cfail(TBK_ERR_INVALID, "counter is NULL");
    if (launches) *launches = c->timed_launches;
    if (window_starts) *window_starts = c->timed_windows;
    if (total_ms) *total_ms = c->timed_ms;
    if (reset) { c->timed_launches = 0; c->timed_windows = 0; c->timed_ms = 0.0; }
    return TBK_OK;
}

extern "C" int tbk_counter_adds_issued(tbk_counter *c, uint64_t *adds) {
    if (!c || !adds) return cfail(TBK_ERR_INVALID, "NULL argument");
    int rc = counter_device(c);
    if (rc) return rc;
    std::vector<unsigned long long> v(1024);
    CHIP(hipMemcpy(v.data(), c->d_used + 1, v.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    uint64_t sum = 0;
    for (unsigned long long x : v) sum += x;
    *adds = sum;
    return TBK_OK;
}

extern "C" int tbk_counter_histogram(tbk_counter *c, uint64_t hist[256]) {
    if (!c || !hist) return cfail(TBK_ERR_INVALID, "NULL argument");
    int rc = counter_device(c);
    if (rc) return rc;
    unsigned long long *d_hist = nullptr;
    CHIP(hipMalloc((void **)&d_hist, 256 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d_hist, 0, 256 * sizeof(unsigned long long));
    if (e == hipSuccess) e = tbk_launch_count_histogram(c->d_lines, c->n_buckets, c->mz, d_hist, nullptr);
    unsigned long long h[256];
    if (e == hipSuccess) e = hipMemcpy(h, d_hist, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d_hist);
    if (e != hipSuccess) return cfail(TBK_ERR_HIP, "tbk_counter_histogram: %s", hipGetErrorString(e));
    for (int i = 0; i < 256; i++) hist[i] = h[i];
    return TBK_OK;
}

extern "C" int tbk_counter_distinct(const tbk_counter *c, uint64_t *distinct) {
    if (!c || !distinct) return cfail(TBK_ERR_INVALID, "NULL argument");
    *distinct = c->used;
    return TBK_OK;
}

extern "C" int tbk_counter_stats(const tbk_counter *c, uint64_t *n_slots, uint64_t *table_bytes, uint64_t *bases_added, uint64_t *reads_added) {
    if (!c) return cfail(TBK_ERR_INVALID, "counter is NULL");
    if (n_slots) *n_slots = (uint64_t)c->n_buckets * TBK_SLOTS_PER_BUCKET;
    if (table_bytes) *table_bytes = (uint64_t)c->n_buckets * 128;
    if (bases_added) *bases_added = c->bases_added;
    if (reads_added) *reads_added = c->reads_added;
    return TBK_OK;
}

// one k-mer per line, k characters + '\n', from lexicographic ranks (base 0 in the top bits)
static bool write_list(const char *path, const uint64_t *lex, uint64_t n, int k, std::string &err) {
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { err = std::string("cannot create ") + path + ": " + strerror(errno); return false; }
    const size_t line = (size_t)k + 1;
    const uint64_t per = (uint64_t)1 << 18;  // lines per piece
    const uint64_t pieces = (n + per - 1) / per;
    std::atomic<uint64_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
        std::vector<char> buf;
        for (uint64_t p; (p = next.fetch_add(1)) < pieces && ok.load();) {
            const uint64_t lo = p * per, hi = std::min(n, lo + per);
            buf.resize((size_t)(hi - lo) * line);
            char *w = buf.data();
            for (uint64_t i = lo; i < hi; i++) {
                const uint64_t v = lex[i];
                for (int b = 0; b < k; b++) *w++ = "ACGT"[(v >> (2 * (k - 1 - b))) & 3u];
                *w++ = '\n';
            }
            size_t done = 0;
            while (done < buf.size()) {
                const ssize_t r = ::pwrite(fd, buf.data() + done, buf.size() - done, (off_t)(lo * line + done));
                if (r <= 0) { ok.store(false); break; }
                done += (size_t)r;
            }
        }
    };
    const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)tbk_host_threads(), pieces));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    const bool closed = ::close(fd) == 0;
    if (!ok.load() || !closed) { err = std::string("write failed: ") + path; return false; }
    return true;
}

extern "C" int tbk_counter_unique(tbk_counter *a, tbk_counter *b, uint32_t min_count, uint32_t max_count, const char *out_path,
                                  uint64_t *n_written) {
    if (!a || !b || !out_path || !n_written) return cfail(TBK_ERR_INVALID, "NULL argument");
    if (a->k != b->k) return cfail(TBK_ERR_INVALID, "the counters have different k (%d and %d)", a->k, b->k);
    if (a->device != b->device) return cfail(TBK_ERR_INVALID, "the counters live on different devices");
    *n_written = 0;
    int rc = counter_device(a);
    if (rc) return rc;
    // upper bound of what can come out: k-mers of A with a counter in range
    uint64_t hist[256];
    rc = tbk_counter_histogram(a, hist);
    if (rc) return rc;
    uint64_t cap = 0;
    for (uint32_t cnt = std::max<uint32_t>(2, min_count); cnt <= std::min<uint32_t>(255, max_count); cnt++) cap += hist[cnt];
    uint64_t n = 0;
    std::vector<uint64_t> h_keys;
    if (cap) {
        uint64_t *d_out = nullptr, *d_sorted = nullptr;
        unsigned long long *d_n = nullptr, got = 0;
        hipError_t e = hipMalloc((void **)&d_out, cap * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&d_sorted, cap * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&d_n, sizeof got);
        if (e == hipSuccess) e = hipMemset(d_n, 0, sizeof got);
        if (e == hipSuccess)
            e = tbk_launch_count_unique(a->d_lines, a->n_buckets, a->mz, b->d_lines, b->n_buckets, b->mz, a->k, min_count, max_count,
                                        d_out, cap, d_n, nullptr);
        if (e == hipSuccess) e = hipMemcpy(&got, d_n, sizeof got, hipMemcpyDeviceToHost);
        n = std::min<uint64_t>(got, cap);
        if (e == hipSuccess && n) e = tbk_launch_sort_u64(d_out, d_sorted, n, 2 * a->k, nullptr);
        if (e == hipSuccess && n) {
            h_keys.resize(n);
            e = hipMemcpy(h_keys.data(), d_sorted, n * sizeof(uint64_t), hipMemcpyDeviceToHost);
        }
        if (d_out) (void)hipFree(d_out);
        if (d_sorted) (void)hipFree(d_sorted);
        if (d_n) (void)hipFree(d_n);
        if (e != hipSuccess) return cfail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "tbk_counter_unique: %s", hipGetErrorString(e));
    }
    std::string err;
    if (!write_list(out_path, h_keys.data(), n, a->k, err)) return cfail(TBK_ERR_IO, "%s", err.c_str());
    *n_written = n;
    return TBK_OK;
}
