// tbk_host.cpp — host side of libtbk_hip.so: the C-ABI declared in include/tbk.h.
//
// Owns device memory, streams and events directly through the HIP runtime (no PyTorch):
// k-mer tables resident in HBM, a ring of pinned/device staging slots so the H2D copy of
// batch i+1 (side stream) overlaps the probe kernel of batch i (compute stream), and the
// text-list parser that mirrors the reference's getline() rules.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <pthread.h>
#include <sched.h>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tbk.h"
#include "tbk_common.h"
#include "tbk_pack.h"

// ---- kernels' launchers (tbk_kernels.hip, tbk_synth.hip) -------------------------------
extern "C" hipError_t tbk_launch_order(uint64_t *, uint64_t, const uint32_t *, const uint32_t *, uint32_t, uint32_t, hipStream_t);
extern "C" hipError_t tbk_launch_insert(uint64_t *, uint32_t, uint32_t, uint32_t, TbkMz, const uint64_t *, uint64_t, uint32_t *, uint32_t *, uint32_t, TbkTableView,
                                        unsigned long long *, unsigned long long *, int *, hipStream_t);
extern "C" hipError_t tbk_launch_contains(TbkTableView, const uint64_t *, uint64_t, uint8_t *, hipStream_t);
extern "C" hipError_t tbk_launch_entry_insert(uint64_t *, uint32_t, uint32_t, TbkMz, int, const uint64_t *, uint64_t, int, int, unsigned long long *, int *, hipStream_t);
extern "C" hipError_t tbk_launch_short_insert(uint64_t *, uint32_t, uint32_t, uint32_t, TbkMz, int, const uint64_t *, uint64_t, int, unsigned long long *, int *, uint32_t, hipStream_t);
extern "C" hipError_t tbk_launch_full_insert(uint64_t *, uint32_t, uint32_t, TbkMz, int, const uint64_t *, uint64_t, int, unsigned long long *, int *, uint32_t, hipStream_t);
extern "C" hipError_t tbk_launch_probe_index(const uint64_t *, uint64_t, uint64_t, int32_t *, uint32_t *, uint64_t, int, hipStream_t);
extern "C" int tbk_probe_has_two_read_kernel(TbkMz);
extern "C" hipError_t tbk_launch_probe_range(const uint8_t *, const uint32_t *, const uint16_t *, const uint64_t *, uint64_t, uint64_t, TbkPairView, int,
                                             int32_t *, uint32_t *, uint64_t, uint64_t, uint64_t, int, int, hipEvent_t, hipStream_t, int);
extern "C" hipError_t tbk_launch_scatter_bad(const uint32_t *, const uint16_t *, uint64_t, uint16_t *, uint64_t, int, hipStream_t);
extern "C" uint64_t tbk_packed_chunks(uint64_t total_bases);
extern "C" uint64_t tbk_probe_passes(uint64_t total);
static constexpr uint64_t TBK_PASS_BASES = 2048;  // window starts per pass (tbk_device.h: TBK_PASS)
extern "C" hipError_t tbk_launch_synth_keys(uint64_t, uint64_t, uint64_t, int, uint64_t *, hipStream_t);
extern "C" hipError_t tbk_launch_synth_reads(uint64_t, uint64_t, uint64_t, uint32_t, uint64_t, uint64_t, uint64_t,
                                             int, int, int, uint8_t *, uint64_t *, hipStream_t);
extern "C" hipError_t tbk_launch_fill(void *, uint64_t, uint64_t, hipStream_t);
extern "C" hipError_t tbk_launch_stream(const void *, uint64_t, uint32_t *, hipStream_t);
extern "C" hipError_t tbk_launch_gather(const void *, uint64_t, int, int, int, uint64_t, uint64_t, uint32_t *,
                                        uint64_t *, hipStream_t);

// ---- errors ----------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// lets the other translation units of the library report through tbk_last_error()
extern "C" void tbk_set_error_(int, const char *msg) { g_err = msg ? msg : ""; }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                               \
    } while (0)

static int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(TBK_ERR_NO_DEVICE, "no HIP device visible (%s); libtbk_hip has no CPU fallback",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(TBK_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    return TBK_OK;
}

static double env_double(const char *name, double dflt) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    double d = strtod(v, &end);
    return end == v ? dflt : d;
}

// ---- options of a classifier (include/tbk.h: tbk_options) ----------------------------------------------------------
// Nothing under tbk_classifier_create* / tbk_pipeline_create* reads the environment: what can be chosen is in the struct.
// tbk_options_from_env is the command-line tools' fallback (the reference configures itself by argparse alone,
// classify_by_kmers.py:14-54; the TBK_* variables are this build's experiment knobs) and is called by the caller, never here.
extern "C" void tbk_options_init(tbk_options *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->size = (uint32_t)sizeof *o;
    o->short_keys = o->entries = o->wide_entries = o->front = -1;
    o->full_keys = -1;
    o->full_load = 2.0;
    o->mod_sampling = o->span3 = o->guests = -1;
    o->minimizer_w = -1;
    o->minimizer_m = 0;
    o->two_read_kernel = 1;
    o->table_load = 0.0;  // 0: the layouts' own (0.08 for the key layouts), capped by the memory budget
    o->entry_load = 0.5; o->wentry_load = 0.25; o->short_load = 2.3;
    o->clustered = 0.003; o->behind_front = 0.05; o->plainly_clustered = 0.12; o->entry_min_ratio = 1.5;
    o->memory_budget_bytes = 0;  // 0: 60 % of the device's total memory
    o->short_line_cap = 32;
    o->packed_h2d = 1;
    o->slice_bases = (uint64_t)384 << 20;
    o->h2d_streams = 1;
    o->verify_build = -1;
}

extern "C" int tbk_options_from_env(tbk_options *o) {
    if (!o) return fail(TBK_ERR_INVALID, "options is NULL");
    if (o->size != sizeof *o) tbk_options_init(o);
    auto tri = [](const char *name, int32_t &v) { const double d = env_double(name, -1); if (getenv(name) && *getenv(name)) v = d > 0 ? 1 : d == 0 ? 0 : -1; };
    auto num = [](const char *name, double &v) { if (getenv(name) && *getenv(name)) v = env_double(name, v); };
    auto inum = [](const char *name, int32_t &v) { if (getenv(name) && *getenv(name)) v = (int32_t)env_double(name, v); };
    auto unum = [](const char *name, uint64_t &v) { if (getenv(name) && *getenv(name)) v = (uint64_t)std::max(0.0, env_double(name, (double)v)); };
    tri("TBK_SHORT", o->short_keys); tri("TBK_ENTRY", o->entries); tri("TBK_ENTRY_WIDE", o->wide_entries); tri("TBK_FRONT", o->front);
    tri("TBK_FULL", o->full_keys); num("TBK_FULL_LOAD", o->full_load);
    tri("TBK_MOD_SAMPLING", o->mod_sampling); tri("TBK_SPAN3", o->span3); tri("TBK_GUESTS", o->guests);
    inum("TBK_MINIMIZER_W", o->minimizer_w); inum("TBK_MINIMIZER_M", o->minimizer_m); inum("TBK_TWO_READ", o->two_read_kernel);
    num("TBK_TABLE_LOAD", o->table_load); num("TBK_ENTRY_LOAD", o->entry_load); num("TBK_WENTRY_LOAD", o->wentry_load); num("TBK_SHORT_LOAD", o->short_load);
    num("TBK_CLUSTERED", o->clustered); num("TBK_BEHIND_FRONT", o->behind_front); num("TBK_PLAINLY_CLUSTERED", o->plainly_clustered);
    num("TBK_ENTRY_MIN_RATIO", o->entry_min_ratio);
    unum("TBK_MEMORY_BUDGET", o->memory_budget_bytes); unum("TBK_TABLE_ALIGN", o->table_align); unum("TBK_SLICE_BASES", o->slice_bases);
    { uint64_t cap = o->short_line_cap; unum("TBK_SHORT_LINE_CAP", cap); o->short_line_cap = (uint32_t)cap; }
    inum("TBK_PROBE_MAX_BLOCKS", o->probe_max_blocks); inum("TBK_PACKED_H2D", o->packed_h2d); inum("TBK_BUILD_TIMING", o->build_timing);
    inum("TBK_FORCE_REPLICA", o->force_replica); inum("TBK_RING_STREAMS", o->ring_streams); inum("TBK_COPY_PRIORITY", o->copy_priority);
    inum("TBK_H2D_STREAMS", o->h2d_streams); inum("TBK_ZERO_COPY", o->zero_copy);
    inum("TBK_REPLICA_COPY", o->replica_copy); tri("TBK_VERIFY_BUILD", o->verify_build);
    return TBK_OK;
}

// the caller's options (NULL: the defaults; a shorter struct of an older caller: its fields over the defaults)
static tbk_options resolve_options(const tbk_options *in) {
    tbk_options o;
    tbk_options_init(&o);
    if (in && in->size >= 8) memcpy(&o, in, std::min<size_t>(in->size, sizeof o));
    o.size = (uint32_t)sizeof o;
    return o;
}

// Host threads worth starting: hardware threads, cut down to the CPU affinity mask and to the
// cgroup CPU quota (a container may see 256 hardware threads and be allowed 16 CPUs' worth of
// time; more runnable threads than that only adds throttling stalls).  TBK_HOST_THREADS overrides.
static long local_ranks() {
    long ranks = (long)env_double("TBK_LOCAL_RANKS", 0);
    if (ranks < 1) ranks = (long)env_double("LOCAL_WORLD_SIZE", 1);
    return std::max<long>(1, ranks);
}

static long usable_cpus() {
    long n = (long)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = std::min<long>(n, CPU_COUNT(&set));
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
    }
    if (quota > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (long)((quota + period - 1) / period)));
    return std::max<long>(1, n);
}

extern "C" int tbk_host_threads(void) {
    static int cached = 0;
    if (cached) return cached;
    long n = usable_cpus();
    // One process per GPU (the launcher's LOCAL_WORLD_SIZE ranks on this node; TBK_LOCAL_RANKS says the same by hand):
    // the ranks share the node's CPUs, so each takes its share - 8 ranks in a 16-CPU cgroup start 2 workers each,
    // not 16 each.  (A single process driving several devices divides inside tbk_pipeline_create: tbk_host_threads_per_feeder_.)
    const long ranks = local_ranks();
    if (ranks > 1) n = std::max<long>(1, n / ranks);
    const double forced = env_double("TBK_HOST_THREADS", 0);
    if (forced >= 1) n = (long)forced;
    cached = (int)std::max<long>(1, n);
    return cached;
}

// The packing share of one feeder of a pipeline over n_devices rings: the CPUs are divided among whatever is more -
// the node's ranks or this process's rings - once, not by both (a process that drives several devices from inside a
// launcher's environment would otherwise get cpus / (ranks * devices)).
extern "C" int tbk_host_threads_per_feeder_(int n_devices) {
    const double forced = env_double("TBK_HOST_THREADS", 0);
    if (forced >= 1) return (int)std::max<long>(1, (long)forced / std::max(1, n_devices));
    return (int)std::max<long>(1, usable_cpus() / std::max<long>(local_ranks(), std::max(1, n_devices)));
}

// ---- NUMA placement ("pinned, NUMA-local buffers, one feeder thread per GPU") -------------------------
// A GPU hangs off one socket's PCIe root; a feeder thread that packs and stages batches on the other socket moves every
// byte across the inter-socket link first.  The node of a device is what the kernel says in
// /sys/bus/pci/devices/<bdf>/numa_node (-1 or a missing file: unknown - nothing is bound); its CPUs are
// /sys/devices/system/node/node<N>/cpulist.  TBK_SYSFS_ROOT replaces "/sys" (tests), TBK_NUMA=0 turns binding off.
// Pinned buffers follow the thread: hipHostMalloc places host memory near the current device by default, and the
// buffers are allocated (and first touched) by the bound feeder thread.
static std::string sysfs_root() {
    const char *r = getenv("TBK_SYSFS_ROOT");
    return r && *r ? std::string(r) : std::string("/sys");
}

extern "C" int tbk_numa_node_of_pci_(const char *bdf) {
    if (!bdf || !*bdf) return -1;
    std::string id(bdf);
    for (char &ch : id) ch = (char)tolower((unsigned char)ch);
    FILE *f = fopen((sysfs_root() + "/bus/pci/devices/" + id + "/numa_node").c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node < 0 ? -1 : node;
}

// the CPUs of a node ("0-63,128-191"), at most `cap` of them written; returns how many the node has (0: unknown node)
extern "C" int tbk_numa_node_cpus_(int node, int *cpus, int cap) {
    if (node < 0) return 0;
    FILE *f = fopen((sysfs_root() + "/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
    if (!f) return 0;
    char buf[4096] = {0};
    const size_t got = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[got] = 0;
    int n = 0;
    const char *p = buf;
    while (*p) {
        while (*p == ',' || *p == ' ' || *p == '\n') p++;
        if (!isdigit((unsigned char)*p)) break;
        char *end = nullptr;
        long lo = strtol(p, &end, 10), hi = lo;
        p = end;
        if (*p == '-') { hi = strtol(p + 1, &end, 10); p = end; }
        for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) { if (cpus && n < cap) cpus[n] = (int)c; n++; }
    }
    return n;
}

// Bind the calling thread (and the threads it starts from now on) to the CPUs of `node` that its current affinity mask
// allows.  Returns the number of CPUs it is bound to, 0 when nothing was changed (unknown node, no CPU in common,
// TBK_NUMA=0, or the kernel refused).
extern "C" int tbk_numa_bind_thread_(int node) {
    const char *off = getenv("TBK_NUMA");
    if (off && *off == '0') return 0;
    std::vector<int> cpus(CPU_SETSIZE);
    const int n = tbk_numa_node_cpus_(node, cpus.data(), (int)cpus.size());
    if (n <= 0) return 0;
    cpu_set_t cur, want;
    if (pthread_getaffinity_np(pthread_self(), sizeof cur, &cur) != 0) return 0;
    CPU_ZERO(&want);
    for (int i = 0; i < std::min(n, (int)cpus.size()); i++) if (CPU_ISSET(cpus[(size_t)i], &cur)) CPU_SET(cpus[(size_t)i], &want);
    const int k = CPU_COUNT(&want);
    if (k == 0 || k == CPU_COUNT(&cur)) return k == 0 ? 0 : k;  // (nothing in common; or already inside the node)
    if (pthread_setaffinity_np(pthread_self(), sizeof want, &want) != 0) return 0;
    return k;
}

extern "C" int tbk_device_numa_node(int device, int *node) {
    if (!node) return fail(TBK_ERR_INVALID, "node is NULL");
    *node = -1;
    char bdf[64] = {0};
    int n_visible = 0;
    if (hipGetDeviceCount(&n_visible) != hipSuccess || n_visible <= 0) { (void)hipGetLastError(); return fail(TBK_ERR_NO_DEVICE, "no HIP device visible"); }
    if (device < 0 || device >= n_visible) return fail(TBK_ERR_INVALID, "device %d out of range (0..%d)", device, n_visible - 1);
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return TBK_OK; }  // (unknown: -1)
    *node = tbk_numa_node_of_pci_(bdf);
    return TBK_OK;
}

extern "C" int tbk_numa_bind_to_device(int device, int *node_out, int *cpus_out) {
    int node = -1;
    const int rc = tbk_device_numa_node(device, &node);
    if (node_out) *node_out = node;
    if (cpus_out) *cpus_out = 0;
    if (rc) return rc;
    const int k = tbk_numa_bind_thread_(node);
    if (cpus_out) *cpus_out = k;
    return TBK_OK;
}

// ---- handles ---------------------------------------------------------------------------
// A k-mer list in HBM: its packed keys, one per list line, verbatim (duplicates and all).
// The hashed forms are built from these: the paired hapA|hapB table by the classifier, and
// a standalone table (64-byte lines) on first use of tbk_table_contains / _distinct.
static void table_born();
static void table_gone();
struct tbk_table {
    tbk_table() { table_born(); }
    ~tbk_table() { table_gone(); }
    tbk_table(const tbk_table &) = delete;
    tbk_table &operator=(const tbk_table &) = delete;
    int device = 0;
    int k = 0;
    uint64_t num_lines = 0;  // what the reference calls num_kmers (c/kmers.c:37)
    int origin = 0;          // how the keys got here: 0 caller's keys / general host parser, 1 GPU parser, 2 binary key cache
    uint64_t *d_keys = nullptr;
    // lazily built standalone table
    uint64_t *d_slots = nullptr;
    uint32_t n_buckets = 0;
    uint64_t distinct = 0;
    bool hashed = false;
    TbkTableView view() const { return TbkTableView{d_slots, n_buckets, 8, 0, TbkMz{0, 0, 0, 0}, 0}; }
};

// What a build of the paired table reads of a list: its packed keys on the CURRENT device.  A tbk_table on its own device, or
// the copy of its keys a replica's build has brought to another device (tbk_classifier_create_multi_opts).
struct ListRef {
    const uint64_t *d_keys;
    uint64_t num_lines;
    ListRef(const tbk_table *t) : d_keys(t->d_keys), num_lines(t->num_lines) {}
    ListRef(const uint64_t *keys, uint64_t n) : d_keys(keys), num_lines(n) {}
    const ListRef *operator->() const { return this; }
};

static constexpr int RING = 3;
static constexpr int TIMING_POOL = 4096;  // events kept for kernel timing before they are folded into sums

struct Slot {
    uint8_t *d_bases = nullptr; size_t cap_bases = 0;
    uint64_t *d_offsets = nullptr; size_t cap_reads = 0;  // capacity in reads (offsets has +1)
    int32_t *d_counts = nullptr;
    uint8_t *h_bases = nullptr; size_t hcap_bases = 0;    // pinned staging for unpinned callers
    uint64_t *h_offsets = nullptr; int32_t *h_counts = nullptr; size_t hcap_reads = 0;
    // packed transfer format (tbk_pack.cpp): code words, dense masks, exceptions; pinned staging for
    // batches packed at submit time
    uint32_t *d_codes = nullptr; uint16_t *d_bad = nullptr; size_t cap_chunks = 0;
    bool bad_clean = false;  // d_bad is all zero (it is between batches: a batch's exceptions are taken out again behind its probe)
    uint32_t *d_exc_chunk = nullptr; uint16_t *d_exc_mask = nullptr; size_t cap_exc = 0;
    uint32_t *h_codes = nullptr; size_t hcap_chunks = 0;
    uint32_t *h_exc_chunk = nullptr; uint16_t *h_exc_mask = nullptr; size_t hcap_exc = 0;
    hipEvent_t copied = nullptr, probed = nullptr, done = nullptr, copied2 = nullptr;
    hipEvent_t sliced[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // a host batch's bases arrive (and are probed) in up to 8 slices
    bool busy = false;
    uint64_t ticket = 0, n_reads = 0;
    int32_t *user_counts = nullptr;
    bool counts_staged = false;
};

// The three streams of a device's rings.  Several rings on ONE device (TBK_DEVICES=0,0,0) share them: the runtime maps
// a process's streams onto a handful of hardware queues (4 by default), and nine streams on four queues put one
// ring's copy in front of another ring's kernels (three rings of their own streams: 122 Gbases/s where one ring
// does 178; GPU_MAX_HW_QUEUES=8: 166 - profiles/r03/ab_rings_hw_queues.log).  `enqueue` keeps one batch's sequence of
// launches together on the shared streams (the feeder threads of the rings submit concurrently).
struct DeviceStreams {
    int device = 0;
    hipStream_t compute = nullptr, copy = nullptr, out = nullptr;
    hipStream_t copy2 = nullptr;  // second H2D stream (TBK_H2D_STREAMS=2): a steady-state batch's bases cross PCIe in two halves on two DMA engines
    std::mutex enqueue;
    std::atomic<int> in_flight{0};  // batches submitted and not yet waited for, over all rings of the device
    ~DeviceStreams() {
        if (hipSetDevice(device) != hipSuccess) return;
        if (compute) (void)hipStreamDestroy(compute);
        if (copy) (void)hipStreamDestroy(copy);
        if (copy2) (void)hipStreamDestroy(copy2);
        if (out) (void)hipStreamDestroy(out);
    }
};

struct tbk_classifier {
    int device = 0;
    int k = 0;
    tbk_options opt = resolve_options(nullptr);  // what it was created with (tbk_classifier_create_opts)
    std::shared_ptr<DeviceStreams> streams;
    uint64_t *d_pair = nullptr;  // n_buckets lines of 128 B (front layout: [A0-3 | B0-3 | A4-7 | B4-7])
    // who frees it: the classifiers of one device share one read-only table (several stream rings on one GPU need
    // no second copy of it); every other device holds its own replica
    std::shared_ptr<uint64_t> pair_owner;
    void *pair_base = nullptr;   // what hipMalloc returned (d_pair may be aligned up inside it: TBK_TABLE_ALIGN)
    void own_pair() {
        const int dev = device;
        void *base = pair_base ? pair_base : (void *)d_pair;
        pair_owner = std::shared_ptr<uint64_t>(d_pair, [dev, base](uint64_t *) { if (base && hipSetDevice(dev) == hipSuccess) (void)hipFree(base); });
        pair_base = nullptr;
    }
    // the table's memory: `bytes`, aligned to TBK_TABLE_ALIGN when that is set (the allocation is made that much larger)
    hipError_t alloc_pair(size_t bytes) {
        const size_t align = (size_t)opt.table_align;
        pair_base = nullptr; d_pair = nullptr;
        hipError_t e = hipMalloc(&pair_base, bytes + align);
        if (e != hipSuccess) { pair_base = nullptr; return e; }
        uintptr_t p = (uintptr_t)pair_base;
        if (align > 1) p = (p + align - 1) / align * align;
        d_pair = (uint64_t *)p;
        return hipSuccess;
    }
    void free_pair() {  // before own_pair(): a table that is being rebuilt
        if (pair_base) (void)hipFree(pair_base);
        else if (d_pair) (void)hipFree(d_pair);
        pair_base = nullptr; d_pair = nullptr;
    }
    uint32_t n_buckets = 0;
    uint64_t distinct_a = 0, distinct_b = 0;
    uint64_t shared = 0;         // hapB list lines left out of the table because hapA holds their key
    TbkMz mz{0, 0, 0};           // how a key picks its bucket (minimizer span or plain hash)
    int replica_copies = 0;      // 0: built here first, or shared; 1: a copy of another classifier's finished table (peer copy); 2: built again here from the lists in the first one's geometry
    int layout_builds = 0;       // times the paired table was built (2: the lists clustered under mod-sampling)
    uint64_t past_half = 0;      // keys that found their own half of their home line full
    uint64_t behind_front = 0;   // keys that are not among the first four slots of their list in their home line (entry layouts: entries behind the line's 32-byte front - four narrow entries, two wide ones; short and full keys: keys behind the front's seven / three)
    uint64_t entries_a = 0, entries_b = 0;  // entry layout (TBK_FLAG_ENTRY): slots the lists' keys take (a run of overlapping keys is one entry)
    uint32_t guests = 0;         // TBK_FLAG_GUESTS (k < 32: a full half's surplus sits, tagged, in the other half of its line before it leaves the line) | TBK_FLAG_FRONT (tbk_common.h)
    uint64_t verified_lines = 0; // list lines looked up again in the finished table (tbk_options.verify_build), all answered as the lists say
    double verify_s = 0;
    uint32_t over_mask = 0;      // short keys (TBK_FLAG_SHORT): the overflow table behind the lines has over_mask + 1 slots of 8 bytes
    uint64_t table_bytes() const { return (uint64_t)n_buckets * 2 * TBK_BUCKET_BYTES + ((guests & TBK_FLAG_SHORT) ? ((uint64_t)over_mask + 1) * 8 : 0); }
    TbkPairView pair() const { return TbkPairView{d_pair, n_buckets, mz, guests, over_mask}; }
    hipStream_t compute = nullptr, copy = nullptr, out = nullptr;  // kernels; H2D of the next batch; D2H of finished counts
    Slot ring[RING];
    uint64_t next_ticket = 1;
    int max_blocks = 0;
    // how tbk_stream_submit moves a host batch's bases: 1 = packed on the host first (0.25 B/base over
    // PCIe), 0 = as ASCII (1 B/base).  TBK_PACKED_H2D, tbk_classifier_set_transfer.
    int packed_h2d = 1;
    int pack_threads = 0;             // host threads a submit-time pack may use (0: all; a pipeline's feeders share them)
    uint64_t slice_bases = (uint64_t)384 << 20;  // an empty ring takes a host batch in slices of at least this many bases (TBK_SLICE_BASES; tests shrink it)
    std::vector<uint32_t> exc_chunk;  // scratch of the packer
    std::vector<uint16_t> exc_mask;
    // scratch for the pass -> read index (launches on `compute` are stream-ordered, so one
    // buffer serves them all)
    uint32_t *d_pass_read = nullptr;
    uint64_t cap_passes = 0;
    uint64_t last_passes = 0;  // passes of the most recent probe (tbk_classifier_last_passes)
    // kernel timing: per probe launch three events - before the pass-index kernel, between the multi-read and
    // the single-read probe kernel, after the latter
    bool timing = false;
    std::vector<hipEvent_t> ev;        // per probe: before the index kernel, after it; per slice: before the multi-read kernel, between the two, after the single-read kernel
    std::vector<uint32_t> ev_slices;   // slices of every timed probe, in order (a probe uses 2 + 3 * slices events)
    size_t ev_used = 0;
    uint64_t timed_launches = 0;
    double timed_ms = 0.0;         // whole probe: index + both kernels
    double timed_single_ms = 0.0;  // the single-read probe kernel alone
    int fold_timing();
    // One read per call (tbk_count_kmers_in_read: the reference's own loop, classify_by_kmers.py:99-102, through the literal drop-in).
    // The read is copied into `small_h` (pinned, mapped into the device's address space) and the single-read kernel reads it
    // THERE, over PCIe - [offsets: 2 x u64 | pad to 64 B | bases]; the pass index of a one-read batch is all zeros and lives in
    // `small_scratch`, written once; the two counters in `small_d_counts` are never cleared - they run on, and the host takes the
    // difference to the previous call's (mod 2^32).  A call is one memcpy on the host, ONE kernel launch, one 8-byte copy back,
    // one stream synchronisation: 105 us -> what INTEGRATION.md C records.
    uint8_t *small_h = nullptr;
    uint32_t *small_h_counts = nullptr, small_last[2] = {0, 0};
    int32_t *small_d_counts = nullptr;
    uint32_t *small_scratch = nullptr;
    uint64_t small_cap = 0, small_pass_cap = 0;
    void small_free() {
        if (small_h) (void)hipHostFree(small_h);
        if (small_h_counts) (void)hipHostFree(small_h_counts);
        if (small_d_counts) (void)hipFree(small_d_counts);
        if (small_scratch) (void)hipFree(small_scratch);
        small_h = nullptr; small_h_counts = nullptr; small_d_counts = nullptr; small_scratch = nullptr; small_cap = small_pass_cap = 0;
    }
};

// ---- library ---------------------------------------------------------------------------
extern "C" int tbk_abi_version(void) { return TBK_ABI_VERSION; }
extern "C" const char *tbk_last_error(void) { return g_err.c_str(); }

extern "C" int tbk_device_count(int *count) {
    if (!count) return fail(TBK_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(TBK_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return TBK_OK;
}

// "<pci bus id> <uuid>": what tells two devices apart in a multi-rank run's record
extern "C" int tbk_device_identity(int device, char *buf, size_t buflen) {
    if (!buf || buflen < 2) return fail(TBK_ERR_INVALID, "buf is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    char bus[64] = "?";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); strcpy(bus, "?"); }
    hipUUID uuid;
    char hex[40] = "?";
    if (hipDeviceGetUuid(&uuid, device) == hipSuccess) {
        for (int i = 0; i < 16; i++) snprintf(hex + 2 * i, 3, "%02x", (unsigned)(unsigned char)uuid.bytes[i]);
    } else {
        (void)hipGetLastError();
    }
    snprintf(buf, buflen, "%s %s", bus, hex);
    return TBK_OK;
}

extern "C" int tbk_device_name(int device, char *buf, size_t buflen) {
    if (!buf || !buflen) return fail(TBK_ERR_INVALID, "buf is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    snprintf(buf, buflen, "%s %s, %d CUs, %.0f GB", p.gcnArchName, p.name, p.multiProcessorCount,
             (double)p.totalGlobalMem / 1e9);
    return TBK_OK;
}

// ---- unit-level host utilities -----------------------------------------------------------
static inline unsigned code_of(unsigned char c) {  // c/kmers.c:54-68: anything else -> 0
    return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
}

extern "C" uint64_t tbk_kmer_to_int(const char *kmer, unsigned char k) {
    uint64_t v = 0;
    for (unsigned i = 0; i < k && i < 32; i++) v |= (uint64_t)code_of((unsigned char)kmer[i]) << (2 * i);
    return v;
}

extern "C" void tbk_reverse_complement(const char *in, char *out, unsigned char k) {
    for (unsigned i = 0; i < k; i++) {
        char c = 0;
        switch (in[i]) { case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break; }
        if (c) out[k - 1 - i] = c;  // c/kmers.c:78-91: other bytes leave the slot untouched
    }
}

// ---- tables ----------------------------------------------------------------------------
static uint32_t buckets_for(uint64_t n_keys, double default_load, size_t line_bytes, double forced_load, uint64_t budget_bytes) {
    // Target load (keys per slot).  Plain hashing: 2 keys per 8-slot half (load 0.25).  Minimizer bucketing puts
    // keys that share a sampled m-mer into one bucket, and real lists cluster further, so it gets 0.64 keys per
    // half: load 0.08, 100 B of HBM per key, 2 x 3e8 keys = 60 GB.  Measured at that scale, resident, same box
    // (profiles/r03/ab_policy.log): uniform lists, front layout: load 0.08 184 Gbases/s, 0.12 (67 B per key) 169,
    // 0.16 (50 B) 152; haplotype-shaped lists in whole lines at 0.08: 138 (front layout at 0.04, 120 GB: 145).
    // The table is capped at the memory budget (tbk_options.memory_budget_bytes; 0: 60 % of the device's memory); bigger
    // lists get a proportionally higher load.  forced_load > 0 (tbk_options.table_load) overrides.
    double load = forced_load;
    const bool forced = load > 0;
    if (!forced) load = default_load;
    if (load < 0.02) load = 0.02;
    if (load > 0.9) load = 0.9;
    double want = (double)n_keys / (TBK_SLOTS_PER_BUCKET * load);
    if (!forced) {
        // the cap is a share of the device's TOTAL memory, not of what happens to be free: the same lists
        // give the same table whatever else lives on the device (and a table that does not fit fails loudly)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double fit = (budget_bytes ? (double)budget_bytes : 0.6 * (double)total_b) / (double)line_bytes;
            const double floor_ = (double)n_keys / (TBK_SLOTS_PER_BUCKET * 0.5);  // never denser than load 0.5 on our own account
            if (want > fit) want = fit > floor_ ? fit : floor_;
        } else {
            (void)hipGetLastError();
        }
    }
    uint64_t nb = (uint64_t)want + 1;
    if (nb < 16) nb = 16;
    if (nb > 0x7FFFFFF0ull) nb = 0x7FFFFFF0ull;  // bit 31 of a bucket index is a flag inside the probe kernel
    return (uint32_t)nb;
}

// Insert n keys into one list's slots of a table (standalone: stride 8, half 0; paired:
// stride 16, half 0 / 8).  The slots must already be filled with TBK_EMPTY.
static int insert_keys(uint64_t *d_slots, uint32_t n_buckets, uint32_t stride, uint32_t half, TbkMz mz, uint32_t *d_overflowed,
                       const uint64_t *d_keys, uint64_t n, uint64_t *distinct_out,
                       TbkTableView skip = TbkTableView{nullptr, 0, 0, 0, TbkMz{0, 0, 0, 0}, 0}, uint64_t *skipped_out = nullptr,
                       uint32_t *d_left_line = nullptr, uint32_t guests = 0, uint64_t *past_out = nullptr, uint64_t *back_out = nullptr) {
    // counters: [0] distinct keys stored, [1] keys dropped because `skip` holds them, [2] keys that found
    // their own half of their home line full, [3] keys stored behind the first four slots of it
    unsigned long long *d_cnt = nullptr, cnt[4] = {0, 0, 0, 0};
    int *d_failed = nullptr;
    hipError_t e = hipMalloc((void **)&d_cnt, sizeof cnt);
    if (e == hipSuccess) e = hipMalloc((void **)&d_failed, sizeof(int));
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, sizeof cnt);
    if (e == hipSuccess) e = hipMemset(d_failed, 0, sizeof(int));
    if (e == hipSuccess) e = tbk_launch_insert(d_slots, n_buckets, stride, half, mz, d_keys, n, d_overflowed, d_left_line, guests, skip, d_cnt, d_cnt + 1, d_failed, nullptr);
    int failed = 0;
    if (e == hipSuccess) e = hipMemcpy(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&failed, d_failed, sizeof failed, hipMemcpyDeviceToHost);
    if (d_cnt) (void)hipFree(d_cnt);
    if (d_failed) (void)hipFree(d_failed);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "table insert: %s", hipGetErrorString(e));
    if (failed) return fail(TBK_ERR_HIP, "table insert overflowed (table full)");
    *distinct_out = cnt[0];
    if (skipped_out) *skipped_out = cnt[1];
    if (past_out) *past_out = cnt[2];
    if (back_out) *back_out = cnt[3] + cnt[2];  // not in the front of their home line: behind it in the same half, or past it
    return TBK_OK;
}

// One bit per half: "a key went past this half" (set by the inserts, turned into the order of
// the half's last two slots by finish_table, then dropped).
static int overflow_bitmap(uint64_t n_halves, uint32_t **out) {
    const size_t bytes = ((size_t)n_halves + 31) / 32 * 4 + 4;
    hipError_t e = hipMalloc((void **)out, bytes);
    if (e == hipSuccess) e = hipMemset(*out, 0, bytes);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "overflow bitmap (%zu bytes): %s", bytes, hipGetErrorString(e));
    return TBK_OK;
}

static int order_table(uint64_t *d_slots, uint64_t n_halves, const uint32_t *d_overflowed, const uint32_t *d_left_line = nullptr,
                       uint32_t flags = 0, uint32_t stride = 8) {
    hipError_t e = tbk_launch_order(d_slots, n_halves, d_overflowed, d_left_line, flags, stride, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(TBK_ERR_HIP, "table order pass: %s", hipGetErrorString(e));
    return TBK_OK;
}

// Build the standalone hashed form of a list on first use.
static int table_hash(tbk_table *t) {
    if (t->hashed) return TBK_OK;
    int rc = use_device(t->device);
    if (rc) return rc;
    t->n_buckets = buckets_for(t->num_lines, 0.25, TBK_BUCKET_BYTES, env_double("TBK_TABLE_LOAD", 0), 0);  // (a list's standalone table - tbk_table_contains, tests - not a classifier's)
    const size_t bytes = (size_t)t->n_buckets * TBK_BUCKET_BYTES;
    HIP_TRY(hipMalloc((void **)&t->d_slots, bytes));
    hipError_t e = hipMemset(t->d_slots, 0xFF, bytes);
    uint32_t *d_over = nullptr;
    if (e != hipSuccess) rc = fail(TBK_ERR_HIP, "hipMemset: %s", hipGetErrorString(e));
    if (!rc) rc = overflow_bitmap(t->n_buckets, &d_over);
    if (!rc) {
        rc = insert_keys(t->d_slots, t->n_buckets, 8, 0, TbkMz{0, 0, 0, 0}, d_over, t->d_keys, t->num_lines, &t->distinct);
        if (!rc) rc = order_table(t->d_slots, t->n_buckets, d_over);
        (void)hipFree(d_over);
    }
    if (rc) { (void)hipFree(t->d_slots); t->d_slots = nullptr; return rc; }
    t->hashed = true;
    return TBK_OK;
}

static int table_new(const uint64_t *src, bool src_on_device, uint64_t n, int k, uint64_t num_lines, int device,
                     tbk_table **out) {
    tbk_table *t = new tbk_table();
    t->device = device; t->k = k; t->num_lines = n;
    (void)num_lines;
    hipError_t e = hipMalloc((void **)&t->d_keys, (n ? n : 1) * sizeof(uint64_t));
    if (e == hipSuccess && n)
        e = hipMemcpy(t->d_keys, src, n * sizeof(uint64_t), src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (t->d_keys) (void)hipFree(t->d_keys);
        delete t;
        return fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "table keys: %s", hipGetErrorString(e));
    }
    *out = t;
    return TBK_OK;
}

extern "C" int tbk_table_create_from_device_keys(const void *d_keys, uint64_t n, int k, int device, tbk_table **out) {
    if (!out) return fail(TBK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (k < 1 || k > 32) return fail(TBK_ERR_INVALID, "k = %d outside 1..32", k);
    if (!n) return fail(TBK_ERR_FORMAT, "empty k-mer list");
    if (!d_keys) return fail(TBK_ERR_INVALID, "keys is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    return table_new((const uint64_t *)d_keys, true, n, k, n, device, out);
}

extern "C" int tbk_table_create_from_keys(const uint64_t *keys, uint64_t n, int k, int device, tbk_table **out) {
    if (!out) return fail(TBK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (k < 1 || k > 32) return fail(TBK_ERR_INVALID, "k = %d outside 1..32", k);
    if (!n) return fail(TBK_ERR_FORMAT, "empty k-mer list");
    if (!keys) return fail(TBK_ERR_INVALID, "keys is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    return table_new(keys, false, n, k, n, device, out);
}

// Text list -> packed keys with the reference's getline() rules (c/kmers.c:124-146,204-221):
// k = bytes of the first line as getline returns them (newline included) minus one; each
// getline success is one k-mer; a line contributes its first k bytes (a byte outside ACGT,
// the newline included, packs as 0).  A line with fewer than k bytes packs what the reference's
// getline buffer holds there: the line, a NUL, then the tail earlier lines left (parse_list).
//
// Lists are gigabytes (22 B per 21-mer line, 6.6 GB per 300 M-line list; the reference parses
// them at 2.3-2.5 M lines/s).  The common case — every line exactly k bytes + '\n' — is
// packed by all host threads in parallel straight from the mapping; anything irregular falls
// back to the sequential general parser, which implements the rules above line by line.
static const uint8_t *g_code_lut() {
    static uint8_t lut[256];
    static bool ready = false;
    if (!ready) {
        memset(lut, 0, sizeof lut);
        lut[(unsigned)'C'] = 1; lut[(unsigned)'G'] = 2; lut[(unsigned)'T'] = 3;
        lut[(unsigned)'\n'] = 0x80;  // marks a newline inside the k bytes: irregular file
        ready = true;
    }
    return lut;
}

static bool parse_regular(const char *data, size_t size, long k, std::vector<uint64_t> &keys) {
    const size_t stride = (size_t)k + 1;
    const bool tail_nl = size % stride == 0;
    if (!tail_nl && (size + 1) % stride != 0) return false;
    const size_t n = (size + 1) / stride;
    if (n == 0) return false;
    keys.resize(n);
    const uint8_t *lut = g_code_lut();
    unsigned hw = (unsigned)tbk_host_threads();
    size_t nt = std::min<size_t>(std::max(1u, hw), 64);
    nt = std::min(nt, std::max<size_t>(1, n / 65536));
    std::atomic<bool> ok{true};
    auto work = [&](size_t lo, size_t hi) {
        const uint8_t *p = (const uint8_t *)data + lo * stride;
        for (size_t i = lo; i < hi && ok.load(std::memory_order_relaxed); i++, p += stride) {
            uint64_t v = 0;
            unsigned flag = 0;
            for (long j = 0; j < k; j++) { const unsigned c = lut[p[j]]; flag |= c; v |= (uint64_t)(c & 3u) << (2 * j); }
            const bool last = i + 1 == n;
            if ((flag & 0x80u) || (!(last && !tail_nl) && p[k] != '\n')) { ok.store(false); return; }
            keys[i] = v;
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nt; t++) pool.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    work(0, n / nt);
    for (auto &th : pool) th.join();
    return ok.load();
}

static int parse_list(const char *path, std::vector<uint64_t> &keys, int &k_out) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(TBK_ERR_IO, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return fail(TBK_ERR_IO, "%s is not a regular file", path); }
    const size_t size = (size_t)st.st_size;
    if (size == 0) { close(fd); return fail(TBK_ERR_FORMAT, "%s is empty: no k-mers", path); }
    const char *data = (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (data == MAP_FAILED) return fail(TBK_ERR_IO, "mmap %s: %s", path, strerror(errno));
    const char *end = data + size;
    const char *nl = (const char *)memchr(data, '\n', size);
    const size_t first_len = nl ? (size_t)(nl - data) + 1 : size;  // as getline counts it
    const long k = (long)first_len - 1;
    if (k < 1 || k > 32) {
        munmap((void *)data, size);
        return fail(TBK_ERR_FORMAT, "%s: first line gives k = %ld (supported: 1..32)", path, k);
    }
    k_out = (int)k;
    if (nl && parse_regular(data, size, k, keys)) {
        munmap((void *)data, size);
        return TBK_OK;
    }
    (void)madvise((void *)data, size, MADV_SEQUENTIAL);
    keys.clear();
    keys.reserve(size / (size_t)(k + 1) + 1);
    // The reference packs the first k bytes of getline()'s buffer whatever the line's length
    // (c/kmers.c:113,204-206).  A line with fewer than k bytes therefore packs its own bytes, the
    // terminating NUL getline wrote, and behind it whatever EARLIER lines left in the buffer:
    // getline reuses (and only ever grows, by realloc) one buffer, and the first line has k + 1
    // bytes, so those bytes are defined.  `sim` is that buffer's first k bytes.  (A blank last line,
    // which many tools leave, is such a line: it counts in num_kmers and stores a key made of the
    // previous line's tail - almost never canonical, hence dead, exactly as in the reference.)
    std::vector<uint8_t> sim((size_t)k, 0);
    const char *p = data;
    while (p < end) {
        const char *e = (const char *)memchr(p, '\n', (size_t)(end - p));
        const size_t got = e ? (size_t)(e - p) + 1 : (size_t)(end - p);
        const size_t take = std::min(got, (size_t)k);
        memcpy(sim.data(), p, take);
        if (got < (size_t)k) sim[got] = 0;  // getline's terminator
        uint64_t v = 0;
        for (long i = 0; i < k; i++) v |= (uint64_t)code_of(sim[(size_t)i]) << (2 * i);
        keys.push_back(v);
        p += got;
    }
    munmap((void *)data, size);
    return TBK_OK;
}

// Host-only: the packed keys of a list file (what tbk_table_create_from_file places in HBM).
extern "C" int tbk_list_parse_file(const char *path, uint64_t **keys_out, uint64_t *n_out, int *k_out) {
    if (!path || !keys_out || !n_out || !k_out) return fail(TBK_ERR_INVALID, "NULL argument");
    *keys_out = nullptr; *n_out = 0; *k_out = 0;
    std::vector<uint64_t> keys;
    int k = 0;
    int rc = parse_list(path, keys, k);
    if (rc) return rc;
    uint64_t *mem = (uint64_t *)malloc((keys.size() ? keys.size() : 1) * sizeof(uint64_t));
    if (!mem) return fail(TBK_ERR_NOMEM, "out of memory");
    memcpy(mem, keys.data(), keys.size() * sizeof(uint64_t));
    *keys_out = mem; *n_out = keys.size(); *k_out = k;
    return TBK_OK;
}
extern "C" void tbk_list_free(uint64_t *keys) { free(keys); }

// ---- lists from files: the GPU parses regular lists; a binary key cache skips the text ------------------
extern "C" hipError_t tbk_launch_parse_lines(const uint8_t *, uint64_t, int, int, uint64_t *, int *, hipStream_t);

// pread of [off, off + n) of a file into `dst` over several host threads (the copy out of the page cache is
// what reading a cached file costs; one thread moves ~5-10 GB/s, PCIe wants 50)
static bool par_pread(int fd, uint8_t *dst, size_t n, size_t off, uint64_t *sum64 = nullptr) {
    const size_t piece = (size_t)8 << 20;
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)tbk_host_threads(), n / piece));
    std::atomic<bool> ok{true};
    std::vector<uint64_t> sums((size_t)nt, 0);
    auto work = [&](int t) {
        size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
        if (sum64) { lo &= ~(size_t)7; hi = t + 1 == nt ? n : (hi & ~(size_t)7); }  // whole 8-byte words per thread
        size_t at = lo;
        while (at < hi) {
            const ssize_t got = ::pread(fd, dst + at, hi - at, (off_t)(off + at));
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) { ok.store(false); return; }
            at += (size_t)got;
        }
        if (sum64) {
            uint64_t acc = 0;
            const uint64_t *w = (const uint64_t *)(dst + lo);
            for (size_t i = 0; i < (hi - lo) / 8; i++) acc += w[i];
            sums[(size_t)t] = acc;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
    work(0);
    for (std::thread &th : pool) th.join();
    if (sum64) for (uint64_t v : sums) *sum64 += v;
    return ok.load();
}

// Two pinned staging buffers and two events: piece p is read from the file into buffer p & 1 while piece
// p - 1 crosses PCIe.  `consume(slot, staged bytes, piece offset, piece bytes)` enqueues the piece's copy
// (and kernel) on `stream`.
// The staging buffers outlive a list: the command line loads two lists one after the other, and pinning 2 x 128 MB
// again for the second costs a third of its 0.18 s.  They are kept until a classifier is made of the lists
// (tbk_classifier_create lets them go) or another size is asked for.
extern "C" void *tbk_pin_alloc_(size_t bytes);
extern "C" void tbk_pin_free_(void *p);
struct ListStaging {
    std::mutex mu;
    uint8_t *h[2] = {nullptr, nullptr};
    size_t bytes = 0;
    void release_locked() {
        for (uint8_t *&b : h) { if (b) tbk_pin_free_(b); b = nullptr; }
        bytes = 0;
    }
};
static ListStaging g_list_staging;
static std::atomic<long> g_live_tables{0};  // tbk_table handles alive: the staging buffers go with the last one at the latest
static void release_list_staging() {
    std::lock_guard<std::mutex> lk(g_list_staging.mu);
    g_list_staging.release_locked();
}
static void table_born() { g_live_tables.fetch_add(1); }
static void table_gone() { if (g_live_tables.fetch_sub(1) <= 1) release_list_staging(); }

template <class F>
static int staged_file_upload(int fd, size_t file_off, size_t bytes, size_t piece_bytes, hipStream_t stream, uint64_t *sum64, F consume) {
    std::lock_guard<std::mutex> staging_lock(g_list_staging.mu);  // (lists are loaded one at a time)
    uint8_t *h[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = TBK_OK;
    hipError_t e = hipSuccess;
    if (g_list_staging.bytes != piece_bytes) g_list_staging.release_locked();
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        if (!g_list_staging.h[i] && !(g_list_staging.h[i] = (uint8_t *)tbk_pin_alloc_(piece_bytes))) e = hipErrorOutOfMemory;   // (huge pages, registered: below)
        h[i] = g_list_staging.h[i];
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) g_list_staging.bytes = piece_bytes; else g_list_staging.release_locked();
    if (e != hipSuccess) rc = fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "list staging: %s", hipGetErrorString(e));
    size_t p = 0;
    for (size_t off = 0; off < bytes && !rc; off += piece_bytes, p++) {
        const int slot = (int)(p & 1);
        const size_t n = std::min(piece_bytes, bytes - off);
        if (p >= 2) {
            e = hipEventSynchronize(ev[slot]);  // the copy that last read this buffer is done
            if (e != hipSuccess) { rc = fail(TBK_ERR_HIP, "list staging: %s", hipGetErrorString(e)); break; }
        }
        if (!par_pread(fd, h[slot], n, file_off + off, sum64)) { rc = fail(TBK_ERR_IO, "reading the list: %s", strerror(errno)); break; }
        rc = consume(h[slot], off, n, slot);
        if (!rc) {
            e = hipEventRecord(ev[slot], stream);
            if (e != hipSuccess) rc = fail(TBK_ERR_HIP, "list staging: %s", hipGetErrorString(e));
        }
    }
    if (hipStreamSynchronize(stream) != hipSuccess && !rc) rc = fail(TBK_ERR_HIP, "list upload failed");
    for (int i = 0; i < 2; i++) if (ev[i]) (void)hipEventDestroy(ev[i]);
    return rc;
}

// The binary key cache of a list, `<list>.tbk`: what parsing the text produces, kept so that the next run
// skips the text.  Valid only for the very file it was made from: size, modification time AND a fingerprint of the
// text itself - lists of one k and line count are equally long to the byte, and cp -p / rsync -t / tar keep the
// modification time, so size + mtime alone would hand a replaced list its predecessor's keys.
struct ListCacheHeader {
    char magic[8];         // "TBKLIST2"
    uint32_t k, reserved;
    uint64_t n_lines;      // = num_kmers of the reference (c/kmers.c:124-146), duplicates and all
    uint64_t src_size;
    int64_t src_mtime_ns;
    uint64_t key_sum;      // wrapping sum of the keys
    uint64_t src_fingerprint;  // source_fingerprint() of the text
};

// A hash of the list text's first and last MiB and of fifteen 64 KiB blocks spread evenly between them (3 MB read from
// the page cache: about a millisecond).  Two different lists agree on all of those only by being the same list there.
static bool source_fingerprint(const char *path, uint64_t size, uint64_t *out) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    std::vector<uint8_t> buf((size_t)1 << 20);
    uint64_t h = 0x9E3779B97F4A7C15ull ^ size;
    bool ok = true;
    auto fold = [&](uint64_t off, size_t n) {
        if (off >= size) return;
        n = (size_t)std::min<uint64_t>(n, size - off);
        size_t at = 0;
        while (at < n) {
            const ssize_t got = ::pread(fd, buf.data() + at, n - at, (off_t)(off + at));
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) { ok = false; return; }
            at += (size_t)got;
        }
        h = (h ^ off) * 0xD1342543DE82EF95ull;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, buf.data() + i, 8); h = (h ^ w) * 0xAF251AF3B0F025B5ull; h ^= h >> 29; }
        for (; i < n; i++) h = (h ^ buf[i]) * 0x100000001B3ull;
    };
    fold(0, (size_t)1 << 20);
    for (int i = 1; i <= 15 && ok; i++) fold(size / 16 * (uint64_t)i, (size_t)64 << 10);
    if (ok && size > ((uint64_t)1 << 20)) fold(size - ((uint64_t)1 << 20), (size_t)1 << 20);
    ::close(fd);
    *out = h;
    return ok;
}

static std::string cache_path_of(const char *path) { return std::string(path) + ".tbk"; }

static int64_t mtime_ns_of(const struct stat &st) { return (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec; }

// 1 = table made from the cache, 0 = no usable cache (never an error: a stale or damaged cache is ignored)
static int table_from_cache(const char *path, const struct stat &src, int device, tbk_table **out) {
    const char *env = getenv("TBK_LIST_CACHE");
    if (env && *env == '0') return 0;
    const std::string cp = cache_path_of(path);
    const int fd = ::open(cp.c_str(), O_RDONLY);
    if (fd < 0) return 0;
    ListCacheHeader hd;
    struct stat cst;
    bool good = ::pread(fd, &hd, sizeof hd, 0) == (ssize_t)sizeof hd && memcmp(hd.magic, "TBKLIST2", 8) == 0 && fstat(fd, &cst) == 0 &&
                hd.k >= 1 && hd.k <= 32 && hd.n_lines > 0 && (uint64_t)cst.st_size == sizeof hd + hd.n_lines * 8 &&
                hd.src_size == (uint64_t)src.st_size && hd.src_mtime_ns == mtime_ns_of(src);
    uint64_t fp = 0;
    good = good && source_fingerprint(path, (uint64_t)src.st_size, &fp) && fp == hd.src_fingerprint;
    if (!good || use_device(device) != TBK_OK) { ::close(fd); return 0; }
    tbk_table *t = new tbk_table();
    t->device = device; t->k = (int)hd.k; t->num_lines = hd.n_lines; t->origin = 2;
    uint64_t sum = 0;
    int rc = TBK_OK;
    if (hipMalloc((void **)&t->d_keys, hd.n_lines * 8) != hipSuccess) { (void)hipGetLastError(); rc = TBK_ERR_NOMEM; }
    if (!rc) {
        uint64_t *d_keys = t->d_keys;
        rc = staged_file_upload(fd, sizeof hd, (size_t)hd.n_lines * 8, (size_t)128 << 20, nullptr, &sum, [&](uint8_t *staged, size_t off, size_t n, int) -> int {
            HIP_TRY(hipMemcpyAsync((uint8_t *)d_keys + off, staged, n, hipMemcpyHostToDevice, nullptr));
            return TBK_OK;
        });
    }
    ::close(fd);
    if (rc || sum != hd.key_sum) {  // damaged: parse the text instead
        if (t->d_keys) (void)hipFree(t->d_keys);
        delete t;
        return 0;
    }
    *out = t;
    return 1;
}

static void write_list_cache(const char *path, const struct stat &src, const tbk_table *t) {
    const char *env = getenv("TBK_LIST_CACHE");
    if (!env || *env != '1') return;  // opt-in: a cache is as big as a third of its list
    std::vector<uint64_t> keys((size_t)t->num_lines);
    if (hipMemcpy(keys.data(), t->d_keys, keys.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
    ListCacheHeader hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "TBKLIST2", 8);
    hd.k = (uint32_t)t->k; hd.n_lines = t->num_lines; hd.src_size = (uint64_t)src.st_size; hd.src_mtime_ns = mtime_ns_of(src);
    if (!source_fingerprint(path, (uint64_t)src.st_size, &hd.src_fingerprint)) return;
    for (uint64_t v : keys) hd.key_sum += v;
    const std::string cp = cache_path_of(path), tmp = cp + ".tmp." + std::to_string((long)getpid());
    const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return;  // a directory we may not write to: no cache, no complaint
    bool ok = ::write(fd, &hd, sizeof hd) == (ssize_t)sizeof hd;
    const char *p = (const char *)keys.data();
    size_t left = keys.size() * 8;
    while (ok && left) {
        const ssize_t w = ::write(fd, p, std::min(left, (size_t)1 << 30));
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) { ok = false; break; }
        p += w; left -= (size_t)w;
    }
    ok = ::close(fd) == 0 && ok;
    if (!ok || ::rename(tmp.c_str(), cp.c_str()) != 0) (void)::unlink(tmp.c_str());
}

// 1 = table made by the GPU parser, 0 = the file is not a regular list (the general parser's), < 0 = error
static int table_from_regular_text(const char *path, const struct stat &st, int device, tbk_table **out) {
    const char *env = getenv("TBK_LIST_GPU_PARSE");
    if (env && *env == '0') return 0;
    const size_t size = (size_t)st.st_size;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return 0;
    char head[40];
    const ssize_t got = ::pread(fd, head, sizeof head, 0);
    const char *nl = got > 0 ? (const char *)memchr(head, '\n', (size_t)got) : nullptr;
    const long k = nl ? (long)(nl - head) : -1;
    const size_t stride = (size_t)k + 1;
    const bool closed_end = k >= 1 && size % stride == 0, open_end = k >= 1 && !closed_end && (size + 1) % stride == 0;
    if (k < 1 || k > 32 || !(closed_end || open_end) || use_device(device) != TBK_OK) { ::close(fd); return 0; }
    const uint64_t n = (size + 1) / stride;
    tbk_table *t = new tbk_table();
    t->device = device; t->k = (int)k; t->num_lines = n; t->origin = 1;
    const size_t piece_lines = std::max<size_t>(1, ((size_t)128 << 20) / stride), piece_bytes = piece_lines * stride;  // (32 MB pieces: the second list 0.087 s instead of 0.052)
    uint8_t *d_text[2] = {nullptr, nullptr};
    int *d_irregular = nullptr, irregular = 0;
    hipError_t e = hipMalloc((void **)&t->d_keys, n * 8);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void **)&d_text[i], piece_bytes + 16);
    if (e == hipSuccess) e = hipMalloc((void **)&d_irregular, sizeof(int));
    if (e == hipSuccess) e = hipMemset(d_irregular, 0, sizeof(int));
    int rc = e == hipSuccess ? TBK_OK : fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "list buffers: %s", hipGetErrorString(e));
    if (!rc) {
        uint64_t *d_keys = t->d_keys;
        rc = staged_file_upload(fd, 0, size, piece_bytes, nullptr, nullptr, [&](uint8_t *staged, size_t off, size_t nb, int slot) -> int {
            // the copy into d_text[slot] waits for the kernel that last read it: same stream, in order
            HIP_TRY(hipMemcpyAsync(d_text[slot], staged, nb, hipMemcpyHostToDevice, nullptr));
            const uint64_t first = off / stride, lines = (nb + 1) / stride;
            HIP_TRY(tbk_launch_parse_lines(d_text[slot], lines, (int)k, open_end && off + nb == size, d_keys + first, d_irregular, nullptr));
            return TBK_OK;
        });
    }
    if (!rc && hipMemcpy(&irregular, d_irregular, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(TBK_ERR_HIP, "list parse: reading the flag failed");
    ::close(fd);
    for (int i = 0; i < 2; i++) if (d_text[i]) (void)hipFree(d_text[i]);
    if (d_irregular) (void)hipFree(d_irregular);
    if (rc || irregular) {
        if (t->d_keys) (void)hipFree(t->d_keys);
        delete t;
        return rc ? rc : 0;
    }
    *out = t;
    return 1;
}

// Replaces create_kmer_hash_set's file handling (c/kmers.c:185-229).  In order: the binary key cache of this
// very file (`<list>.tbk`, used when it matches the list's size and modification time; written after a parse
// when TBK_LIST_CACHE=1); the GPU parser for lists of the regular shape (every line k bytes + newline); the
// general host parser, which implements the reference's getline rules for everything else.  All three give
// the same keys (tests/test_gpu_lists.py).
extern "C" int tbk_table_create_from_file(const char *path, int device, tbk_table **out) {
    if (!out) return fail(TBK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!path) return fail(TBK_ERR_INVALID, "path is NULL");
    struct stat st;
    if (::stat(path, &st) != 0) return fail(TBK_ERR_IO, "cannot open %s: %s", path, strerror(errno));
    if (S_ISREG(st.st_mode) && st.st_size > 0) {
        if (table_from_cache(path, st, device, out) == 1) return TBK_OK;
        const int got = table_from_regular_text(path, st, device, out);
        if (got < 0) return got;
        if (got == 1) { write_list_cache(path, st, *out); return TBK_OK; }
    }
    std::vector<uint64_t> keys;
    int k = 0;
    int rc = parse_list(path, keys, k);
    if (rc) return rc;
    rc = tbk_table_create_from_keys(keys.data(), keys.size(), k, device, out);
    if (!rc && S_ISREG(st.st_mode)) write_list_cache(path, st, *out);
    return rc;
}

// the packed keys of a table, copied to the host (tests; tools)
extern "C" int tbk_table_keys(const tbk_table *t, uint64_t *keys, uint64_t capacity) {
    if (!t || (!keys && capacity)) return fail(TBK_ERR_INVALID, "NULL argument");
    if (capacity < t->num_lines) return fail(TBK_ERR_INVALID, "capacity %llu below the table's %llu keys", (unsigned long long)capacity, (unsigned long long)t->num_lines);
    int rc = use_device(t->device);
    if (rc) return rc;
    if (t->num_lines) HIP_TRY(hipMemcpy(keys, t->d_keys, t->num_lines * 8, hipMemcpyDeviceToHost));
    return TBK_OK;
}

static void drop_cached_classifier(const tbk_table *t);

extern "C" void tbk_table_destroy(tbk_table *t) {
    if (!t) return;
    // (a process that only loads lists - tbk_table_contains, tbk_table_keys, tools - never makes a classifier: the pinned
    // staging buffers of the list loader, 2 x 128 MB, go with its last table)
    // (done by ~tbk_table: table_gone)
    drop_cached_classifier(t);
    if (hipSetDevice(t->device) == hipSuccess) {
        if (t->d_keys) (void)hipFree(t->d_keys);
        if (t->d_slots) (void)hipFree(t->d_slots);
    }
    delete t;
}

extern "C" uint64_t tbk_table_num_kmers(const tbk_table *t) { return t ? t->num_lines : 0; }
extern "C" int tbk_table_k(const tbk_table *t) { return t ? t->k : 0; }
extern "C" int tbk_table_device(const tbk_table *t) { return t ? t->device : -1; }
extern "C" int tbk_table_origin(const tbk_table *t) { return t ? t->origin : -1; }
extern "C" const void *tbk_table_device_keys(const tbk_table *t) { return t ? t->d_keys : nullptr; }
extern "C" uint64_t tbk_table_bytes(const tbk_table *t) {
    return t ? t->num_lines * sizeof(uint64_t) + (t->hashed ? (uint64_t)t->n_buckets * TBK_BUCKET_BYTES : 0) : 0;
}
extern "C" int tbk_table_distinct(tbk_table *t, uint64_t *distinct) {
    if (!t || !distinct) return fail(TBK_ERR_INVALID, "NULL argument");
    int rc = table_hash(t);
    if (rc) return rc;
    *distinct = t->distinct;
    return TBK_OK;
}

extern "C" int tbk_table_contains(tbk_table *t, const uint64_t *keys, uint64_t n, uint8_t *out) {
    if (!t || (n && (!keys || !out))) return fail(TBK_ERR_INVALID, "NULL argument");
    int rc = use_device(t->device);
    if (rc) return rc;
    rc = table_hash(t);
    if (rc) return rc;
    if (!n) return TBK_OK;
    uint64_t *d_keys = nullptr;
    uint8_t *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_keys, n * sizeof(uint64_t)));
    hipError_t e = hipMalloc((void **)&d_out, n);
    if (e == hipSuccess) e = hipMemcpy(d_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = tbk_launch_contains(t->view(), d_keys, n, d_out, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost);
    (void)hipFree(d_keys);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(TBK_ERR_HIP, "tbk_table_contains: %s", hipGetErrorString(e));
    return TBK_OK;
}

// the same with keys and answers in device memory on the list's device (the full-membership sweep: tbk_verify.cpp)
extern "C" int tbk_table_contains_device(tbk_table *t, const void *d_keys, uint64_t n, void *d_out) {
    if (!t || (n && (!d_keys || !d_out))) return fail(TBK_ERR_INVALID, "NULL argument");
    int rc = use_device(t->device);
    if (rc) return rc;
    rc = table_hash(t);
    if (rc || !n) return rc;
    HIP_TRY(tbk_launch_contains(t->view(), (const uint64_t *)d_keys, n, (uint8_t *)d_out, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return TBK_OK;
}

// ---- classifier ------------------------------------------------------------------------
// streams and the ticket ring's events (the current device is the classifier's)
static int classifier_streams(tbk_classifier *c, const tbk_classifier *same_device = nullptr) {
    hipError_t e = hipSuccess;
    if (same_device && same_device->streams && same_device->device == c->device) {
        c->streams = same_device->streams;
    } else {
        c->streams = std::make_shared<DeviceStreams>();
        c->streams->device = c->device;
        e = hipStreamCreateWithFlags(&c->streams->compute, hipStreamNonBlocking);
        if (e == hipSuccess && c->opt.copy_priority != 0) {  // (experiment: the H2D stream at the highest priority the device offers)
            int least = 0, greatest = 0;
            e = hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->streams->copy, hipStreamNonBlocking, greatest);
        } else if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->streams->copy, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->streams->out, hipStreamNonBlocking);
        if (e == hipSuccess && c->opt.h2d_streams >= 2) e = hipStreamCreateWithFlags(&c->streams->copy2, hipStreamNonBlocking);
    }
    c->compute = c->streams->compute; c->copy = c->streams->copy; c->out = c->streams->out;
    for (int i = 0; i < RING && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&c->ring[i].copied, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ring[i].probed, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ring[i].done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ring[i].copied2, hipEventDisableTiming);
        for (int j = 0; j < 8 && e == hipSuccess; j++) e = hipEventCreateWithFlags(&c->ring[i].sliced[j], hipEventDisableTiming);
    }
    if (e != hipSuccess) return fail(TBK_ERR_HIP, "classifier setup: %s", hipGetErrorString(e));
    return TBK_OK;
}

// The paired table for c->mz: allocate, fill with TBK_EMPTY, insert hapA's list, then hapB's minus the
// keys hapA holds: such a key can never count for hapB (hapA is asked first, c/kmers.c:291-294), and
// leaving it out keeps the halves disjoint, so the probe kernel never arbitrates between them.  The
// order pass after each list turns "a key went past this half" / "left the line" into the order of
// the half's last slots, which is what lookups (hapB's inserts included) read.  *past = keys that
// found their own half of their home line full.
// give_up_past != 0: a build whose only purpose may be to find out whether the lists cluster - when hapA's list alone has sent
// more keys than that past their halves (or more than give_up_behind behind their fronts) the answer is known, the table is dropped and *gave_up set (hapB's inserts, which
// look every key up in hapA's crowded half first, are the slow part of such a build).
// pinned_buckets != 0: the table of another device's classifier is built again here, with that one's number of lines.
static int build_pair_table(tbk_classifier *c, ListRef a, ListRef b, double load, uint64_t *past, uint64_t give_up_past = 0, bool *gave_up = nullptr,
                            uint64_t give_up_behind = 0, uint32_t pinned_buckets = 0) {
    c->n_buckets = pinned_buckets ? pinned_buckets
                                  : buckets_for(std::max(a->num_lines, b->num_lines), c->mz.w > 1 ? load : 0.25, 2 * TBK_BUCKET_BYTES, c->opt.table_load, c->opt.memory_budget_bytes);
    const size_t bytes = (size_t)c->n_buckets * 2 * TBK_BUCKET_BYTES;
    hipError_t e = c->alloc_pair(bytes);
    if (e == hipSuccess) e = hipMemset(c->d_pair, 0xFF, bytes);
    if (e != hipSuccess) {
        c->free_pair();
        return fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "paired table (%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    uint32_t *d_over = nullptr, *d_left = nullptr;
    uint64_t past_a = 0, past_b = 0, back_a = 0, back_b = 0;
    int rc = overflow_bitmap((uint64_t)c->n_buckets * 2, &d_over);
    if (!rc && (c->guests & TBK_FLAG_GUESTS)) rc = overflow_bitmap((uint64_t)c->n_buckets * 2, &d_left);
    if (!rc) {
        rc = insert_keys(c->d_pair, c->n_buckets, 16, 0, c->mz, d_over, a->d_keys, a->num_lines, &c->distinct_a,
                         TbkTableView{nullptr, 0, 0, 0, TbkMz{0, 0, 0, 0}, 0}, nullptr, d_left, c->guests, &past_a, &back_a);
        if (!rc && give_up_past && (past_a > give_up_past || (give_up_behind && back_a > give_up_behind))) {
            if (d_over) (void)hipFree(d_over);
            if (d_left) (void)hipFree(d_left);
            c->free_pair();
            c->past_half = past_a; c->behind_front = back_a; c->distinct_b = 0;
            *past = past_a;
            if (gave_up) *gave_up = true;
            return TBK_OK;
        }
        if (!rc) rc = order_table(c->d_pair, (uint64_t)c->n_buckets * 2, d_over, d_left, c->guests, 16);
        if (!rc) rc = insert_keys(c->d_pair, c->n_buckets, 16, 8, c->mz, d_over, b->d_keys, b->num_lines, &c->distinct_b,
                                  TbkTableView{c->d_pair, c->n_buckets, 16, 0, c->mz, c->guests}, &c->shared, d_left, c->guests, &past_b, &back_b);
        if (!rc) rc = order_table(c->d_pair, (uint64_t)c->n_buckets * 2, d_over, d_left, c->guests, 16);
    }
    if (d_over) (void)hipFree(d_over);
    if (d_left) (void)hipFree(d_left);
    if (rc) { c->free_pair(); return rc; }
    c->past_half = past_a + past_b;
    c->behind_front = back_a + back_b;
    *past = past_a + past_b;
    return TBK_OK;
}

// The paired table in entry layout (tbk_common.h "entry layout"): lines of 128 bytes, EMPTY = 0, hapA's list first, then
// hapB's minus the keys hapA holds (tbk_entry_insert_kernel looks them up in hapA's finished half).
// (give_up_behind != 0, full keys: a build that may only show that the lists cluster stops after hapA's list when that alone has put
// more keys than that behind the fronts - *gave_up)
static int build_entry_table(tbk_classifier *c, ListRef a, ListRef b, uint32_t n_buckets, uint64_t give_up_behind = 0, bool *gave_up = nullptr) {
    c->n_buckets = n_buckets;
    const size_t bytes = (size_t)c->n_buckets * 2 * TBK_BUCKET_BYTES;
    hipError_t e = c->alloc_pair(bytes);
    if (e == hipSuccess) e = hipMemset(c->d_pair, 0, bytes);
    unsigned long long *d_cnt = nullptr, cnt[2][8];
    int *d_failed = nullptr, failed = 0;
    memset(cnt, 0, sizeof cnt);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, sizeof cnt[0]);
    if (e == hipSuccess) e = hipMalloc((void **)&d_failed, sizeof(int));
    if (e == hipSuccess) e = hipMemset(d_failed, 0, sizeof(int));
    for (int list = 0; list < 2 && e == hipSuccess; list++) {
        const ListRef t = list ? b : a;
        e = hipMemset(d_cnt, 0, sizeof cnt[0]);
        if (e == hipSuccess) e = (c->guests & TBK_FLAG_FULL)
                                     ? tbk_launch_full_insert(c->d_pair, c->n_buckets, list ? 8u : 0u, c->mz, c->k, t->d_keys, t->num_lines, list, d_cnt, d_failed, 0u, nullptr)
                                     : tbk_launch_entry_insert(c->d_pair, c->n_buckets, list ? 8u : 0u, c->mz, c->k, t->d_keys, t->num_lines, list, (c->guests & TBK_FLAG_WIDE) != 0, d_cnt, d_failed, nullptr);
        if (e == hipSuccess) e = hipMemcpy(cnt[list], d_cnt, sizeof cnt[0], hipMemcpyDeviceToHost);  // (synchronises: hapB's inserts read hapA's finished half)
        if (e == hipSuccess && list == 0 && give_up_behind && cnt[0][3] > give_up_behind) {
            if (gave_up) *gave_up = true;
            break;
        }
    }
    if (e == hipSuccess) e = hipMemcpy(&failed, d_failed, sizeof failed, hipMemcpyDeviceToHost);
    if (d_cnt) (void)hipFree(d_cnt);
    if (d_failed) (void)hipFree(d_failed);
    if (e != hipSuccess || failed) {
        c->free_pair();
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "paired table in entry layout (%zu bytes): %s", bytes, hipGetErrorString(e));
        return fail(TBK_ERR_HIP, "table insert overflowed (table full)");
    }
    c->distinct_a = cnt[0][0]; c->distinct_b = cnt[1][0];
    c->shared = cnt[1][1];
    c->entries_a = cnt[0][2]; c->entries_b = cnt[1][2];
    c->behind_front = cnt[0][3] + cnt[1][3];
    c->past_half = cnt[0][4] + cnt[1][4];
    return TBK_OK;
}

// The paired table of short keys (tbk_common.h "short keys"): n_buckets lines of 32 words, EMPTY = 0, and behind them the
// overflow table (over_slots 64-bit slots, all ones = empty); hapA's list first, then hapB's minus the keys hapA holds.
static int build_short_table(tbk_classifier *c, ListRef a, ListRef b, uint32_t n_buckets, uint32_t over_slots) {
    c->n_buckets = n_buckets;
    c->over_mask = over_slots - 1;
    const size_t line_bytes = (size_t)c->n_buckets * 2 * TBK_BUCKET_BYTES, bytes = line_bytes + (size_t)over_slots * 8;
    hipError_t e = c->alloc_pair(bytes);
    if (e == hipSuccess) e = hipMemset(c->d_pair, 0, line_bytes);
    if (e == hipSuccess) e = hipMemset((char *)c->d_pair + line_bytes, 0xFF, (size_t)over_slots * 8);
    unsigned long long *d_cnt = nullptr, cnt[2][8];
    int *d_failed = nullptr, failed = 0;
    memset(cnt, 0, sizeof cnt);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, sizeof cnt[0]);
    if (e == hipSuccess) e = hipMalloc((void **)&d_failed, sizeof(int));
    if (e == hipSuccess) e = hipMemset(d_failed, 0, sizeof(int));
    for (int list = 0; list < 2 && e == hipSuccess; list++) {
        const ListRef t = list ? b : a;
        e = hipMemset(d_cnt, 0, sizeof cnt[0]);
        if (e == hipSuccess) e = tbk_launch_short_insert(c->d_pair, c->n_buckets, c->over_mask, (uint32_t)list, c->mz, c->k, t->d_keys, t->num_lines, list, d_cnt, d_failed,
                                                              c->opt.short_line_cap ? c->opt.short_line_cap : 32u, nullptr);  // (short_line_cap: tests fill the overflow table)
        if (e == hipSuccess) e = hipMemcpy(cnt[list], d_cnt, sizeof cnt[0], hipMemcpyDeviceToHost);  // (synchronises: hapB's inserts read hapA's finished words)
    }
    if (e == hipSuccess) e = hipMemcpy(&failed, d_failed, sizeof failed, hipMemcpyDeviceToHost);
    if (d_cnt) (void)hipFree(d_cnt);
    if (d_failed) (void)hipFree(d_failed);
    if (e != hipSuccess || failed) {
        c->free_pair();
        c->over_mask = 0;
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "paired table of short keys (%zu bytes): %s", bytes, hipGetErrorString(e));
        return fail(TBK_ERR_HIP, "table insert overflowed (table full)");
    }
    c->distinct_a = cnt[0][0]; c->distinct_b = cnt[1][0];
    c->shared = cnt[1][1];
    c->entries_a = cnt[0][2] + cnt[0][4]; c->entries_b = cnt[1][2] + cnt[1][4];
    c->behind_front = cnt[0][3] + cnt[1][3] + cnt[0][4] + cnt[1][4];
    c->past_half = cnt[0][4] + cnt[1][4];
    return TBK_OK;
}

extern "C" int tbk_classifier_create(const tbk_table *a, const tbk_table *b, tbk_classifier **out) {
    return tbk_classifier_create_opts(a, b, nullptr, out);
}

// What fraction of uniformly hashed keys lies behind a front of `front` keys when a line gets Poisson(lambda) of them:
// E[max(0, X - front)] / lambda.  The full-key layout's tests for "these lists cluster" are set against it (a memory-capped
// table runs at up to six keys per line, where a uniform list already has a third of its keys behind the three-key front).
static double poisson_behind(double lambda, int front) {
    if (lambda <= 0) return 0;
    double p = exp(-lambda), cdf_part = 0, mean_part = 0;  // P(X = x); sum over x <= front of P(x) and of x P(x)
    for (int x = 0; x <= front; x++) { cdf_part += p; mean_part += x * p; p *= lambda / (x + 1); }
    // E[max(0, X - front)] = (lambda - mean_part) - front * (1 - cdf_part)
    return std::max(0.0, ((lambda - mean_part) - front * (1.0 - cdf_part)) / lambda);
}

static int classifier_build(const tbk_table *a, const tbk_table *b, const tbk_options *options, tbk_classifier **out) {
    if (!out) return fail(TBK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!a || !b) return fail(TBK_ERR_INVALID, "table is NULL");
    if (a->device != b->device) return fail(TBK_ERR_INVALID, "tables live on different devices (%d, %d)", a->device, b->device);
    // The reference takes the window length from haplotype_A->k but packs hapB lookups with
    // hapB's own k (c/kmers.c:251-253,278-290) — only meaningful when both are equal.
    if (a->k != b->k) return fail(TBK_ERR_INVALID, "the two k-mer lists have different k (%d and %d)", a->k, b->k);
    int rc = use_device(a->device);
    if (rc) return rc;
    release_list_staging();  // (the lists are loaded: their pinned staging buffers go)
    tbk_classifier *c = new tbk_classifier();
    c->device = a->device;
    c->k = a->k;
    const tbk_options o = c->opt = resolve_options(options);
    c->max_blocks = o.probe_max_blocks;
    c->packed_h2d = o.packed_h2d != 0;
    c->slice_bases = std::max<uint64_t>(2048, o.slice_bases ? o.slice_bases : (uint64_t)384 << 20);
    const uint64_t budget = o.memory_budget_bytes;  // (0: 60 % of the device's total memory)
    // Bucket selection: an m-mer sampled from the k-mer's central span (6 to 8 m-mers, see span_for below;
    // TBK_MINIMIZER_W = 0: plain hashing of the whole key) by mod-sampling, which switches lines 18 % less often than the random
    // minimizer; the load is 0.08 (100 B of HBM per key).  In which layout the probe reads a line is decided by
    // the lists.  Front layout (tbk_common.h): the probe kernel asks for 64 bytes of a line, the first four slots
    // of each list (a list's fifth key of a bucket sits, tagged, in a free front slot of the other list first); a
    // window that misses in a front whose list has keys behind it is queued, and the queue is settled 16 windows
    // at a time from the back half of the line, which the L2 holds already (tbk_kernels.hip: drain_back).  One
    // request per line instead of two: the faster layout on lists whose keys fall evenly into buckets (BASELINE's
    // uniform lists: 184 Gbases/s; at load 0.12, 0.7 % of the keys behind a front: 169 against 151 in whole lines).
    // Lists that cluster the way real find-unique-kmers output does (the k overlapping k-mers around one variant
    // share ~5 sampled m-mers, in both lists at once) overflow the fronts - one window in 13 to 17 needs the back
    // half of its line - and are read in whole lines (two requests per line, both halves at hand).  Measured on
    // haplotype-shaped lists, same box, resident, Gbases/s (profiles/r03/ab_policy.log, ab_policy_whole.log):
    //   k = 21, 2 x 3e8 keys, load 0.08 (60 GB): whole lines, mod-sampling 137, random minimizer 135; front-first
    //     122 / 129 (front-first at load 0.04, 120 GB: 133 / 145 - twice the memory for 6 %);
    //   k = 31, 2 x 1e9 keys (186 GB): whole lines, mod-sampling 153, random minimizer 145.
    // Nothing of that is observable in the results and building the table takes a fraction of a second, so: build
    // front-first; if more than TBK_CLUSTERED (default 0.3 %) of the keys found their own half of their home line
    // full (uniform lists: 1e-5; haplotype-shaped: 2-10 %), or more than TBK_BEHIND_FRONT (default 5 %) lie behind
    // a front, build the same table again in whole lines.  TBK_MOD_SAMPLING=0 pins the random minimizer, TBK_FRONT
    // the layout, TBK_TABLE_LOAD the load.
    const double pin = o.mod_sampling;
    const int w_pin = o.minimizer_w, m_force = o.minimizer_m;
    const uint64_t n_big = std::max(a->num_lines, b->num_lines);
    const uint32_t guests = c->k < 32 && o.guests != 0 ? TBK_FLAG_GUESTS : 0u;
    const double front_pin = o.front;
    // The span: as many m-mers as k leaves room for beside an m long enough for the keys, up to 8 - a window then
    // switches lines with density 3/(2w+1): 0.176 at w = 8 against 0.231 at w = 6 (k = 21 has room for 6 only;
    // k = 25, 2 x 3e8 keys: 208 against 182 Gbases/s; k = 31, 2 x 1e9 keys: 221 against 193 -
    // profiles/r03/ab_spans.log).  That is for the front layout: a longer span also means longer runs of a
    // clustered list's keys in one bucket, and the whole-line kernels of w = 7, 8 keep 4 waves per SIMD, so whole
    // lines stay at w = 6 (k = 31, haplotype-shaped lists: 153 at w = 6, 134 at w = 8).  TBK_MINIMIZER_W pins it.
    auto span_for = [&](bool front_layout) {
        if (w_pin >= 0) return tbk_mz_params(c->k, w_pin, n_big, m_force, pin != 0);
        if (front_layout && pin != 0 && m_force <= 0)
            for (int w = 8; w > 6; w--) {
                const TbkMz z = tbk_mz_params(c->k, w, n_big, m_force, 1);
                if (z.w == w && z.t > 0) return z;
            }
        return tbk_mz_params(c->k, 6, n_big, m_force, pin != 0);
    };
    // The entry layout: a span of 6 m-mers where k leaves room for the flanks in an entry's 30 bits (k = 21 .. 23), shorter
    // spans up to k = 25.  Its table is sized by the ENTRIES, which are known only once it is built: the first build guesses
    // four keys per entry (what runs around SNPs give at w = 6), a second one follows when that was off by more than a
    // quarter.  TBK_ENTRY_LOAD: entries per list and bucket (default 0.5: the four front slots of a line are whoever comes
    // first's, so a bucket overflows its front with its fifth entry - the haplotype-shaped lists of the bench: 2 x 3e8 keys =
    // 1.5e8 entries in 19 GB, 32 bytes per key, 3 % of the entries behind a front; with 3w t-mer positions per span the
    // probe runs 15.45 ms at 0.5, 15.63 at 0.64 (15 GB, 25 bytes per key), 16.45 at 0.8: profiles/r04/ab_loads_span3.log;
    // with 2w positions 0.40 .. 0.64 ran alike: ab_entry_layout.log).
    const double entry_pin = o.entries;
    const bool span3_on = o.span3 != 0;  // (0: narrow entries and short keys rank 2w t-mer positions, as the key layouts do)
    // TBK_BUILD_TIMING=1: every build of the paired table with its duration, on stderr
    const bool build_timing = o.build_timing != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what, bool kept) {
        if (!build_timing) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "tbk build: %-28s %7.3f s  %u lines, %llu + %llu keys, behind a front %llu, past %llu%s\n", what, std::chrono::duration<double>(now - t_last).count(), c->n_buckets,
                (unsigned long long)c->distinct_a, (unsigned long long)c->distinct_b, (unsigned long long)c->behind_front, (unsigned long long)c->past_half, kept ? "  <- kept" : "");
        t_last = now;
    };
    auto try_entry_layout = [&](bool forced) -> bool {
        if (pin == 0 || w_pin == 0 || c->k > 32) return false;
        TbkMz z{0, 0, 0, 0};
        TbkEntryGeom g;
        bool ok = false, wide = false;
        if (o.wide_entries <= 0)
            for (int w = (w_pin > 0 ? w_pin : 6); w >= (w_pin > 0 ? w_pin : 4) && !ok; w--) {
                z = tbk_mz_params(c->k, w, n_big, m_force, 1);
                ok = z.w == w && tbk_entry_geom(c->k, z, &g);
                if (ok && span3_on) z = tbk_mz_span3(z);  // 3w t-mer positions where t stays at 4 or more: 9 % fewer switches of line
            }
        // k-mers too long for a slot's worth of context (k > 25: a k-mer and its neighbours under one m-mer are k + w - 1
        // bases): WIDE entries, 16 bytes, and the longest span k's parity allows, 8 down to 6 (tbk_common.h "wide entries").
        // Two entries of either list in the front: 0.25 entries per list and bucket (TBK_WENTRY_LOAD; 2 x 1e9 haplotype-shaped
        // 31-mers: 102 GB, 51 B per key, 209 Gbases/s resident; 0.20: 128 GB, 215-218; 0.30: 85 GB, 203).  TBK_ENTRY_WIDE=1
        // asks for them at any k.
        // m-mers of 18 bases where k has room (then 17, 16): mod-sampling samples only the m-mers that hold one of the span's
        // smallest t-mers at offset 0 or w, about an eighth of them, and 2 x 2e8 entries over 16-mers crowd the buckets they
        // share whatever the table's size (tbk_common.h "wide entries").
        if (!ok && o.wide_entries != 0)
        {
            // (a span of eight 20-mers ranked over 24 t-mer positions, t = 4, was measured for wide entries: 18.5 ms against 18.1 - the
            // kernel needs 77 registers for it, six waves per SIMD instead of seven; EXPERIMENTS.md)
            for (int mm = (m_force > 0 ? m_force : 18); mm >= (m_force > 0 ? m_force : 16) && !ok; mm--)
                for (int w = (w_pin > 0 ? w_pin : 8); w >= (w_pin > 0 ? w_pin : 6) && !ok; w--) {
                    z = tbk_mz_params(c->k, w, n_big, mm, 1);
                    ok = wide = z.w == w && z.m == mm && z.t > 0 && tbk_wentry_geom(c->k, z, &g);
                }
        }
        if (!ok) return false;
        const TbkMz keep_mz = c->mz;
        const uint32_t keep_flags = c->guests;
        const double el = wide ? std::min(3.5, std::max(0.02, o.wentry_load > 0 ? o.wentry_load : 0.25))   // (four entries per list and line)
                               : std::min(7.0, std::max(0.02, o.entry_load > 0 ? o.entry_load : 0.5));  // (tests crowd the lines: sixteen slots that both lists share)
        c->mz = z;
        c->guests = TBK_FLAG_ENTRY | (wide ? TBK_FLAG_WIDE : 0u);
        double want = (double)n_big / ((wide ? 5.0 : 4.0) * el);
        bool built = false;
        for (int attempt = 0; attempt < 6 && !built; attempt++) {
            uint64_t nb = (uint64_t)want + 16;
            size_t free_b = 0, total_b = 0;
            if (budget) nb = std::min<uint64_t>(nb, budget / 128);
            else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) nb = std::min<uint64_t>(nb, (uint64_t)(0.6 * (double)total_b / 128.0)); else (void)hipGetLastError();
            if (nb > 0x3FFFFFF0ull) nb = 0x3FFFFFF0ull;  // (bits 30 and 31 of a bucket index are flags in the probe's queues)
            c->free_pair();
            const int brc = build_entry_table(c, a, b, (uint32_t)nb);
            lap(wide ? "wide entries" : "entries", false);
            if (brc) {
                // the guess of keys per entry was too high for these lists and the table ran full: twice the room (an allocation
                // that failed, or a table that is full at the device's cap, ends the attempt)
                if ((double)nb + 17.0 < want) break;  // (the table was capped already: more room is not to be had)
                const std::string why = g_err;
                if (why.find("full") == std::string::npos) break;
                want *= 2;
                continue;
            }
            const double target = (double)std::max<uint64_t>(std::max(c->entries_a, c->entries_b), 1) / el;
            if (attempt >= 4 || ((double)nb >= 0.8 * target && (double)nb <= 1.25 * target)) { built = true; break; }
            want = target;
        }
        if (!built) { c->free_pair(); c->mz = keep_mz; c->guests = keep_flags; c->entries_a = c->entries_b = 0; return false; }
        const double ratio = (double)(c->distinct_a + c->distinct_b) / (double)std::max<uint64_t>(1, c->entries_a + c->entries_b);
        if (!forced && ratio < o.entry_min_ratio) { c->free_pair(); c->mz = keep_mz; c->guests = keep_flags; c->entries_a = c->entries_b = 0; return false; }
        return true;
    };
    // Short keys (tbk_common.h): lists whose keys do not merge into entries (BASELINE's uniform k-mers) in 4 bytes a key
    // instead of 8, read through the entry kernels' two-lane window loop: 8 keys of either list in a 32-byte front, 32 in a
    // line.  TBK_SHORT_LOAD keys per line (default 2.3: 56 bytes of HBM per key; 2 x 3e8 uniform 21-mers: 33.4 GB, 2 % of the
    // keys behind a front).  Built first where k and the table's size allow (k = 21: any table of 65536 lines or more,
    // k = 25: 2 GB or more); lists that cluster (more than TBK_BEHIND_FRONT of the keys behind a front) go on to the key
    // layout's test and from there to entries, as before.  TBK_SHORT=0: never, 1: whatever the lists look like.
    const double short_pin = o.short_keys;
    double short_behind = -1;  // the fraction of the keys a short-key build found behind a front (-1: none was built)
    auto try_short_layout = [&](bool forced) -> bool {
        if (pin == 0 || w_pin == 0 || c->k > 31 || c->k < 17) return false;
        if (!forced && o.table_load > 0) return false;  // (the key layouts' load is pinned: the key layouts are meant)
        TbkMz z = span_for(true);  // (the front layout's span: as long as k leaves room for, up to 8 m-mers)
        if (z.w < 2 || z.t <= 0 || z.m > 16) return false;
        if (span3_on) z = tbk_mz_span3(z);
        const double per_line = std::min(24.0, std::max(0.1, o.short_load > 0 ? o.short_load : 2.3));
        uint64_t nb = (uint64_t)((double)(a->num_lines + b->num_lines) / per_line) + 16;
        const uint32_t min_nb = tbk_short_min_buckets(c->k, z);
        if (!min_nb) return false;
        if (nb < min_nb) { if (!forced && (double)min_nb > 2.0 * (double)nb) return false; nb = min_nb; }  // (a small table may be up to twice its size for it)
        if (nb > 0x3FFFFFF0ull) return false;
        TbkShortGeom g;
        if (!tbk_short_geom(c->k, z, (uint32_t)nb, &g)) return false;
        size_t free_b = 0, total_b = 0;
        if (budget) { if ((double)nb * 128.0 > (double)budget) return false; }
        else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) { if ((double)nb * 128.0 > 0.6 * (double)total_b) return false; } else (void)hipGetLastError();
        const TbkMz keep_mz = c->mz;
        const uint32_t keep_flags = c->guests;
        c->mz = z;
        c->guests = TBK_FLAG_SHORT;
        bool built = false;
        uint64_t over = 4096;
        while (over < (a->num_lines + b->num_lines) / 256) over <<= 1;
        for (int attempt = 0; attempt < 4 && !built; attempt++, over <<= 3) {
            c->free_pair();
            if (over > (1ull << 31)) break;
            const int brc = build_short_table(c, a, b, (uint32_t)nb, (uint32_t)over);
            lap("short keys", false);
            if (brc == TBK_OK) { built = true; c->layout_builds++; break; }
            if (g_err.find("full") == std::string::npos) break;
        }
        const double n_keys = (double)std::max<uint64_t>(1, c->distinct_a + c->distinct_b);
        if (built) short_behind = (double)c->behind_front / n_keys;
        if (built && !forced && short_behind > o.behind_front) built = false;  // the lists cluster
        if (!built) { c->free_pair(); c->mz = keep_mz; c->guests = keep_flags; c->over_mask = 0; c->entries_a = c->entries_b = 0; return false; }
        return true;
    };
    // Full keys (tbk_common.h): lists that do not merge and do not fit short keys - uniform 26- to 31-mers, 23- / 25-mers whose
    // tables need m-mers beyond 16 bases - as 64-bit keys in the entry kernels' line: three keys and the line's summary in a
    // 32-byte front read by two lanes, twelve more behind it.  full_load keys per line (default 2.0: 64 bytes of device
    // memory per key; the key layout they had until round 5: 93-100).  Lists that cluster (more than behind_front x 6 of the
    // keys behind a front - a fifth of a uniform list's keys is, with three keys to a front; runs of overlapping k-mers put
    // well over half there) go on to the key layout's test and to entries.  full_keys = 0: never, 1: whatever the lists look like.
    const double full_pin = o.full_keys;
    double sample_behind = -1;  // what the bucket sample of a full-key attempt found behind the fronts (-1: none was taken)
    bool full_clustered = false;  // a full-key attempt (its sample, or its build) has shown the lists to cluster: on to entries without the key layout's test build
    auto try_full_layout = [&](bool forced) -> bool {
        if (pin == 0 || w_pin == 0 || c->k > 31 || c->k < 17) return false;
        if (!forced && o.table_load > 0) return false;  // (the key layouts' load is pinned: the key layouts are meant)
        TbkMz z = span_for(true);
        if (!tbk_full_geom(c->k, z) || z.t != z.m - z.w) return false;
        const double per_line = std::min(12.0, std::max(0.05, o.full_load > 0 ? o.full_load : 2.0));
        uint64_t nb = (uint64_t)((double)(a->num_lines + b->num_lines) / per_line) + 16;
        size_t free_b = 0, total_b = 0;
        double cap = (double)budget;
        if (!budget) { if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap = 0.6 * (double)total_b; else (void)hipGetLastError(); }
        if (cap > 0 && (double)nb * 128.0 > cap) {
            if ((double)(a->num_lines + b->num_lines) / (cap / 128.0) > 6.0) return false;  // (a line holds fifteen keys: beyond six to a line on average the key layout's denser loads do better)
            nb = (uint64_t)(cap / 128.0);
        }
        if (nb > 0x3FFFFFF0ull) nb = 0x3FFFFFF0ull;
        // Do the lists' keys crowd their buckets?  A sample BY BUCKET first: hapA's list hashed as for the whole table, stored only
        // where the home bucket lies in the table's first sixteenth (tbk_full_insert_kernel, only_below) - whatever order the list is
        // in, those lines fill as they would in the whole table.  Uniform keys: 2 % of what is stored lies behind a front (one
        // list, half the load); runs of overlapping k-mers: a third and more.  2 x 1e9 keys: 0.2 s instead of a 3 s build thrown away.
        if (!forced && nb >= ((uint64_t)1 << 22)) {
            const uint32_t sample_nb = (uint32_t)(nb / 16);
            uint64_t *d_sample = nullptr;
            unsigned long long *d_cnt = nullptr, cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int *d_failed = nullptr;
            hipError_t e = hipMalloc((void **)&d_sample, (size_t)sample_nb * 128);
            if (e == hipSuccess) e = hipMemset(d_sample, 0, (size_t)sample_nb * 128);
            if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, sizeof cnt);
            if (e == hipSuccess) e = hipMemset(d_cnt, 0, sizeof cnt);
            if (e == hipSuccess) e = hipMalloc((void **)&d_failed, sizeof(int));
            if (e == hipSuccess) e = hipMemset(d_failed, 0, sizeof(int));
            if (e == hipSuccess) e = tbk_launch_full_insert(d_sample, (uint32_t)nb, 0u, z, c->k, a->d_keys, a->num_lines, 0, d_cnt, d_failed, sample_nb, nullptr);
            if (e == hipSuccess) e = hipMemcpy(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost);
            if (d_sample) (void)hipFree(d_sample);
            if (d_cnt) (void)hipFree(d_cnt);
            if (d_failed) (void)hipFree(d_failed);
            if (e != hipSuccess) { (void)hipGetLastError(); return false; }
            const double f = (double)cnt[3] / (double)std::max<unsigned long long>(1, cnt[2]);
            // (hapA's list alone at this table's load: what a uniform list would show, times 2.5 - never below the option's own figure)
            const double sample_limit = std::max(4.0 * o.behind_front, 2.5 * poisson_behind((double)a->num_lines / (double)nb, 3));
            if (build_timing) fprintf(stderr, "tbk build: full keys, a sixteenth of the buckets: %llu of %llu slots behind a front (%.1f %%; clustered from %.1f %%)\n", cnt[3], cnt[2], 100.0 * f, 100.0 * sample_limit);
            t_last = std::chrono::steady_clock::now();
            if (f > sample_limit) {
                sample_behind = f; full_clustered = true;  // (the lists cluster: the caller goes on to entries without the key layout's test build)
                return false;
            }
        }
        const TbkMz keep_mz = c->mz;
        const uint32_t keep_flags = c->guests;
        c->mz = z;
        c->guests = TBK_FLAG_FULL;
        c->free_pair();
        bool gave_up = false;
        // (hapA's list alone, at half the load: a uniform list has 2 % of its keys behind a front then, a clustered one a third: given up at 20 %)
        const double lam = (double)(a->num_lines + b->num_lines) / (double)nb, lam_a = (double)a->num_lines / (double)nb;
        const double built_limit = std::max(6.0 * o.behind_front, 2.5 * poisson_behind(lam, 3));
        // (given up after hapA's list when that alone - at its own load - is 2.5 times past a uniform list's expectation, never below the option's figure)
        const uint64_t give_up = forced ? 0 : (uint64_t)(std::max(2.0 * o.behind_front * (double)(a->num_lines + b->num_lines), 2.5 * poisson_behind(lam_a, 3) * (double)a->num_lines)) + 1;
        const int brc = build_entry_table(c, a, b, (uint32_t)nb, give_up, &gave_up);
        lap(gave_up ? "full keys (given up after hapA's list)" : "full keys", false);
        bool built = brc == TBK_OK && !gave_up;
        if (built) c->layout_builds++;
        const double n_keys = (double)std::max<uint64_t>(1, c->distinct_a + c->distinct_b);
        if (built && !forced && (double)c->behind_front / n_keys > built_limit) {  // the lists cluster
            sample_behind = std::max(sample_behind, (double)c->behind_front / n_keys);  // (measured: the caller need not build the key layout to find out)
            full_clustered = true;
            built = false;
        }
        if (!built) { c->free_pair(); c->mz = keep_mz; c->guests = keep_flags; c->entries_a = c->entries_b = 0; return false; }
        return true;
    };
    if (entry_pin <= 0 && short_pin != 0 && front_pin < 0 && try_short_layout(short_pin > 0)) {
        c->own_pair();
        rc = classifier_streams(c);
        if (rc) { tbk_classifier_destroy(c); return rc; }
        *out = c;
        return TBK_OK;
    }
    // By themselves only where short keys were ALLOWED and do not apply (k, or m-mers beyond 16 bases: short_behind < 0 - lists a
    // short-key build has measured are not in need of full keys); with short keys switched off the key layouts are meant.
    if (entry_pin <= 0 && front_pin < 0 && (full_pin > 0 || (full_pin < 0 && short_pin < 0 && short_behind < 0)) && try_full_layout(full_pin > 0)) {
        c->own_pair();
        rc = classifier_streams(c);
        if (rc) { tbk_classifier_destroy(c); return rc; }
        *out = c;
        return TBK_OK;
    }
    if (entry_pin > 0 && try_entry_layout(true)) {
        c->layout_builds++;
        c->own_pair();
        rc = classifier_streams(c);
        if (rc) { tbk_classifier_destroy(c); return rc; }
        *out = c;
        return TBK_OK;
    }
    // Lists that the short-key build has just shown to cluster plainly (more than TBK_PLAINLY_CLUSTERED, default 12 %, of
    // the keys behind an 8-slot front; haplotype-shaped lists: 16 %, uniform ones 1.5 %) skip the key layout's test -
    // building a front-first key table of clustered lists only to measure them takes 2 s at 2 x 3e8 keys, ten times a
    // build of entries - and go to entries at once; lists that do not merge come back here.
    if ((short_behind > o.plainly_clustered || full_clustered) && entry_pin != 0 && front_pin < 0 && try_entry_layout(false)) {
        c->layout_builds++;
        lap("(kept)", true);
        c->own_pair();
        rc = classifier_streams(c);
        if (rc) { tbk_classifier_destroy(c); return rc; }
        *out = c;
        return TBK_OK;
    }
    bool front = front_pin != 0;
    for (;;) {
        c->mz = span_for(front);
        if (c->mz.w < 2) front = false;  // (plain hashing and one-m-mer spans have no front layout)
        c->free_pair();
        c->layout_builds++;
        c->guests = guests | (front ? TBK_FLAG_FRONT : 0u);
        uint64_t past = 0;
        // (a front-first build that may be rejected gives up after hapA's list when that alone shows the clustering: the
        // lists' lines bound the distinct keys from above, so the test below would fail for certain)
        const bool testing = front && front_pin < 0;
        bool gave_up = false;
        rc = build_pair_table(c, a, b, 0.08, &past, testing ? (uint64_t)(o.clustered * (double)(a->num_lines + b->num_lines)) + 1 : 0, &gave_up,
                              testing ? (uint64_t)(o.behind_front * (double)(a->num_lines + b->num_lines)) + 1 : 0);
        lap(gave_up ? "key layout, front (given up after hapA's list)" : front ? "key layout, front" : "key layout, whole lines", false);
        if (rc) { c->free_pair(); delete c; return rc; }
        const double n_keys = (double)std::max<uint64_t>(1, c->distinct_a + c->distinct_b);
        const double clustered = (double)past / n_keys, behind = (double)c->behind_front / n_keys;
        if (!gave_up && (!front || front_pin > 0 || (clustered <= o.clustered && behind <= o.behind_front))) break;
        // Clustered lists.  Lists shaped like real find-unique-kmers output cluster because they ARE runs of overlapping
        // k-mers: the entry layout stores a run once (tbk_common.h), which brings them back to a front - two slots per
        // list, asked for by two lanes - in a table a quarter of the size.  Kept when the lists really merge (at least 1.5
        // keys per entry); lists that cluster for another reason go to whole lines, as before.  TBK_ENTRY=0: never.
        if (entry_pin != 0 && try_entry_layout(false)) { c->layout_builds++; break; }
        front = false;  // clustered lists, or too many keys behind the fronts: whole lines
    }
    c->own_pair();
    rc = classifier_streams(c);
    if (rc) { tbk_classifier_destroy(c); return rc; }
    *out = c;
    return TBK_OK;
}

// ---- every build looked at again (tbk_options.verify_build) ----
// The reference stores every list line (c/kmers.c:112-122) and finds every stored canonical key (:245-268).  The entry layouts
// MERGE keys while thousands of threads insert at once - narrow entries OR window bits into a word others are ORing into, wide
// entries guard two words with a lock made of relaxed agent-scope atomics (tbk_kernels.hip: tbk_wentry_insert_kernel; the
// formally sufficient acquire / release costs 25 s per 2e9 keys in L2 write-backs) - and a key lost or misfiled there leaves
// no trace in any counter.  So a table built that way is asked for every line of both lists before it is handed out, by
// default (-1); verify_build = 1 does so for every layout, 0 never.  2 x 3e8 keys: under a second.
extern "C" int tbk_verify_expectations_(tbk_table *a, tbk_table *b, void *d_expect);
extern "C" int tbk_classifier_verify_expect_(tbk_classifier *c, const void *d_keys_a, uint64_t na, const void *d_keys_b, uint64_t nb, int k, const void *d_expect,
                                             uint64_t out[5]);

static bool verify_wanted(const tbk_classifier *c) {
    return c->opt.verify_build > 0 || (c->opt.verify_build < 0 && (c->guests & TBK_FLAG_ENTRY) != 0);
}

// what the lists say of their own lines, one byte per line (hapA's first), on the lists' device; the standalone tables this
// needs are dropped again when they were built for it (32 B per key: 64 GB at 2 x 1e9)
static int list_expectations(const tbk_table *a, const tbk_table *b, uint8_t **d_expect) {
    *d_expect = nullptr;
    int rc = use_device(a->device);
    if (rc) return rc;
    tbk_table *ta = const_cast<tbk_table *>(a), *tb = const_cast<tbk_table *>(b);
    const bool had_a = ta->hashed, had_b = tb->hashed;
    HIP_TRY(hipMalloc((void **)d_expect, a->num_lines + b->num_lines + 8));
    rc = tbk_verify_expectations_(ta, tb, *d_expect);
    auto drop = [](tbk_table *t) { if (t->d_slots) (void)hipFree(t->d_slots); t->d_slots = nullptr; t->hashed = false; };
    if (!had_a) drop(ta);
    if (!had_b && tb != ta) drop(tb);
    if (rc) { (void)hipFree(*d_expect); *d_expect = nullptr; }
    return rc;
}

// c's table asked for every list line (the keys and the expectations where c lives)
static int verify_table(tbk_classifier *c, const uint64_t *ka, uint64_t na, const uint64_t *kb, uint64_t nb, const uint8_t *d_expect) {
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t r[5];
    int rc = tbk_classifier_verify_expect_(c, ka, na, kb, nb, c->k, d_expect, r);
    if (rc) return rc;
    c->verify_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (c->opt.build_timing) fprintf(stderr, "tbk build: device %d, %llu list lines looked up again: %llu wrong  %7.3f s\n", c->device, (unsigned long long)r[0], (unsigned long long)r[3], c->verify_s);
    if (r[3]) return fail(TBK_ERR_HIP, "the table built on device %d answers %llu of %llu list lines wrongly (first: line %llu): not using it", c->device,
                          (unsigned long long)r[3], (unsigned long long)r[0], (unsigned long long)r[4]);
    c->verified_lines = r[0];
    return TBK_OK;
}

extern "C" int tbk_classifier_create_opts(const tbk_table *a, const tbk_table *b, const tbk_options *options, tbk_classifier **out) {
    int rc = classifier_build(a, b, options, out);
    if (rc || !verify_wanted(*out)) return rc;
    uint8_t *d_expect = nullptr;
    rc = list_expectations(a, b, &d_expect);
    if (!rc) rc = verify_table(*out, a->d_keys, a->num_lines, b->d_keys, b->num_lines, d_expect);
    if (d_expect) (void)hipFree(d_expect);
    if (rc) { const std::string msg = g_err; tbk_classifier_destroy(*out); *out = nullptr; g_err = msg; }
    return rc;
}

extern "C" int tbk_classifier_verified(const tbk_classifier *c, uint64_t *lines, double *seconds) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    if (lines) *lines = c->verified_lines;
    if (seconds) *seconds = c->verify_s;
    return TBK_OK;
}

// ---- one classifier per device (tables replicated, reads sharded, no collective) ----
// A replica's classifier: everything of `src` but the table and the streams.
static tbk_classifier *replica_shell(const tbk_classifier *src, int device) {
    tbk_classifier *c = new tbk_classifier();
    c->device = device;
    c->k = src->k;
    c->opt = src->opt;
    c->n_buckets = src->n_buckets;
    c->distinct_a = src->distinct_a; c->distinct_b = src->distinct_b; c->shared = src->shared;
    c->mz = src->mz;
    c->max_blocks = src->max_blocks;
    c->packed_h2d = src->packed_h2d;
    c->slice_bases = src->slice_bases;
    c->guests = src->guests;
    c->layout_builds = src->layout_builds; c->past_half = src->past_half; c->behind_front = src->behind_front;
    c->entries_a = src->entries_a; c->entries_b = src->entries_b;
    c->over_mask = src->over_mask;
    return c;
}

static void enable_peer(int device, int other) {  // (the current device is `device`)
    int can = 0;
    if (device != other && hipDeviceCanAccessPeer(&can, device, other) == hipSuccess && can) {
        const hipError_t pe = hipDeviceEnablePeerAccess(other, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
    }
    (void)hipGetLastError();
}

// `bytes` from `src` on src_device to `dst` on the current device `device`, asynchronously on `stream` (a stream of the
// destination: eight destinations' copies leave one source over seven xGMI links at once instead of queueing behind each other)
static hipError_t copy_from_device(void *dst, int device, const void *src, int src_device, size_t bytes, hipStream_t stream) {
    if (!bytes) return hipSuccess;
    if (device == src_device) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    return hipMemcpyPeerAsync(dst, device, src, src_device, bytes, stream);
}

// The finished table of `src`, copied (TBK_REPLICA_COPY=1, tbk_classifier_replicate, and the fallback of a replica whose
// build from the lists could not be made).  56 to 100 bytes per key cross xGMI.
static int replica_by_copy(const tbk_classifier *src, int device, tbk_classifier *c) {
    const size_t bytes = (size_t)c->table_bytes();
    hipError_t e = c->alloc_pair(bytes);
    hipStream_t stream = nullptr;
    if (e == hipSuccess) {
        // the finished table travels device to device (xGMI when the two are peers; the runtime
        // stages through the host otherwise) - once, outside any timed region
        enable_peer(device, src->device);
        e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = copy_from_device(c->d_pair, device, src->d_pair, src->device, bytes, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (stream) (void)hipStreamDestroy(stream);
        c->replica_copies = 1;
    }
    if (e != hipSuccess) {
        c->free_pair();
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "replicating the paired table (%zu bytes) to device %d: %s", bytes, device,
                    hipGetErrorString(e));
    }
    return TBK_OK;
}

// The table of `src` built AGAIN on `device` from the two lists, in the layout and geometry `src` decided on (layout flags, span,
// lines, overflow slots are pinned; no trial builds).  The lists' keys - 8 bytes per key, where the finished table is 56 to 100 -
// are what crosses xGMI when the lists live on another device.  c/kmers.c:185-229 builds a table once and :245-268 only reads it
// afterwards, so a replica built independently answers like the first as long as it holds the same keys: the counts the build
// reports (distinct keys per list, keys both lists hold) are compared with src's.
// *keys_a / *keys_b: where the lists' keys lie on `device` afterwards (the lists' own memory when that is their device; else
// copies the caller frees: *own_a, *own_b) - the verification reads them once more.
static int replica_by_build(const tbk_classifier *src, const tbk_table *a, const tbk_table *b, int device, tbk_classifier *c, const uint64_t **keys_a,
                            const uint64_t **keys_b, uint64_t **own_a, uint64_t **own_b) {
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t *d_a = nullptr, *d_b = nullptr;
    *keys_a = *keys_b = nullptr; *own_a = *own_b = nullptr;
    const uint64_t *ka = a->d_keys, *kb = b->d_keys;
    double copy_s = 0;
    if (device != a->device) {
        enable_peer(device, a->device);
        hipStream_t stream = nullptr;
        hipError_t e = hipMalloc((void **)&d_a, std::max<size_t>(8, a->num_lines * 8));
        if (e == hipSuccess) e = hipMalloc((void **)&d_b, std::max<size_t>(8, b->num_lines * 8));
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = copy_from_device(d_a, device, a->d_keys, a->device, a->num_lines * 8, stream);
        if (e == hipSuccess) e = copy_from_device(d_b, device, b->d_keys, b->device, b->num_lines * 8, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (stream) (void)hipStreamDestroy(stream);
        if (e != hipSuccess) {
            if (d_a) (void)hipFree(d_a);
            if (d_b) (void)hipFree(d_b);
            (void)hipGetLastError();
            return fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "the lists' keys (%llu bytes) to device %d: %s",
                        (unsigned long long)((a->num_lines + b->num_lines) * 8), device, hipGetErrorString(e));
        }
        ka = d_a; kb = d_b;
        copy_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const ListRef la(ka, a->num_lines), lb(kb, b->num_lines);
    int rc;
    if (src->guests & TBK_FLAG_SHORT) rc = build_short_table(c, la, lb, src->n_buckets, src->over_mask + 1);
    else if (src->guests & (TBK_FLAG_ENTRY | TBK_FLAG_FULL)) rc = build_entry_table(c, la, lb, src->n_buckets);
    else {
        uint64_t past = 0;
        rc = build_pair_table(c, la, lb, 0.08, &past, 0, nullptr, 0, src->n_buckets);
    }
    if (!rc && (c->distinct_a != src->distinct_a || c->distinct_b != src->distinct_b || c->shared != src->shared)) {
        c->free_pair();
        rc = fail(TBK_ERR_HIP, "the table built on device %d holds %llu + %llu keys (%llu in both lists), the one on device %d %llu + %llu (%llu)", device,
                    (unsigned long long)c->distinct_a, (unsigned long long)c->distinct_b, (unsigned long long)c->shared, src->device,
                    (unsigned long long)src->distinct_a, (unsigned long long)src->distinct_b, (unsigned long long)src->shared);
    }
    if (rc) {
        if (d_a) (void)hipFree(d_a);
        if (d_b) (void)hipFree(d_b);
        return rc;
    }
    *keys_a = ka; *keys_b = kb; *own_a = d_a; *own_b = d_b;
    c->replica_copies = 2;
    c->layout_builds = 1;
    if (src->opt.build_timing) {
        (void)hipDeviceSynchronize();
        fprintf(stderr, "tbk build: replica on device %d from the lists  %7.3f s (the lists' keys: %.3f s)  %u lines, %llu + %llu keys\n", device,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), copy_s, c->n_buckets, (unsigned long long)c->distinct_a, (unsigned long long)c->distinct_b);
    }
    return TBK_OK;
}

extern "C" int tbk_classifier_replicate(const tbk_classifier *src, int device, tbk_classifier **out) {
    if (!out) return fail(TBK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!src) return fail(TBK_ERR_INVALID, "classifier is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    tbk_classifier *c = replica_shell(src, device);
    // TBK_FORCE_REPLICA=1: a ring on the device that holds the table gets a full replica of its own all the same,
    // made by the very calls a second GPU's replica is made by (peer query, peer copy) - how a one-GPU box
    // executes and checks the replica path of an 8-GPU node (tests/test_gpu_multi.py).
    const bool force_replica = src->opt.force_replica != 0;
    if (device == src->device && src->pair_owner && !force_replica) {
        // another stream ring on the device that holds the table already: the table is read-only, so it is shared
        c->d_pair = src->d_pair;
        c->pair_owner = src->pair_owner;
        c->replica_copies = 0;
    } else {
        rc = replica_by_copy(src, device, c);
        if (rc) { delete c; return rc; }
        c->own_pair();
    }
    rc = classifier_streams(c, src->opt.ring_streams != 0 ? nullptr : src);  // (TBK_RING_STREAMS=1: streams of its own, the measurement above)
    if (rc) { tbk_classifier_destroy(c); return rc; }
    *out = c;
    return TBK_OK;
}

extern "C" int tbk_classifier_create_multi(const tbk_table *a, const tbk_table *b, const int *devices, int n_devices, tbk_classifier **out) {
    return tbk_classifier_create_multi_opts(a, b, devices, n_devices, nullptr, out);
}

// The fan-out.  The lists are hashed once on their own device, which decides the layout (short keys, entries, ... and the
// table's geometry); every OTHER device then gets the lists' keys (8 B per key over its own xGMI link) and builds the same
// table itself - one host thread per device, all at once.  Until round 5 the finished table (56-100 B per key) was copied to
// one device after another by a blocking hipMemcpyPeer out of the first: seven times 33 GB at configs[2], 102-128 GB at
// configs[4].  tbk_options.replica_copy = 1 keeps the copy (now asynchronous, per destination), and a device whose build
// fails (no room for the lists beside the table) falls back to it.  Further entries of a device that has a table share it.
extern "C" int tbk_classifier_create_multi_opts(const tbk_table *a, const tbk_table *b, const int *devices, int n_devices, const tbk_options *options,
                                                tbk_classifier **out) {
    if (!out || !devices || n_devices < 1) return fail(TBK_ERR_INVALID, "devices/out is NULL or n_devices < 1");
    for (int i = 0; i < n_devices; i++) out[i] = nullptr;
    int n_visible = 0;
    if (hipGetDeviceCount(&n_visible) != hipSuccess || n_visible <= 0) return fail(TBK_ERR_NO_DEVICE, "no HIP device visible; libtbk_hip has no CPU fallback");
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= n_visible) return fail(TBK_ERR_INVALID, "device %d out of range (0..%d)", devices[i], n_visible - 1);
    tbk_classifier *first = nullptr;
    int rc = classifier_build(a, b, options, &first);
    if (rc) return rc;
    // (tbk_options.verify_build: what the lists say of their own lines is worked out once, where the lists are; every table made
    // below - the first, the ones built again from the lists, copies - is asked for all of them on its own device)
    uint8_t *d_expect = nullptr;
    const bool verify = verify_wanted(first);
    if (verify) {
        rc = list_expectations(a, b, &d_expect);
        if (!rc) rc = verify_table(first, a->d_keys, a->num_lines, b->d_keys, b->num_lines, d_expect);
        if (rc) { const std::string msg = g_err; if (d_expect) { (void)hipSetDevice(a->device); (void)hipFree(d_expect); } tbk_classifier_destroy(first); g_err = msg; return rc; }
    }
    const bool force_replica = first->opt.force_replica != 0, by_copy = first->opt.replica_copy != 0;
    const auto t0 = std::chrono::steady_clock::now();
    // who makes a table (a "leader": the first entry of a device other than the lists', or with force_replica every entry)
    // and who shares one (leader_of[i] = the entry whose table entry i shares; -1 = first's)
    std::vector<int> leader_of(n_devices, -2);
    std::vector<int> leaders;
    bool placed = false;
    for (int i = 0; i < n_devices; i++) {
        if (!placed && devices[i] == first->device) { out[i] = first; placed = true; leader_of[i] = i; continue; }
        if (!force_replica) {
            if (devices[i] == first->device) { leader_of[i] = -1; continue; }
            int l = -2;
            for (int j : leaders) if (devices[j] == devices[i]) { l = j; break; }
            if (l >= 0) { leader_of[i] = l; continue; }
        }
        leader_of[i] = i;
        leaders.push_back(i);
    }
    std::vector<int> rcs(n_devices, TBK_OK);
    std::vector<std::string> errs(n_devices);
    auto make = [&](int i) {
        const int device = devices[i];
        int r = use_device(device);
        tbk_classifier *c = nullptr;
        const uint64_t *ka = nullptr, *kb = nullptr;
        uint64_t *own_a = nullptr, *own_b = nullptr;
        if (!r) {
            c = replica_shell(first, device);
            r = by_copy ? TBK_ERR_HIP : replica_by_build(first, a, b, device, c, &ka, &kb, &own_a, &own_b);
            if (r && !by_copy && first->opt.build_timing) fprintf(stderr, "tbk build: replica on device %d from the lists failed (%s): copying the table\n", device, g_err.c_str());
            if (r) {
                delete c;
                c = replica_shell(first, device);
                r = replica_by_copy(first, device, c);
            }
            if (!r) {
                c->own_pair();
                r = classifier_streams(c, first->opt.ring_streams != 0 ? nullptr : first);
            } else { delete c; c = nullptr; }
            if (!r && verify) {
                // the keys (a copy made for the build, the lists' own on their device, or - behind a copied table - fetched now) and
                // the expectations, on this device
                uint8_t *d_exp = d_expect;
                uint8_t *own_exp = nullptr;
                hipError_t e = hipSuccess;
                const uint64_t na = a->num_lines, nb = b->num_lines;
                if (device != a->device) {
                    enable_peer(device, a->device);
                    hipStream_t stream = nullptr;
                    e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
                    if (e == hipSuccess) e = hipMalloc((void **)&own_exp, na + nb + 8);
                    if (e == hipSuccess) e = copy_from_device(own_exp, device, d_expect, a->device, na + nb, stream);
                    if (e == hipSuccess && !ka) {
                        e = hipMalloc((void **)&own_a, std::max<size_t>(8, na * 8));
                        if (e == hipSuccess) e = hipMalloc((void **)&own_b, std::max<size_t>(8, nb * 8));
                        if (e == hipSuccess) e = copy_from_device(own_a, device, a->d_keys, a->device, na * 8, stream);
                        if (e == hipSuccess) e = copy_from_device(own_b, device, b->d_keys, b->device, nb * 8, stream);
                        ka = own_a; kb = own_b;
                    }
                    if (e == hipSuccess) e = hipStreamSynchronize(stream);
                    if (stream) (void)hipStreamDestroy(stream);
                    d_exp = own_exp;
                } else if (!ka) { ka = a->d_keys; kb = b->d_keys; }
                if (e != hipSuccess) { (void)hipGetLastError(); r = fail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "the lists' keys to device %d for verification: %s", device, hipGetErrorString(e)); }
                if (!r) r = verify_table(c, ka, na, kb, nb, d_exp);
                if (own_exp) (void)hipFree(own_exp);
            }
            if (own_a) (void)hipFree(own_a);
            if (own_b) (void)hipFree(own_b);
            if (r && c) { const std::string msg = g_err; tbk_classifier_destroy(c); c = nullptr; g_err = msg; }
        }
        out[i] = c;
        rcs[i] = r;
        if (r) errs[i] = g_err;
    };
    if (leaders.size() == 1) make(leaders[0]);
    else if (!leaders.empty()) {
        std::vector<std::thread> threads;
        for (int i : leaders) threads.emplace_back(make, i);
        for (auto &t : threads) t.join();
        (void)hipSetDevice(first->device);
    }
    for (int i : leaders) if (rcs[i] && !rc) { rc = rcs[i]; g_err = errs[i]; }
    if (d_expect) { (void)hipSetDevice(a->device); (void)hipFree(d_expect); (void)hipSetDevice(first->device); }
    for (int i = 0; i < n_devices && !rc; i++) {
        if (out[i]) continue;
        const tbk_classifier *src = leader_of[i] < 0 ? first : out[leader_of[i]];
        rc = tbk_classifier_replicate(src, devices[i], &out[i]);  // (the device has the table: shared)
        if (!rc) { out[i]->verified_lines = src->verified_lines; out[i]->verify_s = src->verify_s; }
    }
    if (!rc && first->opt.build_timing && !leaders.empty())
        fprintf(stderr, "tbk build: %zu replica(s) %s, all at once: %7.3f s\n", leaders.size(), by_copy ? "copied" : "built from the lists",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    if (rc) {
        const std::string msg = g_err;
        for (int i = 0; i < n_devices; i++) { if (out[i] && out[i] != first) tbk_classifier_destroy(out[i]); out[i] = nullptr; }
        tbk_classifier_destroy(first);
        g_err = msg;
        return rc;
    }
    if (!placed) tbk_classifier_destroy(first);
    return TBK_OK;
}

extern "C" int tbk_classifier_device(const tbk_classifier *c) { return c ? c->device : -1; }

extern "C" int tbk_classifier_layout(const tbk_classifier *c, int *minimizer_w, int *minimizer_m, int *span_offset) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    if (minimizer_w) *minimizer_w = c->mz.w;
    if (minimizer_m) *minimizer_m = c->mz.m;
    if (span_offset) *span_offset = c->mz.o;
    return TBK_OK;
}

extern "C" int tbk_classifier_sampling_t(const tbk_classifier *c) { return c ? c->mz.t : 0; }

extern "C" int tbk_classifier_stats(const tbk_classifier *c, uint64_t *distinct_a, uint64_t *distinct_b,
                                    uint64_t *n_buckets, uint64_t *table_bytes) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    if (distinct_a) *distinct_a = c->distinct_a;
    if (distinct_b) *distinct_b = c->distinct_b;
    if (n_buckets) *n_buckets = c->n_buckets;
    if (table_bytes) *table_bytes = c->table_bytes();
    return TBK_OK;
}

extern "C" int tbk_classifier_build_info(const tbk_classifier *c, int *layout_builds, uint64_t *keys_past_half) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    if (layout_builds) *layout_builds = c->layout_builds;
    if (keys_past_half) *keys_past_half = c->past_half;
    return TBK_OK;
}

extern "C" int tbk_classifier_front(const tbk_classifier *c, int *front, uint64_t *keys_behind_front) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    if (front) *front = (c->guests & TBK_FLAG_FRONT) ? 1 : 0;
    if (keys_behind_front) *keys_behind_front = c->behind_front;
    return TBK_OK;
}

extern "C" int tbk_classifier_entries(const tbk_classifier *c, int *entry_layout, uint64_t *entries_a, uint64_t *entries_b) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    if (entry_layout) *entry_layout = (c->guests & TBK_FLAG_FULL) ? 4 : (c->guests & TBK_FLAG_SHORT) ? 3 : (c->guests & TBK_FLAG_ENTRY) ? ((c->guests & TBK_FLAG_WIDE) ? 2 : 1) : 0;  // (3: short keys, 4: full keys)
    if (entries_a) *entries_a = c->entries_a;
    if (entries_b) *entries_b = c->entries_b;
    return TBK_OK;
}

// Random 64-byte reads (a quad per line, as the probe asks for a front) over THIS table where it lies in HBM:
// lines per second.  The same table in another place of the device's memory measures up to 15 % differently
// (EXPERIMENTS.md, round 3); this is the yardstick for that.
extern "C" int tbk_classifier_calibrate(tbk_classifier *c, double *lines_per_sec) {
    if (!c || !lines_per_sec) return fail(TBK_ERR_INVALID, "NULL argument");
    int rc = use_device(c->device);
    if (rc) return rc;
    const uint64_t bytes = ((uint64_t)c->n_buckets * 2 * TBK_BUCKET_BYTES) & ~(uint64_t)255;
    if (bytes < (1u << 20)) { *lines_per_sec = 0; return TBK_OK; }
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc((void **)&sink, 16));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uint64_t done = 0;
    float ms = 0;
    const uint64_t n_lines = (uint64_t)1 << 27;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = tbk_launch_gather(c->d_pair, bytes, 64, 4, 4, n_lines >> 3, 5, sink, &done, c->compute);  // warm-up
    if (e == hipSuccess) e = hipEventRecord(e0, c->compute);
    if (e == hipSuccess) e = tbk_launch_gather(c->d_pair, bytes, 64, 4, 4, n_lines, 9, sink, &done, c->compute);
    if (e == hipSuccess) e = hipEventRecord(e1, c->compute);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(TBK_ERR_HIP, "tbk_classifier_calibrate: %s", hipGetErrorString(e));
    *lines_per_sec = ms > 0 ? (double)done / (ms * 1e-3) : 0;
    return TBK_OK;
}

// The same over this table in the entry kernels' own request shape (one-wave blocks, `waves_per_simd` of them resident, two lanes x
// 16 bytes of a line's first 32, `inflight` lines per pair before any is used): the ceiling bench.py prices the window loop's line
// rate against - same table, same box, same process (where a table's pages lie moves the rate by +-2 %: EXPERIMENTS.md).
extern "C" hipError_t tbk_launch_gather_pairs(const void *, uint64_t, int, int, uint64_t, uint64_t, uint32_t *, uint64_t *, hipStream_t);
extern "C" int tbk_classifier_calibrate_pairs(tbk_classifier *c, int inflight, int waves_per_simd, uint64_t n_lines, double *lines_per_sec) {
    if (!c || !lines_per_sec) return fail(TBK_ERR_INVALID, "NULL argument");
    int rc = use_device(c->device);
    if (rc) return rc;
    const uint64_t bytes = ((uint64_t)c->n_buckets * 2 * TBK_BUCKET_BYTES) & ~(uint64_t)255;
    if (bytes < (1u << 20) || n_lines < 4096) { *lines_per_sec = 0; return TBK_OK; }
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc((void **)&sink, 16));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uint64_t done = 0;
    float ms = 0;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = tbk_launch_gather_pairs(c->d_pair, bytes, inflight, waves_per_simd, n_lines >> 3, 5, sink, &done, c->compute);  // warm-up
    if (e == hipSuccess) e = hipEventRecord(e0, c->compute);
    if (e == hipSuccess) e = tbk_launch_gather_pairs(c->d_pair, bytes, inflight, waves_per_simd, n_lines, 9, sink, &done, c->compute);
    if (e == hipSuccess) e = hipEventRecord(e1, c->compute);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? TBK_ERR_INVALID : TBK_ERR_HIP, "tbk_classifier_calibrate_pairs: %s", hipGetErrorString(e));
    *lines_per_sec = ms > 0 ? (double)done / (ms * 1e-3) : 0;
    return TBK_OK;
}

// Which memory a classifier's table lies in (two classifiers with equal ids share one table), and whether that
// table is a replica copied from another classifier's.
extern "C" int tbk_classifier_table_id(const tbk_classifier *c, uint64_t *table_id, int *is_replica) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    if (table_id) *table_id = (uint64_t)(uintptr_t)c->d_pair;
    if (is_replica) *is_replica = c->replica_copies;
    return TBK_OK;
}

extern "C" int tbk_classifier_shared_keys(const tbk_classifier *c, uint64_t *n_shared) {
    if (!c || !n_shared) return fail(TBK_ERR_INVALID, "NULL argument");
    *n_shared = c->shared;
    return TBK_OK;
}

extern "C" void tbk_classifier_destroy(tbk_classifier *c) {
    if (!c) return;
    if (hipSetDevice(c->device) == hipSuccess) {
        if (c->compute) (void)hipStreamSynchronize(c->compute);
        if (c->copy) (void)hipStreamSynchronize(c->copy);
        if (c->out) (void)hipStreamSynchronize(c->out);
        for (Slot &s : c->ring) {
            if (s.d_bases) (void)hipFree(s.d_bases);
            if (s.d_offsets) (void)hipFree(s.d_offsets);
            if (s.d_counts) (void)hipFree(s.d_counts);
            if (s.h_bases) (void)hipHostFree(s.h_bases);
            if (s.h_offsets) (void)hipHostFree(s.h_offsets);
            if (s.h_counts) (void)hipHostFree(s.h_counts);
            if (s.d_codes) (void)hipFree(s.d_codes);
            if (s.d_bad) (void)hipFree(s.d_bad);
            if (s.d_exc_chunk) (void)hipFree(s.d_exc_chunk);
            if (s.d_exc_mask) (void)hipFree(s.d_exc_mask);
            if (s.h_codes) (void)hipHostFree(s.h_codes);
            if (s.h_exc_chunk) (void)hipHostFree(s.h_exc_chunk);
            if (s.h_exc_mask) (void)hipHostFree(s.h_exc_mask);
            if (s.copied) (void)hipEventDestroy(s.copied);
            if (s.probed) (void)hipEventDestroy(s.probed);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.copied2) (void)hipEventDestroy(s.copied2);
            for (hipEvent_t e : s.sliced) if (e) (void)hipEventDestroy(e);
        }
        for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
        c->pair_owner.reset();  // (the table itself goes with its last user)
        if (c->d_pass_read) (void)hipFree(c->d_pass_read);
        c->small_free();
        for (const Slot &s : c->ring) if (s.busy && c->streams) c->streams->in_flight--;
        c->streams.reset();  // (the streams go with their last ring)
    }
    delete c;
}

// memcpy of a large buffer over several host threads (staging a pageable batch into pinned memory:
// one thread copies ~10 GB/s, PCIe wants 55)
static void par_memcpy(void *dst, const void *src, size_t n) {
    const size_t piece = (size_t)16 << 20;
    const int nt = (int)std::min<size_t>((size_t)tbk_host_threads(), n / piece);
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) {
        const size_t lo = n * (size_t)t / nt, hi = n * (size_t)(t + 1) / nt;
        pool.emplace_back([=]() { memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    for (std::thread &th : pool) th.join();
}

static bool is_pinned(const void *p) {
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// A slice of a batch's passes and the event (on another stream) behind which its bases are on the device.
struct ProbeSlice { uint64_t pass_lo, pass_hi; hipEvent_t arrived; };

static int launch_probe_timed(tbk_classifier *c, const uint8_t *d_bases, const uint64_t *d_offsets,
                              uint64_t n_reads, uint64_t total, int32_t *d_counts, const uint32_t *d_codes = nullptr,
                              const uint16_t *d_bad16 = nullptr, const ProbeSlice *slices = nullptr, int n_slices = 0) {
    if (n_reads >= 0xFFFFFFF0ull) return fail(TBK_ERR_INVALID, "more than 2^32 reads in one batch");
    if (total == 0) {  // nothing to probe (every read empty): all counts are zero
        HIP_TRY(hipMemsetAsync(d_counts, 0, n_reads * 2 * sizeof(int32_t), c->compute));
        return TBK_OK;
    }
    const uint64_t passes = tbk_probe_passes(total);
    if (passes > c->cap_passes) {
        HIP_TRY(hipStreamSynchronize(c->compute));  // earlier launches may still read the old scratch
        if (c->d_pass_read) HIP_TRY(hipFree(c->d_pass_read));
        c->d_pass_read = nullptr; c->cap_passes = 0;
        const uint64_t cap = (passes + passes / 4 + 1024 + 1) & ~1ull;  // (even: the two-read list's 64-bit entries start at word 2 * cap)
        HIP_TRY(hipMalloc((void **)&c->d_pass_read, (4 * cap + 16) * sizeof(uint32_t)));  // pass -> read, multi-read pass list, two-read (pass, read) list, the lists' lengths
        c->cap_passes = cap;
    }
    const ProbeSlice whole = {0, passes, nullptr};
    if (n_slices <= 0) { slices = &whole; n_slices = 1; }
    // (the per-read counters are cleared by the pass-index kernel)
    hipEvent_t *ev = nullptr;
    if (c->timing) {
        const size_t need = 2 + 3 * (size_t)n_slices;
        if (c->ev_used + need > c->ev.size() && c->ev.size() >= (size_t)TIMING_POOL) {  // fold what we have, then reuse the pool
            HIP_TRY(hipStreamSynchronize(c->compute));
            int rc = c->fold_timing();
            if (rc) return rc;
        }
        while (c->ev_used + need > c->ev.size()) {
            hipEvent_t a;
            HIP_TRY(hipEventCreate(&a));
            c->ev.push_back(a);
        }
        ev = c->ev.data() + c->ev_used;
        c->ev_used += need;
        c->ev_slices.push_back((uint32_t)n_slices);
        c->timed_launches++;
        HIP_TRY(hipEventRecord(ev[0], c->compute));
    }
    HIP_TRY(tbk_launch_probe_index(d_offsets, n_reads, total, d_counts, c->d_pass_read, c->cap_passes, c->opt.two_read_kernel != 0 && tbk_probe_has_two_read_kernel(c->mz), c->compute));
    if (ev) HIP_TRY(hipEventRecord(ev[1], c->compute));
    for (int j = 0; j < n_slices; j++) {
        if (slices[j].arrived) HIP_TRY(hipStreamWaitEvent(c->compute, slices[j].arrived, 0));
        if (ev) HIP_TRY(hipEventRecord(ev[2 + 3 * j], c->compute));  // (behind the wait: the slice's own time starts when its bases are there)
        HIP_TRY(tbk_launch_probe_range(d_bases, d_codes, d_bad16, d_offsets, n_reads, total, c->pair(), c->k, d_counts, c->d_pass_read, c->cap_passes,
                                       slices[j].pass_lo, slices[j].pass_hi, c->max_blocks, c->opt.two_read_kernel != 0, ev ? ev[3 + 3 * j] : nullptr, c->compute, 0));
        if (ev) HIP_TRY(hipEventRecord(ev[4 + 3 * j], c->compute));
    }
    c->last_passes = passes;
    return TBK_OK;
}

int tbk_classifier::fold_timing() {
    size_t at = 0;
    for (uint32_t n_slices : ev_slices) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev[at], ev[at + 1]));  // the pass-index kernel
        timed_ms += ms;
        for (uint32_t j = 0; j < n_slices; j++) {
            float whole = 0, single = 0;
            HIP_TRY(hipEventElapsedTime(&whole, ev[at + 2 + 3 * j], ev[at + 4 + 3 * j]));
            HIP_TRY(hipEventElapsedTime(&single, ev[at + 3 + 3 * j], ev[at + 4 + 3 * j]));
            timed_ms += whole;
            timed_single_ms += single;
        }
        at += 2 + 3 * (size_t)n_slices;
    }
    ev_used = 0;
    ev_slices.clear();
    return TBK_OK;
}

extern "C" int tbk_stream_depth(const tbk_classifier *) { return RING; }

// A host batch's offsets (every entry point that takes one): offsets[0] = 0, non-decreasing.
// offsets[n_reads] sizes the copies and bounds the kernels, so a malformed array must not get past here.
extern "C" int tbk_check_offsets_(const uint64_t *offsets, uint64_t n_reads) {
    if (!offsets) return fail(TBK_ERR_INVALID, "offsets is NULL");
    if (offsets[0] != 0) return fail(TBK_ERR_INVALID, "offsets[0] must be 0");
    for (uint64_t i = 0; i < n_reads; i++)
        if (offsets[i + 1] < offsets[i]) return fail(TBK_ERR_INVALID, "offsets not non-decreasing at read %llu", (unsigned long long)i);
    return TBK_OK;
}

// device buffers of a slot
static int slot_reserve_device(Slot &s, uint64_t total, uint64_t n_reads) {
    const size_t need_b = ((size_t)total + 15) & ~(size_t)15;
    if (need_b > s.cap_bases) {
        if (s.d_bases) HIP_TRY(hipFree(s.d_bases));
        s.d_bases = nullptr; s.cap_bases = 0;
        const size_t cap = std::max(need_b + need_b / 8, (size_t)1 << 20);
        HIP_TRY(hipMalloc((void **)&s.d_bases, cap));
        s.cap_bases = cap;
    }
    if (n_reads > s.cap_reads) {
        if (s.d_offsets) HIP_TRY(hipFree(s.d_offsets));
        if (s.d_counts) HIP_TRY(hipFree(s.d_counts));
        s.d_offsets = nullptr; s.d_counts = nullptr; s.cap_reads = 0;
        const size_t cap = std::max((size_t)n_reads + (size_t)n_reads / 8, (size_t)1024);
        HIP_TRY(hipMalloc((void **)&s.d_offsets, (cap + 1) * sizeof(uint64_t)));
        HIP_TRY(hipMalloc((void **)&s.d_counts, cap * 2 * sizeof(int32_t)));
        s.cap_reads = cap;
    }
    return TBK_OK;
}

// pinned host staging of a slot: `stage_bases` bytes of bases (0 = none), offsets/counts staging
// when `stage_small` / `stage_out`
static int slot_reserve(Slot &s, uint64_t stage_bases, uint64_t n_reads, bool stage_small, bool stage_out) {
    const size_t need_b = ((size_t)stage_bases + 15) & ~(size_t)15;
    if (need_b > s.hcap_bases) {
        if (s.h_bases) HIP_TRY(hipHostFree(s.h_bases));
        s.h_bases = nullptr; s.hcap_bases = 0;
        const size_t cap = std::max(need_b + need_b / 8, (size_t)1 << 20);
        HIP_TRY(hipHostMalloc((void **)&s.h_bases, cap, hipHostMallocPortable));
        s.hcap_bases = cap;
    }
    if ((stage_small || stage_out) && n_reads > s.hcap_reads) {
        if (s.h_offsets) HIP_TRY(hipHostFree(s.h_offsets));
        if (s.h_counts) HIP_TRY(hipHostFree(s.h_counts));
        s.h_offsets = nullptr; s.h_counts = nullptr; s.hcap_reads = 0;
        const size_t cap = std::max((size_t)n_reads + (size_t)n_reads / 8, (size_t)1024);
        HIP_TRY(hipHostMalloc((void **)&s.h_offsets, (cap + 1) * sizeof(uint64_t), hipHostMallocPortable));
        HIP_TRY(hipHostMalloc((void **)&s.h_counts, cap * 2 * sizeof(int32_t), hipHostMallocPortable));
        s.hcap_reads = cap;
    }
    return TBK_OK;
}

// device buffers of a slot for a packed batch
static int slot_reserve_packed(Slot &s, uint64_t n_chunks, uint64_t n_exc) {
    if (n_chunks > s.cap_chunks) {
        if (s.d_codes) HIP_TRY(hipFree(s.d_codes));
        if (s.d_bad) HIP_TRY(hipFree(s.d_bad));
        s.d_codes = nullptr; s.d_bad = nullptr; s.cap_chunks = 0;
        const size_t cap = std::max((size_t)n_chunks + (size_t)n_chunks / 8, (size_t)1 << 16);
        HIP_TRY(hipMalloc((void **)&s.d_codes, cap * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&s.d_bad, (cap + 2) * sizeof(uint16_t)));
        s.cap_chunks = cap;
        s.bad_clean = false;
    }
    if (n_exc > s.cap_exc) {
        if (s.d_exc_chunk) HIP_TRY(hipFree(s.d_exc_chunk));
        if (s.d_exc_mask) HIP_TRY(hipFree(s.d_exc_mask));
        s.d_exc_chunk = nullptr; s.d_exc_mask = nullptr; s.cap_exc = 0;
        const size_t cap = std::max((size_t)n_exc + (size_t)n_exc / 4, (size_t)4096);
        HIP_TRY(hipMalloc((void **)&s.d_exc_chunk, cap * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&s.d_exc_mask, cap * sizeof(uint16_t)));
        s.cap_exc = cap;
    }
    return TBK_OK;
}

// pinned staging of a slot for a batch packed at submit time
static int slot_reserve_packed_host(Slot &s, uint64_t n_chunks, uint64_t n_exc) {
    if (n_chunks > s.hcap_chunks) {
        if (s.h_codes) HIP_TRY(hipHostFree(s.h_codes));
        s.h_codes = nullptr; s.hcap_chunks = 0;
        const size_t cap = std::max((size_t)n_chunks + (size_t)n_chunks / 8, (size_t)1 << 16);
        HIP_TRY(hipHostMalloc((void **)&s.h_codes, cap * sizeof(uint32_t), hipHostMallocPortable));
        s.hcap_chunks = cap;
    }
    if (n_exc > s.hcap_exc) {
        if (s.h_exc_chunk) HIP_TRY(hipHostFree(s.h_exc_chunk));
        if (s.h_exc_mask) HIP_TRY(hipHostFree(s.h_exc_mask));
        s.h_exc_chunk = nullptr; s.h_exc_mask = nullptr; s.hcap_exc = 0;
        const size_t cap = std::max((size_t)n_exc + (size_t)n_exc / 4, (size_t)4096);
        HIP_TRY(hipHostMalloc((void **)&s.h_exc_chunk, cap * sizeof(uint32_t), hipHostMallocPortable));
        HIP_TRY(hipHostMalloc((void **)&s.h_exc_mask, cap * sizeof(uint16_t), hipHostMallocPortable));
        s.hcap_exc = cap;
    }
    return TBK_OK;
}

// The common body of the host-batch submits.  Exactly one of `bases` (ASCII) and `codes` (packed) is
// given; src pointers must stay valid until the ticket is waited for unless they were staged here.
static int submit_host(tbk_classifier *c, const uint8_t *bases, const uint32_t *codes, const uint32_t *exc_chunk, const uint16_t *exc_mask,
                       uint64_t n_exc, const uint64_t *offsets, uint64_t n_reads, int32_t *counts, uint64_t *ticket) {
    if (!c || !offsets || !ticket || (n_reads && !counts)) return fail(TBK_ERR_INVALID, "NULL argument");
    int rc = tbk_check_offsets_(offsets, n_reads);
    if (rc) return rc;
    const uint64_t total = offsets[n_reads];
    if (total && !bases && !codes) return fail(TBK_ERR_INVALID, "bases is NULL");
    if (codes && n_exc && (!exc_chunk || !exc_mask)) return fail(TBK_ERR_INVALID, "exception arrays are NULL");
    rc = use_device(c->device);
    if (rc) return rc;
    const uint64_t tk = c->next_ticket;
    Slot &s = c->ring[tk % RING];
    if (s.busy) return fail(TBK_ERR_STATE, "all %d stream slots are in flight; call tbk_stream_wait(%llu) first", RING,
                            (unsigned long long)s.ticket);
    const uint64_t n_chunks = tbk_packed_chunks(total);
    bool packed = codes != nullptr;
    if (!packed && total && c->packed_h2d) {
        // Pack on the host (all host threads) straight from the caller's memory into the slot's pinned
        // staging: a quarter of the bytes cross PCIe, and a pageable batch needs no staging copy (the
        // packer reads it once, where a copy into pinned memory would read and write it).
        rc = slot_reserve_packed_host(s, n_chunks, 0);
        if (rc) return rc;
        rc = tbk_pack_bases_vec(bases, total, s.h_codes, c->exc_chunk, c->exc_mask, c->pack_threads);
        if (rc) return fail(rc, "%s", g_err.c_str());
        n_exc = c->exc_chunk.size();
        rc = slot_reserve_packed_host(s, n_chunks, n_exc);
        if (rc) return rc;
        if (n_exc) {
            memcpy(s.h_exc_chunk, c->exc_chunk.data(), n_exc * sizeof(uint32_t));
            memcpy(s.h_exc_mask, c->exc_mask.data(), n_exc * sizeof(uint16_t));
        }
        codes = s.h_codes; exc_chunk = s.h_exc_chunk; exc_mask = s.h_exc_mask;
        packed = true;
    } else if (packed && total) {
        // caller-packed arrays: pinned ones are copied from where they lie, pageable ones are staged
        const bool codes_pinned = is_pinned(codes), exc_pinned = n_exc == 0 || (is_pinned(exc_chunk) && is_pinned(exc_mask));
        rc = slot_reserve_packed_host(s, codes_pinned ? 0 : n_chunks, exc_pinned ? 0 : n_exc);
        if (rc) return rc;
        if (!codes_pinned) { par_memcpy(s.h_codes, codes, n_chunks * sizeof(uint32_t)); codes = s.h_codes; }
        if (!exc_pinned) {
            memcpy(s.h_exc_chunk, exc_chunk, n_exc * sizeof(uint32_t));
            memcpy(s.h_exc_mask, exc_mask, n_exc * sizeof(uint16_t));
            exc_chunk = s.h_exc_chunk; exc_mask = s.h_exc_mask;
        }
        for (uint64_t i = 0; i < n_exc; i++)
            if (exc_chunk[i] >= n_chunks) return fail(TBK_ERR_INVALID, "exception %llu names chunk %u of %llu", (unsigned long long)i, exc_chunk[i], (unsigned long long)n_chunks);
    }
    // pinned inputs are copied straight from the caller's memory; pageable ones go through the
    // slot's pinned staging (bases and offsets decided independently: a reader batch has its
    // bases pinned and its small offsets array in ordinary memory)
    const bool bases_pinned = packed || total == 0 || is_pinned(bases);
    const bool offs_pinned = is_pinned(offsets);
    const bool out_pinned = n_reads == 0 || is_pinned(counts);
    rc = slot_reserve(s, bases_pinned ? 0 : total, n_reads, !bases_pinned || !offs_pinned, !out_pinned);
    if (rc) return rc;
    rc = slot_reserve_device(s, packed ? 0 : total, n_reads);
    if (rc) return rc;
    if (packed) {
        rc = slot_reserve_packed(s, n_chunks, n_exc);
        if (rc) return rc;
    }
    s.ticket = tk; s.n_reads = n_reads; s.user_counts = counts; s.counts_staged = !out_pinned;
    if (n_reads && total) {
        const uint64_t *src_o = offsets;
        if (!offs_pinned) { memcpy(s.h_offsets, offsets, (n_reads + 1) * sizeof(uint64_t)); src_o = s.h_offsets; }
        const uint8_t *src_b = bases;
        if (!packed && !bases_pinned) { par_memcpy(s.h_bases, bases, total); src_b = s.h_bases; }
        std::lock_guard<std::mutex> together(c->streams->enqueue);  // (rings sharing the device's streams: one batch's launches stay together)
        // Side stream: the H2D of this batch overlaps the previous batch's kernels.  The small arrays go first; the
        // bases follow - when the ring is empty in up to 8 slices, each with its own event, and the probe of a slice's
        // passes starts as soon as that slice is on the device: an empty pipeline starts computing after an eighth of
        // its first copy instead of after all of it.
        HIP_TRY(hipMemcpyAsync(s.d_offsets, src_o, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->copy));
        if (packed) {
            if (!s.bad_clean) HIP_TRY(hipMemsetAsync(s.d_bad, 0, (s.cap_chunks + 2) * sizeof(uint16_t), c->copy));  // a new buffer, or a batch that failed half-way
            s.bad_clean = false;
            if (n_exc) {
                HIP_TRY(hipMemcpyAsync(s.d_exc_chunk, exc_chunk, n_exc * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy));
                HIP_TRY(hipMemcpyAsync(s.d_exc_mask, exc_mask, n_exc * sizeof(uint16_t), hipMemcpyHostToDevice, c->copy));
            }
        }
        HIP_TRY(hipEventRecord(s.copied, c->copy));
        const uint64_t passes = tbk_probe_passes(total);
        // Slices pay where nothing else keeps the device busy (an empty ring: the first batch of a run); behind a
        // batch that is still in flight one copy and one pair of kernels is best (every kernel launch ends in a
        // partly filled device: eight slices per batch measured 4 % slower in the steady state).
        const bool zero_copy = packed && c->opt.zero_copy != 0 && is_pinned(codes);
        const bool alone = c->streams->in_flight.load() == 0;  // (over all rings of the device)
        const int n_slices = alone ? (int)std::max<uint64_t>(1, std::min<uint64_t>(8, total / c->slice_bases)) : 1;
        ProbeSlice slices[8];
        uint64_t sent = 0;  // chunks (of 16 bases) on their way so far
        for (int j = 0; j < n_slices; j++) {
            slices[j].pass_lo = passes * (uint64_t)j / (uint64_t)n_slices;
            slices[j].pass_hi = passes * (uint64_t)(j + 1) / (uint64_t)n_slices;
            slices[j].arrived = s.sliced[j];
            // a pass reads 130 chunks from its first one: the slice needs the stream up to chunk pass_hi * 128 + 2
            const uint64_t upto = j + 1 == n_slices ? n_chunks : std::min<uint64_t>(n_chunks, slices[j].pass_hi * (TBK_PASS_BASES / 16) + 2);
            if (upto > sent) {
                if (packed && zero_copy) {
                    // (experiment, TBK_ZERO_COPY=1) no copy of the codes at all: the probe's coalesced tile loads read them from the
                    // caller's pinned memory over PCIe themselves
                } else if (packed && n_slices == 1 && c->streams->copy2 && upto - sent >= ((uint64_t)1 << 22)) {
                    // steady state, two H2D streams: the second half of the codes travels on a DMA engine of its own
                    const uint64_t mid = sent + (upto - sent) / 2;
                    HIP_TRY(hipMemcpyAsync(s.d_codes + mid, codes + mid, (upto - mid) * sizeof(uint32_t), hipMemcpyHostToDevice, c->streams->copy2));
                    HIP_TRY(hipEventRecord(s.copied2, c->streams->copy2));
                    HIP_TRY(hipMemcpyAsync(s.d_codes + sent, codes + sent, (mid - sent) * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy));
                    HIP_TRY(hipStreamWaitEvent(c->copy, s.copied2, 0));
                } else if (packed) {
                    HIP_TRY(hipMemcpyAsync(s.d_codes + sent, codes + sent, (upto - sent) * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy));
                } else {
                    const uint64_t b0 = sent * 16, b1 = std::min<uint64_t>(total, upto * 16);
                    HIP_TRY(hipMemcpyAsync(s.d_bases + b0, src_b + b0, b1 - b0, hipMemcpyHostToDevice, c->copy));
                }
                sent = upto;
            }
            HIP_TRY(hipEventRecord(s.sliced[j], c->copy));
        }
        HIP_TRY(hipStreamWaitEvent(c->compute, s.copied, 0));
        // exceptions into the dense masks (the tail of the last partial chunk is masked there too): a kernel, so on the
        // compute stream - the copy stream carries copies only and never waits for a free compute unit
        if (packed) HIP_TRY(tbk_launch_scatter_bad(s.d_exc_chunk, s.d_exc_mask, n_exc, s.d_bad, total, 0, c->compute));
        rc = launch_probe_timed(c, packed ? nullptr : s.d_bases, s.d_offsets, n_reads, total, s.d_counts, packed ? (zero_copy ? const_cast<uint32_t *>(codes) : s.d_codes) : nullptr,
                                packed ? s.d_bad : nullptr, slices, n_slices);
        if (rc) return rc;
        if (packed) {
            // behind the probe: the batch's exceptions out of the dense masks again, which are all zero between batches
            HIP_TRY(tbk_launch_scatter_bad(s.d_exc_chunk, s.d_exc_mask, n_exc, s.d_bad, total, 1, c->compute));
            s.bad_clean = true;
        }
        // the counts travel home on a stream of their own: a copy between two probes on the compute stream costs
        // the next probe a hand-over to the copy engine and back (measured: 1 ms per 25 ms step)
        HIP_TRY(hipEventRecord(s.probed, c->compute));
        HIP_TRY(hipStreamWaitEvent(c->out, s.probed, 0));
        HIP_TRY(hipMemcpyAsync(out_pinned ? counts : s.h_counts, s.d_counts, n_reads * 2 * sizeof(int32_t),
                               hipMemcpyDeviceToHost, c->out));
        HIP_TRY(hipEventRecord(s.done, c->out));
    } else {
        if (n_reads) memset(counts, 0, n_reads * 2 * sizeof(int32_t));
        s.counts_staged = false;
        HIP_TRY(hipEventRecord(s.done, c->compute));
    }
    s.busy = true;
    c->streams->in_flight++;
    c->next_ticket++;
    *ticket = tk;
    return TBK_OK;
}

extern "C" int tbk_stream_submit(tbk_classifier *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                                 int32_t *counts, uint64_t *ticket) {
    return submit_host(c, bases, nullptr, nullptr, nullptr, 0, offsets, n_reads, counts, ticket);
}

extern "C" int tbk_stream_submit_packed(tbk_classifier *c, const uint32_t *codes, const uint32_t *exc_chunk, const uint16_t *exc_mask,
                                        uint64_t n_exc, const uint64_t *offsets, uint64_t n_reads, int32_t *counts, uint64_t *ticket) {
    if (!codes && offsets && n_reads && offsets[n_reads]) return fail(TBK_ERR_INVALID, "codes is NULL");
    static const uint32_t none = 0;
    return submit_host(c, nullptr, codes ? codes : &none, exc_chunk, exc_mask, n_exc, offsets, n_reads, counts, ticket);
}

extern "C" int tbk_classifier_set_transfer(tbk_classifier *c, int packed) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    c->packed_h2d = packed != 0;
    return TBK_OK;
}

extern "C" int tbk_classifier_transfer(const tbk_classifier *c) { return c ? c->packed_h2d : -1; }

// (library-internal) the share of the host threads this classifier's submit-time packing may use
extern "C" int tbk_classifier_set_pack_threads_(tbk_classifier *c, int threads) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    c->pack_threads = threads > 0 ? threads : 0;
    return TBK_OK;
}

extern "C" int tbk_stream_wait(tbk_classifier *c, uint64_t ticket) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    Slot &s = c->ring[ticket % RING];
    if (!s.busy || s.ticket != ticket) return fail(TBK_ERR_STATE, "ticket %llu is not in flight", (unsigned long long)ticket);
    int rc = use_device(c->device);
    if (rc) return rc;
    HIP_TRY(hipEventSynchronize(s.done));
    if (s.counts_staged && s.n_reads) memcpy(s.user_counts, s.h_counts, s.n_reads * 2 * sizeof(int32_t));
    s.busy = false;
    c->streams->in_flight--;
    return TBK_OK;
}

// 1: that ticket's batch is complete (tbk_stream_wait will not block), 0: not yet; < -1: an error
extern "C" int tbk_stream_query(tbk_classifier *c, uint64_t ticket) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL") - 10;
    Slot &s = c->ring[ticket % RING];
    if (!s.busy || s.ticket != ticket) return fail(TBK_ERR_STATE, "ticket %llu is not in flight", (unsigned long long)ticket) - 10;
    if (use_device(c->device)) return TBK_ERR_HIP - 10;
    const hipError_t e = hipEventQuery(s.done);
    if (e == hipSuccess) return 1;
    (void)hipGetLastError();
    return e == hipErrorNotReady ? 0 : TBK_ERR_HIP - 10;
}

extern "C" int tbk_classify_batch(tbk_classifier *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                                  int32_t *counts) {
    uint64_t tk = 0;
    int rc = tbk_stream_submit(c, bases, offsets, n_reads, counts, &tk);
    if (rc) return rc;
    return tbk_stream_wait(c, tk);
}

extern "C" void *tbk_host_alloc(size_t bytes) {
    void *p = nullptr;
    // (experiment: TBK_HOST_ALLOC_FLAGS ORs hipHostMalloc flags in - 4: write-combined, 0x80000000: non-coherent, 0x40000000: coherent;
    // none moved the H2D rate under load, EXPERIMENTS.md)
    static const unsigned extra = (unsigned)env_double("TBK_HOST_ALLOC_FLAGS", 0);
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | extra) != hipSuccess) {
        fail(TBK_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}
extern "C" void tbk_host_free(void *p) { if (p) (void)hipHostFree(p); }

// Pinned staging memory for the big streaming buffers (the reader's GPU windows, the bin writer's text, the batches): transparent huge
// pages mapped by the process itself and registered with the runtime, instead of hipHostMalloc.  Measured on the pool's boxes
// (tools/pin_cost.hip, 432 MB): hipHostMalloc 65-88 ms + hipHostFree 47-78 ms; mmap + MADV_HUGEPAGE + touch 27 ms, hipHostRegister
// 1.5 ms, unregister + munmap 17 ms - and the same 57 GB/s both ways over the link.  A run that holds 2-3 GB of such buffers used to
// spend half a second getting them and giving them back.  Used only as the source or target of hipMemcpyAsync (no kernel reads it
// through a device pointer).  Where mapping or registering fails, and under TBK_PINNED=malloc: hipHostMalloc.
namespace {
struct PinnedMap { void *map; size_t len; bool registered; };
std::mutex pinned_mu;
std::vector<std::pair<void *, PinnedMap>> pinned_maps;   // (a handful of live buffers: a vector does)
}  // namespace

extern "C" void *tbk_pin_alloc_(size_t bytes) {
    if (!bytes) bytes = 1;
    static const bool plain = [] { const char *e = getenv("TBK_PINNED"); return e && strcmp(e, "malloc") == 0; }();
    constexpr size_t HUGE = (size_t)2 << 20;
    if (!plain && bytes >= HUGE) {
        const size_t len = (bytes + HUGE - 1) & ~(HUGE - 1);
        void *m = mmap(nullptr, len + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED) {
            uint8_t *a = (uint8_t *)(((uintptr_t)m + HUGE - 1) & ~(uintptr_t)(HUGE - 1));
            (void)madvise(a, len, MADV_HUGEPAGE);
            (void)madvise(a, len, MADV_DONTFORK);   // (a child - subprocess.run in the caller - must not share pages the device copies into: no copy-on-write under a transfer)
            for (size_t off = 0; off < len; off += 4096) a[off] = 0;   // (the pages exist before they are pinned: faulted in here, two megabytes at a time)
            if (hipHostRegister(a, len, hipHostRegisterPortable) == hipSuccess) {
                std::lock_guard<std::mutex> lk(pinned_mu);
                pinned_maps.emplace_back((void *)a, PinnedMap{m, len + HUGE, true});
                return a;
            }
            (void)hipGetLastError();
            munmap(m, len + HUGE);
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void tbk_pin_free_(void *p) {
    if (!p) return;
    PinnedMap pm{nullptr, 0, false};
    {
        std::lock_guard<std::mutex> lk(pinned_mu);
        for (size_t i = 0; i < pinned_maps.size(); i++)
            if (pinned_maps[i].first == p) { pm = pinned_maps[i].second; pinned_maps[i] = pinned_maps.back(); pinned_maps.pop_back(); break; }
    }
    if (pm.registered) { (void)hipHostUnregister(p); munmap(pm.map, pm.len); return; }
    (void)hipHostFree(p);
}

extern "C" int tbk_classify_device(tbk_classifier *c, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                                   uint64_t total_bases, void *d_counts) {
    if (!c || (n_reads && (!d_offsets || !d_counts)) || (total_bases && !d_bases)) return fail(TBK_ERR_INVALID, "NULL argument");
    if (((uintptr_t)d_bases & 15) != 0) return fail(TBK_ERR_INVALID, "d_bases must be 16-byte aligned");
    int rc = use_device(c->device);
    if (rc) return rc;
    if (!n_reads) return TBK_OK;
    std::lock_guard<std::mutex> together(c->streams->enqueue);
    return launch_probe_timed(c, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_reads, total_bases,
                              (int32_t *)d_counts);
}

// Device-resident batch through the ticket ring: kernel on the compute stream, counts copied
// to the caller's host buffer behind it; tbk_stream_wait(ticket) returns when they are there.
extern "C" int tbk_stream_submit_device(tbk_classifier *c, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                                        uint64_t total_bases, int32_t *counts, uint64_t *ticket) {
    if (!c || !ticket || (n_reads && (!d_offsets || !counts)) || (total_bases && !d_bases)) return fail(TBK_ERR_INVALID, "NULL argument");
    if (((uintptr_t)d_bases & 15) != 0) return fail(TBK_ERR_INVALID, "d_bases must be 16-byte aligned");
    int rc = use_device(c->device);
    if (rc) return rc;
    const uint64_t tk = c->next_ticket;
    Slot &s = c->ring[tk % RING];
    if (s.busy) return fail(TBK_ERR_STATE, "all %d stream slots are in flight; call tbk_stream_wait(%llu) first", RING,
                            (unsigned long long)s.ticket);
    const bool out_pinned = n_reads == 0 || is_pinned(counts);
    rc = slot_reserve(s, 0, n_reads, false, !out_pinned);
    if (rc) return rc;
    rc = slot_reserve_device(s, 0, n_reads);
    if (rc) return rc;
    s.ticket = tk; s.n_reads = n_reads; s.user_counts = counts; s.counts_staged = !out_pinned;
    if (n_reads && total_bases) {
        std::lock_guard<std::mutex> together(c->streams->enqueue);
        rc = launch_probe_timed(c, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_reads, total_bases, s.d_counts);
        if (rc) return rc;
        // the counts travel home on a stream of their own, beside the next batch's kernels
        HIP_TRY(hipEventRecord(s.probed, c->compute));
        HIP_TRY(hipStreamWaitEvent(c->out, s.probed, 0));
        HIP_TRY(hipMemcpyAsync(out_pinned ? counts : s.h_counts, s.d_counts, n_reads * 2 * sizeof(int32_t),
                               hipMemcpyDeviceToHost, c->out));
        HIP_TRY(hipEventRecord(s.done, c->out));
    } else {
        if (n_reads) memset(counts, 0, n_reads * 2 * sizeof(int32_t));
        s.counts_staged = false;
        HIP_TRY(hipEventRecord(s.done, c->compute));
    }
    s.busy = true;
    c->streams->in_flight++;
    c->next_ticket++;
    *ticket = tk;
    return TBK_OK;
}

extern "C" int tbk_classifier_sync(tbk_classifier *c) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    int rc = use_device(c->device);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->copy));
    HIP_TRY(hipStreamSynchronize(c->compute));
    HIP_TRY(hipStreamSynchronize(c->out));
    return TBK_OK;
}

extern "C" int tbk_kernel_timing_enable(tbk_classifier *c, int on) {
    if (!c) return fail(TBK_ERR_INVALID, "classifier is NULL");
    int rc = use_device(c->device);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->compute));
    c->timing = on != 0;
    c->ev_used = 0; c->ev_slices.clear(); c->timed_launches = 0; c->timed_ms = 0.0; c->timed_single_ms = 0.0;
    return TBK_OK;
}

extern "C" int tbk_kernel_timing_read2(tbk_classifier *c, uint64_t *launches, double *total_ms, double *single_ms) {
    if (!c || !launches || !total_ms) return fail(TBK_ERR_INVALID, "NULL argument");
    int rc = use_device(c->device);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->compute));
    rc = c->fold_timing();
    if (rc) return rc;
    *launches = c->timed_launches;
    *total_ms = c->timed_ms;
    if (single_ms) *single_ms = c->timed_single_ms;
    c->timed_launches = 0; c->timed_ms = 0.0; c->timed_single_ms = 0.0;
    return TBK_OK;
}

// the most recent probe's passes (2048 window starts each) and how many of them touched more than one read
// (the multi-read and two-read kernels' share; the rest is the single-read kernel's)
extern "C" int tbk_classifier_last_passes(tbk_classifier *c, uint64_t *n_passes, uint64_t *n_multi) {
    if (!c || !n_passes || !n_multi) return fail(TBK_ERR_INVALID, "NULL argument");
    *n_passes = c->last_passes; *n_multi = 0;
    if (!c->last_passes || !c->d_pass_read) return TBK_OK;
    int rc = use_device(c->device);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->compute));
    uint32_t m[2] = {0, 0};  // passes listed for the multi-read kernel, and for the two-read kernel
    HIP_TRY(hipMemcpy(m, c->d_pass_read + 4 * c->cap_passes, sizeof m, hipMemcpyDeviceToHost));
    *n_multi = (uint64_t)m[0] + m[1];
    return TBK_OK;
}

extern "C" int tbk_kernel_timing_read(tbk_classifier *c, uint64_t *launches, double *total_ms) {
    return tbk_kernel_timing_read2(c, launches, total_ms, nullptr);
}

// ---- single read (compat with kmers.count_kmers_in_read) ----------------------------------
static std::mutex g_cache_mu;
static tbk_classifier *g_cached = nullptr;
static const tbk_table *g_cached_a = nullptr, *g_cached_b = nullptr;

static void drop_cached_classifier(const tbk_table *t) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (g_cached && (g_cached_a == t || g_cached_b == t)) {
        tbk_classifier_destroy(g_cached);
        g_cached = nullptr; g_cached_a = g_cached_b = nullptr;
    }
}

// One read, one launch (tbk_classifier: small_*).  Reads of up to SMALL_READ_MAX bases; longer ones take the batch path.
static constexpr uint64_t SMALL_READ_MAX = (uint64_t)8 << 20;

static int count_small(tbk_classifier *c, const char *read, uint64_t len, int *count_a, int *count_b) {
    int rc = use_device(c->device);
    if (rc) return rc;
    if (len > c->small_cap) {
        c->small_free();
        const uint64_t cap = std::max<uint64_t>((uint64_t)1 << 20, len + len / 2);
        const uint64_t pass_cap = (tbk_probe_passes(cap) + 2) & ~1ull;
        hipError_t e = hipHostMalloc((void **)&c->small_h, 64 + cap + 4096, hipHostMallocPortable | hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostMalloc((void **)&c->small_h_counts, 64, hipHostMallocPortable);
        if (e == hipSuccess) e = hipMalloc((void **)&c->small_d_counts, 64);
        if (e == hipSuccess) e = hipMemset(c->small_d_counts, 0, 64);
        if (e == hipSuccess) e = hipMalloc((void **)&c->small_scratch, (4 * pass_cap + 16) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(c->small_scratch, 0, (4 * pass_cap + 16) * sizeof(uint32_t));  // every pass starts in read 0; no multi-read, no two-read pass
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { (void)hipGetLastError(); c->small_free(); return TBK_ERR_STATE; }
        memset(c->small_h, 0, 64 + cap + 4096);
        c->small_cap = cap; c->small_pass_cap = pass_cap;
        c->small_last[0] = c->small_last[1] = 0;
    }
    uint64_t *offsets = (uint64_t *)c->small_h;
    uint8_t *bases = c->small_h + 64;
    offsets[0] = 0; offsets[1] = len;
    memcpy(bases, read, len);
    std::lock_guard<std::mutex> together(c->streams->enqueue);
    hipError_t e = tbk_launch_probe_range(bases, nullptr, nullptr, offsets, 1, len, c->pair(), c->k, c->small_d_counts, c->small_scratch, c->small_pass_cap, 0,
                                          tbk_probe_passes(len), c->max_blocks, 0, nullptr, c->compute, 1);
    if (e == hipSuccess) e = hipMemcpyAsync(c->small_h_counts, c->small_d_counts, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->compute);
    if (e == hipSuccess) e = hipStreamSynchronize(c->compute);
    if (e != hipSuccess) {
        c->small_free();  // (the running counters are no longer known)
        return fail(TBK_ERR_HIP, "tbk_count_kmers_in_read: %s", hipGetErrorString(e));
    }
    const uint32_t now_a = c->small_h_counts[0], now_b = c->small_h_counts[1];
    *count_a = (int)(uint32_t)(now_a - c->small_last[0]);
    *count_b = (int)(uint32_t)(now_b - c->small_last[1]);
    c->small_last[0] = now_a; c->small_last[1] = now_b;
    return TBK_OK;
}

extern "C" int tbk_count_kmers_in_read(const char *read, int64_t len, const tbk_table *a, const tbk_table *b,
                                       int *count_a, int *count_b) {
    if (!read || !a || !b || !count_a || !count_b) return fail(TBK_ERR_INVALID, "NULL argument");
    if (len < 0) len = (int64_t)strlen(read);
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (!g_cached || g_cached_a != a || g_cached_b != b) {
        if (g_cached) { tbk_classifier_destroy(g_cached); g_cached = nullptr; }
        int rc = tbk_classifier_create(a, b, &g_cached);
        if (rc) return rc;
        g_cached_a = a; g_cached_b = b;
    }
    return tbk_classifier_count_read(g_cached, read, len, count_a, count_b);
}

// The same on a classifier of the caller's (one caller at a time per classifier, like every entry point that takes one).
extern "C" int tbk_classifier_count_read(tbk_classifier *c, const char *read, int64_t len, int *count_a, int *count_b) {
    if (!c || !read || !count_a || !count_b) return fail(TBK_ERR_INVALID, "NULL argument");
    if (len < 0) len = (int64_t)strlen(read);
    if (len < c->k) { *count_a = *count_b = 0; return TBK_OK; }  // c/kmers.c:287: no window
    if ((uint64_t)len <= SMALL_READ_MAX) {
        const int rc = count_small(c, read, (uint64_t)len, count_a, count_b);
        if (rc != TBK_ERR_STATE) return rc;  // (TBK_ERR_STATE: the fast path could not be set up - the batch path below)
    }
    uint64_t offsets[2] = {0, (uint64_t)len};
    int32_t counts[2] = {0, 0};
    int rc = tbk_classify_batch(c, (const uint8_t *)read, offsets, 1, counts);
    if (rc) return rc;
    *count_a = counts[0];
    *count_b = counts[1];
    return TBK_OK;
}

// ---- scoring (host) ------------------------------------------------------------------------
extern "C" int tbk_score_and_bin(const int32_t *counts, uint64_t n_reads, uint64_t num_a, uint64_t num_b,
                                 double *score_a, double *score_b, char *bins) {
    if (n_reads && (!counts || !score_a || !score_b || !bins)) return fail(TBK_ERR_INVALID, "NULL argument");
    if (!num_a || !num_b) return fail(TBK_ERR_INVALID, "a k-mer list is empty (the reference divides by zero here)");
    // classify_by_kmers.py:72-76: 1.0 * max / n ; :104-105 count * factor ; :107-115 strict >
    const uint64_t mx = num_a > num_b ? num_a : num_b;
    const double fa = 1.0 * (double)mx / (double)num_a;
    const double fb = 1.0 * (double)mx / (double)num_b;
    auto work = [=](uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; r++) {
            const double sa = (double)counts[2 * r] * fa, sb = (double)counts[2 * r + 1] * fb;
            score_a[r] = sa;
            score_b[r] = sb;
            bins[r] = sa > sb ? 'A' : (sb > sa ? 'B' : 'U');
        }
    };
    // batches of millions of short reads: split over a few host threads (each read is independent)
    const unsigned hw = (unsigned)tbk_host_threads();
    const uint64_t n_thr = std::min<uint64_t>(std::min<uint64_t>(hw ? hw : 1, 16), n_reads >> 18);
    if (n_thr <= 1) {
        work(0, n_reads);
    } else {
        std::vector<std::thread> pool;
        for (uint64_t t = 0; t < n_thr; t++) pool.emplace_back(work, n_reads * t / n_thr, n_reads * (t + 1) / n_thr);
        for (std::thread &th : pool) th.join();
    }
    return TBK_OK;
}

// ---- device memory helpers ------------------------------------------------------------------
extern "C" int tbk_device_alloc(int device, size_t bytes, void **d_ptr) {
    if (!d_ptr) return fail(TBK_ERR_INVALID, "d_ptr is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 16));
    return TBK_OK;
}
extern "C" int tbk_device_free(int device, void *d_ptr) {
    int rc = use_device(device);
    if (rc) return rc;
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return TBK_OK;
}
extern "C" int tbk_memcpy_h2d(int device, void *d_dst, const void *h_src, size_t bytes) {
    int rc = use_device(device);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return TBK_OK;
}
extern "C" int tbk_memcpy_d2h(int device, void *h_dst, const void *d_src, size_t bytes) {
    int rc = use_device(device);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return TBK_OK;
}
extern "C" int tbk_device_sync(int device) {
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return TBK_OK;
}
extern "C" int tbk_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    int rc = use_device(device);
    if (rc) return rc;
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return TBK_OK;
}

// ---- synthetic workload -----------------------------------------------------------------------
extern "C" int tbk_synth_keys_device(int device, uint64_t seed, uint64_t first, uint64_t n, int k, void *d_keys) {
    if (k < 3 || k > 32) return fail(TBK_ERR_INVALID, "synthetic keys need 3 <= k <= 32");
    if (k < 32 && first + n > (1ull << (2 * (k - 2)))) return fail(TBK_ERR_INVALID, "key index exceeds 4^(k-2)");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(tbk_launch_synth_keys(seed, first, n, k, (uint64_t *)d_keys, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return TBK_OK;
}

extern "C" int tbk_synth_keys_host(uint64_t seed, uint64_t first, uint64_t n, int k, uint64_t *keys) {
    if (k < 3 || k > 32) return fail(TBK_ERR_INVALID, "synthetic keys need 3 <= k <= 32");
    if (n && !keys) return fail(TBK_ERR_INVALID, "keys is NULL");
    for (uint64_t i = 0; i < n; i++) keys[i] = tbk_synth_key(seed, first + i, k);
    return TBK_OK;
}

extern "C" int tbk_synth_reads_device(int device, uint64_t read_seed, uint64_t first_read, uint64_t n_reads,
                                      uint32_t read_len, uint64_t key_seed, uint64_t n_a, uint64_t n_b, int k,
                                      int plant_major, int plant_minor, void *d_bases, void *d_offsets) {
    if (k < 3 || k > 32 || plant_major < 0 || plant_minor < 0) return fail(TBK_ERR_INVALID, "bad synth parameters");
    if (((uintptr_t)d_bases & 15) != 0) return fail(TBK_ERR_INVALID, "d_bases must be 16-byte aligned");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(tbk_launch_synth_reads(read_seed, first_read, n_reads, read_len, key_seed, n_a, n_b, k, plant_major,
                                   plant_minor, (uint8_t *)d_bases, (uint64_t *)d_offsets, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return TBK_OK;
}

extern "C" hipError_t tbk_launch_synth_hap_keys(uint64_t, uint64_t, uint32_t, int, uint64_t *, uint64_t *, uint64_t,
                                                unsigned long long *, hipStream_t);
extern "C" hipError_t tbk_launch_synth_hap_reads(uint64_t, uint64_t, uint32_t, uint64_t, uint64_t, uint64_t, uint32_t, uint32_t,
                                                 uint8_t *, uint64_t *, hipStream_t);

extern "C" int tbk_synth_hap_keys_device(int device, uint64_t seed, uint64_t genome_len, uint32_t snp_per_2p24, int k,
                                         void *d_keys_a, void *d_keys_b, uint64_t capacity, uint64_t *n_keys) {
    if (k < 3 || k > 32 || genome_len < (uint64_t)k || !d_keys_a || !d_keys_b || !n_keys) return fail(TBK_ERR_INVALID, "bad synth parameters");
    int rc = use_device(device);
    if (rc) return rc;
    unsigned long long *d_n = nullptr, n = 0;
    HIP_TRY(hipMalloc((void **)&d_n, sizeof n));
    hipError_t e = hipMemset(d_n, 0, sizeof n);
    if (e == hipSuccess) e = tbk_launch_synth_hap_keys(seed, genome_len, snp_per_2p24, k, (uint64_t *)d_keys_a, (uint64_t *)d_keys_b, capacity, d_n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&n, d_n, sizeof n, hipMemcpyDeviceToHost);
    (void)hipFree(d_n);
    if (e != hipSuccess) return fail(TBK_ERR_HIP, "tbk_synth_hap_keys_device: %s", hipGetErrorString(e));
    *n_keys = n;  // may exceed capacity: only the first `capacity` were written
    return TBK_OK;
}

extern "C" int tbk_synth_hap_reads_device(int device, uint64_t seed, uint64_t genome_len, uint32_t snp_per_2p24,
                                          uint64_t read_seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                                          uint32_t err_per_2p24, void *d_bases, void *d_offsets) {
    if (genome_len < read_len || !read_len) return fail(TBK_ERR_INVALID, "bad synth parameters");
    if (((uintptr_t)d_bases & 15) != 0) return fail(TBK_ERR_INVALID, "d_bases must be 16-byte aligned");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(tbk_launch_synth_hap_reads(seed, genome_len, snp_per_2p24, read_seed, first_read, n_reads, read_len, err_per_2p24,
                                       (uint8_t *)d_bases, (uint64_t *)d_offsets, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return TBK_OK;
}

// ---- calibration --------------------------------------------------------------------------------
extern "C" int tbk_calib_gather(int device, uint64_t footprint, int line_bytes, int lanes_per_line, int inflight,
                                uint64_t n_lines, int reps, double *lines_per_sec, double *ms_per_rep) {
    int rc = use_device(device);
    if (rc) return rc;
    footprint &= ~(uint64_t)255;
    if (footprint < (1u << 20)) return fail(TBK_ERR_INVALID, "footprint too small");
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc(&buf, footprint));
    hipError_t e = hipMalloc((void **)&sink, 16);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uint64_t done = 0;
    float ms = 0;
    if (e == hipSuccess) e = tbk_launch_fill(buf, footprint, 1, nullptr);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = tbk_launch_gather(buf, footprint, line_bytes, lanes_per_line, inflight, n_lines, 7, sink, &done, nullptr);  // warm-up
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps && e == hipSuccess; r++)
        e = tbk_launch_gather(buf, footprint, line_bytes, lanes_per_line, inflight, n_lines, 11 + r, sink, &done, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? TBK_ERR_INVALID : TBK_ERR_HIP, "calib_gather: %s", hipGetErrorString(e));
    if (ms_per_rep) *ms_per_rep = ms / reps;
    if (lines_per_sec) *lines_per_sec = (double)done * reps / (ms * 1e-3);
    return TBK_OK;
}

extern "C" hipError_t tbk_launch_gather_pairs(const void *, uint64_t, int, int, uint64_t, uint64_t, uint32_t *, uint64_t *, hipStream_t);
extern "C" hipError_t tbk_launch_stream_nt(const void *, uint64_t, int, unsigned, uint32_t *, hipStream_t);

// The gather in the entry kernels' shape (one-wave blocks, two lanes x 16 bytes per line, `inflight` lines per pair before any is
// used, `waves_per_simd` resident waves) and a tuned streaming read (`unroll` x 16 bytes per thread, non-temporal, `blocks`
// blocks of 256): the ceilings bench.py prices the probe kernel's line rate and traffic against (tools/calib_ceilings.py sweeps them).
extern "C" int tbk_calib_gather_pairs(int device, uint64_t footprint, int inflight, int waves_per_simd, uint64_t n_lines, int reps, double *lines_per_sec) {
    int rc = use_device(device);
    if (rc) return rc;
    footprint &= ~(uint64_t)255;
    if (footprint < (1u << 20) || reps < 1) return fail(TBK_ERR_INVALID, "footprint too small");
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc(&buf, footprint));
    hipError_t e = hipMalloc((void **)&sink, 16);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uint64_t done = 0;
    float ms = 0;
    if (e == hipSuccess) e = tbk_launch_fill(buf, footprint, 1, nullptr);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = tbk_launch_gather_pairs(buf, footprint, inflight, waves_per_simd, n_lines, 7, sink, &done, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps && e == hipSuccess; r++) e = tbk_launch_gather_pairs(buf, footprint, inflight, waves_per_simd, n_lines, 11 + r, sink, &done, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? TBK_ERR_INVALID : TBK_ERR_HIP, "calib_gather_pairs: %s", hipGetErrorString(e));
    if (lines_per_sec) *lines_per_sec = (double)done * reps / (ms * 1e-3);
    return TBK_OK;
}

extern "C" int tbk_calib_stream_nt(int device, uint64_t footprint, int unroll, int blocks, int reps, double *bytes_per_sec) {
    int rc = use_device(device);
    if (rc) return rc;
    footprint &= ~(uint64_t)255;
    if (footprint < (1u << 20) || reps < 1 || blocks < 1) return fail(TBK_ERR_INVALID, "bad calibration parameters");
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc(&buf, footprint));
    hipError_t e = hipMalloc((void **)&sink, 16);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    if (e == hipSuccess) e = tbk_launch_fill(buf, footprint, 1, nullptr);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = tbk_launch_stream_nt(buf, footprint, unroll, (unsigned)blocks, sink, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps && e == hipSuccess; r++) e = tbk_launch_stream_nt(buf, footprint, unroll, (unsigned)blocks, sink, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? TBK_ERR_INVALID : TBK_ERR_HIP, "calib_stream_nt: %s", hipGetErrorString(e));
    if (bytes_per_sec) *bytes_per_sec = (double)footprint * reps / (ms * 1e-3);
    return TBK_OK;
}

extern "C" hipError_t tbk_launch_atomics(void *, uint64_t, uint32_t, uint32_t, uint64_t, unsigned, hipStream_t);

extern "C" hipError_t tbk_launch_atomics64(void *, uint64_t, uint32_t, uint32_t, uint64_t, unsigned, hipStream_t);
static int calib_atomics(int device, uint64_t footprint, int run, int reps, double *atomics_per_sec, bool wide);
extern "C" int tbk_calib_atomics(int device, uint64_t footprint, int run, int reps, double *atomics_per_sec) {
    return calib_atomics(device, footprint, run, reps, atomics_per_sec, false);
}
// the same with the counting kernel's own instruction: 64-bit adds (two counters at once), `run` (1..4) per line
extern "C" int tbk_calib_atomics64(int device, uint64_t footprint, int run, int reps, double *atomics_per_sec) {
    if (run > 4) return fail(TBK_ERR_INVALID, "run must be 1..4 (a line holds four 64-bit counter words)");
    return calib_atomics(device, footprint, run, reps, atomics_per_sec, true);
}
static int calib_atomics(int device, uint64_t footprint, int run, int reps, double *atomics_per_sec, bool wide) {
    int rc = use_device(device);
    if (rc) return rc;
    if (run < 1 || run > 32) return fail(TBK_ERR_INVALID, "run must be 1..32");
    footprint &= ~(uint64_t)255;
    if (footprint < (1u << 20)) return fail(TBK_ERR_INVALID, "footprint too small");
    void *buf = nullptr;
    HIP_TRY(hipMalloc(&buf, footprint));
    const unsigned blocks = 16384;
    const uint32_t iters = 256;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    hipError_t e = hipMemset(buf, 0, footprint);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    auto launch = wide ? tbk_launch_atomics64 : tbk_launch_atomics;
    if (e == hipSuccess) e = launch(buf, footprint, iters, (uint32_t)run, 1, blocks, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps && e == hipSuccess; r++) e = launch(buf, footprint, iters, (uint32_t)run, 2 + r, blocks, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(TBK_ERR_HIP, "calib_atomics: %s", hipGetErrorString(e));
    if (atomics_per_sec) *atomics_per_sec = (double)blocks * 256 * iters * reps / (ms * 1e-3);
    return TBK_OK;
}

extern "C" int tbk_calib_stream(int device, uint64_t footprint, int reps, double *bytes_per_sec) {
    int rc = use_device(device);
    if (rc) return rc;
    footprint &= ~(uint64_t)255;
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc(&buf, footprint));
    hipError_t e = hipMalloc((void **)&sink, 16);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    if (e == hipSuccess) e = tbk_launch_fill(buf, footprint, 1, nullptr);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = tbk_launch_stream(buf, footprint, sink, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps && e == hipSuccess; r++) e = tbk_launch_stream(buf, footprint, sink, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return fail(TBK_ERR_HIP, "calib_stream: %s", hipGetErrorString(e));
    if (bytes_per_sec) *bytes_per_sec = (double)footprint * reps / (ms * 1e-3);
    return TBK_OK;
}
