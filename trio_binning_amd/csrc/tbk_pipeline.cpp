// tbk_pipeline.cpp — one handle over several devices: a feeder thread + stream ring per device, one ordered queue.
//
// tables replicated in every GPU's HBM, reads dealt in batches to 1/2/4/8 devices, "one host
// feeder thread + pinned ring + 2-3 streams per device", results taken back in input order, no collective.
// The reference's loop is one read at a time on one core (classify_by_kmers.py:99-117); here the caller
// (a reader thread, the native classify loop of tbk_run.cpp, bench.py) hands batches to
// tbk_pipeline_submit, which only queues them, and a feeder takes the next batch as soon as its device's
// ring has a free slot: whatever is left to do on the host for a batch - packing an ASCII batch into the
// transfer format (its share of the host threads), staging pageable arrays, the copies' and kernels' launches
// - happens on that device's feeder, beside the other devices' feeders and off the caller's thread.
// tbk_pipeline_wait(ticket) returns when that batch's counts are in the caller's array, whichever device
// computed them; a device may appear several times in the list (several rings on one GPU: how a one-GPU box
// tests the dealing).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/tbk.h"

extern "C" void tbk_set_error_(int code, const char *msg);
extern "C" int tbk_classifier_set_pack_threads_(tbk_classifier *c, int threads);

namespace {

struct Job {
    uint64_t ticket = 0;
    const uint8_t *bases = nullptr;
    const uint32_t *codes = nullptr, *exc_chunk = nullptr;
    const uint16_t *exc_mask = nullptr;
    uint64_t n_exc = 0;
    const uint64_t *offsets = nullptr;
    uint64_t n_reads = 0;
    int32_t *counts = nullptr;
    bool packed = false;
    // set by the feeder
    bool done = false;
    int rc = 0;
    int device_slot = -1;
    std::string err;
};

}  // namespace

// what a feeder drives: a classifier's stream ring, or (tests, no GPU) a pair of callbacks with the same contract
struct Ring {
    tbk_classifier *c = nullptr;
    tbk_pipeline_test_submit_fn t_submit = nullptr;
    tbk_pipeline_test_wait_fn t_wait = nullptr;
    void *user = nullptr;
    int slot = 0, depth = 1;
    int submit(const Job *j, uint64_t *tk) const {
        if (!c) return t_submit(user, slot, j->bases, j->offsets, j->n_reads, j->counts, tk);
        if (j->packed) return tbk_stream_submit_packed(c, j->codes, j->exc_chunk, j->exc_mask, j->n_exc, j->offsets, j->n_reads, j->counts, tk);
        return tbk_stream_submit(c, j->bases, j->offsets, j->n_reads, j->counts, tk);
    }
    int wait(uint64_t tk) const { return c ? tbk_stream_wait(c, tk) : t_wait(user, slot, tk); }
    // 1: that batch is complete (wait will not block), 0: not yet, -1: cannot tell without waiting (test rings)
    int done(uint64_t tk) const { return c ? tbk_stream_query(c, tk) : -1; }
};

struct tbk_pipeline {
    std::vector<tbk_classifier *> cls;
    std::vector<Ring> rings;
    std::vector<std::thread> feeders;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job *> queue;                       // submitted, not yet taken by a feeder (submission order)
    std::unordered_map<uint64_t, Job *> jobs;      // every job not yet waited for
    uint64_t next_ticket = 1;
    uint64_t in_system = 0;                        // submitted and not yet waited for
    int depth = 0;
    bool stop = false;
    std::vector<uint64_t> batches_by_slot;         // how many batches each ring took (stats)
    std::vector<int> numa_node, numa_cpus;         // per ring: the device's NUMA node (-1 unknown) and the CPUs its feeder is bound to (0: not bound)
    int pack_share = 1;                            // host threads a feeder packs an ASCII batch with
};

extern "C" int tbk_numa_bind_to_device(int device, int *node_out, int *cpus_out);
extern "C" int tbk_host_threads_per_feeder_(int n_devices);

static int pfail(int code, const char *msg) {
    tbk_set_error_(code, msg);
    return code;
}

static void feeder_loop(tbk_pipeline *p, int slot) {
    const Ring &rg = p->rings[(size_t)slot];
    const int ring = rg.depth;
    // this thread packs and stages its device's batches and allocates the ring's pinned buffers (at its first submit):
    // it runs on the CPUs of the socket the device hangs off (tbk_host.cpp "NUMA placement"; unknown node: anywhere)
    if (rg.c) {
        (void)tbk_numa_bind_to_device(tbk_classifier_device(rg.c), &p->numa_node[(size_t)slot], &p->numa_cpus[(size_t)slot]);
        // the packers this feeder starts inherit its mask: no more of them than the CPUs it is bound to
        const int cpus = p->numa_cpus[(size_t)slot];
        if (cpus > 0 && cpus < p->pack_share) (void)tbk_classifier_set_pack_threads_(rg.c, cpus);
    }
    std::deque<std::pair<uint64_t, Job *>> flying;  // (ring ticket, job), oldest first
    auto finish_oldest = [&]() {
        auto [tk, job] = flying.front();
        flying.pop_front();
        const int rc = rg.wait(tk);
        std::lock_guard<std::mutex> lk(p->mu);
        job->rc = rc;
        if (rc) job->err = tbk_last_error();
        job->done = true;
        p->cv_done.notify_all();
    };
    for (;;) {
        Job *job = nullptr;
        bool collect = false;  // take the oldest batch in flight off the ring
        {
            std::unique_lock<std::mutex> lk(p->mu);
            if ((int)flying.size() >= ring) {
                collect = true;  // the ring is full: nothing to do but wait for its oldest batch
            } else if (flying.empty()) {
                p->cv_work.wait(lk, [&] { return p->stop || !p->queue.empty(); });
                if (p->queue.empty()) return;  // stop, nothing queued, nothing in flight
            } else {
                // Batches in flight and room in the ring: whichever comes first - a new batch to submit (its copy
                // must start NOW, beside the kernel that is running, not when that kernel ends) or the oldest
                // batch completing.  The device is polled, the queue is waited on in short naps.
                // The naps back off from 100 us to 1 ms (a new batch still wakes the feeder at once: cv_work): several
                // rings on one device polled at 10 kHz each were runtime calls taken from the reader's and the writer's CPUs.
                // A ring that cannot be polled (test rings) is collected from right away.
                int nap_us = 100;
                while (p->queue.empty() && !p->stop) {
                    lk.unlock();  // (the query is a runtime call: not under the queue's lock)
                    const int d = rg.done(flying.front().first);
                    lk.lock();
                    if (d == 1 || d < 0) { collect = true; break; }
                    if (!p->queue.empty() || p->stop) break;
                    p->cv_work.wait_for(lk, std::chrono::microseconds(nap_us));
                    nap_us = std::min(1000, nap_us * 2);
                }
                if (p->queue.empty() && p->stop) collect = true;
            }
            if (!collect && !p->queue.empty()) {
                job = p->queue.front();
                p->queue.pop_front();
                job->device_slot = slot;
                p->batches_by_slot[(size_t)slot]++;
            }
        }
        if (!job) {
            if (!flying.empty()) finish_oldest();
            continue;
        }
        uint64_t tk = 0;
        const int rc = rg.submit(job, &tk);
        if (rc) {
            std::lock_guard<std::mutex> lk(p->mu);
            job->rc = rc;
            job->err = tbk_last_error();
            job->done = true;
            p->cv_done.notify_all();
            continue;
        }
        flying.emplace_back(tk, job);
    }
}

extern "C" int tbk_pipeline_create(const tbk_table *a, const tbk_table *b, const int *devices, int n_devices, tbk_pipeline **out) {
    return tbk_pipeline_create_opts(a, b, devices, n_devices, nullptr, out);
}

extern "C" int tbk_pipeline_create_opts(const tbk_table *a, const tbk_table *b, const int *devices, int n_devices, const tbk_options *options, tbk_pipeline **out) {
    if (!out) return pfail(TBK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return pfail(TBK_ERR_INVALID, "devices is NULL or n_devices outside 1..64");
    std::vector<tbk_classifier *> cls((size_t)n_devices, nullptr);
    int rc = tbk_classifier_create_multi_opts(a, b, devices, n_devices, options, cls.data());
    if (rc) return rc;
    tbk_pipeline *p = new tbk_pipeline();
    p->cls = cls;
    p->batches_by_slot.assign((size_t)n_devices, 0);
    p->numa_node.assign((size_t)n_devices, -1); p->numa_cpus.assign((size_t)n_devices, 0);
    // an ASCII batch is packed by its feeder with this share of the host threads
    const int share = tbk_host_threads_per_feeder_(n_devices);
    p->pack_share = share;
    for (int i = 0; i < n_devices; i++) {
        tbk_classifier *c = p->cls[(size_t)i];
        (void)tbk_classifier_set_pack_threads_(c, share);
        Ring rg;
        rg.c = c; rg.slot = i; rg.depth = tbk_stream_depth(c);
        p->rings.push_back(rg);
        p->depth += rg.depth;
    }
    for (int i = 0; i < n_devices; i++) p->feeders.emplace_back(feeder_loop, p, i);
    *out = p;
    return TBK_OK;
}

// Testing hook (no GPU needed): the same queue and feeder threads over `n_rings` rings whose submit / wait are
// the caller's callbacks (tests/test_multi_cpu.py drives them with the oracle and finishes batches out of step).
extern "C" int tbk_pipeline_create_test_(int n_rings, int ring_depth, tbk_pipeline_test_submit_fn submit, tbk_pipeline_test_wait_fn wait, void *user,
                                         tbk_pipeline **out) {
    if (!out) return pfail(TBK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_rings < 1 || n_rings > 64 || ring_depth < 1 || !submit || !wait) return pfail(TBK_ERR_INVALID, "bad test pipeline parameters");
    tbk_pipeline *p = new tbk_pipeline();
    p->batches_by_slot.assign((size_t)n_rings, 0);
    p->numa_node.assign((size_t)n_rings, -1); p->numa_cpus.assign((size_t)n_rings, 0);
    for (int i = 0; i < n_rings; i++) {
        Ring rg;
        rg.t_submit = submit; rg.t_wait = wait; rg.user = user; rg.slot = i; rg.depth = ring_depth;
        p->rings.push_back(rg);
        p->depth += ring_depth;
    }
    for (int i = 0; i < n_rings; i++) p->feeders.emplace_back(feeder_loop, p, i);
    *out = p;
    return TBK_OK;
}

extern "C" int tbk_pipeline_depth(const tbk_pipeline *p) { return p ? p->depth : 0; }
// (library-internal) 1 when batches may be submitted in the packed transfer format: real rings whose transfer is packed
extern "C" int tbk_pipeline_takes_packed_(const tbk_pipeline *p) {
    if (!p || p->cls.empty()) return 0;
    for (tbk_classifier *c : p->cls) if (tbk_classifier_transfer(c) != 1) return 0;
    return 1;
}
extern "C" int tbk_pipeline_devices(const tbk_pipeline *p) { return p ? (int)p->rings.size() : 0; }
extern "C" int tbk_pipeline_numa(const tbk_pipeline *p, int slot, int *node, int *cpus) {
    if (!p || slot < 0 || slot >= (int)p->rings.size()) return pfail(TBK_ERR_INVALID, "pipeline is NULL or slot out of range");
    if (node) *node = p->numa_node[(size_t)slot];
    if (cpus) *cpus = p->numa_cpus[(size_t)slot];
    return TBK_OK;
}
extern "C" tbk_classifier *tbk_pipeline_classifier(tbk_pipeline *p, int slot) {
    return p && slot >= 0 && slot < (int)p->cls.size() ? p->cls[(size_t)slot] : nullptr;
}

static int enqueue(tbk_pipeline *p, Job *job, uint64_t *ticket) {
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->stop) { delete job; return pfail(TBK_ERR_STATE, "pipeline is shutting down"); }
    // as many batches in the system as the rings hold plus one waiting per device: a caller that runs
    // further ahead must wait for a ticket first (its buffers are in use until then anyway)
    if (p->in_system >= (uint64_t)p->depth + p->rings.size()) {
        delete job;
        return pfail(TBK_ERR_STATE, "too many batches in flight; call tbk_pipeline_wait first");
    }
    job->ticket = p->next_ticket++;
    p->jobs[job->ticket] = job;
    p->queue.push_back(job);
    p->in_system++;
    *ticket = job->ticket;
    lk.unlock();
    p->cv_work.notify_one();
    return TBK_OK;
}

extern "C" int tbk_pipeline_submit(tbk_pipeline *p, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int32_t *counts,
                                   uint64_t *ticket) {
    if (!p || !offsets || !ticket || (n_reads && !counts)) return pfail(TBK_ERR_INVALID, "NULL argument");
    if (offsets[n_reads] && !bases) return pfail(TBK_ERR_INVALID, "bases is NULL");
    Job *job = new Job();
    job->bases = bases; job->offsets = offsets; job->n_reads = n_reads; job->counts = counts;
    return enqueue(p, job, ticket);
}

extern "C" int tbk_pipeline_submit_packed(tbk_pipeline *p, const uint32_t *codes, const uint32_t *exc_chunk, const uint16_t *exc_mask,
                                          uint64_t n_exc, const uint64_t *offsets, uint64_t n_reads, int32_t *counts, uint64_t *ticket) {
    if (!p || !offsets || !ticket || (n_reads && !counts)) return pfail(TBK_ERR_INVALID, "NULL argument");
    if (offsets[n_reads] && !codes) return pfail(TBK_ERR_INVALID, "codes is NULL");
    if (p->cls.empty()) return pfail(TBK_ERR_INVALID, "a test pipeline takes ASCII batches only");
    Job *job = new Job();
    job->packed = true;
    job->codes = codes; job->exc_chunk = exc_chunk; job->exc_mask = exc_mask; job->n_exc = n_exc;
    job->offsets = offsets; job->n_reads = n_reads; job->counts = counts;
    return enqueue(p, job, ticket);
}

extern "C" int tbk_pipeline_wait(tbk_pipeline *p, uint64_t ticket, int *device_slot) {
    if (!p) return pfail(TBK_ERR_INVALID, "pipeline is NULL");
    std::unique_lock<std::mutex> lk(p->mu);
    auto it = p->jobs.find(ticket);
    if (it == p->jobs.end()) return pfail(TBK_ERR_STATE, "ticket is not in flight");
    Job *job = it->second;
    p->cv_done.wait(lk, [&] { return job->done; });
    p->jobs.erase(it);
    p->in_system--;
    lk.unlock();
    const int rc = job->rc;
    if (device_slot) *device_slot = job->device_slot;
    if (rc) tbk_set_error_(rc, job->err.c_str());
    delete job;
    return rc;
}

extern "C" int tbk_pipeline_batches(const tbk_pipeline *p, uint64_t *per_slot, int n) {
    if (!p || !per_slot) return pfail(TBK_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(const_cast<tbk_pipeline *>(p)->mu);
    for (int i = 0; i < n && i < (int)p->batches_by_slot.size(); i++) per_slot[i] = p->batches_by_slot[(size_t)i];
    return TBK_OK;
}

extern "C" void tbk_pipeline_destroy(tbk_pipeline *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_work.notify_all();
    for (std::thread &t : p->feeders) t.join();
    for (auto &kv : p->jobs) delete kv.second;
    // a device listed twice shares nothing but the table's source: every classifier is its own object
    for (tbk_classifier *c : p->cls) tbk_classifier_destroy(c);
    delete p;
}
