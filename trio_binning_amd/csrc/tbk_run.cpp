// tbk_run.cpp — the read / classify / write loop of classify-by-kmers as native threads.
//
// Replaces the body of classify_by_kmers.main (classify_by_kmers.py:80-117): for every read of the input,
// count_kmers_in_read, the two scores, the bin, the record written to that bin and a TSV line on stdout.
// The reference does that one read at a time in Python; here the stages run side by side on batches:
//     reader thread    tbk_fastx_next: the next batch of records into pinned memory, its bases packed
//                      into the transfer format on the way (tbk_fastx_set_packing);
//     calling thread   tbk_pipeline_submit_packed: up to `depth` batches in flight on the device(s);
//     collector thread tbk_pipeline_wait: the batches taken back in input order, as soon as they are classified;
//     writer thread    tbk_score_and_bin, tbk_bin_writer_write, tbk_format_tsv -> tsv_fd, in input order.
// Batches circulate through a fixed set of buffers (depth + 3), so memory is bounded and nothing is
// allocated per batch.  Output bytes are those of the Python mirror's loop, which are those of the
// reference (tests/test_gpu_cli.py runs both and the reference's recorded output).
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tbk.h"

extern "C" void tbk_set_error_(int code, const char *msg);
extern "C" int tbk_pipeline_takes_packed_(const tbk_pipeline *p);

namespace {

using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t) { return std::chrono::duration<double>(Clock::now() - t).count(); }

struct Item {
    tbk_fastx_batch *batch = nullptr;
    int32_t *counts = nullptr;  // [cap_reads][2], pinned when a device is there
    bool counts_pinned = false;
    size_t cap_reads = 0;
    uint64_t n_reads = 0, n_bases = 0;
};

template <class T>
struct Chan {  // a small blocking queue; close() wakes everybody
    std::mutex mu;
    std::condition_variable cv;
    std::deque<T> q;
    bool closed = false;
    void put(T v) { { std::lock_guard<std::mutex> lk(mu); q.push_back(v); } cv.notify_one(); }
    bool get(T &v) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return closed || !q.empty(); });
        if (q.empty()) return false;
        v = q.front(); q.pop_front();
        return true;
    }
    void close() { { std::lock_guard<std::mutex> lk(mu); closed = true; } cv.notify_all(); }
};

struct Failure {
    std::mutex mu;
    int rc = 0;
    std::string msg;
    void set(int code, const char *m) { std::lock_guard<std::mutex> lk(mu); if (!rc) { rc = code; msg = m ? m : ""; } }
    bool any() { std::lock_guard<std::mutex> lk(mu); return rc != 0; }
};

bool write_fd(int fd, const char *p, size_t n) {
    while (n) {
        const ssize_t w = ::write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; n -= (size_t)w;
    }
    return true;
}

}  // namespace

extern "C" int tbk_classify_file(tbk_pipeline *p, const char *reads_path, uint64_t num_kmers_a, uint64_t num_kmers_b, const char *out_a,
                                 const char *out_b, const char *out_u, int gzip_output, int gzip_level, int tsv_fd, uint64_t batch_bases,
                                 uint64_t batch_reads, tbk_run_stats *stats) {
    if (!p || !reads_path || !out_a || !out_b || !out_u) { tbk_set_error_(TBK_ERR_INVALID, "NULL argument"); return TBK_ERR_INVALID; }
    const auto t_start = Clock::now();
    tbk_run_stats st;
    memset(&st, 0, sizeof st);
    if (!batch_bases) batch_bases = (uint64_t)64 << 20;
    tbk_fastx_reader *reader = nullptr;
    int rc = tbk_fastx_open(reads_path, &reader);
    if (rc) return rc;
    (void)tbk_fastx_set_packing(reader, tbk_pipeline_takes_packed_(p));  // (TBK_PACKED_H2D=0 and test pipelines take ASCII)
    // records of a mapped plain FASTQ stay in the mapping: packed from there, written to their bins from there
    // (TBK_BORROW=0: copied into the batch's arrays first, as the Python-level reader does)
    const char *borrow_env = getenv("TBK_BORROW");
    (void)tbk_fastx_set_borrowing(reader, tbk_pipeline_takes_packed_(p) && !(borrow_env && *borrow_env == '0'));
    {
        // BGZF input is inflated on the (first) device of the pipeline, which is idle most of an end-to-end run (stub rings have none)
        tbk_classifier *c0 = tbk_pipeline_classifier(p, 0);
        if (c0) (void)tbk_fastx_set_device(reader, tbk_classifier_device(c0));
    }
    tbk_bin_writer *writer = nullptr;
    rc = tbk_bin_writer_open(out_a, out_b, out_u, gzip_output, gzip_level, 0, &writer);
    if (rc) { tbk_fastx_close(reader); return rc; }
    if (gzip_output) {
        // the bins' gzip members are coded on the (first) device of the pipeline - it is idle 95 % of an end-to-end run - unless
        // TBK_GZIP_ENCODER says cpu / zlib (a test pipeline of stub rings has no device: the host's encoder)
        tbk_classifier *c0 = tbk_pipeline_classifier(p, 0);
        if (c0) {
            rc = tbk_bin_writer_use_device(writer, tbk_classifier_device(c0));
            if (rc) { tbk_bin_writer_close(writer); tbk_fastx_close(reader); return rc; }
        }
    }

    st.gzip_encoder = !gzip_output ? 0 : tbk_bin_writer_encoder(writer) ? 2 : 1;

    // batches kept submitted: the rings' slots plus one waiting per ring (what the pipeline admits), so that a feeder
    // whose ring has just got room finds its next batch queued already
    const int depth = tbk_pipeline_depth(p) + tbk_pipeline_devices(p);
    const int n_items = depth + 3;  // submitted + one apiece for reader, queues and writer
    std::vector<Item> items((size_t)n_items);
    Chan<Item *> free_q, filled_q, done_q;
    Failure failure;
    for (Item &it : items) {
        rc = tbk_fastx_batch_create(&it.batch);
        if (rc) break;
        free_q.put(&it);
    }
    auto counts_for = [](Item *it, uint64_t n) -> bool {
        if (n <= it->cap_reads) return true;
        if (it->counts) { if (it->counts_pinned) tbk_host_free(it->counts); else free(it->counts); }
        const size_t cap = (size_t)n + (size_t)n / 4 + 1024;
        it->counts = (int32_t *)tbk_host_alloc(cap * 2 * sizeof(int32_t));
        it->counts_pinned = it->counts != nullptr;
        if (!it->counts) it->counts = (int32_t *)malloc(cap * 2 * sizeof(int32_t));  // no device (the testing hook's stub rings)
        it->cap_reads = it->counts ? cap : 0;
        return it->counts != nullptr;
    };

    double read_s = 0, write_s = 0, gpu_wait_s = 0, setup_s = since(t_start), close_s = 0;
    double reader_idle_s = 0, writer_idle_s = 0, feed_idle_s = 0, submit_s = 0, collect_s = 0;   // (each thread's wait for its queue; TBK_WRITE_TIMING prints them)
    const bool write_timing = getenv("TBK_WRITE_TIMING") != nullptr;  // the writer's ms per batch, in tenths of the run, to stderr
    std::vector<double> per_batch_ms;
    std::thread reader_thread, writer_thread;
    if (!rc) {
        reader_thread = std::thread([&] {
            Item *it = nullptr;
            for (;;) {
                const auto t_idle = Clock::now();
                if (failure.any() || !free_q.get(it)) break;
                reader_idle_s += since(t_idle);
                const auto t = Clock::now();
                const int r = tbk_fastx_next(reader, it->batch, batch_bases, batch_reads);
                read_s += since(t);
                if (r) { failure.set(r, tbk_last_error()); break; }
                uint64_t n = 0;
                const uint64_t *off = nullptr;
                (void)tbk_fastx_batch_view(it->batch, &n, nullptr, &off, nullptr, nullptr, nullptr, nullptr, nullptr);
                if (n == 0) break;
                it->n_reads = n;
                it->n_bases = off[n];
                filled_q.put(it);
            }
            filled_q.close();
        });
        writer_thread = std::thread([&] {
            Item *it = nullptr;
            std::vector<double> sa, sb;
            std::vector<char> bins, tsv;
            for (;;) {
                const auto t_idle = Clock::now();
                if (!done_q.get(it)) break;
                writer_idle_s += since(t_idle);
                if (!failure.any()) {
                    const auto t = Clock::now();
                    const uint64_t n = it->n_reads;
                    sa.resize(n); sb.resize(n); bins.resize(n);
                    int r = tbk_score_and_bin(it->counts, n, num_kmers_a, num_kmers_b, sa.data(), sb.data(), bins.data());
                    if (!r) r = tbk_bin_writer_write(writer, it->batch, bins.data());
                    size_t len = 0;
                    if (!r && tsv_fd >= 0) {
                        r = tbk_format_tsv(it->batch, bins.data(), sa.data(), sb.data(), nullptr, 0, &len);
                        if (!r) { tsv.resize(len); r = tbk_format_tsv(it->batch, bins.data(), sa.data(), sb.data(), tsv.data(), tsv.size(), &len); }
                        if (!r && !write_fd(tsv_fd, tsv.data(), len)) { r = TBK_ERR_IO; tbk_set_error_(r, (std::string("writing the TSV: ") + strerror(errno)).c_str()); }
                    }
                    if (r) failure.set(r, tbk_last_error());
                    write_s += since(t);
                    if (write_timing) per_batch_ms.push_back(since(t) * 1e3);
                }
                free_q.put(it);
            }
        });

        // this thread keeps the device(s) fed; a collector takes the batches back in input order and hands them to the writer the moment
        // they are classified (a reader that delivers in bursts - windows of inflated text - must not find its batches parked behind a
        // submit loop that waits for the next one to be read)
        Chan<std::pair<uint64_t, Item *>> flying_q;
        std::mutex fly_mu;
        std::condition_variable fly_cv;
        int in_flight = 0;
        std::thread collector([&] {
            std::pair<uint64_t, Item *> f;
            while (flying_q.get(f)) {
                const auto t = Clock::now();
                const int r = tbk_pipeline_wait(p, f.first, nullptr);
                collect_s += since(t);
                if (r) failure.set(r, tbk_last_error());
                done_q.put(f.second);
                { std::lock_guard<std::mutex> lk(fly_mu); in_flight--; }
                fly_cv.notify_all();
            }
        });
        Item *it = nullptr;
        for (;;) {
            const auto t_idle = Clock::now();
            if (!filled_q.get(it)) break;
            feed_idle_s += since(t_idle);
            if (failure.any()) { free_q.put(it); continue; }
            {   // as many submitted as the pipeline admits
                const auto t = Clock::now();
                std::unique_lock<std::mutex> lk(fly_mu);
                fly_cv.wait(lk, [&] { return in_flight < depth; });
                in_flight++;
                gpu_wait_s += since(t);
            }
            const auto t_submit = Clock::now();
            const uint32_t *codes = nullptr, *exc_chunk = nullptr;
            const uint16_t *exc_mask = nullptr;
            const uint8_t *bases = nullptr;
            const uint64_t *off = nullptr;
            uint64_t n_exc = 0, tk = 0;
            int r = counts_for(it, it->n_reads) ? TBK_OK : TBK_ERR_NOMEM;
            if (!r) r = tbk_fastx_batch_view(it->batch, nullptr, &bases, &off, nullptr, nullptr, nullptr, nullptr, nullptr);
            if (!r) r = tbk_fastx_batch_packed(it->batch, &codes, &exc_chunk, &exc_mask, &n_exc);
            if (!r) {
                if (codes) r = tbk_pipeline_submit_packed(p, codes, exc_chunk, exc_mask, n_exc, off, it->n_reads, it->counts, &tk);
                else r = tbk_pipeline_submit(p, bases, off, it->n_reads, it->counts, &tk);
            }
            if (r) {
                failure.set(r, tbk_last_error());
                free_q.put(it);
                { std::lock_guard<std::mutex> lk(fly_mu); in_flight--; }
                continue;
            }
            st.reads += it->n_reads; st.bases += it->n_bases; st.batches++;
            flying_q.put({tk, it});
            submit_s += since(t_submit);
        }
        flying_q.close();
        collector.join();
        done_q.close();
        writer_thread.join();
        free_q.close();
        reader_thread.join();
    }
    if (write_timing && !per_batch_ms.empty()) {
        std::string line = "tbk-write-timing ms per batch, by tenth of the run:";
        const size_t nb = per_batch_ms.size();
        for (size_t d = 0; d < 10; d++) {
            const size_t lo = nb * d / 10, hi = std::max(lo + 1, nb * (d + 1) / 10);
            double sum = 0;
            for (size_t i = lo; i < hi && i < nb; i++) sum += per_batch_ms[i];
            char buf[32];
            snprintf(buf, sizeof buf, " %.1f", sum / (double)(std::min(hi, nb) - lo));
            line += buf;
        }
        fprintf(stderr, "%s\n", line.c_str());
    }
    const auto t_close = Clock::now();
    int rc_close = tbk_bin_writer_close(writer);
    const double close_writer_s = since(t_close);
    tbk_fastx_close(reader);
    const double close_reader_s = since(t_close) - close_writer_s;
    for (Item &it : items) {
        if (it.batch) tbk_fastx_batch_destroy(it.batch);
        if (it.counts) { if (it.counts_pinned) tbk_host_free(it.counts); else free(it.counts); }
    }
    close_s = since(t_close);
    if (write_timing)
        fprintf(stderr, "tbk-loop-waits the reader waited %.3f s for a free batch, this thread %.3f s for a read one and %.3f s for room in the rings (and spent %.3f s submitting), the collector "
                "%.3f s for the oldest batch submitted, the writer %.3f s for a classified one; %d batches circulate, %d of them submitted\n", reader_idle_s, feed_idle_s, gpu_wait_s, submit_s,
                collect_s, writer_idle_s, n_items, depth);
    if (write_timing)
        fprintf(stderr, "tbk-loop-timing opening the reader, the writer and the batches %.3f s; closing them %.3f s (writer %.3f, reader %.3f, batches %.3f)\n", setup_s, close_s,
                close_writer_s, close_reader_s, close_s - close_writer_s - close_reader_s);
    st.read_s = read_s; st.write_s = write_s; st.gpu_wait_s = gpu_wait_s; st.total_s = since(t_start);
    if (stats) *stats = st;
    if (rc) return rc;
    if (failure.any()) { tbk_set_error_(failure.rc, failure.msg.c_str()); return failure.rc; }
    return rc_close;
}
