// tbk_fastx.cpp — native I/O either side of the hot path:
//
//   tbk_fastx_*      FASTA/FASTQ(.gz) batch reader that reproduces the records of the
//                    reference's readfq (src/trio_binning/seq.py:45-92) byte for byte,
//                    quirks included, and lays each batch out exactly as the classifier's
//                    C-ABI wants it (bases back to back in pinned memory + offsets).
//   tbk_bin_writer_* ordered writer of the three bins with the byte format of Read.print
//                    (seq.py:27-42) and the file naming of open_outfiles (seq.py:117-134),
//                    gzip members deflated in parallel.
//   tbk_format_tsv   the stdout TSV of classify_by_kmers.py:117 including Python's
//                    str(float) formatting.
//
// Host code only (no kernels).  Text is handled as bytes; for ASCII input this is exactly
// Python's behaviour in text mode with universal newlines ("\n", "\r\n" and "\r" all end a
// line).
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <immintrin.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <deque>
#include <condition_variable>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <vector>

#include "../../include/tbk.h"
#include "tbk_inflate.h"
#include "tbk_pack.h"
#include "tbk_gdeflate.h"

uint32_t tbk_crc32(uint32_t crc, const uint8_t *p, size_t n);  // tbk_crc.cpp: zlib's crc32() by carry-less multiplication

extern "C" void tbk_set_error_(int code, const char *msg);  // tbk_host.cpp
extern "C" void *tbk_pin_alloc_(size_t bytes);              // tbk_host.cpp: pinned staging memory (huge pages registered with the runtime)
extern "C" void tbk_pin_free_(void *p);

static int ffail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tbk_set_error_(code, buf);
    return code;
}

// Second pass of the guessing inflater (LineSource::pinflate_loop): 16-bit symbols -> bytes.  A symbol
// is a byte value or a marker 0x8000 + i for byte i of the window `w` in front of the chunk.  Returns
// the OR of everything stored: above 0xFF when a marker pointed before the member's first byte.
static uint32_t resolve_symbols(const uint16_t *src, size_t n, const uint16_t *w, uint8_t *dst) {
    uint32_t seen = 0;
    for (size_t k = 0; k < n; k++) {
        uint32_t v = src[k];
        if (v >= 0x8000u) v = w[v - 0x8000u];
        seen |= v;
        dst[k] = (uint8_t)v;
    }
    return seen;
}
__attribute__((target("avx2"))) static uint32_t resolve_symbols_avx2(const uint16_t *src, size_t n, const uint16_t *w, uint8_t *dst) {
    uint32_t seen = 0;
    __m256i seen_v = _mm256_setzero_si256();
    size_t k = 0;
    for (; k + 32 <= n; k += 32) {  // 32 symbols without a marker among them (the rule, past a chunk's first stretch): narrow and store
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + k)), b = _mm256_loadu_si256((const __m256i *)(src + k + 16));
        if ((uint32_t)_mm256_movemask_epi8(_mm256_or_si256(a, b)) & 0xAAAAAAAAu) {
            seen |= resolve_symbols(src + k, 32, w, dst + k);
            continue;
        }
        seen_v = _mm256_or_si256(seen_v, _mm256_or_si256(a, b));
        _mm256_storeu_si256((__m256i *)(dst + k), _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8));
    }
    alignas(32) uint16_t lanes[16];
    _mm256_store_si256((__m256i *)lanes, seen_v);
    for (int q = 0; q < 16; q++) seen |= lanes[q];
    return seen | resolve_symbols(src + k, n - k, w, dst + k);
}

// =======================================================================================
// line source: plain or gzip file -> lines with universal-newline semantics
// =======================================================================================
// The GPU inflater's windows (bgzf_loop_gpu): the inflater and which of its slots are out.  A slot's text lies in pinned memory of the
// inflater's; it is out while the parser reads it, while it waits in the queue, and while a batch whose records were left in it
// (borrowed, as from a plain file's mapping) is alive.  Every holder has a WindowHold; the slot is the inflater's again with the last
// of them, and the inflater itself goes with the last reference to this object - a batch may outlive its reader.
struct GpuWindows {
    std::mutex mu;
    std::condition_variable cv;
    tbk_ginflate *g = nullptr;
    bool slot_free[TBK_GINFLATE_SLOTS];
    bool slot_sized[TBK_GINFLATE_SLOTS];   // its buffers have been sized (by the worker itself: the first one used; by its helper: the others)
    bool sizing_done = false;              // the helper has gone through the slots (one it could not size ends it: the ring is as deep as it got)
    bool abandoned = false;   // the reader is being closed: the worker must not wait for a slot
    GpuWindows() { for (bool &f : slot_free) f = true; for (bool &f : slot_sized) f = false; }
    GpuWindows(const GpuWindows &) = delete;
    GpuWindows &operator=(const GpuWindows &) = delete;
    ~GpuWindows() { if (g) tbk_ginflate_destroy(g); }
};
struct WindowHold {
    std::shared_ptr<GpuWindows> w;
    int slot;
    WindowHold(std::shared_ptr<GpuWindows> w_, int slot_) : w(std::move(w_)), slot(slot_) {}
    WindowHold(const WindowHold &) = delete;
    WindowHold &operator=(const WindowHold &) = delete;
    ~WindowHold() { { std::lock_guard<std::mutex> lk(w->mu); w->slot_free[slot] = true; } w->cv.notify_all(); }
};

struct LineSource {
    int fd = -1;
    bool gz = false;
    z_stream zs;
    bool zs_live = false;
    bool raw_eof = false;      // no more bytes from the file
    bool text_eof = false;     // no more decoded bytes
    std::vector<uint8_t> zin;  // compressed input window
    size_t zin_pos = 0, zin_end = 0;
    std::vector<uint8_t> buf;  // decoded text window
    size_t pos = 0, end = 0;
    // Where the text at hand lies: `buf`, or - a window the GPU inflater has written - that window's pinned buffer itself, with what
    // the parser had left of the window before copied in front of it (`view`; `held` keeps the window's slot out until the next
    // window has been taken, and the batches whose records stay in the window share it).  pos and end count from text().
    uint8_t *view = nullptr;
    std::shared_ptr<WindowHold> held;
    std::shared_ptr<GpuWindows> windows;
    const uint8_t *text() const { return view ? view : buf.data(); }
    // May a batch leave its records in the window at hand?  Only while two more windows are free: the ring must turn over whatever
    // the batches' owner does with them (a caller who keeps every batch would otherwise stop the reader for good).
    bool window_can_lend() {
        if (!view || !held || !windows) return false;
        std::lock_guard<std::mutex> lk(windows->mu);
        int free_now = 0;
        for (int i = 0; i < TBK_GINFLATE_SLOTS; i++) free_now += windows->slot_sized[i] && windows->slot_free[i];
        return free_now >= 2;
    }
    bool skip_lf = false;      // previous line ended in '\r' at the window edge: swallow a leading '\n'
    std::string err;
    // BGZF (bgzip) files are gzip files whose members are independent blocks of <= 64 KiB that
    // carry their own compressed size in a "BC" extra subfield: such members can be inflated side
    // by side.  Same bytes as the sequential path; taken while every member at hand is such a block.
    bool bgzf = false;
    bool bgzf_seen = false;  // a BGZF member has been read: zero padding may follow it (Python's GzipFile skips it)
    // BGZF blocks inflated on the GPU (tbk_fastx_set_device; csrc/tbk_gdeflate.hip, second half): one wave per block, windows of
    // 96 MB of the file four deep.  -1: on the host's threads.
    int gpu_device = -1;
    uint64_t gpu_windows = 0, gpu_blocks = 0;
    double gpu_stage_s = 0, gpu_wait_s = 0, gpu_slot_wait_s = 0;
    double chunk_wait_s = 0, left_copy_s = 0; uint64_t left_bytes = 0;   // the parser's side: waiting for a window, copying what it had left in front of it
    uint64_t gpu_in_place = 0, gpu_copied = 0;   // windows parsed where the device wrote them / copied into `buf` (more left over than the room in front)
    int threads = 1;
    // Ordinary gzip streams go through the library's own DEFLATE decoder (tbk_inflate.h) on the
    // memory-mapped file: about twice zlib's speed on FASTQ, and that stream is what a run on .gz
    // input waits for.  TBK_INFLATE=zlib keeps zlib's inflate.
    bool fast = false;
    bool threaded = false;     // a worker thread produces the text (own decoder, or BGZF windows)
    TbkInflate inf;
    const uint8_t *map = nullptr;
    size_t map_size = 0;
    size_t member_start = 0;   // index in buf where the current gzip member's output began
    uint32_t member_crc = 0;   // CRC-32 of the member's output so far
    uint64_t member_size = 0;

    bool open_path(const char *path, bool gzip) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open ") + path + ": " + strerror(errno); return false; }
        gz = gzip;
        buf.resize(1 << 22);
        if (gz) {
            zin.resize(1 << 20);
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, 15 + 16) != Z_OK) { err = "inflateInit2 failed"; return false; }
            zs_live = true;
            threads = std::max(1, tbk_host_threads());
            uint8_t head[18];
            const ssize_t n = ::pread(fd, head, sizeof head, 0);
            bgzf = threads > 1 && n == (ssize_t)sizeof head && bgzf_block_size(head, sizeof head) > 0;
            const char *how = getenv("TBK_INFLATE");
            struct stat st;
            if (bgzf && fstat(fd, &st) == 0 && st.st_size > 0) {
                // BGZF blocks are parsed and inflated where the file is mapped: no copy of the compressed bytes through read()
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) { (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL); map = (const uint8_t *)m; map_size = (size_t)st.st_size; }
            }
            if (bgzf && !map) zin.resize((size_t)32 << 20);
            if (!bgzf && !(how && strcmp(how, "zlib") == 0) && fstat(fd, &st) == 0 && st.st_size > 0) {
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                    map = (const uint8_t *)m; map_size = (size_t)st.st_size;
                    inf.reset(map, map_size);
                    member_crc = (uint32_t)crc32(0L, Z_NULL, 0);
                    fast = true;
                }
            }
            threaded = fast || bgzf;
        }
        return true;
    }
    // CRC-32 of buf[lo..hi) folded into the running member CRC, computed by several threads
    void fold_crc(const uint8_t *base, size_t lo, size_t hi) {
        const size_t n = hi - lo;
        if (!n) return;
        const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n >> 20));
        std::vector<uint32_t> part((size_t)nt);
        std::vector<size_t> len((size_t)nt);
        auto one = [&](int t) {
            const size_t a = lo + n * (size_t)t / nt, b = lo + n * (size_t)(t + 1) / nt;
            uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
            c = tbk_crc32(c, base + a, b - a);
            part[(size_t)t] = c; len[(size_t)t] = b - a;
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(one, t);
        one(0);
        for (std::thread &th : pool) th.join();
        for (int t = 0; t < nt; t++) member_crc = (uint32_t)crc32_combine(member_crc, part[(size_t)t], (z_off_t)len[(size_t)t]);
        member_size += n;
    }
    // The decoder runs on its own thread, in its own window (it needs the last 32 KiB of its
    // output in front of the write position), and hands the text over in chunks: inflating and
    // parsing then overlap instead of taking turns.
    // A chunk owns the buffer it was inflated into: [ up to 32 KiB of the text before it | new text ].
    // (ext / slot: the text lies in a pinned output buffer of the GPU inflater instead of `data`; the slot is the inflater's again once
    // the parser has taken the text over)
    struct Chunk { std::vector<uint8_t> data; size_t off = 0, len = 0; bool last = false, fallback = false; std::string err; uint8_t *ext = nullptr; size_t room = 0; std::shared_ptr<WindowHold> hold; };
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Chunk> ready;
    bool stop = false, started = false;
    size_t queue_cap = 3;

    // buffers of chunks the parser is done with, for the next chunks (a fresh 8 MiB vector is a
    // trip to the kernel, a zero-fill and two thousand page faults)
    std::vector<std::vector<uint8_t>> spare;
    std::vector<uint8_t> take_buffer(size_t n) {
        std::vector<uint8_t> v;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!spare.empty()) { v.swap(spare.back()); spare.pop_back(); }
        }
        if (v.size() < n) { std::vector<uint8_t>().swap(v); v.resize(n); }  // a vector that grows copies its old bytes: start anew
        return v;
    }
    void push(Chunk &&c) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || ready.size() < queue_cap; });
        if (stop) return;
        ready.push_back(std::move(c));
        cv.notify_all();
    }
    // bgzip'ed input on the worker thread: windows of blocks inflated side by side, one chunk each
    void bgzf_loop() {
        for (;;) {
            { std::lock_guard<std::mutex> lk(mu); if (stop) return; }
            Chunk c;
            bool at_end = false;
            size_t text_len = 0;
            const int r = bgzf_window(c.data, &text_len, &at_end);
            if (r < 0) { c.err = err; c.last = true; push(std::move(c)); return; }
            if (r == 0) { c.fallback = true; c.data.clear(); push(std::move(c)); return; }  // an ordinary member: the parser thread takes over
            c.off = 0; c.len = text_len; c.last = at_end;
            push(std::move(c));
            if (at_end) return;
        }
    }
    // The blocks of the mapped file from byte `from` on, as many as lie within `span_want` bytes: their deflate streams (relative to
    // `from`), text lengths and CRC-32s.  *span = bytes of whole blocks (and padding) walked.  Returns -1 for a corrupt block.
    int bgzf_scan(size_t from, size_t span_want, std::vector<tbk_ginflate_block> &blks, size_t *span, size_t *out_total) {
        const uint8_t *zbase = map + from;
        const size_t avail = map_size - from;
        size_t p = 0;
        *out_total = 0;
        while (p < avail && p < span_want) {
            if (bgzf_seen && zbase[p] == 0) { p++; continue; }  // zero padding between members
            const size_t bs = bgzf_block_size(zbase + p, avail - p);
            if (bs == 0) break;                                  // not a BGZF block (an ordinary member, or the file is cut off)
            if (bs < 26) { err = "corrupt BGZF block"; return -1; }
            if (p + bs > avail) break;
            const uint8_t *b = zbase + p;
            const size_t xlen = b[10] | ((size_t)b[11] << 8), hdr = 12 + xlen;
            if (hdr + 8 > bs) { err = "corrupt BGZF block"; return -1; }
            const uint32_t crc = (uint32_t)b[bs - 8] | ((uint32_t)b[bs - 7] << 8) | ((uint32_t)b[bs - 6] << 16) | ((uint32_t)b[bs - 5] << 24);
            const uint32_t isize = (uint32_t)b[bs - 4] | ((uint32_t)b[bs - 3] << 8) | ((uint32_t)b[bs - 2] << 16) | ((uint32_t)b[bs - 1] << 24);
            if (isize > (1u << 16)) { err = "corrupt BGZF block"; return -1; }
            blks.push_back(tbk_ginflate_block{(uint64_t)(p + hdr), (uint32_t)(bs - hdr - 8), isize, crc, 0});
            *out_total += isize;
            p += bs;
            bgzf_seen = true;
        }
        *span = p;
        return 0;
    }
    // BGZF on the GPU: windows of the mapped file (TBK_BGZF_GPU_WINDOW bytes, default 96 MB: ~3800 blocks, one wave each) are copied
    // into the inflater's pinned input by a few threads, inflated and CRC-checked on the device, and their text comes back into pinned
    // memory, where the parser reads it.  Four windows deep (TBK_GINFLATE_SLOTS): one is staged while one is on the device, one is with
    // the parser and one more with it or with batches that left their records in it.  Measured at configs[1] scale (Gbases/s end to
    // end): 256 MB x 5: 3.2, 128 MB x 8: 3.6-3.9, x 5: 4.0-4.2, x 4: 4.3-4.5, 96 MB x 4: 4.5 - the pinned memory costs a tenth of a
    // second per 600 MB to get and as much to give back.  Anything that is not a BGZF block (an ordinary member behind the blocks),
    // and any failure to set the inflater up, hands over to the host path exactly where bgzf_loop would be.
    void bgzf_loop_gpu() {
        tbk_ginflate *g = nullptr;
        if (tbk_ginflate_create(gpu_device, &g) != TBK_OK) { bgzf_loop(); return; }
        std::shared_ptr<GpuWindows> w = std::make_shared<GpuWindows>();
        w->g = g;   // (goes with the last window that is out, not with this thread)
        { std::lock_guard<std::mutex> lk(mu); windows = w; }
        const size_t window = std::max<size_t>((size_t)1 << 16, env_size("TBK_BGZF_GPU_WINDOW", (size_t)96 << 20));
        // room in front of a window's text for what the parser has left of the window before (it asks for more when less than a batch's
        // worth, at most 64 MiB and a record, is at hand): the window is then parsed where it lies.  0 = every window is copied.
        const size_t head_room = env_size("TBK_BGZF_GPU_ROOM", (size_t)72 << 20);
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        auto fail_with = [&](const std::string &msg) { Chunk c; c.err = msg; c.last = true; push(std::move(c)); };
        // The windows on the device, oldest first, and the thread that takes them home: it waits for a window's kernels, copies its text
        // into the slot's pinned output and hands it to the parser, while this thread stages the next window.
        struct Flight { int slot; bool last; };
        std::deque<Flight> flights;
        std::mutex fl_mu;
        std::condition_variable fl_cv;
        bool no_more = false, collector_failed = false;
        std::thread collector([&] {
            for (;;) {
                Flight f;
                {
                    std::unique_lock<std::mutex> lk(fl_mu);
                    fl_cv.wait(lk, [&] { return no_more || !flights.empty(); });
                    if (flights.empty()) return;
                    f = flights.front();
                }
                uint8_t *base = nullptr;
                size_t n = 0;
                uint32_t bad = 0;
                const double t0 = now();
                const int rc = tbk_ginflate_wait(g, f.slot, &base, &n, &bad);
                gpu_wait_s += now() - t0;
                bool ok = true;
                if (rc) { fail_with(std::string("inflate: ") + tbk_last_error()); ok = false; }
                else if (bad) { fail_with("inflate: corrupt BGZF block"); ok = false; }
                else {
                    Chunk c;
                    c.ext = base + head_room; c.room = head_room; c.off = 0; c.len = n; c.hold = std::make_shared<WindowHold>(w, f.slot); c.last = f.last;
                    push(std::move(c));
                }
                {
                    std::lock_guard<std::mutex> lk(fl_mu);
                    flights.pop_front();
                    if (!ok) collector_failed = true;
                }
                fl_cv.notify_all();
                if (!ok) return;
            }
        });
        // every window submitted has gone to the parser (true), or the collector has reported an error of its own (false)
        auto drain = [&]() -> bool {
            { std::lock_guard<std::mutex> lk(fl_mu); no_more = true; }
            fl_cv.notify_all();
            if (collector.joinable()) collector.join();
            return !collector_failed;
        };
        struct JoinCollector { decltype(drain) &d; ~JoinCollector() { (void)d(); } } join_collector{drain};
        // the slots' buffers are sized for the first window and a quarter more (what the first window's own buffers get), those of the
        // slots behind the first by a helper thread while the first window is on its way: a tenth of a second each, not in a row in front
        std::thread sizer;
        struct JoinSizer { std::thread &t; ~JoinSizer() { if (t.joinable()) t.join(); } } join_sizer{sizer};
        bool first_window = true;
        for (;;) {
            { std::lock_guard<std::mutex> lk(mu); if (stop) return; }
            { std::lock_guard<std::mutex> lk(fl_mu); if (collector_failed) return; }
            std::vector<tbk_ginflate_block> blks;
            size_t span = 0, out_total = 0;
            if (bgzf_scan(bgzf_map_pos, window, blks, &span, &out_total) < 0) { if (drain()) fail_with(err); return; }
            const bool at_end = bgzf_map_pos + span >= map_size;
            if (blks.empty()) {
                if (!at_end && span > 0) { bgzf_map_pos += span; continue; }   // only padding: drop it and look again
                if (!drain()) return;
                if (at_end) { Chunk c; c.last = true; push(std::move(c)); return; }
                if (map_size - bgzf_map_pos >= 18 && bgzf_block_size(map + bgzf_map_pos, map_size - bgzf_map_pos) == 0) {
                    // an ordinary gzip member follows: the sequential path reads the file itself, from here (as bgzf_window hands over)
                    if (lseek(fd, (off_t)bgzf_map_pos, SEEK_SET) < 0) { fail_with(std::string("lseek: ") + strerror(errno)); return; }
                    zin_pos = zin_end = 0; raw_eof = false;
                    Chunk c; c.fallback = true; push(std::move(c));
                    return;
                }
                fail_with("truncated gzip file");
                return;
            }
            if (out_total == 0) {   // only empty blocks (the end-of-file marker)
                bgzf_map_pos += span;
                if (at_end) { if (drain()) { Chunk c; c.last = true; push(std::move(c)); } return; }
                continue;
            }
            // a slot nobody holds any more
            int slot = -1;
            {
                const double t0 = now();
                std::unique_lock<std::mutex> lk(w->mu);
                w->cv.wait(lk, [&] {
                    if (w->abandoned) return true;
                    int sized = 0;
                    for (int i = 0; i < TBK_GINFLATE_SLOTS; i++) {
                        sized += w->slot_sized[i];
                        if (slot < 0 && w->slot_free[i] && (first_window || w->slot_sized[i])) slot = i;
                    }
                    if (slot >= 0) return true;
                    return !first_window && w->sizing_done && sized < 2;   // (one window cannot turn over: the parser keeps it until the next arrives)
                });
                if (slot < 0) {
                    if (!w->abandoned) { lk.unlock(); if (drain()) fail_with("inflate: no pinned memory for two BGZF windows (TBK_BGZF_INFLATE=cpu inflates on the host)"); }
                    return;
                }
                w->slot_free[slot] = false;
                gpu_slot_wait_s += now() - t0;
            }
            if (first_window) {
                first_window = false;
                const size_t in_b = span + span / 10, n_b = blks.size() + blks.size() / 4, out_b = head_room + out_total + out_total / 10;   // (windows are cut to one size: a tenth of slack; a window that needs more grows its buffers)
                const char *limit = getenv("TBK_BGZF_GPU_SLOTS");   // (tests: as if the memory for more windows than this were not there)
                const int max_slots = limit ? atoi(limit) : TBK_GINFLATE_SLOTS;
                if (max_slots < 2 || tbk_ginflate_reserve(g, slot, in_b, n_b, out_b) != TBK_OK) {
                    // not even one window's buffers: the host's threads inflate, from this very block on
                    { std::lock_guard<std::mutex> lk(w->mu); w->slot_free[slot] = true; }
                    (void)drain();
                    gpu_device = -1;
                    bgzf_loop();
                    return;
                }
                { std::lock_guard<std::mutex> lk(w->mu); w->slot_sized[slot] = true; }
                sizer = std::thread([w, g, slot, in_b, n_b, out_b, max_slots] {
                    int have = 1;
                    for (int i = 0; i < TBK_GINFLATE_SLOTS && have < max_slots; i++) {
                        if (i == slot) continue;
                        { std::lock_guard<std::mutex> lk(w->mu); if (w->abandoned) break; }
                        if (tbk_ginflate_reserve(g, i, in_b, n_b, out_b) != TBK_OK) break;
                        { std::lock_guard<std::mutex> lk(w->mu); w->slot_sized[i] = true; }
                        w->cv.notify_all();
                        have++;
                    }
                    { std::lock_guard<std::mutex> lk(w->mu); w->sizing_done = true; }
                    w->cv.notify_all();
                });
            }
            const double t0 = now();
            uint8_t *in = tbk_ginflate_input(g, slot, span);
            if (!in) { if (drain()) fail_with("inflate: no pinned memory for a BGZF window"); return; }
            {
                const int nt = std::min(4, std::max(1, threads));
                std::vector<std::thread> pool;
                const uint8_t *from = map + bgzf_map_pos;
                for (int t = 0; t < nt; t++) {
                    const size_t lo = span * (size_t)t / nt, hi = span * (size_t)(t + 1) / nt;
                    pool.emplace_back([=] { memcpy(in + lo, from + lo, hi - lo); });
                }
                for (std::thread &th : pool) th.join();
                (void)madvise((void *)(((uintptr_t)from + 4095) & ~(uintptr_t)4095), span > 8192 ? span - 8192 : 0, MADV_DONTNEED);   // (read once)
            }
            gpu_stage_s += now() - t0;
            if (tbk_ginflate_submit(g, slot, span, blks.data(), blks.size(), head_room) != TBK_OK) { const std::string m = std::string("inflate: ") + tbk_last_error(); if (drain()) fail_with(m); return; }
            gpu_windows++; gpu_blocks += blks.size();
            { std::lock_guard<std::mutex> lk(fl_mu); flights.push_back(Flight{slot, at_end}); }
            fl_cv.notify_all();
            bgzf_map_pos += span;
            if (at_end) { (void)drain(); return; }
        }
    }
    void inflate_loop() {
        constexpr size_t HIST = 32768, ROOM = (size_t)8 << 20;
        std::vector<uint8_t> tail;  // the last 32 KiB inflated so far
        for (;;) {
            { std::lock_guard<std::mutex> lk(mu); if (stop) return; }
            Chunk c;
            c.data.resize(HIST + ROOM);
            if (!tail.empty()) memcpy(c.data.data() + HIST - tail.size(), tail.data(), tail.size());
            // the member may have begun before this buffer: then matches may reach back through all of `tail`
            size_t pos = HIST, mstart = member_size >= tail.size() ? HIST - tail.size() : HIST - (size_t)member_size;
            TbkInflate::Status st = TbkInflate::NEED_OUTPUT;
            while (c.data.size() - pos >= 4096) {  // a buffer may hold the ends and beginnings of several members
                const size_t before = pos;
                st = inf.run(c.data.data(), &pos, c.data.size(), mstart);
                if (st == TbkInflate::ERROR) break;
                fold_crc(c.data.data(), before, pos);
                if (st == TbkInflate::MEMBER_DONE) {
                    if (member_crc != inf.trailer_crc() || (uint32_t)member_size != inf.trailer_isize()) { st = TbkInflate::ERROR; c.err = "inflate: gzip CRC or size mismatch"; break; }
                    member_crc = (uint32_t)crc32(0L, Z_NULL, 0);
                    member_size = 0;
                    mstart = pos;
                    continue;
                }
                break;  // NEED_OUTPUT (buffer full) or INPUT_DONE
            }
            if (st == TbkInflate::ERROR) {
                if (c.err.empty()) c.err = std::string("inflate: ") + inf.error();
                c.last = true;
                push(std::move(c));
                return;
            }
            c.off = HIST; c.len = pos - HIST;
            c.last = st == TbkInflate::INPUT_DONE;
            const size_t keep = std::min(HIST, tail.size() + c.len);
            std::vector<uint8_t> next_tail(c.data.data() + pos - keep, c.data.data() + pos);
            tail.swap(next_tail);
            const bool done = c.last;
            if (c.len || c.last) push(std::move(c));
            if (done) return;
        }
    }
    // ---- an ordinary gzip stream on several threads -------------------------------------------------
    // DEFLATE is one chain: a block can be found only by decoding the one before it, and a match may
    // reach 32 KiB back into text that is not there yet.  Both are broken by guessing, and every guess
    // is checked (the idea of pugz / rapidgzip, restated for this reader's chunks).  A round cuts the
    // next stretch of the file into one span per thread.  Thread 0 continues the exact decoder.  Every
    // other thread looks for the first bit of its span where a dynamic-Huffman block can begin
    // (TbkInflate::open_dynamic_block_at: type bits, code counts, complete codes) and decodes from there
    // into 16-bit symbols: a byte value, or - where a match reaches back before the thread's own
    // output - a marker 0x8000 + i for "byte i of the 32 KiB window I do not have".  A thread stops in
    // front of the first block header at or past the next thread's guess.  Then the guesses are
    // checked in order: thread i's output counts only if thread i-1's counted and ended EXACTLY on
    // thread i's starting bit - so by induction every accepted chunk starts on a block boundary of
    // the real chain, and what it decoded is what the sequential decoder decodes there.  The windows
    // are then filled in front to back (only the last 32 KiB of each chunk, one after the other), the
    // markers replaced and the CRCs taken by all threads, and the chunks handed to the parser in order.
    // A guess that fails, a member that ends, a chunk that outgrows its buffer: the round is cut
    // there, the exact decoder takes over that state, and the next round goes on from it.
    // Text, CRC checks and errors are those of inflate_loop().  TBK_PINFLATE=0 turns this off;
    // TBK_PINFLATE_SPAN / TBK_PINFLATE_MIN set the span and the least file size (tests).
    static size_t env_size(const char *name, size_t dflt) {
        const char *e = getenv(name);
        return e && *e ? (size_t)strtoull(e, nullptr, 10) : dflt;
    }
    bool guessing() const {
        const char *e = getenv("TBK_PINFLATE");
        if (e && strcmp(e, "0") == 0) return false;
        return fast && threads >= 3 && map_size >= env_size("TBK_PINFLATE_MIN", (size_t)16 << 20);
    }
    struct GuessJob {
        TbkInflate dec;
        uint64_t start_bit = 0, stop_bit = 0;
        std::unique_ptr<uint16_t[]> sym;  // [ 32 Ki window | output ]; never zero-filled: its pages are first touched by the thread that decodes into them
        size_t sym_cap = 0;
        size_t n = 0;               // output symbols
        TbkInflate::Status st = TbkInflate::ERROR;
        std::vector<uint16_t> window;  // the real window in front of this chunk (filled in when known)
        Chunk out;
        uint32_t crc = 0;
        bool bad_symbol = false;
    };
    // streams being inflated this way right now (find-unique-kmers reads a library's files side by
    // side): they share the host threads instead of each taking all of them
    static std::atomic<int> &guessers() { static std::atomic<int> n{0}; return n; }
    void pinflate_loop() {
        struct Here { Here() { guessers()++; } ~Here() { guessers()--; } } here;
        constexpr size_t HIST = 32768;
        constexpr uint16_t NOTHING = 0x7FFF;  // window position before the member's first byte
        // A round's span per thread: 8 MB of compressed bytes where the file has them for every thread (30 GB of FASTQ as one member,
        // 16 threads: 2.48 Gbases/s through the whole loop against 1.81 with spans of 2 MB - a round ends in three joins, and a short
        // round is mostly joins; profiles/r05/cli_gz_configs1_spans.json), less in a smaller file so that every thread gets one.
        const size_t span_pinned = env_size("TBK_PINFLATE_SPAN", 0);
        const size_t span = span_pinned ? std::max<size_t>(span_pinned, 4096)
                                        : std::min<size_t>((size_t)8 << 20, std::max<size_t>((size_t)1 << 20, map_size / (size_t)std::max(1, std::min(threads, 32))));
        size_t cap = std::max<size_t>(span * 5, (size_t)1 << 16);  // symbols per chunk; grows when chunks hit it
        const int nt_most = std::min(threads, 32);
        { std::lock_guard<std::mutex> lk(mu); queue_cap = (size_t)(2 * nt_most); }
        std::vector<GuessJob> jobs((size_t)nt_most);
        std::vector<uint16_t> tail(HIST, NOTHING);  // window in front of the exact decoder
        int rest = 0, backoff = 1;                  // rounds without guesses after a round that wasted them
        auto run_threads = [&](int n, auto &&fn) {
            std::vector<std::thread> pool;
            for (int t = 1; t < n; t++) pool.emplace_back([&fn, t] { fn(t); });
            fn(0);
            for (std::thread &th : pool) th.join();
        };
        const bool timing = getenv("TBK_PINFLATE_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        for (;;) {
            { std::lock_guard<std::mutex> lk(mu); if (stop) return; }
            const size_t base = (size_t)(inf.bit_position() >> 3);
            const double t0 = now();
            const int nt = std::max(1, nt_most / std::max(1, guessers().load()));
            // ---- guesses ----
            int n_jobs = 1;
            if (rest > 0) {
                rest--;
            } else if (nt >= 2) {
                std::vector<uint64_t> found((size_t)nt, ~0ull);
                run_threads(nt, [&](int t) {
                    if (t == 0) return;
                    const size_t lo = base + (size_t)t * span, hi = std::min(map_size, lo + span);
                    if (lo + 64 >= map_size) return;
                    TbkInflate &d = jobs[(size_t)t].dec;
                    for (uint64_t bit = (uint64_t)lo * 8; bit < (uint64_t)hi * 8; bit++)
                        if (d.open_dynamic_block_at(map, map_size, bit)) { found[(size_t)t] = bit; return; }
                });
                for (int t = 1; t < nt; t++) {
                    if (found[(size_t)t] == ~0ull) continue;
                    if (n_jobs != t) jobs[(size_t)n_jobs].dec = jobs[(size_t)t].dec;
                    jobs[(size_t)n_jobs].start_bit = found[(size_t)t];
                    n_jobs++;
                }
            }
            jobs[0].dec = inf;
            jobs[0].start_bit = inf.bit_position();
            for (int i = 0; i < n_jobs; i++) {
                GuessJob &j = jobs[(size_t)i];
                // the last chunk of a round ends at the first block boundary past its span
                j.stop_bit = i + 1 < n_jobs ? jobs[(size_t)i + 1].start_bit : ((j.start_bit >> 3) + span) * 8;  // may lie past the end of the file: then the file ends first
                if (j.sym_cap < HIST + cap + 512) { j.sym_cap = HIST + cap + 512; j.sym.reset(new uint16_t[j.sym_cap]); }
                if (i == 0) memcpy(j.sym.get(), tail.data(), HIST * 2);
                else for (size_t w = 0; w < HIST; w++) j.sym[w] = (uint16_t)(0x8000u + w);
            }
            const double t1 = now();
            // ---- first pass: decode ----
            run_threads(n_jobs, [&](int t) {
                GuessJob &j = jobs[(size_t)t];
                size_t pos = HIST;
                j.st = j.dec.run16(j.sym.get(), &pos, HIST + cap, j.stop_bit);
                j.n = pos - HIST;
            });
            const double t2 = now();
            // ---- which guesses hold ----
            int good = 1;
            while (good < n_jobs) {
                const GuessJob &prev = jobs[(size_t)good - 1];
                if (prev.st != TbkInflate::BOUNDARY || prev.dec.bit_position() != jobs[(size_t)good].start_bit) break;
                if (jobs[(size_t)good].st == TbkInflate::ERROR) break;  // the exact decoder will meet it and say what it is
                good++;
            }
            if (n_jobs > 1 && good * 2 < n_jobs) { rest = backoff; backoff = std::min(backoff * 2, 64); }
            else if (n_jobs > 1) backoff = 1;
            for (int i = 0; i < good; i++)
                if (jobs[(size_t)i].st == TbkInflate::NEED_OUTPUT) cap = std::min(cap * 2, std::max(cap, span * 16));
            // ---- the windows, front to back ----
            auto window_after = [&](const GuessJob &j, const std::vector<uint16_t> &before) {
                std::vector<uint16_t> w(HIST);
                if (j.st == TbkInflate::MEMBER_DONE) { std::fill(w.begin(), w.end(), NOTHING); return w; }
                const size_t take = std::min(HIST, j.n), keep = HIST - take;
                memcpy(w.data(), before.data() + (HIST - keep), keep * 2);
                const uint16_t *src = j.sym.get() + HIST + j.n - take;
                for (size_t k = 0; k < take; k++) { const uint16_t v = src[k]; w[keep + k] = v >= 0x8000u ? before[v - 0x8000u] : v; }
                return w;
            };
            jobs[0].window = tail;
            for (int i = 1; i < good; i++) jobs[(size_t)i].window = window_after(jobs[(size_t)i - 1], jobs[(size_t)i - 1].window);
            std::vector<uint16_t> next_tail = window_after(jobs[(size_t)good - 1], jobs[(size_t)good - 1].window);
            // ---- second pass: markers -> bytes, CRC ----
            run_threads(good, [&](int t) {
                GuessJob &j = jobs[(size_t)t];
                j.out = Chunk();
                j.out.data = take_buffer(j.n);
                static const bool avx2 = __builtin_cpu_supports("avx2");
                const uint32_t seen = avx2 ? resolve_symbols_avx2(j.sym.get() + HIST, j.n, j.window.data(), j.out.data.data())
                                           : resolve_symbols(j.sym.get() + HIST, j.n, j.window.data(), j.out.data.data());
                j.bad_symbol = seen > 0xFFu;
                uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
                c = tbk_crc32(c, j.out.data.data(), j.n);
                j.crc = c;
                j.out.off = 0; j.out.len = j.n;
            });
            const double t3 = now();
            if (timing) {
                size_t text = 0;
                for (int i = 0; i < good; i++) text += jobs[(size_t)i].n;
                fprintf(stderr, "tbk-pinflate round at byte %zu: %d guesses, %d chunks kept, %.1f MB of text: guess %.1f ms, decode %.1f ms, resolve %.1f ms\n",
                        base, n_jobs - 1, good, text / 1e6, t1 - t0, t2 - t1, t3 - t2);
            }
            // ---- hand over, in order ----
            for (int i = 0; i < good; i++) {
                GuessJob &j = jobs[(size_t)i];
                std::string problem;
                if (j.bad_symbol) problem = "inflate: distance too far back";
                else if (j.st == TbkInflate::ERROR) problem = std::string("inflate: ") + j.dec.error();
                if (problem.empty()) {
                    member_crc = (uint32_t)crc32_combine(member_crc, j.crc, (z_off_t)j.n);
                    member_size += j.n;
                    if (j.st == TbkInflate::MEMBER_DONE) {
                        if (member_crc != j.dec.trailer_crc() || (uint32_t)member_size != j.dec.trailer_isize()) problem = "inflate: gzip CRC or size mismatch";
                        member_crc = (uint32_t)crc32(0L, Z_NULL, 0);
                        member_size = 0;
                    }
                }
                if (!problem.empty()) {
                    Chunk c;
                    c.err = problem; c.last = true;
                    push(std::move(c));
                    return;
                }
                const bool done = j.st == TbkInflate::INPUT_DONE;
                j.out.last = done;
                if (j.out.len || done) push(std::move(j.out));
                if (done) return;
            }
            inf = jobs[(size_t)good - 1].dec;
            tail.swap(next_tail);
        }
    }
    // own decoder: returns like refill()
    bool refill_fast() {
        if (!started) { started = true; worker = std::thread([this] { if (bgzf && gpu_device >= 0 && map) bgzf_loop_gpu(); else if (bgzf) bgzf_loop(); else if (guessing()) pinflate_loop(); else inflate_loop(); }); }
        Chunk c;
        {
            const auto t0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty(); });
            chunk_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            c = std::move(ready.front());
            ready.pop_front();
            cv.notify_all();
        }
        if (!c.err.empty()) { err = c.err; return false; }
        if (c.ext && c.hold && end - pos <= c.room) {
            // a window of the GPU inflater, and room in front of it for what is left of the text at hand: parsed where it lies
            const size_t left = end - pos;
            uint8_t *start = c.ext - left;
            const auto t0 = std::chrono::steady_clock::now();
            if (left) memcpy(start, text() + pos, left);
            left_copy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); left_bytes += left;
            view = start; held = std::move(c.hold); pos = 0; end = left + c.len;   // (the window before this one is the inflater's again, or its batches')
            gpu_in_place++;
            if (c.last) text_eof = true;
            return true;
        }
        if (view) {  // back into `buf` with what is left
            const size_t left = end - pos;
            if (buf.size() < left) buf.resize(left + ((size_t)1 << 20));
            if (left) memcpy(buf.data(), view + pos, left);
            view = nullptr; held.reset();
            pos = 0; end = left;
        }
        if (c.ext) gpu_copied++;
        if (c.fallback) {  // the worker met an ordinary gzip member and has left: zlib goes on from zin[zin_pos..)
            if (worker.joinable()) worker.join();
            started = false; bgzf = false; threaded = false;
            inflateReset(&zs);
            return refill();
        }
        if (pos > 0 && pos == end) { pos = end = 0; }
        if (buf.size() - end < c.len) {
            if (pos > 0) { memmove(buf.data(), buf.data() + pos, end - pos); end -= pos; pos = 0; }
            if (buf.size() - end < c.len) buf.resize(end + c.len + ((size_t)1 << 20));
        }
        if (c.len) {
            const uint8_t *from = (c.ext ? c.ext : c.data.data()) + c.off;
            // (a GPU window is half a gigabyte of text: a few threads copy it)
            const int nt = c.len >= ((size_t)64 << 20) ? std::min(4, std::max(1, threads)) : 1;
            if (nt == 1) memcpy(buf.data() + end, from, c.len);
            else {
                std::vector<std::thread> pool;
                uint8_t *to = buf.data() + end;
                for (int t = 0; t < nt; t++) {
                    const size_t lo = c.len * (size_t)t / nt, hi = c.len * (size_t)(t + 1) / nt;
                    pool.emplace_back([=] { memcpy(to + lo, from + lo, hi - lo); });
                }
                for (std::thread &th : pool) th.join();
            }
        }
        end += c.len;
        c.hold.reset();   // (copied: the window is the inflater's again)
        if (c.last) text_eof = true;
        if (!c.data.empty()) {
            std::lock_guard<std::mutex> lk(mu);
            if (spare.size() < 2 * queue_cap) spare.push_back(std::move(c.data));
        }
        return true;
    }
    // total size of the BGZF block starting at p (0 if p does not start one or n < 18)
    static size_t bgzf_block_size(const uint8_t *p, size_t n) {
        if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
        const size_t xlen = p[10] | ((size_t)p[11] << 8);
        if (xlen < 6 || p[12] != 'B' || p[13] != 'C' || p[14] != 2 || p[15] != 0) return 0;
        return ((size_t)p[16] | ((size_t)p[17] << 8)) + 1;
    }
    // Inflate as many whole BGZF blocks as the compressed window holds, side by side, straight
    // into buf[end..].  Returns 1 bytes produced (or clean end), 0 fall back to the sequential path
    // (the window does not start with a BGZF block), -1 error.
    // `dst` receives the text of the window, *text_len bytes of it (a recycled buffer: it may be longer); *at_end is set when
    // the file is exhausted.  The compressed window is a 32 MB stretch of the mapped file (a window read() into `zin` where the
    // file could not be mapped): enough blocks for every host thread, and no serial copy in front of them.
    size_t bgzf_map_pos = 0;
    int bgzf_window(std::vector<uint8_t> &dst, size_t *text_len, bool *at_end) {
        *text_len = 0;
        for (;;) {
            const uint8_t *zbase;
            size_t zcap;
            if (map) {
                bgzf_map_pos += zin_pos; zin_pos = 0;
                zbase = map + bgzf_map_pos;
                zin_end = std::min<size_t>((size_t)32 << 20, map_size - bgzf_map_pos);
                zcap = zin_end < ((size_t)32 << 20) ? zin_end + 1 : zin_end;  // (a window cut short by the end of the file is not "full")
                raw_eof = bgzf_map_pos + zin_end >= map_size;
            } else {
                // top up the compressed window
                if (zin_pos > 0 && zin_pos < zin_end) memmove(zin.data(), zin.data() + zin_pos, zin_end - zin_pos);
                zin_end -= zin_pos; zin_pos = 0;
                while (!raw_eof && zin_end < zin.size()) {
                    const ssize_t n = ::read(fd, zin.data() + zin_end, zin.size() - zin_end);
                    if (n < 0) { err = std::string("read: ") + strerror(errno); return -1; }
                    if (n == 0) raw_eof = true;
                    zin_end += (size_t)(n > 0 ? n : 0);
                }
                zbase = zin.data();
                zcap = zin.size();
            }
            if (zin_end == 0) { *at_end = true; dst.clear(); return 1; }
            struct Blk { size_t in, in_len, out, out_len; uint32_t crc; size_t whole, whole_len; };
            std::vector<Blk> blks;
            size_t p = 0, out_total = 0;
            while (p < zin_end) {
                if (bgzf_seen && zbase[p] == 0) { p++; continue; }  // zero padding between members
                const size_t bs = bgzf_block_size(zbase + p, zin_end - p);
                if (bs == 0) break;            // not a BGZF block (or its header is cut off)
                if (bs < 26 || p + bs > zin_end) { if (bs < 26) { err = "corrupt BGZF block"; return -1; } break; }
                const uint8_t *b = zbase + p;
                const size_t xlen = b[10] | ((size_t)b[11] << 8), hdr = 12 + xlen;
                if (hdr + 8 > bs) { err = "corrupt BGZF block"; return -1; }
                const uint32_t crc = (uint32_t)b[bs - 8] | ((uint32_t)b[bs - 7] << 8) | ((uint32_t)b[bs - 6] << 16) | ((uint32_t)b[bs - 5] << 24);
                const size_t isize = (size_t)b[bs - 4] | ((size_t)b[bs - 3] << 8) | ((size_t)b[bs - 2] << 16) | ((size_t)b[bs - 1] << 24);
                if (isize > (1u << 16)) { err = "corrupt BGZF block"; return -1; }
                blks.push_back(Blk{p + hdr, bs - hdr - 8, out_total, isize, crc, p, bs});
                out_total += isize;
                p += bs;
                bgzf_seen = true;
            }
            if (blks.empty()) {
                if (p > 0) { zin_pos = p; continue; }  // only padding so far: drop it and look again
                if (zin_end >= 18 && bgzf_block_size(zbase, zin_end) == 0) {  // an ordinary gzip member follows
                    if (map) {  // the sequential path reads the file itself, from here
                        if (lseek(fd, (off_t)bgzf_map_pos, SEEK_SET) < 0) { err = std::string("lseek: ") + strerror(errno); return -1; }
                        zin_pos = zin_end = 0; raw_eof = false;
                    }
                    return 0;
                }
                if (raw_eof) { err = "truncated gzip file"; return -1; }
                if (zin_end >= zcap) { err = "corrupt BGZF block"; return -1; }
                continue;  // header or block cut off by the window: read more
            }
            dst = take_buffer(out_total);  // (recycled: a fresh vector of this size is zero-filled by one thread, every window)
            uint8_t *out = dst.data();
            std::atomic<size_t> next{0};
            std::atomic<bool> ok{true};
            const bool own = !(getenv("TBK_INFLATE") && strcmp(getenv("TBK_INFLATE"), "zlib") == 0);
            auto work = [&]() {
                z_stream z;
                memset(&z, 0, sizeof z);
                if (inflateInit2(&z, -15) != Z_OK) { ok.store(false); return; }
                // own decoder: a block (a whole gzip member) is inflated into a private buffer with the
                // slack the decoder's wide stores need, then copied to its place beside its neighbours
                std::vector<uint8_t> scratch(own ? (1u << 16) + 1024 : 0);
                TbkInflate blk_inf;
                for (size_t i; (i = next.fetch_add(1)) < blks.size() && ok.load();) {
                    const Blk &k = blks[i];
                    bool good;
                    if (own && k.out_len) {
                        blk_inf.reset(zbase + k.whole, k.whole_len);
                        size_t pos = 0;
                        const TbkInflate::Status st = blk_inf.run(scratch.data(), &pos, scratch.size(), 0);
                        good = st == TbkInflate::MEMBER_DONE && pos == k.out_len;
                        if (good) memcpy(out + k.out, scratch.data(), k.out_len);
                    } else {
                        inflateReset(&z);
                        z.next_in = const_cast<uint8_t *>(zbase + k.in); z.avail_in = (uInt)k.in_len;
                        z.next_out = out + k.out; z.avail_out = (uInt)k.out_len;
                        const int rc = k.out_len ? inflate(&z, Z_FINISH) : Z_STREAM_END;  // an empty block (the end-of-file marker) has nothing to inflate
                        good = rc == Z_STREAM_END && z.avail_out == 0;
                    }
                    if (!good || tbk_crc32(0u, out + k.out, k.out_len) != k.crc) ok.store(false);
                }
                inflateEnd(&z);
            };
            const int nt = (int)std::min<size_t>((size_t)threads, blks.size());
            std::vector<std::thread> pool;
            for (int t = 1; t < nt; t++) pool.emplace_back(work);
            work();
            for (std::thread &t : pool) t.join();
            if (!ok.load()) { err = "inflate: corrupt BGZF block"; return -1; }
            zin_pos = p;
            *text_len = out_total;
            if (out_total) return 1;
            // only empty blocks (the BGZF end-of-file marker): go on
        }
    }
    ~LineSource() { close_all(); }
    void close_all() {
        if (started) {
            std::shared_ptr<GpuWindows> w;
            { std::lock_guard<std::mutex> lk(mu); stop = true; w = windows; }
            cv.notify_all();
            if (w) { { std::lock_guard<std::mutex> lk(w->mu); w->abandoned = true; } w->cv.notify_all(); }
            if (worker.joinable()) worker.join();
            started = false;
            ready.clear();   // (windows still queued go back)
            if (gpu_windows && (getenv("TBK_PINFLATE_TIMING") || getenv("TBK_WRITE_TIMING")))
                fprintf(stderr, "tbk-gpu-bgzf %llu windows, %llu blocks inflated on device %d (%llu parsed in place, %llu copied); the worker: staging %.3f s, waiting for the device %.3f s, "
                        "for a free window %.3f s; the parser: waiting for a window %.3f s, copying %.1f MB it had left in front of the next %.3f s\n",
                        (unsigned long long)gpu_windows, (unsigned long long)gpu_blocks, gpu_device, (unsigned long long)gpu_in_place, (unsigned long long)gpu_copied,
                        gpu_stage_s, gpu_wait_s, gpu_slot_wait_s, chunk_wait_s, left_bytes / 1e6, left_copy_s);
        }
        view = nullptr; held.reset(); windows.reset();
        if (map) { munmap((void *)map, map_size); map = nullptr; }
        if (zs_live) { inflateEnd(&zs); zs_live = false; }
        if (fd >= 0) { ::close(fd); fd = -1; }
    }
    // append decoded bytes at buf[end..]; returns false on error; sets text_eof at the end
    bool refill() {
        if (text_eof) return true;
        if (threaded) return refill_fast();  // a worker thread inflates, this one parses
        if (pos > 0 && pos == end) { pos = end = 0; }
        if (buf.size() - end < (1u << 16)) {
            if (pos > (buf.size() >> 1)) {  // compact
                memmove(buf.data(), buf.data() + pos, end - pos);
                end -= pos; pos = 0;
            } else {
                buf.resize(buf.size() * 2);
            }
        }
        if (!gz) {
            ssize_t n = ::read(fd, buf.data() + end, buf.size() - end);
            if (n < 0) { err = std::string("read: ") + strerror(errno); return false; }
            if (n == 0) { text_eof = true; return true; }
            end += (size_t)n;
            return true;
        }
        for (;;) {
            if (zin_pos == zin_end && !raw_eof) {
                ssize_t n = ::read(fd, zin.data(), zin.size());
                if (n < 0) { err = std::string("read: ") + strerror(errno); return false; }
                if (n == 0) raw_eof = true;
                zin_pos = 0; zin_end = (size_t)(n > 0 ? n : 0);
            }
            if (zin_pos == zin_end && raw_eof) { text_eof = true; return true; }
            zs.next_in = zin.data() + zin_pos;
            zs.avail_in = (uInt)(zin_end - zin_pos);
            zs.next_out = buf.data() + end;
            const size_t room = buf.size() - end;
            zs.avail_out = (uInt)std::min<size_t>(room, 1u << 30);
            const uInt out_before = zs.avail_out;
            int rc = inflate(&zs, Z_NO_FLUSH);
            zin_pos = zin_end - zs.avail_in;
            const size_t produced = out_before - zs.avail_out;
            end += produced;
            if (rc == Z_STREAM_END) {
                // another gzip member may follow (Python's GzipFile reads them all), possibly after
                // zero padding (which it skips)
                for (;;) {
                    if (zin_pos == zin_end) {
                        if (raw_eof) { text_eof = true; break; }
                        ssize_t n = ::read(fd, zin.data(), zin.size());
                        if (n < 0) { err = std::string("read: ") + strerror(errno); return false; }
                        if (n == 0) { raw_eof = true; text_eof = true; break; }
                        zin_pos = 0; zin_end = (size_t)n;
                    }
                    if (zin[zin_pos] == 0) { zin_pos++; continue; }
                    inflateReset(&zs);
                    break;
                }
                if (produced || text_eof) return true;
                continue;
            }
            if (rc != Z_OK && rc != Z_BUF_ERROR) { err = std::string("inflate: ") + (zs.msg ? zs.msg : "corrupt gzip data"); return false; }
            if (produced) return true;
            if (rc == Z_BUF_ERROR && zin_pos == zin_end && raw_eof) { err = "truncated gzip file"; return false; }
        }
    }
    // next line; `len` excludes the terminator, `term` says whether there was one
    // returns 1 line, 0 end of text, -1 error
    int next(const uint8_t *&p, size_t &len, bool &term) {
        for (;;) {
            if (skip_lf) {
                if (pos == end) {
                    if (text_eof) { skip_lf = false; return 0; }
                    if (!refill()) return -1;
                    continue;
                }
                if (text()[pos] == '\n') pos++;
                skip_lf = false;
            }
            const uint8_t *base = text();
            size_t i = pos;
            // scan for '\n' or '\r'
            while (i < end) {
                const uint8_t *nl = (const uint8_t *)memchr(base + i, '\n', end - i);
                const size_t stop = nl ? (size_t)(nl - base) : end;
                const uint8_t *cr = (const uint8_t *)memchr(base + i, '\r', stop - i);
                if (cr) { i = (size_t)(cr - base); break; }
                i = stop;
                break;
            }
            if (i < end) {
                p = base + pos; len = i - pos; term = true;
                if (base[i] == '\r') {
                    if (i + 1 < end) pos = i + 1 + (base[i + 1] == '\n' ? 1 : 0);
                    else { pos = i + 1; skip_lf = true; }
                } else {
                    pos = i + 1;
                }
                return 1;
            }
            if (text_eof) {
                if (pos == end) return 0;
                p = base + pos; len = end - pos; term = false;
                pos = end;
                return 1;
            }
            if (!refill()) return -1;
        }
    }
};

// =======================================================================================
// batch
// =======================================================================================
struct FastqRec { uint64_t head, seq, plus, qual, end; uint32_t name_len; };  // line starts of a regular FASTQ record; `end` = start of the next record

// The reader's mapping of a plain input file.  The reader and every borrowed batch made from it hold a reference: the
// mapping goes with its last holder, so a borrowed batch outliving its reader (written, or refilled from another
// reader, after tbk_fastx_close) still points into mapped text, and release_borrowed never touches memory that has
// been given to somebody else.
struct FastxMapping {
    const uint8_t *p = nullptr;
    size_t n = 0;
    FastxMapping(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    FastxMapping(const FastxMapping &) = delete;
    FastxMapping &operator=(const FastxMapping &) = delete;
    ~FastxMapping() { if (p) munmap((void *)p, n); }
};

struct tbk_fastx_batch {
    uint8_t *bases = nullptr;  // pinned (tbk_pin_alloc_) when a device is present, else malloc
    size_t bases_cap = 0;
    bool pinned = false;
    std::vector<uint64_t> base_off{0};
    std::vector<uint8_t> names;
    std::vector<uint64_t> name_off{0};
    std::vector<uint8_t> quals;
    std::vector<uint64_t> qual_off{0};
    std::vector<uint8_t> has_qual;
    uint64_t n_bases = 0;
    // record under construction
    uint64_t rec_seq0 = 0, rec_qual0 = 0;
    // The packed transfer form of `bases` (tbk_pack.cpp: 16 bases to a word + the exceptions), made by the
    // reader itself when asked (tbk_fastx_set_packing): the chunk-parallel scan packs a record's chunks right
    // after it copied the record, from its own cache, so the classify stage copies a quarter of the bytes
    // and nobody reads the batch a second time.  `fused` = every base of the batch came through that scan.
    bool want_packed = false, fused = false, packed_ok = false;
    uint32_t *codes = nullptr;
    size_t codes_cap = 0;
    bool codes_pinned = false;
    std::vector<TbkExc> exc;
    std::vector<uint32_t> exc_chunk;
    std::vector<uint16_t> exc_mask;
    // A borrowed batch (tbk_fastx_set_borrowing): its records lie in the reader's mapping of the input file and
    // were not copied - `recs` says where, names / offsets / has_qual / the packed form are filled as usual, `bases`
    // and `quals` are not.  The bin writer writes such records from the mapping.  Valid until the reader is closed.
    bool borrowed = false;
    const uint8_t *text = nullptr;
    std::vector<FastqRec> recs;
    std::shared_ptr<void> hold;  // keeps `text` where it is while this batch refers to it: the reader's mapping of the file, or a window of the GPU inflater
    bool text_mapped = false;    // `text` is the file's mapping (its pages are let go when the batch is refilled)

    ~tbk_fastx_batch() {
        release();
        if (codes) { if (codes_pinned) tbk_pin_free_(codes); else free(codes); }
    }
    void release() {  // the bases buffer (reserve_bases replaces it when it grows)
        if (bases) { if (pinned) tbk_pin_free_(bases); else free(bases); }
        bases = nullptr; bases_cap = 0;
    }
    bool reserve_codes(size_t need_chunks) {
        if (need_chunks <= codes_cap) return true;
        const size_t cap = std::max<size_t>(need_chunks + need_chunks / 2, (size_t)1 << 18);
        uint32_t *nc = nullptr;
        bool np = false;
        if (pinned && (nc = (uint32_t *)tbk_pin_alloc_(cap * sizeof(uint32_t))) != nullptr) np = true;
        else nc = (uint32_t *)malloc(cap * sizeof(uint32_t));
        if (!nc) return false;
        if (codes && codes_cap) memcpy(nc, codes, codes_cap * sizeof(uint32_t));
        if (codes) { if (codes_pinned) tbk_pin_free_(codes); else free(codes); }
        codes = nc; codes_cap = cap; codes_pinned = np;
        return true;
    }
    // Pinned when a device is there (the batch then goes to the GPU without a staging copy).
    // Pinning is tried once per process: a failing attempt (no device) is slow.
    bool reserve_bases(size_t need) {
        if (need <= bases_cap) return true;
        static std::atomic<int> pin_state{0};  // 0 unknown, 1 works, -1 does not
        size_t cap = bases_cap ? std::max<size_t>(need + need / 2, (size_t)1 << 22) : std::max<size_t>(need, (size_t)1 << 22);
        uint8_t *nb = nullptr;
        bool np = false;
        if (pin_state.load() >= 0) {
            if ((nb = (uint8_t *)tbk_pin_alloc_(cap)) != nullptr) { np = true; pin_state.store(1); }
            else pin_state.store(-1);
        }
        if (!nb) nb = (uint8_t *)malloc(cap);
        if (!nb) return false;
        if (n_bases) memcpy(nb, bases, n_bases);
        release();
        bases = nb; bases_cap = cap; pinned = np;
        return true;
    }
    void clear() {
        base_off.assign(1, 0); names.clear(); name_off.assign(1, 0); quals.clear(); qual_off.assign(1, 0);
        has_qual.clear(); n_bases = 0; rec_seq0 = rec_qual0 = 0;
        fused = want_packed; packed_ok = false;
        exc.clear(); exc_chunk.clear(); exc_mask.clear();
        borrowed = false; text = nullptr; recs.clear(); hold.reset(); text_mapped = false;
    }
    void begin(const uint8_t *name, size_t n) {
        names.insert(names.end(), name, name + n);
        rec_seq0 = n_bases; rec_qual0 = quals.size();
    }
    bool seq(const uint8_t *p, size_t n) {
        if (!reserve_bases(n_bases + n + 16)) return false;
        memcpy(bases + n_bases, p, n);
        n_bases += n;
        fused = false;  // the sequential machine's bytes are packed in one go when the batch is complete
        return true;
    }
    void qual(const uint8_t *p, size_t n) { quals.insert(quals.end(), p, p + n); }
    void finish(bool with_qual) {
        if (!with_qual) quals.resize(rec_qual0);
        name_off.push_back(names.size());
        base_off.push_back(n_bases);
        qual_off.push_back(quals.size());
        has_qual.push_back(with_qual ? 1 : 0);
    }
    uint64_t n_reads() const { return has_qual.size(); }
};

// =======================================================================================
// reader: the record state machine of seq.py:45-83
// =======================================================================================
// ---- chunk-parallel scan of regular 4-line FASTQ ------------------------------------------------
// Plain (uncompressed) FASTQ whose records are "@header / sequence / +... / quality of the same
// length", every line ended by a single '\n': what sequencers and basecallers write.  For such a
// record the state machine below has no choices to make, so records can be found by several threads
// at once and copied into the batch by several threads at once.  The file is mapped; a call takes a
// window of it, cuts the window into one piece per host thread, lets every thread but the first
// guess a record start in its piece, index the records that begin in the piece (checking each one
// for regularity) and report where the last one ended.  The guesses are then checked, not trusted:
// piece t+1 is accepted only if it began exactly where piece t ended, and piece 0 begins where the
// previous call stopped, in the state machine's SEEK state - so by induction the accepted records are
// exactly those the sequential machine would have produced.  The first record that is not regular
// (CR anywhere, a multi-line sequence, a quality line of another length, a blank line, '>' records,
// a last line without newline ...) ends this mode for good: the sequential machine takes over at
// that record's first byte.
struct RegularScan {
    const uint8_t *map = nullptr;
    size_t size = 0, pos = 0;   // pos: a record starts here (or pos == size)
    bool active = false;
    bool inflated = false;        // the same scan over inflated text (gzip input), see regular_next_inflated
    double bytes_per_base = 2.2;  // estimate used to size a call's window
};

// first '\n' or '\r' at or after p (or end)
__attribute__((target("avx2"))) static const uint8_t *find_eol_avx2(const uint8_t *p, const uint8_t *end) {
    const __m256i nl = _mm256_set1_epi8('\n'), cr = _mm256_set1_epi8('\r');
    while (p + 32 <= end) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)p);
        const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(v, nl), _mm256_cmpeq_epi8(v, cr)));
        if (m) return p + __builtin_ctz(m);
        p += 32;
    }
    while (p < end && *p != '\n' && *p != '\r') p++;
    return p;
}
static const uint8_t *find_eol_scalar(const uint8_t *p, const uint8_t *end) {
    while (p < end && *p != '\n' && *p != '\r') p++;
    return p;
}
static inline const uint8_t *find_eol(const uint8_t *p, const uint8_t *end) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? find_eol_avx2(p, end) : find_eol_scalar(p, end);
}

// The record starting at `at`, if it is regular and complete: fills rec and returns true.
static bool regular_record(const uint8_t *d, size_t size, size_t at, FastqRec &rec) {
    const uint8_t *end = d + size;
    if (at >= size || d[at] != '@') return false;
    const uint8_t *e0 = find_eol(d + at, end);
    if (e0 == end || *e0 != '\n') return false;
    const uint8_t *s = e0 + 1;
    if (s < end && (*s == '@' || *s == '+' || *s == '>')) return false;  // would be taken for a header or the separator
    const uint8_t *e1 = find_eol(s, end);
    if (e1 == end || *e1 != '\n') return false;
    const uint8_t *pl = e1 + 1;
    if (pl >= end || *pl != '+') return false;
    const uint8_t *e2 = find_eol(pl, end);
    if (e2 == end || *e2 != '\n') return false;
    const uint8_t *q = e2 + 1;
    const size_t len = (size_t)(e1 - s);
    if ((size_t)(end - q) < len + 1 || q[len] != '\n') return false;   // quality: same length, then newline
    if (find_eol(q, q + len) != q + len) return false;                  // no line end inside it
    rec.head = at; rec.seq = (uint64_t)(s - d); rec.plus = (uint64_t)(pl - d); rec.qual = (uint64_t)(q - d);
    rec.end = rec.qual + len + 1;
    // name = header without '@', up to the first space (seq.py:61)
    const uint8_t *name = d + at + 1;
    const size_t hl = (size_t)(e0 - name);
    const uint8_t *sp = (const uint8_t *)memchr(name, ' ', hl);
    rec.name_len = (uint32_t)(sp ? (size_t)(sp - name) : hl);
    return true;
}

// Why the record at `at` is not a regular, complete record: true = the text at hand ends inside it (more text
// may complete it), false = it is irregular whatever follows (the sequential machine's).
static bool regular_record_cut_off(const uint8_t *d, size_t size, size_t at) {
    const uint8_t *end = d + size;
    if (at >= size) return true;
    if (d[at] != '@') return false;
    const uint8_t *e0 = find_eol(d + at, end);
    if (e0 == end) return true;
    if (*e0 != '\n') return false;
    const uint8_t *s = e0 + 1;
    if (s == end) return true;
    if (*s == '@' || *s == '+' || *s == '>') return false;
    const uint8_t *e1 = find_eol(s, end);
    if (e1 == end) return true;
    if (*e1 != '\n') return false;
    const uint8_t *pl = e1 + 1;
    if (pl >= end) return true;
    if (*pl != '+') return false;
    const uint8_t *e2 = find_eol(pl, end);
    if (e2 == end) return true;
    if (*e2 != '\n') return false;
    const uint8_t *q = e2 + 1;
    const size_t len = (size_t)(e1 - s), have = (size_t)(end - q);
    if (have < len + 1) return find_eol(q, end) == end;  // a line end inside the quality that is there: irregular already
    return false;
}

struct tbk_fastx_reader {
    LineSource src;
    RegularScan scan;
    enum { SEEK, SEQ, QUAL, DONE } state = SEEK;
    std::string pending_name;   // header already consumed for the record whose sequence comes next
    bool have_pending = false;
    uint64_t seq_len = 0;       // current record (QUAL state)
    int64_t qual_have = 0;
    std::shared_ptr<FastxMapping> mapping;  // owner of scan.map (shared with the borrowed batches made from it)
    bool packing = false;       // batches also carry the packed transfer form of their bases
    bool borrowing = false;     // batches of the mapped plain file reference its text instead of copying it (needs packing)
};

static void name_of(const uint8_t *body, size_t n, const uint8_t *&np, size_t &nn) {
    // header without its first character, cut at the first space (seq.py:61)
    np = body + (n ? 1 : 0);
    nn = n ? n - 1 : 0;
    const uint8_t *sp = (const uint8_t *)memchr(np, ' ', nn);
    if (sp) nn = (size_t)(sp - np);
}

extern "C" int tbk_fastx_open(const char *path, tbk_fastx_reader **out) {
    if (!out || !path) return ffail(TBK_ERR_INVALID, "NULL argument");
    *out = nullptr;
    const size_t n = strlen(path);
    const bool gz = n >= 3 && strcmp(path + n - 3, ".gz") == 0;  // seq.py:88: by file name
    tbk_fastx_reader *r = new tbk_fastx_reader();
    if (!r->src.open_path(path, gz)) {
        std::string e = r->src.err;
        r->src.close_all();
        delete r;
        return ffail(TBK_ERR_IO, "%s", e.c_str());
    }
    // plain FASTQ: records are found and copied by several threads while they stay regular
    const char *scan_env = getenv("TBK_FASTQ_SCAN");
    struct stat st;
    if (!gz && !(scan_env && *scan_env == '0') && fstat(r->src.fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, r->src.fd, 0);
        if (m != MAP_FAILED) {
            r->mapping = std::make_shared<FastxMapping>((const uint8_t *)m, (size_t)st.st_size);
            r->scan.map = (const uint8_t *)m;
            r->scan.size = (size_t)st.st_size;
            r->scan.active = r->scan.map[0] == '@';
        }
    }
    // the same scan over inflated text: opt-in (TBK_INFLATED_SCAN=1).  On 16 host threads it is ~5 % slower
    // than the sequential machine behind the inflater (3.4 against 3.6 GB/s): the inflating threads
    // are the limit and the scan's threads compete with them; it pays where cores are plentiful.
    const char *inflated_env = getenv("TBK_INFLATED_SCAN");
    r->scan.inflated = gz && inflated_env && *inflated_env == '1' && !(scan_env && *scan_env == '0');
    *out = r;
    return TBK_OK;
}

extern "C" void tbk_fastx_close(tbk_fastx_reader *r) {
    if (!r) return;
    r->scan.map = nullptr;
    r->mapping.reset();  // (unmapped now, or with the last borrowed batch that still refers to it)
    r->src.close_all();
    delete r;
}

extern "C" int tbk_fastx_batch_create(tbk_fastx_batch **out) {
    if (!out) return ffail(TBK_ERR_INVALID, "NULL argument");
    *out = new tbk_fastx_batch();
    return TBK_OK;
}

extern "C" void tbk_fastx_batch_destroy(tbk_fastx_batch *b) { delete b; }

// `n` bases of a borrowed batch's stream (the records' sequences back to back) from stream position `at`
static void borrowed_bases(const tbk_fastx_batch *b, uint64_t at, size_t n, uint8_t *out) {
    size_t i = (size_t)(std::upper_bound(b->base_off.begin(), b->base_off.end(), at) - b->base_off.begin());
    i = i ? i - 1 : 0;  // the record that holds position `at` (empty records before it share its offset and are skipped)
    for (; n && i < b->recs.size(); i++) {
        const uint64_t lo = b->base_off[i], hi = b->base_off[i + 1];
        if (at >= hi) continue;
        const size_t k = (size_t)std::min<uint64_t>(n, hi - at);
        memcpy(out, b->text + b->recs[i].seq + (at - lo), k);
        out += k; at += k; n -= k;
    }
}

// One batch by the chunk-parallel scan (see RegularScan) of the text d[pos..size): a mapped file
// (final: nothing follows d[size-1]) or the inflated text at hand (more may follow: a record cut off by
// the end of the window is not irregular, only incomplete).  Leaves the batch empty when the record
// at `pos` is not regular and complete.  *new_pos = where the next record starts; *leave = every
// record that chained up was taken and the one after them is not regular (meaningful when final).
static int regular_window(RegularScan &sc, const uint8_t *d, const size_t size, const size_t pos, tbk_fastx_batch *b,
                          uint64_t max_bases, uint64_t max_reads, size_t *new_pos_out, bool *leave, bool may_borrow = false) {
    *new_pos_out = pos;
    *leave = false;
    if (pos >= size) return TBK_OK;
    const size_t left = size - pos;
    size_t want = left;
    if (max_bases != ~0ull) {
        const double est = (double)max_bases * sc.bytes_per_base * 1.02 + (double)((size_t)2 << 20);
        if (est < (double)left) want = (size_t)est;
    }
    const size_t piece_min = (size_t)4 << 20;
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, tbk_host_threads()), want / piece_min));
    struct Piece { size_t lo = 0, hi = 0, begin = 0, end = 0; bool synced = false, bad = false; std::vector<FastqRec> recs; };
    std::vector<Piece> pieces((size_t)nt);
    for (int t = 0; t < nt; t++) {
        pieces[(size_t)t].lo = pos + want * (size_t)t / (size_t)nt;
        pieces[(size_t)t].hi = pos + want * (size_t)(t + 1) / (size_t)nt;
    }
    auto work = [&](int t) {
        Piece &pc = pieces[(size_t)t];
        size_t at = pc.lo;
        FastqRec rec, nxt;
        if (t == 0) {
            pc.synced = true;
        } else {
            // guess: the first line start in the piece where two regular records follow one another
            // (the piece may begin on a line start itself: then that line is the first candidate)
            const uint8_t *e = d[pc.lo - 1] == '\n' ? d + pc.lo - 1 : find_eol(d + pc.lo, d + size);
            size_t ls = (size_t)(e - d) + 1;
            for (int tries = 0; tries < 12 && ls < pc.hi && !pc.synced; tries++) {
                if (e == d + size || *e != '\n') break;  // a CR: not this mode's business
                if (regular_record(d, size, ls, rec) && (rec.end == size || regular_record(d, size, rec.end, nxt))) {
                    at = ls;
                    pc.synced = true;
                    break;
                }
                e = find_eol(d + ls, d + size);
                ls = (size_t)(e - d) + 1;
            }
            if (!pc.synced) return;
        }
        pc.begin = at;
        pc.recs.reserve((pc.hi - pc.lo) / 4096 + 16);
        while (at < pc.hi) {
            if (!regular_record(d, size, at, rec)) { pc.bad = true; break; }
            pc.recs.push_back(rec);
            at = (size_t)rec.end;
        }
        pc.end = at;
    };
    static const bool timing = getenv("TBK_SCAN_TIMING") != nullptr;
    const auto t_a = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
        work(0);
        for (std::thread &th : pool) th.join();
    }
    const auto t_b = std::chrono::steady_clock::now();
    // accept the pieces that chain up: each must begin exactly where the one before it ended
    int n_ok = 1;
    while (n_ok < nt && !pieces[(size_t)n_ok - 1].bad && pieces[(size_t)n_ok].synced && pieces[(size_t)n_ok].begin == pieces[(size_t)n_ok - 1].end) n_ok++;
    // records of this batch: up to and including the one that reaches a limit
    uint64_t n_reads = 0, n_bases = 0, n_name = 0;
    int last_piece = 0;
    size_t last_idx = 0;
    bool full = false;
    for (int t = 0; t < n_ok && !full; t++) {
        const std::vector<FastqRec> &rs = pieces[(size_t)t].recs;
        for (size_t i = 0; i < rs.size(); i++) {
            n_reads++;
            n_bases += rs[i].plus - 1 - rs[i].seq;
            n_name += rs[i].name_len;
            last_piece = t; last_idx = i + 1;
            if (n_reads >= max_reads || n_bases >= max_bases) { full = true; break; }
        }
    }
    if (n_reads == 0) return TBK_OK;  // the record at pos is not regular (or not all there)
    // flat view of the chosen records, then parallel copies into the batch's arrays
    std::vector<const FastqRec *> chosen;
    chosen.reserve((size_t)n_reads);
    for (int t = 0; t <= last_piece; t++) {
        const std::vector<FastqRec> &rs = pieces[(size_t)t].recs;
        const size_t upto = t == last_piece ? last_idx : rs.size();
        for (size_t i = 0; i < upto; i++) chosen.push_back(&rs[i]);
    }
    // appended behind what the batch holds already (inflated text comes window by window)
    const size_t n0 = (size_t)b->n_reads(), on0 = b->names.size(), oq0 = b->quals.size();
    const uint64_t ob0 = b->n_bases;
    const bool pack = b->want_packed && b->fused;
    // Borrowing (d is the reader's mapping of the file, the batch is empty): the records stay where they are.  Their
    // bases are packed straight from the mapping, names and offsets are filled as usual, sequences and qualities
    // are not copied - the bin writer writes the records from the mapping.
    const bool borrow = may_borrow && pack && n0 == 0 && ob0 == 0;
    if (!borrow && !b->reserve_bases((size_t)(ob0 + n_bases) + 16)) return ffail(TBK_ERR_NOMEM, "out of memory sizing a read batch");
    if (pack && !b->reserve_codes((size_t)((ob0 + n_bases + 15) / 16) + 1)) return ffail(TBK_ERR_NOMEM, "out of memory sizing a read batch");
    b->base_off.resize(n0 + (size_t)n_reads + 1);
    b->name_off.resize(n0 + (size_t)n_reads + 1);
    b->qual_off.resize(n0 + (size_t)n_reads + 1);
    b->has_qual.resize(n0 + (size_t)n_reads, 1);
    b->names.resize(on0 + (size_t)n_name);
    if (borrow) { b->borrowed = true; b->text = d; b->recs.resize((size_t)n_reads); }
    else b->quals.resize(oq0 + (size_t)n_bases);
    uint64_t ob = ob0, on = on0, oq = oq0;
    for (size_t i = 0; i < chosen.size(); i++) {
        b->base_off[n0 + i] = ob; b->qual_off[n0 + i] = oq; b->name_off[n0 + i] = on;
        const uint64_t len = chosen[i]->plus - 1 - chosen[i]->seq;
        ob += len; oq += len;
        on += chosen[i]->name_len;
    }
    b->base_off[n0 + chosen.size()] = ob; b->qual_off[n0 + chosen.size()] = oq; b->name_off[n0 + chosen.size()] = on;
    b->n_bases = ob;
    {
        const int ct = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, tbk_host_threads()), (size_t)n_bases / ((size_t)8 << 20)));
        // records [cut(t), cut(t + 1)) are thread t's: split where the bases split evenly (reads differ in length; empty ones have none)
        auto cut = [&](int u) -> size_t {
            if (u <= 0) return 0;
            if (u >= ct) return chosen.size();
            const uint64_t at_b = ob0 + n_bases * (uint64_t)u / (uint64_t)ct;
            const auto first = b->base_off.begin() + (ptrdiff_t)n0;
            return (size_t)(std::lower_bound(first, first + (ptrdiff_t)chosen.size(), at_b) - first);
        };
        std::vector<std::vector<TbkExc>> found(pack ? (size_t)ct : 0);
        auto copy = [&](int t) {
            const size_t first = cut(t), last = cut(t + 1);
            // packing: the 16-base chunks that lie wholly inside this thread's stretch of the stream are its own, and
            // each is packed as soon as the record that completes it has been copied (from this core's cache)
            const uint64_t hi_b = b->base_off[n0 + last];
            uint64_t next_chunk = (b->base_off[n0 + first] + 15) / 16;
            for (size_t i = first; i < last; i++) {
                const FastqRec &rc = *chosen[i];
                const size_t len = (size_t)(rc.plus - 1 - rc.seq);
                if (len) {
                    memcpy(b->bases + b->base_off[n0 + i], d + rc.seq, len);
                    memcpy(b->quals.data() + b->qual_off[n0 + i], d + rc.qual, len);
                }
                if (rc.name_len) memcpy(b->names.data() + b->name_off[n0 + i], d + rc.head + 1, rc.name_len);
                if (pack) {
                    const uint64_t done = std::min(b->base_off[n0 + i + 1], hi_b) / 16;
                    if (done > next_chunk) { tbk_pack_chunk_range_(b->bases, next_chunk, done, b->codes, found[(size_t)t]); next_chunk = done; }
                }
            }
        };
        // the same stretch of a borrowed batch: names copied, bases packed from where they lie.  A chunk is made of
        // the bases of consecutive records: what a record leaves of one waits in `stage` for the next record.
        auto take_borrowed = [&](int t) {
            const size_t first = cut(t), last = cut(t + 1);
            std::vector<TbkExc> &ex = found[(size_t)t];
            uint8_t stage[16];
            unsigned have = 0;
            uint64_t at_b = b->base_off[first];
            unsigned skip = (unsigned)((16 - at_b % 16) % 16);  // a chunk the stretch begins inside is packed below, once whole
            for (size_t i = first; i < last; i++) {
                const FastqRec &rc = *chosen[i];
                b->recs[i] = rc;
                if (rc.name_len) memcpy(b->names.data() + b->name_off[i], d + rc.head + 1, rc.name_len);
                const uint8_t *sq = d + rc.seq;
                size_t n = (size_t)(rc.plus - 1 - rc.seq);
                if (skip && n) { const size_t k = std::min<size_t>(skip, n); sq += k; n -= k; at_b += k; skip -= (unsigned)k; }
                if (have && n) {
                    const size_t k = std::min<size_t>(16 - have, n);
                    memcpy(stage + have, sq, k);
                    have += (unsigned)k; sq += k; n -= k; at_b += k;
                    if (have == 16) { tbk_pack_span_(stage, at_b / 16 - 1, 1, b->codes, ex); have = 0; }
                }
                if (n >= 16) { const size_t whole = n / 16; tbk_pack_span_(sq, at_b / 16, whole, b->codes, ex); sq += 16 * whole; n -= 16 * whole; at_b += 16 * whole; }
                if (n) { memcpy(stage, sq, n); have = (unsigned)n; at_b += n; }
            }
        };
        std::vector<std::thread> pool;
        if (borrow) {
            for (int t = 1; t < ct; t++) pool.emplace_back(take_borrowed, t);
            take_borrowed(0);
        } else {
            for (int t = 1; t < ct; t++) pool.emplace_back(copy, t);
            copy(0);
        }
        for (std::thread &th : pool) th.join();
        if (pack) {
            for (const auto &f : found) b->exc.insert(b->exc.end(), f.begin(), f.end());
            // a chunk that holds the boundary between two threads' stretches (or the start of this window, which
            // the previous window left as its partial last chunk) belongs to neither: packed here, once whole
            uint64_t last_done = ~0ull;
            for (int u = 0; u < ct; u++) {
                const uint64_t pb = b->base_off[n0 + cut(u)];
                const uint64_t c = pb / 16;
                if (pb % 16 == 0 || c == last_done || 16 * c + 16 > ob) continue;
                if (borrow) {
                    uint8_t stage[16];
                    borrowed_bases(b, 16 * c, 16, stage);
                    tbk_pack_span_(stage, c, 1, b->codes, b->exc);
                } else {
                    tbk_pack_chunk_range_(b->bases, c, c + 1, b->codes, b->exc);
                }
                last_done = c;
            }
        }
    }
    if (timing && n_ok < nt)
        fprintf(stderr, "tbk-scan chain broke at piece %d: prev end %zu bad %d, synced %d begin %zu (lo %zu hi %zu)\n", n_ok, pieces[(size_t)n_ok - 1].end,
                (int)pieces[(size_t)n_ok - 1].bad, (int)pieces[(size_t)n_ok].synced, pieces[(size_t)n_ok].begin, pieces[(size_t)n_ok].lo, pieces[(size_t)n_ok].hi);
    if (timing) {
        const auto t_c = std::chrono::steady_clock::now();
        fprintf(stderr, "tbk-scan window %.1f MB, %d pieces (%d chained), %llu reads: index %.1f ms, select+copy %.1f ms\n", (double)want / 1e6, nt, n_ok,
                (unsigned long long)n_reads, std::chrono::duration<double, std::milli>(t_b - t_a).count(), std::chrono::duration<double, std::milli>(t_c - t_b).count());
    }
    const size_t new_pos = (size_t)chosen.back()->end;
    if (n_bases) sc.bytes_per_base = std::max(1.5, (double)(new_pos - pos) / (double)n_bases);
    *new_pos_out = new_pos;
    // everything that chained up was taken and the last piece stopped at an irregular record
    *leave = !full && pieces[(size_t)n_ok - 1].bad && last_piece == n_ok - 1 && last_idx == pieces[(size_t)n_ok - 1].recs.size();
    return TBK_OK;
}

// The mapped plain file: leaves the batch empty when the record at scan.pos is not regular (the mode is
// then off and the sequential machine goes on from there) or the file is exhausted.
static int regular_next(tbk_fastx_reader *r, tbk_fastx_batch *b, uint64_t max_bases, uint64_t max_reads) {
    RegularScan &sc = r->scan;
    if (sc.pos >= sc.size) return TBK_OK;
    size_t new_pos = sc.pos;
    bool leave = false;
    const int rc = regular_window(sc, sc.map, sc.size, sc.pos, b, max_bases, max_reads, &new_pos, &leave, r->borrowing);
    if (rc) return rc;
    if (b->borrowed) { b->hold = r->mapping; b->text_mapped = true; }
    if (b->n_reads() == 0 || leave) sc.active = false;
    sc.pos = new_pos;
    return TBK_OK;
}

// Inflated text (gzip / bgzip input): the same scan over the decoded bytes at hand, while the machine
// stands between records.  The text is taken as it arrives, in windows of up to 64 MiB appended to the
// batch one after the other (the inflating threads work on the next window meanwhile), until the batch
// is full.  A record cut off by the end of a window waits for more text; a record that is not regular
// ends the mode, and the machine reads on from that byte of the same buffer.
static int regular_next_inflated(tbk_fastx_reader *r, tbk_fastx_batch *b, uint64_t max_bases, uint64_t max_reads) {
    LineSource &src = r->src;
    RegularScan &sc = r->scan;
    // size the pinned sequence buffer once, from the batch limit, instead of growing it
    if (max_bases != ~0ull && max_bases <= ((uint64_t)1 << 34) && b->bases_cap < max_bases)
        if (!b->reserve_bases((size_t)max_bases + ((size_t)max_bases >> 3) + (1 << 20)))
            return ffail(TBK_ERR_NOMEM, "out of memory sizing a read batch");
    size_t window = (size_t)64 << 20, at_least = 0;
    for (;;) {
        const uint64_t have_reads = b->n_reads(), have_bases = b->n_bases;
        if (have_reads >= max_reads || have_bases >= max_bases) return TBK_OK;
        size_t want = window;
        if (max_bases != ~0ull) want = (size_t)std::min<double>((double)window, (double)(max_bases - have_bases) * sc.bytes_per_base * 1.02 + (double)((size_t)1 << 20));
        want = std::max(want, at_least);
        while (!src.text_eof && src.end - src.pos < want)
            if (!src.refill()) return ffail(TBK_ERR_IO, "%s", src.err.c_str());
        if (src.pos == src.end) return TBK_OK;  // nothing left: the machine will say so
        if (src.text()[src.pos] != '@') { sc.inflated = false; return TBK_OK; }
        size_t new_pos = src.pos;
        bool leave = false;
        // a window of the GPU inflater is parsed where it lies: like a plain file's mapping, it can keep the records of a batch (the batch
        // then ends with the window at the latest: what the next window brings goes into the next batch)
        const int rc = regular_window(sc, src.text(), src.end, src.pos, b, max_bases - have_bases, max_reads - have_reads, &new_pos, &leave, r->borrowing && src.window_can_lend());
        if (rc) return rc;
        if (b->n_reads() > have_reads) {
            src.pos = new_pos;
            at_least = 0;
            if (b->borrowed) b->hold = src.held;
            if (leave && src.text_eof) { sc.inflated = false; return TBK_OK; }
            if (b->borrowed) return TBK_OK;
            continue;
        }
        // the record at src.pos: cut off by the end of the text at hand (read on: only then is a longer window worth
        // another scan), or not regular whatever follows (the machine's, at once)
        if (!src.text_eof && src.end - src.pos < ((size_t)64 << 20) && regular_record_cut_off(src.text(), src.end, src.pos)) {
            at_least = src.end - src.pos + ((size_t)8 << 20);
            continue;
        }
        sc.inflated = false;
        return TBK_OK;
    }
}

static int fastx_next_records(tbk_fastx_reader *r, tbk_fastx_batch *b, uint64_t max_bases, uint64_t max_reads);

extern "C" int tbk_fastx_set_packing(tbk_fastx_reader *r, int on) {
    if (!r) return ffail(TBK_ERR_INVALID, "NULL argument");
    r->packing = on != 0;
    return TBK_OK;
}

// BGZF input from here on is inflated on `device` (before the first tbk_fastx_next; a file that is not BGZF, or not mapped, is read as
// before).  TBK_BGZF_INFLATE=cpu keeps the host's threads.
extern "C" int tbk_fastx_set_device(tbk_fastx_reader *r, int device) {
    if (!r) return ffail(TBK_ERR_INVALID, "NULL argument");
    const char *how = getenv("TBK_BGZF_INFLATE");
    if (how && strcmp(how, "cpu") == 0) return TBK_OK;
    if (r->src.started) return ffail(TBK_ERR_STATE, "tbk_fastx_set_device after the first read");
    r->src.gpu_device = device;
    // the host's threads do not inflate now: the chunk-parallel scan has them (TBK_INFLATED_SCAN=0 keeps the sequential machine), and
    // the windows it scans can keep a batch's records (tbk_fastx_set_borrowing)
    const char *scan_env = getenv("TBK_FASTQ_SCAN"), *inflated_env = getenv("TBK_INFLATED_SCAN");
    if (r->src.bgzf && r->src.map && !(scan_env && *scan_env == '0') && !(inflated_env && *inflated_env == '0')) r->scan.inflated = true;
    return TBK_OK;
}
// 1 when the reader's BGZF blocks are (being) inflated on a device
extern "C" int tbk_fastx_inflates_on_device(const tbk_fastx_reader *r) { return r && r->src.bgzf && r->src.gpu_device >= 0 && r->src.map ? 1 : 0; }

extern "C" int tbk_fastx_set_borrowing(tbk_fastx_reader *r, int on) {
    if (!r) return ffail(TBK_ERR_INVALID, "NULL argument");
    r->borrowing = on != 0;
    return TBK_OK;
}

extern "C" int tbk_fastx_batch_borrowed(const tbk_fastx_batch *b) { return b && b->borrowed ? 1 : 0; }

extern "C" int tbk_fastx_batch_gather(const tbk_fastx_batch *b, uint8_t *bases, uint64_t bases_cap, uint8_t *quals, uint64_t quals_cap) {
    if (!b) return ffail(TBK_ERR_INVALID, "NULL argument");
    const uint64_t nb = b->base_off.back(), nq = b->qual_off.back();
    if ((bases && bases_cap < nb) || (quals && quals_cap < nq)) return ffail(TBK_ERR_INVALID, "tbk_fastx_batch_gather: buffer too small (%llu bases, %llu quality bytes)",
                                                                           (unsigned long long)nb, (unsigned long long)nq);
    if (!b->borrowed) {
        if (bases && nb) memcpy(bases, b->bases, (size_t)nb);
        if (quals && nq) memcpy(quals, b->quals.data(), (size_t)nq);
        return TBK_OK;
    }
    for (size_t i = 0; i < b->recs.size(); i++) {  // a regular FASTQ record: sequence and quality on one line each, equally long
        const uint64_t len = b->base_off[i + 1] - b->base_off[i];
        if (bases) memcpy(bases + b->base_off[i], b->text + b->recs[i].seq, (size_t)len);
        if (quals && b->has_qual[i]) memcpy(quals + b->qual_off[i], b->text + b->recs[i].qual, (size_t)(b->qual_off[i + 1] - b->qual_off[i]));
    }
    return TBK_OK;
}

// the batch's packed form: what the scan has packed already plus the last, partial chunk - or, for bytes
// that came through the sequential machine, the whole stream in one go on all host threads
static int finish_packing(tbk_fastx_batch *b) {
    if (!b->want_packed || b->n_reads() == 0) return TBK_OK;
    if (!b->reserve_codes((size_t)((b->n_bases + 15) / 16) + 1)) return ffail(TBK_ERR_NOMEM, "out of memory sizing a read batch");
    if (b->fused) {
        if (b->borrowed) {
            const unsigned valid = (unsigned)(b->n_bases % 16);
            uint8_t tail[16];
            if (valid) { borrowed_bases(b, b->n_bases - valid, valid, tail); tbk_pack_tail_bytes_(tail, valid, b->n_bases / 16, b->codes, b->exc); }
        } else {
            tbk_pack_tail_chunk_(b->bases, b->n_bases, b->codes, b->exc);
        }
        b->exc_chunk.resize(b->exc.size());
        b->exc_mask.resize(b->exc.size());
        for (size_t i = 0; i < b->exc.size(); i++) { b->exc_chunk[i] = b->exc[i].chunk; b->exc_mask[i] = b->exc[i].mask; }
    } else {
        const int rc = tbk_pack_bases_vec(b->bases, b->n_bases, b->codes, b->exc_chunk, b->exc_mask, 0);
        if (rc) return rc;
    }
    b->packed_ok = true;
    return TBK_OK;
}

extern "C" int tbk_fastx_next(tbk_fastx_reader *r, tbk_fastx_batch *b, uint64_t max_bases, uint64_t max_reads) {
    if (!r || !b) return ffail(TBK_ERR_INVALID, "NULL argument");
    b->want_packed = r->packing;
    const int rc = fastx_next_records(r, b, max_bases, max_reads);
    if (rc) return rc;
    return finish_packing(b);
}

extern "C" int tbk_fastx_batch_packed(const tbk_fastx_batch *b, const uint32_t **codes, const uint32_t **exc_chunk, const uint16_t **exc_mask,
                                      uint64_t *n_exc) {
    if (!b || !codes || !exc_chunk || !exc_mask || !n_exc) return ffail(TBK_ERR_INVALID, "NULL argument");
    static const uint32_t none32 = 0;
    static const uint16_t none16 = 0;
    *codes = b->packed_ok ? b->codes : nullptr;
    *exc_chunk = b->packed_ok && !b->exc_chunk.empty() ? b->exc_chunk.data() : &none32;
    *exc_mask = b->packed_ok && !b->exc_mask.empty() ? b->exc_mask.data() : &none16;
    *n_exc = b->packed_ok ? b->exc_chunk.size() : 0;
    return TBK_OK;
}

// A borrowed batch that is being refilled: its records have been written, so the pages of the mapping they lie in are
// let go here, on the reader's thread, a batch at a time - left to tbk_fastx_close (or to the process's exit) the
// 30 GB mapping of a BASELINE-sized input costs 0.3-1 s of tear-down behind the last record.
static void release_borrowed(tbk_fastx_batch *b) {
    if (!b->borrowed || !b->text_mapped || !b->text || b->recs.empty()) return;
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = ((uintptr_t)(b->text + b->recs.front().head) + page - 1) / page * page;  // whole pages inside the batch's text only
    const uintptr_t hi = (uintptr_t)(b->text + b->recs.back().end) / page * page;
    if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_DONTNEED);
}

static int fastx_next_records(tbk_fastx_reader *r, tbk_fastx_batch *b, uint64_t max_bases, uint64_t max_reads) {
    release_borrowed(b);
    b->clear();
    if (r->state == tbk_fastx_reader::DONE) return TBK_OK;
    if (max_reads == 0) max_reads = ~0ull;
    if (max_bases == 0) max_bases = ~0ull;
    if (r->scan.map && (r->scan.active || r->scan.pos > 0)) {
        if (r->scan.active) {
            const int rc = regular_next(r, b, max_bases, max_reads);
            if (rc) return rc;
            if (b->n_reads() > 0) return TBK_OK;
            if (r->scan.pos >= r->scan.size) { r->state = tbk_fastx_reader::DONE; return TBK_OK; }
        }
        // the scan has stopped in front of a record that is not regular: the sequential machine reads
        // on from that byte, in its SEEK state
        if (!r->scan.active && r->scan.pos != (size_t)-1) {
            if (lseek(r->src.fd, (off_t)r->scan.pos, SEEK_SET) < 0) return ffail(TBK_ERR_IO, "lseek: %s", strerror(errno));
            r->src.pos = r->src.end = 0;
            r->src.text_eof = false; r->src.skip_lf = false;
            r->state = tbk_fastx_reader::SEEK;
            r->scan.pos = (size_t)-1;  // handed over
        }
    }
    if (r->scan.inflated && r->state == tbk_fastx_reader::SEEK && !r->have_pending && !r->src.skip_lf) {
        const int rc = regular_next_inflated(r, b, max_bases, max_reads);
        if (rc) return rc;
        if (b->n_reads() > 0) return TBK_OK;  // (a batch the scan left unfilled at an irregular record: the machine fills the next one)
    }
    // size the pinned sequence buffer once, from the batch limit, instead of growing it
    if (max_bases != ~0ull && max_bases <= ((uint64_t)1 << 34) && b->bases_cap < max_bases)
        if (!b->reserve_bases((size_t)max_bases + ((size_t)max_bases >> 3) + (1 << 20)))
            return ffail(TBK_ERR_NOMEM, "out of memory sizing a read batch");
    if (r->have_pending) {  // a FASTA record whose header closed the previous batch
        b->begin((const uint8_t *)r->pending_name.data(), r->pending_name.size());
        r->have_pending = false;
    }
    auto full = [&]() { return b->n_reads() >= max_reads || b->n_bases >= max_bases; };
    const uint8_t *p;
    size_t len;
    bool term;
    for (;;) {
        int got = r->src.next(p, len, term);
        if (got < 0) return ffail(TBK_ERR_IO, "%s", r->src.err.c_str());
        if (got == 0) {  // input exhausted
            if (r->state == tbk_fastx_reader::SEQ) b->finish(false);
            else if (r->state == tbk_fastx_reader::QUAL) b->finish(false);  // short quality: FASTA (seq.py:81-83)
            r->state = tbk_fastx_reader::DONE;
            return TBK_OK;
        }
        const uint8_t head = len ? p[0] : (uint8_t)'\n';
        const size_t body = term ? len : (len ? len - 1 : 0);  // line[:-1]
        switch (r->state) {
            case tbk_fastx_reader::SEEK:
                if (head == '>' || head == '@') {
                    if (body == 0) { r->state = tbk_fastx_reader::DONE; return TBK_OK; }
                    const uint8_t *np; size_t nn;
                    name_of(p, body, np, nn);
                    b->begin(np, nn);
                    r->state = tbk_fastx_reader::SEQ;
                }
                break;
            case tbk_fastx_reader::SEQ:
                if (head == '@' || head == '+' || head == '>') {
                    if (body == 0) { b->finish(false); r->state = tbk_fastx_reader::DONE; return TBK_OK; }
                    if (head == '+') {
                        r->seq_len = b->n_bases - b->rec_seq0;
                        r->qual_have = 0;
                        r->state = tbk_fastx_reader::QUAL;
                    } else {
                        b->finish(false);
                        const uint8_t *np; size_t nn;
                        name_of(p, body, np, nn);
                        if (full()) {  // hand the batch over; the new record starts the next one
                            r->pending_name.assign((const char *)np, nn);
                            r->have_pending = true;
                            return TBK_OK;
                        }
                        b->begin(np, nn);
                    }
                } else {
                    if (!b->seq(p, body)) return ffail(TBK_ERR_NOMEM, "out of memory growing a read batch");
                }
                break;
            case tbk_fastx_reader::QUAL:
                b->qual(p, body);
                r->qual_have += (int64_t)len + (term ? 1 : 0) - 1;  // len(line) - 1
                if ((uint64_t)r->qual_have >= r->seq_len) {
                    b->finish(true);
                    r->state = tbk_fastx_reader::SEEK;
                    if (full()) return TBK_OK;
                }
                break;
            default:
                return TBK_OK;
        }
    }
}

extern "C" int tbk_fastx_batch_view(const tbk_fastx_batch *b, uint64_t *n_reads, const uint8_t **bases,
                                    const uint64_t **base_off, const uint8_t **names, const uint64_t **name_off,
                                    const uint8_t **quals, const uint64_t **qual_off, const uint8_t **has_qual) {
    if (!b) return ffail(TBK_ERR_INVALID, "NULL argument");
    static const uint8_t nothing = 0;
    if (n_reads) *n_reads = b->n_reads();
    if (bases) *bases = b->bases && !b->borrowed ? b->bases : &nothing;  // (a borrowed batch has no sequence / quality arrays)
    if (base_off) *base_off = b->base_off.data();
    if (names) *names = b->names.empty() ? &nothing : b->names.data();
    if (name_off) *name_off = b->name_off.data();
    if (quals) *quals = b->quals.empty() || b->borrowed ? &nothing : b->quals.data();
    if (qual_off) *qual_off = b->qual_off.data();
    if (has_qual) *has_qual = b->has_qual.empty() ? &nothing : b->has_qual.data();
    return TBK_OK;
}

// =======================================================================================
// Python's str(float) (repr): shortest round-trip digits; exponent form when the decimal
// point position is <= -4 or > 16; ".0" appended to integers.
// =======================================================================================
static size_t py_float_repr(double v, char *out) {
    char tmp[64];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::scientific);
    *r.ptr = 0;
    // tmp = [-]d[.ddd]e[+-]XX
    char *q = tmp;
    size_t o = 0;
    if (*q == '-') { out[o++] = '-'; q++; }
    if (*q == 'i' || *q == 'n') { strcpy(out + o, *q == 'i' ? "inf" : "nan"); return o + 3; }
    char digits[40];
    int nd = 0;
    while (*q && *q != 'e') { if (*q != '.') digits[nd++] = *q; q++; }
    const int e10 = atoi(q + 1);
    const int decpt = e10 + 1;
    if (decpt <= -4 || decpt > 16) {
        out[o++] = digits[0];
        if (nd > 1) { out[o++] = '.'; memcpy(out + o, digits + 1, nd - 1); o += nd - 1; }
        o += (size_t)sprintf(out + o, "e%c%02d", e10 < 0 ? '-' : '+', e10 < 0 ? -e10 : e10);
        return o;
    }
    if (decpt <= 0) {
        out[o++] = '0'; out[o++] = '.';
        for (int i = 0; i < -decpt; i++) out[o++] = '0';
        memcpy(out + o, digits, nd); o += nd;
        return o;
    }
    if (nd <= decpt) {
        memcpy(out + o, digits, nd); o += nd;
        for (int i = nd; i < decpt; i++) out[o++] = '0';
        out[o++] = '.'; out[o++] = '0';
        return o;
    }
    memcpy(out + o, digits, decpt); o += decpt;
    out[o++] = '.';
    memcpy(out + o, digits + decpt, nd - decpt); o += nd - decpt;
    return o;
}

extern "C" int tbk_format_float(double v, char *out, size_t cap) {
    if (!out || cap < 40) return ffail(TBK_ERR_INVALID, "buffer too small");
    size_t n = py_float_repr(v, out);
    out[n] = 0;
    return (int)n;
}

// name \t bin \t score_a \t score_b \n for every read of the batch (classify_by_kmers.py:117)
extern "C" int tbk_format_tsv(const tbk_fastx_batch *b, const char *bins, const double *score_a, const double *score_b,
                              char *out, size_t cap, size_t *len) {
    if (!b || !len || (b->n_reads() && (!bins || !score_a || !score_b))) return ffail(TBK_ERR_INVALID, "NULL argument");
    const uint64_t n = b->n_reads();
    const size_t need = b->names.size() + n * 72 + 1;
    *len = need;
    if (!out || cap < need) return TBK_OK;  // caller asks for the size first
    size_t o = 0;
    for (uint64_t i = 0; i < n; i++) {
        const size_t nn = b->name_off[i + 1] - b->name_off[i];
        memcpy(out + o, b->names.data() + b->name_off[i], nn); o += nn;
        out[o++] = '\t'; out[o++] = bins[i]; out[o++] = '\t';
        o += py_float_repr(score_a[i], out + o);
        out[o++] = '\t';
        o += py_float_repr(score_b[i], out + o);
        out[o++] = '\n';
    }
    *len = o;
    return TBK_OK;
}

// =======================================================================================
// bin writer
// =======================================================================================
// bytes not yet written: a plain buffer that grows by realloc (a vector would zero-fill gigabytes
// that are overwritten the next moment)
struct TextBuf {
    char *p = nullptr;
    size_t n = 0, cap = 0;
    bool pinned = false;  // pinned (tbk_pin_alloc_): what the GPU gzip encoder copies its text from (set before the first grow_to)
    ~TextBuf() { release(); }
    void release() { if (p) { if (pinned) tbk_pin_free_(p); else free(p); } p = nullptr; cap = 0; }
    bool grow_to(size_t need) {
        if (need <= cap) return true;
        size_t c = std::max(need + need / 4, (size_t)1 << 20);
        if (pinned) {
            char *q = (char *)tbk_pin_alloc_(c);
            if (!q) return false;
            if (n) memcpy(q, p, n);
            if (p) tbk_pin_free_(p);
            p = q; cap = c;
            return true;
        }
        char *q = (char *)realloc(p, c);
        if (!q) return false;
        p = q; cap = c;
        return true;
    }
    char *data() { return p; }
    size_t size() const { return n; }
};

struct BinFile {
    int fd = -1;
    uint64_t file_off = 0;  // bytes written so far (this descriptor's own offset: plain output is written with pwrite)
    bool positional = false;  // a regular file no other bin names: slices may go out with pwrite from several threads;
                              // anything else (FIFO, /dev/stdout, two prefixes naming one file) gets sequential write()
    TextBuf text;           // records not yet written
    TextBuf alt;            // the GPU encoder's second buffer: `text` and `alt` change places at every flush, so that the next batch is gathered
                            // while the device is still fetching this one's text
};

// One thread per bin that writes finished gzip members to the bin's file while the writer thread is already gathering the next
// batch (the GPU encoder's path: the members come back from the device two batches later).
struct IoLane {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<const char *, size_t>> q;
    bool busy = false, stop = false;
    int fd = -1;
    std::string err;
    uint64_t written = 0;
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty()) { if (stop) return; continue; }
            auto [p, n] = q.front();
            q.pop_front();
            busy = true;
            lk.unlock();
            std::string e;
            while (n) {
                const ssize_t k = ::write(fd, p, n);
                if (k < 0) { if (errno == EINTR) continue; e = std::string("write: ") + strerror(errno); break; }
                p += k; n -= (size_t)k; written += (uint64_t)k;
            }
            lk.lock();
            if (!e.empty() && err.empty()) err = e;
            busy = false;
            cv.notify_all();
        }
    }
    void start(int fd_) { fd = fd_; th = std::thread([this] { run(); }); }
    void put(const char *p, size_t n) { { std::lock_guard<std::mutex> lk(mu); q.emplace_back(p, n); } cv.notify_all(); }
    void idle() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return q.empty() && !busy; }); }
    void end() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); }
};

struct tbk_bin_writer {
    BinFile bin[3];  // A, B, U
    bool gz = false;
    int level = 6;
    int threads = 1;
    size_t chunk = (size_t)1 << 20;  // text per gzip member: small enough that one batch keeps every host thread busy
    std::string err;
    // the GPU gzip encoder (tbk_gdeflate.hip; tbk_bin_writer_use_device): members coded on the device, written by a thread per bin
    tbk_gdeflate *gpu = nullptr;
    IoLane lane[3];
    bool lanes_on = false;
    double gpu_submit_s = 0, gpu_crc_s = 0, gpu_text_wait_s = 0, gpu_collect_s = 0, gpu_io_wait_s = 0, fill_s = 0;
};

static bool write_all(int fd, const char *p, size_t n, std::string &err) {
    while (n) {
        ssize_t w = ::write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; err = std::string("write: ") + strerror(errno); return false; }
        p += w; n -= (size_t)w;
    }
    return true;
}

// The zlib path of the gzip members (TBK_GZIP_ENCODER=zlib or a pinned TBK_GZIP_STRATEGY; the default
// is the library's own encoder, tbk_deflate.cpp).
// FASTQ text is bases (no LZ77 match worth having in a 32 KiB window) and qualities.  zlib's default
// strategy spends 8-10 times the time of its match-free ones for nothing (0.43 against 0.43-0.44 of the
// input on HiFi-like records) - and the gzip members are what a run with compressed output waits
// for.  Of the match-free strategies Huffman-only is the faster (about 1.3x) and on qualities that
// vary the smaller (0.429 against 0.442); run-length wins where qualities come in runs (binned or
// constant qualities: a third of the size).  So each member looks at its first 64 KiB: where more
// than 45 % of the bytes equal their predecessor it is deflated with Z_RLE, else with Z_HUFFMAN_ONLY.
// TBK_GZIP_STRATEGY=rle / huffman / default pins one.  Decompressed bytes are the same either way.
static int deflate_strategy(const char *src, size_t n) {
    static const int pinned = [] {
        const char *e = getenv("TBK_GZIP_STRATEGY");
        if (!e) return -1;
        return strcmp(e, "default") == 0 ? Z_DEFAULT_STRATEGY : strcmp(e, "huffman") == 0 ? Z_HUFFMAN_ONLY : strcmp(e, "rle") == 0 ? Z_RLE : -1;
    }();
    if (pinned >= 0) return pinned;
    const size_t look = n < ((size_t)64 << 10) ? n : ((size_t)64 << 10);
    if (look < 2) return Z_RLE;
    size_t same = 0;
    for (size_t i = 1; i < look; i++) same += src[i] == src[i - 1];
    return same * 100 > (look - 1) * 45 ? Z_RLE : Z_HUFFMAN_ONLY;
}

bool tbk_gzip_member_literal(const char *src, size_t n, std::vector<char> &out);  // tbk_deflate.cpp

// one gzip member per chunk; concatenated members are a valid gzip file
static bool deflate_member(const char *src, size_t n, int level, std::vector<char> &dst) {
    // the library's own encoder (tbk_deflate.cpp: Huffman coding per line-aligned block, runs as
    // distance-1 matches) unless zlib is asked for or a zlib strategy is pinned
    static const bool own = !(getenv("TBK_GZIP_ENCODER") && strcmp(getenv("TBK_GZIP_ENCODER"), "zlib") == 0) && !getenv("TBK_GZIP_STRATEGY");
    if (own && level > 0) return tbk_gzip_member_literal(src, n, dst);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, deflate_strategy(src, n)) != Z_OK) return false;
    dst.resize(deflateBound(&zs, n) + 64);
    zs.next_in = (Bytef *)src; zs.avail_in = (uInt)n;
    zs.next_out = (Bytef *)dst.data(); zs.avail_out = (uInt)dst.size();
    int rc = deflate(&zs, Z_FINISH);
    const size_t out = dst.size() - zs.avail_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    dst.resize(out);
    return true;
}

struct Piece { int bin; const char *src; size_t n; std::vector<char> out; bool ok = true; };

static inline double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// finished members to their bins' files: each bin's are handed to its lane in order (the lanes of the three bins run side by side)
static int hand_to_lanes(tbk_bin_writer *w, const std::vector<tbk_gdeflate_out> &outs) {
    const double t0 = wall_now();
    for (IoLane &l : w->lane) l.idle();  // (the bytes handed over last time are on their way to being reused)
    w->gpu_io_wait_s += wall_now() - t0;
    for (IoLane &l : w->lane) if (!l.err.empty()) return ffail(TBK_ERR_IO, "%s", l.err.c_str());
    for (size_t i = 0; i < outs.size();) {
        size_t k = i;
        size_t n = outs[i].n;
        while (k + 1 < outs.size() && outs[k + 1].tag == outs[i].tag && outs[k + 1].data == outs[k].data + outs[k].n) { k++; n += outs[k].n; }
        w->lane[outs[i].tag].put(outs[i].data, n);
        w->bin[outs[i].tag].file_off += n;
        i = k + 1;
    }
    return TBK_OK;
}

// The gzip members of this flush coded on the device (tbk_gdeflate.hip).  Three jobs deep: this flush's text goes to the
// device and is coded while the host sums the members' CRC-32s; the previous flush's members travel home; the one before
// that is handed to the bins' writer threads.  The text buffers are free again when this returns.
static int flush_bins_gpu(tbk_bin_writer *w, bool final, std::vector<Piece> &pieces, const size_t keep[3]) {
    std::vector<tbk_gdeflate_out> outs;
    int rc = TBK_OK;
    bool swapped = false;
    if (!pieces.empty()) {
        double t0 = wall_now();
        std::vector<tbk_gdeflate_member> members;
        members.reserve(pieces.size());
        for (const Piece &pc : pieces) members.push_back(tbk_gdeflate_member{pc.src, pc.n, pc.bin});
        rc = tbk_gdeflate_submit(w->gpu, members.data(), members.size());
        if (rc) return rc;
        w->gpu_submit_s += wall_now() - t0;
        t0 = wall_now();
        static const bool host_crc = getenv("TBK_GZIP_CRC") && strcmp(getenv("TBK_GZIP_CRC"), "host") == 0;  // (default: the device sums them too)
        if (host_crc) {
            std::atomic<size_t> next{0};
            std::vector<uint32_t> crc(pieces.size());
            auto work = [&]() {
                for (size_t i; (i = next.fetch_add(1)) < pieces.size();) crc[i] = tbk_crc32(0, (const uint8_t *)pieces[i].src, pieces[i].n);
            };
            const int nt = (int)std::min<size_t>((size_t)w->threads, pieces.size());
            std::vector<std::thread> pool;
            for (int t = 1; t < nt; t++) pool.emplace_back(work);
            work();
            for (auto &t : pool) t.join();
            for (size_t i = 0; i < pieces.size(); i++) tbk_gdeflate_set_crc(w->gpu, i, crc[i]);
        }
        w->gpu_crc_s += wall_now() - t0;
        t0 = wall_now();
        // the text of THIS flush stays where it is until the device has fetched it; what is left over moves to the other buffer
        // (whose own text - the flush before this one's - must have left: it has, long ago) and the two change places
        rc = tbk_gdeflate_text_done(w->gpu, 1);
        if (rc) return rc;
        w->gpu_text_wait_s += wall_now() - t0;
        for (int b = 0; b < 3; b++) {
            BinFile &f = w->bin[b];
            f.alt.pinned = true;
            if (!f.alt.grow_to(std::max<size_t>(keep[b], 1))) return ffail(TBK_ERR_NOMEM, "out of memory buffering a bin");
            if (keep[b]) memcpy(f.alt.data(), f.text.data() + f.text.size() - keep[b], keep[b]);
            f.alt.n = keep[b];
            std::swap(f.text.p, f.alt.p); std::swap(f.text.n, f.alt.n); std::swap(f.text.cap, f.alt.cap);
        }
        swapped = true;
        t0 = wall_now();
        rc = tbk_gdeflate_collect(w->gpu, false, outs);
        w->gpu_collect_s += wall_now() - t0;
        if (rc) return rc;
        if (!outs.empty()) { rc = hand_to_lanes(w, outs); if (rc) return rc; }
    }
    if (final) {
        while (tbk_gdeflate_in_flight(w->gpu) > 0) {
            const double t0 = wall_now();
            rc = tbk_gdeflate_collect(w->gpu, true, outs);
            w->gpu_collect_s += wall_now() - t0;
            if (rc) return rc;
            if (!outs.empty()) { rc = hand_to_lanes(w, outs); if (rc) return rc; }
        }
        for (IoLane &l : w->lane) l.idle();
        for (IoLane &l : w->lane) if (!l.err.empty()) return ffail(TBK_ERR_IO, "%s", l.err.c_str());
    }
    if (!swapped)
        for (int b = 0; b < 3; b++) {
            BinFile &f = w->bin[b];
            if (keep[b] && keep[b] != f.text.size()) memmove(f.text.data(), f.text.data() + f.text.size() - keep[b], keep[b]);
            f.text.n = keep[b];
        }
    return TBK_OK;
}

static int flush_bins(tbk_bin_writer *w, bool final) {
    // cut every bin's pending text into pieces (whole chunks; everything when final)
    std::vector<Piece> pieces;
    size_t keep[3] = {0, 0, 0};
    for (int b = 0; b < 3; b++) {
        BinFile &f = w->bin[b];
        size_t off = 0;
        const size_t n = f.text.size();
        if (!w->gz) { if (n) pieces.push_back(Piece{b, f.text.data(), n, {}, true}); keep[b] = 0; continue; }
        while (n - off >= w->chunk) { pieces.push_back(Piece{b, f.text.data() + off, w->chunk, {}, true}); off += w->chunk; }
        if (final && n > off) { pieces.push_back(Piece{b, f.text.data() + off, n - off, {}, true}); off = n; }
        keep[b] = n - off;
    }
    if (w->gz && w->gpu) return flush_bins_gpu(w, final, pieces, keep);
    if (w->gz && !pieces.empty()) {
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < pieces.size();) {
                Piece &pc = pieces[i];
                // zlib's avail_in is 32-bit; chunks are far below that
                pc.ok = deflate_member(pc.src, pc.n, w->level, pc.out);
            }
        };
        const int nt = (int)std::min<size_t>((size_t)w->threads, pieces.size());
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    if (w->gz) {
        for (Piece &pc : pieces) {
            if (!pc.ok) return ffail(TBK_ERR_IO, "deflate failed");
            if (!write_all(w->bin[pc.bin].fd, pc.out.data(), pc.out.size(), w->err)) return ffail(TBK_ERR_IO, "%s", w->err.c_str());
            w->bin[pc.bin].file_off += pc.out.size();
        }
    } else {
        // Plain output: the bytes of a bin go to the file at the offset a sequence of write() calls
        // would have reached (each descriptor counts for itself, as the kernel does).  One thread per file: a
        // buffered write holds the file's inode lock, more threads per file only spin for it (see
        // tbk_bin_writer_write).
        std::atomic<int> failed_errno{0};
        std::vector<std::thread> pool;
        for (Piece &pc : pieces) {
            BinFile &f = w->bin[pc.bin];
            if (!f.positional) {
                // not seekable, or a file another bin writes too: the bytes go out in order through the
                // descriptor's own offset, as the reference's open(name, "w") handles do (seq.py:128-134)
                if (!write_all(f.fd, pc.src, pc.n, w->err)) { for (std::thread &th : pool) th.join(); return ffail(TBK_ERR_IO, "%s", w->err.c_str()); }
                f.file_off += pc.n;
                continue;
            }
            const int fd = f.fd;
            const char *src = pc.src;
            const size_t len = pc.n;
            const uint64_t off = f.file_off;
            auto work = [fd, src, len, off, &failed_errno]() {
                size_t done = 0;
                while (done < len) {
                    const ssize_t k = ::pwrite(fd, src + done, len - done, (off_t)(off + done));
                    if (k < 0) { if (errno == EINTR) continue; failed_errno.store(errno); return; }
                    done += (size_t)k;
                }
            };
            f.file_off += pc.n;
            if (w->threads > 1 && len >= ((size_t)1 << 20)) pool.emplace_back(work);  // (plain output: one piece per bin)
            else work();
        }
        for (std::thread &th : pool) th.join();
        if (failed_errno.load()) return ffail(TBK_ERR_IO, "write: %s", strerror(failed_errno.load()));
    }
    for (int b = 0; b < 3; b++) {
        BinFile &f = w->bin[b];
        if (keep[b] && keep[b] != f.text.size()) memmove(f.text.data(), f.text.data() + f.text.size() - keep[b], keep[b]);
        f.text.n = keep[b];
    }
    return TBK_OK;
}

extern "C" int tbk_bin_writer_open(const char *path_a, const char *path_b, const char *path_u, int gzip_output, int level,
                                   int threads, tbk_bin_writer **out) {
    if (!out || !path_a || !path_b || !path_u) return ffail(TBK_ERR_INVALID, "NULL argument");
    *out = nullptr;
    tbk_bin_writer *w = new tbk_bin_writer();
    w->gz = gzip_output != 0;
    w->level = level < 0 ? 6 : std::min(level, 9);
    if (threads <= 0) threads = tbk_host_threads();
    w->threads = std::min(threads, 512);
    const char *paths[3] = {path_a, path_b, path_u};
    for (int b = 0; b < 3; b++) {
        // truncating open, as open(name, "w") / gzip.open(name, "wt") do (seq.py:128-134); when two
        // names coincide the later bin re-opens the same file, exactly like the reference
        w->bin[b].fd = ::open(paths[b], O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (w->bin[b].fd < 0) {
            int e = errno;
            for (int c = 0; c < b; c++) ::close(w->bin[c].fd);
            delete w;
            return ffail(TBK_ERR_IO, "cannot create %s: %s", paths[b], strerror(e));
        }
    }
    // pwrite needs a seekable target of this bin's own: probe each descriptor once
    struct stat st[3];
    bool have[3];
    for (int b = 0; b < 3; b++) {
        have[b] = ::fstat(w->bin[b].fd, &st[b]) == 0;
        w->bin[b].positional = have[b] && S_ISREG(st[b].st_mode) && ::lseek(w->bin[b].fd, 0, SEEK_CUR) >= 0;
    }
    for (int b = 0; b < 3; b++)
        for (int c = 0; c < 3; c++)
            if (c != b && have[b] && have[c] && st[b].st_dev == st[c].st_dev && st[b].st_ino == st[c].st_ino) w->bin[b].positional = false;
    *out = w;
    return TBK_OK;
}

// The bytes Read.print writes for record i (seq.py:27-31: FASTQ when the record has a non-empty quality string, else
// FASTA; name = header up to the first space, bare '+' line), as up to 7 pieces of memory.  A record copied into the
// batch's arrays is gathered from them and from constants.  A borrowed record lies in the mapped input as
// "@name[ comment]\nSEQ\n+[...]\nQUAL\n": what the header's comment and the '+' line's text do not interrupt is
// written as it lies there - a record without either is one piece, and one piece with its neighbours of the same bin.
static inline int record_pieces(const tbk_fastx_batch *b, size_t i, struct iovec *v) {
    static const char at_c = '@', gt_c = '>', nl_c = '\n';
    static const char plus_c[3] = {'\n', '+', '\n'};
    const size_t nn = b->name_off[i + 1] - b->name_off[i];
    const size_t ns = b->base_off[i + 1] - b->base_off[i];
    const size_t nq = b->qual_off[i + 1] - b->qual_off[i];
    const bool fq = b->has_qual[i] && nq > 0;
    int c = 0;
    if (b->borrowed && fq) {
        const FastqRec &r = b->recs[i];
        const uint8_t *t = b->text;
        const bool comment = r.seq - r.head - 2 != r.name_len, bare = r.qual - r.plus == 2;
        if (!comment && bare) { v[c++] = {(void *)(t + r.head), (size_t)(r.end - r.head)}; return c; }
        if (comment) {
            v[c++] = {(void *)(t + r.head), (size_t)1 + r.name_len};                        // "@name"
            if (bare) { v[c++] = {(void *)(t + r.seq - 1), (size_t)(r.end - r.seq + 1)}; return c; }  // "\nSEQ\n+\nQUAL\n"
            v[c++] = {(void *)(t + r.seq - 1), (size_t)(r.plus + 1 - (r.seq - 1))};        // "\nSEQ\n+"
        } else {
            v[c++] = {(void *)(t + r.head), (size_t)(r.plus + 1 - r.head)};                // "@name\nSEQ\n+"
        }
        v[c++] = {(void *)(t + r.qual - 1), (size_t)(r.end - r.qual + 1)};                 // "\nQUAL\n"
        return c;
    }
    v[c++] = {(void *)(fq ? &at_c : &gt_c), 1};
    if (nn) v[c++] = {(void *)(b->names.data() + b->name_off[i]), nn};
    v[c++] = {(void *)&nl_c, 1};
    if (ns) v[c++] = {(void *)((b->borrowed ? b->text + b->recs[i].seq : b->bases + b->base_off[i])), ns};
    if (fq) {
        v[c++] = {(void *)plus_c, 3};
        v[c++] = {(void *)(b->quals.data() + b->qual_off[i]), nq};
    }
    v[c++] = {(void *)&nl_c, 1};
    return c;
}

// append the batch's records to their bins in input order: FASTQ when the record has a
// non-empty quality string, else FASTA (seq.py:27-31)
extern "C" int tbk_bin_writer_write(tbk_bin_writer *w, const tbk_fastx_batch *b, const char *bins) {
    if (!w || !b || (b->n_reads() && !bins)) return ffail(TBK_ERR_INVALID, "NULL argument");
    const uint64_t n = b->n_reads();
    // where every record goes: its bin and its offset in that bin's pending text (one serial pass
    // over the records' lengths), then the bytes are put there by several threads
    std::vector<uint64_t> dst((size_t)n), upto((size_t)n + 1);
    uint64_t at[3] = {w->bin[0].text.size(), w->bin[1].text.size(), w->bin[2].text.size()};
    upto[0] = 0;
    for (uint64_t i = 0; i < n; i++) {
        const int which = bins[i] == 'A' ? 0 : bins[i] == 'B' ? 1 : 2;
        const size_t nn = b->name_off[i + 1] - b->name_off[i];
        const size_t ns = b->base_off[i + 1] - b->base_off[i];
        const size_t nq = b->qual_off[i + 1] - b->qual_off[i];
        const bool fq = b->has_qual[i] && nq > 0;
        const size_t len = 1 + nn + 1 + ns + 1 + (fq ? 2 + nq + 1 : 0);
        dst[(size_t)i] = at[which];
        at[which] += len;
        upto[(size_t)i + 1] = upto[(size_t)i] + len;
    }
    const uint64_t total = upto[(size_t)n];
    // Plain output of long records into regular files: the records go from where they lie - the batch's arrays, or
    // the mapped input of a borrowed batch - straight to their places in the files (pwritev), no copy into a bin
    // buffer first.  What a sequence of write() calls would have produced, byte for byte.  ONE thread per bin: a
    // buffered write holds its file's inode lock, so a second thread on the same file adds nothing but the CPU
    // time it spins for the lock (12.9 GB into two files, GPU box: 2 threads 20.0 GB/s and 1.3 CPU-s, 16 threads
    // 18.7 GB/s and 5.5 CPU-s - profiles/r03/write_threads.log), CPU time the reader's threads need.  Short records
    // (many pieces per megabyte), gzip output and targets that cannot seek take the buffered path below.
    if (!w->gz && n > 0 && total / n >= 4096 && w->bin[0].positional && w->bin[1].positional && w->bin[2].positional &&
        w->bin[0].text.size() == 0 && w->bin[1].text.size() == 0 && w->bin[2].text.size() == 0) {
        std::atomic<int> failed_errno{0};
        auto put = [&](int which) {
            // the bin's records are consecutive in its file: their pieces are gathered, up to ~1000 at a time, into one pwritev
            constexpr int CAP = 1008;
            std::vector<struct iovec> v;
            v.reserve(CAP + 8);
            uint64_t start = 0;
            auto flush = [&]() {
                struct iovec *q = v.data();
                int c = (int)v.size();
                uint64_t off = start;
                while (c > 0 && !failed_errno.load()) {
                    const ssize_t k = ::pwritev(w->bin[which].fd, q, c, (off_t)off);
                    if (k < 0) { if (errno == EINTR) continue; failed_errno.store(errno); return; }
                    off += (uint64_t)k;
                    size_t done = (size_t)k;
                    while (c > 0 && done >= q->iov_len) { done -= q->iov_len; q++; c--; }
                    if (c > 0 && done) { q->iov_base = (char *)q->iov_base + done; q->iov_len -= done; }
                }
                start = off;
                v.clear();
            };
            const char mine = "ABU"[which];
            for (size_t i = 0; i < (size_t)n && !failed_errno.load(); i++) {
                if ((bins[i] == 'A' || bins[i] == 'B' ? bins[i] : 'U') != mine) continue;
                // (dst[i] counts from the bin's buffer start, which is empty here: an offset into this batch's share)
                if (v.empty()) start = w->bin[which].file_off + dst[i];
                struct iovec pc[7];
                const int np = record_pieces(b, i, pc);
                for (int j = 0; j < np; j++) {
                    // (neighbouring records of a borrowed batch that go to the same bin lie back to back in the input)
                    if (!v.empty() && (char *)v.back().iov_base + v.back().iov_len == (char *)pc[j].iov_base) v.back().iov_len += pc[j].iov_len;
                    else v.push_back(pc[j]);
                }
                if ((int)v.size() >= CAP) flush();
            }
            if (!v.empty()) flush();
        };
        std::vector<std::thread> pool;
        if (w->threads > 1) {
            for (int k = 1; k < 3; k++) if (at[k]) pool.emplace_back(put, k);
            if (at[0]) put(0);
        } else {
            for (int k = 0; k < 3; k++) if (at[k]) put(k);
        }
        for (std::thread &th : pool) th.join();
        if (failed_errno.load()) return ffail(TBK_ERR_IO, "write: %s", strerror(failed_errno.load()));
        for (int k = 0; k < 3; k++) w->bin[k].file_off += at[k];
        return TBK_OK;
    }
    for (int k = 0; k < 3; k++) {
        if (!w->bin[k].text.grow_to((size_t)at[k])) return ffail(TBK_ERR_NOMEM, "out of memory buffering a bin");
        w->bin[k].text.n = (size_t)at[k];
    }
    // (the device codes the members: the gather is all the host does with the text, and the reader's threads want the CPUs - six copy 134 MB in 5 ms)
    const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(w->gpu ? std::min(w->threads, 6) : w->threads), total / ((uint64_t)4 << 20)));
    auto fill = [&](int t) {
        auto cut = [&](int u) -> size_t {
            if (u <= 0) return 0;
            if (u >= nt) return (size_t)n;
            return (size_t)(std::lower_bound(upto.begin(), upto.begin() + (ptrdiff_t)n, total * (uint64_t)u / (uint64_t)nt) - upto.begin());
        };
        const size_t last = cut(t + 1);
        for (size_t i = cut(t); i < last; i++) {
            const int which = bins[i] == 'A' ? 0 : bins[i] == 'B' ? 1 : 2;
            char *p = w->bin[which].text.data() + dst[i];
            struct iovec pc[7];
            const int np = record_pieces(b, i, pc);
            for (int j = 0; j < np; j++) { memcpy(p, pc[j].iov_base, pc[j].iov_len); p += pc[j].iov_len; }
        }
    };
    {
        const double t0 = wall_now();
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(fill, t);
        fill(0);
        for (std::thread &th : pool) th.join();
        w->fill_s += wall_now() - t0;
    }
    return flush_bins(w, false);
}

// The gzip members from here on are coded on `device` (tbk_gdeflate.hip) instead of on the host's threads: call it right after
// tbk_bin_writer_open (gzip output; before the first write).  TBK_GZIP_ENCODER=cpu / zlib keeps the host encoders.
extern "C" int tbk_bin_writer_use_device(tbk_bin_writer *w, int device) {
    if (!w) return ffail(TBK_ERR_INVALID, "NULL argument");
    if (!w->gz || w->gpu) return TBK_OK;
    const char *enc = getenv("TBK_GZIP_ENCODER");
    if ((enc && (strcmp(enc, "cpu") == 0 || strcmp(enc, "zlib") == 0)) || getenv("TBK_GZIP_STRATEGY") || w->level == 0) return TBK_OK;
    for (int b = 0; b < 3; b++) if (w->bin[b].text.size()) return ffail(TBK_ERR_STATE, "tbk_bin_writer_use_device after the first write");
    // (two prefixes naming one file: the lanes would interleave their writes where the reference's handles do not)
    for (int b = 0; b < 3; b++) if (!w->bin[b].positional) return TBK_OK;
    int rc = tbk_gdeflate_create(device, &w->gpu);
    if (rc) return rc;
    for (int b = 0; b < 3; b++) { w->bin[b].text.release(); w->bin[b].text.pinned = true; w->lane[b].start(w->bin[b].fd); }
    w->lanes_on = true;
    return TBK_OK;
}

extern "C" int tbk_bin_writer_encoder(const tbk_bin_writer *w) { return w && w->gpu ? 1 : 0; }

extern "C" int tbk_bin_writer_close(tbk_bin_writer *w) {
    if (!w) return TBK_OK;
    int rc = flush_bins(w, true);
    if (w->lanes_on) {
        for (IoLane &l : w->lane) { l.idle(); l.end(); if (!l.err.empty() && !rc) rc = ffail(TBK_ERR_IO, "%s", l.err.c_str()); }
        w->lanes_on = false;
    }
    if (w->gpu) {
        if (getenv("TBK_WRITE_TIMING")) {
            uint64_t tb = 0, mb = 0, nb = 0, nm = 0;
            tbk_gdeflate_stats(w->gpu, &tb, &mb, &nb, &nm);
            fprintf(stderr, "tbk-gpu-gzip %.3f GB of text in %llu members / %llu blocks -> %.3f GB; writer thread: gather %.3f s, submit %.3f s, crc %.3f s, waiting for the text's copy %.3f s, "
                            "collect %.3f s, waiting for the bins' writers %.3f s\n", (double)tb / 1e9, (unsigned long long)nm, (unsigned long long)nb, (double)mb / 1e9, w->fill_s, w->gpu_submit_s,
                    w->gpu_crc_s, w->gpu_text_wait_s, w->gpu_collect_s, w->gpu_io_wait_s);
        }
        tbk_gdeflate_destroy(w->gpu);
        w->gpu = nullptr;
    }
    if (w->gz) {
        // an empty bin is still a valid (empty) gzip file, as gzip.open(...).close() leaves it
        for (int b = 0; b < 3; b++) {
            struct stat st;
            if (fstat(w->bin[b].fd, &st) == 0 && st.st_size == 0) {
                std::vector<char> out;
                if (deflate_member("", 0, w->level, out)) write_all(w->bin[b].fd, out.data(), out.size(), w->err);
            }
        }
    }
    for (int b = 0; b < 3; b++) if (w->bin[b].fd >= 0) ::close(w->bin[b].fd);
    delete w;
    return rc;
}
