// read_paths.cpp: how fast can N threads bring a page-cached file's bytes past the CPU once?
//   pread : 64 MiB windows, every thread preads its 4 MiB pieces into the window buffer, then the threads scan it
//           (a byte sum: the stand-in for the record scan) - what the FASTQ reader does
//   mmap  : the file is mapped, the threads scan the mapping piece by piece (no copy; page faults instead)
//   mmap+p: the same with MAP_POPULATE per 64 MiB window (page tables filled in one go)
// Build: g++ -O2 -pthread tools/read_paths.cpp -o /tmp/read_paths ; run: /tmp/read_paths FILE [threads]
// (the file is written by the caller; tools/gpu_host_side.sh makes one)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static uint64_t scan(const unsigned char *p, size_t n) {  // count newlines: one pass over the bytes
    uint64_t c = 0;
    for (size_t i = 0; i < n; i++) c += p[i] == '\n';
    return c;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: read_paths FILE [threads]\n"); return 2; }
    const int nt = argc > 2 ? atoi(argv[2]) : 16;
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror("open"); return 1; }
    struct stat st;
    fstat(fd, &st);
    const size_t total = (size_t)st.st_size, window = (size_t)64 << 20, piece = (size_t)4 << 20;
    auto pool_run = [&](size_t jobs, auto fn) {
        std::atomic<size_t> next{0};
        auto work = [&]() { for (size_t j; (j = next.fetch_add(1)) < jobs;) fn(j); };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    };
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            std::atomic<uint64_t> lines{0};
            std::vector<unsigned char> buf(mode == 0 ? window : 1);
            const double t0 = now();
            double t_copy = 0;
            for (size_t off = 0; off < total; off += window) {
                const size_t n = std::min(window, total - off), jobs = (n + piece - 1) / piece;
                if (mode == 0) {
                    const double a = now();
                    pool_run(jobs, [&](size_t j) {
                        size_t o = j * piece, len = std::min(piece, n - o), done = 0;
                        while (done < len) { ssize_t k = pread(fd, buf.data() + o + done, len - done, (off_t)(off + o + done)); if (k <= 0) { perror("pread"); exit(1); } done += (size_t)k; }
                    });
                    t_copy += now() - a;
                    pool_run(jobs, [&](size_t j) { size_t o = j * piece; lines += scan(buf.data() + o, std::min(piece, n - o)); });
                } else {
                    unsigned char *m = (unsigned char *)mmap(nullptr, n, PROT_READ, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fd, (off_t)off);
                    if (m == MAP_FAILED) { perror("mmap"); exit(1); }
                    pool_run(jobs, [&](size_t j) { size_t o = j * piece; lines += scan(m + o, std::min(piece, n - o)); });
                    munmap(m, n);
                }
            }
            const double dt = now() - t0;
            printf("%-7s threads %2d: %6.2f GB/s (%.2f s for %.1f GB%s; %llu lines)\n", mode == 0 ? "pread" : mode == 1 ? "mmap" : "mmap+p", nt, total / 1e9 / dt, dt, total / 1e9,
                   mode == 0 ? (", of which pread " + std::to_string(t_copy).substr(0, 4) + " s").c_str() : "", (unsigned long long)lines.load());
            fflush(stdout);
        }
    }
    return 0;
}
