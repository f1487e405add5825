// write_sources.cpp: ONE thread per file, two files - does it matter where the bytes come from?
//   warm   : a 64 MiB buffer used over and over (tools/write_paths.cpp's source)
//   cold   : a buffer as large as the output, filled once by 16 threads (a copied batch's arrays)
//   mapped : a page-cached file of that size, mapped MAP_PRIVATE and touched once by 16 threads (a borrowed batch)
// each written with pwritev of 1000 pieces of 30 KB (what the bin writer issues).
// Build: g++ -O2 -pthread tools/write_sources.cpp -o /tmp/write_sources ; run: /tmp/write_sources DIR [GB]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/uio.h>
#include <unistd.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu_s() {
    struct rusage u;
    getrusage(RUSAGE_SELF, &u);
    return u.ru_utime.tv_sec + u.ru_stime.tv_sec + (u.ru_utime.tv_usec + u.ru_stime.tv_usec) * 1e-6;
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t total = (size_t)(atof(argc > 2 ? argv[2] : "12") * (1 << 30)) / (2 * 30000 * 1000) * (2 * 30000 * 1000);
    const size_t piece = 30000, per_call = 1000;
    auto fill = [&](char *p, size_t n) {
        std::vector<std::thread> pool;
        for (int t = 0; t < 16; t++) pool.emplace_back([=]() { for (size_t i = n * t / 16; i < n * (t + 1) / 16; i++) p[i] = "ACGT"[(i * 2654435761u >> 13) & 3]; });
        for (auto &t : pool) t.join();
    };
    auto touch = [&](const char *p, size_t n) {
        std::atomic<size_t> sum{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < 16; t++) pool.emplace_back([&, t]() { size_t s = 0; for (size_t i = n * t / 16; i < n * (t + 1) / 16; i += 64) s += (unsigned char)p[i]; sum += s; });
        for (auto &t : pool) t.join();
        return sum.load();
    };
    std::vector<char> warm((size_t)64 << 20);
    fill(warm.data(), warm.size());
    char *cold = (char *)malloc(total);
    fill(cold, total);
    const std::string src_name = dir + "/write_sources_in_" + std::to_string(getpid());
    { int fd = open(src_name.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600); for (size_t o = 0; o < total;) { ssize_t k = write(fd, cold + o, std::min<size_t>(total - o, (size_t)1 << 30)); if (k <= 0) { perror("write"); return 1; } o += (size_t)k; } close(fd); sync(); }
    const int in_fd = open(src_name.c_str(), O_RDONLY);
    const char *mapped = (const char *)mmap(nullptr, total, PROT_READ, MAP_PRIVATE, in_fd, 0);
    if (mapped == MAP_FAILED) { perror("mmap"); return 1; }
    touch(mapped, total);
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 3; mode++) {
            const char *label = mode == 0 ? "warm" : mode == 1 ? "cold" : "mapped";
            int fds[2];
            std::string names[2];
            for (int f = 0; f < 2; f++) { names[f] = dir + "/write_sources_" + std::to_string(getpid()) + "_" + std::to_string(f); fds[f] = open(names[f].c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600); }
            const double t0 = now(), c0 = cpu_s();
            auto work = [&](int f) {
                const size_t half = total / 2;
                std::vector<struct iovec> v(per_call);
                for (size_t off = 0; off < half; off += piece * per_call) {
                    for (size_t j = 0; j < per_call; j++) {
                        // the two files take alternate pieces of the source, as two bins take alternate reads
                        const size_t so = 2 * (off + j * piece) + (size_t)f * piece;
                        const char *p = mode == 0 ? warm.data() + so % (warm.size() - piece) : mode == 1 ? cold + so : mapped + so;
                        v[j] = {(void *)p, piece};
                    }
                    size_t done = 0;
                    struct iovec *q = v.data();
                    int c = (int)per_call;
                    while (c > 0) {
                        ssize_t k = pwritev(fds[f], q, c, (off_t)(off + done));
                        if (k < 0) { perror("pwritev"); exit(1); }
                        done += (size_t)k;
                        size_t d = (size_t)k;
                        while (c > 0 && d >= q->iov_len) { d -= q->iov_len; q++; c--; }
                        if (c > 0 && d) { q->iov_base = (char *)q->iov_base + d; q->iov_len -= d; }
                    }
                }
            };
            std::thread other(work, 1);
            work(0);
            other.join();
            const double t1 = now(), c1 = cpu_s();
            for (int f = 0; f < 2; f++) { close(fds[f]); unlink(names[f].c_str()); }
            printf("%-7s 2 files, 1 thread each: %6.2f GB/s (%.2f s for %.1f GB, %.1f CPU-s)\n", label, total / 1e9 / (t1 - t0), t1 - t0, total / 1e9, c1 - c0);
            fflush(stdout);
        }
    unlink(src_name.c_str());
    return 0;
}
