// write_paths.cpp: how fast can N threads put G gigabytes into ONE regular file?
//   pwrite : every thread pwrite()s its 8 MiB slices at their offsets (buffered writes take the inode lock)
//   mmap   : the file is grown with ftruncate, mapped shared, and the threads memcpy their slices into the mapping
//   mmap+fa: the same behind posix_fallocate (space reserved first: ENOSPC instead of SIGBUS)
//   direct : the file is fallocate'd whole, opened O_DIRECT, and the threads pwrite() disjoint aligned 8 MiB slices from
//            an aligned buffer - no page cache, so no copy into it and no inode lock held across one; what the DEVICE takes
//            (round 4: the gate for "plain bins past the inode lock")
// Build: g++ -O2 -pthread tools/write_paths.cpp -o /tmp/write_paths ; run: /tmp/write_paths DIR [GB] [threads] [pwrite-only]
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
#include <sys/stat.h>
#include <unistd.h>

static double cpu_s() {  // user + system time of the process so far
    struct rusage u;
    getrusage(RUSAGE_SELF, &u);
    return u.ru_utime.tv_sec + u.ru_stime.tv_sec + (u.ru_utime.tv_usec + u.ru_stime.tv_usec) * 1e-6;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t total = (size_t)(atof(argc > 2 ? argv[2] : "8") * (1 << 30));
    const int nt = argc > 3 ? atoi(argv[3]) : 16;
    const size_t slice = (size_t)8 << 20, window = (size_t)64 << 20;  // a "batch" share of one bin
    char *src_mem = nullptr;
    if (posix_memalign((void **)&src_mem, 4096, window)) { perror("posix_memalign"); return 1; }
    struct Src { char *p; size_t n; char *data() const { return p; } size_t size() const { return n; } } src{src_mem, window};
    for (size_t i = 0; i < src.size(); i++) src.p[i] = "ACGT"[(i * 2654435761u >> 13) & 3];
    auto run = [&](const char *label, int mode, int files) {
        std::vector<int> fds;
        std::vector<std::string> names;
        for (int f = 0; f < files; f++) {
            names.push_back(dir + "/write_paths_" + std::to_string(getpid()) + "_" + std::to_string(f));
            fds.push_back(open(names.back().c_str(), O_RDWR | O_CREAT | O_TRUNC | (mode == 3 ? O_DIRECT : 0), 0600));
            if (fds.back() < 0) { perror(mode == 3 ? "open O_DIRECT" : "open"); if (mode == 3) { printf("direct   files %d: O_DIRECT not supported here\n", files); for (int g = 0; g < f; g++) { close(fds[g]); unlink(names[g].c_str()); } return; } exit(1); }
            if (mode == 3 && posix_fallocate(fds.back(), 0, (off_t)(total / files))) { perror("fallocate"); exit(1); }
        }
        const double t0 = now(), c0 = cpu_s();
        const size_t per_file = mode == 3 ? total / files / window * window : total / files;  // (O_DIRECT: aligned offsets and lengths)
        for (size_t off = 0; off < per_file; off += window) {
            const size_t n = std::min(window, per_file - off);
            std::vector<char *> maps(files, nullptr);
            if (mode == 1 || mode == 2)
                for (int f = 0; f < files; f++) {
                    if (mode == 2) { if (posix_fallocate(fds[f], (off_t)off, (off_t)n)) { perror("fallocate"); exit(1); } }
                    else if (ftruncate(fds[f], (off_t)(off + n))) { perror("ftruncate"); exit(1); }
                    maps[f] = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fds[f], (off_t)off);
                    if (maps[f] == MAP_FAILED) { perror("mmap"); exit(1); }
                }
            std::atomic<size_t> next{0};
            const size_t per = (n + slice - 1) / slice, jobs = per * files;
            auto work = [&]() {
                for (size_t j; (j = next.fetch_add(1)) < jobs;) {
                    const int f = (int)(j % files);
                    const size_t o = (j / files) * slice, len = std::min(slice, n - o);
                    if (mode == 1 || mode == 2) memcpy(maps[f] + o, src.data() + o, len);
                    else {
                        size_t done = 0;
                        while (done < len) { ssize_t k = pwrite(fds[f], src.data() + o + done, len - done, (off_t)(off + o + done)); if (k < 0) { perror("pwrite"); exit(1); } done += (size_t)k; }
                    }
                }
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < nt; t++) pool.emplace_back(work);
            work();
            for (auto &t : pool) t.join();
            for (int f = 0; f < files; f++) if (maps[f]) munmap(maps[f], n);
        }
        const double t1 = now(), c1 = cpu_s();
        for (int f = 0; f < files; f++) { close(fds[f]); unlink(names[f].c_str()); }
        printf("%-8s files %d threads %2d: %6.2f GB/s (%.2f s for %.1f GB, %.1f CPU-s; unlink %.2f s)\n", label, files, nt, total / 1e9 / (t1 - t0), t1 - t0, total / 1e9, c1 - c0, now() - t1);
        fflush(stdout);
    };
    const bool only_pwrite = argc > 4;
    for (int files : {1, 2, 3}) {
        run("pwrite", 0, files);
        if (only_pwrite) continue;
        run("mmap", 1, files);
        run("mmap+fa", 2, files);
    }
    for (int files : {1, 2, 3}) run("direct", 3, files);
    return 0;
}
