// pin_cost.hip — what a piece of pinned host memory costs to get and to give back, by the way it is made (round 6: the reader's GPU
// windows and the writer's bin buffers are a tenth of a second per 600 MB, twice).
//   hipcc -O2 --offload-arch=gfx950 -o tools/bin/pin_cost tools/pin_cost.hip && ./tools/bin/pin_cost [MB]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 432, n = mb << 20;
    CHECK(hipSetDevice(0));
    void *d = nullptr;
    CHECK(hipMalloc(&d, n));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    auto copy_rate = [&](void *h) { const double t0 = now(); (void)hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); const double t1 = now();
                                    (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); const double t2 = now();
                                    printf("   D2H %.1f GB/s, H2D %.1f GB/s\n", n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9); };
    for (int rep = 0; rep < 2; rep++) {
        struct { const char *name; unsigned flags; } kinds[] = {{"hipHostMalloc portable", hipHostMallocPortable}, {"hipHostMalloc default", hipHostMallocDefault},
                                                                  {"hipHostMalloc non-coherent", hipHostMallocNonCoherent | hipHostMallocPortable}, {"hipHostMalloc numa-user", hipHostMallocNumaUser | hipHostMallocPortable}};
        for (auto &k : kinds) {
            void *h = nullptr;
            double t0 = now();
            hipError_t e = hipHostMalloc(&h, n, k.flags);
            double t1 = now();
            if (e != hipSuccess) { printf("%s: %s\n", k.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            memset(h, 1, n);
            double t2 = now();
            printf("%-28s %4zu MB: alloc %.1f ms, first touch %.1f ms", k.name, mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
            copy_rate(h);
            t0 = now();
            (void)hipHostFree(h);
            printf("   free %.1f ms\n", (now() - t0) * 1e3);
        }
        for (int huge = 0; huge < 2; huge++) {
            double t0 = now();
            void *h = mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            void *a = (void *)(((uintptr_t)h + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
            if (huge) (void)madvise(a, n, MADV_HUGEPAGE);
            memset(a, 1, n);
            double t1 = now();
            hipError_t e = hipHostRegister(a, n, hipHostRegisterPortable);
            double t2 = now();
            if (e != hipSuccess) { printf("hipHostRegister: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); munmap(h, n + (2 << 20)); continue; }
            printf("%-28s %4zu MB: mmap + touch %.1f ms, register %.1f ms", huge ? "mmap(THP) + hipHostRegister" : "mmap + hipHostRegister", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
            copy_rate(a);
            t0 = now();
            (void)hipHostUnregister(a);
            t1 = now();
            munmap(h, n + (2 << 20));
            printf("   unregister %.1f ms, munmap %.1f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
        }
    }
    return 0;
}
