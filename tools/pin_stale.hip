// pin_stale.hip — does a host range that was registered, unregistered and unmapped leave anything behind?  After it, small hipHostMalloc
// buffers receive one byte from the device each; the byte must arrive.
//   hipcc -O2 --offload-arch=gfx950 -o tools/bin/pin_stale tools/pin_stale.hip && ./tools/bin/pin_stale
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: register + unregister + munmap first; 1: nothing first (control); 2: register, munmap WITHOUT unregister
    const size_t n = (size_t)128 << 20, HUGE = (size_t)2 << 20;
    CHECK(hipSetDevice(0));
    uint8_t *d = nullptr;
    CHECK(hipMalloc((void **)&d, n));
    CHECK(hipMemset(d, 'x', n));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uintptr_t lo = 0, hi = 0;
    for (int round = 0; round < 3 && mode != 1; round++) {
        void *m = mmap(nullptr, n + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        uint8_t *a = (uint8_t *)(((uintptr_t)m + HUGE - 1) & ~(uintptr_t)(HUGE - 1));
        (void)madvise(a, n, MADV_HUGEPAGE);
        for (size_t off = 0; off < n; off += 4096) a[off] = 0;
        CHECK(hipHostRegister(a, n, hipHostRegisterPortable));
        memset(a, 'G', n);
        CHECK(hipMemcpyAsync(d, a, n, hipMemcpyHostToDevice, st));
        CHECK(hipMemcpyAsync(a, d, n, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        if (mode == 0) CHECK(hipHostUnregister(a));
        munmap(m, n + HUGE);
        lo = (uintptr_t)a; hi = lo + n;
    }
    CHECK(hipMemset(d, 'x', n));
    CHECK(hipDeviceSynchronize());
    int wrong = 0, inside = 0;
    for (int i = 0; i < 400; i++) {
        const size_t sz = (i % 4 == 0) ? 4200 : (i % 4 == 1) ? 70000 : (i % 4 == 2) ? 600000 : 1900000;
        uint8_t *h = nullptr;
        CHECK(hipHostMalloc((void **)&h, sz, hipHostMallocPortable));
        h[0] = 'G';
        inside += (uintptr_t)h >= lo && (uintptr_t)h < hi;
        CHECK(hipMemcpyAsync(h, d, 1, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        if (h[0] != 'x') { if (wrong < 5) printf("  buffer %d (%zu bytes at %p%s): got %c\n", i, sz, (void *)h, ((uintptr_t)h >= lo && (uintptr_t)h < hi) ? ", inside the old range" : "", h[0]); wrong++; }
        CHECK(hipHostFree(h));
    }
    printf("mode %d: %d of 400 one-byte copies did not arrive; %d buffers lay inside the range registered before\n", mode, wrong, inside);
    return 0;
}
