// tbk_sort.hip — device radix sort of 64-bit keys (rocPRIM): orders the dumped k-mer lists of the
// find-unique-kmers step lexicographically, as kmc_dump writes them.  A plain library operation
// outside every timed path.  rocPRIM is ROCm's own primitives library and takes the element count as
// a size_t: a dump is not limited to 2^31 keys.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

extern "C" hipError_t tbk_launch_sort_u64(const uint64_t *d_in, uint64_t *d_out, uint64_t n, int bits, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    size_t tmp_bytes = 0;
    void *d_tmp = nullptr;
    const size_t count = (size_t)n;
    hipError_t e = rocprim::radix_sort_keys(nullptr, tmp_bytes, d_in, d_out, count, 0u, (unsigned)bits, stream);
    if (e != hipSuccess) return e;
    e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_in, d_out, count, 0u, (unsigned)bits, stream);
    const hipError_t e2 = hipStreamSynchronize(stream);
    (void)hipFree(d_tmp);
    return e != hipSuccess ? e : e2;
}
