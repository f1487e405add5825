// tbk_sort.hip — device radix sort of 64-bit keys (hipCUB / rocPRIM): orders the dumped k-mer
// lists of the find-unique-kmers step lexicographically, as kmc_dump writes them.  A plain
// library operation outside every timed path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

extern "C" hipError_t tbk_launch_sort_u64(const uint64_t *d_in, uint64_t *d_out, uint64_t n, int bits, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (n > 0x7FFFFFF0ull) return hipErrorInvalidValue;
    size_t tmp_bytes = 0;
    void *d_tmp = nullptr;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_in, d_out, (int)n, 0, bits, stream);
    if (e != hipSuccess) return e;
    e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_bytes, d_in, d_out, (int)n, 0, bits, stream);
    const hipError_t e2 = hipStreamSynchronize(stream);
    (void)hipFree(d_tmp);
    return e != hipSuccess ? e : e2;
}
