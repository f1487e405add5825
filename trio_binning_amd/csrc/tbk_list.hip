// tbk_list.hip — the k-mer list's text turned into packed keys on the GPU.
//
// Replaces the per-line work of create_kmer_hash_set (c/kmers.c:185-229: getline, kmer_to_int, add_to_hash)
// for lists in the shape every tool writes them: each line exactly k bytes and a newline (the last line may
// lack the newline).  The text crosses PCIe once, in pieces staged through pinned memory (tbk_host.cpp), and
// one thread per line packs its k bytes with the reference's rule (c/kmers.c:50-72: A=0 C=1 G=2 T=3, any other
// byte 0, base i at bits 2i).  A line that is not of that shape - a newline among its k bytes, anything but a
// newline behind them - raises a flag instead; the host then parses the file with the general parser, which
// implements the reference's getline rules line by line (parse_list).
#include <hip/hip_runtime.h>
#include <stdint.h>

// lines [0, n) of `text` (line i at byte i * (k + 1)); `open_end`: the piece's last line is the file's last
// line and has no newline behind it
__global__ void __launch_bounds__(256)
tbk_parse_lines_kernel(const uint8_t *__restrict__ text, uint64_t n, int k, int open_end, uint64_t *__restrict__ keys,
                       int *__restrict__ irregular) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = text + i * (uint64_t)(k + 1);
    uint64_t v = 0;
    bool bad = false;
    for (int j = 0; j < k; j++) {
        const uint32_t c = p[j];
        const uint64_t code = c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
        bad = bad || c == '\n';
        v |= code << (2 * j);
    }
    if (!(open_end && i + 1 == n) && p[k] != '\n') bad = true;
    keys[i] = v;
    if (bad) *irregular = 1;
}

extern "C" hipError_t tbk_launch_parse_lines(const uint8_t *d_text, uint64_t n, int k, int open_end, uint64_t *d_keys, int *d_irregular,
                                             hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tbk_parse_lines_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_text, n, k, open_end, d_keys, d_irregular);
    return hipGetLastError();
}
