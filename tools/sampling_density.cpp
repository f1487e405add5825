// What the sampling rule itself costs in lines per window (no GPU): random bases, every window's bucket by the probe's own rule
// (csrc/tbk_common.h: tbk_tmer_rank over the span's t-mer positions, the position mod w, the canonical m-mer's bucket), then
//   continuous   line switches of one walk over all windows: the density of the scheme as built (ties, canonical m-mers and all)
//   lanes        the kernel's decomposition: 32 windows per lane, even lanes up, odd lanes down; a lane's first window always
//                fetches; two lanes that START side by side on one bucket are one request (same wave instruction)
//   lanes_l2     the same, with the re-request of a line at a boundary where two lanes END side by side counted as an L2 hit
// against the 0.2105 of the formula (floor((l - t) / w) + 2) / (l - t + 2) and the 0.229 table lines per window the counters show.
//   g++ -O2 -std=c++17 -I trio_binning_amd/csrc -o /tmp/sampling_density tools/sampling_density.cpp && /tmp/sampling_density 21 6 1
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "tbk_common.h"

int main(int argc, char **argv) {
    const int k = argc > 1 ? atoi(argv[1]) : 21, w_want = argc > 2 ? atoi(argv[2]) : 6, span3 = argc > 3 ? atoi(argv[3]) : 1;
    const uint64_t n_bases = argc > 4 ? strtoull(argv[4], nullptr, 10) : 4000000;
    const uint32_t n_buckets = 261131725u;
    TbkMz z = tbk_mz_params(k, w_want, 300000000, 0, 1);
    if (span3) z = tbk_mz_span3(z);
    const int nt = tbk_mz_positions(z);
    std::mt19937_64 rng(12345);
    std::vector<uint8_t> b(n_bases);
    for (auto &x : b) x = (uint8_t)(rng() & 3);
    const uint64_t n_win = n_bases - k + 1, kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    std::vector<uint32_t> bucket(n_win);
    uint64_t fwd = 0;
    for (uint64_t i = 0; i < n_bases; i++) {
        fwd = (fwd >> 2) | ((uint64_t)b[i] << (2 * (k - 1)));   // base i is the window's LAST base: bits grow to the left as in the kernel's streams
        if (i + 1 < (uint64_t)k) continue;
        const uint64_t f = fwd & kmask, rc = tbk_revcomp_packed(f, k);
        uint32_t best = 0xFFFFFFFFu; int x = 0;
        for (int p = 0; p < nt; p++) { const uint32_t r = tbk_tmer_rank(f, z, p); if (r < best) { best = r; x = p; } }
        const int pos = x % z.w;
        const uint32_t mmask = z.m == 16 ? 0xFFFFFFFFu : ((1u << (2 * z.m)) - 1u);
        const uint32_t mx = (uint32_t)(f >> (2 * (z.o + pos))) & mmask, my = (uint32_t)(rc >> (2 * (z.o + z.w - 1 - pos))) & mmask;
        const uint32_t cm = mx < my ? mx : my;
        bucket[i + 1 - k] = (uint32_t)(((uint64_t)tbk_mmer_hash(cm) * n_buckets) >> 32);
    }
    uint64_t cont = 1;
    for (uint64_t i = 1; i < n_win; i++) cont += bucket[i] != bucket[i - 1];
    // lanes of 32 windows: lane L covers [32 L, 32 L + 31]; even lanes walk up, odd lanes down
    uint64_t lanes = 0, lanes_l2 = 0;
    const uint64_t n_lanes = n_win / 32;
    for (uint64_t L = 0; L < n_lanes; L++) {
        const uint64_t lo = 32 * L;
        uint64_t sw = 0;
        for (int j = 1; j < 32; j++) sw += bucket[lo + j] != bucket[lo + j - 1];
        const bool up = (L & 1) == 0;
        // the first window: even lane L starts at lo beside odd lane L - 1's start (lo - 1): one request if both name one bucket
        bool first_counts = true;
        if (up && L > 0 && bucket[lo] == bucket[lo - 1]) first_counts = false;   // (the odd lane L - 1 counted it)
        lanes += sw + (first_counts ? 1 : 0);
        lanes_l2 += sw + (first_counts ? 1 : 0);
        // where two lanes END side by side (even L's last window lo + 31, odd L + 1's last window lo + 32) on one bucket, the line was fetched twice
        if (up && L + 1 < n_lanes && bucket[lo + 31] == bucket[lo + 32]) lanes_l2 -= 1;
    }
    const uint64_t covered = n_lanes * 32;
    const int l = z.w + z.m - 1;
    printf("k=%d w=%d m=%d t=%d positions=%d: formula %.4f; continuous %.4f; lanes %.4f; lanes with end-to-end re-requests as L2 hits %.4f  (%llu windows)\n", k, z.w, z.m, z.t, nt,
           ((l - z.t) / z.w + 2) / (double)(l - z.t + 2), cont / (double)n_win, lanes / (double)covered, lanes_l2 / (double)covered, (unsigned long long)n_win);
    return 0;
}
