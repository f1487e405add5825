// tbk_gdeflate.hip — the gzip members of the bin writer, entropy-coded on the GPU.
//
// The reference writes its three bins through gzip.open unless --no-gzip-output (seq.py:132-134,
// classify_by_kmers.py:86-92): the DEFAULT mode compresses everything it writes.  tbk_deflate.cpp does that on the host
// the way FASTQ text wants it - no LZ77 (four-symbol noise and per-base qualities have nothing to match in 32 KiB), a
// byte histogram per line-aligned block, a canonical Huffman code, the dynamic-block header, the literals; runs of one
// byte as matches at distance 1 - and 16 CPUs do 9 GB/s of it, which is what a run with gzip'ed bins waited for while
// 256 CUs idled.  This file is the same coder as three kernels:
//
//   gd_encode_kernel   one workgroup (256 threads) per BLOCK of text (<= 32 KiB, cut at a line end by the host): the
//                      histogram by LDS atomics, the Huffman tree (rank sort over all threads, the two-queue merge on
//                      one lane), canonical codes, the RFC 1951 3.2.7 header, then every thread codes its own
//                      contiguous stretch of the block at the bit offset a block-wide scan gives it; the block is
//                      assembled in LDS (ds_or) and leaves for its slot in HBM in whole words.  A block ends with its
//                      end-of-block code and an EMPTY STORED block (00 00 FF FF behind the next byte boundary - what
//                      zlib's Z_SYNC_FLUSH writes), so every block starts on a byte and blocks are coded
//                      independently.  A block that would not shrink is written as a stored block.
//   gd_scan_kernel     one workgroup: where every block's bytes go in the dense output (gzip header in front of a
//                      member's first block, final empty block + CRC-32 + ISIZE behind its last).
//   gd_gather_kernel   one workgroup per block: slot -> dense output, and the members' framing bytes.
//
// The host (this file's tbk_gdeflate_*) cuts blocks, moves text in and members out through pinned memory on a stream of
// its own, three jobs deep (a job's text goes in while the previous job's members come out and the one before is
// written to the files), and fills in each member's CRC-32, which it sums (tbk_crc.cpp: PCLMULQDQ) while the device
// codes.  HBM roofline: the kernels read the text twice and write ~0.45 of it twice: ~3 B of HBM traffic per byte
// of text; PCIe carries 1 B in and ~0.45 B out per byte.  Any inflater reads the result; the decompressed bytes are
// what went in (tests/test_gpu_deflate.py: zlib on every member).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tbk.h"
#include "tbk_gdeflate.h"

extern "C" void tbk_set_error_(int, const char *msg);

namespace {

constexpr int GD_T = 256;                      // threads per workgroup
constexpr uint32_t GD_MAX_BLOCK = 16384;       // bytes of text per block (the LDS image of a coded block sets how many workgroups a CU holds: 16 KiB -> five)
constexpr uint32_t GD_HDR_WORDS = 160;         // a dynamic block's header: 17 bits, 19 x 3, at most 287 code-length tokens of at most 7 + 7 bits: under 520 bytes
constexpr uint32_t GD_NSYM = 288;              // literal/length alphabet (286 used)

struct GdBlock {
    uint64_t text_off;   // of the block's first byte in the job's text on the device
    uint64_t slot_off;   // of the block's slot in the slot buffer
    uint32_t n;          // bytes of text
    uint32_t member;     // which member it belongs to
    uint32_t flags;      // 1: first block of its member, 2: last
    uint32_t member_n;   // bytes of text of the whole member (ISIZE)
    uint64_t member_off; // of the member's first byte in the job's text (the CRC: how many bytes of the member follow a stretch)
};

// CRC-32 (the gzip polynomial, bit-reflected) of a concatenation from its parts' CRCs: crc(A || B) = crc(A) * x^(8 |B|) mod P + crc(B)
// (zlib's crc32_combine, in its multmodp / x2nmodp form).  Every lane sums its own stretch of a block with the byte table and
// multiplies the sum by x^(8 * bytes of the MEMBER behind the stretch); the member's CRC is the XOR of all of that (atomicXor per block).
constexpr uint32_t GD_POLY = 0xedb88320u;
__host__ __device__ inline uint32_t gd_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ GD_POLY : b >> 1;
    }
    return p;
}
struct GdX2n { uint32_t v[32]; };  // v[k] = x^(2^k) mod P
__host__ __device__ inline uint32_t gd_x2nmodp(const GdX2n &tab, uint64_t n, unsigned k) {  // x^(n * 2^k) mod P
    uint32_t p = 1u << 31;
    while (n) { if (n & 1u) p = gd_multmodp(tab.v[k & 31u], p); n >>= 1; k++; }
    return p;
}
inline GdX2n gd_x2n_table() {
    GdX2n t;
    uint32_t p = 1u << 30;  // x^1
    t.v[0] = p;
    for (int k = 1; k < 32; k++) t.v[k] = p = gd_multmodp(p, p);
    return t;
}

__host__ __device__ inline uint64_t gd_slot_bytes(uint32_t n) { return ((uint64_t)n + n / 8 + 1024 + 15) & ~(uint64_t)15; }

__device__ const uint16_t gd_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t gd_len_bits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};

// length symbol index (0..28) of a match length 3..258 (RFC 1951 3.2.5)
__device__ inline int gd_len_code(uint32_t len) {
    if (len == 258) return 28;
    int c = 27;
    while (gd_len_base[c] > len) c--;
    return c;
}

// The block's text in LDS: 256 stretches of `segw` words, one per lane, each padded to an ODD number of words so that the lanes'
// byte reads (lane t reads its own stretch front to back) fall into different banks.  The text is fetched from HBM once, in
// whole coalesced words; until round 6's first profile every lane read its stretch byte by byte from global memory, four times
// over - 64 cache lines per wave instruction, 130 k cycles of L1 per block: the kernel ran at the L1's pace (2.8 ms per 134 MB).
constexpr uint32_t GD_SEGW_MAX = ((GD_MAX_BLOCK + 3 + 3) / 4 + GD_T - 1) / GD_T;   // 17 words
constexpr uint32_t GD_TEXT_WORDS = GD_T * (GD_SEGW_MAX | 1u);

struct Lds {
    uint32_t text[GD_TEXT_WORDS];
    uint32_t hdr[GD_HDR_WORDS];   // the block header's bits (the codes go straight to the slot)
    uint32_t freq[GD_NSYM];
    uint32_t w[2 * GD_NSYM];
    uint16_t parent[2 * GD_NSYM];
    uint16_t order[GD_NSYM];
    uint16_t code[GD_NSYM];
    uint8_t len[GD_NSYM];
    uint8_t depth[2 * GD_NSYM];
    uint32_t bl_count[16], next_code[16];
    // the code-length code (19 symbols) and the header's token list
    uint32_t clfreq[19];
    uint8_t cllen[19];
    uint16_t clcode[19];
    uint32_t wave_sum[GD_T / 64];
    uint16_t hop[2 * GD_NSYM], dep[2 * GD_NSYM];   // pointer jumping over the tree: an ancestor of the node, and how many levels above
    uint16_t wcnt[5][16];                           // symbols per code length in each group of 64 symbols (canonical codes by ballot)
    unsigned long long nzmask[5];                   // which symbols have a code (the header's run-length pass walks the set bits)
    uint32_t m, deepest, same, header_bits, total_bits, stored;
};

// Code lengths (<= limit) for the symbols with freq > 0 among freq[0 .. nsym): every thread takes part.  At least two symbols have
// freq > 0.  Leaves sorted by (freq, symbol) - a rank sort over all lanes; the two-queue merge on one lane (inherently serial: m - 1
// steps, the queues' heads kept in registers); depths by pointer jumping over all lanes (log2 m rounds instead of 2m dependent LDS
// reads on one lane).  A tree deeper than the limit is not rebuilt: its over-long leaves are set to the limit and the Kraft sum
// is brought back to one by moving leaves one level down, one unit per move (zlib's gen_bitlen does the same), then the lengths
// are handed out again in order of frequency.
__device__ void gd_huffman(Lds &s, uint32_t *freq, int nsym, int limit, uint8_t *len) {
    const int t = threadIdx.x;
    if (t == 0) s.m = 0;
    __syncthreads();
    for (int a = t; a < nsym; a += GD_T) {
        const uint32_t f = freq[a];
        len[a] = 0;
        if (!f) continue;
        uint32_t r = 0;
        for (int b = 0; b < nsym; b++) { const uint32_t g = freq[b]; r += (g != 0 && (g < f || (g == f && b < a))) ? 1u : 0u; }
        s.order[r] = (uint16_t)a;
        atomicAdd(&s.m, 1u);
    }
    __syncthreads();
    const int m = (int)s.m, nodes = 2 * m - 1;
    for (int i = t; i < m; i += GD_T) s.w[i] = freq[s.order[i]];
    __syncthreads();
    if (t == 0) {
        int leaf = 0, inner = m, next = m;
        uint32_t wl = s.w[0], wi = 0xFFFFFFFFu;   // the queues' heads (all ones: empty)
        while (next < nodes) {
            uint32_t sum = 0;
            int pick[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (wl <= wi) { pick[q] = leaf++; sum += wl; wl = leaf < m ? s.w[leaf] : 0xFFFFFFFFu; }
                else { pick[q] = inner++; sum += wi; wi = inner < next ? s.w[inner] : 0xFFFFFFFFu; }
            }
            s.w[next] = sum;
            s.parent[pick[0]] = s.parent[pick[1]] = (uint16_t)next;
            if (inner == next) wi = sum;   // the inner queue was empty: the new node is its head
            next++;
        }
    }
    __syncthreads();
    // depth of every node: hop = an ancestor, dep = levels up to it; both double until every hop is the root
    const int root = nodes - 1;
    for (int i = t; i < nodes; i += GD_T) { s.hop[i] = (uint16_t)(i == root ? root : s.parent[i]); s.dep[i] = (uint16_t)(i == root ? 0 : 1); }
    __syncthreads();
    for (int span = 1; span < m; span <<= 1) {
        uint16_t nh[3], nd[3];
        int c = 0;
        for (int i = t; i < nodes; i += GD_T, c++) { const int h = s.hop[i]; nh[c] = s.hop[h]; nd[c] = (uint16_t)(s.dep[i] + s.dep[h]); }
        __syncthreads();
        c = 0;
        for (int i = t; i < nodes; i += GD_T, c++) { s.hop[i] = nh[c]; s.dep[i] = nd[c]; }
        __syncthreads();
    }
    if ((int)s.dep[0] <= limit) {   // (leaf 0 is the rarest symbol: no leaf lies deeper)
        for (int i = t; i < m; i += GD_T) len[s.order[i]] = (uint8_t)s.dep[i];
        __syncthreads();
        return;
    }
    if (t < 16) s.bl_count[t] = 0;
    if (t == 0) s.deepest = 0;
    __syncthreads();
    for (int i = t; i < m; i += GD_T) {
        const int d = (int)s.dep[i] < limit ? (int)s.dep[i] : limit;
        atomicAdd(&s.bl_count[d], 1u);
        atomicAdd(&s.deepest, 1u << (limit - d));   // the Kraft sum in units of 2^-limit
    }
    __syncthreads();
    if (t == 0) {
        for (uint32_t excess = s.deepest - (1u << limit); excess > 0; excess--) {
            int bits = limit - 1;
            while (s.bl_count[bits] == 0) bits--;
            s.bl_count[bits]--; s.bl_count[bits + 1] += 2; s.bl_count[limit]--;   // one leaf a level down, beside a leaf that comes up from the limit
        }
        int idx = 0;
        for (int bits = limit; bits >= 1; bits--)
            for (uint32_t c = s.bl_count[bits]; c > 0; c--) len[s.order[idx++]] = (uint8_t)bits;   // the rarest symbols get the longest codes
    }
    __syncthreads();
}

// canonical codes of the lengths, bit-reversed (DEFLATE packs a Huffman code from its most significant bit).  A symbol's code is
// the first code of its length plus the number of smaller symbols of that length: counted by wave ballots (fifteen per group of
// 64 symbols), not by every lane scanning the symbols below it.  Leaves s.bl_count[1 .. 15] = symbols per length.
__device__ void gd_codes(Lds &s, const uint8_t *len, int nsym, uint16_t *code) {
    const int t = threadIdx.x, lane = t & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    int l[2], rank[2] = {0, 0};
    for (int r = 0; r < 2; r++) {
        const int a = r * GD_T + t, group = r == 0 ? (t >> 6) : 4;
        l[r] = a < nsym ? (int)len[a] : 0;
        if (r == 1 && t >= 64) break;   // (symbols 256 .. 287: the first wave's second turn)
        for (int L = 1; L <= 15; L++) {
            const unsigned long long mk = __builtin_amdgcn_ballot_w64(l[r] == L);
            if (l[r] == L) rank[r] = __popcll(mk & below);
            if (lane == 0) s.wcnt[group][L] = (uint16_t)__popcll(mk);
        }
    }
    __syncthreads();
    if (t < 16) {
        uint32_t run = 0;
        for (int g = 0; g < 5; g++) { const uint32_t c = t ? s.wcnt[g][t] : 0; s.wcnt[g][t] = (uint16_t)run; run += c; }   // -> symbols of this length in earlier groups
        s.bl_count[t] = run;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t c = 0;
        s.bl_count[0] = 0;
        for (int L = 1; L <= 15; L++) { c = (c + s.bl_count[L - 1]) << 1; s.next_code[L] = c; }
    }
    __syncthreads();
    for (int r = 0; r < 2; r++) {
        const int a = r * GD_T + t, group = r == 0 ? (t >> 6) : 4;
        if (a >= nsym || (r == 1 && t >= 64)) break;
        if (!l[r]) { code[a] = 0; continue; }
        const uint32_t v = s.next_code[l[r]] + s.wcnt[group][l[r]] + (uint32_t)rank[r];
        code[a] = (uint16_t)(__brev(v) >> (32 - l[r]));
    }
    __syncthreads();
}

// One lane's bit writer.  LDS (the header, assembled before it is known whether the block is coded at all): every word by ds_or.
// GLOBAL (the codes, straight into the block's slot in HBM, which the job zeroed): a lane's stretch of bits is contiguous, so only
// its first and last word are shared with its neighbours - those go in by atomicOr, the words between by plain stores.
template <bool GLOBAL>
struct BitOutT {
    uint32_t *out;
    uint64_t acc;
    uint32_t word;   // index of the word acc's bit 0 belongs to
    int n;           // bits in acc
    bool first;      // the next word to leave is the stretch's first (it may hold a neighbour's bits)
    // CAPTURE: the stretch's first and last word are not written but kept - (head_w, head_v), (tail_w, tail_v) - for gd_merge_edges
    bool capture = false, has_head = false, has_tail = false;
    uint32_t head_w = 0, head_v = 0, tail_w = 0, tail_v = 0;
    __device__ BitOutT(uint32_t *o, uint32_t bitpos) : out(o), acc(0), word(bitpos >> 5), n((int)(bitpos & 31u)), first(true) {}
    __device__ inline void put(uint32_t bits, int len) {
        acc |= (uint64_t)bits << n;
        n += len;
        if (n >= 32) {
            if (GLOBAL && first && capture) { head_w = word; head_v = (uint32_t)acc; has_head = true; }
            else if (!GLOBAL || first) atomicOr(&out[word], (uint32_t)acc);
            else out[word] = (uint32_t)acc;
            first = false;
            acc >>= 32; n -= 32; word++;
        }
    }
    __device__ inline void finish() {
        if (n <= 0) return;
        if (!(GLOBAL && capture) && !(uint32_t)acc) return;   // (a captured border is kept even when its bits are all zero: the lanes' chain of borders must not break)
        if (GLOBAL && capture) {
            if (first) { head_w = word; head_v = (uint32_t)acc; has_head = true; }   // the whole stretch lies in one word
            else { tail_w = word; tail_v = (uint32_t)acc; has_tail = true; }
        } else atomicOr(&out[word], (uint32_t)acc);
    }
};

// The words at the borders of the lanes' stretches, merged in the wave's registers instead of by atomics at the memory side (where
// gfx950 performs device-scope atomics: 700 of them per block wrote 3.6 times the coded bytes).  A lane has a HEAD (its bits in
// the first word it touches) and, when its stretch goes on past that word, a TAIL (its bits in the last).  The heads of the lanes
// that begin in one word are ORed by a segmented suffix scan; the word is written ONCE - by the lane whose tail it is, else by the
// first lane that begins in it - with a plain store.  Only the words a wave shares with its neighbours (its first lane's head
// group, a tail nobody in the wave continues) still go in by atomicOr: a handful per block.
__device__ inline void gd_merge_edges(uint32_t *out, const BitOutT<true> &bo) {
    const int lane = threadIdx.x & 63;
    const uint32_t hw = bo.has_head ? bo.head_w : 0xFFFF0000u + (uint32_t)lane;   // (no head: a key nobody shares)
    uint32_t S = bo.has_head ? bo.head_v : 0u;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_down((int)S, d, 64), k = (uint32_t)__shfl_down((int)hw, d, 64);
        if (lane + d < 64 && k == hw) S |= o;
    }
    const uint32_t prev_hw = (uint32_t)__shfl_up((int)hw, 1, 64);
    const bool leader = bo.has_head && (lane == 0 || prev_hw != hw);
    const uint32_t tw = bo.has_tail ? bo.tail_w : 0xFFFE0000u + (uint32_t)lane;
    const uint32_t prev_tw = (uint32_t)__shfl_up((int)tw, 1, 64);
    const uint32_t next_hw = (uint32_t)__shfl_down((int)hw, 1, 64), next_S = (uint32_t)__shfl_down((int)S, 1, 64);
    const int next_leader = __shfl_down((int)leader, 1, 64);
    // A word is written EITHER by one plain store OR by atomics only - a plain store sits in this XCD's L2 until it is written back,
    // an atomic is performed at the memory side, and the write-back would bury it.  The words this wave may share with its
    // neighbours - where its first lane begins, where its last lane begins and ends - are the atomic ones.
    const uint32_t hw0 = (uint32_t)__shfl((int)hw, 0, 64), hw63 = (uint32_t)__shfl((int)hw, 63, 64), tw63 = (uint32_t)__shfl((int)tw, 63, 64);
    auto shared = [&](uint32_t w) { return w == hw0 || w == hw63 || w == tw63; };
    if (bo.has_tail) {
        const bool with_heads = lane < 63 && next_leader && next_hw == tw;   // the heads that begin in my tail's word travel with it
        const uint32_t v = bo.tail_v | (with_heads ? next_S : 0u);
        if (with_heads && !shared(tw)) out[tw] = v;
        else atomicOr(&out[tw], v);
    }
    if (leader && !(lane > 0 && prev_tw == hw)) {   // (else: the lane before has written this word with its tail)
        if (shared(hw)) atomicOr(&out[hw], S);
        else out[hw] = S;                            // the word begins with this lane's bits
    }
}

using BitOut = BitOutT<true>;
using BitOutLds = BitOutT<false>;

// The tokens of the stretch [lo, hi) of a block: literals, and - where the block is coded with runs - matches at
// distance 1.  A run never crosses a lane's stretch (<= 128 bytes, so a match is <= 128 long); a stretch that begins
// inside its predecessor's run continues it with a match of its own.  MODE 0: count symbols into freq; 1: sum the code
// bits; 2: write the codes.
template <int MODE>
// (src[i] is valid for lo <= i < hi: the lane's stretch in LDS; `prev` is the byte in front of it, -1 at the block's start)
__device__ inline void gd_tokens(Lds &s, const uint8_t *src, int prev, uint32_t lo, uint32_t hi, bool runs, uint32_t &bits, BitOut *bo) {
    auto lit = [&](uint32_t v) {
        if (MODE == 0) atomicAdd(&s.freq[v], 1u);
        else if (MODE == 1) bits += s.len[v];
        else bo->put(s.code[v], s.len[v]);
    };
    auto match = [&](uint32_t len) {
        const int c = gd_len_code(len);
        const uint32_t sym = 257u + (uint32_t)c;
        if (MODE == 0) atomicAdd(&s.freq[sym], 1u);
        else if (MODE == 1) bits += (uint32_t)s.len[sym] + gd_len_bits[c] + 1u;
        else {
            bo->put(s.code[sym], s.len[sym]);
            bo->put(((len - gd_len_base[c]) & 0x1Fu), gd_len_bits[c] + 1);  // the extra bits, then the one-bit distance code (0)
        }
    };
    if (!runs) {
        for (uint32_t i = lo; i < hi; i++) lit(src[i]);
        return;
    }
    uint32_t i = lo;
    if (i < hi && (int)src[i] == prev) {
        const uint8_t v = src[i];
        uint32_t run = 1;
        while (i + run < hi && src[i + run] == v) run++;
        if (run >= 3) match(run);
        else for (uint32_t r = 0; r < run; r++) lit(v);
        i += run;
    }
    while (i < hi) {
        const uint8_t v = src[i];
        uint32_t run = 1;
        while (i + run < hi && src[i + run] == v) run++;
        lit(v);
        const uint32_t left = run - 1;
        if (left >= 3) match(left);
        else for (uint32_t r = 0; r < left; r++) lit(v);
        i += run;
    }
}

__device__ inline uint32_t gd_block_exclusive_scan(uint32_t *wave_sum, uint32_t v, uint32_t &total) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t inc = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
    for (int wv = 0; wv < GD_T / 64; wv++) { if (wv < wave) base += wave_sum[wv]; all += wave_sum[wv]; }
    total = all;
    __syncthreads();
    return base + inc - v;
}

// One bit per byte of the job's text: is it a line end.  (A thread per 32 bytes; the text buffer is padded to a multiple of 32.)
__global__ void __launch_bounds__(GD_T)
gd_newline_kernel(const uint8_t *__restrict__ text, uint64_t n_words, uint32_t *__restrict__ bitmap) {
    const uint64_t wi = (uint64_t)blockIdx.x * GD_T + threadIdx.x;
    if (wi >= n_words) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(text + wi * 32);
    uint32_t bits = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint4 v = p[h];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = w[j] ^ 0x0A0A0A0Au;                                    // a zero byte where the text has '\n'
            const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 in every zero byte
            const uint32_t four = ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
            bits |= four << (h * 16 + j * 4);
        }
    }
    bitmap[wi] = bits;
}

// position of the first line end in text[start, start + count) relative to start, or -1: the whole WAVE looks, 64 words of the
// bitmap (2 KiB of text) per load and four loads in flight, and every lane returns the same answer
__device__ inline int64_t gd_find_nl(const uint32_t *__restrict__ bitmap, uint64_t start, uint64_t count) {
    if (!count) return -1;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t end = start + count, w0 = start >> 5, w_end = (end + 31) >> 5;
    for (uint64_t base = w0; base < w_end; base += 256) {
        uint32_t wv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint64_t wi = base + (uint64_t)q * 64 + lane;
            wv[q] = wi < w_end ? bitmap[wi] : 0u;
            if (wi == w0) wv[q] &= 0xFFFFFFFFu << (start & 31u);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint64_t hit = __builtin_amdgcn_ballot_w64(wv[q] != 0);
            if (hit) {
                const int f = __builtin_ctzll(hit);
                const uint32_t word = (uint32_t)__shfl((int)wv[q], f, 64);
                const uint64_t pos = ((base + (uint64_t)q * 64 + (uint64_t)f) << 5) + (uint64_t)(__ffs((int)word) - 1);
                return pos < end ? (int64_t)(pos - start) : -1;
            }
        }
    }
    return -1;
}

struct GdMember {
    uint64_t text_off;    // of the member's first byte in the job's text
    uint64_t slot_off;    // of its first block's slot
    uint32_t n;           // bytes of text
    uint32_t first_block; // its entries in the block table: [first_block, first_block + n / 8192 + 1)
};

// The blocks of every member (one WAVE per member: a block starts where the one before it ended, members do not wait for each
// other): tbk_deflate.cpp's rule with 32 KiB as the most.  Long lines (the second line of a block runs past 2 KiB): a block ends
// at the first line end past 8 KiB, so that the bases of a long read and its qualities get codes of their own; short lines:
// blocks of 32 KiB.  Entries of the table a member does not use keep n = 0.
__global__ void __launch_bounds__(64)
gd_cut_kernel(const GdMember *__restrict__ members, uint32_t n_members, const uint32_t *__restrict__ bitmap, GdBlock *__restrict__ blocks) {
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    if (j >= n_members) return;
    const GdMember mb = members[j];
    const uint32_t cap = mb.n / 8192 + 1;
    uint32_t k = 0;
    uint64_t slot = mb.slot_off;
    for (uint64_t off = 0; off < mb.n;) {
        uint64_t m = mb.n - off;
        const uint64_t at = mb.text_off + off;
        const int64_t first = gd_find_nl(bitmap, at, m < 2048 ? m : 2048);
        const uint64_t used = first >= 0 ? (uint64_t)first + 1 : m;
        const bool short_lines = first >= 0 && (used == m || gd_find_nl(bitmap, at + used, (m - used) < 2048 ? (m - used) : 2048) >= 0);
        const uint64_t least = short_lines ? GD_MAX_BLOCK : 8192, most = GD_MAX_BLOCK;
        if (m > least) {
            const uint64_t span = (m < most ? m : most) - least;
            const int64_t nl = gd_find_nl(bitmap, at + least, span);
            m = nl >= 0 ? least + (uint64_t)nl + 1 : least + span;
        }
        if (lane == 0) blocks[mb.first_block + k] = GdBlock{at, slot, (uint32_t)m, j, (k == 0 ? 1u : 0u) | (off + m == mb.n ? 2u : 0u), mb.n, mb.text_off};
        slot += gd_slot_bytes((uint32_t)m);
        k++;
        off += m;
    }
    for (uint32_t u = k + lane; u < cap; u += 64) blocks[mb.first_block + u] = GdBlock{0, 0, 0, j, 0, mb.n, mb.text_off};
}

// The members' CRC-32s: a lane per 64-byte chunk counted from the member's END (so that chunk c has exactly 64 c bytes behind it),
// slicing-by-4 over the chunk's aligned words, times x^(8 * 64 c) mod P from two tables of 128 (c = c0 + 128 c1; longer members
// go on with gd_x2nmodp), XOR over the wave, one atomicXor per wave.
struct GdCrcTabs { uint32_t lo[128], hi[128]; };   // lo[j] = x^(512 j), hi[j] = x^(512 * 128 j)  mod P

__global__ void __launch_bounds__(GD_T)
gd_crc_kernel(const uint8_t *__restrict__ text, const GdMember *__restrict__ members, const GdCrcTabs *__restrict__ tabs, GdX2n x2n,
              uint32_t *__restrict__ member_crc) {
    __shared__ uint32_t T[4][256];
    const int t = threadIdx.x;
    {
        uint32_t c = (uint32_t)t;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ GD_POLY : c >> 1;
        T[0][t] = c;
    }
    __syncthreads();
    for (int k = 1; k < 4; k++) { T[k][t] = (T[k - 1][t] >> 8) ^ T[0][T[k - 1][t] & 0xFFu]; __syncthreads(); }
    const GdMember mb = members[blockIdx.y];
    const uint64_t n_chunks = ((uint64_t)mb.n + 63) / 64;
    const uint64_t c = (uint64_t)blockIdx.x * GD_T + t;
    uint32_t part = 0;
    if (c < n_chunks) {
        const uint64_t end = (uint64_t)mb.n - c * 64, begin = end > 64 ? end - 64 : 0;   // the member's bytes [begin, end)
        const uint8_t *p = text + mb.text_off + begin;
        uint32_t len = (uint32_t)(end - begin), crc = 0xFFFFFFFFu;
        while (len && ((uintptr_t)p & 3u)) { crc = T[0][(crc ^ *p++) & 0xFFu] ^ (crc >> 8); len--; }
        for (; len >= 4; len -= 4, p += 4) {
            const uint32_t x = crc ^ *reinterpret_cast<const uint32_t *>(p);
            crc = T[3][x & 0xFFu] ^ T[2][(x >> 8) & 0xFFu] ^ T[1][(x >> 16) & 0xFFu] ^ T[0][x >> 24];
        }
        while (len) { crc = T[0][(crc ^ *p++) & 0xFFu] ^ (crc >> 8); len--; }
        uint32_t mult = gd_multmodp(tabs->lo[c & 127u], tabs->hi[(c >> 7) & 127u]);
        if (c >> 14) mult = gd_multmodp(mult, gd_x2nmodp(x2n, c >> 14, 3 + 6 + 14));
        part = gd_multmodp(mult, ~crc);
    }
    for (int d = 32; d >= 1; d >>= 1) part ^= (uint32_t)__shfl_xor((int)part, d, 64);
    if ((t & 63) == 0 && part) atomicXor(&member_crc[blockIdx.y], part);
}

#ifdef GD_PHASES
__device__ unsigned long long gd_phase[16];
#define GD_MARK(k) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); atomicAdd(&gd_phase[k], now_ - mark_); mark_ = now_; } } while (0)
#else
#define GD_MARK(k) do { } while (0)
#endif

__global__ void __launch_bounds__(GD_T)
gd_encode_kernel(const uint8_t *__restrict__ text, const GdBlock *__restrict__ blocks, uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes) {
    __shared__ Lds s;
    const int t = threadIdx.x;
    const GdBlock b = blocks[blockIdx.x];
    if (b.n == 0) { if (t == 0) sizes[blockIdx.x] = 0; return; }  // an entry its member did not need
#ifdef GD_PHASES
    unsigned long long mark_ = clock64();
#endif
    const uint8_t *gsrc = text + b.text_off;
    uint8_t *slot = slots + b.slot_off;
    const uint32_t n = b.n;
    // the text into LDS: the aligned words that cover it, `shift` bytes of the first one in front of the block
    const uint32_t shift = (uint32_t)(b.text_off & 3u), nw = (n + shift + 3) / 4;
    const uint32_t segw = (nw + GD_T - 1) / GD_T, stride = segw | 1u;
    {
        const uint32_t *g32 = reinterpret_cast<const uint32_t *>(text + (b.text_off & ~(uint64_t)3));
        for (uint32_t w = t; w < nw; w += GD_T) { const uint32_t owner = w / segw; s.text[owner * stride + (w - owner * segw)] = g32[w]; }
    }
    // lane t's stretch: the bytes of its words that belong to the block
    const uint32_t w_lo = (uint32_t)t * segw * 4, w_hi = w_lo + segw * 4;
    const uint32_t lo = w_lo > shift ? min(n, w_lo - shift) : 0u, hi = w_hi > shift ? min(n, w_hi - shift) : 0u;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(s.text) + (size_t)t * stride * 4 + shift - w_lo;   // src[i], lo <= i < hi (wraps below lo: never read there)

    for (uint32_t i = t; i < GD_HDR_WORDS; i += GD_T) s.hdr[i] = 0;
    for (uint32_t i = t; i < GD_NSYM; i += GD_T) s.freq[i] = 0;
    if (t == 0) { s.same = 0; s.stored = 0; }
    __syncthreads();
    GD_MARK(0);  // text staged, LDS cleared
    // bytes that repeat their predecessor: uniform random bases do a quarter of the time and runs of four or more cover
    // 1.6 % of them - nothing to gain; past 40 % there are real runs (constant or binned qualities, homopolymers)
    // (the byte in front of the stretch is the last byte of the lane before: its stretch is full when this one is not empty)
    const int prev = lo > 0 && lo < hi ? (int)reinterpret_cast<const uint8_t *>(s.text)[(size_t)(t - 1) * stride * 4 + segw * 4 - 1] : -1;
    uint32_t same = 0;
    int before = prev;
    for (uint32_t i = lo; i < hi; i++) {
        const int v = (int)src[i];
        same += v == before ? 1u : 0u;
        before = v;
    }
    if (same) atomicAdd(&s.same, same);
    __syncthreads();
    GD_MARK(1);  // same-as-predecessor pass
    const bool runs = (uint64_t)s.same * 5 > (uint64_t)n * 2;
    uint32_t bits = 0;
    gd_tokens<0>(s, src, prev, lo, hi, runs, bits, nullptr);
    if (t == 0) atomicAdd(&s.freq[256], 1u);  // end of block
    __syncthreads();
    int nlit = 257;
    if (runs) { nlit = 286; while (nlit > 257 && s.freq[nlit - 1] == 0) nlit--; }
    GD_MARK(2);  // histogram
    gd_huffman(s, s.freq, nlit, 15, s.len);
    GD_MARK(3);  // tree
    gd_codes(s, s.len, nlit, s.code);
    GD_MARK(4);  // codes

    // ---- the block header: RFC 1951 3.2.7 ----
    // The code lengths go out as code-length symbols: lengths as they are, runs of zeros as 17 (3-10) / 18 (11-138).  Which symbols
    // have a code is a bit mask (wave ballots); one lane walks its set bits - as many steps as there are symbols in use, not 258 -
    // once to count the zero-run symbols (the lengths' own counts are gd_codes' bl_count) and once to write.
    {
        const int lane = t & 63;
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(t < nlit && s.len[t] != 0);
        if (lane == 0) s.nzmask[t >> 6] = mk;
        if (t < 64) {
            const int a = GD_T + t;
            const unsigned long long mk2 = __builtin_amdgcn_ballot_w64((a < nlit && s.len[a] != 0) || (a == nlit && runs));   // (the one distance code: a bit where matches are used)
            if (lane == 0) s.nzmask[4] = mk2;
        }
    }
    __syncthreads();
    const int total = nlit + 1;  // + the one distance code
    // f(zero run before the symbol, the symbol or -1 for the end) over the symbols that have a code, in order
    auto walk = [&](auto &&on_zeros, auto &&on_length) {
        int prev = -1;
        for (int wv = 0; wv < 5; wv++) {
            unsigned long long mk = s.nzmask[wv];
            while (mk) {
                const int a = wv * 64 + __builtin_ctzll(mk);
                mk &= mk - 1;
                if (a - prev - 1 > 0) on_zeros(a - prev - 1);
                on_length(a);
                prev = a;
            }
        }
        if (total - prev - 1 > 0) on_zeros(total - prev - 1);
    };
    if (t == 0) {
        uint32_t c0 = 0, c17 = 0, c18 = 0;
        walk([&](int z) { while (z >= 11) { const int r = z < 138 ? z : 138; c18++; z -= r; } if (z >= 3) { c17++; z = 0; } c0 += (uint32_t)z; }, [&](int) {});
        for (int k = 1; k <= 15; k++) s.clfreq[k] = s.bl_count[k];
        if (runs) s.clfreq[1]++;   // the distance code's length
        s.clfreq[0] = c0; s.clfreq[16] = 0; s.clfreq[17] = c17; s.clfreq[18] = c18;
        int distinct = 0;
        for (int k = 0; k < 19; k++) distinct += s.clfreq[k] != 0;
        if (distinct < 2) s.clfreq[s.clfreq[0] ? 1 : 0]++;  // a code needs two symbols to be complete
    }
    __syncthreads();
    gd_huffman(s, s.clfreq, 19, 7, s.cllen);
    gd_codes(s, s.cllen, 19, s.clcode);
    if (t == 0) {
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19;
        while (hclen > 4 && s.cllen[order[hclen - 1]] == 0) hclen--;
        BitOutLds bo(s.hdr, 0);
        bo.put(0u, 1);                        // not the final block: the member ends with an empty stored block of its own
        bo.put(2u, 2);                        // dynamic Huffman codes
        bo.put((uint32_t)(nlit - 257), 5);    // HLIT
        bo.put(0u, 5);                        // HDIST: 1 distance code
        bo.put((uint32_t)(hclen - 4), 4);
        for (int k = 0; k < hclen; k++) bo.put(s.cllen[order[k]], 3);
        const uint32_t z_code = s.clcode[0], z_len = s.cllen[0], c17 = s.clcode[17], l17 = s.cllen[17], c18 = s.clcode[18], l18 = s.cllen[18];
        walk([&](int z) {
                 while (z >= 11) { const int r = z < 138 ? z : 138; bo.put(c18, (int)l18); bo.put((uint32_t)(r - 11), 7); z -= r; }
                 if (z >= 3) { bo.put(c17, (int)l17); bo.put((uint32_t)(z - 3), 3); z = 0; }
                 for (; z > 0; z--) bo.put(z_code, (int)z_len);
             },
             [&](int a) { const int L = a < nlit ? (int)s.len[a] : 1; bo.put(s.clcode[L], s.cllen[L]); });
        s.header_bits = bo.word * 32u + (uint32_t)bo.n;
        bo.finish();
    }
    __syncthreads();

    GD_MARK(5);  // header (token list, code-length code, bits)
    // ---- where every lane's codes go; does the block shrink at all ----
    bits = 0;
    gd_tokens<1>(s, src, prev, lo, hi, runs, bits, nullptr);
    uint32_t total_bits = 0;
    const uint32_t my_bit = gd_block_exclusive_scan(s.wave_sum, bits, total_bits);
    GD_MARK(6);  // bit counting + scan
    const uint32_t end_bits = s.header_bits + total_bits + s.len[256];   // behind the end-of-block code
    const uint32_t coded_bytes = (end_bits + 3 + 7) / 8 + 4;             // + the empty stored block: 3 header bits, pad, 00 00 FF FF
    if (coded_bytes >= n + 5) {
        // a stored block (starts on a byte, ends on a byte): 00, LEN, ~LEN, the bytes
        if (t == 0) {
            slot[0] = 0; slot[1] = (uint8_t)(n & 0xFF); slot[2] = (uint8_t)(n >> 8); slot[3] = (uint8_t)(~n & 0xFF); slot[4] = (uint8_t)((~n >> 8) & 0xFF);
            sizes[blockIdx.x] = n + 5;
        }
        for (uint32_t i = t; i < n; i += GD_T) slot[5 + i] = gsrc[i];
        return;
    }
    uint32_t *slot32 = reinterpret_cast<uint32_t *>(slot);  // (slots start on 16 bytes and are zero: the job's memset)
    {
        // the header's words: whole ones by plain stores, the last (it shares its word with lane 0's first codes) by OR
        const uint32_t whole = s.header_bits / 32;
        for (uint32_t i = t; i < whole; i += GD_T) slot32[i] = s.hdr[i];
        if (t == 0 && (s.header_bits & 31u)) atomicOr(&slot32[whole], s.hdr[whole]);
        BitOut bo(slot32, s.header_bits + my_bit);
        bo.capture = true;
        gd_tokens<2>(s, src, prev, lo, hi, runs, bits, &bo);
        if (lo < hi && hi == n) {
            // the lane with the block's last bytes goes on: the end-of-block code, then the empty stored block - BFINAL = 0, BTYPE = 00,
            // zero bits up to the byte, LEN = 0, NLEN = FFFF - so that these bits are part of ITS stretch (no second writer in its words)
            bo.put(s.code[256], s.len[256]);
            bo.put(0u, 3);
            const uint32_t at_bit = bo.word * 32u + (uint32_t)bo.n;
            bo.put(0u, (int)((8u - (at_bit & 7u)) & 7u));
            bo.put(0u, 16);
            bo.put(0xFFFFu, 16);
        }
        bo.finish();
        gd_merge_edges(slot32, bo);
    }
    if (t == 0) sizes[blockIdx.x] = coded_bytes;
    GD_MARK(7);  // codes out
}

// dense[] = for every member: 10 bytes of gzip header, its blocks back to back, the final empty stored block (5 bytes),
// CRC-32 and ISIZE (8 bytes).  offsets[b]: where block b's bytes start; member_end[j]: where member j ends.
__global__ void __launch_bounds__(GD_T)
gd_scan_kernel(const GdBlock *__restrict__ blocks, const uint32_t *__restrict__ sizes, uint32_t n_blocks, uint64_t *__restrict__ offsets,
               uint64_t *__restrict__ member_end, uint32_t n_members) {
    __shared__ uint32_t wave_sum[GD_T / 64];
    __shared__ uint64_t carry;
    __shared__ uint32_t used;
    const int t = threadIdx.x;
    if (t == 0) { carry = 0; used = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += GD_T) {
        const uint32_t i = base + t;
        uint32_t v = 0, flags = 0;
        if (i < n_blocks && blocks[i].n) { flags = blocks[i].flags; v = sizes[i] + ((flags & 1u) ? 10u : 0u) + ((flags & 2u) ? 13u : 0u); }
        uint32_t total = 0;
        const uint32_t ex = gd_block_exclusive_scan(wave_sum, v, total);
        const uint64_t start = carry + ex;
        if (i < n_blocks && v) {
            atomicAdd(&used, 1u);
            offsets[i] = start + ((flags & 1u) ? 10u : 0u);
            if (flags & 2u) member_end[blocks[i].member] = start + v;
        }
        __syncthreads();
        if (t == 0) carry += total;
        __syncthreads();
    }
    if (t == 0) member_end[n_members] = used;  // (behind the members' ends: how many blocks the job came to)
}

__global__ void __launch_bounds__(GD_T)
gd_gather_kernel(const GdBlock *__restrict__ blocks, const uint32_t *__restrict__ sizes, const uint64_t *__restrict__ offsets, const uint8_t *__restrict__ slots,
                 uint8_t *__restrict__ dense) {
    const int t = threadIdx.x;
    const GdBlock b = blocks[blockIdx.x];
    if (b.n == 0) return;
    const uint32_t size = sizes[blockIdx.x];
    const uint8_t *src = slots + b.slot_off;
    uint8_t *dst = dense + offsets[blockIdx.x];
    // (the slot starts on 16 bytes, the destination anywhere: words where it happens to be aligned, bytes where not)
    if ((((uintptr_t)dst) & 3u) == 0) {
        const uint32_t words = size / 4;
        for (uint32_t i = t; i < words; i += GD_T) reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
        for (uint32_t i = words * 4 + t; i < size; i += GD_T) dst[i] = src[i];
    } else {
        for (uint32_t i = t; i < size; i += GD_T) dst[i] = src[i];
    }
    if ((b.flags & 1u) && t < 10) {
        const uint8_t head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
        dst[(int)t - 10] = head[t];
    }
    if ((b.flags & 2u) && t < 13) {
        // BFINAL = 1, BTYPE = 00 (one byte with the padding), LEN = 0, NLEN = FFFF; the CRC-32 is the host's; ISIZE
        const uint8_t tail[13] = {0x01, 0x00, 0x00, 0xff, 0xff, 0, 0, 0, 0, (uint8_t)(b.member_n & 0xFF), (uint8_t)((b.member_n >> 8) & 0xFF),
                                  (uint8_t)((b.member_n >> 16) & 0xFF), (uint8_t)((b.member_n >> 24) & 0xFF)};
        dst[size + t] = tail[t];
    }
}

int gfail(int code, const char *what, hipError_t e) {
    char buf[256];
    snprintf(buf, sizeof buf, "GPU gzip encoder: %s: %s", what, hipGetErrorString(e));
    tbk_set_error_(code, buf);
    (void)hipGetLastError();
    return code;
}

extern "C" void *tbk_pin_alloc_(size_t bytes);   // tbk_host.cpp
extern "C" void tbk_pin_free_(void *p);

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n, bool exact = false) {   // (exact: the caller has put its slack into n already)
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t c = exact ? n + 4096 : n + n / 4 + 4096;
        const hipError_t e = hipMalloc(&p, c);
        if (e == hipSuccess) cap = c;
        return e;
    }
    void drop() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n, bool exact = false) {
        if (n <= cap) return hipSuccess;
        if (p) tbk_pin_free_(p);
        p = nullptr; cap = 0;
        const size_t c = exact ? n + 4096 : n + n / 4 + 4096;
        p = tbk_pin_alloc_(c);   // (huge pages registered with the runtime: a third of hipHostMalloc's cost, tbk_host.cpp)
        if (!p) return hipErrorOutOfMemory;
        cap = c;
        return hipSuccess;
    }
    void drop() { if (p) tbk_pin_free_(p); p = nullptr; cap = 0; }
};

struct Job {
    int state = 0;  // 0 free, 1 coding (text in, kernels, the members' ends on their way home), 2 the members on their way home
    DevBuf d_text, d_bitmap, d_members, d_blocks, d_slots, d_sizes, d_offsets, d_member_end, d_dense;  // (d_member_end: the members' ends, the block count, then the members' CRCs)
    PinBuf h_members, h_member_end, h_dense;
    hipEvent_t text_in = nullptr, meta_home = nullptr, dense_home = nullptr;
    std::vector<int> tags;          // per member: the caller's tag (the bin)
    std::vector<uint32_t> crc;      // per member, when the caller has summed it itself (tbk_gdeflate_set_crc); else the device's is used
    std::vector<uint8_t> crc_set;
    std::vector<uint64_t> text_len; // per member
    uint64_t dense_bytes = 0;
    void drop() {
        d_text.drop(); d_bitmap.drop(); d_members.drop(); d_blocks.drop(); d_slots.drop(); d_sizes.drop(); d_offsets.drop(); d_member_end.drop(); d_dense.drop();
        h_members.drop(); h_member_end.drop(); h_dense.drop();
        if (text_in) (void)hipEventDestroy(text_in);
        if (meta_home) (void)hipEventDestroy(meta_home);
        if (dense_home) (void)hipEventDestroy(dense_home);
    }
};

}  // namespace

struct tbk_gdeflate {
    int device = 0;
    // kernels; the text's way in; the members' way out - three streams, so that a job's text arrives and the previous job's members
    // leave while the kernels of the one between them run
    hipStream_t stream = nullptr, stream_in = nullptr, stream_out = nullptr;
    GdX2n x2n = gd_x2n_table();
    GdCrcTabs *d_crc_tabs = nullptr;
    Job jobs[3];
    uint64_t submitted = 0;  // jobs so far; job k lives in jobs[k % 3]
    uint64_t collected = 0;
    // totals (tbk_gdeflate_stats)
    uint64_t text_bytes = 0, member_bytes = 0, blocks = 0, members = 0;
};

int tbk_gdeflate_create(int device, tbk_gdeflate **out) {
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); tbk_set_error_(TBK_ERR_NO_DEVICE, "GPU gzip encoder: no such device"); return TBK_ERR_NO_DEVICE; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "hipSetDevice", e);
    tbk_gdeflate *g = new tbk_gdeflate();
    g->device = device;
    e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream_out, hipStreamNonBlocking);
    for (Job &j : g->jobs) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&j.text_in, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&j.meta_home, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&j.dense_home, hipEventDisableTiming);
    }
    if (e == hipSuccess) {
        GdCrcTabs tabs;
        for (uint32_t j = 0; j < 128; j++) { tabs.lo[j] = gd_x2nmodp(g->x2n, j, 3 + 6); tabs.hi[j] = gd_x2nmodp(g->x2n, j, 3 + 6 + 7); }
        e = hipMalloc((void **)&g->d_crc_tabs, sizeof tabs);
        if (e == hipSuccess) e = hipMemcpy(g->d_crc_tabs, &tabs, sizeof tabs, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { tbk_gdeflate_destroy(g); return gfail(TBK_ERR_HIP, "setup", e); }
    *out = g;
    return TBK_OK;
}

void tbk_gdeflate_destroy(tbk_gdeflate *g) {
    if (!g) return;
    if (hipSetDevice(g->device) == hipSuccess) {
        if (g->stream) (void)hipStreamSynchronize(g->stream);
        if (g->stream_in) { (void)hipStreamSynchronize(g->stream_in); (void)hipStreamDestroy(g->stream_in); }
        if (g->stream_out) { (void)hipStreamSynchronize(g->stream_out); (void)hipStreamDestroy(g->stream_out); }
        for (Job &j : g->jobs) j.drop();
        if (g->d_crc_tabs) (void)hipFree(g->d_crc_tabs);
        if (g->stream) (void)hipStreamDestroy(g->stream);
    }
    delete g;
}

int tbk_gdeflate_submit(tbk_gdeflate *g, const tbk_gdeflate_member *members, size_t n_members) {
    if (!g || (n_members && !members)) { tbk_set_error_(TBK_ERR_INVALID, "GPU gzip encoder: NULL argument"); return TBK_ERR_INVALID; }
    Job &j = g->jobs[g->submitted % 3];
    if (j.state != 0) { tbk_set_error_(TBK_ERR_STATE, "GPU gzip encoder: three jobs in flight; collect first"); return TBK_ERR_STATE; }
    hipError_t e = hipSetDevice(g->device);
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "hipSetDevice", e);
    j.tags.clear(); j.crc.assign(n_members, 0); j.crc_set.assign(n_members, 0); j.text_len.clear();
    // The host never reads the text (it is 30 GB a run): where the line ends are is found on the device (gd_newline_kernel), the
    // blocks are cut there (gd_cut_kernel).  Here: only what follows from the members' lengths - where a member's text, its
    // entries of the block table (as many as it could need: a block is 8 KiB or more, but for a member's last) and its slots begin.
    uint64_t text_total = 0, slot_total = 0, nb = 0;
    e = j.h_members.need((n_members + 1) * sizeof(GdMember));
    if (e != hipSuccess) return gfail(TBK_ERR_NOMEM, "buffers", e);
    GdMember *hm = (GdMember *)j.h_members.p;
    for (size_t i = 0; i < n_members; i++) {
        if (members[i].n > 0xFFFFFFF0ull) { tbk_set_error_(TBK_ERR_INVALID, "GPU gzip encoder: a member of 4 GiB or more"); return TBK_ERR_INVALID; }
        j.tags.push_back(members[i].tag);
        j.text_len.push_back(members[i].n);
        const uint64_t entries = members[i].n / 8192 + 1;
        hm[i] = GdMember{text_total, slot_total, (uint32_t)members[i].n, (uint32_t)nb};
        text_total += members[i].n;
        slot_total += (((uint64_t)members[i].n + members[i].n / 8 + 15) & ~(uint64_t)15) + entries * 1040;
        nb += entries;
    }
    if (n_members > 65535) { tbk_set_error_(TBK_ERR_INVALID, "GPU gzip encoder: more than 65535 members in one job"); return TBK_ERR_INVALID; }
    if (nb > 0x7FFFFFF0ull) { tbk_set_error_(TBK_ERR_INVALID, "GPU gzip encoder: too much text in one job"); return TBK_ERR_INVALID; }
    const uint64_t n_words = (text_total + 31) / 32;
    const uint64_t dense_cap = slot_total + 23 * (uint64_t)n_members + 64;
    e = j.d_text.need(n_words * 32 + 64);
    if (e == hipSuccess) e = j.d_bitmap.need((n_words + 2) * 4);
    if (e == hipSuccess) e = j.d_members.need((n_members + 1) * sizeof(GdMember));
    if (e == hipSuccess) e = j.d_blocks.need((nb + 1) * sizeof(GdBlock));
    if (e == hipSuccess) e = j.d_slots.need(slot_total + 64);
    if (e == hipSuccess) e = j.d_sizes.need((nb + 1) * 4);
    if (e == hipSuccess) e = j.d_offsets.need((nb + 1) * 8);
    if (e == hipSuccess) e = j.d_member_end.need((n_members + 1) * 8 + (n_members + 1) * 4);
    if (e == hipSuccess) e = j.d_dense.need(dense_cap);
    if (e == hipSuccess) e = j.h_member_end.need((n_members + 1) * 8 + (n_members + 1) * 4);
    if (e == hipSuccess) e = j.h_dense.need(dense_cap);
    if (e != hipSuccess) return gfail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "buffers", e);
    // the text: members that lie back to back in the caller's memory (a bin's pieces do) go in one copy
    uint64_t at = 0;
    for (size_t i = 0; i < n_members && e == hipSuccess;) {
        size_t k = i;
        uint64_t run = members[i].n;
        while (k + 1 < n_members && members[k + 1].src == members[k].src + members[k].n) { k++; run += members[k].n; }
        if (run) e = hipMemcpyAsync((uint8_t *)j.d_text.p + at, members[i].src, run, hipMemcpyHostToDevice, g->stream_in);
        at += run;
        i = k + 1;
    }
    if (e == hipSuccess && (text_total & 31)) e = hipMemsetAsync((uint8_t *)j.d_text.p + text_total, 0, 32 - (text_total & 31), g->stream_in);  // (the bitmap's last word reads whole)
    if (e == hipSuccess && n_members) e = hipMemcpyAsync(j.d_members.p, j.h_members.p, n_members * sizeof(GdMember), hipMemcpyHostToDevice, g->stream_in);
    if (e == hipSuccess) e = hipEventRecord(j.text_in, g->stream_in);
    if (e == hipSuccess) e = hipStreamWaitEvent(g->stream, j.text_in, 0);
    if (e == hipSuccess) e = hipMemsetAsync(j.d_member_end.p, 0, (n_members + 1) * 12, g->stream);
    if (e == hipSuccess && slot_total) e = hipMemsetAsync(j.d_slots.p, 0, slot_total, g->stream);  // (the codes are ORed and stored into zeroed slots)
    if (e == hipSuccess && text_total) {
        hipLaunchKernelGGL(gd_newline_kernel, dim3((unsigned)((n_words + GD_T - 1) / GD_T)), dim3(GD_T), 0, g->stream, (const uint8_t *)j.d_text.p, n_words, (uint32_t *)j.d_bitmap.p);
        hipLaunchKernelGGL(gd_cut_kernel, dim3((unsigned)n_members), dim3(64), 0, g->stream, (const GdMember *)j.d_members.p, (uint32_t)n_members,
                           (const uint32_t *)j.d_bitmap.p, (GdBlock *)j.d_blocks.p);
        uint64_t longest = 0;
        for (size_t i = 0; i < n_members; i++) longest = std::max<uint64_t>(longest, members[i].n);
        hipLaunchKernelGGL(gd_crc_kernel, dim3((unsigned)(((longest + 63) / 64 + GD_T - 1) / GD_T), (unsigned)n_members), dim3(GD_T), 0, g->stream, (const uint8_t *)j.d_text.p,
                           (const GdMember *)j.d_members.p, (const GdCrcTabs *)g->d_crc_tabs, g->x2n, (uint32_t *)((uint64_t *)j.d_member_end.p + n_members + 1));
        hipLaunchKernelGGL(gd_encode_kernel, dim3((unsigned)nb), dim3(GD_T), 0, g->stream, (const uint8_t *)j.d_text.p, (const GdBlock *)j.d_blocks.p, (uint8_t *)j.d_slots.p,
                           (uint32_t *)j.d_sizes.p);
        hipLaunchKernelGGL(gd_scan_kernel, dim3(1), dim3(GD_T), 0, g->stream, (const GdBlock *)j.d_blocks.p, (const uint32_t *)j.d_sizes.p, (uint32_t)nb, (uint64_t *)j.d_offsets.p,
                           (uint64_t *)j.d_member_end.p, (uint32_t)n_members);
        hipLaunchKernelGGL(gd_gather_kernel, dim3((unsigned)nb), dim3(GD_T), 0, g->stream, (const GdBlock *)j.d_blocks.p, (const uint32_t *)j.d_sizes.p,
                           (const uint64_t *)j.d_offsets.p, (const uint8_t *)j.d_slots.p, (uint8_t *)j.d_dense.p);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(j.h_member_end.p, j.d_member_end.p, (n_members + 1) * 12, hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipEventRecord(j.meta_home, g->stream);
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "submit", e);
    j.state = 1;
    g->submitted++;
    g->text_bytes += text_total; g->members += n_members;
    return TBK_OK;
}

// back = 0: the newest job's text has left the caller's buffers; 1: the one before it; ...
int tbk_gdeflate_text_done(tbk_gdeflate *g, int back) {
    if (!g || g->submitted <= (uint64_t)back) return TBK_OK;
    Job &j = g->jobs[(g->submitted - 1 - (uint64_t)back) % 3];
    const hipError_t e = hipEventSynchronize(j.text_in);
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "waiting for the text's copy", e);
    return TBK_OK;
}

void tbk_gdeflate_set_crc(tbk_gdeflate *g, size_t member, uint32_t crc) {
    if (!g || g->submitted == 0) return;
    Job &j = g->jobs[(g->submitted - 1) % 3];
    if (member < j.crc.size()) { j.crc[member] = crc; j.crc_set[member] = 1; }
}

// Moves the pipeline on: the job before the newest starts its members' journey home; the oldest job in flight (if it has
// come that far, or `drain`) is handed out.  *out: its members in order - (tag, bytes, length) - valid until the next call.
int tbk_gdeflate_collect(tbk_gdeflate *g, bool drain, std::vector<tbk_gdeflate_out> &out) {
    out.clear();
    if (!g) return TBK_OK;
    hipError_t e = hipSetDevice(g->device);
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "hipSetDevice", e);
    // stage B: every job whose coding has been submitted and whose successor exists (or drain) sends its members home
    for (uint64_t k = g->collected; k < g->submitted; k++) {
        Job &j = g->jobs[k % 3];
        if (j.state != 1) continue;
        if (!drain && k + 1 >= g->submitted) break;  // the newest job: its kernels have only just been queued
        e = hipEventSynchronize(j.meta_home);
        if (e != hipSuccess) return gfail(TBK_ERR_HIP, "waiting for a job", e);
        const size_t nm = j.tags.size();
        const uint64_t *ends = (const uint64_t *)j.h_member_end.p;
        g->blocks += ends[nm];
        uint64_t total = 0;
        // empty members have no block: their bytes are put in by the host below, behind the device's
        for (size_t i = 0; i < nm; i++) if (j.text_len[i]) total = std::max<uint64_t>(total, ends[i]);
        j.dense_bytes = total;
        if (total) e = hipMemcpyAsync(j.h_dense.p, j.d_dense.p, total, hipMemcpyDeviceToHost, g->stream_out);  // (the job's kernels are done: meta_home)
        if (e == hipSuccess) e = hipEventRecord(j.dense_home, g->stream_out);
        if (e != hipSuccess) return gfail(TBK_ERR_HIP, "fetching the members", e);
        j.state = 2;
    }
    // stage C: the oldest job, when its members are home (or drain)
    if (g->collected < g->submitted) {
        Job &j = g->jobs[g->collected % 3];
        if (j.state == 2 && (drain || g->collected + 2 < g->submitted || hipEventQuery(j.dense_home) == hipSuccess)) {
            e = hipEventSynchronize(j.dense_home);
            if (e != hipSuccess) return gfail(TBK_ERR_HIP, "waiting for the members", e);
            const size_t nm = j.tags.size();
            const uint64_t *ends = (const uint64_t *)j.h_member_end.p;
            uint8_t *dense = (uint8_t *)j.h_dense.p;
            uint64_t start = 0, spare = j.dense_bytes;
            for (size_t i = 0; i < nm; i++) {
                if (j.text_len[i] == 0) {
                    // an empty member (an empty bin's file is a valid gzip file, as gzip.open(...).close() leaves it)
                    static const uint8_t empty[23] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff, 0x01, 0x00, 0x00, 0xff, 0xff, 0, 0, 0, 0, 0, 0, 0, 0};
                    memcpy(dense + spare, empty, 23);
                    out.push_back(tbk_gdeflate_out{j.tags[i], (const char *)dense + spare, 23});
                    spare += 23;
                    g->member_bytes += 23;
                    continue;
                }
                const uint64_t end = ends[i];
                const uint32_t *dev_crc = (const uint32_t *)(ends + nm + 1);  // what the encode kernel summed (gd_multmodp)
                const uint32_t crc = j.crc_set[i] ? j.crc[i] : dev_crc[i];
                memcpy(dense + end - 8, &crc, 4);
                out.push_back(tbk_gdeflate_out{j.tags[i], (const char *)dense + start, (size_t)(end - start)});
                g->member_bytes += end - start;
                start = end;
            }
            j.state = 0;
            g->collected++;
        }
    }
    return TBK_OK;
}

int tbk_gdeflate_in_flight(const tbk_gdeflate *g) { return g ? (int)(g->submitted - g->collected) : 0; }

void tbk_gdeflate_stats(const tbk_gdeflate *g, uint64_t *text_bytes, uint64_t *member_bytes, uint64_t *blocks, uint64_t *members) {
    if (text_bytes) *text_bytes = g ? g->text_bytes : 0;
    if (member_bytes) *member_bytes = g ? g->member_bytes : 0;
    if (blocks) *blocks = g ? g->blocks : 0;
    if (members) *members = g ? g->members : 0;
}


// =====================================================================================================================================
// The other direction: bgzf input inflated on the GPU (tbk_ginflate_*; the reader's bgzf path, tbk_fastx.cpp).
// =====================================================================================================================================
// A .fastq.gz written by bgzip / htslib is a chain of independent gzip members of at most 64 KiB of text (the BC extra field says how
// long each one is): the reference reads it through gzip.open like any other (seq.py:86-92), this reader inflated its blocks side by
// side on the host's threads - 6.5 GB/s of text on 16 of them, what a run from bgzf input waited for.  Here ONE WAVE inflates a block,
// some four thousand blocks to a window.  The decoder is written uniformly - every lane runs the same control flow on the same values,
// and the values are kept in scalar registers (readfirstlane behind the wave index and behind every LDS read: the compiler cannot
// know they are uniform); the Huffman tables of the current deflate block live in LDS (built by the wave: canonical order by one
// lane, the 10-bit / 9-bit look-up tables by all); a literal is one store (every lane writes the same byte to the same place: no lane
// mask to set up); a match is copied by the lanes side by side (dst[i] = src[i mod dist]).  22-25 GB/s of text, bound by the 56 scalar
// instructions a symbol costs (tools/ginflate_gate.hip: the measurement this was built on, and three decoders that were not faster).
// Every block's CRC-32 is summed on the device (gd_crc_kernel) and compared with its trailer's; a block that does not decode, or whose
// CRC differs, is counted and the caller told.
constexpr int GI_FAST_BITS = 10, GI_DFAST_BITS = 9;
constexpr int GI_WAVES = 4;

// a fast-table entry: bits 0..3 code length (0: not a short code), 4..7 extra bits, 8: literal, 9: end of block, 16..31 the literal, the
// match length's base or the distance's base
struct GiTables {
    uint32_t fast[1 << GI_FAST_BITS];
    uint32_t dfast[1 << GI_DFAST_BITS];
    uint16_t lsym[288], dsym[32];     // symbols in canonical order
    uint16_t lcount[16], dcount[16];
    uint8_t len[320];
};

__device__ const uint16_t GI_LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t GI_LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t GI_DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t GI_DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t GI_CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// the bit reader: 64 bits in a register pair, refilled from the (read-only) input by 8-byte loads at byte granularity
struct GiBits {
    const uint8_t *p;      // next byte not yet in the buffer
    const uint8_t *end;
    uint64_t buf;
    int cnt;
    __device__ void init(const uint8_t *b, const uint8_t *e) { p = b; end = e; buf = 0; cnt = 0; }
    __device__ inline void refill() {
        // (reads up to 8 bytes past the block's end: the input buffer is padded)
        uint64_t w;
        memcpy(&w, p, 8);
        buf |= w << cnt;
        p += (63 - cnt) >> 3;
        cnt |= 56;
    }
    __device__ inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    __device__ inline void drop(int n) { buf >>= n; cnt -= n; }
    __device__ inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

// canonical order and counts of a code (one lane; n <= 288), then the fast table by all lanes.  kind 0: literal/length code, 1: distance
// code, 2: the code-length code (entry = the symbol in bits 16.., length in bits 0..3)
__device__ void gi_build(const uint8_t *len, int n, uint16_t *count, uint16_t *symbol, uint32_t *fast, int fast_bits, int kind, int lane) {
    if (lane == 0) {
        for (int l = 0; l < 16; l++) count[l] = 0;
        for (int s = 0; s < n; s++) count[len[s]]++;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int s = 0; s < n; s++) if (len[s]) symbol[offs[len[s]]++] = (uint16_t)s;
    }
    for (int i = lane; i < (1 << fast_bits); i += 64) fast[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int total = 0;
    for (int l = 1; l < 16; l++) total += count[l];
    for (int i = lane; i < total; i += 64) {
        int L = 1, first = 0, index = 0;
        while (i >= index + count[L]) { index += count[L]; first = (first + count[L]) << 1; L++; }
        if (L > fast_bits) continue;
        const uint32_t code = (uint32_t)(first + (i - index));
        const uint32_t rev = __brev(code) >> (32 - L);
        const uint32_t sym = symbol[i];
        uint32_t e = (uint32_t)L;
        if (kind == 2) e |= sym << 16;
        else if (kind == 1) e |= sym < 30 ? ((uint32_t)GI_DEXT[sym] << 4) | ((uint32_t)GI_DBASE[sym] << 16) : 0xFFFF0000u;
        else if (sym < 256) e |= 0x100u | (sym << 16);
        else if (sym == 256) e |= 0x200u;
        else e |= sym - 257 < 29 ? ((uint32_t)GI_LEXT[sym - 257] << 4) | ((uint32_t)GI_LBASE[sym - 257] << 16) : 0xFFFF0000u;
        for (uint32_t j = rev; j < (1u << fast_bits); j += 1u << L) fast[j] = e;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// one symbol bit by bit along the canonical order (puff.c's decode): codes longer than the fast table's index
__device__ inline int gi_decode_slow(GiBits &b, const uint16_t *count, const uint16_t *symbol) {
    int code = 0, first = 0, index = 0;
    for (int L = 1; L <= 15; L++) {
        code |= (int)b.take(1);
        const int c = __builtin_amdgcn_readfirstlane((int)count[L]);
        if (code - c < first) return __builtin_amdgcn_readfirstlane((int)symbol[index + (code - first)]);
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}
// the same entry the fast table would have held
__device__ inline uint32_t gi_entry_slow(GiBits &b, const uint16_t *count, const uint16_t *symbol, int kind) {
    const int sym = gi_decode_slow(b, count, symbol);
    if (sym < 0) return 0xFFFF0000u;
    if (kind == 2) return (uint32_t)sym << 16;
    if (kind == 1) return sym < 30 ? ((uint32_t)GI_DEXT[sym] << 4) | ((uint32_t)GI_DBASE[sym] << 16) : 0xFFFF0000u;
    if (sym < 256) return 0x100u | ((uint32_t)sym << 16);
    if (sym == 256) return 0x200u;
    return sym - 257 < 29 ? ((uint32_t)GI_LEXT[sym - 257] << 4) | ((uint32_t)GI_LBASE[sym - 257] << 16) : 0xFFFF0000u;
}
__device__ inline uint32_t gi_lookup(GiBits &b, const uint32_t *fast, int fast_bits, const uint16_t *count, const uint16_t *symbol, int kind) {
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)fast[b.peek(fast_bits)]);
    if (e & 15u) { b.drop((int)(e & 15u)); return e; }
    return gi_entry_slow(b, count, symbol, kind);
}

__global__ void __launch_bounds__(64 * GI_WAVES)
gi_inflate_kernel(const uint8_t *__restrict__ in, const tbk_ginflate_block *__restrict__ blks, const uint64_t *__restrict__ out_offs, uint32_t n_blks, uint8_t *__restrict__ out,
                  uint32_t *__restrict__ bad) {
    __shared__ GiTables tabs[GI_WAVES];
    // (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same in all 64 lanes, and with it everything read through the
    // wave's tables and block - the whole state of the decoder - would live in vector registers: measured, 20 % slower)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    GiTables &T = tabs[wave];
    // The grid is as many waves as the launch wants resident (tbk_ginflate_submit), not one per block: a wave takes the next block off a
    // counter (bad[2], zeroed with the other two) until there is none.  A grid of one wave per block - ten thousand for a window - sits in
    // the dispatcher for most of the kernel's 30 ms, and the kernels of every other stream on the device (the classifier's, the bins'
    // encoder's) wait behind it: measured, not one of them ran beside such a kernel (EXPERIMENTS.md, round 6).
  for (;;) {
    uint32_t next = 0;
    if (lane == 0) next = atomicAdd(bad + 2, 1u);
    const uint32_t bi = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
    if (bi >= n_blks) break;
    const tbk_ginflate_block blk = blks[bi];
    if (blk.out_len == 0) continue;   // (the end-of-file marker: a final empty block, nothing to write)
    GiBits b;
    b.init(in + blk.in_off, in + blk.in_off + blk.in_len);
    uint8_t *dst = out + out_offs[bi];
    uint32_t pos = 0;
    bool fail = false;
    // (every step below writes a byte or uses up input bits; the count of steps a damaged stream can take is bounded all the same)
    uint32_t fuel = blk.out_len + 8u * blk.in_len + 1024u;
    for (bool last = false; !last && !fail;) {
        if (fuel < 8u) { fail = true; break; }
        fuel -= 8u;   // (a deflate block is at least ten bits of input, and its end-of-block symbol one more step)
        b.refill();
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 0) {  // stored
            b.drop(b.cnt & 7);
            // un-read the whole bytes still in the buffer
            b.p -= b.cnt >> 3; b.buf = 0; b.cnt = 0;
            if (b.p + 4 > b.end) { fail = true; break; }
            const uint32_t n = b.p[0] | ((uint32_t)b.p[1] << 8);
            b.p += 4;
            if (b.p + n > b.end || pos + n > blk.out_len) { fail = true; break; }
            for (uint32_t i = lane; i < n; i += 64) dst[pos + i] = b.p[i];
            pos += n; b.p += n;
            continue;
        }
        if (type == 3) { fail = true; break; }
        int nlit = 288, ndist = 30;
        if (type == 1) {
            for (int i = lane; i < 288; i += 64) T.len[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            if (lane < 30) T.len[288 + lane] = 5;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else {
            nlit = (int)b.take(5) + 257; ndist = (int)b.take(5) + 1;
            const int ncl = (int)b.take(4) + 4;
            // the code-length code: lengths into T.len[0..19), its tables into the dist tables' space (built before those)
            if (lane < 19) T.len[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int i = 0; i < ncl; i++) { if (b.cnt < 3) b.refill(); const uint32_t v = b.take(3); if (lane == 0) T.len[GI_CLORD[i]] = (uint8_t)v; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            gi_build(T.len, 19, T.dcount, T.dsym, T.dfast, 7, 2, lane);
            uint8_t prev = 0;
            int i = 0;
            const int want = nlit + ndist;
            while (i < want && !fail) {
                b.refill();
                const uint32_t e = gi_lookup(b, T.dfast, 7, T.dcount, T.dsym, 2);
                const int sym = (int)(e >> 16);
                if (sym > 18) { fail = true; break; }
                if (sym < 16) { if (lane == 0) T.len[i] = (uint8_t)sym; prev = (uint8_t)sym; i++; continue; }
                int rep; uint8_t val = 0;
                if (sym == 16) { val = prev; rep = 3 + (int)b.take(2); }
                else if (sym == 17) rep = 3 + (int)b.take(3);
                else rep = 11 + (int)b.take(7);
                if (i + rep > want) { fail = true; break; }
                for (int r = lane; r < rep; r += 64) T.len[i + r] = val;
                i += rep; prev = val;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (fail) break;
        }
        gi_build(T.len, nlit, T.lcount, T.lsym, T.fast, GI_FAST_BITS, 0, lane);
        gi_build(T.len + nlit, ndist, T.dcount, T.dsym, T.dfast, GI_DFAST_BITS, 1, lane);
        // ---- the symbols ----
        for (;;) {
            if (b.cnt < 32) { if (b.p > b.end + 8) { fail = true; break; } b.refill(); }   // (a corrupt stream must not walk out of its input)
            if (fuel-- == 0u) { fail = true; break; }
            uint32_t e = gi_lookup(b, T.fast, GI_FAST_BITS, T.lcount, T.lsym, 0);
            if (e & 0x100u) {   // a literal: every lane stores the same byte to the same place (no lane mask to set up)
                if (pos >= blk.out_len) { fail = true; break; }
                dst[pos++] = (uint8_t)(e >> 16);
                continue;
            }
            if (e & 0x200u) break;   // end of block
            if ((e >> 16) == 0xFFFFu) { fail = true; break; }
            if (b.cnt < 32) b.refill();
            const uint32_t len = (e >> 16) + b.take((int)((e >> 4) & 15u));
            const uint32_t d = gi_lookup(b, T.dfast, GI_DFAST_BITS, T.dcount, T.dsym, 1);
            if ((d >> 16) == 0xFFFFu) { fail = true; break; }
            if (b.cnt < 16) b.refill();
            const uint32_t dist = (d >> 16) + b.take((int)((d >> 4) & 15u));
            if (dist > pos || pos + len > blk.out_len) { fail = true; break; }
            // the lanes copy side by side; a match that overlaps itself repeats its first `dist` bytes
            const uint8_t *src = dst + pos - dist;
            if (dist >= len) { for (uint32_t i = lane; i < len; i += 64) dst[pos + i] = src[i]; }
            else { for (uint32_t i = lane; i < len; i += 64) dst[pos + i] = src[i % dist]; }
            pos += len;
        }
    }
    if (fail || pos != blk.out_len) {   // the window is refused as a whole: this wave is done
        if (lane == 0) atomicAdd(bad, 1u);
        break;
    }
  }
}


// expected CRC-32s against the ones summed from the inflated text
__global__ void __launch_bounds__(GD_T)
gi_check_kernel(const tbk_ginflate_block *__restrict__ blks, const uint32_t *__restrict__ crc, uint32_t n_blks, uint32_t *__restrict__ bad) {
    const uint32_t i = blockIdx.x * GD_T + threadIdx.x;
    if (i < n_blks && blks[i].out_len && crc[i] != blks[i].crc) atomicAdd(bad + 1, 1u);
}

namespace {
struct GiSlot {
    PinBuf h_in, h_out, h_blocks, h_members, h_offs, h_bad;
    DevBuf d_in, d_out, d_blocks, d_members, d_offs, d_crc, d_bad;
    hipEvent_t done = nullptr;   // the window's kernels
    size_t out_bytes = 0, head = 0;
    bool busy = false;
    void drop() {
        h_in.drop(); h_out.drop(); h_blocks.drop(); h_members.drop(); h_offs.drop(); h_bad.drop();
        d_in.drop(); d_out.drop(); d_blocks.drop(); d_members.drop(); d_offs.drop(); d_crc.drop(); d_bad.drop();
        if (done) (void)hipEventDestroy(done);
    }
};
}  // namespace

struct tbk_ginflate {
    int device = 0;
    hipStream_t stream = nullptr, stream_in = nullptr, stream_out = nullptr;
    hipEvent_t in_done = nullptr;
    GdX2n x2n = gd_x2n_table();
    GdCrcTabs *d_crc_tabs = nullptr;
    GiSlot slots[TBK_GINFLATE_SLOTS];
    uint64_t blocks = 0, text_bytes = 0;
    int cus = 256, resident_wgs = 5;   // the inflate kernel's grid: workgroups (of GI_WAVES waves, 28 KB of LDS) per compute unit
};

int tbk_ginflate_create(int device, tbk_ginflate **out) {
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); tbk_set_error_(TBK_ERR_NO_DEVICE, "GPU inflater: no such device"); return TBK_ERR_NO_DEVICE; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "hipSetDevice", e);
    tbk_ginflate *g = new tbk_ginflate();
    g->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) g->cus = prop.multiProcessorCount;
    // (five such workgroups fill a compute unit's LDS: 5120 waves, one round for a window of 128 MB of the file)
    if (const char *v = getenv("TBK_BGZF_GPU_WGS")) g->resident_wgs = std::max(1, std::min(8, atoi(v)));
    e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream_out, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->in_done, hipEventDisableTiming);
    for (GiSlot &s : g->slots) if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    if (e == hipSuccess) {
        GdCrcTabs tabs;
        for (uint32_t j = 0; j < 128; j++) { tabs.lo[j] = gd_x2nmodp(g->x2n, j, 3 + 6); tabs.hi[j] = gd_x2nmodp(g->x2n, j, 3 + 6 + 7); }
        e = hipMalloc((void **)&g->d_crc_tabs, sizeof tabs);
        if (e == hipSuccess) e = hipMemcpy(g->d_crc_tabs, &tabs, sizeof tabs, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { tbk_ginflate_destroy(g); return gfail(TBK_ERR_HIP, "GPU inflater setup", e); }
    *out = g;
    return TBK_OK;
}

void tbk_ginflate_destroy(tbk_ginflate *g) {
    if (!g) return;
    if (hipSetDevice(g->device) == hipSuccess) {
        for (hipStream_t s : {g->stream, g->stream_in, g->stream_out}) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        for (GiSlot &s : g->slots) s.drop();
        if (g->in_done) (void)hipEventDestroy(g->in_done);
        if (g->d_crc_tabs) (void)hipFree(g->d_crc_tabs);
    }
    delete g;
}

// The slot's pinned buffer for `bytes` of deflated input (the caller copies the window's blocks into it, back to back or as they lie in
// the file), or NULL.
uint8_t *tbk_ginflate_input(tbk_ginflate *g, int slot, size_t bytes) {
    if (!g || slot < 0 || slot >= TBK_GINFLATE_SLOTS || hipSetDevice(g->device) != hipSuccess) return nullptr;
    GiSlot &s = g->slots[slot];
    if (s.h_in.need(bytes + 64) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (uint8_t *)s.h_in.p;
}

// Size a slot's buffers ahead of its first window (a gigabyte of pinned and device memory takes the runtime a tenth of a second: the
// reader's worker has another thread do this for the slots behind the first while the first window is on its way).  Not while the
// slot is in use.
int tbk_ginflate_reserve(tbk_ginflate *g, int slot, size_t in_bytes, size_t n_blocks, size_t out_bytes) {
    if (!g || slot < 0 || slot >= TBK_GINFLATE_SLOTS) { tbk_set_error_(TBK_ERR_INVALID, "GPU inflater: bad argument"); return TBK_ERR_INVALID; }
    hipError_t e = hipSetDevice(g->device);
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "hipSetDevice", e);
    GiSlot &s = g->slots[slot];
    e = s.h_in.need(in_bytes + 64, true);
    if (e == hipSuccess) e = s.h_blocks.need(n_blocks * sizeof(tbk_ginflate_block));
    if (e == hipSuccess) e = s.h_members.need((n_blocks + 1) * sizeof(GdMember));
    if (e == hipSuccess) e = s.h_offs.need((n_blocks + 1) * 8);
    if (e == hipSuccess) e = s.h_bad.need(64);
    if (e == hipSuccess) e = s.h_out.need(out_bytes + 64, true);
    if (e == hipSuccess) e = s.d_in.need(in_bytes + 64, true);
    if (e == hipSuccess) e = s.d_out.need(out_bytes + 64, true);
    if (e == hipSuccess) e = s.d_blocks.need(n_blocks * sizeof(tbk_ginflate_block));
    if (e == hipSuccess) e = s.d_members.need((n_blocks + 1) * sizeof(GdMember));
    if (e == hipSuccess) e = s.d_offs.need((n_blocks + 1) * 8);
    if (e == hipSuccess) e = s.d_crc.need((n_blocks + 1) * 4);
    if (e == hipSuccess) e = s.d_bad.need(64);
    if (e != hipSuccess) return gfail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "GPU inflater buffers", e);
    return TBK_OK;
}

// Queue a window: blocks[i] = where block i's raw deflate stream lies in the slot's input (in_off, in_len), how much text it makes and
// the CRC-32 its trailer names.  The text of the blocks lies back to back in the slot's output, `head` bytes into it (room in front for
// what the parser has left of the window before).  Asynchronous: tbk_ginflate_wait collects.
int tbk_ginflate_submit(tbk_ginflate *g, int slot, size_t in_bytes, const tbk_ginflate_block *blocks, size_t n_blocks, size_t head) {
    if (!g || slot < 0 || slot >= TBK_GINFLATE_SLOTS || !blocks || !n_blocks) { tbk_set_error_(TBK_ERR_INVALID, "GPU inflater: bad argument"); return TBK_ERR_INVALID; }
    hipError_t e = hipSetDevice(g->device);
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "hipSetDevice", e);
    GiSlot &s = g->slots[slot];
    if (s.busy) { tbk_set_error_(TBK_ERR_STATE, "GPU inflater: the slot is in flight"); return TBK_ERR_STATE; }
    if (n_blocks > 0x7FFFFFF0ull / 1) { tbk_set_error_(TBK_ERR_INVALID, "GPU inflater: too many blocks"); return TBK_ERR_INVALID; }
    uint64_t out_total = 0;
    e = s.h_blocks.need(n_blocks * sizeof(tbk_ginflate_block));
    if (e == hipSuccess) e = s.h_members.need((n_blocks + 1) * sizeof(GdMember));
    if (e == hipSuccess) e = s.h_offs.need((n_blocks + 1) * 8);
    if (e == hipSuccess) e = s.h_bad.need(64);
    if (e != hipSuccess) return gfail(TBK_ERR_NOMEM, "GPU inflater buffers", e);
    memcpy(s.h_blocks.p, blocks, n_blocks * sizeof(tbk_ginflate_block));
    GdMember *hm = (GdMember *)s.h_members.p;
    uint64_t *offs = (uint64_t *)s.h_offs.p;
    uint32_t longest = 0;
    for (size_t i = 0; i < n_blocks; i++) {
        if (blocks[i].out_len > 65536u || (uint64_t)blocks[i].in_off + blocks[i].in_len > in_bytes) { tbk_set_error_(TBK_ERR_INVALID, "GPU inflater: a block outside its window"); return TBK_ERR_INVALID; }
        offs[i] = out_total;
        hm[i] = GdMember{out_total, 0, blocks[i].out_len, 0};
        longest = std::max(longest, blocks[i].out_len);
        out_total += blocks[i].out_len;
    }
    s.out_bytes = out_total;
    e = s.d_in.need(in_bytes + 64);
    if (e == hipSuccess) e = s.d_out.need(out_total + 64);
    if (e == hipSuccess) e = s.d_blocks.need(n_blocks * sizeof(tbk_ginflate_block));
    if (e == hipSuccess) e = s.d_members.need((n_blocks + 1) * sizeof(GdMember));
    if (e == hipSuccess) e = s.d_offs.need((n_blocks + 1) * 8);
    if (e == hipSuccess) e = s.d_crc.need((n_blocks + 1) * 4);
    if (e == hipSuccess) e = s.d_bad.need(64);
    if (e == hipSuccess) e = s.h_out.need(head + out_total + 64);
    if (e != hipSuccess) return gfail(e == hipErrorOutOfMemory ? TBK_ERR_NOMEM : TBK_ERR_HIP, "GPU inflater buffers", e);
    e = hipMemcpyAsync(s.d_in.p, s.h_in.p, in_bytes, hipMemcpyHostToDevice, g->stream_in);
    if (e == hipSuccess) e = hipMemsetAsync((uint8_t *)s.d_in.p + in_bytes, 0, 64, g->stream_in);
    if (e == hipSuccess) e = hipMemcpyAsync(s.d_blocks.p, s.h_blocks.p, n_blocks * sizeof(tbk_ginflate_block), hipMemcpyHostToDevice, g->stream_in);
    if (e == hipSuccess) e = hipMemcpyAsync(s.d_members.p, s.h_members.p, n_blocks * sizeof(GdMember), hipMemcpyHostToDevice, g->stream_in);
    if (e == hipSuccess) e = hipMemcpyAsync(s.d_offs.p, s.h_offs.p, n_blocks * 8, hipMemcpyHostToDevice, g->stream_in);
    if (e == hipSuccess) e = hipEventRecord(g->in_done, g->stream_in);
    if (e == hipSuccess) e = hipStreamWaitEvent(g->stream, g->in_done, 0);
    if (e == hipSuccess) e = hipMemsetAsync(s.d_bad.p, 0, 64, g->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s.d_crc.p, 0, (n_blocks + 1) * 4, g->stream);
    if (e == hipSuccess) {
        // g->resident_wgs workgroups per compute unit, each wave of them working through blocks (see the kernel)
        const size_t wgs = std::min<size_t>((n_blocks + GI_WAVES - 1) / GI_WAVES, (size_t)g->cus * (size_t)g->resident_wgs);
        hipLaunchKernelGGL(gi_inflate_kernel, dim3((unsigned)wgs), dim3(64 * GI_WAVES), 0, g->stream, (const uint8_t *)s.d_in.p, (const tbk_ginflate_block *)s.d_blocks.p,
                           (const uint64_t *)s.d_offs.p, (uint32_t)n_blocks, (uint8_t *)s.d_out.p, (uint32_t *)s.d_bad.p);
        // the blocks' CRC-32s: gd_crc_kernel over (block = member); blockIdx.y is limited to 65535: in turns
        for (size_t first = 0; first < n_blocks; first += 65535) {
            const size_t nb = std::min<size_t>(65535, n_blocks - first);
            hipLaunchKernelGGL(gd_crc_kernel, dim3((unsigned)std::max<uint64_t>(1, (((uint64_t)longest + 63) / 64 + GD_T - 1) / GD_T), (unsigned)nb), dim3(GD_T), 0, g->stream, (const uint8_t *)s.d_out.p,
                               (const GdMember *)s.d_members.p + first, (const GdCrcTabs *)g->d_crc_tabs, g->x2n, (uint32_t *)s.d_crc.p + first);
        }
        hipLaunchKernelGGL(gi_check_kernel, dim3((unsigned)((n_blocks + GD_T - 1) / GD_T)), dim3(GD_T), 0, g->stream, (const tbk_ginflate_block *)s.d_blocks.p, (const uint32_t *)s.d_crc.p,
                           (uint32_t)n_blocks, (uint32_t *)s.d_bad.p);
        e = hipGetLastError();
    }
    // The text's copy home is NOT queued here behind the kernels' event: a copy that waits sits at the head of the copy engine's queue,
    // and every copy queued after it - the classifier's batches, the bins' encoder's text - waits with it for the 16 ms of this window's
    // kernel and the copy itself (measured: the classifier's kernels started 11 ms after an inflate kernel's end, to the tenth of a
    // millisecond, never earlier).  tbk_ginflate_wait queues it when the kernels are done.
    if (e == hipSuccess) e = hipEventRecord(s.done, g->stream);
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "GPU inflater submit", e);
    s.head = head;
    s.busy = true;
    g->blocks += n_blocks; g->text_bytes += out_total;
    return TBK_OK;
}

// The window's text (`head` bytes into the slot's pinned output; valid until the slot's next tbk_ginflate_input); *bad = blocks that did
// not decode to their length + blocks whose CRC-32 differs from their trailer's.
int tbk_ginflate_wait(tbk_ginflate *g, int slot, uint8_t **out_base, size_t *text_bytes, uint32_t *bad) {
    if (!g || slot < 0 || slot >= TBK_GINFLATE_SLOTS) { tbk_set_error_(TBK_ERR_INVALID, "GPU inflater: bad argument"); return TBK_ERR_INVALID; }
    GiSlot &s = g->slots[slot];
    if (!s.busy) { tbk_set_error_(TBK_ERR_STATE, "GPU inflater: nothing in flight in the slot"); return TBK_ERR_STATE; }
    hipError_t e = hipSetDevice(g->device);
    if (e == hipSuccess) e = hipEventSynchronize(s.done);
    if (e == hipSuccess && s.out_bytes) e = hipMemcpyAsync((uint8_t *)s.h_out.p + s.head, s.d_out.p, s.out_bytes, hipMemcpyDeviceToHost, g->stream_out);
    if (e == hipSuccess) e = hipMemcpyAsync(s.h_bad.p, s.d_bad.p, 8, hipMemcpyDeviceToHost, g->stream_out);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream_out);
    s.busy = false;
    if (e != hipSuccess) return gfail(TBK_ERR_HIP, "GPU inflater wait", e);
    const uint32_t *b = (const uint32_t *)s.h_bad.p;
    if (out_base) *out_base = (uint8_t *)s.h_out.p;
    if (text_bytes) *text_bytes = s.out_bytes;
    if (bad) *bad = b[0] + b[1];
    return TBK_OK;
}

uint32_t tbk_crc32(uint32_t crc, const uint8_t *p, size_t n);  // tbk_crc.cpp

// C-ABI (include/tbk.h): n_members pieces of text -> as many gzip members, coded on `device`, one job, synchronously.  text holds the
// members back to back (member_len[i] bytes each); the members are written back to back into dst and member_out_len[i] says how long
// each one is.  *need = bytes of dst used (or needed, when cap is too small: TBK_ERR_NOMEM).  What tests and tools call; the bin writer
// drives the same encoder three jobs deep.
extern "C" int tbk_gzip_members_device(int device, const char *text, const uint64_t *member_len, uint64_t n_members, char *dst, uint64_t cap, uint64_t *member_out_len,
                                       uint64_t *need) {
    if ((n_members && (!member_len || !member_out_len)) || !need) { tbk_set_error_(TBK_ERR_INVALID, "tbk_gzip_members_device: NULL argument"); return TBK_ERR_INVALID; }
    tbk_gdeflate *g = nullptr;
    int rc = tbk_gdeflate_create(device, &g);
    if (rc) return rc;
    std::vector<tbk_gdeflate_member> members;
    uint64_t off = 0;
    for (uint64_t i = 0; i < n_members; i++) { members.push_back(tbk_gdeflate_member{text + off, (size_t)member_len[i], 0}); off += member_len[i]; }
    rc = tbk_gdeflate_submit(g, members.data(), members.size());
    if (!rc) {
        rc = tbk_gdeflate_text_done(g, 0);  // (the members' CRC-32s are the device's)
    }
    std::vector<tbk_gdeflate_out> outs;
    uint64_t used = 0, at = 0;
    while (!rc && tbk_gdeflate_in_flight(g) > 0) {
        rc = tbk_gdeflate_collect(g, true, outs);
        for (const tbk_gdeflate_out &o : outs) {
            if (dst && used + o.n <= cap) memcpy(dst + used, o.data, o.n);
            if (at < n_members) member_out_len[at++] = o.n;
            used += o.n;
        }
    }
    tbk_gdeflate_destroy(g);
    if (rc) return rc;
    *need = used;
    if (used > cap || (!dst && used)) { tbk_set_error_(TBK_ERR_NOMEM, "tbk_gzip_members_device: dst too small"); return TBK_ERR_NOMEM; }
    return TBK_OK;
}

#ifdef GD_PHASES
extern "C" int tbk_gdeflate_phases_(unsigned long long out[16]) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(gd_phase), 16 * sizeof(unsigned long long));
    unsigned long long z[16] = {0};
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(gd_phase), z, sizeof z);
    return e == hipSuccess ? 0 : -5;
}
#endif

// C-ABI (include/tbk.h): the encoder timed by itself.  `reps` jobs of the same members (text in PINNED host memory, as the bin writer
// holds it) go through the three-deep ring - text in, kernels, members out, overlapped - and *pipelined_s is the wall time per job in
// the steady state (link included); *kernels_s the kernels' own time per job (HIP events around one job's kernels on an idle
// device).  *out_bytes: bytes of members one job makes.
extern "C" int tbk_gzip_bench_device(int device, const char *text, const uint64_t *member_len, uint64_t n_members, int reps, double *pipelined_s, double *kernels_s,
                                     uint64_t *out_bytes) {
    if (!text || !member_len || !n_members || reps < 4 || !pipelined_s || !kernels_s || !out_bytes) { tbk_set_error_(TBK_ERR_INVALID, "tbk_gzip_bench_device: bad argument"); return TBK_ERR_INVALID; }
    tbk_gdeflate *g = nullptr;
    int rc = tbk_gdeflate_create(device, &g);
    if (rc) return rc;
    std::vector<tbk_gdeflate_member> members;
    uint64_t off = 0;
    for (uint64_t i = 0; i < n_members; i++) { members.push_back(tbk_gdeflate_member{text + off, (size_t)member_len[i], 0}); off += member_len[i]; }
    std::vector<tbk_gdeflate_out> outs;
    // warm-up: three jobs (every slot's buffers exist), drained
    for (int i = 0; i < 3 && !rc; i++) { rc = tbk_gdeflate_submit(g, members.data(), members.size()); if (!rc) rc = tbk_gdeflate_collect(g, false, outs); }
    while (!rc && tbk_gdeflate_in_flight(g) > 0) rc = tbk_gdeflate_collect(g, true, outs);
    uint64_t bytes = 0;
    for (const tbk_gdeflate_out &o : outs) bytes += o.n;
    // one job alone, its kernels between two events
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    if (!rc) {
        hipError_t e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        // (the kernels wait for the text's event: bracket them by recording behind the text's arrival and behind the last kernel)
        if (e == hipSuccess) { rc = tbk_gdeflate_submit(g, members.data(), members.size()); }
        if (e != hipSuccess) rc = gfail(TBK_ERR_HIP, "bench", e);
        while (!rc && tbk_gdeflate_in_flight(g) > 0) rc = tbk_gdeflate_collect(g, true, outs);
        // timed again with the text already resident: the same job's kernels re-launched are what the events bracket
        if (!rc) {
            Job &j = g->jobs[(g->submitted - 1) % 3];
            const size_t nm = members.size();
            uint64_t nb = 0, longest = 0, text_total = 0, slot_total = 0;
            for (size_t i = 0; i < nm; i++) {
                const uint64_t entries = members[i].n / 8192 + 1;
                nb += entries; longest = std::max<uint64_t>(longest, members[i].n); text_total += members[i].n;
                slot_total += (((uint64_t)members[i].n + members[i].n / 8 + 15) & ~(uint64_t)15) + entries * 1040;
            }
            const uint64_t n_words = (text_total + 31) / 32;
            (void)hipEventRecord(e0, g->stream);
            (void)hipMemsetAsync(j.d_member_end.p, 0, (nm + 1) * 12, g->stream);
            (void)hipMemsetAsync(j.d_slots.p, 0, slot_total, g->stream);
            hipLaunchKernelGGL(gd_newline_kernel, dim3((unsigned)((n_words + GD_T - 1) / GD_T)), dim3(GD_T), 0, g->stream, (const uint8_t *)j.d_text.p, n_words, (uint32_t *)j.d_bitmap.p);
            hipLaunchKernelGGL(gd_cut_kernel, dim3((unsigned)nm), dim3(64), 0, g->stream, (const GdMember *)j.d_members.p, (uint32_t)nm, (const uint32_t *)j.d_bitmap.p, (GdBlock *)j.d_blocks.p);
            hipLaunchKernelGGL(gd_crc_kernel, dim3((unsigned)(((longest + 63) / 64 + GD_T - 1) / GD_T), (unsigned)nm), dim3(GD_T), 0, g->stream, (const uint8_t *)j.d_text.p,
                               (const GdMember *)j.d_members.p, (const GdCrcTabs *)g->d_crc_tabs, g->x2n, (uint32_t *)((uint64_t *)j.d_member_end.p + nm + 1));
            hipLaunchKernelGGL(gd_encode_kernel, dim3((unsigned)nb), dim3(GD_T), 0, g->stream, (const uint8_t *)j.d_text.p, (const GdBlock *)j.d_blocks.p, (uint8_t *)j.d_slots.p, (uint32_t *)j.d_sizes.p);
            hipLaunchKernelGGL(gd_scan_kernel, dim3(1), dim3(GD_T), 0, g->stream, (const GdBlock *)j.d_blocks.p, (const uint32_t *)j.d_sizes.p, (uint32_t)nb, (uint64_t *)j.d_offsets.p,
                               (uint64_t *)j.d_member_end.p, (uint32_t)nm);
            hipLaunchKernelGGL(gd_gather_kernel, dim3((unsigned)nb), dim3(GD_T), 0, g->stream, (const GdBlock *)j.d_blocks.p, (const uint32_t *)j.d_sizes.p, (const uint64_t *)j.d_offsets.p,
                               (const uint8_t *)j.d_slots.p, (uint8_t *)j.d_dense.p);
            (void)hipEventRecord(e1, g->stream);
            hipError_t e2 = hipEventSynchronize(e1);
            if (e2 == hipSuccess) e2 = hipEventElapsedTime(&ms, e0, e1);
            if (e2 != hipSuccess) rc = gfail(TBK_ERR_HIP, "bench events", e2);
        }
    }
    // the ring in its steady state
    double wall = 0;
    if (!rc) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps && !rc; i++) {
            rc = tbk_gdeflate_submit(g, members.data(), members.size());
            if (!rc) rc = tbk_gdeflate_collect(g, false, outs);
        }
        while (!rc && tbk_gdeflate_in_flight(g) > 0) rc = tbk_gdeflate_collect(g, true, outs);
        wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    tbk_gdeflate_destroy(g);
    if (rc) return rc;
    *pipelined_s = wall / reps;
    *kernels_s = ms * 1e-3;
    *out_bytes = bytes;
    return TBK_OK;
}

// the blocks of a bgzf file in memory (tbk_bgzf_inflate_device, tbk_bgzf_bench_device)
// C-ABI (include/tbk.h): a whole bgzf file (or a run of its blocks) in host memory -> its text, inflated on `device`, one window,
// synchronously.  What tests and tools call; the reader drives the same inflater four windows deep.  *text_len = bytes of text (also
// when dst is too small: TBK_ERR_NOMEM); bytes that are not a bgzf block (an ordinary gzip member, garbage) end the run with
// TBK_ERR_FORMAT; a block that does not decode or fails its CRC-32: TBK_ERR_FORMAT too.
static int bgzf_blocks_of(const uint8_t *data, uint64_t size, std::vector<tbk_ginflate_block> &blocks, uint64_t *out_total_p) {
    uint64_t p = 0, out_total = 0;
    while (p < size) {
        if (data[p] == 0 && !blocks.empty()) { p++; continue; }   // zero padding between members (Python's gzip skips it too)
        if (size - p < 18 || data[p] != 0x1f || data[p + 1] != 0x8b || data[p + 2] != 8 || !(data[p + 3] & 4)) { tbk_set_error_(TBK_ERR_FORMAT, "not a BGZF block"); return TBK_ERR_FORMAT; }
        const uint64_t xlen = data[p + 10] | ((uint64_t)data[p + 11] << 8);
        if (xlen < 6 || data[p + 12] != 'B' || data[p + 13] != 'C' || data[p + 14] != 2 || data[p + 15] != 0) { tbk_set_error_(TBK_ERR_FORMAT, "not a BGZF block"); return TBK_ERR_FORMAT; }
        const uint64_t bs = ((uint64_t)data[p + 16] | ((uint64_t)data[p + 17] << 8)) + 1, hdr = 12 + xlen;
        if (bs < 26 || hdr + 8 > bs || p + bs > size) { tbk_set_error_(TBK_ERR_FORMAT, "corrupt BGZF block"); return TBK_ERR_FORMAT; }
        const uint8_t *b = data + p;
        const uint32_t crc = (uint32_t)b[bs - 8] | ((uint32_t)b[bs - 7] << 8) | ((uint32_t)b[bs - 6] << 16) | ((uint32_t)b[bs - 5] << 24);
        const uint32_t isize = (uint32_t)b[bs - 4] | ((uint32_t)b[bs - 3] << 8) | ((uint32_t)b[bs - 2] << 16) | ((uint32_t)b[bs - 1] << 24);
        if (isize > (1u << 16)) { tbk_set_error_(TBK_ERR_FORMAT, "corrupt BGZF block"); return TBK_ERR_FORMAT; }
        blocks.push_back(tbk_ginflate_block{p + hdr, (uint32_t)(bs - hdr - 8), isize, crc, 0});
        out_total += isize;
        p += bs;
    }
    *out_total_p = out_total;
    return TBK_OK;
}

extern "C" int tbk_bgzf_inflate_device(int device, const uint8_t *data, uint64_t size, uint8_t *dst, uint64_t cap, uint64_t *text_len) {
    if ((!data && size) || !text_len) { tbk_set_error_(TBK_ERR_INVALID, "tbk_bgzf_inflate_device: NULL argument"); return TBK_ERR_INVALID; }
    *text_len = 0;
    std::vector<tbk_ginflate_block> blocks;
    uint64_t out_total = 0;
    { const int rc0 = bgzf_blocks_of(data, size, blocks, &out_total); if (rc0) return rc0; }
    *text_len = out_total;
    if (blocks.empty()) return TBK_OK;
    tbk_ginflate *g = nullptr;
    int rc = tbk_ginflate_create(device, &g);
    if (rc) return rc;
    uint8_t *in = tbk_ginflate_input(g, 0, (size_t)size);
    if (!in) { tbk_ginflate_destroy(g); tbk_set_error_(TBK_ERR_NOMEM, "GPU inflater: no pinned memory for the input"); return TBK_ERR_NOMEM; }
    memcpy(in, data, (size_t)size);
    rc = tbk_ginflate_submit(g, 0, (size_t)size, blocks.data(), blocks.size(), 0);
    uint8_t *text = nullptr;
    size_t n = 0;
    uint32_t bad = 0;
    if (!rc) rc = tbk_ginflate_wait(g, 0, &text, &n, &bad);
    if (!rc && bad) { tbk_set_error_(TBK_ERR_FORMAT, "inflate: corrupt BGZF block"); rc = TBK_ERR_FORMAT; }
    if (!rc && (n > cap || !dst)) { if (n) { tbk_set_error_(TBK_ERR_NOMEM, "tbk_bgzf_inflate_device: dst too small"); rc = TBK_ERR_NOMEM; } }
    if (!rc && n) memcpy(dst, text, n);
    tbk_ginflate_destroy(g);
    return rc;
}

// C-ABI (include/tbk.h; bench.py's `input_bgzf_inflater` record): the inflater by itself on one window of bgzf blocks in host memory.
// *kernels_s: the window's kernels (inflate, CRC-32, check) between two HIP events, the input resident, mean of `reps` launches;
// *ring_s: wall time per window of `reps` windows through the ring as the reader drives it (staging copy into pinned memory, copy in,
// kernels, text home, two windows in flight).  Every window's text is checked against its blocks' CRC-32s on the device.
extern "C" int tbk_bgzf_bench_device(int device, const uint8_t *data, uint64_t size, int reps, double *ring_s, double *kernels_s, uint64_t *text_bytes) {
    if (!data || !size || reps < 2 || !ring_s || !kernels_s || !text_bytes) { tbk_set_error_(TBK_ERR_INVALID, "tbk_bgzf_bench_device: bad argument"); return TBK_ERR_INVALID; }
    std::vector<tbk_ginflate_block> blocks;
    uint64_t out_total = 0;
    int rc = bgzf_blocks_of(data, size, blocks, &out_total);
    if (rc) return rc;
    if (blocks.empty() || !out_total) { tbk_set_error_(TBK_ERR_INVALID, "tbk_bgzf_bench_device: no blocks"); return TBK_ERR_INVALID; }
    *text_bytes = out_total;
    tbk_ginflate *g = nullptr;
    rc = tbk_ginflate_create(device, &g);
    if (rc) return rc;
    uint8_t *text = nullptr;
    size_t n = 0;
    uint32_t bad = 0;
    auto window = [&](int slot) -> int {
        uint8_t *in = tbk_ginflate_input(g, slot, (size_t)size);
        if (!in) { tbk_set_error_(TBK_ERR_NOMEM, "GPU inflater: no pinned memory for the input"); return TBK_ERR_NOMEM; }
        memcpy(in, data, (size_t)size);
        return tbk_ginflate_submit(g, slot, (size_t)size, blocks.data(), blocks.size(), 0);
    };
    auto home = [&](int slot) -> int {
        int r = tbk_ginflate_wait(g, slot, &text, &n, &bad);
        if (!r && (bad || n != out_total)) { tbk_set_error_(TBK_ERR_FORMAT, "inflate: corrupt BGZF block"); r = TBK_ERR_FORMAT; }
        return r;
    };
    // both slots' buffers exist
    for (int slot = 0; slot < 2 && !rc; slot++) { rc = window(slot); if (!rc) rc = home(slot); }
    // the kernels alone: slot 0's input is on the device
    if (!rc) {
        GiSlot &s = g->slots[0];
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipError_t e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        uint32_t longest = 0;
        for (const tbk_ginflate_block &b : blocks) longest = std::max(longest, b.out_len);
        const size_t nb = blocks.size();
        const size_t wgs = std::min<size_t>((nb + GI_WAVES - 1) / GI_WAVES, (size_t)g->cus * (size_t)g->resident_wgs);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e == hipSuccess) e = hipEventRecord(e0, g->stream);
        for (int i = 0; i < reps && e == hipSuccess; i++) {
            e = hipMemsetAsync(s.d_bad.p, 0, 64, g->stream);
            if (e == hipSuccess) e = hipMemsetAsync(s.d_crc.p, 0, (nb + 1) * 4, g->stream);
            hipLaunchKernelGGL(gi_inflate_kernel, dim3((unsigned)wgs), dim3(64 * GI_WAVES), 0, g->stream, (const uint8_t *)s.d_in.p, (const tbk_ginflate_block *)s.d_blocks.p,
                               (const uint64_t *)s.d_offs.p, (uint32_t)nb, (uint8_t *)s.d_out.p, (uint32_t *)s.d_bad.p);
            for (size_t first = 0; first < nb; first += 65535) {
                const size_t part = std::min<size_t>(65535, nb - first);
                hipLaunchKernelGGL(gd_crc_kernel, dim3((unsigned)std::max<uint64_t>(1, (((uint64_t)longest + 63) / 64 + GD_T - 1) / GD_T), (unsigned)part), dim3(GD_T), 0, g->stream,
                                   (const uint8_t *)s.d_out.p, (const GdMember *)s.d_members.p + first, (const GdCrcTabs *)g->d_crc_tabs, g->x2n, (uint32_t *)s.d_crc.p + first);
            }
            hipLaunchKernelGGL(gi_check_kernel, dim3((unsigned)((nb + GD_T - 1) / GD_T)), dim3(GD_T), 0, g->stream, (const tbk_ginflate_block *)s.d_blocks.p, (const uint32_t *)s.d_crc.p,
                               (uint32_t)nb, (uint32_t *)s.d_bad.p);
            if (e == hipSuccess) e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(e1, g->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        uint32_t h_bad[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpy(h_bad, s.d_bad.p, 8, hipMemcpyDeviceToHost);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (e != hipSuccess) rc = gfail(TBK_ERR_HIP, "bench events", e);
        else if (h_bad[0] + h_bad[1]) { tbk_set_error_(TBK_ERR_FORMAT, "inflate: corrupt BGZF block"); rc = TBK_ERR_FORMAT; }
        *kernels_s = (double)ms * 1e-3 / reps;
    }
    // the ring: window i + 1 staged and queued while window i is on the device
    if (!rc) {
        const auto t0 = std::chrono::steady_clock::now();
        rc = window(0);
        for (int i = 1; i < reps && !rc; i++) {
            rc = window(i & 1);
            if (!rc) rc = home((i - 1) & 1);
        }
        if (!rc) rc = home((reps - 1) & 1);
        *ring_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    }
    tbk_ginflate_destroy(g);
    return rc;
}
