// ginflate_gate.hip — the gate of a GPU inflater for bgzf input ("optionally block-parallel bgzf inflate on the
// device"): can one WAVE per bgzf block, thousands of them side by side, inflate FASTQ text faster than the host's 16 threads (6.5 GB/s)?
//
//   hipcc -O3 --offload-arch=gfx950 -o ginflate_gate tools/ginflate_gate.hip -lz && ./ginflate_gate [MB of text] [zlib level]
//
// The decoder is written UNIFORMLY: every lane of the wave runs the same control flow on the same values (the compiler keeps them in
// scalar registers), the input arrives through scalar loads, the Huffman tables live in LDS (built per deflate block by the wave),
// literals leave by one lane's byte store, matches are copied by the lanes side by side (dst[i] = src[i mod dist]).  The program makes
// its own input - HiFi-like FASTQ text cut into 60 000-byte blocks, each deflated raw by zlib at the given level, what bgzip writes -
// checks every byte of the output and prints the kernel's rate.  No library code: a measurement, kept with its result (EXPERIMENTS.md).
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Blk { uint64_t in_off, out_off; uint32_t in_len, out_len; };

constexpr int FAST_BITS = 10, DFAST_BITS = 9;
constexpr int WAVES_PER_BLOCK = 4;

// a fast-table entry: bits 0..3 code length (0: not a short code), 4..7 extra bits, 8: literal, 9: end of block, 16..31 the literal, the
// match length's base or the distance's base
struct Tables {
    uint32_t fast[1 << FAST_BITS];
    uint32_t dfast[1 << DFAST_BITS];
    uint16_t lsym[288], dsym[32];     // symbols in canonical order
    uint16_t lcount[16], dcount[16];
    uint8_t len[320];
};

__device__ const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// the bit reader: 64 bits in a register pair, refilled from the (read-only) input by 8-byte loads at byte granularity
struct Bits {
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
__device__ void build(const uint8_t *len, int n, uint16_t *count, uint16_t *symbol, uint32_t *fast, int fast_bits, int kind, int lane) {
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
        else if (kind == 1) e |= sym < 30 ? ((uint32_t)DEXT[sym] << 4) | ((uint32_t)DBASE[sym] << 16) : 0xFFFF0000u;
        else if (sym < 256) e |= 0x100u | (sym << 16);
        else if (sym == 256) e |= 0x200u;
        else e |= sym - 257 < 29 ? ((uint32_t)LEXT[sym - 257] << 4) | ((uint32_t)LBASE[sym - 257] << 16) : 0xFFFF0000u;
        for (uint32_t j = rev; j < (1u << fast_bits); j += 1u << L) fast[j] = e;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// one symbol bit by bit along the canonical order (puff.c's decode): codes longer than the fast table's index
template <class B> __device__ inline int decode_slow(B &b, const uint16_t *count, const uint16_t *symbol) {
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
template <class B> __device__ inline uint32_t entry_slow(B &b, const uint16_t *count, const uint16_t *symbol, int kind) {
    const int sym = decode_slow(b, count, symbol);
    if (sym < 0) return 0xFFFF0000u;
    if (kind == 2) return (uint32_t)sym << 16;
    if (kind == 1) return sym < 30 ? ((uint32_t)DEXT[sym] << 4) | ((uint32_t)DBASE[sym] << 16) : 0xFFFF0000u;
    if (sym < 256) return 0x100u | ((uint32_t)sym << 16);
    if (sym == 256) return 0x200u;
    return sym - 257 < 29 ? ((uint32_t)LEXT[sym - 257] << 4) | ((uint32_t)LBASE[sym - 257] << 16) : 0xFFFF0000u;
}
template <class B> __device__ inline uint32_t lookup(B &b, const uint32_t *fast, int fast_bits, const uint16_t *count, const uint16_t *symbol, int kind) {
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)fast[b.peek(fast_bits)]);
    if (e & 15u) { b.drop((int)(e & 15u)); return e; }
    return entry_slow(b, count, symbol, kind);
}

// MODE 0: a match is copied on the spot (load, wait, store).  MODE 1: its store is put off until the next match (or the end of the block):
// the bytes are on their way while the next symbols are decoded.  stats: literals, matches, bytes copied by matches, matches longer than 64.
template <int MODE>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK)
ginflate_kernel(const uint8_t *__restrict__ in, const Blk *__restrict__ blks, uint32_t n_blks, uint8_t *__restrict__ out, int *__restrict__ bad, unsigned long long *__restrict__ stats) {
    __shared__ Tables tabs[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (the compiler must know it is the same in every lane)
    const uint32_t bi = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (bi >= n_blks) return;
    Tables &T = tabs[wave];
    const Blk blk = blks[bi];
    Bits b;
    b.init(in + blk.in_off, in + blk.in_off + blk.in_len);
    uint8_t *dst = out + blk.out_off;
    uint32_t pos = 0;
    bool fail = false;
    uint32_t n_lit = 0, n_match = 0, n_copied = 0, n_long = 0;
    bool pend = false;
    uint32_t pend_pos = 0, pend_len = 0;
    uint8_t pend_val = 0;
    for (bool last = false; !last && !fail;) {
        b.refill();
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 0) {  // stored
            b.drop(b.cnt & 7);
            // un-read the whole bytes still in the buffer
            b.p -= b.cnt >> 3; b.buf = 0; b.cnt = 0;
            const uint32_t n = b.p[0] | ((uint32_t)b.p[1] << 8);
            b.p += 4;
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
            for (int i = 0; i < ncl; i++) { if (b.cnt < 3) b.refill(); const uint32_t v = b.take(3); if (lane == 0) T.len[CLORD[i]] = (uint8_t)v; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            build(T.len, 19, T.dcount, T.dsym, T.dfast, 7, 2, lane);
            uint8_t prev = 0;
            int i = 0;
            const int want = nlit + ndist;
            while (i < want && !fail) {
                b.refill();
                const uint32_t e = lookup(b, T.dfast, 7, T.dcount, T.dsym, 2);
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
        build(T.len, nlit, T.lcount, T.lsym, T.fast, FAST_BITS, 0, lane);
        build(T.len + nlit, ndist, T.dcount, T.dsym, T.dfast, DFAST_BITS, 1, lane);
        // ---- the symbols ----
        for (;;) {
            if (b.cnt < 32) b.refill();
            uint32_t e = lookup(b, T.fast, FAST_BITS, T.lcount, T.lsym, 0);
            if (e & 0x100u) {   // a literal: every lane stores the same byte to the same place (no lane mask to set up)
                if (pos >= blk.out_len) { fail = true; break; }
                dst[pos++] = (uint8_t)(e >> 16);
                n_lit++;
                continue;
            }
            if (e & 0x200u) break;   // end of block
            if ((e >> 16) == 0xFFFFu) { fail = true; break; }
            if (b.cnt < 32) b.refill();
            const uint32_t len = (e >> 16) + b.take((int)((e >> 4) & 15u));
            const uint32_t d = lookup(b, T.dfast, DFAST_BITS, T.dcount, T.dsym, 1);
            if ((d >> 16) == 0xFFFFu) { fail = true; break; }
            if (b.cnt < 16) b.refill();
            const uint32_t dist = (d >> 16) + b.take((int)((d >> 4) & 15u));
            if (dist > pos || pos + len > blk.out_len) { fail = true; break; }
            // the lanes copy side by side; a match that overlaps itself repeats its first `dist` bytes
            const uint8_t *src = dst + pos - dist;
            n_match++; n_copied += len; n_long += len > 64;
            if (MODE == 1) {
                if (pend) { if ((uint32_t)lane < pend_len) dst[pend_pos + lane] = pend_val; pend = false; }
                if (len <= 64) {
                    uint8_t v = 0;
                    if ((uint32_t)lane < len) v = src[dist >= len ? (uint32_t)lane : (uint32_t)lane % dist];
                    pend = true; pend_pos = pos; pend_len = len; pend_val = v;
                    pos += len;
                    continue;
                }
            }
            if (dist >= len) { for (uint32_t i = lane; i < len; i += 64) dst[pos + i] = src[i]; }
            else { for (uint32_t i = lane; i < len; i += 64) dst[pos + i] = src[i % dist]; }
            pos += len;
        }
        if (MODE == 1 && pend) { if ((uint32_t)lane < pend_len) dst[pend_pos + lane] = pend_val; pend = false; }
    }
    if (lane == 0 && stats) { atomicAdd(stats, (unsigned long long)n_lit); atomicAdd(stats + 1, (unsigned long long)n_match); atomicAdd(stats + 2, (unsigned long long)n_copied); atomicAdd(stats + 3, (unsigned long long)n_long); }
    if ((fail || pos != blk.out_len) && lane == 0) atomicAdd(bad, 1);
}


// ---- the second decoder: nothing on the path from one symbol to the next waits for global memory --------------------------------------
// (1) the deflated input is staged through LDS, 512 bytes at a time, and the bit buffer is refilled from there by dwords;
// (2) everything the wave decides on is forced into scalar registers (readfirstlane behind every LDS read);
// (3) literals are gathered eight to a store; (4) matches are not copied when they are decoded: lane k remembers the k-th of them, and
// when 64 are at hand (or the block ends) they are resolved together - every lane whose source lies in front of the group's first match
// copies its own bytes, all loads in flight at once; the others (a source inside the group: runs, neighbours) and the long ones follow
// in order, the lanes side by side as before.
struct Tables2 {
    Tables t;
    uint32_t inbuf[256];
};
struct Bits2 {
    const uint32_t *g;     // the block's input from its dword-aligned start
    uint32_t *lds;         // 256 dwords: chunk c (dwords 128c..128c+127) lies in half c & 1
    uint32_t ndw;          // dwords that may be read (the buffer's padding included)
    uint32_t iw;           // next dword to take
    uint64_t buf;
    int cnt;
    int lane;
    __device__ inline void load_chunk(uint32_t c) {
        const uint32_t base = c * 128u;
        uint32_t a = base + (uint32_t)lane, bq = base + 64u + (uint32_t)lane;
        a = a < ndw ? a : ndw - 1; bq = bq < ndw ? bq : ndw - 1;
        const uint32_t x = g[a], y = g[bq];
        lds[(base & 255u) + (uint32_t)lane] = x;
        lds[(base & 255u) + 64u + (uint32_t)lane] = y;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // start reading at byte q of the input
    __device__ inline void seek(uint32_t q) {
        iw = q >> 2;
        load_chunk(iw >> 7);
        load_chunk((iw >> 7) + 1);
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[iw & 255u]);
        const uint32_t sh = 8u * (q & 3u);
        buf = (uint64_t)(w >> sh);
        cnt = 32 - (int)sh;
        iw++;
        if ((iw & 127u) == 0) load_chunk((iw >> 7) + 1);
    }
    __device__ inline uint32_t byte_pos() const { return iw * 4u - (uint32_t)(cnt >> 3); }   // (cnt a multiple of 8)
    __device__ inline void refill() {
        if (cnt <= 32) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[iw & 255u]);
            buf |= (uint64_t)w << cnt;
            cnt += 32;
            iw++;
            if ((iw & 127u) == 0) load_chunk((iw >> 7) + 1);
        }
    }
    __device__ inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    __device__ inline void drop(int n) { buf >>= n; cnt -= n; }
    __device__ inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

constexpr uint32_t SHORT_MATCH = 16;

__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK)
ginflate_kernel2(const uint8_t *__restrict__ in, uint64_t in_total, const Blk *__restrict__ blks, uint32_t n_blks, uint8_t *__restrict__ out, int *__restrict__ bad) {
    __shared__ Tables2 tabs[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (the compiler must know it is the same in every lane)
    const uint32_t bi = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (bi >= n_blks) return;
    Tables &T = tabs[wave].t;
    const Blk blk = blks[bi];
    Bits2 b;
    const uint64_t start = blk.in_off & ~3ull;
    b.g = (const uint32_t *)(in + start);
    b.lds = tabs[wave].inbuf;
    b.ndw = (uint32_t)((in_total - start) >> 2);
    b.lane = lane;
    b.seek((uint32_t)(blk.in_off - start));
    const uint32_t in_end = (uint32_t)(blk.in_off - start) + blk.in_len;   // the byte behind the block's input
    uint8_t *dst = out + blk.out_off;
    uint32_t pos = 0;
    bool fail = false;
    // literals not yet stored
    uint64_t lit_acc = 0;
    uint32_t lit_n = 0, lit_pos = 0;
    // matches not yet copied: lane k holds the k-th
    uint32_t ntok = 0, tok_pos = 0, tok_len = 0, tok_dist = 0;
    auto flush_lits = [&]() {
        if (lit_n) {
            if ((uint32_t)lane < lit_n) dst[lit_pos + lane] = (uint8_t)(lit_acc >> (8 * lane));
            lit_n = 0; lit_acc = 0;
        }
    };
    auto resolve = [&]() {
        flush_lits();
        if (ntok == 0) return;
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tok_pos);
        const bool active = (uint32_t)lane < ntok;
        const uint32_t src = tok_pos - tok_dist;
        const bool later = active && (src + tok_len > p0 || tok_len > SHORT_MATCH);
        if (active && !later) {
            uint8_t v[SHORT_MATCH];
#pragma unroll
            for (uint32_t j = 0; j < SHORT_MATCH; j++) v[j] = j < tok_len ? dst[src + j] : (uint8_t)0;
#pragma unroll
            for (uint32_t j = 0; j < SHORT_MATCH; j++) if (j < tok_len) dst[tok_pos + j] = v[j];
        }
        uint64_t todo = __ballot(later);
        while (todo) {
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)tok_pos, k), len = (uint32_t)__builtin_amdgcn_readlane((int)tok_len, k),
                           dist = (uint32_t)__builtin_amdgcn_readlane((int)tok_dist, k);
            const uint8_t *s = dst + p - dist;
            if (dist >= len) { for (uint32_t i = lane; i < len; i += 64) dst[p + i] = s[i]; }
            else if (dist == 1) { const uint8_t c = s[0]; for (uint32_t i = lane; i < len; i += 64) dst[p + i] = c; }
            else { for (uint32_t i = lane; i < len; i += 64) dst[p + i] = s[i % dist]; }
        }
        ntok = 0;
    };
    for (bool last = false; !last && !fail;) {
        b.refill();
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 0) {  // stored
            resolve();
            b.drop(b.cnt & 7);
            b.refill();
            const uint32_t n = b.take(16);
            b.refill();
            (void)b.take(16);
            const uint32_t q = b.byte_pos();
            if (q + n > in_end || pos + n > blk.out_len) { fail = true; break; }
            const uint8_t *s = (const uint8_t *)b.g + q;
            for (uint32_t i = lane; i < n; i += 64) dst[pos + i] = s[i];
            pos += n;
            b.seek(q + n);
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
            if (lane < 19) T.len[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int i = 0; i < ncl; i++) { b.refill(); const uint32_t v = b.take(3); if (lane == 0) T.len[CLORD[i]] = (uint8_t)v; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            build(T.len, 19, T.dcount, T.dsym, T.dfast, 7, 2, lane);
            uint8_t prev = 0;
            int i = 0;
            const int want = nlit + ndist;
            while (i < want && !fail) {
                b.refill();
                const uint32_t e = lookup(b, T.dfast, 7, T.dcount, T.dsym, 2);
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
        build(T.len, nlit, T.lcount, T.lsym, T.fast, FAST_BITS, 0, lane);
        build(T.len + nlit, ndist, T.dcount, T.dsym, T.dfast, DFAST_BITS, 1, lane);
        // ---- the symbols ----
        for (;;) {
            b.refill();
            if (b.byte_pos() > in_end + 8) { fail = true; break; }
            const uint32_t e = lookup(b, T.fast, FAST_BITS, T.lcount, T.lsym, 0);
            if (e & 0x100u) {
                if (pos >= blk.out_len) { fail = true; break; }
                if (lit_n == 0) lit_pos = pos;
                lit_acc |= (uint64_t)(e >> 16) << (8 * lit_n);
                lit_n++; pos++;
                if (lit_n == 8) flush_lits();
                continue;
            }
            if (e & 0x200u) break;   // end of block
            if ((e >> 16) == 0xFFFFu) { fail = true; break; }
            b.refill();
            const uint32_t len = (e >> 16) + b.take((int)((e >> 4) & 15u));
            const uint32_t d = lookup(b, T.dfast, DFAST_BITS, T.dcount, T.dsym, 1);
            if ((d >> 16) == 0xFFFFu) { fail = true; break; }
            b.refill();
            const uint32_t dist = (d >> 16) + b.take((int)((d >> 4) & 15u));
            if (dist > pos || pos + len > blk.out_len) { fail = true; break; }
            flush_lits();   // (the gathered literals lie side by side: a match between two of them ends the run)
            if ((uint32_t)lane == ntok) { tok_pos = pos; tok_len = len; tok_dist = dist; }
            ntok++;
            pos += len;
            if (ntok == 64) resolve();
        }
    }
    resolve();
    if ((fail || pos != blk.out_len) && lane == 0) atomicAdd(bad, 1);
}


// ---- the third decoder: the second one laid out for the instruction cache and the scalar unit ------------------------------------------
// Everything that is not the way from one symbol to the next is a function of its own (table builds, the canonical walk for long codes,
// a chunk of input into LDS, a group of matches resolved): the loop that runs 18 000 times per block is a few dozen instructions.
__device__ __noinline__ void build_nl(const uint8_t *len, int n, uint16_t *count, uint16_t *symbol, uint32_t *fast, int fast_bits, int kind, int lane) {
    build(len, n, count, symbol, fast, fast_bits, kind, lane);
}
// the code at the low end of `bits` (15 of them are enough) walked along the canonical order: symbol << 8 | length, or ~0u
__device__ __noinline__ uint32_t slow_code(uint32_t bits, const uint16_t *count, const uint16_t *symbol) {
    int code = 0, first = 0, index = 0;
    for (int L = 1; L <= 15; L++) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int c = __builtin_amdgcn_readfirstlane((int)count[L]);
        if (code - c < first) return ((uint32_t)__builtin_amdgcn_readfirstlane((int)symbol[index + (code - first)]) << 8) | (uint32_t)L;
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return ~0u;
}
__device__ __noinline__ uint32_t entry_of(uint32_t sym, int kind) {
    if (kind == 1) return sym < 30 ? ((uint32_t)DEXT[sym] << 4) | ((uint32_t)DBASE[sym] << 16) : 0xFFFF0000u;
    if (sym < 256) return 0x100u | (sym << 16);
    if (sym == 256) return 0x200u;
    return sym - 257 < 29 ? ((uint32_t)LEXT[sym - 257] << 4) | ((uint32_t)LBASE[sym - 257] << 16) : 0xFFFF0000u;
}
__device__ __noinline__ void load_chunk_nl(const uint32_t *g, uint32_t *lds, uint32_t ndw, uint32_t c, int lane) {
    const uint32_t base = c * 128u;
    uint32_t a = base + (uint32_t)lane, bq = base + 64u + (uint32_t)lane;
    a = a < ndw ? a : ndw - 1; bq = bq < ndw ? bq : ndw - 1;
    const uint32_t x = g[a], y = g[bq];
    lds[(base & 255u) + (uint32_t)lane] = x;
    lds[(base & 255u) + 64u + (uint32_t)lane] = y;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
__device__ __noinline__ void resolve_nl(uint8_t *dst, uint32_t tok_pos, uint32_t tok_len, uint32_t tok_dist, uint32_t ntok, int lane) {
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tok_pos);
    const bool active = (uint32_t)lane < ntok;
    const uint32_t src = tok_pos - tok_dist;
    const bool later = active && (src + tok_len > p0 || tok_len > SHORT_MATCH);
    if (active && !later) {
        uint8_t v[SHORT_MATCH];
#pragma unroll
        for (uint32_t j = 0; j < SHORT_MATCH; j++) v[j] = j < tok_len ? dst[src + j] : (uint8_t)0;
#pragma unroll
        for (uint32_t j = 0; j < SHORT_MATCH; j++) if (j < tok_len) dst[tok_pos + j] = v[j];
    }
    uint64_t todo = __ballot(later);
    while (todo) {
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)tok_pos, k), len = (uint32_t)__builtin_amdgcn_readlane((int)tok_len, k),
                       dist = (uint32_t)__builtin_amdgcn_readlane((int)tok_dist, k);
        const uint8_t *s = dst + p - dist;
        if (dist >= len) { for (uint32_t i = lane; i < len; i += 64) dst[p + i] = s[i]; }
        else if (dist == 1) { const uint8_t c = s[0]; for (uint32_t i = lane; i < len; i += 64) dst[p + i] = c; }
        else { for (uint32_t i = lane; i < len; i += 64) dst[p + i] = s[i % dist]; }
    }
}

struct Bits3 {
    const uint32_t *g;
    uint32_t *lds;
    uint32_t ndw, iw;
    uint64_t buf;
    int cnt;
    int lane;
    __device__ inline void seek(uint32_t q) {
        iw = q >> 2;
        load_chunk_nl(g, lds, ndw, iw >> 7, lane);
        load_chunk_nl(g, lds, ndw, (iw >> 7) + 1, lane);
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[iw & 255u]);
        const uint32_t sh = 8u * (q & 3u);
        buf = (uint64_t)(w >> sh);
        cnt = 32 - (int)sh;
        iw++;
        if ((iw & 127u) == 0) load_chunk_nl(g, lds, ndw, (iw >> 7) + 1, lane);
    }
    __device__ inline uint32_t byte_pos() const { return iw * 4u - (uint32_t)(cnt >> 3); }
    __device__ inline void refill() {
        if (cnt <= 32) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[iw & 255u]);
            buf |= (uint64_t)w << cnt;
            cnt += 32;
            iw++;
            if ((iw & 127u) == 0) load_chunk_nl(g, lds, ndw, (iw >> 7) + 1, lane);
        }
    }
    __device__ inline uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ inline void drop(int n) { buf >>= n; cnt -= n; }
    __device__ inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};
// a table entry for the code at hand: the fast table's, or the long way round
__device__ inline uint32_t lookup3(Bits3 &b, const uint32_t *fast, int fast_bits, const uint16_t *count, const uint16_t *symbol, int kind) {
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)fast[b.peek(fast_bits)]);
    if (__builtin_expect((e & 15u) != 0, 1)) { b.drop((int)(e & 15u)); return e; }
    // (what a function returns is "different in every lane" to the compiler unless it is told otherwise)
    const uint32_t sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)slow_code((uint32_t)b.buf & 0x7FFFu, count, symbol));
    if (sc == ~0u) return 0xFFFF0000u;
    b.drop((int)(sc & 255u));
    return kind == 2 ? (sc >> 8) << 16 : (uint32_t)__builtin_amdgcn_readfirstlane((int)entry_of(sc >> 8, kind));
}

__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK)
ginflate_kernel3(const uint8_t *__restrict__ in, uint64_t in_total, const Blk *__restrict__ blks, uint32_t n_blks, uint8_t *__restrict__ out, int *__restrict__ bad) {
    __shared__ Tables2 tabs[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t bi = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (bi >= n_blks) return;
    Tables &T = tabs[wave].t;
    const Blk blk = blks[bi];
    Bits3 b;
    const uint64_t start = blk.in_off & ~3ull;
    b.g = (const uint32_t *)(in + start);
    b.lds = tabs[wave].inbuf;
    b.ndw = (uint32_t)((in_total - start) >> 2);
    b.lane = lane;
    b.seek((uint32_t)(blk.in_off - start));
    const uint32_t in_end = (uint32_t)(blk.in_off - start) + blk.in_len;
    uint8_t *dst = out + blk.out_off;
    uint32_t pos = 0;
    bool fail = false;
    uint64_t lit_acc = 0;
    uint32_t lit_n = 0, lit_pos = 0;
    uint32_t ntok = 0, tok_pos = 0, tok_len = 0, tok_dist = 0;
#define FLUSH_LITS() do { if (lit_n) { if ((uint32_t)lane < lit_n) dst[lit_pos + lane] = (uint8_t)(lit_acc >> (8 * lane)); lit_n = 0; lit_acc = 0; } } while (0)
#define RESOLVE() do { FLUSH_LITS(); if (ntok) { resolve_nl(dst, tok_pos, tok_len, tok_dist, ntok, lane); ntok = 0; } } while (0)
    for (bool last = false; !last && !fail;) {
        b.refill();
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 0) {  // stored
            RESOLVE();
            b.drop(b.cnt & 7);
            b.refill();
            const uint32_t n = b.take(16);
            b.refill();
            (void)b.take(16);
            const uint32_t q = b.byte_pos();
            if (q + n > in_end || pos + n > blk.out_len) { fail = true; break; }
            const uint8_t *s = (const uint8_t *)b.g + q;
            for (uint32_t i = lane; i < n; i += 64) dst[pos + i] = s[i];
            pos += n;
            b.seek(q + n);
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
            if (lane < 19) T.len[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int i = 0; i < ncl; i++) { b.refill(); const uint32_t v = b.take(3); if (lane == 0) T.len[CLORD[i]] = (uint8_t)v; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            build_nl(T.len, 19, T.dcount, T.dsym, T.dfast, 7, 2, lane);
            uint8_t prev = 0;
            int i = 0;
            const int want = nlit + ndist;
            while (i < want && !fail) {
                b.refill();
                const uint32_t e = lookup3(b, T.dfast, 7, T.dcount, T.dsym, 2);
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
        build_nl(T.len, nlit, T.lcount, T.lsym, T.fast, FAST_BITS, 0, lane);
        build_nl(T.len + nlit, ndist, T.dcount, T.dsym, T.dfast, DFAST_BITS, 1, lane);
        // ---- the symbols ----
        const uint32_t out_len = blk.out_len;
        for (;;) {
            b.refill();
            const uint32_t e = lookup3(b, T.fast, FAST_BITS, T.lcount, T.lsym, 0);
            if (e & 0x100u) {
                if (lit_n == 0) lit_pos = pos;
                lit_acc |= (uint64_t)(e >> 16) << (8 * lit_n);
                lit_n++; pos++;
                if (lit_n == 8) { if (pos > out_len) { fail = true; break; } FLUSH_LITS(); }
                continue;
            }
            if (e & 0x200u) break;   // end of block
            if ((e >> 16) == 0xFFFFu || b.byte_pos() > in_end + 8) { fail = true; break; }
            b.refill();
            const uint32_t len = (e >> 16) + b.take((int)((e >> 4) & 15u));
            const uint32_t d = lookup3(b, T.dfast, DFAST_BITS, T.dcount, T.dsym, 1);
            if ((d >> 16) == 0xFFFFu) { fail = true; break; }
            b.refill();
            const uint32_t dist = (d >> 16) + b.take((int)((d >> 4) & 15u));
            if (dist > pos || pos + len > out_len) { fail = true; break; }
            FLUSH_LITS();
            if ((uint32_t)lane == ntok) { tok_pos = pos; tok_len = len; tok_dist = dist; }
            ntok++;
            pos += len;
            if (ntok == 64) { resolve_nl(dst, tok_pos, tok_len, tok_dist, ntok, lane); ntok = 0; }
        }
        if (pos > out_len) fail = true;
    }
    if (fail) { lit_n = 0; ntok = 0; }
    RESOLVE();
    if ((fail || pos != blk.out_len) && lane == 0) atomicAdd(bad, 1);
#undef FLUSH_LITS
#undef RESOLVE
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 512;
    const int level = argc > 2 ? atoi(argv[2]) : 4;
    // FASTQ text with HiFi-like qualities, 15 kb reads
    std::mt19937_64 rng(7);
    std::normal_distribution<double> q(60, 15);
    std::vector<uint8_t> text;
    text.reserve(mb << 20);
    const int L = 15000;
    for (uint64_t r = 0; text.size() + 2 * L + 40 < (mb << 20); r++) {
        char head[32];
        const int hn = snprintf(head, sizeof head, "@read%09llu c\n", (unsigned long long)r);
        text.insert(text.end(), head, head + hn);
        for (int i = 0; i < L; i++) text.push_back("ACGT"[rng() & 3]);
        text.push_back('\n'); text.push_back('+'); text.push_back('\n');
        for (int i = 0; i < L; i++) {
            double v = (rng() % 10) < 6 ? 93 : q(rng);
            v = v < 2 ? 2 : v > 93 ? 93 : v;
            text.push_back((uint8_t)(33 + (int)v));
        }
        text.push_back('\n');
    }
    // bgzf-like blocks: 60 000 bytes of text each, raw deflate
    std::vector<uint8_t> comp;
    std::vector<Blk> blks;
    std::vector<uint8_t> tmp(80000);
    for (size_t off = 0; off < text.size(); off += 60000) {
        const size_t n = std::min<size_t>(60000, text.size() - off);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = text.data() + off; zs.avail_in = (uInt)n;
        zs.next_out = tmp.data(); zs.avail_out = (uInt)tmp.size();
        deflate(&zs, Z_FINISH);
        const size_t cn = tmp.size() - zs.avail_out;
        deflateEnd(&zs);
        blks.push_back(Blk{comp.size(), off, (uint32_t)cn, (uint32_t)n});
        comp.insert(comp.end(), tmp.begin(), tmp.begin() + cn);
    }
    comp.resize(comp.size() + 64);
    printf("%.1f MB of text, %.1f MB deflated at level %d (%.3f), %zu blocks\n", text.size() / 1e6, comp.size() / 1e6, level, (double)comp.size() / text.size(), blks.size());
    uint8_t *d_in, *d_out;
    Blk *d_blks;
    int *d_bad;
    CHECK(hipMalloc(&d_in, comp.size()));
    CHECK(hipMalloc(&d_out, text.size() + 64));
    CHECK(hipMalloc(&d_blks, blks.size() * sizeof(Blk)));
    CHECK(hipMalloc(&d_bad, 4));
    CHECK(hipMemcpy(d_in, comp.data(), comp.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_blks, blks.data(), blks.size() * sizeof(Blk), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_bad, 0, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((blks.size() + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    unsigned long long *d_stats;
    CHECK(hipMalloc(&d_stats, 32));
    CHECK(hipMemset(d_stats, 0, 32));
    for (int rep = 0; rep < 12; rep++) {
        const int mode = rep / 3;
        CHECK(hipMemset(d_out, 0, text.size()));
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(ginflate_kernel<0>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, nullptr, d_in, d_blks, (uint32_t)blks.size(), d_out, d_bad, rep == 0 ? d_stats : nullptr);
        else if (mode == 3) hipLaunchKernelGGL(ginflate_kernel3, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, nullptr, d_in, (uint64_t)comp.size(), d_blks, (uint32_t)blks.size(), d_out, d_bad);
        else if (mode == 2) hipLaunchKernelGGL(ginflate_kernel2, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, nullptr, d_in, (uint64_t)comp.size(), d_blks, (uint32_t)blks.size(), d_out, d_bad);
        else hipLaunchKernelGGL(ginflate_kernel<1>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, nullptr, d_in, d_blks, (uint32_t)blks.size(), d_out, d_bad, (unsigned long long *)nullptr);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        int bad = 0;
        CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        std::vector<uint8_t> back(text.size());
        CHECK(hipMemcpy(back.data(), d_out, text.size(), hipMemcpyDeviceToHost));
        size_t wrong = 0;
        for (size_t i = 0; i < text.size(); i++) wrong += back[i] != text[i];
        if (rep == 0) {
            unsigned long long st[4];
            CHECK(hipMemcpy(st, d_stats, 32, hipMemcpyDeviceToHost));
            printf("symbols: %llu literals, %llu matches copying %llu bytes (%.1f each), %llu of them longer than 64; per block %.0f literals, %.0f matches\n", st[0], st[1], st[2],
                   (double)st[2] / (double)(st[1] ? st[1] : 1), st[3], (double)st[0] / blks.size(), (double)st[1] / blks.size());
        }
        printf("mode %d run %d: %.2f ms = %.2f GB/s of text (%.2f GB/s deflated); blocks that failed %d, wrong bytes %zu\n", mode, rep, ms, text.size() / ms / 1e6, comp.size() / ms / 1e6, bad, wrong);
        CHECK(hipMemset(d_bad, 0, 4));
    }
    return 0;
}
