// tbk_pack.cpp — the packed transfer format of the streaming path, host side.
//
// A read batch crosses PCIe as 1 byte per base in the reference-shaped interface (ASCII, as
// c/kmers.c:270-276 takes its reads), and PCIe Gen5 x16 (~56 GB/s) then caps a GPU at ~56 Gbases/s
// while the probe kernel does ~150.  The kernel's first act on a 16-base chunk is to turn it into a
// 32-bit word of 2-bit codes and a 16-bit not-ACGT mask (tbk_device.h: pack16); this file does the
// same on the host, so that what crosses the link is
//     codes[chunk]                       uint32, base i of the chunk at bits 2i..2i+1 (A=0 C=1 G=2 T=3,
//                                        c/kmers.c:50-72; the two bits of any other byte mean nothing:
//                                        its mask bit invalidates every window over it)
//     (exc_chunk[j], exc_mask[j])        for the chunks holding a byte outside ACGT (or stream
//                                        positions at or past the end): their index and 16-bit mask
// = 0.25 bytes per base on clean reads.  On the device the exceptions are scattered into a dense
// 16-bit-per-chunk mask array and the probe kernel loads (codes, mask) instead of ASCII - the
// representation it works on anyway, so counts are identical by construction (and by test).
#include <immintrin.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/tbk.h"
#include "tbk_pack.h"

extern "C" void tbk_set_error_(int code, const char *msg);

using Exc = TbkExc;

namespace {

// 8 ASCII bases -> 16 bits of codes, 8 bits of "not ACGT" (the device's pack4, 64 bits wide)
inline void pack8(uint64_t w, uint32_t &code16, uint32_t &bad8) {
    uint64_t c = ((w >> 1) ^ (w >> 2)) & 0x0303030303030303ull;
    const uint64_t lo = c & 0x0101010101010101ull, hi = (c >> 1) & 0x0101010101010101ull;
    const uint64_t expect = 0x4141414141414141ull + 2 * lo + 6 * hi + 11 * (lo & hi);
    const uint64_t diff = w ^ expect;
    const uint64_t nz = ((((diff & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | diff) & 0x8080808080808080ull) >> 7;
    bad8 = (uint32_t)((nz * 0x0102040810204080ull) >> 56);
    c |= c >> 6;
    c = (c | (c >> 12)) & 0x000000FF000000FFull;
    code16 = (uint32_t)((c | (c >> 24)) & 0xFFFFu);
}

inline void pack16_scalar(const uint8_t *p, uint32_t &code, uint32_t &bad) {
    uint64_t a, b;
    memcpy(&a, p, 8);
    memcpy(&b, p + 8, 8);
    uint32_t c0, c1, b0, b1;
    pack8(a, c0, b0);
    pack8(b, c1, b1);
    code = c0 | (c1 << 16);
    bad = b0 | (b1 << 8);
}

void pack_range_scalar(const uint8_t *bases, uint64_t c_lo, uint64_t c_hi, uint32_t *codes, std::vector<Exc> &exc) {
    for (uint64_t c = c_lo; c < c_hi; c++) {
        uint32_t code, bad;
        pack16_scalar(bases + c * 16, code, bad);
        codes[c] = code;
        if (bad) exc.push_back(Exc{(uint32_t)c, (uint16_t)bad});
    }
}

__attribute__((target("avx2"))) void pack_range_avx2(const uint8_t *bases, uint64_t c_lo, uint64_t c_hi, uint32_t *codes, std::vector<Exc> &exc) {
    const __m256i three = _mm256_set1_epi8(3);
    const __m256i lut = _mm256_setr_epi8(0x41, 0x43, 0x47, 0x54, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x41, 0x43, 0x47, 0x54, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i w1 = _mm256_set1_epi16(0x0401);      // code(byte 0) + 4 * code(byte 1)
    const __m256i w2 = _mm256_set1_epi32(0x00100001);  // pair 0 + 16 * pair 1
    const __m256i order = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7);
    uint64_t c = c_lo;
    // 128 bases = 8 chunks per step: four 32-byte loads, one 32-byte store of 8 code words
    for (; c + 8 <= c_hi; c += 8) {
        __m256i m[4];
        uint32_t good[4];
#pragma GCC unroll 4
        for (int i = 0; i < 4; i++) {
            const __m256i v = _mm256_loadu_si256((const __m256i *)(bases + (c + 2 * (uint64_t)i) * 16));
            const __m256i code = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), three);
            good[i] = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_shuffle_epi8(lut, code)));
            m[i] = _mm256_madd_epi16(_mm256_maddubs_epi16(code, w1), w2);  // 8 dwords, each the 8 code bits of 4 bases
        }
        // dwords -> words -> bytes (the packs work per 128-bit lane: lane 0 collects every load's low chunk,
        // lane 1 its high chunk), then the chunks' dwords back into stream order
        const __m256i b = _mm256_packus_epi16(_mm256_packus_epi32(m[0], m[1]), _mm256_packus_epi32(m[2], m[3]));
        _mm256_storeu_si256((__m256i *)(codes + c), _mm256_permutevar8x32_epi32(b, order));
        if ((good[0] & good[1] & good[2] & good[3]) != 0xFFFFFFFFu) {
            for (int i = 0; i < 4; i++) {
                const uint32_t bad = ~good[i];
                if (bad & 0xFFFFu) exc.push_back(Exc{(uint32_t)(c + 2 * (uint64_t)i), (uint16_t)(bad & 0xFFFFu)});
                if (bad >> 16) exc.push_back(Exc{(uint32_t)(c + 2 * (uint64_t)i + 1), (uint16_t)(bad >> 16)});
            }
        }
    }
    if (c < c_hi) pack_range_scalar(bases, c, c_hi, codes, exc);
}

bool have_avx2() {
    static const bool yes = __builtin_cpu_supports("avx2") && !(getenv("TBK_NO_AVX2") && *getenv("TBK_NO_AVX2") == '1');
    return yes;
}

}  // namespace

extern "C" uint64_t tbk_packed_chunks(uint64_t total_bases) { return (total_bases + 15) / 16; }

// The full chunks [c_lo, c_hi) of a base stream, on the calling thread (the FASTX reader packs a record's
// chunks right after it copied the record, while the bytes are still in its cache).
void tbk_pack_chunk_range_(const uint8_t *bases, uint64_t c_lo, uint64_t c_hi, uint32_t *codes, std::vector<TbkExc> &exc) {
    if (c_lo >= c_hi) return;
    if (have_avx2()) pack_range_avx2(bases, c_lo, c_hi, codes, exc);
    else pack_range_scalar(bases, c_lo, c_hi, codes, exc);
}

// The last, partial chunk of a stream of `total` bases (total % 16 != 0): positions at or past the end
// count as not ACGT.
void tbk_pack_tail_chunk_(const uint8_t *bases, uint64_t total, uint32_t *codes, std::vector<TbkExc> &exc) {
    const uint64_t full = total / 16;
    const unsigned valid = (unsigned)(total - full * 16);
    if (!valid) return;
    tbk_pack_tail_bytes_(bases + full * 16, valid, full, codes, exc);
}

// The same from the `valid` (1..15) bytes themselves, wherever they lie: chunk `chunk` of the stream.
void tbk_pack_tail_bytes_(const uint8_t *bytes, unsigned valid, uint64_t chunk, uint32_t *codes, std::vector<TbkExc> &exc) {
    uint8_t tmp[16] = {0};
    memcpy(tmp, bytes, valid);
    uint32_t code, bad;
    pack16_scalar(tmp, code, bad);
    bad |= 0xFFFFu << valid;
    code &= (1u << (2 * valid)) - 1u;
    codes[chunk] = code;
    exc.push_back(TbkExc{(uint32_t)chunk, (uint16_t)bad});
}

// `n` full chunks whose bytes lie at `src` (not in a stream buffer: a record in a mapped file, a staging buffer):
// chunks [c_lo, c_lo + n) of the stream.
void tbk_pack_span_(const uint8_t *src, uint64_t c_lo, uint64_t n, uint32_t *codes, std::vector<TbkExc> &exc) {
    if (!n) return;
    const uint8_t *base = (const uint8_t *)((uintptr_t)src - (uintptr_t)(16 * c_lo));  // the stream this span would be part of
    tbk_pack_chunk_range_(base, c_lo, c_lo + n, codes, exc);
}

// Pack `total` ASCII bases.  codes must hold tbk_packed_chunks(total) words.  The exceptions of the
// whole stream are appended to exc (ordered by chunk).  `threads` <= 0: tbk_host_threads().
int tbk_pack_bases_vec(const uint8_t *bases, uint64_t total, uint32_t *codes, std::vector<uint32_t> &exc_chunk, std::vector<uint16_t> &exc_mask,
                       int threads) {
    exc_chunk.clear();
    exc_mask.clear();
    const uint64_t full = total / 16, n_chunks = tbk_packed_chunks(total);
    if (n_chunks > 0xFFFFFFF0ull) { tbk_set_error_(TBK_ERR_INVALID, "more than 2^36 bases in one batch"); return TBK_ERR_INVALID; }
    if (threads <= 0) threads = tbk_host_threads();
    const uint64_t per_thread_min = (uint64_t)1 << 18;  // 4 Mbases: below that a thread is not worth starting
    int nt = (int)std::min<uint64_t>((uint64_t)threads, std::max<uint64_t>(1, full / per_thread_min));
    std::vector<std::vector<Exc>> found((size_t)nt);
    auto work = [&](int t) {
        const uint64_t lo = full * (uint64_t)t / (uint64_t)nt, hi = full * (uint64_t)(t + 1) / (uint64_t)nt;
        if (have_avx2()) pack_range_avx2(bases, lo, hi, codes, found[(size_t)t]);
        else pack_range_scalar(bases, lo, hi, codes, found[(size_t)t]);
    };
    // Threads are started per call on purpose.  A persistent pool woken through a condition variable was
    // tried and measured on the GPU box (16 CPUs' worth of quota on a 256-thread host): 78-83 Gbases/s
    // through tbk_stream_submit against 123-138 with fresh threads, three interleaved runs each - woken
    // workers are placed next to their waker and stack up on a few cores for the 7 ms a batch takes,
    // fresh threads are placed on idle cores.
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
        work(0);
        for (std::thread &th : pool) th.join();
    }
    size_t n_exc = 0;
    for (const auto &f : found) n_exc += f.size();
    exc_chunk.reserve(n_exc + 1);
    exc_mask.reserve(n_exc + 1);
    for (const auto &f : found)
        for (const Exc &e : f) { exc_chunk.push_back(e.chunk); exc_mask.push_back(e.mask); }
    if (full < n_chunks) {  // the last, partial chunk: positions at or past the end count as not ACGT
        uint8_t tmp[16] = {0};
        const unsigned valid = (unsigned)(total - full * 16);
        memcpy(tmp, bases + full * 16, valid);
        uint32_t code, bad;
        pack16_scalar(tmp, code, bad);
        bad |= 0xFFFFu << valid;
        code &= valid >= 16 ? 0xFFFFFFFFu : ((1u << (2 * valid)) - 1u);
        codes[full] = code;
        exc_chunk.push_back((uint32_t)full);
        exc_mask.push_back((uint16_t)bad);
    }
    return TBK_OK;
}

extern "C" int tbk_pack_bases(const uint8_t *bases, uint64_t total_bases, uint32_t *codes, uint32_t *exc_chunk, uint16_t *exc_mask,
                              uint64_t exc_capacity, uint64_t *n_exc) {
    if ((total_bases && (!bases || !codes)) || !n_exc) { tbk_set_error_(TBK_ERR_INVALID, "NULL argument"); return TBK_ERR_INVALID; }
    std::vector<uint32_t> ec;
    std::vector<uint16_t> em;
    const int rc = tbk_pack_bases_vec(bases, total_bases, codes, ec, em, 0);
    if (rc) return rc;
    *n_exc = ec.size();
    if (ec.size() > exc_capacity) {
        tbk_set_error_(TBK_ERR_NOMEM, "exception arrays too small (see *n_exc)");
        return TBK_ERR_NOMEM;
    }
    if (!ec.empty()) {
        if (!exc_chunk || !exc_mask) { tbk_set_error_(TBK_ERR_INVALID, "NULL argument"); return TBK_ERR_INVALID; }
        memcpy(exc_chunk, ec.data(), ec.size() * sizeof(uint32_t));
        memcpy(exc_mask, em.data(), em.size() * sizeof(uint16_t));
    }
    return TBK_OK;
}
