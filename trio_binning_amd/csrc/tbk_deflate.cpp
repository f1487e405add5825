// tbk_deflate.cpp — the DEFLATE encoder behind the gzip members of the bin writer: entropy coding, and runs.
//
// FASTQ text has nothing for LZ77 to find inside a 32 KiB window (bases are four-symbol noise,
// qualities of long-read data vary from base to base), so what compresses it is entropy coding alone,
// and zlib's Z_HUFFMAN_ONLY does exactly that - through a general-purpose machine that spends most of
// its time not coding.  This file is the same format written directly: per block (a few KiB, cut at a line end) a
// byte histogram, a length-limited canonical Huffman code, the dynamic-block header (RFC 1951
// 3.2.7), then the literals through a 64-bit bit buffer.  Output is an ordinary gzip member (RFC
// 1952): any inflater reads it, decompressed bytes are what went in, CRC-32 and size in the trailer.
// The one redundancy LZ77 does find in FASTQ - runs of one byte (constant or binned qualities, the
// '~' stretches of HiFi reads) - is kept: a block whose bytes mostly repeat their predecessor is coded
// as literals + matches at distance 1, what zlib calls Z_RLE.  TBK_GZIP_ENCODER=zlib sends the bin
// writer's members through zlib instead.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

uint32_t tbk_crc32(uint32_t crc, const uint8_t *p, size_t n);  // tbk_crc.cpp

namespace {

// Bits go into a 64-bit accumulator and leave it eight bytes at a time through an unaligned store;
// the caller guarantees room (ensure()) before each burst, so the hot loop has no capacity checks.
struct BitWriter {
    std::vector<char> &out;
    size_t at;     // bytes of `out` in use (out.size() is the reserved room)
    uint64_t acc = 0;
    int n = 0;     // bits held in acc, < 8 after every flush
    explicit BitWriter(std::vector<char> &o) : out(o), at(o.size()) {}
    void ensure(size_t more) { if (out.size() < at + more + 16) out.resize(at + more + 16); }
    inline void add(uint32_t bits, int len) { acc |= (uint64_t)bits << n; n += len; }  // n + len <= 64
    inline void flush() {
        memcpy(&out[at], &acc, 8);
        at += (size_t)(n >> 3);
        acc >>= (n & ~7);
        n &= 7;
    }
    inline void put(uint32_t bits, int len) { add(bits, len); flush(); }  // len <= 32
    void finish() {  // pad to a byte boundary and give the vector its true size
        if (n > 0) { out[at++] = (char)(acc & 0xFF); acc = 0; n = 0; }
        out.resize(at);
    }
};

// Huffman code lengths (<= limit) for the symbols with freq > 0; the others get 0.  At least two
// symbols must have freq > 0.  Heap-free: nodes sorted once, two-queue merge; when the tree is deeper
// than the limit the frequencies are halved (never to zero) and the tree rebuilt.
void huffman_lengths(const uint32_t *freq_in, int n, int limit, uint8_t *len) {
    std::vector<uint32_t> freq(freq_in, freq_in + n);
    for (;;) {
        struct Node { uint64_t w; int left, right; };
        std::vector<int> order;
        for (int i = 0; i < n; i++) if (freq[(size_t)i]) order.push_back(i);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return freq[(size_t)a] != freq[(size_t)b] ? freq[(size_t)a] < freq[(size_t)b] : a < b; });
        const int m = (int)order.size();
        std::vector<Node> nodes;
        nodes.reserve((size_t)(2 * m));
        for (int i = 0; i < m; i++) nodes.push_back(Node{freq[(size_t)order[(size_t)i]], -1, -1});
        int leaf = 0, inner = m;  // next unused leaf / next unused inner node
        auto take = [&]() -> int {
            if (leaf < m && (inner >= (int)nodes.size() || nodes[(size_t)leaf].w <= nodes[(size_t)inner].w)) return leaf++;
            return inner++;
        };
        while ((int)nodes.size() < 2 * m - 1) {
            const int a = take(), b = take();
            nodes.push_back(Node{nodes[(size_t)a].w + nodes[(size_t)b].w, a, b});
        }
        std::vector<uint8_t> depth(nodes.size(), 0);
        int deepest = 0;
        for (int i = (int)nodes.size() - 1; i >= 0; i--) {
            if (nodes[(size_t)i].left >= 0) {
                depth[(size_t)nodes[(size_t)i].left] = depth[(size_t)nodes[(size_t)i].right] = (uint8_t)(depth[(size_t)i] + 1);
            } else if (depth[(size_t)i] > deepest) {
                deepest = depth[(size_t)i];
            }
        }
        if (deepest <= limit) {
            memset(len, 0, (size_t)n);
            for (int i = 0; i < m; i++) len[order[(size_t)i]] = depth[(size_t)i];
            return;
        }
        for (int i = 0; i < n; i++) if (freq[(size_t)i]) freq[(size_t)i] = (freq[(size_t)i] + 1) / 2;
    }
}

// canonical codes for the lengths, bit-reversed (DEFLATE packs Huffman codes starting at their most significant bit)
void canonical_codes(const uint8_t *len, int n, uint16_t *code) {
    uint32_t count[16] = {0}, next[16] = {0};
    for (int i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    uint32_t c = 0;
    for (int l = 1; l <= 15; l++) { c = (c + count[l - 1]) << 1; next[l] = c; }
    for (int i = 0; i < n; i++) {
        if (!len[i]) { code[i] = 0; continue; }
        uint32_t v = next[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; b++) { r = (r << 1) | (v & 1u); v >>= 1; }
        code[i] = (uint16_t)r;
    }
}

template <int G>
inline void encode_literals(BitWriter &bw, const uint8_t *src, size_t n, const uint32_t *sym) {
    // sym[v] = code | length << 16; G symbols of at most 56/G bits each per flush (7 bits may be waiting)
    size_t i = 0;
    for (; i + G <= n; i += G) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint32_t e = sym[src[i + (size_t)g]];
            bw.add(e & 0xFFFFu, (int)(e >> 16));
        }
        bw.flush();
    }
    for (; i < n; i++) { const uint32_t e = sym[src[i]]; bw.put(e & 0xFFFFu, (int)(e >> 16)); }
}

// length symbol (257..285), extra bits and their value for a match length 3..258 (RFC 1951 3.2.5)
struct LenCode { uint16_t sym; uint8_t extra_bits; uint8_t extra; };
const LenCode *length_codes() {
    static LenCode table[259];
    static const bool once = [] {
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t bits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        for (int len = 3; len <= 258; len++) {
            int c = 28;
            while (base[c] > len) c--;
            if (len == 258) c = 28;
            table[len] = LenCode{(uint16_t)(257 + c), bits[c], (uint8_t)(len - base[c])};
        }
        return true;
    }();
    (void)once;
    return table;
}

// header of a dynamic block for `nlit` literal/length code lengths followed by one distance code length
void put_block_header(BitWriter &bw, const uint8_t *len, int nlit, bool final) {
    const int total = nlit + 1;
    // the code lengths as code-length symbols: lengths as they are, runs of zeros as 17 (3-10) / 18 (11-138)
    struct Cl { uint8_t sym, extra_bits; uint16_t extra; };
    Cl cl[288];
    int ncl = 0;
    for (int k = 0; k < total;) {
        if (len[k] != 0) { cl[ncl++] = Cl{len[k], 0, 0}; k++; continue; }
        int run = 1;
        while (k + run < total && len[k + run] == 0) run++;
        int left = run;
        while (left >= 11) { const int r = std::min(left, 138); cl[ncl++] = Cl{18, 7, (uint16_t)(r - 11)}; left -= r; }
        if (left >= 3) { cl[ncl++] = Cl{17, 3, (uint16_t)(left - 3)}; left = 0; }
        while (left-- > 0) cl[ncl++] = Cl{0, 0, 0};
        k += run;
    }
    uint32_t clfreq[19] = {0};
    for (int k = 0; k < ncl; k++) clfreq[cl[k].sym]++;
    int distinct = 0;
    for (int k = 0; k < 19; k++) distinct += clfreq[k] != 0;
    if (distinct < 2) clfreq[clfreq[0] ? 1 : 0]++;  // a code needs two symbols to be complete
    uint8_t cllen[19];
    huffman_lengths(clfreq, 19, 7, cllen);
    uint16_t clcode[19];
    canonical_codes(cllen, 19, clcode);
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cllen[order[hclen - 1]] == 0) hclen--;
    bw.put(final ? 1u : 0u, 1);
    bw.put(2u, 2);                        // dynamic Huffman codes
    bw.put((uint32_t)(nlit - 257), 5);    // HLIT
    bw.put(0u, 5);                        // HDIST: 1 distance code
    bw.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; k++) bw.put(cllen[order[k]], 3);
    for (int k = 0; k < ncl; k++) {
        bw.put(clcode[cl[k].sym], cllen[cl[k].sym]);
        if (cl[k].extra_bits) bw.put(cl[k].extra, cl[k].extra_bits);
    }
}

// A block whose bytes come in runs (constant or binned qualities, homopolymers): a run is its first
// byte as a literal, then matches at distance 1 - the only distance this encoder knows, so the
// distance code is one 1-bit symbol.  Tokens: a literal is its value, a match is 0x8000 | length.
void encode_block_runs(BitWriter &bw, const uint8_t *src, size_t n, bool final) {
    static thread_local std::vector<uint16_t> tokens;
    tokens.clear();
    tokens.reserve(n);
    uint32_t freq[286] = {0};
    const LenCode *lc = length_codes();
    for (size_t i = 0; i < n;) {
        const uint8_t v = src[i];
        size_t run = 1;
        while (i + run < n && src[i + run] == v) run++;
        tokens.push_back(v);
        freq[v]++;
        size_t left = run - 1;
        while (left >= 3) {
            size_t m = std::min<size_t>(left, 258);
            if (left - m > 0 && left - m < 3) m = left - 3;  // never strand one or two bytes behind a full match
            tokens.push_back((uint16_t)(0x8000u | m));
            freq[lc[m].sym]++;
            left -= m;
        }
        for (; left > 0; left--) { tokens.push_back(v); freq[v]++; }
        i += run;
    }
    freq[256] = 1;
    int nlit = 286;
    while (nlit > 257 && freq[nlit - 1] == 0) nlit--;
    uint8_t len[287];
    huffman_lengths(freq, nlit, 15, len);
    uint16_t code[286];
    canonical_codes(len, nlit, code);
    len[nlit] = 1;  // the distance code: one symbol (distance 1), one bit, value 0
    bw.ensure(2 * n + 1024);
    put_block_header(bw, len, nlit, final);
    for (size_t t = 0; t < tokens.size(); t++) {
        const uint16_t tok = tokens[t];
        if (tok < 0x8000u) {
            bw.add(code[tok], len[tok]);
        } else {
            const LenCode &l = lc[tok & 0x7FFFu];
            bw.add(code[l.sym], len[l.sym]);
            bw.add(l.extra, l.extra_bits);
            bw.add(0u, 1);
        }
        bw.flush();  // at most 15 + 5 + 1 bits per token on top of 7 waiting
    }
    bw.put(code[256], len[256]);
}

void encode_block(BitWriter &bw, const uint8_t *src, size_t n, bool final) {
    // four interleaved histograms: FASTQ repeats a handful of byte values, and one counter per value
    // would serialise on its own store; the same pass counts bytes that repeat their predecessor
    uint32_t h[4][256];
    memset(h, 0, sizeof(h));
    size_t i = 0, same = 0;
    for (; i + 4 <= n; i += 4) { h[0][src[i]]++; h[1][src[i + 1]]++; h[2][src[i + 2]]++; h[3][src[i + 3]]++; }
    for (; i < n; i++) h[0][src[i]]++;
    for (i = 1; i < n; i++) same += src[i] == src[i - 1];
    // uniform random bases repeat their predecessor a quarter of the time and runs of four or more
    // cover 1.6 % of them: nothing to gain.  Past 40 % there are real runs.
    if (same * 5 > n * 2) return encode_block_runs(bw, src, n, final);
    uint32_t freq[257];
    for (int v = 0; v < 256; v++) freq[v] = h[0][v] + h[1][v] + h[2][v] + h[3][v];
    freq[256] = 1;  // end of block
    uint8_t len[258];
    huffman_lengths(freq, 257, 15, len);
    len[257] = 0;  // the one distance code: unused (literals only)
    uint16_t code[257];
    canonical_codes(len, 257, code);
    bw.ensure(2 * n + 1024);  // literals take at most 15 bits each, the header under 400 bytes
    put_block_header(bw, len, 257, final);
    uint32_t sym[256];
    int longest = 0;
    for (int v = 0; v < 256; v++) { sym[v] = (uint32_t)code[v] | ((uint32_t)len[v] << 16); longest = std::max<int>(longest, len[v]); }
    if (longest <= 11) encode_literals<5>(bw, src, n, sym);
    else if (longest <= 14) encode_literals<4>(bw, src, n, sym);
    else encode_literals<3>(bw, src, n, sym);
    bw.put(code[256], len[256]);
}

}  // namespace

// One gzip member holding src[0..n), entropy-coded only.  Appends to `out` (which is cleared first).
bool tbk_gzip_member_literal(const char *src, size_t n, std::vector<char> &out) {
    out.clear();
    out.reserve(n / 2 + 64);
    static const unsigned char head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    out.insert(out.end(), (const char *)head, (const char *)head + 10);
    BitWriter bw(out);
    if (n == 0) {
        bw.ensure(16);
        bw.put(1u, 1); bw.put(1u, 2); bw.put(0u, 7);  // final block, fixed codes, end of block
    } else {
        // Long lines (the second line of a block - after "@name" or "+" - runs past 2 KiB): a block ends
        // at the first line end past 8 KiB, 64 KiB at most, so that the bases of a long read and its
        // qualities get codes of their own - 2 bits a base instead of a code stretched over both
        // alphabets.  Short lines: bases and qualities alternate too fast to separate, and blocks of
        // 32 KiB keep the ~70-byte header under 1/4 %.
        static const size_t pinned = getenv("TBK_GZIP_BLOCK") ? (size_t)atol(getenv("TBK_GZIP_BLOCK")) : 0;
        for (size_t off = 0; off < n;) {
            size_t m = n - off;
            size_t least = pinned;
            if (!least) {
                const char *first = (const char *)memchr(src + off, '\n', std::min<size_t>(m, 2048));
                const size_t used = first ? (size_t)(first - (src + off)) + 1 : m;
                const bool short_lines = first && (used == m || memchr(first + 1, '\n', std::min<size_t>(m - used, 2048)) != nullptr);
                least = short_lines ? (size_t)32 << 10 : (size_t)8 << 10;
            }
            const size_t most = std::max<size_t>(least, (size_t)64 << 10);
            if (m > least) {
                const size_t span = std::min(m, most) - least;
                const void *nl = memchr(src + off + least, '\n', span);
                m = nl ? (size_t)((const char *)nl - (src + off)) + 1 : least + span;
            }
            encode_block(bw, (const uint8_t *)src + off, m, off + m == n);
            off += m;
        }
    }
    bw.finish();
    uLong crc = crc32(0L, Z_NULL, 0);
    crc = tbk_crc32((uint32_t)crc, (const uint8_t *)src, n);
    const uint32_t tail[2] = {(uint32_t)crc, (uint32_t)(n & 0xFFFFFFFFu)};
    out.insert(out.end(), (const char *)tail, (const char *)tail + 8);
    return true;
}

// C-ABI (include/tbk.h): the member for a host buffer
extern "C" int tbk_gzip_member(const char *src, size_t n, char *dst, size_t cap, size_t *len) {
    if ((!src && n) || !len) return -1;  // TBK_ERR_INVALID
    std::vector<char> out;
    tbk_gzip_member_literal(src, n, out);
    *len = out.size();
    if (out.size() > cap || !dst) return -6;  // TBK_ERR_NOMEM: *len says what is needed
    memcpy(dst, out.data(), out.size());
    return 0;
}
