// tbk_inflate.cpp — see tbk_inflate.h.
#include "tbk_inflate.h"

namespace {
const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769,
                                1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}
}  // namespace

// gzip member header (RFC 1952 2.3), at a byte boundary; zero padding between members is skipped
bool TbkInflate::parse_header() {
    const uint8_t *p = ip_;
    while (p < in_end_ && *p == 0) p++;  // padding (Python's gzip skips it as well)
    if (p == in_end_) { ip_ = p; return true; }
    if (in_end_ - p < 18) { err_ = "truncated gzip file"; return false; }
    if (p[0] != 0x1f || p[1] != 0x8b) { err_ = "not a gzip member"; return false; }
    if (p[2] != 8 || (p[3] & 0xE0)) { err_ = "unsupported gzip header"; return false; }
    const int flg = p[3];
    p += 10;
    if (flg & 4) {  // FEXTRA
        if (in_end_ - p < 2) { err_ = "truncated gzip file"; return false; }
        const size_t xlen = p[0] | ((size_t)p[1] << 8);
        p += 2;
        if ((size_t)(in_end_ - p) < xlen) { err_ = "truncated gzip file"; return false; }
        p += xlen;
    }
    for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
        if (flg & f) {
            while (p < in_end_ && *p) p++;
            if (p == in_end_) { err_ = "truncated gzip file"; return false; }
            p++;
        }
    if (flg & 2) {  // FHCRC
        if (in_end_ - p < 2) { err_ = "truncated gzip file"; return false; }
        p += 2;
    }
    ip_ = p;
    bitbuf_ = 0; bitcnt_ = 0;
    state_ = BLOCK_HEAD;
    return true;
}

// Canonical Huffman code -> lookup table.  Codes no longer than `primary_bits` fill the primary
// table directly (bit-reversed, replicated); longer ones go through a sub-table per primary prefix.
bool TbkInflate::build(const uint8_t *lens, int n, uint32_t *table, int table_size, int primary_bits, bool is_dist, bool is_codes) {
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    uint32_t next_code[16];
    uint32_t code = 0;
    long kraft = 0;
    for (int len = 1; len <= 15; len++) {
        code = (code + (uint32_t)count[len - 1]) << 1;
        next_code[len] = code;
        kraft += (long)count[len] << (15 - len);
    }
    if (kraft > (1L << 15)) { err_ = "over-subscribed Huffman code"; return false; }
    // An incomplete code is refused as zlib refuses it (inftrees.c): only a literal/length or
    // distance code made of a single 1-bit code may leave part of the code space unused (and a
    // code with no symbols at all is a table of invalid entries).
    {
        int used = 0, longest_len = 0;
        for (int len = 1; len <= 15; len++) if (count[len]) { used += count[len]; longest_len = len; }
        if (used > 0 && kraft < (1L << 15) && (is_codes || !(used == 1 && longest_len == 1))) { err_ = "incomplete Huffman code"; return false; }
    }
    const int primary_size = 1 << primary_bits;
    const uint32_t bad = entry(0, BAD, 0, 1);
    for (int i = 0; i < table_size; i++) table[i] = bad;
    // longest code behind each primary prefix
    uint8_t longest[1 << LBITS];
    memset(longest, 0, (size_t)primary_size);
    uint32_t codes[320];
    for (int sym = 0; sym < n; sym++) {
        const int len = lens[sym];
        if (!len) continue;
        const uint32_t rev = reverse_bits(next_code[len]++, len);
        codes[sym] = rev;
        if (len > primary_bits) {
            uint8_t &m = longest[rev & (uint32_t)(primary_size - 1)];
            if (len > m) m = (uint8_t)len;
        }
    }
    int next_free = primary_size;
    uint16_t sub_off[1 << LBITS];
    for (int p = 0; p < primary_size; p++) {
        if (!longest[p]) continue;
        const int sub_bits = longest[p] - primary_bits;
        if (next_free + (1 << sub_bits) > table_size) { err_ = "Huffman table overflow"; return false; }
        sub_off[p] = (uint16_t)next_free;
        table[p] = entry((uint32_t)next_free, SUB, (uint32_t)sub_bits, (uint32_t)primary_bits);
        next_free += 1 << sub_bits;
    }
    for (int sym = 0; sym < n; sym++) {
        const int len = lens[sym];
        if (!len) continue;
        uint32_t e;
        if (is_dist) {
            e = sym < 30 ? entry(kDistBase[sym], DIST, kDistExtra[sym], (uint32_t)len) : bad;
        } else if (sym < 256) {
            e = entry((uint32_t)sym, LIT, 0, (uint32_t)len);
        } else if (sym == 256) {
            e = entry(0, EOB, 0, (uint32_t)len);
        } else {
            e = sym <= 285 ? entry(kLenBase[sym - 257], LEN, kLenExtra[sym - 257], (uint32_t)len) : bad;
        }
        const uint32_t rev = codes[sym];
        if (len <= primary_bits) {
            for (uint32_t i = rev; i < (uint32_t)primary_size; i += 1u << len) table[i] = e;
        } else {
            const uint32_t p = rev & (uint32_t)(primary_size - 1);
            const int sub_bits = longest[p] - primary_bits;
            for (uint32_t j = rev >> primary_bits; j < (1u << sub_bits); j += 1u << (len - primary_bits)) table[sub_off[p] + j] = e;
        }
    }
    return true;
}

// Where a primary-table index holds a whole literal code and, behind it, a second whole literal
// code, let one lookup decode both: FASTQ is mostly literals with short codes (2-3 bits for bases).
void TbkInflate::pair_literals() {
    constexpr int N = 1 << LBITS;
    static thread_local uint32_t single[N];
    memcpy(single, lit_, sizeof single);
    for (int i = 0; i < N; i++) {
        const uint32_t e1 = single[i];
        if (((e1 >> 12) & 15u) != LIT) continue;
        const int l1 = (int)(e1 & 0xFF), rem = LBITS - l1;
        if (rem < 1) continue;
        const uint32_t e2 = single[i >> l1];  // the bits above `rem` read as zero: right whenever the code fits in `rem`
        if (((e2 >> 12) & 15u) != LIT || (int)(e2 & 0xFF) > rem) continue;
        lit_[i] = entry((e1 >> 16) | ((e2 >> 16) << 8), LIT2, 0, (uint32_t)(l1 + (int)(e2 & 0xFF)));
    }
}

void TbkInflate::fixed_tables() {
    uint8_t lens[288];
    for (int i = 0; i < 144; i++) lens[i] = 8;
    for (int i = 144; i < 256; i++) lens[i] = 9;
    for (int i = 256; i < 280; i++) lens[i] = 7;
    for (int i = 280; i < 288; i++) lens[i] = 8;
    build(lens, 288, lit_, LSIZE, LBITS, false);
    pair_literals();
    uint8_t dl[32];
    for (int i = 0; i < 32; i++) dl[i] = 5;
    build(dl, 32, dist_, DSIZE, DBITS, true);
}

bool TbkInflate::dynamic_tables() {
    refill();
    const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
    if (hlit > 286 || hdist > 30) { err_ = "bad dynamic block header"; return false; }
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) {
        if (bitcnt_ < 3) refill();
        cl[kClOrder[i]] = (uint8_t)take(3);
    }
    uint32_t cltab[1 << 7];
    if (!build(cl, 19, cltab, 1 << 7, 7, false, true)) return false;  // symbols 0..18 decode as "literals"
    uint8_t lens[320];
    int i = 0;
    while (i < hlit + hdist) {
        refill();
        const uint32_t e = cltab[bitbuf_ & 127u];
        if (((e >> 12) & 15u) != LIT) { err_ = "bad code-length code"; return false; }
        take((int)(e & 0xFF));
        const int sym = (int)(e >> 16);
        if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
        int rep, val = 0;
        if (sym == 16) {
            if (i == 0) { err_ = "repeat with no previous length"; return false; }
            val = lens[i - 1]; rep = 3 + (int)take(2);
        } else if (sym == 17) {
            rep = 3 + (int)take(3);
        } else {
            rep = 11 + (int)take(7);
        }
        if (i + rep > hlit + hdist) { err_ = "code lengths overrun"; return false; }
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) { err_ = "no end-of-block code"; return false; }
    if (!build(lens, hlit, lit_, LSIZE, LBITS, false)) return false;
    pair_literals();
    return build(lens + hlit, hdist, dist_, DSIZE, DBITS, true);
}

bool TbkInflate::read_block_head() {
    refill();
    last_block_ = take(1) != 0;
    const uint32_t type = take(2);
    if (type == 0) {
        byte_align_and_unread();
        if (in_end_ - ip_ < 4) { err_ = "truncated gzip file"; return false; }
        const uint32_t len = ip_[0] | ((uint32_t)ip_[1] << 8), nlen = ip_[2] | ((uint32_t)ip_[3] << 8);
        if ((len ^ nlen) != 0xFFFFu) { err_ = "corrupt stored block"; return false; }
        ip_ += 4;
        stored_left_ = len;
        state_ = STORED;
        return true;
    }
    if (type == 1) { fixed_tables(); state_ = HUFF; return true; }
    if (type == 2) { if (!dynamic_tables()) return false; state_ = HUFF; return true; }
    err_ = "bad block type";
    return false;
}

TbkInflate::Status TbkInflate::run(uint8_t *out, size_t *pos, size_t cap, size_t member_start) {
    return run_impl<uint8_t>(out, pos, cap, member_start, ~0ull);
}

TbkInflate::Status TbkInflate::run16(uint16_t *out, size_t *pos, size_t cap, uint64_t stop_bit) {
    return run_impl<uint16_t>(out, pos, cap, 0, stop_bit);
}

// T = uint8_t: the text.  T = uint16_t: one symbol per element - a byte value, or whatever the caller put
// in front of the write position and a match copied from there (LineSource::pinflate_loop in tbk_fastx.cpp: markers for a
// window that is not known yet).
template <class T>
TbkInflate::Status TbkInflate::run_impl(T *out, size_t *pos, size_t cap, size_t member_start, uint64_t stop_bit) {
    T *op = out + *pos;
    T *const oend = out + cap;
    const T *const floor_ = out + member_start;  // matches may not start before this
    constexpr uint32_t LMASK = (1u << LBITS) - 1, DMASK = (1u << DBITS) - 1;
    for (;;) {
        switch (state_) {
        case HEADER:
            if (!parse_header()) return ERROR;
            if (state_ == HEADER) { *pos = (size_t)(op - out); return INPUT_DONE; }  // nothing but padding left
            break;
        case BLOCK_HEAD:
            if (bit_position() >= stop_bit) { *pos = (size_t)(op - out); return BOUNDARY; }
            if (!read_block_head()) return ERROR;
            if (bits_consumed_past_end()) return fail("truncated gzip file");
            break;
        case STORED: {
            while (stored_left_) {
                const size_t room = (size_t)(oend - op), have = (size_t)(in_end_ - ip_);
                if (!have) return fail("truncated gzip file");
                if (!room) { *pos = (size_t)(op - out); return NEED_OUTPUT; }
                size_t n = stored_left_;
                if (n > room) n = room;
                if (n > have) n = have;
                if (sizeof(T) == 1) memcpy(op, ip_, n);
                else for (size_t i = 0; i < n; i++) op[i] = (T)ip_[i];
                op += n; ip_ += n; stored_left_ -= (uint32_t)n;
            }
            state_ = last_block_ ? TRAILER : BLOCK_HEAD;
            break;
        }
        case HUFF: {
            if (oend - op < 320) { *pos = (size_t)(op - out); return NEED_OUTPUT; }
            T *const olimit = oend - 320;
            bool block_done = false;
            // the decoder state lives in locals inside the loop: the byte stores to `op` could alias the
            // members, which would force a reload of the bit buffer after every literal
            uint64_t bb = bitbuf_;
            int bc = bitcnt_;
            const uint8_t *ip = ip_;
            const uint8_t *const in_end = in_end_;
            const uint32_t *const lit = lit_, *const dst = dist_;
            const char *bad = nullptr;
#define TBK_REFILL()                                                                              \
            do {                                                                                  \
                uint64_t w_;                                                                      \
                if (ip + 8 <= in_end) memcpy(&w_, ip, 8);                                         \
                else { w_ = 0; for (int i_ = 0; i_ < 8 && ip + i_ < in_end; i_++) w_ |= (uint64_t)ip[i_] << (8 * i_); } \
                bb |= w_ << bc; ip += (63 - bc) >> 3; bc |= 56;                                   \
            } while (0)
#define TBK_LOOKUP(e)                                                                             \
            do {                                                                                  \
                e = lit[bb & LMASK];                                                              \
                if (((e >> 12) & 15u) == SUB) e = lit[(e >> 16) + ((bb >> LBITS) & ((1u << ((e >> 8) & 15u)) - 1u))]; \
            } while (0)
            while (op < olimit) {
                if (ip > in_end + 16) { bad = "truncated gzip file"; break; }
                TBK_REFILL();
                uint32_t e;
                TBK_LOOKUP(e);
                if ((e & 0x7000u) == 0) {
                    // up to three lookups per refill (3 x 15 bits <= 56), each worth one or two literals:
                    // both bytes are stored (the second is overwritten if there is only one)
#define TBK_EMIT()                                                                   \
                    do {                                                              \
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);                     \
                        const uint16_t v_ = (uint16_t)(e >> 16);                      \
                        if (sizeof(T) == 1) memcpy(op, &v_, 2);                       \
                        else { op[0] = (T)(v_ & 0xFFu); op[1] = (T)(v_ >> 8); }       \
                        op += 1 + ((e >> 15) & 1u);                                   \
                    } while (0)
                    TBK_EMIT();
                    TBK_LOOKUP(e);
                    if ((e & 0x7000u) == 0) {
                        TBK_EMIT();
                        TBK_LOOKUP(e);
                        if ((e & 0x7000u) == 0) {
                            TBK_EMIT();
                            continue;
                        }
                    }
#undef TBK_EMIT
                    // a length or end-of-block code follows: top the buffer up again first (the
                    // code's low bits are in the buffer already, so the lookup stands)
                    TBK_REFILL();
                    TBK_LOOKUP(e);
                }
                const uint32_t kind = (e >> 12) & 15u;
                if (kind == EOB) {
                    bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                    block_done = true;
                    break;
                }
                if (kind != LEN) { bad = "invalid literal/length code"; break; }
                bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                const int le = (int)((e >> 8) & 15u);
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1ull << le) - 1));
                bb >>= le; bc -= le;
                uint32_t d = dst[bb & DMASK];
                if (((d >> 12) & 15u) == SUB) d = dst[(d >> 16) + ((bb >> DBITS) & ((1u << ((d >> 8) & 15u)) - 1u))];
                if (((d >> 12) & 15u) != DIST) { bad = "invalid distance code"; break; }
                bb >>= (d & 0xFF); bc -= (int)(d & 0xFF);
                const int de = (int)((d >> 8) & 15u);
                const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1ull << de) - 1));
                bb >>= de; bc -= de;
                if ((size_t)(op - floor_) < dist) { bad = "distance too far back"; break; }
                const T *src = op - dist;
                T *const end = op + len;
                constexpr uint32_t PER = 8 / sizeof(T);  // elements per 8-byte copy
                if (dist >= PER) {
                    do { uint64_t w; memcpy(&w, src, 8); memcpy(op, &w, 8); op += PER; src += PER; } while (op < end);
                    op = end;
                } else if (dist == 1) {
                    const T v = *src;
                    if (sizeof(T) == 1) memset(op, (int)v, len);
                    else for (uint32_t i = 0; i < len; i++) op[i] = v;
                    op = end;
                } else {
                    while (op < end) *op++ = *src++;
                }
            }
#undef TBK_REFILL
#undef TBK_LOOKUP
            bitbuf_ = bb; bitcnt_ = bc; ip_ = ip;
            if (bad) return fail(bad);
            if (bits_consumed_past_end()) return fail("truncated gzip file");
            if (block_done) { state_ = last_block_ ? TRAILER : BLOCK_HEAD; break; }
            *pos = (size_t)(op - out);
            return NEED_OUTPUT;
        }
        case TRAILER: {
            byte_align_and_unread();
            if (in_end_ - ip_ < 8) return fail("truncated gzip file");
            t_crc_ = ip_[0] | ((uint32_t)ip_[1] << 8) | ((uint32_t)ip_[2] << 16) | ((uint32_t)ip_[3] << 24);
            t_isize_ = ip_[4] | ((uint32_t)ip_[5] << 8) | ((uint32_t)ip_[6] << 16) | ((uint32_t)ip_[7] << 24);
            ip_ += 8;
            state_ = HEADER;
            *pos = (size_t)(op - out);
            return MEMBER_DONE;
        }
        }
    }
}

template TbkInflate::Status TbkInflate::run_impl<uint8_t>(uint8_t *, size_t *, size_t, size_t, uint64_t);
template TbkInflate::Status TbkInflate::run_impl<uint16_t>(uint16_t *, size_t *, size_t, size_t, uint64_t);

// Does a dynamic-Huffman, non-final block begin at bit `bitpos`?  A cheap look at the first bits
// (type, code counts, a complete code-length code) before the tables are built for real; on success
// the decoder stands behind that block's header.
bool TbkInflate::open_dynamic_block_at(const uint8_t *data, size_t size, uint64_t bitpos) {
    const size_t byte = (size_t)(bitpos >> 3);
    if (byte + 16 > size) return false;
    uint64_t w0, w1;
    memcpy(&w0, data + byte, 8);
    memcpy(&w1, data + byte + 8, 8);
    const int sh = (int)(bitpos & 7);
    const unsigned __int128 v = (((unsigned __int128)w1 << 64) | w0) >> sh;  // 121+ bits from bitpos on; the header's fixed part takes 17 + 57 at most
    const uint32_t lo = (uint32_t)v;
    if ((lo & 7u) != 4u) return false;                     // BFINAL = 0, BTYPE = 2
    if (((lo >> 3) & 31u) > 29u || ((lo >> 8) & 31u) > 29u) return false;  // HLIT, HDIST
    const int n = (int)((lo >> 13) & 15u) + 4;
    uint32_t kraft = 0;
    for (int i = 0; i < n; i++) {
        const uint32_t len = (uint32_t)(v >> (17 + 3 * i)) & 7u;
        if (len) kraft += 128u >> len;
    }
    if (kraft != 128u) return false;
    reset(data, size);
    ip_ = in_ + byte;
    refill();
    take(sh);
    if (!read_block_head()) { err_ = nullptr; return false; }
    return state_ == HUFF && !last_block_;
}
