// tbk_inflate.h — a DEFLATE (RFC 1951) decoder for gzip members (RFC 1952), written for the one
// thing the FASTX reader needs: inflate a memory-mapped .gz file into a text window as fast as one
// core can.  A single gzip stream is one chain of dependencies (LineSource::pinflate_loop in tbk_fastx.cpp breaks it by
// guessing), and for FASTQ(.gz) input that stream is what both CLIs wait for; zlib 1.2.11 delivers ~0.55 GB/s of text here.  This decoder
// works the way the fast ones do: a 64-bit bit buffer refilled eight bytes at a time, one table
// lookup per symbol (11-bit primary table for literals/lengths, 8-bit for distances, sub-tables
// behind them for longer codes), word-wide match copies.
//
// Resumable at symbol boundaries: run() stops when the output window has no room for another match
// and continues from there on the next call, with the last 32 KiB of output still in front of the
// write position (the caller keeps them there).  CRC-32 and ISIZE of every member are checked by the
// caller from what run() reports, so a decoding mistake cannot pass silently.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

class TbkInflate {
public:
    enum Status { NEED_OUTPUT, MEMBER_DONE, INPUT_DONE, ERROR, BOUNDARY };

    void reset(const uint8_t *data, size_t size) {
        in_ = ip_ = data; in_end_ = data + size;
        bitbuf_ = 0; bitcnt_ = 0; state_ = HEADER; last_block_ = false; err_ = nullptr;
    }
    // Decode into out[*pos .. cap).  out[*pos - 32768 .. *pos) must hold the previous output of the
    // current member (as far as it exists).  member_start = *pos at which the current member began
    // (matches may not reach before it).
    Status run(uint8_t *out, size_t *pos, size_t cap, size_t member_start);
    // The same machine with 16-bit output elements: a literal is its byte value; a match copies
    // elements, whatever they are - out[*pos - 32768 .. *pos) holds the window, as byte values where it
    // is known and as markers of the caller's choosing where it is not (LineSource::pinflate_loop, tbk_fastx.cpp).  Returns
    // BOUNDARY in front of the first block header at or past bit `stop_bit` of the input.
    Status run16(uint16_t *out, size_t *pos, size_t cap, uint64_t stop_bit);
    // bits of the input consumed so far (exact between blocks and between symbols)
    uint64_t bit_position() const { return (uint64_t)(ip_ - in_) * 8 - (uint64_t)bitcnt_; }
    // Start decoding at bit `bitpos` if a non-final dynamic-Huffman block can begin there.
    bool open_dynamic_block_at(const uint8_t *data, size_t size, uint64_t bitpos);
    const char *error() const { return err_ ? err_ : ""; }
    // CRC-32 and ISIZE from the trailer of the member that just ended (after MEMBER_DONE)
    uint32_t trailer_crc() const { return t_crc_; }
    uint32_t trailer_isize() const { return t_isize_; }
    bool at_end() const { return state_ == HEADER && bits_consumed_past_end() == 0 && ip_ - (bitcnt_ >> 3) >= in_end_; }

private:
    enum State { HEADER, BLOCK_HEAD, STORED, HUFF, TRAILER };
    static constexpr int LBITS = 11, DBITS = 8;
    static constexpr int LSIZE = (1 << LBITS) + 1024, DSIZE = (1 << DBITS) + 512;
    // table entry: bits 0-7 code length, 8-11 extra bits (or sub-table bits), 12-15 kind, 16-31 value
    // LIT2: two literals decoded by one lookup (value = first | second << 8); the literal kinds are
    // the ones with bits 12-14 clear
    enum Kind { LIT = 0, LEN = 1, EOB = 2, SUB = 3, DIST = 4, LIT2 = 8, BAD = 15 };

    const uint8_t *in_ = nullptr, *ip_ = nullptr, *in_end_ = nullptr;
    uint64_t bitbuf_ = 0;
    int bitcnt_ = 0;
    State state_ = HEADER;
    bool last_block_ = false;
    uint32_t stored_left_ = 0;
    uint32_t t_crc_ = 0, t_isize_ = 0;
    const char *err_ = nullptr;
    uint32_t lit_[LSIZE], dist_[DSIZE];

    static uint32_t entry(uint32_t value, Kind kind, uint32_t extra, uint32_t nbits) {
        return (value << 16) | ((uint32_t)kind << 12) | (extra << 8) | nbits;
    }
    uint64_t load64(const uint8_t *p) const {
        uint64_t v = 0;
        if (p + 8 <= in_end_) { memcpy(&v, p, 8); return v; }
        for (int i = 0; i < 8 && p + i < in_end_; i++) v |= (uint64_t)p[i] << (8 * i);
        return v;  // past the end reads as zero bits; bits_consumed_past_end() tells
    }
    void refill() {
        bitbuf_ |= load64(ip_) << bitcnt_;
        ip_ += (63 - bitcnt_) >> 3;
        bitcnt_ |= 56;
    }
    uint32_t take(int n) {  // n <= 32, after a refill
        const uint32_t v = (uint32_t)(bitbuf_ & ((1ull << n) - 1));
        bitbuf_ >>= n; bitcnt_ -= n;
        return v;
    }
    long bits_consumed_past_end() const {
        const long consumed_bytes_x8 = (long)(ip_ - in_) * 8 - bitcnt_;
        const long have = (long)(in_end_ - in_) * 8;
        return consumed_bytes_x8 > have ? consumed_bytes_x8 - have : 0;
    }
    void byte_align_and_unread() {  // drop the bits up to the next byte boundary, give whole bytes back
        const int drop = bitcnt_ & 7;
        bitbuf_ >>= drop; bitcnt_ -= drop;
        ip_ -= bitcnt_ >> 3;
        bitbuf_ = 0; bitcnt_ = 0;
    }
    Status fail(const char *msg) { err_ = msg; return ERROR; }
    template <class T> Status run_impl(T *out, size_t *pos, size_t cap, size_t member_start, uint64_t stop_bit);
    bool parse_header();
    bool read_block_head();
    bool build(const uint8_t *lens, int n, uint32_t *table, int table_size, int primary_bits, bool is_dist, bool is_codes = false);
    bool dynamic_tables();
    void fixed_tables();
    void pair_literals();
};
