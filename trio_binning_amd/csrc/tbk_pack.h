// tbk_pack.h - the packed transfer format's host-side packer, for the other translation units of the library
// (tbk_pack.cpp has the format; include/tbk.h the public entry points).
#pragma once
#include <stdint.h>

#include <vector>

struct TbkExc { uint32_t chunk; uint16_t mask; };  // a chunk holding a byte outside ACGT (or positions past the end)

// whole stream, all host threads (threads <= 0: tbk_host_threads())
int tbk_pack_bases_vec(const uint8_t *bases, uint64_t total, uint32_t *codes, std::vector<uint32_t> &exc_chunk, std::vector<uint16_t> &exc_mask, int threads);
// full chunks [c_lo, c_hi) on the calling thread; the stream's last, partial chunk
void tbk_pack_chunk_range_(const uint8_t *bases, uint64_t c_lo, uint64_t c_hi, uint32_t *codes, std::vector<TbkExc> &exc);
void tbk_pack_tail_chunk_(const uint8_t *bases, uint64_t total, uint32_t *codes, std::vector<TbkExc> &exc);
// the same for bytes that lie outside a stream buffer: `n` full chunks at `src` = chunks [c_lo, c_lo + n); the last,
// partial chunk from its `valid` bytes
void tbk_pack_span_(const uint8_t *src, uint64_t c_lo, uint64_t n, uint32_t *codes, std::vector<TbkExc> &exc);
void tbk_pack_tail_bytes_(const uint8_t *bytes, unsigned valid, uint64_t chunk, uint32_t *codes, std::vector<TbkExc> &exc);
