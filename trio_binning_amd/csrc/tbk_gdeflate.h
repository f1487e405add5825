// tbk_gdeflate.h — the GPU gzip encoder behind the bin writer (tbk_gdeflate.hip): members in, gzip members out, three jobs deep.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

struct tbk_gdeflate;
struct tbk_gdeflate_member { const char *src; size_t n; int tag; };   // n bytes of text (host memory, ideally pinned); tag: the caller's (the bin)
struct tbk_gdeflate_out { int tag; const char *data; size_t n; };      // one finished gzip member

int tbk_gdeflate_create(int device, tbk_gdeflate **out);
void tbk_gdeflate_destroy(tbk_gdeflate *g);
// Queue a job: the members' text is copied to the device and coded, asynchronously.  The caller then sums each member's CRC-32
// (tbk_gdeflate_set_crc, any time before the job is collected; the device sums them too) and waits for tbk_gdeflate_text_done before it touches the text again.
int tbk_gdeflate_submit(tbk_gdeflate *g, const tbk_gdeflate_member *members, size_t n_members);
void tbk_gdeflate_set_crc(tbk_gdeflate *g, size_t member, uint32_t crc);
int tbk_gdeflate_text_done(tbk_gdeflate *g, int back = 0);   // back: how many jobs before the newest
// Move the pipeline on and take the oldest finished job's members (none when nothing is ready and !drain); the bytes stay valid
// until the call after the next one.
int tbk_gdeflate_collect(tbk_gdeflate *g, bool drain, std::vector<tbk_gdeflate_out> &out);
int tbk_gdeflate_in_flight(const tbk_gdeflate *g);
void tbk_gdeflate_stats(const tbk_gdeflate *g, uint64_t *text_bytes, uint64_t *member_bytes, uint64_t *blocks, uint64_t *members);

// ---- the other direction: bgzf blocks inflated on the device (tbk_gdeflate.hip, second half) -------------------------------------------
struct tbk_ginflate;
struct tbk_ginflate_block { uint64_t in_off; uint32_t in_len, out_len, crc; uint32_t pad_; };   // a raw deflate stream in the window's input; its text's length and CRC-32 (the bgzf trailer's)
constexpr int TBK_GINFLATE_SLOTS = 4;   // one being staged, one on the device, one with the parser, one more (measured at configs[1] scale: 8 -> 3.9, 5 -> 4.2, 4 -> 4.4 Gbases/s: their pinned memory costs more to get and to give back than lending it to the batches saves)
int tbk_ginflate_create(int device, tbk_ginflate **out);
void tbk_ginflate_destroy(tbk_ginflate *g);
uint8_t *tbk_ginflate_input(tbk_ginflate *g, int slot, size_t bytes);
int tbk_ginflate_reserve(tbk_ginflate *g, int slot, size_t in_bytes, size_t n_blocks, size_t out_bytes);
int tbk_ginflate_submit(tbk_ginflate *g, int slot, size_t in_bytes, const tbk_ginflate_block *blocks, size_t n_blocks, size_t head);
int tbk_ginflate_wait(tbk_ginflate *g, int slot, uint8_t **out_base, size_t *text_bytes, uint32_t *bad);
