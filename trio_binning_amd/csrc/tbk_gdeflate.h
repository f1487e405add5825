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
