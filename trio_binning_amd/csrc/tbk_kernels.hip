// tbk_kernels.hip — the MI355X (gfx950, wave64) kernels of the classify-by-kmers path.
//
//   tbk_probe_kernel   replaces the per-window loop of count_kmers_in_read
//                      (c/kmers.c:270-299) and both kmer_in_hash_set probes
//                      (c/kmers.c:245-268) for a whole batch of reads.
//   tbk_insert_kernel, tbk_order_kernel  replace add_to_hash (c/kmers.c:112-122).
//   tbk_contains_kernel  raw-key membership (tests).
//
// Integer/hash work: no MFMA.  The bound is HBM random-line throughput.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "tbk_common.h"
#include "tbk_device.h"

// =======================================================================================
// insert
// =======================================================================================
// Membership of one key, one thread, whole lines (build-time scans and tests; not the hot path).
__device__ __forceinline__ bool tbk_lookup_slow(const TbkTableView t, uint64_t key) {
    if (key >= TBK_NOKEY) return false;
    uint32_t b = tbk_bucket_of(key, t.mz, t.n_buckets);
    for (uint32_t walked = 0; walked <= t.n_buckets; walked++) {
        const uint64_t *line = t.slots + (uint64_t)b * t.stride;
        for (uint32_t s = 0; s < TBK_SLOTS_PER_BUCKET; s++)
            if (line[tbk_slot_at(t.guests, t.stride, t.half, s)] == key) return true;
        if (t.stride == 16 && (t.guests & (TBK_FLAG_FRONT | TBK_FLAG_GUESTS)) == (TBK_FLAG_FRONT | TBK_FLAG_GUESTS))
            for (uint32_t s = 0; s < 4; s++)  // front layout: a key may sit, tagged, in the other list's front of any line on its way
                if (line[tbk_slot_at(t.guests, t.stride, t.half ^ 8u, s)] == (key | TBK_GUEST)) return true;
        if (!(line[tbk_slot_at(t.guests, t.stride, t.half, 6)] > line[tbk_slot_at(t.guests, t.stride, t.half, 7)])) return false;  // no key went past this half
        if (t.guests & TBK_FLAG_GUESTS) {
            for (uint32_t s = 0; s < TBK_SLOTS_PER_BUCKET; s++)
                if (line[tbk_slot_at(t.guests, t.stride, t.half ^ 8u, s)] == (key | TBK_GUEST)) return true;
            if (!(line[tbk_slot_at(t.guests, t.stride, t.half, 4)] > line[tbk_slot_at(t.guests, t.stride, t.half, 5)])) return false;  // keys went past the half, none left the line
        }
        b = tbk_next_bucket(key, t.mz, t.n_buckets, b, walked == 0);
    }
    return false;
}

// One key per thread.  Scan this list's 8 slots of the home bucket; claim the first free
// slot with a 64-bit CAS; a bucket half without a free slot sends the key to the next
// bucket.  Duplicates are detected (the reference stores them twice, c/kmers.c:112-122;
// membership is the same).  `stride`/`half` select a standalone table (8, 0) or the hapA /
// hapB half of a paired table (16, 0 / 8).
__global__ void __launch_bounds__(256)
tbk_insert_kernel(uint64_t *__restrict__ slots, uint32_t n_buckets, uint32_t stride, uint32_t half, TbkMz mz,
                  const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ overflowed,
                  uint32_t *__restrict__ left_line, uint32_t guests, TbkTableView skip,
                  unsigned long long *__restrict__ n_distinct, unsigned long long *__restrict__ n_skipped,
                  unsigned long long *__restrict__ n_past, int *__restrict__ failed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t halves = stride / TBK_SLOTS_PER_BUCKET, which = half / TBK_SLOTS_PER_BUCKET;
    const bool front_guests = stride == 16 && (guests & (TBK_FLAG_FRONT | TBK_FLAG_GUESTS)) == (TBK_FLAG_FRONT | TBK_FLAG_GUESTS);
    unsigned long long mine = 0, skipped = 0, past = 0, back = 0;  // past: keys that found their own half of their home line full; back: keys behind the first four slots of it
    for (; i < n; i += step) {
        const uint64_t key = keys[i];
        if (key >= TBK_NOKEY) continue;  // TBK_EMPTY / TBK_NOKEY: never a canonical key, never stored
        // hapB's list going into a paired table: a key that hapA's (finished) half already holds can
        // never count for hapB (hapA is asked first, c/kmers.c:291-294), so it is not stored and the
        // two halves stay disjoint - the probe kernel never has to arbitrate between them
        if (skip.slots != nullptr && tbk_lookup_slow(skip, key)) { skipped++; continue; }
        // the buckets a lookup of this key may select: one, except when mod-sampling finds the
        // smallest t-mer rank at several positions of the key - then the key goes into the bucket of
        // every tied position (two positions naming the same bucket: the second visit finds the key there).
        // No candidate array: positions are walked twice (arrays indexed at run time live in scratch memory).
        const bool tied_positions = mz.w > 0 && mz.t > 0;
        const int n_pos = tied_positions ? tbk_mz_positions(mz) : 1;
        uint32_t best = 0xFFFFFFFFu;
        if (tied_positions)
            for (int pi = 0; pi < n_pos; pi++) { const uint32_t g = tbk_tmer_rank(key, mz, pi); best = g < best ? g : best; }
        int c = -1;  // copies placed so far - 1: only the first one counts as a stored key
        for (int pi = 0; pi < n_pos; pi++) {
            uint32_t b;
            if (tied_positions) {
                if (tbk_tmer_rank(key, mz, pi) != best) continue;
                b = tbk_bucket_at(key, mz, pi % mz.w, n_buckets);
            } else {
                b = tbk_bucket_of(key, mz, n_buckets);
            }
            c++;
            bool done = false;
            for (uint32_t walked = 0; walked <= n_buckets && !done; walked++) {
                unsigned long long *line = (unsigned long long *)(slots + (uint64_t)b * stride);
                for (uint32_t s = 0; s < TBK_SLOTS_PER_BUCKET && !done; s++) {
                    if (s == 4 && front_guests) {
                        // front layout: the list's four front slots are taken - before the key goes behind the
                        // front it may sit, tagged, in a free front slot of the other list (a window then still
                        // finds it in the 64 bytes it fetches)
                        const unsigned long long tagged = key | TBK_GUEST;
                        for (uint32_t g = 0; g < 4 && !done; g++) {
                            unsigned long long *slot = &line[tbk_slot_at(guests, stride, half ^ 8u, g)];
                            unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (cur == tagged) { done = true; break; }
                            if (cur == TBK_EMPTY) {
                                unsigned long long old = atomicCAS(slot, (unsigned long long)TBK_EMPTY, tagged);
                                if (old == TBK_EMPTY) { if (c == 0) mine++; done = true; }
                                else if (old == tagged) { done = true; }
                            }
                        }
                        if (done) break;
                    }
                    unsigned long long *slot = &line[tbk_slot_at(guests, stride, half, s)];
                    unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur == key) { done = true; break; }
                    if (cur == TBK_EMPTY) {
                        unsigned long long old = atomicCAS(slot, (unsigned long long)TBK_EMPTY, (unsigned long long)key);
                        if (old == TBK_EMPTY) { if (c == 0) { mine++; back += (s >= 4 && walked == 0); } done = true; }
                        else if (old == key) { done = true; }
                        // else: somebody else's key took the slot; keep scanning
                    }
                }
                if (!done) {
                    // this half is full and the key goes past it: remember that (tbk_order_kernel turns
                    // the note into the order of the half's last two slots)
                    const uint64_t bit = (uint64_t)b * halves + which;
                    atomicOr(&overflowed[bit >> 5], 1u << (bit & 31));
                    past += (c == 0 && walked == 0);
                    if (guests & TBK_FLAG_GUESTS) {
                        // ... first into a free slot of the other list's half of the same line, tagged: lookups
                        // hold the whole line, so a guest costs them two compares, not another random line
                        const unsigned long long tagged = key | TBK_GUEST;
                        for (uint32_t s = 0; s < TBK_SLOTS_PER_BUCKET && !done; s++) {
                            unsigned long long *slot = &line[tbk_slot_at(guests, stride, half ^ 8u, s)];
                            unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (cur == tagged) { done = true; break; }
                            if (cur == TBK_EMPTY) {
                                unsigned long long old = atomicCAS(slot, (unsigned long long)TBK_EMPTY, tagged);
                                if (old == TBK_EMPTY) { if (c == 0) mine++; done = true; }
                                else if (old == tagged) { done = true; }
                            }
                        }
                        if (!done) atomicOr(&left_line[bit >> 5], 1u << (bit & 31));  // ... or out of the line (order of slots 4 and 5)
                    }
                    // then along the probe sequence: home bucket, second-choice bucket, and linearly on from there
                    if (!done) b = tbk_next_bucket(key, mz, n_buckets, b, walked == 0);
                }
            }
            if (!done) atomicExch(failed, 1);
        }
    }
    if (mine) atomicAdd(n_distinct, mine);
    if (skipped) atomicAdd(n_skipped, skipped);
    if (past) atomicAdd(n_past, past);
    if (back) atomicAdd(n_past + 1, back);
}

// Entry layout (tbk_common.h): one list key per thread, stored through each of its forms - one per position that
// attains the smallest t-mer rank (as in tbk_insert_kernel), two where the sampled m-mer is its own reverse complement.
// A form joins the first entry of its list, along its m-mer's bucket sequence, whose flanks agree with its own
// (a 64-bit CAS that adds the form's flank bits and its V bit), or takes the first empty slot of the line (both lists fill
// a line's 16 slots in order; half = 0: hapA's list, else hapB's - bit 62 of the slot).  Entries only ever gain bits, so
// "compatible" can only turn into "incompatible" while a thread looks, never back: a key is stored exactly once per form
// however the threads interleave.  Bit 63 of slot 3 / of slot 15 is set when an entry is created behind the front / when
// a form leaves the line.  List lines that are not canonical are dead in the reference (stored verbatim, c/kmers.c:113;
// looked up as min(fwd, rc), c/kmers.c:255) and are not stored.
// cnt: [0] keys stored, [1] hapB keys left out because hapA holds them, [2] entries created, [3] of those behind a
// front, [4] forms that left a line.
__global__ void __launch_bounds__(256)
tbk_entry_insert_kernel(uint64_t *__restrict__ slots, uint32_t n_buckets, uint32_t half, TbkMz mz, TbkEntryGeom g, int k,
                        const uint64_t *__restrict__ keys, uint64_t n, int skip_a, unsigned long long *__restrict__ cnt, int *__restrict__ failed) {
    // Every thread takes a contiguous stretch of the list: a list made by find-unique-kmers holds the windows of a variant
    // side by side, and neighbouring threads working on neighbouring lines would all want the same few entries at once.
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_threads = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t per = (n + n_threads - 1) / n_threads;
    const uint64_t i_end = (tid + 1) * per < n ? (tid + 1) * per : n;
    const unsigned long long FLAG64 = (unsigned long long)TBK_ENTRY_FLAG << 32;
    unsigned long long stored = 0, skipped = 0, created = 0, behind = 0, past = 0;
    const int n_pos = tbk_mz_positions(mz);
    for (uint64_t i = tid * per; i < i_end; i++) {
        const uint64_t key = keys[i];
        if (key >= TBK_NOKEY) continue;
        if (tbk_revcomp_packed(key, k) < key) continue;  // not canonical: never looked up
        uint32_t best = 0xFFFFFFFFu;
        for (int pi = 0; pi < n_pos; pi++) { const uint32_t r = tbk_tmer_rank(key, mz, pi); best = r < best ? r : best; }
        bool first_form = true, drop = false;
        for (int pi = 0; pi < n_pos && !drop; pi++) {
            if (tbk_tmer_rank(key, mz, pi) != best) continue;
            TbkEntryKey forms[2];
            const int nf = tbk_entry_orientations(key, k, mz, g, pi % mz.w, forms);
            for (int f = 0; f < nf; f++) {
                const TbkEntryKey e = forms[f];
                if (first_form && skip_a && tbk_entry_lookup_one(slots, n_buckets, e) == 0) { skipped++; drop = true; break; }  // (hapA's inserts are finished)
                const uint32_t hapb = half ? 1u : 0u;
                const unsigned long long entry = (unsigned long long)e.cm | ((unsigned long long)(e.khi | (hapb ? TBK_ENTRY_HAPB : 0u)) << 32);
                uint32_t b = tbk_entry_bucket(e.cm, n_buckets);
                bool done = false;
                for (uint32_t walked = 0; walked <= n_buckets && !done; walked++) {
                    unsigned long long *line = (unsigned long long *)(slots + (uint64_t)b * 16);
                    for (uint32_t sl = 0; sl < 16 && !done; sl++) {
                        unsigned long long *slot = &line[sl];
                        unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (;;) {
                            if ((cur & ~FLAG64) == 0) {  // empty: mine, unless somebody is quicker
                                const unsigned long long old = atomicCAS(slot, cur, cur | entry);
                                if (old == cur) {
                                    created++; stored += first_form; done = true;
                                    if (sl >= 4) { behind++; atomicOr(&line[3], FLAG64); }
                                    break;
                                }
                                cur = old;
                                continue;
                            }
                            if (!tbk_entry_compatible(cur, e, hapb, mz, g)) break;  // the other list's, or flanks that disagree (and stay so: entries only gain bits) - next slot
                            if (tbk_entry_match(cur, e)) { done = true; break; }  // a duplicate line, or this key's other tied position naming the same m-mer and place
                            const unsigned long long old = atomicCAS(slot, cur, cur | ((unsigned long long)e.khi << 32));
                            if (old == cur) { stored += first_form; done = true; break; }
                            cur = old;
                        }
                    }
                    if (!done) {
                        atomicOr(&line[15], FLAG64);
                        past++;
                        b = tbk_entry_next_bucket(e.cm, n_buckets, b, walked == 0);
                    }
                }
                if (!done) atomicExch(failed, 1);
                first_form = false;
            }
        }
    }
    if (stored) atomicAdd(&cnt[0], stored);
    if (skipped) atomicAdd(&cnt[1], skipped);
    if (created) atomicAdd(&cnt[2], created);
    if (behind) atomicAdd(&cnt[3], behind);
    if (past) atomicAdd(&cnt[4], past);
}

// The same for wide entries (tbk_common.h "wide entries").  An entry is two words and there is no 128-bit compare-and-swap:
// a form takes the piece's lock (bit 63 of word 0, by CAS on that word) to look at word 1, and ORs its flank bits in when
// list and flanks agree with the entry's.  Every change of word 1 is an OR - the flags other threads set in it meanwhile
// are never lost - and a lock is held for a handful of instructions inside ONE trip of the loop: the lanes of a wave that
// wait for it wait for a lane that is not waiting for them.  Finished tables hold no lock.  Both lists fill a line's eight
// pieces in order (half = 0: hapA's list, else hapB's: bit 62 of word 1).
__global__ void __launch_bounds__(256)
tbk_wentry_insert_kernel(uint64_t *__restrict__ slots, uint32_t n_buckets, uint32_t half, TbkMz mz, TbkEntryGeom g, int k,
                         const uint64_t *__restrict__ keys, uint64_t n, int skip_a, unsigned long long *__restrict__ cnt, int *__restrict__ failed) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_threads = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t per = (n + n_threads - 1) / n_threads;  // (a contiguous stretch of the list per thread: see tbk_entry_insert_kernel)
    const uint64_t i_end = (tid + 1) * per < n ? (tid + 1) * per : n;
    const uint32_t hapb = half ? 1u : 0u;
    unsigned long long stored = 0, skipped = 0, created = 0, behind = 0, past = 0;
    const int n_pos = tbk_mz_positions(mz);
    for (uint64_t i = tid * per; i < i_end; i++) {
        const uint64_t key = keys[i];
        if (key >= TBK_NOKEY) continue;
        if (tbk_revcomp_packed(key, k) < key) continue;  // not canonical: never looked up
        uint32_t best = 0xFFFFFFFFu;
        for (int pi = 0; pi < n_pos; pi++) { const uint32_t r = tbk_tmer_rank(key, mz, pi); best = r < best ? r : best; }
        bool first_form = true, drop = false;
        for (int pi = 0; pi < n_pos && !drop; pi++) {
            if (tbk_tmer_rank(key, mz, pi) != best) continue;
            TbkWideKey forms[2];
            const int nf = tbk_wentry_orientations(key, k, mz, g, pi % mz.w, forms);
            for (int f = 0; f < nf; f++) {
                const TbkWideKey e = forms[f];
                if (first_form && skip_a && tbk_wentry_lookup_one(slots, n_buckets, e) == 0) { skipped++; drop = true; break; }  // (hapA's inserts are finished)
                const unsigned long long taken = (unsigned long long)e.cm | TBK_WENTRY_TAKEN;
                const unsigned long long mine1 = (unsigned long long)e.k1 | (hapb ? TBK_WENTRY_HAPB : 0ull);
                uint32_t b = tbk_wentry_bucket(e.cm, n_buckets);
                bool done = false;
                for (uint32_t walked = 0; walked <= n_buckets && !done; walked++) {
                    unsigned long long *line = (unsigned long long *)(slots + (uint64_t)b * 16);
                    for (uint32_t en = 0; en < 8 && !done; en++) {
                        unsigned long long *w0 = &line[2 * en], *w1 = w0 + 1;
                        bool next_piece = false;
                        while (!done && !next_piece) {
                            // Relaxed device-scope atomics only: every access to the two words is an atomic performed where the XCDs
                            // meet, and the order that matters - my OR of word 1 before my unlock of word 0 - is kept by waiting for the
                            // OR's return value before the unlock is issued.  Acquire / release at agent scope would make every
                            // insert invalidate or write back an XCD's whole L2: 2e9 keys took 25 s that way.
                            // What this rests on is the hardware, not the memory model: device-scope atomics on gfx942 / gfx950 are
                            // performed at the memory side (one point of coherence for all XCDs; MI355X_MICROARCH.md, Global float
                            // atomics), in the order they arrive there, and a returned atomic has been performed.  What checks it:
                            // every key of both lists is looked up again after a build - tbk_classifier_verify (the CLI's
                            // TBK_VERIFY_BUILD=1) and, at 2 x 1e9 keys, the full-membership sweep of tests/test_gpu_scale.py.
                            const unsigned long long cur = __hip_atomic_load(w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (cur & TBK_WENTRY_LOCK) continue;                                   // somebody is looking at this piece: again
                            if (cur != 0 && cur != taken) { next_piece = true; break; }           // another m-mer's entry
                            // empty, or an entry of my m-mer: take the lock (an empty piece becomes mine with it)
                            if (atomicCAS(w0, cur, (cur == 0 ? taken : cur) | TBK_WENTRY_LOCK) != cur) continue;
                            const unsigned long long v1 = __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            unsigned long long seen = v1;  // what word 1 held when my change (if any) went in
                            if (cur == 0) {
                                seen = atomicOr(w1, mine1);
                                created++; stored += first_form; done = true;
                            } else if (tbk_wentry_compatible(cur, v1, e, hapb, mz, g)) {
                                if (!tbk_wentry_match(cur, v1, e)) { seen = atomicOr(w1, (unsigned long long)e.k1); stored += first_form; }
                                done = true;
                            } else {
                                next_piece = true;  // the other list's entry, or flanks that disagree (and stay so: entries only gain bits)
                            }
                            // unlock - behind the OR: the empty asm needs the OR's return value in a register, so the wave has waited for it,
                            // and its memory clobber keeps the store below
                            asm volatile("" ::"v"((uint32_t)seen), "v"((uint32_t)(seen >> 32)) : "memory");
                            __hip_atomic_store(w0, cur == 0 ? taken : cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (done && cur == 0 && en >= 2) { behind++; atomicOr(&line[2 * 1 + 1], (unsigned long long)TBK_WENTRY_FLAG); }
                        }
                    }
                    if (!done) {
                        atomicOr(&line[2 * 7 + 1], (unsigned long long)TBK_WENTRY_FLAG);
                        past++;
                        b = tbk_wentry_next_bucket(e.cm, n_buckets, b, walked == 0);
                    }
                }
                if (!done) atomicExch(failed, 1);
                first_form = false;
            }
        }
    }
    if (stored) atomicAdd(&cnt[0], stored);
    if (skipped) atomicAdd(&cnt[1], skipped);
    if (created) atomicAdd(&cnt[2], created);
    if (behind) atomicAdd(&cnt[3], behind);
    if (past) atomicAdd(&cnt[4], past);
}

// Short keys (tbk_common.h "short keys"): a list key is stored through each of its forms as ONE 32-bit word in the first free
// slot of its m-mer's line (32 slots both lists fill in order; half != 0: hapB's list, bit 30 of the word); a form that
// finds all 32 taken puts the canonical key into the overflow table behind the lines (open addressing, 64-bit CAS) and flags
// the line's last slot.  hapB's keys that hapA holds are left out (skip_a: hapA's inserts are finished).
// cnt: [0] keys stored, [1] hapB keys left out, [2] words created, [3] of those behind a front, [4] forms in the overflow table.
__global__ void __launch_bounds__(256)
tbk_short_insert_kernel(uint32_t *__restrict__ lines, uint32_t n_buckets, unsigned long long *__restrict__ over, uint32_t over_mask, uint32_t half, TbkMz mz,
                        TbkShortGeom g, int k, const uint64_t *__restrict__ keys, uint64_t n, int skip_a, unsigned long long *__restrict__ cnt,
                        int *__restrict__ failed, uint32_t line_cap) {  // line_cap: slots of a line the inserts use (32; tests lower it to fill the overflow table)
    unsigned long long stored = 0, skipped = 0, created = 0, behind = 0, past = 0;
    const int n_pos = tbk_mz_positions(mz);
    const uint32_t listbit = half ? TBK_SHORT_HAPB : 0u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        if (key >= TBK_NOKEY) continue;
        if (tbk_revcomp_packed(key, k) < key) continue;  // not canonical: never looked up
        uint32_t best = 0xFFFFFFFFu;
        for (int pi = 0; pi < n_pos; pi++) { const uint32_t r = tbk_tmer_rank(key, mz, pi); best = r < best ? r : best; }
        bool first_form = true, drop = false;
        for (int pi = 0; pi < n_pos && !drop; pi++) {
            if (tbk_tmer_rank(key, mz, pi) != best) continue;
            TbkShortKey forms[2];
            const int nf = tbk_short_orientations(key, k, mz, g, pi % mz.w, n_buckets, forms);
            for (int f = 0; f < nf; f++) {
                const TbkShortKey e = forms[f];
                if (first_form && skip_a && tbk_short_lookup_one(lines, n_buckets, (const uint64_t *)over, over_mask, e, key) == 0) { skipped++; drop = true; break; }
                uint32_t *line = lines + (uint64_t)e.bucket * 32;
                bool done = false;
                for (uint32_t sl = 0; sl < line_cap && !done; sl++) {
                    if (sl == TBK_SHORT_SUMMARY) continue;  // (no key: the line's summary of what lies behind its front)
                    uint32_t cur = __hip_atomic_load(&line[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (;;) {
                        if ((cur & ~TBK_SHORT_FLAG) == 0) {  // empty: mine, unless somebody is quicker
                            const uint32_t old = atomicCAS(&line[sl], cur, cur | e.word | listbit);
                            if (old == cur) {
                                created++; stored += first_form; done = true;
                                if (sl > TBK_SHORT_SUMMARY) { behind++; atomicOr(&line[TBK_SHORT_SUMMARY], TBK_SHORT_FLAG | tbk_short_filter_bit(e.word)); }
                                break;
                            }
                            cur = old;
                            continue;
                        }
                        if ((cur & ~(TBK_SHORT_FLAG | TBK_SHORT_HAPB)) == e.word) done = true;  // a duplicate line, or this key's other tied position naming the same m-mer and place
                        break;
                    }
                }
                if (!done) {
                    // the line is full: the canonical key goes to the overflow table
                    atomicOr(&line[31], TBK_SHORT_FLAG);
                    atomicOr(&line[TBK_SHORT_SUMMARY], TBK_SHORT_FLAG | tbk_short_filter_bit(e.word));  // (a window must get as far as the overflow table)
                    const unsigned long long mine = (unsigned long long)key | ((unsigned long long)(half ? 1u : 0u) << 63);
                    uint32_t at = tbk_short_over_home(key, over_mask);
                    for (uint32_t walked = 0; walked <= over_mask && over_mask != 0 && !done; walked++, at = (at + 1) & over_mask) {
                        unsigned long long cur = __hip_atomic_load(&over[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (cur == TBK_SHORT_EMPTY64) {
                            const unsigned long long old = atomicCAS(&over[at], cur, mine);
                            if (old == cur) { past++; stored += first_form; done = true; break; }
                            cur = old;
                        }
                        if ((cur & ~(1ull << 63)) == key) done = true;  // there already (another form of this key, or a duplicate line)
                    }
                }
                if (!done) atomicExch(failed, 1);
                first_form = false;
            }
        }
    }
    if (stored) atomicAdd(&cnt[0], stored);
    if (skipped) atomicAdd(&cnt[1], skipped);
    if (created) atomicAdd(&cnt[2], created);
    if (behind) atomicAdd(&cnt[3], behind);
    if (past) atomicAdd(&cnt[4], past);
}

// Full keys (tbk_common.h "full keys"): one list key per thread, stored - inverted, with its list's bit - in the first free
// slot of the line of every position that attains the smallest t-mer rank (slots 0, 1, 2, then 4 .. 15: slot 3 is the line's
// summary); a key placed behind the front, or sent on past a full line, sets its bit in the summary of the line it came
// through (a window follows the same path: home line, second-choice bucket, linear).  hapB's keys that hapA holds are left
// out (skip_a: hapA's inserts are finished).
// cnt: [0] keys stored, [1] hapB keys left out, [2] slots taken, [3] of those behind a front, [4] forms that left a line.
__global__ void __launch_bounds__(256)
tbk_full_insert_kernel(uint64_t *__restrict__ slots, uint32_t n_buckets, uint32_t half, TbkMz mz, int k, const uint64_t *__restrict__ keys, uint64_t n, int skip_a,
                       unsigned long long *__restrict__ cnt, int *__restrict__ failed, uint32_t only_below) {
    // only_below != 0: a SAMPLE BY BUCKET - every key is hashed as for n_buckets lines, but only the forms whose home bucket is
    // below only_below are stored (the table has that many lines), and a form that finds its line full is counted, not sent on.
    // Whatever order the list is in, the sampled lines fill exactly as they would in the whole table: what the policy asks to
    // know before it builds it (do the lists' keys crowd their buckets?) for a sixteenth of the random writes.
    unsigned long long stored = 0, skipped = 0, created = 0, behind = 0, past = 0;
    const int n_pos = tbk_mz_positions(mz);
    const unsigned long long listbit = half ? TBK_FULL_HAPB : 0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        if (key >= TBK_FULL_NOKEY) continue;             // cannot be canonical
        if (tbk_revcomp_packed(key, k) < key) continue;  // not canonical: never looked up
        const unsigned long long word = tbk_full_word(key), fbit = tbk_full_filter_bit(word);
        uint32_t best = 0xFFFFFFFFu;
        for (int pi = 0; pi < n_pos; pi++) { const uint32_t r = tbk_tmer_rank(key, mz, pi); best = r < best ? r : best; }
        bool first_form = true, drop = false;
        for (int pi = 0; pi < n_pos && !drop; pi++) {
            if (tbk_tmer_rank(key, mz, pi) != best) continue;
            uint32_t b = tbk_wentry_bucket(tbk_full_mmer(key, mz, pi % mz.w), n_buckets);
            if (only_below && b >= only_below) continue;
            if (first_form && skip_a && tbk_full_lookup_one(slots, n_buckets, key, b, mz) == 0) { skipped++; drop = true; break; }
            bool done = false;
            for (uint32_t walked = 0; walked <= n_buckets && !done; walked++) {
                unsigned long long *line = (unsigned long long *)(slots + (uint64_t)b * 16);
                for (uint32_t sl = 0; sl < 16 && !done; sl++) {
                    if (sl == TBK_FULL_SUMMARY) continue;
                    unsigned long long cur = __hip_atomic_load(&line[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (;;) {
                        if ((cur & ~TBK_FULL_FLAG) == 0) {  // empty: mine, unless somebody is quicker
                            const unsigned long long old = atomicCAS(&line[sl], cur, cur | word | listbit);
                            if (old == cur) {
                                created++; stored += first_form; done = true;
                                if (sl > TBK_FULL_SUMMARY) { behind++; atomicOr(&line[TBK_FULL_SUMMARY], TBK_FULL_FLAG | fbit); }
                                break;
                            }
                            cur = old;
                            continue;
                        }
                        if ((cur & TBK_FULL_KEY) == word) done = true;  // a duplicate line, or this key's other tied position naming the same bucket
                        break;
                    }
                }
                if (!done) {
                    // the line is full: the key goes on, and a window that follows it must get past this line's front
                    atomicOr(&line[15], TBK_FULL_FLAG);
                    atomicOr(&line[TBK_FULL_SUMMARY], TBK_FULL_FLAG | fbit);
                    past++;
                    if (only_below) { done = true; break; }
                    b = tbk_next_bucket(key, mz, n_buckets, b, walked == 0);
                }
            }
            if (!done) atomicExch(failed, 1);
            first_form = false;
        }
    }
    if (stored) atomicAdd(&cnt[0], stored);
    if (skipped) atomicAdd(&cnt[1], skipped);
    if (created) atomicAdd(&cnt[2], created);
    if (behind) atomicAdd(&cnt[3], behind);
    if (past) atomicAdd(&cnt[4], past);
}

// After all inserts: give every full half the order of its last two slots that says whether a key
// went past it (slot 6 > slot 7) or not (slot 6 < slot 7), and - tables with guests - the order of
// slots 4 and 5 that says whether one of those keys left the line (slot 4 > slot 5).  One thread per half.
__global__ void __launch_bounds__(256)
tbk_order_kernel(uint64_t *__restrict__ slots, uint64_t n_halves, const uint32_t *__restrict__ overflowed,
                 const uint32_t *__restrict__ left_line, uint32_t flags, uint32_t stride) {
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_halves) return;
    const uint32_t halves = stride / TBK_SLOTS_PER_BUCKET;
    uint64_t *line = slots + (h / halves) * stride;
    const uint32_t half = (uint32_t)(h % halves) * TBK_SLOTS_PER_BUCKET;
    // slot pairs (2,3), (4,5), (6,7) of a list are 16 contiguous bytes in either layout
    if (flags & TBK_FLAG_FRONT) {
        // front layout: slot 2 > slot 3 says "this list has keys behind the front" (a fifth key, or beyond)
        ulonglong2 *front = reinterpret_cast<ulonglong2 *>(line + tbk_slot_at(flags, stride, half, 2));
        const ulonglong2 f = *front;
        if (f.y != TBK_EMPTY) {
            const bool more = line[tbk_slot_at(flags, stride, half, 4)] != TBK_EMPTY;
            if ((f.x > f.y) != more) *front = make_ulonglong2(f.y, f.x);
        }
        if (flags & TBK_FLAG_GUESTS) {
            // ... and slot 0 > slot 1 says "some of these four slots hold the other list's keys" (tagged;
            // a lone guest in slot 0 moves to slot 1 for that: EMPTY is larger than anything)
            ulonglong2 *head = reinterpret_cast<ulonglong2 *>(line + tbk_slot_at(flags, stride, half, 0));
            const ulonglong2 h0 = *head;
            const bool tagged = ((h0.x != TBK_EMPTY) && (h0.x & TBK_GUEST)) || ((h0.y != TBK_EMPTY) && (h0.y & TBK_GUEST)) ||
                                ((f.x != TBK_EMPTY) && (f.x & TBK_GUEST)) || ((f.y != TBK_EMPTY) && (f.y & TBK_GUEST));
            if ((h0.x > h0.y) != tagged) *head = make_ulonglong2(h0.y, h0.x);
        }
    }
    ulonglong2 *last = reinterpret_cast<ulonglong2 *>(line + tbk_slot_at(flags, stride, half, 6));
    ulonglong2 v = *last;
    if (v.y == TBK_EMPTY) return;  // not full: nothing went past it, and slot 6 <= slot 7 = EMPTY already says so
    const bool past = (overflowed[h >> 5] >> (h & 31)) & 1u;
    if ((v.x > v.y) != past) *last = make_ulonglong2(v.y, v.x);
    if (left_line != nullptr) {
        ulonglong2 *mid = reinterpret_cast<ulonglong2 *>(line + tbk_slot_at(flags, stride, half, 4));
        const ulonglong2 m = *mid;
        const bool left = (left_line[h >> 5] >> (h & 31)) & 1u;
        if ((m.x > m.y) != left) *mid = make_ulonglong2(m.y, m.x);
    }
}

__global__ void __launch_bounds__(256)
tbk_contains_kernel(TbkTableView t, const uint64_t *__restrict__ keys, uint64_t n,
                    uint8_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = tbk_lookup_slow(t, keys[i]) ? 1 : 0;
}

// =======================================================================================
// probe
// =======================================================================================
// Work decomposition.  The batch is one byte stream of `total` bases; window starts are
// cut into PASSes of 2048 consecutive positions.  One wave owns one pass: lane l owns the
// 32 window starts P0+32l .. P0+32l+31 and needs bases P0+32l .. P0+32l+31+k-1 <= 63
// bases = four 16-base chunks.  Each lane loads two 16-byte chunks of the read stream
// (coalesced: 1 KiB per wave-instruction), packs each to 32 bits of 2-bit codes + a 16-bit
// "not ACGT" mask, and stages them in the wave's own LDS region; lanes 0/1 also stage the
// two halo chunks.  Each lane then reads its four chunks back and holds a 128-bit forward
// stream and the 128-bit reverse-complement stream in registers, from which each window's
// forward and reverse-complement k-mer are bit slices (no per-base loop).  A longer run of
// windows per lane means fewer "first window of the lane" fetches that cannot re-use a line.
//
// Probing is quad-cooperative: a bucket is one 128-byte line [8 hapA slots | 8 hapB slots]
// read by the 4 lanes of a quad, each lane taking 16 bytes (2 slots) of the hapA half and
// 16 bytes of the hapB half, so one wave-instruction touches 16 whole lines with 4 adjacent
// lanes per line (coalesced bucket-line loads) and a window costs ONE random line for both
// probes.  In sub-step (j, s) quad q probes window j of its lane s: key and bucket index are
// broadcast inside the quad with DPP quad_perm moves, each lane compares its slots, and the
// v_cmp results are the wave ballots: a key is stored at most once, in one of the two halves
// (hapA has priority over hapB, c/kmers.c:291-294, so a key both lists hold is kept for hapA
// only when the table is built), hence popcount(ballot) is the number of windows that hit.
//
// Per-read attribution.  A pass that lies inside one read (the usual case for long
// reads) accumulates its two counts in scalar registers and issues one atomicAdd pair; its odd
// lanes walk their windows downwards (see probe_pass) so that neighbouring lanes share the
// line at their boundary.  In a pass that touches several reads every lane counts the hits of
// its own windows per read and hands them to per-read tallies in LDS (count_hits).

#ifndef TBK_UNROLL
#define TBK_UNROLL 2      // j-loop unroll of the probe pass
#endif
#ifndef TBK_SAMP_UNROLL
#define TBK_SAMP_UNROLL 4 // j-loop unroll of the mod-sampling variants
#endif
#ifndef TBK_FULL_UNROLL
#define TBK_FULL_UNROLL TBK_SAMP_UNROLL   // ... of the full-key kernels (1: 62 registers with a span of eight - eight waves per SIMD - for sixteen more moves per window)
#endif
#ifndef TBK_MIN_WAVES
#define TBK_MIN_WAVES 5         // waves per SIMD the register allocator must leave room for: single-read passes, front layout ...
#endif
#ifndef TBK_MIN_WAVES_WHOLE
#define TBK_MIN_WAVES_WHOLE 5   // ... single-read passes in whole lines up to W = 6 (W = 7, 8 would spill vector registers: 4) ...
#endif
#ifndef TBK_MIN_WAVES_MULTI
#define TBK_MIN_WAVES_MULTI 4   // ... and multi-read passes (tbk_probe_kernel)
#endif
// Timing diagnostics (tools/build_variant.sh; WRONG COUNTS, never shipped): where does a step's time go?
//   TBK_DIAG_NOLOAD   no window ever fetches a line: the arithmetic alone
//   TBK_DIAG_CHEAP    the bucket is a cheap hash of (position / 4): the line traffic of a scheme of density 0.25
//                     without the minimizer arithmetic
//   TBK_OCC_PAD=n     n more bytes of LDS per block: fewer waves per CU (13000: 3 per SIMD, 20000: 2)
#ifndef TBK_DIAG_NOLOAD
#define TBK_DIAG_NOLOAD 0
#endif
#ifndef TBK_DIAG_CHEAP
#define TBK_DIAG_CHEAP 0
#endif
#ifndef TBK_OCC_PAD
#define TBK_OCC_PAD 0
#endif
constexpr int TBK_WAVES_PER_BLOCK = 1;      // waves of a block share nothing; one-wave blocks schedule best (measured: 1 > 2 > 4 > 8)

#ifdef TBK_COUNTERS
// event counters of a debug build (tools/measure_realistic.py prints them): j-steps, j-steps on the
// careful path, careful sub-steps, walks queued, lane loads of bucket lines (4 per line), walk steps
__device__ unsigned long long tbk_dbg[8];
#define TBK_COUNT(i, n) dbg[i] += (n)
#else
#define TBK_COUNT(i, n)
#endif

struct ProbeArgs {
    // the read stream: ASCII bytes, or (codes != nullptr) already packed 16 bases to a word with a
    // dense 16-bit not-ACGT mask per chunk (the packed transfer format, tbk_pack.cpp)
    const uint32_t *codes;
    const uint16_t *bad16;
    uint64_t n_chunks;
    const uint8_t *bases;
    const uint64_t *offsets;  // n_reads + 1
    uint64_t n_reads;
    uint64_t total;           // offsets[n_reads]
    uint64_t n_passes;
    TbkPairView t;            // hapA | hapB interleaved
    int k;
    int32_t *counts;          // [n_reads][2], zeroed by the caller
    const uint32_t *pass_read;  // [n_passes] read that contains each pass's first position
    const uint32_t *multi_list; // passes that touch more than one read (the multi-read kernel's work list)
    const uint32_t *n_multi;    // their number
    const uint64_t *two_list;   // passes that touch exactly two reads, as pass | first read << 32 (the two-read kernel's work list; empty where that kernel is not built)
    const uint32_t *n_two;
    uint64_t pass_lo, pass_hi;  // this launch's share of the passes (a batch may be probed slice by slice, as its bases arrive)
};

// largest r in [0, n_reads] with offsets[r] <= pos (pos <= total, offsets[n_reads] = total)
__device__ __forceinline__ uint64_t find_read(const uint64_t *offsets, uint64_t n_reads, uint64_t pos) {
    uint64_t lo = 0, hi = n_reads + 1;  // offsets[lo] <= pos < offsets[hi] (virtually +inf)
    while (hi - lo > 1) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// 16 bytes (two slots) of a bucket line.  -DTBK_NT_LOADS marks the load non-temporal: a line is used by the
// quad that fetched it and hardly ever again before it has left the caches.
__device__ __forceinline__ ulonglong2 load_slots(const uint64_t *p) {
#ifdef TBK_NT_LOADS
    typedef unsigned long long tbk_v2 __attribute__((ext_vector_type(2)));
    const tbk_v2 v = __builtin_nontemporal_load(reinterpret_cast<const tbk_v2 *>(p));
    return make_ulonglong2(v.x, v.y);
#else
    return *reinterpret_cast<const ulonglong2 *>(p);
#endif
}

template <int S>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    // DPP quad_perm:[S,S,S,S]
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, S * 0x55, 0xF, 0xF, true);
}

// wave ballot of a predicate (v_cmp result as a 64-bit lane mask)
__device__ __forceinline__ uint64_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// per-quad OR of a lane mask, result at each quad's lane 0 bit
__device__ __forceinline__ uint64_t quad_any(uint64_t m) {
    return (m | (m >> 1) | (m >> 2) | (m >> 3)) & 0x1111111111111111ull;
}

// ---- deferred walks --------------------------------------------------------------------
// A lookup that misses in a home half some key went past must follow the key's probe sequence
// (tbk_next_bucket).  Doing that inside the window loop would stall the wave on one dependent
// HBM access per walk, and on lists shaped like real data (keys in runs of overlapping k-mers
// that share a minimizer) a noticeable share of halves overflows.  So the loop only ENQUEUES
// such lookups in a per-wave LDS queue; drain_walks resolves them 64 at a time, one lane per
// lookup, so that the latency of a walk step is paid once per 64 walks.  The halves are
// disjoint, so a queued lookup counts for hapA if the hapA walk finds the key, else for hapB if
// the hapB walk does.
constexpr int TBK_QCAP = 256;        // whole-line kernels: queue entries per wave; a window-loop step adds at most 128
constexpr int TBK_QCAP_FRONT = 128;  // front kernels: walks are queued by drain_back only, at most 32 per round
constexpr int TBK_BQCAP = 128;       // front kernels: windows waiting for the back half of their line; a step adds at most 64
#ifndef TBK_TMER_LUT
#define TBK_TMER_LUT 1   // entry kernels with 3w t-mer positions at W = 6 (t = 4): t-mer ranks from a 256-entry table in LDS (0: computed)
#endif
#ifndef TBK_SHORT_DRAIN
#define TBK_SHORT_DRAIN (TBK_BQCAP - 64)   // short keys: the back queue is drained when it holds more than this many windows
#endif
constexpr int TBK_QCAP_ENTRY = 96;   // entry kernels: walks are queued by drain_back_entry, at most 16 per round (with 96 entries a block's LDS is 4.6 KB: 32 one-wave blocks per CU, eight waves per SIMD)

// entry: x = key low, y = key high, z = home bucket, w = list (0 hapA, 1 hapB) | (read - first read of pass) << 1

// follow the probe sequence of a key of list `half` (0 hapA, 8 hapB) past its home bucket, until the
// key is found or a half that no key went past
template <bool FRONT>
__device__ __forceinline__ bool walk_one(const TbkPairView t, uint32_t half, uint64_t key, uint32_t bucket, bool pend) {
    constexpr uint32_t LAYOUT = FRONT ? TBK_FLAG_FRONT : 0u;  // compile-time: the whole-line walk keeps its plain pointer arithmetic
    bool found = false, first = true;
    // front layout: the window loop saw the first four slots of the home line only - the walk begins
    // with that line (its other four slots, its guests), not behind it
    bool at_home = FRONT;
    uint32_t guard = 0;
    while (ballot(pend) != 0 && guard++ <= t.n_buckets) {
        if (pend) {
            if (at_home) at_home = false;
            else { bucket = tbk_next_bucket(key, t.mz, t.n_buckets, bucket, first); first = false; }
            // 32 bytes at a time: this rare path must not set the kernel's register high-water mark
            const uint64_t *line = t.slots + (uint64_t)bucket * 16;
            bool hit = false;
            ulonglong2 v2 = make_ulonglong2(0, 0), v3 = make_ulonglong2(0, 0);
#pragma unroll 1
            for (uint32_t i = 0; i < 8; i += 4) {
                v2 = *reinterpret_cast<const ulonglong2 *>(line + tbk_slot_at(LAYOUT, 16, half, i));
                v3 = *reinterpret_cast<const ulonglong2 *>(line + tbk_slot_at(LAYOUT, 16, half, i + 2));
                hit = hit || v2.x == key || v2.y == key || v3.x == key || v3.y == key;
            }
            if (FRONT && !hit && (t.guests & TBK_FLAG_GUESTS)) {
                // front layout: the key may sit, tagged, in the other list's front slots of this line
                const uint64_t tagged = key | TBK_GUEST;
                const ulonglong2 g0 = *reinterpret_cast<const ulonglong2 *>(line + tbk_slot_at(LAYOUT, 16, half ^ 8u, 0));
                const ulonglong2 g1 = *reinterpret_cast<const ulonglong2 *>(line + tbk_slot_at(LAYOUT, 16, half ^ 8u, 2));
                hit = g0.x == tagged || g0.y == tagged || g1.x == tagged || g1.y == tagged;
            }
            found = found || hit;
            bool past = !hit && v3.x > v3.y;  // slot 6 > slot 7: a key went past this half
            if (past && (t.guests & TBK_FLAG_GUESTS)) {
                // ... into the other half of this line (tagged), or - slot 4 > slot 5 - out of the line
                const bool left = v2.x > v2.y;
                const uint64_t tagged = key | TBK_GUEST;
                bool guest = false;
#pragma unroll 1
                for (uint32_t i = 0; i < 8; i += 2) {
                    const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(line + tbk_slot_at(LAYOUT, 16, half ^ 8u, i));
                    guest = guest || w.x == tagged || w.y == tagged;
                }
                found = found || guest;
                past = !guest && left;
            }
            pend = past;
        }
    }
    return found;
}

// Hits of a pass that touches several reads: every lane counts the hits of its own windows while
// they stay in one read and hands the sum to the wave's per-read tallies in LDS (reads r_first ..
// r_first + TBK_RCNT - 1, flushed with one global atomic per read at the end of the pass; reads
// beyond that - passes full of tiny reads - go straight to the global counters).  One global
// atomic per hit would serialise on two addresses per read: on data with dense hits the 14 % of
// passes that straddle a read end then cost 7 % of the whole run.
constexpr int TBK_RCNT = 64;
__device__ __forceinline__ void count_hits(const ProbeArgs &p, uint32_t *rcnt, uint64_t r_first, uint32_t rrel, uint32_t hap, uint32_t n) {
    if (rrel < (uint32_t)TBK_RCNT) atomicAdd(&rcnt[2 * rrel + hap], n);
    else atomicAdd(&p.counts[2 * (r_first + rrel) + hap], (int)n);
}

template <bool MULTI, bool FRONT>
__device__ __forceinline__ void drain_walks(const ProbeArgs &p, const uint4 *q, uint32_t qn, uint64_t r_first,
                                            uint32_t lane, uint32_t &acc_a, uint32_t &acc_b, uint32_t *rcnt) {
    for (uint32_t base = 0; base < qn; base += 64) {
        const bool act = base + lane < qn;
        uint4 it = make_uint4(0, 0, 0, 0);
        if (act) it = q[base + lane];
        const uint64_t key = (uint64_t)it.x | ((uint64_t)it.y << 32);
        const bool found = walk_one<FRONT>(p.t, (it.w & 1u) * 8u, key, it.z, act);
        const bool count_a = found && !(it.w & 1u), count_b = found && (it.w & 1u);
        if (!MULTI) {
            acc_a += (uint32_t)__popcll(ballot(count_a));
            acc_b += (uint32_t)__popcll(ballot(count_b));
        } else {
            if (count_a) count_hits(p, rcnt, r_first, it.w >> 1, 0, 1);
            if (count_b) count_hits(p, rcnt, r_first, it.w >> 1, 1, 1);
        }
    }
}

// Front layout: a window that missed in a front with keys behind it (the order of a list's slots 2 and 3)
// is settled from the BACK half of its home line - the list's slots 4..7 and, tagged, the other list's
// guests there - which the L2 already holds (it fetched the whole 128-byte line).  The window loop only
// queues such windows (owner lane: key, home bucket, which lists have keys behind their front);
// drain_back takes 16 entries at a time, one quad per entry, each lane loading 16 bytes of the back half
// [A4 A5 | A6 A7 | B4 B5 | B6 B7] - the same coalesced 64-byte request shape as the window loop's - so the
// latency of the second look is paid once per 16 windows instead of once per step.  Only a window whose
// key may have LEFT THE LINE (the list's slots 6 > 7: keys went past the half, and 4 > 5: out of the
// line) goes on to the exact walk, which starts over at the home line.
template <bool MULTI>
__device__ __forceinline__ void drain_back(const ProbeArgs &p, const uint4 *bq, uint32_t qb, uint4 *walkq, uint32_t &qn,
                                           uint64_t r_first, uint32_t lane, uint32_t &acc_a, uint32_t &acc_b, uint32_t *rcnt) {
    const uint32_t sub = lane & 3u, quad = lane >> 2;
    const bool guests = (p.t.guests & TBK_FLAG_GUESTS) != 0;
    for (uint32_t base = 0; base < qb; base += 16) {
        const bool act = base + quad < qb;
        uint4 it = make_uint4((uint32_t)TBK_NOKEY, (uint32_t)(TBK_NOKEY >> 32), 0, 0);
        ulonglong2 v = make_ulonglong2(TBK_EMPTY, TBK_EMPTY);
        if (act) {
            it = bq[base + quad];
            v = load_slots(p.t.slots + (uint64_t)it.z * 16 + 8 + sub * 2);
        }
        const uint64_t key = (uint64_t)it.x | ((uint64_t)it.y << 32);
        const uint64_t plain = ballot(v.x == key) | ballot(v.y == key);
        uint64_t tagged = 0;
        if (guests) { const uint64_t kt = key | TBK_GUEST; tagged = ballot(v.x == kt) | ballot(v.y == kt); }
        // quad lanes 0,1 hold hapA's slots 4..7, lanes 2,3 hapB's; a tagged match belongs to the other list
        const uint64_t hit_a = (plain & 0x3333333333333333ull) | (tagged & 0xCCCCCCCCCCCCCCCCull);
        const uint64_t hit_b = (plain & 0xCCCCCCCCCCCCCCCCull) | (tagged & 0x3333333333333333ull);
        // order of a lane's two slots: lane 0 - hapA keys left the line, lane 1 - keys went past hapA's half; lanes 2, 3: hapB
        const uint64_t ord = ballot(v.x > v.y);
        const uint64_t want_a = ballot((it.w & 2u) != 0) & 0x1111111111111111ull, want_b = ballot((it.w & 1u) != 0) & 0x1111111111111111ull;
        const uint64_t miss = ~quad_any(hit_a | hit_b);
        uint64_t walk_a = want_a & miss & (ord >> 1), walk_b = want_b & miss & (ord >> 3);
        if (guests) { walk_a &= ord; walk_b &= ord >> 2; }  // without guests a key past its half is out of the line
        if (!MULTI) {
            acc_a += (uint32_t)__popcll(hit_a);
            acc_b += (uint32_t)__popcll(hit_b);
        } else {
            if ((hit_a >> lane) & 1ull) count_hits(p, rcnt, r_first, it.w >> 2, 0, 1);
            if ((hit_b >> lane) & 1ull) count_hits(p, rcnt, r_first, it.w >> 2, 1, 1);
        }
        const uint64_t queued = walk_a | walk_b;
        if (queued) {
            const uint64_t me = 1ull << lane;
            const uint32_t n_a = (uint32_t)__popcll(walk_a);
            if (walk_a & me) walkq[qn + (uint32_t)__popcll(walk_a & (me - 1))] = make_uint4(it.x, it.y, it.z, (it.w >> 2) << 1);
            if (walk_b & me) walkq[qn + n_a + (uint32_t)__popcll(walk_b & (me - 1))] = make_uint4(it.x, it.y, it.z, ((it.w >> 2) << 1) | 1u);
            qn += n_a + (uint32_t)__popcll(walk_b);
            if (qn > TBK_QCAP_FRONT - 32) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                drain_walks<MULTI, true>(p, walkq, qn, r_first, lane, acc_a, acc_b, rcnt);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                qn = 0;
            }
        }
    }
}

// One wave pass.  W = m-mers per minimizer span (0: plain hashing, one random line per
// window).  MULTI = the pass touches more than one read.
// TWO = the pass touches exactly two reads (15 kb reads: one pass in seven): the single-read pass's code with the
// boundary folded into the lanes' not-ACGT masks and into four scalar ownership masks, instead of MULTI's
// per-lane read bookkeeping and tallies - so that these passes, too, run at five waves per SIMD.
template <int W, bool M64, bool SAMP, bool MULTI, bool FRONT, bool TWO = false>
__device__ __forceinline__ void probe_pass(const ProbeArgs &p, const uint64_t e0, const uint64_t e1,
                                           const uint64_t e2, const uint64_t e3, const uint64_t P0,
                                           const uint64_t r_first, const uint64_t r_first_end, const uint32_t lane,
                                           uint4 *walkq, uint4 *backq, uint32_t *rcnt) {
    const int k = p.k;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t sub = lane & 3u;
    // Forward stream S = bases 0..63 of this lane as 2-bit codes (base i at bits 2i).  Window
    // j's forward k-mer is the low 2k bits of S >> 2j: S is rolled right by one base per
    // window.  R = reverse complement of the 64-base stream; window j's reverse-complement
    // k-mer is bits [128-2k-2j, 128-2j) of R.  R is pre-shifted right by 64-2k once and then
    // rolled LEFT by one base per window, so that k-mer always sits at bits [64, 64+2k):
    // words (t2, t3), no per-window shift.
    //
    // In a pass that lies inside one read, ODD LANES WALK THEIR 32 WINDOWS DOWNWARDS.  A lane's first
    // window can never re-use a line, and 71 % of the time it needs the very line its neighbour
    // lane holds for the adjacent window; with neighbours walking towards each other (or away from
    // each other) the two fetches of that line happen in the same step or a few steps apart, so the
    // second one is served by L2 instead of HBM.  Walking down needs no second code path: the
    // windows of b[0 .. 30+k] taken downwards are the windows of its reverse complement c[i] =
    // comp(b[30+k-i]) taken upwards, with forward and reverse-complement k-mer swapped - the
    // canonical k-mer and the (canonical, central) minimizer do not notice.  So an odd lane starts
    // from S' = R >> 2(33-k), R' = S << 2(33-k) and the bit-reversed not-ACGT mask, and runs the
    // same loop.  The read end is folded into that mask first (bases at or past it count as not
    // ACGT), so window validity is one mask test in either direction.
    unsigned __int128 S128 = (unsigned __int128)(uint32_t)e0 | ((unsigned __int128)(uint32_t)e1 << 32) |
                             ((unsigned __int128)(uint32_t)e2 << 64) | ((unsigned __int128)(uint32_t)e3 << 96);
    unsigned __int128 R128 = (unsigned __int128)rev_pairs(~(uint32_t)e3) | ((unsigned __int128)rev_pairs(~(uint32_t)e2) << 32) |
                             ((unsigned __int128)rev_pairs(~(uint32_t)e1) << 64) | ((unsigned __int128)rev_pairs(~(uint32_t)e0) << 96);
    // not-ACGT flags of the lane's 64 bases, rolled right by one per window: window j is
    // clean when the low k bits are zero
    uint64_t bad64 = (uint64_t)((uint32_t)(e0 >> 32) | ((uint32_t)(e1 >> 32) << 16)) |
                     ((uint64_t)((uint32_t)(e2 >> 32) | ((uint32_t)(e3 >> 32) << 16)) << 32);
    // TWO: lanes whose windows all start in the pass's second read, and the lane (if any) whose windows start in the
    // first read up to the boundary and in the second from it on.  Both kinds see their bases as the second read does;
    // the straddling lane's windows of the first read must, besides, end before the boundary (the window loop's `cross`).
    bool is_second = false, is_strad = false;
    if (TWO) {
        const uint64_t p_first = P0 + (uint64_t)lane * TBK_WPL;
        const uint64_t r2_end = p.offsets[r_first + 2];  // (exists: the pass's second read is read r_first + 1)
        is_second = p_first >= r_first_end;
        const uint64_t inside = is_second ? 64 : r_first_end - p_first;  // bases of this lane before the first read's end
        is_strad = inside < TBK_WPL;
        if (is_second || is_strad) {
            const uint64_t inside2 = r2_end > p_first ? r2_end - p_first : 0;
            if (inside2 < 64) bad64 |= ~0ull << inside2;
        } else if (inside < 64) {
            bad64 |= ~0ull << inside;
        }
    }
    if (!MULTI) {
        if (!TWO) {
            const uint64_t p_first = P0 + (uint64_t)lane * TBK_WPL;
            const uint64_t inside = r_first_end > p_first ? r_first_end - p_first : 0;  // bases of this lane before the read end
            if (inside < 64) bad64 |= ~0ull << inside;
        }
        if ((lane & 1u) && !(TWO && is_strad)) {  // (a two-read pass's straddling lane keeps walking upwards: its windows change reads on the way)
            const int sh = 33 - k;  // >= 1
            const unsigned __int128 s_up = R128 >> (2 * sh), r_up = S128 << (2 * sh);
            S128 = s_up; R128 = r_up;
            bad64 = __brevll(bad64) >> sh;
        }
    }
    uint32_t s0 = (uint32_t)S128, s1 = (uint32_t)(S128 >> 32), s2 = (uint32_t)(S128 >> 64), s3 = (uint32_t)(S128 >> 96);
    uint32_t t0, t1, t2, t3;
    {
        const unsigned __int128 Rs = R128 >> (64 - 2 * k);
        t0 = (uint32_t)Rs; t1 = (uint32_t)(Rs >> 32); t2 = (uint32_t)(Rs >> 64); t3 = (uint32_t)(Rs >> 96);
    }
    uint32_t bad_lo = (uint32_t)bad64, bad_hi = (uint32_t)(bad64 >> 32);
    const uint32_t badk = k == 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);

    // ---- minimizer state -----------------------------------------------------------------
    // win[i] = hash(canonical m-mer starting at base j + o + i) for the current window j: a
    // W-deep shift register.  The newest m-mer (i = W-1) is bits [2(o+W-1), +2m) of rolled S;
    // its reverse complement sits at base o of the reverse-complement k-mer (the span is
    // central), i.e. bits [64+2o, +2m) of rolled R.
    // Two sampling schemes share the machinery (tbk_common.h "bucket selection"):
    //   SAMP = false  random minimizer: win[] holds the hashes of the span's W m-mers;
    //   SAMP = true   mod-sampling: win[] holds (hash & ~15) | position of the span's 2W
    //                 t-mers (t = m - W, always <= 16 bases), position 0 = first of the span.
    // m-mers are 32-bit values for m <= 16 (M64 = false) and 64-bit ones above.  All slices
    // come from the same two 64-bit words, (s1:s0) forward and (t3:t2) reverse complement,
    // because every m-mer / t-mer of the span lies inside the window's k-mer; the piece at
    // forward base offset a sits at reverse-complement base offset k - len - a.
    constexpr int NW = W > 0 ? (SAMP ? 2 * W : W) : 1;
    using win_t = typename std::conditional<(M64 && !SAMP), uint64_t, uint32_t>::type;
    win_t win[NW];
    uint64_t mmask = 0;
    uint32_t tmask = 0, fsh_new = 0, bsh_new = 0, span_o = 0;
    auto mmer_order = [&](uint64_t fwd64, uint64_t rc64, uint32_t fsh, uint32_t bsh) -> win_t {
        if (M64) {
            const uint64_t x = (fwd64 >> fsh) & mmask, y = (rc64 >> bsh) & mmask;
            return (win_t)tbk_mmer_hash64(x < y ? x : y);
        }
        const uint32_t x = (uint32_t)(fwd64 >> fsh) & (uint32_t)mmask, y = (uint32_t)(rc64 >> bsh) & (uint32_t)mmask;
        return (win_t)tbk_mmer_hash(x < y ? x : y);
    };
    auto tmer_rank = [&](uint64_t fwd64, uint64_t rc64, uint32_t fsh, uint32_t bsh, uint32_t pos) -> uint32_t {
        const uint32_t x = (uint32_t)(fwd64 >> fsh) & tmask, y = (uint32_t)(rc64 >> bsh) & tmask;
        return (tbk_mmer_hash(x < y ? x : y) & ~15u) | pos;
    };
    if (W > 0) {
        const int m = p.t.mz.m, o = p.t.mz.o;
        span_o = (uint32_t)o;
        mmask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
        const uint64_t fs = ((uint64_t)s1 << 32) | s0, bs = ((uint64_t)t3 << 32) | t2;
        if (SAMP) {
            const int t = p.t.mz.t;
            tmask = t >= 16 ? 0xFFFFFFFFu : ((1u << (2 * t)) - 1u);
            win[0] = (win_t)0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i + 1 < NW; i++)  // prologue: window 0's first 2W-1 t-mers; tag = index of the t-mer in the lane's stream, mod 16
                win[i + 1] = (win_t)tmer_rank(fs, bs, (uint32_t)(2 * (o + i)), (uint32_t)(2 * (o + NW - 1 - i)), (uint32_t)i);
            fsh_new = (uint32_t)(2 * (o + NW - 1));
            bsh_new = (uint32_t)(2 * o);
        } else {
            win[0] = (win_t)~0ull;
#pragma unroll
            for (int i = 0; i + 1 < W; i++)  // prologue: the W-1 m-mers window 0 shares with window -1
                win[i + 1] = mmer_order(fs, bs, (uint32_t)(2 * (o + i)), (uint32_t)(2 * (o + W - 1 - i)));
            fsh_new = (uint32_t)(2 * (o + W - 1));
            bsh_new = (uint32_t)(2 * o);
        }
    }

    // ---- read bookkeeping ------------------------------------------------------------------
    const uint64_t p_lane = P0 + (uint64_t)lane * TBK_WPL;
    uint64_t rid = r_first;
    uint64_t rend = r_first_end;
    if (MULTI) {
        // This lane's first window start may be in a later read than the pass start.  Gallop
        // forward from the pass's first read (a long read ending inside the pass costs one or
        // two loads per lane; a pass full of tiny reads still takes only O(log) steps).
        const uint64_t pl = p_lane < p.total ? p_lane : p.total;
        uint64_t lo = r_first, step = 1;  // invariant: offsets[lo] <= pl
        while (lo + step <= p.n_reads && p.offsets[lo + step] <= pl) { lo += step; step <<= 1; }
        uint64_t hi = lo + step <= p.n_reads ? lo + step : p.n_reads + 1;  // offsets[hi] > pl (virtually +inf)
        while (hi - lo > 1) {
            const uint64_t mid = lo + ((hi - lo) >> 1);
            if (p.offsets[mid] <= pl) lo = mid; else hi = mid;
        }
        rid = lo;
        rend = rid < p.n_reads ? p.offsets[rid + 1] : p.total;
    }
    // windows j with j + k <= rel_end lie inside the current read
    auto rel = [&](uint64_t pos) -> uint32_t {  // pos relative to this lane's first window, clamped
        return pos > p_lane ? (uint32_t)(pos - p_lane < 0x40000000ull ? pos - p_lane : 0x40000000ull) : 0u;
    };
    uint32_t rel_end = rel(rend);
    uint32_t acc_a = 0, acc_b = 0;  // wave-uniform in the single-read case (TWO: the pass's first read)
    uint32_t lane_a = 0, lane_b = 0;  // multi-read pass: this lane's hits in read `rid`
    // TWO: hits of the second read, and per sub-step s the lanes (all four of a quad) whose quad's window s belongs
    // to the FIRST read at the current step: fixed but for the straddling lane's quad, which changes sides once
    uint32_t acc2_a = 0, acc2_b = 0;
    uint64_t own1[4] = {0, 0, 0, 0};
    uint32_t jb_s = 0xFFFFu;  // the step at which the straddling lane changes sides (wave-uniform; none: never)
    if (TWO) {
        const uint32_t brel = (uint32_t)(r_first_end - P0);  // 1 .. TBK_PASS - 1
        if (brel % TBK_WPL) jb_s = brel % TBK_WPL;
#pragma unroll
        for (int s = 0; s < 4; s++) own1[s] = ballot(((lane & ~3u) + (uint32_t)s) * TBK_WPL < brel);  // owner lane 4q + s starts in the first read
    }

    // the line each quad slot holds from the previous window of the same lane
    ulonglong2 va[4], vb[4];
#pragma unroll
    for (int s = 0; s < 4; s++) { va[s] = make_ulonglong2(0, 0); vb[s] = make_ulonglong2(0, 0); }
    uint32_t last_bk = 0x7FFFFFFFu;  // bucket of this lane's previous valid window (none yet)
    uint32_t qn = 0;                 // queued walks (wave-uniform)
    uint32_t qb = 0;                 // front layout: windows waiting for the back half of their line (wave-uniform)
#ifdef TBK_COUNTERS
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // bucket of the window the rolled streams stand at (window jj of this lane); moves the minimizer
    // state on by one window, so it is called once per window, in order
    auto bucket_here = [&](const int jj) -> uint32_t {
        uint32_t hsel;
        if (TBK_DIAG_CHEAP) {
            hsel = tbk_mmer_hash((uint32_t)((p_lane + (uint64_t)jj) >> 2));
        } else if (W > 0 && SAMP) {
            // mod-sampling: shift in the span's newest t-mer, find the smallest rank (any of the
            // tied ones will do: the table holds the key under each, see tbk_common.h), turn its
            // stream index into a position inside this window's span, sample the m-mer at
            // position mod W
            const uint64_t fs = ((uint64_t)s1 << 32) | s0, bs = ((uint64_t)t3 << 32) | t2;
#pragma unroll
            for (int i = 0; i + 1 < NW; i++) win[i] = win[i + 1];
            win[NW - 1] = (win_t)tmer_rank(fs, bs, fsh_new, bsh_new, (uint32_t)(jj + NW - 1) & 15u);
            uint32_t best = (uint32_t)win[0];
#pragma unroll
            for (int i = 1; i < NW; i++) best = (uint32_t)win[i] < best ? (uint32_t)win[i] : best;
            const uint32_t x = (best - (uint32_t)jj) & 15u;  // 0 .. 2W-1
            const uint32_t pos = x >= (uint32_t)W ? x - (uint32_t)W : x;
            const uint32_t fsh = 2u * (span_o + pos), bsh = 2u * (span_o + (uint32_t)W - 1u - pos);
            if (M64) {
                const uint64_t mx = (fs >> fsh) & mmask, my = (bs >> bsh) & mmask;
                hsel = (uint32_t)tbk_mmer_hash64(mx < my ? mx : my);
            } else {
                const uint32_t mx = (uint32_t)(fs >> fsh) & (uint32_t)mmask, my = (uint32_t)(bs >> bsh) & (uint32_t)mmask;
                hsel = tbk_mmer_hash(mx < my ? mx : my);
            }
        } else if (W > 0) {
            // random minimizer: shift in the newest m-mer of this window's span, take the minimum
#pragma unroll
            for (int i = 0; i + 1 < W; i++) win[i] = win[i + 1];
            win[W - 1] = mmer_order(((uint64_t)s1 << 32) | s0, ((uint64_t)t3 << 32) | t2, fsh_new, bsh_new);
            win_t best = win[0];
#pragma unroll
            for (int i = 1; i < W; i++) best = win[i] < best ? win[i] : best;
            hsel = M64 ? (uint32_t)best : tbk_scramble((uint32_t)best);
        } else {
            const uint64_t f_ = ((uint64_t)s0 | ((uint64_t)s1 << 32)) & kmask, r_ = ((uint64_t)t2 | ((uint64_t)t3 << 32)) & kmask;
            hsel = tbk_mix32(f_ < r_ ? f_ : r_);
        }
        return tbk_reduce(hsel, p.t.n_buckets);
    };
    // mod-sampling: a deeper unroll lets the 2W-deep shift register be renamed instead of moved
    constexpr int kUnroll = SAMP ? TBK_SAMP_UNROLL : TBK_UNROLL;
#pragma unroll kUnroll
    for (int j = 0; j < TBK_WPL; j++) {
        // ---- this lane's window j ---------------------------------------------------
        const uint64_t fwd = ((uint64_t)s0 | ((uint64_t)s1 << 32)) & kmask;
        const uint64_t rc = ((uint64_t)t2 | ((uint64_t)t3 << 32)) & kmask;
        const uint64_t key = fwd < rc ? fwd : rc;
        if (MULTI) {
            // window j starts at or past the current read's end: move to the read that holds it
            while ((uint32_t)j >= rel_end && rid < p.n_reads) {
                // hand the finished read's hits to the wave's per-read tallies
                if (lane_a) { count_hits(p, rcnt, r_first, (uint32_t)(rid - r_first), 0, lane_a); lane_a = 0; }
                if (lane_b) { count_hits(p, rcnt, r_first, (uint32_t)(rid - r_first), 1, lane_b); lane_b = 0; }
                rid++;
                rend = rid < p.n_reads ? p.offsets[rid + 1] : p.total;
                rel_end = rel(rend);
            }
        }
        if (TWO && (uint32_t)j == jb_s) {
            // the straddling lane's windows belong to the second read from here on: its quad's bits of own1 go
            // (sub-step = that lane's place in its quad)
            const uint64_t strad = ballot(is_strad);  // one lane
#pragma unroll
            for (int s = 0; s < 4; s++) if (strad & (0x1111111111111111ull << s)) own1[s] &= ~(quad_any(strad) * 15ull);  // (all four bits of that quad)
        }
        bool ok = (bad_lo & badk) == 0;  // single-read pass: the read end is part of the mask
        if (TWO) {
            const bool cross = (uint32_t)j < jb_s && (uint32_t)(j + k) > jb_s;  // (wave-uniform) a window of the first read that would reach over the boundary
            ok = ok && !(cross && is_strad);
        }
        if (MULTI) ok = ok && (uint32_t)(j + k) <= rel_end && rid < p.n_reads;
        const uint32_t bkt = bucket_here(j);
        // an invalid window keeps the previous bucket (it never forces a fetch) and looks up
        // TBK_NOKEY, which is never stored (it can never hit)
        // Bit 31 of the broadcast bucket says "not the bucket of this lane's previous window": only
        // then do the quad's lanes fetch the line; otherwise they still hold it.  (Bucket indices stay
        // below 2^31: 2^31 lines would be a 256 GB table.)
        const bool fresh = ok && bkt != last_bk;
        const uint32_t my_bk = (ok ? bkt : last_bk) | (fresh ? 0x80000000u : 0u);
        last_bk = my_bk & 0x7FFFFFFFu;
        const uint32_t my_klo = ok ? (uint32_t)key : (uint32_t)TBK_NOKEY;
        const uint32_t my_khi = ok ? (uint32_t)(key >> 32) : (uint32_t)(TBK_NOKEY >> 32);
        const uint32_t my_rid = (uint32_t)rid;
        auto two_rid = [&]() -> uint32_t { return (is_second || (is_strad && (uint32_t)j >= jb_s)) ? 1u : 0u; };  // TWO: 0 / 1 = this lane's window belongs to the pass's first / second read

        // advance to window j+1: S >>= 2, R <<= 2, bad >>= 1
        s0 = (s0 >> 2) | (s1 << 30); s1 = (s1 >> 2) | (s2 << 30); s2 = (s2 >> 2) | (s3 << 30); s3 >>= 2;
        t3 = (t3 << 2) | (t2 >> 30); t2 = (t2 << 2) | (t1 >> 30); t1 = (t1 << 2) | (t0 >> 30); t0 <<= 2;
        bad_lo = (bad_lo >> 1) | (bad_hi << 31); bad_hi >>= 1;

        // ---- four quad sub-steps: the quad's 4 windows, one 128-byte line each ----------
        uint32_t klo[4], khi[4], bk[4];
        bk[0] = quad_bcast<0>(my_bk); bk[1] = quad_bcast<1>(my_bk); bk[2] = quad_bcast<2>(my_bk); bk[3] = quad_bcast<3>(my_bk);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            // fetch only when this window's line differs from the one the slot already holds
            // (minimizer mode: consecutive windows mostly share it)
            TBK_COUNT(4, __popcll(ballot((int32_t)bk[s] < 0)));
            if ((int32_t)bk[s] < 0 && !TBK_DIAG_NOLOAD) {
                const uint64_t *line = p.t.slots + (uint64_t)(bk[s] & 0x7FFFFFFFu) * 16 + sub * 2;
                if (FRONT) {
                    va[s] = load_slots(line);  // the front of the line: [A0 A1 | A2 A3 | B0 B1 | B2 B3], 16 bytes per quad lane
                } else {
                    va[s] = load_slots(line);
                    vb[s] = load_slots(line + 8);
                }
            }
        }

        if constexpr (FRONT) {
            // Front layout (tbk_common.h): quad lanes 0,1 hold hapA's first four slots of the line, lanes 2,3
            // hapB's.  One compare pair serves both lists - which list a hit counts for is the lane it
            // fell on - and the order of a lane's two slots is, on lanes 1 and 3, the list's "look behind
            // the front" flag.  A window that misses in a front with that flag goes to the deferred walk,
            // which begins at the home line; everything else is settled here.
            const uint32_t klo0 = quad_bcast<0>(my_klo), khi0 = quad_bcast<0>(my_khi), klo1 = quad_bcast<1>(my_klo), khi1 = quad_bcast<1>(my_khi);
            const uint32_t klo2 = quad_bcast<2>(my_klo), khi2 = quad_bcast<2>(my_khi), klo3 = quad_bcast<3>(my_klo), khi3 = quad_bcast<3>(my_khi);
            const uint32_t klo[4] = {klo0, klo1, klo2, klo3}, khi[4] = {khi0, khi1, khi2, khi3};
            uint64_t hit[4], more[4], any_hit = 0, any_more = 0;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const uint64_t kk = (uint64_t)klo[s] | ((uint64_t)khi[s] << 32);
                hit[s] = ballot(va[s].x == kk) | ballot(va[s].y == kk);
                // the order of a lane's two slots: lanes 1 and 3 - the list has keys behind its front;
                // lanes 0 and 2 - these four slots hold keys of the OTHER list (tagged)
                more[s] = ballot(va[s].x > va[s].y);
                any_more |= more[s];
            }
            if ((p.t.guests & TBK_FLAG_GUESTS) && (any_more & 0x5555555555555555ull) != 0) {  // (without guests the order of slots 0 and 1 means nothing)
                // A list's fifth key of a bucket sits, tagged, in a free front slot of the other list before it
                // goes behind the front.  Found there it counts for the list whose lanes these are not: the
                // hit moves over to that list's lanes of the quad.
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if ((more[s] & 0x5555555555555555ull) == 0) continue;
                    const uint64_t tagged = ((uint64_t)klo[s] | ((uint64_t)khi[s] << 32)) | TBK_GUEST;
                    const uint64_t g = ballot(va[s].x == tagged) | ballot(va[s].y == tagged);
                    hit[s] |= ((g & 0x3333333333333333ull) << 2) | ((g & 0xCCCCCCCCCCCCCCCCull) >> 2);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {
                more[s] &= 0xAAAAAAAAAAAAAAAAull;
                any_hit |= hit[s];
            }
            any_more &= 0xAAAAAAAAAAAAAAAAull;
            TBK_COUNT(0, 1);
            if (any_more != 0) {
                TBK_COUNT(1, 1);
                // Windows that missed in a front with keys behind it: brought to the bit of the lane that owns the
                // window (quad q's sub-step s window belongs to lane 4q + s), so that every such lane queues its own
                // window - key, home bucket, which lists have keys behind their front - once per step.
                uint64_t need = 0, beh_a = 0, beh_b = 0;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (more[s] == 0) continue;
                    TBK_COUNT(2, 1);
                    const uint64_t ma = (more[s] & 0x2222222222222222ull) >> 1, mb = (more[s] & 0x8888888888888888ull) >> 3;
                    const uint64_t miss = ~quad_any(hit[s]);  // a hit in either list's front is final (the lists are disjoint)
                    need |= ((ma | mb) & miss) << s;
                    beh_a |= ma << s;
                    beh_b |= mb << s;
                }
                need &= ballot(ok);  // an invalid window looks up TBK_NOKEY: never stored, nothing to look for
                if (need) {
                    const uint64_t me = 1ull << lane;
                    if (need & me) {
                        const uint32_t rrel = MULTI ? my_rid - (uint32_t)r_first : TWO ? two_rid() : 0u;
                        backq[qb + (uint32_t)__popcll(need & (me - 1))] =
                            make_uint4(my_klo, my_khi, last_bk, (rrel << 2) | ((uint32_t)((beh_a >> lane) & 1ull) << 1) | (uint32_t)((beh_b >> lane) & 1ull));
                    }
                    qb += (uint32_t)__popcll(need);
                    TBK_COUNT(5, __popcll(need));
                }
            }
            if (any_hit != 0) {
                if (TWO) {
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const uint64_t h1 = hit[s] & own1[s], h2 = hit[s] & ~own1[s];
                        acc_a += (uint32_t)__popcll(h1 & 0x3333333333333333ull);
                        acc_b += (uint32_t)__popcll(h1 & 0xCCCCCCCCCCCCCCCCull);
                        acc2_a += (uint32_t)__popcll(h2 & 0x3333333333333333ull);
                        acc2_b += (uint32_t)__popcll(h2 & 0xCCCCCCCCCCCCCCCCull);
                    }
                } else if (!MULTI) {
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        acc_a += (uint32_t)__popcll(hit[s] & 0x3333333333333333ull);
                        acc_b += (uint32_t)__popcll(hit[s] & 0xCCCCCCCCCCCCCCCCull);
                    }
                } else {
                    uint64_t wa = 0, wb = 0;
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const uint64_t ha = hit[s] & 0x3333333333333333ull, hb = hit[s] & 0xCCCCCCCCCCCCCCCCull;
                        wa |= ((ha | (ha >> 1)) & 0x1111111111111111ull) << s;
                        wb |= (((hb >> 2) | (hb >> 3)) & 0x1111111111111111ull) << s;
                    }
                    lane_a += (uint32_t)(wa >> lane) & 1u;
                    lane_b += (uint32_t)(wb >> lane) & 1u;
                }
            }
        } else {
        // Fast path.  A key is stored at most once, in one of the two halves (hapB keys that hapA
        // holds are dropped at build time), so the raw ballots count windows and hapA-over-hapB
        // priority (c/kmers.c:291-294) needs no work here.  Only a window whose home half was left
        // by some key (slot 6 > slot 7, held by quad lane 3) may have to look further; those go
        // to the exact path.
        klo[0] = quad_bcast<0>(my_klo); khi[0] = quad_bcast<0>(my_khi); klo[1] = quad_bcast<1>(my_klo); khi[1] = quad_bcast<1>(my_khi);
        klo[2] = quad_bcast<2>(my_klo); khi[2] = quad_bcast<2>(my_khi); klo[3] = quad_bcast<3>(my_klo); khi[3] = quad_bcast<3>(my_khi);
        uint64_t hit_a[4], hit_b[4], full_a[4], full_b[4], full_any = 0, any_a = 0, any_b = 0;
        uint64_t kk[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            kk[s] = (uint64_t)klo[s] | ((uint64_t)khi[s] << 32);
            // one ballot per compare, OR-ed as scalars (a ballot of `a || b` costs two extra VALU
            // instructions: the i1 is materialised and compared again)
            hit_a[s] = ballot(va[s].x == kk[s]) | ballot(va[s].y == kk[s]);
            hit_b[s] = ballot(vb[s].x == kk[s]) | ballot(vb[s].y == kk[s]);
            // quad lane 3 holds slots 6 and 7 of each half: slot 6 > slot 7 says a key went past the half
            full_a[s] = ballot(va[s].x > va[s].y) & 0x8888888888888888ull;
            full_b[s] = ballot(vb[s].x > vb[s].y) & 0x8888888888888888ull;
            full_any |= full_a[s] | full_b[s];
            any_a |= hit_a[s];
            any_b |= hit_b[s];
        }
        TBK_COUNT(0, 1);
        uint64_t guests_a = 0, guests_b = 0;  // multi-read passes: windows whose key was found as a guest in the other half (bit of the owning lane)
        if (full_any != 0) {
            TBK_COUNT(1, 1);
            // Careful path: per-window (= per-quad) resolution, everything brought to the quad's
            // lane-0 bit.  A hit in the home line is final (the halves are disjoint: the other list
            // cannot hold the key); a miss in a half that keys went past is queued for a walk.
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if ((full_a[s] | full_b[s]) == 0) continue;  // raw ballots are already exact
                const uint64_t fa = full_a[s] >> 3, fb = full_b[s] >> 3;  // at the quad's lane-0 bit
                TBK_COUNT(2, 1);
                uint64_t walk_a, walk_b;
                if (p.t.guests & TBK_FLAG_GUESTS) {
                    // The keys that went past a half went, tagged, into free slots of the line's other half
                    // first, and the quad holds that half already.  A tagged slot equal to the window's key
                    // IS the key (stored once, and only in its home line's other half when its own half was
                    // full), so these compares count as they stand: no masks, no per-window reduction.
                    const uint64_t tagged = kk[s] | TBK_GUEST;
                    const uint64_t ga = ballot(vb[s].x == tagged) | ballot(vb[s].y == tagged);  // hapA's guests sit in hapB's half
                    const uint64_t gb = ballot(va[s].x == tagged) | ballot(va[s].y == tagged);
                    if (TWO) {
                        acc_a += (uint32_t)__popcll(ga & own1[s]); acc2_a += (uint32_t)__popcll(ga & ~own1[s]);
                        acc_b += (uint32_t)__popcll(gb & own1[s]); acc2_b += (uint32_t)__popcll(gb & ~own1[s]);
                    } else if (!MULTI) {
                        acc_a += (uint32_t)__popcll(ga);
                        acc_b += (uint32_t)__popcll(gb);
                    } else {
                        guests_a |= quad_any(ga) << s;  // brought to the bit of the lane that owns the window (quad lane s)
                        guests_b |= quad_any(gb) << s;
                    }
                    // Only keys that LEFT THE LINE need a walk: quad lane 2 holds slots 4 and 5 of each half,
                    // whose order says whether any did (meaningful where keys went past the half).  Rare.
                    const uint64_t need_a = fa & ((ballot(va[s].x > va[s].y) & 0x4444444444444444ull) >> 2);
                    const uint64_t need_b = fb & ((ballot(vb[s].x > vb[s].y) & 0x4444444444444444ull) >> 2);
                    if ((need_a | need_b) == 0) continue;
                    const uint64_t valid = ballot(kk[s] != TBK_NOKEY) & 0x1111111111111111ull;
                    const uint64_t miss = valid & ~quad_any(hit_a[s] | hit_b[s] | ga | gb);  // a hit anywhere in the line is final
                    walk_a = miss & need_a;
                    walk_b = miss & need_b;
                } else {
                    const uint64_t valid = ballot(kk[s] != TBK_NOKEY) & 0x1111111111111111ull;
                    const uint64_t miss = valid & ~quad_any(hit_a[s] | hit_b[s]);  // a hit in either half is final
                    walk_a = miss & fa;
                    walk_b = miss & fb;
                }
                const uint64_t queued = walk_a | walk_b;
                if (queued) {
                    const uint64_t me = 1ull << lane;
                    // the read of the window: its owner is quad lane s (a constant once unrolled); taken
                    // here, where the whole wave is active - a DPP move cannot read a masked-off lane
                    const uint32_t rid_v = TWO ? two_rid() : my_rid;
                    const uint32_t rid_s = !(MULTI || TWO) ? 0u : s == 0 ? quad_bcast<0>(rid_v) : s == 1 ? quad_bcast<1>(rid_v)
                                                   : s == 2 ? quad_bcast<2>(rid_v) : quad_bcast<3>(rid_v);
                    // one queue entry per (window, list) so that a window's two walks run side by side
                    const uint32_t n_a = (uint32_t)__popcll(walk_a);
                    if (queued & me) {
                        const uint32_t rrel = MULTI ? rid_s - (uint32_t)r_first : TWO ? rid_s : 0u;
                        const uint32_t home = bk[s] & 0x7FFFFFFFu;
                        if (walk_a & me) walkq[qn + (uint32_t)__popcll(walk_a & (me - 1))] = make_uint4(klo[s], khi[s], home, rrel << 1);
                        if (walk_b & me) walkq[qn + n_a + (uint32_t)__popcll(walk_b & (me - 1))] = make_uint4(klo[s], khi[s], home, (rrel << 1) | 1u);
                    }
                    qn += n_a + (uint32_t)__popcll(walk_b);
                    TBK_COUNT(3, __popcll(queued));
                }
            }
        }
        if ((any_a | any_b | full_any) != 0) {
            if (TWO) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    acc_a += (uint32_t)__popcll(hit_a[s] & own1[s]); acc2_a += (uint32_t)__popcll(hit_a[s] & ~own1[s]);
                    acc_b += (uint32_t)__popcll(hit_b[s] & own1[s]); acc2_b += (uint32_t)__popcll(hit_b[s] & ~own1[s]);
                }
            } else if (!MULTI) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    acc_a += (uint32_t)__popcll(hit_a[s]);
                    acc_b += (uint32_t)__popcll(hit_b[s]);
                }
            } else {
                // Bring each window's verdict to the bit of the lane that owns it (quad q's sub-step s
                // window belongs to lane 4q + s) and let every lane count its own windows: its 32
                // windows nearly always lie in one read, so the hits travel to the tallies once per
                // read and lane, not once per hit.
                uint64_t wa = guests_a, wb = guests_b;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    wa |= quad_any(hit_a[s]) << s;
                    wb |= quad_any(hit_b[s]) << s;
                }
                lane_a += (uint32_t)(wa >> lane) & 1u;
                lane_b += (uint32_t)(wb >> lane) & 1u;
            }
        }
        }  // !FRONT
        if (FRONT) {
            if (qb > TBK_BQCAP - 64) {  // make room for the next step's worst case
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                drain_back<MULTI || TWO>(p, backq, qb, walkq, qn, r_first, lane, acc_a, acc_b, rcnt);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                qb = 0;
            }
        } else if (qn > TBK_QCAP - 128) {  // make room for the next step's worst case
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            drain_walks<MULTI || TWO, FRONT>(p, walkq, qn, r_first, lane, acc_a, acc_b, rcnt);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            qn = 0;
        }
    }
    if (FRONT && qb) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        drain_back<MULTI || TWO>(p, backq, qb, walkq, qn, r_first, lane, acc_a, acc_b, rcnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (qn) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        drain_walks<MULTI || TWO, FRONT>(p, walkq, qn, r_first, lane, acc_a, acc_b, rcnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
#ifdef TBK_COUNTERS
    if (lane == 0)
        for (int i = 0; i < 8; i++) if (dbg[i]) atomicAdd(&tbk_dbg[i], dbg[i]);
#endif
    if (MULTI) {
        if (lane_a) count_hits(p, rcnt, r_first, (uint32_t)(rid - r_first), 0, lane_a);
        if (lane_b) count_hits(p, rcnt, r_first, (uint32_t)(rid - r_first), 1, lane_b);
        // flush the per-read tallies: lane l owns read r_first + l
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t ca = rcnt[2 * lane], cb = rcnt[2 * lane + 1];
        if (ca) { atomicAdd(&p.counts[2 * (r_first + lane)], (int)ca); rcnt[2 * lane] = 0; }
        if (cb) { atomicAdd(&p.counts[2 * (r_first + lane) + 1], (int)cb); rcnt[2 * lane + 1] = 0; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (TWO) {
        // the two reads' counts: what the window loop counted in scalars plus what the drains counted into the tallies
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < 4) {
            const uint32_t c = rcnt[lane] + (lane == 0 ? acc_a : lane == 1 ? acc_b : lane == 2 ? acc2_a : acc2_b);
            if (c) atomicAdd(&p.counts[2 * r_first + lane], (int)c);  // counts[read][hap]: (r, A) (r, B) (r + 1, A) (r + 1, B)
            rcnt[lane] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else if (!MULTI) {
        if (lane == 0) {
            if (acc_a) atomicAdd(&p.counts[2 * r_first], (int)acc_a);
            if (acc_b) atomicAdd(&p.counts[2 * r_first + 1], (int)acc_b);
        }
    }
}

// =======================================================================================
// probe, entry layout (tbk_common.h "entry layout")
// =======================================================================================
// The pass of the key layouts with three differences.  (1) A window asks with (canonical m-mer, flank bits, mask)
// instead of its canonical k-mer: the orientation is the sampled m-mer's, which the sampling arithmetic has at hand.
// (2) The probe is PAIR-cooperative: the front of a line is 32 bytes - four slots, or two 16-byte pieces of wide
// entries - that both lists fill in order; each lane of a pair holds half of it, and a window step has two sub-steps
// instead of four - half the broadcasts, compares and scalar ballot work of the quad probe, and 8 registers of line
// data per lane instead of 16.  One wave instruction touches 32 lines.  (3) A slot is compared under the window's mask:
// expected = (m-mer, bfi(mask, flanks, slot.hi)), one v_bfi and one 64-bit compare; which list a hit counts for is a
// bit of the entry, read only in steps that hit.  Flags are bits (bit 63 of the front's last slot: entries behind it).
// A window that misses in a front with entries behind it is queued; drain_back_entry settles eight at a time from the
// line's other 96 bytes (six lanes x 16 bytes, an L2 hit), and only a line that is full and was left by an entry
// (bit 63 of its last slot) sends the window on to the walk.
template <int S>
__device__ __forceinline__ uint32_t pair_bcast(uint32_t v) {
    // DPP quad_perm:[S, S, 2 + S, 2 + S]: lane S of each pair
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, S | (S << 2) | ((2 + S) << 4) | ((2 + S) << 6), 0xF, 0xF, true);
}
__device__ __forceinline__ uint64_t pair_any(uint64_t m) { return (m | (m >> 1)) & 0x5555555555555555ull; }

// walk queue entry: x = m-mer, y = flank bits + V bit, z = home bucket, w = read
// back queue entry: x = m-mer, y = flank bits + V bit, z = home bucket | flags << 30, w = read
__device__ __forceinline__ TbkEntryKey entry_key_of(uint32_t cm, uint32_t khi, int w, int fbits, int vshift) {
    const uint32_t v = khi >> vshift;                         // exactly one V bit
    const int pos = 31 - (int)__clz(v);
    TbkEntryKey e;
    e.cm = cm; e.khi = khi;
    e.mhi = (((1u << fbits) - 1u) << (2 * (w - 1 - pos))) | (1u << (vshift + pos));
    return e;
}

// wide entries (tbk_common.h): the queues carry 48 bits of m-mer and 48 bits of flanks + V - walk / back entry: x = m-mer low
// word, y = k1 low word, z = k1 bits 32..47 | m-mer bits 32..47 << 16, w = home bucket (| flags << 30); the read of a queued
// window (multi-read passes) in a 16-bit array beside the queue
__device__ __forceinline__ TbkWideKey wide_key_of(uint4 it, int w, int fbits, int vshift) {
    TbkWideKey e;
    e.cm = (uint64_t)it.x | ((uint64_t)(it.z >> 16) << 32);
    e.k1 = (uint64_t)it.y | ((uint64_t)(it.z & 0xFFFFu) << 32);
    const uint32_t v = (uint32_t)(e.k1 >> vshift);            // exactly one V bit (none in a filler entry: then nothing matches)
    const int pos = v ? 31 - (int)__clz(v) : 0;
    e.m1 = (((1ull << fbits) - 1ull) << (2 * (w - 1 - pos))) | (1ull << (vshift + pos));
    return e;
}

// one walk per window over the slots / pieces both lists share (`half` unused), *hap = the list of the entry found
template <bool WIDE, class KEY>
__device__ __forceinline__ bool walk_one_entry(const TbkPairView t, uint32_t half, KEY e, uint32_t bucket, bool pend, uint32_t *hap) {
    bool found = false, first = true;
    uint32_t guard = 0;
    *hap = 0;
    while (ballot(pend) != 0 && guard++ <= t.n_buckets) {
        if (pend) {
            if constexpr (WIDE) bucket = tbk_wentry_next_bucket(e.cm, t.n_buckets, bucket, first);
            else bucket = tbk_entry_next_bucket(e.cm, t.n_buckets, bucket, first);
            first = false;
            const uint64_t *line = t.slots + (uint64_t)bucket * 16;
            bool hit = false, ended = false;
            uint64_t last = 0;
            if constexpr (WIDE) {
#pragma unroll 1
                for (uint32_t pc = 0; pc < 8 && !ended && !hit; pc++) {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(line + 2 * pc);
                    if (tbk_wentry_match(v.x, v.y, e)) { hit = true; *hap = (uint32_t)(v.y >> 62) & 1u; }
                    ended = (v.x & TBK_WENTRY_TAKEN) == 0;
                    last = v.y;
                }
            } else {
#pragma unroll 1
                for (uint32_t sl = 0; sl < 16 && !ended && !hit; sl += 2) {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(line + sl);
                    if (tbk_entry_match(v.x, e)) { hit = true; *hap = (uint32_t)(v.x >> 62) & 1u; }
                    else if (tbk_entry_match(v.y, e)) { hit = true; *hap = (uint32_t)(v.y >> 62) & 1u; }
                    ended = (v.y << 1) == 0;  // an empty slot: the line's entries end here
                    last = v.y;
                }
            }
            found = found || hit;
            pend = !hit && !ended && (last >> 63) != 0;  // all taken and an entry went past them
        }
    }
    return found;
}

// full keys (KIND = 3): the walk of a window whose home line is full and was left by a key - second-choice bucket by a hash of the
// canonical key, then linear (tbk_next_bucket); whole lines, no summary test.  q = the stored form asked (tbk_full_word).
__device__ __forceinline__ bool walk_one_full(const TbkPairView t, uint64_t q, uint32_t bucket, bool pend, uint32_t *hap) {
    bool found = false, first = true;
    uint32_t guard = 0;
    *hap = 0;
    const uint64_t canonical = ~q & TBK_FULL_KEY;
    while (ballot(pend) != 0 && guard++ <= t.n_buckets) {
        if (pend) {
            bucket = tbk_next_bucket(canonical, t.mz, t.n_buckets, bucket, first);
            first = false;
            const uint64_t *line = t.slots + (uint64_t)bucket * 16;
            bool hit = false, ended = false;
            uint64_t last = 0;
#pragma unroll 1
            for (uint32_t sl = 0; sl < 16 && !ended && !hit; sl += 2) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(line + sl);
                if ((v.x & TBK_FULL_KEY) == q) { hit = true; *hap = (uint32_t)(v.x >> 62) & 1u; }
                else if (sl + 1 != TBK_FULL_SUMMARY && (v.y & TBK_FULL_KEY) == q) { hit = true; *hap = (uint32_t)(v.y >> 62) & 1u; }
                ended = sl + 1 == TBK_FULL_SUMMARY ? (v.y >> 63) == 0 : (v.y << 1) == 0;  // nothing behind the front / an empty slot: the line's keys end here
                last = v.y;
            }
            found = found || hit;
            pend = !hit && !ended && (last >> 63) != 0;  // all taken and a key went past them
        }
    }
    return found;
}

template <bool MULTI, int KIND>
__device__ __forceinline__ void drain_walks_entry(const ProbeArgs &p, const uint4 *q, const uint16_t *qr, uint32_t qn, uint64_t r_first, uint32_t lane,
                                                  int fbits, int vshift, uint32_t &acc_a, uint32_t &acc_b, uint32_t *rcnt) {
    constexpr bool WIDE = KIND == 1, FULL = KIND == 3;
    for (uint32_t base = 0; base < qn; base += 64) {
        const bool act = base + lane < qn;
        uint4 it = WIDE ? make_uint4(0, 0, 0, 0) : FULL ? make_uint4(1, 0, 0, 0) : make_uint4(TBK_ENTRY_NO_MMER, 1u << vshift, 0, 0);  // (a wide filler has no V bit, a full one asks for no key: they match nothing)
        uint32_t rrel = 0;
        if (act) { it = q[base + lane]; if (WIDE && MULTI) rrel = qr[base + lane]; }
        bool found;
        uint32_t list = 0;
        if constexpr (WIDE) {
            found = walk_one_entry<true>(p.t, 0, wide_key_of(it, p.t.mz.w, fbits, vshift), it.w & 0x3FFFFFFFu, act, &list);
        } else if constexpr (FULL) {
            rrel = it.w;
            found = walk_one_full(p.t, (uint64_t)it.x | ((uint64_t)it.y << 32), it.z, act, &list);
        } else {
            rrel = it.w;
            found = walk_one_entry<false>(p.t, 0, entry_key_of(it.x, it.y, p.t.mz.w, fbits, vshift), it.z, act, &list);
        }
        const bool count_a = found && !list, count_b = found && list;
        if (!MULTI) {
            acc_a += (uint32_t)__popcll(ballot(count_a));
            acc_b += (uint32_t)__popcll(ballot(count_b));
        } else {
            if (count_a) count_hits(p, rcnt, r_first, rrel, 0, 1);
            if (count_b) count_hits(p, rcnt, r_first, rrel, 1, 1);
        }
    }
}

// eight queued windows at a time, eight lanes per window: lanes 0..5 hold the six 16-byte pieces behind the front of the home
// line (its bytes 32..127: two slots of the narrow layout, one wide entry each), lanes 6 and 7 nothing
// (short keys, KIND = 2: the queue entry is x = the word asked, y = home bucket | flags, z / w = the canonical k-mer; the six
// pieces are four slots each; a window that misses in a line whose last slot is flagged asks the overflow table, here)
template <bool MULTI, int KIND>
__device__ __forceinline__ void drain_back_entry(const ProbeArgs &p, const uint4 *bq, const uint16_t *bqr, uint32_t qb, uint4 *walkq, uint16_t *walkr, uint32_t &qn,
                                                 uint64_t r_first, uint32_t lane, int fbits, int vshift, uint32_t &acc_a, uint32_t &acc_b, uint32_t *rcnt) {
    constexpr bool WIDE = KIND == 1, SHORT = KIND == 2, FULL = KIND == 3;
    const uint32_t sub = lane & 7u, oct = lane >> 3;
    for (uint32_t base = 0; base < qb; base += 8) {
        const bool act = base + oct < qb;
        uint4 it = SHORT ? make_uint4(TBK_SHORT_NONE, 0, 0, 0) : WIDE ? make_uint4(0, 0, 0, 0) : FULL ? make_uint4(1, 0, 0, 0) : make_uint4(TBK_ENTRY_NO_MMER, 1u << vshift, 0, 0);
        uint32_t rrel = 0;
        ulonglong2 v = make_ulonglong2(0, 0);
        if (act) {
            it = bq[base + oct];
            if ((WIDE || SHORT) && MULTI) rrel = bqr[base + oct];
            if (sub < 6) v = load_slots(p.t.slots + (uint64_t)((SHORT ? it.y : WIDE ? it.w : it.z) & 0x3FFFFFFFu) * 16 + 4 + sub * 2);
        }
        uint64_t hit, hit_a, hit_b;
        if constexpr (SHORT) {
            const uint32_t qa = it.x, qb2 = it.x | TBK_SHORT_HAPB;
            const uint32_t w0 = (uint32_t)v.x, w1 = (uint32_t)(v.x >> 32), w2 = (uint32_t)v.y, w3 = (uint32_t)(v.y >> 32) & ~TBK_SHORT_FLAG;
            hit_a = ballot(w0 == qa || w1 == qa || w2 == qa || w3 == qa);
            hit_b = ballot(w0 == qb2 || w1 == qb2 || w2 == qb2 || w3 == qb2);
            hit = hit_a | hit_b;
        } else if constexpr (WIDE) {
            // the six pieces behind the front belong to whichever list came: the entry's bit 62 says which
            hit = ballot(tbk_wentry_match(v.x, v.y, wide_key_of(it, p.t.mz.w, fbits, vshift)));
            const uint64_t hapm = ballot(((v.y >> 62) & 1ull) != 0);
            hit_a = hit & ~hapm; hit_b = hit & hapm;
        } else if constexpr (FULL) {
            // slots 4 .. 15: keys of either list, the list in bit 62
            rrel = it.w;
            const uint64_t q = (uint64_t)it.x | ((uint64_t)it.y << 32);
            const uint64_t hx = ballot((v.x & TBK_FULL_KEY) == q), hy = ballot((v.y & TBK_FULL_KEY) == q);
            const uint64_t bx = ballot(((v.x >> 62) & 1ull) != 0), by = ballot(((v.y >> 62) & 1ull) != 0);
            hit = hx | hy;
            hit_a = (hx & ~bx) | (hy & ~by); hit_b = (hx & bx) | (hy & by);
        } else {
            rrel = it.w;
            const TbkEntryKey e = entry_key_of(it.x, it.y, p.t.mz.w, fbits, vshift);
            const uint64_t hx = ballot(tbk_entry_match(v.x, e)), hy = ballot(tbk_entry_match(v.y, e));
            const uint64_t bx = ballot(((v.x >> 62) & 1ull) != 0), by = ballot(((v.y >> 62) & 1ull) != 0);
            hit = hx | hy;
            hit_a = (hx & ~bx) | (hy & ~by); hit_b = (hx & bx) | (hy & by);
        }
        // lane 5 holds the line's last slots / piece: bit 63 of its second word = an entry went past this line
        const uint64_t gone = ballot((v.y >> 63) != 0);
        const uint64_t any = hit_a | hit_b;
        const uint64_t oct_hit = (any | (any >> 1) | (any >> 2) | (any >> 3) | (any >> 4) | (any >> 5)) & 0x0101010101010101ull;
        const uint64_t walk_a = ((gone >> 5) & 0x0101010101010101ull) & ~oct_hit;  // one walk per window, over both lists' entries
        const uint64_t walk_b = 0ull;
        if (!MULTI) {
            acc_a += (uint32_t)__popcll(hit_a);
            acc_b += (uint32_t)__popcll(hit_b);
        } else {
            if ((hit_a >> lane) & 1ull) count_hits(p, rcnt, r_first, rrel, 0, 1);
            if ((hit_b >> lane) & 1ull) count_hits(p, rcnt, r_first, rrel, 1, 1);
        }
        if constexpr (SHORT) {
            // the line is full and a key of its bucket went to the overflow table: the window's first lane asks there
            if (walk_a != 0 && p.t.over_mask != 0) {
                int found = -1;
                if ((walk_a >> lane) & 1ull) {
                    const uint64_t key = (uint64_t)it.z | ((uint64_t)it.w << 32);
                    const uint64_t *over = p.t.slots + (uint64_t)p.t.n_buckets * 16;
                    uint32_t i = tbk_short_over_home(key, p.t.over_mask);
                    for (uint32_t walked = 0; walked <= p.t.over_mask; walked++, i = (i + 1) & p.t.over_mask) {
                        const uint64_t o = over[i];
                        if (o == TBK_SHORT_EMPTY64) break;
                        if ((o & ~(1ull << 63)) == key) { found = (int)(o >> 63); break; }
                    }
                }
                if (!MULTI) {
                    acc_a += (uint32_t)__popcll(ballot(found == 0));
                    acc_b += (uint32_t)__popcll(ballot(found == 1));
                } else if (found >= 0) {
                    count_hits(p, rcnt, r_first, rrel, (uint32_t)found, 1);
                }
            }
            continue;
        }
        const uint64_t queued = walk_a | walk_b;
        if (queued) {
            const uint64_t me = 1ull << lane;
            const uint32_t n_a = (uint32_t)__popcll(walk_a);
            const uint32_t at_a = qn + (uint32_t)__popcll(walk_a & (me - 1));
            if constexpr (WIDE) {
                if (walk_a & me) { walkq[at_a] = make_uint4(it.x, it.y, it.z, it.w & 0x3FFFFFFFu); if (MULTI) walkr[at_a] = (uint16_t)rrel; }
            } else {
                if (walk_a & me) walkq[at_a] = make_uint4(it.x, it.y, it.z & 0x3FFFFFFFu, it.w);
            }
            qn += n_a + (uint32_t)__popcll(walk_b);
            if (qn > TBK_QCAP_ENTRY - 16) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if constexpr (!SHORT) drain_walks_entry<MULTI, KIND>(p, walkq, walkr, qn, r_first, lane, fbits, vshift, acc_a, acc_b, rcnt);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                qn = 0;
            }
        }
    }
}

// LW: the t-mer positions of a span in units of W - 2 (t = m - w) or 3 (t = m - 2w: tbk_mz_span3)
template <int W, bool MULTI, bool TWO, int KIND, int LW>
__device__ __forceinline__ void probe_pass_entry(const ProbeArgs &p, const uint64_t e0, const uint64_t e1, const uint64_t e2, const uint64_t e3,
                                                 const uint64_t P0, const uint64_t r_first, const uint64_t r_first_end, const uint32_t lane,
                                                 uint4 *walkq, uint4 *backq, uint16_t *walkr, uint16_t *backr, uint32_t *rcnt, const uint32_t *tlut) {
    constexpr bool WIDE = KIND == 1, SHORT = KIND == 2, FULL = KIND == 3;
    constexpr bool TLUT = TBK_TMER_LUT && LW == 3 && W == 6 && !TWO && !MULTI;  // t = 4: a t-mer's rank is a table look-up (tbk_probe_entry_kernel fills the table; the two-read and multi-read kernels, at their register limits, compute)
    const int k = p.k;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t sub = lane & 1u;
    // streams, read ends and the odd lanes' downward walk: as in probe_pass
    unsigned __int128 S128 = (unsigned __int128)(uint32_t)e0 | ((unsigned __int128)(uint32_t)e1 << 32) |
                             ((unsigned __int128)(uint32_t)e2 << 64) | ((unsigned __int128)(uint32_t)e3 << 96);
    unsigned __int128 R128 = (unsigned __int128)rev_pairs(~(uint32_t)e3) | ((unsigned __int128)rev_pairs(~(uint32_t)e2) << 32) |
                             ((unsigned __int128)rev_pairs(~(uint32_t)e1) << 64) | ((unsigned __int128)rev_pairs(~(uint32_t)e0) << 96);
    uint64_t bad64 = (uint64_t)((uint32_t)(e0 >> 32) | ((uint32_t)(e1 >> 32) << 16)) |
                     ((uint64_t)((uint32_t)(e2 >> 32) | ((uint32_t)(e3 >> 32) << 16)) << 32);
    bool is_second = false, is_strad = false;
    if (TWO) {
        const uint64_t p_first = P0 + (uint64_t)lane * TBK_WPL;
        const uint64_t r2_end = p.offsets[r_first + 2];
        is_second = p_first >= r_first_end;
        const uint64_t inside = is_second ? 64 : r_first_end - p_first;
        is_strad = inside < TBK_WPL;
        if (is_second || is_strad) {
            const uint64_t inside2 = r2_end > p_first ? r2_end - p_first : 0;
            if (inside2 < 64) bad64 |= ~0ull << inside2;
        } else if (inside < 64) {
            bad64 |= ~0ull << inside;
        }
    }
    if (!MULTI) {
        if (!TWO) {
            const uint64_t p_first = P0 + (uint64_t)lane * TBK_WPL;
            const uint64_t inside = r_first_end > p_first ? r_first_end - p_first : 0;
            if (inside < 64) bad64 |= ~0ull << inside;
        }
        if ((lane & 1u) && !(TWO && is_strad)) {
            const int sh = 33 - k;
            const unsigned __int128 s_up = R128 >> (2 * sh), r_up = S128 << (2 * sh);
            S128 = s_up; R128 = r_up;
            bad64 = __brevll(bad64) >> sh;
        }
    }
    uint32_t s0 = (uint32_t)S128, s1 = (uint32_t)(S128 >> 32), s2 = (uint32_t)(S128 >> 64), s3 = (uint32_t)(S128 >> 96);
    uint32_t t0, t1, t2, t3;
    {
        const unsigned __int128 Rs = R128 >> (64 - 2 * k);
        t0 = (uint32_t)Rs; t1 = (uint32_t)(Rs >> 32); t2 = (uint32_t)(Rs >> 64); t3 = (uint32_t)(Rs >> 96);
    }
    uint32_t bad_lo = (uint32_t)bad64, bad_hi = (uint32_t)(bad64 >> 32);
    const uint32_t badk = k == 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);

    // mod-sampling state (probe_pass, SAMP): ranks of the span's LW x W t-mers, tagged with their stream index mod 16 (32)
    constexpr int NW = LW * W;
    constexpr uint32_t TAGM = NW > 16 ? 31u : 15u;
    uint32_t win[NW];
    const int m = p.t.mz.m, o = p.t.mz.o, tlen = p.t.mz.t;
    const uint32_t span_o = (uint32_t)o;
    const uint64_t mmask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
    const uint32_t tmask = tlen >= 16 ? 0xFFFFFFFFu : ((1u << (2 * tlen)) - 1u);
    auto tmer_rank = [&](uint64_t fwd64, uint64_t rc64, uint32_t fsh, uint32_t bsh, uint32_t pos) -> uint32_t {
        if constexpr (TLUT) return tlut[(uint32_t)(fwd64 >> fsh) & 0xFFu] | pos;  // (canonical form, hash and tag mask are in the table)
        const uint32_t x = (uint32_t)(fwd64 >> fsh) & tmask, y = (uint32_t)(rc64 >> bsh) & tmask;
        return (tbk_mmer_hash(x < y ? x : y) & ~TAGM) | pos;
    };
    {
        const uint64_t fs = ((uint64_t)s1 << 32) | s0, bs = ((uint64_t)t3 << 32) | t2;
        win[0] = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i + 1 < NW; i++) win[i + 1] = tmer_rank(fs, bs, (uint32_t)(2 * (o + i)), (uint32_t)(2 * (o + NW - 1 - i)), (uint32_t)i);
    }
    const uint32_t fsh_new = (uint32_t)(2 * (o + NW - 1)), bsh_new = (uint32_t)(2 * o);
    // entry geometry
    const int fl = o + W - 1, fbits = 2 * (k - m), vshift = 4 * fl;
    const uint32_t fmask = fbits >= 32 ? 0xFFFFFFFFu : ((1u << fbits) - 1u);
    const uint32_t rshift = 31u - (uint32_t)__clz(p.t.n_buckets);  // short keys (tbk_short_geom)

    // read bookkeeping (probe_pass)
    const uint64_t p_lane = P0 + (uint64_t)lane * TBK_WPL;
    uint64_t rid = r_first;
    uint64_t rend = r_first_end;
    if (MULTI) {
        const uint64_t pl = p_lane < p.total ? p_lane : p.total;
        uint64_t lo = r_first, step = 1;
        while (lo + step <= p.n_reads && p.offsets[lo + step] <= pl) { lo += step; step <<= 1; }
        uint64_t hi = lo + step <= p.n_reads ? lo + step : p.n_reads + 1;
        while (hi - lo > 1) {
            const uint64_t mid = lo + ((hi - lo) >> 1);
            if (p.offsets[mid] <= pl) lo = mid; else hi = mid;
        }
        rid = lo;
        rend = rid < p.n_reads ? p.offsets[rid + 1] : p.total;
    }
    auto rel = [&](uint64_t pos) -> uint32_t {
        return pos > p_lane ? (uint32_t)(pos - p_lane < 0x40000000ull ? pos - p_lane : 0x40000000ull) : 0u;
    };
    uint32_t rel_end = rel(rend);
    uint32_t acc_a = 0, acc_b = 0, lane_a = 0, lane_b = 0, acc2_a = 0, acc2_b = 0;
    uint64_t own1[2] = {0, 0};   // TWO: per sub-step the lanes (both of a pair) whose pair's window belongs to the pass's first read
    uint32_t jb_s = 0xFFFFu;
    if (TWO) {
        const uint32_t brel = (uint32_t)(r_first_end - P0);
        if (brel % TBK_WPL) jb_s = brel % TBK_WPL;
#pragma unroll
        for (int s = 0; s < 2; s++) own1[s] = ballot(((lane & ~1u) + (uint32_t)s) * TBK_WPL < brel);
    }
    ulonglong2 va[2];
    va[0] = make_ulonglong2(0, 0); va[1] = make_ulonglong2(0, 0);
    uint32_t last_bk = 0x7FFFFFFFu;
    uint32_t qn = 0, qb = 0;

    constexpr int UNROLL = KIND == 3 ? TBK_FULL_UNROLL : TBK_SAMP_UNROLL;
#pragma unroll UNROLL
    for (int j = 0; j < TBK_WPL; j++) {
        const uint64_t fs = ((uint64_t)s1 << 32) | s0, bs = ((uint64_t)t3 << 32) | t2;
        if (MULTI) {
            while ((uint32_t)j >= rel_end && rid < p.n_reads) {
                if (lane_a) { count_hits(p, rcnt, r_first, (uint32_t)(rid - r_first), 0, lane_a); lane_a = 0; }
                if (lane_b) { count_hits(p, rcnt, r_first, (uint32_t)(rid - r_first), 1, lane_b); lane_b = 0; }
                rid++;
                rend = rid < p.n_reads ? p.offsets[rid + 1] : p.total;
                rel_end = rel(rend);
            }
        }
        if (TWO && (uint32_t)j == jb_s) {
            const uint64_t strad = ballot(is_strad);  // one lane: its pair changes sides in the sub-step that lane owns
#pragma unroll
            for (int s = 0; s < 2; s++) if (strad & (0x5555555555555555ull << s)) own1[s] &= ~(pair_any(strad) * 3ull);  // (both bits of that pair)
        }
        bool ok = (bad_lo & badk) == 0;
        if (TWO) {
            const bool cross = (uint32_t)j < jb_s && (uint32_t)(j + k) > jb_s;
            ok = ok && !(cross && is_strad);
        }
        if (MULTI) ok = ok && (uint32_t)(j + k) <= rel_end && rid < p.n_reads;
        // ---- sampling: newest t-mer in, smallest rank, its position, the m-mer there on both strands ----
#pragma unroll
        for (int i = 0; i + 1 < NW; i++) win[i] = win[i + 1];
        win[NW - 1] = tmer_rank(fs, bs, fsh_new, bsh_new, (uint32_t)(j + NW - 1) & TAGM);
        uint32_t best = win[0];
#pragma unroll
        for (int i = 1; i < NW; i++) best = win[i] < best ? win[i] : best;
        const uint32_t x = (best - (uint32_t)j) & TAGM;
        const uint32_t x1 = LW == 3 && x >= 2u * (uint32_t)W ? x - 2u * (uint32_t)W : x;
        const uint32_t pos = x1 >= (uint32_t)W ? x1 - (uint32_t)W : x1;
        const uint32_t fsh = 2u * (span_o + pos), bsh = 2u * (span_o + (uint32_t)W - 1u - pos);
        // (wide entries: m-mers of up to 24 bases, 64-bit arithmetic and the placement half of the 64-bit hash)
        using mmer_t = typename std::conditional<WIDE || FULL, uint64_t, uint32_t>::type;  // (full keys: the wide entries' m-mer arithmetic and bucket hash)
        const mmer_t mx = (mmer_t)(fs >> fsh) & (mmer_t)mmask, my = (mmer_t)(bs >> bsh) & (mmer_t)mmask;
        const bool fw_or = mx < my;                       // the m-mer is canonical as the forward strand reads it
        const mmer_t cm = fw_or ? mx : my;
        uint32_t bkt, short_r = 0;
        if constexpr (WIDE || FULL) bkt = tbk_reduce((uint32_t)tbk_mmer_hash64(cm), p.t.n_buckets);
        else if constexpr (SHORT) {
            const uint64_t prod = (uint64_t)tbk_mmer_hash((uint32_t)cm) * (uint64_t)p.t.n_buckets;  // (tbk_short_key: bucket and r are the two words of one product)
            bkt = (uint32_t)(prod >> 32);
            short_r = (uint32_t)prod >> rshift;
        } else bkt = tbk_reduce(tbk_mmer_hash(cm), p.t.n_buckets);
        // ---- what this window asks an entry (tbk_entry_key): the k-mer in the m-mer's orientation, cut around the m-mer ----
        const uint64_t orient = (fw_or ? fs : bs) & kmask;
        const uint32_t a = fw_or ? fsh : bsh;             // 2 (o + pos'), pos' = the m-mer's position as `orient` reads
        const uint32_t low = (uint32_t)orient & ((1u << a) - 1u);
        const uint32_t high = (uint32_t)(orient >> (a + 2u * (uint32_t)m));
        const uint32_t posp = (a >> 1) - span_o;
        const uint32_t shw = 2u * ((uint32_t)W - 1u - posp);
        const uint32_t vbit = WIDE ? 0u : 1u << ((uint32_t)vshift + posp);
        // narrow entries: 32 bits of flanks + V; wide entries: the same field 64 bits wide (the V bit lies in its upper word)
        uint32_t my_khi, my_mhi, my_khi2 = 0, my_mhi2 = 0;
        if constexpr (WIDE) {
            const uint64_t k1 = ((uint64_t)(low | (high << a)) << shw) | (1ull << ((uint32_t)vshift + posp));
            const uint64_t m1 = ((uint64_t)fmask << shw) | (1ull << ((uint32_t)vshift + posp));
            my_khi = (uint32_t)k1; my_khi2 = (uint32_t)(k1 >> 32);
            my_mhi = (uint32_t)m1; my_mhi2 = (uint32_t)(m1 >> 32);
        } else if constexpr (SHORT) {
            my_khi = (low | (high << a)) | (posp << (uint32_t)fbits) | (short_r << ((uint32_t)fbits + 3u)) | TBK_SHORT_TAKEN;  // the word this window asks
            my_mhi = 0;
        } else if constexpr (FULL) {
            my_khi = 0; my_mhi = 0;  // (set with cm_ask below: the stored form of the canonical k-mer)
        } else {
            my_khi = ((low | (high << a)) << shw) | vbit;
            my_mhi = (fmask << shw) | vbit;
        }
        // an invalid window asks for an m-mer no entry holds (narrow: T x 16, never canonical; wide: a word 0 no piece can hold)
        uint32_t cm_ask, cm_ask2 = 0;
        if constexpr (WIDE) {
            const uint64_t w0_want = ok ? ((uint64_t)cm | TBK_WENTRY_TAKEN) : ~0ull;
            cm_ask = (uint32_t)w0_want; cm_ask2 = (uint32_t)(w0_want >> 32);
        } else if constexpr (SHORT) {
            cm_ask = ok ? my_khi : TBK_SHORT_NONE;   // (by arithmetic - my_khi | -(min(bad bits, 1)) - measured: nothing; EXPERIMENTS.md "What a select costs")
        } else if constexpr (FULL) {
            // what the slots hold: the canonical k-mer, inverted (tbk_full_word); an invalid window asks for the inverse of TBK_FULL_NOKEY
            const uint64_t kf = fs & kmask, kr = bs & kmask;
            const uint64_t q = ok ? (~(kf < kr ? kf : kr) & TBK_FULL_KEY) : 1ull;
            cm_ask = (uint32_t)q; my_khi = (uint32_t)(q >> 32);
        } else {
            cm_ask = ok ? (uint32_t)cm : TBK_ENTRY_NO_MMER;
        }
        const bool fresh = ok && bkt != last_bk;
        // (one select: a valid window that is not fresh names the bucket the lane holds already.  A select through VCC - v_cndmask_b32
        // in its 32-bit encoding - costs this chip 2 to 5 ordinary vector instructions: profiles/r05/valu_rates_select.log)
        const uint32_t my_bk = fresh ? (bkt | 0x80000000u) : last_bk;
        last_bk = my_bk & 0x7FFFFFFFu;
        const uint32_t my_rid = (uint32_t)rid;
        auto two_rid = [&]() -> uint32_t { return (is_second || (is_strad && (uint32_t)j >= jb_s)) ? 1u : 0u; };

        // advance to window j + 1
        s0 = (s0 >> 2) | (s1 << 30); s1 = (s1 >> 2) | (s2 << 30); s2 = (s2 >> 2) | (s3 << 30); s3 >>= 2;
        t3 = (t3 << 2) | (t2 >> 30); t2 = (t2 << 2) | (t1 >> 30); t1 = (t1 << 2) | (t0 >> 30); t0 <<= 2;
        bad_lo = (bad_lo >> 1) | (bad_hi << 31); bad_hi >>= 1;

        // ---- two pair sub-steps ----
        const uint32_t bk0 = pair_bcast<0>(my_bk), bk1 = pair_bcast<1>(my_bk);
        if ((int32_t)bk0 < 0) va[0] = load_slots(p.t.slots + (uint64_t)(bk0 & 0x7FFFFFFFu) * 16 + sub * 2);
        if ((int32_t)bk1 < 0) va[1] = load_slots(p.t.slots + (uint64_t)(bk1 & 0x7FFFFFFFu) * 16 + sub * 2);
        const uint32_t cm_s[2] = {pair_bcast<0>(cm_ask), pair_bcast<1>(cm_ask)};
        const uint32_t kh_s[2] = {pair_bcast<0>(my_khi), pair_bcast<1>(my_khi)};
        const uint32_t mh_s[2] = {pair_bcast<0>(my_mhi), pair_bcast<1>(my_mhi)};
        uint32_t kh2_s[2] = {0, 0}, mh2_s[2] = {0, 0}, cm2_s[2] = {0, 0};
        if constexpr (WIDE) {
            kh2_s[0] = pair_bcast<0>(my_khi2); kh2_s[1] = pair_bcast<1>(my_khi2);
            mh2_s[0] = pair_bcast<0>(my_mhi2); mh2_s[1] = pair_bcast<1>(my_mhi2);
            cm2_s[0] = pair_bcast<0>(cm_ask2); cm2_s[1] = pair_bcast<1>(cm_ask2);
        }
        uint64_t hit[2], more[2], hitx[2] = {0, 0}, hit_sb[2] = {0, 0};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const uint32_t hx = (uint32_t)(va[s].x >> 32), hy = (uint32_t)(va[s].y >> 32);
            if constexpr (SHORT) {
                // the lane's 16 bytes are four slots of either list: the word asked with and without the list bit
                const uint32_t qa = cm_s[s], qb2 = cm_s[s] | TBK_SHORT_HAPB, w0 = (uint32_t)va[s].x, w2 = (uint32_t)va[s].y, w3 = hy & ~TBK_SHORT_FLAG;
                hitx[s] = ballot(w0 == qa || hx == qa || w2 == qa || w3 == qa);
                hit_sb[s] = ballot(w0 == qb2 || hx == qb2 || w2 == qb2 || w3 == qb2);
                hit[s] = hitx[s] | hit_sb[s];
            } else if constexpr (FULL) {
                // the even lane holds slots 0 and 1, the odd lane slot 2 and - never compared: a 62-bit summary could equal a key - the line's summary
                const uint64_t want = (uint64_t)cm_s[s] | ((uint64_t)kh_s[s] << 32);
                hitx[s] = ballot((va[s].x & TBK_FULL_KEY) == want);
                hit[s] = hitx[s] | (ballot((va[s].y & TBK_FULL_KEY) == want) & 0x5555555555555555ull);
            } else if constexpr (WIDE) {
                // the lane's piece is ONE entry: word 0 = m-mer | taken, word 1 = flanks + V (+ the piece's flag in bit 63, outside every mask)
                const uint32_t ly = (uint32_t)va[s].y;
                const uint64_t want = (uint64_t)((kh_s[s] & mh_s[s]) | (ly & ~mh_s[s])) | ((uint64_t)((kh2_s[s] & mh2_s[s]) | (hy & ~mh2_s[s])) << 32);
                hit[s] = ballot(va[s].x == ((uint64_t)cm_s[s] | ((uint64_t)cm2_s[s] << 32))) & ballot(va[s].y == want);  // (word 0 = m-mer | taken, exactly: no lock in a finished table)
            } else {
                const uint64_t wx = (uint64_t)cm_s[s] | ((uint64_t)((kh_s[s] & mh_s[s]) | (hx & ~mh_s[s])) << 32);
                const uint64_t wy = (uint64_t)cm_s[s] | ((uint64_t)((kh_s[s] & mh_s[s]) | (hy & ~mh_s[s])) << 32);
                hitx[s] = ballot(va[s].x == wx);
                hit[s] = hitx[s] | ballot(va[s].y == wy);
            }
            more[s] = ballot((int32_t)hy < 0);  // bit 63 of the line's slot 3 / piece 1 (the odd lane's second word): entries behind the front
        }
        if ((more[0] | more[1]) != 0) {
            // windows that missed in a front with entries behind it: queued by the lane that owns the window
            uint64_t need = 0, beh_a = 0, beh_b = 0;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (more[s] == 0) continue;
                // only the odd lane's second word (slot 3 / piece 1) carries the flag, for the line
                uint64_t flagged = more[s];
                if constexpr (SHORT) {
                    // ... and that word is the line's summary: behind the front only if the window's own bit is set in it
                    const uint32_t summary = (uint32_t)(va[s].y >> 32);
                    flagged = ballot((int32_t)summary < 0 && (summary & tbk_short_filter_bit(cm_s[s])) != 0);
                }
                if constexpr (FULL) {
                    const uint64_t summary = va[s].y;  // (the odd lane's: slot 3)
                    flagged = ballot((summary >> 63) != 0 && (summary & tbk_full_filter_bit((uint64_t)cm_s[s] | ((uint64_t)kh_s[s] << 32))) != 0);
                }
                const uint64_t ma = 0ull, mb = (flagged >> 1) & 0x5555555555555555ull;
                need |= ((ma | mb) & ~pair_any(hit[s])) << s;
                beh_a |= ma << s;
                beh_b |= mb << s;
            }
            need &= ballot(ok);
            if (need) {
                const uint64_t me = 1ull << lane;
                if (need & me) {
                    const uint32_t rrel = MULTI ? my_rid - (uint32_t)r_first : TWO ? two_rid() : 0u;
                    const uint32_t at = qb + (uint32_t)__popcll(need & (me - 1));
                    const uint32_t home = last_bk | ((uint32_t)((beh_a >> lane) & 1ull) << 30) | ((uint32_t)((beh_b >> lane) & 1ull) << 31);
                    if constexpr (SHORT) {
                        const uint64_t kf = fs & kmask, kr = bs & kmask, ck = kf < kr ? kf : kr;  // the canonical k-mer: what the overflow table holds
                        backq[at] = make_uint4(my_khi, home, (uint32_t)ck, (uint32_t)(ck >> 32));
                        if (MULTI || TWO) backr[at] = (uint16_t)rrel;
                    } else if constexpr (WIDE) {
                        backq[at] = make_uint4((uint32_t)cm, my_khi, (my_khi2 & 0xFFFFu) | ((uint32_t)((uint64_t)cm >> 32) << 16), home);
                        if (MULTI || TWO) backr[at] = (uint16_t)rrel;
                    } else if constexpr (FULL) {
                        backq[at] = make_uint4(cm_ask, my_khi, home, rrel);  // (the stored form asked, home bucket, read: the narrow entries' places)
                    } else {
                        backq[at] = make_uint4((uint32_t)cm, my_khi, home, rrel);
                    }
                }
                qb += (uint32_t)__popcll(need);
            }
        }
        if ((hit[0] | hit[1]) != 0) {
            // which list a hit counts for is the entry's list bit (bit 62 of the slot / of a wide entry's second word)
            uint64_t ha[2], hb[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if constexpr (SHORT) {
                    ha[s] = hitx[s]; hb[s] = hit_sb[s];
                } else if constexpr (WIDE) {
                    const uint64_t bm = ballot(((uint32_t)(va[s].y >> 32) & 0x40000000u) != 0);
                    ha[s] = hit[s] & ~bm; hb[s] = hit[s] & bm;
                } else {
                    const uint64_t bx = ballot(((uint32_t)(va[s].x >> 32) & TBK_ENTRY_HAPB) != 0), by = ballot(((uint32_t)(va[s].y >> 32) & TBK_ENTRY_HAPB) != 0);
                    const uint64_t hy_ = hit[s] & ~hitx[s];  // (a window matches at most one slot of the line)
                    ha[s] = (hitx[s] & ~bx) | (hy_ & ~by); hb[s] = (hitx[s] & bx) | (hy_ & by);
                }
            }
            if (TWO) {
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    acc_a += (uint32_t)__popcll(ha[s] & own1[s]);
                    acc_b += (uint32_t)__popcll(hb[s] & own1[s]);
                    acc2_a += (uint32_t)__popcll(ha[s] & ~own1[s]);
                    acc2_b += (uint32_t)__popcll(hb[s] & ~own1[s]);
                }
            } else if (!MULTI) {
                acc_a += (uint32_t)__popcll(ha[0]) + (uint32_t)__popcll(ha[1]);
                acc_b += (uint32_t)__popcll(hb[0]) + (uint32_t)__popcll(hb[1]);
            } else {
                // to the bit of the lane that owns the window (pair p's sub-step s window belongs to lane 2p + s)
                const uint64_t wa = pair_any(ha[0]) | (pair_any(ha[1]) << 1);
                const uint64_t wb = pair_any(hb[0]) | (pair_any(hb[1]) << 1);
                lane_a += (uint32_t)(wa >> lane) & 1u;
                lane_b += (uint32_t)(wb >> lane) & 1u;
            }
        }
        if (qb > (SHORT ? (uint32_t)TBK_SHORT_DRAIN : (uint32_t)(TBK_BQCAP - 64))) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            drain_back_entry<MULTI || TWO, KIND>(p, backq, backr, qb, walkq, walkr, qn, r_first, lane, fbits, vshift, acc_a, acc_b, rcnt);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            qb = 0;
        }
    }
    if (qb) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        drain_back_entry<MULTI || TWO, KIND>(p, backq, backr, qb, walkq, walkr, qn, r_first, lane, fbits, vshift, acc_a, acc_b, rcnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (!SHORT && qn) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        drain_walks_entry<MULTI || TWO, KIND>(p, walkq, walkr, qn, r_first, lane, fbits, vshift, acc_a, acc_b, rcnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (MULTI) {
        if (lane_a) count_hits(p, rcnt, r_first, (uint32_t)(rid - r_first), 0, lane_a);
        if (lane_b) count_hits(p, rcnt, r_first, (uint32_t)(rid - r_first), 1, lane_b);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t ca = rcnt[2 * lane], cb = rcnt[2 * lane + 1];
        if (ca) { atomicAdd(&p.counts[2 * (r_first + lane)], (int)ca); rcnt[2 * lane] = 0; }
        if (cb) { atomicAdd(&p.counts[2 * (r_first + lane) + 1], (int)cb); rcnt[2 * lane + 1] = 0; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (TWO) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < 4) {
            const uint32_t c = rcnt[lane] + (lane == 0 ? acc_a : lane == 1 ? acc_b : lane == 2 ? acc2_a : acc2_b);
            if (c) atomicAdd(&p.counts[2 * r_first + lane], (int)c);
            rcnt[lane] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else if (!MULTI) {
        if (lane == 0) {
            if (acc_a) atomicAdd(&p.counts[2 * r_first], (int)acc_a);
            if (acc_b) atomicAdd(&p.counts[2 * r_first + 1], (int)acc_b);
        }
    }
}

// Which read contains the first position of each pass (one thread per pass): keeps the
// binary search over the offsets out of the probe kernels' waves.  Passes that touch more than one
// read are listed for the multi-read kernel (*n_multi is zeroed by the launcher).
__global__ void __launch_bounds__(1024)
tbk_pass_index_kernel(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t total, uint64_t n_passes,
                      uint32_t *__restrict__ pass_read, uint32_t *__restrict__ multi_list, uint32_t *__restrict__ n_multi,
                      uint64_t *__restrict__ two_list, uint32_t *__restrict__ n_two, int use_two, int32_t *__restrict__ counts) {
    const uint64_t pass = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool to_multi = false, to_two = false;
    uint32_t r_first_of_pass = 0;
    if (pass < n_passes) {
        const uint64_t P0 = pass * TBK_PASS;
        const uint64_t r_first = find_read(offsets, n_reads, P0);
        pass_read[pass] = (uint32_t)r_first;
        r_first_of_pass = (uint32_t)r_first;
        const uint64_t last_pos = P0 + TBK_PASS - 1 < total ? P0 + TBK_PASS - 1 : total - 1;
        const uint64_t r_end = r_first < n_reads ? offsets[r_first + 1] : total;
        if (last_pos >= r_end) {
            // (a read r_first + 1 exists: r_end <= last_pos < total.)  Exactly two reads: the second one reaches past the pass.
            to_two = use_two && r_end > P0 && offsets[r_first + 2] > last_pos;
            to_multi = !to_two;
        }
    }
    // one atomic per BLOCK and list: the block's waves take their places in LDS first.  (On 15 kb reads every seventh pass is
    // listed; a quarter of a million atomics on one word took 2 ms - the chip's rate for that, 8 ns each - and one per wave
    // still 0.25 ms of a 0.28 ms launch: round 5.)
    __shared__ uint32_t s_count[2], s_base[2];
    if (threadIdx.x < 2) s_count[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t in_wave[2] = {0, 0}, wave_at[2] = {0, 0};
    const bool mine[2] = {to_multi, to_two};
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(mine[l]);
        if (m != 0) {
            const int leader = __builtin_ctzll(m);
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(&s_count[l], (uint32_t)__popcll(m));
            wave_at[l] = (uint32_t)__shfl((int)base, leader, 64);
            in_wave[l] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_count[threadIdx.x] != 0) s_base[threadIdx.x] = atomicAdd(threadIdx.x == 0 ? n_multi : n_two, s_count[threadIdx.x]);
    __syncthreads();
    if (to_multi) multi_list[s_base[0] + wave_at[0] + in_wave[0]] = (uint32_t)pass;
    if (to_two) two_list[s_base[1] + wave_at[1] + in_wave[1]] = pass | ((uint64_t)r_first_of_pass << 32);  // (the two-read kernel's block starts from this one load)
    // the same launch clears the per-read counters the probe kernels add to
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = pass; i < 2 * n_reads; i += step) counts[i] = 0;
}

// Two kernels share probe_pass: one for the passes that lie inside a single read and one for those that
// touch several (MULTI).  They were one kernel until round 3; its register allocation was the
// maximum over both passes (125 VGPRs: 4 waves per SIMD), while a single-read pass alone needs 95-103.
// The probe is bound by memory latency - measured with LDS padding that only lowers occupancy, the
// kernel's time goes 23.8 / 31.6 / 41.4 ms at 4 / 3 / 2 waves per SIMD - so the single-read kernel,
// which does nearly all the work on long reads, is compiled for 5 waves per SIMD where the line is asked for
// front-first (96 VGPRs, 0-9 spills; same box, uniform lists: 23.2 against 24.0 ms, haplotype-shaped lists
// 27.8 against 31.6) and the multi-read kernel keeps 4.  So are the whole-line variants up to W = 6 (96 VGPRs, no
// vector spills; haplotype-shaped lists, same box: 22.6 ms at 5 waves, 25.4 at 4 - profiles/r03/ab_policy_whole.log).
// The pass-index kernel lists the multi-read passes; the single-read kernel skips them.
template <int W, bool M64, bool SAMP, bool FRONT, bool MULTI, bool TWO = false>
__global__ void __launch_bounds__(64 * TBK_WAVES_PER_BLOCK, MULTI ? TBK_MIN_WAVES_MULTI
                                                               : TWO ? (FRONT && W <= 7 ? TBK_MIN_WAVES : 4)  // (front W = 8 and whole lines would spill vector registers at 5)
                                                               : (FRONT ? TBK_MIN_WAVES : (W <= 6 ? TBK_MIN_WAVES_WHOLE : 4)))
tbk_probe_kernel(const ProbeArgs p) {
    // LDS staging of the read tile, one region per wave: a wave only ever reads what it wrote
    // itself, so wave-scope ordering is enough and the waves of a block never wait for each
    // other (no s_barrier in this kernel).
    __shared__ uint64_t stage[TBK_WAVES_PER_BLOCK][TBK_CHUNKS + 2];
    __shared__ uint4 walkq[TBK_WAVES_PER_BLOCK][FRONT ? TBK_QCAP_FRONT : TBK_QCAP];
    __shared__ uint4 backq[TBK_WAVES_PER_BLOCK][FRONT ? TBK_BQCAP : 1];
    __shared__ uint32_t rcnt[TBK_WAVES_PER_BLOCK][MULTI ? 2 * TBK_RCNT : (TWO ? 4 : 1)];
    const uint32_t lane = threadIdx.x & 63u;
#if TBK_OCC_PAD
    __shared__ uint32_t occ_pad[TBK_OCC_PAD / 4];
    occ_pad[threadIdx.x] = 0;
    if (p.k == 99) p.counts[0] = (int32_t)occ_pad[(threadIdx.x * 97u) % (TBK_OCC_PAD / 4)];
#endif
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t per_iter = (uint64_t)gridDim.x * TBK_WAVES_PER_BLOCK;
    if (MULTI) { rcnt[wave][lane] = 0; rcnt[wave][64 + lane] = 0; }  // per-read tallies (zero between passes)
    if (TWO && lane < 4) rcnt[wave][lane] = 0;
    const uint64_t n_work = MULTI ? (uint64_t)*p.n_multi : TWO ? (uint64_t)*p.n_two : p.pass_hi - p.pass_lo;

    // The multi-read kernel walks its list with a fixed grid; the single-read kernel is launched with one block per
    // pass and has no loop around the pass: the loop's live state cost it 8 spilled registers, and a kernel that
    // uses scratch memory at all ran 18-22 ms from one stream to the next where this one runs 18-19 (EXPERIMENTS.md).
    // The two-read kernel is launched with one block per possible list entry (a batch of n reads has fewer than n
    // two-read passes); blocks past the list's end leave at once.
    for (uint64_t item = (uint64_t)blockIdx.x * TBK_WAVES_PER_BLOCK + wave; item < n_work; item += per_iter) {
        const uint64_t two_entry = TWO ? p.two_list[item] : 0;
        const uint64_t pass = MULTI ? (uint64_t)p.multi_list[item] : TWO ? (two_entry & 0xFFFFFFFFull) : p.pass_lo + item;
        if (MULTI && (pass < p.pass_lo || pass >= p.pass_hi)) continue;  // (the list is the whole batch's, in no order)
        if (TWO && (pass < p.pass_lo || pass >= p.pass_hi)) return;
        const uint64_t P0 = pass * TBK_PASS;
        // which read(s) does this pass touch?  (wave-uniform)
        const uint64_t r_first = TWO ? (two_entry >> 32) : (uint64_t)p.pass_read[pass];
        const uint64_t r_end = r_first < p.n_reads ? p.offsets[r_first + 1] : p.total;
        if (!MULTI && !TWO) {
            const uint64_t last_pos = (P0 + TBK_PASS - 1 < p.total ? P0 + TBK_PASS - 1 : p.total - 1);
            if (last_pos >= r_end) return;  // the multi-read or the two-read kernel's
        }
        if (p.codes != nullptr) {  // packed input (wave-uniform): the chunk words are there already
            const uint64_t c0 = P0 / 16 + lane;
            stage[wave][lane] = load_packed_chunk(p.codes, p.bad16, c0, p.n_chunks);
            stage[wave][64 + lane] = load_packed_chunk(p.codes, p.bad16, c0 + 64, p.n_chunks);
            if (lane < 2) stage[wave][128 + lane] = load_packed_chunk(p.codes, p.bad16, c0 + 128, p.n_chunks);
        } else {
            stage[wave][lane] = load_chunk(p.bases, P0 + (uint64_t)lane * 16, p.total);
            stage[wave][64 + lane] = load_chunk(p.bases, P0 + (uint64_t)(64 + lane) * 16, p.total);
            if (lane < 2) stage[wave][128 + lane] = load_chunk(p.bases, P0 + (uint64_t)(128 + lane) * 16, p.total);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint64_t e0 = stage[wave][2 * lane], e1 = stage[wave][2 * lane + 1], e2 = stage[wave][2 * lane + 2],
                       e3 = stage[wave][2 * lane + 3];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        probe_pass<W, M64, SAMP, MULTI, FRONT, TWO>(p, e0, e1, e2, e3, P0, r_first, r_end, lane, walkq[wave], backq[wave], rcnt[wave]);
        if (!MULTI) return;  // one pass per block
    }
}

// The entry layout's probe kernels: the same three (single-read, two-read, multi-read passes) over probe_pass_entry.
// KIND: 0 narrow entries, 1 wide entries, 2 short keys, 3 full keys (tbk_common.h)
template <int W, bool MULTI, bool TWO = false, int KIND = 0, int LW = 2>
__global__ void __launch_bounds__(64 * TBK_WAVES_PER_BLOCK, MULTI ? TBK_MIN_WAVES_MULTI : TBK_MIN_WAVES)
tbk_probe_entry_kernel(const ProbeArgs p) {
    __shared__ uint64_t stage[TBK_WAVES_PER_BLOCK][TBK_CHUNKS + 2];
    __shared__ uint4 walkq[TBK_WAVES_PER_BLOCK][KIND == 2 ? 1 : TBK_QCAP_ENTRY];  // (short keys never walk: their overflow is a table of its own)
    __shared__ uint4 backq[TBK_WAVES_PER_BLOCK][TBK_BQCAP];
    __shared__ uint16_t walkr[TBK_WAVES_PER_BLOCK][KIND == 1 && (MULTI || TWO) ? TBK_QCAP_ENTRY : 1];  // wide entries, short keys: the read of a queued window travels beside the queue
    __shared__ uint16_t backr[TBK_WAVES_PER_BLOCK][(KIND == 1 || KIND == 2) && (MULTI || TWO) ? TBK_BQCAP : 1];
    __shared__ uint32_t rcnt[TBK_WAVES_PER_BLOCK][MULTI ? 2 * TBK_RCNT : (TWO ? 4 : 1)];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t per_iter = (uint64_t)gridDim.x * TBK_WAVES_PER_BLOCK;
    if (MULTI) { rcnt[wave][lane] = 0; rcnt[wave][64 + lane] = 0; }
    if (TWO && lane < 4) rcnt[wave][lane] = 0;
    const uint64_t n_work = MULTI ? (uint64_t)*p.n_multi : TWO ? (uint64_t)*p.n_two : p.pass_hi - p.pass_lo;
    for (uint64_t item = (uint64_t)blockIdx.x * TBK_WAVES_PER_BLOCK + wave; item < n_work; item += per_iter) {
        const uint64_t two_entry = TWO ? p.two_list[item] : 0;
        const uint64_t pass = MULTI ? (uint64_t)p.multi_list[item] : TWO ? (two_entry & 0xFFFFFFFFull) : p.pass_lo + item;
        if (MULTI && (pass < p.pass_lo || pass >= p.pass_hi)) continue;
        if (TWO && (pass < p.pass_lo || pass >= p.pass_hi)) return;
        const uint64_t P0 = pass * TBK_PASS;
        const uint64_t r_first = TWO ? (two_entry >> 32) : (uint64_t)p.pass_read[pass];
        const uint64_t r_end = r_first < p.n_reads ? p.offsets[r_first + 1] : p.total;
        if (!MULTI && !TWO) {
            const uint64_t last_pos = (P0 + TBK_PASS - 1 < p.total ? P0 + TBK_PASS - 1 : p.total - 1);
            if (last_pos >= r_end) return;  // the multi-read or the two-read kernel's
        }
        if (p.codes != nullptr) {
            const uint64_t c0 = P0 / 16 + lane;
            stage[wave][lane] = load_packed_chunk(p.codes, p.bad16, c0, p.n_chunks);
            stage[wave][64 + lane] = load_packed_chunk(p.codes, p.bad16, c0 + 64, p.n_chunks);
            if (lane < 2) stage[wave][128 + lane] = load_packed_chunk(p.codes, p.bad16, c0 + 128, p.n_chunks);
        } else {
            stage[wave][lane] = load_chunk(p.bases, P0 + (uint64_t)lane * 16, p.total);
            stage[wave][64 + lane] = load_chunk(p.bases, P0 + (uint64_t)(64 + lane) * 16, p.total);
            if (lane < 2) stage[wave][128 + lane] = load_chunk(p.bases, P0 + (uint64_t)(128 + lane) * 16, p.total);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint64_t e0 = stage[wave][2 * lane], e1 = stage[wave][2 * lane + 1], e2 = stage[wave][2 * lane + 2],
                       e3 = stage[wave][2 * lane + 3];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if constexpr (TBK_TMER_LUT && LW == 3 && W == 6 && !TWO && !MULTI) {
            // The staged tile is in registers: its LDS now holds the ranks of all 256 t-mers (W = 6 with 3w positions: t = m - 12 = 4) -
            // canonical form, hash and tag mask folded in - so that a window's new t-mer costs the loop one LDS read instead of eleven
            // vector instructions, one of them a multiply.  The loop runs at 90 % of the vector units' issue rate (EXPERIMENTS.md).
            uint32_t *lut = reinterpret_cast<uint32_t *>(stage[wave]);
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) {
                const uint32_t x = 4u * lane + i, y = tbk_revcomp32(x, 4);
                lut[x] = tbk_mmer_hash(x < y ? x : y) & ~31u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        probe_pass_entry<W, MULTI, TWO, KIND, LW>(p, e0, e1, e2, e3, P0, r_first, r_end, lane, walkq[wave], backq[wave], walkr[wave], backr[wave], rcnt[wave],
                                                  reinterpret_cast<const uint32_t *>(stage[wave]));
        if (!MULTI) return;  // one pass per block
    }
}

// =======================================================================================
// launchers (called from tbk_host.cpp)
// =======================================================================================
extern "C" hipError_t tbk_launch_entry_insert(uint64_t *slots, uint32_t n_buckets, uint32_t half, TbkMz mz, int k, const uint64_t *d_keys, uint64_t n,
                                              int skip_a, int wide, unsigned long long *d_cnt, int *d_failed, hipStream_t stream) {
    TbkEntryGeom g;
    if (!(wide ? tbk_wentry_geom(k, mz, &g) : tbk_entry_geom(k, mz, &g))) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (wide) hipLaunchKernelGGL(tbk_wentry_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slots, n_buckets, half, mz, g, k, d_keys, n, skip_a, d_cnt, d_failed);
    else hipLaunchKernelGGL(tbk_entry_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slots, n_buckets, half, mz, g, k, d_keys, n, skip_a, d_cnt, d_failed);
    return hipGetLastError();
}


extern "C" hipError_t tbk_launch_full_insert(uint64_t *slots, uint32_t n_buckets, uint32_t half, TbkMz mz, int k, const uint64_t *d_keys, uint64_t n, int skip_a,
                                             unsigned long long *d_cnt, int *d_failed, uint32_t only_below, hipStream_t stream) {
    if (!tbk_full_geom(k, mz)) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(tbk_full_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slots, n_buckets, half, mz, k, d_keys, n, skip_a, d_cnt, d_failed, only_below);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_short_insert(uint64_t *slots, uint32_t n_buckets, uint32_t over_mask, uint32_t half, TbkMz mz, int k, const uint64_t *d_keys, uint64_t n,
                                              int skip_a, unsigned long long *d_cnt, int *d_failed, uint32_t line_cap, hipStream_t stream) {
    TbkShortGeom g;
    if (!tbk_short_geom(k, mz, n_buckets, &g)) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(tbk_short_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint32_t *)slots, n_buckets, (unsigned long long *)(slots + (uint64_t)n_buckets * 16), over_mask,
                       half, mz, g, k, d_keys, n, skip_a, d_cnt, d_failed, line_cap < 1 ? 1u : line_cap > 32 ? 32u : line_cap);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_insert(uint64_t *slots, uint32_t n_buckets, uint32_t stride, uint32_t half, TbkMz mz,
                                        const uint64_t *d_keys, uint64_t n, uint32_t *d_overflowed, uint32_t *d_left_line, uint32_t guests, TbkTableView skip,
                                        unsigned long long *d_distinct, unsigned long long *d_skipped, int *d_failed,
                                        hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(tbk_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slots, n_buckets, stride,
                       half, mz, d_keys, n, d_overflowed, d_left_line, guests, skip, d_distinct, d_skipped, d_skipped + 1, d_failed);
    return hipGetLastError();
}

// n_halves = n_buckets * stride / 8; d_overflowed has one bit per half (bit index = bucket * halves + which)
extern "C" hipError_t tbk_launch_order(uint64_t *slots, uint64_t n_halves, const uint32_t *d_overflowed, const uint32_t *d_left_line,
                                       uint32_t flags, uint32_t stride, hipStream_t stream) {
    if (n_halves == 0) return hipSuccess;
    hipLaunchKernelGGL(tbk_order_kernel, dim3((unsigned)((n_halves + 255) / 256)), dim3(256), 0, stream, slots, n_halves, d_overflowed, d_left_line, flags, stride);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_contains(TbkTableView t, const uint64_t *d_keys, uint64_t n,
                                          uint8_t *d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tbk_contains_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t,
                       d_keys, n, d_out);
    return hipGetLastError();
}

// Scatter the packed format's exceptions into the dense per-chunk mask array, which is all zero between
// batches: `clear` = 0 ORs the masks in before the probe, `clear` = 1 takes the same entries out again
// after it - a batch touches a handful of entries, and zeroing the whole array (an eighth of the batch's
// bases in bytes) on the copy stream cost more than the batch's H2D copy.  Masks are OR-ed in 32-bit words
// (two chunks per word), so an index listed twice is harmless, and the positions at or past `total` in the
// last, partial chunk are marked here whether or not the caller's packer listed them: the probe kernel
// relies on that mask (load_packed_chunk).
__global__ void __launch_bounds__(256)
tbk_scatter_bad_kernel(const uint32_t *__restrict__ exc_chunk, const uint16_t *__restrict__ exc_mask, uint64_t n,
                       uint16_t *__restrict__ bad16, uint64_t total, int clear) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t *words = reinterpret_cast<uint32_t *>(bad16);  // hipMalloc'ed: 4-byte aligned, capacity even
    if (i < n) {
        const uint32_t c = exc_chunk[i];
        if (clear) atomicAnd(&words[c >> 1], ~(0xFFFFu << (16 * (c & 1u))));
        else atomicOr(&words[c >> 1], (uint32_t)exc_mask[i] << (16 * (c & 1u)));
    }
    if (i == 0 && (total & 15u) != 0) {
        const uint64_t c = total >> 4;
        if (clear) atomicAnd(&words[c >> 1], ~(0xFFFFu << (16 * (c & 1u))));
        else atomicOr(&words[c >> 1], ((0xFFFFu << (total & 15u)) & 0xFFFFu) << (16 * (c & 1u)));
    }
}

extern "C" hipError_t tbk_launch_scatter_bad(const uint32_t *d_exc_chunk, const uint16_t *d_exc_mask, uint64_t n, uint16_t *d_bad16,
                                             uint64_t total, int clear, hipStream_t stream) {
    if (n == 0 && (total & 15u) == 0) return hipSuccess;
    hipLaunchKernelGGL(tbk_scatter_bad_kernel, dim3((unsigned)((std::max<uint64_t>(n, 1) + 255) / 256)), dim3(256), 0, stream, d_exc_chunk, d_exc_mask, n, d_bad16, total, clear);
    return hipGetLastError();
}

// d_codes == nullptr: the read stream is d_bases (ASCII); else it is (d_codes, d_bad16) and d_bases is not read.
// d_scratch: 2 * pass_cap + 16 words (pass -> read index, the multi-read passes' list, their number).
// A probe is launched in two parts so that a batch can be probed slice by slice while its bases are still
// arriving: tbk_launch_probe_index (needs the offsets only) once, then tbk_launch_probe_range for every slice of
// passes [pass_lo, pass_hi) once the bases up to (pass_hi * 2048 + 2080) are there.
static void fill_args(ProbeArgs &p, const uint8_t *d_bases, const uint32_t *d_codes, const uint16_t *d_bad16, const uint64_t *d_offsets,
                      uint64_t n_reads, uint64_t total, TbkPairView t, int k, int32_t *d_counts, uint32_t *d_scratch, uint64_t pass_cap) {
    p.codes = d_codes; p.bad16 = d_bad16; p.n_chunks = (total + 15) / 16;
    p.bases = d_bases; p.offsets = d_offsets; p.n_reads = n_reads; p.total = total;
    p.n_passes = (total + TBK_PASS - 1) / TBK_PASS;
    p.t = t; p.k = k; p.counts = d_counts; p.pass_read = d_scratch; p.multi_list = d_scratch + pass_cap; p.two_list = (const uint64_t *)(d_scratch + 2 * pass_cap);
    p.n_multi = d_scratch + 4 * pass_cap; p.n_two = d_scratch + 4 * pass_cap + 1;  // (tbk_host.cpp sizes the scratch: 4 * pass_cap + 16 words, pass_cap even)
    p.pass_lo = 0; p.pass_hi = p.n_passes;
}

// use_two != 0: passes that touch exactly two reads get a list of their own (the two-read kernel's; built for the
// mod-sampling variants - tbk_probe_has_two_read_kernel)
// (tbk_options.two_read_kernel = 0: two-read passes go to the multi-read kernel, as before round 3)
extern "C" int tbk_probe_has_two_read_kernel(TbkMz mz) { return mz.w >= 2 && mz.t > 0; }

extern "C" hipError_t tbk_launch_probe_index(const uint64_t *d_offsets, uint64_t n_reads, uint64_t total, int32_t *d_counts, uint32_t *d_scratch,
                                             uint64_t pass_cap, int use_two, hipStream_t stream) {
    if (total == 0 || n_reads == 0) return hipSuccess;
    const uint64_t n_passes = (total + TBK_PASS - 1) / TBK_PASS;
    if (n_passes > pass_cap) return hipErrorInvalidValue;
    uint32_t *d_multi = d_scratch + pass_cap, *d_n = d_scratch + 4 * pass_cap;
    uint64_t *d_two = (uint64_t *)(d_scratch + 2 * pass_cap);
    if (pass_cap & 1) return hipErrorInvalidValue;  // (64-bit entries)
    hipError_t e = hipMemsetAsync(d_n, 0, 2 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tbk_pass_index_kernel, dim3((unsigned)((n_passes + 1023) / 1024)), dim3(1024), 0, stream,
                       d_offsets, n_reads, total, n_passes, d_scratch, d_multi, d_n, d_two, d_n + 1, use_two, d_counts);
    return hipGetLastError();
}

extern "C" hipError_t tbk_launch_probe_range(const uint8_t *d_bases, const uint32_t *d_codes, const uint16_t *d_bad16, const uint64_t *d_offsets,
                                             uint64_t n_reads, uint64_t total, TbkPairView t, int k, int32_t *d_counts, uint32_t *d_scratch,
                                             uint64_t pass_cap, uint64_t pass_lo, uint64_t pass_hi, int max_blocks, int use_two, hipEvent_t between, hipStream_t stream,
                                             int only_single) {
    // only_single != 0: the single-read kernel alone (a batch of ONE read has no pass that touches two: tbk_count_kmers_in_read's path)
    if (total == 0 || n_reads == 0 || pass_hi <= pass_lo) return hipSuccess;
    ProbeArgs p;
    fill_args(p, d_bases, d_codes, d_bad16, d_offsets, n_reads, total, t, k, d_counts, d_scratch, pass_cap);
    if (p.n_passes > pass_cap || pass_hi > p.n_passes) return hipErrorInvalidValue;
    p.pass_lo = pass_lo; p.pass_hi = pass_hi;
    const uint64_t blocks = (pass_hi - pass_lo + TBK_WAVES_PER_BLOCK - 1) / TBK_WAVES_PER_BLOCK;  // single-read kernel: one block per pass
    // the multi-read kernel walks its list with a grid that fills the chip a few times over (max_blocks caps it: a test knob)
    uint64_t blocks_multi = std::min<uint64_t>(blocks, 16384);
    if (max_blocks > 0 && blocks_multi > (uint64_t)max_blocks) blocks_multi = (uint64_t)max_blocks;
    const dim3 grid((unsigned)blocks), grid_multi((unsigned)blocks_multi), block(64 * TBK_WAVES_PER_BLOCK);
    // kernel variant: W m-mers per span; 32-bit (m <= 16) or 64-bit m-mers; random-minimizer or
    // mod-sampling selection; front or whole-line layout
    const bool m64 = t.mz.m > 16, samp = t.mz.t > 0, front = (t.guests & TBK_FLAG_FRONT) != 0, entry = (t.guests & TBK_FLAG_ENTRY) != 0;
    if (front && t.mz.w < 2) return hipErrorInvalidValue;  // front tables are built for minimizer spans only (tbk_host.cpp)
    const bool wide = (t.guests & TBK_FLAG_WIDE) != 0, shortk = (t.guests & TBK_FLAG_SHORT) != 0, fullk = (t.guests & TBK_FLAG_FULL) != 0;
    const bool span3 = (entry || shortk) && t.mz.t > 0 && t.mz.t == t.mz.m - 2 * t.mz.w;
    if ((entry || shortk) && t.mz.t > 0 && !span3 && t.mz.t != t.mz.m - t.mz.w) return hipErrorInvalidValue;
    // (the single-read entry kernel of LW = 3, W = 6 ranks t-mers through a 256-entry table in LDS: 4^t entries, t = 4, tags of 5 bits -
    // what tbk_mz_span3 gives at w = 6 today (m = 16); a view with another t must not reach that instantiation)
    if (TBK_TMER_LUT && span3 && t.mz.w == 6 && t.mz.t != 4) return hipErrorInvalidValue;
    if (shortk) {
        TbkShortGeom g;
        if (!tbk_short_geom(k, t.mz, t.n_buckets, &g) || t.n_buckets > 0x3FFFFFFFu) return hipErrorInvalidValue;
    }
    if (fullk && (!tbk_full_geom(k, t.mz) || t.mz.t != t.mz.m - t.mz.w || t.n_buckets > 0x3FFFFFFFu)) return hipErrorInvalidValue;
    if (entry) {
        TbkEntryGeom g;
        if (!(wide ? tbk_wentry_geom(k, t.mz, &g) : tbk_entry_geom(k, t.mz, &g)) || t.n_buckets > 0x3FFFFFFFu) return hipErrorInvalidValue;
    }
    // the two-read kernel: one block per possible list entry (fewer two-read passes than reads, and than passes)
    const uint64_t blocks_two = use_two && tbk_probe_has_two_read_kernel(t.mz) ? std::min<uint64_t>(n_reads > 1 ? n_reads - 1 : 0, p.n_passes) : 0;  // (the list is the whole batch's: every launch walks all of it)
    const dim3 grid_two((unsigned)std::max<uint64_t>(1, blocks_two));
    hipError_t e = hipSuccess;
    for (int which = 2; which >= 0; which--) {  // 2: multi-read passes, 1: two-read passes, 0: single-read passes (timed by itself: `between`)
        if (which == 1 && blocks_two == 0) continue;
        if (which != 0 && only_single) continue;
        // the event in front of the single-read kernel: the host times it by itself (tbk_kernel_timing_read2)
        if (which == 0 && between != nullptr) { e = hipEventRecord(between, stream); if (e != hipSuccess) return e; }
#define TBK_LAUNCH(N, M, S, F) do { if (which == 2) hipLaunchKernelGGL((tbk_probe_kernel<N, M, S, F, true>), grid_multi, block, 0, stream, p); \
                                    else if (which == 0) hipLaunchKernelGGL((tbk_probe_kernel<N, M, S, F, false>), grid, block, 0, stream, p); \
                                    else if (S && N >= 2) hipLaunchKernelGGL((tbk_probe_kernel<N, M, (S && N >= 2), F, false, (S && N >= 2)>), grid_two, block, 0, stream, p); } while (0)
#define TBK_W(N) case N: if (samp && front) { if (m64) TBK_LAUNCH(N, true, true, true); else TBK_LAUNCH(N, false, true, true); } \
                         else if (samp) { if (m64) TBK_LAUNCH(N, true, true, false); else TBK_LAUNCH(N, false, true, false); } \
                         else if (front) { if (m64) TBK_LAUNCH(N, true, false, true); else TBK_LAUNCH(N, false, false, true); } \
                         else { if (m64) TBK_LAUNCH(N, true, false, false); else TBK_LAUNCH(N, false, false, false); } break;
        if (entry || shortk || fullk) {
#define TBK_E1(N, WD, LW) do { if (which == 2) hipLaunchKernelGGL((tbk_probe_entry_kernel<N, true, false, WD, LW>), grid_multi, block, 0, stream, p); \
                         else if (which == 0) hipLaunchKernelGGL((tbk_probe_entry_kernel<N, false, false, WD, LW>), grid, block, 0, stream, p); \
                         else hipLaunchKernelGGL((tbk_probe_entry_kernel<N, false, true, WD, LW>), grid_two, block, 0, stream, p); } while (0)
            // (3w t-mer positions: narrow entries and short keys with spans of up to six m-mers - tbk_mz_span3)
#define TBK_E(N, WD) case N: if (span3) TBK_E1(N, WD, 3); else TBK_E1(N, WD, 2); break;
#define TBK_E2(N, WD) case N: if (span3) return hipErrorInvalidValue; TBK_E1(N, WD, 2); break;
            if (fullk) {
                switch (t.mz.w) {
#ifdef TBK_ONLY_W6
                    TBK_E2(6, 3) TBK_E2(8, 3)
#else
                    TBK_E2(2, 3) TBK_E2(3, 3) TBK_E2(4, 3) TBK_E2(5, 3) TBK_E2(6, 3) TBK_E2(7, 3) TBK_E2(8, 3)
#endif
                    default: return hipErrorInvalidValue;
                }
            } else if (shortk) {
                switch (t.mz.w) {
#ifdef TBK_ONLY_W6
                    TBK_E(6, 2)
#else
                    TBK_E(2, 2) TBK_E(3, 2) TBK_E(4, 2) TBK_E(5, 2) TBK_E(6, 2) TBK_E2(7, 2) TBK_E2(8, 2)
#endif
                    default: return hipErrorInvalidValue;
                }
            } else if (wide) {
                switch (t.mz.w) {
#ifdef TBK_ONLY_W6
                    TBK_E2(6, 1) TBK_E2(8, 1)
#else
                    TBK_E2(2, 1) TBK_E2(3, 1) TBK_E2(4, 1) TBK_E2(5, 1) TBK_E2(6, 1) TBK_E2(7, 1) TBK_E2(8, 1)
#endif
                    default: return hipErrorInvalidValue;
                }
            } else {
                switch (t.mz.w) {
#ifdef TBK_ONLY_W6
                    TBK_E(6, 0)
#else
                    TBK_E(2, 0) TBK_E(3, 0) TBK_E(4, 0) TBK_E(5, 0) TBK_E(6, 0) TBK_E2(7, 0)
#endif
                    default: return hipErrorInvalidValue;
                }
            }
#undef TBK_E
#undef TBK_E2
#undef TBK_E1
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            continue;
        }
        switch (t.mz.w) {
#ifdef TBK_ONLY_W6  // experiment builds (tools/build_variant.sh): the bench configuration's kernels only
            TBK_W(6)
#else
            case 0: TBK_LAUNCH(0, false, false, false); break;
            case 1: if (m64) TBK_LAUNCH(1, true, false, false); else TBK_LAUNCH(1, false, false, false); break;
            TBK_W(2) TBK_W(3) TBK_W(4) TBK_W(5) TBK_W(6) TBK_W(7) TBK_W(8)
#endif
            default: return hipErrorInvalidValue;
        }
#undef TBK_W
#undef TBK_LAUNCH
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

#ifdef TBK_COUNTERS
extern "C" int tbk_debug_counters(unsigned long long out[8], int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tbk_dbg), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(tbk_dbg), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -5;
}
#endif

// number of uint32 entries of pass_read scratch a batch of `total` bases needs
extern "C" uint64_t tbk_probe_passes(uint64_t total) { return (total + TBK_PASS - 1) / TBK_PASS; }
