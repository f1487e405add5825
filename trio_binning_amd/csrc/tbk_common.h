// tbk_common.h — layout and arithmetic shared by host and device code of libtbk_hip.so.
//
// Table layout in HBM (replicated per GPU).
//   The classifier holds the two k-mer lists as two open-addressing tables of uint64 slots
//   that share one bucket index and are interleaved line by line: bucket i is one 128-byte
//   aligned line = [8 hapA slots | 8 hapB slots].  Measured on MI355X (profiles/,
//   DESIGN.md §4): random line reads saturate at ~48 G lines/s whether the line is 64 or
//   128 bytes, so fetching both tables' buckets as ONE 128-byte line halves the cost of a
//   window compared with two independent 64-byte lines.  hapA and hapB stay separate tables
//   (separate slots, separate probes); a key both lists hold is stored for hapA only, because
//   hapA is asked first (c/kmers.c:291-294), so the two halves are disjoint.
//   A slot holds a packed k-mer (base i at bits 2i..2i+1, A=0 C=1 G=2 T=3 — the reference's
//   encoding, c/kmers.c:50-72) or TBK_EMPTY.  A key lives in the first half along its probe
//   sequence (tbk_next_bucket: home bucket, second-choice bucket, then linear) that had a free
//   slot when it was inserted.  Inserts always take the FIRST free slot of a half and nothing
//   is ever removed, so the occupied slots of a half form a prefix.  After the inserts the last
//   two slots of every full half are put in the order that says whether any key went past the
//   half (slot 6 > slot 7) or not (slot 6 < slot 7; a half with a free slot has slot 7 =
//   TBK_EMPTY, the largest value): a lookup that misses stops at the first half nothing went
//   past, and the probe kernel learns that from one compare of two slots it already holds.
//   Before a key leaves its LINE it tries the other list's half of the same line (k < 32): it is
//   stored there with TBK_GUEST set, and only if that half is full too does it go on along its probe
//   sequence.  The order of slots 4 and 5 of a full half that keys went past says which: slot 4 >
//   slot 5 = "some key of this list left the line".  A lookup holds the whole line in registers, so
//   a guest costs it two more compares where a key that left the line costs another random line;
//   lists shaped like real data (the k overlapping k-mers around one variant share ~6 minimizers,
//   in both lists at once) fill single halves long before they fill lines.
//   A standalone list (tbk_table) is just its packed keys in HBM; a single-table form of
//   the same layout (64-byte lines, 8 slots) is built on demand for tbk_table_contains.
//   The reference's layout (8-byte slots + a parallel "full" byte array at load 0.75,
//   c/kmers.c:12-38,160-180) is not observable; only membership is.
//
// TBK_EMPTY = all ones is never a canonical lookup key: for k < 32 every key is < 4^k, and
// for k = 32 all-ones is T x32 whose reverse complement A x32 = 0 is smaller.  A list line
// that encodes to all-ones (only "T"x32) is dropped at insert; it could never be matched
// (the reference stores list lines verbatim, c/kmers.c:113, and looks up min(fwd, rc),
// c/kmers.c:255).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define TBK_HD __host__ __device__ __forceinline__
#else
#define TBK_HD inline
#endif

#define TBK_EMPTY 0xFFFFFFFFFFFFFFFFull
// The key an invalid window (non-ACGT byte, read boundary) looks up.  Like TBK_EMPTY it can
// never be a canonical key (k < 32: >= 4^k; k = 32: "GTTT...T", whose reverse complement
// "AAA...AC" is smaller), so it is never stored — a list line that packs to it is dead in
// the reference too — and a window carrying it can never hit.
#define TBK_NOKEY 0xFFFFFFFFFFFFFFFEull
#define TBK_SLOTS_PER_BUCKET 8
#define TBK_BUCKET_BYTES 64
// A key of one list stored in a free slot of the OTHER list's half of its line (its own half was
// full) carries this tag.  Keys of k < 32 leave bit 63 free (they are below 4^31 = 2^62); a tagged
// key never equals a lookup key, TBK_EMPTY or TBK_NOKEY, so the raw compares of the probe kernel's
// fast path do not see guests at all: only a lookup that found its own half left by keys looks for
// them (in slots it already holds in registers).  k = 32 has no free bit: no guests there.
#define TBK_GUEST 0x8000000000000000ull

// ---- bucket selection --------------------------------------------------------------------
// Not the reference's hash_function (c/kmers.c:98-103): hash values are not observable, so
// the bucket of a key is chosen for the GPU.
//
// Mode "minimizer" (w >= 1): the bucket of a canonical k-mer is chosen by the minimizer of
// its central span of m + w - 1 bases: H = min over the span's w m-mers of hash(canonical
// m-mer), bucket = reduce(scramble(H)).  The span is central and the m-mers are
// canonicalised, so a k-mer and its reverse complement give the same H (the probe kernel
// computes H from the forward strand alone).  Consecutive windows of a read share their
// minimizer 1 - 2/(w+1) of the time, so they probe the SAME 128-byte line and the kernel
// re-uses the line it already holds instead of fetching another random line from HBM.
// (m, w) are picked per k and per table size by tbk_mz_params so that m + w - 1 has k's parity
// (the span must be central) and so that there are enough distinct m-mers for the keys: all
// keys sharing a minimizer land in one bucket, so about n_keys * w / (4^m / 2) keys compete
// for the most popular minimizers.  m <= 16 keeps m-mer arithmetic in 32 bits; bigger tables
// (or TBK_MINIMIZER_M) use longer m-mers on a 64-bit path.
//
// Which m-mer of the span is sampled decides how often consecutive windows switch buckets
// (the "density" of the sampling scheme):
//   * random minimizer (t = 0): the m-mer with the smallest hash; density 2/(w+1) = 0.286 at w=6;
//   * mod-sampling (t = m - w, Groot Koerkamp & Pibiri 2024): rank the span's 2w canonical
//     t-mers by hash, take the position x of the smallest, sample the m-mer at x mod w.
//     Density 3/(2w+1) = 0.231 at w = 6 — 19 % fewer line switches — and the sampled m-mers
//     are not biased towards small hashes, so buckets fill more evenly.
//     Ties (the smallest t-mer hash attained at several positions: repeats, palindromes) are
//     settled on the INSERT side: a key is stored under the bucket of every tied position, so a
//     lookup may pick any of them — it works on the forward strand of the read, where position
//     x of the canonical k-mer appears as x or 2w-1-x, and (2w-1-x) mod w = w-1-(x mod w) is
//     the same m-mer seen from the other strand.  Ties are rare, the extra copies negligible,
//     and a lookup still meets a key at most once (it reads one home bucket).
//     The number of ranked positions is l = w + m - t in general (tbk_mz_positions): the entry
//     layouts and short keys take t = m - 2w where that leaves t >= 4 (tbk_mz_span3), l = 3w
//     positions, density 4/(3w+1) = 0.211 at w = 6; the mirror of position x is l-1-x, and
//     (l-1-x) mod w = w-1-(x mod w) still.
//
// Mode "plain" (w = 0): bucket = reduce(mix32(key)), one random line per window.
struct TbkMz {
    int w;  // m-mers per span (0 = plain mode)
    int m;  // m-mer length (<= 16: 32-bit m-mer arithmetic; 17..32: 64-bit)
    int o;  // first base of the span inside the k-mer: (k - (m + w - 1)) / 2
    int t;  // 0: the span's m-mer with the smallest hash is sampled ("random minimizer");
            // t > 0: mod-sampling — the span's w + m - t t-mers are ranked (t = m - w: 2w of them;
            // t = m - 2w: 3w), and the m-mer at (position of the smallest t-mer) mod w is sampled
};

TBK_HD uint32_t tbk_mix32(uint64_t key) {
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t h = lo * 0x9E3779B1u ^ (hi + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0xC2B2AE3Du;
    h ^= h >> 13;
    return h;
}

// bucket = floor(h * n_buckets / 2^32): uniform over [0, n_buckets) without a division.
TBK_HD uint32_t tbk_reduce(uint32_t h, uint32_t n_buckets) {
    return (uint32_t)(((uint64_t)h * (uint64_t)n_buckets) >> 32);
}

// order of canonical m-mers for minimizer selection (a bijection of 32-bit values; the xor
// keeps poly-A, which packs to 0, from being everybody's minimizer)
TBK_HD uint32_t tbk_mmer_hash(uint32_t cm) {
    cm = (cm ^ 0x5BD1E995u) * 0x9E3779B1u;
    return cm ^ (cm >> 16);
}

// minimizer hashes are small-biased (a minimum of w values): reversing the bits (one full-rate
// instruction; multiplies are quarter rate) moves their well-mixed low bits to the top, where
// tbk_reduce reads them
TBK_HD uint32_t tbk_scramble(uint32_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(h);
#else
    h = ((h >> 1) & 0x55555555u) | ((h & 0x55555555u) << 1);
    h = ((h >> 2) & 0x33333333u) | ((h & 0x33333333u) << 2);
    h = ((h >> 4) & 0x0F0F0F0Fu) | ((h & 0x0F0F0F0Fu) << 4);
    h = ((h >> 8) & 0x00FF00FFu) | ((h & 0x00FF00FFu) << 8);
    return (h >> 16) | (h << 16);
#endif
}

// reverse complement of an m-base packed value, m <= 16
TBK_HD uint32_t tbk_revcomp32(uint32_t x, int m) {
    uint32_t y = ~x;
    y = ((y >> 2) & 0x33333333u) | ((y & 0x33333333u) << 2);
    y = ((y >> 4) & 0x0F0F0F0Fu) | ((y & 0x0F0F0F0Fu) << 4);
    y = ((y >> 8) & 0x00FF00FFu) | ((y & 0x00FF00FFu) << 8);
    y = (y >> 16) | (y << 16);
    return y >> (32 - 2 * m);
}

// Canonical m-mers longer than 16 bases (big tables).  Returns (order << 32) | place: minimizer
// selection compares the whole 64-bit value (so `order` decides), and the bucket is chosen by
// `place`, an independent 32-bit hash of the same m-mer.  A minimum of w order values is
// confined to the lowest ~1/(w+1) of the 32-bit range; using it as the bucket hash as well
// (fine for the 3e8-bucket tables of the 32-bit path) would leave a 1e9-bucket table half
// unused.  `place` of the selected m-mer is uniform over all 32 bits.
TBK_HD uint64_t tbk_mmer_hash64(uint64_t cm) {
    const uint32_t lo = (uint32_t)cm, hi = (uint32_t)(cm >> 32);
    uint32_t order = (lo ^ 0x5BD1E995u) * 0x9E3779B1u ^ (hi + 0x7F4A7C15u) * 0x85EBCA77u;
    order ^= order >> 15;
    order *= 0xC2B2AE3Du;
    order ^= order >> 16;
    uint32_t place = (lo + 0x165667B1u) * 0x27D4EB2Fu ^ (hi ^ 0x9E3779B9u) * 0xC2B2AE35u;
    place ^= place >> 15;
    place *= 0x85EBCA6Bu;
    return ((uint64_t)order << 32) | place;
}

// reverse complement of an m-base packed value, m <= 32
TBK_HD uint64_t tbk_revcomp64(uint64_t x, int m) {
    uint64_t y = ~x;
    y = ((y >> 2) & 0x3333333333333333ull) | ((y & 0x3333333333333333ull) << 2);
    y = ((y >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((y & 0x0F0F0F0F0F0F0F0Full) << 4);
    y = ((y >> 8) & 0x00FF00FF00FF00FFull) | ((y & 0x00FF00FF00FF00FFull) << 8);
    y = ((y >> 16) & 0x0000FFFF0000FFFFull) | ((y & 0x0000FFFF0000FFFFull) << 16);
    y = (y >> 32) | (y << 32);
    return y >> (64 - 2 * m);
}

// (m, w, o) for a given k, wanted w and table size.  m_need = shortest m-mer with at most
// ~0.9 keys per distinct canonical m-mer (the 300 M-key / m = 16 / w = 6 operating point that
// was measured); the preferred m is max(16, m_need), then one shorter (never below m_need or
// 15), then one longer — whichever gives a span m + w - 1 <= k with k's parity; w shrinks
// until something fits.  m_force > 0 pins m (tests).  Plain mode when nothing fits (k < 15).
TBK_HD TbkMz tbk_mz_params(int k, int w_target, uint64_t n_keys, int m_force, int mod_sampling) {
    TbkMz z;
    z.w = 0; z.m = 0; z.o = 0; z.t = 0;
    if (w_target > 8) w_target = 8;
    if (w_target < 1) return z;
    int m_need = 15;
    while (m_need < 32) {
        const double distinct = 0.5 * (double)(1ull << (2 * (m_need > 31 ? 31 : m_need))) * (m_need > 31 ? 4.0 : 1.0);
        if ((double)n_keys * (double)w_target <= 0.9 * distinct) break;
        m_need++;
    }
    const int m0 = m_need > 16 ? m_need : 16;
    for (int w = w_target; w >= 1; w--) {
        int cand[3] = {m0, m0 - 1, m0 + 1};
        if (m_force > 0) { cand[0] = m_force; cand[1] = cand[2] = -1; }
        for (int c = 0; c < 3; c++) {
            const int m = cand[c];
            if (m < 1 || m > 32) continue;
            if (m_force <= 0 && (m < 15 || m < m_need)) continue;
            const int span = m + w - 1;
            if (span <= k && ((k - span) & 1) == 0) {
                z.w = w; z.m = m; z.o = (k - span) / 2;
                // mod-sampling needs t = m - w long enough that two of the 2w t-mers rarely
                // coincide, and 2w <= 16 positions (4-bit position tag)
                if (mod_sampling && w >= 2 && m - w >= 8 && m - w <= 16) z.t = m - w;
                return z;
            }
        }
    }
    return z;
}

// Bucket of a packed key.  For canonical keys this equals what the probe kernel computes
// from the read; for the (dead) non-canonical list lines any value is fine.
// bucket selected by the span's m-mer at position p (mod-sampling)
TBK_HD uint32_t tbk_bucket_at(uint64_t key, TbkMz z, int p, uint32_t n_buckets) {
    if (z.m <= 16) {
        const uint32_t mmask = z.m == 16 ? 0xFFFFFFFFu : ((1u << (2 * z.m)) - 1u);
        const uint32_t x = (uint32_t)(key >> (2 * (z.o + p))) & mmask;
        const uint32_t y = tbk_revcomp32(x, z.m);
        return tbk_reduce(tbk_mmer_hash(x < y ? x : y), n_buckets);  // a sampled m-mer's hash is not small-biased
    }
    const uint64_t mmask = z.m == 32 ? ~0ull : ((1ull << (2 * z.m)) - 1ull);
    const uint64_t x = (key >> (2 * (z.o + p))) & mmask;
    const uint64_t y = tbk_revcomp64(x, z.m);
    return tbk_reduce((uint32_t)tbk_mmer_hash64(x < y ? x : y), n_buckets);
}

// The t-mer positions of a span under mod-sampling: l = w + m - t.  t = m - w ranks the span's 2w t-mers (density
// 3 / (2w + 1)); t = m - 2w ranks 3w of them and switches less often still - density (floor((l - 1) / w) + 2) / (l + 1) =
// 4 / (3w + 1): 0.2105 against 0.2308 at w = 6 (simulated on random sequence: 0.2100 / 0.2308) - at the price of shorter
// t-mers, whose ranks tie more often (t = 4: in 6 % of the windows; a tie is a second copy of the key on the insert side).
// The entry layouts and short keys use it where t stays at 4 or more (tbk_mz_span3); the key layouts keep 2w.
TBK_HD int tbk_mz_positions(TbkMz z) { return z.t > 0 ? z.w + z.m - z.t : z.w; }
// the bits of a rank that carry the position tag in the probe kernels: 4 (up to 16 positions) or 5
TBK_HD uint32_t tbk_mz_tagmask(TbkMz z) { return tbk_mz_positions(z) > 16 ? 31u : 15u; }
TBK_HD TbkMz tbk_mz_span3(TbkMz z) {
    if (z.t > 0 && z.t == z.m - z.w && z.m - 2 * z.w >= 4 && z.m <= 16 && 3 * z.w <= 32) z.t = z.m - 2 * z.w;
    return z;
}

// rank of the span's t-mer at position i: the hash with its low 4 (5) bits cleared (they carry a
// position tag in the probe kernel)
TBK_HD uint32_t tbk_tmer_rank(uint64_t key, TbkMz z, int i) {
    const uint32_t tmask = z.t == 16 ? 0xFFFFFFFFu : ((1u << (2 * z.t)) - 1u);
    const uint32_t x = (uint32_t)(key >> (2 * (z.o + i))) & tmask;
    const uint32_t y = tbk_revcomp32(x, z.t);
    return tbk_mmer_hash(x < y ? x : y) & ~tbk_mz_tagmask(z);
}

// All buckets a lookup of `key` may select: one per position that attains the smallest t-mer
// rank (mod-sampling), else the single bucket.  Returns the number of distinct buckets (<= 16).
TBK_HD int tbk_bucket_candidates(uint64_t key, TbkMz z, uint32_t n_buckets, uint32_t *out) {
    if (z.w == 0 || z.t == 0) { out[0] = 0; return -1; }  // caller uses tbk_bucket_of
    const int nt = tbk_mz_positions(z);
    uint32_t best = 0xFFFFFFFFu;
    for (int i = 0; i < nt; i++) { const uint32_t g = tbk_tmer_rank(key, z, i); best = g < best ? g : best; }
    int n = 0;
    for (int i = 0; i < nt; i++) {
        if (tbk_tmer_rank(key, z, i) != best) continue;
        const uint32_t b = tbk_bucket_at(key, z, i % z.w, n_buckets);
        bool seen = false;
        for (int j = 0; j < n; j++) seen = seen || out[j] == b;
        if (!seen) out[n++] = b;
    }
    return n;
}

// Bucket of a packed key (for mod-sampling: the first of its candidates).  For canonical keys
// this is what the probe kernel computes from the read; for the (dead) non-canonical list
// lines any value is fine.
TBK_HD uint32_t tbk_bucket_of(uint64_t key, TbkMz z, uint32_t n_buckets) {
    if (z.w == 0) return tbk_reduce(tbk_mix32(key), n_buckets);
    if (z.t > 0) {
        const int nt = tbk_mz_positions(z);
        uint32_t best = 0xFFFFFFFFu;
        for (int i = 0; i < nt; i++) {
            const uint32_t g = tbk_tmer_rank(key, z, i) | (uint32_t)i;
            best = g < best ? g : best;
        }
        return tbk_bucket_at(key, z, (int)(best & tbk_mz_tagmask(z)) % z.w, n_buckets);
    }
    uint32_t best = 0xFFFFFFFFu;
    if (z.m <= 16) {
        const uint32_t mmask = z.m == 16 ? 0xFFFFFFFFu : ((1u << (2 * z.m)) - 1u);
        for (int i = 0; i < z.w; i++) {
            const uint32_t x = (uint32_t)(key >> (2 * (z.o + i))) & mmask;
            const uint32_t y = tbk_revcomp32(x, z.m);
            const uint32_t g = tbk_mmer_hash(x < y ? x : y);
            best = g < best ? g : best;
        }
    } else {
        const uint64_t mmask = z.m == 32 ? ~0ull : ((1ull << (2 * z.m)) - 1ull);
        uint64_t best64 = ~0ull;
        for (int i = 0; i < z.w; i++) {
            const uint64_t x = (key >> (2 * (z.o + i))) & mmask;
            const uint64_t y = tbk_revcomp64(x, z.m);
            const uint64_t g = tbk_mmer_hash64(x < y ? x : y);
            best64 = g < best64 ? g : best64;
        }
        return tbk_reduce((uint32_t)best64, n_buckets);
    }
    return tbk_reduce(tbk_scramble(best), n_buckets);
}

// Device view of a table: bucket b's slots for this list start at
// slots[b * stride + half] (stride 8, half 0 for a standalone table; stride 16 and half 0 / 8
// for the hapA / hapB halves of a paired table).
//
// The `guests` word of the views carries two flags:
#define TBK_FLAG_GUESTS 1u   // a full half's surplus may sit, tagged, in the other half of the line
#define TBK_FLAG_FRONT 2u    // paired table in front layout (below)
// Front layout.  A touched line costs the CU's L1 path the same whether 16 or 128 of its bytes are
// wanted, and a whole [8 hapA | 8 hapB] line is two 64-byte requests per quad; with a key or two per
// bucket nearly every window can be answered from four slots of each list.  So the line is laid out
//     [ A0 A1 A2 A3 | B0 B1 B2 B3 | A4 A5 A6 A7 | B4 B5 B6 B7 ]
// and the probe kernel fetches the first 64 bytes only (one 16-byte load per quad lane: lanes 0,1 hold
// hapA's first four slots, lanes 2,3 hapB's).  Slots fill in index order.  With guests (k < 32) a list's
// fifth key of a bucket first tries the free front slots of the OTHER list, tagged with TBK_GUEST - a
// window still finds it in the 64 bytes it fetches, and counts it for the list whose lanes those are
// not; whether four front slots hold such keys is the order of their slots 0 and 1 (slot 0 > slot 1:
// compare the tagged key too).  What lies behind the front - a list's slots 4..7, the guests there,
// everything "past the half" - is announced by the order of the list's slots 2 and 3 (slot 2 > slot 3:
// look further), and a window that misses in such a front is settled exactly by the deferred walk,
// starting at the home line (tbk_order_kernel writes both orders after the inserts).  Logical slot
// numbers, flags in slots 4..7, guests and probe sequences are those of the plain layout: only
// tbk_slot_at() knows where a slot lies.
TBK_HD uint32_t tbk_slot_at(uint32_t flags, uint32_t stride, uint32_t half, uint32_t s) {
    if ((flags & TBK_FLAG_FRONT) && stride == 16) return ((s & 4u) << 1) + (half >> 1) + (s & 3u);
    return half + s;
}
struct TbkTableView {
    const uint64_t *slots;
    uint32_t n_buckets;
    uint32_t stride;  // slots per bucket line (8 or 16)
    uint32_t half;    // first slot of this list inside the line (0 or 8)
    TbkMz mz;         // bucket selection
    uint32_t guests;  // TBK_FLAG_GUESTS (paired table, k < 32) | TBK_FLAG_FRONT
};

// The probe sequence of a key: its home bucket (chosen by the minimizer, shared with its
// neighbours in a read), then - if that half is full and keys went past it - a second-choice
// bucket chosen by a hash of the whole key, then linearly on from there.  A minimizer shared by
// more keys than a half holds (low-complexity sequence: thousands of distinct k-mers around one
// poly-A m-mer) therefore scatters its surplus over the table instead of piling it up in the
// lines next to the home bucket, and a lookup meeting such a bucket pays one more random line,
// not a walk through all of them.  Plain mode's home bucket already is a hash of the key, so its
// sequence is purely linear.
TBK_HD uint32_t tbk_next_bucket(uint64_t key, TbkMz mz, uint32_t n_buckets, uint32_t b, bool leaving_home) {
    if (leaving_home && mz.w > 0) return tbk_reduce(tbk_mix32(key), n_buckets);
    return b + 1 == n_buckets ? 0 : b + 1;
}

// The paired (hapA | hapB) table the probe kernel reads: bucket b = 16 slots = 128 bytes.
struct TbkPairView {
    const uint64_t *slots;  // n_buckets * 16
    uint32_t n_buckets;
    TbkMz mz;               // bucket selection
    uint32_t guests;        // see TbkTableView
    uint32_t over_mask;     // short keys (TBK_FLAG_SHORT): slots of the overflow table behind the lines, minus one (0: none)
};

// ---- entry layout: a run of overlapping list k-mers is stored once ------------------------------------------------
// Real find-unique-kmers output (find_unique_kmers.py:200-233) is the k overlapping k-mers around every variant, and
// those that sample the same m-mer occurrence - 4.3 consecutive windows on average at w = 6 - land in one bucket as
// 4.3 separate keys, in both lists at once: the fronts overflow and the probe needs whole lines.
// All of them are windows of ONE stretch of sequence: the sampled m-mer with up to FL = o + w - 1 bases on either side.
// The entry layout stores that stretch once per (m-mer occurrence, list):
//     lo word   the canonical m-mer (m <= 16 bases)
//     hi word   bits [0, 2 FL)       left flank  - context bases 0 .. FL-1, the base next to the m-mer highest
//               bits [2 FL, 4 FL)    right flank - context bases FL+m .. FL+m+FL-1
//               bits [4 FL, +w)      V: bit p set = "the k-mer whose sampled m-mer sits at span position p is in the list"
//               bit 30               the list: 0 hapA, 1 hapB
//               bit 31               a flag of the slot's place in its line, no part of the entry
// in the orientation in which the m-mer is canonical.  The k-mer at span position p is its m-mer plus the o + p bases
// before it and the (w - 1 - p) + o bases after it: in the flank field those are the CONTIGUOUS 2 (k - m) bits from bit
// 2 (w - 1 - p) on (the left flank's top bases, then the right flank's low ones).  A window with oriented k-mer K and
// position p therefore matches an entry iff   lo == m-mer   and   (hi ^ khi) & mhi == 0,   where khi holds K's flank
// bases at that place plus the bit V[p], and mhi covers exactly those bits: one v_bfi and one 64-bit compare per slot
// (expected = (cm, bfi(mhi, khi, hi))); which list the hit counts for is the entry's bit 30.  Bases of an entry outside
// every valid window's extent are zero and never looked at.  Keys of ONE list that share an m-mer share an entry as long
// as their flanks agree where both define them; otherwise they are separate entries of the bucket.  EMPTY is 0 (no V bit:
// matches nothing).  A window that may not match (a byte outside ACGT, a read end) asks for the m-mer 0xFFFFFFFF, which
// no entry holds: T x 16 is not canonical (its reverse complement A x 16 = 0 is smaller), and shorter m-mers stay below.
// A line is 16 slots that BOTH lists fill in order (the lists are disjoint - hapB keys that hapA holds are left out - so
// a window matches at most one entry of its line); the probe's window loop reads the first four, 32 bytes, with two
// lanes per window.  Bit 31 of slot 3: "more than four entries in this line"; of slot 15: "an entry went past this line"
// (second-choice bucket by a hash of the m-mer, then linear).
// A variant's k-mers cost one slot per bucket instead of four to five: the haplotype-shaped lists of the bench shrink
// from 6.0e8 keys to 1.6e8 entries.  Needs m <= 16 and 4 FL + w <= 30: k = 21 .. 23 with the span of six m-mers,
// k = 24, 25 with shorter ones; longer k-mers take wide entries (below).
#define TBK_FLAG_ENTRY 4u    // `guests` word of the views: the paired table is in entry layout
#define TBK_ENTRY_NO_MMER 0xFFFFFFFFu   // the m-mer an invalid window asks for
#define TBK_ENTRY_FLAG 0x80000000u
#define TBK_ENTRY_HAPB 0x40000000u

struct TbkEntryGeom {
    int fl;      // flank bases kept on each side of the m-mer: o + w - 1
    int fbits;   // bits of one window's flank bases: 2 (k - m)
    int vshift;  // first V bit: 4 fl
};

TBK_HD bool tbk_entry_geom(int k, TbkMz z, TbkEntryGeom *g) {
    if (z.w < 2 || z.t <= 0 || z.m > 16 || z.m < 8) return false;
    const int fl = z.o + z.w - 1;
    if (4 * fl + z.w > 30) return false;
    g->fl = fl; g->fbits = 2 * (k - z.m); g->vshift = 4 * fl;
    return true;
}

struct TbkEntryKey { uint32_t cm, khi, mhi; };

// what a window asks an entry: `oriented` = its k-mer read in the orientation in which the sampled m-mer is canonical,
// pos = that m-mer's position in the span as `oriented` reads
TBK_HD TbkEntryKey tbk_entry_key(uint64_t oriented, TbkMz z, TbkEntryGeom g, int pos) {
    const int a = 2 * (z.o + pos);
    const uint32_t mmask = z.m == 16 ? 0xFFFFFFFFu : ((1u << (2 * z.m)) - 1u);
    TbkEntryKey e;
    e.cm = (uint32_t)(oriented >> a) & mmask;
    const uint32_t low = (uint32_t)oriented & ((1u << a) - 1u);                 // the o + pos bases before the m-mer
    const uint32_t high = a + 2 * z.m >= 64 ? 0u : (uint32_t)(oriented >> (a + 2 * z.m));  // the bases after it
    const uint32_t fw = low | (high << a);
    const int sh = 2 * (z.w - 1 - pos);
    const uint32_t vbit = 1u << (g.vshift + pos);
    e.khi = (fw << sh) | vbit;
    e.mhi = (((1u << g.fbits) - 1u) << sh) | vbit;
    return e;
}

TBK_HD bool tbk_entry_match(uint64_t slot, TbkEntryKey e) {
    return (uint32_t)slot == e.cm && ((((uint32_t)(slot >> 32)) ^ e.khi) & e.mhi) == 0;
}

// the flank bits an entry defines: the union of its valid windows' extents
TBK_HD uint32_t tbk_entry_defined(uint32_t hi, TbkMz z, TbkEntryGeom g) {
    uint32_t d = 0;
    for (int p = 0; p < z.w; p++)
        if ((hi >> (g.vshift + p)) & 1u) d |= ((1u << g.fbits) - 1u) << (2 * (z.w - 1 - p));
    return d;
}

// may the window `e` (a key of list `hapb`) join the entry in `slot`?  Same m-mer, same list, and the flanks agree wherever both define them.
TBK_HD bool tbk_entry_compatible(uint64_t slot, TbkEntryKey e, uint32_t hapb, TbkMz z, TbkEntryGeom g) {
    if ((uint32_t)slot != e.cm) return false;
    const uint32_t hi = (uint32_t)(slot >> 32);
    if (((hi & TBK_ENTRY_HAPB) != 0) != (hapb != 0)) return false;
    const uint32_t mine = e.mhi & ((1u << g.vshift) - 1u);  // my flank bits
    return ((hi ^ e.khi) & mine & tbk_entry_defined(hi, z, g)) == 0;
}

TBK_HD uint32_t tbk_entry_bucket(uint32_t cm, uint32_t n_buckets) { return tbk_reduce(tbk_mmer_hash(cm), n_buckets); }
// an entry that finds the line's sixteen slots taken goes to a second-choice bucket (a hash of the m-mer: every
// window that asks for this m-mer follows the same path), then linearly on
TBK_HD uint32_t tbk_entry_next_bucket(uint32_t cm, uint32_t n_buckets, uint32_t b, bool leaving_home) {
    if (leaving_home) return tbk_reduce(tbk_mix32((uint64_t)cm ^ 0xA5A5A5A500000000ull), n_buckets);
    return b + 1 == n_buckets ? 0 : b + 1;
}

// Orientation(s) of a list key for the m-mer at span position p as the key stands: n = 1, or 2 when that m-mer is its
// own reverse complement (a lookup may then read it either way round).  Returns n; out[i] = what to ask / store.
TBK_HD uint64_t tbk_revcomp_packed(uint64_t x, int k);
TBK_HD int tbk_entry_orientations(uint64_t key, int k, TbkMz z, TbkEntryGeom g, int p, TbkEntryKey *out) {
    const uint32_t mmask = z.m == 16 ? 0xFFFFFFFFu : ((1u << (2 * z.m)) - 1u);
    const uint32_t x = (uint32_t)(key >> (2 * (z.o + p))) & mmask;
    const uint32_t y = tbk_revcomp32(x, z.m);
    int n = 0;
    if (x <= y) out[n++] = tbk_entry_key(key, z, g, p);
    if (x >= y) out[n++] = tbk_entry_key(tbk_revcomp_packed(key, k), z, g, z.w - 1 - p);
    return n;
}

// Which list holds the window (build-time scans, tests, the CPU model): -1 none, 0 hapA, 1 hapB.  Any one of a key's
// (orientation, tied position) forms is stored iff the key is.
TBK_HD int tbk_entry_lookup_one(const uint64_t *slots, uint32_t n_buckets, TbkEntryKey e) {
    uint32_t b = tbk_entry_bucket(e.cm, n_buckets);
    for (uint32_t walked = 0; walked <= n_buckets; walked++) {
        const uint64_t *line = slots + (uint64_t)b * 16;
        for (uint32_t s = 0; s < 16; s++) {
            const uint64_t v = line[s];
            if ((v & ~((uint64_t)TBK_ENTRY_FLAG << 32)) == 0) return -1;  // first empty slot of the line
            if (tbk_entry_match(v, e)) return (int)((v >> 62) & 1ull);
        }
        if (!((line[15] >> 63) & 1ull)) return -1;                        // nothing went past this line
        b = tbk_entry_next_bucket(e.cm, n_buckets, b, walked == 0);
    }
    return -1;
}

// ---- wide entries: the entry layout for k-mers too long for 64 bits of context (k up to 32) -----------------------
// A k-mer plus its neighbours under one sampled m-mer is k + w - 1 bases: 26 at k = 21, 38 at k = 31 - more than a slot.
// A WIDE entry is two slots, one 16-byte piece of the line (what one lane loads):
//     word 0   bits [0, 2m)  the canonical m-mer, m up to 24;  bit 62  "taken";  bit 63  lock (build only)
//     word 1   bits [0, 4 FL)  the flank field, laid out as in a narrow entry;  bits [4 FL, +w)  V  (4 FL + w <= 48);
//              bit 62  the list (0 hapA, 1 hapB);  bit 63  the piece's flag
// and a window matches iff word 0 is its m-mer | taken and (word 1 ^ k1) & m1 == 0 - two v_bfi and two 64-bit compares;
// which list the hit counts for is the entry's bit 62.  The m-mer has room to be LONGER than 16 bases, and needs to be:
// mod-sampling only ever samples m-mers that hold one of the span's smallest t-mers at offset 0 or w - about an eighth
// of all m-mers - so 2 x 2e8 entries over 16-mers meet in the same buckets however roomy the table (a quarter of them lay
// behind a front); 18-mers (the default) have sixteen times the room.  A line is eight pieces that BOTH lists fill in
// order, the front is its first two (32 bytes, pair-cooperative probe as for narrow entries).  Bit 63 of piece 1's
// word 1: "more than two entries in this line"; of piece 7's: "an entry went past this line".  There is no 128-bit
// compare-and-swap: an insert takes the piece's lock (bit 63 of word 0) to check that list and flanks agree with the
// entry's and ORs its bits in.  The lists are disjoint (hapB keys that hapA holds are left out), so a window matches at
// most one entry of its line.
#define TBK_FLAG_WIDE 8u     // `guests` word: the entry table holds wide entries (with TBK_FLAG_ENTRY)
#define TBK_WENTRY_TAKEN 0x4000000000000000ull
#define TBK_WENTRY_LOCK 0x8000000000000000ull
#define TBK_WENTRY_FLAG 0x8000000000000000ull
#define TBK_WENTRY_HAPB 0x4000000000000000ull

TBK_HD bool tbk_wentry_geom(int k, TbkMz z, TbkEntryGeom *g) {
    if (z.w < 2 || z.t <= 0 || z.m > 24 || z.m < 8) return false;
    const int fl = z.o + z.w - 1;
    // 48 bits of flanks + V (the probe's queues carry the upper 16 of them beside the m-mer's upper 16), and a window's
    // flank bases in 32 bits: k - m <= 16
    if (4 * fl + z.w > 48 || 2 * (k - z.m) > 32) return false;
    g->fl = fl; g->fbits = 2 * (k - z.m); g->vshift = 4 * fl;
    return true;
}

struct TbkWideKey { uint64_t cm, k1, m1; };

TBK_HD TbkWideKey tbk_wentry_key(uint64_t oriented, TbkMz z, TbkEntryGeom g, int pos) {
    const int a = 2 * (z.o + pos);
    const uint64_t mmask = (1ull << (2 * z.m)) - 1ull;
    TbkWideKey e;
    e.cm = (oriented >> a) & mmask;
    const uint64_t low = oriented & ((1ull << a) - 1ull);
    const uint64_t high = a + 2 * z.m >= 64 ? 0ull : (oriented >> (a + 2 * z.m));
    const uint64_t fw = low | (high << a);                      // 2 (k - m) <= 32 bits
    const int sh = 2 * (z.w - 1 - pos);
    const uint64_t vbit = 1ull << (g.vshift + pos);
    const uint64_t fmask = (1ull << g.fbits) - 1ull;
    e.k1 = (fw << sh) | vbit;
    e.m1 = (fmask << sh) | vbit;
    return e;
}

// (whichever list the entry belongs to: bit 62 of w1 says)
TBK_HD bool tbk_wentry_match(uint64_t w0, uint64_t w1, TbkWideKey e) {
    return (w0 & ~TBK_WENTRY_LOCK) == (e.cm | TBK_WENTRY_TAKEN) && ((w1 ^ e.k1) & e.m1) == 0;
}

TBK_HD uint64_t tbk_wentry_defined(uint64_t w1, TbkMz z, TbkEntryGeom g) {
    uint64_t d = 0;
    const uint64_t fmask = (1ull << g.fbits) - 1ull;
    for (int p = 0; p < z.w; p++)
        if ((w1 >> (g.vshift + p)) & 1ull) d |= fmask << (2 * (z.w - 1 - p));
    return d;
}

// may the window `e` of list `hapb` join this entry?  Same m-mer, same list, and the flanks agree wherever both define them.
TBK_HD bool tbk_wentry_compatible(uint64_t w0, uint64_t w1, TbkWideKey e, uint32_t hapb, TbkMz z, TbkEntryGeom g) {
    if ((w0 & ~TBK_WENTRY_LOCK) != (e.cm | TBK_WENTRY_TAKEN) || ((w1 >> 62) & 1ull) != (uint64_t)(hapb ? 1 : 0)) return false;
    const uint64_t mine = e.m1 & ((1ull << g.vshift) - 1ull);
    return ((w1 ^ e.k1) & mine & tbk_wentry_defined(w1, z, g)) == 0;
}

// the bucket of a wide entry's m-mer: the placement half of the 64-bit m-mer hash (as the key layouts' long m-mers)
TBK_HD uint32_t tbk_wentry_bucket(uint64_t cm, uint32_t n_buckets) { return tbk_reduce((uint32_t)tbk_mmer_hash64(cm), n_buckets); }
TBK_HD uint32_t tbk_wentry_next_bucket(uint64_t cm, uint32_t n_buckets, uint32_t b, bool leaving_home) {
    if (leaving_home) return tbk_reduce(tbk_mix32(cm ^ 0xA5A5A5A5A5A5A5A5ull), n_buckets);
    return b + 1 == n_buckets ? 0 : b + 1;
}

TBK_HD int tbk_wentry_orientations(uint64_t key, int k, TbkMz z, TbkEntryGeom g, int p, TbkWideKey *out) {
    const uint64_t mmask = (1ull << (2 * z.m)) - 1ull;
    const uint64_t x = (key >> (2 * (z.o + p))) & mmask;
    const uint64_t y = tbk_revcomp64(x, z.m);
    int n = 0;
    if (x <= y) out[n++] = tbk_wentry_key(key, z, g, p);
    if (x >= y) out[n++] = tbk_wentry_key(tbk_revcomp_packed(key, k), z, g, z.w - 1 - p);
    return n;
}

// -1: no entry holds the window; 0 / 1: an entry of hapA / hapB does
TBK_HD int tbk_wentry_lookup_one(const uint64_t *slots, uint32_t n_buckets, TbkWideKey e) {
    uint32_t b = tbk_wentry_bucket(e.cm, n_buckets);
    for (uint32_t walked = 0; walked <= n_buckets; walked++) {
        const uint64_t *line = slots + (uint64_t)b * 16;
        uint64_t last1 = 0;
        for (uint32_t i = 0; i < 8; i++) {
            const uint64_t w0 = line[2 * i], w1 = line[2 * i + 1];
            if (!(w0 & TBK_WENTRY_TAKEN)) return -1;    // first empty piece of the line
            if (tbk_wentry_match(w0, w1, e)) return (int)((w1 >> 62) & 1ull);
            last1 = w1;
        }
        if (!(last1 >> 63)) return -1;                  // nothing went past this line
        b = tbk_wentry_next_bucket(e.cm, n_buckets, b, walked == 0);
    }
    return -1;
}

// ---- short keys: a list k-mer in 32 bits, exactly (lists whose keys do NOT merge into entries) -----------------------
// BASELINE's synthetic lists are uniform random k-mers: no two of them overlap, an entry would hold one key.  For such
// lists the key layout spends 8 bytes a key and its probe reads 64 bytes of a line with four lanes per window.  But a
// window that has found its bucket already knows most of its key: the bucket is a function of the sampled m-mer, and
// what is left is
//     r      which of the few m-mer hashes that map to this bucket it is: the low product word of tbk_reduce
//            (hash x n_buckets) shifted right by rshift = floor(log2 n_buckets) - hashes of one bucket are consecutive,
//            their products n_buckets apart, so distinct hashes give distinct r; tbk_mmer_hash is a bijection of the
//            32-bit m-mers: (bucket, r) <-> m-mer, exactly;  32 - rshift bits
//     pos    the m-mer's position in the span, 3 bits
//     flank  the k - m bases around the m-mer in the orientation in which the m-mer is canonical (as tbk_entry_key)
// - at k = 21 with 2^27.8 buckets that is 5 + 3 + 10 bits.  A SHORT slot is one 32-bit word:
//     bits [0, 29)  flank | pos << fbits | r << (fbits + 3);  bit 29  taken;  bit 30  the list (1 = hapB);  bit 31  a flag
// A line is 32 slots; both lists fill slots 0..6 and then 8..31 in order, and the probe's window loop reads the first eight
// (32 bytes, two lanes per window as for entries, four compares of 32 bits per lane and list).  Slot 7 is no key but the
// line's SUMMARY of what lies behind its front (round 5): bit 31 "slots behind the front are in use", and in bits 0..28 one
// bit per word stored there or in the overflow table (tbk_short_filter_bit of the word).  A window that misses in the front of
// a flagged line looks behind it - the line's other 96 bytes, fetched a second time: by then the line has long left the L2 -
// only when its own word's bit is set: 99.8 % of a read's windows are in no list, and before the summary every one of them
// that met a flagged line (1.4 % of all windows on BASELINE's lists) paid that second fetch; now one in ten of those does.
// A summary never equals a word asked (bit 29, "taken", is clear in it and set in every word).  Bit 31 of slot 31: "a key of
// this bucket is in the overflow table" - open addressing over 64-bit canonical keys (| list << 63) behind the lines, for
// the handful of keys whose m-mer's bucket holds more than 31.  EMPTY is 0; a window that may not match asks 0xFFFFFFFF
// (every compare strips the slot's bit 31 first).  Needs m <= 16 and fbits + 3 + 32 - rshift <= 29:
// k = 21 from 65536 buckets on, k = 23 from 2^20, k = 25 from 2^24.
#define TBK_FLAG_SHORT 16u   // `guests` word of the views: the paired table holds short keys
#define TBK_SHORT_FLAG 0x80000000u
#define TBK_SHORT_HAPB 0x40000000u
#define TBK_SHORT_TAKEN 0x20000000u
#define TBK_SHORT_NONE 0xFFFFFFFFu
#define TBK_SHORT_EMPTY64 0xFFFFFFFFFFFFFFFFull   // empty slot of the overflow table
#define TBK_SHORT_SUMMARY 7                       // the slot of a line that holds its summary
#define TBK_SHORT_FRONT_KEYS 7                    // keys in the 32 bytes the window loop reads

// the bit of its line's summary (bits 0..28 of slot 7) that a word stored behind the front sets
TBK_HD uint32_t tbk_short_filter_bit(uint32_t word) {
    const uint32_t h = (word ^ (word >> 5) ^ (word >> 11)) & 0xFFFFu;
    return 1u << ((h * 29u) >> 16);
}

struct TbkShortGeom {
    int fbits;   // 2 (k - m)
    int rshift;  // floor(log2 n_buckets)
};

TBK_HD bool tbk_short_geom(int k, TbkMz z, uint32_t n_buckets, TbkShortGeom *g) {
    if (z.w < 2 || z.w > 8 || z.t <= 0 || z.m > 16 || z.m < 8 || n_buckets < 2 || k > 31) return false;
    int rs = 0;
    while (rs < 31 && (2u << rs) <= n_buckets) rs++;
    const int fbits = 2 * (k - z.m);
    if (fbits + 3 + (32 - rs) > 29) return false;
    g->fbits = fbits; g->rshift = rs;
    return true;
}

// the fewest buckets a table of short keys may have for k and the span
TBK_HD uint32_t tbk_short_min_buckets(int k, TbkMz z) {
    const int need = 2 * (k - z.m) + 3 + 32 - 29;  // rshift >= this
    return need <= 1 ? 2u : need >= 31 ? 0u : (1u << need);
}

struct TbkShortKey { uint32_t word, bucket; };

// what a window asks: `oriented` and pos as for tbk_entry_key
TBK_HD TbkShortKey tbk_short_key(uint64_t oriented, TbkMz z, TbkShortGeom g, int pos, uint32_t n_buckets) {
    const int a = 2 * (z.o + pos);
    const uint32_t mmask = z.m == 16 ? 0xFFFFFFFFu : ((1u << (2 * z.m)) - 1u);
    const uint32_t cm = (uint32_t)(oriented >> a) & mmask;
    const uint32_t low = (uint32_t)oriented & ((1u << a) - 1u);
    const uint32_t high = a + 2 * z.m >= 64 ? 0u : (uint32_t)(oriented >> (a + 2 * z.m));
    const uint64_t prod = (uint64_t)tbk_mmer_hash(cm) * (uint64_t)n_buckets;
    TbkShortKey e;
    e.bucket = (uint32_t)(prod >> 32);
    e.word = (low | (high << a)) | ((uint32_t)pos << g.fbits) | (((uint32_t)prod >> g.rshift) << (g.fbits + 3)) | TBK_SHORT_TAKEN;
    return e;
}

TBK_HD int tbk_short_orientations(uint64_t key, int k, TbkMz z, TbkShortGeom g, int p, uint32_t n_buckets, TbkShortKey *out) {
    const uint32_t mmask = z.m == 16 ? 0xFFFFFFFFu : ((1u << (2 * z.m)) - 1u);
    const uint32_t x = (uint32_t)(key >> (2 * (z.o + p))) & mmask;
    const uint32_t y = tbk_revcomp32(x, z.m);
    int n = 0;
    if (x <= y) out[n++] = tbk_short_key(key, z, g, p, n_buckets);
    if (x >= y) out[n++] = tbk_short_key(tbk_revcomp_packed(key, k), z, g, z.w - 1 - p, n_buckets);
    return n;
}

TBK_HD uint32_t tbk_short_over_home(uint64_t canonical, uint32_t over_mask) {
    uint64_t h = canonical * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return (uint32_t)(h * 0xBF58476D1CE4E5B9ull >> 32) & over_mask;
}

// -1 none, 0 hapA, 1 hapB.  lines = the table's n_buckets x 32 words; over = the overflow table behind them
// (over_mask + 1 slots; over_mask = 0: none).  as_window != 0: as the probe's window loop decides - behind the front only
// when the summary has the word's bit (build-time scans and tests pass 0 and look everywhere: both must agree).
TBK_HD int tbk_short_lookup_one(const uint32_t *lines, uint32_t n_buckets, const uint64_t *over, uint32_t over_mask, TbkShortKey e, uint64_t canonical, int as_window = 0) {
    const uint32_t *line = lines + (uint64_t)e.bucket * 32;
    for (uint32_t s = 0; s < TBK_SHORT_FRONT_KEYS; s++) {
        const uint32_t v = line[s] & ~TBK_SHORT_FLAG;
        if (v == 0) break;  // the line's keys end here (what its bucket holds beyond - a capped line of the tests - is in the overflow table, and flagged)
        if ((v & ~TBK_SHORT_HAPB) == e.word) return (int)((v >> 30) & 1u);
    }
    const uint32_t summary = line[TBK_SHORT_SUMMARY];
    if (!(summary & TBK_SHORT_FLAG)) return -1;                            // nothing behind the front
    if (as_window && !(summary & tbk_short_filter_bit(e.word))) return -1;  // ... or not this word
    for (uint32_t s = TBK_SHORT_SUMMARY + 1; s < 32; s++) {
        const uint32_t v = line[s] & ~TBK_SHORT_FLAG;
        if (v == 0) break;
        if ((v & ~TBK_SHORT_HAPB) == e.word) return (int)((v >> 30) & 1u);
    }
    if (!(line[31] & TBK_SHORT_FLAG) || !over_mask) return -1;
    for (uint32_t i = tbk_short_over_home(canonical, over_mask), walked = 0; walked <= over_mask; walked++, i = (i + 1) & over_mask) {
        const uint64_t v = over[i];
        if (v == TBK_SHORT_EMPTY64) return -1;
        if ((v & ~(1ull << 63)) == canonical) return (int)(v >> 63);
    }
    return -1;
}

// ---- full keys: 64-bit list k-mers in the entry kernels' line (lists that do not merge, k too long for short keys) ----
// Uniform lists of 26- to 31-mers (and of 23- / 25-mers in tables whose m-mers outgrow 16 bases) do not merge into entries and
// do not fit short keys; until round 5 they stayed in the key layout: 8 + 8 slots per line, a 64-byte front read by four
// lanes, 94-96 registers, five waves per SIMD, 93-100 bytes of device memory per key.  FULL keys put them into the entry
// kernels' line shape: 16 slots, a 32-byte front read by two lanes per window.  A slot =
//     bits [0, 62)  the canonical k-mer, INVERTED (k <= 31: a key is below 2^62) - so that an empty slot is 0, the inverse of
//                   T x 31, which is never canonical (for k < 31 no key reaches it)
//     bit 62        the list (1 = hapB);   bit 63  a flag of the slot's place in its line, no part of the key
// and a window matches a slot iff (slot & KEY) == ~canonical & KEY: one and, one 64-bit compare.  A window that may not
// match asks for the inverse of TBK_FULL_NOKEY ("GTT...T": never canonical either).
// Both lists fill slots 0, 1, 2 and then 4 .. 15 in order.  SLOT 3 IS NO KEY but the line's summary of what lies behind its
// front (as slot 7 of a short-key line): bit 63 "slots behind the front are in use", bits 0..61 one bit per key stored there
// or sent on past the line (tbk_full_filter_bit).  Three keys in the front is few - at two keys per line a fifth of the
// lines holds more - but a window looks behind a front only when the summary has its own key's bit: 99.8 % of a read's
// windows are in no list, and of those that meet a flagged line one in ~25 passes.  The probe never compares slot 3 (a
// 62-bit summary could equal a key).  Bit 63 of slot 15: "a key went past this line" (second-choice bucket by a hash of the
// key, then linear: tbk_next_bucket).  The bucket is the sampled m-mer's, by the placement half of the 64-bit m-mer hash
// whatever m is (tbk_wentry_bucket: the probe's m-mer arithmetic is the wide entries'); a key is stored under the bucket of
// every position that attains the smallest t-mer rank, as in the key layouts.  List lines that are not canonical are dead in
// the reference (stored verbatim, c/kmers.c:113; looked up as min(fwd, rc), c/kmers.c:255) and are not stored; hapB keys
// that hapA's list holds are left out (c/kmers.c:291-294), so a window matches at most one slot of its line.
#define TBK_FLAG_FULL 32u    // `guests` word of the views: the paired table holds full keys
#define TBK_FULL_FLAG 0x8000000000000000ull
#define TBK_FULL_HAPB 0x4000000000000000ull
#define TBK_FULL_KEY 0x3FFFFFFFFFFFFFFFull
#define TBK_FULL_NOKEY 0x3FFFFFFFFFFFFFFEull   // what an invalid window looks up (stored form: 1)
#define TBK_FULL_SUMMARY 3                     // the slot of a line that holds its summary
#define TBK_FULL_FRONT_KEYS 3                  // keys in the 32 bytes the window loop reads

TBK_HD bool tbk_full_geom(int k, TbkMz z) { return k >= 3 && k <= 31 && z.w >= 2 && z.w <= 8 && z.t > 0 && z.m >= 8 && z.m <= 24 && z.t <= 16; }

// the stored form of a canonical key (without list bit and flag)
TBK_HD uint64_t tbk_full_word(uint64_t canonical) { return ~canonical & TBK_FULL_KEY; }

// the bit of its line's summary (bits 0..61 of slot 3) that a key stored behind the front sets; `word` = tbk_full_word
TBK_HD uint64_t tbk_full_filter_bit(uint64_t word) {
    const uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
    const uint32_t h = (lo ^ (lo >> 7) ^ (lo >> 17) ^ hi ^ (hi >> 9)) & 0xFFFFu;
    return 1ull << ((h * 62u) >> 16);
}

// the canonical m-mer of `key` at span position p (m <= 24: 64-bit arithmetic)
TBK_HD uint64_t tbk_full_mmer(uint64_t key, TbkMz z, int p) {
    const uint64_t mmask = (1ull << (2 * z.m)) - 1ull;
    const uint64_t x = (key >> (2 * (z.o + p))) & mmask;
    const uint64_t y = tbk_revcomp64(x, z.m);
    return x < y ? x : y;
}

// -1: neither list holds the canonical key; 0 / 1: hapA's / hapB's does.  `bucket`: the home bucket of one of the key's forms.
// as_window != 0: as the probe's window loop decides (behind the front of the HOME line only when its summary has the key's bit).
TBK_HD int tbk_full_lookup_one(const uint64_t *slots, uint32_t n_buckets, uint64_t canonical, uint32_t bucket, TbkMz z, int as_window = 0) {
    const uint64_t want = tbk_full_word(canonical);
    uint32_t b = bucket;
    for (uint32_t walked = 0; walked <= n_buckets; walked++) {
        const uint64_t *line = slots + (uint64_t)b * 16;
        for (uint32_t s = 0; s < 16; s++) {
            if (s == TBK_FULL_SUMMARY) {
                if (!(line[s] >> 63)) return -1;                                                    // nothing behind this front
                if (as_window && walked == 0 && !(line[s] & tbk_full_filter_bit(want))) return -1;   // ... or not this key
                continue;
            }
            const uint64_t v = line[s];
            if ((v & ~TBK_FULL_FLAG) == 0) return -1;   // first empty slot of the line
            if ((v & TBK_FULL_KEY) == want) return (int)((v >> 62) & 1ull);
        }
        if (!(line[15] >> 63)) return -1;               // nothing went past this line
        b = tbk_next_bucket(canonical, z, n_buckets, b, walked == 0);
    }
    return -1;
}

// ---- synthetic key sequence (bench inputs) ---------------------------------
// Key i of `seed` is a k-mer whose k-2 middle bases are a bijective scramble of i over
// 2(k-2) bits (distinct i -> distinct k-mer) and whose end bases (b0, b_{k-1}) satisfy
// b0 + b_{k-1} < 3, which makes the packed integer strictly smaller than its reverse
// complement's (the top base pair decides: b_{k-1} < 3 - b0), i.e. the k-mer is canonical.
// Requires 3 <= k <= 32 and i < 4^(k-2).
TBK_HD uint64_t tbk_synth_key(uint64_t seed, uint64_t i, int k) {
    const int mb = 2 * (k - 2);
    const uint64_t mm = mb >= 64 ? ~0ull : ((1ull << mb) - 1ull);
    uint64_t x = (i + seed * 0x9E3779B97F4A7C15ull) & mm;
    const int s1 = mb / 2 + 1 > 63 ? 63 : mb / 2 + 1;
    x = (x * 0xD1342543DE82EF95ull) & mm;
    x ^= x >> s1;
    x = (x * 0xAF251AF3B0F025B5ull) & mm;
    x ^= x >> s1;
    x = (x * 0x9E3779B97F4A7C15ull) & mm;
    x ^= x >> s1;
    // ends: 6 admissible (b0, b_{k-1}) pairs, picked by a hash of i
    uint64_t e = (i ^ seed) * 0xC6A4A7935BD1E995ull;
    e ^= e >> 29;
    const uint32_t pick = (uint32_t)((e >> 11) % 6u);
    // pairs: (0,0) (0,1) (0,2) (1,0) (1,1) (2,0)
    const uint32_t b0 = pick < 3 ? 0u : (pick < 5 ? 1u : 2u);
    const uint32_t bl = pick < 3 ? pick : (pick < 5 ? pick - 3u : 0u);
    return (uint64_t)b0 | (x << 2) | ((uint64_t)bl << (2 * (k - 1)));
}

// Reverse complement of a packed k-mer (used by generators and host utilities).
TBK_HD uint64_t tbk_revcomp_packed(uint64_t x, int k) {
    uint64_t y = ~x;
    // reverse the order of the 32 two-bit groups
    y = ((y >> 2) & 0x3333333333333333ull) | ((y & 0x3333333333333333ull) << 2);
    y = ((y >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((y & 0x0F0F0F0F0F0F0F0Full) << 4);
    y = ((y >> 8) & 0x00FF00FF00FF00FFull) | ((y & 0x00FF00FF00FF00FFull) << 8);
    y = ((y >> 16) & 0x0000FFFF0000FFFFull) | ((y & 0x0000FFFF0000FFFFull) << 16);
    y = (y >> 32) | (y << 32);
    return y >> (64 - 2 * k);
}

// splitmix64 step, the counter-based PRNG of the read generator.
TBK_HD uint64_t tbk_splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// ---- synthetic haplotypes (bench inputs shaped like real trio-binning lists) -------------
// An implicit random genome (base at position p = a hash of p) and two haplotypes that each
// differ from it by SNPs at `snp24 / 2^24` per base.  The k-mers covering a position where the
// haplotypes differ are the "haplotype-unique" lists: they come in runs of up to k overlapping
// k-mers that share a handful of minimizers, hapA's and hapB's runs at the same loci - the
// shape of real find-unique-kmers output, unlike tbk_synth_key's uniform keys.
// Bits 24..31 of the shape word add repeats: rep8 / 256 of the genome's 8192-base blocks are copies of
// one of 16 family sequences, each copy diverged from its family at 2 % of its positions (young
// interspersed repeats: with 10 % of a 3 Gbase genome that is ~2300 near-identical copies per family,
// whose variant k-mers all share the family's m-mers - the crowded-bucket case of a real list).
TBK_HD void tbk_hap_bases(uint64_t seed, uint64_t p, uint32_t shape, uint32_t &base_a, uint32_t &base_b) {
    const uint32_t snp24 = shape & 0xFFFFFFu, rep8 = shape >> 24;
    const uint64_t r = tbk_splitmix(seed ^ (p * 0x9E3779B97F4A7C15ull));
    uint32_t base = (uint32_t)r & 3u;
    if (rep8) {
        const uint64_t hb = tbk_splitmix(seed ^ 0xB10CB10CB10Cull ^ ((p >> 13) * 0xD6E8FEB86659FD93ull));
        if ((uint32_t)(hb & 0xFFu) < rep8 && (uint32_t)((r * 0x94D049BB133111EBull) >> 56) >= 5u) {
            const uint64_t fam = (hb >> 8) & 15ull;
            base = (uint32_t)tbk_splitmix(seed ^ ((((fam + 1ull) << 48) | (p & 8191ull)) * 0xA0761D6478BD642Full)) & 3u;
        }
    }
    const uint32_t draw_a = (uint32_t)(r >> 4) & 0xFFFFFFu, draw_b = (uint32_t)(r >> 28) & 0xFFFFFFu;
    const uint32_t alt_a = (base + 1u + (uint32_t)((r >> 52) & 0xFFu) % 3u) & 3u;
    const uint32_t alt_b = (base + 1u + (uint32_t)((r >> 56) & 0xFFu) % 3u) & 3u;
    base_a = draw_a < snp24 ? alt_a : base;
    base_b = draw_b < snp24 ? alt_b : base;
}

