// CPU model of the full-key layout (tbk_common.h "full keys"): a sequential build with the insert rule of
// tbk_full_insert_kernel - 64-bit keys, sixteen slots to a line, slot 3 the line's summary, a key that finds its line full
// sent on by tbk_next_bucket - then the claims the probe kernel relies on are checked against plain set membership:
//   * every canonical list key is found through the bucket of each position that attains the smallest t-mer rank, in its own
//     list only; keys outside the lists are not found - including keys that share their sampled m-mer, and so their line,
//     with list keys;
//   * the summary never hides a key: tbk_full_lookup_one(..., as_window = 1) - behind the front of the HOME line only when
//     its summary has the key's bit, which is how the window loop decides - answers like the exhaustive walk (as_window = 0);
//   * a read window asks with a position chosen from the FORWARD strand alone, on either strand, and gets the set's answer;
//   * crowded tables (many keys per line): lines fill, keys walk on to their second-choice bucket and beyond, and are found;
//   * hapA-over-hapB priority (c/kmers.c:291-294): a key of both lists is stored for hapA only.
// Built and run by tests/test_entry_model.py (g++, no GPU; the CPU sanitizer build never reaches this logic otherwise).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_set>
#include <vector>

#include "../../trio_binning_amd/csrc/tbk_common.h"

static uint64_t canon(uint64_t x, int k) { const uint64_t y = tbk_revcomp_packed(x, k); return x < y ? x : y; }

struct Table {
    std::vector<uint64_t> slots;
    uint32_t n_buckets;
    TbkMz z;
    int k;
    uint64_t taken = 0, behind = 0, past = 0;
};

// the home buckets of a canonical key: one per position that attains the smallest t-mer rank
static int buckets_of(const Table &t, uint64_t key, uint32_t *out) {
    const int nt = tbk_mz_positions(t.z);
    uint32_t best = 0xFFFFFFFFu;
    for (int i = 0; i < nt; i++) { const uint32_t r = tbk_tmer_rank(key, t.z, i); best = r < best ? r : best; }
    int n = 0;
    for (int i = 0; i < nt; i++)
        if (tbk_tmer_rank(key, t.z, i) == best) out[n++] = tbk_wentry_bucket(tbk_full_mmer(key, t.z, i % t.z.w), t.n_buckets);
    return n;
}

static int which_list(const Table &t, uint64_t key, uint64_t *disagreements) {  // -1, 0, 1: the same through every bucket, either way of asking
    uint32_t b[64];
    const int n = buckets_of(t, key, b);
    int ans = -2;
    for (int i = 0; i < n; i++) {
        const int whole = tbk_full_lookup_one(t.slots.data(), t.n_buckets, key, b[i], t.z, 0);
        const int as_window = tbk_full_lookup_one(t.slots.data(), t.n_buckets, key, b[i], t.z, 1);
        if (whole != as_window) (*disagreements)++;
        if (ans != -2 && whole != ans) { fprintf(stderr, "the buckets of one key disagree\n"); exit(3); }
        ans = whole;
    }
    return ans;
}

static void insert_key(Table &t, uint32_t half, uint64_t key, bool skip_a) {
    if (key >= TBK_FULL_NOKEY) return;
    if (key != canon(key, t.k)) return;  // a non-canonical list line is dead in the reference (c/kmers.c:113 vs 251-255)
    uint32_t bs[64];
    const int n = buckets_of(t, key, bs);
    const uint64_t word = tbk_full_word(key), fbit = tbk_full_filter_bit(word), listbit = half ? TBK_FULL_HAPB : 0ull;
    for (int i = 0; i < n; i++) {
        uint32_t b = bs[i];
        if (i == 0 && skip_a && tbk_full_lookup_one(t.slots.data(), t.n_buckets, key, b, t.z) == 0) return;
        bool done = false;
        for (uint32_t walked = 0; walked <= t.n_buckets && !done; walked++) {
            uint64_t *line = t.slots.data() + (uint64_t)b * 16;
            for (uint32_t sl = 0; sl < 16 && !done; sl++) {
                if (sl == TBK_FULL_SUMMARY) continue;
                const uint64_t cur = line[sl];
                if ((cur & ~TBK_FULL_FLAG) == 0) {
                    line[sl] = cur | word | listbit;
                    t.taken++;
                    if (sl > TBK_FULL_SUMMARY) { t.behind++; line[TBK_FULL_SUMMARY] |= TBK_FULL_FLAG | fbit; }
                    done = true;
                } else if ((cur & TBK_FULL_KEY) == word) done = true;
            }
            if (!done) {
                line[15] |= TBK_FULL_FLAG;
                line[TBK_FULL_SUMMARY] |= TBK_FULL_FLAG | fbit;
                t.past++;
                b = tbk_next_bucket(key, t.z, t.n_buckets, b, walked == 0);
            }
        }
        if (!done) { fprintf(stderr, "table full\n"); exit(2); }
    }
}

// the bucket the probe kernel asks for the window whose forward k-mer is `fwd`: the position from the forward strand alone
static uint32_t window_bucket(const Table &t, uint64_t fwd, int pick_last_tie) {
    const int nt = tbk_mz_positions(t.z);
    uint32_t best = 0xFFFFFFFFu;
    int x = 0;
    for (int i = 0; i < nt; i++) {
        const uint32_t r = tbk_tmer_rank(fwd, t.z, i);
        if (r < best || (pick_last_tie && r == best)) { best = r; x = i; }
    }
    return tbk_wentry_bucket(tbk_full_mmer(fwd, t.z, x % t.z.w), t.n_buckets);
}

int main(int argc, char **argv) {
    const int k = argc > 1 ? atoi(argv[1]) : 31;
    const int w_want = argc > 2 ? atoi(argv[2]) : 8;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
    const double per_line = argc > 4 ? atof(argv[4]) : 2.0;   // keys per line the table is sized for
    std::mt19937_64 rng(seed);
    Table t;
    t.k = k;
    t.z = tbk_mz_params(k, w_want, 1000000000ull, 0, 1);
    if (!tbk_full_geom(k, t.z) || t.z.t != t.z.m - t.z.w) { printf("k=%d w=%d: no full keys (w=%d m=%d o=%d t=%d)\n", k, w_want, t.z.w, t.z.m, t.z.o, t.z.t); return 0; }
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    const int G = 50000;
    std::vector<uint8_t> ga(G), gb(G);
    for (int i = 0; i < G; i++) ga[i] = (uint8_t)(rng() & 3);
    for (int i = 20000; i < 20400; i++) ga[i] = (uint8_t)((i / 3) & 1);         // low complexity
    for (int i = 0; i < 3000; i++) ga[30000 + i] = ga[5000 + i];               // a repeat
    for (int i = 0; i < 64; i++) ga[40000 + i] = (uint8_t)(i < 32 ? (i & 3) : 3 - ((63 - i) & 3));  // a palindromic stretch
    gb = ga;
    for (int i = 0; i < G; i++) if (rng() % 150 == 0) gb[i] = (uint8_t)((ga[i] + 1 + rng() % 3) & 3);
    auto kmer_at = [&](const std::vector<uint8_t> &g, int i) { uint64_t x = 0; for (int j = 0; j < k; j++) x |= (uint64_t)g[i + j] << (2 * j); return x; };
    std::unordered_set<uint64_t> all_a, all_b;
    for (int i = 0; i + k <= G; i++) { all_a.insert(canon(kmer_at(ga, i), k)); all_b.insert(canon(kmer_at(gb, i), k)); }
    std::vector<uint64_t> list_a, list_b;
    for (uint64_t x : all_a) if (!all_b.count(x)) list_a.push_back(x);
    for (uint64_t x : all_b) if (!all_a.count(x)) list_b.push_back(x);
    for (int i = 0; i < 20000; i++) list_a.push_back(canon(rng() & kmask, k));
    for (int i = 0; i < 20000; i++) list_b.push_back(canon(rng() & kmask, k));
    // crowded m-mers: hundreds of keys around one stretch of sequence, and as many near misses that are NOT in the lists
    std::vector<uint64_t> near_miss;
    for (int c = 0; c < 6; c++) {
        const uint64_t core = rng() & kmask;
        for (int i = 0; i < 400; i++) {
            const int lo = 2 * (int)(rng() % 3), hi = 2 * (k - 1 - (int)(rng() % 3));
            uint64_t x = core ^ ((rng() & 3ull) << lo) ^ ((rng() & 3ull) << hi) ^ ((rng() & 0xFull) << (2 * (int)(rng() % 2)));
            x = canon(x & kmask, k);
            if (i & 1) (c & 1 ? list_b : list_a).push_back(x); else near_miss.push_back(x);
        }
    }
    for (int i = 0; i < 500; i++) { list_b.push_back(list_a[rng() % list_a.size()]); list_a.push_back(list_a[rng() % list_a.size()]); }
    for (int i = 0; i < 500; i++) { const uint64_t x = rng() & kmask; if (x != canon(x, k)) { list_a.push_back(x); list_b.push_back(x); } }
    std::unordered_set<uint64_t> set_a, set_b;
    for (uint64_t x : list_a) if (x == canon(x, k) && x < TBK_FULL_NOKEY) set_a.insert(x);
    for (uint64_t x : list_b) if (x == canon(x, k) && x < TBK_FULL_NOKEY && !set_a.count(x)) set_b.insert(x);
    t.n_buckets = (uint32_t)((double)(list_a.size() + list_b.size()) / per_line) + 16;
    t.slots.assign((size_t)t.n_buckets * 16, 0);
    for (uint64_t x : list_a) insert_key(t, 0, x, false);
    for (uint64_t x : list_b) insert_key(t, 1, x, true);
    uint64_t bad = 0, disagree = 0;
    for (uint64_t x : set_a) if (which_list(t, x, &disagree) != 0) bad++;
    for (uint64_t x : set_b) if (which_list(t, x, &disagree) != 1) bad++;
    for (uint64_t x : near_miss) { const int want = set_a.count(x) ? 0 : set_b.count(x) ? 1 : -1; if (which_list(t, x, &disagree) != want) bad++; }
    for (int i = 0; i < 200000; i++) { const uint64_t x = canon(rng() & kmask, k); const int want = set_a.count(x) ? 0 : set_b.count(x) ? 1 : -1; if (which_list(t, x, &disagree) != want) bad++; }
    uint64_t windows = 0, hits_a = 0, hits_b = 0;
    for (int strand = 0; strand < 2; strand++)
        for (const std::vector<uint8_t> *hap : {&ga, &gb}) {
            std::vector<uint8_t> r(*hap);
            for (int i = 0; i < G; i++) if (rng() % 300 == 0) r[i] = (uint8_t)(rng() & 3);
            if (strand) { std::vector<uint8_t> q(G); for (int i = 0; i < G; i++) q[i] = (uint8_t)(3 - r[G - 1 - i]); r = q; }
            for (int i = 0; i + k <= G; i++) {
                const uint64_t fwd = kmer_at(r, i), key = canon(fwd, k);
                const int which = key >= TBK_FULL_NOKEY ? -1 : tbk_full_lookup_one(t.slots.data(), t.n_buckets, key, window_bucket(t, fwd, (int)(rng() & 1)), t.z, 1);
                if ((which == 0) != (set_a.count(key) != 0) || (which == 1) != (set_b.count(key) != 0)) bad++;
                windows++; hits_a += which == 0; hits_b += which == 1;
            }
        }
    printf("full k=%d w=%d m=%d o=%d t=%d: %llu keys in %llu slots, %u buckets (%.2f keys per line asked for), %llu behind a front, %llu walks past a line; %llu windows, %llu / %llu hits; "
           "as_window disagreements %llu; mismatches %llu\n",
           k, t.z.w, t.z.m, t.z.o, t.z.t, (unsigned long long)(set_a.size() + set_b.size()), (unsigned long long)t.taken, t.n_buckets, per_line, (unsigned long long)t.behind,
           (unsigned long long)t.past, (unsigned long long)windows, (unsigned long long)hits_a, (unsigned long long)hits_b, (unsigned long long)disagree, (unsigned long long)(bad + disagree));
    return bad + disagree ? 1 : 0;
}
