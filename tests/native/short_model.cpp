// CPU model of the short-key layout (tbk_common.h "short keys"): a sequential build with the insert rule of
// tbk_short_insert_kernel, then the claims the probe kernel relies on are checked against plain set membership:
//   * a 32-bit word plus its bucket names a k-mer EXACTLY: every canonical list key is found through each of its (tied
//     position, orientation) forms, in its own list only; keys outside the lists are not found - including keys that share
//     their sampled m-mer, and therefore their line, with list keys;
//   * a read window asks with (orientation by the sampled m-mer, position) computed from the FORWARD strand alone, as the
//     kernel does, on either strand of the same sequence, and gets the set's answer;
//   * lines that overflow their 31 slots send keys to the overflow table and lookups find them there;
//   * the summary in slot 7 never hides a key: a window that asks as the probe's loop does (behind the front only when the
//     summary has its word's bit) gets the same answer as a scan of the whole line;
//   * hapA-over-hapB priority (c/kmers.c:291-294): a key of both lists is stored for hapA only.
// Built and run by tests/test_entry_model.py (g++, no GPU).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_set>
#include <vector>

#include "../../trio_binning_amd/csrc/tbk_common.h"

static uint64_t canon(uint64_t x, int k) { const uint64_t y = tbk_revcomp_packed(x, k); return x < y ? x : y; }

struct Table {
    std::vector<uint32_t> lines;
    std::vector<uint64_t> over;
    uint32_t n_buckets, over_mask;
    TbkMz z;
    TbkShortGeom g;
    int k;
    uint64_t words = 0, behind = 0, past = 0;
};

static int forms_of(const Table &t, uint64_t key, TbkShortKey *out) {
    const int nt = tbk_mz_positions(t.z);
    uint32_t best = 0xFFFFFFFFu;
    for (int i = 0; i < nt; i++) { const uint32_t r = tbk_tmer_rank(key, t.z, i); best = r < best ? r : best; }
    int n = 0;
    for (int i = 0; i < nt; i++) {
        if (tbk_tmer_rank(key, t.z, i) != best) continue;
        n += tbk_short_orientations(key, t.k, t.z, t.g, i % t.z.w, t.n_buckets, out + n);
    }
    return n;
}

static int which_list(const Table &t, uint64_t key) {  // -1, 0, 1 - the same through every form
    TbkShortKey f[64];
    const int n = forms_of(t, key, f);
    int ans = -2;
    for (int i = 0; i < n; i++) {
        const int h = tbk_short_lookup_one(t.lines.data(), t.n_buckets, t.over.data(), t.over_mask, f[i], key);
        if (ans != -2 && h != ans) { fprintf(stderr, "forms of one key disagree\n"); exit(3); }
        ans = h;
    }
    return ans;
}

static void insert_key(Table &t, uint32_t half, uint64_t key, bool skip_a) {
    if (key != canon(key, t.k)) return;  // a non-canonical list line is dead in the reference (c/kmers.c:113 vs 251-255)
    if (skip_a && which_list(t, key) == 0) return;
    TbkShortKey f[64];
    const int n = forms_of(t, key, f);
    for (int i = 0; i < n; i++) {
        uint32_t *line = t.lines.data() + (uint64_t)f[i].bucket * 32;
        bool done = false;
        for (uint32_t s = 0; s < 32 && !done; s++) {
            if (s == TBK_SHORT_SUMMARY) continue;  // the line's summary, no key
            const uint32_t cur = line[s] & ~TBK_SHORT_FLAG;
            if (cur == 0) {
                line[s] |= f[i].word | (half ? TBK_SHORT_HAPB : 0u);
                t.words++;
                if (s > TBK_SHORT_SUMMARY) { line[TBK_SHORT_SUMMARY] |= TBK_SHORT_FLAG | tbk_short_filter_bit(f[i].word); t.behind++; }
                done = true;
            } else if ((cur & ~TBK_SHORT_HAPB) == f[i].word) done = true;
        }
        if (done) continue;
        line[31] |= TBK_SHORT_FLAG;
        line[TBK_SHORT_SUMMARY] |= TBK_SHORT_FLAG | tbk_short_filter_bit(f[i].word);
        for (uint32_t at = tbk_short_over_home(key, t.over_mask), walked = 0; walked <= t.over_mask && !done; walked++, at = (at + 1) & t.over_mask) {
            if (t.over[at] == TBK_SHORT_EMPTY64) { t.over[at] = key | ((uint64_t)(half ? 1 : 0) << 63); t.past++; done = true; }
            else if ((t.over[at] & ~(1ull << 63)) == key) done = true;
        }
        if (!done) { fprintf(stderr, "overflow table full\n"); exit(2); }
    }
}

// what the probe kernel computes for the window whose forward k-mer is `fwd`: from the forward strand alone
static TbkShortKey window_key(const Table &t, uint64_t fwd, int pick_last_tie) {
    const uint64_t rc = tbk_revcomp_packed(fwd, t.k);
    const int nt = tbk_mz_positions(t.z);
    uint32_t best = 0xFFFFFFFFu;
    int x = 0;
    for (int i = 0; i < nt; i++) {
        const uint32_t r = tbk_tmer_rank(fwd, t.z, i);
        if (r < best || (pick_last_tie && r == best)) { best = r; x = i; }
    }
    const int pos = x % t.z.w;
    const uint32_t mmask = t.z.m == 16 ? 0xFFFFFFFFu : ((1u << (2 * t.z.m)) - 1u);
    const uint32_t mx = (uint32_t)(fwd >> (2 * (t.z.o + pos))) & mmask;
    const uint32_t my = (uint32_t)(rc >> (2 * (t.z.o + t.z.w - 1 - pos))) & mmask;
    if (my != tbk_revcomp32(mx, t.z.m)) { fprintf(stderr, "strand geometry\n"); exit(4); }
    const bool f = mx < my;
    return tbk_short_key(f ? fwd : rc, t.z, t.g, f ? pos : t.z.w - 1 - pos, t.n_buckets);
}

int main(int argc, char **argv) {
    const int k = argc > 1 ? atoi(argv[1]) : 21;
    const int w_want = argc > 2 ? atoi(argv[2]) : 6;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
    const uint32_t n_buckets = argc > 4 ? (uint32_t)strtoul(argv[4], nullptr, 10) : 0;
    std::mt19937_64 rng(seed);
    Table t;
    t.k = k;
    t.z = tbk_mz_params(k, w_want, 1000000, 0, 1);
    if (argc > 5 && atoi(argv[5])) t.z = tbk_mz_span3(t.z);   // 3w t-mer positions (what the library uses where t stays at 4 or more)
    t.n_buckets = n_buckets ? n_buckets : tbk_short_min_buckets(k, t.z) + (uint32_t)(rng() % 1000);
    if (!tbk_short_geom(k, t.z, t.n_buckets, &t.g)) { printf("k=%d w=%d n_buckets=%u: no short keys (w=%d m=%d o=%d t=%d)\n", k, w_want, t.n_buckets, t.z.w, t.z.m, t.z.o, t.z.t); return 0; }
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    const int G = 60000;
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
    // crowded m-mers: hundreds of keys around one stretch of sequence (every one samples an m-mer of that stretch), and as
    // many near misses that are NOT in the lists
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
    for (uint64_t x : list_a) if (x == canon(x, k)) set_a.insert(x);
    for (uint64_t x : list_b) if (x == canon(x, k) && !set_a.count(x)) set_b.insert(x);
    t.lines.assign((size_t)t.n_buckets * 32, 0);
    t.over_mask = 4095;
    t.over.assign(4096, TBK_SHORT_EMPTY64);
    for (uint64_t x : list_a) insert_key(t, 0, x, false);
    for (uint64_t x : list_b) insert_key(t, 1, x, true);
    uint64_t bad = 0;
    for (uint64_t x : set_a) if (which_list(t, x) != 0) bad++;
    for (uint64_t x : set_b) if (which_list(t, x) != 1) bad++;
    for (uint64_t x : near_miss) { const int want = set_a.count(x) ? 0 : set_b.count(x) ? 1 : -1; if (which_list(t, x) != want) bad++; }
    for (int i = 0; i < 200000; i++) { const uint64_t x = canon(rng() & kmask, k); const int want = set_a.count(x) ? 0 : set_b.count(x) ? 1 : -1; if (which_list(t, x) != want) bad++; }
    uint64_t windows = 0, hits_a = 0, hits_b = 0;
    for (int strand = 0; strand < 2; strand++)
        for (const std::vector<uint8_t> *hap : {&ga, &gb}) {
            std::vector<uint8_t> r(*hap);
            for (int i = 0; i < G; i++) if (rng() % 300 == 0) r[i] = (uint8_t)(rng() & 3);
            if (strand) { std::vector<uint8_t> q(G); for (int i = 0; i < G; i++) q[i] = (uint8_t)(3 - r[G - 1 - i]); r = q; }
            for (int i = 0; i + k <= G; i++) {
                const uint64_t fwd = kmer_at(r, i), key = canon(fwd, k);
                // as the window loop decides: the front, then - only if the line's summary has the word's bit - what lies behind it
                const int which = tbk_short_lookup_one(t.lines.data(), t.n_buckets, t.over.data(), t.over_mask, window_key(t, fwd, (int)(rng() & 1)), key, 1);
                if ((which == 0) != (set_a.count(key) != 0) || (which == 1) != (set_b.count(key) != 0)) bad++;
                windows++; hits_a += which == 0; hits_b += which == 1;
            }
        }
    printf("short k=%d w=%d m=%d o=%d t=%d fbits=%d rshift=%d: %llu keys in %llu words, %u buckets, %llu behind a front, %llu in the overflow table; %llu windows, %llu / %llu hits; mismatches %llu\n",
           k, t.z.w, t.z.m, t.z.o, t.z.t, t.g.fbits, t.g.rshift, (unsigned long long)(set_a.size() + set_b.size()), (unsigned long long)t.words, t.n_buckets,
           (unsigned long long)t.behind, (unsigned long long)t.past, (unsigned long long)windows, (unsigned long long)hits_a, (unsigned long long)hits_b, (unsigned long long)bad);
    return bad ? 1 : 0;
}
