// CPU model of the entry layout (tbk_common.h "entry layout"): a sequential build with the insert rule of
// tbk_entry_insert_kernel, then every claim the probe kernel relies on is checked against plain set membership:
//   * every canonical list key is found, through each of its (tied position, orientation) forms, in its own list only;
//   * keys outside the lists are not found;
//   * a read window asks with (orientation by the sampled m-mer, position) computed from the FORWARD strand alone, as
//     the kernel does, on either strand of the same sequence, and gets the set's answer;
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
    std::vector<uint64_t> slots;
    uint32_t n_buckets;
    TbkMz z;
    TbkEntryGeom g;
    int k;
    uint64_t entries = 0, merged = 0, behind = 0, past = 0;
    bool wide = false;
};

// the insert rule (sequential form): first compatible entry of the list along the m-mer's bucket sequence, else the first empty slot
static void insert_form(Table &t, uint32_t half, TbkEntryKey e) {
    const uint32_t hapb = half ? 1 : 0;
    uint32_t b = tbk_entry_bucket(e.cm, t.n_buckets);
    for (uint32_t walked = 0; walked <= t.n_buckets; walked++) {
        uint64_t *line = t.slots.data() + (uint64_t)b * 16;
        for (uint32_t s = 0; s < 16; s++) {
            uint64_t &v = line[s];
            if ((v & ~((uint64_t)TBK_ENTRY_FLAG << 32)) == 0) {
                v |= (uint64_t)e.cm | ((uint64_t)(e.khi | (hapb ? TBK_ENTRY_HAPB : 0u)) << 32);
                t.entries++;
                if (s >= 4) { line[3] |= (uint64_t)TBK_ENTRY_FLAG << 32; t.behind++; }
                return;
            }
            if (tbk_entry_compatible(v, e, hapb, t.z, t.g)) { v |= (uint64_t)e.khi << 32; t.merged++; return; }
        }
        line[15] |= (uint64_t)TBK_ENTRY_FLAG << 32;
        t.past++;
        b = tbk_entry_next_bucket(e.cm, t.n_buckets, b, walked == 0);
    }
    fprintf(stderr, "table full\n");
    exit(2);
}

// ---- the same for wide entries (k up to 32): one entry per 16-byte piece, eight pieces per line shared by both lists ----
static void winsert_form(Table &t, uint32_t half, TbkWideKey e) {
    const uint32_t hapb = half ? 1 : 0;
    uint32_t b = tbk_wentry_bucket(e.cm, t.n_buckets);
    for (uint32_t walked = 0; walked <= t.n_buckets; walked++) {
        uint64_t *line = t.slots.data() + (uint64_t)b * 16;
        for (uint32_t i = 0; i < 8; i++) {
            uint64_t &w0 = line[2 * i], &w1 = line[2 * i + 1];
            if (!(w0 & TBK_WENTRY_TAKEN)) {
                w0 = e.cm | TBK_WENTRY_TAKEN;
                w1 |= e.k1 | (hapb ? TBK_WENTRY_HAPB : 0ull);
                t.entries++;
                if (i >= 2) { line[2 * 1 + 1] |= TBK_WENTRY_FLAG; t.behind++; }
                return;
            }
            if (tbk_wentry_compatible(w0, w1, e, hapb, t.z, t.g)) { w1 |= e.k1; t.merged++; return; }
        }
        line[2 * 7 + 1] |= TBK_WENTRY_FLAG;
        t.past++;
        b = tbk_wentry_next_bucket(e.cm, t.n_buckets, b, walked == 0);
    }
    fprintf(stderr, "table full\n");
    exit(2);
}

static int wforms_of(const Table &t, uint64_t key, TbkWideKey *out) {
    const int nt = tbk_mz_positions(t.z);
    uint32_t best = 0xFFFFFFFFu;
    for (int i = 0; i < nt; i++) { const uint32_t r = tbk_tmer_rank(key, t.z, i); best = r < best ? r : best; }
    int n = 0;
    for (int i = 0; i < nt; i++) {
        if (tbk_tmer_rank(key, t.z, i) != best) continue;
        n += tbk_wentry_orientations(key, t.k, t.z, t.g, i % t.z.w, out + n);
    }
    return n;
}

static bool wcontains(const Table &t, uint32_t half, uint64_t key) {
    TbkWideKey f[64];
    const int n = wforms_of(t, key, f);
    bool any = false, all = true;
    for (int i = 0; i < n; i++) { const bool h = tbk_wentry_lookup_one(t.slots.data(), t.n_buckets, f[i]) == (half ? 1 : 0); any = any || h; all = all && h; }
    if (any != all) { fprintf(stderr, "forms of one key disagree\n"); exit(3); }
    return any;
}

static TbkWideKey wwindow_key(const Table &t, uint64_t fwd, int pick_last_tie) {
    const uint64_t rc = tbk_revcomp_packed(fwd, t.k);
    const int nt = tbk_mz_positions(t.z);
    uint32_t best = 0xFFFFFFFFu;
    int x = 0;
    for (int i = 0; i < nt; i++) {
        const uint32_t r = tbk_tmer_rank(fwd, t.z, i);
        if (r < best || (pick_last_tie && r == best)) { best = r; x = i; }
    }
    const int pos = x % t.z.w;
    const uint64_t mmask = (1ull << (2 * t.z.m)) - 1ull;
    const uint64_t mx = (fwd >> (2 * (t.z.o + pos))) & mmask;
    const uint64_t my = (rc >> (2 * (t.z.o + t.z.w - 1 - pos))) & mmask;
    if (my != tbk_revcomp64(mx, t.z.m)) { fprintf(stderr, "strand geometry\n"); exit(4); }
    const bool f = mx < my;
    return tbk_wentry_key(f ? fwd : rc, t.z, t.g, f ? pos : t.z.w - 1 - pos);
}

// every (tied position, orientation) form of a list key
static int forms_of(const Table &t, uint64_t key, TbkEntryKey *out) {
    const int nt = tbk_mz_positions(t.z);
    uint32_t best = 0xFFFFFFFFu;
    for (int i = 0; i < nt; i++) { const uint32_t r = tbk_tmer_rank(key, t.z, i); best = r < best ? r : best; }
    int n = 0;
    for (int i = 0; i < nt; i++) {
        if (tbk_tmer_rank(key, t.z, i) != best) continue;
        n += tbk_entry_orientations(key, t.k, t.z, t.g, i % t.z.w, out + n);
    }
    return n;
}

static bool wcontains(const Table &t, uint32_t half, uint64_t key);
static bool contains(const Table &t, uint32_t half, uint64_t key) {
    if (t.wide) return wcontains(t, half, key);
    TbkEntryKey f[64];
    const int n = forms_of(t, key, f);
    bool any = false, all = true;
    for (int i = 0; i < n; i++) { const bool h = tbk_entry_lookup_one(t.slots.data(), t.n_buckets, f[i]) == (half ? 1 : 0); any = any || h; all = all && h; }
    if (any != all) { fprintf(stderr, "forms of one key disagree\n"); exit(3); }
    return any;
}

static void insert_key(Table &t, uint32_t half, uint64_t key, const Table *skip_in_a) {
    if (key != canon(key, t.k)) return;  // a non-canonical list line is dead in the reference (c/kmers.c:113 vs 251-255)
    if (skip_in_a && contains(*skip_in_a, 0, key)) return;
    if (t.wide) {
        TbkWideKey wf[64];
        const int wn = wforms_of(t, key, wf);
        for (int i = 0; i < wn; i++) winsert_form(t, half, wf[i]);
        return;
    }
    TbkEntryKey f[64];
    const int n = forms_of(t, key, f);
    for (int i = 0; i < n; i++) insert_form(t, half, f[i]);
}

// what the probe kernel computes for the window whose forward k-mer is `fwd`: from the forward strand alone
static TbkEntryKey window_key(const Table &t, uint64_t fwd, int pick_last_tie) {
    const uint64_t rc = tbk_revcomp_packed(fwd, t.k);
    const int nt = tbk_mz_positions(t.z);
    uint32_t best = 0xFFFFFFFFu;
    int x = 0;
    for (int i = 0; i < nt; i++) {
        const uint32_t r = tbk_tmer_rank(fwd, t.z, i);  // (rank of the canonical t-mer: the same from either strand)
        if (r < best || (pick_last_tie && r == best)) { best = r; x = i; }
    }
    const int pos = x % t.z.w;
    const uint32_t mmask = t.z.m == 16 ? 0xFFFFFFFFu : ((1u << (2 * t.z.m)) - 1u);
    const uint32_t mx = (uint32_t)(fwd >> (2 * (t.z.o + pos))) & mmask;
    const uint32_t my = (uint32_t)(rc >> (2 * (t.z.o + t.z.w - 1 - pos))) & mmask;
    if (my != tbk_revcomp32(mx, t.z.m)) { fprintf(stderr, "strand geometry\n"); exit(4); }
    const bool f = mx < my;
    return tbk_entry_key(f ? fwd : rc, t.z, t.g, f ? pos : t.z.w - 1 - pos);
}

int main(int argc, char **argv) {
    const int k = argc > 1 ? atoi(argv[1]) : 21;
    const int w_want = argc > 2 ? atoi(argv[2]) : 6;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
    const int crowd = argc > 4 ? atoi(argv[4]) : 0;  // 1: a table so small that lines overflow
    const int wide = argc > 5 ? atoi(argv[5]) : 0;   // m: wide entries (16 bytes) with m-mers of that length (1: of 18, the default)
    std::mt19937_64 rng(seed);
    Table t;
    t.k = k;
    t.wide = wide != 0;
    t.z = tbk_mz_params(k, w_want, 1000000, wide ? (wide == 1 ? 18 : wide) : 0, 1);
    if (!wide && argc > 6 && atoi(argv[6])) t.z = tbk_mz_span3(t.z);   // 3w t-mer positions (what the library uses for narrow entries where t stays at 4 or more)
    if (!(wide ? tbk_wentry_geom(k, t.z, &t.g) : tbk_entry_geom(k, t.z, &t.g))) { printf("k=%d w=%d: no entry layout (w=%d m=%d o=%d t=%d)\n", k, w_want, t.z.w, t.z.m, t.z.o, t.z.t); return 0; }
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    // a genome with SNPs between two haplotypes, low-complexity stretches and a repeated segment
    const int G = 60000;
    std::vector<uint8_t> ga(G), gb(G);
    for (int i = 0; i < G; i++) ga[i] = (uint8_t)(rng() & 3);
    for (int i = 20000; i < 20400; i++) ga[i] = (uint8_t)((i / 3) & 1);         // low complexity
    for (int i = 0; i < 3000; i++) ga[30000 + i] = ga[5000 + i];               // a repeat
    for (int i = 0; i < 64; i++) ga[40000 + i] = (uint8_t)(i < 32 ? (i & 3) : 3 - ((63 - i) & 3));  // a palindromic stretch (its own reverse complement)
    gb = ga;
    for (int i = 0; i < G; i++) if (rng() % 150 == 0) gb[i] = (uint8_t)((ga[i] + 1 + rng() % 3) & 3);
    auto kmer_at = [&](const std::vector<uint8_t> &g, int i) { uint64_t x = 0; for (int j = 0; j < k; j++) x |= (uint64_t)g[i + j] << (2 * j); return x; };
    std::unordered_set<uint64_t> all_a, all_b;
    for (int i = 0; i + k <= G; i++) { all_a.insert(canon(kmer_at(ga, i), k)); all_b.insert(canon(kmer_at(gb, i), k)); }
    std::vector<uint64_t> list_a, list_b;
    for (uint64_t x : all_a) if (!all_b.count(x)) list_a.push_back(x);
    for (uint64_t x : all_b) if (!all_a.count(x)) list_b.push_back(x);
    // uniform keys, keys in both lists, duplicates, non-canonical lines
    for (int i = 0; i < 20000; i++) list_a.push_back(canon(rng() & kmask, k));
    for (int i = 0; i < 20000; i++) list_b.push_back(canon(rng() & kmask, k));
    for (int i = 0; i < 500; i++) { list_b.push_back(list_a[rng() % list_a.size()]); list_a.push_back(list_a[rng() % list_a.size()]); }
    for (int i = 0; i < 500; i++) { const uint64_t x = rng() & kmask; if (x != canon(x, k)) { list_a.push_back(x); list_b.push_back(x); } }
    std::unordered_set<uint64_t> set_a, set_b;
    for (uint64_t x : list_a) if (x == canon(x, k)) set_a.insert(x);
    for (uint64_t x : list_b) if (x == canon(x, k) && !set_a.count(x)) set_b.insert(x);
    const uint64_t n_keys = set_a.size() + set_b.size();
    t.n_buckets = crowd ? (uint32_t)(n_keys / (wide ? 6 : 12) + 7) : (uint32_t)(n_keys / (wide ? 2 : 4) + 16);
    t.slots.assign((size_t)t.n_buckets * 16, 0);
    for (uint64_t x : list_a) insert_key(t, 0, x, nullptr);
    for (uint64_t x : list_b) insert_key(t, 8, x, &t);
    uint64_t bad = 0;
    for (uint64_t x : set_a) { if (!contains(t, 0, x)) bad++; if (contains(t, 8, x)) bad++; }
    for (uint64_t x : set_b) { if (!contains(t, 8, x)) bad++; if (contains(t, 0, x)) bad++; }
    for (int i = 0; i < 200000; i++) { const uint64_t x = canon(rng() & kmask, k); if (!set_a.count(x) && contains(t, 0, x)) bad++; if (!set_b.count(x) && contains(t, 8, x)) bad++; }
    // reads: windows of both haplotypes and of their reverse complements, with errors
    uint64_t windows = 0, hits_a = 0, hits_b = 0;
    for (int strand = 0; strand < 2; strand++)
        for (const std::vector<uint8_t> *hap : {&ga, &gb}) {
            std::vector<uint8_t> r(*hap);
            for (int i = 0; i < G; i++) if (rng() % 300 == 0) r[i] = (uint8_t)(rng() & 3);
            if (strand) { std::vector<uint8_t> q(G); for (int i = 0; i < G; i++) q[i] = (uint8_t)(3 - r[G - 1 - i]); r = q; }
            for (int i = 0; i + k <= G; i++) {
                const uint64_t fwd = kmer_at(r, i), key = canon(fwd, k);
                const int tie = (int)(rng() & 1);
                bool in_a, in_b;
                if (t.wide) {
                    const TbkWideKey e = wwindow_key(t, fwd, tie);
                    const int which = tbk_wentry_lookup_one(t.slots.data(), t.n_buckets, e);
                    in_a = which == 0; in_b = which == 1;
                } else {
                    const TbkEntryKey e = window_key(t, fwd, tie);
                    const int which = tbk_entry_lookup_one(t.slots.data(), t.n_buckets, e);
                    in_a = which == 0; in_b = which == 1;
                }
                if (in_a != (set_a.count(key) != 0) || in_b != (set_b.count(key) != 0)) bad++;
                windows++; hits_a += in_a; hits_b += in_b;
            }
        }
    printf("%sk=%d w=%d m=%d o=%d t=%d fl=%d crowd=%d: %llu keys in %llu entries (%llu merges), %u buckets, %llu behind a front, %llu past a line; %llu windows, %llu / %llu hits; mismatches %llu\n",
           wide ? "wide " : "", k, t.z.w, t.z.m, t.z.o, t.z.t, t.g.fl, crowd, (unsigned long long)n_keys, (unsigned long long)t.entries, (unsigned long long)t.merged, t.n_buckets,
           (unsigned long long)t.behind, (unsigned long long)t.past, (unsigned long long)windows, (unsigned long long)hits_a, (unsigned long long)hits_b, (unsigned long long)bad);
    return bad ? 1 : 0;
}
