/*
 * kmers_oracle.c — CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file restates, in plain C, the algorithm of the reference's single native
 * translation unit (/root/reference/c/kmers.c) for the classify-by-kmers path.  It is
 * the parity checker and the CPU baseline; it is NOT part of the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (trio_binning_amd + libtbk_hip.so) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 *   (a) the known-answer tests the reference's own suite holds (tests/test_kmers.py:11,
 *       :21-25, :31-35, :45-49),
 *   (b) golden vectors produced by the real reference (compiled c/kmers.c driven through
 *       the reference's own Python) by tests/golden/make_golden.py, and
 *   (c) when oracle/_ref/kmers_ref.so exists (built from the reference's own source by
 *       oracle/Makefile), the real reference itself on seeded random inputs.
 *
 * Faithful cost profile (this matters for the CPU baseline): two linear-probing tables
 * with hash_size = n*4/3 (load 0.75), a parallel "full" byte array, the reference's
 * 64->32-bit mixer, NON-rolling re-encoding of the window and of its reverse complement
 * for every probe, hapA probed first and hapB only when hapA missed.
 *
 * One documented deviation: a window that contains a byte outside {A,C,G,T}.  The
 * reference reads a stale / uninitialised scratch buffer there (c/kmers.c:78-91,279), so
 * its result is undefined; its own docstring restricts reads to [ACGT]
 * (src/trio_binning/kmers.py:134-135).  The oracle (and the product) define such a window
 * as "no hit".  Pass strict_acgt = 0 to orc_count_kmers_in_read to get the reference's
 * defined half of that behaviour instead (forward strand encodes the byte as 0; the
 * reverse-complement scratch keeps whatever an earlier window left there, starting from
 * zero-filled memory) — used only to cross-check against oracle/_ref on ACGT input.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>

typedef struct {
    uint64_t *slot;      /* packed k-mer per slot            (ref: hash_set.kmers, c/kmers.c:16) */
    unsigned char *used; /* 1 when slot holds a k-mer         (ref: hash_set.full,  c/kmers.c:21) */
    uint64_t n_slots;    /* ref: hash_size = num_kmers*4/3    (c/kmers.c:167)                     */
    uint64_t n_lines;    /* ref: num_kmers, duplicates counted (c/kmers.c:166)                     */
    int k;
} orc_table;

/* ---- 2-bit packing: base i -> bits 2i..2i+1, A=0 C=1 G=2 T=3, anything else 0 ----------
 * (ref: kmer_to_int, c/kmers.c:50-72).  Table-driven instead of a switch. */
static unsigned char g_code[256];
static unsigned char g_comp[256]; /* complement letter, 0 when the byte is not ACGT */
static int g_tables_ready = 0;

static void orc_init_tables(void) {
    if (g_tables_ready) return;
    memset(g_code, 0, sizeof g_code);
    memset(g_comp, 0, sizeof g_comp);
    g_code['C'] = 1; g_code['G'] = 2; g_code['T'] = 3;
    g_comp['A'] = 'T'; g_comp['C'] = 'G'; g_comp['G'] = 'C'; g_comp['T'] = 'A';
    g_tables_ready = 1;
}

uint64_t orc_kmer_to_int(const char *kmer, int k) {
    orc_init_tables();
    uint64_t v = 0;
    for (int i = 0; i < k; i++)
        v |= (uint64_t)g_code[(unsigned char)kmer[i]] << (2 * i);
    return v;
}

/* (ref: reverse_complement, c/kmers.c:74-93) out[k-1-i] = complement(in[i]); a byte that
 * is not ACGT leaves out[k-1-i] untouched, exactly as the reference's switch does. */
void orc_reverse_complement(const char *in, char *out, int k) {
    orc_init_tables();
    for (int i = 0; i < k; i++) {
        unsigned char c = g_comp[(unsigned char)in[i]];
        if (c) out[k - 1 - i] = (char)c;
    }
}

/* (ref: hash_function, c/kmers.c:98-103) */
uint32_t orc_hash(uint64_t x) {
    x = ((x >> 16) ^ x) * 0x45d9f3bULL;
    x = ((x >> 16) ^ x) * 0x45d9f3bULL;
    x = (x >> 16) ^ x;
    return (uint32_t)x;
}

/* (ref: initialize_hash_set, c/kmers.c:160-180) */
static orc_table *orc_table_alloc(int k, uint64_t n_lines) {
    orc_table *t = (orc_table *)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->k = k;
    t->n_lines = n_lines;
    t->n_slots = n_lines * 4 / 3;
    t->slot = (uint64_t *)malloc((t->n_slots ? t->n_slots : 1) * sizeof(uint64_t));
    t->used = (unsigned char *)calloc(t->n_slots ? t->n_slots : 1, 1);
    if (!t->slot || !t->used) { free(t->slot); free(t->used); free(t); return NULL; }
    return t;
}

/* (ref: add_to_hash, c/kmers.c:112-122) verbatim key, no canonicalisation, no dedupe */
static void orc_insert_key(orc_table *t, uint64_t key) {
    uint64_t pos = orc_hash(key) % t->n_slots;
    while (t->used[pos]) pos = (pos + 1) % t->n_slots;
    t->used[pos] = 1;
    t->slot[pos] = key;
}

/* (ref: peek_at_file c/kmers.c:124-146 + create_kmer_hash_set c/kmers.c:185-229)
 * k = bytes getline() returned for the first line, minus one; every getline() success is
 * one k-mer; each line contributes its first k bytes.  Returns NULL when the file cannot
 * be opened or holds no line (the reference would crash / divide by zero there). */
orc_table *orc_table_from_file(const char *path) {
    orc_init_tables();
    FILE *fp = fopen(path, "r");
    if (!fp) return NULL;
    char *line = NULL;
    size_t cap = 0;
    ssize_t got = getline(&line, &cap, fp);
    if (got <= 0) { free(line); fclose(fp); return NULL; }
    int k = (int)got - 1;
    uint64_t n_lines = 1;
    while (getline(&line, &cap, fp) != -1) n_lines++;
    if (k < 1 || k > 32 || n_lines * 4 / 3 < 1) { free(line); fclose(fp); return NULL; }
    orc_table *t = orc_table_alloc(k, n_lines);
    rewind(fp);
    if (t) {
        while ((got = getline(&line, &cap, fp)) != -1) {
            /* a line shorter than k packs getline's buffer as the reference does: the line, the NUL,
             * then the bytes earlier lines of this pass left behind (same buffer, only ever grown) */
            orc_insert_key(t, orc_kmer_to_int(line, k));
        }
    }
    free(line);
    fclose(fp);
    return t;
}

/* Same table, built from already-packed keys (bench.py's cpu_baseline leg: the keys come
 * back from the GPU generator).  Sequential, identical slot placement to the file path. */
orc_table *orc_table_from_keys(const uint64_t *keys, uint64_t n, int k) {
    orc_init_tables();
    if (n * 4 / 3 < 1) return NULL;
    orc_table *t = orc_table_alloc(k, n);
    if (!t) return NULL;
    for (uint64_t i = 0; i < n; i++) orc_insert_key(t, keys[i]);
    return t;
}

/* Multi-threaded variant for the 2x300M-entry bench tables: same layout rule (n*4/3
 * slots, linear probing, first free slot wins) but slots are claimed with a CAS on the
 * "used" byte, so placement within a probe run may differ from the sequential build.
 * Membership, load factor and probe-length distribution are the same. */
typedef struct { orc_table *t; const uint64_t *keys; uint64_t lo, hi; } orc_build_job;

static void *orc_build_worker(void *arg) {
    orc_build_job *j = (orc_build_job *)arg;
    orc_table *t = j->t;
    for (uint64_t i = j->lo; i < j->hi; i++) {
        uint64_t key = j->keys[i];
        uint64_t pos = orc_hash(key) % t->n_slots;
        for (;;) {
            if (!t->used[pos] && __sync_bool_compare_and_swap(&t->used[pos], 0, 1)) {
                t->slot[pos] = key;
                break;
            }
            pos = (pos + 1) % t->n_slots;
        }
    }
    return NULL;
}

orc_table *orc_table_from_keys_mt(const uint64_t *keys, uint64_t n, int k, int threads) {
    orc_init_tables();
    if (n * 4 / 3 < 1) return NULL;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    orc_table *t = orc_table_alloc(k, n);
    if (!t) return NULL;
    pthread_t tid[256];
    orc_build_job job[256];
    for (int w = 0; w < threads; w++) {
        job[w].t = t; job[w].keys = keys;
        job[w].lo = n * (uint64_t)w / threads;
        job[w].hi = n * (uint64_t)(w + 1) / threads;
        pthread_create(&tid[w], NULL, orc_build_worker, &job[w]);
    }
    for (int w = 0; w < threads; w++) pthread_join(tid[w], NULL);
    __sync_synchronize();
    return t;
}

void orc_table_free(orc_table *t) {
    if (!t) return;
    free(t->slot); free(t->used); free(t);
}

uint64_t orc_table_num_kmers(const orc_table *t) { return t->n_lines; }
/* the stored keys in slot order (tests compare them, sorted, with the real reference's arrays) */
uint64_t orc_table_keys(const orc_table *t, uint64_t *out) {
    uint64_t n = 0;
    for (uint64_t i = 0; i < t->n_slots; i++) if (t->used[i]) out[n++] = t->slot[i];
    return n;
}
uint64_t orc_table_hash_size(const orc_table *t) { return t->n_slots; }
int orc_table_k(const orc_table *t) { return t->k; }

/* (ref: kmer_in_hash_set, c/kmers.c:245-268) encode window, write its reverse complement
 * into the caller's scratch, encode that, look up the smaller of the two integers by
 * linear probing until an unused slot. */
static int orc_member(const char *win, char *scratch, const orc_table *t) {
    uint64_t f = orc_kmer_to_int(win, t->k);
    orc_reverse_complement(win, scratch, t->k);
    uint64_t r = orc_kmer_to_int(scratch, t->k);
    uint64_t key = f < r ? f : r;
    uint64_t pos = orc_hash(key) % t->n_slots;
    /* `walked` bounds the probe: with <= 2 list lines the reference's table is 100% full
     * and its loop never ends on a miss (c/kmers.c:167,258-265); the oracle answers 0. */
    for (uint64_t walked = 0; t->used[pos] && walked < t->n_slots; walked++) {
        if (t->slot[pos] == key) return 1;
        pos = (pos + 1) % t->n_slots;
    }
    return 0;
}

/* (ref: count_kmers_in_read, c/kmers.c:270-299) windows 0..len-k; hapA first, hapB only
 * on a hapA miss; k is hapA's k for both tables.  len < 0 means "NUL-terminated".
 * strict_acgt != 0: windows holding a non-ACGT byte score nothing (documented policy). */
void orc_count_kmers_in_read(const char *read, int64_t len, const orc_table *a,
                             const orc_table *b, int strict_acgt, int *count_a,
                             int *count_b) {
    orc_init_tables();
    int k = a->k;
    char *win = (char *)calloc((size_t)k + 1, 1);
    char *scratch = (char *)calloc((size_t)k + 1, 1);
    int ca = 0, cb = 0;
    if (len < 0) len = (int64_t)strlen(read);
    int64_t bad_until = -1; /* windows starting at <= bad_until contain a non-ACGT byte */
    if (strict_acgt)
        for (int64_t j = 0; j < k - 1 && j < len; j++)
            if (!g_comp[(unsigned char)read[j]]) bad_until = j;
    for (int64_t i = 0; i + k <= len; i++) {
        if (strict_acgt) {
            if (!g_comp[(unsigned char)read[i + k - 1]]) bad_until = i + k - 1;
            if (bad_until >= i) continue;
        }
        memcpy(win, read + i, (size_t)k);
        if (orc_member(win, scratch, a)) ca++;
        else if (orc_member(win, scratch, b)) cb++;
    }
    free(win);
    free(scratch);
    *count_a = ca;
    *count_b = cb;
}

/* Batch form used by tests and by bench.py's cpu_baseline: reads are
 * bases[offsets[i] .. offsets[i+1]), counts is int32[n_reads][2].  threads > 1 shards the
 * reads over POSIX threads that share the read-only tables (the most favourable
 * multi-core reading of the single-threaded reference). */
typedef struct {
    const uint8_t *bases; const uint64_t *offsets; uint64_t lo, hi;
    const orc_table *a, *b; int strict; int32_t *counts;
} orc_count_job;

static void *orc_count_worker(void *arg) {
    orc_count_job *j = (orc_count_job *)arg;
    for (uint64_t r = j->lo; r < j->hi; r++) {
        int ca, cb;
        orc_count_kmers_in_read((const char *)j->bases + j->offsets[r],
                                (int64_t)(j->offsets[r + 1] - j->offsets[r]), j->a, j->b,
                                j->strict, &ca, &cb);
        j->counts[2 * r] = ca;
        j->counts[2 * r + 1] = cb;
    }
    return NULL;
}

void orc_count_batch(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                     const orc_table *a, const orc_table *b, int strict_acgt, int threads,
                     int32_t *counts) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if ((uint64_t)threads > n_reads) threads = n_reads ? (int)n_reads : 1;
    pthread_t tid[256];
    orc_count_job job[256];
    for (int w = 0; w < threads; w++) {
        job[w] = (orc_count_job){bases, offsets, n_reads * (uint64_t)w / threads,
                                 n_reads * (uint64_t)(w + 1) / threads, a, b, strict_acgt,
                                 counts};
        if (threads == 1) orc_count_worker(&job[w]);
        else pthread_create(&tid[w], NULL, orc_count_worker, &job[w]);
    }
    if (threads > 1)
        for (int w = 0; w < threads; w++) pthread_join(tid[w], NULL);
}

/* ---- fairness datum (SURVEY 8d "CPU-opt") --------------------------------------------------
 * NOT a restatement of the reference: the same two tables and the same A-then-B rule, but the
 * forward and reverse-complement k-mers are rolled base by base instead of re-encoded per
 * window, and reads are sharded over threads.  bench.py reports its rate next to the faithful
 * port so that the GPU is not only compared with the reference's slowest formulation.  Counts
 * are identical to orc_count_batch with strict_acgt = 1 (tests/test_oracle_golden.py). */
static int orc_lookup_key(const orc_table *t, uint64_t key) {
    uint64_t pos = orc_hash(key) % t->n_slots;
    for (uint64_t walked = 0; t->used[pos] && walked < t->n_slots; walked++) {
        if (t->slot[pos] == key) return 1;
        pos = (pos + 1) % t->n_slots;
    }
    return 0;
}

static void *orc_fast_worker(void *arg) {
    orc_count_job *j = (orc_count_job *)arg;
    const int k = j->a->k;
    const uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    for (uint64_t r = j->lo; r < j->hi; r++) {
        const uint8_t *s = j->bases + j->offsets[r];
        const uint64_t len = j->offsets[r + 1] - j->offsets[r];
        uint64_t fwd = 0, rc = 0;
        int run = 0, ca = 0, cb = 0;  /* run = consecutive ACGT bases ending here */
        for (uint64_t i = 0; i < len; i++) {
            const unsigned char c = s[i];
            if (!g_comp[c]) { run = 0; fwd = rc = 0; continue; }
            const uint64_t code = g_code[c];
            fwd = (fwd >> 2) | (code << (2 * (k - 1)));
            rc = ((rc << 2) | (3 - code)) & mask;
            if (++run >= k) {
                const uint64_t key = fwd < rc ? fwd : rc;
                if (orc_lookup_key(j->a, key)) ca++;
                else if (orc_lookup_key(j->b, key)) cb++;
            }
        }
        j->counts[2 * r] = ca;
        j->counts[2 * r + 1] = cb;
    }
    return NULL;
}

void orc_count_batch_fast(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                          const orc_table *a, const orc_table *b, int threads, int32_t *counts) {
    orc_init_tables();
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if ((uint64_t)threads > n_reads) threads = n_reads ? (int)n_reads : 1;
    pthread_t tid[256];
    orc_count_job job[256];
    for (int w = 0; w < threads; w++) {
        job[w] = (orc_count_job){bases, offsets, n_reads * (uint64_t)w / threads,
                                 n_reads * (uint64_t)(w + 1) / threads, a, b, 1, counts};
        if (threads == 1) orc_fast_worker(&job[w]);
        else pthread_create(&tid[w], NULL, orc_fast_worker, &job[w]);
    }
    if (threads > 1)
        for (int w = 0; w < threads; w++) pthread_join(tid[w], NULL);
}

/* (ref: calculate_scaling_factors classify_by_kmers.py:57-77 and the binning rule
 * classify_by_kmers.py:104-115) float64, same operation order: 1.0*max/n, count*factor,
 * strict '>' both ways, otherwise 'U'. */
void orc_score_and_bin(const int32_t *counts, uint64_t n_reads, uint64_t num_a,
                       uint64_t num_b, double *score_a, double *score_b, char *bins) {
    uint64_t mx = num_a > num_b ? num_a : num_b;
    double fa = 1.0 * (double)mx / (double)num_a;
    double fb = 1.0 * (double)mx / (double)num_b;
    for (uint64_t r = 0; r < n_reads; r++) {
        double sa = counts[2 * r] * fa, sb = counts[2 * r + 1] * fb;
        score_a[r] = sa;
        score_b[r] = sb;
        bins[r] = sa > sb ? 'A' : (sb > sa ? 'B' : 'U');
    }
}

/* ---- find-unique-kmers step: a single-thread canonical k-mer counter -------------------------
 * Checker and CPU datum for the GPU counter (tbk_counter_*).  Restates the reading of `kmc -k<k>`
 * documented in oracle/unique_oracle.py (KMC itself is not available: parity with it is
 * unpinned): canonical k-mers over both strands, windows holding a byte outside ACGT skipped,
 * lower case counted as upper case.  hist[c], c = 1..255 = distinct k-mers whose count, capped at
 * 255, is c; hist[0] = distinct k-mers.  Open addressing, linear probing, table of `slots` entries
 * (a power of two >= 2x the distinct k-mers expected).  Returns 0, or -1 when the table fills up. */
int orc_kmer_histogram(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k,
                       uint64_t slots, uint64_t *hist) {
    uint64_t *keys = malloc(slots * sizeof(uint64_t));
    uint32_t *cnt = calloc(slots, sizeof(uint32_t));
    if (!keys || !cnt || (slots & (slots - 1))) { free(keys); free(cnt); return -1; }
    memset(keys, 0xFF, slots * sizeof(uint64_t));
    const uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    uint64_t used = 0;
    int rc = 0;
    for (uint64_t r = 0; r < n_reads && !rc; r++) {
        uint64_t fwd = 0, rev = 0;
        int run = 0;  /* consecutive ACGT bytes seen */
        for (uint64_t i = offsets[r]; i < offsets[r + 1]; i++) {
            int c;
            switch (bases[i] & 0xDF) { case 'A': c = 0; break; case 'C': c = 1; break; case 'G': c = 2; break; case 'T': c = 3; break; default: c = -1; }
            if (c < 0) { run = 0; fwd = rev = 0; continue; }
            fwd = (fwd >> 2) | ((uint64_t)c << (2 * (k - 1)));
            rev = ((rev << 2) | (uint64_t)(3 - c)) & mask;
            if (++run < k) continue;
            const uint64_t key = fwd < rev ? fwd : rev;
            uint64_t h = key * 0x9E3779B97F4A7C15ULL;
            h ^= h >> 29;
            uint64_t p = h & (slots - 1);
            while (keys[p] != key && keys[p] != ~0ULL) p = (p + 1) & (slots - 1);
            if (keys[p] != key) {
                if (++used > slots - slots / 8) { rc = -1; break; }
                keys[p] = key;
            }
            cnt[p]++;
        }
    }
    if (!rc) {
        memset(hist, 0, 256 * sizeof(uint64_t));
        for (uint64_t p = 0; p < slots; p++)
            if (keys[p] != ~0ULL) { hist[cnt[p] < 255 ? cnt[p] : 255]++; hist[0]++; }
    }
    free(keys);
    free(cnt);
    return rc;
}
