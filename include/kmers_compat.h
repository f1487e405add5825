/*
 * kmers_compat.h — the reference's OWN C symbols, exported by libtbk_hip.so.
 *
 * src/trio_binning/kmers.py:62-86 binds four functions of c/kmers.c by name and reads one
 * struct field through the returned pointer (kmers.py:159: hash_set.contents.num_kmers).
 * With these exports the reference's unmodified binding loads this library in place of its
 * kmers_c<EXT_SUFFIX> ("C. Literal drop-in"): same names, same signatures,
 * and a struct whose leading fields have the layout of c/kmers.c:12-38.
 *
 * They are thin: each forwards to the tbk_* entry point named beside it (include/tbk.h).
 * The reference's C has no error convention (a missing file crashes it, c/kmers.c:131-133);
 * here a failure returns NULL / leaves both counts at -1 and the reason is in
 * tbk_last_error().  There is no CPU fallback: without an MI355X both calls fail that way.
 */
#ifndef KMERS_COMPAT_H
#define KMERS_COMPAT_H

#include <stdint.h>

#include "tbk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* c/kmers.c:12-38.  The reference's arrays live in host memory; here the keys live in HBM, so
 * `kmers` and `full` are NULL.  hash_size is what initialize_hash_set computes (num_kmers * 4 / 3,
 * c/kmers.c:167).  The int fields saturate at INT_MAX (the reference's overflow there is
 * undefined); the exact line count is num_kmers_u64. */
typedef struct hash_set {
    uint64_t *kmers;        /* c/kmers.c:16  (NULL) */
    unsigned char *full;    /* c/kmers.c:21  (NULL) */
    int hash_size;          /* c/kmers.c:27 */
    unsigned char k;        /* c/kmers.c:32 */
    int num_kmers;          /* c/kmers.c:37 */
    /* ---- past the reference's layout ---- */
    uint64_t num_kmers_u64;
    tbk_table *table;       /* the list in HBM (tbk_table_create_from_file) */
} hash_set;

/* c/kmers.c:185-229 -> tbk_table_create_from_file on device TBK_DEVICE (default 0). */
hash_set *create_kmer_hash_set(char *kmer_file_path);
/* c/kmers.c:270-299 -> tbk_count_kmers_in_read; `read` is NUL-terminated. */
void count_kmers_in_read(char *read, hash_set *haplotype_A, hash_set *haplotype_B, int *count_A, int *count_B);
/* c/kmers.c:50-72 -> tbk_kmer_to_int. */
uint64_t kmer_to_int(char *kmer, unsigned char k);
/* c/kmers.c:74-93 -> tbk_reverse_complement. */
void reverse_complement(char *kmer_in, char *kmer_out, unsigned char k);
/* The reference never frees its tables (c/kmers.c:164-172 has no counterpart); this does. */
void free_kmer_hash_set(hash_set *set);

#ifdef __cplusplus
}
#endif
#endif /* KMERS_COMPAT_H */
