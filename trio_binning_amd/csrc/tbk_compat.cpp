// tbk_compat.cpp — the reference's own symbol names (include/kmers_compat.h), forwarding to the
// tbk_* C-ABI, so that the reference's unmodified ctypes binding (src/trio_binning/kmers.py:62-86,159)
// can load libtbk_hip.so in place of its kmers_c extension.
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <new>

#include "../../include/kmers_compat.h"

static int compat_device() {
    const char *v = getenv("TBK_DEVICE");
    return v && *v ? atoi(v) : 0;
}

static int saturate(uint64_t v) { return v > (uint64_t)INT_MAX ? INT_MAX : (int)v; }

extern "C" hash_set *create_kmer_hash_set(char *kmer_file_path) {
    tbk_table *t = nullptr;
    if (tbk_table_create_from_file(kmer_file_path, compat_device(), &t) != TBK_OK) {
        fprintf(stderr, "create_kmer_hash_set(%s): %s\n", kmer_file_path ? kmer_file_path : "(null)", tbk_last_error());
        return nullptr;
    }
    hash_set *hs = new (std::nothrow) hash_set();
    if (!hs) { tbk_table_destroy(t); return nullptr; }
    const uint64_t n = tbk_table_num_kmers(t);
    hs->kmers = nullptr;
    hs->full = nullptr;
    hs->hash_size = saturate(n * 4 / 3);  // c/kmers.c:167
    hs->k = (unsigned char)tbk_table_k(t);
    hs->num_kmers = saturate(n);
    hs->num_kmers_u64 = n;
    hs->table = t;
    // c/kmers.c:136-142 reports the same two facts on stderr
    fprintf(stderr, "Found %llu %d-mers in %s.\n", (unsigned long long)n, (int)hs->k, kmer_file_path);
    return hs;
}

extern "C" void count_kmers_in_read(char *read, hash_set *haplotype_A, hash_set *haplotype_B, int *count_A, int *count_B) {
    if (count_A) *count_A = -1;
    if (count_B) *count_B = -1;
    if (!read || !haplotype_A || !haplotype_B || !count_A || !count_B) return;
    int a = 0, b = 0;
    if (tbk_count_kmers_in_read(read, -1, haplotype_A->table, haplotype_B->table, &a, &b) != TBK_OK) {
        fprintf(stderr, "count_kmers_in_read: %s\n", tbk_last_error());
        return;
    }
    *count_A = a;
    *count_B = b;
}

extern "C" uint64_t kmer_to_int(char *kmer, unsigned char k) { return tbk_kmer_to_int(kmer, k); }

extern "C" void reverse_complement(char *kmer_in, char *kmer_out, unsigned char k) { tbk_reverse_complement(kmer_in, kmer_out, k); }

extern "C" void free_kmer_hash_set(hash_set *set) {
    if (!set) return;
    tbk_table_destroy(set->table);
    delete set;
}
