/*
 * tbk.h — C-ABI of libtbk_hip.so: the MI355X-native classify-by-kmers hot path.
 *
 * This is the drop-in boundary.  The reference crosses from Python into native code
 * through ctypes (src/trio_binning/kmers.py:30-86) into the symbols of c/kmers.c; every
 * entry point below names the reference interface it replaces.  Plain C types only:
 * pointers, sizes, opaque handles.  No torch types, no C++ in the signatures.
 *
 * Conventions
 *   - Functions returning int return TBK_OK (0) or a negative tbk_status; the message for
 *     the calling thread's last failure is tbk_last_error().
 *   - Handles are opaque, created/destroyed explicitly (the reference leaks its tables:
 *     c/kmers.c:164-172 has no destroy).
 *   - "Host" pointers are ordinary process memory; "device" pointers are HIP device
 *     memory on the handle's device.  The library never keeps a caller pointer after the
 *     call returns, except between tbk_stream_submit and the matching tbk_stream_wait.
 *   - A read batch is `bases` = the reads' ASCII bytes back to back, and
 *     `offsets[n_reads+1]`: read i is bases[offsets[i] .. offsets[i+1]).  No separators,
 *     no terminators.  Counts come back as int32 counts[n_reads][2] = {hapA, hapB}.
 *   - Any byte outside {A,C,G,T} (upper case) invalidates every window that contains it:
 *     such a window scores no hit (the reference is undefined there — c/kmers.c:78-91,279;
 *     its docstring restricts reads to [ACGT], kmers.py:134-135).
 *   - Everything that computes on the GPU fails with TBK_ERR_NO_DEVICE when no MI355X is
 *     visible.  There is no CPU fallback in this library.
 *   - Threading: a tbk_table is read-only after creation and may be shared by threads; a
 *     tbk_classifier (its streams, ticket ring and scratch) belongs to one thread at a time,
 *     like the reference's per-call scratch buffers (c/kmers.c:278-279).  tbk_last_error() is
 *     per thread.  Calls block the calling thread only where stated; the Python bindings
 *     release the GIL for the duration of every call (ctypes), as the reference's do.
 */
#ifndef TBK_H
#define TBK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBK_ABI_VERSION 1

typedef enum tbk_status {
    TBK_OK = 0,
    TBK_ERR_INVALID = -1,   /* bad argument (k out of 1..32, NULL, mismatched k, ...)       */
    TBK_ERR_IO = -2,        /* file missing / unreadable (Python side raises IOError,       */
                            /*   as kmers.py:117-118 does)                                  */
    TBK_ERR_FORMAT = -3,    /* malformed k-mer list (empty file, first line gives k > 32)    */
    TBK_ERR_NO_DEVICE = -4, /* no usable HIP device                                         */
    TBK_ERR_HIP = -5,       /* a HIP runtime call failed                                    */
    TBK_ERR_NOMEM = -6,
    TBK_ERR_STATE = -7      /* call sequence error (wait without submit, ...)               */
} tbk_status;

typedef struct tbk_table tbk_table;           /* one k-mer list resident in HBM           */
typedef struct tbk_classifier tbk_classifier; /* (hapA, hapB) + streams + staging buffers */

/* ---- library ---------------------------------------------------------------------- */
int tbk_abi_version(void);
const char *tbk_last_error(void);
int tbk_device_count(int *count);
/* Short description of device `device` ("gfx950 AMD Instinct MI355X, 256 CUs, 288 GB"). */
int tbk_device_name(int device, char *buf, size_t buflen);
int tbk_device_identity(int device, char *buf, size_t buflen);  /* "<pci bus id> <uuid>" */
/* NUMA placement (pinned, NUMA-local buffers, one feeder thread per GPU; the reference is one thread on one
 * socket, c/kmers.c:270-299).  *node = the host NUMA node the device's PCIe root belongs to
 * (/sys/bus/pci/devices/<bdf>/numa_node), -1 when the kernel does not say.  tbk_numa_bind_to_device restricts the
 * CALLING thread - and the threads it starts afterwards - to that node's CPUs inside its current affinity mask
 * (*cpus = how many; 0 = left alone: unknown node, nothing in common, TBK_NUMA=0); a pipeline's feeder threads do
 * this for their device by themselves (tbk_pipeline_numa reports it), and a one-process-per-GPU launcher calls it
 * once per rank before the library starts its worker threads.  Pinned buffers the thread allocates afterwards are
 * local: hipHostMalloc places host memory on the node nearest to the current device. */
int tbk_device_numa_node(int device, int *node);
int tbk_numa_bind_to_device(int device, int *node, int *cpus);
/* (library-internal, exported for the CPU tests: sysfs lookups under TBK_SYSFS_ROOT, and the binding itself) */
int tbk_numa_node_of_pci_(const char *pci_bus_id);
int tbk_numa_node_cpus_(int node, int *cpus, int cap);
int tbk_numa_bind_thread_(int node);

/* ---- unit-level API kept for parity with the reference's ctypes surface ------------ */
/* Replaces kmer_to_int (c/kmers.c:50-72; bound kmers.py:75-82): base i -> bits 2i..2i+1,
 * A=0 C=1 G=2 T=3, any other byte contributes 0.  Host code. */
uint64_t tbk_kmer_to_int(const char *kmer, unsigned char k);
/* Replaces reverse_complement (c/kmers.c:74-93; bound kmers.py:85-93): out[k-1-i] =
 * complement(in[i]); a non-ACGT byte leaves out[k-1-i] untouched.  Host code. */
void tbk_reverse_complement(const char *kmer_in, char *kmer_out, unsigned char k);

/* ---- k-mer lists ------------------------------------------------------------------- */
/* Replaces create_kmer_hash_set (c/kmers.c:185-229; bound kmers.py:62-64,104-122) with
 * peek_at_file's rules (c/kmers.c:124-146): k = length of the first line as getline()
 * returns it, minus one; every line is one k-mer (duplicates counted, a last line without
 * '\n' counted); each line contributes its first k bytes, verbatim (no canonicalisation).
 * The packed keys are placed in HBM on `device`; tbk_classifier_create hashes two such lists
 * into the paired open-addressing table the probe kernel reads.
 * Where the keys come from, in this order: (1) the binary key cache `<path>.tbk` when it was made from this
 * very file (same size and modification time; a stale or damaged cache is ignored; TBK_LIST_CACHE=0: never
 * read one, TBK_LIST_CACHE=1: write one after parsing the text); (2) the GPU parser, for lists in the shape
 * every tool writes them - each line k bytes and a newline - whose text is staged through pinned memory and
 * packed one line per thread (TBK_LIST_GPU_PARSE=0: off); (3) the general host parser (the reference's getline
 * rules line by line) for anything else.  All three give the same keys.  tbk_table_origin says which it was
 * (0 keys from the caller / the host parser, 1 the GPU parser, 2 the cache); tbk_table_keys copies the keys out. */
int tbk_table_create_from_file(const char *path, int device, tbk_table **out);
int tbk_table_origin(const tbk_table *t);
/* The list's packed keys where they lie in HBM on the list's device (num_kmers of them; read-only, valid until the list is
 * destroyed): what the full-membership sweep reads. */
const void *tbk_table_device_keys(const tbk_table *t);
int tbk_table_keys(const tbk_table *t, uint64_t *keys, uint64_t capacity);
/* Host-only half of the above: parse the list (all host threads when every line is k bytes +
 * newline, the sequential general parser otherwise) into malloc'd packed keys; free with
 * tbk_list_free. */
int tbk_list_parse_file(const char *path, uint64_t **keys, uint64_t *n, int *k);
void tbk_list_free(uint64_t *keys);
/* Same from already-packed keys (one per list line) in host memory. */
int tbk_table_create_from_keys(const uint64_t *keys, uint64_t n, int k, int device, tbk_table **out);
/* Same, keys already in device memory on `device` (bench generator).  The keys are copied. */
int tbk_table_create_from_device_keys(const void *d_keys, uint64_t n, int k, int device, tbk_table **out);
void tbk_table_destroy(tbk_table *t);
/* Replaces the field read hash_set->num_kmers (kmers.py:157-159): number of list LINES. */
uint64_t tbk_table_num_kmers(const tbk_table *t);
int tbk_table_k(const tbk_table *t);
int tbk_table_device(const tbk_table *t);
/* Bytes of HBM this list holds (keys + its standalone hashed form once built). */
uint64_t tbk_table_bytes(const tbk_table *t);
/* Distinct keys in the list (builds the standalone hashed form on first use). */
int tbk_table_distinct(tbk_table *t, uint64_t *distinct);
/* Membership of raw packed keys (no canonicalisation): out[i] = 1/0.  Host pointers.
 * Builds the standalone hashed form of the list on first use. */
int tbk_table_contains(tbk_table *t, const uint64_t *keys, uint64_t n, uint8_t *out);

/* ---- the hot path ------------------------------------------------------------------- */
/* Replaces count_kmers_in_read (c/kmers.c:270-299; bound kmers.py:66-73,125-154) for one
 * read: `read` is `len` bytes (len < 0: NUL-terminated).  hapA wins when a k-mer is in both
 * lists.  Both lists must have the same k (TBK_ERR_INVALID otherwise; the reference mixes
 * hapA's window length with hapB's packing there, c/kmers.c:251-253,278-290). */
int tbk_count_kmers_in_read(const char *read, int64_t len, const tbk_table *hap_a,
                            const tbk_table *hap_b, int *count_a, int *count_b);
/* The same call on a classifier of the caller's (tbk_count_kmers_in_read keeps one for the last pair of lists it saw).  The
 * reference's driver calls count_kmers_in_read once per read (classify_by_kmers.py:99-102); a call here is one memcpy into
 * pinned memory, ONE kernel launch that reads the read over PCIe, 8 bytes back and one stream synchronisation (reads of up
 * to 8 Mi bases; longer ones go through tbk_classify_batch). */
int tbk_classifier_count_read(tbk_classifier *c, const char *read, int64_t len, int *count_a, int *count_b);

/* Batch path (what the per-read Python loop classify_by_kmers.py:99-102 becomes).  Creating
 * the classifier builds the two open-addressing tables in HBM on the lists' device: 64-bit
 * slots, 8 per bucket per list, hapA's and hapB's bucket i interleaved in one 128-byte line
 * (initialize_hash_set + add_to_hash, c/kmers.c:112-122,160-180).  The lists may be
 * destroyed afterwards. */
int tbk_classifier_create(const tbk_table *hap_a, const tbk_table *hap_b, tbk_classifier **out);
void tbk_classifier_destroy(tbk_classifier *c);
/* How a classifier is built, as arguments (the reference configures itself through argparse alone,
 * classify_by_kmers.py:14-54): nothing under tbk_classifier_create* / tbk_pipeline_create* reads the environment, and two
 * classifiers of one process may be built with different options at the same time.  tbk_options_init writes the defaults;
 * tbk_classifier_create(a, b, out) = tbk_classifier_create_opts(a, b, NULL, out) = the defaults.  No option changes any
 * result (c/kmers.c:245-299: only membership is observable); they pin what the lists would decide, move the decisions'
 * thresholds and size the table.  tbk_options_from_env overlays the TBK_* variables of the process environment
 * (TBK_SHORT, TBK_ENTRY, TBK_ENTRY_WIDE, TBK_FRONT, TBK_MOD_SAMPLING, TBK_SPAN3, TBK_GUESTS, TBK_MINIMIZER_W/_M,
 * TBK_TABLE_LOAD, TBK_ENTRY_LOAD, TBK_WENTRY_LOAD, TBK_SHORT_LOAD, TBK_CLUSTERED, TBK_BEHIND_FRONT, TBK_PLAINLY_CLUSTERED,
 * TBK_ENTRY_MIN_RATIO, TBK_MEMORY_BUDGET, ...): the command-line tools' fallback, called by THEM (the Python bindings do
 * when no options are given) - never by the library's constructors. */
typedef struct tbk_options {
    uint32_t size;            /* sizeof(tbk_options) of the caller (tbk_options_init sets it): the struct may grow */
    /* the layout of the paired table: -1 the lists decide (short keys -> key layout front-first -> entries -> whole lines,
     * csrc/tbk_common.h), 0 never this one, 1 this one whatever the lists look like */
    int32_t short_keys, entries, wide_entries, front;
    int32_t mod_sampling;     /* -1 / 1: mod-sampling picks the bucket; 0: the random minimizer */
    int32_t span3;            /* -1 / 1: narrow entries and short keys rank 3w t-mer positions; 0: 2w */
    int32_t guests;           /* -1 / 1: key layouts keep a full half's surplus in the other half of its line (k < 32); 0: not */
    int32_t minimizer_w;      /* m-mers per span; -1: as many as k leaves room for (6 .. 8); 0: plain hashing of the key */
    int32_t minimizer_m;      /* m-mer length; 0: by k and the lists' size */
    int32_t two_read_kernel;  /* 0: passes that touch two reads go to the multi-read kernel */
    double table_load;        /* key layouts: keys per slot; 0: 0.08, denser only where the memory budget says so */
    double entry_load, wentry_load, short_load;  /* entries per list and bucket (0.5 / wide 0.25), short keys per line (2.3) */
    double clustered, behind_front, plainly_clustered, entry_min_ratio;  /* the policy's thresholds (0.003, 0.05, 0.12, 1.5) */
    uint64_t memory_budget_bytes;  /* what the paired table may take; 0: 60 % of the device's TOTAL memory (the same lists give
                                      the same table whatever else lives on the device) */
    uint64_t table_align;     /* alignment of the table's first byte (0: the allocator's) */
    uint32_t short_line_cap;  /* slots of a line the short-key inserts use (32; tests lower it to fill the overflow table) */
    int32_t probe_max_blocks; /* cap of the multi-read kernel's grid (0: none; tests) */
    int32_t packed_h2d;       /* 1: host batches cross PCIe in the packed transfer format; 0: as ASCII */
    uint64_t slice_bases;     /* an empty ring takes a host batch in slices of at least this many bases (384 Mi) */
    int32_t build_timing;     /* 1: every build of the paired table with its duration, on stderr */
    int32_t force_replica;    /* 1: a further ring on the table's own device gets a full replica (how one GPU runs the replica path) */
    int32_t ring_streams, copy_priority, h2d_streams, zero_copy;  /* experiments of EXPERIMENTS.md (0, 0, 1, 0) */
    int32_t full_keys;        /* the full-key layout (csrc/tbk_common.h "full keys"): -1 the lists decide, 0 never, 1 pinned */
    double full_load;         /* full keys per line of sixteen slots (2.0: 64 bytes of device memory per key) */
    int32_t replica_copy;     /* 0: the other devices of tbk_classifier_create_multi get the lists' keys and build the table themselves, all at
                                 once; 1: they get a copy of the finished table (asynchronous peer copies, one stream per destination) */
    int32_t verify_build;     /* -1: a table built by inserts that MERGE keys (entries, wide entries) is asked for every line of both lists
                                 before it is handed out (c/kmers.c:112-122: every line is stored); 1: every table; 0: none */
} tbk_options;
void tbk_options_init(tbk_options *o);
int tbk_options_from_env(tbk_options *o);
int tbk_classifier_create_opts(const tbk_table *hap_a, const tbk_table *hap_b, const tbk_options *options, tbk_classifier **out);
/* Several devices (the reference's per-read loop, classify_by_kmers.py:99-102, has no
 * cross-read state, so reads shard over devices and the tables are replicated).  The two lists are
 * hashed once, on their own device; out[i] is a classifier on devices[i] holding a copy of the
 * finished paired table (device-to-device copy, xGMI between peers) with its own streams and
 * ticket ring.  A device may appear more than once (two rings on one GPU).  There is no
 * collective and no cross-device traffic after this call: the caller deals batches to the
 * classifiers and writes results in input order (trio_binning_amd.kmers.MultiClassifier does;
 * tickets of one classifier complete in submission order).  On failure nothing is left behind. */
int tbk_classifier_create_multi(const tbk_table *hap_a, const tbk_table *hap_b, const int *devices, int n_devices,
                                tbk_classifier **out /* [n_devices] */);
int tbk_classifier_create_multi_opts(const tbk_table *hap_a, const tbk_table *hap_b, const int *devices, int n_devices,
                                     const tbk_options *options, tbk_classifier **out /* [n_devices] */);
/* One more classifier over a finished classifier's table, on `device`.  On another device the table is replicated
 * (peer access where the devices are peers, hipMemcpyPeer); on src's own device the read-only table is shared - unless
 * src was created with tbk_options.force_replica, which makes a full replica there too through the same calls (how a one-GPU box executes the
 * replica path).  tbk_classifier_table_id: where a classifier's table lies (equal ids = one shared table) and whether
 * it is such a copy. */
int tbk_classifier_replicate(const tbk_classifier *src, int device, tbk_classifier **out);
int tbk_classifier_table_id(const tbk_classifier *c, uint64_t *table_id, int *is_replica);
int tbk_classifier_device(const tbk_classifier *c);
/* Distinct keys stored per list, bucket lines, bytes of HBM the paired table holds. */
int tbk_classifier_stats(const tbk_classifier *c, uint64_t *distinct_a, uint64_t *distinct_b,
                         uint64_t *n_buckets, uint64_t *table_bytes);
/* Lines of hapB's list whose key hapA's list holds too (0 for lists made by find-unique-kmers).
 * hapA is asked first (c/kmers.c:291-294), so such a key can never count for hapB: it is left out
 * of hapB's half of the table (distinct_b above does not include it), the two halves are disjoint
 * and the probe kernel never arbitrates between them.  get_number_kmers_in_set is unaffected. */
int tbk_classifier_shared_keys(const tbk_classifier *c, uint64_t *n_shared);
/* How a key picks its bucket: the minimizer (w m-mers of length m, starting at base
 * span_offset of the k-mer) of the k-mer's central span, or the whole key when w = 0.
 * tbk_options.minimizer_w and .table_load tune it; neither changes any result. */
int tbk_classifier_layout(const tbk_classifier *c, int *minimizer_w, int *minimizer_m, int *span_offset);
/* 0: the span's m-mer with the smallest hash picks the bucket (random minimizer); t > 0: mod-sampling
 * over the span's t-mers (15 % fewer bucket switches between consecutive windows, more arithmetic per
 * window, longer runs of keys per bucket).  The lists decide: the table is built with mod-sampling
 * and, if more than 0.3 % (TBK_CLUSTERED) of the keys found their own half of their home line full
 * - lists that cluster, as real find-unique-kmers output does - built again with the random
 * minimizer and at half the load (0.04 instead of 0.08 keys per slot: such lists are the ones a
 * roomier table helps).  tbk_options.mod_sampling pins the rule, .table_load the load.  No result
 * depends on either. */
int tbk_classifier_sampling_t(const tbk_classifier *c);
/* How many times the table was built (1 or 2, see above) and how many keys found their own half of
 * their home line full in the layout that was kept. */
int tbk_classifier_build_info(const tbk_classifier *c, int *layout_builds, uint64_t *keys_past_half);
/* Layout of the paired table that stands: front = 1 when the probe kernel fetches the first 64 bytes of a
 * line only (the first four slots of each list; csrc/tbk_common.h "front layout"), and how many keys lie
 * behind that front (settled by the deferred walk).  tbk_options.front pins the layout. */
int tbk_classifier_front(const tbk_classifier *c, int *front, uint64_t *keys_behind_front);
/* Entry layout (csrc/tbk_common.h "entry layout"): lists whose keys come in runs of overlapping k-mers - what
 * find-unique-kmers writes (find_unique_kmers.py:200-233): the k windows over every variant - store a run once per
 * sampled m-mer: the m-mer, up to o + w - 1 bases on either side and one bit per window that is in the list, compared
 * under the window's mask (replaces kmer_in_hash_set's compare, c/kmers.c:245-268; membership is the same).
 * entry_layout = 1 when the table that stands is laid out so (2: WIDE entries of 16 bytes, for k-mers whose context
 * does not fit a slot - k = 26 .. 32); entries_a/_b = entries the lists' keys take.  Chosen for clustered lists whose
 * keys merge (>= tbk_options.entry_min_ratio, default 1.5, keys per entry); .entries = 1 / 0 pins it, .wide_entries the
 * wide form, .entry_load / .wentry_load set the entries per list and bucket (defaults 0.5 / 0.25).
 * entry_layout = 3: SHORT KEYS (csrc/tbk_common.h "short keys") - lists whose keys do not merge (uniform k-mers), each key
 * stored as the 32 bits its bucket does not say already, 32 to a line, plus an overflow table of full keys behind the lines
 * (tbk_classifier_stats' table_bytes counts it); entries_a/_b = words the lists' keys take.  Tried first where k (17 .. ~25,
 * m-mers of at most 16 bases) and the table's size allow; tbk_options.short_keys pins it, .short_load sets the keys per line
 * (default 2.3: 56 bytes of device memory per key; 2.6: 49).  Like the entry layouts it leaves out list lines that are not
 * canonical (no window ever asks for them, c/kmers.c:255): distinct_a/_b count the keys stored.
 * entry_layout = 4: FULL KEYS (csrc/tbk_common.h "full keys") - lists that do not merge and do not fit short keys (uniform 26- to
 * 31-mers): 64-bit keys in the same line shape, three keys and the line's summary in the 32-byte front, twelve more behind it;
 * entries_a/_b = slots the lists' keys take.  tbk_options.full_keys pins it, .full_load sets the keys per line (default 2.0). */
int tbk_classifier_entries(const tbk_classifier *c, int *entry_layout, uint64_t *entries_a, uint64_t *entries_b);
/* Random 64-byte reads, a quad of lanes per line as the probe asks for a front, over this table where it lies
 * in HBM: lines per second (a diagnostic: the same table measures up to 15 % differently from one placement in
 * the device's memory to another). */
int tbk_classifier_calibrate(tbk_classifier *c, double *lines_per_sec);
/* The same in the entry kernels' own request shape - one-wave blocks, waves_per_simd (4..8) of them resident per SIMD, two lanes x
 * 16 bytes of a line's first 32, inflight (1..8) lines per pair before any is used - over n_lines random lines of this table: the
 * ceiling bench.py prices the window loop's line rate against (same table, same box, same process). */
int tbk_classifier_calibrate_pairs(tbk_classifier *c, int inflight, int waves_per_simd, uint64_t n_lines, double *lines_per_sec);

/* Synchronous: host batch in, host counts out (pinned staging + H2D + kernel + D2H). */
int tbk_classify_batch(tbk_classifier *c, const uint8_t *bases, const uint64_t *offsets,
                       uint64_t n_reads, int32_t *counts);

/* Streaming: up to tbk_stream_depth() batches in flight; H2D of batch i+1 runs on a side
 * stream while the kernel of batch i runs.  submit returns a ticket; wait blocks until
 * that batch's counts are in `counts` (the pointer given at submit).  Tickets complete in
 * submission order.  `bases`/`offsets`/`counts` must stay valid until wait returns; buffers
 * from tbk_host_alloc are pinned and are copied without an intermediate staging copy. */
int tbk_stream_depth(const tbk_classifier *c);
int tbk_stream_submit(tbk_classifier *c, const uint8_t *bases, const uint64_t *offsets,
                      uint64_t n_reads, int32_t *counts, uint64_t *ticket);
int tbk_stream_wait(tbk_classifier *c, uint64_t ticket);
/* 1 when that ticket's batch is complete (tbk_stream_wait will not block), 0 when not yet, below -1 on error. */
int tbk_stream_query(tbk_classifier *c, uint64_t ticket);
/* The packed transfer format.  ASCII reads cost a byte per base over PCIe (Gen5 x16: ~56 GB/s, a
 * third of what the probe kernel consumes); the kernel's first step on a 16-base chunk is to turn it
 * into a 32-bit word of 2-bit codes (c/kmers.c:50-72's encoding) and a 16-bit not-ACGT mask, and that
 * step can run on the host instead.  A packed batch is
 *     codes[tbk_packed_chunks(total)]     chunk j = stream positions 16j..16j+15, base i at bits 2i..2i+1
 *     exc_chunk[n_exc], exc_mask[n_exc]   the chunks holding a byte outside ACGT, or positions at or
 *                                         past the end of the stream: chunk index and 16-bit mask
 * = 0.25 bytes per base for clean reads.  Counts are identical to the ASCII path's by construction.
 * The library ORs the exceptions into its dense mask array (an index listed twice is harmless) and marks
 * the positions at or past the end of the stream in the last, partial chunk itself, so a caller's own
 * packer, or a batch re-sliced after packing, need not list that tail.  Code bits of masked positions are
 * ignored.
 * tbk_stream_submit itself packs a host batch this way before the copy (all host threads, straight
 * from the caller's memory into pinned staging) unless TBK_PACKED_H2D=0 / tbk_classifier_set_transfer
 * (c, 0); tbk_pack_bases + tbk_stream_submit_packed let a caller pack ahead of time (a reader thread). */
uint64_t tbk_packed_chunks(uint64_t total_bases);
/* TBK_ERR_NOMEM with *n_exc set when exc_capacity is too small (tbk_packed_chunks(total) always suffices). */
int tbk_pack_bases(const uint8_t *bases, uint64_t total_bases, uint32_t *codes, uint32_t *exc_chunk, uint16_t *exc_mask,
                   uint64_t exc_capacity, uint64_t *n_exc);
int tbk_stream_submit_packed(tbk_classifier *c, const uint32_t *codes, const uint32_t *exc_chunk, const uint16_t *exc_mask,
                             uint64_t n_exc, const uint64_t *offsets, uint64_t n_reads, int32_t *counts, uint64_t *ticket);
/* packed != 0: tbk_stream_submit / tbk_classify_batch pack on the host before the copy (default). */
int tbk_classifier_set_transfer(tbk_classifier *c, int packed);
int tbk_classifier_transfer(const tbk_classifier *c);
void *tbk_host_alloc(size_t bytes);  /* pinned host memory (hipHostMalloc) */
void tbk_host_free(void *p);

/* Device-resident form (inputs already in HBM: bench.py's timed region, and callers that
 * produce reads on the GPU).  d_bases must be 16-byte aligned, d_offsets[0] must be 0 and
 * d_offsets[n_reads] must equal total_bases.  Asynchronous on the classifier's compute stream;
 * tbk_classifier_sync waits for it. */
int tbk_classify_device(tbk_classifier *c, const void *d_bases, const void *d_offsets,
                        uint64_t n_reads, uint64_t total_bases, void *d_counts);
int tbk_classifier_sync(tbk_classifier *c);
/* Device-resident batch through the same ticket ring as tbk_stream_submit: the kernel runs
 * on the compute stream and the counts are copied to `counts` (host; pinned memory from
 * tbk_host_alloc avoids a staging copy) behind it; tbk_stream_wait(ticket) returns when
 * they have arrived.  Lets the host bin batch i while the GPU probes batch i+1. */
int tbk_stream_submit_device(tbk_classifier *c, const void *d_bases, const void *d_offsets,
                             uint64_t n_reads, uint64_t total_bases, int32_t *counts, uint64_t *ticket);

/* ---- one handle over several devices ("one host feeder thread + pinned ring + 2-3 streams per
 * device", tables replicated, reads dealt in batches, no collective) ---------------------------------------
 * tbk_pipeline_create hashes the two lists once (tbk_classifier_create_multi), gives every entry of
 * `devices` a replica with its own stream ring and starts one feeder thread per entry.  submit only queues a
 * batch (ASCII, or already in the packed transfer format); the next feeder whose ring has room takes it and
 * does what is left to do on the host - packing an ASCII batch with its share of the host threads, staging,
 * launching copies and kernels - off the caller's thread.  wait(ticket) returns when that batch's counts
 * are in `counts`, whichever device computed them; *device_slot (optional) = index into `devices` of the ring
 * that did.  Tickets may be waited for in any order; the caller's arrays must stay valid until then.  At most
 * tbk_pipeline_depth() + n_devices batches may be submitted and not yet waited for.  A device may be listed
 * several times: several rings on one GPU, which share that device's (read-only) table.  The handle itself may be used from one thread at a time.
 * Replaces the per-read loop of classify_by_kmers.py:99-102 for any number of GPUs of one node. */
typedef struct tbk_pipeline tbk_pipeline;
int tbk_pipeline_create(const tbk_table *a, const tbk_table *b, const int *devices, int n_devices, tbk_pipeline **out);
int tbk_pipeline_create_opts(const tbk_table *a, const tbk_table *b, const int *devices, int n_devices, const tbk_options *options, tbk_pipeline **out);
void tbk_pipeline_destroy(tbk_pipeline *p);
int tbk_pipeline_depth(const tbk_pipeline *p);     /* sum of the rings' depths */
int tbk_pipeline_devices(const tbk_pipeline *p);
int tbk_pipeline_numa(const tbk_pipeline *p, int slot, int *node, int *cpus);  /* ring `slot`: its device's NUMA node (-1 unknown), CPUs its feeder thread is bound to (0: not bound) */
tbk_classifier *tbk_pipeline_classifier(tbk_pipeline *p, int slot);  /* for stats / timing; never submit to it directly */
int tbk_pipeline_submit(tbk_pipeline *p, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int32_t *counts, uint64_t *ticket);
int tbk_pipeline_submit_packed(tbk_pipeline *p, const uint32_t *codes, const uint32_t *exc_chunk, const uint16_t *exc_mask, uint64_t n_exc,
                               const uint64_t *offsets, uint64_t n_reads, int32_t *counts, uint64_t *ticket);
int tbk_pipeline_wait(tbk_pipeline *p, uint64_t ticket, int *device_slot);
int tbk_pipeline_batches(const tbk_pipeline *p, uint64_t *per_slot, int n);  /* batches each ring has taken so far */
/* Testing hook, no GPU needed: the same queue and feeder threads over n_rings rings whose submit / wait are the
 * caller's callbacks (same contract as tbk_stream_submit / tbk_stream_wait; `slot` says which ring asks). */
typedef int (*tbk_pipeline_test_submit_fn)(void *user, int slot, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                                           int32_t *counts, uint64_t *ticket);
typedef int (*tbk_pipeline_test_wait_fn)(void *user, int slot, uint64_t ticket);
int tbk_pipeline_create_test_(int n_rings, int ring_depth, tbk_pipeline_test_submit_fn submit, tbk_pipeline_test_wait_fn wait, void *user,
                              tbk_pipeline **out);

/* The whole read / classify / write loop of classify-by-kmers as native threads (classify_by_kmers.py:80-117):
 * a reader thread (tbk_fastx_next with packing on), the calling thread feeding the pipeline and taking the
 * batches back in input order, a writer thread (tbk_score_and_bin, tbk_bin_writer_write, tbk_format_tsv ->
 * tsv_fd; tsv_fd < 0: no TSV).  out_a/out_b/out_u are the bins' file names as the reference derives them
 * (seq.py:127-134).  batch_bases / batch_reads bound a batch (0: 64 Mbases / unbounded). */
typedef struct tbk_run_stats {
    uint64_t reads, bases, batches;
    double read_s, gpu_wait_s, write_s, total_s;  /* busy seconds of the reader / waiting for tickets / of the writer; wall */
    int32_t gzip_encoder, reserved_;              /* who coded the bins' gzip members: 0 nobody (plain bins), 1 the host, 2 the device */
} tbk_run_stats;
int tbk_classify_file(tbk_pipeline *p, const char *reads_path, uint64_t num_kmers_a, uint64_t num_kmers_b, const char *out_a,
                      const char *out_b, const char *out_u, int gzip_output, int gzip_level, int tsv_fd, uint64_t batch_bases,
                      uint64_t batch_reads, tbk_run_stats *stats);

/* HIP-event timing of the probe on the stream it is launched on.  A probe is four kernels: the pass
 * index, the kernels of the passes that touch several reads (more than two; exactly two), and the kernel of the
 * passes inside one read (on long reads nearly all the work: the dominant kernel).  While enabled, every probe of
 * this classifier is bracketed by events - before the first kernel, in front of the last, after the last; read
 * returns the number of probes and their summed duration since enable (total_ms: all kernels; single_ms:
 * the single-read kernel alone), and resets. */
int tbk_kernel_timing_enable(tbk_classifier *c, int on);
int tbk_kernel_timing_read(tbk_classifier *c, uint64_t *launches, double *total_ms);
int tbk_kernel_timing_read2(tbk_classifier *c, uint64_t *launches, double *total_ms, double *single_ms);
/* Passes (2048 window starts each) of the classifier's most recent probe, and how many of them touched more
 * than one read: the two-read and multi-read kernels' share of the work; the rest was the single-read kernel's. */
int tbk_classifier_last_passes(tbk_classifier *c, uint64_t *n_passes, uint64_t *n_multi);

/* Replaces calculate_scaling_factors (classify_by_kmers.py:57-77) and the binning rule
 * (classify_by_kmers.py:104-115): float64, same operation order (1.0*max/n, count*factor,
 * strict > both ways, else 'U').  Host code; bins[i] in {'A','B','U'}. */
int tbk_score_and_bin(const int32_t *counts, uint64_t n_reads, uint64_t num_kmers_a,
                      uint64_t num_kmers_b, double *score_a, double *score_b, char *bins);

/* ---- I/O either side of the path ------------------------------------ */
typedef struct tbk_fastx_reader tbk_fastx_reader; /* FASTA/FASTQ(.gz) record source            */
typedef struct tbk_fastx_batch tbk_fastx_batch;   /* one batch of records in C-ABI batch layout */
typedef struct tbk_bin_writer tbk_bin_writer;     /* the three output bins                      */

/* Replaces open_fastx_read + readfq (seq.py:45-92): gzip chosen by the ".gz" name suffix,
 * universal newlines, and exactly readfq's record rules (name = header up to the first
 * space, every line loses its last character, '+' switches to quality, quality is read until
 * its length reaches the sequence's, a short quality section turns the record into FASTA...). */
int tbk_fastx_open(const char *path, tbk_fastx_reader **out);
void tbk_fastx_close(tbk_fastx_reader *r);
int tbk_fastx_batch_create(tbk_fastx_batch **out);
void tbk_fastx_batch_destroy(tbk_fastx_batch *b);
/* Fill `b` with the next records: stops after the record that reaches max_bases or max_reads
 * (0 = no limit).  An empty batch means end of input.  The sequence bytes of a batch lie back
 * to back in pinned host memory: pass them straight to tbk_stream_submit. */
int tbk_fastx_next(tbk_fastx_reader *r, tbk_fastx_batch *b, uint64_t max_bases, uint64_t max_reads);
/* on != 0: every batch of this reader also carries its bases in the packed transfer format (see
 * tbk_stream_submit_packed), made while the records are copied - the chunk-parallel scan packs each record's
 * 16-base chunks right after copying it - so that the classify stage moves a quarter of the bytes and
 * nobody reads the batch a second time.  tbk_fastx_batch_packed hands the arrays out (*codes = NULL when the
 * batch carries none); they stay valid until the batch is refilled or destroyed. */
int tbk_fastx_set_packing(tbk_fastx_reader *r, int on);
/* BGZF input (bgzip / htslib: independent gzip members of <= 64 KiB, which the reference reads through gzip.open like any .gz,
 * seq.py:86-92) inflated on `device` - csrc/tbk_gdeflate.hip: one wave per block, CRC-32s checked there - instead of on the host's
 * threads.  Before the first read; other inputs are read as before.  tbk_fastx_inflates_on_device: 1 when that is what happens.  The text
 * arrives in windows of pinned memory, four deep; the chunk-parallel scan reads it there, and with tbk_fastx_set_borrowing a batch may
 * leave its records in a window (while two more are free) - it keeps the window, and the inflater's memory, alive until it is refilled
 * or destroyed.  When the pinned memory for two windows cannot be had, the host's threads inflate after all. */
int tbk_fastx_set_device(tbk_fastx_reader *r, int device);
int tbk_fastx_inflates_on_device(const tbk_fastx_reader *r);
int tbk_fastx_batch_packed(const tbk_fastx_batch *b, const uint32_t **codes, const uint32_t **exc_chunk, const uint16_t **exc_mask,
                           uint64_t *n_exc);
/* on != 0 (and packing on): batches taken from a plain FASTQ file by the chunk-parallel scan do not copy their
 * records.  The reader maps its input; such a batch records where each record lies, carries names, offsets,
 * has_qual and the packed form as usual - the bases are packed straight from the mapping - and has no `bases` /
 * `quals` arrays (tbk_fastx_batch_view hands out one zero byte for them).  tbk_bin_writer_write writes the
 * records to their bins from the mapping: the bytes Read.print would write (seq.py:27-31), a record without a
 * header comment and with a bare '+' line as it stands in the input.  Such batches hold a reference to
 * the mapping (it is unmapped with its last holder); a batch that is refilled lets the pages of the mapping its old records lay in go (they stay in
 * the page cache: the 30 GB mapping of a large input is not torn down all at once at the end).  Batches of any other input (gzip, FASTA, irregular records) are copied as always;
 * tbk_fastx_batch_borrowed says which kind a batch is.  The loop of tbk_classify_file turns this on
 * (TBK_BORROW=0 turns it off).  A borrowed batch keeps the mapping alive: it stays readable (and writable to
 * the bins) after tbk_fastx_close of its reader, until it is refilled or destroyed.
 * tbk_fastx_batch_gather copies a batch's sequences (base_off[n_reads] bytes) and qualities (qual_off[n_reads] bytes)
 * into the caller's buffers, back to back, wherever they lie - for a borrowed batch the only way to see them as
 * arrays (either pointer may be NULL; TBK_ERR_INVALID when a buffer is too small). */
int tbk_fastx_set_borrowing(tbk_fastx_reader *r, int on);
int tbk_fastx_batch_borrowed(const tbk_fastx_batch *b);
int tbk_fastx_batch_gather(const tbk_fastx_batch *b, uint8_t *bases, uint64_t bases_cap, uint8_t *quals, uint64_t quals_cap);
/* Borrow the batch's arrays: offsets have n_reads+1 entries; has_qual[i] = 1 when the record
 * was read as FASTQ (readfq's qual is not None). */
int tbk_fastx_batch_view(const tbk_fastx_batch *b, uint64_t *n_reads, const uint8_t **bases,
                         const uint64_t **base_off, const uint8_t **names, const uint64_t **name_off,
                         const uint8_t **quals, const uint64_t **qual_off, const uint8_t **has_qual);
/* Replaces open_outfiles + Read.print (seq.py:27-42,98-136): truncating open of the three
 * files; a record is written as FASTQ when it has a non-empty quality string, else as FASTA;
 * gzip output is a sequence of gzip members deflated by `threads` threads (0 = all cores) at
 * `level` (<0: 6).  Decompressed bytes equal the reference's; the container bytes do not
 * (they never do: gzip stores a timestamp). */
int tbk_bin_writer_open(const char *path_a, const char *path_b, const char *path_u, int gzip_output,
                        int level, int threads, tbk_bin_writer **out);
/* bins[i] in {'A','B','U'} for every record of the batch; records keep input order per bin. */
/* gzip output: code the members on `device` (csrc/tbk_gdeflate.hip: per-block Huffman coding as kernels, the members' CRC-32s summed
 * on the host meanwhile, one writer thread per bin) instead of on the host's threads.  Right after tbk_bin_writer_open.  A no-op for
 * plain output, level 0, TBK_GZIP_ENCODER=cpu / zlib.  tbk_bin_writer_encoder: 1 when the device codes, 0 when the host does.
 * The decompressed bytes are the same either way (seq.py:27-42,132-134). */
int tbk_bin_writer_use_device(tbk_bin_writer *w, int device);
/* The same encoder by itself: n_members pieces of text (back to back in `text`, member_len[i] bytes each) -> as many gzip members,
 * back to back in dst (member_out_len[i] bytes each; *need = bytes used, or needed when cap is too small: TBK_ERR_NOMEM). */
int tbk_gzip_members_device(int device, const char *text, const uint64_t *member_len, uint64_t n_members, char *dst, uint64_t cap,
                            uint64_t *member_out_len, uint64_t *need);
/* The encoder timed by itself (tools/measure_gdeflate.py): `reps` jobs of these members (text in pinned host memory: tbk_host_alloc)
 * through the three-deep ring - *pipelined_s per job, the link included - and one job's kernels between HIP events on their own
 * stream - *kernels_s; *out_bytes = bytes of members a job makes. */
int tbk_gzip_bench_device(int device, const char *text, const uint64_t *member_len, uint64_t n_members, int reps, double *pipelined_s,
                          double *kernels_s, uint64_t *out_bytes);
/* The other direction (csrc/tbk_gdeflate.hip, second half): a bgzf file - the chain of independent <= 64 KiB gzip members that bgzip and
 * htslib write, which the reference reads through gzip.open like any .gz (seq.py:86-92) - inflated on `device`, one wave per block, every
 * block's CRC-32 checked there.  data / size: the file's bytes (or a run of whole blocks); *text_len: bytes of text.  The reader
 * (tbk_fastx_set_device) drives the same inflater four windows deep. */
int tbk_bgzf_inflate_device(int device, const uint8_t *data, uint64_t size, uint8_t *dst, uint64_t cap, uint64_t *text_len);
/* The inflater timed by itself (bench.py's `input_bgzf_inflater`): one window of bgzf blocks, `reps` times - *kernels_s per window with the
 * input resident (HIP events around inflate, CRC-32 and check), *ring_s per window through the ring as the reader drives it (staging
 * copy, copy in, kernels, text home; two windows in flight); *text_bytes: a window's text. */
int tbk_bgzf_bench_device(int device, const uint8_t *data, uint64_t size, int reps, double *ring_s, double *kernels_s, uint64_t *text_bytes);
int tbk_bin_writer_encoder(const tbk_bin_writer *w);
int tbk_bin_writer_write(tbk_bin_writer *w, const tbk_fastx_batch *b, const char *bins);
int tbk_bin_writer_close(tbk_bin_writer *w);
/* The writer's own encoder for gzip members whose bytes do not come in runs (FASTQ of long reads:
 * nothing for LZ77 to match, so entropy coding only - dynamic-Huffman DEFLATE blocks cut at line ends,
 * so that bases and qualities get codes of their own).  One complete gzip member for src[0..n);
 * TBK_ERR_NOMEM with *len = the size needed when cap is too small (2n + 1024 always suffices).
 * Members with runs go through zlib (Z_RLE);
 * TBK_GZIP_ENCODER=zlib sends everything there. */
int tbk_gzip_member(const char *src, size_t n, char *dst, size_t cap, size_t *len);
/* zlib's crc32(crc, p, n) - the CRC-32 of gzip members - by carry-less multiplication (PCLMULQDQ folding,
 * csrc/tbk_crc.cpp; zlib's own where the CPU lacks the instruction or TBK_CRC=zlib): what the reader
 * and the bin writer sum their members with. */
uint32_t tbk_crc32_c(uint32_t crc, const uint8_t *p, size_t n);
/* Replaces the stdout line of classify_by_kmers.py:117 for a whole batch:
 * name \t bin \t str(score_a) \t str(score_b) \n with Python's float repr.  Call with out = NULL
 * to get an upper bound of the size in *len. */
int tbk_format_tsv(const tbk_fastx_batch *b, const char *bins, const double *score_a, const double *score_b,
                   char *out, size_t cap, size_t *len);
/* Python's str(float) of one value (returns the length; cap >= 40). */
int tbk_format_float(double v, char *out, size_t cap);

/* ---- device memory helpers for callers without a HIP binding (bench.py, tests) ------- */
int tbk_device_alloc(int device, size_t bytes, void **d_ptr);
int tbk_device_free(int device, void *d_ptr);
int tbk_memcpy_h2d(int device, void *d_dst, const void *h_src, size_t bytes);
int tbk_memcpy_d2h(int device, void *h_dst, const void *d_src, size_t bytes);
int tbk_device_sync(int device);
int tbk_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- synthetic workload generators (BASELINE.json's bench inputs) -------- */
/* Writes keys i in [first, first+n) of the deterministic k-mer sequence for `seed`: each
 * is a distinct canonical k-mer (distinct i -> distinct k-mer), packed as above. */
int tbk_synth_keys_device(int device, uint64_t seed, uint64_t first, uint64_t n, int k, void *d_keys);
/* Host restatement of the same sequence (tests, small sizes). */
int tbk_synth_keys_host(uint64_t seed, uint64_t first, uint64_t n, int k, uint64_t *keys);
/* Fills d_bases with n_reads reads of read_len uniform random ACGT and plants list k-mers
 * (random strand, non-overlapping slots): a read's origin is A/B/none with p .45/.45/.10;
 * origin reads get `plant_major` k-mers of their list and `plant_minor` of the other,
 * origin-less reads get plant_minor+plant_minor.  d_offsets gets n_reads+1 uint64.
 * Keys [0,n_a) of `key_seed` are list A, [n_a, n_a+n_b) list B. */
int tbk_synth_reads_device(int device, uint64_t read_seed, uint64_t first_read, uint64_t n_reads,
                           uint32_t read_len, uint64_t key_seed, uint64_t n_a, uint64_t n_b, int k,
                           int plant_major, int plant_minor, void *d_bases, void *d_offsets);

/* Lists and reads shaped like real trio-binning input instead of uniform keys: an implicit random
 * genome of `genome_len` bases, two haplotypes that each differ from it by SNPs at
 * snp_per_2p24 / 2^24 per base; list A / list B = the canonical k-mers of haplotype A / B that
 * cover a position where the haplotypes differ (runs of up to k overlapping k-mers sharing a few
 * minimizers, both lists clustered at the same loci).  Both lists get *n_keys entries (order not
 * deterministic, sets are); at most `capacity` are written to each of d_keys_a / d_keys_b.
 * Bits 24..31 of snp_per_2p24, when set, add repeats: that many 256ths of the genome's 8192-base
 * blocks are copies of one of 16 family sequences, each copy diverged at 2 % of its positions. */
int tbk_synth_hap_keys_device(int device, uint64_t seed, uint64_t genome_len, uint32_t snp_per_2p24, int k,
                              void *d_keys_a, void *d_keys_b, uint64_t capacity, uint64_t *n_keys);
/* Read r comes from haplotype (first_read + r) & 1, from a hashed position and strand, with
 * substitution errors at err_per_2p24 / 2^24 per base. */
int tbk_synth_hap_reads_device(int device, uint64_t seed, uint64_t genome_len, uint32_t snp_per_2p24,
                               uint64_t read_seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                               uint32_t err_per_2p24, void *d_bases, void *d_offsets);

/* Reads of any lengths (BASELINE configs[4]: "50x ONT ultra-long (N50 100 kb)"; the reference takes a read of any length
 * whole, c/kmers.c:285-287).  tbk_synth_lognormal_lengths (host code) writes offsets[0 .. n_reads]: read first_read + i gets a
 * log-normal length whose base-weighted median (N50) is n50 - ln L ~ N(ln n50 - sigma^2, sigma^2); sigma 0.9: median 44 kb,
 * mean 67 kb, 2.7e-4 of the reads beyond 1 Mb - except that a share short_fraction of the reads is debris, log-uniform
 * between min_len and 5 kb; lengths are clamped to [min_len, max_len].  A read's length depends on (seed, its index) alone.
 * The two generators below fill d_bases for offsets already on the device (d_offsets[n_reads] = total_bases): uniform
 * background with one planted list k-mer per slot_len bases (slot_len 454 = the 33 plants per 15 kb of
 * tbk_synth_reads_device; origin A / B / none as there, every 11th plant of an origin read from the other list), and
 * reads drawn from the two haplotypes (longest_read <= genome_len). */
int tbk_synth_lognormal_lengths(uint64_t seed, uint64_t first_read, uint64_t n_reads, double n50, double sigma, double short_fraction,
                                uint32_t min_len, uint32_t max_len, uint64_t *offsets);
int tbk_synth_reads_ragged_device(int device, uint64_t read_seed, uint64_t first_read, uint64_t n_reads, const void *d_offsets, uint64_t total_bases,
                                  uint64_t key_seed, uint64_t n_a, uint64_t n_b, int k, uint32_t slot_len, void *d_bases);
int tbk_synth_hap_reads_ragged_device(int device, uint64_t seed, uint64_t genome_len, uint32_t snp_per_2p24, uint64_t read_seed, uint64_t first_read,
                                      uint64_t n_reads, const void *d_offsets, uint64_t total_bases, uint64_t longest_read, uint32_t err_per_2p24, void *d_bases);

/* ---- the full-membership sweep (tests/test_gpu_scale.py, bench.py --sweep) -------------------------------------------
 * The reference stores every list line (add_to_hash, c/kmers.c:112-122) and finds every stored canonical key
 * (kmer_in_hash_set, c/kmers.c:245-268).  The paired tables here hold compressed and merged forms of the keys (short keys,
 * entries, wide entries), so that claim is checked key by key at the tables' full size: tbk_classifier_sweep_keys lays the n
 * keys at d_keys (device memory, packed as a list's) out as reads - key i as it stands when i is even, reverse-complemented
 * when odd; keys_per_read = 1: one read of k bases per key (multi-read passes); P > 1: P keys to a read with an 'N' between
 * neighbours, so that a read counts exactly its member keys (single-read and two-read passes) - chunk_keys at a time (0: 2^25),
 * classifies every chunk through tbk_classify_device and compares the counts on the device with what they must be:
 * expect = 1: every key counts (1, 0); 2: (0, 1); 0: (0, 0); or per key from d_expect (uint8, same codes: a read of P keys must count how many of its keys say 1 and how many 2).
 * out[0], out[1] = the sums of the hapA / hapB counts; out[2] = reads that differ from their expectation; out[3] = the index
 * of the first such read (all ones: none).
 * tbk_sweep_expectation_device writes d_expect for arbitrary keys from the two lists' STANDALONE tables (verbatim 64-bit
 * keys, plain hashing - tbk_table_contains; nothing of the paired table's layouts): 1 when hapA's list holds the key's
 * canonical form, else 2 when hapB's does, else 0.  tbk_synth_mutate_keys_device turns keys into near misses: one base
 * substituted (position and base hashed from seed and first + i), canonicalised - a non-member, almost always, that shares
 * m-mer, position and most flank bits with a member.  tbk_table_contains_device: tbk_table_contains with keys and answers
 * in device memory. */
int tbk_table_contains_device(tbk_table *t, const void *d_keys, uint64_t n, void *d_out);
int tbk_synth_mutate_keys_device(int device, const void *d_keys, uint64_t first, uint64_t n, int k, uint64_t seed, void *d_out);
int tbk_sweep_expectation_device(tbk_table *a, tbk_table *b, const void *d_keys, uint64_t n, void *d_expect);
int tbk_classifier_sweep_keys(tbk_classifier *c, const void *d_keys, uint64_t n, int k, uint32_t keys_per_read, int expect, const void *d_expect,
                              uint64_t chunk_keys, uint64_t out[4]);
/* Every line of both lists through the finished table, checked against the lists' standalone tables: the verification a build can
 * be given in the field (a concurrent build - compare-and-swap on slots, the wide entries' per-piece lock - leaves no other
 * trace of a lost or misfiled key).  out = {lines checked, keys counted for hapA, for hapB, lines that differ from what the
 * standalone tables say, the first such line - hapA's lines first, then hapB's; all ones: none}.  Under a second at
 * 2 x 3e8 keys; the standalone tables take 32 bytes of device memory per list line while it runs.  The command-line tool runs it
 * when TBK_VERIFY_BUILD=1. */
int tbk_classifier_verify(tbk_classifier *c, tbk_table *a, tbk_table *b, uint64_t out[5]);
/* What tbk_options.verify_build did when the classifier was made: list lines looked up again (0: not verified) - all of them
 * answered as the lists say, or the constructor would have failed - and the seconds that took. */
int tbk_classifier_verified(const tbk_classifier *c, uint64_t *lines, double *seconds);

/* ---- k-mer counting: the find-unique-kmers step ---------------------------
 * Replaces the KMC subprocesses of find_unique_kmers.py:62-233 by a counting table in HBM.
 * Semantics restated from KMC 3 at the reference's settings (kmc -k<k>, defaults -ci2 -cs255;
 * kmc_tools transform histogram; kmc_tools simple kmers_subtract; kmc_dump -ci -cx): canonical
 * k-mers of all reads, both strands, k-mers holding a symbol outside ACGT skipped, lower case
 * counted as upper case; k-mers seen once are not in the database; counters saturate at 255.
 * KMC itself is not part of the reference checkout: parity with it is unpinned.
 * Limit: the table's counters are 32-bit and neighbours share a 64-bit atomic add, so a k-mer that occurs
 * 2^32 times or more in one library (a satellite k-mer of a very deep read set) would carry into the
 * neighbouring slot's counter; readers cap at 255 long before, but the neighbour's count would be off by
 * the carry.  Not reachable below 4.3e9 occurrences of ONE k-mer; split such a library over two counters. */
typedef struct tbk_counter tbk_counter;
/* Table for `capacity_kmers` distinct k-mers to start with (16 bytes per slot at load <= 0.6: a
 * bucket is one 128-byte line holding 8 keys and their 8 counters).  Before a batch that could
 * fill it the table is rebuilt twice as large (old and new coexist for the move); TBK_ERR_NOMEM
 * when HBM cannot hold that. */
int tbk_counter_create(int k, uint64_t capacity_kmers, int device, tbk_counter **out);
void tbk_counter_destroy(tbk_counter *c);
/* Count the canonical k-mers of a batch of reads (the classifier's batch layout; host memory). */
int tbk_counter_add_batch(tbk_counter *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads);
/* Same for a batch already in HBM (d_bases readable up to total_bases). */
int tbk_counter_add_device(tbk_counter *c, const void *d_bases, const void *d_offsets, uint64_t n_reads, uint64_t total_bases);
/* HIP-event timing of the counting kernel (every launch is bracketed by an event pair on the stream
 * it runs on): launches, window starts they covered and their summed duration since the last reset. */
int tbk_counter_kernel_timing(tbk_counter *c, uint64_t *launches, uint64_t *window_starts, double *total_ms, int reset);
int tbk_counter_adds_issued(tbk_counter *c, uint64_t *adds);
/* hist[c], c = 1..255: number of distinct k-mers whose counter (capped at 255) is c - the rows
 * kmc_tools writes, except that KMC's -ci2 database has no row-1 k-mers (callers zero hist[1]);
 * hist[0]: all distinct k-mers met. */
int tbk_counter_histogram(tbk_counter *c, uint64_t hist[256]);
/* Distinct k-mers met so far (slots taken). */
int tbk_counter_distinct(const tbk_counter *c, uint64_t *distinct);
int tbk_counter_stats(const tbk_counter *c, uint64_t *n_slots, uint64_t *table_bytes, uint64_t *bases_added, uint64_t *reads_added);
/* kmers_subtract + kmc_dump: write to out_path, one k-mer per line in lexicographic order, the
 * k-mers of `a` seen at least twice whose counter lies in [min_count, max_count] and that `b` has
 * seen at most once. */
int tbk_counter_unique(tbk_counter *a, tbk_counter *b, uint32_t min_count, uint32_t max_count, const char *out_path,
                       uint64_t *n_written);

/* Host threads the library starts for its own host-side work (list parsing, gzip members,
 * scoring): hardware threads limited by the CPU affinity mask and the cgroup CPU quota, divided by the
 * number of ranks the launcher started on this node (LOCAL_WORLD_SIZE, or TBK_LOCAL_RANKS): one process per GPU
 * means N processes sharing the node's CPUs.  Env TBK_HOST_THREADS overrides. */
int tbk_host_threads(void);

/* ---- roofline calibration (SURVEY §8d "random-read roofline") -------------------------- */
/* Independent uniformly random line-aligned loads over a `footprint_bytes` buffer:
 * `line_bytes` in {64,128}, `lanes_per_line` in {1,4,8} (16 B per lane when >1, the whole
 * line per lane when 1), `loads_in_flight` per lane in 1..8.  Reports lines/s. */
int tbk_calib_gather(int device, uint64_t footprint_bytes, int line_bytes, int lanes_per_line,
                     int loads_in_flight, uint64_t n_lines, int reps, double *lines_per_sec,
                     double *ms_per_rep);
/* Fire-and-forget 32-bit atomic adds over a `footprint_bytes` buffer: each lane sends `run` (1..32)
 * consecutive adds to consecutive words of one random 128-byte line, then moves to another line.
 * The ceiling of the k-mer counting kernel (one add per k-mer occurrence). */
int tbk_calib_atomics(int device, uint64_t footprint_bytes, int run, int reps, double *atomics_per_sec);
/* The same with the instruction the counting kernel really issues: 64-bit adds that count two neighbouring 32-bit counters
 * at once, `run` (1..4) of them on the four counter words of one random line.  tbk_counter_adds_issued: how many of those
 * the kernel has sent so far - adds per second over this ceiling is the counting kernel's roofline fraction (<= 1). */
int tbk_calib_atomics64(int device, uint64_t footprint_bytes, int run, int reps, double *atomics_per_sec);
/* Streaming read of the same buffer (the 6.3 TB/s figure on this box). */
int tbk_calib_stream(int device, uint64_t footprint_bytes, int reps, double *bytes_per_sec);
/* The same two ceilings in tuned shapes (round 5: a yardstick must not sit below what it measures).  tbk_calib_gather_pairs: the
 * entry kernels' own request shape - one-wave blocks, `waves_per_simd` (1..8) of them resident per SIMD, two lanes x 16 bytes of a
 * line's first 32, `loads_in_flight` (1, 2, 3, 4, 6, 8) independent lines per pair before any is used.  tbk_calib_stream_nt:
 * `unroll` (1, 2, 4, 8) x 16 bytes per thread in flight, non-temporal loads, a grid of `blocks` x 256 threads.
 * tools/calib_ceilings.py sweeps both and writes profiles/calibration.json's ceilings. */
int tbk_calib_gather_pairs(int device, uint64_t footprint_bytes, int loads_in_flight, int waves_per_simd, uint64_t n_lines, int reps, double *lines_per_sec);
int tbk_calib_stream_nt(int device, uint64_t footprint_bytes, int unroll, int blocks, int reps, double *bytes_per_sec);

#ifdef __cplusplus
}
#endif
#endif /* TBK_H */
