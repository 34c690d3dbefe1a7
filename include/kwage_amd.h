/* include/kwage_amd.h -- C ABI of the MI355X-native `kwage` search engine.
 *
 * This is the drop-in boundary for the one hot path of LANL-Bioinformatics/KWAGE: the body
 * of `search()` (reference kwage.cpp:26-31 declaration, :340-541 body) and the two loops in
 * `main` that call it (kwage.cpp:116-148).  The reference has no FFI of its own; a
 * maintainer replaces those loops with the calls below (INTEGRATION.md shows the patch).
 *
 * Conventions
 *   - plain C types only; every function returns KWAGE_OK (0) or a negative status and
 *     records a message retrievable with kwage_last_error() (thread local) -- this replaces
 *     the reference's `throw "file:func: msg"` literals (kwage.cpp:419,452; hash.cpp:92).
 *   - objects are opaque handles, created/destroyed explicitly; buffers returned by the
 *     library stay owned by it until the matching free / destroy call.
 *   - one kwage_ctx == one GPU == one HIP stream.  A ctx and the objects made from it may be
 *     used from one thread at a time; different ctxs are independent (this mirrors the
 *     reference's per-OpenMP-thread ifstream + result map, kwage.cpp:76-89).
 *   - there is NO CPU fallback: without a usable gfx950 device kwage_init fails.
 */
#ifndef KWAGE_AMD_H
#define KWAGE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KWAGE_AMD_ABI_VERSION 1

enum {
	KWAGE_OK = 0,
	KWAGE_ERR_ARG = -1,        /* bad argument / unsupported parameter                    */
	KWAGE_ERR_DEVICE = -2,     /* HIP error (no device, out of memory, launch failure)    */
	KWAGE_ERR_IO = -3,         /* file could not be opened / read / is truncated          */
	KWAGE_ERR_FORMAT = -4,     /* not a KWAGE database / unsupported compression          */
	KWAGE_ERR_HASH = -5,       /* unknown hash function (reference hash.cpp:92)           */
	KWAGE_ERR_STATE = -6       /* call order violated (e.g. search before finalize)       */
};

/* Limits the reference compiles in (word.h:10, bloom.h:20-21). */
#define KWAGE_MAX_WORD_LEN 32
#define KWAGE_MIN_NUM_HASH 1
#define KWAGE_MAX_NUM_HASH 5
#define KWAGE_HASH_MURMUR32 0      /* hash.h:9 MURMUR_HASH_32 */

/* search flags */
#define KWAGE_SEARCH_EARLY_EXIT 1u /* kwage.cpp:437-483: stop a query tile once no column can
                                      still match. Never changes results, only work done.   */
#define KWAGE_SEARCH_TIMING     2u /* record the HIP-event duration of the gather kernel in the result */
#define KWAGE_SEARCH_TIMING_KMER 4u /* with TIMING: also time the k-mer stage (two more events)         */

const char *kwage_last_error(void);
uint32_t kwage_abi_version(void);

/* ------------------------------------------------------------------------------------
 * Device context
 * ---------------------------------------------------------------------------------- */
typedef struct kwage_ctx kwage_ctx;

/* Number of visible HIP devices (0 if none / no driver). Never fails. */
int kwage_device_count(void);
/* Bind a context to HIP device `device` and create its stream. */
int kwage_init(int device, kwage_ctx **out);
void kwage_shutdown(kwage_ctx *ctx);
/* Free / total device memory in bytes. */
int kwage_mem_info(kwage_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
/* Identity of the device behind a context, as one line of `key=value` pairs separated by `;` -- `uuid` (hipDeviceGetUuid,
 * hex), `name`, `arch`, `cus`, `sclk_mhz`, `mclk_mhz`, `hbm_bus_bits`, `pci` (domain:bus:device) -- written NUL-terminated
 * into `buf` (at most `len` bytes; 256 is enough).  The boxes of a pool differ by several per cent in what their HBM
 * delivers: every measurement this repo files carries the line, so two of them can be paired or told apart. */
int kwage_device_fingerprint(kwage_ctx *ctx, char *buf, uint64_t len);
/* Block until everything queued on the context's stream has finished. */
int kwage_sync(kwage_ctx *ctx);

/* Kernel-selection knobs of a context ("walk", "walk_waves", "force_segs", "count_walk", "and_vec", ...: the table in
 * kwage_amd/csrc/engine.hip).  Their values are read ONCE, when the context is created, from the environment
 * (KWAGE_<NAME IN CAPITALS>); afterwards only these calls change them -- nothing on the search path looks at the
 * environment.  For tests and tuning tools: no knob changes any result.  Not while a search is pending. */
int kwage_ctx_set_tuning(kwage_ctx *ctx, const char *name, int64_t value);
int kwage_ctx_get_tuning(kwage_ctx *ctx, const char *name, int64_t *value);
/* Diagnostic: the persistent gather kernels (and_walk_kernel, and_band_walk_kernel, count_walk_kernel) finish the (query,
 * tile) pairs their wave shares cut through small exchange buffers in HBM that they must leave ALL ZERO -- they are
 * cleared when allocated, never per search.  Counts the non-zero 32-bit words of both search slots' buffers:
 * out[0] cut-pair masks, out[1] cut-pair counts + flags (and_walk), out[2] per-query masks, out[3] per-query flags
 * (and_band_walk), out[4] tree arrival counters (count_walk).  Waits for the context's streams first.  tools/soak_walk.py. */
int kwage_ctx_scratch_nonzero(kwage_ctx *ctx, uint64_t out[5]);
/* Diagnostic: what the LAST early-exit search of each slot handed over from its screen launch to its refine launch
 * (kernels.hpp and_screen_kernel / count_screen_kernel): per slot k, out[4k + 0] cluster places, [1] item places and
 * [2] unit places taken (static parts included, i.e. what the refine / emit launches scanned), [3] the capacity of the
 * unit list.  Waits for the context's streams first.  tools/step_breakdown.py, tests. */
int kwage_ctx_refine_stats(kwage_ctx *ctx, uint64_t out[8]);

/* ------------------------------------------------------------------------------------
 * Database group: all columns (samples) that share (kmer_len, num_hash, log_2_filter_len,
 * hash_func), concatenated into ONE wide row-major bit matrix resident in HBM.
 * Replaces the per-file `seekg + slice.read` of kwage.cpp:414-416 / :447-449.
 * Row r, global column c  <->  byte r*row_stride + c/8, bit c%8 (LSB first, bloom.h:143,162).
 * ---------------------------------------------------------------------------------- */
typedef struct kwage_group kwage_group;

typedef struct {
	uint32_t kmer_len;          /* 1..32                    (DBFileHeader::kmer_len)         */
	uint32_t num_hash;          /* 1..5                     (DBFileHeader::num_hash)         */
	uint32_t log_2_filter_len;  /* rows = 1 << this, <= 32  (DBFileHeader::log_2_filter_len) */
	int32_t  hash_func;         /* KWAGE_HASH_MURMUR32      (DBFileHeader::hash_func)        */
} kwage_params;

/* Allocate a group able to hold `column_capacity` columns (rounded up internally so that the
 * row stride is a multiple of 128 bytes). Rows are zero-initialised. */
int kwage_group_create(kwage_ctx *ctx, const kwage_params *params, uint64_t column_capacity,
                       kwage_group **out);
void kwage_group_destroy(kwage_group *g);

/* A SPARSE group: the resident matrix holds only the slices `rows[0..n_rows)` (strictly ascending row indices below
 * 2^log_2_filter_len) of every file added to it -- for a few queries against a database far larger than what they
 * address.  The reference reads only the addressed slices too (one seekg + read per k-mer and hash,
 * kwage.cpp:414-416); here the distinct addressed slices of a whole batch are fetched once per file (host threads,
 * I/O proportional to the queries, not to the database), and searches translate row indices to positions in the
 * list on the device.  Get the rows of a batch from kwage_hash_batch (sort + unique them); searching a batch that
 * addresses a row outside the list fails with KWAGE_ERR_STATE.  kwage_group_add_columns takes images of n_rows rows
 * for such a group; hits, columns and everything else are as for a full group. */
int kwage_group_create_sparse(kwage_ctx *ctx, const kwage_params *params, uint64_t column_capacity,
                              const uint32_t *rows, uint64_t n_rows, kwage_group **out);

/* Append `num_filter` columns from a host image of a file's bit-slice block
 * (2^L rows of ceil(num_filter/8) bytes, `host_row_stride` bytes apart).  The block is placed at
 * the next byte-aligned column; *first_column receives the global index of its column 0. */
int kwage_group_add_columns(kwage_group *g, const void *host_rows, uint64_t host_row_stride,
                            uint32_t num_filter, uint64_t *first_column);

/* Append every column of a NO_COMPRESSION `.db` file (header kwage.h:30-72; body
 * build_db.cpp:243-315), streaming the slice block to the device through pinned staging
 * buffers.  The file's parameters must equal the group's. */
int kwage_group_add_db_file(kwage_group *g, const char *path, uint64_t *first_column,
                            uint32_t *num_filter);

/* The same for `n` files, columns in the order given; first_columns / num_filters (n entries each, may be
 * NULL) receive what kwage_group_add_db_file reports per file.  Prefer this when a group has many files: raw
 * files go file by file through ONE copy-engine pipeline that stays busy across file boundaries (windows of a
 * file are locked in the page cache, copied by SDMA into staging buffers and scattered into the matrix while the
 * next window is being locked: 45-54 GB/s).  Opt-in alternative, KWAGE_LOAD_DIRECT=1: no staging, up to 16 raw
 * files at a time copied side by side by one kernel -- measured slower on a 105 GB matrix of 392 files. */
int kwage_group_add_db_files(kwage_group *g, const char *const *paths, uint32_t n, uint64_t *first_columns,
                             uint32_t *num_filters);

/* Loading progress: while files are loaded through `ctx`, the library adds the bytes of every database file it has
 * passed (whole files, in the order they were given) to *bytes_passed.  The word may lie in memory shared with another
 * process: the CLI's page-cache reader, a child forked before the GPU is touched, keeps a bounded distance ahead of it
 * when the files are not in the page cache.  NULL switches the reporting off (the default). */
void kwage_set_load_progress(kwage_ctx *ctx, volatile uint64_t *bytes_passed);

/* Append `num_columns` synthetic columns: i.i.d. Bernoulli(density_q8/256) bits from a
 * counter-based generator keyed by (seed, row, 64-bit word index) -- generated ON the device. */
int kwage_group_add_random_columns(kwage_group *g, uint64_t num_columns, uint64_t seed,
                                   uint32_t density_q8, uint64_t *first_column);

/* Set individual bits (planting known positives into a synthetic database). */
int kwage_group_set_bits(kwage_group *g, const uint32_t *rows, const uint64_t *columns, uint64_t n);

/* Copy `n` whole rows (row_bytes() bytes each) back to the host, `out_stride` bytes apart. */
int kwage_group_read_rows(kwage_group *g, const uint32_t *rows, uint64_t n, void *out,
                          uint64_t out_stride);

/* No more columns will be added; uploads the valid-column mask. Required before searching. */
int kwage_group_finalize(kwage_group *g);

uint64_t kwage_group_num_columns(const kwage_group *g);   /* valid columns added so far       */
uint64_t kwage_group_column_span(const kwage_group *g);   /* next free global column index    */
uint64_t kwage_group_row_bytes(const kwage_group *g);     /* ceil(column_span/8)              */
uint64_t kwage_group_row_stride(const kwage_group *g);    /* bytes between rows in HBM        */
uint64_t kwage_group_device_bytes(const kwage_group *g);  /* HBM held by the bit matrix       */
int kwage_group_params(const kwage_group *g, kwage_params *out);
/* How the matrix's device block was chosen.  Where a large block lies in HBM decides a few per cent of the gather
 * kernels' rate, so where the device has room for two candidate blocks at once, both are allocated, a few milliseconds
 * of the gather pattern are timed on each and the faster one is kept (matrices of 4 GiB and more; context knobs
 * "group_placement_probe" = 0: no second candidate -- releasing the other block costs seconds per 100 GB while the
 * driver wipes it, so one-shot programs turn it off; "group_contiguous" = 0: plain hipMalloc blocks instead of
 * physically contiguous ones; kwage_ctx_set_tuning, or KWAGE_GROUP_PLACEMENT_PROBE / KWAGE_GROUP_CONTIGUOUS in the
 * environment when the context is created).  candidates = blocks compared (1: no choice was made); *_gbps = the probe's rate on the
 * block kept and on the one released (0 with one candidate); windowed_gbps = the probe on the block kept with all waves
 * reading from the same quarter of it at a time -- where that is more than 3 % faster the block mixes regions of the
 * device's memory and the t = 1 walk kernel takes its rows band after band of the matrix (knob "walk_bands" = -1).
 * Any pointer may be NULL. */
int kwage_group_placement(const kwage_group *g, uint32_t *candidates, double *kept_gbps, double *other_gbps, double *windowed_gbps);

/* ------------------------------------------------------------------------------------
 * Query batch: raw sequences (any case, any characters -- exactly what the reference hands to
 * search(), kwage.cpp:119,137), concatenated, uploaded once and kept resident in HBM.
 * ---------------------------------------------------------------------------------- */
typedef struct kwage_batch kwage_batch;

/* `offsets` has n_queries+1 entries; query i is bytes [offsets[i], offsets[i+1]) of `seqs`. */
int kwage_batch_create(kwage_ctx *ctx, const char *seqs, const uint64_t *offsets,
                       uint32_t n_queries, kwage_batch **out);
void kwage_batch_destroy(kwage_batch *b);
uint32_t kwage_batch_num_queries(const kwage_batch *b);

/* ------------------------------------------------------------------------------------
 * Search: for every query of the batch against every column of the group, what
 * search() computes (kwage.cpp:340-541): canonical k-mer set, MurmurHash3 row indices,
 * row gather + AND (threshold == 1.0f) or per-column counts (threshold < 1), hit extraction.
 * ---------------------------------------------------------------------------------- */
typedef struct {
	uint32_t query;      /* index into the batch                                            */
	uint32_t column;     /* global column of the group                                      */
	uint32_t num_match;  /* MatchResult::num_kmers_found (kwage.cpp:517-518)                */
} kwage_hit;

typedef struct {
	uint64_t n_hits;
	const kwage_hit *hits;          /* sorted by (query, column)                              */
	uint32_t n_queries;
	const uint32_t *num_query_kmer; /* per query: distinct canonical k-mers (kwage.cpp:366)   */
	const uint32_t *query_threshold;/* per query: (unsigned)(float t * n) (kwage.cpp:388); 0 at t==1 */
	uint64_t total_kmers;           /* sum of num_query_kmer                                  */
	uint64_t bit_tests;             /* total_kmers * num_hash * num_columns                   */
	uint64_t algorithmic_bytes;     /* total_kmers * num_hash * ceil(num_columns/8)           */
	float kmer_kernel_ms;           /* with KWAGE_SEARCH_TIMING, else 0                       */
	float search_kernel_ms;         /* the gather + AND / count kernel                        */
	uint32_t search_kernel_launches;/* >1 if the hit buffer had to grow and the kernel re-ran */
	const char *search_kernel;      /* which gather kernel ran, with its template shape: "and_kernel<2,8,nt>",
	                                 * "and_walk_kernel<13,4>", "and_narrow_kernel<4,8>", "count_kernel<10,5>",
	                                 * "count_narrow_kernel<7,1,4>", "...+segments"; "" if none.  Owned by the result. */
} kwage_result;

/* Order `n` hit records by (query, column) in place, on the host (LSD radix sort on the 64-bit key; no device is
 * touched).  What a caller that gathers the unsorted device-resident lists of several shards needs after adding each
 * shard's column base (kwage_amd/distributed.py). */
void kwage_sort_hits(kwage_hit *hits, uint64_t n);

int kwage_search(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags,
                 kwage_result **out);
void kwage_result_free(kwage_result *r);

/* The same search in two halves, for hosts that stream many batches: submit enqueues the whole device
 * pipeline and returns at once; collect waits for it and builds the result.  A context holds at most TWO
 * pending searches (each on its own HIP stream), so the k-mer stage, copy-back and host post-processing
 * of one batch overlap with the gather kernel of the other.  kwage_search == submit + collect. */
typedef struct kwage_pending kwage_pending;
int kwage_search_submit(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags, kwage_pending **out);
int kwage_search_collect(kwage_pending *p, kwage_result **out);   /* consumes p, also on error */
/* 1 when collecting `p` would not wait any more (everything the search queued on the device has finished), 0 while it
 * would, < 0 on a device error.  For hosts that have other work to do meanwhile -- kwage_node feeds the context's slots
 * from inside its RCCL exchange with it.  Does not consume p. */
int kwage_search_poll(kwage_pending *p);

/* Device-side variant for multi-GPU hosts that exchange hit lists themselves (RCCL): hits are
 * left UNSORTED in the caller's device buffer of `capacity` records; *n_hits receives the total
 * found (which may exceed capacity: grow and call again).  num_query_kmer_dev may be NULL or a
 * device buffer of n_queries uint32. */
int kwage_search_device(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags,
                        void *hits_dev, uint64_t capacity, uint64_t *n_hits,
                        void *num_query_kmer_dev);

/* The device-side search in two halves (see kwage_search_submit): the caller alternates between two hit
 * buffers so that the exchange of one step's hits overlaps with the next step's gather kernel.
 * count_dev may be NULL or a device uint64 that receives the total hit count in stream order behind
 * the search kernels -- an exchange buffer can then carry its own record count (header word followed
 * by the records) and go straight into an all-gather without any further device work. */
int kwage_search_device_submit(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags,
                               void *hits_dev, uint64_t capacity, void *count_dev, kwage_pending **out);
/* search_kernel_ms (may be NULL): HIP-event duration of the gather kernel(s) when KWAGE_SEARCH_TIMING was set. */
int kwage_search_device_collect(kwage_pending *p, uint64_t *n_hits, void *num_query_kmer_dev, float *search_kernel_ms);

/* Append mode, for hosts whose result list spans several groups or shards -- the reference appends the matches of every
 * database file to one list per query (kwage.cpp:154-177): *count_dev (a device uint64, required) IS the list's hit
 * counter.  The search appends its records behind the ones already counted there instead of starting at record 0,
 * `column_base` is added to every reported column (each group / shard gets its own range of global column numbers),
 * and reset_count != 0 zeroes the counter in stream order before this search (the first search of a new list).
 * kwage_search_device_collect then reports the RUNNING total; records beyond `capacity` are counted, not stored
 * (grow the buffer and redo the list).  The searches of one context run in submission order. */
int kwage_search_device_append_submit(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags,
                                      void *hits_dev, uint64_t capacity, void *count_dev, uint32_t column_base,
                                      int reset_count, kwage_pending **out);

/* K-mer stage alone (word.h:73-104 + kwage.cpp:362-366 + hash.cpp:176-234 on the device):
 * for query i writes its distinct canonical k-mers to kmers[kmer_offsets[i] ...] (unordered)
 * and their row indices to rows[(kmer_offsets[i]+j)*num_hash + h].  kmer_offsets must hold
 * n_queries+1 entries and is filled with the per-query capacity prefix (max(len-k+1,0));
 * num_query_kmer[i] receives the distinct count.  kmers / rows may be NULL. */
int kwage_hash_batch(kwage_ctx *ctx, const kwage_params *params, kwage_batch *b,
                     uint64_t *kmer_offsets, uint32_t *num_query_kmer,
                     uint64_t *kmers, uint32_t *rows);

/* Streaming-read microbenchmark over the group's bit matrix (achievable HBM peak on this box):
 * reads `bytes` (clamped to the matrix size) `iters` times; returns GB/s. */
int kwage_stream_read_gbps(kwage_group *g, uint64_t bytes, uint32_t iters, double *gbps);

/* ------------------------------------------------------------------------------------
 * Database construction (the format's only writer): replaces build_db() (build_db.cpp:24-456;
 * declared maestro.h:121) -- `.bloom` files in, one `.db` file out, byte-identical to the
 * reference's output; the bit transpose (build_db.cpp:259-304) runs on the device.
 * ---------------------------------------------------------------------------------- */
typedef struct {
	uint64_t bits_transposed;      /* 2^L * n                               */
	float transpose_kernel_ms;     /* sum over chunks, HIP events           */
	uint64_t db_bytes;             /* size of the file written              */
} kwage_build_stats;

int kwage_build_db(kwage_ctx *ctx, const char *out_path, const kwage_params *params,
                   const char *const *bloom_paths, uint32_t n, kwage_build_stats *stats);

/* ------------------------------------------------------------------------------------
 * Bloom filter construction from sequences -- the EXACT k-mer set (genomes / assemblies): every valid
 * canonical k-mer sets bit hash_h & (2^L - 1) for h < num_hash.  (The reference's make_bloom_filter()
 * always runs its counting-Bloom pass, which even at min_kmer_count == 1 may skip a k-mer whose counters
 * were raised by others; that behaviour, for any min_kmer_count, is kwage_bloom_counter_* below.)
 * Output: a `.bloom` file as binary_write<BloomFilter> writes it
 * (binary_io.cpp:182-208), ready for kwage_build_db.
 * ---------------------------------------------------------------------------------- */
typedef struct {
	const char *run_accession;                /* required: 3 letters + 1..10 digits (sra_accession.cpp:27-64) */
	const char *experiment_accession;         /* accessions and texts may be NULL or "" = absent */
	const char *sample_accession;
	const char *study_accession;
	const char *experiment_title;
	const char *experiment_design_description;
	const char *experiment_library_name;
	const char *experiment_library_strategy;
	const char *experiment_library_source;
	const char *experiment_library_selection;
	const char *experiment_instrument_model;
	const char *sample_taxa;
	const char *study_title;
	const char *study_abstract;
	const char *const *attribute_tags;        /* num_attributes entries each */
	const char *const *attribute_values;
	uint32_t num_attributes;
	uint64_t number_of_spots, number_of_bases;
	uint32_t day, month, year;                /* date_received; 0,0,0 = absent */
} kwage_sample_info;

/* optimal_bloom_param (bloom.cpp:10-68): smallest log_2_filter_len in [min,max], then the num_hash in
 * [1,5] with the lowest false-positive probability <= p for `num_kmer` distinct k-mers. */
int kwage_optimal_bloom_param(uint32_t kmer_len, uint64_t num_kmer, float p, uint32_t min_log_2_filter_len,
                              uint32_t max_log_2_filter_len, kwage_params *out);

/* Distinct canonical k-mers over ALL sequences of the batch (one shared set). */
int kwage_count_distinct_kmers(kwage_ctx *ctx, kwage_batch *b, uint32_t kmer_len, uint64_t *count);

/* Bits of the sample's Bloom filter (2^L / 8 bytes, LSB first) computed on the device. */
int kwage_bloom_bits_from_batch(kwage_ctx *ctx, const kwage_params *params, kwage_batch *b,
                                void *bits_out, uint64_t *distinct);

/* Sequences (concatenated + offsets, like kwage_batch_create) -> `.bloom` file. Long sequences are cut
 * into overlapping pieces internally so that every k-mer is seen once by some workgroup. */
int kwage_make_bloom(kwage_ctx *ctx, const kwage_params *params, const char *seqs, const uint64_t *offsets,
                     uint32_t n_seqs, const kwage_sample_info *info, const char *out_path,
                     uint64_t *num_distinct_kmers);

/* ------------------------------------------------------------------------------------
 * Bloom filter construction WITH a minimum k-mer count -- make_bloom_filter()'s counting pass
 * (make_bloom.cpp:76-504, count_words :506-621) with the reference's order-dependent semantics: two
 * 4-bit counting Bloom filters of 2^C elements with conservative update over the k-mer occurrences
 * of the read stream IN ORDER; the occurrence that lifts a k-mer's minimum counter to
 * min_kmer_count sets its five candidate bits in 2^M-bit vectors and counts as one k-mer; at the
 * end optimal_bloom_param(num_kmer) picks (log_2_filter_len, num_hash) and the first num_hash
 * vectors are OR-folded to that length (:336-354).  The device schedule commits occurrences that
 * share a counter in their original order, so counters, bits and num_kmer equal the sequential
 * loop's (kwage_amd/csrc/counter.hip).  Fragments are the sequences handed to _add, in call order
 * (the reference's front end -- NCBI SDK read iterators -- is replaced by whatever reads the caller
 * has, e.g. kwage_seqfile_*).  Note that even min_kmer_count == 1 differs from the exact k-mer set
 * of kwage_make_bloom: a new k-mer whose four counters were already raised by others is skipped.
 * PARITY UNPINNED against a running reference (make_bloom.cpp needs the NCBI SDK): checked against
 * the line-by-line restatement in oracle/.
 * ---------------------------------------------------------------------------------- */
typedef struct kwage_bloom_counter kwage_bloom_counter;

typedef struct {
	uint64_t num_valid_kmer;          /* m_progress.num_kmer                                      */
	uint64_t num_bp;                  /* m_progress.num_bp: bases of all fragments                */
	uint64_t positions;               /* k-mer start positions examined                           */
	uint64_t occurrences_committed;   /* occurrences that changed a counter                       */
	uint64_t chunks, rounds;          /* device chunks processed, commit rounds over all chunks   */
	uint32_t max_rounds;              /* longest commit chain in one chunk                        */
	uint32_t reserved;
	double add_ms;                    /* host wall time inside _add/_flush                        */
} kwage_bloom_counter_stats;

#define KWAGE_BLOOM_SUCCESS 0         /* STATUS_BLOOM_SUCCESS, maestro.h:25 */
#define KWAGE_BLOOM_INVALID 1         /* STATUS_BLOOM_INVALID, maestro.h:27: bound not satisfiable / no k-mers */

/* make_bloom.cpp:105-130: log2 length of the counting filters from the number of bases (0 = unknown -> 32). */
uint32_t kwage_counting_filter_log2(uint64_t num_bp);
/* bloom.cpp:72-121 approximate_max_kmers */
uint64_t kwage_approximate_max_kmers(float false_positive_probability, uint32_t min_log_2_filter_len,
                                     uint32_t max_log_2_filter_len);

int kwage_bloom_counter_create(kwage_ctx *ctx, uint32_t kmer_len, int32_t hash_func, uint32_t min_kmer_count,
                               uint32_t log_2_counting_filter_len, uint32_t max_log_2_filter_len,
                               kwage_bloom_counter **out);
void kwage_bloom_counter_destroy(kwage_bloom_counter *bc);
/* Next sample in the same object (device allocations kept; a 2^32-element object costs ~20 GB to create):
 * counters, candidate bits and totals are zeroed; the counting filters may be smaller than at creation. */
int kwage_bloom_counter_reset(kwage_bloom_counter *bc, uint32_t min_kmer_count, uint32_t log_2_counting_filter_len);
/* Append fragments (concatenated + offsets, like kwage_batch_create) to the read stream. */
int kwage_bloom_counter_add(kwage_bloom_counter *bc, const char *seqs, const uint64_t *offsets, uint32_t n_seqs);
/* Process what is staged on the host so far (add() works in 16 M-position chunks). */
int kwage_bloom_counter_flush(kwage_bloom_counter *bc);
int kwage_bloom_counter_get_stats(kwage_bloom_counter *bc, kwage_bloom_counter_stats *out);
/* Inspection (tests): CountingBloom elements [first, first+n) as bytes (low nibble `first`, high nibble
 * `second`, make_bloom.cpp:59-66) and bytes of candidate bit vector `hash` (LSB first). */
int kwage_bloom_counter_read_counts(kwage_bloom_counter *bc, uint64_t first, uint64_t n, void *out);
int kwage_bloom_counter_read_valid_bits(kwage_bloom_counter *bc, uint32_t hash, uint64_t first_byte, uint64_t nbytes, void *out);
/* make_bloom.cpp:210-216,309-450: *status = KWAGE_BLOOM_INVALID (nothing written) if num_kmer exceeds
 * approximate_max_kmers or no parameters satisfy the bound; else fold, CRC and write the `.bloom` file
 * (out_path may be NULL: parameters only).  `chosen` may be NULL. */
int kwage_bloom_counter_finish(kwage_bloom_counter *bc, float false_positive_probability, uint32_t min_log_2_filter_len,
                               const kwage_sample_info *info, const char *out_path, kwage_params *chosen, int *status);

/* Column-wise re-pack: the columns of several same-parameter `.db` files (raw or compressed) become
 * ONE raw `.db` file with contiguous columns, in file order then column order -- the bit-level work of
 * the reference's merge_db.cpp:268-820 (get_bit/set_bit per bit there; shift-and-OR on the device
 * here), without its file naming / size policy.  The reference `kwage` reads the result. */
int kwage_repack_db(kwage_ctx *ctx, const char *out_path, const char *const *in_paths, uint32_t n);

/* ------------------------------------------------------------------------------------
 * Host-side helpers that mirror the reference's host code for this path (no device needed).
 * ---------------------------------------------------------------------------------- */

/* DBFileHeader, kwage.h:30-72, as serialized by binary_io.cpp:243-265 (44 bytes, LE). */
typedef struct {
	uint32_t magic, version, crc32, kmer_len, num_hash, log_2_filter_len, num_filter;
	int32_t  hash_func;
	uint32_t compression;
	uint64_t info_start;
} kwage_db_header;

int kwage_db_read_header(const char *path, kwage_db_header *out);

/* The listed slices of one .db file (host only; raw or compressed), n * ceil(num_filter / 8) bytes in list order:
 * what the reference's seekg + read per addressed slice fetches (kwage.cpp:414-416), and what the sparse groups
 * (kwage_group_create_sparse) are loaded through -- several threads for long lists. */
int kwage_db_read_slices(const char *path, const uint32_t *rows, uint64_t n, unsigned char *out);

/* Compressed container (host only).  The reference ships a slice CODEC (slice_z.h: raw deflate,
 * windowBits -9, level 9, memLevel 9, default strategy, "store raw unless smaller") but no file
 * layout, and its kwage ignores header.compression.  This repo defines the layout:
 * header.compression = 2, u64 offset[2^L+1], per-slice payloads, then the usual metadata.
 * kwage_group_add_db_file reads both layouts (slices are inflated once, on the host, at load).
 * Parity with the reference is UNPINNED for this container (it has none); it is validated by
 * round trip: decompress(compress(x)) == x byte for byte, and identical search results. */
int kwage_db_compress(const char *in_path, const char *out_path, uint32_t threads);
int kwage_db_decompress(const char *in_path, const char *out_path);

/* Database metadata reader: info_loc[] + FilterInfo records (kwage.cpp:505-515,
 * binary_io.cpp:154-176), loaded once per file instead of two seeks per hit. */
typedef struct kwage_dbinfo kwage_dbinfo;
int kwage_dbinfo_open(const char *path, kwage_dbinfo **out);
void kwage_dbinfo_close(kwage_dbinfo *d);
uint32_t kwage_dbinfo_num_filter(const kwage_dbinfo *d);
/* FilterInfo::csv_string() (bloom.cpp:124-127): the run accession. */
int kwage_dbinfo_csv_string(const kwage_dbinfo *d, uint32_t column, char *buf, size_t buflen);
/* FilterInfo::json_string(prefix) (bloom.cpp:129-326). Returns needed length (excluding NUL). */
int64_t kwage_dbinfo_json_string(const kwage_dbinfo *d, uint32_t column, const char *prefix,
                                 char *buf, size_t buflen);

/* sra_accession.cpp:27-96 */
int kwage_str_to_accession(const char *s, uint64_t *out);
int kwage_accession_to_str(uint64_t acc, char *buf, size_t buflen);

/* SequenceIterator (parse_sequence.cpp:13-262): FASTA / FASTQ, gz transparent. */
typedef struct kwage_seqfile kwage_seqfile;
int kwage_seqfile_open(const char *path, kwage_seqfile **out);
/* Returns 1 and sets the views (valid until the next call) or 0 at end of file, <0 on error. */
int kwage_seqfile_next(kwage_seqfile *f, const char **defline, const char **seq, uint64_t *seq_len);
void kwage_seqfile_close(kwage_seqfile *f);

/* (unsigned)(float threshold * n), kwage.cpp:388 */
uint32_t kwage_query_threshold(float threshold, uint32_t num_query_kmer);

#ifdef __cplusplus
}
#endif

#endif /* KWAGE_AMD_H */
