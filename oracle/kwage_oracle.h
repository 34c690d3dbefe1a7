/* oracle/kwage_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11, scalar) of the reference `kwage` search path
 * (LANL-Bioinformatics/KWAGE, /root/reference).  It exists so that the HIP product under
 * kwage_amd/ can be checked bit-for-bit; it is NEVER linked into, imported by or called
 * from the product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it.
 *
 * PARITY PINNING: the reference has no golden vectors of its own, so
 * this restatement is pinned against outputs of the REFERENCE ITSELF run in the build
 * container: oracle/_ref/kwage and oracle/_ref/ref_tool are compiled from the reference's
 * own sources (oracle/Makefile `ref`), and tests/golden/ holds the fixtures they produced
 * (generator: tests/golden/make_golden.py).  tests/test_oracle_vs_reference.py asserts this
 * file == those fixtures.
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef KWAGE_ORACLE_H
#define KWAGE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* word.cpp:9-23 kmer_word_mask: low 2*k bits set (k may be 32). */
uint64_t kwo_kmer_mask(uint32_t k);

/* word.h:73-104 ForEachDuplexWord + :161-165 ValidWord/CanonicalWord.
 * Writes, for every position whose last k characters are all in [ACGTacgt], the canonical
 * word (min of sense and reverse complement, masked) to out_words[] and the 0-based start
 * of the k-mer (Loc5) to out_pos[] (either may be NULL).  Returns the number written;
 * never more than max(len-k+1,0). */
size_t kwo_canonical_kmers(const char *seq, size_t len, uint32_t k,
                           uint64_t *out_words, uint64_t *out_pos);

/* kwage.cpp:352-366: SORT + unique of the canonical words.  out must hold len entries.
 * Returns num_query_kmer. */
size_t kwo_unique_kmers(const char *seq, size_t len, uint32_t k, uint64_t *out);

/* hash.cpp:114-170 murmur_hash32(string, seed): canonical MurmurHash3_x86_32. */
uint32_t kwo_murmur3_32_bytes(const uint8_t *data, uint32_t len, uint32_t seed);

/* hash.cpp:176-234 murmur_hash32(Word, k, seed): the same hash over the k ASCII bases
 * "ACGT"[2-bit code], 5'->3' (most significant base first, hash.cpp:189-190). */
uint32_t kwo_murmur3_32_word(uint64_t word, uint32_t k, uint32_t seed);

/* hash.cpp:79-94 bigsi_hash(Word,k,seed,func): hash_func 0 == MURMUR_HASH_32 (hash.h:8-13);
 * any other value is the reference's "Unknown hash function" throw -> returns 0 and sets
 * *err to 1 (err may be NULL). */
uint32_t kwo_bigsi_hash(uint64_t word, uint32_t k, uint32_t seed, int32_t hash_func, int *err);

/* kwage.cpp:388: query_threshold = (unsigned)( float(threshold) * num_query_kmer ),
 * float32 multiply, truncation toward zero. */
uint32_t kwo_query_threshold(float threshold, uint32_t num_query_kmer);

/* kwage.cpp:397: number of leading k-mers searched without the early-exit test. */
uint64_t kwo_mid_kmers(float threshold, uint32_t num_query_kmer);

/* One hit of kwage.cpp:489-538: column index within the file and num_match. */
typedef struct {
	uint32_t column;
	uint32_t num_match;
} kwo_hit;

/* kwage.cpp:373-538 restated over an in-memory image of the bit-slice block
 * (file bytes [bloom_start, info_start) of a NO_COMPRESSION .db, i.e. 2^L rows of
 * slice_size = ceil(num_filter/8) bytes, LSB-first within each byte: bloom.h:143,162).
 *
 *   rows          pointer to row 0
 *   row_stride    bytes between rows (== slice_size for a file image)
 *   kmers/n_kmer  sorted unique canonical words (kwo_unique_kmers)
 *   early_exit    1 = follow kwage.cpp:437-483 (stop as the reference does),
 *                 0 = read every row (results are identical; only rows_read differs)
 *   hits/cap      output, ascending column order; returns the total number of hits even if
 *                 it exceeds cap (only the first cap are stored)
 *   rows_read     (may be NULL) number of slice reads performed
 */
size_t kwo_search_image(const uint8_t *rows, size_t row_stride,
                        uint32_t kmer_len, uint32_t num_hash, uint32_t log_2_filter_len,
                        uint32_t num_filter, int32_t hash_func,
                        const uint64_t *kmers, uint32_t n_kmer,
                        float threshold, int early_exit,
                        kwo_hit *hits, size_t cap, uint64_t *rows_read);

/* Same reduction, but the caller supplies one pointer per addressed row in the reference's
 * visiting order [kmer 0 hash 0, kmer 0 hash 1, ..., kmer 1 hash 0, ...] (used when only
 * the addressed rows of a device-resident database were copied back). */
size_t kwo_search_rows(const uint8_t *const *row_ptrs,
                       uint32_t num_hash, uint32_t num_filter, uint32_t n_kmer,
                       float threshold, int early_exit,
                       kwo_hit *hits, size_t cap, uint64_t *rows_read);

/* Convenience: whole search() of kwage.cpp:340-541 for one query string against one image. */
size_t kwo_search_sequence(const uint8_t *rows, size_t row_stride,
                           uint32_t kmer_len, uint32_t num_hash, uint32_t log_2_filter_len,
                           uint32_t num_filter, int32_t hash_func,
                           const char *seq, size_t len,
                           float threshold, int early_exit,
                           kwo_hit *hits, size_t cap,
                           uint32_t *num_query_kmer, uint64_t *rows_read);

/* ---- Bloom filter construction with a minimum k-mer count (make_bloom.cpp) --------------------------
 * PARITY UNPINNED for this block: make_bloom.cpp needs the NCBI SDK headers and cannot be compiled
 * here, and the reference ships no vectors for it.  The functions below restate its source line by
 * line (sequential, one fragment after the other); the pieces it shares with the pinned path
 * (ForEachDuplexWord, bigsi_hash, optimal_bloom_param, binary_write<BloomFilter>) ARE pinned. */

/* make_bloom.cpp:105-130: log2 length of the two 4-bit counting Bloom filters from the number of
 * bases (0 = unknown -> the maximum, 32); clamped to [18, 32]. */
uint32_t kwo_counting_filter_log2(uint64_t num_bp);

/* bloom.cpp:72-121 approximate_max_kmers: the smallest power of two of k-mers for which no
 * (filter length, num_hash) in range meets the false-positive bound. */
uint64_t kwo_approximate_max_kmers(float p, uint32_t min_log_2_filter_len, uint32_t max_log_2_filter_len);

/* State of make_bloom_filter() between fragments: CountingBloom[2^log2_count] (one byte each:
 * low nibble `first`, high nibble `second`, make_bloom.cpp:59-66), MAX_NUM_HASH = 5 bit vectors of
 * 2^max_log2 bits (`valid_bits`, :165-169) and the running num_kmer. */
typedef struct {
	uint32_t kmer_len, min_kmer_count, log2_count, max_log2;
	uint8_t *count;
	uint8_t *valid[5];
	uint64_t num_valid_kmer;
	uint64_t num_bp;
} kwo_counter;

kwo_counter *kwo_counter_new(uint32_t kmer_len, uint32_t min_kmer_count, uint32_t log2_count, uint32_t max_log2);
void kwo_counter_free(kwo_counter *c);

/* count_words (make_bloom.cpp:506-621) for one fragment, preceded by the num_bp bookkeeping of the
 * calling loops (:202,:238,:290). */
void kwo_counter_add(kwo_counter *c, const char *seq, size_t len);

/* make_bloom.cpp:336-354: OR-fold of valid_bits[h], h < num_hash, into a filter of 2^log_2_filter_len
 * bits (out: 2^log_2_filter_len / 8 bytes, log_2_filter_len >= 3). */
void kwo_counter_fold(const kwo_counter *c, uint32_t log_2_filter_len, uint32_t num_hash, uint8_t *out);

#ifdef __cplusplus
}
#endif

#endif /* KWAGE_ORACLE_H */
