/* oracle/kwage_oracle.c -- TEST INFRASTRUCTURE ONLY (see kwage_oracle.h).
 *
 * Plain scalar C restatement of the reference `kwage` search path.  Never part of the
 * product; the product (kwage_amd/) fails loudly without its HIP library instead of
 * falling back to this file.
 */
#include "kwage_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- word.cpp:9-23 ------------------------------------------------------------------ */
uint64_t kwo_kmer_mask(uint32_t k)
{
	uint64_t ret = 0;
	for(uint32_t i = 0; i < 2*k && i < 64; ++i){
		ret |= (UINT64_C(1) << i);
	}
	return ret;
}

/* ---- word.h:73-104, 161-165 ----------------------------------------------------------
 * The rolling words are never cleared on an invalid character; only the run length is
 * reset (word.h:101-103), and a word is valid once k consecutive good bases were seen
 * (word.h:162).  By then every stale bit has been shifted out of the masked window. */
size_t kwo_canonical_kmers(const char *seq, size_t len, uint32_t k,
                           uint64_t *out_words, uint64_t *out_pos)
{
	const uint64_t comp_shift = 2*( (uint64_t)k - 1 );
	const uint64_t mask = kwo_kmer_mask(k);
	uint64_t w = 0, comp_w = 0;
	unsigned int word_len = 0;
	size_t n = 0;

	for(size_t index = 0; index < len; ++index){

		++word_len;

		switch(seq[index]){
			case 'A': case 'a':
				w = (w << 2) | 0;
				comp_w = (comp_w >> 2) | (UINT64_C(3) << comp_shift);
				break;
			case 'T': case 't':
				w = (w << 2) | 3;
				comp_w = (comp_w >> 2) | (UINT64_C(0) << comp_shift);
				break;
			case 'G': case 'g':
				w = (w << 2) | 2;
				comp_w = (comp_w >> 2) | (UINT64_C(1) << comp_shift);
				break;
			case 'C': case 'c':
				w = (w << 2) | 1;
				comp_w = (comp_w >> 2) | (UINT64_C(2) << comp_shift);
				break;
			default:
				word_len = 0;
				break;
		}

		if(word_len >= k){
			const uint64_t s = w & mask;
			const uint64_t a = comp_w & mask;
			if(out_words){ out_words[n] = (s < a) ? s : a; }
			if(out_pos){ out_pos[n] = (index + 1) - k; }
			++n;
		}
	}
	return n;
}

static int cmp_u64(const void *a, const void *b)
{
	const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
	return (x > y) - (x < y);
}

/* ---- kwage.cpp:352-366 -------------------------------------------------------------- */
size_t kwo_unique_kmers(const char *seq, size_t len, uint32_t k, uint64_t *out)
{
	size_t n = kwo_canonical_kmers(seq, len, k, out, NULL);
	if(n == 0){ return 0; }
	qsort(out, n, sizeof(uint64_t), cmp_u64);
	size_t m = 1;
	for(size_t i = 1; i < n; ++i){
		if(out[i] != out[m - 1]){ out[m++] = out[i]; }
	}
	return m;
}

/* ---- hash.cpp:43-57 ----------------------------------------------------------------- */
static inline uint32_t rotl32(uint32_t x, int r){ return (x << r) | (x >> (32 - r)); }

static inline uint32_t fmix32(uint32_t h)
{
	h ^= h >> 16;
	h *= 0x85ebca6bU;
	h ^= h >> 13;
	h *= 0xc2b2ae35U;
	h ^= h >> 16;
	return h;
}

/* ---- hash.cpp:114-170 --------------------------------------------------------------- */
uint32_t kwo_murmur3_32_bytes(const uint8_t *data, uint32_t len, uint32_t seed)
{
	const uint32_t c1 = 0xcc9e2d51U, c2 = 0x1b873593U;
	const uint32_t nblocks = len/4;
	uint32_t h1 = seed;
	uint32_t offset = 0;

	for(uint32_t i = 0; i < nblocks; ++i, offset += 4){
		uint32_t k1 = ((uint32_t)data[offset + 3] << 24) | ((uint32_t)data[offset + 2] << 16) |
		              ((uint32_t)data[offset + 1] << 8)  |  (uint32_t)data[offset + 0];
		k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
		h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1*5 + 0xe6546b64U;
	}

	uint32_t k1 = 0;
	switch(len & 3){
		case 3: k1 ^= (uint32_t)data[offset + 2] << 16; /* fall through */
		case 2: k1 ^= (uint32_t)data[offset + 1] << 8;  /* fall through */
		case 1: k1 ^= (uint32_t)data[offset];
			k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
	}

	h1 ^= len;
	return fmix32(h1);
}

/* ---- hash.cpp:176-234 (BITS_TO_BASE :189-190; bits_to_base word.h:31-34) ------------- */
uint32_t kwo_murmur3_32_word(uint64_t word, uint32_t k, uint32_t seed)
{
	uint8_t ascii[32];
	for(uint32_t i = 0; i < k && i < 32; ++i){
		ascii[i] = (uint8_t)"ACGT"[ (word >> (2*(k - 1 - i))) & 3 ];
	}
	return kwo_murmur3_32_bytes(ascii, k, seed);
}

/* ---- hash.cpp:79-94 ----------------------------------------------------------------- */
uint32_t kwo_bigsi_hash(uint64_t word, uint32_t k, uint32_t seed, int32_t hash_func, int *err)
{
	if(hash_func == 0){ /* MURMUR_HASH_32, hash.h:9 */
		if(err){ *err = 0; }
		return kwo_murmur3_32_word(word, k, seed);
	}
	if(err){ *err = 1; }
	return 0;
}

/* ---- kwage.cpp:388, :397 ------------------------------------------------------------ */
uint32_t kwo_query_threshold(float threshold, uint32_t num_query_kmer)
{
	/* `unsigned int = float*unsigned int`: the count is converted to float, the product is a
	 * float32, and the result truncates toward zero. */
	volatile float prod = threshold * (float)num_query_kmer;
	return (uint32_t)prod;
}

uint64_t kwo_mid_kmers(float threshold, uint32_t num_query_kmer)
{
	volatile float prod = (1.0f - threshold) * (float)num_query_kmer;
	return (uint64_t)(int64_t)prod;
}

/* ---- kwage.cpp:373-538 -------------------------------------------------------------- */
static int any_bit(const uint8_t *v, uint32_t num_filter)
{
	/* bloom.h:333-360 max_bit(): pad bits of the last block do not count. */
	const size_t nblock = (num_filter + 7)/8;
	for(size_t i = 0; i + 1 < nblock; ++i){
		if(v[i]){ return 1; }
	}
	const unsigned rem = num_filter % 8;
	const uint8_t last = v[nblock - 1];
	if(rem == 0){ return last != 0; }
	return (last & ((1u << rem) - 1)) != 0;
}

size_t kwo_search_rows(const uint8_t *const *row_ptrs,
                       uint32_t num_hash, uint32_t num_filter, uint32_t n_kmer,
                       float threshold, int early_exit,
                       kwo_hit *hits, size_t cap, uint64_t *rows_read)
{
	if(rows_read){ *rows_read = 0; }
	if(n_kmer == 0 || num_filter == 0){ return 0; } /* kwage.cpp:369-371 */

	const int complete_match = (threshold == 1.0f);           /* kwage.cpp:349 */
	const size_t nblock = ((size_t)num_filter + 7)/8;         /* kwage.cpp:108-109 */

	uint8_t *mask = NULL;
	uint32_t *count = NULL;
	uint32_t query_threshold = 0;

	if(complete_match){
		mask = (uint8_t*)malloc(nblock);
		memset(mask, 0xFF, nblock);                           /* bloom.h:182-187 */
	}
	else{
		count = (uint32_t*)calloc(num_filter, sizeof(uint32_t));
		query_threshold = kwo_query_threshold(threshold, n_kmer);
	}

	uint8_t *kmer_match = (uint8_t*)malloc(nblock);
	const uint64_t mid = early_exit ? kwo_mid_kmers(threshold, n_kmer) : (uint64_t)n_kmer;
	uint64_t nread = 0;

	for(uint64_t i = 0; i < n_kmer; ++i){

		memset(kmer_match, 0xFF, nblock);

		for(uint32_t h = 0; h < num_hash; ++h){               /* kwage.cpp:409-423 */
			const uint8_t *slice = row_ptrs[i*num_hash + h];
			++nread;
			for(size_t b = 0; b < nblock; ++b){               /* bloom.h:245-262 */
				kmer_match[b] &= slice[b];
			}
		}

		if(complete_match){
			for(size_t b = 0; b < nblock; ++b){ mask[b] &= kmer_match[b]; }
			if(i >= mid && !any_bit(mask, num_filter)){       /* kwage.cpp:466-470 */
				break;
			}
		}
		else{
			for(uint32_t j = 0; j < num_filter; ++j){         /* bloom.h:291-330 */
				count[j] += (kmer_match[j/8] >> (j%8)) & 1;
			}
			if(i >= mid){                                     /* kwage.cpp:478-481 */
				uint32_t mx = 0;
				for(uint32_t j = 0; j < num_filter; ++j){ if(count[j] > mx){ mx = count[j]; } }
				if( (uint64_t)mx + (uint64_t)(n_kmer - i) < (uint64_t)query_threshold ){
					break;
				}
			}
		}
	}

	size_t nhit = 0;
	for(uint32_t j = 0; j < num_filter; ++j){                 /* kwage.cpp:489-538 */
		int matched;
		if(complete_match){ matched = (mask[j/8] >> (j%8)) & 1; }
		else{ matched = (count[j] >= query_threshold); }
		if(matched){
			if(nhit < cap && hits){
				hits[nhit].column = j;
				hits[nhit].num_match = complete_match ? n_kmer : count[j];
			}
			++nhit;
		}
	}

	free(kmer_match);
	free(mask);
	free(count);
	if(rows_read){ *rows_read = nread; }
	return nhit;
}

size_t kwo_search_image(const uint8_t *rows, size_t row_stride,
                        uint32_t kmer_len, uint32_t num_hash, uint32_t log_2_filter_len,
                        uint32_t num_filter, int32_t hash_func,
                        const uint64_t *kmers, uint32_t n_kmer,
                        float threshold, int early_exit,
                        kwo_hit *hits, size_t cap, uint64_t *rows_read)
{
	if(rows_read){ *rows_read = 0; }
	if(n_kmer == 0){ return 0; }

	const uint64_t filter_len = UINT64_C(1) << log_2_filter_len;  /* kwage.h:63-66 */
	const size_t n = (size_t)n_kmer*num_hash;
	const uint8_t **ptrs = (const uint8_t**)malloc(n*sizeof(*ptrs));

	for(uint32_t i = 0; i < n_kmer; ++i){
		for(uint32_t h = 0; h < num_hash; ++h){
			int err = 0;
			/* kwage.cpp:411-414: slice_index = bigsi_hash(...) % filter_len */
			const uint64_t idx = (uint64_t)kwo_bigsi_hash(kmers[i], kmer_len, h, hash_func, &err) % filter_len;
			if(err){ free(ptrs); return (size_t)-1; }
			ptrs[(size_t)i*num_hash + h] = rows + idx*row_stride;
		}
	}

	const size_t ret = kwo_search_rows(ptrs, num_hash, num_filter, n_kmer, threshold,
	                                   early_exit, hits, cap, rows_read);
	free(ptrs);
	return ret;
}

size_t kwo_search_sequence(const uint8_t *rows, size_t row_stride,
                           uint32_t kmer_len, uint32_t num_hash, uint32_t log_2_filter_len,
                           uint32_t num_filter, int32_t hash_func,
                           const char *seq, size_t len,
                           float threshold, int early_exit,
                           kwo_hit *hits, size_t cap,
                           uint32_t *num_query_kmer, uint64_t *rows_read)
{
	uint64_t *kmers = (uint64_t*)malloc((len ? len : 1)*sizeof(uint64_t));
	const uint32_t n = (uint32_t)kwo_unique_kmers(seq, len, kmer_len, kmers);
	if(num_query_kmer){ *num_query_kmer = n; }
	const size_t ret = kwo_search_image(rows, row_stride, kmer_len, num_hash, log_2_filter_len,
	                                    num_filter, hash_func, kmers, n, threshold, early_exit,
	                                    hits, cap, rows_read);
	free(kmers);
	return ret;
}

/* ======================================================================================
 * Bloom filter construction with a minimum k-mer count -- make_bloom.cpp (PARITY UNPINNED,
 * see kwage_oracle.h).
 * ==================================================================================== */

/* make_bloom.cpp:105-130 (MAX_LOG_COUNT_FILTER_LEN 32, MIN 18, COUNT_FILTER_FP 1e-2 :20-24) */
uint32_t kwo_counting_filter_log2(uint64_t num_bp)
{
	uint64_t lg = 32;
	if(num_bp > 0){
		const double counting_length = 1.0/( 1.0 - pow( 1.0 - pow(1.0e-2, 1.0/4.0), 1.0/(2*num_bp) ) );
		/* the reference assigns the double result of ceil() to a size_t member */
		lg = (uint64_t)ceil( log(counting_length)/log(2.0) );
		if(lg > 32){ lg = 32; }
		if(lg < 18){ lg = 18; }
	}
	return (uint32_t)lg;
}

/* bloom.cpp:72-121 */
uint64_t kwo_approximate_max_kmers(float m_p, uint32_t min_lg, uint32_t max_lg)
{
	for(uint64_t log_2_num_kmer = 1; log_2_num_kmer < 64; ++log_2_num_kmer){
		const uint64_t num_kmer = UINT64_C(1) << log_2_num_kmer;
		int valid = 0;
		for(uint64_t lg = min_lg; (lg <= max_lg) && !valid; ++lg){
			float best_p = 10.0f;
			for(uint32_t num_hash = 1; (num_hash <= 5) && !valid; ++num_hash){
				const uint64_t len = UINT64_C(1) << lg;
				const double p = pow(1.0 - pow(1.0 - 1.0/len, num_kmer*num_hash), num_hash);
				if( (p <= m_p) && (p < best_p) ){ valid = 1; }
			}
		}
		if(!valid){ return num_kmer; }
	}
	return UINT64_C(0xFFFFFFFFFFFFFFFF);
}

kwo_counter *kwo_counter_new(uint32_t kmer_len, uint32_t min_kmer_count, uint32_t log2_count, uint32_t max_log2)
{
	kwo_counter *c = (kwo_counter*)calloc(1, sizeof(kwo_counter));
	if(!c){ return NULL; }
	c->kmer_len = kmer_len; c->min_kmer_count = min_kmer_count;
	c->log2_count = log2_count; c->max_log2 = max_log2;
	c->count = (uint8_t*)calloc((size_t)1 << log2_count, 1);                 /* :159 memset 0 */
	size_t vb = (((size_t)1 << max_log2) + 7)/8;
	for(int h = 0; h < 5; ++h){ c->valid[h] = (uint8_t*)calloc(vb, 1); }       /* :165-169 */
	return c;
}

void kwo_counter_free(kwo_counter *c)
{
	if(!c){ return; }
	free(c->count);
	for(int h = 0; h < 5; ++h){ free(c->valid[h]); }
	free(c);
}

/* make_bloom.cpp:506-621 */
void kwo_counter_add(kwo_counter *c, const char *seq, size_t len)
{
	c->num_bp += len;
	if(len == 0){ return; }
	uint64_t *words = (uint64_t*)malloc(len*sizeof(uint64_t));
	const size_t n = kwo_canonical_kmers(seq, len, c->kmer_len, words, NULL);
	const uint64_t count_mask = (UINT64_C(1) << c->log2_count) - 1;     /* :138-148 */
	const uint64_t seq_mask = (UINT64_C(1) << c->max_log2) - 1;
	const uint32_t m = c->min_kmer_count;
	for(size_t i = 0; i < n; ++i){
		uint64_t hi[5];
		for(uint32_t h = 0; h < 5; ++h){ hi[h] = kwo_murmur3_32_word(words[i], c->kmer_len, h); }   /* :530 */
		uint8_t *c0 = &c->count[hi[0] & count_mask], *c1 = &c->count[hi[1] & count_mask];
		uint8_t *c2 = &c->count[hi[2] & count_mask], *c3 = &c->count[hi[3] & count_mask];
		const unsigned f0 = *c0 & 15u, f1 = *c1 & 15u;                    /* .first  :546-547 */
		const unsigned s0 = *c2 >> 4, s1 = *c3 >> 4;                       /* .second :549-550 */
		unsigned mn = f0;
		if(f1 < mn){ mn = f1; }
		if(s0 < mn){ mn = s0; }
		if(s1 < mn){ mn = s1; }
		if(mn < m){                                                        /* :560 */
			if(mn == m - 1){                                               /* :562 */
				++c->num_valid_kmer;
				for(uint32_t h = 0; h < 5; ++h){
					const uint64_t b = hi[h] & seq_mask;
					c->valid[h][b >> 3] |= (uint8_t)(1u << (b & 7));       /* BitVector::set_bit, bloom.h:143 */
				}
			}
			/* :587-602: four separate ++ on 4-bit fields (two hashes landing on the same element
			 * increment it twice; a 4-bit field wraps) */
			if(f0 == mn){ *c0 = (uint8_t)((*c0 & 0xF0u) | ((*c0 + 1u) & 15u)); }
			if(f1 == mn){ *c1 = (uint8_t)((*c1 & 0xF0u) | ((*c1 + 1u) & 15u)); }
			if(s0 == mn){ *c2 = (uint8_t)((*c2 & 0x0Fu) | ((((*c2 >> 4) + 1u) & 15u) << 4)); }
			if(s1 == mn){ *c3 = (uint8_t)((*c3 & 0x0Fu) | ((((*c3 >> 4) + 1u) & 15u) << 4)); }
		}
	}
	free(words);
}

/* make_bloom.cpp:336-354 */
void kwo_counter_fold(const kwo_counter *c, uint32_t log_2_filter_len, uint32_t num_hash, uint8_t *out)
{
	const uint64_t num_dst_block = (UINT64_C(1) << log_2_filter_len)/8;
	const uint64_t num_src_block = (UINT64_C(1) << c->max_log2)/8;
	memset(out, 0, num_dst_block);
	for(uint32_t h = 0; h < num_hash; ++h){
		for(uint64_t i = 0; i < num_src_block; i += num_dst_block){
			for(uint64_t j = 0; j < num_dst_block; ++j){ out[j] |= c->valid[h][i + j]; }
		}
	}
}
