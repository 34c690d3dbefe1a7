// kwage_amd/csrc/counter.hip -- Bloom filter construction with a minimum k-mer count on the device
// (SURVEY.md section 8f rank 4): make_bloom_filter()'s counting pass (make_bloom.cpp:76-504, count_words
// :506-621) with the reference's exact, ORDER-DEPENDENT semantics.
//
// The reference walks the k-mer occurrences of a read set strictly one after the other.  For each it
// reads four 4-bit counters (two per counting Bloom filter, indices hash_0..3 & (2^C - 1)), takes the
// minimum, and if that is below min_kmer_count increments only the counters equal to the minimum
// (conservative update); the occurrence that lifts the minimum to min_kmer_count sets the k-mer's five
// candidate bits (hash_0..4 & (2^M - 1)) and bumps num_kmer.  Two occurrences interact only if they
// share a counter, so any schedule that keeps the original order AMONG OCCURRENCES THAT SHARE A COUNTER
// gives the same counters, bits and num_kmer as the sequential loop.  That is what runs here, per chunk
// of the read stream:
//
//   round 0  hash_claim_kernel   every position: 2-bit pack, canonical word, 5 MurmurHash3 values (kept),
//                                then the claim step below
//   claim                        read the 4 counters; min >= min_kmer_count -> retire (a no-op in the
//                                sequential order too: what it read are updates of EARLIER occurrences
//                                only -- later ones sharing a counter cannot have run yet -- and counters
//                                only grow); else atomicMin(owner[counter], position) on its 4 counters
//   commit_kernel                an occurrence that owns all 4 of its counters is the earliest unfinished
//                                one on each of them: apply the reference's update, release the owners;
//                                the others go to the next round.  An occurrence that retires in a LATER
//                                claim step may still hold owner entries from earlier rounds; it releases
//                                them here (never during a claim step, where a later occurrence could
//                                slip in ahead of an earlier one that had already lost to it)
//
// The globally earliest unfinished occurrence always owns its counters, so every round retires at least
// one; in practice chains are as long as the multiplicity of a k-mer inside one chunk, capped by
// min_kmer_count (<= 15) because a k-mer that reached the threshold retires in the claim step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host.hpp"
#include "internal.h"
#include "kmer_device.hpp"

using namespace kwage;

namespace kwage {
hipStream_t ctx_stream(kwage_ctx *ctx);
int ctx_device(kwage_ctx *ctx);
int write_bloom_file(const char *out_path, const kwage_params *params, const FilterInfo &fi,
                     const unsigned char *bits, uint64_t nbytes);
int sample_info_to_filter_info(const kwage_sample_info *si, FilterInfo &fi);
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess){ \
	return fail(KWAGE_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } } while(0)

namespace {

static constexpr uint32_t NO_OWNER = 0xFFFFFFFFu;
static constexpr uint32_t RETIRE_FLAG = 0x80000000u;     // pending-list entry: only release stale owner entries
static constexpr uint32_t CHUNK_STARTS = 1u << 24;       // k-mer start positions per chunk
static constexpr int CT_THREADS = 256;

struct CounterArgs {
	const char *chars;            // chunk: carry (k-1) + new characters, sequences separated by '\n'
	uint32_t nstart;              // start positions in this chunk
	uint32_t k, min_count;
	uint32_t count_mask;          // 2^C - 1
	uint32_t seq_mask;            // 2^M - 1
	uint32_t *count_words;        // CountingBloom[2^C] viewed as u32 words (low nibble first, high nibble second)
	uint32_t *owner;              // [2^C]
	uint32_t *valid;              // 5 bit vectors of 2^M bits, back to back
	uint64_t valid_words;         // u32 words per vector
	uint32_t *hash;               // [5][CHUNK_STARTS] hash values of the chunk's positions
	const uint32_t *pend_in;      // claim/commit input list (null in round 0)
	uint32_t n_in;
	uint32_t *pend_out;
	unsigned long long *ctr;      // [0] num_valid_kmer, [1] length of pend_out
};

// Append `on` lanes' values to out[] with one atomic per wave; every lane of the wave must call.
__device__ __forceinline__ void wave_append(uint32_t *out, unsigned long long *len, bool on, uint32_t v)
{
	const unsigned long long m = __ballot(on);
	if(m == 0){ return; }
	const uint32_t lane = threadIdx.x & 63;
	unsigned long long base = 0;
	if(lane == (uint32_t)__ffsll((long long)m) - 1){ base = atomicAdd(len, (unsigned long long)__popcll(m)); }
	base = __shfl(base, __ffsll((long long)m) - 1);
	if(on){ out[base + __popcll(m & ((1ull << lane) - 1))] = v; }
}

__device__ __forceinline__ uint32_t count_first(const uint32_t *cw, uint32_t cell)  { return (cw[cell >> 2] >> ((cell & 3)*8)) & 15u; }
__device__ __forceinline__ uint32_t count_second(const uint32_t *cw, uint32_t cell) { return (cw[cell >> 2] >> ((cell & 3)*8 + 4)) & 15u; }

// make_bloom.cpp:546-560: the four counters and their minimum
__device__ __forceinline__ uint32_t min_of_four(const CounterArgs &a, const uint32_t (&c)[4], uint32_t (&v)[4])
{
	v[0] = count_first(a.count_words, c[0]);
	v[1] = count_first(a.count_words, c[1]);
	v[2] = count_second(a.count_words, c[2]);
	v[3] = count_second(a.count_words, c[3]);
	return min(min(v[0], v[1]), min(v[2], v[3]));
}

// claim step for one occurrence; returns false if it retires (:560 false -> nothing happens)
__device__ __forceinline__ bool claim(const CounterArgs &a, uint32_t pos, const uint32_t (&c)[4])
{
	uint32_t v[4];
	if(min_of_four(a, c, v) >= a.min_count){ return false; }
#pragma unroll
	for(int j = 0; j < 4; ++j){ atomicMin(&a.owner[c[j]], pos); }
	return true;
}

__global__ __launch_bounds__(CT_THREADS) void hash_claim_kernel(CounterArgs a)
{
	__shared__ uint8_t codes[CT_THREADS + 32];
	const uint32_t p0 = blockIdx.x*CT_THREADS;
	const uint32_t nload = min((uint32_t)CT_THREADS + a.k - 1, a.nstart + a.k - 1 - p0);
	for(uint32_t i = threadIdx.x; i < nload; i += CT_THREADS){ codes[i] = (uint8_t)base_code(a.chars[p0 + i]); }
	__syncthreads();
	const uint32_t p = p0 + threadIdx.x;
	bool pending = false;
	if(p < a.nstart){
		uint64_t w = 0;
		uint32_t bad = 0;
		for(uint32_t j = 0; j < a.k; ++j){
			const uint32_t cd = codes[threadIdx.x + j];
			bad |= cd & 4u;
			w = (w << 2) | (cd & 3u);
		}
		if(!bad){
			const uint64_t rc = revcomp2(w, a.k);
			const uint64_t canon = (w < rc) ? w : rc;                 // word.h:164
			MurmurKeys mk;
			murmur_keys(canon, a.k, mk);
			uint32_t h[5];
#pragma unroll
			for(uint32_t s = 0; s < 5; ++s){
				h[s] = murmur_finish(mk, a.k, s);                      // make_bloom.cpp:530, seeds 0..4
				a.hash[(uint64_t)s*CHUNK_STARTS + p] = h[s];
			}
			const uint32_t c[4] = { h[0] & a.count_mask, h[1] & a.count_mask, h[2] & a.count_mask, h[3] & a.count_mask };
			pending = claim(a, p, c);
		}
	}
	wave_append(a.pend_out, a.ctr + 1, pending, p);
}

__global__ __launch_bounds__(CT_THREADS) void claim_kernel(CounterArgs a)
{
	const uint32_t i = blockIdx.x*CT_THREADS + threadIdx.x;
	bool pending = false;
	uint32_t p = 0;
	if(i < a.n_in){
		p = a.pend_in[i];
		uint32_t c[4];
#pragma unroll
		for(uint32_t s = 0; s < 4; ++s){ c[s] = a.hash[(uint64_t)s*CHUNK_STARTS + p] & a.count_mask; }
		pending = true;
		if(!claim(a, p, c)){ p |= RETIRE_FLAG; }       // it claimed in an earlier round: the commit step releases that
	}
	wave_append(a.pend_out, a.ctr + 1, pending, p);
}

__global__ __launch_bounds__(CT_THREADS) void commit_kernel(CounterArgs a)
{
	const uint32_t i = blockIdx.x*CT_THREADS + threadIdx.x;
	bool pending = false;
	bool counted = false;
	uint32_t p = 0;
	if(i < a.n_in){
		p = a.pend_in[i];
		const bool retire = (p & RETIRE_FLAG) != 0;
		p &= ~RETIRE_FLAG;
		uint32_t h[5], c[4];
#pragma unroll
		for(uint32_t s = 0; s < 5; ++s){ h[s] = a.hash[(uint64_t)s*CHUNK_STARTS + p]; }
#pragma unroll
		for(uint32_t s = 0; s < 4; ++s){ c[s] = h[s] & a.count_mask; }
		if(retire){
#pragma unroll
			for(int j = 0; j < 4; ++j){ atomicCAS(&a.owner[c[j]], p, NO_OWNER); }
		}
		const bool ready = !retire && (a.owner[c[0]] == p) && (a.owner[c[1]] == p) && (a.owner[c[2]] == p) && (a.owner[c[3]] == p);
		if(!ready){
			pending = !retire;
		}
		else{
			uint32_t v[4];
			const uint32_t mn = min_of_four(a, c, v);
			if(mn < a.min_count){                                              // :560 (always true here: it claimed)
				if(mn == a.min_count - 1){                                     // :562-583
					counted = true;
#pragma unroll
					for(uint32_t s = 0; s < 5; ++s){
						const uint32_t b = h[s] & a.seq_mask;
						atomicOr(&a.valid[(uint64_t)s*a.valid_words + (b >> 5)], 1u << (b & 31));
					}
				}
				// :587-602: four separate ++ on 4-bit fields, each decided by the value READ BEFORE any
				// increment; two hashes on one element increment it twice and the field wraps at 16.
				// Deltas go in with atomicAdd on the enclosing word: other occurrences may be updating
				// neighbouring nibbles of the same word right now, and no delta carries out of its nibble.
				if(c[0] == c[1]){
					if(v[0] == mn){
						const uint32_t nv = (v[0] + 2) & 15u;
						atomicAdd(&a.count_words[c[0] >> 2], (nv - v[0]) << ((c[0] & 3)*8));
					}
				}
				else{
					if(v[0] == mn){ atomicAdd(&a.count_words[c[0] >> 2], 1u << ((c[0] & 3)*8)); }
					if(v[1] == mn){ atomicAdd(&a.count_words[c[1] >> 2], 1u << ((c[1] & 3)*8)); }
				}
				if(c[2] == c[3]){
					if(v[2] == mn){
						const uint32_t nv = (v[2] + 2) & 15u;
						atomicAdd(&a.count_words[c[2] >> 2], (nv - v[2]) << ((c[2] & 3)*8 + 4));
					}
				}
				else{
					if(v[2] == mn){ atomicAdd(&a.count_words[c[2] >> 2], 1u << ((c[2] & 3)*8 + 4)); }
					if(v[3] == mn){ atomicAdd(&a.count_words[c[3] >> 2], 1u << ((c[3] & 3)*8 + 4)); }
				}
			}
#pragma unroll
			for(int j = 0; j < 4; ++j){ a.owner[c[j]] = NO_OWNER; }
		}
	}
	wave_append(a.pend_out, a.ctr + 1, pending, p);
	const unsigned long long m = __ballot(counted);
	if(m && (threadIdx.x & 63) == (uint32_t)__ffsll((long long)m) - 1){ atomicAdd(a.ctr, (unsigned long long)__popcll(m)); }
}

// make_bloom.cpp:336-354: dst = OR over h < num_hash, over blocks of dst length, of valid_bits[h]
__global__ __launch_bounds__(256) void fold_kernel(const uint32_t *valid, uint64_t valid_words, uint32_t num_hash,
                                                  uint32_t *dst, uint64_t dst_words)
{
	const uint64_t j = (uint64_t)blockIdx.x*256 + threadIdx.x;
	if(j >= dst_words){ return; }
	uint32_t acc = 0;
	for(uint32_t h = 0; h < num_hash; ++h){
		const uint32_t *src = valid + (uint64_t)h*valid_words;
		for(uint64_t i = j; i < valid_words; i += dst_words){ acc |= src[i]; }
	}
	dst[j] = acc;
}

}  // namespace

struct kwage_bloom_counter {
	kwage_ctx *ctx = nullptr;
	hipStream_t stream = nullptr;
	uint32_t k = 0, min_count = 0, logc = 0, max_log2 = 0;
	uint32_t logc_cap = 0;                   // counting filters are allocated for 2^logc_cap elements
	uint32_t *d_count = nullptr, *d_owner = nullptr, *d_valid = nullptr, *d_hash = nullptr;
	uint32_t *d_pend[2] = { nullptr, nullptr };
	unsigned long long *d_ctr = nullptr;
	char *d_chars = nullptr;
	char *h_chars = nullptr;                 // pinned staging, CHUNK_STARTS + 64 bytes
	unsigned long long *h_ctr = nullptr;     // pinned
	uint64_t valid_words = 0;
	uint32_t staged = 0;                     // characters in h_chars (carry included)
	bool started = false;                    // the stream has produced at least one character
	kwage_bloom_counter_stats st{};
};

namespace {

int read_ctr(kwage_bloom_counter *bc)
{
	HIP_TRY(hipMemcpyAsync(bc->h_ctr, bc->d_ctr, 2*sizeof(unsigned long long), hipMemcpyDeviceToHost, bc->stream));
	HIP_TRY(hipStreamSynchronize(bc->stream));
	return KWAGE_OK;
}

// Run the staged characters through the counting pass; keep the last k-1 as the next chunk's carry.
int flush_chunk(kwage_bloom_counter *bc, bool final)
{
	const uint32_t k = bc->k;
	if(bc->staged >= k){
		const uint32_t nstart = bc->staged - k + 1;
		HIP_TRY(hipMemcpyAsync(bc->d_chars, bc->h_chars, bc->staged, hipMemcpyHostToDevice, bc->stream));
		CounterArgs a{};
		a.chars = bc->d_chars; a.nstart = nstart; a.k = k; a.min_count = bc->min_count;
		a.count_mask = (uint32_t)((1ull << bc->logc) - 1);
		a.seq_mask = (uint32_t)((1ull << bc->max_log2) - 1);
		a.count_words = bc->d_count; a.owner = bc->d_owner; a.valid = bc->d_valid; a.valid_words = bc->valid_words;
		a.hash = bc->d_hash; a.ctr = bc->d_ctr;
		int cur = 0;
		HIP_TRY(hipMemsetAsync(bc->d_ctr + 1, 0, sizeof(unsigned long long), bc->stream));
		a.pend_in = nullptr; a.n_in = 0; a.pend_out = bc->d_pend[cur];
		hipLaunchKernelGGL(hash_claim_kernel, dim3((nstart + CT_THREADS - 1)/CT_THREADS), dim3(CT_THREADS), 0, bc->stream, a);
		HIP_TRY(hipGetLastError());
		int rc;
		if((rc = read_ctr(bc))){ return rc; }
		uint32_t n = (uint32_t)bc->h_ctr[1];
		bc->st.positions += nstart;
		uint32_t rounds = 0;
		while(n){
			// commit: pend[cur] -> pend[cur^1] (not ready yet)
			HIP_TRY(hipMemsetAsync(bc->d_ctr + 1, 0, sizeof(unsigned long long), bc->stream));
			a.pend_in = bc->d_pend[cur]; a.n_in = n; a.pend_out = bc->d_pend[cur ^ 1];
			hipLaunchKernelGGL(commit_kernel, dim3((n + CT_THREADS - 1)/CT_THREADS), dim3(CT_THREADS), 0, bc->stream, a);
			HIP_TRY(hipGetLastError());
			if((rc = read_ctr(bc))){ return rc; }
			const uint32_t left = (uint32_t)bc->h_ctr[1];
			++rounds;
			bc->st.occurrences_committed += n - left;
			if(left == n){ return fail(KWAGE_ERR_STATE, "kwage_bloom_counter: no progress in a commit round (internal error)"); }
			n = left;
			cur ^= 1;
			if(!n){ break; }
			// claim again: pend[cur] -> pend[cur^1] (still below the threshold)
			HIP_TRY(hipMemsetAsync(bc->d_ctr + 1, 0, sizeof(unsigned long long), bc->stream));
			a.pend_in = bc->d_pend[cur]; a.n_in = n; a.pend_out = bc->d_pend[cur ^ 1];
			hipLaunchKernelGGL(claim_kernel, dim3((n + CT_THREADS - 1)/CT_THREADS), dim3(CT_THREADS), 0, bc->stream, a);
			HIP_TRY(hipGetLastError());
			if((rc = read_ctr(bc))){ return rc; }
			n = (uint32_t)bc->h_ctr[1];
			cur ^= 1;
		}
		bc->st.rounds += rounds;
		bc->st.max_rounds = std::max(bc->st.max_rounds, rounds);
		++bc->st.chunks;
		bc->st.num_valid_kmer = bc->h_ctr[0];
		// carry: the last k-1 characters start k-mers that end in the next chunk
		if(!final && k > 1){ memmove(bc->h_chars, bc->h_chars + bc->staged - (k - 1), k - 1); }
		bc->staged = final ? 0 : (k - 1);
	}
	else if(final){ bc->staged = 0; }
	return KWAGE_OK;
}

}  // namespace

// make_bloom.cpp:105-130
extern "C" uint32_t kwage_counting_filter_log2(uint64_t num_bp)
{
	uint64_t lg = 32;                                   // MAX_LOG_COUNT_FILTER_LEN :20
	if(num_bp > 0){
		const double counting_length = 1.0/( 1.0 - pow( 1.0 - pow(1.0e-2, 1.0/4.0), 1.0/(double)(2*num_bp) ) );
		lg = (uint64_t)ceil( log(counting_length)/log(2.0) );
		lg = std::min<uint64_t>(std::max<uint64_t>(lg, 18), 32);
	}
	return (uint32_t)lg;
}

// bloom.cpp:72-121
extern "C" uint64_t kwage_approximate_max_kmers(float p_bound, uint32_t min_lg, uint32_t max_lg)
{
	for(uint64_t lgn = 1; lgn < 64; ++lgn){
		const uint64_t num_kmer = 1ull << lgn;
		bool valid = false;
		for(uint64_t lg = min_lg; lg <= max_lg && lg < 64 && !valid; ++lg){
			const float best_p = 10.0f;
			for(uint32_t nh = KWAGE_MIN_NUM_HASH; nh <= KWAGE_MAX_NUM_HASH && !valid; ++nh){
				const uint64_t len = 1ull << lg;
				const double p = pow(1.0 - pow(1.0 - 1.0/len, (double)(num_kmer*nh)), (double)nh);
				if((p <= p_bound) && (p < best_p)){ valid = true; }
			}
		}
		if(!valid){ return num_kmer; }
	}
	return ~0ull;
}

extern "C" void kwage_bloom_counter_destroy(kwage_bloom_counter *bc)
{
	if(!bc){ return; }
	(void)hipSetDevice(ctx_device(bc->ctx));
	(void)hipStreamSynchronize(bc->stream);
	if(bc->d_count){ (void)hipFree(bc->d_count); }
	if(bc->d_owner){ (void)hipFree(bc->d_owner); }
	if(bc->d_valid){ (void)hipFree(bc->d_valid); }
	if(bc->d_hash){ (void)hipFree(bc->d_hash); }
	if(bc->d_pend[0]){ (void)hipFree(bc->d_pend[0]); }
	if(bc->d_pend[1]){ (void)hipFree(bc->d_pend[1]); }
	if(bc->d_ctr){ (void)hipFree(bc->d_ctr); }
	if(bc->d_chars){ (void)hipFree(bc->d_chars); }
	if(bc->h_chars){ (void)hipHostFree(bc->h_chars); }
	if(bc->h_ctr){ (void)hipHostFree(bc->h_ctr); }
	delete bc;
}

extern "C" int kwage_bloom_counter_create(kwage_ctx *ctx, uint32_t kmer_len, int32_t hash_func, uint32_t min_kmer_count,
                                          uint32_t log_2_counting_filter_len, uint32_t max_log_2_filter_len,
                                          kwage_bloom_counter **out)
{
	if(!ctx || !out){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_create: NULL argument"); }
	*out = nullptr;
	if(kmer_len < 1 || kmer_len > 32){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_create: kmer_len must be in [1, 32]"); }
	if(hash_func != KWAGE_HASH_MURMUR32){ return fail(KWAGE_ERR_ARG, "bigsi_hash: Unknown hash function"); }          // hash.cpp:107
	if(min_kmer_count < 1 || min_kmer_count > 15){      // make_bloom.cpp:91-93 (MAX_COUNT 15); 0 would never insert anything
		return fail(KWAGE_ERR_ARG, "make_bloom_filter: min_kmer_count must be in [1, 15]");
	}
	if(log_2_counting_filter_len < 2 || log_2_counting_filter_len > 32){
		return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_create: log_2_counting_filter_len must be in [2, 32]");
	}
	if(max_log_2_filter_len < 5 || max_log_2_filter_len > 32){       // options.cpp:774 (32-bit hash)
		return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_create: max_log_2_filter_len must be in [5, 32]");
	}
	HIP_TRY(hipSetDevice(ctx_device(ctx)));
	kwage_bloom_counter *bc = new (std::nothrow) kwage_bloom_counter();
	if(!bc){ return fail(KWAGE_ERR_DEVICE, "out of host memory"); }
	bc->ctx = ctx; bc->stream = ctx_stream(ctx);
	bc->k = kmer_len; bc->min_count = min_kmer_count; bc->logc = bc->logc_cap = log_2_counting_filter_len; bc->max_log2 = max_log_2_filter_len;
	bc->valid_words = (1ull << max_log_2_filter_len)/32;
	const uint64_t cells = 1ull << bc->logc;
	hipError_t e = hipMalloc(&bc->d_count, cells);
	if(e == hipSuccess){ e = hipMalloc(&bc->d_owner, cells*sizeof(uint32_t)); }
	if(e == hipSuccess){ e = hipMalloc(&bc->d_valid, 5*bc->valid_words*sizeof(uint32_t)); }
	if(e == hipSuccess){ e = hipMalloc(&bc->d_hash, 5ull*CHUNK_STARTS*sizeof(uint32_t)); }
	if(e == hipSuccess){ e = hipMalloc(&bc->d_pend[0], (uint64_t)CHUNK_STARTS*sizeof(uint32_t)); }
	if(e == hipSuccess){ e = hipMalloc(&bc->d_pend[1], (uint64_t)CHUNK_STARTS*sizeof(uint32_t)); }
	if(e == hipSuccess){ e = hipMalloc(&bc->d_ctr, 2*sizeof(unsigned long long)); }
	if(e == hipSuccess){ e = hipMalloc(&bc->d_chars, CHUNK_STARTS + 64); }
	if(e == hipSuccess){ e = hipHostMalloc(&bc->h_chars, CHUNK_STARTS + 64, hipHostMallocDefault); }
	if(e == hipSuccess){ e = hipHostMalloc(&bc->h_ctr, 2*sizeof(unsigned long long), hipHostMallocDefault); }
	if(e == hipSuccess){ e = hipMemsetAsync(bc->d_count, 0, cells, bc->stream); }                      // make_bloom.cpp:159
	if(e == hipSuccess){ e = hipMemsetAsync(bc->d_owner, 0xFF, cells*sizeof(uint32_t), bc->stream); }
	if(e == hipSuccess){ e = hipMemsetAsync(bc->d_valid, 0, 5*bc->valid_words*sizeof(uint32_t), bc->stream); }   // :167-169
	if(e == hipSuccess){ e = hipMemsetAsync(bc->d_ctr, 0, 2*sizeof(unsigned long long), bc->stream); }
	if(e == hipSuccess){ e = hipStreamSynchronize(bc->stream); }
	if(e != hipSuccess){
		kwage_bloom_counter_destroy(bc);
		return fail(KWAGE_ERR_DEVICE, "kwage_bloom_counter_create: %s (counting filters 2^%u, candidate bits 5 x 2^%u)",
		            hipGetErrorString(e), log_2_counting_filter_len, max_log_2_filter_len);
	}
	*out = bc;
	return KWAGE_OK;
}

// Start the next sample in the same object: everything is zeroed again (make_bloom.cpp:159,167-169), the
// device allocations are kept.  The counting filters may shrink (a smaller read set), never grow.
extern "C" int kwage_bloom_counter_reset(kwage_bloom_counter *bc, uint32_t min_kmer_count, uint32_t log_2_counting_filter_len)
{
	if(!bc){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_reset: NULL argument"); }
	if(min_kmer_count < 1 || min_kmer_count > 15){ return fail(KWAGE_ERR_ARG, "make_bloom_filter: min_kmer_count must be in [1, 15]"); }
	if(log_2_counting_filter_len < 2 || log_2_counting_filter_len > bc->logc_cap){
		return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_reset: log_2_counting_filter_len must be in [2, %u] (the size the object was created with)", bc->logc_cap);
	}
	HIP_TRY(hipSetDevice(ctx_device(bc->ctx)));
	HIP_TRY(hipStreamSynchronize(bc->stream));
	bc->min_count = min_kmer_count;
	bc->logc = log_2_counting_filter_len;
	bc->staged = 0;
	bc->started = false;
	bc->st = kwage_bloom_counter_stats{};
	// owner[] is all NO_OWNER again whenever a chunk has been processed to the end, so only the counters, the
	// candidate bits and the running totals need clearing
	HIP_TRY(hipMemsetAsync(bc->d_count, 0, 1ull << bc->logc, bc->stream));
	HIP_TRY(hipMemsetAsync(bc->d_valid, 0, 5*bc->valid_words*sizeof(uint32_t), bc->stream));
	HIP_TRY(hipMemsetAsync(bc->d_ctr, 0, 2*sizeof(unsigned long long), bc->stream));
	HIP_TRY(hipStreamSynchronize(bc->stream));
	return KWAGE_OK;
}

extern "C" int kwage_bloom_counter_add(kwage_bloom_counter *bc, const char *seqs, const uint64_t *offsets, uint32_t n_seqs)
{
	if(!bc || !offsets || (!seqs && n_seqs && offsets[n_seqs] > offsets[0])){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_add: NULL argument"); }
	HIP_TRY(hipSetDevice(ctx_device(bc->ctx)));
	const auto t0 = std::chrono::steady_clock::now();
	const uint32_t cap = CHUNK_STARTS + bc->k - 1;       // characters per chunk
	int rc;
	for(uint32_t i = 0; i < n_seqs; ++i){
		if(offsets[i + 1] < offsets[i]){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_add: offsets must be non-decreasing"); }
		// a separator in front of every fragment but the first resets the k-mer run exactly like the start
		// of a new ForEachDuplexWord loop does (word.h:76-81)
		if(bc->started){
			if(bc->staged == cap && (rc = flush_chunk(bc, false))){ return rc; }
			bc->h_chars[bc->staged++] = '\n';
		}
		bc->started = true;
		uint64_t s = offsets[i];
		const uint64_t e = offsets[i + 1];
		bc->st.num_bp += e - s;                          // make_bloom.cpp:202,238,290
		while(s < e){
			if(bc->staged == cap && (rc = flush_chunk(bc, false))){ return rc; }
			const uint64_t take = std::min<uint64_t>(e - s, cap - bc->staged);
			memcpy(bc->h_chars + bc->staged, seqs + s, take);
			bc->staged += (uint32_t)take;
			s += take;
		}
	}
	bc->st.add_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return KWAGE_OK;
}

extern "C" int kwage_bloom_counter_flush(kwage_bloom_counter *bc)
{
	if(!bc){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_flush: NULL argument"); }
	HIP_TRY(hipSetDevice(ctx_device(bc->ctx)));
	const auto t0 = std::chrono::steady_clock::now();
	// not final: later fragments continue the same stream (the carry keeps the separator logic simple)
	int rc = flush_chunk(bc, false);
	bc->st.add_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return rc;
}

extern "C" int kwage_bloom_counter_get_stats(kwage_bloom_counter *bc, kwage_bloom_counter_stats *out)
{
	if(!bc || !out){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_get_stats: NULL argument"); }
	int rc = kwage_bloom_counter_flush(bc);
	if(rc){ return rc; }
	*out = bc->st;
	return KWAGE_OK;
}

extern "C" int kwage_bloom_counter_read_counts(kwage_bloom_counter *bc, uint64_t first, uint64_t n, void *out)
{
	if(!bc || (!out && n)){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_read_counts: NULL argument"); }
	if(first + n > (1ull << bc->logc)){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_read_counts: range outside the counting filter"); }
	int rc = kwage_bloom_counter_flush(bc);
	if(rc){ return rc; }
	if(n){ HIP_TRY(hipMemcpy(out, (const char*)bc->d_count + first, n, hipMemcpyDeviceToHost)); }
	return KWAGE_OK;
}

extern "C" int kwage_bloom_counter_read_valid_bits(kwage_bloom_counter *bc, uint32_t hash, uint64_t first_byte, uint64_t nbytes, void *out)
{
	if(!bc || (!out && nbytes)){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_read_valid_bits: NULL argument"); }
	if(hash >= 5 || first_byte + nbytes > bc->valid_words*4){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_read_valid_bits: range outside the bit vector"); }
	int rc = kwage_bloom_counter_flush(bc);
	if(rc){ return rc; }
	if(nbytes){ HIP_TRY(hipMemcpy(out, (const char*)(bc->d_valid + (uint64_t)hash*bc->valid_words) + first_byte, nbytes, hipMemcpyDeviceToHost)); }
	return KWAGE_OK;
}

extern "C" int kwage_bloom_counter_finish(kwage_bloom_counter *bc, float false_positive_probability, uint32_t min_log_2_filter_len,
                                          const kwage_sample_info *info, const char *out_path, kwage_params *chosen, int *status)
{
	if(!bc || !status){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_finish: NULL argument"); }
	*status = KWAGE_BLOOM_INVALID;
	if(min_log_2_filter_len < 5 || min_log_2_filter_len > bc->max_log2){
		return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_finish: min_log_2_filter_len must be in [5, max_log_2_filter_len]");
	}
	FilterInfo fi;
	int rc;
	if(out_path){
		if(!info){ return fail(KWAGE_ERR_ARG, "kwage_bloom_counter_finish: sample info is required to write a file"); }
		if((rc = sample_info_to_filter_info(info, fi))){ return rc; }
	}
	if((rc = kwage_bloom_counter_flush(bc))){ return rc; }
	const uint64_t num_kmer = bc->st.num_valid_kmer;
	// make_bloom.cpp:210-216 (checked after every fragment there; num_kmer only grows, so the final value decides)
	if(kwage_approximate_max_kmers(false_positive_probability, min_log_2_filter_len, bc->max_log2) < num_kmer){ return KWAGE_OK; }
	kwage_params p{};
	// make_bloom.cpp:309-334: no parameters (or no k-mers: bloom.cpp:16-18) -> STATUS_BLOOM_INVALID
	if(kwage_optimal_bloom_param(bc->k, num_kmer, false_positive_probability, min_log_2_filter_len, bc->max_log2, &p) != KWAGE_OK){ return KWAGE_OK; }
	if(chosen){ *chosen = p; }
	const uint64_t dst_words = (1ull << p.log_2_filter_len)/32;
	uint32_t *d_dst = nullptr;
	HIP_TRY(hipMalloc(&d_dst, dst_words*sizeof(uint32_t)));
	hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((dst_words + 255)/256)), dim3(256), 0, bc->stream,
	                   bc->d_valid, bc->valid_words, p.num_hash, d_dst, dst_words);
	std::vector<unsigned char> bits(dst_words*4);
	hipError_t e = hipGetLastError();
	if(e == hipSuccess){ e = hipMemcpyAsync(bits.data(), d_dst, bits.size(), hipMemcpyDeviceToHost, bc->stream); }
	if(e == hipSuccess){ e = hipStreamSynchronize(bc->stream); }
	(void)hipFree(d_dst);
	if(e != hipSuccess){ return fail(KWAGE_ERR_DEVICE, "kwage_bloom_counter_finish: %s", hipGetErrorString(e)); }
	if(out_path && (rc = write_bloom_file(out_path, &p, fi, bits.data(), bits.size()))){ return rc; }
	*status = KWAGE_BLOOM_SUCCESS;
	return KWAGE_OK;
}
