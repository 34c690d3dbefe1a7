// kwage_amd/csrc/kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the kwage
// search path.  Integer / bit work only, HBM-bandwidth bound: no MFMA anywhere.
//
//   kmer_kernel   word.h:73-104,161-165 (2-bit canonical k-mers) + kwage.cpp:362-366 (distinct
//                 set) + hash.cpp:176-234 (MurmurHash3_x86_32 of the ASCII k-mer) + kwage.cpp:411-412
//                 (row index) + kwage.cpp:388 (float32 threshold).
//   and_kernel    kwage.cpp:404-470 at threshold == 1.0f: gather the addressed bit-slice rows,
//                 AND them (bloom.h:245-262), extract hits (kwage.cpp:489-538).
//   and_walk_kernel  the same reduction for rows of 3..16 KiB: a persistent grid, every wave walks an equal share
//                 of the batch's row list over the whole row width through bounds-checked buffer loads; (query,
//                 tile) pairs cut by a share boundary are finished through a small OR buffer in memory.
//   count_kernel  the threshold < 1 path: per k-mer AND over hashes, then bit-sliced (vertical)
//                 per-column counters in registers instead of bloom.h:291-330's per-bit loop.
//
// Data layout in HBM: one row-major bit matrix per database group; row r starts
// at r*stride (stride a multiple of 128 B); global column c is byte c/8, bit c%8 of the row.
// A wave owns a "tile" = (query, 64*VEC*16 contiguous bytes of every addressed row): each lane
// holds VEC 16-byte vectors, so one global_load_dwordx4 per lane reads 1 KiB of a row per wave,
// fully coalesced; UNROLL rows are kept in flight per wave.
#ifndef KWAGE_AMD_KERNELS_HPP
#define KWAGE_AMD_KERNELS_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kwage_amd.h"
#include "kmer_device.hpp"

namespace kwage {

static constexpr int WAVE = 64;
static constexpr int KM_THREADS = 256;               // largest k-mer workgroup; short queries use 64 or 128
static constexpr uint32_t KM_LDS_SLOTS = 4096;      // 32 KiB of u64 slots: queries up to 2048 positions
static constexpr uint32_t KM_CHUNK = 1024;          // positions per workgroup of a long query (4 tiles of 256)
static constexpr uint64_t KM_EMPTY = ~0ull;         // never a canonical word: min(w, rc) < all-ones
static constexpr int SEARCH_THREADS = 256;          // 4 waves, one tile each

// (2-bit packing + MurmurHash3 device helpers: kmer_device.hpp)

__device__ __forceinline__ uint32_t table_log2(uint64_t npos)
{
	// smallest power of two >= 2*npos, at least 64 slots
	uint32_t lg = 6;
	while((1ull << lg) < 2*npos){ ++lg; }
	return lg;
}

// Insert into an open-addressing set. Returns true when this call created the entry.
// Slot arithmetic is 64-bit: the shared table of a whole sample (Bloom construction) and the table of a query
// above 2^30 positions have 2^32 slots or more.
template <typename PTR>
__device__ __forceinline__ bool set_insert(PTR tab, uint32_t lg, uint64_t w)
{
	const uint64_t mask = (1ull << lg) - 1ull;
	uint64_t s = ((w * 0x9E3779B97F4A7C15ull) >> (64 - lg)) & mask;
	while(true){
		const unsigned long long old = atomicCAS(&tab[s], (unsigned long long)KM_EMPTY, (unsigned long long)w);
		if(old == KM_EMPTY){ return true; }
		if(old == w){ return false; }
		s = (s + 1) & mask;
	}
}

struct KmerArgs {
	const char *seqs;               // concatenated queries
	const uint64_t *seq_off;        // n_queries + 1
	const uint64_t *pos_off;        // n_queries + 1: prefix of max(len-k+1, 0)
	const uint64_t *tab_off;        // per query: slot offset into g_tables (only read for long queries)
	unsigned long long *g_tables;   // pre-filled with KM_EMPTY
	uint32_t k, num_hash, row_mask;
	float threshold;
	int complete_match;
	uint32_t *rows;                 // [pos_off[q] + j][num_hash] row indices (may be null)
	uint64_t *kmers_out;            // [pos_off[q] + j] distinct canonical words (may be null)
	uint32_t *nkmer;                // per query
	uint32_t *qthr;                 // per query threshold (kwage.cpp:388)
	unsigned long long *total_kmers;
	// Bloom construction: all sequences share ONE global set (distinct k-mers of the whole sample) and
	// every new k-mer sets its num_hash bits in `bloom_bits` (bit index = hash & row_mask).
	uint32_t lds_slots;             // capacity of the dynamic-LDS table (a power of two <= KM_LDS_SLOTS)
	uint32_t shared_lg;             // 0 = one set per query (search); else log2 slots of the shared table
	uint32_t *bloom_bits;           // may be null
	// work list (null: workgroup i = sequence i, whole): the query of every workgroup and the first position of its
	// chunk; a query longer than KM_CHUNK positions with a global set is handled by several workgroups
	const uint32_t *chunk_q;
	const uint64_t *chunk_t0;
};

// MULTI: this workgroup handles positions [t_begin, t_end) of a query that other workgroups work on too (they share
// its global set): new k-mers are counted per tile in LDS, ONE global add per tile reserves their places behind
// nkmer[q] (zeroed before the launch), and the threshold is left to kmer_finish_kernel.
template <bool LDS_TAB, bool MULTI, typename TAB>
__device__ __forceinline__ void kmer_body(const KmerArgs &a, uint32_t q, uint64_t s0, uint64_t len,
                                          uint64_t t_begin, uint64_t t_end, TAB tab, uint32_t lg,
                                          uint8_t *codes, uint32_t *count, uint32_t *tile_base)
{
	const uint32_t k = a.k;
	const uint64_t base = a.pos_off[q];
	const uint64_t kmask = (k == 32) ? ~0ull : ((1ull << (2*k)) - 1ull);

	if(LDS_TAB){
		for(uint32_t i = threadIdx.x; i < (1u << lg); i += blockDim.x){ tab[i] = KM_EMPTY; }
	}
	if(threadIdx.x == 0){ *count = 0; }
	__syncthreads();

	for(uint64_t t0 = t_begin; t0 < t_end; t0 += blockDim.x){
		// stage the 2-bit codes of characters [t0, t0 + blockDim.x + k - 1)
		const uint32_t nchar = (uint32_t)min((uint64_t)(blockDim.x + k - 1), len - t0);
		for(uint32_t i = threadIdx.x; i < nchar; i += blockDim.x){
			codes[i] = (uint8_t)base_code(a.seqs[s0 + t0 + i]);
		}
		__syncthreads();

		const uint64_t p = t0 + threadIdx.x;
		bool fresh = false;
		uint64_t canon = 0;
		if(p < t_end){
			uint64_t w = 0;
			uint32_t bad = 0;
			for(uint32_t j = 0; j < k; ++j){
				const uint32_t c = codes[threadIdx.x + j];
				bad |= c >> 2;
				w = (w << 2) | (c & 3u);
			}
			if(!bad){   // ValidWord: k consecutive good bases end here (word.h:162)
				w &= kmask;
				const uint64_t rc = revcomp2(w, k);
				canon = (w < rc) ? w : rc;                         // word.h:165
				fresh = set_insert(tab, lg, canon);                // first time this k-mer is seen
			}
		}
		uint32_t idx = 0;
		if(MULTI){
			if(fresh){ idx = atomicAdd(count, 1u); }               // rank within the tile
			__syncthreads();
			if(threadIdx.x == 0){
				const uint32_t n_new = *count;
				*tile_base = n_new ? atomicAdd(a.nkmer + q, n_new) : 0u;
				*count = 0;
			}
			__syncthreads();
			idx += *tile_base;
		}
		else if(fresh){ idx = atomicAdd(count, 1u); }
		if(fresh){
			if(a.kmers_out){ a.kmers_out[base + idx] = canon; }
			if(a.bloom_bits){
				MurmurKeys mk;
				murmur_keys(canon, k, mk);
				for(uint32_t h = 0; h < a.num_hash; ++h){
					const uint32_t bit = murmur_finish(mk, k, h) & a.row_mask;
					atomicOr(a.bloom_bits + (bit >> 5), 1u << (bit & 31));      // LSB first, bloom.h:162
				}
			}
			if(a.rows){
				MurmurKeys mk;
				murmur_keys(canon, k, mk);
				for(uint32_t h = 0; h < a.num_hash; ++h){  // seed = hash index, kwage.cpp:409-412
					a.rows[(base + idx)*a.num_hash + h] = murmur_finish(mk, k, h) & a.row_mask;
				}
			}
		}
		__syncthreads();
	}

	if(!MULTI && threadIdx.x == 0){
		const uint32_t n = *count;
		a.nkmer[q] = n;
		// kwage.cpp:388: unsigned = float * unsigned  ->  float32 product, truncation
		a.qthr[q] = a.complete_match ? 0u : (uint32_t)__fmul_rn(a.threshold, (float)n);
		// one same-address atomic per workgroup costs ~12 ns at the memory side (100k reads = 1.2 ms): the
		// search path sums nkmer[] on the host instead and passes no counter
		if(n && a.total_kmers){ atomicAdd(a.total_kmers, (unsigned long long)n); }
	}
}

// Launched with 64, 128 or 256 threads and lds_slots*8 bytes of dynamic LDS (the distinct-set table): short
// reads get small workgroups and small tables, so many of them are resident per CU.
__global__ __launch_bounds__(KM_THREADS) void kmer_kernel(KmerArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned long long lds_tab[];
	__shared__ uint8_t codes[KM_THREADS + KWAGE_MAX_WORD_LEN];
	__shared__ uint32_t count, tile_base;

	const uint32_t q = a.chunk_q ? a.chunk_q[blockIdx.x] : blockIdx.x;
	const uint64_t t_begin = a.chunk_q ? a.chunk_t0[blockIdx.x] : 0;
	const uint64_t s0 = a.seq_off[q];
	const uint64_t len = a.seq_off[q + 1] - s0;
	const uint64_t npos = (len >= a.k) ? (len - a.k + 1) : 0;

	if(npos == 0){    // kwage.cpp:369-371: query too short
		if(threadIdx.x == 0){ a.nkmer[q] = 0; a.qthr[q] = 0; }
		return;
	}

	if(a.shared_lg){
		kmer_body<false, false>(a, q, s0, len, 0, npos, a.g_tables, a.shared_lg, codes, &count, &tile_base);
		return;
	}
	const uint32_t lg = table_log2(npos);
	if((1ull << lg) <= a.lds_slots){
		kmer_body<true, false>(a, q, s0, len, 0, npos, lds_tab, lg, codes, &count, &tile_base);
	}
	else if(npos <= KM_CHUNK){
		kmer_body<false, false>(a, q, s0, len, 0, npos, a.g_tables + a.tab_off[q], lg, codes, &count, &tile_base);
	}
	else{             // one of several workgroups on this query (the host cut it the same way, batch_prepare)
		kmer_body<false, true>(a, q, s0, len, t_begin, min(npos, t_begin + (uint64_t)KM_CHUNK), a.g_tables + a.tab_off[q], lg, codes, &count, &tile_base);
	}
}

// Thresholds after a launch in which long queries were counted by several workgroups (kwage.cpp:388 again: the
// single-workgroup queries have written the same value already).
__global__ __launch_bounds__(256) void kmer_finish_kernel(KmerArgs a, uint32_t n_queries)
{
	const uint32_t q = blockIdx.x*blockDim.x + threadIdx.x;
	if(q < n_queries){ a.qthr[q] = a.complete_match ? 0u : (uint32_t)__fmul_rn(a.threshold, (float)a.nkmer[q]); }
}

// Sparse groups (kwage_group_create_sparse): the matrix holds only the rows listed in `map` (ascending).  Translate
// every VALID row index of the batch (the first nkmer[q]*num_hash entries of query q) into its position in the list;
// an index that is not listed is a caller error and is counted in *missing.  wgs_per_query workgroups per query.
__global__ __launch_bounds__(256) void remap_rows_kernel(uint32_t *rows, const uint64_t *pos_off, const uint32_t *nkmer, uint32_t num_hash,
                                                         const uint32_t *map, uint32_t map_len, unsigned long long *missing, uint32_t wgs_per_query)
{
	const uint32_t q = blockIdx.x / wgs_per_query, part = blockIdx.x % wgs_per_query;      // long queries are shared by several workgroups
	uint32_t *rq = rows + pos_off[q]*num_hash;
	const uint64_t n = (uint64_t)nkmer[q]*num_hash;
	for(uint64_t e = (uint64_t)part*blockDim.x + threadIdx.x; e < n; e += (uint64_t)wgs_per_query*blockDim.x){
		const uint32_t r = rq[e];
		uint32_t lo = 0, hi = map_len;                  // first position with map[pos] >= r
		while(lo < hi){
			const uint32_t mid = lo + (hi - lo)/2;
			if(map[mid] < r){ lo = mid + 1; } else { hi = mid; }
		}
		if(lo < map_len && map[lo] == r){ rq[e] = lo; }
		else{ rq[e] = 0; atomicAdd(missing, 1ull); }
	}
}

// ------------------------------------------------------------------------------------------
// search kernels
// ------------------------------------------------------------------------------------------
struct SearchArgs {
	const uint8_t *db;              // bit matrix
	uint64_t stride;                // bytes between rows (multiple of 128)
	uint32_t units_per_row;         // stride / 16
	const uint8_t *valid;           // stride bytes: 1 bits = real columns
	const uint32_t *rows;           // row indices from kmer_kernel
	const uint64_t *pos_off;
	const uint32_t *nkmer;
	const uint32_t *qthr;
	uint32_t num_hash;
	uint32_t chunks;                // tiles per (query, segment) = ceil(units_per_row / (64*VEC))
	uint32_t n_queries;
	kwage_hit *hits;
	unsigned long long cap;
	unsigned long long *hit_count;
	int early_exit;
	// long queries: the k-mer list of a query is cut into `segs` segments of `seg_kmers` k-mers, one
	// wave per (query, segment, tile); partial results meet in `partial` and a combine kernel
	// extracts the hits.  segs == 1: everything happens in one kernel, `partial` is unused.
	uint32_t segs;
	uint32_t seg_kmers;
	uint32_t *partial;              // AND: u32 [query][units*4] masks; count: u32x4 [query][seg][plane][unit]
	// added to every reported column: a host that appends the hits of several groups / shards to ONE list gives each
	// its own range of global column numbers (kwage_search_device_append_submit)
	uint32_t col_base;
	// The run table (null: not kept -- caller-owned device lists are not ordered).  Every reservation of hit slots -- one
	// wave's, or one workgroup's in the narrow kernels -- is a RUN: its records lie together in the list, ascending by
	// (query, column), and the runs of a search have disjoint, totally ordered key ranges.  The reserving lane notes
	// `first slot << 16 | records` under the run's number, which ascends with the keys (query-major: runs_per_query entries
	// per query; the narrow kernels number their workgroups), so ordering a long list is a prefix sum over the table and
	// one copy of every run to its place -- no sort (hit_sort.hip).  Zeroed before the gather stage: most runs are empty.
	unsigned long long *runs;
	uint32_t runs_per_query;
};

// 16 bytes of columns as a clang vector: bitwise operators apply lane-wise, and the nontemporal
// load builtin accepts it.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ u32x4 load16(const u32x4 *p)
{
	if(NT){ return __builtin_nontemporal_load(p); }
	return *p;
}

// Reserve room for `cnt` hit records of this lane: ONE atomic per wave (an inclusive scan over the lanes
// gives each lane its offset), instead of one same-address atomic per hit.  Must be called by every
// active lane of the wave the same number of times (lanes without hits pass 0).
__device__ __forceinline__ unsigned long long reserve_hits(const SearchArgs &a, uint32_t cnt, uint64_t run)
{
	if(!__any(cnt != 0)){ return 0; }            // the rule: a (query, tile) without a single hit -- no scan, no atomic
	uint32_t incl = cnt;
#pragma unroll
	for(int d = 1; d < WAVE; d <<= 1){
		const uint32_t up = __shfl_up(incl, d);
		if((int)(threadIdx.x & (WAVE - 1)) >= d){ incl += up; }
	}
	const uint32_t total = __shfl(incl, WAVE - 1);
	unsigned long long base = 0;
	if(total){
		if((threadIdx.x & (WAVE - 1)) == WAVE - 1){
			base = atomicAdd(a.hit_count, (unsigned long long)total);
			if(a.runs){ a.runs[run] = (base << 16) | total; }           // (a wave reserves at most 64 x 128 slots)
		}
		base = __shfl(base, WAVE - 1);
	}
	return base + (incl - cnt);
}

// The same for a whole workgroup of `nw` waves that ALL reach this point once: the wave totals meet in LDS and one lane
// does the atomic for all of them.  The narrow kernels end with it: their waves are short (a few dozen rows each) and
// most of them carry a hit, so per-wave atomics on the one counter were 10 % of the kernel (100k x 150 bp reads
// against one 2048-column file: 0.513 ms with hits, 0.466 ms without).
struct WgHitScratch { uint32_t total[SEARCH_THREADS/WAVE]; unsigned long long base; };

__device__ __forceinline__ unsigned long long reserve_hits_wg(const SearchArgs &a, uint32_t cnt, WgHitScratch *sc, uint64_t run)
{
	const uint32_t lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6, nw = blockDim.x >> 6;
	uint32_t incl = cnt;
#pragma unroll
	for(int d = 1; d < WAVE; d <<= 1){
		const uint32_t up = __shfl_up(incl, d);
		if((int)lane >= d){ incl += up; }
	}
	if(lane == WAVE - 1){ sc->total[w] = incl; }
	__syncthreads();
	if(threadIdx.x == 0){
		uint32_t sum = 0;
		for(uint32_t v = 0; v < nw; ++v){ sum += sc->total[v]; }
		sc->base = sum ? atomicAdd(a.hit_count, (unsigned long long)sum) : 0ull;
		if(a.runs && sum){ a.runs[run] = (sc->base << 16) | sum; }       // (four waves: at most 32768 slots)
	}
	__syncthreads();
	unsigned long long base = sc->base;
	for(uint32_t v = 0; v < w; ++v){ base += sc->total[v]; }
	return base + (incl - cnt);
}

__device__ __forceinline__ void store_hit(const SearchArgs &a, unsigned long long slot, uint32_t q, uint32_t col, uint32_t nm)
{
	if(slot < a.cap){
		kwage_hit h; h.query = q; h.column = col + a.col_base; h.num_match = nm;
		a.hits[slot] = h;
	}
}

// ---- hit records of a PERSISTENT wave: collected in LDS, reserved and stored once -------------------------------------
// A reservation is a returning atomic on the ONE hit counter.  A tiled kernel's wave makes it at the end of its life and
// nobody waits behind it; a persistent wave streams rows for the whole launch, its later loads return in order behind the
// atomic, and while that queues at the counter's address the wave streams nothing: with hits in half the queries the walk
// form lost 14 % on 150-base reads and 3 % on C2 (profiles/r04_walk_hit_cost.txt -- the atomic, not the record stores, not
// the scans).  So the persistent kernels keep a wave's records in LDS -- one buffer per wave, WBUF_RECS records and
// WBUF_RUNS runs -- and flush them with ONE reservation when the wave's share is done (or the buffer is full; a single
// round with more records than the buffer holds goes the direct way).  The list stays dense and every run keeps its entry
// in the run table, so nothing downstream changes.
static constexpr uint32_t WBUF_RECS = 256, WBUF_RUNS = 64;
// (a run table entry is `first slot << 16 | records`: what ONE reservation can hold must fit 16 bits -- a wave's 64 lanes x
// 128 columns, a narrow kernel's workgroup of SEARCH_THREADS lanes x 128 columns, a wave's LDS buffer)
static_assert(SEARCH_THREADS*128 <= 0xFFFF, "a workgroup's reservation must fit the run table's 16-bit record field");
static_assert(WAVE*128 <= 0xFFFF, "a wave's reservation must fit the run table's 16-bit record field");
static_assert(WBUF_RECS <= 0xFFFF, "a wave's buffered records must fit the run table's 16-bit record field");
struct WaveHitBuf {
	kwage_hit rec[WBUF_RECS];
	unsigned long long run_id[WBUF_RUNS];
	uint32_t run_at[WBUF_RUNS], run_n[WBUF_RUNS];
};
struct WaveHitState { uint32_t n_rec = 0, n_run = 0; };        // wave-uniform

__device__ __forceinline__ void wave_hits_flush(const SearchArgs &a, WaveHitBuf *buf, WaveHitState &st)
{
	if(st.n_rec == 0){ return; }
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the lanes' LDS writes before other lanes read them
	__builtin_amdgcn_wave_barrier();
	unsigned long long base = 0;
	if(lane == WAVE - 1){ base = atomicAdd(a.hit_count, (unsigned long long)st.n_rec); }
	base = __shfl(base, WAVE - 1);
	for(uint32_t i = lane; i < st.n_rec; i += WAVE){
		if(base + i < a.cap){ a.hits[base + i] = buf->rec[i]; }
	}
	if(a.runs){
		for(uint32_t k = lane; k < st.n_run; k += WAVE){ a.runs[buf->run_id[k]] = ((base + buf->run_at[k]) << 16) | buf->run_n[k]; }
	}
	__builtin_amdgcn_wave_barrier();
	st.n_rec = 0;
	st.n_run = 0;
}

// Room for `total` records of one run in the wave's buffer -> this lane's first place in it (lanes in order, `cnt` each);
// returns false when the run does not fit the buffer at all (the caller then reserves and stores directly).
__device__ __forceinline__ bool wave_hits_place(const SearchArgs &a, WaveHitBuf *buf, WaveHitState &st, uint32_t cnt, uint64_t run, uint32_t &at)
{
	uint32_t incl = cnt;
#pragma unroll
	for(int d = 1; d < WAVE; d <<= 1){
		const uint32_t up = __shfl_up(incl, d);
		if((int)(threadIdx.x & (WAVE - 1)) >= d){ incl += up; }
	}
	const uint32_t total = __builtin_amdgcn_readfirstlane(__shfl(incl, WAVE - 1));
	if(total > WBUF_RECS){ wave_hits_flush(a, buf, st); return false; }       // (what is buffered goes first: runs reach the list in the wave's order)
	if(st.n_rec + total > WBUF_RECS || st.n_run == WBUF_RUNS){ wave_hits_flush(a, buf, st); }
	at = st.n_rec + (incl - cnt);
	if((threadIdx.x & (WAVE - 1)) == WAVE - 1){
		buf->run_id[st.n_run] = run;
		buf->run_at[st.n_run] = st.n_rec;
		buf->run_n[st.n_run] = total;
	}
	st.n_rec += total;
	st.n_run += 1;
	return true;
}

// hit extraction at threshold == 1 (kwage.cpp:489-499,517-518), restricted to real columns.
// `on` = this lane holds a real tile position; every lane of the wave must call it.
// `run`: the number of this reservation in the run table (SearchArgs::runs), wave-uniform (workgroup-uniform with `wg`).
// The records of a mask that is already restricted to real columns (`m` = 0 in lanes without a tile position).
__device__ __forceinline__ void emit_masked_hits(const SearchArgs &a, uint32_t q, uint32_t unit, u32x4 m, uint32_t n, uint64_t run, WgHitScratch *wg = nullptr)
{
	const uint32_t cnt = __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);
	unsigned long long slot = wg ? reserve_hits_wg(a, cnt, wg, run) : reserve_hits(a, cnt, run);
#pragma unroll
	for(int d = 0; d < 4; ++d){
		uint32_t bits = m[d];
		while(bits){
			const uint32_t b = __ffs(bits) - 1;
			bits &= bits - 1;
			store_hit(a, slot++, q, unit*128u + d*32u + b, n);       // num_match = num_query_kmer
		}
	}
}

// The same through the wave's LDS buffer (persistent kernels).
__device__ __forceinline__ void emit_masked_hits_buffered(const SearchArgs &a, WaveHitBuf *buf, WaveHitState &st, uint32_t q, uint32_t unit, u32x4 m, uint32_t n, uint64_t run)
{
	const uint32_t cnt = __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);
	if(!__any(cnt != 0)){ return; }
	uint32_t at = 0;
	if(!wave_hits_place(a, buf, st, cnt, run, at)){ emit_masked_hits(a, q, unit, m, n, run); return; }
#pragma unroll
	for(int d = 0; d < 4; ++d){
		uint32_t bits = m[d];
		while(bits){
			const uint32_t b = __ffs(bits) - 1;
			bits &= bits - 1;
			kwage_hit h; h.query = q; h.column = unit*128u + d*32u + b + a.col_base; h.num_match = n;
			buf->rec[at++] = h;
		}
	}
}

__device__ __forceinline__ void emit_mask_hits(const SearchArgs &a, uint32_t q, uint32_t unit, u32x4 acc, uint32_t n, uint64_t run, bool on = true, WgHitScratch *wg = nullptr)
{
	u32x4 m = (u32x4)(0u);
	if(on){ m = acc & reinterpret_cast<const u32x4*>(a.valid)[unit]; }
	const uint32_t cnt = __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);
	unsigned long long slot = wg ? reserve_hits_wg(a, cnt, wg, run) : reserve_hits(a, cnt, run);
#pragma unroll
	for(int d = 0; d < 4; ++d){
		uint32_t bits = m[d];
		while(bits){
			const uint32_t b = __ffs(bits) - 1;
			bits &= bits - 1;
			store_hit(a, slot++, q, unit*128u + d*32u + b, n);       // num_match = num_query_kmer
		}
	}
}

// Decompose a tile id into (query, segment, column tile); all wave-uniform.
__device__ __forceinline__ void tile_coords(const SearchArgs &a, uint64_t tile, uint32_t &q, uint32_t &sg, uint32_t &c)
{
	const uint32_t per_q = a.segs*a.chunks;
	q = __builtin_amdgcn_readfirstlane((uint32_t)(tile / per_q));
	const uint32_t rem = __builtin_amdgcn_readfirstlane((uint32_t)(tile % per_q));
	sg = rem / a.chunks;      // adjacent waves = adjacent column tiles of the SAME rows
	c = rem % a.chunks;
}

// threshold == 1.0f: AND of every addressed row (kwage.cpp:404-470).
//   VEC    16-byte vectors per lane per row (tile = 64*VEC*16 bytes of each row per wave)
//   SEG    the query's row list is split over several waves (see SearchArgs::segs)
// Eight rows in flight per wave, nontemporal loads (every row byte is used exactly once per (query, tile): +4-10 %): 4, 16
// and 32 rows and plain loads were template axes until round 5 -- no selection rule ever reached them.
template <int VEC, bool SEG>
__global__ __launch_bounds__(SEARCH_THREADS) void and_kernel(SearchArgs a)
{
	constexpr int UNROLL = 8;
	constexpr bool NT = true;
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint64_t tile = (uint64_t)blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6);
	if(tile >= (uint64_t)a.n_queries*a.segs*a.chunks){ return; }

	// wave-uniform values -> SGPRs, so row indices come through the scalar cache
	uint32_t q, sg, c;
	tile_coords(a, tile, q, sg, c);
	const uint32_t n = a.nkmer[q];
	if(n == 0){ return; }
	uint32_t k0 = 0, k1 = n;
	if(SEG){
		k0 = sg*a.seg_kmers;
		k1 = min(n, k0 + a.seg_kmers);
		if(k0 >= k1){ return; }
	}
	const uint32_t nrows = (k1 - k0)*a.num_hash;
	const uint32_t *rq = a.rows + (a.pos_off[q] + k0)*a.num_hash;

	uint32_t unit[VEC];     // this lane's 16-byte units within a row (clamped in range: no divergence)
	bool live[VEC];
	u32x4 acc[VEC];
#pragma unroll
	for(int v = 0; v < VEC; ++v){
		const uint32_t u = c*(WAVE*VEC) + v*WAVE + lane;
		live[v] = (u < a.units_per_row);
		unit[v] = live[v] ? u : (a.units_per_row - 1);
		acc[v] = ~(u32x4)(0u);                       // set_all_bits, bloom.h:182-187
	}

	uint32_t i = 0;
	bool dead = false;        // early exit taken: the accumulator is all zero and stays so
	for(; i + UNROLL <= nrows; i += UNROLL){
		u32x4 x[UNROLL][VEC];
#pragma unroll
		for(int u = 0; u < UNROLL; ++u){
			const uint32_t r = rq[i + u];
			const u32x4 *p = reinterpret_cast<const u32x4*>(a.db + (uint64_t)r*a.stride);
#pragma unroll
			for(int v = 0; v < VEC; ++v){ x[u][v] = load16<NT>(p + unit[v]); }
		}
#pragma unroll
		for(int u = 0; u < UNROLL; ++u){
#pragma unroll
			for(int v = 0; v < VEC; ++v){ acc[v] &= x[u][v]; }
		}
		if(a.early_exit){     // kwage.cpp:466-470 per tile: nothing left that could match
			bool nz = false;
#pragma unroll
			for(int v = 0; v < VEC; ++v){ nz |= ((acc[v].x | acc[v].y | acc[v].z | acc[v].w) != 0); }
			if(!__any(nz)){ dead = true; break; }
		}
	}
	for(; !dead && i < nrows; ++i){
		const uint32_t r = rq[i];
		const u32x4 *p = reinterpret_cast<const u32x4*>(a.db + (uint64_t)r*a.stride);
#pragma unroll
		for(int v = 0; v < VEC; ++v){ acc[v] &= load16<NT>(p + unit[v]); }
	}

#pragma unroll
	for(int v = 0; v < VEC; ++v){
		if(SEG){
			// meet the other segments of this (query, tile) in the mask buffer (pre-set to all ones)
			if(live[v]){
				uint32_t *m = a.partial + ((uint64_t)q*a.units_per_row + unit[v])*4;
#pragma unroll
				for(int d = 0; d < 4; ++d){
					if(acc[v][d] != ~0u){ atomicAnd(m + d, acc[v][d]); }
				}
			}
		}
		else{
			emit_mask_hits(a, q, unit[v], acc[v], n, (uint64_t)q*a.runs_per_query + c*VEC + v, live[v]);     // every lane takes part in the wave scan
		}
	}
}

// ---- early exit at threshold == 1: SCREEN, then REFINE -------------------------------------------------------------
// kwage.cpp:466-470 stops a query once no column can match any more.  Per (query, 2 KiB tile) that rule kills a tile
// without a matching column after 8-16 random rows -- but a tile that HOLDS a matching column never dies: in and_kernel
// its wave walks all of the query's rows 2 KiB wide, alone, long after its sibling tiles stopped (C2's shape: ~1500 such
// walks of 970 rows on 256 CUs were the whole launch, and 15/16 of the bytes they fetched belonged to columns that had
// been ruled out after 16 rows).  So the work is cut in two:
//   and_screen_kernel   and_kernel's tile loop (accumulator preset with the real-column mask).  After every group of rows:
//                       no column left -> the tile is done (the reference's exit); columns left in at most `max_groups`
//                       128-byte groups of the tile and at least `min_rows` rows to go -> the tile is HANDED OVER: per
//                       KiB-step with survivors one CLUSTER (the unit of hit reservation: one run of the run table), per
//                       128-byte group with survivors one ITEM (its 128 bytes of mask so far), and per item and segment of
//                       `seg_rows` remaining rows one UNIT, all appended to lists in memory; otherwise it goes on.
//                       A tile that reaches the end of its row list reports its hits itself, as and_kernel does.
//   and_refine_kernel   a persistent grid over the unit list, EIGHT units per wave (8 lanes x 16 B = one 128-byte line of
//                       each row): ANDs the unit's rows of that line only, and folds the result into the item's mask with
//                       atomic ANDs -- skipped where the mask would not change (the rule for a true positive), and the
//                       whole unit is skipped when an earlier one has already emptied the mask.
//   and_refine_emit_kernel  one wave per cluster: the surviving columns of its items, one reservation.
// Lists that are full: the tile simply goes on by itself (reserved places are filled with REFINE_NONE entries).
// A unit carries everything the refine launch needs (no dependent loads of item or query records before its rows).
struct RefineUnit { uint32_t item, r0, r1, unit0, rq_lo, rq_hi, q, pad; };      // rows [r0, r1) of the row list at a.rows + rq; unit0: first 16-byte unit of the 128-byte group; item: whose mask
struct RefineCluster {
	uint32_t q, kstep_groups, first_item, n;      // kstep << 8 | bitmap of the KiB-step's 128-byte groups that are items (ascending: first_item, +1, ...); n = num_query_kmer
	uint32_t first_unit, nseg, pad0, pad1;        // count path: item j's units are first_unit + j*nseg ... (their partial counters: RefineArgs::slab)
};
static constexpr uint32_t REFINE_NONE = 0xFFFFFFFFu;

// Three lists: clusters and units are SCANNED by the launches that follow (places nobody filled hold REFINE_NONE), items
// are only places for masks.  Every list has a static part -- `stat` places per wave of the screen launch, taken without
// any atomic -- and, behind it (from `base` = waves x stat), a dynamic part reserved `chunk` places at a time through
// counters[]: hand-overs come in bursts (all tiles reach their eighth row together), and returning atomics on one cache
// line serialise at the memory side at ~25 ns each -- three per hand-over were 0.2 ms of C2's launch and 27 ms of a
// launch of 100 k short reads (300 k hand-overs).
struct RefineList { uint32_t cap, stat, base, chunk; };
struct RefineArgs {
	uint32_t *counters;             // [0] clusters, [1] items, [2] units reserved in the dynamic parts (zeroed before the stage; may run past the capacities); [4..5]: the tile queue of count_screen_kernel (u64)
	RefineCluster *clusters;
	uint32_t *masks;                // AND: u32x4 [item][8 lanes], the item's 128 bytes of mask; count: u32x4 [item][PLANES][8 lanes], its counters so far
	RefineUnit *units;
	uint32_t *slab;                 // count only: u32x4 [unit][UP planes][8 lanes], the counters of the unit's k-mers
	RefineList lc, li, lu;          // clusters, items, units
	uint32_t seg_rows;              // rows (count: k-mers) per unit
	uint32_t min_rows;              // hand a tile over only when at least this many rows (k-mers) are left
	uint32_t max_groups;            // ... and at most this many of its 128-byte groups hold a surviving column
	uint32_t queue_batch;           // count_screen_kernel: tiles a wave draws from the queue at a time
	uint32_t check_every;           // count_screen_kernel: k-mers between two looks at the bound (8, 16, 32 or 64)
};

// one bit per 8 lanes of a ballot: the 128-byte groups of a KiB-step with a lane set
__device__ __forceinline__ uint32_t group_bits(uint64_t lanes)
{
	uint32_t g = 0;
#pragma unroll
	for(int k = 0; k < 8; ++k){ g |= ((lanes >> (8*k)) & 0xFFull) ? (1u << k) : 0u; }
	return g;
}

// position of the j-th set bit of `bits` (j < popcount)
__device__ __forceinline__ uint32_t nth_set_bit(uint32_t bits, uint32_t j)
{
	for(uint32_t k = 0; k < j; ++k){ bits &= bits - 1; }
	return __ffs(bits) - 1;
}

struct RefineChunk { uint32_t next = 0, end = 0; };            // a wave's private stretch of a list (wave-uniform)

__device__ __forceinline__ void refine_none_clusters(const RefineArgs &ra, uint32_t from, uint32_t to)
{
	to = min(to, ra.lc.cap);
	for(uint32_t i = from + (threadIdx.x & (WAVE - 1)); i < to; i += WAVE){ ra.clusters[i].q = REFINE_NONE; }
}
__device__ __forceinline__ void refine_none_units(const RefineArgs &ra, uint32_t from, uint32_t to)
{
	to = min(to, ra.lu.cap);
	for(uint32_t i = from + (threadIdx.x & (WAVE - 1)); i < to; i += WAVE){ ra.units[i].item = REFINE_NONE; }
}

// `need` consecutive places of list `which` (0 clusters, 1 items, 2 units) from the wave's chunk -> first place, or
// REFINE_NONE when the list is full.  A chunk that is too short is closed (NONE entries where the list is scanned) and a
// new one reserved in the list's dynamic part: `chunk` places, or `need` if that is more (the units of one long query).
// Wave-uniform; every lane calls.
__device__ __forceinline__ uint32_t refine_take(const RefineArgs &ra, int which, RefineChunk &ch, uint32_t need)
{
	const RefineList &ls = (which == 0) ? ra.lc : (which == 1) ? ra.li : ra.lu;
	if(ch.end - ch.next < need){
		if(which == 0){ refine_none_clusters(ra, ch.next, ch.end); } else if(which == 2){ refine_none_units(ra, ch.next, ch.end); }
		ch.next = ch.end = 0;
		if((uint64_t)ls.base + __hip_atomic_load(ra.counters + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ls.cap){ return REFINE_NONE; }      // (full: no more atomics on it)
		const uint32_t want = max(need, ls.chunk);
		uint32_t r = 0;
		if((threadIdx.x & (WAVE - 1)) == 0){ r = atomicAdd(ra.counters + which, want); }
		const uint64_t at = (uint64_t)ls.base + __builtin_amdgcn_readfirstlane(r);
		if(at + want > ls.cap){                 // the tail of the list: too short for this wave; close it
			if(at < ls.cap){
				if(which == 0){ refine_none_clusters(ra, (uint32_t)at, ls.cap); } else if(which == 2){ refine_none_units(ra, (uint32_t)at, ls.cap); }
			}
			return REFINE_NONE;
		}
		ch.next = (uint32_t)at;
		ch.end = (uint32_t)at + want;
	}
	const uint32_t at = ch.next;
	ch.next += need;
	return at;
}

// how far the launches that follow have to scan a list
__device__ __forceinline__ uint32_t refine_list_end(const RefineArgs &ra, int which)
{
	const RefineList &ls = (which == 0) ? ra.lc : (which == 1) ? ra.li : ra.lu;
	return (uint32_t)min((uint64_t)ls.base + ra.counters[which], (uint64_t)ls.cap);
}

// (a PERSISTENT grid: tile t of the batch -- (query, 64*VEC*16 bytes of every row), query-major -- goes to wave t mod waves,
// so the waves of a workgroup screen adjacent tiles of the same rows at the same time)
template <int VEC, int UNROLL>
__global__ __launch_bounds__(SEARCH_THREADS) void and_screen_kernel(SearchArgs a, RefineArgs ra)
{
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint64_t n_tiles = (uint64_t)a.n_queries*a.chunks;
	const uint32_t n_waves = gridDim.x*(blockDim.x/WAVE);
	const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6));
	__shared__ WaveHitBuf hit_bufs[SEARCH_THREADS/WAVE];           // (a persistent wave reserves its hit records once: WaveHitBuf)
	WaveHitBuf *hbuf = &hit_bufs[threadIdx.x >> 6];
	WaveHitState hst;
	RefineChunk cc, ci, cu;               // the wave's static places first
	cc.next = gw*ra.lc.stat; cc.end = cc.next + ra.lc.stat;
	ci.next = gw*ra.li.stat; ci.end = ci.next + ra.li.stat;
	cu.next = gw*ra.lu.stat; cu.end = cu.next + ra.lu.stat;
	bool lists_full = false;
	for(uint64_t tile = gw; tile < n_tiles; tile += n_waves){
		const uint32_t q = __builtin_amdgcn_readfirstlane((uint32_t)(tile / a.chunks));
		const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)(tile % a.chunks));
		const uint32_t n = a.nkmer[q];
		if(n == 0){ continue; }
		const uint32_t nrows = n*a.num_hash;
		const uint64_t rq_off = a.pos_off[q]*a.num_hash;
		const uint32_t *rq = a.rows + rq_off;

		uint32_t unit[VEC];
		u32x4 acc[VEC];
#pragma unroll
		for(int v = 0; v < VEC; ++v){
			const uint32_t u = c*(WAVE*VEC) + v*WAVE + lane;
			const bool live = (u < a.units_per_row);
			unit[v] = live ? u : (a.units_per_row - 1);
			// only real columns can survive: pad bits between file blocks (whatever they hold) never keep a tile alive
			acc[v] = live ? reinterpret_cast<const u32x4*>(a.valid)[u] : (u32x4)(0u);
		}

		bool may_hand_over = !lists_full, gone = false;
		uint32_t i = 0;
		for(; i + UNROLL <= nrows; i += UNROLL){
			u32x4 x[UNROLL][VEC];
#pragma unroll
			for(int u = 0; u < UNROLL; ++u){
				const uint32_t r = rq[i + u];
				const u32x4 *p = reinterpret_cast<const u32x4*>(a.db + (uint64_t)r*a.stride);
#pragma unroll
				for(int v = 0; v < VEC; ++v){ x[u][v] = load16<true>(p + unit[v]); }
			}
#pragma unroll
			for(int u = 0; u < UNROLL; ++u){
#pragma unroll
				for(int v = 0; v < VEC; ++v){ acc[v] &= x[u][v]; }
			}
			uint32_t gb[VEC], ngroups = 0;
#pragma unroll
			for(int v = 0; v < VEC; ++v){
				gb[v] = group_bits(__ballot((acc[v].x | acc[v].y | acc[v].z | acc[v].w) != 0));
				ngroups += __popc(gb[v]);
			}
			if(ngroups == 0){ gone = true; break; }       // kwage.cpp:466-470 per tile: nothing left that could match
			const uint32_t done = i + UNROLL;
			// (a long row list waits for its SECOND group of rows: after eight random rows a quarter of all tiles still holds a
			// chance survivor, after sixteen one in fifteen -- and every item of a long list costs a dozen units of list space)
			if(may_hand_over && ngroups <= ra.max_groups && nrows - done >= ra.min_rows && (done >= 2*UNROLL || nrows < 4*ra.seg_rows)){
				may_hand_over = false;                    // (one attempt per tile)
				const uint32_t nseg = (nrows - done + ra.seg_rows - 1)/ra.seg_rows;
				uint32_t nc = 0;
#pragma unroll
				for(int v = 0; v < VEC; ++v){ nc += gb[v] ? 1u : 0u; }
				const uint32_t nu = ngroups*nseg;             // (ngroups <= 16; the host keeps nseg <= 2^16)
				const uint32_t c0 = refine_take(ra, 0, cc, nc);
				const uint32_t i0 = (c0 != REFINE_NONE) ? refine_take(ra, 1, ci, ngroups) : REFINE_NONE;
				const uint32_t u0 = (i0 != REFINE_NONE) ? refine_take(ra, 2, cu, nu) : REFINE_NONE;
				if(u0 == REFINE_NONE){
					// some list is full: the cluster places taken stay empty, and the tile walks on by itself
					if(c0 != REFINE_NONE){ refine_none_clusters(ra, c0, c0 + nc); }
					lists_full = true;
					continue;
				}
				uint32_t cidx = c0, ii = i0, all_groups = 0;
#pragma unroll
				for(int v = 0; v < VEC; ++v){
					all_groups |= gb[v] << (8*v);
					if(!gb[v]){ continue; }
					if(lane == 0){
						RefineCluster cl; cl.q = q; cl.kstep_groups = ((c*VEC + v) << 8) | gb[v]; cl.first_item = ii; cl.n = n;
						cl.first_unit = 0; cl.nseg = 0; cl.pad0 = 0; cl.pad1 = 0;
						ra.clusters[cidx] = cl;
					}
					const uint32_t k = lane >> 3;                                  // this lane's 128-byte group of the KiB-step
					if((gb[v] >> k) & 1u){
						const uint32_t it = ii + __popc(gb[v] & ((1u << k) - 1u));
						reinterpret_cast<u32x4*>(ra.masks)[(uint64_t)it*8 + (lane & 7u)] = acc[v];
					}
					++cidx;
					ii += __popc(gb[v]);
				}
				for(uint32_t e = lane; e < nu; e += WAVE){
					const uint32_t j = e / nseg, sg = e % nseg;                     // item j of the tile (ascending by column), segment sg
					const uint32_t gpos = nth_set_bit(all_groups, j);               // bit v*8 + k
					RefineUnit un;
					un.item = i0 + j;
					un.r0 = done + sg*ra.seg_rows;
					un.r1 = min(nrows, un.r0 + ra.seg_rows);
					un.unit0 = c*(WAVE*VEC) + gpos*8u;                             // (= v*64 + k*8 within the tile)
					un.rq_lo = (uint32_t)rq_off; un.rq_hi = (uint32_t)(rq_off >> 32);
					un.q = q; un.pad = 0;
					ra.units[u0 + e] = un;
				}
				gone = true;                              // the refine launch takes it from here
				break;
			}
		}
		if(gone){ continue; }
		for(; i < nrows; ++i){
			const uint32_t r = rq[i];
			const u32x4 *p = reinterpret_cast<const u32x4*>(a.db + (uint64_t)r*a.stride);
#pragma unroll
			for(int v = 0; v < VEC; ++v){ acc[v] &= load16<true>(p + unit[v]); }
		}
#pragma unroll
		for(int v = 0; v < VEC; ++v){
			emit_masked_hits_buffered(a, hbuf, hst, q, unit[v], acc[v], n, (uint64_t)q*a.runs_per_query + c*VEC + v);      // (acc is 0 in lanes past the row end)
		}
	}
	wave_hits_flush(a, hbuf, hst);
	// what the wave did not use of its places
	refine_none_clusters(ra, cc.next, cc.end);
	refine_none_units(ra, cu.next, cu.end);
}

// Eight units per wave: lane group g (8 lanes x 16 B = the unit's 128-byte line of every row) walks unit base + g.  The
// unit's row numbers are fetched 64 at a time, eight per lane, before the rows (a row number per lane and row would put
// a second dependent load in front of every group of rows); a lane's neighbours hand them round (ds_bpermute).
template <int UNROLL>
__global__ __launch_bounds__(SEARCH_THREADS) void and_refine_kernel(SearchArgs a, RefineArgs ra)
{
	static_assert(UNROLL == 8 || UNROLL == 16, "eight or sixteen rows in flight");
	const uint32_t lane = threadIdx.x & (WAVE - 1), l = lane & 7u, sh = lane & ~7u;
	const uint32_t n_units = refine_list_end(ra, 2);
	const uint64_t gw = (uint64_t)blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6);
	const uint64_t step = (uint64_t)gridDim.x*(blockDim.x/WAVE)*8;
	for(uint64_t base = gw*8; base < n_units; base += step){
		const uint64_t u = base + (lane >> 3);
		RefineUnit un;
		un.item = REFINE_NONE;
		if(u < n_units){ un = ra.units[u]; }
		bool on = (un.item != REFINE_NONE);
		uint32_t *M = ra.masks;
		u32x4 m = (u32x4)(0u);
		if(on){
			M = ra.masks + ((uint64_t)un.item*8 + l)*4;
#pragma unroll
			for(int d = 0; d < 4; ++d){ m[d] = __hip_atomic_load(M + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		}
		// an earlier unit of the item has emptied the mask: nothing this one finds can matter
		on = on && (((uint32_t)(__ballot((m.x | m.y | m.z | m.w) != 0) >> sh) & 0xFFu) != 0);
		if(on){
			const uint32_t *rq = a.rows + (((uint64_t)un.rq_hi << 32) | un.rq_lo);
			const u32x4 *col = reinterpret_cast<const u32x4*>(a.db) + (un.unit0 + l);
			u32x4 acc = ~(u32x4)(0u);
			bool alive = true;
			for(uint32_t b = un.r0; alive && b < un.r1; b += 64){
				uint32_t idx[8];           // lane l: rows b + l, b + 8 + l, ... (past the end: the last row again -- AND is idempotent)
#pragma unroll
				for(int k = 0; k < 8; ++k){ idx[k] = rq[min(b + 8u*k + l, un.r1 - 1)]; }
#pragma unroll
				for(int k = 0; k < 8; k += UNROLL/8){
					if(b + 8u*k >= un.r1){ break; }
					u32x4 x[UNROLL];
#pragma unroll
					for(int j = 0; j < UNROLL; ++j){
						const uint32_t r = __shfl(idx[k + j/8], sh + (j & 7));
						x[j] = load16<true>(reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(col) + (uint64_t)r*a.stride));
					}
#pragma unroll
					for(int j = 0; j < UNROLL; ++j){ acc &= x[j]; }
					if((((uint32_t)(__ballot((acc.x | acc.y | acc.z | acc.w) != 0) >> sh)) & 0xFFu) == 0){ alive = false; break; }      // the group's 128 bytes are empty
				}
			}
			// fold into the item's mask; where it would not change (m is a subset of acc: the rule for a column that
			// really matches, and masks only ever lose bits) there is nothing to do
#pragma unroll
			for(int d = 0; d < 4; ++d){
				if(m[d] & ~acc[d]){ __hip_atomic_fetch_and(M + d, acc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
			}
		}
	}
}

// (a wave takes every nw-th cluster, four at a time: the four records, then the four sets of masks are requested together;
// the records found go through the wave's LDS buffer and are reserved ONCE per wave -- 150 k clusters with a hit each,
// reserved one by one, were 1.8 ms of returning atomics on the one hit counter)
__global__ __launch_bounds__(SEARCH_THREADS) void and_refine_emit_kernel(SearchArgs a, RefineArgs ra)
{
	constexpr int B = 4;
	__shared__ WaveHitBuf hit_bufs[SEARCH_THREADS/WAVE];
	WaveHitBuf *hbuf = &hit_bufs[threadIdx.x >> 6];
	WaveHitState hst;
	const uint32_t lane = threadIdx.x & (WAVE - 1), l = lane & 7u, g = lane >> 3;
	const uint32_t n_clusters = refine_list_end(ra, 0);
	const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6));
	const uint32_t nw = gridDim.x*(blockDim.x/WAVE);
	for(uint64_t c0 = gw; c0 < n_clusters; c0 += (uint64_t)B*nw){
		RefineCluster cl[B];
		u32x4 m[B];
#pragma unroll
		for(int k = 0; k < B; ++k){
			const uint64_t ci = c0 + (uint64_t)k*nw;
			cl[k].q = REFINE_NONE;
			if(ci < n_clusters){ cl[k] = ra.clusters[ci]; }
		}
#pragma unroll
		for(int k = 0; k < B; ++k){
			const uint32_t groups = cl[k].kstep_groups & 0xFFu;
			m[k] = (u32x4)(0u);
			if(cl[k].q != REFINE_NONE && ((groups >> g) & 1u)){          // this lane's 128-byte group of the KiB-step is an item
				const uint32_t it = cl[k].first_item + __popc(groups & ((1u << g) - 1u));
				m[k] = reinterpret_cast<const u32x4*>(ra.masks)[(uint64_t)it*8 + l];
			}
		}
#pragma unroll
		for(int k = 0; k < B; ++k){
			if(cl[k].q == REFINE_NONE){ continue; }
			const uint32_t kstep = cl[k].kstep_groups >> 8;
			emit_masked_hits_buffered(a, hbuf, hst, cl[k].q, kstep*WAVE + lane, m[k], cl[k].n, (uint64_t)cl[k].q*a.runs_per_query + kstep);
		}
	}
	wave_hits_flush(a, hbuf, hst);
}

// threshold == 1.0f, "walk" form of the gather + AND for rows of 3..16 KiB: a wave walks each of its rows over
// the WHOLE width of a column tile (CH KiB, CH accumulators per lane), UNROLL rows at a time, so a row's
// consecutive KiB are requested back to back by one wave instead of by different waves at different times as in
// and_kernel's (query, 2 KiB tile) form (+1-2.4 % on 12.5 KB rows, nothing on 125 KB rows).
//
// The grid is PERSISTENT and the work statically balanced: the launch has 8 waves per CU (2 workgroups; 16 per CU
// measured 1-2 % slower -- twice the cut pairs -- and a deeper prefetch no better), fewer for small batches, and the batch's concatenated position list -- `slots`: for every
// query and column tile its positions, query-major -- is cut into equal contiguous ranges, one per wave.  Row
// lists are addressed by position (pos_off) and trimmed to the distinct k-mers the k-mer stage found (nkmer),
// so every wave reads the same number of bytes unless a query lost many positions to duplicates or N.  There is
// no "last round": 1030 queries cost 1.03x what 1000 do.
//   - a (query, tile) pair that lies inside one wave's range is reduced in registers and emitted at once;
//   - a pair cut by a range boundary is finished through memory: every wave that holds a part folds its partial
//     mask into the pair's slot (slot = the wave whose range holds the pair's first position, so no two cut
//     pairs share one) -- an all-zero mask by raising a flag, anything else by ORing its complement into
//     `orbuf` -- and adds its k-mer count to the slot's counter; the wave whose add completes the pair's nkmer
//     reads the result back, emits, and leaves slot, flags and counter ZERO again: the buffers are cleared
//     once when they are allocated, never per search.
static constexpr int WALK_WG_WAVES = 8;        // waves per workgroup (= per CU) of a chip-filling launch of the persistent kernels
static constexpr uint32_t WALK_DEAD = 1u;      // some part of the pair has an all-zero mask: nothing to report
static constexpr uint32_t WALK_DIRTY = 2u;     // some part ORed its mask into the slot: clear it when the pair is done

struct WalkArgs {
	uint64_t total_slots;           // column tiles per row x positions of the batch
	uint64_t per_wave;              // slots per wave: ceil(total_slots / waves of the launch)
	uint32_t coltiles;              // column tiles per row (balanced, each <= 16 KiB)
	uint32_t *orbuf;                // [waves][CH][4][64] complemented partial masks of the cut pairs
	uint32_t *done;                 // [waves][2]: k-mers folded into the slot so far, and its WALK_DEAD / WALK_DIRTY flags
};

// (rows, pos_off and nkmer are passed as __restrict__ parameters of their own besides SearchArgs: the kernel stores
// hits and updates the cut-pair slots inside its work loop, and only with the no-alias promise does the compiler
// keep the row-index loads on the scalar path -- otherwise every row descriptor goes through a waterfall loop)
// (launched with workgroups of WALK_WG_WAVES waves and a dynamic-LDS pad of more than half a CU's LDS when the launch
// fills the chip: ONE workgroup per CU, so every CU runs exactly the same number of waves -- with workgroups of four
// waves the dispatcher's placement left some CUs with three workgroups and others with one: +0.8-0.9 % at C2's shape,
// profiles/r03_walk_cu_shapes.txt)
// (amdgpu_waves_per_eu(2, 2): the launch holds 8 waves per CU -- two per SIMD -- by construction, and the compiler must
// know: left to aim for the occupancy its register count would allow, it serialised the UNROLL loads of a step into pairs
// for CH = 4..7 (54 instead of 81 VGPRs, two rows in flight instead of four) once the early-exit test had left the loop in
// round 5 -- C2's columns split 2 and 4 ways lost 18-22 %, profiles/r05_walk_occupancy_hint_ab.txt)
template <int CH, int UNROLL>
__global__ __launch_bounds__(WALK_WG_WAVES*WAVE) __attribute__((amdgpu_waves_per_eu(2, 2))) void and_walk_kernel(SearchArgs a, WalkArgs wa, const uint32_t *__restrict__ rows,
                                                                      const uint64_t *__restrict__ pos_off, const uint32_t *__restrict__ nkmer)
{
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6));
	uint64_t s = (uint64_t)gw*wa.per_wave;
	const uint64_t s1 = min(wa.total_slots, s + wa.per_wave);
	if(s >= s1){ return; }
	__shared__ WaveHitBuf hit_bufs[WALK_WG_WAVES];                // (static: beside the dynamic pad of a chip-filling launch)
	WaveHitBuf *hbuf = &hit_bufs[threadIdx.x >> 6];
	WaveHitState hst;
	const uint32_t ct = wa.coltiles;
	const uint32_t row_bytes = a.units_per_row*16u;
	const uint32_t umax = a.units_per_row - 1;

	// the query that holds slot s: the largest q with ct*pos_off[q] <= s (pos_off[n_queries]*ct = total_slots > s)
	uint32_t q = 0;
	{
		uint32_t hi = a.n_queries;
		while(hi - q > 1){
			const uint32_t mid = q + (hi - q)/2;
			if((uint64_t)ct*pos_off[mid] <= s){ q = mid; } else { hi = mid; }
		}
	}

	while(s < s1){
		const uint64_t p0 = pos_off[q];
		const uint64_t npos = pos_off[q + 1] - p0;
		if(npos == 0){ ++q; continue; }                      // a query shorter than k: no slots (q stays in range: s < total_slots)
		const uint64_t rem = s - (uint64_t)ct*p0;
		const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)(rem / npos));
		const uint32_t j0 = __builtin_amdgcn_readfirstlane((uint32_t)(rem % npos));
		const uint32_t take = (uint32_t)min(npos - j0, s1 - s);
		const uint32_t j1 = j0 + take;
		const uint32_t n = nkmer[q];
		const uint32_t jv1 = min(j1, n);                     // positions past the distinct k-mers hold no rows
		if(j0 < jv1){
			const uint32_t nrows = (jv1 - j0)*a.num_hash;
			const uint32_t *rq = rows + (p0 + j0)*a.num_hash;
			// Rows are read through buffer descriptors (base = the row, num_records = the row's bytes): the address of
			// chunk j is "descriptor + lane offset + scalar j KiB" with no per-chunk vector arithmetic, and lanes past
			// the row end (last chunk(s) of the last column tile) are bounds-checked by the hardware -- they return 0
			// without touching memory, and are never reported.
			const uint32_t u0 = c*CH*WAVE + lane;
			// (CH = 8: eight accumulators are 32 dwords -- exactly the widest register tuple -- and the compiler promoted the array
			// to ONE 1024-bit value whose elements the cut-pair path inserts one by one: 3.2 KB of spills per lane, in the ISA since
			// round 3 and found by tools/isa_check.py in round 5.  One element more and it stays eight separate vectors.)
			u32x4 acc[CH == 8 ? CH + 1 : CH];
#pragma unroll
			for(int j = 0; j < CH; ++j){ acc[j] = ~(u32x4)(0u); }
			for(uint32_t i = 0; i < nrows; i += UNROLL){
				__amdgpu_buffer_rsrc_t rs[UNROLL];
#pragma unroll
				for(int u = 0; u < UNROLL; ++u){
					const uint32_t r = rq[min(i + u, nrows - 1)];        // past the end: the last row again (AND is idempotent)
					rs[u] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.db + (uint64_t)r*a.stride), 0, row_bytes, 0x00020000);
				}
				// (the opposite extreme, all UNROLL x CH KiB requested before the first wait -- 172 VGPRs, 2 waves/SIMD -- measured
				// the same or 1 % slower at every batch size: removed)
#pragma unroll
				for(int j = 0; j < CH; ++j){
					u32x4 x[UNROLL];
#pragma unroll
					for(int u = 0; u < UNROLL; ++u){ x[u] = __builtin_amdgcn_raw_buffer_load_b128(rs[u], u0*16u, j*1024, 2 /* nt */); }
#pragma unroll
					for(int u = 0; u < UNROLL; ++u){ acc[j] &= x[u]; }
					// One KiB-step of UNROLL rows at a time: nothing of step j+1 is requested before step j is in.  What the
					// memory system likes is ~8192 sequential row streams chip-wide (8 waves per CU x 4 rows), each a KiB
					// deep; left alone the scheduler keeps 8-9 KiB per wave in flight, which measures 1 % slower, and more
					// streams (16 waves per CU, or 8 rows) 2-3 % slower (profiles/r02_walk_sizes_schedules.txt).
					// (all CH steps of eight rows of narrow column tiles requested together -- the tiled kernel's "wide" access pattern on
					// the balanced persistent grid -- measured no better in round 4 and was removed in round 5)
					__builtin_amdgcn_sched_barrier(0);
				}
			}

			bool emit = true;
			if(j0 != 0 || jv1 != n){
				// A cut pair: fold this part into the pair's slot; the part that completes the pair emits.
				// A part whose mask is all zero (the rule after a few dozen random rows, unless a column really matches)
				// decides the pair -- nothing can be reported -- so it only raises the slot's DEAD flag; only parts with
				// surviving columns pay for ORing CH KiB of complemented mask into the slot (and raise DIRTY, so that
				// whoever finishes the pair knows the slot has to be cleared).
				const uint32_t slot = (uint32_t)(((uint64_t)ct*p0 + (uint64_t)c*npos)/wa.per_wave);
				uint32_t *ob = wa.orbuf + (uint64_t)slot*(CH*4*WAVE) + lane;
				uint32_t *state = wa.done + 2*(uint64_t)slot;                // [0] k-mers folded so far, [1] WALK_DEAD | WALK_DIRTY
				bool nz = false;
#pragma unroll
				for(int j = 0; j < CH; ++j){ nz |= ((acc[j].x | acc[j].y | acc[j].z | acc[j].w) != 0); }
				const bool live = __any(nz);
				if(live){
#pragma unroll
					for(int j = 0; j < CH; ++j){
#pragma unroll
						for(int d = 0; d < 4; ++d){
							const uint32_t v = ~acc[j][d];
							if(v){ __hip_atomic_fetch_or(ob + (j*4 + d)*WAVE, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
						}
					}
				}
				if(lane == 0){ __hip_atomic_fetch_or(state + 1, live ? WALK_DIRTY : WALK_DEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
				// The ORs and the flag must be performed before the count says so.  Everything this protocol exchanges
				// goes through device-scope atomics, which are performed at the device's point of coherence and
				// acknowledged from there, so waiting for the acknowledgements (vmcnt) orders them.  A C++ release fence
				// would do it too but also writes the L2 back (buffer_wbl2 sc1) once per cut pair, and the waves all
				// reach this point together at the end of the kernel: +0.14 ms per launch, measured in round 2
				// (profiles/r02_walk_sizes_fences.txt; 62 000 soaked launches without a mismatch: profiles/r05_soak.txt).
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				uint32_t old = 0;
				if(lane == 0){ old = __hip_atomic_fetch_add(state, jv1 - j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
				old = __builtin_amdgcn_readfirstlane(old);
				emit = false;
				if(old + (jv1 - j0) == n){                                       // this part completes the pair
					const uint32_t fl = __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if(fl & WALK_DIRTY){
						emit = !(fl & WALK_DEAD);
#pragma unroll
						for(int j = 0; j < CH; ++j){
#pragma unroll
							for(int d = 0; d < 4; ++d){
								if(emit){ acc[j][d] = ~__hip_atomic_load(ob + (j*4 + d)*WAVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
								__hip_atomic_store(ob + (j*4 + d)*WAVE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							}
						}
					}
					if(lane == 0){
						__hip_atomic_store(state, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						__hip_atomic_store(state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
			}
			if(emit){
				// (most pairs end without a surviving column anywhere in the tile: one wave-wide test instead of CH rounds
				// of valid-mask loads and lane counts -- what a pair costs beyond its rows matters when queries are short:
				// 120 rows per query in C3's shape)
				bool any_left = false;
#pragma unroll
				for(int j = 0; j < CH; ++j){ any_left |= ((acc[j].x | acc[j].y | acc[j].z | acc[j].w) != 0); }
				if(__any(any_left)){
					// A pair WITH surviving columns: the CH real-column masks are requested together, and a round without a
					// surviving column in any lane then costs no memory operation at all.  (One emit_mask_hits per round --
					// a dependent load of the mask, then the scan and the atomic, CH times in a row with nothing else of this
					// wave in flight -- made every pair with a hit cost ~145 us: with hits in half the queries the walk form
					// lost 14 % on 150-base reads, profiles/r04_walk_hit_cost.txt.)
#pragma unroll
					for(int j = 0; j < CH; ++j){
						const uint32_t u = u0 + (uint32_t)j*WAVE;
						acc[j] = (u <= umax) ? (acc[j] & reinterpret_cast<const u32x4*>(a.valid)[u]) : (u32x4)(0u);
					}
#pragma unroll
					for(int j = 0; j < CH; ++j){
						emit_masked_hits_buffered(a, hbuf, hst, q, min(u0 + (uint32_t)j*WAVE, umax), acc[j], n, (uint64_t)q*a.runs_per_query + c*CH + j);
					}
				}
			}
		}
		s += take;
		if(j1 == npos && c + 1 == ct){ ++q; }
	}
	wave_hits_flush(a, hbuf, hst);          // one reservation for everything the wave found
}

// ---- the walk form, ADDRESS BAND after ADDRESS BAND ---------------------------------------------------------------
// A random-row gather over a matrix of ~100 GB reads 3-4 % faster when, at any moment, all waves read from the same
// narrow part of it: concurrent accesses to different large regions of the device's memory cost each other
// (tools/micro/placement_probe.hip: a 105 GB block that reads 6.66 TB/s reads 6.91 when the rows are taken window after
// window; every 4 GiB window alone reads 6.95; profiles/r03_placement_probe.txt).  AND is order independent, so the
// rows of every query are bucketed by BAND of the matrix (band_bucket_kernel, one launch) and every wave walks band 0's
// part of its share of the batch, then band 1's, ... -- all waves move up the matrix together.  A (query, band) part is reduced in registers
// as before; parts meet in the query's slot: an all-zero part (the rule after a few dozen random rows) only raises the
// slot's DEAD flag, a part with surviving columns ORs its complemented mask in (DIRTY).  Nothing waits for anything:
// and_band_finish_kernel, queued behind, reads the slots of the queries that are DIRTY and not DEAD, reports their
// columns and leaves every slot zero again.  One column tile only (rows up to 16 KiB).
struct BandArgs {
	uint32_t bands;                 // B <= BAND_MAX
	uint32_t rows_per_band;         // matrix rows per band: ceil(rows of the matrix / B)
	uint32_t *orbuf;                // [n_queries][CH*4*64] complemented partial masks; zero between searches
	uint32_t *state;                // [n_queries] WALK_DEAD | WALK_DIRTY; zero between searches
};
static constexpr uint32_t BAND_MAX = 64;

// The rows of query q, bucketed by band IN PLACE of the query's stretch of the row list: rows2[pos_off[q]*nh + loc[q][b] ...
// + loc[q][b+1]) are q's rows in band b (loc is [n_queries][bands + 1], loc[q][bands] = q's rows).  One workgroup per query;
// the order inside a bucket is free.
__global__ __launch_bounds__(256) void band_bucket_kernel(const uint32_t *__restrict__ rows, const uint64_t *__restrict__ pos_off, const uint32_t *__restrict__ nkmer,
                                                         uint32_t num_hash, uint32_t bands, uint32_t rows_per_band, uint32_t *__restrict__ loc, uint32_t *__restrict__ rows2)
{
	__shared__ uint32_t h[BAND_MAX], cur[BAND_MAX];
	const uint32_t q = blockIdx.x;
	if(threadIdx.x < BAND_MAX){ h[threadIdx.x] = 0; }
	__syncthreads();
	const uint64_t base = pos_off[q]*num_hash;
	const uint32_t n = nkmer[q]*num_hash;
	for(uint32_t i = threadIdx.x; i < n; i += blockDim.x){
		atomicAdd(&h[min(rows[base + i]/rows_per_band, bands - 1)], 1u);
	}
	__syncthreads();
	if(threadIdx.x == 0){
		uint32_t *lq = loc + (uint64_t)q*(bands + 1);
		uint32_t run = 0;
		for(uint32_t b = 0; b < bands; ++b){ cur[b] = run; lq[b] = run; run += h[b]; }
		lq[bands] = run;
	}
	__syncthreads();
	for(uint32_t i = threadIdx.x; i < n; i += blockDim.x){
		const uint32_t r = rows[base + i];
		rows2[base + atomicAdd(&cur[min(r/rows_per_band, bands - 1)], 1u)] = r;
	}
}

// (the grid and a wave's share of the batch's POSITIONS are and_walk_kernel's: wa.total_slots, wa.per_wave; in band b a
// wave takes, of every query piece it holds, the same fraction of the query's band-b bucket as the piece is of the query)
template <int CH, int UNROLL>
__global__ __launch_bounds__(WALK_WG_WAVES*WAVE) __attribute__((amdgpu_waves_per_eu(2, 2))) void and_band_walk_kernel(SearchArgs a, BandArgs ba, WalkArgs wa, const uint32_t *__restrict__ rows2,
                                                                           const uint32_t *__restrict__ loc, const uint64_t *__restrict__ pos_off,
                                                                           const uint32_t *__restrict__ nkmer)
{
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6));
	const uint64_t s_begin = (uint64_t)gw*wa.per_wave;
	const uint64_t s_end = min(wa.total_slots, s_begin + wa.per_wave);
	if(s_begin >= s_end){ return; }
	const uint32_t row_bytes = a.units_per_row*16u;
	const uint32_t b1 = ba.bands + 1;
	// the query that holds position s_begin: the largest q with pos_off[q] <= s_begin
	uint32_t q_begin = 0;
	{
		uint32_t hi = a.n_queries;
		while(hi - q_begin > 1){
			const uint32_t mid = q_begin + (hi - q_begin)/2;
			if(pos_off[mid] <= s_begin){ q_begin = mid; } else { hi = mid; }
		}
	}
	for(uint32_t b = 0; b < ba.bands; ++b){
		uint64_t s = s_begin;
		uint32_t q = q_begin;
		while(s < s_end){
			const uint64_t p0 = pos_off[q];
			const uint64_t npos = pos_off[q + 1] - p0;
			if(npos == 0){ ++q; continue; }                  // a query shorter than k (q stays in range: s < total_slots)
			const uint32_t j0 = (uint32_t)(s - p0);
			const uint32_t take = (uint32_t)min(npos - j0, s_end - s);
			const uint32_t n = nkmer[q];
			const uint32_t jv1 = min(j0 + take, n);              // positions past the distinct k-mers hold no rows
			uint32_t nrows = 0;
			const uint32_t *rq = rows2;
			if(j0 < jv1){
				const uint32_t *lq = loc + (uint64_t)q*b1;
				const uint32_t c0 = lq[b], cnt = lq[b + 1] - c0;
				const uint32_t lo = (uint32_t)((uint64_t)cnt*j0/n), hi = (uint32_t)((uint64_t)cnt*jv1/n);
				nrows = hi - lo;
				rq = rows2 + p0*a.num_hash + c0 + lo;
			}
			if(nrows){
				u32x4 acc[CH];
#pragma unroll
				for(int j = 0; j < CH; ++j){ acc[j] = ~(u32x4)(0u); }
				for(uint32_t i = 0; i < nrows; i += UNROLL){
					// (parts are short here -- a band's share of a query -- so a group past the part's end does not read its last
					// row again as in and_walk_kernel: its descriptors are EMPTY, the loads touch no memory and return 0, and
					// `fill` turns that into all ones)
					__amdgpu_buffer_rsrc_t rs[UNROLL];
					uint32_t fill[UNROLL];
#pragma unroll
					for(int u = 0; u < UNROLL; ++u){
						const bool in = i + u < nrows;
						const uint32_t r = rq[min(i + u, nrows - 1)];
						rs[u] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.db + (uint64_t)r*a.stride), 0, in ? row_bytes : 0u, 0x00020000);
						fill[u] = in ? 0u : ~0u;
					}
#pragma unroll
					for(int j = 0; j < CH; ++j){
						u32x4 x[UNROLL];
#pragma unroll
						for(int u = 0; u < UNROLL; ++u){ x[u] = __builtin_amdgcn_raw_buffer_load_b128(rs[u], lane*16u, j*1024, 2 /* nt */); }
#pragma unroll
						for(int u = 0; u < UNROLL; ++u){ acc[j] &= (x[u] | fill[u]); }
						__builtin_amdgcn_sched_barrier(0);               // one KiB-step of UNROLL rows at a time (and_walk_kernel)
					}
				}
				bool nz = false;
#pragma unroll
				for(int j = 0; j < CH; ++j){ nz |= ((acc[j].x | acc[j].y | acc[j].z | acc[j].w) != 0); }
				const bool live = __any(nz);
				if(live){
					uint32_t *ob = ba.orbuf + (uint64_t)q*(CH*4*WAVE) + lane;
#pragma unroll
					for(int j = 0; j < CH; ++j){
#pragma unroll
						for(int d = 0; d < 4; ++d){
							const uint32_t v = ~acc[j][d];
							if(v){ __hip_atomic_fetch_or(ob + (j*4 + d)*WAVE, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
						}
					}
				}
				if(lane == 0){ __hip_atomic_fetch_or(ba.state + q, live ? WALK_DIRTY : WALK_DEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
			}
			s += take;
			if(j0 + take == npos){ ++q; }
		}
	}
}

// One wave per query: report the columns that survived in every part, leave the query's slot zero.
template <int CH>
__global__ __launch_bounds__(256) void and_band_finish_kernel(SearchArgs a, BandArgs ba, const uint32_t *__restrict__ nkmer)
{
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint32_t q = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6));
	if(q >= a.n_queries){ return; }
	const uint32_t fl = __builtin_amdgcn_readfirstlane(ba.state[q]);
	if(fl == 0){ return; }                                           // a query without rows
	const bool emit = (fl & WALK_DIRTY) && !(fl & WALK_DEAD);
	u32x4 acc[CH];
	if(fl & WALK_DIRTY){
		uint32_t *ob = ba.orbuf + (uint64_t)q*(CH*4*WAVE) + lane;
#pragma unroll
		for(int j = 0; j < CH; ++j){
#pragma unroll
			for(int d = 0; d < 4; ++d){
				acc[j][d] = ~ob[(j*4 + d)*WAVE];
				ob[(j*4 + d)*WAVE] = 0u;
			}
		}
	}
	if(lane == 0){ ba.state[q] = 0u; }
	if(emit){
		const uint32_t n = nkmer[q];
		const uint32_t umax = a.units_per_row - 1;
#pragma unroll
		for(int j = 0; j < CH; ++j){      // (the real-column masks requested together: and_walk_kernel)
			const uint32_t u = lane + (uint32_t)j*WAVE;
			acc[j] = (u <= umax) ? (acc[j] & reinterpret_cast<const u32x4*>(a.valid)[u]) : (u32x4)(0u);
		}
#pragma unroll
		for(int j = 0; j < CH; ++j){
			emit_masked_hits(a, q, min(lane + (uint32_t)j*WAVE, umax), acc[j], n, (uint64_t)q*a.runs_per_query + j);
		}
	}
}

// Narrow databases (a row is at most 64/G 16-byte units, e.g. one 2048-column file = 16 units): G queries
// share a wave, 64/G lanes each, so a wave-load still moves up to 1 KiB.  Row indices are per lane group
// (vector loads, broadcast within the group).  Shorter row lists are padded by re-reading their last row
// (AND is idempotent), so there is no divergence inside the loop.
template <int G, int UNROLL>
__global__ __launch_bounds__(SEARCH_THREADS) void and_narrow_kernel(SearchArgs a)
{
	constexpr uint32_t LG = WAVE/G;
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint32_t l = lane % LG;
	const uint64_t tile = (uint64_t)blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6);
	const uint64_t q64 = tile*G + lane/LG;
	const bool has_q = (q64 < a.n_queries);
	const uint32_t q = has_q ? (uint32_t)q64 : 0;
	const uint32_t n = has_q ? a.nkmer[q] : 0;
	const uint32_t nrows = n*a.num_hash;
	const bool active = (nrows != 0) && (l < a.units_per_row);
	const uint32_t *rq = a.rows + a.pos_off[q]*a.num_hash;
	const uint32_t unit = (l < a.units_per_row) ? l : 0;

	u32x4 acc = ~(u32x4)(0u);
	if(active){
		for(uint32_t i = 0; __any(i < nrows); i += UNROLL){
			u32x4 x[UNROLL];
#pragma unroll
			for(int u = 0; u < UNROLL; ++u){
				const uint32_t idx = min(i + u, nrows - 1);
				const uint32_t r = rq[idx];
				x[u] = load16<true>(reinterpret_cast<const u32x4*>(a.db + (uint64_t)r*a.stride) + unit);
			}
#pragma unroll
			for(int u = 0; u < UNROLL; ++u){ acc &= x[u]; }
			if(a.early_exit && !__any((acc.x | acc.y | acc.z | acc.w) != 0)){ break; }
		}
	}
	__shared__ WgHitScratch wg_scratch;
	emit_mask_hits(a, q, unit, acc, n, blockIdx.x, active, &wg_scratch);      // every wave of the workgroup gets here, exactly once (run = the workgroup: queries ascend with waves and lanes)
}

// Second pass of the segmented AND: one thread per (query, 16-byte unit).
__global__ __launch_bounds__(256) void and_combine_kernel(SearchArgs a)
{
	const uint32_t u0 = blockIdx.x*blockDim.x + threadIdx.x;
	const uint32_t q = blockIdx.y;
	const uint32_t n = a.nkmer[q];
	if(n == 0){ return; }                                  // uniform per workgroup
	const bool on = (u0 < a.units_per_row);
	const uint32_t unit = on ? u0 : 0;
	const u32x4 acc = reinterpret_cast<const u32x4*>(a.partial)[(uint64_t)q*a.units_per_row + unit];
	emit_mask_hits(a, q, unit, acc, n, (uint64_t)q*a.runs_per_query + u0/WAVE, on);      // (a wave = 64 consecutive units of one query)
}

// threshold < 1: count, per column, the k-mers whose every hash row has the bit set
// (kwage.cpp:404-433 with increment_count, bloom.h:291-330).  Counters are bit-sliced:
// plane[p] holds bit p of the counter of each of the lane's 128 columns.
template <int PLANES>
__device__ __forceinline__ void planes_add(u32x4 (&plane)[PLANES], u32x4 carry, int from)
{
#pragma unroll
	for(int p = 0; p < PLANES; ++p){
		if(p < from){ continue; }
		const u32x4 t = plane[p] & carry;
		plane[p] ^= carry;
		carry = t;
	}
}

// carry-save adder: (sum, carry) of three 1-bit vectors
__device__ __forceinline__ void csa(u32x4 &sum, u32x4 &carry, u32x4 a, u32x4 b, u32x4 c)
{
	const u32x4 u = a ^ b;
	carry = (a & b) | (u & c);
	sum = u ^ c;
}

// Bit-parallel comparator: mask of the columns whose bit-sliced counter is >= thr, evaluated plane by
// plane from the top.
template <int PLANES>
__device__ __forceinline__ u32x4 planes_ge(const u32x4 (&plane)[PLANES], uint32_t thr)
{
	u32x4 gt = (u32x4)(0u);
	u32x4 eq = ~(u32x4)(0u);
#pragma unroll
	for(int p = PLANES - 1; p >= 0; --p){
		const u32x4 t4 = (u32x4)(((thr >> p) & 1u) ? ~0u : 0u);
		gt |= eq & plane[p] & ~t4;
		eq &= ~(plane[p] ^ t4);
	}
	u32x4 ge = gt | eq;
	if(PLANES < 32 && (thr >> (PLANES & 31)) != 0){ ge = (u32x4)(0u); }   // a counter of PLANES bits cannot reach thr
	return ge;
}

// columns with count >= thr (kwage.cpp:497); then the count of every surviving column is
// re-assembled from the planes (num_match = match_count[i]).  Every lane of the wave must call it
// (`on` = the lane holds a real tile position): the records are placed with one atomic per wave.
template <int PLANES>
__device__ __forceinline__ void emit_count_hits(const SearchArgs &a, uint32_t q, uint32_t unit,
                                                const u32x4 (&plane)[PLANES], uint32_t thr, uint64_t run, bool on, WgHitScratch *wg = nullptr)
{
	u32x4 ge = (u32x4)(0u);
	if(on){ ge = planes_ge<PLANES>(plane, thr) & reinterpret_cast<const u32x4*>(a.valid)[unit]; }
	const uint32_t nge = __popc(ge.x) + __popc(ge.y) + __popc(ge.z) + __popc(ge.w);
	unsigned long long slot = wg ? reserve_hits_wg(a, nge, wg, run) : reserve_hits(a, nge, run);
#pragma unroll
	for(int d = 0; d < 4; ++d){
		uint32_t bits = ge[d];
		while(bits){
			const uint32_t b = __ffs(bits) - 1;
			bits &= bits - 1;
			uint32_t cnt = 0;
#pragma unroll
			for(int p = 0; p < PLANES; ++p){ cnt |= ((plane[p][d] >> b) & 1u) << p; }
			store_hit(a, slot++, q, unit*128u + d*32u + b, cnt);
		}
	}
}

// The same through the wave's LDS buffer (the persistent count kernel; see WaveHitBuf).
template <int PLANES>
__device__ __forceinline__ void emit_count_hits_buffered(const SearchArgs &a, WaveHitBuf *buf, WaveHitState &st, uint32_t q, uint32_t unit,
                                                         const u32x4 (&plane)[PLANES], uint32_t thr, uint64_t run, bool on)
{
	u32x4 ge = (u32x4)(0u);
	if(on){ ge = planes_ge<PLANES>(plane, thr) & reinterpret_cast<const u32x4*>(a.valid)[unit]; }
	const uint32_t nge = __popc(ge.x) + __popc(ge.y) + __popc(ge.z) + __popc(ge.w);
	if(!__any(nge != 0)){ return; }
	uint32_t at = 0;
	const bool buffered = wave_hits_place(a, buf, st, nge, run, at);
	unsigned long long slot = buffered ? 0ull : reserve_hits(a, nge, run);
#pragma unroll
	for(int d = 0; d < 4; ++d){
		uint32_t bits = ge[d];
		while(bits){
			const uint32_t b = __ffs(bits) - 1;
			bits &= bits - 1;
			uint32_t cnt = 0;
#pragma unroll
			for(int p = 0; p < PLANES; ++p){ cnt |= ((plane[p][d] >> b) & 1u) << p; }
			if(buffered){ kwage_hit h; h.query = q; h.column = unit*128u + d*32u + b + a.col_base; h.num_match = cnt; buf->rec[at++] = h; }
			else{ store_hit(a, slot++, q, unit*128u + d*32u + b, cnt); }
		}
	}
}

// Add two bit-sliced counters: acc (PLANES planes) += b (the first nb planes of `b`, the rest zero).
template <int PLANES, typename LOADB>
__device__ __forceinline__ void planes_accumulate(u32x4 (&acc)[PLANES], int nb, LOADB loadb)
{
	u32x4 carry = (u32x4)(0u);
#pragma unroll
	for(int p = 0; p < PLANES; ++p){
		const u32x4 b = (p < nb) ? loadb(p) : (u32x4)(0u);
		const u32x4 x = acc[p] ^ b;
		const u32x4 cnext = (acc[p] & b) | (carry & x);
		acc[p] = x ^ carry;
		carry = cnext;
	}
}

// Four 1-bit vectors into the bit-sliced counters: a carry-save tree, so that the ripple through the upper planes
// runs once per four k-mers (planes 0 and 1 are the CSA residues).
template <int PLANES>
__device__ __forceinline__ void planes_add4(u32x4 (&plane)[PLANES], u32x4 m0, u32x4 m1, u32x4 m2, u32x4 m3)
{
	if(PLANES >= 3){
		u32x4 twoA, twoB, four, s;
		csa(s, twoA, plane[0], m0, m1);
		csa(plane[0], twoB, s, m2, m3);
		csa(plane[1], four, plane[1], twoA, twoB);
		planes_add<PLANES>(plane, four, 2);
	}
	else{
		planes_add<PLANES>(plane, m0, 0);
		planes_add<PLANES>(plane, m1, 0);
		planes_add<PLANES>(plane, m2, 0);
		planes_add<PLANES>(plane, m3, 0);
	}
}

// Eight at once (7 CSAs, then one ripple from plane 3): the narrow kernels keep eight k-mers' rows in flight per wave.
template <int PLANES>
__device__ __forceinline__ void planes_add8(u32x4 (&plane)[PLANES], const u32x4 (&m)[8])
{
	static_assert(PLANES >= 4, "planes_add8 needs four planes");
	u32x4 s1, s2, s3, c1, c2, c3, c4, t1, f1, f2, e1;
	csa(s1, c1, m[0], m[1], m[2]);
	csa(s2, c2, m[3], m[4], m[5]);
	csa(s3, c3, m[6], m[7], plane[0]);
	csa(plane[0], c4, s1, s2, s3);
	csa(t1, f1, c1, c2, c3);
	csa(plane[1], f2, t1, c4, plane[1]);
	csa(plane[2], e1, f1, f2, plane[2]);
	planes_add<PLANES>(plane, e1, 3);
}

// The counting loop of one (query, 1 KiB column tile): k-mers [0, nk) of the row list `rq` ([k-mer][NH] row indices),
// kmer_match = AND over the NH hash rows (kwage.cpp:409-422), added into the lane's bit-sliced counters
// (increment_count, bloom.h:291-330).  Four k-mers per step: 4*NH row loads in flight.  (Eight per step -- 94 VGPRs, 5
// waves/SIMD -- measured the same with one and two hash functions at C2's shape: 1.916 vs 1.921 ms, 3.752 vs 3.755 ms,
// round 2; not kept.)  `stop(done)` is asked every 64 k-mers whether the tile can be given up (early exit); returns
// false when it was.
template <int PLANES, int NH, typename STOP>
__device__ __forceinline__ bool count_kmers(const uint8_t *db, uint64_t stride, const uint32_t *rq, uint32_t nk, uint32_t unit,
                                            u32x4 (&plane)[PLANES], STOP stop)
{
	constexpr int KPS = 4;
	uint32_t i = 0;
	for(; i + KPS <= nk; i += KPS){
		u32x4 m[KPS];
#pragma unroll
		for(int u = 0; u < KPS; ++u){
			u32x4 x[NH];
#pragma unroll
			for(int h = 0; h < NH; ++h){
				const uint32_t r = rq[(i + u)*NH + h];
				x[h] = load16<true>(reinterpret_cast<const u32x4*>(db + (uint64_t)r*stride) + unit);
			}
			m[u] = x[0];
#pragma unroll
			for(int h = 1; h < NH; ++h){ m[u] &= x[h]; }    // kmer_match &= slice
		}
		planes_add4<PLANES>(plane, m[0], m[1], m[2], m[3]);
		if(((i + KPS) & 63u) == 0 && stop(i + KPS)){ return false; }
	}
	for(; i < nk; ++i){
		u32x4 mm = ~(u32x4)(0u);
#pragma unroll
		for(int h = 0; h < NH; ++h){
			const uint32_t r = rq[i*NH + h];
			mm &= load16<true>(reinterpret_cast<const u32x4*>(db + (uint64_t)r*stride) + unit);
		}
		planes_add<PLANES>(plane, mm, 0);
	}
	return true;
}

template <int PLANES, int NH, bool SEG>
__global__ __launch_bounds__(SEARCH_THREADS) void count_kernel(SearchArgs a)
{
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint64_t tile = (uint64_t)blockIdx.x*(SEARCH_THREADS/WAVE) + (threadIdx.x >> 6);
	if(tile >= (uint64_t)a.n_queries*a.segs*a.chunks){ return; }

	uint32_t q, sg, c;
	tile_coords(a, tile, q, sg, c);
	const uint32_t n = a.nkmer[q];
	if(n == 0){ return; }
	uint32_t k0 = 0, k1 = n;
	if(SEG){
		k0 = sg*a.seg_kmers;
		k1 = min(n, k0 + a.seg_kmers);
		if(k0 >= k1){ return; }
	}
	const uint32_t nk = k1 - k0;
	const uint32_t *rq = a.rows + (a.pos_off[q] + k0)*NH;

	const uint32_t u0 = c*WAVE + lane;
	const bool live = (u0 < a.units_per_row);
	const uint32_t unit = live ? u0 : (a.units_per_row - 1);

	u32x4 plane[PLANES];
#pragma unroll
	for(int p = 0; p < PLANES; ++p){ plane[p] = (u32x4)(0u); }

	// kwage.cpp:478-481 per tile: stop once no column of the tile can still reach the threshold even if every
	// remaining k-mer matched (max count + remaining < threshold)
	const bool whole = count_kmers<PLANES, NH>(a.db, a.stride, rq, nk, unit, plane, [&](uint32_t done) -> bool {
		if(SEG || !a.early_exit){ return false; }
		const uint32_t remaining = nk - done;
		const uint32_t thr = a.qthr[q];
		if(thr <= remaining){ return false; }
		const u32x4 can = planes_ge<PLANES>(plane, thr - remaining);
		return !__any((can.x | can.y | can.z | can.w) != 0);
	});
	if(!whole){ return; }

	if(SEG){
		// partial counters of this segment -> slab [query][segment][plane][unit]
		if(live){
			u32x4 *slab = reinterpret_cast<u32x4*>(a.partial) + ((uint64_t)q*a.segs + sg)*PLANES*a.units_per_row + unit;
#pragma unroll
			for(int p = 0; p < PLANES; ++p){ slab[(uint64_t)p*a.units_per_row] = plane[p]; }
		}
	}
	else{
		emit_count_hits<PLANES>(a, q, unit, plane, a.qthr[q], (uint64_t)q*a.runs_per_query + c, live);
	}
}

// ---- early exit at threshold < 1: SCREEN, then REFINE (and_screen_kernel's idea for the count path) ---------------------
// kwage.cpp:478-481 gives a query up once max count + remaining k-mers < threshold.  Per column that bound cannot prune
// before n - threshold k-mers are in, and then needs a margin the random matches stay below: every tile reads the first
// ~30 % of a query's rows at t = 0.8 whatever else happens -- the launch's floor.  What the tiled count_kernel adds to that
// floor is the tail: a tile that HOLDS a column above the threshold is counted to the end, 1 KiB wide, by one wave.
//   count_screen_kernel  count_kernel's tile loop on a persistent grid; every 8 k-mers (once the bound can bite): no column
//                        can reach the threshold any more -> done; such columns in at most `max_groups` 128-byte groups and
//                        `min_rows` k-mers to go -> the tile is handed over (cluster, items with their counters so far,
//                        units of `seg_rows` k-mers); otherwise it goes on, and reports itself at the end of the list.
//   count_refine_kernel  eight units per wave, 8 lanes x 16 B each: the unit's k-mers counted in UP planes -> slab.
//   count_refine_emit_kernel  per cluster: item counters + the units' counters, threshold, one reservation per wave.
// Hand a (query, KiB tile) over to the refine launch at k-mer `k_done` of `nk`: one cluster, one item (its PLANES counters so
// far) per 128-byte group of `gb`, one unit per item and segment of ra.seg_rows remaining k-mers.  False: a list is full
// (nothing was handed over).  Wave-uniform arguments except `plane`; every lane calls.
template <int PLANES>
__device__ __forceinline__ bool count_hand_over(const RefineArgs &ra, RefineChunk &cc, RefineChunk &ci, RefineChunk &cu, uint32_t q, uint32_t c, uint32_t gb,
                                                const u32x4 (&plane)[PLANES], uint32_t k_done, uint32_t nk, uint64_t rq_off)
{
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint32_t ngroups = __popc(gb);
	const uint32_t nseg = (nk - k_done + ra.seg_rows - 1)/ra.seg_rows;
	const uint32_t nu = ngroups*nseg;
	const uint32_t c0 = refine_take(ra, 0, cc, 1);
	const uint32_t i0 = (c0 != REFINE_NONE) ? refine_take(ra, 1, ci, ngroups) : REFINE_NONE;
	const uint32_t un0 = (i0 != REFINE_NONE) ? refine_take(ra, 2, cu, nu) : REFINE_NONE;
	if(un0 == REFINE_NONE){
		if(c0 != REFINE_NONE){ refine_none_clusters(ra, c0, c0 + 1); }
		return false;
	}
	if(lane == 0){
		RefineCluster cl; cl.q = q; cl.kstep_groups = (c << 8) | gb; cl.first_item = i0; cl.n = nk;
		cl.first_unit = un0; cl.nseg = nseg; cl.pad0 = 0; cl.pad1 = 0;
		ra.clusters[c0] = cl;
	}
	const uint32_t k = lane >> 3;
	if((gb >> k) & 1u){
		const uint32_t it = i0 + __popc(gb & ((1u << k) - 1u));
		u32x4 *dst = reinterpret_cast<u32x4*>(ra.masks) + (uint64_t)it*PLANES*8 + (lane & 7u);
#pragma unroll
		for(int p = 0; p < PLANES; ++p){ dst[p*8] = plane[p]; }
	}
	for(uint32_t e = lane; e < nu; e += WAVE){
		const uint32_t j = e / nseg, sg = e % nseg;
		RefineUnit un;
		un.item = i0 + j;
		un.r0 = k_done + sg*ra.seg_rows;                               // (k-mers, not rows)
		un.r1 = min(nk, un.r0 + ra.seg_rows);
		un.unit0 = c*WAVE + nth_set_bit(gb, j)*8u;
		un.rq_lo = (uint32_t)rq_off; un.rq_hi = (uint32_t)(rq_off >> 32);
		un.q = q; un.pad = 0;
		ra.units[un0 + e] = un;
	}
	return true;
}

// (amdgpu_waves_per_eu: the launch holds as many waves as the registers allow -- the host asks the runtime -- and that is
// 2-3 per SIMD whatever the scheduler does; left alone it aimed higher and issued a step's eight rows two or three at a
// time: see and_walk_kernel)
template <int PLANES, int NH>
__global__ __launch_bounds__(SEARCH_THREADS) __attribute__((amdgpu_waves_per_eu(2, (PLANES <= 10) ? 3 : 2))) void count_screen_kernel(SearchArgs a, RefineArgs ra)
{
	const uint32_t check_mask = ra.check_every - 1;       // k-mers between two looks at the bound (a power of two >= 8)
	constexpr int KPS = (NH <= 2) ? 8 : 4;    // k-mers per step: 8 or 4*NH rows in flight
	__shared__ WaveHitBuf hit_bufs[SEARCH_THREADS/WAVE];
	WaveHitBuf *hbuf = &hit_bufs[threadIdx.x >> 6];
	WaveHitState hst;
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint64_t n_tiles = (uint64_t)a.n_queries*a.chunks;
	const uint32_t n_waves = gridDim.x*(blockDim.x/WAVE);
	const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6));
	RefineChunk cc, ci, cu;
	cc.next = gw*ra.lc.stat; cc.end = cc.next + ra.lc.stat;
	ci.next = gw*ra.li.stat; ci.end = ci.next + ra.li.stat;
	cu.next = gw*ra.lu.stat; cu.end = cu.next + ra.lu.stat;
	bool lists_full = false;
	// Tiles are HEAVY here (hundreds of k-mers before the bound can bite) and a wave gets only a few: dealt out beforehand,
	// 13 000 tiles over 4096 waves are 3 for most waves and 4 for some -- a quarter of the launch spent waiting for those.
	// So the waves draw their tiles, `queue_batch` at a time, from a counter (ra.counters[4..5], zeroed with the others).
	unsigned long long *queue = reinterpret_cast<unsigned long long*>(ra.counters + 4);
	// (every wave's FIRST tile is its own number -- thousands of waves asking the one counter at the same moment would
	// start one after the other, ~25 ns apart -- and the queue holds the tiles from n_waves on)
	uint64_t tile = gw, tile_end = (gw < n_tiles) ? (uint64_t)gw + 1 : (uint64_t)gw;      // (the last workgroup may have a wave too many)
	for(;;){
		if(tile == tile_end){
			unsigned long long t0 = 0;
			if(lane == 0){ t0 = atomicAdd(queue, (unsigned long long)ra.queue_batch); }
			tile = n_waves + (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(t0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)t0));
			if(tile >= n_tiles){ break; }
			tile_end = min(n_tiles, tile + ra.queue_batch);
		}
		const uint32_t q = __builtin_amdgcn_readfirstlane((uint32_t)(tile / a.chunks));
		const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)(tile % a.chunks));
		++tile;
		const uint32_t nk = a.nkmer[q];
		if(nk == 0){ continue; }
		const uint32_t thr = a.qthr[q];
		const uint64_t rq_off = a.pos_off[q]*NH;
		const uint32_t *rq = a.rows + rq_off;
		const uint32_t u0 = c*WAVE + lane;
		const bool live = (u0 < a.units_per_row);
		const uint32_t unit = live ? u0 : (a.units_per_row - 1);
		const u32x4 real = live ? reinterpret_cast<const u32x4*>(a.valid)[unit] : (u32x4)(0u);     // only real columns keep a tile alive

		u32x4 plane[PLANES];
#pragma unroll
		for(int p = 0; p < PLANES; ++p){ plane[p] = (u32x4)(0u); }
		bool may_hand_over = !lists_full, gone = false;
		// A 128-byte group none of whose columns can reach the threshold any more is not READ any more (its eight lanes sit the
		// loads out; their counters stay where they are, below the bound for good): a tile that goes on -- more groups alive than
		// the refine launch takes, a column not on track yet, the lists full -- costs what its live groups cost.
		bool alive = true;
		uint32_t i = 0;
		for(; i + KPS <= nk; ){
			if(alive){
				u32x4 m[KPS];
#pragma unroll
				for(int u = 0; u < KPS; ++u){
					u32x4 x[NH];
#pragma unroll
					for(int h = 0; h < NH; ++h){
						const uint32_t r = rq[(i + u)*NH + h];
						x[h] = load16<true>(reinterpret_cast<const u32x4*>(a.db + (uint64_t)r*a.stride) + unit);
					}
					m[u] = x[0];
#pragma unroll
					for(int h = 1; h < NH; ++h){ m[u] &= x[h]; }    // kmer_match &= slice
				}
				if constexpr(KPS == 8){ planes_add8<PLANES>(plane, m); }
				else{ planes_add4<PLANES>(plane, m[0], m[1], m[2], m[3]); }
			}
			i += KPS;
			const uint32_t remaining = nk - i;
			if((i & check_mask) != 0 || thr <= remaining){ continue; }
			// kwage.cpp:478-481 per column: the columns that can still reach the threshold if every remaining k-mer matches
			const u32x4 can = planes_ge<PLANES>(plane, thr - remaining) & real;
			const uint32_t gb = group_bits(__ballot((can.x | can.y | can.z | can.w) != 0));
			const uint32_t ngroups = __popc(gb);
			if(ngroups == 0){ gone = true; break; }
			alive = ((gb >> (lane >> 3)) & 1u) != 0;
			if(may_hand_over && ngroups <= ra.max_groups && remaining >= ra.min_rows){
				// Only columns that are ON TRACK -- matching at the threshold's rate so far: count >= threshold x i / n -- are worth
				// the refine launch, which counts every handed-over group to the END of the list.  A column that can still reach the
				// threshold but matches at a lower rate (a related genome: 40 % of the k-mers) is ruled out by the bound a few dozen
				// k-mers later; while one is left, the tile goes on as it is.
				const uint32_t on_track = (uint32_t)(((uint64_t)thr*i + nk - 1)/nk);
				const u32x4 off = can & ~planes_ge<PLANES>(plane, on_track);
				if(__any((off.x | off.y | off.z | off.w) != 0)){ continue; }
				may_hand_over = false;                    // (one attempt per tile)
				if(!count_hand_over<PLANES>(ra, cc, ci, cu, q, c, gb, plane, i, nk, rq_off)){
					lists_full = true;
					continue;
				}
				gone = true;
				break;
			}
		}
		if(gone){ continue; }
		for(; i < nk; ++i){
			if(!alive){ continue; }
			u32x4 mm = ~(u32x4)(0u);
#pragma unroll
			for(int h = 0; h < NH; ++h){
				const uint32_t r = rq[i*NH + h];
				mm &= load16<true>(reinterpret_cast<const u32x4*>(a.db + (uint64_t)r*a.stride) + unit);
			}
			planes_add<PLANES>(plane, mm, 0);
		}
		emit_count_hits_buffered<PLANES>(a, hbuf, hst, q, unit, plane, thr, (uint64_t)q*a.runs_per_query + c, live);
	}
	wave_hits_flush(a, hbuf, hst);
	refine_none_clusters(ra, cc.next, cc.end);
	refine_none_units(ra, cu.next, cu.end);
}

// UP counter planes per unit (seg_rows k-mers < 2^UP).  A lane holds the NH row numbers of ONE k-mer of a block of eight
// (fetched together, handed round by ds_bpermute); KPS k-mers' rows are in flight at a time.
template <int NH, int UP>
__global__ __launch_bounds__(SEARCH_THREADS) __attribute__((amdgpu_waves_per_eu(2, 3))) void count_refine_kernel(SearchArgs a, RefineArgs ra)
{
	constexpr int KPS = (NH <= 2) ? 8 : 4;
	const uint32_t lane = threadIdx.x & (WAVE - 1), l = lane & 7u, sh = lane & ~7u;
	const uint32_t n_units = refine_list_end(ra, 2);
	const uint64_t gw = (uint64_t)blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6);
	const uint64_t step = (uint64_t)gridDim.x*(blockDim.x/WAVE)*8;
	for(uint64_t base = gw*8; base < n_units; base += step){
		const uint64_t u = base + (lane >> 3);
		RefineUnit un;
		un.item = REFINE_NONE;
		if(u < n_units){ un = ra.units[u]; }
		if(un.item == REFINE_NONE){ continue; }                    // (lanes of a group agree)
		const uint32_t *rq = a.rows + (((uint64_t)un.rq_hi << 32) | un.rq_lo);
		const uint8_t *col = a.db + (uint64_t)(un.unit0 + l)*16;
		u32x4 plane[UP];
#pragma unroll
		for(int p = 0; p < UP; ++p){ plane[p] = (u32x4)(0u); }
		for(uint32_t b = un.r0; b < un.r1; b += 8){
			uint32_t idx[NH];          // the rows of k-mer b + l (past the end: the last k-mer's -- its match is dropped below)
#pragma unroll
			for(int h = 0; h < NH; ++h){ idx[h] = rq[(uint64_t)min(b + l, un.r1 - 1)*NH + h]; }
			u32x4 m[8];
#pragma unroll
			for(int k0 = 0; k0 < 8; k0 += KPS){
				u32x4 x[KPS][NH];
#pragma unroll
				for(int j = 0; j < KPS; ++j){
#pragma unroll
					for(int h = 0; h < NH; ++h){
						const uint32_t r = __shfl(idx[h], sh + k0 + j);
						x[j][h] = load16<true>(reinterpret_cast<const u32x4*>(col + (uint64_t)r*a.stride));
					}
				}
#pragma unroll
				for(int j = 0; j < KPS; ++j){
					u32x4 mm = x[j][0];
#pragma unroll
					for(int h = 1; h < NH; ++h){ mm &= x[j][h]; }
					m[k0 + j] = (b + k0 + j < un.r1) ? mm : (u32x4)(0u);
				}
			}
			planes_add8<UP>(plane, m);
		}
		u32x4 *dst = reinterpret_cast<u32x4*>(ra.slab) + u*UP*8 + l;
#pragma unroll
		for(int p = 0; p < UP; ++p){ dst[p*8] = plane[p]; }
	}
}

template <int PLANES, int UP>
__global__ __launch_bounds__(SEARCH_THREADS) void count_refine_emit_kernel(SearchArgs a, RefineArgs ra)
{
	__shared__ WaveHitBuf hit_bufs[SEARCH_THREADS/WAVE];
	WaveHitBuf *hbuf = &hit_bufs[threadIdx.x >> 6];
	WaveHitState hst;
	const uint32_t lane = threadIdx.x & (WAVE - 1), l = lane & 7u, g = lane >> 3;
	const uint32_t n_clusters = refine_list_end(ra, 0);
	const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6));
	const uint32_t nw = gridDim.x*(blockDim.x/WAVE);
	for(uint32_t ci = gw; ci < n_clusters; ci += nw){
		const RefineCluster cl = ra.clusters[ci];
		if(cl.q == REFINE_NONE){ continue; }
		const uint32_t groups = cl.kstep_groups & 0xFFu, kstep = cl.kstep_groups >> 8;
		const bool on = ((groups >> g) & 1u) != 0;               // this lane's 128-byte group of the KiB-step is an item
		u32x4 plane[PLANES];
#pragma unroll
		for(int p = 0; p < PLANES; ++p){ plane[p] = (u32x4)(0u); }
		if(on){
			const uint32_t j = __popc(groups & ((1u << g) - 1u));
			const u32x4 *src = reinterpret_cast<const u32x4*>(ra.masks) + (uint64_t)(cl.first_item + j)*PLANES*8 + l;
#pragma unroll
			for(int p = 0; p < PLANES; ++p){ plane[p] = src[p*8]; }
			const u32x4 *sl = reinterpret_cast<const u32x4*>(ra.slab) + ((uint64_t)cl.first_unit + (uint64_t)j*cl.nseg)*UP*8 + l;
			for(uint32_t sg = 0; sg < cl.nseg; ++sg){
				const u32x4 *s2 = sl + (uint64_t)sg*UP*8;
				planes_accumulate<PLANES>(plane, UP < PLANES ? UP : PLANES, [&](int p){ return s2[p*8]; });
			}
		}
		emit_count_hits_buffered<PLANES>(a, hbuf, hst, cl.q, min(kstep*WAVE + lane, a.units_per_row - 1), plane, a.qthr[cl.q], (uint64_t)cl.q*a.runs_per_query + kstep, on);
	}
	wave_hits_flush(a, hbuf, hst);
}

// The same loop with the NEXT four k-mers' rows requested before the current four are added up (16 more VGPRs): for the
// persistent kernel below, whose few waves per CU leave nothing else to cover the adders' time.
// KPS k-mers per step: 4, or 8 (seven carry-save adders, then ONE ripple through the upper planes per eight k-mers): with
// 14 planes and more -- queries above 1 k positions -- the ripple is most of the kernel's instructions, and a wave of the
// persistent grid spends 42 % of its time issuing them (one 100 kb query at t = 0.9, 20 planes: SQ_ACTIVE_INST_ANY /
// SQ_WAVE_CYCLES, profiles/r04_long1t_pmc_occupancy.json); halving the ripples per k-mer is what eight per step buys.
template <int PLANES, int NH, int KPS>
__device__ __forceinline__ void count_kmers_prefetch(const uint8_t *db, uint64_t stride, const uint32_t *rq, uint32_t nk, uint32_t unit,
                                                     u32x4 (&plane)[PLANES])
{
	static_assert(KPS == 4 || KPS == 8, "four or eight k-mers per step");
	// With 14 counter planes and more the matches are counted in a BLOCK counter of seven planes first (120 k-mers at most)
	// and the block is added to the PLANES-plane total once per block: the upper planes are touched once per 120 k-mers
	// instead of once per eight -- the ripple through them was most of the 20-plane form's instructions (a wave spent 42 % of
	// its time issuing them, profiles/r04_long1t_pmc_occupancy.json).
	// (one hash function and up to 20 planes only: with more rows in flight per k-mer, or 32 planes, the 28 registers of the
	// block counter are the ones that spill -- tools/isa_check.py)
	constexpr bool BLOCKS = (KPS == 8 && PLANES >= 14 && PLANES <= 20 && NH == 1);
	constexpr int BP = 7;
	u32x4 blk[BP];
	uint32_t in_blk = 0;
	if constexpr(BLOCKS){
#pragma unroll
		for(int p = 0; p < BP; ++p){ blk[p] = (u32x4)(0u); }
	}
	auto fold = [&]() {
		if constexpr(BLOCKS){
			if(in_blk == 0){ return; }
			planes_accumulate<PLANES>(plane, BP, [&](int p) -> u32x4 { return blk[p < BP ? p : 0]; });
#pragma unroll
			for(int p = 0; p < BP; ++p){ blk[p] = (u32x4)(0u); }
			in_blk = 0;
		}
	};
	auto add = [&](const u32x4 (&m)[KPS]) {
		if constexpr(BLOCKS){
			planes_add8<BP>(blk, m);
			in_blk += 8;
			if(in_blk > 127 - 8){ fold(); }
		}
		else if constexpr(KPS == 8){ planes_add8<PLANES>(plane, m); }
		else{ planes_add4<PLANES>(plane, m[0], m[1], m[2], m[3]); }
	};
	auto fetch = [&](uint32_t i, u32x4 (&m)[KPS]) {
#pragma unroll
		for(int u = 0; u < KPS; ++u){
			u32x4 x[NH];
#pragma unroll
			for(int h = 0; h < NH; ++h){
				const uint32_t r = rq[(i + u)*NH + h];
				x[h] = load16<true>(reinterpret_cast<const u32x4*>(db + (uint64_t)r*stride) + unit);
			}
			m[u] = x[0];
#pragma unroll
			for(int h = 1; h < NH; ++h){ m[u] &= x[h]; }
		}
	};
	uint32_t i = 0;
	if(nk >= KPS){
		u32x4 cur[KPS], nxt[KPS];
		fetch(0, cur);
		for(i = KPS; i + KPS <= nk; i += KPS){
			fetch(i, nxt);
			add(cur);
#pragma unroll
			for(int u = 0; u < KPS; ++u){ cur[u] = nxt[u]; }
		}
		add(cur);
	}
	for(; i < nk; ++i){
		u32x4 mm = ~(u32x4)(0u);
#pragma unroll
		for(int h = 0; h < NH; ++h){
			const uint32_t r = rq[i*NH + h];
			mm &= load16<true>(reinterpret_cast<const u32x4*>(db + (uint64_t)r*stride) + unit);
		}
		if constexpr(BLOCKS){
			planes_add<BP>(blk, mm, 0);
			if(++in_blk == 127){ fold(); }
		}
		else{ planes_add<PLANES>(plane, mm, 0); }
	}
	fold();
}

// The same counts from a PERSISTENT, statically balanced grid (and_walk_kernel's idea for the threshold < 1 path).  The
// tiled kernel above has one wave per (query, 1 KiB column tile): batch sizes whose waves are not a multiple of what the
// chip holds end in a part-filled round, and a single long query needs the segment slab + combine pass.  Here the launch
// has a fixed number of waves per CU and the batch's slot list -- for every query and column tile its positions,
// query-major, exactly and_walk_kernel's -- is cut into equal contiguous shares.  Counters do not fit a wave's registers
// for more than one KiB of columns, so a wave walks its share one (query, KiB tile) part at a time:
//   - a part that is a whole pair is thresholded and emitted at once;
//   - a pair CUT by share boundaries lies in a run of consecutive waves first_w .. last_w (known to each of them from
//     the pair's first slot, its k-mer count and the share size).  The parts are added up as a binary TREE over that
//     run, "last arriver continues": a wave stores its partial counters in its own block of `slab` ([w][1] for the
//     part that starts the pair, [w][0] for a part that began in an earlier wave), then arrives at the node it shares
//     with its sibling subtree (one counter per (wave, level)); the first to arrive is done, the second adds the
//     sibling's block to its registers (ripple-carry adders across the planes), zeroes the counter and climbs a level.
//     Whoever climbs past the root holds the pair's counters and emits.  A pair cut once costs one store, one
//     atomic and one load of PLANES KiB; a 1 Mb query spread over 80 waves is summed in 7 levels, not by one wave.
// Everything the protocol exchanges goes through device-scope stores / atomics and loads (the XCDs' L2s are not
// coherent with one another for plain accesses), ordered by waiting for the stores' acknowledgements (vmcnt).
static constexpr uint32_t CWALK_LEVELS = 32;      // tree depth a 32-bit wave count can need

struct CountWalkArgs {
	uint64_t total_slots;           // column tiles per row x positions of the batch
	uint64_t per_wave;              // slots per wave
	uint32_t coltiles;              // 1 KiB column tiles per row
	uint32_t *slab;                 // [waves][2][PLANES][4][64] partial counters of cut pairs (overwritten before they are read)
	uint32_t *arrived;              // [waves][CWALK_LEVELS] arrivals at the tree node (first wave of the node's subtree, level); zero between searches
	// TRUNC (early exit over few long queries at t < 1): only the first kcut[q] k-mers of query q are walked -- what the bound
	// max + remaining < threshold needs before it can rule a column out (the host's estimate from the matrix's density) -- and
	// the columns that can still reach the threshold go to the refine launch with their counters (count_hand_over)
	const uint64_t *slot_off;       // n_queries + 1: prefix of min(positions, kcut) -- the slots of the launch
	const uint32_t *kcut;           // n_queries
};

// (no amdgpu_waves_per_eu hint here, unlike and_walk_kernel: with it the 14-plane-and-more forms with several hash
// functions requested ONE k-mer's rows at a time -- tools/isa_check.py compares every kernel's loads in flight with round 4's)
template <int PLANES, int NH, bool TRUNC>
__global__ __launch_bounds__(WALK_WG_WAVES*WAVE) void count_walk_kernel(SearchArgs a, CountWalkArgs wa, RefineArgs ra, const uint32_t *__restrict__ rows,
                                                                    const uint64_t *__restrict__ pos_off, const uint32_t *__restrict__ nkmer,
                                                                    const uint32_t *__restrict__ qthr)
{
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6));
	uint64_t s = (uint64_t)gw*wa.per_wave;
	const uint64_t s1 = min(wa.total_slots, s + wa.per_wave);
	if(s >= s1){ return; }
	__shared__ WaveHitBuf hit_bufs[WALK_WG_WAVES];
	WaveHitBuf *hbuf = &hit_bufs[threadIdx.x >> 6];
	WaveHitState hst;
	const uint32_t ct = wa.coltiles;
	// the slots of the launch: every query's positions -- TRUNC: its first kcut positions -- times the column tiles
	const uint64_t *__restrict__ soff = TRUNC ? wa.slot_off : pos_off;
	RefineChunk cc, ci, cu;               // TRUNC: the wave's static places of the refine lists
	if(TRUNC){
		cc.next = gw*ra.lc.stat; cc.end = cc.next + ra.lc.stat;
		ci.next = gw*ra.li.stat; ci.end = ci.next + ra.li.stat;
		cu.next = gw*ra.lu.stat; cu.end = cu.next + ra.lu.stat;
	}

	// the query that holds slot s: the largest q with ct*soff[q] <= s
	uint32_t q = 0;
	{
		uint32_t hi = a.n_queries;
		while(hi - q > 1){
			const uint32_t mid = q + (hi - q)/2;
			if((uint64_t)ct*soff[mid] <= s){ q = mid; } else { hi = mid; }
		}
	}

	while(s < s1){
		const uint64_t p0s = soff[q];
		const uint64_t npos = soff[q + 1] - p0s;
		if(npos == 0){ ++q; continue; }
		const uint64_t rem = s - (uint64_t)ct*p0s;
		const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)(rem / npos));
		const uint32_t j0 = __builtin_amdgcn_readfirstlane((uint32_t)(rem % npos));
		const uint32_t take = (uint32_t)min(npos - j0, s1 - s);
		const uint32_t j1 = j0 + take;
		const uint32_t nk = nkmer[q];
		const uint32_t n = TRUNC ? min(nk, wa.kcut[q]) : nk;     // the k-mers walked here (TRUNC: the rest is the refine launch's)
		const uint64_t p0 = TRUNC ? pos_off[q] : p0s;            // where the query's rows lie
		const uint32_t jv1 = min(j1, n);                     // positions past the distinct k-mers hold no rows
		if(j0 < jv1){
			const uint32_t u0 = c*WAVE + lane;
			const bool live = (u0 < a.units_per_row);
			const uint32_t unit = live ? u0 : (a.units_per_row - 1);
			u32x4 plane[PLANES];
#pragma unroll
			for(int p = 0; p < PLANES; ++p){ plane[p] = (u32x4)(0u); }
			// (eight k-mers per step with 14 counter planes and more, where the ripple through the upper planes is most of the
			// kernel's instructions; the form without the prefetch and four per step at every width were knobs until round 5)
			count_kmers_prefetch<PLANES, NH, (PLANES >= 14) ? 8 : 4>(a.db, a.stride, rows + (p0 + j0)*NH, jv1 - j0, unit, plane);

			bool emit = true;
			if(j0 != 0 || jv1 != n){
				// the run of waves that hold a part of this pair, and this wave's place in it
				const uint64_t pair_start = (uint64_t)ct*p0s + (uint64_t)c*npos;       // slot of the pair's position 0
				const uint32_t first_w = (uint32_t)(pair_start / wa.per_wave);
				const uint32_t parts = (uint32_t)((pair_start + n - 1) / wa.per_wave) - first_w + 1;
				uint32_t rep = gw - first_w;                                           // first wave (relative) of the subtree whose sum this wave holds
				for(uint32_t stride = 1, level = 0; stride < parts; stride <<= 1, ++level){
					const uint32_t parent = rep & ~(2*stride - 1);
					if(rep == parent && rep + stride >= parts){ continue; }              // no sibling at this level: the subtree moves up as it is
					// the block of a subtree's first wave holds the subtree's sum ([..][1] in the wave that starts the pair)
					uint32_t *mine = wa.slab + ((uint64_t)(first_w + rep)*2 + (rep == 0 ? 1 : 0))*(PLANES*4*WAVE) + lane;
#pragma unroll
					for(int p = 0; p < PLANES; ++p){
#pragma unroll
						for(int d = 0; d < 4; ++d){ __hip_atomic_store(mine + (p*4 + d)*WAVE, plane[p][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
					}
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores are performed before the arrival says so
					uint32_t *node = wa.arrived + (uint64_t)(first_w + parent)*CWALK_LEVELS + level;
					uint32_t old = 0;
					if(lane == 0){ old = __hip_atomic_fetch_add(node, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
					old = __builtin_amdgcn_readfirstlane(old);
					if(old == 0){ emit = false; break; }                   // the sibling is still at work: it will take this sum along
					const uint32_t sib = (rep == parent) ? rep + stride : parent;
					const uint32_t *theirs = wa.slab + ((uint64_t)(first_w + sib)*2 + (sib == 0 ? 1 : 0))*(PLANES*4*WAVE) + lane;
					planes_accumulate<PLANES>(plane, PLANES, [&](int p) -> u32x4 {
						u32x4 v;
#pragma unroll
						for(int d = 0; d < 4; ++d){ v[d] = __hip_atomic_load(theirs + (p*4 + d)*WAVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
						return v;
					});
					if(lane == 0){ __hip_atomic_store(node, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
					rep = parent;
				}
			}
			if(emit){
				if(TRUNC && n < nk){
					// kwage.cpp:478-481 per column: the columns that can still reach the threshold if every remaining k-mer matches go on
					const uint32_t thr = qthr[q], remaining = nk - n;
					u32x4 can = live ? reinterpret_cast<const u32x4*>(a.valid)[unit] : (u32x4)(0u);
					if(thr > remaining){ can &= planes_ge<PLANES>(plane, thr - remaining); }
					const uint32_t gb = group_bits(__ballot((can.x | can.y | can.z | can.w) != 0));
					if(gb && !count_hand_over<PLANES>(ra, cc, ci, cu, q, c, gb, plane, n, nk, p0*NH)){
						// the lists are full: this wave counts the pair's remaining k-mers itself and reports it (correct, not balanced)
						count_kmers_prefetch<PLANES, NH, (PLANES >= 14) ? 8 : 4>(a.db, a.stride, rows + (p0 + n)*NH, nk - n, unit, plane);
						emit_count_hits_buffered<PLANES>(a, hbuf, hst, q, unit, plane, thr, (uint64_t)q*a.runs_per_query + c, live);
					}
				}
				else{ emit_count_hits_buffered<PLANES>(a, hbuf, hst, q, unit, plane, qthr[q], (uint64_t)q*a.runs_per_query + c, live); }
			}
		}
		s += take;
		if(j1 == npos && c + 1 == ct){ ++q; }
	}
	wave_hits_flush(a, hbuf, hst);          // one reservation for everything the wave found
	if(TRUNC){
		refine_none_clusters(ra, cc.next, cc.end);
		refine_none_units(ra, cu.next, cu.end);
	}
}

// Narrow databases, count path: G queries per wave (see and_narrow_kernel).  A shorter k-mer list is padded
// with all-zero matches (counting is not idempotent, so padded steps must add nothing).
template <int PLANES, int NH, int G>
__global__ __launch_bounds__(SEARCH_THREADS) void count_narrow_kernel(SearchArgs a)
{
	constexpr int KPS = 8;
	constexpr uint32_t LG = WAVE/G;
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint32_t l = lane % LG;
	const uint64_t tile = (uint64_t)blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6);
	const uint64_t q64 = tile*G + lane/LG;
	const bool has_q = (q64 < a.n_queries);
	const uint32_t q = has_q ? (uint32_t)q64 : 0;
	const uint32_t nk = has_q ? a.nkmer[q] : 0;
	const bool active = (nk != 0) && (l < a.units_per_row);
	const uint32_t *rq = a.rows + a.pos_off[q]*NH;
	const uint32_t unit = (l < a.units_per_row) ? l : 0;

	u32x4 plane[PLANES];
#pragma unroll
	for(int p = 0; p < PLANES; ++p){ plane[p] = (u32x4)(0u); }

	// Eight k-mers per step, and the NEXT step's rows requested before the current step's matches are added up: a narrow
	// launch has few waves (10 k queries of 1 kb against one 2048-column file: 2500, ten per CU), so the bytes in flight
	// come from the loads per wave and nothing else covers the adders' time (4 per step, not pipelined: 4.2 TB/s where the
	// AND form's 8 reach 5.6-5.7; 8 per step: 4.7).
	auto fetch = [&](uint32_t i, u32x4 (&m)[KPS]) {
#pragma unroll
		for(int u = 0; u < KPS; ++u){
			const bool real = (i + u < nk);
			const uint32_t kk = real ? (i + u) : (nk - 1);
			u32x4 x = load16<true>(reinterpret_cast<const u32x4*>(a.db + (uint64_t)rq[kk*NH]*a.stride) + unit);
#pragma unroll
			for(int h = 1; h < NH; ++h){
				x &= load16<true>(reinterpret_cast<const u32x4*>(a.db + (uint64_t)rq[kk*NH + h]*a.stride) + unit);
			}
			m[u] = real ? x : (u32x4)(0u);
		}
	};
	if(active){
		u32x4 cur[KPS], nxt[KPS];
		fetch(0, cur);
		for(uint32_t i = KPS; __any(i < nk); i += KPS){
			fetch(i, nxt);
			planes_add8<PLANES>(plane, cur);
#pragma unroll
			for(int u = 0; u < KPS; ++u){ cur[u] = nxt[u]; }
		}
		planes_add8<PLANES>(plane, cur);
	}
	__shared__ WgHitScratch wg_scratch;
	emit_count_hits<PLANES>(a, q, unit, plane, a.qthr[q], blockIdx.x, active, &wg_scratch);      // every wave of the workgroup gets here, exactly once
}

// Second pass of the segmented count: add the per-segment bit-sliced counters (ripple-carry adders across
// planes, bit-parallel over columns), then threshold + emit.  One workgroup per (query, 64 units); its
// COMBINE_WAVES waves each add every COMBINE_WAVES-th segment -- loads of different segments are independent,
// so they are all in flight together -- and the partial sums meet in LDS as a binary tree.  The slab holds
// `seg_planes` planes per segment (enough for seg_kmers), the result PLANES (enough for the whole query).
static constexpr int COMBINE_WAVES = 4;

template <int PLANES>
__global__ __launch_bounds__(COMBINE_WAVES*WAVE) void count_combine_kernel(SearchArgs a, uint32_t seg_planes)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char combine_lds[];     // (COMBINE_WAVES/2) x PLANES x 64 x 16 B
	u32x4 (*red)[PLANES][WAVE] = reinterpret_cast<u32x4 (*)[PLANES][WAVE]>(combine_lds);
	const uint32_t lane = threadIdx.x & (WAVE - 1);
	const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t u0 = blockIdx.x*WAVE + lane;
	const uint32_t q = blockIdx.y;
	const uint32_t n = a.nkmer[q];
	if(n == 0){ return; }                                  // uniform per workgroup
	const bool on = (u0 < a.units_per_row);
	const uint32_t unit = on ? u0 : 0;
	const uint32_t nseg = (n + a.seg_kmers - 1)/a.seg_kmers;
	const u32x4 *slab = reinterpret_cast<const u32x4*>(a.partial) + (uint64_t)q*a.segs*seg_planes*a.units_per_row + unit;
	u32x4 plane[PLANES];
#pragma unroll
	for(int p = 0; p < PLANES; ++p){ plane[p] = (u32x4)(0u); }
	for(uint32_t sg = w; sg < nseg; sg += COMBINE_WAVES){
		const u32x4 *s2 = slab + (uint64_t)sg*seg_planes*a.units_per_row;
		planes_accumulate<PLANES>(plane, (int)seg_planes, [&](int p){ return s2[(uint64_t)p*a.units_per_row]; });
	}
#pragma unroll
	for(int half = COMBINE_WAVES/2; half >= 1; half >>= 1){
		if(w >= (uint32_t)half && w < 2u*half){
#pragma unroll
			for(int p = 0; p < PLANES; ++p){ red[w - half][p][lane] = plane[p]; }
		}
		__syncthreads();
		if(w < (uint32_t)half){
			planes_accumulate<PLANES>(plane, PLANES, [&](int p){ return red[w][p][lane]; });
		}
		__syncthreads();
	}
	if(w == 0){ emit_count_hits<PLANES>(a, q, unit, plane, a.qthr[q], (uint64_t)q*a.runs_per_query + blockIdx.x, on); }
}

// Streaming read of the matrix: the box's achievable HBM read rate, reported beside every roofline number.  Every wave
// walks a contiguous region of its own, eight 1-KiB loads in flight (tools/micro/power_probe.hip: this pattern streams
// 6.9 TB/s where a grid-stride walk -- round 1/2's probe -- measured 6.4 on the same box; a random-row gather reaches
// 6.7 TB/s of touched bytes).
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4 *src, uint64_t n16, uint32_t *sink)
{
	u32x4 acc = (u32x4)(0u);
	const uint64_t nwaves = (uint64_t)gridDim.x*(blockDim.x/WAVE);
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/WAVE) + (threadIdx.x >> 6);
	const uint64_t per = (n16/nwaves)/(WAVE*8)*(WAVE*8);                 // whole 8 KiB steps; the remainder is left unread
	const u32x4 *p = src + wave*per + (threadIdx.x & (WAVE - 1));
	for(uint64_t i = 0; i < per; i += WAVE*8){
		u32x4 a[8];
#pragma unroll
		for(int u = 0; u < 8; ++u){ a[u] = __builtin_nontemporal_load(p + i + u*WAVE); }
#pragma unroll
		for(int u = 0; u < 8; ++u){ acc ^= a[u]; }
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u){ *sink = 1; }   // keep the loads alive
}

}  // namespace kwage

#endif
