// kwage_amd/csrc/hit_sort.hip -- ordering a large hit list by (query, column) on the device WITHOUT sorting it.
//
// The gather kernels append hits in the order their waves finish; the C ABI returns them sorted by (query, column)
// (include/kwage_amd.h, kwage_result::hits; the reference's own order among its per-thread result lists is the one
// sort.h:22-27 leaves it with).  Lists of up to a few thousand records are sorted by the host while it assembles the
// result.  Longer ones were radix-sorted on the device in rounds 2-3 (pack -> rocPRIM -> unpack: five 8-bit passes over
// 20 bytes per hit).  But the list is not an arbitrary permutation: every reservation of hit slots -- one wave's -- is a
// RUN of records that is already ascending by (query, column), and the runs of one search cover disjoint key ranges
// whose order the reserving wave knows: (query, column tile, step).  The kernels note `first slot << 16 | records` per
// run in a table indexed in key order (kernels.hpp, SearchArgs::runs), so ordering the list is
//   run_block_sums   records and non-empty runs per block of 1024 table entries
//   run_scan_sums    exclusive prefix of both over the blocks (one workgroup)
//   run_compact      per block: prefix over its entries = every run's place in the ordered list; the non-empty runs go,
//                    in key order, into a compact list (first slot in the raw list, first slot in the ordered list)
//   run_copy         one workgroup per 2048 records of the ORDERED list, whatever the runs' lengths (a hit-heavy list is
//                    a few thousand runs of 8192 records, C3's 150 k hits are 150 k runs of one): the runs that cover its
//                    stretch are found by binary search, their starts go to LDS, and every thread moves its records
// -- one pass over the table (8 bytes per wave of the gather launch) and ONE copy of the records, no library primitive.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "internal.h"

namespace kwage {

namespace {

constexpr int ORDER_THREADS = 256;
constexpr int ENTRIES_PER_THREAD = 4;
constexpr int ENTRIES_PER_BLOCK = ORDER_THREADS*ENTRIES_PER_THREAD;
constexpr int SCAN_THREADS = 1024;

__device__ __forceinline__ uint32_t run_records(unsigned long long e) { return (uint32_t)(e & 0xFFFFull); }
__device__ __forceinline__ unsigned long long run_first(unsigned long long e) { return e >> 16; }

// inclusive prefix sum over the workgroup's threads (wave scans by shuffle, wave totals through LDS)
template <int NWAVES>
__device__ __forceinline__ unsigned long long block_inclusive_scan(unsigned long long v, unsigned long long *wave_tot)
{
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
	for(int d = 1; d < 64; d <<= 1){
		const unsigned long long up = __shfl_up(v, d);
		if((int)lane >= d){ v += up; }
	}
	if(lane == 63){ wave_tot[w] = v; }
	__syncthreads();
	unsigned long long before = 0;
	for(uint32_t x = 0; x < w; ++x){ before += wave_tot[x]; }
	__syncthreads();
	return v + before;
}

// sums[b] = records of block b, sums[nblocks + 1 + b] = its non-empty runs
__global__ __launch_bounds__(ORDER_THREADS) void run_block_sums(const unsigned long long *__restrict__ runs, uint64_t n_runs, uint64_t nblocks, unsigned long long *__restrict__ sums)
{
	__shared__ unsigned long long wave_tot[ORDER_THREADS/64];
	const uint64_t i0 = (uint64_t)blockIdx.x*ENTRIES_PER_BLOCK + (uint64_t)threadIdx.x*ENTRIES_PER_THREAD;
	unsigned long long c = 0;       // records in the low 40 bits, non-empty runs above (a block holds at most 1024 x 65535 records)
#pragma unroll
	for(int k = 0; k < ENTRIES_PER_THREAD; ++k){
		if(i0 + k < n_runs){ const uint32_t n = run_records(runs[i0 + k]); c += n + (n ? (1ull << 40) : 0ull); }
	}
	const unsigned long long incl = block_inclusive_scan<ORDER_THREADS/64>(c, wave_tot);
	if(threadIdx.x == ORDER_THREADS - 1){
		sums[blockIdx.x] = incl & ((1ull << 40) - 1);
		sums[nblocks + 1 + blockIdx.x] = incl >> 40;
	}
}

// both halves of sums -> their exclusive prefixes, in place; sums[nblocks] = all records, sums[2*nblocks + 1] = all non-empty runs
__global__ __launch_bounds__(SCAN_THREADS) void run_scan_sums(unsigned long long *sums, uint64_t nblocks)
{
	__shared__ unsigned long long wave_tot[SCAN_THREADS/64];
	const uint64_t per = (nblocks + SCAN_THREADS - 1)/SCAN_THREADS;
	const uint64_t lo = std::min<uint64_t>(nblocks, (uint64_t)threadIdx.x*per), hi = std::min<uint64_t>(nblocks, lo + per);
	for(int half = 0; half < 2; ++half){
		unsigned long long *v = sums + (half ? nblocks + 1 : 0);
		unsigned long long mine = 0;
		for(uint64_t i = lo; i < hi; ++i){ mine += v[i]; }
		const unsigned long long incl = block_inclusive_scan<SCAN_THREADS/64>(mine, wave_tot);
		unsigned long long at = incl - mine;
		for(uint64_t i = lo; i < hi; ++i){ const unsigned long long x = v[i]; v[i] = at; at += x; }
		if(threadIdx.x == SCAN_THREADS - 1){ v[nblocks] = incl; }
		__syncthreads();
	}
}

// the non-empty runs in key order: cfrom[k] = the run's first slot in the raw list, cto[k] = its first slot in the ordered list
__global__ __launch_bounds__(ORDER_THREADS) void run_compact(const unsigned long long *__restrict__ runs, uint64_t n_runs, uint64_t nblocks,
                                                             const unsigned long long *__restrict__ sums, unsigned long long *__restrict__ cfrom, unsigned long long *__restrict__ cto)
{
	__shared__ unsigned long long wave_tot[ORDER_THREADS/64];
	const uint64_t i0 = (uint64_t)blockIdx.x*ENTRIES_PER_BLOCK + (uint64_t)threadIdx.x*ENTRIES_PER_THREAD;
	unsigned long long e[ENTRIES_PER_THREAD];
	unsigned long long c = 0;
#pragma unroll
	for(int k = 0; k < ENTRIES_PER_THREAD; ++k){
		e[k] = (i0 + k < n_runs) ? runs[i0 + k] : 0ull;
		const uint32_t n = run_records(e[k]);
		c += n + (n ? (1ull << 40) : 0ull);
	}
	const unsigned long long incl = block_inclusive_scan<ORDER_THREADS/64>(c, wave_tot);
	const unsigned long long excl = incl - c;
	unsigned long long at = sums[blockIdx.x] + (excl & ((1ull << 40) - 1));          // place in the ordered list
	unsigned long long k_out = sums[nblocks + 1 + blockIdx.x] + (excl >> 40);        // index in the compact list
#pragma unroll
	for(int k = 0; k < ENTRIES_PER_THREAD; ++k){
		const uint32_t n = run_records(e[k]);
		if(n){ cfrom[k_out] = run_first(e[k]); cto[k_out] = at; ++k_out; at += n; }
	}
}

constexpr uint32_t COPY_CHUNK = 2048;          // ordered records per workgroup of run_copy

__global__ __launch_bounds__(ORDER_THREADS) void run_copy(const unsigned long long *__restrict__ cfrom, const unsigned long long *__restrict__ cto, const unsigned long long *__restrict__ sums,
                                                          uint64_t nblocks, const kwage_hit *__restrict__ src, kwage_hit *__restrict__ dst, uint64_t n_hits)
{
	__shared__ unsigned long long lfrom[COPY_CHUNK], lto[COPY_CHUNK];
	__shared__ unsigned long long k_first;
	const unsigned long long n_ne = sums[2*nblocks + 1];          // non-empty runs
	const unsigned long long total = sums[nblocks];              // records the table accounts for
	const unsigned long long o0 = (unsigned long long)blockIdx.x*COPY_CHUNK;
	if(o0 >= total || n_ne == 0){ return; }
	const unsigned long long o1 = std::min<unsigned long long>(std::min<unsigned long long>(total, n_hits), o0 + COPY_CHUNK);
	if(threadIdx.x == 0){
		// the last run that starts at or before o0
		unsigned long long lo = 0, hi = n_ne;
		while(hi - lo > 1){
			const unsigned long long mid = lo + (hi - lo)/2;
			if(cto[mid] <= o0){ lo = mid; } else { hi = mid; }
		}
		k_first = lo;
	}
	__syncthreads();
	const unsigned long long k0 = k_first;
	// the runs that start inside the stretch follow k0: at most COPY_CHUNK - 1 of them (every run holds a record)
	for(uint32_t i = threadIdx.x; i < COPY_CHUNK; i += ORDER_THREADS){
		const unsigned long long k = k0 + i;
		const bool in = k < n_ne && (i == 0 || cto[k] < o1);
		lto[i] = in ? cto[k] : ~0ull;
		lfrom[i] = in ? cfrom[k] : 0ull;
	}
	__syncthreads();
	for(unsigned long long o = o0 + threadIdx.x; o < o1; o += ORDER_THREADS){
		uint32_t lo = 0, hi = COPY_CHUNK;                       // the last local run with lto <= o (lto ascends; unused entries are ~0)
		while(hi - lo > 1){
			const uint32_t mid = lo + (hi - lo)/2;
			if(lto[mid] <= o){ lo = mid; } else { hi = mid; }
		}
		const unsigned long long f = lfrom[lo] + (o - lto[lo]);
		if(f < n_hits){ dst[o] = src[f]; }
	}
}

inline uint64_t align_up(uint64_t x){ return (x + 255) & ~255ull; }

}  // namespace

uint64_t hit_order_scratch_bytes(uint64_t n_hits, uint64_t n_runs)
{
	const uint64_t nblocks = (n_runs + ENTRIES_PER_BLOCK - 1)/ENTRIES_PER_BLOCK;
	const uint64_t compact = std::min(n_runs, n_hits) + 1;
	return align_up(n_hits*sizeof(kwage_hit)) + align_up(2*(nblocks + 1)*sizeof(unsigned long long)) + 2*align_up(compact*sizeof(unsigned long long));
}

int order_hits_by_runs(void *stream, const kwage_hit *d_hits, uint64_t n_hits, const void *runs, uint64_t n_runs,
                       void *scratch, uint64_t scratch_bytes, kwage_hit **d_ordered, const uint64_t **d_total)
{
	if(!scratch || scratch_bytes < hit_order_scratch_bytes(n_hits, n_runs) || !runs || n_runs == 0){
		return fail(KWAGE_ERR_ARG, "hit order: no run table or scratch block too small");
	}
	const uint64_t nblocks = (n_runs + ENTRIES_PER_BLOCK - 1)/ENTRIES_PER_BLOCK;
	const uint64_t chunks = (n_hits + COPY_CHUNK - 1)/COPY_CHUNK;
	if(nblocks > 0x7FFFFFFFull || chunks > 0x7FFFFFFFull){ return fail(KWAGE_ERR_ARG, "hit order: list or run table too long"); }
	hipStream_t st = (hipStream_t)stream;
	const uint64_t compact = std::min(n_runs, n_hits) + 1;
	char *base = (char*)scratch;
	kwage_hit *ordered = (kwage_hit*)base;
	unsigned long long *sums = (unsigned long long*)(base + align_up(n_hits*sizeof(kwage_hit)));
	unsigned long long *cfrom = (unsigned long long*)((char*)sums + align_up(2*(nblocks + 1)*sizeof(unsigned long long)));
	unsigned long long *cto = (unsigned long long*)((char*)cfrom + align_up(compact*sizeof(unsigned long long)));
	const unsigned long long *table = (const unsigned long long*)runs;
	hipLaunchKernelGGL(run_block_sums, dim3((uint32_t)nblocks), dim3(ORDER_THREADS), 0, st, table, n_runs, nblocks, sums);
	hipLaunchKernelGGL(run_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, sums, nblocks);
	hipLaunchKernelGGL(run_compact, dim3((uint32_t)nblocks), dim3(ORDER_THREADS), 0, st, table, n_runs, nblocks, (const unsigned long long*)sums, cfrom, cto);
	hipLaunchKernelGGL(run_copy, dim3((uint32_t)chunks), dim3(ORDER_THREADS), 0, st, (const unsigned long long*)cfrom, (const unsigned long long*)cto,
	                   (const unsigned long long*)sums, nblocks, d_hits, ordered, n_hits);
	const hipError_t e = hipGetLastError();
	if(e != hipSuccess){ return fail(KWAGE_ERR_DEVICE, "hit order: %s", hipGetErrorString(e)); }
	*d_ordered = ordered;
	*d_total = (const uint64_t*)(sums + nblocks);
	return KWAGE_OK;
}

}  // namespace kwage
