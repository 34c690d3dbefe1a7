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
//   run_block_sums   records per block of 1024 table entries
//   run_scan_sums    exclusive prefix over the blocks (one workgroup)
//   run_place        per block: prefix over its entries = every run's place in the ordered list; the non-empty runs
//                    are copied there, a wave per run, dword by dword (coalesced)
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

__global__ __launch_bounds__(ORDER_THREADS) void run_block_sums(const unsigned long long *__restrict__ runs, uint64_t n_runs, unsigned long long *__restrict__ sums)
{
	__shared__ unsigned long long wave_tot[ORDER_THREADS/64];
	const uint64_t i0 = (uint64_t)blockIdx.x*ENTRIES_PER_BLOCK + (uint64_t)threadIdx.x*ENTRIES_PER_THREAD;
	unsigned long long c = 0;
#pragma unroll
	for(int k = 0; k < ENTRIES_PER_THREAD; ++k){ if(i0 + k < n_runs){ c += run_records(runs[i0 + k]); } }
	const unsigned long long incl = block_inclusive_scan<ORDER_THREADS/64>(c, wave_tot);
	if(threadIdx.x == ORDER_THREADS - 1){ sums[blockIdx.x] = incl; }
}

// sums[0 .. nblocks) -> their exclusive prefix, in place; sums[nblocks] = the total
__global__ __launch_bounds__(SCAN_THREADS) void run_scan_sums(unsigned long long *sums, uint64_t nblocks)
{
	__shared__ unsigned long long wave_tot[SCAN_THREADS/64];
	const uint64_t per = (nblocks + SCAN_THREADS - 1)/SCAN_THREADS;
	const uint64_t lo = std::min<uint64_t>(nblocks, (uint64_t)threadIdx.x*per), hi = std::min<uint64_t>(nblocks, lo + per);
	unsigned long long mine = 0;
	for(uint64_t i = lo; i < hi; ++i){ mine += sums[i]; }
	const unsigned long long incl = block_inclusive_scan<SCAN_THREADS/64>(mine, wave_tot);
	unsigned long long at = incl - mine;
	for(uint64_t i = lo; i < hi; ++i){ const unsigned long long v = sums[i]; sums[i] = at; at += v; }
	if(threadIdx.x == SCAN_THREADS - 1){ sums[nblocks] = incl; }
}

__global__ __launch_bounds__(ORDER_THREADS) void run_place(const unsigned long long *__restrict__ runs, uint64_t n_runs, const unsigned long long *__restrict__ block_base,
                                                           const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint64_t n_hits)
{
	__shared__ unsigned long long wave_tot[ORDER_THREADS/64];
	__shared__ unsigned long long from[ENTRIES_PER_BLOCK], to[ENTRIES_PER_BLOCK];
	__shared__ uint32_t len[ENTRIES_PER_BLOCK];
	__shared__ uint32_t listed;
	if(threadIdx.x == 0){ listed = 0; }
	const uint64_t i0 = (uint64_t)blockIdx.x*ENTRIES_PER_BLOCK + (uint64_t)threadIdx.x*ENTRIES_PER_THREAD;
	unsigned long long e[ENTRIES_PER_THREAD];
	unsigned long long c = 0;
#pragma unroll
	for(int k = 0; k < ENTRIES_PER_THREAD; ++k){
		e[k] = (i0 + k < n_runs) ? runs[i0 + k] : 0ull;
		c += run_records(e[k]);
	}
	const unsigned long long incl = block_inclusive_scan<ORDER_THREADS/64>(c, wave_tot);     // (its barriers also publish `listed`)
	unsigned long long at = block_base[blockIdx.x] + incl - c;
#pragma unroll
	for(int k = 0; k < ENTRIES_PER_THREAD; ++k){
		const uint32_t n = run_records(e[k]);
		if(n){
			const uint32_t slot = atomicAdd(&listed, 1u);            // the order inside the list is free
			from[slot] = run_first(e[k]);
			to[slot] = at;
			len[slot] = n;
			at += n;
		}
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for(uint32_t r = w; r < listed; r += ORDER_THREADS/64){
		const unsigned long long f = from[r], t = to[r];
		const uint32_t n = len[r];
		if(f + n > n_hits || t + n > n_hits){ continue; }             // (cannot happen while the table and the counter agree; never write outside)
		const uint32_t *s = src + f*3;
		uint32_t *d = dst + t*3;
		for(uint32_t x = lane; x < n*3u; x += 64){ d[x] = s[x]; }
	}
}

inline uint64_t align_up(uint64_t x){ return (x + 255) & ~255ull; }

}  // namespace

uint64_t hit_order_scratch_bytes(uint64_t n_hits, uint64_t n_runs)
{
	const uint64_t nblocks = (n_runs + ENTRIES_PER_BLOCK - 1)/ENTRIES_PER_BLOCK;
	return align_up(n_hits*sizeof(kwage_hit)) + align_up((nblocks + 1)*sizeof(unsigned long long));
}

int order_hits_by_runs(void *stream, const kwage_hit *d_hits, uint64_t n_hits, const void *runs, uint64_t n_runs,
                       void *scratch, uint64_t scratch_bytes, kwage_hit **d_ordered, const uint64_t **d_total)
{
	if(!scratch || scratch_bytes < hit_order_scratch_bytes(n_hits, n_runs) || !runs || n_runs == 0){
		return fail(KWAGE_ERR_ARG, "hit order: no run table or scratch block too small");
	}
	const uint64_t nblocks = (n_runs + ENTRIES_PER_BLOCK - 1)/ENTRIES_PER_BLOCK;
	if(nblocks > 0x7FFFFFFFull){ return fail(KWAGE_ERR_ARG, "hit order: run table too long"); }
	hipStream_t st = (hipStream_t)stream;
	kwage_hit *ordered = (kwage_hit*)scratch;
	unsigned long long *sums = (unsigned long long*)((char*)scratch + align_up(n_hits*sizeof(kwage_hit)));
	const unsigned long long *table = (const unsigned long long*)runs;
	hipLaunchKernelGGL(run_block_sums, dim3((uint32_t)nblocks), dim3(ORDER_THREADS), 0, st, table, n_runs, sums);
	hipLaunchKernelGGL(run_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, sums, nblocks);
	hipLaunchKernelGGL(run_place, dim3((uint32_t)nblocks), dim3(ORDER_THREADS), 0, st, table, n_runs, (const unsigned long long*)sums,
	                   (const uint32_t*)d_hits, (uint32_t*)ordered, n_hits);
	const hipError_t e = hipGetLastError();
	if(e != hipSuccess){ return fail(KWAGE_ERR_DEVICE, "hit order: %s", hipGetErrorString(e)); }
	*d_ordered = ordered;
	*d_total = (const uint64_t*)(sums + nblocks);
	return KWAGE_OK;
}

}  // namespace kwage
