// kwage_amd/csrc/hit_sort.hip -- ordering a large hit list by (query, column) on the device.
//
// The gather kernels append hits in the order their waves finish.  The C ABI returns them sorted by
// (query, column) (include/kwage_amd.h, kwage_result::hits; the reference's own order among its per-thread
// result lists is the one sort.h:22-27 leaves it with).  Lists of up to a few thousand records are sorted by
// the host while it assembles the result; above that the host sort was the largest part of a search call
// (1.5 M hits: 25 ms next to a 38 ms gather), so the list is sorted where it lies:
//   pack    (query, column, num_match) records -> key = query << column_bits | column, value = num_match
//   sort    rocPRIM's device radix sort over the key's low query_bits + column_bits bits only
//   unpack  sorted keys and values -> records, in place of the unsorted list
// Keys are unique ((query, column) pairs are), so stability does not matter.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "internal.h"

namespace kwage {

namespace {

__global__ __launch_bounds__(256) void pack_hits_kernel(const kwage_hit *__restrict__ hits, uint64_t n, uint32_t column_bits,
                                                        uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x*blockDim.x){
		const kwage_hit h = hits[i];
		keys[i] = ((uint64_t)h.query << column_bits) | h.column;
		vals[i] = h.num_match;
	}
}

__global__ __launch_bounds__(256) void unpack_hits_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t n,
                                                          uint32_t column_bits, kwage_hit *__restrict__ hits)
{
	const uint64_t column_mask = (1ull << column_bits) - 1;
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x*blockDim.x){
		const uint64_t k = keys[i];
		kwage_hit h;
		h.query = (uint32_t)(k >> column_bits);
		h.column = (uint32_t)(k & column_mask);
		h.num_match = vals[i];
		hits[i] = h;
	}
}

uint32_t bits_for(uint64_t count)      // bits that hold 0 .. count-1
{
	if(count <= 1){ return 0; }
	uint32_t b = 0;
	while(b < 64 && ((count - 1) >> b)){ ++b; }
	return b;
}

inline uint64_t align_up(uint64_t x){ return (x + 255) & ~255ull; }

struct Layout {
	uint64_t keys[2], vals[2], temp, total;
	size_t temp_bytes;
};

// Where the two key and value buffers and rocPRIM's own storage lie in the scratch block.
int layout_for(uint64_t n, unsigned end_bit, Layout *l)
{
	rocprim::double_buffer<uint64_t> keys(nullptr, nullptr);
	rocprim::double_buffer<uint32_t> vals(nullptr, nullptr);
	size_t temp = 0;
	if(rocprim::radix_sort_pairs(nullptr, temp, keys, vals, (size_t)n, 0u, end_bit) != hipSuccess){
		return fail(KWAGE_ERR_DEVICE, "hit sort: sizing the device radix sort failed");
	}
	uint64_t at = 0;
	for(int i = 0; i < 2; ++i){ l->keys[i] = at; at += align_up(n*sizeof(uint64_t)); }
	for(int i = 0; i < 2; ++i){ l->vals[i] = at; at += align_up(n*sizeof(uint32_t)); }
	l->temp = at;
	l->temp_bytes = temp;
	l->total = at + align_up(temp);
	return KWAGE_OK;
}

// `column_span`: one more than the largest column index a hit can carry -- the group's column SPAN (files start at
// 16-byte boundaries, so it exceeds the number of valid columns), at most 2^32.
unsigned key_bits(uint32_t n_queries, uint64_t column_span, uint32_t *column_bits)
{
	*column_bits = std::min(32u, bits_for(column_span));
	return std::max(1u, *column_bits + bits_for(n_queries));
}

}  // namespace

int hit_sort_scratch_bytes(uint64_t n_hits, uint32_t n_queries, uint64_t column_span, uint64_t *bytes)
{
	uint32_t cb;
	Layout l;
	int rc = layout_for(n_hits, key_bits(n_queries, column_span, &cb), &l);
	if(rc){ return rc; }
	*bytes = l.total;
	return KWAGE_OK;
}

int sort_hits_on_device(void *stream, kwage_hit *d_hits, uint64_t n_hits, uint32_t n_queries, uint64_t column_span,
                        void *scratch, uint64_t scratch_bytes)
{
	if(n_hits < 2){ return KWAGE_OK; }
	hipStream_t st = (hipStream_t)stream;
	uint32_t cb;
	const unsigned end_bit = key_bits(n_queries, column_span, &cb);
	Layout l;
	int rc = layout_for(n_hits, end_bit, &l);
	if(rc){ return rc; }
	if(!scratch || scratch_bytes < l.total){ return fail(KWAGE_ERR_ARG, "hit sort: scratch block too small"); }
	char *base = (char*)scratch;
	rocprim::double_buffer<uint64_t> keys((uint64_t*)(base + l.keys[0]), (uint64_t*)(base + l.keys[1]));
	rocprim::double_buffer<uint32_t> vals((uint32_t*)(base + l.vals[0]), (uint32_t*)(base + l.vals[1]));
	const unsigned grid = (unsigned)std::min<uint64_t>((n_hits + 255)/256, 256*16);
	hipLaunchKernelGGL(pack_hits_kernel, dim3(grid), dim3(256), 0, st, (const kwage_hit*)d_hits, n_hits, cb, keys.current(), vals.current());
	hipError_t e = hipGetLastError();
	size_t temp = l.temp_bytes;
	if(e == hipSuccess){ e = rocprim::radix_sort_pairs(base + l.temp, temp, keys, vals, (size_t)n_hits, 0u, end_bit, st); }
	if(e == hipSuccess){
		hipLaunchKernelGGL(unpack_hits_kernel, dim3(grid), dim3(256), 0, st, (const uint64_t*)keys.current(), (const uint32_t*)vals.current(), n_hits, cb, d_hits);
		e = hipGetLastError();
	}
	if(e != hipSuccess){ return fail(KWAGE_ERR_DEVICE, "hit sort: %s", hipGetErrorString(e)); }
	return KWAGE_OK;
}

}  // namespace kwage
