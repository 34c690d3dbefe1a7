// kwage_amd/csrc/loader_kernels.hpp -- gfx950 kernels of the database loader (loader.hip): placing staged file rows in
// the matrix, copying locked file windows side by side, synthetic columns, single-bit writes and row read-back.
#ifndef KWAGE_AMD_LOADER_KERNELS_HPP
#define KWAGE_AMD_LOADER_KERNELS_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kwage {

static constexpr int LOADER_WAVE = 64;                 // lanes of a gfx950 wavefront
typedef uint32_t dwords4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// database construction / inspection kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t &x)
{
	x += 0x9E3779B97F4A7C15ull;
	uint64_t z = x;
	z = (z ^ (z >> 30))*0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27))*0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

// 64 i.i.d. Bernoulli(q8/256) bits: fold random words along the binary expansion of q8
// (bit set -> OR, bit clear -> AND), least significant set bit first.
__device__ __forceinline__ uint64_t bernoulli64(uint64_t key, uint32_t q8)
{
	if(q8 >= 256){ return ~0ull; }
	if(q8 == 0){ return 0; }
	uint64_t x = key;
	uint64_t r = 0;
	bool started = false;
	for(int b = 0; b < 8; ++b){
		const bool bit = (q8 >> b) & 1u;
		if(!started){
			if(bit){ r = splitmix64(x); started = true; }
		}
		else{
			const uint64_t d = splitmix64(x);
			r = bit ? (r | d) : (r & d);
		}
	}
	return r;
}

// Fill bytes [byte0, byte0+nbytes) of every row with random bits. byte0 is 8-byte aligned.
__global__ void fill_random_kernel(uint8_t *db, uint64_t stride, uint64_t nrows, uint64_t byte0,
                                   uint64_t nbytes, uint64_t seed, uint32_t q8)
{
	const uint64_t words_per_row = (nbytes + 7)/8;
	const uint64_t total = nrows*words_per_row;
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x*blockDim.x){
		const uint64_t r = i / words_per_row;
		const uint64_t w = i % words_per_row;
		const uint64_t key = seed ^ (r*0xD1342543DE82EF95ull) ^ ((byte0/8 + w)*0xA24BAED4963EE407ull);
		const uint64_t bits = bernoulli64(key, q8);
		uint8_t *dst = db + r*stride + byte0 + w*8;
		const uint64_t remain = nbytes - w*8;
		if(remain >= 8){
			*reinterpret_cast<uint64_t*>(dst) = bits;
		}
		else{
			for(uint64_t b = 0; b < remain; ++b){ dst[b] = (uint8_t)(bits >> (8*b)); }
		}
	}
}

// Copy a contiguous block of rows (src: nrows x width) into the strided matrix at byte offset byte0.
__global__ void place_rows_kernel(uint8_t *db, uint64_t stride, uint64_t row0, uint64_t byte0,
                                  const uint8_t *src, uint64_t src_stride, uint64_t width, uint64_t nrows)
{
	if(((width | byte0 | src_stride) & 3ull) == 0 && ((uintptr_t)src & 3ull) == 0){
		const uint64_t wpr = width/4;
		const uint64_t total = nrows*wpr;
		for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x*blockDim.x){
			const uint64_t r = i / wpr, w = i % wpr;
			*reinterpret_cast<uint32_t*>(db + (row0 + r)*stride + byte0 + 4*w) =
				*reinterpret_cast<const uint32_t*>(src + r*src_stride + 4*w);
		}
	}
	else{
		const uint64_t total = nrows*width;
		for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x*blockDim.x){
			const uint64_t r = i / width, b = i % width;
			db[(row0 + r)*stride + byte0 + b] = src[r*src_stride + b];
		}
	}
}

// Loader, direct path.  Rows of up to LOAD_GANG files -- windows of `.db` files locked in host memory, read over PCIe;
// only dword aligned: the body of a file starts at byte 44 -- to rows row0.. of the strided matrix, every file at its
// own byte column (16-byte aligned, adjacent for full 2048-column files).  UB = bytes per lane: 16 when every row
// length is a multiple of 16, else 4.  Consecutive lanes write consecutive bytes of ONE matrix row across the files
// of the gang, so a 100 KB-wide matrix is written 4 KiB at a time instead of 256 bytes at a time (one file per
// kernel filled a 105 GB matrix at 14 GB/s: every 256-byte piece opened another DRAM page and another TLB entry).
static constexpr uint32_t LOAD_GANG_MAX = 16;
struct GangSource { const uint8_t *src; uint64_t byte0; uint64_t width; };        // window of one file, its byte column in the matrix, its row length
struct GangArgs { GangSource f[LOAD_GANG_MAX]; uint32_t n; };

typedef dwords4 dwords4_unaligned __attribute__((aligned(4)));

// One work item = 64 lanes x UB consecutive bytes of ONE file's window, and the four waves of a workgroup take four
// consecutive items of the same file: a workgroup reads a contiguous 4 KiB (one host page of the page cache) exactly as
// a plain sequential copy would.  Consecutive workgroups take the same stretch of consecutive files, so the pieces that
// are neighbours in a matrix row are written at about the same time, by neighbouring workgroups, into the same DRAM
// pages and through the same TLB entries.
template <int UB>
__global__ __launch_bounds__(256) void copy_rows_gang_kernel(uint8_t *db, uint64_t stride, uint64_t row0, GangArgs ga, uint64_t nrows, uint64_t items)
{
	const uint32_t lane = threadIdx.x & (LOADER_WAVE - 1);
	const uint64_t nwaves = ((uint64_t)gridDim.x*blockDim.x) >> 6;
	for(uint64_t it = ((uint64_t)blockIdx.x*blockDim.x + threadIdx.x) >> 6; it < items; it += nwaves){
		// it = ((stretch*n + file)*4 + quarter): quarter = wave within the workgroup
		const uint32_t fi = __builtin_amdgcn_readfirstlane((uint32_t)((it >> 2) % ga.n));
		const uint64_t chunk = ((it >> 2) / ga.n)*4 + (it & 3);
		const uint64_t width = ga.f[fi].width;
		const uint64_t off = chunk*(LOADER_WAVE*UB) + (uint64_t)lane*UB;        // byte offset within this file's window
		if(off < nrows*width){
			const uint64_t r = off / width, col = off % width;           // UB divides width: a lane never straddles two rows
			const uint8_t *s = ga.f[fi].src + off;
			uint8_t *d = db + (row0 + r)*stride + ga.f[fi].byte0 + col;
			if(UB == 16){ *reinterpret_cast<dwords4*>(d) = *reinterpret_cast<const dwords4_unaligned*>(s); }
			else{ *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<const uint32_t*>(s); }
		}
	}
}

__global__ void set_bits_kernel(uint8_t *db, uint64_t stride, const uint32_t *rows, const uint64_t *cols, uint64_t n)
{
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x*blockDim.x){
		const uint64_t c = cols[i];
		uint32_t *word = reinterpret_cast<uint32_t*>(db + (uint64_t)rows[i]*stride + (c/32)*4);
		atomicOr(word, 1u << (c % 32));      // little endian: bit c%8 of byte c/8 (bloom.h:162)
	}
}

__global__ void gather_rows_kernel(const uint8_t *db, uint64_t stride, const uint32_t *rows, uint64_t n,
                                   uint64_t row_bytes, uint8_t *out)
{
	const uint64_t total = n*row_bytes;
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x*blockDim.x){
		const uint64_t r = i / row_bytes, b = i % row_bytes;
		out[i] = db[(uint64_t)rows[r]*stride + b];
	}
}


// Share of set bits among the real columns of `n_sample` rows spread evenly over the matrix: one workgroup per sampled
// row.  What the early exit at threshold < 1 needs to know before it can rule a column out after the first k-mers of a
// long query (engine.hip: the truncated count walk) -- an estimate, never a condition of correctness.
__global__ __launch_bounds__(256) void density_probe_kernel(const uint8_t *db, uint64_t stride, uint64_t nrows, const uint8_t *valid, uint32_t n_sample,
                                                            unsigned long long *set_bits)
{
	const uint64_t row = (uint64_t)blockIdx.x*nrows/n_sample;
	const dwords4 *r = reinterpret_cast<const dwords4*>(db + row*stride), *v = reinterpret_cast<const dwords4*>(valid);
	unsigned long long c = 0;
	for(uint64_t u = threadIdx.x; u < stride/16; u += blockDim.x){
		const dwords4 x = r[u] & v[u];
		c += __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w);
	}
	for(int d = 32; d >= 1; d >>= 1){ c += __shfl_down(c, d); }
	if((threadIdx.x & 63) == 0 && c){ atomicAdd(set_bits, c); }
}

// The DENSEST column among the same sampled rows: one thread per byte of a row (eight columns), its eight counts over the
// n_sample rows, the largest of them into *max_count.  Columns are not equally dense -- every sample's filter has its own
// fill -- and a bound that rules out the average column leaves every 128-byte group alive that holds a denser one.
__global__ __launch_bounds__(256) void column_density_probe_kernel(const uint8_t *db, uint64_t stride, uint64_t nrows, const uint8_t *valid, uint64_t row_bytes,
                                                                   uint32_t n_sample, unsigned int *max_count)
{
	const uint64_t b = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x;
	if(b >= row_bytes){ return; }
	const uint32_t real = valid[b];
	uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	for(uint32_t i = 0; i < n_sample; ++i){
		const uint32_t x = db[((uint64_t)i*nrows/n_sample)*stride + b] & real;
#pragma unroll
		for(int k = 0; k < 8; ++k){ cnt[k] += (x >> k) & 1u; }
	}
	uint32_t m = 0;
#pragma unroll
	for(int k = 0; k < 8; ++k){ m = max(m, cnt[k]); }
	if(m){ atomicMax(max_count, m); }
}


// The gather kernels' access pattern on an empty block, to compare candidate PLACEMENTS of a matrix (loader.hip,
// choose_placement): every wave reads `rows_per_wave` pseudo-random rows, four at a time, `chunks` KiB-steps of each
// (`lanes` lanes x 16 B per step, never past the row's stride).  One workgroup of 8 waves per CU (dynamic LDS pad).
__global__ __launch_bounds__(512) void placement_probe_kernel(const dwords4 *base, uint64_t nrows, uint64_t stride16, uint32_t chunks, uint32_t lanes,
                                                              uint32_t rows_per_wave, uint32_t windows, uint32_t *sink)
{
	extern __shared__ uint32_t placement_pad[];
	dwords4 acc = (dwords4)(0u);
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/LOADER_WAVE) + (threadIdx.x/LOADER_WAVE);
	const uint32_t lane = threadIdx.x & (LOADER_WAVE - 1);
	uint64_t x = wave*0x9E3779B97F4A7C15ull + 12345;
	for(uint32_t r = 0; r < rows_per_wave; r += 4){
		const dwords4 *p[4];
#pragma unroll
		for(int u = 0; u < 4; ++u){
			x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
			// windows > 1: all waves draw from the same 1/windows of the block at the same time, window after window
			const uint64_t per_win = nrows/windows, win = (uint64_t)(r + u)*windows/rows_per_wave;
			const uint64_t row = win*per_win + (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((x*0x2545F4914F6CDD1Dull) >> 33)) % per_win;
			p[u] = base + row*stride16 + (lane < lanes ? lane : 0);
		}
		for(uint32_t j = 0; j < chunks; ++j){
			dwords4 a[4];
#pragma unroll
			for(int u = 0; u < 4; ++u){ a[u] = __builtin_nontemporal_load(p[u] + (uint64_t)j*LOADER_WAVE); }
#pragma unroll
			for(int u = 0; u < 4; ++u){ acc ^= a[u]; }
		}
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u){ *sink = placement_pad[0]; }   // keep the loads alive
}

}  // namespace kwage

#endif
