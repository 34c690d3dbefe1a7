// Microbenchmark: what is the best HBM READ rate this box gives, over access shapes?  (Sets the practical
// ceiling the gather kernels are compared with.)  hipcc --offload-arch=gfx950 -O3 -o stream_variants stream_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){ printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } }while(0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void grid_stride(const u32x4 *src, uint64_t n16, uint32_t *sink)
{
	u32x4 acc = (u32x4)(0u);
	const uint64_t step = (uint64_t)gridDim.x*blockDim.x;
	uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x;
	for(; i + (U - 1)*step < n16; i += U*step){
		u32x4 a[U];
#pragma unroll
		for(int u = 0; u < U; ++u){ a[u] = NT ? __builtin_nontemporal_load(src + i + u*step) : src[i + u*step]; }
#pragma unroll
		for(int u = 0; u < U; ++u){ acc ^= a[u]; }
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u){ *sink = 1; }
}

// each WAVE walks its own contiguous region, U consecutive KiB per iteration
template <int U, bool NT>
__global__ __launch_bounds__(256) void wave_contig(const u32x4 *src, uint64_t n16, uint32_t *sink)
{
	u32x4 acc = (u32x4)(0u);
	const uint64_t nwaves = (uint64_t)gridDim.x*(blockDim.x/64);
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6);
	const uint64_t per = n16/nwaves/(64*U)*(64*U);
	const u32x4 *p = src + wave*per + (threadIdx.x & 63);
	for(uint64_t i = 0; i < per; i += 64*U){
		u32x4 a[U];
#pragma unroll
		for(int u = 0; u < U; ++u){ a[u] = NT ? __builtin_nontemporal_load(p + i + u*64) : p[i + u*64]; }
#pragma unroll
		for(int u = 0; u < U; ++u){ acc ^= a[u]; }
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u){ *sink = 1; }
}

// random 1-KiB-aligned rows of `row16` units (like the gather kernel: a wave reads whole rows, U in flight)
template <int U>
__global__ __launch_bounds__(256) void random_rows(const u32x4 *src, uint64_t nrows, uint32_t row16, uint64_t rows_per_wave, uint32_t *sink)
{
	u32x4 acc = (u32x4)(0u);
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63;
	uint64_t x = wave*0x9E3779B97F4A7C15ull + 12345;
	for(uint64_t r = 0; r < rows_per_wave; r += U){
		const u32x4 *p[U];
#pragma unroll
		for(int u = 0; u < U; ++u){
			x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
			const uint64_t row = __builtin_amdgcn_readfirstlane((uint32_t)((x*0x2545F4914F6CDD1Dull) >> 32)) % nrows;
			p[u] = src + row*row16;
		}
		for(uint32_t c = lane; c < row16; c += 64){
			u32x4 a[U];
#pragma unroll
			for(int u = 0; u < U; ++u){ a[u] = __builtin_nontemporal_load(p[u] + c); }
#pragma unroll
			for(int u = 0; u < U; ++u){ acc ^= a[u]; }
		}
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u){ *sink = 1; }
}

__device__ __forceinline__ uint64_t row_of(uint32_t q, uint32_t i, uint64_t nrows)
{
	uint64_t x = ((uint64_t)q << 32 | i)*0x9E3779B97F4A7C15ull + 0x1234567;
	x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
	return x % nrows;
}

// the and_kernel shape: a wave = (query, tile of VEC KiB of every row); tiles of a query are adjacent waves
template <int VEC, int U>
__global__ __launch_bounds__(256) void tiled_rows(const u32x4 *src, uint64_t nrows, uint32_t row16, uint32_t nq, uint32_t nk, uint32_t *sink)
{
	const uint32_t tiles = (row16 + 64*VEC - 1)/(64*VEC);
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6);
	if(wave >= (uint64_t)nq*tiles){ return; }
	const uint32_t q = __builtin_amdgcn_readfirstlane((uint32_t)(wave/tiles)), c = __builtin_amdgcn_readfirstlane((uint32_t)(wave % tiles));
	const uint32_t lane = threadIdx.x & 63;
	uint32_t unit[VEC];
	for(int v = 0; v < VEC; ++v){ unit[v] = min(c*64*VEC + v*64 + lane, row16 - 1); }
	u32x4 acc[VEC];
	for(int v = 0; v < VEC; ++v){ acc[v] = ~(u32x4)(0u); }
	for(uint32_t i = 0; i < nk; i += U){
		u32x4 a[U][VEC];
#pragma unroll
		for(int u = 0; u < U; ++u){
			const u32x4 *p = src + row_of(q, i + u, nrows)*row16;
#pragma unroll
			for(int v = 0; v < VEC; ++v){ a[u][v] = __builtin_nontemporal_load(p + unit[v]); }
		}
#pragma unroll
		for(int u = 0; u < U; ++u){
#pragma unroll
			for(int v = 0; v < VEC; ++v){ acc[v] &= a[u][v]; }
		}
	}
	u32x4 t = acc[0];
	for(int v = 1; v < VEC; ++v){ t ^= acc[v]; }
	if((t.x ^ t.y ^ t.z ^ t.w) == 0x12345678u){ *sink = 1; }
}

// balanced tiles: the row's KiB-chunks are dealt to ceil(chunks/VMAX) waves as evenly as possible (no mostly empty last tile)
template <int VMAX, int U>
__global__ __launch_bounds__(256) void balanced_rows(const u32x4 *src, uint64_t nrows, uint32_t row16, uint32_t nq, uint32_t nk, uint32_t *sink)
{
	const uint32_t chunks = (row16 + 63)/64, wpq = (chunks + VMAX - 1)/VMAX, base = chunks/wpq, extra = chunks % wpq;
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6);
	if(wave >= (uint64_t)nq*wpq){ return; }
	const uint32_t q = __builtin_amdgcn_readfirstlane((uint32_t)(wave/wpq)), w = __builtin_amdgcn_readfirstlane((uint32_t)(wave % wpq));
	const uint32_t c0 = w*base + min(w, extra), nv = base + (w < extra ? 1u : 0u);
	const uint32_t lane = threadIdx.x & 63;
	uint32_t unit[VMAX];
	u32x4 acc[VMAX];
	for(int v = 0; v < VMAX; ++v){ unit[v] = min((c0 + min((uint32_t)v, nv - 1))*64 + lane, row16 - 1); acc[v] = ~(u32x4)(0u); }
	for(uint32_t i = 0; i < nk; i += U){
		u32x4 a[U][VMAX];
#pragma unroll
		for(int u = 0; u < U; ++u){
			const u32x4 *p = src + row_of(q, i + u, nrows)*row16;
#pragma unroll
			for(int v = 0; v < VMAX; ++v){ if(v < (int)nv){ a[u][v] = __builtin_nontemporal_load(p + unit[v]); } else { a[u][v] = ~(u32x4)(0u); } }
		}
#pragma unroll
		for(int u = 0; u < U; ++u){
#pragma unroll
			for(int v = 0; v < VMAX; ++v){ acc[v] &= a[u][v]; }
		}
	}
	u32x4 t = acc[0];
	for(int v = 1; v < VMAX; ++v){ t ^= acc[v]; }
	if((t.x ^ t.y ^ t.z ^ t.w) == 0x12345678u){ *sink = 1; }
}

// whole-row walk: a wave = (query, segment of its k-mers); U rows in flight, walked chunk by chunk; CH accumulators
template <int CH, int U>
__global__ __launch_bounds__(256) void walk_rows(const u32x4 *src, uint64_t nrows, uint32_t row16, uint32_t nq, uint32_t nk, uint32_t segs, uint32_t *sink)
{
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6);
	if(wave >= (uint64_t)nq*segs){ return; }
	const uint32_t q = __builtin_amdgcn_readfirstlane((uint32_t)(wave/segs)), sg = __builtin_amdgcn_readfirstlane((uint32_t)(wave % segs));
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t per = (nk + segs - 1)/segs, k0 = sg*per, k1 = min(nk, k0 + per);
	u32x4 acc[CH];
	uint32_t unit[CH];
	for(int c = 0; c < CH; ++c){ acc[c] = ~(u32x4)(0u); unit[c] = min(c*64 + lane, row16 - 1); }
	for(uint32_t i = k0; i < k1; i += U){
		const u32x4 *p[U];
#pragma unroll
		for(int u = 0; u < U; ++u){ p[u] = src + row_of(q, min(i + u, k1 - 1), nrows)*row16; }
#pragma unroll
		for(int c = 0; c < CH; ++c){
			u32x4 a[U];
#pragma unroll
			for(int u = 0; u < U; ++u){ a[u] = __builtin_nontemporal_load(p[u] + unit[c]); }
#pragma unroll
			for(int u = 0; u < U; ++u){ acc[c] &= a[u]; }
		}
	}
	u32x4 t = acc[0];
	for(int c = 1; c < CH; ++c){ t ^= acc[c]; }
	if((t.x ^ t.y ^ t.z ^ t.w) == 0x12345678u){ *sink = 1; }
}

template <typename F>
static double timeit(F launch, double bytes, int iters = 3)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	launch();
	hipEventRecord(a, 0);
	for(int i = 0; i < iters; ++i){ launch(); }
	hipEventRecord(b, 0); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b);
	return bytes*iters/(ms*1e-3)/1e9;
}

int main(int argc, char **argv)
{
	const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 32ull) << 30;
	u32x4 *d; uint32_t *sink;
	CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
	CK(hipMemset(d, 0x5A, bytes));
	const uint64_t n16 = bytes/16;
#define GS(U, NT, G) printf("grid_stride U=%d nt=%d grid=%d: %.0f GB/s\n", U, NT, G, timeit([&]{ hipLaunchKernelGGL((grid_stride<U, NT>), dim3(G), dim3(256), 0, 0, d, n16, sink); }, (double)bytes)); fflush(stdout)
	GS(8, true, 2048); GS(8, false, 2048); GS(4, true, 2048); GS(16, true, 2048); GS(8, true, 4096); GS(8, true, 8192); GS(8, true, 1024); GS(16, true, 1024); GS(4, true, 8192); GS(2, true, 16384);
#define WC(U, NT, G) printf("wave_contig U=%d nt=%d grid=%d: %.0f GB/s\n", U, NT, G, timeit([&]{ hipLaunchKernelGGL((wave_contig<U, NT>), dim3(G), dim3(256), 0, 0, d, n16, sink); }, (double)(n16/(G*4ull)/(64*U)*(64*U)*G*4*16))); fflush(stdout)
	WC(8, true, 2048); WC(8, true, 4096); WC(16, true, 2048); WC(4, true, 4096); WC(8, false, 2048);
	for(uint32_t row_bytes : {1024u, 2048u, 12544u, 125056u}){
		const uint32_t row16 = row_bytes/16;
		const uint64_t nrows = bytes/row_bytes;
		const uint64_t waves = 8192, total_rows = (8ull << 30)/row_bytes, rpw = total_rows/waves/8*8;
		printf("random rows of %u B, 8 in flight: %.0f GB/s\n", row_bytes, timeit([&]{ hipLaunchKernelGGL((random_rows<8>), dim3(waves/4), dim3(256), 0, 0, d, nrows, row16, rpw, sink); }, (double)rpw*waves*row_bytes));
		fflush(stdout);
	}
	{	// C2 shape: 1000 queries x 970 rows of 12544 B out of the whole buffer
		const uint32_t row_bytes = argc > 2 ? (uint32_t)atoi(argv[2]) : 12544;
		const uint32_t row16 = row_bytes/16, nq = 1000, nk = 968;
		const uint64_t nrows = bytes/row_bytes;
		const double by = (double)nq*nk*row_bytes;
		printf("rows of %u bytes\n", row_bytes);
		for(int rep = 0; rep < 3; ++rep){
#define TR(VEC, U) { const uint32_t tiles = (row16 + 64*VEC - 1)/(64*VEC); printf("tiled_rows VEC=%d U=%d (%u waves): %.0f GB/s\n", VEC, U, nq*tiles, timeit([&]{ hipLaunchKernelGGL((tiled_rows<VEC, U>), dim3((nq*tiles + 3)/4), dim3(256), 0, 0, d, nrows, row16, nq, nk, sink); }, by, 5)); fflush(stdout); }
			TR(2, 8); TR(1, 8); TR(4, 4); TR(4, 8);
#define BR(VMAX, U) { const uint32_t chunks = (row16 + 63)/64, wpq = (chunks + VMAX - 1)/VMAX; printf("balanced_rows VMAX=%d U=%d (%u waves): %.0f GB/s\n", VMAX, U, nq*wpq, timeit([&]{ hipLaunchKernelGGL((balanced_rows<VMAX, U>), dim3((nq*wpq + 3)/4), dim3(256), 0, 0, d, nrows, row16, nq, nk, sink); }, by, 5)); fflush(stdout); }
			BR(2, 8); BR(3, 4); BR(3, 8); BR(4, 4); BR(4, 8);
#define WR(U, SEGS) printf("walk_rows CH=13 U=%d segs=%d (%u waves): %.0f GB/s\n", U, SEGS, nq*SEGS, timeit([&]{ hipLaunchKernelGGL((walk_rows<13, U>), dim3((nq*SEGS + 3)/4), dim3(256), 0, 0, d, nrows, row16, nq, nk, SEGS, sink); }, by, 5)); fflush(stdout)
			WR(8, 1); WR(8, 2); WR(8, 4); WR(8, 8); WR(4, 8); WR(4, 4); WR(2, 8); WR(16, 2);
		}
	}
	return 0;
}
