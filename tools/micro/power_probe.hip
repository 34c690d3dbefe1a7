// Microbenchmark: does the SUSTAINED HBM read rate depend on how many CUs do the reading?  A cold MI355X streams
// 6.9-7.0 TB/s and settles at 6.2-6.5 TB/s after a few seconds; if that is a power / thermal
// governor, fewer busy CUs (each with more loads in flight) might hold a higher rate.  Every configuration streams a
// large buffer for several seconds and prints the rate second by second.
//   hipcc --offload-arch=gfx950 -O3 -o power_probe power_probe.hip ;  ./power_probe [seconds per config]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){ printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } }while(0)

// each wave walks its own contiguous region, U KiB per iteration; `lds_pad` bytes of dynamic LDS keep a second
// workgroup off the CU
template <int U>
__global__ __launch_bounds__(1024) void walk(const u32x4 *src, uint64_t n16, uint32_t *sink)
{
	extern __shared__ uint32_t pad[];
	u32x4 acc = (u32x4)(0u);
	const uint64_t nwaves = (uint64_t)gridDim.x*(blockDim.x/64);
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6);
	const uint64_t per = n16/nwaves/(64*U)*(64*U);
	const u32x4 *p = src + wave*per + (threadIdx.x & 63);
	for(uint64_t i = 0; i < per; i += 64*U){
		u32x4 a[U];
#pragma unroll
		for(int u = 0; u < U; ++u){ a[u] = __builtin_nontemporal_load(p + i + u*64); }
#pragma unroll
		for(int u = 0; u < U; ++u){ acc ^= a[u]; }
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u){ *sink = pad[0]; }
}

// and_walk_kernel's access pattern: a wave takes R random rows of `row_kib` KiB at a time and walks them KiB-step after
// KiB-step (R loads of 1 KiB in flight per wave)
// (gap != 0: instead of uniformly random rows, every wave takes ASCENDING rows of a region of its own, one row in `gap`
// with a jitter -- what a batch's row list sorted by address would look like: same bytes, translations and DRAM pages
// visited in order)
template <int R>
__global__ __launch_bounds__(1024) void gather(const u32x4 *src, uint64_t nrows, uint32_t row_kib, uint64_t stride16, uint64_t rows_per_wave, uint32_t *sink, uint32_t gap = 0)
{
	extern __shared__ uint32_t pad[];
	u32x4 acc = (u32x4)(0u);
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63;
	uint64_t x = wave*0x9E3779B97F4A7C15ull + 12345;
	for(uint64_t r = 0; r < rows_per_wave; r += R){
		const u32x4 *p[R];
#pragma unroll
		for(int u = 0; u < R; ++u){
			x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
			uint64_t row = __builtin_amdgcn_readfirstlane((uint32_t)((x*0x2545F4914F6CDD1Dull) >> 33));
			row = gap ? ((wave*rows_per_wave + r + u)*gap + row % gap) % nrows : row % nrows;
			p[u] = src + row*stride16 + lane;
		}
		for(uint32_t j = 0; j < row_kib; ++j){
			u32x4 a[R];
#pragma unroll
			for(int u = 0; u < R; ++u){ a[u] = __builtin_nontemporal_load(p[u] + j*64); }
#pragma unroll
			for(int u = 0; u < R; ++u){ acc ^= a[u]; }
		}
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u){ *sink = pad[0]; }
}

template <int R>
int run_gather(const char *name, int wgs, int threads, size_t lds, const u32x4 *buf, uint64_t bytes, uint32_t *sink, double seconds, uint32_t gap = 0)
{
	CK(hipFuncSetAttribute((const void*)gather<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	const uint32_t row_kib = 13;                       // C2: 12 500-byte rows, 12 544-byte stride
	const uint64_t stride16 = 12544/16, nrows = bytes/12544;
	const uint64_t total_rows = 970000;                 // one C2 step
	const uint64_t waves = (uint64_t)wgs*(threads/64);
	const uint64_t rpw = (total_rows + waves - 1)/waves/R*R + R;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	printf("%-44s", name);
	const auto t0 = std::chrono::steady_clock::now();
	double last_report = 0;
	while(true){
		CK(hipEventRecord(e0, 0));
		for(int r = 0; r < 8; ++r){ hipLaunchKernelGGL(gather<R>, dim3(wgs), dim3(threads), lds, 0, buf, nrows, row_kib, stride16, rpw, sink, gap); }
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if(el - last_report >= 1.0){ printf(" %5.0f", 8.0*waves*rpw*row_kib*1024/ms/1e6); fflush(stdout); last_report = el; }
		if(el >= seconds){ break; }
	}
	printf("  GB/s per second (bytes touched, 13 KiB per row)\n");
	return 0;
}

template <int U>
int run(const char *name, int wgs, int threads, size_t lds, const u32x4 *buf, uint64_t n16, uint32_t *sink, double seconds)
{
	CK(hipFuncSetAttribute((const void*)walk<U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	printf("%-44s", name);
	const auto t0 = std::chrono::steady_clock::now();
	double last_report = 0;
	while(true){
		CK(hipEventRecord(e0, 0));
		for(int r = 0; r < 8; ++r){ hipLaunchKernelGGL(walk<U>, dim3(wgs), dim3(threads), lds, 0, buf, n16, sink); }
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if(el - last_report >= 1.0){ printf(" %5.0f", 8.0*n16*16/ms/1e6); fflush(stdout); last_report = el; }
		if(el >= seconds){ break; }
	}
	printf("  GB/s per second\n");
	return 0;
}

int main(int argc, char **argv)
{
	const double seconds = argc > 1 ? atof(argv[1]) : 6.0;
	const bool full = argc > 2;
	const uint64_t bytes = 96ull << 30;
	u32x4 *buf; uint32_t *sink;
	CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
	CK(hipMemset(buf, 1, bytes));
	const uint64_t n16 = (32ull << 30)/16;
	const size_t big = 100*1024;     // more than half a CU's LDS: one workgroup per CU
	char name[128];
	if(run<4>("stream: 2048 WGs x 256 thr, 4 KiB/wave", 2048, 256, 0, buf, n16, sink, seconds)) return 1;
	if(full){
		for(int wpc : {8, 16}){
			for(int cus : {160, 192, 208, 224, 240, 256}){
				snprintf(name, sizeof(name), "stream: %d CUs x %d waves, 8 KiB/wave", cus, wpc);
				if(run<8>(name, cus, wpc*64, big, buf, n16, sink, seconds)) return 1;
			}
		}
		for(int wpc : {8, 16}){
			for(int cus : {160, 192, 224, 256}){
				snprintf(name, sizeof(name), "gather: %d CUs x %d waves, R=4", cus, wpc);
				if(run_gather<4>(name, cus, wpc*64, big, buf, bytes, sink, seconds)) return 1;
			}
		}
	}
	if(run_gather<4>("gather: 256 CUs x 8 waves, R=4, random rows", 256, 512, big, buf, bytes, sink, seconds)) return 1;
	// the same bytes with the rows in address order per wave (one row in 9 touched, as for one C2 batch over 2^23 rows)
	if(run_gather<4>("gather: 256 CUs x 8 waves, R=4, ASCENDING rows (1 in 9)", 256, 512, big, buf, bytes, sink, seconds, 9)) return 1;
	if(run_gather<4>("gather: 256 CUs x 8 waves, R=4, ascending, every row", 256, 512, big, buf, bytes, sink, seconds, 1)) return 1;
	if(run_gather<8>("gather: 256 CUs x 8 waves, R=8, ASCENDING rows (1 in 9)", 256, 512, big, buf, bytes, sink, seconds, 9)) return 1;
	if(run_gather<4>("gather: 256 CUs x 8 waves, R=4, random rows again", 256, 512, big, buf, bytes, sink, seconds)) return 1;
	return 0;
}
