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
	const uint64_t bytes = 32ull << 30;
	u32x4 *buf; uint32_t *sink;
	CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
	CK(hipMemset(buf, 1, bytes));
	const uint64_t n16 = bytes/16;
	const size_t big = 100*1024;     // more than half a CU's LDS: one workgroup per CU
	if(run<4>("2048 WGs x 256 thr, 4 KiB/wave (all CUs, 32 waves)", 2048, 256, 0, buf, n16, sink, seconds)) return 1;
	if(run<4>("256 WGs x 1024 thr, 1/CU, 4 KiB/wave", 256, 1024, big, buf, n16, sink, seconds)) return 1;
	if(run<8>("192 WGs x 1024 thr, 1/CU, 8 KiB/wave", 192, 1024, big, buf, n16, sink, seconds)) return 1;
	if(run<8>("128 WGs x 1024 thr, 1/CU, 8 KiB/wave", 128, 1024, big, buf, n16, sink, seconds)) return 1;
	if(run<16>("128 WGs x 1024 thr, 1/CU, 16 KiB/wave", 128, 1024, big, buf, n16, sink, seconds)) return 1;
	if(run<16>("64 WGs x 1024 thr, 1/CU, 16 KiB/wave", 64, 1024, big, buf, n16, sink, seconds)) return 1;
	if(run<8>("256 WGs x 512 thr, 1/CU, 8 KiB/wave (8 waves/CU)", 256, 512, big, buf, n16, sink, seconds)) return 1;
	if(run<4>("2048 WGs x 256 thr again", 2048, 256, 0, buf, n16, sink, seconds)) return 1;
	return 0;
}
