// tools/micro/row_width_probe.hip -- what the memory system gives for the gather pattern as a function of the ROW WIDTH:
// the ceiling the strong-scaling shapes (a workload's columns split over 2, 4, 8 GPUs: rows of 6.3, 3.2, 1.6 KB instead of
// 12.5 KB) are measured against.  No row list, no AND, no hits: every wave takes R random rows (register RNG) at a time and
// reads them KiB-step after KiB-step (PACED, the walk kernel's schedule) or all steps at once.
//   rows of W bytes at a stride of W rounded up to 128, 2^23 rows, ~4 GB read per launch, best of 5 launches
//   hipcc -O3 --offload-arch=gfx950 -o row_width_probe row_width_probe.hip && ./row_width_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess){ fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while(0)

template <int R, int STEPS, bool PACED, int LB>
__global__ __launch_bounds__(LB) void gather_rows(const u32x4 *__restrict__ src, uint32_t nrows_mask, uint32_t units, uint64_t stride16, uint32_t groups_per_wave, uint32_t *sink)
{
	extern __shared__ uint32_t pad[];
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6);
	uint64_t x = wave*0x9E3779B97F4A7C15ull + 12345;
	u32x4 acc = (u32x4)(0u);
	for(uint32_t g = 0; g < groups_per_wave; ++g){
		__amdgpu_buffer_rsrc_t rs[R];
#pragma unroll
		for(int u = 0; u < R; ++u){
			x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
			const uint32_t row = __builtin_amdgcn_readfirstlane((uint32_t)((x*0x2545F4914F6CDD1Dull) >> 33)) & nrows_mask;
			rs[u] = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (uint64_t)row*stride16), 0, units*16u, 0x00020000);     // lanes past the row end read nothing
		}
		u32x4 a[STEPS][R];
#pragma unroll
		for(int j = 0; j < STEPS; ++j){
#pragma unroll
			for(int u = 0; u < R; ++u){ a[j][u] = __builtin_amdgcn_raw_buffer_load_b128(rs[u], lane*16u, j*1024, 2); }
			if(PACED){
#pragma unroll
				for(int u = 0; u < R; ++u){ acc ^= a[j][u]; }
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		if(!PACED){
#pragma unroll
			for(int j = 0; j < STEPS; ++j){
#pragma unroll
				for(int u = 0; u < R; ++u){ acc ^= a[j][u]; }
			}
		}
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u){ *sink = pad[0]; }
}

template <int R, int STEPS, bool PACED, int WAVES>
static void run(const char *label, const u32x4 *buf, uint32_t nrows, uint32_t row_bytes, uint64_t stride, uint32_t *sink)
{
	constexpr int waves_per_cu = WAVES;
	const int threads = waves_per_cu*64, wgs = 256;
	const size_t lds = 100*1024;                              // one workgroup per CU
	CK(hipFuncSetAttribute((const void*)gather_rows<R, STEPS, PACED, WAVES*64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	const uint32_t units = (row_bytes + 15)/16;
	const uint64_t waves = (uint64_t)wgs*waves_per_cu;
	const uint64_t rows_total = (4ull << 30)/row_bytes;
	const uint32_t groups = (uint32_t)std::max<uint64_t>(1, rows_total/waves/R);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e30f;
	for(int rep = 0; rep < 6; ++rep){
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((gather_rows<R, STEPS, PACED, WAVES*64>), dim3(wgs), dim3(threads), lds, 0, buf, nrows - 1, units, stride/16, groups, sink);
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		if(rep){ best = std::min(best, ms); }
	}
	const double gbs = (double)waves*groups*R*units*16.0/best/1e6;
	printf("  %-52s %6.0f GB/s  %.3f of 8 TB/s\n", label, gbs, gbs/8000.0);
	fflush(stdout);
	CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int STEPS>
static void width(const u32x4 *buf, uint32_t row_bytes, uint32_t *sink)
{
	const uint64_t stride = ((uint64_t)row_bytes + 127)/128*128;
	const uint32_t nrows = 1u << 23;
	printf("rows of %u bytes (stride %llu, %d KiB-steps, matrix %.1f GB)\n", row_bytes, (unsigned long long)stride, STEPS, (double)stride*nrows/1e9);
	run<4, STEPS, true, 8>("4 rows in flight, paced, 8 waves/CU", buf, nrows, row_bytes, stride, sink);
	run<8, STEPS, true, 8>("8 rows in flight, paced, 8 waves/CU", buf, nrows, row_bytes, stride, sink);
	run<4, STEPS, true, 16>("4 rows in flight, paced, 16 waves/CU", buf, nrows, row_bytes, stride, sink);
	run<8, STEPS, true, 16>("8 rows in flight, paced, 16 waves/CU", buf, nrows, row_bytes, stride, sink);
	if(STEPS <= 4){
		run<4, STEPS, false, 8>("4 rows, all steps at once, 8 waves/CU", buf, nrows, row_bytes, stride, sink);
		run<8, STEPS, false, 8>("8 rows, all steps at once, 8 waves/CU", buf, nrows, row_bytes, stride, sink);
		run<4, STEPS, false, 16>("4 rows, all steps at once, 16 waves/CU", buf, nrows, row_bytes, stride, sink);
		run<8, STEPS, false, 16>("8 rows, all steps at once, 16 waves/CU", buf, nrows, row_bytes, stride, sink);
	}
	if(STEPS <= 2){
		run<16, STEPS, false, 8>("16 rows, all steps at once, 8 waves/CU", buf, nrows, row_bytes, stride, sink);
		run<16, STEPS, false, 16>("16 rows, all steps at once, 16 waves/CU", buf, nrows, row_bytes, stride, sink);
	}
}

int main()
{
	const uint64_t bytes = (12544ull << 23) + (1 << 20);          // the widest matrix of the series
	u32x4 *buf; uint32_t *sink;
	CK(hipMalloc((void**)&buf, bytes)); CK(hipMalloc((void**)&sink, 4));
	CK(hipMemset(buf, 0x5A, bytes));
	width<13>(buf, 12500, sink);
	width<7>(buf, 6272, sink);
	width<4>(buf, 3200, sink);
	width<2>(buf, 1664, sink);
	width<1>(buf, 256, sink);
	return 0;
}
