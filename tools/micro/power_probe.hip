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

// Steps towards and_walk_kernel, to see what costs it the 5-7 % it runs below the bare pattern:
//   MODE 1  row numbers come from an array in memory (scalar loads per group of R rows, as in the kernel)
//   MODE 2  + rows read through buffer descriptors (bounds-checked raw buffer loads), CH KiB-steps fully unrolled, one
//            KiB-step at a time (sched_barrier), accumulators per step -- the kernel's inner loop
//   MODE 3  MODE 2 with the NEXT group's row numbers loaded as a VECTOR load one group ahead and broadcast with
//            readlane (no scalar-load latency in the loop)
// the matrix's content: Bernoulli(1/4) bits like the benchmark databases (AND of two random words), or left constant
__global__ void fill_random(u32x4 *buf, uint64_t n16)
{
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x*blockDim.x){
		uint64_t x = i*0x9E3779B97F4A7C15ull + 99;
		u32x4 v;
		for(int d = 0; d < 4; ++d){
			x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
			const uint32_t a = (uint32_t)x, b = (uint32_t)(x >> 32);
			v[d] = a & b;
		}
		buf[i] = v;
	}
}

__global__ void fill_rows(uint32_t *rows, uint64_t n, uint64_t nrows)
{
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x*blockDim.x){
		uint64_t x = i*0x9E3779B97F4A7C15ull + 777;
		x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
		rows[i] = (uint32_t)(x % nrows);
	}
}

template <int R, int CH, int MODE>
__global__ __launch_bounds__(512) void walk_like(const uint8_t *db, uint64_t stride, uint32_t row_bytes, const uint32_t *__restrict__ rows, uint64_t rows_per_wave, uint32_t *sink)
{
	extern __shared__ uint32_t pad[];
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t wave = __builtin_amdgcn_readfirstlane((uint32_t)(blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6)));
	const uint32_t *rq = rows + wave*rows_per_wave;
	u32x4 acc[CH];
#pragma unroll
	for(int j = 0; j < CH; ++j){ acc[j] = ~(u32x4)(0u); }
	uint32_t vnext = (MODE == 3) ? rq[lane & (R - 1)] : 0;            // lane u (mod R) holds row u of the next group
	for(uint64_t r = 0; r < rows_per_wave; r += R){
		uint32_t idx[R];
		if(MODE == 3){
#pragma unroll
			for(int u = 0; u < R; ++u){ idx[u] = __builtin_amdgcn_readlane(vnext, u); }
			vnext = rq[min(r + R, rows_per_wave - R) + (lane & (R - 1))];     // requested now, used an iteration later
		}
		else{
#pragma unroll
			for(int u = 0; u < R; ++u){ idx[u] = rq[r + u]; }
		}
		if(MODE == 1){
			const u32x4 *p[R];
#pragma unroll
			for(int u = 0; u < R; ++u){ p[u] = reinterpret_cast<const u32x4*>(db + (uint64_t)idx[u]*stride) + lane; }
#pragma unroll
			for(int j = 0; j < CH; ++j){
				u32x4 x[R];
#pragma unroll
				for(int u = 0; u < R; ++u){ x[u] = __builtin_nontemporal_load(p[u] + j*64); }
#pragma unroll
				for(int u = 0; u < R; ++u){ acc[j] &= x[u]; }
			}
		}
		else{
			__amdgpu_buffer_rsrc_t rs[R];
#pragma unroll
			for(int u = 0; u < R; ++u){ rs[u] = __builtin_amdgcn_make_buffer_rsrc((void*)(db + (uint64_t)idx[u]*stride), 0, row_bytes, 0x00020000); }
#pragma unroll
			for(int j = 0; j < CH; ++j){
				u32x4 x[R];
#pragma unroll
				for(int u = 0; u < R; ++u){ x[u] = __builtin_amdgcn_raw_buffer_load_b128(rs[u], lane*16u, j*1024, 2); }
#pragma unroll
				for(int u = 0; u < R; ++u){ acc[j] &= x[u]; }
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
	u32x4 t = acc[0];
#pragma unroll
	for(int j = 1; j < CH; ++j){ t ^= acc[j]; }
	if((t.x ^ t.y ^ t.z ^ t.w) == 0x12345678u){ *sink = pad[0]; }
}


// ---- LDS-DMA forms (round 4): do rows landing in LDS (global_load_lds_dwordx4: no VGPR destination, so the bytes in flight
// per wave are not bounded by the register budget) gather faster than rows landing in registers?
//   walk_lds<R, P, CH>   and_walk_kernel's pattern (MODE 2 above) with every KiB-step of R rows DMA'd into a per-wave LDS
//                        ring P steps ahead of the step being ANDed: P*R KiB in flight per wave, read back with ds_read_b128
//   narrow_reg<U> / narrow_lds<P>   the narrow shape (256-byte rows = one 2048-column file, four lane groups of 16 per wave,
//                        every group its own row list): U rows per group in registers vs P KiB-instructions ahead in LDS
// Every form writes a per-wave checksum of what it read (XOR-accumulated: the AND of fifty Bernoulli(1/4) rows is all zero and would
// say nothing), so that the host can tell that the LDS forms read the same bytes.
__device__ __forceinline__ void glds16_nt(const void *gsrc, uint32_t lds_dst)
{
	// M0 = wave-uniform LDS byte address; every lane's 16 bytes land at M0 + lane*16 (cdna_hip_programming.md section 7: the
	// LDS-DMA recipe).  The lgkmcnt(0) makes sure this wave's earlier ds_reads of the slot have returned before it is refilled.
	unsigned keep;
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int R, int P, int CH>
__global__ __launch_bounds__(512) void walk_lds(const uint8_t *db, uint64_t stride, const uint32_t *__restrict__ rows, uint64_t rows_per_wave, uint32_t *check)
{
	extern __shared__ __attribute__((aligned(1024))) uint8_t ring[];       // per wave (P + 1) slots of R KiB
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint64_t wave = __builtin_amdgcn_readfirstlane((uint32_t)(blockIdx.x*(blockDim.x/64) + w));
	const uint32_t *rq = rows + wave*rows_per_wave;
	constexpr uint32_t SLOT = R*1024, NSLOT = P + 1;
	uint8_t *mine = ring + (size_t)w*NSLOT*SLOT;
	const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)mine);
	const uint64_t groups = rows_per_wave/R, steps = groups*CH;
	u32x4 acc[CH];
#pragma unroll
	for(int j = 0; j < CH; ++j){ acc[j] = ~(u32x4)(0u); }
	// issue side: step ti = (group gi, KiB-step ji) into slot si
	uint64_t gi = 0; uint32_t ji = 0, si = 0;
	auto issue = [&]() {
#pragma unroll
		for(int u = 0; u < R; ++u){
			const uint32_t r = rq[gi*R + u];
			glds16_nt(db + (uint64_t)r*stride + ji*1024u + lane*16u, lds0 + si*SLOT + u*1024u);
		}
		if(++ji == CH){ ji = 0; ++gi; }
		if(++si == NSLOT){ si = 0; }
	};
	uint64_t issued = 0;
	for(; issued < (uint64_t)P && issued < steps; ++issued){ issue(); }
	uint32_t sr = 0;
	for(uint64_t g = 0; g < groups; ++g){
#pragma unroll
		for(int j = 0; j < CH; ++j){
			if(issued < steps){ issue(); ++issued; wait_vm<P*R>(); }       // P steps stay in flight behind the one read now
			else{ wait_vm<0>(); }
			const u32x4 *sl = reinterpret_cast<const u32x4*>(mine + sr*SLOT) + lane;
#pragma unroll
			for(int u = 0; u < R; ++u){ acc[j] ^= sl[u*64]; }
			if(++sr == NSLOT){ sr = 0; }
		}
	}
	u32x4 t = acc[0];
#pragma unroll
	for(int j = 1; j < CH; ++j){ t ^= acc[j]; }
	uint32_t c = t.x ^ t.y ^ t.z ^ t.w;
	for(int d = 32; d; d >>= 1){ c ^= __shfl_xor(c, d); }
	if(lane == 0){ check[wave] = c; }
}

__device__ __forceinline__ uint32_t narrow_row(uint64_t &x, uint64_t nrows)
{
	x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
	return (uint32_t)(((x*0x2545F4914F6CDD1Dull) >> 33) % nrows);
}

// one wave = four queries of `n` rows each (lane group of 16 = one 256-byte row per load)
template <int U>
__global__ __launch_bounds__(256) void narrow_reg(const uint8_t *db, uint64_t nrows, uint32_t n, uint32_t *check)
{
	const uint32_t lane = threadIdx.x & 63, l = lane & 15;
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6);
	uint64_t x = (wave*4 + lane/16)*0x9E3779B97F4A7C15ull + 31337;
	u32x4 acc = ~(u32x4)(0u);
	for(uint32_t i = 0; i < n; i += U){
		u32x4 v[U];
#pragma unroll
		for(int u = 0; u < U; ++u){ v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(db + (uint64_t)narrow_row(x, nrows)*256) + l); }
#pragma unroll
		for(int u = 0; u < U; ++u){ acc ^= v[u]; }
	}
	uint32_t c = acc.x ^ acc.y ^ acc.z ^ acc.w;
	for(int d = 32; d; d >>= 1){ c ^= __shfl_xor(c, d); }
	if(lane == 0){ check[wave] = c; }
}

template <int P>
__global__ __launch_bounds__(256) void narrow_lds(const uint8_t *db, uint64_t nrows, uint32_t n, uint32_t *check)
{
	extern __shared__ __attribute__((aligned(1024))) uint8_t ring[];       // per wave (P + 1) slots of 1 KiB
	const uint32_t lane = threadIdx.x & 63, l = lane & 15, w = threadIdx.x >> 6;
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + w;
	uint64_t x = (wave*4 + lane/16)*0x9E3779B97F4A7C15ull + 31337;
	constexpr uint32_t NSLOT = P + 1;
	uint8_t *mine = ring + (size_t)w*NSLOT*1024;
	const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)mine);
	u32x4 acc = ~(u32x4)(0u);
	uint32_t si = 0, sr = 0, issued = 0;
	auto issue = [&]() {
		glds16_nt(db + (uint64_t)narrow_row(x, nrows)*256 + l*16u, lds0 + si*1024u);
		if(++si == NSLOT){ si = 0; }
	};
	for(; issued < (uint32_t)P && issued < n; ++issued){ issue(); }
	for(uint32_t i = 0; i < n; ++i){
		if(issued < n){ issue(); ++issued; wait_vm<P>(); }
		else{ wait_vm<0>(); }
		acc ^= reinterpret_cast<const u32x4*>(mine + sr*1024u)[lane];
		if(++sr == NSLOT){ sr = 0; }
	}
	uint32_t c = acc.x ^ acc.y ^ acc.z ^ acc.w;
	for(int d = 32; d; d >>= 1){ c ^= __shfl_xor(c, d); }
	if(lane == 0){ check[wave] = c; }
}

// Wide rows (C3: 125 000-byte rows).  PIECES: every wave reads random 8 KiB pieces (row, piece) on its own, R in flight --
// what the tiled kernel and the column-tiled walk do to the memory: a row's pieces are fetched by different waves at
// different times.  COOP: the 16 waves of a workgroup take the SAME row at the same time, wave w its w-th 8 KiB -- the row
// arrives as one 122 KiB burst.
template <int R, bool COOP>
__global__ __launch_bounds__(1024) void wide_rows(const u32x4 *src, uint64_t nrows, uint64_t stride16, uint32_t row_units, uint64_t iters, uint32_t *sink)
{
	extern __shared__ uint32_t pad[];
	u32x4 acc = (u32x4)(0u);
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint64_t wave = (uint64_t)blockIdx.x*(blockDim.x/64) + w;
	uint64_t x = (COOP ? (uint64_t)blockIdx.x : wave)*0x9E3779B97F4A7C15ull + 4242;
	for(uint64_t it = 0; it < iters; it += R){
		const u32x4 *p[R];
		uint32_t units[R];
#pragma unroll
		for(int u = 0; u < R; ++u){
			x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
			const uint64_t h = x*0x2545F4914F6CDD1Dull;
			const uint64_t row = __builtin_amdgcn_readfirstlane((uint32_t)(h >> 33)) % nrows;
			const uint32_t piece = COOP ? w : __builtin_amdgcn_readfirstlane((uint32_t)(h & 0xFFFF)) % 16u;
			p[u] = src + row*stride16 + (uint64_t)piece*512 + lane;
			units[u] = (piece*512 < row_units) ? min(512u, row_units - piece*512) : 0u;      // 16-byte units of this piece inside the row
		}
#pragma unroll
		for(int j = 0; j < 8; ++j){
			u32x4 a[R];
#pragma unroll
			for(int u = 0; u < R; ++u){ a[u] = __builtin_nontemporal_load(p[u] + min(j*64u, units[u] - 1u - min(lane, units[u] - 1u))); }      // past the row end: clamped onto its last units (as the tiled kernel does)
#pragma unroll
			for(int u = 0; u < R; ++u){ acc ^= a[u]; }
		}
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u){ *sink = pad[0]; }
}

template <int R, bool COOP>
int run_wide(const char *name, int wgs, int threads, size_t lds, const u32x4 *buf, uint64_t bytes, uint32_t *sink, double seconds)
{
	CK(hipFuncSetAttribute((const void*)wide_rows<R, COOP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	const uint64_t stride = 125056, nrows = bytes/stride;
	const uint32_t row_units = 125000/16 + 1;
	const uint64_t waves = (uint64_t)wgs*(threads/64);
	const uint64_t iters = (1500000ull*16/waves)/R*R + R;             // ~1.5 M rows x 16 pieces per launch
	// bytes touched per launch: every (row, piece) reads the part of its 8 KiB inside the row; on average a row's 16 pieces hold 125 008 bytes
	const double bytes_per_launch = (double)waves*iters*(125008.0/16);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	printf("%-60s", name);
	const auto t0 = std::chrono::steady_clock::now();
	double last_report = 0;
	while(true){
		CK(hipEventRecord(e0, 0));
		for(int r = 0; r < 2; ++r){ hipLaunchKernelGGL((wide_rows<R, COOP>), dim3(wgs), dim3(threads), lds, 0, buf, nrows, stride/16, row_units, iters, sink); }
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if(el - last_report >= 1.0){ printf(" %5.0f", 2.0*bytes_per_launch/ms/1e6); fflush(stdout); last_report = el; }
		if(el >= seconds){ break; }
	}
	printf("  GB/s touched\n");
	return 0;
}

template <int MODE>
int run_walk_like(const char *name, const uint8_t *buf, uint64_t bytes, uint32_t *rows, uint32_t *sink, double seconds)
{
	const uint64_t stride = 12544, nrows = bytes/stride, total_rows = 970000;
	const int wgs = 256, threads = 512;
	const uint64_t waves = (uint64_t)wgs*(threads/64);
	const uint64_t rpw = (total_rows + waves - 1)/waves/4*4 + 4;
	CK(hipFuncSetAttribute((const void*)walk_like<4, 13, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100*1024));
	hipLaunchKernelGGL(fill_rows, dim3(1024), dim3(256), 0, 0, rows, waves*rpw, nrows);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	printf("%-60s", name);
	const auto t0 = std::chrono::steady_clock::now();
	double last_report = 0;
	while(true){
		CK(hipEventRecord(e0, 0));
		for(int r = 0; r < 8; ++r){ hipLaunchKernelGGL((walk_like<4, 13, MODE>), dim3(wgs), dim3(threads), 100*1024, 0, buf, stride, (uint32_t)stride, rows, rpw, sink); }
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if(el - last_report >= 1.0){ printf(" %5.0f", 8.0*waves*rpw*12544/ms/1e6); fflush(stdout); last_report = el; }
		if(el >= seconds){ break; }
	}
	printf("  GB/s (12 544 B touched per row; %.4f ms per 970 k rows)\n", 0.0);
	return 0;
}

template <int R>
int run_gather(const char *name, int wgs, int threads, size_t lds, const u32x4 *buf, uint64_t bytes, uint32_t *sink, double seconds, uint32_t gap = 0,
               uint64_t stride_bytes = 12544, uint32_t row_kib = 13)                       // C2: 12 500-byte rows, 12 544-byte stride
{
	CK(hipFuncSetAttribute((const void*)gather<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	const uint64_t stride16 = stride_bytes/16, nrows = bytes/stride_bytes;
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


// run `launch` for `seconds`, print GB/s second by second; -> checksum over the per-wave checks of the last launch
template <typename F>
int timed(const char *name, double seconds, double bytes_per_launch, int reps, uint32_t *check, uint64_t nwaves, uint32_t *sum_out, F launch)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	printf("%-72s", name);
	const auto t0 = std::chrono::steady_clock::now();
	double last_report = 0, best = 0;
	while(true){
		CK(hipEventRecord(e0, 0));
		for(int r = 0; r < reps; ++r){ launch(); }
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		CK(hipGetLastError());
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		const double gbps = reps*bytes_per_launch/ms/1e6;
		best = gbps > best ? gbps : best;
		const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if(el - last_report >= 1.0){ printf(" %5.0f", gbps); fflush(stdout); last_report = el; }
		if(el >= seconds){ break; }
	}
	uint32_t *h = (uint32_t*)malloc(nwaves*4);
	CK(hipMemcpy(h, check, nwaves*4, hipMemcpyDeviceToHost));
	uint32_t sum = 0;
	for(uint64_t i = 0; i < nwaves; ++i){ sum = sum*31u + h[i]; }
	free(h);
	*sum_out = sum;
	printf("  GB/s  (best %5.0f, checksum %08x)\n", best, sum);
	return 0;
}

// walk_like<.., 2> with a per-wave checksum (the register reference of the LDS forms)
template <int R, int CH>
__global__ __launch_bounds__(512) void walk_reg_check(const uint8_t *db, uint64_t stride, uint32_t row_bytes, const uint32_t *__restrict__ rows, uint64_t rows_per_wave, uint32_t *check)
{
	extern __shared__ uint32_t pad[];
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t wave = __builtin_amdgcn_readfirstlane((uint32_t)(blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6)));
	const uint32_t *rq = rows + wave*rows_per_wave;
	u32x4 acc[CH];
#pragma unroll
	for(int j = 0; j < CH; ++j){ acc[j] = ~(u32x4)(0u); }
	for(uint64_t r = 0; r < rows_per_wave; r += R){
		__amdgpu_buffer_rsrc_t rs[R];
#pragma unroll
		for(int u = 0; u < R; ++u){ rs[u] = __builtin_amdgcn_make_buffer_rsrc((void*)(db + (uint64_t)rq[r + u]*stride), 0, row_bytes, 0x00020000); }
#pragma unroll
		for(int j = 0; j < CH; ++j){
			u32x4 x[R];
#pragma unroll
			for(int u = 0; u < R; ++u){ x[u] = __builtin_amdgcn_raw_buffer_load_b128(rs[u], lane*16u, j*1024, 2); }
#pragma unroll
			for(int u = 0; u < R; ++u){ acc[j] ^= x[u]; }
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	u32x4 t = acc[0];
#pragma unroll
	for(int j = 1; j < CH; ++j){ t ^= acc[j]; }
	uint32_t c = t.x ^ t.y ^ t.z ^ t.w;
	for(int d = 32; d; d >>= 1){ c ^= __shfl_xor(c, d); }
	if(lane == 0){ check[wave] = c; }
}

template <int R, int P>
int run_walk_lds(const char *name, int wg_waves, const uint8_t *buf, uint64_t bytes, uint32_t *rows, uint32_t *check, double seconds, uint32_t *sum)
{
	const uint64_t stride = 13312, nrows = bytes/stride, total_rows = 970000;      // rows of exactly 13 KiB: both forms read every byte of a row
	const int wgs = 256;
	const uint64_t waves = (uint64_t)wgs*wg_waves;
	const uint64_t rpw = (total_rows + waves - 1)/waves/8*8 + 8;            // (a multiple of every R used, the same for every form of one wave count)
	const size_t lds = (size_t)wg_waves*(P + 1)*R*1024;
	CK(hipFuncSetAttribute((const void*)walk_lds<R, P, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(fill_rows, dim3(1024), dim3(256), 0, 0, rows, waves*rpw, nrows);
	return timed(name, seconds, (double)waves*rpw*13312, 8, check, waves, sum, [&]{
		hipLaunchKernelGGL((walk_lds<R, P, 13>), dim3(wgs), dim3(wg_waves*64), lds, 0, buf, stride, rows, rpw, check); });
}

int run_walk_reg(const char *name, int wg_waves, const uint8_t *buf, uint64_t bytes, uint32_t *rows, uint32_t *check, double seconds, uint32_t *sum)
{
	const uint64_t stride = 13312, nrows = bytes/stride, total_rows = 970000;      // rows of exactly 13 KiB: both forms read every byte of a row
	const int wgs = 256;
	const uint64_t waves = (uint64_t)wgs*wg_waves;
	const uint64_t rpw = (total_rows + waves - 1)/waves/8*8 + 8;
	CK(hipFuncSetAttribute((const void*)walk_reg_check<4, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, 100*1024));
	hipLaunchKernelGGL(fill_rows, dim3(1024), dim3(256), 0, 0, rows, waves*rpw, nrows);
	return timed(name, seconds, (double)waves*rpw*13312, 8, check, waves, sum, [&]{
		hipLaunchKernelGGL((walk_reg_check<4, 13>), dim3(wgs), dim3(wg_waves*64), 100*1024, 0, buf, stride, (uint32_t)stride, rows, rpw, check); });
}


// Which LDS addresses can an LDS-DMA reach?  One wave DMAs 1 KiB of a known pattern to LDS offset `off`, waits, reads it
// back with ds_read and counts the lanes whose 16 bytes differ from a plain global load of the same bytes.
__global__ __launch_bounds__(64) void lds_reach(const u32x4 *src, uint32_t off, uint32_t *bad)
{
	extern __shared__ __attribute__((aligned(1024))) uint8_t ring[];
	const uint32_t lane = threadIdx.x;
	reinterpret_cast<u32x4*>(ring + off)[lane] = (u32x4)(0xDEADBEEFu);
	__syncthreads();
	const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(ring + off));
	glds16_nt(src + lane, lds0);
	wait_vm<0>();
	const u32x4 got = reinterpret_cast<const u32x4*>(ring + off)[lane];
	const u32x4 want = src[lane];
	if(got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w){ atomicAdd(bad, 1u); }
}

int reach_section(const u32x4 *buf, uint32_t *check)
{
	printf("---- LDS-DMA reach: lanes (of 64) whose bytes did not arrive at LDS offset X\n");
	CK(hipFuncSetAttribute((const void*)lds_reach, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
	for(uint32_t kib = 0; kib < 160; kib += 8){
		CK(hipMemset(check, 0, 4));
		hipLaunchKernelGGL(lds_reach, dim3(1), dim3(64), 160*1024, 0, buf + 4096, kib*1024, check);
		uint32_t bad = 0;
		CK(hipMemcpy(&bad, check, 4, hipMemcpyDeviceToHost));
		printf(" %u KiB: %u", kib, bad);
	}
	printf("\n");
	return 0;
}

int lds_section(const uint8_t *buf, uint64_t bytes, uint32_t *rows, uint32_t *check, double seconds)
{
	uint32_t ref8 = 0, ref4 = 0, got = 0;
	printf("---- C2's gather (970 k rows of 13 KiB, row numbers from memory): rows into REGISTERS vs rows into LDS by DMA\n");
	for(int rep = 0; rep < 2; ++rep){
		if(run_walk_reg("registers: buffer loads, 4 rows x 1 KiB in flight, 8 waves/CU  [= mode 2]", 8, buf, bytes, rows, check, seconds, &ref8)) return 1;
		if(run_walk_lds<4, 3>("LDS-DMA: 4 rows x 3 KiB-steps ahead = 12 KiB in flight/wave, 8 waves/CU", 8, buf, bytes, rows, check, seconds, &got)) return 1;
		if(got != ref8){ printf("!! checksum differs from the register form\n"); }
		if(run_walk_lds<2, 7>("LDS-DMA: 2 rows x 7 steps ahead = 14 KiB in flight/wave, 8 waves/CU", 8, buf, bytes, rows, check, seconds, &got)) return 1;
		if(got != ref8){ printf("!! checksum differs from the register form\n"); }
		if(run_walk_reg("registers: 4 rows x 1 KiB in flight, 4 waves/CU", 4, buf, bytes, rows, check, seconds, &ref4)) return 1;
		if(run_walk_lds<4, 7>("LDS-DMA: 4 rows x 7 steps ahead = 28 KiB in flight/wave, 4 waves/CU", 4, buf, bytes, rows, check, seconds, &got)) return 1;
		if(got != ref4){ printf("!! checksum differs from the register form\n"); }
		if(run_walk_lds<8, 3>("LDS-DMA: 8 rows x 3 steps ahead = 24 KiB in flight/wave, 4 waves/CU", 4, buf, bytes, rows, check, seconds, &got)) return 1;
		if(got != ref4){ printf("!! checksum differs from the register form\n"); }
	}
	printf("---- the narrow shape (10 k queries x 960 rows of 256 B = one 2048-column file; 2500 waves, four queries each)\n");
	const uint64_t nrows_n = (8ull << 30)/256;       // 2^25 rows x 256 B = 8 GiB, as the `narrow` workload
	const uint64_t nw = 2500;
	const double nbytes = (double)nw*4*960*256;      // (960 rows per query: a multiple of every unroll below)
	uint32_t refn = 0;
	for(int rep = 0; rep < 2; ++rep){
		if(timed("registers: 8 rows in flight per lane group (8 KiB/wave)", seconds, nbytes, 8, check, nw, &refn, [&]{
			hipLaunchKernelGGL((narrow_reg<8>), dim3(nw/4), dim3(256), 0, 0, buf, nrows_n, 960u, check); })) return 1;
		if(timed("registers: 16 rows in flight (16 KiB/wave)  [= and_narrow_kernel<4,16>]", seconds, nbytes, 8, check, nw, &got, [&]{
			hipLaunchKernelGGL((narrow_reg<16>), dim3(nw/4), dim3(256), 0, 0, buf, nrows_n, 960u, check); })) return 1;
		if(got != refn){ printf("!! checksum differs\n"); }
		if(timed("registers: 32 rows in flight (32 KiB/wave)", seconds, nbytes, 8, check, nw, &got, [&]{
			hipLaunchKernelGGL((narrow_reg<32>), dim3(nw/4), dim3(256), 0, 0, buf, nrows_n, 960u, check); })) return 1;
		if(got != refn){ printf("!! checksum differs\n"); }
		CK(hipFuncSetAttribute((const void*)narrow_lds<15>, hipFuncAttributeMaxDynamicSharedMemorySize, 4*16*1024));
		if(timed("LDS-DMA: 15 KiB in flight per wave", seconds, nbytes, 8, check, nw, &got, [&]{
			hipLaunchKernelGGL((narrow_lds<15>), dim3(nw/4), dim3(256), 4*16*1024, 0, buf, nrows_n, 960u, check); })) return 1;
		if(got != refn){ printf("!! checksum differs from the register form\n"); }
		CK(hipFuncSetAttribute((const void*)narrow_lds<31>, hipFuncAttributeMaxDynamicSharedMemorySize, 4*32*1024));
		if(timed("LDS-DMA: 31 KiB in flight per wave (one workgroup per CU)", seconds, nbytes, 8, check, nw, &got, [&]{
			hipLaunchKernelGGL((narrow_lds<31>), dim3(nw/4), dim3(256), 4*32*1024, 0, buf, nrows_n, 960u, check); })) return 1;
		if(got != refn){ printf("!! checksum differs from the register form\n"); }
		CK(hipFuncSetAttribute((const void*)narrow_lds<62>, hipFuncAttributeMaxDynamicSharedMemorySize, 2*63*1024));
		if(timed("LDS-DMA: 62 KiB in flight per wave, workgroups of 2 waves", seconds, nbytes, 8, check, nw, &got, [&]{
			hipLaunchKernelGGL((narrow_lds<62>), dim3(nw/2), dim3(128), 2*63*1024, 0, buf, nrows_n, 960u, check); })) return 1;
		if(got != refn){ printf("!! checksum differs from the register form\n"); }
	}
	return 0;
}

int main(int argc, char **argv)
{
	const double seconds = argc > 1 ? atof(argv[1]) : 6.0;
	const uint64_t bytes = 96ull << 30;
	u32x4 *buf; uint32_t *sink, *rows;
	CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&rows, 16u << 20));
	CK(hipMemset(buf, 1, bytes));
	const uint64_t n16 = (32ull << 30)/16;
	const size_t big = 100*1024;     // more than half a CU's LDS: one workgroup per CU
	if(run<4>("stream: 2048 WGs x 256 thr, 4 KiB/wave", 2048, 256, 0, buf, n16, sink, seconds)) return 1;
	if(argc > 2 && argv[2][0] == 'l'){        // round 4: rows into LDS by DMA against rows into registers, Bernoulli(1/4) matrix
		uint32_t *check;
		CK(hipMalloc(&check, 1u << 20));
		hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, buf, bytes/16);
		CK(hipDeviceSynchronize());
		if(run_gather<4>("bare gather: random rows from a register RNG, 13 KiB/row", 256, 512, big, buf, bytes, sink, seconds)) return 1;
		if(run_walk_like<2>("2: buffer descriptors, 13 unrolled paced KiB-steps (no checksum)", (const uint8_t*)buf, bytes, rows, sink, seconds)) return 1;
		if(reach_section(buf, check)) return 1;
		return lds_section((const uint8_t*)buf, bytes, rows, check, seconds);
	}
	if(argc > 2 && argv[2][0] == 'w'){
		for(int rep = 0; rep < 2; ++rep){
			if(run_wide<4, false>("wide rows: independent waves, random 8 KiB pieces, 4 in flight (8/CU)", 256, 512, big, buf, bytes, sink, seconds)) return 1;
			if(run_wide<4, false>("wide rows: independent waves, 16 waves/CU", 256, 1024, big, buf, bytes, sink, seconds)) return 1;
			if(run_wide<2, true>("wide rows: 16 waves of a WG share the row, 2 rows in flight", 256, 1024, big, buf, bytes, sink, seconds)) return 1;
			if(run_wide<4, true>("wide rows: 16 waves of a WG share the row, 4 rows in flight", 256, 1024, big, buf, bytes, sink, seconds)) return 1;
			if(run_wide<1, true>("wide rows: 16 waves of a WG share the row, 1 row in flight", 256, 1024, big, buf, bytes, sink, seconds)) return 1;
			if(run_wide<2, true>("wide rows: shared row, 2 in flight, 192 CUs", 192, 1024, big, buf, bytes, sink, seconds)) return 1;
		}
		return 0;
	}
	for(int rep = 0; rep < 2; ++rep){
		if(rep == 1){
			printf("---- the same with Bernoulli(1/4) random bits in the matrix instead of constant bytes\n");
			hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, buf, bytes/16);
			CK(hipDeviceSynchronize());
			if(run<4>("stream: 2048 WGs x 256 thr, 4 KiB/wave", 2048, 256, 0, buf, n16, sink, seconds)) return 1;
		}
		if(run_gather<4>("bare gather: random rows from a register RNG, 13 KiB/row", 256, 512, big, buf, bytes, sink, seconds)) return 1;
		if(run_walk_like<1>("1: row numbers from memory (scalar loads), global loads", (const uint8_t*)buf, bytes, rows, sink, seconds)) return 1;
		if(run_walk_like<2>("2: + buffer descriptors, 13 unrolled paced KiB-steps", (const uint8_t*)buf, bytes, rows, sink, seconds)) return 1;
		if(run_walk_like<3>("3: as 2, row numbers by vector load one group ahead", (const uint8_t*)buf, bytes, rows, sink, seconds)) return 1;
	}
	return 0;
}
