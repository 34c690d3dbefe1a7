// Microbenchmark: the random-row gather (and_walk_kernel's access pattern at C2's shape: 2^23 rows of 12.5 KB, four rows
// in flight per wave, one workgroup of 8 waves per CU) over a 105 GB matrix that is ALLOCATED in different ways.
// tools/placement_probe.py showed the same kernel at 1.89 ... 1.99 ms depending on where the matrix happened to lie.
//   plain     hipMalloc(bytes)
//   contig    hipExtMallocWithFlags(bytes, hipDeviceMallocContiguous)
//   va2g      hipMalloc(bytes + 2 GiB), the matrix starts at the next 2 GiB boundary of the virtual address
//   vmm<N>    hipMemAddressReserve + physical memory created in N-MiB handles, mapped side by side
//   vmma<N>   the same, mapped from a 2 GiB boundary of a larger reservation (hipMemAddressReserve ignores its alignment argument)
//   +<variant>      the block stays allocated until the end (later blocks must lie elsewhere) and is measured again then
//   <variant>:<W>   all waves read from the same 1/W of the matrix at the same time (window after window)
//   hipcc --offload-arch=gfx950 -O3 -o placement_probe placement_probe.hip ;  ./placement_probe [variant ...]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){ printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } }while(0)

template <int R>
__global__ __launch_bounds__(512) void gather(const u32x4 *src, uint64_t nrows, uint32_t row_kib, uint64_t stride16, uint64_t rows_per_wave, uint32_t *sink, uint32_t nwin)
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
			// nwin > 1: all waves draw their rows from the SAME 1/nwin of the matrix at the same time (window after window)
			const uint64_t per_win = nrows/nwin, win = (r + u)*nwin/rows_per_wave;
			const uint64_t row = win*per_win + __builtin_amdgcn_readfirstlane((uint32_t)((x*0x2545F4914F6CDD1Dull) >> 33)) % per_win;
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

struct Block {
	void *base = nullptr;            // what to free
	void *use = nullptr;             // where the matrix starts
	size_t reserved = 0, mapped = 0;
	std::vector<hipMemGenericAllocationHandle_t> handles;
	bool vmm = false;
};

static int allocate(const std::string &how, size_t bytes, int device, Block *b)
{
	const size_t G2 = 2ull << 30;
	if(how == "plain"){
		CK(hipMalloc(&b->base, bytes));
		b->use = b->base;
	}
	else if(how == "contig"){
		CK(hipExtMallocWithFlags(&b->base, bytes, hipDeviceMallocContiguous));
		b->use = b->base;
	}
	else if(how == "va2g"){
		CK(hipMalloc(&b->base, bytes + G2));
		b->use = (void*)(((uintptr_t)b->base + G2 - 1)/G2*G2);
	}
	else if(how.rfind("vmm", 0) == 0){
		const bool aligned = how.rfind("vmma", 0) == 0;          // vmma<N>: the mapping starts at a 2 GiB boundary of a larger reservation
		const size_t chunk = (size_t)atoll(how.c_str() + (aligned ? 4 : 3)) << 20;
		hipMemAllocationProp prop;
		memset(&prop, 0, sizeof(prop));
		prop.type = hipMemAllocationTypePinned;
		prop.location.type = hipMemLocationTypeDevice;
		prop.location.id = device;
		size_t gran = 0;
		CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
		if(chunk == 0 || chunk % gran){ printf("chunk %zu not a multiple of the granularity %zu\n", chunk, gran); return 1; }
		const size_t total = (bytes + chunk - 1)/chunk*chunk;
		CK(hipMemAddressReserve(&b->base, total + (aligned ? G2 : 0), G2, nullptr, 0));
		b->reserved = total + (aligned ? G2 : 0);
		b->mapped = total;
		b->vmm = true;
		b->use = aligned ? (void*)(((uintptr_t)b->base + G2 - 1)/G2*G2) : b->base;
		for(size_t at = 0; at < total; at += chunk){
			hipMemGenericAllocationHandle_t h;
			CK(hipMemCreate(&h, chunk, &prop, 0));
			CK(hipMemMap((char*)b->use + at, chunk, 0, h, 0));
			b->handles.push_back(h);
		}
		hipMemAccessDesc acc;
		memset(&acc, 0, sizeof(acc));
		acc.location.type = hipMemLocationTypeDevice;
		acc.location.id = device;
		acc.flags = hipMemAccessFlagsProtReadWrite;
		CK(hipMemSetAccess(b->use, total, &acc, 1));
	}
	else{ printf("unknown variant %s\n", how.c_str()); return 1; }
	return 0;
}

static int release(Block *b)
{
	if(b->vmm){
		CK(hipMemUnmap(b->use, b->mapped));
		for(auto h : b->handles){ CK(hipMemRelease(h)); }
		CK(hipMemAddressFree(b->base, b->reserved));
	}
	else{ CK(hipFree(b->base)); }
	*b = Block();
	return 0;
}

int main(int argc, char **argv)
{
	std::vector<std::string> variants;
	for(int i = 1; i < argc; ++i){ variants.push_back(argv[i]); }
	if(variants.empty()){ variants = {"plain", "plain", "va2g", "vmm1024", "vmm2048", "vmm256", "plain"}; }
	// PROBE_GB: block size (default: C2's 105 GB = 2^23 rows)
	const uint64_t stride = 12544, row_kib = 13;
	const uint64_t nrows = getenv("PROBE_GB") ? (uint64_t)(atof(getenv("PROBE_GB"))*1e9/stride) : 1ull << 23;            // C2: 100 000 samples = 12 500 B per row, rows 128 B aligned
	const size_t bytes = nrows*stride + (2u << 20);          // (a row is READ as 13 KiB, 768 B more than the stride: slack behind the last row)
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount;
	uint32_t *sink;
	CK(hipMalloc(&sink, 4));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const uint64_t waves = (uint64_t)ncu*8, rows_per_wave = 970000/waves/4*4;   // C2's batch: 970 k rows
	std::vector<Block> kept;
	std::vector<std::string> kept_names;
	for(std::string how : variants){
		bool keep = false;
		if(!how.empty() && how[0] == '+'){ keep = true; how = how.substr(1); }          // "+contig": the block stays allocated (later ones lie elsewhere) and is measured again at the end
		uint32_t nwin = 1;
		const size_t colon = how.find(':');          // "contig:8" = eight windows
		if(colon != std::string::npos){ nwin = (uint32_t)atoi(how.c_str() + colon + 1); how = how.substr(0, colon); }
		Block b;
		if(allocate(how, bytes, 0, &b)){ return 1; }
		CK(hipMemsetAsync(b.use, 0x5a, bytes, 0));
		CK(hipDeviceSynchronize());
		float best = 1e9f, sum = 0;
		const int reps = 9;
		for(int i = 0; i < reps + 1; ++i){
			CK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL(gather<4>, dim3(ncu), dim3(512), 100*1024, 0, (const u32x4*)b.use, nrows, (uint32_t)row_kib, stride/16, rows_per_wave, sink, nwin);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if(i){ best = ms < best ? ms : best; sum += ms; }
		}
		const double touched = (double)waves*rows_per_wave*row_kib*1024;
		printf("%-8s windows %3u at %p  avg %.4f ms  best %.4f ms -> %.0f GB/s touched (avg)\n", how.c_str(), nwin, b.use, sum/reps, best, touched/(sum/reps)/1e6);
		fflush(stdout);
		if(keep){ kept.push_back(b); kept_names.push_back(how); }
		else if(release(&b)){ return 1; }
	}
	// PROBE_FOOTPRINTS="1,4,16": the kept blocks again with the rows drawn from their first F GB only
	std::vector<double> footprints;
	if(const char *e = getenv("PROBE_FOOTPRINTS")){ for(const char *c = e; *c; ){ footprints.push_back(atof(c)); c = strchr(c, ','); if(!c){ break; } ++c; } }
	for(double f : footprints){
		const uint64_t fr = (uint64_t)(f*1e9/stride) < nrows ? (uint64_t)(f*1e9/stride) : nrows;
		for(size_t k = 0; k < kept.size(); ++k){
			float sum = 0;
			for(int i = 0; i < 6; ++i){
				CK(hipEventRecord(e0, 0));
				hipLaunchKernelGGL(gather<4>, dim3(ncu), dim3(512), 100*1024, 0, (const u32x4*)kept[k].use, fr, (uint32_t)row_kib, stride/16, rows_per_wave, sink, 1u);
				CK(hipEventRecord(e1, 0));
				CK(hipEventSynchronize(e1));
				float ms = 0;
				CK(hipEventElapsedTime(&ms, e0, e1));
				if(i){ sum += ms; }
			}
			printf("footprint %6.1f GB of kept block %zu  avg %.4f ms -> %.0f GB/s\n", f, k, sum/5, (double)waves*rows_per_wave*row_kib*1024/(sum/5)/1e6);
		}
	}
	// PROBE_WINDOW_GB=8: the kept blocks again, window by window (rows drawn from [off, off + W) GB only)
	if(const char *e = getenv("PROBE_WINDOW_GB")){
		const uint64_t wr = (uint64_t)(atof(e)*1e9/stride);
		for(size_t k = 0; k < kept.size(); ++k){
			for(uint64_t r0 = 0; r0 + wr <= nrows; r0 += wr){
				float sum = 0;
				for(int i = 0; i < 4; ++i){
					CK(hipEventRecord(e0, 0));
					hipLaunchKernelGGL(gather<4>, dim3(ncu), dim3(512), 100*1024, 0, (const u32x4*)kept[k].use + r0*(stride/16), wr, (uint32_t)row_kib, stride/16, rows_per_wave, sink, 1u);
					CK(hipEventRecord(e1, 0));
					CK(hipEventSynchronize(e1));
					float ms = 0;
					CK(hipEventElapsedTime(&ms, e0, e1));
					if(i){ sum += ms; }
				}
				printf("block %zu window at %6.1f GB (+%s GB)  avg %.4f ms -> %.0f GB/s\n", k, (double)r0*stride/1e9, e, sum/3, (double)waves*rows_per_wave*row_kib*1024/(sum/3)/1e6);
			}
		}
	}
	for(int round = 0; round < 2; ++round){
		for(size_t k = 0; k < kept.size(); ++k){
			float sum = 0;
			for(int i = 0; i < 6; ++i){
				CK(hipEventRecord(e0, 0));
				hipLaunchKernelGGL(gather<4>, dim3(ncu), dim3(512), 100*1024, 0, (const u32x4*)kept[k].use, nrows, (uint32_t)row_kib, stride/16, rows_per_wave, sink, 1u);
				CK(hipEventRecord(e1, 0));
				CK(hipEventSynchronize(e1));
				float ms = 0;
				CK(hipEventElapsedTime(&ms, e0, e1));
				if(i){ sum += ms; }
			}
			printf("again: kept block %zu (%s at %p)  avg %.4f ms -> %.0f GB/s\n", k, kept_names[k].c_str(), kept[k].use, sum/5, (double)waves*rows_per_wave*row_kib*1024/(sum/5)/1e6);
		}
	}
	for(Block &b : kept){ if(release(&b)){ return 1; } }
	return 0;
}
