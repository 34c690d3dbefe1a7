// Loader experiment H: page-cache-resident `.db` files -> staging buffer by the copy engine WITHOUT hipHostRegister:
// file windows are locked through HSA (no device synchronisation), hsa_amd_memory_async_copy (SDMA, linear) moves a
// file's body into one of three device staging buffers, the host waits for the copy's completion signal and launches
// the scatter kernel (HIP) that places the rows in the strided matrix; pinning / un-pinning of the neighbouring files
// overlaps with the copy that is running.
//   hipcc --offload-arch=gfx950 -O2 -o sdma_stage_probe sdma_stage_probe.hip -lhsa-runtime64
//   ./sdma_stage_probe [n_files=16] [matrix files=n_files]     (matrix files > n_files: a wider matrix, the files cycle)
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

static double now(){ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do{ hipError_t e_ = (x); if(e_ != hipSuccess){ printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } }while(0)
#define HS(x) do{ hsa_status_t s_ = (x); if(s_ != HSA_STATUS_SUCCESS){ printf("%s -> %d\n", #x, (int)s_); exit(1); } }while(0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void place_kernel(uint8_t *dst, uint64_t stride, const uint8_t *src, uint64_t width, uint64_t nrows)
{
	const uint64_t upr = width/16, total = nrows*upr;
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x*blockDim.x){
		const uint64_t r = i/upr, u = i%upr;
		*reinterpret_cast<u32x4*>(dst + r*stride + 16*u) = *reinterpret_cast<const u32x4*>(src + r*width + 16*u);
	}
}

int main(int argc, char **argv)
{
	const int nfiles = argc > 1 ? atoi(argv[1]) : 16;
	const int mfiles = argc > 2 ? atoi(argv[2]) : nfiles;            // columns of the matrix, in files
	const uint64_t width = 256, nrows = 1ull << 20, body = width*nrows, fsize = 48 + body;      // body at byte 48 here: SDMA-friendly 16-byte alignment is not needed, but keep dwords
	const uint64_t hdr = 44;
	const uint64_t stride = (uint64_t)mfiles*width;
	std::vector<std::string> paths;
	{
		std::vector<char> buf(fsize);
		for(int f = 0; f < nfiles; ++f){
			for(uint64_t i = 0; i < fsize; i += 4096){ buf[i] = (char)(f + i/4096); }
			paths.push_back("/tmp/sdma_stage_probe_" + std::to_string(f) + ".bin");
			FILE *fp = fopen(paths.back().c_str(), "wb"); fwrite(buf.data(), 1, fsize, fp); fclose(fp);
		}
	}
	uint8_t *dev, *stage[3];
	CK(hipMalloc((void**)&dev, stride*nrows));
	for(auto &s : stage){ CK(hipMalloc((void**)&s, body)); }
	hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	HS(hsa_init());
	static hsa_agent_t gpu_agent, cpu_agent;
	static bool have_gpu = false, have_cpu = false;
	hsa_iterate_agents([](hsa_agent_t a, void*) -> hsa_status_t {
		hsa_device_type_t t;
		hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
		if(t == HSA_DEVICE_TYPE_GPU && !have_gpu){ gpu_agent = a; have_gpu = true; }
		if(t == HSA_DEVICE_TYPE_CPU && !have_cpu){ cpu_agent = a; have_cpu = true; }
		return HSA_STATUS_SUCCESS;
	}, nullptr);
	const long page = sysconf(_SC_PAGESIZE);
	const size_t maplen = (fsize + page - 1)/page*page;
	const int total = mfiles;       // files loaded (cycling through the nfiles on disk)

	for(int rep = 0; rep < 3; ++rep){
		std::vector<hsa_signal_t> sig(total);
		std::vector<hipEvent_t> placed(total);
		std::vector<void*> maps(total, nullptr);
		for(auto &s : sig){ HS(hsa_signal_create(1, 0, nullptr, &s)); }
		for(auto &e : placed){ CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
		double t_lock = 0, t_wait = 0;
		CK(hipDeviceSynchronize());
		const double t0 = now();
		auto finish = [&](int o) {      // copy o is done: scatter it, release its mapping
			double a = now();
			hsa_signal_wait_scacquire(sig[o], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
			t_wait += now() - a;
			hipLaunchKernelGGL(place_kernel, dim3(2048), dim3(256), 0, st, dev + (uint64_t)o*width, stride, (const uint8_t*)stage[o % 3], width, nrows);
			CK(hipGetLastError());
			CK(hipEventRecord(placed[o], st));
			hsa_amd_memory_unlock(maps[o]);
			munmap(maps[o], maplen); maps[o] = nullptr;
		};
		for(int f = 0; f < total; ++f){
			const int fd = open(paths[f % nfiles].c_str(), O_RDONLY);
			double a = now();
			void *p = mmap(nullptr, maplen, PROT_READ, MAP_PRIVATE, fd, 0);
			close(fd);
			if(p == MAP_FAILED){ perror("mmap"); return 1; }
			void *dptr = nullptr;
			HS(hsa_amd_memory_lock(p, maplen, nullptr, 0, &dptr));
			t_lock += now() - a;
			maps[f] = p;
			if(f >= 3){ CK(hipEventSynchronize(placed[f - 3])); }          // stage[f % 3] has been scattered
			HS(hsa_amd_memory_async_copy(stage[f % 3], gpu_agent, (const char*)dptr + hdr, cpu_agent, body, 0, nullptr, sig[f]));
			if(f >= 1){ finish(f - 1); }
		}
		finish(total - 1);
		CK(hipStreamSynchronize(st));
		const double t1 = now();
		std::vector<uint8_t> row(stride);
		CK(hipMemcpy(row.data(), dev + 31*stride, stride, hipMemcpyDeviceToHost));      // file byte 8192 (value f + 2) = row 31, byte (8192 - 44) % 256 = 212
		bool ok = true;
		for(int f = 0; f < total; ++f){ ok = ok && row[(uint64_t)f*width + 212] == (uint8_t)((f % nfiles) + 2); }
		CK(hipMemset(dev, 0, stride*nrows));
		printf("H hsa lock + SDMA linear copy to staging + scatter kernel: %s %6.1f GB/s  (%d files x %.0f MB into a %.1f GB matrix; map+lock %.1f ms, signal wait %.1f ms per file)\n",
		       ok ? "ok " : "BAD", (double)total*body/(t1 - t0)/1e9, total, body/1e6, stride*nrows/1e9, t_lock/total*1e3, t_wait/total*1e3);
		for(auto &s : sig){ hsa_signal_destroy(s); }
		for(auto &e : placed){ (void)hipEventDestroy(e); }
	}
	for(auto &p : paths){ unlink(p.c_str()); }
	return 0;
}
