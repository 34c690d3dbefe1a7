// Loader experiment: page-cache-resident `.db` files -> strided HBM matrix, WITHOUT the staging hop and without
// hipHostRegister/hipHostUnregister (which synchronise the device: copy engine idle 2.3 ms per 256 MB file).
//   A  today's loader: mmap + hipHostRegister, hipMemcpyAsync into a staging buffer + scatter kernel, unregister
//   B  mmap + hsa_amd_memory_lock (no device sync), ONE copy kernel reads the locked mapping over PCIe and writes the
//      strided rows; unlock of file i-1 overlaps with the kernel of file i
//   C  mmap + hipHostRegister + the same copy kernel (reads through hipHostGetDevicePointer)
//   hipcc --offload-arch=gfx950 -O2 -o lock_copy_probe lock_copy_probe.hip -lhsa-runtime64
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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

// rows of `width` bytes (width % 16 == 0), contiguous in src (dword aligned only: a .db body starts at byte 44),
// to dst rows `stride` apart.  One 16-byte unit per thread, grid-stride.
template <int VEC16>
__global__ __launch_bounds__(256) void copy_rows_kernel(uint8_t *dst, uint64_t stride, const uint8_t *src, uint64_t width, uint64_t nrows)
{
	if(VEC16){
		const uint64_t upr = width/16, total = nrows*upr;
		for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x*blockDim.x){
			const uint64_t r = i/upr, u = i%upr;
			const u32x4 v = *reinterpret_cast<const u32x4_a4*>(src + r*width + 16*u);
			*reinterpret_cast<u32x4*>(dst + r*stride + 16*u) = v;
		}
	}
	else{
		const uint64_t wpr = width/4, total = nrows*wpr;
		for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x*blockDim.x){
			const uint64_t r = i/wpr, w = i%wpr;
			*reinterpret_cast<uint32_t*>(dst + r*stride + 4*w) = *reinterpret_cast<const uint32_t*>(src + r*width + 4*w);
		}
	}
}

int main(int argc, char **argv)
{
	const int nfiles = argc > 1 ? atoi(argv[1]) : 16;
	const uint64_t width = 256, nrows = 1ull << 20, body = width*nrows, fsize = 44 + body;     // one reference-format file: 2048 columns x 2^20 rows
	const uint64_t stride = (uint64_t)nfiles*width;
	std::vector<std::string> paths;
	{
		std::vector<char> buf(fsize);
		for(int f = 0; f < nfiles; ++f){
			for(uint64_t i = 0; i < fsize; i += 4096){ buf[i] = (char)(f + i/4096); }
			paths.push_back("/tmp/lock_copy_probe_" + std::to_string(f) + ".bin");
			FILE *fp = fopen(paths.back().c_str(), "wb"); fwrite(buf.data(), 1, fsize, fp); fclose(fp);
		}
	}
	uint8_t *dev, *stage[2];
	CK(hipMalloc((void**)&dev, stride*nrows));
	CK(hipMalloc((void**)&stage[0], body)); CK(hipMalloc((void**)&stage[1], body));
	hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	if(hsa_init() != HSA_STATUS_SUCCESS){ puts("hsa_init failed"); return 1; }
	const long page = sysconf(_SC_PAGESIZE);
	const size_t maplen = (fsize + page - 1)/page*page;
	const int blocks = argc > 2 ? atoi(argv[2]) : 2048;

	// HSA agents for the SDMA variants
	static hsa_agent_t gpu_agent, cpu_agent;
	static bool have_gpu = false, have_cpu = false;
	hsa_iterate_agents([](hsa_agent_t a, void*) -> hsa_status_t {
		hsa_device_type_t t;
		hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
		if(t == HSA_DEVICE_TYPE_GPU && !have_gpu){ gpu_agent = a; have_gpu = true; }
		if(t == HSA_DEVICE_TYPE_CPU && !have_cpu){ cpu_agent = a; have_cpu = true; }
		return HSA_STATUS_SUCCESS;
	}, nullptr);
	std::vector<hsa_signal_t> sigs(nfiles);
	for(auto &sg : sigs){ hsa_signal_create(1, 0, nullptr, &sg); }

	hipStream_t st2; CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
	for(int variant = 0; variant < 10; ++variant){
		// 0 = A (register + memcpy + scatter), 1 = B dword kernel, 2 = B dwordx4 kernel, 3 = C dwordx4 kernel via hipHostRegister,
		// 4 = B dwordx4 with the lock of file i+1 issued while kernel i runs and unlock deferred by two files
		// 5 = D: hsa lock + SDMA rect copy (hsa_amd_memory_async_copy_rect) straight into the strided matrix, unlock deferred by 2
		// 6 = D without MAP_POPULATE (the lock faults the pages in), 7 = B' without MAP_POPULATE
		const char *names[] = {"A register+memcpy+scatter", "B hsa lock + dword copy kernel", "B hsa lock + dwordx4 copy kernel", "C register + dwordx4 copy kernel", "B' hsa lock + dwordx4, unlock deferred by 2",
		                       "D hsa lock + SDMA rect copy, deferred by 2", "D without MAP_POPULATE", "B' without MAP_POPULATE",
		                       "F hsa lock + hipMemcpyAsync(staging) + scatter, deferred by 2", "G B' on two alternating streams"};
		// 8 = F: does HIP's copy recognise memory locked through HSA (then SDMA reads it in place, no register/unregister)?
		// 9 = G: copy kernels of consecutive files on two streams (they overlap on the PCIe link)
		const bool sdma = (variant == 5 || variant == 6);
		const int populate = (variant >= 6) ? 0 : MAP_POPULATE;
		double t_lock = 0, t_unlock = 0, t_map = 0;
		std::vector<void*> maps(nfiles, nullptr);
		std::vector<hipEvent_t> done(nfiles);
		for(auto &e : done){ CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
		CK(hipDeviceSynchronize());
		const double t0 = now();
		const int defer = (variant >= 4) ? 2 : 1;
		for(int f = 0; f < nfiles; ++f){
			const int fd = open(paths[f].c_str(), O_RDONLY);
			double a = now();
			void *p = mmap(nullptr, maplen, PROT_READ, MAP_PRIVATE | populate, fd, 0);
			close(fd);
			t_map += now() - a;
			if(p == MAP_FAILED){ perror("mmap"); return 1; }
			maps[f] = p;
			a = now();
			void *dptr = nullptr;
			if(variant == 0 || variant == 3){
				CK(hipHostRegister(p, maplen, hipHostRegisterReadOnly));
				if(variant == 3){ CK(hipHostGetDevicePointer(&dptr, p, 0)); }
			}
			else{
				if(hsa_amd_memory_lock(p, maplen, nullptr, 0, &dptr) != HSA_STATUS_SUCCESS){ puts("hsa_amd_memory_lock failed"); return 1; }
			}
			t_lock += now() - a;
			// release an older file: its copy has had this file's map + lock time to finish
			if(f >= defer){
				const int o = f - defer;
				if(sdma){ hsa_signal_wait_scacquire(sigs[o], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED); }
				else{ CK(hipEventSynchronize(done[o])); }
				a = now();
				if(variant == 0 || variant == 3){ CK(hipHostUnregister(maps[o])); } else { hsa_amd_memory_unlock(maps[o]); }
				t_unlock += now() - a;
				munmap(maps[o], maplen); maps[o] = nullptr;
			}
			uint8_t *dst = dev + (uint64_t)f*width;
			hipStream_t cs = (variant == 9 && (f & 1)) ? st2 : st;
			if(variant == 0 || variant == 8){
				CK(hipMemcpyAsync(stage[f & 1], (const char*)(variant == 8 ? dptr : p) + 44, body, hipMemcpyHostToDevice, st));
				hipLaunchKernelGGL((copy_rows_kernel<1>), dim3(blocks), dim3(256), 0, st, dst, stride, (const uint8_t*)stage[f & 1], width, nrows);
			}
			else if(sdma){
				hsa_signal_store_relaxed(sigs[f], 1);
				hsa_pitched_ptr_t d = {dst, (size_t)stride, 0}, sp = {(char*)dptr + 44, (size_t)width, 0};
				hsa_dim3_t zero = {0, 0, 0}, range = {(uint32_t)width, (uint32_t)nrows, 1};
				const hsa_status_t hs = hsa_amd_memory_async_copy_rect(&d, &zero, &sp, &zero, &range, gpu_agent, hsaHostToDevice, 0, nullptr, sigs[f]);
				if(hs != HSA_STATUS_SUCCESS){ printf("hsa_amd_memory_async_copy_rect failed: %d\n", (int)hs); return 1; }
			}
			else if(variant == 1){
				hipLaunchKernelGGL((copy_rows_kernel<0>), dim3(blocks), dim3(256), 0, st, dst, stride, (const uint8_t*)dptr + 44, width, nrows);
			}
			else{
				hipLaunchKernelGGL((copy_rows_kernel<1>), dim3(blocks), dim3(256), 0, cs, dst, stride, (const uint8_t*)dptr + 44, width, nrows);
			}
			CK(hipGetLastError());
			CK(hipEventRecord(done[f], cs));
		}
		CK(hipStreamSynchronize(st));
		CK(hipStreamSynchronize(st2));
		if(sdma){ for(int f = 0; f < nfiles; ++f){ hsa_signal_wait_scacquire(sigs[f], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED); } }
		const double t1 = now();
		for(int f = 0; f < nfiles; ++f){
			if(maps[f]){
				if(variant == 0 || variant == 3){ CK(hipHostUnregister(maps[f])); } else { hsa_amd_memory_unlock(maps[f]); }
				munmap(maps[f], maplen);
			}
		}
		// verify a few bytes
		std::vector<uint8_t> row(stride);
		CK(hipMemcpy(row.data(), dev + 16*stride, stride, hipMemcpyDeviceToHost));       // row 16 = file offset 44 + 16*256 = 4140 -> page 1, byte 44
		// file byte 8192 (pattern value f + 2) lies in row (8192 - 44)/256 = 31, column byte (8192 - 44) % 256 = 212
		CK(hipMemcpy(row.data(), dev + 31*stride, stride, hipMemcpyDeviceToHost));
		bool ok = true;
		for(int f = 0; f < nfiles; ++f){ ok = ok && row[(uint64_t)f*width + 212] == (uint8_t)(f + 2); }
		CK(hipMemset(dev, 0, stride*nrows));
		printf("%-46s %s %6.1f GB/s  (%d files x %.0f MB; mmap %.1f ms, lock %.1f ms, unlock %.1f ms per file)\n", names[variant],
		       ok ? "ok " : "BAD", (double)nfiles*body/(t1 - t0)/1e9, nfiles, body/1e6, t_map/nfiles*1e3, t_lock/nfiles*1e3, t_unlock/std::max(nfiles - defer, 1)*1e3);
		for(auto &e : done){ (void)hipEventDestroy(e); }
	}
	for(auto &p : paths){ unlink(p.c_str()); }
	return 0;
}
