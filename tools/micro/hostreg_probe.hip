// Can a page-cache-resident file be DMA'd to the device without the pread copy?  mmap + hipHostRegister + H2D,
// against pread into a pinned buffer + H2D.   hipcc --offload-arch=gfx950 -O2 -o hostreg_probe hostreg_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
static double now(){ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	const size_t n = (argc > 1 ? strtoull(argv[1], 0, 10) : 1024ull) << 20;
	const char *path = "/tmp/hostreg_probe.bin";
	{	// write the file (stays in the page cache)
		FILE *f = fopen(path, "wb");
		char *buf = (char*)malloc(1 << 20); memset(buf, 0x5A, 1 << 20);
		for(size_t i = 0; i < n; i += 1 << 20){ fwrite(buf, 1, 1 << 20, f); }
		fclose(f); free(buf);
	}
	void *dev; if(hipMalloc(&dev, n) != hipSuccess){ puts("hipMalloc failed"); return 1; }
	int fd = open(path, O_RDONLY);
	for(int flags_i = 0; flags_i < 5; ++flags_i){
		const int prot = (flags_i == 2) ? (PROT_READ | PROT_WRITE) : PROT_READ;
		const int mflags = (flags_i == 0) ? MAP_SHARED : MAP_PRIVATE;
		const int pop = (flags_i >= 3) ? 0 : MAP_POPULATE;          // variants 3, 4: no MAP_POPULATE (4: + MADV_HUGEPAGE/WILLNEED)
		double t0 = now();
		void *p = mmap(nullptr, n, prot, mflags | pop, fd, 0);
		if(flags_i == 4 && p != MAP_FAILED){ (void)madvise(p, n, MADV_WILLNEED); }
		double t1 = now();
		if(p == MAP_FAILED){ perror("mmap"); continue; }
		hipError_t e = hipHostRegister(p, n, (flags_i == 2) ? hipHostRegisterDefault : hipHostRegisterReadOnly);
		double t2 = now();
		printf("variant %d (%s, %s): mmap+populate %.3f s, hipHostRegister %.3f s -> %s\n", flags_i, mflags == MAP_SHARED ? "MAP_SHARED" : "MAP_PRIVATE",
		       prot & PROT_WRITE ? "RW" : "RO", t1 - t0, t2 - t1, hipGetErrorString(e));
		if(e == hipSuccess){
			for(int rep = 0; rep < 2; ++rep){
				double a = now();
				e = hipMemcpy(dev, p, n, hipMemcpyHostToDevice);
				double b = now();
				printf("   H2D from the mapping: %.1f GB/s (%s)\n", n/(b - a)/1e9, hipGetErrorString(e));
			}
			double a = now(); (void)hipHostUnregister(p); printf("   unregister %.3f s\n", now() - a);
		}
		else{ (void)hipGetLastError(); }
		munmap(p, n);
	}
	// baseline: pread into pinned + H2D (serial, one thread)
	void *pin; (void)hipHostMalloc(&pin, 64 << 20, hipHostMallocDefault);
	double a = now();
	for(size_t off = 0; off < n; off += 64 << 20){
		if(pread(fd, pin, 64 << 20, off) <= 0){ break; }
		(void)hipMemcpy((char*)dev + off, pin, 64 << 20, hipMemcpyHostToDevice);
	}
	printf("pread -> pinned -> H2D, one thread, serial: %.1f GB/s\n", n/(now() - a)/1e9);
	close(fd); unlink(path);
	return 0;
}
