// tools/micro/graph_gap_probe.hip -- what lies between two gather kernels of consecutive searches, and whether a hipGraph
// would shrink it.  Stand-ins with EXACT durations (they spin on the 100 MHz wall clock): G = a chip-filling kernel of
// 1 ms (the gather stage), K = a small kernel of 20 us (the k-mer stage), a 4 KB copy-back.  The dependency pattern is the
// engine's (engine.hip, enqueue_search_and_copy): K_j on slot stream j%2, the ONE gather stream waits for K_j, G_j, the
// slot stream waits for G_j and copies back; the host keeps two steps in flight.
//
//   a  G back to back on one stream, nothing else                      (the floor: one queue, no dependency packet)
//   b  the engine's streams and events, plain launches + hipEventRecord
//   c  the same with the events riding on G's launch (hipExtLaunchKernelGGL: what the engine does)
//   d  N steps of the same pattern captured into ONE hipGraph, launched once  (what a caller that knows all its batches
//      in advance could do; the C-ABI's callers hand over one batch at a time)
//   e  one hipGraph per step (K -> G -> copy), launched on the gather stream step after step  (K_j+1 can no longer run
//      beside G_j: whole graphs serialise on a stream)
//
// prints, per mode: wall per step - 1 ms = everything that is not the gather kernel.
//   hipcc -O2 --offload-arch=gfx950 -o graph_gap_probe graph_gap_probe.hip && ./graph_gap_probe [steps]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess){ fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while(0)

__global__ void spin_kernel(unsigned long long ticks, unsigned int *sink)
{
	const unsigned long long t0 = wall_clock64();
	unsigned int n = 0;
	while(wall_clock64() - t0 < ticks){ ++n; }          // every wave leaves after `ticks` of the constant 100 MHz clock
	if(n == 0xFFFFFFFFu){ *sink = n; }
}

static double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Rig {
	hipStream_t slot[2], gs;
	hipEvent_t kdone[2], gdone[2], cdone[2], fork, join[2];
	unsigned int *d_sink = nullptr;
	char *d_res = nullptr, *h_res[2] = {nullptr, nullptr};
	static constexpr unsigned long long G_TICKS = 100000, K_TICKS = 2000;      // 1 ms, 20 us
	static constexpr int G_WGS = 256, G_THREADS = 512, K_WGS = 64;

	Rig()
	{
		for(int i = 0; i < 2; ++i){
			CK(hipStreamCreateWithFlags(&slot[i], hipStreamNonBlocking));
			CK(hipEventCreateWithFlags(&kdone[i], hipEventDisableTiming));
			CK(hipEventCreateWithFlags(&gdone[i], hipEventDisableTiming));
			CK(hipEventCreateWithFlags(&cdone[i], hipEventDisableTiming));
			CK(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));
			CK(hipHostMalloc((void**)&h_res[i], 4096, hipHostMallocDefault));
		}
		CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
		CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
		CK(hipMalloc((void**)&d_sink, 4));
		CK(hipMalloc((void**)&d_res, 4096));
	}
	void G(hipStream_t s, hipEvent_t stop = nullptr)
	{
		if(stop){ hipExtLaunchKernelGGL(spin_kernel, dim3(G_WGS), dim3(G_THREADS), 0, s, nullptr, stop, 0u, G_TICKS, d_sink); }
		else{ hipLaunchKernelGGL(spin_kernel, dim3(G_WGS), dim3(G_THREADS), 0, s, G_TICKS, d_sink); }
	}
	void K(hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(K_WGS), dim3(256), 0, s, K_TICKS, d_sink); }
	// one step of the engine's pattern; ride: the gather stage's stop event rides on the launch
	void step(int j, bool ride)
	{
		const int i = j & 1;
		K(slot[i]);
		CK(hipEventRecord(kdone[i], slot[i]));
		CK(hipStreamWaitEvent(gs, kdone[i], 0));
		if(ride){ G(gs, gdone[i]); }
		else{ G(gs); CK(hipEventRecord(gdone[i], gs)); }
		CK(hipStreamWaitEvent(slot[i], gdone[i], 0));
		CK(hipMemcpyAsync(h_res[i], d_res, 4096, hipMemcpyDeviceToHost, slot[i]));
	}
	void sync_all()
	{
		CK(hipStreamSynchronize(gs)); CK(hipStreamSynchronize(slot[0])); CK(hipStreamSynchronize(slot[1]));
	}
};

int main(int argc, char **argv)
{
	const int N = argc > 1 ? atoi(argv[1]) : 200;
	Rig r;
	const double g_ms = (double)Rig::G_TICKS/100000.0;
	auto report = [&](const char *name, double wall_ms) {
		printf("%-72s %8.4f ms per step = gather kernel + %6.1f us\n", name, wall_ms/N, (wall_ms/N - g_ms)*1000.0);
	};
	for(int rep = 0; rep < 2; ++rep){
		// a
		for(int j = 0; j < 10; ++j){ r.G(r.gs); }
		r.sync_all();
		double t0 = now_ms();
		for(int j = 0; j < N; ++j){ r.G(r.gs); }
		r.sync_all();
		report("a  gather kernels back to back on one stream", now_ms() - t0);
		// b, c
		for(int ride = 0; ride < 2; ++ride){
			for(int j = 0; j < 10; ++j){ r.step(j, ride); }
			r.sync_all();
			t0 = now_ms();
			for(int j = 0; j < N; ++j){
				if(j >= 2){ CK(hipEventSynchronize(r.cdone[j & 1])); }          // two steps in flight, as the engine's two slots
				r.step(j, ride);
				CK(hipEventRecord(r.cdone[j & 1], r.slot[j & 1]));
			}
			r.sync_all();
			report(ride ? "c  engine pattern, stop event riding on the gather launch" : "b  engine pattern, plain launches + hipEventRecord", now_ms() - t0);
		}
		// d: N steps in one graph
		{
			hipGraph_t graph; hipGraphExec_t exec;
			CK(hipStreamBeginCapture(r.gs, hipStreamCaptureModeGlobal));
			CK(hipEventRecord(r.fork, r.gs));
			CK(hipStreamWaitEvent(r.slot[0], r.fork, 0));
			CK(hipStreamWaitEvent(r.slot[1], r.fork, 0));
			for(int j = 0; j < N; ++j){ r.step(j, false); }
			for(int i = 0; i < 2; ++i){ CK(hipEventRecord(r.join[i], r.slot[i])); CK(hipStreamWaitEvent(r.gs, r.join[i], 0)); }
			CK(hipStreamEndCapture(r.gs, &graph));
			CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
			CK(hipGraphLaunch(exec, r.gs)); CK(hipStreamSynchronize(r.gs));      // warm
			t0 = now_ms();
			CK(hipGraphLaunch(exec, r.gs)); CK(hipStreamSynchronize(r.gs));
			report("d  the same N steps captured into ONE hipGraph", now_ms() - t0);
			CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
		}
		// e: one graph per step
		{
			hipGraph_t graph; hipGraphExec_t exec;
			CK(hipStreamBeginCapture(r.gs, hipStreamCaptureModeGlobal));
			r.K(r.gs); r.G(r.gs);
			CK(hipMemcpyAsync(r.h_res[0], r.d_res, 4096, hipMemcpyDeviceToHost, r.gs));
			CK(hipStreamEndCapture(r.gs, &graph));
			CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
			for(int j = 0; j < 10; ++j){ CK(hipGraphLaunch(exec, r.gs)); }
			CK(hipStreamSynchronize(r.gs));
			t0 = now_ms();
			for(int j = 0; j < N; ++j){ CK(hipGraphLaunch(exec, r.gs)); }
			CK(hipStreamSynchronize(r.gs));
			report("e  one hipGraph per step (K -> G -> copy), step after step", now_ms() - t0);
			CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
		}
		printf("\n");
	}
	return 0;
}
