// kwage_amd/csrc/loader.hip -- database groups of the C ABI declared in include/kwage_amd.h: the HBM-resident bit matrix
// of all same-parameter columns, how it is allocated, and how columns get into it -- host rows, reference-format .db
// files (raw through the copy-engine pipeline over HSA-locked file windows, compressed through the staged path), sparse
// groups (only the addressed rows of every file), synthetic columns generated on the device.
// Replaces the reference's per-query ifstream seeks into the slice files (kwage.cpp:400-433) with ONE load.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>                 // types only: every HSA function is looked up at run time (see HsaApi)
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "host.hpp"
#include "engine_state.hpp"
#include "loader_kernels.hpp"

using namespace kwage;

namespace {

// hsa_amd_memory_lock / _unlock of the HSA runtime the HIP runtime of this process sits on, looked up at run time
// (no link dependency: the Python binding runs on PyTorch's bundled ROCm, the CLI on the system's).  Unlike
// hipHostRegister / hipHostUnregister they do not synchronise the device, so pinning the next file and
// un-pinning the previous one overlap with the copy that is running.
struct HsaLock {
	typedef int (*lock_fn)(void *host_ptr, size_t size, void *agents, int num_agent, void **agent_ptr);
	typedef int (*unlock_fn)(void *host_ptr);
	lock_fn lock = nullptr;
	unlock_fn unlock = nullptr;
	HsaLock()
	{
		lock = (lock_fn)dlsym(RTLD_DEFAULT, "hsa_amd_memory_lock");
		unlock = (unlock_fn)dlsym(RTLD_DEFAULT, "hsa_amd_memory_unlock");
		if(!lock || !unlock){ lock = nullptr; unlock = nullptr; }
	}
};

const HsaLock &hsa_lock()
{
	static const HsaLock h;
	return h;
}

// The rest of the HSA runtime the loader's copy-engine pipeline needs (same run-time lookup).
struct HsaApi {
	decltype(&hsa_amd_memory_async_copy) async_copy = nullptr;
	decltype(&hsa_signal_create) signal_create = nullptr;
	decltype(&hsa_signal_destroy) signal_destroy = nullptr;
	decltype(&hsa_signal_store_relaxed) signal_store = nullptr;
	decltype(&hsa_signal_wait_scacquire) signal_wait = nullptr;
	decltype(&hsa_iterate_agents) iterate_agents = nullptr;
	decltype(&hsa_agent_get_info) agent_get_info = nullptr;
	decltype(&hsa_amd_pointer_info) pointer_info = nullptr;
	bool ok = false;
	HsaApi()
	{
#define KWAGE_HSA_SYM(member, name) member = (decltype(member))dlsym(RTLD_DEFAULT, name)
		KWAGE_HSA_SYM(async_copy, "hsa_amd_memory_async_copy");
		KWAGE_HSA_SYM(signal_create, "hsa_signal_create");
		KWAGE_HSA_SYM(signal_destroy, "hsa_signal_destroy");
		KWAGE_HSA_SYM(signal_store, "hsa_signal_store_relaxed");
		KWAGE_HSA_SYM(signal_wait, "hsa_signal_wait_scacquire");
		KWAGE_HSA_SYM(iterate_agents, "hsa_iterate_agents");
		KWAGE_HSA_SYM(agent_get_info, "hsa_agent_get_info");
		KWAGE_HSA_SYM(pointer_info, "hsa_amd_pointer_info");
#undef KWAGE_HSA_SYM
		ok = async_copy && signal_create && signal_destroy && signal_store && signal_wait && iterate_agents && agent_get_info && pointer_info;
	}
};

const HsaApi &hsa_api()
{
	static const HsaApi h;
	return h;
}

bool load_env_flag(const char *name, bool fallback)
{
	const char *e = getenv(name);
	return e ? atoi(e) != 0 : fallback;
}

uint32_t grid_for(uint64_t work_items, uint32_t block, uint32_t cap_blocks = 256*8)
{
	const uint64_t b = (work_items + block - 1)/block;
	return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(b, cap_blocks));
}

// Release locked file windows, oldest first, until at most `keep` remain (each after its copy kernel has finished).
void release_locked(kwage_ctx *ctx, size_t keep)
{
	while(ctx->locked.size() > keep){
		kwage_ctx::LockedWindow w = ctx->locked.front();
		ctx->locked.pop_front();
		if(w.done){ (void)hipEventSynchronize(w.done); }
		else{ (void)hipStreamSynchronize(ctx->stream); }
		(void)hsa_lock().unlock(w.base);
		(void)munmap(w.base, w.len);
		if(w.done && w.owns_event){ ctx->spare_events.push_back(w.done); }
	}
}

}  // namespace

namespace kwage {

// Wait for the copies that read the pending file mapping, then unpin and unmap it.
void release_mapping(kwage_ctx *ctx)
{
	release_locked(ctx, 0);
	if(!ctx->map_base){ return; }
	if(ctx->map_done){ (void)hipEventSynchronize(ctx->map_done); }
	else{ (void)hipStreamSynchronize(ctx->stream); }
	(void)hipHostUnregister(ctx->map_base);
	(void)munmap(ctx->map_base, ctx->map_len);
	ctx->map_base = nullptr;
	ctx->map_len = 0;
}

// The NUMA node of a HIP device and the CPUs of that node this process is allowed to run on (sysfs; empty when the
// platform does not say, e.g. a single-node guest).
void find_numa_cpus(int device, int *node, std::vector<int> *cpus)
{
	*node = -1;
	cpus->clear();
	char bus[64] = "";
	if(hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess){ (void)hipGetLastError(); return; }
	for(char *c = bus; *c; ++c){ *c = (char)tolower((unsigned char)*c); }
	char path[256];
	snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
	FILE *f = fopen(path, "r");
	if(!f){ return; }
	int n = -1;
	const int got = fscanf(f, "%d", &n);
	fclose(f);
	if(got != 1 || n < 0){ return; }
	snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", n);
	f = fopen(path, "r");
	if(!f){ return; }
	char list[4096] = "";
	const bool ok = fgets(list, sizeof(list), f) != nullptr;
	fclose(f);
	if(!ok){ return; }
	cpu_set_t allowed;
	CPU_ZERO(&allowed);
	if(sched_getaffinity(0, sizeof(allowed), &allowed) != 0){ return; }
	char *save = nullptr;                 // (strtok_r: contexts are created from several threads at once in the CLI's node mode)
	for(char *tok = strtok_r(list, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)){      // "0-47,96-143"
		int lo = 0, hi = 0;
		const int k = sscanf(tok, "%d-%d", &lo, &hi);
		if(k == 1){ hi = lo; }
		if(k < 1){ continue; }
		for(int c = lo; c <= hi && c < CPU_SETSIZE; ++c){ if(CPU_ISSET(c, &allowed)){ cpus->push_back(c); } }
	}
	*node = n;
}

}  // namespace kwage

namespace {

// Run the calling thread (and the threads it starts meanwhile) on the device's NUMA node for the lifetime of the object.
struct NumaScope {
	cpu_set_t before;
	bool active = false;
	explicit NumaScope(const kwage_ctx *ctx)
	{
		static const bool wanted = []() { const char *e = getenv("KWAGE_LOAD_NUMA"); return !(e && atoi(e) == 0); }();
		if(!wanted || ctx->numa_cpus.empty()){ return; }
		CPU_ZERO(&before);
		if(sched_getaffinity(0, sizeof(before), &before) != 0){ return; }
		cpu_set_t want;
		CPU_ZERO(&want);
		for(int c : ctx->numa_cpus){ CPU_SET(c, &want); }
		if(CPU_EQUAL(&want, &before)){ return; }
		active = sched_setaffinity(0, sizeof(want), &want) == 0;
	}
	~NumaScope() { if(active){ (void)sched_setaffinity(0, sizeof(before), &before); } }
};

}  // namespace

// ------------------------------------------------------------------------------------------
// database group
// ------------------------------------------------------------------------------------------
namespace {

// One device block for a matrix: physically contiguous where the driver has such a block (knob group_contiguous = 0:
// plain hipMalloc, also the fallback).
hipError_t allocate_block(uint64_t bytes, bool contiguous, void **out)
{
	hipError_t e = hipErrorOutOfMemory;
	*out = nullptr;
	if(contiguous && bytes >= (64ull << 20)){
		e = hipExtMallocWithFlags(out, bytes, hipDeviceMallocContiguous);
		if(e != hipSuccess){ (void)hipGetLastError(); *out = nullptr; }
	}
	if(e != hipSuccess){ e = device_malloc(out, bytes); }      // (out of memory: what the contexts' batch pools hold is released first)
	if(e != hipSuccess){ (void)hipGetLastError(); *out = nullptr; }
	return e;
}

// GB/s of the gather pattern (placement_probe_kernel) over an uninitialised block laid out like g's matrix; 0 on error.
double probe_block(kwage_group *g, const void *block, uint32_t windows = 1)
{
	kwage_ctx *ctx = g->ctx;
	const uint64_t stride16 = g->stride/16;
	const uint32_t lanes = (uint32_t)std::min<uint64_t>(LOADER_WAVE, stride16);
	const uint32_t chunks = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, g->stride/1024));
	const uint32_t rows_per_wave = 256;
	const uint32_t wgs = (uint32_t)std::max(ctx->ncu, 1);
	void *sink = nullptr;
	if(hipMalloc(&sink, 4) != hipSuccess){ (void)hipGetLastError(); return 0; }
	hipEvent_t e0 = nullptr, e1 = nullptr;
	double best = 0;
	if(hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess){
		for(int i = 0; i < 6; ++i){                          // (the first launch is the warm-up; the best of five counts)
			(void)hipEventRecord(e0, ctx->stream);
			hipLaunchKernelGGL(placement_probe_kernel, dim3(wgs), dim3(512), 100*1024, ctx->stream, (const dwords4*)block, g->nrows, stride16,
			                   chunks, lanes, rows_per_wave, windows, (uint32_t*)sink);
			(void)hipEventRecord(e1, ctx->stream);
			float ms = 0;
			if(hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0){ best = 0; break; }
			const double gbps = (double)wgs*8*rows_per_wave*chunks*lanes*16/((double)ms*1e-3)/1e9;
			if(i){ best = std::max(best, gbps); }
		}
	}
	if(hipGetLastError() != hipSuccess){ best = 0; }          // (a launch that failed says nothing about the block)
	if(e0){ (void)hipEventDestroy(e0); }
	if(e1){ (void)hipEventDestroy(e1); }
	(void)hipFree(sink);
	return best;
}

// The matrix's device block.  WHERE a large block lies in HBM decides 3-6 % of the gather kernels' rate -- the rate is a
// stable property of the physical region (two 105 GB blocks held at once: 6.88 and 6.64 TB/s, again and again; 23 blocks
// of 12 GB: 18 at 6.90-6.95, five at 6.56-6.71; profiles/r03_placement_probe.txt) -- so where the device has room for
// two candidates at once, both are allocated, the gather pattern is timed on each (a few ms) and the faster one is kept
// (knob group_placement_probe = 0: no second candidate).  Matrices below 4 GiB are not worth it, larger than half the free
// memory have no choice.  The price is paid after the choice: the driver wipes the released block (about 3 s per
// 100 GB) and allocations made meanwhile may wait for it -- worth it for a host that searches a resident database for
// hours, not for a one-shot run (the kwage command line turns the knob off).
hipError_t allocate_matrix(kwage_group *g)
{
	const bool choose = g->ctx->tune.group_placement_probe != 0, contiguous = g->ctx->tune.group_contiguous != 0;
	const auto t0 = std::chrono::steady_clock::now();
	auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
	void *a = nullptr;
	hipError_t e = allocate_block(g->alloc_bytes, contiguous, &a);
	if(e != hipSuccess){ return e; }
	const double ms_a = ms_since(t0);
	g->d_bits = (uint8_t*)a;
	g->placement_candidates = 1;
	if(!choose || g->alloc_bytes < (4ull << 30)){ return hipSuccess; }
	size_t free_bytes = 0, total_bytes = 0;
	double ms_b = 0, ms_probe = 0, ms_release = 0;
	void *b = nullptr;
	if(hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess && free_bytes >= g->alloc_bytes + std::max<uint64_t>(g->alloc_bytes/8, 2ull << 30)){
		const auto t1 = std::chrono::steady_clock::now();
		(void)allocate_block(g->alloc_bytes, contiguous, &b);
		ms_b = ms_since(t1);
	}
	(void)hipGetLastError();
	const auto t2 = std::chrono::steady_clock::now();
	double ra = probe_block(g, a);
	if(b){
		const double rb = probe_block(g, b);
		// only two MEASURED rates are compared: a probe that failed (0) decides nothing -- the first block stays, and the
		// record says that one candidate was measured
		if(ra > 0 && rb > 0){
			g->placement_candidates = 2;
			const bool keep_b = (rb > ra) != (g->ctx->tune.group_placement_probe < 0);     // (knob < 0: keep the SLOWER block -- measurements of what placement costs)
			g->placement_other_gbps = keep_b ? ra : rb;
			if(keep_b){ std::swap(a, b); ra = rb; }
		}
	}
	g->placement_kept_gbps = ra;
	// does the block kept mix regions of the device's memory?  the same pattern, all waves in the same quarter at a time
	// (only where the plain probe of the block worked: the band decision needs both rates)
	g->placement_windowed_gbps = ra > 0 ? probe_block(g, a, 4) : 0;
	// (3 % : the band form's regrouping and finish launches cost about 2 % of a C2 search -- 43 us, rocprofv3 -- so a block
	// whose windowed rate is only 1-2 % above the plain one is better served by the plain walk)
	g->mixes_regions = ra > 0 && g->placement_windowed_gbps > 1.03*ra;
	ms_probe = ms_since(t2);
	g->d_bits = (uint8_t*)a;
	if(b){
		const auto t3 = std::chrono::steady_clock::now();
		(void)hipFree(b);
		ms_release = ms_since(t3);
	}
	if(getenv("KWAGE_VERBOSE")){
		fprintf(stderr, "[kwage] matrix of %.1f GB: gather probe %.0f GB/s on the block kept (%.0f GB/s window after window%s), %.0f GB/s on the other candidate "
		        "(allocations %.0f + %.0f ms, probes %.0f ms, release %.0f ms)\n",
		        (double)g->alloc_bytes/1e9, g->placement_kept_gbps, g->placement_windowed_gbps, g->mixes_regions ? ": it mixes memory regions" : "",
		        g->placement_other_gbps, ms_a, ms_b, ms_probe, ms_release);
	}
	return hipSuccess;
}

// Allocate and clear a group's matrix of `nrows` rows for `column_capacity` columns.
int group_allocate(kwage_ctx *ctx, const kwage_params *params, uint64_t column_capacity, uint64_t nrows, kwage_group **out)
{
	*out = nullptr;
	int rc = check_params(params);
	if(rc){ return rc; }
	if(column_capacity == 0){ return fail(KWAGE_ERR_ARG, "kwage_group_create: column_capacity is 0"); }
	if((rc = set_device(ctx))){ return rc; }

	kwage_group *g = new (std::nothrow) kwage_group();
	if(!g){ return fail(KWAGE_ERR_DEVICE, "out of host memory"); }
	g->ctx = ctx;
	g->params = *params;
	g->nrows = nrows;
	const uint64_t row_bytes = (column_capacity + 7)/8;
	g->stride = (row_bytes + 127)/128*128;
	if(g->stride/16 > 0x7FFFFFFFull){ delete g; return fail(KWAGE_ERR_ARG, "kwage_group_create: row too wide"); }
	g->alloc_bytes = g->stride*g->nrows;
	hipError_t e = allocate_matrix(g);
	if(e != hipSuccess){
		const double gb = (double)g->alloc_bytes/1e9;
		delete g;
		return fail(KWAGE_ERR_DEVICE, "kwage_group_create: hipMalloc of %.3f GB for the bit matrix failed: %s",
		            gb, hipGetErrorString(e));
	}
	e = hipMalloc((void**)&g->d_valid, g->stride);
	if(e != hipSuccess){
		(void)hipFree(g->d_bits);
		delete g;
		return fail(KWAGE_ERR_DEVICE, "kwage_group_create: hipMalloc(valid mask) failed: %s", hipGetErrorString(e));
	}
	e = hipMemsetAsync(g->d_bits, 0, g->alloc_bytes, ctx->stream);
	if(e == hipSuccess){ e = hipMemsetAsync(g->d_valid, 0, g->stride, ctx->stream); }
	if(e == hipSuccess){ e = hipStreamSynchronize(ctx->stream); }
	if(e != hipSuccess){
		(void)hipFree(g->d_bits); (void)hipFree(g->d_valid);
		delete g;
		return fail(KWAGE_ERR_DEVICE, "kwage_group_create: clearing the bit matrix failed: %s", hipGetErrorString(e));
	}
	g->h_valid.assign(g->stride, 0);
	*out = g;
	return KWAGE_OK;
}

}  // namespace

extern "C" int kwage_group_create(kwage_ctx *ctx, const kwage_params *params, uint64_t column_capacity,
                                  kwage_group **out)
{
	if(!ctx || !params || !out){ return fail(KWAGE_ERR_ARG, "kwage_group_create: NULL argument"); }
	*out = nullptr;
	int rc = check_params(params);
	if(rc){ return rc; }
	return group_allocate(ctx, params, column_capacity, 1ull << params->log_2_filter_len, out);
}

extern "C" int kwage_group_create_sparse(kwage_ctx *ctx, const kwage_params *params, uint64_t column_capacity,
                                         const uint32_t *rows, uint64_t n_rows, kwage_group **out)
{
	if(!ctx || !params || !out || !rows){ return fail(KWAGE_ERR_ARG, "kwage_group_create_sparse: NULL argument"); }
	*out = nullptr;
	int rc = check_params(params);
	if(rc){ return rc; }
	if(n_rows == 0 || n_rows > 0xFFFFFFFFull){ return fail(KWAGE_ERR_ARG, "kwage_group_create_sparse: need 1 .. 2^32-1 rows"); }
	const uint64_t filter_len = 1ull << params->log_2_filter_len;
	for(uint64_t i = 0; i < n_rows; ++i){
		if(rows[i] >= filter_len || (i && rows[i] <= rows[i - 1])){
			return fail(KWAGE_ERR_ARG, "kwage_group_create_sparse: rows must be strictly ascending and below 2^%u", params->log_2_filter_len);
		}
	}
	kwage_group *g = nullptr;
	if((rc = group_allocate(ctx, params, column_capacity, n_rows, &g))){ return rc; }
	g->h_row_map.assign(rows, rows + n_rows);
	hipError_t e = hipMalloc((void**)&g->d_row_map, n_rows*sizeof(uint32_t));
	if(e == hipSuccess){ e = hipMemcpy(g->d_row_map, rows, n_rows*sizeof(uint32_t), hipMemcpyHostToDevice); }
	if(e != hipSuccess){
		kwage_group_destroy(g);
		return fail(KWAGE_ERR_DEVICE, "kwage_group_create_sparse: %s", hipGetErrorString(e));
	}
	*out = g;
	return KWAGE_OK;
}

extern "C" void kwage_group_destroy(kwage_group *g)
{
	if(!g){ return; }
	(void)hipSetDevice(g->ctx->device);
	release_mapping(g->ctx);
	(void)hipStreamSynchronize(g->ctx->gather_stream);
	(void)hipStreamSynchronize(g->ctx->slot[0].stream);
	(void)hipStreamSynchronize(g->ctx->slot[1].stream);
	if(g->d_bits){ (void)hipFree(g->d_bits); }
	if(g->d_valid){ (void)hipFree(g->d_valid); }
	if(g->d_row_map){ (void)hipFree(g->d_row_map); }
	delete g;
}

namespace {

// Reserve a 16-byte aligned byte range for `num_filter` new columns; mark them valid.
int group_reserve_columns(kwage_group *g, uint64_t num_filter, uint64_t *byte0)
{
	if(g->finalized){ return fail(KWAGE_ERR_STATE, "group is finalized; no more columns can be added"); }
	if(num_filter == 0){ return fail(KWAGE_ERR_ARG, "cannot add 0 columns"); }
	const uint64_t start = (g->next_byte + 15)/16*16;
	const uint64_t width = (num_filter + 7)/8;
	if(start + width > g->stride){
		return fail(KWAGE_ERR_ARG, "group capacity exceeded: need byte %llu of a %llu-byte row",
		            (unsigned long long)(start + width), (unsigned long long)g->stride);
	}
	for(uint64_t c = 0; c < num_filter; ++c){ g->h_valid[start + c/8] |= (uint8_t)(1u << (c%8)); }
	g->next_byte = start + width;
	g->num_columns += num_filter;
	*byte0 = start;
	return KWAGE_OK;
}

}  // namespace

extern "C" int kwage_group_add_columns(kwage_group *g, const void *host_rows, uint64_t host_row_stride,
                                       uint32_t num_filter, uint64_t *first_column)
{
	if(!g || !host_rows){ return fail(KWAGE_ERR_ARG, "kwage_group_add_columns: NULL argument"); }
	const uint64_t width = ((uint64_t)num_filter + 7)/8;
	if(host_row_stride < width){ return fail(KWAGE_ERR_ARG, "kwage_group_add_columns: host_row_stride < ceil(num_filter/8)"); }
	kwage_ctx *ctx = g->ctx;
	int rc = set_device(ctx);
	if(rc){ return rc; }
	uint64_t byte0 = 0;
	if((rc = group_reserve_columns(g, num_filter, &byte0))){ return rc; }

	// stage through a device buffer in chunks of rows, then scatter into the strided matrix
	const uint64_t chunk_rows = std::max<uint64_t>(1, std::min<uint64_t>(g->nrows, (64ull << 20)/host_row_stride));
	DevBuf stage;
	if((rc = stage.reserve(chunk_rows*host_row_stride))){ return rc; }
	const uint8_t *src = (const uint8_t*)host_rows;
	for(uint64_t r0 = 0; r0 < g->nrows; r0 += chunk_rows){
		const uint64_t nr = std::min(chunk_rows, g->nrows - r0);
		hipError_t e = hipMemcpyAsync(stage.p, src + r0*host_row_stride, (nr - 1)*host_row_stride + width,
		                              hipMemcpyHostToDevice, ctx->stream);
		if(e == hipSuccess){
			hipLaunchKernelGGL(place_rows_kernel, dim3(grid_for(nr*width/4 + 1, 256)), dim3(256), 0, ctx->stream,
			                   g->d_bits, g->stride, r0, byte0, (const uint8_t*)stage.p, host_row_stride, width, nr);
			e = hipGetLastError();
		}
		if(e == hipSuccess){ e = hipStreamSynchronize(ctx->stream); }
		if(e != hipSuccess){ stage.release(); return fail(KWAGE_ERR_DEVICE, "kwage_group_add_columns: %s", hipGetErrorString(e)); }
	}
	stage.release();
	if(first_column){ *first_column = byte0*8; }
	return KWAGE_OK;
}

namespace {

static const uint32_t LOAD_GANG = LOAD_GANG_MAX;      // files whose rows one copy kernel writes side by side (16 x 256 B = 4 KiB per matrix row)

uint64_t load_env_kb(const char *name, uint64_t fallback_bytes)
{
	const char *e = getenv(name);
	return (e && atoll(e) > 0) ? (uint64_t)atoll(e) << 10 : fallback_bytes;
}

// Open a database file for loading into `g`: header checks, optional CRC check, column reservation.
int open_source_for_group(kwage_group *g, const char *path, DbSliceSource &src, uint64_t *byte0)
{
	std::string err;
	if(!src.open(path, err)){ return fail(KWAGE_ERR_IO, "%s", err.c_str()); }
	const kwage_db_header &h = src.header;
	if(h.kmer_len != g->params.kmer_len || h.num_hash != g->params.num_hash ||
	   h.log_2_filter_len != g->params.log_2_filter_len || h.hash_func != g->params.hash_func){
		return fail(KWAGE_ERR_ARG, "%s: parameters (k=%u, hashes=%u, log2 len=%u, func=%d) differ from the group's",
		            path, h.kmer_len, h.num_hash, h.log_2_filter_len, h.hash_func);
	}
	if(h.num_filter == 0){ return fail(KWAGE_ERR_FORMAT, "%s: num_filter is 0", path); }
	// KWAGE_VERIFY_CRC=1: refuse a file whose slice block does not match the CRC32 in its header, as the reference's
	// merge does for its sources (merge_db.cpp:608-614; its `kwage` checks nothing, kwage.cpp:99-105).  A separate
	// pass over the file on the host, so off by default.
	static const bool verify_crc = load_env_flag("KWAGE_VERIFY_CRC", false);
	if(verify_crc){
		uint32_t crc = 0;
		if(!src.slice_crc32(crc, err)){ return fail(KWAGE_ERR_IO, "%s: %s", path, err.c_str()); }
		if(crc != h.crc32){ return fail(KWAGE_ERR_FORMAT, "%s: Invalid CRC32 value (header %08x, slices %08x)", path, h.crc32, crc); }
	}
	return group_reserve_columns(g, h.num_filter, byte0);
}

// Can the direct path take this file?  Raw layout, rows a multiple of 4 bytes, HSA lock available, and asked for
// (KWAGE_LOAD_DIRECT=1: it is NOT the default, see load_gang_direct).
bool direct_loadable(const DbSliceSource &src)
{
	static const bool mmap_ok = load_env_flag("KWAGE_LOAD_MMAP", true), direct_ok = load_env_flag("KWAGE_LOAD_DIRECT", false);
	return mmap_ok && direct_ok && src.header.compression == KWAGE_COMPRESSION_NONE && src.slice_size % 4 == 0 && hsa_lock().lock != nullptr;
}

// Direct path (opt-in, KWAGE_LOAD_DIRECT=1): rows of up to KWAGE_LOAD_GANG (default 16) raw files go from the page cache
// straight into the strided matrix.  Windows of the files are mapped and locked through HSA (no
// hipHostRegister/Unregister, which wait for the device), ONE copy kernel per window reads them over PCIe and writes
// the rows where they belong -- no staging buffer, no second pass over HBM.  The windows of up to two launches stay
// locked behind the one being queued and are released as their kernels finish.
// Why it is not the default (profiles/r02_loader_probe.txt, r02_e2e_cli_32files.txt, r02_load_105gb.txt):
//   32 x 268 MB files (8.6 GB matrix, 8 KiB stride): one file per launch 38-47 GB/s, gangs of 16 35 GB/s, staged path
//     40-41 GB/s -- the kernel's 64-byte PCIe reads cap it below the copy engine's 57 GB/s, so dropping the staging
//     hop buys little (an SDMA rect copy straight into the matrix reaches 33 GB/s, and hipMemcpyAsync does not
//     recognise HSA-locked memory: 19 GB/s);
//   392 files (105 GB matrix, 100 KB stride): 24 GB/s warm and 11 GB/s in the first runs after the files were
//     written, against 30-33 GB/s for the staged path, every time -- 256-byte pieces 100 KB apart miss the TLB and
//     the open DRAM page on every store, which hurts a kernel that holds PCIe reads in flight more than the staged
//     path's short HBM-to-HBM scatter bursts.
// *rows_done = rows of every file that are on their way when the call returns (all of them unless a window could not
// be mapped or locked; the caller finishes the rest through the staged paths).
int load_gang_direct(kwage_group *g, DbSliceSource *const *srcs, const uint64_t *byte0, uint32_t n, uint64_t *rows_done)
{
	kwage_ctx *ctx = g->ctx;
	static const uint64_t window_target = load_env_kb("KWAGE_LOAD_WINDOW_KB", 512ull << 20);
	const long page = sysconf(_SC_PAGESIZE);
	uint64_t total_width = 0;
	bool vec16 = true;
	for(uint32_t i = 0; i < n; ++i){ total_width += srcs[i]->slice_size; vec16 = vec16 && (srcs[i]->slice_size % 16 == 0); }
	const uint32_t ub = vec16 ? 16 : 4;
	// rows per launch: 512 MiB of file windows for one file, up to 2 GiB for a gang (a lock has a fixed cost too)
	const uint64_t win_rows = std::max<uint64_t>(1, std::min<uint64_t>(g->nrows, window_target*std::min<uint32_t>(n, 4)/total_width));
	*rows_done = 0;
	for(uint64_t r0 = 0; r0 < g->nrows; ){
		const uint64_t wr = std::min(win_rows, g->nrows - r0);
		GangArgs ga;
		memset(&ga, 0, sizeof(ga));
		ga.n = n;
		std::vector<kwage_ctx::LockedWindow> wins;
		bool ok = true;
		uint64_t max_bytes = 0;
		for(uint32_t i = 0; i < n && ok; ++i){
			const uint64_t width = srcs[i]->slice_size;
			const uint64_t off = DB_HEADER_BYTES + r0*width, off0 = off/page*page;
			const size_t maplen = (size_t)(off - off0 + wr*width);
			void *base = mmap(nullptr, maplen, PROT_READ, MAP_PRIVATE, srcs[i]->fd, (off_t)off0);      // (the lock faults the pages in)
			if(base == MAP_FAILED){ ok = false; break; }
			void *dev_view = nullptr;
			if(hsa_lock().lock(base, maplen, nullptr, 0, &dev_view) != 0 || !dev_view){ (void)munmap(base, maplen); ok = false; break; }
			wins.push_back(kwage_ctx::LockedWindow{base, maplen, nullptr, false});
			ga.f[i].src = (const uint8_t*)dev_view + (off - off0);
			ga.f[i].byte0 = byte0[i];
			ga.f[i].width = width;
			max_bytes = std::max(max_bytes, wr*width);
		}
		hipEvent_t ev = nullptr;
		if(ok){
			if(!ctx->spare_events.empty()){ ev = ctx->spare_events.back(); ctx->spare_events.pop_back(); }
			else if(hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess){ ev = nullptr; ok = false; }
		}
		if(!ok){
			for(auto &w : wins){ (void)hsa_lock().unlock(w.base); (void)munmap(w.base, w.len); }
			return KWAGE_OK;           // *rows_done tells the caller where to go on
		}
		const uint64_t items = (uint64_t)n*4*((max_bytes + (uint64_t)4*LOADER_WAVE*ub - 1)/((uint64_t)4*LOADER_WAVE*ub));     // (256-lane stretch, file, quarter) triples
		const uint32_t blocks = grid_for(items*LOADER_WAVE, 256, 256*8);
		if(ub == 16){ hipLaunchKernelGGL((copy_rows_gang_kernel<16>), dim3(blocks), dim3(256), 0, ctx->stream, g->d_bits, g->stride, r0, ga, wr, items); }
		else{ hipLaunchKernelGGL((copy_rows_gang_kernel<4>), dim3(blocks), dim3(256), 0, ctx->stream, g->d_bits, g->stride, r0, ga, wr, items); }
		hipError_t e = hipGetLastError();
		if(e == hipSuccess){ e = hipEventRecord(ev, ctx->stream); }
		for(auto &w : wins){ w.done = ev; }              // every window of the launch waits for the same event ...
		wins.back().owns_event = true;                   // ... and the last one to go returns it to the pool
		for(auto &w : wins){ ctx->locked.push_back(w); }
		if(e != hipSuccess){
			release_mapping(ctx);
			return fail(KWAGE_ERR_DEVICE, "loading database rows: %s", hipGetErrorString(e));
		}
		release_locked(ctx, 2*(size_t)n);
		r0 += wr;
		*rows_done = r0;
	}
	return KWAGE_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Copy-engine pipeline for raw files (the default): page cache -> staging buffer -> place_rows_kernel, at the copy
// engine's rate and without hipHostRegister.  Windows of a file (<= 256 MiB) are mapped and locked through HSA -- no
// device synchronisation, so pinning the next window and un-pinning the previous one overlap with the copy that is
// running --, hsa_amd_memory_async_copy (SDMA, linear) moves one window at a time into one of THREE staging buffers,
// the host waits for the copy's completion signal and launches place_rows_kernel behind it on the loading stream.
// Three windows are in flight: one being copied, one being scattered, one being pinned.  16 files x 268 MB:
// 48-52 GB/s; 392 files into a 105 GB matrix: 54 GB/s (tools/micro/sdma_stage_probe.hip, profiles/r02_loader_probe.txt)
// against 40 and 33 GB/s for the hipHostRegister + hipMemcpyAsync form, which stays as the fallback.
// ------------------------------------------------------------------------------------------------------------------
struct SdmaPipe {
	kwage_group *g = nullptr;
	hsa_agent_t gpu{}, cpu{};
	bool usable = false;
	uint64_t chunk_bytes = 0;
	struct Chunk { hsa_signal_t sig{}; bool sig_valid = false; int stage = 0; uint64_t row0 = 0, nr = 0, byte0 = 0, width = 0; };
	Chunk ring[3];
	uint64_t issued = 0, finished = 0;                     // chunk counters (ring index = counter % 3)
	struct Window { void *base; size_t len; uint64_t last_chunk; };
	std::deque<Window> windows;                            // locked windows, oldest first

	int init(kwage_group *grp)
	{
		g = grp;
		const HsaApi &h = hsa_api();
		if(!h.ok || !hsa_lock().lock){ return KWAGE_OK; }
		hsa_amd_pointer_info_t info;
		memset(&info, 0, sizeof(info));
		info.size = sizeof(info);
		if(h.pointer_info(g->d_bits, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || info.type == HSA_EXT_POINTER_TYPE_UNKNOWN){ return KWAGE_OK; }
		gpu = info.agentOwner;                             // the HSA agent behind this context's HIP device
		struct Find { const HsaApi *h; hsa_agent_t cpu; bool found; } f = {&h, {}, false};
		h.iterate_agents([](hsa_agent_t a, void *p) -> hsa_status_t {
			Find *fd = (Find*)p;
			hsa_device_type_t t;
			if(!fd->found && fd->h->agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_CPU){ fd->cpu = a; fd->found = true; }
			return HSA_STATUS_SUCCESS;
		}, &f);
		if(!f.found){ return KWAGE_OK; }
		cpu = f.cpu;
		for(auto &c : ring){ c.sig_valid = false; }
		for(int i = 0; i < 3; ++i){
			if(h.signal_create(1, 0, nullptr, &ring[i].sig) != HSA_STATUS_SUCCESS){ return KWAGE_OK; }
			ring[i].sig_valid = true;
			ring[i].stage = i;
		}
		HIP_TRY(hipStreamSynchronize(g->ctx->stream));       // nothing queued earlier may still read the staging buffers
		usable = true;
		return KWAGE_OK;
	}

	// the oldest unfinished chunk: wait for its copy, scatter it, release windows whose last chunk it was
	int finish_one()
	{
		kwage_ctx *ctx = g->ctx;
		Chunk &c = ring[finished % 3];
		hsa_api().signal_wait(c.sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
		hipLaunchKernelGGL(place_rows_kernel, dim3(grid_for(c.nr*c.width/4 + 1, 256)), dim3(256), 0, ctx->stream,
		                   g->d_bits, g->stride, c.row0, c.byte0, (const uint8_t*)ctx->load_dev[c.stage].p, c.width, c.width, c.nr);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(ctx->load_done[c.stage], ctx->stream));
		while(!windows.empty() && windows.front().last_chunk == finished){
			(void)hsa_lock().unlock(windows.front().base);
			(void)munmap(windows.front().base, windows.front().len);
			windows.pop_front();
		}
		++finished;
		return KWAGE_OK;
	}

	// Queue every row of `src` (raw layout): one window = one copy.  *rows_done = rows that are on their way (all of them
	// unless a window could not be mapped or locked, or the copy engine refused: the caller finishes the rest through the
	// other paths).
	int add_file(DbSliceSource &src, uint64_t byte0, uint64_t *rows_done)
	{
		kwage_ctx *ctx = g->ctx;
		const HsaApi &h = hsa_api();
		// 256 MiB per copy: pinning the next window (~12 us per MB) then takes less time than the copy that is running
		static const uint64_t window_target = load_env_kb("KWAGE_LOAD_WINDOW_KB", 256ull << 20);
		const uint64_t width = src.slice_size;
		const uint64_t win_rows = std::max<uint64_t>(1, std::min<uint64_t>(g->nrows, window_target/width));
		const long page = sysconf(_SC_PAGESIZE);
		int rc;
		for(int i = 0; i < 3; ++i){
			if(ctx->load_dev[i].cap < win_rows*width){
				// growing a staging buffer: every copy into the old one and every kernel that reads it must be done first
				if((rc = flush())){ return rc; }
				HIP_TRY(hipStreamSynchronize(ctx->stream));
				if((rc = ctx->load_dev[i].reserve(win_rows*width))){ return rc; }
			}
			if(!ctx->load_done[i]){ HIP_TRY(hipEventCreateWithFlags(&ctx->load_done[i], hipEventDisableTiming)); HIP_TRY(hipEventRecord(ctx->load_done[i], ctx->stream)); }
		}
		*rows_done = 0;
		for(uint64_t r0 = 0; r0 < g->nrows; ){
			const uint64_t wr = std::min(win_rows, g->nrows - r0);
			const uint64_t off = DB_HEADER_BYTES + r0*width, off0 = off/page*page;
			const size_t maplen = (size_t)(off - off0 + wr*width);
			// pages that are not in the page cache yet: have the kernel read the next two windows while this one is pinned
			// and copied (the lock below faults pages in one by one, at half the rate of a plain sequential read)
			(void)posix_fadvise(src.fd, (off_t)(off0 + maplen), (off_t)(2*win_rows*width), POSIX_FADV_WILLNEED);
			void *base = mmap(nullptr, maplen, PROT_READ, MAP_PRIVATE, src.fd, (off_t)off0);      // (the lock faults the pages in)
			if(base == MAP_FAILED){ return KWAGE_OK; }
			void *dev_view = nullptr;
			if(hsa_lock().lock(base, maplen, nullptr, 0, &dev_view) != 0 || !dev_view){ (void)munmap(base, maplen); return KWAGE_OK; }
			Chunk &c = ring[issued % 3];
			HIP_TRY(hipEventSynchronize(ctx->load_done[c.stage]));          // the scatter kernel that last read this staging buffer
			c.row0 = r0; c.nr = wr; c.byte0 = byte0; c.width = width;
			h.signal_store(c.sig, 1);
			if(h.async_copy(ctx->load_dev[c.stage].p, gpu, (const char*)dev_view + (off - off0), cpu, wr*width, 0, nullptr, c.sig) != HSA_STATUS_SUCCESS){
				(void)hsa_lock().unlock(base);
				(void)munmap(base, maplen);
				usable = false;                                 // the caller flushes and goes on with the staged paths
				return KWAGE_OK;
			}
			windows.push_back(Window{base, maplen, issued});
			++issued;
			if(ctx->load_progress){ __atomic_fetch_add(ctx->load_progress, wr*width, __ATOMIC_RELAXED); }      // pinned: no longer the reader's business
			r0 += wr;
			*rows_done = r0;
			if(issued - finished >= 2){ if((rc = finish_one())){ return rc; } }       // one copy stays in flight while the next window is pinned
		}
		return KWAGE_OK;
	}

	int flush()
	{
		int rc;
		while(finished < issued){ if((rc = finish_one())){ return rc; } }
		return KWAGE_OK;
	}

	~SdmaPipe()
	{
		if(g){ (void)flush(); }
		while(!windows.empty()){ (void)hsa_lock().unlock(windows.front().base); (void)munmap(windows.front().base, windows.front().len); windows.pop_front(); }
		for(auto &c : ring){ if(c.sig_valid){ (void)hsa_api().signal_destroy(c.sig); } }
	}
};

// Staged paths for rows first_row.. of one file: the pinned file mapping feeding the copy engine (raw files, from row 0)
// or pread / inflate into pinned buffers, each followed by place_rows_kernel.
int load_source_rows_staged(kwage_group *g, DbSliceSource &src, const char *path, uint64_t byte0, uint64_t first_row)
{
	kwage_ctx *ctx = g->ctx;
	const kwage_db_header &h = src.header;
	const uint64_t width = src.slice_size;
	std::string err;
	int rc = KWAGE_OK;
	// Raw files: map the file read-only, pin the mapping (hipHostRegister) and let the copy engine read the
	// page cache directly -- no pread copy into a staging buffer (that copy, not PCIe, limited the loader to
	// 30 GB/s; the mapping feeds H2D at the box's 57 GB/s, tools/micro/hostreg_probe.hip).  The copies of THIS
	// file are left in flight when the call returns, so the next file's mmap + pinning (3-4 ms per 256 MB)
	// overlaps with them; the mapping is released by the next call, by finalize, or when the group goes.
	// KWAGE_LOAD_MMAP=0, a compressed file, or a failure to map or pin falls back to the pread path below.
	static const bool mmap_ok = load_env_flag("KWAGE_LOAD_MMAP", true);
	uint64_t first_row_pread = first_row;
	// (KWAGE_LOAD_CHUNK_KB / KWAGE_LOAD_WINDOW_KB shrink the 64 MiB staging chunk and the 512 MiB window: tests)
	static const uint64_t chunk_target = load_env_kb("KWAGE_LOAD_CHUNK_KB", 64ull << 20);
	static const uint64_t window_target = load_env_kb("KWAGE_LOAD_WINDOW_KB", 512ull << 20);
	const uint64_t chunk_rows = std::max<uint64_t>(1, std::min<uint64_t>(g->nrows, chunk_target/width));
	const uint64_t chunk_bytes = chunk_rows*width;
	PinBuf *pin = ctx->load_pin;
	DevBuf *dev = ctx->load_dev;
	hipEvent_t *done = ctx->load_done;
	hipError_t e = hipSuccess;
	for(int i = 0; i < 2 && rc == KWAGE_OK; ++i){
		rc = dev[i].reserve(chunk_bytes);
		if(!rc && !done[i] && hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess){ rc = fail(KWAGE_ERR_DEVICE, "hipEventCreate failed"); }
	}
	if(rc){ return rc; }
	if(first_row == 0 && mmap_ok && h.compression == KWAGE_COMPRESSION_NONE){
		// windows of at most 512 MiB (whole chunks): pinned page-cache pages cannot be evicted, so a file larger
		// than host memory must never be pinned as a whole; two windows are alive at most (one being copied from)
		const uint64_t win_rows = std::max<uint64_t>(chunk_rows, (window_target/chunk_bytes)*chunk_rows);
		const long page = sysconf(_SC_PAGESIZE);
		bool fell_back = false;
		uint64_t r0 = 0;
		int cur = 0;
		for(; r0 < g->nrows; ){
			const uint64_t wr = std::min(win_rows, g->nrows - r0);
			const uint64_t off = DB_HEADER_BYTES + r0*width, off0 = off/page*page;
			const size_t maplen = (size_t)(off - off0 + wr*width);
			void *base = mmap(nullptr, maplen, PROT_READ, MAP_PRIVATE | MAP_POPULATE, src.fd, (off_t)off0);
			if(base == MAP_FAILED){ fell_back = true; break; }
			if(hipHostRegister(base, maplen, hipHostRegisterReadOnly) != hipSuccess){
				(void)hipGetLastError();
				(void)munmap(base, maplen);
				fell_back = true;
				break;
			}
			// Give up the previous window BEFORE queueing this one's copies: hipHostUnregister synchronises the device,
			// so doing it with the new copies in flight would serialise everything (measured: 28 instead of 40 GB/s).
			// The previous copies have had this window's whole mmap + pinning time to finish.
			release_mapping(ctx);
			ctx->map_base = base; ctx->map_len = maplen;
			const unsigned char *rows0 = (const unsigned char*)base + (off - off0);
			for(uint64_t c0 = 0; c0 < wr; c0 += chunk_rows, cur ^= 1){
				const uint64_t nr = std::min(chunk_rows, wr - c0);
				const uint64_t nb = nr*width;
				// staging buffer reuse is safe by stream order: this copy is queued behind the scatter kernel that read it
				e = hipMemcpyAsync(dev[cur].p, rows0 + c0*width, nb, hipMemcpyHostToDevice, ctx->stream);
				if(e == hipSuccess){
					hipLaunchKernelGGL(place_rows_kernel, dim3(grid_for(nb/4 + 1, 256)), dim3(256), 0, ctx->stream,
					                   g->d_bits, g->stride, r0 + c0, byte0, (const uint8_t*)dev[cur].p, width, width, nr);
					e = hipGetLastError();
				}
				if(e != hipSuccess){
					release_mapping(ctx);
					return fail(KWAGE_ERR_DEVICE, "kwage_group_add_db_file: %s", hipGetErrorString(e));
				}
			}
			if(!ctx->map_done && hipEventCreateWithFlags(&ctx->map_done, hipEventDisableTiming) != hipSuccess){ ctx->map_done = nullptr; }
			if(ctx->map_done){ (void)hipEventRecord(ctx->map_done, ctx->stream); }
			r0 += wr;
		}
		if(!fell_back){ return KWAGE_OK; }
		// could not map or pin a window (rows below r0 are already on their way): the pread path does the rest
		first_row_pread = r0;
	}
	release_mapping(ctx);       // the staging buffers below are shared with copies that may still be in flight

	// pread path, double-buffered: fill pinned buffer A (parallel pread / inflate) while buffer B is copied + scattered
	bool used[2] = {false, false};
	for(int i = 0; i < 2 && rc == KWAGE_OK; ++i){ rc = pin[i].reserve(chunk_bytes); }
	int cur = 0;
	for(uint64_t r0 = first_row_pread; r0 < g->nrows && rc == KWAGE_OK; r0 += chunk_rows, cur ^= 1){
		const uint64_t nr = std::min(chunk_rows, g->nrows - r0);
		const uint64_t nb = nr*width;
		if(used[cur]){ e = hipEventSynchronize(done[cur]); if(e != hipSuccess){ rc = fail(KWAGE_ERR_DEVICE, "%s", hipGetErrorString(e)); break; } }
		if(!src.read_rows(r0, nr, (unsigned char*)pin[cur].p, err)){ rc = fail(KWAGE_ERR_IO, "%s: %s", path, err.c_str()); break; }
		e = hipMemcpyAsync(dev[cur].p, pin[cur].p, nb, hipMemcpyHostToDevice, ctx->stream);
		if(e == hipSuccess){
			hipLaunchKernelGGL(place_rows_kernel, dim3(grid_for(nb/4 + 1, 256)), dim3(256), 0, ctx->stream,
			                   g->d_bits, g->stride, r0, byte0, (const uint8_t*)dev[cur].p, width, width, nr);
			e = hipGetLastError();
		}
		if(e == hipSuccess){ e = hipEventRecord(done[cur], ctx->stream); }
		if(e != hipSuccess){ rc = fail(KWAGE_ERR_DEVICE, "kwage_group_add_db_file: %s", hipGetErrorString(e)); break; }
		used[cur] = true;
	}
	(void)hipStreamSynchronize(ctx->stream);
	return rc;
}

// Sparse group: fetch only the listed slices of up to LOAD_GANG files -- I/O proportional to what the queries address,
// like the reference's seekg + read per slice (kwage.cpp:414-416) -- one host thread per file (several per file when
// the list is long) into ONE pinned staging buffer, one copy, then place_rows_kernel per file.
int load_gang_sparse(kwage_group *g, DbSliceSource *const *srcs, const char *const *paths, const uint64_t *byte0, uint32_t n)
{
	kwage_ctx *ctx = g->ctx;
	static const uint64_t chunk_target = load_env_kb("KWAGE_LOAD_CHUNK_KB", 64ull << 20);
	uint64_t total_width = 0;
	for(uint32_t i = 0; i < n; ++i){ total_width += srcs[i]->slice_size; }
	const uint64_t chunk_rows = std::max<uint64_t>(1, std::min<uint64_t>(g->nrows, chunk_target/total_width));
	const uint64_t chunk_bytes = chunk_rows*total_width;
	PinBuf *pin = ctx->load_pin;
	DevBuf *dev = ctx->load_dev;
	hipEvent_t *done = ctx->load_done;
	int rc = KWAGE_OK;
	release_mapping(ctx);       // the staging buffers are shared with copies of an earlier (dense) load
	for(int i = 0; i < 2 && rc == KWAGE_OK; ++i){
		rc = dev[i].reserve(chunk_bytes);
		if(!rc){ rc = pin[i].reserve(chunk_bytes); }
		if(!rc && !done[i] && hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess){ rc = fail(KWAGE_ERR_DEVICE, "hipEventCreate failed"); }
	}
	if(rc){ return rc; }
	bool used[2] = {false, false};
	int cur = 0;
	hipError_t e = hipSuccess;
	for(uint64_t r0 = 0; r0 < g->nrows && rc == KWAGE_OK; r0 += chunk_rows, cur ^= 1){
		const uint64_t nr = std::min(chunk_rows, g->nrows - r0);
		if(used[cur]){ e = hipEventSynchronize(done[cur]); if(e != hipSuccess){ rc = fail(KWAGE_ERR_DEVICE, "%s", hipGetErrorString(e)); break; } }
		// file i's nr slices land at offset nr * (widths of the files before it)
		std::vector<uint64_t> off(n + 1, 0);
		for(uint32_t i = 0; i < n; ++i){ off[i + 1] = off[i] + nr*srcs[i]->slice_size; }
		std::vector<std::string> errs(n);
		std::vector<char> ok(n, 1);
		const unsigned inner = (n == 1) ? 16u : 1u;      // one file: its list is split over threads; a gang: one thread per file
		auto fetch = [&](uint32_t i) {
			ok[i] = srcs[i]->read_row_list(g->h_row_map.data() + r0, nr, (unsigned char*)pin[cur].p + off[i], errs[i], inner) ? 1 : 0;
		};
		std::vector<std::thread> pool;
		for(uint32_t i = 1; i < n; ++i){ pool.emplace_back(fetch, i); }
		fetch(0);
		for(auto &t : pool){ t.join(); }
		for(uint32_t i = 0; i < n; ++i){
			if(!ok[i]){ rc = fail(KWAGE_ERR_IO, "%s: %s", paths[i], errs[i].c_str()); break; }
		}
		if(rc){ break; }
		e = hipMemcpyAsync(dev[cur].p, pin[cur].p, off[n], hipMemcpyHostToDevice, ctx->stream);
		for(uint32_t i = 0; i < n && e == hipSuccess; ++i){
			const uint64_t width = srcs[i]->slice_size;
			hipLaunchKernelGGL(place_rows_kernel, dim3(grid_for(nr*width/4 + 1, 256)), dim3(256), 0, ctx->stream,
			                   g->d_bits, g->stride, r0, byte0[i], (const uint8_t*)dev[cur].p + off[i], width, width, nr);
			e = hipGetLastError();
		}
		if(e == hipSuccess){ e = hipEventRecord(done[cur], ctx->stream); }
		if(e != hipSuccess){ rc = fail(KWAGE_ERR_DEVICE, "kwage_group_add_db_files: %s", hipGetErrorString(e)); break; }
		used[cur] = true;
	}
	(void)hipStreamSynchronize(ctx->stream);
	return rc;
}

}  // namespace

extern "C" int kwage_group_add_db_file(kwage_group *g, const char *path, uint64_t *first_column, uint32_t *num_filter)
{
	if(!g || !path){ return fail(KWAGE_ERR_ARG, "kwage_group_add_db_file: NULL argument"); }
	return kwage_group_add_db_files(g, &path, 1, first_column, num_filter);
}

extern "C" int kwage_group_add_db_files(kwage_group *g, const char *const *paths, uint32_t n, uint64_t *first_columns, uint32_t *num_filters)
{
	if(!g || !paths || n == 0){ return fail(KWAGE_ERR_ARG, "kwage_group_add_db_files: NULL argument"); }
	for(uint32_t i = 0; i < n; ++i){ if(!paths[i]){ return fail(KWAGE_ERR_ARG, "kwage_group_add_db_files: path %u is NULL", i); } }
	int rc = set_device(g->ctx);
	if(rc){ return rc; }
	const NumaScope on_the_gpus_node(g->ctx);      // this thread and the reader threads it starts, until the call returns
	// Files are taken in the order given (that is the column order).  Raw files go through the copy-engine pipeline
	// (SdmaPipe; KWAGE_LOAD_SDMA=0 disables it), or -- opt-in -- LOAD_GANG at a time through the direct copy kernel;
	// compressed files, sparse groups and whatever those paths cannot take go through the staged paths file by file.
	static const bool sdma_ok = load_env_flag("KWAGE_LOAD_SDMA", true) && load_env_flag("KWAGE_LOAD_MMAP", true);
	SdmaPipe pipe;
	if(sdma_ok && !g->d_row_map && !load_env_flag("KWAGE_LOAD_DIRECT", false)){ if((rc = pipe.init(g))){ return rc; } }
	for(uint32_t i0 = 0; i0 < n; ){
		DbSliceSource srcs[LOAD_GANG];
		DbSliceSource *ptrs[LOAD_GANG];
		uint64_t byte0[LOAD_GANG];
		uint32_t cnt = 0, n_direct = 0;
		static const uint32_t gang_max = []() { const char *e = getenv("KWAGE_LOAD_GANG"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= (int)LOAD_GANG) ? (uint32_t)v : LOAD_GANG; }();
		while(i0 + cnt < n && cnt < gang_max){
			DbSliceSource &src = srcs[cnt];
			if((rc = open_source_for_group(g, paths[i0 + cnt], src, &byte0[cnt]))){ return rc; }
			if(first_columns){ first_columns[i0 + cnt] = byte0[cnt]*8; }
			if(num_filters){ num_filters[i0 + cnt] = src.header.num_filter; }
			ptrs[cnt] = &src;
			++cnt;
			if(g->d_row_map){ continue; }                // sparse group: gangs of any files
			// whole-file loading: start reading the head of every file of the gang now (no-op for pages already cached)
			if(src.fd >= 0){ (void)posix_fadvise(src.fd, 0, (off_t)(512ull << 20), POSIX_FADV_WILLNEED); }
			if(!direct_loadable(src)){ break; }          // this file ends the gang and is staged on its own
			n_direct = cnt;
		}
		if(g->d_row_map){          // sparse group: only the listed slices of these files
			const char *gp[LOAD_GANG];
			for(uint32_t k = 0; k < cnt; ++k){ gp[k] = paths[i0 + k]; }
			if((rc = load_gang_sparse(g, ptrs, gp, byte0, cnt))){ return rc; }
			i0 += cnt;
			continue;
		}
		uint64_t rows_done = 0;
		if(n_direct){
			if((rc = load_gang_direct(g, ptrs, byte0, n_direct, &rows_done))){ return rc; }
		}
		for(uint32_t k = 0; k < cnt; ++k){
			const uint64_t progress_before = g->ctx->load_progress ? __atomic_load_n(g->ctx->load_progress, __ATOMIC_RELAXED) : 0;
			uint64_t from = (k < n_direct) ? rows_done : 0;
			if(from == 0 && pipe.usable && srcs[k].header.compression == KWAGE_COMPRESSION_NONE){
				if((rc = pipe.add_file(srcs[k], byte0[k], &from))){ return rc; }
				if(from < g->nrows){ if((rc = pipe.flush())){ return rc; } }       // the staged paths share the staging buffers
			}
			else if(pipe.usable){ if((rc = pipe.flush())){ return rc; } }
			if(from < g->nrows){
				if((rc = load_source_rows_staged(g, srcs[k], paths[i0 + k], byte0[k], from))){ return rc; }
				// its mapped path returns with copies and scatter kernels still in flight on the staging buffers the
				// copy-engine pipeline shares (and records no load_done event): nothing of it may be left when the pipe
				// writes into them again
				if(pipe.usable){ HIP_TRY(hipStreamSynchronize(g->ctx->stream)); }
			}
			if(g->ctx->load_progress){       // the whole file has been passed now (windows reported themselves as they were pinned)
				struct stat st;
				if(fstat(srcs[k].fd, &st) == 0 && st.st_size > 0){
					const uint64_t now = __atomic_load_n(g->ctx->load_progress, __ATOMIC_RELAXED), end = progress_before + (uint64_t)st.st_size;
					if(end > now){ __atomic_fetch_add(g->ctx->load_progress, end - now, __ATOMIC_RELAXED); }
				}
			}
		}
		i0 += cnt;
	}
	if(pipe.usable || pipe.issued){ if((rc = pipe.flush())){ return rc; } }
	return KWAGE_OK;
}

extern "C" int kwage_group_add_random_columns(kwage_group *g, uint64_t num_columns, uint64_t seed,
                                              uint32_t density_q8, uint64_t *first_column)
{
	if(!g){ return fail(KWAGE_ERR_ARG, "kwage_group_add_random_columns: NULL group"); }
	if(density_q8 > 256){ return fail(KWAGE_ERR_ARG, "density_q8 must be in [0,256]"); }
	kwage_ctx *ctx = g->ctx;
	int rc = set_device(ctx);
	if(rc){ return rc; }
	uint64_t byte0 = 0;
	if((rc = group_reserve_columns(g, num_columns, &byte0))){ return rc; }
	const uint64_t width = (num_columns + 7)/8;
	const uint64_t words = g->nrows*((width + 7)/8);
	hipLaunchKernelGGL(fill_random_kernel, dim3(grid_for(words, 256, 256*16)), dim3(256), 0, ctx->stream,
	                   g->d_bits, g->stride, g->nrows, byte0, width, seed, density_q8);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	if(first_column){ *first_column = byte0*8; }
	return KWAGE_OK;
}

extern "C" int kwage_group_set_bits(kwage_group *g, const uint32_t *rows, const uint64_t *columns, uint64_t n)
{
	if(!g || (n && (!rows || !columns))){ return fail(KWAGE_ERR_ARG, "kwage_group_set_bits: NULL argument"); }
	if(n == 0){ return KWAGE_OK; }
	kwage_ctx *ctx = g->ctx;
	int rc = set_device(ctx);
	if(rc){ return rc; }
	for(uint64_t i = 0; i < n; ++i){
		if(rows[i] >= g->nrows || columns[i] >= g->next_byte*8){
			return fail(KWAGE_ERR_ARG, "kwage_group_set_bits: (row %u, column %llu) outside the matrix", rows[i], (unsigned long long)columns[i]);
		}
	}
	DevBuf dr, dc;
	if((rc = dr.reserve(n*sizeof(uint32_t))) || (rc = dc.reserve(n*sizeof(uint64_t)))){ dr.release(); dc.release(); return rc; }
	hipError_t e = hipMemcpyAsync(dr.p, rows, n*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
	if(e == hipSuccess){ e = hipMemcpyAsync(dc.p, columns, n*sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream); }
	if(e == hipSuccess){
		hipLaunchKernelGGL(set_bits_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream,
		                   g->d_bits, g->stride, (const uint32_t*)dr.p, (const uint64_t*)dc.p, n);
		e = hipGetLastError();
	}
	if(e == hipSuccess){ e = hipStreamSynchronize(ctx->stream); }
	dr.release(); dc.release();
	if(e != hipSuccess){ return fail(KWAGE_ERR_DEVICE, "kwage_group_set_bits: %s", hipGetErrorString(e)); }
	return KWAGE_OK;
}

extern "C" int kwage_group_read_rows(kwage_group *g, const uint32_t *rows, uint64_t n, void *out, uint64_t out_stride)
{
	if(!g || (n && (!rows || !out))){ return fail(KWAGE_ERR_ARG, "kwage_group_read_rows: NULL argument"); }
	if(n == 0){ return KWAGE_OK; }
	const uint64_t row_bytes = g->next_byte;
	if(row_bytes == 0){ return fail(KWAGE_ERR_STATE, "kwage_group_read_rows: group has no columns"); }
	if(out_stride < row_bytes){ return fail(KWAGE_ERR_ARG, "kwage_group_read_rows: out_stride < row_bytes"); }
	for(uint64_t i = 0; i < n; ++i){
		if(rows[i] >= g->nrows){ return fail(KWAGE_ERR_ARG, "kwage_group_read_rows: row %u out of range", rows[i]); }
	}
	kwage_ctx *ctx = g->ctx;
	int rc = set_device(ctx);
	if(rc){ return rc; }
	DevBuf dr, dout;
	if((rc = dr.reserve(n*sizeof(uint32_t))) || (rc = dout.reserve(n*row_bytes))){ dr.release(); dout.release(); return rc; }
	hipError_t e = hipMemcpyAsync(dr.p, rows, n*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
	if(e == hipSuccess){
		hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n*row_bytes, 256)), dim3(256), 0, ctx->stream,
		                   (const uint8_t*)g->d_bits, g->stride, (const uint32_t*)dr.p, n, row_bytes, (uint8_t*)dout.p);
		e = hipGetLastError();
	}
	if(e == hipSuccess){
		e = hipMemcpy2DAsync(out, out_stride, dout.p, row_bytes, row_bytes, n, hipMemcpyDeviceToHost, ctx->stream);
	}
	if(e == hipSuccess){ e = hipStreamSynchronize(ctx->stream); }
	dr.release(); dout.release();
	if(e != hipSuccess){ return fail(KWAGE_ERR_DEVICE, "kwage_group_read_rows: %s", hipGetErrorString(e)); }
	return KWAGE_OK;
}

extern "C" int kwage_group_finalize(kwage_group *g)
{
	if(!g){ return fail(KWAGE_ERR_ARG, "kwage_group_finalize: NULL group"); }
	kwage_ctx *ctx = g->ctx;
	int rc = set_device(ctx);
	if(rc){ return rc; }
	HIP_TRY(hipMemcpyAsync(g->d_valid, g->h_valid.data(), g->stride, hipMemcpyHostToDevice, ctx->stream));
	// the matrix's bit density, from a few thousand rows (microseconds): what the early exit over long queries at t < 1 plans with
	// (and its densest column's: what the bound has to rule out before a 128-byte group can be dropped)
	g->density = 0.25;
	g->density_max = 1.0;          // (unknown: nothing is ever planned to be ruled out early)
	if(g->num_columns && g->nrows){
		const uint32_t n_sample = (uint32_t)std::min<uint64_t>(g->nrows, 4096);
		unsigned long long *d_probe = nullptr, h_probe[2] = {0, 0};
		if(hipMalloc((void**)&d_probe, 2*sizeof(unsigned long long)) == hipSuccess){
			hipError_t e = hipMemsetAsync(d_probe, 0, 2*sizeof(unsigned long long), ctx->stream);
			if(e == hipSuccess){
				hipLaunchKernelGGL(density_probe_kernel, dim3(n_sample), dim3(256), 0, ctx->stream, g->d_bits, g->stride, g->nrows, g->d_valid, n_sample, d_probe);
				hipLaunchKernelGGL(column_density_probe_kernel, dim3((uint32_t)((g->next_byte + 255)/256)), dim3(256), 0, ctx->stream, g->d_bits, g->stride, g->nrows, g->d_valid,
				                   g->next_byte, n_sample, (unsigned int*)(d_probe + 1));
				e = hipMemcpyAsync(h_probe, d_probe, sizeof(h_probe), hipMemcpyDeviceToHost, ctx->stream);
			}
			if(e == hipSuccess){ e = hipStreamSynchronize(ctx->stream); }
			if(e == hipSuccess){
				g->density = (double)h_probe[0]/((double)n_sample*(double)g->num_columns);
				g->density_max = (double)(h_probe[1] & 0xFFFFFFFFull)/(double)n_sample;
			}
			else{ (void)hipGetLastError(); }
			(void)hipFree(d_probe);
		}
	}
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	release_mapping(ctx);          // the last file's copies are done
	g->finalized = true;
	return KWAGE_OK;
}

extern "C" uint64_t kwage_group_num_columns(const kwage_group *g) { return g ? g->num_columns : 0; }
extern "C" uint64_t kwage_group_column_span(const kwage_group *g) { return g ? g->next_byte*8 : 0; }
extern "C" uint64_t kwage_group_row_bytes(const kwage_group *g) { return g ? g->next_byte : 0; }
extern "C" uint64_t kwage_group_row_stride(const kwage_group *g) { return g ? g->stride : 0; }
extern "C" uint64_t kwage_group_device_bytes(const kwage_group *g) { return g ? g->alloc_bytes : 0; }

extern "C" int kwage_group_placement(const kwage_group *g, uint32_t *candidates, double *kept_gbps, double *other_gbps, double *windowed_gbps)
{
	if(!g){ return fail(KWAGE_ERR_ARG, "kwage_group_placement: g is NULL"); }
	if(candidates){ *candidates = g->placement_candidates; }
	if(kept_gbps){ *kept_gbps = g->placement_kept_gbps; }
	if(other_gbps){ *other_gbps = g->placement_other_gbps; }
	if(windowed_gbps){ *windowed_gbps = g->placement_windowed_gbps; }
	return KWAGE_OK;
}

extern "C" int kwage_group_params(const kwage_group *g, kwage_params *out)
{
	if(!g || !out){ return fail(KWAGE_ERR_ARG, "kwage_group_params: NULL argument"); }
	*out = g->params;
	return KWAGE_OK;
}
