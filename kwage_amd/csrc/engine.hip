// kwage_amd/csrc/engine.hip -- device side of the C ABI declared in include/kwage_amd.h.
//
// One kwage_ctx = one GPU = one HIP stream.  A kwage_group owns the HBM-resident bit matrix of
// all same-parameter columns; kwage_search() runs, on the context's stream,
//     kmer_kernel  ->  and_kernel | count_kernel  ->  D2H of the hit list
// which together replace the reference's search() (kwage.cpp:340-541) for a whole batch of
// queries.  There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>                 // types only: every HSA function is looked up at run time (see HsaApi)
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <memory>
#include <mutex>
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
#include "internal.h"
#include "kernels.hpp"

using namespace kwage;

#define HIP_TRY(expr)                                                                          \
	do {                                                                                       \
		hipError_t _e = (expr);                                                                \
		if(_e != hipSuccess){                                                                  \
			return fail(KWAGE_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
			            __FILE__, __LINE__);                                                   \
		}                                                                                      \
	} while(0)

namespace {

// A device buffer that only ever grows (scratch reused across searches).
struct DevBuf {
	void *p = nullptr;
	uint64_t cap = 0;
	int reserve(uint64_t bytes)
	{
		if(bytes <= cap){ return KWAGE_OK; }
		if(p){ (void)hipFree(p); p = nullptr; cap = 0; }
		const uint64_t want = std::max<uint64_t>(bytes + bytes/4, 4096);
		HIP_TRY(hipMalloc(&p, want));
		cap = want;
		return KWAGE_OK;
	}
	void release()
	{
		if(p){ (void)hipFree(p); }
		p = nullptr; cap = 0;
	}
};

struct PinBuf {
	void *p = nullptr;
	uint64_t cap = 0;
	int reserve(uint64_t bytes)
	{
		if(bytes <= cap){ return KWAGE_OK; }
		if(p){ (void)hipHostFree(p); p = nullptr; cap = 0; }
		const uint64_t want = std::max<uint64_t>(bytes + bytes/4, 4096);
		HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
		cap = want;
		return KWAGE_OK;
	}
	void release()
	{
		if(p){ (void)hipHostFree(p); }
		p = nullptr; cap = 0;
	}
};

// Pinned host blocks for LONG hit lists, recycled between searches: a list of 100 M records (1.2 GB) crosses PCIe in
// 22 ms, but landing it in fresh pageable memory cost 0.2 s (a page fault per 4 KiB, one thread's memcpy) -- so the
// result array of a long list IS a pinned block, the D2H copy's destination, and kwage_result_free hands it back for
// the next search.  Results may outlive their context: the pool is shared, kwage_shutdown closes it.
struct PinnedPool {
	static const size_t MAX_CACHED_BLOCKS = 2;
	static const uint64_t MAX_CACHED_BYTES = 8ull << 30;
	std::mutex mu;
	bool open = true;
	std::vector<PinBuf> cached;
	int acquire(uint64_t bytes, PinBuf *out)
	{
		{
			std::lock_guard<std::mutex> lock(mu);
			size_t best = cached.size();
			for(size_t i = 0; i < cached.size(); ++i){
				if(cached[i].cap >= bytes && (best == cached.size() || cached[i].cap < cached[best].cap)){ best = i; }
			}
			if(best != cached.size()){
				*out = cached[best];
				cached.erase(cached.begin() + (long)best);
				return KWAGE_OK;
			}
		}
		out->p = nullptr; out->cap = 0;
		return out->reserve(bytes);
	}
	void release(PinBuf &b)
	{
		if(!b.p){ return; }
		{
			std::lock_guard<std::mutex> lock(mu);
			uint64_t held = 0;
			for(const PinBuf &c : cached){ held += c.cap; }
			if(open && cached.size() < MAX_CACHED_BLOCKS && held + b.cap <= MAX_CACHED_BYTES){
				cached.push_back(b);
				b.p = nullptr; b.cap = 0;
				return;
			}
		}
		b.release();
	}
	void close()
	{
		std::lock_guard<std::mutex> lock(mu);
		open = false;
		for(PinBuf &c : cached){ c.release(); }
		cached.clear();
	}
};

}  // namespace

// Everything one in-flight search owns.  A context has two slots so that a second search can be
// submitted (and its k-mer stage run) while the first one's results are still being collected.
struct Slot {
	hipStream_t stream = nullptr;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	hipEvent_t search_done = nullptr;   // recorded behind the slot's gather kernel(s)
	bool search_done_valid = false;
	// scratch, grown on demand and reused
	DevBuf rows, tables, partial;
	// One contiguous result block per search, so that a single D2H copy returns everything:
	//   [counters: 4 x u64 (hits, -, -, sink)] [nkmer: n x u32] [qthr: n x u32] [pad to 16] [hits: cap x 12 B]
	DevBuf result;
	uint64_t *d_counters = nullptr;
	uint32_t *d_nkmer = nullptr, *d_qthr = nullptr;
	kwage_hit *d_hits = nullptr;
	uint64_t hit_cap = 0, head_bytes = 0;
	PinBuf h_stage;        // host image of the head of the result block + the first SPEC_HITS records
	DevBuf sort_scratch;   // key / value buffers of the device hit sort (lists beyond SPEC_HITS only)
	// the submission occupying the slot
	bool busy = false;
	kwage_group *g = nullptr;
	kwage_batch *b = nullptr;
	const struct KmerLayout *lay = nullptr;     // the batch's layout for the group's k-mer length
	float threshold = 1.0f;
	uint32_t flags = 0;
	uint32_t launches = 0;
	char kernel_name[64] = "";          // the gather kernel launch_search_stage picked, with its template shape
	// and_walk_kernel's meeting place for (query, tile) pairs cut by a wave-share boundary: all zero between searches
	DevBuf walk_or, walk_done;
	// count_walk_kernel's: partial counters of cut pairs (overwritten before they are read) and the arrival counters of their trees (zero between searches)
	DevBuf cwalk_slab, cwalk_arrived;
	uint64_t staged_hits = 0;
	kwage_hit *ext_hits = nullptr;      // caller-owned device buffer (kwage_search_device) or null
	uint64_t ext_cap = 0;
	uint64_t *ext_count = nullptr;      // optional device word that receives the hit count in stream order
	// append mode (kwage_search_device_append_submit): *ext_count IS the hit counter -- not zeroed unless asked, so the
	// searches of several groups fill one list -- and col_base is added to every reported column
	bool append = false, append_reset = false;
	uint32_t col_base = 0;
};

// Kernel-selection knobs.  They are parsed ONCE, from the environment, when a context is created, and changed afterwards
// only through kwage_ctx_set_tuning (tests and tuning tools): nothing on the search path reads the environment.
struct Tuning {
	int64_t walk = 4;               // KWAGE_WALK: and_walk_kernel's rows in flight (4 or 2); 0 = always the tiled kernel
	int64_t walk_min_rows = -1;     // KWAGE_WALK_MIN_ROWS: batches with fewer rows use the tiled kernel (-1: 64 rows per wave of the chip)
	int64_t walk_max_kib = 16;      // KWAGE_WALK_MAX_KIB: widest row the walk form takes
	int64_t walk_early_exit = 0;    // KWAGE_WALK_EARLY_EXIT: use the walk form with early exit too (the tiled kernel stops sooner)
	int64_t walk_waves = 0;         // KWAGE_WALK_WAVES: exactly this many waves (tests: shares of every size); 0 = from the CU count
	int64_t walk_fences = 0;        // KWAGE_WALK_FENCES: agent-scope fences around the cut-pair count (measurement only)
	int64_t walk_one_wg_per_cu = 1; // KWAGE_WALK_ONE_WG_PER_CU: chip-filling launches of the persistent kernels use one workgroup of 8 waves per CU (0: workgroups of 4 waves, placed by the dispatcher)
	int64_t and_vec = 0;            // KWAGE_AND_CFG="vec,unroll,nt[,ldsKB[,block waves]]": shape of the tiled AND kernel (0 = by row width)
	int64_t and_unroll = 8;
	int64_t and_nt = 1;
	int64_t and_lds_kb = 0;         //   dynamic LDS per workgroup caps the waves per CU (tuning only)
	int64_t and_block_waves = SEARCH_THREADS/WAVE;
	int64_t narrow = 1;             // KWAGE_NARROW: several queries per wave for rows <= 512 B
	int64_t narrow_unroll = 0;      // KWAGE_NARROW_UNROLL: rows in flight per wave of the narrow AND kernel (0 = by the number of waves; 8, 16)
	int64_t force_segs = 0;         // KWAGE_FORCE_SEGS: cut every query's k-mer list into this many segments (tests)
	int64_t count_walk = 1;         // KWAGE_COUNT_WALK: the persistent count kernel where it applies
	int64_t count_walk_wpc = 8;     // KWAGE_COUNT_WALK_WPC: its waves per CU (8: 6335 GB/s at C2's shape, 12: 6271, 16: 6250, 20: 5876)
	int64_t count_walk_waves = 0;   // KWAGE_COUNT_WALK_WAVES: exactly this many waves (tests)
	int64_t count_walk_min_rows = -1;   // KWAGE_COUNT_WALK_MIN_ROWS: smaller batches use the tiled kernel (-1: 64 rows for each of its waves)
	int64_t count_walk_prefetch = 1;    // KWAGE_COUNT_WALK_PREFETCH: request the next four k-mers' rows before adding the current four (+1.3 % at C2's shape)
	int64_t count_narrow_kps = 8;   // KWAGE_COUNT_NARROW_KPS: k-mers per step of the narrow count kernel (8 or 4)
	int64_t hit_sort_host = 0;      // KWAGE_HIT_SORT=host: order long hit lists on the host (A/B runs, the fallback)
	int64_t hit_copy_piece_kb = 0;  // KWAGE_HIT_COPY_PIECE_KB: piece size of the copy-back of a long hit list (0 = default)
	int64_t shared_table_log2 = 0;  // KWAGE_SHARED_TABLE_LOG2: at least this many slots in a sample's shared distinct set (tests)
};

struct TuningName { const char *name; int64_t Tuning::*field; };
static const TuningName TUNING_NAMES[] = {
	{"walk", &Tuning::walk}, {"walk_min_rows", &Tuning::walk_min_rows}, {"walk_max_kib", &Tuning::walk_max_kib},
	{"walk_early_exit", &Tuning::walk_early_exit}, {"walk_waves", &Tuning::walk_waves}, {"walk_fences", &Tuning::walk_fences}, {"walk_one_wg_per_cu", &Tuning::walk_one_wg_per_cu},
	{"and_vec", &Tuning::and_vec}, {"and_unroll", &Tuning::and_unroll}, {"and_nt", &Tuning::and_nt}, {"and_lds_kb", &Tuning::and_lds_kb},
	{"and_block_waves", &Tuning::and_block_waves}, {"narrow", &Tuning::narrow}, {"narrow_unroll", &Tuning::narrow_unroll}, {"force_segs", &Tuning::force_segs},
	{"count_walk", &Tuning::count_walk}, {"count_walk_wpc", &Tuning::count_walk_wpc}, {"count_walk_waves", &Tuning::count_walk_waves},
	{"count_walk_min_rows", &Tuning::count_walk_min_rows}, {"count_walk_prefetch", &Tuning::count_walk_prefetch},
	{"count_narrow_kps", &Tuning::count_narrow_kps},
	{"hit_sort_host", &Tuning::hit_sort_host}, {"hit_copy_piece_kb", &Tuning::hit_copy_piece_kb}, {"shared_table_log2", &Tuning::shared_table_log2},
};

struct kwage_ctx {
	int device = -1;
	int ncu = 0;                        // compute units of the device (persistent grids are sized from it)
	Tuning tune;
	hipStream_t stream = nullptr;       // == slot[0].stream; loading, building and the synchronous calls use it
	Slot slot[2];
	DevBuf kmers;                       // kwage_hash_batch output
	// database loading: two pinned + two device staging buffers, kept across files
	PinBuf load_pin[2];
	DevBuf load_dev[3];                 // [2] is used by the copy-engine pipeline only (three chunks in flight)
	hipEvent_t load_done[3] = {nullptr, nullptr, nullptr};
	// zero-copy loading: the file mapping whose H2D copies may still be in flight on `stream`
	void *map_base = nullptr;
	size_t map_len = 0;
	hipEvent_t map_done = nullptr;      // recorded behind the last copy that reads the mapping
	// direct loading: file windows locked through HSA whose copy kernels may still be running, oldest first
	struct LockedWindow { void *base; size_t len; hipEvent_t done; bool owns_event; };     // the windows of one launch share its event; the last one owns it
	std::deque<LockedWindow> locked;
	std::vector<hipEvent_t> spare_events;
	volatile uint64_t *load_progress = nullptr;      // kwage_set_load_progress
	std::shared_ptr<PinnedPool> result_pool = std::make_shared<PinnedPool>();      // result arrays of long hit lists
	// CPUs of the NUMA node the device hangs on (empty: unknown, or the process may not run there): database loading
	// runs on them (the page-cache pages it pins and the staging traffic then stay on the GPU's side of the host)
	std::vector<int> numa_cpus;
	int numa_node = -1;
};

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

}  // namespace

struct kwage_group {
	kwage_ctx *ctx = nullptr;
	kwage_params params{};
	uint64_t nrows = 0;
	uint64_t stride = 0;           // bytes, multiple of 128
	uint64_t next_byte = 0;        // next free byte column within a row
	uint64_t num_columns = 0;      // valid columns
	uint8_t *d_bits = nullptr;
	uint8_t *d_valid = nullptr;
	uint64_t alloc_bytes = 0;
	std::vector<uint8_t> h_valid;
	bool finalized = false;
	// sparse group (kwage_group_create_sparse): the matrix holds only the listed rows of every file, in this order
	// (sorted, distinct); row indices from the k-mer stage are translated to positions in the list before the gather
	std::vector<uint32_t> h_row_map;
	uint32_t *d_row_map = nullptr;
};

// Where the k-mer positions of a batch's queries lie for ONE k-mer length: what the k-mer stage and the gather kernels
// index their row lists with.
struct KmerLayout {
	uint32_t k = 0;
	uint64_t total_pos = 0;        // sum over the queries of max(len - k + 1, 0)
	uint64_t max_pos = 0;
	uint64_t table_slots = 0;      // global hash-set slots needed by long queries
	uint64_t *d_pos_off = nullptr; // n+1: position prefix
	uint64_t *d_tab_off = nullptr; // n: slot offset of a long query's global distinct set
	// k-mer stage work list: one workgroup per chunk; a query above KM_LDS_SLOTS/2 positions is cut into chunks of
	// KM_CHUNK positions that share its global distinct set, everything shorter is one chunk
	uint32_t *d_chunk_q = nullptr;     // n_chunks: query of the chunk
	uint64_t *d_chunk_t0 = nullptr;    // n_chunks: its first position within the query
	uint64_t n_chunks = 0;
	bool multi_chunk = false;          // some query has more than one chunk
	std::vector<uint64_t> h_pos_off;
	~KmerLayout()
	{
		if(d_pos_off){ (void)hipFree(d_pos_off); }
		if(d_tab_off){ (void)hipFree(d_tab_off); }
		if(d_chunk_q){ (void)hipFree(d_chunk_q); }
		if(d_chunk_t0){ (void)hipFree(d_chunk_t0); }
	}
};

struct kwage_batch {
	kwage_ctx *ctx = nullptr;
	uint32_t n = 0;
	uint64_t total_len = 0;
	char *d_seqs = nullptr;
	uint64_t *d_seq_off = nullptr;
	std::vector<uint64_t> h_seq_off;
	// One layout per k-mer length the batch has been searched with (a database directory may hold files of several
	// k: two or three in practice).  A layout never changes once built, so searches with different k-mer lengths can
	// be in flight on the same batch side by side.
	std::vector<std::unique_ptr<KmerLayout>> layouts;
};

namespace {

int set_device(kwage_ctx *ctx)
{
	HIP_TRY(hipSetDevice(ctx->device));
	return KWAGE_OK;
}

// Lay the result block out for n queries and at least min_cap hit records. Growing the block keeps
// its head (counters + per-query arrays) when keep_head is set (hit-buffer growth mid-search).
int layout_result(Slot *sl, uint32_t n, uint64_t min_cap, bool keep_head)
{
	const uint64_t head = (32 + 8ull*n + 15)/16*16;
	uint64_t cap = std::max<uint64_t>(min_cap, 1u << 20);
	if(sl->result.cap >= head + sizeof(kwage_hit)){ cap = std::max(cap, (sl->result.cap - head)/sizeof(kwage_hit)); }
	const uint64_t need = head + cap*sizeof(kwage_hit);
	if(need > sl->result.cap){
		if(keep_head && sl->result.p && head == sl->head_bytes){
			void *np = nullptr;
			const uint64_t want = need + need/4;
			HIP_TRY(hipMalloc(&np, want));
			HIP_TRY(hipMemcpyAsync(np, sl->result.p, head, hipMemcpyDeviceToDevice, sl->stream));
			HIP_TRY(hipStreamSynchronize(sl->stream));
			(void)hipFree(sl->result.p);
			sl->result.p = np;
			sl->result.cap = want;
		}
		else{
			int rc = sl->result.reserve(need);
			if(rc){ return rc; }
		}
		cap = (sl->result.cap - head)/sizeof(kwage_hit);
	}
	char *base = (char*)sl->result.p;
	sl->d_counters = (uint64_t*)base;
	sl->d_nkmer = (uint32_t*)(base + 32);
	sl->d_qthr = (uint32_t*)(base + 32 + 4ull*n);
	sl->d_hits = (kwage_hit*)(base + head);
	sl->hit_cap = cap;
	sl->head_bytes = head;
	return KWAGE_OK;
}

uint32_t grid_for(uint64_t work_items, uint32_t block, uint32_t cap_blocks = 256*8)
{
	const uint64_t b = (work_items + block - 1)/block;
	return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(b, cap_blocks));
}

uint32_t host_table_log2(uint64_t npos)
{
	uint32_t lg = 6;
	while((1ull << lg) < 2*npos){ ++lg; }
	return lg;
}

// The batch's layout for k-mer length k: position prefix, global hash-set offsets and the k-mer stage's work list.
// Built on first use (one synchronous upload), kept for the life of the batch.
int batch_prepare(kwage_batch *b, uint32_t k, const KmerLayout **out)
{
	for(const auto &have : b->layouts){ if(have->k == k){ *out = have.get(); return KWAGE_OK; } }
	std::unique_ptr<KmerLayout> L(new (std::nothrow) KmerLayout());
	if(!L){ return fail(KWAGE_ERR_DEVICE, "out of host memory"); }
	L->k = k;
	const uint32_t n = b->n;
	L->h_pos_off.assign((size_t)n + 1, 0);
	std::vector<uint64_t> tab_off(n, 0);
	uint64_t slots = 0, maxp = 0;
	for(uint32_t i = 0; i < n; ++i){
		const uint64_t len = b->h_seq_off[i + 1] - b->h_seq_off[i];
		const uint64_t npos = (len >= k) ? (len - k + 1) : 0;
		L->h_pos_off[i + 1] = L->h_pos_off[i] + npos;
		maxp = std::max(maxp, npos);
		if(npos){
			const uint32_t lg = host_table_log2(npos);
			if((1ull << lg) > KM_LDS_SLOTS){
				tab_off[i] = slots;
				slots += (1ull << lg);
			}
		}
	}
	std::vector<uint32_t> chunk_q;
	std::vector<uint64_t> chunk_t0;
	chunk_q.reserve(n);
	chunk_t0.reserve(n);
	for(uint32_t i = 0; i < n; ++i){
		const uint64_t npos = L->h_pos_off[i + 1] - L->h_pos_off[i];
		const bool is_long = npos && (1ull << host_table_log2(npos)) > KM_LDS_SLOTS;      // the same rule as the table choice above
		if(!is_long || npos <= KM_CHUNK){ chunk_q.push_back(i); chunk_t0.push_back(0); continue; }
		L->multi_chunk = true;
		for(uint64_t t0 = 0; t0 < npos; t0 += KM_CHUNK){ chunk_q.push_back(i); chunk_t0.push_back(t0); }
	}
	if(chunk_q.size() > 0x7FFFFFFFull){ return fail(KWAGE_ERR_ARG, "batch too large for one k-mer launch"); }
	L->n_chunks = chunk_q.size();
	HIP_TRY(hipMalloc(&L->d_chunk_q, std::max<size_t>(chunk_q.size(), 1)*sizeof(uint32_t)));
	HIP_TRY(hipMalloc(&L->d_chunk_t0, std::max<size_t>(chunk_q.size(), 1)*sizeof(uint64_t)));
	HIP_TRY(hipMalloc(&L->d_pos_off, ((size_t)n + 1)*sizeof(uint64_t)));
	HIP_TRY(hipMalloc(&L->d_tab_off, std::max<size_t>(n, 1)*sizeof(uint64_t)));
	// (synchronous copies: the sources are locals, and the layout may be used on either search stream right away)
	if(L->n_chunks){
		HIP_TRY(hipMemcpy(L->d_chunk_q, chunk_q.data(), chunk_q.size()*sizeof(uint32_t), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(L->d_chunk_t0, chunk_t0.data(), chunk_t0.size()*sizeof(uint64_t), hipMemcpyHostToDevice));
	}
	HIP_TRY(hipMemcpy(L->d_pos_off, L->h_pos_off.data(), ((size_t)n + 1)*sizeof(uint64_t), hipMemcpyHostToDevice));
	if(n){ HIP_TRY(hipMemcpy(L->d_tab_off, tab_off.data(), (size_t)n*sizeof(uint64_t), hipMemcpyHostToDevice)); }
	L->total_pos = L->h_pos_off[n];
	L->max_pos = maxp;
	L->table_slots = slots;
	*out = L.get();
	b->layouts.push_back(std::move(L));
	return KWAGE_OK;
}

// Launch the k-mer stage on the ctx stream. rows/kmers_out may be null.
int launch_kmer_stage(Slot *sl, const kwage_params &p, kwage_batch *b, const KmerLayout *L, float threshold,
                      uint32_t *d_rows, uint64_t *d_kmers)
{
	int rc = layout_result(sl, b->n, 0, false);
	if(rc){ return rc; }
	if(L->table_slots){
		if((rc = sl->tables.reserve(L->table_slots*sizeof(uint64_t)))){ return rc; }
		HIP_TRY(hipMemsetAsync(sl->tables.p, 0xFF, L->table_slots*sizeof(uint64_t), sl->stream));
	}
	HIP_TRY(hipMemsetAsync(sl->d_counters, 0, 4*sizeof(uint64_t), sl->stream));
	if(b->n == 0){ return KWAGE_OK; }

	KmerArgs a;
	a.seqs = b->d_seqs;
	a.seq_off = b->d_seq_off;
	a.pos_off = L->d_pos_off;
	a.tab_off = L->d_tab_off;
	a.g_tables = (unsigned long long*)sl->tables.p;
	a.k = p.kmer_len;
	a.num_hash = p.num_hash;
	a.row_mask = (p.log_2_filter_len >= 32) ? 0xFFFFFFFFu : ((1u << p.log_2_filter_len) - 1u);
	a.threshold = threshold;
	a.complete_match = (threshold == 1.0f) ? 1 : 0;      // kwage.cpp:349
	a.rows = d_rows;
	a.kmers_out = d_kmers;
	a.nkmer = sl->d_nkmer;
	a.qthr = sl->d_qthr;
	a.total_kmers = nullptr;           // summed on the host from nkmer[] (a per-workgroup atomic serialises)
	a.shared_lg = 0;
	a.bloom_bits = nullptr;
	a.chunk_q = L->d_chunk_q;
	a.chunk_t0 = L->d_chunk_t0;
	if(L->multi_chunk){      // the chunks of a long query add their new k-mers into nkmer[q]
		HIP_TRY(hipMemsetAsync(sl->d_nkmer, 0, (size_t)b->n*sizeof(uint32_t), sl->stream));
	}
	// workgroup and LDS table sized for the longest query of the batch: queries whose table would not fit
	// KM_LDS_SLOTS use the global tables laid out by batch_prepare (same rule there)
	uint32_t slots = 64;
	while(slots < KM_LDS_SLOTS && slots < 2*L->max_pos){ slots *= 2; }
	a.lds_slots = slots;
	const uint32_t threads = (L->max_pos <= 192) ? 64 : (L->max_pos <= 768) ? 128 : KM_THREADS;
	hipLaunchKernelGGL(kmer_kernel, dim3((uint32_t)L->n_chunks), dim3(threads), (size_t)slots*sizeof(uint64_t), sl->stream, a);
	HIP_TRY(hipGetLastError());
	if(L->multi_chunk){      // thresholds of the queries whose k-mers were counted by several workgroups
		hipLaunchKernelGGL(kmer_finish_kernel, dim3((b->n + 255)/256), dim3(256), 0, sl->stream, a, b->n);
		HIP_TRY(hipGetLastError());
	}
	return KWAGE_OK;
}

static const uint32_t WALK_MIN_ROWS_PER_WAVE = 64;   // and_walk_kernel: below this share per wave the tiled kernel is used
static const uint32_t WALK_WAVES_PER_CU = 8;         // 2 workgroups: 2048 waves measured 1-2 % faster than 4096 (half the cut pairs)

uint32_t search_blocks(const SearchArgs &a)
{
	const uint64_t tiles = (uint64_t)a.n_queries*a.segs*a.chunks;
	return (uint32_t)((tiles + 3)/4);
}

// Launch shape of a persistent kernel that wants `want_waves` waves.  A chip-filling launch (8 waves per CU) uses ONE
// workgroup of 8 waves per CU -- a dynamic-LDS pad of more than half a CU's LDS keeps a second workgroup off --, so
// that every CU runs exactly 8 waves; smaller launches use workgroups of four waves wherever the dispatcher puts them.
struct WalkShape { uint32_t wgs, wg_waves; size_t lds; };
static const size_t WALK_PAD_LDS = 100*1024;

WalkShape walk_shape(const Tuning &tn, uint64_t want_waves, uint64_t ncu)
{
	WalkShape w;
	const bool pinned = tn.walk_one_wg_per_cu && want_waves == ncu*WALK_WG_WAVES;
	w.wg_waves = pinned ? (uint32_t)WALK_WG_WAVES : 4u;
	w.wgs = (uint32_t)((want_waves + w.wg_waves - 1)/w.wg_waves);
	w.lds = pinned ? WALK_PAD_LDS : 0;
	return w;
}

// Shape of the tiled AND kernel: VEC 16-byte vectors per lane, UNROLL rows in flight, nontemporal loads, and (tuning
// only) dynamic LDS per workgroup and waves per workgroup.  Defaults come from measurements on MI355X.
struct AndCfg { int vec, unroll, nt, lds_bytes, block_waves; };

AndCfg and_config(const Tuning &t, uint32_t units_per_row)
{
	AndCfg c;
	c.vec = (t.and_vec == 1 || t.and_vec == 2 || t.and_vec == 4) ? (int)t.and_vec : ((units_per_row >= 4*WAVE) ? 2 : 1);
	c.unroll = (t.and_unroll == 4 || t.and_unroll == 16 || t.and_unroll == 32) ? (int)t.and_unroll : 8;
	if((c.unroll == 16 && c.vec == 4) || (c.unroll == 32 && c.vec != 1)){ c.unroll = 8; }       // shapes that are not instantiated
	c.nt = t.and_nt ? 1 : 0;      // +4-10 % on MI355X: each row byte is consumed once per (query, tile)
	c.lds_bytes = (t.and_lds_kb > 0 && t.and_lds_kb <= 160) ? (int)t.and_lds_kb*1024 : 0;
	c.block_waves = (t.and_block_waves == 1 || t.and_block_waves == 2) ? (int)t.and_block_waves : SEARCH_THREADS/WAVE;
	return c;
}

template <int VEC, int UNROLL, bool NT>
void launch_and(const SearchArgs &a, hipStream_t s, const AndCfg &c)
{
	const uint64_t tiles = (uint64_t)a.n_queries*a.segs*a.chunks;
	const uint32_t bw = (uint32_t)c.block_waves;
	const dim3 grid((uint32_t)((tiles + bw - 1)/bw)), block(bw*WAVE);
	if(a.segs > 1){
		hipLaunchKernelGGL((and_kernel<VEC, UNROLL, NT, true>), grid, block, (size_t)c.lds_bytes, s, a);
	}
	else{
		hipLaunchKernelGGL((and_kernel<VEC, UNROLL, NT, false>), grid, block, (size_t)c.lds_bytes, s, a);
	}
}

template <int VEC, bool NT>
void launch_and_u(const SearchArgs &a, hipStream_t s, const AndCfg &c)
{
	if(c.unroll == 4){ launch_and<VEC, 4, NT>(a, s, c); }
	else if(c.unroll == 16 && VEC < 4){ launch_and<VEC, 16, NT>(a, s, c); }
	else if(c.unroll == 32 && VEC == 1){ launch_and<VEC, 32, NT>(a, s, c); }
	else{ launch_and<VEC, 8, NT>(a, s, c); }
}

template <bool NT>
void launch_and_v(const SearchArgs &a, hipStream_t s, const AndCfg &c)
{
	if(c.vec == 1){ launch_and_u<1, NT>(a, s, c); }
	else if(c.vec == 2){ launch_and_u<2, NT>(a, s, c); }
	else{ launch_and_u<4, NT>(a, s, c); }
}

template <int PLANES, int NH>
void launch_count(const SearchArgs &a, hipStream_t s)
{
	if(a.segs > 1){
		hipLaunchKernelGGL((count_kernel<PLANES, NH, true>), dim3(search_blocks(a)), dim3(SEARCH_THREADS), 0, s, a);
	}
	else{
		hipLaunchKernelGGL((count_kernel<PLANES, NH, false>), dim3(search_blocks(a)), dim3(SEARCH_THREADS), 0, s, a);
	}
}

template <int PLANES, int G, int KPS>
void launch_count_narrow(const SearchArgs &a, hipStream_t s)
{
	const uint64_t waves = ((uint64_t)a.n_queries + G - 1)/G;
	const dim3 grid((uint32_t)((waves + 3)/4)), block(SEARCH_THREADS);
	switch(a.num_hash){
		case 1: hipLaunchKernelGGL((count_narrow_kernel<PLANES, 1, G, KPS>), grid, block, 0, s, a); break;
		case 2: hipLaunchKernelGGL((count_narrow_kernel<PLANES, 2, G, KPS>), grid, block, 0, s, a); break;
		case 3: hipLaunchKernelGGL((count_narrow_kernel<PLANES, 3, G, KPS>), grid, block, 0, s, a); break;
		case 4: hipLaunchKernelGGL((count_narrow_kernel<PLANES, 4, G, KPS>), grid, block, 0, s, a); break;
		default: hipLaunchKernelGGL((count_narrow_kernel<PLANES, 5, G, KPS>), grid, block, 0, s, a); break;
	}
}

template <int PLANES, int G>
void launch_count_narrow_k(const SearchArgs &a, hipStream_t s, int kps)
{
	if(kps == 4){ launch_count_narrow<PLANES, G, 4>(a, s); }
	else{ launch_count_narrow<PLANES, G, 8>(a, s); }
}

template <int PLANES>
void launch_count_nh(const SearchArgs &a, hipStream_t s)
{
	switch(a.num_hash){
		case 1: launch_count<PLANES, 1>(a, s); break;
		case 2: launch_count<PLANES, 2>(a, s); break;
		case 3: launch_count<PLANES, 3>(a, s); break;
		case 4: launch_count<PLANES, 4>(a, s); break;
		default: launch_count<PLANES, 5>(a, s); break;
	}
}

// counter planes for counts up to `max_count`: the smallest instantiated size whose bits hold it
uint32_t planes_for(uint64_t max_count)
{
	uint32_t bits = 1;
	while(bits < 32 && (max_count >> bits) != 0){ ++bits; }
	return (bits <= 7) ? 7 : (bits <= 10) ? 10 : (bits <= 14) ? 14 : (bits <= 20) ? 20 : 32;
}

void launch_count_planes(uint32_t planes, const SearchArgs &a, hipStream_t s)
{
	switch(planes){
		case 7: launch_count_nh<7>(a, s); break;
		case 10: launch_count_nh<10>(a, s); break;
		case 14: launch_count_nh<14>(a, s); break;
		case 20: launch_count_nh<20>(a, s); break;
		default: launch_count_nh<32>(a, s); break;
	}
}

template <int PLANES, int NH, bool PF>
void launch_count_walk(const SearchArgs &a, const CountWalkArgs &wa, const WalkShape &w, hipStream_t s)
{
	if(w.lds > 48*1024){ (void)hipFuncSetAttribute((const void*)count_walk_kernel<PLANES, NH, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w.lds); }
	hipLaunchKernelGGL((count_walk_kernel<PLANES, NH, PF>), dim3(w.wgs), dim3(w.wg_waves*WAVE), w.lds, s, a, wa, a.rows, a.pos_off, a.nkmer, a.qthr);
}

template <int PLANES, bool PF>
void launch_count_walk_nh(const SearchArgs &a, const CountWalkArgs &wa, const WalkShape &w, hipStream_t s)
{
	switch(a.num_hash){
		case 1: launch_count_walk<PLANES, 1, PF>(a, wa, w, s); break;
		case 2: launch_count_walk<PLANES, 2, PF>(a, wa, w, s); break;
		case 3: launch_count_walk<PLANES, 3, PF>(a, wa, w, s); break;
		case 4: launch_count_walk<PLANES, 4, PF>(a, wa, w, s); break;
		default: launch_count_walk<PLANES, 5, PF>(a, wa, w, s); break;
	}
}

template <bool PF>
void launch_count_walk_planes(uint32_t planes, const SearchArgs &a, const CountWalkArgs &wa, const WalkShape &w, hipStream_t s)
{
	switch(planes){
		case 7: launch_count_walk_nh<7, PF>(a, wa, w, s); break;
		case 10: launch_count_walk_nh<10, PF>(a, wa, w, s); break;
		case 14: launch_count_walk_nh<14, PF>(a, wa, w, s); break;
		case 20: launch_count_walk_nh<20, PF>(a, wa, w, s); break;
		default: launch_count_walk_nh<32, PF>(a, wa, w, s); break;
	}
}

template <int PLANES>
int launch_count_combine(const SearchArgs &a, uint32_t seg_planes, hipStream_t s)
{
	const size_t lds = (size_t)(COMBINE_WAVES/2)*PLANES*WAVE*16;
	if(lds > 48*1024){      // 32 planes only (queries above 2^20 positions); the attribute is per device, so set it per launch
		HIP_TRY(hipFuncSetAttribute((const void*)count_combine_kernel<PLANES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	}
	hipLaunchKernelGGL((count_combine_kernel<PLANES>), dim3((a.units_per_row + WAVE - 1)/WAVE, a.n_queries), dim3(COMBINE_WAVES*WAVE), lds, s, a, seg_planes);
	return KWAGE_OK;
}

// How many segments to cut each query's k-mer list into: none while the launch already has
// enough waves to fill the chip; otherwise enough to reach ~TARGET_TILES waves (about 8 per CU, ~2x the bytes in flight that cover HBM latency), but never segments
// shorter than MIN_SEG_KMERS k-mers.  The force_segs knob forces a count (tests).
void choose_segments(SearchArgs &a, uint64_t max_kmers, uint64_t max_segs, int64_t force_segs)
{
	static const uint64_t TARGET_TILES = 2048, MIN_SEG_KMERS = 64;
	const uint64_t MAX_SEGS = max_segs;
	a.segs = 1;
	a.seg_kmers = (uint32_t)std::max<uint64_t>(max_kmers, 1);
	if(a.n_queries > 65535){ return; }      // the combine kernels index queries with gridDim.y
	uint64_t want = 1;
	if(force_segs > 0){ want = (uint64_t)force_segs; }
	else{
		const uint64_t tiles = (uint64_t)a.n_queries*a.chunks;
		if(tiles >= TARGET_TILES || max_kmers < 2*MIN_SEG_KMERS){ return; }
		want = std::min<uint64_t>((TARGET_TILES + tiles - 1)/tiles, max_kmers/MIN_SEG_KMERS);
	}
	want = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(want, MAX_SEGS), std::max<uint64_t>(max_kmers, 1)));
	if(want <= 1){ return; }
	a.seg_kmers = (uint32_t)((max_kmers + want - 1)/want);
	a.segs = (uint32_t)((max_kmers + a.seg_kmers - 1)/a.seg_kmers);
}

// A scratch buffer the kernels leave all zero: cleared when it is (re)allocated, never per search.
int reserve_zeroed(DevBuf &buf, uint64_t bytes, hipStream_t s)
{
	if(bytes <= buf.cap){ return KWAGE_OK; }
	int rc = buf.reserve(bytes);
	if(rc){ return rc; }
	HIP_TRY(hipMemsetAsync(buf.p, 0, buf.cap, s));
	return KWAGE_OK;
}

// Launch the gather+reduce kernel(s) for the current batch.
int launch_search_stage(Slot *sl, kwage_group *g, kwage_batch *b, const KmerLayout *L, float threshold, uint32_t flags,
                        kwage_hit *d_hits, uint64_t cap, unsigned long long *hit_count)
{
	const Tuning &tn = g->ctx->tune;
	const uint64_t ncu = (uint64_t)std::max(g->ctx->ncu, 1);
	SearchArgs a;
	a.db = g->d_bits;
	a.stride = g->stride;
	a.units_per_row = (uint32_t)(g->stride/16);
	a.valid = g->d_valid;
	a.rows = (const uint32_t*)sl->rows.p;
	a.pos_off = L->d_pos_off;
	a.nkmer = sl->d_nkmer;
	a.qthr = sl->d_qthr;
	a.num_hash = g->params.num_hash;
	a.n_queries = b->n;
	a.hits = d_hits;
	a.cap = cap;
	a.hit_count = hit_count;
	a.early_exit = (flags & KWAGE_SEARCH_EARLY_EXIT) ? 1 : 0;
	a.partial = nullptr;
	a.col_base = sl->col_base;
	int rc;

	if(threshold == 1.0f){
		const AndCfg cfg = and_config(tn, a.units_per_row);
		a.chunks = (a.units_per_row + WAVE*cfg.vec - 1)/(WAVE*cfg.vec);
		choose_segments(a, L->max_pos, 4096, tn.force_segs);
		if((uint64_t)a.n_queries*a.segs*a.chunks/4 + 1 > 0x7FFFFFFFull){ return fail(KWAGE_ERR_ARG, "batch too large for one launch"); }
		// narrow rows: several queries per wave (tools/bench_narrow.py)
		if(tn.narrow && a.segs == 1 && a.units_per_row <= 32 && a.n_queries >= 64){
			const uint32_t G = (a.units_per_row <= 4) ? 16 : (a.units_per_row <= 8) ? 8 : (a.units_per_row <= 16) ? 4 : 2;
			const uint64_t waves = ((uint64_t)a.n_queries + G - 1)/G;
			const dim3 grid((uint32_t)((waves + 3)/4)), block(SEARCH_THREADS);
			// few waves (10 k queries of 1 kb: ten per CU): sixteen rows in flight per wave instead of eight
			const int unroll = (tn.narrow_unroll == 8 || tn.narrow_unroll == 16) ? (int)tn.narrow_unroll : ((waves < ncu*16) ? 16 : 8);
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "and_narrow_kernel<%u,%d>", G, unroll);
#define KWAGE_NARROW_CASE(GG) case GG: \
				if(unroll == 16){ hipLaunchKernelGGL((and_narrow_kernel<GG, 16>), grid, block, 0, sl->stream, a); } \
				else{ hipLaunchKernelGGL((and_narrow_kernel<GG, 8>), grid, block, 0, sl->stream, a); } break;
			switch(G){
				KWAGE_NARROW_CASE(16) KWAGE_NARROW_CASE(8) KWAGE_NARROW_CASE(4)
				default: KWAGE_NARROW_CASE(2)
			}
#undef KWAGE_NARROW_CASE
			HIP_TRY(hipGetLastError());
			return KWAGE_OK;
		}
		// rows of 3..16 KiB: the walk form (a persistent grid, every wave walks an equal share of the batch's row
		// list over the whole width of a column tile; kernels.hpp and_walk_kernel).  Worth it once every wave of
		// the chip gets a few dozen rows; smaller batches stay with the tiled kernel and its row-list segments.
		const int walk_unroll = (int)tn.walk;
		const uint64_t walk_min_rows = (tn.walk_min_rows >= 0) ? (uint64_t)tn.walk_min_rows : (uint64_t)WALK_MIN_ROWS_PER_WAVE*256*WALK_WAVES_PER_CU;
		const uint32_t kib = (a.units_per_row + WAVE - 1)/WAVE;
		// (the kernel handles wider rows as several balanced column tiles -- the walk_max_kib knob raises the limit --
		// but 125 KB rows measured no gain over the tiled kernel)
		const uint32_t walk_max_kib = (uint32_t)std::max<int64_t>(tn.walk_max_kib, 0);
		const uint32_t coltiles = (kib + 15)/16, walk_ch = (kib + coltiles - 1)/coltiles;     // balanced tiles of <= 16 KiB
		// With early exit the tiled kernel wins: a tile that holds no candidate column stops after a few rows even
		// when another tile of the same query holds a hit, whereas a walking wave covers the hit column's whole row
		// width and never stops (C2 with early exit: 0.64 ms tiled, 1.29 ms walk).
		const bool walk_ee_ok = !a.early_exit || tn.walk_early_exit;
		const uint64_t walk_slots = (uint64_t)coltiles*L->total_pos;
		if(walk_unroll && walk_ee_ok && kib >= 3 && kib <= walk_max_kib && walk_slots*a.num_hash >= walk_min_rows && walk_slots > 0){
			// WALK_WAVES_PER_CU waves per CU, all resident at once (__launch_bounds__(256, 4) allows twice as many),
			// fewer when the batch is small: a wave should have WALK_MIN_ROWS_PER_WAVE rows to walk
			const uint64_t chip_waves = ncu*WALK_WAVES_PER_CU;
			const uint64_t want_waves = (tn.walk_waves > 0) ? std::min<uint64_t>((uint64_t)tn.walk_waves, walk_slots)
				: std::max<uint64_t>(1, std::min<uint64_t>(chip_waves, walk_slots*a.num_hash/WALK_MIN_ROWS_PER_WAVE));
			const WalkShape shape = walk_shape(tn, want_waves, ncu);
			const uint32_t wgs = shape.wgs;
			const uint64_t waves = (uint64_t)wgs*shape.wg_waves;
			WalkArgs wa;
			wa.total_slots = walk_slots;
			wa.per_wave = (walk_slots + waves - 1)/waves;
			wa.coltiles = coltiles;
			// cut-pair slots: one per wave, 16 KiB each whatever CH is; the kernel leaves them zero
			if((rc = reserve_zeroed(sl->walk_or, waves*16*1024, sl->stream))){ return rc; }
			if((rc = reserve_zeroed(sl->walk_done, waves*2*sizeof(uint32_t), sl->stream))){ return rc; }
			wa.orbuf = (uint32_t*)sl->walk_or.p;
			wa.done = (uint32_t*)sl->walk_done.p;
			wa.full_fences = tn.walk_fences ? 1 : 0;
			a.segs = 1;
			a.chunks = coltiles;
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "and_walk_kernel<%u,%d>", walk_ch, walk_unroll == 2 ? 2 : 4);
			const dim3 grid(wgs), block(shape.wg_waves*WAVE);
#define KWAGE_WALK_LAUNCH(...) do { \
				if(shape.lds > 48*1024){ (void)hipFuncSetAttribute((const void*)and_walk_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shape.lds); } \
				hipLaunchKernelGGL((and_walk_kernel<__VA_ARGS__>), grid, block, shape.lds, sl->stream, a, wa, a.rows, a.pos_off, a.nkmer); } while(0)
#define KWAGE_WALK_CASE(CH) case CH: \
				if(walk_unroll == 2){ KWAGE_WALK_LAUNCH(CH, 2); } else{ KWAGE_WALK_LAUNCH(CH, 4); } break;
			switch(walk_ch){
				KWAGE_WALK_CASE(3) KWAGE_WALK_CASE(4) KWAGE_WALK_CASE(5) KWAGE_WALK_CASE(6) KWAGE_WALK_CASE(7)
				KWAGE_WALK_CASE(8) KWAGE_WALK_CASE(9) KWAGE_WALK_CASE(10) KWAGE_WALK_CASE(11) KWAGE_WALK_CASE(12)
				KWAGE_WALK_CASE(13) KWAGE_WALK_CASE(14) KWAGE_WALK_CASE(15)
				default: KWAGE_WALK_CASE(16)
			}
#undef KWAGE_WALK_CASE
#undef KWAGE_WALK_LAUNCH
			HIP_TRY(hipGetLastError());
			return KWAGE_OK;
		}
		if(a.segs > 1){
			const uint64_t bytes = (uint64_t)a.n_queries*g->stride;
			if((rc = sl->partial.reserve(bytes))){ return rc; }
			HIP_TRY(hipMemsetAsync(sl->partial.p, 0xFF, bytes, sl->stream));
			a.partial = (uint32_t*)sl->partial.p;
		}
		snprintf(sl->kernel_name, sizeof(sl->kernel_name), "and_kernel<%d,%d,%s>%s", cfg.vec, cfg.unroll, cfg.nt ? "nt" : "t", a.segs > 1 ? "+segments" : "");
		if(cfg.nt){ launch_and_v<true>(a, sl->stream, cfg); }
		else{ launch_and_v<false>(a, sl->stream, cfg); }
		if(a.segs > 1){
			hipLaunchKernelGGL(and_combine_kernel, dim3((a.units_per_row + 255)/256, a.n_queries), dim3(256), 0, sl->stream, a);
		}
	}
	else{
		a.chunks = (a.units_per_row + WAVE - 1)/WAVE;
		// counter planes: enough bits for the largest possible num_query_kmer of the batch
		const uint32_t planes = planes_for(L->max_pos);
		const bool narrow = tn.narrow && a.units_per_row <= 32 && a.units_per_row > 8 && a.n_queries >= 64 && planes <= 14;
		// The persistent form (count_walk_kernel): equal shares of the batch's (query, KiB tile, position) list per wave,
		// pairs cut by share boundaries added up as a tree through memory.  Taken when the batch gives every wave of the
		// launch its 64 rows and no early exit is asked for (a persistent wave has no tile of its own to give up): long
		// queries need no segment slab and no combine pass then (1 x 100 kb: 5410 vs 4813 GB/s, 4 x 1 Mb: 6431 vs 5705),
		// no batch size ends in a part-filled round of waves, and in a host's software pipeline its resident waves do not
		// share the CUs with the next batch's k-mer stage the way the tiled kernel's 13 000 workgroups do (C2 at t = 0.8
		// inside bench.py: 1.988 vs 2.077 ms, profiles/r03_c2t_bench_ab.txt).  (Alone on the chip, batches whose tiles
		// all fit in one round are 2-3 % faster through the tiled kernel -- 300 queries 0.582 vs 0.598 ms --: not worth
		// a rule of their own.)  Early exit, tiny batches and forced segment counts go on below.
		if(tn.count_walk && !a.early_exit && !narrow && tn.force_segs <= 0 && L->total_pos > 0){
			const uint64_t slots = (uint64_t)a.chunks*L->total_pos;
			const uint64_t chip_waves = ncu*(uint64_t)std::max<int64_t>(tn.count_walk_wpc, 1);
			const uint64_t min_rows = (tn.count_walk_min_rows >= 0) ? (uint64_t)tn.count_walk_min_rows : (uint64_t)WALK_MIN_ROWS_PER_WAVE*chip_waves;
			const uint64_t want_waves = (tn.count_walk_waves > 0) ? std::min<uint64_t>((uint64_t)tn.count_walk_waves, slots)
				: std::max<uint64_t>(1, std::min<uint64_t>(chip_waves, slots*a.num_hash/WALK_MIN_ROWS_PER_WAVE));
			const WalkShape shape = walk_shape(tn, want_waves, ncu);
			const uint64_t waves = (uint64_t)shape.wgs*shape.wg_waves;
			if(slots*a.num_hash >= min_rows){
				CountWalkArgs wa;
				wa.total_slots = slots;
				wa.per_wave = (slots + waves - 1)/waves;
				wa.coltiles = a.chunks;
				if((rc = sl->cwalk_slab.reserve(waves*2*planes*1024))){ return rc; }
				if((rc = reserve_zeroed(sl->cwalk_arrived, waves*CWALK_LEVELS*sizeof(uint32_t), sl->stream))){ return rc; }
				wa.slab = (uint32_t*)sl->cwalk_slab.p;
				wa.arrived = (uint32_t*)sl->cwalk_arrived.p;
				a.segs = 1;
				snprintf(sl->kernel_name, sizeof(sl->kernel_name), "count_walk_kernel<%u,%u%s>", planes, std::min(a.num_hash, 5u), tn.count_walk_prefetch ? ",pf" : "");
				if(tn.count_walk_prefetch){ launch_count_walk_planes<true>(planes, a, wa, shape, sl->stream); }
				else{ launch_count_walk_planes<false>(planes, a, wa, shape, sl->stream); }
				HIP_TRY(hipGetLastError());
				return KWAGE_OK;
			}
		}
		// Long queries: segments of the k-mer list are counted by different waves into a slab of partial counters
		// and added by count_combine_kernel (a tree per (query, 64 units)); a segment's counters need only the
		// planes its own k-mer count can reach.
		choose_segments(a, L->max_pos, 1024, tn.force_segs);
		uint32_t seg_planes = (a.segs > 1) ? planes_for(a.seg_kmers) : planes;
		// keep the slab of partial counters bounded (1 GiB)
		while(a.segs > 1 && (uint64_t)a.n_queries*a.segs*seg_planes*g->stride > (1ull << 30)){
			const uint64_t want = a.segs/2;
			a.seg_kmers = (uint32_t)((L->max_pos + want - 1)/std::max<uint64_t>(want, 1));
			a.segs = (uint32_t)((L->max_pos + a.seg_kmers - 1)/a.seg_kmers);
			seg_planes = (a.segs > 1) ? planes_for(a.seg_kmers) : planes;
		}
		if((uint64_t)a.n_queries*a.segs*a.chunks/4 + 1 > 0x7FFFFFFFull){ return fail(KWAGE_ERR_ARG, "batch too large for one launch"); }
		if(a.segs > 1){
			if((rc = sl->partial.reserve((uint64_t)a.n_queries*a.segs*seg_planes*g->stride))){ return rc; }
			a.partial = (uint32_t*)sl->partial.p;
		}
		if(narrow && a.segs == 1){
			// one reference file (<= 2048 columns = 16 units) or two: 4 resp. 2 queries per wave
			const int kps = (tn.count_narrow_kps == 4) ? 4 : 8;
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "count_narrow_kernel<%u,%u,%d,%d>", planes, a.num_hash, a.units_per_row <= 16 ? 4 : 2, kps);
			if(a.units_per_row <= 16){
				if(planes == 7){ launch_count_narrow_k<7, 4>(a, sl->stream, kps); }
				else if(planes == 10){ launch_count_narrow_k<10, 4>(a, sl->stream, kps); }
				else{ launch_count_narrow_k<14, 4>(a, sl->stream, kps); }
			}
			else{
				if(planes == 7){ launch_count_narrow_k<7, 2>(a, sl->stream, kps); }
				else if(planes == 10){ launch_count_narrow_k<10, 2>(a, sl->stream, kps); }
				else{ launch_count_narrow_k<14, 2>(a, sl->stream, kps); }
			}
			HIP_TRY(hipGetLastError());
			return KWAGE_OK;
		}
		if(a.segs > 1){
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "count_kernel<%u,%u>+segments->%u", seg_planes, std::min(a.num_hash, 5u), planes);
		}
		else{
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "count_kernel<%u,%u>", planes, std::min(a.num_hash, 5u));
		}
		launch_count_planes(seg_planes, a, sl->stream);
		if(a.segs > 1){
			HIP_TRY(hipGetLastError());
			switch(planes){
				case 7: rc = launch_count_combine<7>(a, seg_planes, sl->stream); break;
				case 10: rc = launch_count_combine<10>(a, seg_planes, sl->stream); break;
				case 14: rc = launch_count_combine<14>(a, seg_planes, sl->stream); break;
				case 20: rc = launch_count_combine<20>(a, seg_planes, sl->stream); break;
				default: rc = launch_count_combine<32>(a, seg_planes, sl->stream); break;
			}
			if(rc){ return rc; }
		}
	}
	HIP_TRY(hipGetLastError());
	return KWAGE_OK;
}

static const uint64_t SPEC_HITS = 8192;     // hit records copied back together with the counters
static const uint64_t SORT_SCRATCH_KEEP = 1ull << 30;     // device sort buffers above this size are freed after use

struct SearchOutcome {
	uint64_t staged_hits = 0;      // hit records already in the slot's h_stage
	uint64_t n_hits = 0;
	uint64_t total_kmers = 0;
	float kmer_ms = 0, search_ms = 0;
	uint32_t launches = 0;
	char kernel_name[64] = "";
};

// Enqueue, on the slot's stream: search kernel(s) + ONE D2H copy that brings back the counters, the
// per-query arrays and (own buffer) the first SPEC_HITS records.
int enqueue_search_and_copy(Slot *sl)
{
	int rc;
	kwage_group *g = sl->g;
	kwage_batch *b = sl->b;
	const bool timing = (sl->flags & KWAGE_SEARCH_TIMING) != 0;
	const bool own = (sl->ext_hits == nullptr && sl->ext_cap == 0);
	const uint64_t cap = own ? sl->hit_cap : sl->ext_cap;
	kwage_hit *d_hits = own ? sl->d_hits : sl->ext_hits;
	if(b->n && g->num_columns){
		// The gather kernels of the two slots must not run side by side (they would only share the HBM
		// bandwidth and stretch each other): this one starts when the other slot's has finished.  The
		// k-mer stage enqueued before and the copy-back enqueued after are what overlaps.
		Slot *other = (sl == &g->ctx->slot[0]) ? &g->ctx->slot[1] : &g->ctx->slot[0];
		if(other->search_done_valid){ HIP_TRY(hipStreamWaitEvent(sl->stream, other->search_done, 0)); }
		if(sl->append && sl->append_reset){ HIP_TRY(hipMemsetAsync(sl->ext_count, 0, sizeof(uint64_t), sl->stream)); }      // a new list starts here
		if(timing){ HIP_TRY(hipEventRecord(sl->ev[2], sl->stream)); }
		unsigned long long *hit_count = sl->append ? (unsigned long long*)sl->ext_count : (unsigned long long*)sl->d_counters;
		if((rc = launch_search_stage(sl, g, b, sl->lay, sl->threshold, sl->flags, d_hits, cap, hit_count))){ return rc; }
		if(timing){ HIP_TRY(hipEventRecord(sl->ev[3], sl->stream)); }
		HIP_TRY(hipEventRecord(sl->search_done, sl->stream));
		sl->search_done_valid = true;
		++sl->launches;
	}
	else if(sl->append && sl->append_reset){ HIP_TRY(hipMemsetAsync(sl->ext_count, 0, sizeof(uint64_t), sl->stream)); }      // nothing to search: an empty list all the same
	if(sl->append){         // the running total of the caller's list comes back with the head of the result block
		HIP_TRY(hipMemcpyAsync(sl->d_counters, sl->ext_count, sizeof(uint64_t), hipMemcpyDeviceToDevice, sl->stream));
	}
	else if(sl->ext_count){      // the caller's exchange buffer carries its own record count (no host round trip)
		HIP_TRY(hipMemcpyAsync(sl->ext_count, sl->d_counters, sizeof(uint64_t), hipMemcpyDeviceToDevice, sl->stream));
	}
	sl->staged_hits = own ? std::min<uint64_t>(SPEC_HITS, cap) : 0;
	const uint64_t bytes = sl->head_bytes + sl->staged_hits*sizeof(kwage_hit);
	if((rc = sl->h_stage.reserve(bytes))){ return rc; }
	HIP_TRY(hipMemcpyAsync(sl->h_stage.p, sl->result.p, bytes, hipMemcpyDeviceToHost, sl->stream));
	return KWAGE_OK;
}

// First half of a search: validate, lay out the slot, enqueue the whole device pipeline. Returns at once.
int submit_search(Slot *sl, kwage_group *g, kwage_batch *b, float threshold, uint32_t flags,
                  kwage_hit *ext_hits, uint64_t ext_cap, uint64_t *ext_count = nullptr,
                  bool append = false, bool append_reset = false, uint32_t col_base = 0)
{
	kwage_ctx *ctx = g->ctx;
	int rc;
	if(sl->busy){ return fail(KWAGE_ERR_STATE, "search slot is busy: collect the pending search first"); }
	if(!g->finalized){ return fail(KWAGE_ERR_STATE, "kwage_group_finalize() must be called before searching"); }
	if(b->ctx != ctx){ return fail(KWAGE_ERR_ARG, "batch and group belong to different contexts"); }
	if(!(threshold > 0.0f) || threshold > 1.0f){      // options.cpp:186-191
		return fail(KWAGE_ERR_ARG, "search threshold must satisfy 0 < t <= 1");
	}
	if((rc = set_device(ctx))){ return rc; }
	const KmerLayout *L = nullptr;
	if((rc = batch_prepare(b, g->params.kmer_len, &L))){ return rc; }
	if(L->max_pos*g->params.num_hash > 0xFFFFFFFFull){      // the kernels index a query's row list with 32 bits
		return fail(KWAGE_ERR_ARG, "a query of %llu k-mer positions x %u hash functions exceeds 2^32 rows", (unsigned long long)L->max_pos, g->params.num_hash);
	}
	if((rc = sl->rows.reserve(std::max<uint64_t>(L->total_pos*g->params.num_hash, 1)*sizeof(uint32_t)))){ return rc; }
	sl->g = g; sl->b = b; sl->lay = L; sl->threshold = threshold; sl->flags = flags;
	sl->ext_hits = ext_hits; sl->ext_cap = ext_cap; sl->ext_count = ext_count;
	sl->append = append; sl->append_reset = append_reset; sl->col_base = col_base;
	sl->launches = 0;

	const bool timing_kmer = (flags & KWAGE_SEARCH_TIMING) && (flags & KWAGE_SEARCH_TIMING_KMER);
	if(timing_kmer){ HIP_TRY(hipEventRecord(sl->ev[0], sl->stream)); }
	if((rc = launch_kmer_stage(sl, g->params, b, L, threshold, (uint32_t*)sl->rows.p, nullptr))){ return rc; }
	if(g->d_row_map && b->n){      // sparse group: row index -> position in the group's row list (counter 2 = indices not listed)
		// (a workgroup per 4096 row indices of the longest query, so that a genome-length query is not one workgroup's job)
		const uint64_t per_q = std::max<uint64_t>(1, (L->max_pos*g->params.num_hash + 4095)/4096);
		const uint64_t wgs_per_q = std::min<uint64_t>(per_q, std::max<uint64_t>(1, 0x7FFFFFFFull/b->n));
		hipLaunchKernelGGL(remap_rows_kernel, dim3((uint32_t)(b->n*wgs_per_q)), dim3(256), 0, sl->stream, (uint32_t*)sl->rows.p, L->d_pos_off, sl->d_nkmer,
		                   g->params.num_hash, g->d_row_map, (uint32_t)g->h_row_map.size(), (unsigned long long*)sl->d_counters + 2, (uint32_t)wgs_per_q);
		HIP_TRY(hipGetLastError());
	}
	if(timing_kmer){ HIP_TRY(hipEventRecord(sl->ev[1], sl->stream)); }
	if((rc = enqueue_search_and_copy(sl))){ return rc; }
	sl->busy = true;
	return KWAGE_OK;
}

// Second half: wait for the slot's stream, grow the hit buffer and re-run the search kernel if it
// overflowed (own buffer only), report counts and timings. Frees the slot.
int collect_search(Slot *sl, SearchOutcome *out)
{
	if(!sl->busy){ return fail(KWAGE_ERR_STATE, "no pending search in this slot"); }
	kwage_ctx *ctx = sl->g->ctx;
	int rc = set_device(ctx);
	if(rc){ sl->busy = false; return rc; }
	const bool own = (sl->ext_hits == nullptr && sl->ext_cap == 0);
	const bool timing = (sl->flags & KWAGE_SEARCH_TIMING) != 0;
	const bool timing_kmer = timing && (sl->flags & KWAGE_SEARCH_TIMING_KMER) != 0;
	sl->busy = false;                       // whatever happens below, the slot is released
	while(true){
		HIP_TRY(hipStreamSynchronize(sl->stream));     // (polling an event instead measured no faster)
		const uint64_t *hc = (const uint64_t*)sl->h_stage.p;
		if(hc[2] != 0){
			return fail(KWAGE_ERR_STATE, "%llu row indices of this batch are not among the rows of the sparse group (it was created for other queries)",
			            (unsigned long long)hc[2]);
		}
		out->n_hits = hc[0];
		out->total_kmers = 0;
		const uint32_t *hn = (const uint32_t*)((const char*)sl->h_stage.p + 32);      // staged nkmer[]
		for(uint32_t i = 0; i < sl->b->n; ++i){ out->total_kmers += hn[i]; }
		const uint64_t cap = own ? sl->hit_cap : sl->ext_cap;
		if(!own || out->n_hits <= cap){ break; }
		// hit buffer too small (e.g. threshold truncated to 0: every column matches): grow, re-run
		if((rc = layout_result(sl, sl->b->n, out->n_hits, true))){ return rc; }
		HIP_TRY(hipMemsetAsync(sl->d_counters, 0, sizeof(uint64_t), sl->stream));   // hit counter only
		if((rc = enqueue_search_and_copy(sl))){ return rc; }
	}
	out->staged_hits = sl->staged_hits;
	out->launches = sl->launches;
	memcpy(out->kernel_name, sl->kernel_name, sizeof(out->kernel_name));
	if(timing_kmer){ HIP_TRY(hipEventElapsedTime(&out->kmer_ms, sl->ev[0], sl->ev[1])); }
	if(timing && sl->launches){ HIP_TRY(hipEventElapsedTime(&out->search_ms, sl->ev[2], sl->ev[3])); }
	return KWAGE_OK;
}

Slot *free_slot(kwage_ctx *ctx)
{
	for(int i = 0; i < 2; ++i){ if(!ctx->slot[i].busy){ return &ctx->slot[i]; } }
	return nullptr;
}

}  // namespace

namespace {

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

// The knobs' values at context creation: KWAGE_<NAME> for every name of TUNING_NAMES, plus the two historic spellings
// KWAGE_AND_CFG="vec,unroll,nt[,ldsKB[,block waves]]" and KWAGE_HIT_SORT=host.
void tuning_from_environment(Tuning *t)
{
	for(const TuningName &tn : TUNING_NAMES){
		std::string env = "KWAGE_";
		for(const char *c = tn.name; *c; ++c){ env += (char)toupper((unsigned char)*c); }
		const char *e = getenv(env.c_str());
		if(e && *e){ t->*(tn.field) = strtoll(e, nullptr, 10); }
	}
	if(const char *e = getenv("KWAGE_AND_CFG")){
		int v = 0, u = 0, n = 0, l = 0, w = 0;
		const int got = sscanf(e, "%d,%d,%d,%d,%d", &v, &u, &n, &l, &w);
		if(got >= 3){
			t->and_vec = v; t->and_unroll = u; t->and_nt = n;
			t->and_lds_kb = (got >= 4) ? l : 0;
			t->and_block_waves = (got >= 5) ? w : SEARCH_THREADS/WAVE;
		}
	}
	if(const char *e = getenv("KWAGE_HIT_SORT")){ t->hit_sort_host = !strcmp(e, "host") ? 1 : 0; }
}

}  // namespace

extern "C" int kwage_ctx_set_tuning(kwage_ctx *ctx, const char *name, int64_t value)
{
	if(!ctx || !name){ return fail(KWAGE_ERR_ARG, "kwage_ctx_set_tuning: NULL argument"); }
	for(int i = 0; i < 2; ++i){ if(ctx->slot[i].busy){ return fail(KWAGE_ERR_STATE, "kwage_ctx_set_tuning: a search is pending on this context"); } }
	for(const TuningName &tn : TUNING_NAMES){
		if(!strcmp(tn.name, name)){ ctx->tune.*(tn.field) = value; return KWAGE_OK; }
	}
	return fail(KWAGE_ERR_ARG, "kwage_ctx_set_tuning: no knob named '%s'", name);
}

extern "C" int kwage_ctx_get_tuning(kwage_ctx *ctx, const char *name, int64_t *value)
{
	if(!ctx || !name || !value){ return fail(KWAGE_ERR_ARG, "kwage_ctx_get_tuning: NULL argument"); }
	for(const TuningName &tn : TUNING_NAMES){
		if(!strcmp(tn.name, name)){ *value = ctx->tune.*(tn.field); return KWAGE_OK; }
	}
	return fail(KWAGE_ERR_ARG, "kwage_ctx_get_tuning: no knob named '%s'", name);
}

namespace kwage {
hipStream_t ctx_stream(kwage_ctx *ctx) { return ctx->stream; }
int ctx_device(kwage_ctx *ctx) { return ctx->device; }
}

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
extern "C" int kwage_device_count(void)
{
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess){ return 0; }
	return n;
}

extern "C" int kwage_init(int device, kwage_ctx **out)
{
	if(!out){ return fail(KWAGE_ERR_ARG, "kwage_init: out is NULL"); }
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if(e != hipSuccess || n == 0){
		return fail(KWAGE_ERR_DEVICE, "kwage_init: no HIP device available (%s); this engine has no CPU fallback",
		            e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
	}
	if(device < 0 || device >= n){ return fail(KWAGE_ERR_ARG, "kwage_init: device %d out of range [0,%d)", device, n); }
	HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if(strncmp(prop.gcnArchName, "gfx950", 6) != 0){
		return fail(KWAGE_ERR_DEVICE, "kwage_init: device %d is %s; this library is built for gfx950 (MI355X) only",
		            device, prop.gcnArchName);
	}
	kwage_ctx *ctx = new (std::nothrow) kwage_ctx();
	if(!ctx){ return fail(KWAGE_ERR_DEVICE, "out of host memory"); }
	ctx->device = device;
	ctx->ncu = prop.multiProcessorCount;
	tuning_from_environment(&ctx->tune);
	find_numa_cpus(device, &ctx->numa_node, &ctx->numa_cpus);
	for(int k = 0; k < 2; ++k){
		Slot *sl = &ctx->slot[k];
		// (a lowest-priority stream was tried so that a caller's small kernels get in between the gather
		// kernel's workgroups: no measurable difference on gfx950, tools/shard_pipe_probe.py)
		HIP_TRY(hipStreamCreateWithFlags(&sl->stream, hipStreamNonBlocking));
		for(int i = 0; i < 4; ++i){ HIP_TRY(hipEventCreate(&sl->ev[i])); }
		HIP_TRY(hipEventCreateWithFlags(&sl->search_done, hipEventDisableTiming));
	}
	ctx->stream = ctx->slot[0].stream;
	*out = ctx;
	return KWAGE_OK;
}

extern "C" void kwage_shutdown(kwage_ctx *ctx)
{
	if(!ctx){ return; }
	(void)hipSetDevice(ctx->device);
	release_mapping(ctx);
	if(ctx->map_done){ (void)hipEventDestroy(ctx->map_done); }
	for(hipEvent_t e : ctx->spare_events){ (void)hipEventDestroy(e); }
	for(int k = 0; k < 2; ++k){
		Slot *sl = &ctx->slot[k];
		if(sl->stream){ (void)hipStreamSynchronize(sl->stream); }
		sl->rows.release(); sl->tables.release(); sl->result.release();
		sl->partial.release(); sl->h_stage.release(); sl->sort_scratch.release();
		sl->walk_or.release(); sl->walk_done.release(); sl->cwalk_slab.release(); sl->cwalk_arrived.release();
		for(int i = 0; i < 4; ++i){ if(sl->ev[i]){ (void)hipEventDestroy(sl->ev[i]); } }
		if(sl->search_done){ (void)hipEventDestroy(sl->search_done); }
		if(sl->stream){ (void)hipStreamDestroy(sl->stream); }
	}
	ctx->kmers.release();
	ctx->result_pool->close();
	for(int i = 0; i < 2; ++i){
		ctx->load_pin[i].release(); ctx->load_dev[i].release();
		if(ctx->load_done[i]){ (void)hipEventDestroy(ctx->load_done[i]); }
		if(i == 1){ ctx->load_dev[2].release(); if(ctx->load_done[2]){ (void)hipEventDestroy(ctx->load_done[2]); } }
	}
	delete ctx;
}

extern "C" void kwage_set_load_progress(kwage_ctx *ctx, volatile uint64_t *bytes_passed)
{
	if(ctx){ ctx->load_progress = bytes_passed; }
}

extern "C" int kwage_mem_info(kwage_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes)
{
	if(!ctx){ return fail(KWAGE_ERR_ARG, "kwage_mem_info: ctx is NULL"); }
	int rc = set_device(ctx);
	if(rc){ return rc; }
	size_t f = 0, t = 0;
	HIP_TRY(hipMemGetInfo(&f, &t));
	if(free_bytes){ *free_bytes = f; }
	if(total_bytes){ *total_bytes = t; }
	return KWAGE_OK;
}

extern "C" int kwage_sync(kwage_ctx *ctx)
{
	if(!ctx){ return fail(KWAGE_ERR_ARG, "kwage_sync: ctx is NULL"); }
	int rc = set_device(ctx);
	if(rc){ return rc; }
	HIP_TRY(hipStreamSynchronize(ctx->slot[0].stream));
	HIP_TRY(hipStreamSynchronize(ctx->slot[1].stream));
	return KWAGE_OK;
}

// ------------------------------------------------------------------------------------------
// database group
// ------------------------------------------------------------------------------------------
namespace {

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
	hipError_t e = hipMalloc((void**)&g->d_bits, g->alloc_bytes);
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

bool load_env_flag(const char *name, bool fallback)
{
	const char *e = getenv(name);
	return e ? atoi(e) != 0 : fallback;
}

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
		const uint64_t items = (uint64_t)n*4*((max_bytes + (uint64_t)4*WAVE*ub - 1)/((uint64_t)4*WAVE*ub));     // (256-lane stretch, file, quarter) triples
		const uint32_t blocks = grid_for(items*WAVE, 256, 256*8);
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

extern "C" int kwage_group_params(const kwage_group *g, kwage_params *out)
{
	if(!g || !out){ return fail(KWAGE_ERR_ARG, "kwage_group_params: NULL argument"); }
	*out = g->params;
	return KWAGE_OK;
}

// ------------------------------------------------------------------------------------------
// query batch
// ------------------------------------------------------------------------------------------
extern "C" int kwage_batch_create(kwage_ctx *ctx, const char *seqs, const uint64_t *offsets, uint32_t n_queries,
                                  kwage_batch **out)
{
	if(!ctx || !offsets || !out){ return fail(KWAGE_ERR_ARG, "kwage_batch_create: NULL argument"); }
	*out = nullptr;
	for(uint32_t i = 0; i < n_queries; ++i){
		if(offsets[i + 1] < offsets[i]){ return fail(KWAGE_ERR_ARG, "kwage_batch_create: offsets must be non-decreasing"); }
		if(offsets[i + 1] - offsets[i] >= (1ull << 31)){ return fail(KWAGE_ERR_ARG, "kwage_batch_create: query %u is longer than 2^31-1 bases", i); }
	}
	if(n_queries > 0x7FFFFFFFu){ return fail(KWAGE_ERR_ARG, "kwage_batch_create: at most 2^31-1 queries per batch"); }
	const uint64_t total = offsets[n_queries] - offsets[0];
	if(total && !seqs){ return fail(KWAGE_ERR_ARG, "kwage_batch_create: seqs is NULL"); }
	int rc = set_device(ctx);
	if(rc){ return rc; }
	kwage_batch *b = new (std::nothrow) kwage_batch();
	if(!b){ return fail(KWAGE_ERR_DEVICE, "out of host memory"); }
	b->ctx = ctx;
	b->n = n_queries;
	b->total_len = total;
	b->h_seq_off.resize((size_t)n_queries + 1);
	for(uint32_t i = 0; i <= n_queries; ++i){ b->h_seq_off[i] = offsets[i] - offsets[0]; }
	hipError_t e = hipMalloc((void**)&b->d_seqs, std::max<uint64_t>(total, 16));
	if(e == hipSuccess){ e = hipMalloc((void**)&b->d_seq_off, ((size_t)n_queries + 1)*sizeof(uint64_t)); }
	if(e == hipSuccess && total){ e = hipMemcpyAsync(b->d_seqs, seqs + offsets[0], total, hipMemcpyHostToDevice, ctx->stream); }
	if(e == hipSuccess){ e = hipMemcpyAsync(b->d_seq_off, b->h_seq_off.data(), ((size_t)n_queries + 1)*sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream); }
	if(e == hipSuccess){ e = hipStreamSynchronize(ctx->stream); }
	if(e != hipSuccess){
		kwage_batch_destroy(b);
		return fail(KWAGE_ERR_DEVICE, "kwage_batch_create: %s", hipGetErrorString(e));
	}
	*out = b;
	return KWAGE_OK;
}

extern "C" void kwage_batch_destroy(kwage_batch *b)
{
	if(!b){ return; }
	(void)hipSetDevice(b->ctx->device);
	(void)hipStreamSynchronize(b->ctx->slot[0].stream);
	(void)hipStreamSynchronize(b->ctx->slot[1].stream);
	if(b->d_seqs){ (void)hipFree(b->d_seqs); }
	if(b->d_seq_off){ (void)hipFree(b->d_seq_off); }
	delete b;      // (its layouts free their device arrays)
}

extern "C" uint32_t kwage_batch_num_queries(const kwage_batch *b) { return b ? b->n : 0; }

// ------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------
namespace {

// f(t, lo, hi) for T contiguous chunks of [0, n), chunk 0 on the calling thread.
template <typename F>
void parallel_chunks(unsigned T, size_t n, F f)
{
	if(T <= 1){ f(0u, (size_t)0, n); return; }
	std::vector<std::thread> pool;
	for(unsigned t = 1; t < T; ++t){ pool.emplace_back(f, t, n*t/T, n*(t + 1)/T); }
	f(0u, (size_t)0, n/T);
	for(std::thread &th : pool){ th.join(); }
}

// Order hits by (query, column) on the host -- the lists of at most SPEC_HITS records that come back with the
// counters (longer ones are sorted on the device, hit_sort.hip), and the merged lists of a sharded search on rank 0
// (kwage_sort_hits): std::sort for short lists, otherwise an LSD radix sort on the 64-bit key (11-bit digits; digits on
// which all keys agree are skipped).  From a million records on the passes run on up to 8 threads (per-thread
// histograms, one prefix over digits x threads, disjoint scatter ranges): eight C3 shares return 9.6 M records per
// step to rank 0, which one thread orders in about the time the step's kernel takes.
void sort_hits(kwage_hit *hits, size_t n)
{
	if(n < 256){
		std::sort(hits, hits + n, [](const kwage_hit &x, const kwage_hit &y){
			return (x.query != y.query) ? (x.query < y.query) : (x.column < y.column);
		});
		return;
	}
	struct Rec { uint64_t key; uint32_t val; };
	const unsigned T = (n >= (1u << 20)) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u;
	std::unique_ptr<Rec[]> a(new Rec[n]), b(new Rec[n]);          // (not value-initialised: every record is written below)
	std::vector<uint64_t> ors(T, 0), ands(T, ~0ull);
	parallel_chunks(T, n, [&](unsigned t, size_t lo, size_t hi){
		uint64_t o = 0, d = ~0ull;
		for(size_t i = lo; i < hi; ++i){
			const uint64_t key = ((uint64_t)hits[i].query << 32) | hits[i].column;
			a[i].key = key;
			a[i].val = hits[i].num_match;
			o |= key; d &= key;
		}
		ors[t] = o; ands[t] = d;
	});
	uint64_t all_or = 0, all_and = ~0ull;
	for(unsigned t = 0; t < T; ++t){ all_or |= ors[t]; all_and &= ands[t]; }
	const uint64_t varying = all_or ^ all_and;
	Rec *src = a.get(), *dst = b.get();
	std::vector<size_t> hist((size_t)T*2048);
	for(int shift = 0; shift < 64; shift += 11){
		const uint64_t mask = 0x7FFull << shift;
		if((varying & mask) == 0){ continue; }
		parallel_chunks(T, n, [&](unsigned t, size_t lo, size_t hi){
			size_t *h = hist.data() + (size_t)t*2048;
			memset(h, 0, 2048*sizeof(size_t));
			for(size_t i = lo; i < hi; ++i){ ++h[(src[i].key >> shift) & 0x7FF]; }
		});
		size_t at = 0;                      // digit-major, then thread: thread t's records with digit d follow those of threads < t
		for(int d = 0; d < 2048; ++d){
			for(unsigned t = 0; t < T; ++t){ const size_t c = hist[(size_t)t*2048 + d]; hist[(size_t)t*2048 + d] = at; at += c; }
		}
		parallel_chunks(T, n, [&](unsigned t, size_t lo, size_t hi){
			size_t *h = hist.data() + (size_t)t*2048;
			for(size_t i = lo; i < hi; ++i){ dst[h[(src[i].key >> shift) & 0x7FF]++] = src[i]; }
		});
		std::swap(src, dst);
	}
	parallel_chunks(T, n, [&](unsigned, size_t lo, size_t hi){
		for(size_t i = lo; i < hi; ++i){
			hits[i].query = (uint32_t)(src[i].key >> 32);
			hits[i].column = (uint32_t)src[i].key;
			hits[i].num_match = src[i].val;
		}
	});
}

struct ResultStorage {
	kwage_result pub;
	std::unique_ptr<kwage_hit[]> hits;            // short lists
	std::shared_ptr<PinnedPool> pool;             // long lists: a pinned block of the context's pool
	PinBuf pinned;
	std::vector<uint32_t> nkmer, qthr;
	char kernel[64];
	~ResultStorage() { if(pool){ pool->release(pinned); } }
};

// Build the host result of a collected search from the slot's staging buffer.
int build_result(Slot *sl, kwage_group *g, kwage_batch *b, const SearchOutcome &so, kwage_result **out)
{
	ResultStorage *rs = new (std::nothrow) ResultStorage();
	if(!rs){ return fail(KWAGE_ERR_DEVICE, "out of host memory"); }
	rs->nkmer.resize(b->n);
	rs->qthr.resize(b->n);
	const uint64_t nq_bytes = (uint64_t)b->n*sizeof(uint32_t);
	const char *hs = (const char*)sl->h_stage.p;        // host image of the result block's head
	if(b->n){
		memcpy(rs->nkmer.data(), hs + 32, nq_bytes);
		memcpy(rs->qthr.data(), hs + 32 + nq_bytes, nq_bytes);
	}
	const uint64_t have = std::min(so.n_hits, so.staged_hits);
	kwage_hit *hits = nullptr;
	if(so.n_hits <= have){
		rs->hits.reset(new (std::nothrow) kwage_hit[std::max<uint64_t>(so.n_hits, 1)]);      // not zero-filled: every record is written below
		if(!rs->hits){ delete rs; return fail(KWAGE_ERR_DEVICE, "out of host memory (%llu hits)", (unsigned long long)so.n_hits); }
		hits = rs->hits.get();
		if(have){ memcpy(hits, hs + sl->head_bytes, have*sizeof(kwage_hit)); }
		// deterministic order; the reference's own order among ties is unspecified (sort.h:22-27)
		sort_hits(hits, so.n_hits);
	} else {
		// A long list is sorted where it lies (hit_sort.hip) and then crosses PCIe ONCE, straight into the result array --
		// a pinned block of the context's pool (PinnedPool above): no staging hop, no host copy, no page faults.
		rs->pool = g->ctx->result_pool;
		int rc2 = rs->pool->acquire(so.n_hits*sizeof(kwage_hit), &rs->pinned);
		if(rc2){ delete rs; return rc2; }
		hits = (kwage_hit*)rs->pinned.p;
		uint64_t scratch = 0;
		const bool host_sort = g->ctx->tune.hit_sort_host != 0;      // the sort of round 1, kept for A/B runs and as the fallback
		const uint64_t column_span = std::min<uint64_t>((uint64_t)sl->col_base + g->stride*8, 1ull << 32);      // no hit carries a column beyond the row (files are padded apart: more than num_columns)
		bool on_device = !host_sort
		                 && hit_sort_scratch_bytes(so.n_hits, b->n, column_span, &scratch) == KWAGE_OK
		                 && sl->sort_scratch.reserve(scratch) == KWAGE_OK
		                 && sort_hits_on_device(sl->stream, sl->d_hits, so.n_hits, b->n, column_span, sl->sort_scratch.p, sl->sort_scratch.cap) == KWAGE_OK;
		if(!on_device){
			(void)hipGetLastError();
			if(!host_sort){      // never silently: the list is still ordered, by the host, and that is slower
				fprintf(stderr, "[kwage_amd] no room for the device hit sort's buffers (%llu bytes) beside the database: %llu hits ordered by the host\n",
				        (unsigned long long)scratch, (unsigned long long)so.n_hits);
			}
		}
		// (pieces only so that a knob can shrink them in tests: one copy is what the link likes)
		const uint64_t piece_kb = (uint64_t)std::max<int64_t>(g->ctx->tune.hit_copy_piece_kb, 0);
		const uint64_t piece = piece_kb ? std::max<uint64_t>(1, (piece_kb << 10)/sizeof(kwage_hit)) : so.n_hits;
		hipError_t e = hipSuccess;
		for(uint64_t at = 0; at < so.n_hits && e == hipSuccess; at += piece){
			const uint64_t m = std::min(piece, so.n_hits - at);
			e = hipMemcpyAsync(hits + at, sl->d_hits + at, m*sizeof(kwage_hit), hipMemcpyDeviceToHost, sl->stream);
		}
		if(e == hipSuccess){ e = hipStreamSynchronize(sl->stream); }
		if(e != hipSuccess){
			(void)hipStreamSynchronize(sl->stream);
			delete rs;
			return fail(KWAGE_ERR_DEVICE, "kwage_search: copying results failed: %s", hipGetErrorString(e));
		}
		if(!on_device){ sort_hits(hits, so.n_hits); }
		if(sl->sort_scratch.cap > SORT_SCRATCH_KEEP){ sl->sort_scratch.release(); }      // a rare giant list: give the memory back
	}

	kwage_result &r = rs->pub;
	r.n_hits = so.n_hits;
	r.hits = hits;
	r.n_queries = b->n;
	r.num_query_kmer = rs->nkmer.data();
	r.query_threshold = rs->qthr.data();
	r.total_kmers = so.total_kmers;
	r.bit_tests = so.total_kmers*g->params.num_hash*g->num_columns;
	r.algorithmic_bytes = so.total_kmers*g->params.num_hash*((g->num_columns + 7)/8);
	r.kmer_kernel_ms = so.kmer_ms;
	r.search_kernel_ms = so.search_ms;
	r.search_kernel_launches = so.launches;
	memcpy(rs->kernel, so.kernel_name, sizeof(rs->kernel));
	r.search_kernel = rs->kernel;
	*out = &rs->pub;
	return KWAGE_OK;
}

}  // namespace

extern "C" void kwage_sort_hits(kwage_hit *hits, uint64_t n)
{
	if(!hits || n < 2){ return; }
	try{ sort_hits(hits, (size_t)n); }
	catch(const std::bad_alloc &){      // no room for the radix sort's two copies of the list: order it in place
		std::sort(hits, hits + n, [](const kwage_hit &x, const kwage_hit &y){
			return (x.query != y.query) ? (x.query < y.query) : (x.column < y.column);
		});
	}
}

struct kwage_pending { Slot *sl; };

extern "C" int kwage_search_submit(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags, kwage_pending **out)
{
	if(!g || !b || !out){ return fail(KWAGE_ERR_ARG, "kwage_search_submit: NULL argument"); }
	*out = nullptr;
	Slot *sl = free_slot(g->ctx);
	if(!sl){ return fail(KWAGE_ERR_STATE, "kwage_search_submit: two searches are already pending on this context"); }
	int rc = submit_search(sl, g, b, threshold, flags, nullptr, 0);
	if(rc){ return rc; }
	kwage_pending *p = new (std::nothrow) kwage_pending();
	if(!p){ SearchOutcome so; (void)collect_search(sl, &so); return fail(KWAGE_ERR_DEVICE, "out of host memory"); }
	p->sl = sl;
	*out = p;
	return KWAGE_OK;
}

extern "C" int kwage_search_collect(kwage_pending *p, kwage_result **out)
{
	if(!p || !out){ return fail(KWAGE_ERR_ARG, "kwage_search_collect: NULL argument"); }
	*out = nullptr;
	Slot *sl = p->sl;
	delete p;
	kwage_group *g = sl->g;
	kwage_batch *b = sl->b;
	SearchOutcome so;
	int rc = collect_search(sl, &so);
	if(rc){ return rc; }
	return build_result(sl, g, b, so, out);
}

extern "C" int kwage_search(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags, kwage_result **out)
{
	if(!g || !b || !out){ return fail(KWAGE_ERR_ARG, "kwage_search: NULL argument"); }
	*out = nullptr;
	static const bool prof = getenv("KWAGE_PROFILE_HOST") != nullptr;
	const auto tp0 = std::chrono::steady_clock::now();
	Slot *sl = free_slot(g->ctx);
	if(!sl){ return fail(KWAGE_ERR_STATE, "kwage_search: two searches are already pending on this context"); }
	int rc = submit_search(sl, g, b, threshold, flags, nullptr, 0);
	if(rc){ return rc; }
	SearchOutcome so;
	if((rc = collect_search(sl, &so))){ return rc; }
	const auto tp1 = std::chrono::steady_clock::now();
	rc = build_result(sl, g, b, so, out);
	if(prof && !rc){
		const auto tp2 = std::chrono::steady_clock::now();
		fprintf(stderr, "[kwage_search] device pipeline + sync %.1f us, host result assembly + sort %.1f us (%llu hits)\n",
		        std::chrono::duration<double, std::micro>(tp1 - tp0).count(),
		        std::chrono::duration<double, std::micro>(tp2 - tp1).count(), (unsigned long long)so.n_hits);
	}
	return rc;
}

extern "C" void kwage_result_free(kwage_result *r)
{
	if(!r){ return; }
	delete reinterpret_cast<ResultStorage*>(r);     // pub is the first member
}

extern "C" int kwage_search_device_submit(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags,
                                          void *hits_dev, uint64_t capacity, void *count_dev, kwage_pending **out)
{
	if(!g || !b || !out || (capacity && !hits_dev)){ return fail(KWAGE_ERR_ARG, "kwage_search_device_submit: NULL argument"); }
	*out = nullptr;
	Slot *sl = free_slot(g->ctx);
	if(!sl){ return fail(KWAGE_ERR_STATE, "kwage_search_device_submit: two searches are already pending on this context"); }
	static kwage_hit dummy;          // non-null marker for "caller-owned buffer" when capacity is 0
	int rc = submit_search(sl, g, b, threshold, flags, capacity ? (kwage_hit*)hits_dev : &dummy, capacity, (uint64_t*)count_dev);
	if(rc){ return rc; }
	kwage_pending *p = new (std::nothrow) kwage_pending();
	if(!p){ SearchOutcome so; (void)collect_search(sl, &so); return fail(KWAGE_ERR_DEVICE, "out of host memory"); }
	p->sl = sl;
	*out = p;
	return KWAGE_OK;
}

extern "C" int kwage_search_device_append_submit(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags,
                                                 void *hits_dev, uint64_t capacity, void *count_dev, uint32_t column_base,
                                                 int reset_count, kwage_pending **out)
{
	if(!g || !b || !out || !count_dev || (capacity && !hits_dev)){ return fail(KWAGE_ERR_ARG, "kwage_search_device_append_submit: NULL argument"); }
	*out = nullptr;
	if((uint64_t)column_base + g->stride*8 > 0x100000000ull){
		return fail(KWAGE_ERR_ARG, "kwage_search_device_append_submit: column base %u + the group's column span exceeds 32 bits", column_base);
	}
	Slot *sl = free_slot(g->ctx);
	if(!sl){ return fail(KWAGE_ERR_STATE, "kwage_search_device_append_submit: two searches are already pending on this context"); }
	static kwage_hit dummy;          // non-null marker for "caller-owned buffer" when capacity is 0
	int rc = submit_search(sl, g, b, threshold, flags, capacity ? (kwage_hit*)hits_dev : &dummy, capacity, (uint64_t*)count_dev,
	                       true, reset_count != 0, column_base);
	if(rc){ return rc; }
	kwage_pending *p = new (std::nothrow) kwage_pending();
	if(!p){ SearchOutcome so; (void)collect_search(sl, &so); return fail(KWAGE_ERR_DEVICE, "out of host memory"); }
	p->sl = sl;
	*out = p;
	return KWAGE_OK;
}

extern "C" int kwage_search_device_collect(kwage_pending *p, uint64_t *n_hits, void *num_query_kmer_dev, float *search_kernel_ms)
{
	if(!p || !n_hits){ return fail(KWAGE_ERR_ARG, "kwage_search_device_collect: NULL argument"); }
	Slot *sl = p->sl;
	delete p;
	kwage_batch *b = sl->b;
	SearchOutcome so;
	int rc = collect_search(sl, &so);
	if(rc){ return rc; }
	*n_hits = so.n_hits;
	if(search_kernel_ms){ *search_kernel_ms = so.search_ms; }
	if(num_query_kmer_dev && b->n){
		HIP_TRY(hipMemcpyAsync(num_query_kmer_dev, sl->d_nkmer, (size_t)b->n*sizeof(uint32_t), hipMemcpyDeviceToDevice, sl->stream));
		HIP_TRY(hipStreamSynchronize(sl->stream));
	}
	return KWAGE_OK;
}

extern "C" int kwage_search_device(kwage_group *g, kwage_batch *b, float threshold, uint32_t flags,
                                   void *hits_dev, uint64_t capacity, uint64_t *n_hits, void *num_query_kmer_dev)
{
	if(!g || !b || !n_hits || (capacity && !hits_dev)){ return fail(KWAGE_ERR_ARG, "kwage_search_device: NULL argument"); }
	Slot *sl = free_slot(g->ctx);
	if(!sl){ return fail(KWAGE_ERR_STATE, "kwage_search_device: two searches are already pending on this context"); }
	// a zero-capacity call (size query) still needs a non-null marker for "caller-owned buffer"
	static kwage_hit dummy;
	int rc = submit_search(sl, g, b, threshold, flags, capacity ? (kwage_hit*)hits_dev : &dummy, capacity);
	if(rc){ return rc; }
	SearchOutcome so;
	if((rc = collect_search(sl, &so))){ return rc; }
	*n_hits = so.n_hits;
	if(num_query_kmer_dev && b->n){
		HIP_TRY(hipMemcpyAsync(num_query_kmer_dev, sl->d_nkmer, (size_t)b->n*sizeof(uint32_t), hipMemcpyDeviceToDevice, sl->stream));
		HIP_TRY(hipStreamSynchronize(sl->stream));
	}
	return KWAGE_OK;
}

extern "C" int kwage_hash_batch(kwage_ctx *ctx, const kwage_params *params, kwage_batch *b,
                                uint64_t *kmer_offsets, uint32_t *num_query_kmer, uint64_t *kmers, uint32_t *rows)
{
	if(!ctx || !params || !b || !kmer_offsets || !num_query_kmer){ return fail(KWAGE_ERR_ARG, "kwage_hash_batch: NULL argument"); }
	int rc = check_params(params);
	if(rc){ return rc; }
	if(b->ctx != ctx){ return fail(KWAGE_ERR_ARG, "kwage_hash_batch: batch belongs to another context"); }
	if((rc = set_device(ctx))){ return rc; }
	Slot *sl = &ctx->slot[0];
	if(sl->busy){ return fail(KWAGE_ERR_STATE, "kwage_hash_batch: a search is pending on this context"); }
	const KmerLayout *L = nullptr;
	if((rc = batch_prepare(b, params->kmer_len, &L))){ return rc; }
	const uint64_t np = std::max<uint64_t>(L->total_pos, 1);
	if((rc = sl->rows.reserve(np*params->num_hash*sizeof(uint32_t)))){ return rc; }
	if((rc = ctx->kmers.reserve(np*sizeof(uint64_t)))){ return rc; }
	if((rc = launch_kmer_stage(sl, *params, b, L, 1.0f, (uint32_t*)sl->rows.p, (uint64_t*)ctx->kmers.p))){ return rc; }
	memcpy(kmer_offsets, L->h_pos_off.data(), ((size_t)b->n + 1)*sizeof(uint64_t));
	if(b->n){ HIP_TRY(hipMemcpyAsync(num_query_kmer, sl->d_nkmer, (size_t)b->n*sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
	if(kmers && L->total_pos){ HIP_TRY(hipMemcpyAsync(kmers, ctx->kmers.p, L->total_pos*sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream)); }
	if(rows && L->total_pos){ HIP_TRY(hipMemcpyAsync(rows, sl->rows.p, L->total_pos*params->num_hash*sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return KWAGE_OK;
}

// ------------------------------------------------------------------------------------------
// Bloom filter construction from sequences (exact k-mer set)
// ------------------------------------------------------------------------------------------
namespace {

// Run the k-mer stage over `b` with ONE shared distinct set; optionally set Bloom bits.
int run_shared_kmer_pass(kwage_ctx *ctx, const kwage_params &p, kwage_batch *b, uint32_t *d_bloom_bits, uint64_t *distinct)
{
	Slot *sl = &ctx->slot[0];
	if(sl->busy){ return fail(KWAGE_ERR_STATE, "a search is pending on this context"); }
	const KmerLayout *L = nullptr;
	int rc = batch_prepare(b, p.kmer_len, &L);
	if(rc){ return rc; }
	if((rc = layout_result(sl, b->n, 0, false))){ return rc; }
	uint32_t lg = 10;
	while((1ull << lg) < 2*std::max<uint64_t>(L->total_pos, 1)){ ++lg; }
	// (the shared_table_log2 knob raises the table size: tests exercise the >= 2^32-slot arithmetic on small inputs)
	if(ctx->tune.shared_table_log2 > 0){ lg = std::max<uint32_t>(lg, (uint32_t)ctx->tune.shared_table_log2); }
	if(lg > 36){ return fail(KWAGE_ERR_ARG, "too many k-mer positions for one sample"); }
	if((rc = sl->tables.reserve((1ull << lg)*sizeof(uint64_t)))){ return rc; }
	HIP_TRY(hipMemsetAsync(sl->tables.p, 0xFF, (1ull << lg)*sizeof(uint64_t), ctx->stream));
	HIP_TRY(hipMemsetAsync(sl->d_counters, 0, 4*sizeof(uint64_t), ctx->stream));
	if(b->n){
		KmerArgs a;
		a.seqs = b->d_seqs; a.seq_off = b->d_seq_off; a.pos_off = L->d_pos_off; a.tab_off = L->d_tab_off;
		a.g_tables = (unsigned long long*)sl->tables.p;
		a.k = p.kmer_len; a.num_hash = p.num_hash;
		a.row_mask = (p.log_2_filter_len >= 32) ? 0xFFFFFFFFu : ((1u << p.log_2_filter_len) - 1u);
		a.threshold = 1.0f; a.complete_match = 1;
		a.rows = nullptr; a.kmers_out = nullptr;
		a.nkmer = sl->d_nkmer; a.qthr = sl->d_qthr;
		a.total_kmers = (unsigned long long*)sl->d_counters + 1;
		a.shared_lg = lg;
		a.bloom_bits = d_bloom_bits;
		a.chunk_q = nullptr; a.chunk_t0 = nullptr;       // one workgroup per sequence
		a.lds_slots = 0;                 // the shared global table is used for every sequence
		hipLaunchKernelGGL(kmer_kernel, dim3(b->n), dim3(KM_THREADS), 0, ctx->stream, a);
		HIP_TRY(hipGetLastError());
	}
	uint64_t h[2] = {0, 0};
	HIP_TRY(hipMemcpyAsync(h, sl->d_counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	if(distinct){ *distinct = h[1]; }
	return KWAGE_OK;
}

}  // namespace

extern "C" int kwage_count_distinct_kmers(kwage_ctx *ctx, kwage_batch *b, uint32_t kmer_len, uint64_t *count)
{
	if(!ctx || !b || !count){ return fail(KWAGE_ERR_ARG, "kwage_count_distinct_kmers: NULL argument"); }
	if(b->ctx != ctx){ return fail(KWAGE_ERR_ARG, "kwage_count_distinct_kmers: batch belongs to another context"); }
	kwage_params p = {kmer_len, 1, 0, KWAGE_HASH_MURMUR32};
	int rc = check_params(&p);
	if(rc){ return rc; }
	if((rc = set_device(ctx))){ return rc; }
	return run_shared_kmer_pass(ctx, p, b, nullptr, count);
}

extern "C" int kwage_bloom_bits_from_batch(kwage_ctx *ctx, const kwage_params *params, kwage_batch *b,
                                           void *bits_out, uint64_t *distinct)
{
	if(!ctx || !params || !b || !bits_out){ return fail(KWAGE_ERR_ARG, "kwage_bloom_bits_from_batch: NULL argument"); }
	if(b->ctx != ctx){ return fail(KWAGE_ERR_ARG, "kwage_bloom_bits_from_batch: batch belongs to another context"); }
	int rc = check_params(params);
	if(rc){ return rc; }
	if((rc = set_device(ctx))){ return rc; }
	const uint64_t nbytes = ((1ull << params->log_2_filter_len) + 7)/8;
	const uint64_t alloc = (nbytes + 3)/4*4;
	DevBuf bits;
	if((rc = bits.reserve(alloc))){ return rc; }
	hipError_t e = hipMemsetAsync(bits.p, 0, alloc, ctx->stream);
	if(e != hipSuccess){ bits.release(); return fail(KWAGE_ERR_DEVICE, "%s", hipGetErrorString(e)); }
	rc = run_shared_kmer_pass(ctx, *params, b, (uint32_t*)bits.p, distinct);
	if(!rc){
		e = hipMemcpy(bits_out, bits.p, nbytes, hipMemcpyDeviceToHost);
		if(e != hipSuccess){ rc = fail(KWAGE_ERR_DEVICE, "%s", hipGetErrorString(e)); }
	}
	bits.release();
	return rc;
}

extern "C" int kwage_stream_read_gbps(kwage_group *g, uint64_t bytes, uint32_t iters, double *gbps)
{
	if(!g || !gbps || iters == 0){ return fail(KWAGE_ERR_ARG, "kwage_stream_read_gbps: bad argument"); }
	kwage_ctx *ctx = g->ctx;
	int rc = set_device(ctx);
	if(rc){ return rc; }
	bytes = std::min(bytes, g->alloc_bytes)/16*16;
	if(bytes == 0){ return fail(KWAGE_ERR_ARG, "kwage_stream_read_gbps: nothing to read"); }
	Slot *sl = &ctx->slot[0];
	if(sl->busy){ return fail(KWAGE_ERR_STATE, "kwage_stream_read_gbps: a search is pending on this context"); }
	if((rc = layout_result(sl, 0, 0, false))){ return rc; }
	uint32_t *sink = (uint32_t*)(sl->d_counters + 3);
	const uint64_t n16 = bytes/16;
	// every wave walks a contiguous region of whole 8 KiB steps: up to 8192 waves, fewer for a small matrix
	const uint64_t step16 = (uint64_t)WAVE*8;
	if(n16 < step16){ return fail(KWAGE_ERR_ARG, "kwage_stream_read_gbps: the matrix is smaller than one 8 KiB step"); }
	const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(256*8, n16/step16/4));
	const uint64_t nwaves = (uint64_t)blocks*4;
	bytes = nwaves*((n16/nwaves)/step16*step16)*16;          // what the kernel reads (a remainder below one step per wave is left out)
	hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const u32x4*)g->d_bits, n16, sink);   // warm-up
	HIP_TRY(hipEventRecord(sl->ev[0], ctx->stream));
	for(uint32_t i = 0; i < iters; ++i){
		hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const u32x4*)g->d_bits, n16, sink);
	}
	HIP_TRY(hipEventRecord(sl->ev[1], ctx->stream));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, sl->ev[0], sl->ev[1]));
	*gbps = (double)bytes*iters/((double)ms*1e-3)/1e9;
	return KWAGE_OK;
}
