// kwage_amd/csrc/engine_state.hpp -- what the translation units of the device library share: the context, the
// database group, the query batch and the buffers they own.  Private to kwage_amd/csrc (the C ABI exposes these types
// as opaque pointers only).
//
//   engine.hip   contexts, tuning knobs, query batches, the search pipeline (k-mer stage, gather kernels, hit lists)
//   loader.hip   database groups: allocation, the loaders (.db files raw and compressed, sparse groups), synthetic columns
#ifndef KWAGE_AMD_ENGINE_STATE_HPP
#define KWAGE_AMD_ENGINE_STATE_HPP

#include <hip/hip_runtime.h>

#include <algorithm>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

#include "internal.h"

#define HIP_TRY(expr)                                                                          \
	do {                                                                                       \
		hipError_t _e = (expr);                                                                \
		if(_e != hipSuccess){                                                                  \
			return fail(KWAGE_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
			            __FILE__, __LINE__);                                                   \
		}                                                                                      \
	} while(0)

struct kwage_group;
struct kwage_batch;
struct kwage_ctx;

namespace kwage {

struct KmerLayout;

// Device memory that is only PARKED -- the blocks of destroyed query batches a context keeps for the next ones (DevPool
// below) -- must never be the reason an allocation fails: every pool is listed here, and an allocation that runs out of
// memory empties them all and tries once more.
struct DevPool;
struct DevPoolRegistry {
	std::mutex mu;
	std::vector<DevPool*> pools;
	static DevPoolRegistry &get() { static DevPoolRegistry r; return r; }
	void drain_all();
};

// hipMalloc; out of memory: the parked blocks of every context's pool are released, then once more.
inline hipError_t device_malloc(void **out, uint64_t bytes)
{
	hipError_t e = hipMalloc(out, bytes);
	if(e == hipErrorOutOfMemory){
		(void)hipGetLastError();
		DevPoolRegistry::get().drain_all();
		e = hipMalloc(out, bytes);
	}
	return e;
}

// A device buffer that only ever grows (scratch reused across searches).
struct DevBuf {
	void *p = nullptr;
	uint64_t cap = 0;
	int reserve(uint64_t bytes)
	{
		if(bytes <= cap){ return KWAGE_OK; }
		if(p){ (void)hipFree(p); p = nullptr; cap = 0; }
		const uint64_t want = std::max<uint64_t>(bytes + bytes/4, 4096);
		HIP_TRY(device_malloc(&p, want));
		cap = want;
		return KWAGE_OK;
	}
	void release()
	{
		if(p){ (void)hipFree(p); }
		p = nullptr; cap = 0;
	}
};

// Device blocks of destroyed query batches, kept for the next ones.  hipFree waits for the whole device: a host that
// streams batches through a context (the command lines: one batch created and one destroyed per step) would drain its
// software pipeline at every batch.  Blocks go back here instead and are reused (best fit within 2x); what exceeds
// KEEP bytes is really freed.
struct DevPool {
	struct Block { void *p; uint64_t cap; };
	std::vector<Block> free_blocks;        // oldest first
	uint64_t held = 0;
	static constexpr uint64_t KEEP = 1ull << 30;
	DevPool() { std::lock_guard<std::mutex> lk(DevPoolRegistry::get().mu); DevPoolRegistry::get().pools.push_back(this); }
	~DevPool()
	{
		std::lock_guard<std::mutex> lk(DevPoolRegistry::get().mu);
		std::vector<DevPool*> &v = DevPoolRegistry::get().pools;
		v.erase(std::remove(v.begin(), v.end(), this), v.end());
	}
	DevPool(const DevPool&) = delete;
	DevPool &operator=(const DevPool&) = delete;
	hipError_t take(uint64_t bytes, void **out, uint64_t *cap)
	{
		bytes = std::max<uint64_t>(bytes, 256);
		size_t best = free_blocks.size();
		for(size_t i = 0; i < free_blocks.size(); ++i){
			if(free_blocks[i].cap >= bytes && free_blocks[i].cap <= 2*bytes + (1u << 20) && (best == free_blocks.size() || free_blocks[i].cap < free_blocks[best].cap)){ best = i; }
		}
		if(best != free_blocks.size()){
			*out = free_blocks[best].p; *cap = free_blocks[best].cap;
			held -= free_blocks[best].cap;
			free_blocks.erase(free_blocks.begin() + (long)best);
			return hipSuccess;
		}
		*cap = (bytes + 65535)/65536*65536;
		hipError_t e = hipMalloc(out, *cap);
		if(e == hipErrorOutOfMemory){          // what is parked here (and in the other contexts' pools) goes first
			(void)hipGetLastError();
			close();
			DevPoolRegistry::get().drain_all();
			e = hipMalloc(out, *cap);
		}
		return e;
	}
	// (over KEEP the OLDEST parked blocks go, not the incoming one: what a streaming host hands back is what it asks for next)
	void give(void *p, uint64_t cap)
	{
		if(!p){ return; }
		if(cap > KEEP){ (void)hipFree(p); return; }
		while(held + cap > KEEP && !free_blocks.empty()){
			(void)hipFree(free_blocks.front().p);
			held -= free_blocks.front().cap;
			free_blocks.erase(free_blocks.begin());
		}
		free_blocks.push_back(Block{p, cap});
		held += cap;
	}
	void close()
	{
		for(Block &b : free_blocks){ (void)hipFree(b.p); }
		free_blocks.clear();
		held = 0;
	}
};

inline void DevPoolRegistry::drain_all()
{
	// (a context is used by one host thread at a time; the registry only guards the LIST: pools are drained by the thread
	// whose allocation failed, which is the thread that owns the pool in every host this library has)
	std::lock_guard<std::mutex> lk(mu);
	for(DevPool *p : pools){ p->close(); }
}

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


// Everything one in-flight search owns.  A context has two slots so that a second search can be
// submitted (and its k-mer stage run) while the first one's results are still being collected.
struct Slot {
	hipStream_t stream = nullptr;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	// the gather stage of BOTH slots runs on the context's one gather stream (kwage_ctx::gather_stream), so that gather
	// kernels follow each other on one hardware queue without a cross-queue dependency between them (~25 us each on
	// gfx950); the slot's own stream carries the k-mer stage before it and the copy-back after it:
	hipEvent_t kmer_done = nullptr;     // recorded on `stream` behind the k-mer stage: the gather stream waits for it
	hipEvent_t gather_done = nullptr;   // recorded on the gather stream behind the slot's gather kernel(s): `stream` waits for it
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
	DevBuf sort_scratch;   // the ordered copy of a long hit list + the block sums of its run table (lists beyond SPEC_HITS only)
	// the run table of the search's own hit list (kernels.hpp SearchArgs::runs): 8 bytes per reservation of hit slots,
	// runs_per_query entries per query, zeroed before every gather stage; n_runs == 0: not kept (caller-owned lists)
	DevBuf runs;
	uint64_t n_runs = 0;
	uint32_t runs_per_query = 0;
	// the submission occupying the slot
	bool busy = false;
	kwage_group *g = nullptr;
	kwage_batch *b = nullptr;
	const KmerLayout *lay = nullptr;     // the batch's layout for the group's k-mer length
	float threshold = 1.0f;
	uint32_t flags = 0;
	uint32_t launches = 0;
	char kernel_name[64] = "";          // the gather kernel launch_search_stage picked, with its template shape
	// and_walk_kernel's meeting place for (query, tile) pairs cut by a wave-share boundary: all zero between searches
	DevBuf walk_or, walk_done;
	// and_band_walk_kernel's: every query's rows bucketed by band of the matrix, the buckets' offsets per query, and
	// the queries' meeting slots (masks + flags: zero between searches)
	DevBuf band_rows, band_loc, band_or, band_state;
	// count_walk_kernel's: partial counters of cut pairs (overwritten before they are read) and the arrival counters of their trees (zero between searches)
	DevBuf cwalk_slab, cwalk_arrived;
	// early exit, screen + refine (kernels.hpp and_screen_kernel): the three counters and the lists of the tiles handed over
	DevBuf ref_counters, ref_clusters, ref_masks, ref_units, ref_slab;
	DevBuf trunc_dev;              // the truncated count walk's per-query arrays: [slot_off: (n + 1) x u64 | kcut: n x u32]
	PinBuf trunc_host;             // their host image (pinned: the copy is queued, the slot keeps it until the search is collected)
	uint32_t ref_base[3] = {0, 0, 0}, ref_cap[3] = {0, 0, 0};      // of the last such search (kwage_ctx_refine_stats)
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
	int64_t walk = 4;               // KWAGE_WALK: 0 = never the walk form (always the tiled kernel)
	int64_t walk_min_rows = -1;     // KWAGE_WALK_MIN_ROWS: batches with fewer rows use the tiled kernel (-1: 64 rows per wave of the chip)
	int64_t walk_max_kib = 16;      // KWAGE_WALK_MAX_KIB: widest row the walk form takes, in KiB-steps (wider rows: the tiled kernel's wide shape, which stays 0.5-3 % ahead on C3 / C4 and C3 split in two; the walk form handles them column tile after column tile when the knob is raised)
	int64_t walk_tile_kib = 16;     // KWAGE_WALK_TILE_KIB: widest column tile of the walk form in KiB-steps (rows wider than it are walked tile after tile)
	int64_t walk_min_kib = 2;       // KWAGE_WALK_MIN_KIB: narrowest row (in KiB-steps) the walk form takes (round 4: 2 -- rows of 1-2 KiB no
	                                //   longer need the tiled kernel's segments + combine pass: 0.263 vs 0.284 ms at C2's columns split 8 ways)
	int64_t walk_waves = 0;         // KWAGE_WALK_WAVES: exactly this many waves (tests: shares of every size); 0 = from the CU count
	int64_t walk_one_wg_per_cu = 1; // KWAGE_WALK_ONE_WG_PER_CU: chip-filling launches of the persistent kernels use one workgroup of 8 waves per CU (0: workgroups of 4 waves, placed by the dispatcher)
	int64_t walk_bands = -1;        // KWAGE_WALK_BANDS: the walk form takes the rows band after band of the matrix, all waves together (and_band_walk_kernel):
	                                //   -1 = three bands where the loader's probe found that the matrix's block mixes regions of the device's memory
	                                //   (the windowed probe > 3 % faster than the plain one), 0 = never, 2..64 = always, that many
	int64_t walk_bands_min_gib = 48;    // KWAGE_WALK_BANDS_MIN_GIB: a forced band count applies to matrices of at least this size
	int64_t and_vec = 0;            // KWAGE_AND_VEC: 16-byte vectors per lane of the tiled AND kernel (1, 2, 4; 0 = by row width)
	int64_t and_wide_min_kib = 16;  // KWAGE_AND_WIDE_MIN_KIB: ... for rows above this many KiB-steps (the walk form takes rows up to walk_max_kib first)
	int64_t and_wide = 1;           // KWAGE_AND_WIDE: rows beyond the walk form's range, no early exit, a chip-filling launch: vec 4, 8 rows, 8 waves per CU (and_config)
	int64_t narrow = 1;             // KWAGE_NARROW: several queries per wave for rows <= 512 B
	int64_t force_segs = 0;         // KWAGE_FORCE_SEGS: cut every query's k-mer list into this many segments (tests)
	int64_t ee_refine = 1;          // KWAGE_EE_REFINE: with early exit, tiles that still hold a candidate column after the first rows are handed over to the
	                                //   refine launch, which reads 128-byte groups on a balanced grid (0: the tile's own wave walks on 1-2 KiB wide)
	int64_t refine_seg_rows = 32;   // KWAGE_REFINE_SEG_ROWS: rows (k-mers at t < 1) per unit of the refine launch (t < 1: at most 120).  32: four units for what is left of a 150-base read
	                                //   (100 k reads: 2.29 vs 2.48 with 64 and 3.18 with 128; 1 kb queries the same with all three -- profiles/r05_refine_knobs_ab.txt)
	int64_t refine_min_rows = 32;   // KWAGE_REFINE_MIN_ROWS: a tile with fewer rows (k-mers) left finishes by itself
	int64_t refine_max_groups = 4;  // KWAGE_REFINE_MAX_GROUPS: a tile is handed over once at most this many of its 128-byte groups hold a candidate column
	int64_t refine_unroll = 8;      // KWAGE_REFINE_UNROLL: rows in flight per 128-byte group in the refine launch (8 or 16)
	int64_t screen_wpc = 20;        // KWAGE_SCREEN_WPC: waves per CU of the persistent screen launch
	int64_t count_screen_wpc = 32;  // KWAGE_COUNT_SCREEN_WPC: at most this many waves per CU in the count path's screen launch (fewer where the kernel's registers hold fewer)
	int64_t count_screen_min_tiles = 8192;  // KWAGE_COUNT_SCREEN_MIN_TILES: at t < 1, batches with fewer (query, KiB tile) pairs keep the tiled kernel and its segments
	int64_t count_trunc = 1;        // KWAGE_COUNT_TRUNC: early exit over few long queries at t < 1: the persistent count kernel over the first k-mers of every
	                                //   query (as many as the bound needs before it can rule a column out), the survivors refined (0: segments, every row read)
	int64_t count_screen_check = 8;     // KWAGE_COUNT_SCREEN_CHECK: k-mers between two looks at the bound in the count path's screen launch (8, 16, 32, 64; 8 = after every step:
	                                //   short reads at t = 0.8 9 % sooner than with 16, 14 % sooner than with 32; C2 at t = 0.8 the same -- profiles/r05_refine_knobs_ab.txt)
	int64_t refine_static = 1;      // KWAGE_REFINE_STATIC: half of every list is dealt out to the screen launch's waves beforehand (0: every place is reserved through the counters -- diagnostics: kwage_ctx_refine_stats then counts the hand-overs exactly)
	int64_t refine_list_cap = 0;    // KWAGE_REFINE_LIST_CAP: capacity of each of the three lists of handed-over tiles (0 = from the batch; tests: full lists)
	int64_t count_walk = 1;         // KWAGE_COUNT_WALK: the persistent count kernel where it applies
	int64_t count_walk_wpc = 8;     // KWAGE_COUNT_WALK_WPC: its waves per CU (8: 6335 GB/s at C2's shape, 12: 6271, 16: 6250, 20: 5876)
	int64_t count_walk_waves = 0;   // KWAGE_COUNT_WALK_WAVES: exactly this many waves (tests)
	int64_t count_walk_min_rows = -1;   // KWAGE_COUNT_WALK_MIN_ROWS: smaller batches use the tiled kernel (-1: 64 rows for each of its waves)
	int64_t hit_sort_host = 0;      // KWAGE_HIT_SORT=host: order long hit lists on the host (A/B runs, the fallback)
	int64_t hit_copy_piece_kb = 0;  // KWAGE_HIT_COPY_PIECE_KB: piece size of the copy-back of a long hit list (0 = default)
	int64_t ext_launch_events = 1;  // KWAGE_EXT_LAUNCH_EVENTS: a gather stage's start / end events ride on its kernel launches (hipExtLaunchKernelGGL: no
	                                //   barrier packets between consecutive gather kernels); 0 = plain hipEventRecord around the stage
	int64_t shared_table_log2 = 0;  // KWAGE_SHARED_TABLE_LOG2: at least this many slots in a sample's shared distinct set (tests)
	// where a group's matrix lies (loader.hip, allocate_matrix) -- read when a group is created
	int64_t group_contiguous = 1;   // KWAGE_GROUP_CONTIGUOUS: ask for a physically contiguous block first (0: plain hipMalloc)
	int64_t group_placement_probe = 1;  // KWAGE_GROUP_PLACEMENT_PROBE: where two candidate blocks fit, time the gather pattern on both and keep the
	                                //   faster (+3-4 % at C2's shape); releasing the other costs ~3 s per 100 GB (the driver wipes it), so one-shot programs turn it off
};

}  // namespace kwage

struct kwage_ctx {
	int device = -1;
	int ncu = 0;                        // compute units of the device (persistent grids are sized from it)
	kwage::Tuning tune;
	hipStream_t stream = nullptr;       // == slot[0].stream; loading, building and the synchronous calls use it
	hipStream_t gather_stream = nullptr;    // the gather kernels of both slots, in submission order (see Slot)
	hipStream_t upload_stream = nullptr;    // query batches and their layouts go to the device here: creating a batch never waits for a pending search
	kwage::Slot slot[2];
	kwage::DevBuf kmers;                       // kwage_hash_batch output
	kwage::DevPool batch_pool;                 // device blocks of destroyed query batches (kwage_batch_destroy never waits for the device)
	// database loading: two pinned + two device staging buffers, kept across files
	kwage::PinBuf load_pin[2];
	kwage::DevBuf load_dev[3];                 // [2] is used by the copy-engine pipeline only (three chunks in flight)
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
	std::shared_ptr<kwage::PinnedPool> result_pool = std::make_shared<kwage::PinnedPool>();      // result arrays of long hit lists
	// CPUs of the NUMA node the device hangs on (empty: unknown, or the process may not run there): database loading
	// runs on them (the page-cache pages it pins and the staging traffic then stay on the GPU's side of the host)
	std::vector<int> numa_cpus;
	int numa_node = -1;
};

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
	// how the matrix's block was chosen (loader.hip, allocate_matrix): candidates compared, the gather probe's GB/s on the kept and on the released one
	uint32_t placement_candidates = 0;
	double placement_kept_gbps = 0, placement_other_gbps = 0;
	// the same probe on the block kept with all waves reading from the same quarter of it at the same time: where that
	// is clearly faster the block mixes regions of the device's memory, and the walk form goes band after band (knob walk_bands = -1)
	double placement_windowed_gbps = 0;
	bool mixes_regions = false;
	std::vector<uint8_t> h_valid;
	bool finalized = false;
	double density = 0.25;         // share of set bits among the real columns, and ...
	double density_max = 1.0;      // ... in the DENSEST column, both from a few thousand rows sampled at finalize: the latter plans the truncated count walk
	// sparse group (kwage_group_create_sparse): the matrix holds only the listed rows of every file, in this order
	// (sorted, distinct); row indices from the k-mer stage are translated to positions in the list before the gather
	std::vector<uint32_t> h_row_map;
	uint32_t *d_row_map = nullptr;
};

namespace kwage {

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
	DevPool *pool = nullptr;           // where the four device arrays came from and go back to (the context's)
	uint64_t cap_pos_off = 0, cap_tab_off = 0, cap_chunk_q = 0, cap_chunk_t0 = 0;
	~KmerLayout()
	{
		if(!pool){ return; }
		pool->give(d_pos_off, cap_pos_off);
		pool->give(d_tab_off, cap_tab_off);
		pool->give(d_chunk_q, cap_chunk_q);
		pool->give(d_chunk_t0, cap_chunk_t0);
	}
};

}  // namespace kwage

struct kwage_batch {
	kwage_ctx *ctx = nullptr;
	uint32_t n = 0;
	uint64_t total_len = 0;
	char *d_seqs = nullptr;
	uint64_t *d_seq_off = nullptr;
	uint64_t cap_seqs = 0, cap_seq_off = 0;      // blocks of the context's batch pool
	std::vector<uint64_t> h_seq_off;
	// One layout per k-mer length the batch has been searched with (a database directory may hold files of several
	// k: two or three in practice).  A layout never changes once built, so searches with different k-mer lengths can
	// be in flight on the same batch side by side.
	std::vector<std::unique_ptr<kwage::KmerLayout>> layouts;
};

namespace kwage {

inline int set_device(kwage_ctx *ctx)
{
	HIP_TRY(hipSetDevice(ctx->device));
	return KWAGE_OK;
}

// loader.hip, called where a context is created and destroyed
void find_numa_cpus(int device, int *node, std::vector<int> *cpus);
void release_mapping(kwage_ctx *ctx);

}  // namespace kwage

#endif
