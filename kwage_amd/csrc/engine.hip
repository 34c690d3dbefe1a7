// kwage_amd/csrc/engine.hip -- device side of the C ABI declared in include/kwage_amd.h: contexts, tuning knobs, query
// batches and the search pipeline.  (Database groups and their loaders: loader.hip; shared state: engine_state.hpp.)
//
// One kwage_ctx = one GPU = two slots, each a HIP stream of its own.  A kwage_group owns the HBM-resident bit matrix of
// all same-parameter columns; a search runs, on its slot's stream,
//     kmer_kernel  ->  and_kernel | and_walk_kernel | count_kernel | count_walk_kernel | ...  ->  D2H of the hit list
// which together replace the reference's search() (kwage.cpp:340-541) for a whole batch of
// queries.  There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cctype>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "host.hpp"
#include "engine_state.hpp"
#include "kernels.hpp"

using namespace kwage;


struct TuningName { const char *name; int64_t Tuning::*field; };
static const TuningName TUNING_NAMES[] = {
	{"walk", &Tuning::walk}, {"walk_min_rows", &Tuning::walk_min_rows}, {"walk_max_kib", &Tuning::walk_max_kib}, {"walk_min_kib", &Tuning::walk_min_kib}, {"walk_tile_kib", &Tuning::walk_tile_kib},
	 {"walk_waves", &Tuning::walk_waves}, {"walk_one_wg_per_cu", &Tuning::walk_one_wg_per_cu},
	{"walk_bands", &Tuning::walk_bands}, {"walk_bands_min_gib", &Tuning::walk_bands_min_gib},
	{"and_vec", &Tuning::and_vec}, {"and_wide", &Tuning::and_wide}, {"and_wide_min_kib", &Tuning::and_wide_min_kib}, {"narrow", &Tuning::narrow}, {"force_segs", &Tuning::force_segs},
	{"ee_refine", &Tuning::ee_refine}, {"refine_seg_rows", &Tuning::refine_seg_rows}, {"refine_min_rows", &Tuning::refine_min_rows}, {"refine_max_groups", &Tuning::refine_max_groups}, {"refine_unroll", &Tuning::refine_unroll}, {"refine_list_cap", &Tuning::refine_list_cap}, {"refine_static", &Tuning::refine_static}, {"screen_wpc", &Tuning::screen_wpc}, {"count_screen_wpc", &Tuning::count_screen_wpc}, {"count_screen_min_tiles", &Tuning::count_screen_min_tiles}, {"count_screen_check", &Tuning::count_screen_check}, {"count_trunc", &Tuning::count_trunc},
	{"count_walk", &Tuning::count_walk}, {"count_walk_wpc", &Tuning::count_walk_wpc}, {"count_walk_waves", &Tuning::count_walk_waves},
	{"count_walk_min_rows", &Tuning::count_walk_min_rows},
	{"hit_sort_host", &Tuning::hit_sort_host}, {"hit_copy_piece_kb", &Tuning::hit_copy_piece_kb}, {"shared_table_log2", &Tuning::shared_table_log2},
	{"ext_launch_events", &Tuning::ext_launch_events}, {"group_contiguous", &Tuning::group_contiguous}, {"group_placement_probe", &Tuning::group_placement_probe},
};

namespace {

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
	L->pool = &b->ctx->batch_pool;
	HIP_TRY(L->pool->take(std::max<size_t>(chunk_q.size(), 1)*sizeof(uint32_t), (void**)&L->d_chunk_q, &L->cap_chunk_q));
	HIP_TRY(L->pool->take(std::max<size_t>(chunk_q.size(), 1)*sizeof(uint64_t), (void**)&L->d_chunk_t0, &L->cap_chunk_t0));
	HIP_TRY(L->pool->take(((size_t)n + 1)*sizeof(uint64_t), (void**)&L->d_pos_off, &L->cap_pos_off));
	HIP_TRY(L->pool->take(std::max<size_t>(n, 1)*sizeof(uint64_t), (void**)&L->d_tab_off, &L->cap_tab_off));
	// (on the context's upload stream, waited for here: the sources are locals, the layout may be used on either search
	// stream right away -- and nothing of this waits for a search that is pending on a slot)
	hipStream_t us = b->ctx->upload_stream;
	if(L->n_chunks){
		HIP_TRY(hipMemcpyAsync(L->d_chunk_q, chunk_q.data(), chunk_q.size()*sizeof(uint32_t), hipMemcpyHostToDevice, us));
		HIP_TRY(hipMemcpyAsync(L->d_chunk_t0, chunk_t0.data(), chunk_t0.size()*sizeof(uint64_t), hipMemcpyHostToDevice, us));
	}
	HIP_TRY(hipMemcpyAsync(L->d_pos_off, L->h_pos_off.data(), ((size_t)n + 1)*sizeof(uint64_t), hipMemcpyHostToDevice, us));
	if(n){ HIP_TRY(hipMemcpyAsync(L->d_tab_off, tab_off.data(), (size_t)n*sizeof(uint64_t), hipMemcpyHostToDevice, us)); }
	HIP_TRY(hipStreamSynchronize(us));
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

// The events of one gather stage, carried by its kernel launches themselves (hipExtLaunchKernelGGL): the first launch of
// a stage takes `start`, the last one `stop` -- the dispatch packets' own start / end timestamps, no barrier packet on the
// gather stream.  Recording them with hipEventRecord put two to three barrier packets between consecutive gather kernels,
// ~5 us each (rocprofv3 kernel trace: 22 us from one gather kernel's end to the next one's start on the same queue).
// Null members: the caller records plain events around the stage instead (knob ext_launch_events = 0).
struct StageEvents { hipEvent_t start = nullptr, stop = nullptr; };

#define KW_GATHER_LAUNCH(ge, first, last, kernel, grid, block, lds, stream, ...) \
	hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)(lds), stream, (first) ? (ge).start : nullptr, (last) ? (ge).stop : nullptr, 0u, __VA_ARGS__)

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

// Shape of the tiled AND kernel: VEC 16-byte vectors per lane (by row width; knob and_vec), and -- the WIDE shape only -- a
// dynamic-LDS pad that caps the waves per CU.  Eight rows in flight and nontemporal loads always.
struct AndCfg { int vec, lds_bytes; };

// `wide`: rows beyond the walk form's range searched without early exit by a launch that fills the chip (C3, C4 and their
// column shares): four vectors per lane, eight rows in flight, and a dynamic-LDS pad that keeps 8 waves per CU resident
// -- 32 KiB in flight per wave, 256 KiB per CU.  As for the persistent kernels, the memory system prefers few deep
// streams to many: against the default shape (32 waves per CU, 16 KiB each) +1.4 % on C3 and on the C4 share in one
// process (12 waves per CU +0.4 %, 16 and 20 the same as the default, 4 waves per CU -20 %; profiles/r04_tune_and_wide_rows.txt).
AndCfg and_config(const Tuning &t, uint32_t units_per_row, bool wide = false)
{
	AndCfg c;
	if(wide && t.and_vec == 0){ c.vec = 4; c.lds_bytes = 80*1024; return c; }
	c.vec = (t.and_vec == 1 || t.and_vec == 2 || t.and_vec == 4) ? (int)t.and_vec : ((units_per_row >= 4*WAVE) ? 2 : 1);
	c.lds_bytes = 0;
	return c;
}

template <int VEC>
void launch_and(const SearchArgs &a, hipStream_t s, const AndCfg &c, const StageEvents &ge)
{
	const uint64_t tiles = (uint64_t)a.n_queries*a.segs*a.chunks;
	const uint32_t bw = SEARCH_THREADS/WAVE;
	const dim3 grid((uint32_t)((tiles + bw - 1)/bw)), block(SEARCH_THREADS);
	if(c.lds_bytes > 48*1024){
		(void)hipFuncSetAttribute((const void*)and_kernel<VEC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds_bytes);
	}
	if(a.segs > 1){
		KW_GATHER_LAUNCH(ge, true, false, (and_kernel<VEC, true>), grid, block, 0, s, a);      // (and_combine_kernel ends the stage)
	}
	else{
		KW_GATHER_LAUNCH(ge, true, true, (and_kernel<VEC, false>), grid, block, c.lds_bytes, s, a);
	}
}

void launch_and_v(const SearchArgs &a, hipStream_t s, const AndCfg &c, const StageEvents &ge)
{
	if(c.vec == 1){ launch_and<1>(a, s, c, ge); }
	else if(c.vec == 2){ launch_and<2>(a, s, c, ge); }
	else{ launch_and<4>(a, s, c, ge); }
}

template <int PLANES, int NH>
void launch_count(const SearchArgs &a, hipStream_t s, const StageEvents &ge)
{
	if(a.segs > 1){
		KW_GATHER_LAUNCH(ge, true, false, (count_kernel<PLANES, NH, true>), dim3(search_blocks(a)), dim3(SEARCH_THREADS), 0, s, a);      // (count_combine_kernel ends the stage)
	}
	else{
		KW_GATHER_LAUNCH(ge, true, true, (count_kernel<PLANES, NH, false>), dim3(search_blocks(a)), dim3(SEARCH_THREADS), 0, s, a);
	}
}

template <int PLANES, int G>
void launch_count_narrow(const SearchArgs &a, hipStream_t s, const StageEvents &ge)
{
	const uint64_t waves = ((uint64_t)a.n_queries + G - 1)/G;
	const dim3 grid((uint32_t)((waves + 3)/4)), block(SEARCH_THREADS);
	switch(a.num_hash){
		case 1: KW_GATHER_LAUNCH(ge, true, true, (count_narrow_kernel<PLANES, 1, G>), grid, block, 0, s, a); break;
		case 2: KW_GATHER_LAUNCH(ge, true, true, (count_narrow_kernel<PLANES, 2, G>), grid, block, 0, s, a); break;
		case 3: KW_GATHER_LAUNCH(ge, true, true, (count_narrow_kernel<PLANES, 3, G>), grid, block, 0, s, a); break;
		case 4: KW_GATHER_LAUNCH(ge, true, true, (count_narrow_kernel<PLANES, 4, G>), grid, block, 0, s, a); break;
		default: KW_GATHER_LAUNCH(ge, true, true, (count_narrow_kernel<PLANES, 5, G>), grid, block, 0, s, a); break;
	}
}

template <int PLANES>
void launch_count_nh(const SearchArgs &a, hipStream_t s, const StageEvents &ge)
{
	switch(a.num_hash){
		case 1: launch_count<PLANES, 1>(a, s, ge); break;
		case 2: launch_count<PLANES, 2>(a, s, ge); break;
		case 3: launch_count<PLANES, 3>(a, s, ge); break;
		case 4: launch_count<PLANES, 4>(a, s, ge); break;
		default: launch_count<PLANES, 5>(a, s, ge); break;
	}
}

// counter planes for counts up to `max_count`: the smallest instantiated size whose bits hold it
uint32_t planes_for(uint64_t max_count)
{
	uint32_t bits = 1;
	while(bits < 32 && (max_count >> bits) != 0){ ++bits; }
	return (bits <= 7) ? 7 : (bits <= 10) ? 10 : (bits <= 14) ? 14 : (bits <= 20) ? 20 : 32;
}

void launch_count_planes(uint32_t planes, const SearchArgs &a, hipStream_t s, const StageEvents &ge)
{
	switch(planes){
		case 7: launch_count_nh<7>(a, s, ge); break;
		case 10: launch_count_nh<10>(a, s, ge); break;
		case 14: launch_count_nh<14>(a, s, ge); break;
		case 20: launch_count_nh<20>(a, s, ge); break;
		default: launch_count_nh<32>(a, s, ge); break;
	}
}

template <int PLANES, int NH, bool TRUNC>
void launch_count_walk(const SearchArgs &a, const CountWalkArgs &wa, const RefineArgs &ra, const WalkShape &w, hipStream_t s, const StageEvents &ge)
{
	if(w.lds > 48*1024){ (void)hipFuncSetAttribute((const void*)count_walk_kernel<PLANES, NH, TRUNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w.lds); }
	// (TRUNC: the refine and emit launches end the stage)
	KW_GATHER_LAUNCH(ge, true, !TRUNC, (count_walk_kernel<PLANES, NH, TRUNC>), dim3(w.wgs), dim3(w.wg_waves*WAVE), w.lds, s, a, wa, ra, a.rows, a.pos_off, a.nkmer, a.qthr);
}

template <int PLANES, bool TRUNC>
void launch_count_walk_nh(const SearchArgs &a, const CountWalkArgs &wa, const RefineArgs &ra, const WalkShape &w, hipStream_t s, const StageEvents &ge)
{
	switch(a.num_hash){
		case 1: launch_count_walk<PLANES, 1, TRUNC>(a, wa, ra, w, s, ge); break;
		case 2: launch_count_walk<PLANES, 2, TRUNC>(a, wa, ra, w, s, ge); break;
		case 3: launch_count_walk<PLANES, 3, TRUNC>(a, wa, ra, w, s, ge); break;
		case 4: launch_count_walk<PLANES, 4, TRUNC>(a, wa, ra, w, s, ge); break;
		default: launch_count_walk<PLANES, 5, TRUNC>(a, wa, ra, w, s, ge); break;
	}
}

template <bool TRUNC>
void launch_count_walk_planes(uint32_t planes, const SearchArgs &a, const CountWalkArgs &wa, const RefineArgs &ra, const WalkShape &w, hipStream_t s, const StageEvents &ge)
{
	switch(planes){
		// (the truncated form counts a query of up to 127 k-mers with ten planes: the caller sizes slab and lists for that)
		case 7: launch_count_walk_nh<TRUNC ? 10 : 7, TRUNC>(a, wa, ra, w, s, ge); break;
		case 10: launch_count_walk_nh<10, TRUNC>(a, wa, ra, w, s, ge); break;
		case 14: launch_count_walk_nh<14, TRUNC>(a, wa, ra, w, s, ge); break;
		case 20: launch_count_walk_nh<20, TRUNC>(a, wa, ra, w, s, ge); break;
		default: launch_count_walk_nh<32, TRUNC>(a, wa, ra, w, s, ge); break;
	}
}

struct RefineSetup { RefineArgs ra; uint32_t refine_wgs, emit_wgs; };

// The refine and emit launches of the count path's early-exit forms (count_screen_kernel, the truncated count walk).
void launch_count_refine_and_emit(uint32_t planes, int up, const SearchArgs &a, const RefineSetup &rs, hipStream_t gs, const StageEvents &ge)
{
	const dim3 block(SEARCH_THREADS);
#define KWAGE_CR_NH(N) case N: if(up == 7){ KW_GATHER_LAUNCH(ge, false, false, (count_refine_kernel<N, 7>), dim3(rs.refine_wgs), block, 0, gs, a, rs.ra); } \
		else{ KW_GATHER_LAUNCH(ge, false, false, (count_refine_kernel<N, 14>), dim3(rs.refine_wgs), block, 0, gs, a, rs.ra); } break;
	switch(std::min(a.num_hash, 5u)){ KWAGE_CR_NH(1) KWAGE_CR_NH(2) KWAGE_CR_NH(3) KWAGE_CR_NH(4) default: KWAGE_CR_NH(5) }
#undef KWAGE_CR_NH
#define KWAGE_CE_P(P) case P: if(up == 7){ KW_GATHER_LAUNCH(ge, false, true, (count_refine_emit_kernel<P, 7>), dim3(rs.emit_wgs), block, 0, gs, a, rs.ra); } \
		else{ KW_GATHER_LAUNCH(ge, false, true, (count_refine_emit_kernel<P, (P >= 14 ? 14 : 7)>), dim3(rs.emit_wgs), block, 0, gs, a, rs.ra); } break;
	switch(planes){ KWAGE_CE_P(7) KWAGE_CE_P(10) KWAGE_CE_P(14) KWAGE_CE_P(20) default: KWAGE_CE_P(32) }
#undef KWAGE_CE_P
}

template <int PLANES>
int launch_count_combine(const SearchArgs &a, uint32_t seg_planes, hipStream_t s, const StageEvents &ge)
{
	const size_t lds = (size_t)(COMBINE_WAVES/2)*PLANES*WAVE*16;
	if(lds > 48*1024){      // 32 planes only (queries above 2^20 positions); the attribute is per device, so set it per launch
		HIP_TRY(hipFuncSetAttribute((const void*)count_combine_kernel<PLANES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	}
	KW_GATHER_LAUNCH(ge, false, true, (count_combine_kernel<PLANES>), dim3((a.units_per_row + WAVE - 1)/WAVE, a.n_queries), dim3(COMBINE_WAVES*WAVE), lds, s, a, seg_planes);
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

// The lists of the early-exit form "screen, then refine" (kernels.hpp and_screen_kernel), sized from the batch: room for
// eight handed-over 128-byte groups per query (three planted columns per query is what the synthetic workloads hold; a
// tile that finds the lists full walks on by itself) and for the units of as many items of average length.  Half of
// every list is static -- its places dealt out to the `screen_waves` waves of the screen launch beforehand, taken without an
// atomic --, the rest is reserved in chunks through three counters, zeroed on the gather stream before every stage.
// `item_bytes`: mask (t = 1: 128) or counters (t < 1: planes x 128) per item; `unit_bytes`: partial counters per unit
// (t < 1 only).
int refine_setup(Slot *sl, const Tuning &tn, const SearchArgs &a, uint64_t total_rows, uint64_t max_rows, uint32_t max_seg, uint64_t item_bytes, uint64_t unit_bytes,
                 uint64_t tiles, uint64_t screen_waves, uint64_t ncu, hipStream_t gs, RefineSetup *out,
                 uint64_t exact_clusters = 0, uint64_t exact_items = 0, uint64_t exact_units = 0)
{
	int rc;
	RefineArgs &ra = out->ra;
	ra.seg_rows = (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.refine_seg_rows, 8), max_seg);
	ra.seg_rows = (uint32_t)std::max<uint64_t>(ra.seg_rows, (max_rows + 65535)/65536);              // (at most 2^16 units per item; the count path passes its own segment length)
	ra.min_rows = (uint32_t)std::max<int64_t>(tn.refine_min_rows, 1);
	ra.max_groups = (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.refine_max_groups, 0), 16);      // (0: nothing is ever handed over)
	uint64_t items = std::min<uint64_t>(std::max<uint64_t>(8ull*a.n_queries, 1u << 16), 1u << 20);
	items = std::max<uint64_t>(std::min<uint64_t>(items, (256ull << 20)/item_bytes), 1024);
	uint64_t units = std::min<uint64_t>(std::max<uint64_t>(32*total_rows/ra.seg_rows, 1u << 18), 1u << 25);       // (32 bytes each: at most 1 GiB)
	if(unit_bytes){ units = std::max<uint64_t>(std::min<uint64_t>(units, (512ull << 20)/unit_bytes), 4096); }
	if(tn.refine_list_cap > 0){ items = units = (uint64_t)std::min<int64_t>(tn.refine_list_cap, 1 << 20); }
	uint64_t clusters = items;
	// (the truncated count walk names its capacities: a place for every pair and group, units by memory)
	if(exact_units){ clusters = exact_clusters; items = exact_items; units = exact_units; }
	if(exact_units && tn.refine_list_cap > 0){ clusters = items = units = (uint64_t)std::min<int64_t>(tn.refine_list_cap, 1 << 20); }       // (tests: full lists)
	// dynamic chunks: what a wave is likely to need for a few of its tiles (a wave with one or two tiles takes exactly what it needs)
	const uint64_t tiles_per_wave = (tiles + screen_waves - 1)/std::max<uint64_t>(screen_waves, 1);
	auto list = [&](uint64_t cap, uint64_t stat_max, uint64_t chunk_max) {
		RefineList ls;
		ls.cap = (uint32_t)cap;
		// (exact capacities: every place is taken exactly when it is needed -- no static parts, no chunks, nothing left unused)
		ls.stat = (tn.refine_static && !exact_units) ? (uint32_t)std::min<uint64_t>(stat_max, cap/2/std::max<uint64_t>(screen_waves, 1)) : 0u;
		ls.base = (uint32_t)(ls.stat*screen_waves);
		ls.chunk = exact_units ? 1u : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(chunk_max, tiles_per_wave/4));
		return ls;
	};
	ra.lc = list(clusters, 32, 32);
	ra.li = list(items, 64, 64);
	ra.lu = list(units, 256, 256);
	if((rc = sl->ref_counters.reserve(8*sizeof(uint32_t)))){ return rc; }
	if((rc = sl->ref_clusters.reserve(clusters*sizeof(RefineCluster)))){ return rc; }
	if((rc = sl->ref_masks.reserve(items*item_bytes))){ return rc; }
	if((rc = sl->ref_units.reserve(units*sizeof(RefineUnit)))){ return rc; }
	if(unit_bytes && (rc = sl->ref_slab.reserve(units*unit_bytes))){ return rc; }
	ra.counters = (uint32_t*)sl->ref_counters.p;
	ra.clusters = (RefineCluster*)sl->ref_clusters.p;
	ra.masks = (uint32_t*)sl->ref_masks.p;
	ra.units = (RefineUnit*)sl->ref_units.p;
	ra.slab = (uint32_t*)sl->ref_slab.p;
	HIP_TRY(hipMemsetAsync(ra.counters, 0, 8*sizeof(uint32_t), gs));
	sl->ref_base[0] = ra.lc.base; sl->ref_base[1] = ra.li.base; sl->ref_base[2] = ra.lu.base;
	sl->ref_cap[0] = ra.lc.cap; sl->ref_cap[1] = ra.li.cap; sl->ref_cap[2] = ra.lu.cap;
	ra.queue_batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(16, tiles/(8*std::max<uint64_t>(screen_waves, 1))));
	ra.check_every = (tn.count_screen_check == 16 || tn.count_screen_check == 32 || tn.count_screen_check == 64) ? (uint32_t)tn.count_screen_check : 8u;
	out->refine_wgs = (uint32_t)(ncu*8);          // 32 waves per CU, eight units each
	out->emit_wgs = (uint32_t)(ncu*4);            // 16 waves per CU, four clusters at a time each
	return KWAGE_OK;
}

// Workgroups of count_screen_kernel<planes, nh> a CU holds at once (registers; asked of the runtime once per shape).
uint32_t count_screen_blocks_per_cu(uint32_t planes, uint32_t nh)
{
	static int cache[3][5] = {};
	const int pi = (planes <= 7) ? 0 : (planes <= 10) ? 1 : 2, ni = (int)std::min(std::max(nh, 1u), 5u) - 1;
	if(cache[pi][ni] == 0){
		int nb = 0;
		hipError_t e = hipErrorUnknown;
#define KWAGE_OCC_NH(P, N) case N: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, count_screen_kernel<P, N>, SEARCH_THREADS, 0); break;
#define KWAGE_OCC_P(I, P) case I: switch(ni + 1){ KWAGE_OCC_NH(P, 1) KWAGE_OCC_NH(P, 2) KWAGE_OCC_NH(P, 3) KWAGE_OCC_NH(P, 4) default: KWAGE_OCC_NH(P, 5) } break;
		switch(pi){ KWAGE_OCC_P(0, 7) KWAGE_OCC_P(1, 10) default: KWAGE_OCC_P(2, 14) }
#undef KWAGE_OCC_P
#undef KWAGE_OCC_NH
		if(e != hipSuccess || nb <= 0){ (void)hipGetLastError(); nb = 2; }
		cache[pi][ni] = nb;
	}
	return (uint32_t)cache[pi][ni];
}

// Launch the gather+reduce kernel(s) for the current batch.
int launch_search_stage(Slot *sl, kwage_group *g, kwage_batch *b, const KmerLayout *L, float threshold, uint32_t flags,
                        kwage_hit *d_hits, uint64_t cap, unsigned long long *hit_count, hipStream_t gs, const StageEvents &ge)
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
	a.runs = sl->n_runs ? (unsigned long long*)sl->runs.p : nullptr;
	a.runs_per_query = sl->runs_per_query;
	int rc;

	if(threshold == 1.0f){
		// The wide shape of the tiled kernel (and_config) where a launch without early exit fills the chip with 4 KiB tiles: for
		// rows beyond the walk form's range.  (Within that range the walk form is ahead for every row-list length since its
		// waves buffer their hit records: a rule that sent batches of short row lists to the wide shape was dropped in round 4,
		// its knob in round 5; profiles/r04_walk_vs_tiled_wide_grid.txt, r04_walk_hit_cost.txt.)
		const uint64_t wide_tiles = (uint64_t)a.n_queries*((a.units_per_row + 4*WAVE - 1)/(4*WAVE));
		const bool wide_ok = tn.and_wide && !(flags & KWAGE_SEARCH_EARLY_EXIT) && wide_tiles >= 8*ncu;
		const bool wide_rows = wide_ok && a.units_per_row > (uint32_t)std::max<int64_t>(tn.and_wide_min_kib, 1)*WAVE;
		const AndCfg cfg = and_config(tn, a.units_per_row, wide_rows);
		a.chunks = (a.units_per_row + WAVE*cfg.vec - 1)/(WAVE*cfg.vec);
		choose_segments(a, L->max_pos, 4096, tn.force_segs);
		if((uint64_t)a.n_queries*a.segs*a.chunks/4 + 1 > 0x7FFFFFFFull){ return fail(KWAGE_ERR_ARG, "batch too large for one launch"); }
		// narrow rows: several queries per wave (tools/bench_narrow.py)
		if(tn.narrow && a.segs == 1 && a.units_per_row <= 32 && a.n_queries >= 64){
			const uint32_t G = (a.units_per_row <= 4) ? 16 : (a.units_per_row <= 8) ? 8 : (a.units_per_row <= 16) ? 4 : 2;
			const uint64_t waves = ((uint64_t)a.n_queries + G - 1)/G;
			const dim3 grid((uint32_t)((waves + 3)/4)), block(SEARCH_THREADS);
			// few waves (10 k queries of 1 kb: ten per CU): sixteen rows in flight per wave instead of eight
			const int unroll = (waves < ncu*16) ? 16 : 8;
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "and_narrow_kernel<%u,%d>", G, unroll);
#define KWAGE_NARROW_CASE(GG) case GG: \
				if(unroll == 16){ KW_GATHER_LAUNCH(ge, true, true, (and_narrow_kernel<GG, 16>), grid, block, 0, gs, a); } \
				else{ KW_GATHER_LAUNCH(ge, true, true, (and_narrow_kernel<GG, 8>), grid, block, 0, gs, a); } break;
			switch(G){
				KWAGE_NARROW_CASE(16) KWAGE_NARROW_CASE(8) KWAGE_NARROW_CASE(4)
				default: KWAGE_NARROW_CASE(2)
			}
#undef KWAGE_NARROW_CASE
			HIP_TRY(hipGetLastError());
			return KWAGE_OK;
		}
		// early exit on rows of a KiB and more: screen, then refine (kernels.hpp and_screen_kernel)
		if(a.early_exit && tn.ee_refine && a.units_per_row >= WAVE && tn.force_segs <= 0){
			RefineSetup rs;
			const int vec = (a.units_per_row >= 4*WAVE) ? 2 : 1;
			a.segs = 1;
			a.chunks = (a.units_per_row + WAVE*vec - 1)/(WAVE*vec);
			const uint64_t tiles = (uint64_t)a.n_queries*a.chunks;
			// (a persistent grid: tile t goes to wave t mod waves; 20 waves per CU is what the kernel's registers allow)
			const uint64_t screen_waves = (std::min<uint64_t>(tiles, ncu*(uint64_t)std::max<int64_t>(tn.screen_wpc, 1)) + 3)/4*4;
			if((rc = refine_setup(sl, tn, a, L->total_pos*a.num_hash, L->max_pos*a.num_hash, 1u << 20, 128, 0, tiles, screen_waves, ncu, gs, &rs))){ return rc; }
			const int unroll = (tn.refine_unroll == 16) ? 16 : 8;
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "and_screen_kernel<%d,8>+refine<%d>", vec, unroll);
			const dim3 grid((uint32_t)(screen_waves/4)), block(SEARCH_THREADS);
			if(vec == 2){ KW_GATHER_LAUNCH(ge, true, false, (and_screen_kernel<2, 8>), grid, block, 0, gs, a, rs.ra); }
			else{ KW_GATHER_LAUNCH(ge, true, false, (and_screen_kernel<1, 8>), grid, block, 0, gs, a, rs.ra); }
			if(unroll == 16){ KW_GATHER_LAUNCH(ge, false, false, (and_refine_kernel<16>), dim3(rs.refine_wgs), block, 0, gs, a, rs.ra); }
			else{ KW_GATHER_LAUNCH(ge, false, false, (and_refine_kernel<8>), dim3(rs.refine_wgs), block, 0, gs, a, rs.ra); }
			KW_GATHER_LAUNCH(ge, false, true, and_refine_emit_kernel, dim3(rs.emit_wgs), block, 0, gs, a, rs.ra);
			HIP_TRY(hipGetLastError());
			return KWAGE_OK;
		}
		// rows of 3..16 KiB: the walk form (a persistent grid, every wave walks an equal share of the batch's row
		// list over the whole width of a column tile; kernels.hpp and_walk_kernel).  Worth it once every wave of
		// the chip gets a few dozen rows; smaller batches stay with the tiled kernel and its row-list segments.
		const int walk_knob = (int)tn.walk;
		const uint64_t walk_min_rows = (tn.walk_min_rows >= 0) ? (uint64_t)tn.walk_min_rows : (uint64_t)WALK_MIN_ROWS_PER_WAVE*256*WALK_WAVES_PER_CU;
		const uint32_t kib = (a.units_per_row + WAVE - 1)/WAVE;
		// (the kernel handles wider rows as several balanced column tiles -- the walk_max_kib knob raises the limit --
		// but 125 KB rows measured no gain over the tiled kernel)
		const uint32_t walk_max_kib = (uint32_t)std::max<int64_t>(tn.walk_max_kib, 0);
		// (knob walk_tile_kib: narrower column tiles -- with 4, eight rows in flight and no pacing a wave requests 4 KiB of each
		// of eight rows at once: the tiled kernel's wide shape on the balanced persistent grid)
		const uint32_t walk_tile = (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.walk_tile_kib, 1), 16);
		const uint32_t coltiles = (kib + walk_tile - 1)/walk_tile, walk_ch = (kib + coltiles - 1)/coltiles;     // balanced tiles of <= walk_tile KiB
		// (early exit never comes here: a walking wave covers a column tile's whole width and has nothing to give up --
		// C2 with early exit measured 1.29 ms through the walk form against 0.64 ms tiled; the screen + refine form above
		// takes rows of a KiB and more, the tiled kernel below the rest)
		const bool walk_ee_ok = !a.early_exit;
		const uint64_t walk_slots = (uint64_t)coltiles*L->total_pos;
		const uint32_t walk_min_kib = (uint32_t)std::max<int64_t>(tn.walk_min_kib, 1);
		// rows in flight per wave: 4; 8 for rows of one or two KiB-steps (a group of four such rows is only 4-8 loads: C2's
		// columns split 8 ways, 1664-byte rows, 0.2626 vs 0.2667 ms)
		const int walk_unroll = (walk_ch <= 2) ? 8 : 4;
		if(walk_knob && walk_ee_ok && kib >= walk_min_kib && kib <= walk_max_kib && walk_slots*a.num_hash >= walk_min_rows && walk_slots > 0){
			// WALK_WAVES_PER_CU waves per CU, all resident at once (__launch_bounds__(256, 4) allows twice as many),
			// fewer when the batch is small: a wave should have WALK_MIN_ROWS_PER_WAVE rows to walk
			const uint64_t chip_waves = ncu*WALK_WAVES_PER_CU;
			const uint64_t want_waves = (tn.walk_waves > 0) ? std::min<uint64_t>((uint64_t)tn.walk_waves, walk_slots)
				: std::max<uint64_t>(1, std::min<uint64_t>(chip_waves, walk_slots*a.num_hash/WALK_MIN_ROWS_PER_WAVE));
			const WalkShape shape = walk_shape(tn, want_waves, ncu);
			const uint32_t wgs = shape.wgs;
			const uint64_t waves = (uint64_t)wgs*shape.wg_waves;
			// Band after band (and_band_walk_kernel): matrices large enough to span several regions of the device's memory,
			// one column tile, a slot of 16 KiB per query affordable.  (With early exit the tiled kernel is the default anyway.)
			const uint64_t band_items = L->total_pos*a.num_hash;
			// (a matrix of at least walk_bands_min_gib either way: the probe's "mixes regions" on a 14 GB matrix of 1.6 KB rows --
			// C2's columns split 8 ways -- sent that shape through the band form at 2.8x the walk form's time)
			const bool big_enough = g->alloc_bytes >= ((uint64_t)std::max<int64_t>(tn.walk_bands_min_gib, 0) << 30);
			const uint32_t bands = !big_enough ? 0u : (tn.walk_bands < 0) ? (g->mixes_regions ? 3u : 0u) : (uint32_t)std::min<int64_t>(tn.walk_bands, BAND_MAX);
			if(bands >= 2 && coltiles == 1 &&
			   (uint64_t)a.n_queries*16*1024 <= (256ull << 20) && band_items < 0x7FFFFFFFull){
				BandArgs ba;
				ba.bands = bands;
				ba.rows_per_band = (uint32_t)((g->nrows + bands - 1)/bands);
				if((rc = sl->band_rows.reserve(band_items*sizeof(uint32_t)))){ return rc; }
				if((rc = sl->band_loc.reserve((uint64_t)a.n_queries*(bands + 1)*sizeof(uint32_t)))){ return rc; }
				if((rc = reserve_zeroed(sl->band_or, (uint64_t)a.n_queries*16*1024, gs))){ return rc; }
				if((rc = reserve_zeroed(sl->band_state, (uint64_t)a.n_queries*sizeof(uint32_t), gs))){ return rc; }
				ba.orbuf = (uint32_t*)sl->band_or.p;
				ba.state = (uint32_t*)sl->band_state.p;
				uint32_t *loc = (uint32_t*)sl->band_loc.p, *rows2 = (uint32_t*)sl->band_rows.p;
				KW_GATHER_LAUNCH(ge, true, false, band_bucket_kernel, dim3(a.n_queries), dim3(256), 0, gs, a.rows, a.pos_off, a.nkmer, a.num_hash,
				                 ba.bands, ba.rows_per_band, loc, rows2);
				WalkArgs wb;
				wb.total_slots = walk_slots;                         // (one column tile: slots = positions)
				wb.per_wave = (walk_slots + waves - 1)/waves;
				wb.coltiles = 1;
				wb.orbuf = nullptr; wb.done = nullptr;
				a.segs = 1;
				a.chunks = 1;
				snprintf(sl->kernel_name, sizeof(sl->kernel_name), "and_band_walk_kernel<%u,4>", walk_ch);
				const dim3 grid(wgs), block(shape.wg_waves*WAVE), fgrid((a.n_queries + 3)/4);
#define KWAGE_BAND_LAUNCH(CH, U) do { \
					if(shape.lds > 48*1024){ (void)hipFuncSetAttribute((const void*)and_band_walk_kernel<CH, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shape.lds); } \
					KW_GATHER_LAUNCH(ge, false, false, (and_band_walk_kernel<CH, U>), grid, block, shape.lds, gs, a, ba, wb, (const uint32_t*)rows2, (const uint32_t*)loc, a.pos_off, a.nkmer); \
					KW_GATHER_LAUNCH(ge, false, true, (and_band_finish_kernel<CH>), fgrid, dim3(256), 0, gs, a, ba, a.nkmer); } while(0)
#define KWAGE_BAND_CASE(CH) case CH: \
					KWAGE_BAND_LAUNCH(CH, 4); break;
				switch(walk_ch){
					KWAGE_BAND_CASE(3) KWAGE_BAND_CASE(4) KWAGE_BAND_CASE(5) KWAGE_BAND_CASE(6) KWAGE_BAND_CASE(7)
					KWAGE_BAND_CASE(8) KWAGE_BAND_CASE(9) KWAGE_BAND_CASE(10) KWAGE_BAND_CASE(11) KWAGE_BAND_CASE(12)
					KWAGE_BAND_CASE(13) KWAGE_BAND_CASE(14) KWAGE_BAND_CASE(15)
					default: KWAGE_BAND_CASE(16)
				}
#undef KWAGE_BAND_CASE
#undef KWAGE_BAND_LAUNCH
				HIP_TRY(hipGetLastError());
				return KWAGE_OK;
			}
			WalkArgs wa;
			wa.total_slots = walk_slots;
			wa.per_wave = (walk_slots + waves - 1)/waves;
			wa.coltiles = coltiles;
			// cut-pair slots: one per wave, 16 KiB each whatever CH is; the kernel leaves them zero
			if((rc = reserve_zeroed(sl->walk_or, waves*16*1024, gs))){ return rc; }
			if((rc = reserve_zeroed(sl->walk_done, waves*2*sizeof(uint32_t), gs))){ return rc; }
			wa.orbuf = (uint32_t*)sl->walk_or.p;
			wa.done = (uint32_t*)sl->walk_done.p;
			a.segs = 1;
			a.chunks = coltiles;
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "and_walk_kernel<%u,%d>", walk_ch, walk_unroll);
			const dim3 grid(wgs), block(shape.wg_waves*WAVE);
#define KWAGE_WALK_LAUNCH(...) do { \
				if(shape.lds > 48*1024){ (void)hipFuncSetAttribute((const void*)and_walk_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shape.lds); } \
				KW_GATHER_LAUNCH(ge, true, true, (and_walk_kernel<__VA_ARGS__>), grid, block, shape.lds, gs, a, wa, a.rows, a.pos_off, a.nkmer); } while(0)
#define KWAGE_WALK_CASE(CH) case CH: KWAGE_WALK_LAUNCH(CH, 4); break;
#define KWAGE_WALK_CASE8(CH) case CH: KWAGE_WALK_LAUNCH(CH, 8); break;
			switch(walk_ch){
				KWAGE_WALK_CASE8(1) KWAGE_WALK_CASE8(2) KWAGE_WALK_CASE(3) KWAGE_WALK_CASE(4) KWAGE_WALK_CASE(5) KWAGE_WALK_CASE(6) KWAGE_WALK_CASE(7)
				KWAGE_WALK_CASE(8) KWAGE_WALK_CASE(9) KWAGE_WALK_CASE(10) KWAGE_WALK_CASE(11) KWAGE_WALK_CASE(12)
				KWAGE_WALK_CASE(13) KWAGE_WALK_CASE(14) KWAGE_WALK_CASE(15)
				default: KWAGE_WALK_CASE(16)
			}
#undef KWAGE_WALK_CASE8
#undef KWAGE_WALK_CASE
#undef KWAGE_WALK_LAUNCH
			HIP_TRY(hipGetLastError());
			return KWAGE_OK;
		}
		if(a.segs > 1){
			const uint64_t bytes = (uint64_t)a.n_queries*g->stride;
			if((rc = sl->partial.reserve(bytes))){ return rc; }
			HIP_TRY(hipMemsetAsync(sl->partial.p, 0xFF, bytes, gs));
			a.partial = (uint32_t*)sl->partial.p;
		}
		snprintf(sl->kernel_name, sizeof(sl->kernel_name), "and_kernel<%d,8,nt>%s", cfg.vec, a.segs > 1 ? "+segments" : "");      // (shape: vectors per lane, rows in flight, nontemporal)
		launch_and_v(a, gs, cfg, ge);
		if(a.segs > 1){
			KW_GATHER_LAUNCH(ge, false, true, and_combine_kernel, dim3((a.units_per_row + 255)/256, a.n_queries), dim3(256), 0, gs, a);
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
		// early exit on rows of a KiB and more, enough (query, KiB tile) pairs to keep a persistent grid busy: screen, then refine
		// (kernels.hpp count_screen_kernel).  Few long queries stay with the segments below: every tile has to read the first
		// (1 - t) n k-mers before the bound can prune anything, and a handful of tiles that long would be the whole launch.
		if(a.early_exit && tn.ee_refine && !narrow && tn.force_segs <= 0 && a.units_per_row >= WAVE
		   && (uint64_t)a.n_queries*a.chunks >= (uint64_t)std::max<int64_t>(tn.count_screen_min_tiles, 1) && planes <= 14){      // (queries of up to 16383 positions: 20 and 32 counter planes + eight rows in flight do not fit a wave's registers)
			const uint64_t tiles = (uint64_t)a.n_queries*a.chunks;
			// k-mers per unit: 64 (7 counter planes per unit); queries above 8192 positions: 1/128 of the longest (14 planes)
			uint32_t seg = (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.refine_seg_rows, 8), 120)/8*8;
			int up = 7;
			if(L->max_pos > 8192){ seg = (uint32_t)((L->max_pos + 127)/128 + 7)/8*8; up = 14; }
			// (as many waves as the kernel's registers let a CU hold -- knob count_screen_wpc caps it; they draw their tiles from a queue)
			const uint64_t wpc = std::min<uint64_t>((uint64_t)std::max<int64_t>(tn.count_screen_wpc, 1), 4ull*count_screen_blocks_per_cu(planes, std::min(a.num_hash, 5u)));
			const uint64_t screen_waves = (std::min<uint64_t>(tiles, ncu*wpc) + 3)/4*4;
			RefineSetup rs;
			if((rc = refine_setup(sl, tn, a, L->total_pos, 0, seg, (uint64_t)planes*128, (uint64_t)up*128, tiles, screen_waves, ncu, gs, &rs))){ return rc; }
			rs.ra.seg_rows = seg;
			a.segs = 1;
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "count_screen_kernel<%u,%u>+refine<%d>", planes, std::min(a.num_hash, 5u), up);
			const dim3 grid((uint32_t)(screen_waves/4)), block(SEARCH_THREADS);
#define KWAGE_CS_NH(P, N) case N: KW_GATHER_LAUNCH(ge, true, false, (count_screen_kernel<P, N>), grid, block, 0, gs, a, rs.ra); break;
#define KWAGE_CS_P(P) case P: switch(std::min(a.num_hash, 5u)){ KWAGE_CS_NH(P, 1) KWAGE_CS_NH(P, 2) KWAGE_CS_NH(P, 3) KWAGE_CS_NH(P, 4) default: KWAGE_CS_NH(P, 5) } break;
			switch(planes){ KWAGE_CS_P(7) KWAGE_CS_P(10) default: KWAGE_CS_P(14) }
#undef KWAGE_CS_P
#undef KWAGE_CS_NH
			launch_count_refine_and_emit(planes, up, a, rs, gs, ge);
			HIP_TRY(hipGetLastError());
			return KWAGE_OK;
		}
		// Early exit over FEW LONG queries (fewer tiles than the screen form wants, or more counter planes than it holds): every
		// tile has to read the first (1 - t) n k-mers plus a margin before kwage.cpp:478-481 can rule anything out, and a handful of
		// tiles that long would be the whole launch -- so that part is walked by the BALANCED persistent count kernel over the
		// first kcut(q) k-mers of every query (count_walk_kernel<.., TRUNC>), the columns that can still reach the threshold are
		// handed over with their counters, and the refine launch counts their remaining k-mers in 128-byte groups.  kcut comes
		// from the density of the matrix's DENSEST column (sampled at finalize): the smallest K with  K - (n - thr)  >=  pm K + 4.5 sqrt(pm (1 - pm) K) + 8,
		// pm = that density ^ num_hash -- an estimate that decides how much is read, never what is reported.
		if(a.early_exit && tn.ee_refine && tn.count_trunc && tn.count_walk && !narrow && tn.force_segs <= 0 && a.units_per_row >= WAVE && L->total_pos > 0){
			const uint32_t nq = a.n_queries;
			const uint32_t tplanes = std::max<uint32_t>(planes, 10);      // (no 7-plane instantiation of this form)
			// The DENSEST column decides: a 128-byte group stays alive while any of its 1024 columns can still reach the threshold,
			// and columns are not equally dense (every sample's filter has its own fill; the synthetic workloads' planted columns
			// are 15 % denser than the rest).  + 3 sigma of what 4096 sampled rows can say about one column.
			const double col = std::min(1.0, std::max(0.0, g->density_max) + 3.0*std::sqrt(0.25/4096.0));
			const double pm = std::min(0.9, std::max(0.0005, std::pow(col, (double)a.num_hash)));
			static const bool trunc_debug = getenv("KWAGE_TRUNC_DEBUG") != nullptr;
			if(trunc_debug){ fprintf(stderr, "[kwage_amd] truncated count walk: sampled density %.5f, densest column %.5f, per-k-mer match probability planned with %.5f\n", g->density, g->density_max, pm); }
			if((rc = sl->trunc_host.reserve(((uint64_t)nq + 1)*sizeof(uint64_t) + (uint64_t)nq*sizeof(uint32_t)))){ return rc; }
			uint64_t *h_soff = (uint64_t*)sl->trunc_host.p;
			uint32_t *h_kcut = (uint32_t*)(h_soff + nq + 1);
			uint64_t walked = 0, max_rest = 0, rest_total = 0;
			h_soff[0] = 0;
			for(uint32_t i = 0; i < nq; ++i){
				const uint64_t npos = L->h_pos_off[i + 1] - L->h_pos_off[i];
				uint64_t kc = npos;
				if(npos >= 64){
					const uint32_t thr = (uint32_t)(threshold*(float)npos);      // kwage.cpp:388 on the positions (an upper bound of the distinct k-mers)
					const double need = (double)(npos - std::min<uint64_t>(thr, npos));
					double K = (need + 8.0)/(1.0 - pm);
					for(int it = 0; it < 8; ++it){ K = (need + 8.0 + 4.5*std::sqrt(pm*(1.0 - pm)*K))/(1.0 - pm); }
					const uint64_t k8 = ((uint64_t)K + 7)/8*8;
					if(k8*20 <= npos*17){ kc = k8; }                              // (worth it when at least 15 % of the list is left out)
				}
				h_kcut[i] = (uint32_t)kc;
				h_soff[i + 1] = h_soff[i] + kc;
				walked += kc;
				max_rest = std::max(max_rest, npos - kc);
				rest_total += npos - kc;
			}
			const uint64_t slots = (uint64_t)a.chunks*walked;
			const uint64_t chip_waves = ncu*(uint64_t)std::max<int64_t>(tn.count_walk_wpc, 1);
			const uint64_t min_rows = (tn.count_walk_min_rows >= 0) ? (uint64_t)tn.count_walk_min_rows : (uint64_t)WALK_MIN_ROWS_PER_WAVE*chip_waves;
			// units of 1/128 of the longest remainder (at least refine_seg_rows k-mers): 7 counter planes per unit up to 120 k-mers, 14 beyond
			uint64_t seg = (uint64_t)std::min<int64_t>(std::max<int64_t>(tn.refine_seg_rows, 8), 120)/8*8;
			seg = std::max<uint64_t>(seg, ((max_rest + 127)/128 + 7)/8*8);
			const int up = (seg <= 120) ? 7 : 14;
			uint64_t units_worst = 0;
			for(uint32_t i = 0; i < nq && seg; ++i){
				const uint64_t rest = (L->h_pos_off[i + 1] - L->h_pos_off[i]) - h_kcut[i];
				units_worst += (uint64_t)a.chunks*8*((rest + seg - 1)/seg);
			}
			// Lists: a place for every pair and every 128-byte group of it (few tiles: that is affordable), and for the units
			// as many as a GiB of partial counters holds -- every group of every pair surviving is not what they are sized for;
			// a pair that finds them full is counted to the end by the wave that holds it (count_walk_kernel).
			const uint64_t tiles = (uint64_t)nq*a.chunks;
			const uint64_t units_cap = std::max<uint64_t>(4096, std::min<uint64_t>(units_worst, (1ull << 30)/((uint64_t)up*128 + sizeof(RefineUnit))));
			if(walked*10 <= L->total_pos*7 && slots*a.num_hash >= min_rows && seg <= 16376 && rest_total > 0 && tiles*8*(uint64_t)tplanes*128 <= (1ull << 30)){
				const uint64_t want_waves = (tn.count_walk_waves > 0) ? std::min<uint64_t>((uint64_t)tn.count_walk_waves, slots)
					: std::max<uint64_t>(1, std::min<uint64_t>(chip_waves, slots*a.num_hash/WALK_MIN_ROWS_PER_WAVE));
				const WalkShape shape = walk_shape(tn, want_waves, ncu);
				const uint64_t waves = (uint64_t)shape.wgs*shape.wg_waves;
				RefineSetup rs;
				if((rc = refine_setup(sl, tn, a, rest_total, 0, (uint32_t)seg, (uint64_t)tplanes*128, (uint64_t)up*128, tiles, waves, ncu, gs, &rs, tiles, tiles*8, units_cap))){ return rc; }
				rs.ra.seg_rows = (uint32_t)seg;
				if((rc = sl->trunc_dev.reserve(((uint64_t)nq + 1)*sizeof(uint64_t) + (uint64_t)nq*sizeof(uint32_t)))){ return rc; }
				HIP_TRY(hipMemcpyAsync(sl->trunc_dev.p, sl->trunc_host.p, ((uint64_t)nq + 1)*sizeof(uint64_t) + (uint64_t)nq*sizeof(uint32_t), hipMemcpyHostToDevice, gs));
				CountWalkArgs wa;
				wa.total_slots = slots;
				wa.per_wave = (slots + waves - 1)/waves;
				wa.coltiles = a.chunks;
				if((rc = sl->cwalk_slab.reserve(waves*2*tplanes*1024))){ return rc; }
				if((rc = reserve_zeroed(sl->cwalk_arrived, waves*CWALK_LEVELS*sizeof(uint32_t), gs))){ return rc; }
				wa.slab = (uint32_t*)sl->cwalk_slab.p;
				wa.arrived = (uint32_t*)sl->cwalk_arrived.p;
				wa.slot_off = (const uint64_t*)sl->trunc_dev.p;
				wa.kcut = (const uint32_t*)((const uint64_t*)sl->trunc_dev.p + nq + 1);
				a.segs = 1;
				snprintf(sl->kernel_name, sizeof(sl->kernel_name), "count_walk_kernel<%u,%u,trunc>+refine<%d>", tplanes, std::min(a.num_hash, 5u), up);
				launch_count_walk_planes<true>(tplanes, a, wa, rs.ra, shape, gs, ge);
				launch_count_refine_and_emit(tplanes, up, a, rs, gs, ge);
				HIP_TRY(hipGetLastError());
				return KWAGE_OK;
			}
		}
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
				if((rc = reserve_zeroed(sl->cwalk_arrived, waves*CWALK_LEVELS*sizeof(uint32_t), gs))){ return rc; }
				wa.slab = (uint32_t*)sl->cwalk_slab.p;
				wa.arrived = (uint32_t*)sl->cwalk_arrived.p;
				a.segs = 1;
				// (shape: planes, hashes, the next step's rows prefetched, eight k-mers per step with 14 planes and more)
				snprintf(sl->kernel_name, sizeof(sl->kernel_name), "count_walk_kernel<%u,%u,pf%s>", planes, std::min(a.num_hash, 5u), planes >= 14 ? ",8" : "");
				wa.slot_off = nullptr; wa.kcut = nullptr;
				launch_count_walk_planes<false>(planes, a, wa, RefineArgs(), shape, gs, ge);
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
			const int kps = 8;
			snprintf(sl->kernel_name, sizeof(sl->kernel_name), "count_narrow_kernel<%u,%u,%d,%d>", planes, a.num_hash, a.units_per_row <= 16 ? 4 : 2, kps);
			if(a.units_per_row <= 16){
				if(planes == 7){ launch_count_narrow<7, 4>(a, gs, ge); }
				else if(planes == 10){ launch_count_narrow<10, 4>(a, gs, ge); }
				else{ launch_count_narrow<14, 4>(a, gs, ge); }
			}
			else{
				if(planes == 7){ launch_count_narrow<7, 2>(a, gs, ge); }
				else if(planes == 10){ launch_count_narrow<10, 2>(a, gs, ge); }
				else{ launch_count_narrow<14, 2>(a, gs, ge); }
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
		launch_count_planes(seg_planes, a, gs, ge);
		if(a.segs > 1){
			HIP_TRY(hipGetLastError());
			switch(planes){
				case 7: rc = launch_count_combine<7>(a, seg_planes, gs, ge); break;
				case 10: rc = launch_count_combine<10>(a, seg_planes, gs, ge); break;
				case 14: rc = launch_count_combine<14>(a, seg_planes, gs, ge); break;
				case 20: rc = launch_count_combine<20>(a, seg_planes, gs, ge); break;
				default: rc = launch_count_combine<32>(a, seg_planes, gs, ge); break;
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
		// The gather stage of both slots goes to the context's ONE gather stream, in submission order: gather kernels
		// never run side by side (they would only share the HBM bandwidth and stretch each other), and the next one
		// starts right behind the previous one on the same hardware queue -- a cross-queue event between two gather
		// kernels cost ~25 us per search (rocprofv3 kernel trace, profiles/r04_*).  What the stage reads (row list,
		// counts, zeroed counters) was queued on the slot's stream: the gather stream waits for that -- normally long
		// done, the k-mer stage ran beside the previous gather kernel -- and the slot's stream, which carries the
		// copy-back, waits for the gather stage.
		hipStream_t gs = g->ctx->gather_stream;
		// (the run table of the search's own list: most runs stay empty; cleared beside the previous gather kernel)
		if(sl->n_runs){ HIP_TRY(hipMemsetAsync(sl->runs.p, 0, sl->n_runs*sizeof(unsigned long long), sl->stream)); }
		HIP_TRY(hipEventRecord(sl->kmer_done, sl->stream));
		HIP_TRY(hipStreamWaitEvent(gs, sl->kmer_done, 0));
		if(sl->append && sl->append_reset){ HIP_TRY(hipMemsetAsync(sl->ext_count, 0, sizeof(uint64_t), gs)); }      // a new list starts here
		// the stage's events ride on its kernel launches (StageEvents above); knob off: plain records around the stage
		const bool ride = g->ctx->tune.ext_launch_events != 0;
		StageEvents ge;
		if(ride){ ge.start = timing ? sl->ev[2] : nullptr; ge.stop = sl->gather_done; }
		else if(timing){ HIP_TRY(hipEventRecord(sl->ev[2], gs)); }
		unsigned long long *hit_count = sl->append ? (unsigned long long*)sl->ext_count : (unsigned long long*)sl->d_counters;
		if((rc = launch_search_stage(sl, g, b, sl->lay, sl->threshold, sl->flags, d_hits, cap, hit_count, gs, ge))){ return rc; }
		if(!ride){ HIP_TRY(hipEventRecord(sl->gather_done, gs)); }
		HIP_TRY(hipStreamWaitEvent(sl->stream, sl->gather_done, 0));
		++sl->launches;
	}
	else if(sl->append && sl->append_reset){
		// nothing to search: an empty list all the same (on the gather stream, where the list's other searches append)
		hipStream_t gs = g->ctx->gather_stream;
		HIP_TRY(hipMemsetAsync(sl->ext_count, 0, sizeof(uint64_t), gs));
		HIP_TRY(hipEventRecord(sl->gather_done, gs));
		HIP_TRY(hipStreamWaitEvent(sl->stream, sl->gather_done, 0));
	}
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
	// The search's own list is returned ordered by (query, column): the gather kernels keep its run table (hit_sort.hip).
	// Entries per query: one per KiB-step of a row, plus what the kernels' tilings round up (and_kernel's VEC, the walk
	// form's balanced column tiles); the narrow kernels number their workgroups (fewer than queries).
	sl->n_runs = 0;
	sl->runs_per_query = 0;
	if(ext_hits == nullptr && ext_cap == 0 && b->n && g->num_columns){
		// (KiB-steps of a row, plus what the tilings round up: and_kernel's 1 / 2 / 4 vectors per lane pad to a multiple of four,
		// the walk form's balanced column tiles to tiles x steps < kib + tiles.  The narrow kernels number their WORKGROUPS,
		// fewer than queries.)
		const uint64_t kib = (g->stride/16 + WAVE - 1)/WAVE;
		const uint64_t walk_tile = (uint64_t)std::min<int64_t>(std::max<int64_t>(ctx->tune.walk_tile_kib, 1), 16);
		sl->runs_per_query = (uint32_t)(kib + (kib + walk_tile - 1)/walk_tile + 4);
		sl->n_runs = (uint64_t)b->n*sl->runs_per_query;
		if((rc = sl->runs.reserve(sl->n_runs*sizeof(unsigned long long)))){ return rc; }
	}

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

int collect_search_steps(Slot *sl, SearchOutcome *out);

// Second half: wait for the slot's stream, grow the hit buffer and re-run the search kernel if it
// overflowed (own buffer only), report counts and timings. Frees the slot.
// A collection that FAILS half-way may leave kernels of this search queued (the k-mer stage or the gather kernels of a
// re-run): the slot is released all the same, so both of its streams are drained first -- the batch's device blocks go
// back to the context's pool the moment the caller destroys it (kwage_batch_destroy no longer waits for the device), and
// nothing may still be reading them then.
int collect_search(Slot *sl, SearchOutcome *out)
{
	if(!sl->busy){ return fail(KWAGE_ERR_STATE, "no pending search in this slot"); }
	kwage_ctx *ctx = sl->g->ctx;
	const int rc = collect_search_steps(sl, out);
	if(rc){
		(void)hipStreamSynchronize(ctx->gather_stream);
		(void)hipStreamSynchronize(sl->stream);
		(void)hipGetLastError();
	}
	return rc;
}

int collect_search_steps(Slot *sl, SearchOutcome *out)
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
	if(timing && sl->launches){ HIP_TRY(hipEventElapsedTime(&out->search_ms, sl->ev[2], sl->gather_done)); }      // (gather_done: the end of the stage's last kernel)
	return KWAGE_OK;
}

Slot *free_slot(kwage_ctx *ctx)
{
	for(int i = 0; i < 2; ++i){ if(!ctx->slot[i].busy){ return &ctx->slot[i]; } }
	return nullptr;
}

}  // namespace

namespace {

// The knobs' values at context creation: KWAGE_<NAME> for every name of TUNING_NAMES, plus the historic spelling
// KWAGE_HIT_SORT=host.
void tuning_from_environment(Tuning *t)
{
	for(const TuningName &tn : TUNING_NAMES){
		std::string env = "KWAGE_";
		for(const char *c = tn.name; *c; ++c){ env += (char)toupper((unsigned char)*c); }
		const char *e = getenv(env.c_str());
		if(e && *e){ t->*(tn.field) = strtoll(e, nullptr, 10); }
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

namespace {
__global__ __launch_bounds__(256) void count_nonzero_kernel(const uint32_t *p, uint64_t n, unsigned long long *out)
{
	unsigned long long c = 0;
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x*blockDim.x){ c += (p[i] != 0); }
	if(c){ atomicAdd(out, c); }
}
}  // namespace

extern "C" int kwage_ctx_scratch_nonzero(kwage_ctx *ctx, uint64_t out[5])
{
	if(!ctx || !out){ return fail(KWAGE_ERR_ARG, "kwage_ctx_scratch_nonzero: NULL argument"); }
	for(int i = 0; i < 2; ++i){ if(ctx->slot[i].busy){ return fail(KWAGE_ERR_STATE, "kwage_ctx_scratch_nonzero: a search is pending on this context"); } }
	int rc = set_device(ctx);
	if(rc){ return rc; }
	HIP_TRY(hipStreamSynchronize(ctx->gather_stream));
	HIP_TRY(hipStreamSynchronize(ctx->slot[0].stream));
	HIP_TRY(hipStreamSynchronize(ctx->slot[1].stream));
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void**)&d, 5*sizeof(unsigned long long)));
	hipError_t e = hipMemsetAsync(d, 0, 5*sizeof(unsigned long long), ctx->stream);
	for(int k = 0; k < 2 && e == hipSuccess; ++k){
		Slot *sl = &ctx->slot[k];
		const DevBuf *bufs[5] = {&sl->walk_or, &sl->walk_done, &sl->band_or, &sl->band_state, &sl->cwalk_arrived};
		for(int j = 0; j < 5; ++j){
			const uint64_t n = bufs[j]->cap/sizeof(uint32_t);
			if(!n){ continue; }
			hipLaunchKernelGGL(count_nonzero_kernel, dim3((uint32_t)std::min<uint64_t>(2048, (n + 255)/256)), dim3(256), 0, ctx->stream,
			                   (const uint32_t*)bufs[j]->p, n, d + j);
		}
		e = hipGetLastError();
	}
	unsigned long long h[5] = {0, 0, 0, 0, 0};
	if(e == hipSuccess){ e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream); }
	if(e == hipSuccess){ e = hipStreamSynchronize(ctx->stream); }
	(void)hipFree(d);
	if(e != hipSuccess){ return fail(KWAGE_ERR_DEVICE, "kwage_ctx_scratch_nonzero: %s", hipGetErrorString(e)); }
	for(int j = 0; j < 5; ++j){ out[j] = h[j]; }
	return KWAGE_OK;
}

extern "C" int kwage_ctx_refine_stats(kwage_ctx *ctx, uint64_t out[8])
{
	if(!ctx || !out){ return fail(KWAGE_ERR_ARG, "kwage_ctx_refine_stats: NULL argument"); }
	for(int i = 0; i < 2; ++i){ if(ctx->slot[i].busy){ return fail(KWAGE_ERR_STATE, "kwage_ctx_refine_stats: a search is pending on this context"); } }
	int rc = set_device(ctx);
	if(rc){ return rc; }
	HIP_TRY(hipStreamSynchronize(ctx->gather_stream));
	for(int k = 0; k < 2; ++k){
		Slot *sl = &ctx->slot[k];
		uint32_t h[4] = {0, 0, 0, 0};
		if(sl->ref_counters.p){ HIP_TRY(hipMemcpy(h, sl->ref_counters.p, sizeof(h), hipMemcpyDeviceToHost)); }
		for(int j = 0; j < 3; ++j){ out[4*k + j] = std::min<uint64_t>((uint64_t)sl->ref_base[j] + h[j], sl->ref_cap[j]); }
		out[4*k + 3] = sl->ref_cap[2];
	}
	return KWAGE_OK;
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
		HIP_TRY(hipEventCreateWithFlags(&sl->kmer_done, hipEventDisableTiming));
		HIP_TRY(hipEventCreate(&sl->gather_done));           // (carries the gather stage's end time too)
	}
	HIP_TRY(hipStreamCreateWithFlags(&ctx->gather_stream, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
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
	if(ctx->gather_stream){ (void)hipStreamSynchronize(ctx->gather_stream); }
	for(int k = 0; k < 2; ++k){
		Slot *sl = &ctx->slot[k];
		if(sl->stream){ (void)hipStreamSynchronize(sl->stream); }
		sl->rows.release(); sl->tables.release(); sl->result.release();
		sl->partial.release(); sl->h_stage.release(); sl->sort_scratch.release(); sl->runs.release();
		sl->walk_or.release(); sl->walk_done.release();
		sl->band_rows.release(); sl->band_loc.release(); sl->band_or.release(); sl->band_state.release(); sl->cwalk_slab.release(); sl->cwalk_arrived.release();
		sl->ref_counters.release(); sl->ref_clusters.release(); sl->ref_masks.release(); sl->ref_units.release(); sl->ref_slab.release();
		sl->trunc_dev.release(); sl->trunc_host.release();
		for(int i = 0; i < 4; ++i){ if(sl->ev[i]){ (void)hipEventDestroy(sl->ev[i]); } }
		if(sl->kmer_done){ (void)hipEventDestroy(sl->kmer_done); }
		if(sl->gather_done){ (void)hipEventDestroy(sl->gather_done); }
		if(sl->stream){ (void)hipStreamDestroy(sl->stream); }
	}
	if(ctx->gather_stream){ (void)hipStreamDestroy(ctx->gather_stream); }
	if(ctx->upload_stream){ (void)hipStreamSynchronize(ctx->upload_stream); (void)hipStreamDestroy(ctx->upload_stream); }
	ctx->kmers.release();
	ctx->batch_pool.close();
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

extern "C" int kwage_device_fingerprint(kwage_ctx *ctx, char *buf, uint64_t len)
{
	if(!ctx || !buf || len == 0){ return fail(KWAGE_ERR_ARG, "kwage_device_fingerprint: NULL argument"); }
	int rc = set_device(ctx);
	if(rc){ return rc; }
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
	hipUUID id;
	memset(&id, 0, sizeof(id));
	(void)hipDeviceGetUuid(&id, ctx->device);
	char hex[2*sizeof(id.bytes) + 1];
	bool text = true;                  // (the runtime hands out sixteen hex DIGITS as characters: kept as they are)
	for(size_t i = 0; i < sizeof(id.bytes); ++i){ text = text && isalnum((unsigned char)id.bytes[i]); }
	if(text){ memcpy(hex, id.bytes, sizeof(id.bytes)); hex[sizeof(id.bytes)] = 0; }
	else{ for(size_t i = 0; i < sizeof(id.bytes); ++i){ snprintf(hex + 2*i, 3, "%02x", (unsigned)(unsigned char)id.bytes[i]); } }
	snprintf(buf, (size_t)len, "uuid=%s;name=%s;arch=%s;cus=%d;sclk_mhz=%d;mclk_mhz=%d;hbm_bus_bits=%d;pci=%04x:%02x:%02x",
	         hex, prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate/1000, prop.memoryClockRate/1000, prop.memoryBusWidth,
	         (unsigned)prop.pciDomainID, (unsigned)prop.pciBusID, (unsigned)prop.pciDeviceID);
	return KWAGE_OK;
}

extern "C" int kwage_sync(kwage_ctx *ctx)
{
	if(!ctx){ return fail(KWAGE_ERR_ARG, "kwage_sync: ctx is NULL"); }
	int rc = set_device(ctx);
	if(rc){ return rc; }
	HIP_TRY(hipStreamSynchronize(ctx->gather_stream));
	HIP_TRY(hipStreamSynchronize(ctx->slot[0].stream));
	HIP_TRY(hipStreamSynchronize(ctx->slot[1].stream));
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
	hipError_t e = ctx->batch_pool.take(std::max<uint64_t>(total, 16), (void**)&b->d_seqs, &b->cap_seqs);
	if(e == hipSuccess){ e = ctx->batch_pool.take(((size_t)n_queries + 1)*sizeof(uint64_t), (void**)&b->d_seq_off, &b->cap_seq_off); }
	// (the context's upload stream: a host that streams batches creates the next one while slot 0 holds a pending search)
	if(e == hipSuccess && total){ e = hipMemcpyAsync(b->d_seqs, seqs + offsets[0], total, hipMemcpyHostToDevice, ctx->upload_stream); }
	if(e == hipSuccess){ e = hipMemcpyAsync(b->d_seq_off, b->h_seq_off.data(), ((size_t)n_queries + 1)*sizeof(uint64_t), hipMemcpyHostToDevice, ctx->upload_stream); }
	if(e == hipSuccess){ e = hipStreamSynchronize(ctx->upload_stream); }
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
	// A search on this batch that has been collected is done with its device arrays; only one still pending would need
	// waiting for (a caller's mistake: collect first).  Nothing here waits for the DEVICE: the blocks go back to the
	// context's pool (hipFree would drain the software pipeline of a host that streams batches through the context).
	for(int k = 0; k < 2; ++k){
		if(b->ctx->slot[k].busy && b->ctx->slot[k].b == b){
			(void)hipStreamSynchronize(b->ctx->gather_stream);
			(void)hipStreamSynchronize(b->ctx->slot[k].stream);
		}
	}
	b->ctx->batch_pool.give(b->d_seqs, b->cap_seqs);
	b->ctx->batch_pool.give(b->d_seq_off, b->cap_seq_off);
	delete b;      // (its layouts give their device arrays back too)
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
		const bool host_sort = g->ctx->tune.hit_sort_host != 0;      // the host's sort of the raw list, kept for A/B runs and as the fallback
		// The list is ordered where it lies, without a sort: a prefix sum over the run table the gather kernels kept and one
		// copy of every run to its place (hit_sort.hip), into the slot's scratch block; the copy-back reads from there.
		const uint64_t scratch = hit_order_scratch_bytes(so.n_hits, sl->n_runs);
		kwage_hit *d_from = sl->d_hits;
		const uint64_t *d_total = nullptr;
		bool on_device = !host_sort && sl->n_runs
		                 && sl->sort_scratch.reserve(scratch) == KWAGE_OK
		                 && order_hits_by_runs(sl->stream, sl->d_hits, so.n_hits, sl->runs.p, sl->n_runs, sl->sort_scratch.p, sl->sort_scratch.cap, &d_from, &d_total) == KWAGE_OK;
		if(!on_device){
			(void)hipGetLastError();
			d_from = sl->d_hits;
			if(!host_sort){      // never silently: the list is still ordered, by the host, and that is slower
				fprintf(stderr, "[kwage_amd] no room for the ordered copy of the hit list (%llu bytes) beside the database: %llu hits ordered by the host\n",
				        (unsigned long long)scratch, (unsigned long long)so.n_hits);
			}
		}
		// (pieces only so that a knob can shrink them in tests: one copy is what the link likes)
		const uint64_t piece_kb = (uint64_t)std::max<int64_t>(g->ctx->tune.hit_copy_piece_kb, 0);
		const uint64_t piece = piece_kb ? std::max<uint64_t>(1, (piece_kb << 10)/sizeof(kwage_hit)) : so.n_hits;
		hipError_t e = hipSuccess;
		for(uint64_t at = 0; at < so.n_hits && e == hipSuccess; at += piece){
			const uint64_t m = std::min(piece, so.n_hits - at);
			e = hipMemcpyAsync(hits + at, d_from + at, m*sizeof(kwage_hit), hipMemcpyDeviceToHost, sl->stream);
		}
		uint64_t placed = so.n_hits;       // records the run table accounts for
		if(e == hipSuccess && on_device){ e = hipMemcpyAsync(&placed, d_total, sizeof(uint64_t), hipMemcpyDeviceToHost, sl->stream); }
		if(e == hipSuccess){ e = hipStreamSynchronize(sl->stream); }
		if(e != hipSuccess){
			(void)hipStreamSynchronize(sl->stream);
			delete rs;
			return fail(KWAGE_ERR_DEVICE, "kwage_search: copying results failed: %s", hipGetErrorString(e));
		}
		if(placed != so.n_hits){
			delete rs;
			return fail(KWAGE_ERR_STATE, "kwage_search: the run table accounts for %llu of %llu hit records", (unsigned long long)placed, (unsigned long long)so.n_hits);
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

extern "C" int kwage_search_poll(kwage_pending *p)
{
	if(!p || !p->sl){ fail(KWAGE_ERR_ARG, "kwage_search_poll: NULL argument"); return -1; }
	if(set_device(p->sl->g->ctx)){ return -1; }
	const hipError_t e = hipStreamQuery(p->sl->stream);      // (the slot's stream ends with the copy-back, which waits for the gather stage)
	if(e == hipSuccess){ return 1; }
	(void)hipGetLastError();
	if(e == hipErrorNotReady){ return 0; }
	fail(KWAGE_ERR_DEVICE, "kwage_search_poll: %s", hipGetErrorString(e));
	return -1;
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
