// kwage_amd/csrc/kwage_node.cpp -- `kwage_node`: the kwage command line on every GPU of a node, ONE PROCESS PER GPU, the
// per-GPU hit lists concatenated on rank 0 by a variable-length gather over RCCL.
//
// Same options, same database files, same CSV / JSON bytes as `kwage` (and therefore as the reference): the option
// parser, query readers, hit records and report writers are the ones `kwage` uses (cli_common.hpp), so the two programs
// cannot drift apart.  What differs is how a node's GPUs are used:
//
//   kwage  (KWAGE_DEVICES=all)   one process, a host thread + context per GPU, hit lists merged in host memory
//   kwage_node                   one process per GPU (forked before anything touches a device), the sample (column)
//                                axis sharded by whole files (a file belongs to the rank that owns its middle column,
//                                the rule of kwage's node mode and of kwage_amd.distributed.partition_files), queries
//                                replicated, and per query batch ONE exchange:  every rank searches its groups with
//                                kwage_search_device_append_submit (all groups append to one device list, the records
//                                carry GLOBAL column numbers), ncclAllGather of the record counts, one grouped
//                                ncclSend / ncclRecv of the records in exact sizes to rank 0 (RCCL has no gatherv),
//                                rank 0 maps columns back to (file, column in file) and writes the report.
//
// This is the reference's only parallel axis (OpenMP over database files, kwage.cpp:76-87) and its critical-section
// merge (kwage.cpp:154-177) at node scale, as BASELINE's north star words it.  The library itself stays free of RCCL:
// only this program links it.  No row data ever crosses xGMI.
//
//   KWAGE_NODE_RANKS    number of ranks (default: the number of visible devices); rank r uses device r
//   KWAGE_NODE_PLAN     1: print the plan (groups, every rank's files, global column bases; under KWAGE_MAX_GROUP_BYTES also
//                       every rank's passes) as JSON and stop; no device is touched or counted -- the number of ranks must
//                       come from KWAGE_NODE_RANKS
//   KWAGE_NODE_STATS    1: rank 0 reports on stderr what the search phase took: wall, sum of gather-kernel time, exchange, filing
//   KWAGE_NODE_REHEARSE 1: rehearsal on a machine with fewer GPUs than ranks -- every rank uses device 0 and the records
//                       travel through a shared host segment instead of RCCL (which refuses two ranks on one device).
//                       The searches still run on the GPU; sharding, global numbering, gather and report are the same
//                       code.  For tests on one-GPU boxes, not for production.
//   KWAGE_EARLY_EXIT, KWAGE_BATCH_BASES, KWAGE_MAX_GROUP_BYTES   as for kwage
//   KWAGE_NODE_COMM_TIMEOUT_S   how long a rank waits for the communicator to come up (default 120 s): a rank whose peers never
//                       arrive exits non-zero instead of waiting forever, and the parent ends the others
// A database that does not fit its ranks' HBM is searched in PASSES, as kwage does it (kwage_main.cpp: units of whole files
// packed into what is free; the query sources are read once per pass; results are additive): every rank plans every
// rank's passes from the file headers and the ranks' budgets (one all-gather), so all of them make the same number of
// exchanges -- a rank whose files are done contributes empty lists.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <new>

#include "cli_common.hpp"

namespace {

#define NODE_HIP(call) do { hipError_t e_ = (call); if(e_ != hipSuccess){ throw string(#call " failed: ") + hipGetErrorString(e_); } } while(0)
#define NODE_NCCL(call) do { ncclResult_t r_ = (call); if(r_ != ncclSuccess){ throw string(#call " failed: ") + ncclGetErrorString(r_); } } while(0)

struct GroupKey {
	uint32_t kmer_len, num_hash, log_2_filter_len; int32_t hash_func;
	bool operator<(const GroupKey &o) const
	{
		return std::tie(kmer_len, num_hash, log_2_filter_len, hash_func) < std::tie(o.kmer_len, o.num_hash, o.log_2_filter_len, o.hash_func);
	}
};

// One rank's share of one group: its files and where each file's columns begin in the rank's matrix.  Every rank can
// compute every rank's share -- and its layout -- from the headers alone.
struct Share {
	vector<uint32_t> files;              // indices into the list of database files
	vector<uint64_t> first_column;       // of each file's block (blocks start at 16-byte boundaries)
	uint64_t span_columns = 0;           // next free column
};

Share share_of(const vector<uint32_t> &group_files, const vector<DbFileEntry> &files, size_t rank, size_t n_ranks)
{
	uint64_t total = 0, before = 0, span = 0;
	for(uint32_t fi : group_files){ total += files[fi].header.num_filter; }
	Share s;
	for(uint32_t fi : group_files){
		const uint64_t nf = files[fi].header.num_filter;
		const size_t owner = min<size_t>(n_ranks - 1, (size_t)(((long double)before + nf/2.0L)*n_ranks/max<uint64_t>(total, 1)));
		if(owner == rank){
			span = (span + 15)/16*16;
			s.files.push_back(fi);
			s.first_column.push_back(span*8);
			span += (nf + 7)/8;
		}
		before += nf;
	}
	s.span_columns = span*8;
	return s;
}

// What one rank holds resident in one pass: a span of whole files of one group, one matrix.  The records it produces carry
// base + column-in-matrix, which is the file's place in the GLOBAL numbering whatever the pass (a unit starts at a file,
// and files follow each other in a unit exactly as in the rank's whole share: blocks at 16-byte boundaries).
struct Unit {
	size_t gi = 0;                       // index into the groups
	vector<uint32_t> files;              // indices into the list of database files
	vector<uint64_t> first_column;       // of each file's block within the unit's matrix
	uint64_t span_columns = 0;
	uint64_t base = 0;                   // global number of the unit's column 0
	uint32_t kmer_len = 0;
	kwage_group *mine = nullptr;
};

// A file's columns in the global numbering of the hit records.
struct ColumnBlock { uint64_t first_global_column; uint32_t file_index, kmer_len; };

struct NodeGroup {
	GroupKey key;
	kwage_params params;
	vector<Share> share;                 // per rank
	vector<uint64_t> base;               // per rank: global number of the rank's column 0 of this group
};

// How rank 0 hands the communicator's unique id to the other ranks: anonymous shared memory mapped before the fork.
struct Bootstrap {
	ncclUniqueId id;
	volatile int ready = 0;
};

double now_seconds() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); }

// KWAGE_NODE_REHEARSE: what the ranks share instead of a communicator (mapped before the fork)
struct Rehearsal {
	pthread_barrier_t barrier;
	uint64_t capacity;                   // records
	uint64_t counts[64];
	kwage_hit *records() { return reinterpret_cast<kwage_hit*>(this + 1); }
};

// The node's plan: the database files grouped by parameters, every group's files dealt to the ranks (share_of), and
// the global column number of every rank's column 0 -- group after group, inside a group rank after rank.  A pure
// function of the file headers: every rank computes the same plan (and KWAGE_NODE_PLAN=1 prints it without a device).
vector<NodeGroup> plan_groups(const vector<DbFileEntry> &files, int n_ranks)
{
	map<GroupKey, vector<uint32_t> > by_key;
	for(size_t i = 0; i < files.size(); ++i){
		const kwage_db_header &h = files[i].header;
		by_key[GroupKey{h.kmer_len, h.num_hash, h.log_2_filter_len, h.hash_func}].push_back((uint32_t)i);
	}
	vector<NodeGroup> groups;
	uint64_t next_base = 0;
	for(const auto &kv : by_key){
		NodeGroup g;
		g.key = kv.first;
		g.params = kwage_params{kv.first.kmer_len, kv.first.num_hash, kv.first.log_2_filter_len, kv.first.hash_func};
		for(int r = 0; r < n_ranks; ++r){
			g.share.push_back(share_of(kv.second, files, (size_t)r, (size_t)n_ranks));
			g.base.push_back(next_base);
			next_base += g.share.back().span_columns;
		}
		groups.push_back(std::move(g));
	}
	if(next_base > (1ull << 32)){ throw "main: more than 2^32 columns in the database"; }
	return groups;
}

// The passes of one rank (kwage_main.cpp's packing): its files of every group, in order, cut into units that fit what is
// left of `budget` bytes in the current pass; a file that does not fit goes to the next pass -- unless the pass is still
// empty: then it is tried alone (and the allocation reports it if it really is too large).
vector<vector<Unit> > plan_passes(const vector<NodeGroup> &groups, const vector<DbFileEntry> &files, size_t rank, uint64_t budget)
{
	vector<vector<Unit> > passes(1);
	uint64_t pass_left = budget;
	for(size_t gi = 0; gi < groups.size(); ++gi){
		const Share &sh = groups[gi].share[rank];
		const uint64_t nrows = 1ull << groups[gi].params.log_2_filter_len;
		for(size_t m0 = 0; m0 < sh.files.size(); ){
			uint64_t span_bytes = 0;
			size_t m1 = m0;
			Unit u;
			while(m1 < sh.files.size()){
				const uint64_t at = (span_bytes + 15)/16*16;
				const uint64_t next = at + ((uint64_t)files[sh.files[m1]].header.num_filter + 7)/8;
				if(((next + 127)/128*128)*nrows > pass_left && (m1 > m0 || !passes.back().empty())){ break; }
				u.files.push_back(sh.files[m1]);
				u.first_column.push_back(at*8);
				span_bytes = next;
				++m1;
			}
			if(m1 > m0){
				u.gi = gi;
				u.span_columns = span_bytes*8;
				u.base = groups[gi].base[rank] + sh.first_column[m0];
				u.kmer_len = groups[gi].params.kmer_len;
				pass_left -= min(pass_left, ((span_bytes + 127)/128*128)*nrows);
				passes.back().push_back(std::move(u));
				m0 = m1;
			}
			if(m0 < sh.files.size()){ passes.emplace_back(); pass_left = budget; }
		}
	}
	if(passes.back().empty() && passes.size() > 1){ passes.pop_back(); }
	return passes;
}

// KWAGE_NODE_PLAN=1: print the plan for n_ranks ranks and stop -- no device is touched (tests; a dry run before a long job)
int print_plan(const vector<string> &db_paths, int n_ranks)
{
	try{
		vector<DbFileEntry> files(db_paths.size());
		for(size_t i = 0; i < db_paths.size(); ++i){
			files[i].path = db_paths[i];
			if(kwage_db_read_header(files[i].path.c_str(), &files[i].header) != KWAGE_OK){
				cerr << kwage_last_error() << endl;
				throw "main: I/O error";
			}
		}
		const vector<NodeGroup> groups = plan_groups(files, n_ranks);
		cout << "{\"ranks\": " << n_ranks << ", \"groups\": [";
		for(size_t gi = 0; gi < groups.size(); ++gi){
			const NodeGroup &g = groups[gi];
			cout << (gi ? ", " : "") << "{\"kmer_len\": " << g.key.kmer_len << ", \"num_hash\": " << g.key.num_hash << ", \"log_2_filter_len\": "
			     << g.key.log_2_filter_len << ", \"hash_func\": " << g.key.hash_func << ", \"shares\": [";
			for(size_t r = 0; r < g.share.size(); ++r){
				cout << (r ? ", " : "") << "{\"rank\": " << r << ", \"global_base\": " << g.base[r] << ", \"span_columns\": " << g.share[r].span_columns << ", \"files\": [";
				for(size_t f = 0; f < g.share[r].files.size(); ++f){
					cout << (f ? ", " : "") << "{\"path\": \"" << files[g.share[r].files[f]].path << "\", \"first_column\": " << g.share[r].first_column[f]
					     << ", \"num_filter\": " << files[g.share[r].files[f]].header.num_filter << "}";
				}
				cout << "]}";
			}
			cout << "]}";
		}
		cout << "]";
		// with a budget (KWAGE_MAX_GROUP_BYTES): every rank's passes as plan_passes cuts them, the ranks padded to the same number
		const uint64_t budget = env_u64("KWAGE_MAX_GROUP_BYTES", 0);
		if(budget){
			size_t n_passes = 1;
			vector<vector<vector<Unit> > > per_rank((size_t)n_ranks);
			for(int r = 0; r < n_ranks; ++r){
				per_rank[(size_t)r] = plan_passes(groups, files, (size_t)r, budget);
				n_passes = max(n_passes, per_rank[(size_t)r].size());
			}
			cout << ", \"budget\": " << budget << ", \"passes\": " << n_passes << ", \"rank_passes\": [";
			for(int r = 0; r < n_ranks; ++r){
				per_rank[(size_t)r].resize(n_passes);
				cout << (r ? ", " : "") << "[";
				for(size_t ps = 0; ps < n_passes; ++ps){
					cout << (ps ? ", " : "") << "[";
					for(size_t ui = 0; ui < per_rank[(size_t)r][ps].size(); ++ui){
						const Unit &u = per_rank[(size_t)r][ps][ui];
						cout << (ui ? ", " : "") << "{\"group\": " << u.gi << ", \"global_base\": " << u.base << ", \"span_columns\": " << u.span_columns << ", \"files\": [";
						for(size_t f = 0; f < u.files.size(); ++f){
							cout << (f ? ", " : "") << "{\"path\": \"" << files[u.files[f]].path << "\", \"first_column\": " << u.first_column[f] << "}";
						}
						cout << "]}";
					}
					cout << "]";
				}
				cout << "]";
			}
			cout << "]";
		}
		cout << "}" << endl;
	}
	catch(const char *error){
		cerr << "Caught the error " << error << endl;
		return EXIT_FAILURE;
	}
	return EXIT_SUCCESS;
}

// One rank's software pipeline over query batches (what kwage_amd/distributed.py's StepPipeline is for the Python host).
// A STEP = one query batch against every group this rank holds; all of a step's searches append to ONE device list
// (kwage_search_device_append_submit: the first search of a step zeroes the list's counter in stream order, every record
// carries a GLOBAL column number).  The context runs two searches at a time: begin() queues a step's searches, finish()
// completes the oldest step, feeding the slots as they come free -- so the next batch's first gather kernels are already
// running while the finished batch's hits are exchanged over RCCL and filed on rank 0.  THREE lists rotate: batch i+2 is
// queued before batch i is exchanged (the collective's kernel waits for wave slots behind the running gather kernel, and
// the device must have the next gather kernel queued while the host waits for it).
struct HitList {
	uint64_t *d_count = nullptr;         // the list's record counter (what the gather kernels add to)
	kwage_hit *d_hits = nullptr;
	uint64_t cap = 0;
};

struct Step {
	QueryBatch q;
	kwage_batch *b = nullptr;
	int list = 0;
	size_t searches = 0, done = 0;
	uint64_t n = 0;                      // records in the list (the running total the last search leaves)
	double kernel_ms = 0;
	map<uint32_t, vector<uint32_t> > nk; // rank 0: num_query_kmer per k-mer length, taken from this rank's own searches
	Findings *found = nullptr;
};

struct RankPipeline {
	kwage_ctx *ctx;
	vector<Unit> *units = nullptr;       // what is resident in the current pass
	int rank;
	float threshold;
	uint32_t flags;
	HitList lists[3];
	uint32_t *d_nk = nullptr;            // where a search leaves num_query_kmer for the host (rank 0)
	uint64_t nk_cap = 0;
	struct Todo { Step *step; size_t gi; bool first; };
	struct Flying { kwage_pending *p; Step *step; size_t gi; };
	deque<Todo> todo;
	deque<Flying> flying;
	int next_list = 0;

	RankPipeline(kwage_ctx *c, int r, float t, uint32_t f) : ctx(c), rank(r), threshold(t), flags(f)
	{
		for(HitList &l : lists){
			l.cap = 1u << 18;
			NODE_HIP(hipMalloc((void**)&l.d_count, sizeof(uint64_t)));
			NODE_HIP(hipMemset(l.d_count, 0, sizeof(uint64_t)));
			NODE_HIP(hipMalloc((void**)&l.d_hits, l.cap*sizeof(kwage_hit)));
		}
	}
	~RankPipeline()
	{
		for(Flying &f : flying){ uint64_t n = 0; (void)kwage_search_device_collect(f.p, &n, nullptr, nullptr); }
		for(HitList &l : lists){ (void)hipFree(l.d_count); (void)hipFree(l.d_hits); }
		if(d_nk){ (void)hipFree(d_nk); }
	}
	void submit(const Todo &t)
	{
		Unit &u = (*units)[t.gi];
		HitList &l = lists[t.step->list];
		kwage_pending *p = nullptr;
		check(kwage_search_device_append_submit(u.mine, t.step->b, threshold, flags | KWAGE_SEARCH_TIMING, l.d_hits, l.cap, l.d_count,
		                                        (uint32_t)u.base, t.first ? 1 : 0, &p));
		flying.push_back(Flying{p, t.step, t.gi});
	}
	void pump() { while(!todo.empty() && flying.size() < 2){ submit(todo.front()); todo.pop_front(); } }
	void collect_oldest()
	{
		Flying f = flying.front();
		flying.pop_front();
		Step *st = f.step;
		const uint32_t k = (*units)[f.gi].kmer_len;
		// rank 0 files the hits and needs num_query_kmer of every query: it comes with the search (one device-to-host copy
		// per distinct k-mer length and batch), not from a k-mer stage of its own
		const bool want_nk = (rank == 0) && st->q.size() && !st->nk.count(k);
		if(want_nk && st->q.size() > nk_cap){
			if(d_nk){ (void)hipFree(d_nk); d_nk = nullptr; }
			nk_cap = st->q.size() + st->q.size()/4;
			NODE_HIP(hipMalloc((void**)&d_nk, nk_cap*sizeof(uint32_t)));
		}
		uint64_t n = 0;
		float ms = 0;
		check(kwage_search_device_collect(f.p, &n, want_nk ? d_nk : nullptr, &ms));
		if(want_nk){
			vector<uint32_t> &nk = st->nk[k];
			nk.resize(st->q.size());
			NODE_HIP(hipMemcpy(nk.data(), d_nk, nk.size()*sizeof(uint32_t), hipMemcpyDeviceToHost));
		}
		st->n = n;
		st->kernel_ms += ms;
		++st->done;
	}
	// Wait for `stream` (the exchange's) WITHOUT leaving the device unfed: a collective queued behind a running gather
	// kernel can take as long as that kernel, and only two searches are queued at a time -- so while the exchange is
	// pending, searches that have finished are collected and the step's next ones submitted.
	// (No sleeping poll: the host always BLOCKS on something that has to complete anyway -- the oldest search in flight while
	// there is one (its collection frees a slot for the step's next search), the exchange's stream itself otherwise.  Filing
	// a batch a kernel's length later costs nothing: it happens beside the following batches' searches.)
	void wait_feeding(hipStream_t stream)
	{
		for(;;){
			const hipError_t e = hipStreamQuery(stream);
			if(e == hipSuccess){ return; }
			(void)hipGetLastError();
			if(e != hipErrorNotReady){ throw string("hipStreamQuery failed: ") + hipGetErrorString(e); }
			if(flying.empty()){ NODE_HIP(hipStreamSynchronize(stream)); return; }
			collect_oldest();
			pump();
		}
	}
	void begin(Step *st)
	{
		st->list = next_list;
		next_list = (next_list + 1) % 3;
		check(kwage_batch_create(ctx, st->q.bases.data(), st->q.offsets.data(), (uint32_t)st->q.size(), &st->b));
		bool first = true;
		for(size_t ui = 0; ui < units->size(); ++ui){
			todo.push_back(Todo{st, ui, first});
			first = false;
			++st->searches;
		}
		// (a rank with nothing resident in this pass still takes part in the exchange: an empty list)
		if(st->searches == 0){ NODE_HIP(hipMemset(lists[st->list].d_count, 0, sizeof(uint64_t))); }
		pump();
	}
	// Complete every search of `st` (the oldest open step).  A list that outgrew its buffer is redone after growing it.
	void finish(Step *st)
	{
		while(st->done < st->searches){ collect_oldest(); pump(); }
		HitList &l = lists[st->list];
		if(st->n > l.cap){
			while(!flying.empty()){ collect_oldest(); }       // the next step's searches write the OTHER list: let them finish
			while(st->n > l.cap){
				(void)hipFree(l.d_hits);
				l.cap = st->n + st->n/4;
				NODE_HIP(hipMalloc((void**)&l.d_hits, l.cap*sizeof(kwage_hit)));
				st->done = 0;
				st->kernel_ms = 0;
				bool first = true;
				for(size_t ui = 0; ui < units->size(); ++ui){
					submit(Todo{st, ui, first});
					first = false;
					collect_oldest();
				}
			}
			pump();
		}
	}
};

int run_rank(int rank, int n_ranks, Bootstrap *boot, const Cli &cli, const vector<string> &db_paths, Rehearsal *rehearsal)
{
	try{
		const time_t started = time(nullptr);
		ofstream fout;
		if(rank == 0 && !cli.output_path.empty()){
			fout.open(cli.output_path.c_str());
			if(!fout){
				cerr << "Unable to open " << cli.output_path << " for writing" << endl;
				return EXIT_FAILURE;
			}
		}
		ostream &out = fout.is_open() ? fout : cout;

		// ---- headers (every rank) and metadata (rank 0, read on first use) of every database file ---------------------
		vector<DbFileEntry> files(db_paths.size());
		vector<DbInfo> infos(rank == 0 ? db_paths.size() : 0);
		for(size_t i = 0; i < db_paths.size(); ++i){
			files[i].path = db_paths[i];
			if(kwage_db_read_header(files[i].path.c_str(), &files[i].header) != KWAGE_OK){
				if(rank == 0){ cerr << kwage_last_error() << endl; }
				throw "main: I/O error";
			}
			string err;
			if(rank == 0 && !infos[i].open(files[i].path, err)){
				cerr << err << endl;
				throw "main: Unable to read header";
			}
		}
		vector<NodeGroup> groups = plan_groups(files, n_ranks);
		// where every file's columns begin in the global numbering (ascending: groups, ranks and files are numbered in order)
		vector<ColumnBlock> blocks;
		for(const NodeGroup &g : groups){
			for(size_t r = 0; r < (size_t)n_ranks; ++r){
				for(size_t f = 0; f < g.share[r].files.size(); ++f){
					blocks.push_back(ColumnBlock{g.base[r] + g.share[r].first_column[f], g.share[r].files[f], g.params.kmer_len});
				}
			}
		}

		// ---- this rank's device, its communicator, its matrices ---------------------------------------------------------
		kwage_ctx *ctx = nullptr;
		check(kwage_init(rehearsal ? 0 : rank, &ctx));
		one_shot_placement(ctx);
		ncclComm_t comm = nullptr;
		hipStream_t stream;
		NODE_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
		if(!rehearsal){
			// stdout is the report: whatever RCCL prints while a communicator comes up (its version banner under
			// NCCL_DEBUG=VERSION goes to stdout whatever NCCL_DEBUG_FILE says) is sent to stderr instead
			cout.flush(); fflush(stdout);
			const int report_fd = dup(STDOUT_FILENO);
			if(report_fd < 0 || dup2(STDERR_FILENO, STDOUT_FILENO) < 0){ throw string("cannot redirect stdout"); }
			// the communicator's id travels through memory the ranks have shared since before the fork: nothing another
			// user of the machine could plant or read (round 3 used a file under /tmp)
			ncclUniqueId id;
			if(rank == 0){
				NODE_NCCL(ncclGetUniqueId(&id));
				memcpy(&boot->id, &id, sizeof(id));
				__sync_synchronize();
				boot->ready = 1;
			}
			else{
				int tries = 0;
				while(!boot->ready && tries++ < 60000){ usleep(1000); }
				if(!boot->ready){ throw string("no RCCL unique id from rank 0"); }
				__sync_synchronize();
				memcpy(&id, &boot->id, sizeof(id));
			}
			// ncclCommInitRank has no timeout of its own: a rank whose peers never arrive (one of them failed before this point)
			// would wait forever.  It runs on a thread; if it is not back in time the process exits non-zero without
			// returning into RCCL, and the parent ends the other ranks.
			ncclResult_t comm_up = ncclInternalError;
			{
				struct Up { mutex mu; condition_variable cv; bool done = false; ncclResult_t rc = ncclInternalError; ncclComm_t comm = nullptr; };
				shared_ptr<Up> up = make_shared<Up>();
				const int device = rank;
				thread([up, n_ranks, id, rank, device]() {
					(void)hipSetDevice(device);
					ncclComm_t c = nullptr;
					const ncclResult_t rc = ncclCommInitRank(&c, n_ranks, id, rank);
					lock_guard<mutex> lk(up->mu);
					up->rc = rc; up->comm = c; up->done = true;
					up->cv.notify_all();
				}).detach();
				unique_lock<mutex> lk(up->mu);
				const uint64_t limit_s = env_u64("KWAGE_NODE_COMM_TIMEOUT_S", 120);
				if(!up->cv.wait_for(lk, chrono::seconds(limit_s), [&]{ return up->done; })){
					cerr << "Caught the error rank " << rank << ": the RCCL communicator did not come up within " << limit_s << " s (a peer is missing)" << endl;
					fflush(nullptr);
					_exit(EXIT_FAILURE);
				}
				comm_up = up->rc;
				comm = up->comm;
			}
			fflush(stdout);
			dup2(report_fd, STDOUT_FILENO);
			close(report_fd);
			if(comm_up != ncclSuccess){ throw string("ncclCommInitRank failed: ") + ncclGetErrorString(comm_up); }
		}

		// ---- the exchange's buffers: every rank's count, rank 0's gathered list (device + pinned host) ------------------
		const uint32_t flags = env_u64("KWAGE_EARLY_EXIT", 1) ? KWAGE_SEARCH_EARLY_EXIT : 0u;
		const uint64_t max_batch_bases = env_u64("KWAGE_BATCH_BASES", 64ull << 20);
		const bool stats = env_u64("KWAGE_NODE_STATS", 0) != 0;
		uint64_t all_cap = 0, host_cap = 0;
		uint64_t *d_counts = nullptr, *h_counts = nullptr;
		kwage_hit *d_all = nullptr, *h_all = nullptr;
		NODE_HIP(hipMalloc((void**)&d_counts, (size_t)n_ranks*sizeof(uint64_t)));
		NODE_HIP(hipHostMalloc((void**)&h_counts, (size_t)n_ranks*sizeof(uint64_t)));
		double sum_kernel_ms = 0, t_exchange = 0, t_file = 0;
		uint64_t n_batches = 0, n_records = 0;
		const double t_search0 = now_seconds();
		double t_first_begin = 0;

		// ---- the passes: every rank's budget to everyone, then every rank plans every rank's passes ---------------------------
		vector<vector<Unit> > my_passes;
		size_t n_passes = 1;
		{
			uint64_t budget = env_u64("KWAGE_MAX_GROUP_BYTES", 0);
			if(budget == 0){
				uint64_t free_b = 0, total_b = 0;
				check(kwage_mem_info(ctx, &free_b, &total_b));
				budget = free_b - free_b/8;              // leave room for staging buffers, row indices, hits
				if(rehearsal){ budget /= (uint64_t)n_ranks; }      // (the rehearsed ranks share one device)
			}
			vector<uint64_t> budgets((size_t)n_ranks, budget);
			if(rehearsal){
				rehearsal->counts[rank] = budget;
				pthread_barrier_wait(&rehearsal->barrier);
				for(int r = 0; r < n_ranks; ++r){ budgets[(size_t)r] = rehearsal->counts[r]; }
				pthread_barrier_wait(&rehearsal->barrier);
			}
			else if(n_ranks > 1){
				uint64_t *d_mine = nullptr;
				NODE_HIP(hipMalloc((void**)&d_mine, sizeof(uint64_t)));
				NODE_HIP(hipMemcpy(d_mine, &budget, sizeof(uint64_t), hipMemcpyHostToDevice));
				NODE_NCCL(ncclAllGather(d_mine, d_counts, 1, ncclUint64, comm, stream));
				NODE_HIP(hipMemcpyAsync(h_counts, d_counts, (size_t)n_ranks*sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
				NODE_HIP(hipStreamSynchronize(stream));
				for(int r = 0; r < n_ranks; ++r){ budgets[(size_t)r] = h_counts[r]; }
				(void)hipFree(d_mine);
			}
			for(int r = 0; r < n_ranks; ++r){
				vector<vector<Unit> > pr = plan_passes(groups, files, (size_t)r, budgets[(size_t)r]);
				n_passes = max(n_passes, pr.size());
				if(r == rank){ my_passes = std::move(pr); }
			}
			my_passes.resize(n_passes);
			if(env_u64("KWAGE_VERBOSE", 0) || stats){
				size_t nu = 0;
				for(const auto &ps : my_passes){ nu += ps.size(); }
				// (one write: the ranks share stderr)
				fprintf(stderr, "[kwage_node] rank %d: budget %llu bytes per pass, %zu pass(es), %zu unit(s) of its own\n", rank, (unsigned long long)budget, n_passes, nu);
			}
		}
		double t_parse_wait = 0;

		{
		RankPipeline pipe(ctx, rank, cli.threshold, flags);

		// The gatherv of one finished step: counts to everyone (one ncclAllGather straight from the lists' counter words),
		// records in exact sizes to rank 0 (one grouped ncclSend / ncclRecv; RCCL has no gatherv); rank 0 files them.
		// The next step's searches are running on the device meanwhile.
		auto exchange_and_file = [&](Step *st) {
			const double t0 = now_seconds();
			const HitList &l = pipe.lists[st->list];
			const uint64_t n_mine = st->n;
			vector<uint64_t> counts((size_t)n_ranks);
			const kwage_hit *hits = nullptr;
			uint64_t total = 0;
			if(rehearsal){
				rehearsal->counts[rank] = n_mine;
				pthread_barrier_wait(&rehearsal->barrier);
				uint64_t at = 0;
				for(int r = 0; r < n_ranks; ++r){
					counts[(size_t)r] = rehearsal->counts[r];
					if(r < rank){ at += counts[(size_t)r]; }
					total += counts[(size_t)r];
				}
				if(total > rehearsal->capacity){ throw string("the rehearsal segment is too small for this hit list"); }
				if(n_mine){ NODE_HIP(hipMemcpy(rehearsal->records() + at, l.d_hits, n_mine*sizeof(kwage_hit), hipMemcpyDeviceToHost)); }
				pthread_barrier_wait(&rehearsal->barrier);
				if(rank == 0 && total){
					if(total > host_cap){
						if(h_all){ (void)hipHostFree(h_all); }
						host_cap = total + total/4;
						NODE_HIP(hipHostMalloc((void**)&h_all, host_cap*sizeof(kwage_hit)));
					}
					memcpy(h_all, rehearsal->records(), total*sizeof(kwage_hit));
					hits = h_all;
				}
				pthread_barrier_wait(&rehearsal->barrier);            // (the segment is free for the next batch)
			}
			else{
				// (the step's searches have been collected: the counter word holds the list's final count)
				NODE_NCCL(ncclAllGather(l.d_count, d_counts, 1, ncclUint64, comm, stream));
				NODE_HIP(hipMemcpyAsync(h_counts, d_counts, (size_t)n_ranks*sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
				pipe.wait_feeding(stream);
				for(int r = 0; r < n_ranks; ++r){ counts[(size_t)r] = h_counts[r]; total += h_counts[r]; }
				if(counts[(size_t)rank] != n_mine){ throw string("the list's counter word disagrees with the count the search returned"); }
				if(rank == 0 && total > all_cap){
					if(d_all){ (void)hipFree(d_all); }
					all_cap = total + total/4;
					NODE_HIP(hipMalloc((void**)&d_all, all_cap*sizeof(kwage_hit)));
				}
				if(rank == 0 && total > host_cap){
					if(h_all){ (void)hipHostFree(h_all); }
					host_cap = total + total/4;
					NODE_HIP(hipHostMalloc((void**)&h_all, host_cap*sizeof(kwage_hit)));
				}
				{
					NODE_NCCL(ncclGroupStart());
					if(rank == 0){
						uint64_t at = counts[0];
						for(int r = 1; r < n_ranks; ++r){
							if(counts[(size_t)r]){ NODE_NCCL(ncclRecv(d_all + at, counts[(size_t)r]*3, ncclUint32, r, comm, stream)); }
							at += counts[(size_t)r];
						}
					}
					else if(n_mine){
						NODE_NCCL(ncclSend(l.d_hits, n_mine*3, ncclUint32, 0, comm, stream));
					}
					NODE_NCCL(ncclGroupEnd());
				}
				if(rank == 0 && total){
					// rank 0's own records go to the host straight from its list, the other ranks' from the gathered block
					if(n_mine){ NODE_HIP(hipMemcpyAsync(h_all, l.d_hits, n_mine*sizeof(kwage_hit), hipMemcpyDeviceToHost, stream)); }
					if(total > n_mine){ NODE_HIP(hipMemcpyAsync(h_all + n_mine, d_all + n_mine, (total - n_mine)*sizeof(kwage_hit), hipMemcpyDeviceToHost, stream)); }
					hits = h_all;
				}
				pipe.wait_feeding(stream);      // (a sender's list is refilled two steps on: the send has left it)
			}
			const double t1 = now_seconds();
			t_exchange += t1 - t0;

			if(rank == 0 && total){
				kwage_sort_hits(h_all, total);                     // by (query, global column): groups, ranks and files in order
				const QueryBatch &q = st->q;
				// num_query_kmer depends on the k-mer length only; lengths of which this rank holds no group (it searched none
				// of them) are the one case that still needs a k-mer stage of its own
				for(const NodeGroup &g : groups){
					if(st->nk.count(g.params.kmer_len)){ continue; }
					while(!pipe.flying.empty()){ pipe.collect_oldest(); pipe.pump(); }      // (kwage_hash_batch wants the context's slots idle)
					vector<uint64_t> off(q.size() + 1);
					vector<uint32_t> nk(q.size());
					check(kwage_hash_batch(ctx, &g.params, st->b, off.data(), nk.data(), nullptr, nullptr));
					st->nk[g.params.kmer_len] = std::move(nk);
				}
				pipe.pump();
				Findings &found = *st->found;
				for(uint64_t i = 0; i < total; ){
					const uint32_t qi = hits[i].query;
					const size_t qid = q.ids[qi];
					vector<Match> &dst = found.by_query.try_emplace(found.by_query.end(), qid)->second;
					if(!q.deflines.empty()){ found.defline.try_emplace(found.defline.end(), qid, q.deflines[qi]); }
					for(; i < total && hits[i].query == qi; ++i){
						// the last block that starts at or before the column
						const ColumnBlock &blk = *(upper_bound(blocks.begin(), blocks.end(), (uint64_t)hits[i].column,
						                                       [](uint64_t col, const ColumnBlock &b) { return col < b.first_global_column; }) - 1);
						Match m;
						m.num_kmers_found = hits[i].num_match;
						m.num_query_kmer = st->nk[blk.kmer_len][qi];
						m.file_index = blk.file_index;
						m.column = (uint32_t)(hits[i].column - blk.first_global_column);
						dst.push_back(m);
					}
				}
			}
			t_file += now_seconds() - t1;
			sum_kernel_ms += st->kernel_ms;
			n_records += total;
			++n_batches;
		};

		// batches stream through: batch i+1 is parsed (on its own thread) and its searches queued BEFORE batch i is
		// finished; batch i is exchanged and filed only AFTER batch i+2 has been queued (three lists rotate)
		unique_ptr<Step> open_step, done_step;             // begun, not finished yet; finished, not exchanged yet
		auto hand_over = [&](unique_ptr<Step> &st) {
			exchange_and_file(st.get());
			kwage_batch_destroy(st->b);
			st.reset();
		};
		auto run_source = [&](QuerySource &source, Findings &found) {
			PrefetchedQueries ahead(source, max_batch_bases);
			for(;;){
				unique_ptr<Step> st(new Step());
				const double t_fill = now_seconds();
				const bool more = ahead.fill(st->q, max_batch_bases);
				t_parse_wait += now_seconds() - t_fill;        // (every rank parses the query files, on a thread of its own: what the searches waited for it)
				if(!more){ break; }
				st->found = &found;
				if(t_first_begin == 0){ t_first_begin = now_seconds(); }
				pipe.begin(st.get());
				if(done_step){ hand_over(done_step); }
				if(open_step){ pipe.finish(open_step.get()); done_step = std::move(open_step); }
				open_step = std::move(st);
				open_step->q.bases = string();                      // resident on the device now
			}
		};
		Findings from_command_line_, from_files_;
		double t_load = 0;
		for(size_t pass = 0; pass < n_passes; ++pass){
			// ---- this pass's matrices (none: the rank's files are done -- it still takes part in every exchange) ------------
			vector<Unit> &units = my_passes[pass];
			const double t_l0 = now_seconds();
			for(Unit &u : units){
				NodeGroup &g = groups[u.gi];
				check(kwage_group_create(ctx, &g.params, u.span_columns, &u.mine));
				vector<const char*> paths;
				for(uint32_t fi : u.files){ paths.push_back(files[fi].path.c_str()); }
				vector<uint64_t> first(paths.size());
				check(kwage_group_add_db_files(u.mine, paths.data(), (uint32_t)paths.size(), first.data(), nullptr));
				if(first != u.first_column){ throw "main: the loaded layout differs from the planned one"; }
				check(kwage_group_finalize(u.mine));
			}
			t_load += now_seconds() - t_l0;
			pipe.units = &units;
			// the query sources are read once per pass (results are additive: a query's matches of this pass join those of the others)
			{
				CommandLineQueries typed(cli.query_seqs);
				run_source(typed, from_command_line_);
			}
			if(!cli.query_files.empty()){
				FileQueries disk(cli.query_files);
				run_source(disk, from_files_);
			}
			if(done_step){ hand_over(done_step); }
			if(open_step){ pipe.finish(open_step.get()); hand_over(open_step); }
			for(Unit &u : units){ if(u.mine){ kwage_group_destroy(u.mine); u.mine = nullptr; } }
		}
		const double t_search = now_seconds() - t_search0;
		if(stats && rank == 0){
			const double t_pipe = t_first_begin ? now_seconds() - t_first_begin : 0;
			fprintf(stderr, "[kwage_node] %llu batches, %llu records: first batch queued -> last batch filed %.3f ms wall (with the first batch's parsing %.3f ms); "
			                "sum of gather-kernel time (rank 0) %.3f ms = %.1f %% of it; exchange %.3f ms, filing %.3f ms (both beside the next batch's searches)\n",
			        (unsigned long long)n_batches, (unsigned long long)n_records, t_pipe*1e3, t_search*1e3,
			        sum_kernel_ms, t_pipe > 0 ? 100.0*sum_kernel_ms/(t_pipe*1e3) : 0.0, t_exchange*1e3, t_file*1e3);
			fprintf(stderr, "[kwage_node] %zu pass(es), loading %.3f s; waited %.3f ms for this rank's own query parser (every rank parses the query files on a prefetch thread)\n",
			        n_passes, t_load, t_parse_wait*1e3);
		}

		// (pipe is destroyed before the context)
		if(comm){ NODE_NCCL(ncclCommDestroy(comm)); }
		(void)hipFree(d_counts); (void)hipHostFree(h_counts);
		if(d_all){ (void)hipFree(d_all); }
		if(h_all){ (void)hipHostFree(h_all); }
		(void)hipStreamDestroy(stream);

		if(rank == 0){
			// order and report exactly as kwage does (kwage_main.cpp; reference kwage.cpp:191-315)
			for(Findings *f : {&from_command_line_, &from_files_}){
				for(auto &kv : f->by_query){
					sort(kv.second.begin(), kv.second.end(), [](const Match &a, const Match &b) {
						return (a.file_index != b.file_index) ? (a.file_index < b.file_index) : (a.column < b.column);
					});
					sort(kv.second.begin(), kv.second.end(), [](const Match &a, const Match &b) { return a.num_kmers_found > b.num_kmers_found; });
				}
			}
			unique_ptr<Report> report;
			if(cli.format == Cli::CSV){ report.reset(new CsvReport(out, infos)); }
			else{ report.reset(new JsonReport(out, cli.threshold, infos)); }
			report->begin(from_command_line_.by_query.size() + from_files_.by_query.size());
			for(const auto &kv : from_command_line_.by_query){ report->query("command line seq " + to_string(kv.first), kv.second); }
			for(const auto &kv : from_files_.by_query){ report->query(from_files_.defline[kv.first], kv.second); }
			report->end();
			cerr << "Search complete in " << (time(nullptr) - started) << " sec" << endl;
		}
		}
		kwage_shutdown(ctx);
	}
	catch(const char *error){
		cerr << "Caught the error " << error << endl;
		return EXIT_FAILURE;
	}
	catch(const string &error){
		cerr << "Caught the error " << error << endl;
		return EXIT_FAILURE;
	}
	catch(...){
		cerr << "Caught an unhandled error" << endl;
		return EXIT_FAILURE;
	}
	return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char *argv[])
{
	setenv("GPU_MAX_HW_QUEUES", "8", 0);
	setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);          // dmabuf IPC (what RCCL needs on this host driver)
	setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);           // stdout is the report: RCCL's debug lines (NCCL_DEBUG) go to stderr
	Cli cli;
	vector<string> db_paths;
	try{
		if(!read_command_line(argc, argv, cli, db_paths)){ return EXIT_SUCCESS; }
	}
	catch(const char *error){
		cerr << "Caught the error " << error << endl;
		return EXIT_FAILURE;
	}
	int n_ranks = (int)env_u64("KWAGE_NODE_RANKS", 0);
	// the plan is a function of the file headers and the rank count: it is printed before anything looks for a device
	if(env_u64("KWAGE_NODE_PLAN", 0)){
		if(n_ranks < 1 || n_ranks > 64){
			cerr << "kwage_node: KWAGE_NODE_PLAN needs the number of ranks in KWAGE_NODE_RANKS (1..64); no device is asked" << endl;
			return EXIT_FAILURE;
		}
		return print_plan(db_paths, n_ranks);
	}
	// the parent touches no GPU: the devices are counted by a short-lived child, the ranks forked before any HIP call
	if(n_ranks <= 0){ n_ranks = device_count_in_child(); }
	if(n_ranks < 1 || n_ranks > 64){
		cerr << "kwage_node: no usable device count (" << n_ranks << "); set KWAGE_NODE_RANKS" << endl;
		return EXIT_FAILURE;
	}
	Rehearsal *rehearsal = nullptr;
	if(env_u64("KWAGE_NODE_REHEARSE", 0)){
		const uint64_t capacity = env_u64("KWAGE_NODE_REHEARSE_RECORDS", 64ull << 20);
		void *seg = mmap(nullptr, sizeof(Rehearsal) + capacity*sizeof(kwage_hit), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if(seg == MAP_FAILED){ perror("mmap"); return EXIT_FAILURE; }
		rehearsal = static_cast<Rehearsal*>(seg);
		rehearsal->capacity = capacity;
		pthread_barrierattr_t shared;
		pthread_barrierattr_init(&shared);
		pthread_barrierattr_setpshared(&shared, PTHREAD_PROCESS_SHARED);
		pthread_barrier_init(&rehearsal->barrier, &shared, (unsigned)n_ranks);
	}
	// where rank 0 leaves the communicator's id for the others: anonymous memory shared with the ranks through the fork
	void *bseg = mmap(nullptr, sizeof(Bootstrap), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
	if(bseg == MAP_FAILED){ perror("mmap"); return EXIT_FAILURE; }
	Bootstrap *boot = new (bseg) Bootstrap();
	vector<pid_t> kids;
	int rc = EXIT_SUCCESS;
	for(int r = 0; r < n_ranks; ++r){
		const pid_t pid = fork();
		if(pid < 0){
			// the ranks already started would wait for this one in the communicator's set-up or the first exchange: end them
			perror("fork");
			rc = EXIT_FAILURE;
			for(pid_t k : kids){ kill(k, SIGTERM); }
			break;
		}
		if(pid == 0){
			const int rank_rc = run_rank(r, n_ranks, boot, cli, db_paths, rehearsal);
			cout.flush();
			fflush(nullptr);
			_exit(rank_rc);
		}
		kids.push_back(pid);
	}
	// a rank that fails would leave the others waiting in the exchange: end them too
	for(size_t left = kids.size(); left; ){
		int st = 0;
		const pid_t done = waitpid(-1, &st, 0);
		if(done < 0){ break; }
		vector<pid_t>::iterator it = find(kids.begin(), kids.end(), done);
		if(it == kids.end()){ continue; }
		*it = 0;
		--left;
		if((!WIFEXITED(st) || WEXITSTATUS(st) != 0) && rc == EXIT_SUCCESS){
			rc = EXIT_FAILURE;
			for(pid_t k : kids){ if(k > 0){ kill(k, SIGTERM); } }
		}
	}
	(void)munmap(bseg, sizeof(Bootstrap));
	return rc;
}
