// kwage_amd/csrc/kwage_node.cpp -- `kwage_node`: the kwage command line on every GPU of a node, ONE PROCESS PER GPU, the
// per-GPU hit lists concatenated on rank 0 by a variable-length gather over RCCL.
//
// Same options, same database files, same CSV / JSON bytes as `kwage` (and therefore as the reference): the option
// parser, query readers, hit filing and report writers ARE kwage_main.cpp's -- this translation unit includes that file
// with its `main` renamed, so the two programs cannot drift apart.  What differs is how a node's GPUs are used:
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
//   KWAGE_NODE_PLAN     1: print the plan (groups, every rank's files, global column bases) as JSON and stop; no device is touched
//   KWAGE_NODE_REHEARSE 1: rehearsal on a machine with fewer GPUs than ranks -- every rank uses device 0 and the records
//                       travel through a shared host segment instead of RCCL (which refuses two ranks on one device).
//                       The searches still run on the GPU; sharding, global numbering, gather and report are the same
//                       code.  For tests on one-GPU boxes, not for production.
//   KWAGE_EARLY_EXIT, KWAGE_BATCH_BASES   as for kwage
// Every parameter group must fit its ranks' HBM in one pass (kwage itself also handles databases that do not).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <pthread.h>
#include <sys/mman.h>

#define main kwage_single_process_main        // kwage_main.cpp's main(): not used here, but its parts are
#include "kwage_main.cpp"
#undef main

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

// A file's columns in the global numbering of the hit records.
struct ColumnBlock { uint64_t first_global_column; uint32_t file_index, kmer_len; };

struct NodeGroup {
	GroupKey key;
	kwage_params params;
	vector<Share> share;                 // per rank
	vector<uint64_t> base;               // per rank: global number of the rank's column 0 of this group
	kwage_group *mine = nullptr;         // this rank's matrix (null: no file of the group here)
};

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
		cout << "]}" << endl;
	}
	catch(const char *error){
		cerr << "Caught the error " << error << endl;
		return EXIT_FAILURE;
	}
	return EXIT_SUCCESS;
}

int run_rank(int rank, int n_ranks, const string &id_path, const Cli &cli, const vector<string> &db_paths, Rehearsal *rehearsal)
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
		NODE_HIP(hipStreamCreate(&stream));
		if(!rehearsal){
			// stdout is the report: whatever RCCL prints while a communicator comes up (its version banner under
			// NCCL_DEBUG=VERSION goes to stdout whatever NCCL_DEBUG_FILE says) is sent to stderr instead
			cout.flush(); fflush(stdout);
			const int report_fd = dup(STDOUT_FILENO);
			if(report_fd < 0 || dup2(STDERR_FILENO, STDOUT_FILENO) < 0){ throw string("cannot redirect stdout"); }
			ncclUniqueId id;
			if(rank == 0){
				NODE_NCCL(ncclGetUniqueId(&id));
				const string tmp = id_path + ".tmp";
				FILE *f = fopen(tmp.c_str(), "wb");
				if(!f || fwrite(&id, sizeof(id), 1, f) != 1){ throw string("cannot write ") + tmp; }
				fclose(f);
				rename(tmp.c_str(), id_path.c_str());
			}
			else{
				FILE *f = nullptr;
				for(int tries = 0; tries < 6000 && !(f = fopen(id_path.c_str(), "rb")); ++tries){ usleep(10000); }
				if(!f || fread(&id, sizeof(id), 1, f) != 1){ throw string("no RCCL unique id from rank 0"); }
				fclose(f);
			}
			const ncclResult_t comm_up = ncclCommInitRank(&comm, n_ranks, id, rank);
			fflush(stdout);
			dup2(report_fd, STDOUT_FILENO);
			close(report_fd);
			if(comm_up != ncclSuccess){ throw string("ncclCommInitRank failed: ") + ncclGetErrorString(comm_up); }
		}

		for(NodeGroup &g : groups){
			const Share &s = g.share[(size_t)rank];
			if(s.files.empty()){ continue; }
			check(kwage_group_create(ctx, &g.params, s.span_columns, &g.mine));
			vector<const char*> paths;
			for(uint32_t fi : s.files){ paths.push_back(files[fi].path.c_str()); }
			vector<uint64_t> first(paths.size());
			check(kwage_group_add_db_files(g.mine, paths.data(), (uint32_t)paths.size(), first.data(), nullptr));
			if(first != s.first_column){ throw "main: the loaded layout differs from the planned one"; }
			check(kwage_group_finalize(g.mine));
		}

		// ---- the exchange's buffers: this rank's list (counter word + records), every rank's count, rank 0's gathered list --
		const uint32_t flags = env_u64("KWAGE_EARLY_EXIT", 1) ? KWAGE_SEARCH_EARLY_EXIT : 0u;
		const uint64_t max_batch_bases = env_u64("KWAGE_BATCH_BASES", 64ull << 20);
		uint64_t cap = 1u << 18, all_cap = 0;
		uint64_t *d_count = nullptr, *d_counts = nullptr;
		kwage_hit *d_hits = nullptr, *d_all = nullptr;
		NODE_HIP(hipMalloc((void**)&d_count, sizeof(uint64_t)));
		NODE_HIP(hipMalloc((void**)&d_counts, (size_t)n_ranks*sizeof(uint64_t)));
		NODE_HIP(hipMalloc((void**)&d_hits, cap*sizeof(kwage_hit)));

		Findings from_command_line, from_files;
		auto search_batch = [&](const QueryBatch &q, Findings &found) {
			kwage_batch *b = nullptr;
			check(kwage_batch_create(ctx, q.bases.data(), q.offsets.data(), (uint32_t)q.size(), &b));
			// all of this rank's groups append to ONE list; records carry global column numbers
			uint64_t n_mine = 0;
			for(;;){
				NODE_HIP(hipMemset(d_count, 0, sizeof(uint64_t)));
				bool first_search = true;
				for(NodeGroup &g : groups){
					if(!g.mine){ continue; }
					kwage_pending *p = nullptr;
					check(kwage_search_device_append_submit(g.mine, b, cli.threshold, flags, d_hits, cap, d_count, (uint32_t)g.base[(size_t)rank],
					                                        first_search ? 1 : 0, &p));
					check(kwage_search_device_collect(p, &n_mine, nullptr, nullptr));
					first_search = false;
				}
				if(n_mine <= cap){ break; }
				(void)hipFree(d_hits);                               // rare: the list outgrew its buffer; grow and search again
				cap = n_mine + n_mine/4;
				NODE_HIP(hipMalloc((void**)&d_hits, cap*sizeof(kwage_hit)));
			}
			// the gatherv: counts to everyone, records in exact sizes to rank 0
			vector<uint64_t> counts((size_t)n_ranks);
			vector<kwage_hit> hits;
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
				if(n_mine){ NODE_HIP(hipMemcpy(rehearsal->records() + at, d_hits, n_mine*sizeof(kwage_hit), hipMemcpyDeviceToHost)); }
				pthread_barrier_wait(&rehearsal->barrier);
				if(rank == 0){ hits.assign(rehearsal->records(), rehearsal->records() + total); }
				pthread_barrier_wait(&rehearsal->barrier);            // (the segment is free for the next batch)
			}
			else{
				NODE_HIP(hipMemcpy(d_counts + rank, &n_mine, sizeof(uint64_t), hipMemcpyHostToDevice));
				NODE_NCCL(ncclAllGather(d_counts + rank, d_counts, 1, ncclUint64, comm, stream));
				NODE_HIP(hipStreamSynchronize(stream));
				NODE_HIP(hipMemcpy(counts.data(), d_counts, counts.size()*sizeof(uint64_t), hipMemcpyDeviceToHost));
				for(uint64_t c : counts){ total += c; }
				if(rank == 0 && total > all_cap){
					if(d_all){ (void)hipFree(d_all); }
					all_cap = total + total/4;
					NODE_HIP(hipMalloc((void**)&d_all, all_cap*sizeof(kwage_hit)));
				}
				NODE_NCCL(ncclGroupStart());
				if(rank == 0){
					uint64_t at = counts[0];
					for(int r = 1; r < n_ranks; ++r){
						if(counts[(size_t)r]){ NODE_NCCL(ncclRecv(d_all + at, counts[(size_t)r]*3, ncclUint32, r, comm, stream)); }
						at += counts[(size_t)r];
					}
				}
				else if(n_mine){
					NODE_NCCL(ncclSend(d_hits, n_mine*3, ncclUint32, 0, comm, stream));
				}
				NODE_NCCL(ncclGroupEnd());
				if(rank == 0 && n_mine){ NODE_HIP(hipMemcpyAsync(d_all, d_hits, n_mine*sizeof(kwage_hit), hipMemcpyDeviceToDevice, stream)); }
				NODE_HIP(hipStreamSynchronize(stream));
				if(rank == 0 && total){
					hits.resize(total);
					NODE_HIP(hipMemcpy(hits.data(), d_all, total*sizeof(kwage_hit), hipMemcpyDeviceToHost));
				}
			}

			if(rank == 0 && total){
				kwage_sort_hits(hits.data(), total);               // by (query, global column): groups, ranks and files in order
				// num_query_kmer depends on the k-mer length only: one k-mer stage per distinct length of the database
				map<uint32_t, vector<uint32_t> > nk_by_k;
				for(const NodeGroup &g : groups){
					if(nk_by_k.count(g.params.kmer_len)){ continue; }
					vector<uint64_t> off(q.size() + 1);
					vector<uint32_t> nk(q.size());
					check(kwage_hash_batch(ctx, &g.params, b, off.data(), nk.data(), nullptr, nullptr));
					nk_by_k[g.params.kmer_len] = std::move(nk);
				}
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
						m.num_query_kmer = nk_by_k[blk.kmer_len][qi];
						m.file_index = blk.file_index;
						m.column = (uint32_t)(hits[i].column - blk.first_global_column);
						dst.push_back(m);
					}
				}
			}
			kwage_batch_destroy(b);
		};

		{
			CommandLineQueries typed(cli.query_seqs);
			QueryBatch q;
			while(typed.fill(q, max_batch_bases)){ search_batch(q, from_command_line); }
		}
		if(!cli.query_files.empty()){
			FileQueries disk(cli.query_files);
			QueryBatch q;
			while(disk.fill(q, max_batch_bases)){ search_batch(q, from_files); }
		}

		if(comm){ NODE_NCCL(ncclCommDestroy(comm)); }
		(void)hipFree(d_count); (void)hipFree(d_counts); (void)hipFree(d_hits);
		if(d_all){ (void)hipFree(d_all); }
		(void)hipStreamDestroy(stream);
		for(NodeGroup &g : groups){ if(g.mine){ kwage_group_destroy(g.mine); } }
		kwage_shutdown(ctx);

		if(rank == 0){
			// order and report exactly as kwage does (kwage_main.cpp; reference kwage.cpp:191-315)
			for(Findings *f : {&from_command_line, &from_files}){
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
			report->begin(from_command_line.by_query.size() + from_files.by_query.size());
			for(const auto &kv : from_command_line.by_query){ report->query("command line seq " + to_string(kv.first), kv.second); }
			for(const auto &kv : from_files.by_query){ report->query(from_files.defline[kv.first], kv.second); }
			report->end();
			cerr << "Search complete in " << (time(nullptr) - started) << " sec" << endl;
		}
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
	// the parent touches no GPU: the devices are counted by a short-lived child, the ranks forked before any HIP call
	int n_ranks = (int)env_u64("KWAGE_NODE_RANKS", 0);
	if(n_ranks <= 0){ n_ranks = device_count_in_child(); }
	if(n_ranks < 1 || n_ranks > 64){
		cerr << "kwage_node: no usable device count (" << n_ranks << "); set KWAGE_NODE_RANKS" << endl;
		return EXIT_FAILURE;
	}
	if(env_u64("KWAGE_NODE_PLAN", 0)){ return print_plan(db_paths, n_ranks); }
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
	char id_path[] = "/tmp/kwage_node_id_XXXXXX";
	const int fd = mkstemp(id_path);
	if(fd < 0){ perror("mkstemp"); return EXIT_FAILURE; }
	close(fd);
	unlink(id_path);                                       // rank 0 creates it when the id is complete
	vector<pid_t> kids;
	for(int r = 0; r < n_ranks; ++r){
		const pid_t pid = fork();
		if(pid < 0){ perror("fork"); return EXIT_FAILURE; }
		if(pid == 0){
			const int rc = run_rank(r, n_ranks, id_path, cli, db_paths, rehearsal);
			cout.flush();
			fflush(nullptr);
			_exit(rc);
		}
		kids.push_back(pid);
	}
	// a rank that fails would leave the others waiting in the exchange: end them too
	int rc = EXIT_SUCCESS;
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
	unlink(id_path);
	return rc;
}
