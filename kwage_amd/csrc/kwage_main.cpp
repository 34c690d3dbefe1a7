// kwage_amd/csrc/kwage_main.cpp -- the `kwage` command-line program, drop-in for the reference's kwage
// (kwage.cpp main, options.cpp:39-192 for the option surface, output.h:35-112 for the report bytes): same
// options, same `.db` files, same CSV / JSON output.  The structure is this repo's own:
//
//   reference:  for each .db file: for each query: search()  -- seek+read one slice per (k-mer, hash)
//   here:       for each parameter group: load all its files' columns into one HBM matrix; as many groups as fit
//               the device are resident together (a pass); then STREAM the queries through the pass in batches
//               (kwage_search_submit / _collect, two in flight): a reader thread parses ahead, each batch is
//               uploaded once and searched against every resident matrix.  Host memory is O(batch + hits), as
//               in the reference (one record at a time, kwage.cpp:129-148), never O(query set).
//
// Results are order-independent, so the loop inversion is invisible in the output.
// Extra knobs are environment variables only, to keep the option surface verbatim:
//   KWAGE_DEVICE      HIP device index (default 0)
//   KWAGE_DEVICES     "all" or "0,1,...": shard the database files over several GPUs
//   KWAGE_EARLY_EXIT  1 = enable the reference's early-exit shortcut on the device (default 1)
//   KWAGE_BATCH_BASES max bases per query batch (default 64 Mi)
//   KWAGE_MAX_GROUP_BYTES  cap on the HBM bit matrices of one pass (default: 7/8 of the free device memory): the
//                     parameter groups (filter sizes) that fit together are resident together and the query files
//                     are read once per pass; a larger database takes several passes over whole files
//   KWAGE_ONE_UNIT_PER_PASS  1 = one parameter group per pass (measurement: the schedule of earlier versions)
//   KWAGE_CACHE_READER  threads of the page-cache reader, a child process that reads database files which are not in
//                     the page cache ahead of the loader (default 4; 0 = none); KWAGE_CACHE_READER_AHEAD_MB (8192)
//   KWAGE_SPARSE      how a SMALL query set (everything fits one batch of KWAGE_SPARSE_BASES, default 4 Mi bases) is
//                     searched: "auto" (default) loads only the slices the queries address when they address at most
//                     1/8 of a group's rows -- I/O proportional to the queries, like the reference's seek + read per
//                     slice --, "1" always does, "0" always loads whole files
//   KWAGE_SPARSE_SCREEN  at -t 1 with slices fetched on demand: the slices of every query's first N k-mers (default 32;
//                     0 = off) are fetched and searched first, and only files that hold a candidate column are
//                     fetched for the rest -- the reads the reference's early exit never makes (kwage.cpp:466-481)
//   KWAGE_VERBOSE     1 = per-stage wall times on stderr
#include "cli_common.hpp"

namespace {

// Batches that were read already, then the rest of the source they came from: the first pass continues where the
// preview of the query files stopped, so a run of one pass reads every query file exactly once, like the reference.
struct ResumedQueries : QuerySource {
	deque<QueryBatch> head;
	QuerySource &rest;
	explicit ResumedQueries(QuerySource &r) : rest(r) {}
	bool fill(QueryBatch &b, uint64_t max_bases) override
	{
		if(head.empty()){ return rest.fill(b, max_bases); }
		b = std::move(head.front());
		head.pop_front();
		return true;
	}
};

// A batch that was read before (a small query set is parsed ONCE and reused for every group and pass).
struct PreloadedQueries : QuerySource {
	const QueryBatch &stored;
	bool given = false;
	explicit PreloadedQueries(const QueryBatch &b) : stored(b) {}
	bool fill(QueryBatch &b, uint64_t) override
	{
		if(given || stored.size() == 0){ return false; }
		b = stored;
		given = true;
		return true;
	}
};

// =====================================================================================================
// search
// =====================================================================================================
// The columns of one loaded group: which file each global column came from.
struct ColumnMap {
	vector<const DbFileEntry*> files;      // increasing first_column
	vector<uint32_t> file_index;
	void locate(uint32_t column, uint32_t &file, uint32_t &local) const
	{
		size_t lo = 0, hi = files.size();
		while(hi - lo > 1){
			const size_t mid = (lo + hi)/2;
			if(files[mid]->first_column <= column){ lo = mid; } else { hi = mid; }
		}
		file = file_index[lo];
		local = (uint32_t)(column - files[lo]->first_column);
	}
};

// The distinct bit-slice rows the queries of `q` address under parameters `p` (sorted), appended to `rows`: the device
// k-mer stage gives them (kwage_hash_batch), exactly the rows search() would seek to (kwage.cpp:404-416).
void addressed_rows(kwage_ctx *ctx, const kwage_params &p, const QueryBatch &q, vector<uint32_t> &rows)
{
	if(q.size() == 0){ return; }
	kwage_batch *b = nullptr;
	check(kwage_batch_create(ctx, q.bases.data(), q.offsets.data(), (uint32_t)q.size(), &b));
	vector<uint64_t> off(q.size() + 1);
	vector<uint32_t> nk(q.size());
	int rc = kwage_hash_batch(ctx, &p, b, off.data(), nk.data(), nullptr, nullptr);      // sizes
	vector<uint32_t> all;
	if(rc == KWAGE_OK && off.back()){
		all.resize(off.back()*p.num_hash);
		rc = kwage_hash_batch(ctx, &p, b, off.data(), nk.data(), nullptr, all.data());
	}
	kwage_batch_destroy(b);
	check(rc);
	for(size_t i = 0; i < q.size(); ++i){
		rows.insert(rows.end(), all.begin() + off[i]*p.num_hash, all.begin() + (off[i] + nk[i])*p.num_hash);
	}
	sort(rows.begin(), rows.end());
	rows.erase(unique(rows.begin(), rows.end()), rows.end());
}

// The hits of one collected batch, translated to (file, column within the file) and filed under their query's id.
void record_hits(const kwage_result &res, const QueryBatch &q, const ColumnMap &cols, Findings &found)
{
	// the list is sorted by (query, column): one look-up per query, and ids mostly arrive in increasing order
	for(uint64_t i = 0; i < res.n_hits; ){
		const uint32_t qi = res.hits[i].query;
		uint64_t j = i;
		while(j < res.n_hits && res.hits[j].query == qi){ ++j; }
		const size_t id = q.ids[qi];
		vector<Match> &dst = found.by_query.try_emplace(found.by_query.end(), id)->second;
		if(!q.deflines.empty()){ found.defline.try_emplace(found.defline.end(), id, q.deflines[qi]); }
		dst.reserve(dst.size() + (size_t)(j - i));
		for(; i < j; ++i){
			const kwage_hit &h = res.hits[i];
			Match m;
			m.num_kmers_found = h.num_match;
			m.num_query_kmer = res.num_query_kmer[qi];
			cols.locate(h.column, m.file_index, m.column);
			dst.push_back(m);
		}
	}
}

// One resident matrix: a span of whole files with equal parameters.
struct ResidentUnit {
	kwage_group *group = nullptr;
	uint32_t kmer_len = 0;
	ColumnMap cols;
};

// Stream one query source through every resident unit: each batch is uploaded once (once per distinct k-mer length:
// the device-side position tables of a batch depend on it) and searched against one unit after the other.  Two
// searches are in flight: the next one is submitted -- and the next batch parsed -- while the device works, then the
// older one is collected and its hits filed.
void search_stream(kwage_ctx *ctx, const vector<ResidentUnit> &units, QuerySource &source, float threshold,
                   uint32_t flags, uint64_t max_batch_bases, Findings &found)
{
	struct Uploaded {               // one query batch on the device, shared by its searches
		QueryBatch q;
		map<uint32_t, kwage_batch*> by_k;
		~Uploaded() { for(auto &kv : by_k){ if(kv.second){ kwage_batch_destroy(kv.second); } } }
	};
	struct InFlight {
		shared_ptr<Uploaded> batch;
		const ResidentUnit *unit = nullptr;
		kwage_pending *pending = nullptr;
	};
	deque<InFlight> flying;
	auto abandon = [&]() {
		for(InFlight &f : flying){
			kwage_result *r = nullptr;
			if(f.pending && kwage_search_collect(f.pending, &r) == KWAGE_OK){ kwage_result_free(r); }
		}
		flying.clear();
	};
	auto finish_oldest = [&]() {
		InFlight f = flying.front();
		flying.pop_front();
		kwage_result *res = nullptr;
		check(kwage_search_collect(f.pending, &res));
		record_hits(*res, f.batch->q, f.unit->cols, found);
		kwage_result_free(res);
	};
	try{
		QueryBatch next;
		while(source.fill(next, max_batch_bases)){
			shared_ptr<Uploaded> up = make_shared<Uploaded>();
			up->q = std::move(next);
			for(const ResidentUnit &u : units){
				kwage_batch *&kb = up->by_k[u.kmer_len];
				if(!kb){ check(kwage_batch_create(ctx, up->q.bases.data(), up->q.offsets.data(), (uint32_t)up->q.size(), &kb)); }
			}
			up->q.bases = string();                                // resident on the device now
			for(const ResidentUnit &u : units){
				if(flying.size() == 2){ finish_oldest(); }
				InFlight f;
				f.batch = up;
				f.unit = &u;
				check(kwage_search_submit(u.group, up->by_k[u.kmer_len], threshold, flags, &f.pending));
				flying.push_back(f);
			}
			next = QueryBatch();
		}
		while(!flying.empty()){ finish_oldest(); }
	}
	catch(...){
		abandon();
		throw;
	}
}

// =====================================================================================================
// page-cache reader
// =====================================================================================================
// Database files that are not in the page cache are read at half the rate the disk gives when the loader's own thread
// faults their pages in while pinning them, and reading and copying then run in turn (profiles/r02_cold_load.txt).
// Helper THREADS made it worse -- they contend with the loader for the process's address-space lock -- so the reading
// ahead is done by a child PROCESS, forked before anything has touched the GPU: it walks the files in the loader's
// order, never more than `lookahead` bytes ahead of what the loader reports as passed (kwage_set_load_progress into a
// word both processes share), asks mincore() whether a 64 MiB piece is resident already (a warm database costs a
// system call per piece) and reads it if not, several pieces at a time.
struct ReaderShared {
	volatile uint64_t passed;       // bytes of the file sequence the loader is done with
	volatile int go, stop;
};

struct CacheReader {
	ReaderShared *shared = nullptr;
	pid_t child = -1;

	static void child_main(ReaderShared *sh, const vector<string> &paths, unsigned nthreads, uint64_t lookahead)
	{
		prctl(PR_SET_PDEATHSIG, SIGKILL);          // never outlive the search
		if(getppid() == 1){ _exit(0); }
		while(!sh->go && !sh->stop){ usleep(500); }
		struct Piece { const string *path; uint64_t offset, length, start; };
		vector<Piece> pieces;
		uint64_t total = 0;
		const uint64_t piece_bytes = 64ull << 20;
		for(const string &path : paths){
			struct stat st;
			if(stat(path.c_str(), &st) != 0 || st.st_size <= 0){ continue; }
			for(uint64_t off = 0; off < (uint64_t)st.st_size; off += piece_bytes){
				pieces.push_back(Piece{&path, off, min<uint64_t>(piece_bytes, (uint64_t)st.st_size - off), total + off});
			}
			total += (uint64_t)st.st_size;
		}
		atomic<size_t> next{0};
		auto run = [&]() {
			const long page = sysconf(_SC_PAGESIZE);
			vector<unsigned char> resident;
			vector<char> buf(4u << 20);
			for(;;){
				const size_t i = next.fetch_add(1);
				if(i >= pieces.size()){ return; }
				const Piece &p = pieces[i];
				while(!sh->stop && p.start > sh->passed + lookahead){ usleep(300); }
				if(sh->stop){ return; }
				if(p.start + p.length <= sh->passed){ continue; }          // the loader has been here already
				const int fd = open(p.path->c_str(), O_RDONLY);
				if(fd < 0){ continue; }
				bool cold = true;
				void *m = mmap(nullptr, (size_t)p.length, PROT_READ, MAP_SHARED, fd, (off_t)p.offset);
				if(m != MAP_FAILED){
					const size_t npages = (size_t)((p.length + page - 1)/page);
					resident.assign(npages, 0);
					size_t present = 0;
					if(mincore(m, (size_t)p.length, resident.data()) == 0){ for(unsigned char r : resident){ present += r & 1; } }
					cold = present < npages;
					(void)munmap(m, (size_t)p.length);
				}
				for(uint64_t done = 0; cold && done < p.length && !sh->stop; ){
					const ssize_t got = pread(fd, buf.data(), (size_t)min<uint64_t>(buf.size(), p.length - done), (off_t)(p.offset + done));
					if(got <= 0){ break; }
					done += (uint64_t)got;
				}
				close(fd);
			}
		};
		vector<thread> pool;
		for(unsigned t = 1; t < nthreads; ++t){ pool.emplace_back(run); }
		run();
		for(thread &t : pool){ t.join(); }
		_exit(0);
	}

	// Call before the first HIP call of the process, and before any thread is started.
	void start(const vector<string> &ordered_paths, unsigned nthreads, uint64_t lookahead)
	{
		void *m = mmap(nullptr, sizeof(ReaderShared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
		if(m == MAP_FAILED){ return; }
		shared = new (m) ReaderShared();
		shared->passed = 0; shared->go = 0; shared->stop = 0;
		cout.flush(); cerr.flush();
		child = fork();
		if(child == 0){ child_main(shared, ordered_paths, nthreads, lookahead); _exit(0); }
		if(child < 0){ (void)munmap(m, sizeof(ReaderShared)); shared = nullptr; }
	}
	void release(bool dense) { if(shared){ if(dense){ shared->go = 1; } else { shared->stop = 1; } } }
	void finish()
	{
		if(!shared){ return; }
		shared->stop = 1;
		if(child > 0){ int status = 0; (void)waitpid(child, &status, 0); child = -1; }
		(void)munmap((void*)shared, sizeof(ReaderShared));
		shared = nullptr;
	}
	~CacheReader() { finish(); }
};

double now_s() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); }

// resident set size of this process in MB, now and at its peak (KWAGE_VERBOSE)
string rss_mb()
{
	ifstream st("/proc/self/status");
	string line, cur = "?", peak = "?";
	while(getline(st, line)){
		if(line.compare(0, 6, "VmRSS:") == 0){ cur = to_string(atol(line.c_str() + 6)/1024); }
		if(line.compare(0, 6, "VmHWM:") == 0){ peak = to_string(atol(line.c_str() + 6)/1024); }
	}
	return cur + " MB resident (peak " + peak + ")";
}

// *hip_is_up: the device count had to come from a HIP call in THIS process (no page-cache reader may be forked then)
vector<int> chosen_devices(bool *hip_is_up)
{
	vector<int> devices;
	*hip_is_up = false;
	if(const char *dl = getenv("KWAGE_DEVICES")){
		if(string(dl) == "all"){
			int n = device_count_in_child();
			if(n < 0){ n = kwage_device_count(); *hip_is_up = true; }
			for(int d = 0; d < n; ++d){ devices.push_back(d); }
		}
		else{
			stringstream ss(dl);
			for(string tok; getline(ss, tok, ','); ){ if(!tok.empty()){ devices.push_back(atoi(tok.c_str())); } }
		}
	}
	if(devices.empty()){ devices.push_back((int)env_u64("KWAGE_DEVICE", 0)); }
	return devices;
}

}  // namespace

int main(int argc, char *argv[])
{
	// two search streams per device context plus the loader's: keep them on hardware queues of their own
	// (HIP's default is 4 queues per device for all streams of the process); an explicit setting wins
	setenv("GPU_MAX_HW_QUEUES", "8", 0);
	try{
		const time_t started = time(nullptr);
		const double t_main = now_s();

		Cli cli;
		vector<string> db_paths;
		if(!read_command_line(argc, argv, cli, db_paths)){ return EXIT_SUCCESS; }

		ofstream fout;
		if(!cli.output_path.empty()){
			fout.open(cli.output_path.c_str());
			if(!fout){
				cerr << "Unable to open " << cli.output_path << " for writing" << endl;
				return EXIT_FAILURE;
			}
		}
		ostream &out = fout.is_open() ? fout : cout;

		// ---- headers + metadata of every database file, read once (kwage.cpp:89-113, 505-515) ------------
		vector<DbFileEntry> files(db_paths.size());
		vector<DbInfo> infos(db_paths.size());
		for(size_t i = 0; i < db_paths.size(); ++i){
			files[i].path = db_paths[i];
			if(kwage_db_read_header(files[i].path.c_str(), &files[i].header) != KWAGE_OK){
				cerr << kwage_last_error() << endl;
				throw "main: I/O error";
			}
			string err;
			if(!infos[i].open(files[i].path, err)){
				cerr << err << endl;
				throw "main: Unable to read header";
			}
		}

		// ---- files grouped by (k, hashes, log2 length, hash function): one HBM matrix per group ------------
		struct GroupKey {
			uint32_t kmer_len, num_hash, log_2_filter_len; int32_t hash_func;
			bool operator<(const GroupKey &o) const
			{
				return std::tie(kmer_len, num_hash, log_2_filter_len, hash_func) < std::tie(o.kmer_len, o.num_hash, o.log_2_filter_len, o.hash_func);
			}
		};
		map<GroupKey, vector<uint32_t> > groups;
		for(size_t i = 0; i < files.size(); ++i){
			const kwage_db_header &h = files[i].header;
			groups[GroupKey{h.kmer_len, h.num_hash, h.log_2_filter_len, h.hash_func}].push_back((uint32_t)i);
		}

		// KWAGE_DEVICES="all" | "0,1,..." shards every group's files (whole files, contiguous, balanced by column
		// count) over several GPUs, one host thread + one kwage_ctx per GPU -- the reference's only parallel axis is
		// the same one (OpenMP over files, kwage.cpp:76-87).  Every worker streams the query files itself.
		bool hip_is_up = false;
		const vector<int> devices = chosen_devices(&hip_is_up);
		const size_t ndev = devices.size();
		const uint32_t flags = env_u64("KWAGE_EARLY_EXIT", 1) ? KWAGE_SEARCH_EARLY_EXIT : 0u;
		const uint64_t max_batch_bases = env_u64("KWAGE_BATCH_BASES", 64ull << 20);
		const uint64_t max_group_bytes = env_u64("KWAGE_MAX_GROUP_BYTES", 0);       // 0 = what is free on the device
		const bool verbose = env_u64("KWAGE_VERBOSE", 0) != 0;

		// A device's share of a group: a file belongs to the device that owns its middle column.
		auto share_of = [&](const vector<uint32_t> &group_files, size_t di) {
			uint64_t total = 0;
			for(uint32_t fi : group_files){ total += files[fi].header.num_filter; }
			vector<uint32_t> members;
			uint64_t before = 0;
			for(uint32_t fi : group_files){
				const uint64_t nf = files[fi].header.num_filter;
				const size_t owner = min<size_t>(ndev - 1, (size_t)(((long double)before + nf/2.0L)*ndev/max<uint64_t>(total, 1)));
				if(owner == di){ members.push_back(fi); }
				before += nf;
			}
			return members;
		};

		// The page-cache readers, one per device (each follows the loading of its device's share): forked here, before
		// the first HIP call and before any thread exists; they wait until the plan says whole files are read.
		vector<CacheReader> readers(ndev);
		vector<vector<string> > load_order(ndev);
		const unsigned reader_threads = hip_is_up ? 0u : (unsigned)min<uint64_t>(env_u64("KWAGE_CACHE_READER", 4), 16);
		for(size_t di = 0; di < ndev && reader_threads; ++di){
			for(const auto &grp_entry : groups){ for(uint32_t fi : share_of(grp_entry.second, di)){ load_order[di].push_back(files[fi].path); } }
			readers[di].start(load_order[di], reader_threads, env_u64("KWAGE_CACHE_READER_AHEAD_MB", 8192) << 20);
		}

		// A small query set is read ONCE, up front, and reused for every group and pass; it is also what makes the sparse
		// path possible (the rows to fetch must be known before the files are read).
		const string sparse_mode = getenv("KWAGE_SPARSE") ? getenv("KWAGE_SPARSE") : "auto";
		const uint64_t small_bases = min<uint64_t>(env_u64("KWAGE_SPARSE_BASES", 4ull << 20), max_batch_bases);
		QueryBatch typed_all, disk_all;
		bool small_set = false;
		// a query set that turns out not to be small: what the preview has read of the files, and the reader to go on with
		unique_ptr<FileQueries> preview_reader(new FileQueries(cli.query_files));
		unique_ptr<ResumedQueries> preview;
		mutex preview_lock;
		{
			CommandLineQueries typed(cli.query_seqs);
			QueryBatch extra;
			const bool any_typed = typed.fill(typed_all, small_bases);
			const bool typed_done = !any_typed || !typed.fill(extra, small_bases);
			const bool any_disk = typed_done && preview_reader->fill(disk_all, small_bases);
			const bool more_disk = any_disk && preview_reader->fill(extra, small_bases);
			small_set = typed_done && !more_disk;
			if(!small_set){
				if(more_disk){
					preview.reset(new ResumedQueries(*preview_reader));
					preview->head.push_back(std::move(disk_all));
					preview->head.push_back(std::move(extra));
				}
				typed_all.clear();
				disk_all.clear();
			}
		}

		// whole files are certain to be read when no sparse group can come about: the reader may start during HIP initialisation
		if(!small_set || sparse_mode == "0"){ for(CacheReader &r : readers){ r.release(true); } }

		Findings from_command_line, from_files;
		mutex merge_lock;
		vector<string> worker_error(ndev);
		const double t_start = now_s();
		if(env_u64("KWAGE_VERBOSE", 0)){ cerr << "[kwage] command line, headers of " << files.size() << " files, query preview: " << (t_start - t_main) << " s" << endl; }

		auto worker = [&](size_t di) {
			try{
				kwage_ctx *ctx = nullptr;
				check(kwage_init(devices[di], &ctx));
				one_shot_placement(ctx);
				if(verbose){ lock_guard<mutex> lk(merge_lock); cerr << "[kwage] device " << devices[di] << " ready: " << rss_mb() << endl; }
				CacheReader &reader = readers[di];
				if(reader.shared){ kwage_set_load_progress(ctx, &reader.shared->passed); }
				double t_load = 0, t_search = 0, t_free = 0, gb_loaded = 0;
				const double t_init = now_s() - t_start;
				Findings local_cmdline, local_files;
				// ---- plan: this device's files, cut into units (spans of whole files with equal parameters, one resident
				// matrix each) and the units packed into passes that fit the HBM that is free.  The queries are streamed
				// once per PASS, against every unit of it: a database whose filter sizes differ (several parameter groups)
				// but which fits the device is searched with a single reading of the query files.  A group larger than a
				// pass is continued in the next one (results are additive).
				struct UnitPlan {
					kwage_params p;
					vector<uint32_t> members;           // indices into files[]
					uint64_t span_bytes = 0, nrows = 0;
					shared_ptr<vector<uint32_t> > sparse_rows;       // null: all slices
				};
				vector<vector<UnitPlan> > passes(1);
				uint64_t budget = max_group_bytes;       // per pass
				if(budget == 0){
					uint64_t free_b = 0, total_b = 0;
					check(kwage_mem_info(ctx, &free_b, &total_b));
					budget = free_b - free_b/8;              // leave room for staging buffers, row indices, hits
				}
				uint64_t pass_left = budget;
				const bool one_unit_per_pass = env_u64("KWAGE_ONE_UNIT_PER_PASS", 0) != 0;      // measurement hook: the schedule before units shared passes

				// ---- t = 1, small query set, slices fetched on demand: SCREEN, THEN FETCH.  The reference's early exit
				// (kwage.cpp:466-481) gives a (query, file) pair up after a handful of slices and never READS the rest -- for a one-shot
				// run against files in the page cache that is most of its speed.  Here: the slices of every query's first
				// KWAGE_SPARSE_SCREEN (default 32) k-mers are fetched from every file and searched (a column that misses one of them
				// cannot hold all of the query's), then only the files that hold a candidate column are fetched again, with the slices
				// of the queries that have one, and those queries searched in full.  Same report; I/O of the order the reference's.
				const uint64_t screen_kmers = env_u64("KWAGE_SPARSE_SCREEN", 32);
				auto load_sparse_unit = [&](const kwage_params &p, const vector<uint32_t> &mem, const vector<uint32_t> &rows, ResidentUnit &u) {
					uint64_t span_bytes = 0;
					vector<const char*> paths;
					for(uint32_t fi : mem){
						span_bytes = (span_bytes + 15)/16*16 + ((uint64_t)files[fi].header.num_filter + 7)/8;
						paths.push_back(files[fi].path.c_str());
					}
					u.kmer_len = p.kmer_len;
					check(kwage_group_create_sparse(ctx, &p, span_bytes*8, rows.data(), rows.size(), &u.group));
					vector<uint64_t> firsts(paths.size());
					check(kwage_group_add_db_files(u.group, paths.data(), (uint32_t)paths.size(), firsts.data(), nullptr));
					for(size_t m = 0; m < mem.size(); ++m){
						DbFileEntry &f = files[mem[m]];
						f.first_column = firsts[m];
						u.cols.files.push_back(&f);
						u.cols.file_index.push_back(mem[m]);
					}
					check(kwage_group_finalize(u.group));
					gb_loaded += (double)kwage_group_row_bytes(u.group)*(double)rows.size()/1e9;
				};
				auto span_fits = [&](const vector<uint32_t> &mem, uint64_t nrows) {
					uint64_t span_bytes = 0;
					for(uint32_t fi : mem){ span_bytes = (span_bytes + 15)/16*16 + ((uint64_t)files[fi].header.num_filter + 7)/8; }
					return ((span_bytes + 127)/128*128)*nrows <= budget;
				};
				auto screen_then_fetch = [&](const kwage_params &p, const vector<uint32_t> &members, size_t n_rows_all) -> bool {
					struct Drop { kwage_group *&g; ~Drop() { if(g){ kwage_group_destroy(g); g = nullptr; } } };
					const QueryBatch *full[2] = {&typed_all, &disk_all};
					QueryBatch heads[2], subs[2];
					const uint64_t head_len = screen_kmers + p.kmer_len - 1;
					for(int s = 0; s < 2; ++s){
						heads[s].clear();
						subs[s].clear();
						for(size_t i = 0; i < full[s]->size(); ++i){
							const uint64_t o = full[s]->offsets[i], len = full[s]->offsets[i + 1] - o;
							heads[s].add(full[s]->ids[i], full[s]->bases.substr(o, min(len, head_len)), nullptr);
						}
					}
					vector<uint32_t> head_rows;
					addressed_rows(ctx, p, heads[0], head_rows);
					addressed_rows(ctx, p, heads[1], head_rows);
					if(head_rows.empty() || head_rows.size()*4 > n_rows_all || !span_fits(members, head_rows.size())){ return false; }
					double t0 = now_s();
					vector<char> file_in(files.size(), 0);
					// per file and query source: the queries with a candidate column in it (increasing; a query with several columns once)
					map<uint32_t, vector<uint32_t> > cand[2];
					vector<uint32_t> everywhere[2];
					{
						ResidentUnit scr;
						Drop drop{scr.group};
						load_sparse_unit(p, members, head_rows, scr);
						t_load += now_s() - t0;
						t0 = now_s();
						for(int s = 0; s < 2; ++s){
							if(heads[s].size() == 0){ continue; }
							kwage_batch *b = nullptr;
							check(kwage_batch_create(ctx, heads[s].bases.data(), heads[s].offsets.data(), (uint32_t)heads[s].size(), &b));
							kwage_result *res = nullptr;
							const int rc = kwage_search(scr.group, b, 1.0f, 0, &res);
							kwage_batch_destroy(b);
							check(rc);
							vector<char> keep(heads[s].size(), 0);
							for(uint64_t i = 0; i < res->n_hits; ++i){
								uint32_t fi = 0, local = 0;
								scr.cols.locate(res->hits[i].column, fi, local);
								file_in[fi] = 1;
								keep[res->hits[i].query] = 1;
								cand[s][fi].push_back(res->hits[i].query);
							}
							for(size_t i = 0; i < heads[s].size(); ++i){
								// a head without a valid k-mer (N's) says nothing about a longer query: that query meets every file
								const uint64_t len = full[s]->offsets[i + 1] - full[s]->offsets[i];
								if(res->num_query_kmer[i] == 0 && len > head_len){
									keep[i] = 1;
									everywhere[s].push_back((uint32_t)i);
									for(uint32_t fi : members){ file_in[fi] = 1; }
								}
								if(keep[i]){
									subs[s].add(full[s]->ids[i], full[s]->bases.substr(full[s]->offsets[i], len), full[s]->deflines.empty() ? nullptr : &full[s]->deflines[i]);
								}
							}
							kwage_result_free(res);
						}
						t_search += now_s() - t0;
					}
					vector<uint32_t> files2;
					for(uint32_t fi : members){ if(file_in[fi]){ files2.push_back(fi); } }
					if(verbose){
						lock_guard<mutex> lk(merge_lock);
						cerr << "[kwage] device " << devices[di] << " screened 2^" << p.log_2_filter_len << " slices: " << head_rows.size() << " slices of " << members.size()
							<< " files for the first " << screen_kmers << " k-mers; " << (subs[0].size() + subs[1].size()) << " of " << (typed_all.size() + disk_all.size())
							<< " queries hold a candidate, in " << files2.size() << " files" << endl;
					}
					if(files2.empty() || subs[0].size() + subs[1].size() == 0){ return true; }       // nothing can match
					// FILE BY FILE where that is much less to fetch: a file gets the slices of ITS OWN candidate queries only (a thousand
					// queries with a candidate each in three of fifty files: a sixteenth of what the union of the queries costs in every
					// candidate file).  Every file is a load and a search of its own, so only for a few hundred files.
					{
						auto width = [&](uint32_t fi) { return ((uint64_t)files[fi].header.num_filter + 7)/8; };
						auto rows_of = [&](int s, uint32_t i) {
							const uint64_t len = full[s]->offsets[i + 1] - full[s]->offsets[i];
							return (len >= p.kmer_len) ? (len - p.kmer_len + 1)*p.num_hash : 0;
						};
						uint64_t all_rows = 0, union_bytes = 0, own_bytes = 0;
						for(int s = 0; s < 2; ++s){ for(size_t i = 0; i < subs[s].size(); ++i){ const uint64_t len = subs[s].offsets[i + 1] - subs[s].offsets[i]; all_rows += (len >= p.kmer_len) ? (len - p.kmer_len + 1)*p.num_hash : 0; } }
						for(uint32_t fi : files2){
							union_bytes += all_rows*width(fi);
							uint64_t own = 0;
							for(int s = 0; s < 2; ++s){
								for(uint32_t i : everywhere[s]){ own += rows_of(s, i); }
								auto it = cand[s].find(fi);
								if(it == cand[s].end()){ continue; }
								uint32_t last = 0xFFFFFFFFu;
								for(uint32_t i : it->second){ if(i != last){ own += rows_of(s, i); last = i; } }
							}
							own_bytes += own*width(fi);
						}
						if(files2.size() <= 256 && own_bytes*2 <= union_bytes){
							for(uint32_t fi : files2){
								QueryBatch own[2];
								for(int s = 0; s < 2; ++s){
									own[s].clear();
									vector<char> in(full[s]->size(), 0);
									for(uint32_t i : everywhere[s]){ in[i] = 1; }
									auto it = cand[s].find(fi);
									if(it != cand[s].end()){ for(uint32_t i : it->second){ in[i] = 1; } }
									for(size_t i = 0; i < full[s]->size(); ++i){
										if(!in[i]){ continue; }
										const uint64_t o = full[s]->offsets[i];
										own[s].add(full[s]->ids[i], full[s]->bases.substr(o, full[s]->offsets[i + 1] - o), full[s]->deflines.empty() ? nullptr : &full[s]->deflines[i]);
									}
								}
								vector<uint32_t> rows_f;
								addressed_rows(ctx, p, own[0], rows_f);
								addressed_rows(ctx, p, own[1], rows_f);
								if(rows_f.empty()){ continue; }
								t0 = now_s();
								vector<ResidentUnit> unit(1);
								Drop drop{unit[0].group};
								load_sparse_unit(p, vector<uint32_t>(1, fi), rows_f, unit[0]);
								t_load += now_s() - t0;
								t0 = now_s();
								PreloadedQueries typed(own[0]), from_disk(own[1]);
								search_stream(ctx, unit, typed, cli.threshold, flags, max_batch_bases, local_cmdline);
								search_stream(ctx, unit, from_disk, cli.threshold, flags, max_batch_bases, local_files);
								t_search += now_s() - t0;
							}
							if(verbose){ lock_guard<mutex> lk(merge_lock); cerr << "[kwage] device " << devices[di] << " candidate files fetched one by one, each with the slices of its own candidate queries" << endl; }
							return true;
						}
					}
					vector<uint32_t> rows2;
					addressed_rows(ctx, p, subs[0], rows2);
					addressed_rows(ctx, p, subs[1], rows2);
					// the candidate files, as many at a time as the budget holds
					for(size_t m0 = 0; m0 < files2.size(); ){
						size_t m1 = m0 + 1;
						while(m1 < files2.size() && span_fits(vector<uint32_t>(files2.begin() + m0, files2.begin() + m1 + 1), rows2.size())){ ++m1; }
						t0 = now_s();
						vector<ResidentUnit> unit(1);
						Drop drop{unit[0].group};
						load_sparse_unit(p, vector<uint32_t>(files2.begin() + m0, files2.begin() + m1), rows2, unit[0]);
						t_load += now_s() - t0;
						t0 = now_s();
						PreloadedQueries typed(subs[0]), from_disk(subs[1]);
						search_stream(ctx, unit, typed, cli.threshold, flags, max_batch_bases, local_cmdline);
						search_stream(ctx, unit, from_disk, cli.threshold, flags, max_batch_bases, local_files);
						t_search += now_s() - t0;
						m0 = m1;
					}
					return true;
				};
				for(const auto &grp_entry : groups){
					const vector<uint32_t> members = share_of(grp_entry.second, di);
					if(members.empty()){ continue; }
					kwage_params p;
					p.kmer_len = grp_entry.first.kmer_len;
					p.num_hash = grp_entry.first.num_hash;
					p.log_2_filter_len = grp_entry.first.log_2_filter_len;
					p.hash_func = grp_entry.first.hash_func;
					// Sparse path: only the slices the (small) query set addresses are fetched and kept resident.
					shared_ptr<vector<uint32_t> > sparse_rows;
					if(small_set && sparse_mode != "0"){
						vector<uint32_t> rows;
						addressed_rows(ctx, p, typed_all, rows);
						addressed_rows(ctx, p, disk_all, rows);
						if((sparse_mode == "1") || (uint64_t)rows.size()*8 <= (1ull << p.log_2_filter_len)){
							if(rows.empty()){ continue; }          // no valid k-mer in any query: nothing can match (kwage.cpp:369-371)
							if(screen_kmers && cli.threshold == 1.0f && screen_then_fetch(p, members, rows.size())){ continue; }
							sparse_rows = make_shared<vector<uint32_t> >(std::move(rows));
						}
					}
					const uint64_t nrows = sparse_rows ? sparse_rows->size() : (1ull << p.log_2_filter_len);
					for(size_t m0 = 0; m0 < members.size(); ){
						uint64_t span_bytes = 0;
						size_t m1 = m0;
						while(m1 < members.size()){
							const uint64_t next = (span_bytes + 15)/16*16 + ((uint64_t)files[members[m1]].header.num_filter + 7)/8;
							// a file that does not fit what is left goes to the next pass -- unless the pass is still empty:
							// then it is tried alone (and the allocation reports it if it really is too large)
							if(((next + 127)/128*128)*nrows > pass_left && (m1 > m0 || !passes.back().empty())){ break; }
							span_bytes = next;
							++m1;
						}
						if(m1 > m0){
							if(one_unit_per_pass && !passes.back().empty()){ passes.emplace_back(); }
							UnitPlan u;
							u.p = p;
							u.members.assign(members.begin() + m0, members.begin() + m1);
							u.span_bytes = span_bytes;
							u.nrows = nrows;
							u.sparse_rows = sparse_rows;
							const uint64_t used = ((span_bytes + 127)/128*128)*nrows;
							pass_left -= min(pass_left, used);
							passes.back().push_back(std::move(u));
							m0 = m1;
						}
						if(m0 < members.size()){ passes.emplace_back(); pass_left = budget; }
					}
				}

				{       // whole files will be read (no unit fetches addressed slices only): let the page-cache reader run ahead
					bool any = false, dense = true;
					for(const vector<UnitPlan> &pass : passes){ for(const UnitPlan &u : pass){ any = true; dense = dense && !u.sparse_rows; } }
					reader.release(any && dense);
				}
				for(const vector<UnitPlan> &pass : passes){
					if(pass.empty()){ continue; }
					// the query files of this pass are read ahead from now on, beside the loading
					unique_ptr<QuerySource> disk_source;
					unique_ptr<PrefetchedQueries> disk_ahead;
					if(!small_set && !cli.query_files.empty()){
						{       // the first pass to get here goes on from the preview; every other one opens the files again
							lock_guard<mutex> lk(preview_lock);
							if(preview){ disk_source = std::move(preview); }
						}
						if(!disk_source){ disk_source.reset(new FileQueries(cli.query_files)); }
						disk_ahead.reset(new PrefetchedQueries(*disk_source, max_batch_bases));
					}
					vector<ResidentUnit> resident;
					resident.reserve(pass.size());
					struct Release {           // the matrices of the pass go when it ends, however it ends
						vector<ResidentUnit> &units;
						double &seconds;
						~Release()
						{
							const double t0 = now_s();
							for(ResidentUnit &u : units){ if(u.group){ kwage_group_destroy(u.group); } }
							seconds += now_s() - t0;
						}
					} release{resident, t_free};
					double t0 = now_s();
					for(const UnitPlan &plan : pass){
						resident.emplace_back();
						ResidentUnit &u = resident.back();
						u.kmer_len = plan.p.kmer_len;
						if(plan.sparse_rows){ check(kwage_group_create_sparse(ctx, &plan.p, plan.span_bytes*8, plan.sparse_rows->data(), plan.sparse_rows->size(), &u.group)); }
						else{ check(kwage_group_create(ctx, &plan.p, plan.span_bytes*8, &u.group)); }
						vector<const char*> paths;
						for(uint32_t fi : plan.members){ paths.push_back(files[fi].path.c_str()); }
						vector<uint64_t> firsts(paths.size());
						check(kwage_group_add_db_files(u.group, paths.data(), (uint32_t)paths.size(), firsts.data(), nullptr));
						for(size_t m = 0; m < plan.members.size(); ++m){
							DbFileEntry &f = files[plan.members[m]];       // each file is touched by exactly one worker
							f.first_column = firsts[m];
							u.cols.files.push_back(&f);
							u.cols.file_index.push_back(plan.members[m]);
						}
						check(kwage_group_finalize(u.group));
						gb_loaded += (double)kwage_group_row_bytes(u.group)*(double)plan.nrows/1e9;
						if(verbose){
							lock_guard<mutex> lk(merge_lock);
							cerr << "[kwage] device " << devices[di] << " unit loaded: " << plan.members.size() << " files, 2^" << plan.p.log_2_filter_len << " slices, "
								<< (plan.sparse_rows ? to_string(plan.nrows) + " addressed slices resident" : string("all resident")) << "; " << rss_mb() << endl;
						}
					}
					t_load += now_s() - t0;
					t0 = now_s();
					if(small_set){
						PreloadedQueries typed(typed_all), from_disk(disk_all);
						search_stream(ctx, resident, typed, cli.threshold, flags, max_batch_bases, local_cmdline);
						search_stream(ctx, resident, from_disk, cli.threshold, flags, max_batch_bases, local_files);
					}
					else{
						CommandLineQueries typed(cli.query_seqs);
						search_stream(ctx, resident, typed, cli.threshold, flags, max_batch_bases, local_cmdline);
						if(disk_ahead){ search_stream(ctx, resident, *disk_ahead, cli.threshold, flags, max_batch_bases, local_files); }
					}
					t_search += now_s() - t0;
					if(verbose){
						lock_guard<mutex> lk(merge_lock);
						cerr << "[kwage] device " << devices[di] << " pass of " << resident.size() << " unit(s) searched" << endl;
					}
				}
				kwage_set_load_progress(ctx, nullptr);
				reader.finish();          // everything is loaded: the reader has nothing left to do
				const double t_down = now_s();
				kwage_shutdown(ctx);
				lock_guard<mutex> lk(merge_lock);        // as the reference's `omp critical` section, kwage.cpp:154-177
				if(verbose){
					cerr << "[kwage] device " << devices[di] << ": init " << t_init << " s, loaded " << gb_loaded << " GB in " << t_load
						<< " s (" << (t_load > 0 ? gb_loaded/t_load : 0) << " GB/s), search " << t_search << " s, freeing groups " << t_free
						<< " s, shutdown " << (now_s() - t_down) << " s; " << rss_mb() << endl;
				}
				from_command_line.absorb(local_cmdline);
				from_files.absorb(local_files);
			}
			catch(const char *error){ worker_error[di] = error; }
			catch(const string &error){ worker_error[di] = error; }
			catch(...){ worker_error[di] = "Unhandled search error"; }
		};

		if(ndev == 1){ worker(0); }
		else{
			vector<thread> pool;
			for(size_t di = 0; di < ndev; ++di){ pool.emplace_back(worker, di); }
			for(thread &t : pool){ t.join(); }
		}
		for(const string &e : worker_error){
			if(!e.empty()){
				cerr << "Caught the search error: " << e << endl;
				throw e;
			}
		}

		const double t_searched = now_s();
		// ---- order: each query's hits as the single-threaded reference collects them (file order, then
		// column), then its unstable descending sort by hits (kwage.cpp:191-201) -------------------------------
		for(Findings *f : {&from_command_line, &from_files}){
			for(auto &kv : f->by_query){
				sort(kv.second.begin(), kv.second.end(), [](const Match &a, const Match &b) {
					return (a.file_index != b.file_index) ? (a.file_index < b.file_index) : (a.column < b.column);
				});
				sort(kv.second.begin(), kv.second.end(), [](const Match &a, const Match &b) { return a.num_kmers_found > b.num_kmers_found; });
			}
		}

		const double t_ordered = now_s();
		unique_ptr<Report> report;
		if(cli.format == Cli::CSV){ report.reset(new CsvReport(out, infos)); }
		else{ report.reset(new JsonReport(out, cli.threshold, infos)); }
		report->begin(from_command_line.by_query.size() + from_files.by_query.size());
		for(const auto &kv : from_command_line.by_query){                       // command-line queries first, by position
			report->query("command line seq " + to_string(kv.first), kv.second);     // kwage.cpp:237-240
		}
		for(const auto &kv : from_files.by_query){                              // then file queries by global id
			report->query(from_files.defline[kv.first], kv.second);
		}
		report->end();
		if(verbose){
			cerr << "[kwage] workers done " << (t_searched - t_main) << " s after the start of main; hits ordered in " << (t_ordered - t_searched)
			     << " s, report written in " << (now_s() - t_ordered) << " s; " << rss_mb() << endl;
		}

		if(verbose){ cerr << "[kwage] " << (now_s() - t_main) << " s from the start of main to its last statement" << endl; }
		cerr << "Search complete in " << (time(nullptr) - started) << " sec" << endl;
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
