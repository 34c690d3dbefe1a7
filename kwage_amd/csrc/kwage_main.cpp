// kwage_amd/csrc/kwage_main.cpp -- the `kwage` command-line program, drop-in for the reference's
// kwage.cpp: same options (options.cpp:39-192), same `.db` files, same CSV / JSON output
// (output.h:35-112), but the two nested search loops of kwage.cpp:86-148 are replaced by
// batched calls into the MI355X engine (include/kwage_amd.h):
//
//   reference:  for each .db file: for each query: search()  -- seek+read one slice per (k-mer, hash)
//   here:       for each parameter group: load all its files' columns into one HBM matrix,
//               then kwage_search() whole batches of queries against it.
//
// Results are order-independent, so the inversion is invisible in the
// output.  Extra knobs are environment variables only, to keep the option surface verbatim:
//   KWAGE_DEVICE      HIP device index (default 0)
//   KWAGE_DEVICES     "all" or "0,1,...": shard the database files over several GPUs
//   KWAGE_EARLY_EXIT  1 = enable the reference's early-exit shortcut on the device (default 1)
//   KWAGE_BATCH_BASES max bases per query batch (default 256 Mi)
//   KWAGE_MAX_GROUP_BYTES  cap on the HBM bit matrix of one pass (default: 7/8 of the free device memory);
//                     larger parameter groups are searched in several passes over whole files
//   KWAGE_VERBOSE     1 = per-stage wall times on stderr
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <mutex>
#include <thread>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include <dirent.h>
#include <getopt.h>
#include <sys/stat.h>

#include "host.hpp"
#include "kwage_amd.h"

using namespace std;
using namespace kwage;

#define KWAGE_VERSION "0.4d"                 // reference kwage.h:4 (usage text)
#define DEFAULT_SEARCH_THRESHOLD 1.0f        // reference options.h:148

namespace {

struct SearchOptions {                       // reference options.h:16-41
	enum { OUTPUT_CSV, OUTPUT_JSON };
	deque<string> query_files, query_seq, subject_files;
	string output_file;
	float threshold = DEFAULT_SEARCH_THRESHOLD;
	int output_format = OUTPUT_JSON;         // options.h:149
	bool quit = false;
};

// Database files: `.db` (options.cpp:30-33) and `.dbz`, which the reference's README.md:260 names but
// its option parser never accepted (this repo's compressed container).
static bool is_db_file(const string &p) { return find_file_extension(p, ".db") || find_file_extension(p, ".dbz"); }

// Breadth-first walk in readdir order: regular files are reported as met, directories queued
// (reference file_util.h:30-125).
struct FindFiles {
	deque<string> targets;
	void add(const string &p) { targets.push_back(p); }
	void run(deque<string> &out_db)
	{
		while(!targets.empty()){
			const string p = targets.front();
			targets.pop_front();
			struct stat st;
			if(stat(p.c_str(), &st) != 0){ throw "FindFiles::next: Unable to stat entry"; }
			if(S_ISREG(st.st_mode)){
				if(is_db_file(p)){ out_db.push_back(p); }
				continue;
			}
			if(!S_ISDIR(st.st_mode)){ throw "FindFiles::next: Unknown filesystem object"; }
			DIR *dp = opendir(p.c_str());
			if(!dp){ throw "FindFiles::next: Unable to open directory for reading"; }
			while(struct dirent *d = readdir(dp)){
				if(d->d_ino == 0 || !strcmp(d->d_name, ".") || !strcmp(d->d_name, "..")){ continue; }
				const string name = p + '/' + d->d_name;
				struct stat ds;
				if(stat(name.c_str(), &ds) != 0){ closedir(dp); throw "FindFiles::next: Unable to stat entry (2)"; }
				if(S_ISDIR(ds.st_mode)){ targets.push_back(name); }
				else if(S_ISREG(ds.st_mode) && is_db_file(name)){ out_db.push_back(name); }
			}
			closedir(dp);
		}
	}
};

void parse_options(int argc, char *argv[], SearchOptions &o)     // reference options.cpp:39-192
{
	const char *options = "o:d:i:t:h?";
	int config_opt = 0, long_index = 0;
	struct option long_opts[] = {
		{"o.csv", false, &config_opt, 1},
		{"o.json", false, &config_opt, 2},
		{0, 0, 0, 0}
	};
	int opt_code;
	opterr = 0;
	bool print_usage = (argc == 1);
	FindFiles ff;

	while((opt_code = getopt_long(argc, argv, options, long_opts, &long_index)) != EOF){
		switch(opt_code){
			case 0:
				if(config_opt == 1){ o.output_format = SearchOptions::OUTPUT_CSV; break; }
				if(config_opt == 2){ o.output_format = SearchOptions::OUTPUT_JSON; break; }
				cerr << "Unknown flag!" << endl;
				break;
			case 'o': o.output_file = optarg; break;
			case 'i': o.query_files.push_back(optarg); break;
			case 'd': ff.add(optarg); break;
			case 't': o.threshold = atof(optarg); break;
			case 'h':
			case '?': print_usage = true; break;
			default:
				cerr << '\"' << (char)opt_code << "\" is not a valid option!" << endl;
				break;
		}
	}

	if(print_usage){
		o.quit = true;
		cerr << "Usage for KWAGE (v. " << KWAGE_VERSION << "):" << endl;
		cerr << "\t[-o <output file>] (default is stdout)" << endl;
		cerr << "\t[--o.csv (output CSV) | --o.json (output JSON)]" << endl;
		cerr << "\t[-t <search threshold>] (default is " << DEFAULT_SEARCH_THRESHOLD << ")" << endl;
		cerr << "\t-d <database search path> (can be repeated)" << endl;
		cerr << "\t[-i <input sequence file>] (can be repeated)" << endl;
		cerr << "\t[<DNA sequence>] (can be repeated)" << endl;
		return;
	}

	for(int i = optind; i < argc; i++){ o.query_seq.push_back(argv[i]); }

	ff.run(o.subject_files);

	if(o.subject_files.empty()){
		cerr << "Please provide at least one database file to search (-d)" << endl;
		o.quit = true;
		return;
	}
	if(o.query_files.empty() && o.query_seq.empty()){
		cerr << "Please provide at least one query sequence or file" << endl;
		o.quit = true;
		return;
	}
	static const char *allowed_sequence_extentions[] = {     // options.cpp:22-28
		".fna", ".fna.gz", ".fasta", ".fasta.gz", ".fa", ".fa.gz", ".fastq", ".fastq.gz", NULL
	};
	for(deque<string>::const_iterator i = o.query_files.begin(); i != o.query_files.end(); ++i){
		bool valid = false;
		for(const char **ext = allowed_sequence_extentions; *ext != NULL; ++ext){
			const size_t ext_len = strlen(*ext);
			if(ext_len > i->size()){ continue; }
			if(i->find(*ext) == (i->size() - ext_len)){ valid = true; break; }     // case-sensitive, first occurrence
		}
		if(!valid){
			cerr << "The query sequence file name, " << *i << ", does not have an allowed file extension" << endl;
			o.quit = true;
			return;
		}
	}
	if((o.threshold <= 0.0) || (o.threshold > 1.0)){
		cerr << "Please provide: 0.0 < search threshold <= 1.0" << endl;
		o.quit = true;
		return;
	}
}

// One hit, before its metadata is fetched (reference output.h:9-33 MatchResult).
struct Match {
	unsigned int num_kmers_found;
	unsigned int num_query_kmer;
	uint32_t file_index;      // index into opt.subject_files
	uint32_t column;          // column within that file
	bool operator<(const Match &rhs) const { return num_kmers_found > rhs.num_kmers_found; }   // descending
};

typedef unordered_map<size_t, deque<Match> > ResultMap;

struct Query {
	size_t id;
	string seq;
};

struct DbFileEntry {
	string path;
	kwage_db_header header;
	uint64_t first_column = 0;
};

void check(int rc)
{
	if(rc != KWAGE_OK){ throw string(kwage_last_error()); }
}

// Search one set of queries against one loaded group; append matches.  Batches are software-pipelined
// through the context's two search slots: batch i+1 is submitted before batch i is collected, so the
// host-side mapping of hits overlaps with the device work of the next batch.
void search_queries(kwage_ctx *ctx, kwage_group *grp, const vector<DbFileEntry*> &files,
                    const vector<uint32_t> &file_index, const vector<Query> &queries, float threshold,
                    uint32_t flags, uint64_t max_batch_bases, ResultMap &results)
{
	struct InFlight {
		kwage_batch *batch = NULL;
		kwage_pending *pending = NULL;
		size_t q0 = 0;
	};
	auto finish = [&](InFlight &f) {
		kwage_result *res = NULL;
		const int rc = kwage_search_collect(f.pending, &res);
		f.pending = NULL;
		if(rc != KWAGE_OK){ kwage_batch_destroy(f.batch); f.batch = NULL; check(rc); }
		for(uint64_t i = 0; i < res->n_hits; ++i){
			const kwage_hit &h = res->hits[i];
			// global column -> (file, local column): files are laid out in increasing first_column
			size_t lo = 0, hi = files.size();
			while(hi - lo > 1){
				const size_t mid = (lo + hi)/2;
				if(files[mid]->first_column <= h.column){ lo = mid; } else { hi = mid; }
			}
			Match m;
			m.num_kmers_found = h.num_match;
			m.num_query_kmer = res->num_query_kmer[h.query];
			m.file_index = file_index[lo];
			m.column = (uint32_t)(h.column - files[lo]->first_column);
			results[queries[f.q0 + h.query].id].push_back(m);
		}
		kwage_result_free(res);
		kwage_batch_destroy(f.batch);
		f.batch = NULL;
	};

	InFlight prev;
	size_t q0 = 0;
	try{
		while(q0 < queries.size()){
			// assemble one batch
			string concat;
			vector<uint64_t> offs(1, 0);
			size_t q1 = q0;
			while(q1 < queries.size() && (q1 == q0 || concat.size() + queries[q1].seq.size() <= max_batch_bases) &&
			      (q1 - q0) < (1u << 24)){
				concat += queries[q1].seq;
				offs.push_back(concat.size());
				++q1;
			}
			InFlight cur;
			cur.q0 = q0;
			check(kwage_batch_create(ctx, concat.data(), offs.data(), (uint32_t)(q1 - q0), &cur.batch));
			const int rc = kwage_search_submit(grp, cur.batch, threshold, flags, &cur.pending);
			if(rc != KWAGE_OK){ kwage_batch_destroy(cur.batch); check(rc); }
			if(prev.pending){ finish(prev); }
			prev = cur;
			q0 = q1;
		}
		if(prev.pending){ finish(prev); }
	}
	catch(...){
		if(prev.pending){ kwage_result *r = NULL; if(kwage_search_collect(prev.pending, &r) == KWAGE_OK){ kwage_result_free(r); } }
		if(prev.batch){ kwage_batch_destroy(prev.batch); }
		throw;
	}
}

// reference output.h:35-54
void write_csv_header(ostream &out) { out << "query,num_kmers,num_kmers_found,percent_kmers_found,sample_metadata\n"; }

void write_csv(ostream &out, const string &query, const deque<Match> &ms, const vector<DbInfo> &infos)
{
	for(deque<Match>::const_iterator i = ms.begin(); i != ms.end(); ++i){
		const float norm = i->num_query_kmer ? 1.0f/i->num_query_kmer : 0.0f;
		FilterInfo fi;
		if(!infos[i->file_index].info(i->column, fi)){ throw "binary_read<FilterInfo>: Unable to read FilterInfo"; }
		out << '"' << query << "\"," << i->num_query_kmer << ',' << i->num_kmers_found << ','
			<< (100.0f*i->num_kmers_found)*norm << ",\"" << fi.csv_string() << '"' << std::endl;
	}
}

// reference output.h:61-112
void write_json_header(ostream &out, bool multiple) { if(multiple){ out << '['; } }

void write_json(ostream &out, const string &query, bool multiple, bool first_match, const float &threshold,
                const deque<Match> &ms, const vector<DbInfo> &infos)
{
	const string prefix = multiple ? "\t" : "";
	out << ((multiple && !first_match) ? "," : "") << '\n' << prefix
		<< "{\n" << prefix << "\t\"query\": \"" << query << "\",\n" << prefix
		<< "\t\"threshold\": "
		<< std::showpoint << std::setprecision(1) << std::fixed << threshold
		<< ",\n" << prefix << "\t\"results\": [";
	for(deque<Match>::const_iterator i = ms.begin(); i != ms.end(); ++i){
		const float norm = i->num_query_kmer ? 1.0f/i->num_query_kmer : 0.0f;
		FilterInfo fi;
		if(!infos[i->file_index].info(i->column, fi)){ throw "binary_read<FilterInfo>: Unable to read FilterInfo"; }
		out << ((i != ms.begin()) ? "," : "")
			<< "\n" << prefix << "\t\t{\n" << prefix
			<< "\t\t\t\"percent_kmers_found\": "
			<< (100.0*i->num_kmers_found)*norm
			<< ",\n" << prefix << "\t\t\t\"num_kmers\": " << i->num_query_kmer
			<< ",\n" << prefix << "\t\t\t\"num_kmers_found\": " << i->num_kmers_found
			<< ",\n" << prefix << "\t\t\t\"sample_metadata\": {\n"
			<< fi.json_string(prefix + "\t\t\t\t")
			<< "\n" << prefix
			<< "\t\t\t}\n" << prefix << "\t\t}";
	}
	if(!ms.empty()){ out << "\n" << prefix << '\t'; }
	out << "]\n" << prefix << "}";
}

void write_json_footer(ostream &out, bool multiple) { if(multiple){ out << "\n]\n"; } }

}  // namespace

int main(int argc, char *argv[])
{
	// two search streams per device context plus the loader's: keep them on hardware queues of their own
	// (HIP's default is 4 queues per device for all streams of the process); an explicit setting wins
	setenv("GPU_MAX_HW_QUEUES", "8", 0);
	try{
		time_t profile = time(NULL);

		SearchOptions opt;
		parse_options(argc, argv, opt);
		if(opt.quit){ return EXIT_SUCCESS; }

		ofstream fout;
		if(!opt.output_file.empty()){
			fout.open(opt.output_file.c_str());
			if(!fout){
				cerr << "Unable to open " << opt.output_file << " for writing" << endl;
				return EXIT_FAILURE;
			}
		}
		ostream &out = fout.is_open() ? fout : cout;

		// ---- headers + metadata of every database file (kwage.cpp:89-113) -------------------
		const size_t num_subject_files = opt.subject_files.size();
		vector<DbFileEntry> files(num_subject_files);
		vector<DbInfo> infos(num_subject_files);
		for(size_t i = 0; i < num_subject_files; ++i){
			files[i].path = opt.subject_files[i];
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

		// ---- queries, parsed once (the reference re-parses per database file) ---------------
		vector<Query> cmdline_queries, file_queries;
		unordered_map<size_t, string> all_file_deflines;
		for(size_t i = 0; i < opt.query_seq.size(); ++i){ cmdline_queries.push_back(Query{i, opt.query_seq[i]}); }
		size_t query_id = 0;
		for(deque<string>::const_iterator qf = opt.query_files.begin(); qf != opt.query_files.end(); ++qf){
			SeqFile sf;
			string err;
			if(!sf.open(*qf, err)){
				cerr << err << endl;
				throw "SequenceIterator::SequenceIterator: Unable to open sequence file";
			}
			int r;
			while((r = sf.next(err)) == 1){
				file_queries.push_back(Query{query_id, sf.seq});
				all_file_deflines[query_id] = sf.curr_defline;
				++query_id;                   // ids run on across files, kwage.cpp:127-147
			}
			if(r < 0){ throw err; }
		}

		// ---- devices --------------------------------------------------------------------------
		// KWAGE_DEVICES="all" | "0,1,..." shards every group's files (whole files, contiguous, balanced
		// by column count) over several GPUs, one host thread + one kwage_ctx per GPU -- the
		// reference's only parallel axis is the same one (OpenMP over files, kwage.cpp:76-87).
		vector<int> devices;
		if(const char *dl = getenv("KWAGE_DEVICES")){
			if(string(dl) == "all"){
				for(int d = 0; d < kwage_device_count(); ++d){ devices.push_back(d); }
			}
			else{
				stringstream ss(dl);
				string tok;
				while(getline(ss, tok, ',')){ if(!tok.empty()){ devices.push_back(atoi(tok.c_str())); } }
			}
		}
		if(devices.empty()){
			const char *dev_env = getenv("KWAGE_DEVICE");
			devices.push_back(dev_env ? atoi(dev_env) : 0);
		}
		const size_t ndev = devices.size();
		const char *ee = getenv("KWAGE_EARLY_EXIT");
		const uint32_t flags = (ee && atoi(ee) == 0) ? 0u : KWAGE_SEARCH_EARLY_EXIT;
		const char *bb = getenv("KWAGE_BATCH_BASES");
		const uint64_t max_batch_bases = bb ? strtoull(bb, NULL, 10) : (256ull << 20);
		const char *mg = getenv("KWAGE_MAX_GROUP_BYTES");
		const uint64_t max_group_bytes = mg ? strtoull(mg, NULL, 10) : 0;       // 0 = what is free on the device

		ResultMap file_search_results, command_line_search_results;

		// ---- group files by (k, hashes, log2 length, hash function) --------------------------
		typedef pair<pair<uint32_t, uint32_t>, pair<uint32_t, int32_t> > Key;
		map<Key, vector<uint32_t> > groups;
		for(size_t i = 0; i < num_subject_files; ++i){
			const kwage_db_header &h = files[i].header;
			groups[Key(make_pair(h.kmer_len, h.num_hash), make_pair(h.log_2_filter_len, h.hash_func))].push_back((uint32_t)i);
		}

		mutex merge_lock;
		vector<string> worker_error(ndev);
		const bool verbose = getenv("KWAGE_VERBOSE") && atoi(getenv("KWAGE_VERBOSE")) != 0;
		auto now = []() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); };
		const double t_start = now();

		auto worker = [&](size_t di) {
			try{
				kwage_ctx *ctx = NULL;
				check(kwage_init(devices[di], &ctx));
				double t_load = 0, t_search = 0, gb_loaded = 0;
				const double t_init = now() - t_start;
				ResultMap local_file_results, local_cmdline_results;
				for(map<Key, vector<uint32_t> >::const_iterator gi = groups.begin(); gi != groups.end(); ++gi){
					// this device's share: a file belongs to the device that owns its middle column
					uint64_t total = 0;
					for(size_t m = 0; m < gi->second.size(); ++m){ total += files[gi->second[m]].header.num_filter; }
					vector<uint32_t> members;
					uint64_t prefix = 0;
					for(size_t m = 0; m < gi->second.size(); ++m){
						const uint64_t nf = files[gi->second[m]].header.num_filter;
						const size_t owner = min<size_t>(ndev - 1, (size_t)(((long double)prefix + nf/2.0L)*ndev/max<uint64_t>(total, 1)));
						if(owner == di){ members.push_back(gi->second[m]); }
						prefix += nf;
					}
					if(members.empty()){ continue; }
					kwage_params p;
					p.kmer_len = gi->first.first.first;
					p.num_hash = gi->first.first.second;
					p.log_2_filter_len = gi->first.second.first;
					p.hash_func = gi->first.second.second;
					// A group larger than the HBM that is free is searched in several passes over
					// sub-groups of whole files (queries are re-run per pass; results are additive).
					uint64_t budget = max_group_bytes;
					if(budget == 0){
						uint64_t free_b = 0, total_b = 0;
						check(kwage_mem_info(ctx, &free_b, &total_b));
						budget = free_b - free_b/8;          // leave room for staging buffers, row indices, hits
					}
					const uint64_t nrows = 1ull << p.log_2_filter_len;
					size_t m0 = 0;
					while(m0 < members.size()){
						uint64_t span_bytes = 0;
						size_t m1 = m0;
						while(m1 < members.size()){
							const uint64_t next = (span_bytes + 15)/16*16 + ((uint64_t)files[members[m1]].header.num_filter + 7)/8;
							if(m1 > m0 && ((next + 127)/128*128)*nrows > budget){ break; }
							span_bytes = next;
							++m1;
						}
						const vector<uint32_t> part(members.begin() + m0, members.begin() + m1);
						kwage_group *grp = NULL;
						double t0 = now();
						check(kwage_group_create(ctx, &p, span_bytes*8, &grp));
						vector<DbFileEntry*> gfiles;
						for(size_t m = 0; m < part.size(); ++m){
							DbFileEntry &f = files[part[m]];       // each file is touched by exactly one worker
							uint32_t nf = 0;
							int rc = kwage_group_add_db_file(grp, f.path.c_str(), &f.first_column, &nf);
							if(rc != KWAGE_OK){ kwage_group_destroy(grp); check(rc); }
							gfiles.push_back(&f);
						}
						check(kwage_group_finalize(grp));
						t_load += now() - t0;
						gb_loaded += (double)kwage_group_row_bytes(grp)*(double)nrows/1e9;
						t0 = now();
						search_queries(ctx, grp, gfiles, part, cmdline_queries, opt.threshold, flags, max_batch_bases,
						               local_cmdline_results);
						search_queries(ctx, grp, gfiles, part, file_queries, opt.threshold, flags, max_batch_bases,
						               local_file_results);
						t_search += now() - t0;
						kwage_group_destroy(grp);
						m0 = m1;
					}
				}
				kwage_shutdown(ctx);
				if(verbose){
					lock_guard<mutex> lk(merge_lock);
					cerr << "[kwage] device " << devices[di] << ": init " << t_init << " s, loaded " << gb_loaded << " GB in " << t_load
						<< " s (" << (t_load > 0 ? gb_loaded/t_load : 0) << " GB/s), search " << t_search << " s" << endl;
				}
				// merge, as the reference's `omp critical` section does (kwage.cpp:154-177)
				lock_guard<mutex> lk(merge_lock);
				for(ResultMap::const_iterator i = local_file_results.begin(); i != local_file_results.end(); ++i){
					deque<Match> &ref = file_search_results[i->first];
					ref.insert(ref.end(), i->second.begin(), i->second.end());
				}
				for(ResultMap::const_iterator i = local_cmdline_results.begin(); i != local_cmdline_results.end(); ++i){
					deque<Match> &ref = command_line_search_results[i->first];
					ref.insert(ref.end(), i->second.begin(), i->second.end());
				}
			}
			catch(const char *error){ worker_error[di] = error; }
			catch(const string &error){ worker_error[di] = error; }
			catch(...){ worker_error[di] = "Unhandled search error"; }
		};

		if(ndev == 1){ worker(0); }
		else{
			vector<thread> pool;
			for(size_t di = 0; di < ndev; ++di){ pool.push_back(thread(worker, di)); }
			for(size_t di = 0; di < ndev; ++di){ pool[di].join(); }
		}
		for(size_t di = 0; di < ndev; ++di){
			if(!worker_error[di].empty()){
				cerr << "Caught the search error: " << worker_error[di] << endl;
				throw worker_error[di];
			}
		}

		// ---- order: as the single-threaded reference builds each deque (file order, then column),
		// then its unstable descending sort by hits (kwage.cpp:191-201) --------------------------
		ResultMap *maps[2] = {&command_line_search_results, &file_search_results};
		for(int k = 0; k < 2; ++k){
			for(ResultMap::iterator i = maps[k]->begin(); i != maps[k]->end(); ++i){
				std::sort(i->second.begin(), i->second.end(), [](const Match &a, const Match &b){
					return (a.file_index != b.file_index) ? (a.file_index < b.file_index) : (a.column < b.column);
				});
				std::sort(i->second.begin(), i->second.end());
			}
		}

		const bool multiple_query_matches = (command_line_search_results.size() + file_search_results.size()) > 1;

		if(opt.output_format == SearchOptions::OUTPUT_CSV){ write_csv_header(out); }
		else{ write_json_header(out, multiple_query_matches); }

		bool first_match = true;
		vector<size_t> id;
		for(ResultMap::const_iterator i = command_line_search_results.begin(); i != command_line_search_results.end(); ++i){ id.push_back(i->first); }
		std::sort(id.begin(), id.end());
		for(vector<size_t>::const_iterator i = id.begin(); i != id.end(); ++i){
			stringstream ssin;
			ssin << "command line seq " << *i;          // kwage.cpp:237-240
			const deque<Match> &ms = command_line_search_results[*i];
			if(opt.output_format == SearchOptions::OUTPUT_CSV){ write_csv(out, ssin.str(), ms, infos); }
			else{ write_json(out, ssin.str(), multiple_query_matches, first_match, opt.threshold, ms, infos); }
			first_match = false;
		}

		id.clear();
		for(ResultMap::const_iterator i = file_search_results.begin(); i != file_search_results.end(); ++i){ id.push_back(i->first); }
		std::sort(id.begin(), id.end());
		for(vector<size_t>::const_iterator i = id.begin(); i != id.end(); ++i){
			const string &defline = all_file_deflines[*i];
			const deque<Match> &ms = file_search_results[*i];
			if(opt.output_format == SearchOptions::OUTPUT_CSV){ write_csv(out, defline, ms, infos); }
			else{ write_json(out, defline, multiple_query_matches, first_match, opt.threshold, ms, infos); }
			first_match = false;
		}

		if(opt.output_format == SearchOptions::OUTPUT_JSON){ write_json_footer(out, multiple_query_matches); }

		profile = time(NULL) - profile;
		cerr << "Search complete in " << profile << " sec" << endl;
	}
	catch(const char *error){
		cerr << "Caught the error " << error << endl;
		return EXIT_FAILURE;
	}
	catch(const string error){
		cerr << "Caught the error " << error << endl;
		return EXIT_FAILURE;
	}
	catch(...){
		cerr << "Caught an unhandled error" << endl;
		return EXIT_FAILURE;
	}
	return EXIT_SUCCESS;
}
