// oracle/ref_driver/kwage_patched_main.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The reference-side binding INTEGRATION.md documents, compiled: what a KWAGE maintainer's `kwage` looks like once the
// per-file search loops of main (kwage.cpp:76-188) call the C ABI of include/kwage_amd.h instead of search()
// (kwage.cpp:340-541).  Everything around the search is the REFERENCE'S OWN code, included / linked in place from
// $(REF) by oracle/Makefile (nothing is copied): SearchOptions (options.h), DBFileHeader + binary_read (kwage.h,
// binary_io.cpp), SequenceIterator (parse_sequence.h), FilterInfo + binary_read (bloom.h), MatchResult and the CSV / JSON
// writers (output.h), SORT (sort.h), keys (keys.h).  `make -C oracle ref` builds it as _ref/kwage_patched; the GPU test
// tests/test_gpu_patched_reference.py runs it beside _ref/kwage on every case of tests/golden/manifest.json and
// compares the bytes.
//
// Differences from the reference's main that a maintainer would see in the patch:
//   - the database files are grouped by (kmer_len, num_hash, log_2_filter_len, hash_func) and every group is loaded into
//     one resident matrix (kwage_group_*), instead of one seekg + read per k-mer and hash (kwage.cpp:414-416);
//   - queries are read ONCE (the reference re-parses every query file for every database file, kwage.cpp:129-148) and
//     searched as batches (kwage_batch_create + kwage_search);
//   - hits come back as (query, column, num_match) records; FilterInfo is fetched the way kwage.cpp:505-515 does and
//     the MatchResults are appended in the order the single-threaded reference produces them (database file by
//     database file, columns ascending), so that the same SORT yields the same bytes.
#include <algorithm>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "kwage.h"
#include "options.h"
#include "sort.h"
#include "bloom.h"
#include "output.h"
#include "parse_sequence.h"
#include "keys.h"
#include "binary_io.h"

#include "kwage_amd.h"

using namespace std;

// the reference's objects expect these globals from their host program (kwage.cpp:34-35)
int mpi_numtasks;
int mpi_rank;

namespace {

void check(int rc)
{
	if(rc != KWAGE_OK){ throw string(kwage_last_error()); }      // caught by main like the reference's throw literals
}

struct SubjectFile {
	DBFileHeader header;
	uint64_t first_column;      // of this file's block in its group's matrix
};

// One hit, keyed so that sorting reproduces the order in which the single-threaded reference appends MatchResults to a
// query's list: database file by database file (kwage.cpp:86), columns ascending (kwage.cpp:490).
struct Hit {
	size_t query, file;
	uint32_t column, num_match, num_query_kmer;
	bool operator<(const Hit &o) const { return tie(query, file, column) < tie(o.query, o.file, o.column); }
};

// kwage.cpp:505-515: the address of the record from the info index, then the record
FilterInfo read_filter_info(ifstream &fsubject, unsigned long int info_start, uint32_t column)
{
	fsubject.seekg(info_start + column*sizeof(unsigned long int));
	unsigned long int info_loc;
	fsubject.read((char*)&info_loc, sizeof(unsigned long int));
	fsubject.seekg(info_loc);
	FilterInfo info;
	binary_read(fsubject, info);
	return info;
}

// Search one batch of queries against every group; ids[i] is the query id of the batch's i-th sequence.
void search_batch(kwage_ctx *ctx, const vector<kwage_group*> &groups, const vector< vector<size_t> > &group_files,
                  const vector<SubjectFile> &files, const vector<string> &seqs, const vector<size_t> &ids, float threshold,
                  vector<Hit> &hits)
{
	if(seqs.empty()){ return; }
	string concat;
	vector<uint64_t> off(1, 0);
	for(size_t i = 0; i < seqs.size(); ++i){ concat += seqs[i]; off.push_back(concat.size()); }
	kwage_batch *b = NULL;
	check(kwage_batch_create(ctx, concat.data(), off.data(), (uint32_t)seqs.size(), &b));
	for(size_t g = 0; g < groups.size(); ++g){
		kwage_result *r = NULL;
		check(kwage_search(groups[g], b, threshold, KWAGE_SEARCH_EARLY_EXIT, &r));
		for(uint64_t i = 0; i < r->n_hits; ++i){
			const kwage_hit &h = r->hits[i];
			// the file whose column block holds the hit
			size_t f = group_files[g][0];
			for(size_t k = 0; k < group_files[g].size(); ++k){
				if(files[group_files[g][k]].first_column <= h.column){ f = group_files[g][k]; }
			}
			Hit x;
			x.query = ids[h.query]; x.file = f; x.column = (uint32_t)(h.column - files[f].first_column);
			x.num_match = h.num_match; x.num_query_kmer = r->num_query_kmer[h.query];
			hits.push_back(x);
		}
		kwage_result_free(r);
	}
	kwage_batch_destroy(b);
}

// Turn hits into the reference's result map (kwage.cpp:517-534): MatchResult(num_match, num_query_kmer, info)
void append_results(const vector<Hit> &sorted_hits, const SearchOptions &opt, const vector<SubjectFile> &files,
                    unordered_map< size_t, deque<MatchResult> > &results)
{
	size_t open_file = (size_t)(-1);
	ifstream fsubject;
	for(size_t i = 0; i < sorted_hits.size(); ++i){
		const Hit &h = sorted_hits[i];
		if(h.file != open_file){
			fsubject.close();
			fsubject.clear();
			fsubject.open(opt.subject_files[h.file].c_str(), ios::binary);
			if(!fsubject){ throw __FILE__ ":main: I/O error"; }
			open_file = h.file;
		}
		results[h.query].push_back(MatchResult(h.num_match, h.num_query_kmer, read_filter_info(fsubject, files[h.file].header.info_start, h.column)));
	}
}

template <class RESULTS>
void write_matches(ostream &out, const SearchOptions &opt, const RESULTS &results, const unordered_map<size_t, string> *names,
                   bool multiple_query_matches, bool &first_match)
{
	vector<size_t> id = keys(results);
	SORT(id.begin(), id.end());
	for(vector<size_t>::const_iterator i = id.begin(); i != id.end(); ++i){
		typename RESULTS::const_iterator iter = results.find(*i);
		string name;
		if(names){
			unordered_map<size_t, string>::const_iterator n = names->find(*i);
			if(n == names->end()){ throw __FILE__ ":main: Unable to lookup query id in file_query_info"; }
			name = n->second;
		}
		else{
			stringstream ssin;
			ssin << "command line seq " << *i;       // kwage.cpp:238
			name = ssin.str();
		}
		if(opt.output_format == SearchOptions::OUTPUT_CSV){ write_csv(out, name, iter->second.begin(), iter->second.end()); }
		else{ write_json(out, name, multiple_query_matches, first_match, opt.threshold, iter->second.begin(), iter->second.end()); }
		first_match = false;
	}
}

}  // namespace

int main(int argc, char *argv[])
{
	try{
		time_t profile = time(NULL);
		SearchOptions opt(argc, argv);
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
		if(opt.output_format != SearchOptions::OUTPUT_CSV && opt.output_format != SearchOptions::OUTPUT_JSON){
			throw __FILE__ ":main: Unknown output file format (1)";
		}

		// ---- headers, read with the reference's own reader (kwage.cpp:89-101) ----------------------------------
		vector<SubjectFile> files(opt.subject_files.size());
		typedef tuple<unsigned int, unsigned int, unsigned int, int> GroupKey;
		map<GroupKey, size_t> group_of;
		vector< vector<size_t> > group_files;
		for(size_t f = 0; f < files.size(); ++f){
			ifstream fsubject(opt.subject_files[f].c_str(), ios::binary);
			if(!fsubject){
				cerr << "Unable to open database file " << opt.subject_files[f] << " for reading" << endl;
				throw __FILE__ ":main: I/O error";
			}
			binary_read(fsubject, files[f].header);
			if(!fsubject){ throw __FILE__ ":main: Unable to read header"; }
			const DBFileHeader &h = files[f].header;
			const GroupKey key(h.kmer_len, h.num_hash, h.log_2_filter_len, (int)h.hash_func);
			if(group_of.find(key) == group_of.end()){ group_of[key] = group_files.size(); group_files.push_back(vector<size_t>()); }
			group_files[group_of[key]].push_back(f);
		}

		// ---- THE PATCH: resident groups instead of per-slice seeks ------------------------------------------------
		kwage_ctx *ctx = NULL;
		check(kwage_init(getenv("KWAGE_DEVICE") ? atoi(getenv("KWAGE_DEVICE")) : 0, &ctx));
		vector<kwage_group*> groups(group_files.size(), NULL);
		for(size_t g = 0; g < group_files.size(); ++g){
			const DBFileHeader &h0 = files[group_files[g][0]].header;
			uint64_t span_bytes = 0;
			vector<const char*> paths;
			for(size_t k = 0; k < group_files[g].size(); ++k){
				const size_t f = group_files[g][k];
				span_bytes = (span_bytes + 15)/16*16 + (files[f].header.num_filter + 7)/8;      // blocks start at 16-byte boundaries
				paths.push_back(opt.subject_files[f].c_str());
			}
			kwage_params p = { h0.kmer_len, h0.num_hash, h0.log_2_filter_len, (int32_t)h0.hash_func };
			check(kwage_group_create(ctx, &p, span_bytes*8, &groups[g]));
			vector<uint64_t> first(paths.size());
			check(kwage_group_add_db_files(groups[g], paths.data(), (uint32_t)paths.size(), first.data(), NULL));
			check(kwage_group_finalize(groups[g]));
			for(size_t k = 0; k < group_files[g].size(); ++k){ files[group_files[g][k]].first_column = first[k]; }
		}

		unordered_map< size_t, deque<MatchResult> > file_search_results, command_line_search_results;
		unordered_map<size_t, string> file_query_info;

		// sequences given on the command line (kwage.cpp:116-124): query id = position on the command line
		{
			vector<string> seqs(opt.query_seq.begin(), opt.query_seq.end());
			vector<size_t> ids;
			for(size_t i = 0; i < seqs.size(); ++i){ ids.push_back(i); }
			vector<Hit> hits;
			search_batch(ctx, groups, group_files, files, seqs, ids, opt.threshold, hits);
			sort(hits.begin(), hits.end());
			append_results(hits, opt, files, command_line_search_results);
		}
		// sequence files (kwage.cpp:127-148): ids run on across the files; batches of at most 64 Mi bases
		{
			vector<Hit> hits;
			vector<string> seqs;
			vector<size_t> ids;
			unordered_map<size_t, string> defline;
			size_t query_id = 0, bases = 0;
			for(deque<string>::const_iterator qf = opt.query_files.begin(); qf != opt.query_files.end(); ++qf){
				SequenceIterator seq_iter(*qf);
				while(seq_iter){
					seqs.push_back(seq_iter.get_seq());
					ids.push_back(query_id);
					defline[query_id] = seq_iter.get_info();
					bases += seqs.back().size();
					if(bases >= (64u << 20)){
						search_batch(ctx, groups, group_files, files, seqs, ids, opt.threshold, hits);
						seqs.clear(); ids.clear(); bases = 0;
					}
					++seq_iter;
					++query_id;
				}
			}
			search_batch(ctx, groups, group_files, files, seqs, ids, opt.threshold, hits);
			sort(hits.begin(), hits.end());
			append_results(hits, opt, files, file_search_results);
			for(size_t i = 0; i < hits.size(); ++i){ file_query_info[hits[i].query] = defline[hits[i].query]; }       // kwage.cpp:137-143
		}
		for(size_t g = 0; g < groups.size(); ++g){ kwage_group_destroy(groups[g]); }
		kwage_shutdown(ctx);
		// ---- end of the patch: from here on the reference's own sort and writers (kwage.cpp:190-315) ------------

		for(unordered_map< size_t, deque<MatchResult> >::iterator i = command_line_search_results.begin(); i != command_line_search_results.end(); ++i){
			SORT(i->second.begin(), i->second.end());
		}
		for(unordered_map< size_t, deque<MatchResult> >::iterator i = file_search_results.begin(); i != file_search_results.end(); ++i){
			SORT(i->second.begin(), i->second.end());
		}
		const bool multiple_query_matches = (command_line_search_results.size() + file_search_results.size()) > 1;
		if(opt.output_format == SearchOptions::OUTPUT_CSV){ write_csv_header(out); }
		else{ write_json_header(out, multiple_query_matches); }
		bool first_match = true;
		write_matches(out, opt, command_line_search_results, NULL, multiple_query_matches, first_match);
		write_matches(out, opt, file_search_results, &file_query_info, multiple_query_matches, first_match);
		if(opt.output_format == SearchOptions::OUTPUT_CSV){ write_csv_footer(out); }
		else{ write_json_footer(out, multiple_query_matches); }

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
