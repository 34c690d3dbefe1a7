// kwage_amd/csrc/cli_common.hpp -- what the two command-line programs share: `kwage` (kwage_main.cpp, one process) and
// `kwage_node` (kwage_node.cpp, one process per GPU).  The option surface of the reference (options.cpp:39-192), the
// query sources (kwage.cpp:116-148), the hit records the report is made from, and the CSV / JSON report writers
// (output.h:35-112) live here ONCE, so the two programs cannot drift apart: same options, same bytes.
// Included by exactly those two translation units (everything is in an unnamed namespace).
#ifndef KWAGE_AMD_CLI_COMMON_HPP
#define KWAGE_AMD_CLI_COMMON_HPP
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <exception>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <atomic>
#include <dirent.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>
#include <getopt.h>
#include <sys/stat.h>

#include "host.hpp"
#include "kwage_amd.h"

using namespace std;
using namespace kwage;

namespace {


// =====================================================================================================
// command line
// =====================================================================================================
struct Cli {
	enum Format { JSON, CSV };                  // JSON is the default (reference options.h:149)
	vector<string> db_roots, query_files, query_seqs;
	string output_path;
	float threshold = 1.0f;                     // reference options.h:148
	Format format = JSON;
	bool show_usage = false;
};

// One row per flag: how getopt sees it and what it does.  Long-only flags get codes above the char range.
struct FlagSpec {
	int code;
	const char *long_name;
	bool takes_value;
	void (*apply)(Cli &, const char *);
};

const FlagSpec FLAG_TABLE[] = {
	{'o', nullptr, true, [](Cli &c, const char *v) { c.output_path = v; }},
	{'d', nullptr, true, [](Cli &c, const char *v) { c.db_roots.push_back(v); }},
	{'i', nullptr, true, [](Cli &c, const char *v) { c.query_files.push_back(v); }},
	{'t', nullptr, true, [](Cli &c, const char *v) { c.threshold = (float)atof(v); }},     // atof -> float, as the reference stores it
	{'h', nullptr, false, [](Cli &c, const char *) { c.show_usage = true; }},
	{'?', nullptr, false, [](Cli &c, const char *) { c.show_usage = true; }},             // also what getopt reports for anything unknown
	{256, "o.csv", false, [](Cli &c, const char *) { c.format = Cli::CSV; }},
	{257, "o.json", false, [](Cli &c, const char *) { c.format = Cli::JSON; }},
};

const char *const USAGE_LINES[] = {
	"Usage for KWAGE (v. 0.4d):",                                   // version string of reference kwage.h:4
	"\t[-o <output file>] (default is stdout)",
	"\t[--o.csv (output CSV) | --o.json (output JSON)]",
	"\t[-t <search threshold>] (default is 1)",
	"\t-d <database search path> (can be repeated)",
	"\t[-i <input sequence file>] (can be repeated)",
	"\t[<DNA sequence>] (can be repeated)",
};

const char *const QUERY_SUFFIXES[] = {".fna", ".fna.gz", ".fasta", ".fasta.gz", ".fa", ".fa.gz", ".fastq", ".fastq.gz"};

// The reference accepts a query file name when the FIRST occurrence of some suffix sits at the very end
// (options.cpp:158-169): "reads.fa.fa" is refused.  Observable, so kept.
bool accepted_query_name(const string &name)
{
	return any_of(begin(QUERY_SUFFIXES), end(QUERY_SUFFIXES), [&](const char *sfx) {
		const size_t n = strlen(sfx);
		return n <= name.size() && name.find(sfx) == name.size() - n;
	});
}

// Database files: `.db` (options.cpp:30-33) and `.dbz`, which the reference's README.md:260 names but
// its option parser never accepted (this repo's compressed container).
bool is_db_file(const string &p) { return find_file_extension(p, ".db") || find_file_extension(p, ".dbz"); }

// Breadth-first walk in readdir order: regular files are reported as met, directories queued
// (reference file_util.h:30-125).
void find_database_files(const vector<string> &roots, vector<string> &out_db)
{
	deque<string> todo(roots.begin(), roots.end());
	while(!todo.empty()){
		const string p = todo.front();
		todo.pop_front();
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
			if(S_ISDIR(ds.st_mode)){ todo.push_back(name); }
			else if(S_ISREG(ds.st_mode) && is_db_file(name)){ out_db.push_back(name); }
		}
		closedir(dp);
	}
}

// Parse argv into `cli` and the list of database files.  Returns false when the program should stop
// (usage shown or a complaint printed) -- with exit status 0, like the reference.
bool read_command_line(int argc, char *argv[], Cli &cli, vector<string> &db_files)
{
	string shorts;
	vector<struct option> longs;
	for(const FlagSpec &f : FLAG_TABLE){
		if(f.long_name){ longs.push_back({f.long_name, f.takes_value ? required_argument : no_argument, nullptr, f.code}); }
		else{ shorts += (char)f.code; if(f.takes_value){ shorts += ':'; } }
	}
	longs.push_back({nullptr, 0, nullptr, 0});
	opterr = 0;
	cli.show_usage = (argc == 1);
	for(int code; (code = getopt_long(argc, argv, shorts.c_str(), longs.data(), nullptr)) != -1; ){
		const FlagSpec *f = find_if(begin(FLAG_TABLE), end(FLAG_TABLE), [&](const FlagSpec &x) { return x.code == code; });
		if(f != end(FLAG_TABLE)){ f->apply(cli, optarg); }
	}
	if(cli.show_usage){
		for(const char *line : USAGE_LINES){ cerr << line << endl; }
		return false;
	}
	cli.query_seqs.assign(argv + optind, argv + argc);          // getopt has moved the non-options to the end
	find_database_files(cli.db_roots, db_files);

	const string *bad_name = nullptr;
	for(const string &q : cli.query_files){ if(!bad_name && !accepted_query_name(q)){ bad_name = &q; } }
	// complaints in the reference's order; the first that applies ends the run
	const struct { bool failed; string text; } checks[] = {
		{db_files.empty(), "Please provide at least one database file to search (-d)"},
		{cli.query_files.empty() && cli.query_seqs.empty(), "Please provide at least one query sequence or file"},
		{bad_name != nullptr, "The query sequence file name, " + (bad_name ? *bad_name : string()) + ", does not have an allowed file extension"},
		{(cli.threshold <= 0.0) || (cli.threshold > 1.0), "Please provide: 0.0 < search threshold <= 1.0"},
	};
	for(const auto &c : checks){
		if(c.failed){ cerr << c.text << endl; return false; }
	}
	return true;
}

// =====================================================================================================
// queries, streamed
// =====================================================================================================
// One hit, before its metadata is fetched (what the reference keeps as output.h:9-33 MatchResult).
struct Match {
	unsigned int num_kmers_found;
	unsigned int num_query_kmer;
	uint32_t file_index;      // index into the list of database files
	uint32_t column;          // column within that file
};

// Everything the report needs, keyed by query id (ordered: the report walks ids ascending).
struct Findings {
	map<size_t, vector<Match> > by_query;
	map<size_t, string> defline;            // file queries only, and only those with a hit (kwage.cpp:137-143)
	void absorb(Findings &other)
	{
		if(by_query.empty() && defline.empty()){ by_query.swap(other.by_query); defline.swap(other.defline); return; }
		for(auto &kv : other.by_query){
			vector<Match> &dst = by_query[kv.first];
			if(dst.empty()){ dst.swap(kv.second); }
			else{ dst.insert(dst.end(), kv.second.begin(), kv.second.end()); }
		}
		for(auto &kv : other.defline){ defline.emplace(kv.first, std::move(kv.second)); }
	}
};

// A batch of queries as kwage_batch_create wants it, plus what maps hits back to the caller's ids.
struct QueryBatch {
	string bases;                   // concatenated sequences
	vector<uint64_t> offsets;       // n + 1
	vector<size_t> ids;
	vector<string> deflines;        // empty for command-line sequences
	size_t size() const { return ids.size(); }
	void clear() { bases.clear(); offsets.assign(1, 0); ids.clear(); deflines.clear(); }
	void add(size_t id, const string &seq, const string *defline)
	{
		bases += seq;
		offsets.push_back(bases.size());
		ids.push_back(id);
		if(defline){ deflines.push_back(*defline); }
	}
};

// Where batches come from.  fill() appends queries until the batch holds `max_bases` (always at least one
// query) and returns false once the source is exhausted and the batch is empty.
struct QuerySource {
	virtual ~QuerySource() {}
	virtual bool fill(QueryBatch &b, uint64_t max_bases) = 0;
};

const size_t MAX_QUERIES_PER_BATCH = 1u << 24;

struct CommandLineQueries : QuerySource {
	const vector<string> &seqs;
	size_t next = 0;
	explicit CommandLineQueries(const vector<string> &s) : seqs(s) {}
	bool fill(QueryBatch &b, uint64_t max_bases) override
	{
		b.clear();
		while(next < seqs.size() && b.size() < MAX_QUERIES_PER_BATCH && (b.size() == 0 || b.bases.size() + seqs[next].size() <= max_bases)){
			b.add(next, seqs[next], nullptr);          // the id is the position on the command line (kwage.cpp:119-125)
			++next;
		}
		return b.size() != 0;
	}
};

// The records of the -i files, one after the other; ids run on across files (kwage.cpp:127-147).
struct FileQueries : QuerySource {
	const vector<string> &paths;
	size_t file = 0, next_id = 0;
	SeqFile reader;
	bool open = false, held = false;            // held: reader.seq / curr_defline is a record that did not fit the last batch
	explicit FileQueries(const vector<string> &p) : paths(p) {}
	bool fill(QueryBatch &b, uint64_t max_bases) override
	{
		b.clear();
		if(max_bases <= (256ull << 20)){ b.bases.reserve(max_bases); }      // address space only until written: no regrowth copies
		string err;
		while(b.size() < MAX_QUERIES_PER_BATCH){
			if(!held){
				if(!open){
					if(file == paths.size()){ break; }
					if(!reader.open(paths[file], err)){
						cerr << err << endl;
						throw "SequenceIterator::SequenceIterator: Unable to open sequence file";
					}
					open = true;
				}
				const int r = reader.next(err);
				if(r < 0){ throw err; }
				if(r == 0){ open = false; ++file; continue; }
				held = true;
			}
			if(b.size() != 0 && b.bases.size() + reader.seq.size() > max_bases){ break; }
			b.add(next_id++, reader.seq, &reader.curr_defline);
			held = false;
		}
		return b.size() != 0;
	}
};

// Batches of another source, parsed ahead on a thread of their own: reading starts when this object is made -- before
// the group's files are loaded -- and runs beside the loading, the device's work and the filing of hits; inflating a
// .gz query file is the slowest stage of many runs.  At most `depth` finished batches wait.
struct PrefetchedQueries : QuerySource {
	QuerySource &inner;
	const uint64_t max_bases;
	const size_t depth;
	mutex lock;
	condition_variable changed;
	deque<QueryBatch> ready;
	bool finished = false, cancelled = false;
	exception_ptr failure;
	thread reader;
	PrefetchedQueries(QuerySource &source, uint64_t batch_bases, size_t queue_depth = 2)
		: inner(source), max_bases(batch_bases), depth(queue_depth), reader([this] { read_ahead(); }) {}
	~PrefetchedQueries() override
	{
		{ lock_guard<mutex> lk(lock); cancelled = true; }
		changed.notify_all();
		reader.join();
	}
	void read_ahead()
	{
		try{
			for(;;){
				QueryBatch b;
				if(!inner.fill(b, max_bases)){ break; }
				unique_lock<mutex> lk(lock);
				changed.wait(lk, [this] { return cancelled || ready.size() < depth; });
				if(cancelled){ return; }
				ready.push_back(std::move(b));
				changed.notify_all();
			}
		}
		catch(...){ failure = current_exception(); }
		lock_guard<mutex> lk(lock);
		finished = true;
		changed.notify_all();
	}
	bool fill(QueryBatch &b, uint64_t) override          // the batch size was fixed when reading began
	{
		unique_lock<mutex> lk(lock);
		changed.wait(lk, [this] { return finished || !ready.empty(); });
		if(ready.empty()){
			if(failure){ exception_ptr f = failure; failure = nullptr; rethrow_exception(f); }
			return false;
		}
		b = std::move(ready.front());
		ready.pop_front();
		changed.notify_all();
		return true;
	}
};


// =====================================================================================================
// search
// =====================================================================================================
struct DbFileEntry {
	string path;
	kwage_db_header header;
	uint64_t first_column = 0;
};

void check(int rc)
{
	if(rc != KWAGE_OK){ throw string(kwage_last_error()); }
}

// =====================================================================================================
// report (bytes as the reference's output.h:35-112 writes them; the text is data, the printer generic)
// =====================================================================================================
// With the search down to milliseconds the report is most of a hit-heavy run (2 M hits: 0.74 s CSV / 1.7 s JSON of a
// 1.8 / 2.7 s run through iostream formatting, tools/e2e_many_reads.py), so it is assembled in a memory buffer that
// goes to the stream a MiB at a time, numbers are formatted by hand, and the two things that repeat -- a column's
// metadata text (a popular sample is in many queries' lists) and the percentage of a (k-mers, found) pair -- are
// made once.
struct TextSink {
	ostream &out;
	string buf;
	explicit TextSink(ostream &o) : out(o) { buf.reserve((1u << 20) + (64u << 10)); }
	void put(char c) { buf.push_back(c); }
	void put(const char *s, size_t n) { buf.append(s, n); }
	void put(const char *s) { buf.append(s); }
	void put(const string &s) { buf.append(s); }
	void put(uint64_t v)
	{
		char tmp[24];
		char *e = tmp + sizeof(tmp), *p = e;
		do{ *--p = (char)('0' + v % 10); v /= 10; } while(v);
		buf.append(p, (size_t)(e - p));
	}
	void drain() { if(buf.size() >= (1u << 20)){ flush(); } }
	void flush() { out.write(buf.data(), (streamsize)buf.size()); buf.clear(); }
};

// The text of percent_kmers_found, remembered per (k-mers, found) pair in a small direct-mapped table.
//   CSV:  float arithmetic and the stream's default float format, 6 significant digits (output.h:43-51)
//   JSON: double arithmetic with a float reciprocal, fixed with one decimal -- the reference's stream flags stay
//         set after the threshold (output.h:82-90)
struct PercentText {
	struct Entry { uint64_t key = ~0ull; char text[24]; uint8_t len = 0; };
	vector<Entry> table;
	bool json;
	explicit PercentText(bool j) : table(4096), json(j) {}
	void put(TextSink &to, const Match &m)
	{
		const uint64_t key = ((uint64_t)m.num_query_kmer << 32) | m.num_kmers_found;
		Entry &e = table[(key*0x9E3779B97F4A7C15ull) >> 52];
		if(e.key != key){
			const float norm = m.num_query_kmer ? 1.0f/m.num_query_kmer : 0.0f;
			const int n = json ? snprintf(e.text, sizeof(e.text), "%.1f", (100.0*m.num_kmers_found)*norm)
			                   : snprintf(e.text, sizeof(e.text), "%.6g", (double)((100.0f*m.num_kmers_found)*norm));
			e.len = (uint8_t)min<int>(n, (int)sizeof(e.text) - 1);
			e.key = key;
		}
		to.put(e.text, e.len);
	}
};

// A column's metadata as the report prints it, formatted the first time the column has a hit.
struct MetadataText {
	const vector<DbInfo> &infos;
	vector<vector<string> > text;           // [file][column]
	vector<vector<char> > made;
	explicit MetadataText(const vector<DbInfo> &i) : infos(i), text(i.size()), made(i.size()) {}
	virtual ~MetadataText() {}
	virtual string format(const FilterInfo &fi) const = 0;
	const string &of(const Match &m)
	{
		if(text[m.file_index].empty()){
			text[m.file_index].resize(infos[m.file_index].header.num_filter);
			made[m.file_index].assign(infos[m.file_index].header.num_filter, 0);
		}
		if(m.column >= made[m.file_index].size()){ throw "binary_read<FilterInfo>: Unable to read FilterInfo"; }
		if(!made[m.file_index][m.column]){
			FilterInfo fi;
			if(!infos[m.file_index].info(m.column, fi)){ throw "binary_read<FilterInfo>: Unable to read FilterInfo"; }
			text[m.file_index][m.column] = format(fi);
			made[m.file_index][m.column] = 1;
		}
		return text[m.file_index][m.column];
	}
};

struct Report {
	virtual ~Report() {}
	virtual void begin(size_t queries_with_hits) = 0;
	virtual void query(const string &name, const vector<Match> &ms) = 0;
	virtual void end() = 0;
};

struct CsvReport : Report {
	TextSink to;
	PercentText percent;
	struct Accession : MetadataText {
		using MetadataText::MetadataText;
		string format(const FilterInfo &fi) const override { return fi.csv_string(); }
	} metadata;
	CsvReport(ostream &o, const vector<DbInfo> &infos) : to(o), percent(false), metadata(infos) {}
	void begin(size_t) override { to.put("query,num_kmers,num_kmers_found,percent_kmers_found,sample_metadata\n"); }
	void query(const string &name, const vector<Match> &ms) override
	{
		for(const Match &m : ms){
			to.put('"'); to.put(name); to.put("\",", 2);
			to.put((uint64_t)m.num_query_kmer); to.put(',');
			to.put((uint64_t)m.num_kmers_found); to.put(',');
			percent.put(to, m);
			to.put(",\"", 2); to.put(metadata.of(m)); to.put("\"\n", 2);
			to.drain();
		}
	}
	void end() override { to.flush(); to.out.flush(); }
};

// A pretty printer with ONE layout rule for objects and arrays: every member starts on a new line, one tab
// deeper than its container; the closing bracket of a non-empty container goes on a line of its own at the
// container's depth; an empty array closes at once.  The reference's hand-written JSON follows this rule
// throughout, including its habit of starting the document with a newline.
struct JsonPrinter {
	TextSink &to;
	struct Level { size_t members; };
	vector<Level> stack;
	int base_depth;                 // -1: the top-level list is not wrapped in [ ]
	string tabs;                    // enough of them for any depth used here
	JsonPrinter(TextSink &t, bool wrapped) : to(t), base_depth(wrapped ? 0 : -1), tabs(16, '\t')
	{
		stack.push_back(Level{0});
		if(wrapped){ to.put('['); }
	}
	size_t depth(int extra = 0) const { const int d = base_depth + (int)stack.size() - 1 + extra; return d > 0 ? (size_t)d : 0; }
	string indent(int extra = 0) const { return string(depth(extra), '\t'); }
	void member()
	{
		if(stack.back().members++){ to.put(','); }
		to.put('\n'); to.put(tabs.data(), depth(1));
	}
	void key(const char *name) { member(); to.put('"'); to.put(name); to.put("\": ", 3); }
	void open(char bracket) { to.put(bracket); stack.push_back(Level{0}); }
	void close(char bracket, bool own_line_even_if_empty)
	{
		const bool any = stack.back().members != 0;
		stack.pop_back();
		if(any || own_line_even_if_empty){ to.put('\n'); to.put(tabs.data(), depth(1)); }
		to.put(bracket);
	}
	void finish() { if(base_depth == 0){ to.put("\n]\n"); } }
};

struct JsonReport : Report {
	TextSink to;
	PercentText percent;
	string threshold_text;
	unique_ptr<JsonPrinter> js;
	// the whole "{ ... }" value of sample_metadata at the depth every result's members have (it is the same for all)
	struct Block : MetadataText {
		string inner_indent, closing_indent;
		using MetadataText::MetadataText;
		string format(const FilterInfo &fi) const override { return "{\n" + fi.json_string(inner_indent) + '\n' + closing_indent + '}'; }
	} metadata;
	JsonReport(ostream &o, float t, const vector<DbInfo> &infos) : to(o), percent(true), metadata(infos)
	{
		char buf[64];
		snprintf(buf, sizeof(buf), "%.1f", t);              // fixed, one decimal (output.h:73-75)
		threshold_text = buf;
	}
	void begin(size_t queries_with_hits) override { js.reset(new JsonPrinter(to, queries_with_hits > 1)); }     // [ ] only around several
	void query(const string &name, const vector<Match> &ms) override
	{
		JsonPrinter &j = *js;
		j.member(); j.open('{');
		j.key("query"); to.put('"'); to.put(name); to.put('"');
		j.key("threshold"); to.put(threshold_text);
		j.key("results"); j.open('[');
		for(const Match &m : ms){
			j.member(); j.open('{');
			j.key("percent_kmers_found"); percent.put(to, m);
			j.key("num_kmers"); to.put((uint64_t)m.num_query_kmer);
			j.key("num_kmers_found"); to.put((uint64_t)m.num_kmers_found);
			j.key("sample_metadata");
			if(metadata.inner_indent.empty()){ metadata.inner_indent = j.indent(2); metadata.closing_indent = j.indent(1); }
			to.put(metadata.of(m));
			j.close('}', true);
			to.drain();
		}
		j.close(']', false);
		j.close('}', true);
	}
	void end() override { if(js){ js->finish(); } to.flush(); to.out.flush(); }
};

uint64_t env_u64(const char *name, uint64_t fallback)
{
	const char *e = getenv(name);
	return e ? strtoull(e, nullptr, 10) : fallback;
}

// A command-line run loads a database once and searches it once: choosing between two candidate placements of a large
// matrix (the library's default, +3-4 % on the gather kernels) costs seconds while the driver wipes the released block,
// which such a run never earns back.  Off unless the environment asks for it.
void one_shot_placement(kwage_ctx *ctx)
{
	if(!getenv("KWAGE_GROUP_PLACEMENT_PROBE")){ (void)kwage_ctx_set_tuning(ctx, "group_placement_probe", 0); }
}

// The number of visible devices WITHOUT starting the HIP runtime in this process: the page-cache readers are forked
// after the devices have been chosen, and a process in which HIP is up (runtime threads, the open KFD) must not be
// forked.  A short-lived child asks the runtime and reports over a pipe.  -1: could not be done.
int device_count_in_child()
{
	int fd[2];
	if(pipe(fd) != 0){ return -1; }
	const pid_t pid = fork();
	if(pid < 0){ close(fd[0]); close(fd[1]); return -1; }
	if(pid == 0){
		close(fd[0]);
		const int n = kwage_device_count();
		const ssize_t w = write(fd[1], &n, sizeof(n));
		_exit(w == (ssize_t)sizeof(n) ? 0 : 1);
	}
	close(fd[1]);
	int n = -1;
	if(read(fd[0], &n, sizeof(n)) != (ssize_t)sizeof(n)){ n = -1; }
	close(fd[0]);
	int status = 0;
	(void)waitpid(pid, &status, 0);
	return n;
}

}  // namespace

#endif
