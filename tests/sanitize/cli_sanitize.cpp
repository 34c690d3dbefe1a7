// The command line's host logic under AddressSanitizer + UBSan (CPU build only), against reference-written fixtures:
//  * the report writers of cli_common.hpp (CsvReport, JsonReport: hand-formatted numbers, the percent table, metadata
//    text made once per column): the hit lists of tests/golden/*/expected_t*.csv -- written by the reference binary --
//    are read back, replayed through both writers and must give the reference's CSV and JSON files byte for byte;
//  * the option parser on a few argument vectors (usage, complaints in the reference's order, thresholds out of range);
//  * the query sources (command line and files, batches cut at a base budget) over the fixtures' query files.
// Built and run by tests/test_host_sanitizers.py.
#include "cli_common.hpp"

// device entry points the header's helpers name; nothing here runs a device
extern "C" int kwage_ctx_set_tuning(kwage_ctx*, const char*, int64_t) { return 0; }
extern "C" int kwage_device_count(void) { return 0; }
extern "C" int kwage_batch_create(kwage_ctx*, const char*, const uint64_t*, uint32_t, kwage_batch**) { return -1; }
extern "C" void kwage_batch_destroy(kwage_batch*) {}
extern "C" int kwage_bloom_bits_from_batch(kwage_ctx*, const kwage_params*, kwage_batch*, void*, uint64_t*) { return -1; }

namespace {

string slurp(const string &p)
{
	ifstream f(p, ios::binary);
	stringstream ss;
	ss << f.rdbuf();
	return ss.str();
}

int fails = 0;
void expect(bool ok, const string &what) { if(!ok){ cerr << "FAILED: " << what << endl; ++fails; } }

// "NAME",num_kmers,num_kmers_found,percent,"ACCESSION"  (the name may hold commas and quotes; the accession does not)
bool parse_csv_line(const string &ln, string &name, uint32_t &nk, uint32_t &found, string &acc)
{
	size_t p4 = ln.rfind(','); if(p4 == string::npos){ return false; }
	size_t p3 = ln.rfind(',', p4 - 1); if(p3 == string::npos){ return false; }
	size_t p2 = ln.rfind(',', p3 - 1); if(p2 == string::npos){ return false; }
	size_t p1 = ln.rfind(',', p2 - 1); if(p1 == string::npos){ return false; }
	if(p1 < 2 || ln[0] != '"' || ln[p1 - 1] != '"' || ln.size() < p4 + 3){ return false; }
	name = ln.substr(1, p1 - 2);
	nk = (uint32_t)strtoul(ln.substr(p1 + 1, p2 - p1 - 1).c_str(), nullptr, 10);
	found = (uint32_t)strtoul(ln.substr(p2 + 1, p3 - p2 - 1).c_str(), nullptr, 10);
	acc = ln.substr(p4 + 2, ln.size() - p4 - 3);
	return true;
}

void replay_case(const string &dir, const vector<string> &db_files, const vector<string> &thresholds)
{
	vector<DbInfo> infos(db_files.size());
	map<string, pair<uint32_t, uint32_t> > where;
	bool unique_acc = true;
	for(size_t i = 0; i < db_files.size(); ++i){
		string err;
		if(!infos[i].open(dir + "/" + db_files[i], err)){ expect(false, err); return; }
		for(uint32_t c = 0; c < infos[i].header.num_filter; ++c){
			FilterInfo fi;
			if(!infos[i].info(c, fi)){ expect(false, "metadata of " + db_files[i]); return; }
			unique_acc = where.emplace(fi.csv_string(), make_pair((uint32_t)i, c)).second && unique_acc;
		}
	}
	if(!unique_acc){ cerr << "(accessions repeat in " << dir << ": hits cannot be mapped back, case skipped)" << endl; return; }
	for(const string &t : thresholds){
		const string csv = slurp(dir + "/expected_t" + t + ".csv"), json = slurp(dir + "/expected_t" + t + ".json");
		expect(!csv.empty() && !json.empty(), dir + ": fixture for t = " + t);
		vector<pair<string, vector<Match> > > queries;
		stringstream lines(csv);
		string ln;
		getline(lines, ln);           // header
		while(getline(lines, ln)){
			string name, acc;
			uint32_t nk = 0, found = 0;
			if(!parse_csv_line(ln, name, nk, found, acc)){ expect(false, "unreadable line: " + ln); continue; }
			auto it = where.find(acc);
			if(it == where.end()){ expect(false, "accession not in the database: " + acc); continue; }
			Match m;
			m.num_kmers_found = found;
			m.num_query_kmer = nk;
			m.file_index = it->second.first;
			m.column = it->second.second;
			if(queries.empty() || queries.back().first != name){ queries.emplace_back(name, vector<Match>()); }
			queries.back().second.push_back(m);
		}
		for(int format = 0; format < 2; ++format){
			ostringstream out;
			unique_ptr<Report> report;
			if(format == 0){ report.reset(new CsvReport(out, infos)); }
			else{ report.reset(new JsonReport(out, strtof(t.c_str(), nullptr), infos)); }
			report->begin(queries.size());
			for(const auto &q : queries){ report->query(q.first, q.second); }
			report->end();
			expect(out.str() == (format == 0 ? csv : json), dir + " t = " + t + (format == 0 ? ": CSV" : ": JSON") + " differs from the reference's file");
		}
		// a column outside the file is refused the way the reference's reader fails
		{
			ostringstream out;
			CsvReport report(out, infos);
			Match m;
			m.num_kmers_found = 1; m.num_query_kmer = 1; m.file_index = 0; m.column = infos[0].header.num_filter;
			bool thrown = false;
			try{ report.query("x", vector<Match>(1, m)); } catch(const char*){ thrown = true; }
			expect(thrown, "column past the file's last is refused");
		}
	}
}

void many_hits()
{
	// a long report through the MiB drain, every (k-mers, found) pair of a range through the percent table twice
	vector<DbInfo> infos(0);
	ostringstream out;
	TextSink sink(out);
	PercentText csv(false), json(true);
	for(int pass = 0; pass < 2; ++pass){
		for(uint32_t nk : {0u, 1u, 3u, 7u, 970u, 4294967295u}){
			for(uint32_t f = 0; f < 2000; f += (f < 20 ? 1 : 97)){
				Match m;
				m.num_query_kmer = nk; m.num_kmers_found = f; m.file_index = 0; m.column = 0;
				csv.put(sink, m); sink.put(','); json.put(sink, m); sink.put('\n');
				sink.put((uint64_t)nk*f); sink.put((uint64_t)0xFFFFFFFFFFFFFFFFull);
				sink.drain();
			}
		}
	}
	for(int i = 0; i < 300000; ++i){ sink.put("0123456789", 10); sink.drain(); }
	sink.flush();
	expect(out.str().size() > 3000000, "text sink wrote everything");
}

void option_parser(const string &golden)
{
	auto run = [&](vector<string> args, bool want, float thr = 1.0f) {
		vector<char*> argv;
		for(string &a : args){ argv.push_back(&a[0]); }
		argv.push_back(nullptr);
		Cli cli;
		vector<string> dbs;
		optind = 1;
		const bool got = read_command_line((int)args.size(), argv.data(), cli, dbs);
		expect(got == want, "command line '" + (args.size() > 1 ? args[1] : string()) + " ...' accepted: " + (got ? "yes" : "no"));
		if(got && want){ expect(cli.threshold == thr && !dbs.empty(), "threshold and database files parsed"); }
	};
	const string db = golden + "/multi/dbs", q = golden + "/multi/reads.fastq";
	run({"kwage"}, false);
	run({"kwage", "-d", db, "-i", q}, true);
	run({"kwage", "-d", db, "-i", q, "-t", "0.7", "--o.csv"}, true, 0.7f);
	run({"kwage", "-d", db, "ACGTACGTACGTACGTACGTACGTACGTACGTACGT", "-t", ".5"}, true, 0.5f);
	run({"kwage", "-d", db, "-i", q, "-t", "0"}, false);
	run({"kwage", "-d", db, "-i", q, "-t", "1.5"}, false);
	run({"kwage", "-d", db, "-i", q, "-t", "abc"}, false);
	run({"kwage", "-d", db, "-i", golden + "/manifest.json"}, false);
	run({"kwage", "-d", golden + "/zslice", "-i", q}, false);
	run({"kwage", "-i", q}, false);
	run({"kwage", "-d", db}, false);
	run({"kwage", "-d", db, "-i", q, "--unknown-flag", "-?"}, false);
}

void query_sources(const string &golden)
{
	const vector<string> files = {golden + "/multi/reads.fastq", golden + "/multi/contigs.fa.gz", golden + "/k32/q.fna"};
	size_t whole = 0;
	for(uint64_t budget : {(uint64_t)1, (uint64_t)300, (uint64_t)100000, (uint64_t)1 << 30}){
		FileQueries src(files);
		QueryBatch b;
		size_t n = 0, bases = 0;
		while(src.fill(b, budget)){
			expect(b.offsets.size() == b.size() + 1 && b.offsets.back() == b.bases.size() && b.deflines.size() == b.size(), "batch bookkeeping");
			n += b.size();
			bases += b.bases.size();
			b = QueryBatch();
		}
		if(!whole){ whole = n*1000003 + bases; }
		expect(n > 0 && whole == n*1000003 + bases, "the same queries whatever the batch budget");
		FileQueries again(files);
		size_t n2 = 0;
		{
			PrefetchedQueries ahead(again, budget);          // (a reader thread behind it)
			QueryBatch c;
			while(ahead.fill(c, budget)){ n2 += c.size(); c = QueryBatch(); }
		}
		expect(n2 == n, "prefetched source gives the same queries");
	}
	const vector<string> typed = {"ACGT", "", "NNNNACGTTTGACCA", string(5000, 'A')};
	CommandLineQueries cl(typed);
	QueryBatch b;
	size_t n = 0;
	while(cl.fill(b, 100)){ n += b.size(); b = QueryBatch(); }
	expect(n == typed.size(), "command-line queries");
}

}  // namespace

int main(int argc, char **argv)
{
	if(argc < 2){ cerr << "usage: cli_sanitize GOLDEN_DIR" << endl; return 1; }
	const string golden = argv[1];
	try{
		replay_case(golden + "/basic", {"db/basic.db"}, {"1.0", "0.8", "0.5", "0.05", "0.0001"});
		replay_case(golden + "/k32", {"k32.db"}, {"1.0", "0.6"});
		many_hits();
		option_parser(golden);
		query_sources(golden);
	}
	catch(const char *e){ cerr << "exception: " << e << endl; return 2; }
	catch(const string &e){ cerr << "exception: " << e << endl; return 2; }
	if(fails){ return 1; }
	cout << "command-line host logic under the sanitizers: reports replayed byte for byte, no report" << endl;
	return 0;
}
