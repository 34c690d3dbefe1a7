// Unit checks of the CLI's host-side pieces that need no GPU: the read-ahead query source, the filing of hits, the
// number formatting of the report writers.  The CLI's translation unit is compiled in with its main() renamed.
//   g++ -std=c++17 -fsanitize=address,undefined -I include -I kwage_amd/csrc tests/native/cli_units_driver.cpp -L kwage_amd/lib -lkwage_amd -lz -pthread
#define main kwage_cli_main
#include "kwage_main.cpp"
#undef main

#include <atomic>
#include <random>

static int failures = 0;
#define EXPECT(cond) do{ if(!(cond)){ ++failures; fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); } }while(0)

namespace {

// n batches of one query each ("seq<i>"), optionally throwing instead of delivering batch `throw_at`
struct CountingSource : QuerySource {
	size_t n, next = 0, throw_at;
	atomic<size_t> produced{0};
	CountingSource(size_t batches, size_t fail_at = ~size_t(0)) : n(batches), throw_at(fail_at) {}
	bool fill(QueryBatch &b, uint64_t) override
	{
		b.clear();
		if(next == throw_at){ ++next; throw string("source failed"); }
		if(next >= n){ return false; }
		const string name = "read " + to_string(next);
		b.add(next, "ACGT" + to_string(next), &name);
		++next;
		++produced;
		return true;
	}
};

void read_ahead_delivers_in_order()
{
	CountingSource src(50);
	PrefetchedQueries ahead(src, 1 << 20, 2);
	QueryBatch b;
	for(size_t i = 0; i < 50; ++i){
		EXPECT(ahead.fill(b, 0));
		EXPECT(b.size() == 1 && b.ids[0] == i && b.deflines[0] == "read " + to_string(i) && b.bases == "ACGT" + to_string(i));
		if(i == 10){
			this_thread::sleep_for(chrono::milliseconds(50));
			EXPECT(src.produced <= i + 1 + 2 + 1);          // delivered + queue depth + the one being offered
		}
	}
	EXPECT(!ahead.fill(b, 0));
	EXPECT(!ahead.fill(b, 0));          // and stays exhausted
}

void read_ahead_passes_errors_on_after_the_good_batches()
{
	CountingSource src(10, 3);
	PrefetchedQueries ahead(src, 1 << 20, 2);
	QueryBatch b;
	for(size_t i = 0; i < 3; ++i){ EXPECT(ahead.fill(b, 0) && b.ids[0] == i); }
	bool thrown = false;
	try{ ahead.fill(b, 0); }
	catch(const string &e){ thrown = (e == "source failed"); }
	EXPECT(thrown);
	EXPECT(!ahead.fill(b, 0));
}

void read_ahead_can_be_dropped_early()
{
	for(int round = 0; round < 20; ++round){
		CountingSource src(1000);
		PrefetchedQueries ahead(src, 1 << 20, 2);
		QueryBatch b;
		if(round % 2){ EXPECT(ahead.fill(b, 0)); }
	}          // destructors: the reader is blocked on a full queue, or still parsing
}

void integers_and_percentages_match_printf()
{
	ostringstream out;
	TextSink to(out);
	const uint64_t edge[] = {0, 1, 9, 10, 99, 100, 4294967295ull, 4294967296ull, 18446744073709551615ull};
	string want;
	for(uint64_t v : edge){ to.put(v); to.put(' '); want += to_string(v) + " "; }
	to.flush();
	EXPECT(out.str() == want);

	mt19937_64 rng(7);
	for(int json = 0; json < 2; ++json){
		PercentText pt(json != 0);
		ostringstream o2;
		TextSink t2(o2);
		string expect;
		for(int i = 0; i < 200000; ++i){          // far more pairs than table entries: replaced entries are recomputed
			Match m{};
			m.num_query_kmer = 1 + (unsigned)(rng() % (i % 3 ? 200 : 3000000));
			m.num_kmers_found = (unsigned)(rng() % (m.num_query_kmer + 1));
			pt.put(t2, m);
			t2.put('\n');
			char buf[64];
			const float norm = 1.0f/m.num_query_kmer;
			if(json){ snprintf(buf, sizeof(buf), "%.1f", (100.0*m.num_kmers_found)*norm); }
			else{ const float p = (100.0f*m.num_kmers_found)*norm; snprintf(buf, sizeof(buf), "%.6g", (double)p); }
			expect += buf; expect += '\n';
			t2.drain();
		}
		t2.flush();
		EXPECT(o2.str() == expect);
	}
}

void hits_are_filed_under_their_queries()
{
	vector<DbFileEntry> files(3);
	files[0].first_column = 0; files[1].first_column = 128; files[2].first_column = 1000;
	ColumnMap cols;
	for(uint32_t i = 0; i < 3; ++i){ cols.files.push_back(&files[i]); cols.file_index.push_back(10 + i); }
	mt19937 rng(3);
	Findings got;
	map<size_t, vector<Match> > want;
	map<size_t, string> want_name;
	size_t next_id = 0;
	for(int pass = 0; pass < 2; ++pass){          // a second group revisits the same ids
		next_id = 0;
		for(int batch = 0; batch < 5; ++batch){
			QueryBatch q;
			q.clear();
			const size_t nq = 1 + rng() % 40;
			for(size_t i = 0; i < nq; ++i){ const string name = "q" + to_string(next_id); q.add(next_id++, "A", &name); }
			vector<kwage_hit> hl;
			vector<uint32_t> nk(nq);
			for(uint32_t i = 0; i < nq; ++i){
				nk[i] = 10 + i;
				const uint32_t nh = rng() % 4 == 0 ? 0 : rng() % 6;
				uint32_t col = rng() % 50;
				for(uint32_t j = 0; j < nh; ++j, col += 1 + rng() % 700){ hl.push_back(kwage_hit{i, col, 1 + j}); }
			}
			kwage_result res{};
			res.n_hits = hl.size(); res.hits = hl.data(); res.n_queries = (uint32_t)nq; res.num_query_kmer = nk.data();
			record_hits(res, q, cols, got);
			for(const kwage_hit &h : hl){
				Match m{};
				m.num_kmers_found = h.num_match; m.num_query_kmer = nk[h.query];
				const int f = h.column >= 1000 ? 2 : (h.column >= 128 ? 1 : 0);
				m.file_index = 10 + f; m.column = h.column - (uint32_t)files[f].first_column;
				want[q.ids[h.query]].push_back(m);
				want_name.emplace(q.ids[h.query], q.deflines[h.query]);
			}
		}
	}
	EXPECT(got.by_query.size() == want.size() && got.defline == want_name);
	for(const auto &kv : want){
		const vector<Match> &g = got.by_query[kv.first];
		EXPECT(g.size() == kv.second.size());
		for(size_t i = 0; i < g.size() && i < kv.second.size(); ++i){
			EXPECT(g[i].file_index == kv.second[i].file_index && g[i].column == kv.second[i].column
			       && g[i].num_kmers_found == kv.second[i].num_kmers_found && g[i].num_query_kmer == kv.second[i].num_query_kmer);
		}
	}
	// merging a worker's findings into an empty and into a non-empty set
	Findings a, b2;
	Findings copy1 = got, copy2 = got;
	a.absorb(copy1);
	EXPECT(a.by_query.size() == got.by_query.size() && a.defline == got.defline);
	a.absorb(copy2);
	for(const auto &kv : got.by_query){ EXPECT(a.by_query[kv.first].size() == 2*kv.second.size()); }
}

void page_cache_reader_starts_stops_and_is_reaped()
{
	char dir[] = "/tmp/kwage_reader_unit_XXXXXX";
	EXPECT(mkdtemp(dir) != nullptr);
	vector<string> paths;
	for(int f = 0; f < 3; ++f){
		paths.push_back(string(dir) + "/f" + to_string(f) + ".bin");
		ofstream o(paths.back().c_str(), ios::binary);
		const string block(1 << 20, (char)('a' + f));
		for(int i = 0; i < 3; ++i){ o << block; }
	}
	for(int mode = 0; mode < 3; ++mode){
		CacheReader r;
		r.start(paths, 2, 1 << 20);
		EXPECT(r.shared != nullptr && r.child > 0);
		if(mode == 1){ r.release(true); usleep(50000); }                          // reads (or finds resident) everything within its look-ahead
		if(mode == 2){ r.release(true); r.shared->passed = 9u << 20; usleep(50000); }       // the loader has passed everything: nothing to do
		r.finish();                                                                // mode 0: never released -- must still end at once
		EXPECT(r.shared == nullptr && r.child == -1);
		int status = 0;
		EXPECT(waitpid(-1, &status, WNOHANG) == -1);                              // no child left behind
	}
	for(const string &p : paths){ remove(p.c_str()); }
	rmdir(dir);
}

}  // namespace

int main()
{
	read_ahead_delivers_in_order();
	read_ahead_passes_errors_on_after_the_good_batches();
	read_ahead_can_be_dropped_early();
	integers_and_percentages_match_printf();
	hits_are_filed_under_their_queries();
	page_cache_reader_starts_stops_and_is_reaped();
	printf("%d failure(s)\n", failures);
	return failures ? 1 : 0;
}
