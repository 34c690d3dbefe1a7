// Host side of the CLI without a GPU: how fast are FASTQ parsing, the filing of hits under their queries and the two
// report writers?  The CLI's own code is compiled in (its main() renamed), fed with a real query file, one real `.db`
// file's metadata and a synthetic hit list (every other read hits `hits_per_read` columns).
//   g++ -O3 -std=c++17 -I../../include -I../../kwage_amd/csrc cli_host_bench.cpp -o cli_host_bench -L../../kwage_amd/lib -lkwage_amd -lz -pthread
//   ./cli_host_bench file.db reads.fastq [hits_per_read=8] [keep reports as PREFIX.csv / PREFIX.json]
#define main kwage_cli_main
#include "kwage_main.cpp"
#undef main

int main(int argc, char **argv)
{
	if(argc < 3){ fprintf(stderr, "usage: cli_host_bench file.db reads.fastq [hits_per_read]\n"); return 1; }
	const uint32_t per = argc > 3 ? (uint32_t)atoi(argv[3]) : 8;
	vector<DbFileEntry> files(1);
	vector<DbInfo> infos(1);
	files[0].path = argv[1];
	string err;
	if(kwage_db_read_header(argv[1], &files[0].header) != KWAGE_OK || !infos[0].open(argv[1], err)){ fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
	ColumnMap cols;
	cols.files.push_back(&files[0]);
	cols.file_index.push_back(0);
	const uint32_t ncol = files[0].header.num_filter;

	vector<string> paths(1, argv[2]);
	FileQueries src(paths);
	Findings found;
	double t_parse = 0, t_file = 0;
	size_t reads = 0, bases = 0, hits = 0;
	QueryBatch q;
	for(;;){
		double t0 = now_s();
		const bool more = src.fill(q, 64ull << 20);
		t_parse += now_s() - t0;
		if(!more){ break; }
		reads += q.size(); bases += q.bases.size();
		vector<kwage_hit> hl;
		vector<uint32_t> nk(q.size());
		for(size_t i = 0; i < nk.size(); ++i){ nk[i] = 70 + (uint32_t)((i/2) % 7 == 3 ? i % 950 : 0); }
		for(uint32_t i = 0; i < q.size(); i += 2){
			for(uint32_t j = 0; j < per; ++j){ hl.push_back(kwage_hit{i, (uint32_t)((i*7 + j*257) % ncol), nk[i] - (j % 3)*(1 + i % 5)}); }
		}
		kwage_result res{};
		res.n_hits = hl.size(); res.hits = hl.data(); res.n_queries = (uint32_t)q.size(); res.num_query_kmer = nk.data();
		t0 = now_s();
		record_hits(res, q, cols, found);
		t_file += now_s() - t0;
		hits += hl.size();
	}
	printf("parsed %zu reads, %.1f M bases in %.3f s (%.0f MB/s of bases); %zu hits filed in %.3f s\n", reads, bases/1e6, t_parse, bases/1e6/t_parse, hits, t_file);
	for(int fmt = 0; fmt < 2; ++fmt){
		const string out_path = argc > 4 ? string(argv[4]) + (fmt ? ".json" : ".csv") : string("/tmp/cli_host_bench.out");
		ofstream out(out_path.c_str());
		unique_ptr<Report> report;
		if(fmt == 0){ report.reset(new CsvReport(out, infos)); } else { report.reset(new JsonReport(out, 0.8f, infos)); }
		const double t0 = now_s();
		report->begin(found.by_query.size());
		for(const auto &kv : found.by_query){ report->query(found.defline[kv.first], kv.second); }
		report->end();
		out.close();
		const double dt = now_s() - t0;
		ifstream sz(out_path.c_str(), ios::ate | ios::binary);
		printf("%s report: %.3f s, %.0f MB (%.0f MB/s, %.0f ns per hit)\n", fmt ? "JSON" : "CSV", dt, (double)sz.tellg()/1e6, (double)sz.tellg()/1e6/dt, dt/hits*1e9);
	}
	if(argc <= 4){ remove("/tmp/cli_host_bench.out"); }
	return 0;
}
