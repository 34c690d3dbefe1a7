// Host side of the C ABI under AddressSanitizer + UBSan (CPU build only: g++ -fsanitize=address,undefined on host.cpp).
// Walks the reference-written fixtures of tests/golden through every host entry point, then the same files DAMAGED
// (truncated at every kind of boundary, bytes flipped): each call may fail, none may touch memory it does not own.
// Built and run by tests/test_host_sanitizers.py; exit code 0 and no sanitizer report is the pass criterion.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>
#include <iterator>

#include "kwage_amd.h"

// the three device entry points host.cpp's kwage_make_bloom calls: not part of this build (the call fails, as without a GPU)
extern "C" int kwage_batch_create(kwage_ctx*, const char*, const uint64_t*, uint32_t, kwage_batch**) { return -1; }
extern "C" void kwage_batch_destroy(kwage_batch*) {}
extern "C" int kwage_bloom_bits_from_batch(kwage_ctx*, const kwage_params*, kwage_batch*, void*, uint64_t*) { return -1; }

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static std::vector<unsigned char> slurp(const std::string &p)
{
	std::ifstream f(p, std::ios::binary);
	return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string &p, const std::vector<unsigned char> &b, size_t n)
{
	std::ofstream f(p, std::ios::binary | std::ios::trunc);
	f.write((const char*)b.data(), (std::streamsize)n);
}

static unsigned long calls = 0, failures = 0;
#define CALL(x) do { ++calls; if((x) != 0){ ++failures; (void)kwage_last_error(); } } while(0)

static void walk_db(const std::string &path, bool expect_ok)
{
	kwage_db_header h;
	memset(&h, 0, sizeof(h));
	const int rc = kwage_db_read_header(path.c_str(), &h);
	++calls;
	if(rc != 0){ ++failures; if(expect_ok){ fprintf(stderr, "header of %s: %s\n", path.c_str(), kwage_last_error()); exit(2); } }
	// slices: a few in range, repeated, unordered; one out of range
	if(rc == 0 && h.log_2_filter_len <= 24){
		const uint64_t nrows = 1ull << h.log_2_filter_len, ss = ((uint64_t)h.num_filter + 7)/8;
		for(uint64_t n : {(uint64_t)1, (uint64_t)70, (uint64_t)5000}){
			if(n*ss > (256ull << 20)){ continue; }          // (a damaged num_filter: the caller sizes this buffer, not the library)
			std::vector<uint32_t> rows(n);
			for(auto &r : rows){ r = (uint32_t)(rnd() % nrows); }
			std::vector<unsigned char> out(n*ss + 1, 0xA5);
			const int r2 = kwage_db_read_slices(path.c_str(), rows.data(), n, out.data());
			++calls;
			if(r2 != 0){ ++failures; if(expect_ok){ fprintf(stderr, "slices of %s: %s\n", path.c_str(), kwage_last_error()); exit(2); } }
			if(out[n*ss] != 0xA5){ fprintf(stderr, "kwage_db_read_slices wrote past its buffer\n"); exit(3); }
		}
		uint32_t bad[2] = {0, (uint32_t)nrows};
		if(2*ss <= (256ull << 20)){
			std::vector<unsigned char> out(2*ss);
			CALL(kwage_db_read_slices(path.c_str(), bad, 2, out.data()));
		}
	}
	// metadata strings into buffers of every awkward size
	kwage_dbinfo *d = nullptr;
	++calls;
	if(kwage_dbinfo_open(path.c_str(), &d) != 0){ ++failures; (void)kwage_last_error(); if(expect_ok){ fprintf(stderr, "dbinfo of %s\n", path.c_str()); exit(2); } return; }
	const uint32_t nf = kwage_dbinfo_num_filter(d);
	for(uint32_t c = 0; c < nf + 2; c += (nf > 64 ? nf/32 : 1)){
		for(size_t len : {(size_t)0, (size_t)1, (size_t)5, (size_t)64, (size_t)4096}){
			std::vector<char> buf(len + 1, 'Z');
			(void)kwage_dbinfo_csv_string(d, c, len ? buf.data() : nullptr, len);
			if(buf[len] != 'Z'){ fprintf(stderr, "csv_string wrote past its buffer\n"); exit(3); }
			const int64_t need = kwage_dbinfo_json_string(d, c, "\t\t", len ? buf.data() : nullptr, len);
			if(buf[len] != 'Z'){ fprintf(stderr, "json_string wrote past its buffer\n"); exit(3); }
			if(need > 0 && (size_t)need < (1u << 20)){
				std::vector<char> full((size_t)need + 2, 'Z');
				(void)kwage_dbinfo_json_string(d, c, "\t\t", full.data(), (size_t)need + 1);
				if(full[(size_t)need + 1] != 'Z'){ fprintf(stderr, "json_string wrote past its buffer (exact size)\n"); exit(3); }
			}
			calls += 2;
		}
	}
	kwage_dbinfo_close(d);
}

static void walk_seqfile(const std::string &path)
{
	kwage_seqfile *f = nullptr;
	++calls;
	if(kwage_seqfile_open(path.c_str(), &f) != 0){ ++failures; (void)kwage_last_error(); return; }
	const char *def = nullptr, *seq = nullptr;
	uint64_t len = 0, total = 0;
	int rc;
	unsigned guard = 0;
	while((rc = kwage_seqfile_next(f, &def, &seq, &len)) == 1 && ++guard < 1000000){
		// touch every byte the views promise
		for(uint64_t i = 0; i < len; ++i){ total += (unsigned char)seq[i]; }
		if(def){ total += strlen(def); }
	}
	++calls;
	if(rc < 0){ ++failures; (void)kwage_last_error(); }
	kwage_seqfile_close(f);
	if(total == 0xFFFFFFFFFFFFFFFFull){ puts(""); }
}

int main(int argc, char **argv)
{
	if(argc < 3){ fprintf(stderr, "usage: host_sanitize GOLDEN_DIR SCRATCH_DIR\n"); return 1; }
	const std::string golden = argv[1], tmp = argv[2];
	if(kwage_abi_version() == 0){ return 1; }
	const std::vector<std::string> dbs = {"basic/db/basic.db", "k32/k32.db", "bloomgen/bloomgen.db", "multi/dbs/k31_L12_h3.db", "multi/dbs/a/k31_L10_h1.db",
	                                      "multi/dbs/a/deeper/k31_L10_h1_b.db", "multi/dbs/b/k15_L11_h2.DB"};
	for(const std::string &rel : dbs){
		const std::string p = golden + "/" + rel;
		walk_db(p, true);
		// the compressed container of the same file, and back
		const std::string z = tmp + "/x.dbz", back = tmp + "/back.db";
		if(kwage_db_compress(p.c_str(), z.c_str(), 3) != 0){ fprintf(stderr, "compress %s: %s\n", p.c_str(), kwage_last_error()); return 2; }
		walk_db(z, true);
		if(kwage_db_decompress(z.c_str(), back.c_str()) != 0){ fprintf(stderr, "decompress: %s\n", kwage_last_error()); return 2; }
		if(slurp(back) != slurp(p)){ fprintf(stderr, "round trip of %s differs\n", p.c_str()); return 2; }
		CALL(kwage_db_compress(z.c_str(), (tmp + "/zz.dbz").c_str(), 1));          // already compressed: refused
		// damaged copies of both layouts: cut at the header, inside the slices / offset table, inside the metadata; bytes flipped
		for(const std::string &src : {p, z}){
			const std::vector<unsigned char> good = slurp(src);
			const std::string bad = tmp + "/bad.db";
			std::vector<size_t> cuts = {0, 1, 43, 44, 45, 100, good.size()/3, good.size()/2, good.size() - good.size()/10, good.size() - 9, good.size() - 1};
			for(size_t c : cuts){
				if(c > good.size()){ continue; }
				spit(bad, good, c);
				walk_db(bad, false);
				CALL(kwage_db_decompress(bad.c_str(), (tmp + "/bad_back.db").c_str()));
				CALL(kwage_db_compress(bad.c_str(), (tmp + "/bad_z.dbz").c_str(), 2));
			}
			for(int trial = 0; trial < 40; ++trial){
				std::vector<unsigned char> b = good;
				const int flips = 1 + (int)(rnd() % 4);
				for(int k = 0; k < flips; ++k){
					// mostly the header, the offset table and the metadata (where lengths and offsets live), sometimes anywhere
					const uint64_t pick = rnd() % 4;
					size_t at = (pick == 0) ? (size_t)(rnd() % 44) : (pick == 1) ? (size_t)(44 + rnd() % 4096) % b.size()
					          : (pick == 2) ? b.size() - 1 - (size_t)(rnd() % (b.size() < 4096 ? b.size() : 4096)) : (size_t)(rnd() % b.size());
					if(at < 4){ at = 4 + at; }       // (keep the magic number: the interesting paths lie behind it)
					b[at] = (unsigned char)(rnd() & 0xFF);
				}
				spit(bad, b, b.size());
				walk_db(bad, false);
				CALL(kwage_db_decompress(bad.c_str(), (tmp + "/bad_back.db").c_str()));
			}
		}
	}
	walk_db(golden + "/multi/dbs/not_a_db.txt", false);
	walk_db(tmp + "/absent.db", false);

	// query files: the fixtures, then damaged / cut copies and random bytes (plain and behind the gzip reader)
	const std::vector<std::string> qs = {"k32/q.fna", "multi/reads.fastq", "multi/contigs.fa.gz", "basic/q.fa"};
	for(const std::string &rel : qs){
		const std::string p = golden + "/" + rel;
		const std::vector<unsigned char> good = slurp(p);
		if(good.empty()){ continue; }
		walk_seqfile(p);
		const std::string ext = (rel.size() > 3 && rel.substr(rel.size() - 3) == ".gz") ? ".fa.gz" : (rel.find("fastq") != std::string::npos ? ".fastq" : ".fa");
		const std::string bad = tmp + "/bad" + ext;
		for(int trial = 0; trial < 60; ++trial){
			std::vector<unsigned char> b = good;
			if(trial % 3 == 0){ b.resize((size_t)(rnd() % (good.size() + 1))); }
			const int flips = (int)(rnd() % 6);
			for(int k = 0; k < flips && !b.empty(); ++k){
				const size_t at = (size_t)(rnd() % b.size());
				const unsigned char repl[] = {'>', '@', '+', '\n', '\r', 0, 'N', ' ', 0xFF, (unsigned char)(rnd() & 0xFF)};
				b[at] = repl[rnd() % sizeof(repl)];
			}
			spit(bad, b, b.size());
			walk_seqfile(bad);
		}
	}
	for(int trial = 0; trial < 30; ++trial){
		std::vector<unsigned char> b((size_t)(rnd() % 6000));
		for(auto &c : b){ c = (unsigned char)(rnd() & 0xFF); }
		if(!b.empty() && trial % 2){ b[0] = (trial % 4 == 1) ? '>' : '@'; }
		for(const char *name : {"/rand.fa", "/rand.fastq"}){ spit(tmp + name, b, b.size()); walk_seqfile(tmp + name); }
	}
	walk_seqfile(tmp + "/absent.fa");

	// accession codec: every decodable string round-trips; garbage is refused or decoded, never overrun
	const char *acc[] = {"SRR1", "ERR000001", "DRR999999999", "SRR", "", "XRR12", "SRR12a", "srr12", "SRR00000000000000000000000000000001", "ERR4294967296", " SRR1", "SRR1 "};
	for(const char *a : acc){
		uint64_t v = 0;
		++calls;
		if(kwage_str_to_accession(a, &v) != 0){ ++failures; (void)kwage_last_error(); continue; }
		for(size_t len : {(size_t)0, (size_t)1, (size_t)4, (size_t)32}){
			std::vector<char> buf(len + 1, 'Z');
			(void)kwage_accession_to_str(v, len ? buf.data() : nullptr, len);
			if(buf[len] != 'Z'){ fprintf(stderr, "accession_to_str wrote past its buffer\n"); return 3; }
		}
	}
	for(int trial = 0; trial < 2000; ++trial){
		char buf[40];
		(void)kwage_accession_to_str(rnd(), buf, sizeof(buf));
		char s[24];
		const size_t n = (size_t)(rnd() % 23);
		for(size_t i = 0; i < n; ++i){ s[i] = "SERDX0123456789 ar"[rnd() % 18]; }
		s[n] = 0;
		uint64_t v;
		(void)kwage_str_to_accession(s, &v);
	}
	// thresholds and Bloom parameters at the edges of their types
	for(float t : {0.0f, 1e-9f, 0.5f, 0.99999994f, 1.0f}){ for(uint32_t n : {0u, 1u, 970u, 16777217u, 0xFFFFFFFFu}){ (void)kwage_query_threshold(t, n); } }
	for(uint64_t nk : {(uint64_t)0, (uint64_t)1, (uint64_t)1000, (uint64_t)5000000000ull}){
		for(float pr : {0.0f, 1e-6f, 0.25f, 0.999f, 1.0f}){
			kwage_params out;
			memset(&out, 0, sizeof(out));
			(void)kwage_optimal_bloom_param(31, nk, pr, 10, 32, &out);
			(void)kwage_optimal_bloom_param(31, nk, pr, 33, 5, &out);
		}
	}
	printf("host entry points under the sanitizers: %lu calls, %lu of them refused their (damaged) input, no report\n", calls, failures);
	return 0;
}
