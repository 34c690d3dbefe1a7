// tests/native/host_sanitize_driver.cpp -- host-side C ABI under AddressSanitizer + UBSan (CPU only; GPU
// sanitizers are not available on the pool).  Built by tests/test_host_sanitizers.py from
// kwage_amd/csrc/host.cpp alone (no HIP), then fed valid files and systematically damaged ones:
// every call must return a status, never crash or read out of bounds.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kwage_amd.h"
#include "host.hpp"        // DbSliceSource: the loader's view of a file (not part of the C ABI, but fed the same damaged files)

static std::vector<unsigned char> slurp(const std::string &p)
{
	std::vector<unsigned char> b;
	FILE *f = fopen(p.c_str(), "rb");
	if(!f){ return b; }
	fseek(f, 0, SEEK_END);
	b.resize((size_t)ftell(f));
	fseek(f, 0, SEEK_SET);
	if(!b.empty() && fread(b.data(), 1, b.size(), f) != b.size()){ b.clear(); }
	fclose(f);
	return b;
}

static void spit(const std::string &p, const std::vector<unsigned char> &b)
{
	FILE *f = fopen(p.c_str(), "wb");
	if(f){ if(!b.empty()){ fwrite(b.data(), 1, b.size(), f); } fclose(f); }
}

static int exercise_db(const std::string &path, bool expect_ok)
{
	kwage_db_header h;
	int rc = kwage_db_read_header(path.c_str(), &h);
	kwage_dbinfo *d = NULL;
	int rc2 = kwage_dbinfo_open(path.c_str(), &d);
	int bad = 0;
	if(rc2 == KWAGE_OK){
		char buf[64];
		const uint32_t n = kwage_dbinfo_num_filter(d);
		for(uint32_t j = 0; j < n + 2 && j < 5000; ++j){
			(void)kwage_dbinfo_csv_string(d, j, buf, sizeof(buf));
			std::vector<char> big(1 << 16);
			(void)kwage_dbinfo_json_string(d, j, "\t", big.data(), big.size());
		}
		kwage_dbinfo_close(d);
	}
	if(expect_ok && (rc != KWAGE_OK || rc2 != KWAGE_OK)){ bad = 1; }
	return bad;
}

// The slice readers the loader uses: whole row ranges, listed rows (sparse groups: first, last, middle, unordered,
// repeated) and the CRC pass, on whatever the file claims to be.
static int exercise_slices(const std::string &path, bool expect_ok)
{
	kwage::DbSliceSource src;
	std::string err;
	if(!src.open(path, err)){ return expect_ok ? 1 : 0; }
	if(src.nrows > (1ull << 22) || src.slice_size > (1u << 20)){ return 0; }      // a damaged header may claim anything: do not allocate for it
	std::vector<uint32_t> rows = {0, (uint32_t)(src.nrows - 1), (uint32_t)(src.nrows/2), 1 % (uint32_t)src.nrows, (uint32_t)(src.nrows/2), 0};
	for(uint32_t i = 0; i < 5000; ++i){ rows.push_back((uint32_t)((i*2654435761ull) % src.nrows)); }
	std::vector<unsigned char> some(rows.size()*src.slice_size + 1), all(src.nrows*src.slice_size + 1);
	const bool a = src.read_row_list(rows.data(), rows.size(), some.data(), err, 4);
	const bool b = src.read_rows(0, src.nrows, all.data(), err);
	int bad = 0;
	if(a && b){
		for(size_t i = 0; i < rows.size(); ++i){
			if(memcmp(some.data() + i*src.slice_size, all.data() + (size_t)rows[i]*src.slice_size, src.slice_size) != 0){ bad = 1; }
		}
	}
	uint32_t crc = 0;
	const bool c = src.slice_crc32(crc, err);
	if(expect_ok && (!a || !b || !c || crc != src.header.crc32)){ bad = 1; }
	const uint32_t beyond = (uint32_t)src.nrows;
	if(src.read_row_list(&beyond, 1, some.data(), err, 1)){ bad = 1; }            // out of range must be refused
	return bad;
}

static int exercise_seq(const std::string &path)
{
	kwage_seqfile *f = NULL;
	if(kwage_seqfile_open(path.c_str(), &f) != KWAGE_OK){ return 0; }
	const char *d, *s;
	uint64_t n;
	int r, count = 0;
	while((r = kwage_seqfile_next(f, &d, &s, &n)) == 1){ count += (int)strlen(d) + (int)(n > 0 ? s[n - 1] : 0); }
	kwage_seqfile_close(f);
	return count;
}

int main(int argc, char **argv)
{
	if(argc < 3){ fprintf(stderr, "usage: driver <golden dir> <tmp dir>\n"); return 2; }
	const std::string golden = argv[1], tmp = argv[2];
	int bad = 0;

	const char *dbs[] = {"/basic/db/basic.db", "/k32/k32.db", "/multi/dbs/b/k15_L11_h2.DB", "/bloomgen/bloomgen.db"};
	for(const char *rel : dbs){
		const std::string p = golden + rel;
		bad += exercise_db(p, true);
		bad += exercise_slices(p, true);
		// compressed container round trip
		const std::string z = tmp + "/x.dbz", back = tmp + "/back.db";
		if(kwage_db_compress(p.c_str(), z.c_str(), 3) != KWAGE_OK){ fprintf(stderr, "compress failed: %s\n", kwage_last_error()); ++bad; continue; }
		bad += exercise_db(z, true);
		bad += exercise_slices(z, true);
		if(kwage_db_decompress(z.c_str(), back.c_str()) != KWAGE_OK || slurp(back) != slurp(p)){ fprintf(stderr, "round trip failed for %s\n", rel); ++bad; }

		// damaged copies: truncations and byte flips of both layouts must fail cleanly or parse safely
		for(const std::string &src : {p, z}){
			const std::vector<unsigned char> orig = slurp(src);
			const size_t cuts[] = {0, 1, 20, 43, 44, 45, 100, orig.size()/2, orig.size() - 9, orig.size() - 1};
			for(size_t c : cuts){
				if(c > orig.size()){ continue; }
				std::vector<unsigned char> t(orig.begin(), orig.begin() + c);
				spit(tmp + "/cut.db", t);
				(void)exercise_db(tmp + "/cut.db", false);
				(void)exercise_slices(tmp + "/cut.db", false);
				(void)kwage_db_decompress((tmp + "/cut.db").c_str(), (tmp + "/cut_out.db").c_str());
				(void)kwage_db_compress((tmp + "/cut.db").c_str(), (tmp + "/cut_out.dbz").c_str(), 2);
			}
			unsigned s = 12345;
			for(int i = 0; i < 200; ++i){
				std::vector<unsigned char> t = orig;
				s = s*1664525u + 1013904223u;
				const size_t pos = (i < 60) ? (s % 44) : (i < 120 && t.size() > 2000 ? t.size() - 1 - (s % 2000) : s % t.size());
				s = s*1664525u + 1013904223u;
				t[pos] ^= (unsigned char)(1u << (s % 8)) | (unsigned char)(s >> 24);
				spit(tmp + "/flip.db", t);
				(void)exercise_db(tmp + "/flip.db", false);
				if(i % 4 == 0){ (void)exercise_slices(tmp + "/flip.db", false); }
				(void)kwage_db_decompress((tmp + "/flip.db").c_str(), (tmp + "/flip_out.db").c_str());
			}
		}
	}

	const char *seqs[] = {"/basic/q.fa", "/multi/reads.fastq", "/multi/contigs.fa.gz", "/k32/q.fna"};
	for(const char *rel : seqs){
		if(exercise_seq(golden + rel) <= 0){ fprintf(stderr, "no records in %s\n", rel); ++bad; }
		const std::vector<unsigned char> orig = slurp(golden + rel);
		const std::string ext = std::string(rel).substr(std::string(rel).find_last_of('/') + 1);
		for(size_t c = 0; c < orig.size(); c += orig.size()/13 + 1){
			std::vector<unsigned char> t(orig.begin(), orig.begin() + c);
			spit(tmp + "/cut_" + ext, t);
			(void)exercise_seq(tmp + "/cut_" + ext);
		}
	}

	// accession codec + threshold on odd input
	const char *acc[] = {"", "SRR", "SRR1", "srr0000000001", "SRR99999999999", "S1R2R3", "\xff\xfe", "ABCDEFGHIJ123"};
	for(const char *a : acc){ uint64_t v; char buf[8]; if(kwage_str_to_accession(a, &v) == KWAGE_OK){ (void)kwage_accession_to_str(v, buf, sizeof(buf)); } }
	char b32[32];
	(void)kwage_accession_to_str(~0ull, b32, sizeof(b32));
	(void)kwage_query_threshold(0.999999f, 0xFFFFFFFFu);
	kwage_params prm;
	(void)kwage_optimal_bloom_param(31, 1ull << 40, 0.25f, 18, 32, &prm);
	(void)kwage_optimal_bloom_param(31, 5, 1e-30f, 1, 63, &prm);

	printf("host sanitize driver: %d failure(s)\n", bad);
	return bad ? 1 : 0;
}
