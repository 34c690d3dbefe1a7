// kwage_amd/csrc/kwage_dbtool.cpp -- command-line front end to the database-side entry points of the C ABI
// (include/kwage_amd.h): what the reference spreads over dump_db / merge_db / the build_db step of maestro.
//
//   kwage_dbtool info <file.db|.dbz>                                header fields (dump_db.cpp:150-200 prints the same)
//   kwage_dbtool accessions <file.db|.dbz>                          run accession of every column
//   kwage_dbtool compress <in.db> <out.dbz> [threads]               raw -> deflate container (host)
//   kwage_dbtool decompress <in.dbz> <out.db>                       container -> raw (host)
//   kwage_dbtool build <out.db> <k> <log2 len> <num hash> <f.bloom>...   .bloom files -> .db   (device bit transpose)
//   kwage_dbtool repack <out.db> <in.db>...                         same-parameter files -> one wide file (device)
//   kwage_dbtool mkbloom <out.bloom> <accession> <k> <log2 len|0> <num hash|0> <seq file> [p]
//                                                                   exact k-mer set of a FASTA/FASTQ -> .bloom; 0 0 = pick the
//                                                                   parameters with optimal_bloom_param(p, default 0.25)
//   kwage_dbtool countbloom <out.bloom> <accession> <k> <min count> <seq file>... [-p p] [-l min log2] [-L max log2]
//                                                                   make_bloom_filter() with a minimum k-mer count (reference
//                                                                   default 5): counting filters sized from the number of bases,
//                                                                   parameters from optimal_bloom_param; exit code 3 = INVALID
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kwage_amd.h"

static int die(const char *what)
{
	fprintf(stderr, "kwage_dbtool: %s: %s\n", what, kwage_last_error());
	return 1;
}

static int usage()
{
	fprintf(stderr,
		"usage: kwage_dbtool info|accessions <file>\n"
		"       kwage_dbtool compress <in.db> <out.dbz> [threads] | decompress <in.dbz> <out.db>\n"
		"       kwage_dbtool build <out.db> <k> <log2 len> <num hash> <f.bloom>...\n"
		"       kwage_dbtool repack <out.db> <in.db>...\n"
		"       kwage_dbtool mkbloom <out.bloom> <accession> <k> <log2 len|0> <num hash|0> <seq file> [p]\n"
		"       kwage_dbtool countbloom <out.bloom> <accession> <k> <min count> <seq file>... [-p p] [-l min log2] [-L max log2]\n");
	return 2;
}

int main(int argc, char **argv)
{
	if(argc < 3){ return usage(); }
	const std::string cmd = argv[1];

	if(cmd == "info" || cmd == "accessions"){
		kwage_db_header h;
		if(kwage_db_read_header(argv[2], &h)){ return die("read header"); }
		if(cmd == "info"){
			printf("magic\t0x%08x\nversion\t%u\ncrc32\t0x%08x\nkmer_len\t%u\nnum_hash\t%u\nlog_2_filter_len\t%u\nnum_filter\t%u\n"
			       "hash_func\t%d\ncompression\t%u\ninfo_start\t%llu\n", h.magic, h.version, h.crc32, h.kmer_len, h.num_hash,
			       h.log_2_filter_len, h.num_filter, h.hash_func, h.compression, (unsigned long long)h.info_start);
			return 0;
		}
		kwage_dbinfo *d = NULL;
		if(kwage_dbinfo_open(argv[2], &d)){ return die("read metadata"); }
		char buf[64];
		for(uint32_t j = 0; j < kwage_dbinfo_num_filter(d); ++j){
			if(kwage_dbinfo_csv_string(d, j, buf, sizeof(buf))){ kwage_dbinfo_close(d); return die("read FilterInfo"); }
			printf("%u\t%s\n", j, buf);
		}
		kwage_dbinfo_close(d);
		return 0;
	}
	if(cmd == "compress" && argc >= 4){
		return kwage_db_compress(argv[2], argv[3], argc > 4 ? (uint32_t)atoi(argv[4]) : 0) ? die("compress") : 0;
	}
	if(cmd == "decompress" && argc == 4){
		return kwage_db_decompress(argv[2], argv[3]) ? die("decompress") : 0;
	}

	// the remaining commands run on the device
	if(cmd != "build" && cmd != "repack" && cmd != "mkbloom" && cmd != "countbloom"){ return usage(); }
	kwage_ctx *ctx = NULL;
	const char *dev = getenv("KWAGE_DEVICE");
	if(kwage_init(dev ? atoi(dev) : 0, &ctx)){ return die("init"); }
	int rc = 2;
	if(cmd == "build" && argc >= 7){
		kwage_params p = {(uint32_t)atoi(argv[3]), (uint32_t)atoi(argv[5]), (uint32_t)atoi(argv[4]), KWAGE_HASH_MURMUR32};
		kwage_build_stats st;
		rc = kwage_build_db(ctx, argv[2], &p, argv + 6, (uint32_t)(argc - 6), &st);
		if(rc){ rc = die("build"); }
		else{ fprintf(stderr, "wrote %s: %llu bytes, transpose kernel %.3f ms\n", argv[2], (unsigned long long)st.db_bytes, st.transpose_kernel_ms); }
	}
	else if(cmd == "repack" && argc >= 4){
		rc = kwage_repack_db(ctx, argv[2], argv + 3, (uint32_t)(argc - 3));
		if(rc){ rc = die("repack"); }
	}
	else if(cmd == "mkbloom" && argc >= 8){
		kwage_seqfile *sf = NULL;
		if(kwage_seqfile_open(argv[7], &sf)){ kwage_shutdown(ctx); return die("open sequence file"); }
		std::string concat;
		std::vector<uint64_t> off(1, 0);
		const char *d, *s;
		uint64_t n;
		int r;
		while((r = kwage_seqfile_next(sf, &d, &s, &n)) == 1){ concat.append(s, n); off.push_back(concat.size()); }
		kwage_seqfile_close(sf);
		if(r < 0){ kwage_shutdown(ctx); return die("read sequence file"); }
		kwage_params p = {(uint32_t)atoi(argv[4]), (uint32_t)atoi(argv[6]), (uint32_t)atoi(argv[5]), KWAGE_HASH_MURMUR32};
		if(p.log_2_filter_len == 0 || p.num_hash == 0){
			kwage_batch *b = NULL;
			uint64_t distinct = 0;
			if(kwage_batch_create(ctx, concat.data(), off.data(), (uint32_t)(off.size() - 1), &b) ||
			   kwage_count_distinct_kmers(ctx, b, p.kmer_len, &distinct)){ kwage_shutdown(ctx); return die("count k-mers"); }
			kwage_batch_destroy(b);
			const float fp = argc > 8 ? (float)atof(argv[8]) : 0.25f;       // options.h:140 DEFAULT_FALSE_POSITIVE_PROBABILITY
			if(kwage_optimal_bloom_param(p.kmer_len, distinct, fp, 18, 32, &p)){ kwage_shutdown(ctx); return die("optimal_bloom_param"); }
			fprintf(stderr, "%llu distinct k-mers -> log_2_filter_len %u, num_hash %u\n", (unsigned long long)distinct, p.log_2_filter_len, p.num_hash);
		}
		kwage_sample_info si;
		memset(&si, 0, sizeof(si));
		si.run_accession = argv[3];
		si.number_of_bases = concat.size();
		rc = kwage_make_bloom(ctx, &p, concat.data(), off.data(), (uint32_t)(off.size() - 1), &si, argv[2], NULL);
		if(rc){ rc = die("mkbloom"); }
	}
	else if(cmd == "countbloom" && argc >= 7){
		float fp = 0.25f;                               // options.h:140,154-155 defaults
		uint32_t min_lg = 18, max_lg = 32;
		std::vector<const char*> files;
		for(int i = 6; i < argc; ++i){
			if(!strcmp(argv[i], "-p") && i + 1 < argc){ fp = (float)atof(argv[++i]); }
			else if(!strcmp(argv[i], "-l") && i + 1 < argc){ min_lg = (uint32_t)atoi(argv[++i]); }
			else if(!strcmp(argv[i], "-L") && i + 1 < argc){ max_lg = (uint32_t)atoi(argv[++i]); }
			else{ files.push_back(argv[i]); }
		}
		// pass 1: number of bases (the reference takes it from the SRA metadata, make_bloom.cpp:108)
		uint64_t num_bp = 0, num_reads = 0;
		const char *d, *s;
		uint64_t n;
		int r = 0;
		for(const char *f : files){
			kwage_seqfile *sf = NULL;
			if(kwage_seqfile_open(f, &sf)){ kwage_shutdown(ctx); return die("open sequence file"); }
			while((r = kwage_seqfile_next(sf, &d, &s, &n)) == 1){ num_bp += n; ++num_reads; }
			kwage_seqfile_close(sf);
			if(r < 0){ kwage_shutdown(ctx); return die("read sequence file"); }
		}
		const uint32_t logc = kwage_counting_filter_log2(num_bp);
		kwage_bloom_counter *bc = NULL;
		if(kwage_bloom_counter_create(ctx, (uint32_t)atoi(argv[4]), KWAGE_HASH_MURMUR32, (uint32_t)atoi(argv[5]), logc, max_lg, &bc)){
			kwage_shutdown(ctx); return die("countbloom");
		}
		// pass 2: fragments in file order, handed over in batches of ~64 MB
		std::string concat;
		std::vector<uint64_t> off(1, 0);
		rc = 0;
		for(size_t fi = 0; fi < files.size() && !rc; ++fi){
			kwage_seqfile *sf = NULL;
			if(kwage_seqfile_open(files[fi], &sf)){ rc = die("open sequence file"); break; }
			while(!rc && (r = kwage_seqfile_next(sf, &d, &s, &n)) == 1){
				concat.append(s, n); off.push_back(concat.size());
				if(concat.size() >= (64u << 20)){
					if(kwage_bloom_counter_add(bc, concat.data(), off.data(), (uint32_t)(off.size() - 1))){ rc = die("countbloom"); }
					concat.clear(); off.assign(1, 0);
				}
			}
			kwage_seqfile_close(sf);
			if(!rc && r < 0){ rc = die("read sequence file"); }
		}
		if(!rc && off.size() > 1 && kwage_bloom_counter_add(bc, concat.data(), off.data(), (uint32_t)(off.size() - 1))){ rc = die("countbloom"); }
		if(!rc){
			kwage_sample_info si;
			memset(&si, 0, sizeof(si));
			si.run_accession = argv[3];
			si.number_of_spots = num_reads;
			si.number_of_bases = num_bp;
			kwage_params p;
			int status = KWAGE_BLOOM_INVALID;
			kwage_bloom_counter_stats st;
			if(kwage_bloom_counter_get_stats(bc, &st) || kwage_bloom_counter_finish(bc, fp, min_lg, &si, argv[2], &p, &status)){ rc = die("countbloom"); }
			else if(status != KWAGE_BLOOM_SUCCESS){
				fprintf(stderr, "%llu bases, %llu k-mers with count >= %s: no Bloom parameters satisfy the bound (STATUS_BLOOM_INVALID)\n",
				        (unsigned long long)st.num_bp, (unsigned long long)st.num_valid_kmer, argv[5]);
				rc = 3;
			}
			else{
				fprintf(stderr, "%llu bases, counting filters 2^%u, %llu k-mers with count >= %s -> log_2_filter_len %u, num_hash %u\n",
				        (unsigned long long)st.num_bp, logc, (unsigned long long)st.num_valid_kmer, argv[5], p.log_2_filter_len, p.num_hash);
			}
		}
		kwage_bloom_counter_destroy(bc);
	}
	else{
		rc = usage();
	}
	kwage_shutdown(ctx);
	return rc;
}
