/* examples/search_example.c -- the C ABI from plain C (C99): load a directory's worth of `.db` files of one
 * parameter set into HBM, search a few sequences, print (query, column, hits).
 *
 *   gcc -std=c99 -Iinclude examples/search_example.c -Lkwage_amd/lib -lkwage_amd -Wl,-rpath,$PWD/kwage_amd/lib -o search_example
 *   ./search_example <threshold> <file1.db> [file2.db ...] -- <SEQ1> [SEQ2 ...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kwage_amd.h"

#define CHECK(call) do { if((call) != KWAGE_OK){ fprintf(stderr, "%s failed: %s\n", #call, kwage_last_error()); return 1; } } while(0)

int main(int argc, char **argv)
{
	int n_db = 0, i, sep = -1;
	float threshold;
	kwage_ctx *ctx = NULL;
	kwage_group *grp = NULL;
	kwage_batch *batch = NULL;
	kwage_result *res = NULL;
	kwage_db_header h0;
	kwage_params p;
	uint64_t span_bytes = 0, *offsets, k;
	char *concat;
	size_t total = 0;

	if(argc < 5){ fprintf(stderr, "usage: %s <threshold> <db>... -- <seq>...\n", argv[0]); return 2; }
	threshold = (float)atof(argv[1]);
	for(i = 2; i < argc; ++i){ if(strcmp(argv[i], "--") == 0){ sep = i; break; } }
	if(sep < 3 || sep == argc - 1){ fprintf(stderr, "need at least one database and one sequence\n"); return 2; }
	n_db = sep - 2;

	CHECK(kwage_init(0, &ctx));
	CHECK(kwage_db_read_header(argv[2], &h0));
	for(i = 0; i < n_db; ++i){
		kwage_db_header h;
		CHECK(kwage_db_read_header(argv[2 + i], &h));
		span_bytes = (span_bytes + 15)/16*16 + ((uint64_t)h.num_filter + 7)/8;      /* blocks are 16-byte aligned */
	}
	p.kmer_len = h0.kmer_len; p.num_hash = h0.num_hash; p.log_2_filter_len = h0.log_2_filter_len; p.hash_func = h0.hash_func;
	CHECK(kwage_group_create(ctx, &p, span_bytes*8, &grp));
	for(i = 0; i < n_db; ++i){
		uint64_t first; uint32_t nf;
		CHECK(kwage_group_add_db_file(grp, argv[2 + i], &first, &nf));
		printf("# %s: columns %llu..%llu\n", argv[2 + i], (unsigned long long)first, (unsigned long long)(first + nf - 1));
	}
	CHECK(kwage_group_finalize(grp));

	{
		const int n_seq = argc - sep - 1;
		offsets = (uint64_t*)calloc((size_t)n_seq + 1, sizeof(uint64_t));
		for(i = 0; i < n_seq; ++i){ total += strlen(argv[sep + 1 + i]); offsets[i + 1] = total; }
		concat = (char*)malloc(total + 1);
		concat[0] = 0;
		for(i = 0; i < n_seq; ++i){ strcat(concat, argv[sep + 1 + i]); }
		CHECK(kwage_batch_create(ctx, concat, offsets, (uint32_t)n_seq, &batch));
	}
	CHECK(kwage_search(grp, batch, threshold, KWAGE_SEARCH_EARLY_EXIT, &res));
	for(k = 0; k < res->n_hits; ++k){
		printf("query %u\tcolumn %u\t%u / %u k-mers\n", res->hits[k].query, res->hits[k].column, res->hits[k].num_match,
		       res->num_query_kmer[res->hits[k].query]);
	}
	printf("# %llu hits, %llu bit-tests\n", (unsigned long long)res->n_hits, (unsigned long long)res->bit_tests);
	kwage_result_free(res);
	kwage_batch_destroy(batch);
	kwage_group_destroy(grp);
	kwage_shutdown(ctx);
	free(concat); free(offsets);
	return 0;
}
