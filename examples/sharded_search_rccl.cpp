// examples/sharded_search_rccl.cpp -- the multi-GPU shape of the path from a C++ host: one PROCESS per GPU, the
// database files column-sharded over the ranks, every rank searches its own columns through the C ABI
// (include/kwage_amd.h) with the hits left in HBM, and ONE variable-length gather over RCCL brings the per-GPU hit
// lists to rank 0:  all_gather of (count, column span) + a grouped ncclSend / ncclRecv with exact sizes (RCCL has no
// native gatherv).  No row data crosses xGMI.  This is what kwage_amd/distributed.py does over torch.distributed; the
// library itself stays free of RCCL.
//
//   hipcc -O2 -std=c++17 -Iinclude examples/sharded_search_rccl.cpp -Lkwage_amd/lib -lkwage_amd -lrccl \
//         -Wl,-rpath,$PWD/kwage_amd/lib -o sharded_search_rccl
//   ./sharded_search_rccl <n_ranks> <threshold> <file1.db> [file2.db ...] -- <SEQ1> [SEQ2 ...]
//
// The parent forks the ranks BEFORE anything touches the GPU and only waits for them; rank r uses device r.  Output
// (rank 0): one line per hit, "query <i>\t<file>\tcolumn <j>\t<found> / <k-mers>", sorted by (query, file, column).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

#include "kwage_amd.h"

#define KW(call) do { if((call) != KWAGE_OK){ fprintf(stderr, "[rank %d] %s failed: %s\n", rank, #call, kwage_last_error()); return 1; } } while(0)
#define HIP(call) do { hipError_t e_ = (call); if(e_ != hipSuccess){ fprintf(stderr, "[rank %d] %s failed: %s\n", rank, #call, hipGetErrorString(e_)); return 1; } } while(0)
#define NCCL(call) do { ncclResult_t r_ = (call); if(r_ != ncclSuccess){ fprintf(stderr, "[rank %d] %s failed: %s\n", rank, #call, ncclGetErrorString(r_)); return 1; } } while(0)

namespace {

struct FileShare { int file; uint64_t first_column; uint32_t num_filter; };

// Whole files, contiguous, balanced by column count: a file belongs to the rank that owns its middle column (the rule
// of the kwage CLI's KWAGE_DEVICES mode and of kwage_amd.distributed.partition_files).  Every rank can compute every
// rank's share -- and the byte layout of its group -- from the headers alone.
std::vector<FileShare> share_of(int r, int n_ranks, const std::vector<kwage_db_header> &hdr, uint64_t *span_bytes)
{
	uint64_t total = 0, before = 0, span = 0;
	for(const auto &h : hdr){ total += h.num_filter; }
	std::vector<FileShare> out;
	for(size_t f = 0; f < hdr.size(); ++f){
		const uint64_t nf = hdr[f].num_filter;
		const int owner = (int)std::min<uint64_t>((uint64_t)n_ranks - 1, (uint64_t)(((long double)before + nf/2.0L)*n_ranks/std::max<uint64_t>(total, 1)));
		if(owner == r){
			span = (span + 15)/16*16;                      // every file starts at a 16-byte aligned byte column
			out.push_back(FileShare{(int)f, span*8, (uint32_t)nf});
			span += (nf + 7)/8;
		}
		before += nf;
	}
	*span_bytes = span;
	return out;
}

int run_rank(int rank, int n_ranks, const std::string &id_path, float threshold, const std::vector<std::string> &db, const std::vector<std::string> &seqs)
{
	// ---- this rank's columns ---------------------------------------------------------------------------------
	std::vector<kwage_db_header> hdr(db.size());
	for(size_t f = 0; f < db.size(); ++f){ KW(kwage_db_read_header(db[f].c_str(), &hdr[f])); }
	for(const auto &h : hdr){
		if(h.kmer_len != hdr[0].kmer_len || h.num_hash != hdr[0].num_hash || h.log_2_filter_len != hdr[0].log_2_filter_len || h.hash_func != hdr[0].hash_func){
			fprintf(stderr, "this example takes database files of ONE parameter set\n");
			return 2;
		}
	}
	kwage_ctx *ctx = nullptr;
	KW(kwage_init(rank, &ctx));
	uint64_t span_bytes = 0;
	const std::vector<FileShare> mine = share_of(rank, n_ranks, hdr, &span_bytes);
	kwage_params p = {hdr[0].kmer_len, hdr[0].num_hash, hdr[0].log_2_filter_len, hdr[0].hash_func};
	kwage_group *grp = nullptr;
	if(!mine.empty()){
		KW(kwage_group_create(ctx, &p, span_bytes*8, &grp));
		std::vector<const char*> paths;
		for(const auto &s : mine){ paths.push_back(db[s.file].c_str()); }
		std::vector<uint64_t> first(paths.size());
		KW(kwage_group_add_db_files(grp, paths.data(), (uint32_t)paths.size(), first.data(), nullptr));
		for(size_t i = 0; i < mine.size(); ++i){
			if(first[i] != mine[i].first_column){ fprintf(stderr, "[rank %d] layout mismatch\n", rank); return 1; }
		}
		KW(kwage_group_finalize(grp));
	}

	// ---- the queries, replicated on every rank (kilobytes) ---------------------------------------------------------
	std::string concat;
	std::vector<uint64_t> off(1, 0);
	for(const auto &s : seqs){ concat += s; off.push_back(concat.size()); }
	kwage_batch *batch = nullptr;
	KW(kwage_batch_create(ctx, concat.data(), off.data(), (uint32_t)seqs.size(), &batch));

	// ---- local search, hits stay in HBM: [n][3] uint32 (query, local column, num_match) ---------------------------
	uint64_t cap = 4096, n_hits = 0;
	uint32_t *d_hits = nullptr, *d_nkmer = nullptr;
	HIP(hipMalloc((void**)&d_nkmer, std::max<size_t>(seqs.size(), 1)*sizeof(uint32_t)));
	HIP(hipMemset(d_nkmer, 0, std::max<size_t>(seqs.size(), 1)*sizeof(uint32_t)));
	HIP(hipMalloc((void**)&d_hits, cap*sizeof(kwage_hit)));
	if(grp){
		for(;;){
			KW(kwage_search_device(grp, batch, threshold, KWAGE_SEARCH_EARLY_EXIT, d_hits, cap, &n_hits, d_nkmer));
			if(n_hits <= cap){ break; }
			HIP(hipFree(d_hits));                          // rare: more hits than room; grow and search again
			cap = n_hits + n_hits/4;
			HIP(hipMalloc((void**)&d_hits, cap*sizeof(kwage_hit)));
		}
	}

	// ---- communicator: rank 0 publishes the unique id through a file the parent named --------------------------------
	ncclUniqueId id;
	if(rank == 0){
		NCCL(ncclGetUniqueId(&id));
		const std::string tmp = id_path + ".tmp";
		FILE *f = fopen(tmp.c_str(), "wb");
		if(!f || fwrite(&id, sizeof(id), 1, f) != 1){ fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
		fclose(f);
		rename(tmp.c_str(), id_path.c_str());
	}
	else{
		FILE *f = nullptr;
		for(int tries = 0; tries < 3000 && !(f = fopen(id_path.c_str(), "rb")); ++tries){ usleep(10000); }
		if(!f || fread(&id, sizeof(id), 1, f) != 1){ fprintf(stderr, "[rank %d] no unique id from rank 0\n", rank); return 1; }
		fclose(f);
	}
	ncclComm_t comm;
	NCCL(ncclCommInitRank(&comm, n_ranks, id, rank));
	hipStream_t stream;
	HIP(hipStreamCreate(&stream));

	// ---- the gatherv: counts (and column spans) to everyone, then hit records with exact sizes to rank 0 ----------------
	uint64_t *d_meta = nullptr;                           // [n_ranks][2]: hits, column span of the rank's group
	HIP(hipMalloc((void**)&d_meta, (size_t)n_ranks*2*sizeof(uint64_t)));
	const uint64_t my_meta[2] = {n_hits, span_bytes*8};
	HIP(hipMemcpy(d_meta + 2*rank, my_meta, sizeof(my_meta), hipMemcpyHostToDevice));
	NCCL(ncclAllGather(d_meta + 2*rank, d_meta, 2, ncclUint64, comm, stream));
	HIP(hipStreamSynchronize(stream));
	std::vector<uint64_t> meta((size_t)n_ranks*2);
	HIP(hipMemcpy(meta.data(), d_meta, meta.size()*sizeof(uint64_t), hipMemcpyDeviceToHost));
	uint64_t total_hits = 0;
	for(int r = 0; r < n_ranks; ++r){ total_hits += meta[2*r]; }
	uint32_t *d_all = nullptr;
	if(rank == 0){ HIP(hipMalloc((void**)&d_all, std::max<uint64_t>(total_hits, 1)*sizeof(kwage_hit))); }
	NCCL(ncclGroupStart());
	if(rank == 0){
		uint64_t at = meta[0];                            // rank 0's own records go first (device-to-device copy below)
		for(int r = 1; r < n_ranks; ++r){
			if(meta[2*r]){ NCCL(ncclRecv(d_all + at*3, meta[2*r]*3, ncclUint32, r, comm, stream)); }
			at += meta[2*r];
		}
	}
	else if(n_hits){
		NCCL(ncclSend(d_hits, n_hits*3, ncclUint32, 0, comm, stream));
	}
	NCCL(ncclGroupEnd());
	if(rank == 0 && n_hits){ HIP(hipMemcpyAsync(d_all, d_hits, n_hits*sizeof(kwage_hit), hipMemcpyDeviceToDevice, stream)); }
	HIP(hipStreamSynchronize(stream));

	// ---- rank 0: columns are disjoint per rank, so the merge is a concatenation; map back to (file, column) and print ---
	if(rank == 0){
		std::vector<kwage_hit> hits(total_hits);
		std::vector<uint32_t> nkmer(seqs.size());
		if(total_hits){ HIP(hipMemcpy(hits.data(), d_all, total_hits*sizeof(kwage_hit), hipMemcpyDeviceToHost)); }
		if(!seqs.empty()){
			// num_query_kmer is the same on every rank that holds a group (same queries, same k)
			if(grp){ HIP(hipMemcpy(nkmer.data(), d_nkmer, seqs.size()*sizeof(uint32_t), hipMemcpyDeviceToHost)); }
		}
		struct Line { uint32_t query; int file; uint32_t column, found; };
		std::vector<Line> lines;
		uint64_t at = 0;
		for(int r = 0; r < n_ranks; ++r){
			uint64_t sb = 0;
			const std::vector<FileShare> share = share_of(r, n_ranks, hdr, &sb);
			for(uint64_t i = 0; i < meta[2*r]; ++i, ++at){
				const kwage_hit &h = hits[at];
				size_t lo = 0;
				while(lo + 1 < share.size() && share[lo + 1].first_column <= h.column){ ++lo; }
				lines.push_back(Line{h.query, share[lo].file, (uint32_t)(h.column - share[lo].first_column), h.num_match});
			}
		}
		std::sort(lines.begin(), lines.end(), [](const Line &a, const Line &b) {
			return a.query != b.query ? a.query < b.query : a.file != b.file ? a.file < b.file : a.column < b.column; });
		for(const auto &l : lines){
			printf("query %u\t%s\tcolumn %u\t%u / %u\n", l.query, db[l.file].c_str(), l.column, l.found, nkmer[l.query]);
		}
		printf("# %llu hits gathered from %d rank(s) over RCCL\n", (unsigned long long)total_hits, n_ranks);
		(void)hipFree(d_all);
	}
	NCCL(ncclCommDestroy(comm));
	(void)hipFree(d_meta); (void)hipFree(d_hits); (void)hipFree(d_nkmer);
	(void)hipStreamDestroy(stream);
	kwage_batch_destroy(batch);
	if(grp){ kwage_group_destroy(grp); }
	kwage_shutdown(ctx);
	return 0;
}

}  // namespace

int main(int argc, char **argv)
{
	int sep = -1;
	for(int i = 3; i < argc; ++i){ if(strcmp(argv[i], "--") == 0){ sep = i; break; } }
	if(argc < 6 || sep < 4 || sep == argc - 1){
		fprintf(stderr, "usage: %s <n_ranks> <threshold> <file.db>... -- <SEQ>...\n", argv[0]);
		return 2;
	}
	const int n_ranks = atoi(argv[1]);
	const float threshold = (float)atof(argv[2]);
	if(n_ranks < 1 || n_ranks > 64){ fprintf(stderr, "n_ranks must be 1..64\n"); return 2; }
	const std::vector<std::string> db(argv + 3, argv + sep), seqs(argv + sep + 1, argv + argc);
	char id_path[] = "/tmp/kwage_rccl_id_XXXXXX";
	const int fd = mkstemp(id_path);
	if(fd < 0){ perror("mkstemp"); return 1; }
	close(fd);
	unlink(id_path);                                       // rank 0 creates it when the id is complete
	setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);          // dmabuf IPC (what RCCL needs on this host driver)
	// the parent touches no GPU: fork first, then every child initialises HIP for its own device
	std::vector<pid_t> kids;
	for(int r = 0; r < n_ranks; ++r){
		const pid_t pid = fork();
		if(pid < 0){ perror("fork"); return 1; }
		if(pid == 0){ const int rc = run_rank(r, n_ranks, id_path, threshold, db, seqs); fflush(nullptr); _exit(rc); }
		kids.push_back(pid);
	}
	int rc = 0;
	for(pid_t k : kids){
		int st = 0;
		waitpid(k, &st, 0);
		if(!WIFEXITED(st) || WEXITSTATUS(st) != 0){ rc = 1; }
	}
	unlink(id_path);
	return rc;
}
