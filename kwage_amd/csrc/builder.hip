// kwage_amd/csrc/builder.hip -- database construction on the device:
// the bit transpose at the heart of the reference's build_db() (build_db.cpp:24-456, loop
// :259-315: for every set bit k of filter j, set bit j of slice k -- one get_bit/set_bit pair per
// bit, single threaded) done as a transpose of 32 x 32 bit blocks in registers on the GPU, writing a `.db` file that is
// byte-identical to the reference's for the same `.bloom` inputs.
//
// `.bloom` file = binary_write<BloomFilter> (binary_io.cpp:182-208): 1 magic byte (0xFF complete),
// BloomParam {u32 kmer_len, u32 log_2_filter_len, u32 num_hash, i32 hash_func} (bloom.h:546-556),
// u32 crc32 of the bit array, FilterInfo (binary_io.cpp:154-163), then 2^L/8 bytes of bits.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "host.hpp"
#include "internal.h"

using namespace kwage;

namespace {

// One workgroup = 8 waves = a tile of FILTERS filters x ROWS slices (TransposeTile: 64 KB in, 64 KB out).
//
// A lane holds a 32 x 32 bit block: one dword (32 consecutive slices) of each of 32 consecutive filters, loaded so
// that LD consecutive lanes read consecutive dwords of the same filter (64- or 128-byte runs per filter), transposes
// it in registers (five rounds of masked half-block swaps, ~480 integer operations for 1024 bits -- the ballot form
// this replaces spent five instructions per 64 bits and was VALU-bound at 0.15 of the HBM rate) and leaves 32 dwords
// in LDS: slice (32d + b), the four bytes of its 32 filters.  The tile is then written out as whole row segments of
// FILTERS/8 bytes, 16 bytes per thread.  LDS columns are rotated by 16 bytes per 32 slices so that the lanes of
// one store spread over the banks (rows are FILTERS/8 = 128 or 64 bytes apart).
template<int LD, int TB_WAVES> struct TransposeTile {
	static constexpr int WAVES = TB_WAVES;
	static constexpr int ROWS = 32*LD;                        // slices per tile
	static constexpr int GROUPS = TB_WAVES*(64/LD);           // 32-filter groups per tile
	static constexpr int FILTERS = 32*GROUPS;
	static constexpr int PITCH = GROUPS;                      // dwords per LDS row
	static_assert(ROWS*PITCH*4 == TB_WAVES*8*1024, "8 KB per wave");
};

__device__ inline void transpose_32x32(uint32_t (&a)[32])
{
	// a[i] bit b  <->  a[b] bit i (both LSB-first)
#pragma unroll
	for(int s = 0; s < 5; ++s){
		const int j = 16 >> s;
		const uint32_t m = s == 0 ? 0x0000FFFFu : s == 1 ? 0x00FF00FFu : s == 2 ? 0x0F0F0F0Fu : s == 3 ? 0x33333333u : 0x55555555u;
#pragma unroll
		for(int k = 0; k < 32; ++k){
			if(k & j){ continue; }
			const uint32_t t = ((a[k] >> j) ^ a[k + j]) & m;
			a[k + j] ^= t;
			a[k] ^= t << j;
		}
	}
}

// A lane's 32 dwords of tile (row0, f0): the loads are ISSUED here and nothing else touches the values on the fast path
// (filters past the last one read the last one's dword; tile_zero_past_end() clears them before the transpose), so the
// caller can leave them in flight behind other work.
template<int LD>
__device__ __forceinline__ void tile_request(uint32_t (&a)[32], const uint8_t *__restrict__ in, uint64_t in_stride, uint32_t n_filters,
                                             uint64_t chunk_rows, uint64_t row0, uint32_t first, uint32_t d)
{
	const uint64_t tile_bytes = (chunk_rows - row0 + 7)/8;    // bytes of a filter left in the chunk from the tile's first slice on
	if(tile_bytes >= 4ull*LD && (in_stride & 3) == 0 && (((uintptr_t)in) & 3) == 0){
		// the whole row range of the tile exists and dwords are aligned (uniform over the workgroup): 32 loads in flight
		// per lane, no branch between them
		const uint32_t last = first < n_filters ? std::min<uint32_t>(31u, n_filters - 1 - first) : 0u;
		const uint8_t *base = in + (uint64_t)(first < n_filters ? first : 0u)*in_stride + row0/8 + 4*d;
#pragma unroll
		for(int i = 0; i < 32; ++i){
			a[i] = *reinterpret_cast<const uint32_t*>(base + (uint64_t)std::min<uint32_t>((uint32_t)i, last)*in_stride);
		}
	}
	else{
		// the last rows of a chunk, or filters shorter than a dword: byte by byte
		const uint32_t have = tile_bytes > 4ull*d ? (uint32_t)std::min<uint64_t>(4, tile_bytes - 4ull*d) : 0;
#pragma unroll 1
		for(int i = 0; i < 32; ++i){
			const uint32_t f = first + i;
			uint32_t v = 0;
			if(f < n_filters){
				const uint8_t *src = in + (uint64_t)f*in_stride + row0/8 + 4*d;
				for(uint32_t b = 0; b < have; ++b){ v |= (uint32_t)src[b] << (8*b); }
			}
			a[i] = v;
		}
	}
}

__device__ __forceinline__ void tile_zero_past_end(uint32_t (&a)[32], uint32_t first, uint32_t n_filters)
{
#pragma unroll
	for(int i = 0; i < 32; ++i){ a[i] = first + i < n_filters ? a[i] : 0u; }
}

// A workgroup takes tiles t = blockIdx.x, + gridDim.x, ... (slice tiles fastest).  The launch has one workgroup per tile.
// (KWAGE_BUILD_PERSISTENT=1: a resident grid instead -- the next tile's 32 loads per lane are then requested before the
// current tile is written out of LDS, so that reads and writes of a CU overlap although the 64 KB tile leaves room for
// only two workgroups per CU; measured 8-10 % slower than leaving the order of the tiles to the dispatcher.)
template<int LD, int TB_WAVES>
__global__ __launch_bounds__(TB_WAVES*64) void transpose_bits_kernel(
	const uint8_t *__restrict__ in, uint64_t in_stride,   // [n_filters][in_stride bytes]: this chunk's bits
	uint32_t n_filters, uint64_t chunk_rows,              // rows in this chunk (multiple of 8)
	uint8_t *__restrict__ out, uint64_t slice_size,       // [chunk_rows][slice_size]
	uint32_t tiles_x, uint32_t n_tiles)                   // slice tiles per filter tile, tiles in all
{
	using T = TransposeTile<LD, TB_WAVES>;
	__shared__ uint32_t tile[T::ROWS*T::PITCH];

	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = threadIdx.x >> 6;
	const uint32_t d = lane % LD;                             // dword of the tile's row range
	const uint32_t g = wave*(64/LD) + lane/LD;                // 32-filter group of the tile
	constexpr uint32_t SEGS = T::PITCH/4;                     // 16-byte segments per row of the tile

	uint32_t t = blockIdx.x;
	if(t >= n_tiles){ return; }
	uint32_t a[32];
	tile_request<LD>(a, in, in_stride, n_filters, chunk_rows, (uint64_t)(t % tiles_x)*T::ROWS, (t / tiles_x)*T::FILTERS + 32*g, d);
	for(;;){
		const uint64_t row0 = (uint64_t)(t % tiles_x)*T::ROWS;
		const uint32_t f0 = (t / tiles_x)*T::FILTERS;
		tile_zero_past_end(a, f0 + 32*g, n_filters);
		transpose_32x32(a);
		{
			const uint32_t col = (g + 4*d) % T::PITCH;        // rotation: 4 dwords (16 bytes) per 32 slices
#pragma unroll
			for(int b = 0; b < 32; ++b){ tile[(32*d + b)*T::PITCH + col] = a[b]; }
		}
		__syncthreads();

		const uint32_t tn = t + gridDim.x;
		if(tn < n_tiles){
			tile_request<LD>(a, in, in_stride, n_filters, chunk_rows, (uint64_t)(tn % tiles_x)*T::ROWS, (tn / tiles_x)*T::FILTERS + 32*g, d);
		}

		// write the tile: ROWS rows x FILTERS/8 bytes; a thread moves 16 bytes
		const uint64_t col0 = f0/8;                           // first output byte of this filter block
		for(uint32_t i = threadIdx.x; i < T::ROWS*SEGS; i += TB_WAVES*64){
			const uint32_t r = i / SEGS, seg = i % SEGS;
			const uint64_t row = row0 + r;
			if(row >= chunk_rows){ continue; }
			const uint64_t cb = col0 + seg*16;
			if(cb >= slice_size){ continue; }
			const uint64_t nb = std::min<uint64_t>(16, slice_size - cb);
			uint8_t *dst = out + row*slice_size + cb;
			const uint4 v = *reinterpret_cast<const uint4*>(&tile[r*T::PITCH + 4*((seg + r/32) % SEGS)]);
			if(nb == 16 && ((((uintptr_t)dst) & 15) == 0)){ *reinterpret_cast<uint4*>(dst) = v; }
			else{
				const uint32_t w[4] = {v.x, v.y, v.z, v.w};
				for(uint64_t b = 0; b < nb; ++b){ dst[b] = (uint8_t)(w[b >> 2] >> (8*(b & 3))); }
			}
		}
		__syncthreads();                                      // the tile is free again
		if(tn >= n_tiles){ break; }
		t = tn;
	}
}

// Re-pack: OR `nbits` columns of a source block (rows of `src_width` bytes, column 0 at bit 0) into
// the destination rows starting at destination column dst_bit0.  One thread per (row, destination dword).
__global__ void pack_columns_kernel(uint32_t *dst, uint64_t dst_stride_words, uint64_t dst_bit0,
                                    const uint8_t *src, uint64_t src_width, uint64_t nbits, uint64_t nrows)
{
	const uint64_t j0 = dst_bit0/32, j1 = (dst_bit0 + nbits + 31)/32;
	const uint64_t per_row = j1 - j0;
	const uint64_t total = nrows*per_row;
	for(uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x*blockDim.x){
		const uint64_t r = i / per_row, j = j0 + i % per_row;
		const uint64_t lo = (j*32 > dst_bit0) ? j*32 : dst_bit0;
		const uint64_t hi = ((j + 1)*32 < dst_bit0 + nbits) ? (j + 1)*32 : (dst_bit0 + nbits);
		if(lo >= hi){ continue; }
		const uint64_t sbit = lo - dst_bit0;                      // first source column of this dword
		const uint32_t nb = (uint32_t)(hi - lo);
		const uint8_t *row = src + r*src_width;
		uint64_t v = 0;
		const uint64_t byte0 = sbit/8;
#pragma unroll
		for(int b = 0; b < 5; ++b){
			if(byte0 + b < src_width){ v |= (uint64_t)row[byte0 + b] << (8*b); }
		}
		v >>= (sbit % 8);
		const uint32_t bits = (uint32_t)(v & ((nb == 32) ? 0xFFFFFFFFull : ((1ull << nb) - 1)));
		if(bits){ atomicOr(dst + r*dst_stride_words + j, bits << (uint32_t)(lo - j*32)); }
	}
}

struct BloomFile {
	int fd = -1;
	const unsigned char *map = nullptr;
	size_t size = 0;
	uint32_t crc = 0;
	FilterInfo info;
	size_t bits_off = 0;
	void close_file()
	{
		if(map){ munmap((void*)map, size); map = nullptr; }
		if(fd >= 0){ close(fd); fd = -1; }
	}
};

// f(0) ... f(nparts - 1), on threads of their own where they can be had.  A thread that cannot be created
// (std::system_error -- which must never cross the C ABI this file exports) leaves its part to the caller.
template <typename F>
void run_parts(unsigned nparts, F f)
{
	std::vector<std::thread> pool;
	std::vector<unsigned> mine;
	for(unsigned t = 1; t < nparts; ++t){
		try{ pool.emplace_back(f, t); }
		catch(...){ mine.push_back(t); }
	}
	if(nparts){ f(0u); }
	for(unsigned t : mine){ f(t); }
	for(auto &th : pool){ th.join(); }
}

// crc32 of a large buffer continued from `crc`: parts in parallel, stitched with crc32_combine
// (the result is identical to one sequential crc32_z call).
uint32_t crc32_parallel(uint32_t crc, const unsigned char *buf, uint64_t len)
{
	const unsigned nthread = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
	if(len < (8u << 20) || nthread == 1){ return (uint32_t)crc32_z(crc, buf, len); }
	const uint64_t part = (len + nthread - 1)/nthread;
	std::vector<uint32_t> pc(nthread, 0);
	std::vector<uint64_t> pl(nthread, 0);
	for(unsigned t = 0; t < nthread; ++t){ pl[t] = ((uint64_t)t*part < len) ? std::min(part, len - (uint64_t)t*part) : 0; }
	run_parts(nthread, [&](unsigned t) { if(pl[t]){ pc[t] = (uint32_t)crc32_z(crc32_z(0L, Z_NULL, 0), buf + (uint64_t)t*part, pl[t]); } });
	uint64_t out = crc;
	for(unsigned t = 0; t < nthread && pl[t]; ++t){ out = crc32_combine(out, pc[t], (z_off_t)pl[t]); }
	return (uint32_t)out;
}

inline uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace

// The device context internals this file needs (defined in engine.hip).
namespace kwage { hipStream_t ctx_stream(kwage_ctx *ctx); int ctx_device(kwage_ctx *ctx); }

extern "C" int kwage_build_db(kwage_ctx *ctx, const char *out_path, const kwage_params *params,
                              const char *const *bloom_paths, uint32_t n, kwage_build_stats *stats)
{
	if(!ctx || !out_path || !params || !bloom_paths){ return fail(KWAGE_ERR_ARG, "kwage_build_db: NULL argument"); }
	if(n == 0){ return fail(KWAGE_ERR_ARG, "build_db: Empty Bloom filter inventory file"); }      // build_db.cpp:30-32
	int rc = check_params(params);
	if(rc){ return rc; }
	if(hipSetDevice(ctx_device(ctx)) != hipSuccess){ return fail(KWAGE_ERR_DEVICE, "hipSetDevice failed"); }
	hipStream_t stream = ctx_stream(ctx);

	const uint64_t filter_len = 1ull << params->log_2_filter_len;
	const uint64_t filter_bytes = (filter_len + 7)/8;
	const uint64_t slice_size = ((uint64_t)n + 7)/8;

	std::vector<BloomFile> files(n);
	auto cleanup = [&]() { for(auto &f : files){ f.close_file(); } };

	// ---- open + validate every .bloom file (build_db.cpp:48-181) ------------------------------
	for(uint32_t i = 0; i < n; ++i){
		BloomFile &f = files[i];
		f.fd = open(bloom_paths[i], O_RDONLY);
		struct stat st;
		if(f.fd < 0 || fstat(f.fd, &st) != 0){ cleanup(); return fail(KWAGE_ERR_IO, "build_db: Unable to open Bloom filter file %s", bloom_paths[i]); }
		f.size = (size_t)st.st_size;
		if(f.size < 21){ cleanup(); return fail(KWAGE_ERR_FORMAT, "build_db: %s is truncated", bloom_paths[i]); }
		f.map = (const unsigned char*)mmap(nullptr, f.size, PROT_READ, MAP_PRIVATE, f.fd, 0);
		if(f.map == MAP_FAILED){ f.map = nullptr; cleanup(); return fail(KWAGE_ERR_IO, "build_db: mmap(%s) failed", bloom_paths[i]); }
		if(f.map[0] != 0xFF){ cleanup(); return fail(KWAGE_ERR_FORMAT, "build_db: Incomplete Bloom filter %s", bloom_paths[i]); }
		const uint32_t k = rd32(f.map + 1), lg = rd32(f.map + 5), nh = rd32(f.map + 9);
		const int32_t hf = (int32_t)rd32(f.map + 13);
		if(k != params->kmer_len || lg != params->log_2_filter_len || nh != params->num_hash || hf != params->hash_func){
			cleanup();
			return fail(KWAGE_ERR_ARG, "build_db: Inconsistent Bloom parameters in %s", bloom_paths[i]);
		}
		f.crc = rd32(f.map + 17);
		size_t used = 0;
		if(!parse_filter_info(f.map + 21, f.size - 21, f.info, &used)){ cleanup(); return fail(KWAGE_ERR_FORMAT, "build_db: Error reading Bloom filter info from %s", bloom_paths[i]); }
		f.bits_off = 21 + used;
		if(f.size - f.bits_off < filter_bytes){ cleanup(); return fail(KWAGE_ERR_FORMAT, "build_db: Error reading filter bytes from %s", bloom_paths[i]); }
	}

	// ---- per-filter CRC32 (build_db.cpp:343-362), host threads; checked BEFORE anything is written
	{
		const unsigned nthread = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
		std::atomic<uint32_t> next(0), bad(0xFFFFFFFFu);
		run_parts(nthread, [&](unsigned) {
			for(uint32_t i = next++; i < n; i = next++){
				uint64_t crc = crc32_z(0L, Z_NULL, 0);
				crc = crc32_z(crc, files[i].map + files[i].bits_off, filter_bytes);
				if((uint32_t)crc != files[i].crc){ bad = i; }
			}
		});
		if(bad != 0xFFFFFFFFu){
			const uint32_t i = bad;
			cleanup();
			return fail(KWAGE_ERR_FORMAT, "build_db: One or more invalid Bloom filter CRC32 values (%s)", bloom_paths[i]);
		}
	}

	FILE *fout = fopen(out_path, "wb");
	if(!fout){ cleanup(); return fail(KWAGE_ERR_IO, "build_db: Unable to open output file for writing"); }

	// header placeholder (rewritten at the end with crc32 + info_start), kwage.h:36-46
	unsigned char hdr[DB_HEADER_BYTES];
	auto put32 = [](unsigned char *p, uint32_t v) { for(int i = 0; i < 4; ++i){ p[i] = (unsigned char)(v >> (8*i)); } };
	auto put64 = [](unsigned char *p, uint64_t v) { for(int i = 0; i < 8; ++i){ p[i] = (unsigned char)(v >> (8*i)); } };
	memset(hdr, 0, sizeof(hdr));
	put32(hdr, KWAGE_MAGIC_NUMBER); put32(hdr + 4, 2 /* CURRENT_DBFILE_VERSION */);
	put32(hdr + 12, params->kmer_len); put32(hdr + 16, params->num_hash); put32(hdr + 20, params->log_2_filter_len);
	put32(hdr + 24, n);   // hash_func stays 0 and compression NO_COMPRESSION, as build_db.cpp:189-199 leaves them
	bool ok = fwrite(hdr, 1, sizeof(hdr), fout) == sizeof(hdr);

	// ---- transpose, chunk by chunk --------------------------------------------------------------
	// chunk rows: bounded so that the input (n x rows/8) and the output (rows x slice) both stay <= 256 MiB (the output is
	// double-buffered: a chunk's CRC32 and file write run on a host thread beside the next chunk's gather, copies and kernel)
	uint64_t chunk_rows = filter_len;
	while(chunk_rows > 1024 && (chunk_rows/8*n > (256ull << 20) || chunk_rows*slice_size > (256ull << 20))){ chunk_rows /= 2; }
	// Filters lie `in_stride` apart in the device's input block, and a lane's 32 loads go to 32 consecutive filters at the
	// same offset: with a power-of-two stride they all fall on the same few memory channels (3.87 TB/s in + out; a stride of
	// an ODD number of 256-byte units: 4.43-4.50, profiles/r04_builder_transpose.txt).  KWAGE_BUILD_PAD = bytes added instead.
	const uint64_t chunk_bytes = (chunk_rows + 7)/8;
	uint64_t in_stride = (chunk_bytes + 255)/256*256;
	if((in_stride/256) % 2 == 0){ in_stride += 256; }
	if(const char *v = getenv("KWAGE_BUILD_PAD")){ in_stride = (chunk_bytes + 3)/4*4 + (uint64_t)std::max<long long>(atoll(v), 0)/4*4; }      // (never below the chunk itself)
	void *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out[2] = {nullptr, nullptr};
	hipError_t e = hipMalloc(&d_in, in_stride*n);
	if(e == hipSuccess){ e = hipMalloc(&d_out, chunk_rows*slice_size); }
	if(e == hipSuccess){ e = hipHostMalloc(&h_in, in_stride*n, hipHostMallocDefault); }
	for(int k = 0; k < 2; ++k){ if(e == hipSuccess){ e = hipHostMalloc(&h_out[k], chunk_rows*slice_size, hipHostMallocDefault); } }
	std::thread writer;                 // CRC32 + fwrite of the previous chunk
	bool writer_ok = true;
	uint32_t n_chunk = 0;
	uint32_t db_crc = 0;            // output_header.crc32 starts at 0 (build_db.cpp:192,307)
	double t_kernel_ms = 0;
	// tile shape, KWAGE_BUILD_TILE: 0 = 1024 filters x 512 slices (64-byte reads per filter, 128-byte row segments out),
	// 1 = 512 x 1024 (128-byte reads, 64-byte segments), 2 = 512 x 512 and 3 = 256 x 1024 with four waves (32 KB of LDS)
	const int tile_shape = []() { const char *v = getenv("KWAGE_BUILD_TILE"); return v ? atoi(v) : 0; }();
	// (a resident grid walking the tiles with the next tile requested before the current one is written out measured 10 %
	// SLOWER than one workgroup per tile -- profiles/r04_builder_transpose.txt -- and went in round 5)
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	if(e == hipSuccess){ e = hipEventCreate(&ev0); }
	if(e == hipSuccess){ e = hipEventCreate(&ev1); }
	for(uint64_t r0 = 0; r0 < filter_len && ok && e == hipSuccess; r0 += chunk_rows){
		const uint64_t nr = std::min(chunk_rows, filter_len - r0);
		const uint64_t nb = (nr + 7)/8;
		{	// this chunk of every filter into the pinned block (host threads: one memcpy per filter, 2048 of up to 256 KB)
			const unsigned nthread = (uint64_t)n*nb < (8u << 20) ? 1u : std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
			run_parts(nthread, [&](unsigned t) {
				for(uint32_t i = t; i < n; i += nthread){ memcpy((char*)h_in + (uint64_t)i*in_stride, files[i].map + files[i].bits_off + r0/8, nb); }
			});
		}
		e = hipMemcpyAsync(d_in, h_in, in_stride*n, hipMemcpyHostToDevice, stream);
		if(e != hipSuccess){ break; }
		(void)hipEventRecord(ev0, stream);
		auto launch = [&](auto tile_tag) {
			using T = decltype(tile_tag);
			const uint64_t tiles_x = (nr + T::ROWS - 1)/T::ROWS, tiles = tiles_x*((n + T::FILTERS - 1)/T::FILTERS);
			const uint32_t wgs = (uint32_t)tiles;
			hipLaunchKernelGGL((transpose_bits_kernel<T::ROWS/32, T::WAVES>), dim3(wgs), dim3(T::WAVES*64), 0, stream,
			                   (const uint8_t*)d_in, in_stride, n, nr, (uint8_t*)d_out, slice_size, (uint32_t)tiles_x, (uint32_t)tiles);
		};
		switch(tile_shape){
		case 1: launch(TransposeTile<32, 8>()); break;
		case 2: launch(TransposeTile<16, 4>()); break;
		case 3: launch(TransposeTile<32, 4>()); break;
		default: launch(TransposeTile<16, 8>()); break;
		}
		(void)hipEventRecord(ev1, stream);
		e = hipGetLastError();
		unsigned char *hb = (unsigned char*)h_out[n_chunk++ & 1];      // (its previous user, two chunks ago, was joined below)
		if(e == hipSuccess){ e = hipMemcpyAsync(hb, d_out, nr*slice_size, hipMemcpyDeviceToHost, stream); }
		if(e == hipSuccess){ e = hipStreamSynchronize(stream); }
		if(e != hipSuccess){ break; }
		float ms = 0;
		(void)hipEventElapsedTime(&ms, ev0, ev1);
		t_kernel_ms += ms;
		if(writer.joinable()){ writer.join(); }
		ok = writer_ok;
		if(!ok){ break; }
		auto write_chunk = [&db_crc, &writer_ok, fout, hb, bytes = nr*slice_size]() {        // (chunks are written in order: one writer at a time)
			db_crc = crc32_parallel(db_crc, hb, bytes);
			writer_ok = fwrite(hb, 1, bytes, fout) == bytes;
		};
		try{ writer = std::thread(write_chunk); }
		catch(...){ write_chunk(); }          // no thread to be had: this chunk is written here (no exception leaves this function)
	}
	if(writer.joinable()){ writer.join(); }
	ok = ok && writer_ok;
	if(d_in){ (void)hipFree(d_in); }
	if(d_out){ (void)hipFree(d_out); }
	if(h_in){ (void)hipHostFree(h_in); }
	for(int k = 0; k < 2; ++k){ if(h_out[k]){ (void)hipHostFree(h_out[k]); } }
	if(ev0){ (void)hipEventDestroy(ev0); }
	if(ev1){ (void)hipEventDestroy(ev1); }
	if(e != hipSuccess){ fclose(fout); cleanup(); return fail(KWAGE_ERR_DEVICE, "kwage_build_db: %s", hipGetErrorString(e)); }

	// ---- metadata index + records (build_db.cpp:371-416), then the final header (:421-427) -------
	const uint64_t info_start = DB_HEADER_BYTES + filter_len*slice_size;
	std::vector<unsigned char> recs;
	std::vector<uint64_t> loc(n);
	uint64_t pos = info_start + 8ull*n;
	for(uint32_t i = 0; i < n; ++i){
		loc[i] = pos;
		const size_t before = recs.size();
		pack_filter_info(files[i].info, recs);
		pos += recs.size() - before;
	}
	std::vector<unsigned char> locb(8ull*n);
	for(uint32_t i = 0; i < n; ++i){ put64(locb.data() + 8ull*i, loc[i]); }
	ok = ok && fwrite(locb.data(), 1, locb.size(), fout) == locb.size();
	ok = ok && (recs.empty() || fwrite(recs.data(), 1, recs.size(), fout) == recs.size());
	put32(hdr + 8, db_crc);
	put64(hdr + 36, info_start);
	ok = ok && fseek(fout, 0, SEEK_SET) == 0 && fwrite(hdr, 1, sizeof(hdr), fout) == sizeof(hdr);
	ok = (fclose(fout) == 0) && ok;
	cleanup();
	if(!ok){ return fail(KWAGE_ERR_IO, "build_db: Error writing database file %s", out_path); }
	if(stats){
		stats->bits_transposed = filter_len*n;
		stats->transpose_kernel_ms = (float)t_kernel_ms;
		stats->db_bytes = pos;
	}
	return KWAGE_OK;
}


// ---------------------------------------------------------------------------------------------
// Column-wise re-pack of several same-parameter `.db` files into one (what the reference's
// merge_db.cpp:268-820 does pairwise with get_bit/set_bit, without its file-size policy): columns keep
// their order (file order, then column), FilterInfo records are copied verbatim, the slice block
// CRC32 and the metadata index are recomputed.
// ---------------------------------------------------------------------------------------------
extern "C" int kwage_repack_db(kwage_ctx *ctx, const char *out_path, const char *const *in_paths, uint32_t n)
{
	if(!ctx || !out_path || !in_paths || n == 0){ return fail(KWAGE_ERR_ARG, "kwage_repack_db: bad argument"); }
	if(hipSetDevice(ctx_device(ctx)) != hipSuccess){ return fail(KWAGE_ERR_DEVICE, "hipSetDevice failed"); }
	hipStream_t stream = ctx_stream(ctx);

	std::vector<DbSliceSource> src(n);
	std::vector<DbInfo> info(n);
	std::string err;
	uint64_t total_cols = 0, max_width = 0;
	for(uint32_t i = 0; i < n; ++i){
		if(!src[i].open(in_paths[i], err) || !info[i].open(in_paths[i], err)){ return fail(KWAGE_ERR_IO, "%s", err.c_str()); }
		const kwage_db_header &h = src[i].header, &h0 = src[0].header;
		if(h.kmer_len != h0.kmer_len || h.num_hash != h0.num_hash || h.log_2_filter_len != h0.log_2_filter_len || h.hash_func != h0.hash_func){
			return fail(KWAGE_ERR_ARG, "kwage_repack_db: %s has different Bloom parameters", in_paths[i]);   // merge_db.cpp condition 1
		}
		total_cols += h.num_filter;
		max_width = std::max(max_width, src[i].slice_size);
	}
	if(total_cols == 0 || total_cols > 0xFFFFFFFFull){ return fail(KWAGE_ERR_ARG, "kwage_repack_db: bad total column count"); }
	const uint64_t nrows = src[0].nrows;
	const uint64_t out_slice = (total_cols + 7)/8;
	const uint64_t dst_words = (out_slice + 3)/4;                 // device rows are padded to whole dwords

	FILE *fout = fopen(out_path, "wb");
	if(!fout){ return fail(KWAGE_ERR_IO, "Unable to open %s for writing", out_path); }
	unsigned char hdr[DB_HEADER_BYTES];
	memset(hdr, 0, sizeof(hdr));
	bool ok = fwrite(hdr, 1, sizeof(hdr), fout) == sizeof(hdr);

	uint64_t chunk_rows = nrows;
	while(chunk_rows > 1 && (chunk_rows*dst_words*4 > (256ull << 20) || chunk_rows*max_width > (256ull << 20))){ chunk_rows /= 2; }
	void *d_dst = nullptr, *d_src = nullptr, *h_buf = nullptr;
	const uint64_t host_bytes = std::max(chunk_rows*dst_words*4, chunk_rows*max_width);
	hipError_t e = hipMalloc(&d_dst, chunk_rows*dst_words*4);
	if(e == hipSuccess){ e = hipMalloc(&d_src, chunk_rows*max_width); }
	if(e == hipSuccess){ e = hipHostMalloc(&h_buf, host_bytes, hipHostMallocDefault); }
	uint32_t crc = 0;
	std::vector<uint32_t> src_crc(n, 0);          // CRC32 of every source's slice block, checked against its header (merge_db.cpp:608-614)
	std::vector<unsigned char> packed;
	for(uint64_t r0 = 0; r0 < nrows && ok && e == hipSuccess; r0 += chunk_rows){
		const uint64_t nr = std::min(chunk_rows, nrows - r0);
		e = hipMemsetAsync(d_dst, 0, nr*dst_words*4, stream);
		uint64_t bit0 = 0;
		for(uint32_t i = 0; i < n && e == hipSuccess; ++i){
			if(!src[i].read_rows(r0, nr, (unsigned char*)h_buf, err)){ ok = false; break; }
			e = hipMemcpyAsync(d_src, h_buf, nr*src[i].slice_size, hipMemcpyHostToDevice, stream);
			if(e != hipSuccess){ break; }
			src_crc[i] = crc32_parallel(src_crc[i], (const unsigned char*)h_buf, nr*src[i].slice_size);      // while the copy runs
			const uint64_t nbits = src[i].header.num_filter;
			const uint64_t work = nr*((bit0 + nbits + 31)/32 - bit0/32);
			hipLaunchKernelGGL(pack_columns_kernel, dim3((uint32_t)std::min<uint64_t>((work + 255)/256, 8192)), dim3(256), 0, stream,
			                   (uint32_t*)d_dst, dst_words, bit0, (const uint8_t*)d_src, src[i].slice_size, nbits, nr);
			e = hipGetLastError();
			if(e == hipSuccess){ e = hipStreamSynchronize(stream); }     // h_buf is reused for the next file
			bit0 += nbits;
		}
		if(!ok || e != hipSuccess){ break; }
		e = hipMemcpyAsync(h_buf, d_dst, nr*dst_words*4, hipMemcpyDeviceToHost, stream);
		if(e == hipSuccess){ e = hipStreamSynchronize(stream); }
		if(e != hipSuccess){ break; }
		// strip the dword padding of the device rows
		const unsigned char *hb = (const unsigned char*)h_buf;
		if(dst_words*4 == out_slice){
			crc = crc32_parallel(crc, hb, nr*out_slice);
			ok = fwrite(hb, 1, nr*out_slice, fout) == nr*out_slice;
		}
		else{
			packed.resize(nr*out_slice);
			for(uint64_t r = 0; r < nr; ++r){ memcpy(packed.data() + r*out_slice, hb + r*dst_words*4, out_slice); }
			crc = crc32_parallel(crc, packed.data(), packed.size());
			ok = fwrite(packed.data(), 1, packed.size(), fout) == packed.size();
		}
	}
	if(d_dst){ (void)hipFree(d_dst); }
	if(d_src){ (void)hipFree(d_src); }
	if(h_buf){ (void)hipHostFree(h_buf); }
	if(e != hipSuccess){ fclose(fout); return fail(KWAGE_ERR_DEVICE, "kwage_repack_db: %s", hipGetErrorString(e)); }
	if(!ok){ fclose(fout); return fail(KWAGE_ERR_IO, "kwage_repack_db: %s", err.empty() ? "I/O error" : err.c_str()); }
	for(uint32_t i = 0; i < n; ++i){
		// the reference's merge refuses a source whose slices do not match the CRC32 in its header (merge_db.cpp:608-614)
		if(src_crc[i] != src[i].header.crc32){
			fclose(fout);
			(void)remove(out_path);
			return fail(KWAGE_ERR_FORMAT, "kwage_repack_db: Invalid CRC32 value for source database file %s (header %08x, slices %08x)",
			            in_paths[i], src[i].header.crc32, src_crc[i]);
		}
	}

	// metadata: records verbatim in column order, fresh index
	const uint64_t info_start = DB_HEADER_BYTES + nrows*out_slice;
	std::vector<uint64_t> loc;
	std::vector<unsigned char> recs;
	uint64_t pos = info_start + 8ull*total_cols;
	for(uint32_t i = 0; i < n; ++i){
		std::string meta_err;
		if(!info[i].load(meta_err)){ fclose(fout); return fail(KWAGE_ERR_IO, "kwage_repack_db: %s", meta_err.c_str()); }
		for(uint32_t j = 0; j < src[i].header.num_filter; ++j){
			const uint64_t l = info[i].info_loc[j];
			if(l < info[i].tail_start || l >= info[i].tail_start + info[i].tail.size()){ fclose(fout); return fail(KWAGE_ERR_FORMAT, "%s: bad metadata index", in_paths[i]); }
			const unsigned char *p = info[i].tail.data() + (l - info[i].tail_start);
			FilterInfo fi;
			size_t used = 0;
			if(!parse_filter_info(p, info[i].tail.size() - (l - info[i].tail_start), fi, &used)){ fclose(fout); return fail(KWAGE_ERR_FORMAT, "%s: bad FilterInfo record", in_paths[i]); }
			loc.push_back(pos);
			recs.insert(recs.end(), p, p + used);
			pos += used;
		}
	}
	auto put32 = [](unsigned char *p, uint32_t v) { for(int i = 0; i < 4; ++i){ p[i] = (unsigned char)(v >> (8*i)); } };
	auto put64 = [](unsigned char *p, uint64_t v) { for(int i = 0; i < 8; ++i){ p[i] = (unsigned char)(v >> (8*i)); } };
	std::vector<unsigned char> locb(8ull*total_cols);
	for(uint64_t i = 0; i < total_cols; ++i){ put64(locb.data() + 8*i, loc[i]); }
	ok = fwrite(locb.data(), 1, locb.size(), fout) == locb.size();
	ok = ok && fwrite(recs.data(), 1, recs.size(), fout) == recs.size();
	const kwage_db_header &h0 = src[0].header;
	put32(hdr, KWAGE_MAGIC_NUMBER); put32(hdr + 4, h0.version); put32(hdr + 8, crc);
	put32(hdr + 12, h0.kmer_len); put32(hdr + 16, h0.num_hash); put32(hdr + 20, h0.log_2_filter_len);
	put32(hdr + 24, (uint32_t)total_cols); put32(hdr + 28, (uint32_t)h0.hash_func); put32(hdr + 32, 0);
	put64(hdr + 36, info_start);
	ok = ok && fseek(fout, 0, SEEK_SET) == 0 && fwrite(hdr, 1, sizeof(hdr), fout) == sizeof(hdr);
	ok = (fclose(fout) == 0) && ok;
	if(!ok){ return fail(KWAGE_ERR_IO, "kwage_repack_db: error writing %s", out_path); }
	return KWAGE_OK;
}
