// kwage_amd/csrc/host.cpp -- host-side half of the C ABI (no device code): error text,
// `.db` header + metadata reader, accession codec, FASTA/FASTQ iterator.  Written from the
// format the reference produces; each function cites the reference lines whose BEHAVIOUR it
// must reproduce (paths relative to the reference tree).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include <atomic>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "host.hpp"
#include "internal.h"

namespace kwage {

static thread_local std::string g_error;

void set_error(const char *fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_error = buf;
}

int fail(int code, const char *fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_error = buf;
	return code;
}

static inline uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint64_t rd64(const unsigned char *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

// kwage.h:36-46 member order, binary_io.cpp:255-265 field-wise little-endian, no padding.
void unpack_db_header(const unsigned char *b, kwage_db_header *h)
{
	h->magic = rd32(b); h->version = rd32(b + 4); h->crc32 = rd32(b + 8); h->kmer_len = rd32(b + 12);
	h->num_hash = rd32(b + 16); h->log_2_filter_len = rd32(b + 20); h->num_filter = rd32(b + 24);
	h->hash_func = (int32_t)rd32(b + 28); h->compression = rd32(b + 32); h->info_start = rd64(b + 36);
}

int check_params(const kwage_params *p)
{
	if(p->kmer_len < 1 || p->kmer_len > KWAGE_MAX_WORD_LEN){       // word.h:10
		return fail(KWAGE_ERR_ARG, "kmer_len %u outside [1,%d]", p->kmer_len, KWAGE_MAX_WORD_LEN);
	}
	if(p->num_hash < KWAGE_MIN_NUM_HASH || p->num_hash > KWAGE_MAX_NUM_HASH){   // bloom.h:20-21
		return fail(KWAGE_ERR_ARG, "num_hash %u outside [%d,%d]", p->num_hash, KWAGE_MIN_NUM_HASH, KWAGE_MAX_NUM_HASH);
	}
	if(p->log_2_filter_len > 32){                                  // 32-bit hash, hash.h:20-21
		return fail(KWAGE_ERR_ARG, "log_2_filter_len %u > 32", p->log_2_filter_len);
	}
	if(p->hash_func != KWAGE_HASH_MURMUR32){                        // hash.cpp:92
		return fail(KWAGE_ERR_HASH, "bigsi_hash: Unknown hash function");
	}
	return KWAGE_OK;
}

// ---- sra_accession.cpp:27-96 ----------------------------------------------------------------
bool str_to_accession(const std::string &s, uint64_t &out)
{
	uint64_t num_letter = 0, num_digit = 0, data = 0;
	for(char ch : s){
		const int u = toupper((unsigned char)ch);
		if(u >= 'A' && u <= 'Z'){ ++num_letter; data = data*26 + (uint64_t)(u - 'A'); }
		else if(ch >= '0' && ch <= '9'){ ++num_digit; data = data*10 + (uint64_t)(ch - '0'); }
	}
	if(num_letter != 3 || num_digit == 0 || num_digit > 10){ return false; }
	out = (num_digit - 1) | (data << 4);
	return out != 0;
}

std::string accession_to_str(uint64_t acc)
{
	std::string ret;
	const uint64_t num_digit = (acc & 0xF) + 1;
	uint64_t data = (acc >> 4) & 0x0FFFFFFFFFFFFFFFull;
	for(uint64_t i = 0; i < num_digit; ++i){ ret.push_back((char)('0' + data % 10)); data /= 10; }
	for(int i = 0; i < 3; ++i){ ret.push_back((char)('A' + data % 26)); data /= 26; }
	std::reverse(ret.begin(), ret.end());
	return ret;
}

// ---- FilterInfo (bloom.h:474-537), serialized by binary_io.cpp:154-176 -----------------------
std::string FilterInfo::csv_string() const { return accession_to_str(run_accession); }   // bloom.cpp:124-127

// bloom.cpp:129-326: only non-empty fields, in this fixed order, joined by ",\n".
std::string FilterInfo::json_string(const std::string &prefix) const
{
	std::ostringstream out;
	bool wrote = false;
	auto sep = [&]() { if(wrote){ out << ",\n"; } wrote = true; };
	auto str_field = [&](const char *name, const std::string &v) {
		if(!v.empty()){ sep(); out << prefix << '"' << name << "\": \"" << v << '"'; }
	};
	auto acc_field = [&](const char *name, uint64_t a) {
		if(a != 0){ sep(); out << prefix << '"' << name << "\": \"" << accession_to_str(a) << '"'; }
	};

	acc_field("run", run_accession);
	if(year != 0 && month != 0 && day != 0){     // Date::is_valid, date.h; printed Y-M-D unpadded, date.cpp:5-10
		sep(); out << prefix << "\"date received\": \"" << year << '-' << month << '-' << day << '"';
	}
	acc_field("experiment", experiment_accession);
	str_field("experiment title", experiment_title);
	str_field("experiment design", experiment_design_description);
	str_field("experiment library name", experiment_library_name);
	str_field("experiment library strategy", experiment_library_strategy);
	str_field("experiment library source", experiment_library_source);
	str_field("experiment library selection", experiment_library_selection);
	str_field("experiment instrument model", experiment_instrument_model);
	acc_field("sample", sample_accession);
	str_field("sample taxa", sample_taxa);
	if(!sample_attributes.empty()){
		sep();
		out << prefix << "\"sample attributes\": [\n";
		bool first = true;
		for(const auto &kv : sample_attributes){
			if(!first){ out << ",\n"; }
			first = false;
			out << prefix << "\t{\n";
			out << prefix << "\t\t\"tag\": \"" << kv.first << "\",\n";
			out << prefix << "\t\t\"value\": \"" << kv.second << "\"\n";
			out << prefix << "\t}";
		}
		out << '\n' << prefix << ']';
	}
	acc_field("study", study_accession);
	str_field("study title", study_title);
	str_field("study abstract", study_abstract);
	return out.str();
}

namespace {

struct Cursor {
	const unsigned char *p, *end;
	bool ok = true;
	uint64_t u64() { if(end - p < 8){ ok = false; return 0; } uint64_t v = rd64(p); p += 8; return v; }
	uint32_t u32() { if(end - p < 4){ ok = false; return 0; } uint32_t v = rd32(p); p += 4; return v; }
	std::string cstr()    // binary_io.cpp:29-53: bytes up to the NUL, no length prefix
	{
		const unsigned char *z = (const unsigned char*)memchr(p, 0, (size_t)(end - p));
		if(!z){ ok = false; p = end; return std::string(); }
		std::string s((const char*)p, (size_t)(z - p));
		p = z + 1;
		return s;
	}
};

}  // namespace

bool parse_filter_info(const unsigned char *buf, size_t len, FilterInfo &fi, size_t *consumed)
{
	Cursor c{buf, buf + len};
	fi = FilterInfo();
	fi.run_accession = c.u64();
	fi.experiment_accession = c.u64();
	fi.experiment_title = c.cstr();
	fi.experiment_design_description = c.cstr();
	fi.experiment_library_name = c.cstr();
	fi.experiment_library_strategy = c.cstr();
	fi.experiment_library_source = c.cstr();
	fi.experiment_library_selection = c.cstr();
	fi.experiment_instrument_model = c.cstr();
	fi.sample_accession = c.u64();
	fi.sample_taxa = c.cstr();
	const uint64_t n = c.u64();                   // binary_io.h:179-205 count, then (key, value) pairs
	for(uint64_t i = 0; i < n && c.ok; ++i){
		std::pair<std::string, std::string> kv;
		kv.first = c.cstr();
		kv.second = c.cstr();
		fi.sample_attributes.insert(kv);          // same container + insertion order as the reference,
		                                          // so iteration order (JSON output) matches
	}
	fi.study_accession = c.u64();
	fi.study_title = c.cstr();
	fi.study_abstract = c.cstr();
	fi.number_of_spots = c.u64();
	fi.number_of_bases = c.u64();
	fi.day = c.u32(); fi.month = c.u32(); fi.year = c.u32();     // date.h:17-20 member order
	if(consumed){ *consumed = (size_t)(c.p - buf); }
	return c.ok;
}

void pack_filter_info(const FilterInfo &fi, std::vector<unsigned char> &out)
{
	auto u64 = [&](uint64_t v) { for(int i = 0; i < 8; ++i){ out.push_back((unsigned char)(v >> (8*i))); } };
	auto u32 = [&](uint32_t v) { for(int i = 0; i < 4; ++i){ out.push_back((unsigned char)(v >> (8*i))); } };
	auto str = [&](const std::string &s) { out.insert(out.end(), s.begin(), s.end()); out.push_back(0); };
	u64(fi.run_accession); u64(fi.experiment_accession);
	str(fi.experiment_title); str(fi.experiment_design_description); str(fi.experiment_library_name);
	str(fi.experiment_library_strategy); str(fi.experiment_library_source); str(fi.experiment_library_selection);
	str(fi.experiment_instrument_model);
	u64(fi.sample_accession);
	str(fi.sample_taxa);
	u64(fi.sample_attributes.size());
	for(const auto &kv : fi.sample_attributes){ str(kv.first); str(kv.second); }
	u64(fi.study_accession);
	str(fi.study_title); str(fi.study_abstract);
	u64(fi.number_of_spots); u64(fi.number_of_bases);
	u32(fi.day); u32(fi.month); u32(fi.year);
}

// ---- database metadata ----------------------------------------------------------------------
bool DbInfo::open(const std::string &file, std::string &err)
{
	std::ifstream fin(file.c_str(), std::ios::binary);
	if(!fin){ err = "Unable to open database file " + file + " for reading"; return false; }
	unsigned char hb[DB_HEADER_BYTES];
	fin.read((char*)hb, DB_HEADER_BYTES);
	if(!fin){ err = file + ": Unable to read header"; return false; }
	unpack_db_header(hb, &header);
	fin.seekg(0, std::ios::end);
	const uint64_t fsize = (uint64_t)fin.tellg();
	if(header.info_start > fsize || fsize - header.info_start < 8ull*header.num_filter){
		err = file + ": metadata index lies outside the file";
		return false;
	}
	path = file;
	tail_start = header.info_start;
	tail_bytes = fsize - tail_start;
	return true;
}

// info_loc[N] then the FilterInfo records (build_db.cpp:371-416): the tail of the file, read once
bool DbInfo::load(std::string &err) const
{
	std::lock_guard<std::mutex> lk(once);
	if(state == 0){
		state = -1;
		std::ifstream fin(path.c_str(), std::ios::binary);
		tail.resize(tail_bytes);
		if(fin){ fin.seekg((std::streamoff)tail_start); fin.read((char*)tail.data(), (std::streamsize)tail.size()); }
		if(fin && tail.size() >= 8ull*header.num_filter){
			info_loc.resize(header.num_filter);
			for(uint32_t j = 0; j < header.num_filter; ++j){ info_loc[j] = rd64(tail.data() + 8ull*j); }
			state = 1;
		}
		else{ tail.clear(); }
	}
	if(state != 1){ err = path + ": Unable to read metadata"; return false; }
	return true;
}

bool DbInfo::info(uint32_t column, FilterInfo &fi) const
{
	if(column >= header.num_filter){ return false; }
	std::string err;
	if(!load(err)){ return false; }
	const uint64_t loc = info_loc[column];          // kwage.cpp:505-515
	if(loc < tail_start || loc >= tail_start + tail.size()){ return false; }
	return parse_filter_info(tail.data() + (loc - tail_start), tail.size() - (loc - tail_start), fi);
}

// ---- slice block reader + the compressed container -----------------------------------------
// Codec = the reference's slice_z.h (dead code there, never wired to a container): raw deflate,
// windowBits -9 (slice_z.h:9), level 9, memLevel 9, Z_DEFAULT_STRATEGY (slice_z.h:169-196); a slice
// is stored raw when deflate does not make it smaller (CompressSlice::compress, slice_z.h:231-251).
// Container (this repo's definition; the reference has none):
//   [44-byte header, compression = 2, crc32 = crc of the UNCOMPRESSED slice block]
//   [u64 offset[2^L + 1]  absolute file offsets; slice i = bytes offset[i] .. offset[i+1])]
//   [slice payloads: length == ceil(N/8) -> raw bytes, otherwise a raw-deflate stream]
//   [u64 info_loc[N]] [FilterInfo records]                       (as in the uncompressed layout)
static const int SLICE_Z_WINDOW_BITS = -9;

static unsigned host_threads(unsigned want)
{
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	return std::max(1u, std::min(want ? want : 16u, hw));
}

static bool pread_all(int fd, void *dst, uint64_t n, uint64_t off)
{
	uint64_t got = 0;
	while(got < n){
		const ssize_t k = pread(fd, (char*)dst + got, n - got, (off_t)(off + got));
		if(k <= 0){ return false; }
		got += (uint64_t)k;
	}
	return true;
}

bool DbSliceSource::open(const std::string &path, std::string &err)
{
	close();
	fd = ::open(path.c_str(), O_RDONLY);
	if(fd < 0){ err = "Unable to open database file " + path + " for reading"; return false; }
	unsigned char hb[DB_HEADER_BYTES];
	struct stat st;
	if(fstat(fd, &st) != 0 || !pread_all(fd, hb, DB_HEADER_BYTES, 0)){ err = path + ": Unable to read header"; return false; }
	unpack_db_header(hb, &header);
	if(header.magic != KWAGE_MAGIC_NUMBER){ err = path + ": not a KWAGE database (bad magic)"; return false; }
	if(header.log_2_filter_len > 32){ err = path + ": log_2_filter_len > 32"; return false; }
	slice_size = ((uint64_t)header.num_filter + 7)/8;
	nrows = 1ull << header.log_2_filter_len;
	const uint64_t fsize = (uint64_t)st.st_size;
	if(header.compression == KWAGE_COMPRESSION_NONE){
		if(fsize < DB_HEADER_BYTES + nrows*slice_size){ err = path + ": file is shorter than header + slices"; return false; }
		return true;
	}
	if(header.compression != KWAGE_COMPRESSION_DEFLATE){
		err = path + ": unsupported compression code (only 0 = none and 2 = deflate container)";
		return false;
	}
	// (the size first: a damaged log_2_filter_len must not make this allocate -- and zero -- a table of up to 32 GiB)
	if(fsize < DB_HEADER_BYTES + 8*(nrows + 1)){ err = path + ": truncated slice offset table"; return false; }
	offsets.resize(nrows + 1);
	if(!pread_all(fd, offsets.data(), 8*(nrows + 1), DB_HEADER_BYTES)){
		err = path + ": truncated slice offset table";
		return false;
	}
	for(uint64_t i = 0; i < nrows; ++i){
		if(offsets[i + 1] < offsets[i] || offsets[i + 1] > fsize || offsets[i + 1] - offsets[i] > slice_size){
			err = path + ": corrupt slice offset table";
			return false;
		}
	}
	return true;
}

void DbSliceSource::close()
{
	if(fd >= 0){ ::close(fd); fd = -1; }
	offsets.clear();
}

bool DbSliceSource::read_rows(uint64_t r0, uint64_t nr, unsigned char *dst, std::string &err)
{
	if(header.compression == KWAGE_COMPRESSION_NONE){
		// one thread moves ~4 GB/s out of the page cache; several in parallel keep up with PCIe
		const uint64_t total = nr*slice_size, base = DB_HEADER_BYTES + r0*slice_size;
		static const unsigned load_threads = []() { const char *e = getenv("KWAGE_LOAD_THREADS"); return (e && atoi(e) > 0) ? (unsigned)atoi(e) : 8u; }();
		const unsigned nt = (total >= (8u << 20)) ? host_threads(load_threads) : 1;
		if(nt == 1){
			if(!pread_all(fd, dst, total, base)){ err = "Error reading slice from file"; return false; }
			return true;
		}
		const uint64_t part = (total/nt + 4095)/4096*4096;
		std::atomic<bool> bad(false);
		std::vector<std::thread> pool;
		for(unsigned t = 0; t < nt; ++t){
			const uint64_t b = (uint64_t)t*part;
			if(b >= total){ break; }
			const uint64_t len = std::min(part, total - b);
			pool.emplace_back([&, b, len]() { if(!pread_all(fd, dst + b, len, base + b)){ bad = true; } });
		}
		for(auto &t : pool){ t.join(); }
		if(bad){ err = "Error reading slice from file"; return false; }
		return true;
	}
	const uint64_t c0 = offsets[r0], c1 = offsets[r0 + nr];
	std::vector<unsigned char> comp(c1 - c0);
	if(!pread_all(fd, comp.data(), c1 - c0, c0)){ err = "Error reading compressed slice from file"; return false; }
	const unsigned nt = host_threads(16);
	std::atomic<uint64_t> next(0);
	std::atomic<bool> bad(false);
	auto work = [&]() {
		z_stream z;
		memset(&z, 0, sizeof(z));
		if(inflateInit2(&z, SLICE_Z_WINDOW_BITS) != Z_OK){ bad = true; return; }
		const uint64_t grain = 1024;
		for(uint64_t b = next.fetch_add(grain); b < nr && !bad; b = next.fetch_add(grain)){
			for(uint64_t i = b; i < std::min(nr, b + grain); ++i){
				const uint64_t o = offsets[r0 + i] - c0, len = offsets[r0 + i + 1] - offsets[r0 + i];
				unsigned char *out = dst + i*slice_size;
				if(len == slice_size){ memcpy(out, comp.data() + o, len); continue; }     // stored raw
				z.next_in = comp.data() + o; z.avail_in = (uInt)len;
				z.next_out = out; z.avail_out = (uInt)slice_size;
				if(inflate(&z, Z_FINISH) != Z_STREAM_END || z.avail_out != 0 || inflateReset(&z) != Z_OK){ bad = true; break; }
			}
		}
		inflateEnd(&z);
	};
	std::vector<std::thread> pool;
	for(unsigned t = 1; t < nt; ++t){ pool.emplace_back(work); }
	work();
	for(auto &t : pool){ t.join(); }
	if(bad){ err = "InflateSlice::inflate: Error in inflate"; return false; }
	return true;
}

bool DbSliceSource::read_row_list(const uint32_t *rows, uint64_t n, unsigned char *dst, std::string &err, unsigned max_threads)
{
	for(uint64_t i = 0; i < n; ++i){ if(rows[i] >= nrows){ err = "slice index out of range"; return false; } }
	const unsigned nt = (n >= 4096 && max_threads > 1) ? host_threads(max_threads) : 1;
	std::atomic<uint64_t> next(0);
	std::atomic<bool> bad(false);
	auto work = [&]() {
		z_stream z;
		const bool packed = (header.compression != KWAGE_COMPRESSION_NONE);
		if(packed){
			memset(&z, 0, sizeof(z));
			if(inflateInit2(&z, SLICE_Z_WINDOW_BITS) != Z_OK){ bad = true; return; }
		}
		std::vector<unsigned char> comp(packed ? slice_size : 0);
		const uint64_t grain = 256;
		for(uint64_t b = next.fetch_add(grain); b < n && !bad; b = next.fetch_add(grain)){
			for(uint64_t i = b; i < std::min(n, b + grain); ++i){
				unsigned char *out = dst + i*slice_size;
				if(!packed){
					if(!pread_all(fd, out, slice_size, DB_HEADER_BYTES + (uint64_t)rows[i]*slice_size)){ bad = true; break; }
					continue;
				}
				const uint64_t o = offsets[rows[i]], len = offsets[rows[i] + 1] - o;
				if(len == slice_size){ if(!pread_all(fd, out, len, o)){ bad = true; break; } continue; }      // stored raw
				if(!pread_all(fd, comp.data(), len, o)){ bad = true; break; }
				z.next_in = comp.data(); z.avail_in = (uInt)len;
				z.next_out = out; z.avail_out = (uInt)slice_size;
				if(inflate(&z, Z_FINISH) != Z_STREAM_END || z.avail_out != 0 || inflateReset(&z) != Z_OK){ bad = true; break; }
			}
		}
		if(packed){ inflateEnd(&z); }
	};
	std::vector<std::thread> pool;
	for(unsigned t = 1; t < nt; ++t){ pool.emplace_back(work); }
	work();
	for(auto &t : pool){ t.join(); }
	if(bad){ err = "Error reading slice from file"; return false; }
	return true;
}

bool DbSliceSource::slice_crc32(uint32_t &crc, std::string &err)
{
	const uint64_t chunk_rows = std::max<uint64_t>(1, std::min<uint64_t>(nrows, (64ull << 20)/std::max<uint64_t>(slice_size, 1)));
	std::vector<unsigned char> buf(chunk_rows*slice_size);
	crc = 0;
	for(uint64_t r0 = 0; r0 < nrows; r0 += chunk_rows){
		const uint64_t nr = std::min(chunk_rows, nrows - r0);
		if(!read_rows(r0, nr, buf.data(), err)){ return false; }
		crc = (uint32_t)crc32_z(crc, buf.data(), nr*slice_size);
	}
	return true;
}

// ---- file_util.cpp:95-121 -------------------------------------------------------------------
bool find_file_extension(const std::string &path, const char *ext)
{
	// case-insensitive; the FIRST occurrence of ext must end the string (ifind.cpp + file_util.cpp:108-121)
	const size_t n = strlen(ext);
	for(size_t s = 0; s + n <= path.size(); ++s){
		size_t i = 0;
		while(i < n && tolower((unsigned char)path[s + i]) == tolower((unsigned char)ext[i])){ ++i; }
		if(i == n){ return s + n == path.size(); }
	}
	return false;
}

// ---- SequenceIterator (parse_sequence.cpp) --------------------------------------------------
static int seq_file_type(const std::string &fn)     // parse_sequence.cpp:13-26
{
	static const char *fa[] = {".fna", ".fna.gz", ".fa", ".fa.gz", ".fasta", ".fasta.gz"};
	for(const char *e : fa){ if(find_file_extension(fn, e)){ return 0; } }
	if(find_file_extension(fn, ".fastq") || find_file_extension(fn, ".fastq.gz")){ return 1; }
	return 2;
}

SeqFile::SeqFile() : fin(nullptr), type(2) {}
SeqFile::~SeqFile() { close(); }

void SeqFile::close()
{
	if(fin){ gzclose((gzFile)fin); fin = nullptr; }
}

// gzgets(fin, out, len) -- at most len-1 characters, stopping after a newline; NULL when nothing is left -- over
// gzread in 256 KiB blocks (plain and gzip files alike): the chunking the reference's 2048-byte buffer makes observable
// stays exactly the same, the byte-at-a-time copy of zlib's own gzgets does not (1.6 -> 2.6 M reads/s for 150-base FASTQ
// records on one core).
char *SeqFile::get_line(char *out, int len)
{
	return get_chunk(out, len) ? out : nullptr;
}

// The same, returning the number of characters stored (0: nothing left).
int SeqFile::get_chunk(char *out, int len)
{
	if(len < 1){ return 0; }
	int n = 0;
	while(n < len - 1){
		if(rpos == rend){
			if(rbuf.empty()){ rbuf.resize(256u << 10); }
			const int got = gzread((gzFile)fin, rbuf.data(), (unsigned)rbuf.size());
			rpos = 0;
			rend = got > 0 ? (size_t)got : 0;
			if(rend == 0){ break; }
		}
		const size_t room = std::min<size_t>((size_t)(len - 1 - n), rend - rpos);
		const char *nl = (const char*)memchr(rbuf.data() + rpos, '\n', room);
		const size_t take = nl ? (size_t)(nl - (rbuf.data() + rpos)) + 1 : room;
		memcpy(out + n, rbuf.data() + rpos, take);
		n += (int)take;
		rpos += take;
		if(nl){ break; }
	}
	out[n] = 0;
	return n;
}

bool SeqFile::open(const std::string &path, std::string &err)
{
	close();
	rpos = rend = 0;
	type = seq_file_type(path);
	if(type == 2){ err = "SequenceIterator: Unknown file type"; return false; }
	fin = gzopen(path.c_str(), "r");      // plain and gzip files alike, parse_sequence.cpp:40
	if(!fin){ err = "Error opening: " + path; return false; }
	(void)gzbuffer((gzFile)fin, 1u << 20);      // zlib's own input buffer is 8 KiB by default: one read() per 8 KiB of a .gz file
	return true;
}

// One chunk as the reference's C-string code sees it: `len` = strlen (a chunk with an embedded NUL ends there), `eol` =
// it holds a '\n' or '\r' (parse_sequence.cpp tests strpbrk(buffer, "\n\r")).  A '\n' can only be the chunk's last
// character, so the common case is decided without a scan.
struct Chunk {
	size_t len;
	bool eol;
	Chunk(const char *b, int stored)
	{
		len = strlen(b);
		eol = (len == (size_t)stored && len && b[len - 1] == '\n') || memchr(b, '\r', len) != nullptr;
	}
};

// Sequence characters as the reference stores them (parse_sequence.cpp:139-146, 203-214): white space dropped, the rest
// upper-cased -- isspace / toupper of the "C" locale (the reference never calls setlocale), as one table look-up per
// character instead of two library calls.
struct SeqCharTable {
	unsigned char t[256];
	SeqCharTable()
	{
		for(int c = 0; c < 256; ++c){
			const bool space = (c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r');
			t[c] = space ? 0 : (unsigned char)((c >= 'a' && c <= 'z') ? c - 'a' + 'A' : c);
		}
	}
};
static const SeqCharTable g_seq_chars;

static inline void append_sequence_chars(std::string &seq, const char *buffer, size_t len)
{
	const size_t before = seq.size();
	seq.resize(before + len);
	char *w = &seq[before];
	for(size_t i = 0; i < len; ++i){
		const unsigned char c = g_seq_chars.t[(unsigned char)buffer[i]];
		*w = (char)c;
		w += (c != 0);
	}
	seq.resize((size_t)(w - seq.data()));
}

// A defline chunk without its end-of-line characters.
static inline void append_defline_chars(std::string &info, const char *buffer, size_t len)
{
	size_t body = len;
	while(body && (buffer[body - 1] == '\n' || buffer[body - 1] == '\r')){ --body; }
	if(memchr(buffer, '\r', body) == nullptr){ info.append(buffer, body); }       // a '\n' cannot be inside
	else{ for(size_t i = 0; i < body; ++i){ if(buffer[i] != '\r'){ info.push_back(buffer[i]); } } }
}

// Returns 1 with (curr_defline, seq) set, 0 at end of file, -1 on a malformed FASTQ record.
int SeqFile::next(std::string &err)
{
	if(!fin){ return 0; }
	const int buffer_len = 2048;            // gzgets chunking is observable in over-long deflines
	char buffer[buffer_len];
	seq.clear();
	int n;

	if(type == 0){      // parse_sequence.cpp:72-151
		std::string &info = scratch;
		while((n = get_chunk(buffer, buffer_len)) != 0){
			Chunk c(buffer, n);
			if(memchr(buffer, '>', c.len) != nullptr){         // ANY line containing '>' is a defline (:86)
				info.clear();
				append_defline_chars(info, buffer, c.len);
				if(!c.eol){
					// :100-108 -- continuation chunks are appended until one holds the end of line;
					// that last chunk is consumed but NOT appended
					while((n = get_chunk(buffer, buffer_len)) != 0){
						Chunk more(buffer, n);
						if(more.eol){ break; }
						append_defline_chars(info, buffer, more.len);
					}
				}
				size_t s = 0;
				while(s < info.size() && (isspace((unsigned char)info[s]) || info[s] == '>')){ ++s; }
				if(!seq.empty()){
					std::swap(curr_defline, next_defline);
					next_defline.assign(info, s, std::string::npos);
					return 1;
				}
				next_defline.assign(info, s, std::string::npos);
			}
			else{
				append_sequence_chars(seq, buffer, c.len);
			}
		}
		if(!seq.empty()){
			std::swap(curr_defline, next_defline);
			return 1;
		}
		close();
		return 0;
	}

	// FASTQ, parse_sequence.cpp:153-262
	std::string &info = scratch;
	info.clear();
	while(true){
		if((n = get_chunk(buffer, buffer_len)) == 0){ close(); return 0; }
		Chunk c(buffer, n);
		append_defline_chars(info, buffer, c.len);
		if(c.eol){ break; }
	}
	size_t s = 0;
	while(s < info.size() && (isspace((unsigned char)info[s]) || info[s] == '@')){ ++s; }
	curr_defline.assign(info, s, std::string::npos);
	while(true){
		if((n = get_chunk(buffer, buffer_len)) == 0){ err = "next_fastq: Unable to read sequence"; return -1; }
		Chunk c(buffer, n);
		append_sequence_chars(seq, buffer, c.len);
		if(c.eol){ break; }
	}
	if((n = get_chunk(buffer, buffer_len)) == 0){ err = "next_fastq: Unable to read '+'"; return -1; }
	if(!Chunk(buffer, n).eol){ err = "next_fastq: Error reading '+' delimiter"; return -1; }
	while(true){
		if((n = get_chunk(buffer, buffer_len)) == 0){ err = "next_fastq: Unable to read quality"; return -1; }
		if(Chunk(buffer, n).eol){ break; }
	}
	if(!seq.empty()){ return 1; }
	close();           // :253-261: an empty sequence ends the iteration
	return 0;
}

}  // namespace kwage

// ---------------------------------------------------------------------------------------------
// C ABI wrappers
// ---------------------------------------------------------------------------------------------
using namespace kwage;

extern "C" const char *kwage_last_error(void) { return g_error.c_str(); }
extern "C" uint32_t kwage_abi_version(void) { return KWAGE_AMD_ABI_VERSION; }

extern "C" int kwage_db_read_header(const char *path, kwage_db_header *out)
{
	if(!path || !out){ return fail(KWAGE_ERR_ARG, "kwage_db_read_header: NULL argument"); }
	FILE *f = fopen(path, "rb");
	if(!f){ return fail(KWAGE_ERR_IO, "Unable to open database file %s for reading", path); }
	unsigned char hb[DB_HEADER_BYTES];
	const size_t got = fread(hb, 1, DB_HEADER_BYTES, f);
	fclose(f);
	if(got != DB_HEADER_BYTES){ return fail(KWAGE_ERR_IO, "%s: Unable to read header", path); }
	unpack_db_header(hb, out);
	// The reference validates nothing here (kwage.cpp:99-105); a wrong magic would make it read
	// garbage.  Refusing such a file is the only deliberate deviation.
	if(out->magic != KWAGE_MAGIC_NUMBER){ return fail(KWAGE_ERR_FORMAT, "%s: not a KWAGE database (bad magic 0x%08x)", path, out->magic); }
	return KWAGE_OK;
}

static int copy_tail(int in_fd, uint64_t in_off, uint64_t in_size, FILE *out)
{
	std::vector<unsigned char> buf(1 << 20);
	while(in_off < in_size){
		const uint64_t n = std::min<uint64_t>(buf.size(), in_size - in_off);
		if(!pread_all(in_fd, buf.data(), n, in_off) || fwrite(buf.data(), 1, n, out) != n){ return -1; }
		in_off += n;
	}
	return 0;
}

static void pack_db_header(const kwage_db_header &h, unsigned char *b)
{
	auto p32 = [](unsigned char *p, uint32_t v) { for(int i = 0; i < 4; ++i){ p[i] = (unsigned char)(v >> (8*i)); } };
	p32(b, h.magic); p32(b + 4, h.version); p32(b + 8, h.crc32); p32(b + 12, h.kmer_len); p32(b + 16, h.num_hash);
	p32(b + 20, h.log_2_filter_len); p32(b + 24, h.num_filter); p32(b + 28, (uint32_t)h.hash_func); p32(b + 32, h.compression);
	p32(b + 36, (uint32_t)h.info_start); p32(b + 40, (uint32_t)(h.info_start >> 32));
}

extern "C" int kwage_db_read_slices(const char *path, const uint32_t *rows, uint64_t n, unsigned char *out)
{
	if(!path || (n && (!rows || !out))){ return fail(KWAGE_ERR_ARG, "kwage_db_read_slices: NULL argument"); }
	DbSliceSource src;
	std::string err;
	if(!src.open(path, err)){ return fail(KWAGE_ERR_IO, "%s", err.c_str()); }
	if(n && !src.read_row_list(rows, n, out, err)){ return fail(KWAGE_ERR_IO, "%s: %s", path, err.c_str()); }
	return KWAGE_OK;
}

extern "C" int kwage_db_compress(const char *in_path, const char *out_path, uint32_t threads)
{
	if(!in_path || !out_path){ return fail(KWAGE_ERR_ARG, "kwage_db_compress: NULL argument"); }
	DbSliceSource src;
	std::string err;
	if(!src.open(in_path, err)){ return fail(KWAGE_ERR_IO, "%s", err.c_str()); }
	if(src.header.compression != KWAGE_COMPRESSION_NONE){ return fail(KWAGE_ERR_FORMAT, "%s is already compressed", in_path); }
	struct stat st;
	fstat(src.fd, &st);
	FILE *out = fopen(out_path, "wb");
	if(!out){ return fail(KWAGE_ERR_IO, "Unable to open %s for writing", out_path); }

	const uint64_t nrows = src.nrows, ss = src.slice_size;
	std::vector<uint64_t> offsets(nrows + 1);
	unsigned char hb[DB_HEADER_BYTES];
	memset(hb, 0, sizeof(hb));
	bool ok = fwrite(hb, 1, sizeof(hb), out) == sizeof(hb);
	ok = ok && fwrite(offsets.data(), 8, nrows + 1, out) == nrows + 1;          // placeholders
	uint64_t pos = DB_HEADER_BYTES + 8*(nrows + 1);
	uint32_t crc = 0;

	const uint64_t chunk_rows = std::max<uint64_t>(1, std::min<uint64_t>(nrows, (64ull << 20)/std::max<uint64_t>(ss, 1)));
	std::vector<unsigned char> raw(chunk_rows*ss), comp(chunk_rows*ss);
	std::vector<uint32_t> clen(chunk_rows);
	const unsigned nt = host_threads(threads);
	for(uint64_t r0 = 0; r0 < nrows && ok; r0 += chunk_rows){
		const uint64_t nr = std::min(chunk_rows, nrows - r0);
		if(!src.read_rows(r0, nr, raw.data(), err)){ ok = false; break; }
		crc = (uint32_t)crc32_z(crc, raw.data(), nr*ss);
		std::atomic<uint64_t> next(0);
		std::atomic<bool> bad(false);
		auto work = [&]() {
			z_stream z;
			memset(&z, 0, sizeof(z));
			// slice_z.h:169-196: level 9, Z_DEFLATED, windowBits -9, memLevel 9, Z_DEFAULT_STRATEGY
			if(deflateInit2(&z, 9, Z_DEFLATED, SLICE_Z_WINDOW_BITS, 9, Z_DEFAULT_STRATEGY) != Z_OK){ bad = true; return; }
			const uint64_t grain = 512;
			for(uint64_t b = next.fetch_add(grain); b < nr; b = next.fetch_add(grain)){
				for(uint64_t i = b; i < std::min(nr, b + grain); ++i){
					z.next_in = raw.data() + i*ss; z.avail_in = (uInt)ss;
					z.next_out = comp.data() + i*ss; z.avail_out = (uInt)ss;      // "smaller than raw" or nothing
					const int zr = deflate(&z, Z_FINISH);
					const uint64_t produced = ss - z.avail_out;
					deflateReset(&z);
					if(zr == Z_STREAM_END && produced < ss){ clen[i] = (uint32_t)produced; }      // slice_z.h:250
					else{ clen[i] = (uint32_t)ss; memcpy(comp.data() + i*ss, raw.data() + i*ss, ss); }
				}
			}
			deflateEnd(&z);
		};
		std::vector<std::thread> pool;
		for(unsigned t = 1; t < nt; ++t){ pool.emplace_back(work); }
		work();
		for(auto &t : pool){ t.join(); }
		if(bad){ ok = false; break; }
		for(uint64_t i = 0; i < nr && ok; ++i){
			offsets[r0 + i] = pos;
			ok = fwrite(comp.data() + i*ss, 1, clen[i], out) == clen[i];
			pos += clen[i];
		}
	}
	offsets[nrows] = pos;
	// metadata: info_loc[] shifted by the size change, records verbatim
	kwage_db_header h = src.header;
	const uint64_t old_info = h.info_start;
	const int64_t shift = (int64_t)pos - (int64_t)old_info;
	std::vector<uint64_t> loc(h.num_filter);
	ok = ok && pread_all(src.fd, loc.data(), 8ull*h.num_filter, old_info);
	for(auto &l : loc){ l = (uint64_t)((int64_t)l + shift); }
	ok = ok && fwrite(loc.data(), 8, h.num_filter, out) == h.num_filter;
	ok = ok && copy_tail(src.fd, old_info + 8ull*h.num_filter, (uint64_t)st.st_size, out) == 0;
	h.compression = KWAGE_COMPRESSION_DEFLATE;
	h.info_start = pos;
	h.crc32 = crc;
	pack_db_header(h, hb);
	ok = ok && fseek(out, 0, SEEK_SET) == 0 && fwrite(hb, 1, sizeof(hb), out) == sizeof(hb);
	ok = ok && fwrite(offsets.data(), 8, nrows + 1, out) == nrows + 1;
	ok = (fclose(out) == 0) && ok;
	if(!ok){ return fail(KWAGE_ERR_IO, "kwage_db_compress: %s", err.empty() ? "I/O error" : err.c_str()); }
	return KWAGE_OK;
}

extern "C" int kwage_db_decompress(const char *in_path, const char *out_path)
{
	if(!in_path || !out_path){ return fail(KWAGE_ERR_ARG, "kwage_db_decompress: NULL argument"); }
	DbSliceSource src;
	std::string err;
	if(!src.open(in_path, err)){ return fail(KWAGE_ERR_IO, "%s", err.c_str()); }
	struct stat st;
	fstat(src.fd, &st);
	FILE *out = fopen(out_path, "wb");
	if(!out){ return fail(KWAGE_ERR_IO, "Unable to open %s for writing", out_path); }
	const uint64_t nrows = src.nrows, ss = src.slice_size;
	kwage_db_header h = src.header;
	const uint64_t old_info = h.info_start;
	h.compression = KWAGE_COMPRESSION_NONE;
	h.info_start = DB_HEADER_BYTES + nrows*ss;
	unsigned char hb[DB_HEADER_BYTES];
	pack_db_header(h, hb);       // crc32 is already the crc of the uncompressed slice block
	bool ok = fwrite(hb, 1, sizeof(hb), out) == sizeof(hb);
	const uint64_t chunk_rows = std::max<uint64_t>(1, std::min<uint64_t>(nrows, (64ull << 20)/std::max<uint64_t>(ss, 1)));
	std::vector<unsigned char> raw(chunk_rows*ss);
	uint32_t crc = 0;
	for(uint64_t r0 = 0; r0 < nrows && ok; r0 += chunk_rows){
		const uint64_t nr = std::min(chunk_rows, nrows - r0);
		if(!src.read_rows(r0, nr, raw.data(), err)){ ok = false; break; }
		crc = (uint32_t)crc32_z(crc, raw.data(), nr*ss);
		ok = fwrite(raw.data(), 1, nr*ss, out) == nr*ss;
	}
	if(ok && crc != h.crc32){ ok = false; err = "CRC32 of the inflated slice block does not match the header"; }
	const int64_t shift = (int64_t)h.info_start - (int64_t)old_info;
	std::vector<uint64_t> loc(h.num_filter);
	ok = ok && pread_all(src.fd, loc.data(), 8ull*h.num_filter, old_info);
	for(auto &l : loc){ l = (uint64_t)((int64_t)l + shift); }
	ok = ok && fwrite(loc.data(), 8, h.num_filter, out) == h.num_filter;
	ok = ok && copy_tail(src.fd, old_info + 8ull*h.num_filter, (uint64_t)st.st_size, out) == 0;
	ok = (fclose(out) == 0) && ok;
	if(!ok){ return fail(KWAGE_ERR_IO, "kwage_db_decompress: %s", err.empty() ? "I/O error" : err.c_str()); }
	return KWAGE_OK;
}

// ---- Bloom construction (host side) ---------------------------------------------------------------
extern "C" int kwage_optimal_bloom_param(uint32_t kmer_len, uint64_t num_kmer, float p_bound, uint32_t min_lg,
                                         uint32_t max_lg, kwage_params *out)
{
	if(!out){ return fail(KWAGE_ERR_ARG, "kwage_optimal_bloom_param: NULL argument"); }
	if(num_kmer == 0){ return fail(KWAGE_ERR_ARG, "optimal_bloom_param: No kmers found"); }      // bloom.cpp:16-18
	out->kmer_len = kmer_len;
	out->hash_func = KWAGE_HASH_MURMUR32;
	out->num_hash = 0;
	bool valid = false;
	for(uint32_t lg = min_lg; lg <= max_lg && lg < 64; ++lg){          // bloom.cpp:36-64
		float best_p = 10.0f;
		for(uint32_t nh = KWAGE_MIN_NUM_HASH; nh <= KWAGE_MAX_NUM_HASH; ++nh){
			const uint64_t len = 1ull << lg;
			// per-filter, per-k-mer false positive probability; num_kmer*num_hash is integer arithmetic
			const double p = pow(1.0 - pow(1.0 - 1.0/len, (double)(num_kmer*nh)), (double)nh);
			if((p <= p_bound) && (p < best_p)){
				best_p = (float)p;
				out->num_hash = nh;
				valid = true;
			}
		}
		if(valid){ out->log_2_filter_len = lg; return KWAGE_OK; }
	}
	return fail(KWAGE_ERR_ARG, "optimal_bloom_param: Unable to satisfy Bloom filter probability bound");
}

static bool opt_accession(const char *s, uint64_t &out, bool required)
{
	out = 0;
	if(!s || !*s){ return !required; }
	return str_to_accession(s, out);
}

namespace kwage {

int sample_info_to_filter_info(const kwage_sample_info *si, FilterInfo &fi)
{
	if(!opt_accession(si->run_accession, fi.run_accession, true) || !opt_accession(si->experiment_accession, fi.experiment_accession, false) ||
	   !opt_accession(si->sample_accession, fi.sample_accession, false) || !opt_accession(si->study_accession, fi.study_accession, false)){
		return fail(KWAGE_ERR_ARG, "str_to_accession: Unable to parse accession string");
	}
	auto txt = [](const char *s) { return std::string(s ? s : ""); };
	fi.experiment_title = txt(si->experiment_title);
	fi.experiment_design_description = txt(si->experiment_design_description);
	fi.experiment_library_name = txt(si->experiment_library_name);
	fi.experiment_library_strategy = txt(si->experiment_library_strategy);
	fi.experiment_library_source = txt(si->experiment_library_source);
	fi.experiment_library_selection = txt(si->experiment_library_selection);
	fi.experiment_instrument_model = txt(si->experiment_instrument_model);
	fi.sample_taxa = txt(si->sample_taxa);
	fi.study_title = txt(si->study_title);
	fi.study_abstract = txt(si->study_abstract);
	for(uint32_t i = 0; i < si->num_attributes; ++i){
		fi.sample_attributes.insert(std::make_pair(txt(si->attribute_tags[i]), txt(si->attribute_values[i])));
	}
	fi.number_of_spots = si->number_of_spots;
	fi.number_of_bases = si->number_of_bases;
	fi.day = si->day; fi.month = si->month; fi.year = si->year;
	return KWAGE_OK;
}

// binary_write<BloomFilter> (binary_io.cpp:182-208)
int write_bloom_file(const char *out_path, const kwage_params *params, const FilterInfo &fi,
                     const unsigned char *bits, uint64_t nbytes)
{
	std::vector<unsigned char> head;
	head.push_back(0xFF);                                               // BLOOM_MAGIC_COMPLETE, bloom.h:28
	auto u32 = [&](uint32_t v) { for(int i = 0; i < 4; ++i){ head.push_back((unsigned char)(v >> (8*i))); } };
	u32(params->kmer_len); u32(params->log_2_filter_len); u32(params->num_hash); u32((uint32_t)params->hash_func);   // bloom.h:550-554
	u32((uint32_t)crc32_z(crc32_z(0L, Z_NULL, 0), bits, nbytes));                                                      // bloom.cpp:328-343
	pack_filter_info(fi, head);
	FILE *f = fopen(out_path, "wb");
	if(!f){ return fail(KWAGE_ERR_IO, "Unable to open %s for writing", out_path); }
	bool ok = fwrite(head.data(), 1, head.size(), f) == head.size() && fwrite(bits, 1, nbytes, f) == nbytes;
	ok = (fclose(f) == 0) && ok;
	if(!ok){ return fail(KWAGE_ERR_IO, "binary_write<BloomFilter>: Unable to write BloomFilter"); }
	return KWAGE_OK;
}

}  // namespace kwage

extern "C" int kwage_make_bloom(kwage_ctx *ctx, const kwage_params *params, const char *seqs, const uint64_t *offsets,
                                uint32_t n_seqs, const kwage_sample_info *si, const char *out_path, uint64_t *num_distinct)
{
	if(!ctx || !params || !offsets || !si || !out_path){ return fail(KWAGE_ERR_ARG, "kwage_make_bloom: NULL argument"); }
	int rc = check_params(params);
	if(rc){ return rc; }
	FilterInfo fi;
	if((rc = sample_info_to_filter_info(si, fi))){ return rc; }

	// cut long sequences into pieces overlapping by k-1 bases: one workgroup walks one piece
	const uint64_t PIECE = 1u << 16;
	const uint32_t k = params->kmer_len;
	std::vector<uint64_t> po(1, 0);
	std::string pieces;
	for(uint32_t i = 0; i < n_seqs; ++i){
		const uint64_t b = offsets[i], e = offsets[i + 1];
		if(e < b){ return fail(KWAGE_ERR_ARG, "kwage_make_bloom: offsets must be non-decreasing"); }
		for(uint64_t s0 = b; s0 < e; s0 += PIECE){
			const uint64_t s1 = std::min(e, s0 + PIECE + (k - 1));
			pieces.append(seqs + s0, s1 - s0);
			po.push_back(pieces.size());
			if(s1 == e){ break; }
		}
	}
	if(po.size() - 1 > 0xFFFFFFFFull){ return fail(KWAGE_ERR_ARG, "kwage_make_bloom: too many sequence pieces"); }
	kwage_batch *batch = nullptr;
	if((rc = kwage_batch_create(ctx, pieces.data(), po.data(), (uint32_t)(po.size() - 1), &batch))){ return rc; }
	const uint64_t nbytes = ((1ull << params->log_2_filter_len) + 7)/8;
	std::vector<unsigned char> bits(nbytes);
	uint64_t distinct = 0;
	rc = kwage_bloom_bits_from_batch(ctx, params, batch, bits.data(), &distinct);
	kwage_batch_destroy(batch);
	if(rc){ return rc; }
	if(num_distinct){ *num_distinct = distinct; }

	return write_bloom_file(out_path, params, fi, bits.data(), bits.size());
}

struct kwage_dbinfo { DbInfo d; };

extern "C" int kwage_dbinfo_open(const char *path, kwage_dbinfo **out)
{
	if(!path || !out){ return fail(KWAGE_ERR_ARG, "kwage_dbinfo_open: NULL argument"); }
	*out = nullptr;
	kwage_dbinfo *d = new (std::nothrow) kwage_dbinfo();
	if(!d){ return fail(KWAGE_ERR_IO, "out of memory"); }
	std::string err;
	if(!d->d.open(path, err)){ delete d; return fail(KWAGE_ERR_IO, "%s", err.c_str()); }
	*out = d;
	return KWAGE_OK;
}

extern "C" void kwage_dbinfo_close(kwage_dbinfo *d) { delete d; }
extern "C" uint32_t kwage_dbinfo_num_filter(const kwage_dbinfo *d) { return d ? d->d.header.num_filter : 0; }

extern "C" int kwage_dbinfo_csv_string(const kwage_dbinfo *d, uint32_t column, char *buf, size_t buflen)
{
	if(!d || !buf){ return fail(KWAGE_ERR_ARG, "kwage_dbinfo_csv_string: NULL argument"); }
	FilterInfo fi;
	if(!d->d.info(column, fi)){ return fail(KWAGE_ERR_FORMAT, "Unable to read FilterInfo of column %u", column); }
	const std::string s = fi.csv_string();
	if(s.size() + 1 > buflen){ return fail(KWAGE_ERR_ARG, "buffer too small"); }
	memcpy(buf, s.c_str(), s.size() + 1);
	return KWAGE_OK;
}

extern "C" int64_t kwage_dbinfo_json_string(const kwage_dbinfo *d, uint32_t column, const char *prefix, char *buf, size_t buflen)
{
	if(!d){ return fail(KWAGE_ERR_ARG, "kwage_dbinfo_json_string: NULL argument"); }
	FilterInfo fi;
	if(!d->d.info(column, fi)){ return fail(KWAGE_ERR_FORMAT, "Unable to read FilterInfo of column %u", column); }
	const std::string s = fi.json_string(prefix ? prefix : "");
	if(buf && buflen){
		const size_t n = std::min(buflen - 1, s.size());
		memcpy(buf, s.data(), n);
		buf[n] = 0;
	}
	return (int64_t)s.size();
}

extern "C" int kwage_str_to_accession(const char *s, uint64_t *out)
{
	if(!s || !out){ return fail(KWAGE_ERR_ARG, "kwage_str_to_accession: NULL argument"); }
	if(!str_to_accession(s, *out)){ return fail(KWAGE_ERR_ARG, "str_to_accession: Unable to parse accession string"); }
	return KWAGE_OK;
}

extern "C" int kwage_accession_to_str(uint64_t acc, char *buf, size_t buflen)
{
	if(!buf){ return fail(KWAGE_ERR_ARG, "kwage_accession_to_str: NULL argument"); }
	const std::string s = accession_to_str(acc);
	if(s.size() + 1 > buflen){ return fail(KWAGE_ERR_ARG, "buffer too small"); }
	memcpy(buf, s.c_str(), s.size() + 1);
	return KWAGE_OK;
}

struct kwage_seqfile { SeqFile f; };

extern "C" int kwage_seqfile_open(const char *path, kwage_seqfile **out)
{
	if(!path || !out){ return fail(KWAGE_ERR_ARG, "kwage_seqfile_open: NULL argument"); }
	*out = nullptr;
	kwage_seqfile *f = new (std::nothrow) kwage_seqfile();
	if(!f){ return fail(KWAGE_ERR_IO, "out of memory"); }
	std::string err;
	if(!f->f.open(path, err)){ delete f; return fail(KWAGE_ERR_IO, "%s", err.c_str()); }
	*out = f;
	return KWAGE_OK;
}

extern "C" int kwage_seqfile_next(kwage_seqfile *f, const char **defline, const char **seq, uint64_t *seq_len)
{
	if(!f){ return fail(KWAGE_ERR_ARG, "kwage_seqfile_next: NULL argument"); }
	std::string err;
	const int r = f->f.next(err);
	if(r < 0){ return fail(KWAGE_ERR_FORMAT, "%s", err.c_str()); }
	if(r == 1){
		if(defline){ *defline = f->f.curr_defline.c_str(); }
		if(seq){ *seq = f->f.seq.c_str(); }
		if(seq_len){ *seq_len = f->f.seq.size(); }
	}
	return r;
}

extern "C" void kwage_seqfile_close(kwage_seqfile *f) { delete f; }

extern "C" uint32_t kwage_query_threshold(float threshold, uint32_t num_query_kmer)
{
	volatile float prod = threshold*(float)num_query_kmer;      // kwage.cpp:388
	return (uint32_t)prod;
}
