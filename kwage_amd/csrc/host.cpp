// kwage_amd/csrc/host.cpp -- host-side half of the C ABI (no device code): error text,
// `.db` header + metadata reader, accession codec, FASTA/FASTQ iterator.  Written from the
// format the reference produces; each function cites the reference lines whose BEHAVIOUR it
// must reproduce (paths relative to the reference tree).
#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

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
bool DbInfo::open(const std::string &path, std::string &err)
{
	std::ifstream fin(path.c_str(), std::ios::binary);
	if(!fin){ err = "Unable to open database file " + path + " for reading"; return false; }
	unsigned char hb[DB_HEADER_BYTES];
	fin.read((char*)hb, DB_HEADER_BYTES);
	if(!fin){ err = path + ": Unable to read header"; return false; }
	unpack_db_header(hb, &header);
	fin.seekg(0, std::ios::end);
	const uint64_t fsize = (uint64_t)fin.tellg();
	if(header.info_start > fsize || fsize - header.info_start < 8ull*header.num_filter){
		err = path + ": metadata index lies outside the file";
		return false;
	}
	// info_loc[N] then the FilterInfo records (build_db.cpp:371-416): read the tail once
	tail_start = header.info_start;
	tail.resize(fsize - tail_start);
	fin.seekg((std::streamoff)tail_start);
	fin.read((char*)tail.data(), (std::streamsize)tail.size());
	if(!fin){ err = path + ": Unable to read metadata"; return false; }
	info_loc.resize(header.num_filter);
	for(uint32_t j = 0; j < header.num_filter; ++j){ info_loc[j] = rd64(tail.data() + 8ull*j); }
	return true;
}

bool DbInfo::info(uint32_t column, FilterInfo &fi) const
{
	if(column >= header.num_filter){ return false; }
	const uint64_t loc = info_loc[column];          // kwage.cpp:505-515
	if(loc < tail_start || loc >= tail_start + tail.size()){ return false; }
	return parse_filter_info(tail.data() + (loc - tail_start), tail.size() - (loc - tail_start), fi);
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

bool SeqFile::open(const std::string &path, std::string &err)
{
	close();
	type = seq_file_type(path);
	if(type == 2){ err = "SequenceIterator: Unknown file type"; return false; }
	fin = gzopen(path.c_str(), "r");      // plain and gzip files alike, parse_sequence.cpp:40
	if(!fin){ err = "Error opening: " + path; return false; }
	return true;
}

static inline bool has_eol(const char *b) { return strpbrk(b, "\n\r") != nullptr; }

// Returns 1 with (curr_defline, seq) set, 0 at end of file, -1 on a malformed FASTQ record.
int SeqFile::next(std::string &err)
{
	if(!fin){ return 0; }
	const int buffer_len = 2048;            // gzgets chunking is observable in over-long deflines
	char buffer[buffer_len];
	gzFile f = (gzFile)fin;
	seq.clear();

	if(type == 0){      // parse_sequence.cpp:72-151
		std::string info;
		while(gzgets(f, buffer, buffer_len)){
			if(strchr(buffer, '>') != nullptr){         // ANY line containing '>' is a defline (:86)
				info.clear();
				for(char *p = buffer; *p; ++p){ if(*p != '\n' && *p != '\r'){ info.push_back(*p); } }
				if(!has_eol(buffer)){
					// :100-108 -- continuation chunks are appended until one holds the end of line;
					// that last chunk is consumed but NOT appended
					while(gzgets(f, buffer, buffer_len) && !has_eol(buffer)){
						for(char *p = buffer; *p; ++p){ if(*p != '\n' && *p != '\r'){ info.push_back(*p); } }
					}
				}
				size_t s = 0;
				while(s < info.size() && (isspace((unsigned char)info[s]) || info[s] == '>')){ ++s; }
				info.erase(0, s);
				if(!seq.empty()){
					std::swap(curr_defline, next_defline);
					next_defline = info;
					return 1;
				}
				next_defline = info;
			}
			else{
				for(char *p = buffer; *p; ++p){
					if(!isspace((unsigned char)*p)){ seq.push_back((char)toupper((unsigned char)*p)); }
				}
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
	std::string info;
	while(true){
		if(gzgets(f, buffer, buffer_len) == nullptr){ close(); return 0; }
		for(char *p = buffer; *p; ++p){ if(*p != '\n' && *p != '\r'){ info.push_back(*p); } }
		if(has_eol(buffer)){ break; }
	}
	size_t s = 0;
	while(s < info.size() && (isspace((unsigned char)info[s]) || info[s] == '@')){ ++s; }
	curr_defline = info.substr(s);
	while(true){
		if(gzgets(f, buffer, buffer_len) == nullptr){ err = "next_fastq: Unable to read sequence"; return -1; }
		for(char *p = buffer; *p; ++p){
			if(!isspace((unsigned char)*p)){ seq.push_back((char)toupper((unsigned char)*p)); }
		}
		if(has_eol(buffer)){ break; }
	}
	if(gzgets(f, buffer, buffer_len) == nullptr){ err = "next_fastq: Unable to read '+'"; return -1; }
	if(!has_eol(buffer)){ err = "next_fastq: Error reading '+' delimiter"; return -1; }
	while(true){
		if(gzgets(f, buffer, buffer_len) == nullptr){ err = "next_fastq: Unable to read quality"; return -1; }
		if(has_eol(buffer)){ break; }
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
