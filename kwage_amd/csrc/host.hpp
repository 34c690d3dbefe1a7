// kwage_amd/csrc/host.hpp -- host-side types shared by the C ABI wrappers and the kwage CLI.
#ifndef KWAGE_AMD_HOST_HPP
#define KWAGE_AMD_HOST_HPP

#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "kwage_amd.h"

namespace kwage {

// Sample metadata of one column: the 18 serialized members of the reference's FilterInfo
// (bloom.h:478-496) in file order.
struct FilterInfo {
	uint64_t run_accession = 0, experiment_accession = 0;
	std::string experiment_title, experiment_design_description, experiment_library_name,
	            experiment_library_strategy, experiment_library_source, experiment_library_selection,
	            experiment_instrument_model;
	uint64_t sample_accession = 0;
	std::string sample_taxa;
	std::unordered_map<std::string, std::string> sample_attributes;   // bloom.h:16 MAP
	uint64_t study_accession = 0;
	std::string study_title, study_abstract;
	uint64_t number_of_spots = 0, number_of_bases = 0;
	uint32_t day = 0, month = 0, year = 0;

	std::string csv_string() const;
	std::string json_string(const std::string &prefix) const;
};

bool parse_filter_info(const unsigned char *buf, size_t len, FilterInfo &fi, size_t *consumed = nullptr);
// binary_write<FilterInfo> (binary_io.cpp:154-163): members in order, strings NUL-terminated, the
// attribute map as size_t count + pairs in the container's iteration order.
void pack_filter_info(const FilterInfo &fi, std::vector<unsigned char> &out);
bool str_to_accession(const std::string &s, uint64_t &out);
std::string accession_to_str(uint64_t acc);
bool find_file_extension(const std::string &path, const char *ext);

// info_loc[] + FilterInfo records of one `.db` file.  open() reads the header and checks where the metadata lies; the
// metadata itself is read on first use, once: a database of a million samples carries a GB of it, a report touches the
// records of the samples that were hit (the reference seeks to them one by one, kwage.cpp:505-515).
struct DbInfo {
	kwage_db_header header{};
	uint64_t tail_start = 0, tail_bytes = 0;
	std::string path;
	mutable std::vector<unsigned char> tail;
	mutable std::vector<uint64_t> info_loc;
	bool open(const std::string &path, std::string &err);
	bool load(std::string &err) const;                  // thread-safe; what info() does first
	bool info(uint32_t column, FilterInfo &fi) const;
private:
	mutable std::mutex once;
	mutable int state = 0;                               // 0 not read yet, 1 read, -1 unreadable
};

// Reader of the bit-slice block of a `.db` file, raw (compression 0, the only layout the reference
// writes: build_db.cpp:197-199) or this repo's compressed container (compression 2, see
// kwage_db_compress in include/kwage_amd.h).  Compressed slices are inflated on the host.
struct DbSliceSource {
	int fd = -1;
	kwage_db_header header{};
	uint64_t slice_size = 0, nrows = 0;
	std::vector<uint64_t> offsets;          // compressed only: nrows + 1 absolute file offsets
	bool open(const std::string &path, std::string &err);
	bool read_rows(uint64_t r0, uint64_t nr, unsigned char *dst, std::string &err);   // nr*slice_size bytes
	// the listed slices (any order), n*slice_size bytes: what the reference's seekg + read per addressed slice does
	// (kwage.cpp:414-416), batched and on several threads
	bool read_row_list(const uint32_t *rows, uint64_t n, unsigned char *dst, std::string &err, unsigned max_threads = 16);
	bool slice_crc32(uint32_t &crc, std::string &err);      // CRC32 of the (uncompressed) slice block, as build_db stores it in the header
	void close();
	DbSliceSource() = default;
	DbSliceSource(const DbSliceSource&) = delete;
	DbSliceSource& operator=(const DbSliceSource&) = delete;
	~DbSliceSource() { close(); }
};

static const uint32_t KWAGE_COMPRESSION_NONE = 0;       // reference kwage.h:16-20 NO_COMPRESSION
static const uint32_t KWAGE_COMPRESSION_DEFLATE = 2;    // reference enum slot RLE_HUFFMAN_COMPRESSION

// FASTA / FASTQ record iterator with the reference SequenceIterator's exact behaviour.
struct SeqFile {
	void *fin;
	int type;
	std::string seq, curr_defline, next_defline;
	std::vector<char> rbuf;            // block buffer behind get_line()
	size_t rpos = 0, rend = 0;
	char *get_line(char *out, int len);
	int get_chunk(char *out, int len);
	std::string scratch;
	SeqFile();
	~SeqFile();
	bool open(const std::string &path, std::string &err);
	int next(std::string &err);
	void close();
	SeqFile(const SeqFile&) = delete;
	SeqFile& operator=(const SeqFile&) = delete;
};

}  // namespace kwage

#endif
