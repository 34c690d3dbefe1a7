// oracle/ref_driver/ref_tool.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A small driver that LINKS THE REFERENCE's own objects (compiled in place from
// /root/reference by oracle/Makefile; nothing is copied) so that golden vectors come from
// the reference's code, not from this repo's restatement:
//
//   ref_tool mkdb <spec>            build .bloom files with the reference's BloomFilter /
//                                   binary_write and transpose them with the reference's
//                                   build_db() (build_db.cpp:24) -> a real `.db` fixture
//   ref_tool kmers <k> <nhash> <seq> print, for every valid k-mer position, the reference's
//                                   CanonicalWord (word.h:165) and bigsi_hash (hash.cpp:79)
//   ref_tool accession <str>        print str_to_accession / accession_to_str round trip
//   ref_tool param <k> <num_kmer> <p> <minL> <maxL>   the reference's optimal_bloom_param (bloom.cpp:10-68)
//   ref_tool build <out.db> <k> <L> <nhash> <list>   run the reference's build_db() on existing
//                                   .bloom files (one path per line in <list>): CPU baseline of
//                                   the device builder
//   ref_tool zslice <slice_bytes> <slices.bin> <out.bin>    run the reference's CompressSlice (slice_z.h:153-267,
//                                   included in place) over consecutive <slice_bytes>-byte slices; per slice
//                                   write u32 payload length + payload (the deflate stream, or the raw slice
//                                   when compress() says "not smaller")
//   ref_tool zinflate <slice_bytes> <records.bin> <out.bin> the inverse with the reference's InflateSlice
//                                   (slice_z.h:12-150): records as written by zslice (payload length ==
//                                   slice_bytes means stored raw) -> the concatenated slices
//
// Spec format for mkdb (one record per line, fields separated by single TABs):
//   DB <out.db> <kmer_len> <log_2_filter_len> <num_hash> <tmp_dir>
//   F  <run_accession> <noise_seed> <noise_bits>     start a new filter (column)
//   S  <sequence>                                    insert all canonical k-mers of sequence
//   M  <field> <value>                               set a FilterInfo string/accession field
//   A  <tag> <value>                                 add a sample attribute
//   N  <spots> <bases>                               number_of_spots / number_of_bases
//   D  <YYYY-MM-DD...>                               date_received
//   KEEP                                             leave the intermediate .bloom files in <tmp_dir>
// The order of F records is the column order of the database.

#include <iostream>
#include <fstream>
#include <sstream>
#include <iomanip>
#include <deque>
#include <vector>
#include <string>
#include <cstdio>
#include <cstdlib>

#include "kwage.h"
#include "bloom.h"
#include "hash.h"
#include "word.h"
#include "maestro.h"
#include "binary_io.h"
#include "sra_accession.h"
#include "date.h"
#include <cstring>
#include "slice_z.h"

using namespace std;

// build_db.cpp:22 expects these globals from its host program.
int mpi_rank = 0;
int mpi_numtasks = 1;

static vector<string> split_tabs(const string &line)
{
	vector<string> out;
	string cur;
	for(size_t i = 0; i < line.size(); ++i){
		if(line[i] == '\t'){ out.push_back(cur); cur.clear(); }
		else if(line[i] != '\r'){ cur.push_back(line[i]); }
	}
	out.push_back(cur);
	return out;
}

static void insert_sequence(BloomFilter &bf, const BloomParam &param, const string &seq)
{
	const size_t filter_len = param.filter_len();

	ForEachDuplexWord(seq.c_str(), seq.c_str() + seq.size(), param.kmer_len)
		if(ValidWord){
			const Word w = CanonicalWord;
			for(uint32_t h = 0; h < param.num_hash; ++h){
				bf.set_bit( bigsi_hash(w, param.kmer_len, h, param.hash_func) % filter_len );
			}
		}
	EndWord
}

struct PendingFilter
{
	BloomFilter *bf;
	FilterInfo info;
	PendingFilter() : bf(NULL) {}
};

static int cmd_mkdb(const char *spec_path)
{
	ifstream fin(spec_path);
	if(!fin){ cerr << "ref_tool: cannot open spec " << spec_path << endl; return 2; }

	string out_db, tmp_dir;
	BloomParam param;
	deque<string> bloom_files;
	PendingFilter cur;
	size_t filter_index = 0;
	bool keep_bloom = false;

	// Flush the current filter to a .bloom file via the reference serializer.
	auto flush = [&]() {
		if(cur.bf == NULL){ return; }
		cur.bf->set_info(cur.info);
		cur.bf->update_crc32();
		ostringstream name;
		name << tmp_dir << "/f" << setw(6) << setfill('0') << filter_index++ << ".bloom";
		ofstream fout(name.str().c_str(), ios::binary);
		binary_write(fout, *cur.bf);
		fout.close();
		bloom_files.push_back(name.str());
		delete cur.bf;
		cur = PendingFilter();
	};

	string line;
	while(getline(fin, line)){
		if(line.empty() || line[0] == '#'){ continue; }
		const vector<string> f = split_tabs(line);
		const string &tag = f[0];

		if(tag == "DB"){
			if(f.size() != 6){ cerr << "ref_tool: bad DB line" << endl; return 2; }
			out_db = f[1];
			param.kmer_len = atoi(f[2].c_str());
			param.log_2_filter_len = atoi(f[3].c_str());
			param.num_hash = atoi(f[4].c_str());
			param.hash_func = MURMUR_HASH_32;
			tmp_dir = f[5];
		}
		else if(tag == "KEEP"){
			keep_bloom = true;
		}
		else if(tag == "F"){
			flush();
			if(f.size() != 4){ cerr << "ref_tool: bad F line" << endl; return 2; }
			cur.bf = new BloomFilter(param);
			cur.bf->unset_all_bits();
			cur.info = FilterInfo();
			cur.info.run_accession = str_to_accession(f[1]);
			// Background noise: a tiny LCG (this driver's own; only has to be reproducible).
			unsigned long long s = strtoull(f[2].c_str(), NULL, 10)*2862933555777941757ULL + 3037000493ULL;
			const unsigned long long nbits = strtoull(f[3].c_str(), NULL, 10);
			for(unsigned long long i = 0; i < nbits; ++i){
				s = s*6364136223846793005ULL + 1442695040888963407ULL;
				cur.bf->set_bit( (s >> 24) % param.filter_len() );
			}
		}
		else if(tag == "S"){
			insert_sequence(*cur.bf, param, f.at(1));
		}
		else if(tag == "M"){
			const string &k = f.at(1); const string &v = f.at(2);
			if(k == "experiment_accession") cur.info.experiment_accession = str_to_accession(v);
			else if(k == "sample_accession") cur.info.sample_accession = str_to_accession(v);
			else if(k == "study_accession") cur.info.study_accession = str_to_accession(v);
			else if(k == "experiment_title") cur.info.experiment_title = v;
			else if(k == "experiment_design_description") cur.info.experiment_design_description = v;
			else if(k == "experiment_library_name") cur.info.experiment_library_name = v;
			else if(k == "experiment_library_strategy") cur.info.experiment_library_strategy = v;
			else if(k == "experiment_library_source") cur.info.experiment_library_source = v;
			else if(k == "experiment_library_selection") cur.info.experiment_library_selection = v;
			else if(k == "experiment_instrument_model") cur.info.experiment_instrument_model = v;
			else if(k == "sample_taxa") cur.info.sample_taxa = v;
			else if(k == "study_title") cur.info.study_title = v;
			else if(k == "study_abstract") cur.info.study_abstract = v;
			else { cerr << "ref_tool: unknown field " << k << endl; return 2; }
		}
		else if(tag == "A"){
			cur.info.sample_attributes.insert( make_pair(f.at(1), f.at(2)) );
		}
		else if(tag == "N"){
			cur.info.number_of_spots = strtoull(f.at(1).c_str(), NULL, 10);
			cur.info.number_of_bases = strtoull(f.at(2).c_str(), NULL, 10);
		}
		else if(tag == "D"){
			cur.info.date_received = Date(f.at(1));
		}
		else{
			cerr << "ref_tool: unknown record " << tag << endl;
			return 2;
		}
	}
	flush();

	if( !build_db(out_db, param, bloom_files) ){
		cerr << "ref_tool: build_db failed" << endl;
		return 1;
	}

	for(deque<string>::const_iterator i = bloom_files.begin(); !keep_bloom && i != bloom_files.end(); ++i){
		remove(i->c_str());
	}
	return 0;
}

static int cmd_kmers(int k, int nhash, const string &seq)
{
	ForEachDuplexWord(seq.c_str(), seq.c_str() + seq.size(), k)
		if(ValidWord){
			const Word w = CanonicalWord;
			cout << Loc5 << '\t' << hex << setw(16) << setfill('0') << w;
			for(int h = 0; h < nhash; ++h){
				cout << '\t' << setw(8) << setfill('0')
					<< (unsigned int)bigsi_hash(w, k, h, MURMUR_HASH_32);
			}
			cout << dec << '\n';
		}
	EndWord
	return 0;
}

// The codec works on slices of at most MAX_COMPRESSED_BYTES (256 bytes = the 2048 filters a database file holds,
// slice_z.h:8): InflateSlice always offers exactly that much output room, so both sides are instantiated with it.
static bool read_all(const char *path, vector<unsigned char> &buf)
{
	ifstream f(path, ios::binary);
	if(!f){ return false; }
	buf.assign(istreambuf_iterator<char>(f), istreambuf_iterator<char>());
	return true;
}

static int cmd_zslice(unsigned int slice_bytes, const char *in_path, const char *out_path)
{
	vector<unsigned char> in;
	if(slice_bytes == 0 || slice_bytes > MAX_COMPRESSED_BYTES || !read_all(in_path, in) || in.size() % slice_bytes){
		cerr << "zslice: bad slice size or input" << endl;
		return 1;
	}
	ofstream out(out_path, ios::binary);
	CompressSlice<MAX_COMPRESSED_BYTES> z;
	for(size_t o = 0; o < in.size(); o += slice_bytes){
		const bool smaller = z.compress(&in[o], slice_bytes);
		const uint32_t len = smaller ? z.size() : slice_bytes;
		out.write((const char*)&len, 4);
		out.write(smaller ? (const char*)z.ptr() : (const char*)&in[o], len);
	}
	return out ? 0 : 1;
}

static int cmd_zinflate(unsigned int slice_bytes, const char *in_path, const char *out_path)
{
	vector<unsigned char> in;
	if(slice_bytes == 0 || slice_bytes > MAX_COMPRESSED_BYTES || !read_all(in_path, in)){
		cerr << "zinflate: bad slice size or input" << endl;
		return 1;
	}
	ofstream out(out_path, ios::binary);
	InflateSlice<MAX_COMPRESSED_BYTES> z;
	for(size_t o = 0; o < in.size(); ){
		uint32_t len;
		if(o + 4 > in.size()){ cerr << "zinflate: truncated record" << endl; return 1; }
		memcpy(&len, &in[o], 4);
		o += 4;
		if(len > slice_bytes || o + len > in.size()){ cerr << "zinflate: bad record length" << endl; return 1; }
		if(len == slice_bytes){ out.write((const char*)&in[o], len); }
		else{
			z.inflate(&in[o], len);
			if(z.size() != slice_bytes){ cerr << "zinflate: slice inflated to " << z.size() << " bytes" << endl; return 1; }
			out.write((const char*)z.ptr(), slice_bytes);
		}
		o += len;
	}
	return out ? 0 : 1;
}

int main(int argc, char *argv[])
{
	try{
		if(argc == 5 && string(argv[1]) == "zslice"){ return cmd_zslice(atoi(argv[2]), argv[3], argv[4]); }
		if(argc == 5 && string(argv[1]) == "zinflate"){ return cmd_zinflate(atoi(argv[2]), argv[3], argv[4]); }
		if(argc == 3 && string(argv[1]) == "mkdb"){
			return cmd_mkdb(argv[2]);
		}
		if(argc == 5 && string(argv[1]) == "kmers"){
			return cmd_kmers(atoi(argv[2]), atoi(argv[3]), argv[4]);
		}
		if(argc == 7 && string(argv[1]) == "param"){
			try{
				const BloomParam p = optimal_bloom_param(atoi(argv[2]), strtoull(argv[3], NULL, 10), (float)atof(argv[4]),
					MURMUR_HASH_32, atoi(argv[5]), atoi(argv[6]));
				cout << p.log_2_filter_len << '\t' << p.num_hash << '\n';
			}
			catch(const char *error){
				cout << "throw" << '\n';
			}
			return 0;
		}
		if(argc == 7 && string(argv[1]) == "build"){
			BloomParam param;
			param.kmer_len = atoi(argv[3]);
			param.log_2_filter_len = atoi(argv[4]);
			param.num_hash = atoi(argv[5]);
			param.hash_func = MURMUR_HASH_32;
			deque<string> bloom_files;
			ifstream fl(argv[6]);
			string line;
			while(getline(fl, line)){ if(!line.empty()){ bloom_files.push_back(line); } }
			return build_db(argv[2], param, bloom_files) ? 0 : 1;
		}
		if(argc == 3 && string(argv[1]) == "accession"){
			const SraAccession a = str_to_accession(argv[2]);
			cout << a << '\t' << accession_to_str(a) << '\n';
			return 0;
		}
		cerr << "usage: ref_tool mkdb <spec> | kmers <k> <nhash> <seq> | accession <str> | build <out.db> <k> <L> <nhash> <list> | zslice <bytes> <in> <out> | zinflate <bytes> <in> <out>" << endl;
		return 2;
	}
	catch(const char *error){
		cerr << "ref_tool: caught " << error << endl;
		return 1;
	}
}
