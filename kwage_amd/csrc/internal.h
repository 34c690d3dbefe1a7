// kwage_amd/csrc/internal.h -- shared by the HIP engine and the host-side helpers.
#ifndef KWAGE_AMD_INTERNAL_H
#define KWAGE_AMD_INTERNAL_H

#include <cstdarg>
#include <cstdint>
#include <string>

#include "kwage_amd.h"

namespace kwage {

// Thread-local error text behind kwage_last_error().
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Parse the 44-byte little-endian DBFileHeader (reference kwage.h:30-72, binary_io.cpp:255-265).
void unpack_db_header(const unsigned char *b, kwage_db_header *h);

// Validate parameters against the limits the reference compiles in.
int check_params(const kwage_params *p);

// Ordering a large hit list by (query, column) where it lies in device memory, from the run table the gather kernels
// keep (hit_sort.hip; kernels.hpp SearchArgs::runs).  `scratch` is a device block of at least hit_order_scratch_bytes();
// the work is queued on `stream` (a hipStream_t) and not waited for.  *d_ordered: the ordered copy of the list (inside
// `scratch`); *d_total: a device word holding the table's record total, which must equal n_hits.
uint64_t hit_order_scratch_bytes(uint64_t n_hits, uint64_t n_runs);
int order_hits_by_runs(void *stream, const kwage_hit *d_hits, uint64_t n_hits, const void *runs, uint64_t n_runs,
                       void *scratch, uint64_t scratch_bytes, kwage_hit **d_ordered, const uint64_t **d_total);

static const uint32_t KWAGE_MAGIC_NUMBER = 0x20191025u;   // reference kwage.h:22
static const uint32_t DB_HEADER_BYTES = 44;

}  // namespace kwage

#endif
