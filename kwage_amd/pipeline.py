"""FASTA/FASTQ per sample -> `.bloom` -> `.db`, entirely through the C ABI (device k-mer stage, device
Bloom bits or device counting pass, device bit transpose).  This is the database-construction side of KWAGE:
exact k-mer sets for assemblies / genomes, the reference's minimum-k-mer-count pass for read sets
(BloomCounter); like the reference's maestro it gives each sample the smallest Bloom parameters that meet the false-positive
bound (optimal_bloom_param, bloom.cpp:10-68) and writes one `.db` per distinct parameter set, at most
2048 samples per file (options.h:137 MAX_NUM_FILTER_CHUNK)."""
from __future__ import annotations

import ctypes as C
import os
import tempfile
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .engine import Batch, Context
from .native import BloomCounterStats, Params, SampleInfo, check, lib

MAX_NUM_FILTER_CHUNK = 2048


def read_sequences(path: str) -> List[Tuple[str, bytes]]:
    """All records of a FASTA/FASTQ(.gz) file through the library's SequenceIterator mirror."""
    L = lib()
    f = C.c_void_p()
    check(L.kwage_seqfile_open(path.encode(), C.byref(f)))
    out = []
    try:
        d, s, n = C.c_char_p(), C.c_char_p(), C.c_uint64()
        while True:
            r = L.kwage_seqfile_next(f, C.byref(d), C.byref(s), C.byref(n))
            check(r if r < 0 else 0)
            if r == 0:
                break
            out.append((d.value.decode("latin-1"), s.value))
    finally:
        L.kwage_seqfile_close(f)
    return out


class BloomCounter:
    """make_bloom_filter()'s counting pass (make_bloom.cpp:76-504) on the device: add() fragments in read
    order, finish() -> (status, Params) and optionally the `.bloom` file.  See include/kwage_amd.h."""

    def __init__(self, ctx: Context, kmer_len: int = 31, min_kmer_count: int = 5, log_2_counting_filter_len: int = 0,
                 max_log_2_filter_len: int = 32, num_bp: int = 0, hash_func: int = 0):
        L = lib()
        if not log_2_counting_filter_len:
            log_2_counting_filter_len = L.kwage_counting_filter_log2(num_bp)          # make_bloom.cpp:105-130
        self.log_2_counting_filter_len, self.max_log_2_filter_len = log_2_counting_filter_len, max_log_2_filter_len
        self._h = C.c_void_p()
        check(L.kwage_bloom_counter_create(ctx._h, kmer_len, hash_func, min_kmer_count, log_2_counting_filter_len,
                                           max_log_2_filter_len, C.byref(self._h)))

    def reset(self, min_kmer_count: int, log_2_counting_filter_len: int = 0, num_bp: int = 0):
        """Next sample in the same object (allocations kept); the counting filters may only shrink."""
        if not log_2_counting_filter_len:
            log_2_counting_filter_len = lib().kwage_counting_filter_log2(num_bp)
        check(lib().kwage_bloom_counter_reset(self._h, min_kmer_count, log_2_counting_filter_len))
        self.log_2_counting_filter_len = log_2_counting_filter_len

    def add(self, seqs: Sequence[bytes]):
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        if seqs:
            offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        check(lib().kwage_bloom_counter_add(self._h, b"".join(seqs), offs.ctypes.data, len(seqs)))

    def stats(self) -> BloomCounterStats:
        st = BloomCounterStats()
        check(lib().kwage_bloom_counter_get_stats(self._h, C.byref(st)))
        return st

    def counts(self, first: int = 0, n: int = None) -> np.ndarray:
        n = (1 << self.log_2_counting_filter_len) - first if n is None else n
        out = np.empty(n, dtype=np.uint8)
        check(lib().kwage_bloom_counter_read_counts(self._h, first, n, out.ctypes.data))
        return out

    def valid_bits(self, h: int) -> np.ndarray:
        out = np.empty((1 << self.max_log_2_filter_len) // 8, dtype=np.uint8)
        check(lib().kwage_bloom_counter_read_valid_bits(self._h, h, 0, out.size, out.ctypes.data))
        return out

    def finish(self, false_positive: float = 0.25, min_log_2_filter_len: int = 18, info: SampleInfo = None,
               out_path: str = None):
        prm, status = Params(), C.c_int()
        check(lib().kwage_bloom_counter_finish(self._h, C.c_float(false_positive), min_log_2_filter_len,
                                               C.byref(info) if info is not None else None,
                                               out_path.encode() if out_path else None, C.byref(prm), C.byref(status)))
        return status.value, prm

    def close(self):
        if self._h:
            lib().kwage_bloom_counter_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def build_databases(ctx: Context, samples: Sequence[Tuple[str, str]], out_prefix: str, kmer_len: int = 31,
                    false_positive: float = 0.25, min_log_2_filter_len: int = 18, max_log_2_filter_len: int = 32,
                    work_dir: Optional[str] = None, min_kmer_count: int = 0) -> List[str]:
    """samples: (run accession, sequence file).  Returns the `.db` files written.
    min_kmer_count 0: exact k-mer set of each sample (assemblies / genomes); 1..15: the reference's counting
    pass for read sets (make_bloom.cpp, its default is 5) -- samples it declares INVALID are left out."""
    L = lib()
    tmp = work_dir or tempfile.mkdtemp(prefix="kwage_bloom_")
    groups: Dict[Tuple[int, int], List[str]] = {}
    counter: Optional[BloomCounter] = None       # one object for all samples; re-created only when a sample needs larger counting filters
    for acc, path in samples:
        seqs = [s for _, s in read_sequences(path)]
        if min_kmer_count:
            si = SampleInfo()
            si.run_accession = acc.encode()
            si.number_of_spots = len(seqs)
            si.number_of_bases = sum(len(s) for s in seqs)
            bloom = os.path.join(tmp, acc + ".bloom")
            logc = L.kwage_counting_filter_log2(si.number_of_bases)
            if counter is not None and logc <= counter.capacity_log2:
                counter.reset(min_kmer_count, logc)
            else:
                if counter is not None:
                    counter.close()
                counter = BloomCounter(ctx, kmer_len, min_kmer_count, logc, max_log_2_filter_len)
                counter.capacity_log2 = logc
            counter.add(seqs)
            status, prm = counter.finish(false_positive, min_log_2_filter_len, si, bloom)
            if status == 0:
                groups.setdefault((prm.log_2_filter_len, prm.num_hash), []).append(bloom)
            continue
        b = Batch(ctx, seqs)
        n = C.c_uint64()
        check(L.kwage_count_distinct_kmers(ctx._h, b._h, kmer_len, C.byref(n)))
        b.close()
        prm = Params()
        check(L.kwage_optimal_bloom_param(kmer_len, n.value, C.c_float(false_positive), min_log_2_filter_len,
                                          max_log_2_filter_len, C.byref(prm)))
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        if seqs:
            offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        si = SampleInfo()
        si.run_accession = acc.encode()
        si.number_of_bases = int(offs[-1])
        bloom = os.path.join(tmp, acc + ".bloom")
        check(L.kwage_make_bloom(ctx._h, C.byref(prm), b"".join(seqs), offs.ctypes.data, len(seqs), C.byref(si),
                                 bloom.encode(), None))
        groups.setdefault((prm.log_2_filter_len, prm.num_hash), []).append(bloom)
    if counter is not None:
        counter.close()
    written = []
    for (lg, nh), blooms in sorted(groups.items()):
        for part, i in enumerate(range(0, len(blooms), MAX_NUM_FILTER_CHUNK)):
            chunk = blooms[i:i + MAX_NUM_FILTER_CHUNK]
            out = "%s_L%d_h%d_%03d.db" % (out_prefix, lg, nh, part)
            arr = (C.c_char_p * len(chunk))(*[p.encode() for p in chunk])
            prm = Params(kmer_len, nh, lg, 0)
            check(L.kwage_build_db(ctx._h, out.encode(), C.byref(prm), arr, len(chunk), None))
            written.append(out)
    return written
