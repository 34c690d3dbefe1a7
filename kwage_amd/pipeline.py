"""FASTA/FASTQ per sample -> `.bloom` -> `.db`, entirely through the C ABI (device k-mer stage, device
Bloom bits, device bit transpose).  This is the database-construction side of KWAGE for inputs whose
every k-mer counts (assemblies / genomes: the reference's min_kmer_count == 1 case); like the
reference's maestro it gives each sample the smallest Bloom parameters that meet the false-positive
bound (optimal_bloom_param, bloom.cpp:10-68) and writes one `.db` per distinct parameter set, at most
2048 samples per file (options.h:137 MAX_NUM_FILTER_CHUNK)."""
from __future__ import annotations

import ctypes as C
import os
import tempfile
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .engine import Batch, Context
from .native import BuildStats, Params, SampleInfo, check, lib

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


def build_databases(ctx: Context, samples: Sequence[Tuple[str, str]], out_prefix: str, kmer_len: int = 31,
                    false_positive: float = 0.25, min_log_2_filter_len: int = 18, max_log_2_filter_len: int = 32,
                    work_dir: Optional[str] = None) -> List[str]:
    """samples: (run accession, sequence file).  Returns the `.db` files written."""
    L = lib()
    tmp = work_dir or tempfile.mkdtemp(prefix="kwage_bloom_")
    groups: Dict[Tuple[int, int], List[str]] = {}
    for acc, path in samples:
        seqs = [s for _, s in read_sequences(path)]
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
