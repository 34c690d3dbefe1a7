"""kwage_amd -- MI355X-native engine for the `kwage` search path of LANL-Bioinformatics/KWAGE.

The product is the C-ABI library built from kwage_amd/csrc (include/kwage_amd.h) and the
`kwage` command-line program next to it; this package is a thin ctypes mirror of that ABI for
tests, bench.py and multi-GPU hosts.  There is no CPU fallback: importing works anywhere, but
every compute entry point raises if the HIP library or a gfx950 device is missing.
"""
from .native import KwageError, lib, lib_path, build_native   # noqa: F401
from .engine import (Context, Group, Batch, Database, FileDatabase, DatabaseHit, SearchResult, PendingSearch, Params, hash_batch,   # noqa: F401
                     SEARCH_EARLY_EXIT, SEARCH_TIMING, SEARCH_TIMING_KMER)

__all__ = ["KwageError", "lib", "lib_path", "build_native", "Context", "Group", "Batch", "Database", "FileDatabase", "DatabaseHit",
           "SearchResult", "PendingSearch", "Params", "hash_batch", "SEARCH_EARLY_EXIT", "SEARCH_TIMING", "SEARCH_TIMING_KMER"]
