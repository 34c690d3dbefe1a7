"""ctypes loader for kwage_amd/lib/libkwage_amd.so (the C ABI of include/kwage_amd.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = os.path.join(_HERE, "lib", "libkwage_amd.so")
KWAGE_BIN = os.path.join(_HERE, "bin", "kwage")
KWAGE_DBTOOL_BIN = os.path.join(_HERE, "bin", "kwage_dbtool")


class KwageError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("kwage_amd error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


# The engine pipelines searches over two HIP streams of its own while the host program (torch, RCCL) runs
# several more.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); when a search
# stream shares a queue with a torch/RCCL stream the pipeline serialises (measured: +0.2 ms on a 1.9 ms step,
# DESIGN.md section 6).  The variable is read when the HIP runtime initialises, i.e. at the first device
# call, so setting it at import time is early enough; an explicit setting by the user wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def lib_path() -> str:
    return _LIB


def build_native(force: bool = False) -> str:
    """Compile the HIP engine + CLI for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _CSRC, "-j4", "all"])
    return _LIB


def ensure_built() -> str:
    """Build the native library + CLI if (and only if) they are missing.  Used by bench.py / smoke();
    compiling the HIP extension is not a fallback -- nothing runs without it."""
    if not (os.path.exists(_LIB) and os.path.exists(KWAGE_BIN) and os.path.exists(KWAGE_DBTOOL_BIN)):
        build_native()
    return _LIB


class Hit(C.Structure):
    _fields_ = [("query", C.c_uint32), ("column", C.c_uint32), ("num_match", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("kmer_len", C.c_uint32), ("num_hash", C.c_uint32),
                ("log_2_filter_len", C.c_uint32), ("hash_func", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("n_hits", C.c_uint64), ("hits", C.POINTER(Hit)), ("n_queries", C.c_uint32),
                ("num_query_kmer", C.POINTER(C.c_uint32)), ("query_threshold", C.POINTER(C.c_uint32)),
                ("total_kmers", C.c_uint64), ("bit_tests", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
                ("kmer_kernel_ms", C.c_float), ("search_kernel_ms", C.c_float),
                ("search_kernel_launches", C.c_uint32), ("search_kernel", C.c_char_p)]


class BuildStats(C.Structure):
    _fields_ = [("bits_transposed", C.c_uint64), ("transpose_kernel_ms", C.c_float), ("db_bytes", C.c_uint64)]


class BloomCounterStats(C.Structure):
    _fields_ = [("num_valid_kmer", C.c_uint64), ("num_bp", C.c_uint64), ("positions", C.c_uint64),
                ("occurrences_committed", C.c_uint64), ("chunks", C.c_uint64), ("rounds", C.c_uint64),
                ("max_rounds", C.c_uint32), ("reserved", C.c_uint32), ("add_ms", C.c_double)]


class SampleInfo(C.Structure):
    _fields_ = [(n, C.c_char_p) for n in ("run_accession", "experiment_accession", "sample_accession", "study_accession",
                                          "experiment_title", "experiment_design_description", "experiment_library_name",
                                          "experiment_library_strategy", "experiment_library_source",
                                          "experiment_library_selection", "experiment_instrument_model", "sample_taxa",
                                          "study_title", "study_abstract")] + \
               [("attribute_tags", C.POINTER(C.c_char_p)), ("attribute_values", C.POINTER(C.c_char_p)),
                ("num_attributes", C.c_uint32), ("number_of_spots", C.c_uint64), ("number_of_bases", C.c_uint64),
                ("day", C.c_uint32), ("month", C.c_uint32), ("year", C.c_uint32)]


class DbHeader(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint32), ("crc32", C.c_uint32),
                ("kmer_len", C.c_uint32), ("num_hash", C.c_uint32), ("log_2_filter_len", C.c_uint32),
                ("num_filter", C.c_uint32), ("hash_func", C.c_int32), ("compression", C.c_uint32),
                ("info_start", C.c_uint64)]


# every symbol include/kwage_amd.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SIGNATURES = [
    ("kwage_last_error", C.c_char_p, []),
    ("kwage_abi_version", C.c_uint32, []),
    ("kwage_device_count", C.c_int, []),
    ("kwage_init", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("kwage_shutdown", None, [_P]),
    ("kwage_mem_info", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("kwage_device_fingerprint", C.c_int, [_P, C.c_char_p, C.c_uint64]),
    ("kwage_set_load_progress", None, [_P, C.POINTER(C.c_uint64)]),
    ("kwage_sort_hits", None, [C.c_void_p, C.c_uint64]),
    ("kwage_sync", C.c_int, [_P]),
    ("kwage_ctx_set_tuning", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("kwage_ctx_get_tuning", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    ("kwage_ctx_scratch_nonzero", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("kwage_ctx_refine_stats", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("kwage_group_create", C.c_int, [_P, C.POINTER(Params), C.c_uint64, C.POINTER(_P)]),
    ("kwage_group_destroy", None, [_P]),
    ("kwage_group_create_sparse", C.c_int, [_P, C.POINTER(Params), C.c_uint64, _P, C.c_uint64, C.POINTER(_P)]),
    ("kwage_group_add_columns", C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("kwage_group_add_db_file", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("kwage_group_add_db_files", C.c_int, [_P, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("kwage_group_add_random_columns", C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("kwage_group_set_bits", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("kwage_group_read_rows", C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64]),
    ("kwage_group_finalize", C.c_int, [_P]),
    ("kwage_group_num_columns", C.c_uint64, [_P]),
    ("kwage_group_column_span", C.c_uint64, [_P]),
    ("kwage_group_row_bytes", C.c_uint64, [_P]),
    ("kwage_group_row_stride", C.c_uint64, [_P]),
    ("kwage_group_device_bytes", C.c_uint64, [_P]),
    ("kwage_group_placement", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("kwage_group_params", C.c_int, [_P, C.POINTER(Params)]),
    ("kwage_batch_create", C.c_int, [_P, C.c_char_p, _P, C.c_uint32, C.POINTER(_P)]),
    ("kwage_batch_destroy", None, [_P]),
    ("kwage_batch_num_queries", C.c_uint32, [_P]),
    ("kwage_search", C.c_int, [_P, _P, C.c_float, C.c_uint32, C.POINTER(C.POINTER(Result))]),
    ("kwage_result_free", None, [C.POINTER(Result)]),
    ("kwage_search_submit", C.c_int, [_P, _P, C.c_float, C.c_uint32, C.POINTER(_P)]),
    ("kwage_search_collect", C.c_int, [_P, C.POINTER(C.POINTER(Result))]),
    ("kwage_search_poll", C.c_int, [_P]),
    ("kwage_search_device", C.c_int, [_P, _P, C.c_float, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64), _P]),
    ("kwage_search_device_submit", C.c_int, [_P, _P, C.c_float, C.c_uint32, _P, C.c_uint64, _P, C.POINTER(_P)]),
    ("kwage_search_device_append_submit", C.c_int, [_P, _P, C.c_float, C.c_uint32, _P, C.c_uint64, _P, C.c_uint32, C.c_int, C.POINTER(_P)]),
    ("kwage_search_device_collect", C.c_int, [_P, C.POINTER(C.c_uint64), _P, C.POINTER(C.c_float)]),
    ("kwage_hash_batch", C.c_int, [_P, C.POINTER(Params), _P, _P, _P, _P, _P]),
    ("kwage_stream_read_gbps", C.c_int, [_P, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    ("kwage_build_db", C.c_int, [_P, C.c_char_p, C.POINTER(Params), C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(BuildStats)]),
    ("kwage_optimal_bloom_param", C.c_int, [C.c_uint32, C.c_uint64, C.c_float, C.c_uint32, C.c_uint32, C.POINTER(Params)]),
    ("kwage_count_distinct_kmers", C.c_int, [_P, _P, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("kwage_bloom_bits_from_batch", C.c_int, [_P, C.POINTER(Params), _P, _P, C.POINTER(C.c_uint64)]),
    ("kwage_make_bloom", C.c_int, [_P, C.POINTER(Params), C.c_char_p, _P, C.c_uint32, C.POINTER(SampleInfo), C.c_char_p, C.POINTER(C.c_uint64)]),
    ("kwage_counting_filter_log2", C.c_uint32, [C.c_uint64]),
    ("kwage_approximate_max_kmers", C.c_uint64, [C.c_float, C.c_uint32, C.c_uint32]),
    ("kwage_bloom_counter_create", C.c_int, [_P, C.c_uint32, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    ("kwage_bloom_counter_destroy", None, [_P]),
    ("kwage_bloom_counter_reset", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("kwage_bloom_counter_add", C.c_int, [_P, C.c_char_p, _P, C.c_uint32]),
    ("kwage_bloom_counter_flush", C.c_int, [_P]),
    ("kwage_bloom_counter_get_stats", C.c_int, [_P, C.POINTER(BloomCounterStats)]),
    ("kwage_bloom_counter_read_counts", C.c_int, [_P, C.c_uint64, C.c_uint64, _P]),
    ("kwage_bloom_counter_read_valid_bits", C.c_int, [_P, C.c_uint32, C.c_uint64, C.c_uint64, _P]),
    ("kwage_bloom_counter_finish", C.c_int, [_P, C.c_float, C.c_uint32, C.POINTER(SampleInfo), C.c_char_p, C.POINTER(Params), C.POINTER(C.c_int)]),
    ("kwage_repack_db", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32]),
    ("kwage_db_read_header", C.c_int, [C.c_char_p, C.POINTER(DbHeader)]),
    ("kwage_db_read_slices", C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.c_uint64, C.c_void_p]),
    ("kwage_db_compress", C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32]),
    ("kwage_db_decompress", C.c_int, [C.c_char_p, C.c_char_p]),
    ("kwage_dbinfo_open", C.c_int, [C.c_char_p, C.POINTER(_P)]),
    ("kwage_dbinfo_close", None, [_P]),
    ("kwage_dbinfo_num_filter", C.c_uint32, [_P]),
    ("kwage_dbinfo_csv_string", C.c_int, [_P, C.c_uint32, C.c_char_p, C.c_size_t]),
    ("kwage_dbinfo_json_string", C.c_int64, [_P, C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t]),
    ("kwage_str_to_accession", C.c_int, [C.c_char_p, C.POINTER(C.c_uint64)]),
    ("kwage_accession_to_str", C.c_int, [C.c_uint64, C.c_char_p, C.c_size_t]),
    ("kwage_seqfile_open", C.c_int, [C.c_char_p, C.POINTER(_P)]),
    ("kwage_seqfile_next", C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]),
    ("kwage_seqfile_close", None, [_P]),
    ("kwage_query_threshold", C.c_uint32, [C.c_float, C.c_uint32]),
]

EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


def _preload_torch_hip_runtime() -> None:
    """One process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64.so (same soname
    as /opt/rocm's); if this library pulled in the system copy first, a later `import torch` would mix
    the two and find no GPUs.  So when torch is installed, load its copy first (without importing torch):
    the soname is then already resolved when libkwage_amd.so is opened.  The `kwage` CLI never loads
    torch and uses the system runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib() -> C.CDLL:
    """The loaded C-ABI library. Raises (loudly) if it has not been built: no fallback."""
    global _lib
    if _lib is None:
        _preload_torch_hip_runtime()
        if not os.path.exists(_LIB):
            raise KwageError(-2, "%s is missing: build it with `make -C kwage_amd/csrc` "
                                 "(or __graft_entry__.build()); there is no CPU fallback" % _LIB)
        L = C.CDLL(_LIB)
        for name, res, args in _SIGNATURES:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise KwageError(rc, lib().kwage_last_error().decode("latin-1"))
