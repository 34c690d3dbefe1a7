"""Python mirror of the C ABI objects (include/kwage_amd.h): Context, Group, Batch, search.

Names and argument meaning follow the reference's search path: a Group is what the
reference reads slice by slice from `.db` files (kwage.cpp:414-416), a Batch is the set of query
strings handed to search() (kwage.cpp:119,137), SearchResult carries what search() appends to
its result map (MatchResult: num_kmers_found, num_query_kmer; kwage.cpp:534).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from . import native
from .native import Params, check, lib

SEARCH_EARLY_EXIT = 1
SEARCH_TIMING = 2
SEARCH_TIMING_KMER = 4

HIT_DTYPE = np.dtype([("query", "<u4"), ("column", "<u4"), ("num_match", "<u4")])


class Context:
    """One GPU + one HIP stream (kwage_ctx)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(lib().kwage_init(device, C.byref(self._h)))
        self.device = device

    def close(self) -> None:
        if self._h:
            lib().kwage_shutdown(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def mem_info(self) -> Tuple[int, int]:
        f, t = C.c_uint64(), C.c_uint64()
        check(lib().kwage_mem_info(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def fingerprint(self) -> dict:
        """Identity of the device (kwage_device_fingerprint): uuid, name, arch, cus, clocks, pci -- filed with every measurement."""
        buf = C.create_string_buffer(512)
        check(lib().kwage_device_fingerprint(self._h, buf, 512))
        out = {}
        for kv in buf.value.decode("latin-1").split(";"):
            k, _, v = kv.partition("=")
            out[k] = int(v) if (v.isdigit() and k not in ("uuid", "pci")) else v
        return out

    def sync(self) -> None:
        check(lib().kwage_sync(self._h))

    def set_tuning(self, name: str, value: int) -> None:
        """One kernel-selection knob of this context (kwage_ctx_set_tuning; tests and tuning tools)."""
        check(lib().kwage_ctx_set_tuning(self._h, name.encode(), int(value)))

    def get_tuning(self, name: str) -> int:
        v = C.c_int64()
        check(lib().kwage_ctx_get_tuning(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def scratch_nonzero(self) -> dict:
        """Non-zero words left in the exchange buffers of the persistent gather kernels (kwage_ctx_scratch_nonzero):
        all zero between searches, or a cut pair was not finished."""
        out = (C.c_uint64 * 5)()
        check(lib().kwage_ctx_scratch_nonzero(self._h, out))
        return dict(zip(("walk_or", "walk_done", "band_or", "band_state", "cwalk_arrived"), (int(x) for x in out)))

    def refine_stats(self) -> list:
        """Per search slot: places of the cluster / item / unit lists the last early-exit search took, and the unit list's
        capacity (kwage_ctx_refine_stats)."""
        out = (C.c_uint64 * 8)()
        check(lib().kwage_ctx_refine_stats(self._h, out))
        return [dict(zip(("clusters", "items", "units", "units_cap"), (int(x) for x in out[4 * k:4 * k + 4]))) for k in range(2)]

    def tuning(self, **knobs):
        """`with ctx.tuning(walk_waves=17, walk_min_rows=1): ...` -- the knobs are set inside the block and put back
        after it."""
        ctx = self

        class _Scope:
            def __enter__(self):
                self.old = {k: ctx.get_tuning(k) for k in knobs}
                for k, v in knobs.items():
                    ctx.set_tuning(k, v)
                return ctx

            def __exit__(self, *exc):
                for k, v in self.old.items():
                    ctx.set_tuning(k, v)
        return _Scope()


class Group:
    """HBM-resident bit matrix of all columns sharing (kmer_len, num_hash, log_2_filter_len, hash_func)."""

    def __init__(self, ctx: Context, kmer_len: int, num_hash: int, log_2_filter_len: int,
                 column_capacity: int, hash_func: int = 0):
        self.ctx = ctx
        self.params = Params(kmer_len, num_hash, log_2_filter_len, hash_func)
        self._h = C.c_void_p()
        self._nrows = 1 << log_2_filter_len          # rows an add_columns() image must hold
        check(lib().kwage_group_create(ctx._h, C.byref(self.params), column_capacity, C.byref(self._h)))

    @classmethod
    def sparse(cls, ctx: Context, kmer_len: int, num_hash: int, log_2_filter_len: int, column_capacity: int,
               rows: np.ndarray, hash_func: int = 0) -> "Group":
        """A group holding only the listed slices (sorted distinct row indices) of every file added to it
        (kwage_group_create_sparse): for a few queries against a large database."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        g = cls.__new__(cls)
        g.ctx = ctx
        g.params = Params(kmer_len, num_hash, log_2_filter_len, hash_func)
        g._h = C.c_void_p()
        g._nrows = int(rows.size)                     # a sparse group's images hold the listed rows only
        check(lib().kwage_group_create_sparse(ctx._h, C.byref(g.params), column_capacity, rows.ctypes.data, rows.size, C.byref(g._h)))
        return g

    def close(self) -> None:
        if self._h:
            lib().kwage_group_destroy(self._h)
            self._h = C.c_void_p()

    def add_columns(self, rows: np.ndarray, num_filter: int) -> int:
        """rows: uint8 [2^L (a sparse group: its listed rows), >= ceil(num_filter/8)] host image of a file's bit-slice block."""
        assert rows.dtype == np.uint8 and rows.ndim == 2 and rows.strides[1] == 1
        # the library reads exactly this many rows of host_row_stride bytes from the pointer
        assert rows.shape[0] == self._nrows and rows.shape[1] >= (num_filter + 7) // 8, (rows.shape, self._nrows, num_filter)
        first = C.c_uint64()
        check(lib().kwage_group_add_columns(self._h, rows.ctypes.data, rows.strides[0], num_filter, C.byref(first)))
        return first.value

    def add_db_file(self, path: str) -> Tuple[int, int]:
        first, nf = C.c_uint64(), C.c_uint32()
        check(lib().kwage_group_add_db_file(self._h, path.encode(), C.byref(first), C.byref(nf)))
        return first.value, nf.value

    def add_db_files(self, paths: Sequence[str]) -> List[Tuple[int, int]]:
        """Several files at once (columns in the order given): raw files stream through one copy-engine pipeline."""
        n = len(paths)
        arr = (C.c_char_p * n)(*[p.encode() for p in paths])
        first, nf = (C.c_uint64 * n)(), (C.c_uint32 * n)()
        check(lib().kwage_group_add_db_files(self._h, arr, n, first, nf))
        return [(int(first[i]), int(nf[i])) for i in range(n)]

    def add_random_columns(self, num_columns: int, seed: int, density_q8: int) -> int:
        first = C.c_uint64()
        check(lib().kwage_group_add_random_columns(self._h, num_columns, seed, density_q8, C.byref(first)))
        return first.value

    def set_bits(self, rows: np.ndarray, columns: np.ndarray) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        columns = np.ascontiguousarray(columns, dtype=np.uint64)
        assert rows.shape == columns.shape
        check(lib().kwage_group_set_bits(self._h, rows.ctypes.data, columns.ctypes.data, rows.size))

    def read_rows(self, rows: Sequence[int]) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.empty((rows.size, self.row_bytes), dtype=np.uint8)
        check(lib().kwage_group_read_rows(self._h, rows.ctypes.data, rows.size, out.ctypes.data, out.strides[0] if rows.size else self.row_bytes))
        return out

    def finalize(self) -> None:
        check(lib().kwage_group_finalize(self._h))

    num_columns = property(lambda self: lib().kwage_group_num_columns(self._h))
    column_span = property(lambda self: lib().kwage_group_column_span(self._h))
    row_bytes = property(lambda self: lib().kwage_group_row_bytes(self._h))
    row_stride = property(lambda self: lib().kwage_group_row_stride(self._h))
    device_bytes = property(lambda self: lib().kwage_group_device_bytes(self._h))

    @property
    def placement(self) -> dict:
        """How the matrix's device block was chosen (kwage_group_placement): candidates compared and the gather probe's
        GB/s on the block kept / released."""
        n, kept, other, win = C.c_uint32(), C.c_double(), C.c_double(), C.c_double()
        check(lib().kwage_group_placement(self._h, C.byref(n), C.byref(kept), C.byref(other), C.byref(win)))
        return {"candidates": n.value, "kept_probe_gbps": round(kept.value, 1), "other_probe_gbps": round(other.value, 1),
                "kept_windowed_probe_gbps": round(win.value, 1)}

    def stream_read_gbps(self, nbytes: int, iters: int = 3) -> float:
        g = C.c_double()
        check(lib().kwage_stream_read_gbps(self._h, nbytes, iters, C.byref(g)))
        return g.value


class Batch:
    """Query strings resident in HBM (kwage_batch)."""

    def __init__(self, ctx: Context, seqs: Sequence[bytes | str]):
        bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
        offs = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        concat = b"".join(bs)
        self.ctx = ctx
        self.n = len(bs)
        self._h = C.c_void_p()
        check(lib().kwage_batch_create(ctx._h, concat, offs.ctypes.data, self.n, C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib().kwage_batch_destroy(self._h)
            self._h = C.c_void_p()


@dataclass
class SearchResult:
    hits: np.ndarray                 # HIT_DTYPE, sorted by (query, column)
    num_query_kmer: np.ndarray       # uint32 per query
    query_threshold: np.ndarray      # uint32 per query
    total_kmers: int
    bit_tests: int
    algorithmic_bytes: int
    kmer_kernel_ms: float
    search_kernel_ms: float
    search_kernel_launches: int
    search_kernel: str = ""

    def per_query(self) -> List[List[Tuple[int, int]]]:
        out: List[List[Tuple[int, int]]] = [[] for _ in range(len(self.num_query_kmer))]
        for q, c, m in self.hits.tolist():
            out[q].append((c, m))
        return out


ZERO_COPY_HITS = 1 << 20      # longer hit lists are handed to numpy in place (the library's pinned block) instead of copied


class _ResultOwner:
    """Keeps a kwage_result alive for as long as a numpy view of its hit array exists; frees it (the pinned block goes
    back to the context's pool) when the view is collected."""

    def __init__(self, res):
        self.res = res

    def __del__(self):
        try:
            lib().kwage_result_free(self.res)
        except Exception:
            pass


def _unpack_result(res) -> SearchResult:
    free_now = True
    try:
        r = res.contents
        n = r.n_hits
        if n > ZERO_COPY_HITS:
            # 100 M records are 1.2 GB: no second copy (and no page faults) on the Python side either
            raw = (C.c_char * (n * HIT_DTYPE.itemsize)).from_address(C.addressof(r.hits.contents))
            raw._owner = _ResultOwner(res)          # numpy keeps `raw` alive, `raw` keeps the result alive
            free_now = False
            hits = np.frombuffer(raw, dtype=HIT_DTYPE)
        else:
            hits = np.empty(n, dtype=HIT_DTYPE)
            if n:
                C.memmove(hits.ctypes.data, r.hits, n * HIT_DTYPE.itemsize)
        nq = r.n_queries
        nk = np.ctypeslib.as_array(r.num_query_kmer, shape=(nq,)).copy() if nq else np.zeros(0, np.uint32)
        qt = np.ctypeslib.as_array(r.query_threshold, shape=(nq,)).copy() if nq else np.zeros(0, np.uint32)
        return SearchResult(hits, nk, qt, r.total_kmers, r.bit_tests, r.algorithmic_bytes,
                            r.kmer_kernel_ms, r.search_kernel_ms, r.search_kernel_launches,
                            (r.search_kernel or b"").decode())
    finally:
        if free_now:
            lib().kwage_result_free(res)


def search(group: Group, batch: Batch, threshold: float, flags: int = 0) -> SearchResult:
    """kwage_search(): every query of the batch against every column of the group."""
    res = C.POINTER(native.Result)()
    check(lib().kwage_search(group._h, batch._h, C.c_float(threshold), flags, C.byref(res)))
    return _unpack_result(res)


class PendingSearch:
    """A submitted search (kwage_search_submit); collect() waits for it and returns the result."""

    def __init__(self, handle):
        self._h = handle

    def collect(self) -> SearchResult:
        h, self._h = self._h, None
        if h is None:
            raise native.KwageError(-6, "search already collected")
        res = C.POINTER(native.Result)()
        check(lib().kwage_search_collect(h, C.byref(res)))
        return _unpack_result(res)


def submit(group: Group, batch: Batch, threshold: float, flags: int = 0) -> PendingSearch:
    """First half of a search: the device pipeline is enqueued, the call returns at once.  At most two
    searches may be pending per context."""
    h = C.c_void_p()
    check(lib().kwage_search_submit(group._h, batch._h, C.c_float(threshold), flags, C.byref(h)))
    return PendingSearch(h)


Group.search = lambda self, batch, threshold, flags=0: search(self, batch, threshold, flags)
Group.submit = lambda self, batch, threshold, flags=0: submit(self, batch, threshold, flags)


def hash_batch(ctx: Context, kmer_len: int, num_hash: int, log_2_filter_len: int, batch: Batch
               ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """Device k-mer stage alone: per query (distinct canonical k-mers, row indices [n, num_hash])."""
    p = Params(kmer_len, num_hash, log_2_filter_len, 0)
    offs = np.zeros(batch.n + 1, dtype=np.uint64)
    nk = np.zeros(max(batch.n, 1), dtype=np.uint32)
    # first call sizes the outputs (offsets are host-side arithmetic)
    check(lib().kwage_hash_batch(ctx._h, C.byref(p), batch._h, offs.ctypes.data, nk.ctypes.data, None, None))
    total = int(offs[-1])
    kmers = np.zeros(max(total, 1), dtype=np.uint64)
    rows = np.zeros(max(total, 1) * num_hash, dtype=np.uint32)
    check(lib().kwage_hash_batch(ctx._h, C.byref(p), batch._h, offs.ctypes.data, nk.ctypes.data,
                                 kmers.ctypes.data, rows.ctypes.data))
    out_k, out_r = [], []
    for i in range(batch.n):
        o, n = int(offs[i]), int(nk[i])
        out_k.append(kmers[o:o + n].copy())
        out_r.append(rows[o * num_hash:(o + n) * num_hash].reshape(n, num_hash).copy())
    return out_k, out_r


class Database:
    """Several Groups searched as one database -- what a KWAGE database directory is: `.db` files with
    different (kmer_len, num_hash, log_2_filter_len) because maestro picks the Bloom parameters per
    sample size (bloom.cpp:10-68 optimal_bloom_param), the adaptive / COBS-style layout of BASELINE
    config C5.  Hits carry the group index; columns are local to the group."""

    def __init__(self, groups: Sequence[Group]):
        self.groups = list(groups)

    @property
    def num_columns(self) -> int:
        return sum(g.num_columns for g in self.groups)

    @property
    def device_bytes(self) -> int:
        return sum(g.device_bytes for g in self.groups)

    def search(self, batch: Batch, threshold: float, flags: int = 0) -> List[SearchResult]:
        # the k-mer stage is re-run per group: row indices depend on log_2_filter_len (kwage.cpp:411-412)
        return [search(g, batch, threshold, flags) for g in self.groups]

    def close(self) -> None:
        for g in self.groups:
            g.close()


@dataclass
class DatabaseHit:
    query: int
    path: str            # `.db` file the sample lives in
    column: int          # column within that file
    accession: str       # FilterInfo::csv_string() (run accession)
    num_kmers_found: int
    num_query_kmer: int


class FileDatabase(Database):
    """A directory tree / list of `.db` (`.dbz`) files loaded the way the `kwage` CLI loads it: files
    grouped by (kmer_len, num_hash, log_2_filter_len, hash_func), each group one HBM matrix; hits are
    mapped back to (file, column) and carry the sample's run accession."""

    def __init__(self, ctx: Context, paths: Sequence[str]):
        import os
        files: List[str] = []
        todo = list(paths)
        while todo:                                   # breadth first, like FindFiles (file_util.h:30-125)
            p = todo.pop(0)
            if os.path.isdir(p):
                for name in os.listdir(p):
                    full = p + "/" + name
                    if os.path.isdir(full):
                        todo.append(full)
                    elif full.lower().endswith((".db", ".dbz")):
                        files.append(full)
            elif os.path.isfile(p):
                files.append(p)
            else:
                raise native.KwageError(-3, "FindFiles::next: Unable to stat entry " + p)
        by_param = {}
        for f in files:
            h = native.DbHeader()
            check(lib().kwage_db_read_header(f.encode(), C.byref(h)))
            by_param.setdefault((h.kmer_len, h.num_hash, h.log_2_filter_len, h.hash_func), []).append((f, h.num_filter))
        groups, self._layout, self._info = [], [], {}
        for (k, nh, lg, hf), members in sorted(by_param.items()):
            span = 0
            for _, nf in members:
                span = (span + 15) // 16 * 16 + (nf + 7) // 8
            g = Group(ctx, k, nh, lg, span * 8, hf)
            firsts = [(first, nf, f) for (first, nf), (f, _) in zip(g.add_db_files([f for f, _ in members]), members)]
            g.finalize()
            groups.append(g)
            self._layout.append(firsts)
        super().__init__(groups)
        self.ctx = ctx
        self.files = files

    def _accession(self, path: str, column: int) -> str:
        d = self._info.get(path)
        if d is None:
            d = C.c_void_p()
            check(lib().kwage_dbinfo_open(path.encode(), C.byref(d)))
            self._info[path] = d
        buf = C.create_string_buffer(64)
        check(lib().kwage_dbinfo_csv_string(d, column, buf, 64))
        return buf.value.decode()

    def search_sequences(self, seqs: Sequence[bytes | str], threshold: float = 1.0, flags: int = SEARCH_EARLY_EXIT) -> List[DatabaseHit]:
        """What `kwage -d ... <seqs>` reports, as records: sorted by query, then descending hits."""
        import bisect
        b = Batch(self.ctx, seqs)
        out: List[DatabaseHit] = []
        try:
            for g, layout in zip(self.groups, self._layout):
                r = search(g, b, threshold, flags)
                starts = [f[0] for f in layout]
                for q, c, m in r.hits.tolist():
                    i = bisect.bisect_right(starts, c) - 1
                    first, _, path = layout[i]
                    out.append(DatabaseHit(q, path, c - first, self._accession(path, c - first), m, int(r.num_query_kmer[q])))
        finally:
            b.close()
        out.sort(key=lambda h: (h.query, -h.num_kmers_found, h.path, h.column))
        return out

    def close(self) -> None:
        for d in self._info.values():
            lib().kwage_dbinfo_close(d)
        self._info = {}
        super().close()
