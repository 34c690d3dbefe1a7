"""Multi-GPU host for the kwage search path: one process per GPU over torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU rehearsal).

The path shards on the SAMPLE (column) axis -- a hit for sample j depends only on column j of
the addressed rows -- so every rank holds all rows of its own contiguous
block of columns, searches it independently (queries are replicated: kilobytes), and the only
exchange is the variable-length gather of hit records to rank 0:

    default ("padded"): ONE collective -- all_gather of fixed-capacity buffers [count | hits...];
             if any rank's count exceeds the capacity every rank sees it, the capacity is doubled
             and the exchange repeated (first steps only)
    "p2p":   counts <- all_gather(one int64 per rank); hits <- one grouped send/recv with exact sizes
             (RCCL has no native gatherv)
    pipelined (bench.py, PipelinedDeviceSearcher + ShardedSearch.exchange_counted): the engine writes
             [u64 count | hits...] itself, so the buffer goes into the all_gather as it stands while the
             next search's gather kernel is already running; merge + sort on the host

Rank 0 concatenates; no merge is needed because column ranges are disjoint.  No row data ever
crosses xGMI.  The reference has no counterpart (its only parallel axis is OpenMP over .db files,
kwage.cpp:76-87); this is what replaces it at node scale.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Sequence, Tuple

import numpy as np

COLUMN_ALIGN = 1024          # shard boundaries fall on 128-byte (1024-column) multiples


def partition_columns(num_columns: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, end) column blocks, one per rank, boundaries multiples of COLUMN_ALIGN
    (except the last end).  Ranks may get an empty block when there are fewer aligned blocks than
    ranks."""
    units = (num_columns + COLUMN_ALIGN - 1) // COLUMN_ALIGN
    out, start_u = [], 0
    for r in range(world):
        n_u = units // world + (1 if r < units % world else 0)
        s, e = start_u * COLUMN_ALIGN, min((start_u + n_u) * COLUMN_ALIGN, num_columns)
        out.append((min(s, num_columns), e))
        start_u += n_u
    return out


def partition_files(num_filters: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Assign whole `.db` files (never split: each is a <=2048-column block written by the
    reference's build_db) to ranks as contiguous [first_file, last_file) ranges with balanced
    column totals (greedy prefix split)."""
    total = int(sum(num_filters))
    owner, prefix = [], 0
    for nf in num_filters:            # a file goes to the rank that owns its middle column
        owner.append(min(world - 1, int((prefix + nf / 2.0) * world / max(total, 1))))
        prefix += nf
    out = []
    for r in range(world):
        idx = [i for i, o in enumerate(owner) if o == r]
        out.append((idx[0], idx[-1] + 1) if idx else ((out[-1][1], out[-1][1]) if out else (0, 0)))
    return out


def gatherv_hits(local_hits, dist, rank: int, world: int, dst: int = 0, device=None):
    """Variable-length gather of [n_i, 3] int32 hit records (query, local column, num_match).

    local_hits: torch tensor on the backend's device (cuda for nccl, cpu for gloo).
    Returns on dst a list of per-rank tensors (exact sizes); elsewhere None."""
    import torch
    dev = local_hits.device if device is None else device
    cnt = torch.tensor([local_hits.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    if rank == dst:
        outs = [torch.empty((c, 3), dtype=torch.int32, device=dev) for c in counts]
        outs[dst] = local_hits
        ops = [dist.P2POp(dist.irecv, outs[r], r) for r in range(world) if r != dst and counts[r] > 0]
    else:
        outs = None
        ops = [dist.P2POp(dist.isend, local_hits.contiguous(), dst)] if counts[rank] > 0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):     # ONE grouped ncclSend/ncclRecv launch on RCCL
            req.wait()
    return outs


def merge_hits(parts: Sequence[np.ndarray], column_base: Sequence[int]) -> np.ndarray:
    """Concatenate per-rank [n,3] (query, local column, num_match) records into global columns,
    sorted by (query, column).  Column ranges are disjoint, so this is a plain concatenation; the order is made by the
    library's host radix sort (kwage_sort_hits: 1.2 M records in tens of ms where numpy's lexsort takes 0.4 s -- more
    than a C3 search step)."""
    rows = []
    for r, p in enumerate(parts):
        p = np.asarray(p).reshape(-1, 3)
        if len(p):
            q = p.astype(np.uint32)           # a copy: the caller's buffer is left alone
            if int(column_base[r]) + int(q[:, 1].max()) >= 1 << 32:
                raise OverflowError("merge_hits: column base %d + local column %d does not fit 32 bits" % (int(column_base[r]), int(q[:, 1].max())))
            q[:, 1] += np.uint32(int(column_base[r]))
            rows.append(q)
    if not rows:
        return np.zeros((0, 3), dtype=np.int64)
    allh = np.ascontiguousarray(np.concatenate(rows))
    from . import native
    native.lib().kwage_sort_hits(allh.ctypes.data, len(allh))
    return allh.astype(np.int64)


def merge_hits_torch(parts, column_base):
    """Device-side merge: records from all ranks -> global columns, sorted by (query, column) with
    one torch.sort of 64-bit keys. parts: list of int32 [n_i, 3] tensors on one device."""
    import torch
    keys, vals = [], []
    for r, p in enumerate(parts):
        if p.shape[0] == 0:
            continue
        q = p[:, 0].to(torch.int64)
        c = p[:, 1].to(torch.int64) + int(column_base[r])
        keys.append((q << 32) | c)
        vals.append(p[:, 2].to(torch.int64))
    if not keys:
        return np.zeros((0, 3), dtype=np.int64)
    key = torch.cat(keys)
    val = torch.cat(vals)
    key, order = torch.sort(key)
    out = torch.stack([key >> 32, key & 0xFFFFFFFF, val[order]], dim=1)
    return out.cpu().numpy()


@dataclass
class ShardedSearch:
    """One rank's view of a column-sharded search.

    search_fn(queries, threshold) -> ([n,3] int array or torch tensor of (query, LOCAL column,
                                      num_match), per-query num_query_kmer array)
    is the rank-local searcher: on a GPU box the HIP engine (device_tensor_search_fn below); tests
    inject a CPU stand-in to rehearse the exchange under gloo."""
    dist: object
    rank: int
    world: int
    local_columns: int                 # column span of this rank's block
    search_fn: Callable
    device: str = "cpu"
    exchange: str = "padded"           # "padded" (one all_gather) or "p2p" (count all_gather + grouped send/recv)
    capacity: int = 4096               # records per rank in the padded exchange; grows on overflow

    def __post_init__(self):
        import torch
        span = torch.tensor([self.local_columns], dtype=torch.int64, device=self.device)
        spans = [torch.zeros_like(span) for _ in range(self.world)]
        self.dist.all_gather(spans, span)
        spans = [int(s.item()) for s in spans]
        self.column_base = [int(sum(spans[:r])) for r in range(self.world)]
        self.total_columns = int(sum(spans))
        self._send = None
        self._crecv = None
        self._chost = None

    def _exchange_padded(self, t):
        """all_gather of [1 + capacity, 3] int32 buffers; row 0 carries the record count."""
        import torch
        n = int(t.shape[0])
        while True:
            cap = self.capacity
            if self._send is None or self._send.shape[0] != cap + 1 or self._send.device != t.device:
                self._send = torch.zeros((cap + 1, 3), dtype=torch.int32, device=t.device)
                self._recv = torch.empty((self.world * (cap + 1), 3), dtype=torch.int32, device=t.device)
            self._send[0, 0] = n
            m = min(n, cap)
            if m:
                self._send[1:1 + m] = t[:m]
            self.dist.all_gather_into_tensor(self._recv, self._send)
            recv = self._recv.view(self.world, cap + 1, 3)
            counts = recv[:, 0, 0].tolist()                       # one small D2H: every rank learns all counts
            if max(counts) <= cap:
                return [recv[r, 1:1 + counts[r]] for r in range(self.world)]
            self.capacity = max(2 * max(counts), 2 * cap)         # same decision on every rank -> no deadlock

    def search(self, queries, threshold: float):
        """`queries` is handed to search_fn unchanged (a list of strings, or a resident Batch for the
        device searchers).  Returns (hits [n,3] with GLOBAL columns sorted by (query, column),
        num_query_kmer) on rank 0, (None, num_query_kmer) elsewhere."""
        import torch
        local, nk = self.search_fn(queries, threshold)
        if isinstance(local, torch.Tensor):
            t = local                      # already on the backend's device (RCCL: stays in HBM)
            if self.device == "cpu" and t.is_cuda:
                t = t.cpu()                # gloo rehearsal of the device searcher
        else:
            t = torch.as_tensor(np.ascontiguousarray(np.asarray(local, dtype=np.int32).reshape(-1, 3)))
            if self.device != "cpu":
                t = t.to(self.device)
        return self.exchange_and_merge(t), nk

    def exchange_counted(self, buf, n: int):
        """Exchange for a buffer that already has the padded layout (row 0 = u64 record count, rows 1.. =
        records; PipelinedDeviceSearcher fills it on the device): one all_gather of the first 1+capacity
        rows, one copy-back to pinned host memory, merge + sort on the host.  No other device work, so it
        overlaps with a running gather kernel.  -> merged global hit list on rank 0, None elsewhere."""
        import torch
        if n > int(buf.shape[0]) - 1:
            raise ValueError("exchange_counted: the buffer holds fewer records than its count says")
        while True:
            cap = self.capacity                            # the same on every rank, whatever the local buffer size
            if int(buf.shape[0]) >= cap + 1:
                send = buf[:cap + 1]
            else:                                          # rare: another rank's count made the capacity outgrow this buffer
                send = torch.zeros((cap + 1, 3), dtype=torch.int32, device=buf.device)
                send[:buf.shape[0]] = buf
            if self._crecv is None or self._crecv.shape[0] != self.world * (cap + 1) or self._crecv.device != buf.device:
                self._crecv = torch.empty((self.world * (cap + 1), 3), dtype=torch.int32, device=buf.device)
                self._chost = torch.empty((self.world * (cap + 1), 3), dtype=torch.int32, pin_memory=buf.is_cuda)
            self.dist.all_gather_into_tensor(self._crecv, send)
            if buf.is_cuda:
                self._chost.copy_(self._crecv, non_blocking=True)
                torch.cuda.current_stream(buf.device).synchronize()
                host = self._chost.numpy().reshape(self.world, cap + 1, 3)
            else:
                host = self._crecv.numpy().reshape(self.world, cap + 1, 3)
            head = host[:, 0, :2].astype(np.int64) & 0xFFFFFFFF
            counts = [int(head[r, 0] | (head[r, 1] << 32)) for r in range(self.world)]
            if max(counts) <= cap:
                break
            self.capacity = max(2 * max(counts), 2 * cap)     # same decision on every rank -> no deadlock
        if self.rank != 0:
            return None
        return merge_hits([host[r, 1:1 + counts[r]] for r in range(self.world)], self.column_base)

    def exchange_and_merge(self, t):
        """The exchange step alone: this rank's [n,3] int32 hit tensor -> merged global hit list on rank 0
        (None elsewhere).  Callers that pipeline searches (bench.py) call it while the next search runs."""
        if self.exchange == "p2p":
            outs = gatherv_hits(t, self.dist, self.rank, self.world, 0, device=t.device)
        else:
            outs = self._exchange_padded(t)
        if self.rank != 0:
            return None
        if outs[0].is_cuda:
            return merge_hits_torch(outs, self.column_base)
        return merge_hits([o.numpy() for o in outs], self.column_base)


def device_search_fn(group, ctx, flags: int = 0):
    """Rank-local searcher backed by the HIP engine (kwage_amd.engine)."""
    from .engine import Batch

    def fn(seqs, threshold):
        b = Batch(ctx, seqs)
        try:
            r = group.search(b, threshold, flags)
        finally:
            b.close()
        h = np.stack([r.hits["query"], r.hits["column"], r.hits["num_match"]], axis=1).astype(np.int64) \
            if len(r.hits) else np.zeros((0, 3), np.int64)
        return h, r.num_query_kmer
    return fn


def device_tensor_search_fn(group, flags: int = 0, device: str = "cuda", initial_capacity: int = 1 << 20):
    """Rank-local searcher that leaves the hit records in HBM (a torch int32 [n,3] tensor written
    directly by kwage_search_device), so the RCCL gatherv sends them without a host round trip.
    `queries` must be a resident kwage_amd.Batch."""
    import ctypes as C
    import torch
    from .native import check, lib

    state = {"buf": torch.empty((initial_capacity, 3), dtype=torch.int32, device=device)}

    def fn(batch, threshold):
        n = C.c_uint64()
        nk = torch.empty((max(batch.n, 1),), dtype=torch.int32, device=device)
        while True:
            buf = state["buf"]
            check(lib().kwage_search_device(group._h, batch._h, C.c_float(threshold), flags,
                                            buf.data_ptr(), buf.shape[0], C.byref(n), nk.data_ptr()))
            if n.value <= buf.shape[0]:
                break
            state["buf"] = torch.empty((int(n.value * 1.25) + 1, 3), dtype=torch.int32, device=device)
        return state["buf"][: n.value], nk[: batch.n]
    return fn


class PipelinedDeviceSearcher:
    """submit()/collect() over kwage_search_device_submit/_collect with two alternating exchange buffers:
    while step i's hits are exchanged over RCCL, step i+1's gather kernel is already running.

    Each buffer is an int32 [1 + capacity, 3] tensor: row 0 holds the u64 record count (written by the
    engine itself in stream order), rows 1.. the (query, local column, num_match) records -- the layout
    ShardedSearch.exchange_counted() all-gathers as it stands.  Small torch kernels queued behind a
    running gather kernel each wait 0.1-0.5 ms for wave slots (tools/concurrency_probe.py), so the
    exchange must not need any device work besides the collective and one copy-back."""

    def __init__(self, group, flags: int = 0, device: str = "cuda", initial_capacity: int = 1 << 18):
        import torch
        self.group, self.flags, self.device = group, flags, device
        # torch.empty, not zeros: a fill kernel would run on torch's stream, unordered with the engine's own
        # streams that write the count word and the records (it could land AFTER them and wipe a result)
        self.bufs = [torch.empty((1 + initial_capacity, 3), dtype=torch.int32, device=device) for _ in range(2)]
        self.turn = 0
        self.last_kernel_ms = 0.0          # HIP-event time of the last collected search (SEARCH_TIMING flag)

    def _submit_into(self, buf, batch, threshold):
        import ctypes as C
        from .native import check, lib
        h = C.c_void_p()
        check(lib().kwage_search_device_submit(self.group._h, batch._h, C.c_float(threshold), self.flags,
                                               buf.data_ptr() + 12, buf.shape[0] - 1, buf.data_ptr(), C.byref(h)))
        return h

    def submit(self, batch, threshold):
        i = self.turn
        self.turn ^= 1
        return (self._submit_into(self.bufs[i], batch, threshold), i, batch, threshold)

    def collect_counted(self, ticket):
        """-> (exchange buffer [1 + capacity, 3] with its count in row 0, n_hits)."""
        import ctypes as C
        import torch
        from .native import check, lib
        h, i, batch, threshold = ticket
        n, ms = C.c_uint64(), C.c_float()
        check(lib().kwage_search_device_collect(h, C.byref(n), None, C.byref(ms)))
        while n.value > self.bufs[i].shape[0] - 1:     # rare: grow this buffer and redo the search
            self.bufs[i] = torch.empty((int(n.value * 1.25) + 2, 3), dtype=torch.int32, device=self.device)
            check(lib().kwage_search_device_collect(self._submit_into(self.bufs[i], batch, threshold), C.byref(n), None, C.byref(ms)))
        self.last_kernel_ms = float(ms.value)
        return self.bufs[i], int(n.value)

    def collect(self, ticket):
        buf, n = self.collect_counted(ticket)
        return buf[1:1 + n]
