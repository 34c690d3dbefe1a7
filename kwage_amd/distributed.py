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


# ------------------------------------------------------------------------------------------------------
# Steps over several groups, one list per step, an exchange proportional to the hits
# ------------------------------------------------------------------------------------------------------
_MIX1, _MIX2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9)


def hits_checksum(records) -> int:
    """Order-independent 64-bit checksum of (query, column, num_match) records: the sum (mod 2^64) of a mixed
    word per record.  A rank computes it over its own list, rank 0 over the merged one; the sums agree iff the
    exchange lost, duplicated or altered nothing (bench.py's exchange_check)."""
    r = np.asarray(records).reshape(-1, 3)
    if r.dtype == np.int32:               # device buffers hold the u32 fields in int32 tensors
        r = r.view(np.uint32)
    r = r.astype(np.uint64)
    if not len(r):
        return 0
    with np.errstate(over="ignore"):
        h = (r[:, 0] * _MIX1 + r[:, 1]) * _MIX2
        h ^= h >> np.uint64(29)
        h = (h + r[:, 2]) * _MIX1
        return int(h.sum(dtype=np.uint64))


def global_column_bases(dist, rank: int, world: int, local_spans: Sequence[int], device: str = "cpu"):
    """Global column numbers for a database of several groups sharded by columns: group after group, inside a group
    rank after rank.  local_spans[g] = column span of this rank's block of group g.
    -> (this rank's base per group, spans[rank][group], total columns).  One all_gather at set-up."""
    import torch
    mine = torch.tensor([int(x) for x in local_spans], dtype=torch.int64, device=device)
    if dist is None or world == 1:
        spans = [[int(x) for x in local_spans]]
    else:
        outs = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(outs, mine)
        spans = [[int(x) for x in o.tolist()] for o in outs]
    bases, at = [], 0
    for g in range(len(local_spans)):
        bases.append(at + sum(spans[r][g] for r in range(rank)))
        at += sum(spans[r][g] for r in range(world))
    if at > 1 << 32:
        raise OverflowError("global_column_bases: %d columns do not fit the 32-bit column field of a hit record" % at)
    return bases, spans, at


class StepPipeline:
    """One rank's software pipeline over STEPS.  A step = one query batch searched against every group this rank holds
    (one group for C2-C4, the eight filter sizes of C5), all of its hits appended to ONE device list through
    kwage_search_device_append_submit: the engine counts into row 0 of the step's buffer (a u64), writes the records
    behind it and adds the group's global column base, so the buffer is ready for the exchange as it stands -- what
    the reference does when every file's matches go to one list per query (kwage.cpp:154-177).

    A context runs two searches at a time (two slots): begin() queues a step's searches, finish() completes the
    oldest step, feeding the slots as they come free -- so the next step's first gather kernels are already running
    while the caller exchanges the finished step's hits.  At most two steps may be open; THREE buffers rotate, so that a
    caller may begin step i+2 right after finish(i) and only then exchange step i's buffer: the exchange's collective is
    a small kernel that waits for wave slots behind the running gather kernel, and with a third buffer the device has
    step i+2's gather kernel queued behind step i+1's while the host waits for it."""

    def __init__(self, groups, column_bases, flags: int = 0, device: str = "cuda", initial_capacity: int = 1 << 18):
        import torch
        assert len(groups) == len(column_bases) and len(groups) >= 1
        self.groups, self.bases, self.flags, self.device = list(groups), [int(b) for b in column_bases], flags, device
        # torch.empty, not zeros: a fill kernel would run on torch's stream, unordered with the engine's streams
        self.bufs = [torch.empty((1 + initial_capacity, 3), dtype=torch.int32, device=device) for _ in range(3)]
        self.steps = []            # open steps, oldest first: dicts
        self.todo = []             # (step, group index) not yet submitted, in order
        self.inflight = []         # (handle, step, group index), oldest first
        self.next_id = 0
        self.last_kernel_ms = 0.0
        self.searches_submitted = 0

    def _submit(self, step, gi):
        import ctypes as C
        from .native import check, lib
        buf = self.bufs[step["buf"]]
        h = C.c_void_p()
        check(lib().kwage_search_device_append_submit(self.groups[gi]._h, step["batch"]._h, C.c_float(step["threshold"]), self.flags,
                                                      buf.data_ptr() + 12, buf.shape[0] - 1, buf.data_ptr(), self.bases[gi],
                                                      1 if gi == 0 else 0, C.byref(h)))
        self.searches_submitted += 1
        return h

    def _collect(self, h):
        import ctypes as C
        from .native import check, lib
        n, ms = C.c_uint64(), C.c_float()
        check(lib().kwage_search_device_collect(h, C.byref(n), None, C.byref(ms)))
        return int(n.value), float(ms.value)

    def _pump(self):
        while self.todo and len(self.inflight) < 2:
            step, gi = self.todo.pop(0)
            self.inflight.append((self._submit(step, gi), step, gi))

    def begin(self, batch, threshold: float):
        if len(self.steps) >= 2:
            raise RuntimeError("StepPipeline: two steps are open already; finish() one first")
        step = {"id": self.next_id, "buf": self.next_id % 3, "batch": batch, "threshold": threshold, "done": 0, "n": 0, "ms": 0.0}
        self.next_id += 1
        self.steps.append(step)
        self.todo += [(step, gi) for gi in range(len(self.groups))]
        self._pump()

    def finish(self):
        """Complete the oldest open step -> (its buffer [1 + capacity, 3] with the count in row 0, n_hits)."""
        import torch
        if not self.steps:
            raise RuntimeError("StepPipeline: no open step")
        step = self.steps[0]
        while step["done"] < len(self.groups):
            h, st, gi = self.inflight.pop(0)
            n, ms = self._collect(h)
            st["done"] += 1
            st["n"] = n                # the running total: the last search of a step leaves the step's total
            st["ms"] += ms
            self._pump()
        self.steps.pop(0)
        if step["n"] > self.bufs[step["buf"]].shape[0] - 1:
            # The list outgrew the buffer (first steps only): let what is in flight for the next step finish (it
            # writes the OTHER buffer), grow this one and redo the step, search by search.
            while self.inflight:
                h, st, gi = self.inflight.pop(0)
                n, ms = self._collect(h)
                st["done"] += 1; st["n"] = n; st["ms"] += ms
            while step["n"] > self.bufs[step["buf"]].shape[0] - 1:
                self.bufs[step["buf"]] = torch.empty((int(step["n"] * 1.25) + 2, 3), dtype=torch.int32, device=self.device)
                step["ms"] = 0.0
                for gi in range(len(self.groups)):
                    n, ms = self._collect(self._submit(step, gi))
                    step["n"] = n; step["ms"] += ms
            self._pump()
        self.last_kernel_ms = step["ms"]
        return self.bufs[step["buf"]], step["n"]

    def drain(self):
        """Finish every open step, discarding the results (error paths)."""
        while self.steps:
            self.finish()


class HitExchange:
    """The one exchange of the sharded search: every rank's hit list (records that already carry GLOBAL column numbers,
    StepPipeline) to rank 0, in bytes proportional to the hits.

      small lists  ONE collective: all_gather of [count | first `spec` records] (48 KB per rank); when every count fits,
                   that is all -- one copy-back to pinned memory, merge on rank 0.
      large lists  what does not fit goes to rank 0 ONLY, in exact sizes, by one grouped send/recv (RCCL has no gatherv);
                   after a step with a large list the first collective carries the counts alone (one 12-byte row per
                   rank) until the lists are small again, so ranks other than 0 receive nothing but counts.

    Every rank takes the same decisions (all of them see all counts).  Nothing here needs device work besides the
    collectives and copy-backs, so it overlaps with the next step's gather kernel.  Works on host tensors with gloo."""

    def __init__(self, dist, rank: int, world: int, spec: int = 4096):
        self.dist, self.rank, self.world, self.spec = dist, rank, world, int(spec)
        self.counts_only = False          # mode of the NEXT step's first collective
        self._recv = self._host = self._pad = None
        self.last_counts = None
        self.stats = {"collectives": 0, "p2p_batches": 0, "bytes_received": 0}

    def _pinned(self, rows, like):
        import torch
        if self._host is None or self._host.shape[0] < rows:
            self._host = torch.empty((max(rows, 1), 3), dtype=torch.int32, pin_memory=like.is_cuda)
        return self._host[:rows]

    def exchange_step(self, buf, n: int):
        """buf: int32 [>= 1 + n, 3], row 0 = u64 record count (as StepPipeline leaves it), rows 1.. the records.
        -> on rank 0 the merged list sorted by (query, column) (int64 [N, 3]); None elsewhere."""
        import torch
        if n > int(buf.shape[0]) - 1:
            raise ValueError("exchange_step: the buffer holds fewer records than its count says")
        W, spec = self.world, (0 if self.counts_only else self.spec)
        rows = 1 + spec
        if int(buf.shape[0]) >= rows:
            send = buf[:rows]
        else:                             # a buffer smaller than the speculative part: pad (rare, tiny buffers)
            if self._pad is None or self._pad.shape[0] != rows or self._pad.device != buf.device:
                self._pad = torch.zeros((rows, 3), dtype=torch.int32, device=buf.device)
            self._pad[:buf.shape[0]] = buf
            send = self._pad
        if self._recv is None or self._recv.shape[0] != W * rows or self._recv.device != buf.device:
            self._recv = torch.empty((W * rows, 3), dtype=torch.int32, device=buf.device)
        if self.dist is not None:
            self.dist.all_gather_into_tensor(self._recv, send)
            self.stats["bytes_received"] += (W - 1) * rows * 12
            self.stats["collectives"] += 1
            gathered = self._recv
        else:
            gathered = send               # no process group at all (one rank): nothing to gather
        if buf.is_cuda:
            host = self._pinned(W * rows, buf)
            host.copy_(gathered, non_blocking=True)
            torch.cuda.current_stream(buf.device).synchronize()
            head = host.numpy().reshape(W, rows, 3)
        else:
            head = gathered.numpy().reshape(W, rows, 3)
        c64 = head[:, 0, :2].astype(np.int64) & 0xFFFFFFFF
        counts = [int(c64[r, 0] | (c64[r, 1] << 32)) for r in range(W)]
        self.last_counts = counts
        rest = [max(0, c - spec) for c in counts]
        parts = [head[r, 1:1 + min(counts[r], spec)].copy() if self.rank == 0 else None for r in range(W)]
        if any(rest):
            tails = {}
            ops = []
            if self.rank == 0:
                for r in range(1, W):
                    if rest[r]:
                        tails[r] = torch.empty((rest[r], 3), dtype=torch.int32, device=buf.device)
                        ops.append(self.dist.P2POp(self.dist.irecv, tails[r], r))
                        self.stats["bytes_received"] += rest[r] * 12
            elif rest[self.rank]:
                ops.append(self.dist.P2POp(self.dist.isend, buf[1 + spec:1 + counts[self.rank]].contiguous(), 0))
            if ops:
                for req in self.dist.batch_isend_irecv(ops):        # ONE grouped ncclSend/ncclRecv launch on RCCL
                    req.wait()
                if buf.is_cuda:
                    # the engine's own streams (not ordered with torch's) refill this buffer two steps on: the send must
                    # have left it before the call returns
                    torch.cuda.current_stream(buf.device).synchronize()
                self.stats["p2p_batches"] += 1
            if self.rank == 0:
                if rest[0]:
                    tails[0] = buf[1 + spec:1 + counts[0]]
                for r, t in tails.items():
                    if t.is_cuda:
                        hp = torch.empty((t.shape[0], 3), dtype=torch.int32, pin_memory=True)
                        hp.copy_(t, non_blocking=True)
                        torch.cuda.current_stream(t.device).synchronize()
                        t = hp
                    parts[r] = np.concatenate([parts[r], t.numpy()]) if len(parts[r]) else t.numpy()
        # the NEXT step's first collective: counts alone after a large list, counts + records after small ones
        self.counts_only = max(counts) > self.spec
        if self.rank != 0:
            return None
        return merge_hits(parts, [0] * W)
