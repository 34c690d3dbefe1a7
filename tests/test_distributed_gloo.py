"""N>1 path rehearsed on CPU: world_size-2 gloo, the real partition + gatherv + merge code of
kwage_amd.distributed with the CPU oracle injected as the rank-local searcher (the HIP engine is
the searcher on GPU boxes).  Checks that column sharding + one variable-length gather reproduces
the unsharded hit list exactly."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_partition_columns():
    from kwage_amd.distributed import partition_columns, COLUMN_ALIGN
    for n in (1, 1000, 1024, 1025, 100_000, 10_000_000):
        for w in (1, 2, 3, 8):
            parts = partition_columns(n, w)
            assert len(parts) == w and parts[0][0] == 0 and parts[-1][1] == n
            for (s0, e0), (s1, e1) in zip(parts, parts[1:]):
                assert e0 == s1 and (s1 % COLUMN_ALIGN == 0 or s1 == n)
            sizes = [e - s for s, e in parts]
            assert max(sizes) - min(sizes) <= 2 * COLUMN_ALIGN or n < COLUMN_ALIGN * w


def test_partition_files():
    from kwage_amd.distributed import partition_files
    nf = [2048] * 10 + [100]
    for w in (1, 2, 4, 8):
        parts = partition_files(nf, w)
        assert parts[0][0] == 0 and parts[-1][1] == len(nf)
        for (s0, e0), (s1, e1) in zip(parts, parts[1:]):
            assert e0 == s1                      # contiguous, no file split or lost
        loads = [sum(nf[s:e]) for s, e in parts]
        assert max(loads) - min(loads) <= 2 * 2048
    assert sum(e - s for s, e in partition_files([5], 4)) == 1


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import kwage_oracle as oracle
    from kwage_amd.distributed import ShardedSearch

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        db = oracle.read_db(os.path.join(GOLDEN, "basic", "db", "basic.db"))
        h = db.header
        # shard the 100 columns at a byte boundary (the engine shards at 1024-column boundaries; a
        # smaller alignment keeps this CPU rehearsal meaningful on a 100-column fixture)
        bounds = [(0, 56), (56, 100)]
        s, e = bounds[rank]
        shard = np.ascontiguousarray(db.rows[:, s // 8:(e + 7) // 8])
        seqs = [q for _, q in oracle.read_sequences(os.path.join(GOLDEN, "basic", "q.fa"))]

        def search_fn(seqs, threshold):
            rows, nk = [], []
            for qi, q in enumerate(seqs):
                kmers = oracle.unique_kmers(q, h.kmer_len)
                nk.append(len(kmers))
                hits, _ = oracle.search_image(shard, shard.shape[1], h.kmer_len, h.num_hash, h.log_2_filter_len,
                                              e - s, kmers, float(np.float32(threshold)))
                rows += [(qi, c, m) for c, m in hits]
            return np.array(rows, dtype=np.int64).reshape(-1, 3), np.array(nk, dtype=np.uint32)

        for exchange, threshold in (("padded", 1.0), ("padded", 0.5), ("padded", 0.0001), ("p2p", 1.0), ("p2p", 0.0001)):
            # capacity 16 forces the grow-and-repeat path of the padded exchange at low thresholds
            ss = ShardedSearch(dist, rank, world, e - s, search_fn, exchange=exchange, capacity=16)
            assert ss.total_columns == 100 and ss.column_base == [b[0] for b in bounds]
            merged, nk = ss.search(seqs, threshold)
            if exchange == "padded" and threshold == 0.0001:
                assert ss.capacity > 16
            if rank == 0:
                exp = []
                for qi, q in enumerate(seqs):
                    kmers = oracle.unique_kmers(q, h.kmer_len)
                    hits, _ = oracle.search_image(db.rows, h.slice_size, h.kmer_len, h.num_hash, h.log_2_filter_len,
                                                  h.num_filter, kmers, float(np.float32(threshold)))
                    exp += [(qi, c, m) for c, m in hits]
                assert merged.tolist() == [list(x) for x in exp], (exchange, threshold)
        # an empty local hit list on one rank must not hang the gather
        ss2 = ShardedSearch(dist, rank, world, e - s,
                            lambda q, t: (np.zeros((0, 3), np.int64) if rank else np.array([[0, 1, 2]]), np.zeros(1, np.uint32)))
        merged, _ = ss2.search(["A"], 1.0)
        if rank == 0:
            assert merged.tolist() == [[0, 1, 2]]
        # exchange_counted: the buffer layout the pipelined device searcher fills (row 0 = u64 count, rows 1.. =
        # records).  Ranks hold buffers of DIFFERENT sizes and the first capacity is too small on rank 1 only.
        import torch
        rng = np.random.default_rng(5 + rank)
        for n_local in ((3, 40), (0, 7), (25, 0)):
            n = n_local[rank]
            rec = np.stack([rng.integers(0, 9, n), rng.permutation(e - s)[:n], rng.integers(1, 99, n)], axis=1).astype(np.int32)
            buf = torch.zeros((1 + (64 if rank == 0 else 41), 3), dtype=torch.int32)
            buf[0, 0] = n
            buf[1:1 + n] = torch.from_numpy(rec)
            ss3 = ShardedSearch(dist, rank, world, e - s, None, capacity=8)
            merged = ss3.exchange_counted(buf, n)
            assert ss3.capacity >= max(n_local) and (max(n_local) <= 8 or ss3.capacity > 8)
            both = [None, None]
            dist.all_gather_object(both, rec.tolist())
            if rank == 0:
                exp = sorted((q, c + bounds[r][0], m) for r in range(world) for q, c, m in both[r])
                assert sorted(map(tuple, merged.tolist())) == exp and merged.tolist() == [list(x) for x in sorted(exp, key=lambda x: (x[0], x[1]))]
            else:
                assert merged is None
        with pytest.raises(ValueError):
            ShardedSearch(dist, rank, world, e - s, None).exchange_counted(torch.zeros((3, 3), dtype=torch.int32), 5)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_search_gloo_world2(tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


class _CountingDist:
    """torch.distributed with a meter: bytes this rank RECEIVES through the calls HitExchange makes."""

    def __init__(self, dist, rank, world):
        self._d, self.rank, self.world = dist, rank, world
        self.P2POp, self.isend, self.irecv = dist.P2POp, dist.isend, dist.irecv
        self.log = []                      # (call, bytes received by this rank)

    def all_gather_into_tensor(self, out, inp):
        self.log.append(("all_gather", (self.world - 1) * inp.numel() * inp.element_size()))
        return self._d.all_gather_into_tensor(out, inp)

    def batch_isend_irecv(self, ops):
        self.log.append(("p2p", sum(op.tensor.numel() * op.tensor.element_size() for op in ops if op.op is self._d.irecv)))
        return self._d.batch_isend_irecv(ops)


def _exchange_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from kwage_amd.distributed import HitExchange, global_column_bases, hits_checksum

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # two groups per rank with ragged spans: global numbering is group after group, rank after rank inside a group
        spans = [1000 + 24 * rank, 3000 - 100 * rank]
        bases, all_spans, total = global_column_bases(dist, rank, world, spans)
        assert all_spans == [[1000 + 24 * r, 3000 - 100 * r] for r in range(world)]
        g0 = sum(s[0] for s in all_spans)
        assert bases == [sum(s[0] for s in all_spans[:rank]), g0 + sum(s[1] for s in all_spans[:rank])] and total == g0 + sum(s[1] for s in all_spans)

        meter = _CountingDist(dist, rank, world)
        hx = HitExchange(meter, rank, world, spec=64)
        rng = np.random.default_rng(100 + rank)

        def make(n, cap=None):
            """a step's buffer as the engine leaves it: row 0 = u64 count, then n records with GLOBAL columns of this rank"""
            q = rng.integers(0, 5000, n)
            c = rank * 10_000_000 + rng.permutation(10_000_000)[:n] if n <= 10_000_000 else None
            rec = np.stack([q, c, rng.integers(1, 970, n)], axis=1).astype(np.int64)
            rec = rec[np.unique(rec[:, 0] * (1 << 32) + rec[:, 1], return_index=True)[1]]      # (query, column) pairs are unique
            n = len(rec)
            buf = torch.zeros((1 + (cap if cap is not None else n), 3), dtype=torch.int32)
            buf[0, 0] = n
            buf[1:1 + n] = torch.from_numpy(rec.astype(np.uint32).view(np.int32))
            return buf, n, rec

        def run(n, cap=None):
            buf, n, rec = make(n, cap)
            merged = hx.exchange_step(buf, n)
            both = [None] * world
            dist.all_gather_object(both, (n, hits_checksum(rec)))
            if rank == 0:
                assert len(merged) == sum(x[0] for x in both)
                assert hits_checksum(merged) == sum(x[1] for x in both) % (1 << 64)
                key = merged[:, 0] * (1 << 32) + merged[:, 1]
                assert np.all(key[1:] > key[:-1])
            else:
                assert merged is None
            return n

        # small lists: one collective, nothing else
        run([5, 0, 64, 17][rank])
        assert [c for c, _ in meter.log] == ["all_gather"] and not hx.counts_only
        # one rank's list outgrows the speculative part: its tail goes to rank 0 alone, in its exact size
        meter.log.clear()
        n = run([3, 200, 0, 9][rank])
        assert [c for c, _ in meter.log][0] == "all_gather" and hx.counts_only
        if rank == 0:
            assert meter.log[1] == ("p2p", (200 - 64) * 12)
        elif rank == 1:
            assert meter.log[1] == ("p2p", 0)                       # it sends, receives nothing
        else:
            assert len(meter.log) == 1                              # ranks with nothing to send do not take part
        # large lists on every rank, twice: from the second step on ranks other than 0 receive nothing but counts
        for step in range(2):
            meter.log.clear()
            n = run(1_000_000, cap=1_000_000 + rank)
            assert hx.counts_only and hx.last_counts[rank] == n
            if rank == 0:
                assert meter.log[0] == ("all_gather", (world - 1) * 12)
                assert meter.log[1] == ("p2p", sum(hx.last_counts[1:]) * 12)
            else:
                assert meter.log == [("all_gather", (world - 1) * 12), ("p2p", 0)], meter.log
        # back to small lists: this step still carries counts only (its records travel by p2p), the next one is padded again
        meter.log.clear()
        run([0, 3, 0, 1][rank])
        assert not hx.counts_only
        meter.log.clear()
        run([2, 0, 1, 0][rank])
        assert meter.log == [("all_gather", (world - 1) * (1 + 64) * 12)]
        # a buffer smaller than the speculative part, and a count that lies
        run(3, cap=5)
        with pytest.raises(ValueError):
            hx.exchange_step(torch.zeros((3, 3), dtype=torch.int32), 5)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_hit_exchange_is_proportional_to_the_hits_gloo_world4(tmp_path):
    """HitExchange (the exchange of bench.py's sharded steps): small lists ride in ONE all_gather; what does not fit goes
    to rank 0 only, in exact sizes; after a large step the first collective carries counts alone, so with 1 M records per
    rank the other ranks receive 36 bytes per step.  Merged list == the ranks' records (count + checksum + order)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_exchange_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(4))
