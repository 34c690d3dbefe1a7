"""One rank of tests/test_gpu_sharded.py: the HIP engine as the rank-local searcher of a column-sharded database,
W ranks sharing device 0, exchange over gloo.  Every rank checks what it can see; rank 0 checks the merged lists against
the unsharded kwage_search on the same device AND against the CPU oracle.  usage: sharded_worker.py <out_dir>
(RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment)."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rand_seq(rng, n):
    return ACGT[rng.integers(0, 4, size=n)].tobytes().decode()


def as_tuples(a):
    return [tuple(int(x) for x in r) for r in np.asarray(a).reshape(-1, 3)]


def main():
    import time
    t_start = time.perf_counter()

    def lap(what):
        print("[rank %s] %6.1f s  %s" % (os.environ["RANK"], time.perf_counter() - t_start, what), flush=True)
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    import kwage_amd as ka
    import kwage_oracle as oracle
    from kwage_amd.distributed import (HitExchange, PipelinedDeviceSearcher, ShardedSearch, StepPipeline, device_search_fn,
                                       device_tensor_search_fn, global_column_bases, partition_columns, partition_files)
    oracle.build()
    lap("imports done")
    ctx = ka.Context(0)                       # before the process group: hardware queues are first come, first served
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ------------------------------------------------------------------------------------------------------
        # A. one 40 000-column group split at 1024-column boundaries (partition_columns)
        # ------------------------------------------------------------------------------------------------------
        rng = np.random.default_rng(2024)                    # the same matrix and queries on every rank
        k, nh, L, n_cols = 31, 2, 12, 40000
        image = (rng.random((1 << L, n_cols // 8)) < 0.995).astype(np.uint8) * np.uint8(255)
        image &= rng.integers(0, 256, size=image.shape, dtype=np.uint8) | rng.integers(0, 256, size=image.shape, dtype=np.uint8)
        genome = rand_seq(rng, 900)
        planted = [3, 1020, 1023, 1024, 9000] + list(range(2048, 2048 + 30))      # all inside the first blocks: other ranks find nothing at t = 1
        for col in planted:
            for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1):
                image[r, col // 8] |= np.uint8(1 << (col % 8))
        seqs = [genome[:400], rand_seq(rng, 300), genome[300:800].lower(), "ACGT", genome[100:160] + "N" + genome[161:400], rand_seq(rng, 120)]
        parts = partition_columns(n_cols, world)
        s, e = parts[rank]
        assert s % 1024 == 0 and (e % 1024 == 0 or e == n_cols)
        g = ka.Group(ctx, k, nh, L, max(e - s, 8))
        if e > s:
            g.add_columns(np.ascontiguousarray(image[:, s // 8:(e + 7) // 8]), e - s)
        g.finalize()
        batch = ka.Batch(ctx, seqs)

        thresholds = (1.0, 0.7, 0.0001)
        expect = {}
        if rank == 0:
            whole = ka.Group(ctx, k, nh, L, n_cols)
            whole.add_columns(image, n_cols)
            whole.finalize()
            for t in thresholds:
                r = whole.search(batch, t)
                got = [(int(q), int(c), int(m)) for q, c, m in zip(r.hits["query"], r.hits["column"], r.hits["num_match"])]
                exp = []
                for qi, q in enumerate(seqs):
                    hits, _ = oracle.search_image(image, image.shape[1], k, nh, L, n_cols, oracle.unique_kmers(q, k), float(np.float32(t)))
                    exp += [(qi, c, m) for c, m in hits]
                assert got == exp, ("unsharded kwage_search != oracle", t, len(got), len(exp))
                expect[t] = exp
            whole.close()
            assert len(expect[1.0]) and all(c in planted for _, c, _ in expect[1.0])
            assert len(expect[0.0001]) == n_cols * sum(1 for q in seqs if len(q) >= k and "N" not in q[:k] or len(oracle.unique_kmers(q, k)))

        span = int(g.column_span) if e > s else 0
        lap("A: matrices resident, oracle expectations ready")
        # (1) synchronous forms: host lists (device_search_fn) and device-resident lists (device_tensor_search_fn), both exchanges
        for exchange in ("padded", "p2p"):
            for fn_name in ("host", "device"):
                fn = device_search_fn(g, ctx) if fn_name == "host" else device_tensor_search_fn(g, 0, "cuda:0", initial_capacity=32)
                ss = ShardedSearch(dist, rank, world, span, fn, device="cpu", exchange=exchange, capacity=16)
                assert ss.column_base[rank] == s and ss.total_columns == n_cols
                for t in thresholds:
                    merged, nk = ss.search(seqs if fn_name == "host" else batch, t)
                    assert [int(x) for x in np.asarray(nk.cpu() if hasattr(nk, "cpu") else nk)] == [len(oracle.unique_kmers(q, k)) for q in seqs]
                    if rank == 0:
                        assert as_tuples(merged) == expect[t], (exchange, fn_name, t, len(merged), len(expect[t]))
                    else:
                        assert merged is None
        lap("A1: synchronous exchanges done")
        # (2) the pipelined single-group searcher + exchange_counted (round 2's bench path)
        pipe = PipelinedDeviceSearcher(g, 0, "cuda:0", initial_capacity=8)
        ss = ShardedSearch(dist, rank, world, span, None, device="cpu", capacity=8)
        tk = pipe.submit(batch, thresholds[0])
        for i, t in enumerate(thresholds):
            nxt = pipe.submit(batch, thresholds[i + 1]) if i + 1 < len(thresholds) else None
            buf, n = pipe.collect_counted(tk)
            merged = ss.exchange_counted(buf[:max(n, ss.capacity) + 1].cpu(), n)
            if rank == 0:
                assert as_tuples(merged) == expect[t], ("exchange_counted", t)
            tk = nxt
        lap("A2: exchange_counted done")
        # (3) the step pipeline (append mode, global columns from the engine) + the hit-proportional exchange; a buffer of
        # 16 records overflows on the rank(s) that hold the planted columns only at t = 0.7, on every rank at t -> 0
        bases, spans, total = global_column_bases(dist, rank, world, [span])
        assert bases == [s] and total == n_cols
        sp = StepPipeline([g], bases, 0, "cuda:0", initial_capacity=16)
        hx = HitExchange(dist, rank, world, spec=64)
        order = [1.0, 0.7, 0.0001, 1.0, 0.0001, 0.7]
        sp.begin(batch, order[0])
        for i, t in enumerate(order):
            if i + 1 < len(order):
                sp.begin(batch, order[i + 1])
            buf, n = sp.finish()
            local = buf[1:1 + n].cpu().numpy().view(np.uint32)
            assert n == 0 or (int(local[:, 1].min()) >= s and int(local[:, 1].max()) < e), "records carry GLOBAL columns of this rank's block"
            if t == 1.0 and s > max(planted):
                assert n == 0                          # a rank without a hit takes part in the exchange all the same
            merged = hx.exchange_step(buf[:n + 1].cpu(), n)
            if rank == 0:
                assert as_tuples(merged) == expect[t], ("step pipeline", i, t, len(merged), len(expect[t]))
        g.close()
        lap("A3: step pipeline done")

        # ------------------------------------------------------------------------------------------------------
        # B. the golden multi/ database: three parameter groups, whole files dealt to the ranks (partition_files)
        # ------------------------------------------------------------------------------------------------------
        files = sorted(glob.glob(os.path.join(GOLDEN, "multi", "dbs", "**", "*.[dD][bB]"), recursive=True))
        dbs = [oracle.read_db(f) for f in files]
        keys = sorted({(d.header.kmer_len, d.header.num_hash, d.header.log_2_filter_len) for d in dbs})
        assert len(files) == 4 and len(keys) == 3
        queries = [q for f in ("reads.fastq", "contigs.fa.gz") for _, q in oracle.read_sequences(os.path.join(GOLDEN, "multi", f))]
        qb = ka.Batch(ctx, queries)
        mine, layout, local_spans = [], [], []          # this rank's groups; per group [(first local column, file index)]
        for key in keys:
            members = [i for i, d in enumerate(dbs) if (d.header.kmer_len, d.header.num_hash, d.header.log_2_filter_len) == key]
            f0, f1 = partition_files([dbs[i].header.num_filter for i in members], world)[rank]
            share = members[f0:f1]
            if not share:
                local_spans.append(0)
                continue
            cap = sum((dbs[i].header.num_filter + 127) // 128 * 128 for i in share)
            grp = ka.Group(ctx, key[0], key[1], key[2], cap)
            firsts = grp.add_db_files([files[i] for i in share])
            grp.finalize()
            mine.append(grp)
            layout.append([(first, i) for (first, _), i in zip(firsts, share)])
            local_spans.append(int(grp.column_span))
        bases_all, spans_all, total = global_column_bases(dist, rank, world, local_spans)
        bases = [b for b, sp_ in zip(bases_all, local_spans) if sp_]
        hx = HitExchange(dist, rank, world, spec=16)
        for t in (1.0, 0.7, 0.0001):
            buf = torch.zeros((1, 3), dtype=torch.int32)
            n = 0
            if mine:
                sp = StepPipeline(mine, bases, ka.SEARCH_EARLY_EXIT if t == 1.0 else 0, "cuda:0", initial_capacity=8)
                sp.begin(qb, t)
                buf, n = sp.finish()
                buf = buf[:n + 1].cpu()
            # every rank's own records, mapped back to (file, column in file), against the oracle on those files
            rec = buf[1:1 + n].numpy().view(np.uint32)
            got = set()
            for q, c, m in rec.tolist():
                gi = max(i for i, b in enumerate(bases) if b <= c)
                first, fi = max((f, i) for f, i in layout[gi] if f <= c - bases[gi])
                got.add((q, fi, c - bases[gi] - first, m))
            assert len(got) == n
            exp = set()
            for lay in layout:
                for _, fi in lay:
                    d = dbs[fi]
                    h = d.header
                    for qi, q in enumerate(queries):
                        hits, _ = oracle.search_image(d.rows, h.slice_size, h.kmer_len, h.num_hash, h.log_2_filter_len, h.num_filter,
                                                      oracle.unique_kmers(q, h.kmer_len), float(np.float32(t)))
                        exp |= {(qi, fi, c, m) for c, m in hits}
            assert got == exp, ("multi: rank-local records != oracle", t, len(got), len(exp))
            merged = hx.exchange_step(buf, n)
            counts = [None] * world
            dist.all_gather_object(counts, n)
            if rank == 0:
                assert len(merged) == sum(counts) and int(merged[:, 1].max()) < total
                key64 = merged[:, 0] * (1 << 32) + merged[:, 1]
                assert np.all(key64[1:] > key64[:-1])
                # the merged list maps back to exactly the oracle's hits over ALL files
                every = set()
                gbase, at = [], 0
                for gi_, key in enumerate(keys):          # [group][rank] -> base
                    gbase.append([at + sum(spans_all[r][gi_] for r in range(rr)) for rr in range(world)])
                    at += sum(spans_all[r][gi_] for r in range(world))
                exp_all = set()
                for fi, d in enumerate(dbs):
                    h = d.header
                    for qi, q in enumerate(queries):
                        hits, _ = oracle.search_image(d.rows, h.slice_size, h.kmer_len, h.num_hash, h.log_2_filter_len, h.num_filter,
                                                      oracle.unique_kmers(q, h.kmer_len), float(np.float32(t)))
                        exp_all |= {(qi, fi, c, m) for c, m in hits}
                assert len(merged) == len(exp_all), ("multi: merged != oracle over all files", t, len(merged), len(exp_all))
                assert sorted(m for _, _, m in as_tuples(merged)) == sorted(m for _, _, _, m in exp_all)
        lap("B: multi/ done")
        for grp in mine:
            grp.close()
        qb.close()
        batch.close()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()
        ctx.close()


if __name__ == "__main__":
    main()
