"""Randomised GPU-vs-oracle sweep: random column counts (1 ... 300k), filter sizes, k, hash counts,
thresholds, batch compositions (empty / duplicate / very long queries -> 14-, 20- and 32-plane counters),
early exit on/off and forced segment counts.  Every case downloads the device-resident bit matrix and
reduces it with the CPU oracle; hit lists must be identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _seq(rng, n):
    return ACGT[rng.integers(0, 4, size=n)].tobytes().decode()


# (24, 36, 45: long queries at t < 1 against wide rows -- the truncated count walk under the list knobs below)
@pytest.mark.parametrize("seed", sorted(set(range(int(os.environ.get("KWAGE_SWEEP_SEEDS", "24")))) | {24, 36, 45}))
def test_random_configuration(oracle, seed, monkeypatch):
    import kwage_amd as ka
    rng = np.random.default_rng(9000 + seed)
    k = int(rng.choice([5, 11, 16, 21, 27, 31, 32]))
    nh = int(rng.integers(1, 6))
    L = int(rng.integers(6, 15))
    n_cols = int(rng.choice([1, 3, 8, 100, 1000, 8191, 8192, 8193, 20000, 100000, 300000]))
    if n_cols >= 100000:
        L = min(L, 10)
    dens = int(rng.choice([64, 128, 200, 240]))
    genome = _seq(rng, int(rng.choice([200, 3000, 40000])))
    lens = [0, k - 1, k, k + 1, 50, 150, 1000] + ([20000] if seed % 3 == 0 else []) + ([70000] if seed % 8 == 0 else [])
    seqs = []
    for n in lens:
        kind = rng.integers(3)
        if kind == 0 and len(genome) >= n:
            a = int(rng.integers(0, len(genome) - n + 1)); s = genome[a:a + n]
        elif kind == 1:
            s = _seq(rng, n)
        else:
            s = (genome[: max(n // 2, 0)] + "N" + _seq(rng, n))[:n]
        seqs.append(s.lower() if rng.random() < 0.2 else s)
    seqs.append(seqs[-1])                       # duplicate query
    with ka.Context(0) as ctx:
        g = ka.Group(ctx, k, nh, L, n_cols)
        g.add_random_columns(n_cols, 100 + seed, dens)
        gb = ka.Batch(ctx, [genome])
        _, rows = ka.hash_batch(ctx, k, nh, L, gb)
        gb.close()
        planted = sorted(set(int(c) for c in rng.integers(0, n_cols, size=3)))
        if rows[0].size:
            for c in planted:
                r = rows[0].reshape(-1)
                g.set_bits(r, np.full(r.shape, c, dtype=np.uint64))
        g.finalize()
        image = g.read_rows(np.arange(1 << L))
        b = ka.Batch(ctx, seqs)
        thresholds = [1.0, float(rng.choice([0.99, 0.9, 0.75, 0.5])), float(rng.choice([0.3, 0.05, 0.0001])) if n_cols <= 20000 else 0.6]
        for thr in thresholds:
            thr32 = float(np.float32(thr))
            exp = []
            for s in seqs:
                kmers = oracle.unique_kmers(s, k)
                e, _ = oracle.search_image(image, image.shape[1], k, nh, L, n_cols, kmers, thr32)
                exp.append((len(kmers), e))
            for flags, force in ((0, 0), (ka.SEARCH_EARLY_EXIT, 0), (0, int(rng.choice([2, 5, 33, 300]))), (ka.SEARCH_EARLY_EXIT, 3)):
                ctx.set_tuning("force_segs", force)
                r = g.search(b, thr, flags)
                per_q = r.per_query()
                for i, (nk, e) in enumerate(exp):
                    assert r.num_query_kmer[i] == nk, (seed, thr, i)
                    assert per_q[i] == e, (seed, k, nh, L, n_cols, thr, flags, force, i, len(per_q[i]), len(e))
            ctx.set_tuning("force_segs", 0)
            if n_cols >= 8192:
                # early exit on rows of a KiB and more: screen + refine, with short segments, every tile handed over, full lists
                for knobs in (dict(refine_seg_rows=8, refine_min_rows=1, refine_max_groups=16), dict(refine_list_cap=3, refine_min_rows=1), dict(refine_unroll=16, refine_max_groups=1)):
                    with ctx.tuning(count_screen_min_tiles=1, **knobs):
                        r = g.search(b, thr, ka.SEARCH_EARLY_EXIT)
                        # (at t < 1 the screen form takes queries of up to 16383 positions: 14 counter planes; longer ones keep the tiled kernel)
                        long_queries = max(len(s) for s in seqs) - k + 1 > 16383
                        # (... or, where the first k-mers the bound needs leave enough of the lists out, the truncated count walk)
                        want = ("and_screen_kernel<",) if thr == 1.0 else (("count_kernel<", "count_walk_kernel<") if long_queries else ("count_screen_kernel<",))
                        assert r.search_kernel.startswith(want) and (("trunc" in r.search_kernel) == r.search_kernel.startswith("count_walk_kernel<")), r.search_kernel
                        assert [(int(n), e) for n, e in zip(r.num_query_kmer, r.per_query())] == exp, (seed, n_cols, knobs)
            if thr < 1.0 and n_cols > 256:
                # the persistent form of the count path (normally for batches that give every wave of the chip a few dozen
                # rows): shares of a few positions / hundreds / more waves than the chip holds, long queries spread over
                # dozens of waves each; twice per case (the kernel must leave its pair counters zero)
                for waves in (0, int(rng.choice([3, 11, 64])), int(rng.choice([700, 2048, 9000]))):
                    with ctx.tuning(count_walk_min_rows=1, count_walk_waves=waves, narrow=0):
                        for _ in range(2):
                            r = g.search(b, thr, 0)
                            assert r.search_kernel.startswith("count_walk_kernel<"), r.search_kernel
                            assert [(int(n), e) for n, e in zip(r.num_query_kmer, r.per_query())] == exp, (seed, n_cols, thr, waves)
            if thr == 1.0 and n_cols > 16384:
                # the walk form of the AND kernel (normally for batches of >= 256k rows and rows of 3..16 KiB), with the
                # batch's positions cut into few / many / very many wave shares: the long queries are then finished
                # through the cut-pair slots by dozens of waves each
                for flags, waves in ((0, 0), (0, 7), (0, 1000), (0, 4096)):
                    with ctx.tuning(walk_min_rows=1, walk_max_kib=64, walk=4, walk_waves=waves):
                        for _ in range(2):         # twice: the kernel must leave its cut-pair slots clean
                            r = g.search(b, thr, flags)
                            assert r.search_kernel.startswith("and_walk_kernel<")
                            assert [(int(n), e) for n, e in zip(r.num_query_kmer, r.per_query())] == exp, (seed, n_cols, waves)
        b.close()
        g.close()
