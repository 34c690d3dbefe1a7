"""Synthetic workloads for the kwage search path (SURVEY.md section 8d / BASELINE.md section 5):
a random Bernoulli(p) bit matrix generated ON the device, `planted` genomes inserted into known
columns (so every configuration has known true positives), and queries that are substrings of
planted genomes mixed with random sequences.  Used by bench.py and the full-size GPU tests.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

from .engine import Batch, Context, Group, hash_batch


@dataclass
class Workload:
    name: str
    num_samples: int          # columns per GPU
    log_2_filter_len: int
    kmer_len: int
    num_hash: int
    num_queries: int
    query_len: int
    threshold: float
    density_q8: int = 64      # per-bit density 64/256 = 0.25 (the reference's per-k-mer FP design point, 1 hash)
    num_genomes: int = 8
    genome_len: int = 4000
    columns_per_genome: int = 3
    hit_fraction: float = 0.5


WORKLOADS: Dict[str, Workload] = {
    # BASELINE.json configs[0]: plumbing
    "c1": Workload("C1: 1k samples x 2^20 bits, 1 hash, one 10 kb query", 1000, 20, 31, 1, 1, 10000, 1.0,
                   genome_len=20000),
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "c2": Workload("C2: 100k samples x 2^23-bit filters, 1k x 1 kb queries, 1 hash, t=1.0", 100_000, 23, 31, 1,
                   1000, 1000, 1.0, num_genomes=32, genome_len=50_000),
    # C2 with rows of exactly 12 KiB (98 304 samples): no ragged last KiB-step in the walk kernels (tools: what does the 212-byte tail of C2's 12 500-byte rows cost?)
    "c2e": Workload("C2 with 98 304 samples (12 KiB rows)", 98_304, 23, 31, 1, 1000, 1000, 1.0, num_genomes=32, genome_len=50_000),
    "c2et": Workload("C2 with 98 304 samples (12 KiB rows), t=0.8", 98_304, 23, 31, 1, 1000, 1000, 0.8, num_genomes=32, genome_len=50_000),
    # C2 through the count path (threshold < 1, one hash): fewest loads in flight per wave
    "c2t": Workload("C2 at t=0.8 (count path, 1 hash): 100k samples x 2^23-bit filters, 1k x 1 kb queries", 100_000, 23, 31, 1,
                    1000, 1000, 0.8, num_genomes=32, genome_len=50_000),
    # few LONG queries against C2's matrix (the early-exit path's other extreme: 200 x 5 kb)
    "c2q5k": Workload("200 x 5 kb queries against 100k samples x 2^23-bit filters, 1 hash, t=1.0", 100_000, 23, 31, 1, 200, 5000, 1.0,
                      num_genomes=32, genome_len=50_000),
    # ... and at t = 0.8: the early-exit path's truncated count walk (few long queries, the count path)
    "c2q5kt": Workload("200 x 5 kb queries against 100k samples x 2^23-bit filters, 1 hash, t=0.8", 100_000, 23, 31, 1, 200, 5000, 0.8,
                       num_genomes=32, genome_len=50_000),
    "c2q100kt": Workload("10 x 100 kb queries against 100k samples x 2^23-bit filters, 1 hash, t=0.8", 100_000, 23, 31, 1, 10, 100_000, 0.8,
                         num_genomes=8, genome_len=400_000),
    # BASELINE.json configs[2]
    "c3": Workload("C3: 1M samples x 2^20-bit filters, 100k x 150 bp queries, 1 hash, t=1.0", 1_000_000, 20, 31, 1,
                   100_000, 150, 1.0, num_genomes=64, genome_len=150_000),
    # BASELINE.json configs[3] (C4), the share of ONE GPU: 10M samples column-sharded over 8 GPUs
    "c4": Workload("C4 per-GPU share: 1.25M samples x 2^20-bit filters, 10k x 1 kb queries, 1 hash, t=1.0", 1_250_000, 20, 31, 1,
                   10_000, 1000, 1.0, num_genomes=64, genome_len=100_000),
    # count path (threshold < 1), 5 hashes: the C5 flavour on one filter size
    "c5s": Workload("C5-single-group: 200k samples x 2^22-bit filters, 5 hashes, 1k x 1 kb queries, t=0.8", 200_000, 22,
                    31, 5, 1000, 1000, 0.8, density_q8=194, num_genomes=32, genome_len=50_000),
    # BASELINE.json configs[4], per-GPU share; the groups are C5_GROUPS (bench.py --workload c5)
    "c5": Workload("C5: adaptive 2^18-2^25-bit filter groups (1.09M samples/GPU), 5 hashes, 10k x 1 kb queries, t=0.8", 0, 0, 31, 5,
                   10_000, 1000, 0.8, density_q8=194, num_genomes=64, genome_len=100_000),
    # the C5 code path (several groups searched back to back) on toy groups: C5_TEST_GROUPS (tests of bench.py)
    "c5tiny": Workload("C5 code path on toy groups 2^10-2^13", 0, 0, 31, 5, 64, 300, 0.8, density_q8=194, num_genomes=4, genome_len=1000),
    # one reference file (2048 columns, 256-byte rows): the several-queries-per-wave kernels (tools/tune_knob.py)
    "narrow": Workload("one 2048-column file x 2^25 rows, 10k x 1 kb queries", 2048, 25, 31, 1, 10_000, 1000, 1.0, num_genomes=16, genome_len=200_000),
    "narrowt": Workload("one 2048-column file x 2^25 rows, 10k x 1 kb queries, t=0.8", 2048, 25, 31, 1, 10_000, 1000, 0.8, num_genomes=16, genome_len=200_000),
    # ONE long query at threshold < 1 against C2's matrix: the count path's persistent kernel with every pair cut across waves
    "long1t": Workload("one 100 kb query against 100k samples x 2^23-bit filters, 1 hash, t=0.9", 100_000, 23, 31, 1, 1, 100_000, 0.9,
                       num_genomes=2, genome_len=200_000, hit_fraction=1.0),
    # small shapes for tests
    "tiny": Workload("tiny", 5000, 14, 31, 2, 64, 300, 1.0, density_q8=128, num_genomes=4, genome_len=1000),
}


# tuning tools: KWAGE_WORKLOAD_CUSTOM="samples,log2 filter len,hashes,queries,query length,threshold[,hit fraction[,columns per planted genome]]" -> WORKLOADS["custom"]
import os as _os
if _os.environ.get("KWAGE_WORKLOAD_CUSTOM"):
    _v = _os.environ["KWAGE_WORKLOAD_CUSTOM"].split(",")
    WORKLOADS["custom"] = Workload("custom: %s samples x 2^%s, %s hash(es), %s x %s bp, t=%s" % tuple(_v[:6]), int(_v[0]), int(_v[1]), 31, int(_v[2]), int(_v[3]), int(_v[4]),
                                   float(_v[5]), density_q8=64 if int(_v[2]) == 1 else 194, num_genomes=64, genome_len=max(150_000, 4 * int(_v[4])),
                                   hit_fraction=float(_v[6]) if len(_v) > 6 else 0.5, columns_per_genome=int(_v[7]) if len(_v) > 7 else 3)

# BASELINE.json configs[4] (C5), the share of ONE GPU: adaptive filter sizes 2^18..2^25, more samples in the
# small filters (as optimal_bloom_param would assign them), 5 hash functions, threshold 0.8.
# sum_g N(g) * 2^g / 8 = 189 GB per GPU; 1.09 M samples per GPU (8.7 M on 8 GPUs).
C5_GROUPS = [(18, 400_000), (19, 300_000), (20, 200_000), (21, 100_000), (22, 50_000), (23, 25_000), (24, 12_500), (25, 6_000)]
C5_TEST_GROUPS = [(10, 3000), (11, 2000), (12, 1000), (13, 500)]


@dataclass
class Synth:
    workload: Workload
    group: Group
    genomes: List[str]
    planted: List[List[int]]                 # per genome: columns holding it
    queries: List[str]
    query_genome: List[int]                  # per query: source genome or -1 (random)
    batch: Batch = field(default=None)


_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _rand_seq(rng: np.random.Generator, n: int) -> str:
    return _ACGT[rng.integers(0, 4, size=n)].tobytes().decode("ascii")


def build(ctx: Context, w: Workload, seed: int = 1, column_seed: int = 0) -> Synth:
    """Create the device-resident database + the query batch for workload `w`.
    `seed` fixes genomes/queries (identical on every rank); `column_seed` varies the random
    columns per rank (each rank holds a different block of samples)."""
    rng = np.random.default_rng(seed)
    g = Group(ctx, w.kmer_len, w.num_hash, w.log_2_filter_len, w.num_samples)
    g.add_random_columns(w.num_samples, seed * 1_000_003 + column_seed, w.density_q8)

    genomes = [_rand_seq(rng, w.genome_len) for _ in range(w.num_genomes)]
    # rows addressed by each genome's k-mers: computed by the device k-mer stage itself
    gb = Batch(ctx, genomes)
    _, rows = hash_batch(ctx, w.kmer_len, w.num_hash, w.log_2_filter_len, gb)
    gb.close()
    crng = np.random.default_rng(seed * 7919 + column_seed)
    planted: List[List[int]] = []
    for gi in range(w.num_genomes):
        cols = sorted(int(c) for c in crng.choice(w.num_samples, size=min(w.columns_per_genome, w.num_samples), replace=False))
        planted.append(cols)
        r = rows[gi].reshape(-1)
        for c in cols:
            g.set_bits(r, np.full(r.shape, c, dtype=np.uint64))
    g.finalize()

    # Hit queries are DISTINCT windows of the planted genomes (no two share a k-mer by construction),
    # so the benchmark gets no cross-query row reuse out of the caches: every addressed row is a fresh
    # random row of the matrix, as the metric's algorithmic-byte count assumes.
    is_hit = rng.random(w.num_queries) < w.hit_fraction
    win_per_genome = w.genome_len // w.query_len if w.genome_len >= w.query_len else 0
    total_windows = win_per_genome * w.num_genomes
    n_hit = int(is_hit.sum())
    if total_windows == 0:
        is_hit[:] = False
        windows = np.zeros(0, dtype=np.int64)
    elif n_hit <= total_windows:
        windows = rng.choice(total_windows, size=n_hit, replace=False)
    else:
        windows = rng.integers(0, total_windows, size=n_hit)      # pool too small: some reuse (tiny shapes only)
    queries, qsrc = [], []
    wi = 0
    for qi in range(w.num_queries):
        if is_hit[qi]:
            gi, wn = divmod(int(windows[wi]), win_per_genome)
            wi += 1
            queries.append(genomes[gi][wn * w.query_len:(wn + 1) * w.query_len])
            qsrc.append(gi)
        else:
            queries.append(_rand_seq(rng, w.query_len))
            qsrc.append(-1)
    s = Synth(w, g, genomes, planted, queries, qsrc)
    s.batch = Batch(ctx, queries)
    return s


def build_multi(ctx: Context, groups: Sequence[Tuple[int, int]], base: Workload, seed: int = 1, column_seed: int = 0):
    """One Synth per (log_2_filter_len, num_samples) group, all sharing genomes and queries (same seed)."""
    from dataclasses import replace
    out = []
    for gi, (lg, ns) in enumerate(groups):
        w = replace(base, name="%s/L%d" % (base.name, lg), log_2_filter_len=lg, num_samples=ns)
        out.append(build(ctx, w, seed=seed, column_seed=column_seed * 1000 + gi))
    return out
