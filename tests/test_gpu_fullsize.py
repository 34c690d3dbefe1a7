"""Full-size GPU checks of every BASELINE.json configuration that fits one GPU -- C2 (100k samples x 2^23-bit
filters, 105 GB resident), C3 (1 M samples x 2^20, 100k x 150 bp reads, 131 GB), the per-GPU share of C4
(1.25 M samples x 2^20, 10k x 1 kb, 164 GB) and the per-GPU share of C5 (eight filter-size groups 2^18..2^25,
5 hashes, t = 0.8, 189 GB) -- through size-independent properties, because no CPU can hold or scan the matrix:

  * planted positives: every query cut from a planted genome reports (at least) the columns the
    genome was planted in, with num_match == num_query_kmer;
  * sampled bit-exactness: for a sample of queries the ADDRESSED rows are copied back from HBM
    and reduced by the CPU oracle -- the hit lists (false positives included) must be identical;
  * idempotence / early-exit invariance: repeated and early-exit searches return the same list.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ka():
    import kwage_amd
    return kwage_amd


def _check_sampled(ka, oracle, s, res, sample, threshold):
    w = s.workload
    per_q = res.per_query()
    for qi in sample:
        kmers = oracle.unique_kmers(s.queries[qi], w.kmer_len)
        assert res.num_query_kmer[qi] == len(kmers)
        rows = oracle.row_indices(kmers, w.kmer_len, w.num_hash, w.log_2_filter_len).reshape(-1)
        matrix = s.group.read_rows(rows)                      # only the rows this query addresses
        exp = oracle.search_row_matrix(matrix, w.num_hash, w.num_samples, len(kmers), float(np.float32(threshold)))
        assert per_q[qi] == exp, (qi, len(per_q[qi]), len(exp))


def _check_planted(s, res, complete_match=True):
    """Every query cut from a planted genome reports the genome's columns with every k-mer found; at threshold 1
    nothing else can be reported with fewer."""
    per_q = res.per_query()
    n_planted_q = 0
    for qi, gi in enumerate(s.query_genome):
        if gi < 0:
            continue
        n_planted_q += 1
        found = dict(per_q[qi])
        assert set(s.planted[gi]) <= set(found), qi
        assert all(found[c] == res.num_query_kmer[qi] for c in s.planted[gi])
        if complete_match:
            assert all(m == res.num_query_kmer[qi] for m in found.values())
        else:
            assert all(res.query_threshold[qi] <= m <= res.num_query_kmer[qi] for m in found.values())
    assert n_planted_q > 0


def _sample_queries(s, n_hit, n_miss):
    hitq = [i for i, g in enumerate(s.query_genome) if g >= 0]
    missq = [i for i, g in enumerate(s.query_genome) if g < 0]
    pick = lambda xs, n: [xs[(len(xs) - 1) * j // max(n - 1, 1)] for j in range(min(n, len(xs)))]     # first ... last
    return pick(hitq, n_hit) + pick(missq, n_miss)


@pytest.mark.timeout(900)
def test_c2_full_size_properties(ka, oracle):
    from kwage_amd import synth
    w = synth.WORKLOADS["c2"]
    with ka.Context(0) as ctx:
        free, total = ctx.mem_info()
        if free < 120e9:
            pytest.skip("needs ~106 GB of free HBM")
        # the library's default placement policy: two candidate blocks for the 105 GB matrix, the gather pattern timed on
        # both, the faster one kept (off for the rest of the suite: conftest.py)
        ctx.set_tuning("group_placement_probe", 1)
        s = synth.build(ctx, w)
        assert s.group.device_bytes == (1 << 23) * 12544
        pl = s.group.placement
        assert pl["candidates"] == 2 and pl["kept_probe_gbps"] >= pl["other_probe_gbps"] > 1000 and pl["kept_windowed_probe_gbps"] > 1000, pl
        r1 = s.group.search(s.batch, 1.0)
        assert r1.total_kmers == 970 * 1000 and r1.algorithmic_bytes == 970 * 1000 * 12500
        # whichever form the loader's probe chose for this block, the other one must report the same columns
        with ctx.tuning(walk_bands=0 if r1.search_kernel.startswith("and_band_walk") else 3, walk_bands_min_gib=0):
            r1b = s.group.search(s.batch, 1.0)
            assert r1b.search_kernel != r1.search_kernel and np.array_equal(r1.hits, r1b.hits), (r1.search_kernel, r1b.search_kernel)
        _check_planted(s, r1)
        hitq = [i for i, g in enumerate(s.query_genome) if g >= 0][:3]
        missq = [i for i, g in enumerate(s.query_genome) if g < 0][:3]
        _check_sampled(ka, oracle, s, r1, hitq + missq, 1.0)
        r2 = s.group.search(s.batch, 1.0, ka.SEARCH_EARLY_EXIT)
        assert np.array_equal(r1.hits, r2.hits)                # kwage.cpp:437-483 never changes results
        r3 = s.group.search(s.batch, 1.0)
        assert np.array_equal(r1.hits, r3.hits)                # idempotent
        # count path on the same resident matrix
        r4 = s.group.search(s.batch, 0.3)
        assert (r4.query_threshold == np.array([oracle.query_threshold(float(np.float32(0.3)), int(n)) for n in r4.num_query_kmer])).all()
        _check_sampled(ka, oracle, s, r4, hitq[:2] + missq[:2], 0.3)
        # a hit-heavy call on the same matrix: at a threshold that truncates to 0 every sample matches every query
        # (kwage.cpp:388,497) -- 2 M records, ordered on the device, fetched in pieces; order in closed form, counts
        # of the first and last query against the oracle on the rows they address
        few = ka.Batch(ctx, s.queries[:20])
        r5 = s.group.search(few, 0.0001)
        assert len(r5.hits) == 20 * w.num_samples
        assert np.array_equal(r5.hits["query"], np.repeat(np.arange(20, dtype=np.uint32), w.num_samples))
        assert np.array_equal(r5.hits["column"], np.tile(np.arange(w.num_samples, dtype=np.uint32), 20))
        _check_sampled(ka, oracle, s, r5, [0, 19], 0.0001)
        few.close()
        s.batch.close()
        s.group.close()


@pytest.mark.timeout(900)
def test_c3_full_size_properties(ka, oracle):
    """BASELINE.json configs[2]: 1 M samples x 2^20-bit filters (125 KB rows, 131 GB), 100 k x 150 bp reads, t = 1.0 --
    one launch of 3.05 M workgroups, more queries than a grid's y dimension holds."""
    from kwage_amd import synth
    w = synth.WORKLOADS["c3"]
    with ka.Context(0) as ctx:
        free, _ = ctx.mem_info()
        if free < 150e9:
            pytest.skip("needs ~135 GB of free HBM")
        s = synth.build(ctx, w)
        assert s.group.device_bytes == (1 << 20) * 125056 and len(s.queries) == 100_000
        r1 = s.group.search(s.batch, 1.0, ka.SEARCH_TIMING)
        assert r1.total_kmers == 120 * 100_000 and r1.algorithmic_bytes == 120 * 100_000 * 125_000
        assert r1.search_kernel.startswith("and_kernel<")
        _check_planted(s, r1)
        _check_sampled(ka, oracle, s, r1, _sample_queries(s, 4, 4), 1.0)
        r2 = s.group.search(s.batch, 1.0, ka.SEARCH_EARLY_EXIT)
        assert np.array_equal(r1.hits, r2.hits)                # kwage.cpp:437-483 never changes results
        r3 = s.group.search(s.batch, 1.0)
        assert np.array_equal(r1.hits, r3.hits)                # idempotent
        s.batch.close()
        s.group.close()


@pytest.mark.timeout(900)
def test_c4_per_gpu_share_full_size_properties(ka, oracle):
    """BASELINE.json configs[3], what ONE of its 8 GPUs holds: 1.25 M samples x 2^20-bit filters (156 KB rows, 164 GB),
    10 k x 1 kb queries, t = 1.0.  (The 8-GPU exchange itself is covered by tests/test_distributed_gloo.py and
    tests/test_bench_launcher.py.)"""
    from kwage_amd import synth
    w = synth.WORKLOADS["c4"]
    with ka.Context(0) as ctx:
        free, _ = ctx.mem_info()
        if free < 185e9:
            pytest.skip("needs ~170 GB of free HBM")
        s = synth.build(ctx, w)
        assert s.group.device_bytes == (1 << 20) * 156288 and len(s.queries) == 10_000
        r1 = s.group.search(s.batch, 1.0)
        assert r1.total_kmers == 970 * 10_000 and r1.algorithmic_bytes == 970 * 10_000 * 156_250
        _check_planted(s, r1)
        _check_sampled(ka, oracle, s, r1, _sample_queries(s, 2, 2), 1.0)
        r2 = s.group.search(s.batch, 1.0, ka.SEARCH_EARLY_EXIT)
        assert np.array_equal(r1.hits, r2.hits)
        r3 = s.group.search(s.batch, 1.0)
        assert np.array_equal(r1.hits, r3.hits)
        s.batch.close()
        s.group.close()


@pytest.mark.timeout(1200)
def test_c5_per_gpu_share_full_size_properties(ka, oracle):
    """BASELINE.json configs[4], what ONE of its 8 GPUs holds: the eight adaptive filter-size groups 2^18 .. 2^25
    (1.09 M samples, 189 GB resident together), 5 hash functions, threshold 0.8 (count path), 10 k x 1 kb queries
    searched against every group."""
    from kwage_amd import synth
    w = synth.WORKLOADS["c5"]
    with ka.Context(0) as ctx:
        free, _ = ctx.mem_info()
        if free < 210e9:
            pytest.skip("needs ~195 GB of free HBM")
        multi = synth.build_multi(ctx, synth.C5_GROUPS, w)
        assert sum(m.group.device_bytes for m in multi) > 185e9
        assert [m.workload.log_2_filter_len for m in multi] == list(range(18, 26))
        for m in multi:
            r1 = m.group.search(multi[0].batch, 0.8)
            assert r1.total_kmers == 970 * 10000
            assert r1.algorithmic_bytes == 970 * 10000 * 5 * ((m.workload.num_samples + 7) // 8)
            assert (r1.query_threshold == 776).all()                      # (unsigned)(0.8f * 970), kwage.cpp:388
            _check_planted(m, r1, complete_match=False)
            _check_sampled(ka, oracle, m, r1, _sample_queries(m, 1, 1), 0.8)
            r2 = m.group.search(multi[0].batch, 0.8, ka.SEARCH_EARLY_EXIT)
            assert np.array_equal(r1.hits, r2.hits)                       # kwage.cpp:478-481 never changes results
        for m in multi:
            m.batch.close()
            m.group.close()


@pytest.mark.timeout(900)
def test_wide_rows_and_five_hashes(ka, oracle):
    """1M samples per row (125 KB rows, as C3/C4) and the 5-hash count path (as C5), on matrices
    small enough to build in seconds."""
    from kwage_amd import synth
    wide = synth.Workload("wide", 1_000_000, 15, 31, 1, 200, 150, 1.0, num_genomes=8, genome_len=6000)
    five = synth.Workload("five", 300_000, 17, 31, 5, 100, 1000, 0.8, density_q8=194, num_genomes=8, genome_len=20000)
    with ka.Context(0) as ctx:
        for w in (wide, five):
            s = synth.build(ctx, w)
            r = s.group.search(s.batch, w.threshold)
            if w.threshold == 1.0:
                _check_planted(s, r)
            hitq = [i for i, g in enumerate(s.query_genome) if g >= 0][:2]
            missq = [i for i, g in enumerate(s.query_genome) if g < 0][:2]
            _check_sampled(ka, oracle, s, r, hitq + missq, w.threshold)
            r2 = s.group.search(s.batch, w.threshold, ka.SEARCH_EARLY_EXIT)
            assert np.array_equal(r.hits, r2.hits)
            # the other reduction on the same matrix
            other = 0.9 if w.threshold == 1.0 else 1.0
            r3 = s.group.search(s.batch, other)
            _check_sampled(ka, oracle, s, r3, hitq[:1] + missq[:1], other)
            s.batch.close()
            s.group.close()


@pytest.mark.timeout(600)
def test_two_to_the_thirty_rows(ka, oracle):
    """log_2_filter_len = 30 (the reference allows up to 32, options.h:153): row indices beyond 2^24 and
    64-bit row offsets (2^30 rows x 128 B = 137 GB)."""
    from kwage_amd import synth
    w = synth.Workload("L30", 1000, 30, 31, 3, 40, 500, 1.0, density_q8=200, num_genomes=2, genome_len=4000)
    with ka.Context(0) as ctx:
        free, _ = ctx.mem_info()
        if free < 150e9:
            pytest.skip("needs ~138 GB of free HBM")
        s = synth.build(ctx, w)
        assert s.group.device_bytes == (1 << 30) * 128
        for thr in (1.0, 0.95):
            r = s.group.search(s.batch, thr)
            if thr == 1.0:
                _check_planted(s, r)
            hitq = [i for i, g in enumerate(s.query_genome) if g >= 0][:2]
            missq = [i for i, g in enumerate(s.query_genome) if g < 0][:2]
            _check_sampled(ka, oracle, s, r, hitq + missq, thr)
        # the row indices really use the upper bits
        k, rows = ka.hash_batch(ctx, 31, 3, 30, s.batch)
        assert max(int(x.max()) for x in rows if x.size) > (1 << 29)
        s.batch.close()
        s.group.close()


@pytest.mark.timeout(600)
def test_c1_reference_shape_cli_vs_reference_binary(ka, oracle, tmp_path):
    """BASELINE.json configs[0] (C1), the reference's own CPU-runnable case at its exact shape: 1 k synthetic
    Bloom filters of 2^20 bits, k = 31, 1 hash, one 10 kb FASTA query.  The same `.db` file and FASTA through the
    REFERENCE binary, this repo's `kwage` CLI and the CPU oracle; three thresholds, CSV and JSON."""
    import os
    import subprocess
    from kwage_amd import native
    if not os.access(oracle.REF_KWAGE, os.X_OK):
        pytest.skip("oracle/_ref/kwage not built")
    rng = np.random.default_rng(2020)
    k, nh, L, ncol = 31, 1, 20, 1000
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, size=30_000)].tobytes().decode()
    query = genome[5_000:15_000]
    a = rng.integers(0, 1 << 63, size=(1 << L, 16), dtype=np.uint64)
    b = rng.integers(0, 1 << 63, size=(1 << L, 16), dtype=np.uint64)
    rows = np.ascontiguousarray((a & b).view(np.uint8)[:, : (ncol + 7) // 8])          # density 0.25, as at the design point
    rows[:, -1] &= np.uint8((1 << (ncol % 8)) - 1) if ncol % 8 else np.uint8(255)
    grows = oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1)
    partial = oracle.row_indices(oracle.unique_kmers(genome[:12_000], k), k, nh, L).reshape(-1)
    for col in (0, 499, 999):
        rows[grows, col // 8] |= np.uint8(1 << (col % 8))
    rows[partial, 77 // 8] |= np.uint8(1 << (77 % 8))                                  # a sample holding 70 % of the query
    infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % j)) for j in range(ncol)]
    db = tmp_path / "db"
    db.mkdir()
    oracle.write_db(str(db / "c1.db"), k, nh, L, rows, ncol, infos)
    fa = tmp_path / "q.fa"
    fa.write_text(">q10kb\n" + "\n".join(query[i:i + 70] for i in range(0, len(query), 70)) + "\n")
    kmers = oracle.unique_kmers(query, k)
    assert len(kmers) == 9970
    for thr in ("1.0", "0.8", "0.6"):
        exp_hits, _ = oracle.search_image(rows, rows.shape[1], k, nh, L, ncol, kmers, float(np.float32(float(thr))))
        for fmt in ("--o.csv", "--o.json"):
            args = ["-d", str(db), "-i", str(fa), "-t", thr, fmt]
            ref = subprocess.run([oracle.REF_KWAGE] + args, capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="1"))
            got = subprocess.run([native.KWAGE_BIN] + args, capture_output=True, text=True)
            assert ref.returncode == 0 and got.returncode == 0, (ref.stderr, got.stderr)
            assert sorted(got.stdout.splitlines()) == sorted(ref.stdout.splitlines()), (thr, fmt)
            if fmt == "--o.csv":
                rep = oracle.parse_csv(got.stdout)["q10kb"]
                assert sorted((a_, nf) for a_, nk, nf, _ in rep) == sorted(("SRR%07d" % c, m) for c, m in exp_hits)
        cols = {c for c, _ in exp_hits}
        assert {0, 499, 999} <= cols and ((77 in cols) == (float(thr) <= 0.7))
