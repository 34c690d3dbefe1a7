"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on identical inputs -- bit-exact hit lists, counts and k-mer sets."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ka():
    import kwage_amd
    return kwage_amd


@pytest.fixture(scope="module")
def ctx(ka):
    c = ka.Context(0)
    yield c
    c.close()


def rand_seq(rng, n):
    return "".join(rng.choice(list("ACGT"), size=n))


# ---------------------------------------------------------------------------------------------
# k-mer stage: word.h:73-104 + kwage.cpp:362-366 + hash.cpp:176-234 on the device
# ---------------------------------------------------------------------------------------------
def test_kmer_stage_golden_vectors(ka, ctx, oracle):
    for case in json.load(open(os.path.join(GOLDEN, "kat_kmers.json"))):
        k, nh, seq = case["k"], case["num_hash"], case["seq"]
        b = ka.Batch(ctx, [seq])
        kmers, rows = ka.hash_batch(ctx, k, nh, 32, b)
        b.close()
        exp = {}
        for e in case["kmers"]:
            exp[int(e["canon"], 16)] = [int(h, 16) for h in e["hash"]]
        got = {int(w): [int(x) for x in r] for w, r in zip(kmers[0], rows[0])}
        assert got == exp, (k, seq)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 15, 16, 17, 21, 27, 28, 29, 30, 31, 32])
def test_kmer_stage_vs_oracle(ka, ctx, oracle, k):
    rng = np.random.default_rng(1000 + k)
    seqs = ["", "A", "ACGT" * 3, "N" * 50, rand_seq(rng, 31), rand_seq(rng, 32), rand_seq(rng, 33),
            rand_seq(rng, 150), rand_seq(rng, 150).lower(), rand_seq(rng, 1000),
            rand_seq(rng, 255 + k), rand_seq(rng, 256 + k), rand_seq(rng, 257 + k),
            rand_seq(rng, 400) + "N" + rand_seq(rng, 10) + "xyz" + rand_seq(rng, 300),
            "ACGTACGTAC" * 100,                       # heavy duplication
            rand_seq(rng, 2047 + k), rand_seq(rng, 2048 + k), rand_seq(rng, 2049 + k),   # LDS/global table edge
            rand_seq(rng, 30000), ("ACGTTGCA" * 8 + "N") * 300]
    b = ka.Batch(ctx, seqs)
    nh, L = 3, 20
    kmers, rows = ka.hash_batch(ctx, k, nh, L, b)
    b.close()
    for s, km, rw in zip(seqs, kmers, rows):
        exp = oracle.unique_kmers(s, k)
        order = np.argsort(km)
        assert np.array_equal(km[order], exp), (k, len(s))
        assert np.array_equal(rw[order], oracle.row_indices(exp, k, nh, L)), (k, len(s))


# ---------------------------------------------------------------------------------------------
# whole search against the reference-written golden databases
# ---------------------------------------------------------------------------------------------
def _device_search(ka, ctx, db_path, seqs, threshold, flags=0):
    from kwage_oracle import read_db
    hdr = read_db(db_path).header
    g = ka.Group(ctx, hdr.kmer_len, hdr.num_hash, hdr.log_2_filter_len, hdr.num_filter)
    first, nf = g.add_db_file(db_path)
    assert (first, nf) == (0, hdr.num_filter)
    g.finalize()
    b = ka.Batch(ctx, seqs)
    r = g.search(b, threshold, flags)
    b.close()
    g.close()
    return r


@pytest.mark.parametrize("rel,qfile", [("basic/db/basic.db", "basic/q.fa"), ("k32/k32.db", "k32/q.fna"),
                                        ("multi/dbs/b/k15_L11_h2.DB", "multi/reads.fastq"),
                                        ("multi/dbs/a/k31_L10_h1.db", "multi/contigs.fa.gz")])
@pytest.mark.parametrize("threshold", [1.0, 0.8, 0.5, 0.05, 0.0001])
def test_search_golden_db_vs_oracle(ka, ctx, oracle, rel, qfile, threshold):
    db_path = os.path.join(GOLDEN, rel)
    db = oracle.read_db(db_path)
    h = db.header
    seqs = [s for _, s in oracle.read_sequences(os.path.join(GOLDEN, qfile))] + ["", "ACGTNACGT"]
    for flags in (0, ka.SEARCH_EARLY_EXIT):
        r = _device_search(ka, ctx, db_path, seqs, threshold, flags)
        per_q = r.per_query()
        thr32 = float(np.float32(threshold))
        for i, s in enumerate(seqs):
            kmers = oracle.unique_kmers(s, h.kmer_len)
            assert r.num_query_kmer[i] == len(kmers)
            exp, _ = oracle.search_image(db.rows, h.slice_size, h.kmer_len, h.num_hash,
                                         h.log_2_filter_len, h.num_filter, kmers, thr32)
            assert per_q[i] == exp, (rel, threshold, i)
            if threshold != 1.0 and len(kmers):
                assert r.query_threshold[i] == oracle.query_threshold(thr32, len(kmers))


def _make_random_db(rng, L, n_cols, density):
    nbytes = (n_cols + 7) // 8
    rows = (rng.random((1 << L, nbytes * 8)) < density)
    rows[:, n_cols:] = True      # garbage in the pad bits of the last byte must never surface
    return np.packbits(rows, axis=1, bitorder="little")


@pytest.mark.parametrize("n_cols,num_hash,k,L", [(1, 1, 31, 8), (7, 2, 21, 9), (8, 3, 31, 10), (100, 1, 31, 10),
                                                  (1000, 5, 15, 9), (8191, 1, 31, 8), (8192, 2, 31, 8),
                                                  (8193, 1, 32, 8), (20000, 3, 31, 7), (70001, 1, 31, 6)])
def test_search_random_db_vs_oracle(ka, ctx, oracle, n_cols, num_hash, k, L):
    """Ragged widths (1 column ... several 1-KiB tiles), every num_hash, AND and count paths."""
    rng = np.random.default_rng(n_cols * 7 + num_hash)
    image = _make_random_db(rng, L, n_cols, 0.5 ** (1.0 / num_hash) if n_cols < 5000 else 0.9)
    seqs = [rand_seq(rng, n) for n in (k, k + 1, 40, 60, 150, 5)] + [rand_seq(rng, 45) * 3, ""]
    g = ka.Group(ctx, k, num_hash, L, n_cols)
    assert g.add_columns(image, n_cols) == 0
    g.finalize()
    # read_rows returns exactly what was uploaded
    some = [0, 1, (1 << L) - 1]
    assert np.array_equal(g.read_rows(some)[:, :image.shape[1]], image[some])
    b = ka.Batch(ctx, seqs)
    for threshold in (1.0, 0.9, 0.5, 0.02):
        thr32 = float(np.float32(threshold))
        for flags in (0, ka.SEARCH_EARLY_EXIT):
            r = g.search(b, threshold, flags)
            per_q = r.per_query()
            for i, s in enumerate(seqs):
                kmers = oracle.unique_kmers(s, k)
                exp, _ = oracle.search_image(image, image.shape[1], k, num_hash, L, n_cols, kmers, thr32)
                assert r.num_query_kmer[i] == len(kmers)
                assert per_q[i] == exp, (n_cols, num_hash, threshold, flags, i)
    b.close()
    g.close()


def test_multiple_files_in_one_group(ka, ctx, oracle):
    """Column concatenation: two reference files with equal parameters share one matrix; hits map
    back to (file, column) and equal the per-file oracle results."""
    pa = os.path.join(GOLDEN, "multi/dbs/a/k31_L10_h1.db")
    pb = os.path.join(GOLDEN, "multi/dbs/a/deeper/k31_L10_h1_b.db")
    da, db = oracle.read_db(pa), oracle.read_db(pb)
    g = ka.Group(ctx, 31, 1, 10, 16 * 8 + 21)
    fa, na = g.add_db_file(pa)
    fb, nb = g.add_db_file(pb)
    assert (fa, na, nb) == (0, 13, 21) and fb == 128        # second file starts 16-byte aligned
    g.finalize()
    assert g.num_columns == 34
    seqs = [s for _, s in oracle.read_sequences(os.path.join(GOLDEN, "multi/contigs.fa.gz"))]
    seqs += [s for _, s in oracle.read_sequences(os.path.join(GOLDEN, "multi/reads.fastq"))]
    b = ka.Batch(ctx, seqs)
    for threshold in (1.0, 0.7, 0.001):
        r = g.search(b, threshold)
        per_q = r.per_query()
        for i, s in enumerate(seqs):
            kmers = oracle.unique_kmers(s, 31)
            ea, _ = oracle.search_image(da.rows, da.header.slice_size, 31, 1, 10, 13, kmers, float(np.float32(threshold)))
            eb, _ = oracle.search_image(db.rows, db.header.slice_size, 31, 1, 10, 21, kmers, float(np.float32(threshold)))
            assert per_q[i] == ea + [(c + fb, m) for c, m in eb], (threshold, i)
    b.close()
    g.close()


@pytest.mark.parametrize("n_queries", [40, 93])
def test_hit_buffer_growth(ka, ctx, oracle, n_queries, monkeypatch):
    """threshold truncating to 0 makes EVERY column match (kwage.cpp:388,497): more hits than the
    initial device buffer holds -> the engine must grow it and still return all of them.  Lists this long are
    sorted on the device and land in a pinned block of the context's pool, in one copy or (knob) in ragged pieces;
    the block goes back to the pool with the result and serves the next long list."""
    n_cols, L, k = 40000, 6, 31
    g = ka.Group(ctx, k, 1, L, n_cols)
    g.add_random_columns(n_cols, 7, 64)
    g.finalize()
    rng = np.random.default_rng(5)
    seqs = [rand_seq(rng, 60) for _ in range(n_queries)]
    b = ka.Batch(ctx, seqs)
    r = g.search(b, 0.001)
    with ctx.tuning(hit_sort_host=1):
        assert np.array_equal(g.search(b, 0.001).hits, r.hits)      # the host's sort of the same list
    with ctx.tuning(hit_copy_piece_kb=5000):
        assert np.array_equal(g.search(b, 0.001).hits, r.hits)      # 4 resp. 9 pieces, the last one ragged
    p1, p2 = g.submit(b, 0.001), g.submit(b, 0.001)                 # two long lists alive at once: two pool blocks
    r1, r2 = p1.collect(), p2.collect()
    assert np.array_equal(r1.hits, r.hits) and np.array_equal(r2.hits, r.hits)
    assert len(r.hits) == n_cols * len(seqs) and r.search_kernel_launches == 2
    assert np.array_equal(r.hits["query"], np.repeat(np.arange(len(seqs), dtype=np.uint32), n_cols))
    assert np.array_equal(r.hits["column"][:n_cols], np.arange(n_cols, dtype=np.uint32))
    # counts: against the oracle on the rows actually resident on the device
    image = g.read_rows(np.arange(1 << L))
    for i in (0, 17, n_queries - 1):
        kmers = oracle.unique_kmers(seqs[i], k)
        exp, _ = oracle.search_image(image, image.shape[1], k, 1, L, n_cols, kmers, float(np.float32(0.001)))
        got = r.hits[r.hits["query"] == i]
        assert [(int(c), int(m)) for c, m in zip(got["column"], got["num_match"])] == exp
    b.close()
    g.close()


def test_poll_and_exchange_buffers(ka, ctx, oracle):
    """kwage_search_poll says when collecting would not wait any more (never consumes the search), and the exchange
    buffers of the persistent kernels read all zero after cut-heavy searches (kwage_ctx_scratch_nonzero): what
    tools/soak_walk.py checks after tens of thousands of launches, once here."""
    import ctypes as C
    import time
    from kwage_amd.native import lib
    rng = np.random.default_rng(99)
    k, nh, L, n_cols = 31, 2, 10, 40000
    image = _make_random_db(rng, L, n_cols, 0.8)
    seqs = [rand_seq(rng, int(rng.choice([40, 64, 150, 300]))) for _ in range(400)]
    g = ka.Group(ctx, k, nh, L, n_cols)
    g.add_columns(image, n_cols)
    g.finalize()
    b = ka.Batch(ctx, seqs)
    with ctx.tuning(walk=0, count_walk=0):
        ref = {t: g.search(b, t) for t in (1.0, 0.9)}
    with ctx.tuning(walk=4, walk_min_rows=1, walk_waves=3001, count_walk_min_rows=1, count_walk_waves=3001, walk_bands=0):
        for t in (1.0, 0.9, 1.0, 0.9):
            p = g.submit(b, t)
            polls = 0
            while True:
                state = lib().kwage_search_poll(p._h)
                assert state in (0, 1)
                if state == 1:
                    break
                polls += 1
                assert polls < 200000
                time.sleep(0.0001)
            assert lib().kwage_search_poll(p._h) == 1          # still there: polling does not consume it
            r = p.collect()
            assert r.search_kernel.startswith("and_walk_kernel<" if t == 1.0 else "count_walk_kernel<")
            assert np.array_equal(r.hits, ref[t].hits)
        with ctx.tuning(walk_bands=5, walk_bands_min_gib=0):
            assert np.array_equal(g.search(b, 1.0).hits, ref[1.0].hits)
        left = ctx.scratch_nonzero()
    assert set(left) == {"walk_or", "walk_done", "band_or", "band_state", "cwalk_arrived"} and not any(left.values()), left
    b.close()
    g.close()


@pytest.mark.parametrize("n_cols", [2048, 4000, 5000, 40000, 300000])
def test_long_lists_come_back_ordered_from_every_kernel_form(ka, ctx, n_cols):
    """Lists above the 8192 records that come back with the counters are ordered on the device WITHOUT a sort: every
    reservation of hit slots is a run that is already ascending, the gather kernels note every run in a table indexed in
    key order, and a prefix sum + one copy per run puts the list in order (hit_sort.hip).  Every kernel that reserves
    slots numbers its runs itself, so every form is driven here to a long list whose order is known in closed form: a
    matrix whose columns are all ones except every seventh (all zero), so every query with a k-mer reports exactly the
    other columns, ascending -- narrow kernels (several queries per wave, one reservation per workgroup), the tiled AND
    kernel with 1 / 2 / 4 vectors per lane, its segments + combine pass, the walk form (one and three column tiles,
    natural shares and shares of a few positions), band after band, and the four forms of the count path."""
    L, k = 6, 31
    nbytes = (n_cols + 7) // 8
    col = np.arange(nbytes * 8)
    bits = ((col % 7 != 3) & (col < n_cols)).astype(np.uint8).reshape(nbytes, 8)
    row = np.packbits(bits, axis=1, bitorder="little").reshape(-1)
    image = np.ascontiguousarray(np.tile(row, (1 << L, 1)))
    one_cols = np.arange(n_cols, dtype=np.uint32)[np.arange(n_cols) % 7 != 3]
    rng = np.random.default_rng(n_cols)
    n_queries = 200 if n_cols <= 5000 else (90 if n_cols <= 40000 else 24)
    seqs = []
    for i in range(n_queries):
        # (at most 120 k-mers: no automatic row-list segments; at least 3: a threshold that truncates to 0 would report the zero columns too)
        n = int(rng.choice([33, 35, 40, 64, 100, 150]))
        seqs.append("ACGT" if i % 9 == 4 else ("N" * n if i % 13 == 5 else rand_seq(rng, n)))
    g = ka.Group(ctx, k, 2, L, n_cols)
    g.add_columns(image, n_cols)
    g.finalize()
    b = ka.Batch(ctx, seqs)

    def check(r, want_kernel):
        assert r.search_kernel.startswith(want_kernel), (r.search_kernel, want_kernel)
        nk = r.num_query_kmer
        live = np.flatnonzero(nk > 0).astype(np.uint32)
        assert len(live) >= n_queries // 2 and len(r.hits) == len(live) * len(one_cols) > 8192
        assert np.array_equal(r.hits["query"], np.repeat(live, len(one_cols))), want_kernel
        assert np.array_equal(r.hits["column"], np.tile(one_cols, len(live))), want_kernel
        assert np.array_equal(r.hits["num_match"], np.repeat(nk[live], len(one_cols))), want_kernel

    units = (nbytes + 127) // 128 * 8
    off = dict(walk=0, count_walk=0, narrow=0, force_segs=0, walk_bands=0)
    if units <= 32:
        with ctx.tuning(**dict(off, narrow=1)):
            check(g.search(b, 1.0), "and_narrow_kernel<")
            if units > 8:
                check(g.search(b, 0.5), "count_narrow_kernel<")
    for vec in (1, 2, 4):
        with ctx.tuning(**dict(off, and_vec=vec)):
            check(g.search(b, 1.0), "and_kernel<%d," % vec)
    with ctx.tuning(**dict(off, force_segs=3)):
        check(g.search(b, 1.0), "and_kernel<")
        assert g.search(b, 1.0).search_kernel.endswith("+segments")
        check(g.search(b, 0.5), "count_kernel<")
    with ctx.tuning(**off):
        check(g.search(b, 0.5), "count_kernel<")
    if units >= 64:
        # early exit, every tile handed over to the refine launch after its first eight rows: one reservation per cluster
        # (= per query and KiB-step), by the emit launch; and with lists of three places: most tiles walk on by themselves
        for knobs in (dict(refine_max_groups=16, refine_min_rows=1, refine_seg_rows=8), dict(refine_max_groups=16, refine_min_rows=1, refine_list_cap=3)):
            with ctx.tuning(**dict(off, count_screen_min_tiles=1, **knobs)):
                check(g.search(b, 1.0, ka.SEARCH_EARLY_EXIT), "and_screen_kernel<")
                check(g.search(b, 0.5, ka.SEARCH_EARLY_EXIT), "count_screen_kernel<")
    for waves in (0, 37):
        with ctx.tuning(**dict(off, count_walk=1, count_walk_min_rows=1, count_walk_waves=waves)):
            check(g.search(b, 0.5), "count_walk_kernel<")
        if units >= 2 * 64:
            with ctx.tuning(**dict(off, walk=4, walk_min_rows=1, walk_max_kib=64, walk_waves=waves)):
                check(g.search(b, 1.0), "and_walk_kernel<")
            if units <= 16 * 64:
                with ctx.tuning(**dict(off, walk=4, walk_min_rows=1, walk_waves=waves, walk_bands=3, walk_bands_min_gib=0)):
                    check(g.search(b, 1.0), "and_band_walk_kernel<")
    b.close()
    g.close()


@pytest.mark.parametrize("n_cols,num_hash,density", [(8192, 1, 0.25), (20000, 2, 0.5), (100000, 1, 0.25), (131073, 3, 0.7)])
def test_early_exit_screen_then_refine(ka, ctx, oracle, n_cols, num_hash, density):
    """Early exit on rows of a KiB and more.  Threshold 1: and_screen_kernel hands every tile that still holds a candidate
    column after its first rows over to and_refine_kernel (128-byte groups, segments of the remaining rows, masks meeting
    by atomic AND) and and_refine_emit_kernel (one reservation per query and KiB-step).  Threshold < 1: count_screen_kernel
    hands over the tiles in which few columns can still reach the threshold (kwage.cpp:478-481 per column), with their
    counters so far; count_refine_kernel counts segments of the remaining k-mers, count_refine_emit_kernel adds up.
    Planted columns in the first, middle and last 128-byte groups, two of them in ONE group and two in one KiB-step, ragged
    query lengths (shorter than one segment ... dozens of segments), exact and mutated copies of the planted genome
    (counts between the threshold and num_query_kmer), queries without a k-mer in between; swept over segment lengths, the
    hand-over rule, both unrolls and list capacities so small that most tiles find the lists full and walk on by
    themselves.  Every variant must return the list of the search without early exit, which is checked against the oracle."""
    rng = np.random.default_rng(n_cols + num_hash)
    k, L = 31, 10
    image = _make_random_db(rng, L, n_cols, density)
    genome = rand_seq(rng, 3000)
    cols = sorted({0, 5, 1030, n_cols // 2, n_cols - 1000, n_cols - 2, n_cols - 1})
    for col in cols:
        for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, num_hash, L).reshape(-1):
            image[r, col // 8] |= np.uint8(1 << (col % 8))
    seqs = []
    for i in range(300):
        n = int(rng.choice([0, 30, 31, 33, 40, 64, 100, 150, 300, 1000, 2500]))
        if i % 2 == 0 and n >= 31:
            a = int(rng.integers(0, len(genome) - n + 1))
            q = list(genome[a:a + n])
            if i % 4 == 0:                      # a copy with a substitution every ~150 bases: most k-mers still match
                for pos in range(int(rng.integers(0, 150)), n, 150):
                    q[pos] = "ACGT"[("ACGT".index(q[pos]) + 1) % 4]
            seqs.append("".join(q))
        elif i % 11 == 0:
            seqs.append("N" * n)
        else:
            seqs.append(rand_seq(rng, n))
    g = ka.Group(ctx, k, num_hash, L, n_cols)
    g.add_columns(image, n_cols)
    g.finalize()
    b = ka.Batch(ctx, seqs)
    variants = [dict(), dict(refine_seg_rows=8, refine_min_rows=1), dict(refine_seg_rows=1000000, refine_unroll=16), dict(refine_max_groups=16, refine_min_rows=1, refine_seg_rows=17),
                dict(refine_max_groups=1), dict(refine_max_groups=0), dict(refine_list_cap=1, refine_min_rows=1), dict(refine_list_cap=7, refine_max_groups=16),
                dict(refine_list_cap=40, refine_seg_rows=8), dict(ee_refine=0)]
    for thr in (1.0, 0.8, 0.5):
        ref = g.search(b, thr, 0)
        thr32 = float(np.float32(thr))
        if thr == 1.0 or n_cols <= 20000:       # (the oracle's per-bit counting loop over 100 k columns is slow)
            exp = [oracle.search_image(image, image.shape[1], k, num_hash, L, n_cols, oracle.unique_kmers(s, k), thr32)[0] for s in seqs]
            assert ref.per_query() == exp
            planted = [e for s, e in zip(seqs, exp) if len(s) >= 31 and s in genome]
            assert planted and all({c for c, _ in e} >= set(cols) for e in planted)
        want = "and_screen_kernel<" if thr == 1.0 else "count_screen_kernel<"
        for knobs in variants:
            with ctx.tuning(count_screen_min_tiles=1, **knobs):
                for rep in range(2):
                    r = g.search(b, thr, ka.SEARCH_EARLY_EXIT)
                    assert r.search_kernel.startswith(want) != (knobs.get("ee_refine") == 0), (r.search_kernel, knobs)
                    assert np.array_equal(r.hits, ref.hits) and np.array_equal(r.num_query_kmer, ref.num_query_kmer), (n_cols, thr, knobs, rep)
    # append mode with early exit (what kwage_node and the multi-GPU hosts submit): the screen / refine launches append to a
    # caller-owned list with a column base, no run table
    import ctypes as C
    import torch
    from kwage_amd.native import check, lib
    for thr in (1.0, 0.8):
        ref = g.search(b, thr, 0)
        buf = torch.empty((1 + len(ref.hits) + 16, 3), dtype=torch.int32, device="cuda:0")
        n, h = C.c_uint64(), C.c_void_p()
        with ctx.tuning(count_screen_min_tiles=1):
            check(lib().kwage_search_device_append_submit(g._h, b._h, C.c_float(thr), ka.SEARCH_EARLY_EXIT, buf.data_ptr() + 12, buf.shape[0] - 1, buf.data_ptr(), 1000, 1, C.byref(h)))
            check(lib().kwage_search_device_collect(h, C.byref(n), None, None))
        assert n.value == len(ref.hits)
        host = buf.cpu().numpy().view(np.uint32)
        got = sorted(map(tuple, host[1:1 + n.value].tolist()))
        assert got == sorted((int(q), int(c) + 1000, int(m)) for q, c, m in zip(ref.hits["query"], ref.hits["column"], ref.hits["num_match"])), (n_cols, thr)
    # one query of 12 k positions among the others: units of 1/128 of it, 14 counter planes per unit (queries above 16383
    # positions -- 20 counter planes -- keep the tiled kernel at t < 1)
    seqs2 = seqs[:40] + [(genome * 5)[:12500], rand_seq(rng, 9000)]
    b2 = ka.Batch(ctx, seqs2)
    for thr in (1.0, 0.7):
        ref = g.search(b2, thr, 0)
        with ctx.tuning(count_screen_min_tiles=1, refine_min_rows=1):
            r = g.search(b2, thr, ka.SEARCH_EARLY_EXIT)
        assert r.search_kernel.startswith("and_screen_kernel<" if thr == 1.0 else "count_screen_kernel<") and (thr == 1.0 or r.search_kernel.endswith("+refine<14>")), r.search_kernel
        assert np.array_equal(r.hits, ref.hits) and np.array_equal(r.num_query_kmer, ref.num_query_kmer), (n_cols, thr)
    b2.close()
    b.close()
    g.close()


@pytest.mark.parametrize("n_cols,num_hash,q8", [(20000, 1, 64), (100000, 2, 160), (40000, 3, 190)])
def test_early_exit_long_queries_truncated_walk(ka, ctx, oracle, n_cols, num_hash, q8):
    """FEW LONG queries at t < 1 with early exit: the truncated count walk -- the persistent count kernel over the first
    k-mers of every query (as many as the bound max + remaining < threshold needs before it can rule out the matrix's
    DENSEST column, sampled at finalize), the columns that can still reach the threshold handed over with their counters
    (after the tree has summed the parts of pairs cut across dozens of waves), their remaining k-mers counted by the refine
    launch.  Exact, mutated and foreign queries of 3 k ... 40 k positions (14 and 20 counter planes), natural wave counts and
    shares of a handful of positions, lists of five places (nearly every pair is then counted to the end by the wave that
    holds it).  Every variant returns the list of the search without early exit, which is checked against the oracle on
    the rows read back from the device."""
    rng = np.random.default_rng(n_cols + num_hash)
    k, L = 31, 18                                  # (2^18 rows: the planted columns stay well below all ones -- the densest column plans the truncation)
    g = ka.Group(ctx, k, num_hash, L, n_cols)
    g.add_random_columns(n_cols, 77 + num_hash, q8)
    genome = rand_seq(rng, 42000)
    gb = ka.Batch(ctx, [genome])
    _, rows = ka.hash_batch(ctx, k, num_hash, L, gb)
    gb.close()
    cols = sorted({0, 1030, n_cols // 2, n_cols - 1})
    for col in cols:
        r = rows[0].reshape(-1)
        g.set_bits(r, np.full(r.shape, col, dtype=np.uint64))
    g.finalize()
    seqs3 = [genome[:3000], genome[100:20100], rand_seq(rng, 12000), genome[5:40005], rand_seq(rng, 3000), genome[7:9007]]
    for j in (1, 3):                               # substitutions every ~90 bases: counts between the threshold and num_query_kmer
        q = list(seqs3[j])
        for pos in range(11, len(q), 90):
            q[pos] = "ACGT"[("ACGT".index(q[pos]) + 1) % 4]
        seqs3.append("".join(q))
    b3 = ka.Batch(ctx, seqs3)
    used = set()
    for thr in (0.9, 0.8, 0.5):
        ref = g.search(b3, thr, 0)
        thr32 = float(np.float32(thr))
        per_q = ref.per_query()
        for qi in (0, 2, 6):                         # against the oracle on the rows these queries address, read back from HBM
            kmers = oracle.unique_kmers(seqs3[qi], k)
            matrix = g.read_rows(oracle.row_indices(kmers, k, num_hash, L).reshape(-1))
            assert per_q[qi] == oracle.search_row_matrix(matrix, num_hash, n_cols, len(kmers), thr32), (n_cols, thr, qi)
        assert {c for c, _ in per_q[0]} >= set(cols)
        for waves in (0, 7, 3001):
            with ctx.tuning(count_walk_min_rows=1, count_walk_waves=waves):
                for rep in range(2):
                    r = g.search(b3, thr, ka.SEARCH_EARLY_EXIT)
                    used.add((thr, r.search_kernel))
                    assert np.array_equal(r.hits, ref.hits) and np.array_equal(r.num_query_kmer, ref.num_query_kmer), (n_cols, thr, waves, rep, r.search_kernel)
        # lists of five places: nearly every pair finds them full and is counted to the end by the wave that holds it
        for waves in (0, 3001):
            with ctx.tuning(count_walk_min_rows=1, count_walk_waves=waves, refine_list_cap=5):
                r = g.search(b3, thr, ka.SEARCH_EARLY_EXIT)
                assert np.array_equal(r.hits, ref.hits) and np.array_equal(r.num_query_kmer, ref.num_query_kmer), (n_cols, thr, waves, "full lists", r.search_kernel)
        with ctx.tuning(count_trunc=0):
            r = g.search(b3, thr, ka.SEARCH_EARLY_EXIT)
            assert "trunc" not in r.search_kernel and np.array_equal(r.hits, ref.hits)
    assert any(t == 0.9 and "trunc>+refine<" in name for t, name in used), used
    left = ctx.scratch_nonzero()
    assert not any(left.values()), left              # the tree's arrival counters are zero again
    b3.close()
    g.close()


@pytest.mark.parametrize("n_cols,n_queries", [(1, 12000), (3, 5000), (9, 1500), (8193, 2), (8193, 33), (100000, 1), (100001, 17), (70000, 300)])
def test_device_hit_sort_key_widths(ka, ctx, n_cols, n_queries, monkeypatch):
    """The device sort packs (query, column) into query_bits + column_bits of one key: widths from 0 bits (one
    query, one column) up, every list longer than the 8192 records that come back with the counters.  At a threshold
    that truncates to 0 every column matches every query with at least one k-mer, so the sorted list is known in
    closed form; the counts are compared with the host-sorted run of the same search."""
    g = ka.Group(ctx, 31, 1, 6, n_cols)
    g.add_random_columns(n_cols, 11, 128)
    g.finalize()
    rng = np.random.default_rng(n_cols + n_queries)
    seqs = [rand_seq(rng, 31 + int(rng.integers(0, 30))) if i % 7 != 3 else "ACGT" for i in range(n_queries)]     # some without a k-mer
    b = ka.Batch(ctx, seqs)
    r = g.search(b, 0.0001)
    live = np.array([i for i, s in enumerate(seqs) if len(s) >= 31], dtype=np.uint32)
    assert len(r.hits) == len(live) * n_cols > 8192
    assert np.array_equal(r.hits["query"], np.repeat(live, n_cols))
    assert np.array_equal(r.hits["column"], np.tile(np.arange(n_cols, dtype=np.uint32), len(live)))
    with ctx.tuning(hit_sort_host=1):
        assert np.array_equal(g.search(b, 0.0001).hits, r.hits)
    b.close()
    g.close()


def test_device_hit_sort_with_files_padded_apart(ka, ctx, monkeypatch):
    """Column indices of a group run over its SPAN: every file starts at a 16-byte boundary, so the largest index
    exceeds the count of valid columns (here 3 blocks of 1000 + 37 + 3000 columns = 4037 < 2^12 span 1024 + 128 + 3000 =
    4152 > 2^12).  The sort key must be sized by the span: sized by the count, bit 12 of a column would spill into the
    query field."""
    blocks = (1000, 37, 3000)
    g = ka.Group(ctx, 31, 1, 6, 1024 + 128 + 3000)
    firsts = [g.add_random_columns(n, 3 + i, 128) for i, n in enumerate(blocks)]
    g.finalize()
    assert firsts == [0, 1024, 1152] and g.num_columns == sum(blocks) < 4096 < firsts[-1] + blocks[-1]
    columns = np.concatenate([np.arange(f, f + n, dtype=np.uint32) for f, n in zip(firsts, blocks)])
    rng = np.random.default_rng(9)
    seqs = [rand_seq(rng, 40) for _ in range(23)]
    b = ka.Batch(ctx, seqs)
    r = g.search(b, 0.0001)
    assert len(r.hits) == len(seqs) * len(columns) > 8192
    assert np.array_equal(r.hits["query"], np.repeat(np.arange(len(seqs), dtype=np.uint32), len(columns)))
    assert np.array_equal(r.hits["column"], np.tile(columns, len(seqs)))
    with ctx.tuning(hit_sort_host=1):
        assert np.array_equal(g.search(b, 0.0001).hits, r.hits)
    b.close()
    g.close()


def test_errors_are_reported_not_swallowed(ka, ctx):
    with pytest.raises(ka.KwageError):
        ka.Group(ctx, 33, 1, 10, 8)                    # k > MAX_WORD_LEN
    with pytest.raises(ka.KwageError):
        ka.Group(ctx, 31, 6, 10, 8)                    # > MAX_NUM_HASH
    with pytest.raises(ka.KwageError) as e:
        ka.Group(ctx, 31, 1, 10, 8, hash_func=1)       # hash.cpp:92
    assert "Unknown hash function" in str(e.value)
    g = ka.Group(ctx, 31, 1, 8, 64)
    g.add_random_columns(64, 1, 128)
    b = ka.Batch(ctx, ["ACGT" * 20])
    with pytest.raises(ka.KwageError):
        g.search(b, 1.0)                               # not finalized
    g.finalize()
    with pytest.raises(ka.KwageError):
        g.search(b, 0.0)                               # options.cpp:186
    with pytest.raises(ka.KwageError):
        g.add_random_columns(8, 1, 128)                # finalized
    with pytest.raises(ka.KwageError):
        g.add_db_file(os.path.join(GOLDEN, "k32/k32.db"))
    b.close()
    g.close()


# ---------------------------------------------------------------------------------------------
# the drop-in CLI against the reference binary's recorded output
# ---------------------------------------------------------------------------------------------
def _cases():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "%s-%s" % (c["name"], c["expected"]))
def test_cli_matches_reference_output(ka, oracle, case):
    from kwage_amd import native
    cdir = os.path.join(GOLDEN, case["name"])
    args = [native.KWAGE_BIN]
    for d in case["db"]:
        args += ["-d", d]
    for q in case["queries"]:
        args += ["-i", q]
    args += ["-t", case["threshold"], "--o." + case["format"]] + case["cmdline"]
    r = subprocess.run(args, cwd=cdir, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    got = r.stdout.decode("latin-1")
    exp = open(os.path.join(cdir, case["expected"]), encoding="latin-1").read()
    if case["format"] == "csv":
        g, e = oracle.parse_csv(got), oracle.parse_csv(exp)
        assert list(g) == list(e)                       # same queries, same order
        for q in e:
            assert sorted(g[q]) == sorted(e[q]), q
            assert [x[2] for x in g[q]] == [x[2] for x in e[q]]    # descending by hits, as the reference prints
    else:
        # order among equal-score hits is unspecified in the reference; compare as multisets of
        # lines plus the exact sequence of scores
        assert sorted(got.splitlines()) == sorted(exp.splitlines())
        score = lambda t: [ln for ln in t.splitlines() if "num_kmers_found" in ln or '"query"' in ln]
        assert score(got) == score(exp)
    if len(case["db"]) == 1 and case["name"] != "multi":
        assert got == exp                               # single file: byte-identical
    # sparse fetch (at t = 1 screened on the first 32 / 2 k-mers of every query first, or not at all) / whole files / streamed
    for mode in ({"KWAGE_SPARSE": "1"}, {"KWAGE_SPARSE": "1", "KWAGE_SPARSE_SCREEN": "2"}, {"KWAGE_SPARSE": "1", "KWAGE_SPARSE_SCREEN": "0"}, {"KWAGE_SPARSE": "0"}, {"KWAGE_SPARSE_BASES": "1"}):
        r2 = subprocess.run(args, cwd=cdir, capture_output=True, env=dict(os.environ, **mode))
        assert r2.returncode == 0 and r2.stdout == r.stdout, mode


def test_cli_sparse_fetch_screened_on_first_kmers(oracle, tmp_path):
    """`kwage` at t = 1 with slices fetched on demand: the slices of every query's first KWAGE_SPARSE_SCREEN k-mers come
    first and only files with a candidate column are fetched for the rest (kwage_main.cpp, screen_then_fetch).  Same bytes
    as without the screen and as the whole-file path, and the reference binary's lines: hits in some files only, a query
    whose head holds no valid k-mer (N's) but whose tail does, one shorter than the head, one shorter than k, one with no
    hit at all; with a budget that takes the candidate files one at a time; verbose line says what was screened."""
    from kwage_amd import native
    rng = np.random.default_rng(4242)
    k, nh, L = 31, 2, 14
    genome, other, third = rand_seq(rng, 3000), rand_seq(rng, 1500), rand_seq(rng, 1200)
    db = tmp_path / "db"
    db.mkdir()
    for f, (ncol, planted) in enumerate(((2048, genome), (300, None), (2048, other), (64, genome), (1000, third))):
        img = _make_random_db(rng, L, ncol, 0.4)
        if planted is not None:
            for r in oracle.row_indices(oracle.unique_kmers(planted, k), k, nh, L).reshape(-1):
                img[r, (5 + f) // 8] |= np.uint8(1 << ((5 + f) % 8))
        infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % (f * 10000 + j + 1))) for j in range(ncol)]
        oracle.write_db(str(db / ("f%d.db" % f)), k, nh, L, img, ncol, infos)
    seqs = [genome[200:1700], "N" * 70 + genome[500:1500], rand_seq(rng, 900), genome[:40], "ACGT", other[100:1400].lower(), genome[1000:1300] + "N" + other[:300]]
    fa = tmp_path / "q.fa"
    fa.write_text("".join(">q%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    args = ["-d", str(db), "-i", str(fa), "-t", "1.0", "--o.csv", genome[300:700]]

    def run(env):
        r = subprocess.run([native.KWAGE_BIN] + args, capture_output=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        return r.stdout, r.stderr.decode()
    plain, _ = run({"KWAGE_SPARSE": "0"})
    assert plain.count(b"\n") > 4
    outs = {}
    for head in ("0", "4", "32", "100000"):
        outs[head], err = run({"KWAGE_SPARSE": "1", "KWAGE_SPARSE_SCREEN": head, "KWAGE_VERBOSE": "1"})
        assert outs[head] == plain, head
        assert ("screened 2^14 slices" in err) == (head in ("4", "32")), (head, err[-1500:])
    # the candidate files one at a time (a budget below one file's rows), and the default mode
    small, err = run({"KWAGE_SPARSE": "1", "KWAGE_SPARSE_SCREEN": "32", "KWAGE_MAX_GROUP_BYTES": str(256 * 3000), "KWAGE_VERBOSE": "1"})
    assert small == plain and " of 8 queries hold a candidate, in 5 files" in err, err[-1500:]     # (5: the query with N's up front meets every file)
    assert run({})[0] == plain
    if os.access(oracle.REF_KWAGE, os.X_OK):
        ref = subprocess.run([oracle.REF_KWAGE] + args, capture_output=True, env=dict(os.environ, OMP_NUM_THREADS="1"))
        assert ref.returncode == 0 and sorted(ref.stdout.splitlines()) == sorted(plain.splitlines())
    # three queries with their candidates in different files: every candidate file is fetched on its own, with its own queries' slices
    fa2 = tmp_path / "q2.fa"
    fa2.write_text(">a\n%s\n>b\n%s\n>c\n%s\n" % (genome[200:1200], other[100:1100], third[:1000]))
    args = ["-d", str(db), "-i", str(fa2), "-t", "1.0", "--o.json"]
    plain, _ = run({"KWAGE_SPARSE": "0"})
    got, err = run({"KWAGE_SPARSE": "1", "KWAGE_VERBOSE": "1"})
    assert got == plain and plain.count(b"num_kmers_found") >= 4
    assert "3 of 3 queries hold a candidate, in 4 files" in err and "fetched one by one" in err, err[-1500:]
    assert run({"KWAGE_SPARSE": "1", "KWAGE_SPARSE_SCREEN": "0"})[0] == plain


def test_sparse_group_gives_the_same_hits(ka, ctx, oracle, tmp_path):
    """kwage_group_create_sparse: only the slices a batch addresses are fetched from the files (raw and compressed) and
    kept resident; hit lists equal those of the full group for every kernel family (AND / count, narrow / tiled / walk
    widths), a batch that addresses other rows is refused, and add_columns takes a compact image."""
    from kwage_amd import native
    import ctypes as C
    rng = np.random.default_rng(99)
    k, nh, L = 31, 3, 13
    genome = rand_seq(rng, 2500)
    files, full_cols = [], 0
    for f, ncol in enumerate((2048, 100, 30000, 2048)):
        img = _make_random_db(rng, L, ncol, 0.8)
        for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1):
            img[r, (7 + f) // 8] |= np.uint8(1 << ((7 + f) % 8))
        infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % (f * 100000 + j))) for j in range(ncol)]
        p = str(tmp_path / ("s%d.db" % f))
        oracle.write_db(p, k, nh, L, img, ncol, infos)
        if f == 1:                                          # one of them in the compressed container
            z = str(tmp_path / "s1.dbz")
            native.check(native.lib().kwage_db_compress(p.encode(), z.encode(), 2))
            p = z
        files.append(p)
        full_cols += ((ncol + 127) // 128) * 128
    seqs = [genome[100:600], rand_seq(rng, 300), genome[1000:1200].lower(), "ACGT", rand_seq(rng, 150), genome[:2000]]
    b = ka.Batch(ctx, seqs)
    _, rows = ka.hash_batch(ctx, k, nh, L, b)
    need = np.unique(np.concatenate([r.reshape(-1) for r in rows]))
    assert 0 < need.size < (1 << L)
    full = ka.Group(ctx, k, nh, L, full_cols)
    firsts_full = full.add_db_files(files)
    full.finalize()
    sp = ka.Group.sparse(ctx, k, nh, L, full_cols, need)
    assert sp.add_db_files(files) == firsts_full and sp.device_bytes < full.device_bytes
    sp.finalize()
    kernels = set()
    for knobs in ({}, dict(walk_min_rows=1, count_walk_min_rows=1, walk_waves=23, count_walk_waves=23)):      # tiled kernels; the persistent ones on translated row lists
        with ctx.tuning(**knobs):
            for thr in (1.0, 0.8, 0.3):
                for flags in (0, ka.SEARCH_EARLY_EXIT):
                    a, c = full.search(b, thr, flags), sp.search(b, thr, flags)
                    assert np.array_equal(a.hits, c.hits) and np.array_equal(a.num_query_kmer, c.num_query_kmer), (thr, flags, knobs)
                    assert len(a.hits) > 0 and a.search_kernel == c.search_kernel
                    kernels.add(c.search_kernel.split("<")[0])
    assert {"and_kernel", "and_walk_kernel", "count_kernel", "count_walk_kernel"} <= kernels, kernels
    other = ka.Batch(ctx, [rand_seq(rng, 400)])
    with pytest.raises(ka.KwageError) as ei:
        sp.search(other, 1.0)
    assert "not among the rows of the sparse group" in str(ei.value)
    assert np.array_equal(sp.search(b, 1.0).hits, full.search(b, 1.0).hits)          # the group is still usable afterwards
    other.close()
    # a compact host image through add_columns
    img = _make_random_db(rng, L, 500, 0.9)
    sp2 = ka.Group.sparse(ctx, k, nh, L, 512, need)
    sp2.add_columns(np.ascontiguousarray(img[need]), 500)
    sp2.finalize()
    dense = ka.Group(ctx, k, nh, L, 512)
    dense.add_columns(img, 500)
    dense.finalize()
    assert np.array_equal(sp2.search(b, 0.5).hits, dense.search(b, 0.5).hits)
    for g in (full, sp, sp2, dense):
        g.close()
    b.close()


# ---------------------------------------------------------------------------------------------
# long queries: the k-mer list is cut into segments handled by different waves and recombined
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("num_hash", [1, 3])
def test_long_queries_are_segmented(ka, ctx, oracle, num_hash, monkeypatch):
    rng = np.random.default_rng(77 + num_hash)
    k, L, n_cols = 31, 12, 3000
    image = _make_random_db(rng, L, n_cols, 0.93)
    genome = rand_seq(rng, 20000)
    seqs = [genome, genome[:9000] + "N" + rand_seq(rng, 7000), rand_seq(rng, 100), "", genome[5000:5300]]
    # make column 5 and 2999 contain the whole genome
    for col in (5, 2999):
        for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, num_hash, L).reshape(-1):
            image[r, col // 8] |= np.uint8(1 << (col % 8))
    g = ka.Group(ctx, k, num_hash, L, n_cols)
    g.add_columns(image, n_cols)
    g.finalize()
    b = ka.Batch(ctx, seqs)
    # (0 = the natural choice: few tiles -> segments; "cw" = the persistent count kernel with a pair spread over
    # up to 40 waves instead of the segment slab)
    # (the persistent kernel takes eight k-mers per step with 14 counter planes and more -- here 20)
    for force in (0, 1, 7, 64, "cw"):
        knobs = dict(force_segs=0, count_walk_min_rows=1, count_walk_waves=1500) if force == "cw" else dict(force_segs=force)
        with ctx.tuning(**knobs):
            for threshold in (1.0, 0.97, 0.5):
                thr32 = float(np.float32(threshold))
                for flags in (0, ka.SEARCH_EARLY_EXIT):
                    r = g.search(b, threshold, flags)
                    if force == "cw" and threshold < 1.0 and not flags:
                        assert r.search_kernel.startswith("count_walk_kernel<") and r.search_kernel.endswith(",8>"), r.search_kernel
                    per_q = r.per_query()
                    for i, s in enumerate(seqs):
                        kmers = oracle.unique_kmers(s, k)
                        exp, _ = oracle.search_image(image, image.shape[1], k, num_hash, L, n_cols, kmers, thr32)
                        assert per_q[i] == exp, (force, threshold, flags, i, len(per_q[i]), len(exp))
                    if threshold == 1.0:
                        assert {5, 2999} <= {c for c, _ in per_q[0]}
    b.close()
    g.close()


def test_queries_above_2_pow_20_positions_use_32_plane_counters(ka, ctx, oracle):
    """A query with more than 2^20 k-mer positions (a bacterial chromosome) needs counters wider than 20 bits: the
    32-plane instantiations of the count kernels -- tiled with segments + tree combine, and the persistent form whose
    cut pairs carry 32 KiB of partial counters through the slab -- against the oracle and each other; plus the append-mode
    search (kwage_search_device_append_submit): two searches filling ONE list with a column base each."""
    import ctypes as C
    from kwage_amd.native import check, lib
    rng = np.random.default_rng(2 ** 20)
    k, nh, L, n_cols = 31, 1, 10, 200
    image = _make_random_db(rng, L, n_cols, 0.5)         # random columns count about n/2: below the threshold
    genome = rand_seq(rng, 1_150_000)
    for col in (5, 199):
        for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1):
            image[r, col // 8] |= np.uint8(1 << (col % 8))
    seqs = [genome, rand_seq(rng, 400), genome[1000:1500]]
    g = ka.Group(ctx, k, nh, L, n_cols)
    g.add_columns(image, n_cols)
    g.finalize()
    b = ka.Batch(ctx, seqs)
    thr = 0.9
    exp = [oracle.search_image(image, image.shape[1], k, nh, L, n_cols, oracle.unique_kmers(s, k), float(np.float32(thr)))[0] for s in seqs]
    assert {5, 199} <= {c for c, _ in exp[0]} and max(m for _, m in exp[0]) > 1 << 20
    seen = set()
    for knobs, flags in ((dict(count_walk=0), 0), (dict(count_walk=0), ka.SEARCH_EARLY_EXIT), (dict(count_walk=1, count_walk_min_rows=1), 0),
                         (dict(count_walk=1, count_walk_min_rows=1, count_walk_waves=333), 0)):
        with ctx.tuning(narrow=0, **knobs):
            r = g.search(b, thr, flags)
        seen.add(r.search_kernel)
        assert r.per_query() == exp, (knobs, flags, r.search_kernel)
    assert any(n.startswith("count_walk_kernel<32,1") for n in seen) and any("->32" in n for n in seen), seen

    # append mode: the same batch searched twice into one caller-owned list -- first with a fresh counter and column base
    # 1000, then appended behind it with base 5000; a list that does not fit is counted, not overrun
    import torch
    buf = torch.empty((1 + 4096, 3), dtype=torch.int32, device="cuda:0")
    n, ms, h = C.c_uint64(), C.c_float(), C.c_void_p()
    for base, reset in ((1000, 1), (5000, 0)):
        check(lib().kwage_search_device_append_submit(g._h, b._h, C.c_float(thr), 0, buf.data_ptr() + 12, buf.shape[0] - 1, buf.data_ptr(), base, reset, C.byref(h)))
        check(lib().kwage_search_device_collect(h, C.byref(n), None, C.byref(ms)))
    flat = sorted((q, c, m) for q, e in enumerate(exp) for c, m in e)
    assert n.value == 2 * len(flat)
    host = buf.cpu().numpy().view(np.uint32)
    assert int(host[0, 0]) == n.value
    got = sorted(map(tuple, host[1:1 + n.value].tolist()))
    assert got == sorted([(q, c + 1000, m) for q, c, m in flat] + [(q, c + 5000, m) for q, c, m in flat])
    check(lib().kwage_search_device_append_submit(g._h, b._h, C.c_float(thr), 0, buf.data_ptr() + 12, 3, buf.data_ptr(), 0, 1, C.byref(h)))
    check(lib().kwage_search_device_collect(h, C.byref(n), None, None))
    assert n.value == len(flat) > 3                      # counted in full, three stored
    with pytest.raises(ka.KwageError):                  # a column base that would leave the 32-bit column field
        check(lib().kwage_search_device_append_submit(g._h, b._h, C.c_float(thr), 0, buf.data_ptr() + 12, 3, buf.data_ptr(), 0xFFFFFF00, 1, C.byref(h)))
    with pytest.raises(ka.KwageError):                  # append mode needs the counter word
        check(lib().kwage_search_device_append_submit(g._h, b._h, C.c_float(thr), 0, buf.data_ptr() + 12, 3, None, 0, 1, C.byref(h)))
    b.close()
    g.close()


def test_cli_sharded_over_two_contexts(ka, oracle):
    """KWAGE_DEVICES shards whole files over devices, one host thread + ctx each.  With one GPU on
    the box both contexts sit on device 0, which still exercises the sharding, threading and merge."""
    from kwage_amd import native
    cdir = os.path.join(GOLDEN, "multi")
    for fmt in ("csv", "json"):
        for thr in ("1.0", "0.7"):
            args = [native.KWAGE_BIN, "-d", "dbs", "-i", "reads.fastq", "-i", "contigs.fa.gz", "-t", thr, "--o." + fmt]
            one = subprocess.run(args, cwd=cdir, capture_output=True, env=dict(os.environ, KWAGE_DEVICES="0"))
            two = subprocess.run(args, cwd=cdir, capture_output=True, env=dict(os.environ, KWAGE_DEVICES="0,0"))
            three = subprocess.run(args, cwd=cdir, capture_output=True, env=dict(os.environ, KWAGE_DEVICES="0,0,0"))
            assert one.returncode == 0 and two.returncode == 0 and three.returncode == 0, two.stderr.decode()
            assert one.stdout == two.stdout == three.stdout      # deterministic, independent of the sharding
            # "all": the devices are counted by a short-lived child process, so that the page-cache readers are still
            # forked from a process in which HIP is not up (KWAGE_VERBOSE names the devices on stderr)
            every = subprocess.run(args, cwd=cdir, capture_output=True, env=dict(os.environ, KWAGE_DEVICES="all", KWAGE_CACHE_READER="2"))
            assert every.returncode == 0 and every.stdout == one.stdout, every.stderr.decode()
            # a database larger than HBM is searched in several passes over whole files: force one file per pass
            passes = subprocess.run(args, cwd=cdir, capture_output=True, env=dict(os.environ, KWAGE_MAX_GROUP_BYTES="1"))
            assert passes.returncode == 0 and passes.stdout == one.stdout
            # many small query batches, software-pipelined through the two search slots
            small = subprocess.run(args, cwd=cdir, capture_output=True, env=dict(os.environ, KWAGE_BATCH_BASES="300"))
            assert small.returncode == 0 and small.stdout == one.stdout
            exp = open(os.path.join(cdir, "expected_t%s.%s" % (thr, fmt)), encoding="latin-1").read()
            assert sorted(two.stdout.decode("latin-1").splitlines()) == sorted(exp.splitlines())


def test_mixed_filter_size_groups_five_hashes(ka, ctx, oracle):
    """BASELINE config C5 at test scale: adaptive filter sizes (several log_2_filter_len groups), 5
    hash functions, threshold 0.8 -- every group against the oracle on the bits resident in HBM."""
    from kwage_amd import synth
    base = synth.Workload("c5-test", 0, 0, 31, 5, 24, 400, 0.8, density_q8=194, num_genomes=4, genome_len=2000)
    parts = synth.build_multi(ctx, synth.C5_TEST_GROUPS, base)
    db = ka.Database([p.group for p in parts])
    assert db.num_columns == sum(n for _, n in synth.C5_TEST_GROUPS)
    for threshold in (0.8, 1.0):
        results = db.search(parts[0].batch, threshold)
        for p, r in zip(parts, results):
            w = p.workload
            image = p.group.read_rows(np.arange(1 << w.log_2_filter_len))
            per_q = r.per_query()
            for qi, q in enumerate(p.queries):
                kmers = oracle.unique_kmers(q, w.kmer_len)
                exp, _ = oracle.search_image(image, image.shape[1], w.kmer_len, w.num_hash, w.log_2_filter_len,
                                             w.num_samples, kmers, float(np.float32(threshold)))
                assert per_q[qi] == exp, (w.log_2_filter_len, threshold, qi)
            # planted genomes are found in every group
            for qi, gi in enumerate(p.query_genome):
                if gi >= 0:
                    assert set(p.planted[gi]) <= {c for c, _ in per_q[qi]}
    for p in parts:
        p.batch.close()
    db.close()


def test_cli_error_and_empty_cases(ka, tmp_path):
    """Failure behaviour of the drop-in: what the reference's `main` does when search() throws
    (kwage.cpp:322-333: message on stderr, EXIT_FAILURE; its OpenMP build aborts instead because the
    throw crosses the parallel region) and what it prints when nothing matches."""
    from kwage_amd import native
    good = os.path.join(GOLDEN, "k32", "k32.db")
    q = os.path.join(GOLDEN, "k32", "q.fna")
    raw = bytearray(open(good, "rb").read())
    raw[28] = 1                                         # hash_func = 1: "Unknown hash function" (hash.cpp:92)
    (tmp_path / "badhash.db").write_bytes(raw)
    (tmp_path / "trunc.db").write_bytes(bytes(raw[:28]) + b"\0" + bytes(raw[29:300]))
    (tmp_path / "empty.fa").write_text("")
    (tmp_path / "onlydef.fa").write_text(">x\n")
    run = lambda *a: subprocess.run([native.KWAGE_BIN] + list(a), capture_output=True, text=True)
    r = run("-d", str(tmp_path / "badhash.db"), "-i", q, "--o.csv")
    assert r.returncode == 1 and "Unknown hash function" in r.stderr and "Caught the" in r.stderr
    r = run("-d", good, "-i", str(tmp_path / "missing.fa"), "--o.csv")
    assert r.returncode == 1 and "Error opening: " in r.stderr
    r = run("-d", str(tmp_path / "trunc.db"), "-i", q, "--o.csv")
    assert r.returncode == 1 and r.stdout == ""
    header = "query,num_kmers,num_kmers_found,percent_kmers_found,sample_metadata\n"
    for qf in ("empty.fa", "onlydef.fa"):
        r = run("-d", good, "-i", str(tmp_path / qf), "--o.csv")
        assert r.returncode == 0 and r.stdout == header and "Search complete in" in r.stderr
    r = run("-d", good, "--o.csv", "ACGT")              # too short for k: silently no result (kwage.cpp:369-371)
    assert r.returncode == 0 and r.stdout == header
    r = run("-d", good, "--o.json", "ACGT")
    assert r.returncode == 0 and r.stdout == ""         # JSON: nothing at all when no query matched
    r = run("-d", str(tmp_path / "nonexist.db"), "--o.csv", "ACGT")
    assert r.returncode == 1 and "FindFiles::next: Unable to stat entry" in r.stderr


def test_search_device_leaves_the_same_hits_in_hbm(ka, ctx):
    """kwage_search_device (the multi-GPU entry point: hits stay in a caller-owned device buffer for
    RCCL) returns the same records as kwage_search, including when the buffer must grow."""
    import torch
    from kwage_amd import synth
    from kwage_amd.distributed import device_tensor_search_fn
    s = synth.build(ctx, synth.WORKLOADS["tiny"])
    for thr in (1.0, 0.6, 0.001):
        ref = s.group.search(s.batch, thr)
        fn = device_tensor_search_fn(s.group, 0, "cuda:0", initial_capacity=16)      # forces the grow-and-retry path
        t, nk = fn(s.batch, thr)
        got = t.cpu().numpy().astype(np.uint32)
        order = np.lexsort((got[:, 1], got[:, 0]))
        got = got[order]
        assert len(got) == len(ref.hits)
        assert np.array_equal(got[:, 0], ref.hits["query"]) and np.array_equal(got[:, 1], ref.hits["column"])
        assert np.array_equal(got[:, 2], ref.hits["num_match"])
        assert np.array_equal(nk.cpu().numpy().astype(np.uint32), ref.num_query_kmer)
    s.batch.close()
    s.group.close()


def test_python_file_database_matches_reference_report(ka, ctx, oracle):
    """kwage_amd.FileDatabase (directory -> groups -> hits with accessions) == the reference's CSV."""
    cdir = os.path.join(GOLDEN, "multi")
    db = ka.FileDatabase(ctx, [os.path.join(cdir, "dbs")])
    assert len(db.files) == 4 and len(db.groups) == 3
    seqs = [s for _, s in oracle.read_sequences(os.path.join(cdir, "reads.fastq"))]
    seqs += [s for _, s in oracle.read_sequences(os.path.join(cdir, "contigs.fa.gz"))]
    names = [d for d, _ in oracle.read_sequences(os.path.join(cdir, "reads.fastq"))] + \
            [d for d, _ in oracle.read_sequences(os.path.join(cdir, "contigs.fa.gz"))]
    for thr, fn in ((1.0, "expected_t1.0.csv"), (0.7, "expected_t0.7.csv")):
        exp = oracle.parse_csv(open(os.path.join(cdir, fn), encoding="latin-1").read())
        got = {}
        for h in db.search_sequences(seqs, thr):
            got.setdefault(names[h.query], []).append((h.accession, h.num_query_kmer, h.num_kmers_found))
        assert set(got) == set(exp)
        for q in exp:
            assert sorted(got[q]) == sorted((a, nk, nf) for a, nk, nf, _ in exp[q])
    db.close()


@pytest.mark.parametrize("n_cols", [1, 100, 300, 512, 600, 1024, 2048, 2049, 4096])
def test_narrow_rows_many_queries(ka, ctx, oracle, n_cols):
    """Rows narrower than a wave-load are searched with several queries per wave (and_narrow_kernel):
    ragged query lengths (including empty / too short / all-N), >= 64 queries so that path is taken."""
    rng = np.random.default_rng(n_cols + 5)
    k, nh, L = 31, 2, 10
    image = _make_random_db(rng, L, n_cols, 0.8)
    genome = rand_seq(rng, 3000)
    for col in {0, n_cols // 2, n_cols - 1}:
        for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1):
            image[r, col // 8] |= np.uint8(1 << (col % 8))
    seqs = []
    for i in range(150):
        n = int(rng.choice([0, 10, 31, 32, 60, 150, 400, 1500]))
        if i % 3 == 0 and n >= 31:
            a = int(rng.integers(0, len(genome) - n + 1)); seqs.append(genome[a:a + n])
        elif i % 7 == 0:
            seqs.append("N" * n)
        else:
            seqs.append(rand_seq(rng, n))
    g = ka.Group(ctx, k, nh, L, n_cols)
    g.add_columns(image, n_cols)
    g.finalize()
    b = ka.Batch(ctx, seqs)
    for thr in (1.0, 0.9):
        exp = [oracle.search_image(image, image.shape[1], k, nh, L, n_cols, oracle.unique_kmers(s, k), float(np.float32(thr)))[0] for s in seqs]
        for flags in (0, ka.SEARCH_EARLY_EXIT):
            r = g.search(b, thr, flags)
            assert r.per_query() == exp, (n_cols, thr, flags)
    assert sum(len(e) for e in exp) > 0
    b.close()
    g.close()


def test_c_example_program(ka, oracle, tmp_path):
    """examples/search_example.c (plain C99 against the ABI) finds what the oracle finds."""
    import shutil
    from conftest import ROOT
    from kwage_amd import native
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "search_example")
    libdir = os.path.dirname(native.lib_path())
    subprocess.check_call([gcc, "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "search_example.c"),
                           "-L", libdir, "-lkwage_amd", "-Wl,-rpath," + libdir, "-o", exe])
    pa = os.path.join(GOLDEN, "multi/dbs/a/k31_L10_h1.db")
    pb = os.path.join(GOLDEN, "multi/dbs/a/deeper/k31_L10_h1_b.db")
    seqs = [s for _, s in oracle.read_sequences(os.path.join(GOLDEN, "multi/contigs.fa.gz"))]
    r = subprocess.run([exe, "1.0", pa, pb, "--"] + seqs, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [(int(ln.split()[1]), int(ln.split()[3])) for ln in r.stdout.splitlines() if ln.startswith("query")]
    exp = []
    for qi, s in enumerate(seqs):
        km = oracle.unique_kmers(s, 31)
        for base, p in ((0, pa), (128, pb)):
            db = oracle.read_db(p)
            hits, _ = oracle.search_image(db.rows, db.header.slice_size, 31, 1, 10, db.header.num_filter, km, 1.0)
            exp += [(qi, base + c) for c, _ in hits]
    assert got == exp and len(got) > 0


def test_submit_collect_pipelining(ka, ctx):
    """kwage_search_submit / _collect: two searches in flight on one context give the same results as the
    synchronous call, in any collect order, including hit-buffer growth inside collect; a third submit
    is refused."""
    from kwage_amd import synth
    s = synth.build(ctx, synth.WORKLOADS["tiny"])
    rng = np.random.default_rng(2)
    other = ka.Batch(ctx, [rand_seq(rng, 200) for _ in range(30)] + s.queries[:10])
    ref = {(id(b), t): s.group.search(b, t) for b in (s.batch, other) for t in (1.0, 0.7, 0.001)}
    for t1, t2 in ((1.0, 0.7), (0.001, 1.0), (0.7, 0.001)):
        p1 = s.group.submit(s.batch, t1, ka.SEARCH_TIMING)
        p2 = s.group.submit(other, t2)
        with pytest.raises(ka.KwageError):
            s.group.submit(other, t1)                   # both slots busy
        with pytest.raises(ka.KwageError):
            ka.hash_batch(ctx, 31, 2, 14, other)        # slot 0 is busy
        r2 = p2.collect()
        r1 = p1.collect()
        assert np.array_equal(r1.hits, ref[(id(s.batch), t1)].hits) and r1.search_kernel_ms > 0
        assert np.array_equal(r2.hits, ref[(id(other), t2)].hits)
        assert np.array_equal(r1.num_query_kmer, ref[(id(s.batch), t1)].num_query_kmer)
        with pytest.raises(ka.KwageError):
            p1.collect()
    # a long pipelined stream of submissions
    pend, got = None, []
    for i in range(12):
        nxt = s.group.submit(s.batch if i % 2 == 0 else other, 1.0)
        if pend is not None:
            got.append(pend.collect())
        pend = nxt
    got.append(pend.collect())
    for i, r in enumerate(got):
        assert np.array_equal(r.hits, ref[(id(s.batch if i % 2 == 0 else other), 1.0)].hits)
    other.close()
    s.batch.close()
    s.group.close()


def test_pipelined_device_search_and_counted_exchange(ka, ctx):
    """kwage_search_device_submit/_collect through PipelinedDeviceSearcher: the exchange buffer carries its own
    u64 count (written by the engine in stream order) followed by the records; two searches in flight, buffer
    growth, and the world-1 RCCL exchange_counted() all reproduce kwage_search."""
    import torch
    import torch.distributed as dist
    from kwage_amd import synth
    from kwage_amd.distributed import PipelinedDeviceSearcher, ShardedSearch
    s = synth.build(ctx, synth.WORKLOADS["tiny"])
    ref = {t: s.group.search(s.batch, t) for t in (1.0, 0.6, 0.001)}
    pipe = PipelinedDeviceSearcher(s.group, 0, "cuda:0", initial_capacity=64)       # 0.001 overflows -> grow + redo
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        ss = ShardedSearch(dist, 0, 1, s.group.num_columns, None, device="cuda:0", capacity=32)
        seq = [1.0, 0.6, 0.001, 1.0, 0.001, 0.6]
        tk = pipe.submit(s.batch, seq[0])
        for i, t in enumerate(seq):
            nxt = pipe.submit(s.batch, seq[i + 1]) if i + 1 < len(seq) else None
            buf, n = pipe.collect_counted(tk)
            head = buf[0].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            assert n == len(ref[t].hits) == int(head[0] | (head[1] << 32))
            merged = ss.exchange_counted(buf, n)
            assert np.array_equal(merged[:, 0], ref[t].hits["query"]) and np.array_equal(merged[:, 1], ref[t].hits["column"])
            assert np.array_equal(merged[:, 2], ref[t].hits["num_match"])
            tk = nxt
        assert ss.capacity >= len(ref[0.001].hits) > 64
    finally:
        if own:
            dist.destroy_process_group()
    s.batch.close()
    s.group.close()


@pytest.mark.parametrize("n_cols", [16500, 40000, 131072, 131073, 300000])
def test_walk_rows_many_queries(ka, ctx, oracle, n_cols, request):
    """Rows of >= 3 KiB take and_walk_kernel when the batch is large (a persistent grid, every wave walks an
    equal share of the batch's positions over a column tile's whole width; pairs cut by a share boundary meet
    in memory): ragged query lengths so that shares start and end inside, between and across queries, empty and
    too-short queries in between, hits in the first, middle and last columns; 131073 columns (17 KiB) needs two
    column tiles whose last chunk lies wholly past the row end, 300000 columns three tiles (by default rows wider
    than 16 KiB and batches below 256k rows stay with the tiled kernel: both limits are moved here).  Every case
    with the natural number of waves and with shares of a handful of positions."""
    scope = ctx.tuning(walk_max_kib=64, walk_min_rows=1, walk_waves=0, walk=4)      # (knobs of the module's shared context)
    scope.__enter__()
    request.addfinalizer(lambda: scope.__exit__(None, None, None))
    rng = np.random.default_rng(n_cols)
    k, nh, L = 31, 2, 10
    image = _make_random_db(rng, L, n_cols, 0.9)
    genome = rand_seq(rng, 1200)
    cols = sorted({0, 1, n_cols // 2, n_cols - 130, n_cols - 1})
    for col in cols:
        for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1):
            image[r, col // 8] |= np.uint8(1 << (col % 8))
    seqs = []
    for i in range(960 if n_cols <= 40000 else 384):          # (the wide matrices return millions of records per search: fewer queries)
        n = int(rng.choice([0, 30, 31, 32, 33, 34, 35, 40, 64, 150, 300]))
        if i % 2 == 0 and n >= 31:
            a = int(rng.integers(0, len(genome) - n + 1)); seqs.append(genome[a:a + n])
        elif i % 11 == 0:
            seqs.append("N" * n)
        else:
            seqs.append(rand_seq(rng, n))
    g = ka.Group(ctx, k, nh, L, n_cols)
    g.add_columns(image, n_cols)
    g.finalize()
    b = ka.Batch(ctx, seqs)
    exp = [oracle.search_image(image, image.shape[1], k, nh, L, n_cols, oracle.unique_kmers(s, k), 1.0)[0] for s in seqs]
    first = None
    for waves in ((0, 5, 3001, 16384) if n_cols < 100000 or n_cols == 131073 else (0, 3001)):       # (wide matrices return millions of records per search)
        ctx.set_tuning("walk_waves", waves)
        for flags, refine in ((0, 1), (ka.SEARCH_EARLY_EXIT, 1), (ka.SEARCH_EARLY_EXIT, 0)):
            ctx.set_tuning("ee_refine", refine)      # early exit never takes the walk form: screen + refine, or (knob) the tiled kernel alone
            r = g.search(b, 1.0, flags)
            assert r.search_kernel.startswith(("and_screen_kernel<" if refine else "and_kernel<") if flags else "and_walk_kernel<"), r.search_kernel
            if first is None:
                first = r
                assert r.per_query() == exp, (n_cols, flags, waves)           # against the oracle once ...
            else:                                                            # ... then hit list against hit list (millions of records)
                assert np.array_equal(r.hits, first.hits) and np.array_equal(r.num_query_kmer, first.num_query_kmer), (n_cols, flags, waves)
    # band after band of the matrix (and_band_walk_kernel: rows regrouped by band, parts meet in per-query slots, a finish
    # kernel reports): 2, 5 and 64 bands, the natural number of waves and shares of a handful of rows, twice each (the
    # slots must be left zero)
    if n_cols <= 131072:
        with ctx.tuning(walk_bands_min_gib=0):
            for bands in (2, 5, 64):
                for waves in (0, 3001):
                    ctx.set_tuning("walk_bands", bands)
                    ctx.set_tuning("walk_waves", waves)
                    for flags in (0, 0):
                        r = g.search(b, 1.0, flags)
                        assert r.search_kernel.startswith("and_band_walk_kernel<"), r.search_kernel
                        assert np.array_equal(r.hits, first.hits) and np.array_equal(r.num_query_kmer, first.num_query_kmer), (n_cols, bands, waves, flags)
            ctx.set_tuning("walk_bands", 0)
    ctx.set_tuning("walk_waves", 0)
    ctx.set_tuning("ee_refine", 1)
    ctx.set_tuning("walk", 0)
    r = g.search(b, 1.0, 0)
    assert r.search_kernel.startswith("and_kernel<") and np.array_equal(r.hits, first.hits)
    planted = [e for s, e in zip(seqs, exp) if len(s) >= 31 and s in genome]
    assert planted and all({c for c, _ in e} >= set(cols) for e in planted)
    # The count path's persistent form (count_walk_kernel) on the same ragged batch: shares of a handful of positions,
    # of thousands, and more waves than the chip holds; twice per case (the kernel must leave its pair counters zero);
    # against the tiled count kernel's list, which is checked against the oracle once.
    # (the ragged batch returns millions of records per search -- one-k-mer queries match most columns -- so the wide
    # matrices get fewer variants)
    small = n_cols <= 40000
    for thr in ((0.9, 0.5) if small else (0.9,)):
        with ctx.tuning(count_walk=0):
            ref = g.search(b, thr, 0)
        assert ref.search_kernel.startswith("count_kernel<"), ref.search_kernel
        thr32 = float(np.float32(thr))
        got = ref.per_query() if small else None
        for i in range(0, len(seqs), 16 if small else 160):          # (the oracle's per-bit counting loop over 300 k columns is slow)
            e = oracle.search_image(image, image.shape[1], k, nh, L, n_cols, oracle.unique_kmers(seqs[i], k), thr32)[0]
            if small:
                assert got[i] == e, (n_cols, thr, i)
            else:
                h = ref.hits[ref.hits["query"] == i]
                assert [(int(c), int(m)) for c, m in zip(h["column"], h["num_match"])] == e, (n_cols, thr, i)
        for waves in ((0, 7, 3001, 30000) if small else (0, 3001)):
            with ctx.tuning(count_walk_waves=waves, count_walk_min_rows=1):
                for rep in range(2):
                    r = g.search(b, thr, 0)
                    assert r.search_kernel.startswith("count_walk_kernel<"), r.search_kernel
                    assert np.array_equal(r.hits, ref.hits) and np.array_equal(r.num_query_kmer, ref.num_query_kmer), (n_cols, thr, waves, rep)
    b.close()
    g.close()


@pytest.mark.parametrize("env", [{}, {"KWAGE_LOAD_MMAP": "0"}, {"KWAGE_LOAD_CHUNK_KB": "8", "KWAGE_LOAD_WINDOW_KB": "20"},
                                 {"KWAGE_LOAD_CHUNK_KB": "3", "KWAGE_LOAD_WINDOW_KB": "3"}, {"KWAGE_LOAD_SDMA": "0"},
                                 {"KWAGE_LOAD_SDMA": "0", "KWAGE_LOAD_CHUNK_KB": "8", "KWAGE_LOAD_WINDOW_KB": "20"}, {"KWAGE_LOAD_DIRECT": "1"},
                                 {"KWAGE_LOAD_DIRECT": "1", "KWAGE_LOAD_GANG": "1"}, {"KWAGE_LOAD_DIRECT": "1", "KWAGE_LOAD_GANG": "3", "KWAGE_LOAD_WINDOW_KB": "20"},
                                 {"KWAGE_LOAD_DIRECT": "1", "KWAGE_LOAD_CHUNK_KB": "3", "KWAGE_LOAD_WINDOW_KB": "3"}])
def test_loader_paths_give_the_same_matrix(ka, oracle, tmp_path, env):
    """kwage_group_add_db_file(s): the copy-engine pipeline (file windows locked through HSA, hsa_amd_memory_async_copy
    into three staging buffers, scatter kernels behind the completion signals; the default), the hipHostRegister form it
    replaces (KWAGE_LOAD_SDMA=0), the pread path, and the opt-in direct path (file windows locked through HSA, one copy
    kernel reads up to 16 files over PCIe into the strided matrix; rows that are dword multiples), each with many
    small windows / chunks -- the resident matrix must be the file's rows, for many files of odd and even widths in
    one group.  (The knobs are read once per process.)"""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/oracle")
import kwage_amd as ka, kwage_oracle as oracle
rng = np.random.default_rng(8)
files = []
for j, ncol in enumerate((100, 2048, 13, 777, 800, 96, 2016) + (2048,) * 19 + (64, 2048)):
    rows = rng.integers(0, 256, size=(1 << 12, (ncol + 7) // 8), dtype=np.uint8)
    if ncol %% 8:
        rows[:, -1] &= np.uint8((1 << (ncol %% 8)) - 1)
    infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%%07d" %% (j * 10000 + i))) for i in range(ncol)]
    p = %r + "/f%%d.db" %% j
    oracle.write_db(p, 31, 2, 12, rows, ncol, infos)
    files.append((p, ncol, rows))
with ka.Context(0) as ctx:
    for together in (True, False):       # kwage_group_add_db_files (gangs of up to 16 raw files side by side) and file by file
        g = ka.Group(ctx, 31, 2, 12, sum(((n + 127) // 128) * 128 for _, n, _ in files))
        if together:
            loaded = g.add_db_files([p for p, _, _ in files])
            assert [nf for _, nf in loaded] == [n for _, n, _ in files]
            firsts = [f for f, _ in loaded]
        else:
            firsts = [g.add_db_file(p)[0] for p, _, _ in files]
        g.finalize()
        image = g.read_rows(np.arange(1 << 12))
        for (p, ncol, rows), first in zip(files, firsts):
            assert first %% 128 == 0
            got = image[:, first // 8: first // 8 + rows.shape[1]]
            assert np.array_equal(got, rows), (p, together)
        g.close()
print("ok")
''' % (ROOT, ROOT, str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
