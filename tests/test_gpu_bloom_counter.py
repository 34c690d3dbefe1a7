"""Device counting pass (kwage_bloom_counter_*, counter.hip) vs the sequential restatement of
make_bloom.cpp:76-621 in oracle/: counting filters, candidate bit vectors, num_kmer, chosen parameters and
the `.bloom` file must be identical -- the device schedule has to reproduce an ORDER-DEPENDENT loop."""
import os

import numpy as np
import pytest

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


def read_set(rng, genome_len, n_reads, read_len, noise=0.01):
    """Overlapping reads of one genome (both strands, a few errors, the odd N and lower case)."""
    g = rand_seq(rng, genome_len)
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    for _ in range(n_reads):
        s = int(rng.integers(0, max(1, genome_len - read_len)))
        r = list(g[s:s + read_len])
        for j in np.nonzero(rng.random(len(r)) < noise)[0]:
            r[j] = "ACGTN"[int(rng.integers(0, 5))]
        r = "".join(r)
        if rng.random() < 0.5:
            r = r.translate(comp)[::-1]
        if rng.random() < 0.1:
            r = r.lower()
        reads.append(r)
    return reads


def run_both(ka, ctx, oracle, reads, k, m, logc, max_log2, batches=1):
    from kwage_amd.pipeline import BloomCounter
    ref = oracle.CountingPass(k, m, logc, max_log2)
    for r in reads:
        ref.add(r)
    bc = BloomCounter(ctx, k, m, logc, max_log2)
    step = max(1, (len(reads) + batches - 1) // batches)
    for i in range(0, len(reads), step):
        bc.add([r.encode() for r in reads[i:i + step]])
    return ref, bc


def assert_same_state(ref, bc):
    st = bc.stats()
    assert st.num_valid_kmer == ref.num_valid_kmer
    assert st.num_bp == ref.num_bp
    got, exp = bc.counts(), ref.counts()
    assert np.array_equal(got, exp), "counting filters differ at %s" % np.nonzero(got != exp)[0][:8]
    for h in range(5):
        assert np.array_equal(bc.valid_bits(h), ref.valid_bits(h)), "candidate bits of hash %d differ" % h
    return st


@pytest.mark.parametrize("k,m,logc", [(31, 1, 18), (31, 2, 18), (31, 5, 18), (31, 15, 18), (21, 3, 12), (32, 2, 16), (15, 4, 10)])
def test_read_set_matches_sequential_loop(ka, ctx, oracle, k, m, logc, tmp_path):
    """Read sets with real coverage; small counting filters make different k-mers share counters all the time."""
    rng = np.random.default_rng(k * 100 + m)
    reads = read_set(rng, 20000, 1500, 150) + ["", "ACGT", "N" * 40, rand_seq(rng, k), rand_seq(rng, k - 1)]
    ref, bc = run_both(ka, ctx, oracle, reads, k, m, logc, 20, batches=3)
    st = assert_same_state(ref, bc)
    assert st.max_rounds >= 1 and st.positions > 0
    # finish: parameters, fold, file
    from kwage_amd.native import SampleInfo
    si = SampleInfo()
    si.run_accession = b"SRR0000042"
    si.number_of_bases = ref.num_bp
    out = str(tmp_path / "x.bloom")
    status, prm = bc.finish(0.25, 18, si, out)
    exp = ref.finish(0.25, 18)
    if exp is None:
        assert status == 1 and not os.path.exists(out)
    else:
        assert status == 0 and (prm.log_2_filter_len, prm.num_hash) == exp[0]
        fi = oracle.FilterInfo()
        fi.run_accession = oracle.str_to_accession("SRR0000042")
        fi.number_of_bases = ref.num_bp
        want = str(tmp_path / "want.bloom")
        oracle.write_bloom(want, k, exp[0][0], exp[0][1], fi, exp[1])
        assert open(out, "rb").read() == open(want, "rb").read()
    bc.close()
    ref.close()


@pytest.mark.parametrize("m", [1, 2, 7, 15])
def test_low_complexity_and_tiny_filters(ka, ctx, oracle, m):
    """Homopolymers / short tandem repeats (one k-mer hundreds of times in a row) and a counting filter of 4
    or 16 elements: every occurrence shares counters with every other, so the whole stream is one dependency
    chain -- including the reference's double increment when two hashes hit one element and the 4-bit wrap
    (make_bloom.cpp:587-602) at min_kmer_count 15."""
    rng = np.random.default_rng(m)
    reads = ["A" * 300, "AC" * 200, "ACG" * 150, "T" * 90 + "N" + "T" * 90, rand_seq(rng, 400), "a" * 64 + "c" * 64]
    for logc in (2, 4, 8):
        for k in (5, 31):
            ref, bc = run_both(ka, ctx, oracle, reads, k, m, logc, 12)
            assert_same_state(ref, bc)
            bc.close()
            ref.close()


def test_k1_and_repeated_reads(ka, ctx, oracle):
    rng = np.random.default_rng(3)
    r = rand_seq(rng, 500)
    for k, m, logc in ((1, 3, 6), (2, 15, 6), (31, 4, 14)):
        ref, bc = run_both(ka, ctx, oracle, [r] * 40, k, m, logc, 10, batches=40)
        assert_same_state(ref, bc)
        bc.close()
        ref.close()


def test_many_chunks_and_carry(ka, ctx, oracle):
    """More than one 16 M-position device chunk, with a fragment that straddles the boundary (the k-1 carry)
    and counters shared across chunks."""
    rng = np.random.default_rng(9)
    genome = rand_seq(rng, 3_000_000)
    reads = [genome[i:i + 2_000_000] for i in (0, 500_000, 1_000_000, 250_000, 0, 700_000, 123_456, 999_999, 1_000_000)]
    assert sum(map(len, reads)) > (1 << 24)
    ref, bc = run_both(ka, ctx, oracle, reads, 31, 3, 22, 24, batches=2)
    st = assert_same_state(ref, bc)
    assert st.chunks >= 2
    status, prm = bc.finish(0.25, 18)
    exp = ref.finish(0.25, 18)
    assert status == 0 and (prm.log_2_filter_len, prm.num_hash) == exp[0]
    bc.close()
    ref.close()


def test_invalid_outcomes_and_errors(ka, ctx, oracle):
    from kwage_amd.pipeline import BloomCounter
    rng = np.random.default_rng(4)
    # no k-mer reaches the count -> optimal_bloom_param throws "No kmers found" -> STATUS_BLOOM_INVALID
    bc = BloomCounter(ctx, 31, 5, 18, 20)
    bc.add([rand_seq(rng, 1000).encode()])
    assert bc.finish(0.25, 18)[0] == 1 and bc.stats().num_valid_kmer == 0
    bc.close()
    # bound not satisfiable within max_log_2_filter_len
    reads = [rand_seq(rng, 60000)]
    ref, bc = run_both(ka, ctx, oracle, reads, 31, 1, 20, 12)
    assert_same_state(ref, bc)
    assert ref.finish(0.01, 10) is None and bc.finish(0.01, 10)[0] == 1
    bc.close()
    ref.close()
    for bad in ((0, 1, 18, 20), (33, 1, 18, 20), (31, 0, 18, 20), (31, 16, 18, 20), (31, 1, 33, 20), (31, 1, 18, 33), (31, 1, 18, 4)):
        with pytest.raises(ka.KwageError):
            BloomCounter(ctx, *bad)
    assert ka.native.lib().kwage_counting_filter_log2(0) == 32
    for n in (1, 1000, 49_000, 50_000, 10**6, 10**8, 4 * 10**8, 5 * 10**8, 10**10):
        assert ka.native.lib().kwage_counting_filter_log2(n) == oracle.counting_filter_log2(n)
    import ctypes as C
    for p, lo, hi in ((0.25, 18, 32), (0.25, 18, 20), (0.01, 10, 12), (0.5, 5, 8)):
        assert ka.native.lib().kwage_approximate_max_kmers(C.c_float(p), lo, hi) == oracle.approximate_max_kmers(p, lo, hi)


def test_read_sets_to_db_to_search(ka, oracle, tmp_path):
    """FASTQ read sets -> countbloom (dbtool and pipeline) -> `.db` -> both this repo's kwage and the REFERENCE
    binary find each sample by a window of its genome; a window with a sequencing error seen once is not in
    (that is what the minimum k-mer count is for)."""
    import subprocess
    from kwage_amd import native, pipeline
    rng = np.random.default_rng(21)
    samples, genomes = [], {}
    for j in range(3):
        g = rand_seq(rng, 6000)
        acc = "SRR%06d" % (900 + j)
        p = tmp_path / (acc + ".fastq")
        with open(p, "w") as f:
            comp = str.maketrans("ACGT", "TGCA")
            for i in range(1200):                                         # ~30x coverage, both strands
                s = int(rng.integers(0, 6000 - 150))
                r = g[s:s + 150]
                if rng.random() < 0.5:
                    r = r.translate(comp)[::-1]
                f.write("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
            # a contaminant seen exactly once: below min_kmer_count 3
            lone = rand_seq(rng, 150)
            f.write("@lone\n%s\n+\n%s\n" % (lone, "I" * 150))
        samples.append((acc, str(p)))
        genomes[acc] = (g, lone)
    # 1. the command-line tool on sample 0; must equal the sequential restatement byte for byte
    acc0, path0 = samples[0]
    out0 = str(tmp_path / "tool.bloom")
    r = subprocess.run([native.KWAGE_DBTOOL_BIN, "countbloom", out0, acc0, "31", "3", path0, "-l", "14", "-L", "22"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    recs = oracle.read_sequences(path0)
    num_bp = sum(len(s) for _, s in recs)
    ref = oracle.CountingPass(31, 3, oracle.counting_filter_log2(num_bp), 22)
    for _, s in recs:
        ref.add(s)
    (lg, nh), bits = ref.finish(0.25, 14)
    (k_, lg_, nh_, hf_), crc, fi, got_bits = oracle.read_bloom(out0)
    assert (k_, lg_, nh_, hf_) == (31, lg, nh, 0) and np.array_equal(got_bits, bits)
    assert fi.csv_string() == acc0 and fi.number_of_bases == num_bp and fi.number_of_spots == len(recs)
    ref.close()
    # 2. pipeline -> .db -> search
    os.makedirs(tmp_path / "w")
    with ka.Context(0) as ctx:
        dbs = pipeline.build_databases(ctx, samples, str(tmp_path / "out"), kmer_len=31, false_positive=0.25,
                                       min_log_2_filter_len=14, max_log_2_filter_len=22, work_dir=str(tmp_path / "w"),
                                       min_kmer_count=3)
    assert sum(oracle.read_db(d).header.num_filter for d in dbs) == 3
    q = tmp_path / "q.fa"
    q.write_text("".join(">%s\n%s\n>%s_lone\n%s\n" % (acc, g[2000:3000], acc, lone) for acc, (g, lone) in genomes.items()))
    dbdir = tmp_path / "dbs"
    os.makedirs(dbdir)
    for d in dbs:
        os.rename(d, dbdir / os.path.basename(d))
    out = subprocess.run([native.KWAGE_BIN, "-d", str(dbdir), "-i", str(q), "--o.csv", "-t", "0.9"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rep = oracle.parse_csv(out.stdout)
    for acc in genomes:
        assert any(a == acc and nf >= 0.95 * nk for a, nk, nf, _ in rep[acc]), rep.get(acc)
        assert not any(a == acc for a, nk, nf, _ in rep.get(acc + "_lone", []))
    if os.access(oracle.REF_KWAGE, os.X_OK):
        refout = subprocess.run([oracle.REF_KWAGE, "-d", str(dbdir), "-i", str(q), "--o.csv", "-t", "0.9"], capture_output=True,
                                text=True, env=dict(os.environ, OMP_NUM_THREADS="1"))
        assert refout.returncode == 0 and sorted(refout.stdout.splitlines()) == sorted(out.stdout.splitlines())


def test_reset_reuses_the_object(ka, ctx, oracle):
    """kwage_bloom_counter_reset: several samples through one object (smaller counting filters allowed, larger
    refused) give what fresh objects give."""
    from kwage_amd.pipeline import BloomCounter
    rng = np.random.default_rng(77)
    bc = BloomCounter(ctx, 31, 3, 20, 20)
    for m, logc in ((3, 20), (2, 18), (5, 19), (1, 12)):
        reads = read_set(rng, 8000, 400, 150)
        bc.reset(m, logc)
        bc.add([r.encode() for r in reads])
        ref = oracle.CountingPass(31, m, logc, 20)
        for r in reads:
            ref.add(r)
        assert_same_state(ref, bc)
        exp = ref.finish(0.25, 14)
        status, prm = bc.finish(0.25, 14)
        assert (status == 1) == (exp is None) and (exp is None or (prm.log_2_filter_len, prm.num_hash) == exp[0])
        ref.close()
    with pytest.raises(ka.KwageError):
        bc.reset(3, 21)
    bc.close()
