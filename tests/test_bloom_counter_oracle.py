"""The min-count Bloom construction restatement (oracle/kwage_oracle.c, make_bloom.cpp:76-621) is PARITY
UNPINNED against a running reference (make_bloom.cpp needs the NCBI SDK).  What can be checked on the CPU:
  * a second, independent restatement (pure Python, written from the same source lines) agrees with the C one;
  * where no two k-mers share a counter the result must equal the exact k-mer set, whose bits ARE pinned
    (bloom_bits_from_sequences is checked against `.bloom` files written by the reference's own code);
  * the closed-form helpers against direct evaluation."""
import math

import numpy as np
import pytest


def py_count_words(state, seq, k, m, logc, max_log2, oracle):
    """count_words, make_bloom.cpp:506-621, one fragment."""
    count, valid = state["count"], state["valid"]
    cmask, smask = (1 << logc) - 1, (1 << max_log2) - 1
    words, _ = oracle.canonical_kmers(seq, k)
    for w in words:
        h = [oracle.hash_word(int(w), k, s) for s in range(5)]
        c = [x & cmask for x in h[:4]]
        f0, f1 = count[c[0]] & 15, count[c[1]] & 15
        s0, s1 = count[c[2]] >> 4, count[c[3]] >> 4
        mn = min(f0, f1, s0, s1)
        if mn < m:
            if mn == m - 1:
                state["num"] += 1
                for s in range(5):
                    b = h[s] & smask
                    valid[s][b >> 3] |= 1 << (b & 7)
            if f0 == mn:
                count[c[0]] = (count[c[0]] & 0xF0) | ((count[c[0]] + 1) & 15)
            if f1 == mn:
                count[c[1]] = (count[c[1]] & 0xF0) | ((count[c[1]] + 1) & 15)
            if s0 == mn:
                count[c[2]] = (count[c[2]] & 0x0F) | ((((count[c[2]] >> 4) + 1) & 15) << 4)
            if s1 == mn:
                count[c[3]] = (count[c[3]] & 0x0F) | ((((count[c[3]] >> 4) + 1) & 15) << 4)


@pytest.mark.parametrize("k,m,logc", [(31, 1, 10), (31, 3, 6), (11, 15, 2), (5, 2, 4), (32, 5, 8)])
def test_c_restatement_equals_python_restatement(oracle, k, m, logc):
    rng = np.random.default_rng(k + m)
    g = "".join(rng.choice(list("ACGT"), size=1500))
    reads = [g[i:i + 200] for i in rng.integers(0, 1300, size=40)] + ["A" * 100, "acgtn" * 30, "", "AC" * 60]
    max_log2 = 12
    c = oracle.CountingPass(k, m, logc, max_log2)
    st = {"count": [0] * (1 << logc), "valid": [[0] * ((1 << max_log2) // 8) for _ in range(5)], "num": 0}
    for r in reads:
        c.add(r)
        py_count_words(st, r, k, m, logc, max_log2, oracle)
    assert c.num_valid_kmer == st["num"] and c.num_bp == sum(map(len, reads))
    assert c.counts().tolist() == st["count"]
    for h in range(5):
        assert c.valid_bits(h).tolist() == st["valid"][h]
    c.close()


def test_without_shared_counters_it_is_the_exact_kmer_set(oracle):
    """min_kmer_count 1, counting filters far larger than the k-mer set: no k-mer is skipped, so the folded
    filter equals the exact-set bits (pinned against the reference's BloomFilter files) for the parameters
    optimal_bloom_param (pinned) chooses."""
    rng = np.random.default_rng(12)
    seqs = ["".join(rng.choice(list("ACGT"), size=3000)) for _ in range(3)]
    c = oracle.CountingPass(31, 1, 26, 20)
    for s in seqs:
        c.add(s)
    distinct = len(np.unique(np.concatenate([oracle.unique_kmers(s, 31) for s in seqs])))
    if int(np.count_nonzero(c.counts())) == 4 * distinct:          # no two k-mers shared a counter
        assert c.num_valid_kmer == distinct
    (lg, nh), bits = c.finish(0.25, 18)
    assert (lg, nh) == oracle.optimal_bloom_param(c.num_valid_kmer, 0.25, 18, 20)
    if c.num_valid_kmer == distinct:
        assert np.array_equal(bits, oracle.bloom_bits_from_sequences(seqs, 31, nh, lg))
    c.close()


def test_min_count_selects_repeated_kmers(oracle):
    """A k-mer seen m times is in, one seen m-1 times is out (large filters, no collisions expected)."""
    rng = np.random.default_rng(5)
    a = "".join(rng.choice(list("ACGT"), size=400))
    b = "".join(rng.choice(list("ACGT"), size=400))
    for m in (2, 3, 5):
        c = oracle.CountingPass(31, m, 24, 18)
        for _ in range(m):
            c.add(a)
        for _ in range(m - 1):
            c.add(b)
        (lg, nh), bits = c.finish(0.25, 18)
        assert c.num_valid_kmer == len(oracle.unique_kmers(a, 31))
        assert np.array_equal(bits, oracle.bloom_bits_from_sequences([a], 31, nh, lg))
        c.close()


def test_counting_filter_length_rule(oracle):
    """make_bloom.cpp:105-130 evaluated directly."""
    assert oracle.counting_filter_log2(0) == 32
    for n in (1, 10, 49_000, 50_000, 123_456, 10**6, 10**7, 10**8, 3 * 10**8, 4 * 10**8, 5 * 10**8, 10**9, 10**11):
        length = 1.0 / (1.0 - math.pow(1.0 - math.pow(1.0e-2, 0.25), 1.0 / (2 * n)))
        lg = min(32, max(18, math.ceil(math.log(length) / math.log(2.0))))
        assert oracle.counting_filter_log2(n) == lg, n


def test_approximate_max_kmers(oracle):
    """bloom.cpp:72-121: smallest 2^j for which optimal_bloom_param (pinned) finds nothing; one below must work."""
    for p, lo, hi in ((0.25, 18, 32), (0.25, 18, 20), (0.01, 10, 12), (0.5, 5, 8)):
        n = oracle.approximate_max_kmers(p, lo, hi)
        assert n & (n - 1) == 0
        assert oracle.optimal_bloom_param(n, p, lo, hi) is None
        assert oracle.optimal_bloom_param(n // 2, p, lo, hi) is not None
