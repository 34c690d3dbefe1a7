"""Bloom construction from sequences (the min_kmer_count == 1 case) and the
parameter choice, pinned by files the REFERENCE wrote (tests/golden/bloomgen: its BloomFilter +
binary_write through oracle/_ref/ref_tool) and by its optimal_bloom_param (kat_optimal_bloom_param.json)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

from kwage_amd import native


def test_optimal_bloom_param_matches_reference(oracle):
    L = native.lib()
    for case in json.load(open(os.path.join(GOLDEN, "kat_optimal_bloom_param.json"))):
        exp = None if case["result"] == ["throw"] else (int(case["result"][0]), int(case["result"][1]))
        assert oracle.optimal_bloom_param(case["num_kmer"], float(case["p"]), case["min"], case["max"]) == exp, case
        out = native.Params()
        rc = L.kwage_optimal_bloom_param(31, case["num_kmer"], C.c_float(float(case["p"])), case["min"], case["max"], C.byref(out))
        if exp is None:
            assert rc != 0 and b"Unable to satisfy" in L.kwage_last_error()
        else:
            assert rc == 0 and (out.log_2_filter_len, out.num_hash, out.kmer_len, out.hash_func) == (exp[0], exp[1], 31, 0)
    out = native.Params()
    assert L.kwage_optimal_bloom_param(31, 0, C.c_float(0.25), 18, 32, C.byref(out)) != 0      # "No kmers found"


def _samples():
    return json.load(open(os.path.join(GOLDEN, "bloomgen", "samples.json")))


def test_oracle_bloom_bits_match_reference_files(oracle):
    spec = _samples()
    for j, fl in enumerate(spec["samples"]):
        prm, crc, fi, bits = oracle.read_bloom(os.path.join(GOLDEN, "bloomgen", "bloom", "f%06d.bloom" % j))
        assert prm == (spec["kmer_len"], spec["log_2_filter_len"], spec["num_hash"], 0)
        got = oracle.bloom_bits_from_sequences(fl["seqs"], spec["kmer_len"], spec["num_hash"], spec["log_2_filter_len"])
        assert np.array_equal(got, bits), j


def _sample_info(fl):
    meta = fl.get("meta", {})
    attrs = fl.get("attrs", [])
    tags = (C.c_char_p * max(len(attrs), 1))(*[a[0].encode() for a in attrs])
    vals = (C.c_char_p * max(len(attrs), 1))(*[a[1].encode() for a in attrs])
    si = native.SampleInfo()
    si.run_accession = fl["acc"].encode()
    for k in ("experiment_accession", "sample_accession", "study_accession", "experiment_title", "sample_taxa", "study_title"):
        if k in meta:
            setattr(si, k, meta[k].encode())
    si.attribute_tags, si.attribute_values, si.num_attributes = tags, vals, len(attrs)
    if "n" in fl:
        si.number_of_spots, si.number_of_bases = fl["n"]
    if "date" in fl:
        y, m, d = fl["date"][:10].split("-")
        si.year, si.month, si.day = int(y), int(m), int(d)
    return si, (tags, vals)


@pytest.mark.gpu
def test_device_make_bloom_is_byte_identical_to_reference(tmp_path):
    import kwage_amd as ka
    spec = _samples()
    L = native.lib()
    prm = native.Params(spec["kmer_len"], spec["num_hash"], spec["log_2_filter_len"], 0)
    with ka.Context(0) as ctx:
        outs = []
        for j, fl in enumerate(spec["samples"]):
            seqs = [s.encode() for s in fl["seqs"]]
            offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(s) for s in seqs])
            si, keep = _sample_info(fl)
            out = str(tmp_path / ("f%06d.bloom" % j))
            nd = C.c_uint64()
            native.check(L.kwage_make_bloom(ctx._h, C.byref(prm), b"".join(seqs), offs.ctypes.data, len(seqs), C.byref(si),
                                            out.encode(), C.byref(nd)))
            ref = open(os.path.join(GOLDEN, "bloomgen", "bloom", "f%06d.bloom" % j), "rb").read()
            assert open(out, "rb").read() == ref, j
            outs.append(out)
        # ... and the whole chain FASTA -> .bloom -> .db on the device reproduces the reference's database
        from test_builder import _device_build
        db = str(tmp_path / "bloomgen.db")
        _device_build(ka, ctx, outs, (spec["kmer_len"], spec["num_hash"], spec["log_2_filter_len"], 0), db)
        assert open(db, "rb").read() == open(os.path.join(GOLDEN, "bloomgen", "bloomgen.db"), "rb").read()


@pytest.mark.gpu
def test_distinct_kmer_count_and_long_sequences(oracle):
    import kwage_amd as ka
    rng = np.random.default_rng(8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    chrom = acgt[rng.integers(0, 4, size=300_000)].tobytes().decode()
    seqs = [chrom, chrom[1000:5000], "ACGT" * 50, chrom[::-1].translate(str.maketrans("ACGT", "TGCA"))[:40000], "N" * 100, ""]
    with ka.Context(0) as ctx:
        b = ka.Batch(ctx, seqs)
        cnt = C.c_uint64()
        native.check(native.lib().kwage_count_distinct_kmers(ctx._h, b._h, 31, C.byref(cnt)))
        b.close()
        exp = len(np.unique(np.concatenate([oracle.unique_kmers(s, 31) for s in seqs if len(s) >= 31])))
        assert cnt.value == exp
        # bits through make_bloom's internal chunking of the 300 kb sequence
        prm = native.Params(31, 3, 22, 0)
        bits = np.zeros((1 << 22) // 8, dtype=np.uint8)
        b = ka.Batch(ctx, seqs)
        nd = C.c_uint64()
        native.check(native.lib().kwage_bloom_bits_from_batch(ctx._h, C.byref(prm), b._h, bits.ctypes.data, C.byref(nd)))
        b.close()
        assert nd.value == exp
        assert np.array_equal(bits, oracle.bloom_bits_from_sequences(seqs, 31, 3, 22))


@pytest.mark.gpu
@pytest.mark.parametrize("lg", [31, 32, 33])
def test_distinct_set_with_2_pow_32_slots_or_more(oracle, lg, monkeypatch):
    """A sample above ~1 G k-mer positions (a human or plant assembly) gets ONE shared distinct set of 2^32 slots or
    more; the probe arithmetic must be 64-bit there (a 32-bit mask wraps to 0 or 1 and the k-mer stage never
    returns).  The shared_table_log2 knob forces such a table (34 / 69 GB of HBM) under a small input."""
    import kwage_amd as ka
    rng = np.random.default_rng(lg)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    chrom = acgt[rng.integers(0, 4, size=50_000)].tobytes().decode()
    seqs = [chrom, chrom[100:9000], "ACGT" * 30]
    exp = len(np.unique(np.concatenate([oracle.unique_kmers(s, 31) for s in seqs])))
    with ka.Context(0) as ctx:
        ctx.set_tuning("shared_table_log2", lg)
        free, _ = ctx.mem_info()
        if free < (8 << lg) + (4 << 30):
            pytest.skip("not enough free HBM for a 2^%d-slot table" % lg)
        b = ka.Batch(ctx, seqs)
        cnt = C.c_uint64()
        native.check(native.lib().kwage_count_distinct_kmers(ctx._h, b._h, 31, C.byref(cnt)))
        assert cnt.value == exp
        prm = native.Params(31, 2, 20, 0)
        bits = np.zeros((1 << 20) // 8, dtype=np.uint8)
        nd = C.c_uint64()
        native.check(native.lib().kwage_bloom_bits_from_batch(ctx._h, C.byref(prm), b._h, bits.ctypes.data, C.byref(nd)))
        b.close()
        assert nd.value == exp
        assert np.array_equal(bits, oracle.bloom_bits_from_sequences(seqs, 31, 2, 20))


@pytest.mark.gpu
def test_pipeline_fasta_to_db_to_search(oracle, tmp_path):
    """FASTA per sample -> device Bloom filters -> device-built `.db` files -> the drop-in CLI finds each
    sample by its own sequence; the REFERENCE binary (when present) agrees on the same files."""
    import subprocess
    import kwage_amd as ka
    from kwage_amd import pipeline
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    samples, genomes = [], {}
    for j, n in enumerate((3000, 3500, 200_000, 2500, 250_000)):          # different sizes -> several parameter groups
        g = acgt[rng.integers(0, 4, size=n)].tobytes().decode()
        acc = "SRR%06d" % (j + 1)
        p = tmp_path / (acc + ".fasta")
        p.write_text(">contig1\n%s\n>contig2\n%s\n" % (g[: n // 2], g[n // 2:]))
        samples.append((acc, str(p)))
        genomes[acc] = g
    with ka.Context(0) as ctx:
        dbs = pipeline.build_databases(ctx, samples, str(tmp_path / "out"), kmer_len=31, false_positive=0.25,
                                       min_log_2_filter_len=14, max_log_2_filter_len=24, work_dir=str(tmp_path))
    hdrs = [oracle.read_db(d).header for d in dbs]
    assert sum(h.num_filter for h in hdrs) == 5 and len({(h.log_2_filter_len, h.num_hash) for h in hdrs}) == len(dbs) >= 2
    for h, d in zip(hdrs, dbs):            # every file carries the parameters optimal_bloom_param picks for its samples
        for j in range(h.num_filter):
            acc = oracle.read_db(d).info(j).csv_string()
            n = len(np.unique(np.concatenate([oracle.unique_kmers(g, 31) for g in (genomes[acc][: len(genomes[acc]) // 2], genomes[acc][len(genomes[acc]) // 2:])])))
            assert oracle.optimal_bloom_param(n, 0.25, 14, 24) == (h.log_2_filter_len, h.num_hash)
    q = tmp_path / "q.fa"
    q.write_text("".join(">%s\n%s\n" % (acc, g[100:1100]) for acc, g in genomes.items()))
    out = subprocess.run([native.KWAGE_BIN, "-d", str(tmp_path), "-i", str(q), "--o.csv"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rep = oracle.parse_csv(out.stdout)
    for acc in genomes:
        assert (acc, 970, 970) in [(a, nk, nf) for a, nk, nf, _ in rep[acc]]      # every sample finds itself
    if os.access(oracle.REF_KWAGE, os.X_OK):
        ref = subprocess.run([oracle.REF_KWAGE, "-d", str(tmp_path), "-i", str(q), "--o.csv"], capture_output=True, text=True,
                             env=dict(os.environ, OMP_NUM_THREADS="1"))
        assert ref.returncode == 0 and sorted(ref.stdout.splitlines()) == sorted(out.stdout.splitlines())
