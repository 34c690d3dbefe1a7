"""Pin the CPU oracle (oracle/) against outputs of the REFERENCE ITSELF.

tests/golden/ was produced by tests/golden/make_golden.py with the reference's own code
(oracle/_ref/kwage = reference `kwage`; oracle/_ref/ref_tool = reference BloomFilter /
build_db / bigsi_hash behind a small driver).  Nothing here touches a GPU or /root/reference.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_kat_kmers_and_hashes(oracle):
    """word.h:73-104,161-165 + hash.cpp:176-234, against ref_tool kmers."""
    for case in _load("kat_kmers.json"):
        k, nh, seq = case["k"], case["num_hash"], case["seq"]
        words, pos = oracle.canonical_kmers(seq, k)
        assert len(words) == len(case["kmers"]), (k, seq)
        for w, p, ref in zip(words, pos, case["kmers"]):
            assert int(p) == ref["pos"]
            assert int(w) == int(ref["canon"], 16)
            for h in range(nh):
                assert oracle.hash_word(int(w), k, h) == int(ref["hash"][h], 16)


def test_kat_survey_table(oracle):
    """The vectors quoted in SURVEY.md section 8(a)."""
    assert oracle.murmur3_32(b"abc", 0) == 0xB3DD93FA
    w, _ = oracle.canonical_kmers("ACGTACGTACGTACGTACGTACGTACGTACG", 31)
    assert int(w[0]) == 0x06C6C6C6C6C6C6C6
    assert [oracle.hash_word(int(w[0]), 31, h) for h in range(5)] == \
        [0x7F7B12A3, 0x16D21CD3, 0x1D86C8AB, 0x14DC07FF, 0xDD02E49A]
    w, _ = oracle.canonical_kmers("T" * 31, 31)
    assert int(w[0]) == 0 and oracle.hash_word(0, 31, 0) == 0x30E9726E
    w, _ = oracle.canonical_kmers("ACGTTGCAACGTTGCAACGTTGCAACGTTGCA", 32)
    assert int(w[0]) == 0x1BE41BE41BE41BE4 and oracle.hash_word(int(w[0]), 32, 0) == 0x5BFF87C0
    w, p = oracle.canonical_kmers("ACGTN" + "ACGT" * 8 + "AC", 31)
    assert list(p) == [5, 6, 7, 8]


def test_kat_accession(oracle):
    for case in _load("kat_accession.json"):
        packed = oracle.str_to_accession(case["str"])
        assert packed == int(case["packed"])
        assert oracle.accession_to_str(packed) == case["round_trip"]


def test_threshold_float_semantics(oracle):
    """kwage.cpp:388: float32 multiply then truncation."""
    f32 = np.float32
    for t in (1.0, 0.8, 0.7, 0.5, 0.05, 0.0001, 0.999999):
        for n in (1, 2, 3, 7, 70, 120, 970, 9970, 1 << 20, (1 << 24) + 1):
            assert oracle.query_threshold(t, n) == int(f32(t) * f32(n))


def test_db_format_round_trip(oracle, tmp_path):
    """Own writer -> own reader, and own reader on the reference-written fixture."""
    db = oracle.read_db(os.path.join(GOLDEN, "basic", "db", "basic.db"))
    h = db.header
    assert (h.magic, h.version) == (0x20191025, 2)
    assert (h.kmer_len, h.num_hash, h.log_2_filter_len, h.num_filter) == (31, 3, 12, 100)
    assert h.compression == 0 and h.hash_func == 0
    assert h.info_start == 44 + 4096 * 13
    import zlib
    assert zlib.crc32(db.rows.tobytes()) & 0xFFFFFFFF == h.crc32   # build_db.cpp:307
    infos = [db.info(j) for j in range(h.num_filter)]
    assert infos[0].csv_string() == "SRR0001000"
    assert infos[9].experiment_title == "title 9" and infos[9].date == (10, 10, 2019)
    assert len(infos[9].sample_attributes) == 5
    out = str(tmp_path / "copy.db")
    oracle.write_db(out, h.kmer_len, h.num_hash, h.log_2_filter_len, db.rows, h.num_filter, infos)
    assert open(out, "rb").read() == db.raw     # byte-identical to the reference's file


def _expected_sets(case):
    txt = open(os.path.join(GOLDEN, case["name"], case["expected"]), encoding="latin-1").read()
    return txt


@pytest.mark.parametrize("case", [c for c in _load("manifest.json")["cases"] if c["format"] == "csv"],
                         ids=lambda c: "%s-t%s-%s" % (c["name"], c["threshold"], len(c["db"])))
def test_search_matches_reference_csv(oracle, case):
    """Whole path (k-mers, hash, AND / count, threshold, hit list) == reference kwage --o.csv."""
    cdir = os.path.join(GOLDEN, case["name"])
    exp = oracle.parse_csv(_expected_sets(case))
    got = oracle.run_search([os.path.join(cdir, d) for d in case["db"]],
                            [os.path.join(cdir, q) for q in case["queries"]],
                            case["cmdline"], float(case["threshold"]))
    # expected keys are deflines; ours are "<id>\t<defline>" for file queries
    got_by_name = {}
    for key, hits in got.items():
        name = key.split("\t", 1)[1] if "\t" in key else key
        got_by_name.setdefault(name, []).extend(hits)
    assert set(got_by_name) == set(exp)
    for q in exp:
        e = sorted((acc, nk, nf) for acc, nk, nf, _ in exp[q])
        g = sorted(got_by_name[q])
        assert g == e, q
        for acc, nk, nf, pct in exp[q]:
            assert oracle.csv_percent(nf, nk) == pct


@pytest.mark.parametrize("early_exit", [False, True])
def test_early_exit_does_not_change_results(oracle, early_exit):
    """kwage.cpp:437-483 only saves work."""
    cdir = os.path.join(GOLDEN, "basic")
    a = oracle.run_search([cdir + "/db"], [cdir + "/q.fa"], [], 0.8, early_exit=early_exit)
    b = oracle.run_search([cdir + "/db"], [cdir + "/q.fa"], [], 0.8, early_exit=not early_exit)
    assert a == b


def test_sequence_reader_quirks(oracle, tmp_path):
    """parse_sequence.cpp:72-262."""
    p = tmp_path / "x.fa"
    p.write_text(">  >a b c\nacgt\nNN gt\n\n>second\n>third\nTTTT\n")
    assert oracle.read_sequences(str(p)) == [("a b c", "ACGTNNGT"), ("third", "TTTT")]
    long_def = ">" + "d" * 3000
    p.write_text(long_def + "\nACGT\n>next\nGG\n")
    recs = oracle.read_sequences(str(p))
    # the defline is cut into 2047-byte gzgets chunks and the chunk holding the EOL is dropped
    assert recs[0] == ("d" * 2046, "ACGT") and recs[1] == ("next", "GG")
    q = tmp_path / "x.fastq"
    q.write_text("@r1 desc\nacgtn\n+\nIIIII\n@r2\nGGCC\n+r2\nIIII\n")
    assert oracle.read_sequences(str(q)) == [("r1 desc", "ACGTN"), ("r2", "GGCC")]
    assert oracle.file_type("a.FASTA.GZ") == "fasta" and oracle.file_type("a.fq") == "unknown"
    assert oracle.file_type("a.fasta.fasta") == "unknown"   # first-occurrence rule, file_util.cpp:108-121
