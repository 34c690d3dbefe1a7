"""CPU-only checks of the product's C ABI: the library loads, exports every declared symbol,
its host-side helpers (db header / metadata / sequence reader / accession codec / threshold)
agree with the oracle and with the reference's golden outputs, and compute entry points fail
LOUDLY without a GPU (no fallback).  No kernel is launched here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

import kwage_amd
from kwage_amd import native


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(native.lib_path()):
        native.build_native()
    return native.lib()


def test_library_exports_every_declared_symbol(L):
    header = open(os.path.join(ROOT, "include", "kwage_amd.h")).read()
    declared = set(re.findall(r"\b(kwage_[a-z0-9_]+)\s*\(", header))
    declared -= {"kwage_amd"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "libkwage_amd.so does not export %s" % name
    assert declared == set(native.EXPORTED_SYMBOLS)
    assert L.kwage_abi_version() == 1


def test_no_cpu_fallback(L):
    """Without a device the engine must refuse, not fall back."""
    if L.kwage_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(kwage_amd.KwageError) as e:
        kwage_amd.Context(0)
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under kwage_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kwage_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".h", ".hip")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, fn)


def test_db_header(L, oracle):
    for rel in ("basic/db/basic.db", "multi/dbs/b/k15_L11_h2.DB", "k32/k32.db"):
        path = os.path.join(GOLDEN, rel)
        h = native.DbHeader()
        native.check(L.kwage_db_read_header(path.encode(), C.byref(h)))
        o = oracle.read_db(path).header
        for f, _ in native.DbHeader._fields_:
            assert getattr(h, f) == getattr(o, f), (rel, f)
    h = native.DbHeader()
    assert L.kwage_db_read_header(os.path.join(GOLDEN, "multi/dbs/not_a_db.txt").encode(), C.byref(h)) != 0
    assert L.kwage_db_read_header(b"/nonexistent.db", C.byref(h)) != 0
    assert b"Unable to open database file" in L.kwage_last_error()


def test_dbinfo_strings_match_reference_json(L, oracle):
    """FilterInfo::csv_string / json_string (bloom.cpp:124-326) against the reference's own JSON."""
    path = os.path.join(GOLDEN, "basic", "db", "basic.db")
    d = C.c_void_p()
    native.check(L.kwage_dbinfo_open(path.encode(), C.byref(d)))
    try:
        assert L.kwage_dbinfo_num_filter(d) == 100
        db = oracle.read_db(path)
        buf = C.create_string_buffer(64)
        for j in range(100):
            native.check(L.kwage_dbinfo_csv_string(d, j, buf, 64))
            assert buf.value.decode() == db.info(j).csv_string()
        # column 9 carries every metadata field; its block appears verbatim in the reference JSON
        prefix = b"\t\t\t\t\t"
        n = L.kwage_dbinfo_json_string(d, 9, prefix, None, 0)
        big = C.create_string_buffer(n + 1)
        assert L.kwage_dbinfo_json_string(d, 9, prefix, big, n + 1) == n
        ref_json = open(os.path.join(GOLDEN, "basic", "expected_t1.0.json"), encoding="latin-1").read()
        assert big.value.decode("latin-1") in ref_json
        assert L.kwage_dbinfo_csv_string(d, 100, buf, 64) != 0     # out of range -> error, not garbage
    finally:
        L.kwage_dbinfo_close(d)


def test_accession_codec(L):
    for case in json.load(open(os.path.join(GOLDEN, "kat_accession.json"))):
        v = C.c_uint64()
        native.check(L.kwage_str_to_accession(case["str"].encode(), C.byref(v)))
        assert v.value == int(case["packed"])
        buf = C.create_string_buffer(32)
        native.check(L.kwage_accession_to_str(v.value, buf, 32))
        assert buf.value.decode() == case["round_trip"]
    v = C.c_uint64()
    assert L.kwage_str_to_accession(b"SR1", C.byref(v)) != 0          # two letters: the reference throws
    assert L.kwage_str_to_accession(b"SRR", C.byref(v)) != 0          # no digits


def _read_all(L, path):
    f = C.c_void_p()
    native.check(L.kwage_seqfile_open(path.encode(), C.byref(f)))
    out = []
    try:
        d, s, n = C.c_char_p(), C.c_char_p(), C.c_uint64()
        while True:
            r = L.kwage_seqfile_next(f, C.byref(d), C.byref(s), C.byref(n))
            if r < 0:
                raise kwage_amd.KwageError(r, L.kwage_last_error().decode())
            if r == 0:
                break
            assert len(s.value) == n.value
            out.append((d.value.decode("latin-1"), s.value.decode("latin-1")))
    finally:
        L.kwage_seqfile_close(f)
    return out


def test_sequence_reader_matches_oracle(L, oracle, tmp_path):
    for rel in ("basic/q.fa", "multi/reads.fastq", "multi/contigs.fa.gz", "k32/q.fna"):
        path = os.path.join(GOLDEN, rel)
        assert _read_all(L, path) == oracle.read_sequences(path), rel
    p = tmp_path / "quirks.fasta"
    p.write_text(">  >a b c\nacgt\nNN gt\r\n\n>second\n>third\nTTTT\n>" + "d" * 5000 + "\nAC\n  GT  \n>tail")
    assert _read_all(L, str(p)) == oracle.read_sequences(str(p))
    assert _read_all(L, str(p))[0] == ("a b c", "ACGTNNGT")
    q = tmp_path / "bad.fastq"
    q.write_text("@r1\nACGT\n+\n")     # quality line missing
    with pytest.raises(kwage_amd.KwageError):
        _read_all(L, str(q))
    f = C.c_void_p()
    assert L.kwage_seqfile_open(str(tmp_path / "x.txt").encode(), C.byref(f)) != 0    # unknown extension


def test_query_threshold(L, oracle):
    for t in (1.0, 0.8, 0.7, 0.5, 0.05, 0.0001):
        for n in (1, 3, 70, 120, 970, 9970, (1 << 24) + 1):
            assert L.kwage_query_threshold(C.c_float(t), n) == oracle.query_threshold(t, n)


def test_sort_hits_orders_by_query_then_column(L):
    """kwage_sort_hits (host, no device): every size around the switch from std::sort to the radix sort, keys that
    differ in one digit only, values at both ends of 32 bits, and sizes on both sides of the switch to the multi-threaded
    passes (2^20 records); against numpy's lexsort, the payload carried along."""
    rng = np.random.default_rng(12)
    for n in (0, 1, 2, 255, 256, 257, 5000, 300_000, (1 << 20) - 1, (1 << 20) + 12345):
        for spread in ("wide", "narrow", "extreme"):
            if spread == "wide":
                q, c = rng.integers(0, 1 << 20, n), rng.integers(0, 1 << 17, n)
            elif spread == "narrow":
                q, c = rng.integers(7, 9, n), rng.integers(0, 3, n) << 11        # one varying digit in each field
            else:
                q, c = rng.choice([0, 1, 0xFFFFFFFE, 0xFFFFFFFF], n), rng.choice([0, 0x7FF, 0x800, 0xFFFFFFFF], n)
            h = np.stack([q, c, np.arange(n)], axis=1).astype(np.uint32)
            want = h[np.lexsort((np.arange(n), h[:, 1], h[:, 0]))]
            got = np.ascontiguousarray(h.copy())
            L.kwage_sort_hits(got.ctypes.data, n)
            assert np.array_equal(got[:, :2], want[:, :2]), (n, spread)
            # equal keys may come in any order: compare the payloads as multisets per key
            if n <= 300_000:
                assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, want.tolist())), (n, spread)
            else:         # (a million Python tuples are slow) the radix sort is stable: payloads keep their order within a key
                assert np.array_equal(got, want), (n, spread)
    L.kwage_sort_hits(None, 0)


def test_cli_usage_and_validation_without_gpu():
    """Option handling mirrors options.cpp:39-192 and needs no device."""
    import subprocess
    exe = native.KWAGE_BIN
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage for KWAGE (v. 0.4d):" in r.stderr and r.stdout == ""
    r = subprocess.run([exe, "-d", os.path.join(GOLDEN, "multi", "dbs", "not_a_db.txt"), "ACGT"], capture_output=True, text=True)
    assert "Please provide at least one database file to search (-d)" in r.stderr
    db = os.path.join(GOLDEN, "k32", "k32.db")
    r = subprocess.run([exe, "-d", db], capture_output=True, text=True)
    assert "Please provide at least one query sequence or file" in r.stderr
    r = subprocess.run([exe, "-d", db, "-i", "reads.txt"], capture_output=True, text=True)
    assert "does not have an allowed file extension" in r.stderr
    r = subprocess.run([exe, "-d", db, "-t", "1.5", "ACGT"], capture_output=True, text=True)
    assert "Please provide: 0.0 < search threshold <= 1.0" in r.stderr
    r = subprocess.run([exe, "-d", db, "-t", "0", "ACGT"], capture_output=True, text=True)
    assert "Please provide: 0.0 < search threshold <= 1.0" in r.stderr


def test_cli_option_handling_matches_the_reference_binary(oracle):
    """Every way the option parser can end a run before the search (usage, complaints, getopt corner cases such as
    abbreviated / ambiguous long options, missing values, `-?`) against the reference binary: same exit status, same
    stdout, same stderr bytes."""
    import subprocess
    if not os.access(oracle.REF_KWAGE, os.X_OK):
        pytest.skip("reference binary not built (oracle/_ref)")
    db = os.path.join(GOLDEN, "k32", "k32.db")
    cases = [[], ["-h"], ["-?"], ["--bogus"], ["-x"], ["--o"], ["--o.c", "-d", db], ["--o.j", "-d", db],
             ["-d", os.path.join(GOLDEN, "multi", "dbs", "not_a_db.txt"), "ACGT"], ["-d", db], ["-d", db, "-i", "reads.txt"],
             ["-d", db, "-i", "x.fa.fa"], ["-d", db, "-i", "x.fastq.gz.fa"], ["-d", db, "-i", "a.fna", "-i", "b.txt"],
             ["-d", db, "-t", "1.5", "ACGT"], ["-d", db, "-t", "0", "ACGT"], ["-d", db, "-t", "-1", "ACGT"], ["-d", db, "-t", "abc", "ACGT"],
             ["-d", db, "-t", "1.0000001", "ACGT"], ["ACGT", "-d", db, "-t"], ["-d"], ["-d", db, "-o"], ["-i", "q.fa"], ["-t", "0.5"],
             ["ACGT", "-h", "-d", db]]
    for argv in cases:
        ref = subprocess.run([oracle.REF_KWAGE] + argv, capture_output=True)
        own = subprocess.run([native.KWAGE_BIN] + argv, capture_output=True)
        assert (own.returncode, own.stdout, own.stderr) == (ref.returncode, ref.stdout, ref.stderr), argv


def test_header_is_plain_c99(tmp_path):
    """include/kwage_amd.h must be consumable from C (the FFI boundary): compile the C example strictly."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-c", os.path.join(ROOT, "examples", "search_example.c"), "-o", str(tmp_path / "ex.o")])
