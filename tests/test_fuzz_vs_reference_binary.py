"""Randomised differential tests against the REFERENCE BINARY itself (oracle/_ref/kwage, built from
the reference's own sources by `make -C oracle ref`; it travels to the GPU box with the snapshot).
Skipped when the binary is absent.  CPU part: oracle == reference.  GPU part: this repo's `kwage`
CLI == reference, on directories of random databases with mixed parameters."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "kwage")
needs_ref = pytest.mark.skipif(not os.access(REF, os.X_OK), reason="reference binary not built (make -C oracle ref)")

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _seq(rng, n):
    return ACGT[rng.integers(0, 4, size=n)].tobytes().decode()


def _make_case(oracle, rng, root, n_files):
    """Random database directory + FASTA; returns (db_dir, fasta, cmdline_seqs)."""
    os.makedirs(os.path.join(root, "db", "sub"), exist_ok=True)
    genomes = [_seq(rng, int(rng.integers(60, 400))) for _ in range(3)]
    param_pool = [(int(rng.integers(1, 33)), int(rng.integers(1, 6)), int(rng.integers(8, 12))) for _ in range(2)]
    for f in range(n_files):
        k, nh, L = param_pool[f % len(param_pool)]
        n = int(rng.choice([1, 7, 8, 9, 31, 64, 65, 130]))
        dens = float(rng.choice([0.05, 0.3, 0.6]))
        bits = rng.random((1 << L, ((n + 7) // 8) * 8)) < dens
        bits[:, n:] = False
        rows = np.packbits(bits, axis=1, bitorder="little")
        for j in range(n):
            if rng.random() < 0.3:      # plant a genome (or a prefix of it) in this column
                g = genomes[int(rng.integers(len(genomes)))]
                g = g if rng.random() < 0.6 else g[: len(g) // 2]
                km = oracle.unique_kmers(g, k)
                if len(km):
                    r = oracle.row_indices(km, k, nh, L).reshape(-1)
                    rows[r, j // 8] |= np.uint8(1 << (j % 8))
        infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%d" % (1000 * f + j + 1))) for j in range(n)]
        sub = "sub" if f % 3 == 2 else ""
        oracle.write_db(os.path.join(root, "db", sub, "f%02d.db" % f), k, nh, L, rows, n, infos)
    # query file: FASTA (sometimes gzip'd, CRLF, wrapped lines, blank lines, over-long deflines, '>' inside a
    # line, odd leading characters) or FASTQ (sometimes with >2047-character sequence lines)
    recs = []
    for i in range(int(rng.integers(3, 9))):
        kind = rng.integers(4)
        g = genomes[int(rng.integers(len(genomes)))]
        if kind == 0:
            s = g
        elif kind == 1:
            a = int(rng.integers(0, len(g) // 2)); s = g[a:a + int(rng.integers(20, len(g)))]
        elif kind == 2:
            s = g[: len(g) // 2] + "N" + _seq(rng, 40)
        else:
            s = _seq(rng, int(rng.integers(5, 200)))
        if rng.random() < 0.3:
            s = s.lower()
        recs.append(("q%d some text" % i, s))
    style = int(rng.integers(5))
    eol = "\r\n" if rng.random() < 0.3 else "\n"
    if style == 4:                                           # FASTQ
        fasta = os.path.join(root, "q.fastq")
        with open(fasta, "w", newline="") as fh:
            for i, (d, s) in enumerate(recs):
                if i == 1:
                    s = s * (2500 // max(len(s), 1) + 1)     # one gzgets chunk is not enough for this line
                fh.write("@%s%s%s%s+%s%s%s" % (d, eol, s, eol, eol, "I" * len(s), eol))
    else:
        fasta = os.path.join(root, "q.fa.gz" if style == 3 else "q.fasta")
        out = []
        for i, (d, s) in enumerate(recs):
            if i == 0 and style == 1:
                d = "  > " + d + " " + "x" * 2600           # longer than the 2048-byte gzgets buffer
            if i == 2 and style == 2:
                d = d + " a>b, \"quoted\""
            out.append(">" + d + eol)
            w = int(rng.choice([60, 70, 1000000]))
            for o in range(0, len(s), w):
                out.append(s[o:o + w] + (" " if rng.random() < 0.1 else "") + eol)
            if rng.random() < 0.2:
                out.append(eol)
        text = "".join(out)
        if style == 3:
            import gzip
            with gzip.open(fasta, "wt", newline="") as fh:
                fh.write(text)
        else:
            with open(fasta, "w", newline="") as fh:
                fh.write(text)
    cmd = [genomes[0][:50]] if rng.random() < 0.5 else []
    return os.path.join(root, "db"), fasta, cmd


def _run(exe, db, fasta, cmd, thr, fmt="csv", env=None):
    r = subprocess.run([exe, "-d", db, "-i", fasta, "-t", thr, "--o." + fmt] + cmd, capture_output=True,
                       env=dict(os.environ, OMP_NUM_THREADS="1", **(env or {})))
    assert r.returncode == 0, r.stderr.decode()
    return r.stdout.decode("latin-1")


@needs_ref
@pytest.mark.parametrize("seed", range(int(os.environ.get("KWAGE_FUZZ_SEEDS_CPU", "30"))))
def test_oracle_equals_reference_binary(oracle, tmp_path, seed):
    rng = np.random.default_rng(4242 + seed)
    db, fasta, cmd = _make_case(oracle, rng, str(tmp_path), int(rng.integers(1, 5)))
    for thr in ("1.0", "%.3f" % rng.uniform(0.05, 0.99), str(rng.choice(["0.0001", "0.999999", "1e-9", "0.5", "0.33333334", "0.99999999", ".75", "1", "1.0e0", "9.9e-1"]))):
        # ("0.99999999" rounds to 1.0f in the reference's `float threshold`: the AND path)
        exp = oracle.parse_csv(_run(REF, db, fasta, cmd, thr))
        got = oracle.run_search([db], [fasta], cmd, float(thr))
        by_name = {}
        for key, hits in got.items():
            by_name.setdefault(key.split("\t", 1)[1] if "\t" in key else key, []).extend(hits)
        assert set(by_name) == set(exp), (seed, thr)
        for qn in exp:
            assert sorted(by_name[qn]) == sorted((a, nk, nf) for a, nk, nf, _ in exp[qn]), (seed, thr, qn)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("KWAGE_FUZZ_SEEDS", "20"))))      # (6-9 s each: CSV + JSON at three thresholds through two binaries; raise for a soak)
def test_cli_equals_reference_binary(oracle, tmp_path, seed):
    from kwage_amd import native
    rng = np.random.default_rng(777 + seed)
    db, fasta, cmd = _make_case(oracle, rng, str(tmp_path), int(rng.integers(1, 7)))
    for thr in ("1.0", "%.3f" % rng.uniform(0.05, 0.99), str(rng.choice(["0.0001", "0.999999", "1e-9", "0.5", "0.33333334", "0.99999999", ".75", "1", "1.0e0", "9.9e-1"]))):
        # ("0.99999999" rounds to 1.0f in the reference's `float threshold`: the AND path)
        for fmt in ("csv", "json"):
            exp = _run(REF, db, fasta, cmd, thr, fmt)
            got = _run(native.KWAGE_BIN, db, fasta, cmd, thr, fmt)
            # tie order is unspecified in the reference (readdir order x unstable sort): compare the
            # multiset of lines and the exact sequence of scores / query names
            assert sorted(got.splitlines()) == sorted(exp.splitlines()), (seed, thr, fmt)
            if fmt == "csv":
                score = lambda t: [ln.rsplit(",", 3)[0:3] for ln in t.splitlines()[1:]]
                assert [s[0] + s[1] + s[2] for s in score(got)] == [s[0] + s[1] + s[2] for s in score(exp)]
                continue
            # the three ways the CLI may fetch the database give the same bytes: only the addressed slices (sparse),
            # whole files with the query set preloaded, whole files with the queries streamed batch by batch
            assert _run(native.KWAGE_BIN, db, fasta, cmd, thr, fmt, {"KWAGE_SPARSE": "1"}) == got, (seed, thr, fmt)
            assert _run(native.KWAGE_BIN, db, fasta, cmd, thr, fmt, {"KWAGE_SPARSE": "0"}) == got, (seed, thr, fmt)
            if float(np.float32(float(thr))) == 1.0:
                # ... and, at t = 1, the sparse fetch screened on every query's first k-mers (2: even these short queries qualify; 0: never)
                for head in ("2", "0"):
                    assert _run(native.KWAGE_BIN, db, fasta, cmd, thr, fmt, {"KWAGE_SPARSE": "1", "KWAGE_SPARSE_SCREEN": head}) == got, (seed, thr, fmt, head)
            if seed % 4 == 0:
                assert _run(native.KWAGE_BIN, db, fasta, cmd, thr, fmt, {"KWAGE_SPARSE_BASES": "1"}) == got, (seed, thr, fmt)
            if seed % 4 == 2:
                # without the early exit the CLI's searches go through the persistent kernels where the knobs (read once,
                # from the environment, when the context is created) send even these tiny batches: and_walk_kernel /
                # count_walk_kernel with a handful of waves, pairs cut by share boundaries added up through memory
                walk = {"KWAGE_EARLY_EXIT": "0", "KWAGE_SPARSE": "0", "KWAGE_NARROW": "0", "KWAGE_WALK_MIN_ROWS": "1", "KWAGE_COUNT_WALK_MIN_ROWS": "1",
                        "KWAGE_COUNT_WALK_WAVES": "37", "KWAGE_WALK_WAVES": "37"}
                assert _run(native.KWAGE_BIN, db, fasta, cmd, thr, fmt, walk) == got, (seed, thr, fmt)
            if seed % 4 == 1:
                # streamed in small batches against several resident units per pass, and against passes so small that
                # files with different parameters (and k-mer lengths) are split over several of them
                small = {"KWAGE_SPARSE_BASES": "1", "KWAGE_BATCH_BASES": "300"}
                assert _run(native.KWAGE_BIN, db, fasta, cmd, thr, fmt, small) == got, (seed, thr, fmt)
                for budget in ("1", "40000"):
                    assert _run(native.KWAGE_BIN, db, fasta, cmd, thr, fmt, dict(small, KWAGE_MAX_GROUP_BYTES=budget)) == got, (seed, thr, fmt, budget)
