"""The `kwage` CLI streams its query files (kwage.cpp:129-148 holds one record at a time): host memory must be
O(batch + hits), never O(query set) -- a 100 M-read FASTQ would otherwise sit in RAM before the GPU sees a byte."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_with_peak_rss(argv, env):
    """-> (exit status, stdout, stderr, peak resident set size of the child in bytes)."""
    import tempfile
    with tempfile.TemporaryFile() as so, tempfile.TemporaryFile() as se:
        p = subprocess.Popen(argv, stdout=so, stderr=se, env=env)
        _, status, ru = os.wait4(p.pid, 0)
        p.returncode = os.waitstatus_to_exitcode(status)
        so.seek(0), se.seek(0)
        return p.returncode, so.read().decode(), se.read().decode(), ru.ru_maxrss * 1024


def test_query_file_larger_than_the_memory_the_cli_may_use(oracle, tmp_path):
    from kwage_amd import native
    rng = np.random.default_rng(77)
    k, nh, L, ncol = 31, 2, 14, 40
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, size=3000)].tobytes().decode()
    rows = np.zeros((1 << L, (ncol + 7) // 8), dtype=np.uint8)
    planted = (3, 17, 39)
    for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1):
        for c in planted:
            rows[r, c // 8] |= np.uint8(1 << (c % 8))
    infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % j)) for j in range(ncol)]
    db = str(tmp_path / "one.db")
    oracle.write_db(db, k, nh, L, rows, ncol, infos)

    # 1.2 M reads x 150 bp = 180 MB of bases, 370 MB of FASTQ; a handful of them come from the planted genome
    n_reads, read_len = 1_200_000, 150
    special = {5: 10, 400_000: 500, 799_999: 1500, n_reads - 1: 2850}        # read index -> offset in the genome
    big = str(tmp_path / "big.fastq")
    small = str(tmp_path / "small.fastq")
    batch_bases = 4 << 20
    n_small = 3 * batch_bases // read_len
    qual = b"I" * read_len
    with open(big, "wb") as fh, open(small, "wb") as fs:
        chunk = 100_000
        for c0 in range(0, n_reads, chunk):
            bases = acgt[rng.integers(0, 4, size=(chunk, read_len))]
            parts = []
            for i in range(chunk):
                idx = c0 + i
                seq = genome[special[idx]:special[idx] + read_len].encode() if idx in special else bases[i].tobytes()
                parts.append(b"@r%d\n%s\n+\n%s\n" % (idx, seq, qual))
            fh.write(b"".join(parts))
            if c0 < n_small:
                # the yardstick: the first three batches' worth of the same file (both search slots and every
                # per-batch buffer of the engine, the HIP runtime's staging pools ... reach their size), 1/14 of the reads
                fs.write(b"".join(parts[:n_small - c0]))
    file_bytes = os.path.getsize(big)
    assert file_bytes > 350 << 20 and os.path.getsize(small) < file_bytes // 12

    env = dict(os.environ, KWAGE_BATCH_BASES=str(batch_bases), KWAGE_VERBOSE="1")
    rc0, out0, err0, rss_small = _run_with_peak_rss([native.KWAGE_BIN, "-d", db, "-i", small, "--o.csv"], env)
    assert rc0 == 0, err0
    rc, out, err, rss_big = _run_with_peak_rss([native.KWAGE_BIN, "-d", db, "-i", big, "--o.csv"], env)
    assert rc == 0, err
    print(err0, err)
    # 14 times the reads must not cost memory: a query set held in RAM would add 180 MB of bases (plus a
    # std::string per read, > 250 MB in all); streaming adds nothing but allocator slack.  What the HIP runtime maps at
    # start-up differs by tens of MB from process to process (the "device 0 ready" line), so the growth of each run over
    # its own start-up is compared, not the absolute peaks.
    def at_start(stderr_text):
        line = [l for l in stderr_text.splitlines() if "ready:" in l][0]
        return int(line.split("ready:")[1].split()[0]) << 20
    grow_small, grow_big = rss_small - at_start(err0), rss_big - at_start(err)
    assert grow_big - grow_small < 64 << 20, (grow_small >> 20, grow_big >> 20, err0, err)

    # and the report is right: exactly the planted reads, each found in exactly the planted samples
    rep = oracle.parse_csv(out)
    assert sorted(rep) == sorted("r%d" % i for i in special)
    assert sorted(oracle.parse_csv(out0)) == sorted("r%d" % i for i in special if i < n_small)
    nk = read_len - k + 1
    want = sorted(("SRR%07d" % c, nk, nk) for c in planted)
    for name in rep:
        assert sorted((acc, n, f) for acc, n, f, _ in rep[name]) == want


def test_batches_of_every_size_give_the_same_report(oracle, tmp_path, golden_dir):
    """KWAGE_BATCH_BASES from 'one query per batch' to 'everything in one' on a golden multi-file case with two -i
    files and command-line sequences: identical bytes (ids run on across files and batches)."""
    from kwage_amd import native
    base = os.path.join(golden_dir, "multi")
    qs = sorted(os.path.join(base, f) for f in os.listdir(base) if f.endswith((".fa", ".fasta", ".fna", ".fastq", ".fa.gz", ".fastq.gz")))
    if not qs:
        pytest.skip("no query files in the golden multi case")
    argv = [native.KWAGE_BIN, "-d", os.path.join(base, "dbs"), "-t", "0.5"]
    for q in qs:
        argv += ["-i", q]
    argv += ["ACGTACGTACGTACGTACGTACGTACGTACGTACGT"]
    outs = {}
    for fmt in ("--o.csv", "--o.json"):
        for bb in ("1", "300", "5000", str(64 << 20)):
            r = subprocess.run(argv + [fmt], capture_output=True, env=dict(os.environ, KWAGE_BATCH_BASES=bb))
            assert r.returncode == 0, r.stderr.decode()
            outs[(fmt, bb)] = r.stdout
        assert len({outs[(fmt, bb)] for bb in ("1", "300", "5000", str(64 << 20))}) == 1, fmt
        assert outs[(fmt, "1")]


def test_query_files_are_read_once_in_a_single_pass(oracle, tmp_path, golden_dir):
    """A run that needs one pass over the database reads every query file exactly once, like the reference
    (kwage.cpp:127-147) -- so a named pipe works as a query file.  Both ways a query set is classified: small (read whole
    by the preview) and not small (KWAGE_SPARSE_BASES=1: the preview stops after two batches and the first pass goes on
    from there)."""
    import threading
    from kwage_amd import native
    base = os.path.join(golden_dir, "basic")
    data = open(os.path.join(base, "q.fa"), "rb").read()
    argv = [native.KWAGE_BIN, "-d", os.path.join(base, "db"), "-t", "0.5", "--o.csv"]
    want = subprocess.run(argv + ["-i", os.path.join(base, "q.fa")], capture_output=True)
    assert want.returncode == 0 and want.stdout.count(b"\n") > 1
    for env_extra in ({}, {"KWAGE_SPARSE_BASES": "1"}, {"KWAGE_SPARSE_BASES": "1", "KWAGE_BATCH_BASES": "100"}):
        fifo = str(tmp_path / ("pipe%d.fa" % len(env_extra)))
        os.mkfifo(fifo)

        def feed():
            with open(fifo, "wb") as fh:        # a second open() by the reader would block for ever: the test would time out
                fh.write(data)
        t = threading.Thread(target=feed, daemon=True)
        t.start()
        got = subprocess.run(argv + ["-i", fifo], capture_output=True, env=dict(os.environ, **env_extra), timeout=120)
        t.join(timeout=10)
        assert got.returncode == 0, got.stderr.decode()
        assert got.stdout == want.stdout, env_extra


def test_page_cache_reader_process_changes_nothing_but_the_loading(tmp_path, golden_dir):
    """The CLI's page-cache reader (a child forked before the GPU is touched) with 0, 1 and 4 threads, on database files
    evicted from the page cache, whole-file loading forced: identical reports, exit status 0, and the child is gone
    when the program ends (its stdout pipe closes: subprocess.run returns)."""
    import shutil
    from kwage_amd import native
    base = os.path.join(golden_dir, "multi")
    dbs = str(tmp_path / "dbs")
    shutil.copytree(os.path.join(base, "dbs"), dbs)
    argv = [native.KWAGE_BIN, "-d", dbs, "-i", os.path.join(base, "reads.fastq"), "-t", "0.7", "--o.json"]
    outs = []
    for threads, devices in (("0", "0"), ("1", "0"), ("4", "0"), ("2", "0,0")):       # the last: one reader per device context
        for root, _, names in os.walk(dbs):
            for n in names:
                fd = os.open(os.path.join(root, n), os.O_RDONLY)
                os.fsync(fd)
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                os.close(fd)
        r = subprocess.run(argv, capture_output=True, env=dict(os.environ, KWAGE_CACHE_READER=threads, KWAGE_DEVICES=devices, KWAGE_SPARSE="0", KWAGE_CACHE_READER_AHEAD_MB="1"), timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] == outs[3] and len(outs[0]) > 100
