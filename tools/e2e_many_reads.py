#!/usr/bin/env python3
"""End to end through the CLI with a hit-heavy read set: where does the wall time go once the search itself is
milliseconds?  n_files reference-format `.db` files x 2048 columns x 2^L rows, a FASTQ of n_reads x 100 bp of which
half are windows of planted genomes (each planted genome sits in one column of EVERY file, so a planted read has
n_files hits), CSV and JSON reports, KWAGE_VERBOSE's per-stage times; the reference `kwage` (16 OpenMP threads) on the
same input beside it when it is built.

    python tools/e2e_many_reads.py [n_files=8] [log2_len=20] [n_reads=500000]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import kwage_oracle as oracle
from kwage_amd import native

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 500_000
ncol, k, nh, read_len = 2048, 31, 1, 100
tmp = tempfile.mkdtemp(prefix="kwage_reads_", dir="/tmp")
try:
    rng = np.random.default_rng(23)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = [acgt[rng.integers(0, 4, size=200_000)].tobytes().decode() for _ in range(8)]
    grows = [np.unique(oracle.row_indices(oracle.unique_kmers(g, k), k, nh, L).reshape(-1)) for g in genomes]
    os.makedirs(os.path.join(tmp, "db"))
    for f in range(n_files):
        a = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
        b = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
        rows = (a & b).view(np.uint8).reshape(1 << L, ncol // 8).copy()
        for gi in range(8):
            col = 100 * gi + f
            rows[grows[gi], col // 8] |= np.uint8(1 << (col % 8))
        infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % (f * ncol + j))) for j in range(ncol)]
        oracle.write_db(os.path.join(tmp, "db", "part%03d.db" % f), k, nh, L, rows, ncol, infos)
    q = os.path.join(tmp, "reads.fastq")
    noise = acgt[rng.integers(0, 4, size=(n_reads // 2 + 1) * read_len)].tobytes().decode()
    qual = "I" * read_len
    with open(q, "w") as fh:
        chunk = []
        for i in range(n_reads):
            if i % 2 == 0:
                g = genomes[(i // 2) % 8]
                off = (i * 37) % (len(g) - read_len)
                s = g[off:off + read_len]
            else:
                j = (i // 2) * read_len
                s = noise[j:j + read_len]
            chunk.append("@read_%d\n%s\n+\n%s\n" % (i, s, qual))
            if len(chunk) == 50_000:
                fh.write("".join(chunk)); chunk = []
        fh.write("".join(chunk))
    print("%d files x %d columns x 2^%d rows (%.1f GB), %d reads x %d bp (%.0f MB FASTQ)" % (n_files, ncol, L, n_files * (1 << L) * 256 / 1e9, n_reads, read_len, os.path.getsize(q) / 1e6))

    outs = {}
    for fmt in ("--o.csv", "--o.json"):
        for thr in ("1.0", "0.8"):
            o = os.path.join(tmp, "ours%s_%s" % (fmt, thr))
            t0 = time.perf_counter()
            r = subprocess.run([native.KWAGE_BIN, "-d", os.path.join(tmp, "db"), "-i", q, fmt, "-t", thr, "-o", o], capture_output=True, env=dict(os.environ, KWAGE_VERBOSE="1"))
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr.decode()
            outs[(fmt, thr)] = o
            print("this repo's kwage %s -t %s: wall %.2f s, report %.0f MB" % (fmt, thr, dt, os.path.getsize(o) / 1e6))
            print("   " + "\n   ".join(l for l in r.stderr.decode().splitlines() if l.startswith("[kwage]") and ("init" in l or "report" in l or "command line" in l)))
    if os.access(oracle.REF_KWAGE, os.X_OK):
        cores = min(os.cpu_count() or 1, 16)
        for fmt, thr in (("--o.csv", "1.0"), ("--o.json", "0.8")):
            o = os.path.join(tmp, "ref")
            t0 = time.perf_counter()
            r = subprocess.run([oracle.REF_KWAGE, "-d", os.path.join(tmp, "db"), "-i", q, fmt, "-t", thr, "-o", o], capture_output=True, env=dict(os.environ, OMP_NUM_THREADS=str(cores)))
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr.decode()
            if fmt == "--o.csv":
                g, e = oracle.parse_csv(open(outs[(fmt, thr)]).read()), oracle.parse_csv(open(o).read())
                same = (list(g) == list(e)) and all(sorted(g[x]) == sorted(e[x]) for x in e)
                hits = sum(len(v) for v in e.values())
            else:
                # with several OpenMP threads the reference's order among equal scores depends on which thread finished first
                import json
                a, b = json.load(open(outs[(fmt, thr)])), json.load(open(o))
                canon = lambda doc: [(e["query"], e["threshold"], sorted(json.dumps(r, sort_keys=True) for r in e["results"])) for e in doc]
                same = canon(a) == canon(b)
                hits = sum(len(e["results"]) for e in b)
            print("reference kwage %s -t %s (%d OpenMP threads, page cache warm): wall %.2f s; reports identical: %s; hits %d" % (fmt, thr, cores, dt, same, hits))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
