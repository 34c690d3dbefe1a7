#!/usr/bin/env python3
"""A database with several filter sizes (parameter groups), end to end through the CLI: all groups that fit the device
are resident together and the query file is read once (default), against one group per pass with the query file read
again for every group (KWAGE_ONE_UNIT_PER_PASS=1, the earlier schedule).  Plain and gzip'd FASTQ.

    python tools/e2e_multi_group.py [n_reads=1000000]"""
import gzip
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

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ncol, k, nh, read_len = 2048, 31, 3, 100
groups = (16, 17, 18, 19, 20, 21)          # log2 filter lengths, two files each
tmp = tempfile.mkdtemp(prefix="kwage_groups_", dir="/tmp")
try:
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, size=100_000)].tobytes().decode()
    km = oracle.unique_kmers(genome, k)
    os.makedirs(os.path.join(tmp, "db"))
    total = 0
    for L in groups:
        grows = np.unique(oracle.row_indices(km, k, nh, L).reshape(-1))
        for f in range(2):
            a = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
            b = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
            rows = (a & b).view(np.uint8).reshape(1 << L, ncol // 8).copy()
            col = 7 * L + f
            rows[grows, col // 8] |= np.uint8(1 << (col % 8))
            infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%02d%05d" % (L, f * ncol + j))) for j in range(ncol)]
            oracle.write_db(os.path.join(tmp, "db", "L%d_%d.db" % (L, f)), k, nh, L, rows, ncol, infos)
            total += (1 << L) * ncol // 8
    q = os.path.join(tmp, "reads.fastq")
    noise = acgt[rng.integers(0, 4, size=n_reads * read_len)].tobytes().decode()
    qual = "I" * read_len
    with open(q, "w") as fh:
        chunk = []
        for i in range(n_reads):
            if i % 50 == 0:
                off = (i * 37) % (len(genome) - read_len)
                s = genome[off:off + read_len]
            else:
                s = noise[i * read_len:(i + 1) * read_len]
            chunk.append("@read_%d\n%s\n+\n%s\n" % (i, s, qual))
            if len(chunk) == 50_000:
                fh.write("".join(chunk)); chunk = []
        fh.write("".join(chunk))
    qz = q + ".gz"
    with open(q, "rb") as src, gzip.open(qz, "wb", compresslevel=4) as dst:
        shutil.copyfileobj(src, dst, 1 << 20)
    print("%d groups (2^%d .. 2^%d slices, %d hashes), 2 files x %d columns each, %.1f GB; %d reads x %d bp (%.0f MB FASTQ, %.0f MB gzip'd)"
          % (len(groups), groups[0], groups[-1], nh, ncol, total / 1e9, n_reads, read_len, os.path.getsize(q) / 1e6, os.path.getsize(qz) / 1e6))
    outs = {}
    for qfile in (q, qz):
        for label, extra in (("all groups resident, queries read once", {}), ("one group per pass (earlier schedule)", {"KWAGE_ONE_UNIT_PER_PASS": "1"})):
            best = None
            for rep in range(2):
                o = os.path.join(tmp, "out.csv")
                t0 = time.perf_counter()
                r = subprocess.run([native.KWAGE_BIN, "-d", os.path.join(tmp, "db"), "-i", qfile, "--o.csv", "-t", "0.9", "-o", o], capture_output=True,
                                   env=dict(os.environ, KWAGE_VERBOSE="1", **extra))
                dt = time.perf_counter() - t0
                assert r.returncode == 0, r.stderr.decode()
                best = dt if best is None else min(best, dt)
            outs[(qfile, label)] = open(o).read()
            line = [l for l in r.stderr.decode().splitlines() if "init" in l and "search" in l][0]
            print("%-14s %-42s wall %.2f s   %s" % (os.path.basename(qfile), label, best, line[line.index("init"):line.index("; ")]))
    texts = list(outs.values())
    print("reports identical: %s; hits %d" % (all(t == texts[0] for t in texts), texts[0].count("\n") - 1))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
