#!/usr/bin/env python3
"""What `kwage_node`'s software pipeline costs beside its gather kernels, on a database of several parameter groups
(the C5 shape at test scale: five filter sizes 2^16 .. 2^20, 24 2048-column files each = 6 KB rows, 12.5 GB, three hash
functions, threshold 0.9 -- wide enough that the device, not the FASTQ parser (2.6 M reads/s), sets the pace) and a FASTQ of reads streamed in many batches (KWAGE_BATCH_BASES):

  * `kwage` (one process)                                    -- the bytes every other run must reproduce
  * `kwage_node`, one rank over RCCL (KWAGE_NODE_RANKS=1)    -- communicator, all-gather of the counts, grouped send/recv
  * `kwage_node`, two and three rehearsed ranks on device 0  (KWAGE_NODE_REHEARSE=1)

each with KWAGE_NODE_STATS=1: rank 0 reports the wall of its search phase (first batch queued -> last batch filed) next
to the sum of its gather-kernel times; the goal is wall within a few per cent of the kernels (batch i's exchange and
filing overlap batch i+1's searches; the next batch is parsed on its own thread).

    python tools/node_pipeline_stats.py [n_reads=1000000] [batch_bases=4194304]
"""
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
batch_bases = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
ncol, k, nh, read_len, files_per_group = 2048, 31, 3, 100, int(os.environ.get("NODE_STATS_FILES_PER_GROUP", "24"))
groups = (16, 17, 18, 19, 20)
NODE_BIN = os.path.join(os.path.dirname(native.KWAGE_BIN), "kwage_node")
tmp = tempfile.mkdtemp(prefix="kwage_node_stats_", dir="/tmp")
try:
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, size=10_000)].tobytes().decode()      # (30 k planted rows: even the 2^16-row group stays far from all ones)
    km = oracle.unique_kmers(genome, k)
    os.makedirs(os.path.join(tmp, "db"))
    total = 0
    t0 = time.perf_counter()
    for L in groups:
        grows = np.unique(oracle.row_indices(km, k, nh, L).reshape(-1))
        a = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
        b = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
        base = (a & b).view(np.uint8).reshape(1 << L, ncol // 8)          # density 0.25: no chance matches at t = 0.9, the hits are the planted ones
        del a, b
        for f in range(files_per_group):
            rows = np.roll(base, 4099 * f, axis=0)
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
                fh.write("".join(chunk))
                chunk = []
        fh.write("".join(chunk))
    print("%d groups (2^%d .. 2^%d slices, %d hashes), %d files x %d columns each, %.1f GB; %d reads x %d bp in batches of %d bases (%d batches); built in %.0f s"
          % (len(groups), groups[0], groups[-1], nh, files_per_group, ncol, total / 1e9, n_reads, read_len, batch_bases,
             -(-n_reads * read_len // batch_bases), time.perf_counter() - t0), flush=True)
    base_env = dict(os.environ, KWAGE_BATCH_BASES=str(batch_bases), KWAGE_EARLY_EXIT="0", KWAGE_NODE_STATS="1", KWAGE_VERBOSE="1")
    argv = ["-d", os.path.join(tmp, "db"), "-i", q, "--o.csv", "-t", "0.9"]
    runs = [("kwage, one process", [native.KWAGE_BIN], {}),
            ("kwage_node, 1 rank over RCCL", [NODE_BIN], {"KWAGE_NODE_RANKS": "1"}),
            ("kwage_node, 2 rehearsed ranks on device 0", [NODE_BIN], {"KWAGE_NODE_RANKS": "2", "KWAGE_NODE_REHEARSE": "1"}),
            ("kwage_node, 3 rehearsed ranks on device 0", [NODE_BIN], {"KWAGE_NODE_RANKS": "3", "KWAGE_NODE_REHEARSE": "1"}),
            # a database twice what a rank may hold at once (KWAGE_MAX_GROUP_BYTES stands in for the free HBM): passes
            ("kwage_node, 1 rank over RCCL, budget = half the database", [NODE_BIN], {"KWAGE_NODE_RANKS": "1", "KWAGE_MAX_GROUP_BYTES": str(total // 2)}),
            ("kwage_node, 2 rehearsed ranks, budget = a quarter of the database each", [NODE_BIN], {"KWAGE_NODE_RANKS": "2", "KWAGE_NODE_REHEARSE": "1", "KWAGE_MAX_GROUP_BYTES": str(total // 4)})]
    outs = []
    for label, prog, extra in runs:
        for rep in range(2):          # (the second run has the files in the page cache and the pinned pools warm)
            o = os.path.join(tmp, "out.csv")
            t0 = time.perf_counter()
            r = subprocess.run(prog + argv + ["-o", o], capture_output=True, env=dict(base_env, **extra))
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr.decode()[-3000:]
        text = open(o).read()
        outs.append(text)
        err = r.stderr.decode()
        stat = [l for l in err.splitlines() if l.startswith("[kwage_node]") or ("init" in l and "search" in l)]
        print("%-44s wall %.2f s  report %d lines%s\n    %s" % (label, dt, text.count("\n"), "" if text == outs[0] else "  !! DIFFERS from kwage's", "\n    ".join(stat)), flush=True)
    assert all(sorted(t.splitlines()) == sorted(outs[0].splitlines()) for t in outs), "the reports differ"
    print("reports: identical lines in all %d runs (%d hits)" % (len(outs), outs[0].count("\n") - 1))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
