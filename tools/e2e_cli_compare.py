#!/usr/bin/env python3
"""End-to-end drop-in check at scale: the same directory of reference-format `.db` files and the same
FASTA through (a) the reference `kwage` (oracle/_ref/kwage, OpenMP over files) and (b) this repo's
`kwage` CLI on one MI355X.  Compares the reports and the wall times (CLI time includes reading the
files, PCIe H2D, search, metadata, printing: the PCIe-inclusive figure DESIGN.md quotes).

    python tools/e2e_cli_compare.py [n_files] [log2_len] [n_queries]"""
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

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n_queries = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
ncol, k, nh = 2048, 31, 1
tmp = tempfile.mkdtemp(prefix="kwage_e2e_", dir="/tmp")
try:
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = [acgt[rng.integers(0, 4, size=50_000)].tobytes().decode() for _ in range(8)]
    queries = []
    for i in range(n_queries):
        if i % 2 == 0:
            g = genomes[(i // 2) % 8]
            off = ((i // 16) * 1000) % 49_000
            queries.append(g[off:off + 1000])
        else:
            queries.append(acgt[rng.integers(0, 4, size=1000)].tobytes().decode())
    grows = [np.unique(oracle.row_indices(oracle.unique_kmers(g, k), k, nh, L).reshape(-1)) for g in genomes]
    os.makedirs(os.path.join(tmp, "db"))
    t0 = time.perf_counter()
    for f in range(n_files):
        a = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
        b = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
        rows = (a & b).view(np.uint8).reshape(1 << L, ncol // 8).copy()
        for gi in range(8):                          # genome gi lives in column 100*gi + f of every file
            col = 100 * gi + f
            rows[grows[gi], col // 8] |= np.uint8(1 << (col % 8))
        infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % (f * ncol + j))) for j in range(ncol)]
        oracle.write_db(os.path.join(tmp, "db", "part%03d.db" % f), k, nh, L, rows, ncol, infos)
    print("wrote %d files x %d columns x 2^%d rows (%.1f GB) in %.1f s" % (n_files, ncol, L, n_files * (1 << L) * 256 / 1e9, time.perf_counter() - t0))
    q = os.path.join(tmp, "q.fa")
    with open(q, "w") as fh:
        for i, s in enumerate(queries):
            fh.write(">query_%d\n%s\n" % (i, s))

    def run(exe, env=None, reps=2):
        best, out = None, None
        for _ in range(reps):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-d", os.path.join(tmp, "db"), "-i", q, "--o.csv"], capture_output=True, env=env)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr.decode()
            best = dt if best is None else min(best, dt)
            out = r.stdout.decode()
            inside = [l.split()[1] for l in r.stderr.decode().splitlines() if "from the start of main" in l]
            if inside:          # KWAGE_VERBOSE: how much of the wall time lies outside main (program start, exit)
                init = [l for l in r.stderr.decode().splitlines() if ": init " in l][0]
                print("   wall %.3f s, inside main %s s (%s)" % (dt, inside[0], init[init.index("init"):init.index(", search")]))
        return best, out

    cores = min(os.cpu_count() or 1, 16)
    t_gpu, out_gpu = run(native.KWAGE_BIN, dict(os.environ, KWAGE_VERBOSE="1"), reps=5)
    r = subprocess.run([native.KWAGE_BIN, "-d", os.path.join(tmp, "db"), "-i", q, "--o.csv", "-o", os.path.join(tmp, "o.csv")], capture_output=True, env=dict(os.environ, KWAGE_VERBOSE="1"))
    print(r.stderr.decode().strip())
    r = subprocess.run([native.KWAGE_BIN, "-d", os.path.join(tmp, "db"), "-i", q, "--o.csv", "-o", os.path.join(tmp, "o2.csv")], capture_output=True,
                       env=dict(os.environ, KWAGE_VERBOSE="1", KWAGE_LOAD_DIRECT="1", KWAGE_LOAD_GANG="1"))
    print("with KWAGE_LOAD_DIRECT=1 KWAGE_LOAD_GANG=1 (copy kernel reading HSA-locked file windows, one file per launch):\n" + r.stderr.decode().strip())
    assert open(os.path.join(tmp, "o.csv")).read() == open(os.path.join(tmp, "o2.csv")).read()
    for rep in range(3):
        for extra in ({}, {"KWAGE_LOAD_NUMA": "0"}, {"KWAGE_LOAD_SDMA": "0"}, {"KWAGE_LOAD_DIRECT": "1", "KWAGE_LOAD_GANG": "1"}, {"KWAGE_LOAD_MMAP": "0"}):
            r = subprocess.run([native.KWAGE_BIN, "-d", os.path.join(tmp, "db"), "-i", q, "--o.csv", "-o", os.path.join(tmp, "o3.csv")], capture_output=True,
                               env=dict(os.environ, KWAGE_VERBOSE="1", **extra))
            print("%s: %s" % (extra or "default (copy-engine pipeline, loading thread on the GPU's NUMA node)", [l for l in r.stderr.decode().splitlines() if "loaded" in l and "GB/s" in l][0]))
    bit_tests = sum(len(oracle.unique_kmers(s, k)) for s in queries) * nh * ncol * n_files
    print("this repo's kwage (1 GPU): wall %.2f s  -> %.1f G bit-tests/s end to end (file read + H2D + search + report)" % (t_gpu, bit_tests / t_gpu / 1e9))
    if os.access(oracle.REF_KWAGE, os.X_OK):
        t_ref, out_ref = run(oracle.REF_KWAGE, dict(os.environ, OMP_NUM_THREADS=str(cores)))
        g, e = oracle.parse_csv(out_gpu), oracle.parse_csv(out_ref)
        same = (list(g) == list(e)) and all(sorted(g[x]) == sorted(e[x]) for x in e)
        print("reference kwage (%d OpenMP threads, page cache warm): wall %.2f s -> %.1f G bit-tests/s; reports identical (as sets per query): %s; "
              "hits %d; speed-up %.1fx" % (cores, t_ref, bit_tests / t_ref / 1e9, same, sum(len(v) for v in e.values()), t_ref / t_gpu))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
