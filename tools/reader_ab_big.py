#!/usr/bin/env python3
"""Does the CLI's page-cache reader cost anything on a large database that IS in the page cache?  n files of 268 MB
written once, then whole-file loads alternating KWAGE_CACHE_READER=4 / 0.   python tools/reader_ab_big.py [n_files=270] [dir=/tmp]"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import kwage_oracle as oracle
from kwage_amd import native

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 270
base = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
L, ncol, k, nh = 20, 2048, 31, 1
tmp = tempfile.mkdtemp(prefix="kwage_ab_", dir=base)
try:
    rng = np.random.default_rng(1)
    rows = rng.integers(0, 256, size=(1 << L, ncol // 8), dtype=np.uint8)
    infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % j)) for j in range(ncol)]
    first = os.path.join(tmp, "part0000.db")
    oracle.write_db(first, k, nh, L, rows, ncol, infos)
    for f in range(1, n_files):
        shutil.copyfile(first, os.path.join(tmp, "part%04d.db" % f))
    print("%d files, %.1f GB" % (n_files, n_files * os.path.getsize(first) / 1e9), flush=True)
    for rep in range(4):
        for threads in ("4", "0"):
            r = subprocess.run([native.KWAGE_BIN, "-d", tmp, "--o.csv", "ACGTACGTACGTAAACCCGGGTTTACGTACGTACGT"], capture_output=True,
                               env=dict(os.environ, KWAGE_VERBOSE="1", KWAGE_SPARSE="0", KWAGE_CACHE_READER=threads))
            assert r.returncode == 0, r.stderr.decode()
            line = [l for l in r.stderr.decode().splitlines() if ": init " in l][0]
            print("reader threads %s: %s" % (threads, line[line.index("init"):line.index(", search")]), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
