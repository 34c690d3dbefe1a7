#!/usr/bin/env python3
"""Loading a database whose files are NOT in the page cache: written, synced, evicted with
posix_fadvise(DONTNEED) before every run.  The disk's own sequential rate (dd-style read of the same evicted
files) beside the loader's default path (HSA-locked file mappings + copy engine) and the pread path.

    python tools/cold_load_probe.py [n_files=16]"""
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

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L, ncol, k, nh = 20, 2048, 31, 1
tmp = tempfile.mkdtemp(prefix="kwage_cold_", dir=os.environ.get("KWAGE_PROBE_DIR", "/tmp"))
try:
    rng = np.random.default_rng(3)
    os.makedirs(os.path.join(tmp, "db"))
    paths = []
    for f in range(n_files):
        rows = rng.integers(0, 256, size=(1 << L, ncol // 8), dtype=np.uint8)
        infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % (f * ncol + j))) for j in range(ncol)]
        p = os.path.join(tmp, "db", "part%03d.db" % f)
        oracle.write_db(p, k, nh, L, rows, ncol, infos)
        paths.append(p)
    gb = sum(os.path.getsize(p) for p in paths) / 1e9

    def evict():
        for p in paths:
            fd = os.open(p, os.O_RDONLY)
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            os.close(fd)

    def resident_fraction():
        # mincore through /proc is not available to python; time a read of the first MiB of each file instead
        t0 = time.perf_counter()
        for p in paths:
            with open(p, "rb") as fh:
                fh.read(1 << 20)
        return time.perf_counter() - t0

    evict()
    t0 = time.perf_counter()
    buf = bytearray(8 << 20)
    for p in paths:
        with open(p, "rb", buffering=0) as fh:
            while fh.readinto(buf):
                pass
    dt = time.perf_counter() - t0
    print("%d files, %.1f GB.  evicted, then read sequentially by one thread: %.2f s = %.2f GB/s" % (n_files, gb, dt, gb / dt))
    t0 = time.perf_counter()
    for p in paths:
        with open(p, "rb", buffering=0) as fh:
            while fh.readinto(buf):
                pass
    dt = time.perf_counter() - t0
    print("read again (page cache): %.2f s = %.2f GB/s" % (dt, gb / dt))
    for label, extra in (("default (reader process, 4 threads)", {}), ("no reader process (KWAGE_CACHE_READER=0)", {"KWAGE_CACHE_READER": "0"}),
                         ("reader process, 1 thread", {"KWAGE_CACHE_READER": "1"}), ("reader process, 8 threads", {"KWAGE_CACHE_READER": "8"}),
                         ("pread path, no reader (KWAGE_LOAD_MMAP=0)", {"KWAGE_LOAD_MMAP": "0", "KWAGE_CACHE_READER": "0"}),
                         ("pread path + reader process", {"KWAGE_LOAD_MMAP": "0"}),
                         ("hipHostRegister path + reader process (KWAGE_LOAD_SDMA=0)", {"KWAGE_LOAD_SDMA": "0"}),
                         ("default (reader process, 4 threads)", {}), ("no reader process (KWAGE_CACHE_READER=0)", {"KWAGE_CACHE_READER": "0"})):
        for state in ("cold", "warm"):
            if state == "cold":
                evict()
            r = subprocess.run([native.KWAGE_BIN, "-d", os.path.join(tmp, "db"), "--o.csv", "ACGTACGTACGTAAACCCGGGTTTACGTACGTACGT"], capture_output=True,
                               env=dict(os.environ, KWAGE_VERBOSE="1", KWAGE_SPARSE="0", **extra))
            assert r.returncode == 0, r.stderr.decode()
            line = [l for l in r.stderr.decode().splitlines() if ": init " in l][0]
            print("%-48s %s: %s" % (label, state, line[line.index("loaded"):line.index(", search")]))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
