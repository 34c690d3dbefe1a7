#!/usr/bin/env python3
"""PCIe-inclusive wall time of loading a database the size of C2 (105 GB) through the `kwage` CLI: N reference-format
files (2048 columns x 2^20 rows, 268 MB each) in a tmpfs / page-cache directory, one short query, KWAGE_VERBOSE=1.

    python tools/load_big_db.py [n_files=392] [dir=/dev/shm]"""
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

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 392
base = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
L, ncol, k, nh = 20, 2048, 31, 1
free = shutil.disk_usage(base).free
need = n_files * ((1 << L) * 256 + 200_000)
if free < need * 1.02:
    n_files = int(free / 1.02 // ((1 << L) * 256 + 200_000))
    print("only %.0f GB free under %s: %d files" % (free / 1e9, base, n_files))
# first-touch placement: write the files from CPUs of the NUMA node the GPU hangs off (KWAGE_BIG_NODE overrides; -1 = leave
# the writer where the scheduler puts it), so that the PCIe reads do not cross the socket interconnect
node = os.environ.get("KWAGE_BIG_NODE")
if node is None:
    try:
        cards = [d for d in sorted(os.listdir("/sys/class/drm")) if d.startswith("card") and d[4:].isdigit()
                 and os.path.exists("/sys/class/drm/%s/device/numa_node" % d)]
        nodes = [open("/sys/class/drm/%s/device/numa_node" % c).read().strip() for c in cards]
        print("GPU numa nodes:", dict(zip(cards, nodes)))
        node = next((x for x in nodes if x not in ("-1", "")), "-1")
    except Exception as exc:
        print("no numa information:", exc)
        node = "-1"
if node != "-1":
    try:
        cpus = set()
        for part in open("/sys/devices/system/node/node%s/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, cpus & os.sched_getaffinity(0) or os.sched_getaffinity(0))
        print("writer AND the kwage runs pinned to node %s (%d cpus)" % (node, len(os.sched_getaffinity(0))))
    except Exception as exc:
        print("could not pin the writer:", exc)
tmp = tempfile.mkdtemp(prefix="kwage_big_", dir=base)
try:
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, size=2000)].tobytes().decode()
    a = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
    b = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
    rows = (a & b).view(np.uint8).reshape(1 << L, ncol // 8).copy()
    for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1):
        rows[r, 5] |= np.uint8(1 << 3)                      # column 43 of every file holds the genome
    infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % j)) for j in range(ncol)]
    os.makedirs(os.path.join(tmp, "db"))
    first = os.path.join(tmp, "db", "part0000.db")
    t0 = time.perf_counter()
    oracle.write_db(first, k, nh, L, rows, ncol, infos)
    for f in range(1, n_files):
        shutil.copyfile(first, os.path.join(tmp, "db", "part%04d.db" % f))
    gb = n_files * (1 << L) * 256 / 1e9
    print("wrote %d files, %.1f GB, in %.1f s" % (n_files, gb, time.perf_counter() - t0), flush=True)
    # (a run that follows a whole-database run pays ~2.7 s of HIP initialisation for the release of the other's 105 GB:
    # the sparse runs are therefore done twice in a row)
    for env_extra in ({"KWAGE_SPARSE": "0"}, {"KWAGE_SPARSE": "0", "KWAGE_LOAD_SDMA": "0"}, {"KWAGE_SPARSE": "0"}, {"KWAGE_SPARSE": "0", "KWAGE_LOAD_SDMA": "0"}, {}, {}):
        t0 = time.perf_counter()
        r = subprocess.run([native.KWAGE_BIN, "-d", os.path.join(tmp, "db"), "--o.csv", genome[100:1100]], capture_output=True, text=True,
                           env=dict(os.environ, KWAGE_VERBOSE="1", **env_extra))
        wall = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        hits = len(r.stdout.strip().splitlines()) - 1
        print("%s: wall %.2f s for %.1f GB (%.1f GB/s end to end, PCIe + metadata + search + report), %d hits (expected %d)"
              % (("whole files, %s" % env_extra) if env_extra else "default (one 1 kb query: only the addressed slices are fetched)", wall, gb, gb / wall, hits, n_files))
        print("   " + "\n   ".join(l for l in r.stderr.strip().splitlines() if l.startswith("[kwage]")), flush=True)
        assert hits >= n_files
    if os.access(oracle.REF_KWAGE, os.X_OK):
        for threads in (16, 1):
            t0 = time.perf_counter()
            r = subprocess.run([oracle.REF_KWAGE, "-d", os.path.join(tmp, "db"), "--o.csv", genome[100:1100]], capture_output=True, text=True,
                               env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
            wall = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr
            print("reference kwage, %d OpenMP thread(s), same query, files in the page cache: wall %.2f s, %d hits" % (threads, wall, len(r.stdout.strip().splitlines()) - 1))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
