#!/usr/bin/env python3
"""How much of a short CLI run lies outside main(): program start (dynamic linking of the HIP runtime) and exit
(runtime teardown)?   python tools/exit_probe.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kwage_amd import native

g = os.path.join(ROOT, "tests", "golden", "basic")
argv = [native.KWAGE_BIN, "-d", os.path.join(g, "db"), "-i", os.path.join(g, "q.fa"), "-t", "0.5", "--o.csv"]
for rep in range(4):
    t0 = time.perf_counter()
    r = subprocess.run(argv, capture_output=True, env=dict(os.environ, KWAGE_VERBOSE="1"))
    wall = time.perf_counter() - t0
    inside = [l for l in r.stderr.decode().splitlines() if "from the start of main" in l][0].split()[1]
    init = [l for l in r.stderr.decode().splitlines() if ": init " in l][0]
    print("wall %.3f s, inside main %s s  (%s)" % (wall, inside, init[init.index("init"):init.index(", loaded")]))
r = subprocess.run(argv, capture_output=True, env=dict(os.environ, LD_DEBUG="statistics"))
print("\n".join(l for l in r.stderr.decode().splitlines() if "total startup time" in l or "time needed for relocation" in l or "load objects" in l))
t0 = time.perf_counter(); subprocess.run(["/bin/true"]); print("fork+exec of /bin/true: %.3f s" % (time.perf_counter() - t0))
