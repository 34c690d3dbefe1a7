#!/usr/bin/env python3
"""What the loader's choice between two candidate placements costs at load time: group creation timed for matrices of
105, 105, 26 and 131 GB in one process; with KWAGE_VERBOSE=1 the loader prints allocation / probe / release times and
the probe's rates.   KWAGE_VERBOSE=1 python tools/placement_timing.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import kwage_amd as ka
ctx = ka.Context(0)
for L, cols in ((23, 100_000), (23, 100_000), (21, 100_000), (20, 1_000_000)):
    t = time.time()
    g = ka.Group(ctx, 31, 1, L, cols)
    print("group create %.2f s" % (time.time() - t), g.placement, flush=True)
    t = time.time(); g.close(); print("close %.2f s" % (time.time() - t), flush=True)
