#!/usr/bin/env python3
"""and_walk_kernel (persistent grid, equal shares of the batch's row list per wave) vs the tiled and_kernel over
batch sizes: small batches (where does the walk form start to pay?) and sizes around what used to be a partly
filled last round of one-workgroup-per-query (1030, 2100 queries).   python tools/walk_sizes.py [waves ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kwage_amd as ka
from kwage_amd import synth

ctx = ka.Context(0)
w = synth.WORKLOADS["c2"]
s = synth.build(ctx, w)
rng = np.random.default_rng(5)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
extra = [acgt[rng.integers(0, 4, size=1000)].tobytes().decode() for _ in range(4000)]
variants = [("tiled", {"walk": 0}), ("walk", {"walk": 4, "walk_min_rows": 1})]
for wv in sys.argv[1:]:
    if True:
        variants.append(("walk/%s waves" % wv, {"walk": 4, "walk_min_rows": 1, "walk_waves": int(wv)}))
for nq in [int(x) for x in os.environ.get("WALK_SIZES", "50,100,200,300,500,700,900,1000,1024,1030,1100,1300,1500,2048,2100,3000,5000").split(",")]:
    qs = (s.queries + extra)[:nq]
    b = ka.Batch(ctx, qs)
    out = []
    for name, knobs in variants:
        with ctx.tuning(**dict({"walk": 4, "walk_min_rows": -1, "walk_waves": 0}, **knobs)):
            ms = [s.group.search(b, 1.0, ka.SEARCH_TIMING).search_kernel_ms for _ in range(6)]
            r = s.group.search(b, 1.0, ka.SEARCH_TIMING)
        out.append("%s %s %.3f ms (%.0f GB/s)" % (name, r.search_kernel, float(np.median(ms[1:])), r.algorithmic_bytes / float(np.median(ms[1:])) / 1e6))
    b.close()
    print("%5d queries: %s" % (nq, " | ".join(out)), flush=True)
