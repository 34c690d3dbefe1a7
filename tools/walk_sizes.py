#!/usr/bin/env python3
"""and_walk_kernel vs and_kernel over batch sizes around the chip's resident capacity (1024 workgroups at 4 per
CU): is there a slow second round?   python tools/walk_sizes.py"""
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
for nq in (900, 1000, 1024, 1030, 1100, 1300, 1500, 2048, 2100, 3000, 5000):
    qs = (s.queries + extra)[:nq]
    b = ka.Batch(ctx, qs)
    out = []
    for v in ("0", "4"):
        os.environ["KWAGE_WALK"] = v
        ms = [s.group.search(b, 1.0, ka.SEARCH_TIMING).search_kernel_ms for _ in range(6)]
        r = s.group.search(b, 1.0, ka.SEARCH_TIMING)
        out.append((r.search_kernel, float(np.median(ms[1:])), r.algorithmic_bytes))
    b.close()
    print("%5d queries: %s %.3f ms (%.0f GB/s) | %s %.3f ms (%.0f GB/s)" % (nq, out[0][0], out[0][1], out[0][2] / out[0][1] / 1e6, out[1][0], out[1][1], out[1][2] / out[1][1] / 1e6), flush=True)
