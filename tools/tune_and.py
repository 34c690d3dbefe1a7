#!/usr/bin/env python3
"""Sweep and_kernel shapes (the and_vec / and_unroll / and_nt / and_lds_kb / and_block_waves knobs) on one resident workload, interleaved
rounds in ONE process (cdna_hip_programming.md section 5.4 rule 24).  Prints median/min kernel ms
and algorithmic GB/s per variant.   python tools/tune_and.py [workload] [rounds] ["vec,unroll,nt,ldsKB,blockwaves;..."]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
variants = [(2, 8, 1, 0, 4), (2, 8, 1, 0, 2), (2, 8, 1, 0, 1), (1, 8, 1, 0, 4), (1, 8, 1, 0, 1), (1, 16, 1, 0, 1),
            (2, 16, 1, 0, 4), (2, 16, 1, 0, 1), (4, 8, 1, 0, 4), (4, 8, 1, 0, 1), (2, 8, 0, 0, 4)]
if len(sys.argv) > 3:
    variants = [tuple(int(x) for x in v.split(",")) for v in sys.argv[3].split(";") if v]
ctx = ka.Context(0)
ctx.set_tuning("walk", 0)      # the tiled kernel is what is swept
s = synth.build(ctx, synth.WORKLOADS[wl])
ms = {v: [] for v in variants}
ref = None
for r in range(rounds):
    for v in variants:
        for name, x in zip(("and_vec", "and_unroll", "and_nt", "and_lds_kb", "and_block_waves"), v):
            ctx.set_tuning(name, x)
        res = s.group.search(s.batch, s.workload.threshold, ka.SEARCH_TIMING)
        key = (len(res.hits), int(res.hits["column"].astype(np.uint64).sum()), int(res.hits["query"].astype(np.uint64).sum()))
        ref = ref or key
        assert key == ref, "variant %r changed the result" % (v,)
        ms[v].append(res.search_kernel_ms)
ab = res.algorithmic_bytes
print("workload %s  algorithmic bytes/launch %.3f GB" % (wl, ab / 1e9))
for v in sorted(variants, key=lambda v: np.median(ms[v])):
    m = np.array(ms[v][1:])
    print("vec=%d unroll=%2d nt=%d ldsKB=%3d waves/block=%d  median %.4f ms  min %.4f ms  -> %.0f GB/s (median)" % (v + (np.median(m), m.min(), ab / np.median(m) / 1e6)))
