#!/usr/bin/env python3
"""and_walk_kernel (the walk knob = rows in flight) against the tiled and_kernel (walk = 0) on one resident workload,
interleaved rounds in ONE process (the same 105 GB allocation: separate processes differ by +-3 % from
physical placement alone).   python tools/tune_walk.py [workload] [rounds] [values]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 9
values = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1", "2", "3", "4"]
ctx = ka.Context(0)
s = synth.build(ctx, synth.WORKLOADS[wl])
ms = {v: [] for v in values}
ref = None
for r in range(rounds):
    for v in values:
        ctx.set_tuning("walk", int(v))
        res = s.group.search(s.batch, s.workload.threshold, ka.SEARCH_TIMING)
        key = (len(res.hits), int(res.hits["column"].astype(np.uint64).sum()), int(res.hits["query"].astype(np.uint64).sum()))
        ref = ref or key
        assert key == ref, "variant %r changed the result" % (v,)
        ms[v].append(res.search_kernel_ms)
ab = res.algorithmic_bytes
print("workload %s  algorithmic bytes/launch %.3f GB, %d rounds" % (wl, ab / 1e9, rounds))
for v in sorted(values, key=lambda v: np.median(ms[v][1:])):
    m = np.array(ms[v][1:])
    print("walk=%s  median %.4f ms  min %.4f ms  max %.4f ms -> %.0f GB/s (median)" % (v, np.median(m), m.min(), m.max(), ab / np.median(m) / 1e6))
