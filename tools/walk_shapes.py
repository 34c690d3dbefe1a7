#!/usr/bin/env python3
"""Launch shape of the persistent kernels when they fill the chip: ONE workgroup of 8 waves per CU (a dynamic-LDS pad keeps a
second workgroup off: every CU runs exactly 8 waves) against workgroups of 4 waves placed by the dispatcher.  (Background,
tools/micro/power_probe.hip: how many CUs read does matter for a sequential stream -- 192 CUs x 16 waves 7.2 TB/s, all 256
CUs 6.9 -- but not for the random-row gather, 6.7 TB/s of touched bytes on 160...256 CUs.)
Interleaved rounds in ONE process on one allocation.   python tools/walk_shapes.py [workload] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 9
ctx = ka.Context(0)
s = synth.build(ctx, synth.WORKLOADS[wl])
variants = [("one WG of 8 waves per CU", dict(walk_one_wg_per_cu=1)), ("WGs of 4 waves, dispatcher's placement", dict(walk_one_wg_per_cu=0))]
ms = {n: [] for n, _ in variants}
ref = None
for r in range(rounds):
    for n, knobs in variants:
        with ctx.tuning(**knobs):
            res = s.group.search(s.batch, s.workload.threshold, ka.SEARCH_TIMING)
        key = (len(res.hits), int(res.hits["column"].astype(np.uint64).sum()), int(res.hits["query"].astype(np.uint64).sum()))
        ref = ref or key
        assert key == ref, "variant %r changed the result" % (n,)
        ms[n].append(res.search_kernel_ms)
ab = res.algorithmic_bytes
print("workload %s  %s  algorithmic bytes/launch %.3f GB, %d rounds" % (wl, res.search_kernel, ab / 1e9, rounds))
for n, _ in variants:
    m = np.array(ms[n][1:])
    print("  %-40s median %.4f ms  min %.4f  max %.4f -> %.0f GB/s (median)" % (n, np.median(m), m.min(), m.max(), ab / np.median(m) / 1e6), flush=True)
print("stream read of this box: %.0f GB/s" % s.group.stream_read_gbps(8 << 30, 3))
