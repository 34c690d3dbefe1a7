#!/usr/bin/env python3
"""Generic in-process A/B: values of one kernel-selection knob of the context (kwage_ctx_set_tuning), interleaved
rounds on one resident workload.   python tools/tune_knob.py KNOB v1,v2,... [workload] [rounds] [other_knob=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

var, values = sys.argv[1], sys.argv[2].split(",")
wl = sys.argv[3] if len(sys.argv) > 3 else "c2"
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 9
ctx = ka.Context(0)
for kv in sys.argv[5:]:
    ctx.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
s = synth.build(ctx, synth.WORKLOADS[wl])
ms = {v: [] for v in values}
ref = None
for r in range(rounds):
    for v in values:
        ctx.set_tuning(var, int(v))
        res = s.group.search(s.batch, s.workload.threshold, ka.SEARCH_TIMING)
        key = (len(res.hits), int(res.hits["column"].astype(np.uint64).sum()), int(res.hits["num_match"].astype(np.uint64).sum()))
        ref = ref or key
        assert key == ref, "value %r changed the result" % (v,)
        ms[v].append(res.search_kernel_ms)
ab = res.algorithmic_bytes
print("workload %s  %s  algorithmic bytes/launch %.3f GB, %d rounds" % (wl, res.search_kernel, ab / 1e9, rounds))
for v in sorted(values, key=lambda v: np.median(ms[v][1:])):
    m = np.array(ms[v][1:])
    print("%s=%s  median %.4f ms  min %.4f ms  max %.4f ms -> %.0f GB/s (median)" % (var, v, np.median(m), m.min(), m.max(), ab / np.median(m) / 1e6))
