import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kwage_amd as ka
from kwage_amd import synth
ctx = ka.Context(0)
s = synth.build(ctx, synth.WORKLOADS["c2"])
fl = ka.SEARCH_TIMING
for _ in range(3): s.group.search(s.batch, 1.0, fl)
ctx.sync()
ts = []
t0 = time.perf_counter()
pend = s.group.submit(s.batch, 1.0, fl); ts.append(("submit", time.perf_counter() - t0))
for i in range(12):
    a = time.perf_counter(); nxt = s.group.submit(s.batch, 1.0, fl); b = time.perf_counter()
    r = pend.collect(); c = time.perf_counter()
    ts.append(("submit %.0f us  collect %.0f us  kernel %.0f us" % ((b - a) * 1e6, (c - b) * 1e6, r.search_kernel_ms * 1e3), c - t0))
    pend = nxt
r = pend.collect()
print("total %.3f ms for 13 steps -> %.3f ms/step" % ((time.perf_counter() - t0) * 1e3, (time.perf_counter() - t0) * 1e3 / 13))
for x in ts: print(x)
