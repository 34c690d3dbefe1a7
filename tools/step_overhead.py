#!/usr/bin/env python3
"""Where does a C2 step spend its time outside the gather kernel?  python tools/step_overhead.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import native, synth

ctx = ka.Context(0)
s = synth.build(ctx, synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"])
L = native.lib()
thr = C.c_float(s.workload.threshold)
for flags in (ka.SEARCH_TIMING | ka.SEARCH_TIMING_KMER, ka.SEARCH_TIMING, 0):
    walls, ks, kms = [], [], []
    for i in range(30):
        res = C.POINTER(native.Result)()
        t0 = time.perf_counter()
        native.check(L.kwage_search(s.group._h, s.batch._h, thr, flags, C.byref(res)))
        t1 = time.perf_counter()
        walls.append((t1 - t0) * 1e3); ks.append(res.contents.search_kernel_ms); kms.append(res.contents.kmer_kernel_ms)
        L.kwage_result_free(res)
    w, k, km = np.median(walls[5:]), np.median(ks[5:]), np.median(kms[5:])
    print("flags=%d  raw C-ABI call wall %.4f ms | search kernel %.4f | kmer stage %.4f | rest %.4f" % (flags, w, k, km, w - k - km))
walls = []
for i in range(30):
    t0 = time.perf_counter()
    r = s.group.search(s.batch, s.workload.threshold, ka.SEARCH_TIMING)
    walls.append((time.perf_counter() - t0) * 1e3)
print("python Group.search wall %.4f ms" % np.median(walls[5:]))
