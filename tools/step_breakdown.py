#!/usr/bin/env python3
"""Whole synchronous search step (raw C-ABI call) against its two device stages for query shapes other than the
bench's: where is time spent outside the gather kernel?   python tools/step_breakdown.py [substring of a shape's name]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import native, synth

ctx = ka.Context(0)
L = native.lib()
base = synth.Workload("shape", 100_000, 20, 31, 1, 1, 1, 1.0, num_genomes=8, genome_len=200_000)
shapes = [("1000 x 1 kb", 1000, 1000), ("100k x 150 bp", 100_000, 150), ("1M x 50 bp", 1_000_000, 50), ("1 x 10 kb", 1, 10_000),
          ("10 x 100 kb", 10, 100_000), ("20k x 40 bp (k+9)", 20_000, 40), ("200 x 5 kb", 200, 5000)]
from dataclasses import replace
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, nq, qlen in shapes:
    if only not in name:
        continue
    w = replace(base, num_queries=nq, query_len=qlen)
    s = synth.build(ctx, w)
    for thr_v, ee in ((1.0, 0), (1.0, ka.SEARCH_EARLY_EXIT), (0.8, 0), (0.8, ka.SEARCH_EARLY_EXIT), (0.0001, 0)):
        if thr_v == 0.0001 and nq * 100_000 > 200_000_000:
            continue                  # every column matches every query: keep the hit list below 200 M records
        thr = C.c_float(thr_v)
        flags = ka.SEARCH_TIMING | ka.SEARCH_TIMING_KMER | ee
        walls, ks, kms, nh = [], [], [], 0
        for i in range(6):
            res = C.POINTER(native.Result)()
            t0 = time.perf_counter()
            native.check(L.kwage_search(s.group._h, s.batch._h, thr, flags, C.byref(res)))
            walls.append((time.perf_counter() - t0) * 1e3); ks.append(res.contents.search_kernel_ms); kms.append(res.contents.kmer_kernel_ms)
            nh = res.contents.n_hits; kern = (res.contents.search_kernel or b"").decode()
            L.kwage_result_free(res)
        wl, k, km = np.median(walls[2:]), np.median(ks[2:]), np.median(kms[2:])
        print("%-18s t=%-6g %s call %9.3f ms | gather %9.3f (%s) | k-mer stage %7.3f | rest %8.3f | hits %d%s"
              % (name, thr_v, "ee" if ee else "  ", wl, k, kern, km, wl - k - km, nh, (" | lists " + str(ctx.refine_stats()[0])) if "screen" in kern else ""), flush=True)
    s.batch.close(); s.group.close()
