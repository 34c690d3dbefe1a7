#!/usr/bin/env python3
"""Few long queries (the C1 shape scaled up): effect of splitting a query's k-mer list over waves.
   python tools/bench_long_query.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

ctx = ka.Context(0)
for name, w in (("1 x 100 kb vs 100k samples", synth.Workload("long1", 100_000, 20, 31, 1, 1, 100_000, 1.0, num_genomes=2, genome_len=200_000)),
                ("4 x 1 Mb vs 100k samples", synth.Workload("long4", 100_000, 20, 31, 1, 4, 1_000_000, 1.0, num_genomes=4, genome_len=1_000_000, hit_fraction=1.0))):
    s = synth.build(ctx, w)
    for thr in (1.0, 0.9):
        for force in ("1", None, "cw"):       # one wave per (query, tile) / segments / persistent count kernel with long part chains
            if force == "cw" and thr == 1.0:
                continue
            ctx.set_tuning("force_segs", int(force) if force == "1" else 0)
            ctx.set_tuning("count_walk", 1 if force == "cw" else 0)
            best = None
            for _ in range(3):
                r = s.group.search(s.batch, thr, ka.SEARCH_TIMING | ka.SEARCH_TIMING_KMER)
                best = r.search_kernel_ms if best is None else min(best, r.search_kernel_ms)
            print("%-28s t=%.1f segs=%-5s kernel %.3f ms  %.0f GB/s  hits %d   (k-mer stage %.3f ms)" % (name, thr, (force or "auto") + " " + r.search_kernel, best, r.algorithmic_bytes / best / 1e6, len(r.hits), r.kmer_kernel_ms))
    s.batch.close(); s.group.close()
