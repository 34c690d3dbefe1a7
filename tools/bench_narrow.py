#!/usr/bin/env python3
"""Narrow databases (rows of a few hundred bytes): how far below the HBM roofline does the one-wave-per-
(query, tile) mapping fall when a row is narrower than one wave-load?   python tools/bench_narrow.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

ctx = ka.Context(0)
for ncol, L, nq, qlen in ((2048, 25, 100_000, 150), (2048, 25, 10_000, 1000), (8192, 24, 100_000, 150), (32768, 22, 100_000, 150)):
    w = synth.Workload("narrow", ncol, L, 31, 1, nq, qlen, 1.0, num_genomes=16, genome_len=200_000,
                       hit_fraction=float(os.environ.get("KWAGE_BENCH_HIT_FRACTION", "0.5")))
    s = synth.build(ctx, w)
    for thr in (1.0, 0.8):
        best = None
        for _ in range(4):
            r = s.group.search(s.batch, thr, ka.SEARCH_TIMING)
            best = r.search_kernel_ms if best is None else min(best, r.search_kernel_ms)
        print("N=%6d (row %5d B) L=%d  %6d x %4d bp  t=%.1f  %s kernel %.3f ms  algorithmic %.0f GB/s  %.1f G bit-tests/s  hits %d"
              % (ncol, (ncol + 7) // 8, L, nq, qlen, thr, r.search_kernel, best, r.algorithmic_bytes / best / 1e6, r.bit_tests / best / 1e6, len(r.hits)), flush=True)
    s.batch.close(); s.group.close()
