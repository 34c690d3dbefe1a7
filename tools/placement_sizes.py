#!/usr/bin/env python3
"""Contiguous or plain device memory for the matrix, by matrix size and row width: one matrix per process (the policy is
read once per process), t = 1, 100 k x 150 bp queries.   python tools/placement_sizes.py COLUMNS LOG2_ROWS"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

ncol, L = int(sys.argv[1]), int(sys.argv[2])
ctx = ka.Context(0)
w = synth.Workload("sizes", ncol, L, 31, 1, 100_000, 150, 1.0, num_genomes=16, genome_len=200_000)
s = synth.build(ctx, w)
ms = []
for _ in range(7):
    r = s.group.search(s.batch, 1.0, ka.SEARCH_TIMING)
    ms.append(r.search_kernel_ms)
m = float(np.median(ms[1:]))
print("contiguous=%s  %7d columns (row %6d B) x 2^%d rows = %6.1f GB  %-24s median %8.3f ms  %5.0f GB/s" %
      (os.environ.get("KWAGE_GROUP_CONTIGUOUS", "1"), ncol, (ncol + 7) // 8, L, s.group.device_bytes / 1e9, r.search_kernel, m, r.algorithmic_bytes / m / 1e6), flush=True)
