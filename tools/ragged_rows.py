#!/usr/bin/env python3
"""What does the ragged last KiB-step of a row cost the walk kernels?  C2's rows are 12 500 bytes = 12 KiB + 212 bytes: the
13th step of a row moves 212 bytes.  The same batch against 98 304 samples (rows of exactly 12 KiB) and against 100 000,
AND path and count path, in one process.   python tools/ragged_rows.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ctx = ka.Context(0)
for wl in ("c2e", "c2", "c2et", "c2t"):
    s = synth.build(ctx, synth.WORKLOADS[wl])
    ms = []
    for r in range(rounds):
        res = s.group.search(s.batch, s.workload.threshold, ka.SEARCH_TIMING)
        ms.append(res.search_kernel_ms)
    m = np.array(ms[1:])
    print("%-5s %-28s row %6d B  median %.4f ms  min %.4f -> %.0f GB/s algorithmic (median), %.1f ns per row" %
          (wl, res.search_kernel, (s.workload.num_samples + 7) // 8, np.median(m), m.min(), res.algorithmic_bytes / np.median(m) / 1e6,
           np.median(m) * 1e6 / res.total_kmers), flush=True)
    s.batch.close(); s.group.close()
