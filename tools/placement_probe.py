#!/usr/bin/env python3
"""Does WHERE the matrix lies in device memory change the gather kernel's rate?  One process, one workload, several
placements: before each build of the matrix a dummy allocation of a given size is made (and kept), so the matrix starts
at another device address / on other physical pages; every placement is timed with the same queries.

    python tools/placement_probe.py [workload] [pad_mb,pad_mb,...] [searches]

Background: the same kernel binary measured 1.89 and 2.02 ms at C2's shape in two processes on one box
(profiles/r03_placement_probe.txt)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
pads = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,0,2,33,1024,5000,0").split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 9

ctx = ka.Context(0)
for pad in pads:
    dummy = ka.Group(ctx, 31, 1, 10, pad * 1024 * 8) if pad else None        # 1024 rows of `pad` KiB: a device block of `pad` MiB
    s = synth.build(ctx, synth.WORKLOADS[wl])
    ms = []
    for _ in range(n):
        res = s.group.search(s.batch, s.workload.threshold, ka.SEARCH_TIMING)
        ms.append(res.search_kernel_ms)
    m = np.array(ms[1:])
    print("pad %6d MB  %s  median %.4f ms  min %.4f  max %.4f -> %.0f GB/s" %
          (pad, res.search_kernel, np.median(m), m.min(), m.max(), res.algorithmic_bytes / np.median(m) / 1e6), flush=True)
    s.batch.close()
    s.group.close()
    if dummy is not None:
        dummy.close()
