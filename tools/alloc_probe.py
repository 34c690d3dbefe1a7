#!/usr/bin/env python3
"""Does the gather kernel's speed depend on WHICH physical memory the 105 GB matrix got?  Build the C2 group
several times in one process (free + allocate again) and time the same searches on each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kwage_amd as ka
from kwage_amd import synth

ctx = ka.Context(0)
w = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    s = synth.build(ctx, w)
    ms = [s.group.search(s.batch, w.threshold, ka.SEARCH_TIMING).search_kernel_ms for _ in range(12)]
    gbps = s.group.stream_read_gbps(min(s.group.device_bytes, 8 << 30), 3)
    print("allocation %d: gather kernel median %.4f ms (min %.4f max %.4f); stream read %.0f GB/s" % (rep, np.median(ms[2:]), min(ms[2:]), max(ms[2:]), gbps), flush=True)
    s.batch.close(); s.group.close()
