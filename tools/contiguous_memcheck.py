#!/usr/bin/env python3
"""Does a physically contiguous matrix cost more device memory than its size?  Free memory before / after groups of
105, 164 and 193 GB (hipExtMallocWithFlags(hipDeviceMallocContiguous) does not round up).   python tools/contiguous_memcheck.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import kwage_amd as ka
ctx = ka.Context(0)
f0, t = ctx.mem_info()
print("free %.2f GB of %.2f" % (f0/1e9, t/1e9))
for L, cols in ((23, 100_000), (20, 1_250_000), (25, 46_000)):
    g = ka.Group(ctx, 31, 1, L, cols)
    f1, _ = ctx.mem_info()
    print("group 2^%d x %d: %.2f GB matrix, free dropped by %.2f GB" % (L, cols, g.device_bytes/1e9, (f0-f1)/1e9))
    g.close()
