#!/usr/bin/env python3
"""Device bit-transpose database builder vs the reference's build_db() (oracle/_ref/ref_tool build)
on the same `.bloom` files.   python tools/bench_builder.py [n_filters] [log2_len] [noref] [copyref]"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import kwage_amd as ka
from kwage_amd import native
import kwage_oracle as oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tmp = tempfile.mkdtemp(prefix="kwage_build_", dir="/tmp")
try:
    rng = np.random.default_rng(3)
    paths = []
    for j in range(n):
        bits = rng.integers(0, 256, size=(1 << L) // 8, dtype=np.uint8) & rng.integers(0, 256, size=(1 << L) // 8, dtype=np.uint8)
        p = os.path.join(tmp, "f%05d.bloom" % j)
        oracle.write_bloom(p, 31, L, 1, oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%d" % (j + 1))), bits)
        paths.append(p)
    lst = os.path.join(tmp, "list.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    with ka.Context(0) as ctx:
        arr = (C.c_char_p * n)(*[p.encode() for p in paths])
        prm = native.Params(31, 1, L, 0)
        st = native.BuildStats()
        out = os.path.join(tmp, "gpu.db")
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            native.check(native.lib().kwage_build_db(ctx._h, out.encode(), C.byref(prm), arr, n, C.byref(st)))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    bits_total = n * (1 << L)
    print("device builder: %d filters x 2^%d bits: wall %.3f s (file I/O + CRC32 + transpose), transpose kernel %.3f ms "
          "= %.1f G bits/s, %.1f GB/s in+out" % (n, L, best, st.transpose_kernel_ms, bits_total / st.transpose_kernel_ms / 1e6,
                                                  2 * bits_total / 8 / st.transpose_kernel_ms / 1e6))
    if "copyref" in sys.argv[3:]:
        # what a kernel that writes as much as it reads can reach on this box: a device-to-device copy of one chunk's size
        import torch
        nbytes = min(bits_total // 8, 512 << 20)
        a = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
        a.fill_(3); b.copy_(a); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            b.copy_(a)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("device-to-device copy of %d MB: %.3f ms = %.1f GB/s in+out" % (nbytes >> 20, ms, 2 * nbytes / ms / 1e6))
    if os.access(oracle.REF_TOOL, os.X_OK) and "noref" not in sys.argv[3:]:
        ref = os.path.join(tmp, "ref.db")
        t0 = time.perf_counter()
        subprocess.check_call([oracle.REF_TOOL, "build", ref, "31", str(L), "1", lst])
        dt = time.perf_counter() - t0
        same = open(ref, "rb").read() == open(out, "rb").read()
        print("reference build_db (1 thread): wall %.2f s = %.3f G bits/s; output byte-identical: %s; speed-up (wall) %.1fx"
              % (dt, bits_total / dt / 1e9, same, dt / best))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
