#!/usr/bin/env python3
"""The count path (threshold < 1) on one resident workload: the tiled count_kernel against the persistent
count_walk_kernel at several waves per CU, interleaved rounds in ONE process on one allocation; then batch sizes
around C2's (the tiled kernel's rounds of waves make some sizes slow).   python tools/tune_count.py [workload] [rounds] [sizes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kwage_amd as ka
from kwage_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c2t"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
sizes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
ctx = ka.Context(0)
s = synth.build(ctx, synth.WORKLOADS[wl])
thr = s.workload.threshold
variants = [("tiled", dict(count_walk=0))] + [("walk %d waves/CU" % w, dict(count_walk=1, count_walk_wpc=w, count_walk_min_rows=1)) for w in (6, 8, 12)]


def run(batch, label):
    ms = {n: [] for n, _ in variants}
    names = {}
    ref = None
    for r in range(rounds):
        for n, knobs in variants:
            with ctx.tuning(**knobs):
                res = s.group.search(batch, thr, ka.SEARCH_TIMING)
            key = (len(res.hits), int(res.hits["column"].astype(np.uint64).sum()), int(res.hits["num_match"].astype(np.uint64).sum()))
            ref = ref or key
            assert key == ref, "variant %r changed the result" % (n,)
            ms[n].append(res.search_kernel_ms)
            names[n] = res.search_kernel
    ab = res.algorithmic_bytes
    print("%s: algorithmic bytes/launch %.3f GB, %d rounds" % (label, ab / 1e9, rounds))
    for n, _ in variants:
        m = np.array(ms[n][1:])
        print("  %-20s %-30s median %.4f ms  min %.4f  max %.4f -> %.0f GB/s (median)" % (n, names[n], np.median(m), m.min(), m.max(), ab / np.median(m) / 1e6), flush=True)


run(s.batch, "workload %s" % wl)
if sizes:
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    extra = [acgt[rng.integers(0, 4, size=s.workload.query_len)].tobytes().decode() for _ in range(max(sizes))]
    for nq in sizes:
        b = ka.Batch(ctx, (s.queries + extra)[:nq])
        run(b, "%d queries" % nq)
        b.close()
print("stream read of this box: %.0f GB/s" % s.group.stream_read_gbps(8 << 30, 3))
