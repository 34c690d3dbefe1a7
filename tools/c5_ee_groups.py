#!/usr/bin/env python3
"""C5's eight filter-size groups one by one: the gather stage without and with early exit (kernel chosen, time from the
events riding on the launches, hand-over list use), to see which group the early-exit time of the whole share is spent in.

    python tools/c5_ee_groups.py [--workload c5] [--reps 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c5")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import kwage_amd as ka
    from kwage_amd import synth
    import bench
    ctx = ka.Context(0)
    w = synth.WORKLOADS[args.workload]
    groups = getattr(synth, bench.MULTI_GROUPS[args.workload])
    members = synth.build_multi(ctx, groups, w, seed=1, column_seed=0)
    batch = members[0].batch
    print("# %s, code %s" % (w.name, bench.kernel_code_hash()))
    tot = [0.0, 0.0]
    for (lg, ns), m in zip(groups, members):
        row = []
        for fl in (ka.SEARCH_TIMING, ka.SEARCH_TIMING | ka.SEARCH_EARLY_EXIT):
            best, name, hits = None, None, None
            for _ in range(args.reps):
                r = m.group.search(batch, w.threshold, fl)
                if best is None or r.search_kernel_ms < best:
                    best, name, hits = r.search_kernel_ms, r.search_kernel, len(r.hits)
            row.append((best, name, hits))
        st = [x for x in ctx.refine_stats() if x.get("units_cap")]
        tot[0] += row[0][0]
        tot[1] += row[1][0]
        print("2^%d rows x %7d samples (%6.1f GB): nominal %9.3f ms %-28s | early exit %9.3f ms %-40s = %.2fx | hits %d / %d | lists %s"
              % (lg, ns, m.group.device_bytes / 1e9, row[0][0], row[0][1], row[1][0], row[1][1], row[0][0] / row[1][0], row[0][2], row[1][2],
                 st[-1] if st else None), flush=True)
    print("sum: nominal %.3f ms, early exit %.3f ms = %.2fx" % (tot[0], tot[1], tot[0] / tot[1]))


if __name__ == "__main__":
    main()
