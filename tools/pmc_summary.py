#!/usr/bin/env python3
"""Condense a rocprofv3 --pmc FETCH_SIZE[,WRITE_SIZE] counter_collection.csv into a per-kernel
summary (JSON + text), applying the gfx950 correction prescribed by MI355X_MICROARCH.md section HBM:
FETCH_SIZE is reported in KiB and counts exactly HALF of the bytes of a wide (16 B/lane) coalesced
read stream, so HBM read bytes = FETCH_SIZE * 1024 * 2.  The stream_read_kernel dispatches in the
same run (a known byte count) calibrate that factor.

    python tools/pmc_summary.py <counter_collection.csv> <workload> <out.json>
"""
import collections
import csv
import json
import sys


def main():
    src, workload, out = sys.argv[1:4]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(src)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary = {"workload": workload, "source": src, "kernels": {}}
    for k, ctrs in agg.items():
        name = k.split("(")[0].replace("void ", "")
        e = {"dispatches": max(len(v) for v in ctrs.values())}
        for c, v in ctrs.items():
            e[c + "_avg_KiB"] = sum(v) / len(v)
        if "FETCH_SIZE" in ctrs:
            f = sum(ctrs["FETCH_SIZE"]) / len(ctrs["FETCH_SIZE"])
            e["hbm_read_bytes_per_launch_corrected"] = int(f * 1024 * 2)
        summary["kernels"][name] = e
    json.dump(summary, open(out, "w"), indent=1)
    for k, e in summary["kernels"].items():
        print(k, e)


if __name__ == "__main__":
    main()
