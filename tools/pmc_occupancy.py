#!/usr/bin/env python3
"""What caps a gather kernel that sits below the others: occupancy / waiting, from one `rocprofv3 --pmc` pass of SQ and
GRBM counters per workload (separate from the FETCH_SIZE pass of tools/pmc_refresh.py: counters in their own run).

    python tools/pmc_occupancy.py [--round r05] [workload ...]        (default: c2 narrow narrowt long1t)

Per gather kernel: waves launched, mean waves resident per CU (SQ_WAVE_CYCLES is in quad-cycles, summed over waves;
GRBM_GUI_ACTIVE is summed over the 8 XCDs), the share of wave time spent parked in s_waitcnt (SQ_WAIT_ANY), issuing
(SQ_ACTIVE_INST_ANY), and the effective clock.  Writes profiles/<round>_<workload>_pmc_occupancy.json (and a copy under
gpurun_out/).  This script never touches the GPU itself: every pass is a child process."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

COUNTERS = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
GATHER = ("and_kernel", "and_walk_kernel", "and_band_walk_kernel", "and_narrow_kernel", "count_kernel", "count_walk_kernel", "count_narrow_kernel",
          "and_screen_kernel", "and_refine_kernel", "and_refine_emit_kernel", "count_screen_kernel", "count_refine_kernel", "count_refine_emit_kernel")


def main():
    args = sys.argv[1:]
    rnd = "r05"
    if args[:1] == ["--round"]:
        rnd, args = args[1], args[2:]
    workloads = args or ["c2", "narrow", "narrowt", "long1t"]
    out_root = os.path.join(ROOT, "gpurun_out", "pmc_" + rnd)
    os.makedirs(out_root, exist_ok=True)
    for wl in workloads:
        d = os.path.join(out_root, wl + "_occupancy")
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        cmd = ["rocprofv3", "--pmc"] + COUNTERS + ["--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--no-sustained", "--no-result-check", "--also", "none", "--steps", "5", "--warmup", "1"]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        print("[pmc_occupancy] %s: rc %d in %.0f s" % (wl, r.returncode, time.time() - t0), flush=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        csvs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not lines or not csvs:
            print(r.stderr[-2000:])
            continue
        line = json.loads(lines[-1])
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(sorted(csvs)[-1])):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        ncu = 256
        out = {"workload": wl, "kernel": line["roofline"]["kernel"], "kernel_ms_under_pmc": line["roofline"]["kernel_ms"], "frac_under_pmc": line["roofline"]["frac"],
               "counters": COUNTERS, "kernels": {}}
        for name, c in per.items():
            if not any(("kwage::" + g + "<") in name for g in GATHER):
                continue
            m = {k: sum(v) / len(v) for k, v in c.items()}
            gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0                     # cycles the kernel was active (mean over the XCDs)
            wave_cycles = 4.0 * m.get("SQ_WAVE_CYCLES", 0)              # quad-cycles -> cycles, summed over all waves
            e = {"dispatches": len(next(iter(c.values()))), "waves": m.get("SQ_WAVES"),
                 "mean_waves_resident_per_cu": round(wave_cycles / (gui * ncu), 2) if gui else None,
                 "share_of_wave_time_waiting": round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 4) if m.get("SQ_WAVE_CYCLES") else None,
                 "share_of_wave_time_issue_stalled": round(m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 4) if m.get("SQ_WAVE_CYCLES") else None,
                 "share_of_wave_time_issuing": round(m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 4) if m.get("SQ_WAVE_CYCLES") else None,
                 "active_cycles": round(gui), "raw_means": {k: round(v, 1) for k, v in m.items()}}
            out["kernels"][name.split("(")[0].replace("void ", "").replace("kwage::", "").replace(" ", "")] = e
            print("[pmc_occupancy] %s %s" % (wl, json.dumps({k: v for k, v in e.items() if k != "raw_means"})), flush=True)
        path = os.path.join(ROOT, "profiles", "%s_%s_pmc_occupancy.json" % (rnd, wl))
        json.dump(out, open(path, "w"), indent=1)
        shutil.copyfile(path, os.path.join(out_root, os.path.basename(path)))


if __name__ == "__main__":
    main()
