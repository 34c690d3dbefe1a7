#!/usr/bin/env python3
"""Regenerate EVERY entry of profiles/pmc_traffic.json in one GPU session: per workload a separate
`rocprofv3 --pmc FETCH_SIZE` pass of the bench command (counters in their own run, no tracing), condensed by
tools/pmc_summary.py's rule (FETCH_SIZE is in KiB and counts half the bytes of a 16 B/lane read stream on gfx950; the
factor is CALIBRATED in each pass on stream_read_kernel's known byte count), written with the hash of the kernel sources +
engine the pass ran on -- bench.py reports an entry as roofline.traffic only while that hash still matches.

    python tools/pmc_refresh.py [--round r03] [workload ...]        (default: c2 c2+bands c2t c3 c4 c5s c5; also: narrow narrowt long1t)

This script never touches the GPU itself: every pass is a child process (`rocprofv3 ... -- python3 bench.py ...`)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (pure Python at import time: no torch, no HIP)

N_GROUPS = {"c5": 8}
GATHER = ("and_kernel", "and_walk_kernel", "and_band_walk_kernel", "and_narrow_kernel", "count_kernel", "count_walk_kernel", "count_narrow_kernel",
          "and_screen_kernel", "and_refine_kernel", "and_refine_emit_kernel", "count_screen_kernel", "count_refine_kernel", "count_refine_emit_kernel")


def short(kernel_name):
    """'void kwage::and_walk_kernel<13, 4>(...)' -> 'and_walk_kernel<13,4>' (bench.py's spelling, without bools)"""
    k = kernel_name.split("(")[0].replace("void ", "").replace("kwage::", "").replace(" ", "")
    return k


def main():
    args = sys.argv[1:]
    rnd = "r05"
    if args[:1] == ["--round"]:
        rnd, args = args[1], args[2:]
    # "c2+bands": the same workload with the walk kernel forced band after band (its entry is keyed "c2@<kernel>")
    # "c2+ee": the same workload searched with the reference's early exit (`bench.py --early-exit`; entry "c2@ee", read by
    # the `early_exit` block of the driver's line): the bytes of ALL the stage's launches (screen + refine + emit) per step
    workloads = args or ["c2", "c2+bands", "c2t", "c3", "c4", "c5s", "c5"]
    out_root = os.path.join(ROOT, "gpurun_out", "pmc_" + rnd)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    rec = json.load(open(path))
    code = bench.kernel_code_hash()
    for spec in workloads:
        wl, banded, ee = spec.split("+")[0], spec.endswith("+bands"), spec.endswith("+ee")
        knobs = {"KWAGE_WALK_BANDS": "3", "KWAGE_WALK_BANDS_MIN_GIB": "0"} if banded else {"KWAGE_WALK_BANDS": "0"}
        d = os.path.join(out_root, spec.replace("+", "_"))
        os.makedirs(d, exist_ok=True)
        cmd = ["rocprofv3", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--no-sustained", "--no-result-check", "--no-early-exit-block", "--also", "none", "--steps", "5", "--warmup", "1"] + (["--early-exit"] if ee else [])
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", **knobs))
        print("[pmc_refresh] %s: rc %d in %.0f s" % (wl, r.returncode, time.time() - t0), flush=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if r.returncode != 0 or not lines:
            print(r.stderr[-2000:])
            continue
        line = json.loads(lines[-1])
        csvs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not csvs:
            print("[pmc_refresh] %s: no counter_collection.csv under %s" % (wl, d))
            continue
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(sorted(csvs)[-1])):
            if row["Counter_Name"] == "FETCH_SIZE":
                agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        # calibration: stream_read_kernel reads min(matrix, 8 GiB) per dispatch (kwage_stream_read_gbps)
        cal = [v for k, vs in agg.items() if "stream_read_kernel" in k for v in vs]
        n16 = min(int(line["config"]["db_bytes_per_gpu"]), 8 << 30) // 16          # (kwage_stream_read_gbps: whole 8 KiB steps per wave)
        nwaves = 4 * max(1, min(2048, n16 // 512 // 4))
        known = nwaves * ((n16 // nwaves) // 512 * 512) * 16
        factor = (known / (sum(cal) / len(cal) * 1024.0)) if cal else 2.0
        gather = {k: vs for k, vs in agg.items() if any(("kwage::" + g + "<") in k for g in GATHER)}
        if not gather:
            print("[pmc_refresh] %s: no gather kernel in the counter file" % wl)
            continue
        ng = N_GROUPS.get(wl, 1)
        dominant = max(gather, key=lambda k: sum(gather[k]))
        # (a stage of several launches -- screen or truncated walk, then refine + emit -- counts once per step and group: the
        # dispatches of its FIRST kernel, which is the one gather kernel of the stage that is not a refine launch)
        steps = sum(len(v) for k, v in gather.items() if "_refine" not in k) / float(ng)
        per_step = sum(sum(v) for v in gather.values()) * 1024.0 * factor / steps
        summary = {"workload": wl, "bench_line": line, "calibration": {"stream_read_dispatches": len(cal), "known_bytes": known, "factor": factor},
                   "kernels": {short(k): {"dispatches": len(v), "FETCH_SIZE_avg_KiB": sum(v) / len(v)} for k, v in agg.items()}}
        sfile = os.path.join(ROOT, "profiles", "%s_%s_pmc_fetch_size.json" % (rnd, spec.replace("+", "_")))
        json.dump(summary, open(sfile, "w"), indent=1)
        entry = {"kernel": line["roofline"]["kernel"], "profiled_kernels": {short(k): len(v) for k, v in gather.items()},
                 "hbm_read_bytes_per_launch": int(per_step), "algorithmic_bytes_per_launch": int(line["roofline"]["algorithmic_bytes_per_launch"]),
                 "ratio": round(per_step / line["roofline"]["algorithmic_bytes_per_launch"], 4), "correction_factor_calibrated": round(factor, 4),
                 "source": os.path.relpath(sfile, ROOT), "code_hash": code, "round": rnd}
        if ng > 1:
            entry["note"] = "per STEP = the %d groups' launches together, as bench.py sums kernel time and algorithmic bytes over the groups" % ng
        if ee:
            entry["per_kernel_bytes_per_step"] = {short(k): int(sum(v) * 1024.0 * factor / steps) for k, v in gather.items()}
        rec[("%s@%s" % (wl, entry["kernel"])) if banded else ("%s@ee" % wl if ee else wl)] = entry
        print("[pmc_refresh] %s: %s  %.3f GB per step for %.3f GB algorithmic = %.4fx (factor %.4f)" %
              (wl, short(dominant), per_step / 1e9, entry["algorithmic_bytes_per_launch"] / 1e9, entry["ratio"], factor), flush=True)
        json.dump(rec, open(path, "w"), indent=1)
        # (gpurun_out/ is what comes back from a GPU box)
        import shutil
        shutil.copyfile(path, os.path.join(out_root, "pmc_traffic.json"))
        shutil.copyfile(sfile, os.path.join(out_root, os.path.basename(sfile)))


if __name__ == "__main__":
    main()
