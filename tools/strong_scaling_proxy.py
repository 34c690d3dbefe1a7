#!/usr/bin/env python3
"""One-GPU proxy of the STRONG-scaling curve ("narrower rows at fixed N" is the limiter).

For C3 (1 M samples, 125 KB rows) and C2 (100 k samples, 12.5 KB rows) the per-GPU shape of a K-way column split --
what rank 0 of `bench.py --gpus K --scaling strong` would hold, K = 1, 2, 4, 8 -- is measured on ONE GPU
(`bench.py --share-of K`), each in a fresh process under `rocprofv3 --kernel-trace --stats`: kernel chosen, its
roofline fraction from the HIP events AND rocprofv3's own average for it, the step time, and what they imply for 1 -> K:

    speedup(K)    = ms_per_step(1) / ms_per_step(K)        (a rank's step; the exchange adds < 1 %)
    efficiency(K) = speedup(K) / K

Rank 0's share is the widest one of the split (partition_columns gives the first ranks the extra 1024-column unit), so
it is the rank the job waits for.

    python tools/strong_scaling_proxy.py [--round r05] [--steps 20] [--no-rocprof] [c3 c2 ...]

Writes profiles/<round>_strong_scaling_proxy.json and profiles/<round>_proxy_<workload>_of<K>_kernel_stats.csv.
This script never touches the GPU itself: every measurement is a child process."""
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
import bench  # noqa: E402  (pure Python at import time)

SPLITS = (1, 2, 4, 8)


def kernel_avg_ms(stats_csv, kernel):
    """rocprofv3's average duration (ms) and call count of the gather kernel `kernel` ('and_walk_kernel<13,4>')."""
    want = kernel.split("<")[0]
    best = None
    for row in csv.DictReader(open(stats_csv)):
        name = row.get("Name", "")
        if ("kwage::" + want + "<") in name:
            avg, calls = float(row["AverageNs"]) / 1e6, int(row["Calls"])
            if best is None or calls * avg > best[0] * best[1]:
                best = (avg, calls, name)
    return best


def main():
    args = sys.argv[1:]
    rnd, steps, rocprof = "r04", 20, True
    while args and args[0].startswith("--"):
        if args[0] == "--round":
            rnd, args = args[1], args[2:]
        elif args[0] == "--steps":
            steps, args = int(args[1]), args[2:]
        elif args[0] == "--no-rocprof":
            rocprof, args = False, args[1:]
        else:
            raise SystemExit("unknown option " + args[0])
    workloads = args or ["c3", "c2"]
    out_root = os.path.join(ROOT, "gpurun_out", "proxy_" + rnd)
    table = {"_about": "one-GPU proxy of strong scaling: per-GPU share of a K-way column split of each workload (bench.py --share-of K); "
                       "speedup = ms_per_step(1)/ms_per_step(K), efficiency = speedup/K; tools/strong_scaling_proxy.py",
             "code_hash": bench.kernel_code_hash(), "steps": steps}
    for wl in workloads:
        rows = []
        for k in SPLITS:
            d = os.path.join(out_root, "%s_of%d" % (wl, k))
            shutil.rmtree(d, ignore_errors=True)
            os.makedirs(d, exist_ok=True)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--share-of", str(k), "--also", "none",
                   "--no-cpu-baseline", "--no-sustained", "--steps", str(steps), "--warmup", "3"]
            if rocprof:
                cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + cmd
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            print("[proxy] %s of %d: rc %d in %.0f s" % (wl, k, r.returncode, time.time() - t0), flush=True)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
            if r.returncode != 0 or not lines:
                print(r.stderr[-3000:], flush=True)
                rows.append({"split": k, "error": "rc %d" % r.returncode})
                continue
            line = json.loads(lines[-1])
            rf = line["roofline"]
            row = {"split": k, "samples_per_gpu": line["config"]["samples_per_gpu"], "row_bytes": line["config"]["row_bytes"],
                   "db_bytes_per_gpu": line["config"]["db_bytes_per_gpu"], "kernel": rf["kernel"],
                   "ms_per_step": line["ms_per_step"], "kernel_ms": rf["kernel_ms"], "achieved_gbps": rf["achieved"], "frac": rf["frac"],
                   "frac_of_measured_stream": rf["frac_of_measured_stream"], "value_g_bit_tests_per_s": line["value"],
                   "result_check_ok": bool(line.get("result_check", {}).get("ok"))}
            if rocprof:
                stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
                if stats:
                    got = kernel_avg_ms(sorted(stats)[-1], rf["kernel"])
                    if got:
                        row["rocprof_kernel_avg_ms"], row["rocprof_calls"], row["rocprof_kernel"] = round(got[0], 4), got[1], got[2].split("(")[0]
                        row["rocprof_frac"] = round(rf["algorithmic_bytes_per_launch"] / (got[0] * 1e-3) / 1e9 / bench.HBM_PEAK_GBPS, 4)
                    dst = os.path.join(ROOT, "profiles", "%s_proxy_%s_of%d_kernel_stats.csv" % (rnd, wl, k))
                    shutil.copyfile(sorted(stats)[-1], dst)
                    shutil.copyfile(dst, os.path.join(out_root, os.path.basename(dst)))      # (gpurun_out/ is what comes back from a GPU box)
                    row["kernel_stats"] = os.path.relpath(dst, ROOT)
            rows.append(row)
        base = next((x for x in rows if x.get("split") == 1 and "error" not in x), None)
        for x in rows:
            if base and "error" not in x:
                x["speedup_vs_1"] = round(base["ms_per_step"] / x["ms_per_step"], 3)
                x["efficiency"] = round(base["ms_per_step"] / x["ms_per_step"] / x["split"], 3)
                x["kernel_only_efficiency"] = round(base["kernel_ms"] / x["kernel_ms"] / x["split"], 3)
        table[wl] = rows
        for x in rows:
            print("[proxy] %s" % json.dumps(x), flush=True)
    path = os.path.join(ROOT, "profiles", "%s_strong_scaling_proxy.json" % rnd)
    with open(path, "w") as fh:
        json.dump(table, fh, indent=1)
        fh.write("\n")
    shutil.copyfile(path, os.path.join(out_root, os.path.basename(path)))
    print("[proxy] wrote " + os.path.relpath(path, ROOT))


if __name__ == "__main__":
    main()
