#!/usr/bin/env python3
"""Render the measurement tables from the files under profiles/ -- nothing in them is typed by hand.

    python tools/render_tables.py            writes profiles/TABLES.md, profiles/README.md and the generated block of DESIGN.md

Sources:
  profiles/rNN_*bench*.json           bench.py lines (one JSON object per file)          -> the per-round workload tables
  profiles/rNN_*kernel_stats.csv      rocprofv3 --kernel-trace --stats summaries         -> kernel averages beside the HIP-event times
  profiles/pmc_traffic.json           HBM bytes per launch from the --pmc passes         -> traffic / algorithmic
  profiles/rNN_strong_scaling_proxy.json, rNN_*_pmc_occupancy.json                      -> their own tables
  profiles/INDEX.json                 what every file is and the command that made it   -> profiles/README.md
DESIGN.md carries the newest round's table between the markers  <!-- GENERATED:measurements BEGIN/END -->.
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
GATHER = ("and_kernel", "and_walk_kernel", "and_band_walk_kernel", "and_narrow_kernel", "count_kernel", "count_walk_kernel", "count_narrow_kernel",
          "and_screen_kernel", "and_refine_kernel", "and_refine_emit_kernel", "count_screen_kernel", "count_refine_kernel", "count_refine_emit_kernel")


def load_line(path):
    txt = open(path).read().strip()
    try:
        return json.loads(txt)
    except Exception:
        for ln in reversed(txt.splitlines()):
            if ln.startswith("{") and '"metric"' in ln:
                return json.loads(ln)
    return None


def short_kernel(name):
    return name.split("(")[0].replace("void ", "").replace("kwage::", "").replace(" ", "")


def kernel_stats(path):
    """-> [(short kernel name, calls, average ms)] for the gather kernels of one rocprofv3 stats file, busiest first."""
    out = []
    for row in csv.DictReader(open(path)):
        name = row.get("Name", "")
        if any(("kwage::" + g + "<") in name for g in GATHER):
            out.append((short_kernel(name), int(row["Calls"]), float(row["AverageNs"]) / 1e6, float(row["TotalDurationNs"])))
    out.sort(key=lambda x: -x[3])
    return [(a, b, c) for a, b, c, _ in out]


def fmt(x, nd=3):
    return "—" if x is None else (("%." + str(nd) + "f") % x)


def bench_rows(rnd):
    rows = []
    for path in sorted(glob.glob(os.path.join(PROF, rnd + "_*.json"))):
        base = os.path.basename(path)
        if "pmc" in base or "proxy" in base:
            continue
        d = load_line(path)
        if not isinstance(d, dict) or "roofline" not in d:
            continue
        rows.append((base, d))
        for name, blk in (d.get("also") or {}).items():
            if isinstance(blk, dict) and "roofline" in blk:
                rows.append((base + " › also." + name, blk))
    return rows


def bench_table(rnd):
    pmc = json.load(open(os.path.join(PROF, "pmc_traffic.json")))
    out = ["| file | workload | GPUs | gather kernel | kernel ms (HIP events) | algorithmic GB/s | frac of 8 TB/s | frac of the box's stream | PMC traffic ÷ algorithmic | ms per step | T bit-tests/s | result_check |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for base, d in bench_rows(rnd):
        r = d["roofline"]
        alg = r.get("algorithmic_bytes_per_launch")
        ratio = (r["traffic"] / alg) if (r.get("traffic") and alg) else None
        if ratio is None:            # the pass may have been taken after the line was written: same kernel name in the record
            for key, e in pmc.items():
                if isinstance(e, dict) and e.get("kernel") == r.get("kernel") and e.get("algorithmic_bytes_per_launch") == alg and e.get("round") == rnd:
                    ratio = e["ratio"]
        rc = d.get("result_check")
        out.append("| `%s` | %s | %s | `%s` | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            base, d.get("config", {}).get("workload", "?").split(":")[0], d.get("n_gpus", d.get("aggregate", {}).get("n_gpus", 1)), r.get("kernel"), fmt(r.get("kernel_ms"), 4), fmt(r.get("achieved"), 0),
            fmt(r.get("frac"), 4), fmt(r.get("frac_of_measured_stream"), 3), fmt(ratio, 4), fmt(d.get("ms_per_step"), 4),
            fmt(d["value"] / 1e3, 2) if d.get("value") else "—", ("ok" if rc.get("ok") else "FAILED") if isinstance(rc, dict) else "—"))
    return out


def early_exit_table(rnd):
    """The `early_exit` blocks of the lines (the same batch on the same resident matrix with KWAGE_SEARCH_EARLY_EXIT) and the
    lines of whole runs with --early-exit; fetched bytes from the "X@ee" PMC passes."""
    pmc = json.load(open(os.path.join(PROF, "pmc_traffic.json")))
    out = ["| file | workload | nominal kernel ms | early-exit stage | its kernel ms | ms per step | speed-up of the kernel | fetched bytes (PMC) | fetched ÷ algorithmic | fetched ÷ time ÷ 8 TB/s | lists identical to the nominal kernel's | result_check |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for base, d in bench_rows(rnd):
        wl = d.get("config", {}).get("workload", "?").split(":")[0]
        ee = d.get("early_exit")
        if isinstance(ee, dict) and "kernel" in ee:
            alg = (ee.get("nominal_rate") or {}).get("algorithmic_bytes")
            fetched, frac = ee.get("fetched_bytes"), ee.get("frac_of_fetched")
            if fetched is None:
                for key, e in pmc.items():
                    if key.endswith("@ee") and isinstance(e, dict) and e.get("kernel") == ee["kernel"] and e.get("algorithmic_bytes_per_launch") == alg and e.get("round") == rnd:
                        fetched = e["hbm_read_bytes_per_launch"]
                        frac = fetched / (ee["kernel_ms"] * 1e-3) / 1e9 / 8000.0
            rc = ee.get("result_check") or {}
            out.append("| `%s` › early_exit | %s | %s | `%s` | %s | %s | %s | %s | %s | %s | %s | %s |" % (
                base, wl, fmt(d["roofline"].get("kernel_ms"), 4), ee["kernel"], fmt(ee.get("kernel_ms"), 4), fmt(ee.get("ms_per_step"), 4), fmt(ee.get("speedup_vs_nominal_kernel"), 2),
                ("%d" % fetched) if fetched else "—", fmt(fetched / alg, 4) if (fetched and alg) else "—", fmt(frac, 4), ee.get("identical_to_nominal"), "ok" if rc.get("ok") else ("—" if not rc else "FAILED")))
        elif d.get("config", {}).get("early_exit"):
            r = d["roofline"]
            alg = r.get("algorithmic_bytes_per_launch")
            fetched = r.get("traffic")
            if fetched is None:
                for key, e in pmc.items():
                    if key.endswith("@ee") and isinstance(e, dict) and e.get("kernel") == r.get("kernel") and e.get("algorithmic_bytes_per_launch") == alg and e.get("round") == rnd:
                        fetched = e["hbm_read_bytes_per_launch"]
            frac = fetched / (r["kernel_ms"] * 1e-3) / 1e9 / 8000.0 if (fetched and r.get("kernel_ms")) else None
            rc = d.get("result_check") or {}
            out.append("| `%s` (whole run with --early-exit) | %s | — | `%s` | %s | %s | — | %s | %s | %s | — | %s |" % (
                base, wl, r.get("kernel"), fmt(r.get("kernel_ms"), 4), fmt(d.get("ms_per_step"), 4), ("%d" % fetched) if fetched else "—",
                fmt(fetched / alg, 4) if (fetched and alg) else "—", fmt(frac, 4), "ok" if rc.get("ok") else ("—" if not rc else "FAILED")))
    return out


def boxes(rnd):
    """uuid -> {pci, stream GB/s seen, files}: which box every line of the round was taken on (config.box)."""
    seen = {}
    for base, d in bench_rows(rnd):
        box = (d.get("config") or {}).get("box")
        if not isinstance(box, dict) or "›" in base:
            continue
        e = seen.setdefault(str(box.get("uuid")), {"pci": box.get("pci"), "name": box.get("name"), "stream_read_gbps": [], "files": []})
        e["stream_read_gbps"].append(box.get("measured_stream_read_gbps"))
        e["files"].append(base)
    for p in sorted(glob.glob(os.path.join(PROF, rnd + "_*.json"))):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        box = d.get("box") if isinstance(d, dict) else None
        if isinstance(box, dict) and "uuid" in box:
            e = seen.setdefault(str(box.get("uuid")), {"pci": box.get("pci"), "name": box.get("name"), "stream_read_gbps": [], "files": []})
            e["stream_read_gbps"].append(box.get("measured_stream_read_gbps"))
            e["files"].append(os.path.basename(p))
    return seen


def boxes_table(rnd):
    out = ["| box (device uuid) | PCI | streaming read seen (GB/s) | files taken on it |", "|---|---|---|---|"]
    for uuid, e in sorted(boxes(rnd).items()):
        vals = [v for v in e["stream_read_gbps"] if v]
        out.append("| `%s` | %s | %s | %s |" % (uuid, e["pci"], ("%.0f – %.0f" % (min(vals), max(vals))) if vals else "—", ", ".join("`%s`" % f for f in e["files"])))
    return out


def cpu_table(rnd):
    out = ["| file | CPU baseline | G bit-tests/s | threads | database | reference wall | whole hit list == the timed kernel's | bounded no-early-exit sample (G bit-tests/s, threads) |", "|---|---|---|---|---|---|---|---|"]
    for base, d in bench_rows(rnd):
        cb = d.get("cpu_baseline")
        if not isinstance(cb, dict) or cb.get("value") is None:
            continue
        idb = cb.get("identical_db") or {}
        smp = cb.get("no_early_exit_sample") or {}
        out.append("| `%s` | %s%s | %s | %s | %s | %s | %s | %s |" % (
            base, cb.get("kind"), "" if cb.get("extrapolated") else ", identical database", fmt(cb.get("value"), 1), cb.get("cores"),
            ("%d files, %.1f GB, 2^%s rows (%s)" % (idb["files"], idb["db_bytes"] / 1e9, idb["log_2_rows"], idb.get("directory"))) if idb.get("files") else "bounded sample (extrapolated)",
            ("%.2f s" % idb["reference_wall_s"]) if idb.get("reference_wall_s") else "—", idb.get("whole_list_identical", "—"),
            ("%s, %s" % (fmt(smp.get("value"), 1), smp.get("cores"))) if smp else "—"))
    return out


def kernel_key(name):
    """'and_kernel<2,8,nt>' / 'and_kernel<2,8,true,false>' -> ('and_kernel', '2', '8'): the kernel and its first two template arguments."""
    base, _, rest = name.partition("<")
    args = [x.strip() for x in rest.rstrip(">").split(",")]
    return (base,) + tuple(args[:2])


def rocprof_table(rnd):
    out = ["| rocprofv3 summary | gather kernel | dispatches | average ms | launches per step | HIP-event ms per step of the same run (`kernel_ms`) | algorithmic bytes ÷ rocprofv3 time ÷ 8 TB/s |", "|---|---|---|---|---|---|---|"]
    for path in sorted(glob.glob(os.path.join(PROF, rnd + "_*kernel_stats.csv"))):
        base = os.path.basename(path)
        ks = kernel_stats(path)
        if not ks:
            continue
        stem = base.replace("_kernel_stats.csv", "")
        blocks = []
        p = os.path.join(PROF, stem + "_bench_under_rocprof.json")
        if os.path.exists(p):
            line = load_line(p)
            if line:
                blocks = [line] + [b for b in (line.get("also") or {}).values() if isinstance(b, dict) and "roofline" in b]
        for k, calls, avg in ks[:3]:
            hip = frac = per = None
            for blk in blocks:
                r = blk["roofline"]
                if kernel_key(r.get("kernel", "")) == kernel_key(k):
                    per = len(blk["config"].get("groups") or [1])
                    hip = r.get("kernel_ms")
                    if r.get("algorithmic_bytes_per_launch"):
                        frac = r["algorithmic_bytes_per_launch"] / (avg * per * 1e-3) / 1e9 / 8000.0
            out.append("| `%s` | `%s` | %d | %s | %s | %s | %s |" % (base, k, calls, fmt(avg, 4), per if per else "—", fmt(hip, 4), fmt(frac, 4)))
    return out


def proxy_table(path):
    d = json.load(open(path))
    out = ["| workload | split | samples per GPU | row bytes | gather kernel | ms per step | kernel ms | frac of 8 TB/s | rocprofv3 average ms | speed-up vs 1 | efficiency | kernel-only efficiency |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for wl, rows in d.items():
        if not isinstance(rows, list):
            continue
        for r in rows:
            if "error" in r:
                continue
            out.append("| %s | %d | %d | %s | `%s` | %s | %s | %s | %s | %s | %s | %s |" % (
                wl, r["split"], r["samples_per_gpu"], r.get("row_bytes"), r["kernel"], fmt(r["ms_per_step"], 4), fmt(r["kernel_ms"], 4), fmt(r["frac"], 4),
                fmt(r.get("rocprof_kernel_avg_ms"), 4), fmt(r.get("speedup_vs_1"), 3), fmt(r.get("efficiency"), 3), fmt(r.get("kernel_only_efficiency"), 3)))
    return out


def occupancy_table(rnd):
    out = ["| file | workload | gather kernel | waves | mean waves resident per CU | share of wave time waiting (s_waitcnt) | issue-stalled | issuing | frac under PMC |", "|---|---|---|---|---|---|---|---|---|"]
    for path in sorted(glob.glob(os.path.join(PROF, rnd + "_*_pmc_occupancy.json"))):
        d = json.load(open(path))
        for k, e in d.get("kernels", {}).items():
            out.append("| `%s` | %s | `%s` | %s | %s | %s | %s | %s | %s |" % (os.path.basename(path), d["workload"], k, fmt(e.get("waves"), 0), fmt(e.get("mean_waves_resident_per_cu"), 2),
                       fmt(e.get("share_of_wave_time_waiting"), 3), fmt(e.get("share_of_wave_time_issue_stalled"), 3), fmt(e.get("share_of_wave_time_issuing"), 3), fmt(d.get("frac_under_pmc"), 4)))
    return out


def pmc_table():
    rec = json.load(open(os.path.join(PROF, "pmc_traffic.json")))
    out = ["| entry | kernel | HBM bytes per launch (FETCH_SIZE × calibrated factor) | algorithmic bytes | ratio | pass | code hash | round |", "|---|---|---|---|---|---|---|---|"]
    for k, e in rec.items():
        if isinstance(e, dict) and "kernel" in e:
            out.append("| %s | `%s` | %d | %d | %s | `%s` | `%s` | %s |" % (k, e["kernel"], e["hbm_read_bytes_per_launch"], e["algorithmic_bytes_per_launch"], fmt(e.get("ratio"), 4),
                       e.get("source"), e.get("code_hash"), e.get("round")))
    return out


def rounds():
    return sorted({m.group(1) for p in glob.glob(os.path.join(PROF, "r[0-9][0-9]_*")) for m in [re.match(r"(r\d\d)_", os.path.basename(p))] if m}, reverse=True)


def render_tables():
    out = ["# profiles/TABLES.md — generated by `tools/render_tables.py` from the files in this directory; do not edit", ""]
    for rnd in rounds():
        out += ["## Round %s — `bench.py` lines (`%s_*.json`)" % (rnd[1:].lstrip("0"), rnd), ""] + bench_table(rnd) + [""]
        et = early_exit_table(rnd)
        if len(et) > 2:
            out += ["### Early exit: the path `kwage` and `kwage_node` run by default (`early_exit` blocks and `--early-exit` runs)", ""] + et + [""]
        ct = cpu_table(rnd)
        if len(ct) > 2:
            out += ["### CPU baseline (the reference `kwage`, OpenMP)", ""] + ct + [""]
        bt = boxes_table(rnd)
        if len(bt) > 2:
            out += ["### Boxes (`config.box`: kwage_device_fingerprint)", ""] + bt + [""]
        rt = rocprof_table(rnd)
        if len(rt) > 2:
            out += ["### rocprofv3 kernel summaries (`%s_*_kernel_stats.csv`)" % rnd, ""] + rt + [""]
        for p in sorted(glob.glob(os.path.join(PROF, rnd + "_strong_scaling_proxy.json"))):
            out += ["### Strong scaling, one-GPU proxy (`%s`)" % os.path.basename(p), ""] + proxy_table(p) + [""]
        ot = occupancy_table(rnd)
        if len(ot) > 2:
            out += ["### Occupancy / waiting of the gather kernels (`%s_*_pmc_occupancy.json`)" % rnd, ""] + ot + [""]
    out += ["## HBM traffic per launch (`pmc_traffic.json`)", ""] + pmc_table() + [""]
    return "\n".join(out) + "\n"


DESIGN_BEGIN = "<!-- GENERATED:measurements BEGIN (tools/render_tables.py; do not edit) -->"
DESIGN_END = "<!-- GENERATED:measurements END -->"


def design_block():
    """The newest round's tables as DESIGN.md carries them between its markers."""
    rnd = rounds()[0]
    block = ["", "`bench.py` lines of round %s (`profiles/%s_*.json`):" % (rnd[1:].lstrip("0"), rnd), ""] + bench_table(rnd)
    et = early_exit_table(rnd)
    if len(et) > 2:
        block += ["", "Early exit -- the path `kwage` and `kwage_node` run by default (`early_exit` blocks of the lines above, and whole runs with `--early-exit`):", ""] + et
    ct = cpu_table(rnd)
    if len(ct) > 2:
        block += ["", "CPU baseline (the reference `kwage`, OpenMP over files):", ""] + ct
    block += ["", "rocprofv3 summaries of the same commands:", ""] + rocprof_table(rnd)
    for p in sorted(glob.glob(os.path.join(PROF, rnd + "_strong_scaling_proxy.json"))):
        block += ["", "Strong scaling, one-GPU proxy (`profiles/%s`):" % os.path.basename(p), ""] + proxy_table(p)
    block += [""]
    return "\n".join(block)


def bootstrap_index():
    """First run: take the hand-written tables of profiles/README.md as the index's descriptions."""
    idx = {}
    path = os.path.join(PROF, "README.md")
    rnd = None
    for ln in open(path).read().splitlines():
        m = re.match(r"## Round (\d)", ln)
        if m:
            rnd = "r0" + m.group(1)
        if ln.startswith("| `") and rnd:
            cells = [c.strip() for c in ln.strip().strip("|").split(" | ")]
            if len(cells) >= 2:
                idx.setdefault(rnd, []).append({"files": cells[0], "what": cells[1], "command": cells[2] if len(cells) > 2 else ""})
    return idx


def render_readme(idx):
    out = ["# profiles/ — measurement evidence (MI355X, gfx950, ROCm 7.2 image)", "",
           "Generated by `tools/render_tables.py` from `INDEX.json` (what each file is, the command that made it) and the directory listing;",
           "the numbers are in `TABLES.md`, generated from the files themselves. History of rounds 1–3 in prose: `HISTORY.md`.", ""]
    listed = set()
    idx = {k: v for k, v in idx.items() if re.match(r"r\d\d$", k)}        # (the "boxes" key is generated, not a round)
    for rnd in sorted(idx, reverse=True):
        out += ["## Round %s (`%s_*`)" % (rnd[1:].lstrip("0"), rnd), "", "| file | what | command that produced it |", "|---|---|---|"]
        for e in idx[rnd]:
            out.append("| %s | %s | %s |" % (e["files"], e["what"], e.get("command", "")))
            listed.update(re.findall(r"`([^`]+)`", e["files"]))
        out.append("")
    every = sorted(os.path.basename(p) for p in glob.glob(os.path.join(PROF, "*")) if os.path.isfile(p))
    missing = [f for f in every if f not in listed and not any(f.startswith(x.rstrip("*…")) for x in listed if x.endswith(("*", "…")))
               and f not in ("README.md", "TABLES.md", "INDEX.json", "HISTORY.md", "pmc_traffic.json")]
    loose = [f for f in missing if not any(f in e["files"] or re.sub(r"^r\d\d_", "", f).split(".")[0].split("_")[0] in e["files"] for r in idx.values() for e in r)]
    if loose:
        out += ["## Not described in INDEX.json", ""] + ["* `%s`" % f for f in loose] + [""]
    return "\n".join(out) + "\n"


def main():
    open(os.path.join(PROF, "TABLES.md"), "w").write(render_tables())
    ipath = os.path.join(PROF, "INDEX.json")
    if not os.path.exists(ipath):
        json.dump(bootstrap_index(), open(ipath, "w"), indent=1, ensure_ascii=False)
    idx = json.load(open(ipath))
    # which box every line was taken on (config.box), per round: kept in the index beside the files' descriptions
    idx["boxes"] = {rnd: boxes(rnd) for rnd in rounds() if boxes(rnd)}
    json.dump(idx, open(ipath, "w"), indent=1, ensure_ascii=False)
    open(os.path.join(PROF, "README.md"), "w").write(render_readme(idx))
    # the newest round's tables inside DESIGN.md
    dpath = os.path.join(ROOT, "DESIGN.md")
    txt = open(dpath).read()
    if DESIGN_BEGIN in txt and DESIGN_END in txt:
        txt = txt[:txt.index(DESIGN_BEGIN) + len(DESIGN_BEGIN)] + design_block() + txt[txt.index(DESIGN_END):]
        open(dpath, "w").write(txt)
    b = DESIGN_BEGIN
    print("rendered profiles/TABLES.md, profiles/README.md%s" % (", DESIGN.md block" if b in txt else ""))


if __name__ == "__main__":
    main()
