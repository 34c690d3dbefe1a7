#!/usr/bin/env python3
"""The reference `kwage` and this repo's `kwage` on IDENTICAL files at C2's size ("identical DB bits
and identical FASTA").

The resident C2 matrix of bench.py (100 000 samples x 2^23-bit filters, seeded synthetic columns + planted genomes) is
read back from HBM band by band (kwage_group_read_rows) and written as reference-format `.db` files of <= 2048 columns
(49 files, 105 GB), the batch's 1 000 x 1 kb queries beside them as FASTA.  Then both binaries search that directory:
`oracle/_ref/kwage` (the reference's own sources, OpenMP over files, all host threads, page cache warm, best of 2) and
`kwage_amd/bin/kwage` (one MI355X; load and search split from KWAGE_VERBOSE), at -t 1.0 and -t 0.8, and the two reports are
compared as sets per query -- the WHOLE hit list, not sampled queries.

    python tools/e2e_c2_identical.py [--workload c2] [--log2-rows L] [--dir D] [--out profiles/r05_c2_identical_db]

Free space is probed first (--dir, $TMPDIR, /tmp, /dev/shm, the repo): when 2^23 rows do not fit anywhere the largest
log2 row count that does is taken instead and the report says so (the same samples, queries and densities; fewer rows)."""
import argparse
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--log2-rows", type=int, default=0, help="filter length of the copy (default: the workload's, or the largest that fits)")
    ap.add_argument("--dir", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_c2_identical_db"))
    ap.add_argument("--thresholds", default="1.0,0.8")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()

    import kwage_oracle as oracle
    import kwage_amd as ka
    from kwage_amd import native, synth
    from dataclasses import replace
    import bench

    w = synth.WORKLOADS[args.workload]
    ncol_file = 2048
    n_files = (w.num_samples + ncol_file - 1) // ncol_file
    row_bytes = (w.num_samples + 7) // 8
    lines = []

    def say(msg):
        print(msg, flush=True)
        lines.append(msg)

    # ---- where do 2^L x row_bytes fit? ------------------------------------------------------------------------------------
    cands = [d for d in (args.dir, os.environ.get("TMPDIR", ""), "/tmp", "/dev/shm", os.path.join(ROOT, "gpurun_out")) if d and os.path.isdir(d)]
    free = {d: shutil.disk_usage(d).free for d in cands}
    mem = {}
    for ln in open("/proc/meminfo"):
        k, v = ln.split(":")
        if k in ("MemTotal", "MemAvailable"):
            mem[k] = int(v.split()[0]) * 1024
    say("host %s: %d CPUs, MemTotal %.0f GB, MemAvailable %.0f GB; free space: %s"
        % (socket.gethostname(), os.cpu_count(), mem.get("MemTotal", 0) / 1e9, mem.get("MemAvailable", 0) / 1e9,
           ", ".join("%s %.0f GB" % (d, f / 1e9) for d, f in free.items())))
    want_L = args.log2_rows or w.log_2_filter_len
    best_dir, L = None, 0
    for d in cands:
        # tmpfs lives in memory: the files AND the page-cache-free host copy must fit beside this process
        room = free[d] - (8 << 30)
        if d.startswith("/dev/shm"):
            room = min(room, mem.get("MemAvailable", 0) - (24 << 30))
        fit = want_L
        while fit > 10 and (1 << fit) * row_bytes + (64 << 20) > room:
            fit -= 1
        if fit > L:
            best_dir, L = d, fit
    if best_dir is None or L < 12:
        say("no directory with room for even 2^12 rows: nothing done")
        return 2
    if L != w.log_2_filter_len:
        say("NOTE: 2^%d rows x %d bytes = %.1f GB do not fit anywhere (largest free: %.0f GB): the copy has 2^%d rows (%.1f GB) -- the disk limit, not a choice"
            % (w.log_2_filter_len, row_bytes, (1 << w.log_2_filter_len) * row_bytes / 1e9, max(free.values()) / 1e9, L, (1 << L) * row_bytes / 1e9))
        w = replace(w, log_2_filter_len=L, name=w.name + " [copy with 2^%d rows]" % L)
    work = os.path.join(best_dir, "kwage_c2_identical_%d" % os.getpid())
    dbdir = os.path.join(work, "db")
    os.makedirs(dbdir)
    try:
        # ---- the matrix bench.py searches, read back and written as the reference's files ----------------------------------
        t0 = time.perf_counter()
        ctx = ka.Context(0)
        s = synth.build(ctx, w, seed=1, column_seed=0)
        say("built %s in %.1f s (device %s)" % (w.name, time.perf_counter() - t0, ctx.fingerprint().get("uuid")))
        nominal = {}
        for t in [float(x) for x in args.thresholds.split(",")]:
            r = s.group.search(s.batch, t, 0)
            nominal[t] = (len(r.hits), int(r.total_kmers))
        total_kmers = nominal[list(nominal)[0]][1]
        exp = bench.export_group_as_db_files(s.group, w, dbdir, ncol_file, progress=True)
        total_bytes, nrows, nh = exp["bytes"], 1 << L, w.num_hash
        say("read back %d rows x %d bytes in bands of %d rows (D2H %.1f s) and wrote %d reference-format .db files, %.1f GB, in %.1f s -> %s"
            % (nrows, s.group.row_bytes, exp["band_rows"], exp["d2h_seconds"], n_files, total_bytes / 1e9, exp["seconds"], dbdir))
        qfile = os.path.join(work, "q.fa")
        with open(qfile, "w") as fh:
            for i, q in enumerate(s.queries):
                fh.write(">query_%d\n%s\n" % (i, q))
        qhash = hashlib.sha256(open(qfile, "rb").read()).hexdigest()[:16]
        fp = ctx.fingerprint()
        stream = s.group.stream_read_gbps(min(s.group.device_bytes, 8 << 30), 3)
        s.batch.close()
        s.group.close()
        ctx.close()

        # ---- both binaries on that directory ------------------------------------------------------------------------------
        cores = os.cpu_count() or 1
        threads = min(cores, n_files)
        bit_tests = total_kmers * nh * w.num_samples
        result = {"workload": args.workload, "name": w.name, "samples": w.num_samples, "log_2_rows": L, "full_size": L == synth.WORKLOADS[args.workload].log_2_filter_len,
                  "files": n_files, "db_bytes": total_bytes, "queries": len(s.queries), "query_len": w.query_len, "queries_sha256_16": qhash,
                  "total_kmers": total_kmers, "bit_tests": bit_tests, "host": socket.gethostname(), "cpus": cores, "box": dict(fp, measured_stream_read_gbps=round(stream, 1)),
                  "code_hash": bench.kernel_code_hash(), "directory": best_dir, "runs": {}}

        def run(exe, thr, env, reps):
            best, out, err = None, None, ""
            for _ in range(reps):
                o = os.path.join(work, "out_%s.csv" % os.path.basename(exe))
                ta = time.perf_counter()
                r = subprocess.run([exe, "-d", dbdir, "-i", qfile, "-t", repr(thr), "--o.csv", "-o", o], capture_output=True, env=env)
                dt = time.perf_counter() - ta
                if r.returncode != 0:
                    raise RuntimeError("%s failed: %s" % (exe, r.stderr.decode()[-2000:]))
                if best is None or dt < best:
                    best, out, err = dt, open(o).read(), r.stderr.decode()
            return best, out, err
        ok_all = True
        for thr in [float(x) for x in args.thresholds.split(",")]:
            have_ref = os.access(oracle.REF_KWAGE, os.X_OK)
            t_ref = out_ref = None
            if have_ref:
                t_ref, out_ref, _ = run(oracle.REF_KWAGE, thr, dict(os.environ, OMP_NUM_THREADS=str(threads)), 2)
            t_gpu, out_gpu, err_gpu = run(native.KWAGE_BIN, thr, dict(os.environ, KWAGE_VERBOSE="1"), 2)
            split = [ln.strip() for ln in err_gpu.splitlines() if ("loaded" in ln and "GB/s" in ln) or "from the start of main" in ln or ": init " in ln or " screened 2^" in ln]
            g = oracle.parse_csv(out_gpu)
            n_hits = sum(len(v) for v in g.values())
            rec = {"threshold": thr, "kwage_amd_wall_s": round(t_gpu, 3), "kwage_amd_verbose": split, "hits": n_hits,
                   "kwage_amd_g_bit_tests_per_s": round(bit_tests / t_gpu / 1e9, 1), "device_search_hits_without_early_exit": nominal[thr][0]}
            if have_ref:
                e = oracle.parse_csv(out_ref)
                same = (list(g) == list(e)) and all(sorted(g[x]) == sorted(e[x]) for x in e)
                rec.update({"reference_wall_s": round(t_ref, 3), "reference_threads": threads, "reference_g_bit_tests_per_s": round(bit_tests / t_ref / 1e9, 2),
                            "reports_identical": bool(same), "reference_hits": sum(len(v) for v in e.values()), "speedup_end_to_end": round(t_ref / t_gpu, 1)})
                ok_all = ok_all and same
                say("t = %g: reference kwage (%d OpenMP threads of %d CPUs, page cache warm, best of 2): wall %.2f s = %.2f G bit-tests/s | kwage_amd/bin/kwage (1 GPU, best of 2): wall %.2f s "
                    "(file read + H2D + search + report) = %.1f G bit-tests/s | reports identical (whole list, as sets per query): %s | hits %d (device search without early exit: %d)"
                    % (thr, threads, cores, t_ref, bit_tests / t_ref / 1e9, t_gpu, bit_tests / t_gpu / 1e9, same, n_hits, nominal[thr][0]))
            else:
                say("t = %g: oracle/_ref/kwage is not here; kwage_amd/bin/kwage wall %.2f s, hits %d" % (thr, t_gpu, n_hits))
                ok_all = False
            for ln in split:
                say("      " + ln)
            result["runs"]["%g" % thr] = rec
        # the reference with the HOST THREADS THE GPU PROGRAM USES for its file reads (16: a GPU's share of the host's CPUs on this
        # pool): the 49-thread figure above is the reference at its best on a whole 256-CPU host
        if os.access(oracle.REF_KWAGE, os.X_OK):
            result["reference_16_threads"] = {}
            for thr in [float(x) for x in args.thresholds.split(",")]:
                t16, out16, _ = run(oracle.REF_KWAGE, thr, dict(os.environ, OMP_NUM_THREADS="16"), 2)
                say("t = %g: reference kwage with OMP_NUM_THREADS=16 (the host threads kwage_amd/bin/kwage reads the files with), best of 2: wall %.2f s = %.2f G bit-tests/s"
                    % (thr, t16, bit_tests / t16 / 1e9))
                result["reference_16_threads"]["%g" % thr] = {"wall_s": round(t16, 3), "g_bit_tests_per_s": round(bit_tests / t16 / 1e9, 2)}
        # t = 1 without the screen on every query's first k-mers (KWAGE_SPARSE_SCREEN=0): every addressed slice of every file fetched
        t_ns, out_ns, err_ns = run(native.KWAGE_BIN, 1.0, dict(os.environ, KWAGE_VERBOSE="1", KWAGE_SPARSE_SCREEN="0"), 2)
        same_ns = oracle.parse_csv(out_ns) == oracle.parse_csv(run(native.KWAGE_BIN, 1.0, dict(os.environ), 1)[1])
        say("t = 1 with KWAGE_SPARSE_SCREEN=0 (every addressed slice of every file fetched, no screen on the queries' first k-mers), best of 2: wall %.2f s; same report: %s" % (t_ns, same_ns))
        for ln in [ln.strip() for ln in err_ns.splitlines() if ("loaded" in ln and "GB/s" in ln)]:
            say("      " + ln)
        result["no_screen"] = {"threshold": 1.0, "wall_s": round(t_ns, 3), "same_report": bool(same_ns), "verbose": [ln.strip() for ln in err_ns.splitlines() if "loaded" in ln and "GB/s" in ln]}
        ok_all = ok_all and same_ns
        # the same command with the WHOLE database loaded (KWAGE_SPARSE=0): what a host that keeps the database resident pays once
        thr0 = float(args.thresholds.split(",")[0])
        t_full, out_full, err_full = run(native.KWAGE_BIN, thr0, dict(os.environ, KWAGE_VERBOSE="1", KWAGE_SPARSE="0"), 1)
        same_full = oracle.parse_csv(out_full) == oracle.parse_csv(run(native.KWAGE_BIN, thr0, dict(os.environ), 1)[1])
        say("t = %g with KWAGE_SPARSE=0 (every row of every file loaded, not only the rows the batch addresses): wall %.2f s; same report: %s" % (thr0, t_full, same_full))
        for ln in [ln.strip() for ln in err_full.splitlines() if ("loaded" in ln and "GB/s" in ln) or "from the start of main" in ln]:
            say("      " + ln)
        result["whole_database_load"] = {"threshold": thr0, "wall_s": round(t_full, 3), "same_report": bool(same_full),
                                         "verbose": [ln.strip() for ln in err_full.splitlines() if "loaded" in ln and "GB/s" in ln]}
        ok_all = ok_all and same_full
        result["reports_identical"] = bool(ok_all)
        say("reports identical: %s" % ok_all)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        open(args.out + ".txt", "w").write("\n".join(lines) + "\n")
        json.dump(result, open(args.out + ".json", "w"), indent=1)
        return 0 if ok_all else 1
    finally:
        if not args.keep:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
