#!/usr/bin/env python3
"""bench.py -- throughput of the kwage search path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5|c5s|c2t|c1]
                    [--scaling weak|strong] [--also auto|none|c3,c4,...] [--share-of K [--share-rank R]]

A "step" is one pass of the hot path over one batch of synthetic queries against the
HBM-resident synthetic database (k-mer pack + MurmurHash3 + row gather + AND/count + hit
compaction + D2H of the sorted hit list).  Inputs (database and query strings) are resident in
HBM before the timed region.  Metric = BASELINE.json's: G k-mer.sample bit-tests/s, with the
achieved HBM GB/s of the gather kernel against the 8 TB/s roofline beside it.

The line the driver reads (ONE JSON line, the last of stdout) is the headline workload -- C2, the configuration
BASELINE.json quotes the metric on, unchanged from round to round -- and carries, beside `roofline`, `sustained`,
`aggregate` and `cpu_baseline`:

  * `result_check`: AFTER the timed region the hit lists the timed kernel produced are checked -- every query cut from
    a planted genome reports the genome's columns, and three sampled queries are bit-exact against the CPU oracle on the
    rows they address, read back from HBM.  The oracle is used here as the checker only, never inside a timed region;
    a mismatch ends the run with a non-zero status on every rank.
  * `early_exit`: the SAME batch against the SAME resident matrix searched with the reference's early exit
    (kwage.cpp:437-483) -- what `kwage` and `kwage_node` run by default -- after the timed region: `ms_per_step`,
    `kernel`, `kernel_ms`, the nominal rate (the batch's algorithmic bytes over that time, labelled as bytes mostly NOT
    read), `fetched_bytes` from the PMC pass of `bench.py --workload X --early-exit` (profiles/pmc_traffic.json "X@ee")
    with `frac_of_fetched` = fetched bytes / kernel time / 8 TB/s, `identical_to_nominal` (the hit lists equal the timed
    kernel's record for record) and its own `result_check`.  Never part of `value`.
  * `also`: the other BASELINE.json configurations this launch can hold, each measured like the headline (own
    `ms_per_step`, `roofline`, `early_exit`, `result_check`, at N > 1 `exchange_check`) after the headline's matrix was
    freed: at N = 1 **C3** (the largest 1-GPU configuration), at N > 1 the per-GPU shares of **C4** and **C5** and
    **`c3_strong`** -- C3's 1 M columns SPLIT over the N ranks: the fixed-total-work curve (its N = 1 point is the
    N = 1 line's `also.c3`), next to the weak-scaling headline.

N > 1: one process per GPU.  Either the caller starts the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE come from the environment), or
plain `python bench.py --gpus N` starts them itself: the parent process -- which never touches the GPU --
runs torch.distributed.run as a child and relays its output and exit code.  The sample (column) axis is
sharded, queries are replicated, every rank searches its block independently and the per-rank hit lists are
concatenated on rank 0 by one small RCCL exchange per step.  No row data ever crosses xGMI.
  --scaling weak   (default) every rank holds `num_samples` columns of its own: per-GPU work is fixed.
  --scaling strong the workload's columns are SPLIT over the ranks at 1024-column boundaries (partition_columns):
                   total work is fixed, rows get narrower as N grows -- SURVEY 8(e)'s real limiter.
  --share-of K     on ONE GPU, hold what rank R (--share-rank, default 0) of a K-way strong split would hold: the
                   one-GPU proxy of the strong-scaling curve (tools/strong_scaling_proxy.py).
Every rank's gather-kernel time goes into the line (`aggregate`): the north-star figure is the fraction of the
AGGREGATE HBM bandwidth, i.e. all ranks' algorithmic bytes over the slowest rank's kernel time.

The CPU baseline (rank 0, N=1 only) times the REFERENCE binary (oracle/_ref/kwage, OpenMP over
<=2048-column .db files) when it travelled with the snapshot, else the repo's C restatement, on
a bounded column subset of the same workload.  It is a reported baseline, not the target.

Rank 0 prints the result as ONE JSON line, the last line of stdout (with NCCL_DEBUG=VERSION in the
environment RCCL prints its version banner to stdout before it).
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SUSTAINED_SECONDS = 2.0  # back-to-back steps after the timed region (DVFS: a 40 ms window could be a burst)
MULTI_GROUPS = {"c5": "C5_GROUPS", "c5tiny": "C5_TEST_GROUPS"}     # workloads of several filter-size groups (kwage_amd/synth.py)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("KWAGE_BENCH_WORKLOAD", "c2"))
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("KWAGE_BENCH_SCALING", "weak"),
                    help="weak: every rank holds the workload's columns; strong: the workload's columns are split over the ranks")
    ap.add_argument("--share-of", type=int, default=0, help="one GPU holding one rank's share of a K-way strong split (the strong-scaling proxy)")
    ap.add_argument("--share-rank", type=int, default=0, help="which rank's share --share-of holds")
    ap.add_argument("--also", default=os.environ.get("KWAGE_BENCH_ALSO", "auto"),
                    help="other configurations measured after the headline: auto (N=1: c3; N>1: c4,c5 -- only when the headline is c2), none, or a comma list")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=("auto", "sample"), default="auto",
                    help="auto: the reference on the IDENTICAL database (the resident matrix written as .db files) where the host has room, plus the bounded sample; sample: the bounded sample only")
    ap.add_argument("--no-result-check", action="store_true", help="skip the post-timing check of the hit lists against the oracle")
    ap.add_argument("--no-sustained", action="store_true", help="skip the %.0f s sustained block after the timed steps" % SUSTAINED_SECONDS)
    ap.add_argument("--early-exit", action="store_true", help="enable the reference's early exit for the WHOLE run (not the nominal figure; the PMC passes behind the `early_exit` blocks use it)")
    ap.add_argument("--no-early-exit-block", action="store_true", help="skip the `early_exit` block (the same batch searched with the reference's early exit, after the timed region)")
    ap.add_argument("--cpu-files", type=int, default=0, help="number of 2048-column .db files of the CPU sample (default: host cores, max 16)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without an external launcher
# ------------------------------------------------------------------------------------------------------
def launch_command(n_ranks, argv, port=None):
    """The child command that starts one rank per GPU.  Pure function of its arguments (tested on CPU)."""
    if port is None:
        with socket.socket() as sk:        # a free port of this host
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n_ranks, argv):
    """Parent of a self-launched multi-GPU run: start the ranks as a CHILD process group, relay their stdout
    (rank 0's JSON line comes last) and return the child's exit code.  Nothing here initialises HIP: a
    process that has touched the GPU must never be replaced or re-used as a launcher on this pool."""
    cmd = launch_command(n_ranks, argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] starting %d ranks: %s" % (n_ranks, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    last_json = None
    for line in proc.stdout:
        if line.startswith("{") and '"metric"' in line:
            last_json = line            # hold the result line back so that it is the LAST line of stdout
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    rc = proc.wait()
    if last_json is not None:
        sys.stdout.write(last_json)
        sys.stdout.flush()
    elif rc == 0:
        print("[bench] the ranks exited without a result line", file=sys.stderr)
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------
def find_room(nbytes, prefer=""):
    """A directory with room for `nbytes` of files (+ slack): `prefer`, $TMPDIR, /tmp, /dev/shm, gpurun_out/.  tmpfs
    lives in memory: there the files must fit beside this process.  -> (dir or None, {dir: free bytes})"""
    cands = [d for d in (prefer, os.environ.get("TMPDIR", ""), "/tmp", "/dev/shm", os.path.join(ROOT, "gpurun_out")) if d and os.path.isdir(d)]
    free = {d: shutil.disk_usage(d).free for d in cands}
    avail = 0
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    for d in cands:
        room = free[d] - (8 << 30)
        if d.startswith("/dev/shm"):
            room = min(room, avail - (24 << 30))
        if nbytes + (64 << 20) <= room:
            return d, free
    return None, free


def export_group_as_db_files(group, w, dbdir, ncol_file=2048, progress=False):
    """The RESIDENT matrix of a synthetic workload, read back from HBM band by band (kwage_group_read_rows) and written as
    reference-format `.db` files of <= ncol_file columns (what `maestro` would produce, options.h:137; layout
    build_db.cpp:189-427), sample j named SRR%07d.  -> {files, bytes, d2h_seconds, seconds}"""
    import concurrent.futures
    import struct
    import zlib
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kwage_oracle as oracle
    L, k, nh = w.log_2_filter_len, w.kmer_len, w.num_hash
    nrows = 1 << L
    n_files = (w.num_samples + ncol_file - 1) // ncol_file
    files = []
    for f in range(n_files):
        ncol = min(ncol_file, w.num_samples - f * ncol_file)
        path = os.path.join(dbdir, "part%03d.db" % f)
        files.append({"path": path, "fd": os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644), "ncol": ncol,
                      "b0": f * ncol_file // 8, "nb": (ncol + 7) // 8, "crc": 0})
    band = max(1, min(nrows, (2 << 30) // group.row_bytes))
    t0 = time.perf_counter()
    t_read = 0.0
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 16)) as pool:
        for r0 in range(0, nrows, band):
            n = min(band, nrows - r0)
            ta = time.perf_counter()
            block = group.read_rows(np.arange(r0, r0 + n, dtype=np.uint32))
            t_read += time.perf_counter() - ta

            def put(fi):
                part = np.ascontiguousarray(block[:, fi["b0"]:fi["b0"] + fi["nb"]])
                fi["crc"] = zlib.crc32(part, fi["crc"])
                os.pwrite(fi["fd"], part, oracle.HEADER_SIZE + r0 * fi["nb"])
            list(pool.map(put, files))
            if progress and (r0 // band) % 8 == 0:
                print("  [export] rows %d / %d written (%.0f s)" % (r0 + n, nrows, time.perf_counter() - t0), file=sys.stderr, flush=True)
    for f, fi in enumerate(files):
        ncol = fi["ncol"]
        hdr = oracle.DBHeader(kmer_len=k, num_hash=nh, log_2_filter_len=L, num_filter=ncol, hash_func=0, compression=0)
        hdr.crc32 = fi["crc"] & 0xFFFFFFFF            # build_db.cpp:307
        hdr.info_start = oracle.HEADER_SIZE + nrows * fi["nb"]
        recs = [oracle.pack_filter_info(oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % (f * ncol_file + j)))) for j in range(ncol)]
        loc, locs = hdr.info_start + 8 * ncol, []
        for rr in recs:
            locs.append(loc)
            loc += len(rr)
        os.pwrite(fi["fd"], hdr.pack(), 0)
        os.pwrite(fi["fd"], struct.pack("<%dQ" % ncol, *locs) + b"".join(recs), hdr.info_start)
        os.close(fi["fd"])
    return {"files": n_files, "bytes": int(sum(os.path.getsize(fi["path"]) for fi in files)), "d2h_seconds": round(t_read, 2),
            "seconds": round(time.perf_counter() - t0, 2), "band_rows": band}


def cpu_baseline_identical(s, w, nominal, max_seconds=90.0):
    """BASELINE.md section 4 as planned: the reference `kwage` (oracle/_ref/kwage: the reference's own sources, OpenMP over
    files, host threads = min(CPUs, files), page cache warm, best of 2) on the IDENTICAL database -- the resident matrix
    read back from HBM and written as <= 2048-column `.db` files -- and the identical queries, and its report compared
    with the timed kernel's hit list: the WHOLE list, per query, as sets of (run accession, k-mers found).
    -> the cpu_baseline block, or None when the reference binary or the room for the files is missing (the caller then
    times the bounded sample).  The reference has no switch for its early exit (kwage.cpp:437-483): the figure is its
    default behaviour on these bits, the counterpart of the line's `early_exit` block; the bounded sample beside it
    (`no_early_exit_sample`) keeps every addressed row read, the counterpart of `value`."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kwage_oracle as oracle
    if not os.access(oracle.REF_KWAGE, os.X_OK) or not w.log_2_filter_len:
        return None
    nbytes = (1 << w.log_2_filter_len) * ((w.num_samples + 7) // 8)
    if nbytes / 4e9 > max_seconds:             # (measured: 5 GB/s read back + written on the pool's boxes)
        return None
    where, free = find_room(nbytes)
    if where is None:
        return None
    work = tempfile.mkdtemp(prefix="kwage_cpu_identical_", dir=where)
    try:
        dbdir = os.path.join(work, "db")
        os.makedirs(dbdir)
        exp = export_group_as_db_files(s.group, w, dbdir)
        qfile = os.path.join(work, "q.fa")
        with open(qfile, "w") as fh:
            for i, q in enumerate(s.queries):
                fh.write(">query_%d\n%s\n" % (i, q))
        cores = os.cpu_count() or 1
        threads = min(cores, exp["files"])
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        best, text = None, None
        for _ in range(2):
            o = os.path.join(work, "ref.csv")
            t0 = time.perf_counter()
            r = subprocess.run([oracle.REF_KWAGE, "-d", dbdir, "-i", qfile, "-t", repr(float(w.threshold)), "--o.csv", "-o", o], env=env, capture_output=True)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError("reference kwage failed: " + r.stderr.decode()[-1000:])
            if best is None or dt < best:
                best, text = dt, open(o).read()
        ref = oracle.parse_csv(text)
        # the timed kernel's list in the report's terms: query name -> {(run accession, k-mers found)}
        hits = nominal.hits
        order = np.argsort(hits["query"], kind="stable")
        hq = hits["query"][order]
        starts = np.searchsorted(hq, np.arange(len(s.queries) + 1))
        same, n_ref = True, 0
        for qi in range(len(s.queries)):
            sl = order[starts[qi]:starts[qi + 1]]
            mine = sorted(("SRR%07d" % int(c), int(n)) for c, n in zip(hits["column"][sl], hits["num_match"][sl]))
            theirs = sorted((acc, found) for acc, _, found, _ in ref.get("query_%d" % qi, []))
            n_ref += len(theirs)
            if mine != theirs:
                same = False
        bit_tests = int(nominal.bit_tests)
        return {"value": round(bit_tests / best / 1e9, 3), "unit": "G bit-tests/s", "cores": threads, "kind": "reference", "extrapolated": False,
                "identical_db": {"files": exp["files"], "db_bytes": exp["bytes"], "log_2_rows": int(w.log_2_filter_len), "columns": int(w.num_samples), "directory": where,
                                 "export_seconds": exp["seconds"], "d2h_seconds": exp["d2h_seconds"], "reference_wall_s": round(best, 3), "threads": threads, "cpus": cores,
                                 "threshold": float(w.threshold), "reference_hits": n_ref, "device_hits": int(len(hits)),
                                 "whole_list_identical": bool(same)},
                "sample": "reference kwage (OpenMP over files, %d threads of %d CPUs) on the IDENTICAL database: the resident matrix read back from HBM and written as %d reference-format "
                          ".db files (%.1f GB, %s) + the identical %d queries, page cache warm, best of 2, wall %.2f s; its early exit (kwage.cpp:437-483) cannot be switched off: "
                          "this is its default behaviour on these bits -- the counterpart of the line's `early_exit` block; whole hit list identical to the timed kernel's: %s"
                          % (threads, cores, exp["files"], exp["bytes"] / 1e9, where, len(s.queries), best, same)}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_baseline(w, queries, n_files):
    """Time the reference CPU `kwage` (or the oracle port) on a bounded sample of workload w:
    n_files .db files x 2048 columns, 2^min(L,20) rows, SAME k / hashes / threshold / queries.
    Every file holds the planted genomes in one column so that the reference's early exit
    (kwage.cpp:466-470) never fires: it reads every addressed row, like the nominal GPU figure."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kwage_oracle as oracle
    cores = os.cpu_count() or 1
    if n_files <= 0:
        n_files = 2*min(cores, 16)         # two files per thread: ~15-30 core-seconds of reference work per run
    L = min(w.log_2_filter_len, 20) if w.log_2_filter_len else 20      # (multi-group workloads carry no single filter length)
    ncol = 2048
    k, nh = w.kmer_len, w.num_hash
    rng = np.random.default_rng(99)
    # sample of the query set sized so the CPU leg stays ~10-30 s
    qs = [q for q in queries if len(q) >= k][:max(1, min(len(queries), 1000))]
    tmp = tempfile.mkdtemp(prefix="kwage_cpu_", dir="/tmp")
    # The CPU leg runs on a column / row SUBSET with a matrix of its own (same density, same queries, same k,
    # hashes and threshold): the GPU matrix (>= 100 GB) does not fit the host.  Cost is linear in columns
    # per addressed row, so the rate carries over -- but it is an extrapolation and says so.
    subset = {"columns": ncol * n_files, "of_columns": int(w.num_samples) if w.num_samples else None,
              "log_2_rows": L, "of_log_2_rows": int(w.log_2_filter_len) if w.log_2_filter_len else None,
              "queries": len(qs), "of_queries": len(queries), "matrix": "own random bits of the same density (seed 99), not the GPU matrix"}
    try:
        # rows addressed by the queries' k-mers (so the planted column matches every query)
        uniq = [oracle.unique_kmers(q, k) for q in qs]
        total_kmers = int(sum(len(u) for u in uniq))
        addressed = np.unique(np.concatenate([oracle.row_indices(u, k, nh, L).reshape(-1) for u in uniq])) \
            if total_kmers else np.zeros(0, np.uint32)
        infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % j)) for j in range(ncol)]
        image = None
        a = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
        b = rng.integers(0, 1 << 63, size=(1 << L, ncol // 64), dtype=np.uint64)
        base = (a & b).view(np.uint8).reshape(1 << L, ncol // 8)               # density ~0.25
        del a, b
        for f in range(n_files):
            rows = np.roll(base, 7919 * f, axis=0)                               # a different matrix per file, same density
            rows[addressed, 0] |= 1                                              # column 0 matches EVERY query: no early exit anywhere
            oracle.write_db(os.path.join(tmp, "s%02d.db" % f), k, nh, L, rows, ncol, infos)
            if f == 0:
                image = rows
        qfile = os.path.join(tmp, "q.fa")
        with open(qfile, "w") as fh:
            for i, q in enumerate(qs):
                fh.write(">q%d\n%s\n" % (i, q))
        bit_tests = total_kmers * nh * ncol * n_files
        if os.access(oracle.REF_KWAGE, os.X_OK):
            env = dict(os.environ, OMP_NUM_THREADS=str(min(cores, 16, n_files)))   # its only parallel axis is files
            best = None
            for _ in range(2):     # first run warms the page cache
                t0 = time.perf_counter()
                r = subprocess.run([oracle.REF_KWAGE, "-d", tmp, "-i", qfile, "-t", repr(float(w.threshold)), "--o.csv",
                                    "-o", os.path.join(tmp, "out.csv")], env=env, capture_output=True)
                dt = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError("reference kwage failed: " + r.stderr.decode())
                best = dt if best is None else min(best, dt)
            return {"value": bit_tests / best / 1e9, "unit": "G bit-tests/s", "cores": min(cores, 16, n_files), "kind": "reference",
                    "extrapolated": True, "subset": subset,
                    "sample": "reference kwage (OpenMP over files), %d files x %d columns x 2^%d rows, %d queries x %d bp, "
                              "page cache warm, best of 2, wall %.2f s; a planted column matches every query, so the early exit never fires and every addressed row is read (as in the nominal GPU figure)"
                              % (n_files, ncol, L, len(qs), w.query_len, best)}
        # fallback: the repo's own C restatement, one thread, one file image in memory
        t0 = time.perf_counter()
        thr = float(np.float32(w.threshold))
        for u in uniq:
            oracle.search_image(image, image.shape[1], k, nh, L, ncol, u, thr, early_exit=True)
        dt = time.perf_counter() - t0
        subset["columns"] = ncol
        return {"value": total_kmers * nh * ncol / dt / 1e9, "unit": "G bit-tests/s", "cores": 1, "kind": "port",
                "extrapolated": True, "subset": subset,
                "sample": "oracle C restatement, 1 thread, 1 file x %d columns x 2^%d rows in memory, %d queries, wall %.2f s"
                          % (ncol, L, len(qs), dt)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernel_code_hash():
    """Identity of the device code and the launch logic a PMC pass was taken on: SHA-256 (first 16 hex digits) over the
    kernel sources and the engine.  profiles/pmc_traffic.json carries it per entry (tools/pmc_refresh.py writes it)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("kernels.hpp", "kmer_device.hpp", "engine_state.hpp", "engine.hip"):
        with open(os.path.join(ROOT, "kwage_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(workload, kernel, early_exit):
    """HBM bytes per launch from the PMC counters: they come from a SEPARATE rocprofv3 --pmc pass of this same
    command (counters cannot be collected from inside the run), kept in profiles/pmc_traffic.json with the kernel
    (name + template shape) and the HASH OF THE CODE they were taken on.  The number is reported only when this run
    used that kernel AND the kernel sources + engine are byte for byte what the pass profiled; otherwise null, with
    the reason in traffic_source.  tools/pmc_refresh.py regenerates every entry in one GPU session.
    Searches with early exit have entries of their own ("c2@ee": the pass ran `bench.py --workload c2 --early-exit`);
    what they fetch depends on the data, which is seeded, not on the box."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        # (a workload whose kernel depends on where its matrix lies -- and_walk_kernel or, band after band,
        # and_band_walk_kernel -- has one pass per kernel: "c2" and "c2@and_band_walk_kernel<13,4>")
        t = rec.get("%s@ee" % workload) if early_exit else (rec.get("%s@%s" % (workload, kernel)) or rec.get(workload))
    except Exception as exc:
        return None, {"file": "profiles/pmc_traffic.json", "status": "unreadable: %r" % (exc,)}
    if not t:
        return None, {"file": "profiles/pmc_traffic.json", "status": "no PMC pass recorded for workload %r%s" % (workload, " with early exit" if early_exit else "")}
    src = {"file": "profiles/pmc_traffic.json", "pmc_pass": t.get("source"), "kernel": t.get("kernel"), "code_hash": t.get("code_hash")}
    if t.get("kernel") != kernel:
        src["status"] = "stale: the PMC pass profiled %s, this run used %s" % (t.get("kernel"), kernel)
        return None, src
    now = kernel_code_hash()
    if t.get("code_hash") != now:
        src["status"] = "stale: the kernel sources / engine changed since the PMC pass (pass %s, now %s)" % (t.get("code_hash"), now)
        return None, src
    src["status"] = "kernel and code hash match"
    return t["hbm_read_bytes_per_launch"], src


# ------------------------------------------------------------------------------------------------------
# which columns a rank holds; which other configurations a launch measures
# ------------------------------------------------------------------------------------------------------
def rank_share(name, scaling, split, part):
    """-> (Workload, groups-or-None, total samples) of workload `name` for rank `part` of `split`.
    weak: every rank holds the workload's own column count.  strong: the workload's columns -- for a multi-group
    workload every group's -- are cut into `split` contiguous blocks at 1024-column boundaries (partition_columns,
    the rule the sharded hosts use) and the rank holds block `part`."""
    from dataclasses import replace
    from kwage_amd import synth
    from kwage_amd.distributed import partition_columns
    w = synth.WORKLOADS[name]
    groups = getattr(synth, MULTI_GROUPS[name]) if name in MULTI_GROUPS else None
    total = int(sum(ns for _, ns in groups)) if groups else int(w.num_samples)
    if scaling == "strong" and split > 1:
        def share(n):
            s, e = partition_columns(int(n), split)[part]
            return e - s
        if groups:
            groups = [(lg, share(ns)) for lg, ns in groups]
            if any(ns <= 0 for _, ns in groups):
                raise ValueError("strong split %d-way leaves rank %d without columns in some group of %s" % (split, part, name))
        else:
            if share(w.num_samples) <= 0:
                raise ValueError("strong split %d-way leaves rank %d without columns of %s" % (split, part, name))
            w = replace(w, num_samples=share(w.num_samples))
    return w, groups, total


def also_workloads(args, world):
    """The BASELINE.json configurations measured after the headline.  auto: only behind the default headline (C2, weak) --
    at N = 1 C3, the largest 1-GPU configuration; at N > 1 the per-GPU shares of the two 8-GPU configurations."""
    sel = (args.also or "none").strip().lower()
    if sel in ("none", "", "0", "off"):
        return []
    if sel == "auto":
        if args.workload != "c2" or args.scaling != "weak" or args.share_of or args.early_exit:
            return []
        # N > 1: the per-GPU shares of the two 8-GPU configurations (weak: every rank a full share) AND the fixed-total-work
        # curve: C3's 1 M columns SPLIT over the N ranks (131 GB / N per GPU) -- what "scaling from 1 to 8 GPUs" means
        # when the work does not grow with N (the N = 1 point of that curve is this same block of the N = 1 line: also.c3)
        return ["c3"] if world == 1 else ["c4", "c5", "c3_strong"]
    return [x for x in (s.strip() for s in sel.split(",")) if x and x != args.workload]      # ("c3_strong": c3 split over the ranks)


# ------------------------------------------------------------------------------------------------------
# result_check: the hit lists of the timed kernel against planted positives and the CPU oracle
# ------------------------------------------------------------------------------------------------------
def result_check(members, results, threshold, n_hit=2, n_miss=1):
    """After the timed region: what the kernel that was timed reported is (a) complete on the planted positives -- every
    query cut from a planted genome reports the genome's columns with every k-mer found -- and (b) bit-exact, false
    positives included, for sampled queries against the CPU oracle reducing the ADDRESSED rows read back from HBM
    (tests/test_gpu_fullsize.py's size-independent checks).  The oracle is the checker here, nothing it does is timed.
    -> dict; ["ok"] False on any mismatch."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kwage_oracle as oracle
    oracle.build()
    t0 = time.perf_counter()
    thr = float(np.float32(threshold))
    out = {"ok": True, "planted_queries": 0, "planted_columns_found": 0, "planted_columns_expected": 0,
           "sampled_queries": 0, "sampled_hits_compared": 0, "sampled_rows_read_back": 0, "mismatches": []}
    for gi, (m, res) in enumerate(zip(members, results)):
        w = m.workload
        # records grouped by query without a Python loop over all of them (C3: 150 k hits)
        order = np.argsort(res.hits["query"], kind="stable")
        hq = res.hits["query"][order]
        starts = np.searchsorted(hq, np.arange(len(m.queries) + 1))

        def per_query(qi):
            sl = order[starts[qi]:starts[qi + 1]]
            return [(int(c), int(n)) for c, n in zip(res.hits["column"][sl], res.hits["num_match"][sl])]
        for qi, src in enumerate(m.query_genome):
            if src < 0:
                continue
            found = dict(per_query(qi))
            nk = int(res.num_query_kmer[qi])
            want = m.planted[src]
            out["planted_queries"] += 1
            out["planted_columns_expected"] += len(want)
            got = sum(1 for c in want if found.get(c) == nk)
            out["planted_columns_found"] += got
            if got != len(want) and len(out["mismatches"]) < 8:
                out["mismatches"].append({"group": gi, "query": qi, "kind": "planted column missing or short of num_query_kmer"})
            lo = nk if threshold == 1.0 else int(res.query_threshold[qi])
            if any(not (lo <= n <= nk) for n in found.values()) and len(out["mismatches"]) < 8:
                out["mismatches"].append({"group": gi, "query": qi, "kind": "num_match outside [threshold, num_query_kmer]"})
        hitq = [i for i, g in enumerate(m.query_genome) if g >= 0]
        missq = [i for i, g in enumerate(m.query_genome) if g < 0]
        pick = lambda xs, n: [xs[(len(xs) - 1) * j // max(n - 1, 1)] for j in range(min(n, len(xs)))]     # first ... last
        for qi in pick(hitq, n_hit) + pick(missq, n_miss):
            kmers = oracle.unique_kmers(m.queries[qi], w.kmer_len)
            rows = oracle.row_indices(kmers, w.kmer_len, w.num_hash, w.log_2_filter_len).reshape(-1)
            matrix = m.group.read_rows(rows)                      # only the rows this query addresses
            exp = oracle.search_row_matrix(matrix, w.num_hash, w.num_samples, len(kmers), thr)
            got = per_query(qi)
            out["sampled_queries"] += 1
            out["sampled_hits_compared"] += len(exp)
            out["sampled_rows_read_back"] += int(len(rows))
            if int(res.num_query_kmer[qi]) != len(kmers) or got != exp:
                if len(out["mismatches"]) < 8:
                    out["mismatches"].append({"group": gi, "query": qi, "kind": "hit list differs from the oracle's",
                                              "device_hits": len(got), "oracle_hits": len(exp)})
    out["ok"] = (not out["mismatches"]) and out["planted_columns_found"] == out["planted_columns_expected"] and out["sampled_queries"] > 0
    out["groups"] = len(members)
    out["oracle"] = "oracle/kwage_oracle.c (kwo_search_rows) on rows read back with kwage_group_read_rows"
    out["seconds"] = round(time.perf_counter() - t0, 2)
    return out


# ------------------------------------------------------------------------------------------------------
# early_exit: the default path of the command-line programs, measured on the resident matrix
# ------------------------------------------------------------------------------------------------------
def early_exit_block(args, name, members, s, threshold, flags, nominal, pmc_applies=True):
    """The batch that was just timed, searched again with KWAGE_SEARCH_EARLY_EXIT: warm-up, then args.steps steps through
    the two search slots exactly like the timed region.  -> {ms_per_step, kernel, kernel_ms, nominal_rate (the
    ALGORITHMIC bytes of the batch over the time -- bytes the early exit mostly does not read, labelled as such),
    fetched_bytes (PMC pass of `bench.py --workload X --early-exit`, profiles/pmc_traffic.json "X@ee") and
    frac_of_fetched = fetched bytes / kernel time / 8 TB/s, identical_to_nominal (the hit lists equal the timed
    kernel's, record for record), result_check}."""
    import numpy as np
    import kwage_amd as ka
    from collections import deque
    fl = flags | ka.SEARCH_EARLY_EXIT
    n_groups = len(members)
    for _ in range(max(args.warmup, 1)):
        for m in members:
            m.group.search(s.batch, threshold, fl)
    pend, acc, last, kernel_ms = deque(), [], [], []
    nsteps = args.steps

    def collect_one():
        acc.append(pend.popleft().collect())
        if len(acc) == n_groups:
            kernel_ms.append(sum(r.search_kernel_ms for r in acc))
            last[:] = acc
            acc.clear()
    t0 = time.perf_counter()
    for j in range(nsteps * n_groups):
        pend.append(members[j % n_groups].group.submit(s.batch, threshold, fl))
        if len(pend) == 2:
            collect_one()
    while pend:
        collect_one()
    dt = time.perf_counter() - t0
    k_ms = float(np.mean(kernel_ms))
    alg = int(sum(r.algorithmic_bytes for r in last))
    kernel = getattr(last[0], "search_kernel", "")
    # (the PMC passes were taken on the whole workload: a share of a strong split has none)
    fetched, fsrc = measured_traffic(name, kernel, True) if pmc_applies else (None, {"status": "no PMC pass for a share of a strong split"})
    same = all(np.array_equal(a.hits, b.hits) and np.array_equal(a.num_query_kmer, b.num_query_kmer) for a, b in zip(last, nominal))
    out = {"ms_per_step": round(dt / nsteps * 1e3, 4), "steps": nsteps, "kernel": kernel, "kernel_ms": round(k_ms, 4),
           "kernel_ms_min": round(float(np.min(kernel_ms)), 4), "kernel_ms_max": round(float(np.max(kernel_ms)), 4),
           "kernel_ms_scope": "gather stage (all its launches)",
           "hits_per_step": int(sum(len(r.hits) for r in last)),
           "nominal_rate": {"algorithmic_bytes": alg, "gbps": round(alg / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                            "note": "the batch's ALGORITHMIC bytes (no early exit) over the early-exit kernel time: bytes mostly NOT read, not an HBM rate"},
           "speedup_vs_nominal_kernel": round(float(np.mean([r.search_kernel_ms for r in nominal])) * n_groups / k_ms, 2) if k_ms > 0 else None,
           "fetched_bytes": fetched, "fetched_source": fsrc,
           "frac_of_fetched": round(fetched / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if (fetched and k_ms > 0) else None,
           "identical_to_nominal": bool(same)}
    try:        # what the screen launch of the last search handed over (places taken of the three lists; a full unit list means tiles walked on by themselves)
        st = [x for x in members[0].group.ctx.refine_stats() if x["units_cap"]]
        if st:
            out["handed_over"] = dict(st[-1], lists_full=bool(st[-1]["units"] >= st[-1]["units_cap"]))
    except Exception:
        pass
    if not args.no_result_check:
        rc = result_check(members, last, threshold)
        out["result_check"] = {k: rc[k] for k in ("ok", "planted_queries", "planted_columns_found", "planted_columns_expected", "sampled_queries",
                                                   "sampled_hits_compared", "mismatches", "seconds") if k in rc}
        out["ok"] = bool(same and rc.get("ok"))
    else:
        out["ok"] = bool(same)
    return out


# ------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------
class Env:
    """What every measurement of a launch shares: the rank's place in the job, its context, the process group."""
    pass


def measure(env, args, name, headline, force_scaling=None):
    """Build workload `name` on this rank, run warm-up + EXACTLY args.steps timed steps (+ the sustained block for the
    headline), check the results, free everything.  -> the JSON block on rank 0, None elsewhere.  Raises SystemExit(3/4)
    on every rank when the exchange or the results fail their check."""
    import numpy as np
    import torch
    import kwage_amd as ka
    from kwage_amd import synth
    rank, local_rank, world, backend, cdev, dist, ctx = env.rank, env.local_rank, env.world, env.backend, env.cdev, env.dist, env.ctx
    sharded, force_sharded = env.sharded, env.force_sharded
    want_sustained = headline and not args.no_sustained

    split = args.share_of if args.share_of > 1 else world
    part = (args.share_rank if args.share_of > 1 else rank)
    scaling = force_scaling or ("strong" if (args.scaling == "strong" or args.share_of > 1) else "weak")
    w, groups, total_samples = rank_share(name, scaling, split, part)
    t_build = time.perf_counter()
    multi = None
    if groups is not None:
        # adaptive filter sizes: several groups searched back to back; everything below treats the first
        # group as `s` for the shared query batch and sums work / kernel time over the groups
        multi = synth.build_multi(ctx, groups, w, seed=1, column_seed=part)
        s = multi[0]
    else:
        s = synth.build(ctx, w, seed=1, column_seed=part)
    t_build = time.perf_counter() - t_build
    flags = ka.SEARCH_TIMING | (ka.SEARCH_EARLY_EXIT if args.early_exit else 0)
    threshold = w.threshold

    members = multi or [s]             # the groups this rank holds: one, or C5's eight filter sizes
    n_groups = len(members)
    ss = hx = None
    bases, total_columns = [0] * n_groups, None
    if sharded:
        from kwage_amd.distributed import HitExchange, ShardedSearch, StepPipeline, device_tensor_search_fn, global_column_bases, hits_checksum
        dev = "cuda:%d" % local_rank
        # global column numbers: group after group, inside a group rank after rank (one all_gather of the spans)
        bases, _, total_columns = global_column_bases(dist, rank, world, [int(m.group.column_span) for m in members], cdev)
        hx = HitExchange(dist, rank, world)
        # the synchronous form (one padded gather per group and step): the fallback if the pipelined search fails somewhere
        ss = [ShardedSearch(dist, rank, world, int(m.group.column_span), device_tensor_search_fn(m.group, flags, dev), device=cdev)
              for m in members]

    def step():
        """One synchronous pass of the hot path; returns (results-or-None, search_kernel_ms, hits delivered to rank 0)."""
        if not sharded:
            rs = [m.group.search(s.batch, threshold, flags) for m in members]
            return rs, sum(r.search_kernel_ms for r in rs), sum(len(r.hits) for r in rs)
        total = 0
        for one in ss:          # one group after the other; each ends in ONE gatherv of its hit list
            merged, _ = one.search(s.batch, threshold)
            total += len(merged) if merged is not None else 0
        return None, 0.0, total

    def sync_all():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if not sharded:
        for _ in range(args.warmup):
            step()
    pipe = None
    pipeline_note = "on"
    kernel_ms = []
    exchange_check = None

    trace = [] if os.environ.get("KWAGE_BENCH_TRACE") == "1" else None

    def local_finish():
        """Complete the oldest step of this rank's pipeline -> (its exchange buffer, n_hits)."""
        t_a = time.perf_counter()
        buf, n = pipe.finish()
        if trace is not None:
            trace.append((time.perf_counter() - t_a, t_a))
        kernel_ms.append(pipe.last_kernel_ms)
        if backend != "nccl":                    # gloo rehearsal: the same exchange on host tensors
            buf = buf[:n + 1].cpu()
        return buf, n

    if not sharded:
        # untimed: let the second search slot allocate its scratch too (the timed loop uses both).  Whole steps only
        # (every group once per step), so that a profiler's per-kernel totals divide by the number of steps.
        for m0 in range(0, n_groups, 2):
            pair = [m.group.submit(s.batch, threshold, flags) for m in members[m0:m0 + 2]]
            for p_ in pair:
                p_.collect()
        if n_groups == 1:
            p1, p2 = s.group.submit(s.batch, threshold, flags), s.group.submit(s.batch, threshold, flags)
            p1.collect()
            p2.collect()
    else:
        # untimed warm-up of both buffers of the step pipeline.  Rank-LOCAL work first (submit + collect, may fail on
        # one rank only, e.g. out of memory), then every rank learns whether all succeeded, and only then the first
        # collective exchange: a rank that failed never leaves the others blocked inside an all_gather.
        pending = []
        try:
            pipe = StepPipeline([m.group for m in members], bases, flags, "cuda:%d" % local_rank)
            pipe.begin(s.batch, threshold)
            pipe.begin(s.batch, threshold)
            pending = [local_finish(), local_finish()]
            ok = 1
        except Exception as exc:       # keep a number rather than none: fall back to the synchronous exchange
            print("[bench] pipelined search failed on rank %d (%r): falling back to the synchronous path" % (rank, exc), file=sys.stderr)
            ok = 0
        agree = torch.tensor([ok], dtype=torch.int32, device=cdev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0:
            pipe, pipeline_note = None, "off (warm-up of the pipelined search failed on some rank)"
            ctx.sync()
        else:
            merged = None
            for buf, n in pending:
                merged = hx.exchange_step(buf, n)
            # ---- untimed self-check of the exchange: what rank 0 holds after the gather is exactly what the ranks
            # found -- every rank's record count and an order-independent checksum of its (query, global column,
            # num_match) records against the merged list.
            buf, n = pending[-1]
            mine = np.array([n, hits_checksum(buf[1:1 + n].cpu().numpy())], dtype=np.uint64).view(np.int64)
            t = torch.from_numpy(mine.copy()).to(cdev)
            outs = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            per = [o.cpu().numpy().view(np.uint64) for o in outs]
            good = 1
            if rank == 0:
                want_n = int(sum(int(x[0]) for x in per))
                want_sum = int(sum(int(x[1]) for x in per) % (1 << 64))
                key = (merged[:, 0].astype(np.uint64) << np.uint64(32)) | merged[:, 1].astype(np.uint64) if len(merged) else np.zeros(0, np.uint64)
                ordered = bool(np.all(key[1:] > key[:-1])) if len(key) > 1 else True       # strictly ascending (query, column): sorted, no duplicate
                in_range = bool(len(merged) == 0 or int(merged[:, 1].max()) < total_columns)
                good = int(len(merged) == want_n and hits_checksum(merged) == want_sum and ordered and in_range)
                exchange_check = {"ok": bool(good), "ranks": world, "hits": int(len(merged)), "hits_per_rank": [int(x[0]) for x in per],
                                  "checksum": "%016x" % hits_checksum(merged), "sum_of_rank_checksums": "%016x" % want_sum,
                                  "sorted_unique": ordered, "columns_in_range": in_range, "total_columns": int(total_columns)}
            flag = torch.tensor([good], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if rank == 0:
                    print(json.dumps({"metric": "exchange_check failed", "workload": w.name, "exchange_check": exchange_check}), flush=True)
                sys.exit(3)

    last_results = []

    def timed_steps(nsteps):
        """nsteps passes, software-pipelined; -> hits of the last step (as delivered to rank 0)."""
        nhits = 0
        if sharded and pipe is not None:
            # multi-GPU: step i+1's searches are queued before step i is finished, and step i+2's right after -- BEFORE step
            # i's hits are exchanged (one small all_gather over RCCL; what does not fit goes to rank 0 alone) and merged on
            # rank 0: the collective's kernel waits for wave slots behind the running gather kernel, and the device must
            # have the next gather kernel queued while the host waits for it (the pipeline's buffers rotate by three)
            pipe.begin(s.batch, threshold)
            if nsteps > 1:
                pipe.begin(s.batch, threshold)
            merged = None
            for i in range(nsteps):
                buf, n = local_finish()
                if i + 2 < nsteps:
                    pipe.begin(s.batch, threshold)
                merged = hx.exchange_step(buf, n)
            nhits = len(merged) if merged is not None else 0
        elif not sharded:
            # The searches of all steps, group after group, stream through the two search slots of the context: search
            # j+1 is submitted (its k-mer stage runs) before search j is collected (copy-back, sort); the gather kernels
            # themselves run back to back, never side by side.  Every step is complete when the region ends.
            from collections import deque
            pend = deque()
            acc = []

            def collect_one():
                acc.append(pend.popleft().collect())
                if len(acc) == n_groups:
                    kernel_ms.append(sum(r.search_kernel_ms for r in acc))
                    last_results[:] = acc
                    acc.clear()
            for j in range(nsteps * n_groups):
                pend.append(members[j % n_groups].group.submit(s.batch, threshold, flags))
                if len(pend) == 2:
                    collect_one()
            while pend:
                collect_one()
            nhits = sum(len(r.hits) for r in last_results)
        else:
            for _ in range(nsteps):
                _, ms, nhits = step()
                kernel_ms.append(ms)
        return nhits

    if sharded:
        # warm-up through the code that is timed (the step pipeline + its exchange); the synchronous form only if that failed
        if pipe is not None:
            timed_steps(max(args.warmup, 1))
        else:
            for _ in range(args.warmup):
                step()
    sync_all()
    t0 = time.perf_counter()
    kernel_ms.clear()
    nhits = timed_steps(args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    if trace:     # host view of the pipeline: time blocked in collect, and collect-to-collect period
        per = np.diff([t for _, t in trace[-args.steps:]])
        print("[trace] collect wait mean %.3f ms; collect-to-collect mean %.3f ms (min %.3f max %.3f)"
              % (np.mean([d for d, _ in trace[-args.steps:]]) * 1e3, per.mean() * 1e3, per.min() * 1e3, per.max() * 1e3), file=sys.stderr)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    timed_kernel_ms = list(kernel_ms)

    # ---- sustained block: the same steps back to back for >= SUSTAINED_SECONDS (the count is derived from the agreed
    # max-over-ranks step time, so every rank runs the same number of steps and collectives) ------------------------
    sustained = None
    sus_kernel_ms = []
    if want_sustained:
        n_sus = int(max(args.steps, min(20000, SUSTAINED_SECONDS / max(dt / args.steps, 1e-6)))) + 1
        sync_all()
        kernel_ms.clear()
        t1 = time.perf_counter()
        timed_steps(n_sus)
        sync_all()
        dts = time.perf_counter() - t1
        if dist is not None:
            tmax = torch.tensor([dts], dtype=torch.float64, device=cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dts = float(tmax.item())
        sus_kernel_ms = list(kernel_ms)
        sustained = {"steps": n_sus, "seconds": round(dts, 3), "ms_per_step": round(dts / n_sus * 1e3, 4)}

    # work per step on THIS rank (weak scaling: identical on every rank; strong: the ranks' column counts differ)
    probe = list(last_results) or [m.group.search(s.batch, threshold, flags) for m in members]
    bit_tests_rank = int(sum(r.bit_tests for r in probe))
    alg_bytes_rank = int(sum(r.algorithmic_bytes for r in probe))
    if sharded and pipe is None:
        # the synchronous sharded path does not return kernel times: measure them with three local searches
        timed_kernel_ms = [sum(m.group.search(s.batch, threshold, flags).search_kernel_ms for m in members) for _ in range(3)]
        sus_kernel_ms = []

    # ---- result_check: the lists the timed kernel produced (rank-local, before any exchange) ------------------------
    rcheck = None
    if not args.no_result_check:
        try:
            rcheck = result_check(members, probe, threshold)
        except Exception as exc:          # a checker that cannot run is a failed check, not a skipped one
            rcheck = {"ok": False, "error": repr(exc)}
    # ---- early_exit: the SAME batch against the SAME resident matrix with the reference's early exit (kwage.cpp:437-483)
    # on -- what `kwage` and `kwage_node` run by default.  After the timed region (never part of `value`); rank-local.
    ee = None
    if not args.no_early_exit_block and not args.early_exit:
        try:
            ee = early_exit_block(args, name, members, s, threshold, flags, probe, pmc_applies=not (scaling == "strong" and split > 1))
        except Exception as exc:
            ee = {"ok": False, "error": repr(exc)}
    ee_ok = 1.0 if (ee is None or ee.get("ok", True)) else 0.0
    probe = probe[0]

    def stats(xs):
        xs = [float(x) for x in xs if x > 0]
        return [float(np.mean(xs)), float(np.min(xs)), float(np.max(xs))] if xs else [0.0, 0.0, 0.0]

    # every rank's kernel time (timed region: mean/min/max; sustained block: mean/min/max), its work per step and the
    # verdict of its result check, gathered on all ranks
    mine = stats(timed_kernel_ms) + (stats(sus_kernel_ms) if sustained else [0.0, 0.0, 0.0]) \
        + [float(bit_tests_rank), float(alg_bytes_rank), (1.0 if (rcheck is None or rcheck.get("ok")) else 0.0) * ee_ok]
    per_rank = [mine]
    if dist is not None:
        t = torch.tensor(mine, dtype=torch.float64, device=cdev)
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        per_rank = [[float(x) for x in o.tolist()] for o in outs]
    checks_ok = all(r[8] == 1.0 for r in per_rank)

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        bit_tests_all = int(sum(r[6] for r in per_rank))
        alg_bytes_all = int(sum(r[7] for r in per_rank))
        value = bit_tests_all * args.steps / dt / 1e9
        k_ms = per_rank[0][0]
        achieved = alg_bytes_rank / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        stream_gbps = s.group.stream_read_gbps(min(s.group.device_bytes, 8 << 30), 3)
        kernel = getattr(probe, "search_kernel", "") or ("and_kernel" if threshold == 1.0 else "count_kernel")
        traffic, traffic_source = (None, {"status": "no PMC pass for a %d-way share" % split}) if scaling == "strong" and split > 1 \
            else measured_traffic(name, kernel, args.early_exit)
        k_means = [r[0] for r in per_rank]
        k_max, k_mean = max(k_means), float(np.mean(k_means))
        # every GPU's algorithmic bytes over the SLOWEST rank's mean kernel time
        agg_achieved = alg_bytes_all / (k_max * 1e-3) / 1e9 if k_max > 0 else 0.0
        out = {
            "metric": "G k-mer·sample bit-tests/sec (+ achieved HBM GB/s of the gather kernel in `roofline`)",
            "value": round(value, 3),
            "unit": "G bit-tests/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": scaling if (world > 1 or split > 1) else "weak", "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": w.name, "samples_per_gpu": int(sum(ns for _, ns in groups)) if multi else w.num_samples,
                       "total_samples": (total_samples if scaling == "strong" else (total_samples * world)),
                       "log_2_filter_len": w.log_2_filter_len,
                       "kmer_len": w.kmer_len, "num_hash": w.num_hash, "queries": w.num_queries, "query_len": w.query_len,
                       "threshold": w.threshold, "early_exit": bool(args.early_exit), "density": w.density_q8 / 256.0,
                       "db_bytes_per_gpu": int(sum(m.group.device_bytes for m in multi)) if multi else int(s.group.device_bytes),
                       "row_bytes": None if multi else int(s.group.row_bytes),
                       "groups": [[lg, ns] for lg, ns in groups] if multi else None,
                       "sharding": ("columns (samples) over %d GPU(s)" % world) + ("; strong: this GPU holds share %d of %d" % (part, split) if split > 1 and scaling == "strong" else ""),
                       "step_pipeline": pipeline_note,
                       "total_kmers_per_step": int(probe.total_kmers) if not multi else None, "hits_per_step": int(nhits),
                       "db_build_s": round(t_build, 2),
                       # which box this is (kwage_device_fingerprint) and what its HBM streams: boxes of the pool differ by several per cent
                       "box": dict(ctx.fingerprint(), measured_stream_read_gbps=round(stream_gbps, 1), host=socket.gethostname()),
                       # how the loader chose each matrix's device block (rank 0): candidates compared, gather-probe GB/s on the kept / released one
                       "matrix_placement": [m.group.placement for m in multi] if multi else s.group.placement,
                       "seeds": {"queries_and_planted_genomes": 1, "columns": "rank (splitmix64 keyed by seed, row, word; kwage_amd/synth.py)"}},
            "hbm_gbps_algorithmic_whole_step": round(alg_bytes_all * args.steps / dt / 1e9, 1),
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": round(k_ms, 4), "kernel_ms_min": round(per_rank[0][1], 4), "kernel_ms_max": round(per_rank[0][2], 4),
                         # HIP events on the search's stream around the gather STAGE: one launch, except for and_band_walk_kernel,
                         # whose bucketing launch and finish launch are inside (rocprofv3's average for that kernel alone is ~2 % less)
                         "kernel_ms_scope": "gather stage" + (" = bucketing launch + and_band_walk_kernel + finish launch" if kernel.startswith("and_band_walk") else " = one launch"),
                         "algorithmic_bytes_per_launch": alg_bytes_rank,
                         "measured_stream_read_gbps": round(stream_gbps, 1),
                         "frac_of_measured_stream": round(achieved / stream_gbps, 4) if stream_gbps else None},
            # all ranks: every GPU's algorithmic bytes over the SLOWEST rank's mean kernel time, against N x 8 TB/s
            "aggregate": {"n_gpus": world, "kernel_ms_per_rank": [round(x, 4) for x in k_means],
                          "kernel_ms_max": round(k_max, 4), "kernel_ms_mean": round(k_mean, 4),
                          "algorithmic_bytes_per_rank": [int(r[7]) for r in per_rank],
                          "achieved": round(agg_achieved, 1), "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                          "aggregate_frac": round(agg_achieved / (HBM_PEAK_GBPS * world), 4)},
        }
        if sustained is not None:
            sk = per_rank[0][3:6]
            sustained.update({"kernel_ms_mean": round(sk[0], 4), "kernel_ms_min": round(sk[1], 4), "kernel_ms_max": round(sk[2], 4),
                              "achieved": round(alg_bytes_rank / (sk[0] * 1e-3) / 1e9, 1) if sk[0] > 0 else None,
                              "frac": round(alg_bytes_rank / (sk[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if sk[0] > 0 else None,
                              "value": round(bit_tests_all / (sustained["ms_per_step"] * 1e-3) / 1e9, 3),
                              "kernel_ms_mean_per_rank": [round(r[3], 4) for r in per_rank]})
            out["sustained"] = sustained
        if sharded:
            rv = None
            try:
                rv = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
            except Exception:
                pass
            out["rccl"] = {"world": world, "backend": backend, "version": rv,
                           "exchange": ("one all_gather of [count | first %d records] per step for all %d group(s); longer lists: the rest to rank 0 only, exact sizes, one grouped send/recv" % (hx.spec, n_groups))
                                       if pipe is not None else "one padded all_gather per group and step",
                           "collectives": hx.stats["collectives"], "p2p_batches": hx.stats["p2p_batches"],
                           "searches_per_step": n_groups}
            if exchange_check is not None:
                out["exchange_check"] = exchange_check
        if ee is not None:
            out["early_exit"] = ee
        if rcheck is not None:
            rcheck["ranks_ok"] = [bool(r[8] == 1.0) for r in per_rank]
            rcheck["checked"] = "rank-local hit lists of the kernel that was timed (%s), after the timed region" % kernel
            out["result_check"] = rcheck
        if force_sharded:
            out["config"]["note"] = "KWAGE_BENCH_FORCE_SHARDED: multi-GPU code path on one rank"
        if headline and world == 1 and not args.no_cpu_baseline:
            # the planned baseline first -- the reference on the IDENTICAL database, whole-list parity included --, where the
            # host has the room (105 GB of files for C2) and the reference binary travelled; the bounded sample always (it is
            # the one that reads every addressed row, like `value`)
            ident = None
            if not multi and args.cpu_baseline != "sample":
                try:
                    ident = cpu_baseline_identical(s, w, probe)
                except Exception as e:
                    ident = {"error": repr(e)}
            try:
                sample = cpu_baseline(w, s.queries, args.cpu_files)
            except Exception as e:   # the baseline is informative; never lose the GPU number over it
                sample = {"value": None, "unit": "G bit-tests/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
            if ident and "error" not in ident:
                ident["no_early_exit_sample"] = sample
                out["cpu_baseline"] = ident
                if not ident["identical_db"]["whole_list_identical"]:
                    checks_ok = False
            else:
                if ident:
                    sample["identical_db"] = ident
                out["cpu_baseline"] = sample

    for m in members:
        m.batch.close()
        m.group.close()
    if not checks_ok:
        if rank == 0:
            print(json.dumps({"metric": "result_check failed", "workload": w.name, "result_check": out.get("result_check") if out else None}), flush=True)
        else:
            print("[bench] rank %d: result_check %r" % (rank, rcheck), file=sys.stderr, flush=True)
        sys.exit(4)
    return out


ALSO_KEYS = ("value", "unit", "ms_per_step", "steps", "warmup", "scaling", "config", "hbm_gbps_algorithmic_whole_step", "roofline",
             "aggregate", "rccl", "exchange_check", "result_check", "early_exit")


def rank_main(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if args.share_of > 1 and (world > 1 or not (0 <= args.share_rank < args.share_of)):
        raise SystemExit("--share-of K is a ONE-GPU proxy (N = 1) and needs 0 <= --share-rank < K")

    import torch
    import kwage_amd as ka
    from kwage_amd import native
    if local_rank == 0:
        native.ensure_built()          # artefacts are git-ignored; normally they travel with the snapshot
    else:
        for _ in range(600):           # other ranks wait for rank 0's build instead of racing it
            if os.path.exists(native.lib_path()) and os.path.exists(native.KWAGE_BIN):
                break
            time.sleep(0.5)

    env = Env()
    env.rank, env.world = rank, world
    # KWAGE_BENCH_BACKEND=gloo + KWAGE_BENCH_ONE_DEVICE=1 rehearse the N>1 code path on a one-GPU box
    env.backend = os.environ.get("KWAGE_BENCH_BACKEND", "nccl")
    if os.environ.get("KWAGE_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
        # ranks that share a device must not each hold a second candidate block for their matrix while the others allocate theirs
        os.environ.setdefault("KWAGE_GROUP_PLACEMENT_PROBE", "0")
    env.local_rank = local_rank
    # KWAGE_BENCH_FORCE_SHARDED=1: take the multi-GPU code path (device-resident hits + RCCL exchange)
    # even with one rank, to measure its per-step overhead on a one-GPU box
    env.force_sharded = os.environ.get("KWAGE_BENCH_FORCE_SHARDED") == "1"
    if env.force_sharded and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    env.sharded = world > 1 or env.force_sharded
    env.dist = None
    env.ctx = ka.Context(local_rank)       # before torch.distributed creates its streams (hardware-queue assignment is first come, first served)
    env.cdev = ("cuda:%d" % local_rank) if env.backend == "nccl" else "cpu"       # where the collectives' tensors live
    if env.sharded:
        import torch.distributed as dist
        env.dist = dist
        torch.cuda.set_device(local_rank)
        if env.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(env.backend)

    out = measure(env, args, args.workload, True)

    # ---- the other BASELINE.json configurations this launch can hold, after the headline's matrix was freed ------------
    names = also_workloads(args, world)
    if names:
        if rank == 0:     # a copy of the headline on stderr first: whatever happens below, the number is on record
            print("[bench] headline (the `also` blocks follow): " + json.dumps(out), file=sys.stderr, flush=True)
        also = {}
        for name in names:
            t0 = time.perf_counter()
            wl, _, how = name.partition("_")          # "c3_strong": workload c3, its columns split over the ranks
            force = "strong" if how == "strong" else None
            if world == 1:
                try:
                    blk = measure(env, args, wl, False, force)
                except SystemExit:
                    raise
                except Exception as exc:          # e.g. the matrix does not fit beside another tenant of the device
                    blk = {"error": repr(exc)}
            else:
                blk = measure(env, args, wl, False, force)       # ranks stay in step: an exception ends the job (the headline is on stderr)
            if rank == 0:
                also[name] = {k: blk[k] for k in ALSO_KEYS if k in blk} if "error" not in blk else blk
                also[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        if rank == 0:
            out["also"] = also
    if rank == 0:
        print(json.dumps(out), flush=True)

    env.ctx.close()
    if env.dist is not None:
        env.dist.barrier()
        env.dist.destroy_process_group()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher above us: start the ranks ourselves, from a process that never touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank_main(args)


if __name__ == "__main__":
    main()
