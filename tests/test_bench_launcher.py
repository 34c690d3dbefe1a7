"""bench.py's contract with the driver: `python bench.py --gpus N` must work without an external launcher (the
parent starts the ranks as a child process and never touches the GPU itself), and with one
(torch.distributed.run sets WORLD_SIZE).  The launcher half is checked here on the CPU; the ranks themselves
run in the GPU test at the bottom (two ranks sharing the box's one GPU, gloo instead of RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_launch_command_is_one_rank_per_gpu_on_localhost():
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5"]
    free = bench.launch_command(2, [])          # picks a free port by itself
    assert 1024 < int(free[free.index("--master-port") + 1]) < 65536


def test_parent_relays_the_result_line_last_and_the_exit_code(monkeypatch, capfd, tmp_path):
    import bench
    stub = tmp_path / "ranks.py"
    stub.write_text("import sys\nprint('{\"metric\": \"m\", \"value\": 1}')\nprint('NCCL version banner')\nsys.exit(int(sys.argv[1]))\n")
    for rc in (0, 3):
        monkeypatch.setattr(bench, "launch_command", lambda n, argv, port=None: [sys.executable, str(stub), str(rc)])
        assert bench.launch_ranks(2, ["--gpus", "2"]) == rc
        out = capfd.readouterr().out.strip().splitlines()
        assert out[0] == "NCCL version banner" and json.loads(out[-1]) == {"metric": "m", "value": 1}
    # ranks that end without a result are a failure even when they exit 0
    stub.write_text("print('nothing')\n")
    monkeypatch.setattr(bench, "launch_command", lambda n, argv, port=None: [sys.executable, str(stub)])
    assert bench.launch_ranks(2, []) != 0


def test_multi_gpu_request_without_a_launcher_starts_ranks_instead_of_exiting(monkeypatch):
    """`python bench.py --gpus 2` with WORLD_SIZE unset goes through launch_ranks (round 1 exited with a usage
    message here); with WORLD_SIZE set the process is a rank."""
    import bench
    seen = {}
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv: seen.setdefault("launch", (n, list(argv))) and 0)
    monkeypatch.setattr(bench, "rank_main", lambda args: seen.setdefault("rank", args.gpus))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    with pytest.raises(SystemExit):
        bench.main()
    assert seen["launch"] == (2, ["--gpus", "2", "--steps", "3"]) and "rank" not in seen
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "2")
    bench.main()
    assert seen == {"rank": 2}
    seen.clear()
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench.main()
    assert seen == {"rank": 1}


def test_traffic_is_reported_only_for_the_kernel_and_code_that_were_profiled(monkeypatch):
    """roofline.traffic comes from a separate PMC pass; it is bound to the kernel's name + template shape AND to the hash
    of the kernel sources + engine the pass ran on: a changed kernel body under the same name reports null."""
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert len(bench.kernel_code_hash()) == 16 and bench.kernel_code_hash() == bench.kernel_code_hash()
    n = 0
    for name, entry in rec.items():
        if name.startswith("_"):
            continue
        n += 1
        assert isinstance(entry.get("code_hash"), str) and len(entry["code_hash"]) == 16, name
        monkeypatch.setattr(bench, "kernel_code_hash", lambda e=entry: e["code_hash"])
        t, src = bench.measured_traffic(name, entry["kernel"], False)
        assert t == entry["hbm_read_bytes_per_launch"] and src["status"] == "kernel and code hash match" and src["kernel"] == entry["kernel"]
        t, src = bench.measured_traffic(name, "some_other_kernel<1,2>", False)
        assert t is None and src["status"].startswith("stale: the PMC pass profiled")
        assert bench.measured_traffic(name, entry["kernel"], True)[0] is None
        monkeypatch.setattr(bench, "kernel_code_hash", lambda: "0123456789abcdef")
        t, src = bench.measured_traffic(name, entry["kernel"], False)
        assert t is None and src["status"].startswith("stale: the kernel sources / engine changed")
    assert n >= 4
    assert bench.measured_traffic("no-such-workload", "k", False)[0] is None
    # a workload whose kernel depends on where its matrix lies has one pass per kernel ("c2" and "c2@<the band form>"):
    # the run's kernel picks the pass
    keyed = [k for k in rec if "@" in k and not k.endswith("@ee")]
    assert keyed
    # searches with early exit have passes of their own ("c2@ee": `bench.py --workload c2 --early-exit`), reported by the
    # line's `early_exit` block only
    for key in [k for k in rec if k.endswith("@ee")]:
        monkeypatch.setattr(bench, "kernel_code_hash", lambda e=rec[key]: e["code_hash"])
        assert bench.measured_traffic(key[:-3], rec[key]["kernel"], True)[0] == rec[key]["hbm_read_bytes_per_launch"]
        assert bench.measured_traffic(key[:-3], rec[key]["kernel"], False)[0] != rec[key]["hbm_read_bytes_per_launch"]
    for key in keyed:
        workload, kern = key.split("@", 1)
        monkeypatch.setattr(bench, "kernel_code_hash", lambda e=rec[key]: e["code_hash"])
        assert rec[key]["kernel"] == kern and rec[workload]["kernel"] != kern
        assert bench.measured_traffic(workload, kern, False)[0] == rec[key]["hbm_read_bytes_per_launch"]
        assert bench.measured_traffic(workload, rec[workload]["kernel"], False)[0] == rec[workload]["hbm_read_bytes_per_launch"]


@pytest.mark.gpu
def test_bench_self_launches_two_ranks_on_one_gpu():
    """The whole N>1 path of bench.py as the driver would start it -- no launcher -- on a one-GPU box: two ranks
    share device 0 and exchange over gloo (RCCL refuses two ranks on one device).  One JSON line with the
    aggregate block, every rank's kernel time, and twice the single-rank work."""
    env = dict(os.environ, KWAGE_BENCH_BACKEND="gloo", KWAGE_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "tiny"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak"
    agg = line["aggregate"]
    assert agg["n_gpus"] == 2 and len(agg["kernel_ms_per_rank"]) == 2 and all(x > 0 for x in agg["kernel_ms_per_rank"])
    assert agg["kernel_ms_max"] >= agg["kernel_ms_mean"] > 0 and 0 < agg["aggregate_frac"] < 1
    assert line["rccl"]["world"] == 2 and line["rccl"]["backend"] == "gloo"
    assert line["sustained"]["steps"] >= 4 and len(line["sustained"]["kernel_ms_mean_per_rank"]) == 2
    assert line["config"]["step_pipeline"] == "on"
    # the run verified its own exchange: every rank's hit count + checksum against the merged list on rank 0
    ec = line["exchange_check"]
    assert ec["ok"] and ec["ranks"] == 2 and ec["hits"] == sum(ec["hits_per_rank"]) and ec["checksum"] == ec["sum_of_rank_checksums"]
    assert ec["sorted_unique"] and ec["columns_in_range"] and line["config"]["hits_per_step"] == ec["hits"]
    assert line["rccl"]["searches_per_step"] == 1
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--workload", "tiny", "--no-cpu-baseline", "--no-sustained"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    single = json.loads(one.stdout.strip().splitlines()[-1])
    assert single["n_gpus"] == 1 and "sustained" not in single
    assert line["roofline"]["algorithmic_bytes_per_launch"] == single["roofline"]["algorithmic_bytes_per_launch"]    # weak scaling: same share per GPU
    # the C5 code path (several filter-size groups per rank: ONE list and ONE exchange per step for all of them, the
    # searches pipelined through the context's two slots) on toy groups, two ranks
    r5 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "c5tiny", "--no-sustained"],
                        capture_output=True, text=True, env=env, timeout=900)
    assert r5.returncode == 0, r5.stderr[-3000:]
    l5 = json.loads(r5.stdout.strip().splitlines()[-1])
    assert l5["n_gpus"] == 2 and len(l5["config"]["groups"]) == 4 and l5["config"]["threshold"] == 0.8
    assert len(l5["aggregate"]["kernel_ms_per_rank"]) == 2 and all(x > 0 for x in l5["aggregate"]["kernel_ms_per_rank"])
    assert l5["roofline"]["kernel"].startswith("count_")
    assert l5["config"]["step_pipeline"] == "on" and l5["rccl"]["searches_per_step"] == 4
    # one collective per step: 2 steps that warm the pipeline's two buffers + 1 warm-up step + 3 timed ones (small lists
    # ride in the first collective whole)
    assert l5["rccl"]["collectives"] == 6 and l5["rccl"]["p2p_batches"] == 0
    e5 = l5["exchange_check"]
    assert e5["ok"] and e5["hits"] == sum(e5["hits_per_rank"]) == l5["config"]["hits_per_step"] and e5["hits"] > 0
    # every line carries the post-timing check of its hit lists (planted positives + sampled queries against the oracle), rank by rank
    for ln in (line, single, l5):
        rc = ln["result_check"]
        assert rc["ok"] and all(rc["ranks_ok"]) and len(rc["ranks_ok"]) == ln["n_gpus"] and rc["sampled_queries"] >= 3 and rc["planted_columns_found"] == rc["planted_columns_expected"] > 0
    # `also` blocks at N > 1 (the driver's 8-GPU run measures the C4 and C5 shares behind the C2 headline): the same code on
    # toy workloads -- every block with its own exchange_check and result_check -- and strong scaling (the columns of the
    # workload split over the ranks at 1024-column boundaries: the ranks' algorithmic bytes differ and add up)
    ra = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "tiny", "--no-sustained",
                         "--also", "c5tiny"], capture_output=True, text=True, env=env, timeout=900)
    assert ra.returncode == 0, ra.stderr[-3000:]
    la = json.loads(ra.stdout.strip().splitlines()[-1])
    blk = la["also"]["c5tiny"]
    assert la["scaling"] == "weak" and la["config"]["total_samples"] == 10000 and la["exchange_check"]["ok"] and la["result_check"]["ok"]
    assert blk["exchange_check"]["ok"] and blk["result_check"]["ok"] and blk["roofline"]["kernel"].startswith("count_") and blk["rccl"]["searches_per_step"] == 4
    rs = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "tiny", "--no-sustained",
                         "--scaling", "strong"], capture_output=True, text=True, env=env, timeout=900)
    assert rs.returncode == 0, rs.stderr[-3000:]
    ls = json.loads(rs.stdout.strip().splitlines()[-1])
    assert ls["scaling"] == "strong" and ls["config"]["total_samples"] == 5000 and ls["config"]["samples_per_gpu"] == 3072 and "also" not in ls
    per = ls["aggregate"]["algorithmic_bytes_per_rank"]
    assert len(per) == 2 and per[0] > per[1] > 0 and ls["exchange_check"]["ok"] and ls["result_check"]["ok"]
    assert ls["exchange_check"]["total_columns"] >= 5000


def test_strong_split_is_the_sharded_hosts_partition():
    """--scaling strong / --share-of K: a rank holds block `part` of partition_columns(total, K) -- the shares add up to
    the workload's columns, every boundary but the last is a 1024-column multiple, and a multi-group workload (C5) is
    cut group by group.  weak: every rank holds the workload as it stands."""
    import bench
    from kwage_amd import synth
    for name in ("c2", "c3"):
        total = synth.WORKLOADS[name].num_samples
        for k in (2, 4, 8):
            shares = [bench.rank_share(name, "strong", k, r) for r in range(k)]
            assert sum(w.num_samples for w, _, _ in shares) == total and all(t == total for _, _, t in shares)
            assert all(w.num_samples % 1024 == 0 for w, _, _ in shares[:-1])
            assert max(w.num_samples for w, _, _ in shares) - min(w.num_samples for w, _, _ in shares) <= 2 * 1024
            assert all(g is None for _, g, _ in shares)
        w, g, t = bench.rank_share(name, "weak", 8, 3)
        assert w.num_samples == total and g is None and t == total
        assert bench.rank_share(name, "strong", 1, 0)[0].num_samples == total
    per_rank = [bench.rank_share("c5", "strong", 4, r)[1] for r in range(4)]
    for gi, (lg, ns) in enumerate(synth.C5_GROUPS):
        assert [g[gi][0] for g in per_rank] == [lg] * 4 and sum(g[gi][1] for g in per_rank) == ns
    with pytest.raises(ValueError):
        bench.rank_share("c5", "strong", 8, 7)            # its smallest group (6000 samples) has six 1024-column units
    assert bench.rank_share("c5", "weak", 8, 0)[1] == synth.C5_GROUPS
    with pytest.raises(ValueError):
        bench.rank_share("tiny", "strong", 8, 7)          # 5000 columns are five 1024-column units: rank 7 of 8 gets none


def test_also_blocks_follow_the_default_headline_only():
    import bench
    a = bench.parse_args([])
    # N = 1: C3.  N > 1: the C4 and C5 per-GPU shares (weak) and C3 split over the ranks (the fixed-total-work curve)
    assert bench.also_workloads(a, 1) == ["c3"] and bench.also_workloads(a, 8) == ["c4", "c5", "c3_strong"] and bench.also_workloads(a, 2) == ["c4", "c5", "c3_strong"]
    assert bench.also_workloads(bench.parse_args(["--also", "c3_strong"]), 2) == ["c3_strong"]
    for argv in (["--workload", "c3"], ["--scaling", "strong"], ["--share-of", "4"], ["--early-exit"], ["--also", "none"]):
        assert bench.also_workloads(bench.parse_args(argv), 1) == [], argv
    assert bench.also_workloads(bench.parse_args(["--also", "c5s,c2t,c2"]), 1) == ["c5s", "c2t"]      # never the headline twice


def test_result_check_accepts_the_oracles_lists_and_refuses_anything_else(oracle):
    """bench.py's post-timing check on a matrix small enough for the CPU: a stand-in group whose rows come from a numpy
    image, hit lists computed by the oracle itself -> ok; one record dropped, altered or added -> not ok."""
    import numpy as np
    import bench
    from types import SimpleNamespace
    from kwage_amd import synth
    from kwage_amd.engine import HIT_DTYPE, SearchResult
    k, nh, L, ncol = 31, 2, 12, 700
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = lambda n: acgt[rng.integers(0, 4, size=n)].tobytes().decode()
    genomes = [seq(600), seq(600)]
    planted = [[3, 400], [77, 699]]
    image = (rng.integers(0, 256, size=(1 << L, (ncol + 7) // 8), dtype=np.uint8) & rng.integers(0, 256, size=(1 << L, (ncol + 7) // 8), dtype=np.uint8))
    image[:, -1] &= np.uint8((1 << (ncol % 8)) - 1)
    for g, cols in zip(genomes, planted):
        rows = oracle.row_indices(oracle.unique_kmers(g, k), k, nh, L).reshape(-1)
        for c in cols:
            image[rows, c // 8] |= np.uint8(1 << (c % 8))
    queries = [genomes[0][:300], seq(300), genomes[1][100:400], seq(300), genomes[0][300:600]]
    qsrc = [0, -1, 1, -1, 0]
    for thr in (1.0, 0.8):
        w = synth.Workload("stub", ncol, L, k, nh, len(queries), 300, thr)
        recs, nk, qt = [], [], []
        for qi, q in enumerate(queries):
            kmers = oracle.unique_kmers(q, k)
            hits, _ = oracle.search_image(image, image.shape[1], k, nh, L, ncol, kmers, float(np.float32(thr)))
            recs += [(qi, c, n) for c, n in hits]
            nk.append(len(kmers))
            qt.append(0 if thr == 1.0 else oracle.query_threshold(float(np.float32(thr)), len(kmers)))
        group = SimpleNamespace(read_rows=lambda rows: np.ascontiguousarray(image[np.asarray(rows, dtype=np.int64)]))
        member = SimpleNamespace(workload=w, group=group, queries=queries, query_genome=qsrc, planted=planted)
        mk = lambda rr: SearchResult(np.array(rr, dtype=HIT_DTYPE), np.array(nk, np.uint32), np.array(qt, np.uint32), sum(nk), 0, 0, 0.0, 0.0, 1, "stub")
        good = bench.result_check([member], [mk(recs)], thr)
        assert good["ok"] and good["planted_queries"] == 3 and good["planted_columns_found"] == good["planted_columns_expected"] == 6
        assert good["sampled_queries"] == 3 and good["sampled_hits_compared"] > 0 and not good["mismatches"]
        first_planted = next(i for i, r in enumerate(recs) if r[0] == 0 and r[1] == 3)
        dropped = recs[:first_planted] + recs[first_planted + 1:]
        altered = [(q, c, n - 1) if i == first_planted else (q, c, n) for i, (q, c, n) in enumerate(recs)]
        added = sorted(recs + [(1, 5, nk[1])]) if not any(r[0] == 1 and r[1] == 5 for r in recs) else None
        for bad in (dropped, altered, added):
            if bad is not None:
                assert not bench.result_check([member], [mk(bad)], thr)["ok"]
