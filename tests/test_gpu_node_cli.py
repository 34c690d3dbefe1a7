"""`kwage_node`: the kwage command line as one process per GPU, the per-GPU hit lists gathered on rank 0 over RCCL
(kwage_amd/csrc/kwage_node.cpp).  It shares kwage's option parser, query readers and report writers (it includes
kwage_main.cpp), so with any number of ranks its report must be kwage's, byte for byte -- and through kwage the
reference's.  A one-GPU box can run ONE rank over RCCL (two ranks on one device are refused): that rank goes through the
whole exchange (communicator, count all-gather, grouped send / recv with nothing to receive, merge, mapping of global
columns back to files) on every golden case and on hostile option lines.  Several ranks are rehearsed with
KWAGE_NODE_REHEARSE=1: all ranks on device 0, the records through a shared host segment where RCCL would carry them --
the file sharding, the global column numbers, the gather order and the mapping back to (file, column) are the same code."""
import json
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT

KWAGE = os.path.join(ROOT, "kwage_amd", "bin", "kwage")
NODE = os.path.join(ROOT, "kwage_amd", "bin", "kwage_node")


def _env(**extra):
    env = dict(os.environ, KWAGE_NODE_RANKS="1")
    env.update(extra)
    env.pop("NCCL_DEBUG", None)          # (RCCL's version banner would land on stderr -- never on stdout: NCCL_DEBUG_FILE -- and differ from kwage's)
    return env


def _cases():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))["cases"]


def _args(case):
    args = []
    for d in case["db"]:
        args += ["-d", d]
    for q in case["queries"]:
        args += ["-i", q]
    return args + ["-t", case["threshold"], "--o." + case["format"]] + case["cmdline"]


def test_node_cli_option_handling_is_kwages_without_a_gpu():
    """Everything that ends a run before the search happens in the parent process, which never touches a device."""
    cdir = os.path.join(GOLDEN, "basic")
    for argv in ([], ["-h"], ["-?"], ["-d", "db", "-t", "7", "ACGT"], ["-d", "no_such_dir", "ACGT"], ["-d", "db"], ["-d", "db", "-i", "reads.txt"],
                 ["--o.cs", "-d", "db", "-t", "0", "ACGT"]):
        a = subprocess.run([KWAGE] + argv, cwd=cdir, capture_output=True, timeout=60)
        b = subprocess.run([NODE] + argv, cwd=cdir, capture_output=True, timeout=60, env=_env())
        assert (b.returncode, b.stdout, b.stderr) == (a.returncode, a.stdout, a.stderr), argv


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases(), ids=lambda c: "%s-%s" % (c["name"], c["expected"]))
def test_node_cli_report_is_kwages(case):
    cdir = os.path.join(GOLDEN, case["name"])
    one = subprocess.run([KWAGE] + _args(case), cwd=cdir, capture_output=True, timeout=300)
    node = subprocess.run([NODE] + _args(case), cwd=cdir, capture_output=True, timeout=300, env=_env())
    assert one.returncode == 0 and node.returncode == 0, (one.stderr.decode(), node.stderr.decode())
    assert node.stdout == one.stdout
    secs = lambda b: re.sub(rb"in \d+ sec", b"in N sec", b)
    assert secs(node.stderr) == secs(one.stderr)
    if len(case["db"]) == 1 and case["name"] != "multi":
        assert node.stdout == open(os.path.join(cdir, case["expected"]), "rb").read()       # the reference's own bytes


@pytest.mark.gpu
def test_node_cli_streams_batches_and_writes_files(tmp_path):
    """Small query batches (several exchanges per run), early exit off (the persistent kernels), -o."""
    cdir = os.path.join(GOLDEN, "multi")
    args = ["-d", "dbs", "-i", "reads.fastq", "-i", "contigs.fa.gz", "-t", "0.7", "--o.json", "TTACAGCCGATGTTAGCGCGCGCTGGAATTACAAAGCTCACTGCCAAGTTAAACCATGGGGCGCGGGTAT"]
    want = subprocess.run([KWAGE] + args, cwd=cdir, capture_output=True, timeout=300)
    assert want.returncode == 0
    # with NCCL_DEBUG=VERSION the banner goes to stderr, the report stays clean
    r = subprocess.run([NODE] + args, cwd=cdir, capture_output=True, timeout=300, env=dict(os.environ, KWAGE_NODE_RANKS="1", NCCL_DEBUG="VERSION"))
    assert r.returncode == 0 and r.stdout == want.stdout
    for env in ({"KWAGE_BATCH_BASES": "300"}, {"KWAGE_EARLY_EXIT": "0", "KWAGE_COUNT_WALK_MIN_ROWS": "1", "KWAGE_NARROW": "0"}):
        out = tmp_path / "node.json"
        r = subprocess.run([NODE] + args + ["-o", str(out)], cwd=cdir, capture_output=True, timeout=300, env=_env(**env))
        assert r.returncode == 0 and r.stdout == b"", r.stderr.decode()
        assert out.read_bytes() == want.stdout, env


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3, 5])
def test_node_cli_several_ranks_rehearsed(ranks, tmp_path):
    """multi/ has files of two k-mer lengths and several filter sizes: with 2, 3 and 5 ranks the groups split unevenly
    and some ranks own no file of a group (5 ranks: some own none at all)."""
    for name, args in (("multi", ["-d", "dbs", "-i", "reads.fastq", "-i", "contigs.fa.gz", "-t", "0.7", "--o.json", "TTACAGCCGATGTTAGCGCGCGCTGGAATTACAAAGCTCACTGCCAAGTTAAACCATGGGGCGCGGGTAT"]),
                       ("multi", ["-d", "dbs", "-i", "reads.fastq", "-t", "1", "--o.csv"])):
        cdir = os.path.join(GOLDEN, name)
        want = subprocess.run([KWAGE] + args, cwd=cdir, capture_output=True, timeout=300)
        assert want.returncode == 0 and want.stdout
        for env in ({}, {"KWAGE_BATCH_BASES": "300", "KWAGE_EARLY_EXIT": "0"}):
            r = subprocess.run([NODE] + args, cwd=cdir, capture_output=True, timeout=300,
                               env=_env(KWAGE_NODE_RANKS=str(ranks), KWAGE_NODE_REHEARSE="1", **env))
            assert r.returncode == 0, r.stderr.decode()
            assert r.stdout == want.stdout, (ranks, args, env)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 2, 3])
def test_node_cli_database_larger_than_hbm_takes_passes(ranks):
    """A database that does not fit its ranks' HBM in one pass (KWAGE_MAX_GROUP_BYTES stands in for the free memory: here
    less than two files' matrices): every rank plans every rank's passes from the headers and the budgets, loads one span
    of whole files per pass, reads the query sources once per pass, and takes part in every exchange -- with an empty list
    once its own files are done.  The report must be kwage's, byte for byte (one rank: over RCCL; more: rehearsed)."""
    cdir = os.path.join(GOLDEN, "multi")
    for args in (["-d", "dbs", "-i", "reads.fastq", "-i", "contigs.fa.gz", "-t", "0.7", "--o.json", "TTACAGCCGATGTTAGCGCGCGCTGGAATTACAAAGCTCACTGCCAAGTTAAACCATGGGGCGCGGGTAT"],
                 ["-d", "dbs", "-i", "reads.fastq", "-t", "1", "--o.csv"]):
        want = subprocess.run([KWAGE] + args, cwd=cdir, capture_output=True, timeout=300)
        assert want.returncode == 0 and want.stdout
        for env in ({"KWAGE_MAX_GROUP_BYTES": "200000"}, {"KWAGE_MAX_GROUP_BYTES": "300000", "KWAGE_BATCH_BASES": "300", "KWAGE_EARLY_EXIT": "0"}):
            extra = {"KWAGE_NODE_REHEARSE": "1"} if ranks > 1 else {}
            r = subprocess.run([NODE] + args, cwd=cdir, capture_output=True, timeout=300,
                               env=_env(KWAGE_NODE_RANKS=str(ranks), KWAGE_NODE_STATS="1", **extra, **env))
            assert r.returncode == 0, r.stderr.decode()
            assert r.stdout == want.stdout, (ranks, args, env)
            passes = [int(x) for x in re.findall(rb"bytes per pass, (\d+) pass\(es\)", r.stderr)]
            assert len(passes) == ranks and len(set(passes)) == 1 and passes[0] >= 2, r.stderr.decode()      # every rank knows the same number of passes


@pytest.mark.gpu
def test_node_cli_a_failing_rank_ends_the_run(tmp_path):
    """A rank that cannot continue (here: a rehearsal segment too small for the hit list) must not leave the others waiting."""
    cdir = os.path.join(GOLDEN, "multi")
    r = subprocess.run([NODE, "-d", "dbs", "-i", "reads.fastq", "-t", "0.5", "--o.csv"], cwd=cdir, capture_output=True, timeout=120,
                       env=_env(KWAGE_NODE_RANKS="2", KWAGE_NODE_REHEARSE="1", KWAGE_NODE_REHEARSE_RECORDS="1"))
    assert r.returncode != 0 and b"rehearsal segment is too small" in r.stderr
    # rank 0 alone fails (it opens the output file); rank 1 is already on its way into the exchange
    r = subprocess.run([NODE, "-d", "dbs", "-i", "reads.fastq", "-t", "0.5", "--o.csv", "-o", "/no_such_dir/out.csv"], cwd=cdir, capture_output=True,
                       timeout=120, env=_env(KWAGE_NODE_RANKS="2", KWAGE_NODE_REHEARSE="1"))
    assert r.returncode != 0 and b"Unable to open" in r.stderr


def test_node_plan_is_the_python_hosts_partition(tmp_path, oracle):
    """KWAGE_NODE_PLAN=1 prints kwage_node's plan without touching a device: files of ragged widths in two parameter
    groups, 1 ... 9 ranks.  Every group's files must be dealt to the ranks exactly as kwage_amd.distributed.partition_files
    deals them (the file that holds a rank's middle column), every file's block must start on a 16-byte boundary behind its
    predecessor, and the global column bases must number groups and ranks in order without overlap."""
    import numpy as np
    from kwage_amd.distributed import partition_files
    rng = np.random.default_rng(5)
    widths = {"a": [int(x) for x in rng.integers(1, 300, size=17)], "b": [2048, 1, 2048, 7, 640]}
    for sub, (L, nh) in (("a", (10, 1)), ("b", (11, 2))):
        os.makedirs(tmp_path / "db" / sub)
        for f, n in enumerate(widths[sub]):
            rows = rng.integers(0, 256, size=(1 << L, (n + 7) // 8), dtype=np.uint8)
            infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % (1000 * f + j))) for j in range(n)]
            oracle.write_db(str(tmp_path / "db" / sub / ("f%02d.db" % f)), 31, nh, L, rows, n, infos)
    for ranks in (1, 2, 3, 5, 9):
        r = subprocess.run([NODE, "-d", "db", "-t", "1", "ACGTACGTACGTACGTACGTACGTACGTACGTAAA"], cwd=tmp_path, capture_output=True, timeout=60,
                           env=_env(KWAGE_NODE_RANKS=str(ranks), KWAGE_NODE_PLAN="1"))
        assert r.returncode == 0, r.stderr.decode()
        plan = json.loads(r.stdout)
        assert plan["ranks"] == ranks and len(plan["groups"]) == 2
        next_base = 0
        for g in plan["groups"]:
            sub = "a" if g["log_2_filter_len"] == 10 else "b"
            files = sorted(os.path.join("db", sub, "f%02d.db" % f) for f in range(len(widths[sub])))       # the CLI's order within a directory
            listed = [f["path"] for s in g["shares"] for f in s["files"]]
            assert sorted(listed) == files
            order = [p for p in listed]                       # rank after rank = the order the files were found in
            nf = [widths[sub][int(os.path.basename(p)[1:3])] for p in order]
            want = partition_files(nf, ranks)
            at = 0
            for s, (lo, hi) in zip(g["shares"], want):
                assert [f["path"] for f in s["files"]] == order[lo:hi], (ranks, sub, s["rank"])
                assert s["global_base"] == next_base
                span = 0
                for f in s["files"]:
                    span = (span + 15) // 16 * 16
                    assert f["first_column"] == span * 8 and f["num_filter"] == nf[at]
                    span += (f["num_filter"] + 7) // 8
                    at += 1
                assert s["span_columns"] == span * 8
                next_base += s["span_columns"]


def test_node_plan_cuts_passes_that_fit_the_budget(tmp_path, oracle):
    """KWAGE_NODE_PLAN=1 with KWAGE_MAX_GROUP_BYTES: every rank's passes as `plan_passes` cuts them (no device touched).  Every
    file of a rank's share lies in exactly one unit, in order; a unit's files follow each other on 16-byte boundaries; the
    matrices of a pass (128-byte row strides x 2^L rows) fit the budget -- except a unit of ONE file that alone exceeds it
    and has a pass to itself; the global base of a unit is its share's base plus the share-relative first column of its first
    file; all ranks have the same number of passes."""
    import numpy as np
    rng = np.random.default_rng(11)
    widths = {"a": [int(x) for x in rng.integers(1, 1500, size=23)], "b": [2048, 1, 2048, 7, 640, 4000]}
    lens = {"a": 10, "b": 12}
    for sub, nh in (("a", 1), ("b", 2)):
        os.makedirs(tmp_path / "db" / sub)
        for f, n in enumerate(widths[sub]):
            rows = rng.integers(0, 256, size=(1 << lens[sub], (n + 7) // 8), dtype=np.uint8)
            infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % (1000 * f + j))) for j in range(n)]
            oracle.write_db(str(tmp_path / "db" / sub / ("f%02d.db" % f)), 31, nh, lens[sub], rows, n, infos)
    nf_of = lambda p: widths[os.path.basename(os.path.dirname(p))][int(os.path.basename(p)[1:3])]
    L_of = lambda p: lens[os.path.basename(os.path.dirname(p))]
    for ranks in (1, 2, 4):
        for budget in (1, 128 << 10, 300 << 10, 1 << 20, 1 << 40):
            r = subprocess.run([NODE, "-d", "db", "-t", "1", "ACGTACGTACGTACGTACGTACGTACGTACGTAAA"], cwd=tmp_path, capture_output=True, timeout=60,
                               env=_env(KWAGE_NODE_RANKS=str(ranks), KWAGE_NODE_PLAN="1", KWAGE_MAX_GROUP_BYTES=str(budget)))
            assert r.returncode == 0, r.stderr.decode()
            plan = json.loads(r.stdout)
            assert plan["budget"] == budget and len(plan["rank_passes"]) == ranks
            assert all(len(ps) == plan["passes"] for ps in plan["rank_passes"])
            assert max(sum(1 for p in ps if p) for ps in plan["rank_passes"]) == plan["passes"] or plan["passes"] == 1       # no empty tail pass everywhere
            if budget == 1 << 40:
                assert plan["passes"] == 1
            for rank, passes in enumerate(plan["rank_passes"]):
                # the files of this rank's shares, group after group, in order
                share_files = [(gi, f["path"], g["shares"][rank]["global_base"], f["first_column"])
                               for gi, g in enumerate(plan["groups"]) for f in g["shares"][rank]["files"]]
                seen = []
                for units in passes:
                    used = 0
                    for u in units:
                        span = 0
                        for f in u["files"]:
                            span = (span + 15) // 16 * 16
                            assert f["first_column"] == span * 8
                            span += (nf_of(f["path"]) + 7) // 8
                            seen.append((u["group"], f["path"]))
                        assert u["span_columns"] == span * 8
                        first = next(x for x in share_files if x[1] == u["files"][0]["path"])
                        assert u["global_base"] == first[2] + first[3] and u["group"] == first[0]
                        assert len({L_of(f["path"]) for f in u["files"]}) == 1
                        used += ((span + 127) // 128 * 128) << L_of(u["files"][0]["path"])
                    alone = len(units) == 1 and len(units[0]["files"]) == 1
                    assert used <= budget or alone, (ranks, budget, rank, used)
                assert seen == [(gi, p) for gi, p, _, _ in share_files], (ranks, budget, rank)
