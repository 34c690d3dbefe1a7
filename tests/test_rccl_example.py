"""examples/sharded_search_rccl.cpp: the C++ multi-GPU host shape of the path -- one process per GPU, database files
column-sharded over the ranks, per-GPU hit lists gathered to rank 0 over RCCL (all_gather of counts + grouped
send/recv).  A one-GPU box can only run it with one rank (RCCL refuses two ranks on one device): that still goes
through the fork, the communicator, the all_gather and the record path, and must report exactly the oracle's hits."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "kwage_amd", "bin", "sharded_search_rccl")


def test_usage_needs_no_gpu():
    if not os.path.exists(EXE):
        pytest.skip("example not built")
    r = subprocess.run([EXE, "1", "1.0", "x.db"], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr


@pytest.mark.gpu
def test_one_rank_gathers_the_oracle_hit_list(oracle, tmp_path):
    rng = np.random.default_rng(21)
    k, nh, L = 31, 2, 12
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, size=600)].tobytes().decode()
    files, images = [], []
    for f, ncol in enumerate((100, 2048, 77)):
        rows = (rng.random((1 << L, ((ncol + 7) // 8) * 8)) < 0.3)
        rows[:, ncol:] = False
        img = np.packbits(rows, axis=1, bitorder="little")
        for r in oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1):
            col = 5 + 3 * f
            img[r, col // 8] |= np.uint8(1 << (col % 8))
        infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%07d" % (f * 10000 + j))) for j in range(ncol)]
        p = str(tmp_path / ("part%d.db" % f))
        oracle.write_db(p, k, nh, L, img, ncol, infos)
        files.append(p); images.append((img, ncol))
    seqs = [genome[50:400], acgt[rng.integers(0, 4, size=200)].tobytes().decode(), "ACGT", genome[:100].lower()]
    for thr in ("1.0", "0.6"):
        r = subprocess.run([EXE, "1", thr] + files + ["--"] + seqs, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got = set()
        for line in r.stdout.splitlines():
            if line.startswith("query "):
                q, path, col, frac = line.split("\t")
                got.add((int(q.split()[1]), path, int(col.split()[1]), int(frac.split("/")[0]), int(frac.split("/")[1])))
        exp = set()
        for qi, s in enumerate(seqs):
            kmers = oracle.unique_kmers(s, k)
            for (img, ncol), path in zip(images, files):
                hits, _ = oracle.search_image(img, img.shape[1], k, nh, L, ncol, kmers, float(np.float32(float(thr))))
                exp |= {(qi, path, c, m, len(kmers)) for c, m in hits}
        assert got == exp and len(exp) >= 6
        assert "gathered from 1 rank(s) over RCCL" in r.stdout
