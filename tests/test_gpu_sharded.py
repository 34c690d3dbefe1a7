"""The sharded search END TO END with the HIP engine as the rank-local searcher: 2 and 4 ranks share device 0 and
exchange over gloo (RCCL refuses two ranks on one device), tests/sharded_worker.py is one rank.  Merged list on rank 0 ==
unsharded kwage_search == CPU oracle for t in {1.0, 0.7, 0.0001}, for the padded and p2p exchanges, the pipelined
exchange_counted, and the step pipeline + hit-proportional exchange (ranks with zero hits, a buffer that overflows on
one rank only); the golden multi/ database dealt out file by file (partition_files), a 40 000-column synthetic group
split at 1024-column boundaries (partition_columns).  What this replaces in the reference: the OpenMP loop over files and
its critical-section merge (kwage.cpp:76-87,154-177)."""
import os
import signal
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.timeout(900)
def test_sharded_hip_search_equals_unsharded_and_oracle(world, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True))
    outs, failed = [], False
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600 if not failed else 20)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)           # the exact process group this test started
            out, _ = p.communicate()
            out += "\n[killed: timeout]"
        failed = failed or p.returncode != 0
        outs.append(out)
    report = "\n".join("---- rank %d (exit %s) ----\n%s" % (rank, p.returncode, out[-3000:]) for rank, (p, out) in enumerate(zip(procs, outs)))
    assert all(p.returncode == 0 for p in procs), report
    assert all(os.path.exists(tmp_path / ("ok%d" % rank)) for rank in range(world)), report
    print(report)          # (shown with -s / on failure: every rank's timing laps)
