"""The reference-side binding, compiled and run: oracle/_ref/kwage_patched is the reference's own option parser, file
readers, MatchResult sort and CSV / JSON writers (compiled in place from its sources by `make -C oracle ref`) around
oracle/ref_driver/kwage_patched_main.cpp, which calls include/kwage_amd.h where the reference's main calls search()
(kwage.cpp:76-188 -> the C ABI).  On every case of tests/golden/manifest.json it must print what the unpatched reference
binary prints, byte for byte (both single-threaded: the reference's tie order depends on its thread merge order,
kwage.cpp:154-177), and what the committed expected output says."""
import json
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

REF = os.path.join(ROOT, "oracle", "_ref", "kwage")
PATCHED = os.path.join(ROOT, "oracle", "_ref", "kwage_patched")


def _cases():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))["cases"]


def _args(case):
    args = []
    for d in case["db"]:
        args += ["-d", d]
    for q in case["queries"]:
        args += ["-i", q]
    return args + ["-t", case["threshold"], "--o." + case["format"]] + case["cmdline"]


@pytest.mark.skipif(not (os.access(REF, os.X_OK) and os.access(PATCHED, os.X_OK)),
                    reason="oracle/_ref/ binaries were not built (they need /root/reference at build time)")
@pytest.mark.parametrize("case", _cases(), ids=lambda c: "%s-%s" % (c["name"], c["expected"]))
def test_patched_reference_prints_what_the_reference_prints(case):
    cdir = os.path.join(GOLDEN, case["name"])
    env = dict(os.environ, OMP_NUM_THREADS="1")
    ref = subprocess.run([REF] + _args(case), cwd=cdir, capture_output=True, env=env, timeout=300)
    got = subprocess.run([PATCHED] + _args(case), cwd=cdir, capture_output=True, env=env, timeout=300)
    assert ref.returncode == 0 and got.returncode == 0, (ref.stderr.decode(), got.stderr.decode())
    assert got.stdout == ref.stdout, "stdout differs from the reference binary's"
    secs = lambda b: re.sub(rb"in \d+ sec", b"in N sec", b)          # wall seconds: HIP initialisation takes one
    assert secs(got.stderr) == secs(ref.stderr)
    if len(case["db"]) == 1 and case["name"] != "multi":
        assert got.stdout == open(os.path.join(cdir, case["expected"]), "rb").read()      # the committed fixture (written by an OpenMP run of the reference)


@pytest.mark.skipif(not os.access(PATCHED, os.X_OK), reason="oracle/_ref/kwage_patched was not built")
def test_patched_reference_keeps_the_reference_command_line_and_errors(tmp_path):
    """The option surface IS the reference's (its own SearchOptions): usage, -o, a failing path."""
    cdir = os.path.join(GOLDEN, "basic")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for argv in ([], ["-h"], ["-d", "db", "-t", "7"], ["-d", "no_such_dir", "ACGT"]):
        a = subprocess.run([REF] + argv, cwd=cdir, capture_output=True, env=env, timeout=120) if os.access(REF, os.X_OK) else None
        b = subprocess.run([PATCHED] + argv, cwd=cdir, capture_output=True, env=env, timeout=120)
        if a is not None:
            assert (b.returncode, b.stdout, b.stderr) == (a.returncode, a.stdout, a.stderr), argv
    out, ref_out = tmp_path / "out.json", tmp_path / "ref.json"
    r = subprocess.run([PATCHED, "-d", "db", "-i", "q.fa", "-t", "0.8", "-o", str(out)], cwd=cdir, capture_output=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout == b"" and out.stat().st_size > 100
    if os.access(REF, os.X_OK):
        subprocess.run([REF, "-d", "db", "-i", "q.fa", "-t", "0.8", "-o", str(ref_out)], cwd=cdir, capture_output=True, env=env, timeout=300, check=True)
        assert out.read_bytes() == ref_out.read_bytes()
