"""Host-side pieces of the CLI that need no GPU -- the read-ahead query source (order, error hand-over, early
destruction), the filing of hits under their queries, the report writers' number formatting against printf, the page-cache reader
process (starts, respects its look-ahead, stops, is reaped) --
compiled from the CLI's own translation unit under AddressSanitizer + UndefinedBehaviorSanitizer."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.timeout(600)
def test_cli_host_units_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "cli_units_driver")
    lib = os.path.join(ROOT, "kwage_amd", "lib")
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "kwage_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "cli_units_driver.cpp"), "-o", exe,
                           "-L", lib, "-lkwage_amd", "-Wl,-rpath," + lib, "-lz", "-pthread"])
    # the engine library is linked for its symbols only (nothing here touches the device); leak checking would report
    # the HIP runtime's start-up allocations
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "0 failure(s)" in r.stdout
