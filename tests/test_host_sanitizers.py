"""Host half of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU
sanitizers are not available).  kwage_amd/csrc/host.cpp is compiled alone (it contains no HIP) with a
small driver that feeds the readers valid golden files and hundreds of truncated / bit-flipped copies."""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT


@pytest.mark.timeout(600)
def test_host_abi_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "host_sanitize_driver")
    # the device-side symbols host.cpp's make_bloom calls are not needed by the driver: stub them at link time
    stubs = tmp_path / "stubs.cpp"
    stubs.write_text('#include "kwage_amd.h"\n'
                     'extern "C" int kwage_batch_create(kwage_ctx*, const char*, const uint64_t*, uint32_t, kwage_batch**){ return KWAGE_ERR_DEVICE; }\n'
                     'extern "C" void kwage_batch_destroy(kwage_batch*){}\n'
                     'extern "C" int kwage_bloom_bits_from_batch(kwage_ctx*, const kwage_params*, kwage_batch*, void*, uint64_t*){ return KWAGE_ERR_DEVICE; }\n')
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "kwage_amd", "csrc"),
           os.path.join(ROOT, "kwage_amd", "csrc", "host.cpp"), os.path.join(ROOT, "tests", "native", "host_sanitize_driver.cpp"),
           str(stubs), "-o", exe, "-lz", "-pthread"]
    subprocess.check_call(cmd)
    work = tmp_path / "work"
    work.mkdir()
    r = subprocess.run([exe, GOLDEN, str(work)], capture_output=True, text=True,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "0 failure(s)" in r.stdout
