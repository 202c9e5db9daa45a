"""The native text side (csrc/svx_text.cpp behind include/svx_text.h) under AddressSanitizer + UBSan and, as a second
build, ThreadSanitizer (both entry points work on several threads): FASTA batch fetches with valid, clipped and invalid
intervals, VCF bodies from random candidate columns — well-formed and with damaged indices, offsets and lengths.  Any
out-of-bounds access, use after free, signed overflow, data race or leak fails the test."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=["address,undefined", "thread"])
def driver(request, tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ is not here")
    exe = str(tmp_path_factory.mktemp("san") / "text_sanitize")
    cmd = [gxx, "-std=c++17", "-g", "-O1", "-fsanitize=" + request.param, "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "text_sanitize.cpp"),
           os.path.join(ROOT, "svim_asm_amd", "csrc", "svx_text.cpp"), "-lpthread", "-o", exe]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        pytest.skip("sanitizer build not possible here:\n" + res.stdout[-2000:])
    return exe


def test_text_side_is_clean_on_random_and_damaged_inputs(driver, tmp_path):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    res = subprocess.run([driver, str(tmp_path), "400"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env,
                         timeout=900)
    assert res.returncode == 0 and "text_sanitize ok" in res.stdout and "WARNING: ThreadSanitizer" not in res.stdout, \
        res.stdout[-4000:]
    assert " formatted, " in res.stdout and " rejected" in res.stdout
