"""The native BAM reader (csrc/svx_bam.cpp behind include/svx_bam.h) under AddressSanitizer + UBSan on the CPU:
every entry point on the reference's own fixtures, the config-1 goldens and a sample with CG-tag long CIGARs,
then on hundreds of damaged copies (flipped bytes, truncations, overwritten length fields, damaged or missing
.bai).  A damaged file may be refused or read as what it now says; any out-of-bounds access, use after free,
signed overflow or leak fails the test; a second build runs the same under ThreadSanitizer (the reader inflates,
indexes and decodes on its own threads).  (GPU sanitizers are not available on the pool; this is the host side of
the ingest, which parses untrusted input.)"""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module", params=["address,undefined", "thread"])
def driver(request, tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("g++ or the HIP headers are not here")
    exe = str(tmp_path_factory.mktemp("san") / "bam_sanitize")
    cmd = [gxx, "-std=c++17", "-g", "-O1", "-fsanitize=" + request.param, "-fno-sanitize-recover=all",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "svim_asm_amd", "csrc"), os.path.join(ROOT, "tests", "native", "bam_sanitize.cpp"),
           os.path.join(ROOT, "svim_asm_amd", "csrc", "svx_bam.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-lpthread",
           "-ldl", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        pytest.skip("sanitizer build not possible here:\n" + res.stdout[-2000:])
    return exe


def _run(exe, scratch, mutations, files):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    res = subprocess.run([exe, str(scratch), str(mutations)] + files, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, env=env, timeout=900)
    assert res.returncode == 0 and "bam_sanitize ok" in res.stdout and "WARNING: ThreadSanitizer" not in res.stdout, \
        res.stdout[-4000:]
    return res.stdout


def test_reader_is_clean_on_fixtures_and_their_damaged_copies(driver, tmp_path):
    files = [os.path.join(GOLD, "chimeric_read.bam"), os.path.join(GOLD, "chimeric_read_errors.bam"),
             os.path.join(GOLD, "config1", "hap1.bam"), os.path.join(GOLD, "config1", "hap2.bam")]
    out = _run(driver, tmp_path, 150, files)
    assert " read," in out


def test_reader_is_clean_with_csi_indices_and_their_damaged_copies(driver, tmp_path):
    """The same with a `.csi` (BGZF-compressed CSI v1) as the index: intact, with random bytes in it, absent."""
    import shutil as sh
    from svim_asm_amd import bamio
    files = []
    for k, name in enumerate(("hap1.bam", "hap2.bam")):
        p = str(tmp_path / ("c%d.bam" % k))
        sh.copy(os.path.join(GOLD, "config1", name), p)
        bamio.index_bam(p, csi=True, min_shift=14 - 2 * k, depth=5 + k)
        files.append(p)
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    out = _run(driver, scratch, 120, files)
    assert " read," in out


def test_reader_is_clean_on_long_cigar_records_and_their_damaged_copies(driver, tmp_path):
    """Records of > 65 535 CIGAR operations (CG:B,I form) that span many BGZF blocks, with a real .bai."""
    from svim_asm_amd import synth_bam
    prm = json.load(open(os.path.join(GOLD, "longcigar_inputs.json")))["params"]
    d = tmp_path / "data"
    d.mkdir()
    _, bams = synth_bam.write_dataset(str(d), seed=prm["seed"], contigs=tuple((n, l) for n, l in prm["contigs"]),
                                      n_shared=prm["n_shared"], n_private=prm["n_private"], median_aln=prm["median_aln"],
                                      mean_m=prm["mean_m"])
    _run(driver, tmp_path, 40, list(bams))


def test_reader_is_clean_on_spec_written_files_and_their_damaged_copies(driver, tmp_path):
    """The awkward cases of tests/test_bam_spec_cases.py — files written from the SAM specification alone by
    tests/spec_bam_writer.py: records across many members, empty members, every aux type around SA, stored members,
    a split block_size field, bins of every level, .bai with and without pseudo-bins — and damaged copies of them."""
    from tests import test_bam_spec_cases as cases
    d = tmp_path / "spec"
    d.mkdir()
    files = [cases.write_case(d, name)[0] for name in sorted(cases.CASES) if name not in ("largest_members", "cigar_of_70000_operations")]
    assert len(files) >= 20
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    out = _run(driver, scratch, 25, files)
    assert " read," in out
