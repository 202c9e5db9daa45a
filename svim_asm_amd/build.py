"""Build libsvx.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m svim_asm_amd.build` or `svim_asm_amd.build.build_lib()`.  hipcc cross-compiles
without a GPU; the resulting .so is git-ignored but travels with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsvx.so")
SOURCES = ["svx_ctx.hip", "svx_cigar.hip", "svx_segments.hip", "svx_pair.hip", "svx_editdist.hip", "svx_linkage.hip", "svx_postpass.hip", "svx_bam.cpp"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, "svx_internal.h"), os.path.join(CSRC, "svx_linkage_dev.h"), os.path.join(ROOT, "include", "svx.h"),
                        os.path.join(ROOT, "include", "svx_bam.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False, out=None, defines=()):
    """Compile every HIP translation unit for gfx950 into svim_asm_amd/libsvx.so (or into `out`, with extra
    `defines` such as the SVX_EXP_* macros of the ablation / test builds)."""
    if out is not None:
        return _compile(out, list(defines), verbose)
    if not force and not is_stale():
        return LIB
    return _compile(LIB, [], verbose)


def _compile(LIB, defines, verbose):
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function"] + defines + [
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB + ".tmp"] + sources() + \
          ["-lz", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    if verbose and res.stdout.strip():
        print(res.stdout, file=sys.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
