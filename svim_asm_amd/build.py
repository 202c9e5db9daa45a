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
SOURCES = ["svx_ctx.hip", "svx_cigar.hip", "svx_segments.hip", "svx_pair.hip", "svx_editdist.hip", "svx_linkage.hip", "svx_postpass.hip", "svx_collect.hip", "svx_inflate.hip", "svx_bam.cpp", "svx_text.cpp", "svx_pairhost.cpp"]


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
    deps = sources() + _headers()
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False, out=None, defines=()):
    """Compile every HIP translation unit for gfx950 into svim_asm_amd/libsvx.so (or into `out`, with extra
    `defines` such as the SVX_EXP_* macros of the ablation / test builds)."""
    if out is not None:
        return _compile(out, list(defines), verbose)
    if not force and not is_stale():
        return LIB
    return _compile(LIB, [], verbose, force)


def _headers():
    return [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + \
           [os.path.join(ROOT, "include", h) for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h")]


def _compile(LIB, defines, verbose, force=False):
    """One object per translation unit (compiled in parallel, kept under build/obj so that touching one source
    recompiles one file), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    import hashlib
    tag = hashlib.sha1(" ".join(defines).encode()).hexdigest()[:8] if defines else "default"
    objdir = os.path.join(ROOT, "build", "obj", tag)
    os.makedirs(objdir, exist_ok=True)
    common = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + \
        list(defines) + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    newest_header = max(os.path.getmtime(h) for h in _headers())

    def one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj, 0, ""
        cmd = common + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return obj, res.returncode, res.stdout

    with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as ex:
        results = list(ex.map(one, sources()))
    failed = [out for _, rc, out in results if rc != 0]
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    if verbose:
        for _, _, out in results:
            if out.strip():
                print(out, file=sys.stderr)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + [o for o, _, _ in results] + \
          ["-lz", "-ldl", "-lpthread"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
