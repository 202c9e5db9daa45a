#!/usr/bin/env python3
"""Where a fresh `svim-asm` process spends its start-up (GPU box): interpreter, imports, library load, HIP context.
    python tools/startup_probe.py"""
import time
t0 = time.perf_counter()
import os, sys  # noqa: E401,E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
marks = [("start", t0)]
import numpy  # noqa: E402,F401
marks.append(("import numpy", time.perf_counter()))
from svim_asm_amd import cli  # noqa: E402,F401
marks.append(("import svim_asm_amd.cli", time.perf_counter()))
from svim_asm_amd import _lib  # noqa: E402
_lib.load()
marks.append(("load libsvx.so", time.perf_counter()))
ctx = _lib.Context(0)
marks.append(("svx_ctx_create (HIP init)", time.perf_counter()))
a = numpy.arange(1 << 20, dtype=numpy.uint32)
ctx.cigar_extract(a, numpy.array([0, len(a)], dtype=numpy.uint64), None, 40)
marks.append(("first kernel call", time.perf_counter()))
ctx.cigar_extract(a, numpy.array([0, len(a)], dtype=numpy.uint64), None, 40)
marks.append(("second kernel call", time.perf_counter()))
for (n0, a0), (n1, a1) in zip(marks, marks[1:]):
    print("%-32s %7.1f ms" % (n1, (a1 - a0) * 1e3))
print("modules loaded:", len(sys.modules), "torch" in sys.modules)
