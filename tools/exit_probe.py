#!/usr/bin/env python3
"""What the end of a process costs once it holds a HIP context (GPU box): wall time of a child from its last line of
output to its exit, (a) context only, (b) context + a page-locked 64 MB buffer + a 256 MB device workspace, each leaving
through os._exit and through the interpreter's normal shutdown.
    python tools/exit_probe.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
from svim_asm_amd import _lib
import numpy as np
mode, how = sys.argv[1], sys.argv[2]
if mode != "none":
    ctx = _lib.Context(0)
    if mode == "work":
        a = np.arange(1 << 24, dtype=np.uint32)
        ctx.cigar_extract(a, np.array([0, len(a)], dtype=np.uint64), None, 40)
sys.stdout.write("%%.6f\n" %% time.time()); sys.stdout.flush()
if how == "fast":
    os._exit(0)
""" % ROOT

for mode in ("none", "ctx", "work"):
    for how in ("fast", "orderly"):
        best = None
        for _ in range(3):
            p = subprocess.Popen([sys.executable, "-c", CHILD, mode, how], stdout=subprocess.PIPE, text=True)
            line = p.stdout.readline()
            p.wait()
            t_end = time.time()
            d = t_end - float(line)
            best = d if best is None else min(best, d)
        print("%-5s %-8s last output -> exit: %.1f ms" % (mode, how, best * 1e3))
