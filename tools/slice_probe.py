#!/usr/bin/env python3
"""Ingest scaling probe (GPU box): record walk and inserted-sequence decode of one BAM at several thread counts,
first and second call on the same handle (second call: the mapping's pages are already faulted in).
    python tools/slice_probe.py DIR/hap1.bam"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svim_asm_amd import bamio

path = sys.argv[1]
for thr in (8, 16, 32, 64, 128):
    t = time.perf_counter()
    f = bamio.AlignmentFile(path, threads=thr).load()
    t_load = time.perf_counter() - t
    rng = np.random.default_rng(0)
    n = 22000
    l = f._cols["l_seq"]
    rec = np.sort(rng.integers(0, len(l), n))
    lo = (rng.random(n) * np.maximum(l[rec] - 500, 1)).astype(np.int64)
    o = np.lexsort((lo, rec)); rec, lo = rec[o], lo[o]
    out = []
    for rep in range(3):
        t = time.perf_counter(); f.sequence_slices_raw(rec, lo, lo + 300); out.append(time.perf_counter() - t)
    print("threads %3d  load %.3f  slices first %.3f second %.3f third %.3f" % (thr, t_load, out[0], out[1], out[2]))
