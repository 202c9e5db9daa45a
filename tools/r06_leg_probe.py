#!/usr/bin/env python3
"""round 6: the device leg of the sequence slices alone (GPU box): both haplotype BAMs of a sample loaded, 22 000 slices of
300 bases each asked of both AT THE SAME TIME (two threads, as COLLECT does), 15 times; median and spread of the slower
call.  SVX_BAM_LEG_PHASES / SVX_BAM_DEVICE_INFLATE / SVX_INFLATE_KERNEL in the environment choose what is measured.
    python tools/r06_leg_probe.py DIR"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from svim_asm_amd import _lib, bamio  # noqa: E402

d = sys.argv[1]
_lib.default_context(0)
files = [bamio.AlignmentFile(os.path.join(d, "hap%d.bam" % k), device=0, threads=bamio.quota_threads(2, 1)) for k in (1, 2)]
for f in files:
    f.device_inflate_percent = bamio.default_device_inflate_percent()
    f.load()
asks = []
for f in files:
    rng = np.random.default_rng(0)
    n = 22000
    l = f._cols["l_seq"]
    rec = np.sort(rng.integers(0, len(l), n))
    lo = (rng.random(n) * np.maximum(l[rec] - 500, 1)).astype(np.int64)
    o = np.lexsort((lo, rec))
    asks.append((rec[o], lo[o]))
both = []
for rep in range(16):
    took = [0.0, 0.0]

    def run(k):
        t = time.perf_counter()
        files[k].sequence_slices_raw(asks[k][0], asks[k][1], asks[k][1] + 300)
        took[k] = time.perf_counter() - t
    th = [threading.Thread(target=run, args=(k,)) for k in (0, 1)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    both.append(time.perf_counter() - t0)
both = sorted(both[1:])
print("phases %s share %s kernel %s: both calls done after median %.1f ms (min %.1f, max %.1f); members on device %s" % (
    os.environ.get("SVX_BAM_LEG_PHASES", "default"), files[0].effective_device_inflate_percent(), os.environ.get("SVX_INFLATE_KERNEL", "3"),
    both[len(both) // 2] * 1e3, both[0] * 1e3, both[-1] * 1e3, [f.device_members for f in files]))
