#!/usr/bin/env python3
"""PAIR stage clocks of the config-5 diploid sample (GPU box): python tools/pair_c5_probe.py DIR  (DIR from
`tools/e2e_bench.py --config5 --keep DIR`); prints pair_distances_s and the total of five runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svim_asm_amd import _lib, bamio, SVIM_COLLECT, SVIM_COMBINE
from svim_asm_amd.fasta import FastaFile
from svim_asm_amd.SVIM_input_parsing import parse_arguments
d = sys.argv[1]
bams = [d + "/hap1.bam", d + "/hap2.bam"]
o = parse_arguments("1.0.3", ["diploid", d + "/wd_probe", bams[0], bams[1], d + "/ref.fa"])
f1, f2 = bamio.AlignmentFile(bams[0]), bamio.AlignmentFile(bams[1])
t1, t2 = SVIM_COLLECT.collect_tables([f1, f2], o)
out = []
for rep in range(5):
    ref = FastaFile(d + "/ref.fa")
    t = time.perf_counter(); p = SVIM_COMBINE.pair_tables(t1, t2, ref, f1, o); dt = time.perf_counter() - t
    out.append((SVIM_COMBINE.LAST_TIMING["pair_distances_s"], dt))
print("rule %s: distances ms %s  pair ms %s  rows %d" % (os.environ.get("SVX_EXP_WFA_RULE", "-"), [round(a * 1e3, 1) for a, _ in out],
                                                        [round(b * 1e3, 1) for _, b in out], len(p)))
