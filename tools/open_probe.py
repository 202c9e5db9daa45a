#!/usr/bin/env python3
"""Open / record walk / close of one BAM through the native reader, four times (GPU box host):
    python tools/open_probe.py DIR/hap1.bam"""
import sys, time
sys.path.insert(0,'/root/repo')
from svim_asm_amd import bamio, _lib
_lib.load()
for rep in range(4):
    t=time.perf_counter(); f=bamio.AlignmentFile(sys.argv[1], threads=32); t1=time.perf_counter()-t
    t=time.perf_counter(); f.load(); t2=time.perf_counter()-t
    t=time.perf_counter(); f.close(); t3=time.perf_counter()-t
    print("open %.4f load %.4f close %.4f"%(t1,t2,t3))
