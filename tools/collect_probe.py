#!/usr/bin/env python3
"""The two product operating points of bench.py on their own (GPU box), for rocprofv3:

    python tools/collect_probe.py [latency_case] [product_point]
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/collect_probe.py product_point

latency_case  = one config-2 sample per svx_collect_batch_dev (what `svim-asm haploid` submits per BAM)
product_point = both haplotype BAMs of a diploid sample in one submission (what `svim-asm diploid` submits)
Prints the legs' JSON objects (the same functions bench.py runs)."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["latency_case", "product_point"]
    args = types.SimpleNamespace(config=2, min_sv_size=40)
    for name in which:
        print(json.dumps({name: getattr(bench, name)(args, 0, torch)}), flush=True)


if __name__ == "__main__":
    main()
