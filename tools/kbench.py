#!/usr/bin/env python3
"""Run single legs of bench.py's extras (for rocprofv3 and A/B work):
    python3 tools/kbench.py latency|pair|editdist [...]
prints one JSON object per leg."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse
    import torch
    import bench
    args = argparse.Namespace(config=2, min_sv_size=40)
    for leg in sys.argv[1:]:
        if leg == "latency":
            print(json.dumps(bench.latency_case(args, 0, torch)))
        elif leg == "pair":
            print(json.dumps(bench.roofline_pair(0)))
        elif leg == "editdist":
            print(json.dumps(bench.roofline_editdist(0, torch.cuda.get_device_properties(0).multi_processor_count)))
        else:
            raise SystemExit("unknown leg " + leg)


if __name__ == "__main__":
    main()
