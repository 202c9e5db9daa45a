#!/usr/bin/env python3
"""cProfile of the product's BAM -> VCF pipeline on the GPU box (host-side hot spots):
    python3 tools/e2e_profile.py --scale 0.25 [--top 40]"""
import argparse
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.25)
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    from tools import e2e_bench
    import tempfile
    d = tempfile.mkdtemp(prefix="svx_prof_")
    e2e_bench.run_e2e(scale=args.scale, keep=d, with_oracle=False, ranks=())  # generates + warms up
    pr = cProfile.Profile()
    pr.enable()
    r = e2e_bench.run_e2e(scale=args.scale, dataset=d, with_oracle=False, ranks=())
    pr.disable()
    print({k: r[k] for k in ("open_index_s", "collect_s", "pair_s", "vcf_s", "product_total_s")})
    pstats.Stats(pr).sort_stats("tottime").print_stats(args.top)
    import shutil
    shutil.rmtree(d)


if __name__ == "__main__":
    main()
