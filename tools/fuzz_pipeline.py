#!/usr/bin/env python3
"""COLLECT and PAIR of the product (GPU kernels + host mirror) against the pinned Python oracle on
fresh random record sets / candidate sets — the generators of tests/test_gpu_pipeline.py with new seeds.

    python tools/fuzz_pipeline.py [--seconds 180] [--seed 1000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import orc, svim_oracle  # noqa: E402
from svim_asm_amd import SVCandidate, SVIM_COLLECT, SVIM_COMBINE  # noqa: E402
from tests import helpers  # noqa: E402

NAMES = ["chr1", "chr10", "chr2", "chrX", "chrUn_1"]
LENGTHS = [3_000_000, 1_500_000, 2_000_000, 800_000, 50_000]


def collect_case(seed):
    rng = np.random.default_rng(seed)
    recs = helpers.random_records(rng, NAMES, LENGTHS, int(rng.integers(1, 200))) + \
        helpers.engineered_split_records(rng, NAMES, LENGTHS, int(rng.integers(0, 300)))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    kw = dict(min_sv_size=int(rng.choice([1, 30, 40, 300])), max_sv_size=int(rng.choice([1000, 3000, 100000])),
              min_mapq=int(rng.choice([0, 20, 60])), query_gap_tolerance=int(rng.choice([0, 50, 500])),
              query_overlap_tolerance=int(rng.choice([0, 50, 500])), reference_gap_tolerance=int(rng.choice([0, 50, 500])),
              reference_overlap_tolerance=int(rng.choice([0, 50, 500])))
    o = helpers.options(**kw)
    got = [helpers.candidate_tuple(c) for c in
           SVIM_COLLECT.analyze_alignment_file_coordsorted(helpers.FakeBam(NAMES, LENGTHS, recs), o)]
    return got == svim_oracle.collect(recs, NAMES, LENGTHS, o), "collect %s" % kw


def pair_case(seed):
    rng = np.random.default_rng(seed)
    L = 30000
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=L)) for n in NAMES}
    lengths = [L] * len(NAMES)
    ref, bam = helpers.FakeFasta(seqs), helpers.FakeBam(NAMES, lengths, [])
    t1 = helpers.random_candidates(rng, NAMES, lengths, seqs, int(rng.integers(0, 200)), "h1")
    t2 = helpers.random_candidates(rng, NAMES, lengths, seqs, int(rng.integers(0, 200)), "h2")
    for c in t1[: len(t1) // 2]:
        if c[0] in ("DEL", "INS", "INV", "DUP_TAN"):
            shift = int(rng.integers(-3, 4))
            lst = list(c)
            lst[2] = max(0, c[2] + shift)
            lst[3] = max(lst[2], c[3] + shift)
            lst[{"DEL": 4, "INS": 4, "INV": 4, "DUP_TAN": 6}[c[0]]] = ("h2_copy",)
            t2.append(tuple(lst))
    o = helpers.options(max_edit_distance=int(rng.choice([0, 10, 50, 200, 5000])),
                        partition_max_distance=int(rng.choice([1, 100, 1000, 100000])))
    c1 = [helpers.build_candidate(t, bam, SVCandidate) for t in t1]
    c2 = [helpers.build_candidate(t, bam, SVCandidate) for t in t2]
    got = [helpers.candidate_tuple(c) for c in SVIM_COMBINE.pair_candidates(c1, c2, ref, bam, o)]
    lens = dict(zip(NAMES, lengths))
    exp = svim_oracle.pair_candidates(helpers.constructed_again(t1, lens), helpers.constructed_again(t2, lens), ref.fetch, NAMES, lengths, lens, o,
                                      edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    return got == exp, "pair med %d pmd %d n %d/%d" % (o.max_edit_distance, o.partition_max_distance, len(t1), len(t2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=180.0)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--oracle-device", action="store_true",
                    help="no GPU here: answer every kernel call with the CPU oracle (tests/helpers.OracleBackedContext) — "
                         "the campaign then exercises the product's HOST logic (columns, constructors, pairing arithmetic)")
    a = ap.parse_args()
    if a.oracle_device:
        from svim_asm_amd import _lib
        ctx = helpers.OracleBackedContext()
        _lib.default_context = lambda device=0: ctx
    t0, seed, n = time.time(), a.seed, [0, 0]
    while time.time() - t0 < a.seconds:
        ok, what = (collect_case if seed % 2 == 0 else pair_case)(seed)
        if not ok:
            print("MISMATCH seed %d: %s" % (seed, what))
            sys.exit(1)
        n[seed % 2] += 1
        seed += 1
    print("fuzz ok: %d collect cases, %d pair cases, seeds %d..%d" % (n[0], n[1], a.seed, seed - 1))


if __name__ == "__main__":
    main()
