#!/usr/bin/env python3
"""Differential campaign of the product's HOST path (columns, constructors, pairing arithmetic, native VCF body; the
device answered by the CPU oracle) against the REAL reference imported from /root/reference — the bodies of
tests/test_product_vs_reference_cpu.py with fresh seeds.  Build container only (the reference does not travel).

    python tools/fuzz_vs_reference.py [--seconds 600] [--seed 5000000]
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import make_golden  # noqa: E402
from svim_asm_amd import SVCandidate, SVIM_COLLECT, SVIM_COMBINE, _lib  # noqa: E402
from tests import helpers  # noqa: E402
from tests.test_oracle_vs_reference import LENGTHS, NAMES, _stub_bam  # noqa: E402


def collect_case(ref, seed):
    rng = np.random.default_rng(seed)
    recs = helpers.random_records(rng, NAMES, LENGTHS, int(rng.integers(1, 120))) + \
        helpers.engineered_split_records(rng, NAMES, LENGTHS, int(rng.integers(0, 160)))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    kw = dict(min_sv_size=int(rng.choice([1, 30, 40, 300])), max_sv_size=int(rng.choice([1000, 3000, 100000])),
              min_mapq=int(rng.choice([0, 20, 60])), query_gap_tolerance=int(rng.choice([0, 50, 500])),
              query_overlap_tolerance=int(rng.choice([0, 50, 500])), reference_gap_tolerance=int(rng.choice([0, 50, 500])),
              reference_overlap_tolerance=int(rng.choice([0, 50, 500])))
    o = helpers.options(**kw)
    exp = [helpers.candidate_tuple(c) for c in ref["COLLECT"].analyze_alignment_file_coordsorted(_stub_bam(recs), o)]
    got = [helpers.candidate_tuple(c) for c in
           SVIM_COLLECT.analyze_alignment_file_coordsorted(helpers.FakeBam(NAMES, LENGTHS, recs), o)]
    return got == exp, "collect %s" % kw


def pair_case(ref, seed):
    rng = np.random.default_rng(seed)
    L = 30000
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=L)) for n in NAMES}
    lengths = [L] * len(NAMES)
    fasta = helpers.FakeFasta(seqs)

    class RefBam(object):
        references = tuple(NAMES)

        def get_reference_length(self, n):
            return lengths[NAMES.index(n)]
    bam = helpers.FakeBam(NAMES, lengths, [])
    t1 = helpers.random_candidates(rng, NAMES, lengths, seqs, int(rng.integers(0, 160)), "h1")
    t2 = helpers.random_candidates(rng, NAMES, lengths, seqs, int(rng.integers(0, 160)), "h2")
    for c in t1[: len(t1) // 2]:
        if c[0] in ("DEL", "INS", "INV", "DUP_TAN"):
            shift = int(rng.integers(-3, 4))
            lst = list(c)
            lst[2] = max(0, c[2] + shift)
            lst[3] = max(lst[2], c[3] + shift)
            lst[{"DEL": 4, "INS": 4, "INV": 4, "DUP_TAN": 6}[c[0]]] = ("h2_copy",)
            t2.append(tuple(lst))
    o = helpers.options(max_edit_distance=int(rng.choice([0, 10, 50, 200, 5000])),
                        partition_max_distance=int(rng.choice([1, 100, 1000, 100000])),
                        query_names=bool(rng.integers(0, 2)), tandem_duplications_as_insertions=bool(rng.integers(0, 2)),
                        interspersed_duplications_as_insertions=bool(rng.integers(0, 2)), symbolic_alleles=bool(rng.random() < 0.25),
                        types=str(rng.choice(["DEL,INS,INV,DUP:TANDEM,DUP:INT,BND", "DEL,INS", "INV,BND,DUP:INT,DUP:TANDEM"])))
    exp_objs = ref["COMBINE"].pair_candidates([helpers.build_candidate(t, RefBam(), ref["CAND"]) for t in t1],
                                              [helpers.build_candidate(t, RefBam(), ref["CAND"]) for t in t2], fasta, RefBam(), o)
    got_objs = SVIM_COMBINE.pair_candidates([helpers.build_candidate(t, bam, SVCandidate) for t in t1],
                                            [helpers.build_candidate(t, bam, SVCandidate) for t in t2], fasta, bam, o)
    if [helpers.candidate_tuple(c) for c in got_objs] != [helpers.candidate_tuple(c) for c in exp_objs]:
        return False, "pair_candidates med %d pmd %d" % (o.max_edit_distance, o.partition_max_distance)
    types = [t.strip() for t in o.types.split(",")]
    out = {}
    for tag, mod, objs in (("ref", ref["COMBINE"], exp_objs), ("got", SVIM_COMBINE, got_objs)):
        wd = tempfile.mkdtemp(prefix="svx_fuzz_ref_")
        o.working_dir = wd
        by = lambda t: [c for c in objs if c.type == t]  # noqa: E731
        mod.write_final_vcf(by("DUP_INT"), by("INV"), by("DUP_TAN"), by("DEL"), by("INS"), by("BND"), "1.0.3", NAMES,
                            lengths, types, helpers.FakeFasta(seqs), o)
        out[tag] = "".join(l for l in open(os.path.join(wd, "variants.vcf")) if not l.startswith("##fileDate="))
        os.remove(os.path.join(wd, "variants.vcf"))
        os.rmdir(wd)
    return out["got"] == out["ref"], "vcf names %s sym %s types %s" % (o.query_names, o.symbolic_alleles, o.types)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--seed", type=int, default=5000000)
    a = ap.parse_args()
    ref = make_golden.load_reference()
    ctx = helpers.OracleBackedContext()
    _lib.default_context = lambda device=0: ctx
    t0, seed, n = time.time(), a.seed, [0, 0]
    while time.time() - t0 < a.seconds:
        ok, what = (collect_case if seed % 2 == 0 else pair_case)(ref, seed)
        if not ok:
            print("MISMATCH seed %d: %s" % (seed, what))
            sys.exit(1)
        n[seed % 2] += 1
        seed += 1
    print("fuzz vs the real reference ok: %d COLLECT cases, %d PAIR + VCF cases, seeds %d..%d" % (n[0], n[1], a.seed, seed - 1))


if __name__ == "__main__":
    main()
