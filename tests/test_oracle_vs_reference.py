"""Randomised differential test: oracle/svim_oracle.py vs the REAL reference imported from
/root/reference (with the stub pysam/edlib of oracle/refstub).  Runs only in the build
container — skipped where the reference is absent (e.g. on the GPU box)."""
import os
import tempfile

import numpy as np
import pytest

from oracle import svim_oracle
from tests import helpers

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src/svim_asm"),
                                reason="reference checkout not available")

NAMES = ["chr1", "chr10", "chr2", "chrX"]
LENGTHS = [3_000_000, 1_500_000, 2_000_000, 800_000]


@pytest.fixture(scope="module")
def ref():
    from oracle import make_golden
    return make_golden.load_reference()


def _stub_bam(records):
    import pysam  # stub, put on sys.path by load_reference()

    class Bam(object):
        references = tuple(NAMES)
        lengths = tuple(LENGTHS)

        def __init__(self):
            self.recs = []
            for r in records:
                a = pysam.AlignedSegment()
                a.query_name = r["qname"]
                a.flag = r["flag"]
                a.reference_id = r["tid"]
                a.reference_start = r["pos"]
                a.mapping_quality = r["mapq"]
                a.cigartuples = r["cigar"]
                a.query_sequence = r["seq"]
                if r.get("sa") is not None:
                    a.set_tags([("SA", r["sa"], "Z")])
                self.recs.append(a)

        def fetch(self, contig=None):
            tid = NAMES.index(contig)
            return iter([a for a in self.recs if a.reference_id == tid])

        def get_tid(self, n):
            return NAMES.index(n) if n in NAMES else -1

        def get_reference_name(self, tid):
            if tid < 0:
                raise ValueError("bad tid")
            return NAMES[tid]

        getrname = get_reference_name

        def get_reference_length(self, n):
            return LENGTHS[NAMES.index(n)]
    return Bam()


@pytest.mark.parametrize("seed", range(8))
def test_collect_matches_reference(ref, seed):
    rng = np.random.default_rng(seed)
    recs = helpers.random_records(rng, NAMES, LENGTHS, 60) + helpers.engineered_split_records(rng, NAMES, LENGTHS, 60)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    kw = [dict(), dict(min_sv_size=30, max_sv_size=3000), dict(min_mapq=0, query_gap_tolerance=500),
          dict(reference_overlap_tolerance=0, query_overlap_tolerance=0)][seed % 4]
    o = helpers.options(**kw)
    exp = [helpers.candidate_tuple(c) for c in ref["COLLECT"].analyze_alignment_file_coordsorted(_stub_bam(recs), o)]
    got = svim_oracle.collect(recs, NAMES, LENGTHS, o)
    assert got == exp
    if seed == 0:
        kinds = {c[0] for c in exp}
        assert {"DEL", "INS", "BND"} <= kinds


def test_all_sv_types_reachable(ref):
    rng = np.random.default_rng(123)
    recs = helpers.engineered_split_records(rng, NAMES, LENGTHS, 1500)
    got = svim_oracle.collect(recs, NAMES, LENGTHS, helpers.options())
    exp = [helpers.candidate_tuple(c) for c in
           ref["COLLECT"].analyze_alignment_file_coordsorted(_stub_bam(recs), helpers.options())]
    assert got == exp
    assert {c[0] for c in got} == {"DEL", "INS", "BND", "DUP_TAN", "DUP_INT", "INV"}


@pytest.mark.parametrize("seed", range(6))
def test_pair_and_vcf_match_reference(ref, seed):
    rng = np.random.default_rng(100 + seed)
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=min(l, 40000))) for n, l in zip(NAMES, LENGTHS)}
    lengths = [len(seqs[n]) for n in NAMES]
    fasta = helpers.FakeFasta(seqs)

    class Bam(object):
        references, lengths_ = tuple(NAMES), tuple(lengths)

        def get_reference_length(self, n):
            return lengths[NAMES.index(n)]
    bam = Bam()
    t1 = helpers.random_candidates(rng, NAMES, lengths, seqs, 120, "h1")
    t2 = helpers.random_candidates(rng, NAMES, lengths, seqs, 120, "h2")
    # a share of haplotype-2 candidates are near-copies of haplotype-1 ones (→ 1/1 calls)
    for c in t1[:60]:
        if c[0] in ("DEL", "INS", "INV", "DUP_TAN"):
            shift = int(rng.integers(-3, 4))
            lst = list(c)
            lst[2] = max(0, c[2] + shift)
            lst[3] = max(lst[2], c[3] + shift)
            rd_i = {"DEL": 4, "INS": 4, "INV": 4, "DUP_TAN": 6}[c[0]]
            lst[rd_i] = ("h2_copy",)
            t2.append(tuple(lst))
    o = helpers.options(max_edit_distance=[200, 10, 50][seed % 3], partition_max_distance=[1000, 100][seed % 2],
                        query_names=bool(seed % 2), tandem_duplications_as_insertions=bool(seed & 2),
                        interspersed_duplications_as_insertions=bool(seed & 4))
    c1 = [helpers.build_candidate(t, bam, ref["CAND"]) for t in t1]
    c2 = [helpers.build_candidate(t, bam, ref["CAND"]) for t in t2]
    exp_objs = ref["COMBINE"].pair_candidates(c1, c2, fasta, bam, o)
    exp = [helpers.candidate_tuple(c) for c in exp_objs]
    ref_lens = dict(zip(NAMES, lengths))
    from oracle import orc
    got = svim_oracle.pair_candidates(helpers.constructed_again(t1, ref_lens), helpers.constructed_again(t2, ref_lens), fasta.fetch, NAMES, lengths, ref_lens, o,
                                      edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    assert got == exp
    assert {c[-1] for c in got} == {"1/1", "1/0", "0/1"}
    # VCF text
    wd = tempfile.mkdtemp()
    o.working_dir = wd
    by = lambda t: [c for c in exp_objs if c.type == t]
    types = [t.strip() for t in o.types.split(",")]
    ref["COMBINE"].write_final_vcf(by("DUP_INT"), by("INV"), by("DUP_TAN"), by("DEL"), by("INS"), by("BND"), "1.0.3",
                                   NAMES, lengths, types, fasta, o)
    exp_vcf = "".join(l for l in open(os.path.join(wd, "variants.vcf")) if not l.startswith("##fileDate="))
    assert svim_oracle.vcf_text(got, fasta.fetch, NAMES, lengths, o) == exp_vcf
