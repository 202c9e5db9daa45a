"""Randomised differential test of the PRODUCT's host path (svim_asm_amd COLLECT / PAIR / VCF writer, the device
answered by the CPU oracle) directly against the REAL reference imported from /root/reference (stub pysam / edlib
of oracle/refstub).  Runs only in the build container — skipped where the reference is absent (GPU box)."""
import os
import tempfile

import numpy as np
import pytest

from svim_asm_amd import SVCandidate, SVIM_COLLECT, SVIM_COMBINE
from tests import helpers
from tests.test_oracle_vs_reference import LENGTHS, NAMES, _stub_bam

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src/svim_asm"),
                                reason="reference checkout not available")


@pytest.fixture(scope="module")
def ref():
    from oracle import make_golden
    return make_golden.load_reference()


@pytest.fixture(autouse=True)
def device_is_the_oracle(monkeypatch):
    helpers.oracle_backed_device(monkeypatch)


@pytest.mark.parametrize("seed", range(6))
def test_collect_matches_the_real_reference(ref, seed):
    rng = np.random.default_rng(5000 + seed)
    recs = helpers.random_records(rng, NAMES, LENGTHS, 60) + helpers.engineered_split_records(rng, NAMES, LENGTHS, 90)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    kw = [dict(), dict(min_sv_size=30, max_sv_size=3000), dict(min_mapq=0, query_gap_tolerance=500),
          dict(reference_overlap_tolerance=0, query_overlap_tolerance=0)][seed % 4]
    o = helpers.options(**kw)
    exp = [helpers.candidate_tuple(c) for c in ref["COLLECT"].analyze_alignment_file_coordsorted(_stub_bam(recs), o)]
    got = [helpers.candidate_tuple(c) for c in
           SVIM_COLLECT.analyze_alignment_file_coordsorted(helpers.FakeBam(NAMES, LENGTHS, recs), o)]
    assert got == exp


@pytest.mark.parametrize("seed", range(4))
def test_pair_and_vcf_match_the_real_reference(ref, seed, tmp_path):
    rng = np.random.default_rng(5100 + seed)
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=30000)) for n in NAMES}
    lengths = [30000] * len(NAMES)
    fasta = helpers.FakeFasta(seqs)

    class RefBam(object):
        references = tuple(NAMES)

        def get_reference_length(self, n):
            return lengths[NAMES.index(n)]
    bam = helpers.FakeBam(NAMES, lengths, [])
    t1 = helpers.random_candidates(rng, NAMES, lengths, seqs, 110, "h1")
    t2 = helpers.random_candidates(rng, NAMES, lengths, seqs, 110, "h2")
    for c in t1[:60]:
        if c[0] in ("DEL", "INS", "INV", "DUP_TAN"):
            shift = int(rng.integers(-3, 4))
            lst = list(c)
            lst[2] = max(0, c[2] + shift)
            lst[3] = max(lst[2], c[3] + shift)
            lst[{"DEL": 4, "INS": 4, "INV": 4, "DUP_TAN": 6}[c[0]]] = ("h2_copy",)
            t2.append(tuple(lst))
    o = helpers.options(max_edit_distance=[200, 10, 50, 0][seed], partition_max_distance=[1000, 100][seed % 2],
                        query_names=bool(seed % 2), tandem_duplications_as_insertions=bool(seed & 2))
    exp_objs = ref["COMBINE"].pair_candidates([helpers.build_candidate(t, RefBam(), ref["CAND"]) for t in t1],
                                              [helpers.build_candidate(t, RefBam(), ref["CAND"]) for t in t2],
                                              fasta, RefBam(), o)
    got_objs = SVIM_COMBINE.pair_candidates([helpers.build_candidate(t, bam, SVCandidate) for t in t1],
                                            [helpers.build_candidate(t, bam, SVCandidate) for t in t2], fasta, bam, o)
    assert [helpers.candidate_tuple(c) for c in got_objs] == [helpers.candidate_tuple(c) for c in exp_objs]
    # VCF text of both writers
    types = [t.strip() for t in o.types.split(",")]
    out = {}
    for tag, mod, objs in (("ref", ref["COMBINE"], exp_objs), ("got", SVIM_COMBINE, got_objs)):
        wd = tempfile.mkdtemp(dir=str(tmp_path))
        o.working_dir = wd
        by = lambda t: [c for c in objs if c.type == t]
        mod.write_final_vcf(by("DUP_INT"), by("INV"), by("DUP_TAN"), by("DEL"), by("INS"), by("BND"), "1.0.3", NAMES,
                            lengths, types, helpers.FakeFasta(seqs), o)
        out[tag] = "".join(l for l in open(os.path.join(wd, "variants.vcf")) if not l.startswith("##fileDate="))
    assert out["got"] == out["ref"]
