"""End-to-end parity of the product path on the GPU: reference function seams, COLLECT, PAIR and
the `svim-asm` command line against the pinned oracle and the reference's golden VCFs."""
import json
import os

import numpy as np
import pytest

from oracle import orc, run_oracle, svim_oracle
from svim_asm_amd import SVCandidate, SVIM_COLLECT, SVIM_COMBINE, SVIM_inter, SVIM_intra, bamio, cli
from tests import helpers
from tests.test_oracle_pins import KNOWN, RUNS, _parse_run

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = ["chr1", "chr10", "chr2", "chrX"]
LENGTHS = [3_000_000, 1_500_000, 2_000_000, 800_000]


def test_analyze_cigar_indel_seam_known_answers(svx_ctx):
    for tuples, expected in KNOWN:  # reference tests/test_intra.py
        assert SVIM_intra.analyze_cigar_indel(tuples, 30) == expected
    assert SVIM_intra.analyze_cigar_indel([], 30) == []
    vec = json.load(open(os.path.join(GOLD, "functions.json")))["analyze_cigar_indel"]
    for case in vec:
        assert SVIM_intra.analyze_cigar_indel([tuple(t) for t in case["tuples"]], case["min_length"]) == \
            [tuple(x) for x in case["out"]]


def test_is_similar_seam():
    assert not SVIM_inter.is_similar("chrI", 0, 100, "chrII", 0, 100)   # reference tests/test_inter.py
    assert SVIM_inter.is_similar("chrI", 0, 100, "chrI", 0, 100)
    assert SVIM_inter.is_similar("chrI", 0, 100, "chrI", 10, 90)
    assert not SVIM_inter.is_similar("chrI", 0, 100, "chrI", 21, 100)


@pytest.mark.parametrize("seed", range(6))
def test_collect_matches_oracle(svx_ctx, seed):
    rng = np.random.default_rng(seed)
    recs = helpers.random_records(rng, NAMES, LENGTHS, 150) + helpers.engineered_split_records(rng, NAMES, LENGTHS, 300)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    kw = [dict(), dict(min_sv_size=30, max_sv_size=3000), dict(min_mapq=0, query_gap_tolerance=500),
          dict(reference_overlap_tolerance=0, query_overlap_tolerance=0)][seed % 4]
    o = helpers.options(**kw)
    got = [helpers.candidate_tuple(c) for c in
           SVIM_COLLECT.analyze_alignment_file_coordsorted(helpers.FakeBam(NAMES, LENGTHS, recs), o)]
    exp = svim_oracle.collect(recs, NAMES, LENGTHS, o)
    assert got == exp
    assert {"DEL", "INS", "BND", "DUP_TAN", "INV"} <= {c[0] for c in exp}


def test_per_alignment_seams_match_batch(svx_ctx):
    rng = np.random.default_rng(42)
    recs = helpers.engineered_split_records(rng, NAMES, LENGTHS, 40)
    bam = helpers.FakeBam(NAMES, LENGTHS, recs)
    o = helpers.options()
    lens = dict(zip(NAMES, LENGTHS))
    for rec, aln in zip(recs, bam.fetch()):
        got = [helpers.candidate_tuple(c) for c in SVIM_intra.analyze_alignment_indel(aln, bam, aln.query_name, o)]
        assert got == svim_oracle.analyze_alignment_indel(rec, NAMES, lens, o.min_sv_size)
        supp = [s for s in SVIM_COLLECT.retrieve_other_alignments(aln, bam) if s.mapping_quality >= o.min_mapq]
        osupp = [s for s in svim_oracle.retrieve_other_alignments(rec, NAMES) if s["mapq"] >= o.min_mapq]
        got = [helpers.candidate_tuple(c) for c in SVIM_inter.analyze_read_segments(aln, supp, bam, o)]
        assert got == svim_oracle.analyze_read_segments(rec, osupp, NAMES, lens, o)


@pytest.mark.parametrize("seed", range(5))
def test_pair_candidates_matches_oracle(svx_ctx, seed):
    rng = np.random.default_rng(200 + seed)
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=40000)) for n in NAMES}
    lengths = [40000] * len(NAMES)
    ref = helpers.FakeFasta(seqs)
    bam = helpers.FakeBam(NAMES, lengths, [])
    t1 = helpers.random_candidates(rng, NAMES, lengths, seqs, 150, "h1")
    t2 = helpers.random_candidates(rng, NAMES, lengths, seqs, 150, "h2")
    for c in t1[:80]:
        if c[0] in ("DEL", "INS", "INV", "DUP_TAN"):
            shift = int(rng.integers(-3, 4))
            lst = list(c)
            lst[2] = max(0, c[2] + shift)
            lst[3] = max(lst[2], c[3] + shift)
            lst[{"DEL": 4, "INS": 4, "INV": 4, "DUP_TAN": 6}[c[0]]] = ("h2_copy",)
            t2.append(tuple(lst))
    o = helpers.options(max_edit_distance=[200, 10, 50][seed % 3], partition_max_distance=[1000, 100][seed % 2])
    c1 = [helpers.build_candidate(t, bam, SVCandidate) for t in t1]
    c2 = [helpers.build_candidate(t, bam, SVCandidate) for t in t2]
    got = [helpers.candidate_tuple(c) for c in SVIM_COMBINE.pair_candidates(c1, c2, ref, bam, o)]
    lens = dict(zip(NAMES, lengths))
    exp = svim_oracle.pair_candidates(helpers.constructed_again(t1, lens), helpers.constructed_again(t2, lens), ref.fetch, NAMES, lengths, lens, o,
                                      edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    assert got == exp
    assert {c[-1] for c in got} == {"1/1", "1/0", "0/1"}
    # form_partitions seam, type by type
    for typ in SVIM_COMBINE.TYPE_ORDER:
        sub = [(1, c) for c in c1 if c.type == typ] + [(2, c) for c in c2 if c.type == typ]
        osub = [(1, t) for t in t1 if t[0] == typ] + [(2, t) for t in t2 if t[0] == typ]
        gp = [[(h, helpers.candidate_tuple(c)) for h, c in p] for p in SVIM_COMBINE.form_partitions(sub, o.partition_max_distance)]
        assert gp == svim_oracle.form_partitions(osub, o.partition_max_distance)


def test_pipeline_vectors_from_the_real_reference(svx_ctx):
    """The product's COLLECT, analyze_read_segments, form_partitions and pair_candidates against vectors
    the REAL reference produced in the build container (oracle/make_golden.py pipeline)."""
    vec = helpers.load_pipeline_vectors()
    names, lengths = vec["names"], vec["lengths"]
    for case in vec["collect"]:
        o = helpers.options(**case["options"])
        bam = helpers.FakeBam(names, lengths, case["records"])
        got = [helpers.candidate_tuple(c) for c in SVIM_COLLECT.analyze_alignment_file_coordsorted(bam, o)]
        assert got == case["out"]
        alns = list(bam.fetch())
        for pr in case["analyze_read_segments"]:
            aln = alns[pr["record"]]
            supp = [s for s in SVIM_COLLECT.retrieve_other_alignments(aln, bam)
                    if not s.is_unmapped and s.mapping_quality >= o.min_mapq]
            got = [helpers.candidate_tuple(c) for c in SVIM_inter.analyze_read_segments(aln, supp, bam, o)]
            assert got == pr["out"]
    for case in vec["pair"]:
        o = helpers.options(**case["options"])
        seqs = case["seqs"]
        plen = [len(seqs[n]) for n in names]
        ref, bam = helpers.FakeFasta(seqs), helpers.FakeBam(names, plen, [])
        c1 = [helpers.build_candidate(t, bam, SVCandidate) for t in case["t1"]]
        c2 = [helpers.build_candidate(t, bam, SVCandidate) for t in case["t2"]]
        for typ, parts in case["form_partitions"].items():
            sub = [(1, c) for c in c1 if c.type == typ] + [(2, c) for c in c2 if c.type == typ]
            where = {id(c): k for k, (_, c) in enumerate(sub)}
            got = [[where[id(c)] for _, c in p] for p in SVIM_COMBINE.form_partitions(sub, o.partition_max_distance)]
            assert got == parts
        got = [helpers.candidate_tuple(c) for c in SVIM_COMBINE.pair_candidates(c1, c2, ref, bam, o)]
        assert got == case["out"]


@pytest.mark.parametrize("name", sorted(RUNS))
def test_cli_reproduces_reference_vcf(svx_ctx, tmp_path, name):
    """`svim-asm haploid|diploid` on the config-1 BAMs == the VCF written by the real reference."""
    pos, kw = _parse_run(RUNS[name])
    argv = list(RUNS[name])
    argv[1] = str(tmp_path)
    for i, a in enumerate(argv):
        if a.endswith(".bam") or a.endswith(".fa"):
            argv[i] = os.path.join(GOLD, "config1", a)
    cli.main(argv)
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == open(os.path.join(GOLD, "config1", name + ".vcf")).read()


def test_collect_on_real_bam_matches_oracle_reader(svx_ctx):
    """bamio-based COLLECT on the golden BAM == oracle COLLECT on records read by the independent stub reader."""
    path = os.path.join(GOLD, "config1", "hap2.bam")
    o = helpers.options()
    got = [helpers.candidate_tuple(c) for c in SVIM_COLLECT.analyze_alignment_file_coordsorted(bamio.AlignmentFile(path), o)]
    exp, _, _ = run_oracle.candidates_from_bam(path, o)
    assert got == exp and len(got) > 50


def test_large_alleles_pair_like_the_oracle(svx_ctx):
    """Partitions of 3+ members whose pairwise haplotype distances are far beyond any band that fits
    in LDS (> 16 000 edits, the limit of the round-1 kernel): scipy's dendrogram above the cut, hence
    the cluster label order and which member supplies the output coordinates, must be the reference's.
    Also: thresholds of any size are accepted (SVIM_COMBINE.py:134-139)."""
    rng = np.random.default_rng(31)
    names, lengths = ["chrL", "chrM"], [260000, 90000]
    seqs = {n: "".join(rng.choice(list("ACGT"), size=l)) for n, l in zip(names, lengths)}
    lens = dict(zip(names, lengths))
    ref, bam = helpers.FakeFasta(seqs), helpers.FakeBam(names, lengths, [])
    from oracle import svim_oracle as O
    c = 130000
    t1 = [O.cand_del("chrL", c - 10000, c + 10000, ["a"], lens),             # same midpoint, very different sizes
          O.cand_del("chrL", c - 2500, c + 2600, ["a2"], lens),
          O.cand_tan("chrM", 30000, 36000, 3, True, ["t1"], lens),
          O.cand_inv("chrM", 60000, 79000, ["i1"], True, lens)]
    t2 = [O.cand_del("chrL", c - 19000, c + 19000, ["b"], lens),
          O.cand_del("chrL", c - 14000, c + 14005, ["c"], lens),
          O.cand_del("chrL", c - 2500, c + 2600, ["c2"], lens),
          O.cand_tan("chrM", 30010, 36020, 1, True, ["t2"], lens),
          O.cand_tan("chrM", 30000, 36000, 3, False, ["t3"], lens),
          O.cand_inv("chrM", 60100, 78900, ["i2"], False, lens),
          O.cand_inv("chrM", 59000, 80000, ["i3"], True, lens)]
    for med in (200, 25000, 3_000_000_000):
        o = helpers.options(max_edit_distance=med)
        c1 = [helpers.build_candidate(t, bam, SVCandidate) for t in t1]
        c2 = [helpers.build_candidate(t, bam, SVCandidate) for t in t2]
        got = [helpers.candidate_tuple(x) for x in SVIM_COMBINE.pair_candidates(c1, c2, ref, bam, o)]
        exp = svim_oracle.pair_candidates(t1, t2, ref.fetch, names, lengths, lens, o,
                                          edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
        assert got == exp, med
    # the compatibility seam returns a distance for any pair of haplotypes, however far apart
    a, b = (1, helpers.build_candidate(t1[0], bam, SVCandidate)), (2, helpers.build_candidate(t2[0], bam, SVCandidate))
    h = SVIM_COMBINE.haplotype_pair(a[1], b[1], ref)
    assert SVIM_COMBINE.compute_distance(a, b, ref) == orc.edit_distance(h[0].encode(), h[1].encode()) > 16000


def test_cohort_command_writes_the_single_sample_vcfs(svx_ctx, tmp_path):
    """svim-asm-cohort: three diploid samples (the config-1 BAMs in both orders) write, per sample, the VCF the real
    reference wrote for that sample alone — as a stream (a group per sample, the default number of worker threads, each with
    a device context of its own), with COLLECT of all six BAMs as ONE device submission (--cohort_group 0), the same with
    that submission cut into several by the op-count limit (here lowered to a few thousand ops), and as groups of two
    samples over three workers."""
    from svim_asm_amd import SVIM_COLLECT, cohort
    g = os.path.join(GOLD, "config1")
    rows = [("s1", "hap1.bam", "hap2.bam", "diploid_default"), ("s2", "hap2.bam", "hap1.bam", None), ("s3", "hap1.bam", "hap2.bam", "diploid_default")]
    manifest = tmp_path / "cohort.tsv"
    manifest.write_text("# working_dir bam1 bam2\n" + "".join("%s %s %s\n" % (tmp_path / wd, os.path.join(g, a), os.path.join(g, b)) for wd, a, b, _ in rows))
    for limit, extra in ((None, []), (None, ["--cohort_group", "0"]), (4000, ["--cohort_group=0", "--cohort_workers", "1"]),
                         (None, ["--cohort_workers", "3", "--cohort_group", "2", "--cohort_lanes", "2"])):
        old = SVIM_COLLECT.MAX_OPS_PER_SUBMISSION
        if limit:
            SVIM_COLLECT.MAX_OPS_PER_SUBMISSION = limit
        try:
            assert cohort.main(["diploid", str(manifest), os.path.join(g, "ref.fa")] + extra) == 0
        finally:
            SVIM_COLLECT.MAX_OPS_PER_SUBMISSION = old
        for wd, a, b, golden in rows:
            got = "".join(l for l in open(tmp_path / wd / "variants.vcf") if not l.startswith("##fileDate="))
            if golden:
                assert got == open(os.path.join(g, golden + ".vcf")).read()
            else:  # the swapped order has no golden of its own: the single-sample command is the reference
                single = tmp_path / "single"
                cli.main(["diploid", str(single), os.path.join(g, a), os.path.join(g, b), os.path.join(g, "ref.fa")])
                assert got == "".join(l for l in open(single / "variants.vcf") if not l.startswith("##fileDate="))
            os.remove(tmp_path / wd / "variants.vcf")
