"""The HOST logic of the product path — COLLECT gathering and assembly, split-read candidates, PAIR recipes and
clustering glue, the CLI and the VCF writer — with the device answered by the CPU oracle
(tests/helpers.OracleBackedContext), so that it is checked in the CPU-only suite as well: against the pinned Python
oracle on seeded inputs, against the function vectors the REAL reference produced, and through the `svim-asm`
command line against the real reference's golden VCFs.  The -m gpu suite makes the same comparisons with the
real kernels."""
import os

import numpy as np
import pytest

from oracle import orc, svim_oracle
from svim_asm_amd import SVCandidate, SVIM_COLLECT, SVIM_COMBINE, SVIM_inter
from tests import helpers
from tests.test_oracle_pins import RUNS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = ["chr1", "chr10", "chr2", "chrX"]
LENGTHS = [3_000_000, 1_500_000, 2_000_000, 800_000]


@pytest.fixture(autouse=True)
def device_is_the_oracle(monkeypatch):
    helpers.oracle_backed_device(monkeypatch)


@pytest.mark.parametrize("seed", range(3))
def test_collect_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    recs = helpers.random_records(rng, NAMES, LENGTHS, 60) + helpers.engineered_split_records(rng, NAMES, LENGTHS, 120)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    o = helpers.options(**[dict(), dict(min_sv_size=30, max_sv_size=3000), dict(min_mapq=0, query_gap_tolerance=500)][seed])
    got = [helpers.candidate_tuple(c) for c in
           SVIM_COLLECT.analyze_alignment_file_coordsorted(helpers.FakeBam(NAMES, LENGTHS, recs), o)]
    assert got == svim_oracle.collect(recs, NAMES, LENGTHS, o)


@pytest.mark.parametrize("chunks", [1, 3])
@pytest.mark.parametrize("seed", range(3))
def test_pair_candidates_matches_oracle(seed, chunks, monkeypatch):
    # (chunks = 3: the distance jobs pipelined in chunks of whole partitions, as PAIR does for crowded samples — windows
    #  and recipes of chunk i + 1 built while a worker thread holds the device for chunk i)
    monkeypatch.setattr(SVIM_COMBINE, "_PAIR_CHUNK_MIN_JOBS", 1 if chunks > 1 else 10 ** 9)
    monkeypatch.setattr(SVIM_COMBINE, "_PAIR_CHUNKS", chunks)
    rng = np.random.default_rng(300 + seed)
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=30000)) for n in NAMES}
    lengths = [30000] * len(NAMES)
    ref, bam = helpers.FakeFasta(seqs), helpers.FakeBam(NAMES, lengths, [])
    t1 = helpers.random_candidates(rng, NAMES, lengths, seqs, 120, "h1")
    t2 = helpers.random_candidates(rng, NAMES, lengths, seqs, 120, "h2")
    for c in t1[:70]:  # near-copies on the other haplotype: two-member partitions on both sides of the threshold
        if c[0] in ("DEL", "INS", "INV", "DUP_TAN"):
            shift = int(rng.integers(-3, 4))
            lst = list(c)
            lst[2] = max(0, c[2] + shift)
            lst[3] = max(lst[2], c[3] + shift)
            lst[{"DEL": 4, "INS": 4, "INV": 4, "DUP_TAN": 6}[c[0]]] = ("h2_copy",)
            t2.append(tuple(lst))
    o = helpers.options(max_edit_distance=[200, 10, -1][seed], partition_max_distance=[1000, 100, 1000][seed])
    c1 = [helpers.build_candidate(t, bam, SVCandidate) for t in t1]
    c2 = [helpers.build_candidate(t, bam, SVCandidate) for t in t2]
    got = [helpers.candidate_tuple(c) for c in SVIM_COMBINE.pair_candidates(c1, c2, ref, bam, o)]
    lens = dict(zip(NAMES, lengths))
    exp = svim_oracle.pair_candidates(helpers.constructed_again(t1, lens), helpers.constructed_again(t2, lens), ref.fetch, NAMES, lengths, lens, o,
                                      edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    assert got == exp


def test_function_vectors_from_the_real_reference():
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
            assert [helpers.candidate_tuple(c) for c in SVIM_inter.analyze_read_segments(aln, supp, bam, o)] == pr["out"]
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
            assert [[where[id(c)] for _, c in p] for p in SVIM_COMBINE.form_partitions(sub, o.partition_max_distance)] == parts
        assert [helpers.candidate_tuple(c) for c in SVIM_COMBINE.pair_candidates(c1, c2, ref, bam, o)] == case["out"]


@pytest.mark.parametrize("name", sorted(RUNS))
def test_cli_reproduces_reference_vcf_config1(tmp_path, name):
    """`svim-asm haploid|diploid` (native BAM reader, host logic, VCF writer) on the config-1 BAMs == the VCF the
    real reference wrote, the kernels answered by the oracle."""
    from svim_asm_amd import cli
    argv = list(RUNS[name])
    argv[1] = str(tmp_path)
    for i, a in enumerate(argv):
        if a.endswith(".bam") or a.endswith(".fa"):
            argv[i] = os.path.join(GOLD, "config1", a)
    cli.main(argv)
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == open(os.path.join(GOLD, "config1", name + ".vcf")).read()


def test_cli_refuses_a_member_damaged_behind_the_bytes_it_needs(tmp_path):
    """htslib checks the CRC32 of every BGZF block it reads (bgzf_read_block under bam.fetch, SVIM_COLLECT.py:65-68): the
    reference aborts on a damaged block.  So does the product by default — also when the damage lies in the tail of a
    member, BEHIND the last byte a record walk or a sequence slice needs (literal bytes of a stored or little-compressed
    block changed: the stream still decodes) — and no VCF is written; with --no_bgzf_crc the same run goes through."""
    import shutil
    from svim_asm_amd import bamio, cli
    src = os.path.join(GOLD, "config1")
    d = tmp_path / "in"
    shutil.copytree(src, d)
    bam = str(d / "hap1.bam")
    data = bytearray(open(bam, "rb").read())
    spans = bamio._bgzf_block_spans(bytes(data))
    # the last bytes of the deflate payload of the biggest member: the end of its SEQ / QUAL bytes
    st, ln = max(((sp[0], sp[1]) for sp in spans if sp[2]), key=lambda x: x[1])
    # find a damage that still inflates (so only the CRC32 can notice it)
    import zlib
    good = zlib.decompress(bytes(data[st:st + ln]), -15)
    done = False
    for back in range(3, min(ln, 4000)):
        for bit in (1, 2, 4, 8, 16, 32, 64, 128):
            trial = bytearray(data[st:st + ln])
            trial[ln - back] ^= bit
            try:
                out = zlib.decompress(bytes(trial), -15)
            except zlib.error:
                continue
            if len(out) == len(good) and out != good and out[:len(good) // 2] == good[:len(good) // 2]:
                data[st + ln - back] ^= bit
                done = True
                break
        if done:
            break
    assert done, "no damage found that leaves a valid deflate stream"
    open(bam, "wb").write(bytes(data))
    argv = ["diploid", str(tmp_path / "wd"), bam, str(d / "hap2.bam"), str(d / "ref.fa")]
    with pytest.raises(ValueError):
        cli.main(argv)
    assert not os.path.exists(tmp_path / "wd" / "variants.vcf")
    # the opt-out reads only what it needs and does not notice damage behind it ... or notices it, when the damaged
    # bytes are among the ones it needs: either way never a crash
    try:
        cli.main(argv[:1] + [str(tmp_path / "wd2")] + argv[2:] + ["--no_bgzf_crc"])
    except ValueError:
        pass


def test_two_haplotypes_with_differently_ordered_headers():
    """The two BAMs of a diploid run need not list their contigs in the same order: COLLECT then submits them
    separately (contig ids mean different things) and PAIR re-expresses the second table's contig ids by NAME in the
    first BAM's header — candidates, pairing and order as the oracle computes them from names."""
    rng = np.random.default_rng(42)
    perm = [2, 0, 3, 1]
    names2, lengths2 = [NAMES[i] for i in perm], [LENGTHS[i] for i in perm]
    recs1 = helpers.random_records(rng, NAMES, LENGTHS, 50) + helpers.engineered_split_records(rng, NAMES, LENGTHS, 60)
    recs1.sort(key=lambda r: (r["tid"], r["pos"]))
    recs2 = helpers.random_records(rng, names2, lengths2, 50) + helpers.engineered_split_records(rng, names2, lengths2, 60)
    recs2.sort(key=lambda r: (r["tid"], r["pos"]))
    o = helpers.options()
    bam1, bam2 = helpers.FakeBam(NAMES, LENGTHS, recs1), helpers.FakeBam(names2, lengths2, recs2)
    t1, t2 = SVIM_COLLECT.collect_tables([bam1, bam2], o)
    exp1, exp2 = svim_oracle.collect(recs1, NAMES, LENGTHS, o), svim_oracle.collect(recs2, names2, lengths2, o)
    assert [helpers.candidate_tuple(c) for c in t1.objects()] == exp1
    assert [helpers.candidate_tuple(c) for c in t2.objects()] == exp2
    seqs = {n: "".join(rng.choice(list("ACGT"), size=l)) for n, l in zip(NAMES, LENGTHS)}
    ref = helpers.FakeFasta(seqs)
    got = [helpers.candidate_tuple(c) for c in SVIM_COMBINE.pair_tables(t1, t2, ref, bam1, o).objects()]
    exp = svim_oracle.pair_candidates(exp1, exp2, ref.fetch, NAMES, LENGTHS, dict(zip(NAMES, LENGTHS)), o,
                                      edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    assert got == exp


def test_breakend_whose_ends_coincide_flips_at_every_construction():
    """CandidateBreakend keeps its ends only when source < dest (contig, then position, both strict): ends on the same
    position are swapped and both directions flipped by every construction — the one in COLLECT and the one PAIR makes
    for its output (SVCandidate.py:352-373, SVIM_COMBINE.py:340-366).  Found by tools/fuzz_pipeline.py seed 980353
    (as a disagreement of the harness's two sides, the product agreeing with the real reference)."""
    lengths = [30000] * len(NAMES)
    lens = dict(zip(NAMES, lengths))
    bam = helpers.FakeBam(NAMES, lengths, [])
    ref = helpers.FakeFasta({n: "A" * 30000 for n in NAMES})
    raw = ("BND", "chr10", 3154, "fwd", "chr10", 3154, "fwd", ("r",), "1/1")
    once = helpers.build_candidate(raw, bam, SVCandidate)
    assert (once.source_direction, once.dest_direction) == ("rev", "rev")
    out = SVIM_COMBINE.pair_candidates([], [once], ref, bam, helpers.options())
    assert [helpers.candidate_tuple(c) for c in out] == [("BND", "chr10", 3154, "fwd", "chr10", 3154, "fwd", ("r",), "0/1")]
    exp = svim_oracle.pair_candidates([], helpers.constructed_again([raw], lens), ref.fetch, NAMES, lengths, lens,
                                      helpers.options())
    assert [tuple(x) for x in exp] == [("BND", "chr10", 3154, "fwd", "chr10", 3154, "fwd", ("r",), "0/1")]


def test_submission_groups_respect_the_op_limit():
    """COLLECT cuts a cohort into several submissions where one would reach svx_collect_batch's 2^32-op limit."""
    class Rec(object):
        def __init__(self, n):
            self.cig_off = np.array([0, n], np.int64)

    class S(object):
        def __init__(self, n, extra):
            self.rec, self.extra_words = Rec(n), [np.zeros(e, np.uint32) for e in extra]
    samples = [S(1000, [3, 3]), S(3000, []), S(500, [10]), S(2000, [1]), S(100, [])]
    groups = SVIM_COLLECT._submission_groups(samples, True, max_ops=4000)
    assert [[samples.index(s) for s in g] for g in groups] == [[0], [1, 2], [3, 4]]
    assert SVIM_COLLECT._submission_groups(samples, True) == [samples]           # far below 2^32: one submission
    assert SVIM_COLLECT._submission_groups(samples, False) == [[s] for s in samples]  # headers differ: one per file
    big = [S((1 << 31) + 5, []), S((1 << 31) + 5, []), S(7, [])]
    assert [len(g) for g in SVIM_COLLECT._submission_groups(big, True)] == [1, 2]


def test_cohort_command_writes_every_samples_vcf(tmp_path, monkeypatch):
    """svim-asm-cohort (host logic; the device answered by the oracle): three diploid samples in one process — every
    sample's VCF is the one the real reference wrote for it alone.  (write_final_vcf closes its FastaFile,
    SVIM_COMBINE.py:466-467: the cohort opens the genome once per sample.)"""
    from svim_asm_amd import cli, cohort
    monkeypatch.setattr(cli, "_warm_device", lambda device: None)
    g = os.path.join(GOLD, "config1")
    rows = [("s1", "hap1.bam", "hap2.bam"), ("s2", "hap1.bam", "hap2.bam"), ("s3", "hap1.bam", "hap2.bam")]
    manifest = tmp_path / "cohort.tsv"
    manifest.write_text("".join("%s %s %s\n" % (tmp_path / wd, os.path.join(g, a), os.path.join(g, b)) for wd, a, b in rows))
    assert cohort.main(["diploid", str(manifest), os.path.join(g, "ref.fa")]) == 0
    for wd, _, _ in rows:
        got = "".join(l for l in open(tmp_path / wd / "variants.vcf") if not l.startswith("##fileDate="))
        assert got == open(os.path.join(g, "diploid_default.vcf")).read()


def test_tables_built_side_by_side_give_the_same_vcf(tmp_path, monkeypatch):
    """The tables of a crowded sample's two haplotypes are built on threads (SVIM_COLLECT.collect_tables): forced
    here on the config-1 BAMs — the VCF is the real reference's."""
    from svim_asm_amd import cli
    monkeypatch.setattr(SVIM_COLLECT, "_TABLES_SIDE_BY_SIDE_FROM", 0)
    g = os.path.join(GOLD, "config1")
    cli.main(["diploid", str(tmp_path), os.path.join(g, "hap1.bam"), os.path.join(g, "hap2.bam"), os.path.join(g, "ref.fa")])
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == open(os.path.join(g, "diploid_default.vcf")).read()
