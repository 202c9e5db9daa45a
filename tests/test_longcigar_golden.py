"""Diploid sample whose contig-spanning alignments carry > 10^5 CIGAR operations, stored the way
samtools/htslib store them (placeholder `<l_seq>S<ref_len>N` + `CG:B,I`, SAM spec §4.2.2).  The VCF
written by the REAL reference (tests/golden/longcigar_diploid.vcf.gz, made by
`oracle/make_golden.py longcigar`; the reference saw the records through the stub pysam, which
restores the CIGAR like htslib's bam_tag2cigar) must be reproduced by the CPU oracle and by the
product CLI on the GPU.  Inputs are regenerated from fixed seeds and checked by the digest of their uncompressed content (a difference fails)."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "longcigar_inputs.json")))


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    from svim_asm_amd import synth_bam
    prm = META["params"]
    d = str(tmp_path_factory.mktemp("longcigar"))
    fasta, bams = synth_bam.write_dataset(d, seed=prm["seed"], contigs=tuple((n, l) for n, l in prm["contigs"]),
                                          n_shared=prm["n_shared"], n_private=prm["n_private"],
                                          median_aln=prm["median_aln"], mean_m=prm["mean_m"])
    from tests import helpers
    helpers.assert_inputs_are_the_golden_ones(META, [fasta] + bams)
    return fasta, bams


def expected_vcf():
    return gzip.open(os.path.join(GOLD, "longcigar_diploid.vcf.gz"), "rb").read().decode()


def test_inputs_really_use_the_cg_tag(dataset):
    from svim_asm_amd import bamio
    fasta, bams = dataset
    for path, n_ops in zip(bams, META["max_cigar_ops"]):
        raw = bamio.bgzf_decompress(path)
        assert raw.count(b"CGBI") >= 1          # the chr1 alignment exceeds 65535 operations
        nat = bamio.AlignmentFile(path)
        assert int(nat._cols["n_cig"].max()) == n_ops > 100000
        py = bamio.AlignmentFile(path, reader="python")
        assert np.array_equal(nat._cigar, py._cigar) and np.array_equal(nat._cols["ref_len"], py._cols["ref_len"])


def test_oracle_reproduces_reference_vcf_longcigar(dataset):
    from oracle import orc, run_oracle
    fasta, bams = dataset
    got = run_oracle.vcf_from_files(bams, fasta, run_oracle.default_options(),
                                    edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    exp = expected_vcf()
    assert got == exp
    assert sum(1 for l in exp.split("\n") if l and l[0] != "#") == META["records"]


@pytest.mark.gpu
def test_collect_on_long_cigars_matches_oracle(svx_ctx, dataset):
    """COLLECT on the GPU (a1+a2 over 145 k-operation alignments, a3 on their split reads) == oracle COLLECT
    on the records read by the independent stub reader."""
    from oracle import run_oracle
    from svim_asm_amd import SVIM_COLLECT, bamio
    from tests import helpers
    fasta, bams = dataset
    o = helpers.options()
    for path in bams:
        got = [helpers.candidate_tuple(c) for c in
               SVIM_COLLECT.analyze_alignment_file_coordsorted(bamio.AlignmentFile(path), o)]
        exp, _, _ = run_oracle.candidates_from_bam(path, o)
        assert got == exp and len(got) > 50


@pytest.mark.gpu
def test_cli_reproduces_reference_vcf_longcigar(svx_ctx, dataset, tmp_path):
    from svim_asm_amd import cli
    fasta, bams = dataset
    cli.main(["diploid", str(tmp_path), bams[0], bams[1], fasta])
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == expected_vcf()
