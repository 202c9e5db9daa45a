"""62 Mbp diploid sample (24 GRCh38-proportioned contigs): the VCF written by the REAL reference
(tests/golden/medium_diploid.vcf.gz, made by oracle/make_golden.py) must be reproduced by the CPU
oracle (CPU test) and by the product CLI on the GPU (-m gpu).  The BAM/FASTA inputs are
regenerated from fixed seeds; the digest of their uncompressed content is checked against the generation-time one (a difference fails)."""
import gzip
import hashlib
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "medium_inputs.json")))


@pytest.fixture(scope="module")
def medium_dataset(tmp_path_factory):
    from svim_asm_amd import synth, synth_bam
    prm = META["params"]
    contigs = tuple((n, max(60000, int(l * prm["scale"]))) for n, l in zip(synth.GRCH38_NAMES, synth.GRCH38_LENGTHS))
    d = str(tmp_path_factory.mktemp("medium"))
    fasta, bams = synth_bam.write_dataset(d, seed=prm["seed"], contigs=contigs, n_shared=prm["n_shared"],
                                          n_private=prm["n_private"], median_aln=prm["median_aln"], mean_m=prm["mean_m"])
    from tests import helpers
    helpers.assert_inputs_are_the_golden_ones(META, [fasta] + bams)
    return fasta, bams


def expected_vcf():
    return gzip.open(os.path.join(GOLD, "medium_diploid.vcf.gz"), "rb").read().decode()


def test_oracle_reproduces_reference_vcf_medium(medium_dataset):
    from oracle import orc, run_oracle
    fasta, bams = medium_dataset
    got = run_oracle.vcf_from_files(bams, fasta, run_oracle.default_options(),
                                    edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    exp = expected_vcf()
    assert got == exp
    assert sum(1 for l in exp.split("\n") if l and l[0] != "#") == META["records"]


@pytest.mark.gpu
def test_cli_reproduces_reference_vcf_medium(svx_ctx, medium_dataset, tmp_path):
    from svim_asm_amd import cli
    fasta, bams = medium_dataset
    cli.main(["diploid", str(tmp_path), bams[0], bams[1], fasta])
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == expected_vcf()


@pytest.mark.gpu
@pytest.mark.spawns_gpu_children
def test_two_rank_cli_reproduces_reference_vcf_medium(medium_dataset, tmp_path):
    """BASELINE config 4 at the medium scale: two ranks (fresh processes, product kernels), contigs
    LPT-packed by their share of the BAM; each rank inflates about half of what one process would."""
    import glob
    import re
    from tests import helpers
    fasta, bams = medium_dataset
    res = helpers.run_cli_ranks(["diploid", str(tmp_path), bams[0], bams[1], fasta], 2)
    for rank, (rc, text) in enumerate(res):
        assert rc == 0, "rank %d failed:\n%s" % (rank, text)
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == expected_vcf()
    spanned = {}
    for path in glob.glob(str(tmp_path / "SVIM_*.log")):
        for m in re.finditer(r"INGEST: rank (\d+)/2 indexed (\d+) records \((\d+) of the (\d+) BGZF", open(path).read()):
            spanned.setdefault(int(m.group(1)), []).append(int(m.group(4)))
    from svim_asm_amd import bamio
    whole = bamio.AlignmentFile(bams[0])
    whole.load()
    total = whole.blocks_spanned
    for rank in (0, 1):
        assert 0.35 * total < spanned[rank][0] < 0.65 * total, (rank, spanned, total)


# ---- the same sample through non-default options (oracle/make_golden.py medium_options): read names in INFO,
# duplications written as insertions, symbolic alleles, strict pairing, a subset of the types
OPTION_RUNS = sorted(META.get("option_runs", {}))


def _expected(name):
    return gzip.open(os.path.join(GOLD, name + ".vcf.gz"), "rb").read().decode()


@pytest.mark.parametrize("name", OPTION_RUNS)
def test_oracle_reproduces_reference_vcf_medium_options(medium_dataset, name):
    from oracle import orc, run_oracle
    from tests.test_oracle_pins import _parse_run
    fasta, bams = medium_dataset
    _, kw = _parse_run(META["option_runs"][name]["options"])
    got = run_oracle.vcf_from_files(bams, fasta, run_oracle.default_options(**kw),
                                    edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    assert got == _expected(name)
    assert sum(1 for l in got.split("\n") if l and l[0] != "#") == META["option_runs"][name]["records"]


@pytest.mark.parametrize("name", OPTION_RUNS)
def test_host_path_reproduces_reference_vcf_medium_options(medium_dataset, tmp_path, monkeypatch, name):
    """The product's host path (native reader, columns, native VCF body) with the device answered by the oracle."""
    from svim_asm_amd import cli
    from tests import helpers
    helpers.oracle_backed_device(monkeypatch)
    fasta, bams = medium_dataset
    cli.main(["diploid", str(tmp_path), bams[0], bams[1], fasta] + META["option_runs"][name]["options"])
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == _expected(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", OPTION_RUNS)
def test_cli_reproduces_reference_vcf_medium_options(svx_ctx, medium_dataset, tmp_path, name):
    from svim_asm_amd import cli
    fasta, bams = medium_dataset
    cli.main(["diploid", str(tmp_path), bams[0], bams[1], fasta] + META["option_runs"][name]["options"])
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == _expected(name)


@pytest.mark.parametrize("how", ["zlib-6", "libdeflate-6", "prefix-only-no-crc", "csi-index"])
def test_host_path_on_other_writers_files_with_the_oracle_as_device(monkeypatch, tmp_path, how):
    """The CPU twin of the test below: reader, host logic and VCF writer of the product with the device answered by the
    oracle (no GPU)."""
    from tests import helpers
    helpers.oracle_backed_device(monkeypatch)
    _other_writers_case(tmp_path, how)


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["zlib-6", "libdeflate-6", "prefix-only-no-crc", "csi-index"])
def test_cli_reproduces_reference_vcf_on_other_writers_files(svx_ctx, tmp_path, how):
    _other_writers_case(tmp_path, how)


def _other_writers_case(tmp_path, how):
    """The same records in files as other writers make them — BGZF members deflated by zlib at level 6 (samtools'
    default), by libdeflate at level 6 (an htslib built with libdeflate: 4-bit literals almost only, the decoder's
    12-bit tables and literal runs) —, read with the opt-out of the whole-member CRC32 check, or indexed by a `.csi`: the product
    CLI writes the REAL reference's VCF for each."""
    from svim_asm_amd import bamio, cli, synth, synth_bam
    prm = META["params"]
    contigs = tuple((n, max(60000, int(l * prm["scale"]))) for n, l in zip(synth.GRCH38_NAMES, synth.GRCH38_LENGTHS))
    level = {"zlib-6": 6, "libdeflate-6": 106}.get(how, 1)
    d = str(tmp_path / "data")
    fasta, bams = synth_bam.write_dataset(d, seed=prm["seed"], contigs=contigs, n_shared=prm["n_shared"],
                                          n_private=prm["n_private"], median_aln=prm["median_aln"], mean_m=prm["mean_m"],
                                          level=level)
    from tests import helpers
    helpers.assert_inputs_are_the_golden_ones(META, [fasta] + bams)   # digests of the uncompressed content
    env = {}
    if how == "prefix-only-no-crc":
        env["SVX_BAM_VERIFY"] = "0"
    if how == "csi-index":
        for b in bams:
            os.remove(b + ".bai")
            bamio.index_bam(b, csi=True, min_shift=14, depth=6)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        out = tmp_path / "wd"
        cli.main(["diploid", str(out), bams[0], bams[1], fasta])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    got = "".join(l for l in open(out / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == expected_vcf()
