"""772 Mbp diploid sample (BASELINE config 3 at 1/4 of GRCh38: 24 contigs, 2 x 231 MB BAM): the VCF the REAL
reference wrote for it (oracle/make_golden.py large) must be reproduced byte for byte by the product CLI on the
GPU, as one process and as four contig-sharded ranks.  The VCF is several MB, so its SHA-256 (##fileDate masked),
size and record counts are committed instead of the text; the inputs are regenerated from fixed seeds and the digest of their
uncompressed content is checked against the generation-time one (a difference fails, it does not skip).  GPU only: regenerating the inputs takes seconds on the
GPU host and minutes in the build container."""
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
META = json.load(open(os.path.join(ROOT, "tests", "golden", "large_inputs.json")))


@pytest.fixture(scope="module")
def large_dataset(tmp_path_factory):
    from svim_asm_amd import synth, synth_bam
    prm = META["params"]
    contigs = tuple((n, max(60000, int(l * prm["scale"]))) for n, l in zip(synth.GRCH38_NAMES, synth.GRCH38_LENGTHS))
    n_shared = max(4, int(prm["sv_per_mbp"] * max(c[1] for c in contigs) / 1e6))
    d = str(tmp_path_factory.mktemp("large"))
    fasta, bams = synth_bam.write_dataset(d, seed=prm["seed"], contigs=contigs, diploid=True, n_shared=n_shared,
                                          n_private=max(2, n_shared // 5), median_aln=prm["median_aln"], mean_m=prm["mean_m"])
    from tests import helpers
    helpers.assert_inputs_are_the_golden_ones(META, [fasta] + bams)
    return fasta, bams


def _check(path):
    got = "".join(l for l in open(path) if not l.startswith("##fileDate="))
    body = [l for l in got.split("\n") if l and l[0] != "#"]
    assert len(body) == META["records"]
    assert [l[:200] for l in body[:3]] == META["first_records"] and [l[:200] for l in body[-3:]] == META["last_records"]
    assert len(got.encode()) == META["vcf_bytes"]
    assert hashlib.sha256(got.encode()).hexdigest() == META["vcf_sha256"]


def test_cli_reproduces_reference_vcf_large(svx_ctx, large_dataset, tmp_path):
    from svim_asm_amd import cli
    fasta, bams = large_dataset
    cli.main(["diploid", str(tmp_path), bams[0], bams[1], fasta])
    _check(tmp_path / "variants.vcf")


def test_cli_with_the_device_leg_reproduces_reference_vcf_large(svx_ctx, large_dataset, tmp_path, monkeypatch):
    """The same command with 60 % of every sequence-slice call's BGZF members inflated and verified on the device
    (svx_bam_set_device_inflate; a one-shot command runs without that share unless SVX_BAM_DEVICE_INFLATE asks for it)."""
    import time
    from svim_asm_amd import bamio, cli
    fasta, bams = large_dataset
    warm = bamio.AlignmentFile(bams[0], device=0)
    warm.load(["chr21"])   # the device lanes come up beside the first load of a process
    time.sleep(0.5)
    monkeypatch.setenv("SVX_BAM_DEVICE_INFLATE", "60")
    monkeypatch.setattr(bamio.AlignmentFile, "device_inflate_min_members", 0)  # (a quarter-size sample: 2 k members in the share)
    seen = []
    real = bamio.AlignmentFile._slices_native

    def spy(self, rec, a, b):
        out = real(self, rec, a, b)
        seen.append((len(rec), self.device_members))
        return out
    monkeypatch.setattr(bamio.AlignmentFile, "_slices_native", spy)
    cli.main(["diploid", str(tmp_path), bams[0], bams[1], fasta])
    _check(tmp_path / "variants.vcf")
    assert seen and max(n for n, _ in seen) >= 2048 and max(m for _, m in seen) > 1000, seen


@pytest.mark.spawns_gpu_children
def test_four_rank_cli_reproduces_reference_vcf_large(large_dataset, tmp_path):
    """BASELINE config 4: four ranks (fresh processes, product kernels, all on device 0 of the one-GPU box)."""
    from tests import helpers
    fasta, bams = large_dataset
    res = helpers.run_cli_ranks(["diploid", str(tmp_path), bams[0], bams[1], fasta], 4)
    for rank, (rc, text) in enumerate(res):
        assert rc == 0, "rank %d failed:\n%s" % (rank, text)
    _check(tmp_path / "variants.vcf")
