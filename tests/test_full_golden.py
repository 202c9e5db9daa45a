"""The configuration the metric is quoted on, under `-m gpu`: the full-size synthetic human diploid sample (BASELINE
config 3 at GRCh38 contig lengths — 3.09 Gbp, 24 contigs, 2 x 924 MB BAM, 3.16 M CIGAR ops per haplotype).  The VCF the
REAL reference wrote for it in the build container (oracle/make_golden.py full: 167 s on one core) must be reproduced
byte for byte by the product command line on the GPU, as one process and as two contig-sharded ranks (config 4).
37 MB of VCF: its SHA-256 (##fileDate masked), size, record counts and first / last records are committed
(tests/golden/full_inputs.json) instead of the text; the inputs are regenerated from fixed seeds (about a minute on the
GPU host) and the digest of their uncompressed content is compared with the generation-time one — a difference FAILS."""
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
META = json.load(open(os.path.join(ROOT, "tests", "golden", "full_inputs.json")))


@pytest.fixture(scope="module")
def full_dataset(tmp_path_factory):
    from svim_asm_amd import synth_bam
    from tools import e2e_bench
    prm = META["params"]
    d = str(tmp_path_factory.mktemp("full"))
    fasta, bams = synth_bam.write_dataset(d, **e2e_bench.dataset_args(prm["scale"], prm["sv_per_mbp"], prm["mean_m"], prm["seed"]))
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


def test_cli_reproduces_reference_vcf_full_size(svx_ctx, full_dataset, tmp_path):
    from svim_asm_amd import cli
    fasta, bams = full_dataset
    cli.main(["diploid", str(tmp_path), bams[0], bams[1], fasta])
    _check(tmp_path / "variants.vcf")


@pytest.mark.spawns_gpu_children
def test_two_rank_cli_reproduces_reference_vcf_full_size(full_dataset, tmp_path):
    """BASELINE config 4 on the full-size sample: two ranks (fresh processes, product kernels, both on device 0 of the
    one-GPU box), contigs LPT-packed over them, rank 0 writes the VCF."""
    from tests import helpers
    fasta, bams = full_dataset
    res = helpers.run_cli_ranks(["diploid", str(tmp_path), bams[0], bams[1], fasta], 2)
    for rank, (rc, text) in enumerate(res):
        assert rc == 0, "rank %d failed:\n%s" % (rank, text)
    _check(tmp_path / "variants.vcf")
