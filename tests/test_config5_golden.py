"""BASELINE config 5 as a diploid end-to-end case (oracle/make_golden.py config5): 10x the small-indel density
(mean M run 400 bp) and SV events packed closely enough that neighbouring events chain inside the pairing
step's 1000 bp — 184 009 candidates (beyond the one-launch limit of the pair sort: its radix plan runs in the
product path), tens of thousands of partitions with 3..10 members (complete linkage in scipy's label order) and
partitions with more than 10 members (dropped, SVIM_COMBINE.py:126-128).  The VCF the REAL reference wrote
(digest, size, counts, first / last records committed) must be reproduced byte for byte by the product CLI on the
GPU.  Inputs are regenerated from fixed seeds and identified by digests of their uncompressed content (a
difference fails)."""
import hashlib
import json
import logging
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
META = json.load(open(os.path.join(ROOT, "tests", "golden", "config5_inputs.json")))


@pytest.fixture(scope="module")
def config5_dataset(tmp_path_factory):
    from svim_asm_amd import synth_bam
    from tests import helpers
    from tools import e2e_bench
    prm = META["params"]
    d = str(tmp_path_factory.mktemp("config5"))
    fasta, bams = synth_bam.write_dataset(d, **e2e_bench.dataset_args(prm["scale"], prm["sv_per_mbp"], prm["mean_m"],
                                                                      prm["seed"], prm["min_gap"]))
    helpers.assert_inputs_are_the_golden_ones(META, [fasta] + bams)
    return fasta, bams


def _check(path):
    got = "".join(l for l in open(path) if not l.startswith("##fileDate="))
    body = [l for l in got.split("\n") if l and l[0] != "#"]
    assert len(body) == META["records"]
    kinds = {}
    for l in body:
        k = l.split("\t")[2].rsplit(".", 1)[0]
        kinds[k] = kinds.get(k, 0) + 1
    assert kinds == META["records_by_id_prefix"]
    assert [l[:200] for l in body[:3]] == META["first_records"] and [l[:200] for l in body[-3:]] == META["last_records"]
    assert len(got.encode()) == META["vcf_bytes"]
    assert hashlib.sha256(got.encode()).hexdigest() == META["vcf_sha256"]


def test_cli_reproduces_reference_vcf_config5(svx_ctx, config5_dataset, tmp_path, caplog):
    from svim_asm_amd import cli
    fasta, bams = config5_dataset
    with caplog.at_level(logging.INFO):
        cli.main(["diploid", str(tmp_path), bams[0], bams[1], fasta])
    _check(tmp_path / "variants.vcf")
    # the same candidate counts went into PAIR as in the reference's run: the sort really saw > 131072 keys
    pairing = [r.getMessage() for r in caplog.records if r.getMessage().startswith("Pairing ")]
    assert pairing == META["reference_log_pairing_lines"]
    assert sum(int(l.split()[1]) for l in pairing) > 131072


def test_config5_has_crowded_and_dropped_partitions(svx_ctx, config5_dataset):
    """The sample really exercises what it is for: partitions of 3..10 members and of more than 10."""
    import numpy as np
    from svim_asm_amd import SVIM_COLLECT, SVIM_COMBINE, bamio
    from svim_asm_amd.table import CandidateTable
    from tests import helpers
    fasta, bams = config5_dataset
    o = helpers.options()
    f1, f2 = bamio.AlignmentFile(bams[0]), bamio.AlignmentFile(bams[1])
    t1, t2 = SVIM_COLLECT.collect_tables([f1, f2], o)
    T = CandidateTable.concat([t1, t2], list(f1.references), list(f1.lengths))
    keys = SVIM_COMBINE._keys_of_table(T)
    perm, part_id, n_parts = svx_ctx.pair_partition(keys, o.partition_max_distance)
    sizes = np.bincount(part_id.astype(np.int64), minlength=n_parts)
    assert len(keys) > 131072
    assert int(((sizes >= 3) & (sizes <= 10)).sum()) > 10000
    assert int((sizes > 10).sum()) > 50


@pytest.mark.spawns_gpu_children
def test_three_rank_cli_reproduces_reference_vcf_config5(config5_dataset, tmp_path):
    """BASELINE config 4 on the config-5 sample: three contig-sharded ranks (fresh processes, product kernels, one
    device), 184 k candidates exchanged as tables, PAIR sharded by key contig over crowded partitions."""
    from tests import helpers
    fasta, bams = config5_dataset
    res = helpers.run_cli_ranks(["diploid", str(tmp_path), bams[0], bams[1], fasta], 3)
    for rank, (rc, text) in enumerate(res):
        assert rc == 0, "rank %d failed:\n%s" % (rank, text)
    _check(tmp_path / "variants.vcf")
