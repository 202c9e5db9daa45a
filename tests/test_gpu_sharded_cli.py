"""BASELINE config 4 on the HIP path: `svim-asm diploid` started as two ranks (fresh processes, one
process per rank as under torch.distributed.run; both on device 0 because the GPU box has one GPU)
runs the product's sharded COLLECT and PAIR with the real kernels and must write the same VCF as
the reference.  Each rank walks only the BGZF ranges of the contigs it owns."""
import glob
import os
import re

import pytest

from tests import helpers

pytestmark = [pytest.mark.gpu, pytest.mark.spawns_gpu_children]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "config1")


def _ingest_lines(workdir):
    """rank -> [(records, inflated, spanned)] parsed from the per-rank log files."""
    out = {}
    for path in glob.glob(os.path.join(workdir, "SVIM_*.log")):
        for m in re.finditer(r"INGEST: rank (\d+)/(\d+) indexed (\d+) records \((\d+) of the (\d+) BGZF", open(path).read()):
            out.setdefault(int(m.group(1)), []).append((int(m.group(3)), int(m.group(4)), int(m.group(5))))
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_cli_reproduces_reference_vcf(tmp_path, world):
    argv = ["diploid", str(tmp_path), os.path.join(GOLD, "hap1.bam"), os.path.join(GOLD, "hap2.bam"),
            os.path.join(GOLD, "ref.fa")]
    res = helpers.run_cli_ranks(argv, world)
    for rank, (rc, text) in enumerate(res):
        assert rc == 0, "rank %d failed:\n%s" % (rank, text)
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == open(os.path.join(GOLD, "diploid_default.vcf")).read()
    ingest = _ingest_lines(str(tmp_path))
    assert sorted(ingest) == list(range(world)) and all(len(v) == 2 for v in ingest.values())
    # no rank indexed the whole file, and together they indexed every placed record exactly once
    from svim_asm_amd import bamio
    for h, name in enumerate(("hap1.bam", "hap2.bam")):
        f = bamio.AlignmentFile(os.path.join(GOLD, name), reader="python")
        placed = int((f._cols["tid"] >= 0).sum())
        per_rank = [ingest[r][h][0] for r in range(world)]
        assert sum(per_rank) == placed and max(per_rank) < placed
    # only rank 0 talks on the console
    assert "Done." in res[0][1] and all("Done." not in text for _, text in res[1:])


def test_sharded_cli_failure_is_loud(tmp_path):
    """A rank that fails must not leave the others waiting in a collective, and the exit status is non-zero."""
    argv = ["diploid", str(tmp_path), os.path.join(GOLD, "hap1.bam"), os.path.join(GOLD, "hap2.bam"),
            os.path.join(GOLD, "does_not_exist.fa")]
    ok = helpers.run_cli_ranks(argv, 2, timeout=300)
    # a missing reference is reported like the reference does (log + return), on every rank, without hanging
    assert all("[timeout]" not in text for _, text in ok)
    bad = helpers.run_cli_ranks(["diploid", str(tmp_path), os.path.join(GOLD, "hap1.bam"),
                                 os.path.join(GOLD, "missing.bam"), os.path.join(GOLD, "ref.fa")], 2, timeout=300)
    assert all("[timeout]" not in text for _, text in bad) and all(rc != 0 for rc, _ in bad)
