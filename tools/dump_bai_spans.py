#!/usr/bin/env python3
"""Compressed bytes per contig of the full-size synthetic diploid sample's two BAMs, as their `.bai` indices attribute them
(svx_bam_contig_spans: what shard.contig_weights feeds the rank plan) -> tests/golden/full_bai_spans.json.

    python tools/dump_bai_spans.py [--scale 1.0] [--dataset DIR] [--out tests/golden/full_bai_spans.json]

The fixture is data: contig names and lengths of the header, the spans of both files, the generator's arguments.  The test
that reads it (tests/test_shard_gloo.py::test_eight_rank_plan_on_the_full_size_spans) needs no BAM."""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "full_bai_spans.json"))
    a = ap.parse_args()
    from svim_asm_amd import bamio, synth_bam
    from tools import e2e_bench
    d = a.dataset or tempfile.mkdtemp(prefix="svx_spans_")
    args = e2e_bench.dataset_args(a.scale)
    if not os.path.exists(os.path.join(d, "hap1.bam")):
        synth_bam.write_dataset(d, **args)
    out = {"generator": "svim_asm_amd.synth_bam.write_dataset(**tools.e2e_bench.dataset_args(%r))" % a.scale,
           "made_by": "tools/dump_bai_spans.py", "files": {}}
    for name in ("hap1.bam", "hap2.bam"):
        f = bamio.AlignmentFile(os.path.join(d, name), device=None)
        spans = f.contig_spans()
        out["references"] = list(f.references)
        out["lengths"] = [int(x) for x in f.lengths]
        out["files"][name] = {"bytes": os.path.getsize(os.path.join(d, name)), "contig_spans": [int(x) for x in spans]}
        f.close()
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out)[:400])


if __name__ == "__main__":
    main()
