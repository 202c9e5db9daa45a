#!/usr/bin/env python3
"""round 6: svim-asm-cohort on N own copies of the full-size sample, a line per setting (GPU box).
    python tools/r06_cohort_ab.py [--scale 1.0] [--n 8] [--out gpurun_out/r06_cohort_ab.jsonl]
Generates the dataset once; settings: workers, the device's share of the sequence-slice inflate work, the wait for a lane."""
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
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--out", default="gpurun_out/r06_cohort_ab.jsonl")
    ap.add_argument("--settings", default="")
    ap.add_argument("--dataset", default="", help="directory with the sample (made there when it has none); kept.  Without: a fresh one, removed at the end")
    a = ap.parse_args()
    from svim_asm_amd import synth_bam
    from tools import e2e_bench
    d = a.dataset or tempfile.mkdtemp(prefix="svx_cohort_")
    os.makedirs(d, exist_ok=True)
    if os.path.exists(os.path.join(d, "hap1.bam")):
        fasta, bams = os.path.join(d, "ref.fa"), [os.path.join(d, "hap1.bam"), os.path.join(d, "hap2.bam")]
    else:
        fasta, bams = synth_bam.write_dataset(d, **e2e_bench.dataset_args(a.scale))
    meta_name, meta = e2e_bench.reference_meta(a.scale, 8.0, 2000)
    import hashlib

    def check(text):
        return None if meta is None else hashlib.sha256(text.encode()).hexdigest() == meta["vcf_sha256"]
    settings = [  # (workers, share %, wait ms, group[, processes on the device])
        (3, 100, 400, 1), (3, 50, 0, 1), (3, 100, 0, 1), (2, 100, 400, 1), (4, 100, 400, 1), (6, 100, 400, 1), (3, 0, 0, 1),
        (4, 100, 400, 2), (3, 100, 400, 1)]
    if a.settings:
        settings = [tuple(int(x) for x in s.split(":")) for s in a.settings.split(",")]
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        for workers, share, wait, group, *more in settings:
            procs = more[0] if more else 1
            os.environ["SVX_BAM_DEVICE_INFLATE"] = str(share)
            os.environ["SVX_COHORT_INFLATE_WAIT_MS"] = str(wait)
            leg = e2e_bench.run_cohort(a.n, bams, fasta, d, 0, check, workers=workers, group=group, procs_per_device=procs)
            leg.update(device_inflate_percent=share, lane_wait_ms=wait)
            leg.pop("output_tail", None)
            line = json.dumps(leg)
            print(line, flush=True)
            f.write(line + "\n")
    print("DATASET", d)
    if not a.dataset and not os.environ.get("SVX_KEEP_DATASET"):
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
