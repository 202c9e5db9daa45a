#!/usr/bin/env python3
"""The reader's upload of its CIGAR pool beside the walk (svx_bam_device_pool), measured on a diploid sample's two BAMs:

    python tools/upload_probe.py --dataset DIR [--repeat 5]        (DIR: ref.fa / hap1.bam / hap2.bam of e2e_bench --keep)
    python tools/upload_probe.py --scale 1.0                        (writes the sample first)

Per BAM: `upload_tail_us` = how long svx_bam_device_pool_wait blocks when it is called the moment svx_bam_load returns —
an upper bound of the time between the end of the walk and the end of the pool's last host-to-device copy.
Per mode: `collect_call_ms` = seconds inside svx_collect_batch of the COLLECT step of both BAMs (median over the
repeats) with the pools taken where the readers put them ("in_hbm") and with the pools uploaded by the call itself
("uploaded": the same run with the readers' copies ignored).  Prints one JSON object."""
import argparse
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--repeat", type=int, default=5)
    args = ap.parse_args()
    from svim_asm_amd import SVIM_COLLECT, _lib, bamio
    from tools import e2e_bench
    d = args.dataset
    if d is None:
        from svim_asm_amd import synth_bam
        d = tempfile.mkdtemp(prefix="svx_upload_", dir="/tmp")
        synth_bam.write_dataset(d, **e2e_bench.dataset_args(args.scale))
    bams = [os.path.join(d, "hap1.bam"), os.path.join(d, "hap2.bam")]
    from svim_asm_amd.SVIM_input_parsing import parse_arguments
    opts = parse_arguments("1.0.3", ["diploid", os.path.join(d, "wd_probe"), bams[0], bams[1], os.path.join(d, "ref.fa")])
    opts.device = 0
    ctx = _lib.default_context(0)
    out = {"dataset": d, "repeat": args.repeat, "upload_tail_us": [], "pool_MB": []}

    # ---- the tail of the upload behind the walk (both BAMs walk at the same time, as COLLECT loads them)
    for _ in range(args.repeat):
        files = [bamio.AlignmentFile(b, device=0) for b in bams]
        tails = [None, None]

        def run(k):
            files[k].load(None)
            pool = files[k].device_pool(wait=True)
            tails[k] = None if pool is None else pool[2]
        th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        out["upload_tail_us"].append(tails)
        out["pool_MB"] = [round(len(f._cigar) * 4 / 1e6, 2) for f in files]
        for f in files:
            f.close()

    # ---- the submission with and without the readers' copies
    real = bamio.AlignmentFile.device_pool
    for mode in ("in_hbm", "uploaded", "in_hbm", "uploaded"):
        bamio.AlignmentFile.device_pool = real if mode == "in_hbm" else (lambda self, wait=False: None)
        calls, submits = [], []
        for _ in range(args.repeat):
            files = [bamio.AlignmentFile(b, device=0) for b in bams]
            t0 = time.perf_counter()
            SVIM_COLLECT.collect_tables(files, opts, ctx)
            calls.append(SVIM_COLLECT.LAST_TIMING["collect_call_s"] * 1e3)
            submits.append(SVIM_COLLECT.LAST_TIMING["submit_s"] * 1e3)
            for f in files:
                f.close()
        out.setdefault("collect_call_ms", {}).setdefault(mode, []).append(round(float(np.median(calls)), 4))
        out.setdefault("submit_ms", {}).setdefault(mode, []).append(round(float(np.median(submits)), 4))
    bamio.AlignmentFile.device_pool = real
    print(json.dumps(out))


if __name__ == "__main__":
    main()
