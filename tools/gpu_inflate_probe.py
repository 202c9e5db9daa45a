#!/usr/bin/env python3
"""Device inflate prototype against the CPU decoders on a real workload (GPU box):

    python tools/gpu_inflate_probe.py [--scale 0.25] [--members 40000]

Writes the synthetic diploid sample at `scale` (1.0 = the full-size sample), takes the first `members` non-empty BGZF
members of hap1.bam (SEQ members mostly: what a run's sequence slices touch), and inflates + CRC-checks them
  * on the device: svx_bgzf_inflate_dev, one lane per member, kernel time by HIP events; upload time beside it,
  * on the CPU: zlib (one thread) and the build's own decoder through the native reader's verify mode,
checks the device's bytes against zlib on a sample of members, and prints one JSON object.  Pre-registered criterion
(VERDICT r03, next 6): the device path has to take the sequence-slice CPU seconds of a full-size run from ~1.3 to
<= ~0.2 with the run's wall time not above today's — i.e. all touched members (~40 k) in well under 50 ms."""
import argparse
import json
import os
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.25)
    ap.add_argument("--members", type=int, default=40000)
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--min-payload", type=int, default=0, help="only members whose compressed payload has at least this many "
                    "bytes (8192: the SEQ members — what sequence slices touch —, not the all-0xFF QUAL members)")
    ap.add_argument("--check-all", action="store_true", help="compare EVERY member's device output with zlib's (not only the first 2000)")
    ap.add_argument("--hap", type=int, default=1, help="1 or 2: hap1.bam or hap2.bam of the sample")
    ap.add_argument("--counts", default="", help="comma-separated member counts: kernel time of the first n selected members each")
    args = ap.parse_args()
    from svim_asm_amd import _lib, bamio, synth_bam
    from tools import e2e_bench
    d = args.dataset or tempfile.mkdtemp(prefix="svx_infl_")
    bam = os.path.join(d, "hap%d.bam" % args.hap)
    if not os.path.exists(bam):
        synth_bam.write_dataset(d, **e2e_bench.dataset_args(args.scale))
    raw = open(bam, "rb").read()
    spans = [sp for sp in bamio._bgzf_block_spans(raw) if sp[2] and sp[1] >= args.min_payload][:args.members]
    payloads = [raw[st:st + ln] for st, ln, *_ in spans]
    isize = [sp[2] for sp in spans]
    t0 = time.perf_counter()
    outs_cpu = [zlib.decompress(p, -15) for p in payloads[:2000]]
    zlib_us = (time.perf_counter() - t0) / len(outs_cpu) * 1e6
    crc = []
    for st, ln, isz, *_ in spans:
        crc.append(int.from_bytes(raw[st + ln:st + ln + 4], "little"))
    ctx = _lib.Context(0)
    ctx.bgzf_inflate(payloads[:64], isize[:64], crc[:64])  # warm-up (code object, workspace)
    t0 = time.perf_counter()
    status, outs, ms = ctx.bgzf_inflate(payloads, isize, crc, keep_output=True)
    call_s = time.perf_counter() - t0
    ok = bool((status == 0).all()) and all(outs[i] == outs_cpu[i] for i in range(len(outs_cpu)))
    checked = len(outs_cpu)
    if args.check_all:
        for i in range(len(outs_cpu), len(payloads)):
            ok = ok and outs[i] == zlib.decompress(payloads[i], -15)
        checked = len(payloads)
    by_count = {}
    for c in [int(x) for x in args.counts.split(",") if x]:
        c = min(c, len(payloads))
        by_count[c] = [round(ctx.bgzf_inflate(payloads[:c], isize[:c], crc[:c], keep_output=False)[2], 2) for _ in range(2)]
    n = len(payloads)
    out_bytes, in_bytes = int(sum(isize)), int(sum(len(p) for p in payloads))
    print(json.dumps({"bam": os.path.basename(bam), "inflate_form": os.environ.get("SVX_INFLATE_KERNEL", "3 (default)"), "members": n, "members_compared_with_zlib": checked, "min_payload": args.min_payload, "kernel_ms_by_member_count": by_count, "compressed_bytes": in_bytes, "inflated_bytes": out_bytes,
                      "device_kernel_ms": ms, "device_inflated_GBps": out_bytes / (ms * 1e-3) / 1e9,
                      "device_us_per_member_amortised": ms * 1e3 / n, "binding_call_s_including_pageable_upload_and_download": call_s,
                      "zlib_us_per_member_one_thread": zlib_us, "zlib_members_per_s_16_threads": 16e6 / zlib_us,
                      "device_members_per_s": n / (ms * 1e-3), "all_status_ok_and_bytes_equal_zlib_on_sample": ok}))


if __name__ == "__main__":
    main()
