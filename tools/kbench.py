#!/usr/bin/env python3
"""Run single legs of bench.py's extras (for rocprofv3 and A/B work):
    python3 tools/kbench.py latency|pair|editdist [...]
prints one JSON object per leg."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def segments_alone(n_reads=64580):
    """a3 (svx_segments_classify_dev + postpass-free) alone at cohort scale: in bench.py it runs beside the
    streaming kernel on a second stream, so its duration there includes the contention."""
    import ctypes as C
    import numpy as np
    from svim_asm_amd import _lib
    ctx = _lib.Context(0)
    rng = np.random.default_rng(77)
    k = rng.integers(2, 5, size=n_reads)
    roff = np.concatenate(([0], np.cumsum(k))).astype(np.uint32)
    n_segs = int(roff[-1])
    segs = np.zeros(n_segs, dtype=_lib.SEG_DTYPE)
    qs = rng.integers(0, 200000, size=n_segs)
    segs["q_start"] = qs
    segs["q_end"] = qs + rng.integers(500, 50000, size=n_segs)
    segs["ref_id"] = rng.integers(0, 24, size=n_segs)
    segs["ref_start"] = rng.integers(0, 50_000_000, size=n_segs)
    segs["ref_end"] = segs["ref_start"] + rng.integers(500, 50000, size=n_segs)
    d_segs, d_roff = ctx.dev_array(segs), ctx.dev_array(roff)
    d_rl = ctx.dev_array(rng.integers(100000, 5000000, size=n_reads).astype(np.int32))
    d_raw = ctx.dev_array(nbytes=32 * n_segs)
    prm = _lib.SegParams(40, 100000, 50, 50, 50, 50)

    def call():
        ctx._check(ctx.lib.svx_segments_classify_dev(ctx.h, d_segs.ptr, n_segs, d_roff.ptr, n_reads, d_rl.ptr,
                                                     C.byref(prm), d_raw.ptr))
    for _ in range(5):
        call()
    ctx.sync()
    import bench
    tot, dom = bench._event_ms(ctx, call, 30)
    return {"reads": n_reads, "segments": n_segs, "ms": dom, "bytes": 56 * n_segs, "GB/s": 56 * n_segs / (dom * 1e-3) / 1e9}


def host_pointer_rate():
    """PCIe-inclusive rate of the host-pointer entry (svx_cigar_extract: H2D of 4 B/op + offsets, kernels, D2H of the
    17 B/signature, synchronous): one config-2 sample (what the CLI hands over per BAM) and a 64-sample cohort."""
    import time
    import numpy as np
    from svim_asm_amd import _lib, synth
    ctx = _lib.Context(0)
    out = []
    for samples in (1, 64):
        base = [synth.synth_cigar_batch(seed=1200 + i, mean_m=4000) for i in range(min(samples, 8))]
        b = synth.concat_batches([base[i % len(base)] for i in range(samples)])
        n_ops = int(b["aln_off"][-1])
        for _ in range(3):
            ctx.cigar_extract(b["cigar"], b["aln_off"], b["ref_start"], 40)
        reps = 50 if samples == 1 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            sig = ctx.cigar_extract(b["cigar"], b["aln_off"], b["ref_start"], 40)
        dt = (time.perf_counter() - t0) / reps
        out.append({"samples": samples, "ops": n_ops, "signatures": len(sig["aln"]), "ms_per_call": dt * 1e3,
                    "ops_per_s": n_ops / dt, "host_bytes_per_call": 4 * n_ops + 12 * (len(b["aln_off"]) - 1) + 17 * len(sig["aln"])})
    return {"entry": "svx_cigar_extract (host pointers, pageable numpy arrays)", "cases": out}


def main():
    import argparse
    import torch
    import bench
    args = argparse.Namespace(config=2, min_sv_size=40)
    for leg in sys.argv[1:]:
        if leg == "latency":
            print(json.dumps(bench.latency_case(args, 0, torch)))
        elif leg == "pair":
            print(json.dumps(bench.roofline_pair(0)))
        elif leg == "editdist":
            print(json.dumps(bench.roofline_editdist(0, torch.cuda.get_device_properties(0).multi_processor_count)))
        elif leg == "hostptr":
            print(json.dumps(host_pointer_rate()))
        elif leg == "segments":
            print(json.dumps(segments_alone()))
        else:
            raise SystemExit("unknown leg " + leg)


if __name__ == "__main__":
    main()
