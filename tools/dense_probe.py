#!/usr/bin/env python3
"""a1+a2 on SV-dense batches (GPU box): what do tiles that leave the staged path cost?

    python tools/dense_probe.py [--reps 30]            # SVX_LIB=build/libsvx_<variant>.so for a variant

Workloads (resident in HBM, svx_cigar_extract_dev, HIP events of the call and of its dominant kernel):
  product   two haplotypes at the density of the full-size synthetic diploid sample (its small contigs carry
            one signature per ~23 ops: 7 % of the 4096-op tiles hold > 128 signatures, 12 % of the 1024-op
            rounds > 32) — the CLI's own submission, streaming path
  sparse    the same two haplotypes at config-2 density (no dense tile): the baseline beside it
  knot      `sparse` with ONE satellite-like stretch (4096 all-indel ops) in the middle
  satellite 1 M ops, every second op an SV-sized indel
Each is checked against the C oracle.  Under `rocprofv3 --kernel-trace --stats` the per-kernel rows
(k_cigar_dense, k_cigar_finish, ...) are the figures profiles/r04_dense_* quote.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workloads():
    from svim_asm_amd import synth
    hap = lambda seed, frac: synth.synth_cigar_batch(seed=seed, mean_m=2000, sv_frac=frac)  # noqa: E731
    sparse = synth.concat_batches([hap(41, 0.015), hap(42, 0.015)])
    # product: most of the genome sparse, a fifth of it at one signature per ~23 ops
    mixed = []
    for seed in (41, 42):
        mixed.append(synth.synth_cigar_batch(seed=seed, mean_m=2000, sv_frac=0.015, ops_target=2_500_000))
        mixed.append(synth.synth_cigar_batch(seed=seed + 10, mean_m=2000, sv_frac=0.09, ops_target=650_000))
    product = synth.concat_batches(mixed)
    knot = {k: v.copy() if isinstance(v, np.ndarray) else v for k, v in sparse.items()}
    mid = len(knot["cigar"]) // 2 // 4096 * 4096 + 100
    rng = np.random.default_rng(3)
    knot["cigar"][mid:mid + 4096] = ((rng.integers(40, 300, 4096) << 4) | rng.integers(1, 3, 4096)).astype(np.uint32)
    sat = synth.synth_cigar_batch(seed=43, mean_m=200, sv_frac=1.0, ops_target=1_000_000)
    return [("sparse", sparse), ("product", product), ("knot", knot), ("satellite", sat)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--min-len", type=int, default=40)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    from svim_asm_amd import _lib
    from oracle import orc
    ctx = _lib.Context(0)
    out = []
    for name, b in workloads():
        if args.only and name not in args.only.split(","):
            continue
        cig, off, rs = b["cigar"], b["aln_off"], b["ref_start"]
        n_ops, n_aln = int(off[-1]), len(off) - 1
        em = (((cig & 15) - 1) < 2) & ((cig >> 4) >= args.min_len)
        exp = orc.cigar_extract(cig, off, rs, args.min_len)
        n_sig = len(exp["aln"])
        d_c, d_o, d_r = ctx.dev_array(cig), ctx.dev_array(off.astype(np.uint64)), ctx.dev_array(rs)
        cap = n_sig + 16
        o = [ctx.dev_array(nbytes=4 * cap) for _ in range(4)] + [ctx.dev_array(nbytes=cap), ctx.dev_array(np.zeros(1, np.uint64))]
        outs = tuple(x.ptr for x in o[:5])
        for path, small in (("streaming", 0), ("small", 1 << 23)):
            if small and n_ops > small:  # beyond the two-launch path
                continue
            ctx.set_small_batch_ops(small)
            tile = 1024 if small else 4096
            cnt = np.add.reduceat(em.astype(np.int64), np.arange(0, n_ops, tile))
            rnd = np.add.reduceat(em.astype(np.int64), np.arange(0, n_ops, 1024))
            call = lambda: ctx.cigar_extract_dev(d_c.ptr, n_ops, d_o.ptr, n_aln, d_r.ptr, args.min_len, outs, cap, o[5].ptr)  # noqa: E731
            for _ in range(3):
                call()
            ctx.sync()
            ctx.set_timing(True)
            tot, dom = [], []
            for _ in range(args.reps):
                call()
                ctx.sync()
                t, d = ctx.last_kernel_ms()
                tot.append(t)
                dom.append(d)
            ctx.set_timing(False)
            ok = int(o[5].download(np.uint64)[0]) == n_sig and all(
                np.array_equal(x.download(np.uint8 if k == "type" else np.uint32, n_sig), exp[k])
                for x, k in zip(o[:5], ("aln", "ref_pos", "read_pos", "len", "type")))
            algo = 4 * n_ops + 16 * n_aln + 17 * n_sig
            out.append({"workload": name, "path": path, "ops": n_ops, "alignments": n_aln, "signatures": n_sig,
                        "tiles": len(cnt), "tiles_over_slab_128": int((cnt > 128).sum()),
                        "rounds_over_32": int((rnd > 32).sum()), "rounds_over_64": int((rnd > 64).sum()),
                        "path_us": float(np.median(tot)) * 1e3, "dominant_us": float(np.median(dom)) * 1e3,
                        "tail_us": float(np.median(np.array(tot) - np.array(dom))) * 1e3,
                        "path_frac_of_8TBs": algo / (float(np.median(tot)) * 1e-3) / 8e12, "bit_exact_vs_oracle": bool(ok)})
            print(json.dumps(out[-1]), flush=True)
        for x in [d_c, d_o, d_r] + o:
            x.free()
    ctx.set_small_batch_ops(1 << 23)
    if not all(r["bit_exact_vs_oracle"] for r in out):
        raise SystemExit("dense_probe: output differs from the oracle")


if __name__ == "__main__":
    main()
