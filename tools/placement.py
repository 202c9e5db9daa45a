"""Does the streaming kernel's time depend on where its buffers lie?  One process, the headline batch
(config 2 x 256), several placements of the input (a pad allocation of varying size in front of it) and a
fresh context (fresh workspace) for each; per placement the median HIP-event time of k_cigar_tiles over 12
launches, plus the device addresses.
    python3 tools/placement.py
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from svim_asm_amd import _lib  # noqa: E402


def main():
    import torch
    args = argparse.Namespace(config=2, samples=256, distinct=8)
    batch = bench.build_batch(args, 0)
    dev = torch.device("cuda", 0)
    cig_np = batch["cigar"]
    n_ops, n_aln = len(cig_np), len(batch["aln_off"]) - 1
    cap = max(1024, n_ops // 64)
    pads = [0, 1 << 12, 1 << 16, 1 << 20, 3 << 20, 1 << 24, 100 << 20, 1 << 30]
    rows = []
    for rep in range(2):
        for pad in pads:
            keep = torch.empty(pad, dtype=torch.uint8, device=dev) if pad else None
            d_cig = torch.from_numpy(cig_np.view(np.int32)).to(dev)
            d_off = torch.from_numpy(batch["aln_off"].astype(np.int64)).to(dev)
            d_rs = torch.from_numpy(batch["ref_start"]).to(dev)
            outs = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(4)] + [torch.empty(cap, dtype=torch.uint8, device=dev)]
            d_n = torch.zeros(1, dtype=torch.int64, device=dev)
            ctx = _lib.Context(0)

            def call():
                ctx.cigar_extract_dev(d_cig.data_ptr(), n_ops, d_off.data_ptr(), n_aln, d_rs.data_ptr(), 40,
                                      tuple(o.data_ptr() for o in outs), cap, d_n.data_ptr())
            for _ in range(3):
                call()
            ctx.sync()
            tot, dom = bench._event_ms(ctx, call, 12)
            row = {"pad": pad, "rep": rep, "cigar_ptr": hex(d_cig.data_ptr()), "kernel_us": round(dom * 1e3, 1), "path_us": round(tot * 1e3, 1)}
            rows.append(row)
            print(json.dumps(row), flush=True)
            ctx.close()
            del d_cig, d_off, d_rs, outs, d_n, keep
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
