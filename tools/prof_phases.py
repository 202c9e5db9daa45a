#!/usr/bin/env python3
"""Per-wave phase clocks of k_cigar_tiles (library built with -DSVX_EXP_PROF; perf experiment only).

    SVX_LIB=$PWD/svim_asm_amd/libsvx_prof.so python tools/prof_phases.py [--samples 64]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--min-len", type=int, default=40)
    a = ap.parse_args()
    args = argparse.Namespace(samples=a.samples, distinct=8, config=2)
    from svim_asm_amd import _lib
    batch = bench.build_batch(args, 0)
    dev = torch.device("cuda", 0)
    n_ops = int(batch["aln_off"][-1]); n_aln = len(batch["aln_off"]) - 1
    d_off = torch.from_numpy(batch["aln_off"].astype(np.int64)).to(dev)
    d_rs = torch.from_numpy(batch["ref_start"]).to(dev)
    d_cig = torch.from_numpy(batch["cigar"].view(np.int32)).to(dev)
    cap = max(1024, n_ops // 16)
    o = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(4)] + \
        [torch.empty(cap, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)]
    ctx = _lib.Context(0)
    torch.cuda.synchronize()
    n_tiles = min(32768, (n_ops + 4095) // 4096)
    buf = np.zeros((n_tiles, 8), dtype=np.uint32)
    f = ctx.lib.svx_debug_prof
    f.argtypes = [C.c_void_p, C.c_uint32]

    def run():
        ctx.cigar_extract_dev(d_cig.data_ptr(), n_ops, d_off.data_ptr(), n_aln, d_rs.data_ptr(), a.min_len,
                              tuple(t.data_ptr() for t in o[:5]), cap, o[5].data_ptr())
    for _ in range(3):
        run()
    ctx.sync()
    assert f(buf.ctypes.data, n_tiles) == 0
    names = ["prologue", "wait+transpose", "walk", "scan+flush", "whole tile"]
    print("tiles", n_tiles)
    for i, n in enumerate(names):
        print("%-16s mean %8.0f  p50 %8.0f  p95 %8.0f clk/wave" % (n, buf[:, i].mean(), np.median(buf[:, i]), np.percentile(buf[:, i], 95)))
    rt = buf[:, 5].astype(np.float64)
    print("realtime (100 MHz) ticks per tile: mean %.0f -> shader clock = %.3f GHz" % (rt.mean(), buf[:, 4].mean() / rt.mean() / 10.0))
    b = buf[:, 6].astype(np.int64)
    b = (b - b.min()) & 0xFFFFFFFF
    e = b + buf[:, 5].astype(np.int64)
    span = (e.max() - b.min()) / 100.0
    ev = np.concatenate([np.stack([b, np.ones_like(b)], 1), np.stack([e, -np.ones_like(e)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    live = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0])
    mean_live = (live[:-1] * dt).sum() / max(dt.sum(), 1)
    print("first begin .. last end: %.1f us; tiles in flight: mean %.0f, max %d (%.1f per CU)" % (span, mean_live, live.max(), live.max() / 256.0))
    # in-flight profile over time (10 slices)
    edges = np.linspace(ev[0, 0], ev[-1, 0], 11)
    idx = np.searchsorted(ev[:, 0], edges[1:-1])
    print("in flight at 10%%..90%% of the span:", [int(live[i]) for i in idx])


if __name__ == "__main__":
    main()
