"""Timeline of k_pair_single (build with tools/mkvar.sh pairclk -DSVX_EXP_PAIRCLK, run with
SVX_LIB=build/libsvx_pairclk.so): per phase, median over workgroups and the slowest one, in µs.
    SVX_LIB=build/libsvx_pairclk.so python3 tools/pairclk.py [n] [random|sample]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svim_asm_amd import _lib  # noqa: E402
from tools.pairbench import sample_keys  # noqa: E402

PHASES = ["slice pass", "barrier A", "windows", "shares", "gather", "sort passes", "flags", "barrier B", "table",
          "output"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000
    shaped = len(sys.argv) > 2 and sys.argv[2] == "sample"
    ctx = _lib.Context(0)
    keys = sample_keys(np.random.default_rng(n), n, shaped)
    bits = int(np.bitwise_or.reduce(keys))
    d_k, d_p, d_id = ctx.dev_array(keys), ctx.dev_array(nbytes=4 * n), ctx.dev_array(nbytes=4 * n)
    d_np = ctx.dev_array(np.zeros(1, np.uint32))
    clk = np.zeros(64 * 16, dtype=np.uint64)
    rows = []
    for rep in range(12):
        ctx._check(ctx.lib.svx_pair_partition_dev_bits(ctx.h, d_k.ptr, n, 1000, bits, d_p.ptr, d_id.ptr, d_np.ptr))
        ctx.sync()
        fn = ctx.lib.svx_debug_pair_clk
        fn.argtypes = [C.c_void_p]
        assert fn(clk.ctypes.data) == 0
        if rep >= 4:
            rows.append(clk.reshape(64, 16).astype(np.int64).copy())
    t = np.stack(rows)  # [rep, wg, mark]
    live = t[0, :, 10] > 0
    t = t[:, live, :11]
    t0 = t[:, :, 0].min(axis=1, keepdims=True)
    print("n=%d %s: %d workgroups; first start -> last end %.2f us" % (
        n, "sample" if shaped else "random", live.sum(), np.median((t[:, :, 10].max(axis=1) - t0[:, 0]) / 100.0)))
    print("start skew (last wg start - first): %.2f us" % np.median((t[:, :, 0].max(axis=1) - t0[:, 0]) / 100.0))
    for k, name in enumerate(PHASES):
        d = (t[:, :, k + 1] - t[:, :, k]) / 100.0
        print("%-12s median %.2f  slowest wg %.2f" % (name, np.median(d), np.median(d.max(axis=1))))
        if os.environ.get("PAIRCLK_PER_WG") == name:
            print("   per wg (median over reps):", " ".join("%.1f" % v for v in np.median(d, axis=0)))
            print("   one rep:", " ".join("%.1f" % v for v in d[0]))


if __name__ == "__main__":
    main()
