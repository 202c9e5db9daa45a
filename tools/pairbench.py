"""Pair sort + partition timings (HIP events over all launches of one svx_pair_partition_dev_bits call):
one-launch path vs radix path, random and sample-shaped key order.
    python3 tools/pairbench.py [n ...]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import orc  # noqa: E402
from svim_asm_amd import _lib  # noqa: E402


def sample_keys(rng, n, shaped):
    typ = rng.choice(6, n, p=[0.45, 0.45, 0.04, 0.03, 0.02, 0.01]).astype(np.uint64)
    contig = rng.integers(0, 24, n).astype(np.uint64)
    pos = rng.integers(0, 248_000_000, n).astype(np.uint64)
    keys = ((typ << np.uint64(8) | contig) << np.uint64(32)) | pos
    if shaped:  # hap-1 list then hap-2 list, each ordered by (type, contig, pos)
        h = n // 2
        keys = np.concatenate([np.sort(keys[:h]), np.sort(keys[h:])])
    return keys


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [10_000, 30_000, 60_000, 88_000, 131_072]
    ctx = _lib.Context(0)
    out = []
    for n in sizes:
        for shaped in (False, True):
            keys = sample_keys(np.random.default_rng(n), n, shaped)
            bits = int(np.bitwise_or.reduce(keys))
            d_k, d_p, d_id = ctx.dev_array(keys), ctx.dev_array(nbytes=4 * n), ctx.dev_array(nbytes=4 * n)
            d_np = ctx.dev_array(np.zeros(1, np.uint32))
            e = orc.pair_partition(keys, 1000)
            row = {"n": n, "order": "sample" if shaped else "random"}
            for path, limit in (("single_us", 131072), ("radix_us", 0)):
                ctx.set_pair_single_launch_max(limit)

                def call():
                    ctx._check(ctx.lib.svx_pair_partition_dev_bits(ctx.h, d_k.ptr, n, 1000, bits, d_p.ptr, d_id.ptr,
                                                                   d_np.ptr))
                for _ in range(5):
                    call()
                ctx.sync()
                tot_ms, _ = bench._event_ms(ctx, call, 40)
                ok = (np.array_equal(d_p.download(np.uint32), e[0]) and np.array_equal(d_id.download(np.uint32), e[1])
                      and int(d_np.download(np.uint32)[0]) == e[2])
                row[path] = round(tot_ms * 1e3, 2)
                row[path.replace("_us", "_ok")] = bool(ok)
            out.append(row)
            print(json.dumps(row), flush=True)
            for d in (d_k, d_p, d_id, d_np):
                d.free()
    ctx.close()


if __name__ == "__main__":
    main()
