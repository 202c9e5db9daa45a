#!/usr/bin/env python3
"""round 6: where a member's wave spends its clocks in k_inflate_wparse (GPU box; a library built with -DSVX_WPARSE_STATS:
tools/mkvar.sh wstats -DSVX_WPARSE_STATS, SVX_LIB=build/libsvx_wstats.so).
    python tools/r06_wave_stats.py [--scale 0.25] [--members 7261] [--level N: recompress the members with zlib level N]"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["header_clk", "headers", "windows", "pass1_clk", "later_pass_clk", "passes", "write_clk", "total_clk", "given_up",
         "window_bits", "used_bits", "tokens", "members"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.25)
    ap.add_argument("--members", type=int, default=7261)
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--level", type=int, default=None)
    a = ap.parse_args()
    from svim_asm_amd import _lib, bamio, synth_bam
    from tools import e2e_bench
    d = a.dataset or tempfile.mkdtemp(prefix="svx_infl_")
    bam = os.path.join(d, "hap1.bam")
    if not os.path.exists(bam):
        synth_bam.write_dataset(d, **e2e_bench.dataset_args(a.scale))
    raw = open(bam, "rb").read()
    spans = [sp for sp in bamio._bgzf_block_spans(raw) if sp[2] and sp[1] >= 8192][:a.members]
    payloads = [raw[st:st + ln] for st, ln, *_ in spans]
    if a.level is not None:
        out = []
        for p in payloads:
            c = zlib.compressobj(a.level, zlib.DEFLATED, -15)
            out.append(c.compress(zlib.decompress(p, -15)) + c.flush())
        payloads = out
    datas = [zlib.decompress(p, -15) for p in payloads[:200]]
    isize = [sp[2] for sp in spans]
    crc = [int.from_bytes(raw[st + ln:st + ln + 4], "little") for st, ln, *_ in spans]
    ctx = _lib.Context(0)
    ctx.bgzf_inflate(payloads[:64], isize[:64], crc[:64])
    stats = (ctypes.c_ulonglong * 16)()
    try:
        fn = ctx.lib.svx_debug_wparse_stats  # (only in a -DSVX_WPARSE_STATS build)
        fn(stats, 1)
    except AttributeError:
        fn = None
    status, outs, ms = ctx.bgzf_inflate(payloads, isize, crc, keep_output=True)
    res = {"members": len(payloads), "kernel_ms": ms, "ok": bool((status == 0).all()) and all(outs[i] == datas[i] for i in range(len(datas))),
           "compressed_mb": sum(map(len, payloads)) / 1e6, "level": a.level}
    if fn is not None:
        fn(stats, 0)
        v = dict(zip(NAMES, list(stats)))
        m = max(1, v["members"])
        res["per_member"] = {k: round(x / m, 1) for k, x in v.items()}
        res["share_of_total_clk"] = {k: round(v[k] / max(1, v["total_clk"]), 3) for k in ("header_clk", "pass1_clk", "later_pass_clk", "write_clk")}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
