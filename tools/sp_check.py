"""Single-pass streaming CIGAR path against the five-launch form on the bench cohort, every output byte.

    python tools/sp_check.py [--samples 256] [--reps 5] [--config 2]

Runs svx_collect_batch_dev on the resident cohort of bench.py (395 M ops at the default) with
svx_ctx_set_cigar_single_pass off, keeps every output (signature SoA, count, segment rows, raw and derived records), then
`reps` times with it on and compares all of it — a race between the tile waves and the scanner workgroup shows up as a
difference here.  The first 2 000 alignments / reads are also checked against the oracle (ResidentCollect.check).
Prints the interleaved HIP-event times of both forms (whole path, dominant kernel) as one JSON line.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def outputs(rc):
    n = rc.n_sig()
    k = min(n, rc.cap)
    got = {"n": np.array([n])}
    for x, key in zip(rc.o[:5], ("aln", "ref_pos", "read_pos", "len", "type")):
        got[key] = x.download(np.uint8 if key == "type" else np.uint32, k).copy()
    n_s, n_r = rc.case["n_segs"], rc.case["n_reads"]
    got["segs"] = rc.d_segs.download(np.int32, 6 * n_s).copy()
    got["raw"] = rc.d_raw.download(np.int32, 8 * n_s).copy()
    got["cnt"] = rc.d_cnt.download(np.uint32, n_r).copy()
    return got


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--min-sv-size", type=int, default=40)
    a = ap.parse_args()
    from svim_asm_amd import _lib
    args = argparse.Namespace(samples=a.samples, distinct=a.distinct, config=a.config, min_sv_size=a.min_sv_size)
    batch = bench.build_batch(args, 0)
    n_ops = int(batch["aln_off"][-1])
    case = bench.chimeric_case(batch, 77)
    ctx = _lib.Context(0)
    rc = bench.ResidentCollect(ctx, batch, case, a.min_sv_size, max(1024, n_ops // 16))

    junk = np.full(n_ops // 64, 0xA5A5A5A5, np.uint32)  # more than the signatures of any bench workload

    def run(sp):
        ctx.set_cigar_single_pass(sp)
        for x in rc.o[:5] + [rc.d_raw, rc.d_segs]:  # nothing of an earlier run survives in the outputs
            n = min(x.nbytes, junk.nbytes)
            ctx._check(ctx.lib.svx_dev_upload(ctx.h, x.ptr, junk.ctypes.data, n))
        ctx.sync()
        rc.step()
        ctx.sync()

    run(False)
    ref = outputs(rc)
    ok_oracle = rc.check(n_aln_checked=2000, n_reads_checked=2000)
    bad = []
    for rep in range(a.reps):
        run(True)
        got = outputs(rc)
        for key in ref:
            if not np.array_equal(ref[key], got[key]):
                bad.append((rep, key, int(np.argmax(ref[key][:len(got[key])] != got[key][:len(ref[key])])) if len(ref[key]) == len(got[key]) else -1))
    ok_oracle_sp = rc.check(n_aln_checked=2000, n_reads_checked=2000)
    run(True)
    stats = ctx.cigar_single_pass_stats()
    import ctypes as C
    raw = np.zeros((stats["tile_waves"] + 2) * 4, np.uint32)
    ctx.lib.svx_debug_sp_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    if ctx.lib.svx_debug_sp_raw(ctx.h, raw.ctypes.data, len(raw)) > 0:
        raw = raw.reshape(-1, 4)
        stats["scanner_phase_us(poll,rest)"] = (raw[-1, :2] / 100.0).tolist()
        t0 = int(raw[-1, 2])
        w = raw[:-2]
        beg = ((w[:, 2].astype(np.int64) - t0) & 0xFFFFFFFF) / 100.0
        end = ((w[:, 3].astype(np.int64) - t0) & 0xFFFFFFFF) / 100.0
        beg = np.where(beg > 1e6, beg - 2**32 / 100.0, beg)
        stats["scanner_end_us"] = ((int(raw[-1, 3]) - t0) & 0xFFFFFFFF) / 100.0
        stats["tile_wave_begin_us(min,med,max)"] = [float(beg.min()), float(np.median(beg)), float(beg.max())]
        stats["tile_wave_end_us(min,med,max)"] = [float(end.min()), float(np.median(end)), float(end.max())]
        # by XCD (workgroup b runs on XCD b % 8; dense index wg -> b = wg + 1 + skipped) and by the order of dispatch
        nw = len(end) // 4
        wg = np.arange(nw)
        b = wg + 1 + wg // 255
        e_wg = end.reshape(-1, 4).max(axis=1)
        stats["wg_end_by_xcd_med"] = [round(float(np.median(e_wg[b % 8 == x])), 1) for x in range(8)]
        stats["wg_end_by_layer_med"] = [round(float(np.median(e_wg[b // 256 == l])), 1) for l in range(6)]
        stats["wave_end_by_wave_in_wg_med"] = [round(float(np.median(end.reshape(-1, 4)[:, k])), 1) for k in range(4)]
        stats["wg_end_percentiles(5,25,50,75,95)"] = [round(float(x), 1) for x in np.percentile(e_wg, [5, 25, 50, 75, 95])]
    # interleaved timing
    t = {True: [], False: []}
    ctx.set_timing(True)
    for i in range(2 * max(5, a.reps)):
        sp = i % 2 == 0
        ctx.set_cigar_single_pass(sp)
        for _ in range(3):
            rc.step()
        ctx.sync()
        t[sp].append(ctx.last_kernel_ms())
    ctx.set_timing(False)
    res = {"n_ops": n_ops, "n_sig": int(ref["n"][0]), "identical": not bad, "differences": bad[:10],
           "oracle_prefix_ok": [bool(ok_oracle), bool(ok_oracle_sp)], "stats": stats,
           "single_pass_us": {"path": [round(x[0] * 1e3, 1) for x in t[True]], "tiles": [round(x[1] * 1e3, 1) for x in t[True]]},
           "five_launch_us": {"path": [round(x[0] * 1e3, 1) for x in t[False]], "tiles": [round(x[1] * 1e3, 1) for x in t[False]]}}
    print(json.dumps(res))
    if bad or not ok_oracle_sp:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
