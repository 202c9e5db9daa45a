#!/usr/bin/env python3
"""bench.py — CIGAR ops/s of the SV-signature hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (a1+a2: CIGAR walk → indel signatures)
over one batch of synthetic haplotype-vs-reference alignments that is already resident in
HBM.  The default workload is BASELINE config 2 (haploid, ~3.1 Gbp, ~5 k alignments,
~1.5 M CIGAR ops per sample) batched as a cohort of `--samples` samples per GPU so that one
step streams more than the 256 MiB Infinity Cache; `--samples 1` gives the single-sample
latency.  Multi-GPU: one process per GPU (torch.distributed / RCCL only for the barrier and
the max-over-ranks reduction; the data path has no collective — alignments shard by sample
and contig), weak scaling: every rank processes its own cohort.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and
`cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=256, help="haplotype samples (assemblies) per GPU per step")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic samples (replicated to --samples)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 5), help="BASELINE config: 2 (mean M run 4000) or 5 (400)")
    ap.add_argument("--min-sv-size", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--layout", default="packed", choices=("packed", "soa"))
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to smoke-test the "
                         "multi-rank logic on a box with fewer GPUs than ranks, together with --share-device)")
    ap.add_argument("--share-device", action="store_true", help="testing only: every rank uses device 0")
    ap.add_argument("--pipeline", action="store_true",
                    help="alternate two contexts between consecutive steps (independent batches overlap; "
                         "per-kernel durations then overlap too, so the default keeps one context)")
    return ap.parse_args()


def build_batch(args, rank):
    from svim_asm_amd import synth
    mean_m = 4000 if args.config == 2 else 400
    distinct = max(1, min(args.distinct, args.samples))
    base = [synth.synth_cigar_batch(seed=1000 * (rank + 1) + args.config * 100 + i, mean_m=mean_m)
            for i in range(distinct)]
    reps = [base[i % distinct] for i in range(args.samples)]
    return synth.concat_batches(reps)


def cpu_baseline(batch, args):
    """C restatement of the reference loop (oracle, kind=port) on 1 core, bounded sample."""
    from oracle import orc
    cig, off, rs = batch["cigar"], batch["aln_off"], batch["ref_start"]
    # bound the sample to ~cpu_seconds: time one pass, then repeat
    max_ops = min(len(cig), 200_000_000)
    a_hi = int(np.searchsorted(off, max_ops, side="right")) - 1
    a_hi = max(a_hi, 1)
    n_ops = int(off[a_hi])
    sub_c = np.ascontiguousarray(cig[:n_ops], np.uint32)
    sub_o = np.ascontiguousarray(off[:a_hi + 1], np.uint64)
    sub_r = np.ascontiguousarray(rs[:a_hi], np.int32)
    bufs = orc.cigar_extract(sub_c, sub_o, sub_r, args.min_sv_size)  # sizes the output buffers (untimed)
    t0 = time.perf_counter()
    orc.cigar_extract_timed(sub_c, sub_o, sub_r, args.min_sv_size, bufs)
    one = time.perf_counter() - t0
    reps = int(max(1, min(200, args.cpu_seconds / max(one, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.cigar_extract_timed(sub_c, sub_o, sub_r, args.min_sv_size, bufs)
    dt = time.perf_counter() - t0
    out = {"value": n_ops * reps / dt, "unit": "CIGAR ops/s", "cores": 1, "kind": "port",
           "sample": "%d passes over %d ops / %d alignments of the bench batch (C restatement of "
                     "SVIM_intra.analyze_cigar_indel, gcc -O2, 1 thread, %.1f s)" % (reps, n_ops, a_hi, dt)}
    # the same C loop on every host core (alignments are independent: one contiguous shard per thread;
    # ctypes releases the GIL) — informative, the reference itself is single-threaded
    try:
        import concurrent.futures as cf
        nthr = max(1, os.cpu_count() or 1)
        cuts = [int(x) for x in np.linspace(0, a_hi, nthr + 1)]
        shards = []
        for i in range(nthr):
            lo, hi = cuts[i], cuts[i + 1]
            if hi > lo:
                o = np.ascontiguousarray(sub_o[lo:hi + 1] - sub_o[lo], np.uint64)
                c = np.ascontiguousarray(sub_c[int(sub_o[lo]):int(sub_o[hi])])
                r = np.ascontiguousarray(sub_r[lo:hi])
                shards.append((c, o, r, orc.cigar_extract(c, o, r, args.min_sv_size)))
        reps_mt = max(1, min(reps, 8))
        with cf.ThreadPoolExecutor(len(shards)) as ex:
            t0 = time.perf_counter()
            for _ in range(reps_mt):
                list(ex.map(lambda sh: orc.cigar_extract_timed(sh[0], sh[1], sh[2], args.min_sv_size, sh[3]), shards))
            dtm = time.perf_counter() - t0
        out["all_cores_value"] = n_ops * reps_mt / dtm
        out["all_cores"] = len(shards)
    except Exception as e:
        out["all_cores_error"] = repr(e)
    # CPython restatement of the same loop (what the reference actually executes), small sample
    try:
        from oracle import svim_oracle
        m = min(n_ops, 1_500_000)
        a2 = max(1, int(np.searchsorted(off, m, side="right")) - 1)
        tuples = [[(int(w) & 15, int(w) >> 4) for w in cig[int(off[a]):int(off[a + 1])]] for a in range(a2)]
        t0 = time.perf_counter()
        for tp in tuples:
            svim_oracle.analyze_cigar_indel(tp, args.min_sv_size)
        dtp = time.perf_counter() - t0
        out["cpython_value"] = int(off[a2]) / dtp
        out["cpython_sample"] = "%d ops, pure-Python restatement, tuples pre-materialised" % int(off[a2])
    except Exception as e:  # the C figure above is the baseline; this one is informative only
        out["cpython_error"] = repr(e)
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the two scalars are reduced
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            except TypeError:  # older signature without device_id
                dist.init_process_group("nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from svim_asm_amd import _lib
    batch = build_batch(args, rank)
    n_ops = int(batch["aln_off"][-1])
    n_aln = len(batch["aln_off"]) - 1

    # --pipeline: two contexts (own HIP stream + own HBM workspace each) alternate between
    # consecutive steps; batches are independent, so step i+1 overlaps the tail of step i.  Default:
    # one context, kernels of consecutive steps run back to back.
    ctxs = [_lib.Context(local_rank), _lib.Context(local_rank)] if args.pipeline else [_lib.Context(local_rank)]
    ctx = ctxs[0]

    cig_np = batch["cigar"]
    d_off = torch.from_numpy(batch["aln_off"].astype(np.int64)).to(dev)
    d_rs = torch.from_numpy(batch["ref_start"]).to(dev)
    if args.layout == "packed":
        d_cig = torch.from_numpy(cig_np.view(np.int32)).to(dev)
        d_op = None
    else:
        d_cig = torch.from_numpy((cig_np >> 4).astype(np.uint32).view(np.int32)).to(dev)
        d_op = torch.from_numpy((cig_np & 15).astype(np.uint8)).to(dev)
    cap = max(1024, n_ops // 16)
    out_sets = []
    for _ in range(len(ctxs)):
        o = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(4)] + \
            [torch.empty(cap, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)]
        out_sets.append(o)
    o_aln, o_ref, o_read, o_len, o_type, d_n = out_sets[0]
    outs = tuple(t.data_ptr() for t in out_sets[0][:5])
    torch.cuda.synchronize(dev)

    # a3 inputs resident in HBM: 5 % of the alignments are primaries of chimeric reads with 1-3
    # SA-derived segments; rows as SVIM_inter.py:66-81 builds them
    rng = np.random.default_rng(77 + rank)
    n_reads = max(1, n_aln // 20)
    k = rng.integers(2, 5, size=n_reads)
    read_off_np = np.concatenate(([0], np.cumsum(k))).astype(np.uint32)
    n_segs = int(read_off_np[-1])
    segs_np = np.zeros(n_segs, dtype=_lib.SEG_DTYPE)
    qs = rng.integers(0, 200000, size=n_segs)
    segs_np["q_start"] = qs
    segs_np["q_end"] = qs + rng.integers(500, 50000, size=n_segs)
    segs_np["ref_id"] = rng.integers(0, 24, size=n_segs)
    segs_np["ref_start"] = rng.integers(0, 50_000_000, size=n_segs)
    segs_np["ref_end"] = segs_np["ref_start"] + rng.integers(500, 50000, size=n_segs)
    segs_np["is_reverse"] = rng.random(n_segs) < 0.1
    d_segs = torch.from_numpy(segs_np.view(np.int32).reshape(-1, 6).copy()).to(dev)
    d_read_off = torch.from_numpy(read_off_np.view(np.int32)).to(dev)
    d_read_len = torch.from_numpy(rng.integers(100000, 5000000, size=n_reads).astype(np.int32)).to(dev)
    d_raw = torch.empty((n_segs, 8), dtype=torch.int32, device=dev)
    seg_prm = _lib.SegParams(args.min_sv_size, 100000, 50, 50, 50, 50)
    import ctypes as C
    # a3 is independent of a1/a2: it runs on a second context (own HIP stream) and overlaps the
    # CIGAR kernels; torch.cuda.synchronize() in barrier() waits for every stream of the device
    ctx2 = _lib.Context(local_rank)

    step_no = [0]

    def step():
        # a1 + a2: every CIGAR op of the batch, once; consecutive steps alternate contexts
        which = step_no[0] % len(ctxs)
        step_no[0] += 1
        c, o = ctxs[which], out_sets[which]
        c.cigar_extract_dev(d_cig.data_ptr(), n_ops, d_off.data_ptr(), n_aln, d_rs.data_ptr(),
                            args.min_sv_size, tuple(t.data_ptr() for t in o[:5]), cap, o[5].data_ptr(),
                            d_op=None if d_op is None else d_op.data_ptr())
        # a3: split-segment decision tree for the chimeric reads of the batch
        ctx2._check(ctx2.lib.svx_segments_classify_dev(ctx2.h, d_segs.data_ptr(), n_segs, d_read_off.data_ptr(),
                                                       n_reads, d_read_len.data_ptr(), C.byref(seg_prm),
                                                       d_raw.data_ptr()))

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)  # uploads ran on torch's stream; the contexts use their own
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    n_sig = int(d_n.item())
    if len(ctxs) > 1 and args.steps > 1 and int(out_sets[1][5].item()) != n_sig:
        raise SystemExit("the two pipelined contexts disagree on the signature count")
    if n_sig > cap:
        raise SystemExit("output capacity too small: %d > %d" % (n_sig, cap))

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_ops], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_ops = int(tot.item())
    else:
        total_ops = n_ops

    # ---- roofline of the dominant kernel (k_cigar_tiles): HIP events on its launch stream around
    # every launch of the a1+a2 path (context 0; event pairs bracket the kernel itself) ----
    torch.cuda.synchronize(dev)
    ctx.set_timing(True)
    k_ms, p_ms = [], []
    for _ in range(max(5, min(20, args.steps))):
        ctx.cigar_extract_dev(d_cig.data_ptr(), n_ops, d_off.data_ptr(), n_aln, d_rs.data_ptr(),
                              args.min_sv_size, outs, cap, d_n.data_ptr(),
                              d_op=None if d_op is None else d_op.data_ptr())
        ctx.sync()
        tot_ms, dom_ms = ctx.last_kernel_ms()
        k_ms.append(dom_ms)
        p_ms.append(tot_ms)
    ctx.set_timing(False)
    k_avg = float(np.mean(k_ms)) * 1e-3
    p_avg = float(np.mean(p_ms)) * 1e-3
    algo_bytes = 4 * n_ops + 16 * n_aln + 17 * n_sig
    achieved = algo_bytes / k_avg / 1e9

    # measured device-copy ceiling of this box (SURVEY.md §8d asks for both denominators): 1 GiB
    # device-to-device copy, bytes read + written over its duration
    copy_gbs = None
    try:
        src = torch.empty(1 << 28, dtype=torch.int32, device=dev)
        dst = torch.empty_like(src)
        src.fill_(1)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize(dev)
        copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
    except Exception:
        copy_gbs = None

    # correctness spot-check of what the timed loop produced (oracle = checker only)
    if rank == 0:
        from oracle import orc
        a_chk = min(n_aln, 2000)
        exp = orc.cigar_extract(cig_np[:int(batch["aln_off"][a_chk])], batch["aln_off"][:a_chk + 1],
                                batch["ref_start"][:a_chk], args.min_sv_size)
        k = len(exp["aln"])
        ok = (np.array_equal(o_ref[:k].cpu().numpy().view(np.uint32), exp["ref_pos"]) and
              np.array_equal(o_read[:k].cpu().numpy().view(np.uint32), exp["read_pos"]) and
              np.array_equal(o_aln[:k].cpu().numpy().view(np.uint32), exp["aln"]) and
              np.array_equal(o_type[:k].cpu().numpy(), exp["type"]))
        if not ok and not os.environ.get("SVX_BENCH_NOCHECK"):  # (ablation builds of tools/ skip the check)
            raise SystemExit("bench output differs from the oracle on the checked prefix")

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        wl_key = "config%d_x%d_%s" % (args.config, args.samples, args.layout)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(wl_key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "CIGAR ops/sec (SV-signature extraction, human genome-genome alignment)",
            "value": total_ops * args.steps / elapsed,
            "unit": "CIGAR ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE config %d (%s, GRCh38-sized, ~5k alignments/sample) x %d samples per GPU "
                            "resident in HBM, %s layout" % (args.config, "haploid 3 Gbp" if args.config == 2 else
                                                            "10x indel density", args.samples, args.layout),
                "ops_per_step_per_gpu": n_ops, "alignments_per_step_per_gpu": n_aln,
                "signatures_per_step_per_gpu": n_sig, "chimeric_reads_per_step_per_gpu": n_reads,
                "segments_per_step_per_gpu": n_segs, "min_sv_size": args.min_sv_size,
                "step": "a1+a2 svx_cigar_extract_dev (stream 1%s) + a3 svx_segments_classify_dev (own stream)"
                        % ("; two contexts alternate between steps" if args.pipeline else ""),
                "parallelism": "sample/contig shards x%d, no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_cigar_tiles", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": k_avg * 1e3,
                "path_ms": p_avg * 1e3, "path_achieved": algo_bytes / p_avg / 1e9,
                "copy_ceiling": copy_gbs, "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
            },
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            res["cpu_baseline"] = cpu_baseline(batch, args)
            res["speedup_vs_cpu_port"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
