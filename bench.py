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
    ap.add_argument("--no-steady", action="store_true", help="skip the 200 further steps behind the timed region "
                    "(`sustained` in the line)")
    ap.add_argument("--prewarm-ms", type=float, default=0.0, help="diagnostic: keep the device busy with a torch fill loop "
                    "for this long right before the warm-up steps (is the slow start of the timed region the device's clocks?)")
    ap.add_argument("--step-trace", type=int, default=0, help="diagnostic: after the timed region leave the device idle "
                    "for two seconds, then run this many steps one by one (synchronised) and print their times to stderr")
    ap.add_argument("--samples", type=int, default=256, help="haplotype samples (assemblies) per GPU per step")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic samples (replicated to --samples)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 5), help="BASELINE config: 2 (mean M run 4000) or 5 (400)")
    ap.add_argument("--min-sv-size", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--layout", default="packed", choices=("packed", "soa"))
    ap.add_argument("--step", default="collect", choices=("collect", "split"),
                    help="collect (default): one step = svx_collect_batch_dev on the cohort, the product's own submission on one "
                         "stream; split: the round 1-3 step, a1+a2 beside a bare decision tree over random rows on a second stream")
    ap.add_argument("--chain-deal", default="table", choices=("table", "equal"),
                    help="how the chimeric reads of the step's submission go to the chain's workgroups: by the table of "
                         "svx_chain_deal (equal op counts; what svx_collect_batch submits) or in equal read counts (A/B)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to smoke-test the "
                         "multi-rank logic on a box with fewer GPUs than ranks, together with --share-device)")
    ap.add_argument("--share-device", action="store_true", help="testing only: every rank uses device 0")
    ap.add_argument("--group-at-1", action="store_true",
                    help="testing only: bring the process group up (rendezvous, communicator, probe all-reduce, the reductions "
                         "and the gather of the rank records) even with ONE rank — the only way to run the nccl = RCCL path of "
                         "this script on a 1-GPU box, where RCCL refuses two ranks on one device")
    ap.add_argument("--init-timeout", type=int, default=180, help="seconds the process group may take to come up (rendezvous + "
                    "first collective) before a rank gives up with exit status 3")
    ap.add_argument("--settle-ms", type=float, default=300.0, help="host idle time in front of the warm-up steps (lets the "
                    "cgroup's CPU quota period roll over: see the comment at the timed region); 0 = none")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip latency_case / roofline_pair / roofline_editdist (N=1) and e2e / e2e_sharded (any N) — "
                         "rank 0 only, all outside the timed region of `value`")
    ap.add_argument("--e2e-scale", type=float, default=1.0,
                    help="BAM->VCF wall-clock legs (e2e, e2e_sharded): fraction of the GRCh38 contig lengths of the "
                         "synthetic diploid sample written as real BAM+FASTA files (1.0 = 3.1 Gbp, the configuration the "
                         "metric is quoted on: ~1 min of generation on the GPU box); 0 skips the legs")
    ap.add_argument("--a3-after-dominant", type=int, default=0,
                    help="1: the a3 launch of a step waits (svx_ctx_wait_dominant) for the step's streaming kernel and "
                         "overlaps the scan / finish tail; 0: it starts at once, beside the streaming kernel")
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


def device_identity(torch, dev):
    """What identifies the physical device a rank ran on: index inside the process, uuid / PCI address where torch
    exposes them, and the variables that narrowed the process's view."""
    p = torch.cuda.get_device_properties(dev)
    out = {"device_index": dev.index, "device_name": p.name}
    for k in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"):
        v = getattr(p, k, None)
        if v is not None:
            out["device_" + k] = str(v) if k == "uuid" else int(v)
    out["visible_devices_env"] = {k: os.environ[k] for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")
                                  if k in os.environ}
    return out


def gather_rank_records(dist, world, record, grouped=None):
    """Every rank's record, in rank order, on every rank (one all_gather_object over the default group)."""
    if not (world > 1 if grouped is None else grouped):
        return [record]
    out = [None] * world
    dist.all_gather_object(out, record)
    return out


RANK_RECORD_KEYS = ("rank", "local_rank", "device_index", "ms_per_step", "sustained_ms", "ops_per_step", "host", "pid")


def multi_rank_fields(records, backend, probe_sum, world, grouped=None):
    """The part of the line that lets a reader verify a --gpus N run without trusting the launcher: which backend, that
    the probe all-reduce really summed over N ranks, and per rank where it ran and how long ITS steps took (`value` uses
    the maximum).  Raises when a record is missing or incomplete — a line is never printed with holes."""
    if len(records) != world or any(r is None for r in records):
        raise RuntimeError("rank records of %d ranks expected, got %r" % (world, records))
    for i, r in enumerate(records):
        missing = [k for k in RANK_RECORD_KEYS if k not in r]
        if missing or r["rank"] != i:
            raise RuntimeError("rank record %d incomplete or out of order: missing %s, rank %r" % (i, missing, r.get("rank")))
    devices = [(r["host"], r.get("device_uuid") or (r.get("device_pci_domain_id"), r.get("device_pci_bus_id"), r.get("device_pci_device_id"), r["device_index"]))
               for r in records]
    return {"backend": backend if (world > 1 if grouped is None else grouped) else None, "collective_world_verified": int(probe_sum),
            "ranks": records, "distinct_devices": len(set(devices))}


def _throttled_us():
    """Microseconds the cgroup has throttled this process's group so far (cpu.stat), None where that is not exposed."""
    for path, key, scale in (("/sys/fs/cgroup/cpu.stat", "throttled_usec", 1.0),
                             ("/sys/fs/cgroup/cpu/cpu.stat", "throttled_time", 1e-3)):
        try:
            with open(path) as fh:
                for line in fh:
                    parts = line.split()
                    if len(parts) == 2 and parts[0] == key:
                        return float(parts[1]) * scale
        except OSError:
            pass
    return None


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
        from tools.e2e_bench import cpu_quota
        out["cpu_quota_cpus"] = cpu_quota()  # not None: the cgroup grants that many CPUs per period, however many
        # threads run — "all_cores_value" is then the rate of that many cores (16 on the GPU pool: 15x the one-core rate)
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


def _event_ms(ctx, fn, reps):
    """HIP-event durations (total, dominant) of `reps` calls of fn() on ctx's stream, one at a time."""
    ctx.set_timing(True)
    tot, dom = [], []
    for _ in range(reps):
        fn()
        ctx.sync()
        t, d = ctx.last_kernel_ms()
        tot.append(t)
        dom.append(d)
    ctx.set_timing(False)
    return float(np.median(tot)), float(np.median(dom))


def chimeric_case(b, seed, frac=0.05):
    """The chimeric reads of a batch as COLLECT hands them to svx_collect_batch: `frac` of the alignments (5 %,
    SURVEY.md §8d config 2) are primaries with 1-3 SA-derived segments, whose CIGARs (S M S, as SVIM_COLLECT.py:33-55
    rebuilds them from the SA tag) follow the records' CIGARs; segment rows, read lengths and reference ends are
    computed on the device from those CIGARs (k_segment_rows / the fused chain), not handed in."""
    rng = np.random.default_rng(seed)
    n_aln = len(b["aln_off"]) - 1
    n_reads = max(1, int(n_aln * frac))
    prim = np.sort(rng.choice(n_aln, size=n_reads, replace=False)).astype(np.int64)
    k = rng.integers(1, 4, size=n_reads)
    n_extra = int(k.sum())
    read_off = np.concatenate(([0], np.cumsum(1 + k))).astype(np.uint32)
    n_segs = int(read_off[-1])
    first = read_off[:-1].astype(np.int64)
    is_first = np.zeros(n_segs, bool)
    is_first[first] = True
    seg_src = np.empty(n_segs, np.uint32)
    seg_src[first] = prim
    seg_src[~is_first] = n_aln + np.arange(n_extra)
    seg_tid = rng.integers(0, 24, size=n_segs).astype(np.int32)
    seg_pos = rng.integers(0, 50_000_000, size=n_segs).astype(np.int32)
    seg_pos[first] = b["ref_start"][prim]
    seg_rev = (rng.random(n_segs) < 0.1).astype(np.uint8)
    seg_qend = np.full(n_segs, -1, np.int32)
    extra_cigar = np.empty(3 * n_extra, np.uint32)
    extra_cigar[0::3] = (rng.integers(1, 200000, size=n_extra).astype(np.uint32) << 4) | 4
    extra_cigar[1::3] = (rng.integers(500, 50000, size=n_extra).astype(np.uint32) << 4) | 0
    extra_cigar[2::3] = (rng.integers(1, 200000, size=n_extra).astype(np.uint32) << 4) | 4
    extra_off = (np.arange(n_extra + 1, dtype=np.uint64) * 3)
    slots = np.diff(read_off.astype(np.int64))
    post_off = np.concatenate(([0], np.cumsum(slots * (slots + 3) // 2))).astype(np.uint64)
    return dict(n_reads=n_reads, n_segs=n_segs, n_extra=n_extra, read_off=read_off, seg_src=seg_src, seg_tid=seg_tid,
                seg_pos=seg_pos, seg_rev=seg_rev, seg_qend=seg_qend, extra_cigar=extra_cigar, extra_off=extra_off,
                post_off=post_off, rank=np.arange(24, dtype=np.int32))


class ResidentCollect:
    """One submission of svx_collect_batch_dev with every input resident in HBM (svx_dev_malloc'ed buffers): what
    svx_collect_batch enqueues between its uploads and its read-backs."""

    def __init__(self, ctx, b, case, min_len, cap=None, use_deal=True):
        from svim_asm_amd import _lib
        self.ctx, self.b, self.case, self.min_len = ctx, b, case, min_len
        n_ops, n_aln = int(b["aln_off"][-1]), len(b["aln_off"]) - 1
        self.n_ops, self.n_aln = n_ops, n_aln
        cigar_all = np.concatenate((b["cigar"], case["extra_cigar"]))
        off_all = np.concatenate((b["aln_off"], n_ops + case["extra_off"][1:])).astype(np.uint64)
        d = self.d = {key: ctx.dev_array(v) for key, v in dict(
            cigar=cigar_all, off=off_all, rs=b["ref_start"], src=case["seg_src"], tid=case["seg_tid"], pos=case["seg_pos"],
            rev=case["seg_rev"], qend=case["seg_qend"], roff=case["read_off"], rank=case["rank"], poff=case["post_off"]).items()}
        self.cap = cap = cap or max(1024, n_ops // 16)
        self.o = [ctx.dev_array(nbytes=4 * cap) for _ in range(4)] + [ctx.dev_array(nbytes=cap), ctx.dev_array(np.zeros(1, np.uint64))]
        self.outs = tuple(x.ptr for x in self.o[:5])
        n_segs, n_reads = case["n_segs"], case["n_reads"]
        self.d_segs, self.d_rl = ctx.dev_array(nbytes=24 * n_segs), ctx.dev_array(nbytes=4 * n_reads)
        self.d_raw = ctx.dev_array(nbytes=32 * n_segs)
        self.d_post, self.d_cnt = ctx.dev_array(nbytes=32 * int(case["post_off"][-1])), ctx.dev_array(nbytes=4 * n_reads)
        self.prm = _lib.SegParams(min_len, 100000, 50, 50, 50, 50)
        # part of the control block, as in svx_collect_batch: the chain's reads dealt to its workgroups by op count
        deal = np.zeros(2 * (n_reads + 2), np.uint32)
        self.n_deal = max(0, int(ctx.lib.svx_chain_deal(case["read_off"].ctypes.data, n_reads, case["seg_src"].ctypes.data,
                                                         off_all.ctypes.data, deal.ctypes.data))) if use_deal else 0
        self.d_deal = ctx.dev_array(deal[:2 * (self.n_deal + 1)]) if self.n_deal else None
        self.dv = _lib.CollectDev(
            d_cigar=d["cigar"].ptr, n_ops=n_ops, d_aln_off=d["off"].ptr, n_aln=n_aln, n_extra=case["n_extra"],
            d_ref_start=d["rs"].ptr, min_len=min_len, d_seg_src=d["src"].ptr, d_seg_tid=d["tid"].ptr, d_seg_pos=d["pos"].ptr,
            d_seg_rev=d["rev"].ptr, d_seg_qend=d["qend"].ptr, n_segs=n_segs, read_off=case["read_off"].ctypes.data,
            d_read_off=d["roff"].ptr, n_reads=n_reads, d_contig_rank=d["rank"].ptr, n_contigs=len(case["rank"]), params=self.prm,
            d_sig=_lib.SigSoa(*self.outs), sig_cap=cap, d_n_sig=self.o[5].ptr, d_segs=self.d_segs.ptr, d_read_len=self.d_rl.ptr,
            d_raw=self.d_raw.ptr, d_post=self.d_post.ptr, post_off=case["post_off"].ctypes.data, d_post_off=d["poff"].ptr,
            d_post_cnt=self.d_cnt.ptr, d_chain_deal=self.d_deal.ptr if self.n_deal else None, n_chain_blocks=self.n_deal)

    def step(self):
        import ctypes as C
        self.ctx._check(self.ctx.lib.svx_collect_batch_dev(self.ctx.h, C.byref(self.dv)))

    def a1a2(self):
        self.ctx.cigar_extract_dev(self.d["cigar"].ptr, self.n_ops, self.d["off"].ptr, self.n_aln, self.d["rs"].ptr, self.min_len,
                                   self.outs, self.cap, self.o[5].ptr)

    def n_sig(self):
        return int(self.o[5].download(np.uint64)[0])

    def check(self, n_aln_checked=None, n_reads_checked=None):
        """Checker only: the signatures of the first alignments against the C oracle; the raw records of the first
        reads against the oracle's CIGAR statistics + decision tree (the rows between them by the host mirror of
        SVIM_inter.py:66-81)."""
        from oracle import orc
        from svim_asm_amd import _lib
        b, case = self.b, self.case
        a_chk = min(self.n_aln, n_aln_checked or self.n_aln)
        exp = orc.cigar_extract(b["cigar"][:int(b["aln_off"][a_chk])], b["aln_off"][:a_chk + 1], b["ref_start"][:a_chk], self.min_len)
        k = len(exp["aln"])
        ok = (a_chk < self.n_aln or self.n_sig() == k) and all(
            np.array_equal(x.download(np.uint8 if key == "type" else np.uint32, k), exp[key])
            for x, key in zip(self.o[:5], ("aln", "ref_pos", "read_pos", "len", "type")))
        r_chk = min(case["n_reads"], n_reads_checked or case["n_reads"])
        n_s = int(case["read_off"][r_chk])
        src = case["seg_src"][:n_s].astype(np.int64)
        cig_x, off_x = case["extra_cigar"], case["extra_off"]
        st = {}
        is_x = src >= self.n_aln
        # statistics of the named alignments only (the primaries' CIGARs and the SA-derived ones)
        pr = src[~is_x]
        parts = [b["cigar"][int(b["aln_off"][a]):int(b["aln_off"][a + 1])] for a in pr.tolist()]
        p_off = np.concatenate(([0], np.cumsum([len(x) for x in parts]))).astype(np.uint64)
        st_p = orc.cigar_stats(np.concatenate(parts) if parts else np.zeros(0, np.uint32), p_off)
        st_x = orc.cigar_stats(cig_x, off_x)
        for key in ("ref_len", "q_start", "q_end", "read_len"):
            v = np.empty(n_s, np.int64)
            v[~is_x] = st_p[key]
            v[is_x] = st_x[key][src[is_x] - self.n_aln]
            st[key] = v
        segs, read_len = _lib.segment_rows(st, case["seg_tid"][:n_s], case["seg_pos"][:n_s], case["seg_rev"][:n_s],
                                           case["seg_qend"][:n_s], case["read_off"][:r_chk + 1])
        raw = orc.segments_classify(segs, case["read_off"][:r_chk + 1], read_len, (self.min_len, 100000, 50, 50, 50, 50))
        ok = ok and np.array_equal(self.d_raw.download(np.int32, 8 * n_s).reshape(-1, 8), raw.view(np.int32).reshape(-1, 8))
        return bool(ok)

    def free(self):
        for x in list(self.d.values()) + self.o + [self.d_segs, self.d_rl, self.d_raw, self.d_post, self.d_cnt] + \
                ([self.d_deal] if self.d_deal else []):
            x.free()


def collect_leg(args, local_rank, torch, batches, what, seed):
    """ONE submission per step through the call path COLLECT uses (svx_collect_batch_dev: a1+a2 and the split-segment
    chain a3, one stream) with the inputs resident in HBM — wall-clock per step —, and the host call
    svx_collect_batch itself (uploads from page-locked memory, the same kernels, two read-backs) beside it."""
    from svim_asm_amd import _lib, synth
    b = batches[0] if len(batches) == 1 else synth.concat_batches(batches)
    case = chimeric_case(b, seed)
    n_ops, n_aln = int(b["aln_off"][-1]), len(b["aln_off"]) - 1
    ctx = _lib.Context(local_rank)
    # ---- the host call (what SVIM_COLLECT issues per sample), PCIe included; the reader's CIGAR pools are page-locked
    pools = []
    for x in batches:
        t = torch.from_numpy(x["cigar"].view(np.int32)).pin_memory()
        pools.append((t, t.numpy().view(np.uint32)))

    def host_call(part_dev=None):
        return ctx.collect_batch([p[1] for p in pools], b["aln_off"], b["ref_start"], args.min_sv_size, case["extra_cigar"],
                                 case["extra_off"], case["seg_src"], case["seg_tid"], case["seg_pos"], case["seg_rev"],
                                 case["seg_qend"], case["read_off"], case["rank"], (args.min_sv_size, 100000, 50, 50, 50, 50),
                                 part_dev=part_dev)
    sig, raw, post, first = host_call()

    def median_ms(call):
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3  # (median: a single call that meets a page fault storm is 10x the others)
    host_ms = median_ms(host_call)
    # the same call when the reader has put its pools into HBM during the walk (svx_bam_device_pool → part_dev): what is
    # left of the submission is the control block's upload, a device-to-device move per pool, kernels and read-backs
    in_hbm = [p[0].cuda(local_rank) for p in pools]
    torch.cuda.synchronize()
    where = [(t.data_ptr(), None) for t in in_hbm]
    again = host_call(where)
    if not (np.array_equal(again[0]["aln"], sig["aln"]) and np.array_equal(again[0]["ref_pos"], sig["ref_pos"])
            and np.array_equal(again[1], raw) and np.array_equal(again[2], post)):
        raise SystemExit("%s: the submission over pools in HBM differs from the one that uploads them" % what)
    host_hbm_ms = median_ms(lambda: host_call(where))
    del in_hbm
    # ---- the same kernel sequence with everything resident: one stream, no host round trip inside a step
    rc = ResidentCollect(ctx, b, case, args.min_sv_size)
    for _ in range(20):
        rc.step()
    ctx.sync()
    steps = 500
    t0 = time.perf_counter()
    for _ in range(steps):
        rc.step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    n_sig = rc.n_sig()
    algo = 4 * n_ops + 16 * n_aln + 17 * n_sig
    path_ms, dom_ms = _event_ms(ctx, rc.a1a2, 20)
    rc.step()
    ctx.sync()
    ok = rc.check() and n_sig == len(sig["aln"]) and np.array_equal(rc.d_raw.download(np.int32).reshape(-1, 8),
                                                                     raw.view(np.int32).reshape(-1, 8))
    cnt = rc.d_cnt.download(np.uint32)
    ok = ok and np.array_equal(np.concatenate(([0], np.cumsum(cnt))), first)
    if not ok:
        raise SystemExit("%s output differs from the oracle" % what)
    small = n_ops <= (1 << 23)
    rc.free()
    ctx.close()
    return {"workload": "%s (%d ops, %d alignments, %d signatures, %d chimeric reads / %d segments)"
                        % (what, n_ops, n_aln, n_sig, case["n_reads"], case["n_segs"]),
            "step": "svx_collect_batch_dev, inputs resident in HBM, one stream: "
                    + ("two launches — k_tiles_a3 (a1+a2 tiles of 1024 ops beside the segment rows and the decision tree of the "
                       "split-segment chain) and k_cigar_finish_small (descriptor scan + final records beside the chain's "
                       "post-passes)" if small else
                       "the five launches of the streaming CIGAR path, the chain inside the finish and dense launches")
                    + " — what svx_collect_batch enqueues between its uploads and its read-backs",
            "launches": 2 if small else 5,
            "ms_per_step": dt * 1e3, "value": n_ops / dt, "unit": "CIGAR ops/s",
            "algorithmic_bytes": algo, "a3_bytes": (24 + 32) * case["n_segs"], "achieved": algo / dt / 1e9,
            "frac": algo / dt / 1e9 / HBM_PEAK_GBS,
            "a1a2_path_ms_hip_events": path_ms, "dominant_kernel_ms": dom_ms,
            "host_call_ms": host_ms, "host_call_note": "svx_collect_batch from page-locked host memory: uploads (%.1f MB), kernels, "
                                                       "two read-backs, two synchronisations — the PCIe-inclusive figure, never `value`"
                                                       % ((4 * n_ops + 8 * n_aln) / 1e6),
            "host_call_pools_in_hbm_ms": host_hbm_ms,
            "host_call_pools_in_hbm_note": "the same call with the CIGAR pools already in HBM (the BAM reader uploads its pool "
                                           "while it assembles it: svx_bam_device_pool, svx_collect_in.part_dev) — what the "
                                           "command line's submission costs between the walk and the results",
            "bit_exact_vs_oracle": True}


def latency_case(args, local_rank, torch):
    """What `svim-asm haploid` submits: ONE config-2 sample."""
    from svim_asm_amd import synth
    b = synth.synth_cigar_batch(seed=1000 + args.config * 100, mean_m=4000 if args.config == 2 else 400)
    return collect_leg(args, local_rank, torch, [b], "BASELINE config %d, ONE sample per submission: what `svim-asm haploid` "
                       "submits per BAM" % args.config, 5)


def product_point(args, local_rank, torch):
    """What `svim-asm diploid` submits: both haplotype BAMs of one sample in ONE batch, at the op count of the
    full-size synthetic diploid sample the BAM -> VCF wall-clock is quoted on (2 x ~3.1 M ops, mean M run 2000) and
    with a fifth of each haplotype at the SV density of that sample's small contigs (one signature per ~23 ops:
    rounds beyond the queue, tiles beyond the slab)."""
    from svim_asm_amd import synth
    haps = []
    for seed in (41, 42):
        haps.append(synth.concat_batches([
            synth.synth_cigar_batch(seed=seed, mean_m=2000, sv_frac=0.015, ops_target=2_500_000),
            synth.synth_cigar_batch(seed=seed + 10, mean_m=2000, sv_frac=0.09, ops_target=650_000)]))
    return collect_leg(args, local_rank, torch, haps, "diploid sample, both haplotype BAMs in ONE submission: what `svim-asm "
                       "diploid` submits", 6)


def _sample_keys(rng, n, presorted):
    """Keys as _pack_keys builds them for a human sample: (type, contig) << 32 | position.  presorted: the
    order PAIR hands over — the haplotype-1 list, then the haplotype-2 list, each grouped by type and ordered
    along the genome (svim-asm:133-148); otherwise the same keys in random order."""
    typ = rng.choice(6, n, p=[0.45, 0.45, 0.04, 0.03, 0.02, 0.01]).astype(np.uint64)
    keys = ((typ << np.uint64(8) | rng.integers(0, 24, n).astype(np.uint64)) << np.uint64(32)) | \
        rng.integers(0, 250_000_000, n).astype(np.uint64)
    if presorted:
        keys = np.concatenate([np.sort(keys[:n // 2]), np.sort(keys[n // 2:])])
    return keys


def roofline_pair(local_rank):
    """a5+a6 (svx_pair_partition_dev_bits): 20 B per candidate over the HIP-event time of
    all its launches: 60 k candidates (one diploid human sample) in the order PAIR hands them over and in
    random order — one launch, k_pair_single — and 600 k (radix path, P + 2 launches)."""
    from svim_asm_amd import _lib
    from oracle import orc
    ctx = _lib.Context(local_rank)
    out = []
    for n, presorted in ((60_000, True), (60_000, False), (600_000, False)):
        keys = _sample_keys(np.random.default_rng(n), n, presorted)
        bits = int(np.bitwise_or.reduce(keys))
        d_k, d_p, d_id = ctx.dev_array(keys), ctx.dev_array(nbytes=4 * n), ctx.dev_array(nbytes=4 * n)
        d_np = ctx.dev_array(np.zeros(1, np.uint32))

        def call():
            ctx._check(ctx.lib.svx_pair_partition_dev_bits(ctx.h, d_k.ptr, n, 1000, bits, d_p.ptr, d_id.ptr, d_np.ptr))
        for _ in range(5):
            call()
        ctx.sync()
        tot_ms, sort_ms = _event_ms(ctx, call, 30)
        e_perm, e_part, e_n = orc.pair_partition(keys, 1000)
        if not (np.array_equal(d_p.download(np.uint32), e_perm) and np.array_equal(d_id.download(np.uint32), e_part)
                and int(d_np.download(np.uint32)[0]) == e_n) and not os.environ.get("SVX_BENCH_NOCHECK"):
            raise SystemExit("roofline_pair output differs from the oracle")
        algo = 20 * n
        out.append({"candidates": n, "key_order": "haplotype lists (as PAIR hands them over)" if presorted else "random",
                    "launches": 1 if n <= 131072 else "P + 2", "ms": tot_ms, "sort_launches_ms": sort_ms,
                    "algorithmic_bytes": algo, "achieved": algo / (tot_ms * 1e-3) / 1e9,
                    "frac": algo / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        for d in (d_k, d_p, d_id, d_np):
            d.free()
    ctx.close()
    return {"bound": "hbm", "kernel": "k_pair_single (<= 131072 candidates) | k_pair_init + k_radix_pass x P + k_partition",
            "unit": "GB/s", "peak": HBM_PEAK_GBS, "bytes_per_candidate": 20, "cases": out, "bit_exact_vs_oracle": True,
            "note": "ms = HIP events around all launches of one call (about 10 us of event and launch overhead "
                    "included; the kernel's own duration is in profiles/); latency-bound at this size, not byte-bound"}


# VALU instructions per systolic step of k_edit_myers<16> (gfx950 ISA of the inner loops, `hipcc -S`: 38 in the steady
# two-step loop of single-strip patterns — 32 for the recurrence, which the compiler fuses below its 36 with v_bfi /
# v_or3, and 6 for the symbol, the match vector and the two delta bits —, about 40 in a pattern's last strip, about 46 in
# the strips that hand their bottom row on; 42 / 50 until the end of round 4, 75 until round 3): a step updates 64 lanes
# x 64 rows of the DP matrix.  The constant is the steady single-strip loop, which the PAIR-like batch mostly runs; the
# other two loops make the formulation's ceiling a little lower than quoted for batches that live in them
EDIT_VALU_PER_STEP = 38
# the recurrence alone (Myers 1999 / Hyyro 2003, one 64-row block and one column): 18 operations on 64-bit words,
# two 32-bit VALU each — the floor any bit-vector formulation on this ISA pays per 64 x 64 lane-cells
EDIT_ALGO_VALU_PER_STEP = 36


def _edit_cells(la, lb, k):
    """DP cells the kernel visits for one pair (strip of 4096 rows x the columns inside Ukkonen's band)."""
    m, n = max(la, lb), min(la, lb)
    if n == 0:
        return 0
    strips = (m + 4095) // 4096
    if strips == 1:
        return n * ((m + 63) // 64) * 64
    cells = 0
    for s in range(strips):
        row_lo, row_hi = s * 4096 + 1, min(m, (s + 1) * 4096)
        c_lo, c_hi = max(1, row_lo - k), min(n, row_hi + k)
        if c_hi >= c_lo:
            cells += (c_hi - c_lo + 1) * ((row_hi - row_lo + 64) // 64) * 64
    return cells


def roofline_editdist(local_rank, n_cu):
    """a7 (svx_edit_distance_batch, Myers/Hyyro bit-vector kernel): DP cell updates per second against the VALU
    issue ceiling of the formulation; two workloads: the PAIR step's typical haplotype pairs (threshold 200) and
    contig-scale alleles (exact)."""
    from svim_asm_amd import _lib
    from oracle import orc
    ctx = _lib.Context(local_rank)
    rng = np.random.default_rng(11)
    cases = []
    for name, n_pairs, lo, hi, k in (("PAIR-like: 20000 haplotype pairs, 240-10200 bases, ~1 % divergence, threshold 200",
                                      20000, 240, 10200, 200),
                                     ("long alleles: 64 pairs of 100 kb, exact", 64, 100_000, 100_001, 0xFFFFFFFF)):
        la = np.exp(rng.uniform(np.log(lo), np.log(hi), n_pairs)).astype(np.int64)
        seqs, a_off, a_len, b_off, b_len, at = [], [], [], [], [], 0
        for L in la.tolist():
            a = rng.integers(0, 4, L).astype(np.uint8)
            b = a.copy()
            nm = max(1, L // 100)
            b[rng.integers(0, L, nm)] = rng.integers(0, 4, nm)
            cut = int(rng.integers(0, 20))
            b = b[cut:]
            a = np.frombuffer(b"ACGT", np.uint8)[a]
            b = np.frombuffer(b"ACGT", np.uint8)[b]
            seqs += [a, b]
            a_off.append(at); a_len.append(len(a)); at += len(a)
            b_off.append(at); b_len.append(len(b)); at += len(b)
        pool = np.concatenate(seqs)
        a_off, b_off = np.array(a_off, np.uint64), np.array(b_off, np.uint64)
        a_len, b_len = np.array(a_len, np.uint32), np.array(b_len, np.uint32)
        res = [None]

        def call():
            res[0] = ctx.edit_distance_batch(pool, a_off, a_len, b_off, b_len, k)
        # the product's plan: wavefront pass (O(n + d^2)) first, bit-vector kernel for what it leaves
        ctx.set_edit_wavefront_cap(1024)
        call()
        plan_ms, _ = _event_ms(ctx, call, 5)
        plan_res = res[0].copy()
        # the bit-vector kernel alone: the figure the VALU ceiling below applies to
        ctx.set_edit_wavefront_cap(0)
        call()
        tot_ms, _ = _event_ms(ctx, call, 5)
        if not np.array_equal(plan_res, res[0]):
            raise SystemExit("roofline_editdist: the two-stage plan and the bit-vector kernel disagree")
        # checker: the C oracle on a bounded subset
        idx = rng.choice(n_pairs, size=min(n_pairs, 200 if hi < 50000 else 4), replace=False)
        for i in idx.tolist():
            a = pool[int(a_off[i]):int(a_off[i]) + int(a_len[i])].tobytes()
            b = pool[int(b_off[i]):int(b_off[i]) + int(b_len[i])].tobytes()
            e = orc.edit_distance_banded(a, b)
            g = int(res[0][i])
            if not (g == e or (k != 0xFFFFFFFF and e > k and g > k)):
                raise SystemExit("roofline_editdist: pair %d: %d != %d" % (i, g, e))
        band = 200 if k != 0xFFFFFFFF else 256
        cells = sum(_edit_cells(int(x), int(y), band) for x, y in zip(a_len, b_len))
        cases.append({"workload": name, "pairs": n_pairs, "ms": tot_ms, "cells": cells,
                      "achieved": cells / (tot_ms * 1e-3) / 1e9, "two_stage_plan_ms": plan_ms,
                      "two_stage_pairs_per_s": n_pairs / (plan_ms * 1e-3)})
    peak = n_cu * 4 * 2.4e9 / (EDIT_VALU_PER_STEP * 4) * 4096 / 1e9
    peak_algo = n_cu * 4 * 2.4e9 / (EDIT_ALGO_VALU_PER_STEP * 4) * 4096 / 1e9
    for c in cases:
        c["frac"] = c["achieved"] / peak_algo          # against the recurrence itself: the honest distance
        c["frac_of_formulation_ceiling"] = c["achieved"] / peak
    ctx.close()
    return {"bound": "valu", "kernel": "k_edit_myers<16>", "unit": "G cell updates/s", "peak": peak_algo,
            "peak_basis": "algorithm-level: %d CUs x 4 SIMDs x 2.4 GHz / (%d VALU x 4 clk per 64x64-cell step); %d = the "
                          "Myers/Hyyro block recurrence alone, 18 64-bit word operations per block and column "
                          "(Eq|Mv; &Pv, +Pv, ^Pv, |X; Pv&D0; Pv|D0, ~, Mv|; two carry extractions; two shifts with their "
                          "carry-ins; D0|Hp, ~, Hn|; Hp&D0; the score update) = 36 32-bit VALU — no symbol fetch, no "
                          "hand-off between blocks, no band bookkeeping" % (n_cu, EDIT_ALGO_VALU_PER_STEP, EDIT_ALGO_VALU_PER_STEP),
            "formulation_peak": peak,
            "formulation_peak_basis": "%d VALU per step as this kernel is written (gfx950 ISA of the inner loop): the recurrence "
                                      "plus the systolic hand-off (DPP wave_shr of the horizontal delta and the text symbol), "
                                      "the match-vector fetch from LDS and the band / strip bookkeeping" % EDIT_VALU_PER_STEP,
            "cases": cases, "exact_vs_oracle_on_sample": True,
            "note": "ms / cells / achieved: the bit-vector kernel alone (svx_ctx_set_edit_wavefront_cap(0)); "
                    "two_stage_plan_ms: the default plan, whose wavefront pass (k_edit_wfa, O(n + d^2)) resolves nearly "
                    "identical haplotypes without visiting the DP cells, so cells/s does not apply to it.  "
                    "ms = HIP events around the launches of one svx_edit_distance_batch call (host-pointer entry: "
                    "includes its small per-launch control copies and the result read-back; the exact case counts "
                    "the cells of the first band only, retries with wider bands are extra work inside the same time)"}


def e2e_leg(scale, local_rank, n_devices=1):
    """BAM -> VCF wall-clock of the product on a synthetic diploid sample (BASELINE config 3; scale 1.0 = GRCh38
    contig lengths, the configuration the metric is quoted on) written as real BAM + FASTA files: the pipeline
    in this process with a phase split, `svim-asm diploid` as a fresh process, and the contig-sharded command
    line as R rank processes (BASELINE config 4).  The VCF is compared with the digest of the VCF the REAL
    reference wrote for the same inputs (tests/golden/full_inputs.json), the inputs are identified by digests
    of their uncompressed content."""
    from tools import e2e_bench
    ranks = sorted(set([1, 2, 4] + ([n_devices] if n_devices > 1 else [])))
    samples = sorted(set([2, 4] + ([n_devices] if n_devices > 1 else [])))
    r = e2e_bench.run_e2e(scale=scale, repeat=5, device=local_rank, ranks=ranks, n_devices=n_devices, samples=samples, cohort=(8, 16))
    best, med = r.get("best_run", r), r.get("median_run", r)
    runs = r.get("all_runs_total_s") or [r["product_total_s"]]
    return {"workload": "svim-asm diploid, BASELINE config 3 at %.3g x GRCh38 contig lengths (%d bp, 2 BAMs of %d / %d bytes, "
                        "%d + %d CIGAR ops)" % (scale, r["genome_bp"], r["bam_bytes"][0], r["bam_bytes"][1],
                                                r["cigar_ops"][0], r["cigar_ops"][1]),
            "wall_s": med["product_total_s"],  # the MEDIAN of the runs below, every one of them with the default ingest:
            # every touched BGZF member inflated whole and its CRC32 checked, as htslib does under the reference
            "wall_s_first_run": runs[0], "wall_s_best": best["product_total_s"], "all_runs_wall_s": runs,
            # wall_s leaves out the unmapping of the genome FASTA, which the reference does inside write_final_vcf
            # (SVIM_COMBINE.py:466-467) and the product defers (the command: to the process's exit, inside command_line_wall_s;
            # a long-lived caller: to a background thread); timed by itself in the same run:
            "reference_release_s": med.get("reference_release_s"),
            "wall_s_including_reference_release": med.get("product_total_including_reference_release_s"),
            "first_run_is_cold": r.get("page_cache_dropped_before_first_run"),  # posix_fadvise(DONTNEED) on both BAMs, their
            # indices and the FASTA right before the first run: it reads its inputs from storage (True = the kernel took
            # every request; the files were written by this process a minute earlier, pages still dirty then are synced first)
            "phases_s": {k: med[k] for k in ("open_index_s", "collect_s", "pair_s", "vcf_s")},
            "pair_stages_s": med.get("pair_stages_s"), "vcf_stages_s": med.get("vcf_stages_s"),
            "collect_stages_s": med.get("collect_stages_s"),  # load_s: the record walks of both BAMs; sequences_wait_s: what PAIR still
            # waited for the inflate of the inserted alleles, which starts in COLLECT and runs beside PAIR's set-up
            "wall_s_prefix_only_no_crc": r.get("prefix_only_no_crc_total_s"),  # one more pass with svx_bam_set_verify(0)
            # (`--no_bgzf_crc`): members inflated only as far as needed, no CRC32 unless a member is inflated to its end
            "wall_s_host_inflate_only": r.get("host_inflate_only_total_s"),  # median of three more runs with the device leg off
            "all_runs_wall_s_host_inflate_only": r.get("host_inflate_only_runs_total_s"),  # (what a one-shot command runs: cli.py)
            "device_inflate_percent": r.get("device_inflate_percent"),  # the device leg of the sequence slices: share of
            "bgzf_members_inflated_on_device": r.get("bgzf_members_inflated_on_device"),  # each call, members per reader
            "cpu_seconds": med.get("cpu_seconds"),  # CPU seconds of all threads per phase, and beside them
            "cpu_seconds_prefix_only_no_crc": r.get("prefix_only_no_crc_cpu_seconds"),
            "cpu_quota_cpus": r.get("cpu_quota_cpus"),  # the CPUs the box's cgroup grants per period: their quotient bounds wall_s
            "command_line_wall_s": r.get("cli_wall_s"),  # `svim-asm diploid` as a fresh process: interpreter + HIP start-up included
            "command_line_all_wall_s": r.get("cli_all_wall_s"),
            "inputs_match_real_reference_run": r.get("inputs_match_real_reference_run"),
            "vcf_matches_real_reference_digest": r.get("vcf_matches_real_reference_digest"),
            "real_reference_fixture": r.get("real_reference_fixture"),
            "real_reference_wall_s_build_container": r.get("real_reference_wall_s_build_container"),
            "oracle_pipeline_wall_s": r.get("oracle_total_s"), "vcf_identical_to_oracle": r.get("vcf_identical"),
            "vcf_records": r["vcf_records"], "cigar_ops": r["cigar_ops"], "candidates": r["candidates"],
            "ingest_threads": r["ingest_threads"], "index_state": r["index_state"], "generate_s": r["generate_s"],
            "devices": n_devices,
            "note": "outside the timed region of `value`; the pipeline functions called in this process the way cli._run "
                    "calls them (garbage collector off for the run, as the command does); wall_s = median of 5 (the first of them cold)"}, \
           {"workload": "the same sample through the real command line as R fresh rank processes (contigs LPT-packed over "
                        "the ranks, one table exchange after COLLECT and one after PAIR, rank 0 writes the VCF); ranks "
                        "share the %d visible device(s)" % n_devices,
            "runs": r["cli_ranks"],
            "note": "wall_s: first process start to last process exit (R = 1: median of three runs; R > 1: the worse of two), "
                    "interpreter start + imports + HIP initialisation of every rank included"}, \
           {"workload": "N independent `svim-asm diploid` processes at once, each on its own copy of the same sample, process k on "
                        "device k mod %d: the unit that scales across the GPUs of a node is the sample (DESIGN §6)" % n_devices,
            "runs": r.get("samples"),
            "note": "samples_per_s = N / (first start to last exit); on a 1-GPU box the processes share device 0 and the box's "
                    "CPU quota (cpu_quota_cpus), which is what bounds them"}, \
           {"workload": "`svim-asm-cohort diploid` as ONE fresh process (device %d) over N own copies of the sample's BAMs, the genome "
                        "FASTA shared: the headline's own workload — many samples through one device — from the BAMs to the VCFs; a "
                        "stream of groups over worker threads with a device context each (svim_asm_amd/cohort.py)" % local_rank,
            "runs": r.get("cohort"),
            "note": "samples_per_s = N / wall-clock of the process, interpreter start and HIP bring-up included (paid once); "
                    "cpu_seconds_per_sample: all threads; every VCF against the real reference's digest; the multi-GPU unit is "
                    "one such process per device"}


def summary_of(res):
    """The legs a reader of the line's LAST 2 000 bytes needs (the driver keeps only that tail), as one compact object:
    the BAM -> VCF wall-clocks, the product's operating points, the cohort leg.  Kept below 1 500 bytes."""
    def get(d, *keys):
        for k in keys:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d

    def r3(x):
        return None if x is None else round(float(x), 4)
    e2e = res.get("e2e") if isinstance(res.get("e2e"), dict) else {}
    s = {
        "ms_per_step": r3(res.get("ms_per_step")), "path_frac": r3(get(res, "roofline", "path_frac")), "frac": r3(get(res, "roofline", "frac")),
        "e2e_wall_s": r3(e2e.get("wall_s")), "e2e_wall_s_host_inflate_only": r3(e2e.get("wall_s_host_inflate_only")),
        "e2e_cpu_s": r3(get(e2e, "cpu_seconds", "total")), "command_line_wall_s": r3(e2e.get("command_line_wall_s")),
        "e2e_vcf_ok": e2e.get("vcf_matches_real_reference_digest"),
        "product_point": [r3(get(res, "product_point", "ms_per_step")), r3(get(res, "product_point", "frac"))],
        "latency_case": [r3(get(res, "latency_case", "ms_per_step")), r3(get(res, "latency_case", "frac"))],
        # (first case of each leg: the PAIR-like batch of 20 000 pairs; the 60 000 keys PAIR hands over)
        "roofline_editdist": [[r3(c.get("frac")), r3(c.get("ms")), r3(c.get("two_stage_plan_ms"))] for c in (get(res, "roofline_editdist", "cases") or [])
                              if isinstance(c, dict)][:2],
        "roofline_pair_frac": r3(((get(res, "roofline_pair", "cases") or [{}])[0] or {}).get("frac")),
        "cpu_quota_cpus": res.get("cpu_quota_cpus"), "cpu_seconds_per_sample": r3(res.get("cpu_seconds_per_sample")),
    }
    for key, fields in (("e2e_cohort", ("samples", "samples_per_s", "cpu_seconds_per_sample")),
                        ("e2e_samples", ("processes", "samples_per_s")), ("e2e_sharded", ("ranks", "wall_s"))):
        runs = get(res, key, "runs")
        if isinstance(runs, list):
            s[key] = [[r3(r.get(f)) if not isinstance(r.get(f), int) else r.get(f) for f in fields] for r in runs if isinstance(r, dict)]
        elif isinstance(res.get(key), dict) and "error" in res[key]:
            s[key] = "error"
    return s


def relaunch_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run from this
    process, which has not touched the GPU, and exit with its status (never exec after GPU init)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus <= 1 or world == args.gpus:
        return
    if world != 1:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with `python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port 29511 bench.py "
                         "--gpus %d ...`" % (args.gpus, world, args.gpus, args.gpus))
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    relaunch_if_needed(args)
    # ONE JSON line on stdout: libraries under torch.distributed print to the C-level stdout (RCCL its version banner
    # when the process ends — behind the line —, gloo its connection notes), so descriptor 1 is pointed at stderr for
    # the whole run and the line is written to a duplicate of the real stdout
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    probe_sum = 1
    grouped = world > 1 or args.group_at_1
    if args.group_at_1 and world == 1:
        import socket as _s
        sk = _s.socket()
        sk.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        sk.close()
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        import threading

        def _give_up():  # a rendezvous or a first collective that never completes must not hang the driver's run
            sys.stderr.write("bench.py: rank %d: the %s process group did not come up within %d s (MASTER_ADDR=%s MASTER_PORT=%s, "
                             "device %d); giving up\n" % (rank, args.backend, args.init_timeout, os.environ.get("MASTER_ADDR"),
                                                          os.environ.get("MASTER_PORT"), local_rank))
            sys.stderr.flush()
            os._exit(3)
        watchdog = threading.Timer(args.init_timeout, _give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            if args.backend == "nccl":
                try:
                    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                            timeout=datetime.timedelta(seconds=args.init_timeout))
                except TypeError:  # older signature without device_id
                    dist.init_process_group("nccl", rank=rank, world_size=world,
                                            timeout=datetime.timedelta(seconds=args.init_timeout))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.init_timeout))
            probe = torch.ones(1, dtype=torch.float64, device=red_dev)  # the first collective builds the communicator
            dist.all_reduce(probe)
            if args.backend == "nccl":
                torch.cuda.synchronize(dev)
            probe_sum = int(probe.item())
            if probe_sum != world:
                raise RuntimeError("all_reduce over %d ranks returned %s" % (world, probe.item()))
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench.py: rank %d: cannot initialise the %s process group: %s: %s\n" % (rank, args.backend, type(e).__name__, e))
            sys.stderr.flush()
            os._exit(3)
        finally:
            watchdog.cancel()

    from svim_asm_amd import _lib
    batch = build_batch(args, rank)
    n_ops = int(batch["aln_off"][-1])
    n_aln = len(batch["aln_off"]) - 1
    if args.layout == "soa" and args.step == "collect":
        raise SystemExit("bench.py: --layout soa needs --step split (svx_collect_batch_dev takes the BAM-native packed words)")

    # --pipeline: two contexts (own HIP stream + own HBM workspace each) alternate between
    # consecutive steps; batches are independent, so step i+1 overlaps the tail of step i.  Default:
    # one context, kernels of consecutive steps run back to back.
    ctxs = [_lib.Context(local_rank), _lib.Context(local_rank)] if args.pipeline else [_lib.Context(local_rank)]
    ctx = ctxs[0]
    cig_np = batch["cigar"]
    cap = max(1024, n_ops // 16)
    import ctypes as C

    if args.step == "collect":
        # ---- the product's own submission: svx_collect_batch_dev on the cohort, chimeric reads with REAL segment
        # CIGARs (SA-derived S M S alignments behind the records' CIGARs): a1+a2 (five launches of the streaming
        # path) and the split-segment chain (rows -> decision tree -> post-passes, one launch) on ONE stream
        case = chimeric_case(batch, 77 + rank)
        n_reads, n_segs = case["n_reads"], case["n_segs"]
        rcs = [ResidentCollect(c, batch, case, args.min_sv_size, cap, use_deal=args.chain_deal == "table") for c in ctxs]
        rc0 = rcs[0]
        d_cig_ptr, d_off_ptr, d_rs_ptr, d_op_ptr = rc0.d["cigar"].ptr, rc0.d["off"].ptr, rc0.d["rs"].ptr, None
        outs, d_n_ptr = rc0.outs, rc0.o[5].ptr
        ctx2 = None
        step_no = [0]

        def step():
            which = step_no[0] % len(rcs)
            step_no[0] += 1
            rcs[which].step()
    else:
        # ---- labelled variant (round 1-3 headline): a1+a2 on one context, a bare svx_segments_classify_dev over
        # RANDOM segment rows on a second context (own stream) beside it — no segment rows from CIGARs, no post-passes
        d_off = torch.from_numpy(batch["aln_off"].astype(np.int64)).to(dev)
        d_rs = torch.from_numpy(batch["ref_start"]).to(dev)
        if args.layout == "packed":
            d_cig = torch.from_numpy(cig_np.view(np.int32)).to(dev)
            d_op = None
        else:
            d_cig = torch.from_numpy((cig_np >> 4).astype(np.uint32).view(np.int32)).to(dev)
            d_op = torch.from_numpy((cig_np & 15).astype(np.uint8)).to(dev)
        out_sets = []
        for _ in range(len(ctxs)):
            o = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(4)] + \
                [torch.empty(cap, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)]
            out_sets.append(o)
        outs = tuple(t.data_ptr() for t in out_sets[0][:5])
        d_n_ptr = out_sets[0][5].data_ptr()
        d_cig_ptr, d_off_ptr, d_rs_ptr = d_cig.data_ptr(), d_off.data_ptr(), d_rs.data_ptr()
        d_op_ptr = None if d_op is None else d_op.data_ptr()
        torch.cuda.synchronize(dev)
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
        ctx2 = _lib.Context(local_rank)
        step_no = [0]

        def step():
            which = step_no[0] % len(ctxs)
            step_no[0] += 1
            c, o = ctxs[which], out_sets[which]
            c.cigar_extract_dev(d_cig_ptr, n_ops, d_off_ptr, n_aln, d_rs_ptr, args.min_sv_size,
                                tuple(t.data_ptr() for t in o[:5]), cap, o[5].data_ptr(), d_op=d_op_ptr)
            if args.a3_after_dominant:
                ctx2.wait_dominant(c)
            ctx2._check(ctx2.lib.svx_segments_classify_dev(ctx2.h, d_segs.data_ptr(), n_segs, d_read_off.data_ptr(),
                                                           n_reads, d_read_len.data_ptr(), C.byref(seg_prm),
                                                           d_raw.data_ptr()))

    def barrier():
        torch.cuda.synchronize(dev)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)  # uploads ran on torch's stream; the contexts use their own
    if args.prewarm_ms > 0:
        junk = torch.empty(1 << 28, dtype=torch.int32, device=dev)
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            junk.fill_(1)
            torch.cuda.synchronize(dev)
        del junk
    # The box grants this process a CPU quota per 100-ms period (16 CPUs on the pool's boxes).  A process that has just
    # burnt it — building and uploading the batch is multi-threaded numpy — is descheduled, every thread of it, until the
    # period ends: once in a dozen runs that stall fell INTO the 7-ms timed region and doubled its figure (0.875 ms per
    # step with a kernel of 0.259 ms and a sustained step of 0.328 ms, profiles/README.md).  So the host sits still for a
    # moment before the warm-up steps, and the line says how long the cgroup throttled it inside the timed region.
    import gc
    gc.collect()
    gc.disable()  # (no collector pause inside the 7 ms either; back on behind the sustained steps)
    if args.settle_ms > 0:
        time.sleep(args.settle_ms / 1e3)
    for _ in range(args.warmup):
        step()
    barrier()
    thr0 = _throttled_us()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    thr1 = _throttled_us()
    host_throttled_ms = None if thr0 is None or thr1 is None else (thr1 - thr0) / 1e3
    own_elapsed = elapsed
    # `value` is the region above — the FIRST `steps` steps behind the warm-up, as the contract asks.  Four more regions of
    # the same length, bracketed the same way, give the spread of a 7-ms measurement (`value_median_of_5`).
    repeats = [elapsed]
    for _ in range(0 if args.no_steady else 4):
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        repeats.append(time.perf_counter() - t1)
    # Sustained rate, reported beside `value` (never instead of it): the timed region above is `steps` steps after
    # `warmup` warm-up steps, as asked; with the driver's 5 + 20 steps that is 9 ms on a device that was idle while the
    # host prepared the batch, and its clocks are still on their way up (the same 20 steps after 50 warm-up steps take
    # 0.31 ms each instead of 0.35).  Here: 200 further steps, bracketed the same way.
    steady = None
    if not args.no_steady:
        barrier()
        t1 = time.perf_counter()
        for _ in range(200):
            step()
        barrier()
        steady_elapsed = time.perf_counter() - t1
        own_steady_elapsed = steady_elapsed
        if grouped:
            t = torch.tensor([steady_elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            steady_elapsed = float(t.item())
        own_steady_ms = own_steady_elapsed / 200 * 1e3
        steady = {"steps": 200, "ms_per_step": steady_elapsed / 200 * 1e3}
    else:
        own_steady_ms = None
    gc.enable()
    if args.step_trace and rank == 0:
        time.sleep(2.0)
        trace = []
        for _ in range(args.step_trace):
            t1 = time.perf_counter()
            step()
            torch.cuda.synchronize(dev)
            trace.append((time.perf_counter() - t1) * 1e3)
        print("step trace after 2 s idle (ms, each step synchronised):", " ".join("%.3f" % x for x in trace), file=sys.stderr)
    ctx.sync()
    n_sig_buf = np.zeros(1, np.uint64)
    ctx._check(ctx.lib.svx_dev_download(ctx.h, n_sig_buf.ctypes.data, d_n_ptr, 8))
    n_sig = int(n_sig_buf[0])
    if args.step == "collect" and len(rcs) > 1 and args.steps > 1 and rcs[1].n_sig() != n_sig:
        raise SystemExit("the two pipelined contexts disagree on the signature count")
    if n_sig > cap:
        raise SystemExit("output capacity too small: %d > %d" % (n_sig, cap))

    if grouped:
        t = torch.tensor(repeats, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        repeats = [float(x) for x in t.tolist()]
        elapsed = repeats[0]
        tot = torch.tensor([n_ops], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_ops = int(tot.item())
    else:
        total_ops = n_ops
    # where every rank ran and how long ITS steps took (all ranks take part; rank 0 prints)
    import socket
    rank_record = dict(device_identity(torch, dev), rank=rank, local_rank=int(os.environ.get("LOCAL_RANK", "0")),
                       ms_per_step=own_elapsed / args.steps * 1e3, sustained_ms=own_steady_ms, ops_per_step=n_ops,
                       host=socket.gethostname(), pid=os.getpid(), shares_device_0=bool(args.share_device))
    rank_fields = multi_rank_fields(gather_rank_records(dist, world, rank_record, grouped), args.backend, probe_sum, world, grouped)

    # ---- roofline of the dominant kernel: HIP events on its launch stream around every launch of the step's own
    # sequence (context 0; the event pair brackets the kernel itself).  The dominant kernel of the step is the streaming
    # tile launch k_cigar_tiles<4096> — with --step collect the split-segment chain rides in the finish and dense
    # launches behind it, so the chain's bytes count for the PATH figure, not for this kernel ----
    torch.cuda.synchronize(dev)
    op_bytes = 5 if args.layout == "soa" else 4  # packed u32 per op; SoA: a u8 op code next to the u32 length
    algo_bytes = op_bytes * n_ops + 16 * n_aln + 17 * n_sig  # SURVEY.md §8(d), whole a1+a2 path
    # what the tile workgroups themselves move of those: the op stream, aln_off, the slab records (16 B per
    # signature) and one 16-B descriptor + one 4-B start index per tile; the 17-B final records are written by
    # the finish kernel and only count for the path figure
    n_tiles = (n_ops + 4095) // 4096
    tile_bytes = op_bytes * n_ops + 8 * n_aln + 16 * n_sig + 20 * n_tiles

    def timed(fn):
        ctx.set_timing(True)
        k_ms, p_ms = [], []
        for _ in range(max(5, min(20, args.steps))):
            fn()
            ctx.sync()
            tot_ms, dom_ms = ctx.last_kernel_ms()
            k_ms.append(dom_ms)
            p_ms.append(tot_ms)
        ctx.set_timing(False)
        return float(np.mean(k_ms)) * 1e-3, float(np.mean(p_ms)) * 1e-3

    def a1a2_only():
        ctx.cigar_extract_dev(d_cig_ptr, n_ops, d_off_ptr, n_aln, d_rs_ptr, args.min_sv_size, outs, cap, d_n_ptr, d_op=d_op_ptr)
    tiles_k, tiles_p = timed(a1a2_only)
    chain_bytes = 0
    if args.step == "collect":
        # bytes of the chain's stage A inside the same launch: the CIGARs of the segments' alignments (4 B per op), per
        # segment its table row in (seg_src, two offsets, tid, pos, rev, qend: 33 B), the 24-B row and the read length
        # out and in again, the 32-B raw record out; per read two offsets and its length
        seg_ops = int((batch["aln_off"][1:] - batch["aln_off"][:-1])[case["seg_src"][case["seg_src"] < n_aln].astype(np.int64)].sum()) + \
            int(len(case["extra_cigar"]))
        chain_bytes = 4 * seg_ops + (33 + 2 * 28 + 32) * n_segs + 12 * n_reads
        k_avg, p_avg = timed(rc0.step)
        kernel_bytes = tile_bytes
    else:
        k_avg, p_avg, kernel_bytes = tiles_k, tiles_p, tile_bytes
    achieved = kernel_bytes / k_avg / 1e9

    # second denominator: the read-only nontemporal stream this box sustains over the very buffer
    # the kernel reads, measured in this run (svx_hbm_read_probe_dev) — <= the spec peak by construction
    try:
        read_gbs = ctx.hbm_read_probe(d_cig_ptr, (op_bytes - (1 if args.layout == "soa" else 0)) * n_ops // 16 * 16, 5)
    except Exception:
        read_gbs = None

    # correctness spot-check of what the timed loop produced (oracle = checker only)
    if rank == 0 and not os.environ.get("SVX_BENCH_NOCHECK"):  # (ablation builds of tools/ skip the check)
        if args.step == "collect":
            rc0.step()
            ctx.sync()
            ok = rc0.check(n_aln_checked=2000, n_reads_checked=2000)
        else:
            from oracle import orc
            a_chk = min(n_aln, 2000)
            exp = orc.cigar_extract(cig_np[:int(batch["aln_off"][a_chk])], batch["aln_off"][:a_chk + 1],
                                    batch["ref_start"][:a_chk], args.min_sv_size)
            k = len(exp["aln"])
            o_aln, o_ref, o_read, o_len, o_type = out_sets[0][:5]
            ok = (np.array_equal(o_ref[:k].cpu().numpy().view(np.uint32), exp["ref_pos"]) and
                  np.array_equal(o_read[:k].cpu().numpy().view(np.uint32), exp["read_pos"]) and
                  np.array_equal(o_aln[:k].cpu().numpy().view(np.uint32), exp["aln"]) and
                  np.array_equal(o_len[:k].cpu().numpy().view(np.uint32), exp["len"]) and
                  np.array_equal(o_type[:k].cpu().numpy(), exp["type"]))
        if not ok:
            raise SystemExit("bench output differs from the oracle on the checked prefix")

    if rank == 0:
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        wl_key = "config%d_x%d_%s" % (args.config, args.samples, args.layout)
        if os.path.exists(tpath):
            try:
                entry = json.load(open(tpath)).get(wl_key, {})
                traffic = entry.get("hbm_bytes_per_launch")
                # the committed counters belong to ONE workload: the bytes this run's launch counts must be the ones the
                # counter passes' launch counted (same ops, alignments, signatures, tiles), else no figure at all
                counted = entry.get("counted_bytes_per_launch")
                if traffic is not None and counted is not None and abs(kernel_bytes - counted) > 0.002 * counted:
                    traffic = None
                if traffic is not None:
                    # not a property of THIS run: PMC counters need their own rocprofv3 passes (separate --pmc runs,
                    # MI355X_MICROARCH.md); the figure is the committed one of the same workload and kernel
                    traffic_source = "%s (%s; collected by the builder with rocprofv3 --pmc, not in this run)" % (
                        entry.get("source"), entry.get("method", "").split(";")[0])
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
            "value_median_of_5": (float(np.median([total_ops * args.steps / e for e in repeats])) if len(repeats) == 5 else None),
            "all_regions_ms_per_step": [e / args.steps * 1e3 for e in repeats],  # the first is `value`'s
            "host_throttled_ms_in_timed_region": host_throttled_ms,  # (cgroup cpu.stat; rank 0's group; None: not exposed)
            "sustained": None if steady is None else dict(steady, value=total_ops * 200 / (steady["ms_per_step"] * 200 * 1e-3),
                                                          note="200 further steps behind the timed region, bracketed the same "
                                                               "way (barrier + synchronize, max over ranks): the rate of a device "
                                                               "whose clocks have come up; `value` is the timed region as asked"),
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
                "step": ("svx_collect_batch_dev on ONE stream (the product's submission: the five launches of the streaming CIGAR "
                         "path with the split-segment chain inside the last two — segment rows from the SA-derived CIGARs and the "
                         "decision tree in the finish launch, the post-passes in the dense-tile launch)%s" if args.step == "collect" else
                         "VARIANT --step split: a1+a2 svx_cigar_extract_dev (stream 1%s) + a bare svx_segments_classify_dev over "
                         "random segment rows (own stream); not what the product submits")
                        % ("; two contexts alternate between steps" if args.pipeline else ""),
                "parallelism": "sample/contig shards x%d, no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_cigar_tiles<false, 4096, 0>",
                "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel_bytes_per_launch": kernel_bytes, "algorithmic_bytes_per_launch": algo_bytes + chain_bytes,
                "chain_bytes_in_the_path": chain_bytes,  # the split-segment chain's stage A (in the finish launch): the segments'
                # CIGARs read again + rows / raw records / table entries
                "kernel_ms": k_avg * 1e3,
                "path_ms": p_avg * 1e3, "path_achieved": (algo_bytes + chain_bytes) / p_avg / 1e9,
                "path_frac": (algo_bytes + chain_bytes) / p_avg / 1e9 / HBM_PEAK_GBS,
                # ... and the same bytes over the step as `value` times it (launch gaps and the device's ramp included)
                "step_frac": (algo_bytes + chain_bytes) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                "a1a2_only": {"kernel": "k_cigar_tiles<false, 4096, 0>", "note": "svx_cigar_extract_dev alone (no chimeric reads in the "
                              "submission: no chain in the finish and dense launches): the call rounds 1-3 timed", "kernel_ms": tiles_k * 1e3,
                              "kernel_bytes_per_launch": tile_bytes,
                              "achieved": tile_bytes / tiles_k / 1e9, "frac": tile_bytes / tiles_k / 1e9 / HBM_PEAK_GBS,
                              "path_ms": tiles_p * 1e3, "path_frac": algo_bytes / tiles_p / 1e9 / HBM_PEAK_GBS},
                "read_ceiling": read_gbs, "frac_of_read_ceiling": (achieved / read_gbs) if read_gbs else None,
                "read_ceiling_note": "read-only nontemporal stream over the same resident CIGAR buffer, measured in this run "
                                     "(svx_hbm_read_probe_dev)",
            },
        }
        res.update(rank_fields)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            res["cpu_baseline"] = cpu_baseline(batch, args)
            res["speedup_vs_cpu_port"] = res["value"] / res["cpu_baseline"]["value"]
    # ---- legs outside the timed region (rank 0).  Several ranks: the process group is closed first and the other
    # ranks leave — their GPUs are then free for the rank processes of the BAM -> VCF legs
    for c in ctxs + ([ctx2] if ctx2 is not None else []):
        c.sync()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
        # this process goes on alone: it is no longer a rank of anything (the BAM -> VCF legs look at the launcher's
        # variables to decide whether they are a contig-sharded run)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK"):
            os.environ.pop(k, None)
    if rank == 0:
        if not args.no_extras:
            if args.step == "collect":  # the cohort is not needed any more
                for r_ in rcs:
                    r_.free()
            else:
                del d_cig, d_op, out_sets
            torch.cuda.empty_cache()
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            legs = []
            if world == 1:
                legs += [("latency_case", lambda: latency_case(args, local_rank, torch)),
                         ("product_point", lambda: product_point(args, local_rank, torch)),
                         ("roofline_pair", lambda: roofline_pair(local_rank)),
                         ("roofline_editdist", lambda: roofline_editdist(local_rank, n_cu))]
            if args.e2e_scale > 0:
                n_dev = world if not args.share_device else 1
                legs.append((("e2e", "e2e_sharded", "e2e_samples", "e2e_cohort"), lambda: e2e_leg(args.e2e_scale, local_rank, n_dev)))
            for name, leg in legs:
                # a leg that fails (its own oracle check included) is reported in its slot: the headline
                # above has been measured and checked already and must still be printed
                try:
                    got = leg()
                    if isinstance(name, tuple):
                        for k, v in zip(name, got):
                            res[k] = v
                    else:
                        res[name] = got
                except BaseException as e:  # noqa: BLE001 — SystemExit of a failed check included
                    if isinstance(e, KeyboardInterrupt):
                        raise
                    for k in (name if isinstance(name, tuple) else (name,)):
                        res[k] = {"error": "%s: %s" % (type(e).__name__, e)}
        # what a scaling record needs to explain itself: on this pool BAM -> VCF is bound by the box's CPU quota
        coh = res.get("e2e_cohort", {}).get("runs") if isinstance(res.get("e2e_cohort"), dict) else None
        if coh:
            best = max((r for r in coh if isinstance(r, dict) and r.get("samples_per_s")), key=lambda r: r["samples"], default=None)
            if best:
                res["cpu_seconds_per_sample"], res["cpu_quota_cpus"] = best.get("cpu_seconds_per_sample"), best.get("cpu_quota_cpus")
        if "cpu_quota_cpus" not in res:
            from tools import e2e_bench
            res["cpu_quota_cpus"] = e2e_bench.cpu_quota()
            sam = res.get("e2e_samples", {}).get("runs") if isinstance(res.get("e2e_samples"), dict) else None
            if sam and isinstance(sam[-1], dict) and sam[-1].get("cpu_seconds_all_processes"):
                res["cpu_seconds_per_sample"] = sam[-1]["cpu_seconds_all_processes"] / sam[-1]["processes"]
        res["summary"] = summary_of(res)  # the LAST key: the driver keeps the line's last 2 000 bytes
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())


if __name__ == "__main__":
    main()
