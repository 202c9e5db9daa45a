#!/usr/bin/env python3
"""End-to-end BAM → VCF wall-clock of the product (in-process phases, the command line as a fresh process,
and the contig-sharded command line as R rank processes) on a synthetic diploid sample written as real
BAM + FASTA files (config 5 with --mean-m 400).

    python tools/e2e_bench.py --scale 1.0 [--keep DIR] [--ranks 1,2,4]

--scale 1.0 = GRCh38 contig lengths (3.1 Gbp, the configuration BASELINE's BAM→VCF wall-clock is quoted
on); 0.25 = the 772 Mbp sample of tests/golden/large_inputs.json.  At those two scales the generated
inputs are the ones the REAL reference was run on in the build container (oracle/make_golden.py full /
large): their identity is checked through digests of their UNCOMPRESSED content and the product's VCF is
compared with the digest of the reference's VCF.  The CPU oracle pipeline (CPython restatement of the
reference, C edit distance) is the checker at other scales (and with --with-oracle), timed beside it.
Prints one JSON object.
"""
import argparse
import hashlib
import json
import logging
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dataset_args(scale, sv_per_mbp=8.0, mean_m=2000, seed=3, min_gap=1500):
    from svim_asm_amd import synth
    contigs = tuple((n, max(60000, int(l * scale))) for n, l in zip(synth.GRCH38_NAMES, synth.GRCH38_LENGTHS))
    n_shared = max(4, int(sv_per_mbp * max(c[1] for c in contigs) / 1e6))
    return dict(seed=seed, contigs=contigs, diploid=True, n_shared=n_shared, n_private=max(2, n_shared // 5),
                median_aln=300000, mean_m=mean_m, min_gap=min_gap)


CONFIG5 = dict(scale=0.05, sv_per_mbp=400.0, mean_m=400, seed=5, min_gap=200)  # oracle/make_golden.py CONFIG5


def cpu_quota():
    """CPUs the cgroup grants this process per scheduling period (cpu.max quota / period), None if unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            return None if quota <= 0 else quota / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            return None


def reference_meta(scale, sv_per_mbp, mean_m, seed=3, min_gap=1500):
    """Committed description of the real reference's run on exactly these generator arguments, if any."""
    for name in ("full_inputs.json", "large_inputs.json", "config5_inputs.json"):
        path = os.path.join(ROOT, "tests", "golden", name)
        if not os.path.exists(path):
            continue
        meta = json.load(open(path))
        prm = meta["params"]
        if abs(scale - prm["scale"]) < 1e-12 and sv_per_mbp == prm["sv_per_mbp"] and mean_m == prm["mean_m"] and \
                seed == prm["seed"] and min_gap == prm.get("min_gap", 1500) and "payload_sha256" in meta:
            return name, meta
    return None, None


def drop_page_cache(paths):
    """Ask the kernel to drop the cached pages of `paths` (fsync + posix_fadvise(DONTNEED)): the run that follows
    reads its inputs from storage.  Returns True when every call succeeded (pages that are mapped or dirty
    elsewhere may stay; a box that refuses says so)."""
    ok = True
    for p in paths:
        try:
            fd = os.open(p, os.O_RDONLY)
            try:
                os.fsync(fd)
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            finally:
                os.close(fd)
        except OSError:
            ok = False
    return ok


def masked(path):
    return "".join(l for l in open(path) if not l.startswith("##fileDate="))


def run_ranks(argv, world_size, n_devices=1, timeout=900):
    """`svim-asm <argv>` as `world_size` fresh rank processes, the environment torch.distributed.run would give
    them (rank r on device r mod n_devices).  Returns (wall seconds from first start to last exit,
    [(returncode, output)] by rank).  The children make their own first GPU call; this process only waits."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    t0 = time.perf_counter()
    for rank in range(world_size):
        env = dict(os.environ)
        env.update(RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_RANK=str(rank % max(1, n_devices)),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if world_size == 1:
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bin", "svim-asm")] + list(argv), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    out = []
    for p in procs:
        try:
            text, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            text, _ = p.communicate()
            text += "\n[timeout]"
        out.append((p.returncode, text))
    return time.perf_counter() - t0, out


def run_samples(n_procs, bams, fasta, out, n_devices, check):
    """N independent `svim-asm diploid` processes at once — the unit that scales across the GPUs of a node is the
    SAMPLE: process k works on its OWN copy of the inputs (no shared page-cache pages) on device
    k mod n_devices.  Returns samples per second over the wall-clock from the first start to the last exit, the
    processes' own wall times and the CPU seconds of all of them."""
    import resource
    dirs = []
    copied = True
    for k in range(n_procs):
        d = os.path.join(out, "sample_copy_%d" % k)
        os.makedirs(d, exist_ok=True)
        for src in [fasta, fasta + ".fai"] + list(bams) + [b + ".bai" for b in bams]:
            dst = os.path.join(d, os.path.basename(src))
            if not os.path.exists(dst):
                try:
                    shutil.copyfile(src, dst)
                except OSError:  # no room for real copies: links (the processes then share page-cache pages — said in the record)
                    copied = False
                    if os.path.exists(dst):
                        os.unlink(dst)
                    os.link(src, dst)
        dirs.append(d)
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    procs, t0 = [], time.perf_counter()
    for k, d in enumerate(dirs):
        env = dict(os.environ)
        for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
            env.pop(name, None)
        wd = os.path.join(d, "wd")
        shutil.rmtree(wd, ignore_errors=True)
        argv = [sys.executable, os.path.join(ROOT, "bin", "svim-asm"), "diploid", wd, os.path.join(d, "hap1.bam"),
                os.path.join(d, "hap2.bam"), os.path.join(d, "ref.fa"), "--device", str(k % max(1, n_devices))]
        procs.append((time.perf_counter(), subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    walls, rcs, tails = [None] * n_procs, [None] * n_procs, [None] * n_procs
    pending = set(range(n_procs))
    while pending:
        for k in list(pending):
            rc = procs[k][1].poll()
            if rc is not None:
                walls[k] = time.perf_counter() - procs[k][0]
                rcs[k] = rc
                tails[k] = procs[k][1].stdout.read()[-400:]
                pending.discard(k)
        time.sleep(0.002)
    wall = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    oks = []
    for d in dirs:
        path = os.path.join(d, "wd", "variants.vcf")
        oks.append(check(masked(path)) if os.path.exists(path) else False)
    leg = {"processes": n_procs, "devices": min(n_procs, max(1, n_devices)), "process_device": [k % max(1, n_devices) for k in range(n_procs)],
           "wall_s": wall, "samples_per_s": n_procs / wall,
           "process_wall_s": walls, "rc": rcs, "cpu_seconds_all_processes": (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime),
           "cpu_quota_cpus": cpu_quota(), "own_copies_of_the_inputs": copied, "vcf_matches_real_reference_digest": oks}
    if any(rcs):
        leg["output_tail"] = tails
    for d in dirs:
        shutil.rmtree(d, ignore_errors=True)
    return leg


def run_cohort(n_samples, bams, fasta, out, device, check, workers=0, group=1, n_devices=1, procs_per_device=1):
    """`svim-asm-cohort diploid` over `n_samples` own copies of the sample's BAMs (the genome FASTA is shared, as it is
    for a real cohort) — ONE fresh process per device (`--device k`, the manifest dealt out round-robin; on one device:
    one process): samples per second over the wall-clock from the first start to the last exit (interpreter start and HIP
    bring-up included, paid once per process), CPU seconds per sample, peak resident set, where each process bound its
    threads, every VCF against the reference's digest.  The headline's own workload — many samples through one device —
    from the BAMs to the VCFs; the unit that scales across the GPUs of a node is this process."""
    import resource
    dirs, copied = [], True
    for k in range(n_samples):
        d = os.path.join(out, "cohort_copy_%d" % k)
        os.makedirs(d, exist_ok=True)
        for src in list(bams) + [b + ".bai" for b in bams]:
            dst = os.path.join(d, os.path.basename(src))
            if not os.path.exists(dst):
                try:
                    shutil.copyfile(src, dst)
                except OSError:  # no room for real copies: links (the readers then share page-cache pages — said in the record)
                    copied = False
                    if os.path.exists(dst):
                        os.unlink(dst)
                    os.link(src, dst)
        shutil.rmtree(os.path.join(d, "wd"), ignore_errors=True)
        dirs.append(d)
    n_proc = max(1, min(n_devices * max(1, procs_per_device), n_samples))  # (procs_per_device > 1: experiments)
    env = dict(os.environ)
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(name, None)
    argvs = []
    for k in range(n_proc):
        manifest = os.path.join(out, "cohort_manifest_%d_of_%d_%d.txt" % (k, n_proc, n_samples))
        with open(manifest, "w") as f:
            for d in dirs[k::n_proc]:
                f.write("%s %s %s\n" % (os.path.join(d, "wd"), os.path.join(d, "hap1.bam"), os.path.join(d, "hap2.bam")))
        argv = [sys.executable, os.path.join(ROOT, "bin", "svim-asm-cohort"), "diploid", manifest, fasta, "--device",
                str((device + k // max(1, procs_per_device)) % max(1, n_devices) if n_devices > 1 else device)]
        if workers:
            argv += ["--cohort_workers", str(workers)]
        if group != 1:
            argv += ["--cohort_group", str(group)]
        argvs.append(argv)
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.perf_counter()
    procs = [subprocess.Popen(a, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for a in argvs]
    outs = [p.communicate()[0] for p in procs]
    wall = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    oks = []
    for d in dirs:
        path = os.path.join(d, "wd", "variants.vcf")
        oks.append(check(masked(path)) if os.path.exists(path) else False)
    cpu = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)
    leg = {"samples": n_samples, "processes": n_proc, "devices": max(1, min(n_devices, n_proc)), "workers": workers or "default", "group": group, "wall_s": wall,
           "samples_per_s": n_samples / wall, "rc": [p.returncode for p in procs], "cpu_seconds": cpu,
           "cpu_seconds_per_sample": cpu / n_samples, "cpu_quota_cpus": cpu_quota(),
           "peak_rss_mb_of_any_child_so_far": ru1.ru_maxrss / 1024.0,  # (RUSAGE_CHILDREN: the largest child this process has waited for)
           "own_copies_of_the_bams": copied, "shared_genome_fasta": True, "vcf_matches_real_reference_digest": oks}
    aff = [l.split("AFFINITY: ", 1)[1].strip() for o in outs for l in o.split("\n") if "AFFINITY: " in l]
    if aff:
        leg["affinity"] = aff
    log = [l for l in outs[0].split("\n") if " worker(s) " in l]
    if log:
        leg["plan"] = log[0].split("******************")[1].strip() if "******************" in log[0] else log[0][-120:]
    if any(p.returncode != 0 for p in procs) or any(o is False for o in oks):
        leg["output_tail"] = [o[-1500:] for o in outs]
    for d in dirs:
        shutil.rmtree(d, ignore_errors=True)
    return leg


def run_e2e(scale=0.1, keep=None, sv_per_mbp=8.0, with_oracle=None, dataset=None, threads=0, repeat=1, device=0,
            ranks=(1, 2, 4), n_devices=1, mean_m=2000, in_process=True, seed=3, min_gap=1500, bam_level=1, samples=(), cohort=()):
    """Generate (or reuse) the dataset, run the product pipeline `repeat` times in this process with phase
    clocks (in_process=False: the caller must not touch the GPU — only the child processes run), then the
    command line as fresh processes (1 rank and the sharded runs), then the checker."""
    from svim_asm_amd import synth_bam
    out = keep or tempfile.mkdtemp(prefix="svx_e2e_")
    res = {"scale": scale, "mean_m": mean_m}
    t0 = time.perf_counter()
    if dataset:
        out = dataset
        fasta, bams = os.path.join(out, "ref.fa"), [os.path.join(out, "hap1.bam"), os.path.join(out, "hap2.bam")]
    else:
        fasta, bams = synth_bam.write_dataset(out, level=bam_level, **dataset_args(scale, sv_per_mbp, mean_m, seed, min_gap))
    res["generate_s"] = time.perf_counter() - t0
    res["genome_bp"] = int(sum(max(60000, int(l * scale)) for l in __import__("svim_asm_amd.synth", fromlist=["x"]).GRCH38_LENGTHS))
    res["bam_bytes"] = [os.path.getsize(b) for b in bams]

    # ---- are these the inputs the REAL reference was run on?  (digests of the uncompressed content)
    meta_name, meta = reference_meta(scale, sv_per_mbp, mean_m, seed, min_gap)
    same_inputs = None
    if meta is not None:
        t0 = time.perf_counter()
        same_inputs = all(synth_bam.payload_digest(b) == meta["payload_sha256"][os.path.basename(b)] for b in bams) and \
            synth_bam.file_digest(fasta) == meta["payload_sha256"][os.path.basename(fasta)]
        res["inputs_match_real_reference_run"] = same_inputs
        res["real_reference_fixture"] = "tests/golden/" + meta_name
        res["real_reference_wall_s_build_container"] = meta.get("reference_wall_s_build_container")
        res["input_digest_s"] = time.perf_counter() - t0

    def check(vcf_text):
        if not same_inputs:
            return None
        return hashlib.sha256(vcf_text.encode()).hexdigest() == meta["vcf_sha256"] and len(vcf_text.encode()) == meta["vcf_bytes"]

    level = logging.getLogger().level
    logging.getLogger().setLevel(logging.WARNING)
    got = None
    if in_process:
        # ---- product: timed phases through the same functions the CLI calls
        from svim_asm_amd import _lib, bamio, shard
        from svim_asm_amd.fasta import FastaFile
        from svim_asm_amd.SVIM_COMBINE import write_vcf_table
        from svim_asm_amd.SVIM_input_parsing import parse_arguments
        wd = os.path.join(out, "wd_product")
        opts = parse_arguments("1.0.3", ["diploid", wd, bams[0], bams[1], fasta])
        opts.device = device
        os.makedirs(wd, exist_ok=True)
        _lib.default_context(device)  # context creation / first touch outside the timed region
        runs = []
        import gc
        res["page_cache_dropped_before_first_run"] = drop_page_cache([fasta, fasta + ".fai"] + list(bams) + [b + ".bai" for b in bams])
        # three more passes behind the opt-out one when the readers' default gives the device a share of the sequence
        # slices' inflate work (bamio.default_device_inflate_percent): the same run with that share at 0
        default_share = bamio.AlignmentFile.device_inflate_percent
        if default_share is None:
            default_share = bamio.default_device_inflate_percent()
        n_host_only = 3 if default_share > 0 and repeat > 1 else 0
        host_only_runs = []
        for rep_no in range(max(1, repeat) + 1 + n_host_only):
            # the last pass: the opt-out (svx_bam_set_verify(0), `--no_bgzf_crc`) — members inflated only as far as
            # needed, CRC32 checked only where a member happens to be inflated to its end; reported beside the runs,
            # not among them.  The runs themselves are the default: every touched member whole + CRC32, as htslib does
            opt_out = rep_no == max(1, repeat)
            host_only = rep_no > max(1, repeat)
            bamio.AlignmentFile.device_inflate_percent = 0 if host_only else default_share
            r = {}
            prof = None
            if os.environ.get("SVX_E2E_PROFILE") and rep_no == max(1, repeat) - 1 and not opt_out:  # cProfile of the last repeat (main thread)
                import cProfile
                prof = cProfile.Profile()
                prof.enable()
            gc.collect()
            gc.disable()  # as cli._run does for the whole command
            t_all = time.perf_counter()
            t = time.perf_counter()
            cpu = [time.process_time()]  # process CPU seconds (all threads) at the phase boundaries
            # as cli._open does: headers and indices here; the record walks of both files run side by side inside COLLECT
            # (the second file is opened on a thread while the first one is, cli._open_ahead)
            import threading
            box = {}
            vf = False if opt_out else None
            th = threading.Thread(target=lambda: box.update(f=bamio.AlignmentFile(bams[1], threads=threads or bamio.ingest_threads(2), device=device, verify=vf)))
            th.start()
            f1 = bamio.AlignmentFile(bams[0], threads=threads or bamio.ingest_threads(2), device=device, verify=vf)
            th.join()
            f2 = box["f"]
            for f in (f1, f2):  # (as cli._open_file: the walks' share of the check rides on the device leg)
                f.defer_verify = not os.environ.get("SVX_BAM_NO_DEFER_VERIFY")
            f1.check_index(), f2.check_index()
            r["open_index_s"] = time.perf_counter() - t
            cpu.append(time.process_time())
            t = time.perf_counter()
            t1, t2 = shard.collect_sharded([f1, f2], opts)
            r["collect_s"] = time.perf_counter() - t
            cpu.append(time.process_time())
            from svim_asm_amd import SVIM_COLLECT
            ref = FastaFile(fasta)
            t = time.perf_counter()
            paired = shard.pair_sharded(t1, t2, ref, f1, opts)
            r["pair_s"] = time.perf_counter() - t
            cpu.append(time.process_time())
            from svim_asm_amd import SVIM_COMBINE
            r["pair_stages_s"] = {k: v for k, v in SVIM_COMBINE.LAST_TIMING.items() if k.startswith("pair_")}
            t = time.perf_counter()
            write_vcf_table(paired, "1.0.3", f1.references, f1.lengths, [x.strip() for x in opts.types.split(",")], ref, opts,
                            release_reference=False)  # as cli._run_steps: the command exits behind the VCF
            r["vcf_s"] = time.perf_counter() - t
            cpu.append(time.process_time())
            # what the run costs in CPU seconds: on a node that grants a process a CPU quota (cgroup cpu.max) the sum,
            # not the thread count, bounds the wall time
            r["cpu_seconds"] = dict(zip(("open_index", "collect", "pair", "vcf"), [b - a for a, b in zip(cpu, cpu[1:])]),
                                    total=cpu[-1] - cpu[0])
            r["vcf_stages_s"] = {k: v for k, v in SVIM_COMBINE.LAST_TIMING.items() if k.startswith("vcf_")}
            # (sequences_wait_s: where the inserted-sequence bytes were first needed — inside PAIR — not inside COLLECT)
            r["collect_stages_s"] = dict(SVIM_COLLECT.LAST_TIMING)
            r["product_total_s"] = time.perf_counter() - t_all
            # NOT in product_total_s: giving the genome's mapping back.  The reference closes its FastaFile inside
            # write_final_vcf (SVIM_COMBINE.py:466-467); the product defers the unmapping (fasta.FastaFile.close) — the command
            # leaves it to the process's exit (its cost is inside command_line_wall_s), a long-lived caller does it on a
            # background thread.  Timed here by itself so that the like-for-like figure can be read off the record:
            t_rel = time.perf_counter()
            from svim_asm_amd import fasta as _fasta_rel
            _fasta_rel.release_deferred(background=False)
            r["reference_release_s"] = time.perf_counter() - t_rel
            r["product_total_including_reference_release_s"] = r["product_total_s"] + r["reference_release_s"]
            # (the command never closes its inputs before it exits; without this the NEXT repeat's open_index_s would
            #  carry the unmapping of this run's 1.8 GB — 7 ms per file — when the names are rebound)
            r["vcf_ok"] = check(masked(os.path.join(wd, "variants.vcf")))  # every pass's VCF, outside its clock
            r["facts"] = {"index_state": f1.index_state(), "bgzf_members_inflated": [f1.blocks_inflated, f2.blocks_inflated],
                          "bgzf_members_walked": [f1.blocks_spanned, f2.blocks_spanned],
                          "candidates": [len(t1), len(t2), len(paired)],
                          "cigar_ops": [int(f1._cols["n_cig"].sum()), int(f2._cols["n_cig"].sum())],
                          # the device leg of the sequence slices (svx_bam_set_device_inflate): its share of each call and
                          # the members the device inflated + verified for the two readers
                          "device_inflate_percent": f1.effective_device_inflate_percent(),
                          "bgzf_members_inflated_on_device": [f1.device_members, f2.device_members]}
            del t1, t2, paired
            f1.close(), f2.close()
            from svim_asm_amd import fasta as _fasta
            _fasta.release_deferred(background=False)  # (the genome's mapping of this repeat, outside every clock)
            gc.enable()
            if prof is not None:
                import pstats
                prof.disable()
                pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(35)
            if host_only:
                host_only_runs.append(r["product_total_s"])
                res.setdefault("host_inflate_only_vcf_ok", []).append(r["vcf_ok"])
                r.pop("facts", None)
            elif opt_out:
                res["prefix_only_no_crc_vcf_ok"] = r["vcf_ok"]
                res["prefix_only_no_crc_total_s"] = r["product_total_s"]
                res["prefix_only_no_crc_cpu_seconds"] = r["cpu_seconds"]["total"]
                last_facts = r.pop("facts")
            else:
                runs.append(r)
        bamio.AlignmentFile.device_inflate_percent = None
        if host_only_runs:
            res["host_inflate_only_runs_total_s"] = host_only_runs
            res["host_inflate_only_total_s"] = sorted(host_only_runs)[len(host_only_runs) // 2]
        res["cpu_quota_cpus"] = cpu_quota()
        res.update(runs[0])
        if len(runs) > 1:
            res["best_run"] = min(runs, key=lambda x: x["product_total_s"])
            res["median_run"] = sorted(runs, key=lambda x: x["product_total_s"])[len(runs) // 2]
            res["all_runs_total_s"] = [r["product_total_s"] for r in runs]
        facts = dict(runs[-1].get("facts") or res.get("facts"))  # (of the last default run: the device lanes of a fresh process
        res.pop("facts", None)                                   #  come up during the first one)
        res.update(facts)
        for r in runs:
            r.pop("facts", None)
        res["ingest_threads"] = threads or bamio.ingest_threads(2)
        got = masked(os.path.join(wd, "variants.vcf"))
        res["vcf_records"] = sum(1 for l in got.split("\n") if l and not l.startswith("#"))
        oks = [r.get("vcf_ok") for r in runs] + [res.get("prefix_only_no_crc_vcf_ok"), check(got)]
        res["vcf_matches_real_reference_digest"] = None if all(o is None for o in oks) else all(o for o in oks if o is not None)

    # ---- the command line itself, as fresh processes: interpreter start, imports, HIP initialisation and log
    # writing included — what `time svim-asm diploid ...` shows; R > 1: contig-sharded ranks (BASELINE config 4)
    sharded = []
    for R in ranks:
        wd_r = os.path.join(out, "wd_cli_r%d" % R)
        walls = []
        for _ in range(3 if R == 1 else 2):  # wall_s: the median (R = 1: of three; R > 1: the worse of two)
            shutil.rmtree(wd_r, ignore_errors=True)
            wall, results = run_ranks(["diploid", wd_r, bams[0], bams[1], fasta], R, n_devices)
            walls.append(wall)
            if any(rc != 0 for rc, _ in results):
                break
        wall = sorted(walls)[len(walls) // 2]
        leg = {"ranks": R, "devices": min(R, max(1, n_devices)), "wall_s": wall, "all_wall_s": walls, "rc": [rc for rc, _ in results]}
        if all(rc == 0 for rc, _ in results):
            text = masked(os.path.join(wd_r, "variants.vcf"))
            if got is None:
                got = text
                res["vcf_records"] = sum(1 for l in got.split("\n") if l and not l.startswith("#"))
            leg["vcf_identical_to_single_process"] = (text == got)
            leg["vcf_matches_real_reference_digest"] = check(text)
            walked = []
            for r in range(R):
                logs = [f for f in os.listdir(wd_r) if f.endswith(".log") and (".rank%d." % r in f or (r == 0 and ".rank" not in f))]
                n = 0
                for f in logs:
                    for line in open(os.path.join(wd_r, f)):
                        if "INGEST: rank" in line:
                            n += int(line.split(" of the ")[1].split(" BGZF")[0])
                walked.append(n)
            leg["bgzf_members_walked_per_rank"] = walked
        else:
            leg["output_tail"] = [t[-600:] for _, t in results]
        sharded.append(leg)
    res["cli_ranks"] = sharded
    if sharded and sharded[0]["ranks"] == 1:
        res["cli_wall_s"] = sharded[0]["wall_s"]
        res["cli_all_wall_s"] = sharded[0]["all_wall_s"]
    if samples:
        res["samples"] = [run_samples(n, bams, fasta, out, n_devices, check) for n in samples]
    if cohort:
        res["cohort"] = [run_cohort(n, bams, fasta, out, device, check, n_devices=n_devices) for n in cohort]

    if with_oracle is None:
        with_oracle = not same_inputs
    if with_oracle and got is not None:
        from oracle import orc, run_oracle
        t = time.perf_counter()
        exp = run_oracle.vcf_from_files(bams, fasta, run_oracle.default_options(),
                                        edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
        res["oracle_total_s"] = time.perf_counter() - t
        res["vcf_identical"] = (got == exp)
    logging.getLogger().setLevel(level)
    if not keep and not dataset:
        shutil.rmtree(out)
    else:
        res["dir"] = out
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.1)
    ap.add_argument("--keep", default=None)
    ap.add_argument("--sv-per-mbp", type=float, default=8.0)
    ap.add_argument("--mean-m", type=int, default=2000, help="mean M-run length of the CIGARs (400: BASELINE config 5)")
    ap.add_argument("--config5", action="store_true", help="BASELINE config 5 as a diploid sample: the generator arguments of "
                    "oracle/make_golden.py config5 (10x small-indel density, crowded partitions, > 131072 candidates)")
    ap.add_argument("--samples", default="", help="comma-separated process counts: N independent `svim-asm diploid` processes at "
                    "once, each on its own copy of the sample (the mode that scales across GPUs)")
    ap.add_argument("--cohort", default="", help="comma-separated sample counts: `svim-asm-cohort diploid` as one process over N own "
                    "copies of the sample's BAMs (samples per second, CPU seconds per sample, peak RSS)")
    ap.add_argument("--with-oracle", action="store_true", help="run the CPU oracle pipeline even when the real reference's digest is available")
    ap.add_argument("--dataset", default=None, help="directory holding ref.fa / hap1.bam / hap2.bam from an earlier --keep run")
    ap.add_argument("--bam-level", type=int, default=1, help="zlib level of the BGZF members of the generated BAMs (1: the "
                    "goldens' writer; 6: what samtools writes by default; 100 + n: libdeflate at level n, what an htslib built with "
                    "libdeflate writes — the records, and so the expected VCF, are the same)")
    ap.add_argument("--threads", type=int, default=0, help="ingest threads (0: one per hardware thread, at most 64)")
    ap.add_argument("--repeat", type=int, default=1, help="repeat the product pipeline, report the best run too")
    ap.add_argument("--ranks", default="1,2,4", help="rank counts of the command-line runs")
    ap.add_argument("--devices", type=int, default=1, help="HIP devices the ranks are spread over")
    ap.add_argument("--no-in-process", action="store_true", help="only the command-line runs (this process never touches the GPU)")
    args = ap.parse_args()
    seed, min_gap = 3, 1500
    if args.config5:
        args.scale, args.sv_per_mbp, args.mean_m, seed, min_gap = (CONFIG5[k] for k in ("scale", "sv_per_mbp", "mean_m", "seed", "min_gap"))
    print(json.dumps(run_e2e(args.scale, args.keep, args.sv_per_mbp, True if args.with_oracle else None, args.dataset,
                             args.threads, args.repeat, ranks=tuple(int(x) for x in args.ranks.split(",") if x),
                             n_devices=args.devices, mean_m=args.mean_m, in_process=not args.no_in_process, seed=seed,
                             min_gap=min_gap, bam_level=args.bam_level,
                             samples=tuple(int(x) for x in args.samples.split(",") if x),
                             cohort=tuple(int(x) for x in args.cohort.split(",") if x))))


if __name__ == "__main__":
    main()
