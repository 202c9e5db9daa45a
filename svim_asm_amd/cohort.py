"""`svim-asm-cohort`: many samples in one process — what the reference does one invocation per sample
(svim-asm:74-141) — as a STREAM: the manifest is cut into groups of G samples (default 1); a group is opened, its COLLECT
goes out as one device submission (svx_collect_batch over every BAM of the group: at G = the whole manifest the batch size
at which the CIGAR walk runs at HBM speed, bench.py's headline workload), then PAIR and the VCF per sample exactly as
`svim-asm haploid|diploid` produces them.  K workers (threads, each with a device context and stream of its own) take
the groups in manifest order, so the ingest of one group — the BAM readers' own threads, the device's share of the
inflate work on by default here — runs while another group is in PAIR or writing its VCF; at most K groups are in memory
whatever the manifest's length.  The process pays interpreter start and HIP bring-up once.

    svim-asm-cohort diploid MANIFEST GENOME [--cohort_workers K] [--cohort_group G] [--cohort_threads T] [--cohort_lanes L] [the options of svim-asm diploid]
    svim-asm-cohort haploid MANIFEST GENOME [--cohort_workers K] [--cohort_group G] [--cohort_threads T] [--cohort_lanes L] [the options of svim-asm haploid]

MANIFEST: one sample per line, whitespace-separated — working_dir bam (haploid) or working_dir bam1 bam2 (diploid);
lines starting with # are skipped.  Every sample gets its own working_dir/variants.vcf, byte-identical to the one
the single-sample command writes.  K defaults to 4 (2 below 12 CPUs' worth of time), G to 1; `--cohort_group 0` = the whole
manifest in one submission (the round-5 behaviour); T threads per BAM reader (default: the process's CPUs shared out among
the readers in flight, default_reader_threads); L inflate lanes on the device (default_lanes: one per three readers in flight).  The reference has no such mode; this is an addition on top of the
drop-in command, which is unchanged."""
import gc
import logging
import os
import sys

from svim_asm_amd import SVIM_COLLECT, _timeline, cli, shard
from svim_asm_amd.fasta import FastaFile
from svim_asm_amd.SVIM_COMBINE import write_vcf_table
from svim_asm_amd.SVIM_input_parsing import parse_arguments


def read_manifest(path, n_bams):
    samples = []
    for no, line in enumerate(open(path), 1):
        fields = line.split()
        if not fields or fields[0].startswith("#"):
            continue
        if len(fields) != 1 + n_bams:
            raise ValueError("%s:%d: expected a working directory and %d BAM path(s)" % (path, no, n_bams))
        samples.append((os.path.abspath(fields[0]), fields[1:]))
    if not samples:
        raise ValueError("%s names no sample" % path)
    return samples


def _take_option(rest, name, default):
    """Removes `name VALUE` (or name=VALUE) from the argument list of the reference's parser; returns int(VALUE)."""
    out, value, k = [], default, 0
    while k < len(rest):
        a = rest[k]
        if a == name and k + 1 < len(rest):
            value = int(rest[k + 1])
            k += 2
            continue
        if a.startswith(name + "="):
            value = int(a.split("=", 1)[1])
            k += 1
            continue
        out.append(a)
        k += 1
    return out, value


COHORT_DEVICE_INFLATE_PERCENT = 100
COHORT_DEVICE_INFLATE_WAIT_MS = int(os.environ.get("SVX_COHORT_INFLATE_WAIT_MS") or 3000)  # (the variable: tools/r06_cohort_ab.py)


def device_numa_cpus(device):
    """(PCI address, NUMA node, CPUs of that node) of a visible HIP device, from svx_device_pci_bus_id and sysfs; node and
    CPUs are None where the platform does not say (a single-node host, a container without the sysfs entries)."""
    import ctypes as C
    from svim_asm_amd import _lib
    buf = C.create_string_buffer(32)
    if _lib.load().svx_device_pci_bus_id(int(device), buf, 32) != 0:
        return None, None, None
    addr = buf.value.decode().lower()
    try:
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % addr).read())
        if node < 0:
            return addr, None, None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        return addr, node, (sorted(cpus) or None)
    except (OSError, ValueError):
        return addr, None, None


def bind_to_device_node(device):
    """Keeps the calling thread — and every thread it starts from now on: a reader's pool, the writers — on the CPUs of
    the NUMA node the device hangs on: with one cohort process per GPU of a node the processes then neither share cores
    nor read their page-locked pools across sockets.  Returns what was done, for the log."""
    addr, node, cpus = device_numa_cpus(device)
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
            return "device %d (%s): NUMA node %d, threads bound to its %d CPUs" % (device, addr, node, len(cpus))
        except OSError as e:
            return "device %d (%s): NUMA node %d, binding refused (%s)" % (device, addr, node, e)
    return "device %d (%s): no NUMA node reported, threads not bound" % (device, addr or "address unknown")


def default_workers():
    from svim_asm_amd import bamio
    return 4 if bamio.host_cpus() >= 12 else 2


def default_lanes(workers, n_bams):
    """Inflate lanes for `workers` workers of `n_bams` readers each: one per three readers in flight, at least the library's
    two.  A lane is held for a call's 50-70 ms and the workers are elsewhere most of the time, so a few lanes serve them;
    every further lane that is busy at the same time shares the same host link and costs CPU-seconds (N = 16 full-size
    samples, 4 workers, one box, twice: 2 lanes 8.4-8.9 samples/s at 0.79 CPU-seconds per sample, 3 lanes 8.6-9.0 at 0.85,
    8 lanes 7.7-8.6 at 1.0; another box at N = 24 with a 400 ms wait: 6.2 on 2 lanes — calls that found no lane in time
    decoded on the threads —, 9.2 on 4, 10.0 on 8: profiles/r06_cohort_lanes.txt).  What a call must not do is give up on the
    lane: COHORT_DEVICE_INFLATE_WAIT_MS."""
    return max(2, min(16, (workers * n_bams + 1) // 3))


def default_reader_threads(workers, n_bams):
    """Threads per BAM reader: the CPUs' worth of time the process gets (hardware threads or the cgroup's quota) shared out
    among the readers of the groups in flight.  Under a quota (cpu.max) a process that runs more threads than it has CPUs
    spends a period's budget in a fraction of the period and then stands still for the rest of it: the single-sample command
    may do that once (its record walk is 1.3 CPU-seconds: inside one 100-ms budget of 16 CPUs), a process that works
    continuously must not."""
    from svim_asm_amd import bamio
    # (one and a half times the CPUs: a reader's threads also wait — for pages, for the device's share of the inflate work)
    return max(2, int(round(1.5 * bamio.host_cpus() / float(max(1, workers * n_bams)))))


def run_group(mode, group, genome, get_ctx, first_no, n_total, workers=1, reader_threads=None):
    """One group of samples from the BAMs to the VCFs on the calling thread's device context (`get_ctx()`: asked for
    behind the record walks, which do not need it — a fresh process's first groups walk their BAMs while the HIP runtime
    comes up): (opts, working dir, BAM paths) per sample.  Returns 0, or 1 after logging why an input was refused (as the
    command does)."""
    from svim_asm_amd import bamio
    from svim_asm_amd.SVIM_COMBINE import pair_tables
    n_bams = 2 if mode == "diploid" else 1
    files = []
    _timeline.mark("group starts", sample=first_no)
    for o, wd, bams in group:
        os.makedirs(wd, exist_ok=True)
        # the files of a sample side by side, each reader with its share of the host's threads among the `workers` groups in
        # flight.  The device's share of the inflate work (the one-shot command opens its files without one: cli._open_file):
        # here the process's wall-clock is its CPU-seconds over the CPUs it may use and nobody waits for ONE sample, so
        # the device takes the WHOLE sequence-slice call of every reader (unless SVX_BAM_DEVICE_INFLATE says otherwise) and a
        # call that finds both of the device's inflate lanes taken sleeps for one instead of spending the CPU seconds
        opened = [cli._open_ahead(path, o, one_shot=False, reader_threads=reader_threads or default_reader_threads(workers, n_bams))
                  for path in bams]
        for k, path in enumerate(bams):
            f = cli._open(path, ("first", "second")[k] if n_bams == 2 else "", o, opened=opened[k])
            if f is None:
                return 1
            if bamio.env_device_inflate_percent() is None:
                f.device_inflate_percent = COHORT_DEVICE_INFLATE_PERCENT
            f.device_inflate_wait_ms = COHORT_DEVICE_INFLATE_WAIT_MS
            files.append(f)
    _timeline.mark("files open", sample=first_no)
    SVIM_COLLECT._load_together(files)  # (collect_tables would do it: here, in front of the first use of the context)
    ctx = get_ctx()
    tables = SVIM_COLLECT.collect_tables(files, group[0][0], ctx=ctx)
    _timeline.mark("COLLECT done", sample=first_no)
    for k, (o, wd, bams) in enumerate(group):
        reference = FastaFile(genome)  # (write_final_vcf closes its FastaFile, SVIM_COMBINE.py:466-467: one per sample)
        mine, mine_files = tables[k * n_bams:(k + 1) * n_bams], files[k * n_bams:(k + 1) * n_bams]
        candidates = pair_tables(mine[0], mine[1], reference, mine_files[0], o, ctx=ctx) if mode == "diploid" else mine[0]
        # as cli._run_steps: a damaged BGZF member among the inserted-sequence bytes must fail the run before a VCF is
        # written, also when nobody reads them (--symbolic_alleles)
        _ = candidates.seqs, [t.seqs for t in mine]
        _timeline.mark("PAIR done", sample=first_no + k)
        write_vcf_table(candidates, cli.__version__, mine_files[0].references, mine_files[0].lengths,
                        [entry.strip() for entry in o.types.split(",")], reference, o)
        _timeline.mark("VCF written", sample=first_no + k)
        logging.info("sample %d of %d: %s/variants.vcf", first_no + k + 1, n_total, wd)
    for f in files:
        f.close()
    del tables, files
    _timeline.mark("files closed", sample=first_no)
    gc.collect()  # (the collector is off while the workers run — the command's 10-15 % —: what a group leaves in cycles goes here)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 3 or argv[0] not in ("haploid", "diploid"):
        print(__doc__)
        return 2
    mode, manifest, genome, rest = argv[0], argv[1], argv[2], argv[3:]
    n_bams = 2 if mode == "diploid" else 1
    if shard.world()[1] > 1:
        # one process = one cohort: under a launcher every rank would collect everything and write the same files
        print("svim-asm-cohort runs as ONE process (WORLD_SIZE=%d): start one cohort per GPU with --device, or shard a "
              "single sample with `svim-asm` under the launcher" % shard.world()[1], file=sys.stderr)
        return 2
    rest, workers = _take_option(rest, "--cohort_workers", 0)
    rest, per_group = _take_option(rest, "--cohort_group", 1)
    rest, reader_threads = _take_option(rest, "--cohort_threads", 0)
    rest, lanes = _take_option(rest, "--cohort_lanes", 0)
    samples = read_manifest(manifest, n_bams)
    _timeline.mark("cohort main")
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)-7.7s]  %(message)s")
    # one options object per sample through the reference's own parser (working dir and BAM paths differ)
    opts = [parse_arguments(cli.__version__, [mode, wd] + bams + [genome] + rest) for wd, bams in samples]
    device = getattr(opts[0], "device", 0) or 0
    cli._warm_device(device)
    per_group = len(samples) if per_group <= 0 else per_group
    groups = [[(opts[k], samples[k][0], samples[k][1]) for k in range(g, min(g + per_group, len(samples)))]
              for g in range(0, len(samples), per_group)]
    workers = max(1, min(workers or default_workers(), len(groups)))
    logging.info("****************** %d samples, %d BAM files: %d group(s) of up to %d, %d worker(s) ******************",
                 len(samples), len(samples) * n_bams, len(groups), per_group, workers)
    import threading
    from svim_asm_amd import _lib
    # inflate lanes of the device (svx_bam_set_inflate_lanes, before the first load): one per reader the workers keep in
    # flight — with the walks' check on the device leg the workers waited for the default's two lanes, not for CPUs
    lanes = max(1, min(16, lanes or int(os.environ.get("SVX_COHORT_LANES") or 0) or default_lanes(workers, n_bams)))
    _lib.load().svx_bam_set_inflate_lanes(lanes)
    gc.collect()
    gc.freeze()
    gc.disable()  # as the command does for its one sample; every worker collects once per group (run_group)
    lock, state = threading.Lock(), {"next": 0, "rc": 0, "error": None}

    def work(worker_no):
        try:
            box = {}

            def get_ctx():
                # worker 0 on the process's context (the one _warm_device is bringing up), the others on their own
                if "ctx" not in box:
                    box["ctx"] = _lib.default_context(device) if worker_no == 0 else _lib.new_context(device)
                    bound = bind_to_device_node(device)  # (behind the context: the device's address needs the runtime up)
                    if worker_no == 0:
                        logging.info("AFFINITY: %s", bound)
                return box["ctx"]
            while True:
                with lock:
                    g = state["next"]
                    if g >= len(groups) or state["rc"] or state["error"]:
                        return
                    state["next"] = g + 1
                rc = run_group(mode, groups[g], genome, get_ctx, g * per_group, len(samples), workers, reader_threads or None)
                if rc:
                    with lock:
                        state["rc"] = rc
                    return
        except BaseException as e:  # noqa: BLE001 — re-raised on the main thread
            with lock:
                state["error"] = state["error"] or e

    threads = [threading.Thread(target=work, args=(k,), name="cohort-%d" % k) for k in range(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    gc.enable()
    _timeline.mark("workers done")
    _timeline.dump()
    if state["error"] is not None:
        raise state["error"]
    return state["rc"]


def entry():
    sys.exit(main())
