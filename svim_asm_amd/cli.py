"""`svim-asm haploid|diploid` driver: same steps, log lines and error handling as the reference
main() (svim-asm:23-185), with BAM/FASTA access from svim_asm_amd.bamio / .fasta and the
COLLECT / PAIR arithmetic on the GPU."""
import logging
import os
import sys
from time import localtime, strftime

from svim_asm_amd import _timeline, bamio, shard
from svim_asm_amd.fasta import FastaFile
from svim_asm_amd.SVIM_COMBINE import write_vcf_table
from svim_asm_amd.table import TYPE_ORDER
from svim_asm_amd.SVIM_input_parsing import parse_arguments

__version__ = "1.0.3"
TYPE_LABELS = (("DEL", "deletion"), ("INV", "inversion"), ("INS", "insertion"), ("DUP_TAN", "tandem duplication"),
               ("DUP_INT", "interspersed duplication"), ("BND", "breakend"))


def _open_file(path, options, one_shot=True, reader_threads=None):
    f = bamio.AlignmentFile(path, device=getattr(options, "device", 0) or 0,
                            threads=reader_threads or bamio.quota_threads(2 if options.sub == "diploid" else 1, shard.world()[1]),
                            verify=False if getattr(options, "no_bgzf_crc", False) else None)
    # The device's share of the sequence slices' inflate work: the readers' own default (bamio.default_device_inflate_percent:
    # under a CPU quota the whole call — a fresh command's wall-clock is the same at every share, its CPU-seconds are 3.4 / 2.6 /
    # 2.0 at 0 / 50 / 100 %, profiles/r06_wave_cli_shares.txt; with a core per thread none).  The command's readers run on as
    # many threads as the quota has CPUs (bamio.quota_threads).  SVX_BAM_DEVICE_INFLATE overrides; svim-asm-cohort sets its own
    # (cohort.py: the whole call and a wait for a free inflate lane — there CPU-seconds count, not one sample's latency).
    # the walks leave the check of the members they touch to the device leg of the sequence slices (bamio: defer_verify; only
    # where a leg can take it, and COLLECT checks on the threads what no leg has taken): a third of the walks' CPU seconds
    f.defer_verify = not os.environ.get("SVX_BAM_NO_DEFER_VERIFY")
    if one_shot:
        asked = bamio.env_device_inflate_percent()
        f.device_inflate_percent = asked if asked is not None else bamio.default_device_inflate_percent()
    return f


def _open_ahead(path, options, one_shot=True, reader_threads=None):
    """Start opening `path` (header, reference dictionary, index) on a thread; returns a function that waits and
    hands back the file — or raises what opening raised — at the point where the caller would have opened it."""
    import threading
    box = {}

    def run():
        try:
            box["file"] = _open_file(path, options, one_shot, reader_threads)
        except BaseException as e:  # noqa: BLE001 — re-raised by the caller at its own time
            box["error"] = e
    th = threading.Thread(target=run, daemon=True)
    th.start()

    def result():
        th.join()
        if "error" in box:
            raise box["error"]
        return box["file"]
    return result


def _open(path, which, options, opened=None):
    """Open one BAM and check sort order and index like the reference (svim-asm:63-72,85-95).
    Returns the alignment file, or None after logging the error."""
    the = {"": "Input", "first": "The first input", "second": "The second input"}[which]
    aln_file = opened() if opened is not None else _open_file(path, options)
    try:
        if aln_file.header["HD"]["SO"] != "coordinate":
            logging.error("{0} BAM file needs to be coordinate-sorted. Exiting..".format(the))
            return None
    except KeyError:
        logging.error("Is the given {0}input BAM file coordinate-sorted? It does not contain a sorting order in "
                      "its header line. Exiting..".format(which + " " if which else ""))
        return None
    try:
        aln_file.check_index()
    except ValueError:
        logging.error("{0} BAM file is missing an index. Please generate with 'samtools index'. "
                      "Exiting..".format(the))
        return None
    return aln_file


def _collect(aln_files, options):
    """COLLECT of all haplotype BAMs of the sample: one device submission per rank (contig shards when
    launched on several GPUs).  Returns one CandidateTable per file."""
    tables = shard.collect_sharded(aln_files, options)
    for aln_file in aln_files:
        logging.info("INGEST: rank {0}/{1} indexed {2} records ({3} of the {4} BGZF members it walked were "
                     "inflated)".format(shard.world()[0], shard.world()[1], len(aln_file), aln_file.blocks_inflated,
                                        aln_file.blocks_spanned))
    return tables


def _init_distributed(options):
    """One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run sets them): rank r drives
    device LOCAL_RANK.  The candidate tables travel over a unix-domain socket (svim_asm_amd/shard.py): no
    torch import, no process group."""
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size <= 1:
        return False
    from svim_asm_amd import _warm
    options.device = _warm.restrict_to_local_rank()
    if options.device is None:
        options.device = int(os.environ.get("LOCAL_RANK", "0"))
    return True


def _warm_device(device):
    """Start HIP initialisation and the creation of the device context (150-350 ms in a fresh process) on a
    thread of its own: the BAM headers, indices and record walks of STEP 1 do not need the device and run
    meanwhile (ctypes releases the GIL for the call).  Errors are not lost: the first real use of the context
    creates it again and raises."""
    import threading
    from svim_asm_amd import _lib

    def create():
        try:
            _lib.default_context(device)
        except Exception:  # noqa: BLE001 — reported by the caller that needs the context
            pass
    threading.Thread(target=create, daemon=True).start()


def main(arguments=None):
    options = parse_arguments(program_version=__version__, arguments=arguments)
    if not options.sub:
        print("Please choose one of the two modes ('haploid' or 'diploid'). See --help for more information.")
        return
    distributed = _init_distributed(options)
    if not distributed:
        from svim_asm_amd import _warm
        options.device = _warm.logical_device(getattr(options, "device", 0) or 0)  # (bin/svim-asm may have narrowed the view)
    _warm_device(getattr(options, "device", 0) or 0)
    try:
        return _main(options)
    except Exception as e:
        if not distributed:
            raise
        # several ranks: a failure must not look like success to the launcher (torch.distributed.run
        # tears the other ranks down on a non-zero exit instead of leaving them in a collective)
        logging.error(e, exc_info=True)
        sys.exit(1)
    finally:
        if distributed:
            shard.shutdown()


def _main(options):

    log_format = logging.Formatter("%(asctime)s [%(levelname)-7.7s]  %(message)s")
    root = logging.getLogger()
    root.setLevel(logging.DEBUG if options.verbose else logging.INFO)
    os.makedirs(options.working_dir, exist_ok=True)  # (several ranks create it at the same moment)
    rank, world_size = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # one log file per rank (rank 0 keeps the reference's name); only rank 0 talks on the console
    suffix = "" if world_size <= 1 or rank == 0 else ".rank{0}".format(rank)
    file_handler = logging.FileHandler("{0}/SVIM_{1}{2}.log".format(options.working_dir,
                                                                  strftime("%y%m%d_%H%M%S", localtime()), suffix),
                                       mode="w")
    handlers = [file_handler] + ([logging.StreamHandler()] if rank == 0 or world_size <= 1 else [])
    for handler in handlers:
        handler.setFormatter(log_format)
        root.addHandler(handler)
    try:
        return _run(options)
    finally:
        for handler in handlers:
            root.removeHandler(handler)
        file_handler.close()


def _run(options):
    # the run allocates a few hundred thousand long-lived objects (records, candidates, VCF lines) and no
    # reference cycles worth collecting before the process exits: generational GC passes over them are
    # pure overhead (10-15 % of the wall-clock at human scale)
    import gc
    gc.disable()
    try:
        return _run_steps(options)
    finally:
        gc.enable()


def _run_steps(options):
    _timeline.mark("main starts")
    logging.info("****************** Start SVIM-asm, version {0} ******************".format(__version__))
    logging.info("CMD: python3 {0}".format(" ".join(sys.argv)))
    logging.info("WORKING DIR: {0}".format(os.path.abspath(options.working_dir)))
    for arg in vars(options):
        logging.info("PARAMETER: {0}, VALUE: {1}".format(arg, getattr(options, arg)))

    logging.info("****************** STEP 1: COLLECT ******************")
    if options.sub == "haploid":
        logging.info("MODE: haploid")
        logging.info("INPUT: {0}".format(os.path.abspath(options.bam_file)))
        aln_file1 = _open(options.bam_file, "", options)
        if aln_file1 is None:
            return
        (sv_candidates,) = _collect([aln_file1], options)
    else:
        logging.info("MODE: diploid")
        logging.info("INPUT1: {0}".format(os.path.abspath(options.bam_file1)))
        logging.info("INPUT2: {0}".format(os.path.abspath(options.bam_file2)))
        second = _open_ahead(options.bam_file2, options)  # opened beside the first one, judged in the reference's order
        aln_file1 = _open(options.bam_file1, "first", options)
        if aln_file1 is None:
            return
        aln_file2 = _open(options.bam_file2, "second", options, opened=second)
        if aln_file2 is None:
            return
        _timeline.mark("files open")
        sv_candidates1, sv_candidates2 = _collect([aln_file1, aln_file2], options)
        if _timeline.enabled():
            from svim_asm_amd import SVIM_COLLECT
            _timeline.mark("COLLECT done", stages=dict(SVIM_COLLECT.LAST_TIMING))

    try:
        reference = FastaFile(options.genome)
    except ValueError:
        logging.error("The given reference genome is missing an index file ({0}.fai). Sequence alleles cannot be "
                      "retrieved.".format(options.genome))
        return
    except IOError:
        logging.error("The given reference genome is missing ({0}). Sequence alleles cannot be "
                      "retrieved.".format(options.genome))
        return

    if options.sub == "diploid":
        logging.info("****************** STEP 2: PAIR ******************")
        _timeline.mark("PAIR starts")
        sv_candidates = shard.pair_sharded(sv_candidates1, sv_candidates2, reference, aln_file1, options)
        if _timeline.enabled():
            from svim_asm_amd import SVIM_COMBINE
            _timeline.mark("PAIR done", stages={k: v for k, v in SVIM_COMBINE.LAST_TIMING.items() if k.startswith("pair_")})
    counts = sv_candidates.counts_by_type()

    logging.info("****************** STEP {0}: OUTPUT ******************".format(2 if options.sub == "haploid" else 3))
    for key, label in TYPE_LABELS:
        logging.info("Found {0} {1} candidates.".format(int(counts[TYPE_ORDER.index(key)]), label))
    # the inserted-sequence bytes are decoded beside PAIR; with --symbolic_alleles (and on ranks other than 0) nobody
    # reads them — but a damaged BGZF member among them must still fail the run, as it would under pysam
    _ = sv_candidates.seqs
    if options.sub == "diploid":
        _ = sv_candidates1.seqs, sv_candidates2.seqs
    if shard.world()[0] != 0:
        return  # every rank holds the full result; rank 0 writes it
    logging.info("Write SV candidates..")
    types_to_output = [entry.strip() for entry in options.types.split(",")]
    write_vcf_table(sv_candidates, __version__, aln_file1.references, aln_file1.lengths, types_to_output, reference,
                    options, release_reference=False)  # (the process ends here: the kernel takes the mappings back)
    if _timeline.enabled():
        from svim_asm_amd import SVIM_COMBINE
        _timeline.mark("VCF written", stages={k: v for k, v in SVIM_COMBINE.LAST_TIMING.items() if k.startswith("vcf_")})
    logging.info("Done.")


def entry():
    try:
        sys.exit(main())
    except Exception as e:  # same top-level convention as the reference (svim-asm:182-185)
        logging.error(e, exc_info=True)


if __name__ == "__main__":
    entry()
