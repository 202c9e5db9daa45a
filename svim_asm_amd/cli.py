"""`svim-asm haploid|diploid` driver: same steps, log lines and error handling as the reference
main() (svim-asm:23-185), with BAM/FASTA access from svim_asm_amd.bamio / .fasta and the
COLLECT / PAIR arithmetic on the GPU."""
import logging
import os
import sys
from time import localtime, strftime

from svim_asm_amd import bamio, shard
from svim_asm_amd.fasta import FastaFile
from svim_asm_amd.SVIM_COMBINE import write_final_vcf
from svim_asm_amd.SVIM_input_parsing import parse_arguments

__version__ = "1.0.3"
TYPE_LABELS = (("DEL", "deletion"), ("INV", "inversion"), ("INS", "insertion"), ("DUP_TAN", "tandem duplication"),
               ("DUP_INT", "interspersed duplication"), ("BND", "breakend"))


def _collect(path, which, options):
    """Open one BAM, check sort order and index like the reference, run COLLECT.
    Returns (alignment_file, candidates) or (None, None) after logging the error."""
    the = {"": "Input", "first": "The first input", "second": "The second input"}[which]
    aln_file = bamio.AlignmentFile(path, device=getattr(options, "device", 0) or 0)
    try:
        if aln_file.header["HD"]["SO"] != "coordinate":
            logging.error("{0} BAM file needs to be coordinate-sorted. Exiting..".format(the))
            return None, None
    except KeyError:
        logging.error("Is the given {0}input BAM file coordinate-sorted? It does not contain a sorting order in "
                      "its header line. Exiting..".format(which + " " if which else ""))
        return None, None
    try:
        aln_file.check_index()
    except ValueError:
        logging.error("{0} BAM file is missing an index. Please generate with 'samtools index'. "
                      "Exiting..".format(the))
        return None, None
    candidates = shard.collect_sharded(aln_file, options)  # contig shards when launched on several GPUs
    logging.info("INGEST: rank {0}/{1} indexed {2} records ({3} of the {4} BGZF members it walked were "
                 "inflated)".format(shard.world()[0], shard.world()[1], len(aln_file), aln_file.blocks_inflated,
                                    aln_file.blocks_spanned))
    return aln_file, candidates


def _init_distributed(options):
    """One process per GPU under torch.distributed.run: rank r drives device LOCAL_RANK."""
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size <= 1:
        return False
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    options.device = int(os.environ.get("LOCAL_RANK", "0"))
    # candidate lists are exchanged as Python objects (a few MB): host-side gloo group
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world_size)
    return True


def main(arguments=None):
    options = parse_arguments(program_version=__version__, arguments=arguments)
    if not options.sub:
        print("Please choose one of the two modes ('haploid' or 'diploid'). See --help for more information.")
        return
    distributed = _init_distributed(options)
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
            import torch.distributed as dist
            dist.destroy_process_group()


def _main(options):

    log_format = logging.Formatter("%(asctime)s [%(levelname)-7.7s]  %(message)s")
    root = logging.getLogger()
    root.setLevel(logging.DEBUG if options.verbose else logging.INFO)
    if not os.path.exists(options.working_dir):
        os.makedirs(options.working_dir)
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
    logging.info("****************** Start SVIM-asm, version {0} ******************".format(__version__))
    logging.info("CMD: python3 {0}".format(" ".join(sys.argv)))
    logging.info("WORKING DIR: {0}".format(os.path.abspath(options.working_dir)))
    for arg in vars(options):
        logging.info("PARAMETER: {0}, VALUE: {1}".format(arg, getattr(options, arg)))

    logging.info("****************** STEP 1: COLLECT ******************")
    if options.sub == "haploid":
        logging.info("MODE: haploid")
        logging.info("INPUT: {0}".format(os.path.abspath(options.bam_file)))
        aln_file1, sv_candidates = _collect(options.bam_file, "", options)
        if aln_file1 is None:
            return
    else:
        logging.info("MODE: diploid")
        logging.info("INPUT1: {0}".format(os.path.abspath(options.bam_file1)))
        logging.info("INPUT2: {0}".format(os.path.abspath(options.bam_file2)))
        aln_file1, sv_candidates1 = _collect(options.bam_file1, "first", options)
        if aln_file1 is None:
            return
        aln_file2, sv_candidates2 = _collect(options.bam_file2, "second", options)
        if aln_file2 is None:
            return

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
        sv_candidates = shard.pair_sharded(sv_candidates1, sv_candidates2, reference, aln_file1, options)
    by_type = {key: [] for key, _ in TYPE_LABELS}
    for candidate in sv_candidates:
        bucket = by_type.get(candidate.type)
        if bucket is not None:
            bucket.append(candidate)

    logging.info("****************** STEP {0}: OUTPUT ******************".format(2 if options.sub == "haploid" else 3))
    for key, label in (TYPE_LABELS[0], TYPE_LABELS[1], TYPE_LABELS[2], TYPE_LABELS[3], TYPE_LABELS[4], TYPE_LABELS[5]):
        logging.info("Found {0} {1} candidates.".format(len(by_type[key]), label))
    if shard.world()[0] != 0:
        return  # every rank holds the full result; rank 0 writes it
    logging.info("Write SV candidates..")
    types_to_output = [entry.strip() for entry in options.types.split(",")]
    write_final_vcf(by_type["DUP_INT"], by_type["INV"], by_type["DUP_TAN"], by_type["DEL"], by_type["INS"],
                    by_type["BND"], __version__, aln_file1.references, aln_file1.lengths, types_to_output, reference,
                    options)
    logging.info("Done.")


def entry():
    try:
        sys.exit(main())
    except Exception as e:  # same top-level convention as the reference (svim-asm:182-185)
        logging.error(e, exc_info=True)


if __name__ == "__main__":
    entry()
