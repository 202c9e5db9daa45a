"""`svim-asm-cohort`: many samples in one process — what the reference does one invocation per sample
(svim-asm:74-141), with the COLLECT step of ALL samples as one device submission (svx_collect_batch over every BAM
of the cohort: the batch size at which the CIGAR walk runs at HBM speed, bench.py's headline workload), then PAIR
and the VCF per sample exactly as `svim-asm haploid|diploid` produces them.

    svim-asm-cohort diploid MANIFEST GENOME [the options of svim-asm diploid]
    svim-asm-cohort haploid MANIFEST GENOME [the options of svim-asm haploid]

MANIFEST: one sample per line, whitespace-separated — working_dir bam (haploid) or working_dir bam1 bam2 (diploid);
lines starting with # are skipped.  Every sample gets its own working_dir/variants.vcf, byte-identical to the one
the single-sample command writes.  The reference has no such mode; this is an addition on top of the drop-in
command, which is unchanged."""
import logging
import os
import sys

from svim_asm_amd import SVIM_COLLECT, cli, shard
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
    samples = read_manifest(manifest, n_bams)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)-7.7s]  %(message)s")
    # one options object per sample through the reference's own parser (working dir and BAM paths differ)
    opts = [parse_arguments(cli.__version__, [mode, wd] + bams + [genome] + rest) for wd, bams in samples]
    cli._warm_device(getattr(opts[0], "device", 0) or 0)
    files = []
    for o, (wd, bams) in zip(opts, samples):
        os.makedirs(wd, exist_ok=True)
        for k, path in enumerate(bams):
            f = cli._open(path, ("first", "second")[k] if n_bams == 2 else "", o)
            if f is None:
                return 1
            files.append(f)
    logging.info("****************** STEP 1: COLLECT (%d samples, %d BAM files, one submission) ******************",
                 len(samples), len(files))
    tables = SVIM_COLLECT.collect_tables(files, opts[0])
    for k, (o, (wd, bams)) in enumerate(zip(opts, samples)):
        reference = FastaFile(genome)  # (write_final_vcf closes its FastaFile, SVIM_COMBINE.py:466-467: one per sample)
        mine, mine_files = tables[k * n_bams:(k + 1) * n_bams], files[k * n_bams:(k + 1) * n_bams]
        if mode == "diploid":
            candidates = shard.pair_sharded(mine[0], mine[1], reference, mine_files[0], o)
        else:
            candidates = mine[0]
        # as cli._run_steps: a damaged BGZF member among the inserted-sequence bytes must fail the run before a VCF is
        # written, also when nobody reads them (--symbolic_alleles)
        _ = candidates.seqs, [t.seqs for t in mine]
        write_vcf_table(candidates, cli.__version__, mine_files[0].references, mine_files[0].lengths,
                        [entry.strip() for entry in o.types.split(",")], reference, o)
        logging.info("sample %d of %d: %s/variants.vcf", k + 1, len(samples), wd)
    return 0


def entry():
    sys.exit(main())
