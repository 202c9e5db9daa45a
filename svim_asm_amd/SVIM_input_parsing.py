"""Command-line options of `svim-asm haploid|diploid`.

Same option names, types and defaults as the reference parser (SVIM_input_parsing.py:7-264);
the two sub-commands share one table of option specs instead of two copied blocks.  One
addition: --device selects the GPU the hot path runs on (default 0).
"""
import argparse
import os
import sys

# (flag, kwargs) — COLLECT thresholds (reference defaults: SVIM_input_parsing.py:43-95)
_COLLECT = [
    ("--min_mapq", dict(type=int, default=20, help="Minimum mapping quality of an alignment to be used")),
    ("--min_sv_size", dict(type=int, default=40, help="Minimum SV size to detect; also the smallest CIGAR "
                                                      "insertion/deletion that becomes a signature")),
    ("--max_sv_size", dict(type=int, default=100000, help="Maximum SV size to detect; larger distances between "
                                                          "split segments become breakends")),
    ("--query_gap_tolerance", dict(type=int, default=50, help="Maximum tolerated gap between adjacent segments "
                                                              "on the query (bp)")),
    ("--query_overlap_tolerance", dict(type=int, default=50, help="Maximum tolerated overlap between adjacent "
                                                                  "segments on the query (bp)")),
    ("--reference_gap_tolerance", dict(type=int, default=50, help="Maximum tolerated gap between adjacent "
                                                                  "segments on the reference (bp), for insertions")),
    ("--reference_overlap_tolerance", dict(type=int, default=50, help="Maximum tolerated overlap between adjacent "
                                                                      "segments on the reference (bp)")),
]
# PAIR thresholds, diploid only (:219-229)
_PAIR = [
    ("--partition_max_distance", dict(type=int, default=1000, help="Maximum distance in bp between SVs in a "
                                                                   "partition")),
    ("--max_edit_distance", dict(type=int, default=200, help="Maximum edit distance between both alleles to be "
                                                             "paired")),
]
# OUTPUT switches (:105-135)
_OUTPUT = [
    ("--sample", dict(type=str, default="Sample", help="Sample ID to include in the output VCF")),
    ("--types", dict(type=str, default="DEL,INS,INV,DUP:TANDEM,DUP:INT,BND",
                     help="SV types to include in the output VCF, comma-separated")),
    ("--symbolic_alleles", dict(action="store_true", help="Use symbolic alleles such as <DEL> instead of "
                                                          "sequence alleles")),
    ("--tandem_duplications_as_insertions", dict(action="store_true", help="Represent tandem duplications as "
                                                                           "insertions (SVTYPE=INS)")),
    ("--interspersed_duplications_as_insertions", dict(action="store_true", help="Represent interspersed "
                                                                                 "duplications as insertions")),
    ("--query_names", dict(action="store_true", help="Output names of supporting query sequences in INFO/READS")),
]


def _add(parser, title, specs):
    group = parser.add_argument_group(title)
    for flag, kwargs in specs:
        group.add_argument(flag, **kwargs)


def parse_arguments(program_version, arguments=None):
    if arguments is None:
        arguments = sys.argv[1:]
    parser = argparse.ArgumentParser(
        formatter_class=argparse.RawDescriptionHelpFormatter,
        description="SVIM-asm compatible structural variant caller for genome-genome alignments; the "
                    "signature extraction and diploid pairing run on an AMD MI355X.\n"
                    "Steps: COLLECT (SV signatures from BAM), PAIR (diploid only), OUTPUT (VCF).")
    subparsers = parser.add_subparsers(help="modes", dest="sub")
    parser.add_argument("--version", "-v", action="version", version="%(prog)s {0}".format(program_version))

    haploid = subparsers.add_parser("haploid", help="Detect SVs from the alignment of an haploid query assembly "
                                                    "to a reference assembly")
    haploid.add_argument("working_dir", type=os.path.abspath, help="Working and output directory (created if missing)")
    haploid.add_argument("bam_file", type=str, help="Coordinate-sorted and indexed BAM file of the query assembly")
    haploid.add_argument("genome", type=str, help="Reference genome FASTA (indexed with .fai)")

    diploid = subparsers.add_parser("diploid", help="Detect SVs from the alignment of a diploid query assembly "
                                                    "(two haplotype BAMs) to a reference assembly")
    diploid.add_argument("working_dir", type=os.path.abspath, help="Working and output directory (created if missing)")
    diploid.add_argument("bam_file1", type=str, help="Coordinate-sorted and indexed BAM file of haplotype 1")
    diploid.add_argument("bam_file2", type=str, help="Coordinate-sorted and indexed BAM file of haplotype 2")
    diploid.add_argument("genome", type=str, help="Reference genome FASTA (indexed with .fai)")

    for sub in (haploid, diploid):
        sub.add_argument("--verbose", action="store_true", help="Enable more verbose logging")
        sub.add_argument("--device", type=int, default=0, help="HIP device index of the GPU to use")
        sub.add_argument("--no_bgzf_crc", action="store_true",
                         help="Do not check the CRC32 of the BGZF blocks of the input BAM(s) (default: every block that is "
                              "read is checked, as htslib does): blocks are then inflated only as far as needed, which "
                              "halves the CPU time of the ingest")
        _add(sub, "COLLECT", _COLLECT)
        if sub is diploid:
            _add(sub, "PAIR", _PAIR)
        _add(sub, "OUTPUT", _OUTPUT)
    return parser.parse_args(arguments)
