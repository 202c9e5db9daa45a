#!/usr/bin/env python3
"""Profile the HOST side of the product pipeline in a GPU-less container by substituting the
device context with the CPU oracle (profiling harness only — never used by the product)."""
import cProfile
import pstats
import sys
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svim_asm_amd import _lib


from tests.helpers import OracleBackedContext as OracleCtx  # noqa: E402


_lib.default_context = lambda device=0: OracleCtx()

if __name__ == "__main__":
    d = sys.argv[1]
    from svim_asm_amd import bamio, shard
    from svim_asm_amd.fasta import FastaFile
    from svim_asm_amd.SVIM_input_parsing import parse_arguments
    bams = [os.path.join(d, "hap1.bam"), os.path.join(d, "hap2.bam")]
    fasta = os.path.join(d, "ref.fa")
    opts = parse_arguments("1.0.3", ["diploid", "/tmp/hp_wd", bams[0], bams[1], fasta])
    pr = cProfile.Profile()
    t = time.time(); f1 = bamio.AlignmentFile(bams[0]).load(); f2 = bamio.AlignmentFile(bams[1]).load(); print("open+index", time.time() - t, f1.index_state(), f1.blocks_inflated, f1.blocks_spanned)
    pr.enable()
    t = time.time(); c1, c2 = shard.collect_sharded([f1, f2], opts); print("collect", time.time() - t)
    ref = FastaFile(fasta)
    t = time.time(); paired = shard.pair_sharded(c1, c2, ref, f1, opts); print("pair", time.time() - t)
    from svim_asm_amd.SVIM_COMBINE import write_vcf_table
    os.makedirs(opts.working_dir, exist_ok=True)
    t = time.time(); write_vcf_table(paired, "1.0.3", f1.references, f1.lengths, [x.strip() for x in opts.types.split(",")], ref, opts); print("vcf", time.time() - t)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
