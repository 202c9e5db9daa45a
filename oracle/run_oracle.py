"""File-level driver of the CPU oracle: BAM(s) + FASTA → VCF text (without ##fileDate).
TEST INFRASTRUCTURE ONLY.  Reads files with the simple pure-Python reader of
oracle/refstub/pysam.py (independent of the product's svim_asm_amd/bamio.py)."""
import argparse
import importlib.util
import os

from oracle import svim_oracle

_HERE = os.path.dirname(os.path.abspath(__file__))


def _stub():
    spec = importlib.util.spec_from_file_location("_oracle_pysam_stub", os.path.join(_HERE, "refstub", "pysam.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def default_options(**kw):
    o = argparse.Namespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, query_gap_tolerance=50,
                           query_overlap_tolerance=50, reference_gap_tolerance=50, reference_overlap_tolerance=50,
                           partition_max_distance=1000, max_edit_distance=200, sample="Sample",
                           types="DEL,INS,INV,DUP:TANDEM,DUP:INT,BND", symbolic_alleles=False,
                           tandem_duplications_as_insertions=False, interspersed_duplications_as_insertions=False,
                           query_names=False)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def read_records(bam_path):
    ps = _stub()
    bam = ps.AlignmentFile(bam_path)
    recs = []
    for a in bam.fetch():
        recs.append(dict(qname=a.query_name, flag=a.flag, tid=a.reference_id, pos=a.reference_start,
                         mapq=a.mapping_quality, cigar=list(a._cigar), seq=a._seq or "",
                         sa=a._tags.get("SA")))
    return recs, list(bam.references), list(bam.lengths)


def candidates_from_bam(bam_path, options):
    recs, names, lengths = read_records(bam_path)
    return svim_oracle.collect(recs, names, lengths, options), names, lengths


def vcf_from_files(bams, fasta_path, options, edit=None):
    ps = _stub()
    fasta = ps.FastaFile(fasta_path)
    ref_lens = dict(zip(fasta.references, fasta.lengths))
    c1, names, lengths = candidates_from_bam(bams[0], options)
    if len(bams) == 2:
        c2, _, _ = candidates_from_bam(bams[1], options)
        kw = {"edit": edit} if edit is not None else {}
        cands = svim_oracle.pair_candidates(c1, c2, fasta.fetch, names, lengths, ref_lens, options, **kw)
    else:
        cands = c1
    return svim_oracle.vcf_text(cands, fasta.fetch, names, lengths, options)
