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
import numpy as np
from oracle import orc
from svim_asm_amd import _lib


class OracleCtx(object):
    def cigar_extract(self, cigar, aln_off, ref_start=None, min_len=40, cap=None, op=None):
        return orc.cigar_extract(cigar, aln_off, ref_start, min_len)

    def cigar_stats(self, cigar, aln_off):
        return orc.cigar_stats(cigar, aln_off)

    def segments_classify(self, segs, read_off, read_len, params):
        prm = [getattr(params, f) for f, _ in params._fields_]
        return orc.segments_classify(np.ascontiguousarray(segs).view(orc.SEG_DTYPE), read_off, read_len, prm).view(_lib.RAW_DTYPE)

    def pair_partition(self, keys, max_dist):
        return orc.pair_partition(keys, max_dist)

    def edit_distance_batch(self, seq, a_off, a_len, b_off, b_len, k_max=0xFFFFFFFF):
        seq = np.ascontiguousarray(seq, np.uint8)
        out = []
        for ao, al, bo, bl in zip(a_off, a_len, b_off, b_len):
            d = orc.edit_distance(seq[ao:ao + al].tobytes(), seq[bo:bo + bl].tobytes())
            out.append(d if d <= k_max else 0xFFFFFFFF)
        return np.array(out, dtype=np.uint32)


    def segments_postpass(self, raw, read_off, contig_rank, params):
        from oracle import svim_oracle
        prm = [getattr(params, f) for f, _ in params._fields_]
        code = {"TANDEM": 1, "DUP_INT": 2, "INV": 3}
        recs, first = [], [0]
        for r in range(len(read_off) - 1):
            rows = [tuple(int(x[k]) for k in ("kind", "a0", "a1", "a2", "a3", "a4", "a5")) for x in raw[read_off[r]:read_off[r + 1]]]
            for t in svim_oracle.postpass_records(rows, list(contig_rank), prm[0], prm[1]):
                recs.append(tuple([code[t[0]]] + [int(v) for v in t[1:]] + [0] * (8 - len(t))))
            first.append(len(recs))
        return np.array(recs, dtype=_lib.RAW_DTYPE) if recs else np.zeros(0, dtype=_lib.RAW_DTYPE), np.array(first, np.int64)

    def haplotype_distance_batch(self, pool, pieces, k_max=0xFFFFFFFF):
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}

        def build(three):
            out = []
            for off, ln, rep, flags in three:
                s = bytes(pool[off:off + ln]).decode("latin-1")
                if flags & 1:
                    s = s.upper()
                if flags & 2:
                    s = "".join(comp.get(b, b) for b in reversed(s))
                out.append(s * rep)
            return "".join(out).encode("latin-1")
        out = []
        for p in range(len(pieces) // 6):
            d = orc.edit_distance(build(pieces[p * 6:p * 6 + 3].tolist()), build(pieces[p * 6 + 3:p * 6 + 6].tolist()))
            out.append(d if d <= k_max else 0xFFFFFFFF)
        return np.array(out, dtype=np.uint32)

    def linkage_cut_batch(self, dist, n_members, cutoff):
        out, at = [], 0
        for n in n_members:
            m = n * (n - 1) // 2
            out.extend(orc.linkage_cut(dist[at:at + m], n, cutoff).tolist() if n > 1 else [1] * n)
            at += m
        return np.array(out, dtype=np.uint32)


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
    t = time.time(); c1 = shard.collect_sharded(f1, opts); c2 = shard.collect_sharded(f2, opts); print("collect", time.time() - t)
    ref = FastaFile(fasta)
    t = time.time(); paired = shard.pair_sharded(c1, c2, ref, f1, opts); print("pair", time.time() - t)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
