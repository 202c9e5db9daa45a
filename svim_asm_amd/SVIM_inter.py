"""Inter-alignment (split-segment) SV signatures.

Mirrors analyze_read_segments(primary, supplementaries, bam, options) (SVIM_inter.py:62-340).
The adjacent-pair decision tree (:91-258) runs on the GPU (svx_segments_classify) and so do the three
per-read post-passes — tandem-duplication merge (:261-290), interspersed duplications from breakend
pairs (:293-320), inversion sweep and complete-linkage clustering (:323-338, :42-60; scipy in the
reference) — svx_segments_postpass, one lane per read.  The host turns the records of both kernels
into Candidate objects.  `analyze_read_segments_batch` is the entry COLLECT uses: two launches for
all reads of a BAM.
"""
import numpy as np

from svim_asm_amd import _lib
from svim_asm_amd.SVCandidate import (CandidateBreakend, CandidateDeletion, CandidateDuplicationInterspersed,
                                      CandidateDuplicationTandem, CandidateInsertion, CandidateInversion)

_DIR = ("fwd", "rev")
_INV_SIDE = ("left_fwd", "left_rev", "right_fwd", "right_rev")


def is_similar(chr1, start1, end1, chr2, start2, end2):
    return chr1 == chr2 and abs(start1 - start2) < 20 and abs(end1 - end2) < 20


def reciprocal_overlap_distance(inversion1, inversion2):
    """Distance of two inversion breakpoints (start, end, side) for the linkage (:19-39)."""
    start1, end1, side1 = inversion1
    start2, end2, side2 = inversion2
    if side1 == side2 or start2 >= end1 or start1 >= end2:
        return 1
    overlap = min(end1, end2) - max(start1, start2)
    return 1 - min(overlap / float(end1 - start1), overlap / float(end2 - start2))


def _inversion_condensed(active_inversions):
    """Condensed distance vector scipy's pdist hands to linkage (:45-47): the metric on the float64 rows
    [start, end, 0 (left) / 1 (right)]."""
    rows = [(float(inv[1]), float(inv[2]), 0.0 if inv[3].split("_")[0] == "left" else 1.0) for inv in active_inversions]
    return [reciprocal_overlap_distance(rows[i], rows[j]) for i in range(len(rows) - 1) for j in range(i + 1, len(rows))]


def _inversion_candidates(active_inversions, labels, query_name, bam):
    clusters = [[] for _ in range(max(labels))]
    for inv, lab in zip(active_inversions, labels):
        clusters[lab - 1].append(inv)
    return [CandidateInversion(cl[0][0], max(i[1] for i in cl), min(i[2] for i in cl), [query_name],
                               len(cl) > 1, bam) for cl in clusters]


def process_overlapping_inversions(active_inversions, query_name, bam, ctx=None):
    """Complete linkage over the breakpoints' reciprocal-overlap distance, cut at 0.3 (:42-60); the
    clustering runs on the GPU (svx_linkage_cut_batch, scipy's label order)."""
    if len(active_inversions) < 2:
        labels = [1] * len(active_inversions)
    else:
        ctx = ctx or _lib.default_context()
        labels = ctx.linkage_cut_batch(_inversion_condensed(active_inversions), [len(active_inversions)], 0.3).tolist()
    return _inversion_candidates(active_inversions, labels, query_name, bam)


def segment_row(alignment):
    """(q_start, q_end, ref_id, ref_start, ref_end, is_reverse) of one alignment (:66-81)."""
    if alignment.is_reverse:
        length = alignment.infer_read_length()
        q_start, q_end = length - alignment.query_alignment_end, length - alignment.query_alignment_start
    else:
        q_start, q_end = alignment.query_alignment_start, alignment.query_alignment_end
    return (q_start, q_end, alignment.reference_id, alignment.reference_start, alignment.reference_end,
            1 if alignment.is_reverse else 0)


def candidates_from_records(raw, post, primary, bam, sequence_slice):
    """Candidates of ONE read in the reference's order (SVIM_inter.py:91-338): the adjacency records of
    svx_segments_classify (INS / DEL / BND, sorted-pair order), then the derived records of
    svx_segments_postpass (tandem duplications, interspersed duplications, inversions)."""
    read_name = primary.query_name
    name = bam.get_reference_name
    sv_candidates = []
    for r in raw:
        kind = int(r["kind"])
        if kind == _lib.RAW_INS:
            a3 = int(r["a3"])
            sv_candidates.append(CandidateInsertion(name(int(r["a0"])), int(r["a1"]), int(r["a2"]), [read_name],
                                                    sequence_slice(a3, a3 + int(r["a4"])), bam))
        elif kind == _lib.RAW_DEL:
            sv_candidates.append(CandidateDeletion(name(int(r["a0"])), int(r["a1"]), int(r["a2"]), [read_name], bam))
        elif kind == _lib.RAW_BND:
            sv_candidates.append(CandidateBreakend(name(int(r["a0"])), int(r["a1"]), _DIR[int(r["a2"])], name(int(r["a3"])),
                                                   int(r["a4"]), _DIR[int(r["a5"])], [read_name], bam))
    for r in post:
        kind = int(r["kind"])
        if kind == _lib.POST_TANDEM:
            sv_candidates.append(CandidateDuplicationTandem(name(int(r["a0"])), int(r["a1"]), int(r["a2"]), int(r["a3"]),
                                                            bool(r["a4"]), [read_name], bam))
        elif kind == _lib.POST_DUP_INT:
            sv_candidates.append(CandidateDuplicationInterspersed(name(int(r["a0"])), int(r["a1"]), int(r["a2"]),
                                                                  name(int(r["a3"])), int(r["a4"]), int(r["a5"]),
                                                                  [read_name], bam))
        elif kind == _lib.POST_INV:
            sv_candidates.append(CandidateInversion(name(int(r["a0"])), int(r["a1"]), int(r["a2"]), [read_name],
                                                    bool(r["a3"]), bam))
    return sv_candidates


def contig_ranks(bam):
    """Rank of every contig name under Python str ordering: the inversion sweep sorts by name (:323)."""
    names = list(bam.references)
    rank = np.zeros(len(names), dtype=np.int32)
    for k, i in enumerate(sorted(range(len(names)), key=lambda i: names[i])):
        rank[i] = k
    return rank


def seg_params(options):
    return _lib.SegParams(int(options.min_sv_size), int(options.max_sv_size), int(options.query_gap_tolerance),
                          int(options.query_overlap_tolerance), int(options.reference_gap_tolerance),
                          int(options.reference_overlap_tolerance))


def _slicer(primary):
    f = getattr(primary, "seq_slice", None)
    if f is not None:
        return f
    return lambda a, b: primary.query_sequence[a:b]


def analyze_read_segments_batch(reads, bam, options, ctx=None, rows=None, read_lens=None):
    """reads: list of (primary, [supplementary, ...]).  Returns one candidate list per read.
    `rows`/`read_lens` may carry precomputed segment rows (COLLECT computes them on the GPU)."""
    ctx = ctx or _lib.default_context()
    if not reads:
        return []
    if rows is None:
        rows = [[segment_row(a) for a in [p] + list(s)] for p, s in reads]
    if read_lens is None:
        read_lens = [p.infer_read_length() for p, _ in reads]
    counts = [len(r) for r in rows]
    read_off = np.concatenate(([0], np.cumsum(counts))).astype(np.uint32)
    segs = np.array([t for r in rows for t in r], dtype=np.int32).reshape(-1, 6)
    segs = np.ascontiguousarray(segs).view(_lib.SEG_DTYPE).reshape(-1)
    prm = seg_params(options)
    raw = ctx.segments_classify(segs, read_off, np.asarray(read_lens, dtype=np.int32), prm)
    post, first = ctx.segments_postpass(raw, read_off, contig_ranks(bam), prm)
    return [candidates_from_records(raw[read_off[i]:read_off[i + 1]], post[first[i]:first[i + 1]], primary, bam,
                                    _slicer(primary)) for i, (primary, _) in enumerate(reads)]


def analyze_read_segments(primary, supplementaries, bam, options):
    return analyze_read_segments_batch([(primary, supplementaries)], bam, options)[0]
