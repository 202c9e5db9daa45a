"""Inter-alignment (split-segment) SV signatures.

Mirrors analyze_read_segments(primary, supplementaries, bam, options) (SVIM_inter.py:62-340).
The adjacent-pair decision tree (:91-258) runs on the GPU (svx_segments_classify); the three
per-read post-passes — tandem-duplication merge (:261-290), interspersed duplications from
breakend pairs (:293-320) and the inversion sweep (:323-338) — run here on the raw records of each
read (a handful per read); the complete-linkage clustering of overlapping inversion breakpoints
(:42-60, scipy in the reference) is svx_linkage_cut_batch, one launch for all reads.
`analyze_read_segments_batch` is the entry COLLECT uses: one launch for all reads.
"""
from fractions import Fraction

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


def _mean(values):
    return Fraction(sum(values), len(values))  # statistics.mean of ints is exact


def candidates_from_raw(raw, primary, bam, options, sequence_slice, inversion_groups=None):
    """Raw records of ONE read (sorted-pair order) → candidates in the reference's order:
    adjacency INS/DEL/BND, then DUP_TAN, then DUP_INT, then INV.  With `inversion_groups` (a list) the
    read's groups of overlapping inversion breakpoints are appended to it instead of being clustered
    here: the caller clusters the groups of all reads with one launch and appends the INV candidates."""
    read_name = primary.query_name
    name = bam.get_reference_name
    sv_candidates, tandems, translocations, inversions = [], [], [], []
    for r in raw:
        kind = int(r["kind"])
        if kind == _lib.RAW_NONE:
            continue
        a0, a1, a2, a3, a4, a5 = (int(r[k]) for k in ("a0", "a1", "a2", "a3", "a4", "a5"))
        if kind == _lib.RAW_INS:
            sv_candidates.append(CandidateInsertion(name(a0), a1, a2, [read_name], sequence_slice(a3, a3 + a4), bam))
        elif kind == _lib.RAW_DEL:
            sv_candidates.append(CandidateDeletion(name(a0), a1, a2, [read_name], bam))
        elif kind == _lib.RAW_BND:
            c1, c2 = (name(a0), name(a3)) if a0 != a3 else (name(a0),) * 2
            sv_candidates.append(CandidateBreakend(c1, a1, _DIR[a2], c2, a4, _DIR[a5], [read_name], bam))
            translocations.append((_DIR[a2], _DIR[a5], c1, a1, c2, a4))
        elif kind == _lib.RAW_TANDEM:
            tandems.append((name(a0), a1, a2, bool(a3), bool(a4)))
        elif kind == _lib.RAW_INV:
            inversions.append((name(a0), a1, a2, _INV_SIDE[a3]))

    # tandem duplications: merge consecutive similar tuples; the direction compared against stays
    # that of the read's FIRST tuple (reference quirk)
    run = None
    for chrom, start, end, fully, direction in tandems:
        if run is None:
            run = dict(chrom=chrom, starts=[start], ends=[end], fully=[fully])
            first_direction = direction
        elif (run["chrom"] == chrom and abs(_mean(run["starts"]) - start) < 20 and
              abs(_mean(run["ends"]) - end) < 20 and first_direction == direction):
            run["starts"].append(start); run["ends"].append(end); run["fully"].append(fully)
        else:
            sv_candidates.append(_tandem_candidate(run, read_name, bam))
            run = dict(chrom=chrom, starts=[start], ends=[end], fully=[fully])
    if run is not None:
        sv_candidates.append(_tandem_candidate(run, read_name, bam))

    # interspersed duplications from pairs of breakends (:293-320)
    lo, hi = options.min_sv_size, options.max_sv_size
    for ti, (t_d1, t_d2, t_c1, t_p1, t_c2, t_p2) in enumerate(translocations):
        for b_d1, b_d2, b_c1, b_p1, b_c2, b_p2 in translocations[:ti]:
            if not (b_d1 == t_d2 and b_d2 == t_d1 and b_c1 == t_c2 and abs(b_p1 - t_p2) < 20 and
                    b_c2 == t_c1 and b_d2 == b_d1):
                continue
            if b_d1 == "fwd":
                length = t_p1 + 1 - b_p2
                if lo <= length <= hi:
                    mid = int(_mean([b_p1 + 1, t_p2]))
                    sv_candidates.append(CandidateDuplicationInterspersed(b_c2, b_p2, t_p1 + 1, b_c1, mid,
                                                                          mid + length, [read_name], bam))
            elif b_d1 == "rev":
                length = b_p2 + 1 - t_p1
                if lo <= length <= hi:
                    mid = int(_mean([b_p1, t_p2 + 1]))
                    sv_candidates.append(CandidateDuplicationInterspersed(b_c2, t_p1, b_p2 + 1, b_c1, mid,
                                                                          mid + length, [read_name], bam))

    # inversions: sweep over sorted breakpoints; the breakpoint that closes a group is dropped
    # (reference quirk).  The groups are clustered later, all reads in one launch.
    groups, active = [], []
    for inv in sorted(inversions, key=lambda i: (i[0], i[1], i[2])):
        if not active:
            active.append(inv)
        elif inv[0] == active[-1][0] and inv[1] < max(i[2] for i in active):
            active.append(inv)
        else:
            groups.append(active)
            active = []
    if active:
        groups.append(active)
    if inversion_groups is None:
        for g in groups:
            sv_candidates.extend(process_overlapping_inversions(g, read_name, bam))
    else:
        inversion_groups.extend(groups)
    return sv_candidates


def _tandem_candidate(run, read_name, bam):
    return CandidateDuplicationTandem(run["chrom"], int(_mean(run["starts"])), int(_mean(run["ends"])),
                                      len(run["starts"]), any(run["fully"]), [read_name], bam)


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
    raw = ctx.segments_classify(segs, read_off, np.asarray(read_lens, dtype=np.int32), seg_params(options))
    out, pending = [], []  # pending: (read index, groups of that read)
    for i, (primary, _) in enumerate(reads):
        groups = []
        out.append(candidates_from_raw(raw[read_off[i]:read_off[i + 1]], primary, bam, options, _slicer(primary), groups))
        if groups:
            pending.append((i, groups))
    # inversion clustering of every read in one launch (single-member groups need none)
    multi = [g for _, groups in pending for g in groups if len(g) > 1]
    labels = iter(())
    if multi:
        flat = [d for g in multi for d in _inversion_condensed(g)]
        lab = ctx.linkage_cut_batch(flat, [len(g) for g in multi], 0.3).tolist()
        pos = np.concatenate(([0], np.cumsum([len(g) for g in multi]))).tolist()
        labels = iter(lab[a:b] for a, b in zip(pos[:-1], pos[1:]))
    for i, groups in pending:
        name = reads[i][0].query_name
        for g in groups:
            out[i].extend(_inversion_candidates(g, next(labels) if len(g) > 1 else [1] * len(g), name, bam))
    return out


def analyze_read_segments(primary, supplementaries, bam, options):
    return analyze_read_segments_batch([(primary, supplementaries)], bam, options)[0]
