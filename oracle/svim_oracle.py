"""Pure-Python CPU restatement of the SVIM-asm hot path (COLLECT → PAIR → VCF text).

TEST INFRASTRUCTURE ONLY — imported from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never from the product package.  Written from SURVEY.md Appendix A; each
function cites the reference lines (under /root/reference/src/svim_asm) it restates.

Pinning (tests/test_oracle_pins.py, tests/test_oracle_vs_reference.py): the reference's own
known-answer vectors (tests/test_intra.py:8-22, tests/test_inter.py:8-11, the two
chimeric_read*.bam fixtures of tests/test_satag.py) and golden vectors / randomised
differential runs against the imported reference in the build container
(oracle/make_golden.py; the reference needs stub pysam/edlib modules, oracle/refstub/).

Data model: an alignment record is a dict
    {qname, flag, tid, pos, mapq, cigar: [(op, len), ...], seq: str, sa: str | None}
a header is (names, lengths); candidates are plain tuples, see cand_*() below.
"""
import re
from collections import defaultdict
from fractions import Fraction

# ----------------------------------------------------------------------------- a1
def analyze_cigar_indel(tuples, min_length):
    """SVIM_intra.py:8-30."""
    pos_ref = pos_read = 0
    out = []
    for op, length in tuples:
        if op in (0, 7, 8):          # :14-16, :27-29
            pos_ref += length
            pos_read += length
        elif op == 1:                # :17-20
            if length >= min_length:
                out.append((pos_ref, pos_read, length, "INS"))
            pos_read += length
        elif op == 2:                # :21-24
            if length >= min_length:
                out.append((pos_ref, pos_read, length, "DEL"))
            pos_ref += length
        elif op == 4:                # :25-26
            pos_read += length
        # N, H, P, B: no branch in :14-29
    return out


# ----------------------------------------------------------------- pysam/htslib facts
def reference_end(rec):
    """htslib bam_endpos: pos + Σ{M,D,N,=,X}, at least 1."""
    rlen = sum(l for o, l in rec["cigar"] if o in (0, 2, 3, 7, 8))
    return rec["pos"] + (rlen if rlen else 1)


def query_alignment_start(rec):
    s = 0
    for o, l in rec["cigar"]:
        if o == 5:
            continue
        if o == 4:
            s += l
        else:
            break
    return s


def query_alignment_end(rec):
    """pysam getQueryEnd: with a stored sequence l_qseq minus trailing soft clips; without one
    (SA-derived segments, SVIM_COLLECT.py:35) leading clips + Σ{M,I,=,X}."""
    lq = len(rec.get("seq") or "")
    cig = rec["cigar"]
    if lq == 0:
        end = 0
        for o, l in cig:
            if o in (0, 1, 7, 8) or (o == 4 and end == 0):
                end += l
        return end
    end = lq
    for k in range(len(cig) - 1, 0, -1):
        o, l = cig[k]
        if o == 5:
            continue
        if o == 4:
            end -= l
        else:
            break
    return end


def infer_read_length(rec):
    return sum(l for o, l in rec["cigar"] if o in (0, 1, 4, 5, 7, 8))


# ----------------------------------------------------------------------- candidates (a2, a5)
def _clamp(start, end, length):
    return max(0, start), min(length, end)


def cand_del(contig, start, end, reads, lens, gt="1/1"):
    assert end >= start                              # SVCandidate.py:40
    s, e = _clamp(start, end, lens[contig])          # :44-46
    return ("DEL", contig, s, e, tuple(reads), gt)


def cand_ins(contig, start, end, reads, seq, lens, gt="1/1"):
    assert end >= start                              # :130
    s, e = _clamp(start, end, lens[contig])          # :134-136
    return ("INS", contig, s, e, tuple(reads), seq, gt)


def cand_inv(contig, start, end, reads, complete, lens, gt="1/1"):
    assert end >= start                              # :83
    s, e = _clamp(start, end, lens[contig])
    return ("INV", contig, s, e, tuple(reads), bool(complete), gt)


def cand_tan(contig, start, end, copies, fully, reads, lens, gt="1/1"):
    assert end >= start                              # :181
    s, e = _clamp(start, end, lens[contig])
    return ("DUP_TAN", contig, s, e, copies, bool(fully), tuple(reads), gt)


def cand_int(sc, ss, se, dc, ds, de, reads, lens, cutpaste=False, gt="1/1"):
    assert se >= ss and de >= ds                     # :266-267
    ss, se = _clamp(ss, se, lens[sc])
    ds, de = _clamp(ds, de, lens[dc])
    return ("DUP_INT", sc, ss, se, dc, ds, de, tuple(reads), bool(cutpaste), gt)


def cand_bnd(sc, ss, sd, dc, ds, dd, reads, lens, gt="1/1"):
    """SVCandidate.py:351-376: lexicographic-min normalisation with direction flip."""
    flip = {"fwd": "rev", "rev": "fwd"}
    if sc < dc or (sc == dc and ss < ds):
        return ("BND", sc, min(lens[sc], max(0, ss)), sd, dc, min(lens[dc], max(0, ds)), dd, tuple(reads), gt)
    return ("BND", dc, min(lens[dc], max(0, ds)), flip[dd], sc, min(lens[sc], max(0, ss)), flip[sd],
            tuple(reads), gt)


def get_key(c):
    """Candidate.get_key (SVCandidate.py:17-19,147-148,292-293,386-387)."""
    t = c[0]
    if t in ("DEL", "INV", "DUP_TAN"):
        return (t, c[1], (c[2] + c[3]) // 2)
    if t == "INS":
        return (t, c[1], c[2])
    if t == "DUP_INT":
        return (t, c[4], c[5])
    return (t, c[1], c[2])  # BND


# ----------------------------------------------------------------------------- a2
def analyze_alignment_indel(rec, names, lens, min_sv_size):
    """SVIM_intra.py:33-44."""
    out = []
    contig = names[rec["tid"]]
    for pos_ref, pos_read, length, typ in analyze_cigar_indel(rec["cigar"], min_sv_size):
        start = rec["pos"] + pos_ref
        if typ == "DEL":
            out.append(cand_del(contig, start, start + length, [rec["qname"]], lens))
        else:
            out.append(cand_ins(contig, start, start + length, [rec["qname"]],
                                rec["seq"][pos_read:pos_read + length], lens))
    return out


# ----------------------------------------------------------------------------- a3
def is_similar(chr1, start1, end1, chr2, start2, end2):
    """SVIM_inter.py:12-16."""
    return chr1 == chr2 and abs(start1 - start2) < 20 and abs(end1 - end2) < 20


def _mean(xs):
    return Fraction(sum(xs), len(xs))  # statistics.mean on ints is exact-rational (A3.8)


def reciprocal_overlap_distance(a, b):
    """SVIM_inter.py:19-39 (float64 arithmetic as in the reference)."""
    s1, e1, d1 = a
    s2, e2, d2 = b
    if d1 == d2 or s2 >= e1 or s1 >= e2:
        return 1
    overlap = min(e1, e2) - (s2 if s2 >= s1 else s1)
    return 1 - min(overlap / float(e1 - s1), overlap / float(e2 - s2))


def _complete_linkage_labels(dist, n, t):
    """scipy linkage(method='complete') + fcluster(criterion='distance') — delegated to scipy,
    the reference's own dependency (SVIM_inter.py:47-48, SVIM_COMBINE.py:134-135,155-156)."""
    import numpy as np
    from scipy.cluster.hierarchy import fcluster, linkage
    z = linkage(np.array(dist, dtype=float), method="complete")
    return list(fcluster(z, t, criterion="distance"))


def process_overlapping_inversions(active, qname, lens):
    """SVIM_inter.py:42-60."""
    if len(active) < 2:
        clusters = [active]
    else:
        rows = [(i[1], i[2], 0 if i[3].split("_")[0] == "left" else 1) for i in active]
        dist = [reciprocal_overlap_distance(rows[i], rows[j])
                for i in range(len(rows) - 1) for j in range(i + 1, len(rows))]
        labels = _complete_linkage_labels(dist, len(rows), 0.3)
        clusters = [[] for _ in range(max(labels))]
        for idx, lab in enumerate(labels):
            clusters[lab - 1].append(active[idx])
    out = []
    for cl in clusters:
        out.append(cand_inv(cl[0][0], max(i[1] for i in cl), min(i[2] for i in cl), [qname], len(cl) > 1, lens))
    return out


def analyze_read_segments(primary, supplementaries, names, lens, o):
    """SVIM_inter.py:62-340.  `o` has min_sv_size, max_sv_size, query_gap_tolerance,
    query_overlap_tolerance, reference_gap_tolerance, reference_overlap_tolerance."""
    qname = primary["qname"]
    segs = []
    for rec in [primary] + supplementaries:          # :64-81
        rev = bool(rec["flag"] & 0x10)
        if rev:
            L = infer_read_length(rec)
            qs, qe = L - query_alignment_end(rec), L - query_alignment_start(rec)
        else:
            qs, qe = query_alignment_start(rec), query_alignment_end(rec)
        segs.append(dict(q_start=qs, q_end=qe, ref_id=rec["tid"], ref_start=rec["pos"],
                         ref_end=reference_end(rec), rev=rev))
    segs.sort(key=lambda s: (s["q_start"], s["q_end"]))  # :83 (stable)
    out, tandems, transl, inversions = [], [], [], []
    L_primary = infer_read_length(primary)

    def bnd(c1, p1, d1, c2, p2, d2):
        out.append(cand_bnd(c1, p1, d1, c2, p2, d2, [qname], lens))
        transl.append((d1, d2, c1, p1, c2, p2))

    for cur, nxt in zip(segs, segs[1:]):             # :91-93
        d_read = nxt["q_start"] - cur["q_end"]       # :95
        if cur["ref_id"] == nxt["ref_id"]:
            chrom = names[cur["ref_id"]]
            if cur["rev"] == nxt["rev"]:
                d_ref = cur["ref_start"] - nxt["ref_end"] if cur["rev"] else nxt["ref_start"] - cur["ref_end"]
                if d_read >= -o.query_overlap_tolerance:                   # :108
                    if d_ref >= -o.reference_overlap_tolerance:            # :110
                        dev = d_read - d_ref
                        if dev >= o.min_sv_size:                           # :113
                            if d_ref <= o.reference_gap_tolerance:         # :115
                                if not cur["rev"]:
                                    seq = primary["seq"][cur["q_end"]:cur["q_end"] + dev]
                                    out.append(cand_ins(chrom, cur["ref_end"], cur["ref_end"] + dev, [qname], seq, lens))
                                else:
                                    a = L_primary - nxt["q_start"]
                                    seq = primary["seq"][a:a + dev]
                                    out.append(cand_ins(chrom, cur["ref_start"], cur["ref_start"] + dev, [qname], seq, lens))
                        elif -o.max_sv_size <= dev <= -o.min_sv_size:      # :123
                            if d_read <= o.query_gap_tolerance:
                                s = nxt["ref_end"] if cur["rev"] else cur["ref_end"]
                                out.append(cand_del(chrom, s, s - dev, [qname], lens))
                        elif dev < -o.max_sv_size:                         # :131
                            if d_read <= o.query_gap_tolerance:
                                if not cur["rev"]:
                                    bnd(chrom, cur["ref_end"] - 1, "fwd", chrom, nxt["ref_start"], "fwd")
                                else:
                                    bnd(chrom, cur["ref_start"], "rev", chrom, nxt["ref_end"] - 1, "rev")
                    else:                                                   # :141
                        if d_read <= o.query_gap_tolerance:
                            dev = d_read - d_ref
                            if dev >= o.min_sv_size:
                                if not cur["rev"]:
                                    if nxt["ref_end"] > cur["ref_start"]:
                                        tandems.append((chrom, nxt["ref_start"], nxt["ref_start"] + dev, True, True))
                                    elif d_ref >= -o.max_sv_size:
                                        tandems.append((chrom, nxt["ref_start"], nxt["ref_start"] + dev, False, True))
                                    else:
                                        bnd(chrom, cur["ref_end"] - 1, "fwd", chrom, nxt["ref_start"], "fwd")
                                else:
                                    if nxt["ref_start"] < cur["ref_end"]:
                                        tandems.append((chrom, cur["ref_start"], cur["ref_start"] + dev, True, False))
                                    elif d_ref >= -o.max_sv_size:
                                        tandems.append((chrom, cur["ref_start"], cur["ref_start"] + dev, False, False))
                                    else:
                                        bnd(chrom, cur["ref_start"], "rev", chrom, nxt["ref_end"] - 1, "rev")
            else:
                if -o.query_overlap_tolerance <= d_read <= o.query_gap_tolerance:   # :175, :201
                    if not cur["rev"]:                                              # :172 fwd → rev
                        dev = d_read - (nxt["ref_end"] - cur["ref_end"])
                        if nxt["ref_start"] - cur["ref_end"] >= -o.reference_overlap_tolerance:
                            if o.min_sv_size <= -dev <= o.max_sv_size:
                                inversions.append((chrom, cur["ref_end"], cur["ref_end"] - dev, "left_fwd"))
                            else:
                                bnd(chrom, cur["ref_end"] - 1, "fwd", chrom, nxt["ref_end"] - 1, "rev")
                        elif cur["ref_start"] - nxt["ref_end"] >= -o.reference_overlap_tolerance:
                            if o.min_sv_size <= dev <= o.max_sv_size:
                                inversions.append((chrom, nxt["ref_end"], nxt["ref_end"] + dev, "left_rev"))
                            else:
                                bnd(chrom, cur["ref_end"] - 1, "fwd", chrom, nxt["ref_end"] - 1, "rev")
                    else:                                                           # :198 rev → fwd
                        dev = d_read - (nxt["ref_start"] - cur["ref_start"])
                        if nxt["ref_start"] - cur["ref_end"] >= -o.reference_overlap_tolerance:
                            if o.min_sv_size <= -dev <= o.max_sv_size:
                                inversions.append((chrom, cur["ref_start"], cur["ref_start"] - dev, "right_fwd"))
                            else:
                                bnd(chrom, cur["ref_start"], "rev", chrom, nxt["ref_start"], "fwd")
                        elif cur["ref_start"] - nxt["ref_end"] >= -o.reference_overlap_tolerance:
                            if o.min_sv_size <= dev <= o.max_sv_size:
                                inversions.append((chrom, nxt["ref_start"], nxt["ref_start"] + dev, "right_rev"))
                            else:
                                bnd(chrom, cur["ref_start"], "rev", chrom, nxt["ref_start"], "fwd")
        else:                                                                        # :224
            c1, c2 = names[cur["ref_id"]], names[nxt["ref_id"]]
            if -o.query_overlap_tolerance <= d_read <= o.query_gap_tolerance:
                if cur["rev"] == nxt["rev"]:
                    if not cur["rev"]:
                        bnd(c1, cur["ref_end"] - 1, "fwd", c2, nxt["ref_start"], "fwd")
                    else:
                        bnd(c1, cur["ref_start"], "rev", c2, nxt["ref_end"] - 1, "rev")
                else:
                    if not cur["rev"]:
                        bnd(c1, cur["ref_end"] - 1, "fwd", c2, nxt["ref_end"] - 1, "rev")
                    else:
                        bnd(c1, cur["ref_start"], "rev", c2, nxt["ref_start"], "fwd")

    # tandem merge (:261-290); current_direction keeps the FIRST tuple's value (A3.7 quirk)
    cur_chr = None
    for td in tandems:
        if cur_chr is None:
            cur_chr, starts, ends, copies, fully, direction = td[0], [td[1]], [td[2]], 1, [td[3]], td[4]
        elif (cur_chr == td[0] and abs(_mean(starts) - td[1]) < 20 and abs(_mean(ends) - td[2]) < 20
              and direction == td[4]):
            starts.append(td[1]); ends.append(td[2]); copies += 1; fully.append(td[3])
        else:
            out.append(cand_tan(cur_chr, int(_mean(starts)), int(_mean(ends)), copies, any(fully), [qname], lens))
            cur_chr, starts, ends, copies, fully = td[0], [td[1]], [td[2]], 1, [td[3]]
    if cur_chr is not None:
        out.append(cand_tan(cur_chr, int(_mean(starts)), int(_mean(ends)), copies, any(fully), [qname], lens))

    # interspersed duplications from breakend pairs (:293-320)
    for ti in range(len(transl)):
        t_d1, t_d2, t_c1, t_p1, t_c2, t_p2 = transl[ti]
        for b_d1, b_d2, b_c1, b_p1, b_c2, b_p2 in transl[:ti]:
            if b_d1 == t_d2 and b_d2 == t_d1 and is_similar(b_c1, b_p1, 0, t_c2, t_p2, 0) \
                    and b_c2 == t_c1 and b_d2 == b_d1:
                if b_d1 == "fwd":
                    length = t_p1 + 1 - b_p2
                    if o.min_sv_size <= length <= o.max_sv_size:
                        m = int(_mean([b_p1 + 1, t_p2]))
                        out.append(cand_int(b_c2, b_p2, t_p1 + 1, b_c1, m, m + length, [qname], lens))
                elif b_d1 == "rev":
                    length = b_p2 + 1 - t_p1
                    if o.min_sv_size <= length <= o.max_sv_size:
                        m = int(_mean([b_p1, t_p2 + 1]))
                        out.append(cand_int(b_c2, t_p1, b_p2 + 1, b_c1, m, m + length, [qname], lens))

    # inversions (:323-338); the inversion that closes a group is DROPPED (A3.10 quirk)
    active = []
    for inv in sorted(inversions, key=lambda i: (i[0], i[1], i[2])):
        if not active:
            active.append(inv)
        elif inv[0] == active[-1][0] and inv[1] < max(i[2] for i in active):
            active.append(inv)
        else:
            out.extend(process_overlapping_inversions(active, qname, lens))
            active = []
    if active:
        out.extend(process_overlapping_inversions(active, qname, lens))
    return out


def postpass_records(raw, contig_rank, min_sv_size, max_sv_size):
    """Record-level restatement of the three post-passes (SVIM_inter.py:260-338) for ONE read: `raw` is
    the read's adjacency records as (kind, a0..a5) tuples in sorted-pair order (kinds: 3 BND, 4 TANDEM,
    5 INV — the layout of svx_raw in include/svx.h); contig_rank[ref_id] is the rank of the contig name
    under Python str ordering (the inversion sort key, :323).  Returns the derived records in the
    reference's order as tuples ("TANDEM", ref, start, end, copies, fully), ("DUP_INT", src_ref,
    src_start, src_end, dst_ref, dst_start, dst_end), ("INV", ref, start, end, complete)."""
    out = []
    tandems = [(r[1], r[2], r[3], bool(r[4]), bool(r[5])) for r in raw if r[0] == 4]
    transl = [(r[3], r[6], r[1], r[2], r[4], r[5]) for r in raw if r[0] == 3]   # (d1, d2, c1, p1, c2, p2)
    inversions = [(r[1], r[2], r[3], r[4]) for r in raw if r[0] == 5]           # (ref, start, end, side 0..3)
    cur_chr = None
    for td in tandems:
        if cur_chr is None:
            cur_chr, starts, ends, copies, fully, direction = td[0], [td[1]], [td[2]], 1, [td[3]], td[4]
        elif (cur_chr == td[0] and abs(_mean(starts) - td[1]) < 20 and abs(_mean(ends) - td[2]) < 20
              and direction == td[4]):
            starts.append(td[1]); ends.append(td[2]); copies += 1; fully.append(td[3])
        else:
            out.append(("TANDEM", cur_chr, int(_mean(starts)), int(_mean(ends)), copies, any(fully)))
            cur_chr, starts, ends, copies, fully = td[0], [td[1]], [td[2]], 1, [td[3]]
    if cur_chr is not None:
        out.append(("TANDEM", cur_chr, int(_mean(starts)), int(_mean(ends)), copies, any(fully)))
    for ti in range(len(transl)):
        t_d1, t_d2, t_c1, t_p1, t_c2, t_p2 = transl[ti]
        for b_d1, b_d2, b_c1, b_p1, b_c2, b_p2 in transl[:ti]:
            if b_d1 == t_d2 and b_d2 == t_d1 and is_similar(b_c1, b_p1, 0, t_c2, t_p2, 0) \
                    and b_c2 == t_c1 and b_d2 == b_d1:
                if b_d1 == 0:
                    length = t_p1 + 1 - b_p2
                    if min_sv_size <= length <= max_sv_size:
                        m = int(_mean([b_p1 + 1, t_p2]))
                        out.append(("DUP_INT", b_c2, b_p2, t_p1 + 1, b_c1, m, m + length))
                else:
                    length = b_p2 + 1 - t_p1
                    if min_sv_size <= length <= max_sv_size:
                        m = int(_mean([b_p1, t_p2 + 1]))
                        out.append(("DUP_INT", b_c2, t_p1, b_p2 + 1, b_c1, m, m + length))

    def flush(active):
        if len(active) < 2:
            clusters = [active]
        else:
            rows = [(i[1], i[2], 0 if i[3] < 2 else 1) for i in active]
            dist = [reciprocal_overlap_distance(rows[i], rows[j])
                    for i in range(len(rows) - 1) for j in range(i + 1, len(rows))]
            labels = _complete_linkage_labels(dist, len(rows), 0.3)
            clusters = [[] for _ in range(max(labels))]
            for idx, lab in enumerate(labels):
                clusters[lab - 1].append(active[idx])
        for cl in clusters:
            out.append(("INV", cl[0][0], max(i[1] for i in cl), min(i[2] for i in cl), len(cl) > 1))

    active = []
    for inv in sorted(inversions, key=lambda i: (contig_rank[i[0]], i[1], i[2])):
        if not active:
            active.append(inv)
        elif inv[0] == active[-1][0] and inv[1] < max(i[2] for i in active):
            active.append(inv)
        else:
            flush(active)
            active = []
    if active:
        flush(active)
    return out


# ----------------------------------------------------------------------------- a4
_CIG_RE = re.compile(r"(\d+)([MIDNSHP=XB])")


def retrieve_other_alignments(rec, names):
    """SVIM_COLLECT.py:8-58: SA tag → pseudo records (empty sequence, flag 2048/2064)."""
    if sum(l for o, l in rec["cigar"] if o == 5) > 0:   # :11
        return []
    if rec.get("sa") is None:                            # :13-16
        return []
    out = []
    for element in rec["sa"].split(";"):
        f = element.split(",")
        if len(f) != 6:                                  # :22-23
            continue
        rname, pos, strand, cigar, mapq, _nm = f[0], int(f[1]), f[2], f[3], int(f[4]), int(f[5])
        if mapq < 0 or mapq > 255:                       # OverflowError → 0 (:42-45)
            mapq = 0
        tuples = [("MIDNSHP=XB".index(c), int(n)) for n, c in _CIG_RE.findall(cigar)]
        if any(l >= (1 << 28) for _, l in tuples):       # OverflowError → entry skipped (:46-50)
            continue
        out.append(dict(qname=rec["qname"], flag=2048 if strand == "+" else 2064,
                        tid=names.index(rname) if rname in names else -1, pos=pos - 1, mapq=mapq,
                        cigar=tuples, seq="", sa=None))
    return out


def collect(records, names, lengths, o):
    """analyze_alignment_file_coordsorted (SVIM_COLLECT.py:61-83) over records in BAM order."""
    lens = dict(zip(names, lengths))
    out = []
    for tid in range(len(names)):                        # contigs in header order (:64)
        for rec in records:
            if rec["tid"] != tid:
                continue
            if rec["flag"] & 0x4 or rec["flag"] & 0x100 or rec["mapq"] < o.min_mapq:   # :71
                continue
            if rec["flag"] & 0x800:                      # supplementary: indels only (:73-74)
                out.extend(analyze_alignment_indel(rec, names, lens, o.min_sv_size))
            else:
                supp = [s for s in retrieve_other_alignments(rec, names)
                        if not (s["flag"] & 0x4) and s["mapq"] >= o.min_mapq]           # :77
                out.extend(analyze_alignment_indel(rec, names, lens, o.min_sv_size))
                out.extend(analyze_read_segments(rec, supp, names, lens, o))
    return out


# ----------------------------------------------------------------------------- a6
def form_partitions(with_hap, max_distance):
    """SVIM_COMBINE.py:15-32."""
    srt = sorted(with_hap, key=lambda e: get_key(e[1]))
    parts, cur = [], []
    for hap, c in srt:
        if cur:
            k, lk = get_key(c), get_key(cur[-1][1])
            if lk[0] != k[0] or lk[1] != k[1] or abs(lk[2] - k[2]) > max_distance:
                parts.append(cur)
                cur = []
        cur.append((hap, c))
    if cur:
        parts.append(cur)
    return parts


# ----------------------------------------------------------------------------- a7
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def edit_distance(a, b):
    """Global unit-cost Levenshtein (edlib.align default NW), two-row DP."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1))
        prev = cur
    return prev[-1]


def haplotype_strings(c1, c2, fetch, ref_lens):
    """The two strings compute_distance aligns (SVIM_COMBINE.py:43-100)."""
    t = c1[0]

    def up(ch, s, e):
        return fetch(ch, s, e).upper()

    if t in ("DEL", "INV", "DUP_TAN"):
        ch = c1[1]
        rs = max(0, min(c1[2], c2[2]) - 100)
        re_ = min(ref_lens[ch], max(c1[3], c2[3]) + 100)
        res = []
        for c in (c1, c2):
            if t == "DEL":
                mid = ""
            elif t == "INV":
                mid = "".join(_COMP.get(b, b) for b in reversed(up(ch, c[2], c[3])))
            else:
                mid = up(ch, c[2], c[3]) * (c[4] + 1)
            res.append(up(ch, rs, c[2]) + mid + up(ch, c[3], re_))
        return res
    if t == "INS":
        ch = c1[1]
        rs = max(0, min(c1[2], c2[2]) - 100)
        re_ = min(ref_lens[ch], max(c1[2], c2[2]) + 100)
        return [up(ch, rs, c[2]) + c[5] + up(ch, c[2], re_) for c in (c1, c2)]
    # DUP_INT
    ch = c1[4]
    rs = max(0, min(c1[5], c2[5]) - 100)
    re_ = min(ref_lens[ch], max(c1[5], c2[5]) + 100)
    return [up(ch, rs, c[5]) + up(c[1], c[2], c[3]) + up(ch, c[5], re_) for c in (c1, c2)]


def compute_distance(e1, e2, fetch, ref_lens, edit=edit_distance):
    """SVIM_COMBINE.py:35-102."""
    if e1[0] == e2[0]:
        return 1000000000
    h1, h2 = haplotype_strings(e1[1], e2[1], fetch, ref_lens)
    return edit(h1, h2)


def pair_haplotypes(partitions, fetch, ref_lens, threshold, edit=edit_distance):
    """SVIM_COMBINE.py:120-140."""
    clusters = []
    for part in partitions:
        if len(part) < 2:
            clusters.append(part)
            continue
        if len(part) > 10:                               # :126-128 dropped
            continue
        dist = [compute_distance(part[i], part[j], fetch, ref_lens, edit)
                for i in range(len(part) - 1) for j in range(i + 1, len(part))]
        labels = _complete_linkage_labels(dist, len(part), threshold)
        new = [[] for _ in range(max(labels))]
        for idx, lab in enumerate(labels):
            new[lab - 1].append(part[idx])
        clusters.extend(new)
    return clusters


def pair_haplotypes_breakends(partitions):
    """SVIM_COMBINE.py:143-161 with span_position_distance_breakends (:105-117)."""
    clusters = []
    for part in partitions:
        if len(part) < 2:
            clusters.append(part)
            continue
        if len(part) > 10:
            continue
        rows = [(h, c[2], c[3], c[5], c[6]) for h, c in part]
        dist = []
        for i in range(len(rows) - 1):
            for j in range(i + 1, len(rows)):
                a, b = rows[i], rows[j]
                if a[0] != b[0] and a[2] == b[2] and a[4] == b[4]:
                    dist.append((abs(a[1] - b[1]) + abs(a[3] - b[3])) / 3000)
                else:
                    dist.append(99999)
        labels = _complete_linkage_labels(dist, len(rows), 0.3)
        new = [[] for _ in range(max(labels))]
        for idx, lab in enumerate(labels):
            new[lab - 1].append(part[idx])
        clusters.extend(new)
    return clusters


# ----------------------------------------------------------------------------- a8
def pair_candidates(c1, c2, fetch, names, lengths, ref_lens, o, edit=edit_distance):
    """SVIM_COMBINE.py:164-366.  ref_lens: FASTA lengths (compute_distance uses the reference
    file's lengths, the ctors the BAM header's)."""
    lens = dict(zip(names, lengths))
    out = []
    for typ in ("DEL", "INV", "INS", "DUP_TAN", "DUP_INT", "BND"):
        both = [(1, c) for c in c1 if c[0] == typ] + [(2, c) for c in c2 if c[0] == typ]
        parts = form_partitions(both, o.partition_max_distance)
        if typ == "BND":
            clusters = pair_haplotypes_breakends(parts)
        else:
            clusters = pair_haplotypes(parts, fetch, ref_lens, o.max_edit_distance, edit)
        for cl in clusters:
            if len(cl) == 1:
                gt = "1/0" if cl[0][0] == 1 else "0/1"
                reads = cl[0][1][_reads_idx(typ)]
                extra = {}
            elif len(cl) == 2:
                gt = "1/1"
                reads = cl[0][1][_reads_idx(typ)] + cl[1][1][_reads_idx(typ)]
                extra = {"other": cl[1][1]}
            else:
                continue                                  # logged as an error, nothing emitted (:205)
            c = cl[0][1]
            other = extra.get("other")
            if typ == "DEL":
                out.append(cand_del(c[1], c[2], c[3], reads, lens, gt))
            elif typ == "INV":
                out.append(cand_inv(c[1], c[2], c[3], reads, c[5] or (other[5] if other else False), lens, gt))
            elif typ == "INS":
                out.append(cand_ins(c[1], c[2], c[3], reads, c[5], lens, gt))
            elif typ == "DUP_TAN":
                copies = c[4] if other is None else round(_py_mean([c[4], other[4]]))    # :290 banker's rounding
                out.append(cand_tan(c[1], c[2], c[3], copies, c[5] or (other[5] if other else False), reads, lens, gt))
            elif typ == "DUP_INT":
                out.append(cand_int(c[1], c[2], c[3], c[4], c[5], c[6], reads, lens,
                                    c[8] or (other[8] if other else False), gt))
            else:
                out.append(cand_bnd(c[1], c[2], c[3], c[4], c[5], c[6], reads, lens, gt))
    return out


def _py_mean(xs):
    from statistics import mean
    return mean(xs)


def _reads_idx(typ):
    return {"DEL": 4, "INS": 4, "INV": 4, "DUP_TAN": 6, "DUP_INT": 7, "BND": 7}[typ]


# ----------------------------------------------------------------------------- VCF text
def _vcf(chrom, pos, ref, alt, filt, info, fmt, sample):
    return "\t".join([chrom, str(pos), "PLACEHOLDERFORID", ref, alt, ".", filt, info, fmt, sample])


def vcf_entries(cands, fetch, o, types):
    """Entry list of write_final_vcf (SVIM_COMBINE.py:428-464) as ((contig,start,end), line, label)."""
    seq_alleles = not o.symbolic_alleles
    rn = o.query_names
    ent = []

    def reads_info(reads):
        return ";READS=" + ",".join(reads) if rn else ""

    def up(ch, s, e):
        return fetch(ch, s, e).upper()

    by = defaultdict(list)
    for c in cands:
        by[c[0]].append(c)
    if "DEL" in types:
        for c in by["DEL"]:
            _, ch, s, e, reads, gt = c
            ref, alt = (up(ch, max(0, s - 1), e), up(ch, max(0, s - 1), s)) if seq_alleles else ("N", "<DEL>")
            ent.append(((ch, max(1, s), e), _vcf(ch, max(1, s), ref, alt, "PASS",
                        "SVTYPE=DEL;END=%d;SVLEN=%d" % (e, s - e) + reads_info(reads), "GT", gt), "DEL"))
    if "INV" in types:
        for c in by["INV"]:
            _, ch, s, e, reads, complete, gt = c
            if seq_alleles:
                ref = up(ch, s, e)
                alt = "".join(_COMP.get(b, b) for b in reversed(ref))
            else:
                ref, alt = "N", "<INV>"
            ent.append(((ch, s + 1, e), _vcf(ch, s + 1, ref, alt, "PASS" if complete else "incomplete_inversion",
                        "SVTYPE=INV;END=%d" % e + reads_info(reads), "GT", gt), "INV"))
    if "INS" in types:
        for c in by["INS"]:
            _, ch, s, e, reads, seq, gt = c
            if seq_alleles:
                ref = up(ch, max(0, s - 1), s)
                alt = ref + seq
            else:
                ref, alt = "N", "<INS>"
            ent.append(((ch, max(1, s), e), _vcf(ch, max(1, s), ref, alt, "PASS",
                        "SVTYPE=INS;END=%d;SVLEN=%d" % (s, e - s) + reads_info(reads), "GT", gt), "INS"))
    for c in by["DUP_TAN"]:
        _, ch, s, e, copies, fully, reads, gt = c
        filt = "PASS" if fully else "not_fully_covered"
        if o.tandem_duplications_as_insertions:
            if "INS" in types:
                if seq_alleles:
                    ref = up(ch, s, e)
                    alt = ref * (copies + 1)
                else:
                    ref, alt = "N", "<INS>"
                ent.append(((ch, s + 1, e), _vcf(ch, s + 1, ref, alt, filt,
                            "SVTYPE=INS;END=%d;SVLEN=%d" % (e, (e - s) * copies) + reads_info(reads), "GT", gt), "INS"))
        elif "DUP:TANDEM" in types:
            ent.append(((ch, s + 1, e), _vcf(ch, s + 1, "N", "<DUP:TANDEM>", filt,
                        "SVTYPE=DUP:TANDEM;END=%d;SVLEN=%d" % (e, e - s) + reads_info(reads), "GT:CN",
                        "%s:%d" % (gt, copies + 1)), "DUP_TANDEM"))
    for c in by["DUP_INT"]:
        _, sc, ss, se, dc, ds, de, reads, cutpaste, gt = c
        cp = "CUTPASTE;" if cutpaste else ""
        if o.interspersed_duplications_as_insertions:
            if "INS" in types:
                if seq_alleles:
                    ref = up(dc, max(0, ds - 1), ds)
                    alt = ref + up(sc, ss, se)
                else:
                    ref, alt = "N", "<INS>"
                ent.append(((dc, max(1, ds), de), _vcf(dc, max(1, ds), ref, alt, "PASS",
                            "SVTYPE=INS;%sEND=%d;SVLEN=%d" % (cp, ds, de - ds) + reads_info(reads), "GT", gt), "INS"))
        elif "DUP:INT" in types:
            ent.append(((sc, ss + 1, se), _vcf(sc, ss + 1, "N", "<DUP:INT>", "PASS",
                        "SVTYPE=DUP:INT;%sEND=%d;SVLEN=%d" % (cp, se, se - ss) + reads_info(reads), "GT", gt), "DUP_INT"))
    if "BND" in types:
        for c in by["BND"]:
            _, sc, ss, sd, dc, ds, dd, reads, gt = c
            fwd_alt = {("fwd", "fwd"): "N[%s:%d[", ("fwd", "rev"): "N]%s:%d]",
                       ("rev", "rev"): "]%s:%d]N", ("rev", "fwd"): "[%s:%d[N"}[(sd, dd)] % (dc, ds + 1)
            rev_alt = {("rev", "rev"): "N[%s:%d[", ("fwd", "rev"): "N]%s:%d]",
                       ("fwd", "fwd"): "]%s:%d]N", ("rev", "fwd"): "[%s:%d[N"}[(sd, dd)] % (sc, ss + 1)
            info = "SVTYPE=BND" + reads_info(reads)
            ent.append(((sc, ss + 1, ss + 2), _vcf(sc, ss + 1, "N", fwd_alt, "PASS", info, "GT", gt), "BND"))
            ent.append(((dc, ds + 1, ds + 2), _vcf(dc, ds + 1, "N", rev_alt, "PASS", info, "GT", gt), "BND"))
    # write_final_vcf builds the list type by type in this order: DEL, INV, INS, DUP_TAN, DUP_INT, BND
    order = {"DEL": 0, "INV": 1, "INS": 2}
    return ent


def natural_key(entry):
    """sorted_nicely key (SVIM_COMBINE.py:369-376)."""
    conv = [int(t) if t.isdigit() else t for t in re.split("([0-9]+)", str(entry[0][0]))]
    return (conv, entry[0][1], entry[0][2])


def vcf_text(cands, fetch, names, lengths, o, version="1.0.3"):
    """write_final_vcf (SVIM_COMBINE.py:379-477) without the ##fileDate line."""
    types = [t.strip() for t in o.types.split(",")]
    L = ["##fileformat=VCFv4.2", "##source=SVIM-asm-v%s" % version]
    L += ["##contig=<ID=%s,length=%d>" % (n, l) for n, l in zip(names, lengths)]
    tan_dup = (not o.tandem_duplications_as_insertions) and "DUP:TANDEM" in types
    int_dup = (not o.interspersed_duplications_as_insertions) and "DUP:INT" in types
    if "DEL" in types:
        L.append('##ALT=<ID=DEL,Description="Deletion">')
    if "INV" in types:
        L.append('##ALT=<ID=INV,Description="Inversion">')
    if tan_dup or int_dup:
        L.append('##ALT=<ID=DUP,Description="Duplication">')
    if tan_dup:
        L.append('##ALT=<ID=DUP:TANDEM,Description="Tandem Duplication">')
    if int_dup:
        L.append('##ALT=<ID=DUP:INT,Description="Interspersed Duplication">')
    if "INS" in types:
        L.append('##ALT=<ID=INS,Description="Insertion">')
    if "BND" in types:
        L.append('##ALT=<ID=BND,Description="Breakend">')
    L.append('##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">')
    L.append('##INFO=<ID=CUTPASTE,Number=0,Type=Flag,Description="Genomic origin of interspersed duplication seems to be deleted">')
    L.append('##INFO=<ID=END,Number=1,Type=Integer,Description="End position of the variant described in this record">')
    L.append('##INFO=<ID=SVLEN,Number=1,Type=Integer,Description="Difference in length between REF and ALT alleles">')
    if o.query_names:
        L.append('##INFO=<ID=READS,Number=.,Type=String,Description="Names of all supporting reads">')
    L.append('##FILTER=<ID=not_fully_covered,Description="Tandem duplication is not fully covered by a contig">')
    L.append('##FILTER=<ID=incomplete_inversion,Description="Only one inversion breakpoint is supported">')
    L.append('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">')
    if tan_dup:
        L.append('##FORMAT=<ID=CN,Number=1,Type=Integer,Description="Copy number of tandem duplication (e.g. 2 for one additional copy)">')
    L.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + o.sample)
    counter = defaultdict(int)
    for src, line, label in sorted(vcf_entries(cands, fetch, o, types), key=natural_key):
        counter[label] += 1
        L.append(line.replace("PLACEHOLDERFORID", "svim_asm.%s.%d" % (label, counter[label]), 1))
    return "\n".join(L) + "\n"
