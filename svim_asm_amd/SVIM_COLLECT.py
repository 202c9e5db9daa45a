"""COLLECT: per-contig driver over a coordinate-sorted BAM, batched for the GPU and columnar on the host.

Mirrors retrieve_other_alignments(main_alignment, bam) (SVIM_COLLECT.py:8-58) and
analyze_alignment_file_coordsorted(bam, options) (SVIM_COLLECT.py:61-83).  The reference walks the
records in a Python `while True / next()` loop and builds one Candidate object per signature; here

  1. the records are COLUMNS (the native reader's arrays, or columns gathered from a duck-typed bam):
     the filters of :71 are one mask, the loop order (contigs in header order, file order inside) one
     stable argsort;
  2. the arithmetic of every record of the sample — or of both haplotype BAMs of a diploid sample — is
     ONE submission to the device (`Context.collect_batch`: CIGAR walk a1+a2 over the reader's CIGAR pool
     as it lies in page-locked memory, CIGAR statistics and segment rows of the chimeric reads, the
     split-segment decision tree and its post-passes a3), with one read-back;
  3. the signatures become rows of a CandidateTable (svim_asm_amd/table.py) with the constructors'
     clamping / normalisation (SVCandidate.py:40-46,83-89,130-136,181-187,266-280,352-373) done on the
     columns, in exactly the reference's order: per alignment its indels (CIGAR order), then its
     segment candidates.
`analyze_alignment_file_coordsorted` returns a CandidateList: the reference's list of objects,
materialised from the table when somebody looks.
"""
import logging
import os
import re
import threading
import time

import numpy as np

from svim_asm_amd import _lib
from svim_asm_amd import SVIM_inter, SVIM_intra
from svim_asm_amd.bamio import AlignedRecord, AlignmentFile
from svim_asm_amd.table import (CandidateList, CandidateTable, F_BOOL, F_DST_REV, F_SRC_REV, NamePool, PendingBytes, T_BND, T_DEL,
                                T_DUP_INT, T_DUP_TAN, T_INS, T_INV, _ranges)

_CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=XB])")
_CIGAR_CODE = {c: i for i, c in enumerate("MIDNSHP=XB")}


def _cigar_string_to_words(cigar):
    pairs = _CIGAR_RE.findall(cigar) if cigar else []
    lens = [int(n) for n, _ in pairs]
    if any(l >= (1 << 28) for l in lens):
        raise OverflowError("CIGAR operation length does not fit 28 bits")
    return np.array([(l << 4) | _CIGAR_CODE[c] for l, (_, c) in zip(lens, pairs)], dtype=np.uint32)


def _parse_sa(sa_string, get_tid):
    """(tid, reference_start, is_reverse, mapping_quality, cigar words, NM) of every usable SA entry
    (SVIM_COLLECT.py:19-57): entries without six fields are skipped, a mapping quality outside uint8 becomes
    0 (:42-45), an entry whose CIGAR overflows is logged and dropped (:46-50)."""
    out = []
    for element in sa_string.split(";"):
        fields = element.split(",")
        if len(fields) != 6:
            continue
        rname, pos, strand, cigar = fields[0], int(fields[1]), fields[2], fields[3]
        mapq, nm = int(fields[4]), int(fields[5])
        try:
            words = _cigar_string_to_words(cigar)
        except OverflowError:
            logging.error("OverflowError while retrieving supplementary CIGAR string. Read name: {0}, "
                          "Position: {1}, CIGAR: {2}".format(rname, pos, cigar))
            continue
        out.append((get_tid(rname), pos - 1, strand != "+", mapq if 0 <= mapq <= 255 else 0, words, nm))
    return out


def _has_hard_clip(words):
    return len(words) > 0 and bool((((words & 15) == 5) & ((words >> 4) > 0)).any())


def retrieve_other_alignments(main_alignment, bam):
    """Reconstruct other alignments of the same read for a given alignment from the SA tag"""
    # reconstruction from the SA tag does not work if the main alignment is hard-clipped
    words = getattr(main_alignment, "cigar_words", None)
    if words is not None:
        if _has_hard_clip(words):
            return []
    elif main_alignment.get_cigar_stats()[0][5] > 0:
        return []
    try:
        sa_tag = main_alignment.get_tag("SA")
    except KeyError:
        return []
    other_alignments = []
    for tid, pos, rev, mapq, cigar_words, nm in _parse_sa(sa_tag, bam.get_tid):
        a = AlignedRecord()
        a.query_name = main_alignment.query_name
        a.flag = 2064 if rev else 2048
        a.reference_id = tid
        a.reference_start = pos
        a.mapping_quality = mapq
        a.cigar_words = cigar_words
        a._tags = {"NM": nm}
        other_alignments.append(a)
    return other_alignments


# ------------------------------------------------------------------------------ records as columns
class _Records(object):
    """The records of one bam COLLECT walks, as columns in FILE order, plus the loop order `order`
    (indices into the columns: contigs in bam.references order, file order inside, :64-65)."""

    def __init__(self):
        self.tid = self.pos = self.flag = self.mapq = self.l_seq = None
        self.cig_off = None      # int64 [n + 1] into `cigar`
        self.cigar = None        # uint32 pool
        self.order = None
        self.n_ref = 0

    def words(self, i):
        return self.cigar[self.cig_off[i]:self.cig_off[i + 1]]


class _NativeRecords(_Records):
    def __init__(self, bam):
        _Records.__init__(self)
        base = getattr(bam, "_bam", bam)
        self.base = base
        tids = [base.get_tid(name) for name in bam.references]
        for name in bam.references:
            logging.info("Processing chromosome {0}...".format(name))
        if base._loaded is None:
            base.load(None)
        elif base._loaded != "all" and not set(t for t in tids if t >= 0) <= set(base._loaded):
            base.load(set(base._loaded) | set(t for t in tids if t >= 0))
        c = base._cols
        self.tid, self.pos, self.flag, self.mapq, self.l_seq = c["tid"], c["pos"], c["flag"], c["mapq"], c["l_seq"]
        self.cig_off, self.cigar = base._cig_off, base._cigar
        self.n_ref = len(base.references)
        rank = np.full(self.n_ref + 1, -1, np.int64)  # slot -1: unplaced records
        for k, t in reversed(list(enumerate(tids))):
            if t >= 0:
                rank[t] = k
        rec_rank = rank[np.where((self.tid >= 0) & (self.tid < self.n_ref), self.tid, -1)]
        sel = np.flatnonzero(rec_rank >= 0)
        r = rec_rank[sel]
        if len(r) > 1 and bool((r[1:] < r[:-1]).any()):
            sel = sel[np.argsort(r, kind="stable")]
        self.order = sel

    def names(self):
        return NamePool(self.base._names_pool, self.base._name_off)

    def name_index(self, i):
        return i

    def sa_string(self, i):
        so = int(self.base._sa_off[i])
        if so < 0:
            return None
        return self.base._aux_pool[so:so + int(self.base._sa_len[i])].decode()

    def query_end_from_sequence(self, i):
        """pysam's query_alignment_end of a record with a stored sequence: l_seq minus the trailing soft
        clips (hard clips skipped); -1 without one (the device derives it from the CIGAR)."""
        l_seq = int(self.l_seq[i])
        if l_seq == 0:
            return -1
        w = self.words(i)
        end = l_seq
        for k in range(len(w) - 1, 0, -1):
            o = int(w[k]) & 15
            if o == 5:
                continue
            if o == 4:
                end -= int(w[k]) >> 4
            else:
                break
        return end

    def sequence_slices(self, rec, lo, hi):
        return self.base.sequence_slices_raw(rec, lo, hi)


class _GenericRecords(_Records):
    """Columns gathered from any bam with the pysam surface of SURVEY.md Appendix B (tests' FakeBam, a real
    pysam.AlignmentFile)."""

    def __init__(self, bam):
        _Records.__init__(self)
        recs = []
        for contig in bam.references:
            logging.info("Processing chromosome {0}...".format(contig))
            recs.extend(bam.fetch(contig=contig))
        self.recs = recs
        n = len(recs)
        self.n_ref = len(bam.references)
        self.tid = np.array([a.reference_id for a in recs], dtype=np.int64).reshape(n)
        self.pos = np.array([a.reference_start for a in recs], dtype=np.int64).reshape(n)
        self.mapq = np.array([a.mapping_quality for a in recs], dtype=np.int64).reshape(n)
        self.flag = np.array([(4 if a.is_unmapped else 0) | (0x100 if a.is_secondary else 0) |
                              (0x800 if a.is_supplementary else 0) | (0x10 if a.is_reverse else 0) for a in recs],
                             dtype=np.int64).reshape(n)
        words = [SVIM_intra.cigar_words_of(a) for a in recs]
        self.cig_off = np.zeros(n + 1, np.int64)
        if n:
            np.cumsum([len(w) for w in words], out=self.cig_off[1:])
        self.cigar = np.concatenate(words).astype(np.uint32, copy=False) if n and self.cig_off[-1] else np.zeros(0, np.uint32)
        self.l_seq = np.array([a._l_seq if hasattr(a, "_l_seq") else len(a.query_sequence or "") for a in recs],
                              dtype=np.int64).reshape(n)
        self.order = np.arange(n, dtype=np.int64)

    def names(self):
        return NamePool.from_strings([a.query_name for a in self.recs])

    def sa_string(self, i):
        try:
            return self.recs[i].get_tag("SA")
        except KeyError:
            return None

    def query_end_from_sequence(self, i):
        a = self.recs[i]
        if getattr(a, "_l_seq", None) == 0:
            return -1  # a product record without a stored sequence: from the CIGAR, on the device
        return int(a.query_alignment_end)

    def sequence_slices(self, rec, lo, hi):
        parts = []
        for i, a, b in zip(np.asarray(rec).tolist(), np.asarray(lo).tolist(), np.asarray(hi).tolist()):
            r = self.recs[i]
            f = getattr(r, "seq_slice", None)
            s = f(a, b) if f is not None else (r.query_sequence or "")[a:b]
            parts.append(s.encode("latin-1"))
        off = np.zeros(len(parts) + 1, np.int64)
        if parts:
            np.cumsum([len(p) for p in parts], out=off[1:])
        return np.frombuffer(b"".join(parts), dtype=np.uint8) if parts else np.zeros(0, np.uint8), off


def _records_of(bam):
    base = getattr(bam, "_bam", bam)
    if isinstance(base, AlignmentFile) and base._h is not None:
        return _NativeRecords(bam)
    return _GenericRecords(bam)


# ------------------------------------------------------------------------------ the batch
class _Sample(object):
    """Host-side state of one bam between the submission and the table."""
    pass


def _prepare(bam, options):
    """Filters, loop order and the chimeric reads of one bam (everything the submission needs)."""
    s = _Sample()
    s.bam = bam
    r = s.rec = _records_of(bam)
    flag = r.flag
    keep = ((flag & 0x4) == 0) & ((flag & 0x100) == 0) & (r.mapq >= options.min_mapq)  # SVIM_COLLECT.py:71
    s.order = r.order[keep[r.order]]
    n = len(r.tid)
    s.place = np.full(n, -1, np.int64)       # record -> its place in the loop (-1: never visited)
    s.place[s.order] = np.arange(len(s.order))
    # chimeric reads: kept primaries whose SA tag yields segments that pass the filters of :77
    s.read_primary, seg_src, seg_tid, seg_pos, seg_rev, seg_qend, counts = [], [], [], [], [], [], []
    s.extra_words = []
    get_tid = bam.get_tid
    has_sa = getattr(getattr(r, "base", None), "_sa_off", None)
    primaries = s.order[(flag[s.order] & 0x800) == 0]
    if has_sa is not None:
        primaries = primaries[has_sa[primaries] >= 0]
    n_ref = r.n_ref
    for i in primaries.tolist():
        sa = r.sa_string(i)
        if sa is None or _has_hard_clip(r.words(i)):
            continue
        good = [e for e in _parse_sa(sa, get_tid) if e[3] >= options.min_mapq]
        if not good:
            continue
        for e in good:
            if e[0] < 0 or e[0] >= n_ref:
                # the reference looks every segment's contig up by id (SVIM_inter.py:99,225-226)
                raise ValueError("reference_id %i out of range 0<=tid<%i" % (e[0], n_ref))
        s.read_primary.append(i)
        counts.append(1 + len(good))
        seg_src.append(i); seg_tid.append(int(r.tid[i])); seg_pos.append(int(r.pos[i]))
        seg_rev.append(1 if int(flag[i]) & 0x10 else 0); seg_qend.append(r.query_end_from_sequence(i))
        for tid, pos, rev, _mapq, words, _nm in good:
            seg_src.append(-1 - len(s.extra_words))  # resolved to n_aln + index once the batch is laid out
            s.extra_words.append(words)
            seg_tid.append(tid); seg_pos.append(pos); seg_rev.append(1 if rev else 0); seg_qend.append(-1)
    s.seg_src = np.asarray(seg_src, dtype=np.int64)
    s.seg_tid = np.asarray(seg_tid, dtype=np.int32)
    s.seg_pos = np.asarray(seg_pos, dtype=np.int32)
    s.seg_rev = np.asarray(seg_rev, dtype=np.uint8)
    s.seg_qend = np.asarray(seg_qend, dtype=np.int32)
    s.seg_count = np.asarray(counts, dtype=np.int64)
    return s


def _submit(samples, options, ctx):
    """ONE device submission for all `samples` (their headers agree): fills s.sig / s.raw / s.post / s.post_first
    with each sample's share of the results (record and segment indices local to the sample)."""
    parts, part_dev, aln_base, ops_base, at_aln, at_ops = [], [], [], [], 0, 0
    for s in samples:
        aln_base.append(at_aln)
        ops_base.append(at_ops)
        parts.append(s.rec.cigar)
        # the reader's own upload of its pool (svx_bam_device_pool): under way since the walk, not part of the submission
        base = getattr(s.rec, "base", None)
        pool = base.device_pool() if base is not None and hasattr(base, "device_pool") and s.rec.cigar is base._cigar else None
        part_dev.append(pool)
        at_aln += len(s.rec.tid)
        at_ops += int(s.rec.cig_off[-1])
    n_aln = at_aln
    aln_off = np.concatenate([s.rec.cig_off[:-1] + b for s, b in zip(samples, ops_base)] + [[at_ops]]).astype(np.uint64)
    ref_start = np.concatenate([s.rec.pos for s in samples]).astype(np.int32)
    extra, extra_len, seg_src, x_at = [], [], [], 0
    for s, b in zip(samples, aln_base):
        src = s.seg_src.copy()
        src[src >= 0] += b
        src[src < 0] = n_aln + x_at + (-1 - src[src < 0])
        seg_src.append(src)
        extra += s.extra_words
        x_at += len(s.extra_words)
    extra_len = [len(w) for w in extra]
    extra_off = np.zeros(len(extra) + 1, np.uint64)
    if extra:
        np.cumsum(extra_len, out=extra_off[1:])
    extra_cigar = np.concatenate(extra).astype(np.uint32, copy=False) if extra and extra_off[-1] else np.zeros(0, np.uint32)
    seg_src = np.concatenate(seg_src).astype(np.uint32) if seg_src else np.zeros(0, np.uint32)
    seg_count = np.concatenate([s.seg_count for s in samples])
    read_off = np.zeros(len(seg_count) + 1, np.uint32)
    np.cumsum(seg_count, out=read_off[1:])
    cat = lambda k, dt: np.concatenate([getattr(s, k) for s in samples]).astype(dt, copy=False)
    t_call = time.perf_counter()
    sig, raw, post, post_first = ctx.collect_batch(
        parts, aln_off, ref_start, options.min_sv_size, extra_cigar, extra_off, seg_src, cat("seg_tid", np.int32),
        cat("seg_pos", np.int32), cat("seg_rev", np.uint8), cat("seg_qend", np.int32), read_off,
        SVIM_inter.contig_ranks(getattr(samples[0].bam, "_bam", samples[0].bam)), SVIM_inter.seg_params(options),
        part_dev=part_dev)
    # (seconds inside svx_collect_batch: uploads — none for pools the reader put into HBM —, kernels, read-backs)
    LAST_TIMING["collect_call_s"] = LAST_TIMING.get("collect_call_s", 0.0) + time.perf_counter() - t_call
    # split the results by sample
    sig_aln = sig["aln"].astype(np.int64)
    cut = np.searchsorted(sig_aln, aln_base + [n_aln], side="left")
    r_at, g_at = 0, 0
    for k, s in enumerate(samples):
        lo, hi = int(cut[k]), int(cut[k + 1])
        s.sig = {key: v[lo:hi] for key, v in sig.items()}
        s.sig_aln = sig_aln[lo:hi] - aln_base[k]
        nr, ng = len(s.seg_count), int(s.seg_count.sum())
        s.raw = raw[g_at:g_at + ng]
        first = np.asarray(post_first[r_at:r_at + nr + 1], dtype=np.int64)
        s.post = post[int(first[0]):int(first[-1])] if nr else post[:0]
        s.post_first = first - (first[0] if nr else 0)
        r_at += nr
        g_at += ng


_TABLES_SIDE_BY_SIDE_FROM = 200000  # signatures of a submission from which its samples' tables are built on threads


def _python_slice(a, b, n):
    """[lo, hi) that s[a:b] selects from a sequence of length n (negative indices wrap, as in the reference's
    primary.query_sequence[...] slices, SVIM_inter.py:117,120)."""
    lo, hi, _ = slice(a, b).indices(n)
    return lo, max(lo, hi)


def _table_of(s, options):
    """Rows of one sample in the reference's order, constructors applied on the columns."""
    r, bam = s.rec, s.bam
    base = getattr(bam, "_bam", bam)
    contigs = list(base.references)
    contig_len = np.array([base.get_reference_length(c) for c in contigs], dtype=np.int64)
    n_ref = len(contigs)
    sig = s.sig
    # ---- indels of the visited records (a1 + a2; SVIM_intra.py:38-43)
    vis = s.place[s.sig_aln] >= 0
    i_aln = s.sig_aln[vis]
    i_start = sig["ref_pos"][vis].astype(np.int64)
    i_len = sig["len"][vis].astype(np.int64)
    i_type = sig["type"][vis]
    i_rpos = sig["read_pos"][vis].astype(np.int64)
    i_place = s.place[i_aln]
    if len(i_place) > 1 and bool((i_place[1:] < i_place[:-1]).any()):
        o = np.argsort(i_place, kind="stable")
        i_aln, i_start, i_len, i_type, i_rpos, i_place = i_aln[o], i_start[o], i_len[o], i_type[o], i_rpos[o], i_place[o]
    n_i = len(i_aln)
    # ---- candidates of the chimeric reads (a3): raw INS / DEL / BND in slot order, then the derived records
    raw, post = s.raw, s.post
    n_reads = len(s.seg_count)
    g_read = np.repeat(np.arange(n_reads), s.seg_count)
    rk = raw["kind"] if len(raw) else np.zeros(0, np.int32)
    rsel = np.flatnonzero((rk == _lib.RAW_INS) | (rk == _lib.RAW_DEL) | (rk == _lib.RAW_BND))
    p_read = np.repeat(np.arange(n_reads), np.diff(s.post_first)) if n_reads else np.zeros(0, np.int64)
    n_r, n_p = len(rsel), len(post)
    n_s = n_r + n_p
    t = CandidateTable(contigs, contig_len, n_i + n_s)
    t.names = r.names()
    # indel rows
    is_del = i_type == _lib.SIG_DEL
    tid_i = r.tid[i_aln]
    t.type[:n_i] = np.where(is_del, T_DEL, T_INS)
    end_i = i_start + i_len
    t.sc[:n_i] = np.where(is_del, tid_i, -1)
    t.ss[:n_i] = np.where(is_del, i_start, 0)
    t.se[:n_i] = np.where(is_del, end_i, 0)
    t.dc[:n_i] = np.where(is_del, -1, tid_i)
    t.ds[:n_i] = np.where(is_del, 0, i_start)
    t.de[:n_i] = np.where(is_del, 0, end_i)
    read_rec = np.empty(n_i + n_s, np.int64)   # the record whose name is the row's read
    read_rec[:n_i] = i_aln
    seq_rec = np.full(n_i + n_s, -1, np.int64)  # rows with an inserted sequence: record, [lo, hi) of its bases
    seq_lo = np.zeros(n_i + n_s, np.int64)
    seq_hi = np.zeros(n_i + n_s, np.int64)
    ins_i = np.flatnonzero(~is_del)
    seq_rec[ins_i] = i_aln[ins_i]
    seq_lo[ins_i] = i_rpos[ins_i]
    seq_hi[ins_i] = i_rpos[ins_i] + i_len[ins_i]
    # segment rows
    if n_s:
        prim = np.asarray(s.read_primary, dtype=np.int64)
        a = {k: np.concatenate((raw[k][rsel], post[k])).astype(np.int64) for k in ("a0", "a1", "a2", "a3", "a4", "a5")}
        kind = np.concatenate((raw["kind"][rsel], post["kind"] + 100)).astype(np.int64)
        s_read = np.concatenate((g_read[rsel], p_read))
        q = slice(n_i, n_i + n_s)
        read_rec[q] = prim[s_read]
        is_ins, is_dl, is_bnd = kind == _lib.RAW_INS, kind == _lib.RAW_DEL, kind == _lib.RAW_BND
        is_tan, is_dint, is_inv = kind == 100 + _lib.POST_TANDEM, kind == 100 + _lib.POST_DUP_INT, kind == 100 + _lib.POST_INV
        t.type[q] = np.select([is_ins, is_dl, is_bnd, is_tan, is_dint, is_inv], [T_INS, T_DEL, T_BND, T_DUP_TAN, T_DUP_INT, T_INV])
        src_like = is_dl | is_tan | is_dint | is_inv | is_bnd
        t.sc[q] = np.where(src_like, a["a0"], -1)
        t.ss[q] = np.where(src_like, a["a1"], 0)
        t.se[q] = np.where(is_dl | is_tan | is_dint | is_inv, a["a2"], 0)
        t.dc[q] = np.where(is_ins, a["a0"], np.where(is_dint | is_bnd, a["a3"], -1))
        t.ds[q] = np.where(is_ins, a["a1"], np.where(is_dint | is_bnd, a["a4"], 0))
        t.de[q] = np.where(is_ins, a["a2"], np.where(is_dint, a["a5"], 0))
        t.copies[q] = np.where(is_tan, a["a3"], 0)
        t.flag[q] = np.where(is_tan, (a["a4"] != 0) * F_BOOL, 0) | np.where(is_inv, (a["a3"] != 0) * F_BOOL, 0) | \
            np.where(is_bnd, (a["a2"] != 0) * F_SRC_REV + (a["a5"] != 0) * F_DST_REV, 0)
        for j in np.flatnonzero(is_ins).tolist():
            p = int(prim[s_read[j]])
            lo, hi = _python_slice(int(a["a3"][j]), int(a["a3"][j] + a["a4"][j]), int(r.l_seq[p]))
            seq_rec[n_i + j], seq_lo[n_i + j], seq_hi[n_i + j] = p, lo, hi
        # the contig ids of the segment rows come from SA tags: the reference looks each up (get_reference_name)
        for col in (t.sc[q], t.dc[q]):
            bad = col[(col >= n_ref) | (col < -1)]
            if len(bad):
                raise ValueError("reference_id %i out of range 0<=tid<%i" % (int(bad[0]), n_ref))
    # ---- reference order: per visited alignment its indels, then (primaries of chimeric reads) its segment rows
    if n_s:
        key = np.concatenate((i_place * 2, s.place[read_rec[n_i:]] * 2 + 1))
        o = np.argsort(key, kind="stable")
        for k in ("type", "sc", "ss", "se", "dc", "ds", "de", "flag", "copies"):
            setattr(t, k, getattr(t, k)[o])
        read_rec, seq_rec, seq_lo, seq_hi = read_rec[o], seq_rec[o], seq_lo[o], seq_hi[o]
    n = n_i + n_s
    t.r_off = np.arange(n + 1, dtype=np.int64)
    t.r_flat = read_rec
    t.rec_tid = r.tid[read_rec]  # the contig each row was collected under: merge key of a contig-sharded run
    # ---- the inserted sequences: decoded in one batch (native reader: only the BGZF members that hold them)
    rows = np.flatnonzero(seq_rec >= 0)
    s.seq_rows, s.seq_job = rows, None
    if len(rows):
        l_seq = r.l_seq[seq_rec[rows]]
        lo = np.minimum(np.maximum(seq_lo[rows], 0), l_seq)
        hi = np.maximum(np.minimum(seq_hi[rows], l_seq), lo)
        s.seq_lo, s.seq_hi = lo, hi
        box = {}

        def decode():
            try:
                box["out"] = r.sequence_slices(seq_rec[rows], lo, hi)
            except BaseException as e:  # noqa: BLE001 — re-raised on the calling thread
                box["error"] = e
        s.seq_job = (threading.Thread(target=decode, daemon=True), box)
        s.seq_job[0].start()
    # ---- the constructors, on the columns
    _apply_constructors(t, n_ref)
    s.table = t
    return t


def _pending_sequences(s):
    """The table's sequence pool while the sample's decoding (started by _table_of) is still running: offsets and
    lengths are known from the requests, the bytes are waited for when somebody reads `table.seqs`."""
    t = s.table
    if s.seq_job is not None:
        job, box = s.seq_job
        cnt = s.seq_hi - s.seq_lo
        off = np.zeros(len(cnt) + 1, np.int64)
        np.cumsum(cnt, out=off[1:])
        t.q_off[s.seq_rows] = off[:-1]
        t.q_len[s.seq_rows] = cnt

        def resolve():
            t0 = time.perf_counter()
            job.join()
            LAST_TIMING["sequences_wait_s"] = LAST_TIMING.get("sequences_wait_s", 0.0) + time.perf_counter() - t0
            if "error" in box:
                raise box["error"]
            _verify_pending(s)  # (nothing left when the call's device leg took the walks' members along)
            pool, got_off = box["out"]
            if not np.array_equal(got_off, off):
                raise ValueError("the reader returned other slice lengths than were asked for")
            return pool
        t.seqs = PendingBytes(int(off[-1]), resolve)
    else:
        _verify_pending(s)  # (no sequence-slice call for this reader: its threads check what the walks left)
    return t


def _verify_pending(s):
    """bamio.AlignmentFile.defer_verify: the members this sample's record walks took bytes from without checking them, if no
    device leg has checked them meanwhile (only ever called when no sequence-slice call of the reader is in flight)."""
    base = getattr(getattr(s, "rec", None), "base", None)
    if isinstance(base, AlignmentFile):
        base.verify_pending()


_WHAT = {T_DEL: "Deletion", T_INV: "Inversion", T_DUP_TAN: "Tandem duplication"}


def _apply_constructors(t, n_ref=None):
    """What the Candidate constructors do to their arguments, on all rows: `assert end >= start`, then
    start = max(0, start), end = min(contig length, end) (SVCandidate.py:40-46,83-89,130-136,181-187,266-280);
    breakends: endpoints ordered by (contig NAME, position) with both directions flipped when swapped, each
    position clamped to [0, contig length] (:352-373)."""
    typ = t.type
    length = np.concatenate((t.contig_len, [0]))  # id -1 -> slot -1
    src = (typ == T_DEL) | (typ == T_INV) | (typ == T_DUP_TAN) | (typ == T_DUP_INT)
    dst = (typ == T_INS) | (typ == T_DUP_INT)
    bad = np.flatnonzero((src & (t.se < t.ss)) | (dst & (t.de < t.ds)))
    if len(bad):
        i = int(bad[0])
        ty = int(typ[i])
        reads = t.names.strings(t.r_flat[t.r_off[i]:t.r_off[i + 1]].tolist())
        if ty == T_INS or (ty == T_DUP_INT and not t.se[i] < t.ss[i]):
            what = "Insertion" if ty == T_INS else "Interspersed duplication destination"
            c, a, b = t.contigs[t.dc[i]], t.de[i], t.ds[i]
        else:
            what = _WHAT.get(ty, "Interspersed duplication source")
            c, a, b = t.contigs[t.sc[i]], t.se[i], t.ss[i]
        raise AssertionError("{0} end ({1}:{2}) is smaller than its start ({1}:{3}). From read {4}".format(
            what, c, a, b, reads))
    unknown = (src & (length[t.sc] < 0)) | (dst & (length[t.dc] < 0))
    is_bnd = typ == T_BND
    unknown |= is_bnd & ((length[t.sc] < 0) | (length[t.dc] < 0))
    if bool(unknown.any()):
        i = int(np.flatnonzero(unknown)[0])
        for cid in (int(t.sc[i]), int(t.dc[i])):
            if cid >= 0 and t.contig_len[cid] < 0:
                raise KeyError("unknown reference %s" % t.contigs[cid])
    t.ss = np.where(src, np.maximum(t.ss, 0), t.ss)
    t.se = np.where(src, np.minimum(t.se, length[t.sc]), t.se)
    t.ds = np.where(dst, np.maximum(t.ds, 0), t.ds)
    t.de = np.where(dst, np.minimum(t.de, length[t.dc]), t.de)
    b = np.flatnonzero(is_bnd)
    if len(b):
        # contig NAMES compare as Python str (:352): rank of every name under that order
        names = t.contigs
        rank = np.empty(len(names) + 1, np.int64)
        uniq = {name: k for k, name in enumerate(sorted(set(names)))}
        rank[:len(names)] = [uniq[name] for name in names]
        rank[-1] = -1
        sc, ss, dc, ds, fl = t.sc[b], t.ss[b], t.dc[b], t.ds[b], t.flag[b]
        keep = (rank[sc] < rank[dc]) | ((rank[sc] == rank[dc]) & (ss < ds))
        swap = ~keep
        s_rev, d_rev = (fl & F_SRC_REV) != 0, (fl & F_DST_REV) != 0
        n_sc, n_ss = np.where(swap, dc, sc), np.where(swap, ds, ss)
        n_dc, n_ds = np.where(swap, sc, dc), np.where(swap, ss, ds)
        n_srev = np.where(swap, ~d_rev, s_rev)   # swapped endpoints read the junction from the other side
        n_drev = np.where(swap, ~s_rev, d_rev)
        t.sc[b], t.dc[b] = n_sc, n_dc
        t.ss[b] = np.minimum(length[n_sc], np.maximum(n_ss, 0))
        t.ds[b] = np.minimum(length[n_dc], np.maximum(n_ds, 0))
        t.flag[b] = (fl & ~np.uint8(F_SRC_REV | F_DST_REV)) | (n_srev * F_SRC_REV + n_drev * F_DST_REV).astype(np.uint8)


def _same_header(bams):
    first = getattr(bams[0], "_bam", bams[0])
    for b in bams[1:]:
        b = getattr(b, "_bam", b)
        if tuple(b.references) != tuple(first.references) or tuple(b.lengths) != tuple(first.lengths):
            return False
    return True


def _load_together(bams):
    """Index the records of all native bams at the same time (each walk runs on the reader's own threads and
    releases the GIL); a contig view loads the contigs it names."""
    jobs = []
    for bam in bams:
        base = getattr(bam, "_bam", bam)
        if not (isinstance(base, AlignmentFile) and base._h is not None):
            continue
        if base._loaded is None:
            want = None if base is bam else list(bam.references)
            jobs.append((base, want))
    if len(jobs) < 2:
        return
    errors = []

    def run(base, want):
        try:
            base.load(want)
        except BaseException as e:  # noqa: BLE001 — re-raised below
            errors.append(e)
    threads = [threading.Thread(target=run, args=j, daemon=True) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if errors:
        raise errors[0]


LAST_TIMING = {}  # seconds per stage of the latest collect_tables call (tools/, bench legs)


MAX_OPS_PER_SUBMISSION = (1 << 32) - 1024  # svx_collect_batch takes fewer than 2^32 CIGAR ops per call (include/svx.h)


def _submission_groups(samples, same_header, max_ops=None):
    """Which samples go out together: all of them in ONE submission when their reference dictionaries agree — also
    the BAMs of many samples of a cohort —, cut into several wherever the op count of a submission (records + SA-derived
    segments) would reach the device entry's limit; one submission per file otherwise."""
    max_ops = MAX_OPS_PER_SUBMISSION if max_ops is None else max_ops
    if not same_header:
        return [[s] for s in samples]
    groups, ops = [[]], 0
    for s in samples:
        n = int(s.rec.cig_off[-1]) + int(sum(len(w) for w in s.extra_words))
        if groups[-1] and ops + n > max_ops:
            groups.append([])
            ops = 0
        groups[-1].append(s)
        ops += n
    return groups


def collect_tables(bams, options, ctx=None):
    """CandidateTable of every bam in `bams` (the two haplotypes of a diploid sample, or the BAMs of a whole cohort
    of samples: svim_asm_amd/cohort.py): one device submission for all of them when their reference dictionaries
    agree, several where one would exceed the 2^32-op limit of svx_collect_batch."""
    tl, cl = time.perf_counter(), time.process_time()
    LAST_TIMING.pop("collect_call_s", None)
    _load_together(bams)
    t0, c0 = time.perf_counter(), time.process_time()
    samples = [_prepare(bam, options) for bam in bams]
    t1 = time.perf_counter()
    # the device context only now: in a fresh process it is still coming up (150-270 ms) while the record walks and the
    # host-side preparation above run — asking for it first made the walks wait for it (round 6: 45 ms of the command)
    ctx = ctx or _lib.default_context(getattr(options, "device", 0) or 0)
    LAST_TIMING["context_wait_s"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    groups = _submission_groups(samples, _same_header(bams))
    for group in groups:
        _submit(group, options, ctx)
    t2 = time.perf_counter()
    # (every sample's table starts its sequence decoding — the readers' threads — before anybody waits for any; the
    #  tables of a crowded pair of haplotypes are ~10 ms of array arithmetic each: side by side)
    if len(samples) > 1 and sum(len(s.sig_aln) for s in samples) > _TABLES_SIDE_BY_SIDE_FROM:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(4, len(samples))) as ex:
            list(ex.map(lambda s: _table_of(s, options), samples))
    else:
        for s in samples:
            _table_of(s, options)
    t3, c3 = time.perf_counter(), time.process_time()
    # (…_cpu_s: CPU seconds of all threads of the process — what a run costs under a CPU quota)
    LAST_TIMING.update(load_s=t0 - tl, prepare_s=t1 - t0, submit_s=t2 - t1, tables_s=t3 - t2, sequences_wait_s=0.0,
                       load_cpu_s=c0 - cl, submit_and_tables_cpu_s=c3 - c0)
    tables = [_pending_sequences(s) for s in samples]
    # The pool could stay pending until PAIR builds its byte pool (SVX_LAZY_SEQS=1: the decoding then runs beside
    # PAIR's keys, sort, windows and recipes).  Measured, interleaved on one box, four pairs of five runs: medians
    # 0.389-0.397 s pending vs 0.371-0.397 s waiting here — the host threads of the two sides compete for the same
    # cores, the overlap buys nothing (profiles/README.md) — so the default waits here.
    if not os.environ.get("SVX_LAZY_SEQS"):
        for t in tables:
            t.seqs
        LAST_TIMING["sequences_cpu_s"] = time.process_time() - c3
    return tables


def analyze_alignment_file_coordsorted(bam, options):
    return CandidateList(collect_tables([bam], options)[0])
