"""PAIR (diploid) + VCF output.

Mirrors form_partitions (SVIM_COMBINE.py:15-32), compute_distance (:35-102),
span_position_distance_breakends (:105-117), pair_haplotypes (:120-140),
pair_haplotypes_breakends (:143-161), pair_candidates (:164-366), sorted_nicely (:369-376) and
write_final_vcf (:379-477).

GPU work: the stable sort by Candidate.get_key() + partition sweep is svx_pair_partition (one
launch for all six SV types: the type is the most significant key field, so partitions and
their order are the ones the reference gets type by type); the pairwise haplotype edit
distances (edlib in the reference) are svx_edit_distance_batch, one launch for all pairs.
Complete linkage + flat cut of the ≤10-member partitions (scipy in the reference) is
svx_linkage_cut_batch, one launch for all partitions of a type, reproducing scipy's label order.
"""
import logging
import re
import time
from collections import defaultdict
from statistics import mean

import numpy as np

from svim_asm_amd import _lib
from svim_asm_amd.SVCandidate import (CandidateBreakend, CandidateDeletion, CandidateDuplicationInterspersed,
                                      CandidateDuplicationTandem, CandidateInsertion, CandidateInversion)

TYPE_ORDER = ("DEL", "INV", "INS", "DUP_TAN", "DUP_INT", "BND")  # processing order of pair_candidates
_TYPE_RANK = {t: i for i, t in enumerate(TYPE_ORDER)}
_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}
SAME_HAPLOTYPE_DISTANCE = 1000000000


def _pack_keys(candidates_with_haplotype):
    """get_key() tuples → u64 `(type rank, contig rank under str order) << 32 | pos`."""
    keys = [c.get_key() for _, c in candidates_with_haplotype]
    if not keys:
        return np.empty(0, dtype=np.uint64)
    types = [k[0] for k in keys]
    contigs = [k[1] for k in keys]
    positions = [k[2] for k in keys]
    contig_rank = {name: i for i, name in enumerate(sorted(set(contigs)))}
    type_rank = {t: _TYPE_RANK.get(t, len(TYPE_ORDER)) for t in set(types)}
    pos = np.array(positions, dtype=object if max(positions) >= (1 << 62) else np.int64)
    bad = np.flatnonzero((pos < 0) | (pos >= (1 << 32)))
    if len(bad):
        raise ValueError("key position out of range: %r" % (positions[int(bad[0])],))
    n = len(keys)
    high = (np.fromiter(map(type_rank.__getitem__, types), dtype=np.uint64, count=n) << np.uint64(24)) | \
        np.fromiter(map(contig_rank.__getitem__, contigs), dtype=np.uint64, count=n)
    return (high << np.uint64(32)) | pos.astype(np.uint64)


def form_partitions(sv_candidates_with_haplotype, max_distance, ctx=None):
    """Form partitions of (haplotype, candidate) pairs: stable sort by key, new partition when
    type or contig differ or consecutive key positions are more than max_distance apart."""
    items = list(sv_candidates_with_haplotype)
    if not items:
        return []
    ctx = ctx or _lib.default_context()
    perm, part_id, n_parts = ctx.pair_partition(_pack_keys(items), max_distance)
    partitions = [[] for _ in range(n_parts)]
    for j, src in enumerate(perm.tolist()):
        partitions[part_id[j]].append(items[src])
    return partitions


# ------------------------------------------------------------------------------ distances
def haplotype_pair(candidate1, candidate2, reference):
    """The two haplotype strings compute_distance aligns (SVIM_COMBINE.py:43-100)."""
    typ = candidate1.type

    def up(contig, start, end):
        return reference.fetch(contig, start, end).upper()

    if typ in ("DEL", "INV", "DUP_TAN"):
        contig = candidate1.source_contig
        region_start = max(0, min(candidate1.source_start, candidate2.source_start) - 100)
        region_end = min(reference.get_reference_length(contig), max(candidate1.source_end, candidate2.source_end) + 100)
        out = []
        for c in (candidate1, candidate2):
            if typ == "DEL":
                middle = ""
            elif typ == "INV":
                middle = "".join(_COMPLEMENT.get(b, b) for b in reversed(up(contig, c.source_start, c.source_end)))
            else:
                middle = up(contig, c.source_start, c.source_end) * (c.copies + 1)
            out.append(up(contig, region_start, c.source_start) + middle + up(contig, c.source_end, region_end))
        return out
    contig = candidate1.dest_contig
    region_start = max(0, min(candidate1.dest_start, candidate2.dest_start) - 100)
    region_end = min(reference.get_reference_length(contig), max(candidate1.dest_start, candidate2.dest_start) + 100)
    out = []
    for c in (candidate1, candidate2):
        middle = c.sequence if typ == "INS" else up(c.source_contig, c.source_start, c.source_end)
        out.append(up(contig, region_start, c.dest_start) + middle + up(contig, c.dest_start, region_end))
    return out


def edit_distances(string_pairs, k_max=0xFFFFFFFF, ctx=None):
    """Batched global edit distance on the GPU (exact at any length); with a threshold, values
    > k_max come back as 0xFFFFFFFF."""
    if not string_pairs:
        return []
    ctx = ctx or _lib.default_context()
    blobs, a_off, a_len, b_off, b_len, pos = [], [], [], [], [], 0
    for a, b in string_pairs:
        ab, bb = a.encode("latin-1"), b.encode("latin-1")
        a_off.append(pos); a_len.append(len(ab)); pos += len(ab)
        b_off.append(pos); b_len.append(len(bb)); pos += len(bb)
        blobs.append(ab); blobs.append(bb)
    pool = np.frombuffer(b"".join(blobs), dtype=np.uint8) if pos else np.zeros(0, np.uint8)
    return ctx.edit_distance_batch(pool, a_off, a_len, b_off, b_len, k_max).tolist()


def compute_distance(candidate_with_haplotype1, candidate_with_haplotype2, reference):
    haplotype1, candidate1 = candidate_with_haplotype1
    haplotype2, candidate2 = candidate_with_haplotype2
    if haplotype1 == haplotype2:
        return SAME_HAPLOTYPE_DISTANCE
    return edit_distances([tuple(haplotype_pair(candidate1, candidate2, reference))])[0]


def span_position_distance_breakends(candidate1, candidate2):
    hap1, pos1a, dir1a, pos1b, dir1b = candidate1
    hap2, pos2a, dir2a, pos2b, dir2b = candidate2
    if hap1 != hap2 and dir1a == dir2a and dir1b == dir2b:
        return (abs(pos1a - pos2a) + abs(pos1b - pos2b)) / 3000
    return 99999


def _clusters_batch(partitions, condensed, threshold, ctx=None):
    """Flat clusters of many partitions with ONE svx_linkage_cut_batch launch: complete linkage cut at
    `threshold`, clusters in scipy's label order (= fcluster(linkage(d, "complete"), t, "distance"),
    SVIM_COMBINE.py:134-139), members in partition order.  partitions[i] has the condensed distance list
    condensed[i]."""
    if not partitions:
        return []
    ctx = ctx or _lib.default_context()
    sizes = np.array([len(p) for p in partitions], dtype=np.uint32)
    flat = np.fromiter((d for c in condensed for d in c), dtype=np.float64, count=int((sizes.astype(np.int64) * (sizes.astype(np.int64) - 1) // 2).sum()))
    labels = ctx.linkage_cut_batch(flat, sizes, float(threshold)).tolist()
    out, at = [], 0
    for partition in partitions:
        lab = labels[at:at + len(partition)]
        at += len(partition)
        clusters = [[] for _ in range(max(lab))]
        for member, l in zip(partition, lab):
            clusters[l - 1].append(member)
        out.append(clusters)
    return out


def _clusters_from_condensed(partition, distances, threshold):
    return _clusters_batch([partition], [distances], threshold)[0]


class _HaplotypePieces(object):
    """Recipes of the haplotype strings compute_distance aligns (SVIM_COMBINE.py:43-100), for the device:
    every haplotype is three pieces (offset, length, repeat, flags) of one byte pool that holds ONE
    reference window per partition (all members lie within partition_max_distance of each other) plus the
    inserted sequences / interspersed-duplication source intervals, once per candidate.  The strings
    themselves are assembled by svx_haplotype_distance_batch on the GPU."""

    _EMPTY = (0, 0, 0, 0)

    def __init__(self, reference):
        self.reference = reference
        self.fetch_bytes = getattr(reference, "fetch_bytes", None)
        self.chunks, self.size = [], 0
        self.off, self.len, self.rep, self.flg = [], [], [], []
        self._middle = {}  # id(candidate) -> piece of its INS sequence / DUP_INT source interval

    def _add(self, data):
        at = self.size
        self.chunks.append(data)
        self.size += len(data)
        return at

    def _window(self, contig, lo, hi):
        if self.fetch_bytes is not None:
            return self._add(self.fetch_bytes(contig, lo, hi))
        return self._add(self.reference.fetch(contig, lo, hi).encode("latin-1"))

    def partition_window(self, partition):
        """(type, pool offset of the window, window start, contig length) of one partition."""
        first = partition[0][1]
        typ = first.type
        if typ in ("DEL", "INV", "DUP_TAN"):
            contig = first.source_contig
            lo = min(c.source_start for _, c in partition)
            hi = max(c.source_end for _, c in partition)
        else:
            contig = first.dest_contig
            lo = min(c.dest_start for _, c in partition)
            hi = max(c.dest_start for _, c in partition)
        length = self.reference.get_reference_length(contig)
        lo, hi = max(0, lo - 100), min(length, hi + 100)
        return typ, self._window(contig, lo, hi), lo, length

    def _middle_piece(self, typ, c):
        m = self._middle.get(id(c))
        if m is None:
            if typ == "INS":  # the inserted sequence as it is: the reference does not fold its case (:74-75)
                data = c.sequence.encode("latin-1")
                m = (self._add(data), len(data), 1, 0)
            else:             # DUP_INT: the source interval, upper-cased like every reference slice (:86-87)
                n = max(0, min(c.source_end, self.reference.get_reference_length(c.source_contig)) - c.source_start)
                m = (self._window(c.source_contig, c.source_start, c.source_end), n, 1, _lib.PIECE_UPPER) if n else self._EMPTY
            self._middle[id(c)] = m
        return m

    def add_pair(self, window, c1, c2):
        typ, base, lo, length = window
        up = _lib.PIECE_UPPER

        def ref(a, b, repeat=1, flags=up):
            """reference[a:b] of the partition's contig, with fetch()'s clamping of the end (:45-99)."""
            b = min(b, length)
            return (base + a - lo, b - a, repeat, flags) if b > a else self._EMPTY

        if typ in ("DEL", "INV", "DUP_TAN"):
            region_start = max(0, min(c1.source_start, c2.source_start) - 100)
            region_end = min(length, max(c1.source_end, c2.source_end) + 100)
            for c in (c1, c2):
                s, e = c.source_start, c.source_end
                if typ == "DEL":
                    mid = self._EMPTY
                elif typ == "INV":
                    mid = ref(s, e, 1, up | _lib.PIECE_REVCOMP)
                else:
                    if c.copies + 1 > 0xFFFF:
                        raise ValueError("tandem duplication with %d copies" % c.copies)
                    mid = ref(s, e, c.copies + 1)
                self._emit((ref(region_start, s), mid, ref(e, region_end)))
        else:
            region_start = max(0, min(c1.dest_start, c2.dest_start) - 100)
            region_end = min(length, max(c1.dest_start, c2.dest_start) + 100)
            for c in (c1, c2):
                d = c.dest_start
                self._emit((ref(region_start, d), self._middle_piece(typ, c), ref(d, region_end)))

    def _emit(self, three):
        for off, ln, rep, flg in three:
            if ln <= 0 or rep <= 0:
                off, ln, rep, flg = self._EMPTY
            self.off.append(off); self.len.append(ln); self.rep.append(rep); self.flg.append(flg)

    def arrays(self, pair_indices):
        """(pool bytes as uint8, HAP_PIECE_DTYPE array of the selected pairs)."""
        pool = np.frombuffer(b"".join(self.chunks), dtype=np.uint8) if self.size else np.zeros(0, np.uint8)
        pieces = np.empty(len(self.off), dtype=_lib.HAP_PIECE_DTYPE)
        pieces["off"] = self.off
        pieces["len"] = self.len
        pieces["repeat"] = self.rep
        pieces["flags"] = self.flg
        idx = (np.asarray(pair_indices, dtype=np.int64)[:, None] * 6 + np.arange(6)).reshape(-1)
        return pool, pieces[idx]


def _pair_two_member_partitions(partitions, which, reference, edit_distance_threshold, ctx):
    """The bulk of a sample: deletion / insertion partitions of exactly two members, one per haplotype.
    Their haplotype recipes (SVIM_COMBINE.py:43-77) are built with array arithmetic — for two members the
    pair's region IS the partition's window — and complete linkage of two points is one comparison:
    fcluster(linkage([d]), t, "distance") puts them in one cluster iff d <= t, else member 0 gets label 1
    and member 1 label 2 (:134-139).  Returns {partition index: paired?}."""
    if not which:
        return {}
    first = [partitions[pi][0][1] for pi in which]
    second = [partitions[pi][1][1] for pi in which]
    typ = first[0].type
    n = len(which)
    up = _lib.PIECE_UPPER
    fetch = getattr(reference, "fetch_bytes", None) or (lambda c, a, b: reference.fetch(c, a, b).encode("latin-1"))
    lengths = {}
    if typ == "DEL":
        contigs = [c.source_contig for c in first]
        s = np.array([[a.source_start, b.source_start] for a, b in zip(first, second)], dtype=np.int64)
        e = np.array([[a.source_end, b.source_end] for a, b in zip(first, second)], dtype=np.int64)
    else:
        contigs = [c.dest_contig for c in first]
        s = np.array([[a.dest_start, b.dest_start] for a, b in zip(first, second)], dtype=np.int64)
        e = s
    for c in contigs:
        if c not in lengths:
            lengths[c] = reference.get_reference_length(c)
    L = np.array([lengths[c] for c in contigs], dtype=np.int64)
    lo = np.maximum(0, s.min(axis=1) - 100)
    hi = np.minimum(L, e.max(axis=1) + 100)
    chunks = [fetch(c, a, b) for c, a, b in zip(contigs, lo.tolist(), hi.tolist())]
    wlen = np.array([len(w) for w in chunks], dtype=np.int64)
    if not np.array_equal(wlen, np.maximum(hi - lo, 0)):
        raise ValueError("reference windows shorter than the index says")
    base = np.concatenate(([0], np.cumsum(wlen)))[:-1]
    pieces = np.zeros((n, 2, 3), dtype=_lib.HAP_PIECE_DTYPE)
    for h in (0, 1):
        # prefix reference[region_start : start] and suffix reference[end : region_end], fetch()'s clamping
        pre = np.maximum(np.minimum(s[:, h], L) - lo, 0)
        suf = np.maximum(hi - e[:, h], 0)
        pieces["off"][:, h, 0] = np.where(pre > 0, base, 0)
        pieces["len"][:, h, 0] = pre
        pieces["repeat"][:, h, 0] = pre > 0
        pieces["flags"][:, h, 0] = np.where(pre > 0, up, 0)
        pieces["off"][:, h, 2] = np.where(suf > 0, base + e[:, h] - lo, 0)
        pieces["len"][:, h, 2] = suf
        pieces["repeat"][:, h, 2] = suf > 0
        pieces["flags"][:, h, 2] = np.where(suf > 0, up, 0)
    if typ == "INS":  # the inserted sequences as they are (:74-75), behind the windows in the pool
        seqs = [c.sequence.encode("latin-1") for pair in zip(first, second) for c in pair]
        slen = np.array([len(x) for x in seqs], dtype=np.int64).reshape(n, 2)
        soff = (int(wlen.sum()) + np.concatenate(([0], np.cumsum(slen.reshape(-1))))[:-1]).reshape(n, 2)
        pieces["off"][:, :, 1] = np.where(slen > 0, soff, 0)
        pieces["len"][:, :, 1] = slen
        pieces["repeat"][:, :, 1] = slen > 0
        chunks += seqs
    pool = np.frombuffer(b"".join(chunks), dtype=np.uint8) if chunks else np.zeros(0, np.uint8)
    k_max = max(min(max(int(edit_distance_threshold), -1), 0xFFFFFFFE), 0)
    d = ctx.haplotype_distance_batch(pool, pieces.reshape(-1), k_max).astype(np.float64)
    d[d == float(0xFFFFFFFF)] = k_max + 1  # "more than the threshold" is all that is known, and all that matters
    paired = d <= float(edit_distance_threshold)
    return dict(zip(which, paired.tolist()))


def pair_haplotypes(partitions, reference, edit_distance_threshold=10, ctx=None):
    """Cluster each partition (2..10 members) by complete linkage over haplotype edit distances.  The
    haplotype strings of all cross-haplotype pairs are assembled on the GPU from one reference window per
    partition, their distances come from one batch per mode, the linkage cuts from one more launch."""
    ctx = ctx or _lib.default_context()
    jobs = []  # (partition index, i, j)
    recipes = _HaplotypePieces(reference)
    two = [pi for pi, p in enumerate(partitions)
           if len(p) == 2 and p[0][0] != p[1][0] and p[0][1].type in ("DEL", "INS")]
    settled = _pair_two_member_partitions(partitions, two, reference, edit_distance_threshold, ctx)
    for pi, partition in enumerate(partitions):
        if len(partition) < 2 or len(partition) > 10 or pi in settled:
            continue
        window = None
        for i in range(len(partition) - 1):
            for j in range(i + 1, len(partition)):
                if partition[i][0] != partition[j][0]:
                    if window is None:
                        window = recipes.partition_window(partition)
                    jobs.append((pi, i, j))
                    recipes.add_pair(window, partition[i][1], partition[j][1])
    # two-member partitions only need "<= threshold?"; larger ones get exact values so that the
    # dendrogram above the cut (hence scipy's cluster label order) is the reference's
    dist = {}
    thr = [k for k, (pi, _, _) in enumerate(jobs) if len(partitions[pi]) == 2]
    exa = [k for k, (pi, _, _) in enumerate(jobs) if len(partitions[pi]) > 2]
    # any threshold the reference accepts: a negative one pairs nothing, one beyond 32 bits everything
    k_max = min(max(int(edit_distance_threshold), -1), 0xFFFFFFFE)
    if thr:
        # (a negative threshold: only "distance > threshold" matters, and every distance is >= 0)
        pool, pieces = recipes.arrays(thr)
        got = ctx.haplotype_distance_batch(pool, pieces, max(k_max, 0)).tolist()
        for k, d in zip(thr, got):
            dist[jobs[k]] = d if d != 0xFFFFFFFF else max(k_max, 0) + 1
    if exa:
        pool, pieces = recipes.arrays(exa)
        for k, d in zip(exa, ctx.haplotype_distance_batch(pool, pieces, 0xFFFFFFFF).tolist()):
            dist[jobs[k]] = d
    todo, condensed = [], []
    for pi, partition in enumerate(partitions):
        if 2 <= len(partition) <= 10 and pi not in settled:
            todo.append(partition)
            condensed.append([dist.get((pi, i, j), SAME_HAPLOTYPE_DISTANCE)
                              for i in range(len(partition) - 1) for j in range(i + 1, len(partition))])
    clustered = iter(_clusters_batch(todo, condensed, edit_distance_threshold, ctx))
    clusters_final = []
    for pi, partition in enumerate(partitions):
        if len(partition) < 2:
            clusters_final.append(partition)
        elif pi in settled:
            if settled[pi]:
                clusters_final.append(partition)
            else:
                clusters_final.append([partition[0]])
                clusters_final.append([partition[1]])
        elif len(partition) > 10:
            # very large partitions tend to be in difficult regions: dropped (SVIM_COMBINE.py:126-128)
            logging.debug("Ignored partition of size {0} and type {1}: {2}".format(
                len(partition), partition[0][1].get_key()[0],
                ",".join("{0}:{1}".format(m[1].get_key()[1], m[1].get_key()[2]) for m in partition)))
        else:
            clusters_final.extend(next(clustered))
    return clusters_final


def pair_haplotypes_breakends(partitions, span_position_distance_threshold=0.3, ctx=None):
    todo, condensed = [], []
    for partition in partitions:
        if 2 <= len(partition) <= 10:
            rows = [(hap, c.get_source()[1], 1 if c.source_direction == "fwd" else 0, c.get_destination()[1],
                     1 if c.dest_direction == "fwd" else 0) for hap, c in partition]
            todo.append(partition)
            condensed.append([span_position_distance_breakends(rows[i], rows[j])
                              for i in range(len(rows) - 1) for j in range(i + 1, len(rows))])
    clustered = iter(_clusters_batch(todo, condensed, span_position_distance_threshold, ctx))
    clusters_final = []
    for partition in partitions:
        if len(partition) < 2:
            clusters_final.append(partition)
        elif len(partition) > 10:
            continue
        else:
            clusters_final.extend(next(clustered))
    return clusters_final


# ------------------------------------------------------------------------------ pairing
def _rebuild(typ, cluster, bam):
    """Candidate of a cluster: coordinates/payload of its first member, reads concatenated,
    flags OR-ed, copy number averaged with round() (banker's rounding) — :184-363."""
    first = cluster[0][1]
    if len(cluster) == 1:
        genotype = "1/0" if cluster[0][0] == 1 else "0/1"
        second, reads = None, first.reads
    else:
        genotype = "1/1"
        second = cluster[1][1]
        reads = first.reads + second.reads
    if typ == "DEL":
        # the constructor would clamp coordinates that are clamped already: same object state from a copy
        assert first.source_end >= first.source_start, \
            "Deletion end ({0}:{1}) is smaller than its start ({0}:{2}). From read {3}".format(
                first.source_contig, first.source_end, first.source_start, reads)
        new = CandidateDeletion.__new__(CandidateDeletion)
        new.__dict__.update(first.__dict__)
        new.reads, new.genotype = reads, genotype
        return new
    if typ == "INV":
        complete = first.complete if second is None else (first.complete or second.complete)
        return CandidateInversion(first.source_contig, first.source_start, first.source_end, reads, complete, bam,
                                  genotype)
    if typ == "INS":
        assert first.dest_end >= first.dest_start, \
            "Insertion end ({0}:{1}) is smaller than its start ({0}:{2}). From read {3}".format(
                first.dest_contig, first.dest_end, first.dest_start, reads)
        new = CandidateInsertion.__new__(CandidateInsertion)
        new.__dict__.update(first.__dict__)
        new.reads, new.genotype = reads, genotype
        return new
    if typ == "DUP_TAN":
        copies = first.copies if second is None else round(mean([first.copies, second.copies]))
        fully = first.fully_covered if second is None else (first.fully_covered or second.fully_covered)
        return CandidateDuplicationTandem(first.source_contig, first.source_start, first.source_end, copies, fully,
                                          reads, bam, genotype)
    if typ == "DUP_INT":
        cutpaste = first.cutpaste if second is None else (first.cutpaste or second.cutpaste)
        return CandidateDuplicationInterspersed(first.source_contig, first.source_start, first.source_end,
                                                first.dest_contig, first.dest_start, first.dest_end, reads, bam,
                                                cutpaste, genotype)
    return CandidateBreakend(first.source_contig, first.source_start, first.source_direction, first.dest_contig,
                             first.dest_start, first.dest_direction, reads, bam, genotype)


_LOG_NAME = {"DEL": "deletions", "INV": "inversions", "INS": "insertions", "DUP_TAN": "tandem duplications",
             "DUP_INT": "interspersed duplications", "BND": "breakends"}


def pair_candidates(sv_candidates1, sv_candidates2, reference, bam, options):
    ctx = _lib.default_context(getattr(options, "device", 0) or 0)
    # one sort/partition launch for all types; the input order per type is hap-1 list then hap-2 list
    per_type = {typ: ([], []) for typ in TYPE_ORDER}
    for hap, cands in ((0, sv_candidates1), (1, sv_candidates2)):
        for c in cands:
            bucket = per_type.get(c.type)
            if bucket is not None:
                bucket[hap].append((hap + 1, c))
    tagged = []
    for typ in TYPE_ORDER:
        tagged += per_type[typ][0]
        tagged += per_type[typ][1]
    partitions = form_partitions(tagged, options.partition_max_distance, ctx=ctx)
    by_type = defaultdict(list)
    for part in partitions:
        by_type[part[0][1].type].append(part)
    paired_candidates = []
    for typ in TYPE_ORDER:
        n = len(per_type[typ][0]) + len(per_type[typ][1])
        logging.info("Pairing {0} {1}...".format(n, _LOG_NAME[typ]))
        if typ == "BND":
            clusters = pair_haplotypes_breakends(by_type[typ])
        else:
            clusters = pair_haplotypes(by_type[typ], reference, options.max_edit_distance, ctx=ctx)
        for cluster in clusters:
            if len(cluster) in (1, 2):
                paired_candidates.append(_rebuild(typ, cluster, bam))
            else:
                logging.error("Cluster size should be either 1 or 2 but is " + str(len(cluster)))
    return paired_candidates


# ------------------------------------------------------------------------------ output
def sorted_nicely(vcf_entries):
    """Natural sort of ((contig, start, end), vcf_string, sv_type) entries: chr10 after chr2."""
    # a sample has a few dozen contig names and tens of thousands of entries: the names are ranked once under
    # the natural key (equal keys — "chr01" and "chr1" — share a rank), the entries sort on integers
    names = set(entry[0][0] for entry in vcf_entries)
    natural = {name: [int(tok) if tok.isdigit() else tok for tok in re.split("([0-9]+)", str(name))] for name in names}
    rank, last = {}, None
    for name in sorted(names, key=natural.__getitem__):
        if last is None or natural[name] != natural[last]:
            r = len(rank)
        rank[name] = r
        last = name
    return sorted(vcf_entries, key=lambda entry: (rank[entry[0][0]], entry[0][1], entry[0][2]))


def _header_lines(version, contig_names, contig_lengths, types_to_output, options):
    tandem_as_dup = (not options.tandem_duplications_as_insertions) and "DUP:TANDEM" in types_to_output
    int_as_dup = (not options.interspersed_duplications_as_insertions) and "DUP:INT" in types_to_output
    yield "##fileformat=VCFv4.2"
    yield "##fileDate={0}".format(time.strftime("%Y-%m-%d|%I:%M:%S%p|%Z|%z"))
    yield "##source=SVIM-asm-v{0}".format(version)
    for name, length in zip(contig_names, contig_lengths):
        yield "##contig=<ID={0},length={1}>".format(name, length)
    alts = [("DEL", "Deletion", "DEL" in types_to_output), ("INV", "Inversion", "INV" in types_to_output),
            ("DUP", "Duplication", tandem_as_dup or int_as_dup), ("DUP:TANDEM", "Tandem Duplication", tandem_as_dup),
            ("DUP:INT", "Interspersed Duplication", int_as_dup), ("INS", "Insertion", "INS" in types_to_output),
            ("BND", "Breakend", "BND" in types_to_output)]
    for ident, text, enabled in alts:
        if enabled:
            yield '##ALT=<ID={0},Description="{1}">'.format(ident, text)
    yield '##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">'
    yield '##INFO=<ID=CUTPASTE,Number=0,Type=Flag,Description="Genomic origin of interspersed duplication seems to be deleted">'
    yield '##INFO=<ID=END,Number=1,Type=Integer,Description="End position of the variant described in this record">'
    yield '##INFO=<ID=SVLEN,Number=1,Type=Integer,Description="Difference in length between REF and ALT alleles">'
    if options.query_names:
        yield '##INFO=<ID=READS,Number=.,Type=String,Description="Names of all supporting reads">'
    yield '##FILTER=<ID=not_fully_covered,Description="Tandem duplication is not fully covered by a contig">'
    yield '##FILTER=<ID=incomplete_inversion,Description="Only one inversion breakpoint is supported">'
    yield '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">'
    if tandem_as_dup:
        yield '##FORMAT=<ID=CN,Number=1,Type=Integer,Description="Copy number of tandem duplication (e.g. 2 for one additional copy)">'
    yield "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + options.sample


def collect_vcf_entries(int_duplication_candidates, inversion_candidates, tandem_duplication_candidates,
                        deletion_candidates, insertion_candidates, breakend_candidates, types_to_output, reference,
                        options):
    """((contig, start, end), line, label) per record, in the reference's list order (:428-464)."""
    seq = not options.symbolic_alleles
    names = options.query_names
    entries = []
    if "DEL" in types_to_output:
        for c in deletion_candidates:
            contig, start, end = c.get_source()
            entries.append(((contig, max(1, start), end), c.get_vcf_entry(seq, reference, names), "DEL"))
    if "INV" in types_to_output:
        for c in inversion_candidates:
            contig, start, end = c.get_source()
            entries.append(((contig, start + 1, end), c.get_vcf_entry(seq, reference, names), "INV"))
    if "INS" in types_to_output:
        for c in insertion_candidates:
            contig, start, end = c.get_destination()
            entries.append(((contig, max(1, start), end), c.get_vcf_entry(seq, reference, names), "INS"))
    if options.tandem_duplications_as_insertions:
        if "INS" in types_to_output:
            for c in tandem_duplication_candidates:
                entries.append(((c.source_contig, c.source_start + 1, c.source_end),
                                c.get_vcf_entry_as_ins(seq, reference, names), "INS"))
    elif "DUP:TANDEM" in types_to_output:
        for c in tandem_duplication_candidates:
            entries.append(((c.source_contig, c.source_start + 1, c.source_end), c.get_vcf_entry_as_dup(names),
                            "DUP_TANDEM"))
    if options.interspersed_duplications_as_insertions:
        if "INS" in types_to_output:
            for c in int_duplication_candidates:
                contig, start, end = c.get_destination()
                entries.append(((contig, max(1, start), end), c.get_vcf_entry_as_ins(seq, reference, names), "INS"))
    elif "DUP:INT" in types_to_output:
        for c in int_duplication_candidates:
            contig, start, end = c.get_source()
            entries.append(((contig, start + 1, end), c.get_vcf_entry_as_dup(names), "DUP_INT"))
    if "BND" in types_to_output:
        for c in breakend_candidates:
            (sc, sp), (dc, dp) = c.get_source(), c.get_destination()
            entries.append(((sc, sp + 1, sp + 2), c.get_vcf_entry(names), "BND"))
            entries.append(((dc, dp + 1, dp + 2), c.get_vcf_entry_reverse(names), "BND"))
    return entries


def write_final_vcf(int_duplication_candidates, inversion_candidates, tandem_duplication_candidates,
                    deletion_candidates, insertion_candidates, breakend_candidates, version, contig_names,
                    contig_lengths, types_to_output, reference, options):
    with open(options.working_dir + "/variants.vcf", "w") as vcf_output:
        for line in _header_lines(version, contig_names, contig_lengths, types_to_output, options):
            print(line, file=vcf_output)
        entries = collect_vcf_entries(int_duplication_candidates, inversion_candidates,
                                      tandem_duplication_candidates, deletion_candidates, insertion_candidates,
                                      breakend_candidates, types_to_output, reference, options)
        if not options.symbolic_alleles:
            reference.close()
        counter = defaultdict(int)
        lines = []
        for _, entry, svtype in sorted_nicely(entries):
            counter[svtype] += 1
            lines.append(entry.replace("PLACEHOLDERFORID", "svim_asm.{0}.{1}".format(svtype, counter[svtype]), 1))
        if lines:
            vcf_output.write("\n".join(lines))
            vcf_output.write("\n")
