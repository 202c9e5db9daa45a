"""PAIR (diploid) + VCF output, columnar.

Mirrors form_partitions (SVIM_COMBINE.py:15-32), compute_distance (:35-102),
span_position_distance_breakends (:105-117), pair_haplotypes (:120-140),
pair_haplotypes_breakends (:143-161), pair_candidates (:164-366), sorted_nicely (:369-376) and
write_final_vcf (:379-477).

The reference keeps (haplotype, Candidate) tuples in Python lists, sorts them with `sorted`, loops over the
partitions and builds a new object per output candidate.  Here the candidates are rows of a CandidateTable
(svim_asm_amd/table.py) from COLLECT to the VCF:
  * keys are packed from the columns; the stable sort by Candidate.get_key() + partition sweep is ONE
    svx_pair_partition launch for all six SV types (the type is the most significant key field, so
    partitions and their order are the ones the reference gets type by type);
  * the cross-haplotype pairs of every partition of 2..10 members are enumerated with array arithmetic (one
    index template per partition size), their haplotype strings are described as three pieces each of one byte
    pool (one reference window per partition, fetched in one native batch) and assembled + aligned on the GPU
    (svx_haplotype_distance_batch; edlib in the reference);
  * complete linkage + flat cut of the partitions with 3..10 members (scipy in the reference) is one
    svx_linkage_cut_batch launch per kind of distance, in scipy's label order; two members need no
    dendrogram: one cluster iff d <= t;
  * the paired candidates are a row selection plus the merged columns (reads concatenated, flags OR-ed,
    copies averaged with banker's rounding, constructors re-applied, :184-363);
  * the VCF record lines are formatted, naturally sorted and numbered by svx_vcf_format (include/svx_text.h).
The reference's seams (`form_partitions`, `pair_candidates`, `write_final_vcf`, ...) keep their signatures;
lists of Candidate objects are converted to a table on the way in, `pair_candidates` returns a CandidateList.
"""
import ctypes as C
import logging
import os
import re
import stat
import time

import numpy as np

from svim_asm_amd import _lib
from svim_asm_amd.SVIM_COLLECT import _apply_constructors
from svim_asm_amd.table import (CandidateList, CandidateTable, F_BOOL, F_DST_REV, F_SRC_REV, GENOTYPES, T_BND, T_DEL,
                                T_DUP_INT, T_DUP_TAN, T_INS, T_INV, TYPE_ORDER, _ranges, as_table)

_TYPE_RANK = {t: i for i, t in enumerate(TYPE_ORDER)}
_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}
SAME_HAPLOTYPE_DISTANCE = 1000000000
BREAKEND_MISMATCH_DISTANCE = 99999


class _NoBam(object):
    references = ()

    def get_reference_length(self, name):
        raise KeyError(name)


def _str_rank(names):
    """Rank of every name under Python str ordering; equal names share a rank."""
    uniq = {name: k for k, name in enumerate(sorted(set(names)))}
    return np.array([uniq[name] for name in names], dtype=np.int64).reshape(len(names))


def _keys_of_table(t):
    """get_key() of every row → u64 `(type rank << 24 | contig rank under str order) << 32 | pos`."""
    if len(t) == 0:
        return np.empty(0, dtype=np.uint64)
    pos = t.key_position()
    bad = np.flatnonzero((pos < 0) | (pos >= (1 << 32)))
    if len(bad):
        raise ValueError("key position out of range: %r" % (int(pos[int(bad[0])]),))
    rank = _str_rank(t.contigs)
    high = (t.type.astype(np.uint64) << np.uint64(24)) | rank[t.key_contig()].astype(np.uint64)
    return (high << np.uint64(32)) | pos.astype(np.uint64)


def _pack_keys(candidates_with_haplotype):
    """get_key() tuples of (haplotype, candidate) items → the u64 keys above."""
    items = list(candidates_with_haplotype)
    if not items:
        return np.empty(0, dtype=np.uint64)
    return _keys_of_table(CandidateTable.from_objects([c for _, c in items], _NoBam()))


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
    return BREAKEND_MISMATCH_DISTANCE


# ------------------------------------------------------------------------------ reference windows
def _fetch_windows(reference, names, ids, lo, hi, upper):
    """reference.fetch(names[ids[i]], lo[i], hi[i]) for all i: (uint8 pool, int64 offsets).  One native batch
    for the product's FastaFile (contig ids go down as they are: no list of 10^5 names); any object with pysam's
    fetch() otherwise."""
    ids = np.asarray(ids)
    batch = getattr(reference, "fetch_batch", None)
    if batch is not None:
        return batch(names, lo, hi, upper=upper, ids=ids)
    parts = []
    for c, a, b in zip(ids.tolist(), np.asarray(lo).tolist(), np.asarray(hi).tolist()):
        s = reference.fetch(names[c], a, b)
        parts.append((s.upper() if upper else s).encode("latin-1"))
    off = np.zeros(len(parts) + 1, np.int64)
    if parts:
        np.cumsum([len(p) for p in parts], out=off[1:])
    return (np.frombuffer(b"".join(parts), dtype=np.uint8) if parts else np.zeros(0, np.uint8)), off


def _reference_lengths(reference, names, used):
    """reference.get_reference_length of the contig ids in `used` (others -1)."""
    out = np.full(len(names), -1, np.int64)
    for cid in np.unique(used).tolist():
        if cid >= 0:
            out[cid] = reference.get_reference_length(names[cid])
    return out


# ------------------------------------------------------------------------------ pairing on columns
class _Clusters(object):
    """Output clusters as columns: partition number, scipy label inside it, first / second member row."""

    def __init__(self):
        self.part, self.label, self.first, self.second, self.size = [], [], [], [], []

    def add(self, part, label, first, second, size):
        self.part.append(np.asarray(part, dtype=np.int64))
        self.label.append(np.asarray(label, dtype=np.int64))
        self.first.append(np.asarray(first, dtype=np.int64))
        self.second.append(np.asarray(second, dtype=np.int64))
        self.size.append(np.asarray(size, dtype=np.int64))

    def ordered(self):
        if not self.part:
            z = np.zeros(0, np.int64)
            return z, z, z, z
        part, label = np.concatenate(self.part), np.concatenate(self.label)
        o = np.lexsort((label, part))  # partitions in sorted-key order, clusters in label order (:136-139)
        return part[o], np.concatenate(self.first)[o], np.concatenate(self.second)[o], np.concatenate(self.size)[o]


def _haplotype_pieces(T, kstart, kend, job_a, job_b, job_part, p_type, p_contig, win_base, win_lo, L_part, reference,
                      seq_split=None):
    """The recipes of the jobs: svx_pair_recipes (include/svx.h; host arithmetic in libsvx.so) for everything but the
    fetch of the interspersed duplications' source intervals.  _haplotype_pieces_numpy below is the same arithmetic as
    array expressions (the form it had until round 5; tests/test_host_logic.py holds the two against each other)."""
    J = len(job_a)
    lib = _lib.load()
    typ = p_type[job_part]
    windows_bytes = int(win_base[-1]) if len(win_base) else 0
    extra_parts, extra_at = [], windows_bytes
    mid_off = mid_len = None
    dint = np.flatnonzero(typ == T_DUP_INT)
    if len(dint):
        rows = (job_a, job_b)
        mid_off, mid_len = np.zeros((J, 2), np.int64), np.zeros((J, 2), np.int64)
        cand = np.unique(np.concatenate((job_a[dint], job_b[dint])))
        Ls = _reference_lengths(reference, T.contigs, T.sc[cand])[T.sc[cand]]
        n_src = np.maximum(0, np.minimum(T.se[cand], Ls) - T.ss[cand])
        pool, off = _fetch_windows(reference, T.contigs, T.sc[cand], T.ss[cand], T.se[cand], False)
        if not np.array_equal(off[1:] - off[:-1], n_src):
            raise ValueError("reference windows shorter than the index says")
        extra_parts.append(pool)
        for h in (0, 1):
            where = np.searchsorted(cand, rows[h][dint])
            mid_off[dint, h] = extra_at + off[:-1][where]
            mid_len[dint, h] = n_src[where]
        extra_at += len(pool)
    keep = []

    def ptr(a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        keep.append(a)
        return a.ctypes.data if a.size else None
    pieces = np.zeros((J, 2, 3), dtype=_lib.HAP_PIECE_DTYPE)
    seq_range, worst = np.zeros(4, np.int64), np.zeros(1, np.int64)
    arg = _lib.RecipeIn(
        n_rows=len(T), type=ptr(T.type, np.uint8), ss=ptr(T.ss, np.int64), se=ptr(T.se, np.int64), ds=ptr(T.ds, np.int64),
        q_off=ptr(T.q_off, np.int64), q_len=ptr(T.q_len, np.int64), copies=ptr(T.copies, np.int64), n_jobs=J,
        job_a=ptr(job_a, np.int64), job_b=ptr(job_b, np.int64), job_part=ptr(job_part, np.int64), n_parts=len(p_type),
        part_type=ptr(p_type, np.int64), part_len=ptr(L_part, np.int64), win_base=ptr(win_base, np.int64),
        win_lo=ptr(win_lo, np.int64), extra_at=extra_at, seq_split=-1 if seq_split is None else int(seq_split),
        mid_off=ptr(mid_off, np.int64) if mid_off is not None else None,
        mid_len=ptr(mid_len, np.int64) if mid_len is not None else None)
    rc = lib.svx_pair_recipes(C.byref(arg), pieces.ctypes.data if J else None, seq_range.ctypes.data, worst.ctypes.data)
    if rc == _lib.SVX_E_TOO_LARGE:
        raise ValueError("tandem duplication with %d copies" % int(worst[0]))
    if rc != 0:
        raise _lib.SvxError(rc, "svx_pair_recipes")
    if int(seq_range[1]) > int(seq_range[0]) or int(seq_range[3]) > int(seq_range[2]):
        seqs = np.asarray(T.seqs, dtype=np.uint8)
        extra_parts.extend(seqs[int(lo_):int(hi_)] for lo_, hi_ in ((seq_range[0], seq_range[1]), (seq_range[2], seq_range[3])) if hi_ > lo_)
    return pieces, extra_parts


def _haplotype_pieces_numpy(T, kstart, kend, job_a, job_b, job_part, p_type, p_contig, win_base, win_lo, L_part, reference,
                            seq_split=None):
    """svx_hap_piece recipes (3 per haplotype, 6 per job) of the strings compute_distance aligns
    (SVIM_COMBINE.py:43-100) for jobs (candidate rows a, b of partition job_part).  Returns (pieces [J, 2, 3],
    extra pool parts appended behind the windows: interspersed-duplication source intervals, inserted sequences)."""
    J = len(job_a)
    up = _lib.PIECE_UPPER
    pieces = np.zeros((J, 2, 3), dtype=_lib.HAP_PIECE_DTYPE)
    typ = p_type[job_part]
    L = L_part[job_part]
    base, lo = win_base[job_part], win_lo[job_part]
    rows = (job_a, job_b)
    s = (kstart[job_a], kstart[job_b])
    e = (kend[job_a], kend[job_b])
    rs = np.maximum(0, np.minimum(s[0], s[1]) - 100)
    re_ = np.minimum(L, np.maximum(e[0], e[1]) + 100)
    windows_bytes = int(win_base[-1]) if len(win_base) else 0

    def put(h, k, off, ln, rep, flags):
        ok = (ln > 0) & (rep > 0)
        pieces["off"][:, h, k] = np.where(ok, off, 0)
        pieces["len"][:, h, k] = np.where(ok, ln, 0)
        pieces["repeat"][:, h, k] = np.where(ok, rep, 0)
        pieces["flags"][:, h, k] = np.where(ok, flags, 0)

    # the middle pieces that are not slices of the partition's window live behind the windows in the pool
    extra_parts, extra_at = [], windows_bytes
    dint = np.flatnonzero(typ == T_DUP_INT)
    mid_off = np.zeros((J, 2), np.int64)
    mid_len = np.zeros((J, 2), np.int64)
    if len(dint):
        # DUP_INT: the source interval, upper-cased like every reference slice (:86-87); one fetch per candidate
        cand = np.unique(np.concatenate((job_a[dint], job_b[dint])))
        Ls = _reference_lengths(reference, T.contigs, T.sc[cand])[T.sc[cand]]
        n_src = np.maximum(0, np.minimum(T.se[cand], Ls) - T.ss[cand])
        pool, off = _fetch_windows(reference, T.contigs, T.sc[cand], T.ss[cand], T.se[cand], False)
        if not np.array_equal(off[1:] - off[:-1], n_src):
            raise ValueError("reference windows shorter than the index says")
        extra_parts.append(pool)
        for h in (0, 1):
            where = np.searchsorted(cand, rows[h][dint])
            mid_off[dint, h] = extra_at + off[:-1][where]
            mid_len[dint, h] = n_src[where]
        extra_at += len(pool)
    ins = typ == T_INS
    want_seqs = bool(ins.any())
    seq_ranges = []  # the stretches of the sequence pool that are appended behind the windows
    if want_seqs:
        # INS: the inserted sequence as it is — the reference does not fold its case (:74-75).  Only the OFFSETS are
        # needed here; the bytes (possibly still being decoded) are appended to the pool at the very end — the stretch
        # of the sequence pool these jobs' alleles lie in (a chunk of PAIR's jobs is a range of partitions, i.e. of
        # contigs: its alleles are a stretch of each haplotype's part of the pool), not the whole pool
        used = np.concatenate((job_a[ins], job_b[ins]))
        used = used[T.q_len[used] > 0]
        # the two haplotype tables' sequences lie one behind the other in the pool: one stretch of each
        split = int(seq_split) if seq_split is not None else 0
        shift = np.zeros(2, np.int64)  # what to subtract from a pool offset below / at or above `split`
        at = extra_at
        for side in (0, 1):
            u = used[(T.q_off[used] >= split) == bool(side)] if seq_split is not None else (used if side == 0 else used[:0])
            if len(u):
                q_lo, q_hi = int(T.q_off[u].min()), int((T.q_off[u] + T.q_len[u]).max())
                seq_ranges.append((q_lo, q_hi))
                shift[side] = q_lo - at
                at += q_hi - q_lo
        for h in (0, 1):
            off_h = T.q_off[rows[h]]
            side_h = (off_h >= split).astype(np.int64) if seq_split is not None else np.zeros(J, np.int64)
            mid_off[:, h] = np.where(ins, off_h - shift[side_h], mid_off[:, h])
            mid_len[:, h] = np.where(ins, T.q_len[rows[h]], mid_len[:, h])
    tan = typ == T_DUP_TAN
    if bool(tan.any()):
        worst = int(max(T.copies[job_a[tan]].max(), T.copies[job_b[tan]].max()))
        if worst + 1 > 0xFFFF:
            raise ValueError("tandem duplication with %d copies" % worst)
    one = np.ones(J, np.int64)
    for h in (0, 1):
        sh, eh = s[h], e[h]
        # reference[region_start : start] and reference[end : region_end], with fetch()'s clamping of the end
        put(h, 0, base + rs - lo, np.minimum(sh, L) - rs, one, up)
        put(h, 2, base + eh - lo, re_ - eh, one, up)
        inner_len = np.minimum(eh, L) - sh
        m_off = np.where((typ == T_INV) | tan, base + sh - lo, mid_off[:, h])
        m_len = np.where((typ == T_INV) | tan, inner_len, mid_len[:, h])
        m_len = np.where(typ == T_DEL, 0, m_len)
        m_rep = np.where(tan, T.copies[rows[h]] + 1, one)
        m_flg = np.where(typ == T_INV, up | _lib.PIECE_REVCOMP, np.where(tan | (typ == T_DUP_INT), up, 0))
        put(h, 1, m_off, m_len, m_rep, m_flg)
    if seq_ranges:
        seqs = np.asarray(T.seqs, dtype=np.uint8)
        extra_parts.extend(seqs[q_lo:q_hi] for q_lo, q_hi in seq_ranges)
    return pieces, extra_parts


LAST_TIMING = {}  # seconds per stage of the latest pair_tables / vcf_body call (tools/, bench legs)
_PAIR_CHUNK_MIN_JOBS = 30000  # PAIR's distance jobs are pipelined in chunks from this many on (a human sample: ~20 k) ...
_PAIR_CHUNKS = max(1, int(os.environ.get("SVX_PAIR_CHUNKS", "2")))  # ... this many chunks (config 5, medians of three interleaved
#                               runs of five: 1 chunk 0.185 s, 2 chunks 0.167 s, 4 chunks 0.171 s, 8 slower; profiles/README.md)


_cpu_mark = [0.0]


def _clock_start():
    _cpu_mark[0] = time.process_time()
    return time.perf_counter()


def _clock(stage, t0):
    """Wall seconds of a stage, and beside them (`…_cpu_s`) the CPU seconds of all threads of the process: under a
    CPU quota (cgroup cpu.max) those, not the thread count, are what a run costs."""
    now, cpu = time.perf_counter(), time.process_time()
    LAST_TIMING[stage] = LAST_TIMING.get(stage, 0.0) + now - t0
    key = stage[:-2] + "_cpu_s" if stage.endswith("_s") else stage + "_cpu"
    LAST_TIMING[key] = LAST_TIMING.get(key, 0.0) + cpu - _cpu_mark[0]
    _cpu_mark[0] = cpu
    return now


def pair_tables(t1, t2, reference, bam, options, ctx=None):
    """pair_candidates on tables: the paired candidates as a CandidateTable, rows in the reference's order."""
    ctx = ctx or _lib.default_context(getattr(options, "device", 0) or 0)
    for k in [k for k in LAST_TIMING if k.startswith("pair_")]:
        del LAST_TIMING[k]
    tc = _clock_start()
    base_bam = getattr(bam, "_bam", bam)
    contigs = list(base_bam.references)
    T = CandidateTable.concat([t1, t2], contigs, [base_bam.get_reference_length(c) for c in contigs])
    n1, n = len(t1), len(t1) + len(t2)
    hap = np.ones(n, np.int64)
    hap[n1:] = 2
    counts = T.counts_by_type()
    for ti, typ in enumerate(TYPE_ORDER):
        logging.info("Pairing {0} {1}...".format(int(counts[ti]), _LOG_NAME[typ]))
    if n == 0:
        return T
    tc = _clock("pair_concat_s", tc)
    # one sort/partition launch for all types; the input order per type is hap-1 list then hap-2 list (:182,...)
    inp = np.lexsort((hap, T.type))
    perm, part_id, n_parts = ctx.pair_partition(_keys_of_table(T)[inp], options.partition_max_distance)
    tc = _clock("pair_sort_s", tc)
    order = inp[perm.astype(np.int64)]
    p_size = np.bincount(part_id.astype(np.int64), minlength=n_parts).astype(np.int64)
    p_start = np.cumsum(p_size) - p_size
    p_type = T.type[order[p_start]].astype(np.int64)
    threshold = options.max_edit_distance
    out = _Clusters()

    # ---- partitions of one member: one cluster (:123-124)
    single = np.flatnonzero(p_size == 1)
    out.add(single, np.ones(len(single)), order[p_start[single]], np.full(len(single), -1), np.ones(len(single)))
    # ---- very large partitions tend to be in difficult regions: dropped (:126-128; breakends silently, :149)
    if logging.getLogger().isEnabledFor(logging.DEBUG):
        for pi in np.flatnonzero((p_size > 10) & (p_type != T_BND)).tolist():
            rows = order[p_start[pi]:p_start[pi] + p_size[pi]]
            kc, kp = T.key_contig()[rows], T.key_position()[rows]
            logging.debug("Ignored partition of size {0} and type {1}: {2}".format(
                int(p_size[pi]), TYPE_ORDER[p_type[pi]], ",".join("{0}:{1}".format(T.contigs[c], p) for c, p in zip(kc.tolist(), kp.tolist()))))

    # ---- partitions of 2..10 members: all pairs (i < j), one index template per size
    src_like = (T.type == T_DEL) | (T.type == T_INV) | (T.type == T_DUP_TAN)
    kstart = np.where(src_like, T.ss, T.ds)   # the interval compute_distance works on: source (DEL, INV, DUP_TAN),
    kend = np.where(src_like, T.se, T.ds)     # destination START twice (INS, DUP_INT; :71,:83)
    classes = []
    for size in range(2, 11):
        P = np.flatnonzero(p_size == size)
        if not len(P):
            continue
        M = order[p_start[P][:, None] + np.arange(size)[None, :]]
        iu, ju = np.triu_indices(size, 1)
        A, B = M[:, iu], M[:, ju]
        cross = hap[A] != hap[B]
        bnd = p_type[P] == T_BND
        classes.append(dict(size=size, P=P, M=M, A=A, B=B, cross=cross, bnd=bnd))
    # haplotype edit distances of the cross-haplotype pairs of the non-breakend partitions
    job_a = [c["A"][~c["bnd"]][c["cross"][~c["bnd"]]] for c in classes]
    job_b = [c["B"][~c["bnd"]][c["cross"][~c["bnd"]]] for c in classes]
    job_p = [np.broadcast_to(c["P"][:, None], c["A"].shape)[~c["bnd"]][c["cross"][~c["bnd"]]] for c in classes]
    job_two = [np.full(len(a), c["size"] == 2) for a, c in zip(job_a, classes)]
    dist = np.zeros(0, np.float64)
    tc = _clock("pair_enumerate_s", tc)
    if classes and sum(len(a) for a in job_a):
        job_a, job_b, job_p, job_two = (np.concatenate(x) for x in (job_a, job_b, job_p, job_two))
        s_sorted, e_sorted = kstart[order], kend[order]
        seg_lo_all = np.minimum.reduceat(s_sorted, p_start)
        seg_hi_all = np.maximum.reduceat(e_sorted, p_start)
        # two-member partitions only need "<= threshold?"; larger ones get exact values so that the dendrogram
        # above the cut (hence scipy's cluster label order) is the reference's.  Any threshold the reference
        # accepts: a negative one pairs nothing, one beyond 32 bits everything
        k_max = max(min(max(int(threshold), -1), 0xFFFFFFFE), 0)

        def prepare(sel):
            """Pool (one reference window per partition with jobs + the alleles behind them), recipes and thresholds of
            the jobs `sel` (None: all)."""
            ja, jb, jp, jtwo = (job_a, job_b, job_p, job_two) if sel is None else (job_a[sel], job_b[sel], job_p[sel], job_two[sel])
            t0 = _clock_start()
            # one reference window per partition with jobs: [min start - 100, max end + 100) of ALL its members (:45-46,...)
            wp = np.unique(jp)
            first_row = order[p_start[wp]]
            p_contig = np.full(n_parts, -1, np.int64)
            p_contig[wp] = T.key_contig()[first_row]
            L_names = _reference_lengths(reference, T.contigs, p_contig[wp])
            L_part = np.zeros(n_parts, np.int64)
            L_part[wp] = L_names[p_contig[wp]]
            wlo = np.maximum(0, seg_lo_all[wp] - 100)
            whi = np.minimum(L_part[wp], seg_hi_all[wp] + 100)
            t0 = _clock("pair_windows_plan_s", t0)
            pool_w, off_w = _fetch_windows(reference, T.contigs, p_contig[wp], wlo, np.maximum(whi, wlo), False)
            t0 = _clock("pair_windows_fetch_s", t0)
            if not np.array_equal(off_w[1:] - off_w[:-1], np.maximum(whi - wlo, 0)):
                raise ValueError("reference windows shorter than the index says")
            win_base = np.zeros(n_parts + 1, np.int64)
            win_base[wp] = off_w[:-1]
            win_base[-1] = off_w[-1]
            win_lo = np.zeros(n_parts, np.int64)
            win_lo[wp] = wlo
            pieces, extra = _haplotype_pieces(T, kstart, kend, ja, jb, jp, p_type, p_contig, win_base, win_lo, L_part, reference,
                                              seq_split=t1.seqs_nbytes)
            pool = np.concatenate([pool_w] + extra) if extra else pool_w
            # one call for both kinds of pairs (per-pair threshold; 0xFFFFFFFF = exact): one upload of the windows and
            # alleles, one assembly of the haplotype strings, one wavefront pass and one bit-vector pass
            per_pair = np.where(jtwo, np.uint32(k_max), np.uint32(0xFFFFFFFF)).astype(np.uint32)
            _clock("pair_recipes_s", t0)
            return pool, pieces.reshape(-1), per_pair

        J = len(job_a)
        n_chunks = 1 if J < _PAIR_CHUNK_MIN_JOBS else _PAIR_CHUNKS
        if n_chunks == 1:
            pool, pieces, per_pair = prepare(None)
            tc = _clock_start()
            dist = ctx.haplotype_distance_batch_mixed(pool, pieces, per_pair).astype(np.float64)
            tc = _clock("pair_distances_s", tc)
        else:
            # a crowded sample (config 5: ~10^5 pairs): the jobs are cut into chunks of whole partitions and the device
            # computes the distances of chunk i (one worker thread holds the context meanwhile; the call sleeps on a
            # blocking event) while this thread fetches the windows and builds the recipes of chunk i + 1
            from concurrent.futures import ThreadPoolExecutor
            per_part = np.bincount(job_p, minlength=n_parts)
            cuts = np.searchsorted(np.cumsum(per_part), J * np.arange(1, n_chunks) / n_chunks, side="left")
            chunk_of = np.searchsorted(cuts, job_p, side="left")
            dist = np.zeros(J, np.float64)
            with ThreadPoolExecutor(max_workers=1) as worker:
                pending = []
                for c in range(n_chunks):
                    sel = np.flatnonzero(chunk_of == c)
                    if not len(sel):
                        continue
                    pending.append((sel, worker.submit(ctx.haplotype_distance_batch_mixed, *prepare(sel))))
                tc = _clock_start()
                for sel, fut in pending:
                    dist[sel] = fut.result().astype(np.float64)
                tc = _clock("pair_distances_wait_s", tc)
        over = job_two & (dist == float(0xFFFFFFFF))
        dist[over] = k_max + 1  # "more than the threshold" is all that is known, and all that matters
        tc = _clock_start()
    # condensed distance vectors per size class (row-major pairs (i < j), :131-133), then the clusters
    at = 0
    for c in classes:
        size, P, M, A, B, cross, bnd = c["size"], c["P"], c["M"], c["A"], c["B"], c["cross"], c["bnd"]
        cond = np.full(A.shape, float(SAME_HAPLOTYPE_DISTANCE))
        nb = ~bnd
        k = int(cross[nb].sum())
        sub = cond[nb]
        sub[cross[nb]] = dist[at:at + k]
        cond[nb] = sub
        at += k
        if bool(bnd.any()):
            # span-position distance of breakends (:105-117); integer sum / 3000 in float64, as Python's true division
            a, b = A[bnd], B[bnd]
            same = cross[bnd] & ((T.flag[a] & (F_SRC_REV | F_DST_REV)) == (T.flag[b] & (F_SRC_REV | F_DST_REV)))
            d = (np.abs(T.ss[a] - T.ss[b]) + np.abs(T.ds[a] - T.ds[b])).astype(np.float64) / 3000.0
            cond[bnd] = np.where(same, d, float(BREAKEND_MISMATCH_DISTANCE))
        cut = np.where(bnd, 0.3, float(threshold))
        if size == 2:
            # complete linkage of two points: one cluster iff d <= t, else member 0 is label 1, member 1 label 2
            together = cond[:, 0] <= cut
            tp = P[together]
            out.add(tp, np.ones(len(tp)), M[together, 0], M[together, 1], np.full(len(tp), 2))
            ap = P[~together]
            for member in (0, 1):
                out.add(ap, np.full(len(ap), member + 1), M[~together, member], np.full(len(ap), -1), np.ones(len(ap)))
            continue
        labels = np.zeros(M.shape, np.int64)
        for sel, t in ((nb, float(threshold)), (bnd, 0.3)):
            if bool(sel.any()):
                k = int(sel.sum())
                labels[sel] = ctx.linkage_cut_batch(cond[sel].reshape(-1), np.full(k, size, np.uint32), t).astype(np.int64).reshape(k, size)
        # clusters in label order, members in partition order (:136-139)
        o = np.argsort(labels, axis=1, kind="stable")
        lab = np.take_along_axis(labels, o, axis=1).reshape(-1)
        rows = np.take_along_axis(M, o, axis=1).reshape(-1)
        part = np.repeat(P, size)
        new = np.ones(len(lab), bool)
        new[1:] = (part[1:] != part[:-1]) | (lab[1:] != lab[:-1])
        st = np.flatnonzero(new)
        sz = np.diff(np.concatenate((st, [len(lab)])))
        second = np.where(sz >= 2, rows[np.minimum(st + 1, len(rows) - 1)], -1)
        out.add(part[st], lab[st], rows[st], second, sz)

    tc = _clock("pair_clusters_s", tc)
    c_part, first, second, size = out.ordered()
    for bad in size[size > 2].tolist():
        logging.error("Cluster size should be either 1 or 2 but is " + str(bad))
    ok = size <= 2
    res = _rebuild_rows(T, hap, first[ok], second[ok])
    _clock("pair_rebuild_s", tc)
    return res


def _rebuild_rows(T, hap, first, second):
    """Candidates of the clusters (:184-363): coordinates and payload of the first member, reads of both
    concatenated, complete / fully_covered / cutpaste OR-ed, copy number averaged with round() — banker's
    rounding —, genotype 1/1 for pairs, 1/0 or 0/1 for singletons by haplotype; the constructors re-applied."""
    paired = second >= 0
    sec = np.where(paired, second, first)
    out = T.take(first)
    gt = {g: i for i, g in enumerate(out.genotypes)}
    for g in GENOTYPES:
        if g not in gt:
            out.genotypes = list(out.genotypes) + [g]
            gt[g] = len(out.genotypes) - 1
    out.gt = np.where(paired, gt["1/1"], np.where(hap[first] == 1, gt["1/0"], gt["0/1"])).astype(np.uint8)
    # reads = first.reads + second.reads
    cnt1 = T.r_off[first + 1] - T.r_off[first]
    cnt2 = np.where(paired, T.r_off[sec + 1] - T.r_off[sec], 0)
    total = cnt1 + cnt2
    r_off = np.zeros(len(first) + 1, np.int64)
    np.cumsum(total, out=r_off[1:])
    starts = np.stack((T.r_off[first], T.r_off[sec]), axis=1).reshape(-1)
    lens = np.stack((cnt1, cnt2), axis=1).reshape(-1)
    out.with_reads(r_off, T.r_flat[_ranges(starts, lens)])
    # flags OR-ed (INV complete :224, DUP_TAN fully_covered :284, DUP_INT cutpaste :319)
    both = paired & ((out.type == T_INV) | (out.type == T_DUP_TAN) | (out.type == T_DUP_INT))
    out.flag = np.where(both, out.flag | (T.flag[sec] & F_BOOL), out.flag).astype(np.uint8)
    # copies = round(mean([c1, c2])) (:290): exact halves go to the even neighbour
    tan = paired & (out.type == T_DUP_TAN)
    if bool(tan.any()):
        total = T.copies[first] + T.copies[sec]
        floor = total // 2
        rounded = np.where(total % 2 == 0, floor, np.where(floor % 2 == 0, floor, floor + 1))
        out.copies = np.where(tan, rounded, out.copies)
    _apply_constructors(out)
    return out


_LOG_NAME = {"DEL": "deletions", "INV": "inversions", "INS": "insertions", "DUP_TAN": "tandem duplications",
             "DUP_INT": "interspersed duplications", "BND": "breakends"}


def pair_candidates(sv_candidates1, sv_candidates2, reference, bam, options):
    t1, t2 = as_table(sv_candidates1, bam), as_table(sv_candidates2, bam)
    return CandidateList(pair_tables(t1, t2, reference, bam, options))


def pair_haplotypes(partitions, reference, edit_distance_threshold=10, ctx=None):
    """Cluster each partition (lists of (haplotype, candidate)) by complete linkage over haplotype edit
    distances (:120-140): the object-level seam, answered by the columnar pairing."""
    return _pair_partitions(partitions, reference, edit_distance_threshold, ctx)


def pair_haplotypes_breakends(partitions, span_position_distance_threshold=0.3, ctx=None):
    if span_position_distance_threshold != 0.3:
        raise ValueError("the breakend cut is fixed at 0.3 (SVIM_COMBINE.py:143)")
    return _pair_partitions(partitions, None, None, ctx)


def _pair_partitions(partitions, reference, threshold, ctx):
    """Clusters (lists of the input tuples) of already formed partitions, through the same arithmetic as
    pair_tables: every partition is kept apart by giving it its own key contig."""
    ctx = ctx or _lib.default_context()
    clusters_final = []
    for partition in partitions:
        if len(partition) < 2:
            clusters_final.append(list(partition))
            continue
        if len(partition) > 10:
            if partition[0][1].type != "BND":
                logging.debug("Ignored partition of size {0} and type {1}: {2}".format(
                    len(partition), partition[0][1].get_key()[0],
                    ",".join("{0}:{1}".format(m[1].get_key()[1], m[1].get_key()[2]) for m in partition)))
            continue
        n = len(partition)
        if partition[0][1].type == "BND":
            rows = [(hap, c.get_source()[1], c.source_direction, c.get_destination()[1], c.dest_direction) for hap, c in partition]
            cond = [span_position_distance_breakends(rows[i], rows[j]) for i in range(n - 1) for j in range(i + 1, n)]
            cut = 0.3
        else:
            pairs = [(i, j) for i in range(n - 1) for j in range(i + 1, n)]
            cross = [(i, j) for i, j in pairs if partition[i][0] != partition[j][0]]
            d = dict(zip(cross, edit_distances([tuple(haplotype_pair(partition[i][1], partition[j][1], reference))
                                                for i, j in cross], ctx=ctx)))
            cond = [d.get(p, SAME_HAPLOTYPE_DISTANCE) for p in pairs]
            cut = threshold
        labels = ctx.linkage_cut_batch(cond, [n], float(cut)).tolist()
        clusters = [[] for _ in range(max(labels))]
        for member, l in zip(partition, labels):
            clusters[l - 1].append(member)
        clusters_final.extend(clusters)
    return clusters_final


# ------------------------------------------------------------------------------ output
def sorted_nicely(vcf_entries):
    """Natural sort of ((contig, start, end), vcf_string, sv_type) entries: chr10 after chr2."""
    rank = _natural_ranks(set(entry[0][0] for entry in vcf_entries))
    return sorted(vcf_entries, key=lambda entry: (rank[entry[0][0]], entry[0][1], entry[0][2]))


def _natural_ranks(names):
    """Rank of every contig name under the reference's key (:373-375: the name split at digit runs, the runs
    as integers); names whose keys are equal — "chr01" and "chr1" — share a rank."""
    natural = {name: [int(tok) if tok.isdigit() else tok for tok in re.split("([0-9]+)", str(name))] for name in names}
    rank, last, r = {}, None, 0
    for name in sorted(natural, key=natural.__getitem__):
        if last is None or natural[name] != natural[last]:
            r = len(rank)
        rank[name] = r
        last = name
    return rank


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


def _pool_of_strings(strings):
    enc = [s.encode("utf-8", "surrogateescape") for s in strings]
    off = np.zeros(len(enc) + 1, np.int64)
    if enc:
        np.cumsum([len(e) for e in enc], out=off[1:])
    return b"".join(enc), off


def vcf_body(table, types_to_output, reference, options, sink=None):
    """The record lines of write_final_vcf (:428-477) for the rows of `table` as bytes (every line ends with a
    newline): entries in the reference's list order, formatted, naturally sorted and numbered by
    svx_vcf_format.  With `sink` (a binary file object) the lines are written to it instead of being returned."""
    lib = _lib.load()
    t = table
    for k in [k for k in LAST_TIMING if k.startswith("vcf_")]:
        del LAST_TIMING[k]
    tc = _clock_start()
    rows_of = lambda ti: np.flatnonzero(t.type == ti)
    kinds, rows = [], []

    def add(kind, r):
        kinds.append(np.full(len(r), kind, np.uint8))
        rows.append(r)
    if "DEL" in types_to_output:
        add(_lib.VCF_DEL, rows_of(T_DEL))
    if "INV" in types_to_output:
        add(_lib.VCF_INV, rows_of(T_INV))
    if "INS" in types_to_output:
        add(_lib.VCF_INS, rows_of(T_INS))
    if options.tandem_duplications_as_insertions:
        if "INS" in types_to_output:
            add(_lib.VCF_DUPTAN_INS, rows_of(T_DUP_TAN))
    elif "DUP:TANDEM" in types_to_output:
        add(_lib.VCF_DUPTAN_DUP, rows_of(T_DUP_TAN))
    if options.interspersed_duplications_as_insertions:
        if "INS" in types_to_output:
            add(_lib.VCF_DUPINT_INS, rows_of(T_DUP_INT))
    elif "DUP:INT" in types_to_output:
        add(_lib.VCF_DUPINT_DUP, rows_of(T_DUP_INT))
    if "BND" in types_to_output:
        b = rows_of(T_BND)  # two entries per breakend, the mate's right behind it (:463-464)
        kinds.append(np.tile(np.array([_lib.VCF_BND, _lib.VCF_BND_REV], np.uint8), len(b)))
        rows.append(np.repeat(b, 2))
    kind = np.concatenate(kinds) if kinds else np.zeros(0, np.uint8)
    row = np.concatenate(rows).astype(np.int64) if rows else np.zeros(0, np.int64)
    ne = len(kind)
    seq = not options.symbolic_alleles
    bases = np.zeros(0, np.uint8)
    b_off = b_len = b2_off = b2_len = np.zeros(ne, np.int64)
    if seq and ne:
        # the REF alleles the formatters fetch (SVCandidate.py:57,105,155,210,301-302), one batch, upper-cased
        ss, se, ds = t.ss[row], t.se[row], t.ds[row]
        on_dst = (kind == _lib.VCF_INS) | (kind == _lib.VCF_DUPINT_INS)
        wants = (kind == _lib.VCF_DEL) | (kind == _lib.VCF_INV) | (kind == _lib.VCF_DUPTAN_INS) | on_dst
        cid = np.where(on_dst, t.dc[row], t.sc[row])
        lo = np.where(kind == _lib.VCF_DEL, np.maximum(0, ss - 1), np.where(on_dst, np.maximum(0, ds - 1), ss))
        hi = np.where(on_dst, ds, se)
        w = np.flatnonzero(wants)
        w2 = np.flatnonzero(kind == _lib.VCF_DUPINT_INS)
        f_cid = np.concatenate((cid[w], t.sc[row][w2]))
        f_lo = np.concatenate((lo[w], ss[w2]))
        f_hi = np.concatenate((hi[w], se[w2]))
        tc = _clock("vcf_entries_s", tc)
        bases, off = _fetch_windows(reference, t.contigs, f_cid, f_lo, f_hi, True)
        tc = _clock("vcf_fetch_s", tc)
        b_off, b_len = np.zeros(ne, np.int64), np.zeros(ne, np.int64)
        b2_off, b2_len = np.zeros(ne, np.int64), np.zeros(ne, np.int64)
        ln = off[1:] - off[:-1]
        b_off[w], b_len[w] = off[:-1][:len(w)], ln[:len(w)]
        b2_off[w2], b2_len[w2] = off[:-1][len(w):], ln[len(w):]
    if seq:
        tc = _clock("vcf_prepare_s", tc)
        reference.close()  # (:466-467)
        tc = _clock("vcf_reference_close_s", tc)
    if ne == 0:
        return b"" if sink is None else None
    natural = _natural_ranks(set(t.contigs))
    contig_rank = np.array([natural[c] for c in t.contigs], dtype=np.int32)
    contig_pool, contig_off = _pool_of_strings(t.contigs)
    gt_pool, gt_off = _pool_of_strings(t.genotypes)
    keep = []  # arrays the struct points into

    def ptr(a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        keep.append(a)
        return a.ctypes.data if a.size else None
    names_pool = t.names.pool if options.query_names else b""
    keep.extend([contig_pool, gt_pool, names_pool])
    arg = _lib.VcfIn(
        n_rows=len(t), sc=ptr(t.sc, np.int32), ss=ptr(t.ss, np.int64), se=ptr(t.se, np.int64), dc=ptr(t.dc, np.int32),
        ds=ptr(t.ds, np.int64), de=ptr(t.de, np.int64), flag=ptr(t.flag, np.uint8), copies=ptr(t.copies, np.int64),
        gt=ptr(t.gt, np.uint8), q_off=ptr(t.q_off, np.int64), q_len=ptr(t.q_len, np.int64),
        r_off=ptr(t.r_off, np.int64), r_flat=ptr(t.r_flat, np.int64), seqs=ptr(t.seqs, np.uint8), seqs_bytes=t.seqs_nbytes,
        names=C.cast(C.c_char_p(names_pool), C.c_void_p).value, name_off=ptr(t.names.off, np.int64), n_names=len(t.names),
        contigs=C.cast(C.c_char_p(contig_pool), C.c_void_p).value, contig_off=ptr(contig_off, np.int64),
        contig_rank=ptr(contig_rank, np.int32), n_contigs=len(t.contigs),
        genotypes=C.cast(C.c_char_p(gt_pool), C.c_void_p).value, genotype_off=ptr(gt_off, np.int64),
        n_genotypes=len(t.genotypes), n_entries=ne, kind=ptr(kind, np.uint8), row=ptr(row, np.uint32),
        bases=ptr(bases, np.uint8), bases_bytes=len(bases), b_off=ptr(b_off, np.int64), b_len=ptr(b_len, np.int64),
        b2_off=ptr(b2_off, np.int64), b2_len=ptr(b2_len, np.int64), sequence_alleles=1 if seq else 0,
        read_names=1 if options.query_names else 0)
    text, n_bytes, n_lines = C.c_void_p(), C.c_uint64(), C.c_uint64()
    tc = _clock("vcf_prepare_s", tc)
    fd = None
    if sink is not None:
        try:
            sink.flush()
            fd = sink.fileno()
            import fcntl
            # (pwrite ignores its offset on a descriptor opened for appending; anything but a regular file has no offsets)
            if not stat.S_ISREG(os.fstat(fd).st_mode) or (fcntl.fcntl(fd, fcntl.F_GETFL) & os.O_APPEND):
                fd = None
        except (AttributeError, OSError, ValueError):  # not a real file: the buffer form below
            fd = None
    if fd is not None:
        # straight from the formatting threads into the file, each stretch at its final place (svx_vcf_write)
        rc = lib.svx_vcf_write(C.byref(arg), fd, C.byref(n_bytes), C.byref(n_lines))
        if rc != 0:
            raise _lib.SvxError(rc, "svx_vcf_write")
        sink.seek(0, os.SEEK_END)
        _clock("vcf_format_s", tc)
        return None
    rc = lib.svx_vcf_format(C.byref(arg), C.byref(text), C.byref(n_bytes), C.byref(n_lines))
    if rc != 0:
        raise _lib.SvxError(rc, "svx_vcf_format")
    tc = _clock("vcf_format_s", tc)
    try:
        if sink is not None:  # straight from the library's buffer into the file: no bytes object in between
            sink.write(memoryview((C.c_char * n_bytes.value).from_address(text.value)) if n_bytes.value else b"")
            return None
        return C.string_at(text, n_bytes.value)
    finally:
        lib.svx_vcf_free(text)
        _clock("vcf_write_s", tc)


def write_vcf_table(table, version, contig_names, contig_lengths, types_to_output, reference, options, release_reference=True):
    """write_final_vcf for a CandidateTable (rows of each type in the order the reference's per-type lists have).
    release_reference=False: the mappings of the reference genome, closed inside (:466-467), are left to the end of the
    process — what the command does, which exits right behind the VCF."""
    with open(options.working_dir + "/variants.vcf", "wb") as vcf_output:
        header = "".join(line + "\n" for line in _header_lines(version, contig_names, contig_lengths, types_to_output, options))
        vcf_output.write(header.encode("utf-8", "surrogateescape"))
        vcf_body(table, types_to_output, reference, options, sink=vcf_output)
    if release_reference:
        from svim_asm_amd import fasta
        fasta.release_deferred()  # the genome's mapping, closed inside vcf_body (:466-467), goes away behind the file, on a thread


def write_final_vcf(int_duplication_candidates, inversion_candidates, tandem_duplication_candidates,
                    deletion_candidates, insertion_candidates, breakend_candidates, version, contig_names,
                    contig_lengths, types_to_output, reference, options):
    class _Header(object):
        references = tuple(contig_names)

        def get_reference_length(self, name):
            return contig_lengths[list(contig_names).index(name)]
    everything = list(deletion_candidates) + list(inversion_candidates) + list(insertion_candidates) + \
        list(tandem_duplication_candidates) + list(int_duplication_candidates) + list(breakend_candidates)
    table = CandidateTable.from_objects(everything, _Header())
    write_vcf_table(table, version, contig_names, contig_lengths, types_to_output, reference, options)
