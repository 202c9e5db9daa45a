"""ctypes wrapper of oracle/libsvx_oracle.so (C restatement of the hot path).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package svim_asm_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsvx_oracle.so")

SEG_DTYPE = np.dtype([("q_start", "<i4"), ("q_end", "<i4"), ("ref_id", "<i4"), ("ref_start", "<i4"),
                      ("ref_end", "<i4"), ("is_reverse", "<i4")])
RAW_DTYPE = np.dtype([("kind", "<i4"), ("a0", "<i4"), ("a1", "<i4"), ("a2", "<i4"), ("a3", "<i4"),
                      ("a4", "<i4"), ("a5", "<i4"), ("pad", "<i4")])


def build(force=False):
    src = os.path.join(HERE, "svx_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-shared", "-o", LIB_PATH, src])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(LIB_PATH)
        P = C.c_void_p
        l.orc_cigar_extract.restype = C.c_uint64
        l.orc_cigar_extract.argtypes = [P, P, C.c_uint32, P, C.c_uint32, P, P, P, P, P, C.c_uint64]
        l.orc_cigar_count.restype = C.c_uint64
        l.orc_cigar_count.argtypes = [P, C.c_uint64, C.c_uint32]
        l.orc_cigar_stats.restype = None
        l.orc_cigar_stats.argtypes = [P, P, C.c_uint32, P, P, P, P, P]
        l.orc_segments_classify.restype = None
        l.orc_segments_classify.argtypes = [P, P, C.c_uint32, P, P, P]
        l.orc_pair_partition.restype = C.c_uint32
        l.orc_pair_partition.argtypes = [P, C.c_uint32, C.c_uint32, P, P]
        l.orc_edit_distance.restype = C.c_uint32
        l.orc_edit_distance.argtypes = [P, C.c_uint32, P, C.c_uint32]
        l.orc_edit_distance_banded.restype = C.c_uint32
        l.orc_edit_distance_banded.argtypes = [P, C.c_uint32, P, C.c_uint32]
        l.orc_linkage_cut.restype = None
        l.orc_linkage_cut.argtypes = [P, C.c_uint32, C.c_double, P]
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data if a is not None else None


def cigar_extract(cigar, aln_off, ref_start=None, min_len=40):
    cigar = np.ascontiguousarray(cigar, np.uint32)
    aln_off = np.ascontiguousarray(aln_off, np.uint64)
    rs = None if ref_start is None else np.ascontiguousarray(ref_start, np.int32)
    n_aln = len(aln_off) - 1 if len(aln_off) else 0
    n_ops = int(aln_off[-1]) if n_aln else 0
    cap = int(lib().orc_cigar_count(_p(cigar), n_ops, int(min_len)))
    out = {"aln": np.empty(cap, np.uint32), "ref_pos": np.empty(cap, np.uint32),
           "read_pos": np.empty(cap, np.uint32), "len": np.empty(cap, np.uint32),
           "type": np.empty(cap, np.uint8)}
    n = lib().orc_cigar_extract(_p(cigar), _p(aln_off), n_aln, _p(rs), int(min_len), _p(out["aln"]),
                                _p(out["ref_pos"]), _p(out["read_pos"]), _p(out["len"]),
                                _p(out["type"]), cap)
    assert n == cap
    return out


def cigar_extract_timed(cigar, aln_off, ref_start, min_len, out):
    """One pass of the C loop into pre-sized buffers `out` (from a previous cigar_extract): what
    bench.py's cpu_baseline times — no counting pre-pass, no allocation.  Returns the count."""
    n_aln = len(aln_off) - 1
    return int(lib().orc_cigar_extract(_p(cigar), _p(aln_off), n_aln, _p(ref_start), int(min_len), _p(out["aln"]),
                                       _p(out["ref_pos"]), _p(out["read_pos"]), _p(out["len"]),
                                       _p(out["type"]), len(out["aln"])))


def cigar_stats(cigar, aln_off):
    cigar = np.ascontiguousarray(cigar, np.uint32)
    aln_off = np.ascontiguousarray(aln_off, np.uint64)
    n_aln = len(aln_off) - 1 if len(aln_off) else 0
    keys = ("ref_len", "q_start", "q_end", "read_len", "n_hard")
    out = {k: np.zeros(n_aln, np.uint32) for k in keys}
    lib().orc_cigar_stats(_p(cigar), _p(aln_off), n_aln, *[_p(out[k]) for k in keys])
    return out


def segment_rows(st, seg_src, seg_tid, seg_pos, seg_rev, seg_qend, read_off):
    """The segment table of analyze_read_segments (SVIM_inter.py:66-81) from the CIGAR statistics `st` of
    cigar_stats() (per ALIGNMENT; segment j is alignment seg_src[j]), one Python step per segment:
      forward  q_start = query_alignment_start, q_end = query_alignment_end          (:75-76)
      reverse  q_start = infer_read_length() - query_alignment_end, q_end = ... - query_alignment_start  (:68-73)
      ref_end = reference_end = reference_start + sum{M,D,N,=,X}, 1 when that is 0 (htslib bam_endpos)
    seg_qend[j] >= 0 stands for the query_alignment_end pysam takes from a stored sequence.  Returns (SEG_DTYPE
    rows, infer_read_length() of every read's first segment: the `primary` of :120)."""
    n = len(seg_src)
    segs = np.zeros(n, SEG_DTYPE)
    for j in range(n):
        a = int(seg_src[j])
        qs, qe, rl = int(st["q_start"][a]), int(st["q_end"][a]), int(st["read_len"][a])
        if int(seg_qend[j]) >= 0:
            qe = int(seg_qend[j])
        ref_len = int(st["ref_len"][a])
        row = ((rl - qe, rl - qs) if seg_rev[j] else (qs, qe)) + (
            int(seg_tid[j]), int(seg_pos[j]), int(seg_pos[j]) + (ref_len if ref_len else 1), 1 if seg_rev[j] else 0)
        for name, v in zip(("q_start", "q_end", "ref_id", "ref_start", "ref_end", "is_reverse"), row):
            segs[j][name] = ((v + (1 << 31)) % (1 << 32)) - (1 << 31)  # int32 wrap-around, as the device's registers
    read_len = np.array([int(st["read_len"][int(seg_src[int(read_off[r])])]) if read_off[r + 1] > read_off[r] else 0
                         for r in range(len(read_off) - 1)], np.int64)
    return segs, (((read_len + (1 << 31)) % (1 << 32)) - (1 << 31)).astype(np.int32)


def segments_classify(segs, read_off, read_len, params):
    segs = np.ascontiguousarray(segs, SEG_DTYPE)
    read_off = np.ascontiguousarray(read_off, np.uint32)
    read_len = np.ascontiguousarray(read_len, np.int32)
    prm = np.ascontiguousarray(params, np.int32)
    out = np.zeros(len(segs), RAW_DTYPE)
    n_reads = len(read_off) - 1 if len(read_off) else 0
    if n_reads:
        lib().orc_segments_classify(_p(segs), _p(read_off), n_reads, _p(read_len), _p(prm), _p(out))
    return out


def pair_partition(keys, max_dist):
    keys = np.ascontiguousarray(keys, np.uint64)
    n = len(keys)
    perm = np.empty(n, np.uint32)
    part = np.empty(n, np.uint32)
    n_parts = lib().orc_pair_partition(_p(keys), n, int(max_dist), _p(perm), _p(part))
    return perm, part, int(n_parts)


def edit_distance(a: bytes, b: bytes):
    aa = np.frombuffer(a, np.uint8) if len(a) else np.zeros(0, np.uint8)
    bb = np.frombuffer(b, np.uint8) if len(b) else np.zeros(0, np.uint8)
    return int(lib().orc_edit_distance(_p(aa), len(a), _p(bb), len(b)))


def edit_distance_banded(a: bytes, b: bytes):
    """Exact distance by the band-doubling DP (for sequences too long for the full matrix)."""
    aa = np.frombuffer(a, np.uint8) if len(a) else np.zeros(0, np.uint8)
    bb = np.frombuffer(b, np.uint8) if len(b) else np.zeros(0, np.uint8)
    return int(lib().orc_edit_distance_banded(_p(aa), len(a), _p(bb), len(b)))


def linkage_cut(condensed, n, cutoff):
    """fcluster(linkage(condensed, "complete"), cutoff, "distance") — scipy's label order."""
    cond = np.ascontiguousarray(condensed, np.float64)
    assert len(cond) == n * (n - 1) // 2
    labels = np.zeros(n, np.uint32)
    lib().orc_linkage_cut(_p(cond), int(n), float(cutoff), _p(labels))
    return labels
