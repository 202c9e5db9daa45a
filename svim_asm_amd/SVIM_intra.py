"""Intra-alignment SV signatures: CIGAR walk on the GPU.

Mirrors the reference seams analyze_cigar_indel(tuples, min_length) (SVIM_intra.py:8-30) and
analyze_alignment_indel(alignment, bam, query_name, options) (SVIM_intra.py:33-44).  The
arithmetic runs in libsvx.so (svx_cigar_extract, include/svx.h); these wrappers only pack
arguments and build Candidate objects.  The batched entry used by COLLECT is
`extract_indel_signatures` — one launch for all alignments of a BAM file.
"""
import numpy as np

from svim_asm_amd import _lib
from svim_asm_amd.SVCandidate import CandidateDeletion, CandidateInsertion

_TYPE_NAME = ("INS", "DEL")


def pack_cigartuples(tuples):
    """[(op, len), ...] → BAM-native u32 words `len << 4 | op`."""
    if tuples is None or len(tuples) == 0:
        return np.zeros(0, dtype=np.uint32)
    a = np.asarray(tuples, dtype=np.int64).reshape(-1, 2)
    return ((a[:, 1] << 4) | (a[:, 0] & 15)).astype(np.uint32)


def cigar_words_of(alignment):
    """Packed CIGAR of a record: zero-copy for bamio records, packed from cigartuples otherwise."""
    w = getattr(alignment, "cigar_words", None)
    if w is not None:
        return w
    return pack_cigartuples(alignment.cigartuples)


def extract_indel_signatures(cigar, aln_off, ref_start, min_length, ctx=None):
    """Batch form: dict(aln, ref_pos, read_pos, len, type) in (alignment, op) order."""
    ctx = ctx or _lib.default_context()
    return ctx.cigar_extract(cigar, aln_off, ref_start, min_length)


def analyze_cigar_indel(tuples, min_length):
    """Parses CIGAR tuples (op, len) and returns indels with a length >= min_length as
    (pos_ref, pos_read, length, "INS"/"DEL"), exactly like the reference function."""
    words = pack_cigartuples(tuples)
    sig = extract_indel_signatures(words, np.array([0, len(words)], dtype=np.uint64), None, min_length)
    return [(int(r), int(q), int(l), _TYPE_NAME[int(t)])
            for r, q, l, t in zip(sig["ref_pos"], sig["read_pos"], sig["len"], sig["type"])]


def _sequence_slice(alignment, a, b):
    f = getattr(alignment, "seq_slice", None)
    if f is not None:
        return f(a, b)
    return alignment.query_sequence[a:b]


def candidates_from_signatures(alignment, bam, query_name, ref_chr, ref_pos, read_pos, length, typ,
                               ins_sequences=None):
    """Signature rows of ONE alignment (absolute ref_pos) → Candidate objects (SVIM_intra.py:38-43).
    `ins_sequences[i]`, when given, is query_sequence[read_pos[i] : read_pos[i] + length[i]] of row i
    (decoded in one batch by the caller)."""
    out = []
    for i, (start, pr, ln, t) in enumerate(zip(ref_pos.tolist(), read_pos.tolist(), length.tolist(), typ.tolist())):
        if t == _lib.SIG_DEL:
            out.append(CandidateDeletion(ref_chr, start, start + ln, [query_name], bam))
        else:
            seq = ins_sequences[i] if ins_sequences is not None else _sequence_slice(alignment, pr, pr + ln)
            out.append(CandidateInsertion(ref_chr, start, start + ln, [query_name], seq, bam))
    return out


def analyze_alignment_indel(alignment, bam, query_name, options):
    ref_chr = bam.getrname(alignment.reference_id)
    words = cigar_words_of(alignment)
    sig = extract_indel_signatures(words, np.array([0, len(words)], dtype=np.uint64),
                                   np.array([alignment.reference_start], dtype=np.int32), options.min_sv_size)
    return candidates_from_signatures(alignment, bam, query_name, ref_chr, sig["ref_pos"].astype(np.int64),
                                      sig["read_pos"], sig["len"], sig["type"])
