"""Contig sharding across the GPUs of one node.

COLLECT: every alignment (and every chimeric read: its SA-derived segments are reconstructed
from the primary's tag, SVIM_COLLECT.py:76) lives wholly in the contig of its record, so contigs
are independent units.  They are distributed over the ranks by LPT bin packing on their CIGAR
op counts; each rank runs the ordinary batched COLLECT on its contigs with its own GPU and the
per-contig candidate lists are exchanged with one all_gather_object (host concat — a few MB;
no RCCL data-path collective is justified) and re-assembled in header-contig order, which is
the reference's output order (SVIM_COLLECT.py:64).
PAIR: partitions never span key contigs (SVIM_COMBINE.py:24-25), so after the exchange each
rank pairs the key contigs it owns and the results are merged type by type in contig-name
(Python str) order — the order the reference's sort produces.
One process per GPU, launched with torch.distributed.run; world size 1 needs no process group.
"""
import os
import sys

import numpy as np


def _dist():
    # importing torch costs seconds: only look for a process group when one can exist
    if "torch.distributed" not in sys.modules and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return None
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def lpt_assign(weights, n_bins):
    """Longest-processing-time bin packing.  Returns bin index per item; deterministic
    (ties broken by item index) so every rank computes the same plan without communication."""
    order = sorted(range(len(weights)), key=lambda i: (-int(weights[i]), i))
    load = [0] * n_bins
    owner = [0] * len(weights)
    for i in order:
        b = min(range(n_bins), key=lambda k: (load[k], k))
        owner[i] = b
        load[b] += int(weights[i])
    return owner


def contig_weights(bam):
    """Work per contig for the rank plan.  With a usable `.bai` the compressed bytes the index
    attributes to each contig (ingest dominates COLLECT and nothing has to be inflated to know
    them); otherwise CIGAR ops per contig, which needs the whole file indexed first."""
    spans = getattr(bam, "contig_spans", None)
    if spans is not None:
        w = spans()
        if w is not None:
            return w
    cols = getattr(bam, "_cols", None)
    n = len(bam.references)
    w = np.zeros(n, dtype=np.int64)
    if cols is not None:
        tid = cols["tid"]
        ok = (tid >= 0) & (tid < n)
        np.add.at(w, tid[ok], cols["n_cig"][ok] + 1)
        return w
    for i, name in enumerate(bam.references):
        for aln in bam.fetch(contig=name):
            words = getattr(aln, "cigar_words", None)
            w[i] += (len(words) if words is not None else len(aln.cigartuples or ())) + 1
    return w


class ContigView(object):
    """A BAM restricted to a subset of contigs for the COLLECT loop; header lookups see the full
    reference dictionary."""

    def __init__(self, bam, contig_indices):
        self._bam = bam
        self.references = tuple(bam.references[i] for i in contig_indices)
        self.lengths = tuple(bam.lengths[i] for i in contig_indices)

    def __getattr__(self, name):
        return getattr(self._bam, name)


def _gather_or_raise(compute):
    """all_gather_object of compute()'s result.  A rank whose compute() raises still takes part in
    the exchange (with an error marker), so the others do not wait for it until the process-group
    timeout; every rank then raises."""
    d = _dist()
    try:
        mine = ("ok", compute())
    except Exception as e:  # noqa: BLE001 — forwarded to every rank below
        import traceback
        mine = ("error", "rank %d: %s\n%s" % (d.get_rank(), e, traceback.format_exc()))
    gathered = [None] * d.get_world_size()
    d.all_gather_object(gathered, mine)
    errors = [payload for status, payload in gathered if status == "error"]
    if errors:
        raise RuntimeError("sharded step failed on %d of %d ranks:\n%s" % (len(errors), len(gathered), "\n".join(errors)))
    return [payload for _, payload in gathered]


def collect_sharded(bam, options, collect_fn=None):
    """Distributed analyze_alignment_file_coordsorted: same return value on every rank."""
    if collect_fn is None:
        from svim_asm_amd.SVIM_COLLECT import analyze_alignment_file_coordsorted as collect_fn
    rank, size = world()
    if size == 1:
        return collect_fn(bam, options)
    owner = lpt_assign(contig_weights(bam), size)
    mine = [i for i, r in enumerate(owner) if r == rank]

    def local():
        load = getattr(bam, "load", None)
        if load is not None:  # walk / inflate only the BGZF ranges of the contigs this rank owns
            load([bam.references[i] for i in mine])
        # one COLLECT call per owned contig keeps the per-contig lists separable
        return [(i, collect_fn(ContigView(bam, [i]), options)) for i in mine]

    merged = {}
    for part in _gather_or_raise(local):
        for i, cands in part:
            merged[i] = cands
    out = []
    for i in range(len(bam.references)):
        out.extend(merged.get(i, []))
    return out


def pair_sharded(sv_candidates1, sv_candidates2, reference, bam, options, pair_fn=None, type_order=None):
    """Distributed pair_candidates: same return value on every rank."""
    if pair_fn is None:
        from svim_asm_amd.SVIM_COMBINE import pair_candidates as pair_fn
    if type_order is None:
        type_order = ("DEL", "INV", "INS", "DUP_TAN", "DUP_INT", "BND")
    rank, size = world()
    if size == 1:
        return pair_fn(sv_candidates1, sv_candidates2, reference, bam, options)
    key_contig = lambda c: c.get_key()[1]
    contigs = sorted(set(key_contig(c) for c in sv_candidates1) | set(key_contig(c) for c in sv_candidates2))
    counts = {n: 0 for n in contigs}
    for c in list(sv_candidates1) + list(sv_candidates2):
        counts[key_contig(c)] += 1
    owner = dict(zip(contigs, lpt_assign([counts[n] for n in contigs], size)))
    mine1 = [c for c in sv_candidates1 if owner[key_contig(c)] == rank]
    mine2 = [c for c in sv_candidates2 if owner[key_contig(c)] == rank]
    gathered = _gather_or_raise(lambda: pair_fn(mine1, mine2, reference, bam, options))
    everything = [c for part in gathered for c in part]
    out = []
    for typ in type_order:  # reference order: type by type, sorted (contig name, position) inside
        out.extend(sorted((c for c in everything if c.type == typ), key=key_contig))
    return out
