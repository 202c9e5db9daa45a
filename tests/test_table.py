"""Columnar candidates (svim_asm_amd/table.py) against the object model they replace: round trips, the
constructors applied on columns vs the Candidate constructors (SVCandidate.py:40-46,83-89,130-136,181-187,266-280,
352-373), contig re-numbering on concatenation, and the native VCF formatter (svx_vcf_format) line by line against
the per-object get_vcf_entry* formatters."""
import argparse

import numpy as np
import pytest

from svim_asm_amd import SVCandidate, SVIM_COMBINE
from svim_asm_amd.SVIM_COLLECT import _apply_constructors
from svim_asm_amd.table import (CandidateList, CandidateTable, F_BOOL, F_DST_REV, F_SRC_REV, NamePool, T_BND, T_DEL,
                                T_DUP_INT, T_DUP_TAN, T_INS, T_INV, TYPE_ORDER)
from tests import helpers

NAMES = ["chr2", "chr10", "chr1", "chrX_random", "10"]
LENGTHS = [5000, 3000, 8000, 1000, 2500]


def _random_objects(rng, n, seqs):
    bam = helpers.FakeBam(NAMES, LENGTHS, [])
    tuples = helpers.random_candidates(rng, NAMES, LENGTHS, seqs, n, "r")
    return bam, tuples, [helpers.build_candidate(t, bam, SVCandidate) for t in tuples]


@pytest.mark.parametrize("seed", range(3))
def test_objects_round_trip_through_the_table(seed):
    rng = np.random.default_rng(seed)
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=l)) for n, l in zip(NAMES, LENGTHS)}
    bam, tuples, objs = _random_objects(rng, 150, seqs)
    for o in objs[::7]:
        o.reads = o.reads + ["second_read", "third"]      # any number of reads per candidate
        o.genotype = "0/1"
    objs[3].genotype = "./."                                  # a genotype string the pairing step never writes
    t = CandidateTable.from_objects(objs, bam)
    back = t.objects()
    assert [helpers.candidate_tuple(c) for c in back] == [helpers.candidate_tuple(c) for c in objs]
    assert [type(a) is type(b) and a.__dict__ == b.__dict__ for a, b in zip(back, objs)] == [True] * len(objs)
    lst = CandidateList(t)
    assert len(lst) == len(objs) and helpers.candidate_tuple(lst[5]) == helpers.candidate_tuple(objs[5])
    assert [c.type for c in lst] == [c.type for c in objs]
    # take / concat keep the rows and re-express contig ids by NAME
    idx = rng.permutation(len(objs))[:60]
    sub = t.take(idx)
    assert [helpers.candidate_tuple(c) for c in sub.objects()] == [helpers.candidate_tuple(objs[i]) for i in idx]
    other = helpers.FakeBam(NAMES[::-1], LENGTHS[::-1], [])
    t2 = CandidateTable.from_objects(objs[:40], other)           # the same candidates numbered against another header
    both = CandidateTable.concat([t, t2], NAMES, LENGTHS)
    assert [helpers.candidate_tuple(c) for c in both.objects()] == [helpers.candidate_tuple(c) for c in objs + objs[:40]]
    assert both.counts_by_type().tolist() == [sum(1 for c in objs + objs[:40] if c.type == ty) for ty in TYPE_ORDER]


def test_wire_format_round_trip_and_refusals():
    """A table travels between ranks as typed arrays behind a JSON header (CandidateTable.to_wire): it comes back row
    for row; a message that is cut, names a type the format does not know or whose columns disagree is refused —
    nothing in it is ever unpickled or executed."""
    import json
    import struct
    rng = np.random.default_rng(9)
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=l)) for n, l in zip(NAMES, LENGTHS)}
    bam, tuples, objs = _random_objects(rng, 120, seqs)
    objs[5].reads = objs[5].reads + ["another_read"]
    objs[9].genotype = "0/1"
    t = CandidateTable.from_objects(objs, bam)
    t.rec_tid = np.arange(len(t), dtype=np.int64) % len(NAMES)
    blob = t.to_wire()
    assert isinstance(blob, bytes) and b"pickle" not in blob[:200]
    back = CandidateTable.from_wire(blob)
    assert [helpers.candidate_tuple(c) for c in back.objects()] == [helpers.candidate_tuple(c) for c in objs]
    assert np.array_equal(back.rec_tid, t.rec_tid) and back.contigs == t.contigs and back.genotypes == t.genotypes
    empty = CandidateTable.from_wire(CandidateTable(NAMES, LENGTHS).to_wire())
    assert len(empty) == 0 and empty.contigs == NAMES
    with pytest.raises(ValueError):
        CandidateTable.from_wire(blob[:len(blob) // 2])
    (n_head,) = struct.unpack_from("<Q", blob, 0)
    head = json.loads(blob[8:8 + n_head])
    head["arrays"][0][1] = "|O"  # an object array: never
    bad_head = json.dumps(head).encode()
    with pytest.raises(ValueError):
        CandidateTable.from_wire(struct.pack("<Q", len(bad_head)) + bad_head + blob[8 + n_head:])
    head = json.loads(blob[8:8 + n_head])
    head["arrays"][1][2] -= 1  # one column a row short
    bad_head = json.dumps(head).encode()
    with pytest.raises(ValueError):
        CandidateTable.from_wire(struct.pack("<Q", len(bad_head)) + bad_head + blob[8 + n_head:])


@pytest.mark.parametrize("seed", range(4))
def test_constructors_on_columns_equal_the_object_constructors(seed):
    rng = np.random.default_rng(100 + seed)
    bam = helpers.FakeBam(NAMES, LENGTHS, [])
    n = 400
    t = CandidateTable(NAMES, LENGTHS, n, names=NamePool.from_strings(["q%d" % i for i in range(n)]))
    t.r_off, t.r_flat = np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int64)
    exp = []
    for i in range(n):
        ty = int(rng.integers(0, 6))
        c1, c2 = int(rng.integers(0, len(NAMES))), int(rng.integers(0, len(NAMES)))
        a = int(rng.integers(-200, LENGTHS[c1]))      # starts inside the contig (or before it), ends anywhere:
        b = a + int(rng.integers(0, 9000))            # a start beyond the contig end fails the re-construction
        d = int(rng.integers(-200, LENGTHS[c2]))      # in the reference too (end clamped below start)
        e = d + int(rng.integers(0, 9000))
        flag = bool(rng.integers(0, 2))
        t.type[i] = ty
        reads = ["q%d" % i]
        if ty == T_DEL:
            t.sc[i], t.ss[i], t.se[i] = c1, a, b
            exp.append(SVCandidate.CandidateDeletion(NAMES[c1], a, b, reads, bam))
        elif ty == T_INV:
            t.sc[i], t.ss[i], t.se[i], t.flag[i] = c1, a, b, F_BOOL * flag
            exp.append(SVCandidate.CandidateInversion(NAMES[c1], a, b, reads, flag, bam))
        elif ty == T_INS:
            t.dc[i], t.ds[i], t.de[i] = c1, a, b
            exp.append(SVCandidate.CandidateInsertion(NAMES[c1], a, b, reads, "", bam))
        elif ty == T_DUP_TAN:
            t.sc[i], t.ss[i], t.se[i], t.copies[i], t.flag[i] = c1, a, b, 3, F_BOOL * flag
            exp.append(SVCandidate.CandidateDuplicationTandem(NAMES[c1], a, b, 3, flag, reads, bam))
        elif ty == T_DUP_INT:
            t.sc[i], t.ss[i], t.se[i], t.dc[i], t.ds[i], t.de[i] = c1, a, b, c2, d, e
            exp.append(SVCandidate.CandidateDuplicationInterspersed(NAMES[c1], a, b, NAMES[c2], d, e, reads, bam))
        else:
            d1, d2 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            if rng.random() < 0.3:
                c2, d = c1, min(LENGTHS[c1] - 1, a + int(rng.integers(-1, 2)))  # same contig, neighbouring or equal positions: the tie rules
            t.sc[i], t.ss[i], t.dc[i], t.ds[i] = c1, a, c2, d
            t.flag[i] = F_SRC_REV * d1 + F_DST_REV * d2
            exp.append(SVCandidate.CandidateBreakend(NAMES[c1], a, ("fwd", "rev")[d1], NAMES[c2], d, ("fwd", "rev")[d2], reads, bam))
    _apply_constructors(t)
    assert [helpers.candidate_tuple(c) for c in t.objects()] == [helpers.candidate_tuple(c) for c in exp]
    # re-applied (what pair_candidates does to every output candidate, SVIM_COMBINE.py:184-363): same as the objects
    again = [helpers.build_candidate(helpers.candidate_tuple(c), bam, SVCandidate) for c in exp]
    _apply_constructors(t)
    assert [helpers.candidate_tuple(c) for c in t.objects()] == [helpers.candidate_tuple(c) for c in again]


def test_constructor_assertion_and_unknown_contig_are_raised():
    bam = helpers.FakeBam(NAMES, LENGTHS, [])
    t = CandidateTable(NAMES, LENGTHS, 2, names=NamePool.from_strings(["a", "b"]))
    t.r_off, t.r_flat = np.arange(3, dtype=np.int64), np.arange(2, dtype=np.int64)
    t.type[:] = [T_DEL, T_INS]
    t.sc[0], t.ss[0], t.se[0] = 1, 50, 60
    t.dc[1], t.ds[1], t.de[1] = 2, 500, 400   # end < start
    with pytest.raises(AssertionError) as ei:
        _apply_constructors(t)
    with pytest.raises(AssertionError) as eo:
        SVCandidate.CandidateInsertion(NAMES[2], 500, 400, ["b"], "", bam)
    assert str(ei.value) == str(eo.value)
    objs = [SVCandidate.CandidateDeletion("chrUn", 5, 9, ["r"], helpers.FakeBam(["chrUn"], [100], []))]
    t = CandidateTable.from_objects(objs, bam)       # a contig the header of `bam` does not have
    with pytest.raises(KeyError):
        _apply_constructors(t)


@pytest.mark.parametrize("seed", list(range(5)) + [14])
def test_native_vcf_lines_equal_the_object_formatters(seed, monkeypatch):
    if seed >= 10:  # the sort of a crowded sample's entries — by contig, the contigs' stretches side by side — on a small table
        monkeypatch.setenv("SVX_VCF_SORT_SPLIT", "0")
        seed -= 10
    rng = np.random.default_rng(200 + seed)
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=l)) for n, l in zip(NAMES, LENGTHS)}
    # (seed 4: more than 4096 entries — the formatter then works on chunks in several threads)
    bam, tuples, objs = _random_objects(rng, 200 if seed < 4 else 6000, seqs)
    for o in objs[::5]:
        o.reads = o.reads + ["extra"]
        o.genotype = ["1/0", "0/1"][int(rng.integers(0, 2))]
    o = argparse.Namespace(symbolic_alleles=bool(seed & 1), query_names=bool(seed & 2), tandem_duplications_as_insertions=seed == 1,
                           interspersed_duplications_as_insertions=seed == 2)
    types = ["DEL", "INS", "INV", "DUP:TANDEM", "DUP:INT", "BND"]
    ref = helpers.FakeFasta(seqs)
    seq = not o.symbolic_alleles
    by = lambda ty: [c for c in objs if c.type == ty]
    # the reference's entry list (SVIM_COMBINE.py:428-464) from the per-object formatters
    entries = []
    for c in by("DEL"):
        entries.append(((c.source_contig, max(1, c.source_start), c.source_end), c.get_vcf_entry(seq, ref, o.query_names), "DEL"))
    for c in by("INV"):
        entries.append(((c.source_contig, c.source_start + 1, c.source_end), c.get_vcf_entry(seq, ref, o.query_names), "INV"))
    for c in by("INS"):
        entries.append(((c.dest_contig, max(1, c.dest_start), c.dest_end), c.get_vcf_entry(seq, ref, o.query_names), "INS"))
    for c in by("DUP_TAN"):
        if o.tandem_duplications_as_insertions:
            entries.append(((c.source_contig, c.source_start + 1, c.source_end), c.get_vcf_entry_as_ins(seq, ref, o.query_names), "INS"))
        else:
            entries.append(((c.source_contig, c.source_start + 1, c.source_end), c.get_vcf_entry_as_dup(o.query_names), "DUP_TANDEM"))
    for c in by("DUP_INT"):
        if o.interspersed_duplications_as_insertions:
            entries.append(((c.dest_contig, max(1, c.dest_start), c.dest_end), c.get_vcf_entry_as_ins(seq, ref, o.query_names), "INS"))
        else:
            entries.append(((c.source_contig, c.source_start + 1, c.source_end), c.get_vcf_entry_as_dup(o.query_names), "DUP_INT"))
    for c in by("BND"):
        entries.append(((c.source_contig, c.source_start + 1, c.source_start + 2), c.get_vcf_entry(o.query_names), "BND"))
        entries.append(((c.dest_contig, c.dest_start + 1, c.dest_start + 2), c.get_vcf_entry_reverse(o.query_names), "BND"))
    counter, exp = {}, []
    for _, line, label in SVIM_COMBINE.sorted_nicely(entries):
        counter[label] = counter.get(label, 0) + 1
        exp.append(line.replace("PLACEHOLDERFORID", "svim_asm.%s.%d" % (label, counter[label]), 1))
    ordered = by("DEL") + by("INV") + by("INS") + by("DUP_TAN") + by("DUP_INT") + by("BND")
    table = CandidateTable.from_objects(ordered, bam)
    got = SVIM_COMBINE.vcf_body(table, types, helpers.FakeFasta(seqs), o).decode().split("\n")
    assert got[-1] == "" and got[:-1] == exp
    # ... and the form the command uses: the same lines written by the formatting threads straight into a file, each
    # stretch at its final place behind the header the caller wrote (svx_vcf_write)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = d + "/out.vcf"
        with open(path, "wb") as fh:
            fh.write(b"##header line\n#CHROM\n")
            assert SVIM_COMBINE.vcf_body(table, types, helpers.FakeFasta(seqs), o, sink=fh) is None
            fh.write(b"trailer written by the caller\n")
        text = open(path, "rb").read().decode().split("\n")
    assert text[:2] == ["##header line", "#CHROM"] and text[2:-2] == exp and text[-2:] == ["trailer written by the caller", ""]
    if seed == 0:  # a sink opened for appending (pwrite would ignore its offsets) or without a descriptor: the buffer form
        import io
        with tempfile.TemporaryDirectory() as d:
            with open(d + "/a.vcf", "wb") as fh:
                fh.write(b"earlier content\n")
            with open(d + "/a.vcf", "ab") as fh:
                SVIM_COMBINE.vcf_body(table, types, helpers.FakeFasta(seqs), o, sink=fh)
            assert open(d + "/a.vcf", "rb").read().decode().split("\n") == ["earlier content"] + exp + [""]
        mem = io.BytesIO()
        SVIM_COMBINE.vcf_body(table, types, helpers.FakeFasta(seqs), o, sink=mem)
        assert mem.getvalue().decode().split("\n") == exp + [""]
