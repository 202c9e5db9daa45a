"""BAM decoding pinned independently of the product's own writer: every file here is written by
tests/spec_bam_writer.py — BGZF, BAM records and the .bai from the SAM specification alone, no import from
svim_asm_amd — with the awkward cases built in by construction, and read back by
  * the native reader (libsvx.so, svim_asm_amd/bamio.py) with whole-member CRC32 verification (the default) and
    with the opt-out (members inflated only as far as needed),
  * the pure-Python reader of bamio.py,
  * the stub pysam the REAL reference runs on in this repository (oracle/refstub/pysam.py),
all three compared with the LITERAL field values the records were written from (and with the pysam semantics of
SURVEY.md Appendix B: reference_end with N and without reference-consuming ops, query_alignment_start / _end and
infer_read_length with S and H, SA among other tags).  Reference surface: SVIM_COLLECT.py:8-58,61-83."""
import os
import sys

import numpy as np
import pytest

from tests import spec_bam_writer as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, I, D, N, S, H, P, EQ, X = range(9)
REFS = [("chrA", 2_000_000), ("chrB", 500_000), ("chrEmpty", 1000), ("chrC", 90_000_000)]


def seq_of(n, seed=0):
    rng = np.random.default_rng(seed)
    return "".join(rng.choice(list("ACGTN"), size=n, p=[0.24, 0.24, 0.24, 0.24, 0.04]))


def rec(name, tid, pos, cigar, flag=0, mapq=60, tags=None, seq=None, seed=0):
    qlen = sum(l for o, l in cigar if o in (M, I, S, EQ, X))
    return dict(name=name, flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cigar, seq=seq_of(qlen, seed) if seq is None else seq,
                tags=tags or [])


def expected_view(r):
    """What pysam shows for the literal record."""
    cig = r["cigar"]
    ref = sum(l for o, l in cig if o in (M, D, N, EQ, X))
    lead = 0
    for o, l in cig:
        if o == H:
            continue
        if o == S:
            lead += l
        else:
            break
    qal = sum(l for o, l in cig if o in (M, I, EQ, X))
    sa = next((v for t, ty, v in r["tags"] if t == "SA"), None)
    l_seq = len(r["seq"])
    if l_seq:  # pysam: from the stored sequence, minus the trailing soft clips
        trail = 0
        for o, l in reversed(cig):
            if o == H:
                continue
            if o == S:
                trail += l
            else:
                break
        q_end = l_seq - trail
    else:
        q_end = lead + qal
    return dict(name=r["name"], flag=r["flag"], tid=r["tid"], pos=r["pos"], mapq=r["mapq"], cigar=[tuple(x) for x in cig],
                sa=sa, seq=r["seq"], ref_end=(r["pos"] + (ref or 1)) if cig else None, q_start=lead, q_end=q_end,
                read_len=sum(l for o, l in cig if o in (M, I, S, H, EQ, X)))


def view_of(a):
    cig = a.cigartuples or []
    return dict(name=a.query_name, flag=a.flag, tid=a.reference_id, pos=a.reference_start, mapq=a.mapping_quality,
                cigar=[(int(o), int(l)) for o, l in cig], sa=a.get_tag("SA") if a.has_tag("SA") else None,
                seq=a.query_sequence or "", ref_end=a.reference_end, q_start=a.query_alignment_start, q_end=a.query_alignment_end,
                read_len=a.infer_read_length() if cig else 0)


def readers(path):
    from svim_asm_amd import bamio
    sys.path.insert(0, os.path.join(ROOT, "oracle", "refstub"))
    try:
        import importlib
        stub = importlib.import_module("pysam")
    finally:
        sys.path.pop(0)
    yield "native, whole members + CRC32", bamio.AlignmentFile(path)
    yield "native, only as far as needed", bamio.AlignmentFile(path, verify=False)
    yield "native, 1 thread", bamio.AlignmentFile(path, threads=1)
    yield "python reader", bamio.AlignmentFile(path, reader="python")
    yield "stub pysam of the reference", stub.AlignmentFile(path)


ALL_AUX = [("XA", "A", "q"), ("Xc", "c", -7), ("XC", "C", 250), ("Xs", "s", -30000), ("XS", "S", 65000), ("Xi", "i", -2_000_000_000),
           ("XI", "I", 4_000_000_000), ("Xf", "f", 1.5), ("XZ", "Z", "some text; with,punctuation"), ("XH", "H", "1AE301")]
ALL_B = [("B" + s, ("B", s), v) for s, v in (("c", [-1, 2, -3]), ("C", [1, 2, 255]), ("s", [-300, 300]), ("S", [1, 65535]),
                                             ("i", [-70000, 70000, 0]), ("I", [4_000_000_000]), ("f", [0.5, -2.25]))]
SA1 = "chrB,1000,-,100S200M50S,60,3;chrC,5000000,+,30S120M200S,20,0;"


def case_plain():
    return [rec("r0", 0, 100, [(M, 50)])], {}


def case_all_op_codes():
    return [rec("ops", 0, 1000, [(H, 10), (S, 20), (M, 30), (I, 5), (D, 7), (N, 1000), (P, 3), (EQ, 11), (X, 2), (M, 9), (S, 4), (H, 6)])], {}


def case_record_spans_many_members():
    return [rec("a", 0, 10, [(M, 40)]), rec("long", 0, 500, [(S, 100), (M, 3000), (I, 60), (M, 2000), (S, 7)], tags=[("SA", "Z", SA1)], seed=3),
            rec("b", 0, 9000, [(M, 40)])], dict(chunk=150)


def case_members_of_64_bytes():
    recs = [rec("r%d" % i, 0, 100 * i, [(M, 20 + i), (D, 45), (M, 30)], seed=i) for i in range(12)]
    return recs, dict(chunk=64)


def case_hard_clipped_primary_with_sa():
    return [rec("hc", 1, 2000, [(H, 5000), (M, 800), (D, 60), (M, 400), (H, 300)], tags=[("NM", "i", 61), ("SA", "Z", SA1)])], {}


def case_sa_after_every_other_tag_type():
    return [rec("t", 0, 77, [(S, 5), (M, 90)], tags=ALL_AUX + ALL_B + [("SA", "Z", SA1)])], dict(chunk=100)


def case_sa_before_every_other_tag_type():
    return [rec("t", 0, 77, [(S, 5), (M, 90)], tags=[("SA", "Z", SA1)] + ALL_B + ALL_AUX)], {}


def case_sa_between_b_arrays_and_lookalikes():
    # tag VALUES that contain the bytes "SAZ" must not be mistaken for the tag
    return [rec("t", 0, 5, [(M, 30)], tags=[("XZ", "Z", "SAZfake,1,+,10M,0,0;"), ("BC", ("B", "C"), [ord("S"), ord("A"), ord("Z"), 0]),
                                            ("SA", "Z", SA1), ("BS", ("B", "S"), [0x4153, 0x005A])])], {}


def case_secondary_supplementary_unmapped_between_placed():
    return [rec("p1", 0, 100, [(M, 500)]), rec("sec", 0, 150, [(M, 80)], flag=256), rec("sup", 0, 200, [(H, 100), (M, 80)], flag=2048),
            rec("unmapped_with_coordinates", 0, 250, [], flag=4, seq=seq_of(33)), rec("dup_qcfail", 0, 300, [(M, 10)], flag=1024 | 512),
            rec("p2", 0, 400, [(M, 500)], flag=16)], {}


def case_unplaced_records_at_the_end():
    return [rec("p", 3, 80_000_000, [(M, 100)]), rec("u1", -1, -1, [], flag=4, seq=seq_of(50)), rec("u2", -1, -1, [], flag=4 | 1, seq="")], {}


def case_empty_members_mid_file():
    recs = [rec("r%d" % i, 0, 1000 * i, [(M, 100), (I, 50), (M, 100)], seed=i) for i in range(6)]
    return recs, dict(chunk=300, empty_after=(0, 1, 2, 5))


def case_no_stored_sequence():
    return [rec("star", 0, 10, [(S, 10), (M, 100), (D, 40), (M, 100), (S, 30)], seq=""), rec("after", 0, 20, [(M, 10)])], {}


def case_odd_and_tiny_sequence_lengths():
    return [rec("one", 0, 1, [(M, 1)]), rec("three", 0, 2, [(M, 3)]), rec("odd", 0, 3, [(S, 1), (M, 98), (S, 2)])], {}


def case_longest_read_name():
    return [rec("n" * 254, 0, 10, [(M, 30)], tags=[("SA", "Z", SA1)])], {}


def case_cigar_of_70000_operations():
    cig = [(M, 10), (I, 1)] * 35000 + [(M, 5)]
    return [rec("before", 0, 5, [(M, 10)]), rec("long_cigar", 0, 100, cig, tags=[("NM", "i", 3), ("SA", "Z", SA1)], seed=5),
            rec("after", 0, 200, [(M, 10)])], {}


def case_references_without_records():
    return [rec("onlyC", 3, 1234, [(M, 100), (D, 50), (M, 100)])], {}


def case_no_pseudo_bins_no_trailer():
    return [rec("r%d" % i, i % 2, 5000 * (i // 2), [(M, 200)], seed=i) for i in range(0, 8, 1) if True], dict(pseudo_bins=False, n_no_coor=False, sort=True)


def case_block_size_field_split_across_members():
    recs = [rec("r%d" % i, 0, 10 * i, [(M, 40)], seed=i) for i in range(4)]
    return recs, dict(split_block_size=True)


def case_stored_members():
    return [rec("r%d" % i, 0, 100 * i, [(M, 300), (D, 100), (M, 300)], seed=i) for i in range(5)], dict(level=0, chunk=500)


def case_largest_members():
    return [rec("big%d" % i, 3, 1_000_000 * i, [(M, 120_000)], seed=i) for i in range(3)], dict(chunk=65280, level=1)


def case_extreme_mapq_and_flags():
    return [rec("q255", 0, 1, [(M, 10)], mapq=255), rec("q0", 0, 2, [(M, 10)], mapq=0, flag=16 | 1 | 2 | 32 | 64),
            rec("q19", 0, 3, [(M, 10)], mapq=19, flag=128)], {}


def case_same_position_ties_keep_file_order():
    return [rec("t%d" % i, 1, 7777, [(M, 10 + i)], seed=i) for i in range(5)], {}


def case_reference_end_without_reference_ops():
    return [rec("ins_only", 0, 500, [(S, 3), (I, 20), (S, 2)]), rec("n_only", 0, 600, [(M, 5), (N, 100000), (M, 5)])], {}


def case_extra_gzip_subfield_before_bc():
    return [rec("r%d" % i, 0, 50 * i, [(M, 40)], seed=i) for i in range(5)], dict(extra_subfield=True, chunk=200)


def case_bins_of_every_level():
    # records whose spans fall into bins of every level of the UCSC scheme (16 kb ... 512 Mb)
    return [rec("l5", 3, 100, [(M, 100)]), rec("l4", 3, 16000, [(M, 1000)]), rec("l3", 3, 130000, [(M, 5), (N, 20000), (M, 5)]),
            rec("l2", 3, 1_000_000, [(M, 5), (N, 200_000), (M, 5)]), rec("l1", 3, 8_000_000, [(M, 5), (N, 1_000_000), (M, 5)]),
            rec("l0", 3, 60_000_000, [(M, 5), (N, 10_000_000), (M, 5)])], {}


CASES = {k[5:]: v for k, v in globals().items() if k.startswith("case_")}


def write_case(tmp_path, name):
    recs, opt = CASES[name]()
    opt = dict(opt)
    if opt.pop("sort", False):
        recs = sorted(recs, key=lambda r: (r["tid"], r["pos"]))
    cuts = None
    if opt.pop("split_block_size", False):
        # cut two bytes into the block_size field of every record after the first
        off = 4 + 4 + len(("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in REFS)).encode()) + 4 + \
            sum(4 + len(n) + 1 + 4 for n, _ in REFS)
        cuts = []
        for r in recs:
            cuts.append(off + 2)
            off += len(W.encode_record(r))
    path = str(tmp_path / (name + ".bam"))
    W.write_bam(path, REFS, recs, cuts=cuts, **opt)
    return path, recs


@pytest.mark.parametrize("name", sorted(CASES))
def test_every_reader_returns_the_literal_records(tmp_path, name):
    path, recs = write_case(tmp_path, name)
    exp = [expected_view(r) for r in recs]
    placed = [e for e in exp if e["tid"] >= 0]
    for label, f in readers(path):
        got = [view_of(a) for a in f.fetch(until_eof=True)] if "stub" in label else [view_of(f.record(i)) for i in range(len(f))]
        assert len(got) == len(exp), (label, len(got), len(exp))
        for g, e in zip(got, exp):
            if not e["cigar"]:
                e = dict(e, ref_end=g["ref_end"], q_start=g["q_start"], q_end=g["q_end"])  # (pysam: None / 0 for unmapped; not compared)
            assert g == e, (label, e["name"][:30], {k: (g[k], e[k]) for k in g if g[k] != e[k]})
        # per-contig fetch through the index: the placed records of that contig, in file order
        for tid, (ref, _) in enumerate(REFS):
            names = [a.query_name for a in f.fetch(ref)]
            assert names == [e["name"] for e in placed if e["tid"] == tid], (label, ref)
        assert f.check_index()
        assert tuple(f.references) == tuple(n for n, _ in REFS) and tuple(f.lengths) == tuple(l for _, l in REFS)


@pytest.mark.parametrize("name", ["record_spans_many_members", "empty_members_mid_file", "members_of_64_bytes", "largest_members"])
def test_sequence_slices_across_members(tmp_path, name):
    """The inserted-sequence bytes COLLECT asks for (SVIM_intra.py:42): arbitrary [a, b) slices of the stored
    sequence, also where the SEQ field straddles member boundaries and empty members."""
    from svim_asm_amd import bamio
    path, recs = write_case(tmp_path, name)
    rng = np.random.default_rng(1)
    for verify in (None, False):
        f = bamio.AlignmentFile(path, verify=verify)
        idx, lo, hi = [], [], []
        for i, r in enumerate(recs):
            n = len(r["seq"])
            for _ in range(6):
                a = int(rng.integers(0, n + 1))
                b = int(rng.integers(a, n + 1))
                idx.append(i); lo.append(a); hi.append(b)
        got = f.sequence_slices(idx, lo, hi)
        assert list(got) == [recs[i]["seq"][a:b] for i, a, b in zip(idx, lo, hi)]


def test_damage_in_any_member_is_refused_by_default(tmp_path):
    """A flipped bit anywhere in the compressed payload of any member: the default reader (whole members + CRC32,
    as htslib) never returns different records silently."""
    from svim_asm_amd import bamio
    path, recs = write_case(tmp_path, "record_spans_many_members")
    data = bytearray(open(path, "rb").read())
    rng = np.random.default_rng(4)
    good = [view_of(bamio.AlignmentFile(path).record(i)) for i in range(len(recs))]
    refused = 0
    for trial in range(40):
        bad = bytearray(data)
        at = int(rng.integers(100, len(data) - 60))
        bad[at] ^= 1 << int(rng.integers(0, 8))
        p2 = str(tmp_path / ("bad%d.bam" % trial))
        open(p2, "wb").write(bytes(bad))
        open(p2 + ".bai", "wb").write(open(path + ".bai", "rb").read())
        try:
            f = bamio.AlignmentFile(p2)
            got = [view_of(f.record(i)) for i in range(len(f))]
        except (ValueError, OSError):
            refused += 1
            continue
        assert got == good, "a damaged file was read as something else"
    assert refused >= 30  # (a flip inside a gzip header's don't-care bytes may leave the records intact)
