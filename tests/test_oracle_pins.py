"""Pin the CPU oracle (oracle/svx_oracle.c, oracle/svim_oracle.py) to the reference:
its own known-answer vectors, its BAM fixtures, and golden vectors produced by running the
real reference in the build container (oracle/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import orc, run_oracle, svim_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# src/tests/test_intra.py:8-22 of the reference, min_length 30
KNOWN = [
    ([(5, 10), (4, 20), (0, 10), (7, 10), (8, 5), (0, 5), (1, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 50, "INS")]),
    ([(5, 10), (4, 20), (0, 30), (2, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 50, "DEL")]),
    ([(5, 10), (4, 20), (0, 30), (2, 40), (1, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 40, "DEL"), (70, 50, 50, "INS")]),
    ([(5, 10), (4, 20), (0, 30), (1, 40), (2, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 40, "INS"), (30, 90, 50, "DEL")]),
]


def c_oracle_indels(tuples, min_len):
    cig = np.array([(l << 4) | o for o, l in tuples], dtype=np.uint32)
    o = orc.cigar_extract(cig, np.array([0, len(cig)], np.uint64), None, min_len)
    return [(int(r), int(q), int(l), "DEL" if t else "INS")
            for r, q, l, t in zip(o["ref_pos"], o["read_pos"], o["len"], o["type"])]


def test_reference_known_answers_test_intra():
    for tuples, expected in KNOWN:
        assert svim_oracle.analyze_cigar_indel(tuples, 30) == expected
        assert c_oracle_indels(tuples, 30) == expected


def test_n_op_quirk():
    tuples = [(0, 10), (3, 1000), (0, 10), (2, 50)]
    assert svim_oracle.analyze_cigar_indel(tuples, 40) == [(20, 20, 50, "DEL")]
    assert c_oracle_indels(tuples, 40) == [(20, 20, 50, "DEL")]


def test_golden_function_vectors():
    vec = json.load(open(os.path.join(GOLD, "functions.json")))
    for case in vec["analyze_cigar_indel"]:
        tuples = [tuple(t) for t in case["tuples"]]
        exp = [tuple(x) for x in case["out"]]
        assert svim_oracle.analyze_cigar_indel(tuples, case["min_length"]) == exp
        assert c_oracle_indels(tuples, case["min_length"]) == exp
    for case in vec["is_similar"]:  # includes the four asserts of the reference's test_inter.py
        assert bool(svim_oracle.is_similar(*case["args"])) == case["out"]


def test_sa_tag_reconstruction_on_reference_fixtures():
    """tests/test_satag.py of the reference: 3 segments equal to records 2-4; 7-field entry
    skipped; mapq -400 becomes 0."""
    vec = json.load(open(os.path.join(GOLD, "functions.json")))["retrieve_other_alignments"]
    for fn, expected in vec.items():
        recs, names, _ = run_oracle.read_records(os.path.join(GOLD, fn))
        prim = [r for r in recs if not r["flag"] & 0x800]
        assert len(prim) == len(expected)
        for rec, exp_rows in zip(prim, expected):
            got = svim_oracle.retrieve_other_alignments(rec, names)
            assert len(got) == len(exp_rows)
            for g, e in zip(got, exp_rows):
                assert "".join("%d%s" % (l, "MIDNSHP=XB"[o]) for o, l in g["cigar"]) == e["cigarstring"]
                assert (g["tid"], g["pos"], g["flag"], g["mapq"]) == (e["reference_id"], e["reference_start"], e["flag"], e["mapping_quality"])
                assert svim_oracle.reference_end(g) == e["reference_end"]
                assert svim_oracle.query_alignment_start(g) == e["query_alignment_start"]
                assert svim_oracle.query_alignment_end(g) == e["query_alignment_end"]
                assert svim_oracle.infer_read_length(g) == e["infer_read_length"]
    # the literal expectations of test_satag.py
    recs, names, _ = run_oracle.read_records(os.path.join(GOLD, "chimeric_read.bam"))
    assert len(recs) == 4
    supp = svim_oracle.retrieve_other_alignments(recs[0], names)
    assert len(supp) == 3
    for s, r in zip(supp, recs[1:]):
        assert s["cigar"] == r["cigar"] and s["tid"] == r["tid"] and s["pos"] == r["pos"]
        assert s["flag"] == r["flag"] and s["mapq"] == r["mapq"]
        assert svim_oracle.reference_end(s) == svim_oracle.reference_end(r)
        assert svim_oracle.query_alignment_start(s) == svim_oracle.query_alignment_start(r)
        assert svim_oracle.query_alignment_end(s) == svim_oracle.query_alignment_end(r)
    recs, names, _ = run_oracle.read_records(os.path.join(GOLD, "chimeric_read_errors.bam"))
    prim = [r for r in recs if not r["flag"] & 0x800]
    assert len(svim_oracle.retrieve_other_alignments(prim[0], names)) == 2
    one = svim_oracle.retrieve_other_alignments(prim[1], names)
    assert len(one) == 1 and one[0]["mapq"] == 0


def _parse_run(argv):
    flags = ("symbolic_alleles", "tandem_duplications_as_insertions", "interspersed_duplications_as_insertions",
             "query_names")
    pos, kw, i = [], {}, 0
    while i < len(argv):
        a = argv[i]
        if a.startswith("--"):
            k = a[2:]
            if k in flags:
                kw[k] = True
                i += 1
            else:
                v = argv[i + 1]
                kw[k] = int(v) if v.lstrip("-").isdigit() else v
                i += 2
        else:
            pos.append(a)
            i += 1
    return pos, kw


RUNS = json.load(open(os.path.join(GOLD, "config1", "runs.json")))


@pytest.mark.parametrize("name", sorted(RUNS))
def test_oracle_pipeline_reproduces_reference_vcf(name):
    """BAM(s) → VCF through the oracle == the VCF the real reference wrote (config 1)."""
    pos, kw = _parse_run(RUNS[name])
    files = [os.path.join(GOLD, "config1", f) for f in pos[2:]]
    got = run_oracle.vcf_from_files(files[:-1], files[-1], run_oracle.default_options(**kw),
                                    edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    assert got == open(os.path.join(GOLD, "config1", name + ".vcf")).read()


def test_edit_distance_c_vs_python():
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(0, 40))).astype(np.uint8))
        b = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(0, 40))).astype(np.uint8))
        assert orc.edit_distance(a, b) == svim_oracle.edit_distance(a.decode(), b.decode())


def test_c_classify_and_partition_agree_with_python_oracle():
    """The C restatements used on the GPU box (orc.segments_classify, orc.pair_partition) agree
    with the pinned Python oracle on random inputs."""
    from tests import helpers
    rng = np.random.default_rng(5)
    names, lengths = ["chr1", "chr10", "chr2"], [2_000_000, 1_500_000, 1_000_000]
    lens = dict(zip(names, lengths))
    # partition: python form_partitions vs C on packed keys
    cands = helpers.random_candidates(rng, names, lengths, None, 400, "a")
    tagged = [(1 + i % 2, c) for i, c in enumerate(cands)]
    for typ in ("DEL", "INS", "INV", "DUP_TAN", "DUP_INT", "BND"):
        sub = [e for e in tagged if e[1][0] == typ]
        parts = svim_oracle.form_partitions(sub, 1000)
        rank = {n: i for i, n in enumerate(sorted(names))}
        keys = np.array([(rank[svim_oracle.get_key(c)[1]] << 32) | svim_oracle.get_key(c)[2] for _, c in sub], dtype=np.uint64)
        perm, part, n_parts = orc.pair_partition(keys, 1000)
        assert n_parts == len(parts)
        flat = [e for p in parts for e in p]
        assert [sub[i] for i in perm] == flat
        assert list(part) == [pi for pi, p in enumerate(parts) for _ in p]


def test_banded_oracle_equals_full_matrix_oracle():
    """orc_edit_distance_banded (used to check long sequences on the GPU box) against the textbook
    DP, including pairs whose optimal path leaves every narrow band (large indels)."""
    import numpy as np
    from oracle import orc
    rng = np.random.default_rng(1)
    dna = np.frombuffer(b"ACGT", np.uint8)
    for it in range(150):
        a = dna[rng.integers(0, 4, int(rng.integers(0, 700)))].tobytes()
        if rng.random() < 0.3:
            b = dna[rng.integers(0, 4, int(rng.integers(0, 700)))].tobytes()
        else:
            b = bytearray(a)
            for _ in range(int(rng.integers(0, 120))):
                pos = int(rng.integers(0, len(b) + 1))
                kind = int(rng.integers(0, 3))
                if kind == 0 and b:
                    del b[min(pos, len(b) - 1)]
                elif kind == 1:
                    b.insert(pos, int(dna[rng.integers(0, 4)]))
                elif b:
                    b[min(pos, len(b) - 1)] = int(dna[rng.integers(0, 4)])
            if rng.random() < 0.4:
                pos = int(rng.integers(0, len(b) + 1))
                b[pos:pos] = dna[rng.integers(0, 4, int(rng.integers(50, 400)))].tobytes()
            b = bytes(b)
        assert orc.edit_distance_banded(a, b) == orc.edit_distance(a, b)
    assert orc.edit_distance_banded(b"", b"ACG") == 3 and orc.edit_distance_banded(b"ACG", b"") == 3


def test_pipeline_vectors_from_the_real_reference():
    """COLLECT, per-read analyze_read_segments, form_partitions and pair_candidates of the oracle against
    vectors the REAL reference produced (oracle/make_golden.py pipeline)."""
    from tests import helpers
    vec = helpers.load_pipeline_vectors()
    names, lengths = vec["names"], vec["lengths"]
    lens = dict(zip(names, lengths))
    for case in vec["collect"]:
        o = helpers.options(**case["options"])
        recs = case["records"]
        assert svim_oracle.collect(recs, names, lengths, o) == case["out"]
        for pr in case["analyze_read_segments"]:
            rec = recs[pr["record"]]
            supp = [s for s in svim_oracle.retrieve_other_alignments(rec, names) if s["mapq"] >= o.min_mapq]
            assert svim_oracle.analyze_read_segments(rec, supp, names, lens, o) == pr["out"]
    for case in vec["pair"]:
        o = helpers.options(**case["options"])
        seqs = case["seqs"]
        plen = [len(seqs[n]) for n in names]
        fasta = helpers.FakeFasta(seqs)
        t1, t2 = case["t1"], case["t2"]
        for typ, parts in case["form_partitions"].items():
            sub = [(1, t) for t in t1 if t[0] == typ] + [(2, t) for t in t2 if t[0] == typ]
            got = svim_oracle.form_partitions(sub, o.partition_max_distance)
            assert got == [[sub[k] for k in p] for p in parts]
        got = svim_oracle.pair_candidates(t1, t2, fasta.fetch, names, plen, dict(zip(names, plen)), o,
                                          edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
        assert got == case["out"]


def _scipy_cut(cond, cutoff):
    from scipy.cluster.hierarchy import fcluster, linkage
    return list(fcluster(linkage(np.array(cond, dtype=float), method="complete"), cutoff, criterion="distance"))


def linkage_cases():
    """Condensed distance vectors that exercise scipy's tie-breaking: every rank pattern (with ties) for
    n <= 4, a seeded sample of them for n = 5, 6, and random vectors with few distinct values to n = 12;
    each with cut-offs below, between and above the values."""
    import itertools
    out = []
    for n in (2, 3, 4):
        m = n * (n - 1) // 2
        for ranks in itertools.product(range(min(m, 4)), repeat=m):
            out.append((n, [float(r) for r in ranks]))
    rng = np.random.default_rng(5)
    for n in (5, 6):
        m = n * (n - 1) // 2
        for _ in range(1500):
            out.append((n, [float(x) for x in rng.integers(0, int(rng.integers(2, 6)), m)]))
    for _ in range(1500):
        n = int(rng.integers(2, 13))
        m = n * (n - 1) // 2
        kind = int(rng.integers(0, 3))
        if kind == 0:    # edit-distance-like ints with same-haplotype sentinels (SVIM_COMBINE.py:37)
            v = rng.integers(0, 400, m).astype(float)
            v[rng.random(m) < 0.4] = 1000000000.0
        elif kind == 1:  # breakend metric (d1 + d2) / 3000 and the 99999 sentinel (:105-117)
            v = rng.integers(0, 2000, m) / 3000
            v[rng.random(m) < 0.3] = 99999
        else:            # inversion metric 1 - relative overlap, many exact 1.0 (SVIM_inter.py:19-39)
            v = 1 - rng.integers(0, 11, m) / 10.0
        out.append((n, [float(x) for x in v]))
    return out


def test_linkage_cut_matches_scipy():
    """orc_linkage_cut == scipy's fcluster(linkage(y, "complete"), t, "distance") including the label
    order (SVIM_COMBINE.py:134-139,155-160; SVIM_inter.py:47-52)."""
    for n, cond in linkage_cases():
        vals = sorted(set(cond))
        cuts = {vals[0] - 0.5, vals[0], vals[-1], vals[-1] + 1, 0.3, 200.0}
        if len(vals) > 1:
            cuts.add((vals[0] + vals[1]) / 2)
            cuts.add(vals[len(vals) // 2])
        for t in sorted(cuts):
            assert list(orc.linkage_cut(cond, n, t)) == _scipy_cut(cond, t), (n, cond, t)
