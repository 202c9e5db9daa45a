"""HIP split-segment classification (svx_segments_classify) vs the CPU oracle.

Reference: adjacent-pair decision tree of analyze_read_segments (SVIM_inter.py:62-258).
"""
import numpy as np
import pytest

from oracle import orc, svim_oracle
from svim_asm_amd import _lib

pytestmark = pytest.mark.gpu
DEFAULT = (40, 100000, 50, 50, 50, 50)  # min_sv, max_sv, qgt, qot, rgt, rot (SVIM_input_parsing.py defaults)


def random_reads(rng, n_reads, max_k, n_contigs=3, spread=300):
    """Segments engineered to sit near every threshold of the decision tree."""
    ks = rng.integers(0, max_k + 1, size=n_reads)
    off = np.concatenate(([0], np.cumsum(ks))).astype(np.uint32)
    n = int(off[-1])
    segs = np.zeros(n, dtype=_lib.SEG_DTYPE)
    read_len = rng.integers(1000, 200000, size=n_reads).astype(np.int32)
    for r in range(n_reads):
        q = 0
        ref = int(rng.integers(0, 1_000_000))
        for i in range(off[r], off[r + 1]):
            ln = int(rng.integers(100, 5000))
            q_start = q + int(rng.choice([-60, -51, -50, -49, 0, 10, 49, 50, 51, 90, 200, 5000]))
            q_start = max(0, q_start)
            segs[i]["q_start"] = q_start
            segs[i]["q_end"] = q_start + ln
            q = q_start + ln
            jump = int(rng.choice([-200000, -100001, -5000, -300, -51, -50, -49, -40, 0, 39, 40, 41, 50, 51, 300, 5000, 100001, 250000]))
            segs[i]["ref_id"] = int(rng.integers(0, n_contigs)) if rng.random() < 0.2 else 0
            segs[i]["ref_start"] = max(0, ref + jump + int(rng.integers(-spread, spread)))
            segs[i]["ref_end"] = segs[i]["ref_start"] + ln + int(rng.integers(-50, 50))
            segs[i]["is_reverse"] = int(rng.random() < 0.3)
            ref = int(segs[i]["ref_end"])
        # shuffle within the read so that the kernel's stable sort matters; add ties
        sl = slice(off[r], off[r + 1])
        if off[r + 1] - off[r] > 1:
            perm = rng.permutation(off[r + 1] - off[r])
            segs[sl] = segs[sl][perm]
            if rng.random() < 0.3:
                segs[off[r] + 1]["q_start"] = segs[off[r]]["q_start"]
                segs[off[r] + 1]["q_end"] = segs[off[r]]["q_end"]
    return segs, off, read_len


@pytest.mark.parametrize("seed", range(5))
def test_random_reads_match_oracle(svx_ctx, seed):
    rng = np.random.default_rng(seed)
    segs, off, rl = random_reads(rng, n_reads=int(rng.integers(1, 3000)), max_k=int(rng.choice([2, 6, 40, 90])))
    for params in (DEFAULT, (40, 1000, 50, 50, 50, 50), (50, 20, 0, 0, 0, 0), (1, 100000, 500, 500, 500, 500)):
        got = svx_ctx.segments_classify(segs, off, rl, params)
        exp = orc.segments_classify(segs, off, rl, params)
        assert np.array_equal(got, exp)
        kinds = set(np.unique(exp["kind"]))
    assert kinds  # non-trivial


def test_every_branch_family_is_exercised():
    """The generator reaches INS, DEL, BND, TANDEM and INV records (oracle-side check, CPU)."""
    rng = np.random.default_rng(0)
    segs, off, rl = random_reads(rng, 4000, 6)
    exp = orc.segments_classify(segs, off, rl, DEFAULT)
    assert set(np.unique(exp["kind"])) == {0, 1, 2, 3, 4, 5}
    inv = exp[exp["kind"] == 5]
    assert set(np.unique(inv["a3"])) == {0, 1, 2, 3}


def test_empty_and_single_segment_reads(svx_ctx):
    segs = np.zeros(3, dtype=_lib.SEG_DTYPE)
    off = np.array([0, 0, 1, 1, 3], dtype=np.uint32)
    rl = np.array([10, 10, 10, 10], dtype=np.int32)
    got = svx_ctx.segments_classify(segs, off, rl, DEFAULT)
    exp = orc.segments_classify(segs, off, rl, DEFAULT)
    assert np.array_equal(got, exp)
    assert svx_ctx.segments_classify(segs[:0], np.zeros(1, np.uint32), rl[:0], DEFAULT).shape == (0,)


def _random_raw_read(rng, n_slots, n_contigs, crowded):
    """Raw records of one read: TANDEM / BND / INV (and a few INS / DEL / NONE) with coordinates crowded enough
    that tandem groups merge, breakend pairs mirror each other and inversion breakpoints overlap."""
    rows = []
    base = int(rng.integers(1000, 50000))
    span = 60 if crowded else 5000
    for _ in range(n_slots):
        kind = int(rng.choice([0, 1, 2, 3, 3, 4, 4, 5, 5, 5]))
        ref = int(rng.integers(0, n_contigs))
        if kind == 4:
            start = base + int(rng.integers(0, span))
            rows.append((4, ref if not crowded else 0, start, start + int(rng.integers(40, 90)), int(rng.integers(0, 2)),
                         int(rng.integers(0, 2)), 0))
        elif kind == 3 and crowded and rng.random() < 0.5 and any(r[0] == 3 and r[3] == r[6] for r in rows):
            # the mirror image of an earlier breakend: an interspersed-duplication pair (:293-320), or nearly
            q = [r for r in rows if r[0] == 3 and r[3] == r[6]][-1]
            delta = int(rng.integers(20, 500))
            rows.append((3, q[4], q[5] + (delta if q[3] == 0 else -delta), q[3], q[1], q[2] + int(rng.integers(-25, 26)), q[3]))
        elif kind == 3:
            d = int(rng.integers(0, 2))
            other = int(rng.integers(0, 2))
            rows.append((3, ref if not crowded else int(rng.integers(0, 2)), base + int(rng.integers(0, span)), d,
                         int(rng.integers(0, n_contigs)) if not crowded else int(rng.integers(0, 2)),
                         base + 300 + int(rng.integers(0, span)), d if rng.random() < 0.7 else other))
        elif kind == 5:
            start = base + int(rng.integers(0, span * 4))
            rows.append((5, ref if not crowded else int(rng.integers(0, 2)), start, start + int(rng.integers(40, 400)),
                         int(rng.integers(0, 4)), 0, 0))
        else:
            rows.append((kind, ref, base, base + 50, 0, 50, 0))
    return rows


@pytest.mark.parametrize("seed", range(4))
def test_postpass_matches_record_level_oracle(svx_ctx, seed):
    """svx_segments_postpass vs oracle/svim_oracle.postpass_records (SVIM_inter.py:260-338) on random raw
    records: merged and split tandem groups, mirrored breakend pairs, overlapping inversion breakpoints (groups of
    more than 10 members included), contig names whose str order differs from the id order."""
    rng = np.random.default_rng(40 + seed)
    n_contigs = 12
    names = ["chr%d" % (i + 1) for i in range(n_contigs)]  # str order: chr1 chr10 chr11 chr12 chr2 ...
    rank = np.zeros(n_contigs, np.int32)
    for k, i in enumerate(sorted(range(n_contigs), key=lambda i: names[i])):
        rank[i] = k
    reads = [_random_raw_read(rng, int(rng.integers(0, 9)), n_contigs, bool(rng.random() < 0.7)) for _ in range(600)]
    reads += [_random_raw_read(rng, int(rng.integers(12, 40)), n_contigs, True) for _ in range(12)]
    reads.append([])
    read_off = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.uint32)
    raw = np.zeros(int(read_off[-1]), dtype=_lib.RAW_DTYPE)
    k = 0
    for r in reads:
        for row in r:
            for name_, v in zip(("kind", "a0", "a1", "a2", "a3", "a4", "a5"), row):
                raw[k][name_] = v
            k += 1
    prm = (40, 100000, 50, 50, 50, 50) if seed % 2 == 0 else (10, 300, 50, 50, 50, 50)
    post, first = svx_ctx.segments_postpass(raw, read_off, rank, prm)
    code = {1: "TANDEM", 2: "DUP_INT", 3: "INV"}
    width = {"TANDEM": 5, "DUP_INT": 6, "INV": 4}
    kinds = set()
    for i, r in enumerate(reads):
        exp = svim_oracle.postpass_records(r, rank.tolist(), prm[0], prm[1])
        got = []
        for rec in post[first[i]:first[i + 1]]:
            kind = code[int(rec["kind"])]
            vals = [int(rec[n]) for n in ("a0", "a1", "a2", "a3", "a4", "a5")][:width[kind]]
            if kind == "TANDEM":
                vals[4] = bool(vals[4])
            if kind == "INV":
                vals[3] = bool(vals[3])
            got.append((kind,) + tuple(vals))
        assert got == exp, (i, r)
        kinds |= {(t[0], t[-1]) if t[0] == "INV" else (t[0],) for t in exp}
    assert {("TANDEM",), ("DUP_INT",), ("INV", True), ("INV", False)} <= kinds


def test_postpass_capacity_is_checked(svx_ctx):
    import ctypes as C
    raw = np.zeros(3, dtype=_lib.RAW_DTYPE)
    read_off = np.array([0, 3], np.uint32)
    out = np.zeros(2, dtype=_lib.RAW_DTYPE)
    out_off = np.array([0, 2], np.uint64)  # needs svx_segments_postpass_bound(3) = 9
    cnt = np.zeros(1, np.uint32)
    rank = np.zeros(1, np.int32)
    prm = _lib.SegParams(40, 100000, 50, 50, 50, 50)
    assert svx_ctx.lib.svx_segments_postpass_bound(3) == 9
    rc = svx_ctx.lib.svx_segments_postpass(svx_ctx.h, raw.ctypes.data, read_off.ctypes.data, 1, rank.ctypes.data, 1,
                                           C.byref(prm), out.ctypes.data, out_off.ctypes.data, cnt.ctypes.data)
    assert rc == _lib.SVX_E_CAPACITY
