"""HIP CIGAR walk (svx_cigar_extract*) vs the CPU oracle — bit-exact, through the C-ABI.

Reference behaviour: analyze_cigar_indel (SVIM_intra.py:8-30) + ref_start add (:36-43).
"""
import numpy as np
import pytest

from oracle import orc
from svim_asm_amd import _lib, synth

pytestmark = pytest.mark.gpu
KEYS = ("aln", "ref_pos", "read_pos", "len", "type")


@pytest.fixture(autouse=True, params=["small_batch", "streaming"])
def cigar_path(request, svx_ctx):
    """Every test of this module runs on both kernel paths: batches up to 2^23 ops in two launches
    (tiles of 1024 ops, k_cigar_finish_small) and the five-launch streaming path (tiles of 4096 ops) that
    larger batches take — forced here with svx_ctx_set_small_batch_ops(0)."""
    svx_ctx.set_small_batch_ops(0 if request.param == "streaming" else 1 << 23)
    yield request.param
    svx_ctx.set_small_batch_ops(1 << 23)


def pack(tuples):
    return np.array([(l << 4) | o for o, l in tuples], dtype=np.uint32)


def assert_same(got, exp):
    for k in KEYS:
        assert len(got[k]) == len(exp[k]), (k, len(got[k]), len(exp[k]))
        assert np.array_equal(got[k], exp[k]), k


# the reference's own known-answer vectors (src/tests/test_intra.py:8-22), min_length 30
KNOWN = [
    ([(5, 10), (4, 20), (0, 10), (7, 10), (8, 5), (0, 5), (1, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 50, "INS")]),
    ([(5, 10), (4, 20), (0, 30), (2, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 50, "DEL")]),
    ([(5, 10), (4, 20), (0, 30), (2, 40), (1, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 40, "DEL"), (70, 50, 50, "INS")]),
    ([(5, 10), (4, 20), (0, 30), (1, 40), (2, 50), (0, 30), (4, 25), (5, 15)], [(30, 50, 40, "INS"), (30, 90, 50, "DEL")]),
]


def test_reference_known_answers(svx_ctx):
    for tuples, indels in KNOWN:
        c = pack(tuples)
        out = svx_ctx.cigar_extract(c, [0, len(c)], None, 30)
        got = [(int(r), int(q), int(l), "DEL" if t else "INS")
               for r, q, l, t in zip(out["ref_pos"], out["read_pos"], out["len"], out["type"])]
        assert got == indels
    # all four as one batch, with an empty alignment in between
    cig = np.concatenate([pack(t) for t, _ in KNOWN])
    lens = [len(t) for t, _ in KNOWN]
    off = np.array([0, lens[0], lens[0], lens[0] + lens[1], sum(lens[:3]), sum(lens)], dtype=np.uint64)
    assert_same(svx_ctx.cigar_extract(cig, off, None, 30), orc.cigar_extract(cig, off, None, 30))


def test_n_op_does_not_advance_reference(svx_ctx):
    # [(0,10),(3,1000),(0,10),(2,50)] -> (20,20,50,"DEL")
    c = pack([(0, 10), (3, 1000), (0, 10), (2, 50)])
    out = svx_ctx.cigar_extract(c, [0, 4], None, 40)
    assert (int(out["ref_pos"][0]), int(out["read_pos"][0]), int(out["len"][0]), int(out["type"][0])) == (20, 20, 50, 1)


def test_empty_inputs(svx_ctx):
    out = svx_ctx.cigar_extract(np.zeros(0, np.uint32), np.zeros(1, np.uint64), None, 40)
    assert all(len(out[k]) == 0 for k in KEYS)
    out = svx_ctx.cigar_extract(np.zeros(0, np.uint32), np.zeros(4, np.uint64), np.zeros(3, np.int32), 40)
    assert all(len(out[k]) == 0 for k in KEYS)


@pytest.mark.parametrize("seed", range(6))
def test_random_ragged_all_opcodes(svx_ctx, seed):
    rng = np.random.default_rng(100 + seed)
    n_aln = int(rng.integers(1, 400))
    cig, off, rs = synth.random_cigar_case(rng, n_aln, max_ops=int(rng.choice([3, 40, 700, 9000])))
    for min_len in (1, 40):
        assert_same(svx_ctx.cigar_extract(cig, off, rs, min_len), orc.cigar_extract(cig, off, rs, min_len))


@pytest.mark.parametrize("seed", range(4))
def test_lengths_up_to_28_bits_mix_fast_and_generic_rounds(svx_ctx, seed):
    """The packed-word walk uses 24-bit multiply-adds when every length of a 1024-op round is
    < 2^24 and the generic masked adds otherwise: sprinkle lengths up to the BAM maximum
    (2^28 - 1) so both kinds of rounds occur inside the same tiles; cursors wrap mod 2^32 exactly
    like the oracle's uint32 arithmetic."""
    rng = np.random.default_rng(900 + seed)
    n_aln = 300
    cig, off, rs = synth.random_cigar_case(rng, n_aln, max_ops=200)
    n = len(cig)
    big = rng.random(n) < (0.0005 if seed < 2 else 0.02)  # rare (most rounds fast) / common
    lens = np.where(big, rng.integers(1 << 24, 1 << 28, size=n), cig >> 4).astype(np.uint32)
    lens[rng.integers(0, n, size=5)] = (1 << 28) - 1
    lens[rng.integers(0, n, size=5)] = 1 << 24
    lens[rng.integers(0, n, size=5)] = (1 << 24) - 1
    cig = (lens << 4) | (cig & 15)
    for min_len in (40, 1 << 24, (1 << 28) - 1):
        assert_same(svx_ctx.cigar_extract(cig, off, rs, min_len), orc.cigar_extract(cig, off, rs, min_len))


@pytest.mark.parametrize("min_len", [0, 1, 2, (1 << 28) - 1, 1 << 28, (1 << 32) - 1])
def test_min_len_edge_values(svx_ctx, min_len):
    """min_len 0 emits every I/D op including zero-length ones (len >= 0); min_len above the
    28-bit length range emits nothing."""
    rng = np.random.default_rng(77)
    cig, off, rs = synth.random_cigar_case(rng, 120, max_ops=300)
    lens = (cig >> 4).copy()
    lens[rng.integers(0, len(cig), size=200)] = 0          # zero-length ops of every kind
    lens[rng.integers(0, len(cig), size=20)] = (1 << 28) - 1
    cig = (lens.astype(np.uint32) << 4) | (cig & 15)
    got, exp = svx_ctx.cigar_extract(cig, off, rs, min_len), orc.cigar_extract(cig, off, rs, min_len)
    assert_same(got, exp)
    if min_len >= 1 << 28:
        assert len(got["aln"]) == 0
    if min_len == 0:
        assert len(got["aln"]) == int(np.isin(cig & 15, (1, 2)).sum())


@pytest.mark.parametrize("n_ops", [1, 15, 16, 17, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 12289])
def test_tile_and_round_boundaries(svx_ctx, n_ops):
    rng = np.random.default_rng(n_ops)
    ops = rng.integers(0, 3, size=n_ops)
    lens = rng.integers(30, 60, size=n_ops)
    cig = ((lens << 4) | ops).astype(np.uint32)
    # one long alignment, then the same ops cut at awkward places (starts at tile/lane edges)
    for cuts in ([], [1], [n_ops // 2], [15, 16, 17, 1024, 4096]):
        cuts = sorted(set(c for c in cuts if 0 < c < n_ops))
        off = np.array([0] + cuts + [n_ops], dtype=np.uint64)
        rs = np.arange(len(off) - 1, dtype=np.int32) * 1000
        assert_same(svx_ctx.cigar_extract(cig, off, rs, 40), orc.cigar_extract(cig, off, rs, 40))


def test_dense_all_indel_tiles_take_the_direct_path(svx_ctx):
    rng = np.random.default_rng(7)
    cig, off, rs = synth.random_cigar_case(rng, 37, max_ops=3000, dense=True)
    exp = orc.cigar_extract(cig, off, rs, 40)
    assert len(exp["aln"]) == len(cig)  # every op emits
    assert_same(svx_ctx.cigar_extract(cig, off, rs, 40), exp)
    # mixed: dense alignments between sparse ones
    c2, o2, r2 = synth.random_cigar_case(rng, 50, max_ops=5000)
    cig3 = np.concatenate([c2, cig, c2])
    off3 = np.concatenate([o2, o2[-1] + off[1:], o2[-1] + off[-1] + o2[1:]]).astype(np.uint64)
    rs3 = np.concatenate([r2, rs, r2])
    assert_same(svx_ctx.cigar_extract(cig3, off3, rs3, 40), orc.cigar_extract(cig3, off3, rs3, 40))


def test_many_empty_alignments(svx_ctx):
    rng = np.random.default_rng(11)
    cig, _, _ = synth.random_cigar_case(rng, 1, max_ops=6000)
    cig = cig[:5000] if len(cig) >= 5000 else np.resize(cig, 5000)
    starts = np.sort(rng.integers(0, 5001, size=3000))
    off = np.concatenate(([0], starts, [5000])).astype(np.uint64)
    rs = rng.integers(0, 1 << 20, size=len(off) - 1).astype(np.int32)
    assert_same(svx_ctx.cigar_extract(cig, off, rs, 40), orc.cigar_extract(cig, off, rs, 40))


def test_capacity_protocol(svx_ctx):
    import ctypes as C
    rng = np.random.default_rng(3)
    cig, off, rs = synth.random_cigar_case(rng, 20, max_ops=500, dense=True)
    n_true = len(orc.cigar_extract(cig, off, rs, 40)["aln"])
    cap = 10
    bufs = [np.full(cap + 8, 0xAB, dt) for dt in (np.uint32, np.uint32, np.uint32, np.uint32, np.uint8)]
    soa = _lib.SigSoa(*[b.ctypes.data for b in bufs])
    n = C.c_uint64(0)
    rc = svx_ctx.lib.svx_cigar_extract(svx_ctx.h, cig.ctypes.data, off.ctypes.data, len(off) - 1,
                                       rs.ctypes.data, 40, soa, cap, C.byref(n))
    assert rc == _lib.SVX_E_CAPACITY and n.value == n_true
    assert all((b[cap:] == 0xAB).all() for b in bufs), "wrote past cap"
    exp = orc.cigar_extract(cig, off, rs, 40)
    assert np.array_equal(bufs[1][:cap], exp["ref_pos"][:cap])


def test_soa_input_variant(svx_ctx):
    rng = np.random.default_rng(5)
    cig, off, rs = synth.random_cigar_case(rng, 120, max_ops=3000, all_ops=False)
    op = (cig & 15).astype(np.uint8)
    ln = (cig >> 4).astype(np.uint32)
    assert_same(svx_ctx.cigar_extract(ln, off, rs, 40, op=op), orc.cigar_extract(cig, off, rs, 40))


@pytest.mark.parametrize("n_ops", [1, 2, 3, 5, 1427, 4099, 17647])
@pytest.mark.parametrize("min_len", [0, 1, 40])
def test_soa_ragged_tail_not_multiple_of_four(svx_ctx, n_ops, min_len):
    """The SoA op bytes are fetched as dwords: the last 1-3 ops of a batch whose length is not a
    multiple of 4 must still be seen (found by tools/fuzz_cigar.py), and the bytes after them must
    not turn into zero-length signatures when min_len is 0."""
    rng = np.random.default_rng(n_ops)
    ops = rng.integers(1, 3, size=n_ops).astype(np.uint8)  # all I/D: the tail always emits
    lens = rng.integers(0, 90, size=n_ops).astype(np.uint32)
    off = np.array([0, n_ops // 2, n_ops], dtype=np.uint64)
    rs = np.array([100, 7], dtype=np.int32)
    cig = (lens << 4) | ops
    exp = orc.cigar_extract(cig, off, rs, min_len)
    assert_same(svx_ctx.cigar_extract(lens, off, rs, min_len, op=ops), exp)
    assert_same(svx_ctx.cigar_extract(cig, off, rs, min_len), exp)


def test_soa_op_codes_above_15_are_noops(svx_ctx):
    """SoA op bytes can hold 16..255 (not representable in BAM): like codes 10-15 they advance
    nothing and emit nothing; rounds containing them take the generic walk, the others the packed one."""
    rng = np.random.default_rng(5)
    n = 20000
    ops = rng.integers(0, 3, size=n).astype(np.uint8)
    lens = rng.integers(30, 60, size=n).astype(np.uint32)
    weird = rng.random(n) < 0.001          # a few rounds only
    ops[weird] = rng.integers(16, 256, size=int(weird.sum())).astype(np.uint8)
    off = np.array([0, 7000, 7000, n], dtype=np.uint64)
    rs = np.array([5, 6, 7], dtype=np.int32)
    got = svx_ctx.cigar_extract(lens, off, rs, 40, op=ops)
    # oracle on the packed form with the weird ops mapped to a no-op code (9 = B)
    cig = (lens << 4) | np.where(weird, 9, ops).astype(np.uint32)
    assert_same(got, orc.cigar_extract(cig, off, rs, 40))


def test_invalid_offsets_rejected(svx_ctx):
    cig = pack([(0, 10), (1, 50)])
    with pytest.raises(_lib.SvxError):
        svx_ctx.cigar_extract(cig, np.array([0, 2, 1], np.uint64), None, 40)
    with pytest.raises(_lib.SvxError):
        svx_ctx.cigar_extract(cig, np.array([1, 2], np.uint64), None, 40)


@pytest.mark.parametrize("cfg", [dict(seed=2, mean_m=4000), dict(seed=5, mean_m=400)])
def test_full_size_configs_match_oracle(svx_ctx, cfg):
    """BASELINE configs 2 (≈1.5 M ops) and 5 (≈15 M ops): whole-batch bit-exact + properties."""
    b = synth.synth_cigar_batch(**cfg)
    got = svx_ctx.cigar_extract(b["cigar"], b["aln_off"], b["ref_start"], 40)
    exp = orc.cigar_extract(b["cigar"], b["aln_off"], b["ref_start"], 40)
    assert_same(got, exp)
    # size-independent properties: (alignment, op) order, thresholds, cursor monotonicity per alignment
    assert np.all(np.diff(got["aln"].astype(np.int64)) >= 0)
    assert np.all(got["len"] >= 40)
    same = np.diff(got["aln"].astype(np.int64)) == 0
    assert np.all(np.diff(got["read_pos"].astype(np.int64))[same] >= 0)
    assert np.all(np.diff(got["ref_pos"].astype(np.int64))[same] >= 0)


@pytest.mark.parametrize("cfg", [dict(seed=31, mean_m=2000, sv_frac=0.015, ops_target=5_000_000),
                                 dict(seed=32, mean_m=2000, sv_frac=0.09, ops_target=3_000_000),
                                 dict(seed=33, mean_m=200, sv_frac=0.5, ops_target=8_000_000)])
def test_diploid_sized_and_sv_dense_batches(svx_ctx, cfg):
    """Both haplotype BAMs of a diploid sample in one batch (2-8 M ops: on the two-launch path every workgroup
    of k_cigar_finish_small folds its share of up to 8192 descriptors in four chunks), at the SV density of the
    full-size synthetic sample's small contigs (a fifth of the tiles leave the staged path: rounds with more
    signatures than the queue holds, tiles beyond the slab) and at satellite density (every tile dense)."""
    b = synth.synth_cigar_batch(**cfg)
    assert (1 << 21) < len(b["cigar"]) < (1 << 23)
    got = svx_ctx.cigar_extract(b["cigar"], b["aln_off"], b["ref_start"], 40)
    assert_same(got, orc.cigar_extract(b["cigar"], b["aln_off"], b["ref_start"], 40))


def test_cohort_batch_equals_per_sample_concat(svx_ctx):
    """Linearity over the batch axis: one launch over K samples == K launches concatenated."""
    bs = [synth.synth_cigar_batch(seed=s, ops_target=200_000) for s in (21, 22, 23)]
    big = synth.concat_batches(bs)
    got = svx_ctx.cigar_extract(big["cigar"], big["aln_off"], big["ref_start"], 40)
    parts = [svx_ctx.cigar_extract(b["cigar"], b["aln_off"], b["ref_start"], 40) for b in bs]
    base = 0
    for b, p in zip(bs, parts):
        p["aln"] = p["aln"] + np.uint32(base)
        base += len(b["aln_off"]) - 1
    for k in KEYS:
        assert np.array_equal(got[k], np.concatenate([p[k] for p in parts])), k


def test_cigar_stats(svx_ctx):
    rng = np.random.default_rng(9)
    cig, off, _ = synth.random_cigar_case(rng, 300, max_ops=700)
    got = svx_ctx.cigar_stats(cig, off)
    exp = orc.cigar_stats(cig, off)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    # leading clips longer than one wave
    c = pack([(5, 3)] + [(4, 2)] * 150 + [(0, 10), (4, 7)])
    got = svx_ctx.cigar_stats(c, [0, len(c)])
    assert (int(got["q_start"][0]), int(got["q_end"][0]), int(got["read_len"][0]), int(got["n_hard"][0])) == (300, 310, 320, 3)


def test_device_pointer_entry_points(svx_ctx):
    """svx_cigar_extract_dev / svx_cigar_extract_soa_dev / svx_cigar_stats_dev on buffers the caller owns in HBM
    (svx_dev_malloc: no torch involved), capacity protocol included: exactly min(count, cap) rows are written."""
    import ctypes as C
    rng = np.random.default_rng(21)
    cig, off, rs = synth.random_cigar_case(rng, 700, max_ops=400)
    n_ops, n_aln = len(cig), len(off) - 1
    exp = orc.cigar_extract(cig, off, rs, 40)
    n_exp = len(exp["aln"])
    assert n_exp > 10
    d_cig, d_off, d_rs = svx_ctx.dev_array(cig), svx_ctx.dev_array(off.astype(np.uint64)), svx_ctx.dev_array(rs)
    d_len = svx_ctx.dev_array((cig >> 4).astype(np.uint32))
    d_op = svx_ctx.dev_array(np.concatenate(((cig & 15).astype(np.uint8), np.zeros(16, np.uint8))))
    for cap in (n_exp + 5, n_exp // 2):
        outs = [svx_ctx.dev_array(nbytes=4 * max(cap, 1)) for _ in range(4)] + [svx_ctx.dev_array(nbytes=max(cap, 1))]
        d_n = svx_ctx.dev_array(np.zeros(1, np.uint64))
        for soa in (False, True):
            sig = _lib.SigSoa(*[o.ptr for o in outs])
            if soa:
                rc = svx_ctx.lib.svx_cigar_extract_soa_dev(svx_ctx.h, d_op.ptr, d_len.ptr, n_ops, d_off.ptr, n_aln, d_rs.ptr,
                                                           40, sig, cap, d_n.ptr)
            else:
                rc = svx_ctx.lib.svx_cigar_extract_dev(svx_ctx.h, d_cig.ptr, n_ops, d_off.ptr, n_aln, d_rs.ptr, 40, sig,
                                                       cap, d_n.ptr)
            assert rc == 0
            assert int(d_n.download(np.uint64)[0]) == n_exp  # the full count, whatever the capacity
            k = min(cap, n_exp)
            for o, key in zip(outs, KEYS):
                got = o.download(np.uint8 if key == "type" else np.uint32, k)
                assert np.array_equal(got, exp[key][:k]), (key, cap, soa)
    # per-alignment statistics on device buffers
    st = {k: svx_ctx.dev_array(nbytes=4 * n_aln) for k in ("ref_len", "q_start", "q_end", "read_len", "n_hard")}
    stats = _lib.AlnStats(*[st[k].ptr for k in ("ref_len", "q_start", "q_end", "read_len", "n_hard")])
    assert svx_ctx.lib.svx_cigar_stats_dev(svx_ctx.h, d_cig.ptr, n_ops, d_off.ptr, n_aln, stats) == 0
    e = orc.cigar_stats(cig, off)
    for k in st:
        assert np.array_equal(st[k].download(np.uint32), e[k]), k
