"""HIP banded Needleman-Wunsch (svx_edit_distance_batch) vs a textbook DP.

Reference call sites: edlib.align(h1, h2)["editDistance"], SVIM_COMBINE.py:50,64,76,88,100.
"""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


def mutate(rng, s, n_edits):
    s = bytearray(s)
    for _ in range(n_edits):
        kind = rng.integers(0, 3)
        pos = int(rng.integers(0, len(s) + 1))
        if kind == 0 and len(s) > 0:
            del s[min(pos, len(s) - 1)]
        elif kind == 1:
            s.insert(pos, int(rng.choice(list(b"ACGT"))))
        elif len(s) > 0:
            s[min(pos, len(s) - 1)] = int(rng.choice(list(b"ACGTacgtN")))
    return bytes(s)


def make_pairs(rng, n, max_len, max_edits):
    seqs, pairs = [], []
    for _ in range(n):
        la = int(rng.integers(0, max_len))
        a = bytes(rng.choice(list(b"ACGT"), size=la).astype(np.uint8))
        if rng.random() < 0.2:
            b = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(0, max_len))).astype(np.uint8))
        else:
            b = mutate(rng, a, int(rng.integers(0, max_edits)))
        pairs.append((a, b))
    pool = b"".join(a + b for a, b in pairs)
    a_off, a_len, b_off, b_len = [], [], [], []
    o = 0
    for a, b in pairs:
        a_off.append(o); a_len.append(len(a)); o += len(a)
        b_off.append(o); b_len.append(len(b)); o += len(b)
    return pairs, np.frombuffer(pool, np.uint8), a_off, a_len, b_off, b_len


def test_exact_mode(svx_ctx):
    rng = np.random.default_rng(0)
    pairs, pool, ao, al, bo, bl = make_pairs(rng, 200, 700, 60)
    got = svx_ctx.edit_distance_batch(pool, ao, al, bo, bl)
    exp = np.array([orc.edit_distance(a, b) for a, b in pairs], dtype=np.uint32)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("k", [0, 1, 10, 200])
def test_threshold_mode(svx_ctx, k):
    rng = np.random.default_rng(k + 1)
    pairs, pool, ao, al, bo, bl = make_pairs(rng, 300, 900, 2 * k + 5)
    got = svx_ctx.edit_distance_batch(pool, ao, al, bo, bl, k_max=k)
    exp = np.array([orc.edit_distance(a, b) for a, b in pairs], dtype=np.int64)
    le = exp <= k
    assert np.array_equal(got[le].astype(np.int64), exp[le])
    assert np.all(got[~le].astype(np.int64) > k)
    assert le.any() and (~le).any()


def test_edge_cases(svx_ctx):
    cases = [(b"", b""), (b"", b"ACGT"), (b"ACGT", b""), (b"A", b"A"), (b"A", b"C"), (b"acgt", b"ACGT"),
             (b"ACGT" * 300, b"ACGT" * 300), (b"A" * 1000, b"C" * 1000)]
    pool = b"".join(a + b for a, b in cases)
    ao, al, bo, bl, o = [], [], [], [], 0
    for a, b in cases:
        ao.append(o); al.append(len(a)); o += len(a)
        bo.append(o); bl.append(len(b)); o += len(b)
    got = svx_ctx.edit_distance_batch(np.frombuffer(pool, np.uint8), ao, al, bo, bl)
    assert list(got) == [0, 4, 4, 0, 1, 4, 0, 1000]
