"""svx_linkage_cut_batch (complete linkage + flat cut on the GPU) vs the C oracle and scipy itself —
labels in scipy's order, bit for bit.

Reference call sites: SVIM_COMBINE.py:134-135,155-156 and SVIM_inter.py:47-48."""
import numpy as np
import pytest

from oracle import orc
from tests.test_oracle_pins import _scipy_cut, linkage_cases

pytestmark = pytest.mark.gpu


def test_batch_matches_oracle_and_scipy(svx_ctx):
    cases = linkage_cases()
    for cutoff in (0.3, 1.0, 2.5, 200.0):
        sizes = [n for n, _ in cases]
        flat = [d for _, c in cases for d in c]
        got = svx_ctx.linkage_cut_batch(flat, sizes, cutoff)
        at = 0
        for k, (n, cond) in enumerate(cases):
            g = list(got[at:at + n])
            at += n
            assert g == list(orc.linkage_cut(cond, n, cutoff)), (n, cond, cutoff)
            if k % 7 == 0:  # scipy is slow per call: a seventh of the cases, all cut-offs
                assert g == _scipy_cut(cond, cutoff), (n, cond, cutoff)


def test_single_members_and_empty(svx_ctx):
    assert list(svx_ctx.linkage_cut_batch([], [], 0.3)) == []
    assert list(svx_ctx.linkage_cut_batch([], [1, 1, 1], 0.3)) == [1, 1, 1]
    assert list(svx_ctx.linkage_cut_batch([5.0], [1, 2, 1], 10.0)) == [1, 1, 1, 1]
    assert list(svx_ctx.linkage_cut_batch([5.0], [1, 2, 1], 4.0)) == [1, 1, 2, 1]


@pytest.mark.parametrize("n", [11, 12, 25, 60, 200])
def test_large_partitions_use_the_hbm_scratch_path(svx_ctx, n):
    """More members than the per-lane LDS slice holds (inversion groups have no size limit,
    SVIM_inter.py:42-60): same labels as scipy; mixed with small partitions in one launch."""
    rng = np.random.default_rng(n)
    m = n * (n - 1) // 2
    big = [1 - rng.integers(0, 11, m) / 10.0, rng.integers(0, 50, m).astype(float), rng.random(m)]
    small = [(3, [1.0, 2.0, 1.0]), (2, [0.2])]
    sizes, flat = [], []
    for b in big:
        for s_n, s_c in small:
            sizes.append(s_n); flat.extend(s_c)
        sizes.append(n); flat.extend(b.tolist())
    for cutoff in (0.3, 0.75, 20.0):
        got = svx_ctx.linkage_cut_batch(flat, sizes, cutoff)
        at = 0
        fi = 0
        for s in sizes:
            mm = s * (s - 1) // 2
            cond = flat[fi:fi + mm]
            fi += mm
            assert list(got[at:at + s]) == _scipy_cut(cond, cutoff) if s > 1 else [1]
            at += s
