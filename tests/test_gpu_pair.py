"""HIP stable radix sort + partition (svx_pair_partition) vs the CPU oracle.

Reference: form_partitions (SVIM_COMBINE.py:15-32).
"""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


def make_keys(rng, n, n_groups, pos_max, dup_frac=0.3):
    grp = rng.integers(0, n_groups, size=n).astype(np.uint64)
    pos = rng.integers(0, pos_max, size=n).astype(np.uint64)
    if n > 10:
        k = int(n * dup_frac)
        src = rng.integers(0, n, size=k)
        dst = rng.integers(0, n, size=k)
        grp[dst] = grp[src]
        pos[dst] = pos[src]  # exact ties: stability is observable
    return (grp << np.uint64(32)) | pos


@pytest.mark.parametrize("n,groups,pos_max", [(0, 1, 10), (1, 1, 10), (2, 1, 5), (63, 2, 100), (64, 3, 1000),
                                              (1000, 6 * 24, 250_000_000), (1024, 1, 2000), (1025, 50, 1 << 31),
                                              (60_000, 24, 250_000_000), (300_000, 6 * 24, 250_000_000)])
def test_matches_oracle(svx_ctx, n, groups, pos_max):
    rng = np.random.default_rng(n + groups)
    keys = make_keys(rng, n, groups, pos_max)
    for max_dist in (0, 1000):
        perm, part, n_parts = svx_ctx.pair_partition(keys, max_dist)
        e_perm, e_part, e_n = orc.pair_partition(keys, max_dist)
        assert n_parts == e_n
        assert np.array_equal(perm, e_perm)
        assert np.array_equal(part, e_part)


def test_properties_full_size(svx_ctx):
    """Sortedness, permutation, stability and partition-rule properties at diploid human scale."""
    rng = np.random.default_rng(1)
    n = 600_000
    keys = make_keys(rng, n, 6 * 24, 250_000_000)
    perm, part, n_parts = svx_ctx.pair_partition(keys, 1000)
    sk = keys[perm]
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))
    assert np.all(sk[1:] >= sk[:-1])
    ties = sk[1:] == sk[:-1]
    assert np.all(perm[1:][ties] > perm[:-1][ties])  # stable
    brk = ((sk[1:] >> np.uint64(32)) != (sk[:-1] >> np.uint64(32))) | \
          ((sk[1:] & np.uint64(0xFFFFFFFF)) - (sk[:-1] & np.uint64(0xFFFFFFFF)) > np.uint64(1000))
    assert np.array_equal(np.diff(part.astype(np.int64)), brk.astype(np.int64))
    assert part[0] == 0 and n_parts == part[-1] + 1
    # idempotence: sorting the sorted keys is the identity permutation
    perm2, part2, _ = svx_ctx.pair_partition(sk, 1000)
    assert np.array_equal(perm2, np.arange(n, dtype=np.uint32)) and np.array_equal(part2, part)
