"""HIP stable radix sort + partition (svx_pair_partition) vs the CPU oracle.

Reference: form_partitions (SVIM_COMBINE.py:15-32).
"""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["single_launch", "radix", "wait_free"])
def pair_path(request, svx_ctx):
    """Every test of this module runs on all three sort plans: batches up to 131072 candidates in one launch
    (k_pair_single: windows of buckets sorted in LDS), the radix plan (P + 2 launches) that larger batches take —
    forced here with svx_ctx_set_pair_single_launch_max(0) —, and the plan without any wait between workgroups
    inside a launch (radix passes + the partition sweep as two launches) that a call falls back to when a wait of
    the other two runs out — forced with svx_ctx_set_pair_wait_free(1)."""
    svx_ctx.set_pair_single_launch_max(0 if request.param == "radix" else 131072)
    svx_ctx.set_pair_wait_free(request.param == "wait_free")
    yield request.param
    svx_ctx.set_pair_single_launch_max(131072)
    svx_ctx.set_pair_wait_free(False)


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


@pytest.mark.parametrize("n", [1, 16384, 16385, 20000, 65536, 70001])
def test_partition_kernel_forms(svx_ctx, n):
    """Around the single-workgroup / multi-workgroup switch of the partition sweep (16 k), with dense
    partitions (every other element opens one) and with a single partition."""
    rng = np.random.default_rng(n)
    for keys in (make_keys(rng, n, 3, 4 * n + 10), (np.arange(n, dtype=np.uint64) * np.uint64(5)),
                 np.zeros(n, dtype=np.uint64), (np.arange(n, dtype=np.uint64)[::-1].copy() << np.uint64(32))):
        for max_dist in (0, 3, 1000):
            perm, part, n_parts = svx_ctx.pair_partition(keys, max_dist)
            e_perm, e_part, e_n = orc.pair_partition(keys, max_dist)
            assert n_parts == e_n and np.array_equal(perm, e_perm) and np.array_equal(part, e_part)


def test_key_layouts(svx_ctx):
    """Every digit plan: all 64 bits live (8 passes of 8 bits), scattered single bits (more than four
    bit fields: merged), one live bit, the layout _pack_keys produces, keys equal in all but the top bit."""
    rng = np.random.default_rng(3)
    n = 5000
    layouts = [rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n).astype(np.uint64),
               rng.integers(0, 1 << 62, n, dtype=np.uint64) & np.uint64(0x8040201008040201),
               rng.integers(0, 2, n).astype(np.uint64) << np.uint64(40),
               ((rng.integers(0, 6, n).astype(np.uint64) << np.uint64(24) | rng.integers(0, 24, n).astype(np.uint64))
                << np.uint64(32)) | rng.integers(0, 250_000_000, n).astype(np.uint64),
               (rng.integers(0, 2, n).astype(np.uint64) << np.uint64(63)) | np.uint64(12345)]
    for keys in layouts:
        perm, part, n_parts = svx_ctx.pair_partition(keys, 1000)
        e_perm, e_part, e_n = orc.pair_partition(keys, 1000)
        assert n_parts == e_n and np.array_equal(perm, e_perm) and np.array_equal(part, e_part)


def test_device_entry_points(svx_ctx):
    """svx_pair_partition_dev (derives the live key bits itself) and svx_pair_partition_dev_bits (the
    caller names them; a superset is fine) on device buffers, repeated on one context."""
    rng = np.random.default_rng(8)
    for n in (3000, 70000):
        keys = make_keys(rng, n, 6 * 24, 250_000_000)
        d_keys = svx_ctx.dev_array(keys)
        d_perm = svx_ctx.dev_array(nbytes=4 * n)
        d_part = svx_ctx.dev_array(nbytes=4 * n)
        d_np = svx_ctx.dev_array(np.zeros(1, np.uint32))
        zeros = np.zeros(n, np.uint32)
        e_perm, e_part, e_n = orc.pair_partition(keys, 1000)
        superset = (0xFF << 32) | 0xFFFFFFFF
        for rep in range(3):
            for bits in (None, int(np.bitwise_or.reduce(keys)), superset):
                for d in (d_perm, d_part):
                    svx_ctx._check(svx_ctx.lib.svx_dev_upload(svx_ctx.h, d.ptr, zeros.ctypes.data, zeros.nbytes))
                if bits is None:
                    rc = svx_ctx.lib.svx_pair_partition_dev(svx_ctx.h, d_keys.ptr, n, 1000, d_perm.ptr, d_part.ptr,
                                                            d_np.ptr)
                else:
                    rc = svx_ctx.lib.svx_pair_partition_dev_bits(svx_ctx.h, d_keys.ptr, n, 1000, bits, d_perm.ptr,
                                                                 d_part.ptr, d_np.ptr)
                assert rc == 0
                assert int(d_np.download(np.uint32)[0]) == e_n
                assert np.array_equal(d_perm.download(np.uint32), e_perm)
                assert np.array_equal(d_part.download(np.uint32), e_part)


def _check(svx_ctx, keys, max_dist):
    perm, part, n_parts = svx_ctx.pair_partition(keys, max_dist)
    e_perm, e_part, e_n = orc.pair_partition(keys, max_dist)
    assert n_parts == e_n and np.array_equal(perm, e_perm) and np.array_equal(part, e_part)


def test_sample_shaped_input(svx_ctx):
    """The order PAIR hands over: the haplotype-1 list, then the haplotype-2 list, each grouped by type and
    ordered along the genome (svim-asm:133-148) — long runs of one bucket per slice; plus the same keys shuffled."""
    rng = np.random.default_rng(21)
    for n_hap in (30_000, 44_000, 65_536):
        haps = []
        for _ in range(2):
            typ = np.sort(rng.choice(6, n_hap, p=[0.45, 0.45, 0.04, 0.03, 0.02, 0.01])).astype(np.uint64)
            contig = rng.integers(0, 24, n_hap).astype(np.uint64)
            pos = rng.integers(0, 248_000_000, n_hap).astype(np.uint64)
            order = np.lexsort((pos, contig, typ))
            haps.append(((typ[order] << np.uint64(24) | contig[order]) << np.uint64(32)) | pos[order])
        keys = np.concatenate(haps)
        _check(svx_ctx, keys, 1000)
        _check(svx_ctx, rng.permutation(keys), 1000)


@pytest.mark.parametrize("n", [5120, 5121, 9000, 131072, 131073])
def test_crowded_buckets(svx_ctx, n):
    """Windows at and beyond what one workgroup sorts in LDS: all keys in one bucket (equal leading bits),
    two crowded buckets beside sparse ones, equal keys only; and the largest one-launch batch."""
    rng = np.random.default_rng(n)
    low = rng.integers(0, 1 << 12, n).astype(np.uint64)
    one_bucket = (np.uint64(5) << np.uint64(32)) | low
    two = np.where(rng.random(n) < 0.5, np.uint64(3) << np.uint64(40), np.uint64(4) << np.uint64(40)) | low
    two[:50] = (rng.integers(0, 1 << 10, 50).astype(np.uint64) << np.uint64(33)) | low[:50]
    # 99 % of the keys below 2^12 under a 30-bit key range: one leading-bits bucket holds nearly everything
    skewed = np.where(rng.random(n) < 0.99, low, rng.integers(0, 1 << 30, n).astype(np.uint64))
    skewed_groups = skewed | (rng.integers(0, 3, n).astype(np.uint64) << np.uint64(32))
    for keys in (one_bucket, two, np.full(n, 7, dtype=np.uint64), skewed, skewed_groups):
        for max_dist in (0, 1000):
            _check(svx_ctx, keys, max_dist)


def test_boundaries_between_windows(svx_ctx):
    """Partitions that continue across bucket and window borders: one group, positions a few apart over
    the whole 28-bit range, so that every border between windows lies inside a partition or opens one
    depending on max_dist alone."""
    rng = np.random.default_rng(5)
    for n in (3000, 40_000, 100_000):
        pos = np.cumsum(rng.integers(1, 7, n)).astype(np.uint64) * np.uint64((1 << 28) // (7 * n))
        keys = rng.permutation((np.uint64(2) << np.uint64(32)) | pos)
        step = int((1 << 28) // (7 * n))
        for max_dist in (0, step, 3 * step, 6 * step, 1 << 28):
            _check(svx_ctx, keys, max_dist)


def test_four_contexts_sort_at_the_same_time():
    """svx.h: up to four contexts may run the one-launch sort on one device at the same time (each launch keeps
    min(64, CUs / 4) workgroups resident behind its barriers).  Four contexts on four host threads, thirty calls
    each, sizes that fill the grid; every result is checked."""
    import threading
    from svim_asm_amd import _lib
    rng = np.random.default_rng(77)
    cases = []
    for n in (60_000, 88_000, 131_072, 45_000):
        keys = make_keys(rng, n, 6 * 24, 250_000_000)
        cases.append((keys, orc.pair_partition(keys, 1000)))
    ctxs = [_lib.Context(0) for _ in range(4)]
    errors = []

    def work(i):
        try:
            keys, (e_perm, e_part, e_n) = cases[i]
            for _ in range(30):
                perm, part, n_parts = ctxs[i].pair_partition(keys, 1000)
                if not (n_parts == e_n and np.array_equal(perm, e_perm) and np.array_equal(part, e_part)):
                    errors.append("context %d: wrong result" % i)
                    return
        except Exception as e:  # noqa: BLE001 — reported below
            errors.append("context %d: %r" % (i, e))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in ctxs:
        c.close()
    assert not errors, errors
