"""svx_chain_deal (include/svx.h): the host arithmetic that deals the chimeric reads of a submission to the workgroups
of the split-segment chain by CIGAR op count.  No device involved: the table only decides WHICH workgroup computes a
read's rows (SVIM_inter.py:66-81) — tests/test_gpu_collect.py checks that results do not depend on it."""
import numpy as np
import pytest

from svim_asm_amd import _lib


def deal(read_off, seg_src, aln_off):
    lib = _lib.load()
    read_off = np.ascontiguousarray(read_off, np.uint32)
    seg_src = np.ascontiguousarray(seg_src, np.uint32)
    aln_off = np.ascontiguousarray(aln_off, np.uint64)
    n_reads = len(read_off) - 1
    out = np.full(2 * (n_reads + 2), 0xFFFFFFFF, np.uint32)
    n = lib.svx_chain_deal(read_off.ctypes.data, n_reads, seg_src.ctypes.data, aln_off.ctypes.data, out.ctypes.data)
    assert n >= 0
    assert (out[2 * (n + 1):] == 0xFFFFFFFF).all() if n else True  # nothing behind the closing entry is written
    return n, out[:2 * (n + 1)].reshape(-1, 2) if n else None


def sample(rng, n_reads, sigma=1.2, n_aln=None):
    k = rng.integers(1, 4, size=n_reads)
    read_off = np.concatenate(([0], np.cumsum(1 + k))).astype(np.uint32)
    n_segs = int(read_off[-1])
    n_aln = n_aln or 4 * n_reads
    ops = np.exp(rng.normal(np.log(150), sigma, size=n_aln)).astype(np.int64) + 3
    aln_off = np.concatenate(([0], np.cumsum(ops), np.cumsum(ops)[-1] + 3 * np.arange(1, n_segs + 1))).astype(np.uint64)
    seg_src = np.empty(n_segs, np.uint32)
    is_first = np.zeros(n_segs, bool)
    is_first[read_off[:-1]] = True
    seg_src[is_first] = rng.choice(n_aln, size=n_reads, replace=False)
    seg_src[~is_first] = n_aln + np.arange(int((~is_first).sum()))
    return read_off, seg_src, aln_off


def costs(read_off, seg_src, aln_off):
    n = (aln_off[seg_src.astype(np.int64) + 1] - aln_off[seg_src]).astype(np.int64)
    seg = np.where(n <= 8, 8, (n + 127) // 128 * 128)
    return 96 + np.add.reduceat(seg, read_off[:-1].astype(np.int64))


@pytest.mark.parametrize("n_reads", [384, 1000, 5000, 70000])
def test_table_is_a_partition_into_consecutive_ranges_of_equal_cost(n_reads):
    rng = np.random.default_rng(n_reads)
    read_off, seg_src, aln_off = sample(rng, n_reads)
    n, t = deal(read_off, seg_src, aln_off)
    per = min(32, max(2, n_reads // 1024))
    assert n == (n_reads + per - 1) // per
    first = t[:, 0].astype(np.int64)
    assert first[0] == 0 and first[-1] == n_reads and (np.diff(first) >= 0).all()
    assert np.array_equal(t[:, 1], read_off[first])
    cost = costs(read_off, seg_src, aln_off)
    pre = np.concatenate(([0], np.cumsum(cost)))
    per_block = pre[first[1:]] - pre[first[:-1]]
    share = pre[-1] / n
    # a workgroup holds at most its share plus one read (the read that crosses the boundary stays whole)
    assert per_block.max() <= share + cost.max() + 1
    # ... and equal COUNTS would have been worse on a skewed sample
    eq = np.add.reduceat(cost, np.arange(0, n_reads, per))
    assert per_block.max() <= eq.max()


def test_small_submissions_get_no_table():
    rng = np.random.default_rng(1)
    for n_reads in (0, 1, 383):
        read_off, seg_src, aln_off = sample(rng, n_reads) if n_reads else (np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(1, np.uint64))
        n, t = deal(read_off, seg_src, aln_off)
        assert n == 0 and t is None


def test_one_giant_read_leaves_the_workgroups_it_skips_empty():
    rng = np.random.default_rng(2)
    read_off, seg_src, aln_off = sample(rng, 2000, sigma=0.3)
    # make read 700's primary a million ops long
    a = int(seg_src[read_off[700]])
    aln_off = aln_off.copy()
    aln_off[a + 1:] += np.uint64(1_000_000)
    n, t = deal(read_off, seg_src, aln_off)
    first = t[:, 0].astype(np.int64)
    owner = np.searchsorted(first, 700, side="right") - 1
    assert first[owner] == 700 and first[owner + 1] == 701          # alone in its workgroup
    assert (np.diff(first) == 0).sum() > n // 3                       # the workgroups its cost spans hold nothing
    assert first[-1] == 2000 and np.array_equal(t[:, 1], read_off[first])


def test_refuses_null_and_decreasing_offsets():
    lib = _lib.load()
    assert lib.svx_chain_deal(None, 500, None, None, None) == _lib.SVX_E_INVALID
    rng = np.random.default_rng(3)
    read_off, seg_src, aln_off = sample(rng, 500)
    bad = read_off.copy()
    bad[10] = bad[11] + 1
    out = np.zeros(2 * 502, np.uint32)
    assert lib.svx_chain_deal(bad.ctypes.data, 500, seg_src.ctypes.data, aln_off.ctypes.data, out.ctypes.data) == _lib.SVX_E_INVALID
