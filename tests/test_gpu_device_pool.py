"""The reader's own upload of its CIGAR pool (svx_bam_device_pool, include/svx_bam.h) and svx_collect_batch taking a
part where it lies in HBM (svx_collect_in.part_dev / part_ready, include/svx.h): same words on the device as in the
page-locked pool, same results as the submission that uploads the pool itself, across reloads of the handle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from svim_asm_amd import SVIM_COLLECT, SVIM_inter, bamio, synth_bam
from tests import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("devpool"))
    # short match runs: 184 k CIGAR ops per haplotype — more than one part of the reader's upload (128 k words each)
    contigs = (("chrA", 900000), ("chrB", 600000), ("chrC", 400000))
    fa, bams = synth_bam.write_dataset(d, seed=23, contigs=contigs, n_shared=14, n_private=4, median_aln=120000, mean_m=12)
    return fa, bams


def device_words(address, n):
    """n uint32 words at a device address, copied back with the HIP runtime itself."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty(n, np.uint32)
    assert hip.hipMemcpy(out.ctypes.data, C.c_void_p(address), n * 4, 2) == 0  # hipMemcpyDeviceToHost
    return out


def test_pool_is_in_hbm_when_the_load_returns(svx_ctx, dataset):
    fa, bams = dataset
    f = bamio.AlignmentFile(bams[0], device=0)
    f.load(None)
    assert f.cigar_pinned and len(f._cigar) > 100000
    pool = f.device_pool()
    assert pool is not None and pool[0] and pool[1]
    address, _none, waited_us = f.device_pool(wait=True)
    assert address == pool[0] and waited_us < 5000.0
    assert np.array_equal(device_words(address, len(f._cigar)), f._cigar)
    # a reload of other contigs: the pool changes, the device copy follows it
    f.load(["chrB"])
    again = f.device_pool(wait=True)
    assert again is not None and len(f._cigar) > 0
    assert np.array_equal(device_words(again[0], len(f._cigar)), f._cigar)
    # pageable pool (no device): no copy, the submission uploads
    g = bamio.AlignmentFile(bams[0])
    g.load(None)
    assert not g.cigar_pinned and g.device_pool() is None


def collect_of(svx_ctx, files, part_dev):
    """svx_collect_batch over the pools of `files` (every record, no segments)."""
    parts = [f._cigar for f in files]
    base = np.cumsum([0] + [len(p) for p in parts])
    aln_off = np.concatenate([f._cig_off[:-1] + b for f, b in zip(files, base)] + [[base[-1]]]).astype(np.uint64)
    ref_start = np.concatenate([f._cols["pos"] for f in files]).astype(np.int32)
    z32, z64 = np.zeros(0, np.uint32), np.zeros(1, np.uint64)
    return svx_ctx.collect_batch(parts, aln_off, ref_start, 40, z32, z64, z32, np.zeros(0, np.int32), np.zeros(0, np.int32),
                                 np.zeros(0, np.uint8), np.zeros(0, np.int32), np.zeros(1, np.uint32),
                                 np.zeros(3, np.int32), seg_params(), part_dev=part_dev)


def seg_params():
    return SVIM_inter.seg_params(helpers.options())


@pytest.mark.parametrize("which", ["both", "first", "second", "no events"])
def test_submission_takes_parts_where_they_lie(svx_ctx, dataset, which):
    fa, bams = dataset
    files = [bamio.AlignmentFile(b, device=0) for b in bams[:2]]
    for f in files:
        f.load(None)
    exp = collect_of(svx_ctx, files, None)
    assert len(exp[0]["aln"]) > 50
    pools = [f.device_pool() for f in files]
    assert all(p is not None for p in pools)
    if which == "first":
        pools[1] = None
    elif which == "second":
        pools[0] = None
    elif which == "no events":
        pools = [(f.device_pool(wait=True)[0], None) for f in files]
    got = collect_of(svx_ctx, files, pools)
    for k in exp[0]:
        assert np.array_equal(got[0][k], exp[0][k]), k
    # the host pointer of a part that is in HBM is not read: hand the call a poisoned copy of it
    if which == "both":
        real = [f._c_cigar for f in files]
        try:
            for f in files:
                f._c_cigar = np.full(len(f._c_cigar), 0xFFFFFFFF, np.uint32)
            got = collect_of(svx_ctx, files, pools)
        finally:
            for f, r in zip(files, real):
                f._c_cigar = r
        for k in exp[0]:
            assert np.array_equal(got[0][k], exp[0][k]), k


def test_collect_of_the_product_goes_through_the_device_pool(svx_ctx, dataset, monkeypatch):
    """analyze_alignment_file_coordsorted hands the reader's device copy to the submission, and finds what the
    submission that uploads finds."""
    fa, bams = dataset
    seen = []
    real = type(svx_ctx).collect_batch

    def spy(self, *a, **kw):
        seen.append(kw.get("part_dev"))
        return real(self, *a, **kw)
    monkeypatch.setattr(type(svx_ctx), "collect_batch", spy)
    o = helpers.options()
    got = [helpers.candidate_tuple(c) for c in
           SVIM_COLLECT.analyze_alignment_file_coordsorted(bamio.AlignmentFile(bams[0], device=0), o)]
    assert seen and seen[0] and seen[0][0] is not None and seen[0][0][0]
    del seen[:]
    exp = [helpers.candidate_tuple(c) for c in
           SVIM_COLLECT.analyze_alignment_file_coordsorted(bamio.AlignmentFile(bams[0]), o)]
    assert seen and (seen[0] is None or seen[0][0] is None)
    assert got == exp and len(got) > 10


def test_switch_in_the_environment(dataset):
    fa, bams = dataset
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from svim_asm_amd import bamio\n"
            "f = bamio.AlignmentFile(%r, device=0); f.load(None)\n"
            "print('POOL', f.cigar_pinned, f.device_pool() is None)\n" % (ROOT, bams[0]))
    env = dict(os.environ, SVX_BAM_DEVICE_POOL="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "POOL True True" in out.stdout, out.stdout + out.stderr


def random_slices(f, rng, n):
    """n slices sorted by (record, first base), as COLLECT asks for them."""
    l_seq = f._cols["l_seq"]
    rec = np.sort(rng.integers(0, len(l_seq), n))
    a = (rng.random(n) * l_seq[rec]).astype(np.int64)
    b = np.minimum(a + rng.integers(1, 3000, n), l_seq[rec])
    order = np.lexsort((a, rec))
    return rec[order].astype(np.uint32), a[order], b[order]


@pytest.mark.parametrize("percent", [25, 60, 100])
def test_device_leg_of_the_sequence_slices(svx_ctx, dataset, percent):
    """svx_bam_set_device_inflate: the members under the first share of a call's slices are inflated and verified by
    the device while the reader's threads take the rest — the same bases as the host alone decodes."""
    fa, bams = dataset
    host = bamio.AlignmentFile(bams[0], device=0)
    host.device_inflate_percent = 0
    host.load(None)
    rec, a, b = random_slices(host, np.random.default_rng(percent), 6000)
    exp, exp_off = host.sequence_slices_raw(rec, a, b)
    assert host.device_members == 0 and len(exp) > 1_000_000
    dev = bamio.AlignmentFile(bams[0], device=0)
    dev.device_inflate_percent = percent
    dev.device_inflate_min_members = 0
    dev.load(None)
    import time
    for attempt in range(200):  # (the lanes come up beside the first load of the process: tens of milliseconds)
        got, got_off = dev.sequence_slices_raw(rec, a, b)
        assert np.array_equal(got_off, exp_off) and np.array_equal(got, exp)
        if dev.device_members:
            break
        time.sleep(0.05)
    assert dev.device_members > 0
    # a second call reuses the lane and the device buffer
    before = dev.device_members
    got, got_off = dev.sequence_slices_raw(rec[:4000], a[:4000], b[:4000])
    exp2, exp2_off = host.sequence_slices_raw(rec[:4000], a[:4000], b[:4000])
    assert np.array_equal(got, exp2) and dev.device_members > before


def test_device_leg_flags_a_damaged_member(svx_ctx, dataset, tmp_path):
    """A member whose payload is damaged (CRC32 no longer fits) under a slice of the device's share: the call fails as it
    does on the host."""
    import shutil
    fa, bams = dataset
    bad = str(tmp_path / "bad.bam")
    shutil.copy(bams[0], bad)
    shutil.copy(bams[0] + ".bai", bad + ".bai")
    probe = bamio.AlignmentFile(bad, device=0)
    probe.load(None)
    rec, a, b = random_slices(probe, np.random.default_rng(5), 6000)
    # flip one byte inside a SEQ member's payload that no record walk reads (the load still succeeds)
    clean = open(bad, "rb").read()
    spans = [sp for sp in bamio._bgzf_block_spans(clean) if sp[2] and sp[1] >= 8192]
    for st, ln, *_ in spans[len(spans) // 3:]:
        raw = bytearray(clean)
        raw[st + ln // 2] ^= 0x55
        open(bad, "wb").write(bytes(raw))
        try:
            bamio.AlignmentFile(bad, device=0).load(None)
            break
        except ValueError:
            continue
    else:
        pytest.skip("every SEQ member of this file holds a record head")
    import time
    results = []
    for percent in (0, 100):
        f = bamio.AlignmentFile(bad, device=0)
        f.device_inflate_percent = percent
        f.device_inflate_min_members = 0
        f.load(None)
        time.sleep(0.5)  # (the lanes of a first load in this process)
        try:
            f.sequence_slices_raw(rec, a, b)
            results.append("ok")
        except ValueError:
            results.append("error")
    assert results == ["error", "error"]


def test_more_readers_than_lanes_wait_for_one(svx_ctx, dataset):
    """Four readers decode their sequence slices at once on a device with two inflate lanes: a call that finds both taken
    sleeps for one (svx_bam_set_device_inflate_wait) — or, without the wait, gives its whole call to the threads —; every
    reader returns the bases the host alone decodes, and readers opened and closed in a row reuse the device and
    page-locked buffers the earlier ones gave back (svx_bam.cpp: buffers kept between handles)."""
    import threading
    import time
    fa, bams = dataset
    host = bamio.AlignmentFile(bams[0], device=0)
    host.device_inflate_percent = 0
    host.load(None)
    rec, a, b = random_slices(host, np.random.default_rng(11), 6000)
    exp, exp_off = host.sequence_slices_raw(rec, a, b)
    for wait_ms in (400, 0):
        readers = []
        for _ in range(4):
            f = bamio.AlignmentFile(bams[0], device=0, threads=4)
            f.device_inflate_percent = 100
            f.device_inflate_min_members = 0
            f.device_inflate_wait_ms = wait_ms
            f.load(None)
            readers.append(f)
        time.sleep(0.3)  # (the lanes of a first load in this process)
        out, errors = [None] * 4, []

        def run(k):
            try:
                out[k] = readers[k].sequence_slices_raw(rec, a, b)
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=run, args=(k,)) for k in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors
        for got, got_off in out:
            assert np.array_equal(got_off, exp_off) and np.array_equal(got, exp)
        on_device = [f.device_members for f in readers]
        if wait_ms:
            assert all(n > 0 for n in on_device), on_device  # everybody got a lane in the end
        else:
            assert sum(1 for n in on_device if n > 0) >= 1     # two lanes: at least somebody; the others decoded on their threads
        for f in readers:
            f.close()


def test_the_walks_check_rides_on_the_device_leg(svx_ctx, dataset, tmp_path):
    """bamio.AlignmentFile.defer_verify (svx_bam_set_defer_verify): the record walks take only the bytes they need and leave
    the members they touched pending; the next sequence-slice call's device leg checks them beside its own members (the same
    records, the same bases, nothing pending afterwards); a member damaged BEHIND the bytes a walk needs lets the load pass
    and fails the sequence call — or verify_pending() — instead."""
    import shutil
    import time
    fa, bams = dataset
    plain = bamio.AlignmentFile(bams[0], device=0)
    plain.device_inflate_percent = 0
    plain.load(None)
    rec, a, b = random_slices(plain, np.random.default_rng(9), 6000)
    exp, exp_off = plain.sequence_slices_raw(rec, a, b)
    for attempt in range(200):
        f = bamio.AlignmentFile(bams[0], device=0)
        f.device_inflate_percent = 100
        f.device_inflate_min_members = 0
        f.defer_verify = True
        f.load(None)
        assert f.pending_members > 0
        for name in ("tid", "pos", "l_seq", "flag"):
            assert np.array_equal(f._cols[name], plain._cols[name])
        assert np.array_equal(f._cigar, plain._cigar)
        got, got_off = f.sequence_slices_raw(rec, a, b)
        assert np.array_equal(got_off, exp_off) and np.array_equal(got, exp)
        assert f.pending_members == 0
        if f.device_members:
            break
        time.sleep(0.05)  # (the lanes come up beside the first load of the process; until then the threads do the check)
    assert f.device_members > 0
    # without a sequence call: verify_pending() on the threads
    g = bamio.AlignmentFile(bams[0], device=0)
    g.device_inflate_percent = 100
    g.defer_verify = True
    g.load(None)
    assert g.pending_members > 0
    g.verify_pending()
    assert g.pending_members == 0
    # ---- a member damaged behind the bytes the walks need: the checking load notices, the deferring load does not — the
    # sequence call (device leg) or verify_pending() (threads) does
    bad = str(tmp_path / "bad.bam")
    shutil.copy(bams[0] + ".bai", bad + ".bai")
    clean = open(bams[0], "rb").read()
    import zlib

    def opened(defer):
        r = bamio.AlignmentFile(bad, device=0)
        r.device_inflate_percent = 100 if defer else 0
        r.device_inflate_min_members = 0
        r.defer_verify = defer
        return r
    found = False
    for st, ln, isz, *_ in bamio._bgzf_block_spans(clean):
        if not isz or ln < 2000 or found:
            continue
        good = zlib.decompress(clean[st:st + ln], -15)
        for back in range(3, 400):
            trial = bytearray(clean[st:st + ln])
            trial[ln - back] ^= 4
            try:
                out = zlib.decompress(bytes(trial), -15)
            except zlib.error:
                continue
            if len(out) != len(good) or out == good or out[:len(good) * 3 // 4] != good[:len(good) * 3 // 4]:
                continue
            raw = bytearray(clean)
            raw[st + ln - back] ^= 4
            open(bad, "wb").write(bytes(raw))
            try:
                opened(False).load(None)
                break  # no walk touches this member: the next one
            except ValueError:
                pass
            try:
                opened(True).load(None)
            except ValueError:
                break  # the damage lies in bytes a walk needs
            found = True
            break
    if not found:
        pytest.skip("no damage found that only a whole-member check notices")
    h = opened(True)
    h.load(None)
    assert h.pending_members > 0
    with pytest.raises(ValueError):
        h.sequence_slices_raw(rec, a, b)
    k = opened(True)
    k.load(None)
    with pytest.raises(ValueError):
        k.verify_pending()
