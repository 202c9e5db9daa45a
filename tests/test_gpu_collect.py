"""svx_collect_batch — COLLECT of a whole sample as one submission — and the asynchronous *_dev entry points it is
made of (svx_segments_rows_dev, svx_segments_postpass_dev) plus svx_linkage_cut_batch_dev and
svx_haplotype_distance_batch_dev, through the C-ABI.  Checked against the composition of the single-purpose
host-pointer entry points (each of which has its own oracle test) and against the oracle directly.

Reference seams: analyze_alignment_file_coordsorted (SVIM_COLLECT.py:61-83), analyze_read_segments
(SVIM_inter.py:62-340), fcluster(linkage(...)) (SVIM_COMBINE.py:134-135), compute_distance (:35-102)."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc
from svim_asm_amd import _lib
from tests import helpers

pytestmark = pytest.mark.gpu
PARAMS = (40, 100000, 50, 50, 50, 50)


@pytest.fixture(autouse=True, params=["fused_chain", "split_chain"])
def chain_form(request, svx_ctx):
    """Every test of this module runs with the split-segment chain of a submission as ONE kernel (inside the tile
    launch of the small-batch path, behind the streaming path) and as the three single-purpose launches
    (svx_ctx_set_split_chain)."""
    svx_ctx.set_split_chain(request.param == "split_chain")
    yield request.param
    svx_ctx.set_split_chain(False)


def long_cigar(rng, n, lead=None):
    """n packed ops of every code (clips in the middle too) behind `lead` leading clips (default: 0-3, sometimes many)."""
    if lead is None:
        lead = int(rng.integers(0, 4)) if rng.random() < 0.9 else int(rng.integers(0, n + 1))
    lead = min(lead, n)
    ops = rng.choice(np.array([0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15], np.uint32), size=n)
    ops[:lead] = rng.choice(np.array([4, 5], np.uint32), size=lead)
    return ((rng.integers(1, 2000, size=n).astype(np.uint32) << 4) | ops).astype(np.uint32)


def random_batch(rng, n_aln, n_parts=2, n_reads=40, max_supp=4, long_read=False, long_aln=0.0, long_max=6000):
    """Records with random CIGARs split over `n_parts` pools, and chimeric reads whose segments name pool records
    (primaries) and extra alignments (SA-derived).  `long_aln`: fraction of the records whose CIGAR has 9 .. `long_max`
    ops (log-uniform) instead of 1-60."""
    tuples = [helpers.random_cigar(rng, int(rng.integers(1, 60)), hard=bool(rng.random() < 0.15)) for _ in range(n_aln)]
    words = [np.array([(l << 4) | o for o, l in t], dtype=np.uint32) for t in tuples]
    if long_aln > 0:
        for a in np.nonzero(rng.random(n_aln) < long_aln)[0].tolist():
            words[a] = long_cigar(rng, int(np.exp(rng.uniform(np.log(9), np.log(long_max)))))
    cut = sorted(rng.integers(0, n_aln + 1, size=n_parts - 1).tolist())
    bounds = [0] + cut + [n_aln]
    parts = [np.concatenate(words[a:b]) if b > a else np.zeros(0, np.uint32) for a, b in zip(bounds, bounds[1:])]
    aln_off = np.concatenate(([0], np.cumsum([len(w) for w in words]))).astype(np.uint64)
    ref_start = rng.integers(0, 1 << 28, size=n_aln).astype(np.int32)
    extra, seg_src, seg_tid, seg_pos, seg_rev, seg_qend, counts = [], [], [], [], [], [], []
    for r in range(n_reads):
        k = int(rng.integers(1, max_supp + 1)) if not (long_read and r == 0) else 23  # > 8 segments: the serial path
        prim = int(rng.integers(0, n_aln))
        seg_src.append(prim); seg_tid.append(int(rng.integers(0, 4))); seg_pos.append(int(ref_start[prim]))
        seg_rev.append(int(rng.random() < 0.3))
        seg_qend.append(int(rng.integers(0, 5000)) if rng.random() < 0.5 else -1)
        for _ in range(k):
            t = helpers.random_cigar(rng, int(rng.integers(1, 12)))
            seg_src.append(n_aln + len(extra))
            extra.append(np.array([(l << 4) | o for o, l in t], dtype=np.uint32))
            seg_tid.append(int(rng.integers(0, 4))); seg_pos.append(int(rng.integers(0, 1 << 27)))
            seg_rev.append(int(rng.random() < 0.3)); seg_qend.append(-1)
        counts.append(1 + k)
    extra_off = np.concatenate(([0], np.cumsum([len(w) for w in extra]))).astype(np.uint64) if extra else np.zeros(1, np.uint64)
    extra_cigar = np.concatenate(extra) if extra else np.zeros(0, np.uint32)
    read_off = np.concatenate(([0], np.cumsum(counts))).astype(np.uint32)
    rank = np.array([2, 0, 3, 1], dtype=np.int32)
    return dict(parts=parts, aln_off=aln_off, ref_start=ref_start, extra_cigar=extra_cigar, extra_off=extra_off,
                seg_src=np.array(seg_src, np.uint32), seg_tid=np.array(seg_tid, np.int32), seg_pos=np.array(seg_pos, np.int32),
                seg_rev=np.array(seg_rev, np.uint8), seg_qend=np.array(seg_qend, np.int32), read_off=read_off, rank=rank)


def call(fn, b, min_len=40, params=PARAMS):
    return fn(b["parts"], b["aln_off"], b["ref_start"], min_len, b["extra_cigar"], b["extra_off"], b["seg_src"],
              b["seg_tid"], b["seg_pos"], b["seg_rev"], b["seg_qend"], b["read_off"], b["rank"], params)


def same(a, b):
    (sa, ra, pa, fa), (sb, rb, pb, fb) = a, b
    assert sorted(sa) == sorted(sb)
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(ra, rb) and np.array_equal(pa, pb) and np.array_equal(fa, fb)


def same_as_oracle(got, exp):
    """One submission against the ORACLE alone — signatures, the adjacency records of every segment slot (rows from
    the oracle's CIGAR statistics, the oracle's decision tree) and the derived records of every read (the record-level
    post-passes of oracle/svim_oracle.py) — not against another composition of this library's kernels."""
    (sig, raw, post, first), (o_sig, o_raw, o_post) = got, exp
    for k in o_sig:
        assert np.array_equal(sig[k], o_sig[k]), k
    assert np.array_equal(raw.view(np.int32).reshape(-1, 8), o_raw.view(np.int32).reshape(-1, 8))
    assert helpers.post_records_as_tuples(post, first) == o_post


@pytest.mark.parametrize("seed", range(6))
def test_one_submission_equals_the_single_purpose_calls(svx_ctx, seed):
    rng = np.random.default_rng(100 + seed)
    b = random_batch(rng, n_aln=int(rng.choice([1, 7, 300, 3000])), n_parts=int(rng.choice([1, 2, 3])),
                     n_reads=int(rng.choice([0, 1, 60, 500])), long_read=seed % 2 == 1)
    for min_len, params in ((40, PARAMS), (1, (1, 100000, 500, 500, 500, 500)), (30, (50, 3000, 0, 0, 0, 0))):
        got = call(svx_ctx.collect_batch, b, min_len, params)
        exp = call(svx_ctx.collect_batch_composed, b, min_len, params)
        same(got, exp)
        same_as_oracle(got, call(helpers.oracle_collect, b, min_len, params))


@pytest.mark.parametrize("seed", range(4))
def test_submission_with_long_primaries(svx_ctx, seed):
    """Chimeric reads whose primaries run from a handful to thousands of ops (the chunk list of the fused chain, the
    whole-workgroup walk beyond 2048 ops), both CIGAR paths."""
    rng = np.random.default_rng(300 + seed)
    b = random_batch(rng, n_aln=int(rng.choice([40, 400])), n_parts=2, n_reads=int(rng.choice([30, 400])), max_supp=3,
                     long_read=seed % 2 == 1, long_aln=0.5, long_max=int(rng.choice([600, 9000])))
    svx_ctx.set_small_batch_ops(0 if seed >= 2 else 1 << 23)
    try:
        got = call(svx_ctx.collect_batch, b)
        same(got, call(svx_ctx.collect_batch_composed, b))
        same_as_oracle(got, call(helpers.oracle_collect, b))
    finally:
        svx_ctx.set_small_batch_ops(1 << 23)


def tiling_batch(rng, n_reads, n_contigs=3):
    """Chimeric reads whose segments tile the read (gaps / overlaps of a few bases around the tolerances) and land
    close to each other on few contigs, both strands: most adjacent pairs are a tandem duplication, an inversion, a
    deletion / insertion or a breakend, so the three post-passes (SVIM_inter.py:260-338) have work — merged tandem runs,
    mirrored breakend pairs, overlapping inversion groups.  A segment's CIGAR is S M S in the record's orientation (the
    primary is a pool record with an indel inside, the others SA-derived)."""
    words, ref_start, extra, seg_src, seg_tid, seg_pos, seg_rev, counts = [], [], [], [], [], [], [], []
    for r in range(n_reads):
        k = int(rng.integers(2, 7))
        style = int(rng.integers(0, 4))  # 0 tandem-like, 1 inversion-like, 2 mirrored breakends, 3 anything
        if style == 0:  # copies of one unit, one after the other on the read, all at (nearly) one reference position
            cuts = int(rng.integers(0, 500)) + np.arange(k + 1) * int(rng.integers(300, 3000))
        else:
            cuts = np.sort(rng.integers(200, 20000, size=k + 1))
        total = int(cuts[-1]) + int(rng.integers(0, 300))
        tid0, base = int(rng.integers(0, n_contigs)), int(rng.integers(10_000, 1_000_000))
        px = base
        for i in range(k):
            jitter = 8 if style == 0 else 60
            qs = int(cuts[i]) + int(rng.integers(-jitter, jitter + 1)) if i else int(cuts[0])
            qe = int(cuts[i + 1])
            qs = min(max(qs, 0), qe - 1)
            rev = {0: 0, 1: i % 2, 2: 0, 3: int(rng.random() < 0.4)}[style]
            tid = tid0 if style != 2 else (tid0 + (i % 2)) % n_contigs
            if style == 0:
                pos = base + int(rng.integers(-6, 7)) + (0 if rng.random() < 0.85 else 5000)
            elif style == 2:  # A (here) B (elsewhere) C (where A ended): the two breakends of an interspersed duplication
                pos = px + int(rng.integers(-8, 9)) if i % 2 == 0 else int(rng.integers(10_000, 1_000_000))
                px += (qe - qs) if i % 2 == 0 else 0
            else:
                pos = base + int(rng.integers(-3000, 3000))
            m = qe - qs
            lead, tail = (total - qe, qs) if rev else (qs, total - qe)
            cig = [(lead << 4) | 4] if lead else []
            if i == 0 and m > 100:  # the primary: a pool record with an insertion inside
                cig += [((m - 60) << 4) | 0, (50 << 4) | 1, (10 << 4) | 0]
            else:
                cig += [(m << 4) | 0]
            cig += [(tail << 4) | 4] if tail else []
            w = np.array(cig, np.uint32)
            if i == 0:
                seg_src.append(len(words)); words.append(w); ref_start.append(pos)
            else:
                seg_src.append(-1 - len(extra)); extra.append(w)
            seg_tid.append(tid); seg_pos.append(pos); seg_rev.append(rev)
        counts.append(k)
    # unrelated records between the primaries
    order = rng.permutation(len(words) + n_reads // 2)
    pool, pool_rs, where = [], [], {}
    for slot in order.tolist():
        if slot < len(words):
            where[slot] = len(pool); pool.append(words[slot]); pool_rs.append(ref_start[slot])
        else:
            t = helpers.random_cigar(rng, int(rng.integers(1, 40)))
            pool.append(np.array([(l << 4) | o for o, l in t], np.uint32)); pool_rs.append(int(rng.integers(0, 1 << 27)))
    n_aln = len(pool)
    src = np.array([where[x] if x >= 0 else n_aln + (-1 - x) for x in seg_src], np.uint32)
    aln_off = np.concatenate(([0], np.cumsum([len(w) for w in pool]))).astype(np.uint64)
    extra_off = np.concatenate(([0], np.cumsum([len(w) for w in extra]))).astype(np.uint64)
    rank = np.array(sorted(range(n_contigs), key=lambda i: "chr%d" % (i * 7 + 1)), np.int32)
    return dict(parts=[np.concatenate(pool)], aln_off=aln_off, ref_start=np.array(pool_rs, np.int32),
                extra_cigar=np.concatenate(extra), extra_off=extra_off, seg_src=src, seg_tid=np.array(seg_tid, np.int32),
                seg_pos=np.array(seg_pos, np.int32), seg_rev=np.array(seg_rev, np.uint8),
                seg_qend=np.full(len(src), -1, np.int32), read_off=np.concatenate(([0], np.cumsum(counts))).astype(np.uint32),
                rank=np.argsort(rank).astype(np.int32))


@pytest.mark.parametrize("seed,streaming", [(0, False), (1, False), (2, True)])
def test_tiling_reads_against_the_oracle(svx_ctx, seed, streaming):
    """The whole chain of one submission — rows, decision tree AND post-passes with plenty of derived records —
    against the oracle alone (no second kernel composition involved)."""
    rng = np.random.default_rng(900 + seed)
    b = tiling_batch(rng, n_reads=[300, 50, 900][seed])
    svx_ctx.set_small_batch_ops(0 if streaming else 1 << 23)
    try:
        for params in (PARAMS, (40, 2000, 50, 50, 50, 50)):
            got = call(svx_ctx.collect_batch, b, 40, params)
            o = call(helpers.oracle_collect, b, 40, params)
            same_as_oracle(got, o)
            kinds = {t[0] for recs in o[2] for t in recs}
            assert {"TANDEM", "INV", "DUP_INT"} <= kinds, kinds
            n_tandem_raw = int((o[1]["kind"] == 4).sum())
            assert sum(t[0] == "TANDEM" for recs in o[2] for t in recs) < n_tandem_raw  # runs were merged
    finally:
        svx_ctx.set_small_batch_ops(1 << 23)


def test_streaming_path_and_capacity_retry(svx_ctx):
    """A batch above the small-batch limit (five-launch streaming path inside the submission) and a tiny first
    capacity guess (dense signatures: the binding retries with the exact count)."""
    rng = np.random.default_rng(7)
    b = random_batch(rng, n_aln=4000, n_parts=2, n_reads=100)
    svx_ctx.set_small_batch_ops(0)
    try:
        got = call(svx_ctx.collect_batch, b, 1)
        same(got, call(svx_ctx.collect_batch_composed, b, 1))
        same_as_oracle(got, call(helpers.oracle_collect, b, 1))
    finally:
        svx_ctx.set_small_batch_ops(1 << 23)


def test_empty_and_degenerate_batches(svx_ctx):
    z32, z64 = np.zeros(0, np.uint32), np.zeros(1, np.uint64)
    sig, raw, post, first = svx_ctx.collect_batch([z32], z64, np.zeros(0, np.int32), 40, z32, z64, z32, np.zeros(0, np.int32),
                                                  np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros(0, np.int32),
                                                  np.zeros(1, np.uint32), np.zeros(0, np.int32), PARAMS)
    assert len(sig["aln"]) == 0 and len(raw) == 0 and len(post) == 0 and first.tolist() == [0]
    # alignments without ops, no chimeric reads
    off = np.array([0, 0, 3, 3], dtype=np.uint64)
    words = np.array([(100 << 4) | 0, (50 << 4) | 2, (10 << 4) | 0], dtype=np.uint32)
    sig, raw, post, first = svx_ctx.collect_batch([words], off, np.array([5, 7, 9], np.int32), 40, z32, z64, z32,
                                                  np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.uint8),
                                                  np.zeros(0, np.int32), np.zeros(1, np.uint32), np.zeros(0, np.int32), PARAMS)
    assert sig["aln"].tolist() == [1] and sig["ref_pos"].tolist() == [107] and sig["len"].tolist() == [50]


def test_bad_arguments_are_refused(svx_ctx):
    rng = np.random.default_rng(3)
    b = random_batch(rng, n_aln=20, n_parts=1, n_reads=3)
    bad = dict(b)
    bad["seg_src"] = b["seg_src"].copy()
    bad["seg_src"][0] = 10_000  # names no alignment
    with pytest.raises(_lib.SvxError):
        call(svx_ctx.collect_batch, bad)
    bad = dict(b)
    bad["aln_off"] = b["aln_off"].copy()
    bad["aln_off"][-1] += 1  # does not end at the pools' op count
    with pytest.raises(_lib.SvxError):
        call(svx_ctx.collect_batch, bad)
    same(call(svx_ctx.collect_batch, b), call(svx_ctx.collect_batch_composed, b))  # the context is still usable


def test_linkage_dev_matches_host_entry(svx_ctx):
    rng = np.random.default_rng(11)
    sizes = rng.integers(1, 13, size=400).astype(np.uint32)
    sizes[5] = 150  # beyond the LDS slice: HBM scratch
    dist = np.concatenate([np.round(rng.random(int(n) * (int(n) - 1) // 2) * 6) / 2 for n in sizes])
    exp = svx_ctx.linkage_cut_batch(dist, sizes, 1.5)
    d_dist, d_n = svx_ctx.dev_array(dist), svx_ctx.dev_array(sizes)
    d_lab = svx_ctx.dev_array(nbytes=4 * int(sizes.sum()))
    svx_ctx._check(svx_ctx.lib.svx_linkage_cut_batch_dev(svx_ctx.h, d_dist.ptr, sizes.ctypes.data, d_n.ptr, len(sizes),
                                                         1.5, d_lab.ptr))
    assert np.array_equal(d_lab.download(np.uint32), exp)


def test_haplotype_distance_dev_matches_host_entry(svx_ctx):
    rng = np.random.default_rng(12)
    pool = np.frombuffer(b"ACGTacgtN", np.uint8)[rng.integers(0, 9, size=40000)]
    n = 300
    pieces = np.zeros(n * 6, dtype=_lib.HAP_PIECE_DTYPE)
    pieces["off"] = rng.integers(0, 30000, size=n * 6)
    pieces["len"] = rng.integers(0, 600, size=n * 6)
    pieces["repeat"] = rng.integers(0, 3, size=n * 6)
    pieces["flags"] = rng.integers(0, 4, size=n * 6)
    for k_max in (0xFFFFFFFF, 50):
        exp = svx_ctx.haplotype_distance_batch(pool, pieces, k_max)
        d_pool = svx_ctx.dev_array(pool)
        got = np.zeros(n, np.uint32)
        svx_ctx._check(svx_ctx.lib.svx_haplotype_distance_batch_dev(svx_ctx.h, d_pool.ptr, len(pool), pieces.ctypes.data, n,
                                                                    k_max, got.ctypes.data))
        assert np.array_equal(got, exp)


def test_postpass_dev_and_rows_dev_raw_calls(svx_ctx):
    """The asynchronous entry points called by hand on svx_dev_malloc'ed buffers, one stream, one synchronisation."""
    rng = np.random.default_rng(13)
    b = random_batch(rng, n_aln=200, n_parts=1, n_reads=80, long_read=True)
    cigar = np.concatenate((b["parts"][0], b["extra_cigar"]))
    n_aln = len(b["aln_off"]) - 1
    off = np.concatenate((b["aln_off"], int(b["aln_off"][-1]) + b["extra_off"][1:])).astype(np.uint64)
    n_segs, n_reads = len(b["seg_src"]), len(b["read_off"]) - 1
    ctx = svx_ctx
    d = {k: ctx.dev_array(v) for k, v in dict(cigar=cigar, off=off, src=b["seg_src"], tid=b["seg_tid"], pos=b["seg_pos"],
                                               rev=b["seg_rev"], qend=b["seg_qend"], roff=b["read_off"], rank=b["rank"]).items()}
    d_segs, d_rl = ctx.dev_array(nbytes=24 * n_segs), ctx.dev_array(nbytes=4 * n_reads)
    d_raw = ctx.dev_array(nbytes=32 * n_segs)
    slots = np.diff(b["read_off"].astype(np.int64))
    post_off = np.concatenate(([0], np.cumsum(slots * (slots + 3) // 2))).astype(np.uint64)
    d_poff, d_post, d_cnt = ctx.dev_array(post_off), ctx.dev_array(nbytes=32 * int(post_off[-1])), ctx.dev_array(nbytes=4 * n_reads)
    prm = _lib.SegParams(*PARAMS)
    ctx._check(ctx.lib.svx_segments_rows_dev(ctx.h, d["cigar"].ptr, d["off"].ptr, d["src"].ptr, d["tid"].ptr, d["pos"].ptr,
                                             d["rev"].ptr, d["qend"].ptr, n_segs, d["roff"].ptr, n_reads, d_segs.ptr, d_rl.ptr))
    ctx._check(ctx.lib.svx_segments_classify_dev(ctx.h, d_segs.ptr, n_segs, d["roff"].ptr, n_reads, d_rl.ptr, C.byref(prm), d_raw.ptr))
    ctx._check(ctx.lib.svx_segments_postpass_dev(ctx.h, d_raw.ptr, b["read_off"].ctypes.data, d["roff"].ptr, n_reads, d["rank"].ptr,
                                                 len(b["rank"]), C.byref(prm), d_post.ptr, post_off.ctypes.data, d_poff.ptr, d_cnt.ptr))
    ctx.sync()
    # expectations from the host-side composition
    st_a = ctx.cigar_stats(cigar, off)
    st = {k: st_a[k].astype(np.int64)[b["seg_src"].astype(np.int64)] for k in ("ref_len", "q_start", "q_end", "read_len")}
    segs, read_len = _lib.segment_rows(st, b["seg_tid"], b["seg_pos"], b["seg_rev"], b["seg_qend"], b["read_off"])
    assert np.array_equal(d_segs.download(np.int32).reshape(-1, 6), segs.view(np.int32).reshape(-1, 6))
    assert np.array_equal(d_rl.download(np.int32), read_len)
    raw = ctx.segments_classify(segs, b["read_off"], read_len, PARAMS)
    assert np.array_equal(d_raw.download(np.int32).reshape(-1, 8), raw.view(np.int32).reshape(-1, 8))
    post, first = ctx.segments_postpass(raw, b["read_off"], b["rank"], PARAMS)
    cnt = d_cnt.download(np.uint32)
    assert np.array_equal(np.concatenate(([0], np.cumsum(cnt))), first)
    got = d_post.download(np.int32).reshape(-1, 8)
    take = np.repeat(post_off[:-1].astype(np.int64) - first[:-1], cnt) + np.arange(int(first[-1]))
    assert np.array_equal(got[take], post.view(np.int32).reshape(-1, 8))


def test_collect_batch_dev_on_resident_buffers(svx_ctx):
    """svx_collect_batch_dev by hand: everything resident, asynchronous, one synchronisation; repeated calls reuse
    the context's second stream and its events."""
    rng = np.random.default_rng(21)
    b = random_batch(rng, n_aln=500, n_parts=1, n_reads=120, long_read=True)
    exp_sig, exp_raw, exp_post, exp_first = call(svx_ctx.collect_batch_composed, b)
    ctx = svx_ctx
    n_aln, n_ops = len(b["aln_off"]) - 1, int(b["aln_off"][-1])
    cigar = np.concatenate((b["parts"][0], b["extra_cigar"]))
    off = np.concatenate((b["aln_off"], n_ops + b["extra_off"][1:])).astype(np.uint64)
    n_segs, n_reads = len(b["seg_src"]), len(b["read_off"]) - 1
    slots = np.diff(b["read_off"].astype(np.int64))
    post_off = np.concatenate(([0], np.cumsum(slots * (slots + 3) // 2))).astype(np.uint64)
    d = {k: ctx.dev_array(v) for k, v in dict(cigar=cigar, off=off, rs=b["ref_start"], src=b["seg_src"], tid=b["seg_tid"],
                                               pos=b["seg_pos"], rev=b["seg_rev"], qend=b["seg_qend"], roff=b["read_off"],
                                               rank=b["rank"], poff=post_off).items()}
    cap = len(exp_sig["aln"]) + 5
    o = [ctx.dev_array(nbytes=4 * cap) for _ in range(4)] + [ctx.dev_array(nbytes=cap), ctx.dev_array(np.zeros(1, np.uint64))]
    d_segs, d_rl = ctx.dev_array(nbytes=24 * n_segs), ctx.dev_array(nbytes=4 * n_reads)
    d_raw, d_post, d_cnt = ctx.dev_array(nbytes=32 * n_segs), ctx.dev_array(nbytes=32 * int(post_off[-1])), ctx.dev_array(nbytes=4 * n_reads)
    dv = _lib.CollectDev(d_cigar=d["cigar"].ptr, n_ops=n_ops, d_aln_off=d["off"].ptr, n_aln=n_aln, n_extra=len(b["extra_off"]) - 1,
                         d_ref_start=d["rs"].ptr, min_len=40, d_seg_src=d["src"].ptr, d_seg_tid=d["tid"].ptr, d_seg_pos=d["pos"].ptr,
                         d_seg_rev=d["rev"].ptr, d_seg_qend=d["qend"].ptr, n_segs=n_segs, read_off=b["read_off"].ctypes.data,
                         d_read_off=d["roff"].ptr, n_reads=n_reads, d_contig_rank=d["rank"].ptr, n_contigs=len(b["rank"]),
                         params=_lib.SegParams(*PARAMS), d_sig=_lib.SigSoa(*[x.ptr for x in o[:5]]), sig_cap=cap, d_n_sig=o[5].ptr,
                         d_segs=d_segs.ptr, d_read_len=d_rl.ptr, d_raw=d_raw.ptr, d_post=d_post.ptr, post_off=post_off.ctypes.data,
                         d_post_off=d["poff"].ptr, d_post_cnt=d_cnt.ptr)
    for _ in range(3):
        ctx._check(ctx.lib.svx_collect_batch_dev(ctx.h, C.byref(dv)))
    ctx.sync()
    n = int(o[5].download(np.uint64)[0])
    assert n == len(exp_sig["aln"])
    for x, key in zip(o[:5], ("aln", "ref_pos", "read_pos", "len", "type")):
        assert np.array_equal(x.download(np.uint32 if key != "type" else np.uint8, n), exp_sig[key]), key
    assert np.array_equal(d_raw.download(np.int32).reshape(-1, 8), exp_raw.view(np.int32).reshape(-1, 8))
    cnt = d_cnt.download(np.uint32)
    assert np.array_equal(np.concatenate(([0], np.cumsum(cnt))), exp_first)
    take = np.repeat(post_off[:-1].astype(np.int64) - exp_first[:-1], cnt) + np.arange(int(exp_first[-1]))
    assert np.array_equal(d_post.download(np.int32).reshape(-1, 8)[take], exp_post.view(np.int32).reshape(-1, 8))


def _sized_cigar(rng, n, lead):
    """n ops: `lead` leading clips (S and H mixed), then random ops of every code (clips in the middle too)."""
    ops = [(int(rng.choice([4, 5])), int(rng.integers(1, 50))) for _ in range(min(lead, n))]
    while len(ops) < n:
        ops.append((int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15])), int(rng.integers(1, 2000))))
    return np.array([(l << 4) | o for o, l in ops], dtype=np.uint32)


@pytest.mark.parametrize("deal", ["equal_counts", "table"])
@pytest.mark.parametrize("streaming", [False, True])
def test_chain_rows_across_alignment_sizes(svx_ctx, streaming, deal):
    """The rows of the fused chain against the oracle's CIGAR statistics for alignments around every size boundary of
    the kernel (a lane up to 8 ops, chunks of 128 ops shared by the sixteen groups of a workgroup beyond), with clip
    prefixes that end inside the first word, run through the whole first chunk (128, 129, 300 clips) or are the whole
    alignment, lengths of 2^24 and more (no 24-bit multiply), one read with several hundred segments (more than one
    batch of 256 per workgroup); reads dealt to the workgroups in equal counts and by a caller's table
    (d_chain_deal: random cuts, empty workgroups among them)."""
    rng = np.random.default_rng(77)
    sizes = [1, 2, 8, 9, 15, 16, 17, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 2047, 2048, 2049, 4095, 4096, 4097,
             8192, 8193, 20000, 70000]
    words = []
    for n in sizes:
        for lead in (0, 1, 2, 3):
            words.append(_sized_cigar(rng, n, lead))
    for lead, n in ((127, 400), (128, 400), (129, 400), (300, 1000), (200, 200), (128, 128), (2048, 2048), (1500, 2049), (9, 9)):
        words.append(_sized_cigar(rng, n, lead))
    for n in (9, 130, 700):  # lengths beyond 24 bits (the sums wrap modulo 2^32 on either side)
        w = _sized_cigar(rng, n, 1)
        at = rng.integers(0, n, size=3)
        w[at] = (rng.integers(1 << 24, 1 << 28, size=3).astype(np.uint32) << 4) | (w[at] & 15)
        words.append(w)
    n_aln = len(words)
    aln_off = np.concatenate(([0], np.cumsum([len(w) for w in words]))).astype(np.uint64)
    n_ops = int(aln_off[-1])
    ref_start = rng.integers(0, 1 << 27, size=n_aln).astype(np.int32)
    # reads: every alignment once as a primary with one or two SA-derived segments; then one read whose 700 segments
    # name alignments of every size
    extra, seg_src, counts = [], [], []
    for a in range(n_aln):
        k = 1 + a % 2
        seg_src.append(a)
        for _ in range(k):
            seg_src.append(n_aln + len(extra))
            extra.append(_sized_cigar(rng, 3, 1))
        counts.append(1 + k)
    small = [a for a in range(n_aln) if len(words[a]) <= 5000]
    many = [int(rng.choice(small)) for _ in range(700)]
    seg_src += many
    counts.append(len(many))
    seg_src = np.array(seg_src, np.uint32)
    n_segs, n_reads = len(seg_src), len(counts)
    read_off = np.concatenate(([0], np.cumsum(counts))).astype(np.uint32)
    seg_tid = rng.integers(0, 4, size=n_segs).astype(np.int32)
    seg_pos = rng.integers(0, 1 << 27, size=n_segs).astype(np.int32)
    seg_rev = (rng.random(n_segs) < 0.3).astype(np.uint8)
    seg_qend = np.where(rng.random(n_segs) < 0.3, rng.integers(0, 5000, size=n_segs), -1).astype(np.int32)
    extra_off = np.concatenate(([0], np.cumsum([len(w) for w in extra]))).astype(np.uint64)
    cigar = np.concatenate(words + extra)
    off = np.concatenate((aln_off, n_ops + extra_off[1:])).astype(np.uint64)
    rank = np.array([2, 0, 3, 1], dtype=np.int32)
    slots = np.diff(read_off.astype(np.int64))
    post_off = np.concatenate(([0], np.cumsum(slots * (slots + 3) // 2))).astype(np.uint64)
    # expectation: the oracle's statistics -> rows (host formula of the binding)
    st_all = orc.cigar_stats(cigar, off)
    exp_segs, exp_rl = orc.segment_rows(st_all, seg_src, seg_tid, seg_pos, seg_rev, seg_qend, read_off)
    exp_sig = orc.cigar_extract(cigar[:n_ops], aln_off, ref_start, 40)
    ctx = svx_ctx
    ctx.set_small_batch_ops(0 if streaming else 1 << 23)
    try:
        for phase in (0,):
            d_all = ctx.dev_array(cigar)
            d = {k: ctx.dev_array(v) for k, v in dict(off=off, rs=ref_start, src=seg_src, tid=seg_tid, pos=seg_pos, rev=seg_rev,
                                                       qend=seg_qend, roff=read_off, rank=rank, poff=post_off).items()}
            cap = len(exp_sig["aln"]) + 5
            o = [ctx.dev_array(nbytes=4 * cap) for _ in range(4)] + [ctx.dev_array(nbytes=cap), ctx.dev_array(np.zeros(1, np.uint64))]
            d_segs, d_rl = ctx.dev_array(nbytes=24 * n_segs), ctx.dev_array(nbytes=4 * n_reads)
            d_raw, d_post, d_cnt = ctx.dev_array(nbytes=32 * n_segs), ctx.dev_array(nbytes=32 * int(post_off[-1])), ctx.dev_array(nbytes=4 * n_reads)
            dv = _lib.CollectDev(d_cigar=d_all.ptr + 4 * phase, n_ops=n_ops, d_aln_off=d["off"].ptr, n_aln=n_aln, n_extra=len(extra),
                                 d_ref_start=d["rs"].ptr, min_len=40, d_seg_src=d["src"].ptr, d_seg_tid=d["tid"].ptr,
                                 d_seg_pos=d["pos"].ptr, d_seg_rev=d["rev"].ptr, d_seg_qend=d["qend"].ptr, n_segs=n_segs,
                                 read_off=read_off.ctypes.data, d_read_off=d["roff"].ptr, n_reads=n_reads, d_contig_rank=d["rank"].ptr,
                                 n_contigs=len(rank), params=_lib.SegParams(*PARAMS), d_sig=_lib.SigSoa(*[x.ptr for x in o[:5]]),
                                 sig_cap=cap, d_n_sig=o[5].ptr, d_segs=d_segs.ptr, d_read_len=d_rl.ptr, d_raw=d_raw.ptr,
                                 d_post=d_post.ptr, post_off=post_off.ctypes.data, d_post_off=d["poff"].ptr, d_post_cnt=d_cnt.ptr)
            if deal == "table":  # consecutive reads per workgroup, cut anywhere; some workgroups get nothing
                cuts = np.sort(rng.integers(0, n_reads + 1, size=40))
                first = np.concatenate(([0], cuts, [n_reads, n_reads])).astype(np.uint32)
                table = np.stack((first, read_off[first]), axis=1).astype(np.uint32).ravel()
                d_deal = ctx.dev_array(table)
                dv.d_chain_deal, dv.n_chain_blocks = d_deal.ptr, len(first) - 1
            ctx._check(ctx.lib.svx_collect_batch_dev(ctx.h, C.byref(dv)))
            ctx.sync()
            got = d_segs.download(np.int32).reshape(-1, 6)
            bad = np.nonzero((got != exp_segs.view(np.int32).reshape(-1, 6)).any(axis=1))[0]
            assert len(bad) == 0, (phase, bad[:5], [int(off[seg_src[j] + 1] - off[seg_src[j]]) for j in bad[:5]])
            assert np.array_equal(d_rl.download(np.int32), exp_rl)
            n = int(o[5].download(np.uint64)[0])
            assert n == len(exp_sig["aln"])
            assert np.array_equal(o[1].download(np.uint32, n), exp_sig["ref_pos"])
            raw = orc.segments_classify(exp_segs, read_off, exp_rl, PARAMS)
            assert np.array_equal(d_raw.download(np.int32).reshape(-1, 8), raw.view(np.int32).reshape(-1, 8))
    finally:
        ctx.set_small_batch_ops(1 << 23)
