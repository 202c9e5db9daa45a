"""HIP bit-vector edit distance (svx_edit_distance_batch) vs a textbook DP (and, for long
sequences, the band-doubling DP pinned against it).

Reference call sites: edlib.align(h1, h2)["editDistance"], SVIM_COMBINE.py:50,64,76,88,100.
"""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["wavefront_1024", "wavefront_48", "bitvector_only"])
def edit_stage(request, svx_ctx):
    """Every test of this module runs with the default two-stage plan (wavefront pass up to 1024 edits, then
    the bit-vector kernel), with a cap so small that most pairs fall through to the second stage, and with
    the bit-vector kernel alone (svx_ctx_set_edit_wavefront_cap)."""
    svx_ctx.set_edit_wavefront_cap({"wavefront_1024": 1024, "wavefront_48": 48, "bitvector_only": 0}[request.param])
    yield request.param
    svx_ctx.set_edit_wavefront_cap(1024)


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


def _pool(pairs):
    pool = b"".join(a + b for a, b in pairs)
    ao, al, bo, bl, o = [], [], [], [], 0
    for a, b in pairs:
        ao.append(o); al.append(len(a)); o += len(a)
        bo.append(o); bl.append(len(b)); o += len(b)
    return np.frombuffer(pool, np.uint8) if pool else np.zeros(0, np.uint8), ao, al, bo, bl


def _rand_dna(rng, n):
    return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].tobytes()


def _edited(rng, s, n_sub, n_indel, max_indel):
    """s with substitutions and a few insertions/deletions of up to max_indel bases (numpy, fast)."""
    a = np.frombuffer(s, np.uint8).copy()
    if n_sub and len(a):
        a[rng.integers(0, len(a), n_sub)] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_sub)]
    parts, last = [], 0
    for cut in sorted(rng.integers(0, len(a) + 1, n_indel).tolist()):
        parts.append(a[last:cut])
        ln = int(rng.integers(1, max_indel + 1))
        if rng.random() < 0.5:
            parts.append(np.frombuffer(_rand_dna(rng, ln), np.uint8))
            last = cut
        else:
            last = min(len(a), cut + ln)
    parts.append(a[last:])
    return np.concatenate(parts).tobytes()


@pytest.mark.parametrize("length", [1000, 10000, 100000])
def test_long_sequences_every_mode(svx_ctx, length):
    """Lengths 10^3..10^5 (1..25 strips of 4096 rows), thresholds 0 / 200 / exact; pairs that are
    identical, within 200 edits, a few thousand edits apart (band widened per pair) and unrelated."""
    rng = np.random.default_rng(length)
    base = _rand_dna(rng, length)
    pairs = [(base, base),
             (base, _edited(rng, base, 40, 6, 20)),                       # within the default threshold
             (_edited(rng, base, 90, 8, 12), base),                       # longer / shorter swapped
             (base, _edited(rng, base, length // 50, 10, 300)),           # > 200, resolved by a wider band
             (base[:length - length // 7], base),                         # length difference alone decides
             (base, _rand_dna(rng, int(length * 0.9)))]                   # unrelated: the band grows to the full matrix
    if length > 30000:
        pairs = pairs[:5]  # the full-matrix oracle needs minutes for 10^5 x 10^5 unrelated bases (kept at 10^4)
    exp = np.array([orc.edit_distance(a, b) if max(len(a), len(b)) <= 10000 else orc.edit_distance_banded(a, b)
                    for a, b in pairs], dtype=np.int64)
    pool, ao, al, bo, bl = _pool(pairs)
    got = svx_ctx.edit_distance_batch(pool, ao, al, bo, bl).astype(np.int64)
    assert np.array_equal(got, exp)
    for k in (0, 200):
        thr = svx_ctx.edit_distance_batch(pool, ao, al, bo, bl, k_max=k).astype(np.int64)
        le = exp <= k
        assert np.array_equal(thr[le], exp[le]) and np.all(thr[~le] > k)
    assert exp[0] == 0 and 0 < exp[1] <= 200 and exp[3] > 200


def test_cut_off_between_strips(svx_ctx):
    """Multi-strip patterns (5-6 strips of 4096 rows) whose distance exceeds the band early, late, or not at all: the
    bit-vector kernel leaves a pair as soon as every value on a strip's last row exceeds the band (Ukkonen), which must
    never change an answer — thresholds exactly at, one below and one above the distance, exact requests (the band is
    widened from where the narrow one failed), edits piled up at the start, at the end, or spread evenly, a length
    difference that alone nearly fills the band.  (With the wavefront pass switched off — the kernel alone — and on:
    the module's fixture.)"""
    rng = np.random.default_rng(4242)
    L = 21000
    base = _rand_dna(rng, L)
    junk = lambda n: _rand_dna(rng, n)
    pairs = [(junk(5000) + base[5000:], base),                  # unrelated first quarter, identical rest
             (base[:18000] + junk(3000), base),                 # identical, then an unrelated tail
             (base, _edited(rng, base, 600, 20, 30)),           # ~3 % spread evenly
             (base[150:], _edited(rng, base, 12, 0, 1)),        # 150-base deletion at the very start + a few substitutions
             (junk(190) + base, base),                          # 190-base insertion in front
             (base, base[:9000] + base[9180:]),                 # 180-base deletion in the middle
             (base, _edited(rng, base, 3, 1, 2)),               # a handful
             (junk(3000) + base[3000:], base[:17000] + junk(4000))]   # bad at both ends
    exp = np.array([orc.edit_distance_banded(a, b) for a, b in pairs], dtype=np.int64)
    pool, ao, al, bo, bl = _pool(pairs)
    assert np.array_equal(svx_ctx.edit_distance_batch(pool, ao, al, bo, bl).astype(np.int64), exp)
    ks = sorted({0, 1, 150, 200, 256, 1000, 5000} | {int(d) + e for d in exp for e in (-1, 0, 1) if int(d) + e >= 0})
    for k in ks:
        thr = svx_ctx.edit_distance_batch(pool, ao, al, bo, bl, k_max=k).astype(np.int64)
        le = exp <= k
        assert np.array_equal(thr[le], exp[le]) and np.all(thr[~le] > k), k
    assert exp[0] > 1000 and exp[1] > 1000 and 150 <= exp[3] <= 200 and exp[4] == 190 and exp[5] == 180


def test_unrelated_long_pair_exact(svx_ctx):
    """3 x 10^4 unrelated bases on each side: every band falls short until the full matrix is visited."""
    rng = np.random.default_rng(77)
    a, b = _rand_dna(rng, 30000), _rand_dna(rng, 28000)
    pool, ao, al, bo, bl = _pool([(a, b)])
    assert int(svx_ctx.edit_distance_batch(pool, ao, al, bo, bl)[0]) == orc.edit_distance(a, b)


def test_strip_and_block_boundaries(svx_ctx):
    """Pattern lengths around 64-row blocks and 4096-row strips, text lengths around the 64-column
    symbol chunks, with edits near both ends."""
    rng = np.random.default_rng(5)
    pairs = []
    for m in (1, 2, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 8191, 8192, 8193, 12289):
        a = _rand_dna(rng, m)
        for b in (a, a[1:], a[:-1], b"T" + a, a + b"G", _edited(rng, a, min(m, 5), min(m, 3), 4)):
            pairs.append((a, b))
    for n in (1, 63, 64, 65, 128, 129):
        a = _rand_dna(rng, 4200)
        pairs.append((a, a[100:100 + n]))
    exp = np.array([orc.edit_distance(a, b) for a, b in pairs], dtype=np.uint32)
    pool, ao, al, bo, bl = _pool(pairs)
    assert np.array_equal(svx_ctx.edit_distance_batch(pool, ao, al, bo, bl), exp)
    got = svx_ctx.edit_distance_batch(pool, ao, al, bo, bl, k_max=3).astype(np.int64)
    le = exp <= 3
    assert np.array_equal(got[le], exp[le].astype(np.int64)) and np.all(got[~le] > 3)


def test_rich_alphabets(svx_ctx):
    """More than 16 distinct bytes in the pattern (IUPAC + lower case + arbitrary bytes, up to all 256):
    the 256-symbol instantiation; mixed in one batch with plain DNA pairs."""
    rng = np.random.default_rng(9)
    pairs = []
    for sigma, n in ((17, 300), (40, 900), (256, 700), (256, 5000), (200, 64)):
        alpha = rng.permutation(256)[:sigma].astype(np.uint8)
        a = alpha[rng.integers(0, sigma, n)].tobytes()
        b = bytearray(a)
        for _ in range(n // 20):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        del b[n // 3:n // 3 + 7]
        pairs.append((a, bytes(b)))
        pairs.append((_rand_dna(rng, n), _rand_dna(rng, n + 5)))
    exp = np.array([orc.edit_distance(a, b) for a, b in pairs], dtype=np.uint32)
    pool, ao, al, bo, bl = _pool(pairs)
    assert np.array_equal(svx_ctx.edit_distance_batch(pool, ao, al, bo, bl), exp)


def test_many_pairs_mixed_lengths(svx_ctx):
    """A PAIR-sized batch (20 k haplotype pairs of 240..2500 bases plus a few long ones)."""
    rng = np.random.default_rng(21)
    pairs = []
    for i in range(20000):
        a = _rand_dna(rng, int(rng.integers(240, 2500)))
        pairs.append((a, _edited(rng, a, int(rng.integers(0, 30)), int(rng.integers(0, 4)), 60)))
    for L in (20000, 50000):
        a = _rand_dna(rng, L)
        pairs.append((a, _edited(rng, a, 100, 5, 50)))
    pool, ao, al, bo, bl = _pool(pairs)
    got = svx_ctx.edit_distance_batch(pool, ao, al, bo, bl, k_max=200).astype(np.int64)
    pick = list(rng.integers(0, 20000, 400)) + [20000, 20001]
    for i in pick:
        a, b = pairs[i]
        e = orc.edit_distance(a, b) if len(a) <= 3000 else orc.edit_distance_banded(a, b)
        assert got[i] == e if e <= 200 else got[i] > 200


def _assemble(pool, pieces):
    """Python restatement of the piece semantics (include/svx.h): the string a haplotype recipe denotes."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    out = []
    for off, ln, rep, flags in pieces:
        s = bytes(pool[off:off + ln]).decode("latin-1")
        if flags & 1:
            s = s.upper()
        if flags & 2:
            s = "".join(comp.get(b, b) for b in reversed(s))
        out.append(s * rep)
    return "".join(out)


def test_haplotype_distance_batch_assembles_like_the_reference(svx_ctx):
    """svx_haplotype_distance_batch: three pieces per haplotype (prefix, middle, suffix) with upper-casing,
    reverse complement and repeats == the strings compute_distance builds (SVIM_COMBINE.py:43-100) followed by
    the exact edit distance."""
    from svim_asm_amd import _lib
    rng = np.random.default_rng(12)
    pool = np.frombuffer("".join(rng.choice(list("ACGTacgtNnRy"), size=60000)).encode(), dtype=np.uint8)
    n_pairs = 400
    pieces = np.zeros(n_pairs * 6, dtype=_lib.HAP_PIECE_DTYPE)
    for k in range(n_pairs * 6):
        kind = k % 3
        if kind == 1:  # middle: absent, reverse complement, repeated, or as it is
            mode = int(rng.integers(0, 4))
            ln = int(rng.integers(0, 900)) if mode else 0
            rep = [0, 1, int(rng.integers(1, 5)), 1][mode]
            flags = [0, 3, 1, 0][mode]
        else:
            ln, rep, flags = int(rng.integers(0, 300)), 1, 1
        if ln == 0 or rep == 0:
            ln = rep = flags = 0
        pieces[k] = (int(rng.integers(0, len(pool) - 1000)) if ln else 0, ln, rep, flags)
    exp = []
    for p in range(n_pairs):
        a = _assemble(pool, pieces[p * 6:p * 6 + 3].tolist())
        b = _assemble(pool, pieces[p * 6 + 3:p * 6 + 6].tolist())
        exp.append(orc.edit_distance(a.encode("latin-1"), b.encode("latin-1")))
    got = svx_ctx.haplotype_distance_batch(pool, pieces, 0xFFFFFFFF)
    assert got.tolist() == exp
    thr = svx_ctx.haplotype_distance_batch(pool, pieces, 200)
    assert all((g == e) if e <= 200 else g == 0xFFFFFFFF for g, e in zip(thr.tolist(), exp))
    # one call with a threshold per pair (the PAIR step's form): 0xFFFFFFFF = exact, else "exact when <= k"
    per_pair = rng.choice(np.array([0xFFFFFFFF, 200, 0, 37, 1 << 31], dtype=np.uint32), size=n_pairs)
    mixed = svx_ctx.haplotype_distance_batch_mixed(pool, pieces, per_pair)
    assert all((g == e) if e <= int(k) else g == 0xFFFFFFFF for g, e, k in zip(mixed.tolist(), exp, per_pair.tolist()))
    # a piece that reads past the pool is rejected, not read
    bad = pieces[:6].copy()
    bad[2] = (len(pool) - 10, 100, 1, 1)
    with pytest.raises(_lib.SvxError):
        svx_ctx.haplotype_distance_batch(pool, bad, 10)


@pytest.mark.parametrize("k", [0, 2, 100])
def test_threshold_applies_to_pairs_with_an_empty_side(svx_ctx, k):
    """Pairs the first stage settles on its own (one side empty: the distance is the other side's length) still
    obey the threshold contract: > k comes back as 0xFFFFFFFF (regression: fuzz seed 20036)."""
    pairs = [(b"", b"ACGT"), (b"ACGT" * 20, b""), (b"A", b"A"), (b"", b""), (b"AC", b"")]
    pool = np.frombuffer(b"".join(a + b for a, b in pairs), np.uint8)
    ao, al, bo, bl, o = [], [], [], [], 0
    for a, b in pairs:
        ao.append(o); al.append(len(a)); o += len(a)
        bo.append(o); bl.append(len(b)); o += len(b)
    got = svx_ctx.edit_distance_batch(pool, ao, al, bo, bl, k_max=k).tolist()
    exp = [max(len(a), len(b)) if (not a or not b) else 0 for a, b in pairs]
    assert got == [e if e <= k else 0xFFFFFFFF for e in exp]


def test_largest_wavefront_cap(svx_ctx):
    """svx_ctx_set_edit_wavefront_cap(4096): the first stage then needs more dynamic LDS than the default launch
    limit; distances between 1024 and 4096 edits are resolved by it."""
    rng = np.random.default_rng(77)
    a = bytes(rng.choice(list(b"ACGT"), size=60000).astype(np.uint8))
    b = mutate(rng, a, 3000)
    pool = np.frombuffer(a + b, np.uint8)
    svx_ctx.set_edit_wavefront_cap(4096)
    try:
        got = int(svx_ctx.edit_distance_batch(pool, [0], [len(a)], [len(a)], [len(b)])[0])
    finally:
        svx_ctx.set_edit_wavefront_cap(1024)
    assert got == orc.edit_distance_banded(a, b) and got > 1024
