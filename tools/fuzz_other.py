#!/usr/bin/env python3
"""Randomised differential tests of the other C-ABI kernels against the C oracle:
svx_segments_classify, svx_segments_postpass, svx_pair_partition, svx_edit_distance_batch,
svx_haplotype_distance_batch, svx_linkage_cut_batch, svx_cigar_stats, svx_collect_batch (the pair sort on all three plans).

    python tools/fuzz_other.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from oracle import orc, svim_oracle  # noqa: E402
from svim_asm_amd import _lib, synth  # noqa: E402
import test_gpu_segments as tseg  # noqa: E402  (random_reads)
import test_gpu_pair as tpair  # noqa: E402  (make_keys)


def fuzz_segments(ctx, rng):
    n_reads = int(rng.choice([1, 3, 40, 700, 5000]))
    max_k = int(rng.choice([0, 1, 2, 5, 8, 9, 30, 120]))
    segs, off, rl = tseg.random_reads(rng, n_reads, max_k, n_contigs=int(rng.integers(1, 5)), spread=int(rng.choice([1, 30, 300, 3000])))
    prm = (int(rng.choice([1, 40, 50, 1000])), int(rng.choice([20, 1000, 100000, 1 << 30])),
           int(rng.choice([0, 50, 500])), int(rng.choice([0, 50, 500])), int(rng.choice([0, 50, 500])), int(rng.choice([0, 50, 500])))
    got = ctx.segments_classify(segs, off, rl, prm)
    exp = orc.segments_classify(segs, off, rl, prm)
    return np.array_equal(got, exp), "segments n_reads %d max_k %d prm %s" % (n_reads, max_k, prm)


def fuzz_pair(ctx, rng):
    n = int(rng.choice([0, 1, 2, 63, 64, 65, 1023, 1024, 1025, 5000, 5121, 20000, 70000, 131072, 131073, 400000]))
    groups = int(rng.choice([1, 2, 24, 144, 5000]))
    pos_max = int(rng.choice([1, 5, 1000, 250_000_000, (1 << 32) - 1]))
    keys = tpair.make_keys(rng, n, groups, pos_max, dup_frac=float(rng.choice([0.0, 0.3, 0.9])))
    shape = str(rng.choice(["as is", "skewed", "runs", "two lists"]))
    if n and shape == "skewed":     # most keys in one bucket of the leading bits
        crowd = rng.random(n) < float(rng.choice([0.5, 0.9, 0.999]))
        keys = np.where(crowd, keys & np.uint64(0xFFF), keys)
    elif n and shape == "runs":     # a few sorted runs one after the other (the run-merging form of the window sort)
        cuts = np.sort(rng.integers(0, n + 1, int(rng.choice([1, 2, 5, 31, 32, 40, 200]))))
        keys = np.concatenate([np.sort(part) for part in np.split(keys, cuts)])
    elif n and shape == "two lists":
        keys = np.concatenate([np.sort(keys[:n // 2]), np.sort(keys[n // 2:])])
    if rng.random() < 0.2 and n:
        keys |= np.uint64(int(rng.integers(0, 1 << 20))) << np.uint64(44)  # high group bits in use
    md = int(rng.choice([0, 1, 1000, 1 << 20, (1 << 32) - 1]))
    plan = str(rng.choice(["one launch", "one launch", "one launch", "radix", "wait-free"]))
    single = plan == "one launch"
    ctx.set_pair_single_launch_max(0 if plan == "radix" else 131072)
    ctx.set_pair_wait_free(plan == "wait-free")
    perm, part, n_parts = ctx.pair_partition(keys, md)
    ctx.set_pair_single_launch_max(131072)
    ctx.set_pair_wait_free(False)
    e_perm, e_part, e_n = orc.pair_partition(keys, md)
    ok = n_parts == e_n and np.array_equal(perm, e_perm) and np.array_equal(part, e_part)
    return ok, "pair n %d groups %d pos_max %d max_dist %d shape %s plan %s" % (n, groups, pos_max, md, shape, plan)


def fuzz_edit(ctx, rng):
    n = int(rng.choice([1, 2, 17, 200]))
    alphabet = np.frombuffer(b"ACGT" if rng.random() < 0.8 else b"ACGTNacgt", dtype=np.uint8)
    seqs, a_off, a_len, b_off, b_len = [], [], [], [], []
    pos = 0
    for _ in range(n):
        la = int(rng.choice([0, 1, 2, 63, 64, 65, 300, 2000]))
        a = alphabet[rng.integers(0, len(alphabet), size=la)]
        mode = rng.random()
        if mode < 0.3:
            b = a.copy()
        elif mode < 0.7 and la > 0:  # a few edits
            b = list(a)
            for _ in range(int(rng.integers(1, 12))):
                j = int(rng.integers(0, len(b) + 1))
                r = rng.random()
                if r < 0.34 and b:
                    b[min(j, len(b) - 1)] = alphabet[int(rng.integers(0, len(alphabet)))]
                elif r < 0.67:
                    b.insert(j, alphabet[int(rng.integers(0, len(alphabet)))])
                elif b:
                    b.pop(min(j, len(b) - 1))
            b = np.array(b, dtype=np.uint8)
        else:
            b = alphabet[rng.integers(0, len(alphabet), size=int(rng.choice([0, 1, 64, 300, 1500])))]
        for arr, offs, lens in ((a, a_off, a_len), (b, b_off, b_len)):
            seqs.append(arr); offs.append(pos); lens.append(len(arr)); pos += len(arr)
    pool = np.concatenate(seqs) if pos else np.zeros(0, np.uint8)
    k = int(rng.choice([0, 1, 5, 200, 5000, 0xFFFFFFFF]))
    ctx.set_edit_wavefront_cap(int(rng.choice([0, 3, 48, 1024, 4096])))
    got = ctx.edit_distance_batch(pool, np.array(a_off, np.uint64), np.array(a_len, np.uint32),
                                  np.array(b_off, np.uint64), np.array(b_len, np.uint32), k)
    ok = True
    for i in range(n):
        d = orc.edit_distance(pool[a_off[i]:a_off[i] + a_len[i]].tobytes(), pool[b_off[i]:b_off[i] + b_len[i]].tobytes())
        e = d if (k == 0xFFFFFFFF or d <= k) else 0xFFFFFFFF
        ok = ok and int(got[i]) == e
    return ok, "edit n %d k %d" % (n, k)


def fuzz_stats(ctx, rng):
    n_aln = int(rng.choice([1, 5, 300, 4000]))
    cig, off, _ = synth.random_cigar_case(rng, n_aln, max_ops=int(rng.choice([0, 3, 70, 700, 20000])))
    if rng.random() < 0.5 and len(cig):  # leading/trailing clips of every flavour
        starts = off[:-1][np.diff(off.astype(np.int64)) > 2].astype(np.int64)
        for s0 in starts[: 200]:
            cig[s0] = (int(rng.integers(0, 500)) << 4) | int(rng.choice([4, 5]))
            cig[s0 + 1] = (int(rng.integers(0, 500)) << 4) | int(rng.choice([4, 5, 0]))
    got = ctx.cigar_stats(cig, off)
    exp = orc.cigar_stats(cig, off)
    ok = all(np.array_equal(got[k], exp[k]) for k in exp)
    return ok, "stats n_aln %d n_ops %d" % (n_aln, len(cig))


def fuzz_linkage(ctx, rng):
    n_parts = int(rng.choice([1, 7, 300, 3000]))
    sizes, flat = [], []
    for _ in range(n_parts):
        n = int(rng.choice([1, 2, 2, 2, 3, 4, 5, 7, 10, 11, 14, 40]))
        m = n * (n - 1) // 2
        kind = int(rng.integers(0, 4))
        if kind == 0:
            v = rng.integers(0, int(rng.choice([2, 5, 400])), m).astype(float)
            v[rng.random(m) < 0.4] = 1000000000.0
        elif kind == 1:
            v = rng.integers(0, 2000, m) / 3000
            v[rng.random(m) < 0.3] = 99999
        elif kind == 2:
            v = 1 - rng.integers(0, 11, m) / 10.0
        else:
            v = rng.random(m)
        sizes.append(n)
        flat.extend(v.tolist())
    cutoff = float(rng.choice([0.3, 0.5, 1.0, 3.0, 200.0, -1.0]))
    got = ctx.linkage_cut_batch(flat, sizes, cutoff)
    at = fi = 0
    ok = True
    for n in sizes:
        m = n * (n - 1) // 2
        exp = orc.linkage_cut(flat[fi:fi + m], n, cutoff) if n > 1 else [1]
        ok = ok and list(got[at:at + n]) == list(exp)
        at += n
        fi += m
    return ok, "linkage n_parts %d cutoff %g" % (n_parts, cutoff)


def fuzz_postpass(ctx, rng):
    n_contigs = int(rng.choice([2, 3, 12]))
    names = ["chr%d" % (i + 1) for i in range(n_contigs)]
    rank = np.zeros(n_contigs, np.int32)
    for k, i in enumerate(sorted(range(n_contigs), key=lambda i: names[i])):
        rank[i] = k
    reads = [tseg._random_raw_read(rng, int(rng.choice([0, 1, 2, 3, 5, 9, 25])), n_contigs, bool(rng.random() < 0.7))
             for _ in range(int(rng.choice([1, 50, 800])))]
    read_off = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.uint32)
    raw = np.zeros(int(read_off[-1]), dtype=_lib.RAW_DTYPE)
    k = 0
    for r in reads:
        for row in r:
            for name_, v in zip(("kind", "a0", "a1", "a2", "a3", "a4", "a5"), row):
                raw[k][name_] = v
            k += 1
    prm = (int(rng.choice([1, 10, 40])), int(rng.choice([300, 100000])), 50, 50, 50, 50)
    post, first = ctx.segments_postpass(raw, read_off, rank, prm)
    code = {1: "TANDEM", 2: "DUP_INT", 3: "INV"}
    width = {"TANDEM": 5, "DUP_INT": 6, "INV": 4}
    ok = True
    for i, r in enumerate(reads):
        exp = svim_oracle.postpass_records(r, rank.tolist(), prm[0], prm[1])
        got = []
        for rec in post[first[i]:first[i + 1]]:
            kind = code[int(rec["kind"])]
            vals = [int(rec[n]) for n in ("a0", "a1", "a2", "a3", "a4", "a5")][:width[kind]]
            if kind != "DUP_INT":
                vals[-1] = bool(vals[-1])
            got.append((kind,) + tuple(vals))
        ok = ok and got == exp
    return ok, "postpass reads %d contigs %d prm %s" % (len(reads), n_contigs, prm)


def fuzz_haplotypes(ctx, rng):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_editdist as ted
    pool = np.frombuffer("".join(rng.choice(list("ACGTacgtNn"), size=int(rng.choice([200, 5000, 40000])))).encode(), dtype=np.uint8)
    n_pairs = int(rng.choice([1, 9, 150]))
    pieces = np.zeros(n_pairs * 6, dtype=_lib.HAP_PIECE_DTYPE)
    for k in range(n_pairs * 6):
        if k % 3 == 1:
            mode = int(rng.integers(0, 4))
            ln = int(rng.integers(0, min(4500, len(pool) // 2))) if mode else 0
            rep = [0, 1, int(rng.integers(1, 4)), 1][mode]
            flags = [0, 3, 1, 0][mode]
        else:
            ln, rep, flags = int(rng.integers(0, 150)), 1, 1
        if ln == 0 or rep == 0:
            ln = rep = flags = 0
        pieces[k] = (int(rng.integers(0, len(pool) - ln)) if ln else 0, ln, rep, flags)
    kmax = int(rng.choice([0, 200, 0xFFFFFFFF]))
    ctx.set_edit_wavefront_cap(int(rng.choice([0, 48, 1024])))
    got = ctx.haplotype_distance_batch(pool, pieces, kmax).tolist()
    ok = True
    for p in range(n_pairs):
        a = ted._assemble(pool, pieces[p * 6:p * 6 + 3].tolist())
        b = ted._assemble(pool, pieces[p * 6 + 3:p * 6 + 6].tolist())
        e = orc.edit_distance_banded(a.encode("latin-1"), b.encode("latin-1"))
        ok = ok and (got[p] == e if (kmax == 0xFFFFFFFF or e <= kmax) else got[p] == 0xFFFFFFFF)
    return ok, "haplotypes pairs %d kmax %d" % (n_pairs, kmax)


def fuzz_collect(ctx, rng):
    """svx_collect_batch (one submission) against the composition of the single-purpose entry points AND against the
    oracle alone (signatures, raw records from oracle-derived rows, derived records from the record-level post-passes);
    random batches and reads whose segments tile the read (plenty of derived records); from 384 reads on the chain's
    reads are dealt by the op-count table."""
    import test_gpu_collect as tcol
    import helpers
    if rng.random() < 0.3:
        b = tcol.tiling_batch(rng, n_reads=int(rng.choice([1, 20, 383, 384, 1500])), n_contigs=int(rng.choice([1, 3, 4])))
    else:
        b = tcol.random_batch(rng, n_aln=int(rng.choice([1, 2, 50, 800, 6000])), n_parts=int(rng.choice([1, 2, 3, 5])),
                              n_reads=int(rng.choice([0, 1, 7, 300, 384, 2000])), max_supp=int(rng.choice([1, 3, 7, 12])),
                              long_read=bool(rng.random() < 0.3), long_aln=float(rng.choice([0.0, 0.0, 0.2, 0.8])),
                              long_max=int(rng.choice([300, 3000, 12000, 40000])))
    min_len = int(rng.choice([1, 30, 40, 500]))
    prm = (int(rng.choice([1, 40, 50, 1000])), int(rng.choice([20, 1000, 100000, 1 << 30])),
           int(rng.choice([0, 50, 500])), int(rng.choice([0, 50, 500])), int(rng.choice([0, 50, 500])), int(rng.choice([0, 50, 500])))
    streaming = bool(rng.random() < 0.25)
    ctx.set_small_batch_ops(0 if streaming else 1 << 23)
    try:
        got = tcol.call(ctx.collect_batch, b, min_len, prm)
        exp = tcol.call(ctx.collect_batch_composed, b, min_len, prm)
    finally:
        ctx.set_small_batch_ops(1 << 23)
    try:
        tcol.same(got, exp)
        tcol.same_as_oracle(got, tcol.call(helpers.oracle_collect, b, min_len, prm))
        ok = True
    except AssertionError:
        ok = False
    return ok, "collect n_aln %d parts %d reads %d min_len %d prm %s streaming %s" % (
        len(b["aln_off"]) - 1, len(b["parts"]), len(b["read_off"]) - 1, min_len, prm, streaming)


def fuzz_inflate(ctx, rng):
    """svx_bgzf_inflate_dev (the kernel of the sequence slices' device leg) against zlib: members of random kinds of
    data at random levels / strategies / memLevels, a share of them damaged (flipped bit, cut short, wrong CRC32, wrong
    ISIZE): status 0 exactly where zlib and the trailer agree, bytes equal there."""
    import zlib

    def deflate(data, level, strategy, mem):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
        return c.compress(data) + c.flush()
    n = int(rng.choice([1, 3, 17, 64, 200]))
    payloads, isize, crc, want_ok, expect = [], [], [], [], []
    for _ in range(n):
        kind = int(rng.integers(0, 7))
        size = int(rng.choice([0, 1, 2, 300, 5000, 65000, 65536]))
        if kind == 0:
            d = rng.choice(np.frombuffer(bytes([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88, 0xFF]), np.uint8), size=size).tobytes()
        elif kind == 1:
            d = bytes(size)
        elif kind == 2:
            d = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
        elif kind == 3:
            d = (b"chr1\t12345\tsvim_asm.DEL.7\tACGTNNNN\t<DEL>\t.\tPASS\tSVTYPE=DEL;END=12400\n" * 1200)[:size]
        elif kind == 4:
            d = b"".join(bytes([int(x)]) * int(k) for x, k in zip(rng.integers(0, 256, 300), rng.integers(1, 400, 300)))[:size]
        elif kind == 5:
            d = rng.integers(0, 4, size, dtype=np.uint8).tobytes()  # tiny alphabet: long codes for the rare symbols
        else:
            base = rng.integers(0, 256, max(1, size // 7), dtype=np.uint8).tobytes()
            d = (base * 8)[:size]                                     # long-distance repeats
        p = deflate(d, int(rng.choice([0, 1, 2, 4, 6, 9])),
                    int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])),
                    int(rng.choice([1, 5, 9])))
        g_isize, g_crc, ok = len(d), zlib.crc32(d) & 0xFFFFFFFF, True
        damage = int(rng.integers(0, 8))
        bad = bytearray(p)
        if damage == 0 and len(bad):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        elif damage == 1 and len(bad) > 1:
            bad = bad[:int(rng.integers(1, len(bad)))]
        elif damage == 2:
            g_crc ^= 1 << int(rng.integers(0, 32))
        elif damage == 3:
            g_isize = max(0, min(65536, g_isize + int(rng.choice([-1, 1, -100, 7]))))
        if damage < 4:
            try:
                o = zlib.decompress(bytes(bad), -15)
                ok = o == d and len(o) == g_isize and (zlib.crc32(o) & 0xFFFFFFFF) == g_crc
            except zlib.error:
                ok = False
        payloads.append(bytes(bad)); isize.append(g_isize); crc.append(g_crc); want_ok.append(ok); expect.append(d)
    status, outs, _ = ctx.bgzf_inflate(payloads, isize, crc)
    good = all((st == 0) == ok for st, ok in zip(status.tolist(), want_ok)) and \
        all(o == e for o, e, ok in zip(outs, expect, want_ok) if ok)
    return good, "inflate n %d" % n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="", help="one of the fuzz_* functions, e.g. pair")
    a = ap.parse_args()
    ctx = _lib.default_context(0)
    fns = [fuzz_segments, fuzz_pair, fuzz_edit, fuzz_stats, fuzz_linkage, fuzz_postpass, fuzz_haplotypes, fuzz_collect, fuzz_inflate]
    if a.only:
        fns = [f for f in fns if f.__name__ == "fuzz_" + a.only]
    counts = {f.__name__: 0 for f in fns}
    t0, seed = time.time(), a.seed
    while time.time() - t0 < a.seconds:
        f = fns[seed % len(fns)]
        rng = np.random.default_rng(seed)
        ok, what = f(ctx, rng)
        if not ok:
            print("MISMATCH seed %d: %s" % (seed, what))
            sys.exit(1)
        counts[f.__name__] += 1
        seed += 1
    print("fuzz ok:", counts, "seeds %d..%d" % (a.seed, seed - 1))


if __name__ == "__main__":
    main()
