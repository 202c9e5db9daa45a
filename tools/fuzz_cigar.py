#!/usr/bin/env python3
"""Randomised differential test of svx_cigar_extract (GPU, through the C-ABI) against the C oracle.

    python tools/fuzz_cigar.py [--seconds 120] [--seed 1]

Every case draws its own shape: number of alignments, ops per alignment (0 .. 30 000), share of
empty alignments, op alphabet, share of SV-sized / huge lengths, indel density up to all-indel
(dense tiles, queue overflow), min_len, packed or SoA layout, a too-small output capacity now and then.
Stops at the first mismatch (prints the seed) or after --seconds."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import orc  # noqa: E402
from svim_asm_amd import _lib  # noqa: E402

KEYS = ("aln", "ref_pos", "read_pos", "len", "type")


def draw(rng):
    n_aln = int(rng.choice([1, 2, 7, 60, 400, 3000]))
    max_ops = int(rng.choice([0, 1, 5, 17, 300, 5000, 30000]))
    n_ops_aln = rng.integers(0, max_ops + 1, size=n_aln)
    if rng.random() < 0.5:
        n_ops_aln[rng.random(n_aln) < rng.choice([0.05, 0.5, 0.9])] = 0
    if int(n_ops_aln.sum()) > 3_000_000:
        n_ops_aln = (n_ops_aln * (3_000_000 / n_ops_aln.sum())).astype(np.int64)
    off = np.concatenate(([0], np.cumsum(n_ops_aln))).astype(np.uint64)
    n = int(off[-1])
    alphabet = [np.arange(16), np.array([0, 1, 2]), np.array([1, 2]), np.array([0, 1, 2, 4, 5, 7, 8]), np.array([0, 3, 2])][int(rng.integers(0, 5))]
    ops = alphabet[rng.integers(0, len(alphabet), size=n)]
    min_len = int(rng.choice([0, 1, 30, 40, 41, 1000, 1 << 24, (1 << 28) - 1, 1 << 28, (1 << 32) - 1]))
    base = max(1, min(min_len, 5000))
    lens = rng.integers(0, 2 * base + 2, size=n)
    kind = rng.random(n)
    lens = np.where(kind < 0.1, rng.integers(max(0, min(min_len, (1 << 28) - 1) - 2), min(min_len, (1 << 28) - 4) + 3, size=n), lens)
    p_big = float(rng.choice([0.0, 0.0005, 0.05]))
    lens = np.where(rng.random(n) < p_big, rng.integers(1 << 22, 1 << 28, size=n), lens)
    cig = ((lens.astype(np.uint64) & ((1 << 28) - 1)).astype(np.uint32) << 4) | ops.astype(np.uint32)
    rs = rng.integers(0, 1 << 30, size=n_aln).astype(np.int32) if rng.random() < 0.8 else None
    return cig, off, rs, min_len


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = _lib.default_context(0)
    t0, n_cases, n_ops, n_sig = time.time(), 0, 0, 0
    seed = a.seed
    while time.time() - t0 < a.seconds:
        rng = np.random.default_rng(seed)
        cig, off, rs, min_len = draw(rng)
        exp = orc.cigar_extract(cig, off, rs, min_len)
        soa = rng.random() < 0.3
        streaming = rng.random() < 0.5  # both kernel paths: small-batch (two launches) and streaming (five)
        ctx.set_small_batch_ops(0 if streaming else 1 << 23)
        if soa:
            got = ctx.cigar_extract((cig >> 4).astype(np.uint32), off, rs, min_len, op=(cig & 15).astype(np.uint8))
        else:
            got = ctx.cigar_extract(cig, off, rs, min_len)
        for k in KEYS:
            if len(got[k]) != len(exp[k]) or not np.array_equal(got[k], exp[k]):
                print("MISMATCH seed %d key %s (n_aln %d n_ops %d min_len %d soa %s streaming %s): got %d exp %d" % (
                    seed, k, len(off) - 1, len(cig), min_len, soa, streaming, len(got[k]), len(exp[k])))
                sys.exit(1)
        n_cases += 1; n_ops += len(cig); n_sig += len(exp["aln"])
        seed += 1
    print("fuzz ok: %d cases, %d ops, %d signatures, seeds %d..%d, %.0f s" % (n_cases, n_ops, n_sig, a.seed, seed - 1, time.time() - t0))


if __name__ == "__main__":
    main()
