#!/usr/bin/env python3
"""The native BAM reader against the pure-Python one on files nobody tuned it for (no GPU): a small synthetic
haplotype BAM is re-blocked at random BGZF payload sizes (1 KiB … 64 KiB: slices and records then span many members)
and re-compressed at random zlib levels / strategies (stored blocks, fixed-code blocks, Huffman-only, RLE), indexed,
and read by both readers — records, CIGARs, tags, and random base slices; with and without whole-member verification;
with the build's decoder and with zlib; with the walks' check deferred (bamio defer_verify).
    python tools/fuzz_bam_reader.py [--seconds 120] [--seed 1]"""
import argparse
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from svim_asm_amd import bamio, synth_bam  # noqa: E402

EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def member(payload, level, strategy):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    body = c.compress(payload) + c.flush()
    assert len(body) + 26 <= 65536
    head = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25)
    return head + body + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


def reblock(src, dst, rng):
    data = bamio.bgzf_decompress(src)
    out, p = [], 0
    while p < len(data):
        size = int(rng.choice([1024, 4096, 20000, 0xFF00])) if rng.random() < 0.8 else int(rng.integers(1, 0xFF00))
        level = int(rng.choice([0, 1, 6, 9]))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
        if level == 0:
            size = min(size, 60000)  # stored blocks grow by five bytes per block
        out.append(member(bytes(data[p:p + size]), level, strategy))
        p += size
        if rng.random() < 0.02:
            out.append(member(b"", 6, zlib.Z_DEFAULT_STRATEGY))  # an empty member in the middle of the file
    open(dst, "wb").write(b"".join(out) + EOF)
    bamio.index_bam(dst)


def columns_digest(f, rng_seed):
    import hashlib
    h = hashlib.sha256()
    for k in ("tid", "pos", "flag", "mapq", "l_seq", "n_cig", "ref_len"):  # (both readers; voffset is compared by the tests)
        h.update(np.ascontiguousarray(f._cols[k]).astype(np.int64).tobytes())
    h.update(np.ascontiguousarray(f._cigar).tobytes())
    for i in range(len(f)):
        r = f.record(i)
        h.update(r.query_name.encode())
        h.update(bytes(r._tags_raw))
    rng = np.random.default_rng(rng_seed)
    n = len(f)
    if n:
        rec = np.sort(rng.integers(0, n, 300))
        l = np.maximum(f._cols["l_seq"][rec], 1)
        lo = (rng.random(300) * l).astype(np.int64)
        ln = rng.choice([1, 2, 7, 300, 5000, 200000], 300)
        o = np.lexsort((lo, rec))
        for s in f.sequence_slices(rec[o], lo[o], lo[o] + ln[o]):
            h.update(s.encode())
        h.update(f.record(int(rec[0])).seq_slice(0, int(l[0])).encode())
    return h.hexdigest()


CHILD = r"""
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from svim_asm_amd import bamio
import fuzz_bam_reader as F
defer = len(sys.argv) > 5 and sys.argv[5] == "defer"
f = bamio.AlignmentFile(sys.argv[1], reader=sys.argv[2], verify=(sys.argv[3] == "1") if sys.argv[2] == "native" else None,
                        device=0 if defer else None)
if defer:  # the walks' check left to the next sequence call (its device leg where there is one, the threads otherwise)
    f.device_inflate_percent = 100
    f.device_inflate_min_members = 0
    f.defer_verify = True
f.load()
print(F.columns_digest(f, int(sys.argv[4])))
""" % (ROOT, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    t_end = time.time() + args.seconds
    seed, cases = args.seed, 0
    tmp = tempfile.mkdtemp(prefix="svx_fuzz_bam_")
    try:
        while time.time() < t_end:
            rng = np.random.default_rng(seed)
            d = os.path.join(tmp, "d")
            shutil.rmtree(d, ignore_errors=True)
            _, bams = synth_bam.write_dataset(d, seed=seed, contigs=(("chr1", 120000), ("chr10", 60000), ("chr2", 40000)),
                                              diploid=False, n_shared=int(rng.integers(3, 30)), n_private=3,
                                              median_aln=int(rng.choice([3000, 30000, 100000])), mean_m=int(rng.choice([50, 400, 2000])))
            dst = os.path.join(tmp, "re.bam")
            reblock(bams[0], dst, rng)
            want = columns_digest(bamio.AlignmentFile(dst, reader="python").load(), seed)
            for env, verify, *more in (({}, "0"), ({}, "1"), ({"SVX_BAM_ZLIB": "1"}, "0"), ({"SVX_BAM_ZLIB": "1"}, "1"), ({}, "1", "defer")):
                got = subprocess.run([sys.executable, "-c", CHILD, dst, "native", verify, str(seed)] + list(more), env=dict(os.environ, **env),
                                     check=True, capture_output=True, text=True).stdout.strip()
                if got != want:
                    keep = os.path.join(ROOT, "gpurun_out", "fuzz_bam_reader_seed%d.bam" % seed)
                    os.makedirs(os.path.dirname(keep), exist_ok=True)
                    shutil.copy(dst, keep)
                    print("MISMATCH seed %d env %r verify %s: kept %s" % (seed, env, verify, keep))
                    return 1
            cases += 1
            seed += 1
        print("fuzz_bam_reader ok: %d files x 5 reader modes, seeds %d..%d" % (cases, args.seed, seed - 1))
        return 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
