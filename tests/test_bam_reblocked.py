"""The native BAM reader on files it was not tuned for: the synthetic haplotype BAM re-blocked at random BGZF
payload sizes (records and base slices span many members) and re-compressed at random zlib levels and strategies
(stored, fixed-code, Huffman-only, RLE blocks; empty members), read by the native reader — the build's decoder and
zlib, inflating what is needed and whole members with CRC32 — and by the pure-Python reader: same records, CIGARs,
tags and bases (tools/fuzz_bam_reader.py is the long-running form; 281 files clean in the round's campaign)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reblocked_recompressed_files_read_the_same():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_bam_reader.py"), "--seconds", "12", "--seed", "7000"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0 and "fuzz_bam_reader ok" in res.stdout, res.stdout[-2000:]
    assert int(res.stdout.split("ok:")[1].split()[0]) >= 3
