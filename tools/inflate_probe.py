#!/usr/bin/env python3
"""Inflate probe (GPU box or build container): per-member cost of the build's DEFLATE decoder (svx_inflate_raw)
beside zlib and, when installed, libdeflate, on the BGZF members of one BAM — SEQ members (poorly compressible:
> 25 000 B compressed) and the others apart — whole and up to the member's middle.
With a third argument N the SEQ members are first re-compressed by libdeflate at level N (htslib is usually built
with libdeflate; its level-6 streams are 4-bit literals almost only, where zlib's are literals and short matches).
    python tools/inflate_probe.py DIR/hap1.bam [n_members [libdeflate_level]]"""
import ctypes as C
import os
import random
import struct
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svim_asm_amd import _lib

lib = _lib.load()
raw = open(sys.argv[1], "rb").read(400 << 20)
n_pick = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
spans, p = [], 0
while p + 18 <= len(raw) - 70000:
    xlen = struct.unpack_from("<H", raw, p + 10)[0]
    bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
    isz = struct.unpack_from("<I", raw, p + bsize - 4)[0]
    spans.append((p + 12 + xlen, bsize - xlen - 20, isz))
    p += bsize
try:
    L = C.CDLL("libdeflate.so.0")
    L.libdeflate_alloc_decompressor.restype = C.c_void_p
    L.libdeflate_deflate_decompress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    ld = L.libdeflate_alloc_decompressor()
except OSError:
    L = None
relevel = int(sys.argv[3]) if len(sys.argv) > 3 else None
if relevel is not None and L is not None:
    L.libdeflate_alloc_compressor.restype = C.c_void_p
    L.libdeflate_alloc_compressor.argtypes = [C.c_int]
    L.libdeflate_deflate_compress.restype = C.c_size_t
    L.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    comp = L.libdeflate_alloc_compressor(relevel)
out = C.create_string_buffer(65536)
got, n = C.c_size_t(), C.c_uint64()
half = (C.c_uint64 * 1)()
for name, grp in (("SEQ members", [s for s in spans if s[1] > 25000]), ("other members", [s for s in spans if s[1] <= 25000 and s[2] > 30000])):
    if not grp:
        continue
    random.seed(1)
    pick = random.sample(grp, min(n_pick, len(grp)))
    blobs = [raw[st:st + ln] for st, ln, _ in pick]
    if relevel is not None and L is not None and name.startswith("SEQ"):
        again = []
        for b, (st, ln, isz) in zip(blobs, pick):
            data = zlib.decompress(b, -15)
            buf = C.create_string_buffer(70000)
            k = L.libdeflate_deflate_compress(comp, data, len(data), buf, 70000)
            again.append(buf.raw[:k])
        blobs = again
        pick = [(0, len(b), isz) for b, (_, _, isz) in zip(blobs, pick)]
        name = "SEQ members re-compressed by libdeflate level %d" % relevel
    res = {}
    t = time.perf_counter()
    for b in blobs:
        zlib.decompress(b, -15)
    res["zlib"] = (time.perf_counter() - t) / len(pick) * 1e6
    t = time.perf_counter()
    for b, (_, ln, isz) in zip(blobs, pick):
        zlib.decompressobj(-15).decompress(b, isz // 2)
    res["zlib_to_half"] = (time.perf_counter() - t) / len(pick) * 1e6
    if L:
        t = time.perf_counter()
        for b, (_, ln, isz) in zip(blobs, pick):
            assert L.libdeflate_deflate_decompress(ld, b, ln, out, isz, C.byref(got)) == 0
        res["libdeflate"] = (time.perf_counter() - t) / len(pick) * 1e6
    t = time.perf_counter()
    for b, (_, ln, isz) in zip(blobs, pick):
        assert lib.svx_inflate_raw(b, ln, out, isz, None, 0, C.byref(n)) == 0 and n.value == isz
    res["own"] = (time.perf_counter() - t) / len(pick) * 1e6
    t = time.perf_counter()
    for b, (_, ln, isz) in zip(blobs, pick):
        half[0] = isz // 2
        lib.svx_inflate_raw(b, ln, out, isz // 2 + 300, half, 1, C.byref(n))  # ends in "more than cap": only the prefix is timed
    res["own_to_half"] = (time.perf_counter() - t) / len(pick) * 1e6
    out2 = C.create_string_buffer(65536)
    n2, ra, rb = C.c_uint64(), C.c_int(), C.c_int()
    END = 2 ** 64 - 1
    t = time.perf_counter()
    for k in range(0, len(pick) - 1, 2):
        (_, la, ia), (_, lb, ib) = pick[k], pick[k + 1]
        lib.svx_inflate_raw_pair(blobs[k], la, out, ia, END, C.byref(n), C.byref(ra), blobs[k + 1], lb, out2, ib, END, C.byref(n2), C.byref(rb))
        assert ra.value == 0 and rb.value == 0
    res["own_pair"] = (time.perf_counter() - t) / (len(pick) // 2 * 2) * 1e6
    t = time.perf_counter()
    for k in range(0, len(pick) - 1, 2):
        (_, la, ia), (_, lb, ib) = pick[k], pick[k + 1]
        lib.svx_inflate_raw_pair(blobs[k], la, out, ia, ia // 2, C.byref(n), C.byref(ra), blobs[k + 1], lb, out2, ib, ib // 2, C.byref(n2), C.byref(rb))
    res["own_pair_to_half"] = (time.perf_counter() - t) / (len(pick) // 2 * 2) * 1e6
    for b, (_, ln, isz) in list(zip(blobs, pick))[:100]:
        lib.svx_inflate_raw(b, ln, out, isz, None, 0, C.byref(n))
        assert out.raw[:isz] == zlib.decompress(b, -15)
    print("%s (%d, avg %d B compressed): " % (name, len(pick), sum(s[1] for s in pick) / len(pick)) +
          "  ".join("%s %.0f" % kv for kv in res.items()) + "  us/member")
