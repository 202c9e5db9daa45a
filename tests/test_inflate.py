"""The build's DEFLATE decoder (svim_asm_amd/csrc/svx_inflate.h, C-ABI svx_inflate_raw) against zlib — the decoder
the reference reaches through pysam → htslib (SVIM_COLLECT.py:68).  Streams are made by zlib's own deflate at every
level and strategy over the kinds of bytes a BAM holds; each is decoded whole and in resumed prefix steps, and
damaged copies must be refused or decode to what zlib decodes them to.  No GPU."""
import ctypes as C
import random
import zlib

import numpy as np
import pytest

from svim_asm_amd import _lib


def inflate(stream, cap, stops=()):
    lib = _lib.load()
    out = (C.c_uint8 * max(cap, 1))()
    st = (C.c_uint64 * max(len(stops), 1))(*stops)
    n = C.c_uint64(0)
    rc = lib.svx_inflate_raw(stream, len(stream), out, cap, st, len(stops), C.byref(n))
    return rc, bytes(out[: n.value])


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8, wbits=-15):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    return c.compress(data) + c.flush()


def seq_like(rng, n):
    """SEQ bytes: two 4-bit bases (1, 2, 4, 8) per byte."""
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)
    a = codes[rng.integers(0, 4, n)]
    b = codes[rng.integers(0, 4, n)]
    return ((a << 4) | b).tobytes()


def kinds(rng, n):
    yield "seq", seq_like(rng, n)
    yield "random", rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    yield "zeros", bytes(n)
    yield "text", (b"chr1\t12345\tsvim_asm.DEL.17\tN\t<DEL>\t.\tPASS\tSVTYPE=DEL;END=12400;SVLEN=-55\tGT\t1/1\n" * (n // 80 + 1))[:n]
    yield "cigar", np.repeat(rng.integers(0, 1 << 20, max(n // 64, 1), dtype=np.uint32), 16).tobytes()[:n]
    yield "skewed", rng.choice(np.arange(256, dtype=np.uint8), n, p=np.r_[0.7, np.full(255, 0.3 / 255)]).tobytes()
    periodic = bytearray()
    while len(periodic) < n:
        unit = bytes(rng.integers(0, 256, int(rng.integers(1, 12)), dtype=np.uint8))
        periodic += unit * int(rng.integers(1, 400))
    yield "periodic", bytes(periodic[:n])


STRATEGIES = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]


def test_known_streams():
    assert inflate(deflate(b""), 0) == (0, b"")
    assert inflate(deflate(b"a"), 1) == (0, b"a")
    assert inflate(b"\x03\x00", 0) == (0, b"")                       # empty fixed-code block
    assert inflate(b"\x01\x00\x00\xff\xff", 0) == (0, b"")           # empty stored block
    assert inflate(b"\x01\x03\x00\xfc\xffabc", 3) == (0, b"abc")
    assert inflate(b"\x01\x03\x00\xfc\xfeabc", 3)[0] != 0             # LEN / NLEN mismatch
    assert inflate(b"\x07", 8)[0] != 0                                # block type 3
    assert inflate(b"", 8)[0] != 0
    assert inflate(deflate(b"abcabcabcabc"), 11)[0] != 0              # more output than the caller allows
    # a distance beyond the start of the output: fixed code, length 3, distance 1, with nothing written yet
    assert inflate(b"\x03\x02\x00", 16)[0] != 0


@pytest.mark.parametrize("size", [0, 1, 2, 319, 320, 321, 1000, 65536])
def test_whole_and_prefix_steps(size):
    rng = np.random.default_rng(size)
    pr = random.Random(size)
    for name, data in kinds(rng, size):
        for level in (0, 1, 6, 9):
            for strategy in STRATEGIES:
                if level != 6 and strategy not in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED):
                    continue
                s = deflate(data, level, strategy, mem=pr.choice([1, 4, 8, 9]))
                rc, got = inflate(s, len(data))
                assert (rc, got) == (0, data), (name, level, strategy)
                stops = sorted(pr.randrange(0, len(data) + 1) for _ in range(pr.randrange(1, 6)))
                rc, got = inflate(s, len(data), stops)
                assert (rc, got) == (0, data), (name, level, strategy, stops)
                if len(data):
                    assert inflate(s, len(data) - 1)[0] != 0


def test_several_blocks_and_flush_points():
    rng = np.random.default_rng(7)
    parts = [d for _, d in kinds(rng, 9000)]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    s = b""
    for i, p in enumerate(parts):
        s += c.compress(p) + c.flush(zlib.Z_FULL_FLUSH if i % 2 else zlib.Z_SYNC_FLUSH)
    s += c.flush()
    data = b"".join(parts)
    assert inflate(s, len(data)) == (0, data)
    assert inflate(s, len(data), [1, 8999, 9000, 9001, 40000]) == (0, data)


def test_prefix_is_what_was_asked_for():
    rng = np.random.default_rng(11)
    data = seq_like(rng, 65536)
    s = deflate(data)
    lib = _lib.load()
    out = (C.c_uint8 * 65536)()
    n = C.c_uint64(0)
    # a stream cut short decodes as far as it goes: the bytes before the cut are the right ones
    rc = lib.svx_inflate_raw(s[: len(s) // 2], len(s) // 2, out, 65536, None, 0, C.byref(n))
    assert rc != 0 and 20000 < n.value < 45000
    assert bytes(out[: n.value - 300]) == data[: n.value - 300]


def zlib_says(stream, cap):
    d = zlib.decompressobj(-15)
    try:
        out = d.decompress(stream, cap + 1)
    except zlib.error:
        return None
    if not d.eof or len(out) > cap:
        return None
    return out


@pytest.mark.parametrize("seed", range(6))
def test_damaged_streams(seed):
    """Bit flips, cuts and garbage: never a crash or an out-of-bounds write (the sanitizer build runs the same
    cases), and whenever this decoder accepts a stream zlib accepts it with the same bytes, and the other way round."""
    rng = np.random.default_rng(1000 + seed)
    pr = random.Random(seed)
    agree_ok = agree_bad = 0
    for name, data in kinds(rng, 3000):
        for strategy in STRATEGIES:
            s = bytearray(deflate(data, 6, strategy))
            for _ in range(60):
                t = bytearray(s)
                how = pr.randrange(4)
                if how == 0:
                    for _ in range(pr.randrange(1, 4)):
                        t[pr.randrange(len(t))] ^= 1 << pr.randrange(8)
                elif how == 1:
                    del t[pr.randrange(len(t)):]
                elif how == 2:
                    k = pr.randrange(len(t))
                    t[k:k + 4] = bytes(pr.randrange(256) for _ in range(4))
                else:
                    t = bytearray(pr.randrange(256) for _ in range(pr.randrange(1, 200)))
                t = bytes(t)
                cap = len(data) + pr.choice([0, 0, 5, 400])
                rc, got = inflate(t, cap)
                want = zlib_says(t, cap)
                if want is None:
                    assert rc != 0, (name, strategy, how, t.hex()[:80])
                    agree_bad += 1
                else:
                    assert (rc, got) == (0, want), (name, strategy, how)
                    agree_ok += 1
    assert agree_bad > 100 and agree_ok > 5


def inflate_pair(sa, cap_a, stop_a, sb, cap_b, stop_b):
    lib = _lib.load()
    oa, ob = (C.c_uint8 * max(cap_a, 1))(), (C.c_uint8 * max(cap_b, 1))()
    na, nb, ra, rb = C.c_uint64(0), C.c_uint64(0), C.c_int(7), C.c_int(7)
    end = 2 ** 64 - 1
    rc = lib.svx_inflate_raw_pair(sa, len(sa), oa, cap_a, end if stop_a is None else stop_a, C.byref(na), C.byref(ra),
                                  sb, len(sb), ob, cap_b, end if stop_b is None else stop_b, C.byref(nb), C.byref(rb))
    assert rc == 0
    return (ra.value, bytes(oa[: na.value])), (rb.value, bytes(ob[: nb.value]))


@pytest.mark.parametrize("seed", range(4))
def test_two_streams_side_by_side_equal_each_alone(seed):
    """svx_inflate_raw_pair (how svx_bam_seq_slices inflates its members): whatever the two streams are — different
    kinds and sizes, one of them damaged, whole or up to a stop — each gets what svx_inflate_raw gives it alone."""
    rng = np.random.default_rng(500 + seed)
    pr = random.Random(seed)
    pool = []
    for size in (0, 3, 400, 5000, 65536):
        for name, data in kinds(rng, size):
            s = deflate(data, pr.choice([1, 6, 9]), pr.choice(STRATEGIES))
            pool.append((s, len(data)))
            if len(s) > 8:
                t = bytearray(s)
                t[pr.randrange(len(t))] ^= 1 << pr.randrange(8)
                pool.append((bytes(t), len(data)))
                pool.append((s[: pr.randrange(len(s))], len(data)))
    n_bad = 0
    for _ in range(150):
        (sa, ca), (sb, cb) = pr.choice(pool), pr.choice(pool)
        for stops in ((None, None), (pr.randrange(ca + 1), pr.randrange(cb + 1)), (None, pr.randrange(cb + 1))):
            got_a, got_b = inflate_pair(sa, ca, stops[0], sb, cb, stops[1])
            for got, s, cap, stop in ((got_a, sa, ca, stops[0]), (got_b, sb, cb, stops[1])):
                lib = _lib.load()
                out = (C.c_uint8 * max(cap, 1))()
                n = C.c_uint64(0)
                if stop is None:
                    rc = lib.svx_inflate_raw(s, len(s), out, cap, None, 0, C.byref(n))
                    assert got == (rc, bytes(out[: n.value]))
                else:  # alone: run to the stop only (svx_inflate_raw would go on to the end afterwards)
                    st = (C.c_uint64 * 1)(stop)
                    rc = lib.svx_inflate_raw(s, len(s), out, cap, st, 1, C.byref(n))
                    whole = bytes(out[: n.value])
                    if got[0] == 0:
                        assert len(got[1]) >= stop and whole[: len(got[1])] == got[1][: len(whole)]
                    else:
                        assert rc != 0
                n_bad += got[0] != 0
    assert n_bad > 20


def test_decoder_under_sanitizers(tmp_path):
    """tests/native/inflate_sanitize.cpp: the same comparison in C++ with AddressSanitizer + UBSan and heap buffers of
    exactly the sizes the decoder is told — whole, resumed and damaged streams, thousands of cases."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "inflate_sanitize")
    cmd = [gxx, "-std=c++17", "-g", "-O2", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(root, "svim_asm_amd", "csrc"), os.path.join(root, "tests", "native", "inflate_sanitize.cpp"),
           "-lz", "-o", exe]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        pytest.skip("sanitizer build not possible here:\n" + res.stdout[-2000:])
    for seed in (1, 2):
        res = subprocess.run([exe, "500", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert res.returncode == 0 and "inflate_sanitize ok" in res.stdout, res.stdout[-3000:]


def _libdeflate():
    try:
        L = C.CDLL("libdeflate.so.0")
    except OSError:
        return None
    L.libdeflate_alloc_compressor.restype = C.c_void_p
    L.libdeflate_alloc_compressor.argtypes = [C.c_int]
    L.libdeflate_deflate_compress.restype = C.c_size_t
    L.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.libdeflate_deflate_compress_bound.restype = C.c_size_t
    L.libdeflate_deflate_compress_bound.argtypes = [C.c_void_p, C.c_size_t]
    L.libdeflate_free_compressor.argtypes = [C.c_void_p]
    return L


@pytest.mark.parametrize("level", [0, 1, 2, 5, 6, 9, 12])
def test_streams_written_by_libdeflate(level):
    """htslib is usually built with libdeflate, whose compressor splits blocks and builds codes differently from
    zlib's (optimal parsing at the high levels, its own choice of stored / fixed / dynamic blocks): its streams decode
    to the same bytes, whole, in prefix steps and side by side — zlib's inflate being the referee."""
    L = _libdeflate()
    if L is None:
        pytest.skip("libdeflate is not installed here")
    rng = np.random.default_rng(level)
    pr = random.Random(level)
    comp = L.libdeflate_alloc_compressor(level)
    assert comp
    streams = []
    try:
        for size in (0, 1, 50, 4000, 65280):
            for name, data in kinds(rng, size):
                bound = L.libdeflate_deflate_compress_bound(comp, len(data))
                buf = C.create_string_buffer(bound + 16)
                n = L.libdeflate_deflate_compress(comp, data, len(data), buf, bound + 16)
                assert n > 0
                s = buf.raw[:n]
                assert zlib.decompress(s, -15) == data
                streams.append((s, data))
                assert inflate(s, len(data)) == (0, data), (name, size)
                stops = sorted(pr.randrange(0, len(data) + 1) for _ in range(3))
                assert inflate(s, len(data), stops) == (0, data), (name, size, stops)
    finally:
        L.libdeflate_free_compressor(comp)
    for _ in range(40):
        (sa, da), (sb, db) = pr.choice(streams), pr.choice(streams)
        assert inflate_pair(sa, len(da), None, sb, len(db), None) == ((0, da), (0, db))
