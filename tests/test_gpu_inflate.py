"""svx_bgzf_inflate_dev (the kernel of the reader's device leg, svx_bam_set_device_inflate): BGZF member payloads inflated and CRC32-checked on the
device, one lane per member, against zlib — what htslib's bgzf_read_block does under every record the reference
reads (SVIM_COLLECT.py:65-68).  Bytes identical for every zlib level / strategy over several kinds of data, stored
and fixed-code members, members of the config-1 golden BAMs; damaged members are flagged, never read past."""
import os
import zlib

import numpy as np
import pytest

from svim_asm_amd import bamio

pytestmark = pytest.mark.gpu


FORMS = {"wave_parse": 1, "lane_parse": 2, "one_pass": 0}


@pytest.fixture(autouse=True, params=list(FORMS))
def inflate_form(request, svx_ctx):
    """Every test of this module runs on the three forms of the device decoder (svx_bgzf_inflate_set_two_pass): the bit
    streams parsed by a wave per member (64 stretches at once, resynchronised) or by a lane per member, matches as tokens,
    then applied and CRC-checked by a wave per member; and the one-launch lane-per-member kernel."""
    was = svx_ctx.lib.svx_bgzf_inflate_set_two_pass(FORMS[request.param])
    yield request.param
    svx_ctx.lib.svx_bgzf_inflate_set_two_pass(was)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "config1")


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return c.compress(data) + c.flush()


def kinds(rng):
    seq = rng.choice(np.frombuffer(bytes([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88, 0xFF]),
                                   np.uint8), size=60000).tobytes()            # 4-bit packed bases, as a SEQ field
    text = (b"chr1\t12345\tsvim_asm.DEL.7\tACGTNNNN\t<DEL>\t.\tPASS\tSVTYPE=DEL;END=12400;SVLEN=-55\tGT\t0/1\n" * 700)[:65000]
    return {"seq": seq, "text": text, "zeros": bytes(65536), "random": rng.integers(0, 256, 65536, dtype=np.uint8).tobytes(),
            "tiny": b"A", "empty": b"", "runs": b"".join(bytes([int(x)]) * int(n) for x, n in zip(rng.integers(0, 256, 400), rng.integers(1, 300, 400)))[:65536]}


def test_every_level_and_strategy_matches_zlib(svx_ctx):
    rng = np.random.default_rng(1)
    payloads, expect = [], []
    for name, data in kinds(rng).items():
        for level in (0, 1, 2, 4, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                for mem in (1, 9):
                    payloads.append(deflate(data, level, strategy, mem))
                    expect.append(data)
    isize = [len(d) for d in expect]
    crc = [zlib.crc32(d) & 0xFFFFFFFF for d in expect]
    status, outs, _ = svx_ctx.bgzf_inflate(payloads, isize, crc)
    assert status.tolist() == [0] * len(payloads)
    assert all(o == e for o, e in zip(outs, expect))


def test_members_of_the_golden_bams(svx_ctx):
    payloads, isize, crc, expect = [], [], [], []
    for name in ("hap1.bam", "hap2.bam"):
        raw = open(os.path.join(GOLD, name), "rb").read()
        for st, ln, isz, *_ in bamio._bgzf_block_spans(raw):
            member_end = st + ln
            payloads.append(raw[st:member_end])
            expect.append(zlib.decompress(raw[st:member_end], -15) if isz else b"")
            isize.append(isz)
            crc.append(zlib.crc32(expect[-1]) & 0xFFFFFFFF)
    assert len(payloads) > 20
    status, outs, _ = svx_ctx.bgzf_inflate(payloads, isize, crc)
    assert status.tolist() == [0] * len(payloads)
    assert all(o == e for o, e in zip(outs, expect))


def test_damaged_members_are_flagged(svx_ctx):
    """Flipped bits, truncated input, wrong trailer values: the status says so (zlib's verdict where zlib can tell,
    the CRC32 where only the checksum can), the neighbours in the same launch are untouched."""
    rng = np.random.default_rng(2)
    data = kinds(rng)
    base = [(deflate(d, lvl), d) for d in (data["seq"], data["text"], data["runs"]) for lvl in (1, 6)]
    payloads, isize, crc, want_ok, expect = [], [], [], [], []
    for p, d in base:  # intact controls
        payloads.append(p); isize.append(len(d)); crc.append(zlib.crc32(d) & 0xFFFFFFFF); want_ok.append(True); expect.append(d)
    for trial in range(200):
        p, d = base[trial % len(base)]
        bad = bytearray(p)
        kind = trial % 4
        good_isize, good_crc = len(d), zlib.crc32(d) & 0xFFFFFFFF
        if kind == 0:
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            bad = bad[:int(rng.integers(1, len(bad)))]
        elif kind == 2:
            good_crc ^= 1 << int(rng.integers(0, 32))
        else:
            good_isize = max(0, good_isize + int(rng.choice([-1, 1, -100, 7])))
        try:
            out = zlib.decompress(bytes(bad), -15)
            same = out == d and len(out) == good_isize and (zlib.crc32(out) & 0xFFFFFFFF) == good_crc
        except zlib.error:
            same = False
        payloads.append(bytes(bad)); isize.append(good_isize); crc.append(good_crc); want_ok.append(same); expect.append(d)
    status, outs, _ = svx_ctx.bgzf_inflate(payloads, isize, crc)
    for k, (st, ok) in enumerate(zip(status.tolist(), want_ok)):
        assert (st == 0) == ok, (k, st, ok)
        if ok:
            assert outs[k] == expect[k]


def test_members_in_slices_and_blocks_that_outlast_their_window(svx_ctx):
    """More members than the token arena holds go out in slices (svx_bgzf_inflate_set_arena: 37 at a time here), members of
    one block each (Z_HUFFMAN_ONLY / Z_RLE at memLevel 9: 64 KiB of symbols in a single block) and of many (memLevel 1:
    a block every 127 symbols' worth of buffer), empty and one-byte members in between."""
    rng = np.random.default_rng(5)
    payloads, expect = [], []
    for k in range(300):
        kind = k % 6
        if kind == 0:
            data = rng.choice(np.frombuffer(b"\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88", np.uint8),
                              size=int(rng.integers(1, 65281))).tobytes()
        elif kind == 1:
            data = bytes(rng.integers(0, 4, int(rng.integers(1, 65281)), dtype=np.uint8) + 65)
        elif kind == 2:
            data = b"" if k % 12 == 2 else b"N"
        elif kind == 3:
            data = (b"ACGT" * 20000)[:int(rng.integers(1000, 65281))]
        elif kind == 4:
            data = rng.integers(0, 256, int(rng.integers(1, 20000)), dtype=np.uint8).tobytes()
        else:
            data = bytes(65280)
        level, strategy, mem = [(1, zlib.Z_DEFAULT_STRATEGY, 8), (6, zlib.Z_DEFAULT_STRATEGY, 8), (9, zlib.Z_HUFFMAN_ONLY, 9),
                                (6, zlib.Z_RLE, 9), (4, zlib.Z_FILTERED, 1), (6, zlib.Z_FIXED, 8)][(k // 6) % 6]
        payloads.append(deflate(data, level, strategy, mem))
        expect.append(data)
    isize = [len(d) for d in expect]
    crc = [zlib.crc32(d) & 0xFFFFFFFF for d in expect]
    was = svx_ctx.lib.svx_bgzf_inflate_set_arena(37)
    try:
        status, outs, _ = svx_ctx.bgzf_inflate(payloads, isize, crc)
    finally:
        svx_ctx.lib.svx_bgzf_inflate_set_arena(was)
    assert status.tolist() == [0] * len(payloads)
    assert all(o == e for o, e in zip(outs, expect))


def test_members_written_by_libdeflate(svx_ctx):
    """htslib is usually built with libdeflate, whose compressor splits blocks and builds codes differently from zlib's
    (optimal parsing at the high levels, its own choice of stored / fixed / dynamic blocks, blocks of any length): members of
    every kind of data at its levels 0 / 1 / 6 / 9 / 12 through the three forms of the device decoder, zlib's inflate being
    the referee."""
    import ctypes as C
    try:
        L = C.CDLL("libdeflate.so.0")
    except OSError:
        pytest.skip("libdeflate is not installed here")
    L.libdeflate_alloc_compressor.restype = C.c_void_p
    L.libdeflate_alloc_compressor.argtypes = [C.c_int]
    L.libdeflate_deflate_compress.restype = C.c_size_t
    L.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.libdeflate_deflate_compress_bound.restype = C.c_size_t
    L.libdeflate_deflate_compress_bound.argtypes = [C.c_void_p, C.c_size_t]
    L.libdeflate_free_compressor.argtypes = [C.c_void_p]
    rng = np.random.default_rng(12)
    payloads, expect = [], []
    for level in (0, 1, 6, 9, 12):
        comp = L.libdeflate_alloc_compressor(level)
        assert comp
        try:
            for rep in range(3):
                for name, data in kinds(rng).items():
                    data = data[:65280]
                    bound = L.libdeflate_deflate_compress_bound(comp, len(data))
                    buf = C.create_string_buffer(bound + 16)
                    n = L.libdeflate_deflate_compress(comp, data, len(data), buf, bound + 16)
                    assert n > 0
                    stream = buf.raw[:n]
                    if len(stream) > 65000:  # (no BGZF member holds that; libdeflate's stored form of random bytes)
                        continue
                    assert zlib.decompress(stream, -15) == data
                    payloads.append(stream)
                    expect.append(data)
        finally:
            L.libdeflate_free_compressor(comp)
    assert len(payloads) > 60
    isize = [len(d) for d in expect]
    crc = [zlib.crc32(d) & 0xFFFFFFFF for d in expect]
    status, outs, _ = svx_ctx.bgzf_inflate(payloads, isize, crc)
    assert status.tolist() == [0] * len(payloads)
    assert all(o == e for o, e in zip(outs, expect))
