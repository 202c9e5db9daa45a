"""BGZF / BAM reader and writer for the host side of the hot path.

pysam/htslib are not available in the target image, so the subset of their behaviour the
reference relies on (call sites svim-asm:63-90, SVIM_COLLECT.py:11-76,
SVIM_intra.py:35-42, SVIM_inter.py:68-120) is provided here from the SAM/BAM specification:
BGZF multi-member gzip, BAM header / reference dictionary, alignment records, aux tags,
4-bit sequence decoding.  The reader is columnar: `AlignmentFile.batch()` hands the GPU path
one flattened BAM-native CIGAR array (`len << 4 | op`) for the whole file with no per-op
Python work; record objects are thin views created on demand.
"""
import ctypes as C
import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_SEQ_LUT = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
_SEQ_PAIR_LUT = np.empty((256, 2), dtype=np.uint8)
for _b in range(256):
    _SEQ_PAIR_LUT[_b, 0] = _SEQ_LUT[_b >> 4]
    _SEQ_PAIR_LUT[_b, 1] = _SEQ_LUT[_b & 15]
_SEQ_ENC = np.zeros(256, dtype=np.uint8) + 15
for _i, _c in enumerate(b"=ACMGRSVTWYHKDBN"):
    _SEQ_ENC[_c] = _i
    _SEQ_ENC[ord(chr(_c).lower())] = _i
CIGAR_OPS = "MIDNSHP=XB"
_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


# ------------------------------------------------------------------------------ BGZF
def _bgzf_block_spans(raw):
    """(payload_start, payload_len, isize, member_start) of every BGZF member in `raw` (SAM spec §4.1)."""
    spans = []
    p, n = 0, len(raw)
    while p + 18 <= n:
        if raw[p] != 0x1F or raw[p + 1] != 0x8B:
            raise ValueError("not a BGZF/gzip member at offset %d" % p)
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        q, end_x, bsize = p + 12, p + 12 + xlen, None
        while q + 4 <= end_x:
            si1, si2, slen = raw[q], raw[q + 1], struct.unpack_from("<H", raw, q + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", raw, q + 4)[0] + 1
            q += 4 + slen
        if bsize is None:
            raise ValueError("gzip member without BGZF BC subfield at offset %d" % p)
        isize = struct.unpack_from("<I", raw, p + bsize - 4)[0]
        spans.append((end_x, bsize - xlen - 20, isize, p))
        p += bsize
    return spans


def bgzf_decompress(path, threads=None):
    """Inflate a whole BGZF file; blocks are independent so they are inflated on a thread pool
    (zlib releases the GIL)."""
    with open(path, "rb") as fh:
        raw = fh.read()
    spans = _bgzf_block_spans(raw)
    out = bytearray(sum(s[2] for s in spans))
    offs = np.concatenate(([0], np.cumsum([s[2] for s in spans]))).astype(np.int64)
    view = memoryview(raw)

    def work(lo, hi):
        for i in range(lo, hi):
            st, ln, isz = spans[i][:3]
            if isz:
                out[offs[i]:offs[i] + isz] = zlib.decompress(view[st:st + ln], -15)

    n = len(spans)
    threads = threads or min(8, os.cpu_count() or 1)
    if n < 64 or threads <= 1:
        work(0, n)
    else:
        step = (n + threads * 4 - 1) // (threads * 4)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda lo: work(lo, min(n, lo + step)), range(0, n, step)))
    return out  # bytearray: no second copy of a multi-GB buffer


class BgzfLazy(object):
    """Random access into the uncompressed stream of a BGZF file, inflating blocks on demand.

    Genome-genome alignment BAMs hold a few thousand records whose SEQ/QUAL fields are hundreds
    of kilobases to megabases long: more than 90 % of the BGZF blocks lie wholly inside those
    fields and are never needed (COLLECT reads headers, names, CIGARs and tags, and the bases of
    the inserted alleles only).  The ISIZE trailer of every block gives its uncompressed length
    without inflating it, so record offsets can be followed through skipped blocks."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self._raw = fh.read()
        spans = _bgzf_block_spans(self._raw)
        self._start = np.array([s[0] for s in spans], dtype=np.int64)
        self._hdr_start = np.array([s[3] for s in spans], dtype=np.int64)
        self._clen = np.array([s[1] for s in spans], dtype=np.int64)
        isize = np.array([s[2] for s in spans], dtype=np.int64)
        self._uoff = np.concatenate(([0], np.cumsum(isize))).astype(np.int64)
        self.size = int(self._uoff[-1])
        self._cache = {}
        self._view = memoryview(self._raw)
        self.blocks_inflated = 0

    def _block(self, i):
        b = self._cache.get(i)
        if b is None:
            st, ln = int(self._start[i]), int(self._clen[i])
            b = zlib.decompress(self._view[st:st + ln], -15) if self._uoff[i + 1] > self._uoff[i] else b""
            self._cache[i] = b
            self.blocks_inflated += 1
        return b

    def read(self, off, n):
        """Bytes [off, off+n) of the uncompressed stream."""
        if n <= 0:
            return b""
        end = min(off + n, self.size)
        i = int(np.searchsorted(self._uoff, off, side="right")) - 1
        parts = []
        while off < end:
            b = self._block(i)
            lo = off - int(self._uoff[i])
            take = min(len(b) - lo, end - off)
            parts.append(b[lo:lo + take])
            off += take
            i += 1
        return parts[0] if len(parts) == 1 else b"".join(parts)

    def prefetch(self, ranges, threads=None):
        """Inflate, in parallel, every block touched by the (offset, length) ranges."""
        need = set()
        for off, n in ranges:
            if n <= 0:
                continue
            i0 = int(np.searchsorted(self._uoff, off, side="right")) - 1
            i1 = int(np.searchsorted(self._uoff, min(off + n, self.size) - 1, side="right")) - 1
            need.update(i for i in range(i0, i1 + 1) if i not in self._cache)
        need = sorted(need)
        threads = threads or min(16, os.cpu_count() or 1)
        if len(need) < 32 or threads <= 1:
            for i in need:
                self._block(i)
            return
        def work(chunk):
            return [(i, zlib.decompress(self._view[int(self._start[i]):int(self._start[i] + self._clen[i])], -15))
                    for i in chunk]
        step = (len(need) + threads * 4 - 1) // (threads * 4)
        with ThreadPoolExecutor(threads) as ex:
            for res in ex.map(work, [need[k:k + step] for k in range(0, len(need), step)]):
                for i, b in res:
                    self._cache[i] = b
                    self.blocks_inflated += 1

    def drop_cache(self):
        self._cache.clear()

    def virtual_offset(self, off):
        """BGZF virtual offset (coffset << 16 | uoffset) of uncompressed offset `off`; positions at
        the end of a member are expressed as the start of the next one (what htslib writes)."""
        i = int(np.searchsorted(self._uoff, off, side="right")) - 1
        i = min(i, len(self._start) - 1)
        hdr = int(self._hdr_start[i])
        return (hdr << 16) | (off - int(self._uoff[i]))


_LIBDEFLATE = [None]


def _libdeflate_compress(chunk, level):
    """Raw DEFLATE stream of `chunk` by libdeflate (levels 100 + n of the writers; what an htslib built with libdeflate
    puts into its BGZF members), or None where the library is not installed."""
    import ctypes as C
    import threading
    if _LIBDEFLATE[0] is None:
        try:
            lib = C.CDLL("libdeflate.so.0")
            lib.libdeflate_alloc_compressor.restype = C.c_void_p
            lib.libdeflate_alloc_compressor.argtypes = [C.c_int]
            lib.libdeflate_deflate_compress.restype = C.c_size_t
            lib.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
            _LIBDEFLATE[0] = (lib, threading.local())
        except OSError:
            _LIBDEFLATE[0] = False
    if not _LIBDEFLATE[0]:
        return None
    lib, tls = _LIBDEFLATE[0]
    comps = tls.__dict__.setdefault("comps", {})
    if level not in comps:
        comps[level] = lib.libdeflate_alloc_compressor(level)   # one per thread and level, kept
    data = bytes(chunk)
    buf = C.create_string_buffer(len(data) + 1024)
    n = lib.libdeflate_deflate_compress(comps[level], data, len(data), buf, len(data) + 1024)
    return buf.raw[:n] if n else None


def _bgzf_member(chunk, level):
    comp = None
    if level >= 100:  # 100 + n: libdeflate at level n (falls back to zlib at min(n, 9) where it is not installed)
        comp = _libdeflate_compress(chunk, level - 100)
        level = min(level - 100, 9)
    if comp is None:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
    return b"".join((struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(comp) + 25), comp,
                     struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))))


def bgzf_compress(data, level=1, threads=None):
    """BGZF-compress `data` (64 KiB minus slack per block) + EOF marker.  Members are independent, so
    they are deflated on a thread pool (zlib releases the GIL); the bytes do not depend on the pool."""
    mv = memoryview(data)
    starts = range(0, len(data), 0xFF00)
    threads = threads or min(32, os.cpu_count() or 1)
    if len(starts) < 64 or threads <= 1:
        parts = [_bgzf_member(mv[p:p + 0xFF00], level) for p in starts]
    else:
        step = 64
        def work(lo):
            return b"".join(_bgzf_member(mv[p:p + 0xFF00], level) for p in starts[lo:lo + step])
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(work, range(0, len(starts), step)))
    return b"".join(parts) + _BGZF_EOF


# ------------------------------------------------------------------------------ header
class _Header(dict):
    """Dict view of the @-lines (enough for header["HD"]["SO"], svim-asm:65)."""


def _parse_header_text(text):
    hdr = _Header()
    for line in text.split("\n"):
        if not line.startswith("@") or len(line) < 3:
            continue
        fields = line.rstrip("\r").split("\t")
        tag = fields[0][1:]
        if tag == "CO":
            hdr.setdefault("CO", []).append("\t".join(fields[1:]))
            continue
        d = {}
        for f in fields[1:]:
            if len(f) >= 3 and f[2] == ":":
                d[f[:2]] = f[3:]
        if tag == "HD":
            hdr["HD"] = d
        else:
            hdr.setdefault(tag, []).append(d)
    return hdr


# ------------------------------------------------------------------------------ records
class AlignedRecord(object):
    """One BAM record (or an SA-derived pseudo record) with pysam-compatible attribute names."""
    __slots__ = ("query_name", "flag", "reference_id", "reference_start", "mapping_quality",
                 "cigar_words", "_seq_packed", "_l_seq", "_seq_str", "_tags_raw", "_tags", "index", "_sa",
                 "_sa_absent", "_seq_fetch")

    def __init__(self):
        self.query_name = None
        self.flag = 0
        self.reference_id = -1
        self.reference_start = -1
        self.mapping_quality = 0
        self.cigar_words = np.zeros(0, np.uint32)
        self._seq_packed = None
        self._l_seq = 0
        self._seq_str = None
        self._tags_raw = None
        self._tags = None
        self.index = -1
        self._sa = None          # SA:Z string located by the native reader (None: parse the aux bytes)
        self._sa_absent = False  # the native reader looked for an SA tag and found none
        self._seq_fetch = None   # callable (a, b) -> str decoding bases straight from the BGZF stream

    is_unmapped = property(lambda s: bool(s.flag & 0x4))
    is_secondary = property(lambda s: bool(s.flag & 0x100))
    is_supplementary = property(lambda s: bool(s.flag & 0x800))
    is_reverse = property(lambda s: bool(s.flag & 0x10))

    @property
    def cigartuples(self):
        w = self.cigar_words
        if len(w) == 0:
            return None
        return list(zip((w & 15).tolist(), (w >> 4).tolist()))

    @property
    def cigarstring(self):
        w = self.cigar_words
        if len(w) == 0:
            return None
        return "".join("%d%s" % (l, CIGAR_OPS[o]) for o, l in zip((w & 15).tolist(), (w >> 4).tolist()))

    def _sum(self, opmask):
        w = self.cigar_words
        return int(((w >> 4) * ((opmask >> (w & 15)) & 1)).sum()) if len(w) else 0

    def get_cigar_stats(self):
        base, cnt = [0] * 11, [0] * 11
        w = self.cigar_words
        for o, l in zip((w & 15).tolist(), (w >> 4).tolist()):
            if o < 10:
                base[o] += l
                cnt[o] += 1
        return base, cnt

    @property
    def reference_end(self):
        if self.is_unmapped or len(self.cigar_words) == 0:
            return None
        rlen = self._sum(0x18D)  # M D N = X
        return self.reference_start + (rlen if rlen else 1)  # htslib bam_endpos

    @property
    def query_alignment_start(self):
        """Σ leading soft clips, skipping hard clips (pysam getQueryStart)."""
        w, s = self.cigar_words, 0
        for k in range(len(w)):
            o = int(w[k]) & 15
            if o == 5:
                continue
            if o == 4:
                s += int(w[k]) >> 4
            else:
                break
        return s

    @property
    def query_alignment_end(self):
        """pysam getQueryEnd: with a stored sequence, l_seq minus the trailing soft clips; without
        one (SA-derived segments) leading clips + Σ{M,I,=,X}."""
        w = self.cigar_words
        if self._l_seq == 0:
            end = 0
            for o, l in zip((w & 15).tolist(), (w >> 4).tolist()):
                if o in (0, 1, 7, 8) or (o == 4 and end == 0):
                    end += l
            return end
        end = self._l_seq
        for k in range(len(w) - 1, 0, -1):
            o = int(w[k]) & 15
            if o == 5:
                continue
            if o == 4:
                end -= int(w[k]) >> 4
            else:
                break
        return end

    def infer_read_length(self):
        if len(self.cigar_words) == 0:
            return None
        return self._sum(0x1B3)  # M I S H = X

    # ---- sequence (BAM orientation, 4-bit packed)
    def seq_slice(self, a, b):
        """query_sequence[a:b] without decoding the whole (contig-sized) read."""
        n = self._l_seq
        a = max(0, min(n, a))
        b = max(a, min(n, b))
        if b <= a:
            return ""
        if self._seq_str is not None:
            return self._seq_str[a:b]
        if self._seq_fetch is not None:
            return self._seq_fetch(a, b)
        by = self._seq_packed[a >> 1:(b + 1) >> 1]
        dec = _SEQ_PAIR_LUT[by].reshape(-1)
        off = a & 1
        return dec[off:off + (b - a)].tobytes().decode("ascii")

    @property
    def query_sequence(self):
        if self._seq_str is None:
            self._seq_str = self.seq_slice(0, self._l_seq) if self._l_seq else None
        return self._seq_str

    # ---- aux tags
    def _parse_tags(self):
        if self._tags is not None:
            return self._tags
        tags = {}
        raw = self._tags_raw
        q, end = 0, len(raw) if raw is not None else 0
        while q + 3 <= end:
            tag = bytes(raw[q:q + 2]).decode()
            typ = chr(raw[q + 2])
            q += 3
            if typ in "ZH":
                e = q
                while raw[e] != 0:
                    e += 1
                tags[tag] = bytes(raw[q:e]).decode()
                q = e + 1
            elif typ == "A":
                tags[tag] = chr(raw[q])
                q += 1
            elif typ in _AUX_FMT:
                fmt, size = _AUX_FMT[typ]
                tags[tag] = struct.unpack_from(fmt, raw, q)[0]
                q += size
            elif typ == "B":
                sub = chr(raw[q])
                cnt = struct.unpack_from("<i", raw, q + 1)[0]
                fmt, size = _AUX_FMT[sub]
                tags[tag] = list(struct.unpack_from("<%d%s" % (cnt, fmt[1]), raw, q + 5))
                q += 5 + cnt * size
            else:
                raise ValueError("unknown aux type %r" % typ)
        self._tags = tags
        return tags

    def get_tag(self, name):
        if name == "SA":
            if self._sa is not None:
                return self._sa
            if self._sa_absent:
                raise KeyError("tag 'SA' not present")
        tags = self._parse_tags()
        if name not in tags:
            raise KeyError("tag '%s' not present" % name)
        return tags[name]

    def has_tag(self, name):
        return name in self._parse_tags()


class _LazySeq(object):
    """The packed 4-bit SEQ field of one record, sliced straight from the lazily inflated stream."""
    __slots__ = ("_z", "_off", "_n")

    def __init__(self, z, off, n):
        self._z, self._off, self._n = z, off, n

    def __len__(self):
        return self._n

    def __getitem__(self, sl):
        a, b, _ = sl.indices(self._n)
        return np.frombuffer(self._z.read(self._off + a, b - a), dtype=np.uint8)


_AUX_FMT = {"c": ("<b", 1), "C": ("<B", 1), "s": ("<h", 2), "S": ("<H", 2), "i": ("<i", 4),
            "I": ("<I", 4), "f": ("<f", 4), "d": ("<d", 8)}


def _restore_long_cigar(words, l_seq, tid, pos, aux):
    """SAM spec §4.2.2 / htslib bam_tag2cigar: a stored CIGAR that starts with a soft clip as long
    as the read is a placeholder when a CG:B,I array is present; returns (real words, aux without CG)
    or None."""
    if len(words) == 0 or tid < 0 or pos < 0:
        return None
    w0 = int(words[0])
    if (w0 & 15) != 4 or (w0 >> 4) != l_seq:
        return None
    q, end = 0, len(aux)
    while q + 3 <= end:
        typ = aux[q + 2]
        start = q
        q += 3
        if typ in b"ZH":
            q = aux.index(b"\x00", q) + 1
        elif typ == 66:  # 'B'
            sub, cnt = chr(aux[q]), struct.unpack_from("<I", aux, q + 1)[0]
            size = _AUX_FMT[sub][1]
            if aux[start:start + 2] == b"CG":
                if sub not in "Ii" or cnt < len(words) or cnt >= (1 << 29):
                    return None
                real = np.frombuffer(aux, dtype="<u4", count=cnt, offset=q + 5)
                return real, aux[:start] + aux[q + 5 + cnt * size:]
            q += 5 + cnt * size
        elif typ == 65:  # 'A'
            q += 1
        else:
            q += _AUX_FMT[chr(typ)][1]
        if aux[start:start + 2] == b"CG":
            return None  # a CG tag that is not an integer array
    return None


def ingest_threads(n_files=1):
    """Threads one reader gets when `n_files` BAMs are walked / decoded at the same time (both haplotypes of a
    diploid sample): a quarter of the hardware threads in total, at most 64 and at least 8 per file.  The inflate
    work scales to 16-32 threads per file on the GPU hosts (shared nodes: more threads than free cores only
    oversubscribe — tools/slice_probe.py, profiles/README.md).  The policy of a long-lived process that handles one
    sample at a time: under a CPU quota its bursts run on a fresh period's budget (in-process BAM -> VCF of the full-size
    sample 0.16-0.18 s; with quota_threads below 0.20-0.21).  A fresh command and a process that works continuously
    size their readers by the quota instead: quota_threads."""
    asked = os.environ.get("SVX_INGEST_THREADS")  # (experiments: tools/cli_timeline.py)
    if asked and asked.isdigit() and int(asked) > 0:
        return int(asked)
    hw = os.cpu_count() or 1
    return int(max(1, min(64, max(8, hw // (4 * max(1, n_files))), hw)))


def quota_threads(n_files=1, processes=1):
    """Threads per reader for a process whose start-up or steady work shares the CPU-quota periods with its readers
    (`svim-asm`, the ranks of a sharded run, svim-asm-cohort): under a quota (cgroup cpu.max below the hardware threads:
    the GPU pool grants 16 CPUs of 256) the readers of all `processes` together get as many threads as the quota has CPUs
    — more only spend a period's budget in a fraction of the period, the whole process then stands still for the rest
    of it, and a process that calls exit() while it is throttled is gone one period later.  Measured on the full-size
    sample, nine fresh `svim-asm diploid` processes each (profiles/r06_cli_timeline.txt): 32 threads per reader
    0.58-0.60 s, one or two periods throttled in every run, 0.10 s from os._exit to gone; 8 per reader 0.46-0.49 s, no
    period throttled, 3 ms to gone.  Without a quota: ingest_threads."""
    asked = os.environ.get("SVX_INGEST_THREADS")
    if asked and asked.isdigit() and int(asked) > 0:
        return int(asked)
    hw, quota = os.cpu_count() or 1, host_cpus()
    if quota < hw:
        return int(max(2, round(quota / float(max(1, n_files) * max(1, processes)))))
    return ingest_threads(n_files)


def host_cpus():
    """CPUs' worth of time this process gets: the hardware threads, or the cgroup's quota when that is smaller
    (cpu.max: 16 of a 256-thread host on the GPU pool)."""
    hw = float(os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            hw = min(hw, float(quota) / float(period))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if quota > 0:
                hw = min(hw, quota / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))
        except (OSError, ValueError):
            pass
    return hw


def default_device_inflate_percent():
    """Share of a sequence-slice call the device inflates (svx_bam_set_device_inflate) unless SVX_BAM_DEVICE_INFLATE says
    otherwise.  The device decodes a full-size sample's 14 k sequence members in ~17 ms (a wave per member,
    csrc/svx_inflate.hip) behind their staging; the host's threads take 45 ms for them when every thread has a core, but
    1.4 CPU-seconds of a run whose wall-clock IS its CPU-seconds over the CPUs it may use when those are few (the pool's
    16-CPU quota).  Measured there on the full-size sample (profiles/r06_wave_e2e.txt, r06_wave_cli_shares.txt): in one
    process 0.135-0.146 s at a share of 50 %, 0.132 at 60, 0.137 at 100 with 2.7 / 2.5 / 1.9 CPU-seconds (0.20-0.24 s and
    3.3 without the device); as a fresh command the same wall-clock at every share with 3.4 / 2.6 / 2.0 CPU-seconds at
    0 / 50 / 100.  So: the whole call with at most 24 CPUs' worth of time, none above (unmeasured there: no such host)."""
    asked = env_device_inflate_percent()
    if asked is not None:
        return asked
    return 100 if host_cpus() <= 24 else 0


def env_device_inflate_percent():
    """SVX_BAM_DEVICE_INFLATE as a percentage 0..100, or None when it is unset, empty or not a number (said once on
    stderr: a typo in the variable must not take down every entry point that imports this module)."""
    env = os.environ.get("SVX_BAM_DEVICE_INFLATE")
    if env in (None, ""):
        return None
    try:
        return max(0, min(100, int(env)))
    except ValueError:
        if env not in _ENV_WARNED:
            _ENV_WARNED.add(env)
            import warnings
            warnings.warn("SVX_BAM_DEVICE_INFLATE=%r is not a percentage (0..100): ignored, the default applies" % env)
        return None


_ENV_WARNED = set()


class _BamColumns(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("l_seq", C.c_void_p),
                ("ref_len", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p), ("cigar_off", C.c_void_p),
                ("cigar", C.c_void_p), ("name_off", C.c_void_p), ("names", C.c_void_p), ("aux_off", C.c_void_p),
                ("aux", C.c_void_p), ("sa_off", C.c_void_p), ("sa_len", C.c_void_p), ("voffset", C.c_void_p),
                ("blocks_inflated", C.c_uint64), ("blocks_spanned", C.c_uint64), ("cigar_pinned", C.c_int),
                ("n_threads", C.c_int)]


def _view(addr, n, dtype):
    """numpy view over `n` items of native memory at `addr` (valid until the next load / close)."""
    dtype = np.dtype(dtype)
    if n == 0 or not addr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class AlignmentFile(object):
    """Coordinate-sorted BAM opened for per-contig streaming (pysam.AlignmentFile surface the
    reference uses).  Records are indexed column-wise by the native reader
    of libsvx.so (include/svx_bam.h): threads walk the file from the record boundaries a `.bai`
    provides and inflate only the BGZF members holding record headers, names, CIGARs and tags.
    `load(contigs)` restricts the walk to the contigs a rank owns; anything that touches the
    records loads the whole file on demand.  reader="python" selects the pure-Python walker
    (kept as the differential reference of the native one; also via SVX_BAM_READER=python).
    verify (default True, or what SVX_BAM_VERIFY says): every touched BGZF member is inflated completely and its CRC32
    checked (what htslib does under the reference); verify=False stops at the last byte needed and checks a member
    only when it happens to be inflated to its end (svx_bam_set_verify, include/svx_bam.h)."""

    def __init__(self, path, mode="rb", threads=None, reader=None, device=None, verify=None):
        self.filename = path
        self._reader = reader or os.environ.get("SVX_BAM_READER", "native")
        self._loaded = None   # None: nothing; "all" or a tuple of tids
        self._h = None
        self._z = None
        self._pin_device = device
        if self._reader == "native":
            from svim_asm_amd import _lib
            self._lib = lib = _lib.load()
            h, err = C.c_void_p(), C.create_string_buffer(512)
            rc = lib.svx_bam_open(os.fsencode(path), int(threads or 0), C.byref(h), err, len(err))
            if rc != 0:
                msg = err.value.decode(errors="replace")
                if not os.path.exists(path):
                    raise FileNotFoundError(msg)
                raise ValueError(msg)
            self._h = h
            if verify is not None:  # None: the library's default (SVX_BAM_VERIFY); see include/svx_bam.h
                lib.svx_bam_set_verify(h, 1 if verify else 0)
            text, l_text, n_ref = C.c_char_p(), C.c_uint64(), C.c_int32()
            tp = C.c_void_p()
            lib.svx_bam_header(h, C.byref(tp), C.byref(l_text), C.byref(n_ref))
            self.text = C.string_at(tp, l_text.value).decode() if l_text.value else ""
            names, lens = [], []
            for tid in range(n_ref.value):
                nm, ln = C.c_char_p(), C.c_int32()
                lib.svx_bam_reference(h, tid, C.byref(nm), C.byref(ln))
                names.append(nm.value.decode())
                lens.append(ln.value)
        else:
            z = self._z = BgzfLazy(path)
            head = z.read(0, 12)
            if head[:4] != b"BAM\x01":
                raise ValueError("%s is not a BAM file" % path)
            l_text = struct.unpack_from("<i", head, 4)[0]
            self.text = z.read(8, l_text).split(b"\x00")[0].decode()
            p = 8 + l_text
            n_ref = struct.unpack_from("<i", z.read(p, 4), 0)[0]
            p += 4
            names, lens = [], []
            for _ in range(n_ref):
                l_name = struct.unpack_from("<i", z.read(p, 4), 0)[0]
                nb = z.read(p + 4, l_name + 4)
                names.append(nb[:l_name - 1].decode())
                lens.append(struct.unpack_from("<i", nb, l_name)[0])
                p += 8 + l_name
            self._rec_start = p
        self.header = _parse_header_text(self.text)
        self.references = tuple(names)
        self.lengths = tuple(lens)
        self._tid = {n: i for i, n in enumerate(names)}

    # ---------------------------------------------------------------- loading
    def index_state(self):
        """0 no index file, 1 usable .bai (parallel / per-contig walks), 2 present but unusable."""
        if self._h is not None:
            return int(self._lib.svx_bam_index_state(self._h))
        base = self.filename
        stem = os.path.splitext(base)[0]
        return 2 if any(os.path.exists(p) for p in (base + ".bai", stem + ".bai", base + ".csi", stem + ".csi")) else 0

    def contig_spans(self):
        """Compressed bytes per contig from the index (None without a usable one)."""
        if self._h is None or self.index_state() != 1:
            return None
        span = np.zeros(len(self.references), dtype=np.uint64)
        if self._lib.svx_bam_contig_spans(self._h, span.ctypes.data) != 0:
            return None
        return span.astype(np.int64)

    # share of a sequence-slice call's members that the pinned device inflates and verifies beside the reader's threads
    # (svx_bam_set_device_inflate); None = default_device_inflate_percent() above, asked when the file is loaded — the
    # environment and the CPU quota as they are THEN, not as they were when the module was imported
    device_inflate_percent = None

    def effective_device_inflate_percent(self):
        pct = self.device_inflate_percent
        return default_device_inflate_percent() if pct is None else int(pct)
    # the share goes to the device only when it holds that many members (svx_bam.h; SVX_BAM_DEVICE_INFLATE_MIN for experiments)
    device_inflate_min_members = int(os.environ.get("SVX_BAM_DEVICE_INFLATE_MIN") or 500)
    device_inflate_wait_ms = 0         # how long a call waits for one of the device's inflate lanes (svx_bam.h)
    # svx_bam_set_defer_verify: the record walks leave the check of the members they touch to the device leg of the next
    # sequence-slice call (a third of the walks' CPU time instead of all of it).  Whoever sets this calls verify_pending()
    # before trusting the records when no sequence_slices_raw call follows (SVIM_COLLECT.collect_tables does).
    defer_verify = False

    @property
    def device_members(self):
        return int(self._lib.svx_bam_device_members(self._h)) if self._h is not None else 0

    def set_device(self, device):
        """HIP device whose context page-locks the CIGAR pool of later loads (None: pageable)."""
        self._pin_device = device

    def device_pool(self, wait=False):
        """(device address, hipEvent_t) of the loaded CIGAR pool's copy in HBM, or None when there is none
        (svx_bam_device_pool: the reader uploads the page-locked pool while it assembles it).  With `wait`, blocks
        until the copy is complete and returns (address, None, microseconds waited)."""
        if self._h is None or self._loaded is None:
            return None
        d, n, ev = C.c_void_p(), C.c_uint64(), C.c_void_p()
        if self._lib.svx_bam_device_pool(self._h, C.byref(d), C.byref(n), C.byref(ev)) != 0 or not d.value:
            return None
        if int(n.value) != len(self._c_cigar):
            return None
        if wait:
            us = C.c_double()
            if self._lib.svx_bam_device_pool_wait(self._h, C.byref(us)) != 0:
                raise ValueError("%s: %s" % (self.filename, self._lib.svx_bam_last_error(self._h).decode(errors="replace")))
            return d.value, None, float(us.value)
        return d.value, ev.value

    def load(self, contigs=None):
        """Index the records of `contigs` (names or tids; None = the whole file)."""
        if contigs is None:
            want = "all"
            tids = None
        else:
            tids = sorted(set(self._tid[c] if isinstance(c, str) else int(c) for c in contigs))
            want = tuple(tids)
        if self._loaded == want:
            return self
        if self._h is not None:
            self._lib.svx_bam_set_pinned_device(self._h, -1 if self._pin_device is None else int(self._pin_device))
            self._lib.svx_bam_set_device_inflate(self._h, 0 if self._pin_device is None else self.effective_device_inflate_percent())
            self._lib.svx_bam_set_device_inflate_min(self._h, int(self.device_inflate_min_members))
            self._lib.svx_bam_set_device_inflate_wait(self._h, int(self.device_inflate_wait_ms))
            self._lib.svx_bam_set_defer_verify(self._h, 1 if self.defer_verify else 0)
            if tids is None:
                rc = self._lib.svx_bam_load(self._h, None, 0)
            else:
                arr = np.asarray(tids, dtype=np.int32)
                rc = self._lib.svx_bam_load(self._h, arr.ctypes.data, len(arr))
            if rc != 0:
                raise ValueError("%s: %s" % (self.filename, self._lib.svx_bam_last_error(self._h).decode(errors="replace")))
            self._bind_native_columns()
        else:
            self._index_records_python(None if tids is None else set(tids))
        self._loaded = want
        return self

    def verify_pending(self):
        """Check (on the reader's threads) the members the record walks took bytes from and no device leg has checked yet
        (defer_verify); ValueError for a damaged one, as load() itself raises without the deferral.  No-op otherwise."""
        if self._h is not None and self._lib.svx_bam_verify_pending(self._h) != 0:
            raise ValueError("%s: %s" % (self.filename, self._lib.svx_bam_last_error(self._h).decode(errors="replace")))

    @property
    def pending_members(self):
        return int(self._lib.svx_bam_pending_members(self._h)) if self._h is not None else 0

    def _ensure(self):
        if self._loaded is None:
            self.load(None)

    def _bind_native_columns(self):
        c = _BamColumns()
        self._lib.svx_bam_get_columns(self._h, C.byref(c))
        n = int(c.n_records)
        self.n_records = n
        cig_off = _view(c.cigar_off, n + 1, np.uint64).astype(np.int64)
        self._c_cols = {"tid": _view(c.tid, n, np.int32).astype(np.int64), "pos": _view(c.pos, n, np.int32).astype(np.int64),
                       "mapq": _view(c.mapq, n, np.uint8).astype(np.int64), "flag": _view(c.flag, n, np.uint16).astype(np.int64),
                       "n_cig": np.diff(cig_off), "l_seq": _view(c.l_seq, n, np.int32).astype(np.int64),
                       "ref_len": _view(c.ref_len, n, np.int32).astype(np.int64),
                       "voffset": _view(c.voffset, n, np.uint64).copy()}
        self._c_cig_off = cig_off
        self._c_cigar = _view(c.cigar, int(cig_off[-1]) if n else 0, np.uint32)  # zero-copy (page-locked pool)
        self._name_off = _view(c.name_off, n + 1, np.uint64).astype(np.int64)
        self._names_pool = C.string_at(c.names, int(self._name_off[-1])) if n and self._name_off[-1] else b""
        self._aux_off = _view(c.aux_off, n + 1, np.uint64).astype(np.int64)
        self._aux_pool = C.string_at(c.aux, int(self._aux_off[-1])) if n and self._aux_off[-1] else b""
        self._sa_off = _view(c.sa_off, n, np.int64).copy()
        self._sa_len = _view(c.sa_len, n, np.uint32).astype(np.int64)
        self.cigar_pinned = bool(c.cigar_pinned)

    @property
    def blocks_inflated(self):
        if self._h is not None:
            c = _BamColumns()
            self._lib.svx_bam_get_columns(self._h, C.byref(c))
            return int(c.blocks_inflated)
        return self._z.blocks_inflated

    @property
    def blocks_spanned(self):
        if self._h is not None:
            c = _BamColumns()
            self._lib.svx_bam_get_columns(self._h, C.byref(c))
            return int(c.blocks_spanned)
        return len(self._z._start)

    # the column containers load the file on first touch
    @property
    def _cols(self):
        self._ensure()
        return self._c_cols

    @property
    def _cigar(self):
        self._ensure()
        return self._c_cigar

    @property
    def _cig_off(self):
        self._ensure()
        return self._c_cig_off

    # ---- pure-Python walker: one sequential pass over the record headers; the name, CIGAR words
    # and tag bytes of every record are copied out, SEQ/QUAL are skipped
    def _index_records_python(self, keep_tids):
        z, p, n = self._z, self._rec_start, self._z.size
        cols = {k: [] for k in ("seq_off", "tid", "pos", "mapq", "flag", "n_cig", "l_seq", "ref_len", "voffset")}
        names, tags, cig_parts = [], [], []
        unpack = struct.Struct("<iiiBBHHHi").unpack_from
        while p + 36 <= n:
            bs, tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq = unpack(z.read(p, 36), 0)
            q = p + 36
            if keep_tids is not None and tid not in keep_tids:
                p += 4 + bs
                continue
            body = z.read(q, l_rn + 4 * n_cig)
            words = np.frombuffer(body, dtype="<u4", count=n_cig, offset=l_rn)
            q += l_rn + 4 * n_cig
            tag_off = q + (l_seq + 1) // 2 + l_seq
            aux = z.read(tag_off, p + 4 + bs - tag_off)
            real = _restore_long_cigar(words, l_seq, tid, pos, aux)
            if real is not None:
                words, aux = real
            names.append(body[:l_rn - 1].decode())
            cig_parts.append(words.tobytes())
            tags.append(aux)
            cols["seq_off"].append(q); cols["tid"].append(tid); cols["pos"].append(pos)
            cols["mapq"].append(mapq); cols["flag"].append(flag); cols["n_cig"].append(len(words))
            cols["l_seq"].append(l_seq)
            cols["ref_len"].append(int(((words >> 4) * ((0x18D >> (words & 15)) & 1)).sum()) if len(words) else 0)
            cols["voffset"].append(z.virtual_offset(p))
            p += 4 + bs
        self._c_cols = {k: np.asarray(v, dtype=np.uint64 if k == "voffset" else np.int64) for k, v in cols.items()}
        self.n_records = len(names)
        self._names, self._tags = names, tags
        self._c_cigar = np.frombuffer(b"".join(cig_parts), dtype="<u4") if cig_parts else np.zeros(0, np.uint32)
        self._c_cig_off = np.concatenate(([0], np.cumsum(self._c_cols["n_cig"]))).astype(np.int64) \
            if self.n_records else np.zeros(1, np.int64)
        self.cigar_pinned = False

    def __len__(self):
        self._ensure()
        return self.n_records

    def record(self, i):
        c = self._cols
        r = AlignedRecord()
        r.index = i
        r.reference_id = int(c["tid"][i])
        r.reference_start = int(c["pos"][i])
        r.mapping_quality = int(c["mapq"][i])
        r.flag = int(c["flag"][i])
        r.cigar_words = self._cigar[self._cig_off[i]:self._cig_off[i + 1]]
        r._l_seq = int(c["l_seq"][i])
        if self._h is not None:
            r.query_name = self._names_pool[self._name_off[i]:self._name_off[i + 1]].decode()
            r._tags_raw = self._aux_pool[self._aux_off[i]:self._aux_off[i + 1]]
            so = int(self._sa_off[i])
            if so >= 0:
                r._sa = self._aux_pool[so:so + int(self._sa_len[i])].decode()
            else:
                r._sa_absent = True
            r._seq_fetch = lambda a, b, _i=i: self.sequence_slices([_i], [a], [b])[0]
        else:
            r.query_name = self._names[i]
            r._seq_packed = _LazySeq(self._z, int(c["seq_off"][i]), (r._l_seq + 1) // 2)
            r._tags_raw = self._tags[i]
        return r

    def sequence_slices(self, rec, begin, end):
        """query_sequence[begin:end] of many records at once (list of str), decoded from the BGZF
        stream by the reader's threads; only the members holding those bases are inflated."""
        self._ensure()
        rec = np.ascontiguousarray(rec, dtype=np.uint32)
        n = len(rec)
        if n == 0:
            return []
        l_seq = self._cols["l_seq"][rec.astype(np.int64)]
        a = np.minimum(np.maximum(np.asarray(begin, dtype=np.int64), 0), l_seq)
        b = np.maximum(np.minimum(np.asarray(end, dtype=np.int64), l_seq), a)
        if self._h is None:
            out = []
            for i, x, y in zip(rec.tolist(), a.tolist(), b.tolist()):
                out.append(self.record(i).seq_slice(x, y))
            return out
        buf, off = self._slices_native(rec, a, b)
        text = buf.tobytes().decode("ascii")
        o = off.tolist()
        return [text[o[k]:o[k + 1]] for k in range(n)]

    def _slices_native(self, rec, a, b):
        n = len(rec)
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(b - a, out=off[1:])
        buf = np.empty(int(off[-1]), dtype=np.uint8)
        a32, b32 = a.astype(np.uint32), b.astype(np.uint32)
        rc = self._lib.svx_bam_seq_slices(self._h, rec.ctypes.data, a32.ctypes.data, b32.ctypes.data, n,
                                          off.ctypes.data, buf.ctypes.data)
        if rc != 0:
            raise ValueError("%s: %s" % (self.filename, self._lib.svx_bam_last_error(self._h).decode(errors="replace")))
        return buf, off.astype(np.int64)

    def sequence_slices_raw(self, rec, begin, end):
        """The same bases as one uint8 pool + offsets [n + 1] (what the columnar COLLECT keeps: no str per allele)."""
        self._ensure()
        rec = np.ascontiguousarray(rec, dtype=np.uint32)
        if len(rec) == 0:
            return np.zeros(0, np.uint8), np.zeros(1, np.int64)
        l_seq = self._cols["l_seq"][rec.astype(np.int64)]
        a = np.minimum(np.maximum(np.asarray(begin, dtype=np.int64), 0), l_seq)
        b = np.maximum(np.minimum(np.asarray(end, dtype=np.int64), l_seq), a)
        if self._h is None:
            parts = [self.record(i).seq_slice(x, y).encode("ascii") for i, x, y in zip(rec.tolist(), a.tolist(), b.tolist())]
            off = np.zeros(len(parts) + 1, np.int64)
            np.cumsum([len(p) for p in parts], out=off[1:])
            return np.frombuffer(b"".join(parts), dtype=np.uint8), off
        return self._slices_native(rec, a, b)

    def prefetch_sequence(self, requests):
        """requests: iterable of (record index, first base, last base + 1) that will be sliced soon
        (pure-Python reader: the covering BGZF blocks are inflated on a thread pool)."""
        if self._h is not None:
            return
        c = self._cols
        self._z.prefetch([(int(c["seq_off"][i]) + (a >> 1), ((b + 1) >> 1) - (a >> 1)) for i, a, b in requests])

    def batch(self, indices=None):
        """Flattened BAM-native CIGAR of the selected records:
        (cigar u32[n_ops], aln_off u64[n+1], ref_start i32[n], tid i32[n])."""
        c = self._cols
        if indices is None:
            cigar, n_cig, idx = self._cigar, c["n_cig"], slice(None)
        else:
            idx = np.asarray(indices, dtype=np.int64)
            n_cig = c["n_cig"][idx]
            parts = [self._cigar[self._cig_off[i]:self._cig_off[i + 1]] for i in idx.tolist()]
            cigar = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
        aln_off = np.concatenate(([0], np.cumsum(n_cig))).astype(np.uint64)
        return (np.ascontiguousarray(cigar, dtype=np.uint32), aln_off,
                c["pos"][idx].astype(np.int32), c["tid"][idx].astype(np.int32))

    # ---- pysam-compatible surface
    def check_index(self):
        if self.index_state() == 0:
            raise ValueError("mapping information not recorded in index or index not available")
        return True

    def indices_of_contig(self, tid):
        return np.nonzero(self._cols["tid"] == tid)[0]

    def fetch(self, contig=None, until_eof=False):
        if contig is None:
            self._ensure()
            return (self.record(i) for i in range(self.n_records))
        tid = self.get_tid(contig)
        if self._loaded is None or (self._loaded != "all" and tid not in self._loaded):
            self.load(None if self._loaded is None else set(self._loaded) | {tid})
        return (self.record(int(i)) for i in self.indices_of_contig(tid))

    def get_tid(self, name):
        return self._tid.get(name, -1)

    def get_reference_name(self, tid):
        if tid < 0 or tid >= len(self.references):
            raise ValueError("reference_id %i out of range 0<=tid<%i" % (tid, len(self.references)))
        return self.references[tid]

    getrname = get_reference_name

    def get_reference_length(self, name):
        by_name = self.__dict__.get("_len_by_name")
        if by_name is None:  # called once or twice per candidate: one dict lookup instead of two calls
            by_name = self.__dict__["_len_by_name"] = dict(zip(self.references, self.lengths))
        try:
            return by_name[name]
        except KeyError:
            raise KeyError("unknown reference %s" % name)

    def close(self):
        self._z = None
        if self._h is not None:
            self._c_cigar = None
            self._lib.svx_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------ writer
def encode_seq(seq):
    """ASCII bases → BAM 4-bit packed bytes."""
    a = np.frombuffer(seq.encode("ascii") if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    codes = _SEQ_ENC[a]
    if len(codes) & 1:
        codes = np.append(codes, 0)
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8).tobytes()


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def encode_record(qname, flag, tid, pos, mapq, cigar_words, seq, tags=None, seq_packed=None, l_seq=None):
    """One BAM alignment record (block_size prefix included).  `tags`: list of (tag, type, value)
    with any SAM aux type (A c C s S i I f d Z H, and B as (subtype, values)).  A CIGAR of more than 65535 operations does not fit the 16-bit
    n_cigar_op field: it is stored as the CG:B,I array behind the placeholder `<l_seq>S<ref_len>N`
    (SAM spec §4.2.2), as samtools/htslib write it."""
    cw = np.ascontiguousarray(cigar_words, dtype="<u4")
    name = qname.encode() + b"\x00"
    if seq_packed is None:
        l_seq = len(seq)
        seq_packed = encode_seq(seq)
    rlen = int(((cw >> 4).astype(np.int64) * ((0x18D >> (cw & 15)) & 1)).sum()) if len(cw) else 0
    end = pos + (rlen if rlen else 1)
    aux = b""
    for tag, typ, val in (tags or []):
        if typ == "Z":
            aux += tag.encode() + b"Z" + val.encode() + b"\x00"
        elif typ == "i":
            aux += tag.encode() + b"i" + struct.pack("<i", val)
        elif typ == "A":
            aux += tag.encode() + b"A" + val.encode()[:1]
        elif typ == "H":
            aux += tag.encode() + b"H" + val.encode() + b"\x00"
        elif typ in _AUX_FMT:     # c C s S I f d
            aux += tag.encode() + typ.encode() + struct.pack(_AUX_FMT[typ][0], val)
        elif typ == "B":          # val = (subtype, values)
            sub, vals = val
            aux += tag.encode() + b"B" + sub.encode() + struct.pack("<i", len(vals)) + \
                struct.pack("<%d%s" % (len(vals), _AUX_FMT[sub][0][1]), *vals)
        else:
            raise ValueError("unsupported aux type " + typ)
    stored = cw
    if len(cw) > 65535:
        stored = np.array([(l_seq << 4) | 4, (rlen << 4) | 3], dtype="<u4")
        aux += b"CGBI" + struct.pack("<I", len(cw)) + cw.tobytes()
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(name), mapq, _reg2bin(max(pos, 0), max(end, 1)),
                       len(stored), flag, l_seq, -1, -1, 0)
    body += name + stored.tobytes() + seq_packed + b"\xff" * l_seq + aux
    return struct.pack("<i", len(body)) + body


def record_extent(blob):
    """(tid, pos, end, flag) of an encoded record; end as htslib's bam_endpos (pos + 1 when the
    CIGAR consumes no reference), the real CIGAR taken from CG when the stored one is a placeholder."""
    bs, tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiiBBHHHi", blob, 0)
    words = np.frombuffer(blob, dtype="<u4", count=n_cig, offset=36 + l_rn)
    rlen = int(((words >> 4).astype(np.int64) * ((0x18D >> (words & 15)) & 1)).sum()) if n_cig else 0
    unmapped = bool(flag & 4)
    return tid, pos, pos + (rlen if rlen and not unmapped else 1), flag


_BAI_NO_OFFSET = (1 << 64) - 1


def build_bai(n_ref, tid, beg, end, flag, voff, voff_end):
    """Bytes of a `.bai` (SAM spec §5.2) for records given in file order: bins with their chunks
    (consecutive records of one bin share a chunk), the 16 kb linear index (offset of the first
    record overlapping each window, gaps back-filled like htslib), the per-contig metadata
    pseudo-bin 37450 and the count of unplaced reads."""
    out = [b"BAI\x01", struct.pack("<i", n_ref)]
    tid = np.asarray(tid, dtype=np.int64)
    n_no_coor = int((tid < 0).sum())
    for r in range(n_ref):
        idx = np.nonzero(tid == r)[0]
        if len(idx) == 0:
            out.append(struct.pack("<ii", 0, 0))
            continue
        bins, order = {}, []
        last_bin = None
        n_win = (max(int(end[i]) for i in idx) - 1 >> 14) + 1
        lin = np.full(n_win, _BAI_NO_OFFSET, dtype=np.uint64)
        n_mapped = n_unmapped = 0
        for i in idx.tolist():
            b, e = max(int(beg[i]), 0), max(int(end[i]), 1)
            bn = _reg2bin(b, e)
            if bn == last_bin:
                bins[bn][-1][1] = int(voff_end[i])
            else:
                if bn not in bins:
                    bins[bn] = []
                    order.append(bn)
                bins[bn].append([int(voff[i]), int(voff_end[i])])
            last_bin = bn
            w0, w1 = b >> 14, (e - 1) >> 14
            np.minimum(lin[w0:w1 + 1], np.uint64(voff[i]), out=lin[w0:w1 + 1])
            if int(flag[i]) & 4:
                n_unmapped += 1
            else:
                n_mapped += 1
        for w in range(n_win - 2, -1, -1):
            if lin[w] == _BAI_NO_OFFSET:
                lin[w] = lin[w + 1]
        out.append(struct.pack("<i", len(order) + 1))
        for bn in order:
            out.append(struct.pack("<Ii", bn, len(bins[bn])))
            for cb, ce in bins[bn]:
                out.append(struct.pack("<QQ", cb, ce))
        out.append(struct.pack("<Ii", 37450, 2))
        out.append(struct.pack("<QQQQ", int(voff[idx[0]]), int(voff_end[idx[-1]]), n_mapped, n_unmapped))
        out.append(struct.pack("<i", n_win))
        out.append(lin.astype("<u8").tobytes())
    out.append(struct.pack("<Q", n_no_coor))
    return b"".join(out)


def write_bam(path, references, lengths, record_blobs, sort_order="coordinate", level=1, write_index=True):
    """Write a BAM file from already-encoded records (see encode_record) and its `.bai`."""
    text = "@HD\tVN:1.6\tSO:%s\n" % sort_order
    text += "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(references, lengths))
    tb = text.encode()
    hdr = b"BAM\x01" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(references))
    for n, l in zip(references, lengths):
        nb = n.encode() + b"\x00"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    data = hdr + b"".join(record_blobs)
    comp = bgzf_compress(data, level)
    with open(path, "wb") as fh:
        fh.write(comp)
    if write_index:
        # member start offsets of the file just written (fixed 0xFF00-byte payloads, then the EOF member)
        spans = _bgzf_block_spans(comp)
        starts = [sp[3] for sp in spans]

        def voffset(u):  # end-of-member positions are written as the start of the next member
            if u >= len(data):
                return starts[-1] << 16  # the EOF member
            return (starts[u // 0xFF00] << 16) | (u % 0xFF00)
        u = len(hdr)
        tid, beg, end, flag, vo, ve = [], [], [], [], [], []
        for blob in record_blobs:
            t, b, e, f = record_extent(blob)
            tid.append(t); beg.append(b); end.append(e); flag.append(f)
            vo.append(voffset(u))
            u += len(blob)
            ve.append(voffset(u))
        with open(path + ".bai", "wb") as fh:
            fh.write(build_bai(len(references), tid, beg, end, flag, vo, ve))


def _reg2bin_csi(beg, end, min_shift, depth):
    """CSI v1 specification, reg2bin for an index of `depth` levels above `min_shift`-bit leaves."""
    end -= 1
    s, t = min_shift, ((1 << depth * 3) - 1) // 7
    for l in range(depth, 0, -1):
        if beg >> s == end >> s:
            return t + (beg >> s)
        s += 3
        t -= 1 << ((l - 1) * 3)
    return 0


def build_csi(n_ref, tid, beg, end, flag, voff, voff_end, min_shift=14, depth=5):
    """Bytes of a `.csi` (CSI v1; what `samtools index -c` writes) for records given in file order: per sequence the
    bins with their chunks and `loffset` (virtual offset of the first record that overlaps the bin), the metadata
    pseudo-bin, then the count of unplaced reads; BGZF-compressed."""
    out = [b"CSI\x01", struct.pack("<iii", min_shift, depth, 0), struct.pack("<i", n_ref)]
    tid = np.asarray(tid, dtype=np.int64)
    meta = ((1 << (depth + 1) * 3) - 1) // 7 + 1
    for r in range(n_ref):
        idx = np.nonzero(tid == r)[0]
        if len(idx) == 0:
            out.append(struct.pack("<i", 0))
            continue
        bins, order, loff, last_bin = {}, [], {}, None
        n_mapped = n_unmapped = 0
        for i in idx.tolist():
            b, e = max(int(beg[i]), 0), max(int(end[i]), 1)
            bn = _reg2bin_csi(b, e, min_shift, depth)
            if bn == last_bin:
                bins[bn][-1][1] = int(voff_end[i])
            else:
                if bn not in bins:
                    bins[bn] = []
                    order.append(bn)
                bins[bn].append([int(voff[i]), int(voff_end[i])])
            last_bin = bn
            # every bin the record overlaps, on every level, learns the first record that reaches into it
            s, t = min_shift, ((1 << depth * 3) - 1) // 7
            for l in range(depth, -1, -1):
                for k in range(b >> s, ((e - 1) >> s) + 1):
                    loff.setdefault(t + k, int(voff[i]))
                s += 3
                t -= 1 << ((l - 1) * 3) if l > 0 else 0
            if int(flag[i]) & 4:
                n_unmapped += 1
            else:
                n_mapped += 1
        out.append(struct.pack("<i", len(order) + 1))
        for bn in order:
            out.append(struct.pack("<IQi", bn, loff.get(bn, 0), len(bins[bn])))
            for cb, ce in bins[bn]:
                out.append(struct.pack("<QQ", cb, ce))
        out.append(struct.pack("<IQi", meta, 0, 2))
        out.append(struct.pack("<QQQQ", int(voff[idx[0]]), int(voff_end[idx[-1]]), n_mapped, n_unmapped))
    out.append(struct.pack("<Q", int((tid < 0).sum())))
    return bgzf_compress(b"".join(out), 6)


def index_bam(path, out=None, csi=False, min_shift=14, depth=5):
    """`samtools index` (csi=True: `samtools index -c`) for the files this package reads: walk the records once
    (native reader, sequential without an index) and write `<path>.bai` / `<path>.csi`."""
    f = AlignmentFile(path, reader="native")
    # ignore whatever index is there: virtual offsets come from the walk itself
    c = f.load(None)._cols
    n = f.n_records
    vo = c["voffset"].astype(np.uint64)
    size = os.path.getsize(path)
    with open(path, "rb") as fh:
        fh.seek(max(0, size - 28))
        eof_at = size - 28 if fh.read(28) == _BGZF_EOF else size
    ve = np.concatenate((vo[1:], [np.uint64(eof_at << 16)])) if n else vo
    unmapped = (c["flag"] & 4) != 0
    rl = np.where((c["ref_len"] > 0) & ~unmapped, c["ref_len"], 1)
    if csi:
        data = build_csi(len(f.references), c["tid"], c["pos"], c["pos"] + rl, c["flag"], vo, ve, min_shift, depth)
    else:
        data = build_bai(len(f.references), c["tid"], c["pos"], c["pos"] + rl, c["flag"], vo, ve)
    f.close()
    out = out or (path + (".csi" if csi else ".bai"))
    with open(out, "wb") as fh:
        fh.write(data)
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 3 and sys.argv[1] == "index":
        print(index_bam(sys.argv[2]))
    else:
        print("usage: python -m svim_asm_amd.bamio index <file.bam>")
