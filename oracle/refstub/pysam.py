"""Stand-in for the `pysam` module so that the REAL reference (/root/reference) can be imported
and run end-to-end in the build container, where pysam/htslib are not installed.

TEST INFRASTRUCTURE ONLY (used by oracle/make_golden.py to generate golden vectors and by
oracle pin tests).  It implements just the htslib/pysam semantics the reference touches
 with a small, deliberately simple pure-Python BAM/FASTA reader that
shares no code with the product's reader (svim_asm_amd/bamio.py).
"""
import gzip
import os
import re
import struct

_CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=XB])")
_CODES = "MIDNSHP=XB"
_SEQ = "=ACMGRSVTWYHKDBN"
_HEX2BASE = str.maketrans("0123456789abcdef", _SEQ)


class AlignedSegment(object):
    def __init__(self, header=None):
        self.query_name = None
        self._seq = ""
        self.flag = 0
        self.reference_id = -1
        self.reference_start = -1
        self._mapq = 0
        self._cigar = []  # list of (op, len)
        self.next_reference_id = -1
        self.next_reference_start = -1
        self.template_length = 0
        self.query_qualities = None
        self._tags = {}

    # ---- flag bits
    @property
    def is_unmapped(self):
        return bool(self.flag & 0x4)

    @property
    def is_secondary(self):
        return bool(self.flag & 0x100)

    @property
    def is_supplementary(self):
        return bool(self.flag & 0x800)

    @property
    def is_reverse(self):
        return bool(self.flag & 0x10)

    # ---- uint8 mapping quality (Cython raises OverflowError outside 0..255)
    @property
    def mapping_quality(self):
        return self._mapq

    @mapping_quality.setter
    def mapping_quality(self, v):
        v = int(v)
        if v < 0 or v > 255:
            raise OverflowError("value does not fit uint8_t")
        self._mapq = v

    # ---- sequence
    @property
    def query_sequence(self):
        return self._seq if self._seq else None if self._seq is None else self._seq

    @query_sequence.setter
    def query_sequence(self, s):
        self._seq = s if s is not None else ""

    # ---- CIGAR
    @property
    def cigarstring(self):
        if not self._cigar:
            return None
        return "".join("%d%s" % (l, _CODES[o]) for o, l in self._cigar)

    @cigarstring.setter
    def cigarstring(self, s):
        if s is None or len(s) == 0:
            self._cigar = []
            return
        tuples = [(_CODES.index(c), int(n)) for n, c in _CIGAR_RE.findall(s)]
        for _, l in tuples:
            if l >= (1 << 28):
                raise OverflowError("CIGAR length does not fit 28 bits")
        self._cigar = tuples

    @property
    def cigartuples(self):
        return list(self._cigar) if self._cigar else None

    @cigartuples.setter
    def cigartuples(self, t):
        self._cigar = [(int(o), int(l)) for o, l in (t or [])]

    def get_cigar_stats(self):
        base = [0] * 11
        cnt = [0] * 11
        for o, l in self._cigar:
            base[o] += l
            cnt[o] += 1
        nm = self._tags.get("NM")
        if nm is not None:
            base[10] = nm
        return base, cnt

    @property
    def reference_end(self):
        if self.is_unmapped or not self._cigar:
            return None
        rlen = sum(l for o, l in self._cigar if o in (0, 2, 3, 7, 8))
        if rlen == 0:
            rlen = 1  # htslib bam_endpos
        return self.reference_start + rlen

    @property
    def query_alignment_start(self):
        start = 0
        for o, l in self._cigar:
            if o == 5:
                continue
            elif o == 4:
                start += l
            else:
                break
        return start

    @property
    def query_alignment_end(self):
        lq = len(self._seq) if self._seq else 0
        if lq == 0:
            end = 0
            for o, l in self._cigar:
                if o in (0, 1, 7, 8) or (o == 4 and end == 0):
                    end += l
            return end
        end = lq
        for k in range(len(self._cigar) - 1, 0, -1):
            o, l = self._cigar[k]
            if o == 5:
                continue
            elif o == 4:
                end -= l
            else:
                break
        return end

    def infer_read_length(self):
        if not self._cigar:
            return None
        return sum(l for o, l in self._cigar if o in (0, 1, 4, 5, 7, 8))

    def infer_query_length(self):
        if not self._cigar:
            return None
        return sum(l for o, l in self._cigar if o in (0, 1, 4, 7, 8))

    # ---- tags
    def set_tags(self, tags):
        self._tags = {}
        for t in tags:
            self._tags[t[0]] = t[1]

    def get_tag(self, name):
        if name not in self._tags:
            raise KeyError("tag '%s' not present" % name)
        return self._tags[name]

    def has_tag(self, name):
        return name in self._tags


class _Header(dict):
    pass


def _read_bgzf(path):
    # BGZF is a series of gzip members; python's gzip handles multi-member files
    with gzip.open(path, "rb") as f:
        return f.read()


def _parse_header_text(text):
    hdr = _Header()
    for line in text.split("\n"):
        if not line.startswith("@"):
            continue
        fields = line.split("\t")
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


class AlignmentFile(object):
    def __init__(self, path, mode="rb"):
        self.filename = path
        data = _read_bgzf(path)
        if data[:4] != b"BAM\x01":
            raise ValueError("not a BAM file")
        l_text, = struct.unpack_from("<i", data, 4)
        text = data[8:8 + l_text].split(b"\x00")[0].decode()
        p = 8 + l_text
        n_ref, = struct.unpack_from("<i", data, p)
        p += 4
        self.references = []
        self.lengths = []
        for _ in range(n_ref):
            l_name, = struct.unpack_from("<i", data, p)
            p += 4
            self.references.append(data[p:p + l_name - 1].decode())
            p += l_name
            l_ref, = struct.unpack_from("<i", data, p)
            p += 4
            self.lengths.append(l_ref)
        self.references = tuple(self.references)
        self.lengths = tuple(self.lengths)
        self.header = _parse_header_text(text)
        self._records = []
        n = len(data)
        while p + 4 <= n:
            bs, = struct.unpack_from("<i", data, p)
            p += 4
            self._records.append(self._parse_record(data, p, bs))
            p += bs

    @staticmethod
    def _parse_record(data, p, bs):
        tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, ntid, npos, tlen = struct.unpack_from("<iiBBHHHiiii", data, p)
        q = p + 32
        a = AlignedSegment()
        a.query_name = data[q:q + l_rn - 1].decode()
        q += l_rn
        words = struct.unpack_from("<%dI" % n_cig, data, q) if n_cig else ()
        q += 4 * n_cig
        a._cigar = [(w & 15, w >> 4) for w in words]
        nb = (l_seq + 1) // 2
        sb = data[q:q + nb]
        q += nb
        # one hex digit per 4-bit base code, then a character translation (fast, still independent
        # of the product's numpy decoder)
        a._seq = sb.hex().translate(_HEX2BASE)[:l_seq]
        q += l_seq  # qualities
        a.flag = flag
        a.reference_id = tid
        a.reference_start = pos
        a._mapq = mapq
        a.next_reference_id = ntid
        a.next_reference_start = npos
        a.template_length = tlen
        end = p + bs
        tags = {}
        cg_subtype = None
        while q < end:
            tag = data[q:q + 2].decode()
            typ = chr(data[q + 2])
            q += 3
            if typ == "Z" or typ == "H":
                e = data.index(b"\x00", q)
                tags[tag] = data[q:e].decode()
                q = e + 1
            elif typ == "A":
                tags[tag] = chr(data[q]); q += 1
            elif typ in "cCsSiIf":
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}[typ]
                tags[tag], = struct.unpack_from(fmt, data, q)
                q += struct.calcsize(fmt)
            elif typ == "B":
                sub = chr(data[q]); cnt, = struct.unpack_from("<i", data, q + 1)
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
                tags[tag] = list(struct.unpack_from("<%d%s" % (cnt, fmt), data, q + 5))
                if tag == "CG":
                    cg_subtype = sub
                q += 5 + cnt * struct.calcsize(fmt)
            else:
                raise ValueError("unknown aux type " + typ)
        # htslib bam_tag2cigar (applied by sam_read1 / pysam on every record): a CIGAR longer than
        # 65535 operations is stored in the CG:B,I tag behind a `<l_seq>S<ref_len>N` placeholder
        # (SAM spec §4.2.2); the real CIGAR replaces the placeholder and the tag disappears
        if (cg_subtype in ("I", "i") and a._cigar and tid >= 0 and pos >= 0 and a._cigar[0] == (4, l_seq)
                and len(a._cigar) <= len(tags["CG"]) < (1 << 29)):
            a._cigar = [(w & 15, w >> 4) for w in (v & 0xFFFFFFFF for v in tags.pop("CG"))]
        a._tags = tags
        return a

    def check_index(self):
        if not (os.path.exists(self.filename + ".bai") or os.path.exists(self.filename + ".csi") or
                os.path.exists(os.path.splitext(self.filename)[0] + ".bai")):
            raise ValueError("mapping information not recorded in index or index not available")
        return True

    def fetch(self, contig=None, until_eof=False):
        if contig is None:
            return iter(list(self._records))
        tid = self.get_tid(contig)
        # htslib index queries return mapped-region overlaps only: placed records of this tid
        return iter([r for r in self._records if r.reference_id == tid])

    def get_tid(self, name):
        try:
            return self.references.index(name)
        except ValueError:
            return -1

    def getrname(self, tid):
        return self.get_reference_name(tid)

    def get_reference_name(self, tid):
        if tid < 0 or tid >= len(self.references):
            raise ValueError("reference_id %i out of range" % tid)
        return self.references[tid]

    def get_reference_length(self, name):
        tid = self.get_tid(name)
        if tid < 0:
            raise KeyError("unknown reference " + str(name))
        return self.lengths[tid]

    def close(self):
        pass


class FastaFile(object):
    def __init__(self, path):
        if not os.path.exists(path):
            raise IOError("file `%s` not found" % path)
        if not os.path.exists(path + ".fai"):
            raise ValueError("no index for " + path)
        self._idx = {}
        self.references = []
        self.lengths = []
        for line in open(path + ".fai"):
            f = line.rstrip("\n").split("\t")
            if len(f) < 5:
                continue
            self._idx[f[0]] = tuple(int(x) for x in f[1:5])
            self.references.append(f[0])
            self.lengths.append(int(f[1]))
        self._fh = open(path, "rb")

    def get_reference_length(self, name):
        return self._idx[name][0]

    def fetch(self, reference, start=None, end=None):
        length, offset, lb, lw = self._idx[reference]
        start = 0 if start is None else start
        end = length if end is None else end
        if start < 0:
            raise ValueError("start out of range (%i)" % start)
        if end < start:
            raise ValueError("end out of range")
        end = min(end, length)
        if start >= end:
            return ""
        b0 = offset + (start // lb) * lw + start % lb
        b1 = offset + ((end - 1) // lb) * lw + (end - 1) % lb + 1
        self._fh.seek(b0)
        raw = self._fh.read(b1 - b0)
        return raw.replace(b"\n", b"").replace(b"\r", b"").decode()

    def close(self):
        self._fh.close()
