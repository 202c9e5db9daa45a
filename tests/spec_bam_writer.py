"""A BAM / BGZF / BAI writer for tests, written from the SAM specification alone (SAMv1 §4.1 BGZF, §4.2 BAM,
§5.2 BAI; hts-specs) — it imports NOTHING from svim_asm_amd, so a file it writes is not the product's own
writer talking to the product's own reader.  Test infrastructure only.

Records are given as plain dicts of literal field values (what a test then expects every reader to hand back):
    name, flag, tid, pos, mapq, cigar [(op, len), ...], seq (str, "" for '*'), qual (bytes or None),
    tags [(tag, type, value), ...] in file order — type one of A c C s S i I f Z H, or ("B", subtype) with a list.
"""
import struct
import zlib

CIGAR_OPS = "MIDNSHP=XB"
SEQ_CODES = "=ACMGRSVTWYHKDBN"


def reg2bin(beg, end):  # SAMv1 §5.3
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def reference_length(cigar):
    return sum(ln for op, ln in cigar if op in (0, 2, 3, 7, 8))  # M D N = X


def encode_aux(tags):
    out = bytearray()
    for tag, typ, val in tags:
        out += tag.encode()
        if isinstance(typ, tuple):  # ("B", subtype)
            sub = typ[1]
            out += b"B" + sub.encode() + struct.pack("<i", len(val))
            out += struct.pack("<%d%s" % (len(val), {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]), *val)
        elif typ in "ZH":
            out += typ.encode() + val.encode() + b"\x00"
        elif typ == "A":
            out += b"A" + val.encode()
        else:
            out += typ.encode() + struct.pack({"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}[typ], val)
    return bytes(out)


def encode_record(r, long_cigar_as_tag=True):
    """One alignment record (§4.2) including its block_size prefix.  A CIGAR of more than 65535 operations goes into
    a CG:B,I tag behind the `<l_seq>S<ref_len>N` placeholder (§4.2.2)."""
    name = r["name"].encode() + b"\x00"
    cigar = list(r.get("cigar") or [])
    seq = r.get("seq", "")
    l_seq = len(seq)
    tags = list(r.get("tags") or [])
    end = r["pos"] + (reference_length(cigar) or 1)
    if len(cigar) > 65535 and long_cigar_as_tag:
        words = [(ln << 4) | op for op, ln in cigar]
        tags = tags + [("CG", ("B", "I"), words)]
        cigar = [(4, l_seq), (3, reference_length(cigar))]
    packed = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(seq):
        code = SEQ_CODES.index(ch.upper()) if ch.upper() in SEQ_CODES else 15
        packed[i // 2] |= code << (4 if i % 2 == 0 else 0)
    qual = r.get("qual")
    qual = bytes([0xFF]) * l_seq if qual is None else bytes(qual)
    assert len(qual) == l_seq
    body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r["mapq"], reg2bin(max(r["pos"], 0), max(end, 1)) if r["tid"] >= 0 else 4680,
                       len(cigar), r["flag"], l_seq, r.get("next_tid", -1), r.get("next_pos", -1), r.get("tlen", 0))
    body += name + b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in cigar) + bytes(packed) + qual + encode_aux(tags)
    return struct.pack("<i", len(body)) + body


def bgzf_member(payload, level=6):
    """One BGZF block (§4.1): gzip member with the BC extra field; payload at most 65280 bytes here."""
    assert len(payload) <= 65536
    if level == 0:
        comp = zlib.compressobj(0, zlib.DEFLATED, -15)
    else:
        comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = comp.compress(payload) + comp.flush()
    bsize = 12 + 6 + len(cdata) + 8
    assert bsize <= 65536
    head = struct.pack("<BBBBIBBH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return head + cdata + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


EOF_MARKER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_bam(path, refs, records, header_text=None, cuts=None, chunk=65280, level=6, empty_after=(), index=True,
              pseudo_bins=True, n_no_coor=True, extra_subfield=False):
    """refs: [(name, length)]; records in file order.  The uncompressed stream is cut into members at the byte
    positions `cuts` (absolute offsets into the stream) or every `chunk` bytes; after member number k in
    `empty_after` an EMPTY member (ISIZE 0) is inserted.  Returns the virtual offset (coffset << 16 | uoffset) of
    every record's start and end."""
    if header_text is None:
        header_text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in refs)
    text = header_text.encode()
    stream = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs)))
    for n, l in refs:
        stream += struct.pack("<i", len(n) + 1) + n.encode() + b"\x00" + struct.pack("<i", l)
    starts, ends = [], []
    for r in records:
        starts.append(len(stream))
        stream += encode_record(r)
        ends.append(len(stream))
    if cuts is None:
        cuts = list(range(chunk, len(stream), chunk))
    bounds = [0] + sorted(set(c for c in cuts if 0 < c < len(stream))) + [len(stream)]
    out = bytearray()
    member_at = []  # (uncompressed start, uncompressed end, compressed offset)
    k = 0
    for a, b in zip(bounds, bounds[1:]):
        while b - a > 65280:
            member_at.append((a, a + 65280, len(out)))
            out += bgzf_member(bytes(stream[a:a + 65280]), level)
            a += 65280
        member_at.append((a, b, len(out)))
        m = bgzf_member(bytes(stream[a:b]), level)
        if extra_subfield:  # another extra subfield in front of BC: readers must walk the subfields
            cdata = m[18:]
            bsize = 12 + 6 + 6 + len(cdata)
            m = struct.pack("<BBBBIBBH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 12) + b"XY" + struct.pack("<HH", 2, 0xABCD) + b"BC" + \
                struct.pack("<HH", 2, bsize - 1) + cdata
        out += m
        if k in empty_after:
            out += bgzf_member(b"", level)
        k += 1
    out += EOF_MARKER
    with open(path, "wb") as fh:
        fh.write(bytes(out))

    def voffset(u):  # canonical: a position at the end of a member is the start of the next one that holds data
        for a, b, c in member_at:
            if a <= u < b:
                return (c << 16) | (u - a)
        return (len(out) - len(EOF_MARKER)) << 16  # end of data: the EOF block

    v_start = [voffset(u) for u in starts]
    v_end = [voffset(u) for u in ends]
    if index:
        write_bai(path + ".bai", len(refs), records, v_start, v_end, pseudo_bins, n_no_coor)
    return v_start, v_end


def write_bai(path, n_ref, records, v_start, v_end, pseudo_bins=True, n_no_coor=True):
    """§5.2: per reference the bins with their chunks, the 16 kbp linear index, optionally the pseudo-bin 37450 and
    the trailing count of unplaced reads."""
    per = [dict(bins={}, lin={}, first=None, last=None, mapped=0, unmapped=0) for _ in range(n_ref)]
    unplaced = 0
    for r, vs, ve in zip(records, v_start, v_end):
        if r["tid"] < 0:
            unplaced += 1
            continue
        d = per[r["tid"]]
        end = r["pos"] + (reference_length(r.get("cigar") or []) or 1)
        b = reg2bin(r["pos"], end)
        chunks = d["bins"].setdefault(b, [])
        if chunks and chunks[-1][1] == vs:
            chunks[-1][1] = ve
        else:
            chunks.append([vs, ve])
        for w in range(r["pos"] >> 14, ((end - 1) >> 14) + 1):
            d["lin"].setdefault(w, vs)
        d["first"] = vs if d["first"] is None else d["first"]
        d["last"] = ve
        if r["flag"] & 4:
            d["unmapped"] += 1
        else:
            d["mapped"] += 1
    out = bytearray(b"BAI\x01" + struct.pack("<i", n_ref))
    for d in per:
        n_bin = len(d["bins"]) + (1 if pseudo_bins and d["first"] is not None else 0)
        out += struct.pack("<i", n_bin)
        for b in sorted(d["bins"]):
            out += struct.pack("<Ii", b, len(d["bins"][b]))
            for vs, ve in d["bins"][b]:
                out += struct.pack("<QQ", vs, ve)
        if pseudo_bins and d["first"] is not None:
            out += struct.pack("<Ii", 37450, 2) + struct.pack("<QQ", d["first"], d["last"]) + struct.pack("<QQ", d["mapped"], d["unmapped"])
        n_intv = (max(d["lin"]) + 1) if d["lin"] else 0
        out += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):
            last = d["lin"].get(w, last)
            out += struct.pack("<Q", last)
    if n_no_coor:
        out += struct.pack("<Q", unplaced)
    with open(path, "wb") as fh:
        fh.write(bytes(out))
