"""Synthetic genome-genome alignments as real BAM + FASTA files.

A haplotype is simulated against a random reference: contigs are tiled with primary
alignments whose CIGARs carry small indels, SV-sized insertions/deletions (shared between the
two haplotypes of a diploid sample, shifted by a few bp, or private), soft clips, and a set of
split reads (primary + SA tag [+ supplementary records]) engineered to reach every branch
family of the split-segment analysis: insertion, deletion, breakend, tandem duplication,
interspersed duplication, inversion.  Sequences are consistent with the CIGARs (M copies the
reference with a low substitution rate, I inserts random bases), so INS alleles and haplotype
edit distances are meaningful.  Deterministic for a given seed.
"""
import numpy as np

from svim_asm_amd import bamio
from svim_asm_amd.fasta import write_fasta

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b

CONFIG1_CONTIGS = (("chr1", 200000), ("chr10", 150000), ("chr2", 100000))


def random_genome(seed, contigs):
    rng = np.random.default_rng(seed)
    return {name: _BASES[rng.integers(0, 4, size=length)] for name, length in contigs}


def revcomp(a):
    return _COMP[a[::-1]]


def _rand_bases(rng, n):
    return _BASES[rng.integers(0, 4, size=n)]


class Event(object):
    """An SV carried inside a primary alignment's CIGAR."""
    __slots__ = ("pos", "kind", "length", "seq")

    def __init__(self, pos, kind, length, seq=None):
        self.pos, self.kind, self.length, self.seq = pos, kind, length, seq


def make_events(rng, contig_len, n, min_len=40, max_len=2000, min_gap=1500):
    """n non-overlapping SV-sized insertion/deletion events on one contig, at least `min_gap` bp apart (below the
    pairing step's partition distance of 1000 neighbouring events chain into crowded partitions: config 5)."""
    out, pos = [], 2000
    for _ in range(n):
        pos += int(rng.integers(min_gap, max(min_gap + 1, (contig_len - 4000) // max(n, 1))))
        if pos + max_len + 3000 >= contig_len:
            break
        length = int(np.exp(rng.uniform(np.log(min_len), np.log(max_len))))
        if rng.random() < 0.5:
            out.append(Event(pos, "D", length))
            pos += length
        else:
            out.append(Event(pos, "I", length, _rand_bases(rng, length)))
    return out


def derive_haplotype_events(rng, shared, private_n, contig_len, shift_frac=0.25, keep_frac=0.75):
    """Events of one haplotype: a subset of the shared set (some shifted by ±1..50 bp or with an
    altered insertion allele — edit distance straddling the pairing threshold) + private ones."""
    out = []
    for ev in shared:
        if rng.random() > keep_frac:
            continue
        e = Event(ev.pos, ev.kind, ev.length, None if ev.seq is None else ev.seq.copy())
        if rng.random() < shift_frac:
            e.pos += int(rng.integers(-50, 51))
            if e.kind == "I":
                k = int(rng.integers(0, min(e.length, 400)))
                idx = rng.integers(0, e.length, size=k)
                e.seq[idx] = _rand_bases(rng, k)
        out.append(e)
    priv = make_events(rng, contig_len, private_n)
    taken = sorted((e.pos, e.pos + (e.length if e.kind == "D" else 1)) for e in out)
    for p in priv:
        if all(p.pos + p.length + 200 < a or p.pos > b + 200 for a, b in taken):
            out.append(p)
    out.sort(key=lambda e: e.pos)
    # enforce spacing so that CIGARs stay well-formed
    spaced, last_end = [], 0
    for e in out:
        if e.pos >= last_end + 100:
            spaced.append(e)
            last_end = e.pos + (e.length if e.kind == "D" else 0)
    return spaced


def _tile_alignment(rng, ref, a, b, events, mean_m, sub_rate, soft_clip):
    """CIGAR words + query bases of one forward-strand alignment covering reference [a, b)."""
    ops, seq_parts = [], []

    def push(op, ln):
        if ln <= 0:
            return
        if ops and ops[-1][0] == op:
            ops[-1][1] += ln
        else:
            ops.append([op, ln])

    def match(lo, hi):
        if hi <= lo:
            return
        chunk = ref[lo:hi].copy()
        if sub_rate > 0 and hi - lo > 0:
            k = rng.binomial(hi - lo, sub_rate)
            if k:
                idx = rng.integers(0, hi - lo, size=k)
                chunk[idx] = _rand_bases(rng, k)
        seq_parts.append(chunk)
        push(0, hi - lo)

    if soft_clip[0]:
        seq_parts.append(_rand_bases(rng, soft_clip[0]))
        push(4, soft_clip[0])
    pos = a
    evs = [e for e in events if a + 50 <= e.pos and e.pos + (e.length if e.kind == "D" else 0) <= b - 50]
    ei = 0
    first = True
    while pos < b:
        nxt_ev = evs[ei].pos if ei < len(evs) else b
        run = int(rng.geometric(1.0 / mean_m))
        stop = min(pos + run, nxt_ev, b)
        if first and stop == pos:      # an alignment starts with a match
            stop = min(pos + 1, b)
        first = False
        match(pos, stop)
        pos = stop
        if pos >= b:
            break
        if ei < len(evs) and pos == evs[ei].pos:
            e = evs[ei]
            ei += 1
            if e.kind == "D":
                push(2, e.length)
                pos += e.length
            else:
                seq_parts.append(e.seq)
                push(1, e.length)
            match(pos, min(pos + 1, b))   # keep indels separated by a match
            pos = min(pos + 1, b)
        else:
            ln = int(rng.integers(1, 30))
            if rng.random() < 0.5 and pos + ln + 2 < min(nxt_ev, b):
                push(2, ln)
                pos += ln
            else:
                seq_parts.append(_rand_bases(rng, ln))
                push(1, ln)
            match(pos, min(pos + 1, min(nxt_ev, b)))
            pos = min(pos + 1, min(nxt_ev, b)) if pos < min(nxt_ev, b) else pos
    if ops and ops[-1][0] in (1, 2):   # and ends with one
        ops.pop() if ops[-1][0] == 2 else None
    if soft_clip[1]:
        seq_parts.append(_rand_bases(rng, soft_clip[1]))
        push(4, soft_clip[1])
    words = np.array([(ln << 4) | op for op, ln in ops], dtype=np.uint32)
    seq = np.concatenate(seq_parts) if seq_parts else np.zeros(0, np.uint8)
    # keep SEQ length consistent with the CIGAR
    qlen = int(sum(ln for op, ln in ops if op in (0, 1, 4)))
    assert qlen == len(seq), (qlen, len(seq))
    return words, seq


def _split_read(rng, genome, names, segments, qname, gaps=None, emit_supplementary=True, mapq=60,
                primary_index=0):
    """Records of one chimeric read.  segments: [(contig, ref_start, length, reverse)], laid out
    on the read in list order with `gaps[i]` read bases between segment i and i+1 (negative =
    overlap).  Returns record dicts (primary with SA tag, optional supplementary records)."""
    gaps = gaps or [0] * (len(segments) - 1)
    # read coordinates
    q, spans = 0, []
    for i, (_, _, ln, _) in enumerate(segments):
        spans.append((q, q + ln))
        q += ln + (gaps[i] if i < len(gaps) else 0)
    read_len = max(e for _, e in spans)
    read = _rand_bases(rng, read_len)
    for (contig, rs, ln, rev), (qs, qe) in zip(segments, spans):
        piece = genome[contig][rs:rs + ln]
        read[qs:qe] = revcomp(piece) if rev else piece

    def cigar_for(i):
        (_, _, ln, rev), (qs, qe) = segments[i], spans[i]
        before, after = (read_len - qe, qs) if rev else (qs, read_len - qe)
        ops = []
        if before:
            ops.append((before << 4) | 4)
        ops.append((ln << 4) | 0)
        if after:
            ops.append((after << 4) | 4)
        return np.array(ops, dtype=np.uint32)

    def cigar_string(words):
        return "".join("%d%s" % (int(w) >> 4, "MIDNSHP=XB"[int(w) & 15]) for w in words)

    recs = []
    for i, (contig, rs, ln, rev) in enumerate(segments):
        if i != primary_index and not emit_supplementary:
            continue
        others = [j for j in range(len(segments)) if j != i]
        sa = "".join("%s,%d,%s,%s,%d,0;" % (segments[j][0], segments[j][1] + 1, "-" if segments[j][3] else "+",
                                             cigar_string(cigar_for(j)), mapq) for j in others)
        seq = revcomp(read) if rev else read
        recs.append(dict(qname=qname, flag=(16 if rev else 0) | (0 if i == primary_index else 2048),
                         tid=names.index(contig), pos=rs, mapq=mapq, cigar=cigar_for(i), seq=seq, sa=sa))
    return recs


def split_read_zoo(rng, genome, names, lengths, tag):
    """Chimeric reads covering each branch family of analyze_read_segments."""
    c0, c1 = names[0], names[-1]
    L0 = lengths[0]
    recs = []
    base = L0 // 8
    # insertion: read gap of 300 bp, reference contiguous
    recs += _split_read(rng, genome, names, [(c0, base, 3000, False), (c0, base + 3000, 2500, False)],
                        "split_ins_" + tag, gaps=[300])
    # deletion: 5 kb reference gap
    recs += _split_read(rng, genome, names, [(c0, 2 * base, 2500, False), (c0, 2 * base + 7500, 2500, False)],
                        "split_del_" + tag)
    # deletion seen from the reverse strand
    recs += _split_read(rng, genome, names, [(c0, 2 * base + 20000, 2000, True), (c0, 2 * base + 15000, 2500, True)],
                        "split_delrev_" + tag)
    # breakend between contigs
    recs += _split_read(rng, genome, names, [(c0, 3 * base, 3000, False), (c1, 10000, 3000, False)],
                        "split_bnd_" + tag)
    # tandem duplication: second segment starts inside the first on the reference (twice → 2 copies)
    recs += _split_read(rng, genome, names, [(c0, 4 * base, 3000, False), (c0, 4 * base + 2000, 3000, False),
                                             (c0, 4 * base + 4002, 3000, False)], "split_tan_" + tag)
    # interspersed duplication: chr A → chr B → back to chr A where it left off
    recs += _split_read(rng, genome, names, [(c0, 5 * base, 2500, False), (c1, 30000, 800, False),
                                             (c0, 5 * base + 2500, 2500, False)], "split_dupint_" + tag)
    # inversion with both breakpoints: fwd → rev → fwd
    recs += _split_read(rng, genome, names, [(c0, 6 * base, 3000, False), (c0, 6 * base + 3000, 1500, True),
                                             (c0, 6 * base + 4500, 3000, False)], "split_inv_" + tag)
    # a lone inversion breakpoint (incomplete) and a very large jump (breakend on one contig)
    recs += _split_read(rng, genome, names, [(c0, 7 * base, 2500, False), (c0, 7 * base + 4000, 1500, True)],
                        "split_inv1_" + tag)
    recs += _split_read(rng, genome, names, [(c0, 1000, 2000, False), (c0, min(L0 - 3000, 1000 + 150000), 2000, False)],
                        "split_far_" + tag)
    return recs


def simulate_haplotype(seed, genome, contigs, events_by_contig, median_aln=30000, mean_m=2000, tag="h",
                       with_splits=True, n_filtered=4):
    """All records of one haplotype BAM, coordinate-sorted.  Returns list of record dicts."""
    rng = np.random.default_rng(seed)
    names = [c[0] for c in contigs]
    lengths = [c[1] for c in contigs]
    recs = []
    for tid, (name, clen) in enumerate(contigs):
        ref = genome[name]
        events = events_by_contig.get(name, [])
        pos, k = int(rng.integers(0, 500)), 0
        while pos < clen - 2000:
            span = int(np.exp(rng.normal(np.log(median_aln), 0.6))) + 1000
            end = min(clen, pos + span)
            # do not cut through an event
            for e in events:
                lo, hi = e.pos - 60, e.pos + (e.length if e.kind == "D" else 0) + 60
                if lo <= end <= hi:
                    end = min(clen, hi + 100)
            clip = (int(rng.integers(1, 500)) if rng.random() < 0.3 else 0,
                    int(rng.integers(1, 500)) if rng.random() < 0.3 else 0)
            words, seq = _tile_alignment(rng, ref, pos, end, events, mean_m, 0.001, clip)
            mapq = 60 if rng.random() > 0.03 else int(rng.integers(0, 20))
            recs.append(dict(qname="%s_ctg_%s_%d" % (tag, name, k), flag=0, tid=tid, pos=pos, mapq=mapq,
                             cigar=words, seq=seq, sa=None))
            k += 1
            pos = end + int(rng.integers(0, 300))
    if with_splits:
        recs += split_read_zoo(rng, genome, names, lengths, tag)
    # records the filters must drop: secondary, unmapped-flagged, low MAPQ with an SV-sized indel
    for i in range(n_filtered):
        name, clen = contigs[i % len(contigs)]
        p = int(rng.integers(1000, clen - 5000))
        words = np.array([(1000 << 4) | 0, (200 << 4) | 2, (1000 << 4) | 0], dtype=np.uint32)
        seq = np.concatenate((genome[name][p:p + 1000], genome[name][p + 1200:p + 2200]))
        flag, mapq = [(256, 60), (4, 60), (0, 5), (2048, 3)][i % 4]
        recs.append(dict(qname="%s_filtered_%d" % (tag, i), flag=flag, tid=names.index(name), pos=p, mapq=mapq,
                         cigar=words, seq=seq, sa=None))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    return recs


def write_bam(path, contigs, records, level=1):
    names = [c[0] for c in contigs]
    lengths = [c[1] for c in contigs]
    blobs = []
    for r in records:
        tags = [("SA", "Z", r["sa"])] if r.get("sa") else []
        seq = r["seq"]
        blobs.append(bamio.encode_record(r["qname"], r["flag"], r["tid"], r["pos"], r["mapq"], r["cigar"],
                                         None, tags, seq_packed=bamio.encode_seq(seq.tobytes()), l_seq=len(seq)))
    bamio.write_bam(path, names, lengths, blobs, level=level)


def payload_digest(path, chunk_members=4096):
    """SHA-256 of the UNCOMPRESSED content of a BGZF file (the inflated members in file order): the identity
    of a generated BAM independent of the deflate implementation that wrote it (zlib builds differ in the
    compressed bytes they produce for the same input, never in what those bytes inflate to)."""
    import hashlib
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    with open(path, "rb") as fh:
        raw = fh.read()
    spans = bamio._bgzf_block_spans(raw)
    view = memoryview(raw)
    h = hashlib.sha256()

    def inflate(span):
        st, ln, isz = span[:3]
        return zlib.decompress(view[st:st + ln], -15) if isz else b""
    with ThreadPoolExecutor(min(8, __import__("os").cpu_count() or 1)) as ex:
        for lo in range(0, len(spans), chunk_members):
            for part in ex.map(inflate, spans[lo:lo + chunk_members]):
                h.update(part)
    return h.hexdigest()


def file_digest(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for block in iter(lambda: fh.read(1 << 24), b""):
            h.update(block)
    return h.hexdigest()


def write_dataset(outdir, seed=1, contigs=CONFIG1_CONTIGS, diploid=True, n_shared=25, n_private=6,
                  median_aln=30000, mean_m=2000, dense_cluster=True, with_splits=True, min_gap=1500, level=1):
    """FASTA + one or two haplotype BAMs under `outdir`; returns their paths."""
    import os
    os.makedirs(outdir, exist_ok=True)
    genome = random_genome(seed, contigs)
    fasta = os.path.join(outdir, "ref.fa")
    write_fasta(fasta, [c[0] for c in contigs], [genome[c[0]] for c in contigs])
    rng = np.random.default_rng(seed + 1000)
    shared = {name: make_events(rng, length, n_shared, min_gap=min_gap) for name, length in contigs}
    # a knot of six small deletions within 1 kb near the end of the first contig, identical in
    # both haplotypes: 12 candidates in one partition, which the pairing step drops (> 10)
    knot = []
    if dense_cluster:
        kname, klen = contigs[0]
        knot = [Event(klen - 4500 + 170 * i, "D", 45 + i) for i in range(6)]
    bams = []
    for h in range(2 if diploid else 1):
        hrng = np.random.default_rng(seed + 2000 + h)
        events = {name: derive_haplotype_events(hrng, shared[name], n_private, length) for name, length in contigs}
        if knot:
            events[contigs[0][0]] = sorted(events[contigs[0][0]] + knot, key=lambda e: e.pos)
        recs = simulate_haplotype(seed + 3000 + h, genome, contigs, events, median_aln, mean_m, tag="h%d" % (h + 1),
                                  with_splits=with_splits)
        path = os.path.join(outdir, "hap%d.bam" % (h + 1))
        write_bam(path, contigs, recs, level=level)  # (deflate level of the BGZF members; the records do not depend on it)
        bams.append(path)
    return fasta, bams
