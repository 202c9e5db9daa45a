"""Shared test helpers: random alignment records, adapters to the product's / the reference's
object models, canonical candidate tuples."""
import argparse

import numpy as np

CODES = "MIDNSHP=XB"


def options(**kw):
    o = argparse.Namespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, query_gap_tolerance=50,
                           query_overlap_tolerance=50, reference_gap_tolerance=50, reference_overlap_tolerance=50,
                           partition_max_distance=1000, max_edit_distance=200, sample="Sample",
                           types="DEL,INS,INV,DUP:TANDEM,DUP:INT,BND", symbolic_alleles=False,
                           tandem_duplications_as_insertions=False, interspersed_duplications_as_insertions=False,
                           query_names=False, device=0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def cigar_string(tuples):
    return "".join("%d%s" % (l, CODES[o]) for o, l in tuples)


def random_cigar(rng, n_ops, sv_rate=0.15, clip=True, hard=False):
    ops = []
    if clip and rng.random() < 0.4:
        ops.append((5 if hard and rng.random() < 0.5 else 4, int(rng.integers(1, 3000))))
    for _ in range(n_ops):
        r = rng.random()
        if r < 0.55:
            ops.append((int(rng.choice([0, 7, 8])), int(rng.integers(1, 3000))))
        elif r < 0.9:
            ln = int(rng.integers(30, 600)) if rng.random() < sv_rate else int(rng.integers(1, 45))
            ops.append((int(rng.integers(1, 3)), ln))
        else:
            ops.append((int(rng.choice([3, 6, 9])), int(rng.integers(1, 500))))
    if clip and rng.random() < 0.4:
        ops.append((4, int(rng.integers(1, 3000))))
    if not any(o in (0, 7, 8) for o, _ in ops):
        ops.insert(1 if ops and ops[0][0] in (4, 5) else 0, (0, int(rng.integers(1, 500))))
    return ops


def query_len(tuples):
    return sum(l for o, l in tuples if o in (0, 1, 4, 7, 8))


def random_sa(rng, names, lengths, read_len):
    """SA tag with segments near interesting thresholds + occasional malformed entries."""
    parts = []
    for _ in range(int(rng.integers(1, 5))):
        name_i = int(rng.integers(0, len(names)))
        ln = int(rng.integers(200, max(201, min(read_len, 4000))))
        before = int(rng.integers(0, max(1, read_len - ln)))
        after = max(0, read_len - before - ln)
        mid = [(0, ln)]
        if rng.random() < 0.3:
            a = ln // 2
            mid = [(0, a), (int(rng.integers(1, 3)), int(rng.integers(1, 60))), (0, ln - a)]
        cig = ([(4, before)] if before else []) + mid + ([(4, after)] if after else [])
        pos = int(rng.integers(1, max(2, lengths[name_i] - ln)))
        mapq = int(rng.choice([60, 60, 60, 30, 19, 20, 0, -400, 300]))
        entry = "%s,%d,%s,%s,%d,%d" % (names[name_i], pos, rng.choice(["+", "-"]), cigar_string(cig), mapq,
                                       int(rng.integers(0, 50)))
        r = rng.random()
        if r < 0.05:
            entry += ",extra"
        elif r < 0.08:
            entry = ",".join(entry.split(",")[:5])
        parts.append(entry)
    return ";".join(parts) + (";" if rng.random() < 0.8 else "")


def random_records(rng, names, lengths, n, split_frac=0.4):
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for i in range(n):
        tid = int(rng.integers(0, len(names)))
        cig = random_cigar(rng, int(rng.integers(1, 40)), hard=rng.random() < 0.1)
        flag = int(rng.choice([0, 0, 0, 16, 2048, 2064, 256, 4, 272]))
        mapq = int(rng.choice([60, 60, 60, 20, 19, 0]))
        qlen = query_len(cig)
        seq = bases[rng.integers(0, 4, size=qlen)].tobytes().decode()
        sa = None
        if rng.random() < split_frac:
            hard = sum(l for o, l in cig if o == 5)
            sa = random_sa(rng, names, lengths, qlen + hard)
        recs.append(dict(qname="read%d" % i, flag=flag, tid=tid, pos=int(rng.integers(0, lengths[tid])), mapq=mapq,
                         cigar=cig, seq=seq, sa=sa))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    return recs


def engineered_split_records(rng, names, lengths, n):
    """Primaries whose SA segments abut the primary on the read and sit near every tolerance on
    the reference, so that all branch families fire (INS, DEL, BND, tandem, DUP_INT, INV)."""
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    deltas = [-200000, -100001, -5000, -600, -51, -50, -49, -41, -40, 0, 39, 40, 41, 49, 50, 51, 300, 5000, 100001]
    for i in range(n):
        tid = int(rng.integers(0, len(names)))
        k = int(rng.integers(2, 5))
        seg_len = [int(rng.integers(300, 3000)) for _ in range(k)]
        gaps = [int(rng.choice([-60, -51, -50, -10, 0, 0, 0, 30, 50, 51, 100, 400])) for _ in range(k - 1)]
        q = [0]
        for j in range(k - 1):
            q.append(max(0, q[-1] + seg_len[j] + gaps[j]))
        read_len = max(a + b for a, b in zip(q, seg_len))
        pos0 = int(rng.integers(200000, lengths[tid] - 300000)) if lengths[tid] > 600000 else int(rng.integers(0, max(1, lengths[tid] // 2)))
        strands = [bool(rng.random() < 0.3) for _ in range(k)]
        refs, p = [], pos0
        for j in range(k):
            t = tid if rng.random() < 0.85 else int(rng.integers(0, len(names)))
            p = max(0, p + int(rng.choice(deltas)) + (seg_len[j - 1] if j else 0))
            refs.append((t, min(p, max(0, lengths[t] - 1))))

        if k == 3 and rng.random() < 0.3:   # A → elsewhere → back to where A ended: interspersed duplication
            rev = bool(rng.random() < 0.3)
            strands = [rev, rev, rev]
            t2 = int(rng.integers(0, len(names)))
            p2 = int(rng.integers(0, max(1, lengths[t2] - 5000)))
            jitter = int(rng.choice([-25, -19, -5, 0, 0, 7, 19, 21]))
            if not rev:
                refs = [(tid, pos0), (t2, p2), (tid, pos0 + seg_len[0] + jitter)]
            else:
                refs = [(tid, pos0 + seg_len[2] + jitter), (t2, p2), (tid, pos0)]
            q = [0, seg_len[0], seg_len[0] + seg_len[1]]
            read_len = sum(seg_len)

        def cig(j):
            before, after = q[j], read_len - q[j] - seg_len[j]
            if strands[j]:
                before, after = after, before
            return ([(4, before)] if before else []) + [(0, seg_len[j])] + ([(4, after)] if after else [])

        pi = int(rng.integers(0, k))
        seq = bases[rng.integers(0, 4, size=read_len)].tobytes().decode()
        sa = "".join("%s,%d,%s,%s,60,0;" % (names[refs[j][0]], refs[j][1] + 1, "-" if strands[j] else "+",
                                            cigar_string(cig(j))) for j in range(k) if j != pi)
        recs.append(dict(qname="split%d" % i, flag=16 if strands[pi] else 0, tid=refs[pi][0], pos=refs[pi][1], mapq=60,
                         cigar=cig(pi), seq=seq, sa=sa))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    return recs


# ---- adapters ------------------------------------------------------------------------------
class FakeBam(object):
    """pysam.AlignmentFile-like view over record dicts, serving the product's record class."""

    def __init__(self, names, lengths, records):
        from svim_asm_amd import bamio
        self.references = tuple(names)
        self.lengths = tuple(lengths)
        self._recs = []
        for r in records:
            a = bamio.AlignedRecord()
            a.query_name = r["qname"]
            a.flag = r["flag"]
            a.reference_id = r["tid"]
            a.reference_start = r["pos"]
            a.mapping_quality = r["mapq"]
            a.cigar_words = np.array([(l << 4) | o for o, l in r["cigar"]], dtype=np.uint32)
            a._seq_str = r["seq"]
            a._l_seq = len(r["seq"])
            a._tags = {"SA": r["sa"]} if r.get("sa") is not None else {}
            self._recs.append(a)

    def fetch(self, contig=None, until_eof=False):
        if contig is None:
            return iter(self._recs)
        tid = self.get_tid(contig)
        return iter([a for a in self._recs if a.reference_id == tid])

    def get_tid(self, name):
        return self.references.index(name) if name in self.references else -1

    def get_reference_name(self, tid):
        if tid < 0 or tid >= len(self.references):
            raise ValueError("reference_id %i out of range" % tid)
        return self.references[tid]

    getrname = get_reference_name

    def get_reference_length(self, name):
        return self.lengths[self.references.index(name)]


class FakeFasta(object):
    def __init__(self, seqs):
        self._s = seqs
        self.references = list(seqs)
        self.lengths = [len(v) for v in seqs.values()]

    def fetch(self, ref, start=None, end=None):
        s = self._s[ref]
        start = 0 if start is None else start
        end = len(s) if end is None else min(end, len(s))
        return s[start:end] if start < end else ""

    def get_reference_length(self, ref):
        return len(self._s[ref])

    def close(self):
        pass


def candidate_tuple(c):
    t = c.type
    if t == "DEL":
        return ("DEL", c.source_contig, c.source_start, c.source_end, tuple(c.reads), c.genotype)
    if t == "INS":
        return ("INS", c.dest_contig, c.dest_start, c.dest_end, tuple(c.reads), c.sequence, c.genotype)
    if t == "INV":
        return ("INV", c.source_contig, c.source_start, c.source_end, tuple(c.reads), bool(c.complete), c.genotype)
    if t == "DUP_TAN":
        return ("DUP_TAN", c.source_contig, c.source_start, c.source_end, int(c.copies), bool(c.fully_covered),
                tuple(c.reads), c.genotype)
    if t == "DUP_INT":
        return ("DUP_INT", c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end,
                tuple(c.reads), bool(c.cutpaste), c.genotype)
    return ("BND", c.source_contig, c.source_start, c.source_direction, c.dest_contig, c.dest_start,
            c.dest_direction, tuple(c.reads), c.genotype)


def build_candidate(tup, bam, cls):
    """Canonical tuple → Candidate object of module `cls` (product or reference SVCandidate)."""
    t = tup[0]
    if t == "DEL":
        return cls.CandidateDeletion(tup[1], tup[2], tup[3], list(tup[4]), bam, tup[5])
    if t == "INS":
        return cls.CandidateInsertion(tup[1], tup[2], tup[3], list(tup[4]), tup[5], bam, tup[6])
    if t == "INV":
        return cls.CandidateInversion(tup[1], tup[2], tup[3], list(tup[4]), tup[5], bam, tup[6])
    if t == "DUP_TAN":
        return cls.CandidateDuplicationTandem(tup[1], tup[2], tup[3], tup[4], tup[5], list(tup[6]), bam, tup[7])
    if t == "DUP_INT":
        return cls.CandidateDuplicationInterspersed(tup[1], tup[2], tup[3], tup[4], tup[5], tup[6], list(tup[7]), bam,
                                                    tup[8], tup[9])
    return cls.CandidateBreakend(tup[1], tup[2], tup[3], tup[4], tup[5], tup[6], list(tup[7]), bam, tup[8])


def random_candidates(rng, names, lengths, seqs, n, hap_tag):
    """Random candidate tuples of all six types, clustered so that partitions of size 1..12 arise."""
    from oracle import svim_oracle as O
    lens = dict(zip(names, lengths))
    out = []
    anchors = [(names[int(rng.integers(0, len(names)))], int(rng.integers(500, 20000))) for _ in range(max(3, n // 6))]
    for i in range(n):
        contig, anchor = anchors[int(rng.integers(0, len(anchors)))]
        pos = max(0, min(lens[contig] - 10, anchor + int(rng.integers(-700, 700))))
        ln = int(rng.integers(40, 400))
        reads = ["%s_r%d" % (hap_tag, i)]
        t = rng.choice(["DEL", "DEL", "INS", "INS", "INV", "DUP_TAN", "DUP_INT", "BND"])
        if t == "DEL":
            out.append(O.cand_del(contig, pos, pos + ln, reads, lens))
        elif t == "INS":
            seq = "".join(rng.choice(list("ACGTacgtN"), size=ln))
            out.append(O.cand_ins(contig, pos, pos + ln, reads, seq, lens))
        elif t == "INV":
            out.append(O.cand_inv(contig, pos, pos + ln, reads, bool(rng.random() < 0.5), lens))
        elif t == "DUP_TAN":
            out.append(O.cand_tan(contig, pos, pos + ln, int(rng.integers(1, 5)), bool(rng.random() < 0.5), reads, lens))
        elif t == "DUP_INT":
            c2 = names[int(rng.integers(0, len(names)))]
            p2 = int(rng.integers(0, lens[c2] - 500))
            out.append(O.cand_int(c2, p2, p2 + ln, contig, pos, pos + ln, reads, lens, bool(rng.random() < 0.3)))
        else:
            c2 = names[int(rng.integers(0, len(names)))]
            p2 = max(0, min(lens[c2] - 1, anchor + int(rng.integers(-400, 400))))
            out.append(O.cand_bnd(contig, pos, rng.choice(["fwd", "rev"]), c2, p2, rng.choice(["fwd", "rev"]), reads, lens))
    return out


def constructed_again(tuples, lens):
    """Candidate tuples as they are after one more pass through their constructor.  The differential harnesses hand
    the product (or the reference) objects built from the tuples with `build_candidate` — the tuples' second
    construction, random_candidates made them with the oracle's constructors — and the oracle the tuples themselves.
    Every constructor is idempotent except CandidateBreakend on a breakend whose two ends are the same position: it
    swaps the ends and flips both directions every time (SVCandidate.py:352-373, `<` on both comparisons), so the
    oracle's side must have been through it as often as the other side."""
    from oracle import svim_oracle as O
    out = []
    for t in tuples:
        if t[0] == "BND":
            t = O.cand_bnd(t[1], t[2], t[3], t[4], t[5], t[6], t[7], lens, t[8])
        out.append(t)
    return out


def run_cli_ranks(argv, world_size, timeout=600):
    """Run `svim-asm <argv>` as `world_size` fresh processes (one rank each, all on HIP device 0 — the
    GPU box has one GPU), the way torch.distributed.run would start them.  Every child makes its own
    first GPU call.  Returns [(returncode, combined output)] by rank."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    code = "import sys; sys.path.insert(0, %r); from svim_asm_amd import cli; cli.main(%r)" % (root, list(argv))
    procs = []
    for rank in range(world_size):
        env = dict(os.environ)
        env.update(RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    out = []
    for p in procs:
        try:
            text, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            text, _ = p.communicate()
            text += "\n[timeout]"
        out.append((p.returncode, text))
    return out


def assert_inputs_are_the_golden_ones(meta, files):
    """The regenerated BAM / FASTA inputs must be the ones the REAL reference was run on when the golden VCF
    was made: compared through digests of their UNCOMPRESSED content (synth_bam.payload_digest — independent of
    the zlib build that compressed them).  A difference is a failure, never a skip: a golden that silently
    drops out of the suite checks nothing."""
    import os
    from svim_asm_amd import synth_bam
    for f in files:
        name = os.path.basename(f)
        got = synth_bam.payload_digest(f) if name.endswith(".bam") else synth_bam.file_digest(f)
        assert got == meta["payload_sha256"][name], \
            "%s: regenerated content differs from the input of the golden run (generator drift): %s" % (name, got)


# ---- reference-generated function vectors (tests/golden/pipeline_vectors.json.gz) --------------
def _tuplify(x):
    return tuple(_tuplify(v) for v in x) if isinstance(x, list) else x


def load_pipeline_vectors():
    """COLLECT / analyze_read_segments / form_partitions / pair_candidates vectors written by the real
    reference (oracle/make_golden.py pipeline).  JSON lists become the tuples the comparisons use."""
    import gzip
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_vectors.json.gz")
    with gzip.open(path, "rt") as fh:
        vec = json.load(fh)
    for c in vec["collect"]:
        for r in c["records"]:
            r["cigar"] = [tuple(t) for t in r["cigar"]]
        c["out"] = [_tuplify(t) for t in c["out"]]
        for pr in c["analyze_read_segments"]:
            pr["out"] = [_tuplify(t) for t in pr["out"]]
    for p in vec["pair"]:
        p["t1"] = [_tuplify(t) for t in p["t1"]]
        p["t2"] = [_tuplify(t) for t in p["t2"]]
        p["out"] = [_tuplify(t) for t in p["out"]]
    return vec


# ---- the device replaced by the oracle (CPU-side tests of the host logic) -------------------------
class OracleBackedContext(object):
    """Stands in for svim_asm_amd._svxlib().Context with every device entry point answered by the CPU oracle, so
    that the HOST logic of the product (record gathering, candidate assembly, haplotype recipes, clustering
    glue, VCF) can be checked where there is no GPU.  TEST INFRASTRUCTURE: the product never sees it (tests
    patch _svxlib().default_context); the GPU tests make the same comparisons with the real kernels."""

    def cigar_extract(self, cigar, aln_off, ref_start=None, min_len=40, cap=None, op=None):
        return _orc().cigar_extract(cigar, aln_off, ref_start, min_len)

    def cigar_stats(self, cigar, aln_off):
        return _orc().cigar_stats(cigar, aln_off)

    def segments_classify(self, segs, read_off, read_len, params):
        prm = [getattr(params, f) for f, _ in params._fields_]
        return _orc().segments_classify(np.ascontiguousarray(segs).view(_orc().SEG_DTYPE), read_off, read_len, prm).view(_svxlib().RAW_DTYPE)

    def pair_partition(self, keys, max_dist):
        return _orc().pair_partition(keys, max_dist)

    def collect_batch(self, *args, **kw):
        # the product's own composition of the single-purpose calls, each answered by the oracle above / below
        return _svxlib().Context.collect_batch_composed(self, *args, **kw)

    def edit_distance_batch(self, seq, a_off, a_len, b_off, b_len, k_max=0xFFFFFFFF):
        seq = np.ascontiguousarray(seq, np.uint8)
        out = []
        for ao, al, bo, bl in zip(a_off, a_len, b_off, b_len):
            d = _orc().edit_distance(seq[ao:ao + al].tobytes(), seq[bo:bo + bl].tobytes())
            out.append(d if d <= k_max else 0xFFFFFFFF)
        return np.array(out, dtype=np.uint32)


    def segments_postpass(self, raw, read_off, contig_rank, params):
        from oracle import svim_oracle
        prm = [getattr(params, f) for f, _ in params._fields_]
        code = {"TANDEM": 1, "DUP_INT": 2, "INV": 3}
        recs, first = [], [0]
        for r in range(len(read_off) - 1):
            rows = [tuple(int(x[k]) for k in ("kind", "a0", "a1", "a2", "a3", "a4", "a5")) for x in raw[read_off[r]:read_off[r + 1]]]
            for t in svim_oracle.postpass_records(rows, list(contig_rank), prm[0], prm[1]):
                recs.append(tuple([code[t[0]]] + [int(v) for v in t[1:]] + [0] * (8 - len(t))))
            first.append(len(recs))
        return np.array(recs, dtype=_svxlib().RAW_DTYPE) if recs else np.zeros(0, dtype=_svxlib().RAW_DTYPE), np.array(first, np.int64)

    def resident(self, host_bytes):
        return host_bytes  # (no device here: the "resident" pool is the array itself)

    def haplotype_distance_batch(self, pool, pieces, k_max=0xFFFFFFFF):
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}

        def build(three):
            out = []
            for off, ln, rep, flags in three:
                s = bytes(pool[off:off + ln]).decode("latin-1")
                if flags & 1:
                    s = s.upper()
                if flags & 2:
                    s = "".join(comp.get(b, b) for b in reversed(s))
                out.append(s * rep)
            return "".join(out).encode("latin-1")
        out = []
        for p in range(len(pieces) // 6):
            d = _orc().edit_distance(build(pieces[p * 6:p * 6 + 3].tolist()), build(pieces[p * 6 + 3:p * 6 + 6].tolist()))
            out.append(d if d <= k_max else 0xFFFFFFFF)
        return np.array(out, dtype=np.uint32)

    def haplotype_distance_batch_mixed(self, pool, pieces, k_max):
        exact = self.haplotype_distance_batch(pool, pieces)
        k_max = np.asarray(k_max, dtype=np.uint32)
        return np.where(exact <= k_max, exact, np.uint32(0xFFFFFFFF)).astype(np.uint32)

    def linkage_cut_batch(self, dist, n_members, cutoff):
        out, at = [], 0
        for n in n_members:
            m = n * (n - 1) // 2
            out.extend(_orc().linkage_cut(dist[at:at + m], n, cutoff).tolist() if n > 1 else [1] * n)
            at += m
        return np.array(out, dtype=np.uint32)


def _orc():
    from oracle import orc
    return orc


def _svxlib():
    from svim_asm_amd import _lib
    return _lib


def oracle_backed_device(monkeypatch):
    """Patch the product so that every kernel call is answered by the oracle (pytest monkeypatch)."""
    ctx = OracleBackedContext()
    monkeypatch.setattr(_svxlib(), "default_context", lambda device=0: ctx)
    monkeypatch.setattr(_svxlib(), "new_context", lambda device=0: OracleBackedContext())  # (svim-asm-cohort's other workers)
    return ctx


# ---- COLLECT of one submission by the checker alone (tests/test_gpu_collect.py)
_POST_CODE = {1: "TANDEM", 2: "DUP_INT", 3: "INV"}
_POST_WIDTH = {"TANDEM": 5, "DUP_INT": 6, "INV": 4}


def post_records_as_tuples(post, first):
    """The packed derived records of svx_segments_postpass / svx_collect_batch as the tuples of
    oracle/svim_oracle.postpass_records, one list per read."""
    out = []
    for r in range(len(first) - 1):
        recs = []
        for rec in post[int(first[r]):int(first[r + 1])]:
            kind = _POST_CODE[int(rec["kind"])]
            vals = [int(rec[n]) for n in ("a0", "a1", "a2", "a3", "a4", "a5")][:_POST_WIDTH[kind]]
            if kind == "TANDEM":
                vals[4] = bool(vals[4])
            if kind == "INV":
                vals[3] = bool(vals[3])
            recs.append((kind,) + tuple(vals))
        out.append(recs)
    return out


def oracle_collect(cigar_parts, aln_off, ref_start, min_len, extra_cigar, extra_off, seg_src, seg_tid, seg_pos, seg_rev,
                   seg_qend, read_off, contig_rank, params):
    """What svx_collect_batch must return, from the oracle only: signatures (orc.cigar_extract, SVIM_intra.py:8-44),
    segment rows from the oracle's CIGAR statistics (orc.segment_rows, SVIM_inter.py:66-81), the decision tree
    (orc.segments_classify, :83-258) and the record-level post-passes (svim_oracle.postpass_records, :260-338).
    Returns (sig, raw, post tuples per read)."""
    from oracle import orc, svim_oracle
    cigar = np.concatenate(cigar_parts) if len(cigar_parts) else np.zeros(0, np.uint32)
    sig = orc.cigar_extract(cigar, aln_off, ref_start, min_len)
    n_reads = len(read_off) - 1 if len(read_off) else 0
    if n_reads <= 0 or len(seg_src) == 0:
        return sig, np.zeros(0, dtype=orc.RAW_DTYPE), [[] for _ in range(max(n_reads, 0))]
    n_ops = int(aln_off[-1])
    off_all = np.concatenate((np.asarray(aln_off, np.uint64), n_ops + np.asarray(extra_off, np.uint64)[1:])).astype(np.uint64)
    st = orc.cigar_stats(np.concatenate((cigar, np.asarray(extra_cigar, np.uint32))), off_all)
    segs, read_len = orc.segment_rows(st, seg_src, seg_tid, seg_pos, seg_rev, seg_qend, read_off)
    raw = orc.segments_classify(segs, read_off, read_len, [int(x) for x in params])
    post = []
    rank = [int(x) for x in contig_rank]
    for r in range(n_reads):
        rows = [tuple(int(raw[j][n]) for n in ("kind", "a0", "a1", "a2", "a3", "a4", "a5"))
                for j in range(int(read_off[r]), int(read_off[r + 1]))]
        post.append(svim_oracle.postpass_records(rows, rank, int(params[0]), int(params[1])))
    return sig, raw, post
