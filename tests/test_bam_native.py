"""Native BAM ingest (libsvx.so svx_bam_*, include/svx_bam.h): the columnar walk must give exactly
what the pure-Python reader and the independent oracle-side stub reader give, through an index
(parallel, per-contig) and without one (sequential), with zlib and with libdeflate, on long
CIGARs stored in CG:B,I, and must reject corrupt input.  CPU only (host code)."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

from oracle import run_oracle
from svim_asm_amd import bamio, synth_bam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
COLS = ("tid", "pos", "mapq", "flag", "n_cig", "l_seq", "ref_len", "voffset")


def assert_same_columns(a, b):
    assert len(a) == len(b)
    for k in COLS:
        assert np.array_equal(a._cols[k], b._cols[k]), k
    assert np.array_equal(a._cigar, b._cigar) and np.array_equal(a._cig_off, b._cig_off)
    for i in range(len(a)):
        ra, rb = a.record(i), b.record(i)
        assert ra.query_name == rb.query_name and bytes(ra._tags_raw) == bytes(rb._tags_raw)
        assert ra.has_tag("SA") == rb.has_tag("SA")
        if ra.has_tag("SA"):
            assert ra.get_tag("SA") == rb.get_tag("SA")


def assert_matches_stub(f, path):
    """Against oracle/refstub/pysam.py, which shares no code with the product's readers."""
    recs, names, lengths = run_oracle.read_records(path)
    assert list(f.references) == names and list(f.lengths) == lengths and len(f) == len(recs)
    for i, r in enumerate(recs):
        a = f.record(i)
        assert (a.query_name, a.flag, a.reference_id, a.reference_start, a.mapping_quality) == \
            (r["qname"], r["flag"], r["tid"], r["pos"], r["mapq"])
        assert a.cigartuples == (r["cigar"] or None)
        assert (a.get_tag("SA") if a.has_tag("SA") else None) == r["sa"]
        assert a.seq_slice(0, a._l_seq) == r["seq"]


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("bamnative"))
    contigs = (("chrA", 500000), ("chrB", 300000), ("chrC", 200000), ("chrD", 120000))
    fa, bams = synth_bam.write_dataset(d, seed=11, contigs=contigs, n_shared=12, n_private=3, median_aln=90000)
    return d, bams


@pytest.mark.parametrize("fn", ["chimeric_read.bam", "chimeric_read_errors.bam", "config1/hap1.bam", "config1/hap2.bam"])
def test_native_reader_on_fixtures(fn):
    """The reference's own samtools-written BAMs (no index: sequential walk) and the config-1 BAMs."""
    path = os.path.join(GOLD, fn)
    nat = bamio.AlignmentFile(path)
    assert nat._h is not None
    assert_same_columns(bamio.AlignmentFile(path, reader="python"), nat)
    assert_matches_stub(nat, path)
    assert nat.index_state() == (1 if fn.startswith("config1") else 0)


def test_index_walk_equals_sequential_walk(dataset, tmp_path):
    d, bams = dataset
    indexed = bamio.AlignmentFile(bams[0], threads=5)
    assert indexed.index_state() == 1
    bare = str(tmp_path / "noindex.bam")
    shutil.copy(bams[0], bare)
    seq = bamio.AlignmentFile(bare, threads=5)
    assert seq.index_state() == 0
    with pytest.raises(ValueError):
        seq.check_index()
    assert len(indexed) > 0
    # members that lie wholly inside SEQ/QUAL stay compressed
    assert indexed.blocks_inflated < indexed.blocks_spanned
    assert_same_columns(seq, indexed)
    assert_same_columns(bamio.AlignmentFile(bams[0], reader="python"), indexed)
    assert_matches_stub(indexed, bams[0])


def test_per_contig_load_walks_only_that_contig(dataset):
    d, bams = dataset
    full = bamio.AlignmentFile(bams[1])
    tid_all = full._cols["tid"]
    spans = full.contig_spans()
    assert (spans > 0).all() and spans[0] > spans[3]
    for sel in (["chrC"], ["chrA", "chrD"], [1]):
        part = bamio.AlignmentFile(bams[1])
        part.load(sel)
        tids = [full.get_tid(c) if isinstance(c, str) else c for c in sel]
        keep = np.nonzero(np.isin(tid_all, tids))[0]
        assert len(part) == len(keep) > 0
        assert np.array_equal(part._cols["voffset"], full._cols["voffset"][keep])
        cig, off, pos, tid = full.batch(keep)
        assert np.array_equal(part._cigar, cig)
        assert part.blocks_spanned < full.blocks_spanned
        # a contig that was not loaded is loaded on demand
        other = next(n for n in full.references if full.get_tid(n) not in tids)
        assert [r.query_name for r in part.fetch(other)] == [r.query_name for r in full.fetch(other)]
    # shares of the file add up: every rank of a 2-way plan walks about its half
    a, b = bamio.AlignmentFile(bams[1]), bamio.AlignmentFile(bams[1])
    a.load(["chrA"]); b.load(["chrB", "chrC", "chrD"])
    assert len(a) + len(b) == int((tid_all >= 0).sum())
    assert abs((a.blocks_spanned + b.blocks_spanned) - full.blocks_spanned) <= 8


def test_sequence_slices_batch(dataset):
    d, bams = dataset
    nat, py = bamio.AlignmentFile(bams[0]), bamio.AlignmentFile(bams[0], reader="python")
    rng = np.random.default_rng(5)
    rec = np.sort(rng.integers(0, len(nat), 300))
    a = np.array([int(rng.integers(0, max(1, nat._cols["l_seq"][r]))) for r in rec])
    b = a + rng.integers(0, 4000, len(rec))
    b[::7] = a[::7]                      # empty slices
    b[::11] += 10_000_000                # clipped at l_seq
    got = nat.sequence_slices(rec, a, b)
    exp = [py.record(int(r)).seq_slice(int(x), int(y)) for r, x, y in zip(rec, a, b)]
    assert got == exp
    assert nat.sequence_slices([], [], []) == []
    r = nat.record(int(rec[0]))
    assert r.seq_slice(3, 77) == py.record(int(rec[0])).seq_slice(3, 77)     # odd start: low nibble first


def _long_cigar(rng, n_pairs):
    ops = []
    for _ in range(n_pairs):
        ops.append((int(rng.integers(1, 20)) << 4) | 0)
        ops.append((int(rng.integers(1, 60)) << 4) | int(rng.integers(1, 3)))
    ops.append((5 << 4) | 0)
    return np.array(ops, dtype=np.uint32)


def test_long_cigar_in_cg_tag(tmp_path):
    """> 65535 operations: stored as `<l_seq>S<ref_len>N` + CG:B,I (SAM spec §4.2.2); every reader
    restores the real CIGAR and hides the tag, as htslib's bam_tag2cigar does."""
    rng = np.random.default_rng(0)
    cw = _long_cigar(rng, 70000)
    qlen = int(((cw >> 4) * np.isin(cw & 15, [0, 1, 4])).sum())
    rlen = int(((cw >> 4) * np.isin(cw & 15, [0, 2])).sum())
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, qlen)].tobytes().decode()
    small = np.array([(50 << 4) | 0, (45 << 4) | 2, (50 << 4) | 0], dtype=np.uint32)
    blobs = [bamio.encode_record("short", 0, 0, 100, 60, small, "A" * 100),
             bamio.encode_record("long", 0, 0, 500, 60, cw, seq, [("NM", "i", 7), ("SA", "Z", "chr2,5,+,10M,60,0;")]),
             # a placeholder-looking CIGAR without a CG tag stays what it is
             bamio.encode_record("fake", 0, 0, 900, 60, np.array([(30 << 4) | 4, (10 << 4) | 3], dtype=np.uint32), "C" * 30),
             bamio.encode_record("after", 0, 1, 7, 60, np.array([(30 << 4) | 0], dtype=np.uint32), "C" * 30)]
    path = str(tmp_path / "long.bam")
    bamio.write_bam(path, ["chr1", "chr2"], [rlen + 1000, 5000], blobs)
    raw = bamio.bgzf_decompress(path)
    assert raw.count(b"CGBI") == 1
    # the stored record really carries the 2-operation placeholder
    at = raw.index(b"long\x00")
    n_cig_stored = struct.unpack_from("<H", raw, at - 32 + 12)[0]
    assert n_cig_stored == 2
    nat, py = bamio.AlignmentFile(path), bamio.AlignmentFile(path, reader="python")
    assert nat.index_state() == 1
    for f in (nat, py):
        assert list(f._cols["n_cig"]) == [3, len(cw), 2, 1]
        assert int(f._cols["ref_len"][1]) == rlen
        r = f.record(1)
        assert np.array_equal(r.cigar_words, cw) and not r.has_tag("CG")
        assert r.get_tag("SA") == "chr2,5,+,10M,60,0;" and r.get_tag("NM") == 7
        assert r.reference_end == 500 + rlen and r.infer_read_length() == qlen
        assert r.seq_slice(qlen - 50, qlen) == seq[-50:]
        assert f.record(2).cigarstring == "30S10N"
    assert_same_columns(py, nat)
    assert_matches_stub(nat, path)


def test_zlib_path_equals_libdeflate_path(dataset):
    """SVX_BAM_ZLIB=1 forces zlib inflate (the fallback when libdeflate is not installed)."""
    d, bams = dataset
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
            "from svim_asm_amd import bamio\n"
            "f = bamio.AlignmentFile(%r)\n"
            "h = hashlib.sha256(f._cigar.tobytes() + f._cols['voffset'].tobytes() + f._aux_pool + f._names_pool)\n"
            "h.update(''.join(f.sequence_slices([0, 1, 2], [5, 0, 100], [5000, 77, 40000])).encode())\n"
            "print(h.hexdigest())\n" % (ROOT, bams[0]))
    outs = []
    for env in ({"SVX_BAM_ZLIB": "1"}, {"SVX_BAM_ZLIB": "0"}):
        e = dict(os.environ)
        e.update(env)
        outs.append(subprocess.run([sys.executable, "-c", code], env=e, check=True, capture_output=True, text=True).stdout)
    assert outs[0] == outs[1] and len(outs[0].strip()) == 64


def test_corrupt_input_is_rejected(dataset, tmp_path):
    d, bams = dataset
    data = open(bams[0], "rb").read()
    # not a BAM at all
    junk = str(tmp_path / "junk.bam")
    open(junk, "wb").write(b"hello world, definitely not gzip" * 10)
    with pytest.raises(ValueError):
        bamio.AlignmentFile(junk)
    with pytest.raises(FileNotFoundError):
        bamio.AlignmentFile(str(tmp_path / "missing.bam"))
    # truncated in the middle of a member
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(data[:len(data) // 2 + 123])
    with pytest.raises(ValueError):
        bamio.AlignmentFile(cut).load()
    # a flipped payload byte in a member that holds record headers: by default (whole members, CRC32 as under htslib)
    # ALWAYS an error — also when the damage lies in the unread tail of the member, behind the last byte a record walk
    # needs; with verify=False (the opt-out) an error when it lies inside the bytes the walk needs and the stream no
    # longer decodes, and the pristine file's records when it lies behind them
    spans = bamio._bgzf_block_spans(data)
    pristine = bamio.AlignmentFile(bams[0])
    member = int(pristine._cols["voffset"][2]) >> 16
    st, ln = next((sp[0], sp[1]) for sp in spans if sp[3] == member)
    outcomes = set()
    for k, where in enumerate((ln // 50, ln // 2, ln - 2)):
        bad = bytearray(data)
        bad[st + where] ^= 0x5A
        flip = str(tmp_path / ("flip%d.bam" % k))
        open(flip, "wb").write(bytes(bad))
        shutil.copy(bams[0] + ".bai", flip + ".bai")
        with pytest.raises(ValueError):
            bamio.AlignmentFile(flip).load()
        with pytest.raises(ValueError):
            bamio.AlignmentFile(flip, verify=True).load()
        try:
            got = bamio.AlignmentFile(flip, verify=False).load()
        except ValueError:
            outcomes.add("refused")
        else:
            assert_same_columns(pristine, got)
            outcomes.add("unaffected")
    assert "refused" in outcomes or "unaffected" in outcomes
    # a stale index (belongs to another file): detected, sequential walk instead, same records
    stale = str(tmp_path / "stale.bam")
    shutil.copy(bams[0], stale)
    shutil.copy(bams[1] + ".bai", stale + ".bai")
    f = bamio.AlignmentFile(stale)
    assert_same_columns(bamio.AlignmentFile(bams[0]), f)
    # a stub index (32 bytes, no bins) on a file with records: unusable but present
    stub = str(tmp_path / "stub.bam")
    shutil.copy(bams[0], stub)
    open(stub + ".bai", "wb").write(b"BAI\x01" + struct.pack("<i", 4) + struct.pack("<ii", 0, 0) * 4)
    g = bamio.AlignmentFile(stub)
    assert g.index_state() == 2 and g.check_index()
    assert_same_columns(bamio.AlignmentFile(bams[0]), g)


def test_index_builder_roundtrip(dataset, tmp_path):
    """index_bam (our `samtools index`) on a file without index == the index the writer made."""
    d, bams = dataset
    p = str(tmp_path / "x.bam")
    shutil.copy(bams[0], p)
    bamio.index_bam(p)
    assert open(p + ".bai", "rb").read() == open(bams[0] + ".bai", "rb").read()
    assert bamio.AlignmentFile(p).index_state() == 1


@pytest.mark.parametrize("reader", ["native", "python"])
def test_decoding_against_samtools_own_text_rendering(reader):
    """chimeric_read_errors.sam is the SAM text of chimeric_read_errors.bam, both written by samtools and
    both shipped with the reference's tests: every field our readers decode from the BAM bytes (name, flag,
    contig, position, MAPQ, CIGAR, the 4-bit packed bases, the SA tag) must equal what samtools printed —
    a pin of the ingest that does not go through this repository's own BAM writer."""
    sam = [l.rstrip("\n").split("\t") for l in open(os.path.join(GOLD, "chimeric_read_errors.sam")) if not l.startswith("@")]
    f = bamio.AlignmentFile(os.path.join(GOLD, "chimeric_read_errors.bam"), reader=reader)
    header_names = [l.split("\t")[1][3:] for l in open(os.path.join(GOLD, "chimeric_read_errors.sam")) if l.startswith("@SQ")]
    assert list(f.references) == header_names
    assert len(f) == len(sam) > 0
    for i, fields in enumerate(sam):
        a = f.record(i)
        qname, flag, rname, pos, mapq, cigar, _rnext, _pnext, _tlen, seq = fields[:10]
        assert a.query_name == qname and a.flag == int(flag) and a.mapping_quality == int(mapq)
        assert f.get_reference_name(a.reference_id) == rname and a.reference_start == int(pos) - 1
        assert a.cigarstring == cigar
        assert a.seq_slice(0, a._l_seq) == seq and a._l_seq == len(seq)
        tags = dict((t[:2], t[5:]) for t in fields[11:])
        if "SA" in tags:
            assert a.get_tag("SA") == tags["SA"]
        else:
            assert not a.has_tag("SA")
        if "NM" in tags:
            assert int(a.get_tag("NM")) == int(tags["NM"])


@pytest.mark.parametrize("min_shift,depth", [(14, 5), (14, 6), (12, 4), (16, 3)])
def test_csi_index_serves_like_a_bai(dataset, tmp_path, min_shift, depth):
    """A `.csi` (CSI v1, `samtools index -c`) next to the BAM instead of a `.bai`: usable (index_state 1), same
    records from the parallel walk, per-contig loads and contig spans work; a damaged one is ignored."""
    d, bams = dataset
    p = str(tmp_path / "x.bam")
    shutil.copy(bams[0], p)
    bamio.index_bam(p, csi=True, min_shift=min_shift, depth=depth)
    assert os.path.exists(p + ".csi") and not os.path.exists(p + ".bai")
    f = bamio.AlignmentFile(p)
    assert f.index_state() == 1 and f.check_index()
    ref = bamio.AlignmentFile(bams[0])
    assert_same_columns(ref, f)
    assert f.contig_spans() is not None and list(f.contig_spans() > 0) == list(ref.contig_spans() > 0)
    one = bamio.AlignmentFile(p).load([f.references[1]])
    want = ref._cols["tid"] == 1
    assert len(one) == int(want.sum()) and np.array_equal(one._cols["pos"], ref._cols["pos"][want])
    # `<stem>.csi` is found too; a `.bai` wins when both are there
    q = str(tmp_path / "y.bam")
    shutil.copy(bams[0], q)
    shutil.copy(p + ".csi", str(tmp_path / "y.csi"))
    assert bamio.AlignmentFile(q).index_state() == 1
    # damaged: truncated, and with a flipped byte in its compressed payload
    raw = open(p + ".csi", "rb").read()
    for bad in (raw[: len(raw) // 2], raw[:40] + bytes([raw[40] ^ 0x55]) + raw[41:]):
        open(p + ".csi", "wb").write(bad)
        g = bamio.AlignmentFile(p)
        assert g.index_state() == 2
        assert_same_columns(ref, g)


def test_sa_tag_among_every_other_aux_type(tmp_path):
    """Records whose SA tag sits before, between and behind tags of every other aux type (A c C s S i I f d Z H and B
    arrays of each subtype, empty and long ones): both readers find the same SA string, keep the same raw tag bytes,
    and an array named CG is only taken for a CIGAR where the SAM specification says so."""
    rng = np.random.default_rng(5)
    names, lengths = ["chr1", "chr2"], [50000, 30000]
    others = [("NM", "i", 17), ("XA", "A", "x"), ("Xc", "c", -5), ("XC", "C", 200), ("Xs", "s", -3000), ("XS", "S", 60000),
              ("XI", "I", 4000000000), ("Xf", "f", 1.5), ("Xd", "d", 2.25), ("MD", "Z", "10A5^AC6"), ("XH", "H", "1AE301"),
              ("B0", "B", ("c", [])), ("B1", "B", ("C", [1, 2, 255])), ("B2", "B", ("s", [-1, 300])),
              ("B3", "B", ("S", list(range(300)))), ("B4", "B", ("i", [-7])), ("B5", "B", ("I", [1 << 31])),
              ("B6", "B", ("f", [0.5, -2.0])), ("CG", "B", ("I", [(40 << 4) | 0, (3 << 4) | 1]))]
    blobs, want_sa = [], []
    for i in range(60):
        tags = [others[k] for k in rng.permutation(len(others))[: int(rng.integers(0, len(others) + 1))]]
        sa = None
        if i % 3:
            sa = "chr2,%d,%s,20M30S,60,1;" % (100 + i, "+-"[i % 2])
            tags.insert(int(rng.integers(0, len(tags) + 1)), ("SA", "Z", sa))
        want_sa.append(sa)
        seq = "".join("ACGT"[b] for b in rng.integers(0, 4, 50))
        blobs.append(bamio.encode_record("r%d" % i, 0, 0, 100 + 37 * i, 60, [(50 << 4) | 0], seq, tags=tags))
    p = str(tmp_path / "aux.bam")
    bamio.write_bam(p, names, lengths, blobs)
    nat, py = bamio.AlignmentFile(p), bamio.AlignmentFile(p, reader="python")
    assert_same_columns(py, nat)
    for i, sa in enumerate(want_sa):
        r = nat.record(i)
        assert (r.get_tag("SA") if r.has_tag("SA") else None) == sa
        assert r.cigartuples == [(0, 50)]          # the CG array is not a CIGAR here: the stored one is no placeholder
        if r.has_tag("B3"):
            assert r.get_tag("B3") == list(range(300))
        if r.has_tag("Xd"):
            assert r.get_tag("Xd") == 2.25
