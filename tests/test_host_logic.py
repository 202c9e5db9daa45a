"""Host-side logic of the product that needs no GPU: BAM/FASTA I/O, candidate model, split-read
post-passes (fed with raw records from the C oracle), VCF writer, CLI precondition errors."""
import logging
import os

import numpy as np
import pytest

from oracle import orc, svim_oracle
from svim_asm_amd import SVCandidate, SVIM_COLLECT, SVIM_COMBINE, SVIM_inter, bamio, fasta
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = ["chr1", "chr10", "chr2", "chrX"]
LENGTHS = [3_000_000, 1_500_000, 2_000_000, 800_000]


def test_bam_reader_on_reference_fixture():
    f = bamio.AlignmentFile(os.path.join(GOLD, "chimeric_read.bam"))
    assert len(f) == 4 and f.header["HD"]["SO"] == "queryname"
    recs = list(f.fetch())
    assert [len(r.cigar_words) for r in recs] == [638, 272, 93, 77]
    assert [r.flag for r in recs] == [0, 2048, 2048, 2048]
    assert all(r._l_seq == 9900 and r.reference_id == 20 for r in recs)
    assert recs[0].get_tag("SA").count(";") == 3
    with pytest.raises(ValueError):
        f.check_index()
    cig, off, pos, tid = f.batch()
    assert list(off) == [0, 638, 910, 1003, 1080] and len(cig) == 1080


def test_sa_reconstruction_matches_test_satag_expectations():
    """retrieve_other_alignments (product) on the reference's fixtures — tests/test_satag.py."""
    f = bamio.AlignmentFile(os.path.join(GOLD, "chimeric_read.bam"))
    recs = list(f.fetch())
    supp = SVIM_COLLECT.retrieve_other_alignments(recs[0], f)
    assert len(supp) == 3
    for s, r in zip(supp, recs[1:]):
        assert s.cigarstring == r.cigarstring
        assert (s.reference_id, s.reference_start, s.reference_end) == (r.reference_id, r.reference_start, r.reference_end)
        assert (s.flag, s.mapping_quality, s.query_name) == (r.flag, r.mapping_quality, r.query_name)
        assert s.query_alignment_start == r.query_alignment_start
        assert s.query_alignment_end == r.query_alignment_end
    g = bamio.AlignmentFile(os.path.join(GOLD, "chimeric_read_errors.bam"))
    prim = [r for r in g.fetch() if not r.is_supplementary]
    assert len(SVIM_COLLECT.retrieve_other_alignments(prim[0], g)) == 2       # 7-field entry skipped
    one = SVIM_COLLECT.retrieve_other_alignments(prim[1], g)
    assert len(one) == 1 and one[0].mapping_quality == 0                     # mapq -400 → 0


def test_bam_and_fasta_roundtrip(tmp_path):
    from svim_asm_amd import synth_bam
    fa, bams = synth_bam.write_dataset(str(tmp_path), seed=3, contigs=(("a", 30000), ("b", 20000)), n_shared=4,
                                       n_private=1, median_aln=8000, dense_cluster=False, with_splits=False)
    f = bamio.AlignmentFile(bams[0])
    assert f.check_index() and f.header["HD"]["SO"] == "coordinate" and f.references == ("a", "b")
    ref = fasta.FastaFile(fa)
    assert ref.get_reference_length("a") == 30000
    whole = ref.fetch("a", 0, 30000)
    assert len(whole) == 30000 and ref.fetch("a", 59, 125) == whole[59:125] and ref.fetch("a", 29990, 40000) == whole[29990:]
    assert ref.fetch("a", 10, 10) == ""
    for r in f.fetch():
        if r.flag == 0 and len(r.cigar_words) > 3 and not r.has_tag("SA"):
            # M stretches of the query reproduce the reference up to the low substitution rate
            s = r.query_sequence
            assert r.seq_slice(5, 50) == s[5:50] and r.seq_slice(4, 51) == s[4:51]
            break
    with pytest.raises(IOError):
        fasta.FastaFile(str(tmp_path / "missing.fa"))
    os.remove(fa + ".fai")
    with pytest.raises(ValueError):
        fasta.FastaFile(fa)


def test_candidate_model_clamps_keys_and_breakend_normalisation():
    bam = helpers.FakeBam(NAMES, LENGTHS, [])
    d = SVCandidate.CandidateDeletion("chr2", -5, 2_000_010, ["r"], bam)
    assert d.get_source() == ("chr2", 0, 2_000_000) and d.get_key() == ("DEL", "chr2", 1_000_000)
    with pytest.raises(AssertionError):
        SVCandidate.CandidateDeletion("chr2", 10, 9, ["r"], bam)
    i = SVCandidate.CandidateInsertion("chr1", 100, 160, ["r"], "ACGT", bam)
    assert i.get_key() == ("INS", "chr1", 100) and i.get_destination() == ("chr1", 100, 160)
    b = SVCandidate.CandidateBreakend("chr2", 500, "fwd", "chr10", 700, "rev", ["r"], bam)
    # "chr10" < "chr2" as strings: endpoints swap and both directions flip
    assert (b.source_contig, b.source_start, b.source_direction) == ("chr10", 700, "fwd")
    assert (b.dest_contig, b.dest_start, b.dest_direction) == ("chr2", 500, "rev")
    same = SVCandidate.CandidateBreakend("chr1", 5, "fwd", "chr1", 5, "fwd", ["r"], bam)
    assert (same.source_direction, same.dest_direction) == ("rev", "rev")
    t = SVCandidate.CandidateDuplicationTandem("chr1", 10, 110, 2, True, ["r"], bam)
    assert t.get_destination() == ("chr1", 110, 310)


@pytest.mark.parametrize("seed", range(4))
def test_split_read_post_passes_match_oracle(seed):
    """host candidate assembly over oracle records (C decision tree + record-level post-passes) == the pinned
    Python oracle of analyze_read_segments, per read."""
    rng = np.random.default_rng(seed)
    recs = helpers.engineered_split_records(rng, NAMES, LENGTHS, 400)
    o = helpers.options(**([{}, dict(min_sv_size=30, max_sv_size=2000)][seed % 2]))
    bam = helpers.FakeBam(NAMES, LENGTHS, recs)
    lens = dict(zip(NAMES, LENGTHS))
    prm = (o.min_sv_size, o.max_sv_size, o.query_gap_tolerance, o.query_overlap_tolerance,
           o.reference_gap_tolerance, o.reference_overlap_tolerance)
    kinds = set()
    for rec, prim in zip(recs, bam.fetch()):
        supp = [s for s in SVIM_COLLECT.retrieve_other_alignments(prim, bam) if s.mapping_quality >= o.min_mapq]
        rows = [SVIM_inter.segment_row(a) for a in [prim] + supp]
        segs = np.array(rows, dtype=np.int32).view(orc.SEG_DTYPE).reshape(-1)
        raw = orc.segments_classify(segs, np.array([0, len(rows)], np.uint32),
                                    np.array([prim.infer_read_length()], np.int32), prm)
        # the post-passes run on the GPU in the product: here the record-level oracle produces the records
        # the host turns into candidates
        rank = SVIM_inter.contig_ranks(bam)
        rows_ = [tuple(int(r[k]) for k in ("kind", "a0", "a1", "a2", "a3", "a4", "a5")) for r in raw]
        post = np.zeros(64, dtype=orc.RAW_DTYPE)
        code = {"TANDEM": 1, "DUP_INT": 2, "INV": 3}
        recs_ = svim_oracle.postpass_records(rows_, rank.tolist(), o.min_sv_size, o.max_sv_size)
        for k, t in enumerate(recs_):
            vals = [code[t[0]]] + [int(v) for v in t[1:]]
            for name_, v in zip(("kind", "a0", "a1", "a2", "a3", "a4", "a5"), vals + [0] * (7 - len(vals))):
                post[k][name_] = v
        got = [helpers.candidate_tuple(c) for c in
               SVIM_inter.candidates_from_records(raw, post[:len(recs_)], prim, bam, lambda a, b: prim.query_sequence[a:b])]
        osupp = [s for s in svim_oracle.retrieve_other_alignments(rec, NAMES) if s["mapq"] >= o.min_mapq]
        exp = svim_oracle.analyze_read_segments(rec, osupp, NAMES, lens, o)
        assert got == exp
        kinds |= {c[0] for c in exp}
    assert kinds == {"DEL", "INS", "BND", "DUP_TAN", "DUP_INT", "INV"}


@pytest.mark.parametrize("seed", range(3))
def test_vcf_writer_matches_oracle(tmp_path, seed):
    rng = np.random.default_rng(seed)
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=30000)) for n in NAMES}
    lengths = [30000] * len(NAMES)
    ref = helpers.FakeFasta(seqs)
    bam = helpers.FakeBam(NAMES, lengths, [])
    tuples = helpers.random_candidates(rng, NAMES, lengths, seqs, 150, "x")
    cands = [helpers.build_candidate(t, bam, SVCandidate) for t in tuples]
    assert [helpers.candidate_tuple(c) for c in cands] == tuples
    o = helpers.options(working_dir=str(tmp_path), query_names=bool(seed & 1),
                        tandem_duplications_as_insertions=seed == 1, interspersed_duplications_as_insertions=seed == 2,
                        types=["DEL,INS,INV,DUP:TANDEM,DUP:INT,BND", "DEL,INS", "INV,BND,DUP:INT"][seed])
    by = lambda t: [c for c in cands if c.type == t]
    SVIM_COMBINE.write_final_vcf(by("DUP_INT"), by("INV"), by("DUP_TAN"), by("DEL"), by("INS"), by("BND"), "1.0.3",
                                 NAMES, lengths, [t.strip() for t in o.types.split(",")], ref, o)
    got = "".join(l for l in open(tmp_path / "variants.vcf") if not l.startswith("##fileDate="))
    assert got == svim_oracle.vcf_text(tuples, ref.fetch, NAMES, lengths, o)


def test_sorted_nicely_is_natural_and_stable():
    e = [(("chr10", 5, 6), "a", "DEL"), (("chr2", 5, 6), "b", "DEL"), (("chr2", 5, 6), "c", "INS"), (("chr1", 9, 9), "d", "DEL")]
    assert [x[1] for x in SVIM_COMBINE.sorted_nicely(e)] == ["d", "b", "c", "a"]


def test_sorted_nicely_follows_the_reference_rule_on_awkward_names():
    """The reference's key (SVIM_COMBINE.py:367-373): the contig name split at digit runs, the runs as
    integers, then start and end — "chr01" and "chr1" compare equal there and fall through to the positions."""
    import random
    import re

    def reference_order(entries):
        convert = lambda text: int(text) if text.isdigit() else text  # noqa: E731
        return sorted(entries, key=lambda e: ([convert(c) for c in re.split("([0-9]+)", str(e[0][0]))], e[0][1], e[0][2]))
    rnd = random.Random(3)
    names = ["chr1", "chr01", "chr10", "chr2", "chrX", "chrUn_KI270", "chr2_random", "1", "10", "2", "MT", "chr001", "a1b2", "a1b10"]
    for _ in range(300):
        e = [((rnd.choice(names), rnd.randrange(4), rnd.randrange(4)), "line%d" % i, "DEL") for i in range(rnd.randrange(0, 50))]
        assert SVIM_COMBINE.sorted_nicely(e) == reference_order(e)


def test_pack_keys_orders_like_the_get_key_tuples():
    """_pack_keys: u64 order == order of the (type, contig, position) tuples of get_key() with the contigs under
    Python's str order (SVIM_COMBINE.py:17) — for every candidate class, and out-of-range positions are refused."""
    bam = helpers.FakeBam(NAMES, LENGTHS, [])
    rng = np.random.default_rng(11)
    items = []
    for i in range(600):
        c, s = NAMES[int(rng.integers(len(NAMES)))], int(rng.integers(0, 900_000))
        kind = i % 6
        if kind == 0:
            cand = SVCandidate.CandidateDeletion(c, s, s + 50, ["r"], bam)
        elif kind == 1:
            cand = SVCandidate.CandidateInsertion(c, s, s + 50, ["r"], "A" * 50, bam)
        elif kind == 2:
            cand = SVCandidate.CandidateInversion(c, s, s + 500, ["r"], True, bam)
        elif kind == 3:
            cand = SVCandidate.CandidateDuplicationTandem(c, s, s + 100, 2, True, ["r"], bam)
        elif kind == 4:
            cand = SVCandidate.CandidateDuplicationInterspersed(c, s, s + 100, NAMES[0], s + 5, s + 105, ["r"], bam)
        else:
            cand = SVCandidate.CandidateBreakend(c, s, "fwd", NAMES[1], s + 7, "rev", ["r"], bam)
        items.append((1 + i % 2, cand))
    packed = SVIM_COMBINE._pack_keys(items)
    tuples = [(SVIM_COMBINE.TYPE_ORDER.index(c.get_key()[0]), c.get_key()[1], c.get_key()[2]) for _, c in items]
    by_tuple = sorted(range(len(items)), key=lambda i: tuples[i])
    by_packed = sorted(range(len(items)), key=lambda i: int(packed[i]))
    assert by_tuple == by_packed  # both sorts are stable: equal keys keep input order in both
    for i, j in zip(by_tuple, by_tuple[1:]):
        assert (tuples[i] == tuples[j]) == (packed[i] == packed[j])
    assert SVIM_COMBINE._pack_keys([]).dtype == np.uint64 and len(SVIM_COMBINE._pack_keys([])) == 0
    far = SVCandidate.CandidateInsertion("chr1", 10, 20, ["r"], "A", bam)
    far.dest_start = 1 << 32
    with pytest.raises(ValueError):
        SVIM_COMBINE._pack_keys([(1, far)])
    far.dest_start = -1
    with pytest.raises(ValueError):
        SVIM_COMBINE._pack_keys([(1, far)])


def test_cli_rejects_unsorted_bam_without_touching_the_gpu(tmp_path, caplog):
    from svim_asm_amd import cli
    wd = tmp_path / "wd"
    with caplog.at_level(logging.ERROR):
        cli.main(["haploid", str(wd), os.path.join(GOLD, "chimeric_read.bam"), os.path.join(GOLD, "config1", "ref.fa")])
    assert "needs to be coordinate-sorted" in caplog.text
    assert not (wd / "variants.vcf").exists()


@pytest.mark.parametrize("seed", range(4))
def test_native_recipes_equal_the_array_form(seed):
    """svx_pair_recipes (libsvx.so, host arithmetic) against _haplotype_pieces_numpy — the recipes of compute_distance
    (SVIM_COMBINE.py:43-100) as array expressions — on random candidate columns of every type: windows clipped at both
    contig ends, empty and absent alleles, tandem copies, interspersed duplications whose source intervals are fetched,
    one and two haplotype tables' sequence pools, more than 8192 jobs (the threaded form)."""
    from svim_asm_amd.table import CandidateTable, T_DEL, T_DUP_INT, T_DUP_TAN, T_INS, T_INV
    rng = np.random.default_rng(7000 + seed)
    names = ["c%d" % i for i in range(4)]
    lens = [int(x) for x in rng.integers(3000, 9000, size=4)]
    seqs = {n: "".join(rng.choice(list("ACGTacgtN"), size=l)) for n, l in zip(names, lens)}
    ref = helpers.FakeFasta(seqs)
    n = 600
    pool_bytes = 50_000
    T = CandidateTable(names, lens, n, seqs=rng.integers(65, 91, size=pool_bytes).astype(np.uint8))
    T.type[:] = rng.choice([T_DEL, T_INV, T_INS, T_DUP_TAN, T_DUP_INT], size=n)
    T.sc[:] = rng.integers(0, 4, size=n)
    T.dc[:] = T.sc
    L_row = np.array(lens)[T.sc]
    T.ss[:] = rng.integers(0, L_row)
    T.se[:] = np.minimum(T.ss + rng.integers(0, 400, size=n), L_row + rng.integers(0, 3, size=n))  # some ends behind the contig
    T.ds[:] = rng.integers(0, L_row)
    T.de[:] = T.ds
    split = 30_000 if seed % 2 == 0 else None
    hap2 = rng.random(n) < 0.5
    T.q_len[:] = np.where(T.type == T_INS, rng.integers(0, 300, size=n), 0)
    T.q_off[:] = np.where(hap2 & (split is not None), rng.integers(30_000, 49_000, size=n), rng.integers(0, 29_000, size=n))
    T.copies[:] = np.where(T.type == T_DUP_TAN, rng.integers(0, 5, size=n), 0)
    src_like = (T.type == T_DEL) | (T.type == T_INV) | (T.type == T_DUP_TAN)
    kstart, kend = np.where(src_like, T.ss, T.ds), np.where(src_like, T.se, T.ds)
    # partitions: rows of one type on one contig; a job = two rows of a partition
    J = [50, 700, 9000, 0][seed]
    n_parts = 40
    p_type = rng.choice([T_DEL, T_INV, T_INS, T_DUP_TAN, T_DUP_INT], size=n_parts).astype(np.int64)
    p_contig = rng.integers(0, 4, size=n_parts).astype(np.int64)
    members = [np.flatnonzero((T.type == p_type[p]) & (T.sc == p_contig[p])) for p in range(n_parts)]
    ok_parts = [p for p in range(n_parts) if len(members[p]) >= 2]
    job_part = rng.choice(ok_parts, size=J).astype(np.int64) if J else np.zeros(0, np.int64)
    job_a = np.array([rng.choice(members[p]) for p in job_part], np.int64)
    job_b = np.array([rng.choice(members[p]) for p in job_part], np.int64)
    L_part = np.array(lens, np.int64)[p_contig]
    seg_lo = np.array([kstart[m].min() if len(m) else 0 for m in members])
    seg_hi = np.array([kend[m].max() if len(m) else 0 for m in members])
    wlo = np.maximum(0, seg_lo - 100)
    whi = np.minimum(L_part, seg_hi + 100)
    win_base = np.concatenate(([0], np.cumsum(np.maximum(whi - wlo, 0)))).astype(np.int64)
    args = (T, kstart, kend, job_a, job_b, job_part, p_type, p_contig, win_base, wlo.astype(np.int64), L_part, ref)
    got, got_extra = SVIM_COMBINE._haplotype_pieces(*args, seq_split=split)
    exp, exp_extra = SVIM_COMBINE._haplotype_pieces_numpy(*args, seq_split=split)
    assert got.shape == exp.shape
    for f in ("off", "len", "repeat", "flags"):
        assert np.array_equal(got[f], exp[f]), f
    assert len(got_extra) == len(exp_extra) and all(np.array_equal(a, b) for a, b in zip(got_extra, exp_extra))
    if J:
        T.copies[job_a[0]] = 70000
        p_type[job_part[0]] = T_DUP_TAN
        with pytest.raises(ValueError):
            SVIM_COMBINE._haplotype_pieces(*args, seq_split=split)


def test_device_inflate_share_policy(monkeypatch):
    """bamio.default_device_inflate_percent: the environment wins; otherwise the whole call with at most 24 CPUs' worth of
    time (hardware threads or cgroup quota), none above."""
    from svim_asm_amd import bamio
    monkeypatch.setenv("SVX_BAM_DEVICE_INFLATE", "35")
    assert bamio.default_device_inflate_percent() == 35
    monkeypatch.setenv("SVX_BAM_DEVICE_INFLATE", "250")
    assert bamio.default_device_inflate_percent() == 100
    monkeypatch.delenv("SVX_BAM_DEVICE_INFLATE")
    monkeypatch.setattr(bamio, "host_cpus", lambda: 16.0)
    assert bamio.default_device_inflate_percent() == 100
    monkeypatch.setattr(bamio, "host_cpus", lambda: 64.0)
    assert bamio.default_device_inflate_percent() == 0
    # not a number: a warning and the default, never an exception (the module is imported by every entry point)
    monkeypatch.setenv("SVX_BAM_DEVICE_INFLATE", "on")
    with pytest.warns(UserWarning):
        bamio._ENV_WARNED.clear()
        assert bamio.default_device_inflate_percent() == 0
    monkeypatch.setattr(bamio, "host_cpus", lambda: 8.0)
    assert bamio.default_device_inflate_percent() == 100
    # the class default is resolved when a file is loaded, not at import
    assert bamio.AlignmentFile.device_inflate_percent is None


def test_host_cpus_is_the_smaller_of_threads_and_quota():
    import os
    from svim_asm_amd import bamio
    n = bamio.host_cpus()
    assert 0 < n <= (os.cpu_count() or 1)


def test_command_device_share_policy(monkeypatch, tmp_path):
    """cli._open_file: under a CPU quota each sequence-slice call goes to the device (the readers' default), with a core
    per thread none of it; SVX_BAM_DEVICE_INFLATE overrides either."""
    import types
    from svim_asm_amd import bamio, cli, synth_bam
    fa, bams = synth_bam.write_dataset(str(tmp_path), seed=3, contigs=(("chrA", 80000), ("chrB", 60000)), n_shared=3, n_private=1,
                                       median_aln=20000)
    opts = types.SimpleNamespace(device=0, sub="diploid", no_bgzf_crc=False)
    monkeypatch.setattr(bamio.AlignmentFile, "device_inflate_percent", 50)
    monkeypatch.delenv("SVX_BAM_DEVICE_INFLATE", raising=False)
    monkeypatch.setattr(bamio, "host_cpus", lambda: 64.0)
    assert cli._open_file(bams[0], opts).device_inflate_percent == 0
    monkeypatch.setattr(bamio, "host_cpus", lambda: 16.0)
    assert cli._open_file(bams[0], opts).device_inflate_percent == 100
    monkeypatch.setenv("SVX_BAM_DEVICE_INFLATE", "40")
    assert cli._open_file(bams[0], opts).device_inflate_percent == 40
    assert bamio.AlignmentFile(bams[0], device=0).device_inflate_percent == 50  # (the library's readers keep their default)


def test_job_token_names_the_job_not_the_shell(monkeypatch):
    """shard._job_token: SVX_JOB_TOKEN wins; SVX_RENDEZVOUS alone names a hand-launched job whatever shell or wrapper
    started each rank; torchrun's default run id "none" names nothing (the parent process + address + port do)."""
    import os
    from svim_asm_amd import shard
    for name in ("SVX_JOB_TOKEN", "SVX_RENDEZVOUS", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv("MASTER_PORT", "29511")
    by_parent = shard._job_token()
    assert str(os.getppid()).encode() in by_parent and b"29511" in by_parent
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    assert shard._job_token() == by_parent
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job-7")
    assert shard._job_token() == b"run-job-7"
    monkeypatch.setenv("SVX_RENDEZVOUS", "my-hand-launched-job")
    assert shard._job_token() == b"rendezvous-my-hand-launched-job"
    monkeypatch.setattr(os, "getppid", lambda: 1)  # another shell: the same token
    assert shard._job_token() == b"rendezvous-my-hand-launched-job"
    monkeypatch.setenv("SVX_JOB_TOKEN", "t0k")
    assert shard._job_token() == b"t0k"


def test_quota_threads_share_the_quota_among_readers_and_ranks(monkeypatch):
    """bamio.quota_threads: under a CPU quota the readers of all processes of a run together get the quota's CPUs (at
    least two each); without a quota the library's policy (ingest_threads); SVX_INGEST_THREADS overrides both."""
    import os
    from svim_asm_amd import bamio
    monkeypatch.delenv("SVX_INGEST_THREADS", raising=False)
    monkeypatch.setattr(os, "cpu_count", lambda: 256)
    monkeypatch.setattr(bamio, "host_cpus", lambda: 16.0)
    assert bamio.quota_threads(2) == 8 and bamio.quota_threads(1) == 16
    assert bamio.quota_threads(2, processes=2) == 4 and bamio.quota_threads(2, processes=4) == 2 and bamio.quota_threads(2, processes=8) == 2
    assert bamio.ingest_threads(2) == 32  # the long-lived caller's bursts: a quarter of the hardware threads in total
    monkeypatch.setattr(bamio, "host_cpus", lambda: 256.0)
    assert bamio.quota_threads(2) == bamio.ingest_threads(2) == 32
    monkeypatch.setenv("SVX_INGEST_THREADS", "5")
    assert bamio.quota_threads(2) == bamio.ingest_threads(2) == 5


def test_cohort_options_and_plan(monkeypatch):
    """svim-asm-cohort's own options are taken out of the argument list before the reference's parser sees it; workers and
    reader threads follow the CPUs the process may use."""
    from svim_asm_amd import bamio, cohort
    rest, k = cohort._take_option(["--min_sv_size", "50", "--cohort_workers", "6", "--types=DEL"], "--cohort_workers", 0)
    assert rest == ["--min_sv_size", "50", "--types=DEL"] and k == 6
    rest, g = cohort._take_option(["--cohort_group=0", "--symbolic_alleles"], "--cohort_group", 1)
    assert rest == ["--symbolic_alleles"] and g == 0
    assert cohort._take_option(["--symbolic_alleles"], "--cohort_threads", 0) == (["--symbolic_alleles"], 0)
    monkeypatch.setattr(bamio, "host_cpus", lambda: 16.0)
    assert cohort.default_workers() == 4 and cohort.default_reader_threads(4, 2) == 3 and cohort.default_reader_threads(3, 2) == 4
    monkeypatch.setattr(bamio, "host_cpus", lambda: 8.0)
    assert cohort.default_workers() == 2 and cohort.default_reader_threads(2, 2) == 3
    monkeypatch.setattr(bamio, "host_cpus", lambda: 2.0)
    assert cohort.default_reader_threads(2, 2) == 2  # never below two
    # inflate lanes: one per three readers in flight, the library's two at least; the library takes 1..16
    assert [cohort.default_lanes(w, 2) for w in (1, 2, 4, 6, 8, 32)] == [2, 2, 3, 4, 5, 16] and cohort.default_lanes(4, 1) == 2
    assert cohort._take_option(["--cohort_lanes", "5"], "--cohort_lanes", 0) == ([], 5)
    assert cohort.COHORT_DEVICE_INFLATE_WAIT_MS >= 1000  # (a call must not give up on its lane: cohort.default_lanes)
    from svim_asm_amd import _lib
    lib = _lib.load()
    assert lib.svx_bam_set_inflate_lanes(0) != 0 and lib.svx_bam_set_inflate_lanes(17) != 0
    assert lib.svx_bam_set_inflate_lanes(4) == 0 and lib.svx_bam_set_inflate_lanes(2) == 0


def test_cohort_manifest_errors(tmp_path):
    from svim_asm_amd import cohort
    m = tmp_path / "m.txt"
    m.write_text("# comment\n\nwd1 a.bam b.bam\n")
    assert cohort.read_manifest(str(m), 2) == [(os.path.abspath("wd1"), ["a.bam", "b.bam"])]
    with pytest.raises(ValueError):
        cohort.read_manifest(str(m), 1)  # a diploid line in a haploid cohort
    m.write_text("# nothing\n")
    with pytest.raises(ValueError):
        cohort.read_manifest(str(m), 2)


def test_timeline_marks_only_when_asked(tmp_path, monkeypatch):
    """svim_asm_amd._timeline: one dictionary look-up without SVX_CLI_TIMELINE; with it, JSON lines of wall clock, process
    CPU seconds and thread per mark."""
    import importlib
    import json
    from svim_asm_amd import _timeline
    monkeypatch.delenv("SVX_CLI_TIMELINE", raising=False)
    tl = importlib.reload(_timeline)
    tl.mark("nothing")
    tl.dump()
    assert not tl.enabled() and tl._marks == []
    path = tmp_path / "tl.jsonl"
    monkeypatch.setenv("SVX_CLI_TIMELINE", str(path))
    tl = importlib.reload(_timeline)
    tl.mark("a")
    tl.mark("b", stages={"x": 0.5})
    tl.dump()
    rows = [json.loads(l) for l in open(path)]
    assert [r["name"] for r in rows] == ["a", "b"] and rows[1]["stages"] == {"x": 0.5}
    assert rows[0]["t"] <= rows[1]["t"] and rows[0]["cpu"] <= rows[1]["cpu"] and rows[0]["thread"] == "MainThread"
    monkeypatch.delenv("SVX_CLI_TIMELINE")
    importlib.reload(_timeline)


def test_device_numa_lookup_reads_sysfs(monkeypatch, tmp_path):
    """cohort.device_numa_cpus: the device's PCI address (svx_device_pci_bus_id) -> numa_node -> that node's cpulist, cut
    down to the CPUs the process may run on; (address, None, None) where the platform reports no node."""
    import builtins
    import os
    from svim_asm_amd import _lib, cohort

    class FakeLib:
        def svx_device_pci_bus_id(self, device, buf, n):
            buf.value = b"0000:72:00.0"
            return 0
    monkeypatch.setattr(_lib, "load", lambda: FakeLib())
    files = {"/sys/bus/pci/devices/0000:72:00.0/numa_node": "1\n", "/sys/devices/system/node/node1/cpulist": "2-3,6,64-65\n"}
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path in files:
            import io
            return io.StringIO(files[path])
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(0, 8)))
    assert cohort.device_numa_cpus(0) == ("0000:72:00.0", 1, [2, 3, 6])
    files["/sys/bus/pci/devices/0000:72:00.0/numa_node"] = "-1\n"
    assert cohort.device_numa_cpus(0) == ("0000:72:00.0", None, None)


def test_deferred_check_falls_back_to_the_threads(tmp_path):
    """bamio.AlignmentFile.defer_verify without a usable device: whatever the walks leave pending is checked by the
    reader's threads — at the end of the next sequence-slice call or in verify_pending() — and nothing stays pending; a
    reader without a device share or a pinned device does not defer at all."""
    import numpy as np
    from svim_asm_amd import bamio, synth_bam
    fa, bams = synth_bam.write_dataset(str(tmp_path), seed=5, contigs=(("chrA", 300000), ("chrB", 200000)), n_shared=6, n_private=2,
                                       median_aln=60000)
    plain = bamio.AlignmentFile(bams[0])
    plain.defer_verify = True
    plain.load(None)
    assert plain.pending_members == 0  # (no pinned device: the walks check as they go)
    f = bamio.AlignmentFile(bams[0], device=0)
    f.device_inflate_percent = 100
    f.defer_verify = True
    f.load(None)
    if f.pending_members == 0:
        pytest.skip("this build has no device kernels registered: nothing is deferred")
    for name in ("tid", "pos", "l_seq", "flag"):
        assert np.array_equal(f._cols[name], plain._cols[name])
    l_seq = f._cols["l_seq"]
    rec = np.arange(len(l_seq), dtype=np.uint32)
    a = np.zeros(len(rec), np.int64)
    b = np.minimum(l_seq, 50).astype(np.int64)
    got = f.sequence_slices_raw(rec, a, b)
    exp = plain.sequence_slices_raw(rec, a, b)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
    assert f.pending_members == 0
    g = bamio.AlignmentFile(bams[0], device=0)
    g.device_inflate_percent = 100
    g.defer_verify = True
    g.load(None)
    assert g.pending_members > 0
    g.verify_pending()
    assert g.pending_members == 0
