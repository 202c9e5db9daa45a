"""Multi-process (gloo, world_size 2, CPU) test of the contig-sharding path: plan, exchange and
re-assembly order.  The per-shard compute is injected (oracle-backed) because the product's
kernels need a GPU; the sharding / merge logic under test is the product's."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "config1")


def test_lpt_assign_is_deterministic_and_balanced():
    from svim_asm_amd.shard import lpt_assign
    w = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]
    owner = lpt_assign(w, 8)
    assert owner == lpt_assign(list(w), 8)
    loads = [sum(x for x, o in zip(w, owner) if o == r) for r in range(8)]
    assert max(loads) <= 1.1 * sum(w) / 8
    assert lpt_assign([], 4) == [] and lpt_assign([5], 4) == [0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, transport):
    """One rank: the PRODUCT's sharded COLLECT + PAIR (native BAM reader, columnar host logic, table exchange)
    with the device answered by the oracle — there is no GPU here.  transport "gloo": the exchange runs over a
    torch.distributed process group; "socket": over the product's own unix-domain socket group, the way the
    svim-asm CLI runs under torch.distributed.run (RANK / WORLD_SIZE in the environment, no torch)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = None
    if transport == "gloo":
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    try:
        from svim_asm_amd import _lib, bamio, fasta, shard
        from tests import helpers
        ctx = helpers.OracleBackedContext()
        _lib.default_context = lambda device=0: ctx
        o = helpers.options()
        ref = fasta.FastaFile(os.path.join(GOLD, "ref.fa"))
        b1 = bamio.AlignmentFile(os.path.join(GOLD, "hap1.bam"))
        b2 = bamio.AlignmentFile(os.path.join(GOLD, "hap2.bam"))
        t1, t2 = shard.collect_sharded([b1, b2], o)
        paired = shard.pair_sharded(t1, t2, ref, b1, o)
        assert "torch" not in sys.modules or transport == "gloo", "the socket transport must not import torch"
        q.put((rank, [helpers.candidate_tuple(c) for c in t1.objects()], [helpers.candidate_tuple(c) for c in t2.objects()],
               [helpers.candidate_tuple(c) for c in paired.objects()], len(b1), len(b2)))
    finally:
        if dist is not None:
            dist.destroy_process_group()
        else:
            from svim_asm_amd import shard
            shard.shutdown()


@pytest.mark.parametrize("transport,world", [("gloo", 2), ("socket", 2), ("socket", 8)])
def test_sharded_collect_and_pair_equal_single_process(transport, world):
    """world 8 on the three contigs of the config-1 sample: five ranks own no contig at all — they still meet the
    others at both exchanges and end with the same tables (the rendezvous of eight, empty shards, header-order re-assembly)."""
    import torch.multiprocessing as mp
    from oracle import orc, run_oracle, svim_oracle
    from svim_asm_amd import bamio
    from tests import helpers
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    o = helpers.options()
    exp1, names, lengths = run_oracle.candidates_from_bam(os.path.join(GOLD, "hap1.bam"), o)
    exp2, _, _ = run_oracle.candidates_from_bam(os.path.join(GOLD, "hap2.bam"), o)
    ps = run_oracle._stub()
    fa = ps.FastaFile(os.path.join(GOLD, "ref.fa"))
    exp_pair = svim_oracle.pair_candidates(exp1, exp2, fa.fetch, names, lengths, dict(zip(fa.references, fa.lengths)), o,
                                           edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
    total = len(bamio.AlignmentFile(os.path.join(GOLD, "hap1.bam")))
    for rank, c1, c2, paired, n1, n2 in results:
        assert c1 == exp1, "rank %d collect order differs (hap 1)" % rank
        assert c2 == exp2, "rank %d collect order differs (hap 2)" % rank
        assert paired == exp_pair, "rank %d pair order differs" % rank
        assert n1 < total, "rank %d indexed %d of %d records: contig-restricted ingest" % (rank, n1, total)
    assert sorted(r[0] for r in results) == list(range(world))
    with_records = sum(1 for r in results if r[4] > 0)
    n_contigs = len(bamio.AlignmentFile(os.path.join(GOLD, "hap1.bam")).references)
    assert with_records == min(world, n_contigs), "every contig owned by exactly one rank, the other ranks empty"
    assert sum(r[4] for r in results) == total, "the ranks' records add up to the file's: no contig twice, none missing"


def test_eight_rank_plan_on_the_full_size_spans():
    """The rank plan of BASELINE config 4 (8 GPUs) on the full-size diploid sample, from the compressed bytes the two
    `.bai` indices attribute to each contig (tests/golden/full_bai_spans.json, tools/dump_bai_spans.py): every contig
    owned exactly once, the heaviest rank within 1.15 of the mean for 2, 4 and 8 ranks, the same plan on every rank
    (deterministic), and the merged rows back in header order whatever the owners are."""
    import json
    import numpy as np
    from svim_asm_amd.shard import lpt_assign
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "full_bai_spans.json")))
    w = np.asarray(fx["files"]["hap1.bam"]["contig_spans"], dtype=np.int64) + np.asarray(fx["files"]["hap2.bam"]["contig_spans"], dtype=np.int64)
    assert len(w) == len(fx["references"]) == 24 and (w > 0).all()  # chr1..22, X, Y
    for size in (2, 4, 8):
        owner = lpt_assign(w, size)
        assert owner == lpt_assign(list(w), size) and len(owner) == len(w) and set(owner) == set(range(size))
        loads = np.bincount(owner, weights=w, minlength=size)
        assert loads.sum() == w.sum()
        assert loads.max() <= 1.15 * w.sum() / size, (size, loads.max() / (w.sum() / size))
        # re-assembly as collect_sharded does it: every rank's rows carry their contig index; a stable sort by it puts
        # the concatenation of the ranks' parts (each in header order for its own contigs) back in header order
        parts = [np.array([c for c in range(len(w)) if owner[c] == r for _ in range(3)]) for r in range(size)]
        rec_tid = np.concatenate(parts)
        merged = rec_tid[np.argsort(rec_tid, kind="stable")]
        assert (merged == np.repeat(np.arange(len(w)), 3)).all()


def test_exchange_socket_lives_in_a_private_directory(tmp_path, monkeypatch):
    """The rank exchange binds a PATH socket inside <tmp>/svx-<uid>, a directory that must belong to the user and be
    closed to everybody else (mode 0700): a directory somebody else could enter — or pre-create — is refused instead of
    used.  (The peers' uid is checked on top of that, SO_PEERCRED, and the messages are typed arrays, never pickles.)"""
    import stat
    from svim_asm_amd import shard
    monkeypatch.delenv("XDG_RUNTIME_DIR", raising=False)
    import tempfile
    short = tempfile.mkdtemp(prefix="svx", dir="/tmp")  # (a deep TMPDIR makes the product fall back to /tmp: 107-byte socket paths)
    monkeypatch.setenv("TMPDIR", short)
    monkeypatch.setattr(tempfile, "tempdir", None)
    monkeypatch.setenv("MASTER_PORT", "12345")
    path = shard._rendezvous_path()
    d = os.path.dirname(path)
    assert d == os.path.join(short, "svx-%d" % os.getuid()) and path.endswith(".sock")
    assert stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    monkeypatch.setenv("MASTER_PORT", "12346")
    assert shard._rendezvous_path() != path  # another job, another socket
    os.chmod(d, 0o755)
    with pytest.raises(RuntimeError):
        shard._rendezvous_path()
    os.chmod(d, 0o700)
    monkeypatch.setattr(tempfile, "tempdir", None)


def test_rank_0_turns_strangers_away_and_still_meets_its_own_ranks(tmp_path, monkeypatch):
    """Two jobs of one user with the same MASTER_PORT and no run id share the socket path.  A peer that connects and says
    nothing, one that announces another job's token and one that claims a rank out of range are turned away — they
    neither hang rank 0 (the announcement is read under the rendezvous deadline) nor make the job fail — and the
    job's own rank is still met."""
    import socket
    import tempfile
    import threading
    import time
    from svim_asm_amd import shard
    monkeypatch.delenv("XDG_RUNTIME_DIR", raising=False)
    short = tempfile.mkdtemp(prefix="svx", dir="/tmp")  # (a unix socket path holds 107 bytes: pytest's tmp_path can be longer)
    monkeypatch.setenv("TMPDIR", short)
    monkeypatch.setattr(tempfile, "tempdir", None)
    monkeypatch.setenv("MASTER_PORT", "23456")
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    monkeypatch.setenv("SVX_JOB_TOKEN", "job-A")
    path = shard._rendezvous_path()
    out = {}

    def rank0():
        try:
            out["g0"] = shard._SocketGroup(0, 2, timeout=20.0)
        except Exception as e:  # noqa: BLE001
            out["err"] = e
    t0 = threading.Thread(target=rank0)
    t0.start()
    deadline = time.time() + 10
    while not os.path.exists(path) and time.time() < deadline:
        time.sleep(0.01)
    silent = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    silent.connect(path)                      # says nothing (kept open: only the deadline can end it) ...
    other = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    other.connect(path)
    shard._send_msg(other, [b"1", b"job-B"])  # ... a rank 1 of ANOTHER job ...
    wrong = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    wrong.connect(path)
    shard._send_msg(wrong, [b"7", b"job-A"])  # ... and a rank that does not exist
    silent.close()
    g1 = shard._SocketGroup(1, 2, timeout=20.0)  # the job's own rank 1 (same token)
    t0.join(30)
    assert "err" not in out, out.get("err")
    g0 = out["g0"]
    assert g0.peers[1] is not None
    for s in (other, wrong):
        s.settimeout(5)
        assert s.recv(1) == b""  # turned away: rank 0 closed them
        s.close()
    g0.close()
    g1.close()


def test_rank_0_gives_up_with_a_deadline_when_only_strangers_come(tmp_path, monkeypatch):
    import socket
    import tempfile
    import threading
    import time
    from svim_asm_amd import shard
    monkeypatch.delenv("XDG_RUNTIME_DIR", raising=False)
    short = tempfile.mkdtemp(prefix="svx", dir="/tmp")
    monkeypatch.setenv("TMPDIR", short)
    monkeypatch.setattr(tempfile, "tempdir", None)
    monkeypatch.setenv("MASTER_PORT", "23457")
    monkeypatch.setenv("SVX_JOB_TOKEN", "job-A")
    path = shard._rendezvous_path()
    out = {}

    def rank0():
        t = time.time()
        try:
            shard._SocketGroup(0, 2, timeout=1.5)
        except RuntimeError as e:
            out["err"], out["s"] = str(e), time.time() - t
    t0 = threading.Thread(target=rank0)
    t0.start()
    deadline = time.time() + 10
    while not os.path.exists(path) and time.time() < deadline:
        time.sleep(0.01)
    silent = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    silent.connect(path)  # connects, never announces itself, never closes
    t0.join(20)
    silent.close()
    assert "only 1 of 2 ranks" in out["err"] and out["s"] < 6.0
    assert not os.path.exists(path)  # the listener is gone with the failed rendezvous
