"""Contig sharding across the GPUs of one node.

COLLECT: every alignment (and every chimeric read: its SA-derived segments are reconstructed
from the primary's tag, SVIM_COLLECT.py:76) lives wholly in the contig of its record, so contigs
are independent units.  They are distributed over the ranks by LPT bin packing on the compressed
bytes the `.bai` attributes to them; each rank walks and inflates only its own contigs of BOTH
haplotype BAMs, runs the ordinary batched COLLECT (one device submission) on its own GPU, and the
candidate TABLES (numpy columns, a few MB) are exchanged once and re-assembled in header-contig
order, which is the reference's output order (SVIM_COLLECT.py:64).
PAIR: partitions never span key contigs (SVIM_COMBINE.py:24-25), so after the exchange each
rank pairs the key contigs it owns and the results are merged type by type in contig-name
(Python str) order — the order the reference's sort produces.

The path has no device-side exchange step (no RCCL collective is justified for a few MB of host
columns).  One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run sets them.
The exchange itself needs no torch: rank 0 listens on a unix-domain socket named after the job
(MASTER_PORT + run id) inside a directory only the user can enter, checks every peer's uid, and the
tables travel as length-prefixed blobs of typed arrays (CandidateTable.to_wire) — nothing is unpickled.
When the caller has initialised a torch.distributed process group (the CPU tests do, with gloo), that
group's all_gather_object is used instead.
"""
import os
import sys
import time

import numpy as np

from svim_asm_amd.table import CandidateTable


# ------------------------------------------------------------------------------ the group
def _torch_group():
    if "torch.distributed" not in sys.modules:
        return None
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def world():
    d = _torch_group()
    if d is not None:
        return d.get_rank(), d.get_world_size()
    size = int(os.environ.get("WORLD_SIZE", "1"))
    return (int(os.environ.get("RANK", "0")), size) if size > 1 else (0, 1)


def _rendezvous_path():
    """Path of the job's exchange socket: inside a directory only this user can enter (mode 0700, owned by the
    user, checked), named after the job (MASTER_PORT + run id, or SVX_RENDEZVOUS)."""
    import hashlib
    import stat
    import tempfile
    uid = os.getuid()
    base = os.environ.get("XDG_RUNTIME_DIR")
    if not (base and os.path.isdir(base) and os.stat(base).st_uid == uid):
        base = tempfile.gettempdir()
    if len(os.fsencode(base)) + len("/svx-%d/" % uid) + 24 + len(".sock") > 107 and os.path.isdir("/tmp"):
        base = "/tmp"  # a unix socket path holds 107 bytes: a deep TMPDIR would not leave room for the name
    d = os.path.join(base, "svx-%d" % uid)
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != uid or (st.st_mode & 0o077):
        raise RuntimeError("%s must be a directory owned by uid %d with mode 0700 (found mode %o, uid %d): refusing to "
                           "exchange candidate tables through it" % (d, uid, st.st_mode & 0o7777, st.st_uid))
    job = os.environ.get("SVX_RENDEZVOUS") or "svx-%s-%s" % (os.environ.get("MASTER_PORT", "0"),
                                                              os.environ.get("TORCHELASTIC_RUN_ID", "job"))
    return os.path.join(d, hashlib.sha256(job.encode()).hexdigest()[:24] + ".sock")


def _job_token():
    """What the ranks of ONE job share and another job of the same user (same MASTER_PORT, no run id: the same socket
    path) does not: SVX_JOB_TOKEN if set; else SVX_RENDEZVOUS — the documented way to name a job whose ranks are started
    by hand, from different shells or through per-rank wrappers —; else the launcher's run id (torchrun's default, the
    constant "none", names nothing); else the parent process together with the job's address and port — a launcher starts
    all ranks of a node from one agent."""
    token = os.environ.get("SVX_JOB_TOKEN")
    if token:
        return token.encode()
    named = os.environ.get("SVX_RENDEZVOUS")
    if named:
        return ("rendezvous-" + named).encode()
    run_id = os.environ.get("TORCHELASTIC_RUN_ID")
    if run_id and run_id != "none":
        return ("run-" + run_id).encode()
    return ("ppid-%d-%s-%s" % (os.getppid(), os.environ.get("MASTER_ADDR", ""), os.environ.get("MASTER_PORT", ""))).encode()


def _send_msg(sock, parts):
    """One message = 8-byte count of parts, then every part as 8-byte length + bytes."""
    import struct
    sock.sendall(struct.pack("<Q", len(parts)))
    for p in parts:
        sock.sendall(struct.pack("<Q", len(p)))
        sock.sendall(p)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the exchange socket")
        got += k
    return buf


def _recv_msg(sock, max_bytes=1 << 40):
    import struct
    (n_parts,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n_parts > 1 << 20:
        raise ValueError("exchange message with %d parts" % n_parts)
    parts = []
    for _ in range(n_parts):
        (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
        if n > max_bytes:
            raise ValueError("exchange message part of %d bytes" % n)
        parts.append(bytes(_recv_exact(sock, n)))
    return parts


def _encode(obj):
    """(status, payload) of _gather_or_raise as message parts: b"ok" + one wire blob per table, or b"error" + text.
    No pickle: what a peer sends can only become tables (CandidateTable.from_wire) or a string."""
    status, payload = obj
    if status == "error":
        return [b"error", str(payload).encode("utf-8", "replace")]
    tables = payload if isinstance(payload, (list, tuple)) else [payload]
    return [b"ok", b"L" if isinstance(payload, (list, tuple)) else b"1"] + [t.to_wire() for t in tables]


def _decode(parts):
    if parts[0] == b"error":
        return ("error", parts[1].decode("utf-8", "replace"))
    if parts[0] != b"ok" or len(parts) < 2 or parts[1] not in (b"L", b"1"):
        raise ValueError("malformed exchange message")
    tables = [CandidateTable.from_wire(p) for p in parts[2:]]
    return ("ok", tables if parts[1] == b"L" else tables[0])


class _SocketGroup(object):
    """all_gather of (status, tables) over a unix-domain socket: rank 0 collects and redistributes.  The socket
    is a path inside a directory of mode 0700 owned by the user, every peer's uid is checked (SO_PEERCRED), and
    the messages are length-prefixed table blobs (CandidateTable.to_wire) — nothing is unpickled."""

    def __init__(self, rank, size, timeout=120.0):
        import socket
        self.rank, self.size = rank, size
        self.path = _rendezvous_path()
        if rank == 0:
            if os.path.exists(self.path):
                probe = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    probe.connect(self.path)
                except OSError:
                    os.unlink(self.path)  # left behind by a job that died
                else:
                    probe.close()
                    raise RuntimeError("another job of this user is using the rendezvous %s (same MASTER_PORT and run id): "
                                       "set SVX_RENDEZVOUS to a unique name" % self.path)
                finally:
                    probe.close()
            self.listener = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            old = os.umask(0o177)
            try:
                self.listener.bind(self.path)
            finally:
                os.umask(old)
            self.listener.listen(size)
            self.listener.settimeout(timeout)  # a rank that never shows up must not block rank 0 for ever
            self.peers = [None] * size
            deadline = time.time() + timeout
            strangers = 0
            while any(p is None for p in self.peers[1:]):
                left = deadline - time.time()
                try:
                    if left <= 0:
                        raise OSError("timed out")
                    self.listener.settimeout(left)
                    conn, _ = self.listener.accept()
                except OSError as e:  # socket.timeout
                    self.close()
                    raise RuntimeError("rank 0: only %d of %d ranks reached the exchange within %.0f s (%s; %d connection(s) "
                                       "of other jobs turned away — ranks started from different shells share SVX_JOB_TOKEN)"
                                       % (1 + sum(p is not None for p in self.peers), size, timeout, e, strangers))
                try:
                    self._check_peer(conn)
                    # the announcement follows the connect at once: a peer that connects and says nothing is given five
                    # seconds (never more than the rendezvous has left), then the next connection is looked at
                    conn.settimeout(max(0.05, min(5.0, deadline - time.time())))
                    hello = _recv_msg(conn, 256)
                    r = int(hello[0])
                    token = hello[1] if len(hello) > 1 else b""
                except (OSError, ValueError, IndexError, RuntimeError):
                    conn.close()
                    strangers += 1
                    continue
                if token != _job_token() or not 0 < r < size or self.peers[r] is not None:
                    conn.close()  # a rank of ANOTHER job that found this listener (or a duplicate): turned away, not fatal
                    strangers += 1
                    continue
                conn.settimeout(None)
                self.peers[r] = conn
        else:
            deadline = time.time() + timeout
            while True:
                self.conn = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    self.conn.connect(self.path)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    self.conn.close()
                    if time.time() > deadline:
                        raise RuntimeError("rank %d: rank 0 did not open the exchange socket within %.0f s" % (rank, timeout))
                    time.sleep(0.005)
            self._check_peer(self.conn)
            _send_msg(self.conn, [str(rank).encode(), _job_token()])

    @staticmethod
    def _check_peer(conn):
        import socket
        import struct
        if hasattr(socket, "SO_PEERCRED"):
            _pid, uid, _gid = struct.unpack("3i", conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i")))
            if uid != os.getuid():
                conn.close()
                raise RuntimeError("exchange socket: peer runs as uid %d, this job as %d" % (uid, os.getuid()))

    def all_gather(self, obj):
        mine = _encode(obj)
        if self.rank == 0:
            everything = [mine] + [None] * (self.size - 1)
            for r in range(1, self.size):
                everything[r] = _recv_msg(self.peers[r])
            flat = [str(len(m)).encode() for m in everything] + [p for m in everything for p in m]
            for r in range(1, self.size):
                _send_msg(self.peers[r], flat)
            return [obj] + [_decode(m) for m in everything[1:]]
        _send_msg(self.conn, mine)
        flat = _recv_msg(self.conn)
        counts = [int(x) for x in flat[:self.size]]
        out, at = [], self.size
        for r, c in enumerate(counts):
            out.append(obj if r == self.rank else _decode(flat[at:at + c]))
            at += c
        return out

    def close(self):
        if self.rank == 0:
            for c in getattr(self, "peers", [])[1:]:
                if c is not None:
                    c.close()
            self.listener.close()
            try:
                os.unlink(self.path)
            except OSError:
                pass
        else:
            self.conn.close()


_group = None


def _all_gather(obj):
    global _group
    d = _torch_group()
    if d is not None:
        gathered = [None] * d.get_world_size()
        d.all_gather_object(gathered, obj)
        return gathered
    if _group is None:
        rank, size = world()
        _group = _SocketGroup(rank, size)
    return _group.all_gather(obj)


def shutdown():
    global _group
    if _group is not None:
        _group.close()
        _group = None


def _gather_or_raise(compute):
    """all-gather of compute()'s result.  A rank whose compute() raises still takes part in the exchange
    (with an error marker), so the others do not wait for it until a timeout; every rank then raises."""
    rank = world()[0]
    try:
        mine = ("ok", compute())
    except Exception as e:  # noqa: BLE001 — forwarded to every rank below
        import traceback
        mine = ("error", "rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
    gathered = _all_gather(mine)
    errors = [payload for status, payload in gathered if status == "error"]
    if errors:
        raise RuntimeError("sharded step failed on %d of %d ranks:\n%s" % (len(errors), len(gathered), "\n".join(errors)))
    return [payload for _, payload in gathered]


# ------------------------------------------------------------------------------ the plan
def lpt_assign(weights, n_bins):
    """Longest-processing-time bin packing.  Returns bin index per item; deterministic
    (ties broken by item index) so every rank computes the same plan without communication."""
    order = sorted(range(len(weights)), key=lambda i: (-int(weights[i]), i))
    load = [0] * n_bins
    owner = [0] * len(weights)
    for i in order:
        b = min(range(n_bins), key=lambda k: (load[k], k))
        owner[i] = b
        load[b] += int(weights[i])
    return owner


def contig_weights(bam):
    """Work per contig for the rank plan.  With a usable `.bai` the compressed bytes the index
    attributes to each contig (ingest dominates COLLECT and nothing has to be inflated to know
    them); otherwise CIGAR ops per contig, which needs the whole file indexed first."""
    spans = getattr(bam, "contig_spans", None)
    if spans is not None:
        w = spans()
        if w is not None:
            return w
    cols = getattr(bam, "_cols", None)
    n = len(bam.references)
    w = np.zeros(n, dtype=np.int64)
    if cols is not None:
        tid = cols["tid"]
        ok = (tid >= 0) & (tid < n)
        np.add.at(w, tid[ok], cols["n_cig"][ok] + 1)
        return w
    for i, name in enumerate(bam.references):
        for aln in bam.fetch(contig=name):
            words = getattr(aln, "cigar_words", None)
            w[i] += (len(words) if words is not None else len(aln.cigartuples or ())) + 1
    return w


class ContigView(object):
    """A BAM restricted to a subset of contigs for the COLLECT loop; header lookups see the full
    reference dictionary."""

    def __init__(self, bam, contig_indices):
        self._bam = bam
        self.references = tuple(bam.references[i] for i in contig_indices)
        self.lengths = tuple(bam.lengths[i] for i in contig_indices)

    def __getattr__(self, name):
        return getattr(self._bam, name)


# ------------------------------------------------------------------------------ COLLECT / PAIR
def collect_sharded(bams, options, collect_fn=None):
    """Distributed COLLECT of the haplotype BAMs of one sample: the CandidateTable of every bam, the same
    on every rank."""
    if collect_fn is None:
        from svim_asm_amd.SVIM_COLLECT import collect_tables as collect_fn
    bams = list(bams)
    rank, size = world()
    if size == 1:
        return collect_fn(bams, options)
    weights = sum(np.asarray(contig_weights(b), dtype=np.int64) for b in bams)
    owner = lpt_assign(weights, size)
    mine = [i for i, r in enumerate(owner) if r == rank]

    def local():
        # a view makes COLLECT walk / inflate only the BGZF ranges of the contigs this rank owns
        tables = collect_fn([ContigView(b, mine) for b in bams], options)
        for t in tables:
            if getattr(t, "rec_tid", None) is None:
                raise RuntimeError("COLLECT tables of a sharded run need the record contig of every row")
        return tables

    parts = _gather_or_raise(local)
    out = []
    for k, bam in enumerate(bams):
        merged = CandidateTable.concat([p[k] for p in parts], list(bam.references),
                                       [bam.get_reference_length(c) for c in bam.references])
        rec_tid = np.concatenate([p[k].rec_tid for p in parts])
        # every rank's rows are in header order for its own contigs, and the contigs are disjoint
        o = np.argsort(rec_tid, kind="stable")
        merged = merged.take(o)
        merged.rec_tid = rec_tid[o]
        out.append(merged)
    return out


def pair_sharded(table1, table2, reference, bam, options, pair_fn=None):
    """Distributed pair_candidates on tables: the same CandidateTable on every rank."""
    if pair_fn is None:
        from svim_asm_amd.SVIM_COMBINE import pair_tables as pair_fn
    rank, size = world()
    if size == 1:
        return pair_fn(table1, table2, reference, bam, options)
    from svim_asm_amd.SVIM_COMBINE import _str_rank
    # key contigs by NAME (the two tables may number contigs differently)
    names1 = np.array(table1.contigs, dtype=object)[table1.key_contig()] if len(table1) else np.zeros(0, object)
    names2 = np.array(table2.contigs, dtype=object)[table2.key_contig()] if len(table2) else np.zeros(0, object)
    counts = {}
    for name in names1.tolist() + names2.tolist():
        counts[name] = counts.get(name, 0) + 1
    contigs = sorted(counts)
    owner = dict(zip(contigs, lpt_assign([counts[c] for c in contigs], size)))
    mine1 = np.flatnonzero(np.fromiter((owner[c] == rank for c in names1.tolist()), dtype=bool, count=len(names1)))
    mine2 = np.flatnonzero(np.fromiter((owner[c] == rank for c in names2.tolist()), dtype=bool, count=len(names2)))
    gathered = _gather_or_raise(lambda: pair_fn(table1.take(mine1), table2.take(mine2), reference, bam, options))
    base = getattr(bam, "_bam", bam)
    everything = CandidateTable.concat(gathered, list(base.references), [base.get_reference_length(c) for c in base.references])
    # reference order: type by type, (contig name, position) inside — every rank's part is sorted already
    o = np.lexsort((_str_rank(everything.contigs)[everything.key_contig()], everything.type))
    return everything.take(o)
