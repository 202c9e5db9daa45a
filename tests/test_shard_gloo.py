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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import orc, run_oracle, svim_oracle
        from svim_asm_amd import SVCandidate, bamio, fasta, shard
        from tests import helpers
        o = helpers.options()
        ref = fasta.FastaFile(os.path.join(GOLD, "ref.fa"))

        def collect_fn(bam_view, options):
            # oracle-backed stand-in for the GPU COLLECT, restricted to the view's contigs
            recs, names, lengths = run_oracle.read_records(bam_view.filename)
            keep = {names.index(n) for n in bam_view.references}
            tuples = svim_oracle.collect([r for r in recs if r["tid"] in keep], names, lengths, options)
            return [helpers.build_candidate(t, bam_view, SVCandidate) for t in tuples]

        def pair_fn(c1, c2, reference, bam, options):
            names, lengths = list(bam.references), list(bam.lengths)
            t = svim_oracle.pair_candidates([helpers.candidate_tuple(c) for c in c1],
                                            [helpers.candidate_tuple(c) for c in c2], reference.fetch, names, lengths,
                                            dict(zip(reference.references, reference.lengths)), options,
                                            edit=lambda a, b: orc.edit_distance(a.encode(), b.encode()))
            return [helpers.build_candidate(x, bam, SVCandidate) for x in t]

        b1 = bamio.AlignmentFile(os.path.join(GOLD, "hap1.bam"))
        b2 = bamio.AlignmentFile(os.path.join(GOLD, "hap2.bam"))
        c1 = shard.collect_sharded(b1, o, collect_fn)
        c2 = shard.collect_sharded(b2, o, collect_fn)
        paired = shard.pair_sharded(c1, c2, ref, b1, o, pair_fn)
        q.put((rank, [helpers.candidate_tuple(c) for c in c1], [helpers.candidate_tuple(c) for c in paired]))
    finally:
        dist.destroy_process_group()


def test_sharded_collect_and_pair_equal_single_process():
    import torch.multiprocessing as mp
    from oracle import orc, run_oracle, svim_oracle
    from tests import helpers
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
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
    for rank, c1, paired in results:
        assert c1 == exp1, "rank %d collect order differs" % rank
        assert paired == exp_pair, "rank %d pair order differs" % rank
