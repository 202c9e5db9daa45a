"""bench.py --gpus N: the line carries, per rank, where it ran and how long ITS steps took, the backend and the sum the
probe all-reduce returned — enough to verify a multi-GPU record without trusting the launcher.  World size 2
over gloo on the CPU: the gathering and the schema; no device involved (a fake device record stands in)."""
import importlib.util
import multiprocessing as mp
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _record(rank, **kw):
    r = dict(rank=rank, local_rank=rank, device_index=rank, device_uuid="GPU-%04d" % rank, ms_per_step=0.35 + rank * 0.01,
             sustained_ms=0.31, ops_per_step=395_000_000, host="box", pid=1000 + rank)
    r.update(kw)
    return r


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bench = _bench()
        probe = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(probe)
        recs = bench.gather_rank_records(dist, world, _record(rank))
        q.put((rank, bench.multi_rank_fields(recs, "gloo", int(probe.item()), world)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_fill_the_schema():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1]  # every rank holds the same picture
    f = got[0]
    assert f["backend"] == "gloo" and f["collective_world_verified"] == 2 and f["distinct_devices"] == 2
    assert [r["rank"] for r in f["ranks"]] == [0, 1]
    bench = _bench()
    for r in f["ranks"]:
        assert all(k in r for k in bench.RANK_RECORD_KEYS)
        assert r["ms_per_step"] > 0 and r["device_uuid"].startswith("GPU-")


def test_single_rank_line_has_the_same_fields():
    bench = _bench()
    f = bench.multi_rank_fields(bench.gather_rank_records(None, 1, _record(0)), "nccl", 1, 1)
    assert f["backend"] is None and f["collective_world_verified"] == 1 and len(f["ranks"]) == 1 and f["distinct_devices"] == 1


def test_holes_are_refused_and_shared_devices_are_counted_once():
    bench = _bench()
    with pytest.raises(RuntimeError):
        bench.multi_rank_fields([_record(0), None], "nccl", 2, 2)
    bad = _record(1)
    del bad["ms_per_step"]
    with pytest.raises(RuntimeError):
        bench.multi_rank_fields([_record(0), bad], "nccl", 2, 2)
    with pytest.raises(RuntimeError):
        bench.multi_rank_fields([_record(1), _record(0)], "nccl", 2, 2)  # out of order
    shared = [_record(0), _record(1, device_index=0, device_uuid="GPU-0000")]
    assert bench.multi_rank_fields(shared, "nccl", 2, 2)["distinct_devices"] == 1
