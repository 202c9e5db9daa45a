"""Context plumbing of the C-ABI (include/svx.h): caller-owned streams, timing, device buffers, pipelining of two
contexts, error returns."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc
from svim_asm_amd import _lib, synth

pytestmark = pytest.mark.gpu
KEYS = ("aln", "ref_pos", "read_pos", "len", "type")


def _dev_case(ctx, cig, off, rs, cap):
    bufs = dict(cig=ctx.dev_array(cig), off=ctx.dev_array(off.astype(np.uint64)), rs=ctx.dev_array(rs),
                outs=[ctx.dev_array(nbytes=4 * cap) for _ in range(4)] + [ctx.dev_array(nbytes=cap)],
                n=ctx.dev_array(np.zeros(1, np.uint64)))
    return bufs


def _run(ctx, b, n_ops, n_aln, cap):
    ctx.cigar_extract_dev(b["cig"].ptr, n_ops, b["off"].ptr, n_aln, b["rs"].ptr, 40, tuple(o.ptr for o in b["outs"]), cap,
                          b["n"].ptr)


def _check(b, exp):
    n = int(b["n"].download(np.uint64)[0])
    assert n == len(exp["aln"])
    for o, key in zip(b["outs"], KEYS):
        assert np.array_equal(o.download(np.uint8 if key == "type" else np.uint32, n), exp[key]), key


def test_context_on_the_default_stream_and_error_returns():
    ctx = _lib.Context(0, stream=0)  # svx_ctx_create_on_stream(device, NULL): the default stream
    rng = np.random.default_rng(2)
    cig, off, rs = synth.random_cigar_case(rng, 300, max_ops=500)
    got = ctx.cigar_extract(cig, off, rs, 40)
    exp = orc.cigar_extract(cig, off, rs, 40)
    for k in KEYS:
        assert np.array_equal(got[k], exp[k])
    ctx.close()
    h = C.c_void_p()
    assert _lib.load().svx_ctx_create(4096, C.byref(h)) == _lib.SVX_E_NODEVICE and not h.value
    assert _lib.load().svx_ctx_sync(None) == _lib.SVX_E_INVALID
    assert _lib.load().svx_device_count() >= 1
    assert b"svx" in _lib.load().svx_version()


def test_timing_brackets_the_last_device_call(svx_ctx):
    rng = np.random.default_rng(3)
    cig, off, rs = synth.random_cigar_case(rng, 2000, max_ops=3000)
    exp = orc.cigar_extract(cig, off, rs, 40)
    cap = len(exp["aln"]) + 8
    b = _dev_case(svx_ctx, cig, off, rs, cap)
    with pytest.raises(_lib.SvxError):
        svx_ctx.last_kernel_ms()  # nothing timed yet
    svx_ctx.set_timing(True)
    try:
        for ops in (1 << 23, 0):  # both kernel paths
            svx_ctx.set_small_batch_ops(ops)
            _run(svx_ctx, b, len(cig), len(off) - 1, cap)
            svx_ctx.sync()
            total, dominant = svx_ctx.last_kernel_ms()
            assert 0 < dominant <= total < 1000
            _check(b, exp)
    finally:
        svx_ctx.set_timing(False)
        svx_ctx.set_small_batch_ops(1 << 23)


def test_two_contexts_pipelined_with_wait_dominant():
    """svx_ctx_wait_dominant orders a context's launches after the other's streaming kernel (bench.py --pipeline);
    results are those of the un-pipelined calls."""
    a, b = _lib.Context(0), _lib.Context(0)
    rng = np.random.default_rng(4)
    cases = []
    for ctx in (a, b):
        ctx.set_small_batch_ops(1 << 21)
        cig, off, rs = synth.random_cigar_case(rng, 30000, max_ops=400)  # > 2 M ops: streaming path
        exp = orc.cigar_extract(cig, off, rs, 40)
        cap = len(exp["aln"]) + 8
        cases.append((ctx, cig, off, _dev_case(ctx, cig, off, rs, cap), cap, exp))
    assert len(cases[0][1]) > (1 << 21)
    for rep in range(6):
        ctx, cig, off, bufs, cap, _ = cases[rep % 2]
        ctx.wait_dominant(cases[(rep + 1) % 2][0])
        _run(ctx, bufs, len(cig), len(off) - 1, cap)
    for ctx, _, _, bufs, _, exp in cases:
        ctx.sync()
        _check(bufs, exp)
    a.close()
    b.close()


def test_device_buffers_round_trip(svx_ctx):
    x = np.arange(100003, dtype=np.uint32) * np.uint32(2654435761)
    d = svx_ctx.dev_array(x)
    assert np.array_equal(d.download(np.uint32), x)
    assert np.array_equal(d.download(np.uint32, 17), x[:17])
    d.free()
    p = C.c_void_p()
    assert svx_ctx.lib.svx_dev_malloc(svx_ctx.h, 0, C.byref(p)) == 0 and p.value  # zero bytes: a valid pointer
    assert svx_ctx.lib.svx_dev_free(svx_ctx.h, p) == 0
    assert svx_ctx.lib.svx_dev_free(svx_ctx.h, None) == 0
    assert svx_ctx.lib.svx_dev_upload(svx_ctx.h, None, x.ctypes.data, 16) == _lib.SVX_E_INVALID


_BARRIER_CHILD = r'''
import numpy as np
from oracle import orc
from svim_asm_amd import _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(4)
big = (rng.integers(0, 100, 70_000).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 28, 70_000).astype(np.uint64)
small = big[:700]
b_perm, b_part, b_n = orc.pair_partition(big, 1000)
retries = 0
for limit in (131072, 0):                       # one-launch plan (59 workgroups), radix plan (k_partition: 18)
    ctx.set_pair_single_launch_max(limit)
    for rep in range(2):
        # every wait of this build runs out at once: the host-pointer call notices, re-runs itself on the plan
        # without waits inside a launch and SUCCEEDS with the reference's answer
        perm, part, n_parts = ctx.pair_partition(big, 1000)
        assert n_parts == b_n and np.array_equal(perm, b_perm) and np.array_equal(part, b_part)
        retries += 1
        assert ctx.pair_retries() == retries
        perm, part, n_parts = ctx.pair_partition(small, 1000)   # one workgroup: nobody to wait for, no retry
        e_perm, e_part, e_n = orc.pair_partition(small, 1000)
        assert n_parts == e_n and np.array_equal(perm, e_perm) and np.array_equal(part, e_part)
        assert ctx.pair_retries() == retries
# the asynchronous entry reports at the next synchronisation
d_k, d_p, d_id = ctx.dev_array(big), ctx.dev_array(nbytes=4 * len(big)), ctx.dev_array(nbytes=4 * len(big))
d_np = ctx.dev_array(np.zeros(1, np.uint32))
ctx.set_pair_single_launch_max(131072)
ctx._check(ctx.lib.svx_pair_partition_dev_bits(ctx.h, d_k.ptr, len(big), 1000, int(np.bitwise_or.reduce(big)), d_p.ptr,
                                               d_id.ptr, d_np.ptr))
try:
    ctx.sync()
    raise SystemExit("svx_ctx_sync did not report the failed launch")
except _lib.SvxError as e:
    assert "ran out" in str(e), str(e)
assert ctx.barrier_timed_out() and not ctx.barrier_timed_out()   # the caller-visible flag, cleared by reading it
ctx.sync()
# ... and an asynchronous caller that cannot be re-run behind its back sees the flag and selects the wait-free plan itself
ctx.set_pair_wait_free(True)
ctx._check(ctx.lib.svx_pair_partition_dev_bits(ctx.h, d_k.ptr, len(big), 1000, int(np.bitwise_or.reduce(big)), d_p.ptr,
                                               d_id.ptr, d_np.ptr))
ctx.sync()
assert np.array_equal(d_p.download(np.uint32), b_perm) and np.array_equal(d_id.download(np.uint32), b_part)
assert int(d_np.download(np.uint32)[0]) == b_n
print("barrier child ok")
'''


def test_a_wait_between_workgroups_that_runs_out_is_an_error_not_a_hang(tmp_path):
    """The kernels that wait for their other workgroups inside a launch (k_pair_single, k_partition) bound the
    wait; a build whose bound is zero turns every such wait into the failure: svx_pair_partition re-runs the call
    on the wait-free plan and succeeds with the oracle's output; the next svx_ctx_sync after an asynchronous call
    returns SVX_E_HIP with a text; nothing hangs, the context works on."""
    import os
    import subprocess
    import sys
    from svim_asm_amd import build
    try:
        build._hipcc()
    except RuntimeError:
        pytest.skip("no hipcc on this box: the zero-bound build cannot be made")
    lib = build.build_lib(out=str(tmp_path / "libsvx_wait0.so"), defines=["-DSVX_EXP_BARRIER_TICKS=0ull"])
    env = dict(os.environ, SVX_LIB=lib, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = subprocess.run([sys.executable, "-c", _BARRIER_CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=600)
    assert res.returncode == 0 and "barrier child ok" in res.stdout, res.stdout
