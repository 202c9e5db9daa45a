import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "spawns_gpu_children: starts fresh rank processes that make their own first GPU "
                                       "call; scheduled before the tests that use the GPU in the pytest process")


def pytest_collection_modifyitems(config, items):
    # tests that start rank processes of their own go first: their children then come up on an idle device.  (A
    # parent that HAS initialised the GPU may still start fresh child processes — bench.py's BAM -> VCF legs and
    # tests/test_gpu_ctx.py do; what must never happen is an os.exec* of a process that has touched the GPU.)
    first = [i for i in items if i.get_closest_marker("spawns_gpu_children")]
    rest = [i for i in items if not i.get_closest_marker("spawns_gpu_children")]
    items[:] = first + rest


@pytest.fixture(scope="session")
def svx_ctx():
    """HIP context on cuda:0 through the C-ABI; fails loudly when no device (no CPU fallback)."""
    from svim_asm_amd import _lib
    return _lib.Context(0)
