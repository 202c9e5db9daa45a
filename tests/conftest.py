import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def svx_ctx():
    """HIP context on cuda:0 through the C-ABI; fails loudly when no device (no CPU fallback)."""
    from svim_asm_amd import _lib
    return _lib.Context(0)
