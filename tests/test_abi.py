"""C-ABI surface: libsvx.so loads on a CPU-only box and exports every symbol include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include")))
                   if h.endswith(".h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for name in ("svx_ctx_create", "svx_cigar_extract", "svx_cigar_extract_dev", "svx_cigar_extract_soa",
                 "svx_segments_classify", "svx_pair_partition", "svx_edit_distance_batch", "svx_last_error"):
        assert name in syms


def test_library_exports_every_declared_symbol():
    from svim_asm_amd import _lib, build
    build.build_lib()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, "libsvx.so lacks: %s" % missing
    # and the ctypes binding table covers the same set
    assert sorted(_lib.SYMBOLS) == _declared_symbols()


def test_version_and_device_count_without_gpu():
    from svim_asm_amd import _lib
    lib = _lib.load()
    assert lib.svx_version().startswith(b"svx")
    assert lib.svx_device_count() >= 0


def test_no_cpu_fallback_without_device():
    """Without a HIP device the product path must fail loudly, not compute on the CPU."""
    from svim_asm_amd import _lib
    lib = _lib.load()
    if lib.svx_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_lib.SvxError) as ei:
        _lib.Context(0)
    assert ei.value.status == _lib.SVX_E_NODEVICE


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "svim_asm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libsvx_oracle" not in src, f
