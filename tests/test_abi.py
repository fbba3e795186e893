"""CPU tests of the drop-in boundary: libvgl_hip.so loads without a GPU, exports every symbol include/vgl_hip.h
declares, the ctypes table matches the header, and the product fails loudly (no CPU fallback) when no device exists."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vgl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vgl_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from vectorgraphlibrary_amd import lib
    L = ctypes.CDLL(lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/vgl_hip.h but not exported"
    assert sorted(lib.EXPORTED_SYMBOLS) == syms, "ctypes table and header disagree"
    assert L.vgl_hip_abi_version() == 1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vectorgraphlibrary_amd import lib
    from vectorgraphlibrary_amd.api import Context
    with pytest.raises(lib.VglHipError):
        Context(0)
    L = lib.load()
    h = ctypes.c_void_p()
    assert L.vgl_hip_ctx_create(0, None, ctypes.byref(h)) != 0
    assert b"vgl_hip" in L.vgl_hip_last_error()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "vectorgraphlibrary_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "vgl_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
