"""CPU-only checks of the product library: the C ABI shared object loads, exports every symbol include/lsfm.h
declares, and refuses to run without a device (no CPU fallback)."""
import os
import re

import pytest


def _header_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "lsfm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lsfm_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_whole_c_abi():
    from linearsfm_amd import api
    if not os.path.exists(api.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = api.lib()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(api.EXPORTS) == syms


def test_context_fails_loudly_without_gpu():
    import torch
    from linearsfm_amd import api
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.LsfmError):
        api.Context(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under linearsfm_amd/ or include/ may reference it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for base in ("linearsfm_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(root, base)):
            if "build" in dp:
                continue
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|oracle/|lsfm_oracle|pyoracle", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_rccl_hook_library_exports_its_header():
    """liblsfm_rccl.so (RCCL behind lsfm_allreduce_fn for C / C++ hosts) loads and exports what include/lsfm_rccl.h declares."""
    import ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "linearsfm_amd", "liblsfm_rccl.so")
    if not os.path.exists(path):
        import __graft_entry__ as g
        g.build()
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "lsfm_rccl.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(lsfm_rccl_[a-z_0-9]+)\s*\(", txt)))
    assert len(syms) >= 6
    L = ctypes.CDLL(path)
    for s in syms:
        assert hasattr(L, s), s
