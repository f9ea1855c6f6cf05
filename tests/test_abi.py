"""CPU-only: the C-ABI library loads and exports every symbol include/st3r.h declares, and
the ctypes table in starst3r_amd/_lib.py covers exactly that set (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "st3r.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(st3r_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    from starst3r_amd import build
    return build.build()


def test_header_declares_functions():
    fns = header_functions()
    assert "st3r_gs_train_fwd_bwd" in fns and "st3r_adam_step" in fns and len(fns) >= 15


def test_library_exports_every_declared_symbol(built_lib):
    L = ctypes.CDLL(built_lib)
    for fn in header_functions():
        assert hasattr(L, fn), f"{fn} declared in include/st3r.h but not exported by libst3r_hip.so"
    L.st3r_version.restype = ctypes.c_int
    assert L.st3r_version() >= 100


def test_ctypes_table_matches_header(built_lib):
    from starst3r_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()
    _lib.lib()  # binds every signature; raises if a symbol is missing


def test_product_fails_loudly_without_gpu():
    import torch
    from starst3r_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.St3rError):
        ops.Context("cpu")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "starst3r_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "oracle/_build" not in txt and "libgs_oracle" not in txt, f
