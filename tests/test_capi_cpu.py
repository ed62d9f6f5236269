"""CPU-side checks of the boundary: libvsel.so builds/loads and exports every symbol include/vsel.h declares;
the product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from visionselector_amd import build
    return build.build_native(verbose=False)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vsel.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vsel_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(libpath):
    lib = ctypes.CDLL(libpath)
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vsel.h but not exported by libvsel.so"


def test_ctypes_signatures_cover_header(libpath):
    from visionselector_amd import _native
    assert sorted(_native.SIGNATURES) == declared_symbols()
    lib = _native.lib()
    assert lib.vsel_version().decode().startswith("vsel")


def test_argument_validation_without_gpu(libpath):
    """Pure host-side validation paths (they return before any HIP call)."""
    from visionselector_amd import _native as N
    lib = N.lib()
    seg = N.Segments(1, 8, 8, 9, 9, None, None)          # k > rows
    assert lib.vsel_topk_select(None, 16, ctypes.byref(seg), 16, None) == 1
    assert b"k=9" in lib.vsel_last_error()
    assert lib.vsel_soft_topk_fwd(None, 16, 1, 8, 8, 16, 16) == 1        # needs 0 < k < n (reference assert)
    assert lib.vsel_soft_topk_fwd(None, 16, 1, 8, 0, 16, 16) == 1
    seg = N.Segments(2, 64, 128, 4, 8, None, None)
    assert lib.vsel_lis_workspace_bytes(ctypes.byref(seg), 3584, 1792) > 0
    sc = N.Scorer(16, 16, 16, 16, 60, 32, 0)              # D not a multiple of 8
    assert lib.vsel_lis_scores(None, 16, 0, ctypes.byref(seg), ctypes.byref(sc), 16, 1 << 30, 16) == 4
    sc = N.Scorer(16, 16, 16, 16, 64, 32, 0)
    assert lib.vsel_lis_scores(None, 16, 0, ctypes.byref(seg), ctypes.byref(sc), 16, 8, 16) == 2   # workspace too small


def test_ops_refuse_cpu_tensors(libpath):
    from visionselector_amd import ops
    h = torch.zeros(8, 64)
    w = torch.zeros(32, 64)
    b = torch.zeros(32)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.lis_select(h, w, b, w, b, 2)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.soft_topk_fwd(torch.zeros(1, 8), 2)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from visionselector_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU"):
        _native.lib()
