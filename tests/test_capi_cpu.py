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


def declared_symbols(header="vsel.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef VSEL_TRACE.*?#endif", "", text, flags=re.S)      # only in -DVSEL_TRACE builds (tools/trace_*.py)
    return sorted(set(re.findall(r"\b(vsel_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(libpath):
    lib = ctypes.CDLL(libpath)
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vsel.h but not exported by libvsel.so"


def test_every_exported_symbol_is_declared(libpath):
    """No undeclared entry points: every exported vsel_* symbol is in include/vsel.h or include/vsel_debug.h."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("vsel_")})
    declared = set(declared_symbols("vsel.h")) | set(declared_symbols("vsel_debug.h"))
    assert exported and set(exported) == declared, sorted(set(exported) ^ declared)


def test_debug_knobs_set_get_restore(libpath):
    from visionselector_amd import _native as N
    lib = N.lib()
    assert sorted(N.DEBUG_SIGNATURES) == declared_symbols("vsel_debug.h")
    lib.vsel_debug_reset()
    defaults = {n: N.debug_get(n) for n in N.KNOBS}
    assert defaults["lis_small_path"] == int(os.environ.get("VSEL_SMALL_PATH", 4)) and defaults["attn_bwd_split"] == -1
    with N.debug_knob("lis_small_path", 100):
        assert N.debug_get("lis_small_path") == 8              # clamped to the knob's range
        with N.debug_knob(attn_waves=8, attn_bwd_split=1):
            assert N.debug_get("attn_waves") == 8 and N.debug_get("attn_bwd_split") == 1
        assert N.debug_get("attn_waves") == 0 and N.debug_get("attn_bwd_split") == -1
    with pytest.raises(ZeroDivisionError):
        with N.debug_knob("attn_pack", 0):
            1 / 0
    assert {n: N.debug_get(n) for n in N.KNOBS} == defaults    # restored, also after an exception
    assert lib.vsel_debug_set(99, 1, None) == 1 and b"unknown knob" in lib.vsel_last_error()
    with pytest.raises(KeyError):
        N.debug_knob("nope", 1)
    prev = ctypes.c_int(-5)
    assert lib.vsel_debug_set(N.KNOBS["attn_pack"], 1, ctypes.byref(prev)) == 0 and prev.value == 2
    lib.vsel_debug_reset()
    assert N.debug_get("attn_pack") == 2
    text = open(os.path.join(ROOT, "include", "vsel_debug.h")).read()
    for name, idx in N.KNOBS.items():                           # the Python table mirrors the header's enum
        assert re.search(rf"VSEL_KNOB_{name.upper()} = {idx}\b", text), name


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
