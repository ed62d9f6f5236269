"""`replace_qwen25vl(args, model, method)` / `replace_llavaov15(args, model, method)` with the reference's call shape
(qwen-evaluation/token_compression/monkeypatch.py:50-107, llava-ov-15/compression_method/monkeypatch.py:34-70).

In the reference, `selector` (and `dynamic`) have NO branch here: they are selected by loading the `*_Selector` model
class (lmms-eval/lmms_eval/models/qwen2_5_vl_with_token_compression.py:114-117); every branch that exists is one of the
third-party baselines (FastV, VisionZip, PruMerge+, DivPrune, DART, HoloV), which are out of scope of this library.

The drop-in packages (`dropin/token_compression`, `dropin/compression_method`) are namespace portions: when the
reference's own package root is also on sys.path, its baseline modules still resolve behind ours, and a baseline method
name is forwarded to the reference's monkeypatch module found there (`forward_to_reference`)."""
from __future__ import annotations

import importlib.util
import os
import sys

_BASELINES = ("visionzip", "fastv", "prumerge+", "divprune", "dart", "holov", "visionzip_official")
_PASSTHROUGH = (None, "", "selector", "dynamic", "origin", "none")


def forward_to_reference(package: str, own_file: str, func: str):
    """The function `func` of another `monkeypatch.py` in a later portion of namespace package `package`, or None.
    The module is loaded as `<package>._reference_monkeypatch` so that its relative imports (`from .visionzip import`,
    qwen-evaluation/token_compression/monkeypatch.py:10) resolve inside the same namespace package."""
    pkg = sys.modules.get(package)
    own_dir = os.path.dirname(os.path.abspath(own_file))
    for portion in list(getattr(pkg, "__path__", ())):
        cand = os.path.join(portion, "monkeypatch.py")
        if os.path.abspath(portion) == own_dir or not os.path.isfile(cand):
            continue
        name = f"{package}._reference_monkeypatch"
        mod = sys.modules.get(name)
        if mod is None:
            spec = importlib.util.spec_from_file_location(name, cand)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                del sys.modules[name]
                raise
        return getattr(mod, func)
    return None


def _replace(args, model, method, who, reference_fn=None):
    if method in _PASSTHROUGH:
        return model                      # same as the reference: nothing to patch for these
    if method in _BASELINES:
        if reference_fn is not None:
            return reference_fn(args, model, method)
        raise NotImplementedError(
            f"{who}: '{method}' is a third-party baseline compressor of the reference and is not part of "
            "visionselector_amd (only the VisionSelector LIS path is implemented); put the reference's package root "
            "behind dropin/ on sys.path to reach its own branch.")
    raise ValueError(f"{who}: unknown compression method '{method}'")


def replace_qwen25vl(args, model, method, _reference_fn=None):
    return _replace(args, model, method, "replace_qwen25vl", _reference_fn)


def replace_llavaov15(args, model, method, _reference_fn=None):
    return _replace(args, model, method, "replace_llavaov15", _reference_fn)
