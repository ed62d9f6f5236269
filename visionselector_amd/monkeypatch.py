"""`replace_qwen25vl(args, model, method)` / `replace_llavaov15(args, model, method)` with the reference's call shape
(qwen-evaluation/token_compression/monkeypatch.py:50-107, llava-ov-15/compression_method/monkeypatch.py).

In the reference, `selector` (and `dynamic`) have NO branch here: they are selected by loading the `*_Selector` model
class (lmms-eval/lmms_eval/models/qwen2_5_vl_with_token_compression.py:114-117); every branch that exists is one of the
third-party baselines (FastV, VisionZip, PruMerge+, DivPrune, DART, HoloV), which are out of scope of this library."""
from __future__ import annotations

_BASELINES = ("visionzip", "fastv", "prumerge+", "divprune", "dart", "holov", "visionzip_official")


def _replace(args, model, method, who):
    if method in (None, "", "selector", "dynamic", "origin", "none"):
        return model                      # same as the reference: nothing to patch for these
    if method in _BASELINES:
        raise NotImplementedError(
            f"{who}: '{method}' is a third-party baseline compressor of the reference and is not part of "
            "visionselector_amd (only the VisionSelector LIS path is implemented).")
    raise ValueError(f"{who}: unknown compression method '{method}'")


def replace_qwen25vl(args, model, method):
    return _replace(args, model, method, "replace_qwen25vl")


def replace_llavaov15(args, model, method):
    return _replace(args, model, method, "replace_llavaov15")
