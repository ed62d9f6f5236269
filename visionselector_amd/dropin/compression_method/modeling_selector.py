"""Import-path shim for `compression_method.modeling_selector` (reference: llava-ov-15/compression_method/modeling_selector.py).

The reference module defines three classes by subclassing the OV model code it vendors (`llavaonevision1_5`), and its
consumer imports them by name:

    from compression_method.modeling_selector import LLaVAOneVision1_5_ForConditionalGeneration_Selector
    (lmms-eval/lmms_eval/models/llava_onevision1_5_with_token_compression.py:20, used at :105-108)

This shim resolves the same three names lazily (PEP 562): on first access it imports
`llavaonevision1_5.modeling_llavaonevision1_5` from the caller's path -- the same import the reference module runs at
`modeling_selector.py:6-17` -- and builds the classes once on those bases with `make_llavaov15_selector_classes`.
Nothing of the OV model code is needed just to import this module."""
from visionselector_amd.selector import TopK, _find_ts, topk  # noqa: F401
from visionselector_amd.selector import TransformerScorer  # noqa: F401
from visionselector_amd.hf_llavaov15 import (  # noqa: F401
    llavaov15_vision_tower_forward_selector_eval,
    llavaov15_vlmodel_forward_selector_eval,
    make_llavaov15_selector_classes,
)

_SELECTOR_CLASSES = (
    "RiceTransformerPretrainedModel_Selector",                    # modeling_selector.py:68
    "LLaVAOneVision1_5_Model_Selector",                           # :188
    "LLaVAOneVision1_5_ForConditionalGeneration_Selector",        # :339
)


def _build_selector_classes():
    try:
        import llavaonevision1_5.modeling_llavaonevision1_5 as ov
    except ImportError as e:
        raise ImportError(
            "compression_method.modeling_selector: the *_Selector classes subclass the LLaVA-OV-1.5 model code; put the "
            "package root that holds `llavaonevision1_5/` (the reference's llava-ov-15/) on sys.path") from e
    built = make_llavaov15_selector_classes(
        ov.RiceTransformerPretrainedModel, ov.LLaVAOneVision1_5_Model, ov.LLaVAOneVision1_5_ForConditionalGeneration,
        text_model_cls=getattr(ov, "LLaVAOneVision1_5_TextModel", None))
    for name, cls in zip(_SELECTOR_CLASSES, built):
        cls.__module__ = __name__
        cls.__qualname__ = name
        globals()[name] = cls


def __getattr__(name):
    if name in _SELECTOR_CLASSES:
        _build_selector_classes()
        return globals()[name]
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__():
    return sorted(set(globals()) | set(_SELECTOR_CLASSES))
