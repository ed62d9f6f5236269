"""Import-path shim for `compression_method.modeling_selector` (reference: llava-ov-15/compression_method/modeling_selector.py).
The three *_Selector classes subclass the OV model code vendored in the reference, so they are built from the caller's base
classes: see make_llavaov15_selector_classes and INTEGRATION.md section 3."""
from visionselector_amd.selector import TopK, _find_ts, topk  # noqa: F401
from visionselector_amd.selector import TransformerScorer  # noqa: F401
from visionselector_amd.hf_llavaov15 import (  # noqa: F401
    llavaov15_vision_tower_forward_selector_eval,
    llavaov15_vlmodel_forward_selector_eval,
    make_llavaov15_selector_classes,
)
