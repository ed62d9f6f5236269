"""Import-path shim for `compression_method.selector_model`
(reference: qwen-vl-finetune/compression_method/selector_model.py, llava-ov-15/compression_method/selector_model.py)."""
from visionselector_amd.selector import TopK, _find_ts, topk  # noqa: F401
from visionselector_amd.hf_qwen25vl import (  # noqa: F401
    qwen25vl_generation_forward_selector,
    qwen25vl_vision_tower_forward_selector,
)
from visionselector_amd.hf_generic import make_vision_tower_forward_selector  # noqa: F401
from visionselector_amd.hf_llavaov15 import (  # noqa: F401
    install_selector_llavaov15,
    llavaov15_generation_forward_selector,
    llavaov15_vision_tower_forward_selector,
    llavaov15_vlmodel_forward_selector,
)
