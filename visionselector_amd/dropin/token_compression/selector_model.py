"""Import-path shim for `token_compression.selector_model` (reference: qwen-evaluation/token_compression/selector_model.py)."""
from visionselector_amd.selector import TopK, _find_ts, topk  # noqa: F401
from visionselector_amd.hf_qwen25vl import (  # noqa: F401
    Qwen2_5_VisionTransformerPretrainedModel_Selector,
    Qwen2_5_VLForConditionalGeneration_Selector,
)
