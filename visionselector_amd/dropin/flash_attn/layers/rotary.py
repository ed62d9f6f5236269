"""`from flash_attn.layers.rotary import apply_rotary_emb` (qwen-vl-finetune/compression_method/selector_model.py:31)."""
from visionselector_amd.flash_attn_compat import apply_rotary_emb  # noqa: F401
