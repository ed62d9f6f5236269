"""`from flash_attn.flash_attn_interface import flash_attn_varlen_func` (qwen-vl-finetune/qwenvl/train/trainer.py:7)."""
from visionselector_amd.flash_attn_compat import flash_attn_func, flash_attn_varlen_func  # noqa: F401
