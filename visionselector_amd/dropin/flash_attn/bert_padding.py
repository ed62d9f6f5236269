"""`from flash_attn.bert_padding import index_first_axis, pad_input, unpad_input`
(qwen-evaluation/token_compression/selector_model.py:26)."""
from visionselector_amd.flash_attn_compat import index_first_axis, pad_input, unpad_input  # noqa: F401
