"""Vision tower alone (Qwen2.5-VL-7B ViT geometry, random-init bf16, 96x96 patches) for rocprofv3 / timing."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import Qwen2_5_VLConfig  # noqa: E402
from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf  # noqa: E402

from visionselector_amd.attention import ATTN_NAME_PACKED, replace_qwen2_vl_attention_class  # noqa: E402

replace_qwen2_vl_attention_class()
vc = Qwen2_5_VLConfig(vision_config=dict(depth=32, hidden_size=1280, num_heads=16, intermediate_size=3420, out_hidden_size=3584,
                                         patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=112,
                                         fullatt_block_indexes=[7, 15, 23, 31], in_channels=3)).vision_config
vc._attn_implementation = os.environ.get("TOWER_ATTN", ATTN_NAME_PACKED)
torch.manual_seed(0)
torch.set_default_dtype(torch.bfloat16)
with torch.device("cuda"):
    tower = hf.Qwen2_5_VisionTransformerPretrainedModel(vc).eval()
torch.set_default_dtype(torch.float32)
pix = torch.randn(96 * 96, 3 * 2 * 14 * 14, device="cuda").bfloat16()
grid = torch.tensor([[1, 96, 96]], device="cuda")
with torch.no_grad():
    for _ in range(2):
        tower(pix, grid)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = int(os.environ.get("ITERS", "3"))
    for _ in range(n):
        tower(pix, grid)
    torch.cuda.synchronize()
print({"tower_ms": (time.perf_counter() - t0) / n * 1e3, "attn": vc._attn_implementation})
