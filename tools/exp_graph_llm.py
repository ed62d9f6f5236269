"""hipGraph capture of a Qwen2.5-VL text-model prefill through the vsel_varlen attention (debug / measurement)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import Qwen2_5_VLTextConfig  # noqa: E402
from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf  # noqa: E402

from visionselector_amd.attention import ATTN_NAME, replace_qwen2_vl_attention_class  # noqa: E402

replace_qwen2_vl_attention_class()
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
impl = sys.argv[2] if len(sys.argv) > 2 else ATTN_NAME
cfg = Qwen2_5_VLTextConfig(hidden_size=3584, intermediate_size=18944, num_hidden_layers=layers, num_attention_heads=28,
                           num_key_value_heads=4, vocab_size=1024, max_position_embeddings=32768,
                           rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1000000.0))
cfg._attn_implementation = impl
torch.set_default_dtype(torch.bfloat16)
with torch.device("cuda"):
    model = hf.Qwen2_5_VLTextModel(cfg).eval()
torch.set_default_dtype(torch.float32)
for L in (524, 2368):
    x = torch.randn(1, L, 3584, device="cuda", dtype=torch.bfloat16) * 0.02
    pos = torch.arange(L, device="cuda")[None, None, :].expand(3, 1, L).contiguous()
    with torch.no_grad():
        for _ in range(2):
            eager = model(inputs_embeds=x, position_ids=pos, use_cache=False).last_hidden_state
    torch.cuda.synchronize()
    if os.environ.get("WITH_PROF"):
        from visionselector_amd import _native
        _native.profile_start()
        with torch.no_grad():
            model(inputs_embeds=x, position_ids=pos, use_cache=False)
        torch.cuda.synchronize()
        print(_native.profile_stop(), flush=True)
    print("eager ok", L, flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        for _ in range(5):
            model(inputs_embeds=x, position_ids=pos, use_cache=False)
    e1.record()
    torch.cuda.synchronize()
    t_eager = e0.elapsed_time(e1) / 5
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        model(inputs_embeds=x, position_ids=pos, use_cache=False)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        y = model(inputs_embeds=x, position_ids=pos, use_cache=False).last_hidden_state
    print("captured", L, flush=True)
    graph.replay()
    torch.cuda.synchronize()
    print("replayed", L, bool(torch.equal(y, eager)), flush=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    print({"L": L, "layers": layers, "impl": impl, "eager_ms": t_eager, "graph_ms": e0.elapsed_time(e1) / 5}, flush=True)
