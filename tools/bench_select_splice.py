"""LIS + splice per call, before / after the select->splice fusion.

  unfused  vsel_lis_select (kept rows -> [k, D]) + vsel_splice(_batched) (copies them again)
  fused    vsel_lis_select_splice, one-launch form (knob lis_splice_fused = 1) and general form (= 0)
Qwen2.5-VL-7B geometry (D = D_llm = 3584, Hd = 1792), 20 % retain, 64 text tokens per prompt.  Prints one JSON line per case.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N  # noqa: E402
from visionselector_amd import ops  # noqa: E402

IMG = 151655
d, hd = 3584, 1792
g = torch.Generator(device="cuda").manual_seed(0)
wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]


def timed(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(visual_lens, n_text=64, budget=0.2):
    ks = [max(1, int(n * budget)) for n in visual_lens]
    seq_lens = [n + n_text for n in visual_lens]
    ids = torch.cat([torch.cat((torch.arange(10, 10 + n_text // 2), torch.full((n,), IMG), torch.arange(50, 50 + n_text - n_text // 2)))
                     for n in visual_lens]).cuda()
    t = ids.numel()
    h = torch.randn(sum(visual_lens), d, device="cuda", generator=g).bfloat16()
    emb = torch.randn(t, d, device="cuda", generator=g).bfloat16()
    pos = torch.arange(t, device="cuda").repeat(3, 1).contiguous()
    single = len(visual_lens) == 1

    def unfused():
        if single:
            out, idx, sc = ops.lis_select(h, wq, bq, wk, bk, ks[0])
            return ops.splice(ids[None], emb[None], IMG, idx, out, visual_lens[0], position_ids=pos[:, None, :])
        out, idx, sc = ops.lis_select_varlen(h, visual_lens, ks, wq, bq, wk, bk)
        return ops.splice_batched(ids, emb, IMG, seq_lens, visual_lens, ks, idx, out, position_ids=pos)

    def fused():
        return ops.lis_select_splice(h, wq, bq, wk, bk, ids, emb, IMG, seq_lens, visual_lens, ks, position_ids=pos)

    def lis_only():
        return ops.lis_select(h, wq, bq, wk, bk, ks[0]) if single else ops.lis_select_varlen(h, visual_lens, ks, wq, bq, wk, bk)

    res = {"prompts": len(visual_lens), "visual_tokens": sum(visual_lens), "kept": sum(ks), "out_rows": t - sum(visual_lens) + sum(ks)}
    res["lis_select_only_us"] = round(timed(lis_only), 2)
    res["unfused_select_then_splice_us"] = round(timed(unfused), 2)
    with N.debug_knob("lis_splice_fused", 1):
        res["fused_us"] = round(timed(fused), 2)
        N.profile_start()
        fused()
        res["fused_launches"] = {k_: c for k_, (_, c) in N.profile_stop().items()}
        if single and 0 < ks[0] < visual_lens[0]:
            # what the *_Selector prefill enqueues per image (hf_generic.select_and_splice): the fused select-splice AND the soft
            # top-k that fills visual.last_combined_scores (EV/token_compression/selector_model.py:190)
            sc = fused()["scores"]
            res["soft_topk_alone_us"] = round(timed(lambda: ops.soft_topk_fwd(sc[None], ks[0])), 2)
            res["eval_call_two_launches_us"] = round(timed(lambda: ops.soft_topk_fwd(fused()["scores"][None], ks[0])), 2)
            res["eval_call_us"] = round(timed(lambda: ops.lis_select_splice(h, wq, bq, wk, bk, ids, emb, IMG, seq_lens, visual_lens, ks,
                                                                            position_ids=pos, soft=True)), 2)
    with N.debug_knob("lis_splice_fused", 0):
        res["fused_general_form_us"] = round(timed(fused), 2)
    N.profile_start()
    unfused()
    res["unfused_launches"] = {k_: c for k_, (_, c) in N.profile_stop().items()}
    res["per_prompt_us"] = {"unfused": round(res["unfused_select_then_splice_us"] / len(visual_lens), 2),
                            "fused": round(min(res["fused_us"], res["fused_general_form_us"]) / len(visual_lens), 2)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    case([2304])
    case([576])
    case([2304] * 4)
    case([2304] * 8)
    rng = torch.Generator().manual_seed(1)
    case([int(x) for x in torch.randint(576, 4097, (64,), generator=rng)])          # config 5: dynamic resolution batch
