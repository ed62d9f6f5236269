"""End-to-end drop-in checks on a tiny random Qwen2.5-VL (installed transformers classes) on the GPU:
the *_Selector inference classes (splice + prefill) and the bound training forwards (soft mask + constraint loss +
gradients) against an eager torch restatement of the reference block run through the same stock model."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

IMG, VID, VSTART, VEND = 10, 11, 12, 13


def tiny_config():
    from transformers import Qwen2_5_VLConfig
    return Qwen2_5_VLConfig(
        text_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                         num_key_value_heads=2, vocab_size=64, max_position_embeddings=4096,
                         rope_parameters=dict(rope_type="default", mrope_section=[4, 6, 6], rope_theta=10000.0)),
        vision_config=dict(depth=2, hidden_size=64, num_heads=4, intermediate_size=128, out_hidden_size=128, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1],
                           in_channels=3),
        image_token_id=IMG, video_token_id=VID, vision_start_token_id=VSTART, vision_end_token_id=VEND)


def make_inputs(grid=(1, 16, 16), n_pre=5, n_post=9, seed=0):
    g = torch.Generator().manual_seed(seed)
    n_patches = grid[0] * grid[1] * grid[2]
    n_vis = n_patches // 4
    pix = torch.randn(n_patches, 3 * 2 * 14 * 14, generator=g)
    pre = torch.randint(20, 60, (n_pre,), generator=g)
    post = torch.randint(20, 60, (n_post,), generator=g)
    ids = torch.cat((pre, torch.tensor([VSTART]), torch.full((n_vis,), IMG), torch.tensor([VEND]), post))[None]
    mm = (ids == IMG).int()
    return dict(input_ids=ids.cuda(), attention_mask=torch.ones_like(ids).cuda(), pixel_values=pix.cuda(),
                image_grid_thw=torch.tensor([list(grid)]).cuda(), mm_token_type_ids=mm.cuda()), n_vis


def randomize_scorer(scorer, seed=1):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in scorer.parameters():
            p.copy_((0.05 * torch.randn(p.shape, generator=g)).to(p.device))


def eager_scores(h, scorer):
    """reference formulation in torch (selector_scorer.py:47-53), the checker for this file"""
    k = F.linear(h, scorer.k_proj.weight, scorer.k_proj.bias)
    q = F.linear(h, scorer.q_proj.weight, scorer.q_proj.bias)
    return (q @ k.T / scorer.hidden_dim ** 0.5).mean(-1)


@pytest.fixture(scope="module")
def selector_model():
    assert torch.cuda.is_available()
    from visionselector_amd.hf_qwen25vl import Qwen2_5_VLForConditionalGeneration_Selector
    torch.manual_seed(0)
    m = Qwen2_5_VLForConditionalGeneration_Selector(tiny_config()).cuda().float().eval()
    randomize_scorer(m.visual.importance_scorer)
    return m


def test_selector_state_dict_keys(selector_model):
    keys = [k for k in selector_model.state_dict() if "importance_scorer" in k]
    assert sorted(k.split("importance_scorer.")[1] for k in keys) == ["k_proj.bias", "k_proj.weight", "q_proj.bias", "q_proj.weight"]
    assert all(k.startswith("model.visual.importance_scorer.") for k in keys)


def test_selector_soft_scores_in_the_reference_bf16_arithmetic(selector_model):
    """visual.soft_topk_bf16_reference = True: last_combined_scores comes from vsel_soft_topk_fwd_bf16ref -- the reference's _find_ts with
    every step rounded to bf16 (its bf16 scorers' stalled bisection, sum != k) -- instead of the fp32 root; selection, logits and
    the default behaviour are untouched."""
    from visionselector_amd import ops
    m = selector_model
    m.visual.budgets = 0.25
    inp, n_vis = make_inputs()
    k = max(1, int(n_vis * 0.25))
    outs = {}
    try:
        for flag in (False, True):
            m.visual.soft_topk_bf16_reference = flag
            m.model.rope_deltas = None
            with torch.no_grad():
                o = m(**inp)
            outs[flag] = (o.logits.clone(), m.visual.last_selected_indices.clone(), m.visual.last_combined_scores.float().clone())
    finally:
        m.visual.soft_topk_bf16_reference = False
    assert torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1])
    assert abs(float(outs[False][2].sum()) - k) < 1e-2
    ps = outs[True][2]
    assert torch.equal(ps, ps.bfloat16().float())                                     # bf16 values
    assert float((ps - outs[False][2]).abs().max()) <= 2e-2 and abs(float(ps.sum()) - k) <= 0.02 * k + 1.0


@pytest.mark.parametrize("budget", [0.25, 0.5])
def test_selector_prefill_matches_manual_splice(selector_model, budget):
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    m = selector_model
    m.visual.budgets = budget
    m.model.rope_deltas = None
    inp, n_vis = make_inputs()
    with torch.no_grad():
        out = m(**inp)
        k = max(1, int(n_vis * budget))
        idx = m.visual.last_selected_indices
        assert idx.shape == (k,) and bool((idx[1:] > idx[:-1]).all())
        assert abs(float(m.visual.last_combined_scores.sum()) - k) < 1e-2
        L = inp["input_ids"].shape[1]
        assert out.logits.shape[1] == L - n_vis + k
        # manual path through the STOCK classes: merged tokens -> eager scores -> topk/sort -> spliced embeds/positions
        merged = hf.Qwen2_5_VisionTransformerPretrainedModel.forward(m.visual, inp["pixel_values"], inp["image_grid_thw"]).pooler_output
        s = eager_scores(merged, m.visual.importance_scorer)
        ref_idx = s.topk(k).indices.sort().values
        assert torch.equal(ref_idx, idx), "kept indices = eager reference formulation"
        ids = inp["input_ids"]
        img_pos = torch.where(ids == IMG)[1]
        sel = torch.cat((img_pos[ref_idx], torch.where(ids != IMG)[1])).sort().values
        emb = m.get_input_embeddings()(ids)
        emb[0, img_pos] = merged
        emb = emb[:, sel]
        m.model.rope_deltas = None
        pos, _ = m.model.get_rope_index(ids, image_grid_thw=inp["image_grid_thw"], attention_mask=inp["attention_mask"],
                                        mm_token_type_ids=inp["mm_token_type_ids"])
        ref = hf.Qwen2_5_VLForConditionalGeneration.forward(
            m, inputs_embeds=emb, position_ids=pos[:, :, sel], attention_mask=inp["attention_mask"][:, sel])
        assert float((out.logits - ref.logits).abs().max()) <= 1e-4 * max(1.0, float(ref.logits.abs().max()))


def test_selector_generate_runs_on_compressed_cache(selector_model):
    m = selector_model
    m.visual.budgets = 0.25
    m.model.rope_deltas = None
    inp, n_vis = make_inputs(seed=3)
    with torch.no_grad():
        first = m(**inp).logits[0, -1].argmax()
        m.model.rope_deltas = None
        gen = m.generate(**inp, max_new_tokens=4, do_sample=False)
    L = inp["input_ids"].shape[1]
    assert gen.shape[1] == L + 4
    assert int(gen[0, L]) == int(first)


def test_training_forward_backward_matches_eager_block():
    """install_selector on a stock model: loss = CE + w * BCE and the scorer gradients equal autograd through an eager
    torch restatement of the block (reference formulation + 64-step bisection + TopK.backward closed form)."""
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    from visionselector_amd.hf_qwen25vl import install_selector
    torch.manual_seed(0)
    model = hf.Qwen2_5_VLForConditionalGeneration(tiny_config()).cuda().float().train()
    install_selector(model, budget=0.25, regularization_weight=0.7)
    visual = model.model.visual
    randomize_scorer(visual.importance_scorer, seed=2)
    for n, p in model.named_parameters():
        p.requires_grad = "importance_scorer" in n                 # set_model(): only the compressor is tuned
    inp, n_vis = make_inputs(seed=5)
    labels = inp["input_ids"].clone()
    labels[inp["input_ids"] == IMG] = -100
    out = model(**inp, labels=labels)
    out.loss.backward()
    got = {n: p.grad.clone() for n, p in visual.importance_scorer.named_parameters()}

    # ---- eager checker ---------------------------------------------------------------------------------------
    class TopKRef(torch.autograd.Function):
        @staticmethod
        def forward(ctx, xs, k):
            lo = -xs.max(dim=1, keepdims=True).values - 10
            hi = -xs.min(dim=1, keepdims=True).values + 10
            for _ in range(64):
                mid = (hi + lo) / 2
                mask = torch.sigmoid(xs + mid).sum(dim=1) < k
                lo[mask] = mid[mask]
                hi[~mask] = mid[~mask]
            ts = (lo + hi) / 2
            ctx.save_for_backward(xs, ts)
            return torch.sigmoid(xs + ts)

        @staticmethod
        def backward(ctx, g):
            xs, ts = ctx.saved_tensors
            p = torch.sigmoid(xs + ts)
            v = p * (1 - p)
            uv = g * v
            return uv - uv.sum(dim=1, keepdims=True) * v / v.sum(dim=1, keepdims=True), None

    for p in visual.importance_scorer.parameters():
        p.grad = None
    merged = hf.Qwen2_5_VisionTransformerPretrainedModel.forward(visual, inp["pixel_values"], inp["image_grid_thw"]).pooler_output.detach()
    s = eager_scores(merged, visual.importance_scorer)
    k = int(n_vis * 0.25)
    ps = TopKRef.apply(s[None], k)[0]
    y = torch.zeros_like(s).scatter_(0, s.topk(k).indices, 1.0)
    emb = model.get_input_embeddings()(inp["input_ids"]).detach()
    emb = emb.masked_scatter((inp["input_ids"] == IMG)[..., None].expand_as(emb), ps[:, None] * merged)
    ref = hf.Qwen2_5_VLForConditionalGeneration.forward(model, input_ids=inp["input_ids"], inputs_embeds=emb,
                                                        attention_mask=inp["attention_mask"], labels=labels,
                                                        image_grid_thw=inp["image_grid_thw"],
                                                        mm_token_type_ids=inp["mm_token_type_ids"])
    ref_loss = ref.loss + 0.7 * F.binary_cross_entropy(ps, y)
    ref_loss.backward()
    assert abs(float(out.loss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    for n, p in visual.importance_scorer.named_parameters():
        scale = max(float(p.grad.abs().max()), 1e-8)
        # TOLERANCE 2e-3 of the tensor's max: fp32 eager autograd through the N x N matmul vs the closed form
        assert float((got[n] - p.grad).abs().max()) <= 2e-3 * scale + 1e-7, n
    with pytest.raises(ValueError, match="do not match"):        # reference: ValueError on token-count mismatch
        bad = dict(inp)
        bad["input_ids"] = inp["input_ids"].clone()
        bad["input_ids"][0, 0] = IMG
        model(**bad)


def test_attention_interface_registration():
    from transformers import AttentionInterface
    from visionselector_amd.attention import ATTN_NAME, replace_qwen2_vl_attention_class, vsel_attention_forward
    assert replace_qwen2_vl_attention_class() == ATTN_NAME
    assert AttentionInterface()[ATTN_NAME] is vsel_attention_forward
    # prefill call in the interface layout: [B, H, L, d]
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(2, 8, 200, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(2, 2, 200, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(2, 2, 200, 128, device="cuda", generator=g).bfloat16()
    with torch.no_grad():
        out, _ = vsel_attention_forward(None, q, k, v, None, scaling=128 ** -0.5, is_causal=True)
        ref = F.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(4, 1), v.float().repeat_interleave(4, 1),
                                             is_causal=True).transpose(1, 2)
    assert out.shape == (2, 200, 8, 128)
    assert float((out.float() - ref).abs().max()) <= 2e-2 and float((out.float() - ref).abs().mean()) <= 1e-3
    # training: gradients flow through the native backward (the reference trains through flash_attn's), vs SDPA autograd
    qg, kg, vg = [t.clone().requires_grad_(True) for t in (q, k, v)]
    out_g, _ = vsel_attention_forward(None, qg, kg, vg, None, scaling=128 ** -0.5, is_causal=True)
    assert torch.equal(out_g, out)
    go = torch.randn(out.shape, device="cuda", generator=g).bfloat16()
    out_g.backward(go)
    qr, kr, vr = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
    F.scaled_dot_product_attention(qr, kr.repeat_interleave(4, 1), vr.repeat_interleave(4, 1),
                                   is_causal=True).transpose(1, 2).backward(go.float())
    for got, want in ((qg.grad, qr.grad), (kg.grad, kr.grad), (vg.grad, vr.grad)):
        # TOLERANCE 2^-6 of the tensor's max magnitude: bf16 gradients + bf16-rounded P / dS operands
        assert float((got.float() - want).abs().max()) <= 2 ** -6 * float(want.abs().max())


def test_native_lis_trainer_reduces_loss_and_checkpoints(tmp_path):
    """LIS-only training loop (no HF Trainer / DeepSpeed): curriculum weight, AdamW, clip, checkpoint with the
    reference's key names, resume."""
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    from visionselector_amd.hf_qwen25vl import install_selector
    from visionselector_amd.trainer import LisTrainer, load_scorer_state_dict, scorer_state_dict
    torch.manual_seed(0)
    model = hf.Qwen2_5_VLForConditionalGeneration(tiny_config()).cuda().float().train()
    install_selector(model, budget=0.25)
    randomize_scorer(model.model.visual.importance_scorer, seed=4)
    logs = []
    tr = LisTrainer(model, max_steps=6, lr=1e-2, reg_weight_start=0.1, reg_weight_end=2.0, log=logs.append)
    assert all(("importance_scorer" in n) == p.requires_grad for n, p in model.named_parameters())
    batches = []
    for seed in (5, 6):
        inp, _ = make_inputs(seed=seed)
        labels = inp["input_ids"].clone()
        labels[inp["input_ids"] == IMG] = -100
        batches.append(dict(inp, labels=labels))
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses = [tr.train_step(batches) for _ in range(6)]
    assert model.regularization_weight == pytest.approx(0.1 + 1.9 * 5 / 6)       # weight used by the last step
    assert any("Set regularization_weight to: 0.4167" in s for s in logs)         # step 1: 0.1 + 1.9/6
    assert all(np.isfinite(losses))
    for n, p in model.named_parameters():
        changed = not torch.equal(before[n], p.detach())
        assert changed == ("importance_scorer" in n), n                             # only the scorer moved
    sd = scorer_state_dict(model)
    assert sorted(sd) == ["visual.importance_scorer.k_proj.bias", "visual.importance_scorer.k_proj.weight",
                          "visual.importance_scorer.q_proj.bias", "visual.importance_scorer.q_proj.weight"]
    path = str(tmp_path / "ck" / "lis.pt")
    tr.save(path)
    model2 = hf.Qwen2_5_VLForConditionalGeneration(tiny_config()).cuda().float()
    install_selector(model2, budget=0.25)
    tr2 = LisTrainer(model2, max_steps=6, lr=1e-2)
    tr2.resume(path)
    assert tr2.global_step == 6
    for a, b in zip(model.model.visual.importance_scorer.parameters(), model2.model.visual.importance_scorer.parameters()):
        assert torch.equal(a, b)
    with pytest.raises(KeyError):
        load_scorer_state_dict(model2, {"visual.importance_scorer.q_proj.weight": sd["visual.importance_scorer.q_proj.weight"]})


def test_generic_tower_wrapper_llavaov_style():
    """make_vision_tower_forward_selector on a stand-in tower (LLaVA-OV's Rice ViT is vendored reference code): scorer
    4096 -> 2048 as modeling_selector.py:101, joint scoring of 8 frames x 729 tokens, 1-D position splice (pos_rows = 1)."""
    from visionselector_amd import ops
    from visionselector_amd.hf_generic import make_vision_tower_forward_selector
    from visionselector_amd.selector import TransformerScorer

    class Tower(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.importance_scorer = TransformerScorer(4096, 2048)
            self.budgets = 0.2

        def base_forward(self, hidden_states, grid_thw):
            return hidden_states            # stand-in for patch embed + blocks + merger

    tower = Tower().cuda().bfloat16()
    randomize_scorer(tower.importance_scorer, seed=8)
    fwd_eval = make_vision_tower_forward_selector(Tower.base_forward, "eval")
    fwd_train = make_vision_tower_forward_selector(Tower.base_forward, "train")
    g = torch.Generator(device="cuda").manual_seed(0)
    n = 8 * 729
    h = torch.randn(n, 4096, device="cuda", generator=g).bfloat16()
    out, idx, total = fwd_eval(tower, h, None)
    k = max(1, int(n * 0.2))
    assert total == n and out.shape == (k, 4096) and torch.equal(out, h[idx])
    s = eager_scores(h.float(), tower.importance_scorer.float())
    tower.bfloat16()
    assert len(set(idx.tolist()) ^ set(s.topk(k).indices.tolist())) <= 4      # bf16 weights cast: boundary ties only
    h_new, ps, y = fwd_train(tower, h, None)
    assert h_new.shape == h.shape and abs(float(ps.sum()) - int(n * 0.2)) < 0.05 and int(y.sum()) == int(n * 0.2)
    # 1-D positions (OV :311-314): ids / embeds / cache_position sliced by the same kernel
    ids = torch.cat((torch.arange(20, 30), torch.full((n,), IMG), torch.arange(30, 36)))[None].cuda()
    emb = torch.randn(1, ids.shape[1], 64, device="cuda", generator=g).bfloat16()
    pos1d = torch.arange(ids.shape[1], device="cuda")[None, None, :]
    sel, new_ids, new_emb, new_pos, _ = ops.splice(ids, emb, IMG, idx, out[:, :64].contiguous(), n, position_ids=pos1d, check=True)
    assert new_pos.shape == (1, 1, ids.shape[1] - n + k) and torch.equal(new_pos[0, 0], sel)


def test_select_splice_fusion_is_transparent(selector_model):
    """The *_Selector prefill writes the kept rows once, from the merger's output into inputs_embeds' (vsel_lis_select_splice):
    same logits / indices / soft scores, bit for bit, as visual() followed by vsel_splice, with and without the un-reorder
    fusion; the unfused launches are gone from the fused forward."""
    from visionselector_amd import _native as N
    m = selector_model
    m.visual.budgets = 0.25
    inp, n_vis = make_inputs(grid=(1, 32, 32), seed=10)
    res = {}
    for fuse_unreorder in (True, False):
        m.visual.fuse_unreorder = fuse_unreorder
        for fuse in (True, False):
            m.fuse_select_splice = fuse
            m.model.rope_deltas = None
            N.profile_start()
            with torch.no_grad():
                o = m(**inp)
            prof = N.profile_stop()
            assert (("select_splice_small_kernel" in prof) == fuse) and (("splice_embed_kernel" in prof) == (not fuse)), prof
            res[(fuse_unreorder, fuse)] = (o.logits.clone(), m.visual.last_selected_indices.clone(),
                                           m.visual.last_combined_scores.clone(), m.model.rope_deltas.clone())
        for a, b in zip(res[(fuse_unreorder, True)], res[(fuse_unreorder, False)]):
            assert torch.equal(a, b)
    m.visual.fuse_unreorder = True
    m.fuse_select_splice = True


def test_fused_prefill_honours_tower_hooks_and_instance_forward(selector_model):
    """The fused prefill bypasses `nn.Module.__call__` on the tower, so it is taken only when that call would do nothing
    else: with a forward / pre-forward hook or an instance-level `forward` (accelerate's device_map hook, MethodType
    patches) the prefill goes through `self.visual(...)`, the hook fires, and the logits are the fused path's."""
    import types
    from visionselector_amd import _native as N
    m = selector_model
    m.visual.budgets = 0.25
    inp, _ = make_inputs(grid=(1, 32, 32), seed=12)
    with torch.no_grad():
        ref = m(**inp).logits.clone()
    calls = []

    def run(expect_fused):
        m.model.rope_deltas = None
        N.profile_start()
        with torch.no_grad():
            o = m(**inp)
        prof = N.profile_stop()
        assert ("select_splice_small_kernel" in prof) == expect_fused, prof
        assert torch.equal(o.logits, ref)

    run(True)
    h = m.visual.register_forward_hook(lambda mod, args, out: calls.append("post"))
    run(False)
    h.remove()
    h = m.visual.register_forward_pre_hook(lambda mod, args: calls.append("pre"))
    run(False)
    h.remove()
    assert calls == ["post", "pre"]
    cls_forward = type(m.visual).forward

    def patched(self, *a, **k):
        calls.append("instance")
        return cls_forward(self, *a, **k)

    m.visual.forward = types.MethodType(patched, m.visual)
    try:
        run(False)
    finally:
        del m.visual.forward
    assert calls[-1] == "instance"
    run(True)


def test_unreorder_fusion_is_transparent(selector_model):
    """The inference tower skips transformers' `merged[reverse_indices, :]` gather (vsel_lis_select_permuted): same kept
    tokens / indices / logits as with the gather executed."""
    m = selector_model
    m.visual.budgets = 0.25
    inp, n_vis = make_inputs(grid=(1, 32, 32), seed=9)        # 256 merged tokens, several 4x4-token windows
    outs = []
    for fuse in (True, False):
        m.visual.fuse_unreorder = fuse
        m.model.rope_deltas = None
        with torch.no_grad():
            o = m(**inp)
        outs.append((o.logits.clone(), m.visual.last_selected_indices.clone(), m.visual.last_combined_scores.clone()))
    m.visual.fuse_unreorder = True
    assert torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 1e-5 * max(1.0, float(outs[1][0].abs().max()))
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= 1e-5
    # the fused path really took the permutation from the tower (window order != natural order for this grid)
    from visionselector_amd import hf_generic
    calls = []
    orig = hf_generic._select_block_permuted
    hf_generic._select_block_permuted = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    m.fuse_select_splice = False          # the tower's own forward (visual() -> (tokens, indices, N)), as the reference calls it
    try:
        with torch.no_grad():
            m.model.rope_deltas = None
            m(**inp)
    finally:
        hf_generic._select_block_permuted = orig
        m.fuse_select_splice = True
    assert calls, "permuted path not taken"
    # and the select->splice form hands the same row maps to vsel_lis_select_splice
    seen = []
    orig_ss = hf_generic.ops.lis_select_splice
    hf_generic.ops.lis_select_splice = lambda *a, **k: (seen.append(k.get("logical_to_physical")), orig_ss(*a, **k))[1]
    try:
        with torch.no_grad():
            m.model.rope_deltas = None
            m(**inp)
    finally:
        hf_generic.ops.lis_select_splice = orig_ss
    assert seen and seen[0] is not None and not torch.equal(seen[0].cpu(), torch.arange(n_vis))


def test_training_through_native_attention_matches_sdpa():
    """The reference trains the LIS through the frozen LLM whose attention is flash_attn_varlen_func (trainer.py:101-113).
    Here: the LLM's attention is vsel_varlen (native forward + backward); the scorer gradients must match the same model
    with torch SDPA attention.  bf16 model, head_dim 128."""
    from transformers import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    from visionselector_amd.attention import ATTN_NAME, replace_qwen2_vl_attention_class
    from visionselector_amd.hf_qwen25vl import install_selector
    replace_qwen2_vl_attention_class()
    cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                         num_key_value_heads=1, vocab_size=64, max_position_embeddings=4096,
                         rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=10000.0)),
        vision_config=dict(depth=2, hidden_size=64, num_heads=4, intermediate_size=128, out_hidden_size=256, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1],
                           in_channels=3),
        image_token_id=IMG, video_token_id=VID, vision_start_token_id=VSTART, vision_end_token_id=VEND)
    torch.manual_seed(0)
    model = hf.Qwen2_5_VLForConditionalGeneration(cfg).cuda().bfloat16().train()
    install_selector(model, budget=0.25, regularization_weight=0.7)
    visual = model.model.visual
    randomize_scorer(visual.importance_scorer, seed=2)
    for n, p in model.named_parameters():
        p.requires_grad = "importance_scorer" in n
    inp, n_vis = make_inputs(grid=(1, 32, 32), seed=5)
    inp["pixel_values"] = inp["pixel_values"].bfloat16()
    labels = inp["input_ids"].clone()
    labels[inp["input_ids"] == IMG] = -100

    def run(impl):
        model.model.language_model.config._attn_implementation = impl      # LLM only; the vision tower (head_dim 16) stays SDPA
        model.model.visual.config._attn_implementation = "sdpa"
        for p in visual.importance_scorer.parameters():
            p.grad = None
        out = model(**inp, labels=labels)
        out.loss.backward()
        return float(out.loss.detach()), {n: p.grad.float().clone() for n, p in visual.importance_scorer.named_parameters()}

    loss_ref, g_ref = run("sdpa")
    from visionselector_amd import _native as N
    N.profile_start()
    loss_got, g_got = run(ATTN_NAME)
    prof = N.profile_stop()
    assert prof["varlen_attn_fwd_kernel"][1] == 2 and prof["attn_bwd_dq_kernel"][1] == 2, prof       # 2 layers, native path ran
    assert abs(loss_got - loss_ref) <= 2e-2 * max(1.0, abs(loss_ref))
    for n in g_ref:
        scale = float(g_ref[n].abs().max())
        # TOLERANCE 6 % of the tensor's max: two bf16 attention implementations (different summation order and rounding of
        # P) back-propagated through 2 bf16 transformer layers
        assert float((g_got[n] - g_ref[n]).abs().max()) <= 6e-2 * scale + 1e-6, (n, float((g_got[n] - g_ref[n]).abs().max()), scale)


def test_merger_colsum_fusion_is_transparent(selector_model):
    """Single-sweep LIS (section 8f N2): the merger's GELU is replaced by vsel_gelu_colsum during the tower forward and the LIS
    block takes sum_rows(H) = sum_rows(G) W2^T + N b2 instead of sweeping H.  Same kept tokens and logits as the two-sweep
    path; the native profile shows the fused kernel and no column-sum sweep."""
    from visionselector_amd import _native as N
    m = selector_model
    m.visual.budgets = 0.25
    inp, n_vis = make_inputs(grid=(1, 32, 32), seed=11)
    outs = []
    for fuse in (True, False):
        m.visual.fuse_merger_colsum = fuse
        m.model.rope_deltas = None
        N.profile_start()
        with torch.no_grad():
            o = m(**inp)
        prof = N.profile_stop()
        outs.append((o.logits.clone(), m.visual.last_selected_indices.clone(), m.visual.last_combined_scores.clone(), prof))
    m.visual.fuse_merger_colsum = None            # back to the default: off (opt-in: True or "auto")
    assert "gelu_colsum_kernel" in outs[0][3] and "colsum_partial_kernel" not in outs[0][3], outs[0][3].keys()
    assert "gelu_colsum_kernel" not in outs[1][3] and "colsum_partial_kernel" in outs[1][3]
    assert torch.equal(outs[0][1], outs[1][1])
    # TOLERANCE: the mean of H comes from fp32 sums of G through an fp32 GEMM instead of fp32 sums of the rounded H rows
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= 1e-4
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 1e-4 * max(1.0, float(outs[1][0].abs().max()))
    assert isinstance(m.visual.merger.mlp[1], torch.nn.GELU)            # the stock module is back in place


# ---------------------------------------------------------------------------------------------------
# LLaVA-OneVision-1.5 surface (duck-typed: the OV model code is vendored in the reference, not in transformers)
# ---------------------------------------------------------------------------------------------------
class _ToyRice(torch.nn.Module):
    """Rice-ViT-shaped tower: forward(hidden_states [n_patches, C], grid_thw) -> merged tokens [n_patches / 4, D]."""

    def __init__(self, c=24, d=128):
        super().__init__()
        self.proj = torch.nn.Linear(c, 32)
        self.merger = torch.nn.Sequential(torch.nn.Linear(4 * 32, 64), torch.nn.GELU(), torch.nn.Linear(64, d))
        self.config = type("C", (), {"out_hidden_size": d})()

    @property
    def dtype(self):
        return self.proj.weight.dtype

    def forward(self, hidden_states, grid_thw, is_verifying=False):
        x = torch.tanh(self.proj(hidden_states))
        if is_verifying:
            return x
        return self.merger(x.view(-1, 4 * 32))


class _ToyOVModel(torch.nn.Module):
    def __init__(self, tower, d=128, vocab=64):
        super().__init__()
        from transformers import Qwen2Config, Qwen2Model
        self.visual = tower
        self.language_model = Qwen2Model(Qwen2Config(hidden_size=d, intermediate_size=256, num_hidden_layers=2,
                                                     num_attention_heads=4, num_key_value_heads=2, vocab_size=vocab,
                                                     max_position_embeddings=2048))
        self.config = type("C", (), {"image_token_id": 60, "video_token_id": 61, "output_attentions": False,
                                     "output_hidden_states": False, "use_return_dict": True, "vocab_size": vocab})()
        self.rope_deltas = None

    def get_input_embeddings(self):
        return self.language_model.embed_tokens

    def get_image_features(self, pixel_values, image_grid_thw=None):
        return self.visual(pixel_values.type(self.visual.dtype), grid_thw=image_grid_thw)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                pixel_values=None, image_grid_thw=None, **kw):
        """The UNPATCHED vl-model forward (what the checker calls): scatter all image features, run the LLM."""
        emb = self.get_input_embeddings()(input_ids) if inputs_embeds is None else inputs_embeds
        if pixel_values is not None:
            feats = self.get_image_features(pixel_values, image_grid_thw)
            emb = emb.masked_scatter((input_ids == self.config.image_token_id)[..., None].expand_as(emb), feats.to(emb.dtype))
        return self.language_model(inputs_embeds=emb, attention_mask=attention_mask, position_ids=position_ids,
                                   past_key_values=past_key_values, return_dict=True)


class _ToyOVForCG(torch.nn.Module):
    def __init__(self, vl, d=128, vocab=64):
        super().__init__()
        from transformers.loss.loss_utils import ForCausalLMLoss
        self.model = vl
        self.lm_head = torch.nn.Linear(d, vocab, bias=False)
        self.config = vl.config
        self.loss_function = ForCausalLMLoss
        self.regularization_weight = 0.0


def _ov_inputs(n_vis=64, n_pre=4, n_post=7, seed=3):
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn(4 * n_vis, 24, generator=g)
    ids = torch.cat((torch.randint(5, 50, (n_pre,), generator=g), torch.full((n_vis,), 60),
                     torch.randint(5, 50, (n_post,), generator=g)))[None]
    return dict(input_ids=ids.cuda(), attention_mask=torch.ones_like(ids).cuda(), pixel_values=pix.cuda(),
                image_grid_thw=torch.tensor([[1, 16, 16]]).cuda())


def test_llavaov15_training_surface_matches_eager_block():
    """llavaov15_{vision_tower,vlmodel,generation}_forward_selector bound with types.MethodType as
    train_sft_visionselector.py:219-225 does: loss = CE + w * BCE and scorer gradients equal autograd through the eager
    restatement of the block."""
    from visionselector_amd.hf_llavaov15 import install_selector_llavaov15
    torch.manual_seed(0)
    model = _ToyOVForCG(_ToyOVModel(_ToyRice())).cuda().float().train()
    install_selector_llavaov15(model, budget=0.25, regularization_weight=0.9)
    visual = model.model.visual
    randomize_scorer(visual.importance_scorer, seed=6)
    for n, p in model.named_parameters():
        p.requires_grad = "importance_scorer" in n
    inp = _ov_inputs()
    labels = inp["input_ids"].clone()
    labels[inp["input_ids"] == 60] = -100
    out = model(**inp, labels=labels)
    out.loss.backward()
    got = {n: p.grad.clone() for n, p in visual.importance_scorer.named_parameters()}
    h_new, img_mask, cmask = visual(inp["pixel_values"], inp["image_grid_thw"])
    assert h_new.shape == (64, 128) and img_mask.shape == (64,) and float(cmask.sum()) == 16.0
    assert visual(inp["pixel_values"], inp["image_grid_thw"], is_verifying=True).shape == (256, 32)

    # ---- eager checker (reference formulation; selector_model.py:120-141, :365-367) ----
    for p in visual.importance_scorer.parameters():
        p.grad = None
    merged = _ToyRice.forward(visual, inp["pixel_values"], inp["image_grid_thw"]).detach()
    s = eager_scores(merged, visual.importance_scorer)
    k = int(64 * 0.25)
    lo = -s.max() - 10
    hi = -s.min() + 10
    with torch.no_grad():
        for _ in range(64):
            mid = (hi + lo) / 2
            if torch.sigmoid(s + mid).sum() < k:
                lo = mid
            else:
                hi = mid
        ts = (lo + hi) / 2
    # implicit differentiation of sum(sigmoid(s + t)) = k  (TopK.backward closed form, selector_model.py:26-36)
    p = torch.sigmoid(s + ts)
    v = (p * (1 - p)).detach()
    ps = p.detach() + (s - s.detach()) * v - ((s - s.detach()) * v).sum() * v / v.sum()
    y = torch.zeros_like(s).scatter_(0, s.topk(k).indices, 1.0).detach()
    emb = model.model.get_input_embeddings()(inp["input_ids"]).detach()
    emb = emb.masked_scatter((inp["input_ids"] == 60)[..., None].expand_as(emb), ps[:, None] * merged)
    hs = model.model.language_model(inputs_embeds=emb, attention_mask=inp["attention_mask"], return_dict=True).last_hidden_state
    logits = model.lm_head(hs)
    ref_loss = model.loss_function(logits=logits, labels=labels, vocab_size=64) + 0.9 * F.binary_cross_entropy(ps, y)
    ref_loss.backward()
    assert abs(float(out.loss.detach()) - float(ref_loss.detach())) <= 1e-4 * max(1.0, abs(float(ref_loss.detach())))
    for n, prm in visual.importance_scorer.named_parameters():
        scale = max(float(prm.grad.abs().max()), 1e-8)
        # TOLERANCE 2e-3 of the tensor's max: fp32 eager autograd through the N x N matmul vs the closed form
        assert float((got[n] - prm.grad).abs().max()) <= 2e-3 * scale + 1e-7, n
    bad = dict(inp)
    bad["input_ids"] = inp["input_ids"].clone()
    bad["input_ids"][0, 0] = 60
    with pytest.raises(ValueError, match="do not match"):
        model(**bad)


def test_llavaov15_inference_classes_splice_like_the_reference():
    """make_llavaov15_selector_classes on stand-in bases: the prefill keeps k image tokens, ids / embeds / 1-D positions /
    mask / cache_position are spliced (modeling_selector.py:245-314), and the LLM output equals a manual torch splice."""
    from visionselector_amd.hf_llavaov15 import (llavaov15_vision_tower_forward_selector_eval,
                                                 llavaov15_vlmodel_forward_selector_eval)
    from visionselector_amd.selector import TransformerScorer
    import types
    torch.manual_seed(1)
    vl = _ToyOVModel(_ToyRice()).cuda().float().eval()
    vl.visual.importance_scorer = TransformerScorer(128, 64).cuda()
    randomize_scorer(vl.visual.importance_scorer, seed=8)
    vl.visual.budgets = 0.25
    vl.visual.forward = types.MethodType(llavaov15_vision_tower_forward_selector_eval, vl.visual)
    inp = _ov_inputs(seed=4)
    with torch.no_grad():
        tokens, idx, total = vl.visual(inp["pixel_values"], inp["image_grid_thw"])
        assert tokens.shape == (16, 128) and total == 64 and torch.equal(idx, idx.sort().values)
        assert abs(float(vl.visual.last_combined_scores.sum()) - 16.0) <= 1e-3
        out, n_vis = llavaov15_vlmodel_forward_selector_eval(vl, **inp)
        # manual splice with torch ops
        ids = inp["input_ids"]
        pos_img = torch.where(ids == 60)[1]
        sel = torch.cat((pos_img[idx], torch.where(ids != 60)[1])).sort().values
        emb = vl.get_input_embeddings()(ids)[:, sel, :]
        new_ids = ids[:, sel]
        emb = emb.masked_scatter((new_ids == 60)[..., None].expand_as(emb), tokens)
        ref = vl.language_model(inputs_embeds=emb, attention_mask=inp["attention_mask"][:, sel], position_ids=sel[None],
                                return_dict=True).last_hidden_state
    assert n_vis == 64 and out.last_hidden_state.shape == (1, 16 + 11, 128)
    assert float((out.last_hidden_state - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


def test_llavaov15_class_factory_builds_the_three_selector_classes():
    """make_llavaov15_selector_classes on PreTrainedModel-shaped toy bases (config-constructed, _from_config, post_init):
    names, scorer attachment (in_features from the tower config), forwards and the (output, visual_token_num) contract."""
    from visionselector_amd.hf_llavaov15 import make_llavaov15_selector_classes

    class Base(torch.nn.Module):
        @classmethod
        def _from_config(cls, config):
            return cls(config)

        def post_init(self):
            pass

    class Rice(Base, _ToyRice):
        def __init__(self, config):
            _ToyRice.__init__(self)
            self.config = config

    class VL(Base):
        def __init__(self, config):
            torch.nn.Module.__init__(self)
            self.config = config
            self.visual = Rice(config.vision_config)
            inner = _ToyOVModel(self.visual)
            self.language_model = inner.language_model
            self.get_input_embeddings = inner.get_input_embeddings

        def get_image_features(self, pixel_values, image_grid_thw=None):
            return self.visual(pixel_values.type(self.visual.dtype), grid_thw=image_grid_thw)

    class CG(Base):
        def __init__(self, config):
            torch.nn.Module.__init__(self)
            self.config = config

    ns = lambda **k: type("C", (), k)()  # noqa: E731
    cfg = ns(vision_config=ns(out_hidden_size=128), text_config=ns(hidden_size=128, vocab_size=64), image_token_id=60,
             video_token_id=61, output_attentions=False, output_hidden_states=False, use_return_dict=True, vocab_size=64)
    RiceSel, VLSel, CGSel = make_llavaov15_selector_classes(Rice, VL, CG)
    assert (RiceSel.__name__, VLSel.__name__, CGSel.__name__) == (
        "RiceTransformerPretrainedModel_Selector", "LLaVAOneVision1_5_Model_Selector",
        "LLaVAOneVision1_5_ForConditionalGeneration_Selector")
    torch.manual_seed(2)
    m = CGSel(cfg).cuda().float().eval()
    assert isinstance(m.model, VLSel) and isinstance(m.model.visual, RiceSel)
    sc = m.model.visual.importance_scorer
    assert (sc.in_features, sc.hidden_dim) == (128, 64) and m.model.visual.budgets == 1.0
    randomize_scorer(sc, seed=3)
    m.model.visual.budgets = 0.5
    inp = _ov_inputs(seed=6)
    with torch.no_grad():
        out, n_vis = m.model(**inp)
    assert n_vis == 64 and out.last_hidden_state.shape == (1, 32 + 11, 128)
    assert m.lm_head.weight.shape == (64, 128)


@pytest.mark.parametrize("impl_name", ["vsel_varlen", "vsel_flash_varlen"])
def test_native_attention_prefill_then_decode_matches_sdpa(impl_name):
    """attn_implementation = vsel_varlen / vsel_flash_varlen (the name transformers treats as a flash flavour: no 4-D mask,
    var-len metadata passed through) for a whole generate()-style run: prefill (var-len kernel), then decode steps and a
    3-token chunk against the cache (paged kernel, one page per batch row), vs the same bf16 model on SDPA."""
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from visionselector_amd.attention import ATTN_NAME, replace_qwen2_vl_attention_class
    replace_qwen2_vl_attention_class()
    torch.manual_seed(0)
    cfg = Qwen2Config(hidden_size=512, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=128, max_position_embeddings=2048, head_dim=128)
    model = Qwen2ForCausalLM(cfg).cuda().bfloat16().eval()
    ids = torch.randint(0, 128, (2, 37), device="cuda")
    steps = [torch.randint(0, 128, (2, 1), device="cuda"), torch.randint(0, 128, (2, 1), device="cuda"),
             torch.randint(0, 128, (2, 3), device="cuda")]

    def run(impl):
        model.config._attn_implementation = impl
        outs = []
        with torch.no_grad():
            o = model(input_ids=ids, use_cache=True)
            outs.append(o.logits.float())
            past = o.past_key_values
            for nxt in steps:
                o = model(input_ids=nxt, past_key_values=past, use_cache=True)
                outs.append(o.logits.float())
                past = o.past_key_values
        return outs

    ref = run("sdpa")
    from visionselector_amd import _native as N
    N.profile_start()
    got = run(impl_name)
    prof = N.profile_stop()
    assert prof["varlen_attn_fwd_kernel"][1] == 2 * 4, prof            # 2 layers x (prefill + 3 cache steps), all native
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        # TOLERANCE 3e-2 of the logits' max magnitude: two bf16 attention implementations through 2 bf16 layers
        assert float((a - b).abs().max()) <= 3e-2 * max(1.0, float(b.abs().max()))


def test_vision_tower_through_packed_native_attention_matches_sdpa():
    """Qwen2.5-VL vision tower (window + full attention layers, head_dim 64 here, 80 in the 7B) with the packed var-len
    kernel (ATTN_NAME_PACKED: one launch per layer over cu_seqlens) vs transformers' per-window SDPA loop."""
    from transformers import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    from visionselector_amd import _native as N
    from visionselector_amd.attention import ATTN_NAME_PACKED, replace_qwen2_vl_attention_class
    replace_qwen2_vl_attention_class()
    vc = Qwen2_5_VLConfig(vision_config=dict(depth=4, hidden_size=256, num_heads=4, intermediate_size=512, out_hidden_size=128,
                                             patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=112,
                                             fullatt_block_indexes=[1, 3], in_channels=3)).vision_config
    torch.manual_seed(0)
    tower = hf.Qwen2_5_VisionTransformerPretrainedModel(vc).cuda().bfloat16().eval()
    g = torch.Generator().manual_seed(2)
    pix = torch.randn(2 * 24 * 24 + 16 * 16, 3 * 2 * 14 * 14, generator=g).bfloat16().cuda()
    grid = torch.tensor([[2, 24, 24], [1, 16, 16]]).cuda()                    # a 2-frame video-like grid and a small image
    outs = {}
    for impl in ("sdpa", ATTN_NAME_PACKED):
        tower.config._attn_implementation = impl
        N.profile_start()
        with torch.no_grad():
            outs[impl] = tower(pix, grid).pooler_output.float()
        prof = N.profile_stop()
        if impl == ATTN_NAME_PACKED:
            assert prof["varlen_attn_fwd_kernel"][1] == 4, prof                # ONE launch per layer
    ref, got = outs["sdpa"], outs[ATTN_NAME_PACKED]
    # TOLERANCE 3e-2 of the output's max: two bf16 attention implementations through 4 bf16 ViT blocks + merger
    assert float((got - ref).abs().max()) <= 3e-2 * float(ref.abs().max())


def test_native_attention_padded_prefill_batch_matches_sdpa():
    """A right- and a left-padded prefill batch through the flash-flavoured interface (2-D token mask): unpad -> one packed
    var-len launch per layer -> pad back, vs SDPA with the same mask; decode against a padded cache raises."""
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from visionselector_amd.attention import ATTN_NAME_PACKED, replace_qwen2_vl_attention_class
    replace_qwen2_vl_attention_class()
    torch.manual_seed(0)
    cfg = Qwen2Config(hidden_size=512, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=128, max_position_embeddings=2048, head_dim=128)
    model = Qwen2ForCausalLM(cfg).cuda().bfloat16().eval()
    ids = torch.randint(0, 128, (3, 40), device="cuda")
    for side in ("right", "left"):
        mask = torch.ones(3, 40, dtype=torch.int64, device="cuda")
        if side == "right":
            mask[0, 25:] = 0
            mask[2, 33:] = 0
        else:
            mask[0, :15] = 0
            mask[2, :7] = 0
        outs = {}
        for impl in ("sdpa", ATTN_NAME_PACKED):
            model.config._attn_implementation = impl
            with torch.no_grad():
                outs[impl] = model(input_ids=ids, attention_mask=mask).logits.float()
        keep = mask.bool()
        a, b = outs[ATTN_NAME_PACKED][keep], outs["sdpa"][keep]             # logits at real tokens
        # TOLERANCE 3e-2 of the logits' max magnitude: two bf16 attention implementations through 2 bf16 layers
        assert float((a - b).abs().max()) <= 3e-2 * max(1.0, float(b.abs().max())), side
    model.config._attn_implementation = ATTN_NAME_PACKED
    with torch.no_grad():
        o = model(input_ids=ids, attention_mask=mask, use_cache=True)
        with pytest.raises(NotImplementedError, match="padded"):
            model(input_ids=ids[:, :1], attention_mask=torch.cat((mask, mask[:, :1] * 0 + 1), 1),
                  past_key_values=o.past_key_values, use_cache=True)


def test_model_constructs_and_loads_with_flash_flavoured_name(tmp_path):
    """attn_implementation="vsel_flash_varlen" at construction and at from_pretrained time (where the reference's harness
    passes "flash_attention_2"): transformers' flash pre-loading must accept the registered name."""
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from visionselector_amd.attention import ATTN_NAME_PACKED, replace_qwen2_vl_attention_class
    replace_qwen2_vl_attention_class()
    cfg = Qwen2Config(hidden_size=256, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                      num_key_value_heads=1, vocab_size=64, max_position_embeddings=512, head_dim=128)
    cfg._attn_implementation = ATTN_NAME_PACKED
    torch.manual_seed(0)
    m = Qwen2ForCausalLM(cfg).cuda().bfloat16().eval()
    assert m.config._attn_implementation == ATTN_NAME_PACKED
    m.save_pretrained(tmp_path / "m")
    m2 = Qwen2ForCausalLM.from_pretrained(tmp_path / "m", attn_implementation=ATTN_NAME_PACKED, dtype=torch.bfloat16).cuda().eval()
    assert m2.config._attn_implementation == ATTN_NAME_PACKED
    ids = torch.randint(0, 64, (1, 33), device="cuda")
    from visionselector_amd import _native as N
    N.profile_start()
    with torch.no_grad():
        a, b = m(input_ids=ids).logits, m2(input_ids=ids).logits
    assert N.profile_stop()["varlen_attn_fwd_kernel"][1] == 2
    # (m's rotary buffer went through .bfloat16(), m2's did not: compare loosely)
    assert float((a.float() - b.float()).abs().max()) <= 3e-2 * max(1.0, float(a.float().abs().max()))


def test_packed_prefill_matches_per_prompt_forward():
    """BASELINE config 5 at model level: three prompts with different image sizes (one with two images, one text-only ... no:
    one small image) served in ONE packed pass (tower once, ragged LIS, vsel_splice_batched, var-len attention over cu_seqlens')
    give the same kept tokens and last-token logits as the batch-1 *_Selector forward run prompt by prompt."""
    from transformers import Qwen2_5_VLConfig
    from visionselector_amd.attention import ATTN_NAME_PACKED, replace_qwen2_vl_attention_class
    from visionselector_amd.hf_qwen25vl import Qwen2_5_VLForConditionalGeneration_Selector
    from visionselector_amd.packed import packed_prefill
    replace_qwen2_vl_attention_class()
    cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                         num_key_value_heads=1, vocab_size=64, max_position_embeddings=4096,
                         rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=10000.0)),
        vision_config=dict(depth=2, hidden_size=64, num_heads=4, intermediate_size=128, out_hidden_size=256, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1],
                           in_channels=3),
        image_token_id=IMG, video_token_id=VID, vision_start_token_id=VSTART, vision_end_token_id=VEND)
    torch.manual_seed(0)
    model = Qwen2_5_VLForConditionalGeneration_Selector(cfg).cuda().bfloat16().eval()
    model.model.language_model.config._attn_implementation = ATTN_NAME_PACKED
    model.model.visual.config._attn_implementation = "sdpa"                  # head_dim 16 tower
    randomize_scorer(model.visual.importance_scorer, seed=5)
    model.visual.budgets = 0.25
    g = torch.Generator().manual_seed(3)
    grids = [[(1, 16, 16)], [(1, 32, 16), (1, 8, 8)], [(1, 24, 24)]]          # prompt 1 holds two images
    prompts, pix, flat_grids = [], [], []
    for gl in grids:
        parts = [torch.randint(20, 60, (int(torch.randint(3, 9, (1,), generator=g)),), generator=g)]
        for (t, hh, ww) in gl:
            n_vis = t * hh * ww // 4
            parts += [torch.tensor([VSTART]), torch.full((n_vis,), IMG), torch.tensor([VEND])]
            pix.append(torch.randn(t * hh * ww, 3 * 2 * 14 * 14, generator=g))
            flat_grids.append([t, hh, ww])
        parts.append(torch.randint(20, 60, (int(torch.randint(4, 12, (1,), generator=g)),), generator=g))
        prompts.append(torch.cat(parts))
    pix_all = torch.cat(pix).bfloat16().cuda()
    grid_all = torch.tensor(flat_grids).cuda()
    res = packed_prefill(model, prompts, pix_all, grid_all, [len(gl) for gl in grids])
    assert res["logits"].shape == (3, 64)
    # per-prompt reference: the batch-1 selector forward
    p0 = g0 = 0
    for b, (ids, gl) in enumerate(zip(prompts, grids)):
        n_patch = sum(t * hh * ww for t, hh, ww in gl)
        inp = dict(input_ids=ids[None].cuda(), attention_mask=torch.ones_like(ids)[None].cuda(),
                   pixel_values=pix_all[p0:p0 + n_patch], image_grid_thw=grid_all[g0:g0 + len(gl)],
                   mm_token_type_ids=(ids == IMG).int()[None].cuda())
        model.model.rope_deltas = None
        with torch.no_grad():
            o = model(**inp)
        n_vis = int((ids == IMG).sum())
        assert res["kept"][b] == max(1, int(n_vis * 0.25)) == int(model.visual.last_selected_indices.numel())
        a0, a1 = int(res["cu_seqlens"][b]), int(res["cu_seqlens"][b + 1])
        assert a1 - a0 == o.logits.shape[1]
        ref = o.logits[0, -1].float()
        got = res["logits"][b].float()
        # TOLERANCE 3e-2 of the logits' max: the packed pass and the per-prompt pass run different GEMM shapes in bf16
        assert float((got - ref).abs().max()) <= 3e-2 * max(1.0, float(ref.abs().max())), b
        p0 += n_patch
        g0 += len(gl)
    with pytest.raises(ValueError, match="do not match"):
        packed_prefill(model, [prompts[0][1:], prompts[1], prompts[2]][::-1], pix_all, grid_all, [1, 2, 1])


def test_selector_loads_reference_era_checkpoint_layout(tmp_path):
    """Released VisionSelector checkpoints were written by the reference on transformers 4.50: keys `visual.*`
    (incl. `visual.importance_scorer.{q_proj,k_proj}.{weight,bias}`), `model.layers.*`, `lm_head.*`.  The drop-in class must
    load that layout through from_pretrained (transformers 5.x renames it to model.visual.* / model.language_model.*)."""
    from safetensors.torch import save_file
    from visionselector_amd.hf_qwen25vl import Qwen2_5_VLForConditionalGeneration_Selector
    torch.manual_seed(0)
    src = Qwen2_5_VLForConditionalGeneration_Selector(tiny_config()).float()
    randomize_scorer(src.visual.importance_scorer, seed=12)
    old = {}
    for k, v in src.state_dict().items():
        if k.startswith("model.visual."):
            nk = k[len("model."):]                                   # visual.*
        elif k.startswith("model.language_model."):
            nk = "model." + k[len("model.language_model."):]         # model.layers.*, model.embed_tokens.*, model.norm.*
        else:
            nk = k                                                   # lm_head.*
        old[nk] = v.detach().clone().contiguous()
    assert "visual.importance_scorer.q_proj.weight" in old and "model.layers.0.self_attn.q_proj.weight" in old
    d = tmp_path / "ckpt"
    d.mkdir()
    src.config.save_pretrained(d)
    save_file(old, str(d / "model.safetensors"), metadata={"format": "pt"})
    m = Qwen2_5_VLForConditionalGeneration_Selector.from_pretrained(d, dtype=torch.float32)
    sd = m.state_dict()
    for k, v in src.state_dict().items():
        assert torch.equal(sd[k], v), k
    assert m.visual.budgets == 1.0 and isinstance(m.visual.importance_scorer.q_proj, torch.nn.Linear)


def test_selector_video_prompt_generates(selector_model):
    """Video branch (pixel_values_videos / video_grid_thw, EV/token_compression/selector_model.py:264-298): the kept video
    tokens are spliced by the same kernel, generation continues on the compressed cache, and the prefill equals a manual
    torch splice of the same kept tokens."""
    m = selector_model
    m.visual.budgets = 0.25
    g = torch.Generator().manual_seed(13)
    grid = (2, 16, 16)                                            # 2 temporal patches -> 128 merged tokens
    n_patches = grid[0] * grid[1] * grid[2]
    n_vis = n_patches // 4
    pix = torch.randn(n_patches, 3 * 2 * 14 * 14, generator=g)
    ids = torch.cat((torch.randint(20, 60, (6,), generator=g), torch.tensor([VSTART]), torch.full((n_vis,), VID),
                     torch.tensor([VEND]), torch.randint(20, 60, (8,), generator=g)))[None]
    mm = torch.zeros_like(ids, dtype=torch.int32)
    mm[ids == VID] = 2
    inp = dict(input_ids=ids.cuda(), attention_mask=torch.ones_like(ids).cuda(), pixel_values_videos=pix.cuda(),
               video_grid_thw=torch.tensor([list(grid)]).cuda(), mm_token_type_ids=mm.cuda())
    with torch.no_grad():
        m.model.rope_deltas = None
        out = m(**inp)
        k = int(m.visual.last_selected_indices.numel())
        assert k == max(1, int(n_vis * 0.25)) and out.logits.shape[1] == ids.shape[1] - n_vis + k
        first = out.logits[0, -1].argmax()
        m.model.rope_deltas = None
        gen = m.generate(**inp, max_new_tokens=3, do_sample=False)
    assert gen.shape[1] == ids.shape[1] + 3 and int(gen[0, ids.shape[1]]) == int(first)


def test_training_with_gradient_checkpointing_matches_plain_backward():
    """HF training enables gradient checkpointing (qwen-vl-finetune/scripts/sft_7b.sh: --gradient_checkpointing True): the
    native autograd functions (LIS block, var-len attention) must give the same scorer gradients when the LLM layers are
    re-run inside backward."""
    from transformers import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    from visionselector_amd.attention import ATTN_NAME_PACKED, replace_qwen2_vl_attention_class
    from visionselector_amd.hf_qwen25vl import install_selector
    replace_qwen2_vl_attention_class()
    cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                         num_key_value_heads=1, vocab_size=64, max_position_embeddings=4096,
                         rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=10000.0)),
        vision_config=dict(depth=2, hidden_size=64, num_heads=4, intermediate_size=128, out_hidden_size=256, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1],
                           in_channels=3),
        image_token_id=IMG, video_token_id=VID, vision_start_token_id=VSTART, vision_end_token_id=VEND)
    torch.manual_seed(0)
    model = hf.Qwen2_5_VLForConditionalGeneration(cfg).cuda().bfloat16().train()
    install_selector(model, budget=0.25, regularization_weight=0.7)
    model.model.language_model.config._attn_implementation = ATTN_NAME_PACKED
    visual = model.model.visual
    randomize_scorer(visual.importance_scorer, seed=2)
    for n, p in model.named_parameters():
        p.requires_grad = "importance_scorer" in n
    inp, _ = make_inputs(grid=(1, 32, 32), seed=5)
    inp["pixel_values"] = inp["pixel_values"].bfloat16()
    labels = inp["input_ids"].clone()
    labels[inp["input_ids"] == IMG] = -100
    grads = []
    for ckpt in (False, True):
        if ckpt:
            model.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"use_reentrant": False})
        for p in visual.importance_scorer.parameters():
            p.grad = None
        out = model(**inp, labels=labels)
        out.loss.backward()
        grads.append((float(out.loss.detach()), [p.grad.float().clone() for p in visual.importance_scorer.parameters()]))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * max(1.0, abs(grads[0][0]))
    for a, b in zip(grads[0][1], grads[1][1]):
        assert torch.equal(a, b)            # deterministic kernels: recomputation reproduces the same bits


def test_eval_time_lines_through_generate_and_log_reader(selector_model, capfd, tmp_path, monkeypatch):
    """N3: EVAL_TIME=true -> the *_Selector forward prints the prefill lines (EV/token_compression/selector_model.py:353-359),
    timed_generate the adapter's latency / memory lines (lmms-eval/.../qwen2_5_vl_with_token_compression.py:370-394), and the
    log reader returns the four averages qwen-evaluation/extract_time.py:4-69 prints."""
    from visionselector_amd import evaltime
    m = selector_model
    m.visual.budgets = 0.25
    monkeypatch.setenv("EVAL_TIME", "true")
    n_samples = 3
    kept = []
    for s in range(n_samples):
        inp, n_vis = make_inputs(seed=20 + s)
        m.model.rope_deltas = None
        with torch.no_grad():
            gen = evaltime.timed_generate(m, **inp, max_new_tokens=3, do_sample=False)
        assert gen.shape[1] == inp["input_ids"].shape[1] + 3
        kept.append(n_vis)
    text = capfd.readouterr().out
    lines = text.splitlines()
    for needle in ("Input visual token number is:", "Generation prefill time is:", "Generation latency time is:",
                   "after generation memory:"):
        assert sum(needle in ln for ln in lines) == n_samples, (needle, text)
    log = tmp_path / "log_eval.log"
    log.write_text(text)
    s = evaltime.summarize_log(str(log))
    assert s["samples"] == n_samples
    assert s["avg_visual_tokens"] == pytest.approx(sum(kept) / n_samples)     # the tower's total_token_num (EV :187,357)
    assert 0 < s["avg_prefill_ms"] <= s["avg_latency_ms"]
    assert s["avg_max_memory_GB"] > 0
    monkeypatch.setenv("EVAL_TIME", "false")
    inp, _ = make_inputs(seed=29)
    m.model.rope_deltas = None
    with torch.no_grad():
        evaltime.timed_generate(m, **inp, max_new_tokens=2, do_sample=False)
    assert "Generation" not in capfd.readouterr().out


@pytest.mark.parametrize("impl", ["sdpa", "vsel_varlen"])
def test_text_only_request_after_an_image_request_is_not_truncated(impl):
    """A compressed image prompt leaves per-request state behind (dropped-column count for the decode mask, video mask);
    the next request on an empty cache must not inherit it: text-only generate() == the stock model's."""
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    from visionselector_amd.attention import replace_qwen2_vl_attention_class
    from visionselector_amd.hf_qwen25vl import Qwen2_5_VLForConditionalGeneration_Selector
    replace_qwen2_vl_attention_class()
    cfg = tiny_config()
    if impl != "sdpa":                         # the HIP attention needs head_dim 128: 512 / 4 heads, tower output to match
        cfg.text_config.hidden_size, cfg.text_config.intermediate_size = 512, 512
        cfg.vision_config.out_hidden_size = 512
        rp = dict(cfg.text_config.rope_parameters)
        rp["mrope_section"] = [16, 24, 24]                                        # sums to head_dim / 2
        cfg.text_config.rope_parameters = rp
    torch.manual_seed(0)
    dt = torch.bfloat16 if impl != "sdpa" else torch.float32
    m = Qwen2_5_VLForConditionalGeneration_Selector(cfg).cuda().to(dt).eval()
    randomize_scorer(m.visual.importance_scorer)
    m.config.text_config._attn_implementation = impl
    m.model.language_model.config._attn_implementation = impl
    m.visual.budgets = 0.25
    inp, n_vis = make_inputs(seed=31)
    inp["pixel_values"] = inp["pixel_values"].to(dt)
    with torch.no_grad():
        m.generate(**inp, max_new_tokens=3, do_sample=False)
    assert m._n_dropped == n_vis - max(1, int(n_vis * 0.25))
    ids = torch.randint(20, 60, (2, 12), generator=torch.Generator().manual_seed(5)).cuda()
    mask = torch.ones_like(ids)
    if impl == "sdpa":
        mask[1, :4] = 0            # left-padded batch: a truncated mask would be visible at once
    text_kwargs = dict(input_ids=ids[:1] if impl != "sdpa" else ids, attention_mask=mask[:1] if impl != "sdpa" else mask)
    with torch.no_grad():
        got = m.generate(**text_kwargs, max_new_tokens=4, do_sample=False)
        ref = hf.Qwen2_5_VLForConditionalGeneration.generate(m, **text_kwargs, max_new_tokens=4, do_sample=False)
    assert m._n_dropped == 0
    assert torch.equal(got, ref)


def test_video_branch_publishes_text_image_mask_on_every_layer(selector_model):
    """EV/token_compression/selector_model.py:295-298: text_image_mask on the text model and on each layer's self_attn; a
    following image request clears it."""
    m = selector_model
    m.visual.budgets = 0.5
    g = torch.Generator().manual_seed(17)
    grid = (2, 16, 16)
    n_patches = grid[0] * grid[1] * grid[2]
    n_vis = n_patches // 4
    pix = torch.randn(n_patches, 3 * 2 * 14 * 14, generator=g)
    ids = torch.cat((torch.randint(20, 60, (4,), generator=g), torch.tensor([VSTART]), torch.full((n_vis,), VID),
                     torch.tensor([VEND]), torch.randint(20, 60, (5,), generator=g)))[None]
    mm = torch.zeros_like(ids, dtype=torch.int32)
    mm[ids == VID] = 2
    with torch.no_grad():
        m.model.rope_deltas = None
        out = m(input_ids=ids.cuda(), attention_mask=torch.ones_like(ids).cuda(), pixel_values_videos=pix.cuda(),
                video_grid_thw=torch.tensor([list(grid)]).cuda(), mm_token_type_ids=mm.cuda())
    lm = m.model.language_model
    k = n_vis // 2
    assert lm.text_image_mask.shape == (1, out.logits.shape[1]) and int((~lm.text_image_mask).sum()) == k
    for layer in lm.layers:
        assert layer.self_attn.text_image_mask is lm.text_image_mask
    inp, _ = make_inputs(seed=2)
    with torch.no_grad():
        m.model.rope_deltas = None
        m(**inp)
    assert lm.text_image_mask is None and all(layer.self_attn.text_image_mask is None for layer in lm.layers)


def test_splice_reports_placeholder_feature_count_mismatch():
    """Placeholder count in input_ids != visual feature count: the reference raises ValueError
    (FT/compression_method/selector_model.py:210-213).  ops.splice(check=True) -- what the *_Selector forward uses -- raises
    the same; unchecked, the kernels still never read past their inputs (rows they cannot produce come out as zeros)."""
    from visionselector_amd import ops
    g = torch.Generator().manual_seed(4)
    n_vis, k, d = 64, 16, 128
    all_idx = torch.sort(torch.randperm(n_vis, generator=g)[:k]).values.cuda()
    vis = torch.randn(k, d, generator=g).cuda().bfloat16()
    for n_placeholders in (n_vis - 3, n_vis + 3):
        ids = torch.cat((torch.randint(20, 60, (5,), generator=g), torch.full((n_placeholders,), IMG),
                         torch.randint(20, 60, (9,), generator=g)))[None].cuda()
        emb = torch.randn(1, ids.shape[1], d, generator=g).cuda().bfloat16()
        with pytest.raises(ValueError, match="do not match"):
            ops.splice(ids, emb, IMG, all_idx, vis, n_vis, check=True)
        sel, new_ids, new_emb, _, _ = ops.splice(ids, emb, IMG, all_idx, vis, n_vis, check=False)
        torch.cuda.synchronize()
        l_out = ids.shape[1] - n_vis + k
        assert new_emb.shape == (1, l_out, d) and bool(torch.isfinite(new_emb.float()).all())
        if n_placeholders > n_vis:        # 3 placeholders have no feature row: 3 output rows cannot be produced
            produced = int((sel >= 0).sum())
            assert produced == l_out - 3 and float(new_emb[0, produced:].float().abs().max()) == 0.0
            assert bool((new_ids[0, produced:] == -1).all())


def test_ops_follow_the_device_of_their_tensors():
    """ops.* switch to the tensors' device before launching (ATen semantics): on a multi-GPU box a call with tensors on
    cuda:1 while cuda:0 is current runs on cuda:1's stream; on one GPU the guard is a no-op."""
    from visionselector_amd import ops
    dev = torch.device("cuda", torch.cuda.device_count() - 1)
    g = torch.Generator().manual_seed(3)
    h = torch.randn(96, 64, generator=g).to(dev).bfloat16()
    wq, wk = (0.05 * torch.randn(32, 64, generator=g)).to(dev).bfloat16(), (0.05 * torch.randn(32, 64, generator=g)).to(dev).bfloat16()
    bq, bk = torch.zeros(32, device=dev).bfloat16(), torch.zeros(32, device=dev).bfloat16()
    torch.cuda.set_device(0)
    out, idx, scores = ops.lis_select(h, wq, bq, wk, bk, 20)
    assert out.device == dev and torch.cuda.current_device() == 0
    torch.cuda.synchronize(dev)
    assert torch.equal(out, h[idx])
    ref = eager_scores(h.float(), type("S", (), {"k_proj": type("L", (), {"weight": wk.float(), "bias": bk.float()}),
                                                  "q_proj": type("L", (), {"weight": wq.float(), "bias": bq.float()}),
                                                  "hidden_dim": 32}))
    assert float((scores - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


# ---- LLaVA-OV surface against goldens produced by the reference's own OV classes (tests/golden/ov*.npz) -------------------
@pytest.mark.parametrize("name", ["8x81", "3ragged"])
def test_llavaov15_tower_block_matches_reference_golden(golden_dir, name):
    """llavaov15_vision_tower_forward_selector_eval around a tower that hands over the reference's LIS input: indices
    bit-exact, kept rows exact, last_combined_scores within 1e-5 (fp32) of RiceTransformerPretrainedModel_Selector.forward
    (llava-ov-15/compression_method/modeling_selector.py:173-184), several images scored jointly."""
    import os
    import types
    from oracle import inputs as oin
    from visionselector_amd.hf_generic import make_vision_tower_forward_selector
    from visionselector_amd.selector import TransformerScorer
    g = np.load(os.path.join(golden_dir, f"ovlis_{name}.npz"))
    d, hd, n = int(g["d"]), int(g["hd"]), int(g["n"])
    c = oin.make_case(d, hd, n, int(g["seed"]))
    m = g["lis_rowmap"]
    h = torch.from_numpy(np.where((m >= 0)[:, None], c["h"][np.clip(m, 0, None)], np.float32(0)).astype(np.float32)).cuda()

    class Tower(torch.nn.Module):
        def forward(self, hidden_states, grid_thw, **kw):
            return hidden_states

    for dt, exact in ((torch.float32, True), (torch.bfloat16, False)):
        tower = Tower()
        tower.importance_scorer = TransformerScorer(d, hd).cuda().to(dt)
        with torch.no_grad():
            tower.importance_scorer.q_proj.weight.copy_(torch.from_numpy(c["wq"]))
            tower.importance_scorer.q_proj.bias.copy_(torch.from_numpy(c["bq"]))
            tower.importance_scorer.k_proj.weight.copy_(torch.from_numpy(c["wk"]))
            tower.importance_scorer.k_proj.bias.copy_(torch.from_numpy(c["bk"]))
        tower.forward = types.MethodType(make_vision_tower_forward_selector(Tower.forward, "eval"), tower)
        for r in oin.BUDGETS:
            tag = str(r).replace(".", "p")
            tower.budgets = r
            with torch.no_grad():
                kept, idx, total = tower(h.to(dt), torch.from_numpy(g["grids"]).cuda())
            assert total == n and np.array_equal(idx.cpu().numpy(), g[f"idx_{tag}"])        # inputs are bf16-representable
            assert torch.equal(kept, h.to(dt)[idx])
            tol = 1e-5 if exact else 4e-3                                                       # bf16 storage of the probabilities
            assert np.abs(tower.last_combined_scores.float().cpu().numpy() - g[f"ps_{tag}"]).max() <= tol


@pytest.mark.parametrize("name", ["a", "b"])
def test_llavaov15_model_splice_matches_reference_golden(golden_dir, name):
    """llavaov15_vlmodel_forward_selector_eval hands the language model exactly what LLaVAOneVision1_5_Model_Selector.forward
    does (modeling_selector.py:259-276, :308-314): embeds, 1-D position_ids, cache_position, attention_mask -- bit-exact."""
    import os
    from oracle import inputs as oin
    from visionselector_amd.hf_llavaov15 import llavaov15_vlmodel_forward_selector_eval
    g = np.load(os.path.join(golden_dir, f"ovsplice_{name}.npz"))
    n_visual, d = int(g["n_visual"]), int(g["d_llm"])
    image_token = 151655
    ids = torch.from_numpy(oin.make_prompt(n_visual, int(g["n_pre"]), int(g["n_post"]), image_token, int(g["seed"]))).cuda()
    L = ids.shape[1]
    seen = {}

    class LM(torch.nn.Module):
        def forward(self, **kw):
            seen.update(kw)
            import types as _t
            return _t.SimpleNamespace(last_hidden_state=kw["inputs_embeds"], past_key_values=None, hidden_states=None, attentions=None)

    class VL(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.language_model = LM()
            self.config = type("C", (), {"image_token_id": image_token, "video_token_id": 151656, "output_attentions": False,
                                         "output_hidden_states": False, "use_return_dict": True})()
            self.rope_deltas = None

        def get_input_embeddings(self):
            ar = torch.arange(d, device="cuda")
            return lambda t: (((t[..., None] * 31 + ar * 17) % 257).float() / 257.0)

        def get_image_features(self, pixel_values, grid):
            return torch.from_numpy(g["vis_embeds"]).cuda(), torch.from_numpy(g["all_idx"]).cuda(), n_visual

    kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=torch.zeros(1, 1, device="cuda"),
              image_grid_thw=torch.tensor([[1, 2, n_visual // 2]]).cuda(), use_cache=False, cache_position=torch.arange(L, device="cuda"))
    if bool(g["with_position_ids"]):
        kw["position_ids"] = (torch.arange(L, device="cuda") + 5)[None]
    with torch.no_grad():
        _, n_vis = llavaov15_vlmodel_forward_selector_eval(VL(), **kw)
    assert n_vis == n_visual
    assert np.array_equal(seen["inputs_embeds"].cpu().numpy(), g["inputs_embeds"])
    assert np.array_equal(seen["position_ids"].cpu().numpy(), g["position_ids"])
    assert np.array_equal(seen["cache_position"].cpu().numpy(), g["cache_position"])
    assert np.array_equal(seen["attention_mask"].cpu().numpy(), g["attention_mask"])
