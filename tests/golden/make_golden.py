#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's own modules.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite only read
the committed ``*.npz`` files.  Nothing from the reference is copied: the script calls the
reference's functions on seeded inputs (oracle/inputs.py) and stores inputs' seeds + outputs.

What is executed from the reference (paths under /root/reference):
  * qwen-vl-finetune/compression_method/selector_scorer.py   TransformerScorer.forward
  * qwen-vl-finetune/compression_method/selector_model.py    topk/TopK/_find_ts and
        qwen25vl_vision_tower_forward_selector (the training LIS block, lines 158-173) run on a
        stub vision tower whose patch_embed/blocks/merger are identities
  * qwen-evaluation/token_compression/selector_model.py      Qwen2_5_VisionTransformerPretrainedModel_Selector.forward
        (inference LIS block, lines 182-194) on the same stub, and
        Qwen2_5_VLForConditionalGeneration_Selector.forward (splice, lines 243-320) on a stub LLM that
        records what it is handed; get_rope_index is the vendored one
        (qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:1550).
  * qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py          Qwen2_5_VLAttention.forward (the in-tree EAGER attention,
        lines 749-800: repeat_kv, QK^T/sqrt(d) + mask, fp32 softmax, PV) and torch autograd through it, with
        selection-matrix projections and an identity rotation so that it computes attention of the given q, k, v
  * llava-ov-15/compression_method/modeling_selector.py      RiceTransformerPretrainedModel_Selector.forward (CLS insertion /
        removal loops :131-166 and the inference LIS block :173-184) on a stub Rice tower (identity patch_embed / blocks /
        merger), and LLaVAOneVision1_5_Model_Selector.forward (splice :259-276, 1-D position_ids / cache_position /
        attention_mask selection :308-314) on a recording language model

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--ov-only | --bf16-only]
        (--bf16-only: the same LIS cases with the reference modules and tokens in bfloat16 -> lisbf16_*.npz)
"""
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import inputs as oin  # noqa: E402

torch.set_grad_enabled(True)
IMAGE_TOKEN, VIDEO_TOKEN, VSTART, VEND = 151655, 151656, 151652, 151653


def _purge(prefixes):
    for k in [k for k in sys.modules if k.split(".")[0] in prefixes]:
        del sys.modules[k]


def load_ft():
    sys.path.insert(0, f"{REF}/qwen-vl-finetune")
    from compression_method import selector_model as ft
    from compression_method.selector_scorer import TransformerScorer
    sys.path.remove(f"{REF}/qwen-vl-finetune")
    _purge({"compression_method"})
    return ft, TransformerScorer


def load_ev():
    for name in ("flash_attn", "flash_attn.bert_padding", "flash_attn.layers", "flash_attn.layers.rotary"):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m
    fa = sys.modules["flash_attn"]
    fa.flash_attn_func = fa.flash_attn_varlen_func = None
    bp = sys.modules["flash_attn.bert_padding"]
    bp.index_first_axis = bp.pad_input = bp.unpad_input = None
    sys.modules["flash_attn.layers.rotary"].apply_rotary_emb = None
    sys.path.insert(0, f"{REF}/qwen-evaluation")
    from token_compression import selector_model as ev
    sys.path.remove(f"{REF}/qwen-evaluation")
    return ev


class StubTower:
    """Identity vision tower around the reference's LIS block."""

    def __init__(self, scorer, budgets, n, training):
        self.importance_scorer = scorer
        self.budgets = budgets
        self.n = n
        self.spatial_merge_unit = 1
        self.blocks = []
        self.fullatt_block_indexes = []
        self.gradient_checkpointing = False
        self.training = training

    def patch_embed(self, x):
        return x

    def rot_pos_emb(self, grid):
        return torch.zeros(self.n, 2)

    def get_window_index(self, grid):
        return torch.arange(self.n), [0, self.n]

    def merger(self, x):
        return x


def build_scorer(TransformerScorer, case):
    hd, d = case["wq"].shape
    m = TransformerScorer(in_features=d, hidden_dim=hd)
    with torch.no_grad():
        m.q_proj.weight.copy_(torch.from_numpy(case["wq"]))
        m.q_proj.bias.copy_(torch.from_numpy(case["bq"]))
        m.k_proj.weight.copy_(torch.from_numpy(case["wk"]))
        m.k_proj.bias.copy_(torch.from_numpy(case["bk"]))
    return m


def gen_lis_case(name, d, hd, n, seed, ft, ev, TransformerScorer):
    case = oin.make_case(d, hd, n, seed)
    scorer = build_scorer(TransformerScorer, case)
    h = torch.from_numpy(case["h"])
    grid = torch.tensor([[1, 1, n]])
    out = {"d": d, "hd": hd, "n": n, "seed": seed}

    # ---- inference (EV) -------------------------------------------------------------------------
    for r in oin.BUDGETS:
        stub = StubTower(scorer, r, n, training=False)
        with torch.no_grad():
            h_new, idx, total = ev.Qwen2_5_VisionTransformerPretrainedModel_Selector.forward(stub, h, grid)
        tag = str(r).replace(".", "p")
        out[f"idx_{tag}"] = idx.numpy().astype(np.int64)
        out[f"ps_{tag}"] = stub.last_combined_scores.numpy().astype(np.float32)
        assert total == n and torch.equal(h_new, h[idx])
        assert torch.equal(stub.last_selected_indices, idx)
    with torch.no_grad():
        scores = scorer(h[None])[0]
    out["scores"] = scores.numpy().astype(np.float32)
    srt = np.sort(out["scores"])[::-1]
    out["gaps"] = np.array([srt[max(1, int(n * r)) - 1] - srt[max(1, int(n * r))] for r in oin.BUDGETS], np.float32)
    out["score_std"] = np.float32(out["scores"].std())

    # ---- differentiable top-k in isolation (FT) -------------------------------------------------
    k = int(n * 0.2)
    xs = scores[None].clone().requires_grad_(True)
    ps = ft.topk(xs, k)
    g = torch.from_numpy(oin.make_vec(n, seed + 1000))[None]
    ps.backward(g)
    with torch.no_grad():
        ts, _ = ft._find_ts(scores[None], k)
    out["topk_k"] = k
    out["topk_ps"] = ps.detach().numpy()[0].astype(np.float32)
    out["topk_ts"] = np.float32(ts.item())
    out["topk_grad"] = xs.grad.numpy()[0].astype(np.float32)

    # ---- training LIS block + constraint loss + backward (FT) -----------------------------------
    reg_w = 0.7
    stub = StubTower(scorer, 0.2, n, training=True)
    for p in scorer.parameters():
        p.grad = None
    hg = h.clone().requires_grad_(True)
    h_new, img_mask, cmask = ft.qwen25vl_vision_tower_forward_selector(stub, hg, grid)
    gmat = torch.from_numpy(
        np.random.default_rng(seed + 2000).standard_normal((n, d), dtype=np.float32) / np.float32(d) ** 0.5)
    bce = F.binary_cross_entropy(img_mask, cmask)            # FT/.../selector_model.py:310
    loss = (h_new * gmat).sum() + reg_w * bce                # :311 with a linear stand-in for the LLM loss
    loss.backward()
    out["train_reg_w"] = np.float32(reg_w)
    out["train_ps"] = img_mask.detach().numpy().astype(np.float32)
    out["train_y"] = cmask.numpy().astype(np.float32)
    out["train_bce"] = np.float32(bce.item())
    out["train_loss"] = np.float32(loss.item())
    out["train_hnew_rowsum"] = h_new.detach().double().sum(1).numpy()
    u_d = oin.make_vec(d, seed + 3000).astype(np.float64)
    v_h = oin.make_vec(hd, seed + 3001).astype(np.float64)
    v_n = oin.make_vec(n, seed + 3002).astype(np.float64)
    gq = scorer.q_proj.weight.grad.double().numpy()
    gk = scorer.k_proj.weight.grad.double().numpy()
    gx = hg.grad.double().numpy()
    out["dbq"] = scorer.q_proj.bias.grad.numpy().astype(np.float32)
    out["dbk"] = scorer.k_proj.bias.grad.numpy().astype(np.float32)
    out["dwq_u"] = gq @ u_d
    out["v_dwq"] = v_h @ gq
    out["dwk_u"] = gk @ u_d
    out["v_dwk"] = v_h @ gk
    out["dx_u"] = gx @ u_d
    out["v_dx"] = v_n @ gx
    if n * d <= 4096:
        out["dwq"] = gq.astype(np.float32)
        out["dwk"] = gk.astype(np.float32)
        out["dx"] = gx.astype(np.float32)
        out["train_hnew"] = h_new.detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, f"lis_{name}.npz"), **out)
    print(f"{name}: N={n} gaps/std={out['gaps'] / out['score_std']} bce={out['train_bce']:.6f} ts={out['topk_ts']:.6f}")


def gen_lis_bf16_case(name, d, hd, n, seed, ft, ev, TransformerScorer):
    """The reference as its users run it: scorer module and tokens in bfloat16 (FT/qwenvl/train/train_qwen_selector.py:175-180
    loads and trains in bf16; `_find_ts` runs in the dtype of its input, FT/compression_method/selector_model.py:72-86).
    Same seeded inputs as lis_<name>.npz (bf16-representable), every module op rounding to bf16 as torch CPU does.
    Stored: bf16 scores, EV selection (indices, soft mask `last_combined_scores`) for the three budgets, `_find_ts` in bf16,
    the training forward (soft mask, constraint mask, BCE), the tie census at each k boundary, and the bf16 BACKWARD:
    `TopK.backward` for a seeded g and the training block's autograd gradients (dWq, dWk, dx as the projections the fp32
    fixtures use; db in full)."""
    case = oin.make_case(d, hd, n, seed)
    scorer = build_scorer(TransformerScorer, case).bfloat16()
    h = torch.from_numpy(case["h"]).bfloat16()
    grid = torch.tensor([[1, 1, n]])
    out = {"d": d, "hd": hd, "n": n, "seed": seed}
    with torch.no_grad():
        scores = scorer(h[None])[0]
    assert scores.dtype == torch.bfloat16
    out["scores_bf16"] = scores.float().numpy()
    sf = out["scores_bf16"]
    for r in oin.BUDGETS:
        stub = StubTower(scorer, r, n, training=False)
        with torch.no_grad():
            h_new, idx, total = ev.Qwen2_5_VisionTransformerPretrainedModel_Selector.forward(stub, h, grid)
        tag = str(r).replace(".", "p")
        k = max(1, int(n * r))
        assert idx.numel() == k and torch.equal(h_new, h[idx])
        out[f"idx_bf16_{tag}"] = idx.numpy().astype(np.int64)
        ps = stub.last_combined_scores
        assert ps.dtype == torch.bfloat16
        out[f"ps_bf16_{tag}"] = ps.float().numpy()
        out[f"sum_ps_bf16_{tag}"] = np.float64(ps.double().sum().item())
        kth = np.sort(sf)[::-1][k - 1]
        out[f"ties_at_kth_{tag}"] = np.int64((sf == kth).sum())          # candidates tied at the k-th value (torch.topk
        out[f"above_kth_{tag}"] = np.int64((sf > kth).sum())              # breaks them in an unspecified order)
    k = int(n * 0.2)
    with torch.no_grad():
        ts, ps = ft._find_ts(scores[None], k)
    out["topk_k"] = k
    out["ts_bf16"] = np.float32(ts.float().item())
    out["ps_bf16"] = ps[0].float().numpy()
    out["sum_ps_bf16"] = np.float64(ps.double().sum().item())
    stub = StubTower(scorer, 0.2, n, training=True)
    with torch.no_grad():
        h_new, img_mask, cmask = ft.qwen25vl_vision_tower_forward_selector(stub, h, grid)
        bce = F.binary_cross_entropy(img_mask.float(), cmask.float())
    out["train_ps_bf16"] = img_mask.float().numpy()
    out["train_y_bf16"] = cmask.float().numpy()
    out["train_bce_bf16"] = np.float32(bce.item())
    out["train_hnew_rowsum_bf16"] = h_new.double().sum(1).numpy()
    # ---- bf16 BACKWARD (the reference trains in bf16: train_qwen_selector.py:175-180 loads bf16 and DeepSpeed's bf16 engine
    # casts the scorer added at :190-197; selector_model.py:60-70 TopK.backward, :158-173 block, :308-313 BCE) -------------
    xs = scores[None].clone().requires_grad_(True)
    ps_t = ft.topk(xs, k)                                           # TopK.apply on bf16 scores
    gvec = torch.from_numpy(oin.make_vec(n, seed + 1000))[None].bfloat16()
    ps_t.backward(gvec)
    assert xs.grad.dtype == torch.bfloat16
    out["topk_grad_bf16"] = xs.grad[0].float().numpy()
    reg_w = 0.7
    stub = StubTower(scorer, 0.2, n, training=True)
    for p in scorer.parameters():
        p.grad = None
    hg = h.clone().requires_grad_(True)
    h_new, img_mask, cmask = ft.qwen25vl_vision_tower_forward_selector(stub, hg, grid)
    gmat = torch.from_numpy(
        np.random.default_rng(seed + 2000).standard_normal((n, d), dtype=np.float32) / np.float32(d) ** 0.5).bfloat16()
    bce_b = F.binary_cross_entropy(img_mask, cmask)                 # bf16 in, bf16 out, as the reference's line :310 runs
    loss = (h_new * gmat).sum() + reg_w * bce_b
    loss.backward()
    assert hg.grad.dtype == torch.bfloat16 and scorer.q_proj.weight.grad.dtype == torch.bfloat16
    out["bwd_reg_w"] = np.float32(reg_w)
    out["bwd_bce_bf16"] = np.float32(bce_b.float().item())
    u_d = oin.make_vec(d, seed + 3000).astype(np.float64)
    v_h = oin.make_vec(hd, seed + 3001).astype(np.float64)
    v_n = oin.make_vec(n, seed + 3002).astype(np.float64)
    gq = scorer.q_proj.weight.grad.double().numpy()
    gk = scorer.k_proj.weight.grad.double().numpy()
    gx = hg.grad.double().numpy()
    out["bwd_dbq"] = scorer.q_proj.bias.grad.float().numpy()
    out["bwd_dbk"] = scorer.k_proj.bias.grad.float().numpy()
    out["bwd_dwq_u"], out["bwd_v_dwq"] = gq @ u_d, v_h @ gq
    out["bwd_dwk_u"], out["bwd_v_dwk"] = gk @ u_d, v_h @ gk
    out["bwd_dx_u"], out["bwd_v_dx"] = gx @ u_d, v_n @ gx
    out["bwd_max_abs"] = np.array([np.abs(gq).max(), np.abs(gk).max(), np.abs(gx).max()], np.float64)
    if n * d <= 4096:
        out["bwd_dwq"], out["bwd_dwk"], out["bwd_dx"] = gq.astype(np.float32), gk.astype(np.float32), gx.astype(np.float32)
    g32 = np.load(os.path.join(HERE, f"lis_{name}.npz"))
    d32 = np.abs(sf - g32["scores"]).max()
    sym = {t: len(set(out[f"idx_bf16_{t}"]) ^ set(g32[f"idx_{t}"])) for t in ("0p1", "0p2", "0p5")}
    np.savez_compressed(os.path.join(HERE, f"lisbf16_{name}.npz"), **out)
    print(f"bf16 {name}: N={n} max|s_bf16 - s_fp32|={d32:.3e} (max|s|={np.abs(g32['scores']).max():.3f}) "
          f"sym.diff vs fp32 idx={sym} sum(ps)={out['sum_ps_bf16']:.3f} (k={k}) ts={out['ts_bf16']:.5f}")


# ---------------------------------------------------------------------------------------------------
# splice goldens
# ---------------------------------------------------------------------------------------------------


def embed_fn(ids, d_llm):
    """Deterministic exact-in-fp32 stand-in for embed_tokens (the test recomputes it in numpy)."""
    ar = torch.arange(d_llm, dtype=torch.int64)
    return (((ids[..., None] * 31 + ar * 17) % 257).float() / 257.0)


class RecordingLLM:
    def __init__(self, d_llm):
        self.d_llm = d_llm
        self.seen = None

    def embed_tokens(self, ids):
        return embed_fn(ids, self.d_llm)

    def __call__(self, **kw):
        self.seen = kw
        hs = kw["inputs_embeds"]
        o = types.SimpleNamespace(past_key_values=None, hidden_states=None, attentions=None)
        return _Out(hs, o)


class _Out:
    def __init__(self, hs, ns):
        self._hs = hs
        self.past_key_values = None
        self.hidden_states = None
        self.attentions = None

    def __getitem__(self, i):
        assert i == 0
        return self._hs


def gen_splice_case(name, ev, kind, n_visual, grid, n_pre, n_post, k, seed, d_llm=32):
    vis_id = IMAGE_TOKEN if kind == "image" else VIDEO_TOKEN
    ids = torch.from_numpy(oin.make_prompt(n_visual, n_pre, n_post, vis_id, seed))
    rng = np.random.default_rng(seed + 1)
    all_idx = np.sort(rng.choice(n_visual, size=k, replace=False)).astype(np.int64)
    vis_embeds = rng.standard_normal((k, d_llm), dtype=np.float32)

    class Visual:
        dtype = torch.float32

        def __call__(self, pixels, grid_thw=None):
            return torch.from_numpy(vis_embeds), torch.from_numpy(all_idx), n_visual

    cfg = types.SimpleNamespace(
        output_attentions=False, output_hidden_states=False, use_return_dict=True,
        image_token_id=IMAGE_TOKEN, video_token_id=VIDEO_TOKEN, vision_start_token_id=VSTART, vocab_size=8,
        vision_config=types.SimpleNamespace(spatial_merge_size=2, tokens_per_second=2))
    llm = RecordingLLM(d_llm)
    stub = types.SimpleNamespace(config=cfg, model=llm, visual=Visual(), rope_deltas=None,
                                 lm_head=lambda x: x[..., :8], base_model=types.SimpleNamespace(layers=[]))
    base = ev.Qwen2_5_VLForConditionalGeneration_Selector.__mro__[1]
    stub.get_rope_index = types.MethodType(base.get_rope_index, stub)
    grid_t = torch.tensor([grid])
    kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), cache_position=torch.arange(ids.shape[1]))
    if kind == "image":
        kw.update(pixel_values=torch.zeros(1, 1), image_grid_thw=grid_t)
    else:
        kw.update(pixel_values_videos=torch.zeros(1, 1), video_grid_thw=grid_t,
                  second_per_grid_ts=torch.tensor([1.0]))
    with torch.no_grad():
        ev.Qwen2_5_VLForConditionalGeneration_Selector.forward(stub, **kw)
    seen = llm.seen
    out = dict(kind=kind, n_visual=n_visual, grid=np.array(grid), n_pre=n_pre, n_post=n_post, k=k, seed=seed,
               d_llm=d_llm, all_idx=all_idx, vis_embeds=vis_embeds,
               position_ids_full=stub.get_rope_index(
                   ids, grid_t if kind == "image" else None, grid_t if kind == "video" else None,
                   torch.tensor([1.0]) if kind == "video" else None, torch.ones_like(ids))[0].numpy(),
               position_ids=seen["position_ids"].numpy(), attention_mask=seen["attention_mask"].numpy(),
               inputs_embeds=seen["inputs_embeds"].numpy(), rope_deltas=stub.rope_deltas.numpy())
    np.savez_compressed(os.path.join(HERE, f"splice_{name}.npz"), **out)
    print(f"splice {name}: L={ids.shape[1]} -> L'={seen['inputs_embeds'].shape[1]}")


def gen_attention_case(name, lens, hq, hkv, d, causal, seed):
    """Outputs and q/k/v gradients of the reference's eager attention module on seeded inputs, sequence by sequence."""
    sys.path.insert(0, f"{REF}/qwen-evaluation")
    from qwen25vl import modeling_qwen2_5_vl as m
    sys.path.remove(f"{REF}/qwen-evaluation")
    q, k, v, dout = oin.make_attention_inputs(lens, hq, hkv, d, seed)
    att = object.__new__(m.Qwen2_5_VLAttention)                  # no config / rotary construction (4.50-era API)
    torch.nn.Module.__init__(att)
    width = (hq + 2 * hkv) * d
    att.hidden_size, att.num_heads, att.head_dim = width, hq, d
    att.num_key_value_heads, att.num_key_value_groups = hkv, hq // hkv
    att.is_causal, att.attention_dropout, att.layer_idx = True, 0.0, 0
    att.rope_scaling = {"mrope_section": [d // 2 - 2 * (d // 6), d // 6, d // 6]}
    eye = torch.eye(width)
    att.q_proj = torch.nn.Linear(width, hq * d)
    att.k_proj = torch.nn.Linear(width, hkv * d)
    att.v_proj = torch.nn.Linear(width, hkv * d)
    att.o_proj = torch.nn.Linear(hq * d, hq * d, bias=False)
    with torch.no_grad():
        att.q_proj.weight.copy_(eye[: hq * d]); att.q_proj.bias.zero_()
        att.k_proj.weight.copy_(eye[hq * d: (hq + hkv) * d]); att.k_proj.bias.zero_()
        att.v_proj.weight.copy_(eye[(hq + hkv) * d:]); att.v_proj.bias.zero_()
        att.o_proj.weight.copy_(torch.eye(hq * d))
    att.eval()
    outs, dqs, dks, dvs = [], [], [], []
    a = 0
    for n in lens:
        x = torch.cat([torch.from_numpy(t_[a:a + n]).reshape(n, -1) for t_ in (q, k, v)], dim=1)[None].clone().requires_grad_(True)
        cos = torch.ones(3, 1, n, d)
        sin = torch.zeros(3, 1, n, d)
        mask = None
        if causal:
            mask = torch.full((n, n), torch.finfo(torch.float32).min).triu(1)[None, None]
        y = att(x, attention_mask=mask, position_embeddings=(cos, sin))[0]
        y.backward(torch.from_numpy(dout[a:a + n]).reshape(1, n, hq * d))
        g = x.grad[0]
        outs.append(y.detach()[0].reshape(n, hq, d).numpy())
        dqs.append(g[:, : hq * d].reshape(n, hq, d).numpy())
        dks.append(g[:, hq * d: (hq + hkv) * d].reshape(n, hkv, d).numpy())
        dvs.append(g[:, (hq + hkv) * d:].reshape(n, hkv, d).numpy())
        a += n
    np.savez_compressed(os.path.join(HERE, f"attn_eager_{name}.npz"), lens=np.asarray(lens), hq=hq, hkv=hkv, d=d,
                        causal=causal, seed=seed, out=np.concatenate(outs), dq=np.concatenate(dqs), dk=np.concatenate(dks),
                        dv=np.concatenate(dvs))
    print(f"attention {name}: T={sum(lens)} hq={hq} hkv={hkv} d={d} causal={causal}")


# ---------------------------------------------------------------------------------------------------
# LLaVA-OneVision-1.5 goldens (the reference's OV classes run as they are)
# ---------------------------------------------------------------------------------------------------
OV_IMAGE_TOKEN = 151655


def load_ov():
    sys.path.insert(0, f"{REF}/llava-ov-15")
    from compression_method import modeling_selector as ov
    from compression_method.selector_scorer import TransformerScorer as OVScorer
    sys.path.remove(f"{REF}/llava-ov-15")
    return ov, OVScorer


def gen_ov_lis_case(name, ov, OVScorer, d, hd, grids, seed):
    """RiceTransformerPretrainedModel_Selector.forward with identity tower pieces: several images (grid_thw rows) scored
    JOINTLY (the mean runs over all tokens of the call), budgets 0.1 / 0.2 / 0.5."""
    n = int(sum(t * h * w for t, h, w in grids))
    case = oin.make_case(d, hd, n, seed)
    scorer = build_scorer(OVScorer, case)
    h = torch.from_numpy(case["h"])
    out = {"d": d, "hd": hd, "n": n, "seed": seed, "grids": np.asarray(grids, np.int64)}

    def stub_tower(width, cls_value, scorer_, budget, seen):
        def merger(x):
            seen.append(x)
            return x
        return types.SimpleNamespace(
            patch_embed=lambda x: x, rot_pos_emb=lambda g: torch.zeros(n, 4), class_embedding=torch.full((width,), cls_value),
            class_pos_emb=torch.zeros(4), pre_layernorm=lambda x: x, blocks=[], gradient_checkpointing=False, training=False,
            merger=merger, importance_scorer=scorer_, budgets=budget)

    # Which row of the tower input reaches the LIS block at each position?  The reference's CLS insertion / removal loops
    # (:131-166) run as they are on a row-id tensor (CLS = -1): with several images the removal reads seg_start+1 of the
    # CLS-extended layout for every segment, so rows of later images are shifted -- that IS the reference's merger input.
    seen = []
    ov.RiceTransformerPretrainedModel_Selector.forward(
        stub_tower(1, -1.0, lambda x: x[..., 0], 0.5, seen), torch.arange(n, dtype=torch.float32)[:, None], torch.tensor(grids))
    rowmap = seen[0][:, 0].to(torch.int64)
    out["lis_rowmap"] = rowmap.numpy()
    h_lis = torch.where((rowmap >= 0)[:, None], h[rowmap.clamp(min=0)], torch.zeros(()))
    for r in oin.BUDGETS:
        seen = []
        stub = stub_tower(d, 0.0, scorer, r, seen)
        with torch.no_grad():
            kept, idx, total = ov.RiceTransformerPretrainedModel_Selector.forward(stub, h, torch.tensor(grids))
        assert torch.equal(seen[0], h_lis)
        assert total == n and torch.equal(kept, h_lis[idx]) and torch.equal(stub.last_selected_indices, idx)
        tag = str(r).replace(".", "p")
        out[f"idx_{tag}"] = idx.numpy().astype(np.int64)
        out[f"ps_{tag}"] = stub.last_combined_scores.numpy().astype(np.float32)
    h = h_lis
    with torch.no_grad():
        out["scores"] = scorer(h[None])[0].numpy().astype(np.float32)
    srt = np.sort(out["scores"])[::-1]
    out["gaps"] = np.array([srt[max(1, int(n * r)) - 1] - srt[max(1, int(n * r))] for r in oin.BUDGETS], np.float32)
    out["score_std"] = np.float32(out["scores"].std())
    np.savez_compressed(os.path.join(HERE, f"ovlis_{name}.npz"), **out)
    print(f"ov lis {name}: N={n} (images {len(grids)}) gaps/std={out['gaps'] / out['score_std']}")


def gen_ov_splice_case(name, ov, n_visual, n_pre, n_post, k, seed, d_llm=32, with_position_ids=False):
    """LLaVAOneVision1_5_Model_Selector.forward on a recording language model: what the LLM is handed after the splice."""
    ids = torch.from_numpy(oin.make_prompt(n_visual, n_pre, n_post, OV_IMAGE_TOKEN, seed))
    rng = np.random.default_rng(seed + 1)
    all_idx = np.sort(rng.choice(n_visual, size=k, replace=False)).astype(np.int64)
    vis_embeds = rng.standard_normal((k, d_llm), dtype=np.float32)
    seen = {}

    def language_model(**kw):
        seen.update(kw)
        return types.SimpleNamespace(last_hidden_state=kw["inputs_embeds"], past_key_values=None, hidden_states=None,
                                     attentions=None)

    stub = types.SimpleNamespace(
        config=types.SimpleNamespace(output_attentions=False, output_hidden_states=False, use_return_dict=True,
                                     image_token_id=OV_IMAGE_TOKEN, video_token_id=151656),
        get_input_embeddings=lambda: (lambda t: embed_fn(t, d_llm)),
        get_image_features=lambda pv, grid: (torch.from_numpy(vis_embeds), torch.from_numpy(all_idx), n_visual),
        language_model=language_model, rope_deltas=None)
    L = ids.shape[1]
    kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=torch.zeros(1, 1),
              image_grid_thw=torch.tensor([[1, 2, n_visual // 2]]), use_cache=False,
              cache_position=torch.arange(L))        # generate() always passes it (the reference indexes it with positions of
                                                     # the ORIGINAL prompt, :312, so its own arange over L' would not do)
    if with_position_ids:
        kw["position_ids"] = (torch.arange(L) + 5)[None]          # a caller-supplied 1-D position row is sliced the same way
    with torch.no_grad():
        _, visual_token_num = ov.LLaVAOneVision1_5_Model_Selector.forward(stub, **kw)
    assert visual_token_num == n_visual
    out = dict(n_visual=n_visual, n_pre=n_pre, n_post=n_post, k=k, seed=seed, d_llm=d_llm, all_idx=all_idx,
               vis_embeds=vis_embeds, with_position_ids=with_position_ids,
               position_ids=seen["position_ids"].numpy(), cache_position=seen["cache_position"].numpy(),
               attention_mask=seen["attention_mask"].numpy(), inputs_embeds=seen["inputs_embeds"].numpy())
    np.savez_compressed(os.path.join(HERE, f"ovsplice_{name}.npz"), **out)
    print(f"ov splice {name}: L={L} -> L'={seen['inputs_embeds'].shape[1]}")


def main():
    if "--ov-only" in sys.argv:
        ov, OVScorer = load_ov()
        gen_ov_lis_case("8x81", ov, OVScorer, 256, 128, [[1, 9, 9]] * 8, 41)
        gen_ov_lis_case("3ragged", ov, OVScorer, 512, 256, [[1, 10, 12], [1, 6, 7], [2, 8, 8]], 42)
        gen_ov_splice_case("a", ov, 64, 7, 12, 12, 51)
        gen_ov_splice_case("b", ov, 290, 20, 33, 58, 52, with_position_ids=True)
        return
    ft, TransformerScorer = load_ft()
    ev = load_ev()
    torch.manual_seed(0)
    if "--bf16-only" in sys.argv:
        for name, d, hd, n, seed in oin.GOLDEN_CASES:
            gen_lis_bf16_case(name, d, hd, n, seed, ft, ev, TransformerScorer)
        return
    for name, d, hd, n, seed in oin.GOLDEN_CASES:
        gen_lis_case(name, d, hd, n, seed, ft, ev, TransformerScorer)
    gen_splice_case("image_a", ev, "image", 64, (1, 16, 16), 7, 12, 12, 21)
    gen_splice_case("image_b", ev, "image", 256, (1, 32, 32), 20, 33, 51, 22)
    gen_splice_case("video_a", ev, "video", 128, (2, 16, 16), 9, 14, 25, 23)
    for case in oin.ATTN_GOLDEN_CASES:
        gen_attention_case(*case)
    print("LLaVA-OV goldens: run again with --ov-only (separate process: the OV tree has its own compression_method package)")


if __name__ == "__main__":
    main()
