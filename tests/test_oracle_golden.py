"""The numpy oracle (oracle/) against vectors produced by the reference itself
(tests/golden/make_golden.py imports /root/reference; only the .npz files are read here)."""
import os

import numpy as np
import pytest

from oracle import inputs as oin
from oracle import lis, splice

CASES = {c[0]: c for c in oin.GOLDEN_CASES}
IMAGE_TOKEN, VIDEO_TOKEN = 151655, 151656


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"lis_{name}.npz"))


@pytest.fixture(scope="module")
def cases():
    cache = {}

    def get(name):
        if name not in cache:
            _, d, hd, n, seed = CASES[name]
            cache[name] = oin.make_case(d, hd, n, seed)
        return cache[name]

    return get


@pytest.mark.parametrize("name", list(CASES))
def test_scores_reference_formulation(golden_dir, cases, name):
    g = load(golden_dir, name)
    c = cases(name)
    s = lis.scorer_reference(c["h"][None], c["wq"], c["bq"], c["wk"], c["bk"])[0]
    # fp32 GEMM summation order differs between OpenBLAS and torch's CPU BLAS: a few ulp of the score scale
    assert np.abs(s - g["scores"]).max() <= 2e-6 * max(1.0, np.abs(g["scores"]).max())


@pytest.mark.parametrize("name", list(CASES))
def test_scores_collapsed_equals_reference(golden_dir, cases, name):
    g = load(golden_dir, name)
    c = cases(name)
    s = lis.scorer_collapsed(c["h"][None], c["wq"], c["bq"], c["wk"], c["bk"])[0]
    assert np.abs(s - g["scores"]).max() <= 2e-6 * max(1.0, np.abs(g["scores"]).max())
    for r in oin.BUDGETS:
        k = lis.budget_k_eval(c["h"].shape[0], r)
        assert np.array_equal(lis.hard_topk_indices(s.astype(np.float32), k), g["idx_" + str(r).replace(".", "p")])


@pytest.mark.parametrize("name", list(CASES))
def test_hard_topk_indices(golden_dir, name):
    g = load(golden_dir, name)
    n = int(g["n"])
    for r in oin.BUDGETS:
        k = lis.budget_k_eval(n, r)
        idx = lis.hard_topk_indices(g["scores"], k)
        ref = g["idx_" + str(r).replace(".", "p")]
        assert idx.dtype == np.int64 and idx.shape == (k,)
        assert np.array_equal(idx, ref)           # bit-exact: golden seeds are tie-free at the boundary
        assert np.all(np.diff(idx) > 0)


def test_budget_truncation():
    # int() of a Python double, SURVEY.md section 7 hard part 3
    assert lis.budget_k_eval(2304, 0.2) == 460
    assert lis.budget_k_eval(576, 0.2) == 115
    assert lis.budget_k_eval(256, 0.2) == 51
    assert lis.budget_k_eval(3, 0.1) == 1 and lis.budget_k_train(3, 0.1) == 0
    assert lis.budget_k_eval(5832, 0.2) == 1166


def test_hard_topk_ties_and_specials():
    s = np.array([1, 2, 2, 2, 2, 2, 0, 2, 2, 3], np.float32)
    assert lis.hard_topk_indices(s, 5).tolist() == [1, 2, 3, 4, 9]     # lowest index wins among the tied 2s
    s = np.array([0.0, -0.0, np.nan, -np.inf, np.inf, -1.0], np.float32)
    assert lis.hard_topk_indices(s, 2).tolist() == [2, 4]              # NaN is greatest (torch.topk convention)
    assert lis.hard_topk_indices(s, 4).tolist() == [0, 1, 2, 4]        # -0.0 == +0.0, index order


@pytest.mark.parametrize("name", list(CASES))
def test_soft_topk_forward_backward(golden_dir, name):
    g = load(golden_dir, name)
    k = int(g["topk_k"])
    ts, ps = lis.find_ts(g["scores"][None], k)
    assert abs(float(ts[0, 0]) - float(g["topk_ts"])) <= 2e-5
    assert np.abs(ps[0] - g["topk_ps"]).max() <= 1e-5
    assert abs(ps.sum() - k) <= 1e-2
    gvec = oin.make_vec(int(g["n"]), int(g["seed"]) + 1000)[None]
    grad = lis.soft_topk_backward(gvec, g["scores"][None], np.float32(g["topk_ts"]))[0]
    assert np.abs(grad - g["topk_grad"]).max() <= 1e-5 * max(1.0, np.abs(g["topk_grad"]).max())
    # inference path also fills last_combined_scores with the soft mask (EV :190), k = max(1, int(N r))
    for r in oin.BUDGETS:
        kk = lis.budget_k_eval(int(g["n"]), r)
        assert np.abs(lis.soft_topk(g["scores"][None], kk)[0] - g["ps_" + str(r).replace(".", "p")]).max() <= 1e-5


@pytest.mark.parametrize("name", ["tiny", "qwen3b_256", "qwen3b_576", "qwen7b_2304"])
def test_train_forward_and_loss(golden_dir, cases, name):
    g = load(golden_dir, name)
    c = cases(name)
    h_new, ps, y, scores, ts = lis.train_forward(c["h"], c["wq"], c["bq"], c["wk"], c["bk"], 0.2)
    assert np.array_equal(y, g["train_y"])
    assert np.abs(ps - g["train_ps"]).max() <= 1e-5
    assert abs(float(lis.bce_mean(ps, y)) - float(g["train_bce"])) <= 1e-5
    assert np.abs(h_new.astype(np.float64).sum(1) - g["train_hnew_rowsum"]).max() <= 1e-3
    if "train_hnew" in g.files:
        assert np.abs(h_new - g["train_hnew"]).max() <= 1e-5


@pytest.mark.parametrize("name", ["tiny", "qwen3b_256", "qwen3b_576"])
def test_train_backward_closed_form(golden_dir, cases, name):
    g = load(golden_dir, name)
    c = cases(name)
    _, d, hd, n, seed = CASES[name]
    gmat = np.random.default_rng(seed + 2000).standard_normal((n, d), dtype=np.float32) / np.float32(d) ** 0.5
    out = lis.train_backward(c["h"], c["wq"], c["bq"], c["wk"], c["bk"], 0.2, gmat, float(g["train_reg_w"]))
    u_d = oin.make_vec(d, seed + 3000).astype(np.float64)
    v_h = oin.make_vec(hd, seed + 3001).astype(np.float64)
    v_n = oin.make_vec(n, seed + 3002).astype(np.float64)

    def close(a, b, rtol=2e-3):
        scale = max(np.abs(b).max(), 1e-12)
        assert np.abs(a - b).max() <= rtol * scale, (np.abs(a - b).max(), scale)

    close(out["dbq"], g["dbq"], rtol=5e-3) if np.abs(g["dbq"]).max() > 1e-6 else None
    close(out["dbk"], g["dbk"])
    close(out["dwq"] @ u_d, g["dwq_u"])
    close(v_h @ out["dwq"], g["v_dwq"])
    close(out["dwk"] @ u_d, g["dwk_u"])
    close(v_h @ out["dwk"], g["v_dwk"])
    close(out["dx"] @ u_d, g["dx_u"])
    close(v_n @ out["dx"], g["v_dx"])
    if "dwq" in g.files:
        close(out["dwq"], g["dwq"])
        close(out["dwk"], g["dwk"])
        close(out["dx"], g["dx"])
    # explicit op-by-op backward agrees with the closed form
    ex = lis.lis_backward_explicit(c["h"], c["wq"], c["bq"], c["wk"], c["bk"], out["dscores"])
    cl = lis.lis_backward_closed(c["h"], c["wq"], c["bq"], c["wk"], c["bk"], out["dscores"])
    for key in ("dwq", "dwk", "dbk", "dx"):
        close(cl[key], ex[key], rtol=1e-9)


def test_curriculum_weight():
    # FT/qwenvl/train/train_qwen_selector.py:66-79 with the script's 0.1 -> 2.0
    assert lis.curriculum_weight(0, 100, 0.1, 2.0) == 0.1
    assert lis.curriculum_weight(50, 100, 0.1, 2.0) == pytest.approx(1.05)
    assert lis.curriculum_weight(100, 100, 0.1, 2.0) == pytest.approx(2.0)
    assert lis.curriculum_weight(150, 100, 0.1, 2.0) == pytest.approx(2.0)
    assert lis.curriculum_weight(5, -1, 0.1, 2.0) == 0.1


def _embed_np(ids, d_llm):
    ar = np.arange(d_llm, dtype=np.int64)
    return (((ids[..., None] * 31 + ar * 17) % 257).astype(np.float32) / np.float32(257.0))


@pytest.mark.parametrize("name", ["image_a", "image_b", "video_a"])
def test_splice(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"splice_{name}.npz"))
    kind = str(g["kind"])
    vis = IMAGE_TOKEN if kind == "image" else VIDEO_TOKEN
    ids = oin.make_prompt(int(g["n_visual"]), int(g["n_pre"]), int(g["n_post"]), vis, int(g["seed"]))
    if kind == "image":
        sel, new_ids = splice.splice_image(ids, vis, g["all_idx"])
    else:
        sel, new_ids, timask = splice.splice_video(ids, vis, g["all_idx"])
        assert timask.sum() == new_ids.shape[1] - int(g["k"])
    emb = splice.splice_embeds(_embed_np(ids, int(g["d_llm"])), new_ids, sel, vis, g["vis_embeds"])
    assert np.array_equal(emb, g["inputs_embeds"])
    pos, am = splice.slice_positions(g["position_ids_full"], np.ones_like(ids), sel)
    assert np.array_equal(pos, g["position_ids"])
    assert np.array_equal(am, g["attention_mask"])
    assert new_ids.shape[1] == ids.shape[1] - int(g["n_visual"]) + int(g["k"])


@pytest.mark.parametrize("name", ["tiny", "qwen3b_256", "qwen7b_2304"])
def test_torch_cpu_restatement(golden_dir, cases, name):
    """oracle/lis_torch.py (bench.py's cpu_baseline leg) reproduces the goldens too."""
    import torch
    from oracle import lis_torch
    g = load(golden_dir, name)
    c = cases(name)
    t = {k: torch.from_numpy(v) for k, v in c.items()}
    for r in oin.BUDGETS:
        h_new, idx, scores = lis_torch.select_forward(t["h"], t["wq"], t["bq"], t["wk"], t["bk"], r)
        assert np.array_equal(idx.numpy(), g["idx_" + str(r).replace(".", "p")])
        assert np.abs(scores.numpy() - g["scores"]).max() <= 2e-6 * max(1.0, np.abs(g["scores"]).max())
        assert torch.equal(h_new, t["h"][idx])


# ---------------------------------------------------------------------------------------------------
# the reference's own bf16 run (lisbf16_*.npz; make_golden.py --bf16-only): what a user of the released bf16
# checkpoints sees.  SURVEY.md section 7 hard parts 1(iii) and 5: "fixtures should pin both".
# ---------------------------------------------------------------------------------------------------
def load_bf16(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"lisbf16_{name}.npz"))


@pytest.mark.parametrize("name", list(CASES))
def test_fp32_oracle_vs_reference_bf16_run(golden_dir, name):
    """How far the reference's bf16 run sits from its own fp32 run (= from the oracle): scores within 1e-3, selected sets
    differ only inside the tie class / rounding band at the k boundary, <= 1 % of k."""
    import parity
    g32, g16 = load(golden_dir, name), load_bf16(golden_dir, name)
    idx = {t: g32[f"idx_{t}"] for t in ("0p1", "0p2", "0p5")}
    m = parity.lis_bf16_metrics(g32["scores"], idx, g16)
    assert m["max_abs_dscore"] <= parity.BF16_SCORE_TOL * max(1.0, m["max_abs_ref"])
    for t in idx:
        assert m[f"symdiff_{t}"] <= max(2 * m[f"ties_at_kth_{t}"], 2, int(parity.BF16_IDX_FRAC * m[f"k_{t}"]))
        assert m[f"symdiff_max_dist_to_kth_{t}"] <= 2 * parity.BF16_SCORE_TOL
    # bisection in bf16 stalls: sum(ps) != k, ts off by up to the bf16 spacing at |t| ~ 1.4
    k = int(g16["topk_k"])
    assert abs(float(g16["sum_ps_bf16"]) - k) <= 0.01 * k + 0.1
    assert abs(float(g16["ts_bf16"]) - float(g32["topk_ts"])) <= 2.0 ** -7
    assert np.abs(g16["ps_bf16"] - g32["topk_ps"]).max() <= parity.BF16_PS_TOL
    ts, ps = lis.find_ts(g16["scores_bf16"][None], k)            # fp32 oracle on the reference's bf16 scores
    assert np.abs(ps[0] - g16["ps_bf16"]).max() <= parity.BF16_PS_TOL
    assert int((g16["train_y_bf16"] != g32["train_y"]).sum()) <= max(2, int(0.01 * k))


@pytest.mark.parametrize("name", ["tiny", "qwen3b_256", "qwen3b_576", "qwen7b_2304", "ov8b_5832"])
def test_find_ts_bf16_reference_reproduces_the_reference_bf16_run_bit_for_bit(golden_dir, name):
    """oracle/lis.py::find_ts_bf16_reference = _find_ts with every operation rounded to bf16 (what the reference's bf16 scorer makes of
    EV/token_compression/selector_model.py:75-89): from the fixture's `scores_bf16` it gives the reference's own `ts_bf16` and `ps_bf16`
    -- the stalled bisection, sum(ps) != k -- BIT FOR BIT, on all five fixtures."""
    g = load_bf16(golden_dir, name)
    k = int(g["topk_k"])
    ts, ps = lis.find_ts_bf16_reference(g["scores_bf16"][None], k)
    assert float(ts[0, 0]) == float(g["ts_bf16"])
    assert np.array_equal(ps[0], g["ps_bf16"])
    assert float(ps[0].sum(dtype=np.float64)) == float(g["sum_ps_bf16"])


@pytest.mark.parametrize("name", ["tiny", "qwen3b_256", "qwen7b_2304"])
def test_torch_cpu_restatement_bf16(golden_dir, cases, name):
    """oracle/lis_torch.py run in bfloat16 (bench.py's `bf16_reference_formulation` CPU leg) is the reference's bf16 run: same
    ATen ops in the same order.  bf16 GEMM blocking may differ between hosts, hence one bf16 ulp instead of bit equality."""
    import torch
    from oracle import lis_torch
    g = load_bf16(golden_dir, name)
    c = cases(name)
    t = {k: torch.from_numpy(v).bfloat16() for k, v in c.items()}
    h_new, idx, scores = lis_torch.select_forward(t["h"], t["wq"], t["bq"], t["wk"], t["bk"], 0.2)
    assert scores.dtype == torch.bfloat16
    s = scores.float().numpy()
    ref = g["scores_bf16"]
    assert np.abs(s - ref).max() <= 2.0 ** -7 * np.abs(ref).max()
    assert (s == ref).mean() >= 0.9
    assert len(set(idx.tolist()) ^ set(g["idx_bf16_0p2"].tolist())) <= max(2, 2 * int(g["ties_at_kth_0p2"]))
    ts, ps = lis_torch.find_ts(torch.from_numpy(ref).bfloat16()[None], int(g["topk_k"]))
    assert ts.dtype == torch.bfloat16 and float(ts) == float(g["ts_bf16"])
    assert np.array_equal(ps[0].float().numpy(), g["ps_bf16"])


def _eager_attention_torch(q, k, v, cu, causal):
    """torch restatement of the eager formula (modeling_qwen2_5_vl.py:777-797) in fp64, differentiable."""
    import torch
    t, hq, d = q.shape
    rep = hq // k.shape[1]
    outs = []
    for a, b in zip(cu[:-1], cu[1:]):
        qq = q[a:b].transpose(0, 1)
        kk = k[a:b].repeat_interleave(rep, dim=1).transpose(0, 1)
        vv = v[a:b].repeat_interleave(rep, dim=1).transpose(0, 1)
        w = qq @ kk.transpose(1, 2) / (d ** 0.5)
        if causal:
            n = b - a
            w = w.masked_fill(~torch.tril(torch.ones(n, n, dtype=torch.bool)), float("-inf"))
        outs.append((torch.softmax(w, dim=-1) @ vv).transpose(0, 1))
    return torch.cat(outs, 0)


@pytest.mark.parametrize("causal", [True, False])
def test_attention_oracle_forward_backward_match_torch_autograd(causal):
    """Pins oracle.attention.varlen_attention / varlen_attention_backward against torch autograd of the same eager formula
    (flash_attn, which the reference calls, is absent: parity with it is unpinned)."""
    import torch
    from oracle import attention as oattn
    rng = np.random.default_rng(3)
    cu = [0, 37, 37, 90]
    q = rng.standard_normal((90, 4, 16))
    k = rng.standard_normal((90, 2, 16))
    v = rng.standard_normal((90, 2, 16))
    do = rng.standard_normal((90, 4, 16))
    tq, tk, tv = [torch.tensor(x, requires_grad=True) for x in (q, k, v)]
    out = _eager_attention_torch(tq, tk, tv, cu, causal)
    out.backward(torch.tensor(do))
    assert np.allclose(oattn.varlen_attention(q, k, v, np.asarray(cu), causal=causal), out.detach().numpy(), atol=1e-12)
    dq, dk, dv = oattn.varlen_attention_backward(q, k, v, np.asarray(cu), do, causal=causal)
    assert np.allclose(dq, tq.grad.numpy(), atol=1e-11)
    assert np.allclose(dk, tk.grad.numpy(), atol=1e-11)
    assert np.allclose(dv, tv.grad.numpy(), atol=1e-11)


@pytest.mark.parametrize("name", [c[0] for c in oin.ATTN_GOLDEN_CASES])
def test_attention_oracle_matches_reference_eager_module(golden_dir, name):
    """oracle.attention (forward and closed-form backward) against outputs / autograd gradients of the REFERENCE's own eager
    attention module (qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py Qwen2_5_VLAttention.forward, run by
    tests/golden/make_golden.py).  fp32 reference vs fp64 oracle."""
    from oracle import attention as oattn
    g = np.load(os.path.join(golden_dir, f"attn_eager_{name}.npz"))
    lens, hq, hkv, d = [int(x) for x in g["lens"]], int(g["hq"]), int(g["hkv"]), int(g["d"])
    causal = bool(g["causal"])
    q, k, v, dout = oin.make_attention_inputs(lens, hq, hkv, d, int(g["seed"]))
    cu = np.concatenate(([0], np.cumsum(lens)))
    out = oattn.varlen_attention(q, k, v, cu, causal=causal)
    dq, dk, dv = oattn.varlen_attention_backward(q, k, v, cu, dout, causal=causal)
    for got, want in ((out, g["out"]), (dq, g["dq"]), (dk, g["dk"]), (dv, g["dv"])):
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


# ---- LLaVA-OneVision-1.5 goldens (produced by the reference's OWN OV classes, make_golden.py --ov-only) ----------------
OV_LIS = ["8x81", "3ragged"]
OV_SPLICE = ["a", "b"]


def ov_lis_input(g):
    """Tokens that reach the LIS block of RiceTransformerPretrainedModel_Selector.forward for the seeded tower input: the
    reference's CLS insertion / removal loops shift the rows of every image after the first (lis_rowmap, -1 = the zero CLS)."""
    c = oin.make_case(int(g["d"]), int(g["hd"]), int(g["n"]), int(g["seed"]))
    m = g["lis_rowmap"]
    h = np.where((m >= 0)[:, None], c["h"][np.clip(m, 0, None)], np.float32(0))
    return h.astype(np.float32), c


@pytest.mark.parametrize("name", OV_LIS)
def test_ov_tower_lis_block(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"ovlis_{name}.npz"))
    h, c = ov_lis_input(g)
    s = lis.scorer_reference(h[None], c["wq"], c["bq"], c["wk"], c["bk"])[0]
    assert np.abs(s - g["scores"]).max() <= 2e-6 * max(1.0, np.abs(g["scores"]).max())
    n = int(g["n"])
    for r in oin.BUDGETS:                                  # several images, ONE joint selection (modeling_selector.py:173-180)
        tag = str(r).replace(".", "p")
        k = lis.budget_k_eval(n, r)
        assert np.array_equal(lis.hard_topk_indices(g["scores"], k), g[f"idx_{tag}"])
        _, ps = lis.find_ts(g["scores"][None], k)
        assert np.abs(ps[0] - g[f"ps_{tag}"]).max() <= 1e-5


@pytest.mark.parametrize("name", OV_SPLICE)
def test_ov_model_splice(golden_dir, name):
    """LLaVAOneVision1_5_Model_Selector.forward (modeling_selector.py:259-276, :308-314): 1-D position_ids, cache_position and
    attention_mask are index-selected with the kept positions."""
    g = np.load(os.path.join(golden_dir, f"ovsplice_{name}.npz"))
    ids = oin.make_prompt(int(g["n_visual"]), int(g["n_pre"]), int(g["n_post"]), IMAGE_TOKEN, int(g["seed"]))
    L = ids.shape[1]
    sel, new_ids = splice.splice_image(ids, IMAGE_TOKEN, g["all_idx"])
    d = int(g["d_llm"])
    emb = (((ids[..., None] * 31 + np.arange(d) * 17) % 257).astype(np.float32) / 257.0)
    out = splice.splice_embeds(emb, new_ids, sel, IMAGE_TOKEN, g["vis_embeds"])
    assert np.array_equal(out, g["inputs_embeds"])
    pos_in = (np.arange(L) + 5)[None] if bool(g["with_position_ids"]) else np.arange(L)[None]
    pos, am = splice.slice_positions(pos_in, np.ones((1, L), np.int64), sel)
    assert np.array_equal(pos, g["position_ids"]) and np.array_equal(am, g["attention_mask"])
    assert np.array_equal(np.arange(L)[sel], g["cache_position"])


@pytest.mark.parametrize("name", list(CASES))
def test_fp64_oracle_vs_reference_bf16_backward(golden_dir, cases, name):
    """The reference's bf16 BACKWARD (TopK.backward + the training block's autograd in bfloat16,
    FT/compression_method/selector_model.py:60-70, :158-173, :308-313) against the fp64 closed form on the same
    bf16-representable inputs: how much of the gate tests/parity.BF16_BWD_TOL is the reference's own bf16 rounding."""
    import parity
    g = load_bf16(golden_dir, name)
    c = cases(name)
    n, d = int(g["n"]), int(g["d"])
    k = int(g["topk_k"])
    sc = g["scores_bf16"]
    ts, _ = lis.find_ts(sc[None], k)
    gvec = parity.bf16_round(oin.make_vec(n, int(g["seed"]) + 1000)).astype(np.float32)
    tg = lis.soft_topk_backward(gvec[None], sc[None], np.float32(ts[0, 0]))[0]
    gmat = parity.bf16_round(np.random.default_rng(int(g["seed"]) + 2000).standard_normal((n, d), dtype=np.float32)
                             / np.float32(d) ** 0.5)
    ref = lis.train_backward(c["h"], c["wq"], c["bq"], c["wk"], c["bk"], 0.2, gmat, float(g["bwd_reg_w"]))
    m = parity.bf16_bwd_metrics({"topk_grad": tg, "dwq": ref["dwq"], "dbq": ref["dbq"], "dwk": ref["dwk"], "dbk": ref["dbk"],
                                 "dx": ref["dx"]}, g)
    for key, tol in parity.BF16_BWD_TOL.items():
        if key in m:
            assert m[key] <= tol, (name, key, m)
