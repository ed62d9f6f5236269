"""GPU parity tests of the differentiable top-k and the training LIS block (forward + closed-form backward)
against the golden vectors produced by the reference's autograd and against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import inputs as oin
from oracle import lis as olis

pytestmark = pytest.mark.gpu
CASES = {c[0]: c for c in oin.GOLDEN_CASES}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from visionselector_amd import ops as _ops
    return _ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"lis_{name}.npz"))


def close(a, b, rtol, what=""):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-12)
    err = np.abs(a - b).max()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


# ---------------------------------------------------------------------------------------------------
# differentiable top-k  (TOLERANCE: fp32 bisection; |dps| <= 1e-5, well inside the 1e-3 of north_star)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(CASES))
def test_soft_topk_forward_backward_golden(ops, golden_dir, name):
    g = load(golden_dir, name)
    k = int(g["topk_k"])
    xs = dev(g["scores"])[None]
    ps, ts = ops.soft_topk_fwd(xs, k)
    assert abs(float(ts[0]) - float(g["topk_ts"])) <= 2e-5
    assert np.abs(ps[0].cpu().numpy() - g["topk_ps"]).max() <= 1e-5
    assert abs(float(ps.sum()) - k) <= 1e-2
    gvec = dev(oin.make_vec(int(g["n"]), int(g["seed"]) + 1000))[None]
    grad = ops.soft_topk_bwd(gvec, xs, ts)
    close(grad[0].cpu().numpy(), g["topk_grad"], 2e-5, "TopK.backward")
    for r in oin.BUDGETS:       # the inference path's last_combined_scores (EV :190)
        kk = olis.budget_k_eval(int(g["n"]), r)
        ps2, _ = ops.soft_topk_fwd(xs, kk)
        assert np.abs(ps2[0].cpu().numpy() - g["ps_" + str(r).replace(".", "p")]).max() <= 1e-5


def test_soft_topk_batched_and_edges(ops):
    rng = np.random.default_rng(3)
    xs = rng.standard_normal((7, 5000)).astype(np.float32) * 3
    for k in (1, 17, 4999):
        ps, ts = ops.soft_topk_fwd(dev(xs), k)
        ts_ref, ps_ref = olis.find_ts(xs, k)
        # k = n-1 is ill-conditioned in fp32 (the sum sits at ~5000 with 5e-4 spacing, so the reference's own
        # `sum < k` decisions are rounding noise in the late iterations): looser there
        tol = 2e-5 if k < 100 else 1e-3
        assert np.abs(ps.cpu().numpy() - ps_ref).max() <= tol
        assert np.abs(ts.cpu().numpy() - ts_ref[:, 0]).max() <= (1e-4 if k < 100 else 5e-2)
        assert np.abs(ps.sum(1).cpu().numpy() - k).max() <= 2e-2
    from visionselector_amd._native import VselError
    with pytest.raises(VselError, match="0 < k < n"):      # reference: assert 0 < k < n (selector_model.py:75)
        ops.soft_topk_fwd(dev(xs), 5000)
    with pytest.raises(VselError, match="0 < k < n"):
        ops.soft_topk_fwd(dev(xs), 0)


def test_soft_topk_saturated_and_wide_spread_scores_converge(ops):
    """Rows on which a Newton step is useless far from the root (f' = sum sigmoid (1 - sigmoid) ~ 0): two clusters 60 apart with k
    between their sizes' boundary, scores spread over +-200, a single outlier, and a row that is constant.  The iteration must fall
    back to a real bisection of the reference's bracket and still return the root: |sum(ps) - k| small and ps within the usual gate of
    the reference's 64-step bisection (oracle/lis.py::find_ts)."""
    rng = np.random.default_rng(11)
    n = 2304
    rows = []
    a = np.concatenate((np.full(300, 30.0), np.full(n - 300, -30.0))) + 0.01 * rng.standard_normal(n)      # two saturated clusters
    rows.append(a)
    rows.append(rng.uniform(-200, 200, n))                                                                   # wide spread
    b = 0.05 * rng.standard_normal(n)
    b[7] = 500.0                                                                                             # one huge outlier
    rows.append(b)
    rows.append(np.full(n, 0.25))                                                                            # constant row
    rows.append(np.concatenate((np.full(460, 80.0), np.full(n - 460, -80.0))))                               # k exactly at the cluster edge
    xs = np.stack(rows).astype(np.float32)
    for k in (1, 300, 460, 461, 1152, n - 1):
        ps, ts = ops.soft_topk_fwd(dev(xs), k)
        ps = ps.cpu().numpy()
        assert np.isfinite(ps).all() and np.isfinite(ts.cpu().numpy()).all()
        assert np.abs(ps.sum(1) - k).max() <= 5e-3 * max(1.0, k / 460), (k, ps.sum(1))
        _, ps_ref = olis.find_ts(xs, k)
        # (a plateau of f -- the cluster-edge rows -- leaves t undetermined over an interval: compare ps, not t)
        assert np.abs(ps - ps_ref).max() <= 2e-4, (k, np.abs(ps - ps_ref).max())


# ---------------------------------------------------------------------------------------------------
# training block forward
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "qwen3b_256", "qwen3b_576", "qwen7b_2304", "ov8b_5832"])
@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_train_forward_golden(ops, golden_dir, name, storage):
    g = load(golden_dir, name)
    _, d, hd, n, seed = CASES[name]
    c = oin.make_case(d, hd, n, seed)
    dt = torch.float32 if storage == "f32" else torch.bfloat16
    h, wq, bq, wk, bk = (dev(c[x], dt) for x in ("h", "wq", "bq", "wk", "bk"))
    k = olis.budget_k_train(n, 0.2)
    h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, k)
    assert np.array_equal(y.cpu().numpy(), g["train_y"]), "constraint mask = hard top-k of the same scores"
    assert np.abs(ps.cpu().numpy() - g["train_ps"]).max() <= 1e-5
    assert abs(float(bce[0]) - float(g["train_bce"])) <= 1e-5
    ref_rows = g["train_hnew_rowsum"]
    got = h_new.double().sum(1).cpu().numpy()
    # TOLERANCE: fp32 storage -> fp32 roundoff; bf16 storage -> h_new is rounded to bf16 (2^-9 relative per element)
    tol = 1e-3 if storage == "f32" else 0.35
    assert np.abs(got - ref_rows).max() <= tol
    if storage == "bf16":
        exp = (ps[:, None] * h.float()).bfloat16()
        assert torch.equal(h_new, exp), "h_new = (ps * h).type(bf16) with round-to-nearest-even"
    if "train_hnew" in g.files and storage == "f32":
        assert np.abs(h_new.cpu().numpy() - g["train_hnew"]).max() <= 1e-5


# ---------------------------------------------------------------------------------------------------
# training block backward vs the reference's autograd (golden projections / full tensors)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "qwen3b_256", "qwen3b_576", "qwen7b_2304", "ov8b_5832"])
@pytest.mark.parametrize("fused_bce", [True, False])
def test_train_backward_golden(ops, golden_dir, name, fused_bce):
    g = load(golden_dir, name)
    _, d, hd, n, seed = CASES[name]
    c = oin.make_case(d, hd, n, seed)
    h, wq, bq, wk, bk = (dev(c[x], torch.float32) for x in ("h", "wq", "bq", "wk", "bk"))
    k = olis.budget_k_train(n, 0.2)
    reg_w = float(g["train_reg_w"])
    h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, k)
    gmat = dev(np.random.default_rng(seed + 2000).standard_normal((n, d), dtype=np.float32) / np.float32(d) ** 0.5)
    if fused_bce:
        outs = ops.lis_train_bwd(gmat, h, wq, bq, wk, bk, ps, y, scores, ts, None, reg_w, need_dh=True)
    else:       # drop-in path: BCE computed by torch outside the block, its gradient arrives as d_ps_ext
        p = ps.clone().requires_grad_(True)
        (reg_w * torch.nn.functional.binary_cross_entropy(p, y)).backward()
        outs = ops.lis_train_bwd(gmat, h, wq, bq, wk, bk, ps, y, scores, ts, p.grad.contiguous(), 0.0, need_dh=True)
    dwq, dbq, dwk, dbk, dh = (t.double().cpu().numpy() for t in outs)
    u_d = oin.make_vec(d, seed + 3000).astype(np.float64)
    v_h = oin.make_vec(hd, seed + 3001).astype(np.float64)
    v_n = oin.make_vec(n, seed + 3002).astype(np.float64)
    # TOLERANCE 2e-3 relative to each tensor's max: the reference's own fp32 autograd (N x N matmul backward) carries
    # ~1e-4 relative noise; both sides agree with the fp64 closed form (tests/test_oracle_golden.py) to that level.
    close(dbk, g["dbk"], 2e-3, "dbk")
    close(dwq @ u_d, g["dwq_u"], 2e-3, "dWq u")
    close(v_h @ dwq, g["v_dwq"], 2e-3, "v dWq")
    close(dwk @ u_d, g["dwk_u"], 2e-3, "dWk u")
    close(v_h @ dwk, g["v_dwk"], 2e-3, "v dWk")
    close(dh @ u_d, g["dx_u"], 2e-3, "dx u")
    close(v_n @ dh, g["v_dx"], 2e-3, "v dx")
    if "dwq" in g.files:
        close(dwq, g["dwq"], 2e-3, "dWq")
        close(dwk, g["dwk"], 2e-3, "dWk")
        close(dh, g["dx"], 2e-3, "dx")
    # and against the fp64 oracle (tighter)
    ref = olis.train_backward(c["h"], c["wq"], c["bq"], c["wk"], c["bk"], 0.2, gmat.cpu().numpy(), reg_w)
    close(dwq, ref["dwq"], 3e-4, "dWq vs fp64 oracle")
    close(dwk, ref["dwk"], 3e-4, "dWk vs fp64 oracle")
    close(dbk, ref["dbk"], 3e-4, "dbk vs fp64 oracle")
    close(dh, ref["dx"], 3e-4, "dx vs fp64 oracle")
    assert np.abs(dbq).max() <= 1e-4 * max(np.abs(ref["dwq"]).max(), 1e-12) * d + 1e-6   # dbq = kbar rs sum(g) ~ 0


def test_scores_backward_standalone(ops):
    """Backward of TransformerScorer.forward alone (vsel_lis_scores_bwd) vs the op-by-op fp64 backward."""
    d, hd, n = 512, 256, 300
    c = oin.make_case(d, hd, n, 77)
    gvec = oin.make_vec(n, 78)
    ref = olis.lis_backward_explicit(c["h"], c["wq"], c["bq"], c["wk"], c["bk"], gvec)
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-5)):
        h, wq, bq, wk, bk = (dev(c[x], dt) for x in ("h", "wq", "bq", "wk", "bk"))
        dwq, dbq, dwk, dbk, dh = ops.lis_scores_bwd(dev(gvec), h, wq, bq, wk, bk, need_dh=True)
        close(dwq.cpu().numpy(), ref["dwq"], tol, "dWq")
        close(dwk.cpu().numpy(), ref["dwk"], tol, "dWk")
        close(dbq.cpu().numpy(), ref["dbq"], tol, "dbq")
        close(dbk.cpu().numpy(), ref["dbk"], tol, "dbk")
        close(dh.float().cpu().numpy(), ref["dx"], tol if dt == torch.float32 else 1e-2, "dx")


def test_train_bf16_end_to_end_tolerance(ops):
    """bf16 storage (what training runs in): soft mask within 1e-3 of the fp32 oracle (north_star tolerance)."""
    d, hd, n = 3584, 1792, 1024          # config 3: 16 images x 64 tokens packed
    c = oin.make_case(d, hd, n, 41)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    k = olis.budget_k_train(n, 0.2)
    h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, k)
    hn_ref, ps_ref, y_ref, s_ref, ts_ref = olis.train_forward(c["h"], c["wq"], c["bq"], c["wk"], c["bk"], 0.2)
    assert np.abs(ps.cpu().numpy() - ps_ref).max() <= 1e-3
    assert np.array_equal(y.cpu().numpy(), y_ref)
    assert abs(float(bce[0]) - float(olis.bce_mean(ps_ref, y_ref))) <= 1e-4
    assert np.abs(h_new.float().cpu().numpy() - hn_ref).max() <= 1e-3 * max(1.0, np.abs(hn_ref).max()) * 8


def test_train_bwd_factors_rebuild_the_dense_gradients(ops):
    """vsel_lis_train_bwd_factors: dWq = a (x) gx, dWk = dk (x) xsum, same dbq / dbk / dh as the dense backward; through
    ddp.LisFactorSync (world 1) the rebuilt gradients equal the dense ones to fp32 rounding."""
    from visionselector_amd.ddp import LisFactorSync
    d, hd, n, k = 2048, 1024, 600, 120
    c = oin.make_case(d, hd, n, 91)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    g = torch.Generator(device="cuda").manual_seed(5)
    dhn = (torch.randn(n, d, device="cuda", generator=g) / d ** 0.5).bfloat16()
    h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, k)
    dwq, dbq, dwk, dbk, dh = ops.lis_train_bwd(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, None, 0.7, need_dh=True)
    payload, dh2 = ops.lis_train_bwd_factors(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, None, 0.7, need_dh=True)
    a, gx, dk, xs, dbq2, dbk2 = ops.factor_payload_split(payload, hd, d)
    assert torch.equal(dbq2, dbq) and torch.equal(dbk2, dbk) and torch.equal(dh2, dh)
    assert torch.equal(torch.outer(a, gx), dwq) and torch.equal(torch.outer(dk, xs), dwk)       # the dense kernel writes a_r * b_c
    params = [torch.nn.Parameter(t.float()) for t in (wq, bq, wk, bk)]
    sync = LisFactorSync(params)
    for _ in range(2):                                      # two identical micro-batches: the gradients add
        ops.lis_train_bwd_factors(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, None, 0.7, out=sync.new_row(h.device))
    sync.sync()
    for p, ref in zip(params, (dwq, dbq, dwk, dbk)):
        assert float((p.grad - 2 * ref).abs().max()) <= 2e-6 * max(1e-30, float(ref.abs().max())) + 1e-12


@pytest.mark.parametrize("shape", [(3584, 1792, 2304), (2048, 1024, 600), (4096, 2048, 4000), (2048, 1024, 37)])
def test_train_backward_four_launch_form_is_bit_identical(ops, shape):
    """The fused training backward (knob train_fused: soft top-k backward in the prologue of the weighted column sums, both projections in
    one launch, the finish arithmetic inside the launch of both rank-1 writes -- a finish kernel of its own in the factor form) against the
    ten-launch chain: every output bit for bit, dense and rank-1-factor forms, with and without the token gradient and the external BCE
    gradient."""
    from visionselector_amd import _native as N
    d, hd, n = shape
    k = max(1, int(n * 0.2))
    c = oin.make_case(d, hd, n, 123)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    g = torch.Generator(device="cuda").manual_seed(11)
    dhn = (torch.randn(n, d, device="cuda", generator=g) / d ** 0.5).bfloat16()
    ext = torch.randn(n, device="cuda", generator=g) * 1e-3
    h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, k)
    for need_dh in (False, True):
        for d_ps_ext, w in ((None, 0.7), (ext, 0.0), (ext, 1.3)):
            outs = {}
            for fused in (1, 0):
                with N.debug_knob("train_fused", fused):
                    N.profile_start()
                    dense = ops.lis_train_bwd(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, d_ps_ext, w, need_dh=need_dh)
                    torch.cuda.synchronize()
                    names = set(N.profile_stop())
                    fac = ops.lis_train_bwd_factors(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, d_ps_ext, w, need_dh=need_dh)
                assert ("outer_finish_pair_kernel" in names) == bool(fused) and ("soft_topk_bwd_kernel" in names) == (not fused), names
                outs[fused] = [t for t in dense if t is not None] + [t for t in (fac if isinstance(fac, (tuple, list)) else [fac]) if t is not None]
            assert len(outs[0]) == len(outs[1])
            for a_, b_ in zip(outs[1], outs[0]):
                assert torch.equal(a_, b_)


# ---------------------------------------------------------------------------------------------------
# against the reference's OWN bf16 backward (tests/golden/lisbf16_*.npz `topk_grad_bf16`, `bwd_*`): the reference trains with
# bf16 modules and tokens (FT/qwenvl/train/train_qwen_selector.py:175-180; TopK.backward FT/compression_method/
# selector_model.py:60-70; block :158-173; BCE :308-313).  TOLERANCE: tests/parity.BF16_BWD_TOL (2 x the observed margins,
# dominated by the reference's own bf16 rounding -- tests/test_oracle_golden.py measures the same distances for the fp64 form).
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(CASES))
def test_train_backward_vs_reference_bf16_run(ops, golden_dir, name):
    import parity
    g = np.load(os.path.join(golden_dir, f"lisbf16_{name}.npz"))
    _, d, hd, n, seed = CASES[name]
    c = oin.make_case(d, hd, n, seed)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    k = int(g["topk_k"])
    # TopK.backward on the reference's bf16 scores with the bf16-rounded seeded g
    xs = dev(g["scores_bf16"])[None]
    _, ts = ops.soft_topk_fwd(xs, k)
    gvec = dev(oin.make_vec(n, seed + 1000)).bfloat16().float()[None]
    topk_grad = ops.soft_topk_bwd(gvec, xs, ts)[0]
    # the training block: bf16 tokens / weights / upstream gradient, BCE fused with the reference's weight
    h_new, ps, y, scores, tts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, olis.budget_k_train(n, 0.2))
    assert abs(float(bce[0]) - float(g["bwd_bce_bf16"])) <= 2.0 ** -8 * max(1.0, float(g["bwd_bce_bf16"])) + 1e-3
    gmat = dev(np.random.default_rng(seed + 2000).standard_normal((n, d), dtype=np.float32) / np.float32(d) ** 0.5).bfloat16()
    dwq, dbq, dwk, dbk, dh = ops.lis_train_bwd(gmat, h, wq, bq, wk, bk, ps, y, scores, tts, None, float(g["bwd_reg_w"]),
                                               need_dh=True)
    got = {"topk_grad": topk_grad, "dwq": dwq, "dbq": dbq, "dwk": dwk, "dbk": dbk, "dx": dh}
    parity.check_bf16_bwd(f"lis_train_bwd[{name}]", {k_: v.double().cpu().numpy() for k_, v in got.items()}, g)
