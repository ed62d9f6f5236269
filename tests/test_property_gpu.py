"""Property / fuzz tests (hypothesis) of the integer-exact kernels against the oracle: hard top-k select (ties, NaN,
infinities, signed zeros, every k), the device splice (arbitrary visual-token layouts) and the soft top-k invariants."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import lis as olis
from oracle import splice as osplice

pytestmark = pytest.mark.gpu
SETTINGS = dict(max_examples=60, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
IMG = 151655


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from visionselector_amd import ops as _ops
    return _ops


specials = st.sampled_from([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1.0, -1.0, 1e-38, -1e-38, 3.4e38])
score_values = st.one_of(st.floats(width=32, allow_nan=True, allow_infinity=True), specials,
                         st.integers(-3, 3).map(float))            # small integers -> many exact ties


@settings(**SETTINGS)
@given(data=st.data())
def test_hard_topk_matches_oracle_on_arbitrary_floats(ops, data):
    n = data.draw(st.integers(1, 3000))
    vals = data.draw(st.lists(score_values, min_size=n, max_size=n))
    k = data.draw(st.integers(1, n))
    s = np.array(vals, np.float32)
    idx, mask = ops.hard_topk(torch.from_numpy(s).cuda(), k, want_mask=True)
    ref = olis.hard_topk_indices(s, k)
    assert np.array_equal(idx.cpu().numpy(), ref)
    m = np.zeros(n, np.float32)
    m[ref] = 1
    assert np.array_equal(mask.cpu().numpy(), m)


@settings(**SETTINGS)
@given(data=st.data())
def test_hard_topk_batched_rows(ops, data):
    b = data.draw(st.integers(1, 9))
    n = data.draw(st.integers(1, 1500))
    k = data.draw(st.integers(1, n))
    seed = data.draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    s = np.round(rng.standard_normal((b, n)) * 4).astype(np.float32) / 4          # coarse grid: ties everywhere
    idx = ops.hard_topk(torch.from_numpy(s).cuda(), k).cpu().numpy()
    for i in range(b):
        assert np.array_equal(idx[i], olis.hard_topk_indices(s[i], k))


@settings(**SETTINGS)
@given(data=st.data())
def test_splice_arbitrary_layouts(ops, data):
    L = data.draw(st.integers(2, 2500))
    n_vis = data.draw(st.integers(1, L))
    k = data.draw(st.integers(0, n_vis))
    seed = data.draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, 50, L).astype(np.int64)
    ids[np.sort(rng.choice(L, n_vis, replace=False))] = IMG
    idx = np.sort(rng.choice(n_vis, k, replace=False)).astype(np.int64)
    d = 8
    emb = rng.standard_normal((1, L, d), dtype=np.float32)
    ve = rng.standard_normal((k, d), dtype=np.float32)
    pos = rng.integers(0, 1 << 40, (3, 1, L)).astype(np.int64)
    am = rng.integers(0, 2, (1, L)).astype(np.int64)
    sel, new_ids, new_emb, new_pos, new_am = ops.splice(
        torch.from_numpy(ids)[None].cuda(), torch.from_numpy(emb).cuda(), IMG, torch.from_numpy(idx).cuda(),
        torch.from_numpy(ve).cuda(), n_vis, position_ids=torch.from_numpy(pos).cuda(), attention_mask=torch.from_numpy(am).cuda(),
        check=True)
    ref_sel, ref_ids = osplice.splice_image(ids[None], IMG, idx)
    assert np.array_equal(sel.cpu().numpy(), ref_sel) and np.array_equal(new_ids.cpu().numpy(), ref_ids)
    assert np.array_equal(new_emb.cpu().numpy(), osplice.splice_embeds(emb, ref_ids, ref_sel, IMG, ve))
    rp, ra = osplice.slice_positions(pos, am, ref_sel)
    assert np.array_equal(new_pos.cpu().numpy(), rp) and np.array_equal(new_am.cpu().numpy(), ra)


@settings(max_examples=25, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(data=st.data())
def test_soft_topk_invariants(ops, data):
    """sum(ps) = k (to fp32 resolution), monotone in the scores, identical rows give identical masks, and the HIP
    bisection agrees with the oracle's 64-step loop."""
    n = data.draw(st.integers(3, 5000))
    k = data.draw(st.integers(1, n - 1))
    scale = data.draw(st.sampled_from([1e-4, 0.1, 1.0, 8.0]))
    seed = data.draw(st.integers(0, 2 ** 31 - 1))
    x = (np.random.default_rng(seed).standard_normal((2, n)) * scale).astype(np.float32)
    x[1] = x[0]
    ps, ts = ops.soft_topk_fwd(torch.from_numpy(x).cuda(), k)
    ps = ps.cpu().numpy()
    assert np.array_equal(ps[0], ps[1])
    assert abs(ps[0].sum(dtype=np.float64) - k) <= max(2e-2, 3e-5 * n)
    order = np.argsort(x[0], kind="stable")
    assert np.all(np.diff(ps[0][order]) >= -1e-7)
    ts_ref, ps_ref = olis.find_ts(x[:1], k)
    # TOLERANCE: both sides bisect in fp32; when only a handful of tokens sit on one side of the threshold (k or n - k of a
    # few) or the scores are spread over many sigmoids' widths, sum(sigmoid(x + t)) is flat in t at fp32 resolution and the
    # two summation orders settle up to ~1e-4 apart (measured: n=4095, k=4093, scale 8 -> 1.1e-4)
    well_conditioned = min(k, n - k) >= max(2, n // 50) and 0.1 <= scale <= 1.0
    assert np.abs(ps[0] - ps_ref[0]).max() <= (2e-5 if well_conditioned else 2e-3)


# ---------------------------------------------------------------------------------------------------
# producer-side fusion: GELU + column sums, single-sweep LIS
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols,n_seg", [(300, 5120, 1), (1024, 512, 4), (7, 64, 1), (129 * 3, 1280, 3)])
def test_gelu_colsum_matches_torch(rows, cols, n_seg, dt):
    """y is bit-identical to nn.GELU() (erf form) on the same device and the sums are those of the ROUNDED y."""
    from visionselector_amd import ops
    g = torch.Generator(device="cuda").manual_seed(rows + cols)
    x = (2.5 * torch.randn(rows, cols, device="cuda", generator=g)).to(dt)
    y, sums = ops.gelu_colsum(x, n_seg)
    ref = torch.nn.functional.gelu(x)
    assert torch.equal(y, ref)
    want = ref.double().view(n_seg, rows // n_seg, cols).sum(1)
    # TOLERANCE: fp32 accumulation of <= 1024 values of magnitude <= ~10: 1e-5 relative to the column's absolute sum
    scale = ref.double().abs().view(n_seg, rows // n_seg, cols).sum(1)
    assert float(((sums.double() - want).abs() / (scale + 1e-6)).max()) <= 1e-5
    # deterministic
    y2, sums2 = ops.gelu_colsum(x, n_seg)
    assert torch.equal(sums, sums2)
    # the sums-free form of the same kernel (col_sums == NULL; bench.py's baseline for pricing the sums): same y
    y3, none = ops.gelu_colsum(x, n_seg, sums=False)
    assert none is None and torch.equal(y3, ref)


@pytest.mark.parametrize("storage", ["bf16", "f32"])
@pytest.mark.parametrize("name", ["qwen7b_2304", "ov8b_5832", "qwen3b_576"])
def test_presummed_select_matches_reference_goldens(golden_dir, name, storage):
    """vsel_lis_select_presummed (single sweep, column sums supplied by the producer) against the REFERENCE's goldens:
    indices bit-exact for the three budgets, scores to the same bar as the two-sweep path, natural and window order."""
    import os
    from oracle import inputs as oin
    from visionselector_amd import ops
    g = np.load(os.path.join(golden_dir, f"lis_{name}.npz"))
    _, d, hd, n, seed = {c[0]: c for c in oin.GOLDEN_CASES}[name]
    c = oin.make_case(d, hd, n, seed)
    dt = torch.bfloat16 if storage == "bf16" else torch.float32
    h, wq, bq, wk, bk = (torch.from_numpy(c[x]).cuda().to(dt) for x in ("h", "wq", "bq", "wk", "bk"))
    sums = torch.from_numpy(c["h"].astype(np.float64).sum(0).astype(np.float32)).cuda()    # independent of any kernel of ours
    scale = max(1.0, float(np.abs(g["scores"]).max()))
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed))
    h_phys = torch.empty_like(h)
    h_phys[perm] = h
    p2l = torch.empty_like(perm)
    p2l[perm] = torch.arange(n, device="cuda")
    for r in oin.BUDGETS:
        k = olis.budget_k_eval(n, r)
        want = g["idx_" + str(r).replace(".", "p")]
        out, idx, sc = ops.lis_select_presummed(h, sums, wq, bq, wk, bk, k)
        assert np.abs(sc.cpu().numpy() - g["scores"]).max() <= 4e-6 * scale
        assert np.array_equal(idx.cpu().numpy(), want) and torch.equal(out, h[idx])
        out, idx, sc = ops.lis_select_presummed(h_phys, sums, wq, bq, wk, bk, k, logical_to_physical=perm, physical_to_logical=p2l)
        assert np.abs(sc.cpu().numpy() - g["scores"]).max() <= 4e-6 * scale
        assert np.array_equal(idx.cpu().numpy(), want) and torch.equal(out, h[idx])


@pytest.mark.parametrize("s,cin,cout,dt", [(1, 5120, 3584, torch.bfloat16), (3, 5120, 3584, torch.bfloat16), (8, 1280, 2048, torch.bfloat16),
                                           (9, 5120, 3584, torch.bfloat16), (128, 5120, 3584, torch.bfloat16), (40, 512, 96, torch.bfloat16),
                                           (2, 520, 77, torch.float32), (11, 64, 40, torch.float32)])
def test_colsum_linear_matches_fp64(s, cin, cout, dt):
    """vsel_colsum_linear (sum_rows(G) -> sum_rows(H) through the merger's last Linear, stored weight, fp32 accumulate) against
    fp64: both forms (wave per output row up to 8 segments, bf16x3 MFMA beyond), bias scaled by the row count; deterministic."""
    from visionselector_amd import ops
    g = torch.Generator(device="cuda").manual_seed(s * 7 + cin)
    n = 2304
    gs = (torch.randn(s, cin, device="cuda", generator=g) * 40 + 300).float()          # column sums of a GELU output: large, positive-ish
    w = (0.02 * torch.randn(cout, cin, device="cuda", generator=g)).to(dt)
    b = (0.1 * torch.randn(cout, device="cuda", generator=g)).to(dt)
    out = ops.colsum_linear(gs, w, b, n)
    ref = gs.double() @ w.double().t() + n * b.double()
    # TOLERANCE: fp32 accumulation of Cin products: 2e-6 of sum |in . w| + |N b|
    scale = gs.double().abs() @ w.double().abs().t() + n * b.double().abs()
    assert float(((out.double() - ref).abs() / scale).max()) <= 2e-6
    assert torch.equal(out, ops.colsum_linear(gs, w, b, n))
    out0 = ops.colsum_linear(gs, w, None, n)
    assert float(((out0.double() - gs.double() @ w.double().t()).abs() / scale).max()) <= 2e-6


@pytest.mark.parametrize("n,seed", [(2304, 5), (576, 6)])
def test_merger_colsum_path_matches_oracle(n, seed):
    """The whole N2 chain -- vsel_gelu_colsum in the merger, sum_rows(H) by linearity of the merger's last Linear
    (hf_generic.merger_col_sums), vsel_lis_select_presummed -- against the numpy ORACLE evaluated on the tokens the merger
    actually emitted (reference: Qwen2_5_VLPatchMerger, EV/qwen25vl/modeling_qwen2_5_vl.py:148-161, then the LIS block
    EV/token_compression/selector_model.py:182-189).  The linearity form sums H BEFORE its bf16 rounding, so the mean differs
    from the mean of the stored tokens by the averaged rounding noise (~2^-9 |h| / sqrt(N)): scores within 3e-4 * scale (not 4e-6;
    observed 1.5e-4 at N = 576, 0.7e-4 at N = 2304), indices equal
    wherever the oracle's k boundary gap exceeds twice the observed score difference (asserted, so a silent mismatch cannot
    pass)."""
    from visionselector_amd import hf_generic, ops
    g = torch.Generator(device="cuda").manual_seed(seed)
    din, d, hd = 5120, 3584, 1792
    x = (1.5 * torch.randn(n, din, device="cuda", generator=g)).bfloat16()
    last = torch.nn.Linear(din, d).cuda().bfloat16()
    with torch.no_grad():
        last.weight.copy_(0.02 * torch.randn(d, din, device="cuda", generator=g))
        last.bias.copy_(0.1 * torch.randn(d, device="cuda", generator=g))
    wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    y, gsum = ops.gelu_colsum(x, 1)
    assert torch.equal(y, torch.nn.functional.gelu(x))
    with torch.no_grad():
        h = last(y)                                           # the merged tokens the LIS block is handed (bf16)
        col_sums = hf_generic.merger_col_sums(gsum, last, n)
    f = lambda t: t.float().cpu().numpy()  # noqa: E731
    ref = olis.scorer_collapsed(f(h)[None], f(wq), f(bq), f(wk), f(bk))[0]
    scale = max(1.0, float(np.abs(ref).max()))
    srt = np.sort(ref)[::-1]
    for r in (0.1, 0.2, 0.5):
        k = olis.budget_k_eval(n, r)
        out, idx, sc = ops.lis_select_presummed(h, col_sums, wq, bq, wk, bk, k)
        dsc = float(np.abs(sc.cpu().numpy() - ref).max())
        assert dsc <= 3e-4 * scale, dsc
        want = olis.hard_topk_indices(ref.astype(np.float32), k)
        got = idx.cpu().numpy()
        if srt[k - 1] - srt[k] > 2 * dsc:
            assert np.array_equal(got, want)
        else:                                                 # boundary closer than the rounding noise of the mean
            sym = set(got.tolist()) ^ set(want.tolist())
            assert len(sym) <= 2 and all(abs(ref[i] - srt[k - 1]) <= 2 * dsc for i in sym)
        assert torch.equal(out, h[idx])


def test_presummed_select_equals_two_sweep_select():
    """vsel_lis_select_presummed with the true column sums selects the same rows as the two-sweep path (scores agree to
    fp32 rounding of the mean), with and without the row permutation."""
    from visionselector_amd import ops
    g = torch.Generator(device="cuda").manual_seed(77)
    b, n, d, hd, k = 3, 2304, 3584, 1792, 460
    h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
    wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    out0, idx0, sc0 = ops.lis_select(h, wq, bq, wk, bk, k)
    sums = h.double().sum(1).float().contiguous()
    out1, idx1, sc1 = ops.lis_select_presummed(h, sums, wq, bq, wk, bk, k)
    assert torch.equal(idx0, idx1) and torch.equal(out0, out1)
    assert float((sc0 - sc1).abs().max()) <= 1e-6 * max(1.0, float(sc0.abs().max()))
    # permuted form, one segment
    perm = torch.randperm(n, device="cuda", generator=g)
    h_phys = torch.empty_like(h[0])
    h_phys[perm] = h[0]                      # logical row i lives at physical row perm[i]
    p2l = torch.empty_like(perm)
    p2l[perm] = torch.arange(n, device="cuda")
    out2, idx2, sc2 = ops.lis_select_presummed(h_phys, sums[0].contiguous(), wq, bq, wk, bk, k, logical_to_physical=perm,
                                               physical_to_logical=p2l)
    assert torch.equal(idx2, idx0[0]) and torch.equal(out2, out0[0])
    assert float((sc2 - sc0[0]).abs().max()) <= 1e-6 * max(1.0, float(sc0.abs().max()))
    with pytest.raises(ValueError):
        ops.lis_select_presummed(h, sums[:, :8].contiguous(), wq, bq, wk, bk, k)
