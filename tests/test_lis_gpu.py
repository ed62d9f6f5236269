"""GPU parity tests of the LIS inference path: HIP kernels (through the C-ABI) vs the numpy oracle
and the golden vectors produced by the reference.  Run on the MI355X box:  pytest -m gpu."""
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
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from visionselector_amd import ops as _ops
    return _ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"lis_{name}.npz"))


def tag(r):
    return "idx_" + str(r).replace(".", "p")


# ---------------------------------------------------------------------------------------------------
# hard top-k on GIVEN scores: bit-exact indices (SURVEY.md section 7 hard part 1, contract (i))
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(CASES))
def test_select_on_reference_scores_is_bit_exact(ops, golden_dir, name):
    g = load(golden_dir, name)
    s = dev(g["scores"])
    n = s.numel()
    for r in oin.BUDGETS:
        k = olis.budget_k_eval(n, r)
        idx, mask = ops.hard_topk(s, k, want_mask=True)
        assert idx.dtype == torch.int64
        assert np.array_equal(idx.cpu().numpy(), g[tag(r)])
        y = np.zeros(n, np.float32)
        y[g[tag(r)]] = 1
        assert np.array_equal(mask.cpu().numpy(), y)


def test_select_ties_nan_and_signed_zero(ops):
    s = dev(np.array([1, 2, 2, 2, 2, 2, 0, 2, 2, 3], np.float32))
    assert ops.hard_topk(s, 5).tolist() == [1, 2, 3, 4, 9]          # lowest index first among equals
    s = dev(np.array([0.0, -0.0, np.nan, -np.inf, np.inf, -1.0], np.float32))
    assert ops.hard_topk(s, 2).tolist() == [2, 4]                   # NaN greatest (torch.topk convention)
    assert ops.hard_topk(s, 4).tolist() == [0, 1, 2, 4]
    assert ops.hard_topk(s, 6).tolist() == [0, 1, 2, 3, 4, 5]       # k == n
    # all-equal scores (the shipped near-zero init in the limit): first k indices
    s = dev(np.full(1000, 0.25, np.float32))
    assert ops.hard_topk(s, 200).tolist() == list(range(200))


@pytest.mark.parametrize("n,k", [(1, 1), (2, 1), (63, 7), (64, 64), (65, 1), (1023, 500), (1024, 1), (1025, 1024),
                                 (4096, 819), (5000, 4999), (40000, 8000)])
def test_select_random_sizes_match_oracle(ops, n, k):
    rng = np.random.default_rng(n * 31 + k)
    s = rng.standard_normal(n).astype(np.float32)
    s[rng.integers(0, n, max(1, n // 10))] = s[0]                   # inject ties
    idx = ops.hard_topk(dev(s), k).cpu().numpy()
    assert np.array_equal(idx, olis.hard_topk_indices(s, k))


def test_select_batched_rows_independent(ops):
    rng = np.random.default_rng(5)
    s = rng.standard_normal((37, 777)).astype(np.float32)
    idx = ops.hard_topk(dev(s), 155).cpu().numpy()
    for b in range(37):
        assert np.array_equal(idx[b], olis.hard_topk_indices(s[b], 155))


# ---------------------------------------------------------------------------------------------------
# end to end: scores computed by the HIP path, indices vs the fp32 reference (contract (ii))
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_lis_select_matches_reference(ops, golden_dir, name, storage):
    g = load(golden_dir, name)
    _, d, hd, n, seed = CASES[name]
    c = oin.make_case(d, hd, n, seed)
    dt = torch.float32 if storage == "f32" else torch.bfloat16     # inputs are bf16-representable: same values
    h, wq, bq, wk, bk = (dev(c[x], dt) for x in ("h", "wq", "bq", "wk", "bk"))
    scale = max(1.0, float(np.abs(g["scores"]).max()))
    for r in oin.BUDGETS:
        k = olis.budget_k_eval(n, r)
        out, idx, scores = ops.lis_select(h, wq, bq, wk, bk, k)
        # TOLERANCE: fp32 accumulation in a different order than the reference's GEMMs -> a few ulp of the scale
        assert np.abs(scores.cpu().numpy() - g["scores"]).max() <= 4e-6 * scale
        assert np.array_equal(idx.cpu().numpy(), g[tag(r)]), "selected indices must be bit-exact (tie-free seeds)"
        assert torch.equal(out, h[idx])
    s2 = ops.lis_scores(h[None], wq, bq, wk, bk)
    assert torch.equal(s2[0], scores)


def test_lis_select_bench_batch_forms_match_reference(ops, golden_dir):
    """The forms bench.py times -- B = 128 images of the 7B geometry: sweep 1 by one wave per (image, column slab), the register radix
    select as its own launch, the gather with non-temporal stores -- against the reference's golden: the golden image sits at three
    places of a batch of random images (images are scored independently, so its scores / indices / rows must be the golden's whatever
    surrounds it), and the profile proves which kernels ran."""
    from visionselector_amd import _native as N
    g = load(golden_dir, "qwen7b_2304")
    _, d, hd, n, seed = CASES["qwen7b_2304"]
    c = oin.make_case(d, hd, n, seed)
    hg, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    b = 128
    gen = torch.Generator(device="cuda").manual_seed(77)
    h = torch.randn(b, n, d, device="cuda", generator=gen).bfloat16()
    where = (0, 63, 127)
    for i in where:
        h[i] = hg
    scale = max(1.0, float(np.abs(g["scores"]).max()))
    k = olis.budget_k_eval(n, 0.2)
    N.profile_start()
    out, idx, scores = ops.lis_select(h, wq, bq, wk, bk, k)
    prof = N.profile_stop()
    assert {"colsum_seg_kernel", "score_kernel", "topk_select_kernel", "gather_rows_kernel"} <= set(prof), prof
    for i in where:
        assert np.abs(scores[i].cpu().numpy() - g["scores"]).max() <= 4e-6 * scale
        assert np.array_equal(idx[i].cpu().numpy(), g[tag(0.2)]), "selected indices must be bit-exact"
        assert torch.equal(out[i], hg[idx[i]])
    assert torch.equal(scores[0], scores[63]) and torch.equal(scores[0], scores[127])


# ---------------------------------------------------------------------------------------------------
# against the reference's OWN bf16 run (contract (iii): bf16 scores within 1e-3, index symmetric difference reported):
# tests/golden/lisbf16_*.npz = the reference modules and tokens in bfloat16, every op rounding to bf16
# (FT/qwenvl/train/train_qwen_selector.py:175-180; EV/token_compression/selector_model.py:182-194).
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(CASES))
def test_lis_select_vs_reference_bf16_run(ops, golden_dir, name):
    import parity
    g = np.load(os.path.join(golden_dir, f"lisbf16_{name}.npz"))
    _, d, hd, n, seed = CASES[name]
    c = oin.make_case(d, hd, n, seed)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    idx_by = {}
    scores = None
    for r in oin.BUDGETS:
        k = olis.budget_k_eval(n, r)
        out, idx, scores = ops.lis_select(h, wq, bq, wk, bk, k)
        idx_by[str(r).replace(".", "p")] = idx.cpu().numpy()
        assert torch.equal(out, h[idx])
    m = parity.check_lis_bf16(f"lis_select[{name}]", scores.cpu().numpy(), idx_by, g)
    # the soft mask the reference publishes as last_combined_scores (EV/...:190), computed by its bf16 _find_ts
    k = int(g["topk_k"])
    ps, ts = ops.soft_topk_fwd(scores[None], k)
    ps = ps[0].cpu().numpy()
    sm = {"max_abs_dps": float(np.abs(ps - g["ps_bf16"]).max()), "abs_dts": abs(float(ts[0]) - float(g["ts_bf16"])),
          "sum_ps": float(ps.sum(dtype=np.float64)), "sum_ps_reference_bf16": float(g["sum_ps_bf16"]), "k": k,
          "max_abs_dps_after_bf16_store": float(np.abs(torch.from_numpy(ps).bfloat16().float().numpy() - g["ps_bf16"]).max())}
    parity.record(f"soft_topk[{name}]", "soft_bf16", sm)
    assert sm["max_abs_dps"] <= parity.BF16_PS_TOL and sm["abs_dts"] <= 2.0 ** -7, sm
    assert abs(sm["sum_ps"] - k) <= 1e-2                     # ours sums to k; the reference's bf16 bisection does not
    # ... and the opt-in bf16-reference mode (vsel_soft_topk_fwd_bf16ref): on the REFERENCE's bf16 scores it reproduces the reference's
    # stalled bisection exactly -- ts bit for bit; ps bit for bit but for elements whose fp32 sigmoid falls within an ulp of a bf16
    # rounding boundary (libm vs device expf): at most 2 of them, one bf16 step apart
    sref = torch.from_numpy(g["scores_bf16"]).cuda()
    psr, tsr = ops.soft_topk_fwd(sref[None], k, bf16_reference=True)
    psr = psr[0].cpu().numpy()
    assert float(tsr[0]) == float(g["ts_bf16"])
    bad = np.flatnonzero(psr != g["ps_bf16"])
    assert len(bad) <= 2 and (len(bad) == 0 or np.abs(psr[bad] - g["ps_bf16"][bad]).max() <= 2.0 ** -8), (len(bad), psr[bad], g["ps_bf16"][bad])
    assert np.array_equal(psr, olis.find_ts_bf16_reference(g["scores_bf16"][None], k)[1][0]) or len(bad) <= 2
    # on OUR scores (fp32 accumulation; within 1e-3 of the reference's) the mode lands on the same bf16 threshold or its neighbour
    pso, tso = ops.soft_topk_fwd(scores[None], k, bf16_reference=True)
    assert abs(float(tso[0]) - float(g["ts_bf16"])) <= 2.0 ** -6
    assert np.abs(pso[0].cpu().numpy() - g["ps_bf16"]).max() <= 2 * parity.BF16_PS_TOL
    # training forward (FT/compression_method/selector_model.py:158-173) against the reference's bf16 training forward
    h_new, tps, y, tsc, tts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, olis.budget_k_train(n, 0.2))
    assert np.abs(tps.cpu().numpy() - g["train_ps_bf16"]).max() <= parity.BF16_PS_TOL
    assert int((y.cpu().numpy() != g["train_y_bf16"]).sum()) <= max(2 * int(g["ties_at_kth_0p2"]), 2, int(0.01 * k))
    assert abs(float(bce[0]) - float(g["train_bce_bf16"])) <= 1e-3
    rs = h_new.double().sum(1).cpu().numpy()
    ref = g["train_hnew_rowsum_bf16"]
    # H' rows: ps * H rounded to bf16 on both sides; ps differs by <= BF16_PS_TOL -> row sums within that x sum |h|
    assert np.abs(rs - ref).max() <= parity.BF16_PS_TOL * float(np.abs(c["h"]).sum(1).max())
    assert m["max_abs_dscore"] <= 1e-3


def test_lis_select_deterministic(ops):
    c = oin.make_case(3584, 1792, 2304, 99)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    a = ops.lis_select(h, wq, bq, wk, bk, 460)
    b = ops.lis_select(h, wq, bq, wk, bk, 460)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("d,hd,n", [(64, 32, 5), (72, 40, 33), (512, 100, 130), (1032, 520, 257), (8192, 64, 70)])
def test_lis_generic_shapes(ops, d, hd, n):
    """Shapes off the fast paths (D not a multiple of 512, odd Hd, tiny N)."""
    c = oin.make_case(d, hd, n, 7)
    ref = olis.scorer_reference(c["h"][None], c["wq"], c["bq"], c["wk"], c["bk"])[0]
    for dt in (torch.float32, torch.bfloat16):
        h, wq, bq, wk, bk = (dev(c[x], dt) for x in ("h", "wq", "bq", "wk", "bk"))
        k = max(1, n // 3)
        out, idx, scores = ops.lis_select(h, wq, bq, wk, bk, k)
        s = scores.cpu().numpy()
        assert np.abs(s - ref).max() <= 4e-6 * max(1.0, np.abs(ref).max())
        assert np.array_equal(idx.cpu().numpy(), olis.hard_topk_indices(s, k))   # select is exact on our own scores
        assert torch.equal(out, h[idx])


def test_lis_batched_uniform(ops):
    """[B,N,D]: each batch item is scored with its own token mean (reference: mean over dim -1 per batch item)."""
    b, n, d, hd = 5, 300, 2048, 1024
    c = oin.make_case(d, hd, n, 21, batch=b)
    ref = olis.scorer_reference(c["h"], c["wq"], c["bq"], c["wk"], c["bk"])
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    out, idx, scores = ops.lis_select(h, wq, bq, wk, bk, 60)
    s = scores.cpu().numpy()
    assert s.shape == (b, n) and np.abs(s - ref).max() <= 4e-6 * max(1.0, np.abs(ref).max())
    for i in range(b):
        assert np.array_equal(idx[i].cpu().numpy(), olis.hard_topk_indices(s[i], 60))
        assert torch.equal(out[i], h[i][idx[i]])


def test_lis_ragged_segments(ops):
    """Mixed-resolution batch (BASELINE config 5, per-segment budgets): every segment equals a separate call."""
    d, hd = 2048, 1024
    lens = [576, 1024, 61, 2304, 1, 900]
    ks = [olis.budget_k_eval(n, 0.2) for n in lens]
    c = oin.make_case(d, hd, sum(lens), 33)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    out, idx, scores = ops.lis_select_varlen(h, lens, ks, wq, bq, wk, bk)
    ro = oo = 0
    for n, k in zip(lens, ks):
        seg = c["h"][ro:ro + n]
        ref = olis.scorer_reference(seg[None], c["wq"], c["bq"], c["wk"], c["bk"])[0]
        s = scores[ro:ro + n].cpu().numpy()
        assert np.abs(s - ref).max() <= 4e-6 * max(1.0, np.abs(ref).max())
        assert np.array_equal(idx[oo:oo + k].cpu().numpy(), olis.hard_topk_indices(s, k))
        assert torch.equal(out[oo:oo + k], h[ro:ro + n][idx[oo:oo + k]])
        ro += n
        oo += k


def test_segment_sum_sweep_is_bit_identical(ops):
    """Many (segment, column slab) pairs: sweep 1 runs as colsum_seg_kernel -- one wave streams a segment slab and adds its 128-row
    chunks up itself, no per-chunk partials (knob lis_seg_sums: from how many pairs; 1 = whenever the batched form runs, 0 = never).
    Scores, indices and rows must equal the per-chunk form's BIT FOR BIT, and each segment those of a separate call (small-batch
    form): ragged lengths incl. 1 / 128 / 129 rows, bf16 and fp32 tokens, a uniform batch, the presummed entry untouched."""
    from visionselector_amd._native import debug_knob
    d, hd = 1024, 512
    rng = np.random.default_rng(5)
    lens = [int(x) for x in rng.integers(100, 700, 40)]
    lens[3], lens[10], lens[20] = 128, 129, 1
    ks = [max(1, n // 5) for n in lens]
    c = oin.make_case(d, hd, sum(lens), 44)
    for dt in (torch.bfloat16, torch.float32):
        h, wq, bq, wk, bk = (dev(c[x], dt) for x in ("h", "wq", "bq", "wk", "bk"))
        res = []
        for mn in (0, 1, 100, 200):          # chunked form; 16- / 8- / 4-byte loads per lane (2 / 4 / 8 slabs of the 1024 columns)
            with debug_knob("lis_seg_sums", mn):
                res.append(ops.lis_select_varlen(h, lens, ks, wq, bq, wk, bk))
        for other in res[1:]:
            for a_, b_ in zip(res[0], other):
                assert torch.equal(a_, b_)
        out, idx, scores = res[1]
        ro = oo = 0
        for j, (n, k) in enumerate(zip(lens, ks)):
            if j in (3, 10, 20) or j < 6:                    # a sample of segments (all the edge lengths) against separate calls
                o1, i1, s1 = ops.lis_select(h[ro:ro + n][None].contiguous(), wq, bq, wk, bk, k)
                assert torch.equal(scores[ro:ro + n], s1[0]) and torch.equal(idx[oo:oo + k], i1[0]) and torch.equal(out[oo:oo + k], o1[0])
            ro += n
            oo += k
    c2 = oin.make_case(d, hd, 24 * 300, 45)
    h, wq, bq, wk, bk = (dev(c2[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    hb = h.view(24, 300, d)
    res = []
    for mn in (0, 1):
        with debug_knob("lis_seg_sums", mn):
            res.append(ops.lis_select(hb, wq, bq, wk, bk, 60) + (ops.lis_scores(hb, wq, bq, wk, bk),))
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)


def test_near_zero_init_edge_case(ops):
    """The shipped init (std 1e-4, zero bias) gives scores ~1e-5: still selects exactly what its own scores say."""
    c = oin.make_case(2048, 1024, 576, 3, near_zero_init=True)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    out, idx, scores = ops.lis_select(h, wq, bq, wk, bk, 115)
    s = scores.cpu().numpy()
    ref = olis.scorer_reference(c["h"][None], c["wq"], c["bq"], c["wk"], c["bk"])[0]
    assert np.abs(s - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-12
    assert np.array_equal(idx.cpu().numpy(), olis.hard_topk_indices(s, 115))


def test_errors_are_loud(ops):
    from visionselector_amd._native import VselError
    c = oin.make_case(64, 32, 16, 1)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    with pytest.raises(VselError, match="k=17"):
        ops.lis_select(h, wq, bq, wk, bk, 17)
    with pytest.raises(VselError):
        ops.lis_select(h, wq, bq, wk, bk, 0)
    with pytest.raises(TypeError):
        ops.lis_select(h.half(), wq, bq, wk, bk, 2)
    with pytest.raises(ValueError):
        ops.lis_select(h[:, :32].contiguous(), wq, bq, wk, bk, 2)


# ---------------------------------------------------------------------------------------------------
# BASELINE.json full size (config 1: 7B geometry, N=2304, 10/20/50 %), size-independent properties
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("budget", [0.1, 0.2, 0.5])
def test_full_size_properties(ops, budget):
    b, n, d, hd = 32, 2304, 3584, 1792
    gen = torch.Generator(device="cuda").manual_seed(1234)
    h = torch.randn(b, n, d, device="cuda", generator=gen).bfloat16()
    wq = (0.02 * torch.randn(hd, d, device="cuda", generator=gen)).bfloat16()
    wk = (0.02 * torch.randn(hd, d, device="cuda", generator=gen)).bfloat16()
    bq = (0.02 * torch.randn(hd, device="cuda", generator=gen)).bfloat16()
    bk = (0.02 * torch.randn(hd, device="cuda", generator=gen)).bfloat16()
    k = olis.budget_k_eval(n, budget)
    out, idx, scores = ops.lis_select(h, wq, bq, wk, bk, k)
    assert out.shape == (b, k, d) and idx.shape == (b, k) and scores.shape == (b, n)
    assert bool((idx[:, 1:] > idx[:, :-1]).all()), "ascending, unique"
    assert bool((idx >= 0).all()) and bool((idx < n).all())
    sel = torch.gather(scores, 1, idx)
    thr = sel.min(dim=1, keepdim=True).values
    assert bool(((scores > thr).sum(1) < k).all()) and bool(((scores >= thr).sum(1) >= k).all())
    assert torch.equal(out, torch.gather(h, 1, idx[:, :, None].expand(-1, -1, d)))
    # linearity of the scorer in the query side: scores are affine in each token given the mean; check
    # against an independent fp64 evaluation of the collapsed form for 2 batch items on the host
    for bi in (0, b - 1):
        ref = olis.scorer_collapsed(h[bi].float().cpu().numpy()[None], wq.float().cpu().numpy(), bq.float().cpu().numpy(),
                                    wk.float().cpu().numpy(), bk.float().cpu().numpy())[0]
        s = scores[bi].cpu().numpy()
        assert np.abs(s - ref).max() <= 4e-6 * max(1.0, np.abs(ref).max())
        ridx = olis.hard_topk_indices(ref.astype(np.float32), k)
        # boundary-tie tolerant: symmetric difference only among scores within fp32 noise of the threshold
        diff = np.setxor1d(ridx, idx[bi].cpu().numpy())
        assert all(abs(ref[j] - np.sort(ref)[::-1][k - 1]) <= 1e-5 for j in diff)


def test_two_half_pipeline_equals_single_stream(ops):
    """The opt-in aux-stream software pipeline (>= 32 segments, knob lis_pipeline / VSEL_PIPELINE=1) must give bit-identical
    results to the single-stream, single-piece order."""
    from visionselector_amd._native import debug_get, debug_knob
    c = oin.make_case(2048, 1024, 300, 5, batch=35)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    lens = [300, 17, 1000, 64, 5, 700, 33, 900, 12] * 4
    ks = [max(1, n // 5) for n in lens]
    c2 = oin.make_case(2048, 1024, sum(lens), 6)
    h2 = dev(c2["h"], torch.bfloat16)
    before = debug_get("lis_pipeline")
    with debug_knob("lis_pipeline", 1):
        a = ops.lis_select(h, wq, bq, wk, bk, 77)
        a2 = ops.lis_select_varlen(h2, lens, ks, wq, bq, wk, bk)
        with debug_knob("lis_pipeline", 0):
            b = ops.lis_select(h, wq, bq, wk, bk, 77)
            b2 = ops.lis_select_varlen(h2, lens, ks, wq, bq, wk, bk)
        torch.cuda.synchronize()
        for x, y in zip(a + a2, b + b2):
            assert torch.equal(x, y)
        # back-to-back calls on the same stream reuse the aux stream / events correctly
        outs = [ops.lis_select(h, wq, bq, wk, bk, 77) for _ in range(20)]
        torch.cuda.synchronize()
    assert debug_get("lis_pipeline") == before             # restored whatever happened inside
    for o in outs:
        for x, y in zip(o, a):
            assert torch.equal(x, y)


def test_lis_select_is_graph_capturable(ops):
    """No hipMalloc / sync inside the C-ABI: the whole path can be captured into a hipGraph and replayed."""
    c = oin.make_case(2048, 1024, 576, 9, batch=3)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    ref = ops.lis_select(h, wq, bq, wk, bk, 115)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.lis_select(h, wq, bq, wk, bk, 115)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        out = ops.lis_select(h, wq, bq, wk, bk, 115)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(ref, out):
        assert torch.equal(a, b)


def test_permuted_select_equals_unreorder_then_select(ops):
    """vsel_lis_select_permuted on the window-ordered tensor == un-reorder gather (EV :179-181) followed by vsel_lis_select,
    bit for bit (scores, indices, kept rows), without materialising the un-reordered copy."""
    b, n, d, hd, k = 6, 1000, 2048, 1024, 200
    c = oin.make_case(d, hd, n, 17, batch=b)
    h_log, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    g = torch.Generator().manual_seed(3)
    # per-image permutation (window_index permutes merged tokens inside each image), global row numbers
    p2l = torch.cat([torch.randperm(n, generator=g) + i * n for i in range(b)]).cuda()      # window_index
    l2p = torch.argsort(p2l)                                                                  # reverse_indices
    h_phys = torch.empty_like(h_log.view(b * n, d))
    h_phys[l2p] = h_log.view(b * n, d)              # so that h_phys[l2p] == h_log  (hidden_states[reverse_indices])
    h_phys = h_phys.view(b, n, d)
    assert torch.equal(h_phys.view(b * n, d)[l2p].view(b, n, d), h_log)
    ref = ops.lis_select(h_log, wq, bq, wk, bk, k)
    got = ops.lis_select_permuted(h_phys, l2p, p2l, wq, bq, wk, bk, k)
    assert torch.equal(got[1], ref[1]) and torch.equal(got[0], ref[0])
    # scores: same per-row dot products; the column mean is summed in a different row order -> fp32 roundoff only
    assert float((got[2] - ref[2]).abs().max()) <= 4e-6 * max(1.0, float(ref[2].abs().max()))


def _small_path(on):
    """with _small_path(True): the small-batch form for up to 8 segments (the library default is 4); False: never."""
    from visionselector_amd._native import debug_knob
    return debug_knob("lis_small_path", 8 if on else 0)


@pytest.mark.parametrize("d,hd,n,k,batches", [(3584, 1792, 2304, 460, (1, 2, 4, 8)),      # Qwen2.5-VL-7B, the reference's eval call
                                                (2048, 1024, 576, 115, (1, 3)),              # 3B
                                                (4096, 2048, 5832, 1166, (1,)),              # LLaVA-OV 8 x 729 jointly
                                                (64, 32, 40, 8, (1, 8)), (512, 112, 130, 1, (2,)), (1040, 528, 9000, 4500, (1,))])
@pytest.mark.parametrize("storage", ["bf16", "f32"])
def test_small_batch_path_is_bit_identical(ops, d, hd, n, k, batches, storage):
    """The five-launch small-batch form (csrc/lis_small.h, <= 8 segments) against the batched nine-launch form on the same
    inputs: scores, indices and kept rows bit for bit; also the scores-only entry."""
    from visionselector_amd import _native
    dt = torch.bfloat16 if storage == "bf16" else torch.float32
    for b in batches:
        c = oin.make_case(d, hd, n, 100 + b, batch=b)
        h = dev(c["h"], dt)
        wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("wq", "bq", "wk", "bk"))
        with _small_path(False):
            ref = ops.lis_select(h, wq, bq, wk, bk, k)
            ref_s = ops.lis_scores(h, wq, bq, wk, bk)
        with _small_path(True):
            _native.profile_start()
            got = ops.lis_select(h, wq, bq, wk, bk, k)
            prof = _native.profile_stop()
            got_s = ops.lis_scores(h, wq, bq, wk, bk)
        assert "proj_nt_small_kernel" in prof and "gemm_nt_bf16x3_kernel" not in prof, prof       # the small form really ran
        assert sum(c_ for _, c_ in prof.values()) == (5 if n <= 8192 else 6), prof
        for x, y in zip(got, ref):
            assert torch.equal(x, y)
        assert torch.equal(got_s, ref_s) and torch.equal(got_s, got[2])


def test_small_batch_path_ragged_permuted_presummed(ops):
    """Ragged segments, the un-reorder-fused form and the producer-supplied column sums through the small-batch kernels."""
    d, hd = 2048, 1024
    lens = [300, 17, 1000, 64, 5]
    ks = [max(1, n // 5) for n in lens]
    c = oin.make_case(d, hd, sum(lens), 77)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    n = 640
    c2 = oin.make_case(d, hd, n, 78)
    h2 = dev(c2["h"], torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    l2p = torch.randperm(n, generator=g).cuda()
    p2l = torch.empty_like(l2p)
    p2l[l2p] = torch.arange(n, device="cuda")
    sums = h2.float().sum(0, keepdim=True).contiguous()
    outs = {}
    for on in (False, True):
        with _small_path(on):
            outs[on] = (ops.lis_select_varlen(h, lens, ks, wq, bq, wk, bk) + ops.lis_select_permuted(h2, l2p, p2l, wq, bq, wk, bk, 128)
                        + ops.lis_select_presummed(h2, sums, wq, bq, wk, bk, 128, logical_to_physical=l2p, physical_to_logical=p2l))
    for x, y in zip(outs[True], outs[False]):
        assert torch.equal(x, y)
    # and the permuted form equals "un-reorder, then select" (kept rows / indices exact)
    out_p, idx_p, _ = outs[True][3:6]
    out_r, idx_r, _ = ops.lis_select(h2[l2p].contiguous(), wq, bq, wk, bk, 128)
    assert torch.equal(idx_p, idx_r) and torch.equal(out_p, out_r)


def test_fused_select_gather_is_bit_identical(ops):
    """Mid-size batches run the radix select inside the gather workgroups (one launch less): same indices / rows / scores as the
    two-launch form, uniform and ragged."""
    from visionselector_amd import _native
    d, hd, n, k, b = 2048, 1024, 1100, 220, 9
    c = oin.make_case(d, hd, n, 31, batch=b)
    h, wq, bq, wk, bk = (dev(c[x], torch.bfloat16) for x in ("h", "wq", "bq", "wk", "bk"))
    lens = [700, 33, 1500, 64, 9, 412, 800]
    ks = [max(1, m // 4) for m in lens]
    c2 = oin.make_case(d, hd, sum(lens), 32)
    h2 = dev(c2["h"], torch.bfloat16)
    outs = {}
    for limit in (0, 64):
        with _native.debug_knob("lis_fused_select", limit):
            _native.profile_start()
            a = ops.lis_select(h, wq, bq, wk, bk, k)
            prof = _native.profile_stop()
            assert ("select_gather_small_kernel" in prof) == (limit > 0), prof
            outs[limit] = a + ops.lis_select_varlen(h2, lens, ks, wq, bq, wk, bk)
    for x, y in zip(outs[0], outs[64]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("d,hd,n,k,b", [(2048, 1024, 576, 115, 56), (3584, 1792, 300, 150, 40), (4096, 2048, 729, 145, 33)])
def test_flat_resident_gather_is_bit_identical(ops, d, hd, n, k, b):
    """Uniform batches of >= 4096 kept rows gather through a flat list of kept rows dealt to resident waves (knob lis_gather = 10 x
    workgroups per CU + rows in flight; round 6) instead of one workgroup per few rows of a segment: the same rows to the same places,
    with and without the producer's row permutation (vsel_lis_select_permuted), whatever the depth."""
    from visionselector_amd import _native
    g = torch.Generator(device="cuda").manual_seed(d + n)
    h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
    wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    perm = torch.cat([torch.randperm(n, device="cuda", generator=g) + i * n for i in range(b)])       # logical -> physical, per image
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(b * n, device="cuda")
    outs = {}
    for form in (0, 82, 43, 24):
        with _native.debug_knob(lis_gather=form, lis_fused_select=0):
            outs[form] = ops.lis_select(h, wq, bq, wk, bk, k) + ops.lis_select_permuted(h, perm, inv, wq, bq, wk, bk, k)
    out0, idx0 = outs[0][0], outs[0][1]
    assert torch.equal(out0, torch.gather(h, 1, idx0[..., None].expand(-1, -1, d)))
    for form in (82, 43, 24):
        for x, y in zip(outs[0], outs[form]):
            assert torch.equal(x, y), form
