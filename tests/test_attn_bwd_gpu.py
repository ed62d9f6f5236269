"""GPU parity of the var-len attention backward (vsel_varlen_attn_bwd) against the closed-form fp64 oracle
(oracle/attention.py::varlen_attention_backward, itself pinned to torch autograd of the reference's eager formula on CPU).
Parity with flash_attn's backward (what the reference calls, trainer.py:101-113) is UNPINNED: flash_attn is absent."""
import numpy as np
import pytest
import torch

import parity
from oracle import attention as oattn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from visionselector_amd import ops as _ops
    return _ops


def _case(lens, hq, hkv, seed):
    rng = np.random.default_rng(seed)
    total = sum(lens)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s, dtype=np.float32)).bfloat16()  # noqa: E731
    q, k, v, do = f(total, hq, 128), f(total, hkv, 128), f(total, hkv, 128), f(total, hq, 128)
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    return q, k, v, do, cu


def _run(ops, lens, hq, hkv, causal, seed):
    q, k, v, do, cu = _case(lens, hq, hkv, seed)
    cu_d = torch.from_numpy(cu).cuda()
    out, lse = ops.varlen_attn_fwd_lse(q.cuda(), k.cuda(), v.cuda(), cu_d, max(lens), causal=causal)
    dq, dk, dv = ops.varlen_attn_bwd(do.cuda(), q.cuda(), k.cuda(), v.cuda(), out, lse, cu_d, max(lens), causal=causal)
    return (q, k, v, do, cu), out, lse, (dq, dk, dv)


def _test_name():
    import os
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]


def _rel(got, ref):
    """max |err| relative to the tensor's max magnitude: outputs are bf16 and P / dS are rounded to bf16 before the
    second contractions, exactly as in the forward kernel and in flash-attn.  Logs the observed margin."""
    got = got.float().cpu().numpy().astype(np.float64)
    parity.record(_test_name(), "attn_grad", parity.grad_metrics(got, ref))
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-3)      # floor: a single-key row has dS == 0 exactly


def _fwd_ok(got, ref):
    """forward gate of tests/parity.py: <= 2 bf16 ulps vs the bf16-rounded reference, mean |err| <= 1e-3."""
    got = got.detach().float().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    parity.check_fwd(_test_name(), got, np.asarray(ref, np.float64))
    return True


@pytest.mark.parametrize("lens,hq,hkv", [([100], 4, 2), ([128], 2, 2), ([129], 4, 1), ([1], 2, 1), ([64, 65, 3, 200], 4, 2),
                                         ([300, 17], 14, 2), ([257], 8, 8)])
@pytest.mark.parametrize("causal", [True, False])
def test_attention_backward_matches_oracle(ops, lens, hq, hkv, causal):
    (q, k, v, do, cu), out, lse, (dq, dk, dv) = _run(ops, lens, hq, hkv, causal, seed=sum(lens) + hq)
    a = [x.float().numpy() for x in (q, k, v)]
    rq, rk, rv = oattn.varlen_attention_backward(*a, cu, do.float().numpy(), causal=causal)
    # tolerance: 2 bf16 ulps of the tensor's magnitude (2^-7) -- bf16 outputs + bf16-rounded P / dS operands
    assert _rel(dq, rq) <= 2 ** -6, _rel(dq, rq)
    assert _rel(dk, rk) <= 2 ** -6, _rel(dk, rk)
    assert _rel(dv, rv) <= 2 ** -6, _rel(dv, rv)


def test_forward_lse_matches_oracle(ops):
    lens = [70, 200, 5]
    (q, k, v, do, cu), out, lse, _ = _run(ops, lens, 4, 2, True, seed=9)
    qf, kf = q.float().numpy().astype(np.float64), k.float().numpy().astype(np.float64)
    ref = np.zeros((sum(lens), 4))
    for a, b in zip(cu[:-1], cu[1:]):
        for h in range(4):
            w = qf[a:b, h] @ kf[a:b, h // 2].T / np.sqrt(128.0)
            w = np.where(np.tril(np.ones((b - a, b - a), bool)), w, -np.inf)
            m = w.max(axis=1)
            ref[a:b, h] = m + np.log(np.exp(w - m[:, None]).sum(axis=1))
    assert np.abs(lse.cpu().numpy() - ref).max() <= 1e-4
    # and the lse-returning forward writes the same output bits as the plain forward
    out2 = ops.varlen_attn(q.cuda(), k.cuda(), v.cuda(), torch.from_numpy(cu).cuda(), max(lens), causal=True)
    assert torch.equal(out, out2)


def test_backward_is_deterministic_and_batch_invariant(ops):
    """No float atomics: reruns are bit-identical, and a sequence's gradients do not depend on what it is packed with."""
    lens = [333, 140, 700]
    (q, k, v, do, cu), out, lse, g1 = _run(ops, lens, 8, 2, True, seed=4)
    _, _, _, g2 = _run(ops, lens, 8, 2, True, seed=4)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    s0, s1 = int(cu[1]), int(cu[2])
    cu1 = torch.tensor([0, s1 - s0], dtype=torch.int32).cuda()
    sl = lambda x: x[s0:s1].contiguous().cuda()  # noqa: E731
    o1, l1 = ops.varlen_attn_fwd_lse(sl(q), sl(k), sl(v), cu1, s1 - s0)
    h1 = ops.varlen_attn_bwd(sl(do), sl(q), sl(k), sl(v), o1, l1, cu1, s1 - s0)
    for a, b in zip(g1, h1):
        assert torch.equal(a[s0:s1], b)


def test_backward_linearity_and_full_size(ops):
    """Full-size (7B geometry, L = 2368) property checks: the backward is linear in dout (up to bf16 rounding), and
    sum(dout * out) == sum(dq * q) == sum(dk * k) for ... no -- softmax invariances: rows of dS sum to zero, hence
    sum_d dq.q == sum_d dk.k per (sequence, group) (both equal sum_ij dS_ij S_ij)."""
    lens = [2368, 524]
    hq, hkv = 28, 4
    (q, k, v, do, cu), out, lse, (dq, dk, dv) = _run(ops, lens, hq, hkv, True, seed=21)
    qd, kd = q.cuda().float(), k.cuda().float()
    for a, b in zip(cu[:-1], cu[1:]):
        lhs = (dq[a:b].float() * qd[a:b]).sum().item()
        rhs = (dk[a:b].float() * kd[a:b]).sum().item()
        scale = max(abs(lhs), (dq[a:b].float() * qd[a:b]).abs().sum().item() * 2 ** -8)
        assert abs(lhs - rhs) <= 2 ** -5 * scale + 1e-3, (lhs, rhs)
    # dV = P^T dO: sum over keys of dv equals sum over queries of dout per group (P rows sum to 1)
    for g in range(hkv):
        lhs = dv[:, g].float().sum(0)
        rhs = do.cuda()[:, g * 7:(g + 1) * 7].float().sum((0, 1))
        assert (lhs - rhs).abs().max().item() <= 2 ** -6 * rhs.abs().max().item() + 0.5
    # spot-check one head group of the short sequence against the oracle
    a, b = int(cu[1]), int(cu[2])
    sl = lambda x: x[a:b].float().numpy()  # noqa: E731
    rq, rk, rv = oattn.varlen_attention_backward(sl(q)[:, :7], sl(k)[:, :1], sl(v)[:, :1], np.array([0, b - a]), sl(do)[:, :7])
    assert _rel(dq[a:b, :7], rq) <= 2 ** -6 and _rel(dk[a:b, :1], rk) <= 2 ** -6 and _rel(dv[a:b, :1], rv) <= 2 ** -6


def test_split_and_grouped_dkdv_agree(ops):
    """The per-q-head split (fp32 partials + ordered group sum) and the in-kernel group loop compute the same sums in the
    same order per head; they may differ only by the fp32 association across heads -> compare within 1 bf16 ulp."""
    from visionselector_amd._native import debug_knob
    with debug_knob("attn_bwd_split", 0):
        _, _, _, g0 = _run(ops, [300, 77, 513], 14, 2, True, seed=8)
    with debug_knob("attn_bwd_split", 1):
        _, _, _, g1 = _run(ops, [300, 77, 513], 14, 2, True, seed=8)
    assert torch.equal(g0[0], g1[0])                                   # dQ does not depend on the mode
    for a, b in zip(g0[1:], g1[1:]):
        assert float((a.float() - b.float()).abs().max()) <= 2 ** -7 * float(a.float().abs().max())


def test_four_and_eight_wave_dkdv_agree(ops):
    """dK / dV with four waves per workgroup (K / V operands in registers, one wave per SIMD) and with eight (two per SIMD: the
    query tile split across wave pairs, K / V fragments from LDS, pair sums handed over through LDS): the same contractions with a
    different fp32 association over the queries -> within 1 bf16 ulp; dQ does not depend on the form; each form is deterministic."""
    from visionselector_amd._native import debug_knob
    res = {}
    for w in (4, 8):
        with debug_knob("attn_bwd_waves", w):
            _, _, _, g = _run(ops, [300, 77, 513, 1200], 14, 2, True, seed=9)
            _, _, _, g2 = _run(ops, [300, 77, 513, 1200], 14, 2, True, seed=9)
            with debug_knob("attn_bwd_split", 1):
                _, _, _, gs = _run(ops, [300, 77, 513], 14, 2, True, seed=8)
        for a, b in zip(g, g2):
            assert torch.equal(a, b)
        res[w] = (g, gs)
    assert torch.equal(res[4][0][0], res[8][0][0])
    for a, b in list(zip(res[4][0][1:], res[8][0][1:])) + list(zip(res[4][1][1:], res[8][1][1:])):
        assert float((a.float() - b.float()).abs().max()) <= 2 ** -7 * float(a.float().abs().max())


def test_xcd_local_work_queues_do_not_change_results(ops):
    """Work items on XCD-local queues (a (sequence, kv head) pair's items on one XCD, attn_common.h XcdQueue) or on the single
    heaviest-first queue: placement only -- forward output, log-sum-exp, dQ, dK and dV are bit-identical, for ragged batches whose
    queues empty at different times (stealing) and in both dK / dV forms."""
    from visionselector_amd import _native as N
    lens = [700, 130, 1500, 64, 900, 333, 1100, 257, 640, 1024, 12, 777, 1300]      # 13 x 4 pairs; > 512 forward / dQ items,
    res = {}                                                                         # 13 x 12 x 4 = 624 dK / dV items
    for mode in (0, 1):
        for waves in (8, 4, 64):           # 64: the one-wave-per-SIMD passes (csrc/attn_bwd_dq64.hip, attn_bwd_dkdv64.hip) forced on
            new = int(waves == 64)
            with N.debug_knob(attn_xcd_queue=mode, attn_bwd_waves=min(waves, 8), attn_bwd_split=0, attn_bwd_dq64=new, attn_bwd_dkdv64=new):
                N.profile_start()
                _, out, lse, g = _run(ops, lens, 28, 4, True, seed=21)
                prof = N.profile_stop()
                names = ("attn_bwd_dq64_kernel", "attn_bwd_dkdv64_kernel") if new else ("attn_bwd_dq_kernel", "attn_bwd_dkdv_kernel")
                assert prof[names[0]][1] == 1 and prof[names[1]][1] == 1, prof
                res[(mode, waves)] = (out, lse) + tuple(g)
    for waves in (8, 4, 64):
        for a, b in zip(res[(0, waves)], res[(1, waves)]):
            assert torch.equal(a, b)
    for a, b in zip(res[(0, 4)], res[(0, 64)]):            # and the 64-row passes reproduce the four-wave form bit for bit
        assert torch.equal(a, b)


def test_static_deal_of_work_items_does_not_change_results(ops):
    """Work items handed out by the atomic queue (knob attn_static = 0) or dealt out statically (1: workgroup b takes items b, 2 G - 1 - b,
    2 G + b, ... of the heaviest-first list; attn_common.h static_deal_item): placement only -- forward output, log-sum-exp, dQ, dK, dV
    bit-identical in the 32-rows-per-wave kernels (both dK / dV item forms) and in the 64-rows forms, for ragged batches whose item
    lists hold empty items, and the library's own rule (-1) takes the static deal on a grid with few rounds of items."""
    from visionselector_amd import _native as N
    for lens, hq, hkv in (([524] * 5, 28, 4), ([700, 130, 1500, 64, 900, 333, 1100, 257, 640, 1024, 12, 777, 1300], 28, 4),
                          ([2368], 28, 4), ([1622, 620, 993], 8, 2)):
        res = {}
        for mode in (0, 1, -1):
            for form in ("w4", "w4split", "r64"):
                new = int(form == "r64")
                with N.debug_knob(attn_static=mode, attn_bwd_waves=4, attn_bwd_split=int(form == "w4split"), attn_bwd_dq64=new,
                                  attn_bwd_dkdv64=new, attn_rows64=new):
                    _, out, lse, g = _run(ops, lens, hq, hkv, True, seed=23)
                    res[(mode, form)] = (out, lse) + tuple(g)
        for form in ("w4", "w4split", "r64"):
            for mode in (1, -1):
                for a, b in zip(res[(0, form)], res[(mode, form)]):
                    assert torch.equal(a, b), (lens, form, mode)


def test_queue_skipping_empty_runs_does_not_change_results(ops):
    """Ragged batches on the single work queue: the shared counter jumping over runs of empty items (knob attn_skip_empty = 1, default;
    attn_common.h queue_skip_empty_run) or every item handed out (0): placement only -- forward output, log-sum-exp, dQ, dK, dV
    bit-identical in the 32-rows-per-wave kernels (both dK / dV item forms) and the 64-rows forms, on a bimodal batch (long empty runs),
    on short runs and when the LAST sequences are the short ones (runs that end a level)."""
    from visionselector_amd import _native as N
    for lens, hq, hkv in (([300] * 30 + [1800] * 6, 8, 2), ([1500, 90, 700, 64, 1100, 257, 40, 1024, 12, 777, 1300, 130], 28, 4),
                          ([2100] + [150] * 70, 4, 4)):
        res = {}
        for mode in (0, 1):
            for form in ("w4", "w4split", "r64"):
                new = int(form == "r64")
                with N.debug_knob(attn_skip_empty=mode, attn_static=0, attn_xcd_queue=0, attn_bwd_waves=4, attn_bwd_split=int(form == "w4split"),
                                  attn_bwd_dq64=new, attn_bwd_dkdv64=new, attn_rows64=new):
                    _, out, lse, g = _run(ops, lens, hq, hkv, True, seed=29)
                    res[(mode, form)] = (out, lse) + tuple(g)
        for form in ("w4", "w4split", "r64"):
            for a, b in zip(res[(0, form)], res[(1, form)]):
                assert torch.equal(a, b), (lens[:3], form)
        for a, b in zip(res[(0, "w4")], res[(0, "r64")]):
            assert torch.equal(a, b)


def test_flash_attn_compat_functions(ops):
    """flash_attn_varlen_func / flash_attn_func with the flash-attn call shapes: same packing (differentiable), different
    query / key packings (forward only, bottom-right causal) and the batched dense form."""
    from visionselector_amd.flash_attn_compat import flash_attn_func, flash_attn_varlen_func
    q, k, v, do, cu = _case([150, 64, 9], 4, 2, seed=31)
    cu_d = torch.from_numpy(cu).cuda()
    qd, kd, vd = [t.cuda().requires_grad_(True) for t in (q, k, v)]
    out = flash_attn_varlen_func(qd, kd, vd, cu_d, cu_d, 150, 150, causal=True)
    out.backward(do.cuda())
    a = [x.float().numpy() for x in (q, k, v)]
    ref = oattn.varlen_attention(*a, cu, causal=True)
    rq, rk, rv = oattn.varlen_attention_backward(*a, cu, do.float().numpy(), causal=True)
    assert _fwd_ok(out, ref)
    assert _rel(qd.grad, rq) <= 2 ** -6 and _rel(kd.grad, rk) <= 2 ** -6 and _rel(vd.grad, rv) <= 2 ** -6
    # different packings: 3 sequences with (q, k) lengths (5, 70), (1, 33), (40, 40)
    rng = np.random.default_rng(5)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s, dtype=np.float32)).bfloat16()  # noqa: E731
    q2, k2, v2 = f(46, 4, 128), f(143, 2, 128), f(143, 2, 128)
    cq = torch.tensor([0, 5, 6, 46], dtype=torch.int32).cuda()
    ck = torch.tensor([0, 70, 103, 143], dtype=torch.int32).cuda()
    o2 = flash_attn_varlen_func(q2.cuda(), k2.cuda(), v2.cuda(), cq, ck, 40, 70, causal=True)
    # oracle through the paged form: one page per sequence
    pages = np.zeros((3, 70, 2, 128), np.float32)
    vpages = np.zeros((3, 70, 2, 128), np.float32)
    for s, (a0, b0) in enumerate([(0, 70), (70, 103), (103, 143)]):
        pages[s, : b0 - a0] = k2[a0:b0].float().numpy()
        vpages[s, : b0 - a0] = v2[a0:b0].float().numpy()
    ref2 = oattn.paged_attention(q2.float().numpy(), pages, vpages, cq.cpu().numpy(), np.array([70, 33, 40]),
                                 np.array([[0], [1], [2]]), causal=True)
    assert _fwd_ok(o2, ref2)
    with pytest.raises(RuntimeError, match="no backward"):
        flash_attn_varlen_func(q2.cuda().requires_grad_(True), k2.cuda(), v2.cuda(), cq, ck, 40, 70, causal=True)
    # dense batched form
    qb, kb, vb = f(2, 100, 4, 128), f(2, 100, 2, 128), f(2, 100, 2, 128)
    ob = flash_attn_func(qb.cuda(), kb.cuda(), vb.cuda(), causal=True)
    refb = oattn.varlen_attention(qb.reshape(200, 4, 128).float().numpy(), kb.reshape(200, 2, 128).float().numpy(),
                                  vb.reshape(200, 2, 128).float().numpy(), np.array([0, 100, 200]), causal=True)
    assert ob.shape == (2, 100, 4, 128)
    assert _fwd_ok(ob.reshape(200, 4, 128), refb)


@pytest.mark.parametrize("name", ["d128_causal", "d128_full"])
def test_kernels_match_reference_eager_attention_golden(ops, golden_dir, name):
    """Forward and backward HIP kernels against the REFERENCE's eager attention module (outputs and autograd gradients
    committed in tests/golden/attn_eager_*.npz): bf16-representable inputs, fp32 reference."""
    import os
    from oracle import inputs as oin
    g = np.load(os.path.join(golden_dir, f"attn_eager_{name}.npz"))
    lens, hq, hkv, d = [int(x) for x in g["lens"]], int(g["hq"]), int(g["hkv"]), int(g["d"])
    causal = bool(g["causal"])
    q, k, v, dout = [torch.from_numpy(x).bfloat16().cuda() for x in oin.make_attention_inputs(lens, hq, hkv, d, int(g["seed"]))]
    cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens), causal=causal)
    dq, dk, dv = ops.varlen_attn_bwd(dout, q, k, v, out, lse, cu, max(lens), causal=causal)
    # TOLERANCES as in the oracle-based tests: forward <= 2 bf16 ulps vs the bf16-rounded reference output (tests/parity.py),
    # gradients 2^-6 of the max
    assert _fwd_ok(out, g["out"])
    assert _rel(dq, g["dq"].astype(np.float64)) <= 2 ** -6
    assert _rel(dk, g["dk"].astype(np.float64)) <= 2 ** -6
    assert _rel(dv, g["dv"].astype(np.float64)) <= 2 ** -6


# ---- dQ pass with 64 query rows per wave (csrc/attn_bwd_dq64.hip, knob attn_bwd_dq64) ---------------------------------------------
@pytest.mark.parametrize("lens,hq,hkv", [([1], 2, 1), ([64], 4, 4), ([65], 4, 2), ([256], 4, 4), ([257], 4, 1), ([300, 129, 64], 4, 2),
                                         ([37, 700, 256, 129], 28, 4), ([1230], 32, 8), ([2368], 4, 4)])
@pytest.mark.parametrize("causal", [True, False])
def test_dq64_pass_is_bit_identical_to_the_reference_dq_kernel(lens, hq, hkv, causal):
    """The hand-scheduled 64-rows-per-wave dQ pass performs, per query row, the operations of attn_bwd_dq_kernel in the same order:
    dQ is bit-identical, and so are dK / dV, which consume the D = rowsum(dO o O) and lse * log2(e) it leaves in the workspace.
    (The oracle parity of these gradients is test_backward_matches_oracle's; this is the form-vs-form gate.)  Ragged lengths
    exercise partial query tiles aligned to the END of each sequence, partial key tiles, waves without rows and diagonal masks."""
    import torch
    from visionselector_amd import _native as N, ops
    g = torch.Generator(device="cuda").manual_seed(17 + len(lens))
    T = sum(lens)
    q = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens), causal=causal)
    res = []
    for dq64 in (0, 1):
        with N.debug_knob(attn_bwd_dq64=dq64):
            N.profile_start()
            res.append(ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, max(lens), causal=causal))
            prof = N.profile_stop()
            assert ("attn_bwd_dq64_kernel" in prof) == bool(dq64), prof
    for a, b in zip(*res):
        assert torch.equal(a, b)


# ---- dK / dV pass, one wave per SIMD with the unit pipeline (csrc/attn_bwd_dkdv64.hip, knob attn_bwd_dkdv64) -----------------------
@pytest.mark.parametrize("lens,hq,hkv", [([1], 2, 1), ([64], 4, 4), ([65], 4, 2), ([129], 2, 2), ([193], 4, 1), ([256], 4, 4),
                                         ([300, 129, 64], 4, 2), ([37, 700, 256, 129], 28, 4), ([1230], 32, 8), ([2368], 4, 4)])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("split", [0, 1])
def test_dkdv64_pass_is_bit_identical_to_the_four_wave_dkdv_kernel(lens, hq, hkv, causal, split):
    """The hand-scheduled dK / dV pass keeps the four-wave kernel's work split (wave w owns keys 32 w .. 32 w + 31 of a 128-key item,
    the group's q heads looped inside) and its summation order (query tiles from the sequence's end downward, heads inner): dK and dV are bit-identical to that form's.  (Oracle
    parity of the gradients is test_backward_matches_oracle's; this is the form-vs-form gate.)  Lengths with remainders 1 / 65 / 129
    exercise the clamped tile loads (one valid row), partial key blocks, key blocks above the diagonal and the two-sided mask.
    split = 1: the per-q-head item form (raw fp32 partial rows + attn_bwd_group_sum_kernel) in both kernels."""
    if split and hq == hkv:
        pytest.skip("no group to split")
    import torch
    from visionselector_amd import _native as N, ops
    g = torch.Generator(device="cuda").manual_seed(23 + len(lens))
    T = sum(lens)
    q = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens), causal=causal)
    res = []
    for dkdv64 in (0, 1):
        with N.debug_knob(attn_bwd_dkdv64=dkdv64, attn_bwd_waves=4, attn_bwd_split=split):
            N.profile_start()
            res.append(ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, max(lens), causal=causal))
            prof = N.profile_stop()
            assert ("attn_bwd_dkdv64_kernel" in prof) == bool(dkdv64) and ("attn_bwd_group_sum_kernel" in prof) == bool(split), prof
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("lens,hq,hkv", [([300, 129, 64], 4, 2), ([37, 700, 256, 129], 28, 4), ([1230, 65], 32, 8), ([2368], 8, 4), ([4096, 1], 8, 2)])
def test_dkdv_walk_direction_is_a_rule_of_the_item(lens, hq, hkv):
    """Round 5: dK / dV items of odd kv heads walk their query tiles upward, even ones downward from the sequence's end (knob
    attn_bwd_updown, attn_common.h dkdv_walks_up).  Against the all-downward setting: dQ bit-identical (another pass), dK / dV of EVEN kv heads
    bit-identical, of odd ones within a bf16 rounding; the 4-wave and the 64-row kernel give the same bits under either setting (the 8-wave
    kernel, whose fp32 association differs anyway, within a rounding); a sequence's gradients do not depend on its place in the batch (reversed batch order)."""
    import torch
    from visionselector_amd import _native as N, ops
    g = torch.Generator(device="cuda").manual_seed(57 + len(lens))
    T = sum(lens)
    q = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens))
    res = {}
    for ud in (0, 1):
        for form in ({"attn_bwd_dkdv64": 0, "attn_bwd_waves": 4}, {"attn_bwd_dkdv64": 0, "attn_bwd_waves": 8}, {"attn_bwd_dkdv64": 1}):
            with N.debug_knob(attn_bwd_updown=ud, attn_bwd_split=0, **form):
                got = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, max(lens))
            if ud in res:
                for i, (a, b) in enumerate(zip(got, res[ud])):
                    if form.get("attn_bwd_waves") == 8 and i > 0:     # (the 8-wave kernel adds two half-tile sums per key block: another fp32 association)
                        assert float((a.float() - b.float()).abs().max()) <= 2 ** -7 * float(b.float().abs().max()), (ud, form)
                    else:
                        assert torch.equal(a, b), (ud, form)
            else:
                res[ud] = got
    assert torch.equal(res[0][0], res[1][0])
    for i in (1, 2):
        assert torch.equal(res[0][i][:, 0::2], res[1][i][:, 0::2])
        assert float((res[0][i].float() - res[1][i].float()).abs().max()) <= 2 ** -7 * float(res[0][i].float().abs().max())
        assert not torch.equal(res[0][i][:, 1::2], res[1][i][:, 1::2]) or max(lens) <= 64     # (the odd heads really took the other order)
    # reversed batch order: every sequence's rows come out with the same bits
    order = list(range(len(lens)))[::-1]
    starts = [0] + list(torch.tensor(lens).cumsum(0).tolist())
    rows = torch.cat([torch.arange(starts[i], starts[i + 1]) for i in order]).cuda()
    cu_r = torch.tensor([0] + list(torch.tensor([lens[i] for i in order]).cumsum(0)), dtype=torch.int32, device="cuda")
    got_r = ops.varlen_attn_bwd(do[rows].contiguous(), q[rows].contiguous(), k[rows].contiguous(), v[rows].contiguous(), out[rows].contiguous(),
                                lse[rows].contiguous(), cu_r, max(lens))
    base = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, max(lens))
    for a, b in zip(got_r, base):
        assert torch.equal(a, b[rows])


@pytest.mark.parametrize("lens,hq,hkv", [([1100, 1300], 28, 4), ([1025, 129, 64, 2000], 8, 2), ([1500], 16, 2), ([1100, 700], 32, 8)])
def test_dkdv_group_in_parts_matches_the_formula_and_the_other_item_forms(lens, hq, hkv):
    """dK / dV with a group's q heads cut in 2 / 3 / 4 PARTS (knob attn_bwd_split = k; csrc/attn_bwd_dkdv64.hip: a part = one item that
    loops over its heads and leaves an fp32 partial in the rows of its first head, attn_bwd_group_sum_kernel adds the parts in ascending
    order): against the fp64 eager formula at the backward test's bound, dQ bit-identical to the other forms (the dQ pass does not
    depend on it), dK / dV within a bf16 rounding of the unsplit form, reruns bit-identical, ragged key blocks / rep = 4, 7, 8."""
    import torch
    from visionselector_amd import _native as N, ops
    g = torch.Generator(device="cuda").manual_seed(31 + len(lens))
    T = sum(lens)
    q = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens))
    refs, o = [], 0
    for L in lens:
        sl = slice(o, o + L)
        qq, kk, vv = (t[sl].double().transpose(0, 1).requires_grad_(True) for t in (q, k, v))
        kr, vr = (t.repeat_interleave(hq // hkv, 0) for t in (kk, vv))
        sc = qq @ kr.transpose(1, 2) / 128 ** 0.5
        sc = sc.masked_fill(torch.ones(L, L, device="cuda", dtype=torch.bool).triu(1), float("-inf"))
        (torch.softmax(sc, -1) @ vr).backward(do[sl].double().transpose(0, 1))
        refs.append((sl, kk.grad.transpose(0, 1), vv.grad.transpose(0, 1)))
        o += L
    res = {}
    for split in (0, 1, 2, 3, 4):
        with N.debug_knob(attn_bwd_split=split, attn_bwd_dkdv64=1):
            N.profile_start()
            res[split] = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, max(lens))
            prof = N.profile_stop()
            assert "attn_bwd_dkdv64_kernel" in prof and ("attn_bwd_group_sum_kernel" in prof) == bool(split), prof
            again = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, max(lens))
            for a, b in zip(res[split], again):
                assert torch.equal(a, b)
        assert torch.equal(res[split][0], res[0][0])
        for sl, rk, rv in refs:
            for got, ref in ((res[split][1], rk), (res[split][2], rv)):
                assert ((got[sl].double() - ref).abs().max() / ref.abs().max()).item() <= 2 ** -6
        for i in (1, 2):
            assert float((res[split][i].float() - res[0][i].float()).abs().max()) <= 2 ** -7 * float(res[0][i].float().abs().max())
    if hq // hkv >= 4:          # by itself: these grids have less than a round or two of unsplit items -> per-head or part form (partials + group sum)
        with N.debug_knob(attn_bwd_dkdv64=1):      # (the per-head form below 2048 tokens is otherwise the 8-wave kernel's: another fp32 association)
            N.profile_start()
            auto = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, max(lens))
            prof = N.profile_stop()
        assert "attn_bwd_group_sum_kernel" in prof, prof
        assert any(all(torch.equal(a, b) for a, b in zip(auto, res[sp])) for sp in (1, 2, 3, 4))


@pytest.mark.parametrize("seed", range(10))
def test_scheduling_rules_on_random_batches(seed):
    """The library's own choices (static deal on few rounds of items, the counter jumping over empty runs, the dK / dV item form by item
    count / sequence length / raggedness, 32- vs 64-rows-per-wave kernels) on random packed batches -- 1 .. 40 sequences of 0 .. 2600
    tokens (a zero-length sequence included), GQA 7 / 4 / 8 / 1 -- against the same call with every scheduling rule switched off (atomic
    queue, every item handed out, the group's heads inside the dK / dV item): forward output, log-sum-exp and dQ bit-identical, dK / dV
    within a bf16 rounding, and all gradients against the fp64 eager formula at the backward test's bound."""
    import random
    import torch
    from visionselector_amd import _native as N, ops
    rnd = random.Random(1000 + seed)
    hq, hkv = rnd.choice([(28, 4), (8, 2), (32, 8), (4, 4), (16, 2)])
    n_seq = rnd.choice([1, 2, 3, 5, 8, 13, 24, 40])
    hi = rnd.choice([300, 700, 1300, 2600]) if n_seq <= 13 else rnd.choice([200, 500])
    lens = [rnd.randint(1, hi) for _ in range(n_seq)]
    if n_seq >= 3:
        lens[rnd.randrange(n_seq)] = 0
    if seed % 3 == 0:
        lens[rnd.randrange(n_seq)] = hi                                  # one long sequence among short ones
    if sum(lens) == 0:
        lens[0] = 17
    T = sum(lens)
    g = torch.Generator(device="cuda").manual_seed(seed)
    q = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    L = max(lens)

    def run():
        out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
        return (out, lse) + tuple(ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L))
    auto = run()
    with N.debug_knob(attn_static=0, attn_skip_empty=0, attn_bwd_split=0):
        base = run()
    for i in range(3):
        assert torch.equal(auto[i], base[i]), (lens, hq, hkv, i)
    for i in (3, 4):
        assert float((auto[i].float() - base[i].float()).abs().max()) <= 2 ** -7 * float(base[i].float().abs().max()), (lens, hq, hkv, i)
    assert all(bool(torch.isfinite(t.float()).all()) for t in (auto[0], auto[2], auto[3], auto[4]))
    o = 0
    for Ls in lens:
        sl = slice(o, o + Ls)
        o += Ls
        if Ls == 0:
            continue
        qq, kk, vv = (t[sl].double().transpose(0, 1).requires_grad_(True) for t in (q, k, v))
        kr, vr = (t.repeat_interleave(hq // hkv, 0) for t in (kk, vv))
        sc = (qq @ kr.transpose(1, 2) / 128 ** 0.5).masked_fill(torch.ones(Ls, Ls, device="cuda", dtype=torch.bool).triu(1), float("-inf"))
        (torch.softmax(sc, -1) @ vr).backward(do[sl].double().transpose(0, 1))
        for got, ref in ((auto[2], qq.grad), (auto[3], kk.grad), (auto[4], vv.grad)):
            ref = ref.transpose(0, 1)
            assert ((got[sl].double() - ref).abs().max() / max(ref.abs().max().item(), 1e-3)).item() <= 2 ** -6, (lens, hq, hkv)


def test_long_sequences_take_the_64_row_passes_by_default_and_match_the_oracle_bound():
    """From 1024 tokens in the longest sequence the library picks attn_bwd_dq64_kernel and attn_bwd_dkdv64_kernel (the per-q-head
    split form of the latter from 2048) by itself; gradients against the fp64 eager formula on the bf16-rounded inputs, at the backward test's bound."""
    import torch
    from visionselector_amd import _native as N, ops
    lens, hq, hkv = [2100, 700], 4, 4
    g = torch.Generator(device="cuda").manual_seed(5)
    T = sum(lens)
    q = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0, lens[0], T], dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens))
    N.profile_start()
    dq, dk, dv = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, max(lens))
    prof = N.profile_stop()
    assert "attn_bwd_dq64_kernel" in prof and "attn_bwd_dkdv64_kernel" in prof, prof
    o = 0
    for L in lens:
        sl = slice(o, o + L)
        qq, kk, vv = (t[sl].double().transpose(0, 1).requires_grad_(True) for t in (q, k, v))
        s = qq @ kk.transpose(1, 2) / 128 ** 0.5
        s = s.masked_fill(torch.ones(L, L, device="cuda", dtype=torch.bool).triu(1), float("-inf"))
        (torch.softmax(s, -1) @ vv).backward(do[sl].double().transpose(0, 1))
        for got, ref in ((dq, qq.grad), (dk, kk.grad), (dv, vv.grad)):
            ref = ref.transpose(0, 1)
            assert ((got[sl].double() - ref).abs().max() / ref.abs().max()).item() <= 2 ** -6
        o += L
