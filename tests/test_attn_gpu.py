"""GPU parity tests of the var-len causal GQA attention kernel against the eager-formula oracle (oracle/attention.py,
fp64).  Parity is UNPINNED against flash_attn itself (third party, not in the reference tree) -- see the oracle header."""

import numpy as np
import pytest
import torch

from oracle import attention as oattn
import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from visionselector_amd import ops as _ops
    return _ops


def make_qkv(total, hq, hkv, seed, spike=False, d=128):
    rng = np.random.default_rng(seed)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s, dtype=np.float32)).bfloat16()  # noqa: E731
    q, k, v = f(total, hq, d), f(total, hkv, d), f(total, hkv, d)
    if spike:      # force large running-max jumps late in the sequence (online-softmax rescale path)
        k[total // 2] *= 6
        k[total - 3] *= 9
    return q, k, v


def run_case(ops, lens, hq, hkv, causal, seed, spike=False, d=128):
    total = sum(lens)
    q, k, v = make_qkv(total, hq, hkv, seed, spike, d)
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    out = ops.varlen_attn(q.cuda(), k.cuda(), v.cuda(), torch.from_numpy(cu).cuda(), max(lens), causal=causal)
    ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, causal=causal)
    return out.float().cpu().numpy().astype(np.float64), ref


# TOLERANCE (north_star: "attention outputs within 1e-3 bf16").  Inputs are bf16, P is rounded to bf16 before P.V (as
# flash-attn does) and the output is STORED as bf16, so the comparison is bf16 against bf16: the kernel output vs the fp64
# eager-formula oracle rounded to bf16, in bf16 ulps (tests/parity.py: <= 1 ulp for every element, ulp taken at the
# scale of its output row; <= 8 ulps and >= 93 % of the elements within ONE ulp with the ulp floor at rowmax / 8), plus
# mean |err| <= 1e-3 * max(1, |O|max) against the un-rounded oracle.  The observed maxima of every case are logged
# (gpurun_out/parity/r03_parity.jsonl -> profiles/r03_parity.json).
def _case():
    import os
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]


def check(got, ref, tag=""):
    return parity.check_fwd(_case() + tag, np.asarray(got, dtype=np.float64), ref)


@pytest.mark.parametrize("lens,hq,hkv", [([100], 4, 2), ([128], 2, 2), ([129], 4, 1), ([1], 2, 1), ([64, 65, 3, 200], 4, 2),
                                         ([524], 28, 4), ([37, 300, 5], 8, 8),
                                         ([1230], 32, 8), ([793, 210], 32, 8)])      # LLaVA-OV-1.5 heads (32 q / 8 kv)
@pytest.mark.parametrize("causal", [True, False])
def test_attention_matches_eager_oracle(ops, lens, hq, hkv, causal):
    check(*run_case(ops, lens, hq, hkv, causal, seed=len(lens) * 7 + hq))


def test_attention_rescale_path(ops):
    """Keys with huge logits late in the sequence force the running max to jump (online-softmax rescale)."""
    check(*run_case(ops, [700], 4, 2, True, seed=5, spike=True))
    check(*run_case(ops, [333, 260], 4, 4, False, seed=6, spike=True))


def test_attention_lazy_exponent_ramp(ops):
    """The row's reference exponent moves only when a tile's maximum exceeds it by more than 8 log2 units (csrc/attn.hip kLazyTau):
    keys whose logits creep up tile after tile keep it stale (p up to 2^8) for many tiles, then a jump moves it.  Oracle parity at
    the usual gate, and the 4- / 8-wave forms (different sets of rows per wave) stay bit-identical."""
    from visionselector_amd._native import debug_knob
    lens = [900, 333]
    total = sum(lens)
    q, k, v = make_qkv(total, 4, 2, 23)
    ramp = torch.linspace(0.2, 2.5, total).view(-1, 1, 1)          # logits grow ~12x along the sequence, a little per 64-key tile
    k = (k.float() * ramp).bfloat16()
    k[700] *= 5                                                     # ... and one jump far past the slack
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cu_t = torch.from_numpy(cu).cuda()
    for causal in (True, False):
        ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, causal=causal)
        outs = []
        for nw in (4, 8):
            with debug_knob("attn_split", 0), debug_knob("attn_waves", nw):
                outs.append(ops.varlen_attn(q.cuda(), k.cuda(), v.cuda(), cu_t, max(lens), causal=causal))
        assert torch.equal(outs[0], outs[1])
        check(outs[0].float().cpu().numpy(), ref, f"[causal={causal}]")


def test_tail_first_tiling_is_bit_identical_and_matches_oracle(ops):
    """Query tiles aligned to the end of each sequence (the partial tile is the first, cheapest one under the causal mask) against
    the start-aligned tiling: same key tiles and the same arithmetic per query -> bit-identical outputs and log-sum-exps, for
    ragged packings, both workgroup sizes, the two-stream form and a key sequence longer than the queries (bottom-right mask)."""
    from visionselector_amd._native import debug_knob
    lens = [300, 77, 513, 1200, 1, 129]
    total = sum(lens)
    q, k, v = make_qkv(total, 4, 2, 31, spike=True)
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cu_t = torch.from_numpy(cu).cuda()
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, causal=True)
    for nw, split in ((4, 0), (8, 0), (0, 1)):
        res = []
        for tf in (0, 1):
            with debug_knob("attn_tail_first", tf), debug_knob("attn_waves", nw), debug_knob("attn_split", split):
                o = ops.varlen_attn(qc, kc, vc, cu_t, max(lens), causal=True)
                o2, lse = ops.varlen_attn_fwd_lse(qc, kc, vc, cu_t, max(lens), causal=True)
            res.append((o, o2, lse))
        for a, b in zip(res[0], res[1]):
            assert torch.equal(a, b)
        if split == 0:
            check(res[1][0].float().cpu().numpy(), ref, f"[nw={nw}]")
    # chunked prefill against longer key sequences: queries are the LAST qlen positions of each key sequence
    qlens, klens = [70, 200, 33], [300, 200, 1000]
    qq, _, _ = make_qkv(sum(qlens), 4, 2, 32)
    _, kk, vv = make_qkv(sum(klens), 4, 2, 33)
    cq = torch.from_numpy(np.concatenate(([0], np.cumsum(qlens))).astype(np.int32)).cuda()
    ck = torch.from_numpy(np.concatenate(([0], np.cumsum(klens))).astype(np.int32)).cuda()
    outs = []
    for tf in (0, 1):
        with debug_knob("attn_tail_first", tf), debug_knob("attn_split", 0):
            outs.append(ops.varlen_attn_kv(qq.cuda(), kk.cuda(), vv.cuda(), cq, ck, max(qlens), causal=True))
    assert torch.equal(outs[0], outs[1])
    # paged cache with scattered pages, chunked prefill / plain prefill / decode / a q > k sequence (leading rows see no key)
    rng = np.random.default_rng(77)
    hq, hkv, page = 8, 2, 16
    qlens, klens = [70, 200, 1, 40, 300], [300, 200, 517, 25, 900]
    pages_per = [-(-kl // page) for kl in klens]
    perm = rng.permutation(sum(pages_per) + 3)
    bt = np.zeros((len(qlens), max(pages_per)), np.int32)
    cur = 0
    for i, pp in enumerate(pages_per):
        bt[i, :pp] = perm[cur:cur + pp]
        cur += pp
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh, dtype=np.float32)).bfloat16()  # noqa: E731
    qp, kcache, vcache = f(sum(qlens), hq, 128), f(len(perm), page, hkv, 128), f(len(perm), page, hkv, 128)
    cu_q = np.concatenate(([0], np.cumsum(qlens))).astype(np.int32)
    outs = []
    for tf in (0, 1):
        with debug_knob("attn_tail_first", tf), debug_knob("attn_split", 0), debug_knob("attn_pack", 0):
            outs.append(ops.paged_attn(qp.cuda(), kcache.cuda(), vcache.cuda(), torch.from_numpy(cu_q).cuda(),
                                       torch.tensor(klens, dtype=torch.int32).cuda(), torch.from_numpy(bt).cuda(), max(qlens), causal=True))
    assert torch.equal(outs[0], outs[1])
    refp = oattn.paged_attention(qp.float().numpy(), kcache.float().numpy(), vcache.float().numpy(), cu_q, np.array(klens), bt, causal=True)
    check(outs[1].float().cpu().numpy().astype(np.float64), refp, "[paged]")


def test_attention_transpose_read_equals_plain_reads(ops):
    """The ds_read_b64_tr_b16 V^T fragments and plain 16-bit column reads give bit-identical outputs."""
    from visionselector_amd._native import debug_knob
    q, k, v = make_qkv(600, 8, 2, 11)
    cu = torch.tensor([0, 250, 600], dtype=torch.int32).cuda()
    with debug_knob("attn_split", 0):                      # same single-stream schedule on both sides
        a = ops.varlen_attn(q.cuda(), k.cuda(), v.cuda(), cu, 350)
        with debug_knob("attn_use_tr", 0):
            b = ops.varlen_attn(q.cuda(), k.cuda(), v.cuda(), cu, 350)
    assert torch.equal(a, b)


def test_attention_full_size_properties(ops):
    """Qwen2.5-VL-7B geometry at the uncompressed length (N + 64 = 2368): rows are convex combinations of V rows
    (|O| <= max|V|), first token attends only to itself, batch-of-sequences == separate calls."""
    hq, hkv = 28, 4
    lens = [2368, 524]
    q, k, v = make_qkv(sum(lens), hq, hkv, 3)
    cu = torch.tensor([0, lens[0], sum(lens)], dtype=torch.int32).cuda()
    qg, kg, vg = q.cuda(), k.cuda(), v.cuda()
    out = ops.varlen_attn(qg, kg, vg, cu, max(lens))
    assert bool(torch.isfinite(out.float()).all())
    assert float(out.float().abs().max()) <= float(v.float().abs().max()) * 1.01
    for s0 in (0, lens[0]):
        exp = vg[s0].float().repeat_interleave(hq // hkv, dim=0)
        assert float((out[s0].float() - exp).abs().max()) <= 2e-2        # softmax over one key = that V row (bf16)
    cu1 = torch.tensor([0, lens[1]], dtype=torch.int32).cuda()
    tail = [t[lens[0]:].contiguous() for t in (qg, kg, vg)]
    # the short sequence alone is a small grid -> two KV streams per workgroup (different summation order): equal to bf16
    # rounding; with the same schedule (split off) a sequence's output does not depend on what it is packed with, bit for bit
    o2 = ops.varlen_attn(*tail, cu1, lens[1])
    assert float((out[lens[0]:].float() - o2.float()).abs().max()) <= 2 ** -6 * float(o2.float().abs().max())
    with _split(0):
        assert torch.equal(out[lens[0]:], ops.varlen_attn(*tail, cu1, lens[1]))
    # spot-check a slice against the oracle (one kv group, last 64 queries of the long sequence)
    sl = slice(lens[0] - 64, lens[0])
    ref = oattn.varlen_attention(q[:lens[0], :7].float().numpy(), k[:lens[0], :1].float().numpy(),
                                 v[:lens[0], :1].float().numpy(), np.array([0, lens[0]]))[sl]
    check(out[sl, :7].float().cpu().numpy(), ref)


@pytest.mark.parametrize("page_size", [16, 64, 100])
@pytest.mark.parametrize("causal", [True, False])
def test_paged_attention_matches_oracle(ops, page_size, causal):
    """Paged KV cache, per-sequence key lengths != query lengths: chunked prefill (q < k), plain prefill (q == k),
    decode (q = 1) and a degenerate q > k sequence (leading rows see no key -> zeros)."""
    hq, hkv = 8, 2
    qlens = [70, 200, 1, 40]
    klens = [300, 200, 517, 25]
    rng = np.random.default_rng(page_size)
    pages_per = [-(-k // page_size) for k in klens]
    max_pages = max(pages_per)
    n_pages = sum(pages_per) + 3
    perm = rng.permutation(n_pages)                      # scattered physical pages
    bt = np.zeros((len(qlens), max_pages), np.int32)
    cur = 0
    for s, pp in enumerate(pages_per):
        bt[s, :pp] = perm[cur:cur + pp]
        cur += pp
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh, dtype=np.float32)).bfloat16()  # noqa: E731
    q = f(sum(qlens), hq, 128)
    kc, vc = f(n_pages, page_size, hkv, 128), f(n_pages, page_size, hkv, 128)
    cu_q = np.concatenate(([0], np.cumsum(qlens))).astype(np.int32)
    out = ops.paged_attn(q.cuda(), kc.cuda(), vc.cuda(), torch.from_numpy(cu_q).cuda(),
                         torch.tensor(klens, dtype=torch.int32).cuda(), torch.from_numpy(bt).cuda(), max(qlens), causal=causal)
    ref = oattn.paged_attention(q.float().numpy(), kc.float().numpy(), vc.float().numpy(), cu_q, np.array(klens), bt, causal=causal)
    check(out.float().cpu().numpy().astype(np.float64), ref)
    if causal:   # q > k sequence: its first qlen - klen rows see no key
        a = int(cu_q[3])
        assert float(out[a:a + 15].float().abs().max()) == 0.0


def test_paged_equals_contiguous_when_pages_are_in_order(ops):
    """Identity page table + seqlens_k == query lengths reproduces vsel_varlen_attn_fwd bit for bit."""
    lens = [130, 257, 64]
    q, k, v = make_qkv(sum(lens), 8, 2, 31)
    cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32).cuda()
    a = ops.varlen_attn(q.cuda(), k.cuda(), v.cuda(), cu, max(lens))
    page = 1
    bt = torch.zeros(len(lens), max(lens), dtype=torch.int32)
    off = 0
    for s, n in enumerate(lens):
        bt[s, :n] = torch.arange(off, off + n, dtype=torch.int32)
        off += n
    b = ops.paged_attn(q.cuda(), k.cuda().view(-1, page, 2, 128), v.cuda().view(-1, page, 2, 128), cu,
                       torch.tensor(lens, dtype=torch.int32).cuda(), bt.cuda(), max(lens))
    assert torch.equal(a, b)


def test_workgroup_shapes_agree(ops):
    """4-wave (128-query) and 8-wave (256-query) workgroups produce bit-identical outputs on a ragged batch."""
    from visionselector_amd._native import debug_knob
    lens = [700, 33, 256, 257, 1500]
    q, k, v = make_qkv(sum(lens), 8, 2, 41)
    cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32).cuda()
    outs = []
    for nw in (4, 8):
        with debug_knob("attn_waves", nw):
            outs.append(ops.varlen_attn(q.cuda(), k.cuda(), v.cuda(), cu, max(lens)))
    assert torch.equal(outs[0], outs[1])
    ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu.cpu().numpy())
    check(outs[1].float().cpu().numpy(), ref)


@pytest.mark.parametrize("d", [80, 64])
@pytest.mark.parametrize("lens,hq,hkv,causal", [([64] * 9, 16, 16, False), ([576], 16, 16, False), ([100, 64, 3, 333], 4, 2, True),
                                                 ([64, 64, 16, 48], 2, 2, False)])
def test_attention_vision_tower_head_dims(ops, d, lens, hq, hkv, causal):
    """head_dim 80 (Qwen2.5-VL ViT) and 64 (Rice ViT): packed non-causal windows of 64 tokens, full attention, ragged tails."""
    check(*run_case(ops, lens, hq, hkv, causal, seed=d + len(lens), d=d))


def test_attention_rejects_unsupported_head_dim(ops):
    from visionselector_amd._native import VselError
    q = torch.zeros(8, 2, 96, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(VselError, match="head_dim"):
        ops.varlen_attn(q, q, q, torch.tensor([0, 8], dtype=torch.int32, device="cuda"), 8)


@pytest.mark.parametrize("hq,hkv,qlens,klens", [(28, 4, [1, 1, 1, 1, 1], [524, 37, 64, 1, 2368]),      # decode, 7B heads
                                                  (16, 2, [3, 1, 4], [100, 999, 4]),                       # rep 8, chunks of <= 4
                                                  (8, 8, [20, 32, 1], [300, 32, 5])])                      # rep 1
@pytest.mark.parametrize("causal", [True, False])
def test_gqa_packed_decode_matches_oracle_and_unpacked_kernel(ops, hq, hkv, qlens, klens, causal):
    """Decode / short chunks against a paged cache: one wave serves the whole GQA group (lane = (query, head) pair), K/V
    streamed once per kv head.  Against the oracle, and against the per-head kernel (pack mode 0)."""
    page_size = 64
    rng = np.random.default_rng(hq + len(qlens))
    pages_per = [-(-kk // page_size) for kk in klens]
    n_pages = sum(pages_per) + 2
    perm = rng.permutation(n_pages)
    bt = np.zeros((len(qlens), max(pages_per)), np.int32)
    cur = 0
    for s, pp in enumerate(pages_per):
        bt[s, :pp] = perm[cur:cur + pp]
        cur += pp
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh, dtype=np.float32)).bfloat16()  # noqa: E731
    q = f(sum(qlens), hq, 128)
    kc, vc = f(n_pages, page_size, hkv, 128), f(n_pages, page_size, hkv, 128)
    cu_q = np.concatenate(([0], np.cumsum(qlens))).astype(np.int32)
    args = (q.cuda(), kc.cuda(), vc.cuda(), torch.from_numpy(cu_q).cuda(), torch.tensor(klens, dtype=torch.int32).cuda(),
            torch.from_numpy(bt).cuda(), max(qlens))
    from visionselector_amd import _native as N
    outs = {}
    for mode in (0, 1):                # compare the packed form with the plain per-head form (not the two-stream one)
        with N.debug_knob(attn_split=0, attn_pack=mode):
            N.profile_start()
            outs[mode] = ops.paged_attn(*args, causal=causal)
            prof = N.profile_stop()
            assert prof["varlen_attn_fwd_kernel"][1] == 1
    ref = oattn.paged_attention(q.float().numpy(), kc.float().numpy(), vc.float().numpy(), cu_q, np.array(klens), bt, causal=causal)
    for mode in (0, 1):
        check(outs[mode].float().cpu().numpy().astype(np.float64), ref)
    # same tiles, same order of operations per (query, head): the two forms agree to the last bit
    assert torch.equal(outs[0], outs[1])


def _split(mode):
    """with _split(0): ...  -- force / forbid the two-KV-stream form for the block (restored on exit)."""
    from visionselector_amd._native import debug_knob
    return debug_knob("attn_split", mode)


@pytest.mark.parametrize("lens,hq,hkv", [([524], 28, 4), ([65], 4, 2), ([1], 2, 1), ([300, 129, 64], 4, 2), ([2368], 4, 4)])
@pytest.mark.parametrize("causal", [True, False])
def test_two_kv_stream_workgroups_match_oracle_and_single_stream(ops, lens, hq, hkv, causal):
    """Small grids run with two KV streams per workgroup (the 4-wave groups split the KV tiles and merge their softmax
    states): against the oracle, and against the single-stream kernel within bf16 rounding (different summation order)."""
    q, k, v = make_qkv(sum(lens), hq, hkv, seed=len(lens) + hq)
    cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
    outs = {}
    for mode in (0, 1):
        with _split(mode):
            outs[mode] = ops.varlen_attn_fwd_lse(q.cuda(), k.cuda(), v.cuda(), cu, max(lens), causal=causal)
    ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu.cpu().numpy(), causal=causal)
    for mode in (0, 1):
        check(outs[mode][0].float().cpu().numpy().astype(np.float64), ref)
    assert float((outs[0][0].float() - outs[1][0].float()).abs().max()) <= 2 ** -6 * max(1.0, float(outs[0][0].float().abs().max()))
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-4          # log-sum-exp agrees


def test_two_kv_streams_with_paged_cache(ops):
    """Chunked prefill / decode of ONE sequence against a long paged cache: the two-stream form is the default there."""
    rng = np.random.default_rng(5)
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh, dtype=np.float32)).bfloat16()  # noqa: E731
    hq, hkv, page = 8, 2, 64
    for qlen, klen in ((1, 1000), (40, 700), (128, 128)):
        n_pages = -(-klen // page)
        bt = torch.from_numpy(rng.permutation(n_pages).astype(np.int32))[None]
        q, kc, vc = f(qlen, hq, 128), f(n_pages, page, hkv, 128), f(n_pages, page, hkv, 128)
        cu_q = torch.tensor([0, qlen], dtype=torch.int32)
        out = ops.paged_attn(q.cuda(), kc.cuda(), vc.cuda(), cu_q.cuda(), torch.tensor([klen], dtype=torch.int32).cuda(), bt.cuda(), qlen)
        ref = oattn.paged_attention(q.float().numpy(), kc.float().numpy(), vc.float().numpy(), cu_q.numpy(), np.array([klen]),
                                    bt.numpy(), causal=True)
        check(out.float().cpu().numpy().astype(np.float64), ref)


@pytest.mark.parametrize("b,hq,hkv,lq,lk,causal", [(2, 8, 2, 200, 200, True), (1, 28, 4, 524, 524, True), (3, 4, 4, 1, 77, True),
                                                    (2, 4, 2, 5, 300, True), (2, 4, 2, 130, 130, False), (70, 8, 1, 1, 200, True)])
def test_head_major_strided_entry_matches_packed(ops, b, hq, hkv, lq, lk, causal):
    """vsel_varlen_attn_fwd_strided on HuggingFace-layout tensors [B, H, L, d] == the packed entries on transposed copies
    (bit for bit: same kernels, only the addressing differs), incl. decode against a longer cache and the GQA-packed form."""
    rng = np.random.default_rng(b + lq)
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh, dtype=np.float32)).bfloat16().cuda()  # noqa: E731
    q, k, v = f(b, hq, lq, 128), f(b, hkv, lk, 128), f(b, hkv, lk, 128)
    out = ops.attn_head_major(q, k, v, causal=causal)
    assert out.shape == (b, lq, hq, 128)
    qp = q.transpose(1, 2).reshape(b * lq, hq, 128).contiguous()
    kp = k.transpose(1, 2).reshape(b * lk, hkv, 128).contiguous()
    vp = v.transpose(1, 2).reshape(b * lk, hkv, 128).contiguous()
    cu_q = torch.arange(0, (b + 1) * lq, lq, dtype=torch.int32).cuda()
    if lq == lk:
        ref = ops.varlen_attn(qp, kp, vp, cu_q, lq, causal=causal)
    else:
        cu_k = torch.arange(0, (b + 1) * lk, lk, dtype=torch.int32).cuda()
        ref = ops.varlen_attn_kv(qp, kp, vp, cu_q, cu_k, lq, causal=causal)
    assert torch.equal(out.reshape(b * lq, hq, 128), ref)
    # HF hands v over as a transposed VIEW of the packed projection output: same result without a copy
    v_view = vp.view(b, lk, hkv, 128).transpose(1, 2)
    assert not v_view.is_contiguous() or lk == 1 or hkv == 1
    assert torch.equal(ops.attn_head_major(q, k, v_view, causal=causal), out)


# ---- 64 rows per wave, one wave per SIMD, hand-scheduled tile loop (csrc/attn_fwd64.hip, knob attn_rows64) ----------------------
def _rows64_pair(ops, q, k, v, cu, L, causal, lse=False):
    """(4- / 8-wave form, 64-rows-per-wave form) on the same inputs; the profile proves which kernel ran"""
    from visionselector_amd import _native as N
    outs = []
    for r64 in (0, 1):
        with N.debug_knob(attn_rows64=r64, attn_split=0, attn_key_parts=0):      # (the two-KV-stream form and key-range parts sum in a different order)
            N.profile_start()
            outs.append(ops.varlen_attn_fwd_lse(q, k, v, cu, L, causal=causal) if lse else ops.varlen_attn(q, k, v, cu, L, causal=causal))
            prof = N.profile_stop()
            assert ("attn_fwd64_kernel" in prof) == bool(r64), prof
    return outs


@pytest.mark.parametrize("lens,hq,hkv", [([1], 2, 1), ([64], 4, 4), ([65], 4, 2), ([256], 4, 4), ([257], 4, 1), ([300, 129, 64], 4, 2),
                                         ([37, 700, 256, 129], 28, 4), ([1230], 32, 8), ([2368], 4, 4)])
@pytest.mark.parametrize("causal", [True, False])
def test_rows64_forward_matches_oracle_and_other_forms_bit_for_bit(ops, lens, hq, hkv, causal):
    """Same arithmetic per query row as the 4- / 8-wave kernels (the software pipeline only moves WHEN things are computed): outputs
    bit-identical, and within the usual gate of the fp64 oracle.  Ragged lengths exercise partial K / V tiles (row-clamped loads),
    partial query tiles (padding lanes), waves without a visible key and masks on the diagonal and at the end of the keys."""
    q, k, v = make_qkv(sum(lens), hq, hkv, 101 + len(lens))
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cu_t = torch.from_numpy(cu).cuda()
    a, b = _rows64_pair(ops, q.cuda(), k.cuda(), v.cuda(), cu_t, max(lens), causal)
    assert torch.equal(a, b)
    ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, causal=causal)
    check(b.float().cpu().numpy(), ref)


def test_rows64_forward_rescale_and_lazy_exponent(ops):
    """Reference exponents that move in the middle of a row (spiked keys, ramped logits): the rescale of O sits behind the P V MFMAs
    of the previous tile in the pipelined loop -- same results as the sequential forms, output and log-sum-exp."""
    lens = [900, 333, 1500]
    total = sum(lens)
    q, k, v = make_qkv(total, 4, 2, 29, spike=True)
    ramp = torch.linspace(0.2, 2.5, total).view(-1, 1, 1)
    k = (k.float() * ramp).bfloat16()
    k[700] *= 5
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cu_t = torch.from_numpy(cu).cuda()
    for causal in (True, False):
        (ao, al), (bo, bl) = _rows64_pair(ops, q.cuda(), k.cuda(), v.cuda(), cu_t, max(lens), causal, lse=True)
        assert torch.equal(ao, bo) and torch.equal(al, bl)
        ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, causal=causal)
        check(bo.float().cpu().numpy(), ref, f"[causal={causal}]")


def test_rows64_forward_with_longer_key_sequences(ops):
    """Queries against longer key sequences (chunked prefill: bottom-right aligned causal mask, shift = klen - qlen), keys at their own
    offsets: the 64-rows-per-wave form through vsel_varlen_attn_fwd_kv agrees bit for bit with the per-head kernel."""
    from visionselector_amd import _native as N
    qlens, klens = [100, 256, 33], [700, 256, 512]
    rng = np.random.default_rng(77)
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh, dtype=np.float32)).bfloat16()  # noqa: E731
    q, k, v = f(sum(qlens), 8, 128), f(sum(klens), 2, 128), f(sum(klens), 2, 128)
    cu_q = torch.tensor(np.concatenate(([0], np.cumsum(qlens))), dtype=torch.int32).cuda()
    cu_k = torch.tensor(np.concatenate(([0], np.cumsum(klens))), dtype=torch.int32).cuda()
    outs = []
    for r64 in (0, 1):
        with N.debug_knob(attn_rows64=r64, attn_split=0, attn_pack=0):
            outs.append(ops.varlen_attn_kv(q.cuda(), k.cuda(), v.cuda(), cu_q, cu_k, max(qlens), causal=True))
    assert torch.equal(outs[0], outs[1])


def test_rows64_is_the_default_for_large_grids(ops):
    """16 x 2368 tokens at 7B heads is the regime of the 8-wave form: the 64-rows-per-wave kernel takes it by default."""
    from visionselector_amd import _native as N
    L, n = 2368, 16
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(n * L, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(n * L, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(n * L, 4, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, n * L + 1, L, dtype=torch.int32, device="cuda")
    N.profile_start()
    a = ops.varlen_attn(q, k, v, cu, L)
    prof = N.profile_stop()
    assert "attn_fwd64_kernel" in prof, prof
    with N.debug_knob(attn_rows64=0):
        b = ops.varlen_attn(q, k, v, cu, L)
    assert torch.equal(a, b)


@pytest.mark.parametrize("b,hq,hkv,lq,lk,causal", [(2, 8, 2, 300, 300, True), (1, 4, 4, 2368, 2368, True), (2, 4, 2, 70, 700, True),
                                                    (2, 4, 2, 257, 257, False)])
def test_rows64_forward_on_head_major_strided_tensors(ops, b, hq, hkv, lq, lk, causal):
    """HuggingFace-layout tensors [B, H, L, d] (row stride d, head stride L d; V as a transposed view with its own strides) through the
    64-rows-per-wave form: its Q rows come by whole-row direct-to-LDS loads at the caller's row stride -- bit-identical to the other
    forms on the same strided tensors."""
    from visionselector_amd import _native as N
    rng = np.random.default_rng(3 * b + lq)
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh, dtype=np.float32)).bfloat16().cuda()  # noqa: E731
    q, k = f(b, hq, lq, 128), f(b, hkv, lk, 128)
    v = f(b, lk, hkv, 128).transpose(1, 2)                 # a view: row stride Hkv d, head stride d
    outs = []
    for r64 in (0, 1):
        with N.debug_knob(attn_rows64=r64, attn_split=0, attn_pack=0):
            N.profile_start()
            outs.append(ops.attn_head_major(q, k, v, causal=causal))
            prof = N.profile_stop()
            assert ("attn_fwd64_kernel" in prof) == bool(r64), prof
    assert torch.equal(outs[0], outs[1])


# ---- group-shared forward for short sequences (csrc/attn_fwd_gqa.hip, round 5) ----------------------------------------------------------
def _gqa_pair(ops, q, k, v, cu, L, causal, lse=False):
    """(per-head 4- / 8-wave form, group-shared form) on the same inputs; the profile proves which kernel ran"""
    from visionselector_amd import _native as N
    outs = []
    for g, form, name in ((0, 0, None), (1, 1, "attn_fwd_gqa64_kernel"), (1, 0, "attn_fwd_gqa_kernel")):
        with N.debug_knob(attn_gqa=g, attn_gqa_form=form, attn_split=0, attn_rows64=0):
            N.profile_start()
            outs.append(ops.varlen_attn_fwd_lse(q, k, v, cu, L, causal=causal) if lse else ops.varlen_attn(q, k, v, cu, L, causal=causal))
            prof = N.profile_stop()
            assert [n for n in prof if "gqa" in n] == ([name] if name else []), prof
    # the two group-shared forms agree bit for bit; the caller compares outs[0] (per head) with outs[1]
    for b in outs[2:]:
        a = outs[1]
        if lse:
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        else:
            assert torch.equal(a, b)
    return outs[:2]


@pytest.mark.parametrize("lens,hq,hkv", [([524], 28, 4), ([1], 28, 4), ([31, 32, 33, 64, 65, 1, 127, 300], 28, 4), ([524] * 5, 28, 4),
                                         ([793, 210, 17], 32, 8), ([300, 129, 64], 16, 2), ([257, 96], 4, 2), ([100, 333], 6, 2),
                                         ([1230, 64], 12, 2)])
@pytest.mark.parametrize("causal", [True, False])
def test_gqa_shared_forward_matches_oracle_and_per_head_forms_bit_for_bit(ops, lens, hq, hkv, causal):
    """One workgroup per (query tile, kv head) serving the whole q-head group: same arithmetic per query row as the per-head kernels
    (64-key tiles from key 0, skipped blocks are exactly the fully masked ones) -> outputs and log-sum-exps bit-identical, and within
    the usual gate of the fp64 oracle.  Group sizes 7 (7B: 7 head waves + a helper), 4 (LLaVA-OV: two 32-query slices), 8 (3B),
    2 / 3 / 5 / 6; ragged lengths exercise empty items (levels a shorter sequence does not reach), partial query tiles, partial key
    tiles, single-tile items and the item-to-item prefetch across sequences."""
    q, k, v = make_qkv(sum(lens), hq, hkv, 211 + len(lens) + hq)
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cu_t = torch.from_numpy(cu).cuda()
    (ao, al), (bo, bl) = _gqa_pair(ops, q.cuda(), k.cuda(), v.cuda(), cu_t, max(lens), causal, lse=True)
    assert torch.equal(ao, bo)
    assert torch.equal(al, bl)
    ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, causal=causal)
    check(bo.float().cpu().numpy(), ref)


@pytest.mark.parametrize("lens,hq,hkv", [([31, 32, 33, 64, 65, 1, 127, 300], 28, 4), ([524] * 8, 28, 4), ([793, 210, 17], 32, 8),
                                         ([900, 5, 700, 64, 1, 333, 2, 450, 800, 120], 28, 4), ([294] * 20, 16, 2)])
def test_gqa_shared_forward_dealt_and_queued_items_agree(ops, lens, hq, hkv):
    """More items than workgroups: the group-shared kernels either deal the heaviest-first item list out (workgroup b: items b, 2 G - 1 - b,
    2 G + b, ...; uniform batches by default, knob attn_static = 1 anywhere) or draw it from the queue behind two dealt rounds (ragged
    batches, attn_static = 0: a workgroup whose own item is empty starts on its mirror item).  Placement only: every setting of both forms
    gives the per-head kernels' bits, also when most first-round items are empty."""
    from visionselector_amd import _native as N
    q, k, v = make_qkv(sum(lens), hq, hkv, 77 + len(lens))
    cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
    q, k, v = q.cuda(), k.cuda(), v.cuda()
    with N.debug_knob(attn_gqa=0, attn_split=0, attn_rows64=0):
        ref = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens))
    for form in (0, 1):
        for static in (0, 1, -1):
            with N.debug_knob(attn_gqa=1, attn_gqa_form=form, attn_static=static, attn_split=0, attn_rows64=0):
                got = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens))
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (form, static)


@pytest.mark.parametrize("lens,hq,hkv,cap", [([2368], 4, 4, -1), ([2368], 8, 2, -1), ([2368], 28, 4, 1), ([2112, 2112], 8, 2, 1), ([4096], 4, 1, 9),
                                             ([2049], 6, 2, 8), ([8192], 4, 2, -1), ([3000], 12, 4, 12)])
def test_key_range_parts_forward_matches_oracle_and_unsplit_form(ops, lens, hq, hkv, cap):
    """One long sequence (or a few of equal length) through vsel_varlen_attn_fwd_ws: 256-query items cut over key ranges, fp32 partials +
    (m, l) in a workspace, merged in part order (csrc/attn_fwd64_parts.hip).  Within the usual gate of the fp64 oracle, within a bf16 rounding
    of the unsplit form (another fp32 association), bit-identical between two runs; cap = knob attn_key_parts (-1: the library's own rule --
    few heads; 1: forced with its own cap; n: n key tiles per part); partial last tiles (2049, 3000), GQA 1 / 3 / 4 / 7, two sequences."""
    from visionselector_amd import _native as N
    q, k, v = make_qkv(sum(lens), hq, hkv, 91 + hq)
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cu_t = torch.from_numpy(cu).cuda()
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    with N.debug_knob(attn_key_parts=0):
        base = ops.varlen_attn(qc, kc, vc, cu_t, max(lens))
    with N.debug_knob(attn_key_parts=cap):
        N.profile_start()
        got = ops.varlen_attn(qc, kc, vc, cu_t, max(lens), key_parts=True)
        prof = N.profile_stop()
        again = ops.varlen_attn(qc, kc, vc, cu_t, max(lens), key_parts=True)
        N.profile_start()
        ops.varlen_attn(qc, kc, vc, cu_t, max(lens))            # the default never takes the parts form (opt-in: ADVICE r5)
        assert "attn_fwd64_parts_kernel" not in N.profile_stop()
    assert "attn_fwd64_parts_kernel" in prof and "attn_fwd64_merge_kernel" in prof, prof
    assert torch.equal(got, again)
    assert float((got.float() - base.float()).abs().max()) <= 2 ** -7 * float(base.float().abs().max())
    ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, causal=True)
    check(got.float().cpu().numpy(), ref)


def test_key_range_parts_are_not_taken_for_other_batches(ops):
    """vsel_varlen_attn_fwd_workspace_bytes is 0 -- and the workspace-free forms run -- for ragged batches, non-causal calls, short
    sequences, and (by the library's own rule) more than half a round of items; a too-small workspace is refused loudly."""
    import ctypes as C
    from visionselector_amd import _native as N
    lib = N.lib()
    wsb = lib.vsel_varlen_attn_fwd_workspace_bytes
    assert wsb(1, 2368, 2368, 4, 4, 128, 1) > 0
    assert wsb(1, 2368, 2368, 28, 4, 128, 1) == 0            # 280 items: the 7B geometry keeps the unsplit form (measured: slower in parts)
    assert wsb(2, 2368, 4000, 4, 4, 128, 1) == 0             # ragged
    assert wsb(1, 2368, 2368, 4, 4, 128, 0) == 0             # not causal
    assert wsb(1, 1024, 1024, 4, 4, 128, 1) == 0             # short
    q, k, v = make_qkv(2368, 4, 4, 5)
    cu_t = torch.tensor([0, 2368], dtype=torch.int32, device="cuda")
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    out = torch.empty_like(qc)
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    rc = lib.vsel_varlen_attn_fwd_ws(None, qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), cu_t.data_ptr(), 1, 2368, 2368, 4, 4, 128,
                                     128 ** -0.5, 1, out.data_ptr(), None, ws.data_ptr(), ws.numel())
    assert rc == 2, rc                                        # VSEL_ERR_WORKSPACE
    lse = torch.empty(2368, 4, dtype=torch.float32, device="cuda")
    need = wsb(1, 2368, 2368, 4, 4, 128, 1)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    N.check(lib.vsel_varlen_attn_fwd_ws(None, qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), cu_t.data_ptr(), 1, 2368, 2368, 4, 4, 128,
                                        128 ** -0.5, 1, out.data_ptr(), lse.data_ptr(), ws.data_ptr(), ws.numel()))
    with N.debug_knob(attn_key_parts=0):
        o2, l2 = ops.varlen_attn_fwd_lse(qc, kc, vc, cu_t, 2368)
    assert float((out.float() - o2.float()).abs().max()) <= 2 ** -7 * float(o2.float().abs().max())
    assert float((lse - l2).abs().max()) <= 1e-4 * max(1.0, float(l2.abs().max()))


def test_gqa_shared_forward_rescale_and_lazy_exponent(ops):
    lens = [900, 333, 1500]
    total = sum(lens)
    q, k, v = make_qkv(total, 8, 2, 31, spike=True)
    ramp = torch.linspace(0.2, 2.5, total).view(-1, 1, 1)
    k = (k.float() * ramp).bfloat16()
    k[700] *= 5
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cu_t = torch.from_numpy(cu).cuda()
    for causal in (True, False):
        (ao, al), (bo, bl) = _gqa_pair(ops, q.cuda(), k.cuda(), v.cuda(), cu_t, max(lens), causal, lse=True)
        assert torch.equal(ao, bo) and torch.equal(al, bl)
        ref = oattn.varlen_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, causal=causal)
        check(bo.float().cpu().numpy(), ref, f"[causal={causal}]")


def test_gqa_shared_forward_many_ragged_prompts_is_deterministic_and_default(ops):
    """Config 5's shape: 64 compressed prompts of 131 .. 947 tokens at 7B heads -- more items than workgroups (work queue drawn two items
    ahead, runs of empty items skipped), the default form for such grids; two runs and the per-head form agree bit for bit."""
    from visionselector_amd import _native as N
    rng = np.random.default_rng(5)
    lens = [int(x) for x in rng.integers(131, 948, size=64)]
    total = sum(lens)
    g = torch.Generator(device="cuda").manual_seed(11)
    q = torch.randn(total, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(total, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(total, 4, 128, device="cuda", generator=g).bfloat16()
    cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
    N.profile_start()
    a = ops.varlen_attn(q, k, v, cu, max(lens))
    prof = N.profile_stop()
    assert "attn_fwd_gqa_kernel" in prof, prof
    b = ops.varlen_attn(q, k, v, cu, max(lens))
    assert torch.equal(a, b)
    with N.debug_knob(attn_gqa=0):
        c = ops.varlen_attn(q, k, v, cu, max(lens))
    assert torch.equal(a, c)


@pytest.mark.parametrize("b,hq,hkv,l,causal", [(3, 28, 4, 524, True), (2, 32, 8, 300, True), (2, 16, 2, 257, False)])
def test_gqa_shared_forward_on_head_major_strided_tensors(ops, b, hq, hkv, l, causal):
    """HuggingFace-layout tensors [B, H, L, d] through the group-shared form: bit-identical to the per-head form on the same views."""
    from visionselector_amd import _native as N
    rng = np.random.default_rng(5 * b + l)
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh, dtype=np.float32)).bfloat16().cuda()  # noqa: E731
    q, k = f(b, hq, l, 128), f(b, hkv, l, 128)
    v = f(b, l, hkv, 128).transpose(1, 2)
    outs = []
    for g in (0, 1):
        with N.debug_knob(attn_gqa=g, attn_split=0, attn_rows64=0, attn_pack=0):
            N.profile_start()
            outs.append(ops.attn_head_major(q, k, v, causal=causal))
            prof = N.profile_stop()
            assert any("gqa" in n for n in prof) == bool(g), prof
    assert torch.equal(outs[0], outs[1])


def test_work_queue_slots_across_streams(ops):
    """Queued attention launches on many streams at once: every stream owns a work-queue counter slot (csrc/common.hip), so concurrent
    launches never share a counter -- results equal the single-stream result on each of 70 streams (more than the 64 slots: idle streams'
    slots are taken over)."""
    lens = [524] * 16
    total = sum(lens)
    g = torch.Generator(device="cuda").manual_seed(21)
    q = torch.randn(total, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(total, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(total, 4, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, total + 1, 524, dtype=torch.int32, device="cuda")
    ref = ops.varlen_attn(q, k, v, cu, 524)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(70)]
    outs = []
    for rnd in range(2):
        for st in streams:
            with torch.cuda.stream(st):
                outs.append(ops.varlen_attn(q, k, v, cu, 524))
        torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)
