"""GPU parity of the fused device splice (vsel_splice) against the reference's goldens (tests/golden/splice_*.npz,
produced by running the reference's own Qwen2_5_VLForConditionalGeneration_Selector.forward) and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import inputs as oin
from oracle import splice as osplice

pytestmark = pytest.mark.gpu
IMAGE_TOKEN, VIDEO_TOKEN = 151655, 151656


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from visionselector_amd import ops as _ops
    return _ops


def _embed(ids, d_llm):
    ar = torch.arange(d_llm, dtype=torch.int64)
    return ((ids[..., None] * 31 + ar * 17) % 257).float() / 257.0


@pytest.mark.parametrize("name", ["image_a", "image_b", "video_a"])
def test_splice_matches_reference_golden(ops, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"splice_{name}.npz"))
    vis = IMAGE_TOKEN if str(g["kind"]) == "image" else VIDEO_TOKEN
    ids = torch.from_numpy(oin.make_prompt(int(g["n_visual"]), int(g["n_pre"]), int(g["n_post"]), vis, int(g["seed"])))
    emb = _embed(ids, int(g["d_llm"]))
    sel, new_ids, new_emb, new_pos, new_am = ops.splice(
        ids.cuda(), emb.cuda(), vis, torch.from_numpy(g["all_idx"]).cuda(), torch.from_numpy(g["vis_embeds"]).cuda(),
        int(g["n_visual"]), position_ids=torch.from_numpy(g["position_ids_full"]).cuda(),
        attention_mask=torch.ones_like(ids).cuda(), check=True)
    assert np.array_equal(new_emb.cpu().numpy(), g["inputs_embeds"])          # bit-exact copies
    assert np.array_equal(new_pos.cpu().numpy(), g["position_ids"])
    assert np.array_equal(new_am.cpu().numpy(), g["attention_mask"])
    if str(g["kind"]) == "image":
        ref_sel, ref_ids = osplice.splice_image(ids.numpy(), vis, g["all_idx"])
    else:
        ref_sel, ref_ids, _ = osplice.splice_video(ids.numpy(), vis, g["all_idx"])
    assert np.array_equal(sel.cpu().numpy(), ref_sel) and np.array_equal(new_ids.cpu().numpy(), ref_ids)


@pytest.mark.parametrize("L,n_vis,k,dt", [(5000, 4096, 819, torch.bfloat16), (70, 64, 1, torch.float32),
                                           (3000, 2304, 2304, torch.bfloat16), (1500, 1030, 0, torch.bfloat16)])
def test_splice_random_layouts_match_oracle(ops, L, n_vis, k, dt):
    """Visual tokens interleaved with text (multi-image prompts), long sequences (several 1024-chunks), k = N and k = 0."""
    rng = np.random.default_rng(L + k)
    ids = rng.integers(10, 1000, L).astype(np.int64)
    vis_pos = np.sort(rng.choice(L, n_vis, replace=False))
    ids[vis_pos] = IMAGE_TOKEN
    idx = np.sort(rng.choice(n_vis, k, replace=False)).astype(np.int64)
    d = 256
    emb = torch.from_numpy(rng.standard_normal((1, L, d), dtype=np.float32)).to(dt)
    ve = torch.from_numpy(rng.standard_normal((k, d), dtype=np.float32)).to(dt)
    pos = torch.from_numpy(rng.integers(0, 9999, (3, 1, L)).astype(np.int64))
    am = torch.from_numpy(rng.integers(0, 2, (1, L)).astype(np.int64))
    sel, new_ids, new_emb, new_pos, new_am = ops.splice(torch.from_numpy(ids)[None].cuda(), emb.cuda(), IMAGE_TOKEN,
                                                        torch.from_numpy(idx).cuda(), ve.cuda(), n_vis, position_ids=pos.cuda(),
                                                        attention_mask=am.cuda(), check=True)
    ref_sel, ref_ids = osplice.splice_image(ids[None], IMAGE_TOKEN, idx)
    assert np.array_equal(sel.cpu().numpy(), ref_sel) and np.array_equal(new_ids.cpu().numpy(), ref_ids)
    ref_emb = osplice.splice_embeds(emb.float().numpy(), ref_ids, ref_sel, IMAGE_TOKEN, ve.float().numpy())
    assert np.array_equal(new_emb.float().cpu().numpy(), ref_emb)
    rp, ra = osplice.slice_positions(pos.numpy(), am.numpy(), ref_sel)
    assert np.array_equal(new_pos.cpu().numpy(), rp) and np.array_equal(new_am.cpu().numpy(), ra)


def test_splice_token_count_mismatch_raises(ops):
    ids = torch.tensor([[5, IMAGE_TOKEN, IMAGE_TOKEN, 7]]).cuda()
    emb = torch.zeros(1, 4, 8).cuda()
    with pytest.raises(ValueError, match="do not match"):      # reference: ValueError (FT selector_model.py:210-213)
        ops.splice(ids, emb, IMAGE_TOKEN, torch.tensor([0]).cuda(), torch.zeros(1, 8).cuda(), 3, check=True)
    with pytest.raises(ValueError, match="single batch"):      # reference assert (EV :270)
        ops.splice(ids.repeat(2, 1), emb.repeat(2, 1, 1), IMAGE_TOKEN, torch.tensor([0]).cuda(), torch.zeros(1, 8).cuda(), 2)


def _packed_case(rng, seq_lens, visual_lens, ks, d, dt):
    ids_l, idx_l = [], []
    for l_s, n_s, k_s in zip(seq_lens, visual_lens, ks):
        ids = rng.integers(10, 1000, l_s).astype(np.int64)
        ids[np.sort(rng.choice(l_s, n_s, replace=False))] = IMAGE_TOKEN
        ids_l.append(ids)
        idx_l.append(np.sort(rng.choice(n_s, k_s, replace=False)).astype(np.int64))
    ids = np.concatenate(ids_l)
    idx = np.concatenate(idx_l) if sum(ks) else np.zeros(0, np.int64)
    t = ids.shape[0]
    emb = torch.from_numpy(rng.standard_normal((t, d), dtype=np.float32)).to(dt)
    ve = torch.from_numpy(rng.standard_normal((sum(ks), d), dtype=np.float32)).to(dt)
    pos = torch.from_numpy(rng.integers(0, 9999, (3, t)).astype(np.int64))
    return ids, idx, emb, ve, pos


@pytest.mark.parametrize("seq_lens,visual_lens,ks,dt", [
    ([700, 2400, 90, 1300], [576, 2304, 0, 1100], [115, 460, 0, 1100], torch.bfloat16),   # no-image prompt, k = N prompt
    ([40], [30], [6], torch.float32),                                                      # S = 1 equals the batch-1 entry
    ([4200, 64, 64, 3000, 1025], [4096, 1, 64, 2900, 1024], [819, 1, 0, 580, 204], torch.bfloat16),
])
def test_splice_batched_matches_per_prompt_reference(ops, seq_lens, visual_lens, ks, dt):
    """vsel_splice_batched == the reference's batch-1 index algebra applied per prompt and concatenated (bit-exact), and
    its cu_seqlens' are the per-prompt output lengths."""
    rng = np.random.default_rng(sum(seq_lens) + sum(ks))
    ids, idx, emb, ve, pos = _packed_case(rng, seq_lens, visual_lens, ks, 128, dt)
    sel, new_ids, new_emb, new_pos, cu = ops.splice_batched(
        torch.from_numpy(ids).cuda(), emb.cuda(), IMAGE_TOKEN, seq_lens, visual_lens, ks, torch.from_numpy(idx).cuda(),
        ve.cuda(), position_ids=pos.cuda(), check=True)
    r_sel, r_ids, r_emb, r_pos, r_cu = osplice.splice_packed(ids, emb.float().numpy(), IMAGE_TOKEN, seq_lens, visual_lens, ks,
                                                             idx, ve.float().numpy(), pos.numpy())
    assert np.array_equal(cu.cpu().numpy(), r_cu)
    assert np.array_equal(sel.cpu().numpy(), r_sel) and np.array_equal(new_ids.cpu().numpy(), r_ids)
    assert np.array_equal(new_emb.float().cpu().numpy(), r_emb) and np.array_equal(new_pos.cpu().numpy(), r_pos)
    if len(seq_lens) == 1:
        s1 = ops.splice(torch.from_numpy(ids)[None].cuda(), emb[None].cuda(), IMAGE_TOKEN, torch.from_numpy(idx).cuda(), ve.cuda(),
                        visual_lens[0], position_ids=pos[:, None].cuda())
        assert torch.equal(s1[0], sel) and torch.equal(s1[1][0], new_ids) and torch.equal(s1[2][0], new_emb)


def test_splice_batched_reports_mismatch(ops):
    """A prompt whose visual-token count disagrees with visual_lens raises the reference's ValueError under check=True."""
    rng = np.random.default_rng(5)
    ids, idx, emb, ve, pos = _packed_case(rng, [100, 200], [50, 120], [10, 24], 64, torch.bfloat16)
    ids[np.where(ids[:100] == IMAGE_TOKEN)[0][0]] = 11          # prompt 0 now holds 49 visual tokens
    with pytest.raises(ValueError, match="do not match"):
        ops.splice_batched(torch.from_numpy(ids).cuda(), emb.cuda(), IMAGE_TOKEN, [100, 200], [50, 120], [10, 24],
                           torch.from_numpy(idx).cuda(), ve.cuda(), check=True)
    with pytest.raises(ValueError):
        ops.splice_batched(torch.from_numpy(ids).cuda(), emb.cuda(), IMAGE_TOKEN, [100, 200], [50, 220], [10, 24],
                           torch.from_numpy(idx).cuda(), ve.cuda())


def test_select_splice_attend_packed_pipeline(ops):
    """Config C5 end to end on device: ragged LIS select -> packed splice -> var-len attention over cu_seqlens', against
    the oracle run prompt by prompt."""
    from oracle import attention as oattn
    from oracle import lis as olis
    rng = np.random.default_rng(77)
    d, hd = 256, 128
    visual_lens, texts = [576, 1024, 300], [40, 17, 64]
    ks = [int(n * 0.2) for n in visual_lens]
    seq_lens = [n + t for n, t in zip(visual_lens, texts)]
    h = torch.from_numpy(rng.standard_normal((sum(visual_lens), d), dtype=np.float32))
    wq, wk = [torch.from_numpy(rng.standard_normal((hd, d), dtype=np.float32) * 0.02) for _ in range(2)]
    bq, bk = [torch.from_numpy(rng.standard_normal(hd, dtype=np.float32) * 0.02) for _ in range(2)]
    out, idx, _ = ops.lis_select_varlen(h.cuda(), visual_lens, ks, wq.cuda(), bq.cuda(), wk.cuda(), bk.cuda())
    ids_l = []
    for n, t in zip(visual_lens, texts):
        ids_l.append(np.concatenate([rng.integers(10, 1000, t // 2), np.full(n, IMAGE_TOKEN), rng.integers(10, 1000, t - t // 2)]))
    ids = np.concatenate(ids_l).astype(np.int64)
    emb = torch.from_numpy(rng.standard_normal((ids.shape[0], d), dtype=np.float32))
    sel, new_ids, new_emb, _, cu = ops.splice_batched(torch.from_numpy(ids).cuda(), emb.cuda(), IMAGE_TOKEN, seq_lens, visual_lens,
                                                      ks, idx, out, check=True)
    # oracle: per-prompt select + splice
    o_idx, v0 = [], 0
    for n, k in zip(visual_lens, ks):
        s = olis.scorer_collapsed(h[None, v0:v0 + n].numpy(), wq.numpy(), bq.numpy(), wk.numpy(), bk.numpy())[0]
        o_idx.append(olis.hard_topk_indices(np.asarray(s, np.float32), k))
        v0 += n
    o_idx = np.concatenate(o_idx)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    offs = np.repeat(np.cumsum([0] + visual_lens[:-1]), ks)
    r_sel, r_ids, r_emb, _, r_cu = osplice.splice_packed(ids, emb.numpy(), IMAGE_TOKEN, seq_lens, visual_lens, ks, o_idx,
                                                         h.numpy()[o_idx + offs])
    assert np.array_equal(cu.cpu().numpy(), r_cu) and np.array_equal(sel.cpu().numpy(), r_sel)
    assert np.array_equal(new_emb.cpu().numpy(), r_emb)
    # attention over the compressed packed batch (2 q heads, 1 kv head, d 128 taken from the embeddings)
    x = new_emb.to(torch.bfloat16)
    q = x[:, :256].reshape(-1, 2, 128).contiguous()
    kk = x[:, 128:256].reshape(-1, 1, 128).contiguous()
    vv = x[:, :128].reshape(-1, 1, 128).contiguous()
    o = ops.varlen_attn(q, kk, vv, cu, max(seq_lens))
    ref = oattn.varlen_attention(q.float().cpu().numpy(), kk.float().cpu().numpy(), vv.float().cpu().numpy(), r_cu)
    assert np.abs(o.float().cpu().numpy() - ref).max() <= 2e-2          # bf16 output of values ~N(0,1)


def test_config5_real_geometry_select_splice_attend(ops):
    """BASELINE config 5 at its real geometry: Qwen2.5-VL-7B (D 3584, Hd 1792, 28 / 4 heads, head_dim 128), 8 prompts with
    576 .. 4096 visual tokens and 16 .. 128 text tokens, per-prompt k = int(0.2 N): ragged select == each prompt scored alone
    (bit for bit) and == the fp64 oracle's indices; packed splice offsets; var-len attention over the compressed packing ==
    each sequence attended alone (bit for bit, same schedule) and within the forward gate of the oracle on one sequence."""
    import parity
    from oracle import attention as oattn
    from oracle import lis as olis
    d, hd, hq, hkv = 3584, 1792, 28, 4
    rng = np.random.default_rng(5)
    n_vis = [int(x) for x in rng.integers(576, 4097, 8)]
    n_vis[0], n_vis[1] = 576, 4096                                   # both ends of the range
    n_txt = [int(x) for x in rng.integers(16, 129, 8)]
    ks = [int(n * 0.2) for n in n_vis]
    seq = [n + t for n, t in zip(n_vis, n_txt)]
    g = torch.Generator(device="cuda").manual_seed(50)
    h = torch.randn(sum(n_vis), d, device="cuda", generator=g).bfloat16()
    wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    out, idx, scores = ops.lis_select_varlen(h, n_vis, ks, wq, bq, wk, bk)
    v0 = o0 = 0
    for i, (n, k) in enumerate(zip(n_vis, ks)):
        o1, i1, s1 = ops.lis_select(h[v0:v0 + n].contiguous(), wq, bq, wk, bk, k)
        assert torch.equal(i1, idx[o0:o0 + k]) and torch.equal(o1, out[o0:o0 + k]) and torch.equal(s1, scores[v0:v0 + n])
        if i in (0, 1, 5):
            f = lambda t: t.float().cpu().numpy()  # noqa: E731
            ref = olis.scorer_collapsed(f(h[v0:v0 + n])[None], f(wq), f(bq), f(wk), f(bk))[0]
            assert np.array_equal(olis.hard_topk_indices(ref.astype(np.float32), k), i1.cpu().numpy())
        v0 += n
        o0 += k
    ids = torch.cat([torch.cat((torch.randint(10, 1000, (t // 2,)), torch.full((n,), IMAGE_TOKEN), torch.randint(10, 1000, (t - t // 2,))))
                     for n, t in zip(n_vis, n_txt)]).cuda()
    emb = torch.randn(sum(seq), d, device="cuda", generator=g).bfloat16()
    pos = torch.arange(sum(seq), device="cuda")[None].expand(3, -1).contiguous()
    sel, new_ids, new_emb, new_pos, cu_c = ops.splice_batched(ids, emb, IMAGE_TOKEN, seq, n_vis, ks, idx, out, position_ids=pos, check=True)
    seq_c = [k + t for k, t in zip(ks, n_txt)]
    assert cu_c.tolist() == np.concatenate(([0], np.cumsum(seq_c))).tolist()
    assert int((new_ids == IMAGE_TOKEN).sum()) == sum(ks) and torch.equal(new_pos[0], sel)
    assert torch.equal(new_emb[new_ids == IMAGE_TOKEN], out)
    t = sum(seq_c)
    q = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
    kk = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
    vv = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
    o = ops.varlen_attn(q, kk, vv, cu_c, max(seq_c))
    from visionselector_amd._native import debug_knob
    with debug_knob("attn_split", 0):        # single-stream schedule on both sides: packing invariance is bit-exact
        o_same = ops.varlen_attn(q, kk, vv, cu_c, max(seq_c))
        a = int(cu_c[3]); b = int(cu_c[4])
        alone = ops.varlen_attn(q[a:b].contiguous(), kk[a:b].contiguous(), vv[a:b].contiguous(),
                                torch.tensor([0, b - a], dtype=torch.int32, device="cuda"), b - a)
    assert torch.equal(o_same[a:b], alone)
    ref = oattn.varlen_attention(q[a:b, :7].float().cpu().numpy(), kk[a:b, :1].float().cpu().numpy(), vv[a:b, :1].float().cpu().numpy(),
                                 np.array([0, b - a]))
    parity.check_fwd("test_config5_real_geometry", o[a:b, :7].float().cpu().numpy().astype(np.float64), ref)


# ---------------------------------------------------------------------------------------------------
# select -> splice in one call, kept rows written once (vsel_lis_select_splice / vsel_topk_select_splice)
# ---------------------------------------------------------------------------------------------------
def _forms():
    """(knob value, kernel that must have run): the one-launch form and the three-launch general form."""
    return ((1, "select_splice_small_kernel"), (0, "splice_index_seg_kernel"))


@pytest.mark.parametrize("name", ["image_a", "image_b", "video_a"])
def test_topk_select_splice_matches_reference_golden(ops, golden_dir, name):
    """The reference's own splice goldens through the fused entry: scores that make top-k pick exactly the golden's
    all_indices, a token tensor whose kept rows are the golden's kept embeddings -> every output bit-exact, both forms."""
    from visionselector_amd import _native as N
    g = np.load(os.path.join(golden_dir, f"splice_{name}.npz"))
    vis = IMAGE_TOKEN if str(g["kind"]) == "image" else VIDEO_TOKEN
    n, k, d = int(g["n_visual"]), int(g["k"]), int(g["d_llm"])
    ids = torch.from_numpy(oin.make_prompt(n, int(g["n_pre"]), int(g["n_post"]), vis, int(g["seed"])))
    emb = _embed(ids, d)[0]
    rng = np.random.default_rng(1)
    h = rng.standard_normal((n, d), dtype=np.float32)
    h[g["all_idx"]] = g["vis_embeds"]
    scores = (rng.random(n, dtype=np.float32) * 0.5).astype(np.float32)
    scores[g["all_idx"]] += 1.0                                      # kept ranks strictly above the rest
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(3))
    h_phys = torch.empty(n, d)
    h_phys[perm] = torch.from_numpy(h)                               # logical row i lives at physical row perm[i]
    for knob, kernel in _forms():
        for hh, l2p in ((torch.from_numpy(h), None), (h_phys, perm)):
            with N.debug_knob("lis_splice_fused", knob):
                N.profile_start()
                o = ops.topk_select_splice(torch.from_numpy(scores).cuda(), hh.cuda(), ids[0].cuda(), emb.cuda(), vis, [ids.shape[1]],
                                           [n], [k], position_ids=torch.from_numpy(g["position_ids_full"]).cuda()[:, 0, :],
                                           attention_mask=torch.ones_like(ids[0]).cuda(),
                                           logical_to_physical=None if l2p is None else l2p.cuda(), check=True)
                prof = N.profile_stop()
            assert kernel in prof, prof
            assert np.array_equal(o["idx"].cpu().numpy(), g["all_idx"])
            assert np.array_equal(o["inputs_embeds"].cpu().numpy()[None], g["inputs_embeds"])
            assert np.array_equal(o["position_ids"].cpu().numpy()[:, None, :], g["position_ids"])
            assert np.array_equal(o["attention_mask"].cpu().numpy()[None], g["attention_mask"])
            if str(g["kind"]) == "image":
                ref_sel, ref_ids = osplice.splice_image(ids.numpy(), vis, g["all_idx"])
            else:
                ref_sel, ref_ids, _ = osplice.splice_video(ids.numpy(), vis, g["all_idx"])
            assert np.array_equal(o["selected_indices"].cpu().numpy(), ref_sel)
            assert np.array_equal(o["input_ids"].cpu().numpy()[None], ref_ids)


@pytest.mark.parametrize("name", ["a", "b"])
def test_topk_select_splice_matches_ov_golden(ops, golden_dir, name):
    """LLaVA-OV goldens (1-D position row; the reference's LLaVAOneVision1_5_Model_Selector.forward)."""
    g = np.load(os.path.join(golden_dir, f"ovsplice_{name}.npz"))
    n, k, d = int(g["n_visual"]), int(g["k"]), int(g["d_llm"])
    ids = torch.from_numpy(oin.make_prompt(n, int(g["n_pre"]), int(g["n_post"]), IMAGE_TOKEN, int(g["seed"])))
    L = ids.shape[1]
    emb = _embed(ids, d)[0]
    rng = np.random.default_rng(2)
    h = rng.standard_normal((n, d), dtype=np.float32)
    h[g["all_idx"]] = g["vis_embeds"]
    scores = np.zeros(n, np.float32)
    scores[g["all_idx"]] = 1.0                                       # ties among the rest: lower index first, never reached
    pos_in = (torch.arange(L) + (5 if bool(g["with_position_ids"]) else 0))[None]
    o = ops.topk_select_splice(torch.from_numpy(scores).cuda(), torch.from_numpy(h).cuda(), ids[0].cuda(), emb.cuda(), IMAGE_TOKEN,
                               [L], [n], [k], position_ids=torch.cat([pos_in, torch.arange(L)[None]]).cuda(),
                               attention_mask=torch.ones(L, dtype=torch.int64).cuda(), check=True)
    assert np.array_equal(o["inputs_embeds"].cpu().numpy()[None], g["inputs_embeds"])
    assert np.array_equal(o["position_ids"][0].cpu().numpy(), np.asarray(g["position_ids"]).reshape(-1))
    assert np.array_equal(o["position_ids"][1].cpu().numpy(), np.asarray(g["cache_position"]).reshape(-1))
    assert np.array_equal(o["attention_mask"].cpu().numpy()[None], g["attention_mask"])


@pytest.mark.parametrize("seq_lens,visual_lens,ks,d,hd,dt", [
    ([2368], [2304], [460], 3584, 1792, torch.bfloat16),                    # the reference's call: one 1344^2 image, 20 %
    ([2368], [2304], [1152], 3584, 1792, torch.bfloat16),
    ([700], [576], [115], 2048, 1024, torch.float32),
    ([5900], [5832], [1166], 4096, 2048, torch.bfloat16),                   # LLaVA-OV 8 x 729 jointly
    ([700, 2400, 1300], [576, 2304, 1100], [115, 460, 1100], 2048, 1024, torch.bfloat16),     # packed; k = N prompt
    ([4200, 64, 3000, 1025, 90, 5000, 33, 800, 640], [4096, 1, 2900, 1024, 60, 4800, 2, 700, 600],
     [819, 1, 580, 204, 12, 960, 1, 140, 120], 512, 128, torch.bfloat16),   # 9 prompts: general form even with the knob on
    ([40000], [39000], [7800], 64, 32, torch.float32),                      # > 32 768 visual tokens: general form
])
def test_lis_select_splice_equals_select_then_splice(ops, seq_lens, visual_lens, ks, d, hd, dt):
    """vsel_lis_select_splice == vsel_lis_select + vsel_splice(_batched), every output bit for bit, in both forms; the
    unfused pair is itself pinned to the reference (test_lis_gpu / the tests above), visual tokens interleaved with text."""
    from visionselector_amd import _native as N
    rng = np.random.default_rng(sum(seq_lens) + sum(ks))
    ids, _, emb, _, pos = _packed_case(rng, seq_lens, visual_lens, ks, d, dt)
    c = oin.make_case(d, hd, sum(visual_lens), 9)
    h = torch.from_numpy(c["h"]).to(dt).cuda()
    wq, bq, wk, bk = (torch.from_numpy(c[x]).bfloat16().cuda() for x in ("wq", "bq", "wk", "bk"))
    am = torch.from_numpy(rng.integers(0, 2, ids.shape[0]).astype(np.int64)).cuda()
    ids_t, emb_t, pos_t = torch.from_numpy(ids).cuda(), emb.cuda(), pos.cuda()
    out, idx, scores = ops.lis_select_varlen(h, visual_lens, ks, wq, bq, wk, bk)
    sel, new_ids, new_emb, new_pos, cu = ops.splice_batched(ids_t, emb_t, IMAGE_TOKEN, seq_lens, visual_lens, ks, idx, out,
                                                            position_ids=pos_t, check=True)
    for knob, kernel in _forms():
        with N.debug_knob("lis_splice_fused", knob):
            N.profile_start()
            o = ops.lis_select_splice(h, wq, bq, wk, bk, ids_t, emb_t, IMAGE_TOKEN, seq_lens, visual_lens, ks, position_ids=pos_t,
                                      attention_mask=am, check=True)
            prof = N.profile_stop()
        small = knob == 1 and len(seq_lens) <= 8 and max(visual_lens) <= 32768
        assert ("select_splice_small_kernel" in prof) == small and ("splice_index_seg_kernel" in prof) == (not small), prof
        assert "gather_rows_kernel" not in prof and "select_gather_small_kernel" not in prof      # no [k, D] staging
        assert torch.equal(o["idx"], idx) and torch.equal(o["scores"], scores)
        assert torch.equal(o["selected_indices"], sel) and torch.equal(o["input_ids"], new_ids)
        assert torch.equal(o["inputs_embeds"], new_emb) and torch.equal(o["position_ids"], new_pos)
        assert torch.equal(o["cu_seqlens"], cu) and torch.equal(o["attention_mask"], am[sel])


def test_lis_select_splice_permuted_presummed_and_mismatch(ops):
    """Window-ordered tokens (row maps) and producer column sums through the fused entry == the unfused entries; a prompt whose
    placeholder count disagrees with its segment raises (reference: ValueError, FT/compression_method/selector_model.py:210-213)
    and never reads out of bounds."""
    d, hd, n, k, L = 2048, 1024, 640, 128, 700
    rng = np.random.default_rng(4)
    ids, _, emb, _, pos = _packed_case(rng, [L], [n], [k], d, torch.bfloat16)
    c = oin.make_case(d, hd, n, 78)
    h = torch.from_numpy(c["h"]).bfloat16().cuda()
    wq, bq, wk, bk = (torch.from_numpy(c[x]).bfloat16().cuda() for x in ("wq", "bq", "wk", "bk"))
    l2p = torch.randperm(n, generator=torch.Generator().manual_seed(3)).cuda()
    p2l = torch.empty_like(l2p)
    p2l[l2p] = torch.arange(n, device="cuda")
    sums = h.float().sum(0, keepdim=True).contiguous()
    ids_t, emb_t, pos_t = torch.from_numpy(ids).cuda(), emb.cuda(), pos.cuda()
    out, idx, scores = ops.lis_select_presummed(h, sums, wq, bq, wk, bk, k, logical_to_physical=l2p, physical_to_logical=p2l)
    ref = ops.splice(ids_t[None], emb_t[None], IMAGE_TOKEN, idx, out, n, position_ids=pos_t[:, None, :])
    o = ops.lis_select_splice(h, wq, bq, wk, bk, ids_t, emb_t, IMAGE_TOKEN, [L], [n], [k], position_ids=pos_t, col_sums=sums,
                              logical_to_physical=l2p, physical_to_logical=p2l, check=True)
    assert torch.equal(o["idx"], idx) and torch.equal(o["scores"], scores)
    assert torch.equal(o["selected_indices"], ref[0]) and torch.equal(o["input_ids"], ref[1][0])
    assert torch.equal(o["inputs_embeds"], ref[2][0]) and torch.equal(o["position_ids"], ref[3][:, 0, :])
    bad = ids_t.clone()
    bad[(bad == IMAGE_TOKEN).nonzero()[:5, 0]] = 11                  # five placeholders short
    from visionselector_amd import _native as N
    for knob in (1, 0):
        with N.debug_knob("lis_splice_fused", knob):
            with pytest.raises(ValueError, match="do not match"):
                ops.lis_select_splice(h, wq, bq, wk, bk, bad, emb_t, IMAGE_TOKEN, [L], [n], [k], check=True)
    with pytest.raises(ValueError, match="same dtype and width"):
        ops.lis_select_splice(h, wq, bq, wk, bk, ids_t, emb_t.float(), IMAGE_TOKEN, [L], [n], [k])


def test_select_splice_argument_hygiene(ops, monkeypatch):
    """The raw pointers handed to the C-ABI are validated first: int32 / short row maps and a strided `input_ids` view
    raise, and a caller whose `max_len_out` is below a prompt's true L' is REPORTED (stats[3], hence
    ValueError under check=True) instead of leaving the uncovered rows of the outputs uninitialised."""
    d, hd = 2048, 1024
    seq_lens, visual_lens, ks = [700, 400], [640, 256], [128, 51]
    rng = np.random.default_rng(9)
    ids, _, emb, _, pos = _packed_case(rng, seq_lens, visual_lens, ks, d, torch.bfloat16)
    n = sum(visual_lens)
    c = oin.make_case(d, hd, n, 79)
    h = torch.from_numpy(c["h"]).bfloat16().cuda()
    wq, bq, wk, bk = (torch.from_numpy(c[x]).bfloat16().cuda() for x in ("wq", "bq", "wk", "bk"))
    ids_t, emb_t = torch.from_numpy(ids).cuda(), emb.cuda()
    ref = ops.lis_select_splice(h, wq, bq, wk, bk, ids_t, emb_t, IMAGE_TOKEN, seq_lens, visual_lens, ks, check=True)
    strided = torch.stack([ids_t, ids_t + 1], dim=1)[:, 0]                # stride 2 view of the same ids
    assert not strided.is_contiguous()
    with pytest.raises(RuntimeError, match="contiguous"):                 # refused loudly, never read with the wrong stride
        ops.lis_select_splice(h, wq, bq, wk, bk, strided, emb_t, IMAGE_TOKEN, seq_lens, visual_lens, ks, check=True)
    ident = torch.arange(n, device="cuda")
    with pytest.raises(TypeError, match="int64"):
        ops.lis_select_splice(h, wq, bq, wk, bk, ids_t, emb_t, IMAGE_TOKEN, seq_lens, visual_lens, ks,
                              logical_to_physical=ident.int(), physical_to_logical=ident)
    with pytest.raises(ValueError, match="one entry per token row"):
        ops.topk_select_splice(ref["scores"], h, ids_t, emb_t, IMAGE_TOKEN, seq_lens, visual_lens, ks,
                               logical_to_physical=ident[:-1])
    common = ops._select_splice_common

    def short_launch(*a, **k):
        r = list(common(*a, **k))
        r[8] = r[8] // 2                                                  # max_len_out: half the longest prompt's L'
        return tuple(r)

    monkeypatch.setattr(ops, "_select_splice_common", short_launch)
    with pytest.raises(ValueError, match="do not match"):
        ops.lis_select_splice(h, wq, bq, wk, bk, ids_t, emb_t, IMAGE_TOKEN, seq_lens, visual_lens, ks, check=True)
    with pytest.raises(RuntimeError, match="max_len_out"):              # one prompt: must be exactly its L'
        ops.lis_select_splice(h[:640], wq, bq, wk, bk, ids_t[:700], emb_t[:700], IMAGE_TOKEN, [700], [640], [128], check=True)


@pytest.mark.parametrize("n,k,L", [(2304, 460, 2368), (576, 115, 640), (256, 51, 300), (4096, 819, 4200), (5000, 1000, 5100), (640, 640, 700)])
def test_select_splice_soft_outputs_equal_soft_topk(ops, n, k, L):
    """soft=True: the soft top-k the reference's eval forward publishes as last_combined_scores (EV/token_compression/
    selector_model.py:190) comes out of the select-splice call -- one extra workgroup of the same launch up to 4096 tokens, own
    launch beyond -- and equals vsel_soft_topk_fwd on the same scores bit for bit, in both forms of the call; nothing else
    changes; k == N publishes nothing (the reference asserts 0 < k < n)."""
    from visionselector_amd import _native as N
    d, hd = 2048, 1024
    rng = np.random.default_rng(n)
    ids, _, emb, _, pos = _packed_case(rng, [L], [n], [k], d, torch.bfloat16)
    c = oin.make_case(d, hd, n, 80)
    h = torch.from_numpy(c["h"]).bfloat16().cuda()
    wq, bq, wk, bk = (torch.from_numpy(c[x]).bfloat16().cuda() for x in ("wq", "bq", "wk", "bk"))
    ids_t, emb_t, pos_t = torch.from_numpy(ids).cuda(), emb.cuda(), pos.cuda()
    base = ops.lis_select_splice(h, wq, bq, wk, bk, ids_t, emb_t, IMAGE_TOKEN, [L], [n], [k], position_ids=pos_t, check=True)
    for knob in (1, 0):
        with N.debug_knob("lis_splice_fused", knob):
            N.profile_start()
            o = ops.lis_select_splice(h, wq, bq, wk, bk, ids_t, emb_t, IMAGE_TOKEN, [L], [n], [k], position_ids=pos_t, check=True,
                                      soft=True)
            prof = N.profile_stop()
        for key in ("idx", "scores", "selected_indices", "input_ids", "inputs_embeds", "position_ids"):
            assert torch.equal(o[key], base[key]), key
        if k == n:
            assert o["soft_ps"] is None and "soft_topk_fwd_kernel" not in prof
            continue
        ps, ts = ops.soft_topk_fwd(o["scores"][None], k)
        assert torch.equal(o["soft_ps"], ps[0]) and torch.equal(o["soft_ts"], ts)
        assert ("soft_topk_fwd_kernel" in prof) == (knob == 0 or n > 4096), prof.keys()      # inside the launch when it can be
        assert abs(float(ps.sum()) - k) <= 1e-2
