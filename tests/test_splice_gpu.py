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
