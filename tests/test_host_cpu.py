"""Host-side logic on CPU: module API / state-dict compatibility, curriculum, splice index algebra vs the goldens,
monkeypatch call shapes, and the data-parallel gradient exchange over a world_size-2 gloo group."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import inputs as oin
from oracle import lis as olis

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IMAGE_TOKEN, VIDEO_TOKEN = 151655, 151656


def test_scorer_module_matches_reference_layout():
    from visionselector_amd.selector import TransformerScorer
    torch.manual_seed(0)
    m = TransformerScorer(in_features=3584)                      # reference default hidden_dim=1792
    assert (m.in_features, m.hidden_dim) == (3584, 1792)
    sd = m.state_dict()
    assert sorted(sd) == ["k_proj.bias", "k_proj.weight", "q_proj.bias", "q_proj.weight"]
    assert sd["q_proj.weight"].shape == (1792, 3584) and sd["k_proj.bias"].shape == (1792,)
    assert sum(p.numel() for p in m.parameters()) == 12_848_640   # README.md:47 "12.85M"
    assert float(sd["q_proj.bias"].abs().max()) == 0.0 and float(sd["k_proj.bias"].abs().max()) == 0.0
    assert 0.5e-4 < float(sd["q_proj.weight"].std()) < 1.5e-4     # init_scale 1e-4
    m2 = TransformerScorer(2048, 1024)
    m2.load_state_dict({k: torch.zeros_like(v) for k, v in m2.state_dict().items()})
    with pytest.raises(RuntimeError, match="GPU only"):           # no CPU fallback
        m2(torch.zeros(1, 4, 2048))


def test_dropin_import_paths():
    sys.path.insert(0, os.path.join(ROOT, "visionselector_amd", "dropin"))
    try:
        from compression_method.selector_model import (TopK, _find_ts, qwen25vl_generation_forward_selector,  # noqa: F401
                                                       qwen25vl_vision_tower_forward_selector, topk)
        from compression_method.selector_scorer import TransformerScorer  # noqa: F401
        from token_compression.monkeypatch import replace_qwen25vl
        from token_compression.selector_model import (Qwen2_5_VisionTransformerPretrainedModel_Selector,  # noqa: F401
                                                      Qwen2_5_VLForConditionalGeneration_Selector)
        from compression_method.monkeypatch import replace_llavaov15
        from compression_method.selector_model import (llavaov15_generation_forward_selector,  # noqa: F401
                                                       llavaov15_vision_tower_forward_selector,
                                                       llavaov15_vlmodel_forward_selector)
        from compression_method.modeling_selector import make_llavaov15_selector_classes  # noqa: F401
    finally:
        sys.path.pop(0)
    sentinel = object()
    assert replace_qwen25vl(None, sentinel, "selector") is sentinel       # reference: no branch for selector
    assert replace_llavaov15(None, sentinel, "selector") is sentinel
    with pytest.raises(NotImplementedError):
        replace_qwen25vl(None, sentinel, "fastv")
    with pytest.raises(ValueError):
        replace_qwen25vl(None, sentinel, "nonsense")


def test_curriculum_weight_matches_oracle():
    from visionselector_amd.selector import curriculum_weight
    for step, total in [(0, 100), (1, 3), (50, 100), (100, 100), (170, 100), (5, 0), (5, -1)]:
        assert curriculum_weight(step, total, 0.1, 2.0) == olis.curriculum_weight(step, total, 0.1, 2.0)
    assert curriculum_weight(10, 10) == pytest.approx(3.0)        # class defaults 0.1 -> 3.0 (train_qwen_selector.py:61)


def _embed(ids, d_llm):
    ar = torch.arange(d_llm, dtype=torch.int64)
    return ((ids[..., None] * 31 + ar * 17) % 257).float() / 257.0


def test_shard_units_round_robin():
    from visionselector_amd.ddp import shard_units
    got = sorted(i for r in range(8) for i in shard_units(37, r, 8))
    assert got == list(range(37))
    assert list(shard_units(10, 3, 4)) == [3, 7]


# ---------------------------------------------------------------------------------------------------
# multi-rank gradient exchange (gloo, world_size 2)
# ---------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from visionselector_amd.ddp import LisGradSync
        from visionselector_amd.selector import TransformerScorer
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        m = TransformerScorer(64, 32, init_scale=0.02)
        sync = LisGradSync(m.parameters())
        assert sync.numel == 2 * (64 * 32 + 32)
        sync.broadcast_parameters(0)
        g = torch.Generator().manual_seed(7 + rank)
        for p in m.parameters():
            p.grad = torch.randn(p.shape, generator=g)
        if rank == 1:
            m.q_proj.bias.grad = None                      # a rank without a grad contributes zeros
        sync.sync()
        torch.save({"params": [p.detach().clone() for p in m.parameters()],
                    "grads": [p.grad.clone() for p in m.parameters()]}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_lis_grad_sync_gloo_world2(tmp_path):
    world = 2
    mp.spawn(_ddp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                           # broadcast from rank 0
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)                           # identical after the all-reduce
    # expected mean of the per-rank grads (single-process accumulation)
    from visionselector_amd.selector import TransformerScorer
    shapes = [p.shape for p in TransformerScorer(64, 32).parameters()]
    names = [n for n, _ in TransformerScorer(64, 32).named_parameters()]
    per_rank = []
    for rank in range(world):
        g = torch.Generator().manual_seed(7 + rank)
        per_rank.append([torch.randn(s, generator=g) for s in shapes])
    per_rank[1][names.index("q_proj.bias")] = torch.zeros(32)
    for i, got in enumerate(r0["grads"]):
        exp = (per_rank[0][i] + per_rank[1][i]) / world
        assert torch.allclose(got, exp, atol=1e-7)


def _ddp_view_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from visionselector_amd.ddp import LisGradSync
        torch.manual_seed(3)
        lin = torch.nn.Linear(8, 4)                        # stands in for the scorer (its forward needs the GPU library)
        sync = LisGradSync(lin.parameters(), bucket_view=True)
        views = sync.views()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(lin.parameters(), views))
        x = torch.full((2, 8), float(rank + 1))
        for _ in range(2):                                 # two micro-batches accumulate IN PLACE into the bucket views
            lin(x).sum().backward()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(lin.parameters(), views))
        sync.sync()                                        # one all-reduce on the bucket, no copies
        g1 = [p.grad.clone() for p in lin.parameters()]
        lin.zero_grad(set_to_none=True)                    # a caller that drops the views: sync() re-attaches them
        lin(x).sum().backward()
        sync.sync()
        g2 = [p.grad.clone() for p in lin.parameters()]
        sync.zero_grads()
        assert all(float(p.grad.abs().max()) == 0.0 for p in lin.parameters())
        torch.save({"g1": g1, "g2": g2}, os.path.join(out_dir, f"v{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _factor_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from visionselector_amd.ddp import LisFactorSync
        from visionselector_amd.selector import TransformerScorer
        torch.manual_seed(0)
        m = TransformerScorer(48, 24, init_scale=0.02)
        params = (m.q_proj.weight, m.q_proj.bias, m.k_proj.weight, m.k_proj.bias)
        sync = LisFactorSync(params)
        assert sync.row == 2 * (24 + 48) + 2 * 24
        g = torch.Generator().manual_seed(50 + rank)
        for _ in range(3):                                 # three micro-batches per rank
            sync.add(torch.randn(sync.row, generator=g))
        sync.sync()
        torch.save([p.grad.clone() for p in params], os.path.join(out_dir, f"f{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_lis_factor_sync_gloo_world2(tmp_path):
    """Rank-1-factor exchange: all-gather of the per-micro-batch payload rows, dense mean gradient rebuilt on every rank ==
    the mean over ranks of the summed dense gradients (what LisGradSync's all-reduce leaves)."""
    world, hd, d = 2, 24, 48
    mp.spawn(_factor_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "f0.pt")
    r1 = torch.load(tmp_path / "f1.pt")
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)
    exp = [torch.zeros(hd, d, dtype=torch.float64), torch.zeros(hd, dtype=torch.float64),
           torch.zeros(hd, d, dtype=torch.float64), torch.zeros(hd, dtype=torch.float64)]
    for rank in range(world):
        g = torch.Generator().manual_seed(50 + rank)
        for _ in range(3):
            row = torch.randn(2 * (hd + d) + 2 * hd, generator=g).double()
            a, gx, dk, xs, dbq, dbk = torch.split(row, [hd, d, hd, d, hd, hd])
            exp[0] += torch.outer(a, gx)
            exp[1] += dbq
            exp[2] += torch.outer(dk, xs)
            exp[3] += dbk
    for got, e in zip(r0, exp):
        assert torch.allclose(got.double(), e / world, rtol=1e-5, atol=1e-5)


def test_lis_grad_sync_bucket_view_gloo_world2(tmp_path):
    """bucket_view: p.grad are views of the flat bucket; the data-parallel mean is one all-reduce with no pack / unpack."""
    world = 2
    mp.spawn(_ddp_view_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "v0.pt"), torch.load(tmp_path / "v1.pt")
    for key in ("g1", "g2"):
        for a, b in zip(r0[key], r1[key]):
            assert torch.equal(a, b)
    # d(sum(Wx + b))/dW = sum over the batch of x; rank r feeds x = r + 1, two rows, two micro-batches
    exp_w1 = torch.full((4, 8), (2 * 2 * 1.0 + 2 * 2 * 2.0) / 2)
    exp_b1 = torch.full((4,), (2 * 2 + 2 * 2) / 2.0)
    assert torch.allclose(r0["g1"][0], exp_w1) and torch.allclose(r0["g1"][1], exp_b1)
    assert torch.allclose(r0["g2"][0], exp_w1 / 2) and torch.allclose(r0["g2"][1], exp_b1 / 2)
    with pytest.raises(TypeError):
        from visionselector_amd.ddp import LisGradSync
        LisGradSync(torch.nn.Linear(2, 2).bfloat16().parameters(), bucket_view=True)


# ---- LisTrainer(exchange="factors") vs the dense bucket, world 2 ------------------------------------------------------
class _CpuLisFunction(torch.autograd.Function):
    """TEST stand-in for selector._LisTrainFunction on the host (the product's forward needs the GPU library): collapsed scores
    and the closed-form backward of SURVEY.md section 7 hard part 4, leaving either dense gradients or -- when
    selector.factor_sink is active -- one payload row a | gx | dk | xsum | dbq | dbk, exactly as the HIP backward does."""

    @staticmethod
    def forward(ctx, h, wq, bq, wk, bk):
        hd = wq.shape[0]
        kbar = wk @ h.mean(0) + bk
        ctx.save_for_backward(h, wq, bq, kbar)
        return (h @ (wq.t() @ kbar) + bq @ kbar) / hd ** 0.5

    @staticmethod
    def backward(ctx, g):
        from visionselector_amd import selector
        h, wq, bq, kbar = ctx.saved_tensors
        n, hd = h.shape[0], wq.shape[0]
        a = kbar / hd ** 0.5
        gx = g @ h
        dk = (wq @ gx + bq * g.sum()) / (n * hd ** 0.5)
        xsum = h.sum(0)
        dbq, dbk = a * g.sum(), n * dk
        sink = selector.active_factor_sink()
        if sink is not None:
            sink.add(torch.cat([a, gx, dk, xsum, dbq, dbk]))
            return None, None, None, None, None
        return None, torch.outer(a, gx), dbq, torch.outer(dk, xsum), dbk


class _CpuLisModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        from visionselector_amd.selector import TransformerScorer
        self.visual = torch.nn.Module()
        self.visual.importance_scorer = TransformerScorer(48, 24, init_scale=0.05)
        self.frozen = torch.nn.Linear(4, 4)
        self.regularization_weight = 0.0

    def forward(self, h, c):
        sc = self.visual.importance_scorer
        s = _CpuLisFunction.apply(h, sc.q_proj.weight, sc.q_proj.bias, sc.k_proj.weight, sc.k_proj.bias)
        loss = (c * s).sum() + self.regularization_weight * (s ** 2).mean()
        return types.SimpleNamespace(loss=loss)


def _trainer_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from visionselector_amd.trainer import LisTrainer
        res = {}
        for mode in ("dense", "factors"):
            torch.manual_seed(11 + rank)                   # different init per rank: the trainer broadcasts rank 0's scorer
            model = _CpuLisModel()
            with torch.no_grad():
                for p in model.visual.importance_scorer.parameters():
                    if p.dim() == 1:
                        p.normal_(std=0.05)
            tr = LisTrainer(model, max_steps=4, lr=1e-2, exchange=mode, log=lambda *_: None)
            g = torch.Generator().manual_seed(900 + rank)  # each rank its own micro-batches
            grads = []
            for step in range(3):
                batches = [dict(h=torch.randn(30 + 5 * mb, 48, generator=g), c=torch.randn(30 + 5 * mb, generator=g))
                           for mb in range(2)]
                tr.train_step(batches)
                grads.append([p.grad.detach().clone().float() for p in tr.params])
            res[mode] = {"grads": grads, "params": {n: p.detach().clone() for n, p in model.named_parameters() if "scorer" in n}}
        torch.save(res, os.path.join(out_dir, f"t{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_lis_trainer_factor_exchange_matches_dense_gloo_world2(tmp_path):
    """LisTrainer(exchange="factors") (57 KB payload rows, all-gather, local rebuild) takes the same optimizer steps as the
    dense bucket all-reduce, to fp32 rounding, over three steps of two micro-batches on two ranks; ranks stay identical."""
    world = 2
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"t{i}.pt") for i in range(world)]
    for mode in ("dense", "factors"):
        for n in r[0][mode]["params"]:
            assert torch.equal(r[0][mode]["params"][n], r[1][mode]["params"][n]), (mode, n)
    for step in range(3):
        for gd, gf in zip(r[0]["dense"]["grads"][step], r[0]["factors"]["grads"][step]):
            assert torch.allclose(gd, gf, rtol=1e-4, atol=1e-6 * float(gd.abs().max()) + 1e-9)   # post-clip gradients
    for n, pd in r[0]["dense"]["params"].items():
        pf = r[0]["factors"]["params"][n]
        assert torch.allclose(pd, pf, rtol=1e-4, atol=1e-5), n
        assert not torch.equal(pd, torch.zeros_like(pd))


def test_lis_factor_sync_buffer_growth_and_device_spelling():
    """new_row: grows only when full ('cuda' vs 'cuda:0' used to double the buffer on every call)."""
    from visionselector_amd.ddp import LisFactorSync
    from visionselector_amd.selector import TransformerScorer
    m = TransformerScorer(16, 8)
    sync = LisFactorSync((m.q_proj.weight, m.q_proj.bias, m.k_proj.weight, m.k_proj.bias))
    caps = []
    for step in range(5):
        for _ in range(3):
            sync.new_row("cpu").zero_()
            caps.append(sync._buf.shape[0])
        sync.sync()
        sync.zero_grads()
    assert max(caps) == 4                                   # three rows per step never outgrow the initial four
    rows = [sync.new_row(torch.device("cpu")) for _ in range(9)]
    for i, r_ in enumerate(rows[:4]):
        r_.fill_(float(i))
    assert sync._buf.shape[0] == 16 and sync._n == 9
    assert LisFactorSync._same_device(torch.device("cpu"), torch.device("cpu"))
    assert not LisFactorSync._same_device(torch.device("cpu"), torch.device("cuda", 0))


def _count_mismatch_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from visionselector_amd.ddp import LisFactorSync
        from visionselector_amd.selector import TransformerScorer
        m = TransformerScorer(16, 8)
        sync = LisFactorSync((m.q_proj.weight, m.q_proj.bias, m.k_proj.weight, m.k_proj.bias))
        for _ in range(2 + rank):                            # rank 1 adds one row more
            sync.add(torch.zeros(sync.row))
        try:
            sync.sync()
            msg = "no error"
        except RuntimeError as e:
            msg = str(e)
        open(os.path.join(out_dir, f"c{rank}.txt"), "w").write(msg)
    finally:
        dist.destroy_process_group()


def test_lis_factor_sync_row_count_mismatch_raises(tmp_path):
    mp.spawn(_count_mismatch_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r_ in range(2):
        assert "different numbers of micro-batch rows: [2, 3]" in (tmp_path / f"c{r_}.txt").read_text()


def test_eval_time_log_reader(tmp_path):
    """The EVAL_TIME lines our *_Selector.forward / timed_generate print are averaged like qwen-evaluation/extract_time.py."""
    from visionselector_amd.evaltime import parse_log, summarize_log
    log = tmp_path / "log_eval.log"
    log.write_text("\n".join([
        "using selector",
        "Input visual token number is: 2304", "Generation prefill time is: 15.5", "Generation latency time is: 120.25",
        "after generation memory: 17179869184",
        "noise Input visual token number is: 576", "Generation prefill time is: 4.5", "Generation latency time is: 80.75",
        "after generation memory: 8589934592", "Generation prefill time is: 0.0",            # zero values are dropped
    ]))
    s = summarize_log(str(log))
    assert s["samples"] == 2 and s["avg_prefill_ms"] == 10.0 and s["avg_latency_ms"] == 100.5
    assert s["avg_visual_tokens"] == 1440.0 and s["avg_max_memory_GB"] == 12.0
    assert parse_log(["nothing here"]) == {"memory_bytes": [], "latency_ms": [], "prefill_ms": [], "visual_tokens": []}


def test_flash_attn_shim_helpers_cpu():
    """dropin/flash_attn: module paths the reference imports resolve, and the padding / rotary helpers (pure index and
    elementwise torch ops) behave like flash-attn 2.7's."""
    sys.path.insert(0, os.path.join(ROOT, "visionselector_amd", "dropin"))
    try:
        for m in [k for k in sys.modules if k == "flash_attn" or k.startswith("flash_attn.")]:
            del sys.modules[m]
        from flash_attn import flash_attn_func, flash_attn_varlen_func  # noqa: F401
        from flash_attn.bert_padding import index_first_axis, pad_input, unpad_input
        from flash_attn.flash_attn_interface import flash_attn_varlen_func as f2
        from flash_attn.layers.rotary import apply_rotary_emb
        assert f2 is flash_attn_varlen_func
    finally:
        sys.path.pop(0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 4, generator=g)
    mask = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [1, 0, 0, 0, 0]])
    tok, idx, cu, mx, lens = unpad_input(x, mask)
    assert tok.shape == (9, 4) and cu.tolist() == [0, 3, 8, 9] and cu.dtype == torch.int32 and mx == 5 and lens.tolist() == [3, 5, 1]
    assert torch.equal(tok, x.reshape(15, 4)[idx]) and torch.equal(index_first_axis(x.reshape(15, 4), idx), tok)
    back = pad_input(tok, idx, 3, 5)
    assert torch.equal(back, x * mask[..., None])
    # rotary: matches the textbook rotation on the first rotary_dim features, both layouts
    xs = torch.randn(2, 6, 3, 16, generator=g)
    ang = torch.randn(6, 4, generator=g)
    cos, sin = ang.cos(), ang.sin()
    out = apply_rotary_emb(xs, cos, sin)
    x1, x2 = xs[..., :4], xs[..., 4:8]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    assert torch.allclose(out[..., :4], x1 * c - x2 * s, atol=1e-6) and torch.allclose(out[..., 4:8], x1 * s + x2 * c, atol=1e-6)
    assert torch.equal(out[..., 8:], xs[..., 8:])
    outi = apply_rotary_emb(xs, cos, sin, interleaved=True)
    assert torch.allclose(outi[..., 0:8:2], xs[..., 0:8:2] * c - xs[..., 1:8:2] * s, atol=1e-6)
    with pytest.raises(NotImplementedError):
        flash_attn_varlen_func(xs, xs, xs, cu, cu, 5, 5, dropout_p=0.1)


# ---- bench.py launcher (python bench.py --gpus N must start N ranks by itself) -------------------------------------
def test_bench_spawn_command_is_the_driver_form():
    import bench
    cmd = bench.spawn_command(4, ["--gpus", "4", "--steps", "3"], port=29517)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    # a free port is picked when none is given
    auto = bench.spawn_command(2, [])
    assert int(auto[auto.index("--master-port") + 1]) > 0


def test_bench_gpus_without_enough_devices_exits_loudly():
    import subprocess
    import sys as _sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([_sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "only" in r.stderr and "GPU" in r.stderr


def test_bench_spawn_ranks_propagates_rank_exit_code(tmp_path, monkeypatch):
    """spawn_ranks returns the launcher's exit code: a rank that dies makes `python bench.py --gpus N` exit non-zero."""
    import bench
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\nsys.exit(7 if os.environ.get('RANK') == '1' else 0)\n")
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 2)
    real = bench.spawn_command

    def fake(n, argv, port=None):
        cmd = real(n, argv, port)
        cmd[cmd.index(os.path.join(ROOT, "bench.py"))] = str(script)
        return cmd

    monkeypatch.setattr(bench, "spawn_command", fake)
    assert bench.spawn_ranks(2, []) != 0
    script.write_text("import sys\nsys.exit(0)\n")
    assert bench.spawn_ranks(2, []) == 0


def test_every_kernel_of_the_lis_step_has_a_byte_model_in_bench():
    """bench.py builds `roofline` from the kernel the HIP-event clock names: every kernel name the LIS inference path can mark
    (VSEL_AFTER_LAUNCH in csrc/lis_kernels.h / lis_small.h) must be priced by bench.kernel_bytes, or `roofline.frac` could be empty again
    (round 5's driver run: frac = None under `pytest -x` hid 201 parity tests)."""
    import re
    import importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    names = set()
    for f in ("lis_kernels.h", "lis_small.h"):
        names |= set(re.findall(r'VSEL_AFTER_LAUNCH\(\s*\w+\s*,\s*"(\w+)"\s*\)', open(os.path.join(root, "visionselector_amd", "csrc", f)).read()))
    assert {"colsum_seg_kernel", "score_kernel", "gather_rows_kernel", "topk_select_kernel", "w_finish_kernel"} <= names
    for nm in sorted(names):
        kb = bench.kernel_bytes(nm, 8, 2304, 3584, 1792, 460)
        assert kb is not None and kb > 0, f"bench.kernel_bytes has no model for {nm}"
