"""bench.py contract on the GPU box: one JSON line with the roofline / parity objects, and the multi-GPU launch path
(self-spawn under torch.distributed.run, one rank per GPU over RCCL) with whatever torch.cuda.device_count() offers."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run_bench(*args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_one_gpu_small():
    res = _run_bench("--steps", "3", "--warmup", "1", "--images", "8", "--no-cpu-baseline", "--no-attn", "--no-llm", "--no-train")
    assert res["metric"] == "vision_tokens_scored_selected_per_sec" and res["n_gpus"] == 1 and res["steps"] == 3
    rf = res["roofline"]
    # frac is ALWAYS a number (round 5's driver run died here on None: the clock had named a helper kernel without a byte model); at 8
    # images the step is host-bound and tokens re-read out of the Infinity Cache can beat the HBM figure, so only the sign is pinned here
    assert res["value"] > 0 and rf["bound"] == "hbm" and isinstance(rf["frac"], float) and rf["frac"] > 0
    assert isinstance(rf["host_bound_step"], bool)
    for name, row in res["kernels"].items():
        assert row["algorithmic_bytes_per_launch"], f"kernel {name} of the step has no byte model in bench.kernel_bytes"
        assert row["avg_us"] > 0
    p = res["parity"]
    assert p["idx_equal_fp64_oracle"] and p["idx_equal_own_scores"] and p["gather_exact"]


def test_bench_self_spawns_one_rank_per_gpu_over_rccl():
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU visible: the multi-rank RCCL path needs >= 2")
    n = 2 if n < 4 else 4
    res = _run_bench("--gpus", str(n), "--steps", "3", "--warmup", "1", "--images", "8", "--no-cpu-baseline", "--no-attn",
                     "--no-llm")
    assert res["n_gpus"] == n and res["scaling"] == "weak" and res["value"] > 0
    tr = res["train_step"]["packed_16x64_tokens"]
    assert res["train_step"]["world"] == n and tr["allreduce_busbw_GBps"] is not None and tr["allreduce_busbw_GBps"] > 0


def _rccl_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from visionselector_amd.ddp import LisGradSync
    ps = [torch.nn.Parameter(torch.zeros(8, 16, device="cuda")), torch.nn.Parameter(torch.zeros(8, device="cuda"))]
    sync = LisGradSync(ps, None, bucket_view=True)
    sync.zero_grads()
    for p in ps:
        p.grad.fill_(float(rank + 1))
    sync.sync()
    torch.cuda.synchronize()
    want = sum(range(1, world + 1)) / world
    ok = all(bool(torch.allclose(p.grad, torch.full_like(p.grad, want))) for p in ps)
    # the rank-1-factor exchange: all-gather of payload rows over RCCL, dense rebuild by vsel_lis_factors_to_grads
    from visionselector_amd.ddp import LisFactorSync
    hd, d = 8, 16
    fp = [torch.nn.Parameter(torch.zeros(hd, d, device="cuda")), torch.nn.Parameter(torch.zeros(hd, device="cuda")),
          torch.nn.Parameter(torch.zeros(hd, d, device="cuda")), torch.nn.Parameter(torch.zeros(hd, device="cuda"))]
    fs = LisFactorSync(fp)
    g = torch.Generator().manual_seed(1000 + rank)
    fs.add(torch.randn(fs.row, generator=g).cuda())
    fs.sync()
    torch.cuda.synchronize()
    exp = torch.zeros(hd, d, dtype=torch.float64)
    for r in range(world):
        row = torch.randn(fs.row, generator=torch.Generator().manual_seed(1000 + r)).double()
        exp += torch.outer(row[:hd], row[hd:hd + d])
    ok = ok and bool(torch.allclose(fp[0].grad.double().cpu(), exp / world, rtol=1e-5, atol=1e-6))
    if rank == 0:
        out.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_lis_grad_sync_over_rccl_all_ranks():
    """ddp.LisGradSync (one flat fp32 bucket, one all-reduce) over RCCL with one process per visible GPU."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU visible: RCCL with > 1 rank needs >= 2")
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    mp.spawn(_rccl_worker, args=(n, port, q), nprocs=n, join=True)
    assert q.get(timeout=60)
