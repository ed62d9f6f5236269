#!/usr/bin/env python3
"""bench.py -- vision tokens scored+selected per second on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (fused LIS score + hard top-k + gather, `vsel_lis_select`) over one
batch of synthetic visual tokens H[B, N, D] that is already resident in HBM.  Workload = BASELINE.json
configs[1]: Qwen2.5-VL-7B geometry (D=3584, Hd=1792), N_vis=2304 (1344x1344), 20 % retain (k=460).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--images B] [--budget r]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: images are independent units -> each rank processes its own B images,
no data-path collective ("weak" scaling); only the barrier and the max-over-ranks of the time use RCCL.

Prints ONE JSON line on rank 0 (see the keys at the bottom).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
D, HD, N_VIS = 3584, 1792, 2304


def algorithmic_bytes(b, n, d, hd, k, e=2):
    """tokens read once + kept rows written + weights once + scores + indices."""
    return b * n * d * e + b * k * d * e + 2 * hd * d * e + 2 * hd * e + b * n * 4 + b * k * 8


def kernel_bytes(name, b, n, d, hd, k, e=2):
    """Algorithmic bytes of ONE launch of a named kernel of the LIS step ('kernels'); b = images that launch covers.
    The three streaming kernels carry SURVEY.md 8(d)'s terms.  The projection helpers between the two sweeps (a few us each,
    launch-bound) are priced at their compulsory traffic -- operands read once, results written once, split-K partials not counted --
    so that EVERY kernel of the step has a byte model and `roofline.frac` is a number whichever kernel the clock names."""
    planes = 3 * 2                                                  # bf16x3 operand planes (csrc/proj_bf16x3.h): 3 x 2 B per element
    table = {
        "colsum_partial_kernel": b * n * d * e,                    # first sweep of the tokens
        "colsum_seg_kernel": b * n * d * e,                        # ... in its many-segments form (one wave per image slab)
        "score_kernel": b * n * d * e + b * n * 4,                 # second sweep + scores out
        "gather_rows_kernel": 2 * b * k * d * e + b * k * 8,       # read kept rows + write them
        "topk_select_kernel": b * n * 4 + b * k * 8,               # scores in, ascending indices out
        # projections between the sweeps, bf16x3 MFMA form (weights streamed once per launch)
        "colsum_finish_split_kernel": b * d * 4 + b * d * planes,
        "gemm_nt_bf16x3_kernel": hd * d * e + b * d * planes + b * hd * 4,
        "kbar_finish_split_kernel": b * hd * 4 + 2 * hd * e + b * hd * planes + b * hd * 4,
        "gemm_nn_bf16x3_kernel": hd * d * e + b * hd * planes + b * d * 4,
        "w_finish_kernel": 2 * b * d * 4,
        # ... generic fp32-input form
        "colsum_finish_kernel": 2 * b * d * 4,
        "gemm_nt_kernel": hd * d * e + b * d * 4 + b * hd * 4,
        "kbar_finish_kernel": 2 * b * hd * 4 + 2 * hd * e,
        "gemm_nn_kernel": hd * d * e + b * hd * 4 + b * d * 4,
        "slice_sum_kernel": 2 * b * d * 4,
        # small-batch form (csrc/lis_small.h)
        "score_small_kernel": b * n * d * e + b * n * 4,
        "select_gather_small_kernel": 2 * b * k * d * e + b * k * 8,
        "proj_nt_small_kernel": hd * d * e,                        # Wk streamed once
        "proj_nn_small_kernel": hd * d * e,                        # Wq streamed once
    }
    return table.get(name)


def cpu_baseline(budget, seconds_budget=14.0):
    """The reference formulation on the host cores (oracle/lis_torch.py = same ATen ops as the reference).
    The thread count is swept (all cores is NOT the fastest on a many-core host for one 2304-token image)
    and the best configuration is reported; `cores` = the threads of that configuration."""
    from oracle import lis_torch
    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(1234)
    h = torch.randn(N_VIS, D, generator=g).bfloat16().float()
    wq = (0.02 * torch.randn(HD, D, generator=g)).bfloat16().float()
    wk = (0.02 * torch.randn(HD, D, generator=g)).bfloat16().float()
    bq = (0.02 * torch.randn(HD, generator=g)).bfloat16().float()
    bk = (0.02 * torch.randn(HD, generator=g)).bfloat16().float()
    cands = sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu} or {ncpu})
    best = None
    per = seconds_budget / len(cands)
    total = 0
    for threads in cands:
        torch.set_num_threads(threads)
        lis_torch.select_forward(h, wq, bq, wk, bk, budget)
        times = []
        t_end = time.perf_counter() + per
        while len(times) < 20 and (time.perf_counter() < t_end or len(times) < 2):
            t0 = time.perf_counter()
            lis_torch.select_forward(h, wq, bq, wk, bk, budget)
            times.append(time.perf_counter() - t0)
        total += len(times)
        if best is None or min(times) < best[0]:
            best = (min(times), threads, sorted(times)[len(times) // 2])
    # honesty line: the collapsed algebra on the same host cores (what a CPU port of OUR formulation would do)
    torch.set_num_threads(best[1])
    lis_torch.select_forward_collapsed(h, wq, bq, wk, bk, budget)
    ct = []
    for _ in range(10):
        t0 = time.perf_counter()
        lis_torch.select_forward_collapsed(h, wq, bq, wk, bk, budget)
        ct.append(time.perf_counter() - t0)
    # the reference's own dtype on a GPU is bf16; the same ATen ops in bf16 on the host cores (SURVEY.md section 8d asks for both)
    bt = []
    try:
        hb, wqb, bqb, wkb, bkb = (t.bfloat16() for t in (h, wq, bq, wk, bk))
        lis_torch.select_forward(hb, wqb, bqb, wkb, bkb, budget)
        for _ in range(5):
            t0 = time.perf_counter()
            lis_torch.select_forward(hb, wqb, bqb, wkb, bkb, budget)
            bt.append(time.perf_counter() - t0)
    except Exception:
        bt = []
    return {"value": N_VIS / best[0], "unit": "tokens/s", "cores": best[1], "kind": "port", "host_cpus": ncpu,
            "bf16_reference_formulation": ({"tokens_per_s": N_VIS / min(bt), "ms_per_image": min(bt) * 1e3, "cores": best[1]}
                                           if bt else None),
            "collapsed_formulation": {"tokens_per_s": N_VIS / min(ct), "ms_per_image": min(ct) * 1e3, "cores": best[1],
                                      "note": "same selection with the scorer algebraically collapsed (not what the reference runs)"},
            "sample": f"{total} images of N={N_VIS}, D={D}, Hd={HD} over thread counts {cands}: fp32 reference formulation "
                      f"(2 GEMMs + NxN matmul + mean + topk + sort + gather) in torch CPU, best-of at {best[1]} threads",
            "ms_per_image": best[0] * 1e3, "ms_per_image_median": best[2] * 1e3}


def spawn_command(n_gpus, argv, port=None):
    """argv of the one-process-per-GPU launch of this script on ONE node (the form the driver uses for N > 1)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n_gpus, argv):
    """Re-execute under torch.distributed.run with one rank per GPU; returns the launcher's exit code."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    return subprocess.call(spawn_command(n_gpus, argv), env=env)


def pmc_traffic(kernel, b):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json; separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if that (kernel, B) was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tab = json.load(f)
        return tab.get(str(b), {}).get(kernel, {}).get("hbm_bytes")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--images", type=int, default=128, help="images per step per GPU (B)")
    ap.add_argument("--budget", type=float, default=0.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-attn", action="store_true")
    ap.add_argument("--no-llm", action="store_true", help="skip the whole-LLM prefill leg (7B random-init weights)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step leg (config C3: fwd + bwd + grad all-reduce)")
    ap.add_argument("--no-single-sweep", action="store_true", help="skip the producer-side column-sum leg (SURVEY 8f N2)")
    ap.add_argument("--no-batch1", action="store_true", help="skip the one-image-per-call leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE config sweep / config-5 legs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (what the reference's scripts do with torchrun /
        # accelerate launch: qwen-vl-finetune/scripts/sft_7b.sh:71-74, qwen-evaluation/run_selector.sh:11-18)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    from visionselector_amd import _native, ops

    # what the collective library itself saw (world > 1): the line's n_gpus is otherwise only the launcher's WORLD_SIZE
    rccl = None
    if dist:
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        props = torch.cuda.get_device_properties(local_rank)
        me = str(getattr(props, "uuid", "")) or f"{props.name}:{getattr(props, 'pci_bus_id', local_rank)}:{local_rank}"
        ids = [None] * world
        dist.all_gather_object(ids, me)
        rccl = {"ranks_seen": int(ones.item()), "distinct_devices": len(set(ids)), "backend": dist.get_backend()}

    b, n, d, hd = args.images, N_VIS, D, HD
    k = max(1, int(n * args.budget))
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    h = torch.randn(b, n, d, device="cuda", generator=gen).bfloat16()
    wq = (0.02 * torch.randn(hd, d, device="cuda", generator=gen)).bfloat16()
    wk = (0.02 * torch.randn(hd, d, device="cuda", generator=gen)).bfloat16()
    bq = (0.02 * torch.randn(hd, device="cuda", generator=gen)).bfloat16()
    bk = (0.02 * torch.randn(hd, device="cuda", generator=gen)).bfloat16()

    def step():
        return ops.lis_select(h, wq, bq, wk, bk, k)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    # timed region: EXACTLY K steps, nothing but the hot path on the stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, idx, scores = step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    # instrumented repeat of the same K steps: libvsel records a HIP event after every kernel on the launch
    # stream (per-kernel durations for the roofline).  The events themselves cost a few us per kernel boundary,
    # so `value` comes from the un-instrumented region above; both step times are reported.
    _native.profile_start()
    tp0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    tp1 = time.perf_counter()
    prof = _native.profile_stop()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    if dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * b * n / (elapsed / args.steps)

    # ---- parity gate on the timed data (one image, fp64 collapsed oracle on the host) ----------------
    parity = None
    if rank == 0:
        import numpy as np
        from oracle import lis as olis
        f = lambda t: t.float().cpu().numpy()  # noqa: E731
        ref = olis.scorer_collapsed(f(h[0])[None], f(wq), f(bq), f(wk), f(bk))[0]
        s0 = scores[0].cpu().numpy()
        ridx = olis.hard_topk_indices(ref.astype(np.float32), k)
        parity = {"max_abs_dscore": float(np.abs(s0 - ref).max()),
                  "idx_equal_fp64_oracle": bool(np.array_equal(ridx, idx[0].cpu().numpy())),
                  "idx_equal_own_scores": bool(np.array_equal(olis.hard_topk_indices(s0, k), idx[0].cpu().numpy())),
                  "gather_exact": bool(torch.equal(out[0], h[0][idx[0]]))}

    # ---- training step (config C3), every rank: LIS fwd + bwd into one flat fp32 bucket + all-reduce over RCCL --------
    train = None
    if not args.no_train:
        try:
            train = bench_train_step(ops, dist, world, rank)
        except Exception as e:  # optional leg: never lose the headline line over it
            train = {"error": str(e)[:300]}

    if rank != 0:
        if dist:
            dist.barrier()              # rank 0 is still producing the report (attention leg); leave together
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events recorded by libvsel on the launch stream) -------
    kern = {name: {"avg_us": ms / calls * 1e3, "launches_per_step": calls / args.steps, "share": 0.0}
            for name, (ms, calls) in prof.items()}
    tot_ms = sum(ms for ms, _ in prof.values()) or 1.0
    for name, (ms, _) in prof.items():
        kern[name]["share"] = ms / tot_ms
    # the dominant kernel = most summed time AMONG the kernels kernel_bytes() prices (all of the step's are; a kernel added without
    # a byte model can therefore never leave `frac` empty), with the whole path's figure as the last resort
    for name in kern:
        bl = b / max(1.0, kern[name]["launches_per_step"])
        kb_ = kernel_bytes(name, bl, n, d, hd, k)
        kern[name]["algorithmic_bytes_per_launch"] = kb_
        kern[name]["achieved_GBps"] = (kb_ / (kern[name]["avg_us"] * 1e-6) / 1e9) if kb_ and kern[name]["avg_us"] > 0 else None
    priced = {nm: v for nm, v in prof.items() if kern[nm]["achieved_GBps"]}
    dom = max(priced.items(), key=lambda kv: kv[1][0])[0] if priced else None
    path_bytes = algorithmic_bytes(b, n, d, hd, k)
    path_gbs = path_bytes / (ms_per_step * 1e-3) / 1e9
    step_us_kernels = sum(v["avg_us"] * v["launches_per_step"] for v in kern.values())
    # per-kernel intervals are [begin mark, end mark] on the launch stream (csrc/common.hip): they hold the kernel and two marker
    # packets.  When the kernels of a step add up to well under the step's wall time the step is bound by the host's launch rate, and
    # the kernel named "dominant" is only the longest of several few-us kernels -- said on the line so nobody reads a roofline into it.
    host_bound = bool(step_us_kernels < 0.6 * ms_per_step * 1e3)
    if dom is not None:
        # (with VSEL_PIPELINE=1 libvsel cuts a call of >= 32 images into two halves: a launch then covers b / launches_per_step images)
        b_launch = b / max(1.0, kern[dom]["launches_per_step"])
        kb = kern[dom]["algorithmic_bytes_per_launch"]
        avg_s = kern[dom]["avg_us"] * 1e-6
        ach = kern[dom]["achieved_GBps"]
        roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, int(b_launch)),
                    "algorithmic_bytes_per_launch": kb, "avg_launch_us": avg_s * 1e6, "images_per_launch": b_launch,
                    "host_bound_step": host_bound, "kernel_us_per_step": step_us_kernels,
                    "note": "achieved = algorithmic bytes of one launch of this kernel / its HIP-event duration (instrumented pass: "
                            "the same launches as the product path on one stream, a begin mark in front of and an end mark behind "
                            "every kernel); traffic = bytes/launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json)"
                            + ("; HOST-BOUND step (kernels sum to < 60 % of the step): per-kernel fractions are not a roofline "
                               "statement at this batch size" if host_bound else "")}
    else:
        roofline = {"bound": "hbm", "kernel": "(whole step)", "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": path_gbs / HBM_PEAK_GBS, "traffic": None, "host_bound_step": host_bound,
                    "note": "no per-kernel timing available: SURVEY 8(d) path bytes / step time"}
    path = {"algorithmic_bytes_per_step": path_bytes, "achieved_GBps": path_gbs, "frac_of_8TBps": path_gbs / HBM_PEAK_GBS}

    res = {
        "metric": "vision_tokens_scored_selected_per_sec", "value": value, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "accumulate": "f32",
        "data": "synthetic",
        "config": {"workload": "Qwen2.5-VL-7B LIS score + hard top-k + gather (vsel_lis_select), N_vis=2304, D=3584, "
                               "Hd=1792, 20% retain (k=460)" if args.budget == 0.2 else
                               f"Qwen2.5-VL-7B LIS select, N_vis=2304, D=3584, Hd=1792, budget={args.budget} (k={k})",
                   "images_per_step_per_gpu": b, "n_vis": n, "d": d, "hd": hd, "k": k, "budget": args.budget,
                   "sharding": f"{world} independent replicas, one process per GPU, no data-path collective"},
        "ms_per_step_instrumented": (tp1 - tp0) / args.steps * 1e3,
        "roofline": roofline, "roofline_path": path, "kernels": kern, "parity": parity,
    }

    if rccl is not None:
        res["rccl"] = rccl
    if train is not None:
        res["train_step"] = train
    # ---- single-sweep LIS: column sums from the merger's GELU, sum production timed inside ----------
    if not args.no_single_sweep:
        try:
            res["single_sweep"] = bench_single_sweep(ops, h, wq, bq, wk, bk, k, idx, path_bytes)
        except Exception as e:  # optional leg: never lose the headline line over it
            res["single_sweep"] = {"error": str(e)[:300]}
    # ---- the reference's real evaluation call: ONE image per call (EV/token_compression/selector_model.py:182-194, assert :270) ---
    if not args.no_batch1:
        try:
            res["batch1"] = bench_batch1(ops, h, wq, bq, wk, bk, k)
        except Exception as e:  # optional leg: never lose the headline line over it
            res["batch1"] = {"error": str(e)[:300]}
    # ---- prefill attention at the compressed vs the full length (second half of the metric) ----------
    if not args.no_attn:
        try:
            res["prefill_attention"] = bench_attention(ops, k)
        except Exception as e:  # the attention kernel is optional for this line
            res["prefill_attention"] = {"error": str(e)[:200]}
    # ---- the other BASELINE.json configurations on one GPU (parity-tested cases; timed here so that the driver's line carries them) ----
    if not args.no_configs:
        del h, out, idx, scores
        torch.cuda.empty_cache()
        try:
            res["config_sweep"] = bench_config_sweep(ops)
        except Exception as e:  # optional leg
            res["config_sweep"] = {"error": str(e)[:300]}
        try:
            res["config5"] = bench_config5(ops, _native)
        except Exception as e:  # optional leg
            res["config5"] = {"error": str(e)[:300]}
    if not args.no_llm:
        try:
            res["prefill_llm"] = bench_llm_prefill(_native, k)
        except Exception as e:  # optional leg: never lose the headline line over it
            res["prefill_llm"] = {"error": str(e)[:300]}
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args.budget)
        res["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
    print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_single_sweep(ops, h, wq, bq, wk, bk, k, idx_two_sweep, path_bytes, iters=10):
    """SURVEY.md section 8f N2 on the headline workload: the merger hands the LIS the column sums of its tokens, so the first
    of the two HBM sweeps is gone.  Sum production is TIMED INSIDE: vsel_gelu_colsum replaces the merger's GELU on the
    [B N, 5120] hidden activation (Qwen2_5_VLPatchMerger, EV/qwen25vl/modeling_qwen2_5_vl.py:148-161) and
    vsel_colsum_linear carries the sums through the merger's last Linear; what the LIS is charged is
        (gelu_colsum - the same GELU kernel without sums) + colsum_linear + lis_select_presummed.
    Parity of the leg: with the tokens' true column sums the selection equals the two-sweep path's."""
    b, n, d = h.shape
    cmid = 5120                                                  # merger hidden width at 7B: 4 x 1280
    gen = torch.Generator(device="cuda").manual_seed(99)
    x = torch.randn(b * n, cmid, device="cuda", generator=gen).bfloat16()
    w2 = (0.02 * torch.randn(d, cmid, device="cuda", generator=gen)).bfloat16()
    b2 = (0.02 * torch.randn(d, device="cuda", generator=gen)).bfloat16()

    def ev_time(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    _, gsum = ops.gelu_colsum(x, b)
    cs = ops.colsum_linear(gsum, w2, b2, n)

    def chain():
        _, g_ = ops.gelu_colsum(x, b)
        c_ = ops.colsum_linear(g_, w2, b2, n)
        return ops.lis_select_presummed(h, c_, wq, bq, wk, bk, k)

    t_gelu_torch = ev_time(lambda: torch.nn.functional.gelu(x))
    t_gelu_own = ev_time(lambda: ops.gelu_colsum(x, b, sums=False))      # the SAME streaming kernel without its sums
    t_gelu_cs = ev_time(lambda: ops.gelu_colsum(x, b))
    t_lin = ev_time(lambda: ops.colsum_linear(gsum, w2, b2, n))
    t_pre = ev_time(lambda: ops.lis_select_presummed(h, cs, wq, bq, wk, bk, k))
    t_chain = ev_time(chain)
    true_sums = h.float().sum(1).contiguous()
    _, idx1, _ = ops.lis_select_presummed(h, true_sums, wq, bq, wk, bk, k)
    # What the LIS is charged: the sums' extra cost over this library's own sums-free GELU, their trip through the merger's last
    # Linear, and the presummed select.  (Round 3 subtracted torch's GELU instead, which is slower than ours by more than the sums
    # cost and credited that difference to the LIS: 0.71 of the path roofline where free sums would give 0.54.)
    charged_us = max(0.0, t_gelu_cs - t_gelu_own) + t_lin + t_pre
    del x
    return {"what": "vsel_gelu_colsum (in place of the merger's GELU) + vsel_colsum_linear + vsel_lis_select_presummed; "
                    "lis_charged_us = (gelu_colsum - the same kernel without sums) + colsum_linear + lis_select_presummed",
            "torch_gelu_us": t_gelu_torch, "gelu_without_sums_us": t_gelu_own, "gelu_colsum_us": t_gelu_cs,
            "colsum_linear_us": t_lin, "lis_select_presummed_us": t_pre, "chain_us": t_chain,
            "lis_charged_us": charged_us, "step_us": charged_us, "tokens_per_s": b * n / (charged_us * 1e-6),
            "roofline_path": {"algorithmic_bytes_per_step": path_bytes, "achieved_GBps": path_bytes / (charged_us * 1e-6) / 1e9,
                              "frac_of_8TBps": path_bytes / (charged_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
            "chain_minus_torch_gelu_us": t_chain - t_gelu_torch,
            "idx_equal_two_sweep_with_true_sums": bool(torch.equal(idx1, idx_two_sweep))}


def bench_batch1(ops, h, wq, bq, wk, bk, k, n_text=64, iters=200):
    """One image per call, what the reference's evaluation harness issues (batch 1): (a) vsel_lis_select alone and (b) everything the
    *_Selector prefill enqueues for the image -- scores, hard top-k, splice of ids / embeddings / M-RoPE positions AND the soft top-k
    that fills visual.last_combined_scores (EV :190) -- vsel_lis_select_splice with its soft outputs.  us per call, back to back
    on one stream (HIP events around `iters` calls); path roofline of (a) on the section-8(d) bytes of one image."""
    n, d = h.shape[1], h.shape[2]
    img = 151655
    h1 = h[0].contiguous()
    ids = torch.cat((torch.arange(10, 10 + n_text // 2), torch.full((n,), img), torch.arange(50, 50 + n_text - n_text // 2))).cuda()
    gen = torch.Generator(device="cuda").manual_seed(7)
    emb = torch.randn(ids.numel(), d, device="cuda", generator=gen).bfloat16()
    pos = torch.arange(ids.numel(), device="cuda").repeat(3, 1).contiguous()

    def ev_time(fn):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    t_sel = ev_time(lambda: ops.lis_select(h1, wq, bq, wk, bk, k))
    t_call = ev_time(lambda: ops.lis_select_splice(h1, wq, bq, wk, bk, ids, emb, img, [ids.numel()], [n], [k], position_ids=pos,
                                                   soft=True))
    bytes1 = algorithmic_bytes(1, n, d, wq.shape[0], k)
    return {"what": "one image per call (the reference's evaluation call): lis_select_us = vsel_lis_select; eval_call_us = "
                    "vsel_lis_select_splice with soft outputs (scores + hard top-k + splice + the soft top-k of last_combined_scores)",
            "lis_select_us": t_sel, "eval_call_us": t_call, "tokens_per_s": n / (t_sel * 1e-6),
            "roofline_path": {"algorithmic_bytes_per_call": bytes1, "achieved_GBps": bytes1 / (t_sel * 1e-6) / 1e9,
                              "frac_of_8TBps": bytes1 / (t_sel * 1e-6) / 1e9 / HBM_PEAK_GBS}}


def bench_train_step(ops, dist, world, rank, iters=20):
    """BASELINE config 3: one data-parallel training step of the LIS block per rank -- forward (scores, soft top-k, mask
    apply, constraint mask, BCE), backward (closed-form fp32 gradients written straight into ONE flat bucket) and the mean
    all-reduce of the 12 848 640 gradients over RCCL (nothing to exchange at world 1).  Curriculum weight per step as
    train_qwen_selector.py:60-92.  Max over ranks; tokens/s is the whole-job aggregate."""
    from visionselector_amd.selector import curriculum_weight
    d, hd = D, HD
    wgen = torch.Generator(device="cuda").manual_seed(4321)          # ONE scorer, replicated on every rank (data parallel:
    wq = (0.02 * torch.randn(hd, d, device="cuda", generator=wgen)).bfloat16()   # the mean of the gradients is a gradient of
    wk = (0.02 * torch.randn(hd, d, device="cuda", generator=wgen)).bfloat16()   # these weights only if they are the same)
    bq = (0.02 * torch.randn(hd, device="cuda", generator=wgen)).bfloat16()
    bk = (0.02 * torch.randn(hd, device="cuda", generator=wgen)).bfloat16()
    gen = torch.Generator(device="cuda").manual_seed(8765 + rank)    # each rank its own micro-batches
    bucket = torch.zeros(2 * (hd * d + hd), dtype=torch.float32, device="cuda")
    views, off = [], 0
    for shape in ((hd, d), (hd,), (hd, d), (hd,)):
        cnt = shape[0] * (shape[1] if len(shape) > 1 else 1)
        views.append(bucket[off:off + cnt].view(*shape))
        off += cnt
    out = {"grad_elems": bucket.numel(), "grad_bytes_fp32": bucket.numel() * 4, "world": world}
    for name, n in (("packed_16x64_tokens", 1024), ("one_image_2304_tokens", 2304)):
        k = int(n * 0.2)
        h = torch.randn(n, d, device="cuda", generator=gen).bfloat16()
        dhn = (torch.randn(n, d, device="cuda", generator=gen) / d ** 0.5).bfloat16()

        def compute(step_no):
            w = curriculum_weight(step_no, 1000, 0.1, 2.0)
            h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, k)
            ops.lis_train_bwd(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, None, w, need_dh=False, out=views)

        def exchange():
            if dist:
                dist.all_reduce(bucket)
                bucket.div_(world)

        for i in range(3):
            compute(i)
            exchange()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        for i in range(iters):
            compute(i)
            exchange()
        ev[1].record()
        for i in range(iters):
            compute(i)
        ev[2].record()
        for i in range(iters):
            exchange()
        ev[3].record()
        torch.cuda.synchronize()
        t = torch.tensor([ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])],
                         dtype=torch.float64, device="cuda") / iters * 1e3
        if dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        step_us, comp_us, exch_us = [float(x) for x in t.tolist()]
        # SURVEY.md 8(d) training bytes: forward 2 N D e (tokens in, masked tokens out) + both weight matrices; backward 2 N D e (dH', H)
        # + 2 Hd D 4 (fp32 gradients out).  ~11 dependent launches of 3 - 10 us each: launch-bound, the fraction says how far.
        tbytes = 2 * n * d * 2 + 2 * hd * d * 2 + 2 * n * d * 2 + 2 * hd * d * 4
        out[name] = {"n_tokens_per_rank": n, "k": k, "step_us": step_us, "fwd_bwd_us": comp_us, "grad_allreduce_us": exch_us,
                     "tokens_per_s": world * n / (step_us * 1e-6),
                     "roofline": {"bound": "hbm", "algorithmic_bytes": tbytes, "achieved": tbytes / (comp_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": tbytes / (comp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                  "note": "forward + backward of one micro-batch (fwd_bwd_us), SURVEY 8(d) training bytes"},
                     "allreduce_busbw_GBps": (2 * (world - 1) / world * bucket.numel() * 4 / (exch_us * 1e-6) / 1e9) if dist else None}

        # the same step with the weight gradients exchanged as rank-1 factors (ddp.LisFactorSync): the backward
        # writes one 57 KB payload row instead of two dense [Hd, D] gradients, the exchange is an all-gather of the rows and
        # every rank rebuilds the mean gradient with two [Hd, R] x [R, D] GEMMs
        from visionselector_amd.ddp import LisFactorSync
        params = [torch.nn.Parameter(t.float(), requires_grad=True) for t in (wq, bq, wk, bk)]
        fsync = LisFactorSync(params, check_counts=False)       # (every rank adds exactly one row per step here: no per-step count exchange / host sync)

        def compute_f(step_no):
            w = curriculum_weight(step_no, 1000, 0.1, 2.0)
            h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, k)
            ops.lis_train_bwd_factors(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, None, w, need_dh=False,
                                      out=fsync.new_row(h.device))

        for i in range(3):
            compute_f(i)
            fsync.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for i in range(iters):
            compute_f(i)
            fsync.sync()
        ev[1].record()
        torch.cuda.synchronize()
        tf_ = torch.tensor([ev[0].elapsed_time(ev[1])], dtype=torch.float64, device="cuda") / iters * 1e3
        if dist:
            dist.all_reduce(tf_, op=dist.ReduceOp.MAX)
        out[name]["rank1_factors"] = {"step_us": float(tf_[0]), "payload_bytes_per_rank": fsync.row * 4,
                                      "tokens_per_s": world * n / (float(tf_[0]) * 1e-6),
                                      "note": "fwd + factor backward + all-gather of the payload rows + dense rebuild on every rank"}
    return out


def bench_attention(ops, k, text=64, hq=28, hkv=4, dh=128, layers=28, iters=200):
    """Var-len causal GQA attention (Qwen2.5-VL-7B geometry) at L' = k + 64 vs L = N + 64, one sequence."""
    out = {}
    for tag, L in (("retain20", k + text), ("full", N_VIS + text)):
        gen = torch.Generator(device="cuda").manual_seed(7)
        q = torch.randn(L, hq, dh, device="cuda", generator=gen).bfloat16()
        kk = torch.randn(L, hkv, dh, device="cuda", generator=gen).bfloat16()
        v = torch.randn(L, hkv, dh, device="cuda", generator=gen).bfloat16()
        cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
        for _ in range(20):            # (a ~15 us kernel: enough launches queued that the Python call rate is not what is timed)
            ops.varlen_attn(q, kk, v, cu, L)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.varlen_attn(q, kk, v, cu, L)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        flops = 4.0 * L * L * hq * dh / 2
        tf = flops / (ms * 1e-3) / 1e12
        out[tag] = {"L": L, "ms_per_layer": ms, "ms_28_layers": ms * layers, "tflops": tf,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / MFMA_BF16_PEAK_TFLOPS, "traffic": None}}
    out["speedup"] = out["full"]["ms_per_layer"] / out["retain20"]["ms_per_layer"]
    # the same kernel where it is not latency-bound: a packed batch (16 sequences x 4096 tokens, the training shape)
    try:
        n_seq, L = 16, 4096
        gen = torch.Generator(device="cuda").manual_seed(8)
        q = torch.randn(n_seq * L, hq, dh, device="cuda", generator=gen).bfloat16()
        kk = torch.randn(n_seq * L, hkv, dh, device="cuda", generator=gen).bfloat16()
        v = torch.randn(n_seq * L, hkv, dh, device="cuda", generator=gen).bfloat16()
        cu = torch.arange(0, n_seq * L + 1, L, dtype=torch.int32, device="cuda")
        for _ in range(5):
            ops.varlen_attn(q, kk, v, cu, L)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.varlen_attn(q, kk, v, cu, L)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        tf = 4.0 * L * L * hq * dh / 2 * n_seq / (ms * 1e-3) / 1e12
        out["packed_16x4096"] = {"ms": ms, "tflops": tf,
                                 "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                              "frac": tf / MFMA_BF16_PEAK_TFLOPS, "traffic": None}}
        # ... and its backward (training): dQ, dK, dV from dO on the same batch, 5 algorithmic contractions
        # (2.5 x the forward's FLOPs; the two-pass kernels execute 7)
        do = torch.randn(n_seq * L, hq, dh, device="cuda", generator=gen).bfloat16()
        o, lse = ops.varlen_attn_fwd_lse(q, kk, v, cu, L)
        for _ in range(5):
            ops.varlen_attn_bwd(do, q, kk, v, o, lse, cu, L)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.varlen_attn_bwd(do, q, kk, v, o, lse, cu, L)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tf = 10.0 * L * L * hq * dh / 2 * n_seq / (ms * 1e-3) / 1e12
        out["packed_16x4096_backward"] = {"ms": ms, "tflops_algorithmic": tf,
                                          "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                       "frac": tf / MFMA_BF16_PEAK_TFLOPS, "traffic": None}}
    except Exception as e:  # optional
        out.setdefault("packed_16x4096", {"error": str(e)[:200]})
        out.setdefault("packed_16x4096_backward", {"error": str(e)[:200]})
    # ... and the packed batches a deployment of the compressed prompts serves (config 5): 8 / 32 prompts of L' = k + 64
    # tokens (forward), and 4 training sequences of the uncompressed length (forward + backward)
    for tag, n_seq, L, bwd in (("packed_8xLc", 8, k + text, False), ("packed_32xLc", 32, k + text, False), ("packed_4xL_train", 4, N_VIS + text, True)):
        try:
            gen = torch.Generator(device="cuda").manual_seed(9)
            q = torch.randn(n_seq * L, hq, dh, device="cuda", generator=gen).bfloat16()
            kk = torch.randn(n_seq * L, hkv, dh, device="cuda", generator=gen).bfloat16()
            v = torch.randn(n_seq * L, hkv, dh, device="cuda", generator=gen).bfloat16()
            cu = torch.arange(0, n_seq * L + 1, L, dtype=torch.int32, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timed(fn, n):
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n
            from visionselector_amd import _native
            _native.profile_start()
            ops.varlen_attn(q, kk, v, cu, L)
            kern = sorted(_native.profile_stop())
            # (best of three timed loops: at 30 - 100 us per call the host's call rate is close to the kernel time, and one slow stretch of
            # the Python loop would be booked as kernel time)
            ms = min(timed(lambda: ops.varlen_attn(q, kk, v, cu, L), 100) for _ in range(3))
            fl = 4.0 * L * L * hq * dh / 2 * n_seq
            tf_ = fl / (ms * 1e-3) / 1e12
            out[tag] = {"n_seq": n_seq, "L": L, "fwd_ms": ms, "fwd_tflops": tf_, "kernel": kern,
                        "roofline": {"bound": "mfma", "achieved": tf_, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": tf_ / MFMA_BF16_PEAK_TFLOPS, "traffic": None}}
            if bwd:
                do = torch.randn(n_seq * L, hq, dh, device="cuda", generator=gen).bfloat16()
                o, lse = ops.varlen_attn_fwd_lse(q, kk, v, cu, L)
                ms = timed(lambda: ops.varlen_attn_bwd(do, q, kk, v, o, lse, cu, L), 30)
                out[tag].update({"bwd_ms": ms, "bwd_tflops_algorithmic": 2.5 * fl / (ms * 1e-3) / 1e12})
        except Exception as e:  # optional
            out[tag] = {"error": str(e)[:200]}
    return out


def _ev_us(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def bench_config_sweep(ops, iters=10):
    """BASELINE.json configs 2 (7B, 10 / 20 / 50 % retain), 1 / 2's 3B geometry at batch and 4 (LLaVA-OV-1.5-8B, 8 x 729 tokens scored
    jointly): vsel_lis_select on resident tokens, us per step, tokens/s and the SURVEY 8(d) path bytes against 8 TB/s."""
    rows = []
    g = torch.Generator(device="cuda").manual_seed(5)
    mk = lambda *sh: (0.02 * torch.randn(*sh, device="cuda", generator=g)).bfloat16()  # noqa: E731
    for name, d, hd, n, b, budgets in (("Qwen2.5-VL-7B N=2304", 3584, 1792, 2304, 128, (0.1, 0.2, 0.5)),
                                       ("Qwen2.5-VL-3B N=576", 2048, 1024, 576, 512, (0.2,)),
                                       ("LLaVA-OV-1.5-8B 8x729 joint", 4096, 2048, 5832, 48, (0.2,))):
        h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
        wq, bq, wk, bk = mk(hd, d), mk(hd), mk(hd, d), mk(hd)
        for r in budgets:
            k = max(1, int(n * r))
            us = _ev_us(lambda: ops.lis_select(h, wq, bq, wk, bk, k), iters)
            by = algorithmic_bytes(b, n, d, hd, k)
            rows.append({"config": name, "images_per_step": b, "n_vis": n, "d": d, "hd": hd, "k": k, "us_per_step": us,
                         "tokens_per_s": b * n / (us * 1e-6),
                         "roofline_path": {"bound": "hbm", "algorithmic_bytes_per_step": by, "achieved_GBps": by / (us * 1e-6) / 1e9,
                                           "frac_of_8TBps": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}})
        del h
    torch.cuda.empty_cache()
    return rows


def bench_config5(ops, _native, n_prompts=64, iters=5):
    """BASELINE.json config 5 on one GPU: a dynamic-resolution batch of 64 prompts, N_i ~ U{576..4096} visual + T_i ~ U{16..128} text tokens,
    k_i = int(0.2 N_i).  Ragged LIS select; scores -> spliced prompt in one call (vsel_lis_select_splice); var-len causal attention over the
    compressed packing vs over the uncompressed one, per layer (7B heads).  The 8-GPU form shards prompts over ranks (no collective)."""
    import numpy as np
    IMG = 151655
    d, hd, hq, hkv = D, HD, 28, 4
    rng = np.random.default_rng(n_prompts)
    n_vis = [int(x) for x in rng.integers(576, 4097, n_prompts)]
    n_txt = [int(x) for x in rng.integers(16, 129, n_prompts)]
    ks = [int(x * 0.2) for x in n_vis]
    seq = [a + t for a, t in zip(n_vis, n_txt)]
    seq_c = [a + t for a, t in zip(ks, n_txt)]
    g = torch.Generator(device="cuda").manual_seed(6)
    mk = lambda *sh: (0.02 * torch.randn(*sh, device="cuda", generator=g)).bfloat16()  # noqa: E731
    wq, bq, wk, bk = mk(hd, d), mk(hd), mk(hd, d), mk(hd)
    h = torch.randn(sum(n_vis), d, device="cuda", generator=g).bfloat16()
    ids = torch.cat([torch.cat((torch.randint(10, 1000, (t // 2,)), torch.full((a,), IMG), torch.randint(10, 1000, (t - t // 2,))))
                     for a, t in zip(n_vis, n_txt)]).cuda()
    emb = torch.randn(sum(seq), d, device="cuda", generator=g).bfloat16()
    pos = torch.arange(sum(seq), device="cuda")[None].expand(3, -1).contiguous()
    e = 2
    out = {"prompts": n_prompts, "visual_tokens": sum(n_vis), "kept": sum(ks), "packed_len": sum(seq), "packed_len_compressed": sum(seq_c)}
    us = _ev_us(lambda: ops.lis_select_varlen(h, n_vis, ks, wq, bq, wk, bk), iters)
    by = sum(n_vis) * d * e + sum(ks) * d * e + 2 * hd * d * e + 2 * hd * e + sum(n_vis) * 4 + sum(ks) * 8
    out["lis_select_ragged"] = {"us": us, "tokens_per_s": sum(n_vis) / (us * 1e-6),
                                "roofline": {"bound": "hbm", "algorithmic_bytes": by, "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}}
    us = _ev_us(lambda: ops.lis_select_splice(h, wq, bq, wk, bk, ids, emb, IMG, seq, n_vis, ks, position_ids=pos), iters)
    # tokens read once + text rows read + the spliced prompt written + weights + scores
    by = sum(n_vis) * d * e + sum(n_txt) * d * e + sum(seq_c) * d * e + 2 * hd * d * e + sum(n_vis) * 4 + sum(seq) * 8
    out["lis_select_splice"] = {"us": us, "tokens_per_s": sum(n_vis) / (us * 1e-6),
                                "roofline": {"bound": "hbm", "algorithmic_bytes": by, "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}}
    del h, emb
    for tag, lens in (("attention_compressed", seq_c), ("attention_uncompressed", seq)):
        t = sum(lens)
        q = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
        kk = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
        v = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
        cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32, device="cuda")
        _native.profile_start()
        ops.varlen_attn(q, kk, v, cu, max(lens))
        kern = sorted(_native.profile_stop())
        us = _ev_us(lambda: ops.varlen_attn(q, kk, v, cu, max(lens)), iters)
        fl = sum(4.0 * L * L * hq * 128 / 2 for L in lens)
        tf = fl / (us * 1e-6) / 1e12
        out[tag] = {"us_per_layer": us, "tflops": tf, "kernel": kern,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS}}
        del q, kk, v
    out["attention_speedup"] = out["attention_uncompressed"]["us_per_layer"] / out["attention_compressed"]["us_per_layer"]
    torch.cuda.empty_cache()
    return out


@torch.no_grad()
def bench_llm_prefill(_native, k, text=64, iters=5):
    """Second half of the BASELINE metric: "prefill ms at 20 % retain, Qwen2.5-VL-7B".  The 28-layer text model of the 7B
    geometry with random-init bf16 weights (no checkpoints offline), attention through the registered `vsel_varlen`
    kernel, GEMMs through PyTorch-ROCm; prefill of L' = k + 64 vs L = N + 64 tokens (inputs_embeds, M-RoPE positions)."""
    from transformers import Qwen2_5_VLTextConfig
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    from visionselector_amd.attention import ATTN_NAME, replace_qwen2_vl_attention_class
    replace_qwen2_vl_attention_class()
    cfg = Qwen2_5_VLTextConfig(hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
                               num_key_value_heads=4, vocab_size=152064, max_position_embeddings=32768,
                               rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1000000.0))
    cfg._attn_implementation = ATTN_NAME
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("cuda"):
            model = hf.Qwen2_5_VLTextModel(cfg).eval()
    finally:
        torch.set_default_dtype(torch.float32)
    out = {"model": "Qwen2.5-VL-7B text model geometry, random-init bf16", "params_B": sum(p.numel() for p in model.parameters()) / 1e9,
           "attention": ATTN_NAME}
    for tag, L in (("retain20", k + text), ("full", N_VIS + text)):
        x = torch.randn(1, L, 3584, device="cuda", dtype=torch.bfloat16) * 0.02
        pos = torch.arange(L, device="cuda")[None, None, :].expand(3, 1, L).contiguous()
        for _ in range(2):
            model(inputs_embeds=x, position_ids=pos, use_cache=False)
        torch.cuda.synchronize()
        _native.profile_start()
        model(inputs_embeds=x, position_ids=pos, use_cache=False)
        torch.cuda.synchronize()
        prof = _native.profile_stop()      # (from 2048 tokens the forward is attn_fwd64_kernel, csrc/attn_fwd64.hip)
        calls = sum(prof.get(nm, (0.0, 0))[1] for nm in ("varlen_attn_fwd_kernel", "attn_fwd64_kernel", "attn_fwd_gqa_kernel", "attn_fwd_gqa64_kernel"))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            model(inputs_embeds=x, position_ids=pos, use_cache=False)
        e1.record()
        torch.cuda.synchronize()
        out[tag] = {"L": L, "prefill_ms": e0.elapsed_time(e1) / iters, "vsel_attention_launches_per_forward": calls}
    out["speedup"] = out["full"]["prefill_ms"] / out["retain20"]["prefill_ms"]
    del model
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
