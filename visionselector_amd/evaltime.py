"""EVAL_TIME log format of the reference's evaluation harness.

With EVAL_TIME=true the reference prints, per sample,
    Input visual token number is: <N>           (qwen-evaluation/token_compression/selector_model.py:357)
    Generation prefill time is: <ms>            (:358; printed by our *_Selector.forward too, hf_qwen25vl.py)
    Generation latency time is: <ms>            (lmms-eval/lmms_eval/models/qwen2_5_vl_with_token_compression.py:391)
    after generation memory: <bytes>            (:392)
and qwen-evaluation/extract_time.py averages them.  `timed_generate` is the adapter-side half for a harness that calls
`model.generate` itself (torch.cuda.Event works unchanged on ROCm); `summarize_log` is the log reader.
"""
from __future__ import annotations

import os
import re
from typing import Dict, Iterable, List

import torch

_PATTERNS = {
    "memory_bytes": re.compile(r"after generation memory:\s*(\d+)"),
    "latency_ms": re.compile(r"Generation latency time is:\s*([0-9.]+)"),
    "prefill_ms": re.compile(r"Generation prefill time is:\s*([0-9.]+)"),
    "visual_tokens": re.compile(r"Input visual token number is:\s*([0-9.]+)"),
}


def eval_time_enabled() -> bool:
    return os.environ.get("EVAL_TIME", "").lower() == "true"


def timed_generate(model, **generate_kwargs):
    """model.generate(**kwargs); under EVAL_TIME=true also prints the latency and peak-memory lines of the lmms-eval adapter
    and resets the peak-memory counter, like the reference."""
    if not eval_time_enabled():
        return model.generate(**generate_kwargs)
    device = next(model.parameters()).device
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    out = model.generate(**generate_kwargs)
    end.record()
    torch.cuda.synchronize()
    print(f"Generation latency time is: {start.elapsed_time(end)}")
    print("after generation memory:", torch.cuda.max_memory_allocated(device))
    torch.cuda.reset_peak_memory_stats(device)
    return out


def parse_log(lines: Iterable[str]) -> Dict[str, List[float]]:
    """Every occurrence of the four quantities, in file order.  Zero values are dropped, as extract_time.py drops them
    (it tests the parsed number for truth)."""
    found: Dict[str, List[float]] = {k: [] for k in _PATTERNS}
    for line in lines:
        for key, pat in _PATTERNS.items():
            m = pat.search(line)
            if m:
                v = float(m.group(1))
                if key in ("memory_bytes", "visual_tokens"):
                    v = float(int(v))
                if v:
                    found[key].append(v)
    return found


def summarize_log(path: str) -> Dict[str, float]:
    """The four averages extract_time.py prints: max memory in GB (2^30), prefill ms, latency ms, visual tokens."""
    with open(path, "r") as f:
        found = parse_log(f)
    mean = lambda xs: float(sum(xs)) / len(xs) if xs else float("nan")  # noqa: E731
    return {"avg_max_memory_GB": mean(found["memory_bytes"]) / (1024 ** 3), "avg_prefill_ms": mean(found["prefill_ms"]),
            "avg_latency_ms": mean(found["latency_ms"]), "avg_visual_tokens": mean(found["visual_tokens"]),
            "samples": len(found["prefill_ms"])}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="average the EVAL_TIME quantities of an evaluation log")
    ap.add_argument("--path", type=str, default="./log_eval.log")
    s = summarize_log(ap.parse_args().path)
    print(f"Average max memory: {s['avg_max_memory_GB']} GB")
    print(f"Average prefill time: {s['avg_prefill_ms']} mSces")
    print(f"Average latency: {s['avg_latency_ms']} mSces")
    print(f"Average visual token num: {s['avg_visual_tokens']}")
