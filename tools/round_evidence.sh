#!/bin/bash
set -u
ROOT="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is set there)}"
cd "$ROOT"
mkdir -p gpurun_out/r03_final
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r03_final/pytest_gpu.txt
timeout 1500 bash tools/collect_round.sh r03_final > gpurun_out/r03_final/collect.log 2>&1
timeout 200 python tools/trace_small.py 1 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_final/small_batch_timeline.txt
timeout 200 python tools/trace_attn.py 524 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_final/attention_timeline.txt
timeout 120 tools/launch_floor > gpurun_out/r03_final/launch_floor.txt 2>&1
timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_final/bench_attn.txt
cat gpurun_out/r03_final/pytest_gpu.txt; head -c 600 gpurun_out/r03_final/bench_default.json
