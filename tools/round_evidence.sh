#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_final
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r02_final/pytest_gpu.txt
timeout 1500 bash tools/collect_round.sh r02_final > gpurun_out/r02_final/collect.log 2>&1
timeout 200 python tools/trace_small.py 1 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_final/small_batch_timeline.txt
timeout 200 python tools/trace_attn.py 524 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_final/attention_timeline.txt
timeout 120 tools/launch_floor > gpurun_out/r02_final/launch_floor.txt 2>&1
timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_final/bench_attn.txt
cat gpurun_out/r02_final/pytest_gpu.txt; head -c 600 gpurun_out/r02_final/bench_default.json
