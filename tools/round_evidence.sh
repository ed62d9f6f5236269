#!/bin/bash
# Everything under profiles/r04_* in one gpurun call: the -m gpu suite (parity margins), bench + rocprofv3 + PMC (collect_round.sh),
# timelines, sweeps.  Copy what is to be judged from gpurun_out/r04_final/ into profiles/.
set -u
ROOT="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is set there)}"
cd "$ROOT"
O=gpurun_out/r04_final
mkdir -p $O
rm -f gpurun_out/parity/r04_parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.txt
python tools/merge_parity.py gpurun_out/parity/r04_parity.jsonl $O/parity.json > /dev/null 2>&1
timeout 1500 bash tools/collect_round.sh r04_final > $O/collect.log 2>&1
g() { grep -v amdgpu.ids; }
# (the s_memtime timelines of round 3 need a -DVSEL_TRACE build: tools/trace_*.py; not part of the routine evidence)
(for m in 0 1; do echo "== VSEL_ATTN_XCD_QUEUE=$m"; VSEL_ATTN_XCD_QUEUE=$m timeout 300 python tools/bench_attn.py --big 2>&1 | g; VSEL_ATTN_XCD_QUEUE=$m timeout 300 python tools/bench_attn_bwd.py --big 2>&1 | g; done) > $O/attn_xcd_queue_ab.txt
timeout 900 bash tools/pmc_attn_mem.sh 16 4096 r04_final/pmc_attn_mem > /dev/null 2>&1
cp gpurun_out/r04_final/pmc_attn_mem/summary.txt $O/attn_l2_counters.txt 2>/dev/null
timeout 300 python tools/bench_attn.py 2>&1 | g > $O/bench_attn.txt
timeout 300 python tools/bench_attn_bwd.py 2>&1 | g > $O/bench_attn_bwd.txt
timeout 300 python tools/bench_sdpa_ref.py 2>&1 | g > $O/sdpa_ref.jsonl
timeout 300 python tools/bench_select_splice.py 2>&1 | g > $O/select_splice.jsonl
timeout 300 python tools/exp_merger_fusion.py 2>&1 | g > $O/merger_fusion.jsonl
timeout 600 python tools/sweep.py 2>&1 | g > $O/config_sweep.json
timeout 600 python tools/bench_c5.py 2>&1 | g > $O/config5.jsonl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_tr_bench tools/lds_tr_bench.hip > /dev/null 2>&1 && timeout 120 /tmp/lds_tr_bench > $O/lds_read_rates.txt 2>&1
timeout 200 bash tools/clock_under_load.sh 2>&1 | g > $O/clock_under_load.txt
timeout 300 python tools/exp_dkdv64_shapes.py 2>&1 | g > $O/dkdv64_shapes.txt
timeout 1500 bash tools/smoke_tools.sh > /dev/null 2>&1; cp gpurun_out/r04_tools_smoke.txt $O/tools_smoke.txt 2>/dev/null
cat $O/pytest_gpu.txt; head -c 600 $O/bench_default.json
