cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 600 python -m pytest tests/test_lis_gpu.py tests/test_property_gpu.py tests/test_bench_gpu.py -m gpu -q 2>&1 | tail -8
for P in 0 1; do VSEL_PIPELINE=$P timeout 300 python tools/run_lis.py 128 50; VSEL_PIPELINE=$P timeout 300 python tools/run_lis.py 32 100; VSEL_PIPELINE=$P timeout 300 python tools/run_lis.py 64 100; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02d/pipeline.txt
timeout 600 python bench.py --no-attn --no-llm --no-train --no-cpu-baseline > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err; python -c "
import json; r=json.load(open('gpurun_out/r02d/bench.json')); print(r['value'], r['ms_per_step'], r['ms_per_step_instrumented'], r['roofline'], r['roofline_path']); print({k:(round(v['avg_us'],1), v['launches_per_step']) for k,v in r['kernels'].items()})"
