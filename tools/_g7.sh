cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 600 python -m pytest tests/test_property_gpu.py -m gpu -q -k gelu 2>&1 | tail -3
python tools/exp_merger_fusion.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02h/merger_fusion.jsonl
