cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_attn_gpu.py tests/test_attn_bwd_gpu.py -m gpu -q -x 2>&1 | tail -5
python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
