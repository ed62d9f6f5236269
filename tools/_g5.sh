cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12
for B in 1 2 4 8 32 128; do timeout 300 python tools/run_lis.py $B 300; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02f/batch_sweep.txt
