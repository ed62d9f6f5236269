cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests/test_lis_gpu.py tests/test_property_gpu.py -m gpu -q -x 2>&1 | tail -15
for B in 1 2 4 8; do for P in 0 1; do VSEL_SMALL_PATH=$P timeout 120 python tools/run_lis.py $B 500; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02e/small.txt
export TMPDIR=/tmp
for B in 1 4; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e/trace_b$B -o t -- python $GRAFT_REPO_ROOT/tools/run_lis.py $B 200 > /dev/null 2>&1)
python tools/trace_gaps.py $(find gpurun_out/r02e/trace_b$B -name '*kernel_trace.csv' | head -1) 5 | tee gpurun_out/r02e/gaps_b$B.txt
done
