cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
rm -f gpurun_out/parity/r02_parity.jsonl
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r02c/pytest.log
cat gpurun_out/r02c/pytest.log | tail -40
