cd $GRAFT_REPO_ROOT
for f in tools/*.py; do
  timeout 90 python $f > /tmp/out.txt 2>&1; rc=$?
  echo "$rc $f $(grep -v amdgpu.ids /tmp/out.txt | tail -1 | cut -c1-150)"
done > gpurun_out/r06_tools_smoke.txt 2>&1
cat gpurun_out/r06_tools_smoke.txt
