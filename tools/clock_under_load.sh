#!/bin/bash
# Shader clock and power while a kernel runs (rocm-smi sampled beside a looping process): the dense-MFMA peak of the guide is quoted at
# the 2.4 GHz boost clock; what the attention kernels can reach scales with the clock the part holds under their load.
#   bash tools/clock_under_load.sh                      # on the GPU box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() {  # $1 = label, $2 = python body run in a loop for ~6 s
  python - "$2" <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
from visionselector_amd import ops
g = torch.Generator(device="cuda").manual_seed(7)
n, L = 16, 4096
T = n * L
q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
do = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
o, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
h = torch.randn(64 * 2304, 3584, device="cuda", generator=g).bfloat16()
t0 = time.time()
while time.time() - t0 < 7:
    for _ in range(20):
        exec(sys.argv[1])
    torch.cuda.synchronize()
PY
  pid=$!
  sleep 4
  for i in 1 2 3 4; do
    echo "$1: $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr -s ' ' | tr '\n' ';')"
    sleep 0.5
  done
  wait $pid
}
echo "idle: $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr -s ' ' | tr '\n' ';')"
run "attention forward 16x4096" "ops.varlen_attn(q, k, v, cu, L)"
run "attention backward 16x4096" "ops.varlen_attn_bwd(do, q, k, v, o, lse, cu, L)"
run "copy 1 GB (HBM-bound)" "h2 = h.clone()"
