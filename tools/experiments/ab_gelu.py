"""us of the merger-side GELU (+ column sums) on the headline shape for a given build of the library (--lib)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionselector_amd import _native as N
if "--lib" in sys.argv:
    N.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import torch
from visionselector_amd import ops
b, n, c = 128, 2304, 5120
x = torch.randn(b * n, c, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).bfloat16()
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
y, s = ops.gelu_colsum(x, b)
ref = torch.nn.functional.gelu(x)
a1 = t(lambda: ops.gelu_colsum(x, b)); a0 = t(lambda: ops.gelu_colsum(x, b, sums=False)); a2 = t(lambda: ops.gelu_colsum(x, b)); a3 = t(lambda: ops.gelu_colsum(x, b, sums=False))
print(os.path.basename(N.LIB_PATH or "libvsel.so"), "with sums %.1f / %.1f us  no sums %.1f / %.1f us  torch %.1f us  equal_to_torch=%s  sum_checksum=%.6e" % (
    a1, a2, a0, a3, t(lambda: torch.nn.functional.gelu(x)), bool(torch.equal(y, ref)), float(s.double().sum())))
