import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops, _native
for nseq, L in ((1, 2368), (32, 524), (4, 2368), (16, 2368), (16, 4096), (4, 8192)):
    g = torch.Generator(device="cuda").manual_seed(7)
    T = nseq * L
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    outs = {}
    for nw in (4, 8):
        with _native.debug_knob("attn_waves", nw):
            for _ in range(3): o = ops.varlen_attn(q, k, v, cu, L)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): o = ops.varlen_attn(q, k, v, cu, L)
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        outs[nw] = o
        print(f"n_seq={nseq} L={L} waves/WG={nw}: {ms*1e3:8.1f} us  {4.0*L*L*28*128/2*nseq/(ms*1e-3)/1e12:6.0f} TF")
    print("   identical:", torch.equal(outs[4], outs[8]))
