"""The reference's evaluation call for one image in a loop (vsel_lis_select_splice with soft outputs; then plain vsel_lis_select): for rocprofv3 --kernel-trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionselector_amd import ops
d, hd, n, k, n_text, img = 3584, 1792, 2304, 460, 64, 151655
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(n, d, device="cuda", generator=g).bfloat16()
wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
ids = torch.cat((torch.arange(10, 10 + n_text // 2), torch.full((n,), img), torch.arange(50, 50 + n_text - n_text // 2))).cuda()
emb = torch.randn(ids.numel(), d, device="cuda", generator=g).bfloat16()
pos = torch.arange(ids.numel(), device="cuda").repeat(3, 1).contiguous()
mode = sys.argv[1] if len(sys.argv) > 1 else "eval"
for _ in range(60):
    if mode == "eval":
        ops.lis_select_splice(h, wq, bq, wk, bk, ids, emb, img, [ids.numel()], [n], [k], position_ids=pos, soft=True)
    else:
        ops.lis_select(h, wq, bq, wk, bk, k)
torch.cuda.synchronize()
