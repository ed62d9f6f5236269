"""Randomised stress of the attention scheduling paths (placement only: every setting must give the per-head / reference-form bits).
Forward: the group-shared forms with their item lists dealt, drawn, and by default, over random uniform / ragged batches and head geometries;
    python tools/stress_attn_sched.py [n_seeds]            # forward
    python tools/stress_attn_sched.py bwd [n_seeds]        # dK / dV: walk directions x XCD queues x kernel forms"""
import sys
if len(sys.argv) > 1 and sys.argv[1] == "bwd":
    sys.argv.pop(1)
    MODE = "bwd"
else:
    MODE = "fwd"
def _fwd():
    import sys, os, random
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    from visionselector_amd import ops, _native as N
    bad = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
        rnd = random.Random(seed)
        hq, hkv = rnd.choice([(28, 4), (32, 8), (16, 2), (8, 2), (12, 2), (6, 2)])
        n_seq = rnd.choice([1, 2, 3, 5, 8, 13, 21, 34, 55, 80])
        hi = rnd.choice([40, 200, 600, 1100, 1600])
        uniform = rnd.random() < 0.3
        lens = [hi] * n_seq if uniform else [rnd.randint(1, hi) for _ in range(n_seq)]
        if sum(lens) > 60000: lens = lens[: max(1, 60000 // hi)]
        total = sum(lens)
        g = torch.Generator(device="cuda").manual_seed(seed)
        q = torch.randn(total, hq, 128, device="cuda", generator=g).bfloat16()
        k = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
        v = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
        cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
        causal = rnd.random() < 0.8
        with N.debug_knob(attn_gqa=0, attn_split=0, attn_rows64=0):
            ref = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens), causal=causal)
        for form in (0, 1):
            for st in (-1, 0, 1):
                with N.debug_knob(attn_gqa=1, attn_gqa_form=form, attn_static=st, attn_split=0, attn_rows64=0):
                    got = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens), causal=causal)
                if not (torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])):
                    bad += 1
                    print("MISMATCH", seed, hq, hkv, n_seq, hi, uniform, causal, form, st)
        dflt = ops.varlen_attn_fwd_lse(q, k, v, cu, max(lens), causal=causal)
        if not torch.equal(dflt[0], ref[0]):
            bad += 1
            print("MISMATCH default", seed)
    print("stress done, mismatches:", bad)
def _bwd():
    import sys, os, random
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    from visionselector_amd import ops, _native as N
    bad = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
        rnd = random.Random(1000 + seed)
        hq, hkv = rnd.choice([(28, 4), (32, 8), (16, 2), (8, 2), (4, 4), (8, 4)])
        n_seq = rnd.choice([1, 2, 3, 5, 8, 13, 21, 40])
        hi = rnd.choice([100, 600, 1300, 2600, 5000])
        lens = [rnd.randint(1, hi) for _ in range(n_seq)]
        while sum(lens) > 40000: lens.pop()
        total = sum(lens)
        g = torch.Generator(device="cuda").manual_seed(seed)
        q = torch.randn(total, hq, 128, device="cuda", generator=g).bfloat16()
        k = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
        v = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
        do = torch.randn(total, hq, 128, device="cuda", generator=g).bfloat16()
        cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
        L = max(lens)
        out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
        res = {}
        for ud in (0, 1):
            for name, form in (("w4", dict(attn_bwd_dkdv64=0, attn_bwd_waves=4)), ("d64", dict(attn_bwd_dkdv64=1))):
                for xq in (0, 1):
                    with N.debug_knob(attn_bwd_updown=ud, attn_bwd_split=0, attn_xcd_queue=xq, **form):
                        res[(ud, name, xq)] = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
        for ud in (0, 1):
            base = res[(ud, "w4", 0)]
            for key in [(ud, "w4", 1), (ud, "d64", 0), (ud, "d64", 1)]:
                if not all(torch.equal(a, b) for a, b in zip(res[key], base)):
                    bad += 1; print("MISMATCH", seed, hq, hkv, lens[:5], key)
        for i in (1, 2):
            a, b = res[(0, "w4", 0)][i].float(), res[(1, "w4", 0)][i].float()
            if float((a - b).abs().max()) > 2 ** -7 * float(a.abs().max()) or not torch.isfinite(b).all():
                bad += 1; print("TOL", seed, i)
    print("stress bwd done, problems:", bad)
(_bwd if MODE == "bwd" else _fwd)()
