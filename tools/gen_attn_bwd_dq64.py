#!/usr/bin/env python3
"""Generator of the hand-scheduled per-item body of attn_bwd_dq64_kernel (visionselector_amd/csrc/attn_bwd_dq64.hip).

    python tools/gen_attn_bwd_dq64.py            # rewrites visionselector_amd/csrc/attn_bwd_dq64_body.inc

The dQ pass of the attention backward in the structure of the 64-rows-per-wave forward (tools/gen_attn_fwd64.py): four waves, one per
SIMD, 64 queries (two 32-row blocks A / B) per wave, the whole 512-entry register file.  Arithmetic = attn_bwd_dq_kernel (attn_bwd.hip)
operation for operation per query row: S^T = K Q^T, dP^T = V dO^T, P = exp2(S c - lse2), dS = bf16(P (dP - D)), dQ^T += K^T dS^T per
32-key block, and D = rowsum(dO o O), lse2 = lse log2(e) left in the workspace for the dK / dV kernel -- dQ, D and lse2 are bit-identical.

Units.  A (key block n, row block X) unit = 8 + 8 MFMAs "SdP" (S and dP alternating, K / V row fragments from LDS), 72 VALU (exp2, dS)
and 8 MFMAs "dQ" (K^T fragments by transposed reads, held in registers for both row blocks of the key block).  The MFMA stream of a
64-key tile t (key blocks 2t, 2t+1):
    SdP(2t,A) dQ(2t-1,B) SdP(2t,B) dQ(2t,A) SdP(2t+1,A) dQ(2t,B) SdP(2t+1,B) dQ(2t+1,A)        [dQ(2t+1,B) opens the next tile]
and the VALU of a unit runs in the gaps of the 8 + 16 MFMAs that follow its SdP -- the other row block's.  The last unit of a tile
crosses the step barrier with everything it needs in registers (S, dP, the K^T fragments), so the K / V ring has two slots.

Register map (per lane)
  a[0:127]    dQ^T accumulators, block b, d-tile dt: a[64 b + 16 dt .. +15]
  a[128:191]  Q^T fragments, a[192:255] dO^T fragments (block b, k-step st: + 32 b + 4 st)
  v[32:63]    S (block b: + 16 b), v[64:95] dP, v[96:111] dS bf16 (block b, 16-key half m: + 8 b + 4 m)
  v[112:143]  K / V row fragments (8 slots of 4), v[144:175] K^T fragments (half m, d-tile dt: + 16 m + 4 dt)
  v[176:183]  row-fragment LDS addresses (k-step), v[184:191] transposed-fragment addresses (2 dt + hi)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_attn_fwd64 import Gen, v, vr, a, ar, s, sr, place, spread            # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("DQ64_OUT", os.path.join(ROOT, "visionselector_amd", "csrc", "attn_bwd_dq64_body.inc"))

A_DQ, A_Q, A_DO = 0, 128, 192
V_S, V_DP, V_DS, V_KV, V_KT = 32, 64, 96, 112, 144
V_RA, V_TR = 176, 184
V_LSE2, V_DSUM, V_KMAX = 192, 194, 196
V_T = 198                                         # T0..T7
V_LO = 206                                        # [4] per-lane byte offsets of the four K / V slices a wave loads per tile
V_P8, V_LANE4, V_LANE, V_HH8, V_NEGINF, V_KR = 210, 211, 212, 213, 214, 215      # KR[2]
V_QLO = 217                                       # [4] per-lane byte offsets of a Q / dO / O / dQ row slice by slice phase
V_U = 222                                         # U0..U9 (even-aligned pairs)
V_X = 232                                         # 232, 233: byte offsets of this lane's rows in lse / D / lse2; 234 = (lane >> 4) * row stride
V_RS = 234
FIRST_V, LAST_V = 32, 255

S_T, S_NT, S_NW, S_MFIRST, S_LEN, S_SL2, S_LDSW, S_W4 = 40, 41, 42, 43, 44, 45, 46, 47
S_KPTR, S_VPTR = 48, 50
S_KSTEP, S_SCALE, S_KRS2, S_QRS2 = 52, 53, 54, 55
S_KB0, S_VB0 = 56, 58
S_TMP = 60                                        # 60..63
S_EXEC = 64
S_RING, S_QST, S_RST, S_NVALID = 66, 67, 68, 69   # LDS base; this wave's 16 KiB of the staging area / of the ring; rows of this wave that exist
S_FST = 70                                        # bytes between consecutive rows of lse / D / lse2 (hq * 4)
S_M0SAVE, S_LENM1 = 74, 75
S_TMP2 = 76                                       # 76..77
S_NSTEADY = 78
S_QPTR = 80                                       # 80..81
FIRST_S, LAST_S = 40, 87

KBUF = 16384
OPT = {
    "dma": "1,3,5,7,9,11,13,15",                  # gaps of a steady step's first SdP batch that carry the next tile's eight slices
    "split": "24",                                # VALU of a unit issued beside the dQ batch right behind its SdP; the rest beside the next SdP batch
    "ko": "",                                     # knock-outs (timing only, WRONG results): valu, lds, dma, mfma joined by "+"
    "bar": "mid",                                 # mid: the step's barrier in front of its last dQ batch, whose gaps then carry the NEXT tile's
                                                  # first row fragments across the step boundary (tools/gen_attn_bwd_dkdv64.py has the argument)
}
for kv in os.environ.get("DQ64_OPTS", "").split(","):
    if "=" in kv:
        key, val = kv.split("=", 1)
        assert key in OPT, key
        OPT[key] = val.replace("/", ",")


# ---- instruction streams ----------------------------------------------------------------------------------------------------------
def kv_read(g, tensor, kb, st, ks):
    """K / V row fragment (key block kb of the tile, k-step st) into slot (st & 3) + (4 if V)"""
    slot = (st & 3) + (4 if tensor == "v" else 0)
    off = (0 if tensor == "k" else 2 * KBUF) + ks * KBUF + kb * 8192
    return (f"ds_read_b128 {vr(V_KV + 4 * slot, 4)}, {v(V_RA + st)} offset:{off}", (tensor, kb, st))


def sdp_mfma(g, i, kb, blk, wait=True):
    """MFMA i of a SdP batch: k-step i // 2; even = S (K fragment x Q), odd = dP (V fragment x dO)"""
    st, is_dp = i // 2, i % 2
    if wait and not is_dp and st in (0, 4, 6):
        # counted waits at three points of the batch: the first eight fragments; then pairs of k-steps of the second eight, whose
        # reads were issued one MFMA behind the last user of their slot (the V fragment of k-step 7 in gap 8)
        g.need(("v", kb, {0: 3, 4: 5, 6: 7}[st]))
    frag = vr(V_KV + 4 * ((st & 3) + (4 if is_dp else 0)), 4)
    dst = vr((V_DP if is_dp else V_S) + 16 * blk, 16)
    src = ar((A_DO if is_dp else A_Q) + 32 * blk + 4 * st, 4)
    g.e(f"v_mfma_f32_32x32x16_bf16 {dst}, {frag}, {src}, {'0' if st == 0 else dst}")


def kt_reads(kb, ks):
    """the sixteen transposed reads of a key block's K^T fragments"""
    out = []
    for m in range(2):
        for dt in range(4):
            for hi in range(2):
                off = ks * KBUF + (32 * kb + 16 * m) * 256
                out.append((f"ds_read_b64_tr_b16 {vr(V_KT + 16 * m + 4 * dt + 2 * hi, 2)}, {v(V_TR + 2 * dt + hi)} offset:{off}", ("t", kb, m, dt, hi)))
    return out


def dq_mfma(g, i, kb, blk, wait):
    m, dt = i // 4, i % 4
    if wait and i == 0:
        g.need(("t", kb, 1, 3, 1))
    acc = ar(A_DQ + 64 * blk + 16 * dt, 16)
    g.e(f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(V_KT + 16 * m + 4 * dt, 4)}, {vr(V_DS + 8 * blk + 4 * m, 4)}, {acc}")


def unit_valu(blk):
    """p = exp2(s c - lse2) in place, dS = bf16(p (dP - D)): 72 VALU, skewed so that no result meets its consumer within two instructions"""
    sb, db = V_S + 16 * blk, V_DP + 16 * blk
    ops = []
    for i in range(16 + 7):
        if i < 16:
            ops.append(f"v_fma_f32 {v(sb + i)}, {v(sb + i)}, {s(S_SL2)}, -{v(V_LSE2 + blk)}")
        r = i - 1
        if 0 <= r < 16:
            ops.append(f"v_sub_f32 {v(db + r)}, {v(db + r)}, {v(V_DSUM + blk)}")
        r = i - 3
        if 0 <= r < 16:
            ops.append(f"v_exp_f32 {v(sb + r)}, {v(sb + r)}")
        r = i - 5
        if 0 <= r < 16:
            ops.append(f"v_mul_f32 {v(db + r)}, {v(sb + r)}, {v(db + r)}")
        r = i - 6
        if 0 <= r < 16 and r % 2 == 1:
            ops.append(f"v_cvt_pk_bf16_f32 {v(V_DS + 8 * blk + 4 * (r >> 3) + ((r & 7) >> 1))}, {v(db + r - 1)}, {v(db + r)}")
    assert len(ops) == 72
    return ops


def mask_unit(g, blk, kb, tile_reg, uniq):
    """S <- -inf where the key lies beyond the lane's last visible key (then p = exp2(-inf) = 0, as the reference's select)"""
    g.e(f"s_cmp_ge_i32 {s(tile_reg)}, {s(S_MFIRST)}")
    g.e(f"s_cbranch_scc0 {g.lref('Lnm' + uniq)}")
    g.e("s_nop 15")
    g.e(f"s_lshl_b32 {s(S_TMP2)}, {s(tile_reg)}, 6")
    g.e(f"v_sub_u32 {v(V_KR + blk)}, {v(V_KMAX + blk)}, {s(S_TMP2)}")
    g.e(f"v_sub_u32 {v(V_KR + blk)}, {v(V_KR + blk)}, {v(V_HH8)}")
    for r in range(16):
        const = 32 * kb + 16 * (r >> 3) + (r & 7)
        reg = V_S + 16 * blk + r
        g.e(f"v_cmp_gt_i32 vcc, {const}, {v(V_KR + blk)}")
        g.e(f"v_cndmask_b32 {v(reg)}, {v(reg)}, {v(V_NEGINF)}, vcc")
    g.label("Lnm" + uniq)


def dma_tile(g, tensor, slot, tile_reg, uniq):
    """one 16 KiB tile of K or V into ring slot `slot` (four 1 KiB slices per wave; partial tiles clamp the source row); advances the pointer"""
    ptr, b0 = (S_KPTR, S_KB0) if tensor == "k" else (S_VPTR, S_VB0)
    lds0 = (0 if tensor == "k" else 2 * KBUF) + slot * KBUF
    g.e(f"s_lshl_b32 {s(S_TMP2)}, {s(tile_reg)}, 6")
    g.e(f"s_add_i32 {s(S_TMP2 + 1)}, {s(S_TMP2)}, 64")
    g.e(f"s_cmp_gt_i32 {s(S_TMP2 + 1)}, {s(S_LEN)}")
    g.e(f"s_cbranch_scc1 {g.lref('Ltail' + uniq)}")
    for u in range(4):
        g.e(f"s_add_u32 m0, {s(S_LDSW)}, {lds0 + u * 4096}")
        g.e("s_nop 0")
        g.e(f"global_load_lds_dwordx4 {v(V_LO + u)}, {sr(ptr)}")
    g.e(f"s_branch {g.lref('Ldone' + uniq)}")
    g.label("Ltail" + uniq)
    g.e(f"s_add_i32 {s(S_TMP2)}, {s(S_TMP2)}, {s(S_W4)}")
    for u in range(4):
        g.e(f"v_add_u32 {v(V_U)}, {s(S_TMP2)}, {v(V_LANE4)}")
        g.e(f"v_min_i32 {v(V_U)}, {v(V_U)}, {s(S_LENM1)}")
        g.e(f"v_mul_lo_u32 {v(V_U)}, {v(V_U)}, {s(S_KRS2)}")
        g.e(f"v_add_u32 {v(V_U)}, {v(V_U)}, {v(V_P8)}")
        g.e(f"s_add_u32 m0, {s(S_LDSW)}, {lds0 + u * 4096}")
        g.e(f"s_add_i32 {s(S_TMP2)}, {s(S_TMP2)}, 16")
        g.e(f"global_load_lds_dwordx4 {v(V_U)}, {sr(b0)}")
    g.label("Ldone" + uniq)
    g.e(f"s_add_u32 {s(ptr)}, {s(ptr)}, {s(S_KSTEP)}")
    g.e(f"s_addc_u32 {s(ptr + 1)}, {s(ptr + 1)}, 0")


def dma_pieces(tensor, slot):
    ptr = S_KPTR if tensor == "k" else S_VPTR
    lds0 = (0 if tensor == "k" else 2 * KBUF) + slot * KBUF
    pieces = [(f"s_add_u32 m0, {s(S_LDSW)}, {lds0 + u * 4096}", f"global_load_lds_dwordx4 {v(V_LO + u)}, {sr(ptr)}") for u in range(4)]
    adv = [f"s_add_u32 {s(ptr)}, {s(ptr)}, {s(S_KSTEP)}", f"s_addc_u32 {s(ptr + 1)}, {s(ptr + 1)}, 0"]
    return pieces, adv


KO = set(OPT["ko"].split("+")) - {""}


def emit_batch(g, n, mfma_fn, gaps):
    for i in range(n):
        if "mfma" in KO:
            mark = len(g.lines)
            mfma_fn(i)
            g.lines[mark:] = [ln for ln in g.lines[mark:] if not ln.startswith("v_mfma")]
        else:
            mfma_fn(i)
        for ins in gaps[i]:
            if isinstance(ins, tuple):
                if "lds" in KO:
                    g.done.add(ins[1])
                else:
                    g.lds(ins[0], ins[1])
            elif "valu" in KO and ins.split()[0] in ("v_fma_f32", "v_sub_f32", "v_exp_f32", "v_mul_f32", "v_cvt_pk_bf16_f32"):
                pass
            elif "dma" in KO and (ins.startswith("global_load_lds") or ins.startswith("s_add_u32 m0")):
                pass
            else:
                g.e(ins)


def first_reads(kb, ks):
    return [kv_read(None, t, kb, st, ks) for st in range(4) for t in ("k", "v")]


def second_reads(kb, ks):
    """(gap, read) of the k-steps 4..7: a slot is free one MFMA after its last user"""
    out = []
    for st in range(4):
        out.append((2 * st + 1, kv_read(None, "k", kb, st + 4, ks)))
        out.append((2 * st + 2, kv_read(None, "v", kb, st + 4, ks)))
    return out


VALU_SPLIT = int(OPT["split"])
BAR_MID = OPT["bar"] == "mid"
CARRY = [None]


def carry_reads():
    """reads in flight across a step boundary, oldest first (bar=mid); found by a dry run of a step body, the same for every body"""
    if not BAR_MID or "lds" in KO:
        return []
    if CARRY[0] is None:
        dry = Gen()
        dry.out = [tg for _, tg in first_reads(0, 0)]
        step_body(dry, 1, "dry", True, True)
    return list(CARRY[0])


def step_body(g, par, tag, has_prev, steady):
    """tile t = s[S_T]: its four units; the last unit's VALU tail and dQ run at the head of the next step (or in the drain).
    On entry with has_prev: S / dP / K^T of unit (2t-1, B) in registers, its first VALU_SPLIT VALU done."""
    ks = vs = par
    e = g.e
    # the first SdP batch's first eight fragments (bar=mid: in flight since the previous step's last batch)
    if not BAR_MID:
        for ins, tag_ in first_reads(0, ks):
            g.lds(ins, tag_)
    order = [(0, 0), (0, 1), (1, 0), (1, 1)]                 # (key block, row block) in MFMA-stream order
    for ui, (kb, blk) in enumerate(order):
        # ---- SdP(kb, blk) beside the VALU tail of the previous unit ----
        gaps = [[] for _ in range(17)]
        for gp, rd in second_reads(kb, ks):
            gaps[gp].append(rd)
        prev = order[ui - 1] if ui > 0 else (1, 1)
        if ui > 0 or has_prev:
            tail = unit_valu(prev[1])[VALU_SPLIT:]
            place(gaps, tail, spread(len(tail), 0, 15))
        if blk == 1:                                         # the key block's K^T fragments for the dQ batches of both row blocks
            kt = kt_reads(kb, ks)
            place(gaps, kt, spread(16, 0, 15))
        if ui == 0 and steady:
            where = [int(x) for x in OPT["dma"].split(",")]
            for (tensor, w4) in (("k", where[:4]), ("v", where[4:])):
                pieces, adv = dma_pieces(tensor, 1 - par)
                for w, (m0w, ld) in zip(w4, pieces):
                    rest = gaps[w]
                    gaps[w] = [m0w] + (rest[:1] if rest else ["s_nop 0"]) + [ld] + rest[1:]
                gaps[w4[-1] + 1] = adv + gaps[w4[-1] + 1]
        emit_batch(g, 16, lambda i: sdp_mfma(g, i, kb, blk), gaps)
        for ins in gaps[16]:
            e(ins)
        mask_unit(g, blk, kb, S_T, f"{tag}u{ui}")
        # ---- dQ of the previous unit beside the head of this unit's VALU and the next SdP's first reads ----
        if ui > 0 or has_prev:
            pkb, pblk = prev
            gq = [[] for _ in range(8)]
            head = unit_valu(blk)[:VALU_SPLIT]
            place(gq, head, spread(len(head), 3, 7))
            if ui < 3:
                nkb = order[ui + 1][0]
                place(gq, first_reads(nkb, ks), [1, 1, 2, 2, 3, 3, 4, 4])
            elif BAR_MID:
                # every wave is past its reads of this tile (the next loads overwrite its slots) and the next tile, loaded during the
                # first batch, has landed
                e("s_waitcnt vmcnt(0)")
                e("s_barrier")
                place(gq, first_reads(0, 1 - ks), [1, 1, 2, 2, 3, 3, 4, 4])
            # K^T of the previous unit's key block: read beside its row-block-A SdP... held since then (row block B reuses them)
            emit_batch(g, 8, lambda i: dq_mfma(g, i, pkb, pblk, wait=(pblk == 0)), gq)
        else:
            # first step of an item: no previous unit; this unit's VALU head and the next reads without cover
            for ins in unit_valu(blk)[:VALU_SPLIT]:
                e(ins)
            for ins, tag_ in first_reads(order[ui + 1][0], ks):
                g.lds(ins, tag_)
    if not BAR_MID or "lds" in KO:
        assert not [t for t in g.out if t[0] in ("k", "v")], g.out
    else:
        if CARRY[0] is None:
            CARRY[0] = list(g.out)
        assert g.out == CARRY[0], (g.out, CARRY[0])


def drain(g):
    """the last unit (B of the wave's last key block): the rest of its VALU, then its dQ from registers"""
    for ins in unit_valu(1)[VALU_SPLIT:]:
        g.e(ins)
    g.e("s_nop 1")
    for i in range(8):
        dq_mfma(g, i, 1, 1, wait=False)


def gen_step(g, par):
    P = f"p{par}"
    e = g.e
    g.label("Lstep" + P)
    if not BAR_MID:
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
    e(f"s_cmp_lt_i32 {s(S_T)}, {s(S_NSTEADY)}")
    e(f"s_cbranch_scc0 {g.lref('Lgen' + P)}")
    e(f"s_cmp_gt_i32 {s(S_T)}, 0")
    e(f"s_cbranch_scc0 {g.lref('Lgen' + P)}")
    g.out = carry_reads()
    step_body(g, par, "s" + P, True, True)
    e(f"s_branch {g.lref('Lend' + P)}")
    g.label("Lgen" + P)
    g.out = carry_reads()
    # K(t + 1), V(t + 1) -> slots 1 - par
    e(f"s_add_i32 {s(S_TMP)}, {s(S_T)}, 1")
    e(f"s_cmp_lt_i32 {s(S_TMP)}, {s(S_NT)}")
    e(f"s_cbranch_scc0 {g.lref('Lnoload' + P)}")
    dma_tile(g, "k", 1 - par, S_TMP, "k" + P)
    dma_tile(g, "v", 1 - par, S_TMP, "v" + P)
    g.label("Lnoload" + P)
    e(f"s_cmp_lt_i32 {s(S_T)}, {s(S_NW)}")
    e(f"s_cbranch_scc0 {g.lref('Lnotfull' + P)}")
    e(f"s_cmp_gt_i32 {s(S_T)}, 0")
    e(f"s_cbranch_scc0 {g.lref('Lfirst' + P)}")
    step_body(g, par, "g" + P, True, False)
    e(f"s_branch {g.lref('Lend' + P)}")
    g.label("Lfirst" + P)
    g.out = carry_reads()
    if par == 0:
        step_body(g, par, "f" + P, False, False)
    e(f"s_branch {g.lref('Lend' + P)}")
    g.label("Lnotfull" + P)
    g.out = carry_reads()                                     # (a wave past its last tile: in flight only in its first such step)
    if BAR_MID:
        e("s_waitcnt vmcnt(0)")                               # its share of the next tile's loads, then the step's barrier
        e("s_barrier")
    e(f"s_cmp_eq_u32 {s(S_T)}, {s(S_NW)}")                    # the step right behind the wave's last tile: its last unit is still open
    e(f"s_cbranch_scc0 {g.lref('Lend' + P)}")
    e(f"s_cmp_gt_i32 {s(S_NW)}, 0")
    e(f"s_cbranch_scc0 {g.lref('Lend' + P)}")
    drain(g)
    g.label("Lend" + P)
    e(f"s_add_i32 {s(S_T)}, {s(S_T)}, 1")
    e(f"s_cmp_lt_i32 {s(S_T)}, {s(S_NT)}")
    e(f"s_cbranch_scc0 {g.lref('Lepi')}")
    if par == 1:
        e(f"s_branch {g.lref('Lstepp0')}")


def row_dma(g, base_name, lds_reg, uniq):
    """the wave's 64 rows of one [T, hq, 128] tensor into 16 KiB of LDS: 16 slices of 4 whole rows (row r, chunk c at c ^ swz(r))"""
    e = g.e
    e(f"s_mov_b64 {sr(S_QPTR)}, %[{base_name}]")
    e(f"s_cmp_gt_i32 {s(S_NVALID)}, 63")
    e(f"s_cbranch_scc0 {g.lref('Lrp' + uniq)}")
    for i in range(16):
        e(f"s_add_u32 m0, {s(lds_reg)}, {1024 * i}")
        e("s_nop 0")
        e(f"global_load_lds_dwordx4 {v(V_QLO + (i & 3))}, {sr(S_QPTR)}")
        e(f"s_add_u32 {s(S_QPTR)}, {s(S_QPTR)}, {s(S_TMP + 1)}")
        e(f"s_addc_u32 {s(S_QPTR + 1)}, {s(S_QPTR + 1)}, 0")
    e(f"s_branch {g.lref('Lrd' + uniq)}")
    g.label("Lrp" + uniq)
    e(f"s_cmp_gt_i32 {s(S_NVALID)}, 0")
    e(f"s_cbranch_scc0 {g.lref('Lrd' + uniq)}")
    e(f"s_add_i32 {s(S_TMP2)}, {s(S_NVALID)}, -1")
    for i in range(16):
        e(f"v_add_u32 {v(V_U)}, {4 * i}, {v(V_LANE4)}")
        e(f"v_min_i32 {v(V_U)}, {v(V_U)}, {s(S_TMP2)}")
        e(f"v_mul_lo_u32 {v(V_U)}, {v(V_U)}, {s(S_QRS2)}")
        e(f"v_sub_u32 {v(V_U + 1)}, {v(V_QLO + (i & 3))}, {v(V_RS)}")
        e(f"v_add_u32 {v(V_U)}, {v(V_U)}, {v(V_U + 1)}")
        e(f"s_add_u32 m0, {s(lds_reg)}, {1024 * i}")
        e("s_nop 0")
        e(f"global_load_lds_dwordx4 {v(V_U)}, {sr(S_QPTR)}")
    g.label("Lrd" + uniq)


def gen_body():
    g = Gen()
    e = g.e
    e(f"s_mov_b64 {sr(S_EXEC)}, exec")
    e(f"s_mov_b32 {s(S_M0SAVE)}, m0")
    for dst, name in ((S_NT, "ntiles"), (S_NW, "nw"), (S_MFIRST, "mfirst"), (S_LEN, "len"), (S_SL2, "sl2"), (S_SCALE, "scale"), (S_KRS2, "krs2"),
                      (S_QRS2, "qrs2"), (S_NVALID, "nvalid"), (S_FST, "fstride")):
        e(f"s_mov_b32 {s(dst)}, %[{name}]")
    e(f"s_lshl_b32 {s(S_W4)}, %[wave], 2")
    e(f"s_mov_b32 {s(S_RING)}, %[ldsbase]")
    e(f"s_lshl_b32 {s(S_TMP)}, %[wave], 10")
    e(f"s_add_u32 {s(S_LDSW)}, {s(S_RING)}, {s(S_TMP)}")
    e(f"s_lshl_b32 {s(S_TMP)}, %[wave], 14")
    e(f"s_add_u32 {s(S_RST)}, {s(S_RING)}, {s(S_TMP)}")        # this wave's 16 KiB of the (still empty) ring: second staging area
    e(f"s_add_u32 {s(S_QST)}, {s(S_RST)}, 0x10000")            # ... and of the staging area behind the ring
    e(f"s_add_i32 {s(S_LENM1)}, {s(S_LEN)}, -1")
    # steady steps: full steps t >= 1 whose next tile exists and is whole: t < min(nw, ntiles - 1, len / 64 - 1)
    e(f"s_add_i32 {s(S_TMP)}, {s(S_NT)}, -1")
    e(f"s_min_i32 {s(S_NSTEADY)}, {s(S_NW)}, {s(S_TMP)}")
    e(f"s_ashr_i32 {s(S_TMP)}, {s(S_LEN)}, 6")
    e(f"s_add_i32 {s(S_TMP)}, {s(S_TMP)}, -1")
    e(f"s_min_i32 {s(S_NSTEADY)}, {s(S_NSTEADY)}, {s(S_TMP)}")
    e(f"s_mov_b64 {sr(S_KB0)}, %[kbase]")
    e(f"s_mov_b64 {sr(S_VB0)}, %[vbase]")
    e(f"s_mul_i32 {s(S_TMP)}, {s(S_W4)}, {s(S_KRS2)}")
    e(f"s_add_u32 {s(S_KPTR)}, {s(S_KB0)}, {s(S_TMP)}")
    e(f"s_addc_u32 {s(S_KPTR + 1)}, {s(S_KB0 + 1)}, 0")
    e(f"s_add_u32 {s(S_VPTR)}, {s(S_VB0)}, {s(S_TMP)}")
    e(f"s_addc_u32 {s(S_VPTR + 1)}, {s(S_VB0 + 1)}, 0")
    e(f"s_lshl_b32 {s(S_KSTEP)}, {s(S_KRS2)}, 6")
    # lane-derived constants (as in gen_attn_fwd64.py; emulated against attn_common.h there)
    T = [V_T + i for i in range(6)]
    e(f"v_mbcnt_lo_u32_b32 {v(V_LANE)}, -1, 0")
    e(f"v_mbcnt_hi_u32_b32 {v(V_LANE)}, -1, {v(V_LANE)}")
    e(f"v_lshrrev_b32 {v(V_LANE4)}, 4, {v(V_LANE)}")
    e(f"v_lshrrev_b32 {v(V_HH8)}, 5, {v(V_LANE)}")
    e(f"v_lshlrev_b32 {v(V_HH8)}, 3, {v(V_HH8)}")
    e(f"s_and_b32 {s(S_TMP)}, %[wave], 3")
    e(f"v_and_b32 {v(T[0])}, 15, {v(V_LANE)}")
    e(f"v_lshlrev_b32 {v(T[1])}, 2, {v(V_LANE4)}")
    e(f"v_or_b32 {v(T[1])}, {s(S_TMP)}, {v(T[1])}")
    e(f"v_xor_b32 {v(T[0])}, {v(T[0])}, {v(T[1])}")
    e(f"v_lshlrev_b32 {v(V_P8)}, 4, {v(T[0])}")

    def swz_of(row, dst, tmp):
        e(f"v_and_b32 {v(dst)}, 3, {v(row)}")
        e(f"v_lshlrev_b32 {v(dst)}, 2, {v(dst)}")
        e(f"v_bfe_u32 {v(tmp)}, {v(row)}, 2, 2")
        e(f"v_or_b32 {v(dst)}, {v(dst)}, {v(tmp)}")
    e(f"v_and_b32 {v(T[0])}, 31, {v(V_LANE)}")
    e(f"v_and_b32 {v(T[1])}, 0x13, {v(T[0])}")
    e(f"v_and_b32 {v(T[2])}, 4, {v(T[0])}")
    e(f"v_lshl_or_b32 {v(T[1])}, {v(T[2])}, 1, {v(T[1])}")
    e(f"v_and_b32 {v(T[2])}, 8, {v(T[0])}")
    e(f"v_lshrrev_b32 {v(T[2])}, 1, {v(T[2])}")
    e(f"v_or_b32 {v(T[1])}, {v(T[1])}, {v(T[2])}")
    swz_of(T[1], T[2], T[3])
    e(f"v_lshrrev_b32 {v(T[3])}, 5, {v(V_LANE)}")
    e(f"v_xor_b32 {v(T[3])}, {v(T[3])}, {v(T[2])}")
    e(f"v_lshlrev_b32 {v(V_RA)}, 8, {v(T[1])}")
    e(f"v_lshl_or_b32 {v(V_RA)}, {v(T[3])}, 4, {v(V_RA)}")
    e(f"v_and_b32 {v(T[0])}, 15, {v(V_LANE)}")
    e(f"v_lshrrev_b32 {v(T[1])}, 2, {v(T[0])}")
    e(f"v_add_u32 {v(T[1])}, {v(T[1])}, {v(V_HH8)}")
    swz_of(T[1], T[2], T[3])
    e(f"v_and_b32 {v(T[3])}, 1, {v(V_LANE4)}")
    e(f"v_lshlrev_b32 {v(T[3])}, 1, {v(T[3])}")
    e(f"v_bfe_u32 {v(T[4])}, {v(T[0])}, 1, 1")
    e(f"v_or_b32 {v(T[3])}, {v(T[3])}, {v(T[4])}")
    e(f"v_xor_b32 {v(T[3])}, {v(T[3])}, {v(T[2])}")
    e(f"v_lshlrev_b32 {v(V_TR)}, 8, {v(T[1])}")
    e(f"v_lshl_or_b32 {v(V_TR)}, {v(T[3])}, 4, {v(V_TR)}")
    e(f"v_and_b32 {v(T[4])}, 1, {v(T[0])}")
    e(f"v_lshl_or_b32 {v(V_TR)}, {v(T[4])}, 3, {v(V_TR)}")
    for st in range(1, 8):
        e(f"v_xor_b32 {v(V_RA + st)}, {st << 5}, {v(V_RA)}")
    for dt in range(1, 4):
        e(f"v_xor_b32 {v(V_TR + 2 * dt)}, {dt << 6}, {v(V_TR)}")
    for dt in range(4):
        e(f"v_xor_b32 {v(V_TR + 2 * dt + 1)}, 16, {v(V_TR + 2 * dt)}")
        e(f"v_add_u32 {v(V_TR + 2 * dt + 1)}, 0x400, {v(V_TR + 2 * dt + 1)}")
    for i in range(8):
        e(f"v_add_u32 {v(V_RA + i)}, {s(S_RING)}, {v(V_RA + i)}")
        e(f"v_add_u32 {v(V_TR + i)}, {s(S_RING)}, {v(V_TR + i)}")
    e(f"v_mov_b32 {v(V_KMAX)}, %[kmaxa]")
    e(f"v_mov_b32 {v(V_KMAX + 1)}, %[kmaxb]")
    e(f"v_mov_b32 {v(V_NEGINF)}, 0xff800000")
    # K / V slice offsets
    e(f"s_lshl_b32 {s(S_TMP)}, {s(S_KRS2)}, 4")
    e(f"v_mul_lo_u32 {v(V_LO)}, {v(V_LANE4)}, {s(S_KRS2)}")
    e(f"v_add_u32 {v(V_LO)}, {v(V_LO)}, {v(V_P8)}")
    for u in range(1, 4):
        e(f"v_add_u32 {v(V_LO + u)}, {v(V_LO + u - 1)}, {s(S_TMP)}")
    # row-slice offsets of the [T, hq, 128] tensors
    e(f"v_mul_lo_u32 {v(V_RS)}, {v(V_LANE4)}, {s(S_QRS2)}")
    e(f"v_and_b32 {v(V_T + 1)}, 15, {v(V_LANE)}")
    e(f"v_lshlrev_b32 {v(V_T + 2)}, 2, {v(V_LANE4)}")
    for kq in range(4):
        e(f"v_or_b32 {v(V_QLO + kq)}, {kq}, {v(V_T + 2)}")
        e(f"v_xor_b32 {v(V_QLO + kq)}, {v(V_QLO + kq)}, {v(V_T + 1)}")
        e(f"v_lshl_add_u32 {v(V_QLO + kq)}, {v(V_QLO + kq)}, 4, {v(V_RS)}")
    e(f"s_lshl_b32 {s(S_TMP + 1)}, {s(S_QRS2)}, 2")
    e(f"s_mov_b32 {s(S_T)}, 0")
    # ---- round 1: Q -> staging area, dO -> this wave's part of the ring ----
    row_dma(g, "qbase", S_QST, "q")
    row_dma(g, "dobase", S_RST, "do")
    # lse of this lane's rows (clamped), while the rows travel
    for b in range(2):
        e(f"v_and_b32 {v(V_U + 2)}, 31, {v(V_LANE)}")
        e(f"v_add_u32 {v(V_U + 2)}, {32 * b}, {v(V_U + 2)}")
        e(f"s_add_i32 {s(S_TMP2)}, {s(S_NVALID)}, -1")
        e(f"s_max_i32 {s(S_TMP2)}, {s(S_TMP2)}, 0")
        e(f"v_min_i32 {v(V_U + 2)}, {v(V_U + 2)}, {s(S_TMP2)}")
        e(f"v_mul_lo_u32 {v(V_X + b)}, {v(V_U + 2)}, {s(S_FST)}")              # byte offset of the row in lse / D / lse2
        e(f"global_load_dword {v(V_LSE2 + b)}, {v(V_X + b)}, %[lsebase]")
    for i in range(128):
        e(f"v_accvgpr_write_b32 {a(A_DQ + i)}, 0")
    e("s_waitcnt vmcnt(0)")
    for b in range(2):
        e(f"v_mul_f32 {v(V_LSE2 + b)}, 0x3fb8aa3b, {v(V_LSE2 + b)}")          # lse * log2(e)
    # fragment addresses of the staged rows: lane (j, hh) takes chunk 2 st + hh of rows j and 32 + j
    QR = [V_U + 2 + i for i in range(8)]
    e(f"v_and_b32 {v(V_T)}, 31, {v(V_LANE)}")
    e(f"v_and_b32 {v(V_T + 1)}, 3, {v(V_T)}")
    e(f"v_lshlrev_b32 {v(V_T + 1)}, 2, {v(V_T + 1)}")
    e(f"v_bfe_u32 {v(V_T + 2)}, {v(V_T)}, 2, 2")
    e(f"v_or_b32 {v(V_T + 1)}, {v(V_T + 1)}, {v(V_T + 2)}")
    e(f"v_lshrrev_b32 {v(V_T + 2)}, 5, {v(V_LANE)}")
    e(f"v_xor_b32 {v(V_T + 1)}, {v(V_T + 1)}, {v(V_T + 2)}")
    e(f"v_lshlrev_b32 {v(V_T)}, 8, {v(V_T)}")
    e(f"v_lshl_or_b32 {v(V_T)}, {v(V_T + 1)}, 4, {v(V_T)}")
    for st in range(8):
        e(f"v_xor_b32 {v(QR[st])}, {st << 5}, {v(V_T)}")
    # Q -> a[128:191]; dO -> a[192:255] and, block by block, into VGPRs next to O for D
    n = 0
    for b in range(2):
        for st in range(8):
            e(f"v_add_u32 {v(V_T + 1)}, {s(S_QST)}, {v(QR[st])}")
            g.lds(f"ds_read_b128 {ar(A_Q + 32 * b + 4 * st, 4)}, {v(V_T + 1)} offset:{8192 * b}", ("q", n))
            e(f"v_add_u32 {v(V_T + 2)}, {s(S_RST)}, {v(QR[st])}")
            g.lds(f"ds_read_b128 {ar(A_DO + 32 * b + 4 * st, 4)}, {v(V_T + 2)} offset:{8192 * b}", ("o", n))
            n += 1
            if len(g.out) >= 12:
                g.need(g.out[3])
    g.drain()
    # ---- round 2: O -> staging area (Q has been read); and, once EVERY wave has read its dO rows out of the ring, the first K / V tiles:
    # issued behind O, so that the counted wait below leaves them in flight under the D arithmetic ----
    row_dma(g, "obase", S_QST, "o")
    # D's second operand: dO once more, into VGPRs this time (block A now, block B below) -- read BEFORE the ring is overwritten
    DOV, OV = V_S, V_DP                                      # 32 registers each: eight 16-byte chunks
    DOV2 = V_KV                                              # block B's dO chunks wait in the K / V / K^T fragment registers (32)
    for b, dst in ((0, DOV), (1, DOV2)):
        for st in range(8):
            e(f"v_add_u32 {v(V_T + 1)}, {s(S_RST)}, {v(QR[st])}")
            g.lds(f"ds_read_b128 {vr(dst + 4 * st, 4)}, {v(V_T + 1)} offset:{8192 * b}", ("dv", b, st))
            if len(g.out) >= 12:
                g.need(g.out[3])
    g.drain()
    e("s_barrier")                                           # ... by every wave: now the ring belongs to K / V
    dma_tile(g, "k", 0, S_T, "pk0")
    dma_tile(g, "v", 0, S_T, "pv0")
    e("s_waitcnt vmcnt(8)")                                  # the sixteen O slices have landed; K(0) / V(0) stay in flight
    # D = sum_d dO O of this lane's rows: the lane's 64 features in the order of attn_bwd_dq_kernel (k-step, dword, low half then high),
    # then the two lane halves (hh = 0: own + other; hh = 1: other + own)
    for b, dreg0 in ((0, DOV), (1, DOV2)):
        for st in range(8):
            e(f"v_add_u32 {v(V_T + 2)}, {s(S_QST)}, {v(QR[st])}")
            g.lds(f"ds_read_b128 {vr(OV + 4 * st, 4)}, {v(V_T + 2)} offset:{8192 * b}", ("ov", b, st))
        g.drain()
        acc = V_DSUM + b
        e(f"v_mov_b32 {v(acc)}, 0")
        for st in range(8):
            for i in range(4):
                dreg, oreg = dreg0 + 4 * st + i, OV + 4 * st + i
                e(f"v_lshlrev_b32 {v(V_T + 1)}, 16, {v(dreg)}")
                e(f"v_lshlrev_b32 {v(V_T + 2)}, 16, {v(oreg)}")
                e(f"v_and_b32 {v(V_T + 3)}, 0xffff0000, {v(dreg)}")
                e(f"v_and_b32 {v(V_T + 4)}, 0xffff0000, {v(oreg)}")
                e(f"v_fma_f32 {v(acc)}, {v(V_T + 1)}, {v(V_T + 2)}, {v(acc)}")
                e(f"v_fma_f32 {v(acc)}, {v(V_T + 3)}, {v(V_T + 4)}, {v(acc)}")
        e(f"v_mov_b32 {v(V_T + 1)}, {v(acc)}")
        e(f"v_mov_b32 {v(V_T + 2)}, {v(acc)}")
        e("s_nop 1")
        e(f"v_permlane32_swap_b32 {v(V_T + 1)}, {v(V_T + 2)}")               # T1 = [lo, lo], T2 = [hi, hi]
        e(f"v_add_f32 {v(acc)}, {v(V_T + 1)}, {v(V_T + 2)}")                  # hh = 0: own + other; hh = 1: other + own  (= lo + hi in both)
    # D and lse2 of the rows that exist, by the hh = 0 lanes
    for b in range(2):
        e(f"v_and_b32 {v(V_T + 1)}, 31, {v(V_LANE)}")
        e(f"v_add_u32 {v(V_T + 1)}, {32 * b}, {v(V_T + 1)}")
        e(f"v_cmp_gt_i32 vcc, {s(S_NVALID)}, {v(V_T + 1)}")
        e(f"v_cmp_gt_u32 {sr(S_TMP)}, 32, {v(V_LANE)}")
        e(f"s_and_b64 exec, vcc, {sr(S_TMP)}")
        e(f"global_store_dword {v(V_X + b)}, {v(V_DSUM + b)}, %[dvecbase]")
        e(f"global_store_dword {v(V_X + b)}, {v(V_LSE2 + b)}, %[lse2base]")
        e(f"s_mov_b64 exec, {sr(S_EXEC)}")
    # ---- tile loop ----
    if BAR_MID:
        assert not g.out, g.out
        e("s_waitcnt vmcnt(0)")                               # K(0) / V(0) by every wave
        e("s_barrier")
        for ins, tag_ in first_reads(0, 0):
            if "lds" not in KO:
                g.lds(ins, tag_)
        while g.out and g.out != carry_reads():                # (the step bodies count the in-flight reads from this state)
            g.need(g.out[0])
    gen_step(g, 0)
    gen_step(g, 1)
    # ---- epilogue ----
    g.label("Lepi")
    g.out = carry_reads()
    e(f"s_cmp_eq_u32 {s(S_NW)}, {s(S_NT)}")
    e(f"s_cbranch_scc0 {g.lref('Lnodrain')}")
    e(f"s_cmp_gt_i32 {s(S_NW)}, 0")
    e(f"s_cbranch_scc0 {g.lref('Lnodrain')}")
    drain(g)
    g.label("Lnodrain")
    e("s_nop 15")
    e("s_barrier")                                           # every wave is done with the ring: it becomes the output staging
    WA, RD, AD = V_T, V_T + 1, V_T + 2
    e(f"v_and_b32 {v(WA)}, 31, {v(V_LANE)}")
    e(f"v_lshlrev_b32 {v(WA)}, 8, {v(WA)}")
    e(f"v_add_u32 {v(WA)}, {v(WA)}, {v(V_HH8)}")
    e(f"v_and_b32 {v(AD)}, 15, {v(V_LANE)}")
    e(f"v_lshlrev_b32 {v(AD)}, 4, {v(AD)}")
    e(f"v_xor_b32 {v(WA)}, {v(WA)}, {v(AD)}")
    e(f"v_add_u32 {v(WA)}, {s(S_RST)}, {v(WA)}")
    e(f"v_lshlrev_b32 {v(RD)}, 4, {v(V_LANE)}")
    e(f"v_add_u32 {v(RD)}, {s(S_RST)}, {v(RD)}")
    tmp = [V_U + i for i in range(4)]
    assert tmp[0] % 2 == 0
    for b in range(2):
        for dt in range(4):
            for g4 in range(4):
                base = A_DQ + 64 * b + 16 * dt + 4 * g4
                for i in range(4):
                    e(f"v_accvgpr_read_b32 {v(tmp[i])}, {a(base + i)}")
                for i in range(4):
                    e(f"v_mul_f32 {v(tmp[i])}, {s(S_SCALE)}, {v(tmp[i])}")
                e(f"v_cvt_pk_bf16_f32 {v(tmp[0])}, {v(tmp[0])}, {v(tmp[1])}")
                e(f"v_cvt_pk_bf16_f32 {v(tmp[1])}, {v(tmp[2])}, {v(tmp[3])}")
                e(f"v_xor_b32 {v(AD)}, {(4 * dt + g4) << 4}, {v(WA)}")
                e(f"ds_write_b64 {v(AD)}, {vr(tmp[0], 2)} offset:{8192 * b}")
    e("s_waitcnt lgkmcnt(0)")
    VO = [V_KR, V_KR + 1, V_X, V_X + 1]
    e(f"v_and_b32 {v(AD)}, 15, {v(V_LANE)}")
    for kq in range(4):
        e(f"v_add_u32 {v(VO[kq])}, {4 * kq}, {v(V_LANE4)}")
        e(f"v_xor_b32 {v(VO[kq])}, {v(VO[kq])}, {v(AD)}")
        e(f"v_lshl_add_u32 {v(VO[kq])}, {v(VO[kq])}, 4, {v(V_RS)}")
    e(f"s_mov_b64 {sr(S_TMP)}, %[dqbase]")
    e(f"s_lshl_b32 {s(S_TMP2)}, {s(S_QRS2)}, 2")
    e(f"s_mov_b32 {s(S_TMP2 + 1)}, {s(S_NVALID)}")
    for i in range(12):
        e(f"ds_read_b128 {vr(V_S + 4 * i, 4)}, {v(RD)} offset:{1024 * i}")
    e("s_waitcnt lgkmcnt(8)")
    for i in range(12, 16):
        e(f"ds_read_b128 {vr(V_S + 4 * i, 4)}, {v(RD)} offset:{1024 * i}")
    for i in range(16):
        if i >= 4:
            e(f"s_waitcnt lgkmcnt({15 - i})")
        e(f"v_cmp_gt_i32 vcc, {s(S_TMP2 + 1)}, {v(V_LANE4)}")
        e("s_mov_b64 exec, vcc")
        e(f"global_store_dwordx4 {v(VO[i % 4])}, {vr(V_S + 4 * i, 4)}, {sr(S_TMP)}")
        e(f"s_mov_b64 exec, {sr(S_EXEC)}")
        e(f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, {s(S_TMP2)}")
        e(f"s_addc_u32 {s(S_TMP + 1)}, {s(S_TMP + 1)}, 0")
        e(f"s_add_i32 {s(S_TMP2 + 1)}, {s(S_TMP2 + 1)}, -4")
    e(f"s_mov_b32 m0, {s(S_M0SAVE)}")
    return g


def clobbers():
    names = [f"v{i}" for i in range(FIRST_V, LAST_V + 1)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(FIRST_S, LAST_S + 1)]
    return names + ["vcc", "scc", "memory"]


def main():
    g = gen_body()
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_attn_bwd_dq64.py -- do not edit; the per-item body of attn_bwd_dq64_kernel as one inline-asm statement.\n")
        f.write(f"// {len(g.lines)} lines; options {OPT}\n")
        f.write("#define VSEL_DQ64_ASM_TEXT \\\n")
        for ln in g.lines:
            f.write(f'  "{ln}\\n\\t" \\\n')
        f.write('  ""\n')
        f.write("#define VSEL_DQ64_ASM_CLOBBERS \\\n  ")
        f.write(", ".join(f'"{c}"' for c in clobbers()))
        f.write("\n")
    n_mfma = sum(1 for ln in g.lines if ln.startswith("v_mfma"))
    print(f"wrote {OUT}: {len(g.lines)} instructions / labels, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
