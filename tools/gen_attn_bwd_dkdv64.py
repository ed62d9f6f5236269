#!/usr/bin/env python3
"""Generator of the hand-scheduled per-item body of attn_bwd_dkdv64_kernel (visionselector_amd/csrc/attn_bwd_dkdv64.hip).

    python tools/gen_attn_bwd_dkdv64.py          # rewrites visionselector_amd/csrc/attn_bwd_dkdv64_body.inc

The dK / dV pass of the attention backward as ONE 512-register wave per SIMD with the MFMA stream software-pipelined inside the wave
(the structure of tools/gen_attn_fwd64.py / gen_attn_bwd_dq64.py).  Work split and arithmetic = attn_bwd_dkdv_kernel<false> (attn_bwd.hip,
the four-wave form): item = (128-key block, kv head, sequence), wave w owns keys 32 w .. 32 w + 31 with its K / V fragments in registers
and loops over the q heads of the group and over 64-query Q / dO tiles in LDS; per tile, for the query blocks qb = 0, 1:
S = Q K^T and dP = dO V^T (lane = key), P = exp2(S c - lse2[q]), dS = P (dP - D[q]), dV^T += dO^T P, dK^T += Q^T dS -- the same MFMAs
per accumulator in the same order, so dK / dV are bit-identical to that kernel's.

Units.  A (tile, query block) unit = 16 MFMAs "SdP" (S / dP alternating; Q / dO row fragments from LDS), 80 VALU and 16 MFMAs "dVdK"
(dV / dK alternating; dO^T / Q^T fragments by transposed reads).  MFMA stream of tile t:
    SdP(t,0) dVdK(t-1,1) SdP(t,1) dVdK(t,0)        [dVdK(t,1) runs in step t + 1: the Q / dO rings have THREE slots]
a unit's VALU runs in the gaps of the 32 MFMAs that follow its SdP (the other query block's): first the 32 operations that read the
per-query lse2 / D (so those registers are free for the next unit's broadcast ds_read_b128 in the first gaps of ITS SdP), then exp / dS /
conversions.  The step's barrier stands in front of its last batch (every wave is then past the slot the next loads overwrite, the next
tile has landed), whose gaps carry the next tile's first row fragments across the step boundary.
Per wave and tile the 64 MFMAs (2048 cycles) want 64 KiB of operands from LDS -- 16 KiB each of Q, dO rows (ds_read_b128) and of
Q^T, dO^T (ds_read_b64_tr_b16, half the rate per byte) -- i.e. ~1800 LDS cycles per CU and tile with the direct-to-LDS writes: the LDS
pipe is as busy as the MFMA pipe, and with <= 15 reads in flight per wave the schedule variants (options below) move the time by 1 - 3 %.

Register map (per lane)
  a[0:63] dK^T accumulators (d-tile dt: + 16 dt), a[64:127] dV^T, a[128:159] K fragments (k-step st: + 4 st), a[160:191] V fragments,
  a[192:223] Q / dO row fragments (8 slots of 4), a[224:255] dO^T / Q^T fragments (8 slots of 4)
  v[32:63] S (query block qb: + 16 qb), v[64:95] dP, v[96:111] P bf16 (qb, 16-query half m: + 8 qb + 4 m), v[112:127] dS bf16,
  v[128:143] lse2 of the unit's 16 query rows of this lane half (m: + 8 m), v[144:159] D
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_attn_fwd64 import Gen, v, vr, a, ar, s, sr, place, spread            # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("DKDV64_OUT", os.path.join(ROOT, "visionselector_amd", "csrc", "attn_bwd_dkdv64_body.inc"))

A_DK, A_DV, A_KF, A_VF, A_ROW, A_TR = 0, 64, 128, 160, 192, 224
V_S, V_DP, V_P, V_DS, V_L, V_D = 32, 64, 96, 112, 128, 144
V_RA, V_TR = 160, 168
V_T = 176                                         # T0..T7
V_LO = 184                                        # [4] per-lane byte offsets of the four Q / dO slices a wave loads per tile
V_P8, V_LANE4, V_LANE, V_HH8, V_NEGINF = 188, 189, 190, 191, 192
V_KREL = 193                                      # kw0 + j - 8 hh   (this lane's key, relative; the tile's query base comes off per tile)
V_QLIM = 194                                      # len - 8 hh
V_REL, V_QL = 195, 196                            # per (unit, m): key_rel and query limit against the element index e
V_LA = 197                                        # LDS address of lse slot 0 + 32 hh bytes
V_U = 198                                         # U0..U9
V_X = 232                                         # [16] four more transposed-fragment slots (trx=1)
V_RA2, V_TR2 = 212, 220                           # the same fragment addresses inside the dO ring (ds offsets are 16 bits: the ring base rides in the register)
FIRST_V, LAST_V = 32, 255

S_IT, S_NIT, S_TPH, S_QBEG, S_LEN, S_SL2, S_LDSW, S_W4 = 40, 41, 42, 43, 44, 45, 46, 47
S_LQT, S_LHEAD = 48, 49                           # (query tile, head) of the NEXT tile to load
S_CQT = 50                                        # query tile being processed
S_SCALE, S_QRS2, S_FST = 51, 52, 53               # scale; bytes per row of q / dO; bytes per row of lse2 / D
S_QB0, S_DOB0 = 54, 56                            # 64-bit: q / dO at (sequence start, head 0)
S_LB0, S_DB0 = 58, 60                             # 64-bit: lse2 / D at (sequence start, head 0)
S_TMP = 62                                        # 62..65
S_EXEC = 66
S_RING, S_KW0, S_CAUSAL, S_LCNT = 68, 69, 70, 71  # LDS base; first key of this wave; causal flag; tiles loaded so far
S_PTR = 72                                        # 72..73
S_M0SAVE, S_LENM1 = 74, 75
S_TMP2 = 76                                       # 76..79
S_NF = 80                                         # number of leading steps that run the steady form
S_WAVE = 81
S_PTR2 = 82                                       # 82..83
S_FLAG = 84                                       # this tile needs the mask for this wave
S_SL2P = 86                                       # 86..87: sl2 twice (operand pair of the packed fma)
S_NS0 = 42                                        # order=td: first step that may run the steady form (s42 is otherwise unused)
S_H0, S_HEND, S_CH, S_NH = 85, 88, 89, 90         # order=td: the item's first head, one past its last, heads done on the compute side's tile, head count
S_STEP = 91                                       # order=td: -64 (tiles from the end downward) or +64 (%[dirup]: from the item's first query upward)
FIRST_S, LAST_S = 40, 91

KBUF = 16384
NQ = 3                                            # ring slots of Q and of dO
LSE_BASE = 2 * NQ * KBUF                          # lse2[NQ][64] floats, then D[NQ][64]
D_BASE = LSE_BASE + NQ * 256
LDS_BYTES = D_BASE + NQ * 256

# defaults = the measured best (profiles/r04_dkdv64_ab.txt); bar=top, valu=skew, split=24, tail=11 is the first working form
OPT = {"dma": "1,3,5,7,9,11,13,15", "split": "40", "ko": "", "bar": "mid", "valu": "ab", "trx": "0", "tail": "13", "fw": "0", "fr": "9", "pk": "0",
       "order": "td"}
for kv in os.environ.get("DKDV64_OPTS", "").split(","):
    if "=" in kv:
        key, val = kv.split("=", 1)
        assert key in OPT, key
        OPT[key] = val.replace("/", ",")
VALU_SPLIT = int(OPT["split"])
KO = set(OPT["ko"].split("+")) - {""}
VALU_AB = OPT["valu"] == "ab"      # a unit's VALU as stage A (the 32 operations that read lse2 / D) then stage B (exp, dS, conversions)
TRX = OPT["trx"] == "1"            # the m = 0 fragments of d-tiles 2, 3 in four arch-VGPR slots: every transposed read is issued >= 7 MFMAs ahead
PK = OPT["pk"] == "1"              # packed fp32 VALU (v_pk_fma / add / mul_f32: two registers per instruction, the same IEEE operations): a wave alone
                                   # on its SIMD issues one instruction per 4 cycles, so the instruction COUNT beside the MFMAs is what costs
FINE_WAITS = OPT["fw"] == "1"     # SdP waits per row fragment instead of per group of them
FR_GAP = int(OPT["fr"])            # first dVdK gap that carries the next SdP's first row reads (two per gap)
TAIL_HI = int(OPT["tail"])         # last SdP gap that takes VALU of the previous unit
ORDER_TD = OPT["order"] == "td"   # tiles outer from the sequence's END downward, the item's heads inner (ht: heads outer, tiles ascending): the
                                   # items of one sequence (key blocks, causal: each starts at its own first query) then sit on the SAME Q / dO
                                   # tile at the same time, and a ragged sequence's one partial tile comes first, every later step steady
BAR_STAG = OPT["bar"] == "stag"    # bar=mid with waves 2, 3 at the barrier one batch earlier: they then run one batch behind waves 0, 1, so
                                   # two waves are in an SdP batch (row reads) while two are in a dVdK batch (half-rate transposed reads)
BAR_MID = OPT["bar"] in ("mid", "stag")      # the step's barrier in front of its last batch, whose gaps then carry the NEXT tile's first reads


# ---- streams ----------------------------------------------------------------------------------------------------------------------
def row_read(tensor, qb, st, slot):
    """Q / dO row fragment (query block qb, k-step st) of ring slot `slot` into fragment slot (st & 3) + (4 if dO)"""
    fs = (st & 3) + (4 if tensor == "do" else 0)
    off = slot * KBUF + qb * 8192
    return (f"ds_read_b128 {ar(A_ROW + 4 * fs, 4)}, {v((V_RA if tensor == 'q' else V_RA2) + st)} offset:{off}", (tensor, qb, st))


def sdp_mfma(g, i, qb):
    st, is_dp = i // 2, i % 2
    if FINE_WAITS:
        g.need(("do" if is_dp else "q", qb, st))
    elif not is_dp and st in (0, 4, 6):
        g.need(("do", qb, {0: 3, 4: 5, 6: 7}[st]))
    frag = ar(A_ROW + 4 * ((st & 3) + (4 if is_dp else 0)), 4)
    dst = vr((V_DP if is_dp else V_S) + 16 * qb, 16)
    g.e(f"v_mfma_f32_32x32x16_bf16 {dst}, {frag}, {ar((A_VF if is_dp else A_KF) + 4 * st, 4)}, {'0' if st == 0 else dst}")


def first_reads(qb, slot):
    return [row_read(t, qb, st, slot) for st in range(4) for t in ("q", "do")]


def second_reads(qb, slot):
    out = []
    for st in range(4):
        out.append((2 * st + 1, row_read("q", qb, st + 4, slot)))
        out.append((2 * st + 2, row_read("do", qb, st + 4, slot)))
    return out


def tr_frag(m, dt, k, lo=0, n=4):
    """registers of the transposed fragment (16-query half m, d-tile dt, k = 0 dO^T / 1 Q^T).  Eight accumulator-file slots: a half's
    four d-tiles, a slot pair free one MFMA pair behind its m = 0 use; with trx, m = 0 keeps d-tiles 2, 3 in arch VGPRs and m = 1 has
    slots 4 .. 7 (d-tiles 0, 1) and 0 .. 3 (d-tiles 2, 3) to itself."""
    if not TRX:
        return ar(A_TR + 4 * (2 * dt + k) + lo, n)
    if m == 0:
        return ar(A_TR + 4 * (2 * dt + k) + lo, n) if dt < 2 else vr(V_X + 4 * (2 * (dt - 2) + k) + lo, n)
    return ar(A_TR + 16 + 4 * (2 * dt + k) + lo, n) if dt < 2 else ar(A_TR + 4 * (2 * (dt - 2) + k) + lo, n)


def tr_reads(qb, m, dt, slot):
    """the four transposed reads of (query block, 16-query half, d-tile): dO^T fragment into slot 2 (dt & 3 ...) and Q^T behind it"""
    out = []
    for tensor, k in (("do", 0), ("q", 1)):
        for hi in range(2):
            off = slot * KBUF + (32 * qb + 16 * m) * 256
            out.append((f"ds_read_b64_tr_b16 {tr_frag(m, dt, k, 2 * hi, 2)}, {v((V_TR if tensor == 'q' else V_TR2) + 2 * dt + hi)} offset:{off}",
                        ("t" + tensor, qb, m, dt, hi)))
    return out


def dvdk_mfma(g, i, qb):
    """MFMA i of a dVdK batch: (m, dt) = (i // 8, (i % 8) // 2); even = dV (dO^T x P), odd = dK (Q^T x dS)"""
    m, dt, is_dk = i // 8, (i % 8) // 2, i % 2
    if not is_dk:
        g.need(("tq", qb, m, dt, 1))                         # the pair's four reads (dO^T then Q^T)
    acc = ar((A_DK if is_dk else A_DV) + 16 * dt, 16)
    frag = tr_frag(m, dt, 1 if is_dk else 0)
    b = vr((V_DS if is_dk else V_P) + 8 * qb + 4 * m, 4)
    g.e(f"v_mfma_f32_32x32x16_bf16 {acc}, {frag}, {b}, {acc}")


def unit_valu(qb):
    """p = exp2(s c - lse2[q]) in place, dS = bf16(p (dP - D[q])), P = bf16(p): 80 VALU, skewed (no result used within two instructions)"""
    sb, db = V_S + 16 * qb, V_DP + 16 * qb
    ops = []
    if PK:
        assert VALU_AB and sb % 2 == 0 and db % 2 == 0 and V_L % 2 == 0 and V_D % 2 == 0
        for j in range(8):
            i = 2 * j
            ops.append(f"v_pk_fma_f32 {vr(sb + i, 2)}, {vr(sb + i, 2)}, {sr(S_SL2P)}, {vr(V_L + i, 2)} neg_lo:[0,0,1] neg_hi:[0,0,1]")
            ops.append(f"v_pk_add_f32 {vr(db + i, 2)}, {vr(db + i, 2)}, {vr(V_D + i, 2)} neg_lo:[0,1] neg_hi:[0,1]")
        for j in range(8 + 2):
            if j < 8:
                ops.append(f"v_exp_f32 {v(sb + 2 * j)}, {v(sb + 2 * j)}")
                ops.append(f"v_exp_f32 {v(sb + 2 * j + 1)}, {v(sb + 2 * j + 1)}")
            r = 2 * (j - 1)
            if 0 <= r < 16:
                ops.append(f"v_pk_mul_f32 {vr(db + r, 2)}, {vr(sb + r, 2)}, {vr(db + r, 2)}")
                ops.append(f"v_cvt_pk_bf16_f32 {v(V_P + 8 * qb + 4 * (r >> 3) + ((r & 7) >> 1))}, {v(sb + r)}, {v(sb + r + 1)}")
            r = 2 * (j - 2)
            if 0 <= r < 16:
                ops.append(f"v_cvt_pk_bf16_f32 {v(V_DS + 8 * qb + 4 * (r >> 3) + ((r & 7) >> 1))}, {v(db + r)}, {v(db + r + 1)}")
        assert len(ops) == 56 and VALU_SPLIT >= 16
        return ops
    if VALU_AB:
        for i in range(16):
            ops.append(f"v_fma_f32 {v(sb + i)}, {v(sb + i)}, {s(S_SL2)}, -{v(V_L + i)}")
            ops.append(f"v_sub_f32 {v(db + i)}, {v(db + i)}, {v(V_D + i)}")
        for i in range(16 + 5):
            if i < 16:
                ops.append(f"v_exp_f32 {v(sb + i)}, {v(sb + i)}")
            r = i - 2
            if 0 <= r < 16:
                ops.append(f"v_mul_f32 {v(db + r)}, {v(sb + r)}, {v(db + r)}")
            r = i - 3
            if 0 <= r < 16 and r % 2 == 1:
                ops.append(f"v_cvt_pk_bf16_f32 {v(V_P + 8 * qb + 4 * (r >> 3) + ((r & 7) >> 1))}, {v(sb + r - 1)}, {v(sb + r)}")
            r = i - 4
            if 0 <= r < 16 and r % 2 == 1:
                ops.append(f"v_cvt_pk_bf16_f32 {v(V_DS + 8 * qb + 4 * (r >> 3) + ((r & 7) >> 1))}, {v(db + r - 1)}, {v(db + r)}")
        assert len(ops) == 80 and VALU_SPLIT >= 32          # (lse2 / D registers are free once the head has run)
        return ops
    for i in range(16 + 8):
        if i < 16:
            ops.append(f"v_fma_f32 {v(sb + i)}, {v(sb + i)}, {s(S_SL2)}, -{v(V_L + i)}")
        r = i - 1
        if 0 <= r < 16:
            ops.append(f"v_sub_f32 {v(db + r)}, {v(db + r)}, {v(V_D + r)}")
        r = i - 3
        if 0 <= r < 16:
            ops.append(f"v_exp_f32 {v(sb + r)}, {v(sb + r)}")
        r = i - 5
        if 0 <= r < 16:
            ops.append(f"v_mul_f32 {v(db + r)}, {v(sb + r)}, {v(db + r)}")
        r = i - 6
        if 0 <= r < 16 and r % 2 == 1:
            ops.append(f"v_cvt_pk_bf16_f32 {v(V_P + 8 * qb + 4 * (r >> 3) + ((r & 7) >> 1))}, {v(sb + r - 1)}, {v(sb + r)}")
        r = i - 7
        if 0 <= r < 16 and r % 2 == 1:
            ops.append(f"v_cvt_pk_bf16_f32 {v(V_DS + 8 * qb + 4 * (r >> 3) + ((r & 7) >> 1))}, {v(db + r - 1)}, {v(db + r)}")
    assert len(ops) == 80
    return ops


def ld_reads(qb, slot):
    """lse2 / D of the unit's query rows: registers r = 8 m + e hold rows 32 qb + 16 m + 8 hh + e (the lane half comes through V_LA)"""
    out = []
    for m in range(2):
        for h in range(2):
            off = slot * 256 + (32 * qb + 16 * m) * 4 + 16 * h                 # (V_LA points at lse2 slot 0)
            out.append((f"ds_read_b128 {vr(V_L + 8 * m + 4 * h, 4)}, {v(V_LA)} offset:{off}", ("l", qb, m, h)))
            out.append((f"ds_read_b128 {vr(V_D + 8 * m + 4 * h, 4)}, {v(V_LA)} offset:{off + D_BASE - LSE_BASE}", ("d", qb, m, h)))
    return out


def mask_unit(g, qb, uniq):
    """S <- -inf where the key is not visible to the query (causal) or the query row does not exist: register 8 m + e is masked iff
    e < rel or e >= qlim, rel = key - (qt + 32 qb + 16 m + 8 hh), qlim = len - (qt + 32 qb + 16 m + 8 hh).  Only tiles that
    touch the causal diagonal of this wave's keys or the end of the sequence come here (s[S_FLAG] != 0)."""
    g.e(f"s_cmp_lg_u32 {s(S_FLAG)}, 0")
    g.e(f"s_cbranch_scc0 {g.lref('Lnm' + uniq)}")
    g.e("s_nop 15")
    for m in range(2):
        g.e(f"s_add_i32 {s(S_TMP2)}, {s(S_CQT)}, {32 * qb + 16 * m}")
        g.e(f"v_sub_u32 {v(V_REL)}, {v(V_KREL)}, {s(S_TMP2)}")
        g.e(f"v_sub_u32 {v(V_QL)}, {v(V_QLIM)}, {s(S_TMP2)}")
        g.e(f"s_cmp_lg_u32 {s(S_CAUSAL)}, 0")                                # not causal: no key bound
        g.e(f"s_cbranch_scc1 {g.lref('Lc' + uniq + str(m))}")
        g.e(f"v_mov_b32 {v(V_REL)}, 0xc0000000")
        g.label("Lc" + uniq + str(m))
        for e in range(8):
            reg = V_S + 16 * qb + 8 * m + e
            g.e(f"v_cmp_gt_i32 {sr(S_TMP)}, {v(V_REL)}, {e}")                 # rel > e: the key lies behind this query
            g.e(f"v_cmp_le_i32 vcc, {v(V_QL)}, {e}")                          # qlim <= e: the query row is padding
            g.e(f"s_or_b64 vcc, vcc, {sr(S_TMP)}")
            g.e(f"v_cndmask_b32 {v(reg)}, {v(reg)}, {v(V_NEGINF)}, vcc")
    g.label("Lnm" + uniq)


def issue(g, text, tag):
    """an LDS read with its wait bookkeeping (timing-only knock-outs: ko=row / tr / ld leave a class of reads out)"""
    kind = "tr" if tag[0] in ("tq", "tdo") else "row" if tag[0] in ("q", "do") else "ld"
    if kind in KO:
        g.strict = False
        return
    if len(g.out) >= 14:                                      # lgkmcnt counts 15: retire the oldest (long landed) read first
        g.need(g.out[0])
    g.lds(text, tag)


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
                issue(g, ins[0], ins[1])
            elif "dma" in KO and ins.startswith("global_load_lds"):
                pass
            elif "valu" in KO and ins.split()[0] in ("v_fma_f32", "v_sub_f32", "v_exp_f32", "v_mul_f32", "v_cvt_pk_bf16_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32"):
                pass
            else:
                g.e(ins)


def tile_flags(g):
    """s[S_FLAG] = this tile needs the mask for this wave: (causal and kw0 + 31 > qt) or qt + 64 > len"""
    e = g.e
    e(f"s_add_i32 {s(S_TMP2)}, {s(S_KW0)}, 31")
    e(f"s_cmp_gt_i32 {s(S_TMP2)}, {s(S_CQT)}")
    e(f"s_cselect_b32 {s(S_FLAG)}, {s(S_CAUSAL)}, 0")
    e(f"s_add_i32 {s(S_TMP2)}, {s(S_CQT)}, 64")
    e(f"s_cmp_gt_i32 {s(S_TMP2)}, {s(S_LEN)}")
    e(f"s_cselect_b32 {s(S_TMP2 + 1)}, 1, 0")
    e(f"s_or_b32 {s(S_FLAG)}, {s(S_FLAG)}, {s(S_TMP2 + 1)}")


def advance_cqt(g):
    e = g.e
    if ORDER_TD:
        e(f"s_add_i32 {s(S_CH)}, {s(S_CH)}, 1")
        e(f"s_cmp_eq_u32 {s(S_CH)}, {s(S_NH)}")
        e(f"s_cselect_b32 {s(S_TMP2)}, {s(S_STEP)}, 0")
        e(f"s_cselect_b32 {s(S_CH)}, 0, {s(S_CH)}")
        e(f"s_add_i32 {s(S_CQT)}, {s(S_CQT)}, {s(S_TMP2)}")
        return
    e(f"s_add_i32 {s(S_CQT)}, {s(S_CQT)}, 64")
    e(f"s_cmp_ge_i32 {s(S_CQT)}, {s(S_LEN)}")
    e(f"s_cselect_b32 {s(S_CQT)}, {s(S_QBEG)}, {s(S_CQT)}")


def load_addr(g):
    """s[S_PTR] / s[S_PTR2] = Q / dO address of row 4 * wave of the next tile to load: base(head 0) + (qt + 4 wave) * row bytes + head * 256"""
    e = g.e
    e(f"s_add_i32 {s(S_TMP + 2)}, {s(S_LQT)}, {s(S_W4)}")
    e(f"s_mul_i32 {s(S_TMP)}, {s(S_TMP + 2)}, {s(S_QRS2)}")
    e(f"s_mul_hi_u32 {s(S_TMP + 1)}, {s(S_TMP + 2)}, {s(S_QRS2)}")
    e(f"s_lshl_b32 {s(S_TMP + 2)}, {s(S_LHEAD)}, 8")
    e(f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, {s(S_TMP + 2)}")
    e(f"s_addc_u32 {s(S_TMP + 1)}, {s(S_TMP + 1)}, 0")
    e(f"s_add_u32 {s(S_PTR)}, {s(S_QB0)}, {s(S_TMP)}")
    e(f"s_addc_u32 {s(S_PTR + 1)}, {s(S_QB0 + 1)}, {s(S_TMP + 1)}")
    e(f"s_add_u32 {s(S_PTR2)}, {s(S_DOB0)}, {s(S_TMP)}")
    e(f"s_addc_u32 {s(S_PTR2 + 1)}, {s(S_DOB0 + 1)}, {s(S_TMP + 1)}")


def load_pieces(slot):
    """a WHOLE tile's eight slices of this wave: [(m0 write, load)]"""
    out = []
    for tensor, ptr in (("q", S_PTR), ("do", S_PTR2)):
        lds0 = (0 if tensor == "q" else NQ * KBUF) + slot * KBUF
        for u in range(4):
            out.append((f"s_add_u32 m0, {s(S_LDSW)}, {lds0 + u * 4096}", f"global_load_lds_dwordx4 {v(V_LO + u)}, {sr(ptr)}"))
    return out


def load_tail(g, slot, uniq):
    """lse2 (wave 0) / D (wave 1) of the tile's 64 query rows (clamped), then the load cursor moves on"""
    e = g.e
    e(f"s_cmp_lt_u32 {s(S_WAVE)}, 2")
    e(f"s_cbranch_scc0 {g.lref('Lnold' + uniq)}")
    e(f"v_add_u32 {v(V_U)}, {s(S_LQT)}, {v(V_LANE)}")
    e(f"v_min_i32 {v(V_U)}, {v(V_U)}, {s(S_LENM1)}")
    e(f"v_mul_lo_u32 {v(V_U)}, {v(V_U)}, {s(S_FST)}")
    e(f"s_lshl_b32 {s(S_TMP2)}, {s(S_LHEAD)}, 2")
    e(f"v_add_u32 {v(V_U)}, {s(S_TMP2)}, {v(V_U)}")
    e(f"s_cmp_eq_u32 {s(S_WAVE)}, 0")
    e(f"s_cselect_b64 {sr(S_TMP)}, {sr(S_LB0)}, {sr(S_DB0)}")
    e(f"s_mov_b32 {s(S_TMP2)}, {LSE_BASE + slot * 256}")
    e(f"s_mov_b32 {s(S_TMP2 + 1)}, {D_BASE + slot * 256}")
    e(f"s_cmp_eq_u32 {s(S_WAVE)}, 0")
    e(f"s_cselect_b32 {s(S_TMP2)}, {s(S_TMP2)}, {s(S_TMP2 + 1)}")
    e(f"s_add_u32 m0, {s(S_RING)}, {s(S_TMP2)}")
    e("s_nop 0")
    e(f"global_load_lds_dword {v(V_U)}, {sr(S_TMP)}")
    g.label("Lnold" + uniq)
    if ORDER_TD:
        e(f"s_add_i32 {s(S_LHEAD)}, {s(S_LHEAD)}, 1")
        e(f"s_cmp_eq_u32 {s(S_LHEAD)}, {s(S_HEND)}")
        e(f"s_cselect_b32 {s(S_TMP2)}, {s(S_STEP)}, 0")
        e(f"s_cselect_b32 {s(S_LHEAD)}, {s(S_H0)}, {s(S_LHEAD)}")
        e(f"s_add_i32 {s(S_LQT)}, {s(S_LQT)}, {s(S_TMP2)}")
        e(f"s_add_i32 {s(S_LCNT)}, {s(S_LCNT)}, 1")
        return
    e(f"s_add_i32 {s(S_LQT)}, {s(S_LQT)}, 64")
    e(f"s_cmp_ge_i32 {s(S_LQT)}, {s(S_LEN)}")
    e(f"s_cselect_b32 {s(S_TMP2)}, 1, 0")
    e(f"s_cselect_b32 {s(S_LQT)}, {s(S_QBEG)}, {s(S_LQT)}")
    e(f"s_add_i32 {s(S_LHEAD)}, {s(S_LHEAD)}, {s(S_TMP2)}")
    e(f"s_add_i32 {s(S_LCNT)}, {s(S_LCNT)}, 1")


def load_tile(g, slot, uniq):
    """the next tile of Q, dO, lse2, D into ring slot `slot`, any tile (a partial one clamps its source rows per lane)"""
    e = g.e
    load_addr(g)
    e(f"s_add_i32 {s(S_TMP2)}, {s(S_LQT)}, 64")
    e(f"s_cmp_gt_i32 {s(S_TMP2)}, {s(S_LEN)}")
    e(f"s_cbranch_scc1 {g.lref('Ltail' + uniq)}")
    for m0w, ld in load_pieces(slot):
        e(m0w)
        e("s_nop 0")
        e(ld)
    e(f"s_branch {g.lref('Lrows' + uniq)}")
    g.label("Ltail" + uniq)
    # clamped form: the pointers go back to the tile's row 0; lane offset = min(4 (wave + 4 u) + (lane >> 4), len - 1 - qt) * row bytes + part
    e(f"s_mul_i32 {s(S_TMP)}, {s(S_W4)}, {s(S_QRS2)}")
    for ptr in (S_PTR, S_PTR2):
        e(f"s_sub_u32 {s(ptr)}, {s(ptr)}, {s(S_TMP)}")
        e(f"s_subb_u32 {s(ptr + 1)}, {s(ptr + 1)}, 0")
    e(f"s_sub_i32 {s(S_TMP2 + 1)}, {s(S_LENM1)}, {s(S_LQT)}")
    for tensor, ptr in (("q", S_PTR), ("do", S_PTR2)):
        lds0 = (0 if tensor == "q" else NQ * KBUF) + slot * KBUF
        for u in range(4):
            e(f"s_add_i32 {s(S_TMP2)}, {s(S_W4)}, {16 * u}")
            e(f"v_add_u32 {v(V_U)}, {s(S_TMP2)}, {v(V_LANE4)}")
            e(f"v_min_i32 {v(V_U)}, {v(V_U)}, {s(S_TMP2 + 1)}")
            e(f"v_mul_lo_u32 {v(V_U)}, {v(V_U)}, {s(S_QRS2)}")
            e(f"v_add_u32 {v(V_U)}, {v(V_U)}, {v(V_P8)}")
            e(f"s_add_u32 m0, {s(S_LDSW)}, {lds0 + u * 4096}")
            e("s_nop 0")
            e(f"global_load_lds_dwordx4 {v(V_U)}, {sr(ptr)}")
    g.label("Lrows" + uniq)
    load_tail(g, slot, uniq)


def tr_plan(pqb, ps):
    """-> (reads riding in the SdP batch in front [(gap, read)], reads in the dVdK batch's own gaps [(gap, read)])"""
    if not TRX:
        pre = list(zip([9, 9, 10, 10, 11, 11, 12, 12], tr_first(pqb, ps)))
        own = [(dt - 2, r) for dt in (2, 3) for r in tr_reads(pqb, 0, dt, ps)]            # the rest of m = 0
        own += [(2 * dt + 2, r) for dt in range(4) for r in tr_reads(pqb, 1, dt, ps)]      # m = 1: behind the slot pair's m = 0 use
        return pre, sorted(own, key=lambda x: x[0])
    m0 = [r for dt in range(4) for r in tr_reads(pqb, 0, dt, ps)]
    pre = list(zip([8 + i // 2 for i in range(16)], m0))                                   # gaps 8 .. 15, a fragment per gap
    own = list(zip([i // 2 for i in range(8)], [r for dt in (0, 1) for r in tr_reads(pqb, 1, dt, ps)]))          # slots 4 .. 7
    own += list(zip([4 + i // 2 for i in range(8)], [r for dt in (2, 3) for r in tr_reads(pqb, 1, dt, ps)]))     # slots 0 .. 3, free after MFMA 3
    return pre, own


def tr_first(pqb, ps):
    """m = 0 fragments of d-tiles 0, 1 (eight reads): issued in the gaps of the SdP batch in front of the dVdK batch that uses them"""
    return [r for dt in range(2) for r in tr_reads(pqb, 0, dt, ps)]


def step_body(g, idx, tag, has_prev, steady):
    """tile it = s[S_IT] in ring slot idx (= it % 3); the previous tile (slot (idx + 2) % 3) still holds the deferred unit's fragments"""
    slot, pslot, nslot = idx, (idx + 2) % 3, (idx + 1) % 3
    e = g.e
    tile_flags(g)
    if not BAR_MID:
        for ins, tg in first_reads(0, slot):
            issue(g, ins, tg)
    if steady:
        load_addr(g)
    for qb in range(2):
        have_prev = qb == 1 or has_prev
        pqb, ps = (0, slot) if qb == 1 else (1, pslot)
        # ---- SdP(t, qb) beside the VALU tail of the previous unit; this unit's lse2 / D reads in the last gaps ----
        gaps = [[] for _ in range(17)]
        for gp, rd in second_reads(qb, slot):
            gaps[gp].append(rd)
        if have_prev:
            tail = unit_valu(1 - qb)[VALU_SPLIT:]
            place(gaps, tail, spread(len(tail), 0, TAIL_HI))
            for gp, rd in tr_plan(pqb, ps)[0]:
                gaps[gp].append(rd)
        if VALU_AB:
            for gp, rd in zip(spread(8, 0, 3), ld_reads(qb, slot)):       # (in front of the gap's other reads: consumed first)
                gaps[gp].insert(0, rd)
        else:
            place(gaps, ld_reads(qb, slot), spread(8, 12, 15))
        if BAR_STAG and qb == 1:
            e(f"s_cmp_lt_u32 {s(S_WAVE)}, 2")
            e(f"s_cbranch_scc1 {g.lref('Lb3' + tag)}")
            e("s_waitcnt vmcnt(0)")
            e("s_barrier")
            g.label("Lb3" + tag)
        if qb == 0 and steady:
            where = [int(x) for x in OPT["dma"].split(",")]
            for w, (m0w, ld) in zip(where, load_pieces(nslot)):
                rest = gaps[w]
                gaps[w] = [m0w] + (rest[:1] if rest else ["s_nop 0"]) + [ld] + rest[1:]
        emit_batch(g, 16, lambda i: sdp_mfma(g, i, qb), gaps)
        if qb == 0 and steady:
            load_tail(g, nslot, tag)
        mask_unit(g, qb, f"{tag}q{qb}")
        # ---- dVdK of the previous unit beside the head of this unit's VALU, the transposed reads and the next SdP's first reads ----
        g.need(("d", qb, 1, 1))                                # this unit's lse2 / D have landed
        if BAR_MID and qb == 1:
            # every wave is past its last read of the slot the next step's loads overwrite (the deferred unit's, batch 2), and the
            # next tile (loaded during batch 1) has landed: its first fragments can ride in this batch instead of behind a barrier
            if BAR_STAG:
                e(f"s_cmp_lt_u32 {s(S_WAVE)}, 2")
                e(f"s_cbranch_scc0 {g.lref('Lb4' + tag)}")
            e("s_waitcnt vmcnt(0)")
            e("s_barrier")
            if BAR_STAG:
                g.label("Lb4" + tag)
        if have_prev:
            gq = [[] for _ in range(17)]
            head = unit_valu(qb)[:VALU_SPLIT]
            place(gq, head, spread(len(head), 3, 15))
            for gp, rd in tr_plan(pqb, ps)[1]:
                gq[gp].append(rd)
            if qb == 0:
                place(gq, first_reads(1, slot), [FR_GAP + i // 2 for i in range(8)])
            elif BAR_MID:
                place(gq, first_reads(0, nslot), [FR_GAP + i // 2 for i in range(8)])
            emit_batch(g, 16, lambda i: dvdk_mfma(g, i, pqb), gq)
        else:
            for ins in unit_valu(qb)[:VALU_SPLIT]:
                e(ins)
            for ins, tg in first_reads(1, slot):
                issue(g, ins, tg)
    advance_cqt(g)
    if CARRY[0] is None:
        CARRY[0] = list(g.out)
    assert g.out == CARRY[0], (g.out, CARRY[0])


CARRY = [None]


def carry_reads():
    """reads in flight across a step boundary, oldest first (bar=mid: those of the next tile's first row fragments that the last batch
    did not retire; a slot without a next tile is read all the same).  Found by a dry run of a step body, the same for every body."""
    if not BAR_MID:
        return []
    if CARRY[0] is None:
        dry = Gen()
        dry.out = [tg for _, tg in first_reads(0, 0)] if "row" not in KO else []
        step_body(dry, 1, "dry", True, True)
    return list(CARRY[0])


def drain(g, pslot):
    """the last unit (query block 1 of the last tile): the rest of its VALU, then its dVdK"""
    for ins in unit_valu(1)[VALU_SPLIT:]:
        g.e(ins)
    pre, own = tr_plan(1, pslot)
    for _, (ins, tg) in pre:
        issue(g, ins, tg)
    gq = [[] for _ in range(17)]
    for gp, rd in own:
        gq[gp].append(rd)
    emit_batch(g, 16, lambda i: dvdk_mfma(g, i, 1), gq)
    assert not g.out


def gen_step(g, idx):
    P = f"p{idx}"
    e = g.e
    g.label("Lstep" + P)
    if not BAR_MID:
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
    e(f"s_cmp_lt_i32 {s(S_IT)}, {s(S_NF)}")
    e(f"s_cbranch_scc0 {g.lref('Lgen' + P)}")
    if ORDER_TD:
        e(f"s_cmp_ge_i32 {s(S_IT)}, {s(S_NS0)}")
    else:
        e(f"s_cmp_gt_i32 {s(S_IT)}, 0")
    e(f"s_cbranch_scc0 {g.lref('Lgen' + P)}")
    g.out = carry_reads()
    step_body(g, idx, "s" + P, True, True)
    e(f"s_branch {g.lref('Lend' + P)}")
    g.label("Lgen" + P)
    g.out = carry_reads()
    e(f"s_cmp_lt_i32 {s(S_LCNT)}, {s(S_NIT)}")
    e(f"s_cbranch_scc0 {g.lref('Lnoload' + P)}")
    load_tile(g, (idx + 1) % 3, "g" + P)
    g.label("Lnoload" + P)
    e(f"s_cmp_gt_i32 {s(S_IT)}, 0")
    e(f"s_cbranch_scc0 {g.lref('Lfirst' + P)}")
    step_body(g, idx, "g" + P, True, False)
    e(f"s_branch {g.lref('Lend' + P)}")
    g.label("Lfirst" + P)
    g.out = carry_reads()
    if idx == 0:
        step_body(g, idx, "f" + P, False, False)
    g.label("Lend" + P)
    e(f"s_add_i32 {s(S_IT)}, {s(S_IT)}, 1")
    e(f"s_cmp_lt_i32 {s(S_IT)}, {s(S_NIT)}")
    e(f"s_cbranch_scc0 {g.lref('Ldrain' + P)}")
    if idx == 2:
        e(f"s_branch {g.lref('Lstepp0')}")


def gen_body():
    g = Gen()
    e = g.e
    e(f"s_mov_b64 {sr(S_EXEC)}, exec")
    e(f"s_mov_b32 {s(S_M0SAVE)}, m0")
    for dst, name in ((S_NIT, "niter"), (S_QBEG, "qbegin"), (S_LEN, "len"), (S_SL2, "sl2"), (S_SCALE, "scale"),
                      (S_QRS2, "qrs2"), (S_FST, "fstride"), (S_KW0, "kw0"), (S_CAUSAL, "causal"), (S_WAVE, "wave"), (S_LHEAD, "head0")):
        e(f"s_mov_b32 {s(dst)}, %[{name}]")
    e(f"s_mov_b32 {s(S_SL2P)}, {s(S_SL2)}")
    e(f"s_mov_b32 {s(S_SL2P + 1)}, {s(S_SL2)}")
    e(f"s_lshl_b32 {s(S_W4)}, {s(S_WAVE)}, 2")
    e(f"s_mov_b32 {s(S_RING)}, %[ldsbase]")
    e(f"s_lshl_b32 {s(S_TMP)}, {s(S_WAVE)}, 10")
    e(f"s_add_u32 {s(S_LDSW)}, {s(S_RING)}, {s(S_TMP)}")
    e(f"s_add_i32 {s(S_LENM1)}, {s(S_LEN)}, -1")
    e(f"s_mov_b64 {sr(S_QB0)}, %[qbase]")
    e(f"s_mov_b64 {sr(S_DOB0)}, %[dobase]")
    e(f"s_mov_b64 {sr(S_LB0)}, %[lsebase]")
    e(f"s_mov_b64 {sr(S_DB0)}, %[dbase]")
    e(f"s_mov_b32 {s(S_LQT)}, {s(S_QBEG)}")
    e(f"s_mov_b32 {s(S_CQT)}, {s(S_QBEG)}")
    e(f"s_mov_b32 {s(S_LCNT)}, 0")
    e(f"s_mov_b32 {s(S_IT)}, 0")
    if ORDER_TD:
        # Direction %[dirup] = 0: cursors start on the LAST tile (the one partial tile of a ragged sequence), qt = qbegin +
        # 64 (ceil((len - qbegin) / 64) - 1), and walk down; steps [0, nheads) work on it.  %[dirup] = 1: from qbegin upward, the partial tile
        # is the last one (steps [niter - nheads, niter)).  A step is steady when its NEXT tile is whole:
        #   down:  max(1, partial ? nheads - 1 : 1) <= it < niter - 1          up:  1 <= it < (partial ? niter - nheads : niter) - 1
        e(f"s_mov_b32 {s(S_NH)}, %[nheads]")
        e(f"s_mov_b32 {s(S_H0)}, {s(S_LHEAD)}")
        e(f"s_add_i32 {s(S_HEND)}, {s(S_LHEAD)}, {s(S_NH)}")
        e(f"s_mov_b32 {s(S_CH)}, 0")
        e(f"s_sub_i32 {s(S_TMP)}, {s(S_LEN)}, {s(S_QBEG)}")
        e(f"s_add_i32 {s(S_TMP)}, {s(S_TMP)}, -1")
        e(f"s_andn2_b32 {s(S_TMP)}, {s(S_TMP)}, 63")
        e(f"s_add_i32 {s(S_TMP)}, {s(S_QBEG)}, {s(S_TMP)}")                       # the last tile
        e(f"s_cmp_lg_u32 %[dirup], 0")
        e(f"s_cselect_b32 {s(S_LQT)}, {s(S_QBEG)}, {s(S_TMP)}")
        e(f"s_cselect_b32 {s(S_STEP)}, 64, -64")
        e(f"s_mov_b32 {s(S_CQT)}, {s(S_LQT)}")
        e(f"s_sub_i32 {s(S_TMP)}, {s(S_LEN)}, {s(S_QBEG)}")
        e(f"s_and_b32 {s(S_TMP)}, {s(S_TMP)}, 63")                                # != 0: the head's last tile is partial
        e(f"s_add_i32 {s(S_TMP + 1)}, {s(S_NH)}, -1")
        e(f"s_sub_i32 {s(S_TMP + 2)}, {s(S_NIT)}, {s(S_NH)}")
        e(f"s_cmp_eq_u32 {s(S_TMP)}, 0")
        e(f"s_cselect_b32 {s(S_NS0)}, 1, {s(S_TMP + 1)}")                         # down: nheads - 1 when partial
        e(f"s_cselect_b32 {s(S_NF)}, {s(S_NIT)}, {s(S_TMP + 2)}")                 # up: niter - nheads when partial
        e(f"s_cmp_lg_u32 %[dirup], 0")
        e(f"s_cselect_b32 {s(S_NS0)}, 1, {s(S_NS0)}")
        e(f"s_cselect_b32 {s(S_NF)}, {s(S_NF)}, {s(S_NIT)}")
        e(f"s_max_i32 {s(S_NS0)}, {s(S_NS0)}, 1")
        e(f"s_add_i32 {s(S_NF)}, {s(S_NF)}, -1")
    # steady steps: it >= 1 with a next tile that is whole.  Tiles of a head are whole except its last one when (len - qbegin) % 64 != 0:
    # then no step is steady (the generic form clamps); else every step with a next tile is.
    if not ORDER_TD:
        e(f"s_sub_i32 {s(S_TMP)}, {s(S_LEN)}, {s(S_QBEG)}")
        e(f"s_and_b32 {s(S_TMP)}, {s(S_TMP)}, 63")
        e(f"s_cmp_eq_u32 {s(S_TMP)}, 0")
        e(f"s_cselect_b32 {s(S_NF)}, {s(S_NIT)}, 0")
        e(f"s_add_i32 {s(S_NF)}, {s(S_NF)}, -1")
    # lane-derived constants (as in gen_attn_fwd64.py)
    T = [V_T + i for i in range(6)]
    e(f"v_mbcnt_lo_u32_b32 {v(V_LANE)}, -1, 0")
    e(f"v_mbcnt_hi_u32_b32 {v(V_LANE)}, -1, {v(V_LANE)}")
    e(f"v_lshrrev_b32 {v(V_LANE4)}, 4, {v(V_LANE)}")
    e(f"v_lshrrev_b32 {v(V_HH8)}, 5, {v(V_LANE)}")
    e(f"v_lshlrev_b32 {v(V_HH8)}, 3, {v(V_HH8)}")
    e(f"s_and_b32 {s(S_TMP)}, {s(S_WAVE)}, 3")
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
    e(f"v_mov_b32 {v(V_NEGINF)}, 0xff800000")
    # slice offsets of a Q / dO tile
    e(f"s_lshl_b32 {s(S_TMP)}, {s(S_QRS2)}, 4")
    e(f"v_mul_lo_u32 {v(V_LO)}, {v(V_LANE4)}, {s(S_QRS2)}")
    e(f"v_add_u32 {v(V_LO)}, {v(V_LO)}, {v(V_P8)}")
    for u in range(1, 4):
        e(f"v_add_u32 {v(V_LO + u)}, {v(V_LO + u - 1)}, {s(S_TMP)}")
    # mask helpers: key of this lane relative to 8 hh; query limit relative to 8 hh; lse / D broadcast address
    e(f"v_and_b32 {v(T[0])}, 31, {v(V_LANE)}")
    e(f"v_add_u32 {v(V_KREL)}, {s(S_KW0)}, {v(T[0])}")
    e(f"v_sub_u32 {v(V_KREL)}, {v(V_KREL)}, {v(V_HH8)}")
    e(f"v_sub_u32 {v(V_QLIM)}, {s(S_LEN)}, {v(V_HH8)}")
    e(f"v_lshlrev_b32 {v(V_LA)}, 2, {v(V_HH8)}")
    e(f"v_add_u32 {v(V_LA)}, {s(S_RING)}, {v(V_LA)}")
    e(f"v_add_u32 {v(V_LA)}, {LSE_BASE}, {v(V_LA)}")
    for i in range(8):
        e(f"v_add_u32 {v(V_RA2 + i)}, {NQ * KBUF}, {v(V_RA + i)}")
        e(f"v_add_u32 {v(V_TR2 + i)}, {NQ * KBUF}, {v(V_TR + i)}")
    # ---- K / V fragments of this lane's key (clamped row; operand B of S / dP): 2 x 8 per-lane loads, once per item ----
    for st in range(8):
        e(f"global_load_dwordx4 {ar(A_KF + 4 * st, 4)}, %[kptr], off offset:{32 * st}")
        e(f"global_load_dwordx4 {ar(A_VF + 4 * st, 4)}, %[vptr], off offset:{32 * st}")
    for i in range(128):
        e(f"v_accvgpr_write_b32 {a(A_DK + i)}, 0")
    # ---- first tile ----
    load_tile(g, 0, "pro")
    if BAR_MID:
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        for ins, tg in first_reads(0, 0):
            issue(g, ins, tg)
        while g.out and g.out != carry_reads():                # (the step bodies count the in-flight reads from this state)
            g.need(g.out[0])
    # ---- tile loop: three step bodies (ring slot = it % 3) ----
    for idx in range(3):
        gen_step(g, idx)
    # ---- the last unit is still open: its fragments sit in the slot of the last tile ----
    for idx in range(3):
        g.label(f"Ldrainp{idx}")
        g.out = carry_reads()
        drain(g, idx)
        e(f"s_branch {g.lref('Lepi')}")
    g.label("Lepi")
    e("s_nop 15")
    # dK * scale, dV as bf16: lane = key row (hh halves of each 8-feature group), 8 bytes per store
    tmp = [V_U + i for i in range(4)]
    assert tmp[0] % 2 == 0
    e(f"v_cmp_ne_u32 vcc, 0, %[kvalid]")
    e("s_mov_b64 exec, vcc")
    # split form (item = one q head): the raw fp32 accumulators into this head's partial rows, straight from the accumulator file
    e("s_cmp_lg_u32 %[split], 0")
    e(f"s_cbranch_scc0 {g.lref('Lbf16')}")
    for acc, ptr in ((A_DK, "dkptr"), (A_DV, "dvptr")):
        for dt in range(4):
            for g4 in range(4):
                e(f"global_store_dwordx4 %[{ptr}], {ar(acc + 16 * dt + 4 * g4, 4)}, off offset:{128 * dt + 32 * g4}")
    e(f"s_branch {g.lref('Lstored')}")
    g.label("Lbf16")
    for acc, ptr, scaled in ((A_DK, "dkptr", True), (A_DV, "dvptr", False)):
        for dt in range(4):
            for g4 in range(4):
                base = acc + 16 * dt + 4 * g4
                for i in range(4):
                    e(f"v_accvgpr_read_b32 {v(tmp[i])}, {a(base + i)}")
                if scaled:
                    for i in range(4):
                        e(f"v_mul_f32 {v(tmp[i])}, {s(S_SCALE)}, {v(tmp[i])}")
                e(f"v_cvt_pk_bf16_f32 {v(tmp[0])}, {v(tmp[0])}, {v(tmp[1])}")
                e(f"v_cvt_pk_bf16_f32 {v(tmp[1])}, {v(tmp[2])}, {v(tmp[3])}")
                e(f"global_store_dwordx2 %[{ptr}], {vr(tmp[0], 2)}, off offset:{64 * dt + 16 * g4}")
    g.label("Lstored")
    e(f"s_mov_b64 exec, {sr(S_EXEC)}")
    e(f"s_mov_b32 m0, {s(S_M0SAVE)}")
    return g


def clobbers():
    names = [f"v{i}" for i in range(FIRST_V, LAST_V + 1)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(FIRST_S, LAST_S + 1)]
    return names + ["vcc", "scc", "memory"]


def main():
    g = gen_body()
    if "wait" in KO:                                           # timing only: the reads stay, their waits go
        g.lines = [ln for ln in g.lines if not ln.startswith("s_waitcnt lgkmcnt")]
    if "bar" in KO:                                            # timing only: no step barriers (and no load waits in front of them)
        g.lines = [ln for ln in g.lines if ln not in ("s_barrier", "s_waitcnt vmcnt(0)")]
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_attn_bwd_dkdv64.py -- do not edit; the per-item body of attn_bwd_dkdv64_kernel as one inline-asm statement.\n")
        f.write(f"// {len(g.lines)} lines; options {OPT}\n")
        f.write(f"#define VSEL_DKDV64_LDS_BYTES {LDS_BYTES}\n")
        f.write("#define VSEL_DKDV64_ASM_TEXT \\\n")
        for ln in g.lines:
            f.write(f'  "{ln}\\n\\t" \\\n')
        f.write('  ""\n')
        f.write("#define VSEL_DKDV64_ASM_CLOBBERS \\\n  ")
        f.write(", ".join(f'"{c}"' for c in clobbers()))
        f.write("\n")
    n_mfma = sum(1 for ln in g.lines if ln.startswith("v_mfma"))
    print(f"wrote {OUT}: {len(g.lines)} instructions / labels, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
