#!/usr/bin/env python3
"""Generator of the hand-scheduled per-item body of attn_fwd64_kernel (visionselector_amd/csrc/attn_fwd64.hip).

    python tools/gen_attn_fwd64.py            # rewrites visionselector_amd/csrc/attn_fwd64_body.inc

hipcc cannot be talked into the register plan this kernel needs (one 512-register wave per SIMD: O and Q in accumulator registers,
two score tiles in arch VGPRs, K / V fragments read from LDS straight into accumulator registers) nor into placing the softmax
VALU between the MFMAs: the C++ form of the same loop compiled to 690 spilled VGPRs and 1400 v_accvgpr copies.  The per-item body is
therefore ONE inline-asm statement with fixed registers, emitted by this script: an explicit register map, an instruction stream per
phase, and a placement of "filler" instructions (VALU, LDS reads) into the gaps behind each MFMA.  The C++ kernel keeps the
persistent item loop, the work queue, all address arithmetic and the log-sum-exp.

Arithmetic = varlen_attn_fwd_kernel<true, 8, 128> (attn.hip) operation for operation per query row, so outputs are bit-identical.

Register map (per lane)
  a[0:127]    O^T accumulators, block b, d-tile dt: a[64 b + 16 dt .. +15]
  a[128:191]  Q^T fragments, block b, k-step st: a[128 + 32 b + 4 st .. +3]
  a[192:223]  K fragments (8 slots of 4)
  a[224:255]  V^T fragments (2 batches x 4 d-tiles x 4)
  v[32:95]    score tile SA, v[96:159] score tile SB (block b, key block kb: +32 b + 16 kb)
  v[160:191]  P^T bf16 fragments, block b, key group g: v[160 + 16 b + 4 g .. +3]
  v[192:199]  K row-fragment LDS addresses (k-step st), v[200:207] V transposed-fragment LDS addresses (2 dt + hi)
  v208.. scalars of the online softmax and temporaries (below)
Hazards handled by construction (cdna_hip_programming.md 5.7, LLVM GCNHazardRecognizer numbers for gfx940+):
  MFMA result -> VALU / accvgpr read: >= 11 wait states (an s_nop 15 where nothing else separates them); v_exp result not used by
  the next two instructions; VALU write -> v_permlane32_swap: 2 wait states; m0 write -> LDS-DMA: 1 wait state.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("F64_OUT", os.path.join(ROOT, "visionselector_amd", "csrc", "attn_fwd64_body.inc"))

# ---- register map ---------------------------------------------------------------------------------------------------------------
A_O, A_Q, A_K, A_V = 0, 128, 192, 224
V_SA, V_SB, V_P = 32, 96, 160
V_RA, V_TR = 192, 200
V_M, V_L, V_PS, V_MX = 208, 210, 212, 214        # [2] each
V_KMAX = 216                                      # [2]
V_T = 218                                         # T0..T7 temporaries 218..225
V_LOK, V_LOV = 226, 230                           # [4] each: per-lane byte offsets of the four slices a wave loads per tile
V_P8, V_LANE4 = 234, 235                          # part8 * 2 bytes ; lane >> 4
V_LANE, V_NEGINF, V_KR = 244, 245, 246            # KR[2] = 246, 247
V_QLO = 240                                       # [4] per-lane byte offsets of a Q slice by slice phase (the old output pointers' registers)
V_HH8 = 248
V_U = 249                                         # U0..U6 more temporaries 249..255
FIRST_V, LAST_V = 32, 255

S_T, S_NT, S_NW, S_MFIRST, S_LEN, S_C, S_LDSW, S_W4 = 40, 41, 42, 43, 44, 45, 46, 47
S_KPTR, S_VPTR = 48, 50
S_KSTEP, S_VSTEP, S_KRS2, S_VRS2 = 52, 53, 54, 55
S_KB0, S_VB0 = 56, 58
S_TMP = 60                                        # 60..63
S_EXEC = 64
S_RING, S_QST, S_QRS2 = 66, 67, 68                # LDS base; this wave's Q staging area; Q row stride in bytes
S_QPTR = 80                                       # 80..81 running Q row pointer
S_QPTRB = 82                                      # 82..83 heads mode: block B's first Q row
S_MV = 70                                         # 70..73: "moves" masks of blocks A / B
S_M0SAVE, S_LENM1 = 74, 75
S_TMP2 = 76                                       # 76..77
S_NSTEADY = 78
S_TR = 84                                         # trace: 84..85 stamp, 86 previous stamp, 87.. accumulators (12)
N_ACC = 12
FIRST_S, LAST_S = 40, 99                           # (s100 / s101 are reserved: flat_scratch)

KBUF = 16384


def nk():
    return OPT["kring"]


def v_base():
    """LDS offset of V slot 0 (behind the K ring)"""
    return nk() * KBUF


def v(i):
    return f"v{i}"


def vr(i, n):
    return f"v[{i}:{i + n - 1}]"


def a(i):
    return f"a{i}"


def ar(i, n):
    return f"a[{i}:{i + n - 1}]"


def s(i):
    return f"s{i}"


def sr(i, n=2):
    return f"s[{i}:{i + n - 1}]"


class Gen:
    def __init__(self):
        self.lines = []
        self.out = []          # outstanding LDS reads, oldest first (tags)
        self.done = set()      # tags already waited for (since the last phase start)
        self.strict = True

    def e(self, text):
        self.lines.append(text)

    def label(self, name):
        self.lines.append(f"{name}%=:")

    def lref(self, name):
        return f"{name}%="

    # ---- LDS reads with counted waits ----
    def lds(self, text, tag):
        assert len(self.out) < 15, "lgkmcnt is a 4-bit counter"
        self.out.append(tag)
        self.done.discard(tag)
        self.e(text)

    def need(self, tag):
        if tag in self.out:
            idx = self.out.index(tag)
            self.e(f"s_waitcnt lgkmcnt({len(self.out) - idx - 1})")
            self.done.update(self.out[:idx + 1])
            self.out = self.out[idx + 1:]
        else:
            assert tag in self.done or not self.strict, f"fragment {tag} is consumed before its read was issued"

    def drain(self):
        if self.out:
            self.e("s_waitcnt lgkmcnt(0)")
            self.out = []


def stamp(g, k):
    """trace builds: cycles since the previous stamp are added to accumulator k (no LDS read may be outstanding: s_memtime
    returns through lgkmcnt)"""
    if not OPT["trace"]:
        return
    g.done.update(g.out)                                   # (trace builds wait for whatever is in flight here)
    g.out = []
    g.e(f"s_memtime {sr(S_TR)}")
    g.e("s_waitcnt lgkmcnt(0)")
    if k >= 0:
        g.e(f"s_sub_u32 {s(S_TMP2)}, {s(S_TR)}, {s(S_TR + 2)}")
        g.e(f"s_add_u32 {s(S_TR + 3 + k)}, {s(S_TR + 3 + k)}, {s(S_TMP2)}")
    g.e(f"s_mov_b32 {s(S_TR + 2)}, {s(S_TR)}")


def s_regs(par):
    """(score tile holding tile t, score tile receiving tile t + 1)"""
    return (V_SA, V_SB) if par == 0 else (V_SB, V_SA)


# ---- instruction streams ----------------------------------------------------------------------------------------------------------
def frag_kb_st(f):
    """fragment f of a K tile in issue order -> (key block, k-step).  srot: k-step major, so the MFMAs rotate over the four score
    accumulators (block A / B x key block 0 / 1) and a dependent accumulate sits four MFMAs behind its producer, not two"""
    return (f % 2, f // 2) if OPT["srot"] else (f // 8, f % 8)


def k_read(g, f, ks):
    """K fragment f (frag_kb_st) of K slot ks into fragment slot f % 8"""
    kb, st = frag_kb_st(f)
    g.lds(f"ds_read_b128 {ar(A_K + 4 * (f % 8), 4)}, {v(V_RA + st)} offset:{ks * KBUF + kb * 8192}", ("k", f))


def s_mfma(g, i, sn, emit=True, acc_dst=False):
    """MFMA i of S = K Q^T: fragment i // 2, block i % 2"""
    f, b = i // 2, i % 2
    kb, st = frag_kb_st(f)
    # one counted wait per group of four fragments (their reads were issued >= 4 gaps ago): every s_waitcnt is an issue slot
    g.need(("k", f | 3) if OPT["wgrp"] else ("k", f))
    dst = vr(sn + 32 * b + 16 * kb, 16)
    if acc_dst:                                  # timing experiment: the same MFMAs with accumulator-register destinations
        dst = ar(32 * b + 16 * kb, 16)
    if emit:
        g.e(f"v_mfma_f32_32x32x16_bf16 {dst}, {ar(A_K + 4 * (f % 8), 4)}, {ar(A_Q + 32 * b + 4 * st, 4)}, {'0' if st == 0 else dst}")


def v_frag(grp, dt):
    """accumulator registers of V^T fragment (key group, d-tile).  pvsplit: all sixteen stay resident through the step (key groups
    2 and 3 in the K fragment registers, which are dead during P V)"""
    if OPT["pvsplit"]:
        return (A_V, A_V + 16, A_K, A_K + 16)[grp] + 4 * dt
    return A_V + 16 * (grp & 1) + 4 * dt


def v_reads(g, grp, dt, vs):
    """the two transposed reads of V^T fragment (key group grp, d-tile dt) of V slot vs"""
    off = vs * KBUF + (32 * (grp >> 1) + 16 * (grp & 1)) * 256          # (the address registers point at V slot 0: ds offsets are 16 bits)
    base = v_frag(grp, dt)
    return [(f"ds_read_b64_tr_b16 {ar(base + 2 * hi, 2)}, {v(V_TR + 2 * dt + hi)} offset:{off}", ("v", grp, dt, hi)) for hi in range(2)]


def pv_mfma(g, i, emit=True):
    if OPT["pvsplit"]:                           # block A over all key groups, then block B from the resident fragments
        b, grp, dt = i // 16, (i % 16) // 4, i % 4
    else:
        grp, dt, b = i // 8, (i % 8) // 2, i % 2
    if OPT["wgrp"]:
        g.need(("v", grp, 3, 1))                 # the whole key group's eight reads
    else:
        g.need(("v", grp, dt, 0))
        g.need(("v", grp, dt, 1))
    acc = ar(A_O + 64 * b + 16 * dt, 16)
    if emit:
        g.e(f"v_mfma_f32_32x32x16_bf16 {acc}, {ar(v_frag(grp, dt), 4)}, {vr(V_P + 16 * b + 4 * grp, 4)}, {acc}")


def finish_chunk(sc, b, kb, first):
    """p = exp2(s c - m) in place, row sum in order, bf16 packing: 56 VALU, software-pipelined so that no result is used by
    the two instructions behind its producer"""
    base = sc + 32 * b + 16 * kb
    ops = []
    for i in range(16 + 6):
        if i < 16:
            ops.append(f"v_fma_f32 {v(base + i)}, {v(base + i)}, {s(S_C)}, -{v(V_M + b)}")
        r = i - 2
        if 0 <= r < 16:
            ops.append(f"v_exp_f32 {v(base + r)}, {v(base + r)}")
        r = i - 4
        if 0 <= r < 16:
            if first and r == 0:
                ops.append(f"v_mov_b32 {v(V_PS + b)}, {v(base + r)}")           # 0 + p = p exactly (p >= 0)
            else:
                ops.append(f"v_add_f32 {v(V_PS + b)}, {v(V_PS + b)}, {v(base + r)}")
        r = i - 5
        if 0 <= r < 16 and r % 2 == 1:
            half, w = r >> 3, (r & 7) >> 1
            ops.append(f"v_cvt_pk_bf16_f32 {v(V_P + 16 * b + 4 * (2 * kb + half) + w)}, {v(base + r - 1)}, {v(base + r)}")
    assert len(ops) == 56
    return ops


def finish_a(sc):
    return finish_chunk(sc, 0, 0, True) + finish_chunk(sc, 0, 1, False) + [f"v_add_f32 {v(V_L)}, {v(V_L)}, {v(V_PS)}"]


def max_chunk(sn, b, kb, first):
    """running maximum of the 16 scores of (block b, key block kb) into MX[b]: v_max3 chain"""
    base = sn + 32 * b + 16 * kb
    ops = []
    if first:
        ops.append(f"v_max3_f32 {v(V_MX + b)}, {v(base)}, {v(base + 1)}, {v(base + 2)}")
        rest = list(range(3, 16))
    else:
        rest = list(range(16))
    while len(rest) >= 2:
        ops.append(f"v_max3_f32 {v(V_MX + b)}, {v(V_MX + b)}, {v(base + rest[0])}, {v(base + rest[1])}")
        rest = rest[2:]
    if rest:
        ops.append(f"v_max_f32 {v(V_MX + b)}, {v(V_MX + b)}, {v(base + rest[0])}")
    return ops


def decide_head():
    """row maxima across the lane halves, candidate exponents, "moves" masks (s[S_MV..+1] block A, s[S_MV+2..+3] block B):
    straight-line VALU, placed in the last gaps of phase Y"""
    t = [V_T, V_T + 1]          # copies for the swap / candidates
    u = [V_T + 2, V_T + 3]      # m_run + tau
    ops = [f"v_mov_b32 {v(t[0])}, {v(V_MX)}", f"v_mov_b32 {v(t[1])}, {v(V_MX + 1)}", "s_nop 1",
           f"v_permlane32_swap_b32 {v(t[0])}, {v(V_MX)}", f"v_permlane32_swap_b32 {v(t[1])}, {v(V_MX + 1)}",
           f"v_add_f32 {v(u[0])}, 0x41000000, {v(V_M)}", f"v_add_f32 {v(u[1])}, 0x41000000, {v(V_M + 1)}",
           f"v_max_f32 {v(V_MX)}, {v(V_MX)}, {v(t[0])}", f"v_max_f32 {v(V_MX + 1)}, {v(V_MX + 1)}, {v(t[1])}",
           f"v_mul_f32 {v(t[0])}, {s(S_C)}, {v(V_MX)}", f"v_mul_f32 {v(t[1])}, {s(S_C)}, {v(V_MX + 1)}",
           f"v_max_f32 {v(t[0])}, {v(V_M)}, {v(t[0])}", f"v_max_f32 {v(t[1])}, {v(V_M + 1)}, {v(t[1])}",
           f"v_cmp_gt_f32 {sr(S_MV)}, {v(t[0])}, {v(u[0])}", f"v_cmp_gt_f32 {sr(S_MV + 2)}, {v(t[1])}, {v(u[1])}"]
    return ops


def decide_tail(g, name, first_tile=False):
    """the wave-uniform branch and the rare rescale: m <- candidate where it moved, alpha = exp2(m_old - m_new), l *= alpha,
    O *= alpha (skipped on an item's first tile, where l = O = 0)"""
    g.e(f"s_or_b64 {sr(S_TMP)}, {sr(S_MV)}, {sr(S_MV + 2)}")
    g.e(f"s_cbranch_scc0 {g.lref(name)}")
    al = [V_T + 4, V_T + 5]
    for b in range(2):
        g.e(f"v_cndmask_b32 {v(V_T + 2 + b)}, {v(V_M + b)}, {v(V_T + b)}, {sr(S_MV + 2 * b)}")      # m_new
    for b in range(2):
        g.e(f"v_sub_f32 {v(al[b])}, {v(V_M + b)}, {v(V_T + 2 + b)}")
    for b in range(2):
        g.e(f"v_exp_f32 {v(al[b])}, {v(al[b])}")
    for b in range(2):
        g.e(f"v_mov_b32 {v(V_M + b)}, {v(V_T + 2 + b)}")
    for b in range(2):
        g.e(f"v_mul_f32 {v(V_L + b)}, {v(V_L + b)}, {v(al[b])}")
    if not first_tile:
        g.e("s_nop 15")                               # the P V MFMAs in flight write O
        tmp = [V_U + i for i in range(6)]
        regs = [(b, i) for b in range(2) for i in range(64)]
        n = len(regs)
        for i in range(n + 4):                        # read(i) | multiply(i - 2) | write(i - 4): a temporary is free again two steps
            if i < n:                                 # before its next read
                b, r = regs[i]
                g.e(f"v_accvgpr_read_b32 {v(tmp[i % 6])}, {a(A_O + 64 * b + r)}")
            j = i - 2
            if 0 <= j < n:
                g.e(f"v_mul_f32 {v(tmp[j % 6])}, {v(tmp[j % 6])}, {v(al[regs[j][0]])}")
            k = i - 4
            if 0 <= k < n:
                b, r = regs[k]
                g.e(f"v_accvgpr_write_b32 {a(A_O + 64 * b + r)}, {v(tmp[k % 6])}")
        g.e("s_nop 4")
    g.label(name)


def rescale_selfcheck():
    """the read / multiply / write pipeline above must touch every register exactly once in each role"""
    g = Gen()
    decide_tail(g, "x")
    rd = [l for l in g.lines if l.startswith("v_accvgpr_read")]
    wr = [l for l in g.lines if l.startswith("v_accvgpr_write")]
    assert len(rd) == 128 and len(wr) == 128 and len(set(rd)) == 128 and len(set(wr)) == 128, (len(rd), len(wr))
    # a temporary is rewritten only after its write-back has been issued
    live = {}
    temps = {v(V_U + i) for i in range(6)}
    for l in g.lines:
        w = l.replace(",", " ").split()
        if w[0] == "v_accvgpr_read_b32":
            assert live.get(w[1], "free") == "free", l
            live[w[1]] = "read"
        elif w[0] == "v_mul_f32" and w[1] in temps:
            assert live[w[1]] == "read", l
            live[w[1]] = "mul"
        elif w[0] == "v_accvgpr_write_b32":
            assert live[w[2]] == "mul", l
            live[w[2]] = "free"
    assert all(x == "free" for x in live.values())


def mask_block(g, sx, tile_expr_reg):
    """s <- -inf where the key lies beyond the lane's last visible key; tile index in SGPR tile_expr_reg"""
    g.e("s_nop 15")
    g.e(f"s_lshl_b32 {s(S_TMP)}, {s(tile_expr_reg)}, 6")
    for b in range(2):
        g.e(f"v_sub_u32 {v(V_KR + b)}, {v(V_KMAX + b)}, {s(S_TMP)}")
    for b in range(2):
        g.e(f"v_sub_u32 {v(V_KR + b)}, {v(V_KR + b)}, {v(V_HH8)}")
    for b in range(2):
        for kb in range(2):
            for r in range(16):
                const = 32 * kb + 16 * (r >> 3) + (r & 7)
                reg = sx + 32 * b + 16 * kb + r
                g.e(f"v_cmp_gt_i32 vcc, {const}, {v(V_KR + b)}")
                g.e(f"v_cndmask_b32 {v(reg)}, {v(reg)}, {v(V_NEGINF)}, vcc")


def dma_ctx(tensor):
    """registers of a tile load: running pointer, 64-row step, per-lane slice offsets, row stride, row-0 pointer, key count, key count - 1,
    LDS base (+ wave * 1024)"""
    if tensor == "k":
        c = dict(ptr=S_KPTR, step=S_KSTEP, lo=V_LOK, rs2=S_KRS2, b0=S_KB0)
    else:
        c = dict(ptr=S_VPTR, step=S_VSTEP, lo=V_LOV, rs2=S_VRS2, b0=S_VB0)
    c.update(len=S_LEN, lenm1=S_LENM1, ldsw=S_LDSW, tensor=tensor)
    return c


def dma_tile(g, tensor, slot, tile_reg, uniq, ctx=None):
    """one 16 KiB tile of K or V into ring slot `slot`: four 1 KiB slices per wave.  tile_reg = SGPR holding the tile index (for the
    partial-tile test).  Advances the running pointer."""
    c = ctx or dma_ctx(tensor)
    ptr, step, lo, rs2, b0 = c["ptr"], c["step"], c["lo"], c["rs2"], c["b0"]
    lds0 = (0 if tensor == "k" else v_base()) + slot * KBUF
    g.e(f"s_lshl_b32 {s(S_TMP2)}, {s(tile_reg)}, 6")
    g.e(f"s_add_i32 {s(S_TMP2 + 1)}, {s(S_TMP2)}, 64")
    g.e(f"s_cmp_gt_i32 {s(S_TMP2 + 1)}, {s(c['len'])}")
    g.e(f"s_cbranch_scc1 {g.lref('Ltail' + uniq)}")
    for u in range(4):
        g.e(f"s_add_u32 m0, {s(c['ldsw'])}, {lds0 + u * 4096}")
        g.e("s_nop 0")
        g.e(f"global_load_lds_dwordx4 {v(lo + u)}, {sr(ptr)}")
    g.e(f"s_branch {g.lref('Ldone' + uniq)}")
    g.label("Ltail" + uniq)
    g.e(f"s_add_i32 {s(S_TMP2)}, {s(S_TMP2)}, {s(S_W4)}")                 # first row of this wave's first slice
    for u in range(4):
        g.e(f"v_add_u32 {v(V_U)}, {s(S_TMP2)}, {v(V_LANE4)}")
        g.e(f"v_min_i32 {v(V_U)}, {v(V_U)}, {s(c['lenm1'])}")
        g.e(f"v_mul_lo_u32 {v(V_U)}, {v(V_U)}, {s(rs2)}")
        g.e(f"v_add_u32 {v(V_U)}, {v(V_U)}, {v(V_P8)}")
        g.e(f"s_add_u32 m0, {s(c['ldsw'])}, {lds0 + u * 4096}")
        g.e(f"s_add_i32 {s(S_TMP2)}, {s(S_TMP2)}, 16")
        g.e(f"global_load_lds_dwordx4 {v(V_U)}, {sr(b0)}")
    g.label("Ldone" + uniq)
    g.e(f"s_add_u32 {s(ptr)}, {s(ptr)}, {s(step)}")
    g.e(f"s_addc_u32 {s(ptr + 1)}, {s(ptr + 1)}, 0")


def place(gaps, stream, where):
    """append the instructions of `stream` to the gaps listed in `where` (gap index per instruction, non-decreasing)"""
    assert len(where) == len(stream), (len(where), len(stream))
    for w, ins in zip(where, stream):
        gaps[w].append(ins)


def spread(n, lo, hi):
    """n instructions spread evenly over gaps lo .. hi (inclusive)"""
    width = hi - lo + 1
    return [lo + (i * width) // n for i in range(n)]


def emit_phase(g, n_mfma, mfma_fn, gaps):
    for i in range(n_mfma):
        mfma_fn(i)
        for ins in gaps[i]:
            if isinstance(ins, tuple):
                g.lds(ins[0], ins[1])
            else:
                g.e(ins)


# ---- schedule options (F64_OPTS="key=value,..." overrides; tools/ab_fwd64.py builds one library per option set) -------------------
OPT = {
    "move_chunk": 1,      # exponentials of (block B, key block 1) beside P V instead of beside K Q^T (balances the two phases)
    "dmak": "x:1,3,5,7",  # steady steps: where the four K(t + 2) slices are issued: "top" | "x:<gaps>" | "y:<gaps>"
    "dmav": "x:9,11,13,15",   # ... and the four V(t + 1) slices
    "pre": 12,            # VALU of the exponentials issued before the first MFMA of phase X (covers the first K reads' latency)
    "vg0": "22,29",       # phase-X gaps over which the first V^T group's reads are spread
    "qlds": 1,            # Q through LDS: 16 direct-to-LDS loads of whole rows per wave (8 lines each) and 16 ds_read_b128 instead of 16
                          # per-lane loads at a 7 KB stride (32 lines each); the staging area is the 64 KiB behind the ring
    "kring": 2,           # K ring slots: 3 = K three tiles ahead, so that K(t + 2) has been visible since the PREVIOUS step's barrier and
                          # the first K fragments of a step are read at the end of the step before (LDS latency out of phase X: measured
                          # -80 cycles in X, +110 elsewhere and one more non-steady step per item: 1146 vs 1160 TFLOP/s; not the default)
    "pvsplit": 0,         # P V: all of block A, then all of block B from V^T fragments that stay resident; block B's exponentials
                          # then run beside block A's MFMAs and only block A's beside K Q^T
    "mxx": 0,             # with pvsplit and srot = 0: the maxima of key block 0 already beside the last K Q^T MFMAs
    "early": 0,           # exponentials of the NEXT tile (block A first) issued at the end of a step, in front of the barrier wait
    "srot": 1,            # S MFMAs rotate over the four score accumulators (k-step major fragment order)
    "wgrp": 1,            # one s_waitcnt per group of four K fragments / per V key group instead of one per fragment
    "trace": 0,           # s_memtime stamps summed per wave and added to a global table at the end of every item (timing builds only)
    "ko": "",             # knock-outs for timing experiments (WRONG results): any of fin,max,dma,bar,lds,mfma joined by "+"
    "heads": 0,           # 1: the wave's two 32-row blocks are the SAME 32 queries of TWO q heads of a GQA group (attn_fwd_gqa64.hip): block B's
                          # Q rows / output rows sit %[qhs2] / %[ohs2] bytes behind block A's instead of 32 rows below, both blocks have
                          # %[nvalid] / %[nvalidb] rows (nvalidb = 0: no second head, nothing stored for it)
    "xitem": 0,           # 1 (with heads): items are pipelined into each other.  %[flags] bit 0: this item's Q fragments are already in
                          # a[128:191] and its K(0) / K(1) in the ring (prefetched by the previous body), bit 1: its V(0) too; bit 3 / 4:
                          # the same for the NEXT item -- its Q rows (%[nxq]), K(0) (%[nxk0]), K(1) (%[nxk1]; = K(0) when it has one tile)
                          # and V(0) (%[nxv]) are loaded in the gaps of this item's last tile (whole tiles and 32 whole rows only: the
                          # caller clears the bits otherwise), its Q fragments read before the epilogue; bit 2: block B has a head
                          # (replaces %[nvalidb]).  The output tile leaves through the wave's own Q staging area (no barrier).
    "qearly": 0,          # 1 (with xitem): the next item's 16 Q slices are issued at the TOP of the item's last step, in front of its softmax
                          # arithmetic (the epilogue waits for them first thing: issued in the P V gaps they had ~300 cycles of lead on a
                          # ~3 000-cycle round trip); K(0) / K(1) / V(0) stay in the P V gaps and the epilogue waits for the Q slices only.
                          # (A step earlier still -- top of the second-to-last step -- was slower: 123.0 vs 114.1 us at 32 x 524, the
                          # slices then sit in front of that step's vmcnt(0) + barrier and their issue is not under any MFMA.)
    "raw": 0,             # 1: the body also serves key-range PARTS of an item: inputs %[raw], %[prawa], %[prawb] (see the epilogue)
    "epi": 0,             # 1: epilogue with packed multiplies (v_pk_mul_f32: half the normalisation instructions, the same IEEE products) and block
                          # A's rows read back and stored while block B is still being converted (the LDS round trip and the store issue of
                          # one block under the VALU work of the other); the next item's Q fragments are waited for after the 1 / l arithmetic
}
for kv in os.environ.get("F64_OPTS", "").replace(";", ",").split(","):
    if "=" in kv:
        key, val = kv.split("=", 1)
        assert key in OPT, key
        OPT[key] = type(OPT[key])(val) if not isinstance(OPT[key], str) else val.replace("/", ",")


def dma_pieces(tensor, slot):
    """in-gap form of one full tile's four slices: [(m0 write, DMA)] + the pointer advance"""
    ptr, step, lo = (S_KPTR, S_KSTEP, V_LOK) if tensor == "k" else (S_VPTR, S_VSTEP, V_LOV)
    lds0 = (0 if tensor == "k" else v_base()) + slot * KBUF
    pieces = [(f"s_add_u32 m0, {s(S_LDSW)}, {lds0 + u * 4096}", f"global_load_lds_dwordx4 {v(lo + u)}, {sr(ptr)}") for u in range(4)]
    adv = [f"s_add_u32 {s(ptr)}, {s(ptr)}, {s(step)}", f"s_addc_u32 {s(ptr + 1)}, {s(ptr + 1)}, 0"]
    return pieces, adv


def insert_dma(gaps, where, pieces, adv):
    """m0 write first in its gap, the load behind the gap's first other instruction (>= 1 wait state after the m0 write)"""
    for w, (m0w, ld) in zip(where, pieces):
        rest = gaps[w]
        gaps[w] = [m0w] + (rest[:1] if rest else ["s_nop 0"]) + [ld] + rest[1:]
    gaps[where[-1] + 1] = adv + gaps[where[-1] + 1]


def parse_where(spec):
    if spec == "top":
        return None, None
    ph, lst = spec.split(":")
    return ph, [int(x) for x in lst.split(",")]


def full_step(g, par, tag, steady, ks, kdma, ks_next):
    """S(t + 1) beside the exponentials of tile t; mask; P V beside the maxima of tile t + 1; reference exponents.  s[S_TMP] = t + 1.
    steady: K(t + 2) / V(t + 1) are full tiles that exist -- their direct-to-LDS loads ride in the gaps."""
    sc, sn = s_regs(par)
    vs = par
    if nk() == 3:
        seed = [("k", f) for f in range(8)]                  # issued at the end of the previous step / prologue
        assert g.out == seed or (OPT["trace"] and not g.out and all(t in g.done for t in seed))
    else:
        for f in range(8):
            k_read(g, f, ks)
    gx = [[] for _ in range(33)]
    gy = [[] for _ in range(33)]
    fin = finish_a(sc)[OPT["early"]:] + finish_chunk(sc, 1, 0, True)      # (the first "early" ones ran at the end of the previous step)
    fin_b1 = finish_chunk(sc, 1, 1, False) + [f"v_add_f32 {v(V_L + 1)}, {v(V_L + 1)}, {v(V_PS + 1)}"]
    if OPT["pvsplit"]:
        fin = finish_a(sc)[OPT["early"]:]
        fin_b0 = finish_chunk(sc, 1, 0, True)
    elif not OPT["move_chunk"]:
        fin += fin_b1
    pre = fin[:OPT["pre"]]
    fin = fin[OPT["pre"]:]
    for ins in pre:                                          # the first K fragments are still on their way
        g.e(ins)
    place(gx, fin, spread(len(fin), 0, 31))
    for f in range(8, 16):                                   # second K batch: slot f % 8 is free one MFMA after its last user
        kb, st = frag_kb_st(f)
        gx[2 * (f - 8) + 2].append((f"ds_read_b128 {ar(A_K + 4 * (f % 8), 4)}, {v(V_RA + st)} offset:{ks * KBUF + kb * 8192}", ("k", f)))
    vg0 = [r for dt in range(4) for r in v_reads(g, 0, dt, vs)]
    vg1 = [r for dt in range(4) for r in v_reads(g, 1, dt, vs)]
    mx = []
    ca, cb = max_chunk(sn, 0, 0, True), max_chunk(sn, 1, 0, True)
    for x, y in zip(ca, cb):
        mx += [x, y]
    mx_kb0 = list(mx)
    ca, cb = max_chunk(sn, 0, 1, False), max_chunk(sn, 1, 1, False)
    for x, y in zip(ca, cb):
        mx += [x, y]
    if OPT["pvsplit"]:
        # X: block A's exponentials, K reads, V key groups 0 and 1 (last read in Y: lgkmcnt counts 15), [maxima of key block 0]
        place(gx, vg0, spread(8, 16, 23))
        place(gx, vg1[:7], spread(7, 24, 30))
        gy[0].append(vg1[7])
        for grp, lo in ((2, 1), (3, 5)):
            reads = [r for dt in range(4) for r in v_reads(g, grp, dt, vs)]
            place(gy, reads, [lo + k // 2 for k in range(8)])
        if OPT["mxx"]:                                       # (k-step-minor fragment order only: key block 0 is complete after MFMA 15)
            assert not OPT["srot"]
            place(gx, mx_kb0, spread(len(mx_kb0), 22, 31))
            mx = mx[len(mx_kb0):]
        place(gy, fin_b0, spread(len(fin_b0), 0, 13))        # P of (B, key groups 0 / 1) feeds MFMA 16 at the earliest
        place(gy, fin_b1, spread(len(fin_b1), 8, 22))        # ... of key groups 2 / 3 MFMA 24
        place(gy, mx, spread(len(mx), 4, 27))
    else:
        lo, hi = [int(x) for x in OPT["vg0"].split(",")]
        place(gx, vg0, spread(8, lo, hi))
        place(gy, vg1, [0, 0, 1, 1, 2, 2, 3, 3])
        for grp in (2, 3):
            for dt in range(4):
                where = 8 * (grp - 2) + 2 * dt + 2 + (2 if grp == 2 else 0)
                for r in v_reads(g, grp, dt, vs):
                    gy[where].append(r)
    if OPT["pvsplit"]:
        pass
    elif OPT["move_chunk"]:
        place(gy, fin_b1, spread(len(fin_b1), 0, 13))        # P of (B, key block 1) feeds MFMA 17 at the earliest
        place(gy, mx, spread(len(mx), 12, 27))
    else:
        place(gy, mx, spread(len(mx), 2, 25))
    dh = decide_head()
    place(gy, dh, spread(len(dh), 28, 31))
    ko = set(OPT["ko"].split("+")) if steady else set()
    if "fin" in ko:
        gx = [[i for i in gap if not (isinstance(i, str) and i.split()[0] in ("v_fma_f32", "v_exp_f32", "v_add_f32", "v_mov_b32", "v_cvt_pk_bf16_f32"))] for gap in gx]
        gy = [[i for i in gap if not (isinstance(i, str) and i.split()[0] in ("v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32"))] for gap in gy]
    if "exp" in ko:
        gx = [[(i.replace("v_exp_f32", "v_mov_b32") if isinstance(i, str) else i) for i in gap] for gap in gx]
        gy = [[(i.replace("v_exp_f32", "v_mov_b32") if isinstance(i, str) else i) for i in gap] for gap in gy]
    if "max" in ko:
        gy = [[i for i in gap if not (isinstance(i, str) and i.split()[0] in ("v_max3_f32", "v_max_f32"))] for gap in gy]
    if "lds" in ko:
        gx = [[i for i in gap if not isinstance(i, tuple)] for gap in gx]
        gy = [[i for i in gap if not isinstance(i, tuple)] for gap in gy]
        g.out = []
        g.strict = False
    if steady and "dma" not in ko:
        for tensor, slot, spec in (("k", kdma, OPT["dmak"]), ("v", 1 - par, OPT["dmav"])):
            ph, where = parse_where(spec)
            pieces, adv = dma_pieces(tensor, slot)
            if ph is None:
                for m0w, ld in pieces:
                    g.e(m0w)
                    g.e("s_nop 0")
                    g.e(ld)
                for ins in adv:
                    g.e(ins)
            else:
                insert_dma(gx if ph == "x" else gy, where, pieces, adv)
    assert not gx[32] or all(isinstance(i, str) and i.startswith("s_add") for i in gx[32])
    emit_phase(g, 32, lambda i: s_mfma(g, i, sn, "mfma" not in ko, "sacc" in ko), gx)
    for ins in gx[32]:
        g.e(ins)
    if OPT["trace"] and steady:
        saved = list(g.out)
        g.e("s_waitcnt lgkmcnt(0)")
        g.out = []
        stamp(g, 1)
        g.done.update(saved)
    # mask of tile t + 1
    g.e(f"s_cmp_ge_i32 {s(S_TMP)}, {s(S_MFIRST)}")
    g.e(f"s_cbranch_scc0 {g.lref('Lnomask' + tag)}")
    mask_block(g, sn, S_TMP)
    g.label("Lnomask" + tag)
    emit_phase(g, 32, lambda i: pv_mfma(g, i, "mfma" not in ko), gy)
    for ins in gy[32]:
        g.e(ins)
    if ko:
        g.drain()
        g.strict = True
    assert not g.out
    if steady:
        stamp(g, 2)
    decide_tail(g, "Lkeep" + tag)
    for ins in finish_a(sn)[:OPT["early"]]:                  # tile t + 1's exponentials start here: VALU is free while the wave
        g.e(ins)                                             # waits for the others at the next barrier
    if nk() == 3:                                            # K(t + 2) has been visible since this step's barrier: the next step's first
        for f in range(8):                                   # fragments are on their way while the wave waits at the next barrier
            k_read(g, f, ks_next)


def q_frag_reads(g):
    """Q^T fragments out of the wave's staging area into a[128:191]: lane (j, hh) takes chunk 2 st + hh of rows j and 32 + j"""
    e = g.e
    QR = [V_T + i for i in range(8)]
    e(f"v_and_b32 {v(V_U)}, 31, {v(V_LANE)}")                          # j
    e(f"v_and_b32 {v(V_U + 1)}, 3, {v(V_U)}")
    e(f"v_lshlrev_b32 {v(V_U + 1)}, 2, {v(V_U + 1)}")
    e(f"v_bfe_u32 {v(V_U + 2)}, {v(V_U)}, 2, 2")
    e(f"v_or_b32 {v(V_U + 1)}, {v(V_U + 1)}, {v(V_U + 2)}")            # swz(j)
    e(f"v_lshrrev_b32 {v(V_U + 2)}, 5, {v(V_LANE)}")
    e(f"v_xor_b32 {v(V_U + 1)}, {v(V_U + 1)}, {v(V_U + 2)}")           # hh ^ swz(j)
    e(f"v_lshlrev_b32 {v(V_U)}, 8, {v(V_U)}")
    e(f"v_lshl_or_b32 {v(V_U)}, {v(V_U + 1)}, 4, {v(V_U)}")
    e(f"v_add_u32 {v(QR[0])}, {s(S_QST)}, {v(V_U)}")                    # (the staging base is a multiple of 256: the XOR below commutes)
    for st in range(1, 8):
        e(f"v_xor_b32 {v(QR[st])}, {st << 5}, {v(QR[0])}")
    n = 0
    for b in range(2):
        for st in range(8):
            g.lds(f"ds_read_b128 {ar(A_Q + 32 * b + 4 * st, 4)}, {v(QR[st])} offset:{8192 * b}", ("q", n))
            n += 1
        g.need(("q", 8 * b + 1))


# xitem: registers of the next item's loads.  In an item's last step nothing of the CURRENT item is loaded any more, so its running
# pointers are free: S_QPTR / S_QPTRB (Q), S_KPTR (K(0) then K(1)), S_VPTR (V(0)); S_KB0 holds the V(0) exec mask.
def prefetch_setup(g):
    e = g.e
    e(f"s_mov_b64 {sr(S_QPTR)}, %[nxq]")
    e(f"s_add_u32 {s(S_QPTRB)}, {s(S_QPTR)}, %[qhs2]")
    e(f"s_addc_u32 {s(S_QPTRB + 1)}, {s(S_QPTR + 1)}, 0")
    e(f"s_lshl_b32 {s(S_TMP + 1)}, {s(S_QRS2)}, 2")                      # four Q rows
    e(f"s_mul_i32 {s(S_TMP2)}, {s(S_W4)}, {s(S_KRS2)}")                  # this wave's first slice starts at row 4 * wave
    e(f"s_mov_b64 {sr(S_KPTR)}, %[nxk0]")
    e(f"s_add_u32 {s(S_KPTR)}, {s(S_KPTR)}, {s(S_TMP2)}")
    e(f"s_addc_u32 {s(S_KPTR + 1)}, {s(S_KPTR + 1)}, 0")
    e(f"s_mov_b64 {sr(S_KB0)}, %[nxk1]")
    e(f"s_add_u32 {s(S_KB0)}, {s(S_KB0)}, {s(S_TMP2)}")
    e(f"s_addc_u32 {s(S_KB0 + 1)}, {s(S_KB0 + 1)}, 0")
    e(f"s_mul_i32 {s(S_TMP2)}, {s(S_W4)}, {s(S_VRS2)}")
    e(f"s_mov_b64 {sr(S_VPTR)}, %[nxv]")
    e(f"s_add_u32 {s(S_VPTR)}, {s(S_VPTR)}, {s(S_TMP2)}")
    e(f"s_addc_u32 {s(S_VPTR + 1)}, {s(S_VPTR + 1)}, 0")
    e(f"s_bitcmp1_b32 %[flags], 4")                                       # V(0) too?  exec mask of its four loads
    e(f"s_cselect_b64 {sr(S_VB0)}, {sr(S_EXEC)}, 0")


def prefetch_pieces():
    """[(instructions in front of the load, load, instructions behind it)]: 16 Q slices, K(0), K(1) (K slots 0 / 1), V(0) (V slot 0)"""
    out = []
    for i in range(16):
        pre = [f"s_add_u32 m0, {s(S_QST)}, {1024 * i}"]
        src = S_QPTR if i < 8 else S_QPTRB
        post = [f"s_add_u32 {s(src)}, {s(src)}, {s(S_TMP + 1)}", f"s_addc_u32 {s(src + 1)}, {s(src + 1)}, 0"]
        out.append((pre, f"global_load_lds_dwordx4 {v(V_QLO + (i & 3))}, {sr(src)}", post))
    for slot, ptr in ((0, S_KPTR), (1, S_KB0)):
        for u in range(4):
            out.append(([f"s_add_u32 m0, {s(S_LDSW)}, {slot * KBUF + u * 4096}"], f"global_load_lds_dwordx4 {v(V_LOK + u)}, {sr(ptr)}", []))
    for u in range(4):
        out.append(([f"s_add_u32 m0, {s(S_LDSW)}, {v_base() + u * 4096}", f"s_mov_b64 exec, {sr(S_VB0)}"],
                    f"global_load_lds_dwordx4 {v(V_LOV + u)}, {sr(S_VPTR)}", [f"s_mov_b64 exec, {sr(S_EXEC)}"]))
    return out


def gen_step(g, idx, period):
    """step t with t % period == idx"""
    par = idx & 1
    sc, sn = s_regs(par)
    vs = par
    if nk() == 3:
        ks, kdma, ks_next, ahead = (idx + 1) % 3, idx % 3, (idx + 2) % 3, 3       # K(t + 1) read, K(t + 3) loaded, K(t + 2) read early
    else:
        ks, kdma, ks_next, ahead = 1 - par, par, None, 2
    seed = [("k", f) for f in range(8)] if nk() == 3 else []
    P = f"p{idx}"
    g.label(f"Lstep{P}")
    g.e("s_waitcnt vmcnt(0)")
    if "bar" not in OPT["ko"].split("+"):
        g.e("s_barrier")
    g.e(f"s_add_i32 {s(S_TMP)}, {s(S_T)}, 1")                 # stays t + 1 through a full step (mask test)
    g.e(f"s_cmp_lt_i32 {s(S_T)}, {s(S_NSTEADY)}")
    g.e(f"s_cbranch_scc0 {g.lref('Lgen' + P)}")
    # ---------------- steady full step: the next tiles' loads ride in the gaps ----------------
    g.out = list(seed)
    stamp(g, 0)
    if OPT["trace"]:
        g.e(f"s_add_u32 {s(S_TR + 3 + 7)}, {s(S_TR + 3 + 7)}, 1")
    full_step(g, par, "s" + P, True, ks, kdma, ks_next)
    stamp(g, 3)
    g.e(f"s_branch {g.lref('Lend' + P)}")
    # ---------------- generic step: loads up front (conditional, partial tiles), then full / last / nothing ----------------
    g.out = []
    g.label("Lgen" + P)
    # K(t + ahead) -> K slot kdma, V(t + 1) -> V slot 1 - par
    g.e(f"s_add_i32 {s(S_TMP)}, {s(S_T)}, {ahead}")
    g.e(f"s_cmp_lt_i32 {s(S_TMP)}, {s(S_NT)}")
    g.e(f"s_cbranch_scc0 {g.lref('LnoK' + P)}")
    dma_tile(g, "k", kdma, S_TMP, "k" + P)
    g.label("LnoK" + P)
    g.e(f"s_add_i32 {s(S_TMP)}, {s(S_T)}, 1")
    g.e(f"s_cmp_lt_i32 {s(S_TMP)}, {s(S_NT)}")
    g.e(f"s_cbranch_scc0 {g.lref('LnoV' + P)}")
    dma_tile(g, "v", 1 - par, S_TMP, "v" + P)
    g.label("LnoV" + P)
    g.e(f"s_cmp_ge_i32 {s(S_T)}, {s(S_NW)}")
    g.e(f"s_cbranch_scc1 {g.lref('Lidle' + P)}")
    g.e(f"s_add_i32 {s(S_TMP)}, {s(S_T)}, 1")
    g.e(f"s_cmp_lt_i32 {s(S_TMP)}, {s(S_NW)}")
    g.e(f"s_cbranch_scc0 {g.lref('Llast' + P)}")
    g.out = list(seed)
    full_step(g, par, "g" + P, False, ks, kdma, ks_next)
    stamp(g, 4)
    g.e(f"s_branch {g.lref('Lend' + P)}")

    # ---------------- the wave's last tile: nothing to overlap with ----------------
    def last_tile(tagx, pieces):
        g.out = list(seed)                                   # (K fragments read ahead for a step that does not come: never used)
        if pieces and OPT["qearly"]:
            for pre, ld, post in pieces[:16]:
                for ins in pre + ["s_nop 0", ld] + post:
                    g.e(ins)
            pieces = pieces[16:]
        for ins in finish_a(sc)[OPT["early"]:] + finish_chunk(sc, 1, 0, True) + finish_chunk(sc, 1, 1, False) + \
                [f"v_add_f32 {v(V_L + 1)}, {v(V_L + 1)}, {v(V_PS + 1)}"]:
            g.e(ins)
        gl = [[] for _ in range(33)]
        for text, tag in [r for dt in range(4) for r in v_reads(g, 0, dt, vs)]:
            if len(g.out) >= 14:
                g.need(g.out[0])
            g.lds(text, tag)
        place(gl, [r for dt in range(4) for r in v_reads(g, 1, dt, vs)], [0, 0, 1, 1, 2, 2, 3, 3])
        for grp in (2, 3):
            for dt in range(4):
                where = 8 * (grp - 2) + 2 * dt + 2 + (2 if grp == 2 else 0)
                if OPT["pvsplit"]:
                    where = 4 * (grp - 1) + dt                   # key group g feeds MFMA 4 g of the block-A pass
                for r in v_reads(g, grp, dt, vs):
                    gl[where].append(r)
        # the next item's loads: one per gap behind the V reads of the gap (an m0 write needs one wait state in front of its load)
        for n_, (pre, ld, post) in enumerate(pieces):
            w = 2 + n_ if n_ < 28 else 31
            gl[w] = gl[w] + pre + ["s_nop 0", ld] + post
        emit_phase(g, 32, lambda i: pv_mfma(g, i), gl[:32])
        assert not g.out and not gl[32]
        stamp(g, 4)

    g.label("Llast" + P)
    if OPT["xitem"]:
        # the ITEM's last step (t + 1 == ntiles) with a next item to prefetch for?  (a wave whose last tile comes earlier -- other query
        # slices of a wider tile -- issues the loads in its idle last step below)
        g.e(f"s_add_i32 {s(S_TMP)}, {s(S_T)}, 1")
        g.e(f"s_cmp_eq_u32 {s(S_TMP)}, {s(S_NT)}")
        g.e(f"s_cbranch_scc0 {g.lref('Llastplain' + P)}")
        g.e(f"s_bitcmp1_b32 %[flags], 3")
        g.e(f"s_cbranch_scc0 {g.lref('Llastplain' + P)}")
        prefetch_setup(g)
        last_tile("x", prefetch_pieces())
        g.e(f"s_branch {g.lref('Lend' + P)}")
        g.label("Llastplain" + P)
    last_tile("", [])

    g.e(f"s_branch {g.lref('Lend' + P)}")
    g.label("Lidle" + P)
    if OPT["xitem"]:
        g.e(f"s_add_i32 {s(S_TMP)}, {s(S_T)}, 1")
        g.e(f"s_cmp_eq_u32 {s(S_TMP)}, {s(S_NT)}")
        g.e(f"s_cbranch_scc0 {g.lref('Lidlenp' + P)}")
        g.e(f"s_bitcmp1_b32 %[flags], 3")
        g.e(f"s_cbranch_scc0 {g.lref('Lidlenp' + P)}")
        prefetch_setup(g)
        for pre, ld, post in prefetch_pieces():
            for ins in pre + ["s_nop 0", ld] + post:
                g.e(ins)
        g.label("Lidlenp" + P)
    stamp(g, 4)
    g.label("Lend" + P)
    g.e(f"s_add_i32 {s(S_T)}, {s(S_T)}, 1")
    g.e(f"s_cmp_lt_i32 {s(S_T)}, {s(S_NT)}")
    g.e(f"s_cbranch_scc0 {g.lref('Lepi')}")
    if idx == period - 1:
        g.e(f"s_branch {g.lref('Lstepp0')}")


def epilogue_pipelined(g, xitem):
    """option epi=1 (see OPT): the epilogue of the item, same values and same stores as the plain form"""
    e = g.e
    e("s_nop 15")
    T0, T1, X, D0, R, N, E1, Q_ = [V_T + i for i in range(8)]
    if xitem:
        # the next item's Q rows have landed in this wave's staging area (its own loads): fragments into a[128:191]; waited for below
        e(f"s_bitcmp1_b32 %[flags], 3")
        e(f"s_cbranch_scc0 {g.lref('Lnoqn')}")
        e(f"s_waitcnt vmcnt({8 if OPT['qearly'] else 0})")           # (qearly: the 16 Q slices are the oldest loads; 8 K + 4 masked V behind them)
        q_frag_reads(g)
        g.out = []                                                   # (drained by the s_waitcnt lgkmcnt(0) in front of the first ds_write)
        g.label("Lnoqn")
    else:
        e("s_barrier")
    # 1 / l of both blocks, each in the LOW register of an even-aligned pair (the packed multiply takes it for both halves)
    INVP = [V_U + 1, V_T + 4]
    assert INVP[0] % 2 == 0 and INVP[1] % 2 == 0
    SAVE = [V_U + 2, V_U]                                            # (the T registers are the division's temporaries: results parked here)
    for b in range(2):
        e(f"v_mov_b32 {v(T0)}, {v(V_L + b)}")
        e(f"v_mov_b32 {v(T1)}, {v(V_L + b)}")
        e("s_nop 1")
        e(f"v_permlane32_swap_b32 {v(T0)}, {v(T1)}")
        e(f"v_add_f32 {v(X)}, {v(T0)}, {v(T1)}")
        e(f"v_div_scale_f32 {v(D0)}, {sr(S_TMP)}, {v(X)}, {v(X)}, 1.0")
        e(f"v_rcp_f32 {v(R)}, {v(D0)}")
        e(f"v_div_scale_f32 {v(N)}, vcc, 1.0, {v(X)}, 1.0")
        e("s_nop 0")
        e(f"v_fma_f32 {v(E1)}, -{v(D0)}, {v(R)}, 1.0")
        e(f"v_fmac_f32 {v(R)}, {v(E1)}, {v(R)}")
        e(f"v_mul_f32 {v(Q_)}, {v(N)}, {v(R)}")
        e(f"v_fma_f32 {v(E1)}, -{v(D0)}, {v(Q_)}, {v(N)}")
        e(f"v_fmac_f32 {v(Q_)}, {v(E1)}, {v(R)}")
        e(f"v_fma_f32 {v(D0)}, -{v(D0)}, {v(Q_)}, {v(N)}")
        e(f"v_div_fmas_f32 {v(D0)}, {v(D0)}, {v(R)}, {v(Q_)}")
        e(f"v_div_fixup_f32 {v(SAVE[b])}, {v(D0)}, {v(X)}, 1.0")
        e(f"v_cmp_lt_f32 vcc, 0, {v(X)}")
        e(f"v_cndmask_b32 {v(SAVE[b])}, 0, {v(SAVE[b])}, vcc")
    e(f"v_mov_b32 {v(INVP[0])}, {v(SAVE[0])}")
    e(f"v_mov_b32 {v(INVP[1])}, {v(SAVE[1])}")
    WA, RD, AD = V_T, V_T + 1, V_T + 2
    if xitem:
        e(f"s_mov_b32 {s(S_TMP)}, {s(S_QST)}")
    else:
        e(f"s_lshl_b32 {s(S_TMP)}, %[wave], 14")
        e(f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, {s(S_RING)}")
    e(f"v_and_b32 {v(WA)}, 31, {v(V_LANE)}")
    e(f"v_lshlrev_b32 {v(WA)}, 8, {v(WA)}")
    e(f"v_add_u32 {v(WA)}, {v(WA)}, {v(V_HH8)}")
    e(f"v_and_b32 {v(AD)}, 15, {v(V_LANE)}")
    e(f"v_lshlrev_b32 {v(AD)}, 4, {v(AD)}")
    e(f"v_xor_b32 {v(WA)}, {v(WA)}, {v(AD)}")
    e(f"v_add_u32 {v(WA)}, {s(S_TMP)}, {v(WA)}")
    e(f"v_lshlrev_b32 {v(RD)}, 4, {v(V_LANE)}")
    e(f"v_add_u32 {v(RD)}, {s(S_TMP)}, {v(RD)}")
    # global offsets of the four row phases: (lane >> 4) * row stride + 16 * ((lane & 15) ^ (4 k + (lane >> 4)))
    VO = [V_KR, V_KR + 1, V_MX, V_MX + 1]
    WB = V_T + 3
    e(f"v_and_b32 {v(AD)}, 15, {v(V_LANE)}")
    e(f"v_mul_lo_u32 {v(WB)}, {v(V_LANE4)}, %[ostride]")
    for k in range(4):
        e(f"v_add_u32 {v(VO[k])}, {4 * k}, {v(V_LANE4)}")
        e(f"v_xor_b32 {v(VO[k])}, {v(VO[k])}, {v(AD)}")
        e(f"v_lshl_add_u32 {v(VO[k])}, {v(VO[k])}, 4, {v(WB)}")
    e(f"s_mov_b64 {sr(S_TMP)}, %[obase]")
    e(f"s_lshl_b32 {s(S_TMP2)}, %[ostride], 2")
    e(f"s_mov_b32 {s(S_TMP2 + 1)}, %[nvalid]")
    tmp = [V_U + 3 + i for i in range(4)]
    assert tmp[0] % 2 == 0 and tmp[3] <= LAST_V
    e("s_waitcnt lgkmcnt(0)")                                        # (the next item's Q fragments: the staging area is free now)

    def convert(b, dt, g4):
        base = A_O + 64 * b + 16 * dt + 4 * g4
        for i in range(4):
            e(f"v_accvgpr_read_b32 {v(tmp[i])}, {a(base + i)}")
        e(f"v_pk_mul_f32 {vr(tmp[0], 2)}, {vr(tmp[0], 2)}, {vr(INVP[b], 2)} op_sel_hi:[1,0]")
        e(f"v_pk_mul_f32 {vr(tmp[2], 2)}, {vr(tmp[2], 2)}, {vr(INVP[b], 2)} op_sel_hi:[1,0]")
        e(f"v_cvt_pk_bf16_f32 {v(tmp[0])}, {v(tmp[0])}, {v(tmp[1])}")
        e(f"v_cvt_pk_bf16_f32 {v(tmp[1])}, {v(tmp[2])}, {v(tmp[3])}")
        e(f"v_xor_b32 {v(AD)}, {(4 * dt + g4) << 4}, {v(WA)}")
        e(f"ds_write_b64 {v(AD)}, {vr(tmp[0], 2)} offset:{8192 * b}")

    def store(i):
        if OPT["heads"] and i == 8:                                 # block B: the second head's rows (none when there is no second head)
            e(f"s_mov_b64 {sr(S_TMP)}, %[obase]")
            e(f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, %[ohs2]")
            e(f"s_addc_u32 {s(S_TMP + 1)}, {s(S_TMP + 1)}, 0")
            if OPT["xitem"]:
                e(f"s_bitcmp1_b32 %[flags], 2")
                e(f"s_cselect_b32 {s(S_TMP2 + 1)}, %[nvalid], 0")
            else:
                e(f"s_mov_b32 {s(S_TMP2 + 1)}, %[nvalidb]")
        e(f"v_cmp_gt_i32 vcc, {s(S_TMP2 + 1)}, {v(V_LANE4)}")        # row 4 i + (lane >> 4) of this wave exists
        e("s_mov_b64 exec, vcc")
        e(f"global_store_dwordx4 {v(VO[i % 4])}, {vr(V_SA + 4 * i, 4)}, {sr(S_TMP)}")
        e(f"s_mov_b64 exec, {sr(S_EXEC)}")
        e(f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, {s(S_TMP2)}")
        e(f"s_addc_u32 {s(S_TMP + 1)}, {s(S_TMP + 1)}, 0")
        e(f"s_add_i32 {s(S_TMP2 + 1)}, {s(S_TMP2 + 1)}, -4")

    # block A converted and written; its eight row groups read back; block B converted with block A's stores between the groups
    for dt in range(4):
        for g4 in range(4):
            convert(0, dt, g4)
    e("s_waitcnt lgkmcnt(0)")
    for i in range(8):
        e(f"ds_read_b128 {vr(V_SA + 4 * i, 4)}, {v(RD)} offset:{1024 * i}")
    n = 0
    for dt in range(4):
        for g4 in range(4):
            convert(1, dt, g4)                                       # (one more LDS operation behind the reads: they return in order)
            n += 1
            if n % 2 == 0:
                i = n // 2 - 1                                       # reads 0 .. i have landed when at most (7 - i) reads + n writes are outstanding
                e(f"s_waitcnt lgkmcnt({min(13, 7 - i + n)})")     # (capped: never more than 15 LDS operations outstanding)
                store(i)
    e("s_waitcnt lgkmcnt(0)")
    for i in range(8, 16):
        e(f"ds_read_b128 {vr(V_SA + 4 * i, 4)}, {v(RD)} offset:{1024 * i}")
    for i in range(8, 16):
        e(f"s_waitcnt lgkmcnt({15 - i})")
        store(i)


def gen_body():
    rescale_selfcheck()
    assert nk() == 2, "kring = 3 (measured slower) predates the Q staging area at +64 KiB: its five tiles would overlap it"
    g = Gen()
    e = g.e
    # ---- inputs into the fixed registers ----
    e(f"s_mov_b64 {sr(S_EXEC)}, exec")
    e(f"s_mov_b32 {s(S_M0SAVE)}, m0")
    if OPT["trace"]:
        for k in range(N_ACC):
            e(f"s_mov_b32 {s(S_TR + 3 + k)}, 0")
        stamp(g, -1)
    for dst, name in ((S_NT, "ntiles"), (S_NW, "nw"), (S_MFIRST, "mfirst"), (S_LEN, "len"), (S_C, "c"), (S_KRS2, "krs2"), (S_VRS2, "vrs2")):
        e(f"s_mov_b32 {s(dst)}, %[{name}]")
    e(f"s_lshl_b32 {s(S_W4)}, %[wave], 2")
    e(f"s_mov_b32 {s(S_RING)}, %[ldsbase]")
    e(f"s_lshl_b32 {s(S_TMP)}, %[wave], 10")
    e(f"s_add_u32 {s(S_LDSW)}, {s(S_RING)}, {s(S_TMP)}")
    e(f"s_lshl_b32 {s(S_TMP)}, %[wave], 14")                 # Q staging: 64 KiB behind the ring, 16 KiB per wave
    e(f"s_add_u32 {s(S_QST)}, {s(S_RING)}, {s(S_TMP)}")
    e(f"s_add_u32 {s(S_QST)}, {s(S_QST)}, 0x10000")
    e(f"s_mov_b32 {s(S_QRS2)}, %[qrs2]")
    e(f"s_add_i32 {s(S_LENM1)}, {s(S_LEN)}, -1")
    # steps 0 .. NSTEADY - 1 are full steps whose next tiles K(t + ring), V(t + 1) exist and are whole: min(nw - 1, ntiles - ring, len / 64 - ring)
    e(f"s_add_i32 {s(S_NSTEADY)}, {s(S_NW)}, -1")
    e(f"s_add_i32 {s(S_TMP)}, {s(S_NT)}, {-nk()}")
    e(f"s_min_i32 {s(S_NSTEADY)}, {s(S_NSTEADY)}, {s(S_TMP)}")
    e(f"s_ashr_i32 {s(S_TMP)}, {s(S_LEN)}, 6")
    e(f"s_add_i32 {s(S_TMP)}, {s(S_TMP)}, {-nk()}")
    e(f"s_min_i32 {s(S_NSTEADY)}, {s(S_NSTEADY)}, {s(S_TMP)}")
    e(f"s_mov_b64 {sr(S_KB0)}, %[kbase]")
    e(f"s_mov_b64 {sr(S_VB0)}, %[vbase]")
    # running pointers start at row 4 * wave of tile 0
    e(f"s_mul_i32 {s(S_TMP)}, {s(S_W4)}, {s(S_KRS2)}")
    e(f"s_add_u32 {s(S_KPTR)}, {s(S_KB0)}, {s(S_TMP)}")
    e(f"s_addc_u32 {s(S_KPTR + 1)}, {s(S_KB0 + 1)}, 0")
    e(f"s_mul_i32 {s(S_TMP)}, {s(S_W4)}, {s(S_VRS2)}")
    e(f"s_add_u32 {s(S_VPTR)}, {s(S_VB0)}, {s(S_TMP)}")
    e(f"s_addc_u32 {s(S_VPTR + 1)}, {s(S_VB0 + 1)}, 0")
    e(f"s_lshl_b32 {s(S_KSTEP)}, {s(S_KRS2)}, 6")
    e(f"s_lshl_b32 {s(S_VSTEP)}, {s(S_VRS2)}, 6")
    # lane-derived constants (attn_common.h: slice_src_part, make_row_addr, make_tr_addr; emulated against them in the generator's
    # self-check): lane, lane >> 4, 8 * (lane >> 5), the byte offset of this lane's source part in a slice, the first K row-fragment
    # and V^T fragment offsets
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

    def swz_of(row, dst, tmp):                              # ((row & 3) << 2) | ((row >> 2) & 3)
        e(f"v_and_b32 {v(dst)}, 3, {v(row)}")
        e(f"v_lshlrev_b32 {v(dst)}, 2, {v(dst)}")
        e(f"v_bfe_u32 {v(tmp)}, {v(row)}, 2, 2")
        e(f"v_or_b32 {v(dst)}, {v(dst)}, {v(tmp)}")
    # row_addr[0]: row = perm_row(lane & 31), part = (lane >> 5) ^ swz(row)
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
    # tr_addr[0][0]: p16 = lane & 15, row = 8 hh + (p16 >> 2), part = (2 ((lane >> 4) & 1) + ((p16 & 3) >> 1)) ^ swz(row), + 8 (p16 & 1)
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
    e(f"v_mov_b32 {v(V_KMAX)}, %[kmaxa]")
    e(f"v_mov_b32 {v(V_KMAX + 1)}, %[kmaxb]")
    e(f"v_mov_b32 {v(V_NEGINF)}, 0xff800000")
    # slice offsets: lane_off + u * 16 rows
    e(f"s_lshl_b32 {s(S_TMP)}, {s(S_KRS2)}, 4")
    e(f"s_lshl_b32 {s(S_TMP + 1)}, {s(S_VRS2)}, 4")
    e(f"v_mul_lo_u32 {v(V_LOK)}, {v(V_LANE4)}, {s(S_KRS2)}")
    e(f"v_mul_lo_u32 {v(V_LOV)}, {v(V_LANE4)}, {s(S_VRS2)}")
    e(f"v_add_u32 {v(V_LOK)}, {v(V_LOK)}, {v(V_P8)}")
    e(f"v_add_u32 {v(V_LOV)}, {v(V_LOV)}, {v(V_P8)}")
    for u in range(1, 4):
        e(f"v_add_u32 {v(V_LOK + u)}, {v(V_LOK + u - 1)}, {s(S_TMP)}")
        e(f"v_add_u32 {v(V_LOV + u)}, {v(V_LOV + u - 1)}, {s(S_TMP + 1)}")
    # LDS fragment addresses: row_addr[st] = row_addr[0] ^ (st << 5); tr_addr[dt][0] = tr_addr[0][0] ^ (dt << 6),
    # tr_addr[dt][1] = (tr_addr[dt][0] ^ 16) + 1024   (attn_common.h make_row_addr / make_tr_addr; checked by the C++ side)
    for st in range(1, 8):
        e(f"v_xor_b32 {v(V_RA + st)}, {st << 5}, {v(V_RA)}")
    for dt in range(1, 4):
        e(f"v_xor_b32 {v(V_TR + 2 * dt)}, {dt << 6}, {v(V_TR)}")
    for dt in range(4):
        e(f"v_xor_b32 {v(V_TR + 2 * dt + 1)}, 16, {v(V_TR + 2 * dt)}")
        e(f"v_add_u32 {v(V_TR + 2 * dt + 1)}, 0x400, {v(V_TR + 2 * dt + 1)}")
    for i in range(8):                                # (the relations hold for offsets inside a tile: the LDS base comes last)
        e(f"v_add_u32 {v(V_RA + i)}, {s(S_RING)}, {v(V_RA + i)}")
        e(f"v_add_u32 {v(V_TR + i)}, {s(S_RING)}, {v(V_TR + i)}")
    for i in range(8):
        e(f"v_add_u32 {v(V_TR + i)}, {v_base()}, {v(V_TR + i)}")
    # ---- Q fragments straight into accumulator registers; O = 0; softmax state ----
    e(f"s_mov_b32 {s(S_T)}, 0")
    # Q: 16 slices of 4 whole rows per wave, direct to LDS (row r, 16-byte chunk c at c ^ swz(r), as the K tiles); a wave whose rows
    # do not all exist clamps the source row per lane (the replayed rows are never stored)
    e(f"v_mul_lo_u32 {v(V_T)}, {v(V_LANE4)}, {s(S_QRS2)}")
    e(f"v_and_b32 {v(V_T + 1)}, 15, {v(V_LANE)}")
    e(f"v_lshlrev_b32 {v(V_T + 2)}, 2, {v(V_LANE4)}")
    for kq in range(4):
        e(f"v_or_b32 {v(V_QLO + kq)}, {kq}, {v(V_T + 2)}")
        e(f"v_xor_b32 {v(V_QLO + kq)}, {v(V_QLO + kq)}, {v(V_T + 1)}")
        e(f"v_lshl_add_u32 {v(V_QLO + kq)}, {v(V_QLO + kq)}, 4, {v(V_T)}")
    xitem = OPT["xitem"]
    assert not xitem or OPT["heads"], "xitem is built on the heads form"
    if xitem:
        e(f"s_bitcmp1_b32 %[flags], 0")                                   # Q fragments, K(0), K(1) came with the previous item's last tile
        e(f"s_cbranch_scc1 {g.lref('Lpkdone')}")
    e(f"s_mov_b64 {sr(S_QPTR)}, %[qbase]")
    e(f"s_lshl_b32 {s(S_TMP + 1)}, {s(S_QRS2)}, 2")
    heads = OPT["heads"]
    if heads:
        e(f"s_add_u32 {s(S_QPTRB)}, {s(S_QPTR)}, %[qhs2]")
        e(f"s_addc_u32 {s(S_QPTRB + 1)}, {s(S_QPTR + 1)}, 0")
    e(f"s_cmp_gt_i32 %[nvalid], {31 if heads else 63}")
    e(f"s_cbranch_scc0 {g.lref('Lqpart')}")
    for i in range(16):
        if heads and i == 8:                                   # block B: the same rows of the second head
            e(f"s_mov_b64 {sr(S_QPTR)}, {sr(S_QPTRB)}")
        e(f"s_add_u32 m0, {s(S_QST)}, {1024 * i}")
        e("s_nop 0")
        e(f"global_load_lds_dwordx4 {v(V_QLO + (i & 3))}, {sr(S_QPTR)}")
        e(f"s_add_u32 {s(S_QPTR)}, {s(S_QPTR)}, {s(S_TMP + 1)}")
        e(f"s_addc_u32 {s(S_QPTR + 1)}, {s(S_QPTR + 1)}, 0")
    e(f"s_branch {g.lref('Lqdone')}")
    g.label("Lqpart")
    e(f"s_cmp_gt_i32 %[nvalid], 0")
    e(f"s_cbranch_scc0 {g.lref('Lqdone')}")
    e(f"s_add_i32 {s(S_TMP2)}, %[nvalid], -1")
    for i in range(16):
        e(f"v_add_u32 {v(V_U)}, {4 * (i & 7 if heads else i)}, {v(V_LANE4)}")
        e(f"v_min_i32 {v(V_U)}, {v(V_U)}, {s(S_TMP2)}")
        e(f"v_mul_lo_u32 {v(V_U)}, {v(V_U)}, {s(S_QRS2)}")
        e(f"v_sub_u32 {v(V_U + 1)}, {v(V_QLO + (i & 3))}, {v(V_T)}")          # the chunk term of this slice phase
        e(f"v_add_u32 {v(V_U)}, {v(V_U)}, {v(V_U + 1)}")
        e(f"s_add_u32 m0, {s(S_QST)}, {1024 * i}")
        e("s_nop 0")
        e(f"global_load_lds_dwordx4 {v(V_U)}, {sr(S_QPTRB if heads and i >= 8 else S_QPTR)}")
    g.label("Lqdone")
    dma_tile(g, "k", 0, S_T, "pk0")
    if xitem:                                                 # (K(1) in front of V(0): the skip labels below)
        e(f"s_cmp_gt_i32 {s(S_NT)}, 1")
        e(f"s_cbranch_scc0 {g.lref('Lp1')}")
        e(f"s_mov_b32 {s(S_TMP)}, 1")
        dma_tile(g, "k", 1, S_TMP, "pk1")
        g.label("Lp1")
        e(f"s_branch {g.lref('Lpv')}")
        g.label("Lpkdone")
        # prefetched: the running K pointer the steps continue from = row 4 * wave of tile 2 (the prologue's own loads advance it twice)
        e(f"s_lshl_b32 {s(S_TMP)}, {s(S_KSTEP)}, 1")
        e(f"s_add_u32 {s(S_KPTR)}, {s(S_KPTR)}, {s(S_TMP)}")
        e(f"s_addc_u32 {s(S_KPTR + 1)}, {s(S_KPTR + 1)}, 0")
        g.label("Lpv")
        e(f"s_bitcmp1_b32 %[flags], 1")
        e(f"s_cbranch_scc1 {g.lref('Lpvskip')}")
        dma_tile(g, "v", 0, S_T, "pv0")
        e(f"s_branch {g.lref('Lpvdone')}")
        g.label("Lpvskip")
        e(f"s_add_u32 {s(S_VPTR)}, {s(S_VPTR)}, {s(S_VSTEP)}")
        e(f"s_addc_u32 {s(S_VPTR + 1)}, {s(S_VPTR + 1)}, 0")
        g.label("Lpvdone")
    else:
        dma_tile(g, "v", 0, S_T, "pv0")
        e(f"s_cmp_gt_i32 {s(S_NT)}, 1")
        e(f"s_cbranch_scc0 {g.lref('Lp1')}")
        e(f"s_mov_b32 {s(S_TMP)}, 1")
        dma_tile(g, "k", 1, S_TMP, "pk1")
        g.label("Lp1")
    for i in range(128):
        e(f"v_accvgpr_write_b32 {a(A_O + i)}, 0")
    for b in range(2):
        e(f"v_mov_b32 {v(V_M + b)}, 0xf149f2ca")       # -1e30f
        e(f"v_mov_b32 {v(V_L + b)}, 0")
    stamp(g, 8)
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    stamp(g, 9)
    e(f"s_cmp_gt_i32 {s(S_NW)}, 0")
    e(f"s_cbranch_scc0 {g.lref('Lstepp0')}")
    # ---- Q^T fragments out of the staging area (unless the previous item's body left them in a[128:191]) ----
    if xitem:
        e(f"s_bitcmp1_b32 %[flags], 0")
        e(f"s_cbranch_scc1 {g.lref('Lqfdone')}")
    q_frag_reads(g)
    if xitem:
        g.drain()
        g.label("Lqfdone")
    # ---- S(0) without overlap, mask, maxima, first reference exponents ----
    for f in range(8):
        k_read(g, f, 0)
    g0 = [[] for _ in range(32)]
    for f in range(8, 16):
        kb, st = frag_kb_st(f)
        g0[2 * (f - 8) + 2].append((f"ds_read_b128 {ar(A_K + 4 * (f % 8), 4)}, {v(V_RA + st)} offset:{kb * 8192}", ("k", f)))
    emit_phase(g, 32, lambda i: s_mfma(g, i, V_SA), g0)
    stamp(g, 10)
    e(f"s_cmp_ge_i32 {s(S_T)}, {s(S_MFIRST)}")
    e(f"s_cbranch_scc0 {g.lref('Lnomask0')}")
    mask_block(g, V_SA, S_T)
    g.label("Lnomask0")
    e("s_nop 15")
    ca, cb = max_chunk(V_SA, 0, 0, True), max_chunk(V_SA, 1, 0, True)
    for x, y in zip(ca, cb):
        e(x)
        e(y)
    ca, cb = max_chunk(V_SA, 0, 1, False), max_chunk(V_SA, 1, 1, False)
    for x, y in zip(ca, cb):
        e(x)
        e(y)
    for ins in decide_head():
        e(ins)
    decide_tail(g, "Lkeep0", first_tile=True)
    for ins in finish_a(V_SA)[:OPT["early"]]:
        e(ins)
    if nk() == 3:
        for f in range(8):                                   # step 0's first K(1) fragments
            k_read(g, f, 1)
        g.out = []
    # ---- tile loop ----
    stamp(g, 5)
    period = 6 if nk() == 3 else 2
    for idx in range(period):
        gen_step(g, idx, period)
    # ---- epilogue: O^T / l as bf16.  A lane holds 4 consecutive features of ONE row per register quad, so direct stores are 8 bytes
    # at a 7 KB stride (32 of them per lane: 11-16k cycles of store issue per item, measured).  The tile goes through LDS instead (the
    # K / V rings are free once every wave has passed the barrier below): ds_write_b64 with the 16-byte chunk c of row r at position
    # c ^ (r & 15) (conflict-free for the 16 lanes of a group), read back as whole rows (ds_read_b128, 4 rows per instruction) and
    # stored as 16 x 1 KiB of whole 256-byte rows.
    g.label("Lepi")
    if OPT["raw"]:
        # key-range PARTS of an item (attn_fwd64_parts.hip): %[raw] != 0 -> the un-normalised fp32 accumulators leave as they are, lane (j, hh)
        # writing features 32 dt + 8 g4 + 4 hh .. + 3 of its row at %[prawa] / %[prawb] (per-lane row pointers, + 16 hh bytes), rows that
        # exist only; (m, l) go back through the outputs as always and the merge kernel does the rest.  No barrier: nothing shared is touched.
        assert not xitem and not OPT["heads"]
        e("s_cmp_lg_u32 %[raw], 0")
        e(f"s_cbranch_scc0 {g.lref('Lnotraw')}")
        e("s_nop 15")
        for b, ptr in ((0, "prawa"), (1, "prawb")):
            e(f"v_and_b32 {v(V_T)}, 31, {v(V_LANE)}")
            if b:
                e(f"v_add_u32 {v(V_T)}, 32, {v(V_T)}")
            e(f"v_cmp_gt_i32 vcc, %[nvalid], {v(V_T)}")
            e("s_mov_b64 exec, vcc")
            for dt in range(4):
                for g4 in range(4):
                    e(f"global_store_dwordx4 %[{ptr}], {ar(A_O + 64 * b + 16 * dt + 4 * g4, 4)}, off offset:{128 * dt + 32 * g4}")
            e(f"s_mov_b64 exec, {sr(S_EXEC)}")
        e(f"s_branch {g.lref('Lrawdone')}")
        g.label("Lnotraw")
    if OPT["epi"]:
        epilogue_pipelined(g, xitem)
    else:
        e("s_nop 15")
        if xitem:
            # the next item's Q rows have landed in this wave's staging area (its own loads): fragments into a[128:191] -- the current item's are
            # dead since its last S -- and the area is free for the output tile.  Wave-private: no barrier.
            e(f"s_bitcmp1_b32 %[flags], 3")
            e(f"s_cbranch_scc0 {g.lref('Lnoqn')}")
            e(f"s_waitcnt vmcnt({8 if OPT['qearly'] else 0})")
            q_frag_reads(g)
            g.drain()
            g.label("Lnoqn")
        else:
            e("s_barrier")
        T0, T1, X, D0, R, N, E1, Q_ = [V_T + i for i in range(8)]
        INV = [V_U, V_U + 1]
        for b in range(2):
            e(f"v_mov_b32 {v(T0)}, {v(V_L + b)}")
            e(f"v_mov_b32 {v(T1)}, {v(V_L + b)}")
            e("s_nop 1")
            e(f"v_permlane32_swap_b32 {v(T0)}, {v(T1)}")
            e(f"v_add_f32 {v(X)}, {v(T0)}, {v(T1)}")
            # 1.0f / x exactly as hipcc expands it (v_div_scale / v_rcp / Newton / v_div_fmas / v_div_fixup)
            e(f"v_div_scale_f32 {v(D0)}, {sr(S_TMP)}, {v(X)}, {v(X)}, 1.0")
            e(f"v_rcp_f32 {v(R)}, {v(D0)}")
            e(f"v_div_scale_f32 {v(N)}, vcc, 1.0, {v(X)}, 1.0")
            e("s_nop 0")
            e(f"v_fma_f32 {v(E1)}, -{v(D0)}, {v(R)}, 1.0")
            e(f"v_fmac_f32 {v(R)}, {v(E1)}, {v(R)}")
            e(f"v_mul_f32 {v(Q_)}, {v(N)}, {v(R)}")
            e(f"v_fma_f32 {v(E1)}, -{v(D0)}, {v(Q_)}, {v(N)}")
            e(f"v_fmac_f32 {v(Q_)}, {v(E1)}, {v(R)}")
            e(f"v_fma_f32 {v(D0)}, -{v(D0)}, {v(Q_)}, {v(N)}")
            e(f"v_div_fmas_f32 {v(D0)}, {v(D0)}, {v(R)}, {v(Q_)}")
            e(f"v_div_fixup_f32 {v(INV[b])}, {v(D0)}, {v(X)}, 1.0")
            e(f"v_cmp_lt_f32 vcc, 0, {v(X)}")
            e(f"v_cndmask_b32 {v(INV[b])}, 0, {v(INV[b])}, vcc")
        # this wave's 16 KiB of LDS: base + wave * 16384 (base is a multiple of 1024: the XOR below commutes with the add)
        WA, RD, AD = V_T, V_T + 1, V_T + 2
        if xitem:
            e(f"s_mov_b32 {s(S_TMP)}, {s(S_QST)}")                  # (1 KiB-aligned like the ring: the XOR below commutes with the add)
        else:
            e(f"s_lshl_b32 {s(S_TMP)}, %[wave], 14")
            e(f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, {s(S_RING)}")
        e(f"v_and_b32 {v(WA)}, 31, {v(V_LANE)}")
        e(f"v_lshlrev_b32 {v(WA)}, 8, {v(WA)}")                         # row * 256
        e(f"v_add_u32 {v(WA)}, {v(WA)}, {v(V_HH8)}")                     # + 8 hh
        e(f"v_and_b32 {v(AD)}, 15, {v(V_LANE)}")
        e(f"v_lshlrev_b32 {v(AD)}, 4, {v(AD)}")                          # (row & 15) << 4
        e(f"v_xor_b32 {v(WA)}, {v(WA)}, {v(AD)}")
        e(f"v_add_u32 {v(WA)}, {s(S_TMP)}, {v(WA)}")
        e(f"v_lshlrev_b32 {v(RD)}, 4, {v(V_LANE)}")
        e(f"v_add_u32 {v(RD)}, {s(S_TMP)}, {v(RD)}")
        tmp = [V_U + 3 + i for i in range(4)]                         # (64-bit VGPR operands must start at an even register)
        assert tmp[0] % 2 == 0 and tmp[3] <= LAST_V
        for b in range(2):
            for dt in range(4):
                for g4 in range(4):
                    base = A_O + 64 * b + 16 * dt + 4 * g4
                    for i in range(4):
                        e(f"v_accvgpr_read_b32 {v(tmp[i])}, {a(base + i)}")
                    for i in range(4):
                        e(f"v_mul_f32 {v(tmp[i])}, {v(tmp[i])}, {v(INV[b])}")
                    e(f"v_cvt_pk_bf16_f32 {v(tmp[0])}, {v(tmp[0])}, {v(tmp[1])}")
                    e(f"v_cvt_pk_bf16_f32 {v(tmp[1])}, {v(tmp[2])}, {v(tmp[3])}")
                    e(f"v_xor_b32 {v(AD)}, {(4 * dt + g4) << 4}, {v(WA)}")
                    e(f"ds_write_b64 {v(AD)}, {vr(tmp[0], 2)} offset:{8192 * b}")
        e("s_waitcnt lgkmcnt(0)")
        # global offsets of the four row phases: (lane >> 4) * row stride + 16 * ((lane & 15) ^ (4 k + (lane >> 4)))
        VO = [V_KR, V_KR + 1, V_MX, V_MX + 1]
        e(f"v_and_b32 {v(AD)}, 15, {v(V_LANE)}")
        e(f"v_mul_lo_u32 {v(WA)}, {v(V_LANE4)}, %[ostride]")
        for k in range(4):
            e(f"v_add_u32 {v(VO[k])}, {4 * k}, {v(V_LANE4)}")
            e(f"v_xor_b32 {v(VO[k])}, {v(VO[k])}, {v(AD)}")
            e(f"v_lshl_add_u32 {v(VO[k])}, {v(VO[k])}, 4, {v(WA)}")
        e(f"s_mov_b64 {sr(S_TMP)}, %[obase]")
        e(f"s_lshl_b32 {s(S_TMP2)}, %[ostride], 2")
        e(f"s_mov_b32 {s(S_TMP2 + 1)}, %[nvalid]")
        for i in range(12):
            e(f"ds_read_b128 {vr(V_SA + 4 * i, 4)}, {v(RD)} offset:{1024 * i}")
        e("s_waitcnt lgkmcnt(8)")                                  # (lgkmcnt counts at most 15 outstanding operations)
        for i in range(12, 16):
            e(f"ds_read_b128 {vr(V_SA + 4 * i, 4)}, {v(RD)} offset:{1024 * i}")
        for i in range(16):
            if i >= 4:
                e(f"s_waitcnt lgkmcnt({15 - i})")
            if OPT["heads"] and i == 8:                                 # block B: the second head's rows (none when there is no second head)
                e(f"s_mov_b64 {sr(S_TMP)}, %[obase]")
                e(f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, %[ohs2]")
                e(f"s_addc_u32 {s(S_TMP + 1)}, {s(S_TMP + 1)}, 0")
                if OPT["xitem"]:
                    e(f"s_bitcmp1_b32 %[flags], 2")
                    e(f"s_cselect_b32 {s(S_TMP2 + 1)}, %[nvalid], 0")
                else:
                    e(f"s_mov_b32 {s(S_TMP2 + 1)}, %[nvalidb]")
            e(f"v_cmp_gt_i32 vcc, {s(S_TMP2 + 1)}, {v(V_LANE4)}")        # row 4 i + (lane >> 4) of this wave exists
            e("s_mov_b64 exec, vcc")
            e(f"global_store_dwordx4 {v(VO[i % 4])}, {vr(V_SA + 4 * i, 4)}, {sr(S_TMP)}")
            e(f"s_mov_b64 exec, {sr(S_EXEC)}")
            e(f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, {s(S_TMP2)}")
            e(f"s_addc_u32 {s(S_TMP + 1)}, {s(S_TMP + 1)}, 0")
            e(f"s_add_i32 {s(S_TMP2 + 1)}, {s(S_TMP2 + 1)}, -4")
    if OPT["trace"]:
        e("s_waitcnt vmcnt(0)")
        stamp(g, 6)
        e("s_mov_b64 exec, 1")
        e(f"v_mov_b32 {v(V_T + 1)}, 0")
        for k in range(N_ACC):
            e(f"v_mov_b32 {v(V_T)}, {s(S_TR + 3 + k)}")
            e(f"global_atomic_add {v(V_T + 1)}, {v(V_T)}, %[dbg] offset:{4 * k}")
        e(f"s_mov_b64 exec, {sr(S_EXEC)}")
    if OPT["raw"]:
        g.label("Lrawdone")
    for b in range(2):
        e(f"v_mov_b32 %[m{b}], {v(V_M + b)}")
        e(f"v_mov_b32 %[l{b}], {v(V_L + b)}")
    e(f"s_mov_b32 m0, {s(S_M0SAVE)}")
    return g


def clobbers():
    names = [f"v{i}" for i in range(FIRST_V, LAST_V + 1)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(FIRST_S, LAST_S + 1)]
    return names + ["vcc", "scc", "memory"]


def main():
    g = gen_body()
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_attn_fwd64.py -- do not edit; the per-item body of attn_fwd64_kernel as one inline-asm statement.\n")
        f.write(f"// {len(g.lines)} lines; options {OPT}\n")
        pre = os.environ.get("F64_PREFIX", "VSEL_FWD64")
        f.write(f"#define {pre}_LDS_BYTES {2 * 65536}\n")
        f.write(f"#define {pre}_ASM_TEXT \\\n")
        for ln in g.lines:
            f.write(f'  "{ln}\\n\\t" \\\n')
        f.write('  ""\n')
        f.write(f"#define {pre}_ASM_CLOBBERS \\\n  ")
        cl = clobbers()
        f.write(", ".join(f'"{c}"' for c in cl))
        f.write("\n")
    n_mfma = sum(1 for ln in g.lines if ln.startswith("v_mfma"))
    print(f"wrote {OUT}: {len(g.lines)} instructions / labels, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
