// Var-len causal GQA attention forward for SHORT sequences (gfx950, bf16, head_dim 128): the compressed prefill the selector produces
// (L' = k + text tokens: a few hundred to ~2000 tokens per prompt, batches of prompts packed by cu_seqlens).
//
// Same call sites as attn.hip (flash_attn_varlen_func / the FA2 prefill of
//   qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:827-918, qwen-vl-finetune/qwenvl/train/trainer.py:101-113,
//   llava-ov-15/llavaonevision1_5/modeling_llavaonevision1_5.py:601-705) and the same arithmetic per query row, bit for bit:
// key tiles of 64 aligned to key 0, S^T = K Q^T, lazy-exponent online softmax in the exp2 domain, P rounded to bf16, O^T += V^T P^T.
//
// What differs is who shares what.  In attn.hip a workgroup is 128 / 256 queries of ONE q head: each of a GQA group's 7 (Qwen2.5-VL-7B) /
// 8 (3B) / 4 (LLaVA-OV) q heads streams the group's K / V by itself, an item is 128 x 64 granular under the causal diagonal, and every
// item pays its own prologue (Q rows at a 7 KB stride, the first K / V tile) with nothing to hide it behind.  At L' = 524 that is
// 0.06 - 0.17 of the matrix peak.  Here:
//   * ONE workgroup (8 waves) serves a kv head's WHOLE q-head group on one query tile: wave w = (query slice w / rep, q head w % rep),
//     32 queries per wave, 8 / rep slices per tile (7B: 7 head waves + 1 helper wave on a 32-query tile).  A K / V tile is loaded to LDS
//     once (direct-to-LDS, source-swizzled, all 8 waves issue) and read by every head: rep x fewer tile loads, items, draws, epilogues;
//   * the causal diagonal at 32-key granularity: query tiles start at multiples of 32, so the last key tile of a tile is either whole or
//     its second 32-key block is masked for every query -- that block is skipped (bit-identical: its p would be exactly 0);
//   * the NEXT item's Q rows and first K / V tile are in flight under the LAST tile of the current item: Q arrives through a per-wave
//     8 KiB LDS staging area (whole 256-byte rows by direct-to-LDS loads, no registers), the K / V ring keeps alternating across items,
//     and the work queue is drawn TWO items ahead (the atomic's round trip is never waited for); one barrier per key tile and none
//     between items.
#include "attn_common.h"
// Timing experiments (knock-outs with WRONG results, loader / priority variants) compile only with -DVSEL_EXPERIMENT, which
// visionselector_amd/build.py refuses for the shipped library: one stray -DVSEL_GQA_KO_* can no longer ship an incorrect kernel.
#if (defined(VSEL_GQA_KO_HALF) || defined(VSEL_GQA_KO_S) || defined(VSEL_GQA_KO_MAX) || defined(VSEL_GQA_KO_EXP) || defined(VSEL_GQA_KO_PROBS) || \
     defined(VSEL_GQA_KO_PV) || defined(VSEL_GQA_KO_DMA) || defined(VSEL_GQA_KO_BAR) || defined(VSEL_GQA_KO_EPI) || defined(VSEL_GQA_LOADER) ||   \
     defined(VSEL_GQA_PRIO)) && !defined(VSEL_EXPERIMENT)
#error "VSEL_GQA_KO_* / VSEL_GQA_LOADER / VSEL_GQA_PRIO are timing experiments: add -DVSEL_EXPERIMENT (tools/build_variant.sh), never in the shipped library"
#endif
#include <atomic>

#include <algorithm>
#include <type_traits>

namespace vsel {

using namespace attn;

namespace gqa {
constexpr int kBuf = kTileBytes;                 // 16 KiB per K or V tile
constexpr int kKV = 4 * kBuf;                    // K[2], V[2]
constexpr int kQRegion = 32 * kRowBytes;         // 8 KiB: one wave's 32 query rows
constexpr int kQ0 = kKV;                         // Q staging: 8 waves
constexpr int kCtl = kKV + 8 * kQRegion;         // control words (candidate item)
constexpr int kLds = kCtl + 16;
}  // namespace gqa

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ int g_gqa_work_counter[64 * 8];       // 64 launch slots, one counter each, 32 bytes apart (concurrent streams do not share a line's atomics)

__device__ __forceinline__ bf16x8_t gq_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8_t, v); }
// x ^ K as an asm statement: the compiler would hoist the (loop-invariant) variants out of the item loop and keep every one of them in a
// register across the tile loop, where there is none to spare
template <int K>
__device__ __forceinline__ uint32_t xor_imm(uint32_t x) {
  uint32_t r;
  asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "n"(K), "v"(x));
  return r;
}
// the lane index, re-derived where it is used outside the tile loop (loads, epilogue): everything computed from the kernel's one `lane`
// would be loop-invariant, hoisted, and held in a register across the tile loop
__device__ __forceinline__ int opaque_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// {x, x of lane ^ 32} in some order without the LDS crossbar round trip of __shfl_xor (ds_bpermute_b32): v_permlane32_swap exchanges
// lanes 32..63 of its first operand with lanes 0..31 of its second.  Enough for a commutative combine (max; a + b and b + a are the same
// float).  Inline asm: __builtin_amdgcn_permlane32_swap(u, u) has its two results folded into one by hipcc (ROCm 7.2) -- max(r0, r1)
// became r0.  (s_nop 1: a VALU write needs 2 wait states in front of the swap; the compiler does not see into the asm.)
__device__ __forceinline__ void both_halves(float x, float& a, float& b) {
  a = x;
  b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

#ifdef VSEL_GQA_TRACE
// tools/trace_gqa.py: s_memtime stamps (shader cycles) of waves 0 and 5 of the first 8 workgroups, (tag << 56) | time, in program order
__device__ unsigned long long g_gqa_trace[8][2][1024];
#define GQA_STAMP(tag)                                                                                                  \
  do {                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (tr_on && tr_n < 1024) g_gqa_trace[blockIdx.x][wave == 0 ? 0 : 1][tr_n++] = ((unsigned long long)(tag) << 56) | (__builtin_readcyclecounter() & 0xffffffffffffffull); \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  } while (0)
#else
#define GQA_STAMP(tag) do {} while (0)
#endif

struct GqaItem {
  int seq, kvh, q0, qs, qlen, n_tiles;           // n_tiles == 0: not an item (empty or past the end of the list)
};

__global__ __launch_bounds__(512, 2) void attn_fwd_gqa_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                              const uint16_t* __restrict__ v, const int32_t* __restrict__ cu, int hq, int hkv,
                                                              float scale_log2e, int causal, uint16_t* __restrict__ out, int q_tiles, int n_seq,
                                                              int* __restrict__ counter, PagedKV pg, float* __restrict__ lse) {
  using namespace gqa;
  constexpr int kSteps = 8, kDTiles = 4, kHeadDim = 128;
  // ONE __shared__ object (attn.hip: a second one makes hipcc drain vmcnt in front of every tile's first ds_read)
  __shared__ __attribute__((aligned(1024))) char smem[kLds];
  // control words: s_cand[item parity] = the candidate drawn two items ahead, published in front of an item's FIRST barrier and read
  // behind it (two slots: a wave may reach the next item's first barrier -- after a single-tile item -- while another has not read yet);
  // s_slow = the redraw of the (rare) empty-candidate path, which has its own two barriers
  int* const s_cand = reinterpret_cast<int*>(smem + kCtl);
  int& s_slow = *reinterpret_cast<int*>(smem + kCtl + 8);
  const int rep = hq / hkv;
  const int QW = 8 / rep;                          // 32-query slices per tile
  const int kBlockQ = 32 * QW;
  const int n_pairs = n_seq * hkv;
  const int n_items = q_tiles * n_pairs;
  const int64_t q_rs = pg.q_row_stride ? pg.q_row_stride : (int64_t)hq * kHeadDim;
  const int64_t q_hs = pg.q_row_stride ? pg.q_head_stride : kHeadDim;
  const int64_t kv_rs = pg.kv_row_stride ? pg.kv_row_stride : (int64_t)hkv * kHeadDim;
  const int64_t kv_hs = pg.kv_row_stride ? pg.kv_head_stride : kHeadDim;
  const int64_t v_rs = pg.v_row_stride ? pg.v_row_stride : kv_rs;
  const int64_t v_hs = pg.v_row_stride ? pg.v_head_stride : kv_hs;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef VSEL_GQA_KO_HALF
  const bool head_wave = wave < 4;                 // timing experiment (WRONG results): one computing wave per SIMD
#else
  const bool head_wave = wave < QW * rep;          // (7 q heads: wave 7 only helps with the tile loads)
#endif
#ifdef VSEL_GQA_LOADER
  const bool one_loader = QW * rep == 7;           // exactly one spare wave: it issues every tile load
#else
  const bool one_loader = false;
#endif
  const int qw = head_wave ? wave / rep : 0;       // this wave's 32-query slice
  const int hl = head_wave ? wave % rep : 0;       // ... and q head inside the group
  const int j = lane & 31, hh = lane >> 5;
#ifdef VSEL_GQA_PRIO
  // waves w and w + 4 share a SIMD and the older one wins every arbitration: static priority for the younger half
  if (wave >= 4) __builtin_amdgcn_s_setprio(VSEL_GQA_PRIO);
#endif
#ifdef VSEL_GQA_TRACE
  const bool tr_on = blockIdx.x < 8 && (wave == 0 || wave == 5) && lane == 0;
  int tr_n = 0;
#endif

  // per-lane LDS addresses: K row fragments (rows permuted so that a lane's P registers are 8 consecutive keys), V transposed
  // fragments, this wave's Q rows (B operand: lane (j, hh) reads features 16 st + 8 hh .. +7 of query row j)
  // (chunk_off(row, 2 st + hh) = chunk_off(row, hh) ^ (32 st): the Q addresses are rebuilt from one base once per item)
  uint32_t row_addr_u[kSteps], tr_addr_u[kDTiles][2];
  const uint32_t q_addr0 = lds_u32(smem) + kQ0 + wave * kQRegion + chunk_off(j, hh);
  {
    int row_addr[kSteps], tr_addr[kDTiles][2];
    make_row_addr<kSteps>(row_addr, j, hh);
    make_tr_addr<kDTiles>(tr_addr, lane);
#pragma unroll
    for (int st = 0; st < kSteps; ++st) row_addr_u[st] = lds_u32(smem) + row_addr[st];
#pragma unroll
    for (int dt = 0; dt < kDTiles; ++dt) { tr_addr_u[dt][0] = lds_u32(smem) + tr_addr[dt][0]; tr_addr_u[dt][1] = lds_u32(smem) + tr_addr[dt][1]; }
  }
  // direct-to-LDS loads: instruction i covers tile rows 4 i .. 4 i + 3, lane l lands at (row 4 i + (l >> 4), position l & 15) and
  // fetches source part (l & 15) ^ swz(row).  K / V: wave w issues slices w and w + 8 (swizzle key w & 3 for both); Q: a wave loads
  // its own 8 slices (key i & 3 per slice).
  const uint32_t k_rs_b = (uint32_t)(kv_rs * 2), v_rs_b = (uint32_t)(v_rs * 2), q_rs_b = (uint32_t)(q_rs * 2);
  // whole tiles: a wave-uniform row pointer (scalar ALU) + a per-lane offset that never changes
  const uint32_t lane_off_k = (uint32_t)(lane >> 4) * k_rs_b + (uint32_t)slice_src_part(lane, wave) * 16u;
  const uint32_t lane_off_v = (uint32_t)(lane >> 4) * v_rs_b + (uint32_t)slice_src_part(lane, wave) * 16u;

  auto decode = [&](int item) -> GqaItem {
    GqaItem it{0, 0, 0, 0, 0, 0};
    if (item >= n_items) return it;
    const int level = item / n_pairs, pair = item - level * n_pairs;
    it.seq = pair / hkv;
    it.kvh = pair - it.seq * hkv;
    it.qs = cu[it.seq];
    it.qlen = cu[it.seq + 1] - it.qs;
    it.q0 = (q_tiles - 1 - level) * kBlockQ;       // level 0 = the last (heaviest) tile of the longest sequence
    if (it.q0 >= it.qlen) return it;
    const int kv_end = causal ? min(it.qlen, it.q0 + kBlockQ) : it.qlen;
    it.n_tiles = (kv_end + kTileK - 1) / kTileK;
    return it;
  };
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // K / V tile t of the (sequence, kv head) whose row 0 is at kp / vp (qlen rows) -> ring slot BUF: rows past the end of the sequence
  // replay its last row (finite values, masked later).  kp / vp are per-ITEM values (item_bases below): recomputing the 64-bit
  // products of the sequence base in every round cost ~50 scalar instructions per tile (PMC: 5.7 SALU per MFMA in this kernel).
  auto item_kbase = [&](const GqaItem& it) { return reinterpret_cast<const char*>(k + (int64_t)it.qs * hkv * kHeadDim + it.kvh * kv_hs); };
  auto item_vbase = [&](const GqaItem& it) { return reinterpret_cast<const char*>(v + (int64_t)it.qs * hkv * kHeadDim + it.kvh * v_hs); };
  auto load_tile_at = [&](const char* kp, const char* vp, int qlen, int t, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    if (t * kTileK + kTileK <= qlen) {
      const uint32_t r = (uint32_t)(t * kTileK + 4 * wave);          // (row * stride < 2^32: max_seqlen <= 16384 rows of <= 64 KiB)
      const char* kt = kp + r * k_rs_b;
      const char* vt = vp + r * v_rs_b;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        __builtin_amdgcn_global_load_lds((gptr_t)(kt + lane_off_k), (lptr_t)(smem + BUF * kBuf + (wave + 8 * u) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vt + lane_off_v), (lptr_t)(smem + (2 + BUF) * kBuf + (wave + 8 * u) * 1024), 16, 0, 0);
        kt += 32 * k_rs_b;
        vt += 32 * v_rs_b;
      }
      return;
    }
    const int l = opaque_lane();
    const uint32_t kv_part_b = (uint32_t)slice_src_part(l, wave) * 16u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = wave + 8 * u;
      const uint32_t row = (uint32_t)min(t * kTileK + 4 * i + (l >> 4), qlen - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(kp + (row * k_rs_b + kv_part_b)), (lptr_t)(smem + BUF * kBuf + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(vp + (row * v_rs_b + kv_part_b)), (lptr_t)(smem + (2 + BUF) * kBuf + i * 1024), 16, 0, 0);
    }
  };
  auto load_tile = [&](const GqaItem& it, int t, auto buf_c) { load_tile_at(item_kbase(it), item_vbase(it), it.qlen, t, buf_c); };
  // the whole tile by ONE wave (7 q heads per group: wave 7 has no head and nothing else to do -- the head waves then never stall on
  // the vector-memory pipe, whose address stage a 32 KiB tile pair occupies for ~500 cycles per round)
  auto load_tile_all = [&](const GqaItem& it, int t, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const char* kp = reinterpret_cast<const char*>(k + (int64_t)it.qs * hkv * kHeadDim + it.kvh * kv_hs);
    const char* vp = reinterpret_cast<const char*>(v + (int64_t)it.qs * hkv * kHeadDim + it.kvh * v_hs);
    const int l = opaque_lane();
    const uint32_t part0_b = (uint32_t)((l & 15) ^ ((l >> 4) << 2)) * 16u;
    const int r0 = t * kTileK + (l >> 4);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t row = (uint32_t)min(r0 + 4 * i, it.qlen - 1);
      const uint32_t part_b = part0_b ^ (uint32_t)(16 * (i & 3));
      __builtin_amdgcn_global_load_lds((gptr_t)(kp + (row * k_rs_b + part_b)), (lptr_t)(smem + BUF * kBuf + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(vp + (row * v_rs_b + part_b)), (lptr_t)(smem + (2 + BUF) * kBuf + i * 1024), 16, 0, 0);
    }
  };
  // this wave's 32 query rows of `it` -> its staging area (rows past the end of the sequence replay the last query)
  auto load_q = [&](const GqaItem& it) {
    if (!head_wave) return;
    const char* qp = reinterpret_cast<const char*>(q + (int64_t)it.qs * hq * kHeadDim + (it.kvh * rep + hl) * q_hs);
    const int l = opaque_lane();
    const uint32_t q_part_b = (uint32_t)((l & 15) ^ ((l >> 4) << 2)) * 16u;
    const int r0 = it.q0 + 32 * qw + (l >> 4);
    static_for<0, 8>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      const uint32_t row = (uint32_t)min(r0 + 4 * i, it.qlen - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(qp + (row * q_rs_b + xor_imm<16 * (i & 3)>(q_part_b))),
                                       (lptr_t)(smem + kQ0 + wave * kQRegion + i * 1024), 16, 0, 0);
    });
  };
  // a candidate that is not an item: move the shared counter past the run of empty items it starts (attn_common.h) and draw again.
  // Uniform over the workgroup (cand comes out of LDS); costs two barriers and an atomic round trip per empty candidate.
  // counter == nullptr: the item list is DEALT, not drawn -- workgroup b takes items b, 2 G - 1 - b, 2 G + b, ... (attn_common.h,
  // static_deal_item): with every draw two items ahead the queue hands the first three rounds out in arrival order anyway, and without
  // the mirroring a workgroup that starts on a level's heaviest item keeps the heaviest of every later round (8 x 524: 9 + 5 + 1 tiles
  // against 5 + 1 on the last workgroup).  Only thread 0 draws: deal_round is its private count.
  int deal_round = 1;
  auto draw = [&]() -> int { return counter ? atomicAdd(counter, 1) : static_deal_item(deal_round++); };
  auto validate = [&](int cand) -> GqaItem {
    for (;;) {
      GqaItem it = decode(cand);
      if (it.n_tiles > 0 || cand >= n_items) return it;
      const int level = cand / n_pairs, pair = cand - level * n_pairs;
      if (counter && pair % hkv == 0)
        queue_skip_empty_run(counter, tid, cu, n_seq, hkv, level, pair / hkv,
                             [&](int lv, int ql) { return (q_tiles - 1 - lv) * kBlockQ < ql; });
      if (tid == 0) s_slow = draw();
      __syncthreads();
      cand = __builtin_amdgcn_readfirstlane(s_slow);
      __syncthreads();
    }
  };

  // ---- prologue: the first item is the workgroup's own index, the second is drawn (and waited for) once ----------------------------
  // Rounds 0 and 1 are DEALT in either mode -- items b and 2 G - 1 - b, the mirror (the queue's counter starts at 2 G).  A workgroup whose
  // own item is empty (ragged batches) starts on its mirror item and takes the second one from the queue.
  GqaItem cur = decode((int)blockIdx.x);
  const bool own_empty = counter && cur.n_tiles == 0;
  cur = validate(own_empty ? static_deal_item(1) : (int)blockIdx.x);
  if (cur.n_tiles == 0) return;
  load_q(cur);
  load_tile(cur, 0, std::integral_constant<int, 0>{});
  if (tid == 0) s_slow = (counter && !own_empty) ? static_deal_item(1) : draw();
  __syncthreads();                                  // (also: the loads above have landed for every wave -- vmcnt(0) in front of it)
  int cand = __builtin_amdgcn_readfirstlane(s_slow);
  __syncthreads();
  GqaItem next = validate(cand);
  GqaItem next2{0, 0, 0, 0, 0, 0};                  // the item after `next`: decoded during the current item's first round
  int pend = 0;                                     // thread 0: the draw for that item, in flight
  if (tid == 0) pend = draw();
  int ipar = 0;                                     // item parity (slot of s_cand)
  int t = 0;                                        // tile of the current item

  // per-item state of the wave
  u32x4 qf[kSteps];
  f32x16 o[kDTiles];
  float m_run = -1e30f, l_run = 0.f;
  const char* cur_kp = item_kbase(cur);             // row 0 of the current item's (sequence, kv head) in K / V
  const char* cur_vp = item_vbase(cur);
  int wave_qmin = cur.q0 + qw * 32;
  bool wave_has_rows = head_wave && wave_qmin < cur.qlen;
  int my_q = min(wave_qmin + j, cur.qlen - 1);
  // Q^T fragments out of the staging area (the area is free from then on: output staging, then the next item's rows)
  auto read_q = [&]() {
    static_for<0, kSteps>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      qf[st] = lds_read_b128_asm<0>(xor_imm<32 * st>(q_addr0));
    });
    lds_wait8<0>(qf);
  };
  if (wave_has_rows) read_q();
#pragma unroll
  for (int dt = 0; dt < kDTiles; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  GQA_STAMP(1);

  // one 64-key tile out of ring slot CUR; nkb = visible 32-key blocks of it for this wave's queries (1 or 2, wave-uniform)
  auto tile_body = [&](auto cur_c, int tt, int nkb, auto&& issue_loads) {
    constexpr int CUR = decltype(cur_c)::value;
    const bool two = nkb == 2;
    const int len = cur.qlen;
    // ---- S^T = K Q^T: the second block's K fragments are on their way while the first block's MFMAs run; the direct-to-LDS loads of
    // the next tile are issued behind those MFMAs (their issue costs ~100 cycles each, free while the matrix pipe works).  The second
    // block's reads are issued whether or not it is visible (no counted wait may sit in one arm of a branch, see below).
    f32x16 s[2];                      // s[1] stays undefined when the second block is skipped (a zero fill outside the branch would be 16 moves)
    u32x4 ka0[8], ka1[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) ka0[st] = lds_read_b128_asm<CUR * kBuf>(row_addr_u[st]);
#pragma unroll
    for (int st = 0; st < 7; ++st) ka1[st] = lds_read_b128_asm<CUR * kBuf + 32 * kRowBytes>(row_addr_u[st]);
    lds_wait8<7>(ka0);
    {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int st = 0; st < 8; ++st)
#ifndef VSEL_GQA_KO_S
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gq_bf16x8(ka0[st]), gq_bf16x8(qf[st]), acc, 0, 0, 0);
#else
        acc[st] += __builtin_bit_cast(float, ka0[st][0]) * __builtin_bit_cast(float, qf[st][0]);
#endif
      s[0] = acc;
    }
    issue_loads();
    ka1[7] = lds_read_b128_asm<CUR * kBuf + 32 * kRowBytes>(row_addr_u[7]);
    // the first group of V fragments does not depend on P: its transpose reads fly under the softmax
    u32x2 vr0[8], vr1[8], vr2[8], vr3[8];     // (vr2 / vr3: the second block's groups)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      vr0[2 * dt] = lds_read_tr16_b64_asm<(2 + CUR) * kBuf>(tr_addr_u[dt][0]);
      vr0[2 * dt + 1] = lds_read_tr16_b64_asm<(2 + CUR) * kBuf>(tr_addr_u[dt][1]);
    }
    lds_wait8<8>(ka1);
    if (two) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int st = 0; st < 8; ++st)
#ifndef VSEL_GQA_KO_S
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gq_bf16x8(ka1[st]), gq_bf16x8(qf[st]), acc, 0, 0, 0);
#else
        acc[st] += __builtin_bit_cast(float, ka1[st][0]) * __builtin_bit_cast(float, qf[st][0]);
#endif
      s[1] = acc;
    }
    // ---- mask + online softmax: attn.hip's arithmetic, element for element -------------------------------------------------------
    // a 32-key block needs the mask when its last key lies past the first query of the wave (causal) or past the end of the keys;
    // with query tiles at multiples of 32 that is the diagonal block alone
    const int kmax = causal ? min(len - 1, my_q) : len - 1;
    const int kbase0 = tt * kTileK + 8 * hh;
    float mx = -INFINITY;
    auto mask_max = [&](auto kb_c) {
      constexpr int KB = decltype(kb_c)::value;
      const bool need_mask = __builtin_amdgcn_readfirstlane(
          (int)((tt * kTileK + 32 * KB + 32 > len) || (causal && (tt * kTileK + 32 * KB + 31 > wave_qmin)))) != 0;
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase0 + 32 * KB + 16 * (r >> 3) + (r & 7);
          const float val = key <= kmax ? s[KB][r] : -INFINITY;
          s[KB][r] = val;
          mx = fmaxf(mx, val);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[KB][r]);
      }
    };
#ifndef VSEL_GQA_KO_MAX
    mask_max(std::integral_constant<int, 0>{});
    if (two) mask_max(std::integral_constant<int, 1>{});
#else
    mx = s[0][3];
#endif
    {
      float ha, hb;
      both_halves(mx, ha, hb);
      mx = fmaxf(ha, hb);
    }
    const float m_cand = fmaxf(m_run, mx * scale_log2e);
    const bool moves = m_cand > m_run + kLazyTau;
    if (__any(moves)) {
      const float m_new = moves ? m_cand : m_run;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < kDTiles; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    float psum = 0.f;
    bf16x8_t pf[2][2];
    auto probs = [&](auto kb_c) {
      constexpr int KB = decltype(kb_c)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef VSEL_GQA_KO_EXP
        const float p = fmaf(s[KB][r], scale_log2e, -m_run);
#else
        const float p = __builtin_amdgcn_exp2f(fmaf(s[KB][r], scale_log2e, -m_run));
#endif
        psum += p;
        pf[KB][r >> 3][r & 7] = (__bf16)p;
      }
    };
#ifndef VSEL_GQA_KO_PROBS
    probs(std::integral_constant<int, 0>{});
    if (two) probs(std::integral_constant<int, 1>{});
#else
    pf[0][0] = gq_bf16x8(qf[0]); pf[0][1] = gq_bf16x8(qf[1]); pf[1][0] = gq_bf16x8(qf[2]); pf[1][1] = gq_bf16x8(qf[3]); psum = s[0][0];
#endif
    l_run += psum;
    // ---- O^T += V^T P^T: (32-key block, 16-key half) groups of 8 transpose reads feeding 4 MFMAs, the next group's reads in flight
    auto issue = [&](auto g_c, u32x2 (&dst)[8]) {
      constexpr int G = decltype(g_c)::value;
      constexpr int OFF = (2 + CUR) * kBuf + (32 * (G >> 1) + 16 * (G & 1)) * kRowBytes;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dst[2 * dt] = lds_read_tr16_b64_asm<OFF>(tr_addr_u[dt][0]);
        dst[2 * dt + 1] = lds_read_tr16_b64_asm<OFF>(tr_addr_u[dt][1]);
      }
    };
    auto pv = [&](auto g_c, u32x2 (&src)[8]) {
      constexpr int G = decltype(g_c)::value;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 w = {src[2 * dt][0], src[2 * dt][1], src[2 * dt + 1][0], src[2 * dt + 1][1]};
#ifndef VSEL_GQA_KO_PV
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gq_bf16x8(w), pf[G >> 1][G & 1], o[dt], 0, 0, 0);
#else
        o[dt][0] += __builtin_bit_cast(float, w[0]) + (float)pf[G >> 1][G & 1][0];
#endif
      }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // Control flow around asm-issued reads: no if / else with MFMAs in both arms (hipcc then keeps O in two register homes and copies
    // all 64 registers at the merge), and no counted wait in two arms (the merge copies the named registers -- in the arm where the
    // copy lands in front of the wait it reads registers whose data has not arrived).  Group 2's reads are therefore issued whether or
    // not the second block is visible (the tile is in LDS either way) and drained unused when it is not.
    issue(I1{}, vr1);
    lds_wait8<8>(vr0);
    pv(I0{}, vr0);
    issue(I2{}, vr2);
    lds_wait8<8>(vr1);
    pv(I1{}, vr1);
    if (two) {
      issue(I3{}, vr3);
      lds_wait8<8>(vr2);
      pv(I2{}, vr2);
      lds_wait8<0>(vr3);
      pv(I3{}, vr3);
    } else {
      // (vr2 is dead on this path: a wait that names it would make hipcc copy the eight registers first)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue of the current item: O^T[d][query] / l as bf16.  A lane holds 4 consecutive features of ONE row per register quad, so
  // direct stores would be 8 bytes at a 7 KB stride, 16 per lane -- store-issue-bound (measured: 4 - 9k cycles per item and wave).  The
  // tile goes through the wave's own staging area instead: the 16-byte chunk c of row r at position c ^ (r & 15), read back as whole
  // rows (4 rows per instruction) and stored as 8 x 1 KiB of whole 256-byte rows.  Wave-private: no barrier.
  auto epilogue = [&]() {
    float l_a, l_b;
    both_halves(l_run, l_a, l_b);
    const float l_tot = l_a + l_b;
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int el = opaque_lane();
    const int e_j = el & 31, e_hh = el >> 5, e_l4 = el >> 4, e_p = el & 15;
    const int head = cur.kvh * rep + hl;
    if (lse && e_hh == 0 && wave_qmin + e_j < cur.qlen)
      lse[(int64_t)(cur.qs + wave_qmin + e_j) * hq + head] = l_tot > 0.f ? (m_run + log2f(l_tot)) * 0.6931471805599453f : -INFINITY;
    char* const stage = smem + kQ0 + wave * kQRegion;
    const uint32_t wa = (uint32_t)(e_j * kRowBytes + 8 * e_hh) ^ (uint32_t)((e_j & 15) << 4);
    // (round 6: the conversion spelled out -- two v_pk_mul_f32 and two v_cvt_pk_bf16_f32 per four values.  Left to hipcc, `(__bf16)(o * inv)`
    // element by element became 48 conversions + 16 v_perm + 16 v_alignbit + 35 moves per wave and item, and the row stores a 64-bit
    // address and an exec round trip each: 290 VALU per item, ~2.8 - 4 k cycles of an item's ~27 k with no MFMA in them.  Same RNE rounding:
    // the same bits.)
    const f32x2 inv2 = {inv, inv};
#pragma unroll
    for (int dt = 0; dt < kDTiles; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x2 lo = f32x2{o[dt][4 * g4], o[dt][4 * g4 + 1]} * inv2;
        const f32x2 hi = f32x2{o[dt][4 * g4 + 2], o[dt][4 * g4 + 3]} * inv2;
        u32x2 pk;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[0]) : "v"(lo[0]), "v"(lo[1]));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[1]) : "v"(hi[0]), "v"(hi[1]));
        *reinterpret_cast<u32x2*>(stage + (wa ^ (uint32_t)((4 * dt + g4) << 4))) = pk;
      }
    // rows 4 i + (lane >> 4), position lane & 15 holds chunk (lane & 15) ^ (row & 15)
    char* const ob = reinterpret_cast<char*>(out + ((int64_t)(cur.qs + wave_qmin) * hq + head) * kHeadDim);
    const uint32_t ostride = (uint32_t)hq * kHeadDim * 2;            // a multiple of 256 bytes: the chunk bits below never carry into it
    const int rows = cur.qlen - wave_qmin;             // rows of this wave that exist (> 0 here)
    u32x4 rowv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowv[i] = *reinterpret_cast<const u32x4*>(stage + i * 1024 + el * 16);
    // byte offset of row 4 i + e_l4, chunk e_p ^ ((4 i + e_l4) & 15):  (e_l4 ostride + i 4 ostride) ^ ((e_p ^ e_l4) << 4) ^ (((4 i) & 15) << 4)
    uint32_t off = (uint32_t)e_l4 * ostride + (uint32_t)((e_p ^ e_l4) << 4);
    const uint32_t step = 4u * ostride;
    const int rows_left = rows - e_l4;                 // row 4 i + e_l4 exists  <=>  4 i < rows_left
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (4 * i < rows_left) *reinterpret_cast<u32x4*>(ob + (off ^ (uint32_t)(((4 * i) & 15) << 4))) = rowv[i];
      off += step;
    }
  };

  // one round = one key tile of the current item out of ring slot CUR; the last round of an item also closes it and opens the next.
  // Returns false when the workgroup is done.
  auto round = [&](auto cur_c) -> bool {
    constexpr int CUR = decltype(cur_c)::value;
    using NXT = std::integral_constant<int, 1 - CUR>;
    const bool last = t + 1 == cur.n_tiles;
    const bool first = t == 0;
    GQA_STAMP(2);
    auto issue_loads = [&]() {
#ifdef VSEL_GQA_KO_DMA
      return;
#endif
      if (one_loader) {
        if (!head_wave) {
          if (!last) load_tile_all(cur, t + 1, NXT{});
          else if (next.n_tiles > 0) load_tile_all(next, 0, NXT{});
        } else if (last && next.n_tiles > 0) {
          load_q(next);
        }
        return;
      }
      if (!last) {
        load_tile_at(cur_kp, cur_vp, cur.qlen, t + 1, NXT{});
      } else if (next.n_tiles > 0) {                 // the next item's rows and first tile fly under this item's last tile
        load_q(next);
        load_tile(next, 0, NXT{});
      }
    };
    // visible 32-key blocks of this tile for the wave's queries (uniform per wave)
    int nkb = 0;
    if (wave_has_rows) {
      const int wave_qmax = min(wave_qmin + 31, cur.qlen - 1);
      if (!causal) nkb = (t * kTileK + 32 < cur.qlen) ? 2 : 1;
      else nkb = (t * kTileK > wave_qmax) ? 0 : ((t * kTileK + 32 > wave_qmax) ? 1 : 2);
    }
    nkb = __builtin_amdgcn_readfirstlane(nkb);
    if (nkb > 0) tile_body(cur_c, t, nkb, issue_loads);
    else issue_loads();
    GQA_STAMP(4);
    if (first && tid == 0) s_cand[ipar] = pend;      // (drawn at the item switch, a whole tile ago)
#ifndef VSEL_GQA_KO_BAR
    __syncthreads();                                 // vmcnt(0) in front of it: every wave's direct loads have landed at the release
#endif
    GQA_STAMP(5);
    if (first) {
      // the item after next: index arithmetic, two dependent cu_seqlens loads and (rarely) the empty-candidate path -- here, in the
      // shadow of the item's remaining tiles, not between two items where every wave would wait for it
      cand = __builtin_amdgcn_readfirstlane(s_cand[ipar]);
      ipar ^= 1;
      next2 = validate(cand);
    }
    if (!last) {
      ++t;
      return true;
    }
    // ---- the item is complete: its rows leave through the staging area, which the next item's Q fragments vacate first -------------
    const bool had_rows = wave_has_rows;
    const int next_qmin = next.q0 + qw * 32;
    const bool next_has_rows = head_wave && next.n_tiles > 0 && next_qmin < next.qlen;
    if (next_has_rows) read_q();                     // (the current item's fragments are dead: its last S is done)
#ifndef VSEL_GQA_KO_EPI
    if (had_rows) epilogue();
#else
    if (had_rows && o[0][0] == 123.f) epilogue();
#endif
    GQA_STAMP(6);
    if (next.n_tiles == 0) return false;
#pragma unroll
    for (int dt = 0; dt < kDTiles; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    m_run = -1e30f;
    l_run = 0.f;
    cur = next;
    cur_kp = item_kbase(cur);
    cur_vp = item_vbase(cur);
    t = 0;
    wave_qmin = next_qmin;
    wave_has_rows = next_has_rows;
    my_q = min(wave_qmin + j, cur.qlen - 1);
    next = next2;
    if (tid == 0) pend = draw();
    GQA_STAMP(7);
    return true;
  };
  for (;;) {
    if (!round(std::integral_constant<int, 0>{})) return;
    if (!round(std::integral_constant<int, 1>{})) return;
  }
}

namespace attn {

// q_tiles = ceil(max_seqlen / (32 * (8 / rep))); counter: one int, set to twice the grid size by the launcher (nullptr: every round dealt)
int attn_fwd_gqa_launch(hipStream_t st, const void* q, const void* k, const void* v, const int32_t* cu_q, int64_t n_seq,
                        int64_t max_seqlen_q, int64_t hq, int64_t hkv, float scale, int causal, void* out, const PagedKV& pg, float* lse,
                        bool deal) {
  const int rep = (int)(hq / hkv);
  const int block_q = 32 * (8 / rep);
  const int64_t q_tiles = cdiv(max_seqlen_q, block_q);
  const int64_t n_items = q_tiles * hkv * n_seq;
  if (n_items >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "too many attention work items");
  const int grid = (int)std::min<int64_t>(n_items, 256);                      // 128 KiB of LDS: one workgroup per CU
  int* counter = nullptr;                         // deal: no counter (the kernel deals the list out: items b, 2 G - 1 - b, ...)
  int slot = -1;                                  // (a counter slot per launch: common.h, queue_slot_acquire)
  if (!deal) {
    if (int rc = queue_slot_acquire(kSlotGqa, st, &slot)) return rc;
    int* counters = nullptr;
    VSEL_HIP_CHECK(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_gqa_work_counter)));
    VSEL_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(counters + 8 * slot), 2 * grid, 1, st));     // (two rounds are dealt: items b and 2 G - 1 - b)
    counter = counters + 8 * slot;
  }
  VSEL_LAUNCH(attn_fwd_gqa_kernel, dim3(grid), dim3(512), 0, st, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, cu_q,
                     (int)hq, (int)hkv, scale * 1.4426950408889634f, causal, (uint16_t*)out, (int)q_tiles, (int)n_seq, counter, pg, lse);
  if (!deal) queue_slot_launched(kSlotGqa, slot, st);
  VSEL_AFTER_LAUNCH(st, "attn_fwd_gqa_kernel");
  return VSEL_OK;
}

}  // namespace attn
}  // namespace vsel

#ifdef VSEL_GQA_TRACE
extern "C" int vsel_debug_read_gqa_trace(unsigned long long* out, int clear) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vsel::g_gqa_trace), sizeof(vsel::g_gqa_trace)) != hipSuccess) return VSEL_ERR_HIP;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(vsel::g_gqa_trace)) != hipSuccess || hipMemset(p, 0, sizeof(vsel::g_gqa_trace)) != hipSuccess) return VSEL_ERR_HIP;
  }
  return VSEL_OK;
}
#endif
