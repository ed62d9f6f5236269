// Var-len causal GQA attention BACKWARD (gfx950, bf16, head_dim 128): dQ, dK, dV for the training path.
//
// The reference trains through flash_attn_varlen_func (qwen-vl-finetune/qwenvl/train/trainer.py:101-113, installed by
// replace_qwen2_vl_attention_class, :150-160): the LLM is frozen but the LIS gradient flows back through every attention
// layer, so the backward of the same op is on the training hot path.  Math = autograd of the in-tree eager formula
// (qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:777-797):
//   P = softmax(S), S = scale * Q K^T (+ causal);  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - D), D_i = sum_d dO_id O_id;
//   dQ = scale * dS K;  dK = scale * dS^T Q;  GQA: dK / dV of a kv head sum over the q heads of its group.
//
// Deterministic by construction (no float atomics, reruns are bit-identical): two kernels, each owning its outputs.
//   attn_bwd_dq_kernel    item = (128-query tile, q head, sequence), loops over 64-key K/V tiles in LDS:
//                         S^T = K Q^T, dP^T = V dO^T (lane = query, so lse / D are per-lane scalars),
//                         dQ^T += K^T dS^T (K^T fragments by ds_read_b64_tr_b16 from the same tile).  Runs FIRST and also leaves
//                         D and the exp2-domain log-sum-exp of its rows in the workspace for the dK / dV kernel (no separate
//                         pass over dO and O).
//   attn_bwd_dkdv_kernel  item = (128-key block, kv head, sequence), K / V fragments stay in registers, loops over the q heads
//                         of the group and over 64-query Q / dO tiles in LDS: S = Q K^T, dP = dO V^T (lane = key),
//                         dV^T += dO^T P, dK^T += Q^T dS (transposed fragments by ds_read_b64_tr_b16).
// P is recomputed from the forward's log-sum-exp (vsel_varlen_attn_fwd_lse).  One LDS layout serves both the row (b128) and
// the transposed (b64_tr) reads without bank conflicts: 256-byte rows, 16-byte part p of row r stored at part p ^ swz(r),
// swz(r) = ((r & 3) << 2) | ((r >> 2) & 3).
#include "attn_common.h"
#include <atomic>

#include <algorithm>
#include <cmath>
#include <type_traits>

namespace vsel {

namespace bwd {

using namespace attn;      // tile layout + fragment addressing shared with the forward (attn_common.h)

constexpr int kD = 128;            // head_dim
constexpr int kRowB = kRowBytes;
constexpr int kTile = kTileRows;   // rows per LDS tile
constexpr int kTileB = kTileBytes;
constexpr float kLog2e = 1.4426950408889634f;

// two fp32 -> packed bf16 pair, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32; same rounding as f32_to_bf16_bits for
// every finite value).  The software form keeps five rounding constants live in registers across the whole item loop, which is
// what pushed the 8-wave dK/dV kernel past 256 registers.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

__device__ int g_bwd_counter[64 * 8];        // 64 launch slots x 8 XCD-local queues (attn_common.h, XcdQueue)

#ifdef VSEL_TRACE
// s_memtime (shader clock) stamps of ONE steady-state dK/dV tile of workgroup 0, one row per wave (tools/trace_attn_bwd.py)
__device__ unsigned long long g_bwd_tile_trace[8][8];
#define VSEL_BWD_STAMP(slot)                                                                        \
  do {                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (trace_on && lane == 0) g_bwd_tile_trace[wave][slot] = __builtin_readcyclecounter();          \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
#else
#define VSEL_BWD_STAMP(slot) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// dQ: the forward's loop with two extra contractions.  4 waves x 32 queries.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    const uint16_t* __restrict__ dout, const uint16_t* __restrict__ out_fwd, const float* __restrict__ lse,
    float* __restrict__ dvec, float* __restrict__ lse2_out,
    const int32_t* __restrict__ cu, int hq, int hkv, float scale, int causal, uint16_t* __restrict__ dq, int q_tiles, int n_seq,
    int slot, int xcd_local) {
  // ONE __shared__ object (a second one makes hipcc drain the direct-to-LDS prefetch with s_waitcnt vmcnt(0) before the first
  // ds_read of every tile: attn.hip)
  __shared__ __attribute__((aligned(16))) char smem[4 * kTileB + 16];  // K[2], V[2], work-item slot
  int& s_item = *reinterpret_cast<int*>(smem + 4 * kTileB);
  char* const k_sm = smem;
  char* const v_sm = smem + 2 * kTileB;
  const int n_items = q_tiles * hq * n_seq;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const float sl2 = scale * kLog2e;
  // per-lane LDS byte offsets inside a tile; the buffer, the 32-key block and the 16-row step add immediates
  int row_addr[8], tr_addr[4][2];
  make_row_addr<8>(row_addr, j, hh);
  make_tr_addr<4>(tr_addr, lane);
  uint32_t row_u[8], tr_u[4][2];                                 // the same as LDS byte addresses (asm reads: base VGPR + immediate)
#pragma unroll
  for (int st = 0; st < 8; ++st) row_u[st] = lds_u32(smem) + row_addr[st];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { tr_u[dt][0] = lds_u32(smem) + tr_addr[dt][0]; tr_u[dt][1] = lds_u32(smem) + tr_addr[dt][1]; }

  // work items: (sequence, kv head) pairs on XCD-local queues; a pair's items = query tiles heaviest first, the group's q heads inner
  const int rep_q = hq / hkv;
  XcdQueue wq{&g_bwd_counter[8 * max(slot, 0)], n_seq * hkv, q_tiles * rep_q, xcc_id(), 0};
  for (int round = 0;; ++round) {
    int t_end, head, seq;
    if (slot < 0 || xcd_local != 1) {
      if (slot == -1 && round > 0) return;
      const int item = slot == -2 ? static_deal_item(round) : slot < 0 ? (int)blockIdx.x : global_queue_next(wq.counters, n_items, &s_item, tid);
      if (item < 0 || item >= n_items) return;
      t_end = item / (hq * n_seq);                               // query tile counted from the heaviest one
      const int rest = item % (hq * n_seq);
      head = rest % hq, seq = rest / hq;
    } else {
      const int item = xcd_queue_next(wq, &s_item, tid);
      if (item < 0) return;
      const int pair = item / wq.per_pair, r = item % wq.per_pair;
      seq = pair / hkv;
      t_end = r / rep_q;
      head = (pair % hkv) * rep_q + r % rep_q;
    }
    const int qs = cu[seq];
    const int len = cu[seq + 1] - qs;
    // Under the causal mask the query tiles are aligned to the END of the sequence, so that its partial tile is the first one
    // (one or two key tiles) instead of the last (len / 64 of them for a handful of queries) -- attn.hip, TAIL; a query's
    // arithmetic does not depend on its tile: bit-identical dQ.
    int q0, q_lim;                                               // this tile's queries: [q0, min(q0 + 128, q_lim))
    // an empty item of the single queue: one workgroup per empty (level, sequence) group moves the counter past the empty run (attn_common.h)
    auto skip_run = [&]() {
      if (slot >= 0 && xcd_local == 2 && head == 0)
        queue_skip_empty_run(wq.counters, tid, cu, n_seq, hq, t_end, seq, [&](int level, int ql) {
          return causal ? ql - level * 128 > 0 : (q_tiles - 1 - level) * 128 < ql;
        });
    };
    if (causal) {
      q_lim = len - t_end * 128;
      if (q_lim <= 0) { skip_run(); continue; }
      q0 = max(0, q_lim - 128);
    } else {
      q_lim = len;
      q0 = (q_tiles - 1 - t_end) * 128;
      if (q0 >= len) { skip_run(); continue; }
    }
    const int kvh = head / (hq / hkv);
    const int my_q = min(q0 + wave * 32 + j, q_lim - 1);
    const bool q_valid = (q0 + wave * 32 + j) < q_lim;
    const int wave_qmax = min(q0 + wave * 32 + 31, q_lim - 1);

    u32x4 qf[8], dof[8];
    {
      const int64_t ro = ((int64_t)(qs + my_q) * hq + head) * kD + 8 * hh;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        qf[st] = *reinterpret_cast<const u32x4*>(q + ro + 16 * st);
        dof[st] = *reinterpret_cast<const u32x4*>(dout + ro + 16 * st);
      }
    }
    // D = sum_d dO * O of this lane's query and its log-sum-exp in the exp2 domain, computed HERE (every valid query row belongs to
    // exactly one dQ item) and left in the workspace for the dK / dV kernel, which runs after this one: the separate pass over
    // dO and O (attn_bwd_dot_kernel, 2.4 % of the backward at 16 x 4096) is gone.  Lane (j, hh) holds 64 of the row's 128
    // features; the two halves are added in the order (features 8 hh' .. of hh' = 0) + (hh' = 1) in both lanes.
    float dpart = 0.f;
    {
      const int64_t ro = ((int64_t)(qs + my_q) * hq + head) * kD + 8 * hh;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const u32x4 ov = *reinterpret_cast<const u32x4*>(out_fwd + ro + 16 * st);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dpart = fmaf(__uint_as_float(dof[st][i] << 16), __uint_as_float(ov[i] << 16), dpart);
          dpart = fmaf(__uint_as_float(dof[st][i] & 0xffff0000u), __uint_as_float(ov[i] & 0xffff0000u), dpart);
        }
      }
    }
    const float dother = __shfl_xor(dpart, 32, 64);
    const float dsum = hh == 0 ? dpart + dother : dother + dpart;
    const float lse2 = lse[(int64_t)(qs + my_q) * hq + head] * kLog2e;
    if (q_valid && hh == 0) {
      dvec[(int64_t)(qs + my_q) * hq + head] = dsum;
      lse2_out[(int64_t)(qs + my_q) * hq + head] = lse2;
    }
    f32x16 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

    const int kv_end = causal ? q_lim : len;
    const int n_tiles = (kv_end + kTile - 1) / kTile;
    // K / V tiles go global -> LDS directly (global_load_lds_dwordx4), swizzle applied to the source address (see the
    // dK / dV kernel below); the mask is evaluated only on tiles that touch the causal diagonal or the end of the keys.
    const int ld_part = slice_src_part(lane, wave) * 8;
    const uint32_t lane_off_kv = (uint32_t)(((lane >> 4) * hkv * kD + ld_part) * 2);     // bytes from a slice's first row, this lane
    const int64_t slice_step_kv = (int64_t)16 * hkv * kD * 2;                             // bytes from slice i to slice i + 4
    auto load_tile = [&](int t, int buf) {
      typedef const __attribute__((address_space(1))) void* gptr_t;
      typedef __attribute__((address_space(3))) void* lptr_t;
      if (t * kTile + kTile <= len) {
        // full tile: wave-uniform 64-bit row base (scalar ALU) + per-lane 32-bit offset computed once per item (the clamped form
        // below costs every load its own min and 64-bit multiply on the vector ALU)
        const int64_t row0 = ((int64_t)(qs + t * kTile + 4 * wave) * hkv + kvh) * kD;
        const char* kp = reinterpret_cast<const char*>(k + row0);
        const char* vp = reinterpret_cast<const char*>(v + row0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = wave + 4 * u;
          __builtin_amdgcn_global_load_lds((gptr_t)(kp + lane_off_kv), (lptr_t)(k_sm + buf * kTileB + i * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gptr_t)(vp + lane_off_kv), (lptr_t)(v_sm + buf * kTileB + i * 1024), 16, 0, 0);
          kp += slice_step_kv;
          vp += slice_step_kv;
        }
        return;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = wave + 4 * u;
        const int kpos = min(t * kTile + 4 * i + (lane >> 4), len - 1);
        const int64_t off = ((int64_t)(qs + kpos) * hkv + kvh) * kD + ld_part;
        __builtin_amdgcn_global_load_lds((gptr_t)(k + off), (lptr_t)(k_sm + buf * kTileB + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(v + off), (lptr_t)(v_sm + buf * kTileB + i * 1024), 16, 0, 0);
      }
    };
    load_tile(0, 0);
    __syncthreads();

    const int kmax = causal ? min(len - 1, my_q) : len - 1;      // last visible key of this lane's query
    // LDS reads of the tile body are issued from inline asm in batches with counted waits (attn_common.h: with the builtins hipcc
    // emits "2 ds_read, s_waitcnt lgkmcnt(0), 2 MFMAs" per k-step -- every MFMA pair behind a full LDS round trip -- and puts
    // s_waitcnt vmcnt(0) in front of the first transposed read of every tile, which drains the direct-to-LDS prefetch of the NEXT
    // tile in the middle of this one).  Same MFMAs in the same order as before: bit-identical dQ.
    auto tile_body = [&](auto cur_c, int t) {
      constexpr int CUR = decltype(cur_c)::value;
      auto block = [&](auto kb_c) {
        constexpr int KB = decltype(kb_c)::value;
        constexpr int kKOff = CUR * kTileB + KB * 32 * kRowB, kVOff = (2 + CUR) * kTileB + KB * 32 * kRowB;
        const int key0 = t * kTile + 32 * KB;
        const bool active = __builtin_amdgcn_readfirstlane((int)(key0 < len && (!causal || key0 <= wave_qmax))) != 0;
        if (!active) return;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        // S^T = K Q^T, dP^T = V dO^T: two batches of 8 row-fragment reads (K and V of 4 k-steps), each feeding 8 MFMAs
        auto sdp_batch = [&](auto h_c) {
          constexpr int H4 = decltype(h_c)::value;
          u32x4 kv[8];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            kv[2 * u] = lds_read_b128_asm<kKOff>(row_u[4 * H4 + u]);
            kv[2 * u + 1] = lds_read_b128_asm<kVOff>(row_u[4 * H4 + u]);
          }
          lds_wait8<0>(kv);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kv[2 * u]), as_bf16x8(qf[4 * H4 + u]), s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kv[2 * u + 1]), as_bf16x8(dof[4 * H4 + u]), dp, 0, 0, 0);
          }
        };
        sdp_batch(std::integral_constant<int, 0>{});
        sdp_batch(std::integral_constant<int, 1>{});
        // K^T fragments of the first 16-key half do not depend on dS: in flight under the P / dS arithmetic
        u32x2 g0[8], g1[8];
        auto issue = [&](auto m_c, u32x2 (&dst)[8]) {
          constexpr int M = decltype(m_c)::value;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            dst[2 * dt] = lds_read_tr16_b64_asm<kKOff + 16 * M * kRowB>(tr_u[dt][0]);
            dst[2 * dt + 1] = lds_read_tr16_b64_asm<kKOff + 16 * M * kRowB>(tr_u[dt][1]);
          }
        };
        issue(std::integral_constant<int, 0>{}, g0);
        const bool need_mask = __builtin_amdgcn_readfirstlane(
            (int)(key0 + 32 > len || (causal && key0 + 31 > q0 + wave * 32))) != 0;
        bf16x8_t dsf[2];
        if (need_mask) {
          const int rel = kmax - (key0 + 8 * hh);                 // register r's key visible iff 16*(r>>3) + (r&7) <= rel
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = (16 * (r >> 3) + (r & 7)) <= rel ? __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lse2)) : 0.f;
            dsf[r >> 3][r & 7] = (__bf16)(p * (dp[r] - dsum));
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lse2));
            dsf[r >> 3][r & 7] = (__bf16)(p * (dp[r] - dsum));
          }
        }
        // dQ^T += K^T dS^T: the second half's reads go out behind the first half's wait, counted lgkmcnt
        auto dq_mfma = [&](auto m_c, u32x2 (&src)[8]) {
          constexpr int M = decltype(m_c)::value;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const u32x4 w = {src[2 * dt][0], src[2 * dt][1], src[2 * dt + 1][0], src[2 * dt + 1][1]};
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w), dsf[M], acc[dt], 0, 0, 0);
          }
        };
        issue(std::integral_constant<int, 1>{}, g1);
        lds_wait8<8>(g0);
        dq_mfma(std::integral_constant<int, 0>{}, g0);
        lds_wait8<0>(g1);
        dq_mfma(std::integral_constant<int, 1>{}, g1);
      };
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
    };
    for (int t = 0; t < n_tiles; t += 2) {
      if (t + 1 < n_tiles) load_tile(t + 1, 1);
      tile_body(std::integral_constant<int, 0>{}, t);
      __syncthreads();                       // also drains this wave's global_load_lds queue before the release
      if (t + 1 >= n_tiles) break;
      if (t + 2 < n_tiles) load_tile(t + 2, 0);
      tile_body(std::integral_constant<int, 1>{}, t + 1);
      __syncthreads();
    }

    if (q_valid) {
      uint16_t* op = dq + ((int64_t)(qs + my_q) * hq + head) * kD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = 32 * dt + 8 * g4 + 4 * hh;
          uint2 pk;
          pk.x = pack_bf16x2(acc[dt][4 * g4] * scale, acc[dt][4 * g4 + 1] * scale);
          pk.y = pack_bf16x2(acc[dt][4 * g4 + 2] * scale, acc[dt][4 * g4 + 3] * scale);
          *reinterpret_cast<uint2*>(op + d0) = pk;
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK, dV: 4 waves x 32 keys; K / V fragments in registers, Q / dO tiles of 64 queries streamed through LDS.
// ---------------------------------------------------------------------------------------------------------------------
// SPLIT = false: item = (key block, kv head, sequence), the q heads of the group are looped inside and dK / dV leave as bf16.
// SPLIT = true (few items: short or few sequences): item = (key block, Q head, sequence), 7x (rep x) more parallelism and an
// rep x shorter critical path; each item writes its fp32 partial [T, hq, 128] and attn_bwd_group_sum_kernel adds the heads
// of a group in fixed order -- still no atomics.
template <bool SPLIT>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    const uint16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ dvec,
    const int32_t* __restrict__ cu, int hq, int hkv, float scale, int causal, uint16_t* __restrict__ dk,
    uint16_t* __restrict__ dv, float* __restrict__ dk_part, float* __restrict__ dv_part, int k_blocks, int n_seq, int slot,
    int xcd_local_arg) {
  // ONE __shared__ object (see attn_bwd_dq_kernel): Q[2], dO[2], lse[2][kTile], D[2][kTile], work-item slot
  __shared__ __attribute__((aligned(16))) char smem[4 * kTileB + 4 * kTile * sizeof(float) + 16];
  float (*lse_sm)[kTile] = reinterpret_cast<float (*)[kTile]>(smem + 4 * kTileB);
  float (*d_sm)[kTile] = reinterpret_cast<float (*)[kTile]>(smem + 4 * kTileB + 2 * kTile * sizeof(float));
  int& s_item = *reinterpret_cast<int*>(smem + 4 * kTileB + 4 * kTile * sizeof(float));
  char* const q_sm = smem;
  char* const do_sm = smem + 2 * kTileB;
  const int heads_per_item_dim = SPLIT ? hq : hkv;
  const int n_items = k_blocks * heads_per_item_dim * n_seq;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const float sl2 = scale * kLog2e;
  const int rep = hq / hkv;
  // per-lane LDS byte offsets inside a tile; the buffer, the 32-query sub-block and the 16-row step add immediates
  int row_addr[8], tr_addr[4][2];
  make_row_addr<8>(row_addr, j, hh);
  make_tr_addr<4>(tr_addr, lane);

  // work items: (sequence, kv head) pairs on XCD-local queues; a pair's items = (SPLIT: q head of the group, outer) key blocks,
  // block 0 first -- under the causal mask it is seen by the most queries
  const int updown = xcd_local_arg >> 4, xcd_local = xcd_local_arg & 15;      // (bit 4: attn_common.h, dkdv_walks_up)
  XcdQueue wq{&g_bwd_counter[8 * max(slot, 0)], n_seq * hkv, k_blocks * (SPLIT ? rep : 1), xcc_id(), 0, updown};
  for (int round = 0;; ++round) {
    int kblock, hsel, seq;
    if (slot < 0 || xcd_local != 1) {
      if (slot == -1 && round > 0) return;
      const int item = slot == -2 ? static_deal_item(round) : slot < 0 ? (int)blockIdx.x : global_queue_next(wq.counters, n_items, &s_item, tid);
      if (item < 0 || item >= n_items) return;
      kblock = item / (heads_per_item_dim * n_seq);
      const int rest = item % (heads_per_item_dim * n_seq);
      hsel = rest % heads_per_item_dim, seq = rest / heads_per_item_dim;
    } else {
      const int item = xcd_queue_next(wq, &s_item, tid);
      if (item < 0) return;
      const int pair = item / wq.per_pair, r = item % wq.per_pair;
      seq = pair / hkv;
      kblock = r % k_blocks;
      hsel = SPLIT ? (pair % hkv) * rep + r / k_blocks : pair % hkv;
    }
    const int kvh = SPLIT ? hsel / rep : hsel;
    const int qs = cu[seq];
    const int len = cu[seq + 1] - qs;
    const int k0 = kblock * 128;
    if (k0 >= len) {      // empty item of the single queue: move the counter past the empty run (attn_common.h), once per (level, sequence) group
      if (slot >= 0 && xcd_local == 2 && hsel == 0)
        queue_skip_empty_run(wq.counters, tid, cu, n_seq, heads_per_item_dim, kblock, seq, [&](int level, int ql) { return level * 128 < ql; });
      continue;
    }
    const int kw0 = k0 + 32 * wave;
    const int my_k = min(kw0 + j, len - 1);
    const bool k_valid = (kw0 + j) < len;

    u32x4 kf[8], vf[8];
    {
      const int64_t ro = ((int64_t)(qs + my_k) * hkv + kvh) * kD + 8 * hh;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        kf[st] = *reinterpret_cast<const u32x4*>(k + ro + 16 * st);
        vf[st] = *reinterpret_cast<const u32x4*>(v + ro + 16 * st);
      }
    }
    f32x16 dka[4], dva[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dka[dt][r] = 0.f; dva[dt][r] = 0.f; }

    const int q_begin = causal ? k0 : 0;                         // k0 is a multiple of 128, hence of the 64-query tile
    const int tiles_per_head = (len - q_begin + kTile - 1) / kTile;
    const int n_iter = SPLIT ? tiles_per_head : tiles_per_head * rep;

    // Q / dO tiles go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write pass).  The LDS
    // image of such a load is lane-linear (wave-uniform base + 16 * lane), so the swizzle is applied to the SOURCE address:
    // LDS position (row, part') receives global part part' ^ swz(row).  A wave-instruction covers 4 rows; wave w issues
    // instructions w, w+4, w+8, w+12, for which swz(row) = ((lane >> 4) << 2) | w is a per-lane constant.
    // Tile order: query tiles OUTER from the sequence's END downward, the item's q heads INNER.  Under the causal mask the items of a
    // sequence (its key blocks) start at different first queries but all end at the last one, so walking down from the end puts the
    // items that run side by side on the SAME Q / dO tile at the same time (one HBM fetch, the rest L2 hits), and a ragged sequence's
    // one partial tile is the first tile of every head instead of the last.  attn_bwd_dkdv8_kernel and the generated
    // attn_bwd_dkdv64_kernel (tools/gen_attn_bwd_dkdv64.py, order=td) walk the same way: dK / dV are bit-identical across the forms.
    const int first_head = SPLIT ? hsel : kvh * rep, n_heads = SPLIT ? 1 : rep;
    const int q_last = q_begin + (tiles_per_head - 1) * kTile;
    const bool up = dkdv_walks_up(kvh, updown, len) != 0;             // odd kv heads walk UPWARD from the item's first query (attn_common.h)
    const int q_first = up ? q_begin : q_last, q_step = up ? kTile : -kTile;
    int ld_qt = q_first, ld_head = first_head;                   // (query tile, head) of the NEXT tile to load
    const int ld_part = slice_src_part(lane, wave) * 8;
    auto load_tile = [&](int buf) {
      typedef const __attribute__((address_space(1))) void* gptr_t;
      typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = wave + 4 * u;
        const int qpos = min(ld_qt + 4 * i + (lane >> 4), len - 1);
        const int64_t off = ((int64_t)(qs + qpos) * hq + ld_head) * kD + ld_part;
        __builtin_amdgcn_global_load_lds((gptr_t)(q + off), (lptr_t)(q_sm + buf * kTileB + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(dout + off), (lptr_t)(do_sm + buf * kTileB + i * 1024), 16, 0, 0);
      }
      if (wave < 2) {
        const int64_t o = (int64_t)(qs + min(ld_qt + lane, len - 1)) * hq + ld_head;
        if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(lse + o), (lptr_t)(&lse_sm[buf][0]), 4, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(dvec + o), (lptr_t)(&d_sm[buf][0]), 4, 0, 0);
      }
      if (++ld_head == first_head + n_heads) { ld_head = first_head; ld_qt += q_step; }
    };
    load_tile(0);
    __syncthreads();

#ifdef VSEL_TRACE
    bool trace_on = false;
#endif
    int qt = q_first, qt_heads = 0;                              // query tile being processed; heads done on it
    // one 64-query tile from LDS buffer CUR (compile-time, so that every LDS address is a per-lane base + an immediate)
    auto tile_body = [&](auto cur_c) {
      constexpr int CUR = decltype(cur_c)::value;
      const bool visible = __builtin_amdgcn_readfirstlane((int)(!causal || qt + kTile - 1 >= kw0)) != 0;
      if (visible) {
        // the masked variant also zeroes the padded query rows of a head's last tile (their Q / dO rows replay row len-1)
        const bool need_mask = __builtin_amdgcn_readfirstlane((int)((causal && kw0 + 31 > qt) || qt + kTile > len)) != 0;
        const char* qtile = smem + CUR * kTileB;
        const char* dotile = smem + (2 + CUR) * kTileB;
        VSEL_BWD_STAMP(0);
        f32x16 s[2], dp[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[qb][r] = 0.f; dp[qb][r] = 0.f; }
#pragma unroll
          for (int st = 0; st < 8; ++st) {
            const bf16x8_t aq = as_bf16x8(*reinterpret_cast<const u32x4*>(qtile + row_addr[st] + qb * 32 * kRowB));
            const bf16x8_t ad = as_bf16x8(*reinterpret_cast<const u32x4*>(dotile + row_addr[st] + qb * 32 * kRowB));
            s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, as_bf16x8(kf[st]), s[qb], 0, 0, 0);
            dp[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, as_bf16x8(vf[st]), dp[qb], 0, 0, 0);
          }
        }
        VSEL_BWD_STAMP(1);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          bf16x8_t pf[2], dsf[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int rb = 32 * qb + 16 * m;                      // + 8*hh through the per-lane base
            const f32x4 l0 = *reinterpret_cast<const f32x4*>(&lse_sm[CUR][rb] + 8 * hh);
            const f32x4 l1 = *reinterpret_cast<const f32x4*>(&lse_sm[CUR][rb + 4] + 8 * hh);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(&d_sm[CUR][rb] + 8 * hh);
            const f32x4 e1 = *reinterpret_cast<const f32x4*>(&d_sm[CUR][rb + 4] + 8 * hh);
            const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
            const float dv8[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
            if (need_mask) {
              const int rel = causal ? kw0 + j - (qt + rb + 8 * hh) : -(1 << 30);    // key visible iff rel <= e
              const int qlim = len - (qt + rb + 8 * hh);                              // query row real iff e < qlim
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int r = 8 * m + e;
                const float p = (rel <= e && e < qlim) ? __builtin_amdgcn_exp2f(fmaf(s[qb][r], sl2, -lv[e])) : 0.f;
                pf[m][e] = (__bf16)p;
                dsf[m][e] = (__bf16)(p * (dp[qb][r] - dv8[e]));
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int r = 8 * m + e;
                const float p = __builtin_amdgcn_exp2f(fmaf(s[qb][r], sl2, -lv[e]));
                pf[m][e] = (__bf16)p;
                dsf[m][e] = (__bf16)(p * (dp[qb][r] - dv8[e]));
              }
            }
          }
          VSEL_BWD_STAMP(2 + 2 * qb);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              typedef __attribute__((address_space(3))) bf16x4_t* lds_p;
              const int off = (32 * qb + 16 * m) * kRowB;
              const bf16x4_t d_lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(dotile + tr_addr[dt][0] + off));
              const bf16x4_t d_hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(dotile + tr_addr[dt][1] + off));
              const bf16x4_t q_lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(qtile + tr_addr[dt][0] + off));
              const bf16x4_t q_hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(qtile + tr_addr[dt][1] + off));
              dva[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(d_lo, d_hi, 0, 1, 2, 3, 4, 5, 6, 7), pf[m],
                                                                dva[dt], 0, 0, 0);
              dka[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(q_lo, q_hi, 0, 1, 2, 3, 4, 5, 6, 7), dsf[m],
                                                                dka[dt], 0, 0, 0);
            }
          VSEL_BWD_STAMP(3 + 2 * qb);
        }
      }
      if (++qt_heads == n_heads) { qt_heads = 0; qt += q_step; }
    };

    for (int it = 0; it < n_iter; it += 2) {
#ifdef VSEL_TRACE
      trace_on = blockIdx.x == 0 && round == 0 && it == ((n_iter / 2) & ~1);   // mid-item: steady state (the first tiles of a launch see every workgroup's prologue burst)
#endif
      if (it + 1 < n_iter) load_tile(1);
      tile_body(std::integral_constant<int, 0>{});
      VSEL_BWD_STAMP(6);
      __syncthreads();                       // also drains this wave's global_load_lds queue (vmcnt(0)) before the release
      VSEL_BWD_STAMP(7);
#ifdef VSEL_TRACE
      trace_on = false;
#endif
      if (it + 1 >= n_iter) break;
      if (it + 2 < n_iter) load_tile(0);
      tile_body(std::integral_constant<int, 1>{});
      __syncthreads();
    }

    if (SPLIT) {
      if (k_valid) {
        const int64_t ro = ((int64_t)(qs + my_k) * hq + hsel) * kD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = 32 * dt + 8 * g4 + 4 * hh;
            *reinterpret_cast<f32x4*>(dk_part + ro + d0) =
                f32x4{dka[dt][4 * g4], dka[dt][4 * g4 + 1], dka[dt][4 * g4 + 2], dka[dt][4 * g4 + 3]};
            *reinterpret_cast<f32x4*>(dv_part + ro + d0) =
                f32x4{dva[dt][4 * g4], dva[dt][4 * g4 + 1], dva[dt][4 * g4 + 2], dva[dt][4 * g4 + 3]};
          }
      }
    } else if (k_valid) {
      const int64_t ro = ((int64_t)(qs + my_k) * hkv + kvh) * kD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = 32 * dt + 8 * g4 + 4 * hh;
          uint2 pk;
          pk.x = pack_bf16x2(dka[dt][4 * g4] * scale, dka[dt][4 * g4 + 1] * scale);
          pk.y = pack_bf16x2(dka[dt][4 * g4 + 2] * scale, dka[dt][4 * g4 + 3] * scale);
          *reinterpret_cast<uint2*>(dk + ro + d0) = pk;
          pk.x = pack_bf16x2(dva[dt][4 * g4], dva[dt][4 * g4 + 1]);
          pk.y = pack_bf16x2(dva[dt][4 * g4 + 2], dva[dt][4 * g4 + 3]);
          *reinterpret_cast<uint2*>(dv + ro + d0) = pk;
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK, dV with TWO waves per SIMD (round 3).  The 4-wave kernel above holds 128 accumulator + 64 K / V operand registers per wave,
// i.e. one wave per SIMD, and a lone wave retires one LDS wave-instruction per ~29 cycles whatever its width (tools/lds_tr_bench.hip):
// its transposed-read phases (two reads per MFMA) cannot beat ~58 cycles per MFMA and its P / dS arithmetic runs with the matrix
// pipe idle (profiles/r03_attn_bwd_tile_timeline.txt).  Here a workgroup is 8 waves on the same 128 keys: wave (kb, qh) owns key
// block kb (32 keys) and the 32-query HALF qh of every 64-query tile, so each SIMD carries a pair of waves with independent
// instruction streams (the per-wave LDS rate doubles per SIMD, one wave's VALU runs under the other's MFMAs).  To fit 256 registers
// the K / V fragments are not resident: the item's 128-key K and V tiles are staged in LDS once (64 KiB) and read as row fragments
// next to the Q / dO fragments.  Each wave of a pair accumulates dK / dV over its own half of the queries; at the end of the item the
// qh = 1 waves hand their fp32 accumulators over through LDS and the qh = 0 waves add them (own + partner, fixed order) and store --
// still no atomics, reruns bit-identical.
// LDS: K 32 KiB | V 32 KiB | Q[2] 32 KiB | dO[2] 32 KiB | lse[2][64], D[2][64] | work-item slot  (the first 128 KiB double as the
// hand-over buffer).
// ---------------------------------------------------------------------------------------------------------------------
template <bool SPLIT>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkdv2_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    const uint16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ dvec,
    const int32_t* __restrict__ cu, int hq, int hkv, float scale, int causal, uint16_t* __restrict__ dk,
    uint16_t* __restrict__ dv, float* __restrict__ dk_part, float* __restrict__ dv_part, int k_blocks, int n_seq, int slot,
    int xcd_local_arg, int split_heads) {
  constexpr int kKV = 2 * kTileB;                               // one 128-key tile
  __shared__ __attribute__((aligned(16))) char smem[2 * kKV + 4 * kTileB + 4 * kTile * sizeof(float) + 16];
  char* const k_sm = smem;
  char* const v_sm = smem + kKV;
  char* const q_sm = smem + 2 * kKV;
  char* const do_sm = smem + 2 * kKV + 2 * kTileB;
  float (*lse_sm)[kTile] = reinterpret_cast<float (*)[kTile]>(smem + 2 * kKV + 4 * kTileB);
  float (*d_sm)[kTile] = reinterpret_cast<float (*)[kTile]>(smem + 2 * kKV + 4 * kTileB + 2 * kTile * sizeof(float));
  int& s_item = *reinterpret_cast<int*>(smem + 2 * kKV + 4 * kTileB + 4 * kTile * sizeof(float));
  // SPLIT: split_heads q heads per item (1: one head; k > 1: the group in ceil(rep / k) parts, a part's partial in its first head's rows --
  // attn_bwd_dkdv64.hip, bwd_split_heads below)
  const int parts = SPLIT ? (hq / hkv + split_heads - 1) / split_heads : 1;
  const int heads_per_item_dim = hkv * parts;
  const int n_items = k_blocks * heads_per_item_dim * n_seq;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = wave & 3, qh = wave >> 2;
  const int j = lane & 31, hh = lane >> 5;
  const float sl2 = scale * kLog2e;
  const int rep = hq / hkv;
  // Row-fragment addresses are rebuilt per use instead of held in 16 registers (256 per wave is the budget of two waves per SIMD):
  // chunk_off(row, 2 st + hh) = row * 256 + (((2 st) ^ (swz(row) ^ hh)) << 4), i.e. a per-lane base and a per-lane XOR term against
  // the compile-time constant 32 st -- two VALU per address.
  int tr_addr[4][2];
  make_tr_addr<4>(tr_addr, lane);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { tr_addr[dt][0] += qh * 32 * kRowB; tr_addr[dt][1] += qh * 32 * kRowB; }
  uint32_t tr_u[4][2];                                           // LDS byte addresses of the per-lane bases (for the asm reads)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {      // (based at the Q / dO area: the 16-bit ds offset field then reaches both tensors' buffers)
    tr_u[dt][0] = lds_u32(smem) + 2 * kKV + tr_addr[dt][0];
    tr_u[dt][1] = lds_u32(smem) + 2 * kKV + tr_addr[dt][1];
  }
  const int q_row = perm_row(j) + 32 * qh;                       // A fragment row of the wave's own 32-query half
  const int q_base = q_row * kRowB, q_x = (swz(q_row) ^ hh) << 4;
  const int kv_row = 32 * kb + j;                                // B fragment: key kw0 + j, features 16 st + 8 hh .. + 7
  const int kv_base = kv_row * kRowB, kv_x = (swz(kv_row) ^ hh) << 4;
  auto q_addr = [&](int st) { return q_base + ((32 * st) ^ q_x); };
  auto kv_addr = [&](int st) { return kv_base + ((32 * st) ^ kv_x); };

  // work items: (sequence, kv head) pairs on XCD-local queues; a pair's items = (SPLIT: q head of the group, outer) key blocks,
  // block 0 first -- under the causal mask it is seen by the most queries
  const int updown = xcd_local_arg >> 4, xcd_local = xcd_local_arg & 15;
  XcdQueue wq{&g_bwd_counter[8 * max(slot, 0)], n_seq * hkv, k_blocks * parts, xcc_id(), 0, updown};
  for (int round = 0;; ++round) {
    int kblock, hsel, seq;
    if (slot < 0 || xcd_local != 1) {
      if (slot == -1 && round > 0) return;
      const int item = slot == -2 ? static_deal_item(round) : slot < 0 ? (int)blockIdx.x : global_queue_next(wq.counters, n_items, &s_item, tid);
      if (item < 0 || item >= n_items) return;
      kblock = item / (heads_per_item_dim * n_seq);
      const int rest = item % (heads_per_item_dim * n_seq);
      hsel = rest % heads_per_item_dim, seq = rest / heads_per_item_dim;
    } else {
      const int item = xcd_queue_next(wq, &s_item, tid);
      if (item < 0) return;
      const int pair = item / wq.per_pair, r = item % wq.per_pair;
      seq = pair / hkv;
      kblock = r % k_blocks;
      hsel = (pair % hkv) * parts + r / k_blocks;
    }
    const int kvh = hsel / parts;
    const int first_head = kvh * rep + (SPLIT ? (hsel % parts) * split_heads : 0);
    const int n_heads = SPLIT ? min(split_heads, rep - (hsel % parts) * split_heads) : rep;
    const int qs = cu[seq];
    const int len = cu[seq + 1] - qs;
    const int k0 = kblock * 128;
    if (k0 >= len) {      // empty item of the single queue: move the counter past the empty run (attn_common.h), once per (level, sequence) group
      if (slot >= 0 && xcd_local == 2 && hsel == 0)
        queue_skip_empty_run(wq.counters, tid, cu, n_seq, heads_per_item_dim, kblock, seq, [&](int level, int ql) { return level * 128 < ql; });
      continue;
    }
    const int kw0 = k0 + 32 * kb;
    const int my_k = min(kw0 + j, len - 1);
    const bool k_valid = (kw0 + j) < len;

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int ld_part = slice_src_part(lane, wave) * 8;
    const uint32_t lane_off_q = (uint32_t)(((lane >> 4) * hq * kD + ld_part) * 2);     // bytes from a slice's first row, this lane
    const int64_t slice_step_q = (int64_t)32 * hq * kD * 2;                             // bytes from slice i to slice i + 8
    {
      // the item's K / V tiles: 32 one-KiB slices each, wave w issues slices w, w + 8, w + 16, w + 24 (source-swizzled)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = wave + 8 * u;
        const int kpos = min(k0 + 4 * i + (lane >> 4), len - 1);
        const int64_t off = ((int64_t)(qs + kpos) * hkv + kvh) * kD + ld_part;
        __builtin_amdgcn_global_load_lds((gptr_t)(k + off), (lptr_t)(k_sm + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(v + off), (lptr_t)(v_sm + i * 1024), 16, 0, 0);
      }
    }
    f32x16 dka[4], dva[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dka[dt][r] = 0.f; dva[dt][r] = 0.f; }

    const int q_begin = causal ? k0 : 0;
    const int tiles_per_head = (len - q_begin + kTile - 1) / kTile;
    const int n_iter = tiles_per_head * n_heads;
    const int q_last = q_begin + (tiles_per_head - 1) * kTile;   // tiles outer from the END downward (odd kv heads: upward), heads inner (attn_bwd_dkdv_kernel)
    const bool up = dkdv_walks_up(kvh, updown, len) != 0;
    const int q_first = up ? q_begin : q_last, q_step = up ? kTile : -kTile;
    int ld_qt = q_first, ld_head = first_head;
    auto load_tile = [&](int buf) {
      if (ld_qt + kTile <= len) {
        // full tile: a wave-uniform 64-bit row base on the scalar ALU plus the per-lane 32-bit offset computed once per item
        // (the clamped form below costs every load its own min and 64-bit multiply on the vector ALU -- attn.hip, load_kv)
        const int64_t row0 = ((int64_t)(qs + ld_qt + 4 * wave) * hq + ld_head) * kD;
        const char* qp = reinterpret_cast<const char*>(q + row0);
        const char* dp = reinterpret_cast<const char*>(dout + row0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = wave + 8 * u;
          __builtin_amdgcn_global_load_lds((gptr_t)(qp + lane_off_q), (lptr_t)(q_sm + buf * kTileB + i * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gptr_t)(dp + lane_off_q), (lptr_t)(do_sm + buf * kTileB + i * 1024), 16, 0, 0);
          qp += slice_step_q;
          dp += slice_step_q;
        }
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = wave + 8 * u;
          const int qpos = min(ld_qt + 4 * i + (lane >> 4), len - 1);
          const int64_t off = ((int64_t)(qs + qpos) * hq + ld_head) * kD + ld_part;
          __builtin_amdgcn_global_load_lds((gptr_t)(q + off), (lptr_t)(q_sm + buf * kTileB + i * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gptr_t)(dout + off), (lptr_t)(do_sm + buf * kTileB + i * 1024), 16, 0, 0);
        }
      }
      if (wave < 2) {
        const int64_t o = (int64_t)(qs + min(ld_qt + lane, len - 1)) * hq + ld_head;
        if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(lse + o), (lptr_t)(&lse_sm[buf][0]), 4, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(dvec + o), (lptr_t)(&d_sm[buf][0]), 4, 0, 0);
      }
      if (++ld_head == first_head + n_heads) { ld_head = first_head; ld_qt += q_step; }
    };
    load_tile(0);
    __syncthreads();

#ifdef VSEL_TRACE
    bool trace_on = false;
#endif
    int qt = q_first, qt_heads = 0;
    auto tile_body = [&](auto cur_c) {
      constexpr int CUR = decltype(cur_c)::value;
      const int qw0 = qt + 32 * qh;                              // first query of the wave's half
      const bool visible = __builtin_amdgcn_readfirstlane((int)((!causal || qw0 + 31 >= kw0) && qw0 < len)) != 0;
      if (visible) {
        const bool need_mask = __builtin_amdgcn_readfirstlane((int)((causal && kw0 + 31 > qw0) || qw0 + 32 > len)) != 0;
        const char* qtile = q_sm + CUR * kTileB;
        const char* dotile = do_sm + CUR * kTileB;
        VSEL_BWD_STAMP(0);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          const bf16x8_t aq = as_bf16x8(*reinterpret_cast<const u32x4*>(qtile + q_addr(st)));
          const bf16x8_t ad = as_bf16x8(*reinterpret_cast<const u32x4*>(dotile + q_addr(st)));
          const bf16x8_t kf = as_bf16x8(*reinterpret_cast<const u32x4*>(k_sm + kv_addr(st)));
          const bf16x8_t vf = as_bf16x8(*reinterpret_cast<const u32x4*>(v_sm + kv_addr(st)));
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf, s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf, dp, 0, 0, 0);
        }
        // Priority by phase: 1 through the two MFMA phases, 0 through the P / dS arithmetic.  The two waves of a SIMD then take turns --
        // the one feeding the matrix pipe wins the issue arbitration over its partner's VALU work -- instead of one wave winning
        // every phase (whichever half is older or statically raised finished ~1000 cycles early and waited at the barrier).
        // Same-box A/B: dK/dV kernel -1.6 % against a static bump of the second-dispatched half, -2.5 % against none.
        __builtin_amdgcn_s_setprio(0);
        VSEL_BWD_STAMP(1);
        bf16x8_t pf[2], dsf[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int rb = 32 * qh + 16 * m;                        // + 8*hh through the per-lane base
          const f32x4 l0 = *reinterpret_cast<const f32x4*>(&lse_sm[CUR][rb] + 8 * hh);
          const f32x4 l1 = *reinterpret_cast<const f32x4*>(&lse_sm[CUR][rb + 4] + 8 * hh);
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(&d_sm[CUR][rb] + 8 * hh);
          const f32x4 e1 = *reinterpret_cast<const f32x4*>(&d_sm[CUR][rb + 4] + 8 * hh);
          const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
          const float dv8[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
          if (need_mask) {
            const int rel = causal ? kw0 + j - (qt + rb + 8 * hh) : -(1 << 30);    // key visible iff rel <= e
            const int qlim = len - (qt + rb + 8 * hh);                              // query row real iff e < qlim
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int r = 8 * m + e;
              const float p = (rel <= e && e < qlim) ? __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lv[e])) : 0.f;
              pf[m][e] = (__bf16)p;
              dsf[m][e] = (__bf16)(p * (dp[r] - dv8[e]));
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int r = 8 * m + e;
              const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lv[e]));
              pf[m][e] = (__bf16)p;
              dsf[m][e] = (__bf16)(p * (dp[r] - dv8[e]));
            }
          }
        }
        __builtin_amdgcn_s_setprio(1);
        VSEL_BWD_STAMP(2);
        // The transposed dO / Q fragments are read from inline asm, a group (16-query half m, d-tile dt) = 4 reads feeding 2 MFMAs,
        // the next group in flight behind the current one's MFMAs, counted lgkmcnt waits.  With the read BUILTIN hipcc puts
        // `s_waitcnt vmcnt(0)` in front of the first transposed read of every tile (the trap attn.hip documents): the direct-to-LDS
        // prefetch of the NEXT tile, issued at the top of this one, was drained HERE, with only the S / dP and P / dS phases to
        // cover its ~2 us, instead of at the barrier a whole tile later.
        {
          constexpr int kQOff = CUR * kTileB, kDoOff = 2 * kTileB + CUR * kTileB;      // relative to the Q / dO area (tr_u)
          auto issue = [&](auto g_c, u32x2 (&dst)[4]) __attribute__((always_inline)) {
            constexpr int G = decltype(g_c)::value, M = G >> 2, DT = G & 3;
            dst[0] = lds_read_tr16_b64_asm<kDoOff + 16 * M * kRowB>(tr_u[DT][0]);
            dst[1] = lds_read_tr16_b64_asm<kDoOff + 16 * M * kRowB>(tr_u[DT][1]);
            dst[2] = lds_read_tr16_b64_asm<kQOff + 16 * M * kRowB>(tr_u[DT][0]);
            dst[3] = lds_read_tr16_b64_asm<kQOff + 16 * M * kRowB>(tr_u[DT][1]);
          };
          auto fma2 = [&](auto g_c, u32x2 (&src)[4]) __attribute__((always_inline)) {
            constexpr int G = decltype(g_c)::value, M = G >> 2, DT = G & 3;
            const u32x4 wd = {src[0][0], src[0][1], src[1][0], src[1][1]};
            const u32x4 wq = {src[2][0], src[2][1], src[3][0], src[3][1]};
            dva[DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wd), pf[M], dva[DT], 0, 0, 0);
            dka[DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wq), dsf[M], dka[DT], 0, 0, 0);
          };
          u32x2 ra[4], rb[4];
          issue(std::integral_constant<int, 0>{}, ra);
          issue(std::integral_constant<int, 1>{}, rb);
          lds_wait4<4>(ra); fma2(std::integral_constant<int, 0>{}, ra); issue(std::integral_constant<int, 2>{}, ra);
          lds_wait4<4>(rb); fma2(std::integral_constant<int, 1>{}, rb); issue(std::integral_constant<int, 3>{}, rb);
          lds_wait4<4>(ra); fma2(std::integral_constant<int, 2>{}, ra); issue(std::integral_constant<int, 4>{}, ra);
          lds_wait4<4>(rb); fma2(std::integral_constant<int, 3>{}, rb); issue(std::integral_constant<int, 5>{}, rb);
          lds_wait4<4>(ra); fma2(std::integral_constant<int, 4>{}, ra); issue(std::integral_constant<int, 6>{}, ra);
          lds_wait4<4>(rb); fma2(std::integral_constant<int, 5>{}, rb); issue(std::integral_constant<int, 7>{}, rb);
          lds_wait4<4>(ra); fma2(std::integral_constant<int, 6>{}, ra);
          lds_wait4<0>(rb); fma2(std::integral_constant<int, 7>{}, rb);
        }
        VSEL_BWD_STAMP(3);
      }
      if (++qt_heads == n_heads) { qt_heads = 0; qt += q_step; }
    };

    for (int it = 0; it < n_iter; it += 2) {
#ifdef VSEL_TRACE
      trace_on = blockIdx.x == 0 && round == 0 && it == ((n_iter / 2) & ~1);   // mid-item: steady state (the first tiles of a launch see every workgroup's prologue burst)
#endif
      if (it + 1 < n_iter) load_tile(1);
      tile_body(std::integral_constant<int, 0>{});
      VSEL_BWD_STAMP(4);
      __syncthreads();
      VSEL_BWD_STAMP(5);
#ifdef VSEL_TRACE
      trace_on = false;
#endif
      if (it + 1 >= n_iter) break;
      if (it + 2 < n_iter) load_tile(0);
      tile_body(std::integral_constant<int, 1>{});
      __syncthreads();
    }
    // ---- pair hand-over: qh = 1 -> LDS [kb][dk / dv][dt][r / 4][lane][4], qh = 0 adds (own + partner) and stores ------------
    // (the loop's last barrier has passed: nobody reads the tiles any more, and no direct-to-LDS load is in flight)
    float* xch = reinterpret_cast<float*>(smem) + (size_t)kb * (2 * 4 * 4 * 64 * 4);
    if (qh == 1) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          *reinterpret_cast<f32x4*>(xch + ((0 * 4 + dt) * 4 + g4) * 256 + lane * 4) =
              f32x4{dka[dt][4 * g4], dka[dt][4 * g4 + 1], dka[dt][4 * g4 + 2], dka[dt][4 * g4 + 3]};
          *reinterpret_cast<f32x4*>(xch + ((1 * 4 + dt) * 4 + g4) * 256 + lane * 4) =
              f32x4{dva[dt][4 * g4], dva[dt][4 * g4 + 1], dva[dt][4 * g4 + 2], dva[dt][4 * g4 + 3]};
        }
    }
    __syncthreads();
    if (qh == 0) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 pk = *reinterpret_cast<const f32x4*>(xch + ((0 * 4 + dt) * 4 + g4) * 256 + lane * 4);
          const f32x4 pv = *reinterpret_cast<const f32x4*>(xch + ((1 * 4 + dt) * 4 + g4) * 256 + lane * 4);
#pragma unroll
          for (int z = 0; z < 4; ++z) { dka[dt][4 * g4 + z] += pk[z]; dva[dt][4 * g4 + z] += pv[z]; }
        }
      if (SPLIT) {
        if (k_valid) {
          const int64_t ro = ((int64_t)(qs + my_k) * hq + first_head) * kD;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const int d0 = 32 * dt + 8 * g4 + 4 * hh;
              *reinterpret_cast<f32x4*>(dk_part + ro + d0) =
                  f32x4{dka[dt][4 * g4], dka[dt][4 * g4 + 1], dka[dt][4 * g4 + 2], dka[dt][4 * g4 + 3]};
              *reinterpret_cast<f32x4*>(dv_part + ro + d0) =
                  f32x4{dva[dt][4 * g4], dva[dt][4 * g4 + 1], dva[dt][4 * g4 + 2], dva[dt][4 * g4 + 3]};
            }
        }
      } else if (k_valid) {
        const int64_t ro = ((int64_t)(qs + my_k) * hkv + kvh) * kD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = 32 * dt + 8 * g4 + 4 * hh;
            uint2 pk;
            pk.x = pack_bf16x2(dka[dt][4 * g4] * scale, dka[dt][4 * g4 + 1] * scale);
            pk.y = pack_bf16x2(dka[dt][4 * g4 + 2] * scale, dka[dt][4 * g4 + 3] * scale);
            *reinterpret_cast<uint2*>(dk + ro + d0) = pk;
            pk.x = pack_bf16x2(dva[dt][4 * g4], dva[dt][4 * g4 + 1]);
            pk.y = pack_bf16x2(dva[dt][4 * g4 + 2], dva[dt][4 * g4 + 3]);
            *reinterpret_cast<uint2*>(dv + ro + d0) = pk;
          }
      }
    }
    __syncthreads();                           // the hand-over buffer is the next item's K / V / Q / dO
  }
}

// dK[t, g, :] = scale * sum_{h in group g} dk_part[t, h, :] (heads added in ascending order), dV likewise without the scale.
// One thread per 4 consecutive d.
__global__ __launch_bounds__(256) void attn_bwd_group_sum_kernel(const float* __restrict__ dk_part, const float* __restrict__ dv_part,
                                                                 int64_t total, int hq, int hkv, float scale,
                                                                 uint16_t* __restrict__ dk, uint16_t* __restrict__ dv, int step) {
  // step = q heads per item of the launch in front: partial rows exist for heads 0, step, 2 step, ... of a group
  const int rep = hq / hkv;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over total * hkv * 32
  if (idx >= total * hkv * 32) return;
  const int c = (int)(idx & 31);
  const int64_t tg = idx >> 5;
  const int g = (int)(tg % hkv);
  const int64_t t = tg / hkv;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < rep; r += step) {
    const int64_t o = ((t * hq + g * rep + r) * kD) + 4 * c;
    a += *reinterpret_cast<const f32x4*>(dk_part + o);
    b += *reinterpret_cast<const f32x4*>(dv_part + o);
  }
  uint2 pk;
  pk.x = pack_bf16x2(a.x * scale, a.y * scale);
  pk.y = pack_bf16x2(a.z * scale, a.w * scale);
  *reinterpret_cast<uint2*>(dk + tg * kD + 4 * c) = pk;
  pk.x = pack_bf16x2(b.x, b.y);
  pk.y = pack_bf16x2(b.z, b.w);
  *reinterpret_cast<uint2*>(dv + tg * kD + 4 * c) = pk;
}

}  // namespace bwd
}  // namespace vsel

using namespace vsel;

#ifdef VSEL_TRACE
extern "C" int vsel_debug_read_bwd_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(bwd::g_bwd_tile_trace), sizeof(bwd::g_bwd_tile_trace)) == hipSuccess ? VSEL_OK : VSEL_ERR_HIP;
}
#endif

// Items of the in-kernel-group dK/dV pass below which the per-q-head split (fp32 partials + group sum) is used instead:
// measured (tools/bench_attn_bwd.py): the split wins up to ~300 items (4 x 2368: 635 vs 802 us) and loses at 576 (16 x 1100).
// Re-measured with the 8-wave dK/dV kernel (whole backward, us, auto / no split / split): 1 x 2368 (76 items) 210 / 505 / 216,
// 2 x 2368 (152) 377 / 552 / 369, 8 x 524 (160) 175 / 173 / 165, 8 x 1100 (288) 443 / 383 / 438, 4 x 2368 (304) 728 / 738 / 725,
// 3 x 4096 (384) 1426 / 1398 / 1414, 6 x 2368 (456) 1086 / 940 / 1099: the crossover is below 288 items now.
static constexpr int64_t kSplitBelowItems = 288;
static constexpr int64_t kPartItemsTarget = 400;     // groups are cut in the fewest parts that give this many items (bwd_split_heads)
// knob VSEL_KNOB_ATTN_BWD_SPLIT (include/vsel_debug.h): -1 = choose by item count, 0 / 1 = force (tests)

namespace vsel { namespace bwd {
int dq64_launch(hipStream_t st, const void* q, const void* k, const void* v, const void* dout, const void* out, const float* lse,
                float* dvec, float* lse2, const int32_t* cu, int64_t n_seq, int64_t max_seqlen, int64_t hq, int64_t hkv, float scale,
                int causal, void* dq, int xcd_local);          // attn_bwd_dq64.hip
int dkdv64_launch(hipStream_t st, const void* q, const void* k, const void* v, const void* dout, const float* lse2, const float* dvec,
                  const int32_t* cu, int64_t n_seq, int64_t max_seqlen, int64_t hq, int64_t hkv, float scale, int causal, void* dk, void* dv,
                  float* dk_part, float* dv_part, int split, int xcd_local);   // attn_bwd_dkdv64.hip (split = q heads per item, 0 = the group)
} }

// the 64-rows-per-wave dQ pass from this many tokens in the longest sequence (same-process A/B, tools/exp_dq64_shapes.py,
// profiles/r04_dq64_shapes.txt: 1 x 2368 +5 %, 16 x 2368 +12 %, 1 x 4096 +17 %, 16 x 4096 +17 %, 2 x 8192 +17 %; 16 x 1100 -1 %, 32 x 524 -13 %)
constexpr int64_t kDq64FromTokens = 1024;         // (profiles/r04_dq64_shapes.txt: +2 ... +3 % at 1100, +9 ... +20 % from 2368, -10 % at 32 x 524)
constexpr int64_t kDkdv64FromTokens = 1024;       // q heads inside the item (profiles/r04_dkdv64_shapes.txt: +4 ... +8 % at 1100, +15 ... +22 % from 2368, level at 524)
constexpr int64_t kDkdv64SplitFromTokens = 2048;  // per-q-head split form (few items): +5 % at 1 x 2368, +12 % at 1 x 4096, +15 % at 1 x 8192

// -> q heads per dK / dV item: 0 = the whole group inside the item, 1 = one q head per item, k > 1 = the group in ceil(rep / k) PARTS (the
// one-wave-per-SIMD kernel and the 8-wave kernel; the 4-wave kernel falls back to 1).  The unsplit grid is causal-imbalanced while it has fewer than ~2 rounds of items (heaviest item / mean
// load = 128 / (n_seq * key blocks)): the heaviest item IS the run time.  Splitting a group's heads over P items shortens it P x at the
// price of P fp32 partial rows per key and the group-sum launch; measured (tools/exp_dkdv_parts.py, profiles/r04_dkdv_parts.txt; us, group /
// per head / best part form): 3 x 1100 186 / 115 / 95 (4 parts), 4 x 1100 183 / 143 / 114 (4), 5 x 1100 183 / 178 / 137 (2 - 3), 6 x 1100
// 187 / 215 / 149 (2), 4 x 2000 326 / 349 / 260 (2), 4 x 2368 364 / 401 / 318 (2), 3 x 4096 693 / 742 / 633 (2); 8-wave kernel: 6 x 524
// 115 / 88 / 69 (4), 10 x 524 114 / 134 / 87 (2), 8 x 800 161 / 180 / 119 (2), 12 x 300 70 / 88 / 59 (2); from 400 items (fewer for shorter
// sequences) the unsplit form wins, below ~100 the per-head form.  Rule: the fewest parts that give >= kPartItemsTarget items.
static int bwd_split_heads(int64_t total, int64_t n_seq, int64_t max_seqlen, int64_t hq, int64_t hkv) {
  if (hq == hkv) return 0;
  const int rep = (int)(hq / hkv);
  const int forced = knob(VSEL_KNOB_ATTN_BWD_SPLIT);
  const int g_dkdv64 = knob(VSEL_KNOB_ATTN_BWD_DKDV64);
  // (the kernels that have the part form: dkdv64 and the 8-wave kernel)
  const bool parts_ok = rep >= 4 && (g_dkdv64 == 1 || (g_dkdv64 < 0 && max_seqlen >= kDkdv64FromTokens) || knob(VSEL_KNOB_ATTN_BWD_WAVES) == 8);
  if (forced >= 0) return forced >= 2 ? (parts_ok ? (rep + forced - 1) / forced : 0) : forced;       // (forced k >= 2: k parts per group)
  int64_t items = cdiv(max_seqlen, 128) * hkv * n_seq;
  if (!parts_ok) return items < kSplitBelowItems ? 1 : 0;
  // ragged batches: the list's length comes from the LONGEST sequence, the load from all of them -- heaviest item / mean load =
  // 512 nb_max / (hkv sum nb_i^2), between [512 / (hkv n nb_mean)] and that times nb_max / nb_mean: count the items a uniform batch of the
  // MEAN length would have and divide by sqrt(max / mean) (uniform batches: unchanged; dividing by max / mean itself over-split 24 - 32
  // prompts of 131 ... 947 tokens, -2 ... -11 %; counting by the longest sequence missed +10 ... +25 % on 4 - 16 of them)
  const int64_t mean_len = cdiv(total, n_seq);
  if (mean_len < max_seqlen)
    items = std::max<int64_t>(1, (int64_t)((double)(cdiv(mean_len, 128) * hkv * n_seq) / std::sqrt((double)max_seqlen / (double)mean_len)));
  // the unsplit form from this many items (short items carry more fixed cost each: the crossover falls with the sequence length)
  const int64_t group_from = max_seqlen >= 2048 ? 400 : max_seqlen >= 1024 ? 320 : max_seqlen >= 450 ? 300 : 250;
  if (items >= group_from) return 0;
  const int parts = (int)cdiv(max_seqlen >= 450 ? kPartItemsTarget : 250, items);
  if (parts > 4) return 1;
  return (rep + parts - 1) / parts;                 // (1 when rep <= parts: the per-head form)
}
static bool bwd_use_split(int64_t total, int64_t n_seq, int64_t max_seqlen, int64_t hq, int64_t hkv) { return bwd_split_heads(total, n_seq, max_seqlen, hq, hkv) != 0; }

static size_t bwd_workspace_bytes_for(int64_t total, int64_t hq, bool split) {
  size_t bytes = (((size_t)total * (size_t)hq * sizeof(float) + 255) & ~(size_t)255) * 2;   // D, lse * log2(e)
  if (split) bytes += 2 * (size_t)total * (size_t)hq * bwd::kD * sizeof(float);             // dK / dV partials per q head
  return bytes;
}
extern "C" size_t vsel_varlen_attn_bwd_workspace_bytes(int64_t total, int64_t hq, int64_t hkv, int64_t n_seq, int64_t max_seqlen) {
  if (total < 1 || hq < 1 || hkv < 1 || n_seq < 1 || max_seqlen < 1) return 0;
  return bwd_workspace_bytes_for(total, hq, bwd_use_split(total, n_seq, max_seqlen, hq, hkv));
}

extern "C" int vsel_varlen_attn_bwd(void* stream, const void* dout, const void* q, const void* k, const void* v, const void* out,
                                    const float* lse, const int32_t* cu_seqlens, int64_t n_seq, int64_t max_seqlen,
                                    int64_t total, int64_t hq, int64_t hkv, int64_t d, float scale, int causal, void* workspace,
                                    size_t workspace_bytes, void* dq, void* dk, void* dv) {
  if (!dout || !q || !k || !v || !out || !lse || !cu_seqlens || !workspace || !dq || !dk || !dv)
    return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (d != bwd::kD) return fail(VSEL_ERR_UNSUPPORTED, "head_dim %lld != 128", (long long)d);
  if (n_seq < 1 || max_seqlen < 1 || total < 1 || hq < 1 || hkv < 1 || hq % hkv != 0 || n_seq > (1 << 24) || hq > 65535)
    return fail(VSEL_ERR_INVALID, "bad attention shape (n_seq=%lld max_seqlen=%lld hq=%lld hkv=%lld)", (long long)n_seq,
                (long long)max_seqlen, (long long)hq, (long long)hkv);
  if (((uintptr_t)dout | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dq | (uintptr_t)dk |
       (uintptr_t)dv | (uintptr_t)workspace) & 15)
    return fail(VSEL_ERR_INVALID, "attention tensors must be 16-byte aligned");
  // the dK / dV item form is decided ONCE per call (the knobs are read here and nowhere below): the workspace check and the launch cannot
  // disagree when a knob changes on another thread in between.  NOTE: the form depends on the call's total / mean / longest length, and each
  // form associates the fp32 sums of a group's q heads differently -- unlike the forward, a sequence's dK / dV may differ by one bf16
  // rounding with what it is packed with (every form is deterministic and within the gradient gate of the fp64 oracle).
  const int split_heads0 = bwd_split_heads(total, n_seq, max_seqlen, hq, hkv);
  if (workspace_bytes < bwd_workspace_bytes_for(total, hq, split_heads0 != 0))
    return fail(VSEL_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  const int64_t rows = total * hq;
  const size_t d_bytes = ((size_t)rows * sizeof(float) + 255) & ~(size_t)255;
  float* dvec = (float*)workspace;
  float* lse2 = (float*)((char*)workspace + d_bytes);
  int* counters = nullptr;
  VSEL_HIP_CHECK(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(bwd::g_bwd_counter)));
  // (a counter slot per queued launch: common.h, queue_slot_acquire; `taken` = the slot to report as launched, or -1)
  auto take_slot = [&](int64_t n_items, int64_t resident, int& slot, int& taken) -> int {
    slot = -1;
    taken = -1;
    if (attn_static_deal(n_items, resident, true)) {
      slot = -2;
    } else if (n_items > resident) {
      if (int rc = queue_slot_acquire(kSlotBwd, st, &taken)) return rc;
      slot = taken;
      if (hipMemsetAsync(counters + 8 * slot, 0, 8 * sizeof(int), st) != hipSuccess) return fail(VSEL_ERR_HIP, "hipMemsetAsync(counter)");
    }
    return VSEL_OK;
  };
  // dQ: on from 2048 tokens x 32 pairs (16 x 2368 1078 -> 1041 us, 16 x 4096 2912 -> 2866; 4 x 2368 +8 %, 4 x 8192 +0.7 % stay off);
  // dK / dV: from 4096 tokens x 16 pairs (16 x 4096 3950 -> 3865 us, 4 x 8192 3820 -> 3810; 16 x 2368 -- 19 key blocks per pair, a
  // ragged fit on 32 CUs -- +8.5 % stays off)
  // (2 = single queue whose counter jumps over runs of empty items: ragged batches, knob attn_skip_empty)
  const int skip_empty = n_seq > 1 && knob(VSEL_KNOB_ATTN_SKIP_EMPTY) != 0 ? 2 : 0;
  const int xcd_local_dq = attn_use_xcd_queues(max_seqlen, n_seq * hkv, 2048, 32) ? 1 : skip_empty;
  // bit 4: odd kv heads walk their query tiles upward, pairs queued in couples (attn_common.h, dkdv_walks_up; knob attn_bwd_updown)
  const int xcd_local_dkdv = (attn_use_xcd_queues(max_seqlen, n_seq * hkv, 4096, 16) ? 1 : skip_empty) |
                             (causal && hkv % 2 == 0 && knob(VSEL_KNOB_ATTN_BWD_UPDOWN) != 0 ? 16 : 0);
  // dQ first: it also leaves D = rowsum(dO * O) and the exp2-domain log-sum-exp in the workspace for the dK / dV kernel
  const int g_dq64 = knob(VSEL_KNOB_ATTN_BWD_DQ64);
  if (g_dq64 == 1 || (g_dq64 < 0 && kDq64FromTokens > 0 && max_seqlen >= kDq64FromTokens)) {
    // long sequences: 64 query rows per wave, one wave per SIMD, hand-scheduled unit pipeline (attn_bwd_dq64.hip); same dQ / D / lse2
    if (int rc = bwd::dq64_launch(st, q, k, v, dout, out, lse, dvec, lse2, cu_seqlens, n_seq, max_seqlen, hq, hkv, scale, causal, dq,
                                  xcd_local_dq))
      return rc;
  } else {
    const int q_tiles = (int)cdiv(max_seqlen, 128);
    const int64_t n_items = (int64_t)q_tiles * hq * n_seq;
    if (n_items >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "too many attention work items");
    int slot, taken;
    if (int rc = take_slot(n_items, 512, slot, taken)) return rc;
    VSEL_LAUNCH(bwd::attn_bwd_dq_kernel, dim3((unsigned)std::min<int64_t>(n_items, 512)), dim3(256), 0, st,
                       (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (const uint16_t*)dout, (const uint16_t*)out, lse, dvec, lse2,
                       cu_seqlens,
                       (int)hq, (int)hkv, scale, causal, (uint16_t*)dq, q_tiles, (int)n_seq, slot, xcd_local_dq);
    queue_slot_launched(kSlotBwd, taken, st);
    VSEL_AFTER_LAUNCH(st, "attn_bwd_dq_kernel");
  }
  const int g_dkdv64 = knob(VSEL_KNOB_ATTN_BWD_DKDV64);
  {
    int split_heads = split_heads0;
    const bool split = split_heads != 0;
    const bool dkdv64 = g_dkdv64 == 1 || (g_dkdv64 < 0 && max_seqlen >= (split_heads == 1 ? kDkdv64SplitFromTokens : kDkdv64FromTokens));
    if (!dkdv64 && split_heads > 1 && knob(VSEL_KNOB_ATTN_BWD_WAVES) != 8) split_heads = 1;     // (the 4-wave kernel: per-head items only)
    const int k_blocks = (int)cdiv(max_seqlen, 128);
    const int64_t n_items = (int64_t)k_blocks * hkv * (split ? cdiv(hq / hkv, split_heads) : 1) * n_seq;
    if (n_items >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "too many attention work items");
    float* dk_part = split ? (float*)((char*)workspace + 2 * d_bytes) : nullptr;
    float* dv_part = split ? dk_part + (size_t)rows * bwd::kD : nullptr;
    if (dkdv64) {
      // long sequences: one wave per SIMD with the unit pipeline (attn_bwd_dkdv64.hip), either item form; dK / dV as the 4-wave kernel's
      if (int rc = bwd::dkdv64_launch(st, q, k, v, dout, lse2, dvec, cu_seqlens, n_seq, max_seqlen, hq, hkv, scale, causal, dk, dv, dk_part,
                                      dv_part, split_heads, xcd_local_dkdv))
        return rc;
    } else {
      int slot, taken;
      if (int rc = take_slot(n_items, 256, slot, taken)) return rc;
      const dim3 grid((unsigned)std::min<int64_t>(n_items, 256));
      const bool w8 = knob(VSEL_KNOB_ATTN_BWD_WAVES) == 8;
#define VSEL_DKDV_ARGS (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (const uint16_t*)dout, lse2, dvec, cu_seqlens, (int)hq, \
                         (int)hkv, scale, causal, (uint16_t*)dk, (uint16_t*)dv, dk_part, dv_part, k_blocks, (int)n_seq, slot, xcd_local_dkdv
      if (w8) {
        if (split) VSEL_LAUNCH((bwd::attn_bwd_dkdv2_kernel<true>), grid, dim3(512), 0, st, VSEL_DKDV_ARGS, split_heads);
        else VSEL_LAUNCH((bwd::attn_bwd_dkdv2_kernel<false>), grid, dim3(512), 0, st, VSEL_DKDV_ARGS, 0);
      } else {
        if (split) VSEL_LAUNCH((bwd::attn_bwd_dkdv_kernel<true>), grid, dim3(256), 0, st, VSEL_DKDV_ARGS);
        else VSEL_LAUNCH((bwd::attn_bwd_dkdv_kernel<false>), grid, dim3(256), 0, st, VSEL_DKDV_ARGS);
      }
#undef VSEL_DKDV_ARGS
      queue_slot_launched(kSlotBwd, taken, st);
      VSEL_AFTER_LAUNCH(st, "attn_bwd_dkdv_kernel");
    }
    if (split) {
      // rows past a sequence's end never exist in the packed layout, so every (t, h) partial row was written
      VSEL_LAUNCH(bwd::attn_bwd_group_sum_kernel, dim3((unsigned)cdiv(total * hkv * 32, 256)), dim3(256), 0, st, dk_part,
                         dv_part, total, (int)hq, (int)hkv, scale, (uint16_t*)dk, (uint16_t*)dv, split_heads);
      VSEL_AFTER_LAUNCH(st, "attn_bwd_group_sum_kernel");
    }
  }
  return VSEL_OK;
}
