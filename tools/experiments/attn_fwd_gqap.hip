// Group-shared short-sequence attention forward, SOFTWARE-PIPELINED form (gfx950, bf16, head_dim 128).
//
// Same items, sharing, work queue, Q / output staging and arithmetic per query row as attn_fwd_gqa.hip (one 8-wave workgroup serves a kv
// head's whole q-head group on a 32-query tile; reference call sites: qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:827-918,
// qwen-vl-finetune/qwenvl/train/trainer.py:101-113).  What differs is WHEN a wave does what.  In the sequential form a wave's tile is one
// dependent chain -- K fragments out of LDS, S^T = K Q^T, maxima, exponentials, V fragments, O^T += V^T P^T, barrier -- and measured alone
// on its SIMD a wave needs ~3 200 cycles per tile for 1 024 cycles of matrix work; two waves per SIMD in lockstep share it at ~4 600
// (profiles/EXPERIMENTS.md, round 5: knock-outs -- without ANY MFMA the kernel is 13 % faster, without the barrier 20 %).  Here, as in
// attn_fwd64.hip's generated loop but compiler-scheduled at two waves per SIMD:
//     phase X   S(t+1) = K(t+1) Q^T   (16 MFMAs)  beside  p = exp2(S(t) c - m), row sums, bf16 packing of tile t          (VALU)
//     phase Y   O^T  += V(t)^T P(t)^T (16 MFMAs)  beside  the (masked) row maxima of S(t+1)                                  (VALU)
//     then the reference exponent of tile t + 1 (rare rescale of O behind the P V MFMAs), one barrier.
// K is two tiles ahead (three ring slots: K(t+1) has been in LDS since the previous barrier), V one (two slots); the tile loads ride
// between the MFMAs of phase X.  An item's first S is computed alone (its Q fragments arrive with the item); every other S overlaps.
// The operations of a row and their order are those of the sequential forms: outputs and log-sum-exps are bit-identical (tests force all
// forms).  The 32-key skip of attn_fwd_gqa.hip (second half of a diagonal tile) is NOT taken here: the masked block contributes exact zeros.
#include "attn_common.h"
#include <atomic>

#include <algorithm>
#include <type_traits>

namespace vsel {

using namespace attn;

namespace gqap {
constexpr int kBuf = kTileBytes;                 // 16 KiB per K or V tile
constexpr int kKSlots = 3, kVSlots = 2;
constexpr int kV0 = kKSlots * kBuf;              // K[3] at 0, V[2] behind
constexpr int kQRegion = 32 * kRowBytes;         // 8 KiB: one wave's 32 query rows (also its output staging)
constexpr int kQ0 = (kKSlots + kVSlots) * kBuf;
constexpr int kCtl = kQ0 + 8 * kQRegion;
constexpr int kLds = kCtl + 16;

__device__ __forceinline__ bf16x8_t b8(u32x4 v) { return __builtin_bit_cast(bf16x8_t, v); }
template <int K>
__device__ __forceinline__ uint32_t xor_imm(uint32_t x) {
  uint32_t r;
  asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "n"(K), "v"(x));
  return r;
}
__device__ __forceinline__ int opaque_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ void both_halves(float x, float& a, float& b) {      // (attn_fwd_gqa.hip)
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
struct Item {
  int seq, kvh, q0, qs, qlen, n_tiles;           // n_tiles == 0: not an item
};
}  // namespace gqap

__device__ int g_gqap_work_counter[64];

__global__ __launch_bounds__(512, 2) void attn_fwd_gqap_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                               const uint16_t* __restrict__ v, const int32_t* __restrict__ cu, int hq, int hkv,
                                                               float scale_log2e, int causal, uint16_t* __restrict__ out, int q_tiles, int n_seq,
                                                               int* __restrict__ counter, PagedKV pg, float* __restrict__ lse) {
  using namespace gqap;
  constexpr int kSteps = 8, kDTiles = 4, kHeadDim = 128;
  __shared__ __attribute__((aligned(1024))) char smem[kLds];      // ONE __shared__ object (attn.hip)
  int* const s_cand = reinterpret_cast<int*>(smem + kCtl);
  int& s_slow = *reinterpret_cast<int*>(smem + kCtl + 8);
  const int rep = hq / hkv;
  const int QW = 8 / rep;
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
  const bool head_wave = wave < QW * rep;
  const int qw = head_wave ? wave / rep : 0;
  const int hl = head_wave ? wave % rep : 0;
  const int j = lane & 31, hh = lane >> 5;

  // per-lane LDS addresses.  ra[st]: K row fragment of k-step st in the K slot whose S comes next; ta[dt][hi]: V transposed fragment in the
  // V slot whose P V comes next.  Both sets are advanced in place when a tile has been consumed (the reads carry immediates only).
  // (attn_common.h: row_addr[st] = row_addr[0] ^ (st << 5); tr_addr[dt][0] = tr_addr[0][0] ^ (dt << 6), tr_addr[dt][1] = (tr_addr[dt][0] ^ 16) +
  // 1024 -- for offsets inside a tile at a 1 KiB-aligned base: ONE register per operand kind, one or two VALU per read, and 14 registers
  // back for the two score tiles this form keeps)
  uint32_t ra0, ta0;
  {
    int row_addr[kSteps], tr_addr[kDTiles][2];
    make_row_addr<kSteps>(row_addr, j, hh);
    make_tr_addr<kDTiles>(tr_addr, lane);
    ra0 = lds_u32(smem) + row_addr[0];
    ta0 = lds_u32(smem) + kV0 + tr_addr[0][0];
  }
  int ks_read = 0, vs_read = 0;                    // slots ra / ta point at
  auto advance_k = [&]() {                         // uniform delta: +1 slot, or back to slot 0
    const uint32_t d = ks_read == kKSlots - 1 ? (uint32_t)(-(kKSlots - 1) * kBuf) : (uint32_t)kBuf;
    ks_read = ks_read == kKSlots - 1 ? 0 : ks_read + 1;
    // (asm: hipcc otherwise keeps one precomputed address set per slot -- 24 + 16 registers -- and selects between them)
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(ra0) : "s"(d));
  };
  auto advance_v = [&]() {
    const uint32_t d = vs_read == 1 ? (uint32_t)(-kBuf) : (uint32_t)kBuf;
    vs_read ^= 1;
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(ta0) : "s"(d));
  };
  const uint32_t k_rs_b = (uint32_t)(kv_rs * 2), v_rs_b = (uint32_t)(v_rs * 2), q_rs_b = (uint32_t)(q_rs * 2);

  auto decode = [&](int item) -> Item {
    Item it{0, 0, 0, 0, 0, 0};
    if (item >= n_items) return it;
    const int level = item / n_pairs, pair = item - level * n_pairs;
    it.seq = pair / hkv;
    it.kvh = pair - it.seq * hkv;
    it.qs = cu[it.seq];
    it.qlen = cu[it.seq + 1] - it.qs;
    it.q0 = (q_tiles - 1 - level) * kBlockQ;
    if (it.q0 >= it.qlen) return it;
    const int kv_end = causal ? min(it.qlen, it.q0 + kBlockQ) : it.qlen;
    it.n_tiles = (kv_end + kTileK - 1) / kTileK;
    return it;
  };
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // this wave's two 1-KiB slices (rows 4 w .. and 4 (w + 8) ..) of tile t of K (which = 0) or V (which = 1) of `it` -> ring slot `slot`;
  // rows past the end of the sequence replay its last row (finite, masked later)
  auto load_kv = [&](const Item& it, int t, int which, int slot) {
    const char* base = reinterpret_cast<const char*>((which ? v : k) + (int64_t)it.qs * hkv * kHeadDim + it.kvh * (which ? v_hs : kv_hs));
    const uint32_t rs_b = which ? v_rs_b : k_rs_b;
    char* const dst = smem + (which ? kV0 : 0) + slot * kBuf;
    const int l = opaque_lane();
    const uint32_t part_b = (uint32_t)slice_src_part(l, wave) * 16u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = wave + 8 * u;
      const uint32_t row = (uint32_t)min(t * kTileK + 4 * i + (l >> 4), it.qlen - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(base + (row * rs_b + part_b)), (lptr_t)(dst + i * 1024), 16, 0, 0);
    }
  };
  auto load_q = [&](const Item& it) {
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
  auto validate = [&](int cand) -> Item {
    for (;;) {
      Item it = decode(cand);
      if (it.n_tiles > 0 || cand >= n_items) return it;
      const int level = cand / n_pairs, pair = cand - level * n_pairs;
      if (pair % hkv == 0)
        queue_skip_empty_run(counter, tid, cu, n_seq, hkv, level, pair / hkv,
                             [&](int lv, int ql) { return (q_tiles - 1 - lv) * kBlockQ < ql; });
      if (tid == 0) s_slow = atomicAdd(counter, 1);
      __syncthreads();
      cand = __builtin_amdgcn_readfirstlane(s_slow);
      __syncthreads();
    }
  };

  // ---- prologue ---------------------------------------------------------------------------------------------------------------------
  Item cur = validate((int)blockIdx.x);
  if (cur.n_tiles == 0) return;
  int gk = 0;                                       // global index of the current tile: K slot gk % 3, V slot gk % 2
  int ks_cur = 0, vs_cur = 0;                       // = gk % 3, gk % 2
  auto k_slot = [&](int ahead) { const int s = ks_cur + ahead; return s >= kKSlots ? s - kKSlots : s; };
  load_q(cur);
  load_kv(cur, 0, 0, 0);
  load_kv(cur, 0, 1, 0);
  if (cur.n_tiles > 1) load_kv(cur, 1, 0, 1);
  if (tid == 0) s_slow = atomicAdd(counter, 1);
  __syncthreads();
  int cand = __builtin_amdgcn_readfirstlane(s_slow);
  __syncthreads();
  Item next = validate(cand);
  Item next2{0, 0, 0, 0, 0, 0};
  int pend = 0;
  if (tid == 0) pend = atomicAdd(counter, 1);
  int ipar = 0;
  int t = 0;

  // per-item state of the wave
  u32x4 qf[kSteps];
  f32x16 o[kDTiles];
  f32x16 sc[2][2];                                  // sc[p]: the score tile of parity p, two 32-key blocks
  float m_run = -1e30f, l_run = 0.f, mx = -INFINITY;
  int wave_qmin = cur.q0 + qw * 32;
  bool wave_has_rows = head_wave && wave_qmin < cur.qlen;
  int my_q = min(wave_qmin + (opaque_lane() & 31), cur.qlen - 1);
  auto read_q = [&]() {
    const int rl = opaque_lane();                   // (nothing lane-derived is kept across the tile loop)
    const uint32_t q_addr0 = lds_u32(smem) + kQ0 + wave * kQRegion + chunk_off(rl & 31, rl >> 5);
    static_for<0, kSteps>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      qf[st] = lds_read_b128_asm<0>(xor_imm<32 * st>(q_addr0));
    });
    lds_wait8<0>(qf);
  };
  auto zero_o = [&]() {
#pragma unroll
    for (int dt = 0; dt < kDTiles; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  };
  // mask of the tile tt held in `s` (both blocks) + running maximum into mx
  auto mask_max = [&](f32x16 (&s)[2], int tt) {
    const int len = cur.qlen;
    const int kmax = causal ? min(len - 1, my_q) : len - 1;
    // key = tile base + 8 hh + (compile-time offset of the element) <= kmax  <=>  offset <= rel: one register, immediates in the compares
    // (with the keys themselves the compiler keeps 32 loop-invariant "8 hh | offset" values in registers -- and spills them)
    const int rel = kmax - (tt * kTileK + 8 * (opaque_lane() >> 5));
    mx = -INFINITY;
    static_for<0, 2>([&](auto kb_c) {
      constexpr int KB = decltype(kb_c)::value;
      const bool need_mask = __builtin_amdgcn_readfirstlane(
          (int)((tt * kTileK + 32 * KB + 32 > len) || (causal && (tt * kTileK + 32 * KB + 31 > wave_qmin)))) != 0;
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float val = (32 * KB + 16 * (r >> 3) + (r & 7)) <= rel ? s[KB][r] : -INFINITY;
          s[KB][r] = val;
          mx = fmaxf(mx, val);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[KB][r]);
      }
    });
  };
  // the reference exponent for the tile whose maximum is in mx (attn.hip's lazy rule), rare rescale of l and O
  auto decide = [&]() {
    float ha, hb;
    both_halves(mx, ha, hb);
    const float mrow = fmaxf(ha, hb);
    const float m_cand = fmaxf(m_run, mrow * scale_log2e);
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
  };
  // S of the tile in the K slot ra points at -> s, alone (an item's first tile)
  auto s_alone = [&](f32x16 (&s)[2]) {
    static_for<0, 2>([&](auto kb_c) {
      constexpr int KB = decltype(kb_c)::value;
      u32x4 ka[8];
      static_for<0, 8>([&](auto st_c) {
        constexpr int st = decltype(st_c)::value;
        ka[st] = lds_read_b128_asm<KB * 32 * kRowBytes>(xor_imm<32 * st>(ra0));
      });
      lds_wait8<0>(ka);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int st = 0; st < 8; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8(ka[st]), b8(qf[st]), acc, 0, 0, 0);
      s[KB] = acc;
    });
  };
  if (wave_has_rows) {
    read_q();
    s_alone(sc[0]);
    mask_max(sc[0], 0);
    zero_o();
    decide();
  } else {
    zero_o();
  }
  advance_k();                                      // ra -> the slot of tile 1

  // ---- epilogue (attn_fwd_gqa.hip): O^T / l through the wave's staging area as whole rows -------------------------------------------------
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
#pragma unroll
    for (int dt = 0; dt < kDTiles; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4_t pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(o[dt][4 * g4 + e] * inv);
        *reinterpret_cast<bf16x4_t*>(stage + (wa ^ (uint32_t)((4 * dt + g4) << 4))) = pk;
      }
    char* const ob = reinterpret_cast<char*>(out + ((int64_t)(cur.qs + wave_qmin) * hq + head) * kHeadDim);
    const uint32_t ostride = (uint32_t)hq * kHeadDim * 2;
    const int rows = cur.qlen - wave_qmin;
    u32x4 rowv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowv[i] = *reinterpret_cast<const u32x4*>(stage + i * 1024 + el * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 4 * i + e_l4;
      const uint32_t c = (uint32_t)(e_p ^ (r & 15));
      if (r < rows) *reinterpret_cast<u32x4*>(ob + (uint32_t)r * ostride + c * 16u) = rowv[i];
    }
  };

  // ---- one round = tile t of the current item: its exponentials and P V, beside S and the maxima of tile t + 1 ----------------------------
  auto round = [&](auto par_c) -> bool {
    constexpr int PAR = decltype(par_c)::value;
    f32x16 (&s_cur)[2] = sc[PAR];
    f32x16 (&s_nxt)[2] = sc[1 - PAR];
    const bool last = t + 1 == cur.n_tiles;
    const bool first = t == 0;
    // the direct-to-LDS loads of this round, in pieces (one call = this wave's two slices of one tile, or its 8 Q slices):
    //   not last: K(t + 2) [piece 0, if it exists; else the next item's K(0) when t + 2 == n], V(t + 1) [piece 1]
    //   last    : next item's Q [piece 0], its K(0) when this item has a single tile [piece 0], its K(1) [piece 1], its V(0) [piece 1]
    auto piece = [&](int p) {
      const bool have_next = next.n_tiles > 0;
      if (!last) {
        if (p == 0) {
          if (t + 2 < cur.n_tiles) load_kv(cur, t + 2, 0, k_slot(2));
          else if (have_next) load_kv(next, 0, 0, k_slot(2));
        } else {
          load_kv(cur, t + 1, 1, vs_cur ^ 1);
        }
      } else if (have_next) {
        if (p == 0) {
          load_q(next);
          if (cur.n_tiles == 1) load_kv(next, 0, 0, k_slot(1));
        } else {
          if (next.n_tiles > 1) load_kv(next, 1, 0, k_slot(2));
          load_kv(next, 0, 1, vs_cur ^ 1);
        }
      }
    };
    bf16x8_t pf[2][2];
    if (wave_has_rows) {
      float psum = 0.f;
      // four elements of tile t: p = exp2(s c - m), row sum in order, bf16 packing (chunk c: block c >> 2, elements 4 (c & 3) ..)
      auto probs_chunk = [&](auto c_c) {
        constexpr int C = decltype(c_c)::value;
        constexpr int KB = C >> 2, R0 = 4 * (C & 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s_cur[KB][R0 + e], scale_log2e, -m_run));
          psum += p;
          pf[KB][(R0 + e) >> 3][(R0 + e) & 7] = (__bf16)p;
        }
      };
      if (!last) {
        // ---- phase X: S(t + 1) beside the exponentials of tile t's FIRST block (the second block's run beside P V below: the two
        // phases carry the same VALU load, and its eight P registers are not live while S(t + 1) needs its sixteen) ----------------------
        u32x4 ka[8];
        static_for<0, 8>([&](auto st_c) {
          constexpr int st = decltype(st_c)::value;
          ka[st] = lds_read_b128_asm<0>(xor_imm<32 * st>(ra0));
        });
        probs_chunk(std::integral_constant<int, 0>{});
        lds_wait8<0>(ka);
#pragma unroll
        for (int r = 0; r < 16; ++r) s_nxt[0][r] = 0.f;
        static_for<0, 4>([&](auto u_c) {
          constexpr int U = decltype(u_c)::value;
          s_nxt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8(ka[2 * U]), b8(qf[2 * U]), s_nxt[0], 0, 0, 0);
          s_nxt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8(ka[2 * U + 1]), b8(qf[2 * U + 1]), s_nxt[0], 0, 0, 0);
          if constexpr (U == 1) probs_chunk(std::integral_constant<int, 1>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        piece(0);                                    // (control flow: never between an asm-issued read and its wait)
        static_for<0, 8>([&](auto st_c) {
          constexpr int st = decltype(st_c)::value;
          ka[st] = lds_read_b128_asm<32 * kRowBytes>(xor_imm<32 * st>(ra0));
        });
        probs_chunk(std::integral_constant<int, 2>{});
        lds_wait8<0>(ka);
#pragma unroll
        for (int r = 0; r < 16; ++r) s_nxt[1][r] = 0.f;
        static_for<0, 4>([&](auto u_c) {
          constexpr int U = decltype(u_c)::value;
          s_nxt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8(ka[2 * U]), b8(qf[2 * U]), s_nxt[1], 0, 0, 0);
          s_nxt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8(ka[2 * U + 1]), b8(qf[2 * U + 1]), s_nxt[1], 0, 0, 0);
          if constexpr (U == 1) probs_chunk(std::integral_constant<int, 3>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        piece(1);
      } else {
        piece(0);
        static_for<0, 4>([&](auto c_c) { probs_chunk(c_c); });
        piece(1);
      }
      // ---- phase Y: O^T += V(t)^T P(t)^T beside the second block's exponentials and the masked maxima of S(t + 1) ------------------------
      u32x2 vr0[8], vr1[8];
      auto issue = [&](auto g_c, u32x2 (&dst)[8]) {
        constexpr int G = decltype(g_c)::value;
        constexpr int OFF = (32 * (G >> 1) + 16 * (G & 1)) * kRowBytes;
        static_for<0, 4>([&](auto dt_c) {
          constexpr int dt = decltype(dt_c)::value;
          const uint32_t a0 = xor_imm<64 * dt>(ta0);
          dst[2 * dt] = lds_read_tr16_b64_asm<OFF>(a0);
          dst[2 * dt + 1] = lds_read_tr16_b64_asm<OFF + 1024>(xor_imm<16>(a0));
        });
      };
      auto pv = [&](auto g_c, u32x2 (&src)[8]) {
        constexpr int G = decltype(g_c)::value;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const u32x4 w = {src[2 * dt][0], src[2 * dt][1], src[2 * dt + 1][0], src[2 * dt + 1][1]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8(w), pf[G >> 1][G & 1], o[dt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      issue(I0{}, vr0);
      issue(I1{}, vr1);
      probs_chunk(std::integral_constant<int, 4>{});
      lds_wait8<8>(vr0);
      pv(I0{}, vr0);
      probs_chunk(std::integral_constant<int, 5>{});
      issue(I2{}, vr0);
      lds_wait8<8>(vr1);
      pv(I1{}, vr1);
      probs_chunk(std::integral_constant<int, 6>{});
      probs_chunk(std::integral_constant<int, 7>{});
      l_run += psum;
      issue(I3{}, vr1);
      lds_wait8<8>(vr0);
      pv(I2{}, vr0);
      lds_wait8<0>(vr1);
      pv(I3{}, vr1);
      if (!last) {
        mask_max(s_nxt, t + 1);
        decide();
      }
    } else {
      piece(0);
      piece(1);
    }
    if (first && tid == 0) s_cand[ipar] = pend;
    __syncthreads();                                 // vmcnt(0) in front of it: every wave's direct loads have landed at the release
    if (first) {
      cand = __builtin_amdgcn_readfirstlane(s_cand[ipar]);
      ipar ^= 1;
      next2 = validate(cand);
    }
    // the tile is consumed: V read slot moves on; the K read slot moved when S(t + 1) was computed (or moves now, past the next item's K(0))
    advance_v();
    vs_cur ^= 1;
    ks_cur = k_slot(1);
    ++gk;
    if (!last) {
      advance_k();
      ++t;
      return true;
    }
    // ---- the item is complete ------------------------------------------------------------------------------------------------------------
    const bool had_rows = wave_has_rows;
    const int next_qmin = next.q0 + qw * 32;
    const bool next_has_rows = head_wave && next.n_tiles > 0 && next_qmin < next.qlen;
    if (next_has_rows) read_q();
    if (had_rows) epilogue();
    if (next.n_tiles == 0) return false;
    zero_o();
    m_run = -1e30f;
    l_run = 0.f;
    cur = next;
    t = 0;
    wave_qmin = next_qmin;
    wave_has_rows = next_has_rows;
    my_q = min(wave_qmin + (opaque_lane() & 31), cur.qlen - 1);
    next = next2;
    if (tid == 0) pend = atomicAdd(counter, 1);
    // the new item's first S, alone: its K(0) and Q fragments are in LDS since the barrier above
    if (wave_has_rows) {
      s_alone(s_nxt);
      mask_max(s_nxt, 0);
      decide();
    }
    advance_k();
    return true;
  };
  for (;;) {
    if (!round(std::integral_constant<int, 0>{})) return;
    if (!round(std::integral_constant<int, 1>{})) return;
  }
}

namespace attn {

int attn_fwd_gqap_launch(hipStream_t st, const void* q, const void* k, const void* v, const int32_t* cu_q, int64_t n_seq,
                         int64_t max_seqlen_q, int64_t hq, int64_t hkv, float scale, int causal, void* out, const PagedKV& pg, float* lse) {
  const int rep = (int)(hq / hkv);
  const int block_q = 32 * (8 / rep);
  const int64_t q_tiles = cdiv(max_seqlen_q, block_q);
  const int64_t n_items = q_tiles * hkv * n_seq;
  if (n_items >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "too many attention work items");
  int slot = -1;
  if (int rc = queue_slot_acquire(kSlotGqa, st, &slot)) return rc;
  int* counters = nullptr;
  VSEL_HIP_CHECK(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_gqap_work_counter)));
  const int grid = (int)std::min<int64_t>(n_items, 256);
  VSEL_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(counters + slot), grid, 1, st));
  hipLaunchKernelGGL(attn_fwd_gqap_kernel, dim3(grid), dim3(512), 0, st, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, cu_q,
                     (int)hq, (int)hkv, scale * 1.4426950408889634f, causal, (uint16_t*)out, (int)q_tiles, (int)n_seq, counters + slot, pg, lse);
  queue_slot_launched(kSlotGqa, slot, st);
  VSEL_AFTER_LAUNCH(st, "attn_fwd_gqap_kernel");
  return VSEL_OK;
}

}  // namespace attn
}  // namespace vsel
