// Var-len causal attention forward for ONE long sequence (or a few of equal length): attn_fwd64's items cut over KEY RANGES.
//
// attn_fwd64_kernel's item is (256-query tile, q head); under the causal mask the tile at the end of a 2368-token sequence walks 37
// key tiles, one after the other, while 280 items share 256 CUs: the launch runs as long as that one chain (58.5 us per layer) although
// the work, spread evenly, is 43 us (profiles/EXPERIMENTS.md, round 5).  Here a tile with more than `cap` key tiles is cut into P parts
// of consecutive key tiles; a part is an item of its own that runs the SAME generated loop (tools/gen_attn_fwd64.py, option raw = 1) on
// its keys -- K / V base moved to the part's first key, lengths and mask bounds taken relative to it -- and, instead of normalising, leaves
// its fp32 accumulators and (m, l) in the caller's workspace; attn_fwd64_merge_kernel adds the parts of a row in part order:
//     M = max m_p,  O = sum_p 2^(m_p - M) O_p,  L = sum_p 2^(m_p - M) l_p,  out = bf16(O / L),  lse = (M + log2 L) ln 2.
// Part boundaries are multiples of 64 keys at or below the tile's first query, so every part but the last sees all of its keys (no mask)
// and every row sees at least one key of the last part.  Tiles with P = 1 write their output directly, as attn_fwd64_kernel does.
// Deterministic (fixed part order); not bit-identical to the unsplit forms (another fp32 association: within a bf16 rounding, the usual
// 1-ulp oracle gate applies).  Reference call sites: as attn_fwd64.hip.  Uniform batches only (every sequence max_seqlen tokens):
// the item list is a host-side table per tile level, dealt out with mirrored rounds.
#include "attn_common.h"
#include "attn_fwd64_parts_body.inc"

#include <algorithm>

namespace vsel {

using namespace attn;

namespace {
__device__ __forceinline__ const uint16_t* uniform_ptr_p(const uint16_t* p) {
  const uint64_t u = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (const uint16_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
constexpr int kLdsP = VSEL_FWD64P_LDS_BYTES;
constexpr int kBlockQP = 256;
constexpr int kHeadDimP = 128;
constexpr int kMaxLevels = 64;
}  // namespace

// parts of a tile with n_tiles key tiles (host and device): as few as keep a part at <= cap tiles, the last part not shorter than the four
// diagonal tiles of a 256-query tile
__host__ __device__ inline int parts_of(int n_tiles, int cap) {
  int p = (n_tiles + cap - 1) / cap;
  while (p > 1) {
    const int tpp = (n_tiles + p - 1) / p;
    if (n_tiles - (p - 1) * tpp >= 4) break;
    --p;
  }
  return p < 1 ? 1 : p;
}

__global__ __launch_bounds__(256, 1) void attn_fwd64_parts_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                                  const uint16_t* __restrict__ v, int len, int hq, int hkv,
                                                                  float scale_log2e, uint16_t* __restrict__ out, int n_seq,
                                                                  PartsPlan plan, float* __restrict__ opart, float* __restrict__ ml,
                                                                  float* __restrict__ lse) {
  __shared__ __attribute__((aligned(1024))) char smem[kLdsP];
  const int rep = hq / hkv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int lds_base = (int)lds_u32(smem);
  const int64_t row_elems = (int64_t)hq * kHeadDimP, kv_row = (int64_t)hkv * kHeadDimP;
  const int64_t total = (int64_t)n_seq * len;
  for (int round = 0;; ++round) {
    const int item = __builtin_amdgcn_readfirstlane(static_deal_item(round));
    if (item >= plan.n_items) return;
    int level = 0;
    while (level + 1 < plan.q_tiles && item >= plan.level_off[level + 1]) ++level;
    const int r = item - plan.level_off[level];
    const int q0 = (plan.q_tiles - 1 - level) * kBlockQP;
    const int kv_end = min(len, q0 + kBlockQP);
    const int n_tiles_all = (kv_end + kTileK - 1) / kTileK;
    const int n_parts = parts_of(n_tiles_all, plan.cap);
    const int part = r / (hq * n_seq), rest = r - part * (hq * n_seq);
    const int seq = rest / hq, head = rest - seq * hq;
    const int kvh = head / rep;
    const int qs = seq * len;
    const int tpp = (n_tiles_all + n_parts - 1) / n_parts;
    const int t_lo = part * tpp, t_hi = min(n_tiles_all, t_lo + tpp);
    const bool last = part == n_parts - 1;
    const int k_lo = t_lo * kTileK;                                       // (<= q0 for every part: parts_of)
    const int len_p = (last ? kv_end : t_hi * kTileK) - k_lo;             // keys of this part
    const int n_tiles = t_hi - t_lo;

    const int wave_qmin = q0 + 64 * wave;
    const int wave_qmax = min(wave_qmin + 63, len - 1);
    const int vq_a = wave_qmin + j, vq_b = wave_qmin + 32 + j;
    const int my_qa = min(vq_a, len - 1), my_qb = min(vq_b, len - 1);
    const int valid = (vq_a < len ? 1 : 0) | (vq_b < len ? 2 : 0);
    // tiles this wave computes / first tile that needs the mask, relative to the part's first key (attn_fwd64.hip with shift = -k_lo)
    int n_w = wave_qmin < len ? n_tiles : 0;
    int mfirst = len_p / kTileK;
    int kmax_a = len_p - 1, kmax_b = len_p - 1;
    if (last) {
      if (n_w > 0) n_w = min(n_tiles, (wave_qmax - k_lo) / kTileK + 1);
      const int x = wave_qmin - k_lo;
      mfirst = min(mfirst, x / kTileK + ((x % kTileK) != kTileK - 1 ? 0 : 1));
      kmax_a = min(len_p - 1, my_qa - k_lo);
      kmax_b = min(len_p - 1, my_qb - k_lo);
    }
    n_w = __builtin_amdgcn_readfirstlane(n_w);
    mfirst = __builtin_amdgcn_readfirstlane(mfirst);

    const uint16_t* const qbase = uniform_ptr_p(q + ((int64_t)(qs + wave_qmin) * hq + head) * kHeadDimP);
    const uint16_t* const kbase = uniform_ptr_p(k + ((int64_t)(qs + k_lo) * hkv + kvh) * kHeadDimP);
    const uint16_t* const vbase = uniform_ptr_p(v + ((int64_t)(qs + k_lo) * hkv + kvh) * kHeadDimP);
    const uint16_t* const obase = uniform_ptr_p(out + ((int64_t)(qs + wave_qmin) * hq + head) * kHeadDimP);
    const int qrs2 = (int)(row_elems * 2), krs2 = (int)(kv_row * 2);
    const int ostride = hq * kHeadDimP * 2;
    const int nvalid = __builtin_amdgcn_readfirstlane(len - wave_qmin);
    const int len_u = __builtin_amdgcn_readfirstlane(len_p), ntiles_u = __builtin_amdgcn_readfirstlane(n_tiles);
    const int raw = __builtin_amdgcn_readfirstlane(n_parts > 1 ? 1 : 0);
    float* const prawa = opart + (((int64_t)part * total + qs + my_qa) * hq + head) * kHeadDimP + 4 * hh;
    float* const prawb = opart + (((int64_t)part * total + qs + my_qb) * hq + head) * kHeadDimP + 4 * hh;
    float m0, m1, l0, l1;
    asm volatile(VSEL_FWD64P_ASM_TEXT
                 : [m0] "=&v"(m0), [m1] "=&v"(m1), [l0] "=&v"(l0), [l1] "=&v"(l1)
                 : [qbase] "s"(qbase), [qrs2] "s"(qrs2), [obase] "s"(obase), [ostride] "s"(ostride), [nvalid] "s"(nvalid), [kbase] "s"(kbase),
                   [vbase] "s"(vbase), [krs2] "s"(krs2), [vrs2] "s"(krs2), [ntiles] "s"(ntiles_u), [nw] "s"(n_w), [mfirst] "s"(mfirst),
                   [len] "s"(len_u), [c] "s"(scale_log2e), [wave] "s"(wave), [ldsbase] "s"(lds_base), [kmaxa] "v"(kmax_a), [kmaxb] "v"(kmax_b),
                   [raw] "s"(raw), [prawa] "v"(prawa), [prawb] "v"(prawb)
                 : VSEL_FWD64P_ASM_CLOBBERS);
    const float lt0 = l0 + __shfl_xor(l0, 32, 64), lt1 = l1 + __shfl_xor(l1, 32, 64);
    if (raw) {
      if (hh == 0) {
        if (valid & 1) *reinterpret_cast<float2*>(ml + (((int64_t)part * total + qs + my_qa) * hq + head) * 2) = make_float2(m0, lt0);
        if (valid & 2) *reinterpret_cast<float2*>(ml + (((int64_t)part * total + qs + my_qb) * hq + head) * 2) = make_float2(m1, lt1);
      }
    } else if (lse && hh == 0) {
      if (valid & 1) lse[(int64_t)(qs + my_qa) * hq + head] = lt0 > 0.f ? (m0 + log2f(lt0)) * 0.6931471805599453f : -INFINITY;
      if (valid & 2) lse[(int64_t)(qs + my_qb) * hq + head] = lt1 > 0.f ? (m1 + log2f(lt1)) * 0.6931471805599453f : -INFINITY;
    }
    __syncthreads();                   // the next item's first loads overwrite ring slots / staging rows other waves may still read
  }
}

// one wave per (row, head) of a split tile: lane = two features.  grid = split rows * hq / 4 workgroups of 4 waves.
__global__ __launch_bounds__(256) void attn_fwd64_merge_kernel(const float* __restrict__ opart, const float* __restrict__ ml, int len, int hq,
                                                               int n_seq, PartsPlan plan, int first_split_row, int split_rows,
                                                               uint16_t* __restrict__ out, float* __restrict__ lse) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t per_seq = (int64_t)split_rows * hq;
  if (w >= per_seq * n_seq) return;
  const int seq = (int)(w / per_seq);
  const int64_t rr = w - seq * per_seq;
  const int row = first_split_row + (int)(rr / hq), head = (int)(rr % hq);
  const int level = plan.q_tiles - 1 - row / kBlockQP;
  const int kv_end = min(len, (row / kBlockQP) * kBlockQP + kBlockQP);
  const int n_parts = parts_of((kv_end + kTileK - 1) / kTileK, plan.cap);
  (void)level;
  if (n_parts < 2) return;                 // (written directly by the parts kernel)
  const int64_t total = (int64_t)n_seq * len;
  const int64_t grow = (int64_t)seq * len + row;
  float mx = -INFINITY;
  for (int p = 0; p < n_parts; ++p) mx = fmaxf(mx, ml[(((int64_t)p * total + grow) * hq + head) * 2]);
  float o0 = 0.f, o1 = 0.f, lsum = 0.f;
  for (int p = 0; p < n_parts; ++p) {
    const int64_t base = ((int64_t)p * total + grow) * hq + head;
    const float2 mlp = *reinterpret_cast<const float2*>(ml + base * 2);
    const float wgt = exp2f(mlp.x - mx);
    const float2 op = *reinterpret_cast<const float2*>(opart + base * kHeadDimP + 2 * lane);
    o0 = fmaf(wgt, op.x, o0);
    o1 = fmaf(wgt, op.y, o1);
    lsum = fmaf(wgt, mlp.y, lsum);
  }
  const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
  const uint32_t pk = f32_to_bf16_bits(o0 * inv) | (f32_to_bf16_bits(o1 * inv) << 16);
  *reinterpret_cast<uint32_t*>(out + (grow * hq + head) * kHeadDimP + 2 * lane) = pk;
  if (lse && lane == 0) lse[grow * hq + head] = lsum > 0.f ? (mx + log2f(lsum)) * 0.6931471805599453f : -INFINITY;
}

namespace attn {

// plan for a uniform batch of n_seq sequences of `len` tokens; cap = key tiles per part.  false: nothing to split
bool fwd64_parts_plan(int64_t n_seq, int64_t len, int64_t hq, int cap, PartsPlan* plan) {
  const int q_tiles = (int)cdiv(len, kBlockQP);
  if (q_tiles > kMaxLevels) return false;
  plan->q_tiles = q_tiles;
  plan->cap = cap;
  plan->max_parts = 1;
  int off = 0;
  for (int l = 0; l < q_tiles; ++l) {
    plan->level_off[l] = off;
    const int q0 = (q_tiles - 1 - l) * kBlockQP;
    const int kv_end = (int)std::min<int64_t>(len, q0 + kBlockQP);
    const int p = parts_of((kv_end + kTileK - 1) / kTileK, cap);
    plan->max_parts = std::max(plan->max_parts, p);
    off += p * (int)(hq * n_seq);
  }
  plan->level_off[q_tiles] = off;
  for (int l = q_tiles + 1; l <= kMaxLevels; ++l) plan->level_off[l] = off;
  plan->n_items = off;
  return plan->max_parts > 1;
}

size_t fwd64_parts_workspace_bytes(int64_t n_seq, int64_t len, int64_t hq, int max_parts) {
  return (size_t)max_parts * (size_t)(n_seq * len) * (size_t)hq * (kHeadDimP + 2) * sizeof(float);
}

int attn_fwd64_parts_launch(hipStream_t st, const void* q, const void* k, const void* v, int64_t n_seq, int64_t len, int64_t hq, int64_t hkv,
                            float scale, void* out, float* lse, const PartsPlan& plan, void* ws, size_t ws_bytes) {
  if (ws_bytes < fwd64_parts_workspace_bytes(n_seq, len, hq, plan.max_parts))
    return fail(VSEL_ERR_WORKSPACE, "attention workspace %zu B < required %zu B", ws_bytes, fwd64_parts_workspace_bytes(n_seq, len, hq, plan.max_parts));
  float* opart = (float*)ws;
  float* ml = opart + (size_t)plan.max_parts * (size_t)(n_seq * len) * (size_t)hq * kHeadDimP;
  const dim3 grid((unsigned)std::min(plan.n_items, 256));
  VSEL_LAUNCH(attn_fwd64_parts_kernel, grid, dim3(256), 0, st, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (int)len, (int)hq,
                     (int)hkv, scale * 1.4426950408889634f, (uint16_t*)out, (int)n_seq, plan, opart, ml, lse);
  VSEL_AFTER_LAUNCH(st, "attn_fwd64_parts_kernel");
  // rows of the split tiles: the LAST query tiles (levels 0 .. s - 1 are split, parts_of is monotone in the tile's key count)
  int split_levels = 0;
  for (int l = 0; l < plan.q_tiles; ++l) {
    const int q0 = (plan.q_tiles - 1 - l) * kBlockQP;
    const int kv_end = (int)std::min<int64_t>(len, q0 + kBlockQP);
    if (parts_of((kv_end + kTileK - 1) / kTileK, plan.cap) > 1) split_levels = l + 1;
  }
  const int first_split_row = (plan.q_tiles - split_levels) * kBlockQP;
  const int split_rows = (int)len - first_split_row;
  const int64_t waves = (int64_t)split_rows * hq * n_seq;
  VSEL_LAUNCH(attn_fwd64_merge_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, opart, ml, (int)len, (int)hq, (int)n_seq, plan,
                     first_split_row, split_rows, (uint16_t*)out, lse);
  VSEL_AFTER_LAUNCH(st, "attn_fwd64_merge_kernel");
  return VSEL_OK;
}

}  // namespace attn
}  // namespace vsel
