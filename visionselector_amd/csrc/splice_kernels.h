// Device side of the sequence splice (see splice.hip for the reference lines it replaces): the ordered scan that turns
// (input_ids, kept visual ranks) into the index arrays of the compressed sequence and a per-row source descriptor, and the
// row copy that materialises inputs_embeds'.  Shared by splice.hip (vsel_splice, vsel_splice_batched) and select_splice.hip
// (vsel_lis_select_splice: kept rows are read straight from the token tensor H, never staged in a [k, D] tensor).
#pragma once
#include "common.h"

#include <algorithm>
#include <limits.h>

namespace vsel {

constexpr int kSpliceThreads = 1024;
constexpr int kMaxVisualBits = 1 << 18;   // 262 144 visual tokens -> 32 KiB bitmap

// One sequence's slice of the packed arrays: positions [p0, p0+L), its visual tokens hold local ranks [0, nvis), its kept
// visual rows are all_indices[j0 .. j0+k) (local ranks, ascending) and its output rows start at q0.
struct SpliceSeq {
  int p0, L, nvis, j0, k, q0;
};

// Ordered scan of one sequence by one 1024-thread workgroup.  Returns (visual found, rows kept, visual kept) to thread 0.
__device__ __forceinline__ void splice_index_body(
    const SpliceSeq sq, uint32_t* bitmap, const int64_t* __restrict__ ids, int64_t visual_id,
    const int64_t* __restrict__ all_indices, const int64_t* __restrict__ pos, int pos_rows, int64_t pos_stride,
    const int64_t* __restrict__ mask, int64_t* __restrict__ sel, int64_t* __restrict__ new_ids,
    int64_t* __restrict__ new_pos, int64_t* __restrict__ new_mask, int32_t* __restrict__ src, int l_out, int q_end,
    uint32_t& out_vis, uint32_t& out_keep, uint32_t& out_kv, int64_t vis_row_base = -1,
    const int64_t* __restrict__ src_map = nullptr) {
  // vis_row_base >= 0: the kept visual rows are read from the token tensor H itself -- the descriptor of a kept visual token of
  // local rank r is -(physical row of H) - 1 with physical row = src_map ? src_map[vis_row_base + r] : vis_row_base + r
  // (vsel_lis_select_splice); otherwise -(j0 + j) - 1 = its row in the staged [k, D] tensor (vsel_splice).
  __shared__ uint32_t wv[16], wk[16], wj[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int words = (sq.nvis + 31) >> 5;
  for (int i = tid; i < words; i += kSpliceThreads) bitmap[i] = 0u;
  __syncthreads();
  for (int j = tid; j < sq.k; j += kSpliceThreads) {
    const int64_t r = all_indices[sq.j0 + j];
    if (r >= 0 && r < sq.nvis) atomicOr(&bitmap[r >> 5], 1u << (r & 31));
  }
  __syncthreads();
  uint32_t run_vis = 0, run_keep = 0, run_kv = 0;   // running counts: visual tokens, kept positions, kept visual tokens
  for (int c0 = 0; c0 < sq.L; c0 += kSpliceThreads) {
    const int p = c0 + tid;
    const bool valid = p < sq.L;
    const int64_t id = valid ? ids[sq.p0 + p] : 0;
    const bool is_vis = valid && id == visual_id;
    const unsigned long long bvis = __ballot(is_vis);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) wv[wave] = __popcll(bvis);
    __syncthreads();
    uint32_t vis_rank = run_vis + __popcll(bvis & below), tot_vis = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { if (w < wave) vis_rank += wv[w]; tot_vis += wv[w]; }
    const bool kept_vis = is_vis && vis_rank < (uint32_t)sq.nvis && ((bitmap[vis_rank >> 5] >> (vis_rank & 31)) & 1u);
    const bool keep = valid && (!is_vis || kept_vis);
    const unsigned long long bkeep = __ballot(keep), bkv = __ballot(kept_vis);
    if (lane == 0) { wk[wave] = __popcll(bkeep); wj[wave] = __popcll(bkv); }
    __syncthreads();
    uint32_t q = run_keep + __popcll(bkeep & below), j = run_kv + __popcll(bkv & below), tot_keep = 0, tot_kv = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) { q += wk[w]; j += wj[w]; }
      tot_keep += wk[w];
      tot_kv += wj[w];
    }
    const int qo = sq.q0 + (int)q;
    if (keep && qo < q_end) {
      sel[qo] = sq.p0 + p;
      new_ids[qo] = id;
      if (mask) new_mask[qo] = mask[sq.p0 + p];
      for (int r = 0; r < pos_rows; ++r) new_pos[(int64_t)r * l_out + qo] = pos[(int64_t)r * pos_stride + sq.p0 + p];
      if (kept_vis && vis_row_base >= 0) {
        const int64_t lrow = vis_row_base + vis_rank;
        src[qo] = -(int32_t)(src_map ? src_map[lrow] : lrow) - 1;
      } else {
        src[qo] = kept_vis ? -(int32_t)(sq.j0 + j + 1) : sq.p0 + p;
      }
    }
    run_vis += tot_vis;
    run_keep += tot_keep;
    run_kv += tot_kv;
    __syncthreads();
  }
  // rows the scan did not produce (input_ids holds fewer kept positions than the descriptors promise: the reference raises
  // ValueError, FT/compression_method/selector_model.py:210-213; here stats report it): mark them so that the embedding
  // copy never dereferences an uninitialised source row
  for (int q = sq.q0 + (int)run_keep + tid; q < q_end; q += kSpliceThreads) {
    src[q] = INT32_MAX;
    sel[q] = -1;
    new_ids[q] = -1;
  }
  out_vis = run_vis;
  out_keep = run_keep;
  out_kv = run_kv;
}

static __attribute__((unused)) __global__ __launch_bounds__(kSpliceThreads) void splice_index_kernel(
    const int64_t* __restrict__ ids, int L, int64_t visual_id, const int64_t* __restrict__ all_indices, int k, int n_visual,
    const int64_t* __restrict__ pos, int pos_rows, const int64_t* __restrict__ mask, int64_t* __restrict__ sel,
    int64_t* __restrict__ new_ids, int64_t* __restrict__ new_pos, int64_t* __restrict__ new_mask,
    int32_t* __restrict__ src, int32_t* __restrict__ stats, int l_out) {
  extern __shared__ uint32_t bitmap[];
  const SpliceSeq sq{0, L, n_visual, 0, k, 0};
  uint32_t nv, nk, nj;
  splice_index_body(sq, bitmap, ids, visual_id, all_indices, pos, pos_rows, L, mask, sel, new_ids, new_pos, new_mask, src,
                    l_out, l_out, nv, nk, nj);
  if (threadIdx.x == 0) {
    stats[0] = (int32_t)nv;   // visual tokens found in input_ids (must equal n_visual)
    stats[1] = (int32_t)nk;   // output length (must equal L')
    stats[2] = (int32_t)nj;   // kept visual tokens (must equal k)
  }
}

// Packed batch: workgroup s splices sequence s.  Output offsets need no scan: sequence s starts at
// cu_seqlens[s] - cu_visual[s] + cu_kept[s].  stats (zeroed by the host wrapper) accumulate with integer atomics.
static __attribute__((unused)) __global__ __launch_bounds__(kSpliceThreads) void splice_index_batched_kernel(
    const int64_t* __restrict__ ids, const int32_t* __restrict__ cu_seqlens, const int32_t* __restrict__ cu_visual,
    const int32_t* __restrict__ cu_kept, int n_seq, int max_visual, int64_t visual_id, const int64_t* __restrict__ all_indices,
    const int64_t* __restrict__ pos, int pos_rows, int total_in, int64_t* __restrict__ sel, int64_t* __restrict__ new_ids,
    int64_t* __restrict__ new_pos, int32_t* __restrict__ src, int32_t* __restrict__ cu_out, int32_t* __restrict__ stats,
    int l_out) {
  extern __shared__ uint32_t bitmap[];
  const int s = blockIdx.x;
  SpliceSeq sq;
  sq.p0 = cu_seqlens[s];
  sq.L = cu_seqlens[s + 1] - sq.p0;
  sq.nvis = cu_visual[s + 1] - cu_visual[s];
  sq.j0 = cu_kept[s];
  sq.k = cu_kept[s + 1] - sq.j0;
  sq.q0 = sq.p0 - cu_visual[s] + sq.j0;
  const int len_out = sq.L - sq.nvis + sq.k;
  const bool sane = sq.L >= 0 && sq.nvis >= 0 && sq.nvis <= max_visual && sq.nvis <= sq.L && sq.k >= 0 && sq.k <= sq.nvis &&
                    sq.p0 >= 0 && sq.p0 + sq.L <= total_in && sq.q0 >= 0 && sq.q0 + len_out <= l_out;
  if (threadIdx.x == 0) {
    cu_out[s] = sq.q0;
    if (s == n_seq - 1) cu_out[n_seq] = sq.q0 + len_out;
  }
  if (!sane) {                       // uniform per workgroup: descriptors that disagree are reported, nothing is written
    if (threadIdx.x == 0) atomicAdd(&stats[3], 1);
    return;
  }
  uint32_t nv, nk, nj;
  splice_index_body(sq, bitmap, ids, visual_id, all_indices, pos, pos_rows, total_in, nullptr, sel, new_ids, new_pos, nullptr,
                    src, l_out, sq.q0 + len_out, nv, nk, nj);
  if (threadIdx.x == 0) {
    atomicAdd(&stats[0], (int32_t)nv);
    atomicAdd(&stats[1], (int32_t)nk);
    atomicAdd(&stats[2], (int32_t)nj);
    if ((int)nv != sq.nvis || (int)nk != len_out || (int)nj != sq.k) atomicAdd(&stats[3], 1);
  }
}

// n_rows / n_vis bound the two source tensors: a descriptor outside them (a splice whose token counts disagree with
// input_ids, reported through stats) yields a zero row instead of an out-of-bounds read.
template <typename T>
__global__ __launch_bounds__(256) void splice_embed_kernel(const T* __restrict__ embeds, const T* __restrict__ vis,
                                                           const int32_t* __restrict__ src, int l_out, int d, int n_rows,
                                                           int n_vis, T* __restrict__ out) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = blockIdx.x * 4 + wave; q < l_out; q += gridDim.x * 4) {
    const int s = src[q];
    const bool ok = s >= 0 ? s < n_rows : (-(int64_t)s - 1) < n_vis;
    const T* from = s >= 0 ? embeds + (int64_t)s * d : vis + (int64_t)(-s - 1) * d;
    const u32x4* sp = reinterpret_cast<const u32x4*>(from);
    u32x4* dp = reinterpret_cast<u32x4*>(out + (int64_t)q * d);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (int v = lane; v < d / V; v += 64) dp[v] = ok ? sp[v] : zero;
  }
}

}  // namespace vsel
