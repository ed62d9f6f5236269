// Sequence splice after selection, on device and without a host sync.
//
// Reference (batch 1): qwen-evaluation/token_compression/selector_model.py:246-262 (image), :264-290 (video),
// :311-320 (M-RoPE position_ids / attention_mask slice); llava-ov-15/compression_method/modeling_selector.py:259-276,311-314.
// The reference builds the kept-position list with torch.where / cat / sort / index / masked_scatter (a dozen launches and
// a device->host sync for the dynamic shapes).  Here the output length L' = L - N + k is known on the host, so:
//   splice_index_kernel   one workgroup: bitmap of kept visual ranks in LDS, then an ordered ballot/popcount scan over the
//                         L positions emits selected_indices, input_ids', attention_mask', position_ids'[r][.] and a
//                         source descriptor per output row (text row p, or kept visual row j)
//   splice_embed_kernel   wave per output row: copy D_llm elements from inputs_embeds[p] or from the kept visual rows
// A position p is kept iff it is not a visual token, or its rank among the visual tokens is in all_indices.  For the
// reference's video branch (one contiguous <vision_start> video... <vision_end> block) this is the same set as :284-287.
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
    uint32_t& out_vis, uint32_t& out_keep, uint32_t& out_kv) {
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
      src[qo] = kept_vis ? -(int32_t)(sq.j0 + j + 1) : sq.p0 + p;
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

__global__ __launch_bounds__(kSpliceThreads) void splice_index_kernel(
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
__global__ __launch_bounds__(kSpliceThreads) void splice_index_batched_kernel(
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

using namespace vsel;

extern "C" int vsel_splice(void* stream, const int64_t* input_ids, int64_t seq_len, int64_t visual_token_id,
                           const int64_t* all_indices, int64_t k, int64_t n_visual, const void* inputs_embeds,
                           const void* visual_embeds, vsel_dtype dtype, int64_t d_llm, const int64_t* position_ids,
                           int64_t pos_rows, const int64_t* attention_mask, int64_t* selected_indices, int64_t* new_input_ids,
                           void* new_inputs_embeds, int64_t* new_position_ids, int64_t* new_attention_mask,
                           int32_t* src_scratch, int32_t* stats) {
  if (!input_ids || !inputs_embeds || !selected_indices || !new_input_ids || !new_inputs_embeds || !src_scratch || !stats ||
      (k > 0 && (!all_indices || !visual_embeds)))
    return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (seq_len < 1 || seq_len >= (1ll << 31) || k < 0 || n_visual < k || n_visual > seq_len)
    return fail(VSEL_ERR_INVALID, "bad splice sizes (L=%lld, N=%lld, k=%lld)", (long long)seq_len, (long long)n_visual, (long long)k);
  if (n_visual > kMaxVisualBits) return fail(VSEL_ERR_UNSUPPORTED, "more than %d visual tokens", kMaxVisualBits);
  if (pos_rows < 0 || pos_rows > 4 || (pos_rows > 0 && (!position_ids || !new_position_ids)))
    return fail(VSEL_ERR_INVALID, "bad position_ids arguments");
  if ((attention_mask == nullptr) != (new_attention_mask == nullptr)) return fail(VSEL_ERR_INVALID, "attention_mask in/out mismatch");
  const int vec = dtype == VSEL_BF16 ? 8 : 4;
  if (dtype != VSEL_BF16 && dtype != VSEL_F32) return fail(VSEL_ERR_INVALID, "bad dtype");
  if (d_llm < vec || d_llm % vec) return fail(VSEL_ERR_UNSUPPORTED, "d_llm must be a multiple of %d", vec);
  if (((uintptr_t)inputs_embeds | (uintptr_t)(k > 0 ? visual_embeds : nullptr) | (uintptr_t)new_inputs_embeds) & 15)
    return fail(VSEL_ERR_INVALID, "embeddings must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  const int l_out = (int)(seq_len - n_visual + k);
  const size_t lds = (size_t)((n_visual + 31) / 32) * 4 + 16;
  hipLaunchKernelGGL(splice_index_kernel, dim3(1), dim3(kSpliceThreads), lds, st, input_ids, (int)seq_len, visual_token_id,
                     all_indices, (int)k, (int)n_visual, position_ids, (int)pos_rows, attention_mask, selected_indices,
                     new_input_ids, new_position_ids, new_attention_mask, src_scratch, stats, l_out);
  VSEL_AFTER_LAUNCH(st, "splice_index_kernel");
  const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(l_out, 4), 2048);
  if (l_out > 0) {
    if (dtype == VSEL_BF16)
      hipLaunchKernelGGL((splice_embed_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)inputs_embeds,
                         (const bf16_t*)visual_embeds, src_scratch, l_out, (int)d_llm, (int)seq_len, (int)k,
                         (bf16_t*)new_inputs_embeds);
    else
      hipLaunchKernelGGL((splice_embed_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)inputs_embeds,
                         (const float*)visual_embeds, src_scratch, l_out, (int)d_llm, (int)seq_len, (int)k,
                         (float*)new_inputs_embeds);
    VSEL_AFTER_LAUNCH(st, "splice_embed_kernel");
  }
  return VSEL_OK;
}

extern "C" int vsel_splice_batched(void* stream, const int64_t* input_ids, int64_t total_len, const int32_t* cu_seqlens,
                                   const int32_t* cu_visual, const int32_t* cu_kept, int64_t n_seq, int64_t max_visual,
                                   int64_t total_visual, int64_t total_kept, int64_t visual_token_id,
                                   const int64_t* all_indices, const void* inputs_embeds, const void* visual_embeds,
                                   vsel_dtype dtype, int64_t d_llm, const int64_t* position_ids, int64_t pos_rows,
                                   int64_t* selected_indices, int64_t* new_input_ids, void* new_inputs_embeds,
                                   int64_t* new_position_ids, int32_t* cu_seqlens_out, int32_t* src_scratch, int32_t* stats) {
  if (!input_ids || !cu_seqlens || !cu_visual || !cu_kept || !inputs_embeds || !selected_indices || !new_input_ids ||
      !new_inputs_embeds || !cu_seqlens_out || !src_scratch || !stats || (total_kept > 0 && (!all_indices || !visual_embeds)))
    return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (n_seq < 1 || n_seq > 65535 || total_len < 1 || total_len >= (1ll << 31) || total_kept < 0 || total_visual < total_kept ||
      total_visual > total_len || max_visual < 0 || max_visual > total_visual)
    return fail(VSEL_ERR_INVALID, "bad batched splice sizes (S=%lld, T=%lld, N=%lld, K=%lld)", (long long)n_seq,
                (long long)total_len, (long long)total_visual, (long long)total_kept);
  if (max_visual > kMaxVisualBits) return fail(VSEL_ERR_UNSUPPORTED, "more than %d visual tokens in one sequence", kMaxVisualBits);
  if (pos_rows < 0 || pos_rows > 4 || (pos_rows > 0 && (!position_ids || !new_position_ids)))
    return fail(VSEL_ERR_INVALID, "bad position_ids arguments");
  const int vec = dtype == VSEL_BF16 ? 8 : 4;
  if (dtype != VSEL_BF16 && dtype != VSEL_F32) return fail(VSEL_ERR_INVALID, "bad dtype");
  if (d_llm < vec || d_llm % vec) return fail(VSEL_ERR_UNSUPPORTED, "d_llm must be a multiple of %d", vec);
  if (((uintptr_t)inputs_embeds | (uintptr_t)(total_kept > 0 ? visual_embeds : nullptr) | (uintptr_t)new_inputs_embeds) & 15)
    return fail(VSEL_ERR_INVALID, "embeddings must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  const int l_out = (int)(total_len - total_visual + total_kept);
  if (hipMemsetAsync(stats, 0, 4 * sizeof(int32_t), st) != hipSuccess) return fail(VSEL_ERR_HIP, "hipMemsetAsync(stats)");
  // a sequence whose descriptors are rejected writes nothing: pre-mark every source row invalid (0x7f7f7f7f is past any row)
  if (l_out > 0 && hipMemsetAsync(src_scratch, 0x7f, (size_t)l_out * sizeof(int32_t), st) != hipSuccess)
    return fail(VSEL_ERR_HIP, "hipMemsetAsync(src)");
  const size_t lds = (size_t)((max_visual + 31) / 32) * 4 + 16;
  hipLaunchKernelGGL(splice_index_batched_kernel, dim3((unsigned)n_seq), dim3(kSpliceThreads), lds, st, input_ids, cu_seqlens,
                     cu_visual, cu_kept, (int)n_seq, (int)max_visual, visual_token_id, all_indices, position_ids, (int)pos_rows,
                     (int)total_len, selected_indices, new_input_ids, new_position_ids, src_scratch, cu_seqlens_out, stats, l_out);
  VSEL_AFTER_LAUNCH(st, "splice_index_batched_kernel");
  const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(l_out, 4), 2048);
  if (l_out > 0) {
    if (dtype == VSEL_BF16)
      hipLaunchKernelGGL((splice_embed_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)inputs_embeds,
                         (const bf16_t*)visual_embeds, src_scratch, l_out, (int)d_llm, (int)total_len, (int)total_kept,
                         (bf16_t*)new_inputs_embeds);
    else
      hipLaunchKernelGGL((splice_embed_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)inputs_embeds,
                         (const float*)visual_embeds, src_scratch, l_out, (int)d_llm, (int)total_len, (int)total_kept,
                         (float*)new_inputs_embeds);
    VSEL_AFTER_LAUNCH(st, "splice_embed_kernel");
  }
  return VSEL_OK;
}
