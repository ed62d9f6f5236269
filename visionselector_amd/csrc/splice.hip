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
#include "splice_kernels.h"


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
  VSEL_LAUNCH(splice_index_kernel, dim3(1), dim3(kSpliceThreads), lds, st, input_ids, (int)seq_len, visual_token_id,
                     all_indices, (int)k, (int)n_visual, position_ids, (int)pos_rows, attention_mask, selected_indices,
                     new_input_ids, new_position_ids, new_attention_mask, src_scratch, stats, l_out);
  VSEL_AFTER_LAUNCH(st, "splice_index_kernel");
  const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(l_out, 4), 2048);
  if (l_out > 0) {
    if (dtype == VSEL_BF16)
      VSEL_LAUNCH((splice_embed_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)inputs_embeds,
                         (const bf16_t*)visual_embeds, src_scratch, l_out, (int)d_llm, (int)seq_len, (int)k,
                         (bf16_t*)new_inputs_embeds);
    else
      VSEL_LAUNCH((splice_embed_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)inputs_embeds,
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
  VSEL_LAUNCH(splice_index_batched_kernel, dim3((unsigned)n_seq), dim3(kSpliceThreads), lds, st, input_ids, cu_seqlens,
                     cu_visual, cu_kept, (int)n_seq, (int)max_visual, visual_token_id, all_indices, position_ids, (int)pos_rows,
                     (int)total_len, selected_indices, new_input_ids, new_position_ids, src_scratch, cu_seqlens_out, stats, l_out);
  VSEL_AFTER_LAUNCH(st, "splice_index_batched_kernel");
  const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(l_out, 4), 2048);
  if (l_out > 0) {
    if (dtype == VSEL_BF16)
      VSEL_LAUNCH((splice_embed_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)inputs_embeds,
                         (const bf16_t*)visual_embeds, src_scratch, l_out, (int)d_llm, (int)total_len, (int)total_kept,
                         (bf16_t*)new_inputs_embeds);
    else
      VSEL_LAUNCH((splice_embed_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)inputs_embeds,
                         (const float*)visual_embeds, src_scratch, l_out, (int)d_llm, (int)total_len, (int)total_kept,
                         (float*)new_inputs_embeds);
    VSEL_AFTER_LAUNCH(st, "splice_embed_kernel");
  }
  return VSEL_OK;
}
