/*
 * vsel.h -- C-ABI of libvsel.so, the MI355X (gfx950) implementation of the VisionSelector hot path.
 *
 * Plain pointers and sizes only: no torch / HIP types in the signatures.  Every device pointer is a
 * raw HBM address (torch: tensor.data_ptr()), `stream` is a hipStream_t passed as void* (torch:
 * torch.cuda.current_stream().cuda_stream; NULL = the default stream).  All calls are asynchronous
 * on `stream`; none synchronises.  Return value: 0 = ok, otherwise a vsel_status; the message is
 * available from vsel_last_error() (thread-local).  Kernels are deterministic (no float atomics).
 *
 * Each entry point names the reference interface it replaces (paths under the reference repo;
 * FT = qwen-vl-finetune, EV = qwen-evaluation, OV = llava-ov-15).
 */
#ifndef VSEL_H
#define VSEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  VSEL_OK = 0,
  VSEL_ERR_INVALID = 1,      /* bad argument (shape, alignment, k out of range, null pointer) */
  VSEL_ERR_WORKSPACE = 2,    /* workspace too small: call the matching *_workspace_bytes()       */
  VSEL_ERR_HIP = 3,          /* a HIP runtime call / launch failed                               */
  VSEL_ERR_UNSUPPORTED = 4,  /* shape outside what the kernels implement                         */
  VSEL_ERR_BUSY = 5          /* more than 64 streams have launches of one attention kernel family in flight (a work-queue counter
                                slot per stream): synchronise one of them and call again                                            */
} vsel_status;
/* Streams and graphs (attention entries whose work items are queued).  A queued launch uses the work-queue counter slot of ITS stream
 * handle: pass explicit streams -- hipStreamPerThread (one handle value for a different stream in every host thread) is refused with
 * VSEL_ERR_UNSUPPORTED.  A captured graph bakes in the slot of its capture stream: replay it on that stream (or at least never
 * concurrently with itself or with eager launches of the same kernel family on the capture stream). */

typedef enum {
  VSEL_BF16 = 0,             /* tokens / weights stored as bfloat16                              */
  VSEL_F32 = 1               /* tokens / weights stored as float32                               */
} vsel_dtype;

const char* vsel_version(void);
const char* vsel_last_error(void);

/* Measurement hook (no reference counterpart; the reference times with torch.cuda.Event around the whole
 * forward, EV/token_compression/selector_model.py:228-234,353-359).  Between start and stop every kernel
 * libvsel launches is bracketed by HIP events recorded on the launch stream.  stop() synchronises the last
 * event and returns, per kernel name (comma-joined into `names`), the summed elapsed ms and launch count. */
int vsel_profile_start(void);
int vsel_profile_stop(char* names, size_t names_len, float* total_ms, int64_t* calls, int max_entries, int* n_entries);

/* ---------------------------------------------------------------------------------------------
 * Segments.  The token tensor is H[T, D] row-major.  A call scores `n_seg` independent segments
 * ("calls" of the reference: one image, or all images of a sample jointly -- the mean over tokens
 * in TransformerScorer.forward is over one segment):
 *   uniform:  seg_rows == NULL, every segment has `rows_per_seg` rows  (H viewed as [B, N, D])
 *   ragged :  seg_rows = DEVICE int32[n_seg + 1] row offsets (cu_seqlens style), seg_out = DEVICE
 *             int32[n_seg + 1] offsets of the kept rows (cumsum of k_s); `rows_per_seg` = max rows,
 *             `k` = max k_s.  k_s is computed on the host exactly as the reference does:
 *             max(1, int(N_s * budgets)) (EV/token_compression/selector_model.py:186).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t n_seg;           /* B                                                  */
  int64_t rows_per_seg;    /* N (uniform) or max_s N_s (ragged)                  */
  int64_t total_rows;      /* T = sum_s N_s                                      */
  int64_t k;               /* kept rows per segment (uniform) or max_s k_s       */
  int64_t total_out;       /* sum_s k_s                                          */
  const int32_t* seg_rows; /* DEVICE, or NULL for uniform                        */
  const int32_t* seg_out;  /* DEVICE, or NULL for uniform                        */
} vsel_segments;

/* Scorer parameters: TransformerScorer.{q_proj,k_proj}.{weight [Hd, D], bias [Hd]}
 * (FT/compression_method/selector_scorer.py:12-22), all of dtype `wdtype`, contiguous. */
typedef struct {
  const void* wq;
  const void* bq;
  const void* wk;
  const void* bk;
  int64_t d;               /* in_features  D                                     */
  int64_t hd;              /* hidden_dim   Hd                                    */
  vsel_dtype wdtype;
} vsel_scorer;

/* -------- inference: score + hard top-k + gather ------------------------------------------------
 * Replaces, in one call, EV/token_compression/selector_model.py:184-189
 *   learned_scores = importance_scorer(hidden_states[None])           (selector_scorer.py:34-55)
 *   all_indices    = learned_scores.topk(k).indices.sort().values
 *   hidden_new     = hidden_states[all_indices, :]
 * (same lines in OV/compression_method/modeling_selector.py:173-180).
 *   h      [T, D]  dtype `hdtype`
 *   out    [total_out, D] same dtype          (kept rows, segment-major, ascending index inside)
 *   idx    int64 [total_out]                  (index LOCAL to the segment, ascending)
 *   scores float32 [T]                        (the learned scores, fp32 accumulate)
 * Tie rule: larger score first, equal scores -> lower index first; NaN sorts greatest.          */
size_t vsel_lis_workspace_bytes(const vsel_segments* seg, int64_t d, int64_t hd);
int vsel_lis_select(void* stream, const void* h, vsel_dtype hdtype, const vsel_segments* seg,
                    const vsel_scorer* scorer, void* workspace, size_t workspace_bytes,
                    void* out, int64_t* idx, float* scores);

/* Same, for tokens stored in a PHYSICAL row order that differs from the LOGICAL order the reference selects in.
 * Qwen2.5-VL's merger emits tokens in window order and the reference un-reorders them with a full gather
 *   reverse_indices = argsort(window_index); hidden_states = hidden_states[reverse_indices, :]
 * (EV/token_compression/selector_model.py:179-181) right before the LIS block.  Passing
 *   logical_to_physical = reverse_indices, physical_to_logical = window_index   (int64 [T], DEVICE, global row numbers;
 *   the permutation must keep every segment's rows inside the segment)
 * gives the same scores / idx / out as running vsel_lis_select on the un-reordered tensor, without materialising it
 * (the column mean is order-invariant; scores are scattered to logical positions; kept rows are gathered through the map). */
int vsel_lis_select_permuted(void* stream, const void* h_physical, vsel_dtype hdtype, const vsel_segments* seg,
                             const vsel_scorer* scorer, void* workspace, size_t workspace_bytes,
                             const int64_t* logical_to_physical, const int64_t* physical_to_logical,
                             void* out, int64_t* idx, float* scores);

/* Scores only: TransformerScorer.forward (FT/compression_method/selector_scorer.py:34-55).      */
int vsel_lis_scores(void* stream, const void* h, vsel_dtype hdtype, const vsel_segments* seg,
                    const vsel_scorer* scorer, void* workspace, size_t workspace_bytes, float* scores);

/* Hard top-k on given scores: scores.topk(k).indices.sort().values
 * (EV/token_compression/selector_model.py:187-188).  Optionally also writes the 0/1 constraint mask
 * of FT/compression_method/selector_model.py:168-171 (mask may be NULL).                         */
int vsel_topk_select(void* stream, const float* scores, const vsel_segments* seg, int64_t* idx, float* mask);

/* Row gather hidden_states[idx, :] (EV/token_compression/selector_model.py:189).                 */
int vsel_gather_rows(void* stream, const void* h, vsel_dtype hdtype, int64_t d, const vsel_segments* seg,
                     const int64_t* idx, void* out);

/* -------- differentiable top-k ------------------------------------------------------------------
 * TopK.forward / _find_ts (FT/compression_method/selector_model.py:53-58,72-86): the t with
 * sum(sigmoid(x + t)) = k, fp32 -- the root the reference's 64 bisection steps converge to, found by
 * bracketed Newton steps on the same bracket (|t - reference t| ~ 1e-6, not bit-equal; the fused
 * training tail and this entry sum in different orders and agree to that noise too).
 * xs [B, N] float32 -> ps [B, N], ts [B].
 * Reference asserts 0 < k < N (:75) -> VSEL_ERR_INVALID.                                          */
int vsel_soft_topk_fwd(void* stream, const float* xs, int64_t b, int64_t n, int64_t k, float* ps, float* ts);
/* The same entry in the reference's own BFLOAT16 arithmetic (its released scorers run in bf16: EV/token_compression/
 * selector_model.py:75-89 then rounds lo / hi / mid, x + mid, the sigmoid and the sum -- spacing 2 at 256 .. 512 -- to bf16, and
 * the 64-step bisection stalls on a bf16 neighbour of the root, e.g. sum(ps) = 459.35 for k = 460).  xs [B, N] float32 (rounded
 * to bf16 on entry) -> ps, ts float32 holding bf16 values: the reference's last_combined_scores (:190) on the same
 * scores: ts bit for bit on every fixture (tests/golden/lisbf16_*.npz); ps bit for bit but for isolated elements (<= 2 per fixture)
 * one bf16 step apart, where the fp32 sigmoid sits on a bf16 rounding boundary (device expf vs libm).  Opt-in (selector attribute soft_topk_bf16_reference); the default returns the fp32 root. */
int vsel_soft_topk_fwd_bf16ref(void* stream, const float* xs, int64_t b, int64_t n, int64_t k, float* ps, float* ts);
/* TopK.backward (FT/compression_method/selector_model.py:60-70).                                  */
int vsel_soft_topk_bwd(void* stream, const float* grad_ps, const float* xs, const float* ts, int64_t b, int64_t n,
                       float* grad_xs);

/* -------- training LIS block --------------------------------------------------------------------
 * Forward of FT/compression_method/selector_model.py:158-173 (OV/compression_method/selector_model.py:127-142)
 * for ONE segment (the reference scores all images of the micro-batch jointly):
 *   scores = scorer(h); k given (host: int(N * budgets)); ps = topk(scores, k); h_new = (ps[:,None] * h).type(dtype)
 *   y = scatter(zeros, scores.topk(k).indices, 1);  bce = mean BCE(ps, y) with ATen's log clamp at -100 (:310)
 * Outputs: h_new [N, D] (hdtype), ps [N], y [N], scores [N], ts [1], bce [1]  (all float32 but h_new). */
size_t vsel_lis_train_workspace_bytes(int64_t n, int64_t d, int64_t hd);
int vsel_lis_train_fwd(void* stream, const void* h, vsel_dtype hdtype, int64_t n, int64_t k, const vsel_scorer* scorer,
                       void* workspace, size_t workspace_bytes, void* h_new, float* ps, float* y, float* scores,
                       float* ts, float* bce);
/* Backward of the same block.  The gradient reaching ps is
 *     dps_i = sum_d d_hnew[i,d] * h[i,d]  +  d_ps_ext[i]  +  dl_dbce * dBCE/dps_i
 *   d_ps_ext (nullable): gradient that arrived at ps from outside the block (the drop-in path calls
 *     F.binary_cross_entropy(img_mask, constraint_img_mask) in torch, FT/.../selector_model.py:310);
 *   dl_dbce: weight of the fused BCE term (the native trainer passes regularization_weight, else 0).
 *   out: dwq [Hd, D], dbq [Hd], dwk [Hd, D], dbk [Hd] float32 (overwritten); dh [N, D] (hdtype) or NULL.
 * Uses the closed form of autograd through selector_scorer.py:47-53 and TopK.backward (:60-70):
 * both weight gradients are rank-1 (SURVEY.md section 7 hard part 4).                              */
int vsel_lis_train_bwd(void* stream, const void* d_hnew, const void* h, vsel_dtype hdtype, int64_t n,
                       const vsel_scorer* scorer, const float* ps, const float* y, const float* scores,
                       const float* ts, const float* d_ps_ext, float dl_dbce, void* workspace,
                       size_t workspace_bytes, float* dwq, float* dbq, float* dwk, float* dbk, void* dh);

/* The same backward with the two weight gradients as their rank-1 FACTORS ("rank-1-factor all-gather",
 * the data-parallel exchange of qwen-vl-finetune/scripts/sft_7b.sh:9-12,71-74 with 2 (Hd + D) + 2 Hd floats per micro-batch
 * instead of 2 Hd D + 2 Hd):   dWq = a (x) gx,   dWk = dk (x) xsum.
 *   out: factors float32 [2 (Hd + D)] = a [Hd] | gx [D] | dk [Hd] | xsum [D]; dbq, dbk float32 [Hd]; dh [N, D] or NULL.
 * Nothing of size Hd x D is written (vsel_lis_train_bwd writes 2 Hd D floats per call).                              */
int vsel_lis_train_bwd_factors(void* stream, const void* d_hnew, const void* h, vsel_dtype hdtype, int64_t n,
                               const vsel_scorer* scorer, const float* ps, const float* y, const float* scores,
                               const float* ts, const float* d_ps_ext, float dl_dbce, void* workspace,
                               size_t workspace_bytes, float* factors, float* dbq, float* dbk, void* dh);

/* Dense gradients from R payload rows of vsel_lis_train_bwd_factors (row = a | gx | dk | xsum | dbq | dbk, 2 (Hd + D) + 2 Hd
 * floats each; own micro-batches and / or the rows all-gathered from the other ranks):
 *   dwq = scale sum_i a_i (x) gx_i,  dwk = scale sum_i dk_i (x) xsum_i,  dbq = scale sum_i dbq_i,  dbk likewise  (i in order).
 * scale = 1 / world gives the mean over ranks that the reference's DDP / ZeRO all-reduce leaves in .grad.                 */
int vsel_lis_factors_to_grads(void* stream, const float* payload, int64_t n_rows, int64_t hd, int64_t d, float scale,
                              float* dwq, float* dbq, float* dwk, float* dbk);

/* Backward of TransformerScorer.forward alone (autograd through FT/compression_method/selector_scorer.py:47-53)
 * for one segment: g = dL/dscores [N] float32 -> dwq, dbq, dwk, dbk float32 (overwritten), dh [N, D] or NULL.
 * Workspace: vsel_lis_train_workspace_bytes(n, d, hd).                                             */
int vsel_lis_scores_bwd(void* stream, const float* g, const void* h, vsel_dtype hdtype, int64_t n,
                        const vsel_scorer* scorer, void* workspace, size_t workspace_bytes, float* dwq, float* dbq,
                        float* dwk, float* dbk, void* dh);

/* -------- sequence splice after selection (batch 1, like the reference) ---------------------------
 * Replaces the index algebra of EV/token_compression/selector_model.py:246-262 (image), :264-290 (video) and the
 * position_ids / attention_mask slices of :311-320 (OV/compression_method/modeling_selector.py:259-276, 311-314) with two
 * launches and no host sync.  A position is kept iff it is not `visual_token_id`, or its rank among the visual tokens is
 * in `all_indices` (ascending, as vsel_lis_select returns it).  L' = seq_len - n_visual + k.
 *   in : input_ids int64 [L]; all_indices int64 [k]; inputs_embeds [L, d_llm]; visual_embeds [k, d_llm] (kept rows);
 *        position_ids int64 [pos_rows, L] (3 for M-RoPE, 1 for 1-D, 0 = none); attention_mask int64 [L] or NULL
 *   out: selected_indices, new_input_ids int64 [L']; new_inputs_embeds [L', d_llm]; new_position_ids [pos_rows, L'];
 *        new_attention_mask [L'] or NULL; src_scratch int32 [L']; stats int32 [3] = {visual tokens found, rows written,
 *        kept visual rows} (device; the caller may check them against n_visual / L' / k -- the reference raises
 *        ValueError on a token-count mismatch, FT/compression_method/selector_model.py:210-213).                      */
int vsel_splice(void* stream, const int64_t* input_ids, int64_t seq_len, int64_t visual_token_id,
                const int64_t* all_indices, int64_t k, int64_t n_visual, const void* inputs_embeds,
                const void* visual_embeds, vsel_dtype dtype, int64_t d_llm, const int64_t* position_ids,
                int64_t pos_rows, const int64_t* attention_mask, int64_t* selected_indices, int64_t* new_input_ids,
                void* new_inputs_embeds, int64_t* new_position_ids, int64_t* new_attention_mask,
                int32_t* src_scratch, int32_t* stats);

/* -------- producer-side fusion: single-sweep LIS ---------------------------------
 * The tokens come out of the patch merger  ln_q -> Linear -> GELU -> Linear  (Qwen2_5_VLPatchMerger,
 * EV/qwen25vl/modeling_qwen2_5_vl.py:148-161; RicePatchMerger, OV/llavaonevision1_5/modeling_llavaonevision1_5.py:255-268)
 * and the scorer needs their row mean first (selector_scorer.py:47-53 collapsed: DESIGN.md section 3).  The last Linear is
 * linear, so  sum_rows(H) = sum_rows(G) W2^T + N b2  with G the GELU output:
 *   vsel_gelu_colsum  replaces the merger's GELU launch: y = GELU(x) (erf form, fp32 math, as nn.GELU()) and
 *     col_sums [n_seg, cols] fp32 = per-segment column sums of y as rounded to `dtype` -- no extra HBM traffic
 *     (col_sums == NULL: y only, workspace unused -- the same streaming GELU without its sums, which is what a
 *     benchmark must subtract to price the sums);
 *   vsel_lis_select_presummed  = vsel_lis_select (row maps NULL) / vsel_lis_select_permuted (row maps given) with the
 *     column sums of the tokens supplied by the caller (col_sums [n_seg, D] fp32), skipping the first sweep over H.
 * The caller forms sum_rows(H) from sum_rows(G) with one skinny fp32 GEMM (a library call).                          */
size_t vsel_gelu_colsum_workspace_bytes(const vsel_segments* seg, int64_t cols);
int vsel_gelu_colsum(void* stream, const void* x, vsel_dtype dtype, const vsel_segments* seg, int64_t cols, void* y,
                     float* col_sums, void* workspace, size_t workspace_bytes);
/* sum_rows(H) from sum_rows(G) through the merger's last Linear H = G W^T + b (Qwen2_5_VLPatchMerger.mlp[2],
 * EV/qwen25vl/modeling_qwen2_5_vl.py:148-161), on the STORED weight:  out[s][j] = sum_i in[s][i] W[j][i] + N_s b[j].
 *   col_sums_in fp32 [n_seg, cin] (vsel_gelu_colsum's output); weight [cout, cin] row-major, bias [cout] or NULL, both `wdtype`;
 *   N_s from `seg`; col_sums_out fp32 [n_seg, cout] = what vsel_lis_select_presummed / vsel_lis_select_splice take.
 * fp32 accumulation (bf16x3-exact MFMA operands beyond 8 segments); deterministic.                                  */
size_t vsel_colsum_linear_workspace_bytes(int64_t n_seg, int64_t cin, int64_t cout);
int vsel_colsum_linear(void* stream, const float* col_sums_in, const vsel_segments* seg, const void* weight, const void* bias,
                       vsel_dtype wdtype, int64_t cin, int64_t cout, float* col_sums_out, void* workspace,
                       size_t workspace_bytes);
int vsel_lis_select_presummed(void* stream, const void* h, vsel_dtype hdtype, const vsel_segments* seg,
                              const vsel_scorer* scorer, const float* col_sums, void* workspace, size_t workspace_bytes,
                              const int64_t* logical_to_physical, const int64_t* physical_to_logical, void* out,
                              int64_t* idx, float* scores);

/* -------- packed-batch splice ----------------------------------------------
 * The reference's generation forward is batch 1 (`assert ... "selector only support single batch"`,
 * EV/token_compression/selector_model.py:270; OV/compression_method/modeling_selector.py:259).  This entry applies the same
 * index algebra (:246-262, :264-290, :311-320) to S prompts packed back to back -- the layout the var-len prefill of
 * FT/qwenvl/train/trainer.py:79-113 consumes -- in two launches, and emits cu_seqlens' for vsel_varlen_attn_fwd.
 * Sequence s owns packed positions [cu_seqlens[s], cu_seqlens[s+1]), visual rows [cu_visual[s], cu_visual[s+1]) (its LIS
 * segment, as passed to vsel_lis_select) and kept rows all_indices[cu_kept[s] .. cu_kept[s+1]) holding LOCAL ranks inside
 * that segment, ascending (exactly vsel_lis_select's idx output for ragged segments).  No scan is needed for the output
 * offsets: cu_seqlens'[s] = cu_seqlens[s] - cu_visual[s] + cu_kept[s].
 *   in : input_ids int64 [T]; cu_seqlens, cu_visual, cu_kept int32 [S+1] (device); max_visual >= every segment's size;
 *        inputs_embeds [T, d_llm]; visual_embeds [K, d_llm]; position_ids int64 [pos_rows, T] or NULL (pos_rows = 0)
 *   out: selected_indices, new_input_ids int64 [T']; new_inputs_embeds [T', d_llm]; new_position_ids [pos_rows, T'];
 *        cu_seqlens_out int32 [S+1]; src_scratch int32 [T']; stats int32 [4] = {visual tokens found, rows written, kept
 *        visual rows, sequences whose token counts disagree with cu_visual / cu_kept (0 on success)}.  T' = T - N + K. */
int vsel_splice_batched(void* stream, const int64_t* input_ids, int64_t total_len, const int32_t* cu_seqlens,
                        const int32_t* cu_visual, const int32_t* cu_kept, int64_t n_seq, int64_t max_visual,
                        int64_t total_visual, int64_t total_kept, int64_t visual_token_id, const int64_t* all_indices,
                        const void* inputs_embeds, const void* visual_embeds, vsel_dtype dtype, int64_t d_llm,
                        const int64_t* position_ids, int64_t pos_rows, int64_t* selected_indices, int64_t* new_input_ids,
                        void* new_inputs_embeds, int64_t* new_position_ids, int32_t* cu_seqlens_out, int32_t* src_scratch,
                        int32_t* stats);

/* -------- score + hard top-k + splice in one call: the kept rows are written ONCE ---------------------------------
 * Replaces EV/token_compression/selector_model.py:184-189 (scores, topk + sort, hidden_states[all_indices, :]) TOGETHER WITH
 * the splice of :246-262 / :264-290 / :311-320 (OV/compression_method/modeling_selector.py:173-180, :259-276, :311-314).
 * The reference materialises hidden_states_new [k, D] and masked_scatter()s it into inputs_embeds; vsel_lis_select +
 * vsel_splice(_batched) do the same with two copies of k x D.  Here the kept rows go straight from the token tensor into
 * new_inputs_embeds (token width D must equal the LLM width, same dtype): one k x D write + read and, for a few prompts, two
 * launches less.  Every output is bit-identical to vsel_lis_select(_permuted / _presummed) followed by vsel_splice(_batched).
 *   seg          one LIS segment per PROMPT (n_seg = S prompts; all images of a prompt are scored jointly, EV :184-186); prompt s
 *                holds exactly its segment's rows as placeholder tokens, in order
 *   col_sums     NULL, or the producer's column sums (vsel_lis_select_presummed); row maps NULL or both given (.._permuted)
 *   input_ids    int64 [T] = S prompts back to back; cu_seqlens int32 [S + 1] DEVICE (may be NULL when S == 1: one prompt of T)
 *   max_len_out  max over prompts of L_s - N_s + k_s (host value; sizes the launch)
 *   inputs_embeds [T, D]; position_ids int64 [pos_rows, T] or NULL; attention_mask int64 [T] or NULL
 *   out: idx int64 [sum k] (local ranks, ascending -- vsel_lis_select's idx), scores fp32 [sum N]; selected_indices,
 *        new_input_ids int64 [T']; new_inputs_embeds [T', D]; new_position_ids [pos_rows, T']; new_attention_mask [T'] or NULL;
 *        cu_seqlens_out int32 [S + 1] (may be NULL when S == 1); src_scratch int32 [T']; stats int32 [4] = {visual tokens
 *        found, rows written, kept visual rows, prompts whose token counts disagree (0 on success)}.  T' = T - sum N + sum k.
 *        soft_ps fp32 [sum N], soft_ts fp32 [S] (both or neither; NULL = not wanted): the soft top-k of the same scores, i.e.
 *        vsel_soft_topk_fwd(scores, k) bit for bit -- what the reference's eval forward publishes next to the selection as
 *        visual.last_combined_scores (EV/token_compression/selector_model.py:190).  For one prompt of <= 4096 visual tokens
 *        (the reference's call) it is computed by one extra workgroup of the select-splice launch instead of a launch of its own;
 *        uniform segments only, written only when 0 < k < N (the reference asserts that, :75).
 * Workspace: vsel_lis_workspace_bytes(seg, D, Hd).                                                                        */
int vsel_lis_select_splice(void* stream, const void* h, vsel_dtype hdtype, const vsel_segments* seg,
                           const vsel_scorer* scorer, void* workspace, size_t workspace_bytes, const float* col_sums,
                           const int64_t* logical_to_physical, const int64_t* physical_to_logical,
                           const int64_t* input_ids, int64_t total_len, const int32_t* cu_seqlens, int64_t max_len_out,
                           int64_t visual_token_id, const void* inputs_embeds, const int64_t* position_ids, int64_t pos_rows,
                           const int64_t* attention_mask, int64_t* idx, float* scores, int64_t* selected_indices,
                           int64_t* new_input_ids, void* new_inputs_embeds, int64_t* new_position_ids,
                           int64_t* new_attention_mask, int32_t* cu_seqlens_out, int32_t* src_scratch, int32_t* stats,
                           float* soft_ps, float* soft_ts);

/* The same on GIVEN scores (fp32 [sum N], logical order; an input here): hard top-k (vsel_topk_select's tie rule) + splice with
 * the kept rows read from h [sum N, d] (through logical_to_physical when not NULL).  EV :187-189 + :246-262.             */
int vsel_topk_select_splice(void* stream, const void* h, vsel_dtype hdtype, int64_t d, const vsel_segments* seg,
                            const float* scores, const int64_t* logical_to_physical, const int64_t* input_ids,
                            int64_t total_len, const int32_t* cu_seqlens, int64_t max_len_out, int64_t visual_token_id,
                            const void* inputs_embeds, const int64_t* position_ids, int64_t pos_rows,
                            const int64_t* attention_mask, int64_t* idx, int64_t* selected_indices, int64_t* new_input_ids,
                            void* new_inputs_embeds, int64_t* new_position_ids, int64_t* new_attention_mask,
                            int32_t* cu_seqlens_out, int32_t* src_scratch, int32_t* stats, float* soft_ps, float* soft_ts);

/* -------- var-len causal attention (compressed-sequence prefill) --------------------------------
 * Replaces flash_attn_varlen_func as called by FT/qwenvl/train/trainer.py:101-113 and the FA2 prefill
 * of EV/qwen25vl/modeling_qwen2_5_vl.py:900 / OV/llavaonevision1_5/modeling_llavaonevision1_5.py:686.
 *   q [T, Hq, d], k/v [T, Hkv, d] bf16, already rotated; cu_seqlens DEVICE int32 [n_seq + 1]
 *   (the tensor the reference passes as `attention_mask`); out [T, Hq, d] bf16.  d in {128 (the LLMs), 80, 64 (vision towers)}.
 * Math: softmax(q k^T * scale + causal) v with fp32 softmax (EV/qwen25vl/modeling_qwen2_5_vl.py:777-797). */
int vsel_varlen_attn_fwd(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens,
                         int64_t n_seq, int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d,
                         float scale, int causal, void* out);

/* Same kernel against a PAGED key/value cache (serving): queries are a packed var-len batch (cu_seqlens_q), sequence s
 * has seqlens_k[s] keys living in pages block_table[s][0 .. ceil(seqlens_k[s] / page_size)) of the pools
 * k_cache / v_cache [n_pages, page_size, Hkv, d] bf16.  Causal masking is bottom-right aligned (query i of a sequence sees
 * keys <= i + (seqlens_k - q_len)), i.e. chunked prefill / prefill-with-prefix / decode against the compressed cache.
 * No reference counterpart beyond the flash-attn call sites above (the reference keeps a contiguous HF DynamicCache);
 * the north-star asks for a "paged / var-len" kernel.  Rows that see no key output zeros.                               */
int vsel_paged_attn_fwd(void* stream, const void* q, const void* k_cache, const void* v_cache,
                        const int32_t* cu_seqlens_q, const int32_t* seqlens_k, const int32_t* block_table,
                        int64_t max_pages_per_seq, int64_t page_size, int64_t n_seq, int64_t max_seqlen_q, int64_t hq,
                        int64_t hkv, int64_t d, float scale, int causal, void* out);

/* Same kernel with separate query / key packings (flash_attn_varlen_func's general form: cu_seqlens_q != cu_seqlens_k):
 * sequence s owns query rows [cu_seqlens_q[s], cu_seqlens_q[s+1]) of q and key rows [cu_seqlens_k[s], +seqlens_k[s]) of
 * k / v (contiguous, [total_k, hkv, d]); causal masking is bottom-right aligned (flash-attn >= 2.1).  Forward only.     */
int vsel_varlen_attn_fwd_kv(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens_q,
                            const int32_t* cu_seqlens_k, const int32_t* seqlens_k, int64_t n_seq, int64_t max_seqlen_q,
                            int64_t hq, int64_t hkv, int64_t d, float scale, int causal, void* out);

/* Strided form of the two entries above, for callers that hold q / k / v head-major ([B, H, L, d], the layout of
 * HuggingFace attention modules after the rotary embedding, EV/qwen25vl/modeling_qwen2_5_vl.py:765-775): element (sequence s,
 * row l, head h) of q lives at  cu_seqlens_q[s] * hq * d + l * q_row_stride + h * q_head_stride  (k and v likewise with
 * cu_seqlens_k or cu_seqlens_q, hkv and their own strides -- HF keeps v as a transposed view of the packed projection); packed [T, H, d] is row stride H * d, head stride d, head-major
 * is row stride d, head stride L * d with cu = [0, L, 2L, ...].  cu_seqlens_k / seqlens_k NULL: keys share the query
 * packing.  out is always packed [T, hq, d].  Saves the two transposing copies per layer of the HF path.             */
int vsel_varlen_attn_fwd_strided(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens_q,
                                 const int32_t* cu_seqlens_k, const int32_t* seqlens_k, int64_t n_seq, int64_t max_seqlen_q,
                                 int64_t hq, int64_t hkv, int64_t d, int64_t q_row_stride, int64_t q_head_stride,
                                 int64_t k_row_stride, int64_t k_head_stride, int64_t v_row_stride, int64_t v_head_stride,
                                 float scale, int causal, void* out);

/* -------- the same forward with a caller workspace: key-range parts for ONE long sequence ------------------------
 * vsel_varlen_attn_fwd (and _lse when lse != NULL) on batches the launch cannot balance as they are -- one uncompressed prompt, or a
 * few of equal length: 256-query items cut over key ranges, fp32 partial outputs + (m, l) in `ws`, merged in part order by a second
 * launch (csrc/attn_fwd64_parts.hip; the reference's call is the same flash_attn_varlen_func, EV/qwen25vl/modeling_qwen2_5_vl.py:900).
 * vsel_varlen_attn_fwd_workspace_bytes: bytes the call would use for these shapes, 0 when it would run the workspace-free forms
 * (then ws may be NULL; the call is vsel_varlen_attn_fwd[_lse]).  Deterministic; not bit-identical to the workspace-free forms
 * (another fp32 association: within a bf16 rounding).  head_dim 128, causal, every sequence max_seqlen tokens.                    */
size_t vsel_varlen_attn_fwd_workspace_bytes(int64_t n_seq, int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d, int causal);
int vsel_varlen_attn_fwd_ws(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens, int64_t n_seq,
                            int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d, float scale, int causal, void* out,
                            float* lse, void* ws, size_t ws_bytes);

/* -------- var-len attention for TRAINING: forward that also saves the log-sum-exp, and the backward -------------
 * The reference trains the LIS through the frozen LLM with flash_attn_varlen_func (FT/qwenvl/train/trainer.py:101-113,
 * patched in by replace_qwen2_vl_attention_class :150-160), so dQ / dK / dV of the same op are on the training path.
 * Math = autograd of the eager formula EV/qwen25vl/modeling_qwen2_5_vl.py:777-797 (see csrc/attn_bwd.hip).
 *   vsel_varlen_attn_fwd_lse: as vsel_varlen_attn_fwd, plus lse float32 [total, hq] = log sum_j exp(scale * q_i.k_j) over the
 *     visible keys (natural log).
 *   vsel_varlen_attn_bwd: dout, out [total, hq, 128] bf16, lse from the forward -> dq [total, hq, 128], dk, dv
 *     [total, hkv, 128] bf16 (overwritten; GQA groups are summed in fp32 inside the kernel).  q and k share cu_seqlens.
 *     Deterministic: no float atomics, reruns are bit-identical.  Workspace: vsel_varlen_attn_bwd_workspace_bytes.     */
int vsel_varlen_attn_fwd_lse(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens,
                             int64_t n_seq, int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d,
                             float scale, int causal, void* out, float* lse);
size_t vsel_varlen_attn_bwd_workspace_bytes(int64_t total, int64_t hq, int64_t hkv, int64_t n_seq, int64_t max_seqlen);
int vsel_varlen_attn_bwd(void* stream, const void* dout, const void* q, const void* k, const void* v, const void* out,
                         const float* lse, const int32_t* cu_seqlens, int64_t n_seq, int64_t max_seqlen, int64_t total,
                         int64_t hq, int64_t hkv, int64_t d, float scale, int causal, void* workspace,
                         size_t workspace_bytes, void* dq, void* dk, void* dv);

#ifdef __cplusplus
}
#endif
#endif /* VSEL_H */
