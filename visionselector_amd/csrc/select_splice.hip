// vsel_lis_select_splice: LIS scores -> hard top-k -> sequence splice, with the kept rows written ONCE, straight from the
// token tensor H into inputs_embeds' (SURVEY.md section 8f N1 on top of section 8a A2/A3/A9).
//
// Reference: qwen-evaluation/token_compression/selector_model.py:184-189 (scores, topk + sort, hidden_states[all_indices, :])
// followed by :246-262 / :264-290 / :311-320 (the index algebra that drops the unselected placeholders and scatters the kept
// rows into inputs_embeds); llava-ov-15/compression_method/modeling_selector.py:173-180, :259-276, :311-314.  The reference
// materialises hidden_states_new [k, D] and then masked_scatter()s it; vsel_lis_select + vsel_splice did the same with two
// copies of k x D.  Here the [k, D] tensor never exists:
//   few prompts (<= kSelectSpliceMaxSeq, <= 32 768 visual tokens each): ONE launch after the scores.  Every 1024-thread
//     workgroup repeats the segment's integer radix select (keys in registers, as topk_select_reg_kernel), leaves the kept ranks
//     as a bitmap in LDS, runs the ordered scan over the prompt's positions (as splice_index_body) and produces the output rows
//     [qb, qe) it owns: index arrays + row copies from inputs_embeds or from H.  idx (all_indices) is written by slices.
//   otherwise: topk_select -> splice_index_seg_kernel (one workgroup per prompt, descriptors point into H) -> splice_embed_kernel.
// Integer arithmetic and raw 16-byte copies only: bit-identical to vsel_lis_select + vsel_splice(_batched) in every output.
#include "lis_kernels.h"
#include "softtopk.h"
#include "lis_small.h"
#include "proj_bf16x3.h"
#include "splice_kernels.h"

namespace vsel {

constexpr int kSelectSpliceMaxSeq = 8;          // prompts per call served by the one-launch form
constexpr int kSelectSpliceMaxRows = 16;        // output rows per workgroup (upper bound of rows_per_block)

// geometry of prompt s: positions [p0, p0 + L), visual rows [rb, rb + nvis) of H (its LIS segment), kept ranks idx[ob .. ob + ko),
// output rows [q0, q0 + len_out)
struct PromptGeom {
  int p0, L, nvis, ko, q0, len_out;
  int64_t rb, ob;
  bool sane;
};
__device__ __forceinline__ PromptGeom prompt_geom(const SegView& sv, const int32_t* __restrict__ cu_seqlens, int s, int total_len,
                                                  int l_out_total) {
  PromptGeom g;
  g.p0 = cu_seqlens ? cu_seqlens[s] : 0;
  g.L = cu_seqlens ? cu_seqlens[s + 1] - g.p0 : total_len;
  g.nvis = sv.n_rows(s);
  g.rb = sv.row_begin(s);
  g.ob = sv.out_begin(s);
  g.ko = min(sv.n_out(s), g.nvis);
  g.q0 = g.p0 - (int)g.rb + (int)g.ob;
  g.len_out = g.L - g.nvis + g.ko;
  g.sane = g.L >= 0 && g.nvis >= 1 && g.nvis <= g.L && g.ko >= 0 && g.p0 >= 0 && g.p0 + g.L <= total_len && g.q0 >= 0 &&
           g.q0 + g.len_out <= l_out_total;
  return g;
}

// ---- general form: one workgroup per prompt, kept ranks from idx (written by the select launch before) -------------------
static __global__ __launch_bounds__(kSpliceThreads) void splice_index_seg_kernel(
    const int64_t* __restrict__ ids, const int32_t* __restrict__ cu_seqlens, SegView sv, int n_seq, int total_len,
    int64_t visual_id, const int64_t* __restrict__ idx, const int64_t* __restrict__ src_map, const int64_t* __restrict__ pos,
    int pos_rows, const int64_t* __restrict__ mask, int64_t* __restrict__ sel, int64_t* __restrict__ new_ids,
    int64_t* __restrict__ new_pos, int64_t* __restrict__ new_mask, int32_t* __restrict__ src, int32_t* __restrict__ cu_out,
    int32_t* __restrict__ stats, int l_out) {
  extern __shared__ uint32_t bitmap[];
  const int s = blockIdx.x;
  const PromptGeom g = prompt_geom(sv, cu_seqlens, s, total_len, l_out);
  if (threadIdx.x == 0 && cu_out) {
    cu_out[s] = g.q0;
    if (s == n_seq - 1) cu_out[n_seq] = g.q0 + g.len_out;
  }
  if (!g.sane) {
    if (threadIdx.x == 0) atomicAdd(&stats[3], 1);
    return;
  }
  const SpliceSeq sq{g.p0, g.L, g.nvis, (int)g.ob, g.ko, g.q0};
  uint32_t nv, nk, nj;
  splice_index_body(sq, bitmap, ids, visual_id, idx, pos, pos_rows, total_len, mask, sel, new_ids, new_pos, new_mask, src, l_out,
                    g.q0 + g.len_out, nv, nk, nj, g.rb, src_map);
  if (threadIdx.x == 0) {
    atomicAdd(&stats[0], (int32_t)nv);
    atomicAdd(&stats[1], (int32_t)nk);
    atomicAdd(&stats[2], (int32_t)nj);
    if ((int)nv != g.nvis || (int)nk != g.len_out || (int)nj != g.ko) atomicAdd(&stats[3], 1);
  }
}

// soft top-k of one row (softtopk.hip)
int launch_soft_topk_fwd(hipStream_t st, const float* xs, int64_t b, int64_t n, int64_t k, float* ps, float* ts);

// optional extra outputs of the select-splice entries: the soft top-k of the scores the reference's eval forward also publishes
// (visual.last_combined_scores, EV/token_compression/selector_model.py:190): ps fp32 [sum N], ts fp32 [n_seg]; NULL = not wanted
struct SoftOut { float* ps; float* ts; };

// ---- one-launch form ---------------------------------------------------------------------------------------------------
// grid (ceil(max len_out / rows_per_block), n_seq), block 1024.  single != 0: one prompt, stats are stored (no memset / atomics).
template <typename T, int KPT>
__global__ __launch_bounds__(kSpliceThreads) void select_splice_small_kernel(
    const T* __restrict__ h, const float* __restrict__ scores, SegView sv, int d, const int64_t* __restrict__ src_map,
    const int64_t* __restrict__ ids, const int32_t* __restrict__ cu_seqlens, int n_seq, int total_len, int64_t visual_id,
    const T* __restrict__ embeds, const int64_t* __restrict__ pos, int pos_rows, const int64_t* __restrict__ mask,
    int64_t* __restrict__ idx, int64_t* __restrict__ sel, int64_t* __restrict__ new_ids, T* __restrict__ new_embeds,
    int64_t* __restrict__ new_pos, int64_t* __restrict__ new_mask, int32_t* __restrict__ cu_out, int32_t* __restrict__ stats,
    int l_out, int rows_per_block, int h_rows, int single, SoftOut soft) {
  constexpr int V = Elem<T>::kVec;
  constexpr int NW = kSpliceThreads / 64;
  const int s = blockIdx.y, b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const PromptGeom g = prompt_geom(sv, cu_seqlens, s, total_len, l_out);
  if (soft.ps && b == (int)gridDim.x - 1) {
    // one EXTRA workgroup per launch (single prompt, <= 4096 visual tokens, 0 < k < n: launch_select_splice): the soft top-k of the
    // same scores, with vsel_soft_topk_fwd's arithmetic (bit-identical), beside the select / splice workgroups instead of in a
    // launch of its own behind them (24 us of one CU at 2304 tokens before round 4, then 2 launches -> 1)
    __shared__ float soft_red[6][NW];
    // (rows, n and k come from the host-validated segment view, not from the prompt: a prompt whose placeholder count does not
    // match -- !g.sane, reported through stats[3] -- still gets the soft top-k of its scores instead of uninitialised memory)
    if (g.nvis >= 2 && g.ko >= 1 && g.ko < g.nvis) soft_topk_row_256x16<NW>(scores + g.rb, g.nvis, g.ko, soft.ps + g.rb, soft.ts + s, soft_red);
    return;
  }
  if (b == 0 && tid == 0 && cu_out) {
    cu_out[s] = g.q0;
    if (s == n_seq - 1) cu_out[n_seq] = g.q0 + g.len_out;
  }
  if (!g.sane) {                                    // uniform per prompt: reported, nothing written
    if (b == 0 && tid == 0) {
      if (single) { stats[0] = stats[1] = stats[2] = 0; stats[3] = 1; }
      else atomicAdd(&stats[3], 1);
    }
    return;
  }
  const int nblk = max(1, (g.len_out + rows_per_block - 1) / rows_per_block);      // active workgroups of this prompt
  if (b >= nblk) return;
  const int qb = b * rows_per_block, qe = min(g.len_out, qb + rows_per_block);     // own output rows (local to the prompt)
  const int kpb = (g.ko + nblk - 1) / nblk;
  const int jb = b * kpb, je = min(g.ko, jb + kpb);                                // own slice of idx
  const int n = g.nvis;
  const float* sc = scores + g.rb;

  __shared__ uint32_t hist[4][256];
  __shared__ uint32_t wtot[NW][2];
  __shared__ unsigned long long kept_bits[NW * KPT];
  __shared__ uint32_t wv[NW], wk[NW], wj[NW];
  __shared__ int32_t rows_src[kSelectSpliceMaxRows];

  // ---- radix select of the k-th largest key (topk_select_reg_kernel's arithmetic) --------------------------------------
  const int kpw = (n + kSpliceThreads - 1) / kSpliceThreads;      // 64-element groups per wave (<= KPT)
  const int e0 = wave * kpw * 64 + lane;
  uint32_t key[KPT];
  {
    float raw[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) raw[j] = sc[min(e0 + 64 * j, n - 1)];
    hist[tid >> 8][tid & 255] = 0;
    if (tid < kSelectSpliceMaxRows) rows_src[tid] = INT32_MAX;
#pragma unroll
    for (int j = 0; j < KPT; ++j) key[j] = (j < kpw && e0 + 64 * j < n) ? order_key(raw[j]) : 0u;
  }
  __syncthreads();
  uint32_t thr = 0, need = 0;
  if (g.ko > 0) {
    uint32_t prefix = 0, maskbits = 0, kk = (uint32_t)g.ko;
#pragma unroll 1
    for (int pass = 3; pass >= 0; --pass) {
      const int shift = 8 * pass;
      uint32_t* hp = hist[pass];
#pragma unroll
      for (int j = 0; j < KPT; ++j)
        if (j < kpw && e0 + 64 * j < n && (key[j] & maskbits) == prefix) atomicAdd(&hp[(key[j] >> shift) & 255u], 1u);
      __syncthreads();
      const u32x4 cv = *reinterpret_cast<const u32x4*>(&hp[252 - 4 * lane]);
      const uint32_t cs[4] = {cv[3], cv[2], cv[1], cv[0]};
      const uint32_t tot = cs[0] + cs[1] + cs[2] + cs[3];
      uint32_t run = wave_prefix_sum_u32(tot) - tot;
      uint32_t found = 0xffffffffu, found_kk = 0;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        if (run < kk && run + cs[bb] >= kk) {
          found = 255 - 4 * lane - bb;
          found_kk = kk - run;
        }
        run += cs[bb];
      }
      const unsigned long long who = __ballot(found != 0xffffffffu);      // exactly one lane (ko >= 1)
      const int srcl = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
      const uint32_t bin = (uint32_t)__builtin_amdgcn_readlane((int)found, srcl);
      kk = (uint32_t)__builtin_amdgcn_readlane((int)found_kk, srcl);
      prefix |= bin << shift;
      maskbits |= 0xffu << shift;
    }
    thr = prefix;
    need = kk;
  }
  // ---- kept ranks: bitmap in LDS (word w * kpw + j = elements of wave w's group j) + this workgroup's slice of idx ------
  unsigned long long bgt[KPT], beq[KPT];
  uint32_t my_gt = 0, my_eq = 0;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const bool valid = g.ko > 0 && j < kpw && e0 + 64 * j < n;
    bgt[j] = __ballot(valid && key[j] > thr);
    beq[j] = __ballot(valid && key[j] == thr);
    my_gt += __popcll(bgt[j]);
    my_eq += __popcll(beq[j]);
  }
  if (lane == 0) { wtot[wave][0] = my_gt; wtot[wave][1] = my_eq; }
  __syncthreads();
  {
    uint32_t run_gt = 0, run_eq = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w < wave) { run_gt += wtot[w][0]; run_eq += wtot[w][1]; }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      if (j < kpw) {                                                      // uniform
        const bool gt = (bgt[j] >> lane) & 1ull;
        const bool eq = (beq[j] >> lane) & 1ull;
        const uint32_t gt_before = run_gt + __popcll(bgt[j] & below), eq_before = run_eq + __popcll(beq[j] & below);
        const bool kept = gt || (eq && eq_before < need);
        const unsigned long long bk = __ballot(kept);
        if (lane == 0) kept_bits[wave * kpw + j] = bk;
        if (kept) {
          const int posn = (int)(gt_before + min(eq_before, need));
          if (posn >= jb && posn < je) idx[g.ob + posn] = (int64_t)(e0 + 64 * j);
        }
        run_gt += __popcll(bgt[j]);
        run_eq += __popcll(beq[j]);
      }
    }
  }
  __syncthreads();
  // ---- ordered scan over the prompt's positions (splice_index_body's arithmetic), own rows only ---------------------------
  uint32_t run_vis = 0, run_keep = 0, run_kv = 0;
  for (int c0 = 0; c0 < g.L; c0 += kSpliceThreads) {
    const int p = c0 + tid;
    const bool valid = p < g.L;
    const int64_t id = valid ? ids[g.p0 + p] : 0;
    const bool is_vis = valid && id == visual_id;
    const unsigned long long bvis = __ballot(is_vis);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) wv[wave] = __popcll(bvis);
    __syncthreads();
    uint32_t vis_rank = run_vis + __popcll(bvis & below), tot_vis = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { if (w < wave) vis_rank += wv[w]; tot_vis += wv[w]; }
    const bool kept_vis = is_vis && vis_rank < (uint32_t)n && ((kept_bits[vis_rank >> 6] >> (vis_rank & 63)) & 1ull);
    const bool keep = valid && (!is_vis || kept_vis);
    const unsigned long long bkeep = __ballot(keep), bkv = __ballot(kept_vis);
    if (lane == 0) { wk[wave] = __popcll(bkeep); wj[wave] = __popcll(bkv); }
    __syncthreads();
    uint32_t q = run_keep + __popcll(bkeep & below), tot_keep = 0, tot_kv = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      if (w < wave) q += wk[w];
      tot_keep += wk[w];
      tot_kv += wj[w];
    }
    if (keep && (int)q >= qb && (int)q < qe) {
      const int qo = g.q0 + (int)q;
      sel[qo] = g.p0 + p;
      new_ids[qo] = id;
      if (mask) new_mask[qo] = mask[g.p0 + p];
      for (int r = 0; r < pos_rows; ++r) new_pos[(int64_t)r * l_out + qo] = pos[(int64_t)r * total_len + g.p0 + p];
      if (kept_vis) {
        const int64_t lrow = g.rb + vis_rank;
        rows_src[(int)q - qb] = -(int32_t)(src_map ? src_map[lrow] : lrow) - 1;
      } else {
        rows_src[(int)q - qb] = g.p0 + p;
      }
    }
    run_vis += tot_vis;
    run_keep += tot_keep;
    run_kv += tot_kv;
    __syncthreads();
    if (b != 0 && (int)run_keep >= qe) break;         // uniform: later positions belong to other workgroups (block 0 counts on)
  }
  if (b == 0 && tid == 0) {
    // nblk > gridDim.x: the caller's max_len_out is below this prompt's L', so the grid has no workgroup for its last rows
    const bool bad = (int)run_vis != g.nvis || (int)run_keep != g.len_out || (int)run_kv != g.ko || nblk > (int)gridDim.x - (soft.ps ? 1 : 0);
    if (single) {
      stats[0] = (int32_t)run_vis;
      stats[1] = (int32_t)run_keep;
      stats[2] = (int32_t)run_kv;
      stats[3] = bad ? 1 : 0;
    } else {
      atomicAdd(&stats[0], (int32_t)run_vis);
      atomicAdd(&stats[1], (int32_t)run_keep);
      atomicAdd(&stats[2], (int32_t)run_kv);
      if (bad) atomicAdd(&stats[3], 1);
    }
  }
  // rows the scan did not produce (input_ids holds fewer kept positions than the descriptors promise; stats report it): marked,
  // and their embedding rows are zeros -- never an uninitialised or out-of-bounds source
  const int nrows = qe - qb;
  if (tid < nrows && rows_src[tid] == INT32_MAX) {
    sel[g.q0 + qb + tid] = -1;
    new_ids[g.q0 + qb + tid] = -1;
  }
  // ---- row copies: 4 rows x (up to 1024 x 16 B) in flight -------------------------------------------------------------
  const int vpr = d / V;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (int r0 = 0; r0 < nrows; r0 += 4) {
    const u32x4* sp[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sr = __builtin_amdgcn_readfirstlane(rows_src[min(r0 + u, nrows - 1)]);
      ok[u] = sr >= 0 ? sr < total_len : (-(int64_t)sr - 1) < h_rows;
      const T* from = sr >= 0 ? embeds + (int64_t)(ok[u] ? sr : 0) * d : h + (int64_t)(ok[u] ? -(int64_t)sr - 1 : 0) * d;
      sp[u] = reinterpret_cast<const u32x4*>(from);
    }
    for (int v0 = 0; v0 < vpr; v0 += kSpliceThreads) {
      const int v = v0 + tid, vc = min(v, vpr - 1);
      u32x4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = sp[u][vc];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r0 + u < nrows && v < vpr)
          reinterpret_cast<u32x4*>(new_embeds + (int64_t)(g.q0 + qb + r0 + u) * d)[v] = ok[u] ? x[u] : zero;
    }
  }
}

template <typename T>
static int launch_select_splice(hipStream_t st, const T* h, int d, const vsel_segments* seg, const float* scores,
                                const int64_t* l2p, const int64_t* ids, const int32_t* cu_seqlens, int64_t total_len,
                                int64_t max_len_out, int64_t visual_id, const T* embeds, const int64_t* pos, int pos_rows,
                                const int64_t* mask, int64_t* idx, int64_t* sel, int64_t* new_ids, T* new_embeds,
                                int64_t* new_pos, int64_t* new_mask, int32_t* cu_out, int32_t* src_scratch, int32_t* stats,
                                int64_t l_out, SoftOut soft) {
  const SegView sv = make_view(seg);
  const int S = (int)seg->n_seg;
  const int64_t maxn = seg->rows_per_seg;
  const bool fused = knob(VSEL_KNOB_LIS_SPLICE_FUSED) != 0 && S <= kSelectSpliceMaxSeq && maxn <= 32 * kSpliceThreads;
  // soft top-k wanted: inside the one-launch form for one prompt of <= 4096 tokens (the reference's call), else its own launches
  const bool soft_wanted = soft.ps != nullptr;
  const bool soft_inside = soft_wanted && fused && S == 1 && maxn <= 4096 && seg->k > 0 && seg->k < maxn;
  auto soft_after = [&]() -> int {
    if (!soft_wanted || soft_inside) return VSEL_OK;
    if (seg->seg_rows != nullptr) return fail(VSEL_ERR_UNSUPPORTED, "soft top-k outputs need uniform segments (one prompt in the reference)");
    if (seg->k <= 0 || seg->k >= maxn) return VSEL_OK;            // reference: assert 0 < k < n; nothing to publish
    return launch_soft_topk_fwd(st, scores, S, maxn, seg->k, soft.ps, soft.ts);
  };
  const SoftOut soft_k = soft_inside ? soft : SoftOut{nullptr, nullptr};
  if (fused) {
    const int single = S == 1;
    if (!single && hipMemsetAsync(stats, 0, 4 * sizeof(int32_t), st) != hipSuccess) return fail(VSEL_ERR_HIP, "hipMemsetAsync(stats)");
    int64_t rpb = kSelectSpliceMaxRows;
    while (rpb > 2 && S * cdiv(max_len_out, rpb) < 128) rpb >>= 1;
    const dim3 grid((unsigned)std::max<int64_t>(1, cdiv(max_len_out, rpb)) + (soft_inside ? 1u : 0u), (unsigned)S);
#define VSEL_SS_LAUNCH(KPT)                                                                                                 \
    VSEL_LAUNCH((select_splice_small_kernel<T, KPT>), grid, dim3(kSpliceThreads), 0, st, h, scores, sv, d, l2p, ids,   \
                       cu_seqlens, S, (int)total_len, visual_id, embeds, pos, pos_rows, mask, idx, sel, new_ids, new_embeds, \
                       new_pos, new_mask, cu_out, stats, (int)l_out, (int)rpb, (int)seg->total_rows, single, soft_k)
    if (maxn <= 4 * kSpliceThreads) VSEL_SS_LAUNCH(4);
    else if (maxn <= 8 * kSpliceThreads) VSEL_SS_LAUNCH(8);
    else VSEL_SS_LAUNCH(32);
#undef VSEL_SS_LAUNCH
    VSEL_AFTER_LAUNCH(st, "select_splice_small_kernel");
    return soft_after();
  }
  int rc = launch_select(st, scores, seg, idx, nullptr);
  if (rc) return rc;
  if (hipMemsetAsync(stats, 0, 4 * sizeof(int32_t), st) != hipSuccess) return fail(VSEL_ERR_HIP, "hipMemsetAsync(stats)");
  if (l_out > 0 && hipMemsetAsync(src_scratch, 0x7f, (size_t)l_out * sizeof(int32_t), st) != hipSuccess)
    return fail(VSEL_ERR_HIP, "hipMemsetAsync(src)");
  const size_t lds = (size_t)((maxn + 31) / 32) * 4 + 16;
  VSEL_LAUNCH(splice_index_seg_kernel, dim3((unsigned)S), dim3(kSpliceThreads), lds, st, ids, cu_seqlens, sv, S,
                     (int)total_len, visual_id, idx, l2p, pos, pos_rows, mask, sel, new_ids, new_pos, new_mask, src_scratch,
                     cu_out, stats, (int)l_out);
  VSEL_AFTER_LAUNCH(st, "splice_index_seg_kernel");
  if (l_out > 0) {
    const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(l_out, 4), 2048);
    VSEL_LAUNCH((splice_embed_kernel<T>), dim3(blocks), dim3(256), 0, st, embeds, h, src_scratch, (int)l_out, d,
                       (int)total_len, (int)seg->total_rows, new_embeds);
    VSEL_AFTER_LAUNCH(st, "splice_embed_kernel");
  }
  return soft_after();
}

template <typename T, typename TW>
static int scores_for_splice(hipStream_t st, const T* h, const vsel_segments* seg, const vsel_scorer* sc, char* ws,
                             const LisPlan& p, float* scores, const int64_t* p2l, const float* col_sums) {
  int rc = VSEL_OK;
  if (!col_sums && (rc = run_colsum<T>(st, h, seg, (int)sc->d, ws, p))) return rc;
  if (seg->n_seg <= knob(VSEL_KNOB_LIS_SMALL_PATH) && small_path_ok(seg, sc, p)) {
    if ((rc = run_proj_small(st, seg, sc, ws, p, col_sums))) return rc;
    return run_score_small<T>(st, h, seg, sc, ws, p, scores, p2l);
  }
  if ((rc = run_proj<TW>(st, seg, sc, ws, p, col_sums))) return rc;
  return run_score<T>(st, h, seg, sc, ws, p, scores, p2l);
}

}  // namespace vsel

using namespace vsel;

// shared argument checks of the two entry points
static int splice_args_check(const void* h, vsel_dtype hdtype, const vsel_segments* seg, const int64_t* input_ids, int64_t total_len,
                             const int32_t* cu_seqlens, int64_t max_len_out, const void* inputs_embeds, const int64_t* position_ids,
                             int64_t pos_rows, const int64_t* attention_mask, const int64_t* idx, const float* scores,
                             const int64_t* selected_indices, const int64_t* new_input_ids, const void* new_inputs_embeds,
                             const int64_t* new_position_ids, const int64_t* new_attention_mask, const int32_t* cu_seqlens_out,
                             const int32_t* src_scratch, const int32_t* stats, int64_t d, int64_t* l_out) {
  if (!h || !input_ids || !inputs_embeds || !idx || !scores || !selected_indices || !new_input_ids || !new_inputs_embeds ||
      !src_scratch || !stats)
    return fail(VSEL_ERR_INVALID, "NULL pointer");
  int rc = check_segments_impl(seg, true);
  if (rc) return rc;
  if (seg->n_seg > 65535) return fail(VSEL_ERR_UNSUPPORTED, "more than 65535 prompts");
  if (seg->n_seg > 1 && (!cu_seqlens || !cu_seqlens_out))
    return fail(VSEL_ERR_INVALID, "several prompts need cu_seqlens / cu_seqlens_out");
  if (pos_rows < 0 || pos_rows > 4 || (pos_rows > 0 && (!position_ids || !new_position_ids)))
    return fail(VSEL_ERR_INVALID, "bad position_ids arguments");
  if ((attention_mask == nullptr) != (new_attention_mask == nullptr)) return fail(VSEL_ERR_INVALID, "attention_mask in/out mismatch");
  if (hdtype != VSEL_BF16 && hdtype != VSEL_F32) return fail(VSEL_ERR_INVALID, "bad dtype");
  *l_out = total_len - seg->total_rows + seg->total_out;
  if (total_len < 1 || total_len >= (1ll << 31) || seg->total_rows > total_len || seg->total_rows >= (1ll << 31) || *l_out < 1 ||
      max_len_out < 1 || max_len_out > *l_out)
    return fail(VSEL_ERR_INVALID, "bad sizes (T=%lld, visual=%lld, kept=%lld, max L'=%lld)", (long long)total_len,
                (long long)seg->total_rows, (long long)seg->total_out, (long long)max_len_out);
  if (seg->n_seg == 1 && max_len_out != *l_out)
    return fail(VSEL_ERR_INVALID, "one prompt: max_len_out (%lld) must be its L' (%lld)", (long long)max_len_out, (long long)*l_out);
  if (seg->rows_per_seg > kMaxVisualBits) return fail(VSEL_ERR_UNSUPPORTED, "more than %d visual tokens in one prompt", kMaxVisualBits);
  if (((uintptr_t)h | (uintptr_t)inputs_embeds | (uintptr_t)new_inputs_embeds) & 15)
    return fail(VSEL_ERR_INVALID, "h / embeddings must be 16-byte aligned");
  const int vec = hdtype == VSEL_BF16 ? 8 : 4;
  if (d < vec || d % vec) return fail(VSEL_ERR_UNSUPPORTED, "token width must be a multiple of %d", vec);
  return VSEL_OK;
}

static int select_splice_dispatch(hipStream_t st, const void* h, vsel_dtype hdtype, int64_t d, const vsel_segments* seg,
                                  const float* scores, const int64_t* l2p, const int64_t* input_ids, const int32_t* cu_seqlens,
                                  int64_t total_len, int64_t max_len_out, int64_t visual_token_id, const void* inputs_embeds,
                                  const int64_t* position_ids, int64_t pos_rows, const int64_t* attention_mask, int64_t* idx,
                                  int64_t* selected_indices, int64_t* new_input_ids, void* new_inputs_embeds,
                                  int64_t* new_position_ids, int64_t* new_attention_mask, int32_t* cu_seqlens_out,
                                  int32_t* src_scratch, int32_t* stats, int64_t l_out, SoftOut soft) {
  if (hdtype == VSEL_BF16)
    return launch_select_splice<bf16_t>(st, (const bf16_t*)h, (int)d, seg, scores, l2p, input_ids, cu_seqlens, total_len,
                                        max_len_out, visual_token_id, (const bf16_t*)inputs_embeds, position_ids, (int)pos_rows,
                                        attention_mask, idx, selected_indices, new_input_ids, (bf16_t*)new_inputs_embeds,
                                        new_position_ids, new_attention_mask, cu_seqlens_out, src_scratch, stats, l_out, soft);
  return launch_select_splice<float>(st, (const float*)h, (int)d, seg, scores, l2p, input_ids, cu_seqlens, total_len, max_len_out,
                                     visual_token_id, (const float*)inputs_embeds, position_ids, (int)pos_rows, attention_mask, idx,
                                     selected_indices, new_input_ids, (float*)new_inputs_embeds, new_position_ids,
                                     new_attention_mask, cu_seqlens_out, src_scratch, stats, l_out, soft);
}

extern "C" int vsel_lis_select_splice(void* stream, const void* h, vsel_dtype hdtype, const vsel_segments* seg,
                                      const vsel_scorer* scorer, void* workspace, size_t workspace_bytes, const float* col_sums,
                                      const int64_t* logical_to_physical, const int64_t* physical_to_logical,
                                      const int64_t* input_ids, int64_t total_len, const int32_t* cu_seqlens,
                                      int64_t max_len_out, int64_t visual_token_id, const void* inputs_embeds,
                                      const int64_t* position_ids, int64_t pos_rows, const int64_t* attention_mask, int64_t* idx,
                                      float* scores, int64_t* selected_indices, int64_t* new_input_ids, void* new_inputs_embeds,
                                      int64_t* new_position_ids, int64_t* new_attention_mask, int32_t* cu_seqlens_out,
                                      int32_t* src_scratch, int32_t* stats, float* soft_ps, float* soft_ts) {
  if (!scorer) return fail(VSEL_ERR_INVALID, "scorer is NULL");
  if ((soft_ps == nullptr) != (soft_ts == nullptr)) return fail(VSEL_ERR_INVALID, "give both soft_ps and soft_ts or neither");
  int64_t l_out = 0;
  int rc = splice_args_check(h, hdtype, seg, input_ids, total_len, cu_seqlens, max_len_out, inputs_embeds, position_ids, pos_rows,
                             attention_mask, idx, scores, selected_indices, new_input_ids, new_inputs_embeds, new_position_ids,
                             new_attention_mask, cu_seqlens_out, src_scratch, stats, scorer->d, &l_out);
  if (rc) return rc;
  if ((logical_to_physical == nullptr) != (physical_to_logical == nullptr)) return fail(VSEL_ERR_INVALID, "give both row maps or neither");
  if ((rc = check_scorer(scorer, hdtype))) return rc;
  const LisPlan p = make_plan(seg->n_seg, seg->rows_per_seg, scorer->d, scorer->hd);
  if (!workspace || workspace_bytes < p.total) return fail(VSEL_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, p.total);
  if (((uintptr_t)workspace | (uintptr_t)scorer->wq | (uintptr_t)scorer->wk | (uintptr_t)col_sums) & 15)
    return fail(VSEL_ERR_INVALID, "workspace / weights / col_sums must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  char* ws = (char*)workspace;
  const int64_t* p2l = physical_to_logical;
  if (hdtype == VSEL_BF16)
    rc = scorer->wdtype == VSEL_BF16 ? scores_for_splice<bf16_t, bf16_t>(st, (const bf16_t*)h, seg, scorer, ws, p, scores, p2l, col_sums)
                                     : scores_for_splice<bf16_t, float>(st, (const bf16_t*)h, seg, scorer, ws, p, scores, p2l, col_sums);
  else
    rc = scorer->wdtype == VSEL_BF16 ? scores_for_splice<float, bf16_t>(st, (const float*)h, seg, scorer, ws, p, scores, p2l, col_sums)
                                     : scores_for_splice<float, float>(st, (const float*)h, seg, scorer, ws, p, scores, p2l, col_sums);
  if (rc) return rc;
  return select_splice_dispatch(st, h, hdtype, scorer->d, seg, scores, logical_to_physical, input_ids, cu_seqlens, total_len,
                                max_len_out, visual_token_id, inputs_embeds, position_ids, pos_rows, attention_mask, idx,
                                selected_indices, new_input_ids, new_inputs_embeds, new_position_ids, new_attention_mask,
                                cu_seqlens_out, src_scratch, stats, l_out, SoftOut{soft_ps, soft_ts});
}

extern "C" int vsel_topk_select_splice(void* stream, const void* h, vsel_dtype hdtype, int64_t d, const vsel_segments* seg,
                                       const float* scores, const int64_t* logical_to_physical, const int64_t* input_ids,
                                       int64_t total_len, const int32_t* cu_seqlens, int64_t max_len_out, int64_t visual_token_id,
                                       const void* inputs_embeds, const int64_t* position_ids, int64_t pos_rows,
                                       const int64_t* attention_mask, int64_t* idx, int64_t* selected_indices,
                                       int64_t* new_input_ids, void* new_inputs_embeds, int64_t* new_position_ids,
                                       int64_t* new_attention_mask, int32_t* cu_seqlens_out, int32_t* src_scratch, int32_t* stats,
                                       float* soft_ps, float* soft_ts) {
  if ((soft_ps == nullptr) != (soft_ts == nullptr)) return fail(VSEL_ERR_INVALID, "give both soft_ps and soft_ts or neither");
  int64_t l_out = 0;
  int rc = splice_args_check(h, hdtype, seg, input_ids, total_len, cu_seqlens, max_len_out, inputs_embeds, position_ids, pos_rows,
                             attention_mask, idx, scores, selected_indices, new_input_ids, new_inputs_embeds, new_position_ids,
                             new_attention_mask, cu_seqlens_out, src_scratch, stats, d, &l_out);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  return select_splice_dispatch(st, h, hdtype, d, seg, scores, logical_to_physical, input_ids, cu_seqlens, total_len, max_len_out,
                                visual_token_id, inputs_embeds, position_ids, pos_rows, attention_mask, idx, selected_indices,
                                new_input_ids, new_inputs_embeds, new_position_ids, new_attention_mask, cu_seqlens_out, src_scratch,
                                stats, l_out, SoftOut{soft_ps, soft_ts});
}
