// Fused GELU + per-segment column sums (producer-side fusion for the patch merger).
//
// The merger that produces the visual tokens is  ln_q -> Linear(4C, 4C) -> GELU -> Linear(4C, D)
// (Qwen2_5_VLPatchMerger, qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:148-161; RicePatchMerger,
// llava-ov-15/llavaonevision1_5/modeling_llavaonevision1_5.py:255-268).  The LIS needs mean_rows(H) before it can score
// (its first HBM sweep over H).  The last Linear is linear:  sum_rows(H) = sum_rows(G) W2^T + N b2  with G = GELU(...).
// This kernel replaces the GELU launch: it writes G exactly as nn.GELU() would (erf form, fp32 math, rounded to the
// activation dtype) and accumulates the column sums of the ROUNDED G on the way -- no extra HBM traffic -- so that
// vsel_lis_select_presummed can skip sweep 1.  Deterministic: fixed 128-row chunks, then a fixed-order finish.
#include "common.h"
#include "lis_kernels.h"

namespace vsel {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

constexpr int kGeluRows = 128;     // rows per workgroup (32-row chunks measured 2.3x slower at 147 k rows: tools/exp_merger_fusion.py)

template <typename T>
__global__ __launch_bounds__(256) void gelu_colsum_kernel(const T* __restrict__ x, SegView sv, int c, int row_splits,
                                                          T* __restrict__ y, float* __restrict__ partial) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.z, rs = blockIdx.y;
  const int col = (blockIdx.x * 64 + lane) * V;
  const int n = sv.n_rows(s);
  const int64_t r0 = sv.row_begin(s);
  const int rb = rs * kGeluRows;
  const int re = min(n, rb + kGeluRows);
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
  auto rounded = [](float v) -> float {
    if constexpr (sizeof(T) == 2) return bf16_to_f32(f32_to_bf16_bits(v));
    else return v;
  };
  if (col < c) {
    const int64_t base = r0 * (int64_t)c + col;
    int r = rb + wave;
    // software-pipelined: the next two rows' loads are in flight while this row's erf runs (the erf body is emitted once)
    float cur[V], nx1[V], nx2[V];
    if (r < re) load_vec(x + base + (int64_t)r * c, cur);
    if (r + 4 < re) load_vec(x + base + (int64_t)(r + 4) * c, nx1);
    for (; r < re; r += 4) {
      if (r + 8 < re) load_vec(x + base + (int64_t)(r + 8) * c, nx2);
#pragma unroll
      for (int i = 0; i < V; ++i) cur[i] = gelu_erf(cur[i]);
      store_vec(y + base + (int64_t)r * c, cur);
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] += rounded(cur[i]);
#pragma unroll
      for (int i = 0; i < V; ++i) { cur[i] = nx1[i]; nx1[i] = nx2[i]; }
    }
  }
  __shared__ float red[4][64][V + 1];
#pragma unroll
  for (int i = 0; i < V; ++i) red[wave][lane][i] = acc[i];
  __syncthreads();
  if (wave == 0 && col < c) {
    float* dst = partial + ((int64_t)(s * row_splits + rs) * c + col);
#pragma unroll
    for (int i = 0; i < V; ++i) dst[i] = (red[0][lane][i] + red[1][lane][i]) + (red[2][lane][i] + red[3][lane][i]);
  }
}

static __global__ __launch_bounds__(256) void gelu_colsum_finish_kernel(const float* __restrict__ partial, int c, int row_splits,
                                                                        float* __restrict__ col_sums) {
  const int s = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= c) return;
  const float* p = partial + (int64_t)s * row_splits * c + col;
  float acc = 0.f;
  for (int rs = 0; rs < row_splits; ++rs) acc += p[(int64_t)rs * c];
  col_sums[(int64_t)s * c + col] = acc;
}

}  // namespace vsel

using namespace vsel;

extern "C" size_t vsel_gelu_colsum_workspace_bytes(const vsel_segments* seg, int64_t cols) {
  if (!seg || seg->n_seg < 1 || seg->rows_per_seg < 1 || cols < 1) return 0;
  return (size_t)seg->n_seg * (size_t)cdiv(seg->rows_per_seg, kGeluRows) * (size_t)cols * sizeof(float);
}

extern "C" int vsel_gelu_colsum(void* stream, const void* x, vsel_dtype dtype, const vsel_segments* seg, int64_t cols, void* y,
                                float* col_sums, void* workspace, size_t workspace_bytes) {
  if (!x || !y || !col_sums || !workspace) return fail(VSEL_ERR_INVALID, "NULL pointer");
  int rc = check_segments_impl(seg, false);
  if (rc) return rc;
  if (dtype != VSEL_BF16 && dtype != VSEL_F32) return fail(VSEL_ERR_INVALID, "bad dtype");
  const int vec = dtype == VSEL_BF16 ? 8 : 4;
  if (cols < vec || cols % vec) return fail(VSEL_ERR_UNSUPPORTED, "cols must be a multiple of %d", vec);
  if (((uintptr_t)x | (uintptr_t)y) & 15) return fail(VSEL_ERR_INVALID, "x / y must be 16-byte aligned");
  if (workspace_bytes < vsel_gelu_colsum_workspace_bytes(seg, cols)) return fail(VSEL_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  const SegView sv = make_view(seg);
  const int S = (int)seg->n_seg;
  const int row_splits = (int)cdiv(seg->rows_per_seg, kGeluRows);
  float* partial = (float*)workspace;
  if (dtype == VSEL_BF16)
    hipLaunchKernelGGL((gelu_colsum_kernel<bf16_t>), dim3((unsigned)cdiv(cols, 64 * 8), row_splits, S), dim3(256), 0, st,
                       (const bf16_t*)x, sv, (int)cols, row_splits, (bf16_t*)y, partial);
  else
    hipLaunchKernelGGL((gelu_colsum_kernel<float>), dim3((unsigned)cdiv(cols, 64 * 4), row_splits, S), dim3(256), 0, st,
                       (const float*)x, sv, (int)cols, row_splits, (float*)y, partial);
  VSEL_AFTER_LAUNCH(st, "gelu_colsum_kernel");
  hipLaunchKernelGGL(gelu_colsum_finish_kernel, dim3((unsigned)cdiv(cols, 256), S), dim3(256), 0, st, partial, (int)cols, row_splits,
                     col_sums);
  VSEL_AFTER_LAUNCH(st, "gelu_colsum_finish_kernel");
  return VSEL_OK;
}
