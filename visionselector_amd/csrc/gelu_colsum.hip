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

// Rows per workgroup: 128 when that already gives several rounds of workgroups, halved (down to 16) otherwise -- a 128-row
// workgroup runs ~100 us of erf, so at 36 k rows (2880 workgroups on 2048 slots) the tail round cost 15-30 % (207 vs 159 us for
// torch's GELU).  The column sums are a fixed function of (rows per segment, columns), not bit-identical across batch shapes; the
// bit-exact path is the two-sweep one.
constexpr int kGeluRowsMax = 128;
inline int gelu_rows_per_wg(const vsel_segments* seg, int64_t cols, int vec) {
  int rows = kGeluRowsMax;
  while (rows > 16 && seg->n_seg * cdiv(seg->rows_per_seg, rows) * cdiv(cols, 64 * vec) < 6144) rows >>= 1;
  return rows;
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_colsum_kernel(const T* __restrict__ x, SegView sv, int c, int row_splits, int rows_per_wg,
                                                          T* __restrict__ y, float* __restrict__ partial) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.z, rs = blockIdx.y;
  const int col = (blockIdx.x * 64 + lane) * V;
  const int n = sv.n_rows(s);
  const int64_t r0 = sv.row_begin(s);
  const int rb = rs * rows_per_wg;
  const int re = min(n, rb + rows_per_wg);
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
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
      if constexpr (sizeof(T) == 2) {
        // round once: the stored bf16 bits are also what the column sums add up
        uint32_t bits[V];
#pragma unroll
        for (int i = 0; i < V; ++i) bits[i] = f32_to_bf16_bits(cur[i]);
        u32x4 pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[i] = bits[2 * i] | (bits[2 * i + 1] << 16);
        *reinterpret_cast<u32x4*>(y + base + (int64_t)r * c) = pk;
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] += __uint_as_float(bits[i] << 16);
      } else {
        store_vec(y + base + (int64_t)r * c, cur);
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] += cur[i];
      }
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

// col_sums[s][col] = sum_rs partial[s][rs][col] in a FIXED two-level order: 16 interleaved groups (rs = g, g + 16, ...) summed in
// rs order, then the 16 group sums in g order.  grid (ceil(c / 64), S), block 1024 = 64 columns x 16 groups.  (The first version --
// one thread per column walking all row splits -- took 413 us at 147 k rows, 1152 splits: longer than half the GELU itself.)
static __global__ __launch_bounds__(1024) void gelu_colsum_finish_kernel(const float* __restrict__ partial, int c, int row_splits,
                                                                         float* __restrict__ col_sums) {
  const int s = blockIdx.y;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  __shared__ float red[16][64];
  float acc = 0.f;
  if (col < c) {
    const float* p = partial + (int64_t)s * row_splits * c + col;
    int rs = g;
    for (; rs + 7 * 16 < row_splits; rs += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(rs + 16 * u) * c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; rs < row_splits; rs += 16) acc += p[(int64_t)rs * c];
  }
  red[g][lane] = acc;
  __syncthreads();
  if (g == 0 && col < c) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][lane];
    col_sums[(int64_t)s * c + col] = t;
  }
}

}  // namespace vsel

using namespace vsel;

extern "C" size_t vsel_gelu_colsum_workspace_bytes(const vsel_segments* seg, int64_t cols) {
  if (!seg || seg->n_seg < 1 || seg->rows_per_seg < 1 || cols < 1) return 0;
  // the partials of the chunking the launch will use: rows per workgroup depend on the vector width (dtype), which this query
  // does not take, so the larger of the two (bf16: 8 elements per lane, fp32: 4) -- not the 16-row worst case, which was 8x too
  // large exactly where the fusion is on (147 456 x 5120: 23.6 MB instead of 188 MB)
  const int rows = gelu_rows_per_wg(seg, cols, 8) < gelu_rows_per_wg(seg, cols, 4) ? gelu_rows_per_wg(seg, cols, 8)
                                                                                  : gelu_rows_per_wg(seg, cols, 4);
  return (size_t)seg->n_seg * (size_t)cdiv(seg->rows_per_seg, rows) * (size_t)cols * sizeof(float);
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
  const int rows_per_wg = gelu_rows_per_wg(seg, cols, vec);
  const int row_splits = (int)cdiv(seg->rows_per_seg, rows_per_wg);
  float* partial = (float*)workspace;
  if (dtype == VSEL_BF16)
    hipLaunchKernelGGL((gelu_colsum_kernel<bf16_t>), dim3((unsigned)cdiv(cols, 64 * 8), row_splits, S), dim3(256), 0, st,
                       (const bf16_t*)x, sv, (int)cols, row_splits, rows_per_wg, (bf16_t*)y, partial);
  else
    hipLaunchKernelGGL((gelu_colsum_kernel<float>), dim3((unsigned)cdiv(cols, 64 * 4), row_splits, S), dim3(256), 0, st,
                       (const float*)x, sv, (int)cols, row_splits, rows_per_wg, (float*)y, partial);
  VSEL_AFTER_LAUNCH(st, "gelu_colsum_kernel");
  hipLaunchKernelGGL(gelu_colsum_finish_kernel, dim3((unsigned)cdiv(cols, 64), S), dim3(1024), 0, st, partial, (int)cols, row_splits,
                     col_sums);
  VSEL_AFTER_LAUNCH(st, "gelu_colsum_finish_kernel");
  return VSEL_OK;
}
