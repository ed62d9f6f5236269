// Fused GELU + per-segment column sums (producer-side fusion for the patch merger).
//
// The merger that produces the visual tokens is  ln_q -> Linear(4C, 4C) -> GELU -> Linear(4C, D)
// (Qwen2_5_VLPatchMerger, qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:148-161; RicePatchMerger,
// llava-ov-15/llavaonevision1_5/modeling_llavaonevision1_5.py:255-268).  The LIS needs mean_rows(H) before it can score
// (its first HBM sweep over H).  The last Linear is linear:  sum_rows(H) = sum_rows(G) W2^T + N b2  with G = GELU(...).
// This kernel replaces the GELU launch: it writes G exactly as nn.GELU() would (erf form, fp32 math, rounded to the
// activation dtype) and accumulates the column sums of the ROUNDED G on the way -- no extra HBM traffic -- so that
// vsel_lis_select_presummed can skip sweep 1.  Deterministic: fixed row chunks per workgroup, then a fixed-order finish.
#include "common.h"
#include "lis_kernels.h"

namespace vsel {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Rows per workgroup: 128 when that already gives several rounds of workgroups, halved (down to 16) otherwise -- a 128-row
// workgroup runs ~100 us of erf, so at 36 k rows (2880 workgroups on 2048 slots) the tail round cost 15-30 % (207 vs 159 us for
// torch's GELU).  The column sums are a fixed function of (rows per segment, columns), not bit-identical across batch shapes; the
// bit-exact path is the two-sweep one.  (Round 3 also tried ONE running sum per workgroup over statically strided 16-row chunks
// -- a bounded number of partials whatever the row count: same time at 18 k tokens, 8 % slower at 147 k and 295 k (1458 vs
// 1355 us): the partial traffic was never the cost, and contiguous rows per workgroup stream better.  Not kept.)
constexpr int kGeluRowsMax = 128;
inline int gelu_rows_per_wg(const vsel_segments* seg, int64_t cols, int vec) {
  int rows = kGeluRowsMax;
  while (rows > 16 && seg->n_seg * cdiv(seg->rows_per_seg, rows) * cdiv(cols, 64 * vec) < 6144) rows >>= 1;
  return rows;
}

// NT: the activation is far larger than the caches and both x and y are touched once here -> non-temporal loads and stores
// (chosen from 384 MB of activation up, as the LIS sweeps do): 1355 vs 1402 us (torch) at 294 912 x 5120, before 1.07x torch's.
// SUMS = false: the same streaming GELU without the column sums (col_sums == NULL at the entry point): the merger's GELU as this
// library would run it anyway -- what bench.py subtracts when it charges the LIS for the sums (not torch's slower kernel).
template <typename T, bool NT, bool SUMS = true>
__global__ __launch_bounds__(256) void gelu_colsum_kernel(const T* __restrict__ x, SegView sv, int c, int row_splits, int rows_per_wg,
                                                          T* __restrict__ y, float* __restrict__ partial) {
  constexpr int V = Elem<T>::kVec;
  typedef typename std::conditional<sizeof(T) == 2, u32x4, f32x4>::type raw_t;
  auto ld = [](const T* p) -> raw_t {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(p));
    else return *reinterpret_cast<const raw_t*>(p);
  };
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
    // software-pipelined on the RAW 16-byte vectors (unpacked only when consumed, so no wait is forced at the load): the next
    // two rows are in flight while this row's erf runs
    raw_t cur, nx1, nx2;
    if (r < re) cur = ld(x + base + (int64_t)r * c);
    if (r + 4 < re) nx1 = ld(x + base + (int64_t)(r + 4) * c);
    for (; r < re; r += 4) {
      if (r + 8 < re) nx2 = ld(x + base + (int64_t)(r + 8) * c);
      float v[V];
      if constexpr (sizeof(T) == 2) {
        unpack_vec<T>(cur, v);
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = cur[i];
      }
#pragma unroll
      for (int i = 0; i < V; ++i) v[i] = gelu_erf(v[i]);
      if constexpr (sizeof(T) == 2) {
        // round once: the stored bf16 bits are also what the column sums add up
        // (v_cvt_pk_bf16_f32: round to nearest even, f32_to_bf16_bits' bits on every finite value and infinity -- one instruction per
        // pair instead of ~11; a NaN comes out as a quiet NaN of the input's sign instead of the canonical 0x7fc0)
        u32x4 pk;
        uint32_t bits[V];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
          bits[2 * i] = pk[i] & 0xffffu;
          bits[2 * i + 1] = pk[i] >> 16;
        }
        if constexpr (NT) __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(y + base + (int64_t)r * c));
        else *reinterpret_cast<u32x4*>(y + base + (int64_t)r * c) = pk;
#pragma unroll
        for (int i = 0; i < V; ++i) if constexpr (SUMS) acc[i] += __uint_as_float(bits[i] << 16);
      } else {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(y + base + (int64_t)r * c));
        else *reinterpret_cast<f32x4*>(y + base + (int64_t)r * c) = o;
#pragma unroll
        for (int i = 0; i < V; ++i) if constexpr (SUMS) acc[i] += v[i];
      }
      cur = nx1;
      nx1 = nx2;
    }
  }
  if constexpr (SUMS) {
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


// -------------------------------------------------------------------------------------------------------------------------
// sum_rows(H) from sum_rows(G) for H = G W^T + b (the merger's last Linear; W [Cout, Cin] row-major as stored, bias [Cout]):
//   out[s][j] = sum_i in[s][i] W[j][i] + N_s b[j]
// on the STORED (bf16) weight with fp32 accumulation -- torch's route was weight.float().t() (a 73 MB fp32 copy per call at 7B)
// + an fp32 addmm.  Few segments: one wave per output row, the weight row (Cin x 2 B, contiguous) read once with every load of
// a batch in flight, fp32 FMA against the <= kLinSmallSeg input rows, fixed-order lane reduction.  Many segments: the scorer's
// own bf16x3 MFMA projection (proj_bf16x3.h: planes of in / N_s, gemm_nt, slabs) and a finish that adds the bias and scales
// back by N_s.  Deterministic either way.
// -------------------------------------------------------------------------------------------------------------------------
constexpr int kLinSmallSeg = 8;

template <typename TW>
__global__ __launch_bounds__(256) void colsum_linear_small_kernel(const float* __restrict__ in, const TW* __restrict__ w,
                                                                  const TW* __restrict__ bias, SegView sv, int S, int cin, int cout,
                                                                  float* __restrict__ out) {
  constexpr int V = Elem<TW>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  if (j >= cout) return;
  const TW* wr = w + (int64_t)j * cin;
  float acc[kLinSmallSeg];
#pragma unroll
  for (int s = 0; s < kLinSmallSeg; ++s) acc[s] = 0.f;
  constexpr int U = 5;                                   // weight vectors in flight per lane (5120 / (64 x 8) = 10 = 2 batches)
  for (int c0 = lane * V; c0 < cin; c0 += 64 * V * U) {
    float wv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) load_vec(wr + min(c0 + 64 * V * u, cin - V), wv[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 64 * V * u;
      if (c < cin) {
#pragma unroll
        for (int s = 0; s < kLinSmallSeg; ++s) {
          if (s < S) {
            const float* g = in + (int64_t)s * cin + c;
#pragma unroll
            for (int q = 0; q < V; q += 4) {
              const f32x4 gv = *reinterpret_cast<const f32x4*>(g + q);
              acc[s] = fmaf(wv[u][q], gv[0], acc[s]);
              acc[s] = fmaf(wv[u][q + 1], gv[1], acc[s]);
              acc[s] = fmaf(wv[u][q + 2], gv[2], acc[s]);
              acc[s] = fmaf(wv[u][q + 3], gv[3], acc[s]);
            }
          }
        }
      }
    }
  }
  const float b = bias ? load_elem(bias + j) : 0.f;
#pragma unroll
  for (int s = 0; s < kLinSmallSeg; ++s) {
    if (s < S) {
      const float t = wave_sum(acc[s]);
      if (lane == 0) out[(int64_t)s * cout + j] = t + (float)sv.n_rows(s) * b;
    }
  }
}

// out[m][n] = (sum_ks part[ks][n][m] + b[n]) N_m   (the planes held in / N_m).  grid (ceil(m_pad/32), ceil(N/8)), block 256.
static __global__ __launch_bounds__(256) void colsum_linear_finish_kernel(const float* __restrict__ part, int KS, int M, int N, int m_pad,
                                                                          const uint16_t* __restrict__ bias, SegView sv,
                                                                          float* __restrict__ out) {
  const int m = blockIdx.x * 32 + (threadIdx.x & 31);
  const int n = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (m < M && n < N) {
    const float v = strided_sum(part + (int64_t)n * m_pad + m, KS, (int64_t)N * m_pad) + (bias ? bf16_to_f32(bias[n]) : 0.f);
    out[(int64_t)m * N + n] = v * (float)sv.n_rows(m);
  }
}

struct LinPlan { int ks, kslice, m_pad; size_t off_planes, off_slabs, total; };
static LinPlan lin_plan(int64_t S, int64_t cin, int64_t cout) {
  LinPlan p{};
  p.kslice = (int)std::min<int64_t>(kSliceNT, cdiv(cin, 16) * 16);
  p.ks = (int)cdiv(cin, p.kslice);
  p.m_pad = 32 * (int)cdiv(S, 32);
  p.off_planes = 0;
  p.off_slabs = align_up((size_t)3 * p.m_pad * cin * sizeof(uint16_t), 256);
  p.total = p.off_slabs + align_up((size_t)p.ks * cout * p.m_pad * sizeof(float), 256);
  return p;
}

}  // namespace vsel

using namespace vsel;

extern "C" size_t vsel_gelu_colsum_workspace_bytes(const vsel_segments* seg, int64_t cols) {
  if (!seg || seg->n_seg < 1 || seg->rows_per_seg < 1 || cols < 1) return 0;
  // the partials of the chunking the launch will use: rows per workgroup depend on the vector width (dtype), which this query
  // does not take, so the larger of the two (bf16: 8 elements per lane, fp32: 4) -- not the 16-row worst case, which was 8x too
  // large exactly where the fusion is on (147 456 x 5120: 23.6 MB instead of 188 MB)
  const int rows = std::min(gelu_rows_per_wg(seg, cols, 8), gelu_rows_per_wg(seg, cols, 4));
  return (size_t)seg->n_seg * (size_t)cdiv(seg->rows_per_seg, rows) * (size_t)cols * sizeof(float);
}

extern "C" int vsel_gelu_colsum(void* stream, const void* x, vsel_dtype dtype, const vsel_segments* seg, int64_t cols, void* y,
                                float* col_sums, void* workspace, size_t workspace_bytes) {
  if (!x || !y || (col_sums && !workspace)) return fail(VSEL_ERR_INVALID, "NULL pointer");
  int rc = check_segments_impl(seg, false);
  if (rc) return rc;
  if (dtype != VSEL_BF16 && dtype != VSEL_F32) return fail(VSEL_ERR_INVALID, "bad dtype");
  const int vec = dtype == VSEL_BF16 ? 8 : 4;
  if (cols < vec || cols % vec) return fail(VSEL_ERR_UNSUPPORTED, "cols must be a multiple of %d", vec);
  if (((uintptr_t)x | (uintptr_t)y) & 15) return fail(VSEL_ERR_INVALID, "x / y must be 16-byte aligned");
  if (col_sums && workspace_bytes < vsel_gelu_colsum_workspace_bytes(seg, cols)) return fail(VSEL_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  const SegView sv = make_view(seg);
  const int S = (int)seg->n_seg;
  const int rows_per_wg = gelu_rows_per_wg(seg, cols, vec);
  const int row_splits = (int)cdiv(seg->rows_per_seg, rows_per_wg);
  float* partial = (float*)workspace;
  const bool nt = seg->total_rows * cols * (dtype == VSEL_BF16 ? 2 : 4) >= kStreamBytes;   // (from 16 / 64 MB up: no difference)
  const dim3 grid((unsigned)cdiv(cols, 64 * vec), row_splits, S);
#define VSEL_GELU_LAUNCH(T, NTV, SUMSV)                                                                                          \
  VSEL_LAUNCH((gelu_colsum_kernel<T, NTV, SUMSV>), grid, dim3(256), 0, st, (const T*)x, sv, (int)cols, row_splits, rows_per_wg, \
                     (T*)y, partial)
  const bool sums = col_sums != nullptr;
  if (dtype == VSEL_BF16) {
    if (sums) { if (nt) VSEL_GELU_LAUNCH(bf16_t, true, true); else VSEL_GELU_LAUNCH(bf16_t, false, true); }
    else { if (nt) VSEL_GELU_LAUNCH(bf16_t, true, false); else VSEL_GELU_LAUNCH(bf16_t, false, false); }
  } else {
    if (sums) { if (nt) VSEL_GELU_LAUNCH(float, true, true); else VSEL_GELU_LAUNCH(float, false, true); }
    else { if (nt) VSEL_GELU_LAUNCH(float, true, false); else VSEL_GELU_LAUNCH(float, false, false); }
  }
#undef VSEL_GELU_LAUNCH
  VSEL_AFTER_LAUNCH(st, sums ? "gelu_colsum_kernel" : "gelu_kernel");
  if (!sums) return VSEL_OK;
  VSEL_LAUNCH(gelu_colsum_finish_kernel, dim3((unsigned)cdiv(cols, 64), S), dim3(1024), 0, st, partial, (int)cols, row_splits,
                     col_sums);
  VSEL_AFTER_LAUNCH(st, "gelu_colsum_finish_kernel");
  return VSEL_OK;
}

extern "C" size_t vsel_colsum_linear_workspace_bytes(int64_t n_seg, int64_t cin, int64_t cout) {
  if (n_seg < 1 || cin < 1 || cout < 1) return 0;
  return n_seg <= kLinSmallSeg ? 16 : lin_plan(n_seg, cin, cout).total;
}

extern "C" int vsel_colsum_linear(void* stream, const float* col_sums_in, const vsel_segments* seg, const void* weight,
                                  const void* bias, vsel_dtype wdtype, int64_t cin, int64_t cout, float* col_sums_out,
                                  void* workspace, size_t workspace_bytes) {
  if (!col_sums_in || !weight || !col_sums_out) return fail(VSEL_ERR_INVALID, "NULL pointer");
  int rc = check_segments_impl(seg, false);
  if (rc) return rc;
  if (wdtype != VSEL_BF16 && wdtype != VSEL_F32) return fail(VSEL_ERR_INVALID, "bad dtype");
  const int vec = wdtype == VSEL_BF16 ? 8 : 4;
  if (cin < vec || cin % vec || cout < 1) return fail(VSEL_ERR_UNSUPPORTED, "Cin must be a multiple of %d", vec);
  if (((uintptr_t)col_sums_in | (uintptr_t)weight | (uintptr_t)col_sums_out) & 15)
    return fail(VSEL_ERR_INVALID, "col_sums / weight must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  const SegView sv = make_view(seg);
  const int S = (int)seg->n_seg;
  const bool mfma = S > kLinSmallSeg && wdtype == VSEL_BF16 && cin % 16 == 0;
  if (!mfma) {
    // <= kLinSmallSeg segments per launch (fp32 weights with many segments take several launches: not a path the models use)
    for (int s0 = 0; s0 < S; s0 += kLinSmallSeg) {
      const int sc = std::min(kLinSmallSeg, S - s0);
      SegView svc = sv;
      if (seg->seg_rows) svc.seg_rows = seg->seg_rows + s0;          // n_rows(s) of the chunk's segments
      const dim3 grid((unsigned)cdiv(cout, 4));
      if (wdtype == VSEL_BF16)
        VSEL_LAUNCH((colsum_linear_small_kernel<bf16_t>), grid, dim3(256), 0, st, col_sums_in + (int64_t)s0 * cin,
                           (const bf16_t*)weight, (const bf16_t*)bias, svc, sc, (int)cin, (int)cout, col_sums_out + (int64_t)s0 * cout);
      else
        VSEL_LAUNCH((colsum_linear_small_kernel<float>), grid, dim3(256), 0, st, col_sums_in + (int64_t)s0 * cin,
                           (const float*)weight, (const float*)bias, svc, sc, (int)cin, (int)cout, col_sums_out + (int64_t)s0 * cout);
      VSEL_AFTER_LAUNCH(st, "colsum_linear_small_kernel");
    }
    return VSEL_OK;
  }
  const LinPlan p = lin_plan(S, cin, cout);
  if (!workspace || workspace_bytes < p.total) return fail(VSEL_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, p.total);
  if ((uintptr_t)workspace & 15) return fail(VSEL_ERR_INVALID, "workspace must be 16-byte aligned");
  uint16_t* planes = (uint16_t*)((char*)workspace + p.off_planes);
  float* slabs = (float*)((char*)workspace + p.off_slabs);
  const unsigned mtiles = (unsigned)(p.m_pad / 32);
  // planes of in[s][.] / N_s (row_splits = 1: the "partials" are the sums themselves); rows m >= S of a tile are never read back
  VSEL_LAUNCH(colsum_finish_split_kernel, dim3((unsigned)cdiv(cin, 256), S), dim3(256), 0, st, col_sums_in, sv, (int)cin, 1, S,
                     planes);
  VSEL_AFTER_LAUNCH(st, "colsum_finish_split_kernel");
  VSEL_LAUNCH(gemm_nt_bf16x3_kernel<kProjWaves>, dim3((unsigned)cdiv(cout, 64), (unsigned)cdiv(mtiles, kProjWaves), p.ks), dim3(64 * kProjWaves), 0, st, planes,
                     (const uint16_t*)weight, S, (int)cout, (int)cin, p.kslice, slabs, (int)mtiles);
  VSEL_AFTER_LAUNCH(st, "gemm_nt_bf16x3_kernel");
  VSEL_LAUNCH(colsum_linear_finish_kernel, dim3(mtiles, (unsigned)cdiv(cout, 8)), dim3(256), 0, st, slabs, p.ks, S, (int)cout,
                     p.m_pad, (const uint16_t*)bias, sv, col_sums_out);
  VSEL_AFTER_LAUNCH(st, "colsum_linear_finish_kernel");
  return VSEL_OK;
}
