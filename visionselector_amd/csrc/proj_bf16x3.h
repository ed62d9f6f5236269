// Skinny projections of the scorer on the bf16 MFMA with fp32-exact operands ("bf16x3").
//
//   kbar[m][h] = sum_d xbar[m][d] Wk[h][d] + bk[h]          (NT:  [M, D] x [Hd, D]^T)
//   w[m][d]    = sum_h kbar[m][h] Wq[h][d]                   (NN:  [M, Hd] x [Hd, D])
//
// The weights are bf16 already (exact).  The fp32 activation is split into three bf16 planes
// x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 24 significand bits), every
// bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the result
// has fp32-GEMM accuracy at 1/5 of the fp32-MFMA cycles.  M = number of segments (images) in the call.
//
// MFMA operand layout (32x32x16): lane = 32*kg + i holds A[i][8*kg .. 8*kg+7] / B[8*kg .. +7][i]; the same
// k-assignment is used for both operands, C[row][col]: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma once
#include "common.h"

namespace vsel {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8_t as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8_t, v); }

__device__ __forceinline__ void split3(float x, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
  b1 = f32_to_bf16_bits(x);
  const float r1 = x - bf16_to_f32(b1);
  b2 = f32_to_bf16_bits(r1);
  const float r2 = r1 - bf16_to_f32(b2);
  b3 = f32_to_bf16_bits(r2);
}

// sum_{j<n} p[j * stride] in index order, with 8 independent loads in flight (a plain loop serialises on latency)
__device__ __forceinline__ float strided_sum(const float* __restrict__ p, int n, int64_t stride) {
  float acc = 0.f;
  int j = 0;
  for (; j + 8 <= n; j += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(j + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; j < n; ++j) acc += p[(int64_t)j * stride];
  return acc;
}

// xs[p][m][c] (bf16 planes) = split3( sum_rs partial[m][rs][c] / N_m )
static __global__ __launch_bounds__(256) void colsum_finish_split_kernel(const float* __restrict__ partial, SegView sv,
                                                                         int d, int row_splits, int M,
                                                                         uint16_t* __restrict__ xs) {
  const int s = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const float* p = partial + (int64_t)s * row_splits * d + c;
  const float x = strided_sum(p, row_splits, d) / (float)sv.n_rows(s);
  uint32_t b1, b2, b3;
  split3(x, b1, b2, b3);
  const int64_t plane = (int64_t)M * d;
  xs[(int64_t)s * d + c] = (uint16_t)b1;
  xs[plane + (int64_t)s * d + c] = (uint16_t)b2;
  xs[2 * plane + (int64_t)s * d + c] = (uint16_t)b3;
}

// NT: part[ks][m][n] = sum_{k in slice} x[m][k] w[n][k].  grid (N/32, M/32, KS), one wave per block.
// Requires K % 16 == 0.  Rows beyond N / M are clamped (their results are never stored).
static __global__ __launch_bounds__(64) void gemm_nt_bf16x3_kernel(const uint16_t* __restrict__ xs,
                                                                   const uint16_t* __restrict__ w, int M, int N, int K,
                                                                   int kslice, float* __restrict__ part) {
  const int lane = threadIdx.x;
  const int i = lane & 31, kg = lane >> 5;
  const int n_row = min(blockIdx.x * 32 + i, N - 1);
  const int m_row = min(blockIdx.y * 32 + i, M - 1);
  const int ks = blockIdx.z;
  const int k_begin = ks * kslice;
  const int k_end = min(K, k_begin + kslice);
  const int64_t plane = (int64_t)M * K;
  const uint16_t* wp = w + (int64_t)n_row * K + 8 * kg;
  const uint16_t* xp = xs + (int64_t)m_row * K + 8 * kg;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // 4 k-steps (64 k) per iteration: 16 independent 16-byte loads in flight per lane before the 12 MFMAs
  int k0 = k_begin;
  for (; k0 + 64 <= k_end; k0 += 64) {
    u32x4 a[4], b1[4], b2[4], b3[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const u32x4*>(wp + k0 + 16 * u);
      b1[u] = *reinterpret_cast<const u32x4*>(xp + k0 + 16 * u);
      b2[u] = *reinterpret_cast<const u32x4*>(xp + plane + k0 + 16 * u);
      b3[u] = *reinterpret_cast<const u32x4*>(xp + 2 * plane + k0 + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[u]), as_bf16x8(b3[u]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[u]), as_bf16x8(b2[u]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[u]), as_bf16x8(b1[u]), acc, 0, 0, 0);
    }
  }
  for (; k0 < k_end; k0 += 16) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(wp + k0);
    const u32x4 b1 = *reinterpret_cast<const u32x4*>(xp + k0);
    const u32x4 b2 = *reinterpret_cast<const u32x4*>(xp + plane + k0);
    const u32x4 b3 = *reinterpret_cast<const u32x4*>(xp + 2 * plane + k0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b3), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b2), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b1), acc, 0, 0, 0);
  }
  const int m = blockIdx.y * 32 + (lane & 31);
  if (m < M) {
    float* dst = part + ((int64_t)ks * M + m) * N + blockIdx.x * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (blockIdx.x * 32 + row < N) dst[row] = acc[r];
    }
  }
}

// kbar[m][h] = sum_ks part[ks][m][h] + bk[h]  -> fp32 copy + bf16x3 planes; cpart[m][bx] = sum_{h in block} bq[h] kbar[m][h]
static __global__ __launch_bounds__(256) void kbar_finish_split_kernel(const float* __restrict__ part, int KS, int M, int N,
                                                                       const uint16_t* __restrict__ bk,
                                                                       const uint16_t* __restrict__ bq,
                                                                       float* __restrict__ kbar, uint16_t* __restrict__ ksp,
                                                                       float* __restrict__ cpart) {
  const int m = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  float cacc = 0.f;
  if (n < N) {
    float v = strided_sum(part + (int64_t)m * N + n, KS, (int64_t)M * N);
    v += bf16_to_f32(bk[n]);
    kbar[(int64_t)m * N + n] = v;
    uint32_t b1, b2, b3;
    split3(v, b1, b2, b3);
    const int64_t plane = (int64_t)M * N;
    ksp[(int64_t)m * N + n] = (uint16_t)b1;
    ksp[plane + (int64_t)m * N + n] = (uint16_t)b2;
    ksp[2 * plane + (int64_t)m * N + n] = (uint16_t)b3;
    cacc = bf16_to_f32(bq[n]) * v;
  }
  cacc = wave_sum(cacc);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cacc;
  __syncthreads();
  if (threadIdx.x == 0) cpart[(int64_t)m * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// NN: part[ks][m][n] = sum_{k in slice} x[m][k] w[k][n], w row-major [K, N].  grid (N/256, M/32, KS), one wave.
// Lane (i, kg) owns 8 consecutive n (n0 + 8i .. +7) and k rows k0 + 8kg .. +7: eight 16-byte row loads, then a
// register transpose (v_perm_b32) builds, for each t in 0..7, the A fragment {w[k0+8kg+e][n0+8i+t]}_e of the
// 32x32 tile "n = n0 + 8*i' + t".  Requires K % 16 == 0 and N % 8 == 0.
static __global__ __launch_bounds__(64) void gemm_nn_bf16x3_kernel(const uint16_t* __restrict__ xs,
                                                                   const uint16_t* __restrict__ w, int M, int N, int K,
                                                                   int kslice, float* __restrict__ part) {
  const int lane = threadIdx.x;
  const int i = lane & 31, kg = lane >> 5;
  const int nb = min(blockIdx.x * 256 + 8 * i, N - 8);
  const int m_row = min(blockIdx.y * 32 + i, M - 1);
  const int ks = blockIdx.z;
  const int k_begin = ks * kslice;
  const int k_end = min(K, k_begin + kslice);
  const int64_t plane = (int64_t)M * K;
  const uint16_t* xp = xs + (int64_t)m_row * K + 8 * kg;
  const uint16_t* wp = w + (int64_t)(8 * kg) * N + nb;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  u32x4 wv[8], b1, b2, b3;
#pragma unroll
  for (int e = 0; e < 8; ++e) wv[e] = *reinterpret_cast<const u32x4*>(wp + (int64_t)(k_begin + e) * N);
  b1 = *reinterpret_cast<const u32x4*>(xp + k_begin);
  b2 = *reinterpret_cast<const u32x4*>(xp + plane + k_begin);
  b3 = *reinterpret_cast<const u32x4*>(xp + 2 * plane + k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += 16) {
    // software prefetch of the next step (the last iteration re-reads its own step)
    const int kn = (k0 + 16 < k_end) ? k0 + 16 : k0;
    u32x4 nwv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) nwv[e] = *reinterpret_cast<const u32x4*>(wp + (int64_t)(kn + e) * N);
    const u32x4 nb1 = *reinterpret_cast<const u32x4*>(xp + kn);
    const u32x4 nb2 = *reinterpret_cast<const u32x4*>(xp + plane + kn);
    const u32x4 nb3 = *reinterpret_cast<const u32x4*>(xp + 2 * plane + kn);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      // element t of vector e sits in dword t>>1, half t&1; pack (e even -> low half, e odd -> high half)
      const uint32_t sel = (t & 1) ? 0x07060302u : 0x05040100u;
      u32x4 a;
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = __builtin_amdgcn_perm(wv[2 * q + 1][t >> 1], wv[2 * q][t >> 1], sel);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b3), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b2), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b1), acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[e] = nwv[e];
    b1 = nb1; b2 = nb2; b3 = nb3;
  }
  const int m = blockIdx.y * 32 + (lane & 31);
  if (m < M) {
    float* dst = part + ((int64_t)ks * M + m) * N + blockIdx.x * 256;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int irow = (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (blockIdx.x * 256 + 8 * irow < N) {
        f32x4 lo = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        f32x4 hi = {acc[4][r], acc[5][r], acc[6][r], acc[7][r]};
        *reinterpret_cast<f32x4*>(dst + 8 * irow) = lo;
        *reinterpret_cast<f32x4*>(dst + 8 * irow + 4) = hi;
      }
    }
  }
}

// w[m][n] = sum_ks part[ks][m][n];  c[m] = sum_j cpart[m][j]  (block x == 0 of each m)
static __global__ __launch_bounds__(256) void w_finish_kernel(const float* __restrict__ part, int KS, int M, int N,
                                                              const float* __restrict__ cpart, int n_cpart,
                                                              float* __restrict__ w, float* __restrict__ c) {
  const int m = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) {
    w[(int64_t)m * N + n] = strided_sum(part + (int64_t)m * N + n, KS, (int64_t)M * N);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float t = 0.f;
    for (int j = 0; j < n_cpart; ++j) t += cpart[(int64_t)m * n_cpart + j];
    c[m] = t;
  }
}

}  // namespace vsel
