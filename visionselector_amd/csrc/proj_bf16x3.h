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
//
// Memory shapes chosen for the load/store path, not the math (the MFMA work is ~1 us of chip time; the first version
// spent 20-40 us in 32-rows-x-32-bytes fragment-shaped loads and 4-byte scattered slab stores):
//   * the activation planes are produced by our own finish kernels, so they are stored FRAGMENT-MAJOR:
//       plane[p][m_tile][k_step][lane][8]  ->  a wave's B operand for one k-step is ONE contiguous 1 KiB load
//   * split-K slabs are stored [ks][n][m_pad] (m fastest): an MFMA C column (32 lanes = 32 consecutive m) is one
//     128-byte line per store instead of 32 scattered dwords
//   * the NT kernel covers two 32-row weight tiles per wave, halving the re-reads of the activation planes
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

// sum_{j<n} p[j * stride] in index order.  24 loads in flight per batch, UNCONDITIONAL with a clamped index (a plain loop
// serialises on latency; an 8-wide batch plus a scalar tail made 18 partials four round trips, 14 slabs seven)
__device__ __forceinline__ float strided_sum(const float* __restrict__ p, int n, int64_t stride) {
  float acc = 0.f;
  for (int j = 0; j < n; j += 24) {
    float v[24];
#pragma unroll
    for (int u = 0; u < 24; ++u) v[u] = p[(int64_t)min(j + u, n - 1) * stride];
#pragma unroll
    for (int u = 0; u < 24; ++u)
      if (j + u < n) acc += v[u];
  }
  return acc;
}

// fragment-major offset (in bf16 elements) of activation element (m, k) of plane p; K % 16 == 0
__device__ __forceinline__ int64_t frag_off(int p, int m, int k, int m_tiles, int k_steps) {
  const int mt = m >> 5, i = m & 31, ks = k >> 4, kg = (k >> 3) & 1, e = k & 7;
  return ((((int64_t)p * m_tiles + mt) * k_steps + ks) * 64 + (32 * kg + i)) * 8 + e;
}

// xs (fragment-major bf16x3 planes) = split3( sum_rs partial[m][rs][c] / N_m )
static __attribute__((unused)) __global__ __launch_bounds__(256) void colsum_finish_split_kernel(const float* __restrict__ partial, SegView sv,
                                                                         int d, int row_splits, int M,
                                                                         uint16_t* __restrict__ xs) {
  const int s = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const float* p = partial + (int64_t)s * row_splits * d + c;
  const float x = strided_sum(p, row_splits, d) / (float)sv.n_rows(s);
  uint32_t b1, b2, b3;
  split3(x, b1, b2, b3);
  const int mt = (M + 31) >> 5, ksteps = d >> 4;
  xs[frag_off(0, s, c, mt, ksteps)] = (uint16_t)b1;
  xs[frag_off(1, s, c, mt, ksteps)] = (uint16_t)b2;
  xs[frag_off(2, s, c, mt, ksteps)] = (uint16_t)b3;
}

// NT: part[ks][n][m_pad] = sum_{k in slice} x[m][k] w[n][k].  grid (ceil(N/64), m_tiles, KS), one wave per block:
// two 32-row weight tiles x one 32-column activation tile.  Requires K % 16 == 0.  m_pad = 32 * m_tiles.
template <int WPB>
static __attribute__((unused)) __global__ __launch_bounds__(64 * WPB) void gemm_nt_bf16x3_kernel(const uint16_t* __restrict__ xs,
                                                                   const uint16_t* __restrict__ w, int /*M*/, int N, int K,
                                                                   int kslice, float* __restrict__ part, int m_tiles) {
  // WPB independent waves per workgroup on WPB m-tiles and the SAME weight tile: fetched from L2 once per CU.  Each wave's
  // arithmetic is what it was: bit-identical slabs.
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, kg = lane >> 5;
  const int mt = blockIdx.y * WPB + (int)(threadIdx.x >> 6), ksteps = K >> 4;
  if (mt >= m_tiles) return;
  const int n0 = blockIdx.x * 64;
  const int row0 = min(n0 + i, N - 1), row1 = min(n0 + 32 + i, N - 1);
  const int ks = blockIdx.z;
  const int k_begin = ks * kslice;
  const int k_end = min(K, k_begin + kslice);
  const uint16_t* wp0 = w + (int64_t)row0 * K + 8 * kg;
  const uint16_t* wp1 = w + (int64_t)row1 * K + 8 * kg;
  const int64_t plane = (int64_t)m_tiles * ksteps * 512;          // elements per plane
  const uint16_t* xp = xs + ((int64_t)mt * ksteps * 64 + lane) * 8;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  // The kernel is latency-bound (one wave per block, ~400 waves): issue the loads of 8 k-steps (40 x 16 B per lane)
  // back to back, then run their 48 MFMAs; a 256-wide k-slice is two such round trips.
  int k0 = k_begin;
  constexpr int KB = 8;
  for (; k0 + 16 * KB <= k_end; k0 += 16 * KB) {
    u32x4 a0[KB], a1[KB], b1[KB], b2[KB], b3[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int64_t xo = (int64_t)((k0 >> 4) + u) * 512;
      a0[u] = *reinterpret_cast<const u32x4*>(wp0 + k0 + 16 * u);
      a1[u] = *reinterpret_cast<const u32x4*>(wp1 + k0 + 16 * u);
      b1[u] = *reinterpret_cast<const u32x4*>(xp + xo);
      b2[u] = *reinterpret_cast<const u32x4*>(xp + plane + xo);
      b3[u] = *reinterpret_cast<const u32x4*>(xp + 2 * plane + xo);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep all loads above the MFMAs (the scheduler otherwise sinks them to save VGPRs)
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0[u]), as_bf16x8(b3[u]), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1[u]), as_bf16x8(b3[u]), acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0[u]), as_bf16x8(b2[u]), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1[u]), as_bf16x8(b2[u]), acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0[u]), as_bf16x8(b1[u]), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1[u]), as_bf16x8(b1[u]), acc1, 0, 0, 0);
    }
  }
  for (; k0 < k_end; k0 += 16) {
    const int64_t xo = (int64_t)(k0 >> 4) * 512;
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(wp0 + k0);
    const u32x4 a1 = *reinterpret_cast<const u32x4*>(wp1 + k0);
    const u32x4 b1 = *reinterpret_cast<const u32x4*>(xp + xo);
    const u32x4 b2 = *reinterpret_cast<const u32x4*>(xp + plane + xo);
    const u32x4 b3 = *reinterpret_cast<const u32x4*>(xp + 2 * plane + xo);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0), as_bf16x8(b3), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1), as_bf16x8(b3), acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0), as_bf16x8(b2), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1), as_bf16x8(b2), acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0), as_bf16x8(b1), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1), as_bf16x8(b1), acc1, 0, 0, 0);
  }
  // C[row = n][col = m]; slab layout [ks][n][m_pad]: 32 lanes of one C row -> 128 contiguous bytes
  const int m_pad = 32 * m_tiles;
  float* dst = part + ((int64_t)ks * N) * m_pad + 32 * mt + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
    if (n0 + row < N) dst[(int64_t)(n0 + row) * m_pad] = acc0[r];
    if (n0 + 32 + row < N) dst[(int64_t)(n0 + 32 + row) * m_pad] = acc1[r];
  }
}

// kbar[m][h] = sum_ks part[ks][h][m] + bk[h]  -> fp32 copy + fragment-major bf16x3 planes;
// cpart[m][by] = sum_{h in block} bq[h] kbar[m][h].  grid (ceil(m_pad/32), ceil(N/8)), block 256 = 32 m x 8 h.
static __attribute__((unused)) __global__ __launch_bounds__(256) void kbar_finish_split_kernel(const float* __restrict__ part, int KS, int M, int N,
                                                                       int m_pad, const uint16_t* __restrict__ bk,
                                                                       const uint16_t* __restrict__ bq,
                                                                       float* __restrict__ kbar, uint16_t* __restrict__ ksp,
                                                                       float* __restrict__ cpart) {
  const int m = blockIdx.x * 32 + (threadIdx.x & 31);
  const int n = blockIdx.y * 8 + (threadIdx.x >> 5);
  float cterm = 0.f;
  if (m < M && n < N) {
    float v = strided_sum(part + (int64_t)n * m_pad + m, KS, (int64_t)N * m_pad);
    v += bf16_to_f32(bk[n]);
    kbar[(int64_t)m * N + n] = v;
    uint32_t b1, b2, b3;
    split3(v, b1, b2, b3);
    const int mt = m_pad >> 5, ksteps = N >> 4;
    ksp[frag_off(0, m, n, mt, ksteps)] = (uint16_t)b1;
    ksp[frag_off(1, m, n, mt, ksteps)] = (uint16_t)b2;
    ksp[frag_off(2, m, n, mt, ksteps)] = (uint16_t)b3;
    cterm = bf16_to_f32(bq[n]) * v;
  }
  // per-m sum over the block's 8 h values, fixed order
  __shared__ float red[8][33];
  red[threadIdx.x >> 5][threadIdx.x & 31] = cterm;
  __syncthreads();
  if (threadIdx.x < 32 && m < M) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[j][threadIdx.x];
    cpart[(int64_t)m * gridDim.y + blockIdx.y] = t;
  }
}

// NN: part[ks][n][m_pad] = sum_{k in slice} x[m][k] w[k][n], w row-major [K, N].  grid (ceil(N/256), m_tiles, KS), one wave.
// Lane (i, kg) owns 8 consecutive n (n0 + 8i .. +7) and k rows k0 + 8kg .. +7: eight 16-byte row loads, then a
// register transpose (v_perm_b32) builds, for each t in 0..7, the A fragment {w[k0+8kg+e][n0+8i+t]}_e of the
// 32x32 tile "n = n0 + 8*i' + t".  Requires K % 16 == 0 and N % 8 == 0.
template <int WPB>
static __attribute__((unused)) __global__ __launch_bounds__(64 * WPB) void gemm_nn_bf16x3_kernel(const uint16_t* __restrict__ xs,
                                                                   const uint16_t* __restrict__ w, int /*M*/, int N, int K,
                                                                   int kslice, float* __restrict__ part, int m_tiles) {
  // WPB independent waves per workgroup on WPB m-tiles and the SAME weight tile (72 % of this kernel's L2 traffic): fetched from
  // L2 once per CU.  Each wave's arithmetic is what it was: bit-identical slabs.
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, kg = lane >> 5;
  const int mt = blockIdx.y * WPB + (int)(threadIdx.x >> 6), ksteps = K >> 4;
  if (mt >= m_tiles) return;
  const int nb = min(blockIdx.x * 256 + 8 * i, N - 8);
  const int ks = blockIdx.z;
  const int k_begin = ks * kslice;
  const int k_end = min(K, k_begin + kslice);
  const int64_t plane = (int64_t)m_tiles * ksteps * 512;
  const uint16_t* xp = xs + ((int64_t)mt * ksteps * 64 + lane) * 8;
  const uint16_t* wp = w + (int64_t)(8 * kg) * N + nb;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // latency-bound like the NT kernel: the loads of KB k-steps (KB x 11 x 16 B per lane) are issued together
  auto step = [&](const u32x4 (&wv)[8], u32x4 b1, u32x4 b2, u32x4 b3) {
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
  };
  constexpr int KB = 4;
  int k0 = k_begin;
  for (; k0 + 16 * KB <= k_end; k0 += 16 * KB) {
    u32x4 wv[KB][8], b1[KB], b2[KB], b3[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) {
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[u][e] = *reinterpret_cast<const u32x4*>(wp + (int64_t)(k0 + 16 * u + e) * N);
      const int64_t xo = (int64_t)((k0 >> 4) + u) * 512;
      b1[u] = *reinterpret_cast<const u32x4*>(xp + xo);
      b2[u] = *reinterpret_cast<const u32x4*>(xp + plane + xo);
      b3[u] = *reinterpret_cast<const u32x4*>(xp + 2 * plane + xo);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KB; ++u) step(wv[u], b1[u], b2[u], b3[u]);
  }
  for (; k0 < k_end; k0 += 16) {
    u32x4 wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[e] = *reinterpret_cast<const u32x4*>(wp + (int64_t)(k0 + e) * N);
    const int64_t xo = (int64_t)(k0 >> 4) * 512;
    step(wv, *reinterpret_cast<const u32x4*>(xp + xo), *reinterpret_cast<const u32x4*>(xp + plane + xo),
         *reinterpret_cast<const u32x4*>(xp + 2 * plane + xo));
  }
  // acc[t][r] = C[n = n0 + 8*irow + t][m = 32*mt + (lane & 31)]; slab [ks][n][m_pad]
  const int m_pad = 32 * m_tiles;
  float* dst = part + ((int64_t)ks * N) * m_pad + 32 * mt + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int irow = (r & 3) + 8 * (r >> 2) + 4 * kg;
    const int n = blockIdx.x * 256 + 8 * irow;
    if (n < N) {
#pragma unroll
      for (int t = 0; t < 8; ++t) dst[(int64_t)(n + t) * m_pad] = acc[t][r];
    }
  }
}

// w[m][n] = sum_ks part[ks][n][m];  c[m] = sum_j cpart[m][j].  grid (ceil(m_pad/32), ceil(N/8)), block 256 = 32 m x 8 n.
static __attribute__((unused)) __global__ __launch_bounds__(256) void w_finish_kernel(const float* __restrict__ part, int KS, int M, int N, int m_pad,
                                                              const float* __restrict__ cpart, int n_cpart,
                                                              float* __restrict__ w, float* __restrict__ c) {
  const int m = blockIdx.x * 32 + (threadIdx.x & 31);
  const int n = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (m < M && n < N) w[(int64_t)m * N + n] = strided_sum(part + (int64_t)n * m_pad + m, KS, (int64_t)N * m_pad);
  if (blockIdx.y == 0) {
    // c[m]: thread (m, nj) sums cpart[m][nj], [nj + 8], ... then the 8 partials are added in fixed order
    const int nj = threadIdx.x >> 5;
    __shared__ float red[8][33];
    red[nj][threadIdx.x & 31] = (m < M && nj < n_cpart) ? strided_sum(cpart + (int64_t)m * n_cpart + nj, (n_cpart - nj + 7) / 8, 8) : 0.f;
    __syncthreads();
    if (threadIdx.x < 32 && m < M) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t += red[j][threadIdx.x];
      c[m] = t;
    }
  }
}

}  // namespace vsel
