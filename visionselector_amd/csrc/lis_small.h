// Small-batch form of the LIS inference path (1 .. kSmallMaxSeg segments per call; the reference's real evaluation call is ONE
// image: qwen-evaluation/token_compression/selector_model.py:182-194, assert :270).
//
// On MI355X every dependent launch costs >= ~4 us whatever it computes (dispatch, first dependent loads from another XCD's
// write-back, end-of-kernel release), and an in-kernel grid barrier costs the same (MI355X_MICROARCH.md, barrier-xcd 4.1 us), so
// the lever at one image is the NUMBER OF DEPENDENT PHASES, not their size.  The nine launches of the batched form
//     colsum_partial, colsum_finish_split, gemm_nt, kbar_finish_split, gemm_nn, w_finish, score, topk_select, gather_rows
// become five by letting every workgroup of a consumer rebuild, redundantly and from L2, the tiny reduction its producer left
// open (for a handful of segments that is a few KB per workgroup):
//     colsum_partial | proj_nt_small (xbar slice in the prologue) | proj_nn_small (kbar slice + c partials in the prologue)
//                    | score_small (w, c in the prologue)         | select_gather_small (radix select per workgroup, then its rows)
// The arithmetic is the batched form's, operation for operation: same 128-row sweep-1 chunks, same bf16x3 planes, same MFMA
// sequence per (tile, 256 / 128-wide k-slice), same slab order, same integer select -- scores, indices and rows are bit-identical
// to the nine-launch path (tests/test_lis_gpu.py::test_small_batch_path_is_bit_identical).
#pragma once
#include "lis_kernels.h"
#include <stdlib.h>

namespace vsel {

constexpr int kSmallMaxSeg = 8;            // rows of the M dimension a workgroup rebuilds in its prologue
constexpr int kSmallSelectThreads = 256;

// -------------------------------------------------------------------------------------------------------------------------
// P1  part1[ks][m][n] = sum_{k in slice ks} xbar[m][k] Wk[n][k]      (bf16x3 planes of xbar built in the prologue)
//     grid (ceil(N / 32), KS), one wave = ONE 32-row MFMA tile: at one image the chain of 48 dependent MFMAs per tile is the
//     kernel's arithmetic floor, so tiles are spread over as many SIMDs as there are (784 waves at the 7B geometry).
//     partial: sweep-1 partials [S][row_splits][K] (or the producer's column sums, row_splits 1)
// -------------------------------------------------------------------------------------------------------------------------
//     split_stride: floats between consecutive row splits of one segment (K for the sweep-1 partials); mean = 0: the sums are
//     used as they are instead of being divided by the segment's row count (training backward: Wq (sum_i g_i x_i))
__device__ __forceinline__ void proj_nt_small_body(const float* __restrict__ partial, const SegView& sv, int S, int row_splits,
                                                   const uint16_t* __restrict__ w, int N, int K, int kslice,
                                                   float* __restrict__ part, int64_t split_stride, int mean) {
  __shared__ __attribute__((aligned(16))) uint16_t xs[3][kSmallMaxSeg][kSliceNT];
  const int lane = threadIdx.x;
  const int i = lane & 31, kg = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int row0 = min(n0 + i, N - 1);
  const int ks = blockIdx.y;
  const int k_begin = ks * kslice;
  const int k_end = min(K, k_begin + kslice);
  const uint16_t* wp0 = w + (int64_t)row0 * K + 8 * kg;
  VSEL_STAMP(1, 0);
  constexpr int KB = kSliceNT / 16;      // the whole 256-wide slice: one wave per SIMD at these batch sizes, registers are free
  // the weights do not depend on the prologue: put the loads of the whole slice in flight first (one round trip, overlapped
  // with the partial-sum loads below)
  u32x4 a0[KB];
  const bool full = k_begin + 16 * KB <= k_end;
  if (full) {
#pragma unroll
    for (int u = 0; u < KB; ++u) a0[u] = *reinterpret_cast<const u32x4*>(wp0 + k_begin + 16 * u);
  }
  // prologue: xbar[m][k] = (sum_rs partial[m][rs][k]) / N_m for the slice, in rs order (= colsum_finish_split_kernel), split in 3.
  // Lane owns 4 consecutive k (one 16-byte load per partial); up to 24 partials per batch are in flight together.
  const int kq = k_begin + 4 * lane;
  const bool kq_ok = kq < k_end;                     // K % 16 == 0: a 4-group is inside the slice or outside it
  for (int m = 0; m < S; ++m) {
    const float nf = (float)sv.n_rows(m);
    const float* pc = partial + (int64_t)m * row_splits * split_stride + (kq_ok ? kq : k_begin);   // lanes past the slice read a valid dummy
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < row_splits; r0 += 24) {
      f32x4 v[24];
#pragma unroll
      for (int q = 0; q < 24; ++q)       // UNCONDITIONAL loads (clamped index): a per-element "load or zero" on a runtime bound makes
                                         // hipcc branch around every load and wait for each one (cdna_hip_programming.md, trap (c))
        v[q] = *reinterpret_cast<const f32x4*>(pc + (int64_t)min(r0 + q, row_splits - 1) * split_stride);
#pragma unroll
      for (int q = 0; q < 24; ++q)
        if (r0 + q < row_splits) acc += v[q];
    }
    uint32_t b[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      b[0][e] = b[1][e] = b[2][e] = 0u;
      if (kq_ok) split3(mean ? acc[e] / nf : acc[e], b[0][e], b[1][e], b[2][e]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      uint2 pk;
      pk.x = b[pl][0] | (b[pl][1] << 16);
      pk.y = b[pl][2] | (b[pl][3] << 16);
      *reinterpret_cast<uint2*>(&xs[pl][m][4 * lane]) = pk;
    }
  }
  __syncthreads();
  VSEL_STAMP(1, 1);
  const bool act = i < S;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  auto ldb = [&](int p, int koff) -> u32x4 {          // B fragment: lane (i, kg) holds xbar[i][koff + 8 kg .. + 7]
    return act ? *reinterpret_cast<const u32x4*>(&xs[p][i][koff + 8 * kg]) : zero;
  };
  f32x16 acc0;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
  int k0 = k_begin;
  if (full) {
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int ko = 16 * u;
      const u32x4 b1 = ldb(0, ko), b2 = ldb(1, ko), b3 = ldb(2, ko);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0[u]), as_bf16x8(b3), acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0[u]), as_bf16x8(b2), acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0[u]), as_bf16x8(b1), acc0, 0, 0, 0);
    }
    k0 = k_end;
  }
  for (; k0 < k_end; k0 += 16) {
    const u32x4 x0 = *reinterpret_cast<const u32x4*>(wp0 + k0);
    const int ko = k0 - k_begin;
    const u32x4 b1 = ldb(0, ko), b2 = ldb(1, ko), b3 = ldb(2, ko);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(x0), as_bf16x8(b3), acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(x0), as_bf16x8(b2), acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(x0), as_bf16x8(b1), acc0, 0, 0, 0);
  }
  // C[row = n][col = m]; compact slab [ks][m][n] (only the S live columns are stored)
#ifdef VSEL_TRACE
  asm volatile("" ::"v"(acc0[0]));
#endif
  VSEL_STAMP(1, 2);
  if (act) {
    float* dst = part + ((int64_t)ks * S + i) * N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (n0 + row < N) dst[n0 + row] = acc0[r];
    }
  }
  VSEL_STAMP_DRAIN(1, 3);
}

static __attribute__((unused)) __global__ __launch_bounds__(64) void proj_nt_small_kernel(const float* __restrict__ partial, SegView sv, int S, int row_splits,
                                                                  const uint16_t* __restrict__ w, int N, int K, int kslice,
                                                                  float* __restrict__ part, int64_t split_stride, int mean) {
  proj_nt_small_body(partial, sv, S, row_splits, w, N, K, kslice, part, split_stride, mean);
}

// Two independent P1 products of the same shape in ONE launch (training backward: kbar = Wk xbar and dk_raw = Wq (sum_i g_i x_i), both
// from the weighted column-sum partials): grid (ceil(N / 32), KS, 2), blockIdx.z picks the operand set.  Same body, same bits.
static __attribute__((unused)) __global__ __launch_bounds__(64) void proj_nt_small_pair_kernel(const float* __restrict__ partial0, const float* __restrict__ partial1,
                                                                       SegView sv, int S, int row_splits, const uint16_t* __restrict__ w0,
                                                                       const uint16_t* __restrict__ w1, int N, int K, int kslice,
                                                                       float* __restrict__ part0, float* __restrict__ part1_, int64_t split_stride,
                                                                       int mean0, int mean1) {
  if (blockIdx.z == 0) proj_nt_small_body(partial0, sv, S, row_splits, w0, N, K, kslice, part0, split_stride, mean0);
  else proj_nt_small_body(partial1, sv, S, row_splits, w1, N, K, kslice, part1_, split_stride, mean1);
}

// -------------------------------------------------------------------------------------------------------------------------
// P2  part2[ks][m][n] = sum_{h in slice ks} kbar[m][h] Wq[h][n],  kbar[m][h] = sum_ks1 part1[ks1][m][h] + bk[h] (prologue);
//     workgroups of n-tile 0 also leave cpart[m][h / 8] = sum_{8 h} bq[h] kbar[m][h].  grid (ceil(N / 128), KS2), one wave:
//     lane (i, kg) loads 8 bytes = Wq[h][n0 + 4 i .. + 3] per k-row, MFMA tile t takes column 4 i + t (four tiles, 96 MFMAs per
//     wave; the 256-column / eight-tile form of the batched kernel spends 3.9 us in its 192 MFMAs at one image).
// -------------------------------------------------------------------------------------------------------------------------
static __attribute__((unused)) __global__ __launch_bounds__(64) void proj_nn_small_kernel(const float* __restrict__ part1, int KS1, int S,
                                                                  const uint16_t* __restrict__ bk, const uint16_t* __restrict__ bq,
                                                                  const uint16_t* __restrict__ w, int N, int K, int kslice,
                                                                  float* __restrict__ part2, float* __restrict__ cpart, int n_cpart) {
  __shared__ __attribute__((aligned(16))) uint16_t ksl[3][kSmallMaxSeg][kSliceNN];
  __shared__ float ct[kSmallMaxSeg][kSliceNN];
  const int lane = threadIdx.x;
  const int i = lane & 31, kg = lane >> 5;
  const int nb = min(blockIdx.x * 128 + 4 * i, N - 4);
  const int ks = blockIdx.y;
  const int k_begin = ks * kslice;
  const int k_end = min(K, k_begin + kslice);
  const uint16_t* wp = w + (int64_t)(8 * kg) * N + nb;
  VSEL_STAMP(2, 0);
  constexpr int KB = kSliceNN / 16;      // the whole 128-deep slice in flight (64 x 8 bytes per lane)
  u32x2 wv[KB][8];
  const bool full = k_begin + 16 * KB <= k_end;
  if (full) {
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[u][e] = *reinterpret_cast<const u32x2*>(wp + (int64_t)(k_begin + 16 * u + e) * N);
  }
  // prologue: kbar slice (= kbar_finish_split_kernel: slabs in ks1 order, then + bk), split in 3; c terms
  static_assert(kSliceNN == 128, "lane owns 2 consecutive h of the 128-wide slice");
  const int hq = k_begin + 2 * lane;
  const bool hq_ok = hq < k_end;                     // K % 16 == 0
  const float bk0 = hq_ok ? bf16_to_f32(bk[hq]) : 0.f, bk1 = hq_ok ? bf16_to_f32(bk[hq + 1]) : 0.f;
  const float bq0 = hq_ok ? bf16_to_f32(bq[hq]) : 0.f, bq1 = hq_ok ? bf16_to_f32(bq[hq + 1]) : 0.f;
  for (int m = 0; m < S; ++m) {
    float2 v = make_float2(0.f, 0.f);
    const float* p1 = part1 + (int64_t)m * K + (hq_ok ? hq : k_begin);
    const int64_t stride = (int64_t)S * K;
    for (int s0 = 0; s0 < KS1; s0 += 16) {             // all slab loads of a batch in flight (unconditional, clamped), added in ks1 order
      float2 t[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] = *reinterpret_cast<const float2*>(p1 + min(s0 + q, KS1 - 1) * stride);
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (s0 + q < KS1) { v.x += t[q].x; v.y += t[q].y; }
    }
    uint32_t b[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
    float c0 = 0.f, c1 = 0.f;
    if (hq_ok) {
      const float k0v = v.x + bk0, k1v = v.y + bk1;
      split3(k0v, b[0][0], b[1][0], b[2][0]);
      split3(k1v, b[0][1], b[1][1], b[2][1]);
      c0 = bq0 * k0v;
      c1 = bq1 * k1v;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint32_t*>(&ksl[pl][m][2 * lane]) = b[pl][0] | (b[pl][1] << 16);
    *reinterpret_cast<float2*>(&ct[m][2 * lane]) = make_float2(c0, c1);
  }
  __syncthreads();
  VSEL_STAMP(2, 1);
  if (blockIdx.x == 0) {
    // cpart[m][by] = sum of the block's 8 consecutive h in order (kbar_finish_split_kernel's LDS reduction)
    const int blocks = (k_end - k_begin + 7) / 8;
    for (int e = lane; e < S * blocks; e += 64) {
      const int m = e / blocks, q = e % blocks;
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t += ct[m][8 * q + j];
      cpart[(int64_t)m * n_cpart + (k_begin >> 3) + q] = t;
    }
  }
  const bool act = i < S;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  auto ldb = [&](int p, int koff) -> u32x4 {
    return act ? *reinterpret_cast<const u32x4*>(&ksl[p][i][koff + 8 * kg]) : zero;
  };
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto step = [&](const u32x2 (&x)[8], u32x4 b1, u32x4 b2, u32x4 b3) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t sel = (t & 1) ? 0x07060302u : 0x05040100u;
      u32x4 a;
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = __builtin_amdgcn_perm(x[2 * q + 1][t >> 1], x[2 * q][t >> 1], sel);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b3), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b2), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b1), acc[t], 0, 0, 0);
    }
  };
  int k0 = k_begin;
  if (full) {
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int ko = 16 * u;
      step(wv[u], ldb(0, ko), ldb(1, ko), ldb(2, ko));
    }
    k0 = k_end;
  }
  for (; k0 < k_end; k0 += 16) {
    u32x2 x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = *reinterpret_cast<const u32x2*>(wp + (int64_t)(k0 + e) * N);
    const int ko = k0 - k_begin;
    step(x, ldb(0, ko), ldb(1, ko), ldb(2, ko));
  }
  // acc[t][r] = C[n = n0 + 4 irow + t][m = lane & 31]; compact slab [ks][m][n]: 4 consecutive n per (lane, r)
#ifdef VSEL_TRACE
  asm volatile("" ::"v"(acc[0][0]), "v"(acc[3][0]));
#endif
  VSEL_STAMP(2, 2);
  if (act) {
    float* dst = part2 + ((int64_t)ks * S + i) * N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int irow = (r & 3) + 8 * (r >> 2) + 4 * kg;
      const int n = blockIdx.x * 128 + 4 * irow;
      if (n < N) {
        const f32x4 v4 = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        *reinterpret_cast<f32x4*>(dst + n) = v4;
      }
    }
  }
  VSEL_STAMP_DRAIN(2, 3);
}

// -------------------------------------------------------------------------------------------------------------------------
// P3  sweep 2 with w[s] = sum_ks part2[ks][s][:] (ks order = w_finish_kernel) and c[s] rebuilt in LDS by every workgroup.
//     grid (row_chunks, S), block 512 (8 waves: rows rb + wave + 8 u).  Dynamic LDS: d + 16 floats.
//     One memory round trip: the wave's first two token rows (they do not depend on the projections), the c partials and the
//     slab loads of the w rebuild (2 column groups x <= 16 slabs per thread) are issued back to back before anything is
//     consumed (tools/trace_small.py: two 32-load batches, then the c chain, then the rows took 5.6 + 1.1 + 3.3 us in a row).
//     Per row the arithmetic is score_rows' (dot_raw over the column groups in order, wave_sum, (a + c) / sqrt(Hd)).
// -------------------------------------------------------------------------------------------------------------------------
constexpr int kSmallScoreThreads = 512;

template <typename T, int ITERS>
__global__ __launch_bounds__(kSmallScoreThreads) void score_small_kernel(const T* __restrict__ h, SegView sv, int d, int S,
                                                                         const float* __restrict__ part2, int KS2,
                                                                         const float* __restrict__ cpart, int n_cpart,
                                                                         float sqrt_hd, int rows_per_block,
                                                                         float* __restrict__ scores,
                                                                         const int64_t* __restrict__ out_map) {
  constexpr int V = Elem<T>::kVec;
  constexpr int XI = ITERS > 0 ? ITERS : 1;
  constexpr int NT = kSmallScoreThreads, NWV = NT / 64;
  extern __shared__ __attribute__((aligned(16))) float wl[];      // [d] then 16 floats for c
  float* cred = wl + d;
  const int s = blockIdx.y;
  const int n = sv.n_rows(s);
  const int rb = blockIdx.x * rows_per_block;
  if (rb >= n) return;
  const int re = min(n, rb + rows_per_block);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = sv.row_begin(s);
  VSEL_STAMP(3, 0);
  // (1) rows rb + wave, rb + wave + NWV as raw 16-byte vectors (clamped: rows past the end are loaded and not used)
  const T* base = h + r0 * (int64_t)d + lane * V;
  u32x4 x[2][XI];
  if constexpr (ITERS > 0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = min(rb + wave + NWV * u, n - 1);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) x[u][it] = *reinterpret_cast<const u32x4*>(base + (int64_t)row * d + it * 64 * V);
    }
  }
  // (2) c[s]: w_finish_kernel's order -- 8 strided partial sums (nj, nj + 8, ...) then those 8 in order.  Thread nj < 8 loads its
  // whole chain now (clamped, <= 32 per batch in flight with everything else) and adds it in order after the slab loads are out
  const float* cps = cpart + (int64_t)s * n_cpart;
  float ct[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) ct[q] = cps[min((tid & 7) + 8 * q, n_cpart - 1)];
  // (3) w: two 4-column groups per thread and pass, all their slab loads in flight together, added in ks order
  const float* src = part2 + (int64_t)s * d;
  const int64_t slab = (int64_t)S * d;
  const int nc4 = d / 4;
  for (int c4 = tid; c4 < nc4; c4 += 2 * NT) {
    const int c4b = c4 + NT;
    const bool has_b = c4b < nc4;
    const int c4bc = has_b ? c4b : c4;               // unconditional loads (clamped): see proj_nt_small_kernel
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KS2; k0 += 16) {
      f32x4 va[16], vb[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int64_t ku = min(k0 + u, KS2 - 1);
        va[u] = *reinterpret_cast<const f32x4*>(src + ku * slab + 4 * c4);
        vb[u] = *reinterpret_cast<const f32x4*>(src + ku * slab + 4 * c4bc);
      }
      asm volatile("" ::: "memory");                 // every load above is issued before the first add below waits for one
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (k0 + u < KS2) { a += va[u]; b += vb[u]; }
    }
    *reinterpret_cast<f32x4*>(wl + 4 * c4) = a;
    if (has_b) *reinterpret_cast<f32x4*>(wl + 4 * c4b) = b;
  }
  if (tid < 8) {
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q)
      if (tid + 8 * q < n_cpart) a += ct[q];
    for (int e0 = tid + 256; e0 < n_cpart; e0 += 8) a += cps[e0];      // n_cpart > 256 (Hd > 2048): the tail, in order
    cred[tid] = a;
  }
  __syncthreads();
  VSEL_STAMP(3, 1);
  VSEL_STAMP(3, 2);
  float cs = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) cs += cred[j];
  if constexpr (ITERS > 0) {
    float wr[ITERS][V];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int q = 0; q < V; q += 4) {
        float t4[4];
        load_vec(wl + (it * 64 + lane) * V + q, t4);
#pragma unroll
        for (int z = 0; z < 4; ++z) wr[it][q + z] = t4[z];
      }
    }
    for (int r = rb + wave; r < re; r += 2 * NWV) {
      if (r != rb + wave) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = min(r + NWV * u, n - 1);
#pragma unroll
          for (int it = 0; it < ITERS; ++it) x[u][it] = *reinterpret_cast<const u32x4*>(base + (int64_t)row * d + it * 64 * V);
        }
      }
      float a[2] = {0.f, 0.f};
#pragma unroll
      for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int u = 0; u < 2; ++u) a[u] = dot_raw<T>(x[u][it], wr[it], a[u]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        a[u] = wave_sum(a[u]);
        const int row = r + NWV * u;
        if (lane == 0 && row < re) scores[out_map ? out_map[r0 + row] : r0 + row] = (a[u] + cs) / sqrt_hd;
      }
    }
  } else {
    for (int r = rb + wave; r < re; r += NWV) {
      const T* p0 = h + (r0 + r) * (int64_t)d;
      float a0 = 0.f;
      for (int col = lane * V; col < d; col += 64 * V) {
        float x0[V];
        load_vec(p0 + col, x0);
#pragma unroll
        for (int q = 0; q < V; q += 4) {
          float t4[4];
          load_vec(wl + col + q, t4);
#pragma unroll
          for (int z = 0; z < 4; ++z) a0 = fmaf(x0[q + z], t4[z], a0);
        }
      }
      a0 = wave_sum(a0);
      if (lane == 0) scores[out_map ? out_map[r0 + r] : r0 + r] = (a0 + cs) / sqrt_hd;
    }
  }
  VSEL_STAMP_DRAIN(3, 3);
}

// -------------------------------------------------------------------------------------------------------------------------
// P4  hard top-k + gather in one launch: every workgroup runs the segment's radix select itself, keeps the source rows of ITS
//     output rows, writes their idx entries and copies them.  Keys live in registers (KPT per thread, N <= 256 KPT);
//     wave w owns the CONTIGUOUS elements [w * span, (w + 1) * span), so the ordered compaction needs one exchange of wave
//     totals instead of two barriers per 256 elements; the four histograms are separate LDS arrays zeroed once and every wave
//     scans them itself: 7 barriers per workgroup in all.  Integer arithmetic only -> the same indices as topk_select_kernel
//     (larger key first, then lower index).  grid (ceil(k / rows_per_block), S), block 256.
//     Latency notes (tools/trace_small.py): the key loads are UNCONDITIONAL (clamped index) -- "load if in range" made hipcc
//     wait for each of the 9 loads in turn (3.0 us to the first barrier); the bin scan is a DPP prefix sum (six dependent
//     ds_bpermute round trips per pass before); the row copy keeps 8 x 16 B per thread in flight.
// -------------------------------------------------------------------------------------------------------------------------
template <typename T, int KPT>
__global__ __launch_bounds__(256) void select_gather_small_kernel(const T* __restrict__ h, const float* __restrict__ scores,
                                                                  SegView sv, int d, int64_t* __restrict__ idx,
                                                                  T* __restrict__ out, int rows_per_block,
                                                                  const int64_t* __restrict__ src_map) {
  constexpr int V = Elem<T>::kVec;
  constexpr int NW = kSmallSelectThreads / 64;
  const int s = blockIdx.y;
  const int n = sv.n_rows(s);
  const int ko = min(sv.n_out(s), n);
  const int jb = blockIdx.x * rows_per_block;
  if (jb >= ko) return;
  const int je = min(ko, jb + rows_per_block);
  const int64_t rb = sv.row_begin(s), ob = sv.out_begin(s);
  const float* sc = scores + rb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  __shared__ uint32_t hist[4][256];
  __shared__ uint32_t wtot[NW][2];
  __shared__ int rows[64];

  VSEL_STAMP(4, 0);
  const int kpw = (n + kSmallSelectThreads - 1) / kSmallSelectThreads;     // 64-element groups per wave (<= KPT)
  const int e0 = wave * kpw * 64 + lane;
  uint32_t key[KPT];
  {
    float raw[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) raw[j] = sc[min(e0 + 64 * j, n - 1)];
#pragma unroll
    for (int p = 0; p < 4; ++p) hist[p][tid] = 0;
#pragma unroll
    for (int j = 0; j < KPT; ++j) key[j] = (j < kpw && e0 + 64 * j < n) ? order_key(raw[j]) : 0u;
  }
  __syncthreads();
#ifdef VSEL_TRACE
  asm volatile("" ::"v"(key[0]), "v"(key[KPT - 1]));
#endif
  VSEL_STAMP(4, 1);
  uint32_t prefix = 0, maskbits = 0, kk = (uint32_t)ko;
#pragma unroll 1
  for (int pass = 3; pass >= 0; --pass) {        // NOT unrolled: the code of one pass is fetched once (instruction cache)
    const int shift = 8 * pass;
    uint32_t* hp = hist[pass];
#pragma unroll
    for (int j = 0; j < KPT; ++j)      // exec-masked, no branches ("if (j >= kpw) break" measured 0.8 vs 0.62 us per pass)
      if (j < kpw && e0 + 64 * j < n && (key[j] & maskbits) == prefix) atomicAdd(&hp[(key[j] >> shift) & 255u], 1u);
    __syncthreads();
    // every wave finds the bin where the count from the top reaches kk.  Lane L owns bins 255 - 4 L .. 252 - 4 L (top first), so
    // "keys in the bins above mine" is an exclusive prefix sum over the lanes
    const u32x4 cv = *reinterpret_cast<const u32x4*>(&hp[252 - 4 * lane]);
    const uint32_t cs[4] = {cv[3], cv[2], cv[1], cv[0]};
    const uint32_t tot = cs[0] + cs[1] + cs[2] + cs[3];
    uint32_t run = wave_prefix_sum_u32(tot) - tot;
    uint32_t found = 0xffffffffu;
    uint32_t found_kk = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (run < kk && run + cs[b] >= kk) {
        found = 255 - 4 * lane - b;
        found_kk = kk - run;
      }
      run += cs[b];
    }
    const unsigned long long who = __ballot(found != 0xffffffffu);      // exactly one lane
    const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
    const uint32_t bin = (uint32_t)__builtin_amdgcn_readlane((int)found, src);
    kk = (uint32_t)__builtin_amdgcn_readlane((int)found_kk, src);
    prefix |= bin << shift;
    maskbits |= 0xffu << shift;
    VSEL_STAMP(5, pass);
  }
  const uint32_t thr = prefix, need = kk;
  VSEL_STAMP(4, 2);
  // ordered compaction, only as far as THIS workgroup's output rows [jb, je) need it: positions are monotone in the element
  // index, so a wave / a 64-element group whose position range misses [jb, je) is skipped with scalar arithmetic on the
  // ballots (computing every position in every workgroup was 1.3 us).  Every workgroup writes the idx entries of its rows.
  unsigned long long bgt[KPT], beq[KPT];
  uint32_t my_gt = 0, my_eq = 0;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const bool valid = j < kpw && e0 + 64 * j < n;
    bgt[j] = __ballot(valid && key[j] > thr);
    beq[j] = __ballot(valid && key[j] == thr);
    my_gt += __popcll(bgt[j]);
    my_eq += __popcll(beq[j]);
  }
  VSEL_STAMP(4, 5);
  if (lane == 0) { wtot[wave][0] = my_gt; wtot[wave][1] = my_eq; }
  __syncthreads();
  VSEL_STAMP(4, 6);
  uint32_t run_gt = 0, run_eq = 0;
#pragma unroll
  for (int wv = 0; wv < NW; ++wv)
    if (wv < wave) { run_gt += wtot[wv][0]; run_eq += wtot[wv][1]; }
  run_gt = (uint32_t)__builtin_amdgcn_readfirstlane((int)run_gt);
  run_eq = (uint32_t)__builtin_amdgcn_readfirstlane((int)run_eq);
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const uint32_t cg = (uint32_t)__popcll(bgt[j]), ce = (uint32_t)__popcll(beq[j]);
    // selected elements of this group occupy positions [p0, p1)
    const uint32_t p0 = run_gt + min(run_eq, need), p1 = run_gt + cg + min(run_eq + ce, need);
    if (p1 > (uint32_t)jb && p0 < (uint32_t)je) {          // uniform
      const int e = e0 + 64 * j;
      const bool gt = (bgt[j] >> lane) & 1ull;
      const bool eq = (beq[j] >> lane) & 1ull;
      const uint32_t gt_before = run_gt + __popcll(bgt[j] & below), eq_before = run_eq + __popcll(beq[j] & below);
      if (gt || (eq && eq_before < need)) {
        const int pos = (int)(gt_before + min(eq_before, need));
        if (pos >= jb && pos < je) {
          rows[pos - jb] = e;
          idx[ob + pos] = (int64_t)e;
        }
      }
    }
    run_gt += cg;
    run_eq += ce;
  }
  VSEL_STAMP(4, 7);
  __syncthreads();
  VSEL_STAMP(4, 3);
  // row copy: 4 rows x 2 column groups of 256 x 16 B in flight per thread (clamped duplicates past the end hit L1)
  const int vpr = d / V;
  for (int j0 = jb; j0 < je; j0 += 4) {
    const u32x4* sp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t lsrc = rb + __builtin_amdgcn_readfirstlane(rows[min(j0 + u, je - 1) - jb]);
      const int64_t src = src_map ? src_map[lsrc] : lsrc;
      sp[u] = reinterpret_cast<const u32x4*>(h + src * d);
    }
    for (int v0 = 0; v0 < vpr; v0 += 512) {
      const int va = v0 + tid, vb = v0 + 256 + tid;
      const int vac = min(va, vpr - 1), vbc = min(vb, vpr - 1);
      u32x4 xa[4], xb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xa[u] = sp[u][vac];
        xb[u] = sp[u][vbc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j0 + u < je) {
          u32x4* dp = reinterpret_cast<u32x4*>(out + (ob + j0 + u) * d);
          if (va < vpr) dp[va] = xa[u];
          if (vb < vpr) dp[vb] = xb[u];
        }
      }
    }
  }
  VSEL_STAMP_DRAIN(4, 4);
}

// -------------------------------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------------------------------
inline bool small_path_ok(const vsel_segments* seg, const vsel_scorer* sc, const LisPlan& p) {
  return seg->n_seg <= kSmallMaxSeg && sc->wdtype == VSEL_BF16 && p.mfma_bf16 && p.kslice1 <= kSliceNT && p.kslice2 <= kSliceNN &&
         sc->d % 8 == 0 && sc->hd % 8 == 0;
}

// sweep 1 partials (or the producer's column sums) -> part2 slabs + c partials: two launches
inline int run_proj_small(hipStream_t st, const vsel_segments* seg, const vsel_scorer* sc, char* ws, const LisPlan& p,
                          const float* col_sums) {
  const int d = (int)sc->d, hd = (int)sc->hd, S = (int)seg->n_seg;
  const float* partial = col_sums ? col_sums : (const float*)(ws + p.off_partial);
  const int row_splits = col_sums ? 1 : p.row_splits;
  float* part1 = (float*)(ws + p.off_part1);
  float* part2 = (float*)(ws + p.off_part2);
  float* cpart = (float*)(ws + p.off_cpart);
  VSEL_LAUNCH(proj_nt_small_kernel, dim3((unsigned)cdiv(hd, 32), p.ks1), dim3(64), 0, st, partial, make_view(seg), S,
                     row_splits, (const uint16_t*)sc->wk, hd, d, p.kslice1, part1, (int64_t)d, 1);
  VSEL_AFTER_LAUNCH(st, "proj_nt_small_kernel");
  VSEL_LAUNCH(proj_nn_small_kernel, dim3((unsigned)cdiv(d, 128), p.ks2), dim3(64), 0, st, part1, p.ks1, S,
                     (const uint16_t*)sc->bk, (const uint16_t*)sc->bq, (const uint16_t*)sc->wq, d, hd, p.kslice2, part2, cpart,
                     p.n_cpart);
  VSEL_AFTER_LAUNCH(st, "proj_nn_small_kernel");
  return VSEL_OK;
}

template <typename T>
inline int run_score_small(hipStream_t st, const T* h, const vsel_segments* seg, const vsel_scorer* sc, char* ws, const LisPlan& p,
                           float* scores, const int64_t* out_map) {
  constexpr int V = Elem<T>::kVec;
  const int d = (int)sc->d, S = (int)seg->n_seg;
  // every workgroup re-reads KS2 x d x 4 bytes of slabs from L2: keep the grid near one workgroup per CU
  int64_t rpb = 64;
  while (rpb > 8 && seg->n_seg * cdiv(seg->rows_per_seg, rpb) < 128) rpb >>= 1;    // 72 / 144 / 288 / 576 workgroups at one
                                                                                   // image: 13.3 / 11.7 / 11.8 / 17.7 us
  const dim3 grid((unsigned)cdiv(seg->rows_per_seg, rpb), (unsigned)S);
  const float sq = (float)sqrt((double)sc->hd);
  const size_t lds = ((size_t)d + 16) * sizeof(float);
  const float* part2 = (const float*)(ws + p.off_part2);
  const float* cpart = (const float*)(ws + p.off_cpart);
  const int iters = (d % (64 * V) == 0) ? d / (64 * V) : 0;
#define VSEL_SCORE_SMALL_CASE(I)                                                                                          \
  case I:                                                                                                                 \
    VSEL_LAUNCH((score_small_kernel<T, I>), grid, dim3(kSmallScoreThreads), lds, st, h, make_view(seg), d, S, part2, p.ks2, cpart, \
                       p.n_cpart, sq, (int)rpb, scores, out_map);                                                         \
    break;
  switch (iters) {
    VSEL_SCORE_SMALL_CASE(1) VSEL_SCORE_SMALL_CASE(2) VSEL_SCORE_SMALL_CASE(3) VSEL_SCORE_SMALL_CASE(4)
    VSEL_SCORE_SMALL_CASE(5) VSEL_SCORE_SMALL_CASE(6) VSEL_SCORE_SMALL_CASE(7) VSEL_SCORE_SMALL_CASE(8)
    default:
      VSEL_LAUNCH((score_small_kernel<T, 0>), grid, dim3(kSmallScoreThreads), lds, st, h, make_view(seg), d, S, part2, p.ks2, cpart,
                         p.n_cpart, sq, (int)rpb, scores, out_map);
  }
#undef VSEL_SCORE_SMALL_CASE
  VSEL_AFTER_LAUNCH(st, "score_small_kernel");
  return VSEL_OK;
}

template <typename T>
inline int launch_select_gather_small(hipStream_t st, const T* h, int d, const vsel_segments* seg, const float* scores,
                                      int64_t* idx, T* out, const int64_t* src_map) {
  const int64_t maxn = seg->rows_per_seg;
  if (maxn > 32 * kSmallSelectThreads) {          // keys do not fit the register file: the two-launch form
    int rc = launch_select(st, scores, seg, idx, nullptr);
    if (rc) return rc;
    return launch_gather<T>(st, h, d, seg, idx, out, src_map);
  }
  int64_t rpb = 16;
  while (rpb > 2 && seg->n_seg * cdiv(seg->k, rpb) < 128) rpb >>= 1;              // 58 / 115 / 230 / 460 workgroups at one image:
                                                                                   // 14.7 / 12.1 / 12.1 / 13.7 us
  const dim3 grid((unsigned)cdiv(seg->k, rpb), (unsigned)seg->n_seg);
  const SegView sv = make_view(seg);
  if (maxn <= 12 * kSmallSelectThreads)
    VSEL_LAUNCH((select_gather_small_kernel<T, 12>), grid, dim3(256), 0, st, h, scores, sv, d, idx, out, (int)rpb, src_map);
  else
    VSEL_LAUNCH((select_gather_small_kernel<T, 32>), grid, dim3(256), 0, st, h, scores, sv, d, idx, out, (int)rpb, src_map);
  VSEL_AFTER_LAUNCH(st, "select_gather_small_kernel");
  return VSEL_OK;
}

}  // namespace vsel
