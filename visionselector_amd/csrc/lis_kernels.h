// LIS inference path for gfx950: column mean -> kbar / w projections (fp32 MFMA) -> scores ->
// radix-select hard top-k with order-preserving compaction -> row gather.
//
// Formulation (exact algebra, SURVEY.md section 7 hard part 2): the reference computes
//   k = Wk x + bk, q = Wq x + bq, s_i = mean_j(q_i . k_j) / sqrt(Hd)
// (reference: qwen-vl-finetune/compression_method/selector_scorer.py:47-53).  mean_j(q_i . k_j) =
// q_i . kbar with kbar = Wk xbar + bk, so
//   s_i = (x_i . w + c) / sqrt(Hd),   w = Wq^T kbar,  c = bq . kbar,  xbar = mean_i x_i.
// That turns 78 GFLOP of GEMM into two sweeps over the token tensor (HBM-bound) plus two skinny
// projections [S, D] x [D, Hd] and [S, Hd] x [Hd, D] which run on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation, so index selection can be compared
// bit-for-bit with the fp32 CPU oracle on tie-free inputs).
#pragma once
#include "common.h"
#ifndef VSEL_PROJ_WAVES
#define VSEL_PROJ_WAVES 4
#endif
#include "proj_bf16x3.h"
#include <algorithm>
#include <type_traits>
#include <math.h>

namespace vsel {

// =================================================================================================
constexpr int kProjWaves = VSEL_PROJ_WAVES;   // waves per workgroup of the bf16x3 projection GEMMs (independent waves sharing one operand tile through L1)
constexpr int kRowsPerChunk = 128;   // sweep-1 row chunk (batch-invariant summation order)
constexpr int kSliceNT = 256;        // split-K slice of the kbar projection (fixed => batch-invariant)
constexpr int kSliceNN = 128;        // split-K slice of the w projection

// K1  partial column sums.  grid (col_tiles, row_splits, n_seg), block 256 (4 waves).
//     A wave reads 64 lanes x 16 B = 1 KiB contiguous of one row; the block's 4 waves interleave rows and reduce through
//     LDS.  (A barrier-free variant -- each wave owning a whole column tile of the chunk -- is 2 % faster in isolation
//     but 2 % slower inside the pipeline, where sweep 1 overlaps the write-back of the previous gather; tools/membw2.hip
//     shows the bare access pattern reaches the 6.2-6.3 TB/s ceiling either way.)
// =================================================================================================
// NT: non-temporal loads (the token tensor is streamed once per sweep and does not fit the caches); chosen per launch by
// stream_policy() -- below ~384 MB the default policy is faster because sweep 2 and the gather hit what sweep 1 left in the
// Infinity Cache (tools/exp_small_batch.py: B = 8 images 86 vs 96 us, B = 32 271 vs 238 us).  Results are identical.
template <typename T> __device__ __forceinline__ void unpack_vec(u32x4 raw, float (&v)[Elem<T>::kVec]);
template <> __device__ __forceinline__ void unpack_vec<bf16_t>(u32x4 raw, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(raw[i] << 16);
    v[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void unpack_vec<float>(u32x4 raw, float (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(raw[i]);
}

template <typename T, bool NT, bool DEEP = false>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ h, SegView sv, int d,
                                                             int row_splits, float* __restrict__ partial) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.z, rs = blockIdx.y;
  const int col = (blockIdx.x * 64 + lane) * V;
  const int n = sv.n_rows(s);
  const int64_t r0 = sv.row_begin(s);
  // fixed 128-row chunks: a segment's partial sums (and therefore its scores, bit for bit) do not depend on what
  // else is in the batch; chunks past the segment's end contribute exact zeros
  const int rb = rs * kRowsPerChunk;
  const int re = min(n, rb + kRowsPerChunk);
  VSEL_STAMP(0, 0);
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
  auto ld = [](const T* p, float (&v)[V]) {
    if constexpr (NT) load_vec_stream(p, v);
    else load_vec(p, v);
  };
  if (col < d) {
    const T* base = h + (r0 * (int64_t)d + col);
    int r = rb + wave;
    if constexpr (DEEP) {
      // a handful of segments (latency-bound, one wave per SIMD): the wave's whole share of a full chunk -- 32 rows -- in flight
      // at once, as raw 16-byte vectors; added in exactly the order of the loop below
      if (r + 124 < re) {
        u32x4 raw[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) raw[q] = *reinterpret_cast<const u32x4*>(base + (int64_t)(r + 4 * q) * d);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[8][V];
#pragma unroll
          for (int q = 0; q < 8; ++q) unpack_vec<T>(raw[8 * g + q], v[q]);
#pragma unroll
          for (int i = 0; i < V; ++i) {
            acc[i] = ((acc[i] + v[0][i]) + v[1][i]) + (v[2][i] + v[3][i]);
            acc[i] = ((acc[i] + v[4][i]) + v[5][i]) + (v[6][i] + v[7][i]);
          }
        }
        r += 128;
      }
    }
    // 8 rows in flight per wave (same-box A/B at B = 128: ~1 % faster than 4).  Summation order = two consecutive 4-row steps,
    // so results are unchanged.
    for (; r + 28 < re; r += 32) {
      float v0[V], v1[V], v2[V], v3[V], v4[V], v5[V], v6[V], v7[V];
      ld(base + (int64_t)r * d, v0);
      ld(base + (int64_t)(r + 4) * d, v1);
      ld(base + (int64_t)(r + 8) * d, v2);
      ld(base + (int64_t)(r + 12) * d, v3);
      ld(base + (int64_t)(r + 16) * d, v4);
      ld(base + (int64_t)(r + 20) * d, v5);
      ld(base + (int64_t)(r + 24) * d, v6);
      ld(base + (int64_t)(r + 28) * d, v7);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        acc[i] = ((acc[i] + v0[i]) + v1[i]) + (v2[i] + v3[i]);
        acc[i] = ((acc[i] + v4[i]) + v5[i]) + (v6[i] + v7[i]);
      }
    }
    for (; r + 12 < re; r += 16) {
      float v0[V], v1[V], v2[V], v3[V];
      ld(base + (int64_t)r * d, v0);
      ld(base + (int64_t)(r + 4) * d, v1);
      ld(base + (int64_t)(r + 8) * d, v2);
      ld(base + (int64_t)(r + 12) * d, v3);
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] = ((acc[i] + v0[i]) + v1[i]) + (v2[i] + v3[i]);
    }
    for (; r < re; r += 4) {
      float v0[V];
      ld(base + (int64_t)r * d, v0);
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] += v0[i];
    }
  }
  __shared__ float red[4][64][V + 1];
#pragma unroll
  for (int i = 0; i < V; ++i) red[wave][lane][i] = acc[i];
  __syncthreads();
  VSEL_STAMP(0, 1);
  if (wave == 0 && col < d) {
    float out[V];
#pragma unroll
    for (int i = 0; i < V; ++i) out[i] = (red[0][lane][i] + red[1][lane][i]) + (red[2][lane][i] + red[3][lane][i]);
    float* dst = partial + ((int64_t)(s * row_splits + rs) * d + col);
#pragma unroll
    for (int i = 0; i < V; i += 4) {
      f32x4 o = {out[i], out[i + 1], out[i + 2], out[i + 3]};
      *reinterpret_cast<f32x4*>(dst + i) = o;
    }
  }
  VSEL_STAMP_DRAIN(0, 2);
}

// K1 for many segments: ONE WAVE owns a (segment, 64-lane column slab), streams it from the first row to the last and plays the
// four waves of colsum_partial_kernel itself -- accumulator set w takes rows w, w + 4, ... of each 128-row chunk, four rows at a
// time ((acc + v0) + v1) + (v2 + v3), the last < 4 one by one; per chunk (set 0 + set 1) + (set 2 + set 3); across chunks the
// finish kernels' order (acc = 0; acc += chunk 0, 1, ...).  The segment's sums therefore come out bit-identical to partials +
// finish, with no LDS, no barrier and no [S][chunks][D] partial round trip.  Why it pays (B = 128, 7B geometry, same-box numbers in
// DESIGN.md section 5): the chunked form is 16 128 short workgroups that each end in a cross-wave reduction behind a barrier and
// write 33 MB of partials in 2-KiB pieces into the read stream -- 335-360 us; without the epilogue and the stores the same loads
// take 306-313 us; this kernel reads the tensor in ~301 us, a little under the score sweep.  The four waves of a workgroup take
// four ADJACENT slabs of the same rows.  32 rows (32 x 16 B per lane) in flight per wave.  Output: sums[s][col] = the finish
// kernels' input with row_splits = 1.  grid ceil(col_tiles * n_seg / 4), block 256 (four independent waves).  One wave streams a
// whole segment slab, so this needs many (segment, slab) pairs -- seg_sums_form() below.
// NW = 32-bit words per lane and load (4 / 2 / 1: 16- / 8- / 4-byte loads, i.e. 512 / 256 / 128 bf16 columns per slab): narrower slabs
// give more (segment, slab) pairs -- more waves -- for smaller batches; the rows in flight grow to keep 128 registers of loads.
template <typename T, bool NT, int NW>
__global__ __launch_bounds__(256) void colsum_seg_kernel(const T* __restrict__ h, SegView sv, int d, int col_tiles, int n_seg,
                                                         float* __restrict__ sums, uint16_t* __restrict__ xs) {
  constexpr int V = NW * (sizeof(T) == 2 ? 2 : 1);               // elements per lane
  constexpr int RIF = 32 * (4 / NW);                             // rows in flight in the main loop: 32 / 64 / 128
  typedef uint32_t raw_t __attribute__((ext_vector_type(NW)));
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);         // (segment, slab), wave-uniform
  if (unit >= col_tiles * n_seg) return;
  const int s = unit / col_tiles;
  const int col = ((unit - s * col_tiles) * 64 + lane) * V;
  const int n = sv.n_rows(s);
  const int64_t r0 = sv.row_begin(s);
  const bool col_ok = col < d;
  const T* base = h + (r0 * (int64_t)d + (col_ok ? col : 0));
  auto ldraw = [](const T* p) -> raw_t {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(p));
    else return *reinterpret_cast<const raw_t*>(p);
  };
  auto unpack = [](raw_t raw, float (&v)[V]) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      uint32_t wd;
      wd = raw[i];                                            // (a 1-element ext vector indexes like the others)
      if constexpr (sizeof(T) == 2) {
        v[2 * i] = __uint_as_float(wd << 16);
        v[2 * i + 1] = __uint_as_float(wd & 0xffff0000u);
      } else {
        v[i] = __uint_as_float(wd);
      }
    }
  };
  float tot[V];
#pragma unroll
  for (int i = 0; i < V; ++i) tot[i] = 0.f;
  // (chunks past the segment's end, which the chunked form adds as exact zeros, are skipped: tot + 0.0f == tot, and tot is never
  // -0.0f because it starts from +0.0f)
  for (int rb = 0; rb < n; rb += kRowsPerChunk) {
    const int re = min(n, rb + kRowsPerChunk);
    float acc[4][V];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int i = 0; i < V; ++i) acc[w][i] = 0.f;
    // four-row steps of every set over the rows blk .. blk + 16 G - 1 (each set: G consecutive steps, in row order)
    auto steps = [&](auto g_c, int blk) __attribute__((always_inline)) {
      constexpr int G = decltype(g_c)::value;
      raw_t x[16 * G];
#pragma unroll
      for (int q = 0; q < 16 * G; ++q) x[q] = ldraw(base + (int64_t)(blk + q) * d);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float v0[V], v1[V], v2[V], v3[V];
          unpack(x[16 * g + w], v0);
          unpack(x[16 * g + w + 4], v1);
          unpack(x[16 * g + w + 8], v2);
          unpack(x[16 * g + w + 12], v3);
#pragma unroll
          for (int i = 0; i < V; ++i) acc[w][i] = ((acc[w][i] + v0[i]) + v1[i]) + (v2[i] + v3[i]);
        }
    };
    int b = rb;
    if constexpr (RIF > 32)
      for (; b + RIF - 1 < re; b += RIF) steps(std::integral_constant<int, RIF / 16>{}, b);
    for (; b + 31 < re; b += 32) steps(std::integral_constant<int, 2>{}, b);   // every set: two full steps in rows b .. b + 31
#pragma unroll
    for (int w = 0; w < 4; ++w) {                                  // the segment's last, partial chunk: set by set
      int r = b + w;
      for (; r + 12 < re; r += 16) {
        float v0[V], v1[V], v2[V], v3[V];
        unpack(ldraw(base + (int64_t)r * d), v0);
        unpack(ldraw(base + (int64_t)(r + 4) * d), v1);
        unpack(ldraw(base + (int64_t)(r + 8) * d), v2);
        unpack(ldraw(base + (int64_t)(r + 12) * d), v3);
#pragma unroll
        for (int i = 0; i < V; ++i) acc[w][i] = ((acc[w][i] + v0[i]) + v1[i]) + (v2[i] + v3[i]);
      }
      for (; r < re; r += 4) {
        float v0[V];
        unpack(ldraw(base + (int64_t)r * d), v0);
#pragma unroll
        for (int i = 0; i < V; ++i) acc[w][i] += v0[i];
      }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) tot[i] += (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
  }
  if (col_ok) {
    float* dst = sums + ((int64_t)s * d + col);
#pragma unroll
    for (int i = 0; i < V; ++i) dst[i] = tot[i];                   // (V consecutive floats: the compiler merges them into one or two stores)
    if (xs) {
      // ... and x-bar = sums / N split into the three bf16 planes of the MFMA projections, fragment-major: what
      // colsum_finish_split_kernel would compute from the sums ((0 + sum) / N, the same bits), one launch less.  A lane's V
      // columns are consecutive elements e of ONE fragment row (V divides 8).
      const float fn = (float)n;
      const int mt = (n_seg + 31) >> 5, ksteps = d >> 4;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        uint16_t* dstp = xs + frag_off(pl, s, col, mt, ksteps);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          uint32_t bt[3];
          split3((0.f + tot[i]) / fn, bt[0], bt[1], bt[2]);
          dstp[i] = (uint16_t)bt[pl];
        }
      }
    }
  }
}

// K1b  xbar[s][c] = sum_rs partial[s][rs][c] / N_s   (fixed order)
static __attribute__((unused)) __global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, SegView sv, int d,
                                                            int row_splits, float* __restrict__ xbar) {
  const int s = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const float* p = partial + (int64_t)s * row_splits * d + c;
  float acc = 0.f;
  for (int rs = 0; rs < row_splits; ++rs) acc += p[(int64_t)rs * d];
  xbar[(int64_t)s * d + c] = acc / (float)sv.n_rows(s);
}

// =================================================================================================
// K2  skinny GEMM "NT":  part[ks][m][n] = sum_{k in slice ks} x[m][k] * w[n][k]
//     x fp32 [M, K], w (bf16|f32) [N, K] row-major.  One wave per (n-tile 32, m-tile 32, k-slice).
//     v_mfma_f32_32x32x2_f32: A[i][kk] = w[n0+i][.], B[kk][j] = x[m0+j][.]; lane = 32*kk + i.
//     Each lane owns 8 consecutive k per 16-wide step (16-B weight load), MFMA step t pairs element t.
// =================================================================================================
template <typename TW>
__device__ __forceinline__ void load8_guard(const TW* row, int kb, int k_end, bool row_ok, float (&v)[8]) {
  constexpr int V = Elem<TW>::kVec;
  if (row_ok && kb + 8 <= k_end) {
    if constexpr (V == 8) {
      load_vec(row + kb, v);
    } else {
      float a[4], b[4];
      load_vec(row + kb, a);
      load_vec(row + kb + 4, b);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (row_ok && kb + i < k_end) ? load_elem(row + kb + i) : 0.f;
  }
}

template <typename TW>
__global__ __launch_bounds__(64) void gemm_nt_kernel(const float* __restrict__ x, const TW* __restrict__ w, int M,
                                                     int N, int K, int kslice, float* __restrict__ part) {
  const int lane = threadIdx.x;
  const int i = lane & 31, kk = lane >> 5;
  const int n_row = blockIdx.x * 32 + i;
  const int m_row = blockIdx.y * 32 + i;
  const int ks = blockIdx.z;
  const int k_begin = ks * kslice;
  const int k_end = min(K, k_begin + kslice);
  const bool n_ok = n_row < N, m_ok = m_row < M;
  const TW* wrow = w + (int64_t)(n_ok ? n_row : 0) * K;
  const float* xrow = x + (int64_t)(m_ok ? m_row : 0) * K;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = k_begin; k0 < k_end; k0 += 16) {
    const int kb = k0 + 8 * kk;
    float a[8], b[8];
    load8_guard<TW>(wrow, kb, k_end, n_ok, a);
    load8_guard<float>(xrow, kb, k_end, m_ok, b);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
  }
  // C[row = n][col = m]: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int m = blockIdx.y * 32 + (lane & 31);
  if (m < M) {
    float* dst = part + ((int64_t)ks * M + m) * N + blockIdx.x * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
      if ((int)blockIdx.x * 32 + row < N) dst[row] = acc[r];
    }
  }
}

// K2b  kbar[m][n] = sum_ks part[ks][m][n] + bk[n];  c[m] = sum_n bq[n] * kbar[m][n].  One block per m.
template <typename TW>
__global__ __launch_bounds__(256) void kbar_finish_kernel(const float* __restrict__ part, int KS, int M, int N,
                                                          const TW* __restrict__ bk, const TW* __restrict__ bq,
                                                          float* __restrict__ kbar, float* __restrict__ c) {
  const int m = blockIdx.x;
  float cacc = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    float v = 0.f;
    for (int k0 = 0; k0 < KS; k0 += 16) {        // 16 slab loads in flight (clamped, unconditional), added in ks order: the plain
      float t[16];                               // loop was KS dependent round trips per element (29 us at Hd = 1792)
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = part[((int64_t)min(k0 + u, KS - 1) * M + m) * N + n];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (k0 + u < KS) v += t[u];
    }
    v += load_elem(bk + n);
    kbar[(int64_t)m * N + n] = v;
    cacc += load_elem(bq + n) * v;
  }
  cacc = wave_sum(cacc);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cacc;
  __syncthreads();
  if (threadIdx.x == 0) c[m] = (red[0] + red[1]) + (red[2] + red[3]);
}

// =================================================================================================
// K3  skinny GEMM "NN":  part[ks][m][n] = sum_{k in slice} x[m][k] * w[k][n],  w (bf16|f32) [K, N].
//     One wave per (n-tile 256, m-tile 32, k-slice).  Lane (i, kk) loads 8 consecutive n of row k
//     (16-B coalesced) and feeds 8 accumulators: acc[t] is the 32x32 tile for n = n0 + 8*i' + t.
// =================================================================================================
template <typename TW>
__global__ __launch_bounds__(64) void gemm_nn_kernel(const float* __restrict__ x, const TW* __restrict__ w, int M,
                                                     int N, int K, int kslice, float* __restrict__ part) {
  const int lane = threadIdx.x;
  const int i = lane & 31, kk = lane >> 5;
  const int nbase = blockIdx.x * 256 + 8 * i;
  const int m_row = blockIdx.y * 32 + i;
  const int ks = blockIdx.z;
  const int k_begin = ks * kslice;
  const int k_end = min(K, k_begin + kslice);
  const bool n_ok = nbase < N, m_ok = m_row < M;   // N % 8 == 0 (checked on the host)
  const float* xrow = x + (int64_t)(m_ok ? m_row : 0) * K;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int k0 = k_begin; k0 < k_end; k0 += 8) {
    const int kb = k0 + 4 * kk;
    float xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = (m_ok && kb + u < k_end) ? xrow[kb + u] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float wv[8];
      const int k = kb + u;
      if (n_ok && k < k_end) {
        const TW* p = w + (int64_t)k * N + nbase;
        if constexpr (Elem<TW>::kVec == 8) {
          load_vec(p, wv);
        } else {
          float a[4], b[4];
          load_vec(p, a);
          load_vec(p + 4, b);
#pragma unroll
          for (int q = 0; q < 4; ++q) { wv[q] = a[q]; wv[4 + q] = b[q]; }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) wv[q] = 0.f;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t], xv[u], acc[t], 0, 0, 0);
    }
  }
  const int m = blockIdx.y * 32 + (lane & 31);
  if (m < M) {
    float* dst = part + ((int64_t)ks * M + m) * N + blockIdx.x * 256;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int irow = (r & 3) + 8 * (r >> 2) + 4 * kk;
      const int n = blockIdx.x * 256 + 8 * irow;
      if (n < N) {
        f32x4 lo = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        f32x4 hi = {acc[4][r], acc[5][r], acc[6][r], acc[7][r]};
        *reinterpret_cast<f32x4*>(dst + 8 * irow) = lo;
        *reinterpret_cast<f32x4*>(dst + 8 * irow + 4) = hi;
      }
    }
  }
}

// K3b  out[e] = sum_ks part[ks][e]
static __attribute__((unused)) __global__ __launch_bounds__(256) void slice_sum_kernel(const float* __restrict__ part, int KS, int64_t count,
                                                        float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  float v = 0.f;
  for (int ks = 0; ks < KS; ++ks) v += part[(int64_t)ks * count + e];
  out[e] = v;
}

template <typename T>
__device__ __forceinline__ float dot_raw(u32x4 raw, const float (&w)[Elem<T>::kVec], float acc);
template <>
__device__ __forceinline__ float dot_raw<bf16_t>(u32x4 raw, const float (&w)[8], float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc = fmaf(__uint_as_float(raw[i] << 16), w[2 * i], acc);
    acc = fmaf(__uint_as_float(raw[i] & 0xffff0000u), w[2 * i + 1], acc);
  }
  return acc;
}
template <>
__device__ __forceinline__ float dot_raw<float>(u32x4 raw, const float (&w)[4], float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = fmaf(__uint_as_float(raw[i]), w[i], acc);
  return acc;
}

// =================================================================================================
// K4  scores[row] = (x_row . w[s] + c[s]) / sqrt(Hd).  grid (row_chunks, n_seg), block 256.
//     ITERS > 0: D == ITERS * 64 * V exactly and w[s] lives in registers; ITERS == 0: generic.
// =================================================================================================
// body shared by score_kernel (w, c from global memory) and score_small_kernel (lis_small.h: w, c rebuilt in LDS by the block):
// rows [rb, re) of segment s, ws = the segment's w [d], cs = its c.
template <typename T, int ITERS, bool NT>
__device__ __forceinline__ void score_rows(const T* __restrict__ h, int d, const float* ws, float cs, float sqrt_hd, int64_t r0,
                                           int rb, int re, float* __restrict__ scores, const int64_t* __restrict__ out_map) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (ITERS > 0) {
    float wr[ITERS][V];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int q = 0; q < V; q += 4) {
        float t4[4];
        load_vec(ws + (it * 64 + lane) * V + q, t4);
#pragma unroll
        for (int z = 0; z < 4; ++z) wr[it][q + z] = t4[z];
      }
    }
    // rows are kept as raw 16-byte vectors until the FMAs (4 VGPRs per load instead of 8 converted floats)
    auto ldraw = [](const T* p) -> u32x4 {
      if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
      else return *reinterpret_cast<const u32x4*>(p);
    };
    const T* base = h + r0 * (int64_t)d + lane * V;
    int r = rb + wave;
    for (; r + 12 < re; r += 16) {       // 4 rows in flight per wave (same-box A/B: ~1 % faster than 2)
      u32x4 x[4][ITERS];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
          x[u][it] = ldraw(base + (int64_t)(r + 4 * u) * d + it * 64 * V);
      float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = dot_raw<T>(x[u][it], wr[it], a[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = wave_sum(a[u]);
        if (lane == 0) scores[out_map ? out_map[r0 + r + 4 * u] : r0 + r + 4 * u] = (a[u] + cs) / sqrt_hd;
      }
    }
    for (; r + 4 < re; r += 8) {
      const T* p0 = base + (int64_t)r * d;
      const T* p1 = base + (int64_t)(r + 4) * d;
      u32x4 x0[ITERS], x1[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) x0[it] = ldraw(p0 + it * 64 * V);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) x1[it] = ldraw(p1 + it * 64 * V);
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        a0 = dot_raw<T>(x0[it], wr[it], a0);
        a1 = dot_raw<T>(x1[it], wr[it], a1);
      }
      a0 = wave_sum(a0);
      a1 = wave_sum(a1);
      if (lane == 0) {
        scores[out_map ? out_map[r0 + r] : r0 + r] = (a0 + cs) / sqrt_hd;
        scores[out_map ? out_map[r0 + r + 4] : r0 + r + 4] = (a1 + cs) / sqrt_hd;
      }
    }
    for (; r < re; r += 4) {
      const T* p0 = base + (int64_t)r * d;
      u32x4 x0[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) x0[it] = ldraw(p0 + it * 64 * V);
      float a0 = 0.f;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) a0 = dot_raw<T>(x0[it], wr[it], a0);
      a0 = wave_sum(a0);
      if (lane == 0) scores[out_map ? out_map[r0 + r] : r0 + r] = (a0 + cs) / sqrt_hd;
    }
  } else {
    for (int r = rb + wave; r < re; r += 4) {
      const T* p0 = h + (r0 + r) * (int64_t)d;
      float a0 = 0.f;
      for (int col = lane * V; col < d; col += 64 * V) {
        float x0[V];
        load_vec(p0 + col, x0);
#pragma unroll
        for (int q = 0; q < V; q += 4) {
          float t4[4];
          load_vec(ws + col + q, t4);
#pragma unroll
          for (int z = 0; z < 4; ++z) a0 = fmaf(x0[q + z], t4[z], a0);
        }
      }
      a0 = wave_sum(a0);
      if (lane == 0) scores[out_map ? out_map[r0 + r] : r0 + r] = (a0 + cs) / sqrt_hd;
    }
  }
}

template <typename T, int ITERS, bool NT>
__global__ __launch_bounds__(256) void score_kernel(const T* __restrict__ h, SegView sv, int d,
                                                    const float* __restrict__ w, const float* __restrict__ c,
                                                    float sqrt_hd, int rows_per_block, float* __restrict__ scores,
                                                    const int64_t* __restrict__ out_map) {
  const int s = blockIdx.y;
  const int n = sv.n_rows(s);
  const int rb = blockIdx.x * rows_per_block;
  if (rb >= n) return;
  score_rows<T, ITERS, NT>(h, d, w + (int64_t)s * d, c[s], sqrt_hd, sv.row_begin(s), rb, min(n, rb + rows_per_block), scores, out_map);
}

// =================================================================================================
// K5  hard top-k per segment: 4-pass radix-256 select of the k-th largest key, then one ordered
//     compaction pass (ballot + popcount prefix) that emits ascending indices -- the reference's
//     topk(k).indices.sort() without a sort.  One block of 1024 threads per segment.
// =================================================================================================
__device__ __forceinline__ uint32_t order_key(float x) {
  const float f = x + 0.0f;  // -0.0 -> +0.0
  const uint32_t u = __float_as_uint(f);
  if (f != f) return 0xffffffffu;  // NaN sorts greatest (torch.topk convention)
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static __attribute__((unused)) __global__ __launch_bounds__(1024) void topk_select_kernel(const float* __restrict__ scores, SegView sv,
                                                           int64_t* __restrict__ idx, float* __restrict__ mask) {
  const int s = blockIdx.x;
  const int n = sv.n_rows(s);
  const int k = min(sv.n_out(s), n);
  const float* sc = scores + sv.row_begin(s);
  int64_t* out = idx ? idx + sv.out_begin(s) : nullptr;
  float* mk = mask ? mask + sv.row_begin(s) : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_bin, sh_kk;
  __shared__ uint32_t wtot_gt[16], wtot_eq[16];

  uint32_t prefix = 0, maskbits = 0, kk = (uint32_t)k;
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const int shift = 8 * pass;
    for (int i = tid; i < n; i += 1024) {
      const uint32_t key = order_key(sc[i]);
      if ((key & maskbits) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      // lane owns bins 4*lane .. 4*lane+3; find the bin where the count from the top reaches kk
      const uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
      const uint32_t tot = c0 + c1 + c2 + c3;
      uint32_t suf = tot;  // inclusive suffix sum over lanes (lanes >= this one)
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_down(suf, off, 64);
        if (lane + off < 64) suf += o;
      }
      uint32_t run = suf - tot;  // keys in bins above this lane's bins
      const uint32_t cs[4] = {c3, c2, c1, c0};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (run < kk && run + cs[b] >= kk) {
          sh_bin = 4 * lane + (3 - b);
          sh_kk = kk - run;
        }
        run += cs[b];
      }
    }
    __syncthreads();
    prefix |= sh_bin << shift;
    maskbits |= 0xffu << shift;
    kk = sh_kk;
    __syncthreads();
  }
  const uint32_t thr = prefix;   // key of the k-th largest element
  const uint32_t need = kk;      // how many elements equal to thr are kept (lowest index first)

  uint32_t run_gt = 0, run_eq = 0;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + tid;
    const bool valid = i < n;
    const uint32_t key = valid ? order_key(sc[i]) : 0u;
    const bool gt = valid && key > thr;
    const bool eq = valid && key == thr;
    const unsigned long long bgt = __ballot(gt), beq = __ballot(eq);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
      wtot_gt[wave] = __popcll(bgt);
      wtot_eq[wave] = __popcll(beq);
    }
    __syncthreads();
    uint32_t gt_before = run_gt + __popcll(bgt & below), eq_before = run_eq + __popcll(beq & below);
    uint32_t tot_gt = 0, tot_eq = 0;
#pragma unroll
    for (int wv = 0; wv < 16; ++wv) {
      const uint32_t g = wtot_gt[wv], e = wtot_eq[wv];
      if (wv < wave) { gt_before += g; eq_before += e; }
      tot_gt += g;
      tot_eq += e;
    }
    const bool sel = gt || (eq && eq_before < need);
    if (sel && out) out[gt_before + min(eq_before, need)] = (int64_t)i;
    if (valid && mk) mk[i] = sel ? 1.0f : 0.0f;
    run_gt += tot_gt;
    run_eq += tot_eq;
    __syncthreads();
  }
}

// K5r  the same select with the keys held in registers (N <= 1024 KPT): one unconditional (clamped) load per key instead of a
//      reload in each of the five sweeps, every wave scans the 256 bins itself with a DPP prefix sum (no broadcast through LDS,
//      one barrier per pass instead of three), wave w owns the contiguous elements [w span, (w + 1) span) so the ordered
//      compaction needs one exchange of wave totals.  Integer arithmetic only: the same indices / mask as topk_select_kernel.
//      tools/trace_small.py found the per-key "load if in range", the ds_bpermute scan and the per-sweep reloads to be most of
//      the 8.5-8.9 us the generic kernel takes per call.
template <int KPT>
__global__ __launch_bounds__(1024) void topk_select_reg_kernel(const float* __restrict__ scores, SegView sv,
                                                               int64_t* __restrict__ idx, float* __restrict__ mask) {
  constexpr int NW = 16;
  const int s = blockIdx.x;
  const int n = sv.n_rows(s);
  const int k = min(sv.n_out(s), n);
  const float* sc = scores + sv.row_begin(s);
  int64_t* out = idx ? idx + sv.out_begin(s) : nullptr;
  float* mk = mask ? mask + sv.row_begin(s) : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  __shared__ uint32_t hist[4][256];
  __shared__ uint32_t wtot[NW][2];

  if (n <= 0) return;
  if (k <= 0) {                                      // nothing kept (the API rejects k < 1; ragged calls may carry an empty output)
    if (mk) for (int i = tid; i < n; i += 1024) mk[i] = 0.f;
    return;
  }
  const int kpw = (n + 1023) / 1024;                 // 64-element groups per wave (<= KPT)
  const int e0 = wave * kpw * 64 + lane;
  uint32_t key[KPT];
  {
    float raw[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) raw[j] = sc[min(e0 + 64 * j, n - 1)];
    hist[tid >> 8][tid & 255] = 0;
#pragma unroll
    for (int j = 0; j < KPT; ++j) key[j] = (j < kpw && e0 + 64 * j < n) ? order_key(raw[j]) : 0u;
  }
  __syncthreads();
  uint32_t prefix = 0, maskbits = 0, kk = (uint32_t)k;
#pragma unroll 1
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = 8 * pass;
    uint32_t* hp = hist[pass];
#pragma unroll
    for (int j = 0; j < KPT; ++j)
      if (j < kpw && e0 + 64 * j < n && (key[j] & maskbits) == prefix) atomicAdd(&hp[(key[j] >> shift) & 255u], 1u);
    __syncthreads();
    // lane L owns bins 255 - 4 L .. 252 - 4 L (top first): "keys in the bins above mine" is an exclusive prefix sum over lanes
    const u32x4 cv = *reinterpret_cast<const u32x4*>(&hp[252 - 4 * lane]);
    const uint32_t cs[4] = {cv[3], cv[2], cv[1], cv[0]};
    const uint32_t tot = cs[0] + cs[1] + cs[2] + cs[3];
    uint32_t run = wave_prefix_sum_u32(tot) - tot;
    uint32_t found = 0xffffffffu, found_kk = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (run < kk && run + cs[b] >= kk) {
        found = 255 - 4 * lane - b;
        found_kk = kk - run;
      }
      run += cs[b];
    }
    const unsigned long long who = __ballot(found != 0xffffffffu);      // exactly one lane (k >= 1)
    const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
    const uint32_t bin = (uint32_t)__builtin_amdgcn_readlane((int)found, src);
    kk = (uint32_t)__builtin_amdgcn_readlane((int)found_kk, src);
    prefix |= bin << shift;
    maskbits |= 0xffu << shift;
  }
  const uint32_t thr = prefix, need = kk;
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
  if (lane == 0) { wtot[wave][0] = my_gt; wtot[wave][1] = my_eq; }
  __syncthreads();
  uint32_t run_gt = 0, run_eq = 0;
#pragma unroll
  for (int wv = 0; wv < NW; ++wv)
    if (wv < wave) { run_gt += wtot[wv][0]; run_eq += wtot[wv][1]; }
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int e = e0 + 64 * j;
    const bool valid = j < kpw && e < n;
    const bool gt = (bgt[j] >> lane) & 1ull;
    const bool eq = (beq[j] >> lane) & 1ull;
    const uint32_t gt_before = run_gt + __popcll(bgt[j] & below), eq_before = run_eq + __popcll(beq[j] & below);
    const bool sel = gt || (eq && eq_before < need);
    if (sel && out) out[gt_before + min(eq_before, need)] = (int64_t)e;
    if (valid && mk) mk[e] = sel ? 1.0f : 0.0f;
    run_gt += __popcll(bgt[j]);
    run_eq += __popcll(beq[j]);
  }
}

// =================================================================================================
// K6  row gather out[ob + j] = h[rb + idx[ob + j]].  grid (row_chunks, n_seg), block 256, wave per row.
// =================================================================================================
// ITERS > 0: D == ITERS * 64 * kVec (2048 / 3584 / 4096 in bf16: 4 / 7 / 8) -- compile-time trip count, the whole row (ITERS x
// 16 B per lane) is loaded before the first store and the NEXT row's index is fetched under the current row's copy.
// ITERS == 0: any D (run-time trip count).
template <typename T, bool NT, int ITERS = 0>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ h, SegView sv, int d,
                                                          const int64_t* __restrict__ idx, T* __restrict__ out,
                                                          int rows_per_block, const int64_t* __restrict__ src_map) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.y;
  const int ko = sv.n_out(s);
  const int jb = blockIdx.x * rows_per_block;
  if (jb >= ko) return;
  const int je = min(ko, jb + rows_per_block);
  const int64_t ob = sv.out_begin(s), rb = sv.row_begin(s);
  if constexpr (ITERS > 0) {
    int j = jb + wave;
    if (j >= je) return;
    int64_t lsrc = rb + idx[ob + j];
    int64_t src = src_map ? src_map[lsrc] : lsrc;
    // two rows in flight per wave: row j + 4 is loaded before row j is stored, so a wave never sits idle between the last store of
    // one row and the first data of the next (the copy is latency-bound per wave: 7 KB per round trip)
    auto load_row = [&](u32x4 (&x)[ITERS], int64_t srow) {
      const u32x4* sp = reinterpret_cast<const u32x4*>(h + srow * d) + lane;
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        if constexpr (NT) x[i] = __builtin_nontemporal_load(sp + 64 * i);     // read once: do not displace the next sweep's lines
        else x[i] = sp[64 * i];
      }
    };
    auto store_row = [&](const u32x4 (&x)[ITERS], int jrow) {
      u32x4* dp = reinterpret_cast<u32x4*>(out + (ob + jrow) * d) + lane;
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        if constexpr (NT) __builtin_nontemporal_store(x[i], dp + 64 * i);      // keeps the write-back from slowing the next sweep 1
        else dp[64 * i] = x[i];
      }
    };
    auto next_src = [&](int jrow) -> int64_t {
      const int64_t l = rb + idx[ob + jrow];
      return src_map ? src_map[l] : l;
    };
    u32x4 xa[ITERS], xb[ITERS];
    load_row(xa, src);
    for (; j < je; j += 8) {
      const bool has_b = j + 4 < je;
      if (has_b) load_row(xb, next_src(j + 4));
      store_row(xa, j);
      if (!has_b) break;
      if (j + 8 < je) load_row(xa, next_src(j + 8));
      store_row(xb, j + 4);
    }
  } else {
    for (int j = jb + wave; j < je; j += 4) {
      const int64_t lsrc = rb + idx[ob + j];
      const int64_t src = src_map ? src_map[lsrc] : lsrc;
      const u32x4* sp = reinterpret_cast<const u32x4*>(h + src * d);
      u32x4* dp = reinterpret_cast<u32x4*>(out + (ob + j) * d);
      for (int v = lane; v < d / V; v += 64) {
        if constexpr (NT) __builtin_nontemporal_store(sp[v], dp + v);
        else dp[v] = sp[v];
      }
    }
  }
}

// =================================================================================================
// host side
// =================================================================================================
// Cache policy of the three kernels that touch the token tensor: non-temporal once the tensor clearly exceeds the Infinity
// Cache (256 MB) + L2 (32 MB).  Measured crossover between 264 MB (default policy 4 % faster) and 528 MB (nt 12 % faster).
constexpr int64_t kStreamBytes = 384ll << 20;
// The policy is a property of the WHOLE API call: when a call is cut into halves (lis.hip) the halves inherit the call's row count.
inline thread_local int64_t tl_policy_rows = -1;
struct PolicyScope {
  explicit PolicyScope(int64_t rows) { tl_policy_rows = rows; }
  ~PolicyScope() { tl_policy_rows = -1; }
};
inline bool stream_policy(int64_t total_rows, int64_t d, size_t elem) {
  const int64_t rows = tl_policy_rows >= 0 ? tl_policy_rows : total_rows;
  return rows * d * (int64_t)elem >= kStreamBytes;
}

struct LisPlan {
  int64_t S, maxn, d, hd;
  int row_splits;
  int ks1, kslice1;  // kbar projection  [S, D] x [Hd, D]^T
  int ks2, kslice2;  // w projection     [S, Hd] x [Hd, D]
  int n_cpart;       // blocks of 256 over Hd
  bool mfma_bf16;    // shapes allow the bf16x3 MFMA projections (K % 16 == 0); weights must also be bf16
  size_t off_partial, off_xbar, off_part1, off_kbar, off_c, off_part2, off_w, off_xs, off_ksp, off_cpart, total;
};

inline LisPlan make_plan(int64_t S, int64_t maxn, int64_t d, int64_t hd) {
  LisPlan p{};
  p.S = S; p.maxn = maxn; p.d = d; p.hd = hd;
  // Batch-invariant plan: the summation tree of one segment never depends on how many segments share the call
  // (fixed 128-row chunks in sweep 1, fixed split-K slices in the projections), so batching, the two-half pipeline
  // and the ragged form all reproduce a single-image call bit for bit.
  p.row_splits = (int)std::max<int64_t>(1, cdiv(maxn, kRowsPerChunk));
  p.mfma_bf16 = (d % 16 == 0) && (hd % 16 == 0);
  p.kslice1 = (int)std::min<int64_t>(kSliceNT, cdiv(d, 16) * 16);
  p.ks1 = (int)cdiv(d, p.kslice1);
  p.kslice2 = (int)std::min<int64_t>(kSliceNN, cdiv(hd, 16) * 16);
  p.ks2 = (int)cdiv(hd, p.kslice2);
  p.n_cpart = (int)cdiv(hd, 8);
  const int64_t m_pad = 32 * cdiv(S, 32);
  size_t o = 0;
  auto take = [&](size_t nfloat) { size_t r = o; o += align_up(nfloat * sizeof(float), 256); return r; };
  p.off_partial = take((size_t)S * p.row_splits * d);
  p.off_xbar = take((size_t)S * d);
  p.off_part1 = take((size_t)p.ks1 * m_pad * hd);
  p.off_kbar = take((size_t)S * hd);
  p.off_c = take((size_t)S);
  p.off_part2 = take((size_t)p.ks2 * m_pad * d);
  p.off_w = take((size_t)S * d);
  p.off_xs = take(((size_t)3 * m_pad * d + 1) / 2);
  p.off_ksp = take(((size_t)3 * m_pad * hd + 1) / 2);
  p.off_cpart = take((size_t)S * p.n_cpart);
  p.total = o;
  return p;
}

inline int check_segments_impl(const vsel_segments* seg, bool need_k) {
  if (!seg) return fail(VSEL_ERR_INVALID, "segments is NULL");
  if (seg->n_seg < 1 || seg->rows_per_seg < 1 || seg->total_rows < 1)
    return fail(VSEL_ERR_INVALID, "empty segments (n_seg=%lld rows_per_seg=%lld total_rows=%lld)",
                (long long)seg->n_seg, (long long)seg->rows_per_seg, (long long)seg->total_rows);
  if (seg->n_seg > 65535) return fail(VSEL_ERR_UNSUPPORTED, "n_seg %lld > 65535", (long long)seg->n_seg);
  if (seg->total_rows >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "total_rows must fit int32");
  if (!seg->seg_rows && seg->total_rows != seg->n_seg * seg->rows_per_seg)
    return fail(VSEL_ERR_INVALID, "uniform segments: total_rows != n_seg * rows_per_seg");
  if (need_k) {
    if (seg->k < 1 || seg->k > seg->rows_per_seg)
      return fail(VSEL_ERR_INVALID, "k=%lld must satisfy 1 <= k <= rows_per_seg=%lld", (long long)seg->k,
                  (long long)seg->rows_per_seg);
    if ((seg->seg_rows != nullptr) != (seg->seg_out != nullptr))
      return fail(VSEL_ERR_INVALID, "seg_rows and seg_out must both be given (ragged) or both NULL (uniform)");
    if (!seg->seg_out && seg->total_out != seg->n_seg * seg->k)
      return fail(VSEL_ERR_INVALID, "uniform segments: total_out != n_seg * k");
  }
  return VSEL_OK;
}

inline int check_scorer(const vsel_scorer* sc, vsel_dtype hdtype) {
  if (!sc || !sc->wq || !sc->bq || !sc->wk || !sc->bk) return fail(VSEL_ERR_INVALID, "scorer parameter pointer is NULL");
  if (sc->d < 8 || sc->hd < 1) return fail(VSEL_ERR_INVALID, "bad scorer dims d=%lld hd=%lld", (long long)sc->d, (long long)sc->hd);
  if (sc->d % 8 != 0) return fail(VSEL_ERR_UNSUPPORTED, "in_features D=%lld must be a multiple of 8 (16-byte rows)", (long long)sc->d);
  if (sc->wdtype != VSEL_BF16 && sc->wdtype != VSEL_F32) return fail(VSEL_ERR_INVALID, "bad weight dtype");
  if (hdtype != VSEL_BF16 && hdtype != VSEL_F32) return fail(VSEL_ERR_INVALID, "bad token dtype");
  return VSEL_OK;
}

template <typename T>
inline int launch_score(hipStream_t st, const T* h, const SegView& sv, const vsel_segments* seg, int d,
                        const float* w, const float* c, int hd, float* scores, const int64_t* out_map = nullptr) {
  constexpr int V = Elem<T>::kVec;
  // rows per block: keep >= ~2048 blocks when there is enough work, 8..64 rows per block
  int64_t rpb = 64;
  while (rpb > 8 && seg->n_seg * cdiv(seg->rows_per_seg, rpb) < 2048) rpb >>= 1;
  dim3 grid((unsigned)cdiv(seg->rows_per_seg, rpb), (unsigned)seg->n_seg);
  const float sq = (float)sqrt((double)hd);
  const bool nt = stream_policy(seg->total_rows, d, sizeof(T));
  const int iters = (d % (64 * V) == 0) ? d / (64 * V) : 0;
#define VSEL_SCORE_CASE(I)                                                                              \
  case I:                                                                                               \
    if (nt) VSEL_LAUNCH((score_kernel<T, I, true>), grid, dim3(256), 0, st, h, sv, d, w, c, sq, (int)rpb, scores, out_map);  \
    else VSEL_LAUNCH((score_kernel<T, I, false>), grid, dim3(256), 0, st, h, sv, d, w, c, sq, (int)rpb, scores, out_map);    \
    break;
  switch (iters) {
    VSEL_SCORE_CASE(1) VSEL_SCORE_CASE(2) VSEL_SCORE_CASE(3) VSEL_SCORE_CASE(4)
    VSEL_SCORE_CASE(5) VSEL_SCORE_CASE(6) VSEL_SCORE_CASE(7) VSEL_SCORE_CASE(8)
    default: VSEL_LAUNCH((score_kernel<T, 0, false>), grid, dim3(256), 0, st, h, sv, d, w, c, sq, (int)rpb, scores, out_map);
  }
#undef VSEL_SCORE_CASE
  VSEL_AFTER_LAUNCH(st, "score_kernel");
  return VSEL_OK;
}

// sweep 1 as colsum_seg_kernel (the segments' sums, one "chunk") instead of per-chunk partials: when some slab width (512 / 256 /
// 128 bf16 columns) gives knob LIS_SEG_SUMS (segment, slab) pairs = waves (default 896 = 3.5 per CU; 0 = never; measured per call at
// the 7B geometry, chunked vs this: B = 32 231 vs 228 us, 64 446 vs 428, 96 721 vs 685, 128 848-895 vs 826-830; with 672 pairs no
// gain, with one 512-column slab per wave at B = 64 a loss: 483 vs 442), never for the <= 8 segments the small-batch form may take
// (its kernels rebuild x-bar from chunk partials), more than one chunk, and no segment longer than 1.375 x the average: a wave
// streams its (segment, slab) alone, at most twice as fast once the others are done, so the longest segment's waves finish last --
// tools/exp_ragged_lis.py, profiles/r04_ragged_lis.txt, call time chunked vs this form by max / mean: 1.0 +1.5 %, 1.13 +1.8 ... +3.2 %,
// 1.33 +1.2 ... 0 %, 1.5 - 1.7 (N_i ~ U{576..4096}) -4 % (32 prompts), -7 % (64), -10 % (128: sweep 1 446 -> 336 us).  Same sums bit for
// bit either way: batch invariance is untouched.
inline bool seg_sums_form(const LisPlan& p, const vsel_segments* seg) {
  const int mn = knob(VSEL_KNOB_LIS_SEG_SUMS);
  // (pairs counted for the narrowest slab of bf16 tokens, 128 columns, whatever the token type: the projection stage does not know it)
  return mn > 0 && p.S > 8 && p.S * cdiv(p.d, 128) >= mn && p.row_splits > 1 && p.d % 8 == 0 && seg->total_rows * 11 >= (int64_t)p.S * p.maxn * 8;
}

template <typename T>
inline int launch_colsum(hipStream_t st, const T* h, const SegView& sv, int d, int S, int row_splits, float* partial,
                         int64_t total_rows, bool seg_sums = false, uint16_t* xs = nullptr) {
  constexpr int V = Elem<T>::kVec;
  if (seg_sums) {
    // the widest slab (16- / 8- / 4-byte loads per lane) that still gives knob LIS_SEG_SUMS (segment, slab) pairs
    const int mn = std::max(1, knob(VSEL_KNOB_LIS_SEG_SUMS));
    const int epw = sizeof(T) == 2 ? 2 : 1;                       // elements per 32-bit word
    int nw = 4;
    while (nw > 1 && ((int64_t)S * cdiv(d, 64 * nw * epw) < mn || d % (nw * epw) != 0)) nw >>= 1;
    const int col_tiles = (int)cdiv(d, 64 * nw * epw);
    const dim3 g2((unsigned)cdiv((int64_t)col_tiles * S, 4));
    const bool nt = stream_policy(total_rows, d, sizeof(T));
#define VSEL_SEG_LAUNCH(NWV)                                                                                                    \
    do {                                                                                                                        \
      if (nt) VSEL_LAUNCH((colsum_seg_kernel<T, true, NWV>), g2, dim3(256), 0, st, h, sv, d, col_tiles, S, partial, xs); \
      else VSEL_LAUNCH((colsum_seg_kernel<T, false, NWV>), g2, dim3(256), 0, st, h, sv, d, col_tiles, S, partial, xs);   \
    } while (0)
    if (nw == 4) VSEL_SEG_LAUNCH(4); else if (nw == 2) VSEL_SEG_LAUNCH(2); else VSEL_SEG_LAUNCH(1);
#undef VSEL_SEG_LAUNCH
    VSEL_AFTER_LAUNCH(st, "colsum_seg_kernel");
    return VSEL_OK;
  }
  const dim3 grid((unsigned)cdiv(d, 64 * V), row_splits, S);
  if ((int64_t)grid.x * grid.y * grid.z <= 1024)       // <= one workgroup per SIMD-quad: latency-bound, use the deep form
    VSEL_LAUNCH((colsum_partial_kernel<T, false, true>), grid, dim3(256), 0, st, h, sv, d, row_splits, partial);
  else if (stream_policy(total_rows, d, sizeof(T)))
    VSEL_LAUNCH((colsum_partial_kernel<T, true>), grid, dim3(256), 0, st, h, sv, d, row_splits, partial);
  else
    VSEL_LAUNCH((colsum_partial_kernel<T, false>), grid, dim3(256), 0, st, h, sv, d, row_splits, partial);
  VSEL_AFTER_LAUNCH(st, "colsum_partial_kernel");
  return VSEL_OK;
}

// stage 1: sweep 1 (column-sum partials)
template <typename T>
inline int run_colsum(hipStream_t st, const T* h, const vsel_segments* seg, int d, char* ws, const LisPlan& p) {
  return launch_colsum<T>(st, h, make_view(seg), d, (int)seg->n_seg, p.row_splits, (float*)(ws + p.off_partial), seg->total_rows,
                          seg_sums_form(p, seg), p.mfma_bf16 ? (uint16_t*)(ws + p.off_xs) : nullptr);
}

// stage 2: partials -> xbar -> kbar -> (w, c)   (small, latency-bound kernels)
// col_sums (optional): [S, D] fp32 column sums of the tokens supplied by the producer (vsel_lis_select_presummed); they take
// the place of the sweep-1 partials as a single "chunk", everything downstream is unchanged.
template <typename TW>
inline int run_proj(hipStream_t st, const vsel_segments* seg, const vsel_scorer* sc, char* ws, const LisPlan& p_in,
                    const float* col_sums = nullptr) {
  const SegView sv = make_view(seg);
  const int d = (int)sc->d, hd = (int)sc->hd, S = (int)seg->n_seg;
  LisPlan p = p_in;
  if (col_sums || seg_sums_form(p_in, seg)) p.row_splits = 1;     // the producer's column sums / colsum_seg_kernel's: one "chunk"
  const float* partial = col_sums ? col_sums : (const float*)(ws + p.off_partial);
  float* xbar = (float*)(ws + p.off_xbar);
  float* part1 = (float*)(ws + p.off_part1);
  float* kbar = (float*)(ws + p.off_kbar);
  float* c = (float*)(ws + p.off_c);
  float* part2 = (float*)(ws + p.off_part2);
  float* w = (float*)(ws + p.off_w);
  if constexpr (std::is_same<TW, bf16_t>::value) {
    if (p.mfma_bf16) {
      // bf16x3 MFMA projections (proj_bf16x3.h)
      uint16_t* xs = (uint16_t*)(ws + p.off_xs);
      uint16_t* ksp = (uint16_t*)(ws + p.off_ksp);
      float* cpart = (float*)(ws + p.off_cpart);
      if (col_sums || !seg_sums_form(p_in, seg)) {       // (colsum_seg_kernel leaves the planes itself)
        VSEL_LAUNCH(colsum_finish_split_kernel, dim3((unsigned)cdiv(d, 256), S), dim3(256), 0, st, partial, sv, d,
                           p.row_splits, S, xs);
        VSEL_AFTER_LAUNCH(st, "colsum_finish_split_kernel");
      }
      const unsigned mtiles = (unsigned)cdiv(S, 32);
      const int m_pad = 32 * (int)mtiles;
      VSEL_LAUNCH(gemm_nt_bf16x3_kernel<kProjWaves>, dim3((unsigned)cdiv(hd, 64), (unsigned)cdiv(mtiles, kProjWaves), p.ks1), dim3(64 * kProjWaves), 0, st, xs,
                         (const uint16_t*)sc->wk, S, hd, d, p.kslice1, part1, (int)mtiles);
      VSEL_AFTER_LAUNCH(st, "gemm_nt_bf16x3_kernel");
      VSEL_LAUNCH(kbar_finish_split_kernel, dim3(mtiles, p.n_cpart), dim3(256), 0, st, part1, p.ks1, S, hd, m_pad,
                         (const uint16_t*)sc->bk, (const uint16_t*)sc->bq, kbar, ksp, cpart);
      VSEL_AFTER_LAUNCH(st, "kbar_finish_split_kernel");
      VSEL_LAUNCH(gemm_nn_bf16x3_kernel<kProjWaves>, dim3((unsigned)cdiv(d, 256), (unsigned)cdiv(mtiles, kProjWaves), p.ks2), dim3(64 * kProjWaves), 0, st, ksp,
                         (const uint16_t*)sc->wq, S, d, hd, p.kslice2, part2, (int)mtiles);
      VSEL_AFTER_LAUNCH(st, "gemm_nn_bf16x3_kernel");
      VSEL_LAUNCH(w_finish_kernel, dim3(mtiles, (unsigned)cdiv(d, 8)), dim3(256), 0, st, part2, p.ks2, S, d, m_pad,
                         cpart, p.n_cpart, w, c);
      VSEL_AFTER_LAUNCH(st, "w_finish_kernel");
      return VSEL_OK;
    }
  }
  // generic path: fp32-input MFMA (fp32 weights, or K not a multiple of 16)
  VSEL_LAUNCH(colsum_finish_kernel, dim3((unsigned)cdiv(d, 256), S), dim3(256), 0, st, partial, sv, d,
                     p.row_splits, xbar);
  VSEL_AFTER_LAUNCH(st, "colsum_finish_kernel");
  VSEL_LAUNCH((gemm_nt_kernel<TW>), dim3((unsigned)cdiv(hd, 32), (unsigned)cdiv(S, 32), p.ks1), dim3(64), 0, st,
                     xbar, (const TW*)sc->wk, S, hd, d, p.kslice1, part1);
  VSEL_AFTER_LAUNCH(st, "gemm_nt_kernel");
  VSEL_LAUNCH((kbar_finish_kernel<TW>), dim3(S), dim3(256), 0, st, part1, p.ks1, S, hd, (const TW*)sc->bk,
                     (const TW*)sc->bq, kbar, c);
  VSEL_AFTER_LAUNCH(st, "kbar_finish_kernel");
  VSEL_LAUNCH((gemm_nn_kernel<TW>), dim3((unsigned)cdiv(d, 256), (unsigned)cdiv(S, 32), p.ks2), dim3(64), 0, st,
                     kbar, (const TW*)sc->wq, S, d, hd, p.kslice2, part2);
  VSEL_AFTER_LAUNCH(st, "gemm_nn_kernel");
  VSEL_LAUNCH(slice_sum_kernel, dim3((unsigned)cdiv((int64_t)S * d, 256)), dim3(256), 0, st, part2, p.ks2,
                     (int64_t)S * d, w);
  VSEL_AFTER_LAUNCH(st, "slice_sum_kernel");
  return VSEL_OK;
}

// stage 3: sweep 2 (scores)
template <typename T>
inline int run_score(hipStream_t st, const T* h, const vsel_segments* seg, const vsel_scorer* sc, char* ws, const LisPlan& p,
                     float* scores, const int64_t* out_map = nullptr) {
  return launch_score<T>(st, h, make_view(seg), seg, (int)sc->d, (const float*)(ws + p.off_w), (const float*)(ws + p.off_c),
                         (int)sc->hd, scores, out_map);
}

template <typename T, typename TW>
inline int run_scores(hipStream_t st, const T* h, const vsel_segments* seg, const vsel_scorer* sc, char* ws,
                      const LisPlan& p, float* scores) {
  int rc = run_colsum<T>(st, h, seg, (int)sc->d, ws, p);
  if (rc) return rc;
  rc = run_proj<TW>(st, seg, sc, ws, p);
  if (rc) return rc;
  return run_score<T>(st, h, seg, sc, ws, p, scores);
}

template <typename T>
inline int run_scores_w(hipStream_t st, const T* h, const vsel_segments* seg, const vsel_scorer* sc, char* ws,
                        const LisPlan& p, float* scores) {
  if (sc->wdtype == VSEL_BF16) return run_scores<T, bf16_t>(st, h, seg, sc, ws, p, scores);
  return run_scores<T, float>(st, h, seg, sc, ws, p, scores);
}

inline int launch_select(hipStream_t st, const float* scores, const vsel_segments* seg, int64_t* idx, float* mask) {
  const int64_t maxn = seg->rows_per_seg;            // (= the longest segment for ragged calls)
  if (maxn <= 4 * 1024)
    VSEL_LAUNCH(topk_select_reg_kernel<4>, dim3((unsigned)seg->n_seg), dim3(1024), 0, st, scores, make_view(seg), idx, mask);
  else if (maxn <= 16 * 1024)
    VSEL_LAUNCH(topk_select_reg_kernel<16>, dim3((unsigned)seg->n_seg), dim3(1024), 0, st, scores, make_view(seg), idx, mask);
  else
    VSEL_LAUNCH(topk_select_kernel, dim3((unsigned)seg->n_seg), dim3(1024), 0, st, scores, make_view(seg), idx, mask);
  VSEL_AFTER_LAUNCH(st, "topk_select_kernel");
  return VSEL_OK;
}

// The gather for UNIFORM segments as a flat list of kept rows dealt to a fixed set of waves (knob LIS_GATHER): wave w of W copies
// kept rows w, w + W, w + 2 W ... of the call (row g = segment g / k, rank g % k), DEPTH rows of ITERS x 16 B per lane in flight per wave.
// The workgroups stay resident for the whole launch (no per-8-rows workgroup churn, no third partial round of workgroups) and a wave's
// next row is on its way while the current one is stored.  The wave index is made uniform, so the index fetch is a scalar load and the
// row pointers live in SGPRs.  Same bytes to the same places as gather_rows_kernel.
template <typename T, bool NT, int ITERS, int DEPTH>
__global__ __launch_bounds__(256) void gather_rows_flat_kernel(const T* __restrict__ h, int rows_per_seg, int k, int d,
                                                               const int64_t* __restrict__ idx, T* __restrict__ out, int total,
                                                               int w_div_k, int w_mod_k, const int64_t* __restrict__ src_map) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = gridDim.x * 4;
  int g = blockIdx.x * 4 + wave;                     // kept row of the call (= output row)
  if (g >= total) return;
  int sg = g / k, jr = g - sg * k;                   // its segment and rank, advanced by W per step without dividing again
  auto advance = [&]() {
    g += W;
    sg += w_div_k;
    jr += w_mod_k;
    if (jr >= k) { jr -= k; ++sg; }
  };
  auto load_row = [&](u32x4 (&x)[ITERS]) {
    const int64_t l = (int64_t)sg * rows_per_seg + idx[g];
    const int64_t srow = src_map ? src_map[l] : l;
    const u32x4* sp = reinterpret_cast<const u32x4*>(h + srow * d) + lane;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      if constexpr (NT) x[i] = __builtin_nontemporal_load(sp + 64 * i);
      else x[i] = sp[64 * i];
    }
  };
  auto store_row = [&](const u32x4 (&x)[ITERS], int grow) {
    u32x4* dp = reinterpret_cast<u32x4*>(out + (int64_t)grow * d) + lane;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      if constexpr (NT) __builtin_nontemporal_store(x[i], dp + 64 * i);
      else dp[64 * i] = x[i];
    }
  };
  u32x4 x[DEPTH][ITERS];
  int gs = g;                                        // the row the next store writes
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    if (g < total) load_row(x[u]);
    advance();
  }
  auto step = [&](u32x4 (&xu)[ITERS]) {
    store_row(xu, gs);
    gs += W;
    if (g < total) load_row(xu);
    advance();
  };
  for (;;) {
    if (gs >= total) break;
    step(x[0]);
    if (gs >= total) break;
    step(x[1]);
    if constexpr (DEPTH > 2) {
      if (gs >= total) break;
      step(x[2]);
    }
    if constexpr (DEPTH > 3) {
      if (gs >= total) break;
      step(x[3]);
    }
  }
}

template <typename T>
inline int launch_gather(hipStream_t st, const T* h, int d, const vsel_segments* seg, const int64_t* idx, T* out,
                         const int64_t* src_map = nullptr) {
  int64_t rpb = 32;
  while (rpb > 4 && seg->n_seg * cdiv(seg->k, rpb) < 2048) rpb >>= 1;
  const dim3 grid((unsigned)cdiv(seg->k, rpb), (unsigned)seg->n_seg);
  const bool nt = stream_policy(seg->total_rows, d, sizeof(T));
  constexpr int V = Elem<T>::kVec;
  const int iters = ( d % (64 * V) == 0 && d / (64 * V) <= 8) ? d / (64 * V) : 0;
  // flat resident form (uniform segments, whole-chunk rows): knob LIS_GATHER = 10 * workgroups-per-CU + rows in flight per wave
  const int gform = knob(VSEL_KNOB_LIS_GATHER);
  const int64_t total_out = seg->n_seg * seg->k;
  if (gform > 0 && iters > 0 && !seg->seg_rows && !seg->seg_out && total_out < (1ll << 30) && total_out >= 4096) {
    const int depth = gform % 10, per_cu = std::max(1, gform / 10);
    const int64_t G = std::min<int64_t>(cdiv(total_out, 4), 256 * per_cu);
    const int W = (int)G * 4, kk = (int)seg->k;
#define VSEL_FLAT_CASE2(I, DP)                                                                                                         \
    do {                                                                                                                               \
      if (nt) VSEL_LAUNCH((gather_rows_flat_kernel<T, true, I, DP>), dim3((unsigned)G), dim3(256), 0, st, h, (int)seg->rows_per_seg, kk, d, \
                          idx, out, (int)total_out, W / kk, W % kk, src_map);                                                         \
      else VSEL_LAUNCH((gather_rows_flat_kernel<T, false, I, DP>), dim3((unsigned)G), dim3(256), 0, st, h, (int)seg->rows_per_seg, kk, d,   \
                       idx, out, (int)total_out, W / kk, W % kk, src_map);                                                            \
    } while (0)
#define VSEL_FLAT_CASE(I)                                                                                                              \
    case I:                                                                                                                            \
      if (depth <= 2) VSEL_FLAT_CASE2(I, 2); else if (depth == 3) VSEL_FLAT_CASE2(I, 3); else VSEL_FLAT_CASE2(I, 4);                   \
      break;
    switch (iters) {
      VSEL_FLAT_CASE(4) VSEL_FLAT_CASE(7) VSEL_FLAT_CASE(8)
      default: goto per_segment_form;
    }
#undef VSEL_FLAT_CASE
#undef VSEL_FLAT_CASE2
    VSEL_AFTER_LAUNCH(st, "gather_rows_kernel");
    return VSEL_OK;
  }
per_segment_form:
#define VSEL_GATHER_CASE(I)                                                                                                   \
  case I:                                                                                                                     \
    if (nt) VSEL_LAUNCH((gather_rows_kernel<T, true, I>), grid, dim3(256), 0, st, h, make_view(seg), d, idx, out, (int)rpb, src_map); \
    else VSEL_LAUNCH((gather_rows_kernel<T, false, I>), grid, dim3(256), 0, st, h, make_view(seg), d, idx, out, (int)rpb, src_map);   \
    break;
  switch (iters) {
    VSEL_GATHER_CASE(1) VSEL_GATHER_CASE(2) VSEL_GATHER_CASE(3) VSEL_GATHER_CASE(4)
    VSEL_GATHER_CASE(5) VSEL_GATHER_CASE(6) VSEL_GATHER_CASE(7) VSEL_GATHER_CASE(8)
    default:
      if (nt) VSEL_LAUNCH((gather_rows_kernel<T, true, 0>), grid, dim3(256), 0, st, h, make_view(seg), d, idx, out, (int)rpb, src_map);
      else VSEL_LAUNCH((gather_rows_kernel<T, false, 0>), grid, dim3(256), 0, st, h, make_view(seg), d, idx, out, (int)rpb, src_map);
  }
#undef VSEL_GATHER_CASE
  VSEL_AFTER_LAUNCH(st, "gather_rows_kernel");
  return VSEL_OK;
}

}  // namespace vsel
