// Training LIS block (reference: qwen-vl-finetune/compression_method/selector_model.py:158-173 forward,
// :308-311 constraint loss, :60-70 TopK.backward; llava-ov-15/compression_method/selector_model.py:127-142).
//
// forward : scores (lis_kernels.h) -> soft top-k (the root of _find_ts by bracketed Newton steps, softtopk.h) -> h_new = ps * h -> hard mask y -> BCE
// backward: closed form of autograd through the scorer (SURVEY.md section 7 hard part 4).  With g = dL/dscores,
//           rs = 1/sqrt(Hd), xbar = mean x, kbar = Wk xbar + bk:
//             dWq = (kbar rs) (x) sum_i g_i x_i        dbq = kbar rs sum_i g_i
//             dk  = (Wq sum_i g_i x_i + bq sum_i g_i) rs / N
//             dWk = dk (x) sum_i x_i                   dbk = N dk
//             dx_i = ps_i dh'_i + g_i rs Wq^T kbar + Wk^T dk
//           i.e. two more sweeps over the token tensor (row dots, weighted column sums) + GEMVs + rank-1 writes.
#include "lis_kernels.h"
#include "lis_small.h"
#include "softtopk.h"

namespace vsel {

int launch_soft_topk_fwd(hipStream_t st, const float* xs, int64_t b, int64_t n, int64_t k, float* ps, float* ts);
int launch_soft_topk_bwd(hipStream_t st, const float* g, const float* xs, const float* ts, int64_t b, int64_t n, float* gx);

// h_new[i, :] = (ps[i] * h[i, :]).type(dtype)     (:164-166)   wave per row
template <typename T>
__global__ __launch_bounds__(256) void mask_apply_kernel(const T* __restrict__ h, const float* __restrict__ ps, int n,
                                                         int d, T* __restrict__ out) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
    const float p = ps[r];
    const T* src = h + (int64_t)r * d;
    T* dst = out + (int64_t)r * d;
    // the row's column groups are loaded 8 at a time (clamped, unconditional): a plain loop waits for every load in turn
    for (int c0 = lane * V; c0 < d; c0 += 8 * 64 * V) {
      float v[8][V];
#pragma unroll
      for (int u = 0; u < 8; ++u) load_vec(src + min(c0 + u * 64 * V, d - V), v[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u * 64 * V;
        if (c < d) {
#pragma unroll
          for (int q = 0; q < V; ++q) v[u][q] *= p;
          store_vec(dst + c, v[u]);
        }
      }
    }
  }
}

// bce = mean_i -(y_i max(log p_i, -100) + (1 - y_i) max(log(1 - p_i), -100))     (:310, ATen clamp)
__global__ __launch_bounds__(1024) void bce_kernel(const float* __restrict__ ps, const float* __restrict__ y, int n,
                                                   float* __restrict__ out) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float p = ps[i], t = y[i];
    acc += (t - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - t * fmaxf(logf(p), -100.0f);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = t / (float)n;
  }
}

// The forward's tail for one row of N <= 256 KPT scores in ONE launch (three before: soft top-k, hard top-k mask, BCE):
//   ts, ps = _find_ts(scores, k)                 (find_ts_newton, as soft_topk_fwd_kernel)
//   y      = hard top-k mask of the scores       (topk_select_reg_kernel's integer select: larger key first, then lower index)
//   bce    = mean_i BCE(ps_i, y_i)               (ATen's -100 clamp; fixed summation order: wave, then waves 0..3)
// Wave w owns the contiguous elements [w span, (w + 1) span), element j of a lane = e0 + 64 j.
template <int KPT>
__global__ __launch_bounds__(256) void train_tail_kernel(const float* __restrict__ xs, int n, int k, float* __restrict__ ps,
                                                         float* __restrict__ ts, float* __restrict__ y,
                                                         float* __restrict__ bce) {
  constexpr int NT = 256, NW = 4;
  __shared__ float red[6][NW];
  __shared__ uint32_t hist[4][256];
  __shared__ uint32_t wtot[NW][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kpw = (n + NT - 1) / NT;                  // 64-element groups per wave (<= KPT)
  const int e0 = wave * kpw * 64 + lane;
  float xr[KPT];
  float mx = -INFINITY, mn = INFINITY, sx = 0.f;
#pragma unroll
  for (int j = 0; j < KPT; ++j) xr[j] = xs[min(e0 + 64 * j, n - 1)];      // unconditional (clamped) loads
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const bool ok = j < kpw && e0 + 64 * j < n;
    if (ok) { mx = fmaxf(mx, xr[j]); mn = fminf(mn, xr[j]); sx += xr[j]; }
    else xr[j] = -INFINITY;                           // sigmoid(-inf + t) = 0: padding never contributes
  }
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) hist[p4][tid] = 0;
  mx = wave_max(mx);
  mn = wave_min(mn);
  if (lane == 0) { red[4][wave] = mx; red[5][wave] = mn; }
  __syncthreads();
  mx = red[4][0]; mn = red[5][0];
#pragma unroll
  for (int w = 1; w < NW; ++w) { mx = fmaxf(mx, red[4][w]); mn = fminf(mn, red[5][w]); }
  // ---- _find_ts (selector_model.py:72-86): the same root by bracketed Newton steps (softtopk.h, find_ts_newton) ----------------
  __syncthreads();                                  // red[4..5] are reused by the iteration
  const float t = find_ts_newton<NW, NW, KPT>(xr, sx, mx, mn, n, k, red);
  if (tid == 0) ts[0] = t;
  // ---- hard top-k mask: 4-pass radix select on the ordered keys, then the ordered tie rule ------------------------------
  uint32_t key[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) key[j] = (j < kpw && e0 + 64 * j < n) ? order_key(xr[j]) : 0u;
  uint32_t prefix = 0, maskbits = 0, kk = (uint32_t)k;
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
    for (int b = 0; b < 4; ++b) {
      if (run < kk && run + cs[b] >= kk) {
        found = 255 - 4 * lane - b;
        found_kk = kk - run;
      }
      run += cs[b];
    }
    const unsigned long long who = __ballot(found != 0xffffffffu);
    const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
    const uint32_t bin = (uint32_t)__builtin_amdgcn_readlane((int)found, src);
    kk = (uint32_t)__builtin_amdgcn_readlane((int)found_kk, src);
    prefix |= bin << shift;
    maskbits |= 0xffu << shift;
  }
  const uint32_t thr = prefix, need = kk;
  uint32_t my_eq = 0;
  unsigned long long beq[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    beq[j] = __ballot(j < kpw && e0 + 64 * j < n && key[j] == thr);
    my_eq += __popcll(beq[j]);
  }
  if (lane == 0) wtot[wave][1] = my_eq;
  __syncthreads();
  uint32_t run_eq = 0;
#pragma unroll
  for (int wv = 0; wv < NW; ++wv)
    if (wv < wave) run_eq += wtot[wv][1];
  const unsigned long long below = (1ull << lane) - 1ull;
  // ---- ps, y, BCE ---------------------------------------------------------------------------------------------------
  float bacc = 0.f;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int e = e0 + 64 * j;
    const bool valid = j < kpw && e < n;
    const bool eq = (beq[j] >> lane) & 1ull;
    const uint32_t eq_before = run_eq + __popcll(beq[j] & below);
    const bool sel = valid && (key[j] > thr || (eq && eq_before < need));
    if (valid) {
      const float p = sigmoidf_ref(xr[j] + t);          // :86
      const float yv = sel ? 1.0f : 0.0f;
      ps[e] = p;
      y[e] = yv;
      bacc += (yv - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - yv * fmaxf(logf(p), -100.0f);
    }
    run_eq += __popcll(beq[j]);
  }
  const float btot = block_sum<NW>(bacc, red, 0);
  if (tid == 0) bce[0] = btot / (float)n;
}

// dps[i] = <d_hnew[i,:], h[i,:]> + d_ps_ext[i] + dl_dbce * (p - y) / max((1 - p) p, 1e-12) / N     wave per row
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ dhn, const T* __restrict__ h,
                                                     const float* __restrict__ ps, const float* __restrict__ y,
                                                     const float* __restrict__ d_ps_ext, float dl_dbce, int n, int d,
                                                     float* __restrict__ dps) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
    const T* a = dhn + (int64_t)r * d;
    const T* b = h + (int64_t)r * d;
    float acc = 0.f;
    for (int c0 = lane * V; c0 < d; c0 += 4 * 64 * V) {      // 2 x 4 loads in flight (clamped); same order of the FMAs
      float va[4][V], vb[4][V];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = min(c0 + u * 64 * V, d - V);
        load_vec(a + cc, va[u]);
        load_vec(b + cc, vb[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + u * 64 * V < d) {
#pragma unroll
          for (int q = 0; q < V; ++q) acc = fmaf(va[u][q], vb[u][q], acc);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const float p = ps[r];
      float v = acc;
      if (d_ps_ext) v += d_ps_ext[r];
      if (dl_dbce != 0.f) v += dl_dbce * ((p - y[r]) / fmaxf((1.0f - p) * p, 1e-12f)) / (float)n;
      dps[r] = v;
    }
  }
}

// partial[rs][0][c] = sum_i x[i][c], partial[rs][1][c] = sum_i g[i] x[i][c]   over the block's rows
// FUSED (one row of <= 4096 scores, the training backward): g is not read but made here -- the workgroup's prologue repeats
// soft_topk_bwd_kernel<256>'s two block sums over (dps, scores) (same threads, same order: the same bits in every workgroup) and each
// wave turns dps[r] into g[r] for the rows it adds; the workgroups of column tile 0 also leave g[r] in g_out (sum_i g_i, dh).  One
// launch and one dependent round trip fewer than soft_topk_bwd + this sweep.
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void wcolsum_partial_kernel(const T* __restrict__ h, const float* __restrict__ g, int n,
                                                              int d, int row_splits, float* __restrict__ partial,
                                                              const float* __restrict__ xs, const float* __restrict__ ts,
                                                              float* __restrict__ g_out) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rs = blockIdx.y;
  const int col = (blockIdx.x * 64 + lane) * V;
  const int rows_per = (n + row_splits - 1) / row_splits;
  const int rb = rs * rows_per, re = min(n, rb + rows_per);
  __shared__ float red[4][64][2 * V + 1];
  float t = 0.f, sv = 1.f, suv = 0.f;
  if constexpr (FUSED) {
    t = ts[0];
    soft_topk_bwd_sums<256>(g, xs, t, n, reinterpret_cast<float (*)[4]>(&red[0][0][0]), sv, suv);   // (g = dps here)
    __syncthreads();                                   // red is reused below
  }
  float a0[V], a1[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
  if (col < d || (FUSED && blockIdx.x == 0)) {
    const int colc = min(col, d - V);
    for (int r0 = rb + wave; r0 < re; r0 += 32) {          // 8 rows in flight per wave (clamped); rows are added in order
      float v[8][V], gr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = min(r0 + 4 * u, re - 1);
        load_vec(h + (int64_t)r * d + colc, v[u]);
        if constexpr (FUSED) gr[u] = soft_topk_bwd_elem(g[r], xs[r], t, sv, suv);
        else gr[u] = g[r];
      }
      if constexpr (FUSED) {
        if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (r0 + 4 * u < re) g_out[r0 + 4 * u] = gr[u];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + 4 * u < re) {
#pragma unroll
          for (int i = 0; i < V; ++i) { a0[i] += v[u][i]; a1[i] = fmaf(gr[u], v[u][i], a1[i]); }
        }
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) { red[wave][lane][i] = a0[i]; red[wave][lane][V + i] = a1[i]; }
  __syncthreads();
  if (wave == 0 && col < d) {
    float* dst = partial + (int64_t)rs * 2 * d;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      dst[col + i] = (red[0][lane][i] + red[1][lane][i]) + (red[2][lane][i] + red[3][lane][i]);
      dst[d + col + i] = (red[0][lane][V + i] + red[1][lane][V + i]) + (red[2][lane][V + i] + red[3][lane][V + i]);
    }
  }
}

// xsum[c], gx[c], xbar[c] = xsum / N; block 0 also reduces sg = sum_i g_i
__global__ __launch_bounds__(256) void wcolsum_finish_kernel(const float* __restrict__ partial, const float* __restrict__ g,
                                                             int n, int d, int row_splits, float* __restrict__ xsum,
                                                             float* __restrict__ gx, float* __restrict__ xbar,
                                                             float* __restrict__ sg) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < d) {
    float a = 0.f, b = 0.f;
    for (int r0 = 0; r0 < row_splits; r0 += 16) {            // 2 x 16 loads in flight (clamped), added in rs order
      float ta[16], tb[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int64_t rs = min(r0 + u, row_splits - 1);
        ta[u] = partial[rs * 2 * d + c];
        tb[u] = partial[rs * 2 * d + d + c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (r0 + u < row_splits) { a += ta[u]; b += tb[u]; }
    }
    xsum[c] = a;
    gx[c] = b;
    xbar[c] = a / (float)n;
  }
  if (blockIdx.x == 0) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += g[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sg[0] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// dk[h] = (sum_ks part[ks][h] + bq[h] sg) rs / N ; dbk[h] = N dk[h] ; dbq[h] = kbar[h] rs sg ; a[h] = kbar[h] rs
template <typename TW>
__global__ __launch_bounds__(256) void dk_finish_kernel(const float* __restrict__ part, int KS, int hd,
                                                        const TW* __restrict__ bq, const float* __restrict__ kbar,
                                                        const float* __restrict__ sg, float rs, int n,
                                                        float* __restrict__ dk, float* __restrict__ a,
                                                        float* __restrict__ dbq, float* __restrict__ dbk) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= hd) return;
  const float v = strided_sum(part + h, KS, hd);             // (ks order, all slab loads in flight)
  const float s = sg[0];
  const float dkv = __builtin_fmaf(load_elem(bq + h), s, v) * rs / (float)n;     // (written out: the same bits in train_bwd_finish_kernel)
  dk[h] = dkv;
  dbk[h] = dkv * (float)n;
  const float kb = kbar[h] * rs;
  a[h] = kb;
  dbq[h] = kb * s;
}

// wcolsum_finish + kbar_finish + dk_finish of ONE row in one launch (training backward, small-batch form): block b finishes columns
// 256 b .. of xsum / gx / xbar and elements 256 b .. of kbar / dk / a / dbq / dbk; every block that needs sum_i g_i rebuilds it
// itself (<= 4096 values out of L2, wcolsum_finish_kernel's order).  Element for element the three kernels' arithmetic.
template <typename TW>
__global__ __launch_bounds__(256) void train_bwd_finish_kernel(const float* __restrict__ wpart, const float* __restrict__ g, int n, int d,
                                                               int row_splits, const float* __restrict__ part1, const float* __restrict__ dkraw,
                                                               int KS, int hd, const TW* __restrict__ bk, const TW* __restrict__ bq, float rs,
                                                               float* __restrict__ xsum, float* __restrict__ gx, float* __restrict__ xbar,
                                                               float* __restrict__ sg_out, float* __restrict__ kbar, float* __restrict__ dk,
                                                               float* __restrict__ a, float* __restrict__ dbq, float* __restrict__ dbk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < d) {
    float va = 0.f, vb = 0.f;
    for (int r0 = 0; r0 < row_splits; r0 += 16) {            // 2 x 16 loads in flight (clamped), added in rs order
      float ta[16], tb[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int64_t rsp = min(r0 + u, row_splits - 1);
        ta[u] = wpart[rsp * 2 * d + c];
        tb[u] = wpart[rsp * 2 * d + d + c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (r0 + u < row_splits) { va += ta[u]; vb += tb[u]; }
    }
    xsum[c] = va;
    gx[c] = vb;
    xbar[c] = va / (float)n;
  }
  if (blockIdx.x * 256 >= hd && blockIdx.x != 0) return;     // (block-uniform)
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += g[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  const float s = (red[0] + red[1]) + (red[2] + red[3]);
  if (blockIdx.x == 0 && threadIdx.x == 0) sg_out[0] = s;
  const int hh = c;
  if (hh >= hd) return;
  float v = 0.f;                                             // kbar_finish_kernel (M = 1)
  for (int k0 = 0; k0 < KS; k0 += 16) {
    float tk[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) tk[u] = part1[(int64_t)min(k0 + u, KS - 1) * hd + hh];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + u < KS) v += tk[u];
  }
  v += load_elem(bk + hh);
  kbar[hh] = v;
  const float vd = strided_sum(dkraw + hh, KS, hd);          // dk_finish_kernel
  const float dkv = __builtin_fmaf(load_elem(bq + hh), s, vd) * rs / (float)n;
  dk[hh] = dkv;
  dbk[hh] = dkv * (float)n;
  const float kb = v * rs;
  a[hh] = kb;
  dbq[hh] = kb * s;
}

// both rank-1 weight gradients in one launch: blockIdx.z picks (a0 (x) b0 -> out0) or (a1 (x) b1 -> out1)
__global__ __launch_bounds__(256) void outer_pair_kernel(const float* __restrict__ a0, const float* __restrict__ b0, float* __restrict__ out0,
                                                         const float* __restrict__ a1, const float* __restrict__ b1, float* __restrict__ out1,
                                                         int rows, int cols) {
  const float* a = blockIdx.z ? a1 : a0;
  const float* b = blockIdx.z ? b1 : b0;
  float* out = blockIdx.z ? out1 : out0;
  const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c4 >= cols) return;
  const f32x4 bv = *reinterpret_cast<const f32x4*>(b + c4);
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    const float av = a[r];
    f32x4 o = {av * bv[0], av * bv[1], av * bv[2], av * bv[3]};
    *reinterpret_cast<f32x4*>(out + (int64_t)r * cols + c4) = o;
  }
}

// Rank-1-factor exchange (vsel_lis_factors_to_grads): R payload rows a_i | gx_i | dk_i | xsum_i | dbq_i | dbk_i ->
//   dwq[r][c] = scale sum_i a_i[r] gx_i[c],  dwk[r][c] = scale sum_i dk_i[r] xsum_i[c]   (i in order; fp32)
// grid (ceil(cols / 1024), row blocks, 2 = {dwq, dwk}); a block stages its <= 64 x R left factors through LDS.
__global__ __launch_bounds__(256) void outer_sum_kernel(const float* __restrict__ payload, int R, int64_t row_stride, int hd, int d,
                                                        float scale, float* __restrict__ dwq, float* __restrict__ dwk) {
  const int which = blockIdx.z;
  const float* left = payload + (which ? hd + d : 0);              // a / dk   [hd]
  const float* right = payload + (which ? 2 * hd + d : hd);         // gx / xsum [d]
  float* out = which ? dwk : dwq;
  const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool c_ok = c4 < d;
  f32x4 bv[8];
  for (int r = blockIdx.y; r < hd; r += gridDim.y) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < R; i0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) bv[u] = *reinterpret_cast<const f32x4*>(right + (int64_t)min(i0 + u, R - 1) * row_stride + (c_ok ? c4 : 0));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < R) {
          const float av = left[(int64_t)(i0 + u) * row_stride + r];
          acc[0] = fmaf(av, bv[u][0], acc[0]);
          acc[1] = fmaf(av, bv[u][1], acc[1]);
          acc[2] = fmaf(av, bv[u][2], acc[2]);
          acc[3] = fmaf(av, bv[u][3], acc[3]);
        }
    }
    if (c_ok) *reinterpret_cast<f32x4*>(out + (int64_t)r * d + c4) = acc * scale;
  }
}

// dbq[h] = scale sum_i dbq_i[h], dbk likewise (i in order)
__global__ __launch_bounds__(256) void bias_sum_kernel(const float* __restrict__ payload, int R, int64_t row_stride, int hd, int d,
                                                       float scale, float* __restrict__ dbq, float* __restrict__ dbk) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= hd) return;
  const float* p = payload + 2 * (hd + d) + h;
  float s0 = 0.f, s1 = 0.f;
  for (int i = 0; i < R; ++i) {
    s0 += p[(int64_t)i * row_stride];
    s1 += p[(int64_t)i * row_stride + hd];
  }
  dbq[h] = s0 * scale;
  dbk[h] = s1 * scale;
}

// dh[i,:] = ps[i] d_hnew[i,:] + (g[i] rs) w[:] + u[:]      (first term dropped when dhn == NULL)
template <typename T>
__global__ __launch_bounds__(256) void dh_kernel(const T* __restrict__ dhn, const float* __restrict__ ps,
                                                 const float* __restrict__ g, const float* __restrict__ w,
                                                 const float* __restrict__ u, float rs, int n, int d, T* __restrict__ dh) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
    const float p = dhn ? ps[r] : 0.f, gr = g[r] * rs;
    for (int c = lane * V; c < d; c += 64 * V) {
      float v[V];
      if (dhn) {
        load_vec(dhn + (int64_t)r * d + c, v);
      } else {
#pragma unroll
        for (int q = 0; q < V; ++q) v[q] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < V; ++q) v[q] = fmaf(p, v[q], fmaf(gr, w[c + q], u[c + q]));
      store_vec(dh + (int64_t)r * d + c, v);
    }
  }
}

struct TrainPlan {
  LisPlan lis;
  int wsplits;
  size_t off_dps, off_g, off_wpart, off_xsum, off_gx, off_sg, off_dkraw, off_dk, off_a, off_u, total;
};

static TrainPlan make_train_plan(int64_t n, int64_t d, int64_t hd) {
  TrainPlan t{};
  t.lis = make_plan(1, n, d, hd);
  t.wsplits = t.lis.row_splits;
  size_t o = t.lis.total;
  auto take = [&](size_t nfloat) { size_t r = o; o += align_up(nfloat * sizeof(float), 256); return r; };
  t.off_dps = take(n);
  t.off_g = take(n);
  t.off_wpart = take((size_t)t.wsplits * 2 * d);
  t.off_xsum = take(d);
  t.off_gx = take(d);
  t.off_sg = take(1);
  t.off_dkraw = take((size_t)t.lis.ks1 * hd);
  t.off_dk = take(hd);
  t.off_a = take(hd);
  t.off_u = take(d);
  t.total = o;
  return t;
}

template <typename T, typename TW>
static int train_fwd_impl(hipStream_t st, const T* h, int64_t n, int64_t k, const vsel_scorer* sc, char* ws,
                          const TrainPlan& tp, T* h_new, float* ps, float* y, float* scores, float* ts, float* bce) {
  vsel_segments seg{1, n, n, k, k, nullptr, nullptr};
  int rc;
  if (small_path_ok(&seg, sc, tp.lis)) {
    // one segment: the four-launch small-batch form of the scores (lis_small.h; bit-identical to run_scores)
    if ((rc = run_colsum<T>(st, h, &seg, (int)sc->d, ws, tp.lis))) return rc;
    if ((rc = run_proj_small(st, &seg, sc, ws, tp.lis, nullptr))) return rc;
    rc = run_score_small<T>(st, h, &seg, sc, ws, tp.lis, scores, nullptr);
  } else {
    rc = run_scores<T, TW>(st, h, &seg, sc, ws, tp.lis, scores);
  }
  if (rc) return rc;
  const int d = (int)sc->d;
  if (n <= 4096) {
    // soft top-k + hard top-k mask + BCE of the one row in one launch
    if (n <= 1024) VSEL_LAUNCH(train_tail_kernel<4>, dim3(1), dim3(256), 0, st, scores, (int)n, (int)k, ps, ts, y, bce);
    else VSEL_LAUNCH(train_tail_kernel<16>, dim3(1), dim3(256), 0, st, scores, (int)n, (int)k, ps, ts, y, bce);
    VSEL_AFTER_LAUNCH(st, "train_tail_kernel");
    VSEL_LAUNCH((mask_apply_kernel<T>), dim3((unsigned)std::min<int64_t>(cdiv(n, 4), 4096)), dim3(256), 0, st, h, ps,
                       (int)n, d, h_new);
    VSEL_AFTER_LAUNCH(st, "mask_apply_kernel");
    return VSEL_OK;
  }
  rc = launch_soft_topk_fwd(st, scores, 1, n, k, ps, ts);
  if (rc) return rc;
  rc = launch_select(st, scores, &seg, nullptr, y);
  if (rc) return rc;
  VSEL_LAUNCH((mask_apply_kernel<T>), dim3((unsigned)std::min<int64_t>(cdiv(n, 4), 4096)), dim3(256), 0, st, h, ps,
                     (int)n, d, h_new);
  VSEL_AFTER_LAUNCH(st, "mask_apply_kernel");
  VSEL_LAUNCH(bce_kernel, dim3(1), dim3(1024), 0, st, ps, y, (int)n, bce);
  VSEL_AFTER_LAUNCH(st, "bce_kernel");
  return VSEL_OK;
}

// Backward of scores = scorer(h) given g = dL/dscores [N] (g must live outside the workspace or at tp.off_g).
template <typename T, typename TW>
static int scores_bwd_impl(hipStream_t st, const float* g, const T* h, int64_t n, const vsel_scorer* sc, char* ws,
                           const TrainPlan& tp, float* dwq, float* dbq, float* dwk, float* dbk, T* dh, const T* dhn,
                           const float* ps, float* factors = nullptr, const float* fuse_dps = nullptr,
                           const float* fuse_scores = nullptr, const float* fuse_ts = nullptr) {
  constexpr int V = Elem<T>::kVec;
  const int d = (int)sc->d, hd = (int)sc->hd;
  const LisPlan& p = tp.lis;
  float* wpart = (float*)(ws + tp.off_wpart);
  // factors != NULL: the rank-1 factors of the weight gradients go to the caller (a | gx | dk | xsum) and the two dense
  // [Hd, D] writes are skipped (vsel_lis_train_bwd_factors)
  float* xsum = factors ? factors + 2 * hd + d : (float*)(ws + tp.off_xsum);
  float* gx = factors ? factors + hd : (float*)(ws + tp.off_gx);
  float* sg = (float*)(ws + tp.off_sg);
  float* dkraw = (float*)(ws + tp.off_dkraw);
  float* dk = factors ? factors + hd + d : (float*)(ws + tp.off_dk);
  float* a = factors ? factors : (float*)(ws + tp.off_a);
  float* u = (float*)(ws + tp.off_u);
  float* xbar = (float*)(ws + p.off_xbar);
  float* part1 = (float*)(ws + p.off_part1);
  float* kbar = (float*)(ws + p.off_kbar);
  float* c = (float*)(ws + p.off_c);
  float* part2 = (float*)(ws + p.off_part2);
  float* w = (float*)(ws + p.off_w);
  const float rs = 1.0f / (float)sqrt((double)hd);
  const unsigned row_blocks = (unsigned)std::min<int64_t>(cdiv(n, 4), 4096);

  vsel_segments seg1{1, n, n, 1, 1, nullptr, nullptr};
  const bool small = small_path_ok(&seg1, sc, p);
  // fused = dps / scores / ts given instead of g (train_bwd_impl, one row of <= 4096 scores, bf16 weights on the small-batch form):
  // five launches -- row dots | weighted column sums with the soft top-k backward in their prologue | both projections | one finish |
  // both rank-1 writes -- instead of ten (knob TRAIN_FUSED; the same bits either way)
  const bool fused = fuse_dps != nullptr;
  if (fused) {
    if constexpr (std::is_same<TW, bf16_t>::value) {
      float* gbuf = (float*)(ws + tp.off_g);
      VSEL_LAUNCH((wcolsum_partial_kernel<T, true>), dim3((unsigned)cdiv(d, 64 * V), tp.wsplits), dim3(256), 0, st, h, fuse_dps, (int)n,
                  d, tp.wsplits, wpart, fuse_scores, fuse_ts, gbuf);
      VSEL_AFTER_LAUNCH(st, "wcolsum_partial_kernel");
      VSEL_LAUNCH(proj_nt_small_pair_kernel, dim3((unsigned)cdiv(hd, 32), p.ks1, 2), dim3(64), 0, st, wpart, wpart + d, make_view(&seg1), 1,
                  tp.wsplits, (const uint16_t*)sc->wk, (const uint16_t*)sc->wq, hd, d, p.kslice1, part1, dkraw, (int64_t)2 * d, 1, 0);
      VSEL_AFTER_LAUNCH(st, "proj_nt_small_pair_kernel");
      VSEL_LAUNCH((train_bwd_finish_kernel<TW>), dim3((unsigned)std::max<int64_t>(cdiv(d, 256), cdiv(hd, 256))), dim3(256), 0, st, wpart, gbuf,
                  (int)n, d, tp.wsplits, part1, dkraw, p.ks1, hd, (const TW*)sc->bk, (const TW*)sc->bq, rs, xsum, gx, xbar, sg, kbar, dk, a,
                  dbq, dbk);
      VSEL_AFTER_LAUNCH(st, "train_bwd_finish_kernel");
      if (!factors) {
        const dim3 og((unsigned)cdiv(d, 1024), (unsigned)std::min<int>(hd, 512), 2);
        VSEL_LAUNCH(outer_pair_kernel, og, dim3(256), 0, st, a, gx, dwq, dk, xsum, dwk, hd, d);
        VSEL_AFTER_LAUNCH(st, "outer_pair_kernel");
      }
    }
    g = (const float*)(ws + tp.off_g);               // (dh below)
  } else {
  VSEL_LAUNCH((wcolsum_partial_kernel<T, false>), dim3((unsigned)cdiv(d, 64 * V), tp.wsplits), dim3(256), 0, st, h, g, (int)n,
                     d, tp.wsplits, wpart, nullptr, nullptr, nullptr);
  VSEL_AFTER_LAUNCH(st, "wcolsum_partial_kernel");
  VSEL_LAUNCH(wcolsum_finish_kernel, dim3((unsigned)cdiv(d, 256)), dim3(256), 0, st, wpart, g, (int)n, d, tp.wsplits,
                     xsum, gx, xbar, sg);
  VSEL_AFTER_LAUNCH(st, "wcolsum_finish_kernel");
  // kbar = Wk xbar + bk.  bf16 weights: the forward's single-wave-per-tile bf16x3 kernel, its prologue summing the row splits
  // of wpart itself (same bits as the forward's kbar); else the fp32-input MFMA form
  if (small) {
    VSEL_LAUNCH(proj_nt_small_kernel, dim3((unsigned)cdiv(hd, 32), p.ks1), dim3(64), 0, st, wpart, make_view(&seg1), 1,
                       tp.wsplits, (const uint16_t*)sc->wk, hd, d, p.kslice1, part1, (int64_t)2 * d, 1);
    VSEL_AFTER_LAUNCH(st, "proj_nt_small_kernel");
  } else {
    VSEL_LAUNCH((gemm_nt_kernel<TW>), dim3((unsigned)cdiv(hd, 32), 1, p.ks1), dim3(64), 0, st, xbar, (const TW*)sc->wk, 1,
                       hd, d, p.kslice1, part1);
    VSEL_AFTER_LAUNCH(st, "gemm_nt_kernel");
  }
  VSEL_LAUNCH((kbar_finish_kernel<TW>), dim3(1), dim3(256), 0, st, part1, p.ks1, 1, hd, (const TW*)sc->bk,
                     (const TW*)sc->bq, kbar, c);
  VSEL_AFTER_LAUNCH(st, "kbar_finish_kernel");
  // dk = (Wq gx + bq sg) rs / N
  if (small) {
    VSEL_LAUNCH(proj_nt_small_kernel, dim3((unsigned)cdiv(hd, 32), p.ks1), dim3(64), 0, st, wpart + d, make_view(&seg1), 1,
                       tp.wsplits, (const uint16_t*)sc->wq, hd, d, p.kslice1, dkraw, (int64_t)2 * d, 0);
    VSEL_AFTER_LAUNCH(st, "proj_nt_small_kernel");
  } else {
    VSEL_LAUNCH((gemm_nt_kernel<TW>), dim3((unsigned)cdiv(hd, 32), 1, p.ks1), dim3(64), 0, st, gx, (const TW*)sc->wq, 1,
                       hd, d, p.kslice1, dkraw);
    VSEL_AFTER_LAUNCH(st, "gemm_nt_kernel");
  }
  VSEL_LAUNCH((dk_finish_kernel<TW>), dim3((unsigned)cdiv(hd, 256)), dim3(256), 0, st, dkraw, p.ks1, hd,
                     (const TW*)sc->bq, kbar, sg, rs, (int)n, dk, a, dbq, dbk);
  VSEL_AFTER_LAUNCH(st, "dk_finish_kernel");
  if (!factors) {
    const dim3 og((unsigned)cdiv(d, 1024), (unsigned)std::min<int>(hd, 512), 2);
    VSEL_LAUNCH(outer_pair_kernel, og, dim3(256), 0, st, a, gx, dwq, dk, xsum, dwk, hd, d);
    VSEL_AFTER_LAUNCH(st, "outer_pair_kernel");
  }
  }
  if (dh) {
    VSEL_LAUNCH((gemm_nn_kernel<TW>), dim3((unsigned)cdiv(d, 256), 1, p.ks2), dim3(64), 0, st, kbar, (const TW*)sc->wq, 1,
                       d, hd, p.kslice2, part2);
    VSEL_AFTER_LAUNCH(st, "gemm_nn_kernel");
    VSEL_LAUNCH(slice_sum_kernel, dim3((unsigned)cdiv(d, 256)), dim3(256), 0, st, part2, p.ks2, (int64_t)d, w);
    VSEL_AFTER_LAUNCH(st, "slice_sum_kernel");
    VSEL_LAUNCH((gemm_nn_kernel<TW>), dim3((unsigned)cdiv(d, 256), 1, p.ks2), dim3(64), 0, st, dk, (const TW*)sc->wk, 1, d,
                       hd, p.kslice2, part2);
    VSEL_AFTER_LAUNCH(st, "gemm_nn_kernel");
    VSEL_LAUNCH(slice_sum_kernel, dim3((unsigned)cdiv(d, 256)), dim3(256), 0, st, part2, p.ks2, (int64_t)d, u);
    VSEL_AFTER_LAUNCH(st, "slice_sum_kernel");
    VSEL_LAUNCH((dh_kernel<T>), dim3(row_blocks), dim3(256), 0, st, dhn, ps, g, w, u, rs, (int)n, d, dh);
    VSEL_AFTER_LAUNCH(st, "dh_kernel");
  }
  return VSEL_OK;
}

template <typename T, typename TW>
static int train_bwd_impl(hipStream_t st, const T* dhn, const T* h, int64_t n, const vsel_scorer* sc, const float* ps,
                          const float* y, const float* scores, const float* ts, const float* d_ps_ext, float dl_dbce,
                          char* ws, const TrainPlan& tp, float* dwq, float* dbq, float* dwk, float* dbk, T* dh,
                          float* factors = nullptr) {
  const int d = (int)sc->d;
  float* dps = (float*)(ws + tp.off_dps);
  float* g = (float*)(ws + tp.off_g);
  const unsigned row_blocks = (unsigned)std::min<int64_t>(cdiv(n, 4), 4096);
  VSEL_LAUNCH((rowdot_kernel<T>), dim3(row_blocks), dim3(256), 0, st, dhn, h, ps, y, d_ps_ext, dl_dbce, (int)n, d, dps);
  VSEL_AFTER_LAUNCH(st, "rowdot_kernel");
  vsel_segments seg1{1, n, n, 1, 1, nullptr, nullptr};
  if (std::is_same<TW, bf16_t>::value && n <= 4096 && knob(VSEL_KNOB_TRAIN_FUSED) != 0 && small_path_ok(&seg1, sc, tp.lis))
    return scores_bwd_impl<T, TW>(st, nullptr, h, n, sc, ws, tp, dwq, dbq, dwk, dbk, dh, dhn, ps, factors, dps, scores, ts);
  int rc = launch_soft_topk_bwd(st, dps, scores, ts, 1, n, g);
  if (rc) return rc;
  return scores_bwd_impl<T, TW>(st, g, h, n, sc, ws, tp, dwq, dbq, dwk, dbk, dh, dhn, ps, factors);
}

}  // namespace vsel

using namespace vsel;

extern "C" size_t vsel_lis_train_workspace_bytes(int64_t n, int64_t d, int64_t hd) {
  if (n < 1 || d < 1 || hd < 1) return 0;
  return make_train_plan(n, d, hd).total;
}

static int train_checks(const void* h, vsel_dtype hdtype, int64_t n, const vsel_scorer* sc, void* ws, size_t ws_bytes,
                        TrainPlan* tp) {
  if (!h) return fail(VSEL_ERR_INVALID, "h is NULL");
  if (n < 1 || n >= (1ll << 31)) return fail(VSEL_ERR_INVALID, "bad n=%lld", (long long)n);
  int st = check_scorer(sc, hdtype);
  if (st) return st;
  *tp = make_train_plan(n, sc->d, sc->hd);
  if (!ws || ws_bytes < tp->total) return fail(VSEL_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, tp->total);
  if (((uintptr_t)h | (uintptr_t)ws | (uintptr_t)sc->wq | (uintptr_t)sc->wk) & 15)
    return fail(VSEL_ERR_INVALID, "h / workspace / weights must be 16-byte aligned");
  return VSEL_OK;
}

#define VSEL_DISPATCH2(hdtype, wdtype, CALL)                                        \
  do {                                                                              \
    if ((hdtype) == VSEL_BF16 && (wdtype) == VSEL_BF16) { using T = bf16_t; using TW = bf16_t; return CALL; } \
    if ((hdtype) == VSEL_BF16 && (wdtype) == VSEL_F32) { using T = bf16_t; using TW = float; return CALL; }   \
    if ((hdtype) == VSEL_F32 && (wdtype) == VSEL_BF16) { using T = float; using TW = bf16_t; return CALL; }   \
    { using T = float; using TW = float; return CALL; }                             \
  } while (0)

extern "C" int vsel_lis_train_fwd(void* stream, const void* h, vsel_dtype hdtype, int64_t n, int64_t k,
                                  const vsel_scorer* sc, void* ws, size_t ws_bytes, void* h_new, float* ps, float* y,
                                  float* scores, float* ts, float* bce) {
  TrainPlan tp;
  int st = train_checks(h, hdtype, n, sc, ws, ws_bytes, &tp);
  if (st) return st;
  if (!h_new || !ps || !y || !scores || !ts || !bce) return fail(VSEL_ERR_INVALID, "output pointer is NULL");
  // the reference's _find_ts asserts 0 < k < n (FT/compression_method/selector_model.py:75)
  if (!(0 < k && k < n)) return fail(VSEL_ERR_INVALID, "training needs 0 < k < n (k=%lld, n=%lld)", (long long)k, (long long)n);
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  VSEL_DISPATCH2(hdtype, sc->wdtype,
                 (train_fwd_impl<T, TW>(s, (const T*)h, n, k, sc, (char*)ws, tp, (T*)h_new, ps, y, scores, ts, bce)));
}

extern "C" int vsel_lis_train_bwd(void* stream, const void* d_hnew, const void* h, vsel_dtype hdtype, int64_t n,
                                  const vsel_scorer* sc, const float* ps, const float* y, const float* scores,
                                  const float* ts, const float* d_ps_ext, float dl_dbce, void* ws, size_t ws_bytes,
                                  float* dwq, float* dbq, float* dwk, float* dbk, void* dh) {
  TrainPlan tp;
  int st = train_checks(h, hdtype, n, sc, ws, ws_bytes, &tp);
  if (st) return st;
  if (!d_hnew || !ps || !y || !scores || !ts || !dwq || !dbq || !dwk || !dbk) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (sc->d % 4) return fail(VSEL_ERR_UNSUPPORTED, "D must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  VSEL_DISPATCH2(hdtype, sc->wdtype,
                 (train_bwd_impl<T, TW>(s, (const T*)d_hnew, (const T*)h, n, sc, ps, y, scores, ts, d_ps_ext, dl_dbce,
                                        (char*)ws, tp, dwq, dbq, dwk, dbk, (T*)dh)));
}

extern "C" int vsel_lis_train_bwd_factors(void* stream, const void* d_hnew, const void* h, vsel_dtype hdtype, int64_t n,
                                          const vsel_scorer* sc, const float* ps, const float* y, const float* scores,
                                          const float* ts, const float* d_ps_ext, float dl_dbce, void* ws, size_t ws_bytes,
                                          float* factors, float* dbq, float* dbk, void* dh) {
  TrainPlan tp;
  int st = train_checks(h, hdtype, n, sc, ws, ws_bytes, &tp);
  if (st) return st;
  if (!d_hnew || !ps || !y || !scores || !ts || !factors || !dbq || !dbk) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (sc->d % 4 || sc->hd % 4) return fail(VSEL_ERR_UNSUPPORTED, "D and Hd must be multiples of 4");
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  VSEL_DISPATCH2(hdtype, sc->wdtype,
                 (train_bwd_impl<T, TW>(s, (const T*)d_hnew, (const T*)h, n, sc, ps, y, scores, ts, d_ps_ext, dl_dbce,
                                        (char*)ws, tp, nullptr, dbq, nullptr, dbk, (T*)dh, factors)));
}

extern "C" int vsel_lis_factors_to_grads(void* stream, const float* payload, int64_t n_rows, int64_t hd, int64_t d, float scale,
                                        float* dwq, float* dbq, float* dwk, float* dbk) {
  if (!payload || !dwq || !dbq || !dwk || !dbk) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (n_rows < 1 || hd < 1 || d < 1 || n_rows > (1 << 20)) return fail(VSEL_ERR_INVALID, "bad shape");
  if (d % 4 || hd % 4) return fail(VSEL_ERR_UNSUPPORTED, "D and Hd must be multiples of 4");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  const int64_t row = 2 * (hd + d) + 2 * hd;
  const dim3 og((unsigned)cdiv(d, 1024), (unsigned)std::min<int64_t>(hd, 512), 2);
  VSEL_LAUNCH(outer_sum_kernel, og, dim3(256), 0, st, payload, (int)n_rows, row, (int)hd, (int)d, scale, dwq, dwk);
  VSEL_AFTER_LAUNCH(st, "outer_sum_kernel");
  VSEL_LAUNCH(bias_sum_kernel, dim3((unsigned)cdiv(hd, 256)), dim3(256), 0, st, payload, (int)n_rows, row, (int)hd, (int)d,
                     scale, dbq, dbk);
  VSEL_AFTER_LAUNCH(st, "bias_sum_kernel");
  return VSEL_OK;
}

extern "C" int vsel_lis_scores_bwd(void* stream, const float* g, const void* h, vsel_dtype hdtype, int64_t n,
                                   const vsel_scorer* sc, void* ws, size_t ws_bytes, float* dwq, float* dbq, float* dwk,
                                   float* dbk, void* dh) {
  TrainPlan tp;
  int st = train_checks(h, hdtype, n, sc, ws, ws_bytes, &tp);
  if (st) return st;
  if (!g || !dwq || !dbq || !dwk || !dbk) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (sc->d % 4) return fail(VSEL_ERR_UNSUPPORTED, "D must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  VSEL_DISPATCH2(hdtype, sc->wdtype,
                 (scores_bwd_impl<T, TW>(s, g, (const T*)h, n, sc, (char*)ws, tp, dwq, dbq, dwk, dbk, (T*)dh, (const T*)nullptr,
                                         (const float*)nullptr)));
}
