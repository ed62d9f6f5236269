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
// BCE: one EXTRA workgroup (the last) computes the constraint loss of the row instead of rows of h_new -- bce = mean_i
// -(y_i max(log p_i, -100) + (1 - y_i) max(log(1 - p_i), -100)) (:310, ATen clamp) with the thread -> element map and the summation
// order train_tail_kernel used when it still did this itself (thread (wave, lane): elements wave kpw 64 + lane + 64 j in j order, then
// wave, then waves 0..3); ps comes from the tail's workgroup 0, y from its workgroup 1 (launch order), so the loss no longer sits on the
// soft top-k's critical path.
template <typename T, bool BCE>
__global__ __launch_bounds__(256) void mask_apply_kernel(const T* __restrict__ h, const float* __restrict__ ps, int n,
                                                         int d, T* __restrict__ out, const float* __restrict__ y,
                                                         float* __restrict__ bce) {
  constexpr int V = Elem<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_blocks = BCE ? gridDim.x - 1 : gridDim.x;
  if constexpr (BCE) {
    if ((int)blockIdx.x == row_blocks) {
      __shared__ float red[2][4];
      const int kpw = (n + 255) / 256;
      const int e0 = wave * kpw * 64 + lane;
      float bacc = 0.f;
      for (int j0 = 0; j0 < kpw; j0 += 16) {           // 2 x 16 loads in flight (clamped), added in j order
        float tp[16], ty[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int e = min(e0 + 64 * (j0 + u), n - 1);
          tp[u] = ps[e];
          ty[u] = y[e];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (j0 + u < kpw && e0 + 64 * (j0 + u) < n) {
            const float p = tp[u], yv = ty[u];
            bacc += (yv - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - yv * fmaxf(logf(p), -100.0f);
          }
      }
      const float btot = block_sum<4>(bacc, red, 0);
      if (threadIdx.x == 0) bce[0] = btot / (float)n;
      return;
    }
  }
  for (int r = blockIdx.x * 4 + wave; r < n; r += row_blocks * 4) {
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

// The forward's tail for one row of N <= 256 KPT scores in ONE launch of TWO workgroups (three launches before round 2, one workgroup
// until round 6: the root and the select are independent, and run back to back they were 12.8 us of a 49 us forward):
//   workgroup 0:  ts, ps = _find_ts(scores, k)   (find_ts_newton, as soft_topk_fwd_kernel)
//   workgroup 1:  y      = hard top-k mask of the scores (topk_select_reg_kernel's integer select: larger key first, then lower index)
//   bce = mean_i BCE(ps_i, y_i) is left to mask_apply_kernel's extra workgroup (next launch).
// Wave w owns the contiguous elements [w span, (w + 1) span), element j of a lane = e0 + 64 j.
template <int KPT>
__global__ __launch_bounds__(256) void train_tail_kernel(const float* __restrict__ xs, int n, int k, float* __restrict__ ps,
                                                         float* __restrict__ ts, float* __restrict__ y) {
  constexpr int NT = 256, NW = 4;
  __shared__ float red[6][NW];
  __shared__ uint32_t hist[4][256];
  __shared__ uint32_t wtot[NW][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kpw = (n + NT - 1) / NT;                  // 64-element groups per wave (<= KPT)
  const int e0 = wave * kpw * 64 + lane;
  float xr[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) xr[j] = xs[min(e0 + 64 * j, n - 1)];      // unconditional (clamped) loads
  if (blockIdx.x == 0) {
    float mx = -INFINITY, mn = INFINITY, sx = 0.f;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const bool ok = j < kpw && e0 + 64 * j < n;
      if (ok) { mx = fmaxf(mx, xr[j]); mn = fminf(mn, xr[j]); sx += xr[j]; }
      else xr[j] = -INFINITY;                           // sigmoid(-inf + t) = 0: padding never contributes
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    if (lane == 0) { red[4][wave] = mx; red[5][wave] = mn; }
    __syncthreads();
    mx = red[4][0]; mn = red[5][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) { mx = fmaxf(mx, red[4][w]); mn = fminf(mn, red[5][w]); }
    // ---- _find_ts (selector_model.py:72-86): the same root by bracketed Newton steps (softtopk.h, find_ts_newton) --------------
    __syncthreads();                                  // red[4..5] are reused by the iteration
    const float t = find_ts_newton<NW, NW, KPT>(xr, sx, mx, mn, n, k, red);
    if (tid == 0) ts[0] = t;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int e = e0 + 64 * j;
      if (j < kpw && e < n) ps[e] = sigmoidf_ref(xr[j] + t);          // :86
    }
    return;
  }
  // ---- hard top-k mask: 4-pass radix select on the ordered keys, then the ordered tie rule ------------------------------
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) hist[p4][tid] = 0;
  uint32_t key[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) key[j] = (j < kpw && e0 + 64 * j < n) ? order_key(xr[j]) : 0u;
  __syncthreads();                                    // the zeroed histograms
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
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int e = e0 + 64 * j;
    const bool valid = j < kpw && e < n;
    const bool eq = (beq[j] >> lane) & 1ull;
    const uint32_t eq_before = run_eq + __popcll(beq[j] & below);
    const bool sel = valid && (key[j] > thr || (eq && eq_before < need));
    if (valid) y[e] = sel ? 1.0f : 0.0f;
    run_eq += __popcll(beq[j]);
  }
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
    if (d <= 8 * 64 * V) {
      // both rows whole in flight (2 x <= 8 loads per lane, clamped): one round trip per row pair instead of two; the FMAs in the
      // order of the loop below
      float va[8][V], vb[8][V];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int cc = min(lane * V + u * 64 * V, d - V);
        load_vec(a + cc, va[u]);
        load_vec(b + cc, vb[u]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (lane * V + u * 64 * V < d) {
#pragma unroll
          for (int q = 0; q < V; ++q) acc = fmaf(va[u][q], vb[u][q], acc);
        }
    } else
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
// NWORD = 32-bit words per lane and row (4 / 2 / 1: 16- / 8- / 4-byte loads): the narrower forms give 2 x / 4 x the column tiles
// when (column tiles x row chunks) would leave CUs idle -- one image is 7 x 18 = 126 workgroups at 16 bytes per lane, each pulling
// 128 KB through ONE CU's memory pipe.  Per column the rows are added in the same order whatever the width: the same bits.
template <typename T, bool FUSED, int NWORD>
__global__ __launch_bounds__(256) void wcolsum_partial_kernel(const T* __restrict__ h, const float* __restrict__ g, int n,
                                                              int d, int row_splits, float* __restrict__ partial,
                                                              const float* __restrict__ xs, const float* __restrict__ ts,
                                                              float* __restrict__ g_out) {
  constexpr int V = NWORD * (4 / (int)sizeof(T));                   // elements per lane
  typedef uint32_t raw_t __attribute__((ext_vector_type(NWORD)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rs = blockIdx.y;
  const int col = (blockIdx.x * 64 + lane) * V;
  const int rows_per = (n + row_splits - 1) / row_splits;           // <= 128 (make_train_plan: the forward's 128-row chunks)
  const int rb = rs * rows_per, re = min(n, rb + rows_per);
  __shared__ float red[4][64][2 * V + 1];
  __shared__ float sred[2][4];
  // (1) the wave's whole share of the chunk -- rows rb + wave + 4 q, q < 32 -- goes in flight at once as raw vectors (the launch is
  // latency-bound: one round trip instead of four), under the prologue below
  raw_t raw[32];
  {
    const T* base = h + min(col, d - V);
#pragma unroll
    for (int q = 0; q < 32; ++q) raw[q] = *reinterpret_cast<const raw_t*>(base + (int64_t)min(rb + wave + 4 * q, re - 1) * d);
  }
  // (2) g of those rows: lane q holds row q's (one soft top-k backward element per lane instead of 32 uniform ones per wave)
  const int myr = rb + wave + 4 * (lane & 31);
  float gl = 0.f;
  if constexpr (FUSED) {
    const float t = ts[0];
    float sv, suv;
    soft_topk_bwd_sums<256>(g, xs, t, n, sred, sv, suv);            // (g = dps here)
    if (myr < re) gl = soft_topk_bwd_elem(g[myr], xs[myr], t, sv, suv);
    if (blockIdx.x == 0 && lane < 32 && myr < re) g_out[myr] = gl;
  } else {
    if (myr < re) gl = g[myr];
  }
  // (3) rows added in order q = 0, 1, ... (the order of the 8-rows-per-step loop this replaces)
  float a0[V], a1[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    if (rb + wave + 4 * q < re) {                      // (wave-uniform)
      const float gr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl), q));
      float v[V];
#pragma unroll
      for (int wd = 0; wd < NWORD; ++wd) {
        const uint32_t word = raw[q][wd];
        if constexpr (sizeof(T) == 2) {
          v[2 * wd] = __uint_as_float(word << 16);
          v[2 * wd + 1] = __uint_as_float(word & 0xffff0000u);
        } else {
          v[wd] = __uint_as_float(word);
        }
      }
#pragma unroll
      for (int i = 0; i < V; ++i) { a0[i] += v[i]; a1[i] = fmaf(gr, v[i], a1[i]); }
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

// the widest loads (16 / 8 / 4 bytes per lane) that still give ~200 workgroups; D must be a multiple of the lane's element count
template <typename T, bool FUSED>
static int launch_wcolsum(hipStream_t st, const T* h, const float* g, int n, int d, int wsplits, float* wpart, const float* xs,
                          const float* ts, float* g_out) {
  constexpr int E = 4 / (int)sizeof(T);                              // elements per 32-bit word
  int nw = 4;
  while (nw > 1 && (cdiv(d, 64 * nw * E) * wsplits < 200 || d % (nw * E) != 0)) nw >>= 1;
  if (d % (nw * E) != 0) return fail(VSEL_ERR_UNSUPPORTED, "D must be a multiple of %d", nw * E);
  const dim3 grid((unsigned)cdiv(d, 64 * nw * E), wsplits);
  if (nw == 4) VSEL_LAUNCH((wcolsum_partial_kernel<T, FUSED, 4>), grid, dim3(256), 0, st, h, g, n, d, wsplits, wpart, xs, ts, g_out);
  else if (nw == 2) VSEL_LAUNCH((wcolsum_partial_kernel<T, FUSED, 2>), grid, dim3(256), 0, st, h, g, n, d, wsplits, wpart, xs, ts, g_out);
  else VSEL_LAUNCH((wcolsum_partial_kernel<T, FUSED, 1>), grid, dim3(256), 0, st, h, g, n, d, wsplits, wpart, xs, ts, g_out);
  VSEL_AFTER_LAUNCH(st, "wcolsum_partial_kernel");
  return VSEL_OK;
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
  // Everything a thread will add is LOADED first (clamped, unconditional: one round trip for the whole kernel instead of one per
  // stage -- the stages are a few hundred bytes each and every one of them used to wait ~1.5 us for its first line), then added in the
  // three kernels' orders.  Rare shapes with more than 16 row splits / 24 k-slices / 16 x 256 scores take the looped tails below.
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int cc = min(c, d - 1), hh = min(c, hd - 1);
  const bool do_h = blockIdx.x * 256 < hd || blockIdx.x == 0;           // (block-uniform) this block needs sum_i g_i
  float ta[16], tb[16], tg[16], tk[16], td[24];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int64_t rsp = min(u, row_splits - 1);
    ta[u] = wpart[rsp * 2 * d + cc];
    tb[u] = wpart[rsp * 2 * d + d + cc];
  }
  if (do_h) {
#pragma unroll
    for (int u = 0; u < 16; ++u) tg[u] = g[min((int)threadIdx.x + 256 * u, n - 1)];
#pragma unroll
    for (int u = 0; u < 16; ++u) tk[u] = part1[(int64_t)min(u, KS - 1) * hd + hh];
#pragma unroll
    for (int u = 0; u < 24; ++u) td[u] = dkraw[(int64_t)min(u, KS - 1) * hd + hh];
  }
  const float bkv = load_elem(bk + hh), bqv = load_elem(bq + hh);
  if (c < d) {
    float va = 0.f, vb = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (u < row_splits) { va += ta[u]; vb += tb[u]; }
    for (int r0 = 16; r0 < row_splits; r0 += 16) {           // (more than 16 row splits: wcolsum_finish_kernel's loop, same order)
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
  if (!do_h) return;
  __shared__ float red[4];
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if ((int)threadIdx.x + 256 * u < n) acc += tg[u];
  for (int i = threadIdx.x + 4096; i < n; i += 256) acc += g[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  const float s = (red[0] + red[1]) + (red[2] + red[3]);
  if (blockIdx.x == 0 && threadIdx.x == 0) sg_out[0] = s;
  if (c >= hd) return;
  float v = 0.f;                                             // kbar_finish_kernel (M = 1): ks order, 16 per batch
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if (u < KS) v += tk[u];
  for (int k0 = 16; k0 < KS; k0 += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) tk[u] = part1[(int64_t)min(k0 + u, KS - 1) * hd + hh];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + u < KS) v += tk[u];
  }
  v += bkv;
  kbar[hh] = v;
  float vd = 0.f;                                            // dk_finish_kernel: strided_sum's order (ks order, 24 per batch)
#pragma unroll
  for (int u = 0; u < 24; ++u)
    if (u < KS) vd += td[u];
  for (int k0 = 24; k0 < KS; k0 += 24) {
#pragma unroll
    for (int u = 0; u < 24; ++u) td[u] = dkraw[(int64_t)min(k0 + u, KS - 1) * hd + hh];
#pragma unroll
    for (int u = 0; u < 24; ++u)
      if (k0 + u < KS) vd += td[u];
  }
  const float dkv = __builtin_fmaf(bqv, s, vd) * rs / (float)n;
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

// train_bwd_finish_kernel AND both rank-1 writes in one launch (the dense backward; the factor form has no rank-1 writes and keeps the
// finish kernel): grid (ceil(D / 1024), kOuterRowGroups, 2 = {dWq, dWk}).  A workgroup rebuilds what it multiplies -- its 1024
// columns of gx (dWq) / xsum (dWk) from the row-chunk partials, the a (dWq) / dk (dWk) of its <= 64 rows from the projection slabs,
// sum_i g_i -- all loads up front, operation for operation the finish kernel's arithmetic (same bits), then streams its rows of the
// gradient.  The row group 0 workgroups also leave gx / xsum / xbar, the column tile 0 workgroups kbar, a, dbq (dWq side) / dk, dbk
// (dWk side), workgroup (0, 0, 0) sum_i g_i: everything the finish kernel left for the token-gradient stage.
constexpr int kOuterRowGroups = 64;     // (32 / 128 row groups and non-temporal stores measured within +-1.5 us of this: profiles/EXPERIMENTS.md)
template <typename TW>
__global__ __launch_bounds__(256) void outer_finish_pair_kernel(const float* __restrict__ wpart, const float* __restrict__ g, int n, int d,
                                                                int row_splits, const float* __restrict__ part1, const float* __restrict__ dkraw,
                                                                int KS, int hd, const TW* __restrict__ bk, const TW* __restrict__ bq, float rs,
                                                                float* __restrict__ xsum, float* __restrict__ gx, float* __restrict__ xbar,
                                                                float* __restrict__ sg_out, float* __restrict__ kbar, float* __restrict__ dk,
                                                                float* __restrict__ a, float* __restrict__ dbq, float* __restrict__ dbk,
                                                                float* __restrict__ dwq, float* __restrict__ dwk) {
  const int which = blockIdx.z;                                       // 0: dWq = a (x) gx      1: dWk = dk (x) xsum
  const int tid = threadIdx.x;
  const int c4 = (blockIdx.x * 256 + tid) * 4;
  const bool c_ok = c4 < d;
  const int by = blockIdx.y;
  const int my_row = by + kOuterRowGroups * tid;                       // thread t < rows of the block: its row factor
  const bool r_ok = tid < 64 && my_row < hd;
  const int hh = r_ok ? my_row : 0;
  __shared__ float red[4];
  __shared__ float left[64];
  // ---- all loads first ---------------------------------------------------------------------------------------------------
  f32x4 tc[16];
  {
    const float* src = wpart + (which ? 0 : d) + (c_ok ? c4 : 0);
#pragma unroll
    for (int u = 0; u < 16; ++u) tc[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)min(u, row_splits - 1) * 2 * d);
  }
  float tg[16], tk[24];
#pragma unroll
  for (int u = 0; u < 16; ++u) tg[u] = g[min(tid + 256 * u, n - 1)];
  {
    const float* src = (which ? dkraw : part1) + hh;
#pragma unroll
    for (int u = 0; u < 24; ++u) tk[u] = src[(int64_t)min(u, KS - 1) * hd];
  }
  const float bkv = load_elem(bk + hh), bqv = load_elem(bq + hh);
  // ---- column factor: the row-chunk partials in rs order (wcolsum_finish_kernel) -------------------------------------------
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if (u < row_splits) bv += tc[u];
  for (int r0 = 16; r0 < row_splits; r0 += 16) {
    const float* src = wpart + (which ? 0 : d) + (c_ok ? c4 : 0);
#pragma unroll
    for (int u = 0; u < 16; ++u) tc[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)min(r0 + u, row_splits - 1) * 2 * d);
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (r0 + u < row_splits) bv += tc[u];
  }
  if (by == 0 && c_ok) {
    if (which) {
      *reinterpret_cast<f32x4*>(xsum + c4) = bv;
      const float nf = (float)n;
      f32x4 xb = {bv[0] / nf, bv[1] / nf, bv[2] / nf, bv[3] / nf};
      *reinterpret_cast<f32x4*>(xbar + c4) = xb;
    } else {
      *reinterpret_cast<f32x4*>(gx + c4) = bv;
    }
  }
  // ---- sum_i g_i (wcolsum_finish_kernel's order) ----------------------------------------------------------------------------
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if (tid + 256 * u < n) acc += tg[u];
  for (int i = tid + 4096; i < n; i += 256) acc += g[i];
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  const float s = (red[0] + red[1]) + (red[2] + red[3]);
  if (blockIdx.x == 0 && by == 0 && which == 0 && tid == 0) sg_out[0] = s;
  // ---- row factor of this thread's row ------------------------------------------------------------------------------------
  if (tid < 64) {
    float lf = 0.f;
    if (which == 0) {
      float v = 0.f;                                            // kbar_finish_kernel: ks order, 16 per batch
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (u < KS) v += tk[u];
      for (int k0 = 16; k0 < KS; k0 += 16) {
        float t2[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t2[u] = part1[(int64_t)min(k0 + u, KS - 1) * hd + hh];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (k0 + u < KS) v += t2[u];
      }
      v += bkv;
      const float kb = v * rs;
      lf = kb;
      if (blockIdx.x == 0 && r_ok) {
        kbar[hh] = v;
        a[hh] = kb;
        dbq[hh] = kb * s;
      }
    } else {
      float vd = 0.f;                                           // dk_finish_kernel: strided_sum's order (24 per batch)
#pragma unroll
      for (int u = 0; u < 24; ++u)
        if (u < KS) vd += tk[u];
      for (int k0 = 24; k0 < KS; k0 += 24) {
        float t2[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) t2[u] = dkraw[(int64_t)min(k0 + u, KS - 1) * hd + hh];
#pragma unroll
        for (int u = 0; u < 24; ++u)
          if (k0 + u < KS) vd += t2[u];
      }
      const float dkv = __builtin_fmaf(bqv, s, vd) * rs / (float)n;
      lf = dkv;
      if (blockIdx.x == 0 && r_ok) {
        dk[hh] = dkv;
        dbk[hh] = dkv * (float)n;
      }
    }
    left[tid] = lf;
  }
  __syncthreads();
  // ---- the workgroup's rows of the gradient ----------------------------------------------------------------------------------
  if (!c_ok) return;
  float* out = which ? dwk : dwq;
  for (int i = 0, r = by; r < hd; ++i, r += kOuterRowGroups) {
    const float av = left[i];
    f32x4 o = {av * bv[0], av * bv[1], av * bv[2], av * bv[3]};
    *reinterpret_cast<f32x4*>(out + (int64_t)r * d + c4) = o;
  }
}

// Rank-1-factor exchange (vsel_lis_factors_to_grads): R payload rows a_i | gx_i | dk_i | xsum_i | dbq_i | dbk_i ->
//   dwq[r][c] = scale sum_i a_i[r] gx_i[c],  dwk[r][c] = scale sum_i dk_i[r] xsum_i[c]   (i in order; fp32)
// grid (ceil(cols / 1024), row blocks, 2 = {dwq, dwk}); a block stages its <= 64 x R left factors through LDS.
__global__ __launch_bounds__(256) void outer_sum_kernel(const float* __restrict__ payload, int R, int64_t row_stride, int hd, int d,
                                                        float scale, float* __restrict__ dwq, float* __restrict__ dwk,
                                                        float* __restrict__ dbq, float* __restrict__ dbk) {
  const int which = blockIdx.z;
  // the bias gradients ride along (they were a launch of their own): the column-tile-0 workgroups sum dbq (z = 0) / dbk (z = 1) of
  // their rows over the payload rows in order -- bias_sum_kernel's arithmetic
  if (blockIdx.x == 0 && dbq) {
    for (int r = blockIdx.y + (int)gridDim.y * (int)threadIdx.x; r < hd; r += (int)gridDim.y * 256) {
      const float* pb = payload + 2 * (hd + d) + (which ? hd : 0) + r;
      float sb = 0.f;
      for (int i = 0; i < R; ++i) sb += pb[(int64_t)i * row_stride];
      (which ? dbk : dbq)[r] = sb * scale;
    }
  }
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
    if (n <= 1024) VSEL_LAUNCH(train_tail_kernel<4>, dim3(2), dim3(256), 0, st, scores, (int)n, (int)k, ps, ts, y);
    else VSEL_LAUNCH(train_tail_kernel<16>, dim3(2), dim3(256), 0, st, scores, (int)n, (int)k, ps, ts, y);
    VSEL_AFTER_LAUNCH(st, "train_tail_kernel");
    VSEL_LAUNCH((mask_apply_kernel<T, true>), dim3((unsigned)std::min<int64_t>(cdiv(n, 4), 4096) + 1), dim3(256), 0, st, h, ps,
                       (int)n, d, h_new, y, bce);
    VSEL_AFTER_LAUNCH(st, "mask_apply_kernel");
    return VSEL_OK;
  }
  rc = launch_soft_topk_fwd(st, scores, 1, n, k, ps, ts);
  if (rc) return rc;
  rc = launch_select(st, scores, &seg, nullptr, y);
  if (rc) return rc;
  VSEL_LAUNCH((mask_apply_kernel<T, false>), dim3((unsigned)std::min<int64_t>(cdiv(n, 4), 4096)), dim3(256), 0, st, h, ps,
                     (int)n, d, h_new, nullptr, nullptr);
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
      if (int rc = launch_wcolsum<T, true>(st, h, fuse_dps, (int)n, d, tp.wsplits, wpart, fuse_scores, fuse_ts, gbuf)) return rc;
      VSEL_LAUNCH(proj_nt_small_pair_kernel, dim3((unsigned)cdiv(hd, 32), p.ks1, 2), dim3(64), 0, st, wpart, wpart + d, make_view(&seg1), 1,
                  tp.wsplits, (const uint16_t*)sc->wk, (const uint16_t*)sc->wq, hd, d, p.kslice1, part1, dkraw, (int64_t)2 * d, 1, 0);
      VSEL_AFTER_LAUNCH(st, "proj_nt_small_pair_kernel");
      if (!factors && hd <= 64 * kOuterRowGroups) {
        VSEL_LAUNCH((outer_finish_pair_kernel<TW>), dim3((unsigned)cdiv(d, 1024), kOuterRowGroups, 2), dim3(256), 0, st, wpart, gbuf, (int)n, d,
                    tp.wsplits, part1, dkraw, p.ks1, hd, (const TW*)sc->bk, (const TW*)sc->bq, rs, xsum, gx, xbar, sg, kbar, dk, a, dbq, dbk,
                    dwq, dwk);
        VSEL_AFTER_LAUNCH(st, "outer_finish_pair_kernel");
      } else {
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
    }
    g = (const float*)(ws + tp.off_g);               // (dh below)
  } else {
  if (int rc = launch_wcolsum<T, false>(st, h, g, (int)n, d, tp.wsplits, wpart, nullptr, nullptr, nullptr)) return rc;
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
  VSEL_LAUNCH(outer_sum_kernel, og, dim3(256), 0, st, payload, (int)n_rows, row, (int)hd, (int)d, scale, dwq, dwk, dbq, dbk);
  VSEL_AFTER_LAUNCH(st, "outer_sum_kernel");
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
