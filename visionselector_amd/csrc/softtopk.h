// Pieces shared by the differentiable top-k kernels (softtopk.hip) and the fused training forward tail (train.hip).
#pragma once
#include "common.h"

namespace vsel {

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }
// 4 instructions instead of ~25 (v_exp_f32 / v_rcp_f32: about 1 ulp each); relative error < 1e-6.  Only used to DECIDE a bisection
// step whose sum is further from k than that error allows (below); every value that is returned comes from sigmoidf_ref.
__device__ __forceinline__ float sigmoidf_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// Sum over the block, identical value in every thread, fixed order.  `slot` alternates so one barrier
// per call is enough.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float (*red)[NW], int slot) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[slot][wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) t += red[slot][w];
  return t;
}

// ---- TopK.backward (FT/compression_method/selector_model.py:60-70), shared by soft_topk_bwd_kernel and the training backward's fused
// sweep (train.hip): the two block sums over one row by NT threads in a fixed order, and one element of the gradient.
// Contraction is OFF inside both and every fused multiply-add is written out: hipcc vectorises the standalone kernel's loops (packed
// fp32 pairs) and not the fused sweep's, and left to itself it contracts a * b + c in one and not in the other -- the two forms then
// differ in the last bit of g, i.e. in all of dbq = kbar rs sum_i g_i (a sum that is 0 in exact arithmetic).
template <int NT>
__device__ __forceinline__ void soft_topk_bwd_sums(const float* __restrict__ g, const float* __restrict__ x, float t, int n,
                                                   float (*red)[NT / 64], float& sv_out, float& suv_out) {
#pragma clang fp contract(off)
  float sv = 0.f, suv = 0.f;
  for (int i = threadIdx.x; i < n; i += NT) {
    const float p = sigmoidf_ref(x[i] + t);
    const float v = p * (1.0f - p);             // :66  sigmoid'(x + t)
    sv += v;
    suv = __builtin_fmaf(g[i], v, suv);
  }
  sv_out = block_sum<NT / 64>(sv, red, 0);      // :67
  suv_out = block_sum<NT / 64>(suv, red, 1);    // :70 uv.sum()
}
__device__ __forceinline__ float soft_topk_bwd_elem(float gi, float xi, float t, float sv, float suv) {
#pragma clang fp contract(off)
  const float p = sigmoidf_ref(xi + t);
  const float v = p * (1.0f - p);
  const float q = (-suv * v) / sv;              // :70
  return __builtin_fmaf(gi, v, q);              // :69, :71  uv - uv.sum() v / v.sum()
}

// ---- the threshold of _find_ts (selector_model.py:72-86) -------------------------------------------------------------------
// The reference bisects 64 times for the t with sum_i sigmoid(x_i + t) = k, from lo = -max(x) - 10, hi = -min(x) + 10.  Bisection
// costs one block-wide reduction per bit of t (26-27 dependent reductions until lo and hi are adjacent floats; 24 us for one row
// of 2304 scores on one CU -- longer than the whole LIS + splice of that image).  The SAME root is found here by a bracketed Newton
// iteration on the reference's own bracket: f(t) = sum sigmoid(x + t) is increasing with f' = sum sigmoid (1 - sigmoid) for free,
// the start t0 = logit(k / n) - mean(x) is exact when the scores' spread is small against the sigmoid's width (the scorer's are:
// std 0.05-0.2), and a step that leaves (lo, hi) is replaced by the bracket's midpoint, so convergence is unconditional.  3-5
// reductions instead of 27.  The result is the root to fp32 summation noise: |t - reference t| ~ 1e-7 .. 1e-6 (tests: 2e-5),
// |ps - reference ps| <= 1e-6 (tests: 1e-5), sum(ps) = k to ~1e-4.  Deterministic: fixed thread -> element map, fixed order.
//
// xr[E]: the caller's register-resident elements (padding = -INFINITY: sigmoid(-inf + t) = 0), sx = this thread's sum of its valid
// elements, mx / mn = the row's maximum / minimum (block-uniform).  NW = waves whose partials are added (waves >= NW of a
// larger workgroup must pass all-padding rows and still call: the barriers are workgroup-wide).  red: float[6][NW_BLOCK].
template <int NW, int NWB, int E>
__device__ __forceinline__ float find_ts_newton(const float (&xr)[E], float sx, float mx, float mn, int n, int k, float (*red)[NWB]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto block_sum2 = [&](float a, float b, int slot, float& ta, float& tb) {
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { red[slot][wave] = a; red[slot + 1][wave] = b; }
    __syncthreads();
    ta = 0.f; tb = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { ta += red[slot][w]; tb += red[slot + 1][w]; }
  };
  float lo = -mx - 10.0f;   // :78
  float hi = -mn + 10.0f;   // :79
  const float kf = (float)k, nf = (float)n;
  float xsum, unused;
  block_sum2(sx, 0.f, 0, xsum, unused);
  float t = logf(kf / (nf - kf)) - xsum / nf;
  t = fminf(fmaxf(t, lo), hi);
  // Far from the root a step only has to point the right way: the 4-instruction sigmoid (relative error < 1e-6) drives the
  // iteration until a step is shorter than 1e-4, the reference-accurate one (expf + division, ~30 instructions) finishes it --
  // the returned t is a fixed point of the ACCURATE iteration; the switch is block-uniform (every thread holds the same st, dt).
  bool accurate = false;
  for (int it = 0; it < 64; ++it) {
    float s = 0.f, d = 0.f;
    if (accurate) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float p = sigmoidf_ref(xr[e] + t);
        s += p;
        d = fmaf(p, 1.0f - p, d);
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float p = sigmoidf_fast(xr[e] + t);
        s += p;
        d = fmaf(p, 1.0f - p, d);
      }
    }
    float st, dt;
    block_sum2(s, d, 2 + 2 * (it & 1), st, dt);      // slots alternate: one barrier per step is enough
    float tn = t - (st - kf) / fmaxf(dt, 1e-30f);
    // the bracket of :82-84.  Every ACCURATE evaluation updates it; a fast one (relative error < 1e-6 per term, < 1e-6 n on the sum)
    // only when the sum is further from k than that error can explain -- so the root never leaves (lo, hi), and rows whose sigmoids are
    // saturated or whose scores are spread wide (f' ~ 0: the Newton step shoots out of the bracket) fall back to a true bisection
    // instead of sitting on a clamped end until the iteration cap.
    if (accurate || fabsf(st - kf) > 8e-6f * nf) {
      if (st < kf) lo = t; else hi = t;
    }
    // (non-strict: st == k exactly gives tn == t == the bracket end just set -- that is convergence, not an escape)
    if (!(tn >= lo && tn <= hi)) tn = 0.5f * (lo + hi);
    const float step = fabsf(tn - t);
    // converged: a Newton step from an accurate evaluation no longer moves t by more than ~8 ulps.  (Not 1 ulp: the fp32 sum of n
    // sigmoids carries ~1e-4 of rounding noise, i.e. steps of ~3e-7 for ever; quadratic convergence means the step BEFORE a
    // 1e-6 step was ~1e-3, so the t returned is the root to that noise -- |dt| ~ 3e-7, |dps| <= 1e-7.)
    const bool done = accurate && step <= 1e-6f * fmaxf(1.0f, fabsf(t));
    if (!accurate && (step <= 1e-4f || it >= 12)) accurate = true;
    t = tn;
    if (done) break;
  }
  return t;
}

// One row of n <= 256 * 16 scores: element i lives in thread i % 256 (register i / 256) of the FIRST FOUR waves of the workgroup;
// a larger workgroup (the extra block of select_splice_small_kernel, 1024 threads) runs the same arithmetic with idle waves.
// E = registers per thread (4 / 8 / 12 / 16, the smallest that holds the row): padding contributes exact zeros to every sum, so
// the result does not depend on E.
template <int NWB, int E>
__device__ __forceinline__ void soft_topk_row_256(const float* __restrict__ x, int n, int k, float* __restrict__ ps,
                                                  float* __restrict__ ts, float (*red)[NWB]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float xr[E];
  float mx = -INFINITY, mn = INFINITY, sx = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = tid + e * 256;
    const bool ok = tid < 256 && i < n;
    const float v = ok ? x[i] : 0.f;
    xr[e] = ok ? v : -INFINITY;              // sigmoid(-inf + t) = 0: padding never contributes
    if (ok) { mx = fmaxf(mx, v); mn = fminf(mn, v); sx += v; }
  }
  mx = wave_max(mx);
  mn = wave_min(mn);
  if (lane == 0) { red[4][wave] = mx; red[5][wave] = mn; }
  __syncthreads();
  mx = red[4][0]; mn = red[5][0];
#pragma unroll
  for (int w = 1; w < 4; ++w) { mx = fmaxf(mx, red[4][w]); mn = fminf(mn, red[5][w]); }
  __syncthreads();                           // red[4..5] are reused by the iteration's alternating slots
  const float t = find_ts_newton<4, NWB, E>(xr, sx, mx, mn, n, k, red);
  if (tid == 0) ts[0] = t;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = tid + e * 256;
    if (tid < 256 && i < n) ps[i] = sigmoidf_ref(xr[e] + t);   // :86
  }
}
template <int NWB>
__device__ __forceinline__ void soft_topk_row_256x16(const float* __restrict__ x, int n, int k, float* __restrict__ ps,
                                                     float* __restrict__ ts, float (*red)[NWB]) {
  if (n <= 1024) soft_topk_row_256<NWB, 4>(x, n, k, ps, ts, red);            // block-uniform
  else if (n <= 2048) soft_topk_row_256<NWB, 8>(x, n, k, ps, ts, red);
  else if (n <= 3072) soft_topk_row_256<NWB, 12>(x, n, k, ps, ts, red);
  else soft_topk_row_256<NWB, 16>(x, n, k, ps, ts, red);
}

}  // namespace vsel
