// Differentiable top-k (reference: qwen-vl-finetune/compression_method/selector_model.py:53-88).
//
// forward  = _find_ts (:72-86): 64 fixed bisection steps for t with sum_i sigmoid(x_i + t) = k,
//            lo = -max(x) - 10, hi = -min(x) + 10; returns t and ps = sigmoid(x + t).
// backward = TopK.backward (:60-70): v = sigmoid'(x + t), grad = g*v - (sum g*v) * v / sum v.
//
// The reference issues ~320 tiny launches for the forward; here one workgroup per row keeps lo/hi in
// registers and does the 64 reductions with wave shuffles + one LDS hop.  All arithmetic is fp32 (the
// reference's bf16 run stalls after ~10 steps at bf16 spacing -- SURVEY.md section 7 hard part 5).
#include "softtopk.h"

namespace vsel {

// EPT > 0: the row lives in registers (EPT elements per thread, n <= NT * EPT); EPT == 0: re-read from global (huge rows).
// Early exit: once `mid` equals `lo` or `hi` (adjacent floats) the remaining iterations of the reference's fixed 64-step
// loop cannot change (lo, hi) any more, so stopping there returns the same ts bit for bit.
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void soft_topk_fwd_kernel(const float* __restrict__ xs, int n, int k,
                                                           float* __restrict__ ps, float* __restrict__ ts) {
  constexpr int NW = NT / 64;
  constexpr int E = EPT > 0 ? EPT : 1;
  __shared__ float red[6][NW];
  const int row = blockIdx.x;
  const float* x = xs + (int64_t)row * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  float xr[E];
  float mx = -INFINITY, mn = INFINITY;
  if constexpr (EPT > 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = tid + e * NT;
      const bool ok = i < n;
      const float v = ok ? x[i] : 0.f;
      xr[e] = ok ? v : -INFINITY;            // sigmoid(-inf + t) = 0: padding never contributes
      if (ok) { mx = fmaxf(mx, v); mn = fminf(mn, v); }
    }
  } else {
    for (int i = tid; i < n; i += NT) {
      const float v = x[i];
      mx = fmaxf(mx, v);
      mn = fminf(mn, v);
    }
  }
  mx = wave_max(mx);
  mn = wave_min(mn);
  if (lane == 0) { red[4][wave] = mx; red[5][wave] = mn; }
  __syncthreads();
  mx = red[4][0]; mn = red[5][0];
#pragma unroll
  for (int w = 1; w < NW; ++w) { mx = fmaxf(mx, red[4][w]); mn = fminf(mn, red[5][w]); }

  float lo = -mx - 10.0f;   // :78
  float hi = -mn + 10.0f;   // :79
  const float kf = (float)k;
  // A step only needs the SIGN of (sum - k).  While the bracket is wide that sign is decided by a sum of 4-instruction sigmoids
  // (error bound: 2e-6 per element incl. the summation, + 1e-3); the step is redone with the reference-accurate sigmoid when
  // the cheap sum is within the bound of k -- the sequence of (lo, hi) is the all-accurate one, step for step, at a third of the
  // arithmetic (the row sits on ONE CU: 26 steps x N accurate sigmoids were 31 us at N = 2304).
  const float decisive = 2e-6f * (float)n + 1e-3f;
  for (int it = 0; it < 64; ++it) {             // :80
    const float mid = (hi + lo) / 2.0f;         // :81
    float acc = 0.f;
    if constexpr (EPT > 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) acc += sigmoidf_fast(xr[e] + mid);
    } else {
      for (int i = tid; i < n; i += NT) acc += sigmoidf_fast(x[i] + mid);
    }
    float sum = block_sum<NW>(acc, red, (2 * it) & 3);
    if (fabsf(sum - kf) <= decisive) {          // uniform
      acc = 0.f;
      if constexpr (EPT > 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc += sigmoidf_ref(xr[e] + mid);
      } else {
        for (int i = tid; i < n; i += NT) acc += sigmoidf_ref(x[i] + mid);
      }
      sum = block_sum<NW>(acc, red, (2 * it + 1) & 3);
    }
    const bool fixed_point = (mid == lo) || (mid == hi);
    if (sum < kf) lo = mid; else hi = mid;      // :82-84
    if (fixed_point) break;
  }
  const float t = (lo + hi) / 2.0f;             // :85
  if (tid == 0) ts[row] = t;
  if constexpr (EPT > 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = tid + e * NT;
      if (i < n) ps[(int64_t)row * n + i] = sigmoidf_ref(xr[e] + t);   // :86
    }
  } else {
    for (int i = tid; i < n; i += NT) ps[(int64_t)row * n + i] = sigmoidf_ref(x[i] + t);
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void soft_topk_bwd_kernel(const float* __restrict__ grad_ps,
                                                           const float* __restrict__ xs,
                                                           const float* __restrict__ ts, int n,
                                                           float* __restrict__ grad_xs) {
  constexpr int NW = NT / 64;
  __shared__ float red[2][NW];
  const int row = blockIdx.x;
  const float* x = xs + (int64_t)row * n;
  const float* g = grad_ps + (int64_t)row * n;
  const float t = ts[row];
  const int tid = threadIdx.x;
  float sv = 0.f, suv = 0.f;
  for (int i = tid; i < n; i += NT) {
    const float p = sigmoidf_ref(x[i] + t);
    const float v = p * (1.0f - p);             // :66  sigmoid'(x + t)
    sv += v;
    suv += g[i] * v;
  }
  sv = block_sum<NW>(sv, red, 0);               // :67
  suv = block_sum<NW>(suv, red, 1);             // :70 uv.sum()
  for (int i = tid; i < n; i += NT) {
    const float p = sigmoidf_ref(x[i] + t);
    const float v = p * (1.0f - p);
    const float uv = g[i] * v;                  // :69
    grad_xs[(int64_t)row * n + i] = (-suv * v) / sv + uv;   // :70-71
  }
}

int launch_soft_topk_fwd(hipStream_t st, const float* xs, int64_t b, int64_t n, int64_t k, float* ps, float* ts) {
  if (n <= 1024)
    hipLaunchKernelGGL((soft_topk_fwd_kernel<256, 4>), dim3((unsigned)b), dim3(256), 0, st, xs, (int)n, (int)k, ps, ts);
  else if (n <= 4096)      // one wave per SIMD: a step is issue-bound (16 waves: 0.8 us per step at N = 2304, 4 waves: see bench_train)
    hipLaunchKernelGGL((soft_topk_fwd_kernel<256, 16>), dim3((unsigned)b), dim3(256), 0, st, xs, (int)n, (int)k, ps, ts);
  else if (n <= 16384)
    hipLaunchKernelGGL((soft_topk_fwd_kernel<1024, 16>), dim3((unsigned)b), dim3(1024), 0, st, xs, (int)n, (int)k, ps, ts);
  else
    hipLaunchKernelGGL((soft_topk_fwd_kernel<1024, 0>), dim3((unsigned)b), dim3(1024), 0, st, xs, (int)n, (int)k, ps, ts);
  VSEL_AFTER_LAUNCH(st, "soft_topk_fwd_kernel");
  return VSEL_OK;
}

int launch_soft_topk_bwd(hipStream_t st, const float* g, const float* xs, const float* ts, int64_t b, int64_t n,
                         float* gx) {
  if (n <= 4096)
    hipLaunchKernelGGL((soft_topk_bwd_kernel<256>), dim3((unsigned)b), dim3(256), 0, st, g, xs, ts, (int)n, gx);
  else
    hipLaunchKernelGGL((soft_topk_bwd_kernel<1024>), dim3((unsigned)b), dim3(1024), 0, st, g, xs, ts, (int)n, gx);
  VSEL_AFTER_LAUNCH(st, "soft_topk_bwd_kernel");
  return VSEL_OK;
}

}  // namespace vsel

using namespace vsel;

extern "C" int vsel_soft_topk_fwd(void* stream, const float* xs, int64_t b, int64_t n, int64_t k, float* ps,
                                  float* ts) {
  if (!xs || !ps || !ts) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (b < 1 || n < 1 || b > 0x7fffffff || n > 0x7fffffff) return fail(VSEL_ERR_INVALID, "bad shape [%lld, %lld]", (long long)b, (long long)n);
  // the reference asserts 0 < k < n (selector_model.py:75)
  if (!(0 < k && k < n)) return fail(VSEL_ERR_INVALID, "soft top-k needs 0 < k < n (k=%lld, n=%lld)", (long long)k, (long long)n);
  VSEL_PROF_BEGIN(stream);
  return launch_soft_topk_fwd((hipStream_t)stream, xs, b, n, k, ps, ts);
}

extern "C" int vsel_soft_topk_bwd(void* stream, const float* grad_ps, const float* xs, const float* ts, int64_t b,
                                  int64_t n, float* grad_xs) {
  if (!grad_ps || !xs || !ts || !grad_xs) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (b < 1 || n < 1 || b > 0x7fffffff || n > 0x7fffffff) return fail(VSEL_ERR_INVALID, "bad shape [%lld, %lld]", (long long)b, (long long)n);
  VSEL_PROF_BEGIN(stream);
  return launch_soft_topk_bwd((hipStream_t)stream, grad_ps, xs, ts, b, n, grad_xs);
}
