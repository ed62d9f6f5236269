// Differentiable top-k (reference: qwen-vl-finetune/compression_method/selector_model.py:53-88).
//
// forward  = _find_ts (:72-86): 64 fixed bisection steps for t with sum_i sigmoid(x_i + t) = k,
//            lo = -max(x) - 10, hi = -min(x) + 10; returns t and ps = sigmoid(x + t).
// backward = TopK.backward (:60-70): v = sigmoid'(x + t), grad = g*v - (sum g*v) * v / sum v.
//
// The reference issues ~320 tiny launches for the forward; here one workgroup per row finds the same root with a bracketed Newton
// iteration (softtopk.h, find_ts_newton: 3-5 block reductions instead of the 27 of a bisection to float resolution).  All
// arithmetic is fp32 (the reference's bf16 run stalls after ~10 steps at bf16 spacing -- SURVEY.md section 7 hard part 5).
#include "softtopk.h"

namespace vsel {

// EPT > 0: the row lives in registers (EPT elements per thread, n <= NT * EPT); EPT == 0: re-read from global (huge rows).
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void soft_topk_fwd_kernel(const float* __restrict__ xs, int n, int k,
                                                           float* __restrict__ ps, float* __restrict__ ts) {
  constexpr int NW = NT / 64;
  __shared__ float red[6][NW];
  const int row = blockIdx.x;
  const float* x = xs + (int64_t)row * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (NT == 256 && EPT == 16) {
    soft_topk_row_256x16<NW>(x, n, k, ps + (int64_t)row * n, ts + row, red);
    return;
  }
  constexpr int E = EPT > 0 ? EPT : 1;
  float xr[E];
  float mx = -INFINITY, mn = INFINITY, sx = 0.f;
  if constexpr (EPT > 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = tid + e * NT;
      const bool ok = i < n;
      const float v = ok ? x[i] : 0.f;
      xr[e] = ok ? v : -INFINITY;
      if (ok) { mx = fmaxf(mx, v); mn = fminf(mn, v); sx += v; }
    }
  } else {
    for (int i = tid; i < n; i += NT) {
      const float v = x[i];
      mx = fmaxf(mx, v);
      mn = fminf(mn, v);
      sx += v;
    }
  }
  mx = wave_max(mx);
  mn = wave_min(mn);
  if (lane == 0) { red[4][wave] = mx; red[5][wave] = mn; }
  __syncthreads();
  mx = red[4][0]; mn = red[5][0];
#pragma unroll
  for (int w = 1; w < NW; ++w) { mx = fmaxf(mx, red[4][w]); mn = fminf(mn, red[5][w]); }
  __syncthreads();
  float t;
  if constexpr (EPT > 0) {
    t = find_ts_newton<NW, NW, E>(xr, sx, mx, mn, n, k, red);
  } else {
    // huge rows (n > 16384): the same iteration with the elements re-read from global (L2-resident: 64 KB+)
    float lo = -mx - 10.0f, hi = -mn + 10.0f;
    const float kf = (float)k, nf = (float)n;
    const float xsum = block_sum<NW>(sx, red, 0);
    t = fminf(fmaxf(logf(kf / (nf - kf)) - xsum / nf, lo), hi);
    for (int it = 0; it < 48; ++it) {
      float s = 0.f, d = 0.f;
      for (int i = tid; i < n; i += NT) {
        const float p = sigmoidf_ref(x[i] + t);
        s += p;
        d = fmaf(p, 1.0f - p, d);
      }
      const float st = block_sum<NW>(s, red, 1 + 2 * (it & 1));
      const float dt = block_sum<NW>(d, red, 2 + 2 * (it & 1));
      if (st < kf) lo = t; else hi = t;
      float tn = t - (st - kf) / fmaxf(dt, 1e-30f);
      if (!(tn >= lo && tn <= hi)) tn = 0.5f * (lo + hi);
      const bool done = fabsf(tn - t) <= 1e-6f * fmaxf(1.0f, fabsf(t));
      t = tn;
      if (done) break;
    }
  }
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

// The reference's OWN arithmetic when its scorer runs in bfloat16 (the released checkpoints do): _find_ts on a bf16 tensor rounds every
// operation to bf16 -- lo / hi / mid, x + mid, the sigmoid, and the SUM (fp32 accumulation, bf16 result: spacing 2 at 256 .. 512) -- so
// the bisection stalls after ~10 of its 64 steps on the bf16 neighbour of the root (EV/token_compression/selector_model.py:75-89; e.g.
// sum ps = 459.35 for k = 460).  This kernel restates exactly that, rounding step for rounding step (oracle/lis.py,
// find_ts_bf16_reference, pinned bit for bit on the reference's bf16 fixtures); the default entry returns the fp32 root instead.
// One workgroup per row; x rounded to bf16 on entry; the loop stops when a step leaves lo and hi unchanged (every later step repeats it).
__device__ __forceinline__ float round_bf16(float x) { return bf16_to_f32(f32_to_bf16_bits(x)); }

template <int NT>
__global__ __launch_bounds__(NT) void soft_topk_fwd_bf16ref_kernel(const float* __restrict__ xs, int n, int k, float* __restrict__ ps,
                                                                   float* __restrict__ ts) {
  constexpr int NW = NT / 64;
  __shared__ float red[6][NW];
  const int row = blockIdx.x;
  const float* x = xs + (int64_t)row * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY, mn = INFINITY;
  for (int i = tid; i < n; i += NT) {
    const float v = round_bf16(x[i]);
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
  }
  mx = wave_max(mx);
  mn = wave_min(mn);
  if (lane == 0) { red[4][wave] = mx; red[5][wave] = mn; }
  __syncthreads();
  mx = red[4][0]; mn = red[5][0];
#pragma unroll
  for (int w = 1; w < NW; ++w) { mx = fmaxf(mx, red[4][w]); mn = fminf(mn, red[5][w]); }
  __syncthreads();
  float lo = round_bf16(round_bf16(-mx) - 10.0f), hi = round_bf16(round_bf16(-mn) + 10.0f);     // :80-81
  const float kf = (float)k;
  for (int it = 0; it < 64; ++it) {                                                              // :82
    const float mid = round_bf16(round_bf16(hi + lo) * 0.5f);                                    // :83
    float s = 0.f;
    for (int i = tid; i < n; i += NT) {
      float a = round_bf16(round_bf16(x[i]) + mid);
      asm volatile("" : "+v"(a));                                                                // (see the output loop below)
      s += round_bf16(sigmoidf_ref(a));
    }
    const float st = round_bf16(block_sum<NW>(s, red, it & 1 ? 1 : 0));                          // :84 (.sum of a bf16 tensor: fp32 inside, bf16 out)
    const float lo2 = st < kf ? mid : lo, hi2 = st < kf ? hi : mid;                              // :85-86
    const int same = __builtin_amdgcn_readfirstlane((lo2 == lo && hi2 == hi) ? 1 : 0);       // (the same value in every lane: a scalar branch)
    lo = lo2;
    hi = hi2;
    if (same) break;
  }
  const float t = round_bf16(round_bf16(lo + hi) * 0.5f);                                        // :87
  if (tid == 0) ts[row] = t;
  float* prow = ps + (int64_t)row * n;
  for (int i = tid; i < n; i += NT) {
    float a = round_bf16(round_bf16(x[i]) + t);
    // (hipcc, ROCm 7.2, unrolls this loop by two and packs the pair's "1 + exp(-a)" into v_pk_add_f32 -- the negation of the FIRST element
    // was lost on the way: every thread's first output came out as sigmoid(-a).  An opaque copy keeps the elements apart.)
    asm volatile("" : "+v"(a));
    prow[i] = round_bf16(sigmoidf_ref(a));                                                        // :88
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
  float sv, suv;
  soft_topk_bwd_sums<NT>(g, x, t, n, red, sv, suv);
  for (int i = tid; i < n; i += NT) grad_xs[(int64_t)row * n + i] = soft_topk_bwd_elem(g[i], x[i], t, sv, suv);
}

int launch_soft_topk_fwd(hipStream_t st, const float* xs, int64_t b, int64_t n, int64_t k, float* ps, float* ts) {
  if (n <= 4096)           // four waves, one per SIMD; the row routine picks 4 / 8 / 12 / 16 registers per thread from n
    VSEL_LAUNCH((soft_topk_fwd_kernel<256, 16>), dim3((unsigned)b), dim3(256), 0, st, xs, (int)n, (int)k, ps, ts);
  else if (n <= 16384)
    VSEL_LAUNCH((soft_topk_fwd_kernel<1024, 16>), dim3((unsigned)b), dim3(1024), 0, st, xs, (int)n, (int)k, ps, ts);
  else
    VSEL_LAUNCH((soft_topk_fwd_kernel<1024, 0>), dim3((unsigned)b), dim3(1024), 0, st, xs, (int)n, (int)k, ps, ts);
  VSEL_AFTER_LAUNCH(st, "soft_topk_fwd_kernel");
  return VSEL_OK;
}

int launch_soft_topk_bwd(hipStream_t st, const float* g, const float* xs, const float* ts, int64_t b, int64_t n,
                         float* gx) {
  if (n <= 4096)
    VSEL_LAUNCH((soft_topk_bwd_kernel<256>), dim3((unsigned)b), dim3(256), 0, st, g, xs, ts, (int)n, gx);
  else
    VSEL_LAUNCH((soft_topk_bwd_kernel<1024>), dim3((unsigned)b), dim3(1024), 0, st, g, xs, ts, (int)n, gx);
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

extern "C" int vsel_soft_topk_fwd_bf16ref(void* stream, const float* xs, int64_t b, int64_t n, int64_t k, float* ps, float* ts) {
  if (!xs || !ps || !ts) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (b < 1 || n < 1 || b > 0x7fffffff || n > 0x7fffffff) return fail(VSEL_ERR_INVALID, "bad shape [%lld, %lld]", (long long)b, (long long)n);
  if (!(0 < k && k < n)) return fail(VSEL_ERR_INVALID, "soft top-k needs 0 < k < n (k=%lld, n=%lld)", (long long)k, (long long)n);
  VSEL_PROF_BEGIN(stream);
  VSEL_LAUNCH((soft_topk_fwd_bf16ref_kernel<1024>), dim3((unsigned)b), dim3(1024), 0, (hipStream_t)stream, xs, (int)n, (int)k, ps, ts);
  VSEL_AFTER_LAUNCH((hipStream_t)stream, "soft_topk_fwd_bf16ref_kernel");
  return VSEL_OK;
}

extern "C" int vsel_soft_topk_bwd(void* stream, const float* grad_ps, const float* xs, const float* ts, int64_t b,
                                  int64_t n, float* grad_xs) {
  if (!grad_ps || !xs || !ts || !grad_xs) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (b < 1 || n < 1 || b > 0x7fffffff || n > 0x7fffffff) return fail(VSEL_ERR_INVALID, "bad shape [%lld, %lld]", (long long)b, (long long)n);
  VSEL_PROF_BEGIN(stream);
  return launch_soft_topk_bwd((hipStream_t)stream, grad_ps, xs, ts, b, n, grad_xs);
}
