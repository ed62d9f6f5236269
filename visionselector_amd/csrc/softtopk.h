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

}  // namespace vsel
