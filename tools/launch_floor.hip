// What does ONE dependent phase cost on MI355X?  Chains of small kernels where every workgroup reads what the previous kernel
// wrote (all-to-all: block b reads the 16-byte records of all blocks) and writes its own record -- the dependency shape of the
// LIS projection stages at one image.  Build: hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o tools/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void empty_kernel() {}

// every block: sum `nread` 16-byte records of `in` (one per lane, strided over blocks), write one record
__global__ __launch_bounds__(256) void hop_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int nread) {
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nread; i += 256) a += in[i];
  __shared__ f32x4 red[256];
  red[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 256; i += 64) t += red[i];
    out[blockIdx.x] = t * 0.5f;
  }
}

// the same chain inside ONE launch: phases separated by a grid barrier (one agent-scope counter, monotonic; every workgroup
// adds 1 with release and polls with acquire).  PRE > 0: every phase also has PRE independent 16-byte loads per thread (the
// weights of the LIS projections) that a persistent kernel may issue BEFORE the barrier
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, int* hung) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1 << 22)) { *hung = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int PRE>
__global__ __launch_bounds__(256) void persistent_kernel(f32x4* __restrict__ a, f32x4* __restrict__ b, int nread, int chain,
                                                         unsigned* counter, unsigned base, const f32x4* __restrict__ wts,
                                                         int* hung) {
  __shared__ f32x4 red[256];
  for (int c = 0; c < chain; ++c) {
    const f32x4* in = (c & 1) ? b : a;
    f32x4* out = (c & 1) ? a : b;
    f32x4 pre[PRE > 0 ? PRE : 1];
#pragma unroll
    for (int q = 0; q < PRE; ++q) pre[q] = wts[((size_t)(c * gridDim.x + blockIdx.x) * PRE + q) * 256 + threadIdx.x];
    if (c > 0) grid_barrier(counter, base + (unsigned)c * gridDim.x, hung);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nread; i += 256) acc += __builtin_nontemporal_load(&in[i]);
#pragma unroll
    for (int q = 0; q < PRE; ++q) acc += pre[q];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < 256; i += 64) t += red[i];
      out[blockIdx.x] = t * 0.5f;
    }
  }
}

template <int PRE>
__global__ __launch_bounds__(256) void hop_pre_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int nread,
                                                      const f32x4* __restrict__ wts, int c) {
  f32x4 pre[PRE > 0 ? PRE : 1];
#pragma unroll
  for (int q = 0; q < PRE; ++q) pre[q] = wts[((size_t)(c * gridDim.x + blockIdx.x) * PRE + q) * 256 + threadIdx.x];
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nread; i += 256) a += in[i];
#pragma unroll
  for (int q = 0; q < PRE; ++q) a += pre[q];
  __shared__ f32x4 red[256];
  red[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 256; i += 64) t += red[i];
    out[blockIdx.x] = t * 0.5f;
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f32x4 *a, *b; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
  const int reps = 200;
  for (int chain : {1, 5, 9}) {
    for (int grid : {1, 128, 256, 512}) {
      auto run = [&](bool hop, int nread) {
        for (int w = 0; w < 20; ++w) for (int c = 0; c < chain; ++c) {
          if (hop) hipLaunchKernelGGL(hop_kernel, dim3(grid), dim3(256), 0, st, (c & 1) ? b : a, (c & 1) ? a : b, nread);
          else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st);
        }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) for (int c = 0; c < chain; ++c) {
          if (hop) hipLaunchKernelGGL(hop_kernel, dim3(grid), dim3(256), 0, st, (c & 1) ? b : a, (c & 1) ? a : b, nread);
          else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / reps;
      };
      printf("chain %d x grid %3d: empty %6.2f us/chain | hop (read %d records) %6.2f | hop (read 4096 records = 64 KB) %6.2f\n", chain, grid,
             run(false, 0), grid, run(true, grid), run(true, 4096));
    }
  }
  // persistent form
  unsigned* counter; CK(hipMalloc(&counter, 64)); CK(hipMemset(counter, 0, 64));
  int* hung; CK(hipMalloc(&hung, 4)); CK(hipMemset(hung, 0, 4));
  constexpr int PRE = 8;
  f32x4* wts; CK(hipMalloc(&wts, (size_t)9 * 256 * PRE * 256 * 16)); CK(hipMemset(wts, 0, (size_t)9 * 256 * PRE * 256 * 16));
  unsigned base = 0;
  for (int chain : {1, 5, 9}) {
    for (int grid : {128, 256}) {
      auto run = [&](int mode, int nread) {      // 0: persistent, 1: persistent + PRE loads, 2: launches + PRE loads
        auto once = [&]() {
          if (mode == 0) {
            hipLaunchKernelGGL(persistent_kernel<0>, dim3(grid), dim3(256), 0, st, a, b, nread, chain, counter, base, wts, hung);
            base += (unsigned)(chain - 1) * grid;
          } else if (mode == 1) {
            hipLaunchKernelGGL(persistent_kernel<PRE>, dim3(grid), dim3(256), 0, st, a, b, nread, chain, counter, base, wts, hung);
            base += (unsigned)(chain - 1) * grid;
          } else {
            for (int c = 0; c < chain; ++c)
              hipLaunchKernelGGL(hop_pre_kernel<PRE>, dim3(grid), dim3(256), 0, st, (c & 1) ? b : a, (c & 1) ? a : b, nread, wts, c);
          }
        };
        for (int w = 0; w < 20; ++w) once();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) once();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / reps;
      };
      printf("chain %d x grid %3d, read 4096 records: ONE persistent launch %6.2f us/chain | + %d weight loads per thread and phase: persistent %6.2f, launches %6.2f\n",
             chain, grid, run(0, 4096), PRE, run(1, 4096), run(2, 4096));
    }
  }
  int h_hung = 0; CK(hipMemcpy(&h_hung, hung, 4, hipMemcpyDeviceToHost));
  printf("barrier spin limit hit: %d\n", h_hung);
  return 0;
}
