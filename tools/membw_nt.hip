// Streaming-read bandwidth with default-policy vs non-temporal loads (global_load_dwordx4 ... nt) on gfx950, 2 GiB footprint.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membw_nt.hip -o tools/membw_nt ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ p, size_t n_vec, unsigned* sink) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  unsigned acc = 0;
  for (; i + 256 * (UNROLL - 1) < n_vec; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + 256 * u) : p[i + 256 * u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int U, bool NT>
static double run(const u32x4* a, size_t n_vec, unsigned* sink, int grid, hipEvent_t e0, hipEvent_t e1) {
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, a, n_vec, sink);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, a, n_vec, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)n_vec * 16 * 10 / (ms * 1e-3) / 1e9;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  u32x4* a; unsigned* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t n_vec = bytes / 16;
  printf("2 GiB streaming read, GB/s (grid blocks x 256 threads, unroll 4 / 8):\n");
  for (int grid : {2048, 4096, 8192, 16384}) {
    printf(" grid %5d: default u4 %6.0f u8 %6.0f | non-temporal u4 %6.0f u8 %6.0f\n", grid,
           run<4, false>(a, n_vec, sink, grid, e0, e1), run<8, false>(a, n_vec, sink, grid, e0, e1),
           run<4, true>(a, n_vec, sink, grid, e0, e1), run<8, true>(a, n_vec, sink, grid, e0, e1));
  }
  return 0;
}
