// Micro-benchmark: read bandwidth vs footprint (HBM vs Infinity Cache vs L2) and copy bandwidth on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o /tmp/membw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ p, size_t n_vec, unsigned* sink) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  unsigned acc = 0;
  for (; i + 256 * (UNROLL - 1) < n_vec; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = p[i + 256 * u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ p, u32x4* __restrict__ q, size_t n_vec) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (; i + 256 * (UNROLL - 1) < n_vec; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = p[i + 256 * u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) q[i + 256 * u] = v[u];
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main() {
  const size_t max_bytes = (size_t)4 << 30;
  u32x4 *a, *b; unsigned* sink;
  CK(hipMalloc(&a, max_bytes)); CK(hipMalloc(&b, max_bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, max_bytes)); CK(hipMemset(b, 2, max_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sizes_mb[] = {8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048, 4096};
  const int grids[] = {1024, 2048, 4096, 8192};
  printf("read  bandwidth (GB/s): rows = footprint MB, cols = grid blocks (unroll 8)\n");
  for (size_t mb : sizes_mb) {
    const size_t n_vec = (mb << 20) / 16;
    printf("%5zu MB:", mb);
    for (int g : grids) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(read_kernel<8>, dim3(g), dim3(256), 0, 0, a, n_vec, sink);
      const int reps = mb <= 256 ? 50 : 10;
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_kernel<8>, dim3(g), dim3(256), 0, 0, a, n_vec, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf(" %7.0f", (double)(mb << 20) * reps / (ms * 1e-3) / 1e9);
    }
    printf("\n");
  }
  printf("read unroll sweep at 2048 MB, grid 4096: ");
  {
    const size_t n_vec = ((size_t)2048 << 20) / 16; float ms;
#define RUN(U) CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(read_kernel<U>, dim3(4096), dim3(256), 0, 0, a, n_vec, sink); \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); printf(" u%d=%.0f", U, 2048.0 * 1048576 * 10 / (ms * 1e-3) / 1e9);
    RUN(1) RUN(2) RUN(4) RUN(8) RUN(16)
    printf("\n");
  }
  printf("copy bandwidth (GB/s, read+write bytes): ");
  for (size_t mb : {64, 128, 1024, 4096}) {
    const size_t n_vec = (mb << 20) / 16; float ms;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(copy_kernel<4>, dim3(4096), dim3(256), 0, 0, a, b, n_vec);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(copy_kernel<4>, dim3(4096), dim3(256), 0, 0, a, b, n_vec);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf(" %zuMB=%.0f", mb, 2.0 * (mb << 20) * 10 / (ms * 1e-3) / 1e9);
  }
  printf("\n");
  // two-pass pattern: read chunk (pass 1) then read it again (pass 2), chunk by chunk over a 2 GB buffer
  printf("chunked two-pass over 2048 MB (GB/s of bytes touched = 2x footprint): ");
  for (size_t chunk_mb : {16, 32, 64, 128, 256, 2048}) {
    const size_t total_mb = 2048; float ms;
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r)
      for (size_t off = 0; off < total_mb; off += chunk_mb) {
        const u32x4* base = a + (off << 20) / 16; const size_t n_vec = (chunk_mb << 20) / 16;
        hipLaunchKernelGGL(read_kernel<8>, dim3(2048), dim3(256), 0, 0, base, n_vec, sink);
        hipLaunchKernelGGL(read_kernel<8>, dim3(2048), dim3(256), 0, 0, base, n_vec, sink);
      }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf(" %zuMB=%.0f", chunk_mb, 2.0 * total_mb * 1048576 * 3 / (ms * 1e-3) / 1e9);
  }
  printf("\n");
  return 0;
}
