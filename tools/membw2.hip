// Access-pattern micro-benchmark for the LIS sweeps: [B*N rows] x [D = 3584 bf16 = 7 KiB per row], 2.1 GB.
//   tile : colsum_partial_kernel's pattern (block = 128 rows x 1 KiB column tile, wave w takes rows w, w+4, ..)
//   rows : score_kernel's pattern (block = 64 rows, wave reads whole 7 KiB rows, two rows in flight)
//   lin  : linear grid-stride (the membw ceiling)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void tile_kernel(const u32x4* __restrict__ p, int rows_per_chunk, int row_vecs, unsigned* sink, int unroll8) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t rb = (size_t)blockIdx.y * rows_per_chunk;
  const u32x4* base = p + rb * row_vecs + blockIdx.x * 64 + lane;
  unsigned acc = 0;
  if (unroll8) {
    for (int r = wave; r + 28 < rows_per_chunk; r += 32) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(r + 4 * u) * row_vecs];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
  } else {
    for (int r = wave; r + 12 < rows_per_chunk; r += 16) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = base[(size_t)(r + 4 * u) * row_vecs];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// the real arithmetic of sweep 1: bf16 -> fp32 widening + 8 fp32 accumulators per lane, wave owns one column tile
__global__ __launch_bounds__(256) void colsum_like_kernel(const u32x4* __restrict__ p, int rows_per_chunk, int row_vecs, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= 7) return;
  const size_t rb = (size_t)blockIdx.y * rows_per_chunk;
  const u32x4* base = p + rb * row_vecs + tile * 64 + lane;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r + 8 <= rows_per_chunk; r += 8) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(r + u) * row_vecs];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] += __uint_as_float(v[u][i] << 16);
        acc[2 * i + 1] += __uint_as_float(v[u][i] & 0xffff0000u);
      }
  }
  float* dst = out + ((size_t)blockIdx.y * 7 + tile) * 512 + lane * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i] = acc[i];
}

__global__ __launch_bounds__(256) void rows_kernel(const u32x4* __restrict__ p, int rows_per_block, int row_vecs, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t rb = (size_t)blockIdx.x * rows_per_block;
  unsigned acc = 0;
  for (int r = wave; r + 4 < rows_per_block; r += 8) {
    const u32x4* p0 = p + (rb + r) * row_vecs + lane;
    const u32x4* p1 = p + (rb + r + 4) * row_vecs + lane;
    u32x4 a[7], b[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) a[i] = p0[i * 64];
#pragma unroll
    for (int i = 0; i < 7; ++i) b[i] = p1[i * 64];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc ^= a[i][0] ^ a[i][1] ^ b[i][2] ^ b[i][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void lin_kernel(const u32x4* __restrict__ p, size_t n_vec, unsigned* sink) {
  size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * 8;
  unsigned acc = 0;
  for (; i + 256 * 7 < n_vec; i += stride) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int B = 128, N = 2304, row_vecs = 448;       // 448 x 16 B = 7168 B per row
  const size_t rows = (size_t)B * N, n_vec = rows * row_vecs, bytes = n_vec * 16;
  u32x4* a; unsigned* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int w = 0; w < 2; ++w) launch();
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %7.1f us  %6.0f GB/s\n", name, ms * 100, bytes * 10.0 / (ms * 1e-3) / 1e9);
  };
  timeit("lin grid 4096", [&] { hipLaunchKernelGGL(lin_kernel, dim3(4096), dim3(256), 0, 0, a, n_vec, sink); });
  for (int rpc : {128, 256, 512, 2304}) {
    char nm[64];
    snprintf(nm, 64, "tile %d rows x 1KiB (4 in flight)", rpc);
    timeit(nm, [&] { hipLaunchKernelGGL(tile_kernel, dim3(7, (unsigned)(rows / rpc)), dim3(256), 0, 0, a, rpc, row_vecs, sink, 0); });
    snprintf(nm, 64, "tile %d rows x 1KiB (8 in flight)", rpc);
    timeit(nm, [&] { hipLaunchKernelGGL(tile_kernel, dim3(7, (unsigned)(rows / rpc)), dim3(256), 0, 0, a, rpc, row_vecs, sink, 1); });
  }
  float* outp; CK(hipMalloc(&outp, (rows / 128) * 7 * 512 * 4));
  timeit("colsum-like 128 rows, wave per tile", [&] { hipLaunchKernelGGL(colsum_like_kernel, dim3(2, (unsigned)(rows / 128)), dim3(256), 0, 0, a, 128, row_vecs, outp); });
  for (int rpb : {64, 128}) {
    char nm[64];
    snprintf(nm, 64, "rows %d per block, 7 KiB rows", rpb);
    timeit(nm, [&] { hipLaunchKernelGGL(rows_kernel, dim3((unsigned)(rows / rpb)), dim3(256), 0, 0, a, rpb, row_vecs, sink); });
  }
  return 0;
}
