// Does a sweep served by the Infinity Cache (MALL) run CONCURRENTLY with an HBM sweep at more than the HBM rate?
// Decides whether a chunked LIS pipeline (sweep 1 of chunk c+1 from HBM beside sweep 2 / gather of chunk c from the MALL) can beat
// the two-full-sweeps floor.  Build: hipcc --offload-arch=gfx950 -O3 tools/membw3.hip -o tools/membw3 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

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

static double now_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  const size_t total = (size_t)2 << 30;
  u32x4* a; unsigned* sink;
  CK(hipMalloc(&a, total)); CK(hipMalloc(&sink, 4)); CK(hipMemset(a, 1, total));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<hipEvent_t> ev(64);
  for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));

  // 1. MALL-resident re-read rate vs loads in flight (footprint 128 MB, read 16x per timing)
  printf("1. Infinity-Cache-resident reads, 128 MB footprint, GB/s (grid x unroll):\n");
  for (int grid : {1024, 2048, 4096}) {
    const size_t nv = ((size_t)128 << 20) / 16;
    auto run = [&](auto kern) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s1, a, nv, sink);
      CK(hipEventRecord(e0, s1));
      for (int r = 0; r < 16; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s1, a, nv, sink);
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      return (double)nv * 16 * 16 / (now_ms(e0, e1) * 1e-3) / 1e9;
    };
    printf("  grid %4d: u4 %6.0f  u8 %6.0f  u16 %6.0f | nt u8 %6.0f\n", grid, run(read_kernel<4, false>), run(read_kernel<8, false>),
           run(read_kernel<16, false>), run(read_kernel<8, true>));
  }

  // 2. two sweeps over 2 GiB in chunks: sequential (sweep1(c), sweep2(c) on one stream) vs pipelined on two streams
  //    (sweep2(c) on s2 beside sweep1(c+1) on s1).  GB/s of bytes touched (= 2 x 2 GiB).
  printf("2. chunked two-sweep over 2 GiB, GB/s of bytes touched (2 x footprint): sequential | two-stream pipelined | pipelined, sweep 1 nt\n");
  for (size_t chunk_mb : {32, 64, 96, 128, 192, 256}) {
    const size_t cb = chunk_mb << 20, nc = total / cb, nv = cb / 16;
    const int grid = 2048;
    auto seq = [&]() {
      CK(hipEventRecord(e0, s1));
      for (size_t c = 0; c < nc; ++c) {
        hipLaunchKernelGGL((read_kernel<8, false>), dim3(grid), dim3(256), 0, s1, a + c * nv, nv, sink);
        hipLaunchKernelGGL((read_kernel<8, false>), dim3(grid), dim3(256), 0, s1, a + c * nv, nv, sink);
      }
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      return 2.0 * total / (now_ms(e0, e1) * 1e-3) / 1e9;
    };
    auto pipe = [&](bool nt1) {
      CK(hipEventRecord(e0, s1));
      CK(hipStreamWaitEvent(s2, e0, 0));
      for (size_t c = 0; c < nc; ++c) {
        if (nt1) hipLaunchKernelGGL((read_kernel<8, true>), dim3(grid), dim3(256), 0, s1, a + c * nv, nv, sink);
        else hipLaunchKernelGGL((read_kernel<8, false>), dim3(grid), dim3(256), 0, s1, a + c * nv, nv, sink);
        CK(hipEventRecord(ev[c % 64], s1));
        CK(hipStreamWaitEvent(s2, ev[c % 64], 0));
        hipLaunchKernelGGL((read_kernel<8, false>), dim3(grid), dim3(256), 0, s2, a + c * nv, nv, sink);
      }
      CK(hipEventRecord(ev[63], s2));
      CK(hipStreamWaitEvent(s1, ev[63], 0));
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      return 2.0 * total / (now_ms(e0, e1) * 1e-3) / 1e9;
    };
    seq(); pipe(false);
    const double a1 = seq(), a2 = pipe(false), a3 = pipe(true);
    printf("  chunk %3zu MB: %6.0f | %6.0f | %6.0f\n", chunk_mb, a1, a2, a3);
  }

  // 3. pure concurrency: one 2 GiB HBM sweep on s1 beside repeated sweeps of a 96 MB resident buffer on s2
  {
    const size_t nv_big = (total - ((size_t)256 << 20)) / 16, nv_small = ((size_t)96 << 20) / 16;
    const u32x4* small = a + nv_big + (((size_t)64 << 20) / 16);
    for (int w = 0; w < 4; ++w) hipLaunchKernelGGL((read_kernel<8, false>), dim3(2048), dim3(256), 0, s2, small, nv_small, sink);
    CK(hipStreamSynchronize(s2));
    CK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL((read_kernel<8, true>), dim3(4096), dim3(256), 0, s1, a, nv_big, sink);
    CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
    const double t_big = now_ms(e0, e1);
    CK(hipEventRecord(e0, s2));
    for (int r = 0; r < 16; ++r) hipLaunchKernelGGL((read_kernel<8, false>), dim3(2048), dim3(256), 0, s2, small, nv_small, sink);
    CK(hipEventRecord(e1, s2)); CK(hipEventSynchronize(e1));
    const double t_small = now_ms(e0, e1);
    CK(hipEventRecord(e0, s1));
    CK(hipStreamWaitEvent(s2, e0, 0));
    hipLaunchKernelGGL((read_kernel<8, true>), dim3(4096), dim3(256), 0, s1, a, nv_big, sink);
    for (int r = 0; r < 16; ++r) hipLaunchKernelGGL((read_kernel<8, false>), dim3(2048), dim3(256), 0, s2, small, nv_small, sink);
    CK(hipEventRecord(ev[0], s2));
    CK(hipStreamWaitEvent(s1, ev[0], 0));
    CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
    const double t_both = now_ms(e0, e1);
    const double gb = (nv_big * 16.0 + 16.0 * nv_small * 16.0) / 1e9;
    printf("3. HBM sweep (%.2f GB, nt) alone %.3f ms (%.0f GB/s); 16 x 96 MB resident sweeps alone %.3f ms (%.0f GB/s); both concurrently %.3f ms (%.0f GB/s aggregate; serial sum %.3f ms)\n",
           nv_big * 16.0 / 1e9, t_big, nv_big * 16.0 / t_big / 1e6, t_small, 16.0 * nv_small * 16.0 / t_small / 1e6, t_both, gb / t_both * 1e3, t_big + t_small);
  }
  return 0;
}
