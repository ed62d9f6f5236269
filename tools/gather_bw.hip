// What does the kept-row gather's access pattern allow?  58 880 rows of 7 168 B (B = 128 images x 460 kept rows of 3584 bf16) copied out of a
// 2.1 GB tensor into a contiguous destination: contiguous vs sorted-random source rows, default vs non-temporal loads / stores, resident
// waves x rows in flight, and the read half / write half alone.  GB/s = (read + written bytes) / time.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/gather_bw.hip -o /tmp/gather_bw && /tmp/gather_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITERS = 7;       // 7 x 64 lanes x 16 B = 7168 B per row

// MODE 0 copy, 1 read only (xor into a sink), 2 write only
template <bool NTL, bool NTS, int DEPTH, int MODE>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, const int* __restrict__ rows, u32x4* __restrict__ dst, int total, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = gridDim.x * 4;
  int g = blockIdx.x * 4 + wave;
  u32x4 x[DEPTH][ITERS];
  unsigned acc = 0;
  auto load = [&](u32x4 (&v)[ITERS], int gi) {
    const u32x4* sp = src + (size_t)rows[gi] * (ITERS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) v[i] = (MODE == 2) ? u32x4{(unsigned)gi, 1u, 2u, 3u} : (NTL ? __builtin_nontemporal_load(sp + 64 * i) : sp[64 * i]);
  };
  auto store = [&](const u32x4 (&v)[ITERS], int gi) {
    u32x4* dp = dst + (size_t)gi * (ITERS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      if (MODE == 1) acc ^= v[i][0] ^ v[i][3];
      else if (NTS) __builtin_nontemporal_store(v[i], dp + 64 * i);
      else dp[64 * i] = v[i];
    }
  };
  int gl = g;
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) { if (gl < total) load(x[u], gl); gl += W; }
  auto step = [&](u32x4 (&xu)[ITERS]) {
    store(xu, g);
    g += W;
    if (gl < total) load(xu, gl);
    gl += W;
  };
  for (;;) {
    if (g >= total) break;
    step(x[0]);
    if (g >= total) break;
    step(x[1]);
    if constexpr (DEPTH > 2) { if (g >= total) break; step(x[2]); }
    if constexpr (DEPTH > 3) { if (g >= total) break; step(x[3]); }
  }
  if (MODE == 1 && acc == 0x12345u) sink[0] = acc;
}

template <bool NTL, bool NTS, int DEPTH, int MODE>
static void run(const char* tag, const u32x4* src, const int* rows, u32x4* dst, int total, unsigned* sink, int per_cu) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * per_cu;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<NTL, NTS, DEPTH, MODE>), dim3(grid), dim3(256), 0, 0, src, rows, dst, total, sink);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<NTL, NTS, DEPTH, MODE>), dim3(grid), dim3(256), 0, 0, src, rows, dst, total, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms / reps * 1e3;
  const double bytes = (double)total * ITERS * 1024 * (MODE == 0 ? 2 : 1);
  printf("%-44s wg/CU %d  %7.1f us  %7.0f GB/s\n", tag, per_cu, us, bytes / us / 1e3);
}

// a whole-tensor streaming read in front of every copy (what the score sweep leaves behind: caches full of the tensor's tail)
__global__ __launch_bounds__(256) void sweep(const u32x4* __restrict__ p, size_t n_vec, unsigned* sink) {
  size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * 8;
  unsigned acc = 0;
  for (; i + 256 * 7 < n_vec; i += stride) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + 256 * u);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <bool NTL, bool NTS>
static void run_after_sweep(const char* tag, const u32x4* src, size_t src_vec, const int* rows, u32x4* dst, int total, unsigned* sink, int per_cu) {
  const int reps = 10;
  std::vector<hipEvent_t> evs(2 * reps);
  for (auto& x : evs) CK(hipEventCreate(&x));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, src, src_vec, sink);
    CK(hipEventRecord(evs[2 * i]));
    hipLaunchKernelGGL((k<NTL, NTS, 2, 0>), dim3(256 * per_cu), dim3(256), 0, 0, src, rows, dst, total, sink);
    CK(hipEventRecord(evs[2 * i + 1]));
  }
  CK(hipDeviceSynchronize());
  double us = 0;
  for (int i = 2; i < reps; ++i) { float ms; CK(hipEventElapsedTime(&ms, evs[2 * i], evs[2 * i + 1])); us += ms * 1e3; }
  us /= (reps - 2);
  printf("%-44s wg/CU %d  %7.1f us  %7.0f GB/s   (each copy right behind a 2.1 GB streaming read)\n", tag, per_cu, us, (double)total * ITERS * 1024 * 2 / us / 1e3);
}

int main() {
  const int B = 128, N = 2304, K = 460, total = B * K;
  const size_t row_vec = ITERS * 64;
  u32x4 *src, *dst; int *rows_c, *rows_r; unsigned* sink;
  CK(hipMalloc(&src, (size_t)B * N * row_vec * 16)); CK(hipMalloc(&dst, (size_t)total * row_vec * 16));
  CK(hipMalloc(&rows_c, total * 4)); CK(hipMalloc(&rows_r, total * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(src, 1, (size_t)B * N * row_vec * 16));
  std::vector<int> rc(total), rr(total);
  srand(1);
  for (int b = 0; b < B; ++b) {
    std::vector<int> perm(N);
    for (int i = 0; i < N; ++i) perm[i] = i;
    for (int i = N - 1; i > 0; --i) std::swap(perm[i], perm[rand() % (i + 1)]);
    std::sort(perm.begin(), perm.begin() + K);
    for (int j = 0; j < K; ++j) { rr[b * K + j] = b * N + perm[j]; rc[b * K + j] = b * N + j; }
  }
  CK(hipMemcpy(rows_c, rc.data(), total * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rows_r, rr.data(), total * 4, hipMemcpyHostToDevice));
  printf("rows %d x 7168 B: %.1f MB read + the same written\n", total, total * 7168.0 / 1e6);
  for (int pc : {4, 8}) {
    run<true, true, 2, 0>("copy, random kept rows, nt loads + nt stores", src, rows_r, dst, total, sink, pc);
    run<false, true, 2, 0>("copy, random kept rows, plain loads + nt stores", src, rows_r, dst, total, sink, pc);
    run<true, false, 2, 0>("copy, random kept rows, nt loads + plain stores", src, rows_r, dst, total, sink, pc);
    run<false, false, 2, 0>("copy, random kept rows, plain + plain", src, rows_r, dst, total, sink, pc);
    run<true, true, 2, 0>("copy, first 460 rows of each image (contiguous)", src, rows_c, dst, total, sink, pc);
    run<true, true, 2, 1>("read half only, random kept rows", src, rows_r, dst, total, sink, pc);
    run<true, true, 2, 2>("write half only", src, rows_r, dst, total, sink, pc);
  }
  run_after_sweep<true, true>("copy, random kept rows, nt + nt", src, (size_t)B * N * row_vec, rows_r, dst, total, sink, 8);
  run_after_sweep<false, true>("copy, random kept rows, plain loads + nt stores", src, (size_t)B * N * row_vec, rows_r, dst, total, sink, 8);
  run_after_sweep<true, false>("copy, random kept rows, nt loads + plain stores", src, (size_t)B * N * row_vec, rows_r, dst, total, sink, 8);
  run<true, true, 4, 0>("copy, random kept rows, nt + nt, 4 rows in flight", src, rows_r, dst, total, sink, 4);
  run<true, true, 3, 0>("copy, random kept rows, nt + nt, 3 rows in flight", src, rows_r, dst, total, sink, 5);
  return 0;
}
