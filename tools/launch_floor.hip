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
  return 0;
}
