// LDS read throughput microbenchmark on gfx950: what a wave-instruction of each flavour costs at the CU level with the attention
// kernels' tile layout (256-byte rows, 16-byte part p of row r at p ^ swz(r)) -- in particular ds_read_b64_tr_b16 vs ds_read_b128.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_tr_bench tools/lds_tr_bench.hip && tools/lds_tr_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int chunk_off(int row, int part, bool swizzle) { return row * 256 + (((swizzle ? swz(row) : 0) ^ part) << 4); }

template <int MODE>   // 0: tr16_b64 swizzled (the kernels' addresses), 1: tr16_b64 unswizzled, 2: plain b64 at the same addresses, 3: b128 row reads, 4: b64 row reads
__global__ __launch_bounds__(512) void bench(unsigned long long* out, int iters) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t addr[8];
  const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  if (MODE <= 2) {
    const int p16 = lane & 15, hh = lane >> 5;
    for (int dt = 0; dt < 4; ++dt)
      for (int hi = 0; hi < 2; ++hi)
        addr[2 * dt + hi] = base + chunk_off(8 * hh + (p16 >> 2) + 4 * hi, 4 * dt + 2 * ((lane >> 4) & 1) + ((p16 & 3) >> 1), MODE != 1) + 8 * (p16 & 1);
  } else {
    const int j = lane & 31, hh = lane >> 5;
    const int row = (j & 0x13) | ((j & 4) << 1) | ((j & 8) >> 1);
    for (int st = 0; st < 8; ++st) addr[st] = base + chunk_off(row, 2 * st + hh, true);
  }
  uint32_t acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int off = (it & 1) * 16 * 256;       // two 16-row steps of the tile, immediates in the real kernels
    if (MODE == 3) {
      u32x4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(r[u]) : "v"(addr[u] + off));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= r[u][0] ^ r[u][3];
    } else {
      u32x2 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (MODE <= 1) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[u]) : "v"(addr[u] + off));
        else asm volatile("ds_read_b64 %0, %1" : "=v"(r[u]) : "v"(addr[u] + off));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= r[u][0] ^ r[u][1];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0 + (acc == 0x12345u ? 1 : 0);
}

template <int MODE>
static void run(const char* name, int waves) {
  unsigned long long* d;
  hipMalloc(&d, 256 * 8 * sizeof(unsigned long long));
  const int iters = 2000;
  hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(64 * waves), 65536, 0, d, iters);
  hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(64 * waves), 65536, 0, d, iters);
  hipDeviceSynchronize();
  unsigned long long h[256 * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < waves; ++w) s += (double)h[b * 8 + w];
  const double per_wave = s / (256.0 * waves) / (iters * 8.0);
  printf("%-44s %d waves/CU: %6.1f cycles per instruction per wave = %5.1f cycles per instruction at the CU (%5.1f B/clk/CU)\n", name, waves,
         per_wave, per_wave / waves, (MODE == 3 ? 1024.0 : 512.0) / (per_wave / waves));
  hipFree(d);
}

int main() {
  for (int waves : {4, 8}) {
    run<0>("ds_read_b64_tr_b16, swizzled tile (kernels)", waves);
    run<1>("ds_read_b64_tr_b16, unswizzled rows", waves);
    run<2>("ds_read_b64 at the tr addresses", waves);
    run<3>("ds_read_b128 row fragments, swizzled", waves);
  }
  return 0;
}
