// Micro-benchmark (hipcc --offload-arch=gfx950 -O3 fma_latency.hip -o fma_latency): issue cost of fp64 FMA chains at one / two / four waves per SIMD, and
// the shader clock the chip sustains while doing it (s_memtime ticks against the
// constant-rate s_memrealtime).  One wave per SIMD is the occupancy of the fused IK kernels.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH>
__global__ __launch_bounds__(64) void chains(double* out, long long* clk, int iters, double a, double b) {
  extern __shared__ double pad[];
  double x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 64 / CH; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) x[c] = fma(x[c], a, b);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  if (s == 12345.678) pad[threadIdx.x] = s;
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int CH>
void run(double* d, long long* clk, int waves_per_simd) {
  const int iters = 20000;
  const int lds = 160 * 1024 / (4 * waves_per_simd) - 512;
  const int blocks = 1024 * waves_per_simd;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<CH>, dim3(blocks), dim3(64), lds, 0, d, clk, 10, 0.999, 1e-3);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(chains<CH>, dim3(blocks), dim3(64), lds, 0, d, clk, iters, 0.999, 1e-3);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[2];
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double fmas = (double)iters * 64 * waves_per_simd;   // per SIMD
  const double tflops = (double)iters * 64 * blocks * 64 * 2 / (ms * 1e-3) / 1e12;
  printf("waves/SIMD=%d chains=%d  %.3f ms  %.2f ns per wave-FMA per SIMD  %.1f TFLOP/s  clock64/wall_clock64 = %.3f\n",
         waves_per_simd, CH, ms, ms * 1e6 / fmas, tflops, (double)h[0] / (double)h[1]);
}

int main() {
  double* d;
  long long* clk;
  (void)hipMalloc(&d, 4096 * 64 * 8);
  (void)hipMalloc(&clk, 16);
  run<1>(d, clk, 1); run<8>(d, clk, 1);
  run<1>(d, clk, 2); run<8>(d, clk, 2);
  run<1>(d, clk, 4); run<8>(d, clk, 4);
  (void)hipFree(d);
  return 0;
}
