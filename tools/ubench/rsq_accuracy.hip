// Micro-benchmark (hipcc --offload-arch=gfx950 -O3 rsq_accuracy.hip -o rsq_accuracy): relative
// error of the hardware fp64 reciprocal / reciprocal-square-root seeds and of one and two
// Newton steps on them, over 1 M arguments spread over 24 binades.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double y0 = __builtin_amdgcn_rsq(v);
  const double hx = 0.5 * v;
  double y1 = fma(fma(-hx * y0, y0, 0.5), y0, y0);
  double y2 = fma(fma(-hx * y1, y1, 0.5), y1, y1);
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = fma(fma(-v, r0, 1.0), r0, r0);
  double r2 = fma(fma(-v, r1, 1.0), r1, r1);
  out[6 * i + 0] = y0; out[6 * i + 1] = y1; out[6 * i + 2] = y2;
  out[6 * i + 3] = r0; out[6 * i + 4] = r1; out[6 * i + 5] = r2;
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(6 * (size_t)n);
  unsigned s = 1;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    const double m = 1.0 + (s >> 8) / 16777216.0;
    x[i] = std::ldexp(m, (i % 24) - 12);
  }
  double *dx, *dout;
  (void)hipMalloc(&dx, n * 8);
  (void)hipMalloc(&dout, 6 * (size_t)n * 8);
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  (void)hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
  double e[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double rs = 1.0L / sqrtl((long double)x[i]), rc = 1.0L / (long double)x[i];
    for (int j = 0; j < 3; ++j) e[j] = fmax(e[j], (double)fabsl((o[6 * (size_t)i + j] - rs) / rs));
    for (int j = 3; j < 6; ++j) e[j] = fmax(e[j], (double)fabsl((o[6 * (size_t)i + j] - rc) / rc));
  }
  printf("rsq: seed %.3e  1 step %.3e  2 steps %.3e\n", e[0], e[1], e[2]);
  printf("rcp: seed %.3e  1 step %.3e  2 steps %.3e\n", e[3], e[4], e[5]);
  return 0;
}
