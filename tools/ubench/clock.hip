// Shader clock seen by short, sparse kernels: clock64() (s_memtime, shader cycles) against wall_clock64() (100 MHz),
// for a lone one-wave kernel, for 64 workgroups (the any-rank factor's grid) and right after a long busy kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o clock tools/ubench/clock.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_probe(long long* out, int iters) {
  long long c0 = clock64(), w0 = wall_clock64();
  double x = threadIdx.x * 1e-9 + 1.0;
  for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;  // dependent f64 FMA chain
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = w1 - w0;
  }
  if (x == 12345.678) out[2] = 1;
}
__global__ void k_busy(double* out, int iters) {
  double x = threadIdx.x * 1e-9 + 1.0, y = 2.0;
  for (int i = 0; i < iters; ++i) {
    x = x * 1.0000001 + 1e-9;
    y = y * 0.9999999 + 1e-9;
  }
  if (x + y == 12345.678) out[0] = 1;
}
int main() {
  long long* d;
  double* dd;
  hipMalloc(&d, 64);
  hipMalloc(&dd, 64);
  long long h[3];
  const int iters = 100000;
  for (int rep = 0; rep < 3; ++rep) {
    for (int grid : {1, 64, 1024}) {
      hipLaunchKernelGGL(k_probe, dim3(grid), dim3(256), 0, 0, d, iters);
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("grid %4d: %lld shader cycles in %.1f us -> %.0f MHz; %.2f cycles per dependent f64 FMA\n", grid, h[0], h[1] / 100.0,
             (double)h[0] / (h[1] / 100.0), (double)h[0] / iters);
    }
    hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, 0, dd, 4000000);
    hipLaunchKernelGGL(k_probe, dim3(64), dim3(256), 0, 0, d, iters);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("after a long busy kernel, grid 64: %.0f MHz\n", (double)h[0] / (h[1] / 100.0));
  }
  return 0;
}
