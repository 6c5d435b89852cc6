// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (not part of the product).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, int iters, double a0, double b0) {
  v4f64 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = (v4f64){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
  }
  double s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int threads, int iters) {
  double* d;
  hipMalloc(&d, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0000001, 0.9999999);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)blocks * (threads / 64) * iters * NACC;
  const double tflops = n_mfma * 2048.0 / (ms * 1e-3) / 1e12;
  printf("NACC=%d blocks=%d threads=%d: %.3f ms, %.1f TFLOP/s f64, %.1f ns per MFMA per wave-slot\n", NACC, blocks, threads, ms,
         tflops, ms * 1e6 / (iters * NACC));
  hipFree(d);
}
int main() {
  run<1>(256, 256, 20000);
  run<4>(256, 256, 5000);
  run<8>(256, 256, 2500);
  run<4>(256 * 2, 256, 5000);
  run<4>(256 * 4, 256, 5000);
  run<8>(256 * 4, 256, 2500);
  return 0;
}
