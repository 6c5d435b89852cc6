// HBM store ceilings on one MI355X for the shapes the sample GEMM could write (4.1 GB per launch):
//   linear: every workgroup streams one contiguous 512 KB slab;  rows: 128 rows x 512 B per step at a 4000 B row stride
//   (the GEMM's tile), rows128: the same at a 4096 B stride (128-byte aligned rows)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k_linear(double* y, size_t per_wg) {
  double* p = y + (size_t)blockIdx.x * per_wg;
  for (size_t i = threadIdx.x; i < per_wg; i += 512) p[i] = (double)i;
}
__global__ void __launch_bounds__(512) k_rows(double* y, int stride, int Lg, int S) {
  // workgroup = (edge, 128-row block); per step 64 columns: wave w rows 16 w .. 16 w + 15, lane -> column
  const int edge = blockIdx.x >> 3, rb = blockIdx.x & 7;
  double* ye = y + (size_t)edge * S * stride;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j0 = 0; j0 < Lg; j0 += 64)
    for (int r = 0; r < 16; ++r) {
      const int s = rb * 128 + 16 * w + r, j = j0 + lane;
      if (s < S && j < Lg) ye[(size_t)s * stride + j] = (double)j;
    }
}
int main() {
  const size_t B = 1024, S = 1000;
  double* y;
  hipMalloc(&y, B * S * 512 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int mode = 0; mode < 6; ++mode) {
    float best = 1e9;
    const size_t lds = mode == 3 ? 50 * 1024 : mode == 4 ? 76 * 1024 : mode == 5 ? 150 * 1024 : 0;  // 3, 2, 1 workgroups per CU
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_linear, dim3(8192), dim3(512), 0, 0, y, (size_t)B * S * 500 / 8192);
      if (mode == 1) hipLaunchKernelGGL(k_rows, dim3(8192), dim3(512), 0, 0, y, 500, 500, 1000);
      if (mode == 2) hipLaunchKernelGGL(k_rows, dim3(8192), dim3(512), 0, 0, y, 512, 500, 1000);
      if (mode >= 3) hipLaunchKernelGGL(k_rows, dim3(8192), dim3(512), lds, 0, y, 500, 500, 1000);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("mode %d (%s): %.3f ms, %.2f TB/s\n", mode, mode == 0 ? "linear" : mode == 1 ? "rows, stride 4000 B" : mode == 2 ? "rows, stride 4096 B" : mode == 3 ? "rows 4000 B, 3 workgroups per CU" : mode == 4 ? "rows 4000 B, 2 workgroups per CU" : "rows 4000 B, 1 workgroup per CU", best,
           1024.0 * 1000 * 500 * 8 / (best * 1e-3) / 1e12);
  }
  return 0;
}
