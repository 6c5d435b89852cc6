// What keeps the f64 matrix pipe of the sample GEMM at ~50 %?  The kernel's loop rebuilt piece by piece: 8 waves per
// workgroup, 2 workgroups per CU, 8 tiles x 4 column groups x 18 MFMAs per wave (= one workgroup of k_sample_gemm_mfma_r<18>).
//   MODE 0: MFMAs on register operands, one accumulator chain      1: + B operand from LDS (stride 65)
//        2: + two barriers per tile                               3: + chunk staging (global -> registers -> LDS)
//        4: + the 16 stores per tile    5: stores with 1/18 of the MFMAs    6: as 5, 512 contiguous bytes per store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define KS 18
template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) k(const double* A, double* Y, int Lg, int S) {
  extern __shared__ double s_fa[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int s0 = (blockIdx.x & 7) * 128;
  const double* Ae = A + (size_t)(blockIdx.x >> 3) * 72 * Lg;
  double* Ye = Y + (size_t)(blockIdx.x >> 3) * S * Lg;
  double areg[KS];
  for (int q = 0; q < KS; ++q) areg[q] = 1.0 + 1e-9 * (tid + q);
  for (int e = tid; e < 72 * 65; e += 512) s_fa[e] = 1.0 + 1e-9 * e;
  constexpr int PF = 9;
  double pf[PF];
  for (int u = 0; u < PF; ++u) pf[u] = 1.0;
  __syncthreads();
  for (int j0 = 0; j0 < Lg; j0 += 64) {
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int e = tid + 512 * u;
        s_fa[(e >> 6) * 65 + (e & 63)] = pf[u];
      }
    }
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3 && j0 + 64 < Lg) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int e = tid + 512 * u;
        const int kk = e >> 6, j = j0 + 64 + (e & 63);
        pf[u] = (j < Lg) ? Ae[(size_t)kk * Lg + j] : 0.0;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < (MODE >= 5 ? 1 : KS); ++q) {
        const double b = MODE >= 1 ? s_fa[(4 * q + lq) * 65 + li + 16 * t] : areg[(q + 1) % KS];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q], b, acc, 0, 0, 0);
      }
      const int j = j0 + 16 * t + li;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int sidx = s0 + 16 * w + lq + 4 * g;
        if (MODE == 6) {  // same bytes, 512 contiguous bytes per wave instruction (lane -> column), rows 16 w + 4 t + g
          const int r2 = s0 + 16 * w + 4 * t + g, c2 = j0 + lane;
          if (r2 < S && c2 < Lg) Ye[(size_t)r2 * Lg + c2] = acc[g];
        } else if (MODE >= 4) {
          if (sidx < S && j < Lg) Ye[(size_t)sidx * Lg + j] = acc[g];
        } else if (acc[g] == 1.2345e300) {
          Ye[(size_t)sidx * Lg + j] = acc[g];
        }
      }
    }
  }
}
template <int MODE>
void run(const double* A, double* Y, int B) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = 72 * 65 * 8;
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(8 * B), dim3(512), lds, 0, A, Y, 500, 1000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("mode %d: %.3f ms  (%.1f TFLOP/s)\n", MODE, best, 1024.0 * 128 * 8 * 512 * 72 * 2 / (best * 1e-3) / 1e12 * B / 1024);
}
int main() {
  const int B = 1024;
  double *A, *Y;
  hipMalloc(&A, (size_t)B * 72 * 500 * 8);
  hipMalloc(&Y, (size_t)B * 1000 * 500 * 8);
  hipMemset(A, 0, (size_t)B * 72 * 500 * 8);
  run<0>(A, Y, B);
  run<1>(A, Y, B);
  run<2>(A, Y, B);
  run<3>(A, Y, B);
  run<4>(A, Y, B);
  run<5>(A, Y, B);
  run<6>(A, Y, B);
  return 0;
}
