// The sample GEMM's loop (one workgroup of k_sample_gemm_mfma_r<18>: 8 waves, 2 workgroups per CU, 8 tiles x 4 column
// groups x 18 MFMAs per wave, chunk staged global -> registers -> LDS) with its STORE SIDE varied:
//   ST 0: today's form -- one 8-byte store per accumulator register: 4 rows x 128 contiguous bytes per instruction
//   ST 1: the columns of two groups interleaved (group a <- even columns, group b <- odd columns of a 32-column half, the
//         chunk permuted accordingly when it is staged): a lane holds two ADJACENT columns of a row -> one 16-byte store:
//         4 rows x 256 contiguous bytes per instruction, half the store instructions
//   ST 2: four groups interleaved (lane <- 4 adjacent columns): two 16-byte stores per row, 4 rows x 512 bytes per pair
//   EPI: the epilogue arithmetic ((acc + mu) * y_s) or none (mean and scale folded into the operands)
//   pitch: 500 doubles (rows 4 000 bytes apart: every other 128-byte run straddles three lines) or 512
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));
#define KS 18
#define LDA 80
template <int ST, bool EPI>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
k(const double* __restrict__ A, double* __restrict__ Y, const double* __restrict__ mean, int Lg, int S, int pitch, double y_s) {
  extern __shared__ double s_fa[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int s0 = (blockIdx.x & 7) * 128;
  const __attribute__((address_space(1))) double* Ae = (const __attribute__((address_space(1))) double*)(A + (size_t)(blockIdx.x >> 3) * 72 * Lg);
  __attribute__((address_space(1))) double* Ye = (__attribute__((address_space(1))) double*)(Y + (size_t)(blockIdx.x >> 3) * S * pitch);
  double* s_mu = s_fa + 72 * LDA;
  for (int j = tid; j < Lg; j += 512) s_mu[j] = mean[j];
  double areg[KS];
  for (int q = 0; q < KS; ++q) areg[q] = 1.0 + 1e-9 * (tid + q);
  constexpr int PF = 9;
  double pf[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int e = tid + 512 * u;
    pf[u] = Ae[(size_t)(e >> 6) * Lg + (e & 63)];
  }
  for (int j0 = 0; j0 < Lg; j0 += 64) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int e = tid + 512 * u;
      const int c = e & 63;
      // position of column c in the staged row: plain, pairs (32 p + 16 (c & 1) + (c >> 1 & 15)), or quads (16 (c & 3) + (c >> 2))
      const int pos = ST == 0 ? c : ST == 1 ? (c & 32) + 16 * (c & 1) + ((c >> 1) & 15) : 16 * (c & 3) + (c >> 2);
      s_fa[(e >> 6) * LDA + pos] = pf[u];
    }
    __syncthreads();
    if (j0 + 64 < Lg) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int e = tid + 512 * u;
        const int kk = e >> 6, j = j0 + 64 + (e & 63);
        pf[u] = (j < Lg) ? Ae[(size_t)kk * Lg + j] : 0.0;
      }
    }
    if (ST == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < KS; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q], s_fa[(4 * q + lq) * LDA + li + 16 * t], acc, 0, 0, 0);
        const int j = j0 + 16 * t + li;
        if (j >= Lg) continue;
        const double mu = EPI ? s_mu[j] : 0.0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int sidx = s0 + 16 * w + lq + 4 * g;
          if (sidx < S) Ye[(size_t)sidx * pitch + j] = EPI ? (acc[g] + mu) * y_s : acc[g];
        }
      }
    } else if (ST == 1) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        v4f64 a0 = (v4f64){0.0, 0.0, 0.0, 0.0}, a1 = a0;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q], s_fa[(4 * q + lq) * LDA + 32 * p + li], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q], s_fa[(4 * q + lq) * LDA + 32 * p + 16 + li], a1, 0, 0, 0);
        }
        const int j = j0 + 32 * p + 2 * li;
        if (j >= Lg) continue;  // (Lg even: a pair is inside or outside)
        v2f64 mu = (v2f64){0.0, 0.0};
        if (EPI) mu = *(const v2f64*)&s_mu[j];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int sidx = s0 + 16 * w + lq + 4 * g;
          v2f64 o;
          o.x = EPI ? (a0[g] + mu.x) * y_s : a0[g];
          o.y = EPI ? (a1[g] + mu.y) * y_s : a1[g];
          if (sidx < S) *(__attribute__((address_space(1))) v2f64*)&Ye[(size_t)sidx * pitch + j] = o;
        }
      }
    } else {
      v4f64 a[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < KS; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q], s_fa[(4 * q + lq) * LDA + 16 * t + li], a[t], 0, 0, 0);
      const int j = j0 + 4 * li;
      if (j < Lg) {
        v2f64 m0 = (v2f64){0.0, 0.0}, m1 = m0;
        if (EPI) { m0 = *(const v2f64*)&s_mu[j]; m1 = *(const v2f64*)&s_mu[j + 2]; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int sidx = s0 + 16 * w + lq + 4 * g;
          v2f64 o0, o1;
          o0.x = EPI ? (a[0][g] + m0.x) * y_s : a[0][g];
          o0.y = EPI ? (a[1][g] + m0.y) * y_s : a[1][g];
          o1.x = EPI ? (a[2][g] + m1.x) * y_s : a[2][g];
          o1.y = EPI ? (a[3][g] + m1.y) * y_s : a[3][g];
          if (sidx < S) {
            *(__attribute__((address_space(1))) v2f64*)&Ye[(size_t)sidx * pitch + j] = o0;
            *(__attribute__((address_space(1))) v2f64*)&Ye[(size_t)sidx * pitch + j + 2] = o1;
          }
        }
      }
    }
  }
}
template <int ST, bool EPI>
void run(const double* A, double* Y, const double* mean, int B, int pitch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = (72 * LDA + 512) * 8;
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ST, EPI>), dim3(8 * B), dim3(512), lds, 0, A, Y, mean, 500, 1000, pitch, 1.25);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("store form %d, epilogue %d, pitch %d: %.3f ms  (%.1f TFLOP/s, %.2f TB/s)\n", ST, (int)EPI, pitch, best,
         1000.0 * 500 * 72 * 2 * B / (best * 1e-3) / 1e12, 1000.0 * 500 * 8 * B / (best * 1e-3) / 1e12);
}
int main() {
  const int B = 1024;
  double *A, *Y, *mean;
  hipMalloc(&A, (size_t)B * 72 * 500 * 8);
  hipMalloc(&Y, (size_t)B * 1000 * 512 * 8);
  hipMalloc(&mean, 512 * 8);
  hipMemset(A, 0, (size_t)B * 72 * 500 * 8);
  hipMemset(mean, 0, 512 * 8);
  for (int pitch = 500; pitch <= 512; pitch += 12) {
    run<0, true>(A, Y, mean, B, pitch);
    run<0, false>(A, Y, mean, B, pitch);
    run<1, true>(A, Y, mean, B, pitch);
    run<1, false>(A, Y, mean, B, pitch);
    run<2, true>(A, Y, mean, B, pitch);
    run<2, false>(A, Y, mean, B, pitch);
  }
  return 0;
}
