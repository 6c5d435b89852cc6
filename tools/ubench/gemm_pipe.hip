// What keeps the f64 matrix pipe of the sample GEMM at ~50 %?  The kernel's loop rebuilt piece by piece: 8 waves per
// workgroup, 2 workgroups per CU, 8 tiles x 4 column groups x 18 MFMAs per wave (= one workgroup of k_sample_gemm_mfma_r<18>).
//   MODE 0: MFMAs on register operands, one accumulator chain      1: + B operand from LDS (stride 65)
//        2: + two barriers per tile                               3: + chunk staging (global -> registers -> LDS)
//        4: + the 16 stores per tile    5: stores with 1/18 of the MFMAs    6: as 5, 512 contiguous bytes per store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define KS 18
// a workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global store
#define LDS_BARRIER()                                                   \
  do {                                                                  \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");     \
    __builtin_amdgcn_s_barrier();                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");     \
  } while (0)
template <int MODE, bool RAWBAR>
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
    if (MODE >= 2 && MODE != 8) { if (RAWBAR) LDS_BARRIER(); else __syncthreads(); }
    if (MODE >= 3 && MODE != 8) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int e = tid + 512 * u;
        s_fa[(e >> 6) * 65 + (e & 63)] = pf[u];
      }
    }
    if (MODE >= 2 && MODE != 8) { if (RAWBAR) LDS_BARRIER(); else __syncthreads(); }
    if (MODE >= 3 && MODE != 8 && j0 + 64 < Lg) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int e = tid + 512 * u;
        const int kk = e >> 6, j = j0 + 64 + (e & 63);
        pf[u] = (j < Lg) ? Ae[(size_t)kk * Lg + j] : 0.0;
      }
    }
    if (MODE == 7) {
      // stores of column group t issued inside the MFMA chain of group t + 1 (after its 4th MFMA)
      v4f64 accs[4];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        if (t < 4) accs[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          if (t < 4) accs[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q], s_fa[(4 * q + lq) * 65 + li + 16 * t], accs[t], 0, 0, 0);
          if (q == 3 && t > 0) {
            const int j = j0 + 16 * (t - 1) + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int sidx = s0 + 16 * w + lq + 4 * g;
              if (sidx < S && j < Lg) Ye[(size_t)sidx * Lg + j] = accs[t - 1][g];
            }
          }
        }
      }
    } else {
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
        if (MODE == 6 || MODE == 8) {  // same bytes, 512 contiguous bytes per wave instruction (lane -> column), rows 16 w + 4 t + g
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
}
// mode 9: chunk double-buffered in LDS; the next chunk's loads are waited for and written to the other buffer INSIDE the
// tile (after the third column group), one LDS-only barrier per tile: no wait on global stores anywhere in the loop
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) k9(const double* A, double* Y, int Lg, int S) {
  extern __shared__ double s_fa2[];  // [2][72 * 65]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int s0 = (blockIdx.x & 7) * 128;
  const double* Ae = A + (size_t)(blockIdx.x >> 3) * 72 * Lg;
  double* Ye = Y + (size_t)(blockIdx.x >> 3) * S * Lg;
  double areg[KS];
  for (int q = 0; q < KS; ++q) areg[q] = 1.0 + 1e-9 * (tid + q);
  constexpr int PF = 9;
  double pf[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int e = tid + 512 * u;
    pf[u] = Ae[(size_t)(e >> 6) * Lg + (e & 63)];
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int e = tid + 512 * u;
    s_fa2[(e >> 6) * 65 + (e & 63)] = pf[u];
  }
  LDS_BARRIER();
  int buf = 0;
  for (int j0 = 0; j0 < Lg; j0 += 64) {
    const double* cur = s_fa2 + buf * 72 * 65;
    double* nxt = s_fa2 + (buf ^ 1) * 72 * 65;
    const bool more = j0 + 64 < Lg;
    if (more) {
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
      for (int q = 0; q < KS; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q], cur[(4 * q + lq) * 65 + li + 16 * t], acc, 0, 0, 0);
      if (t == 3 && more) {  // the next chunk into the other buffer before this group's stores are issued
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int e = tid + 512 * u;
          nxt[(e >> 6) * 65 + (e & 63)] = pf[u];
        }
      }
      const int j = j0 + 16 * t + li;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int sidx = s0 + 16 * w + lq + 4 * g;
        if (sidx < S && j < Lg) Ye[(size_t)sidx * Lg + j] = acc[g];
      }
    }
    LDS_BARRIER();
    buf ^= 1;
  }
}
template <int MODE, bool RAWBAR = false>
void run(const double* A, double* Y, int B) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = 72 * 65 * 8;
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, RAWBAR>), dim3(8 * B), dim3(512), lds, 0, A, Y, 500, 1000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("mode %d%s: %.3f ms  (%.1f TFLOP/s)\n", MODE, RAWBAR ? " (LDS-only barrier)" : "", best, 1024.0 * 128 * 8 * 512 * 72 * 2 / (best * 1e-3) / 1e12 * B / 1024);
}
int main() {
  hipFuncSetAttribute((const void*)k9, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
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
  run<7>(A, Y, B);
  run<8>(A, Y, B);
  {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k9, dim3(8 * B), dim3(512), 2 * 72 * 65 * 8, 0, A, Y, 500, 1000);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("mode 9 (double-buffered chunk, no store waits): %.3f ms\n", best);
  }
  run<4, true>(A, Y, B);
  run<5, true>(A, Y, B);
  run<6, true>(A, Y, B);
  return 0;
}
