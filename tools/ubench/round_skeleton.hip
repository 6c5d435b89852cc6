// What does the skeleton of a Jacobi round cost?  One workgroup; wave 0 runs  [8 LDS reads -> rotation-parameter chain -> 3 LDS
// writes]; the other waves only meet it at the barrier (or do a block's worth of LDS traffic: WORK = 1).  Cycles per trip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/round_skeleton.hip -o gpurun_scratch/round_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void jp1(double app, double apq, double aqq, double& c, double& s) {
  const bool on = fabs(apq) > 1e-300 && apq * apq > 1e-36 * fabs(app * aqq);
  const double d = aqq - app, hh = 2.0 * apq;
  const double rho2 = d * d + hh * hh;
  double y = __builtin_amdgcn_rsq(rho2);
  y = y * (1.5 - 0.5 * rho2 * y * y);
  const double den = fabs(d) + rho2 * y;
  double iv = __builtin_amdgcn_rcp(den);
  iv = iv * (2.0 - den * iv);
  const double t = (d >= 0.0 ? hh : -hh) * iv;
  const double u = 1.0 + t * t;
  double cc = __builtin_amdgcn_rsq(u);
  cc = cc * (1.5 - 0.5 * u * cc * cc);
  cc = cc * (1.5 - 0.5 * u * cc * cc);
  c = on ? cc : 1.0;
  s = on ? t * cc : 0.0;
}
template <int MODE, int WORK>  // MODE 0: chain only; 1: + LDS reads / writes; 2: + barrier
__global__ void k(double* out, long long* cyc, int n) {
  __shared__ double sm[8192];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 8192; i += blockDim.x) sm[i] = 1.0 + 1e-3 * i;
  __syncthreads();
  double app = 3.0 + lane * 1e-3, apq = 0.3, aqq = 6.0, c = 1.0, s = 0.0;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    const int P = i & 1;
    if (w == 0) {
      if (MODE >= 1) {
        const double* b = sm + P * 4096 + lane;
        const double v = ((b[0] + b[64]) + (b[128] + b[192])) + ((b[256] + b[320]) + (b[384] + b[448]));
        apq = v * 1e-3;
      }
      jp1(app, apq, aqq, c, s);
      app = __builtin_fma(c, 0.25, app);
      aqq = __builtin_fma(s, 0.125, aqq);
      if (MODE >= 1) {
        double* o = sm + (P ^ 1) * 4096 + lane;
        o[0] = c; o[64] = s; o[128] = app;
      } else {
        apq = __builtin_fma(s, 0.5, 0.1);
      }
    } else if (WORK) {
      const double* b = sm + P * 4096 + 512 + tid;
      const double v0 = b[0], v1 = b[1024], v2 = b[2048], v3 = b[3072 - 512];
      double* o = sm + (P ^ 1) * 4096 + 512 + tid;
      o[0] = v0 * 1.0001 + v1; o[1024] = v1 * 1.0001 - v0; o[2048] = v2 + v3; o[3072 - 512] = v3 - v2;
    }
    if (MODE >= 2) __syncthreads();
  }
  if (tid == 0) cyc[0] = clock64() - t0;
  out[tid] = app + apq + aqq + sm[tid];
}
template <int MODE, int WORK>
void run(const char* name, int threads) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&cyc, 8);
  long long h = 0;
  const int n = 2000;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<MODE, WORK>), dim3(1), dim3(threads), 0, 0, out, cyc, n);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("%-72s %4d threads: %.0f cycles per trip\n", name, threads, (double)h / n);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0, 0>("chain only", 64);
  run<1, 0>("8 LDS reads -> chain -> 3 LDS writes", 64);
  run<2, 0>("... + barrier, the other waves idle", 64);
  run<2, 0>("... + barrier, the other waves idle", 256);
  run<2, 0>("... + barrier, the other waves idle", 512);
  run<2, 0>("... + barrier, the other waves idle", 768);
  run<2, 1>("... + barrier, the other waves: 4 reads, 4 writes each", 512);
  run<2, 1>("... + barrier, the other waves: 4 reads, 4 writes each", 768);
  return 0;
}
