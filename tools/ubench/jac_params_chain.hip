// Latency of the Jacobi rotation-parameter chain (csrc/gpet_k_factor.inc: jac_params), one wave alone on its SIMD, each call's
// inputs depending on the previous call's outputs (as a round's parameters depend on the previous round's).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/jac_params_chain.hip -o gpurun_scratch/jac_params_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
// V0: the product's form (branches on the two "nothing to rotate" tests)
__device__ __forceinline__ void jp0(double app, double apq, double aqq, double& c, double& s) {
  c = 1.0; s = 0.0;
  if (fabs(apq) > 1e-300 && apq * apq > 1e-36 * fabs(app * aqq)) {
    const double d = aqq - app, hh = 2.0 * apq;
    const double rho2 = d * d + hh * hh;
    double y = __builtin_amdgcn_rsq(rho2);
    y = y * (1.5 - 0.5 * rho2 * y * y);
    const double den = fabs(d) + rho2 * y;
    double iv = __builtin_amdgcn_rcp(den);
    iv = iv * (2.0 - den * iv);
    const double t = (d >= 0.0 ? hh : -hh) * iv;
    const double u = 1.0 + t * t;
    c = __builtin_amdgcn_rsq(u);
    c = c * (1.5 - 0.5 * u * c * c);
    c = c * (1.5 - 0.5 * u * c * c);
    s = t * c;
  }
}
// V1: the same arithmetic, branchless (the tests select at the end)
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
// V2: two reciprocal square roots, no reciprocal: c^2 = (1 + |d| / rho) / 2, s = h sgn(d) / (2 rho c)
__device__ __forceinline__ void jp2(double app, double apq, double aqq, double& c, double& s) {
  const bool on = fabs(apq) > 1e-300 && apq * apq > 1e-36 * fabs(app * aqq);
  const double d = aqq - app, hh = 2.0 * apq;
  const double rho2 = d * d + hh * hh;
  double y = __builtin_amdgcn_rsq(rho2);
  y = y * (1.5 - 0.5 * rho2 * y * y);
  y = y * (1.5 - 0.5 * rho2 * y * y);
  const double c2 = 0.5 + 0.5 * fabs(d) * y;
  double z = __builtin_amdgcn_rsq(c2);
  z = z * (1.5 - 0.5 * c2 * z * z);
  z = z * (1.5 - 0.5 * c2 * z * z);
  const double w = (d >= 0.0 ? 0.5 : -0.5) * hh * y;
  c = on ? c2 * z : 1.0;
  s = on ? w * z : 0.0;
}
template <int V>
__global__ void k(double* out, long long* cyc, double seed) {
  double app = seed + threadIdx.x * 1e-3, apq = 0.3, aqq = 2.0 * seed;
  double c, s;
  long long t0 = clock64();
#pragma unroll 4
  for (int i = 0; i < REP; ++i) {
    if (V == 0) jp0(app, apq, aqq, c, s);
    if (V == 1) jp1(app, apq, aqq, c, s);
    if (V == 2) jp2(app, apq, aqq, c, s);
    // (the next call's inputs from this call's outputs: one dependent multiply-add each)
    app = __builtin_fma(c, 0.25, app);
    apq = __builtin_fma(s, 0.5, 0.1);
    aqq = __builtin_fma(s, 0.125, aqq);
  }
  long long t1 = clock64();
  out[threadIdx.x] = app + apq + aqq + c + s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V>
void run(const char* name) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
  long long h = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<V>), dim3(1), dim3(64), 0, 0, out, cyc, 3.0);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("%-60s %.1f cycles per call (+ one dependent fma)\n", name, (double)h / REP);
  (void)hipFree(out); (void)hipFree(cyc);
}
static void clock_probe();
int main() {
  run<0>("jac_params, product form (two branches)");
  run<1>("the same, branchless");
  run<2>("two rsq + four Newton steps, no rcp, branchless");
  clock_probe();
  return 0;
}
// effective clock of a kernel that is ONE wave on the whole GPU (the single-edge Jacobi's situation): cycles by clock64 against
// hipEvent time
__global__ void k_long(double* out, long long* cyc, int n) {
  double app = 3.0 + threadIdx.x * 1e-3, apq = 0.3, aqq = 6.0, c, s;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    jp1(app, apq, aqq, c, s);
    app = __builtin_fma(c, 0.25, app);
    apq = __builtin_fma(s, 0.5, 0.1);
    aqq = __builtin_fma(s, 0.125, aqq);
  }
  out[threadIdx.x] = app + apq + aqq;
  if (threadIdx.x == 0) cyc[0] = clock64() - t0;
}
static void clock_probe() {
  {
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int n : {1000, 100000}) {
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_long, dim3(1), dim3(64), 0, 0, out, cyc, n);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      long long h = 0; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("one wave, %d calls: %lld clock64 ticks in %.3f ms -> %.2f GHz if a tick is a shader cycle\n", n, h, ms, (double)h / ms * 1e-6);
    }
  }
}
