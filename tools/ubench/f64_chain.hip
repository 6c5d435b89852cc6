// Dependent-chain latency of the f64 instructions the Jacobi's rotation parameters are made of, one wave alone on its SIMD
// (clock64 around 256 dependent instructions), and the issue cost of independent ones (4 chains interleaved).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_chain.hip -o /tmp/f64_chain && /tmp/f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int OP, int CH>
__global__ void k(double* out, long long* cyc, double seed) {
  double x[CH];
  for (int c = 0; c < CH; ++c) x[c] = seed + threadIdx.x * 1e-3 + c;
  const double a = 1.0000001, b = 1e-9;
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < REP; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (OP == 0) x[c] = __builtin_fma(x[c], a, b);
      if (OP == 1) x[c] = x[c] * a;
      if (OP == 2) x[c] = x[c] + b;
      if (OP == 3) x[c] = __builtin_amdgcn_rsq(x[c]) + 2.0;   // rsq + add
      if (OP == 4) x[c] = __builtin_amdgcn_rcp(x[c]) + 2.0;   // rcp + add
      if (OP == 5) { float f = (float)x[c]; f = __builtin_amdgcn_rsqf(f); x[c] = (double)f + 2.0; }  // cvt, rsq_f32, cvt, add
      if (OP == 6) x[c] = __builtin_amdgcn_rsq(x[c]);  // rsq alone (converges to 1)
    }
    asm volatile("" : "+v"(x[0]));
  }
  long long t1 = clock64();
  double s = 0.0;
  for (int c = 0; c < CH; ++c) s += x[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP, int CH>
void run(const char* name) {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  long long h = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<OP, CH>), dim3(1), dim3(64), 0, 0, out, cyc, 3.0);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("%-34s chains %d: %.1f cycles per step (%.1f per instruction group)\n", name, CH, (double)h / REP, (double)h / REP / CH);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 1>("v_fma_f64 dependent"); run<0, 4>("v_fma_f64 x4 independent");
  run<1, 1>("v_mul_f64 dependent"); run<2, 1>("v_add_f64 dependent");
  run<3, 1>("v_rsq_f64 + v_add_f64 dependent"); run<3, 4>("v_rsq_f64 + add x4");
  run<4, 1>("v_rcp_f64 + v_add_f64 dependent");
  run<5, 1>("cvt + v_rsq_f32 + cvt + add dep");
  run<6, 1>("v_rsq_f64 dependent");
  return 0;
}
