// Cost of moving a double one lane along the wave (what the rotation-log pass and the register rows of the Jacobi do every
// round): dependent chain  x = shift(x) + b, one wave alone on its SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lane_shift.hip -o gpurun_scratch/lane_shift
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int CTRL>
__device__ __forceinline__ double dpp2(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int OP>
__global__ void k(double* out, long long* cyc) {
  double x = 1.0 + threadIdx.x * 1e-3;
  const int lane = threadIdx.x;
  asm volatile("" : "+v"(x));
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    double y;
    if (OP == 0) y = dpp2<0x130>(x);       // wave_shl:1
    if (OP == 1) y = dpp2<0x138>(x);       // wave_shr:1
    if (OP == 2) y = dpp2<0x101>(x);       // row_shl:1
    if (OP == 3) y = dpp2<0x111>(x);       // row_shr:1
    if (OP == 4) y = __shfl_down(x, 1, 64);  // ds_bpermute x 2
    if (OP == 5) {                          // row_shr:1 + the row boundaries by v_readlane (three lanes 15, 31, 47 into lanes 16, 32, 48)
      y = dpp2<0x111>(x);
      const long long b = __double_as_longlong(x);
#pragma unroll
      for (int r = 1; r < 4; ++r) {
        const int lo = __builtin_amdgcn_readlane((int)b, 16 * r - 1), hi = __builtin_amdgcn_readlane((int)(b >> 32), 16 * r - 1);
        const double e = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        y = lane == 16 * r ? e : y;
      }
    }
    if (OP == 6) y = x;                     // nothing (the add alone)
    x = y + 1e-9;
    asm volatile("" : "+v"(x));
  }
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP>
void run(const char* name) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
  long long h = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<OP>), dim3(1), dim3(64), 0, 0, out, cyc);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("%-64s %.1f cycles per step\n", name, (double)h / REP);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<6>("v_add_f64 alone");
  run<0>("2 x v_mov_b32_dpp wave_shl:1 + add");
  run<1>("2 x v_mov_b32_dpp wave_shr:1 + add");
  run<2>("2 x v_mov_b32_dpp row_shl:1 + add");
  run<3>("2 x v_mov_b32_dpp row_shr:1 + add");
  run<4>("__shfl_down (2 x ds_bpermute_b32) + add");
  run<5>("row_shr:1 + 3 x (2 v_readlane + select) + add");
  return 0;
}
