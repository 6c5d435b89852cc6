// Cost of the register-resident W update (rotate, then move the circle one seat on with whole-wave DPP shifts):
// cycles per round for one workgroup of 8 waves, NI components per thread, pieces switched off one at a time.
//   hipcc --offload-arch=gfx950 -O3 wpass.hip -o wpass && ./wpass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NI 9
__device__ __forceinline__ double lane_up(double v, double keep) {
  const long long b = __double_as_longlong(v), o = __double_as_longlong(keep);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)b, 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), 0x138, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_down(double v, double keep) {
  const long long b = __double_as_longlong(v), o = __double_as_longlong(keep);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)b, 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), 0x130, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int MODE>  // 0 full; 1 no shifts; 2 no selects; 3 rotation only from registers (no LDS read)
__global__ void __launch_bounds__(512) k(double* out, long long* cyc, int rounds, int half) {
  __shared__ double2 LOG[71 * 36 + 64];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 71 * 36 + 64; e += 512) LOG[e] = make_double2(0.8, 0.6);
  __syncthreads();
  double wx[NI], wy[NI];
  for (int t = 0; t < NI; ++t) {
    wx[t] = 1.0 + tid + t;
    wy[t] = 2.0 + tid - t;
  }
  const bool lane_is0 = lane == 0, lane_last = lane == half - 1;
  const long long t0 = clock64();
  for (int rr = 0; rr < rounds; ++rr) {
    const int round = rr % 71;
    double2 rk = MODE == 3 ? make_double2(0.8, 0.6) : LOG[round * half + lane];
    const double c = rk.x, s_ = rk.y;
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const double x = c * wx[t] - s_ * wy[t], y = s_ * wx[t] + c * wy[t];
      if (MODE == 1 || MODE == 3) {
        wx[t] = x;
        wy[t] = y;
      } else if (MODE == 2) {
        wx[t] = lane_down(x, x);
        wy[t] = lane_up(y, y);
      } else {
        const double xd = lane_down(x, x);
        const double yu = lane_up(lane_is0 ? x : y, y);
        wx[t] = lane_last ? y : xd;
        wy[t] = yu;
      }
    }
  }
  const long long t1 = clock64();
  double acc = 0.0;
  for (int t = 0; t < NI; ++t) acc += wx[t] + wy[t];
  out[blockIdx.x * 512 + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out;
  long long* cyc;
  const int nb = 1024;
  hipMalloc(&out, nb * 512 * sizeof(double));
  hipMalloc(&cyc, nb * sizeof(long long));
  const int rounds = 426;
  for (int blocks : {256, 512, 1024}) {
    for (int mode = 0; mode < 4; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, cyc, rounds, 36);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, cyc, rounds, 36);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, cyc, rounds, 36);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, out, cyc, rounds, 36);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long h;
      hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
      printf("blocks %4d mode %d: %.3f ms, block 0: %.0f clock64 ticks per round\n", blocks, mode, ms, (double)h / rounds);
    }
  }
  return 0;
}
