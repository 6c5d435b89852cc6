// Do the whole-wave DPP shifts exist on gfx950, and which way do they move data?
//   hipcc --offload-arch=gfx950 -O3 dpp_shift.hip -o dpp_shift && ./dpp_shift
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  const int v = 100 + lane;
  // old = -1: what a lane without a source keeps (bound_ctrl = false)
  out[lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xf, 0xf, false);  // wave_ror:1
  out[192 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);  // wave_rol:1
}
int main() {
  int* d;
  hipMalloc(&d, 256 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"wave_shr1", "wave_shl1", "wave_ror1", "wave_rol1"};
  for (int t = 0; t < 4; ++t) {
    printf("%s:", names[t]);
    for (int l : {0, 1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 62, 63}) printf(" [%d]=%d", l, h[64 * t + l]);
    printf("\n");
  }
  return 0;
}
