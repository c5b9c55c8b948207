// probe: what v_mov_b32_dpp wave_shl:1 / wave_shr:1 move on gfx950 (whole-wavefront shifts, GFX9 only)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int v = 100 + threadIdx.x;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false);        // wave_shl:1
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);   // wave_shr:1
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xF, 0xF, false);  // wave_rol:1
}
int main() {
  int* d; hipMalloc(&d, 192 * 4); int h[192];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const int L[] = {0, 1, 2, 14, 15, 16, 17, 31, 32, 33, 62, 63};
  for (int t = 0; t < 3; ++t) { printf("%s:", t == 0 ? "wave_shl:1" : t == 1 ? "wave_shr:1" : "wave_rol:1"); for (int l : L) printf(" [%d]=%d", l, h[64 * t + l]); printf("\n"); }
  return 0;
}
