// Hardware probe: does gfx950 serve unaligned 16-bit LDS reads (ds_read_u16 at odd byte addresses)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint16_t __attribute__((aligned(1))) u16u;
__global__ void k(uint32_t* out) {
  __shared__ __align__(16) uint8_t buf[256];
  for (int i = threadIdx.x; i < 256; i += 64) buf[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const int o = threadIdx.x * 3 + 1;   // mix of odd and even offsets
  uint32_t v;
  asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(buf + o) ) : "memory");
  out[threadIdx.x] = v;
}
int main() {
  uint32_t* d; hipMalloc(&d, 64 * 4);
  k<<<1, 64>>>(d);
  uint32_t h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) { int o = t * 3 + 1; uint32_t want = (uint8_t)(o * 7 + 3) | ((uint32_t)(uint8_t)((o + 1) * 7 + 3) << 8); if (h[t] != want) ++bad; }
  printf("lds unaligned u16: %s (%d mismatches)\n", bad ? "NOT SUPPORTED" : "ok", bad);
  return 0;
}
