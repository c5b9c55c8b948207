// probe: LDS atomic throughput on gfx950 -- ds_add_f64 / ds_add_u64 / ds_add_f32 / ds_add_u32, plain ds_write_b64, per CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_atomics.hip -o /tmp/lds_atomics ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE> __global__ void __launch_bounds__(512) k(double* out, int iters, int stride, int spread) {
  __shared__ double s[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = 0;
  __syncthreads();
  // every lane hits its own slot (conflict free when stride == 1), `spread` slots apart per iteration
  int idx = (threadIdx.x * stride) & 8191;
  if (stride < 0) idx = (int)(((threadIdx.x * 2654435761u) >> 7) % 171u) * (-stride);      // stride < 0: a random "pair" per lane, |stride| doubles per pair
  if (stride == -1000) idx = (int)((threadIdx.x * 37u) % 171u) * 43;                        // distinct pairs within a wave (37 is coprime to 171), residues mod 16 as they fall
  if (stride == -1001) { const int l = threadIdx.x & 63; idx = ((l & 15) * 3 + (l >> 4) * 48) % 171 * 43; }   // distinct pairs, 43 * pair mod 16 perfectly balanced (4 lanes per f64 bank pair)
  double v = 1.0 + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int a = (idx + u * spread) & 8191;
      if (MODE == 0) unsafeAtomicAdd(&s[a], v);
      else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long*>(&s[a]), (unsigned long long)__double_as_longlong(v));
      else if (MODE == 2) unsafeAtomicAdd(reinterpret_cast<float*>(&s[a]), (float)v);
      else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned*>(&s[a]), (unsigned)threadIdx.x);
      else if (MODE == 4) s[a] = v;
      else { double t = s[a]; s[a] = t + v; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[1] + s[77];
}
template <int MODE> void run(const char* name, int stride, int spread) {
  double* d; hipMalloc(&d, 4096 * sizeof(double));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, 10, stride, spread);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, iters, stride, spread);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ops = (double)blocks * 512 * iters * 16;          // lane-operations
  printf("%-22s stride %2d spread %4d: %8.3f ms  %7.2f lane-ops/clk/CU (2.4 GHz, 256 CUs)\n", name, stride, spread, ms, ops / (ms * 1e-3) / 2.4e9 / 256);
  hipFree(d);
}
int main() {
  run<0>("ds_add_f64", 1, 512); run<0>("ds_add_f64", 43, 1); run<0>("ds_add_f64 same-addr", 0, 1);
  run<0>("ds_add_f64 rand*43", -43, 1); run<0>("ds_add_f64 rand*42", -42, 1); run<0>("ds_add_f64 rand*47", -47, 1); run<5>("rmw b64 rand*43", -43, 1);
  run<0>("f64 distinct pairs", -1000, 1); run<0>("f64 distinct+balanced", -1001, 1);
  run<1>("ds_add_u64", 1, 512); run<1>("ds_add_u64", 43, 1);
  run<2>("ds_add_f32", 1, 512); run<3>("ds_add_u32", 1, 512);
  run<4>("ds_write_b64", 1, 512); run<5>("read+add+write b64", 1, 512);
  return 0;
}
