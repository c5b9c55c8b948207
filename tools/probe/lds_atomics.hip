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
  // which modulus matters?  pair p sits at double 43 p: 43 = 11 (mod 32), 11 * 3 = 1 (mod 32), 43 * 3 = 1 (mod 64)
  if (stride == -1002) { const int l = threadIdx.x & 63; idx = ((3 * (l & 15)) % 32 + 32 * (l >> 4)) * 43; }   // balanced mod 16 (4 per class), only half of the classes mod 32
  if (stride == -1003) { const int l = threadIdx.x & 63; idx = ((3 * (l & 31)) % 32 + 32 * (l >> 5)) * 43; }   // balanced mod 32 (2 per class)
  if (stride == -1004) { const int l = threadIdx.x & 63; idx = ((3 * l) % 64) * 43; }                           // all 64 classes mod 64 distinct
  if (stride == -1005) { const int l = threadIdx.x & 63; idx = ((3 * (l & 7)) % 8 + 8 * (l >> 3)) * 43; }        // balanced mod 8 (8 per class): classes mod 16 half used
  // which lanes compete?  stride = -(2000 + x): lanes l and l ^ x get the same double modulo 64 (different addresses), all others differ
  if (stride <= -2000 && stride > -2100) { const int l = threadIdx.x & 63, x = -stride - 2000; idx = (l & ~x) + 64 * ((l & x) ? 1 : 0) + 128 * (threadIdx.x >> 6); }
  // stride = -(2100 + m): lane l at double l * m (which strides are conflict free?)
  if (stride <= -2100 && stride > -2300) { const int l = threadIdx.x & 63, m = -stride - 2100; idx = l * m + (threadIdx.x >> 6) * 3; }
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
  run<0>("f64 bal16 half32", -1002, 1); run<0>("f64 bal32", -1003, 1); run<0>("f64 bal64", -1004, 1); run<0>("f64 bal8", -1005, 1);
  for (int x : {0, 1, 2, 4, 8, 16, 32, 3, 7, 15}) run<0>("f64 l,l^x same bank", -2000 - x, 1);
  for (int m : {1, 2, 4, 8, 16, 32, 64, 3, 5, 6, 12, 24, 48}) run<0>("f64 lane*m", -2100 - m, 1);
  run<1>("ds_add_u64", 1, 512); run<1>("ds_add_u64", 43, 1);
  run<2>("ds_add_f32", 1, 512); run<3>("ds_add_u32", 1, 512);
  run<4>("ds_write_b64", 1, 512); run<5>("read+add+write b64", 1, 512);
  return 0;
}
