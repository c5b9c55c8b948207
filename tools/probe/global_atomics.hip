// Probe: cost of summing the range slices of the reduced system with global FP64 atomics instead of writing slices.
// 256 workgroups x 512 threads; workgroup (window w = b / 16, slice r = b % 16) adds (or stores) 6512 doubles into window w's copy
// (atomic) or into its own slice (store).  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics global_atomics.hip -o /tmp/ga && /tmp/ga
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 6512
__global__ void __launch_bounds__(512) k_store(double* out, int rep) {
  double* o = out + (size_t)blockIdx.x * N;
  for (int r = 0; r < rep; ++r)
    for (int i = threadIdx.x; i < N; i += 512) o[i] = (double)(i + r);
}
__global__ void __launch_bounds__(512) k_atomic(double* out, int rep, int rot) {
  double* o = out + (size_t)(blockIdx.x / 16) * N;
  const int sh = rot ? (blockIdx.x % 16) * (N / 16) : 0;       // rot: the 16 slices of a window start at different offsets
  for (int r = 0; r < rep; ++r)
    for (int i = threadIdx.x; i < N; i += 512) { int j = i + sh; if (j >= N) j -= N; unsafeAtomicAdd(o + j, (double)(i + r)); }
}
__global__ void __launch_bounds__(512) k_atomic_safe(double* out, int rep) {
  double* o = out + (size_t)(blockIdx.x / 16) * N;
  for (int r = 0; r < rep; ++r)
    for (int i = threadIdx.x; i < N; i += 512) atomicAdd(o + i, (double)(i + r));
}
int main() {
  double* d; hipMalloc(&d, (size_t)256 * N * 8); hipMemset(d, 0, (size_t)256 * N * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int which = 0; which < 4; ++which) {
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(a);
      for (int l = 0; l < 20; ++l) {
        if (which == 0) hipLaunchKernelGGL(k_store, dim3(256), dim3(512), 0, 0, d, 1);
        if (which == 1) hipLaunchKernelGGL(k_atomic, dim3(256), dim3(512), 0, 0, d, 1, 0);
        if (which == 2) hipLaunchKernelGGL(k_atomic, dim3(256), dim3(512), 0, 0, d, 1, 1);
        if (which == 3) hipLaunchKernelGGL(k_atomic_safe, dim3(256), dim3(512), 0, 0, d, 1);
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (it == 2) printf("%-28s %7.2f us per launch (256 workgroups x 6512 doubles)\n", which == 0 ? "plain stores to own slice" : which == 1 ? "unsafeAtomicAdd f64" : which == 2 ? "unsafeAtomicAdd f64 rotated" : "atomicAdd f64", 1e3 * ms / 20);
    }
  }
  double h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost); printf("check %g %g\n", h[0], h[1]);
  return 0;
}
