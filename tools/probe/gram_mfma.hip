// tools/probe/gram_mfma.hip -- developer probe behind DESIGN.md's "no MFMA for the pose Gram" decision.
// The per-pose block of the local BA is the 7x7 Gram of [J sqrt(w) | r sqrt(w)] over the pose's residual rows (K = 2 x edges).  This
// program builds many such Grams from rows already in memory two ways and times both on the device:
//   vec   one lane per row, 28 unique products in FP64 FMAs, wave reduce-scatter at the end (the shape of k_ba_lin_poses)
//   mfma  v_mfma_f64_16x16x4_f64 on Xt X with the 7 columns padded to 16 (4 rows per instruction, (7/16)^2 = 19 % of the tile useful)
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 tools/probe/gram_mfma.hip -o /tmp/gram_mfma && /tmp/gram_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double v4d __attribute__((ext_vector_type(4)));
#define ROWS_PER_GRAM 8192        /* 4096 edges of one pose */

__global__ void __launch_bounds__(256) k_vec(const double* __restrict__ X, int ngram, double* __restrict__ out) {
  __shared__ double sh[4][28];
  const int g = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double* x = X + (size_t)g * ROWS_PER_GRAM * 8;
  double acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = 0;
  for (int r = threadIdx.x; r < ROWS_PER_GRAM; r += 256) {
    double v[7];
    const double2* p = reinterpret_cast<const double2*>(x + 8 * (size_t)r);
    const double2 a = p[0], b = p[1], c = p[2], d = p[3];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = d.x;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = i; j < 7; ++j) { acc[k] = __builtin_fma(v[i], v[j], acc[k]); ++k; }
  }
#pragma unroll
  for (int i = 0; i < 28; ++i) {
    double s = acc[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) sh[wv][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 28) out[(size_t)g * 28 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__global__ void __launch_bounds__(256) k_mfma(const double* __restrict__ X, int ngram, double* __restrict__ out) {
  __shared__ double sh[4][49];
  const int g = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double* x = X + (size_t)g * ROWS_PER_GRAM * 8;
  const int col = lane & 15, kr = lane >> 4;                  // this lane supplies X[row0 + kr][col] as both operands (A = Xt, B = X)
  v4d c = {0, 0, 0, 0};
  for (int r0 = wv * 4; r0 < ROWS_PER_GRAM; r0 += 16) {       // 4 waves x 4 rows per step
    const double v = col < 7 ? x[8 * (size_t)(r0 + kr) + col] : 0.0;
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, c, 0, 0, 0);
  }
  // D[lane / 16 + 4 * i][lane % 16] = c[i]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = kr + 4 * i;
    if (m < 7 && col < 7) sh[wv][7 * m + col] = c[i];
  }
  __syncthreads();
  if (threadIdx.x < 28) {
    int k = threadIdx.x, i = 0;
    while (k >= 7 - i) { k -= 7 - i; ++i; }
    const int j = i + k;
    out[(size_t)g * 28 + threadIdx.x] = sh[0][7 * i + j] + sh[1][7 * i + j] + sh[2][7 * i + j] + sh[3][7 * i + j];
  }
}

int main() {
  const int ngram = 19 * 64;                                   // 19 free poses x 64 windows
  const size_t n = (size_t)ngram * ROWS_PER_GRAM * 8;
  std::vector<double> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) h[i] = (i & 7) == 7 ? 0.0 : (rand() / (double)RAND_MAX - 0.5);
  double *dX, *o1, *o2;
  hipMalloc(&dX, n * 8); hipMalloc(&o1, (size_t)ngram * 28 * 8); hipMalloc(&o2, (size_t)ngram * 28 * 8);
  hipMemcpy(dX, h.data(), n * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2] = {0, 0};
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 10; ++it) {
        if (which == 0) hipLaunchKernelGGL(k_vec, dim3(ngram), dim3(256), 0, 0, dX, ngram, o1);
        else hipLaunchKernelGGL(k_mfma, dim3(ngram), dim3(256), 0, 0, dX, ngram, o2);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[which], e0, e1);
    }
  }
  std::vector<double> a((size_t)ngram * 28), b((size_t)ngram * 28);
  hipMemcpy(a.data(), o1, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (size_t i = 0; i < a.size(); ++i) worst = fmax(worst, fabs(a[i] - b[i]) / (fabs(a[i]) + 1e-9));
  if (getenv("GRAM_DUMP")) for (int i = 0; i < 28; ++i) printf("%d vec %.6f mfma %.6f\n", i, a[i], b[i]);
  const double bytes = (double)n * 8, rows = (double)ngram * ROWS_PER_GRAM;
  printf("Gram of %d x %d rows x 7 (%.0f MB): vec %.1f us (%.0f GB/s, %.2f GFMA/s useful)   mfma_f64_16x16x4 %.1f us (%.0f GB/s)   max rel diff %.2e\n",
         ngram, ROWS_PER_GRAM, bytes / 1e6, 100 * ms[0], bytes / (ms[0] * 1e-4) / 1e9, rows * 28 / (ms[0] * 1e-4) / 1e9, 100 * ms[1],
         bytes / (ms[1] * 1e-4) / 1e9, worst);
  return 0;
}
