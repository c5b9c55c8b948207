// Probe: do v_mfma_f64_16x16x4_f64 and v_fma_f64 of DIFFERENT wavefronts of one SIMD overlap on gfx950, or do they share the FP64 datapath?
// (The FP64 matrix rate of the chip equals its FP64 vector rate; the Schur kernel of the local BA keeps both kinds of instruction busy.)
// Workgroups of 512 threads, one per CU: wavefronts 0-3 and 4-7 land pairwise on the four SIMDs.
//   mode 0: all eight wavefronts run a chain-free stream of v_fma_f64                (vector only, 2 wavefronts per SIMD)
//   mode 1: all eight run v_mfma_f64_16x16x4_f64 on four independent accumulators    (matrix only)
//   mode 2: wavefronts 0-3 vector, 4-7 matrix, the same instruction counts as above  (one of each per SIMD)
//   mode 3 / 4: only wavefronts 0-3 work (vector / matrix): what one wavefront per SIMD reaches alone
// If mode 2 takes about max(mode 3, mode 4) the pipes overlap; if it takes their sum they do not.
// Also: issue cost of v_mov_b32 DPP and of ds_read_b64 streams next to the FMAs (modes 5, 6).
// hipcc --offload-arch=gfx950 -O3 f64_pipes.hip -o /tmp/f64p && /tmp/f64p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define NV 4096   /* vector FMAs per wavefront and repetition: 16 independent chains */
#define NM 256    /* MFMAs per wavefront and repetition: 4 independent accumulators */
__device__ __forceinline__ double vec_stream(double x, int n) {
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = x + i;
  for (int it = 0; it < n / 16; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = __builtin_fma(a[i], 1.0000001, 0.5);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  return s;
}
__device__ __forceinline__ double mat_stream(double x, int n) {
  v4d c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = (v4d){x, x, x, x};
  for (int it = 0; it < n / 4; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1.0 + i, c[i], 0, 0, 0);
  }
  return c[0][0] + c[1][1] + c[2][2] + c[3][3];
}
__device__ __forceinline__ double dpp_stream(double x, int n) {      // 16 chains of (two DPP moves + one add): the quad exchange of the Schur kernel
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = x + i;
  for (int it = 0; it < n / 8; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int lo = __builtin_amdgcn_mov_dpp(__double2loint(a[i]), 0x55, 0xF, 0xF, true);
      const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(a[i]), 0x55, 0xF, 0xF, true);
      a[i] += __hiloint2double(hi, lo);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  return s;
}
__global__ void __launch_bounds__(512) k_probe(double* out, int mode, int rep) {
  const int wave = threadIdx.x >> 6;
  double x = 1.0 + 1e-9 * threadIdx.x, s = 0;
  for (int r = 0; r < rep; ++r) {
    if (mode == 0) s += vec_stream(x + r, NV);
    else if (mode == 1) s += mat_stream(x + r, NM);
    else if (mode == 2) s += wave < 4 ? vec_stream(x + r, NV) : mat_stream(x + r, NM);
    else if (mode == 3) { if (wave < 4) s += vec_stream(x + r, NV); }
    else if (mode == 4) { if (wave < 4) s += mat_stream(x + r, NM); }
    else if (mode == 5) { if (wave < 4) s += dpp_stream(x + r, NV / 4); }
    else if (mode == 6) s += wave < 4 ? dpp_stream(x + r, NV / 4) : mat_stream(x + r, NM);
  }
  if (s == 123.456) out[threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"vector FMA x 2 wavefronts / SIMD", "MFMA f64 x 2 wavefronts / SIMD", "1 vector + 1 MFMA wavefront / SIMD", "vector FMA, 1 wavefront / SIMD",
                         "MFMA f64, 1 wavefront / SIMD", "DPP exchange, 1 wavefront / SIMD", "1 DPP + 1 MFMA wavefront / SIMD"};
  const int rep = 8;
  for (int mode = 0; mode < 7; ++mode) {
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
      hipEventRecord(a);
      hipLaunchKernelGGL(k_probe, dim3(256), dim3(512), 0, 0, d, mode, rep);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (it > 0 && ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * 2.4e9 / rep;      // cycles per repetition at 2.4 GHz
    printf("mode %d  %-36s %8.1f us   %8.0f cycles per repetition (%d FMAs = %d issue cycles; %d MFMAs)\n", mode, names[mode], 1e3 * best, cyc, NV, NV * 4, NM);
  }
  return 0;
}
