// cms_area_kernels.hip -- Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cpp:158-176, 728-744) and Frame::GetFeaturesInArea
// (src/Frame.cpp:36-72, 251-716) on the device: the key points of a frame never leave HBM between extraction and matching.
//
//   k_area_grid    one workgroup per frame: every key point gets the key (cell << 12 | index); a rank sort in LDS (keys are
//                  unique, n <= 4095) lists the indices cell-major with ascending index inside a cell -- the order the
//                  reference's per-cell vectors have -- and a start offset is written for each of the 5 x 50 x 50 cells.
//   k_area_query   one thread per query: cms_area_rects() (the reference's 41 unfolding cases as a table, cms_area_table.h)
//                  yields up to three cell rectangles; the thread walks them exactly like AddCells (ix outer, iy inner, level
//                  and canvas-distance test) and either counts its candidates (pass 0) or writes them (pass 1).
//   k_area_scan    exclusive scan of the counts -> CSR offsets.
// The CSR lists feed k_hamming_best2 directly; candidate order (which decides Hamming ties) equals the reference's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cms_types.h"
#include "cms_area_table.h"

#define CMS_AREA_CELLS (5 * CMS_AREA_G * CMS_AREA_G)
#define CMS_AREA_MAXKP 4095

extern "C" __global__ void __launch_bounds__(1024)
k_area_grid(const CmsKeyPoint* __restrict__ kps, const int* __restrict__ kp_cnt, int kp_cap, int F, float inv,
            uint16_t* __restrict__ sorted_idx, int* __restrict__ cell_start, int* __restrict__ n_valid_out) {
  __shared__ uint32_t keys[CMS_AREA_MAXKP + 1];
  __shared__ uint32_t sorted[CMS_AREA_MAXKP + 1];
  __shared__ int s_nvalid;
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const int n = min(kp_cnt[b], CMS_AREA_MAXKP);
  const CmsKeyPoint* kp = kps + (size_t)b * kp_cap;
  if (tid == 0) s_nvalid = 0;
  __syncthreads();
  for (int i = tid; i < n; i += T) {
    const float x = kp[i].x, y = kp[i].y;
    const double fi = (double)(x / (float)F), fj = (double)(y / (float)F);     // FaceInCubemap(cv::Point2f) (CamModelGeneral.h:445-456)
    int f = -1;
    if (fi >= 0 && fi < 1 && fj >= 1 && fj < 2) f = 1;
    else if (fi >= 1 && fi < 2 && fj >= 0 && fj < 1) f = 3;
    else if (fi >= 1 && fi < 2 && fj >= 1 && fj < 2) f = 0;
    else if (fi >= 1 && fi < 2 && fj >= 2 && fj < 3) f = 4;
    else if (fi >= 2 && fi < 3 && fj >= 1 && fj < 2) f = 2;
    uint32_t key = 0xFFFFF000u | (uint32_t)i;                                   // not on a face: sorted behind every cell
    if (f >= 0) {
      const int px = (int)(x * inv) % CMS_AREA_G, py = (int)(y * inv) % CMS_AREA_G;     // PosInGrid (Frame.cpp:734-742), mnMinX = 0
      key = ((uint32_t)((f * CMS_AREA_G + px) * CMS_AREA_G + py) << 12) | (uint32_t)i;
      atomicAdd(&s_nvalid, 1);
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int i = tid; i < n; i += T) {
    const uint32_t k = keys[i];
    int r = 0;
    for (int q = 0; q < n; ++q) r += keys[q] < k ? 1 : 0;
    sorted[r] = k;
  }
  __syncthreads();
  const int nv = s_nvalid;
  uint16_t* si = sorted_idx + (size_t)b * kp_cap;
  int* cs = cell_start + (size_t)b * (CMS_AREA_CELLS + 1);
  for (int s = tid; s < nv; s += T) {
    const int c = (int)(sorted[s] >> 12), prev = s ? (int)(sorted[s - 1] >> 12) : -1;
    si[s] = (uint16_t)(sorted[s] & 0xFFFu);
    for (int cc = prev + 1; cc <= c; ++cc) cs[cc] = s;
  }
  const int last = nv ? (int)(sorted[nv - 1] >> 12) : -1;
  for (int cc = last + 1 + tid; cc <= CMS_AREA_CELLS; cc += T) cs[cc] = nv;
  if (tid == 0) n_valid_out[b] = nv;
}

struct CmsAreaArgs {
  const CmsKeyPoint* kp;        // the frame's key points
  const uint16_t* sorted_idx;   // cell-major index list of the frame
  const int* cell_start;        // CMS_AREA_CELLS + 1 offsets into sorted_idx
  const float* qx; const float* qy; const float* qr; const int* qmin; const int* qmax;
  int nq, F; float inv;
  int* cnt;                     // pass 0: candidates per query
  const int* off;               // pass 1: CSR offsets
  int* idx; int cap; int idx_base;
};

extern "C" __global__ void __launch_bounds__(64) k_area_query(CmsAreaArgs a, int pass) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.nq) return;
  const float x = a.qx[q], y = a.qy[q], r = a.qr[q];
  const int minLevel = a.qmin[q], maxLevel = a.qmax[q];
  const bool check = (minLevel > 0) || (maxLevel >= 0);
  CmsAreaRectI rc[3];
  const int nr = cms_area_rects(x, y, r, a.F, a.inv, rc);
  int n = 0;
  const int base = pass ? a.off[q] : 0;
  for (int k = 0; k < nr; ++k) {
    const int x0 = max(0, rc[k].x0), x1 = min(CMS_AREA_G - 1, rc[k].x1), y0 = max(0, rc[k].y0), y1 = min(CMS_AREA_G - 1, rc[k].y1);   // AddCells' clamp
    for (int ix = x0; ix <= x1; ++ix) {
      if (y0 > y1) break;
      // the cells (ix, y0 .. y1) of a face are consecutive in the cell-major list: one range per ix
      const int c0 = (rc[k].face * CMS_AREA_G + ix) * CMS_AREA_G + y0;
      const int s0 = a.cell_start[c0], s1 = a.cell_start[c0 + (y1 - y0) + 1];
      for (int s = s0; s < s1; ++s) {
        const int j = a.sorted_idx[s];
        const CmsKeyPoint p = a.kp[j];
        if (check) {
          if (p.octave < minLevel) continue;
          if (maxLevel >= 0 && p.octave > maxLevel) continue;
        }
        if (fabsf(p.x - x) < r && fabsf(p.y - y) < r) {
          if (pass && base + n < a.cap) a.idx[base + n] = a.idx_base + j;
          ++n;
        }
      }
    }
  }
  if (!pass) a.cnt[q] = n;
}

// off[0] = 0, off[q + 1] = sum cnt[0..q]; *total = off[nq]
extern "C" __global__ void __launch_bounds__(1024) k_area_scan(const int* __restrict__ cnt, int nq, int* __restrict__ off, int* __restrict__ total) {
  __shared__ int part[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) { carry = 0; off[0] = 0; }
  __syncthreads();
  for (int base = 0; base < nq; base += 1024) {
    const int i = base + tid;
    const int v = i < nq ? cnt[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    int wbase = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { const int pv = part[w]; if (w < wv) wbase += pv; tot += pv; }
    if (i < nq) off[i + 1] = carry + wbase + incl;
    __syncthreads();
    if (tid == 0) carry += tot;
    __syncthreads();
  }
  if (tid == 0 && total) *total = carry;
}
