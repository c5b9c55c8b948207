// cms_area_kernels.hip -- Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cpp:158-176, 728-744) and Frame::GetFeaturesInArea
// (src/Frame.cpp:36-72, 251-716) on the device: the key points of a frame never leave HBM between extraction and matching.
//
//   k_area_grid    one workgroup per frame: every key point gets the key (cell << 14 | index); a rank sort in LDS (keys are
//                  unique, n <= 16383: the 3 x nFeatures extractor of the initialisation included) lists the indices cell-major with
//                  ascending index inside a cell -- the order the
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
#define CMS_AREA_MAXKP 16383

extern "C" __global__ void __launch_bounds__(1024)
k_area_grid(const CmsKeyPoint* __restrict__ kps, const int* __restrict__ kp_cnt, int kp_cap, int F, float inv,
            uint16_t* __restrict__ sorted_idx, int* __restrict__ cell_start, int* __restrict__ n_valid_out) {
  extern __shared__ uint32_t area_lds[];                 // keys [kp_cap + 1] | sorted [kp_cap + 1]
  uint32_t* keys = area_lds;
  uint32_t* sorted = area_lds + (kp_cap + 1);
  __shared__ int s_nvalid;
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const int n = min(kp_cnt[b], min(kp_cap, CMS_AREA_MAXKP));      // the LDS arrays hold kp_cap + 1 entries each
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
    uint32_t key = 0xFFFFC000u | (uint32_t)i;                                   // not on a face: sorted behind every cell
    if (f >= 0) {
      const int px = (int)(x * inv) % CMS_AREA_G, py = (int)(y * inv) % CMS_AREA_G;     // PosInGrid (Frame.cpp:734-742), mnMinX = 0
      key = ((uint32_t)((f * CMS_AREA_G + px) * CMS_AREA_G + py) << 14) | (uint32_t)i;
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
    const int c = (int)(sorted[s] >> 14), prev = s ? (int)(sorted[s - 1] >> 14) : -1;
    si[s] = (uint16_t)(sorted[s] & 0x3FFFu);
    for (int cc = prev + 1; cc <= c; ++cc) cs[cc] = s;
  }
  const int last = nv ? (int)(sorted[nv - 1] >> 14) : -1;
  for (int cc = last + 1 + tid; cc <= CMS_AREA_CELLS; cc += T) cs[cc] = nv;
  if (tid == 0) n_valid_out[b] = nv;
}

struct CmsAreaArgs {
  const CmsKeyPoint* kp;        // the frame's key points
  const uint16_t* sorted_idx;   // cell-major index list of the frame
  const int* cell_start;        // CMS_AREA_CELLS + 1 offsets into sorted_idx
  const float* qx; const float* qy; const float* qr; const int* qmin; const int* qmax;
  const int* q_frame;           // optional: frame of the batch a query addresses (nullptr: all queries address `kp`'s frame)
  int kp_cap;                   // frame stride of kp / sorted_idx (cell_start: CMS_AREA_CELLS + 1) when q_frame is given
  int nq, F; float inv;
  int* cnt;                     // pass 0: candidates per query
  const int* off;               // pass 1: CSR offsets
  int* idx; int cap; int idx_base;
  int* tmp;                     // optional: nq x CMS_AREA_TMP -- pass 0 leaves the first hits of every query here (final values, in order) and pass 1
                                // only copies them for the queries that have no more than that (nearly all: ~2 candidates per window on average)
};
#define CMS_AREA_TMP 8

// Eight lanes per query (eight queries per wavefront): the lanes of a group take the cell columns ix of a rectangle in turn, so the
// dependent loads of a query (cell offsets -> index list -> key point) run eight wide; a group-wide prefix sum of the per-column
// hit counts keeps the output in AddCells' order (ix outer, iy inner, index order inside a cell).
#define CMS_AREA_QL 8
extern "C" __global__ void __launch_bounds__(256) k_area_query(CmsAreaArgs a, int pass) {
  const int gl = threadIdx.x & (CMS_AREA_QL - 1);
  const int q = (blockIdx.x * blockDim.x + threadIdx.x) / CMS_AREA_QL;
  const int qq = q < a.nq ? q : 0;
  const float x = a.qx[qq], y = a.qy[qq], r = a.qr[qq];
  const bool live = q < a.nq && !(r < 0.0f);            // r < 0: "no window" (a map point outside the frustum), empty list
  const int minLevel = a.qmin[qq], maxLevel = a.qmax[qq];
  const bool check = (minLevel > 0) || (maxLevel >= 0);
  CmsAreaRectI rc[3];
  const int base = (pass && live) ? a.off[qq] : 0;
  bool copied = false;
  if (pass && a.tmp && live) {                             // the search of pass 0 kept this query's hits: copy, no second search
    const int c = a.cnt[qq];
    if (c <= CMS_AREA_TMP) {
      if (gl < c && base + gl < a.cap) a.idx[base + gl] = a.tmp[(size_t)qq * CMS_AREA_TMP + gl];
      copied = true;
    }
  }
  const int nr = (live && !copied) ? cms_area_rects(x, y, r, a.F, a.inv, rc) : 0;
  int n = 0;                                               // hits of the whole query so far (same in all lanes of the group)
  int* tmpq = (!pass && a.tmp) ? a.tmp + (size_t)qq * CMS_AREA_TMP : nullptr;
  const int fr = a.q_frame ? a.q_frame[qq] : 0;
  const CmsKeyPoint* kp = a.kp + (size_t)fr * a.kp_cap;
  const uint16_t* sorted_idx = a.sorted_idx + (size_t)fr * a.kp_cap;
  const int* cell_start = a.cell_start + (size_t)fr * (CMS_AREA_CELLS + 1);
  const int idx_base = a.idx_base + (a.q_frame ? fr * a.kp_cap : 0);
  // the groups of a wavefront walk different rectangle shapes: loop bounds are made group-uniform via shuffles inside the group
  for (int k = 0; k < 3; ++k) {
    const bool has = k < nr;
    const int x0 = has ? max(0, rc[k].x0) : 0, x1 = has ? min(CMS_AREA_G - 1, rc[k].x1) : -1;
    const int y0 = has ? max(0, rc[k].y0) : 0, y1 = has ? min(CMS_AREA_G - 1, rc[k].y1) : -1;        // AddCells' clamp
    const int face = has ? rc[k].face : 0;
    const int ncol = (x1 >= x0 && y1 >= y0) ? x1 - x0 + 1 : 0;
    // every group of the wave iterates max-over-wave column chunks; idle groups just carry zeros through the shuffles
    int maxcol = ncol;
    for (int o = 32; o > 0; o >>= 1) maxcol = max(maxcol, __shfl_xor(maxcol, o));
    for (int cb = 0; cb < maxcol; cb += CMS_AREA_QL) {
      const int ix = x0 + cb + gl;
      const bool col = cb + gl < ncol;
      int s0 = 0, s1 = 0;
      if (col) {
        const int c0 = (face * CMS_AREA_G + ix) * CMS_AREA_G + y0;      // cells (ix, y0 .. y1) are consecutive in the list
        s0 = cell_start[c0]; s1 = cell_start[c0 + (y1 - y0) + 1];
      }
      // this lane's hits in its column; the first four are kept (a column rarely holds more)
      int h0 = 0, h1 = 0, h2 = 0, h3 = 0;                  // registers, not an indexed array (that would live in scratch)
      int nh = 0;
      for (int s = s0; s < s1; ++s) {
        const int j = sorted_idx[s];
        const CmsKeyPoint p = kp[j];
        if (check) {
          if (p.octave < minLevel) continue;
          if (maxLevel >= 0 && p.octave > maxLevel) continue;
        }
        if (fabsf(p.x - x) < r && fabsf(p.y - y) < r) {
          if (nh == 0) h0 = j; else if (nh == 1) h1 = j; else if (nh == 2) h2 = j; else if (nh == 3) h3 = j;
          ++nh;
        }
      }
      // exclusive prefix of nh over the 8 lanes of the group
      int incl = nh;
#pragma unroll
      for (int o = 1; o < CMS_AREA_QL; o <<= 1) { const int t = __shfl_up(incl, o, CMS_AREA_QL); if (gl >= o) incl += t; }
      const int tot = __shfl(incl, CMS_AREA_QL - 1, CMS_AREA_QL);
      if (tmpq && nh > 0) {                                 // pass 0: the first CMS_AREA_TMP hits of the query, already in output order
        int w = n + incl - nh;
        if (w < CMS_AREA_TMP) {
          if (nh <= 4) {
            tmpq[w] = idx_base + h0;
            if (nh > 1 && w + 1 < CMS_AREA_TMP) tmpq[w + 1] = idx_base + h1;
            if (nh > 2 && w + 2 < CMS_AREA_TMP) tmpq[w + 2] = idx_base + h2;
            if (nh > 3 && w + 3 < CMS_AREA_TMP) tmpq[w + 3] = idx_base + h3;
          } else {
            for (int s = s0; s < s1 && w < CMS_AREA_TMP; ++s) {
              const int j = sorted_idx[s];
              const CmsKeyPoint p = kp[j];
              if (check && (p.octave < minLevel || (maxLevel >= 0 && p.octave > maxLevel))) continue;
              if (fabsf(p.x - x) < r && fabsf(p.y - y) < r) { tmpq[w] = idx_base + j; ++w; }
            }
          }
        }
      }
      if (pass && nh > 0) {
        int w = base + n + incl - nh;
        if (nh <= 4) {
          if (w < a.cap) a.idx[w] = idx_base + h0;
          if (nh > 1 && w + 1 < a.cap) a.idx[w + 1] = idx_base + h1;
          if (nh > 2 && w + 2 < a.cap) a.idx[w + 2] = idx_base + h2;
          if (nh > 3 && w + 3 < a.cap) a.idx[w + 3] = idx_base + h3;
        } else {                                            // crowded column: walk it again instead of a bigger local list
          for (int s = s0; s < s1; ++s) {
            const int j = sorted_idx[s];
            const CmsKeyPoint p = kp[j];
            if (check && (p.octave < minLevel || (maxLevel >= 0 && p.octave > maxLevel))) continue;
            if (fabsf(p.x - x) < r && fabsf(p.y - y) < r) { if (w < a.cap) a.idx[w] = idx_base + j; ++w; }
          }
        }
      }
      n += tot;
    }
  }
  if (!pass && q < a.nq && gl == 0) a.cnt[q] = n;             // 0 for a "no window" query
}

// CSR offsets: off[0] = 0, off[q + 1] = sum cnt[0..q], *total = off[nq].  Two launches: per-1024 block sums, then every block adds
// the sums of the blocks before it (a few hundred values at most) to its own scan.
extern "C" __global__ void __launch_bounds__(1024) k_area_blocksum(const int* __restrict__ cnt, int nq, int* __restrict__ bsum) {
  __shared__ int part[16];
  const int i = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int v = i < nq ? cnt[i] : 0;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane == 0) part[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += part[w]; bsum[blockIdx.x] = t; }
}
extern "C" __global__ void __launch_bounds__(1024) k_area_scan(const int* __restrict__ cnt, int nq, const int* __restrict__ bsum,
                                                                int* __restrict__ off, int* __restrict__ total) {
  __shared__ int part[16];
  __shared__ int s_before;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int before = 0;
  for (int b = tid; b < (int)blockIdx.x; b += 1024) before += bsum[b];
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
  if (lane == 0) part[wv] = before;
  __syncthreads();
  if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += part[w]; s_before = t; }
  __syncthreads();
  const int i = blockIdx.x * 1024 + tid;
  const int v = i < nq ? cnt[i] : 0;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  __syncthreads();
  if (lane == 63) part[wv] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wv; ++w) wbase += part[w];
  const int mine = s_before + wbase + incl;
  if (i < nq) off[i + 1] = mine;
  if (i == 0) off[0] = 0;
  if (i == nq - 1 && total) *total = mine;
}
