// cms_extract_kernels.hip -- gfx950 kernels for the per-frame extraction path, batched over B frames.
//
//   k_remap        fisheye -> 3F x 3F cubemap cross   (System::CvtFisheyeToCubeMap_reverseQuery_withInterpolation,
//                  System.cpp:327-355 == 5 x cv::remap INTER_LINEAR with the LUT of System.cpp:301-324)
//   k_resize       pyramid level l from level l-1       (ORBextractor::ComputePyramid, ORBExtractor.cpp:928-953)
//   k_fast_cells   FAST-9/16 score + per-cell NMS + ini/min threshold fallback, one wavefront per ~31x31 cell
//                  (ComputeKeyPointsOctTree cell loop, ORBExtractor.cpp:764-803, == one cv::FAST call per cell)
//   k_quadtree     DistributeOctTree (ORBExtractor.cpp:511-737), one workgroup per (frame, level)
//   k_cull         scale to image coords, face / border / mask cull, ordered compaction (ORBExtractor.cpp:887-904,915-921)
//   k_describe     IC_Angle (48-75) + GaussianBlur 7x7 (907-908, fused: only the 37x37 neighbourhood the taps can
//                  reach is blurred) + steered BRIEF (79-118), one wavefront per key point
//
// All integer results are bit-exact with the CPU oracle; no host round trip between stages.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cms_types.h"
#include "cms_detmath.h"
#include "cms_quadtree_core.h"

// ballot of a boolean: the builtin takes the i1 as it is (HIP's __ballot goes through an int and costs a select + compare per use)
#define CMS_BALLOT(p) ((unsigned long long)__builtin_amdgcn_ballot_w64((bool)(p)))
#define LANE_PREFIX(mask) ((int)__builtin_amdgcn_mbcnt_hi((uint32_t)((mask) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(mask), 0u)))

// ------------------------------------------------------------------------------------------------ remap
// LUT entry (one u32 per canvas pixel): X[0:11) | Y[11:22) | ax[22:27) | ay[27:32)  -- cv::remap's 5-bit fixed point.
// One thread = 4 consecutive canvas pixels of CMS_REMAP_FPT frames of the batch: one 16-byte LUT load and one decode serve all
// of them (the LUT is the largest stream of this kernel), then per frame 8 unaligned 16-bit gathers (the source image is cache
// resident) and one dword store.
#ifndef CMS_REMAP_FPT
#define CMS_REMAP_FPT 4
#endif
#ifndef CMS_REMAP_ZSPLIT
#define CMS_REMAP_ZSPLIT 64  /* frame groups processed side by side (measured: walking several groups per workgroup to re-use the
                                decoded LUT entry is slower -- 0.19 ms at 16 groups side by side, 0.23 ms at one) */
#endif
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;
extern "C" __global__ void __launch_bounds__(256)
k_remap(const uint8_t* __restrict__ fish, size_t fish_pitch, int fstride, int Iw, int Ih,
        const uint32_t* __restrict__ lut, int lut_stride, uint8_t* __restrict__ pyr, size_t pyr_bytes,
        int W, int stride0, int F, int write_corners, int B) {
  const int x0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  const int y = blockIdx.y;
  if (x0 >= W) return;
  const bool mid_row = (y >= F && y < 2 * F);
  const int ngroups = (B + CMS_REMAP_FPT - 1) / CMS_REMAP_FPT;
  if (!mid_row && (x0 + 3 < F || x0 >= 2 * F)) {  // corner block of the cross: kept at 0 (cubemap_lafida.cpp:110-111)
    if (write_corners)                              // already 0 unless a caller-supplied canvas was here before
      for (int f = blockIdx.z; f < B; f += gridDim.z) *reinterpret_cast<uint32_t*>(pyr + (size_t)f * pyr_bytes + (size_t)y * stride0 + x0) = 0u;
    return;
  }
  const uint4 e4 = *reinterpret_cast<const uint4*>(lut + (size_t)y * lut_stride + x0);
  const uint32_t e[4] = {e4.x, e4.y, e4.z, e4.w};
  // BORDER_CONSTANT(0) per tap: a valid map value just below Iw / Ih can round up to X == Iw / Y == Ih.
  // The two taps of a row come from ONE unaligned 16-bit load (the gather is bound by the number of address requests, not by
  // bytes); the byte after the last pixel of a row is readable (row padding / next row / 256-byte slack) and masked out.
  int soff[4], w00[4], w01[4], w10[4], w11[4];
  bool ld0[4], ld1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int x = x0 + i;
    const bool valid = x < W && (mid_row || (x >= F && x < 2 * F));   // pixels of a straddling quad inside a corner block get 0
    const int X = e[i] & 0x7FF, Y = (e[i] >> 11) & 0x7FF, ax = (e[i] >> 22) & 31, ay = e[i] >> 27;
    const bool x0in = X < Iw, y0in = Y < Ih, x1in = X + 1 < Iw, y1in = Y + 1 < Ih;
    soff[i] = Y * fstride + X;
    ld0[i] = valid && y0in; ld1[i] = valid && y1in;
    w00[i] = x0in ? (32 - ay) * (32 - ax) : 0; w01[i] = x1in ? (32 - ay) * ax : 0;
    w10[i] = x0in ? ay * (32 - ax) : 0;        w11[i] = x1in ? ay * ax : 0;
  }
  // the decoded LUT entry serves every frame group this workgroup walks (gridDim.z of them run side by side)
  for (int fg = blockIdx.z; fg < ngroups; fg += gridDim.z) {
    const int b0 = fg * CMS_REMAP_FPT, nb = min(CMS_REMAP_FPT, B - b0);
    uint8_t* dst = pyr + (size_t)b0 * pyr_bytes + (size_t)y * stride0 + x0;
#pragma unroll
    for (int f = 0; f < CMS_REMAP_FPT; ++f) {
      if (f >= nb) break;
      const uint8_t* src = fish + (size_t)(b0 + f) * fish_pitch;
      uint32_t r0[4], r1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r0[i] = ld0[i] ? (uint32_t)*reinterpret_cast<const u16_unaligned*>(src + soff[i]) : 0u;
        r1[i] = ld1[i] ? (uint32_t)*reinterpret_cast<const u16_unaligned*>(src + soff[i] + fstride) : 0u;
      }
      uint32_t out = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (32-ay)((32-ax) p00 + ax p01) + ay((32-ax) p10 + ax p11): integer arithmetic, any association gives the same value
        const int S = w00[i] * (int)(r0[i] & 0xFF) + w01[i] * (int)(r0[i] >> 8) + w10[i] * (int)(r1[i] & 0xFF) + w11[i] * (int)(r1[i] >> 8);
        out |= (uint32_t)((S + 512) >> 10) << (8 * i);
      }
      *reinterpret_cast<uint32_t*>(dst + (size_t)f * pyr_bytes) = out;
    }
  }
}

// ------------------------------------------------------------------------------------------------ resize
// One workgroup (64 x 4 threads) produces a 256 x CMS_RZ_ROWS destination tile.  The source rectangle it needs (about 310 x 12
// pixels at scale 1.2) is staged in LDS with coalesced dword loads; the 2 x 2 taps of every destination pixel are then
// byte reads from LDS (a byte gather straight from global memory is bound by the texture-address path, not by HBM).
// `ls` = LDS row stride in bytes (multiple of 4, <= 512).
#ifndef CMS_RZ_ROWS
#define CMS_RZ_ROWS 16
#endif
extern "C" __global__ void __launch_bounds__(256)
k_resize(uint8_t* __restrict__ pyr, size_t pyr_bytes, CmsLevel src, CmsLevel dst,
         const CmsResizeTab* __restrict__ tabx, const CmsResizeTab* __restrict__ taby, int ls, int skip_zero, int scale_lo, int scale_hi) {
  extern __shared__ __align__(16) uint8_t rtile[];
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
  const int xb = blockIdx.x * 256, yb = blockIdx.y * CMS_RZ_ROWS, b = blockIdx.z;
  const int xl = min(xb + 255, dst.w - 1), yl = min(yb + CMS_RZ_ROWS - 1, dst.h - 1);
  // tile inside the constant-zero corner region of a remapped cross (see CmsLevel::zlo): source and destination are 0 already
  if (skip_zero && (xl < dst.zlo || xb >= dst.w - dst.zhi) && (yl < dst.zlo || yb >= dst.h - dst.zhi)) return;
  // staged source rectangle: a conservative superset computed arithmetically (tap index s(d) = floor((d + 0.5) scale - 0.5)
  // lies in [floor(d scale) - 1, floor((d + 1) scale)]), so the pixel loads do not wait for the coefficient-table loads.
  // scale_lo / scale_hi = the ratio rounded down / up to 16 fractional bits: (d scale_lo) >> 16 <= floor(d scale) <=
  // (d scale_hi + 65535) >> 16, all of it wave-uniform integer arithmetic on the scalar unit (the double-precision floor this
  // replaces ran on the vector unit at half rate, 16 instructions per wave)
  const int c0 = max(((xb * scale_lo) >> 16) - 1, 0) & ~15;
  const int c1 = min((((xl + 1) * scale_hi + 65535) >> 16) + 1, src.w - 1);
  const int r0 = min(max(((yb * scale_lo) >> 16) - 1, 0), src.h - 1);
  const int r1 = min((((yl + 1) * scale_hi + 65535) >> 16) + 1, src.h - 1);
  const int nq = ((c1 - c0) >> 4) + 1, nr = r1 - r0 + 1;          // 16-byte columns (<= 32), rows
  const uint8_t* simg = pyr + (size_t)b * pyr_bytes + src.off;
  const int x0 = xb + 4 * tx;
  // every global load of this thread (coefficient entries, then its share of the source rectangle, 16 bytes per load: level
  // rows are 128-byte aligned) is issued before the first dependent use, so the workgroup pays one memory latency, not one per row
  CmsResizeTab t4[4], tyr[CMS_RZ_ROWS / 4];
#pragma unroll
  for (int i = 0; i < 4; ++i) t4[i] = tabx[min(x0 + i, dst.w - 1)];
#pragma unroll
  for (int rr = 0; rr < CMS_RZ_ROWS / 4; ++rr) tyr[rr] = taby[min(yb + ty + 4 * rr, dst.h - 1)];
  {
    const int c = tid & 31, rs = tid >> 5;
    const uint8_t* gp = simg + (size_t)r0 * src.stride + c0 + 16 * c;
    constexpr int NLD = (CMS_RZ_ROWS * 2 + 5 + 7) / 8;               // rows staged <= 1.9 * CMS_RZ_ROWS + 5
    uint4 tmp[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int r = rs + 8 * k;
      tmp[k] = (c < nq && r < nr) ? *reinterpret_cast<const uint4*>(gp + (uint32_t)__mul24(r, src.stride)) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int r = rs + 8 * k;
      if (c < nq && r < nr) reinterpret_cast<uint4*>(rtile + __mul24(r, ls))[c] = tmp[k];
    }
  }
  __syncthreads();
  if (x0 >= dst.w) return;
  int ca[4], cb[4], a0[4], a1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const CmsResizeTab t = t4[i];
    ca[i] = (int)t.s - c0;
    cb[i] = min((int)t.s + 1, src.w - 1) - c0;
    a0[i] = t.a0; a1[i] = t.a1;
  }
#pragma unroll
  for (int rr = 0; rr < CMS_RZ_ROWS / 4; ++rr) {
    const int y = yb + ty + 4 * rr;
    if (y >= dst.h) break;
    const CmsResizeTab tyy = tyr[rr];
    const uint8_t* S0 = rtile + __mul24(min(max((int)tyy.s, 0), src.h - 1) - r0, ls);
    const uint8_t* S1 = rtile + __mul24(min(max((int)tyy.s + 1, 0), src.h - 1) - r0, ls);
    const int b0 = tyy.a0, b1 = tyy.a1;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // 8-bit pixels x 11-bit weights: 24-bit multiplies (full rate; the 32-bit integer multiply is not)
      // (the right tap keeps its own address: written as left + 1 the two byte reads are fused into one unaligned 16-bit LDS read,
      // which runs 4x slower)
      const int h0 = __mul24(S0[ca[i]], a0[i]) + __mul24(S0[cb[i]], a1[i]);
      const int h1 = __mul24(S1[ca[i]], a0[i]) + __mul24(S1[cb[i]], a1[i]);
      // weights are in [0, 2048], h >> 4 < 2^16: the masks tell the compiler so and it picks the 24-bit multiply
      const int v = (int)((((uint32_t)b0 & 0xFFFu) * (((uint32_t)h0 >> 4) & 0xFFFFu) >> 16) +
                          (((uint32_t)b1 & 0xFFFu) * (((uint32_t)h1 >> 4) & 0xFFFFu) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 0xFF) << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(pyr + (size_t)b * pyr_bytes + dst.off + (size_t)y * dst.stride + x0) = out;
  }
}

// ------------------------------------------------------------------------------------------------ resize, two levels per launch
// Level l + 1 is computed from level l, so a launch per level reads every intermediate level back from HBM right after writing it (and pays a
// launch's latency per level: seven small launches per batch).  Here a workgroup produces a 256 x CMS_RZ_ROWS tile of level l + 1 AND the part of
// level l that tile is computed from: it stages the rectangle of level l - 1 behind it in LDS, computes its rectangle of level l into LDS,
// stores the part of it that it OWNS (rectangles of neighbouring workgroups overlap by a few pixels: the boundaries between them are the left /
// top edges of the rectangles, which ascend with the tile index), and computes its tile of level l + 1 from the LDS copy.  Same integer
// arithmetic per pixel as k_resize (cv::resize INTER_LINEAR, 11-bit coefficients): bit-identical levels; about 10 % of level l is computed twice
// (the overlaps); level l is never read back.  With CMS_RESIZE_FUSED=1 levels (1, 2), (3, 4), (5, 6) go through this kernel, level 7 through k_resize.
// NOT the default: measured 2.42 against 0.89 ms per 256 frames for the whole pyramid (round 5) -- the pyramid is bound by per-pixel integer work and LDS
// byte reads (~0.5 ms of vector issue), not by the bytes of the intermediate level, and this kernel's second stage (an irregular rectangle, tables through LDS,
// 30 KB of LDS per workgroup) runs that work less efficiently.  Kept as a documented experiment; its index logic is what the CPU replay verified.
//   src = level l - 1, mid = level l, dst = level l + 1; tab?1 = mid's tables (from src), tab?2 = dst's tables (from mid)
//   ls = LDS row stride of the staged src rectangle, la = of the mid rectangle (both multiples of 16); lo? / hi?: the ratios w(l-1) / w(l) and
//   w(l) / w(l+1) rounded down / up to 16 fractional bits
#define CMS_RZ2_SROWS 40          /* rows of src a workgroup may stage: 16 x 1.2 + 4 rows of mid, x 1.2 + 4 rows of src = 32 at the pyramid's ratio */
#define CMS_RZ2_AROWS 26          /* rows of mid */
struct CmsResize2 {
  CmsLevel src, mid, dst;
  const CmsResizeTab* tabx1; const CmsResizeTab* taby1; const CmsResizeTab* tabx2; const CmsResizeTab* taby2;
  int ls, la, arows, lo1, hi1, lo2, hi2, skip_zero;
};
extern "C" __global__ void __launch_bounds__(256)
k_resize2(uint8_t* __restrict__ pyr, size_t pyr_bytes, CmsResize2 a) {
  extern __shared__ __align__(16) uint8_t rtile[];
  const CmsLevel& src = a.src; const CmsLevel& mid = a.mid; const CmsLevel& dst = a.dst;
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
  const int xb = blockIdx.x * 256, yb = blockIdx.y * CMS_RZ_ROWS, b = blockIdx.z;
  const int xl = min(xb + 255, dst.w - 1), yl = min(yb + CMS_RZ_ROWS - 1, dst.h - 1);
  // the rectangle of mid this tile is computed from (k_resize's bounds) and the part of it this workgroup owns: up to where the next tile's
  // rectangle begins (the last tile of a row / column: up to the level's edge)
  auto lo_of = [&](int d, int lo, int lim) { return min(max(((d * lo) >> 16) - 1, 0), lim); };
  const int ac0 = lo_of(xb, a.lo2, mid.w - 1) & ~15;
  const int ac1 = min((((xl + 1) * a.hi2 + 65535) >> 16) + 1, mid.w - 1);
  const int ar0 = lo_of(yb, a.lo2, mid.h - 1);
  const int ar1 = min((((yl + 1) * a.hi2 + 65535) >> 16) + 1, mid.h - 1);
  const int ox1 = (int)blockIdx.x + 1 < (int)gridDim.x ? (lo_of(xb + 256, a.lo2, mid.w - 1) & ~15) : mid.w;
  const int oy1 = (int)blockIdx.y + 1 < (int)gridDim.y ? lo_of(yb + CMS_RZ_ROWS, a.lo2, mid.h - 1) : mid.h;
  // both the tile of dst and the owned part of mid inside the constant-zero corner region of a remapped cross: nothing to do
  if (a.skip_zero && (xl < dst.zlo || xb >= dst.w - dst.zhi) && (yl < dst.zlo || yb >= dst.h - dst.zhi) &&
      (ox1 - 1 < mid.zlo || ac0 >= mid.w - mid.zhi) && (oy1 - 1 < mid.zlo || ar0 >= mid.h - mid.zhi)) return;
  // ... and the rectangle of src behind the rectangle of mid
  const int sc0 = lo_of(ac0, a.lo1, src.w - 1) & ~15;
  const int sc1 = min((((ac1 + 1) * a.hi1 + 65535) >> 16) + 1, src.w - 1);
  const int sr0 = lo_of(ar0, a.lo1, src.h - 1);
  const int sr1 = min((((ar1 + 1) * a.hi1 + 65535) >> 16) + 1, src.h - 1);
  const int nq = ((sc1 - sc0) >> 4) + 1, nr = sr1 - sr0 + 1;      // 16-byte columns (<= 32), rows (<= CMS_RZ2_SROWS)
  uint8_t* tS = rtile;                                             // staged src rectangle, row stride ls
  uint8_t* tA = rtile + (size_t)a.ls * CMS_RZ2_SROWS;              // mid rectangle, row stride la (a.arows rows)
  CmsResizeTab* tX = reinterpret_cast<CmsResizeTab*>(tA + (size_t)a.la * a.arows);      // mid's x table for columns ac0 .. ac1
  const uint8_t* simg = pyr + (size_t)b * pyr_bytes + src.off;
  {
    const int c = tid & 31, rs = tid >> 5;
    const uint8_t* gp = simg + (size_t)sr0 * src.stride + sc0 + 16 * c;
    uint4 tmp[CMS_RZ2_SROWS / 8];
#pragma unroll
    for (int k = 0; k < CMS_RZ2_SROWS / 8; ++k) {
      const int r = rs + 8 * k;
      tmp[k] = (c < nq && r < nr) ? *reinterpret_cast<const uint4*>(gp + (uint32_t)__mul24(r, src.stride)) : make_uint4(0, 0, 0, 0);
    }
    const int nax = ac1 - ac0 + 1;
    for (int i = tid; i < nax; i += 256) tX[i] = a.tabx1[min(ac0 + i, mid.w - 1)];
#pragma unroll
    for (int k = 0; k < CMS_RZ2_SROWS / 8; ++k) {
      const int r = rs + 8 * k;
      if (c < nq && r < nr) reinterpret_cast<uint4*>(tS + __mul24(r, a.ls))[c] = tmp[k];
    }
  }
  __syncthreads();
  // ---- the rectangle of mid: four pixels per task, tasks over (row, dword column)
  {
    const int ndw = ((ac1 - ac0) >> 2) + 1, nar = ar1 - ar0 + 1;
    uint8_t* mimg = pyr + (size_t)b * pyr_bytes + mid.off;
    for (int row = ty; row < nar; row += 4) {
      const int y = ar0 + row;
      const CmsResizeTab tyy = a.taby1[y];
      const uint8_t* S0 = tS + __mul24(min(max((int)tyy.s, 0), src.h - 1) - sr0, a.ls);
      const uint8_t* S1 = tS + __mul24(min(max((int)tyy.s + 1, 0), src.h - 1) - sr0, a.ls);
      const int b0 = tyy.a0, b1 = tyy.a1;
      for (int dw = tx; dw < ndw; dw += 64) {
        const int x0 = ac0 + 4 * dw;
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const CmsResizeTab t = tX[min(4 * dw + i, ac1 - ac0)];
          const int ca = (int)t.s - sc0, cb = min((int)t.s + 1, src.w - 1) - sc0;
          const int h0 = __mul24(S0[ca], t.a0) + __mul24(S0[cb], t.a1);
          const int h1 = __mul24(S1[ca], t.a0) + __mul24(S1[cb], t.a1);
          const int v = (int)((((uint32_t)b0 & 0xFFFu) * (((uint32_t)h0 >> 4) & 0xFFFFu) >> 16) +
                              (((uint32_t)b1 & 0xFFFu) * (((uint32_t)h1 >> 4) & 0xFFFFu) >> 16) + 2) >> 2;
          out |= (uint32_t)(v & 0xFF) << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(tA + __mul24(row, a.la) + 4 * dw) = out;
        if (x0 < ox1 && y < oy1) *reinterpret_cast<uint32_t*>(mimg + (size_t)y * mid.stride + x0) = out;      // (x0 and ox1 are multiples of 4 or the level's width)
      }
    }
  }
  __syncthreads();
  // ---- the tile of dst from the LDS copy of mid: k_resize's body
  const int x0 = xb + 4 * tx;
  if (x0 >= dst.w) return;
  int ca[4], cb[4], a0[4], a1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const CmsResizeTab t = a.tabx2[min(x0 + i, dst.w - 1)];
    ca[i] = (int)t.s - ac0;
    cb[i] = min((int)t.s + 1, mid.w - 1) - ac0;
    a0[i] = t.a0; a1[i] = t.a1;
  }
#pragma unroll
  for (int rr = 0; rr < CMS_RZ_ROWS / 4; ++rr) {
    const int y = yb + ty + 4 * rr;
    if (y >= dst.h) break;
    const CmsResizeTab tyy = a.taby2[y];
    const uint8_t* S0 = tA + __mul24(min(max((int)tyy.s, 0), mid.h - 1) - ar0, a.la);
    const uint8_t* S1 = tA + __mul24(min(max((int)tyy.s + 1, 0), mid.h - 1) - ar0, a.la);
    const int b0 = tyy.a0, b1 = tyy.a1;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int h0 = __mul24(S0[ca[i]], a0[i]) + __mul24(S0[cb[i]], a1[i]);
      const int h1 = __mul24(S1[ca[i]], a0[i]) + __mul24(S1[cb[i]], a1[i]);
      const int v = (int)((((uint32_t)b0 & 0xFFFu) * (((uint32_t)h0 >> 4) & 0xFFFFu) >> 16) +
                          (((uint32_t)b1 & 0xFFFu) * (((uint32_t)h1 >> 4) & 0xFFFFu) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 0xFF) << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(pyr + (size_t)b * pyr_bytes + dst.off + (size_t)y * dst.stride + x0) = out;
  }
}

// ------------------------------------------------------------------------------------------------ FAST cells
// Arc score A(p) = max over the 16 contiguous 9-arcs and both polarities of min |v - x|; cornerScore<16> == A - 1.
// p is a corner at threshold t  <=>  A(p) > t.   Returns A-1 if A > t else 0.
__device__ __forceinline__ int fast_arc_score(const int d[16], int t) {
  uint32_t md = 0, mb = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { md |= (d[k] > t ? 1u : 0u) << k; mb |= (-d[k] > t ? 1u : 0u) << k; }
  auto has9 = [](uint32_t m) -> bool {
    m |= m << 16;                    // unroll the circle
    uint32_t r = m & (m >> 1);       // 2 consecutive
    r &= r >> 2;                     // 4
    r &= r >> 4;                     // 8
    r &= m >> 8;                     // 9
    return (r & 0xFFFFu) != 0;
  };
  const bool cd = has9(md), cb = has9(mb);
  if (!cd && !cb) return 0;
  int e[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) e[k] = cd ? d[k] : -d[k];
  int m2[16], m4[16], m8[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m2[k] = min(e[k], e[(k + 1) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) m8[k] = min(m4[k], m4[(k + 4) & 15]);
  int A = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) A = max(A, min(m8[k], e[(k + 8) & 15]));
  return A - 1;
}

// Same score on packed 16-bit lanes, both polarities at once: lane pair e[k] = (v - p_k, p_k - v); a 9-arc of one polarity
// always overlaps any 9-arc of the other (9 + 9 > 16), so at most one of the two arc scores exceeds t and max(dark, bright)
// is the score of the polarity that makes p a corner.  111 packed operations instead of ~200 scalar ones.
typedef short s2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int fast_arc_score_pk(const uint8_t* c, int ts, int v, int t) {
  const int off[16] = {3 * ts, 3 * ts + 1, 2 * ts + 2, ts + 3, 3, -ts + 3, -2 * ts + 2, -3 * ts + 1,
                       -3 * ts, -3 * ts - 1, -2 * ts - 2, -ts - 3, -3, ts - 3, 2 * ts - 2, 3 * ts - 1};
  s2_t e[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const s2_t x = __builtin_bit_cast(s2_t, (uint32_t)v | ((uint32_t)c[off[k]] << 16));     // (v, p_k)
    e[k] = x - __builtin_shufflevector(x, x, 1, 0);                                           // (v - p_k, p_k - v)
  }
  // minima of the sixteen cyclic 9-windows [k, k+8] by block prefix / suffix minima over the unrolled ring x[j] = e[j & 15], blocks
  // [0,8] [9,17] [18,23]: 29 + 14 minima instead of 64 for the doubling scheme
#define EX(j) e[(j) & 15]
  s2_t S0[9], P1[9], S1[9], P2[6];
  S0[8] = EX(8);
#pragma unroll
  for (int j = 7; j >= 0; --j) S0[j] = __builtin_elementwise_min(EX(j), S0[j + 1]);            // min x[j..8]
  P1[0] = EX(9);
#pragma unroll
  for (int j = 1; j < 9; ++j) P1[j] = __builtin_elementwise_min(P1[j - 1], EX(9 + j));          // min x[9..9+j]
  S1[8] = EX(17);
#pragma unroll
  for (int j = 7; j >= 0; --j) S1[j] = __builtin_elementwise_min(EX(9 + j), S1[j + 1]);        // min x[9+j..17]
  P2[0] = EX(18);
#pragma unroll
  for (int j = 1; j < 6; ++j) P2[j] = __builtin_elementwise_min(P2[j - 1], EX(18 + j));         // min x[18..18+j]
#undef EX
  s2_t a = __builtin_elementwise_max(S0[0], S1[0]);                                             // windows 0 and 9
#pragma unroll
  for (int k = 1; k <= 8; ++k) a = __builtin_elementwise_max(a, __builtin_elementwise_min(S0[k], P1[k - 1]));     // [k,8] + [9,k+8]
#pragma unroll
  for (int k = 10; k <= 15; ++k) a = __builtin_elementwise_max(a, __builtin_elementwise_min(S1[k - 9], P2[k - 10]));   // [k,17] + [18,k+8]
  const int A = max((int)a.x, (int)a.y);
  return A > t ? A - 1 : 0;
}

// The wavefronts of a workgroup work on different cells and never exchange data: all synchronisation is inside a
// wavefront (LDS operations of one wave complete in order; the fence only pins the compiler's order).
#ifndef CMS_FAST_LIST
#define CMS_FAST_LIST 1     /* launch only the host-listed cells that can hold a corner (0.342 -> 0.329 ms per 32 frames) */
#endif
#ifndef CMS_FAST_SYNC_WG
#define CMS_FAST_SYNC_WG 0
#endif
#if CMS_FAST_SYNC_WG
#define WAVE_SYNC() __syncthreads()
#else
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
#ifndef CMS_FAST_LD
#define CMS_FAST_LD 8       /* row loads in flight per lane while staging the ROI (16-byte loads were measured slower here) */
#endif
__device__ const int k_fast_recip[32] = {0, 65537, 32769, 21846, 16385, 13108, 10923, 9363, 8193, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4097,
                                         3856, 3641, 3450, 3277, 3121, 2979, 2850, 2731, 2622, 2521, 2428, 2341, 2260, 2185, 2115};
#ifndef CMS_FAST_REFINE_MIN
#define CMS_FAST_REFINE_MIN 128
#endif
#ifndef CMS_FAST_WPB
#define CMS_FAST_WPB 1      /* cells (wavefronts) per workgroup; measured: 1 -> 0.34 ms, 4 -> 0.38 ms per 32 frames */
#endif
extern "C" __global__ void __launch_bounds__(64 * CMS_FAST_WPB)
k_fast_cells(const uint8_t* __restrict__ pyr, size_t pyr_bytes, CmsGeom g, const int* __restrict__ cell_list, int n_list,
             uint32_t* __restrict__ cell_cand, int* __restrict__ cell_cnt, int* __restrict__ overflow) {
  extern __shared__ __align__(16) uint8_t smem_all[];
#if CMS_FAST_WPB == 1
  const int lane = threadIdx.x, wave = 0;           // compile-time LDS base: addresses fold into the instruction offsets
#else
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#endif
  const int slot = blockIdx.x * CMS_FAST_WPB + wave;
#if CMS_FAST_LIST
  if (slot >= n_list) return;
  // host-built list of the cells that can hold a corner (see cms_ctx_create); an entry carries level | row | column
  const int entry = cell_list[slot];
  const int l = entry & 15, ci = (entry >> 4) & 255, cj = (entry >> 12) & 255;
  const CmsLevel& lv = g.lv[l];
  const int cid = lv.cell0 + ci * lv.nCols + cj;
#else
  if (slot >= g.total_cells) return;
  const int cid = slot;
  int l = 0;
  for (int k = 1; k < g.nlevels; ++k) if (cid >= g.lv[k].cell0) l = k;
  const CmsLevel& lv = g.lv[l];
  const int lc = cid - lv.cell0;
  const int ci = lc / lv.nCols, cj = lc - ci * lv.nCols;
#endif
  uint8_t* smem = smem_all + (size_t)wave * g.fast_cell_lds;
  const int b = blockIdx.y;
  const int maxBX = lv.w - CMS_MINB, maxBY = lv.h - CMS_MINB;
  const int iniY = CMS_MINB + ci * lv.hCell, iniX = CMS_MINB + cj * lv.wCell;
  if (iniY >= maxBY - 3 || iniX >= maxBX - 6) return;       // ORBExtractor.cpp:769,777
  const int maxY = min(iniY + lv.hCell + 6, maxBY), maxX = min(iniX + lv.wCell + 6, maxBX);
  const int ex0 = iniX + 3, ex1 = maxX - 3, ey0 = iniY + 3, ey1 = maxY - 3;   // pixels cv::FAST evaluates in this ROI
  const int ew = ex1 - ex0, eh = ey1 - ey0;
  if (ew <= 0 || eh <= 0) return;
  // 4/9 of the cross are corner blocks that k_remap keeps at 0: a cell whose whole ROI lies in that constant region (at this
  // level: [0, zlo) or [w - zhi, w) in both directions) has no corner by definition -- identical result, no work
  if (g.skip_zero_cells && (maxX <= lv.zlo || iniX >= lv.w - lv.zhi) && (maxY <= lv.zlo || iniY >= lv.h - lv.zhi)) return;
  if (g.dbg_stop == 9) return;

  uint8_t* tile = smem;                                             // [tile_h][tile_stride]
  uint8_t* sc = tile + g.tile_h * g.tile_stride;                     // [sc_h][sc_stride], zero border
  uint16_t* list = reinterpret_cast<uint16_t*>(sc + g.sc_h * g.sc_stride);
  const int ts = g.tile_stride, ss = g.sc_stride;

  // ---- stage the ROI in LDS with aligned dword loads (lane -> (row, dword) fixed, rows advance by 64 / ndw)
  const uint8_t* img = pyr + (size_t)b * pyr_bytes + lv.off;
  const int ax0 = iniX & ~3;
  const int ndw = (maxX - ax0 + 3) >> 2, th = maxY - iniY;
  {
    // all loads of a lane are issued before the first LDS store: one memory latency per cell instead of one per row group
    // lane / ndw and 64 / ndw without integer division (ndw <= 16, lane < 64: floor(x / n) == (x * (65536 / n + 1)) >> 16)
    const int rcp = k_fast_recip[ndw & 31];
    const int lr = __mul24(lane, rcp) >> 16, lc2 = lane - __mul24(lr, ndw), rstep = (64 * rcp) >> 16;
    const uint8_t* gp = img + (size_t)iniY * lv.stride + ax0 + 4 * lc2;
    for (int rbase = 0; rbase < th; rbase += CMS_FAST_LD * rstep) {      // one pass for the usual ~37-row ROI
      uint32_t tmp[CMS_FAST_LD];
#pragma unroll
      for (int k = 0; k < CMS_FAST_LD; ++k) {
        const int r = rbase + lr + k * rstep;
        tmp[k] = (lr < rstep && r < th) ? *reinterpret_cast<const uint32_t*>(gp + (uint32_t)__mul24(r, lv.stride)) : 0u;
      }
#pragma unroll
      for (int k = 0; k < CMS_FAST_LD; ++k) {
        const int r = rbase + lr + k * rstep;
        if (lr < rstep && r < th) reinterpret_cast<uint32_t*>(tile + __mul24(r, ts))[lc2] = tmp[k];
      }
    }
  }
  for (int idx = lane; idx < (g.sc_h * ss) >> 2; idx += 64) reinterpret_cast<uint32_t*>(sc)[idx] = 0u;
  WAVE_SYNC();
  if (g.dbg_stop == 1) return;

  // ---- phase A: 4-point compass pre-test at minTh on 4 pixels per lane (5 dword LDS reads per quad instead of 20 byte
  // reads), survivors are compacted into an LDS list of (py << 6 | px) codes
  const int t = g.min_th;
  const int lx0 = ex0 - ax0, lx1 = ex1 - ax0;           // evaluated LDS columns [lx0, lx1)
  const int q0 = lx0 >> 2, nq = ((lx1 - 1) >> 2) - q0 + 1;
  const int qrcp = k_fast_recip[nq & 31];                      // (32-bit integer multiplies and divisions are quarter rate: 24-bit forms throughout)
  const int qr = __mul24(lane, qrcp) >> 16, qc = lane - __mul24(qr, nq), qrows = (64 * qrcp) >> 16;
  // which of this lane's 4 columns are evaluated (row independent): kept as wave masks, so that the per-pixel survivor mask below
  // is scalar arithmetic on compare results (a ballot of a combined boolean costs a select + compare per use)
  bool colok[4];
  unsigned long long colm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int lxi = 4 * (q0 + qc) + i; colok[i] = lxi >= lx0 && lxi < lx1; colm[i] = CMS_BALLOT(colok[i]); }
  int L = 0;
  // the lane's quad in its first row; rows advance by `qrows`: one pointer increment per round instead of five row products
  const uint8_t* prow = tile + __mul24(ey0 - iniY + qr, ts) + 4 * (q0 + qc);
  const uint8_t* prow0 = tile + __mul24(ey0 - iniY, ts) + 4 * q0;         // row 0, first quad: valid for every lane
  const int rstride = __mul24(qrows, ts), ts3 = 3 * ts;
  const int q = q0 + qc;
  const unsigned long long qrm = CMS_BALLOT(qr < qrows);
  for (int rb = 0; rb < eh; rb += qrows, prow += rstride) {
    const int py = rb + qr;
    const bool rowok = qr < qrows && py < eh;
    const unsigned long long rowm = CMS_BALLOT(py < eh) & qrm;      // == ballot(rowok), from plain compare masks
    // a lane outside the rows reads its first-row quad again (always staged) and is masked out of the result below
    const uint8_t* pr = rowok ? prow : prow0;
    const uint32_t cm = *reinterpret_cast<const uint32_t*>(pr - 4), cc = *reinterpret_cast<const uint32_t*>(pr), cp = *reinterpret_cast<const uint32_t*>(pr + 4);
    const uint32_t up = *reinterpret_cast<const uint32_t*>(pr - ts3), dn = *reinterpret_cast<const uint32_t*>(pr + ts3);
    // the four compass points are the four (vertical, horizontal) combinations of {p0, p8} x {p4, p12}, so
    //   "some adjacent pair is darker than v - t"   <=>  max(min(p0, p8), min(p4, p12)) < v - t
    //   "some adjacent pair is brighter than v + t" <=>  min(max(p0, p8), max(p4, p12)) > v + t
    // -- six min/max and two compares per pixel, all on byte i of five dwords (x+3 and x-3 are brought into the lane's byte
    // positions by two byte-align operations per quad).
    const uint32_t e4 = __builtin_amdgcn_alignbyte(cp, cc, 3);     // bytes lx+3 .. lx+6
    const uint32_t w4 = __builtin_amdgcn_alignbyte(cc, cm, 1);     // bytes lx-3 .. lx
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lx = 4 * q + i;
      const int v = (cc >> (8 * i)) & 0xFF;
      const int p4 = (e4 >> (8 * i)) & 0xFF, p12 = (w4 >> (8 * i)) & 0xFF;
      const int p0 = (dn >> (8 * i)) & 0xFF, p8 = (up >> (8 * i)) & 0xFF;
      const int dmax = max(min(p0, p8), min(p4, p12));
      const int bmin = min(max(p0, p8), max(p4, p12));
      const unsigned long long m = (CMS_BALLOT(dmax < v - t) | CMS_BALLOT(bmin > v + t)) & colm[i] & rowm;
      const bool pass = __builtin_amdgcn_inverse_ballot_w64(m);
      if (pass) list[L + LANE_PREFIX(m)] = (uint16_t)((py << 6) | (lx - lx0));
      L += __popcll(m);
    }
  }
  WAVE_SYNC();
  if (g.dbg_stop == 2) return;

  // ---- phase A2: 8-point refinement on the list.  Nine contiguous ring pixels always cover four consecutive even ring
  // positions (0,2,..,14), so a corner needs 4 cyclically consecutive "darker" (or "brighter") bits among those eight.
  // (only for long lists: for L <= 64 the refinement is one more pass of 64 lanes in front of a ring pass that costs the same with 12 live
  // lanes as with 47; measured at 256 frames: always 1.634 ms, L > 64 1.605, L > 128 1.588 -- CMS_FAST_REFINE_MIN)
  if (L > CMS_FAST_REFINE_MIN) {
    int L2 = 0;
    for (int base = 0; base < L; base += 64) {
      // every lane runs the test (lanes past the end on entry 0, masked out of the result): the sixteen compare results are wave
      // masks and the run test is scalar and/or on them -- building per-lane bit fields cost a select and an or per bit
      const int k = base + lane;
      const int code = list[k < L ? k : 0];
      const int py = code >> 6, px = code & 63;
      const uint8_t* c = tile + __mul24(ey0 - iniY + py, ts) + (lx0 + px);
      const int v = c[0];
      const int e[8] = {v - c[3 * ts], v - c[2 * ts + 2], v - c[3], v - c[-2 * ts + 2],
                        v - c[-3 * ts], v - c[-2 * ts - 2], v - c[-3], v - c[2 * ts - 2]};
      unsigned long long dk[8], br[8], d2[8], b2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { dk[i] = CMS_BALLOT(e[i] > t); br[i] = CMS_BALLOT(e[i] < -t); }
#pragma unroll
      for (int i = 0; i < 8; ++i) { d2[i] = dk[i] & dk[(i + 1) & 7]; b2[i] = br[i] & br[(i + 1) & 7]; }
      unsigned long long m = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) m |= (d2[i] & d2[(i + 2) & 7]) | (b2[i] & b2[(i + 2) & 7]);
      m &= CMS_BALLOT(k < L);
      WAVE_SYNC();                         // every lane has read its entry before the compacted list is written
      if (__builtin_amdgcn_inverse_ballot_w64(m)) list[L2 + LANE_PREFIX(m)] = (uint16_t)code;
      L2 += __popcll(m);
    }
    L = L2;
  }
  WAVE_SYNC();
  if (g.dbg_stop == 3) return;

  // ---- phase B: full 16-pixel ring score for the listed pixels
  for (int base = 0; base < L; base += 64) {
    const int k = base + lane;
    if (k < L) {
      const int code = list[k];
      const int py = code >> 6, px = code & 63;
      const uint8_t* c = tile + __mul24(ey0 - iniY + py, ts) + (lx0 + px);
      const int v = c[0];
      const int S = fast_arc_score_pk(c, ts, v, t);
      if (S > 0) sc[__mul24(py + 1, ss) + px + 1] = (uint8_t)S;
      else list[k] = 0xFFFFu;
    }
  }
  WAVE_SYNC();
  if (g.dbg_stop == 4) return;

  // ---- phase C: strict 8-neighbour maximum inside this cell, iniTh else minTh, emit
  int n_all = 0, n_ini = 0;
  for (int base = 0; base < L; base += 64) {
    // all nine scores are read at once (a short-circuit chain waits for eight LDS round trips in a row) by every lane -- dead
    // entries and lanes past the end look at pixel (0, 0) and are masked out; the comparisons meet as wave masks
    const int k = base + lane;
    const int code = list[k < L ? k : 0];
    const unsigned long long livem = CMS_BALLOT(k < L) & CMS_BALLOT(code != 0xFFFF);
    const int cd = code != 0xFFFF ? code : 0;
    const int py = cd >> 6, px = cd & 63;
    const uint8_t* s = sc + __mul24(py + 1, ss) + px + 1;
    const int S = s[0];
    const int n0 = s[-1], n1 = s[1], n2 = s[-ss - 1], n3 = s[-ss], n4 = s[-ss + 1], n5 = s[ss - 1], n6 = s[ss], n7 = s[ss + 1];
    const unsigned long long keepm = livem & CMS_BALLOT(S > n0) & CMS_BALLOT(S > n1) & CMS_BALLOT(S > n2) & CMS_BALLOT(S > n3) &
                                     CMS_BALLOT(S > n4) & CMS_BALLOT(S > n5) & CMS_BALLOT(S > n6) & CMS_BALLOT(S > n7);
    if (__builtin_amdgcn_inverse_ballot_w64(livem & ~keepm)) list[k] = 0xFFFFu;
    n_all += __popcll(keepm);
    n_ini += __popcll(keepm & CMS_BALLOT(S >= g.ini_th));
  }
  const int n_emit = n_ini > 0 ? n_ini : n_all;
  if (n_emit == 0) return;            // cell_cnt was zeroed before the launch
  // every cell owns a fixed slot: no atomics, no wait on a returning atomic; k_quadtree compacts the slots of its level
  const size_t cell = (size_t)b * g.total_cells + cid;
  if (lane == 0) cell_cnt[cell] = min(n_emit, g.cell_cap);
  uint32_t* out = cell_cand + cell * g.cell_cap;
  const int basepos = 0;
  const bool use_ini = n_ini > 0;
  int off = 0;
  for (int base = 0; base < L; base += 64) {
    const int k = base + lane;
    bool emit = false;
    int S = 0, px = 0, py = 0;
    if (k < L && list[k] != 0xFFFFu) {
      const int code = list[k];
      py = code >> 6; px = code & 63;
      S = sc[__mul24(py + 1, ss) + px + 1];
      emit = use_ini ? (S >= g.ini_th) : true;
    }
    const unsigned long long m = CMS_BALLOT(emit);
    if (emit) {
      const int pos = basepos + off + LANE_PREFIX(m);
      if (pos < g.cell_cap) out[pos] = (uint32_t)(ex0 + px) | ((uint32_t)(ey0 + py) << 12) | ((uint32_t)S << 24);
      else *overflow = 1;
    }
    off += __popcll(m);
  }
}

// ------------------------------------------------------------------------------------------------ quadtree
extern "C" __global__ void __launch_bounds__(512)
k_quadtree(CmsGeom g, const uint32_t* __restrict__ cell_cand, const int* __restrict__ cell_cnt, uint32_t* __restrict__ cand,
           int* __restrict__ cand_cnt, int* __restrict__ overflow, uint16_t* __restrict__ node_of,
           uint32_t* __restrict__ qt_out, int* __restrict__ qt_cnt, int* __restrict__ walk_cnt) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int l = blockIdx.x, b = blockIdx.y;
  if (walk_cnt && l == 0 && b == 0 && threadIdx.x == 0) *walk_cnt = 0;      // length of the batch's key-point walk (k_cull appends to it, k_describe reads it)
  const CmsLevel& lv = g.lv[l];
  const int maxn = g.qt_maxn;
  QtWork w;
  uint8_t* p = smem;
  w.maxn = maxn;
  w.childcnt = reinterpret_cast<uint32_t*>(p); p += 16 * (size_t)maxn;
  w.rect[0] = reinterpret_cast<QtRect*>(p); p += 8 * (size_t)maxn;
  w.rect[1] = reinterpret_cast<QtRect*>(p); p += 8 * (size_t)maxn;
  w.childpos = reinterpret_cast<uint16_t*>(p); p += 8 * (size_t)maxn;
  w.cnt[0] = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)maxn;
  w.cnt[1] = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)maxn;
  w.s0 = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)maxn;
  w.s1 = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)maxn;
  w.skey = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)maxn;
  w.part = reinterpret_cast<uint32_t*>(p); p += 4 * 512;
  w.sc = reinterpret_cast<int*>(p); p += 64;
  w.proc = reinterpret_cast<uint16_t*>(p); p += 2 * (size_t)maxn;
  w.flag = p; p += maxn;
  w.isex = p; p += maxn;
  QtParams P;
  // ---- gather this level's per-cell candidate slots into one contiguous list (cell-major order), no atomics:
  // chunks of 512 cells, block-wide exclusive scan of the cell counts, every thread copies its cell's entries.
  uint32_t* cwr = cand + (size_t)b * g.cand_total + lv.cand_off;
  {
    const int ncells = lv.nCols * lv.nRows;
    const int* cc = cell_cnt + (size_t)b * g.total_cells + lv.cell0;
    const uint32_t* cs = cell_cand + ((size_t)b * g.total_cells + lv.cell0) * g.cell_cap;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // thread t owns the cells [t * per, (t + 1) * per): one block-wide scan for the whole level (not one per 512 cells), all
    // count loads in flight together, then every thread copies its cells' entries with four loads in flight
    const int per = (ncells + 511) / 512;
    const int cb = (int)threadIdx.x * per, ce = min(cb + per, ncells);
    int mine = 0;
    for (int cell = cb; cell < ce; ++cell) mine += cc[cell];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) w.part[wv] = (uint32_t)incl;
    __syncthreads();
    int wbase = 0, running = 0;
    for (int q = 0; q < 8; ++q) { const int v = (int)w.part[q]; if (q < wv) wbase += v; running += v; }
    int off = wbase + incl - mine;
    for (int cell = cb; cell < ce; ++cell) {
      const int cnt = cc[cell];
      const uint32_t* src = cs + (size_t)cell * g.cell_cap;
      int i = 0;
      for (; i + 3 < cnt; i += 4) {
        const uint32_t v0 = src[i], v1 = src[i + 1], v2 = src[i + 2], v3 = src[i + 3];
        if (off + i + 3 < lv.cand_cap) { cwr[off + i] = v0; cwr[off + i + 1] = v1; cwr[off + i + 2] = v2; cwr[off + i + 3] = v3; }
        else *overflow = 1;
      }
      for (; i < cnt; ++i) {
        if (off + i < lv.cand_cap) cwr[off + i] = src[i];
        else *overflow = 1;
      }
      off += cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) cand_cnt[b * g.nlevels + l] = min(running, lv.cand_cap);
    P.n = min(running, lv.cand_cap);
  }
  __syncthreads();
  P.N = lv.quota;
  P.width = lv.w - 2 * CMS_MINB; P.height = lv.h - 2 * CMS_MINB;
  P.minB = CMS_MINB; P.wCell = lv.wCell; P.hCell = lv.hCell; P.nCols = lv.nCols;
  const uint32_t* c = cand + (size_t)b * g.cand_total + lv.cand_off;
  uint16_t* no = node_of + (size_t)b * g.cand_total + lv.cand_off;
  uint32_t* out = qt_out + (size_t)b * g.kp_cap + lv.kp_off;
  if (g.dbg_stop == 30) { if (threadIdx.x == 0) qt_cnt[b * g.nlevels + l] = 0; return; }      // developer switch: gather only
  const int S = qt_run(P, c, no, w, out, g.dbg_stop >= 31 ? g.dbg_stop - 30 : 0);
  if (threadIdx.x == 0) qt_cnt[b * g.nlevels + l] = S;
}

#define CMS_ORDER_MAX 65535      /* key points per frame the processing order of k_describe is built for (16-bit index) */
// ------------------------------------------------------------------------------------------------ cull + compaction
// One workgroup per frame walks the levels in order; survivors keep (level, list) order (ORBExtractor.cpp:875-921).
extern "C" __global__ void __launch_bounds__(256)
k_cull(CmsGeom g, const uint32_t* __restrict__ qt_out, const int* __restrict__ qt_cnt, const uint8_t* __restrict__ mask,
       int mstride, CmsKeyPoint* __restrict__ kps, uint32_t* __restrict__ aux, int* __restrict__ kp_cnt, uint32_t* __restrict__ walk,
       uint32_t* __restrict__ walk_aux, int* __restrict__ walk_cnt, float* __restrict__ rays) {
  __shared__ int wsum[4];
  __shared__ int s_walk_base;
  __shared__ uint32_t skey[512];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int total = 0;
  const float Ff = (float)g.F;
  for (int l = 0; l < g.nlevels; ++l) {
    const CmsLevel& lv = g.lv[l];
    const int m = qt_cnt[b * g.nlevels + l];
    const uint32_t* src = qt_out + (size_t)b * g.kp_cap + lv.kp_off;
    for (int base = 0; base < m; base += 256) {
      const int k = base + tid;
      bool keep = false;
      float px = 0.f, py = 0.f;
      uint32_t e = 0;
      if (k < m) {
        e = src[k];
        px = (float)(int)(e & 0xFFF) * lv.scale;           // keypoint.pt * scale (Point2f * float)
        py = (float)(int)((e >> 12) & 0xFFF) * lv.scale;
        const float fi = px / Ff, fj = py / Ff;             // FaceInCubemap(Point2f): float / int
        const bool face = (fi >= 0 && fi < 1 && fj >= 1 && fj < 2) || (fi >= 1 && fi < 2 && fj >= 0 && fj < 3) ||
                          (fi >= 2 && fi < 3 && fj >= 1 && fj < 2);
        const int ix = (int)(px + 0.5f), iy = (int)(py + 0.5f);
        keep = face && !(px < 0 || ix >= g.W || py < 0 || iy >= g.W);
        if (keep) keep = mask[(size_t)iy * mstride + ix] != 0;
      }
      const unsigned long long mk = CMS_BALLOT(keep);
      if (lane == 0) wsum[wave] = __popcll(mk);
      __syncthreads();
      int off = total;
      for (int q = 0; q < wave; ++q) off += wsum[q];
      const int chunk = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      if (keep) {
        const int pos = off + LANE_PREFIX(mk);
        CmsKeyPoint kp;
        kp.x = px; kp.y = py; kp.size = lv.patch_size; kp.angle = -1.f; kp.response = (float)(e >> 24); kp.octave = l;
        kps[(size_t)b * g.kp_cap + pos] = kp;
        aux[(size_t)b * g.kp_cap + pos] = (e & 0xFFFFFFu) | ((uint32_t)l << 24);
        if (rays) {
          // Frame::ComputeKeyPointRays -> CamModelGeneral::TransformCubemapToRays (src/Frame.cpp:746-760, include/CamModelGeneral.h:494-513): the pixel's
          // place inside its face in double, through the face intrinsics fx = fy = cx = cy = F / 2, cast to float, turned into rig axes
          // (cvtFacesToRig, :388-414), normalised with a double norm and a double reciprocal (cv::norm, Vec3f * double)
          const float fi = px / Ff, fj = py / Ff;
          const double Fd = (double)g.F, hF = Fd / 2.0;
          double di = (double)px, dj = (double)py;
          di = di - (double)((int)(di / Fd) * g.F); dj = dj - (double)((int)(dj / Fd) * g.F);
          const float lx = (float)((di - hF) * 1.0 / hF), ly = (float)((dj - hF) * 1.0 / hF), lz = 1.0f;
          float rx, ry, rz;
          if (fi >= 0 && fi < 1) { rx = -lz; ry = ly; rz = lx; }                       // LEFT   (FaceInCubemap(Point2f), :445-456: the order of its tests)
          else if (fi >= 1 && fi < 2 && fj >= 0 && fj < 1) { rx = lx; ry = -lz; rz = ly; }   // UPPER
          else if (fi >= 1 && fi < 2 && fj >= 1 && fj < 2) { rx = lx; ry = ly; rz = lz; }    // FRONT
          else if (fi >= 1 && fi < 2) { rx = lx; ry = lz; rz = -ly; }                  // LOWER
          else { rx = lz; ry = ly; rz = -lx; }                                         // RIGHT
          const double nrm = sqrt((double)rx * (double)rx + (double)ry * (double)ry + (double)rz * (double)rz);
          const double sc = nrm > 0 ? 1. / nrm : 0.;
          float* rp = rays + 3 * ((size_t)b * g.kp_cap + pos);
          rp[0] = (float)((double)rx * sc); rp[1] = (float)((double)ry * sc); rp[2] = (float)((double)rz * sc);
        }
      }
      total += chunk;
      __syncthreads();
    }
  }
  if (tid == 0) kp_cnt[b] = total;
  // ---- the order k_describe WORKS in (its output keeps the list order above): by level and band of 64 rows.  A key point's 43 patch rows
  // are 48-byte pieces of 128-byte lines; walked in list order (the octree's node order, scattered over the level) every key point fetched
  // its ~44 lines from HBM on its own -- 3.2x the patch bytes (profiles/r01, r02_describe_order.txt).  Counting sort on (level, band) in LDS;
  // the frame's sorted list is then APPENDED to one walk of the whole batch (one atomic per frame: the frames' order in the walk does not
  // matter, a frame's key points stay together).  k_describe cuts the walk into eight contiguous segments, one per XCD -- neighbours in the
  // walk share patch rows and meet in one L2 -- with equal numbers of key points per XCD whatever the frames hold (round 2's version pinned
  // frame f to XCD f mod 8).  Optional (CMS_DESC_SPATIAL_ORDER=1): see cms_ctx_create for the measurements.
  if (walk) {
    __syncthreads();                                       // the block's own aux[] stores; skey free
    for (int i = tid; i < 512; i += 256) skey[i] = 0;
    if (tid == 0) s_walk_base = atomicAdd(walk_cnt, total);
    __syncthreads();
    for (int i = tid; i < total; i += 256) {
      const uint32_t a = aux[(size_t)b * g.kp_cap + i];
      atomicAdd(&skey[min(255u, (a >> 24) * 32 + (((a >> 12) & 0xFFF) >> 6))], 1u);
    }
    __syncthreads();
    if (tid < 64) {                                        // exclusive prefix over the 256 buckets: four per lane
      const uint32_t c0 = skey[4 * tid], c1 = skey[4 * tid + 1], c2 = skey[4 * tid + 2], c3 = skey[4 * tid + 3];
      uint32_t incl = c0 + c1 + c2 + c3;
      for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
      const uint32_t base = incl - (c0 + c1 + c2 + c3);
      skey[256 + 4 * tid] = base; skey[256 + 4 * tid + 1] = base + c0; skey[256 + 4 * tid + 2] = base + c0 + c1; skey[256 + 4 * tid + 3] = base + c0 + c1 + c2;
    }
    __syncthreads();
    const size_t wb = (size_t)s_walk_base;
    for (int i = tid; i < total; i += 256) {
      const uint32_t a = aux[(size_t)b * g.kp_cap + i];
      const uint32_t pos = atomicAdd(&skey[256 + min(255u, (a >> 24) * 32 + (((a >> 12) & 0xFFF) >> 6))], 1u);
      walk[wb + pos] = ((uint32_t)b << 16) | (uint32_t)i;  // frame | key point (both < 65536: CMS_ORDER_MAX)
      walk_aux[wb + pos] = a;
    }
  }
}

// ------------------------------------------------------------------------------------------------ orientation + rBRIEF
// One wavefront per surviving key point.  A 43x43 raw patch (taps reach +-18, the 7-tap blur +-3 more) is staged in
// LDS with REFLECT_101 at the level edges, IC_Angle runs on its centre, then a separable fixed-point 7x7 Gaussian
// ([18,34,49,55,49,34,18], (sum+32768)>>16) produces the 37x37 blurred neighbourhood the 512 steered taps read.
#define PR 21
#define PW 43
#define PS 48
#define BW 37
#define BS 40
#define RW 40            /* row-sum stride (16-bit entries): 10 quads per row */
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}
#ifndef CMS_DESC_WPB
#define CMS_DESC_WPB 1
#endif
// Byte masks of the radius-15 disc rows: row |v| keeps bytes u = -umax .. umax of its 31 (index u + 15); eight dwords per row.
#define DM(um, j) ((uint32_t)((4 * (j) + 0 >= 15 - (um) && 4 * (j) + 0 <= 15 + (um)) ? 0x000000FFu : 0u) | \
                   (uint32_t)((4 * (j) + 1 >= 15 - (um) && 4 * (j) + 1 <= 15 + (um)) ? 0x0000FF00u : 0u) | \
                   (uint32_t)((4 * (j) + 2 >= 15 - (um) && 4 * (j) + 2 <= 15 + (um)) ? 0x00FF0000u : 0u) | \
                   (uint32_t)((4 * (j) + 3 >= 15 - (um) && 4 * (j) + 3 <= 15 + (um)) ? 0xFF000000u : 0u))
#define DMROW(um) DM(um, 0), DM(um, 1), DM(um, 2), DM(um, 3), DM(um, 4), DM(um, 5), DM(um, 6), DM(um, 7)
__device__ __constant__ __align__(16) uint32_t k_disc_mask[16 * 8] = {
  DMROW(15), DMROW(15), DMROW(15), DMROW(15), DMROW(14), DMROW(14), DMROW(14), DMROW(13), DMROW(13), DMROW(12), DMROW(11), DMROW(10),
  DMROW(9), DMROW(8), DMROW(6), DMROW(3)};
#undef DMROW
#undef DM
// SSE2: the column pass follows an x86 OpenCV's float tie rule (cms_set_gaussian_mode).  A compile-time switch: as a run-time test inside
// the column loop it cost 30 of the loop's 71 vector instructions per step even when off (short predicated blocks are issued, not skipped).
template <bool SSE2>
__device__ __forceinline__ void describe_body(const uint8_t* __restrict__ pyr, size_t pyr_bytes, const CmsGeom& g, CmsKeyPoint* __restrict__ kps,
           const uint32_t* __restrict__ aux, const int* __restrict__ kp_cnt, const float* __restrict__ pattern,
           uint8_t* __restrict__ desc, const uint32_t* __restrict__ walk, const uint32_t* __restrict__ walk_aux, const int* __restrict__ walk_cnt) {
  // CMS_DESC_WPB key points per workgroup, one per wavefront, no data shared between them (wave-level synchronisation only)
  __shared__ __align__(16) uint8_t raw4[CMS_DESC_WPB][PW * PS + 16];
  __shared__ __align__(16) uint32_t rowp4[CMS_DESC_WPB][((PW + 1) / 2) * RW];      // row sums of rows 2m (low half) and 2m + 1 (high half), column by column
  __shared__ __align__(4) uint8_t blr4[CMS_DESC_WPB][BW * BS];
#if CMS_DESC_WPB == 1
  const int wave = 0, lane = threadIdx.x;
#else
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#endif
  uint8_t* raw = raw4[wave]; uint32_t* rowp = rowp4[wave]; uint8_t* blr = blr4[wave];
  // walk != nullptr (1-D grid): workgroups go round-robin to the 8 XCDs; XCD x works through the x-th eighth of the batch's walk (k_cull) in
  // order, so that patch rows shared by neighbouring key points are fetched into that XCD's L2 once
  int b, k;
  uint32_t a;
  if (walk) {
    const int L = blockIdx.x * CMS_DESC_WPB + wave, t = L >> 3;
    const int N = *walk_cnt, per = (N + 7) >> 3;
    const int pos = (L & 7) * per + t;
    if (t >= per || pos >= N) return;
    const uint32_t w = walk[pos];
    a = walk_aux[pos];
    b = (int)(w >> 16); k = (int)(w & 0xFFFFu);
  } else {
    b = blockIdx.y; k = blockIdx.x * CMS_DESC_WPB + wave;
    if (k >= kp_cnt[b]) return;
    a = aux[(size_t)b * g.kp_cap + k];
  }
  const int cx = a & 0xFFF, cy = (a >> 12) & 0xFFF, l = a >> 24;
  const CmsLevel& lv = g.lv[l];
  const uint8_t* img = pyr + (size_t)b * pyr_bytes + lv.off;
  int off = 0;                       // patch column c lives at raw[r * PS + off + c]
  if (cx - PR >= 0 && cx + PR < lv.w && cy - PR >= 0 && cy + PR < lv.h) {
    // interior (the normal case): 43 rows x 12 aligned dwords, all 9 loads of a lane in flight before the first LDS store
    const int ax = (cx - PR) & ~3;
    off = cx - PR - ax;
    const uint8_t* gp = img + (size_t)(cy - PR) * lv.stride + ax;
    uint32_t tmp[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      // idx / 12 as a 24-bit multiply by the reciprocal (exact for idx < 580); row offset as a 24-bit multiply, 32-bit offset
      const int idx = lane + 64 * k, r = __mul24(idx, 5462) >> 16, c = idx - r * 12;
      tmp[k] = idx < PW * 12 ? *reinterpret_cast<const uint32_t*>(gp + (uint32_t)(__mul24(r, lv.stride) + 4 * c)) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int idx = lane + 64 * k;
      if (idx < PW * 12) reinterpret_cast<uint32_t*>(raw)[idx] = tmp[k];     // PS == 48 == 12 dwords per row
    }
  } else {
    for (int idx = lane; idx < PW * PW; idx += 64) {   // patch crosses the level edge: REFLECT_101, byte by byte
      const int r = idx / PW, c = idx - r * PW;
      const int yy = reflect101(cy - PR + r, lv.h), xx = reflect101(cx - PR + c, lv.w);
      raw[r * PS + c] = img[(size_t)yy * lv.stride + xx];
    }
  }
  const uint8_t* rawp = raw + off;
  WAVE_SYNC();
  // ---- IC_Angle: intensity centroid over the radius-15 disc (umax of ORBExtractor.cpp:426-441)
  // Row v of the disc = 31 bytes around the centre: eight dwords (byte-aligned with the patch offset), bytes outside |u| <= umax(v)
  // masked off, then sum p by v_sad_u8 and sum (u + 15) p by v_dot4_u32_u8 -- 35 vector instructions per row instead of a 31-step loop.
  int m10 = 0, m01 = 0;
  if (lane < 31) {
    const int v = lane - 15, av = v < 0 ? -v : v;
    const int b0 = off + (PR - 15);                                         // first byte of the disc row inside the staged row
    const uint32_t* row32 = reinterpret_cast<const uint32_t*>(raw + (PR + v) * PS) + (b0 >> 2);
    const uint32_t sh = (uint32_t)(b0 & 3);
    uint32_t d[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) d[j] = row32[j];
    const uint4 mk0 = reinterpret_cast<const uint4*>(k_disc_mask)[2 * av], mk1 = reinterpret_cast<const uint4*>(k_disc_mask)[2 * av + 1];
    const uint32_t mk[8] = {mk0.x, mk0.y, mk0.z, mk0.w, mk1.x, mk1.y, mk1.z, mk1.w};
    uint32_t sum = 0, mom = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t w = __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh) & mk[j];          // disc bytes 4j .. 4j+3  (u = 4j+i - 15)
      sum = __builtin_amdgcn_sad_u8(w, 0u, sum);
      mom = __builtin_amdgcn_udot4(w, (uint32_t)(4 * j) * 0x01010101u + 0x03020100u, mom, false);
    }
    m10 = (int)mom - 15 * (int)sum;
    m01 = v * (int)sum;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
  const float angle = cms_fast_atan2((float)m01, (float)m10);
  // ---- separable Gaussian, rows then columns
  // Row pass, four outputs per lane and step: 12 patch bytes come in as four aligned dwords (funnel-shifted by the patch's
  // byte offset), every output is two v_dot4_u32_u8 -- the row sum is at most 255 * 257 = 65535, so it is stored in 16 bits.
  // Outputs c = 37..39 of a row are computed from bytes that exist and are never read.
  {
    const uint32_t* raw32 = reinterpret_cast<const uint32_t*>(raw);
    for (int task = lane; task < PW * (RW / 4); task += 64) {
      static_assert(RW == 40, "reciprocals below are for RW / 4 == 10 and RW / 2 == 20");
      const int r = __mul24(task, 6554) >> 16, q = task - r * (RW / 4);           // task / 10, exact for task < 494
      const uint32_t* dp = raw32 + r * (PS / 4) + q;
      const uint32_t d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3];
      const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, off), w1 = __builtin_amdgcn_alignbyte(d2, d1, off),
                     w2 = __builtin_amdgcn_alignbyte(d3, d2, off);                        // patch bytes 4q .. 4q+11
      // seven taps of one output = two 4-byte dot products: bytes c .. c+3 against (18, 34, 49, 55), bytes c+4 .. c+7 against (49, 34, 18, 0)
      const uint32_t ka = 18u | (34u << 8) | (49u << 16) | (55u << 24), kb = 49u | (34u << 8) | (18u << 16);
      uint32_t o[4];
      o[0] = __builtin_amdgcn_udot4(w1, kb, __builtin_amdgcn_udot4(w0, ka, 0u, false), false);
#pragma unroll
      for (int i = 1; i < 4; ++i) {
        const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, i), hi = __builtin_amdgcn_alignbyte(w2, w1, i);
        o[i] = __builtin_amdgcn_udot4(hi, kb, __builtin_amdgcn_udot4(lo, ka, 0u, false), false);
      }
      // row sums <= 255 * 257 = 65535: 16 bits.  Rows 2m and 2m + 1 share a dword per column, so that the column pass can take two
      // vertical taps per v_dot2_u32_u16
      uint16_t* dst = reinterpret_cast<uint16_t*>(rowp + (r >> 1) * RW + 4 * q) + (r & 1);
      dst[0] = (uint16_t)o[0]; dst[2] = (uint16_t)o[1]; dst[4] = (uint16_t)o[2]; dst[6] = (uint16_t)o[3];
    }
  }
  WAVE_SYNC();
  // Column pass, four neighbouring columns per lane and step.  Output row ro needs the row sums ro .. ro + 6: four row PAIRS starting at pair
  // ro >> 1 -- for an even ro the taps (18 34)(49 55)(49 34)(18 -), for an odd one (- 18)(34 49)(55 49)(34 18) -- four dot products of
  // 16-bit pairs per output instead of seven multiplies and as many unpacking operations.
  for (int task = lane; task < BW * (RW / 4); task += 64) {
    const int r = __mul24(task, 6554) >> 16, q = task - r * (RW / 4);             // task / 10, exact for task < 494
    const uint4* p = reinterpret_cast<const uint4*>(rowp + (r >> 1) * RW) + q;
    const uint4 e0 = p[0], e1 = p[RW / 4], e2 = p[2 * (RW / 4)], e3 = p[3 * (RW / 4)];
    const bool odd = (r & 1) != 0;
    const uint32_t w0 = odd ? (18u << 16) : (18u | (34u << 16)), w1 = odd ? (34u | (49u << 16)) : (49u | (55u << 16)),
                   w2 = odd ? (55u | (49u << 16)) : (49u | (34u << 16)), w3 = odd ? (34u | (18u << 16)) : 18u;
    auto col = [&](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) -> int {
      uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a0), __builtin_bit_cast(us2_t, w0), 0u, false);
      acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a1), __builtin_bit_cast(us2_t, w1), acc, false);
      acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a2), __builtin_bit_cast(us2_t, w2), acc, false);
      acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a3), __builtin_bit_cast(us2_t, w3), acc, false);
      return (int)acc;
    };
    const int sv[4] = {col(e0.x, e1.x, e2.x, e3.x), col(e0.y, e1.y, e2.y, e3.y), col(e0.z, e1.z, e2.z, e3.z), col(e0.w, e1.w, e2.w, e3.w)};
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int v = min((sv[i] + 32768) >> 16, 255);
      if (SSE2) {
        // an x86 OpenCV <= 3.2 evaluates the column pass in float with round-half-to-EVEN for the columns its SSE2 loops cover
        // (x < width & ~3): every product and partial sum is exact in binary32, so the result
        // differs from the integer formula exactly on ties (sum mod 65536 == 32768) with an even quotient
        const int x0 = cx - 18 + 4 * q + i, xvec = lv.w & ~3;
        if (x0 < xvec && (sv[i] & 0xFFFF) == 0x8000 && ((sv[i] >> 16) & 1) == 0) v = min(sv[i] >> 16, 255);
      }
      out |= (uint32_t)v << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(blr + r * BS + 4 * q) = out;
  }
  WAVE_SYNC();
  // ---- steered BRIEF: lane i evaluates tests 4i .. 4i+3
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float sb, ca;
  cms_sincosf(angle * factorPI, &sb, &ca);
  const float4* pt4 = reinterpret_cast<const float4*>(pattern) + 4 * lane;      // float table: no int8 -> float conversions (32 per lane)
  const uint8_t* ctr = blr + 18 * BS + 18;
  int nib = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 pq = pt4[t];
    const float x0 = pq.x, y0 = pq.y, x1 = pq.z, y1 = pq.w;
    const int v0 = ctr[cms_cv_round(x0 * sb + y0 * ca) * BS + cms_cv_round(x0 * ca - y0 * sb)];
    const int v1 = ctr[cms_cv_round(x1 * sb + y1 * ca) * BS + cms_cv_round(x1 * ca - y1 * sb)];
    nib |= (v0 < v1 ? 1 : 0) << t;
  }
  const int other = __shfl_xor(nib, 1);
  if ((lane & 1) == 0) desc[((size_t)b * g.kp_cap + k) * 32 + (lane >> 1)] = (uint8_t)(nib | (other << 4));
  if (lane == 0) kps[(size_t)b * g.kp_cap + k].angle = angle;
}
extern "C" __global__ void __launch_bounds__(64 * CMS_DESC_WPB)
k_describe(const uint8_t* __restrict__ pyr, size_t pyr_bytes, CmsGeom g, CmsKeyPoint* __restrict__ kps,
           const uint32_t* __restrict__ aux, const int* __restrict__ kp_cnt, const float* __restrict__ pattern,
           uint8_t* __restrict__ desc, const uint32_t* __restrict__ walk, const uint32_t* __restrict__ walk_aux, const int* __restrict__ walk_cnt) {
  describe_body<false>(pyr, pyr_bytes, g, kps, aux, kp_cnt, pattern, desc, walk, walk_aux, walk_cnt);
}
extern "C" __global__ void __launch_bounds__(64 * CMS_DESC_WPB)
k_describe_sse2(const uint8_t* __restrict__ pyr, size_t pyr_bytes, CmsGeom g, CmsKeyPoint* __restrict__ kps,
                const uint32_t* __restrict__ aux, const int* __restrict__ kp_cnt, const float* __restrict__ pattern,
                uint8_t* __restrict__ desc, const uint32_t* __restrict__ walk, const uint32_t* __restrict__ walk_aux, const int* __restrict__ walk_cnt) {
  describe_body<true>(pyr, pyr_bytes, g, kps, aux, kp_cnt, pattern, desc, walk, walk_aux, walk_cnt);
}
