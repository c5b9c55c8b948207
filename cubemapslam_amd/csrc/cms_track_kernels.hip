// cms_track_kernels.hip -- "track local map" on the device: Frame::isInFrustum (src/Frame.cpp:197-249) for every local map point
// and ORBMatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBMatcher.cpp:50-128), the pair
// Tracking::SearchLocalPoints runs once per frame (src/Tracking.cpp:794-846).
//
//   k_in_frustum       one thread per map point: project with the frame's float pose, the five visibility tests, PredictScale;
//                      it also writes the GetFeaturesInArea query of the point (x, y, r, level-1, level; r < 0 = not in view), so the
//                      window query (cms_area_kernels.hip) follows without a host round trip
//   k_search_local     one workgroup per frame.  The reference's loop is a sequential greedy: a key point taken by map point i is
//                      skipped by every later map point (ORBMatcher.cpp:91-93, :121).  The same result is reached in parallel
//                      rounds: a map point is decided in the round in which it is the LOWEST undecided point among all undecided
//                      points sharing any of its still-free candidates -- then every earlier point that could take one of its
//                      candidates has already spoken.  Points decided in one round have disjoint free candidates, so their claims
//                      cannot collide; the lowest undecided point always qualifies, so the loop ends.  Hamming distances of all
//                      (point, candidate) pairs are computed once, before the rounds.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct CmsFrustumArgs {
  const float* pose15;      // per frame: Rcw (9, row major) | tcw (3) | Ow (3), the float members Frame::UpdatePoseMatrices leaves
  const int* mp_frame;      // frame of every map point (nullptr: all frame 0)
  int n;
  const float* P; const float* normal; const float* min_dist; const float* max_dist;   // mWorldPos, mNormalVector, mfMinDistance, mfMaxDistance
  float viewing_cos_limit, log_scale, th;
  int nlevels, F;
  int bounds_scaled;        // 1: min_dist / max_dist are MapPoint::GetMinDistanceInvariance() / GetMaxDistanceInvariance() (0.8f / 1.2f applied)
  float sf[16];             // mvScaleFactors
  uint8_t* in_view; float* proj_x; float* proj_y; int* level; float* view_cos;            // mbTrackInView, mTrackProjX/Y, mnTrackScaleLevel, mTrackViewCos
  float* qr; int* qmin; int* qmax;                                                      // window of SearchByProjection (ORBMatcher.cpp:69-75)
};

// Scale-invariance bounds of a map point (MapPoint.cpp:375-385) and mfMaxDistance itself, from what the caller handed over
// (cms_set_distance_bounds_mode): the raw members, or the public getters' values
__device__ __forceinline__ void track_distance_bounds(int scaled, float min_in, float max_in, float& minDistance, float& maxDistance, float& raw_max) {
  if (scaled) {
    maxDistance = max_in; minDistance = min_in;
    float r = (float)((double)max_in / (double)1.2f);
    const float lo = __uint_as_float(__float_as_uint(r) - 1u), hi = __uint_as_float(__float_as_uint(r) + 1u);
    if (r > 0.0f && __fmul_rn(1.2f, lo) == max_in) r = lo;
    else if (__fmul_rn(1.2f, r) != max_in && __fmul_rn(1.2f, hi) == max_in) r = hi;
    raw_max = r;
  } else {
    raw_max = max_in; maxDistance = __fmul_rn(1.2f, max_in); minDistance = __fmul_rn(0.8f, min_in);
  }
}

// CamModelGeneral::TransformRaysToCubemap (src/CamModelGeneral.cpp:95-154): face choice on float ratios, pixel through the double
// intrinsics (fx = fy = cx = cy = F / 2 are double members, so `_x * fx / _z + cx` is evaluated in double and narrowed on assignment)
__host__ __device__ __forceinline__ int track_rays_to_cubemap(int F, float x, float y, float z, float& up, float& vp) {
  const double f = F / 2.0;
  float lx, ly, lz, ox, oy;
  int face;
  if (z > 0 && x / z <= 1 && x / z >= -1 && y / z <= 1 && y / z >= -1) { face = 0; lx = x; ly = y; lz = z; ox = (float)F; oy = (float)F; }
  else if (x > 0 && y / x <= 1 && y / x >= -1 && z / x <= 1 && z / x >= -1) { face = 2; lx = -z; ly = y; lz = x; ox = (float)(2 * F); oy = (float)F; }
  else if (x < 0 && y / (-x) <= 1 && y / (-x) >= -1 && z / (-x) <= 1 && z / (-x) >= -1) { face = 1; lx = z; ly = y; lz = -x; ox = 0.0f; oy = (float)F; }
  else if (y > 0 && x / y <= 1 && x / y >= -1 && z / y <= 1 && z / y >= -1) { face = 4; lx = x; ly = -z; lz = y; ox = (float)F; oy = (float)(2 * F); }
  else if (y < 0 && x / (-y) <= 1 && x / (-y) >= -1 && z / (-y) <= 1 && z / (-y) >= -1) { face = 3; lx = x; ly = z; lz = -y; ox = (float)F; oy = 0.0f; }
  else { up = -1.0f; vp = -1.0f; return -1; }
  up = (float)((double)lx * f / (double)lz + f);
  vp = (float)((double)ly * f / (double)lz + f);
  if (up < 0 || up >= F || vp < 0 || vp >= F) return -1;
  if (ox != 0.0f) up += ox;          // the reference adds nothing on the first column / row of faces
  if (oy != 0.0f) vp += oy;
  return face;
}

extern "C" __global__ void __launch_bounds__(256) k_in_frustum(CmsFrustumArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float* ps = a.pose15 + 15 * (size_t)(a.mp_frame ? a.mp_frame[i] : 0);
  uint8_t vis = 0; float u = -1.0f, v = -1.0f, vc = 0.0f; int lvl = -1;
  const float px = a.P[3 * (size_t)i], py = a.P[3 * (size_t)i + 1], pz = a.P[3 * (size_t)i + 2];
  float Pc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {      // mRcw*P+mtcw: float products and sums left to right, + tcw through double (cv::gemm's small path)
    float t = __fmul_rn(ps[3 * r], px);
    t = __fadd_rn(t, __fmul_rn(ps[3 * r + 1], py));
    t = __fadd_rn(t, __fmul_rn(ps[3 * r + 2], pz));
    Pc[r] = (float)((double)t * 1.0 + (double)ps[9 + r] * 1.0);
  }
  const float mnMax = (float)(3 * a.F);
  do {
    const int face = track_rays_to_cubemap(a.F, Pc[0], Pc[1], Pc[2], u, v);
    if (face < 0) break;
    if (u < 0.0f || u > mnMax || v < 0.0f || v > mnMax) break;
    // mfMaxDistance itself (PredictScale's ratio) and the two invariance bounds.  Handed the public getters' values, the bounds are used as
    // they are and mfMaxDistance is recovered as the float r with 1.2f * r == bound (the product rounds up to two neighbouring r onto one
    // bound when it crosses a power of two; the smaller one is taken -- PredictScale can then differ from the reference only where
    // log(ratio) / log(scale factor) sits within an ulp of an integer)
    float maxd, maxDistance, minDistance;
    track_distance_bounds(a.bounds_scaled, a.min_dist[i], a.max_dist[i], minDistance, maxDistance, maxd);
    const float PO[3] = {__fsub_rn(px, ps[12]), __fsub_rn(py, ps[13]), __fsub_rn(pz, ps[14])};
    double s = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) s = __dadd_rn(s, __dmul_rn((double)PO[k], (double)PO[k]));
    const float dist = (float)sqrt(s);
    if (dist < minDistance || dist > maxDistance) break;
    double d = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) d = __dadd_rn(d, __dmul_rn((double)PO[k], (double)a.normal[3 * (size_t)i + k]));
    vc = (float)(d / (double)dist);
    if (vc < a.viewing_cos_limit) break;
    // MapPoint::PredictScale (MapPoint.cpp:404-419): ceil(logf(ratio) / mfLogScaleFactor); the float log is taken as the rounded
    // double log (glibc's logf is within 0.82 ulp of it; the two can only part where the quotient sits within an ulp of an integer)
    const float ratio = maxd / dist;
    const float lg = (float)log((double)ratio);
    int ns = (int)ceilf(lg / a.log_scale);
    if (ns < 0) ns = 0; else if (ns >= a.nlevels) ns = a.nlevels - 1;
    lvl = ns; vis = 1;
  } while (false);
  if (!vis) { u = -1.0f; v = -1.0f; vc = 0.0f; }
  a.in_view[i] = vis; a.proj_x[i] = u; a.proj_y[i] = v; a.level[i] = lvl; a.view_cos[i] = vc;
  if (a.qr) {
    float r = (double)vc > 0.998 ? 2.5f : 4.0f;                    // ORBMatcher::RadiusByViewingCos (ORBMatcher.cpp:380-386)
    if (a.th != 1.0f) r = __fmul_rn(r, a.th);
    a.qr[i] = vis ? __fmul_rn(r, a.sf[lvl]) : -1.0f;
    a.qmin[i] = lvl - 1; a.qmax[i] = lvl;
  }
}

struct CmsSearchLocalArgs {
  const int* mp_off;          // B + 1: map points [mp_off[f], mp_off[f+1]) belong to frame f, in the reference's list order
  const uint4* mp_desc;       // 32 B per map point (MapPoint::GetDescriptor)
  const int* cand_off; const int* cand_idx;   // CSR from the window query; indices are batch rows (frame * kp_cap + key point)
  const uint4* t_desc; const CmsKeyPoint* kp; int kp_cap;
  uint16_t* pair_dist;        // one entry per candidate pair (scratch)
  int* kp_mp;                 // per batch row, in/out: >= 0 = holds a map point with observations (skipped), new matches get the point's index
  int* mp_match;              // per map point: matched batch row or -1
  int* rounds;                // per frame (optional): rounds the greedy needed
  float nnratio; int th_high;
  int frame0;                 // key-point rows of workgroup w are those of frame frame0 + w (single-frame entry point)
  const int* total; int cap;  // optional: *total > cap means the candidate lists were cut short -- do nothing, the host repeats the query
};

__device__ __forceinline__ int track_hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

#define CMS_TRACK_KPMAX 4096
extern "C" __global__ void __launch_bounds__(1024) k_search_local(CmsSearchLocalArgs a) {
  __shared__ int min_open[CMS_TRACK_KPMAX];           // lowest undecided map point that still wants key point k
  const int f = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  if (a.total && *a.total > a.cap) return;
  const int m0 = a.mp_off[f], m1 = a.mp_off[f + 1];
  const int row0 = (f + a.frame0) * a.kp_cap;
  // ---- distances of all pairs of this frame, 4 lanes per map point
  {
    const int sub = tid & 3;
    for (int i = m0 + (tid >> 2); i < m1; i += nt >> 2) {
      const int c0 = a.cand_off[i], c1 = a.cand_off[i + 1];
      if (c0 == c1) continue;
      const uint4 q0 = a.mp_desc[2 * (size_t)i], q1 = a.mp_desc[2 * (size_t)i + 1];
      for (int c = c0 + sub; c < c1; c += 4) {
        const size_t row = (size_t)a.cand_idx[c];
        a.pair_dist[c] = (uint16_t)track_hamming256(q0, q1, a.t_desc[2 * row], a.t_desc[2 * row + 1]);
      }
    }
  }
  for (int i = m0 + tid; i < m1; i += nt) a.mp_match[i] = a.cand_off[i] == a.cand_off[i + 1] ? -1 : -2;   // -2 = undecided
  __syncthreads();
  int round = 0;
  for (;;) {
    for (int k = tid; k < a.kp_cap; k += nt) min_open[k] = 0x7FFFFFFF;
    __syncthreads();
    for (int i = m0 + tid; i < m1; i += nt) {
      if (a.mp_match[i] != -2) continue;
      for (int c = a.cand_off[i]; c < a.cand_off[i + 1]; ++c) {
        const int row = a.cand_idx[c];
        if (a.kp_mp[row] < 0) atomicMin(&min_open[row - row0], i);
      }
    }
    __syncthreads();
    // which of my points are the lowest claimant of every free candidate they have?  (decided before anybody writes a claim; a ready
    // point is marked -3 in its own mp_match entry -- only its owner thread reads that entry -- so a thread may own any number of points)
    int open = 0;
    for (int i = m0 + tid; i < m1; i += nt) {
      if (a.mp_match[i] != -2) continue;
      bool ok = true;
      for (int c = a.cand_off[i]; c < a.cand_off[i + 1] && ok; ++c) {
        const int row = a.cand_idx[c];
        ok = a.kp_mp[row] >= 0 || min_open[row - row0] == i;
      }
      if (ok) a.mp_match[i] = -3; else ++open;
    }
    __syncthreads();
    for (int i = m0 + tid; i < m1; i += nt) {
      if (a.mp_match[i] != -3) continue;
      // the reference's scan (ORBMatcher.cpp:84-113) over the candidates in GetFeaturesInArea's order
      int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestRow = -1;
      for (int c = a.cand_off[i]; c < a.cand_off[i + 1]; ++c) {
        const int row = a.cand_idx[c];
        if (a.kp_mp[row] >= 0) continue;
        const int dist = a.pair_dist[c];
        if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = a.kp[row].octave; bestRow = row; }
        else if (dist < bestDist2) { bestLevel2 = a.kp[row].octave; bestDist2 = dist; }
      }
      int m = -1;
      // nnratio < 0: no second-best test (the frame-to-frame SearchByProjection, ORBMatcher.cpp:207)
      const bool ratio_ok = a.nnratio < 0.0f || !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(a.nnratio, (float)bestDist2));
      if (bestDist <= a.th_high && ratio_ok) m = bestRow;
      if (m >= 0) a.kp_mp[m] = i;
      a.mp_match[i] = m;
    }
    ++round;
    if (__syncthreads_count(open > 0) == 0) break;
  }
  if (a.rounds && tid == 0) a.rounds[f] = round;
}


// ---- ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBMatcher.cpp:130-251), the matcher of TrackWithMotionModel:
// k_project_last projects the last frame's map points with the current pose and writes their windows; the window query and k_search_local
// (nnratio < 0: best distance only) follow; k_rot_filter applies the rotation-consistency histogram (ComputeThreeMaxima, :905-946).
struct CmsProjectLastArgs {
  const float* pose12;        // per current frame: Rcw (9, row major) | tcw (3)
  const int* q_frame;         // current frame searched by every query (nullptr: 0)
  int n; const uint8_t* valid; const float* Xw; const int* oct;
  float th, cos_fov; int F; float sf[16];
  float* qx; float* qy; float* qr; int* qmin; int* qmax;
};
extern "C" __global__ void __launch_bounds__(256) k_project_last(CmsProjectLastArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  float u = -1.0f, v = -1.0f, r = -1.0f;
  const int o = a.oct[i];
  if (a.valid[i]) {
    const float* ps = a.pose12 + 12 * (size_t)(a.q_frame ? a.q_frame[i] : 0);
    const float p0 = a.Xw[3 * (size_t)i], p1 = a.Xw[3 * (size_t)i + 1], p2 = a.Xw[3 * (size_t)i + 2];
    float xc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float t = __fmul_rn(ps[3 * k], p0);
      t = __fadd_rn(t, __fmul_rn(ps[3 * k + 1], p1));
      t = __fadd_rn(t, __fmul_rn(ps[3 * k + 2], p2));
      xc[k] = (float)((double)t * 1.0 + (double)ps[9 + k] * 1.0);
    }
    if (!(xc[2] < a.cos_fov) && track_rays_to_cubemap(a.F, xc[0], xc[1], xc[2], u, v) >= 0) r = __fmul_rn(a.th, a.sf[o & 15]);
  }
  a.qx[i] = u; a.qy[i] = v; a.qr[i] = r; a.qmin[i] = o - 1; a.qmax[i] = o + 1;
}

struct CmsRotFilterArgs {
  const int* mp_off; const float* last_angle; const CmsKeyPoint* kp; int* kp_mp; int* mp_match; int* n_matches; int check_orientation;
  const int* total; int cap;  // as in CmsSearchLocalArgs
};
extern "C" __global__ void __launch_bounds__(1024) k_rot_filter(CmsRotFilterArgs a) {
  __shared__ int hist[32];
  __shared__ int keep[3];
  __shared__ int s_n;
  const int f = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  if (a.total && *a.total > a.cap) return;
  const int m0 = a.mp_off[f], m1 = a.mp_off[f + 1];
  if (tid < 32) hist[tid] = 0;
  if (tid == 0) s_n = 0;
  __syncthreads();
  auto bin_of = [&](int i, int row) {
    float rot = __fsub_rn(a.last_angle[i], a.kp[row].angle);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, 1.0f / 12));
    return bin == 30 ? 0 : bin;
  };
  for (int i = m0 + tid; i < m1; i += nt) {
    const int row = a.mp_match[i];
    if (row >= 0 && a.check_orientation) atomicAdd(&hist[bin_of(i, row)], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int b = 0; b < 30; ++b) {
      const int s = hist[b];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = b; }
      else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = b; }
      else if (s > max3) { max3 = s; i3 = b; }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) i3 = -1;
    keep[0] = i1; keep[1] = i2; keep[2] = i3;
  }
  __syncthreads();
  int mine = 0;
  for (int i = m0 + tid; i < m1; i += nt) {
    const int row = a.mp_match[i];
    if (row < 0) continue;
    if (a.check_orientation) {
      const int b = bin_of(i, row);
      if (b != keep[0] && b != keep[1] && b != keep[2]) { a.kp_mp[row] = -1; a.mp_match[i] = -1; continue; }
    }
    ++mine;
  }
  atomicAdd(&s_n, mine);
  __syncthreads();
  if (tid == 0 && a.n_matches) a.n_matches[f] = s_n;
}

// ---- ORBMatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBMatcher.cpp:676-794), the matcher of
// Tracking::MonocularInitialization.  The window query (level 0 only) is the shared cms_area path; k_init_dist evaluates the Hamming
// distance of every candidate pair in parallel; k_init_greedy then walks the level-0 key points of F1 in order like the reference does
// -- a key point of F2 may be taken over by a later, strictly better match, so the loop is inherently sequential in i1 -- with the 64
// lanes of one wavefront sharing each scan (two smallest keys dist << 16 | position = the sequential scan's best / second best).
// This runs for the first frames of a sequence only; it is latency, not throughput.
struct CmsInitArgs {
  int nq; const int* q_i1;                    // level-0 key points of F1, ascending
  const int* cand_off; const int* cand_idx;   // CSR from the window query; indices are batch rows of frame b2
  const uint4* desc1; const uint4* t_desc; const CmsKeyPoint* kp2; int row0, kp_cap;
  const float* angle1; uint16_t* pair_dist;
  int n1; int* matches12; float* prev_matched; int* n_matches; int8_t* bin_of;
  float nnratio; int check_orientation;
  const int* total; int cap;
};
extern "C" __global__ void __launch_bounds__(64) k_init_dist(CmsInitArgs a) {
  if (*a.total > a.cap) return;
  const int q = blockIdx.x, i1 = a.q_i1[q];
  const uint4 q0 = a.desc1[2 * (size_t)i1], q1 = a.desc1[2 * (size_t)i1 + 1];
  for (int c = a.cand_off[q] + threadIdx.x; c < a.cand_off[q + 1]; c += 64) {
    const size_t row = (size_t)a.cand_idx[c];
    a.pair_dist[c] = (uint16_t)track_hamming256(q0, q1, a.t_desc[2 * row], a.t_desc[2 * row + 1]);
  }
}
extern "C" __global__ void __launch_bounds__(64) k_init_greedy(CmsInitArgs a) {
  extern __shared__ int init_lds[];            // vMatchedDistance [kp_cap] | vnMatches21 [kp_cap]
  __shared__ int hist[32];
  int* md = init_lds;
  int* m21 = init_lds + a.kp_cap;
  const int lane = threadIdx.x;
  for (int k = lane; k < a.kp_cap; k += 64) { md[k] = 0x7FFFFFFF; m21[k] = -1; }
  if (lane < 32) hist[lane] = 0;
  for (int i = lane; i < a.n1; i += 64) { a.matches12[i] = -1; a.bin_of[i] = -1; }
  __syncthreads();
  if (*a.total > a.cap) return;
  int nmatches = 0;
  for (int q = 0; q < a.nq; ++q) {
    const int i1 = a.q_i1[q], c0 = a.cand_off[q], c1 = a.cand_off[q + 1];
    if (c0 == c1) continue;
    unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;                 // the two smallest keys this lane has seen
    for (int c = c0 + lane; c < c1; c += 64) {
      const int i2 = a.cand_idx[c] - a.row0;
      const unsigned dist = a.pair_dist[c];
      if (md[i2] <= (int)dist) continue;                         // prevent several features of F1 from mapping to one of F2 (:720-722)
      const unsigned key = (dist << 16) | (unsigned)(c - c0);
      if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
    }
    for (int o = 32; o > 0; o >>= 1) {                           // merge the lanes' pairs of smallest keys
      const unsigned o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
      const unsigned lo = min(k1, o1), hi = max(k1, o1);
      k2 = min(hi, min(k2, o2)); k1 = lo;
    }
    if (k1 == 0xFFFFFFFFu) continue;
    const int bestDist = (int)(k1 >> 16);
    const int bestDist2 = k2 == 0xFFFFFFFFu ? 0x7FFFFFFF : (int)(k2 >> 16);
    if (bestDist <= 50 && (float)bestDist < __fmul_rn((float)bestDist2, a.nnratio)) {      // TH_LOW, mfNNratio (:733-739)
      const int row = a.cand_idx[c0 + (int)(k1 & 0xFFFFu)], i2 = row - a.row0;
      const int old = m21[i2];
      __syncthreads();                                           // every lane has read the old state before lane 0 rewrites it
      if (lane == 0) {
        if (old >= 0) a.matches12[old] = -1;
        a.matches12[i1] = i2; m21[i2] = i1; md[i2] = bestDist;
        if (a.check_orientation) {
          float rot = __fsub_rn(a.angle1[i1], a.kp2[row].angle);
          if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
          int bin = (int)roundf(__fmul_rn(rot, 1.0f / 12));
          if (bin == 30) bin = 0;
          ++hist[bin];                                           // the entry stays even if the match is taken over later (:750-759)
          a.bin_of[i1] = (int8_t)bin;
        }
      }
      nmatches += old >= 0 ? 0 : 1;
      __syncthreads();
    }
  }
  __syncthreads();
  __threadfence_block();
  if (a.check_orientation) {
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int b = 0; b < 30; ++b) {
      const int s = hist[b];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = b; }
      else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = b; }
      else if (s > max3) { max3 = s; i3 = b; }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) i3 = -1;
    int removed = 0;
    for (int i = lane; i < a.n1; i += 64) {
      const int b = a.bin_of[i];
      if (b < 0 || b == i1 || b == i2 || b == i3) continue;
      if (a.matches12[i] >= 0) { a.matches12[i] = -1; ++removed; }
    }
    for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
    nmatches -= removed;
  }
  for (int i = lane; i < a.n1; i += 64) {
    const int m = a.matches12[i];
    if (m >= 0) { const CmsKeyPoint kp = a.kp2[a.row0 + m]; a.prev_matched[2 * i] = kp.x; a.prev_matched[2 * i + 1] = kp.y; }
  }
  if (lane == 0) *a.n_matches = nmatches;
}
