// oracle/orc_tri.cpp -- CPU ORACLE (test infrastructure only, see orc_api.h) for the mapping thread's steps either side of the local
// bundle adjustment:
//   LocalMapping::CreateNewMapPoints (bearing-vector version)      src/LocalMapping.cpp:209-386, ComputeE12 :469-482
//   ORBMatcher::SearchForTriangulation / CheckDistEpipolarLine      src/ORBMatcher.cpp:971-1125, :388-407, ComputeThreeMaxima :905-946
//   CamModelGeneral::GetVectorSigma(key, normalRig, sigma)          src/CamModelGeneral.cpp:307-333, GetPosInFace include/CamModelGeneral.h:205-209
//   ORBMatcher::Fuse(KeyFrame*, vpMapPoints, th)                    src/ORBMatcher.cpp:1127-1226 (search half; the map surgery stays with the caller)
//
// DBoW2 is needed only to PRODUCE a key frame's FeatureVector (node id -> feature indices); the functions here consume it as data.
//
// cv::Mat / cv::Matx arithmetic (OpenCV is not vendored; "parity unpinned") is taken as:
//   A*B, A*B+C without transposed operands, 3x3 by 3x3 or 3x1: cv::gemm's small-matrix path -- float products summed left to right in
//        float, result (float)((double)t * alpha + (double)c * beta)
//   products with a transposed operand (R1w*R2w.t()) or alpha != 1 (-R1w*...): the generic path -- double accumulation, one rounding
//   Mat::dot: double accumulation;  Matx/Vec::dot: float accumulation;  cv::norm: sqrt of the double sum of double squares
//   a*M1 + b*M2 row expressions: cv::addWeighted in float, (m1*a + m2*b), one temporary per parenthesis
//   cv::SVD::compute on a 4x4 CV_32F: one-sided Jacobi on the columns (JacobiSVDImpl_<float>): double dot products and norms, float
//        rotations, eps = 2*FLT_EPSILON, at most 30 sweeps, singular values sorted descending with the matching row swaps of Vt
#include "orc_api.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
inline float gemm3_small(const float* a /*row, stride 1*/, const float* b /*column, stride bs*/, int bs) {
  float t = a[0] * b[0];
  t = t + a[1] * b[bs];
  t = t + a[2] * b[2 * bs];
  return t;
}
inline void mat3_vec_small(const float* A, const float* x, const float* c, float* out) {   // A*x (+ c)
  for (int r = 0; r < 3; ++r) {
    const float t = gemm3_small(A + 3 * r, x, 1);
    out[r] = c ? (float)((double)t * 1.0 + (double)c[r] * 1.0) : (float)((double)t * 1.0);
  }
}
inline double ddot3(const float* a, const float* b) { double s = 0; for (int k = 0; k < 3; ++k) s += (double)a[k] * (double)b[k]; return s; }
inline double dnorm3(const float* a) { return std::sqrt(ddot3(a, a)); }

// CamModelGeneral::GetPosInFace<float>
inline void pos_in_face(int F, float uc, float vc, float* u, float* v) {
  const int i = (int)std::floor(uc / F), j = (int)std::floor(vc / F);
  *u = uc - i * F; *v = vc - j * F;
}
inline void rig_to_face(const float* g, int face, float* l) {   // CamModelGeneral.h:417-443
  switch (face) {
    case ORC_FACE_FRONT: l[0] = g[0]; l[1] = g[1]; l[2] = g[2]; break;
    case ORC_FACE_LEFT: l[0] = g[2]; l[1] = g[1]; l[2] = -g[0]; break;
    case ORC_FACE_RIGHT: l[0] = -g[2]; l[1] = g[1]; l[2] = g[0]; break;
    case ORC_FACE_LOWER: l[0] = g[0]; l[1] = -g[2]; l[2] = g[1]; break;
    case ORC_FACE_UPPER: l[0] = g[0]; l[1] = g[2]; l[2] = -g[1]; break;
    default: l[0] = 0; l[1] = 0; l[2] = 0;
  }
}
// CamModelGeneral::GetVectorSigma(key, normalRig, sigmaInPixel = 1) (CamModelGeneral.cpp:307-333)
float vector_sigma(const orc_camera* cam, float kx, float ky, const float* normalRig) {
  const int F = cam->face;
  const double fx = F / 2.0, cx = F / 2.0, cy = F / 2.0;
  const float sigmaInPixel = 1.0f;
  float nc[3];
  rig_to_face(normalRig, orc_face_in_cubemap(cam, kx, ky), nc);
  const float epi[3] = {nc[1], -nc[0], 0.0f}, ver[3] = {nc[0], nc[1], 0.0f};
  float u, v;
  pos_in_face(F, kx, ky, &u, &v);
  const float OP[3] = {(float)(u - cx), (float)(v - cy), 0.0f};
  auto fdot = [](const float* a, const float* b) { float s = 0; for (int k = 0; k < 3; ++k) s += a[k] * b[k]; return s; };
  float OO1 = (float)(fdot(OP, epi) / dnorm3(epi)); if (OO1 < 0) OO1 = -OO1;
  const float CO1 = (float)std::sqrt(OO1 * OO1 + fx * fx);
  float PO1 = (float)(fdot(OP, ver) / dnorm3(ver)); if (PO1 < 0) PO1 = -PO1;
  const float tan1 = PO1 / CO1;
  const float tan2 = (PO1 + sigmaInPixel) / CO1;
  const float tan3 = (tan2 - tan1) / (1 + tan1 * tan2);
  return 1.0f / std::sqrt(1.0f / (tan3 * tan3) + 1);
}
// ORBMatcher::CheckDistEpipolarLine (ORBMatcher.cpp:388-407)
bool check_epipolar(const orc_camera* cam, const float* ray1, const float* ray2, float k2x, float k2y, const float* E, float sigma2_oct) {
  const float a = ray1[0] * E[0] + ray1[1] * E[3] + ray1[2] * E[6];
  const float b = ray1[0] * E[1] + ray1[1] * E[4] + ray1[2] * E[7];
  const float c = ray1[0] * E[2] + ray1[1] * E[5] + ray1[2] * E[8];
  const float num = a * ray2[0] + b * ray2[1] + c * ray2[2];
  const float den = a * a + b * b + c * c;
  if (den == 0) return false;
  const float n[3] = {a, b, c};
  const float sigma = vector_sigma(cam, k2x, k2y, n);
  const float sigmaSquare = sigma * sigma;
  const float dsqr = num * num / (den * sigmaSquare * sigma2_oct);
  return dsqr < 3.84;
}
// cv::SVD::compute(A, w, u, vt, MODIFY_A | FULL_UV) for a 4x4 float A; returns vt.row(3)
void svd4_last_row(const float A[16], float out[4]) {
  float At[16], Vt[16];
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) At[4 * i + k] = A[4 * k + i];       // rows of At = columns of A
  double W[4];
  const int m = 4, n = 4;
  const float eps = FLT_EPSILON * 2;
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) { const float t = At[4 * i + k]; sd += (double)t * t; }
    W[i] = sd;
    for (int k = 0; k < n; ++k) Vt[4 * i + k] = 0;
    Vt[4 * i + i] = 1;
  }
  for (int iter = 0; iter < 30; ++iter) {
    bool changed = false;
    for (int i = 0; i < n - 1; ++i)
      for (int j = i + 1; j < n; ++j) {
        float *Ai = At + 4 * i, *Aj = At + 4 * j;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; ++k) p += (double)Ai[k] * Aj[k];
        if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = hypot((double)p, beta);
        float c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = (float)std::sqrt(delta / gamma);
          c = (float)(p / (gamma * s * 2));
        } else {
          c = (float)std::sqrt((gamma + beta) / (gamma * 2));
          s = (float)(p / (gamma * c * 2));
        }
        a = b = 0;
        for (int k = 0; k < m; ++k) {
          const float t0 = c * Ai[k] + s * Aj[k];
          const float t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += (double)t0 * t0; b += (double)t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        float *Vi = Vt + 4 * i, *Vj = Vt + 4 * j;
        for (int k = 0; k < n; ++k) {
          const float t0 = c * Vi[k] + s * Vj[k];
          const float t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0; Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) { const float t = At[4 * i + k]; sd += (double)t * t; }
    W[i] = std::sqrt(sd);
  }
  for (int i = 0; i < n - 1; ++i) {
    int j = i;
    for (int k = i + 1; k < n; ++k) if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      for (int k = 0; k < m; ++k) std::swap(At[4 * i + k], At[4 * j + k]);
      for (int k = 0; k < n; ++k) std::swap(Vt[4 * i + k], Vt[4 * j + k]);
    }
  }
  for (int k = 0; k < 4; ++k) out[k] = Vt[12 + k];
}
// one row of the triangulation system: r_a*(T.row(ia)+T.row(ib)) - (r_b + r_c)*T.row(ic)   (LocalMapping.cpp:296-299)
inline void tri_row(const float* T /*3x4*/, int ia, int ib, int ic, float ra, float rb, float rc, float* out) {
  const float g = -(rb + rc);
  for (int k = 0; k < 4; ++k) {
    const float tmp = T[4 * ia + k] * ra + T[4 * ib + k] * ra;     // ra*(row + row): addWeighted(row, ra, row, ra)
    out[k] = tmp * 1.0f + T[4 * ic + k] * g;                        // tmp - (rb+rc)*row: addWeighted(tmp, 1, row, -(rb+rc))
  }
}
}  // namespace

// LocalMapping::ComputeE12 (LocalMapping.cpp:469-482)
extern "C" void orc_compute_e12(const float* R1w, const float* t1w, const float* R2w, const float* t2w, float* E12) {
  float R12[9], M[9], t12[3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += (double)R1w[3 * r + k] * (double)R2w[3 * c + k];
      R12[3 * r + c] = (float)(s * 1.0);
      M[3 * r + c] = (float)(s * -1.0);                             // (-R1w)*R2w.t(): alpha = -1
    }
  mat3_vec_small(M, t2w, t1w, t12);
  const float tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) E12[3 * r + c] = (float)((double)gemm3_small(tx + 3 * r, R12 + c, 3) * 1.0);
}

// ORBMatcher::SearchForTriangulation: matches12[i1] = index in kf2 or -1; returns nmatches
extern "C" int orc_search_for_triangulation(const orc_camera* cam, const orc_keyframe* kf1, const orc_keyframe* kf2, const float* E12,
                                            const float* scale_factors, const float* level_sigma2, int check_orientation, int* matches12) {
  const int TH_LOW = 50, HISTO_LENGTH = 12;
  float C2[3], ex, ey;
  mat3_vec_small(kf2->Rcw, kf1->Ow, kf2->tcw, C2);
  orc_rays_to_cubemap(cam, C2[0], C2[1], C2[2], &ex, &ey);           // the face is not looked at (ORBMatcher.cpp:982)
  int nmatches = 0;
  for (int i = 0; i < kf1->n; ++i) matches12[i] = -1;
  const int nBins = (int)std::ceil(360.0f / HISTO_LENGTH);
  std::vector<std::vector<int>> rotHist(nBins);
  const float factor = 1.0f / HISTO_LENGTH;
  int f1 = 0, f2 = 0;
  while (f1 < kf1->nnodes && f2 < kf2->nnodes) {
    if (kf1->node_id[f1] == kf2->node_id[f2]) {
      for (int a = kf1->node_off[f1]; a < kf1->node_off[f1 + 1]; ++a) {
        const int idx1 = kf1->node_feat[a];
        if (kf1->mp[idx1] >= 0) continue;
        const orc_keypoint& kp1 = kf1->kps[idx1];
        const float* ray1 = kf1->rays + 3 * (size_t)idx1;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int b = kf2->node_off[f2]; b < kf2->node_off[f2 + 1]; ++b) {
          const int idx2 = kf2->node_feat[b];
          if (kf2->mp[idx2] >= 0) continue;                           // vbMatched2 is never set in the reference
          const int dist = orc_descriptor_distance(kf1->desc + 32 * (size_t)idx1, kf2->desc + 32 * (size_t)idx2);
          if (dist > TH_LOW || dist > bestDist) continue;
          const orc_keypoint& kp2 = kf2->kps[idx2];
          const float distex = ex - kp2.x, distey = ey - kp2.y;
          if (distex * distex + distey * distey < 100 * scale_factors[kp2.octave]) continue;
          if (check_epipolar(cam, ray1, kf2->rays + 3 * (size_t)idx2, kp2.x, kp2.y, E12, level_sigma2[kp2.octave])) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
          matches12[idx1] = bestIdx2; ++nmatches;
          if (check_orientation) {
            float rot = kp1.angle - kf2->kps[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == nBins) bin = 0;
            rotHist[bin].push_back(idx1);
          }
        }
      }
      ++f1; ++f2;
    } else if (kf1->node_id[f1] < kf2->node_id[f2]) {
      f1 = (int)(std::lower_bound(kf1->node_id, kf1->node_id + kf1->nnodes, kf2->node_id[f2]) - kf1->node_id);
    } else {
      f2 = (int)(std::lower_bound(kf2->node_id, kf2->node_id + kf2->nnodes, kf1->node_id[f1]) - kf2->node_id);
    }
  }
  if (check_orientation) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < nBins; ++i) {
      const int s = (int)rotHist[i].size();
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; } else if (max3 < 0.1f * (float)max1) ind3 = -1;
    for (int i = 0; i < nBins; ++i) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int j : rotHist[i]) { matches12[j] = -1; --nmatches; }
    }
  }
  return nmatches;
}

// the body of CreateNewMapPoints' inner loop for one matched pair (LocalMapping.cpp:266-357); returns 1 and x3D when the point survives
extern "C" int orc_triangulate_match(const orc_camera* cam, const orc_keyframe* kf1, const orc_keyframe* kf2, int idx1, int idx2,
                                     const float* scale_factors, const float* level_sigma2, float ratio_factor, float* x3d_out) {
  const orc_keypoint &kp1 = kf1->kps[idx1], &kp2 = kf2->kps[idx2];
  const float *r1 = kf1->rays + 3 * (size_t)idx1, *r2 = kf2->rays + 3 * (size_t)idx2;
  float Rwc1[9], Rwc2[9], T1[12], T2[12];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      Rwc1[3 * r + c] = kf1->Rcw[3 * c + r]; Rwc2[3 * r + c] = kf2->Rcw[3 * c + r];
      T1[4 * r + c] = kf1->Rcw[3 * r + c]; T2[4 * r + c] = kf2->Rcw[3 * r + c];
    }
  for (int r = 0; r < 3; ++r) { T1[4 * r + 3] = kf1->tcw[r]; T2[4 * r + 3] = kf2->tcw[r]; }
  float ray1[3], ray2[3];
  mat3_vec_small(Rwc1, r1, nullptr, ray1);
  mat3_vec_small(Rwc2, r2, nullptr, ray2);
  const float cosParallaxRays = (float)(ddot3(ray1, ray2) / (dnorm3(ray1) * dnorm3(ray2)));
  const float cosParallaxStereo = cosParallaxRays + 1;
  if (!(cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && cosParallaxRays < 0.9998)) return 0;
  float A[16], v4[4];
  tri_row(T1, 1, 2, 0, r1[0], r1[1], r1[2], A);
  tri_row(T1, 0, 2, 1, r1[1], r1[0], r1[2], A + 4);
  tri_row(T2, 1, 2, 0, r2[0], r2[1], r2[2], A + 8);
  tri_row(T2, 0, 2, 1, r2[1], r2[0], r2[2], A + 12);
  svd4_last_row(A, v4);
  if (v4[3] == 0) return 0;
  float x3D[3];
  const float inv_w = (float)(1.0 / (double)v4[3]);                   // Mat / scalar = convertTo with scale 1/s, carried as float for CV_32F
  for (int k = 0; k < 3; ++k) x3D[k] = v4[k] * inv_w;
  const float cosFov = orc_cos_fov_th(cam);
  float xc1[3], xc2[3];
  mat3_vec_small(kf1->Rcw, x3D, kf1->tcw, xc1);
  const float d1 = (float)(xc1[2] / dnorm3(xc1));
  if (d1 <= cosFov) return 0;
  mat3_vec_small(kf2->Rcw, x3D, kf2->tcw, xc2);
  const float d2 = (float)(xc2[2] / dnorm3(xc2));
  if (d2 <= cosFov) return 0;
  {
    const float x = (float)(ddot3(kf1->Rcw, x3D) + kf1->tcw[0]), y = (float)(ddot3(kf1->Rcw + 3, x3D) + kf1->tcw[1]),
                z = (float)(ddot3(kf1->Rcw + 6, x3D) + kf1->tcw[2]);
    float u, v;
    orc_rays_to_cubemap(cam, x, y, z, &u, &v);
    const float eX = u - kp1.x, eY = v - kp1.y;
    if ((eX * eX + eY * eY) > 5.991 * level_sigma2[kp1.octave]) return 0;
  }
  {
    const float x = (float)(ddot3(kf2->Rcw, x3D) + kf2->tcw[0]), y = (float)(ddot3(kf2->Rcw + 3, x3D) + kf2->tcw[1]),
                z = (float)(ddot3(kf2->Rcw + 6, x3D) + kf2->tcw[2]);
    float u, v;
    orc_rays_to_cubemap(cam, x, y, z, &u, &v);
    const float eX = u - kp2.x, eY = v - kp2.y;
    if ((eX * eX + eY * eY) > 5.991 * level_sigma2[kp2.octave]) return 0;
  }
  const float n1[3] = {x3D[0] - kf1->Ow[0], x3D[1] - kf1->Ow[1], x3D[2] - kf1->Ow[2]};
  const float n2[3] = {x3D[0] - kf2->Ow[0], x3D[1] - kf2->Ow[1], x3D[2] - kf2->Ow[2]};
  const float dist1 = (float)dnorm3(n1), dist2 = (float)dnorm3(n2);
  if (dist1 == 0 || dist2 == 0) return 0;
  const float ratioDist = dist2 / dist1;
  const float ratioOctave = scale_factors[kp1.octave] / scale_factors[kp2.octave];
  if (ratioDist * ratio_factor < ratioOctave || ratioDist > ratioOctave * ratio_factor) return 0;
  x3d_out[0] = x3D[0]; x3d_out[1] = x3D[1]; x3d_out[2] = x3D[2];
  return 1;
}

// LocalMapping::CreateNewMapPoints over the given neighbours (covisibility order).  Returns nnew; out_* get (neighbour, idx1, idx2,
// x3D) in creation order.  cur_mp is updated like KeyFrame::AddMapPoint does (so later neighbours skip the new points).
extern "C" int orc_create_new_map_points(const orc_camera* cam, const orc_keyframe* cur, int nneigh, const orc_keyframe* neigh,
                                         const float* scale_factors, const float* level_sigma2, int* cur_mp_inout, int* out_neigh,
                                         int* out_idx1, int* out_idx2, float* out_x3d, int cap) {
  orc_keyframe k1 = *cur;
  k1.mp = cur_mp_inout;
  const float ratioFactor = 1.5f * scale_factors[1];                 // 1.5f * mfScaleFactor
  std::vector<int> m12((size_t)std::max(cur->n, 1));
  int nnew = 0;
  for (int i = 0; i < nneigh; ++i) {
    const orc_keyframe* k2 = neigh + i;
    const float vB[3] = {k2->Ow[0] - k1.Ow[0], k2->Ow[1] - k1.Ow[1], k2->Ow[2] - k1.Ow[2]};
    const float baseline = (float)dnorm3(vB);
    const float ratioBaselineDepth = baseline / k2->median_depth;
    if (ratioBaselineDepth < 0.01) continue;
    float E12[9];
    orc_compute_e12(k1.Rcw, k1.tcw, k2->Rcw, k2->tcw, E12);
    orc_search_for_triangulation(cam, &k1, k2, E12, scale_factors, level_sigma2, 0, m12.data());   // ORBMatcher matcher(0.6, false)
    for (int idx1 = 0; idx1 < k1.n; ++idx1) {
      const int idx2 = m12[idx1];
      if (idx2 < 0) continue;
      float x[3];
      if (!orc_triangulate_match(cam, &k1, k2, idx1, idx2, scale_factors, level_sigma2, ratioFactor, x)) continue;
      if (nnew < cap) { out_neigh[nnew] = i; out_idx1[nnew] = idx1; out_idx2[nnew] = idx2; std::memcpy(out_x3d + 3 * (size_t)nnew, x, 12); }
      cur_mp_inout[idx1] = 1 << 30;                                  // mpCurrentKeyFrame->AddMapPoint(pMP, idx1)
      ++nnew;
    }
  }
  return nnew;
}

// Search half of ORBMatcher::Fuse(pKF, vpMapPoints, th): per map point the key point it would be fused with (or -1) and the distance.
// skip[i] != 0 stands for !pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF).
extern "C" void orc_fuse_search(const orc_camera* cam, const orc_keyframe* kf, int nmp, const uint8_t* skip, const float* P, const float* normal,
                                const float* min_dist, const float* max_dist, const uint8_t* mp_desc, float th, const float* scale_factors,
                                const float* inv_level_sigma2, int nlevels, int* best_idx, int* best_dist) {
  const int TH_LOW = 50;
  const float mnMax = (float)(3 * cam->face);
  const float logScale = std::log(scale_factors[1]);
  std::vector<float> kx((size_t)kf->n), ky((size_t)kf->n);
  std::vector<int> ko((size_t)kf->n);
  for (int k = 0; k < kf->n; ++k) { kx[k] = kf->kps[k].x; ky[k] = kf->kps[k].y; ko[k] = kf->kps[k].octave; }
  std::vector<float> qx, qy, qr;
  std::vector<int> lo, hi, qi, qlvl;
  for (int i = 0; i < nmp; ++i) {
    best_idx[i] = -1; best_dist[i] = 256;
    if (skip && skip[i]) continue;
    const float* p = P + 3 * (size_t)i;
    float pc[3], u, v;
    mat3_vec_small(kf->Rcw, p, kf->tcw, pc);
    orc_rays_to_cubemap(cam, pc[0], pc[1], pc[2], &u, &v);
    if (!(u >= 0.0f && u < mnMax && v >= 0.0f && v < mnMax)) continue;            // KeyFrame::IsInImage
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    const float PO[3] = {p[0] - kf->Ow[0], p[1] - kf->Ow[1], p[2] - kf->Ow[2]};
    const float dist3D = (float)dnorm3(PO);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    if (ddot3(PO, normal + 3 * (size_t)i) < 0.5 * dist3D) continue;
    const float ratio = max_dist[i] / dist3D;
    int nScale = (int)std::ceil(std::log(ratio) / logScale);
    if (nScale < 0) nScale = 0; else if (nScale >= nlevels) nScale = nlevels - 1;
    qx.push_back(u); qy.push_back(v); qr.push_back(th * scale_factors[nScale]); lo.push_back(-1); hi.push_back(-1); qi.push_back(i); qlvl.push_back(nScale);
  }
  const int nq = (int)qi.size();
  std::vector<int> off((size_t)nq + 1, 0), idx((size_t)64 * nq + 1024);
  const int tot = orc_features_in_area(cam, kf->n, kx.data(), ky.data(), ko.data(), nq, qx.data(), qy.data(), qr.data(), lo.data(), hi.data(), off.data(),
                                       idx.data(), (int)idx.size());
  if (tot > (int)idx.size()) {
    idx.resize((size_t)tot);
    orc_features_in_area(cam, kf->n, kx.data(), ky.data(), ko.data(), nq, qx.data(), qy.data(), qr.data(), lo.data(), hi.data(), off.data(), idx.data(), (int)idx.size());
  }
  for (int q = 0; q < nq; ++q) {
    const int i = qi[q], nPredictedLevel = qlvl[q];
    int bestDist = 256, bestIdx = -1;
    for (int c = off[q]; c < off[q + 1]; ++c) {
      const int k = idx[c];
      const int kpLevel = ko[k];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const float ex = qx[q] - kx[k], ey = qy[q] - ky[k];
      const float e2 = ex * ex + ey * ey;
      if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
      const int dist = orc_descriptor_distance(mp_desc + 32 * (size_t)i, kf->desc + 32 * (size_t)k);
      if (dist < bestDist) { bestDist = dist; bestIdx = k; }
    }
    if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; best_dist[i] = bestDist; }
  }
}

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cpp:243-308) for a batch of map points: obs_off[npts+1] into the concatenated
// descriptors of each point's (non-bad) observations, in the std::map order the caller walked them; best_idx[p] = index inside the
// point's list of the descriptor with the least median distance to the rest (first one on ties), -1 for a point without observations.
extern "C" void orc_distinctive_descriptors(int npts, const int* obs_off, const uint8_t* desc, int* best_idx) {
  for (int p = 0; p < npts; ++p) {
    const int N = obs_off[p + 1] - obs_off[p];
    best_idx[p] = -1;
    if (N <= 0) continue;
    const uint8_t* D = desc + 32 * (size_t)obs_off[p];
    std::vector<float> dist((size_t)N * N);
    for (int i = 0; i < N; ++i) {
      dist[(size_t)i * N + i] = 0;
      for (int j = i + 1; j < N; ++j) {
        const int d = orc_descriptor_distance(D + 32 * (size_t)i, D + 32 * (size_t)j);
        dist[(size_t)i * N + j] = (float)d; dist[(size_t)j * N + i] = (float)d;
      }
    }
    int BestMedian = 0x7fffffff, BestIdx = 0;
    for (int i = 0; i < N; ++i) {
      std::vector<int> v(dist.begin() + (size_t)i * N, dist.begin() + (size_t)(i + 1) * N);
      std::sort(v.begin(), v.end());
      const int median = v[(size_t)(0.5 * (N - 1))];
      if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    best_idx[p] = BestIdx;
  }
}

// MapPoint::UpdateNormalAndDepth (src/MapPoint.cpp:332-373): normal = mean of the unit vectors from the observing key frames' centres
// (cv::Mat float arithmetic: addWeighted(normal, 1, normali, 1/norm) per observation, then scale by (float)(1/n)), dist to the
// reference key frame, mfMaxDistance = dist * scaleFactor[level of the reference observation], mfMinDistance = max / scaleFactor[nLevels-1].
extern "C" void orc_update_normal_and_depth(int npts, const int* obs_off, const float* pos, const float* obs_Ow, const float* ref_Ow,
                                            const int* ref_level, const float* scale_factors, int nlevels, float* normal, float* min_dist,
                                            float* max_dist) {
  for (int p = 0; p < npts; ++p) {
    const int N = obs_off[p + 1] - obs_off[p];
    if (N <= 0) continue;                                             // observations.empty(): nothing is touched
    const float* P = pos + 3 * (size_t)p;
    float nrm[3] = {0, 0, 0};
    for (int o = obs_off[p]; o < obs_off[p + 1]; ++o) {
      const float ni[3] = {P[0] - obs_Ow[3 * (size_t)o], P[1] - obs_Ow[3 * (size_t)o + 1], P[2] - obs_Ow[3 * (size_t)o + 2]};
      const float beta = (float)(1.0 / dnorm3(ni));
      for (int k = 0; k < 3; ++k) nrm[k] = nrm[k] * 1.0f + ni[k] * beta;
    }
    const float PC[3] = {P[0] - ref_Ow[3 * (size_t)p], P[1] - ref_Ow[3 * (size_t)p + 1], P[2] - ref_Ow[3 * (size_t)p + 2]};
    const float dist = (float)dnorm3(PC);
    const float mx = dist * scale_factors[ref_level[p]];
    max_dist[p] = mx;
    min_dist[p] = mx / scale_factors[nlevels - 1];
    const float inv_n = (float)(1.0 / (double)N);
    for (int k = 0; k < 3; ++k) normal[3 * (size_t)p + k] = nrm[k] * inv_n;
  }
}
