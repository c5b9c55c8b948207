// oracle/orc_track.cpp -- CPU ORACLE (test infrastructure only, see orc_api.h) for the "track local map" projection + search:
//   Frame::isInFrustum                      src/Frame.cpp:197-249
//   MapPoint::PredictScale(dist, Frame*)    src/MapPoint.cpp:404-419, Get{Min,Max}DistanceInvariance :375-385
//   ORBMatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)   src/ORBMatcher.cpp:50-128, RadiusByViewingCos :380-386
// as Tracking::SearchLocalPoints drives them (src/Tracking.cpp:794-846: isInFrustum(pMP, 0.5), ORBMatcher(0.8), th = 1 or 5).
//
// cv::Mat arithmetic the reference leans on (OpenCV is not vendored; "parity unpinned") is taken as:
//   mRcw*P+mtcw   3x3 * 3x1 + 3x1 in CV_32F = cv::gemm's small-matrix path: t = a0*b0 + a1*b1 + a2*b2 in float (left to right, no
//                 contraction), result (float)((double)t * 1.0 + (double)c * 1.0)
//   P-mOw         float subtraction
//   cv::norm      sqrt of the double sum of double squares, then narrowed by `const float dist = ...`
//   PO.dot(Pn)    double sum of double products; `/dist` is double / float, narrowed by `const float viewCos = ...`
//   log(ratio)    <cmath>'s float overload (logf), divided by the float mfLogScaleFactor = logf(mfScaleFactor) (Frame.cpp:113)
#include "orc_api.h"
#include <cmath>
#include <vector>

extern "C" int orc_is_in_frustum(const orc_camera* cam, const float* Rcw, const float* tcw, const float* Ow, int n, const float* P,
                                 const float* normal, const float* min_dist, const float* max_dist, float viewing_cos_limit,
                                 float scale_factor, int nlevels, uint8_t* in_view, float* proj_x, float* proj_y, int* level,
                                 float* view_cos) {
  const float mnMinX = 0.0f, mnMinY = 0.0f, mnMaxX = (float)(3 * cam->face), mnMaxY = (float)(3 * cam->face);   // Frame.cpp:144-147
  const float logScale = std::log(scale_factor);                                                              // Frame.cpp:113
  int nin = 0;
  for (int i = 0; i < n; ++i) {
    in_view[i] = 0; proj_x[i] = -1.0f; proj_y[i] = -1.0f; level[i] = -1; view_cos[i] = 0.0f;                       // mbTrackInView = false
    const float* p = P + 3 * i;
    float Pc[3];
    for (int r = 0; r < 3; ++r) {
      float t = Rcw[3 * r] * p[0];
      t = t + Rcw[3 * r + 1] * p[1];
      t = t + Rcw[3 * r + 2] * p[2];
      Pc[r] = (float)((double)t * 1.0 + (double)tcw[r] * 1.0);
    }
    float u, v;
    const int face = orc_rays_to_cubemap(cam, Pc[0], Pc[1], Pc[2], &u, &v);
    if (face == ORC_FACE_UNKNOWN) continue;
    if (u < mnMinX || u > mnMaxX) continue;
    if (v < mnMinY || v > mnMaxY) continue;
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    const float PO[3] = {p[0] - Ow[0], p[1] - Ow[1], p[2] - Ow[2]};
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (double)PO[k] * (double)PO[k];
    const float dist = (float)std::sqrt(s);
    if (dist < minDistance || dist > maxDistance) continue;
    double d = 0;
    for (int k = 0; k < 3; ++k) d += (double)PO[k] * (double)normal[3 * i + k];
    const float viewCos = (float)(d / dist);
    if (viewCos < viewing_cos_limit) continue;
    const float ratio = max_dist[i] / dist;                                                                     // MapPoint.cpp:409
    int nScale = (int)std::ceil(std::log(ratio) / logScale);                                                    // float log, float divide
    if (nScale < 0) nScale = 0; else if (nScale >= nlevels) nScale = nlevels - 1;
    in_view[i] = 1; proj_x[i] = u; proj_y[i] = v; level[i] = nScale; view_cos[i] = viewCos;
    ++nin;
  }
  return nin;
}

// SearchByProjection(F, vpMapPoints, th) over the map points isInFrustum marked; returns nmatches.  kp_mp (in/out, one int per
// key point of F): < 0 = key point free, otherwise the index of the map point it holds (entries the caller passes >= 0 stand for
// "mvpMapPoints[idx] with Observations() > 0"; new matches are written as the map point's index in the list).
extern "C" int orc_search_local_points(const orc_camera* cam, int nkp, const float* kx, const float* ky, const int* koct,
                                       const uint8_t* kdesc, const float* scale_factors, int nmp, const uint8_t* in_view,
                                       const float* proj_x, const float* proj_y, const int* level, const float* view_cos,
                                       const uint8_t* mp_desc, float th, float nnratio, int th_high, int* kp_mp, int* mp_match) {
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  // the windows do not depend on the matching state: all GetFeaturesInArea calls first (the frame grid is built once, as
  // Frame::AssignFeaturesToGrid does), then the reference's loop over the map points
  std::vector<float> qx, qy, qr;
  std::vector<int> lo, hi, qi;
  for (int i = 0; i < nmp; ++i) {
    mp_match[i] = -1;
    if (!in_view[i]) continue;
    const int nPredictedLevel = level[i];
    float r = view_cos[i] > 0.998 ? 2.5f : 4.0f;                     // RadiusByViewingCos: float vs the double literal 0.998
    if (bFactor) r *= th;
    qx.push_back(proj_x[i]); qy.push_back(proj_y[i]); qr.push_back(r * scale_factors[nPredictedLevel]);
    lo.push_back(nPredictedLevel - 1); hi.push_back(nPredictedLevel); qi.push_back(i);
  }
  const int nq = (int)qi.size();
  std::vector<int> off((size_t)nq + 1, 0), idx((size_t)64 * nq + 1024);
  int tot = orc_features_in_area(cam, nkp, kx, ky, koct, nq, qx.data(), qy.data(), qr.data(), lo.data(), hi.data(), off.data(), idx.data(), (int)idx.size());
  if (tot > (int)idx.size()) {
    idx.resize((size_t)tot);
    orc_features_in_area(cam, nkp, kx, ky, koct, nq, qx.data(), qy.data(), qr.data(), lo.data(), hi.data(), off.data(), idx.data(), (int)idx.size());
  }
  for (int q = 0; q < nq; ++q) {
    const int i = qi[q];
    const int* cand = idx.data() + off[q];
    const int nc = off[q + 1] - off[q];
    if (nc == 0) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int c = 0; c < nc; ++c) {
      const int k = cand[c];
      if (kp_mp[k] >= 0) continue;
      const int dist = orc_descriptor_distance(mp_desc + 32 * (size_t)i, kdesc + 32 * (size_t)k);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = koct[k]; bestIdx = k; }
      else if (dist < bestDist2) { bestLevel2 = koct[k]; bestDist2 = dist; }
    }
    if (bestDist <= th_high) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      kp_mp[bestIdx] = i;
      mp_match[i] = bestIdx;
      ++nmatches;
    }
  }
  return nmatches;
}

// ORBMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBMatcher.cpp:130-251), the matcher of
// Tracking::TrackWithMotionModel.  Rcw / tcw: CurrentFrame.mTcw (float).  Per key point i of the last frame: valid[i] = it holds a map
// point and is not an outlier, Xw = that point's world position, mp_desc = its descriptor (MapPoint::GetDescriptor), oct / angle = the
// last frame's key point.  kp_mp (in/out, per key point of the current frame): >= 0 = taken; new matches store i.  Returns nmatches
// (after the rotation-consistency filter when check_orientation).
extern "C" int orc_search_by_projection_frames(const orc_camera* cam, const float* Rcw, const float* tcw, int nkp, const float* kx, const float* ky,
                                               const int* koct, const float* kangle, const uint8_t* kdesc, const float* scale_factors, int nlast,
                                               const uint8_t* valid, const float* Xw, const int* loct, const float* langle, const uint8_t* mp_desc,
                                               float th, int check_orientation, int th_high, int* kp_mp, int* match) {
  const int HISTO_LENGTH = 12;
  const int nBins = (int)std::ceil(360.0f / HISTO_LENGTH);
  const float factor = 1.0f / HISTO_LENGTH;
  const float cosFov = orc_cos_fov_th(cam);
  std::vector<std::vector<int>> rotHist(nBins);
  std::vector<float> qx, qy, qr;
  std::vector<int> lo, hi, qi;
  for (int i = 0; i < nlast; ++i) {
    match[i] = -1;
    if (!valid[i]) continue;
    const float* p = Xw + 3 * (size_t)i;
    float xc[3];
    for (int r = 0; r < 3; ++r) {
      float t = Rcw[3 * r] * p[0];
      t = t + Rcw[3 * r + 1] * p[1];
      t = t + Rcw[3 * r + 2] * p[2];
      xc[r] = (float)((double)t * 1.0 + (double)tcw[r] * 1.0);
    }
    if (xc[2] < cosFov) continue;
    float u, v;
    if (orc_rays_to_cubemap(cam, xc[0], xc[1], xc[2], &u, &v) == ORC_FACE_UNKNOWN) continue;
    const int o = loct[i];
    qx.push_back(u); qy.push_back(v); qr.push_back(th * scale_factors[o]); lo.push_back(o - 1); hi.push_back(o + 1); qi.push_back(i);
  }
  const int nq = (int)qi.size();
  std::vector<int> off((size_t)nq + 1, 0), idx((size_t)64 * nq + 1024);
  const int tot = orc_features_in_area(cam, nkp, kx, ky, koct, nq, qx.data(), qy.data(), qr.data(), lo.data(), hi.data(), off.data(), idx.data(), (int)idx.size());
  if (tot > (int)idx.size()) {
    idx.resize((size_t)tot);
    orc_features_in_area(cam, nkp, kx, ky, koct, nq, qx.data(), qy.data(), qr.data(), lo.data(), hi.data(), off.data(), idx.data(), (int)idx.size());
  }
  int nmatches = 0;
  for (int q = 0; q < nq; ++q) {
    const int i = qi[q];
    int bestDist = 256, bestIdx2 = -1;
    for (int c = off[q]; c < off[q + 1]; ++c) {
      const int i2 = idx[c];
      if (kp_mp[i2] >= 0) continue;
      const int dist = orc_descriptor_distance(mp_desc + 32 * (size_t)i, kdesc + 32 * (size_t)i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= th_high) {
      kp_mp[bestIdx2] = i; match[i] = bestIdx2; ++nmatches;
      if (check_orientation) {
        float rot = langle[i] - kangle[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == nBins) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
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
      for (int i2 : rotHist[i]) { match[kp_mp[i2]] = -1; kp_mp[i2] = -1; --nmatches; }   // CurrentFrame.mvpMapPoints[...] = NULL
    }
  }
  return nmatches;
}

// ORBMatcher::SearchForInitialization(Frame& F1, Frame& F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBMatcher.cpp:676-794), the
// matcher of Tracking::MonocularInitialization (Tracking.cpp:428-429: ORBMatcher(0.9, true), windowSize 100).  Level-0 key points of F1
// only (:693-696); candidates = F2.GetFeaturesInArea(prev_matched[i1], windowSize, 0, 0); a candidate already matched at a distance
// <= this one is skipped (:720-722); accepted if bestDist <= TH_LOW (50) and bestDist < (float)bestDist2 * nnratio (:736-739); a key
// point of F2 that was matched before is taken over (:741-745); the rotation histogram keeps the i1 of every accepted match, also of
// those taken over later (:750-759), and the filter clears what is still matched (:763-783); finally vbPrevMatched follows the
// matches (:786-789).  matches12: n1 ints (index in F2 or -1); prev_matched: n1 x 2 floats, in/out.  Returns nmatches.
extern "C" int orc_search_for_initialization(const orc_camera* cam, int n1, const float* k1x, const float* k1y, const int* k1oct, const float* k1angle,
                                             const uint8_t* desc1, int n2, const float* k2x, const float* k2y, const int* k2oct, const float* k2angle,
                                             const uint8_t* desc2, float* prev_matched, int window_size, float nnratio, int check_orientation,
                                             int* matches12) {
  const int HISTO_LENGTH = 12, TH_LOW = 50;
  const int nBins = (int)std::ceil(360.0f / HISTO_LENGTH);
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<std::vector<int>> rotHist(nBins);
  int nmatches = 0;
  for (int i = 0; i < n1; ++i) matches12[i] = -1;
  std::vector<int> vMatchedDistance((size_t)n2, 0x7FFFFFFF), vnMatches21((size_t)n2, -1);
  // the windows do not depend on the matching state: all GetFeaturesInArea calls first
  std::vector<float> qx, qy, qr;
  std::vector<int> lo, hi, qi;
  for (int i1 = 0; i1 < n1; ++i1) {
    if (k1oct[i1] > 0) continue;
    qx.push_back(prev_matched[2 * i1]); qy.push_back(prev_matched[2 * i1 + 1]); qr.push_back((float)window_size); lo.push_back(0); hi.push_back(0); qi.push_back(i1);
  }
  const int nq = (int)qi.size();
  std::vector<int> off((size_t)nq + 1, 0), idx((size_t)256 * nq + 1024);
  const int tot = orc_features_in_area(cam, n2, k2x, k2y, k2oct, nq, qx.data(), qy.data(), qr.data(), lo.data(), hi.data(), off.data(), idx.data(), (int)idx.size());
  if (tot > (int)idx.size()) {
    idx.resize((size_t)tot);
    orc_features_in_area(cam, n2, k2x, k2y, k2oct, nq, qx.data(), qy.data(), qr.data(), lo.data(), hi.data(), off.data(), idx.data(), (int)idx.size());
  }
  for (int q = 0; q < nq; ++q) {
    const int i1 = qi[q];
    if (off[q + 1] == off[q]) continue;
    int bestDist = 0x7FFFFFFF, bestDist2 = 0x7FFFFFFF, bestIdx2 = -1;
    for (int c = off[q]; c < off[q + 1]; ++c) {
      const int i2 = idx[c];
      const int dist = orc_descriptor_distance(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= TH_LOW) {
      if ((float)bestDist < (float)bestDist2 * nnratio) {
        if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; --nmatches; }
        matches12[i1] = bestIdx2; vnMatches21[bestIdx2] = i1; vMatchedDistance[bestIdx2] = bestDist; ++nmatches;
        if (check_orientation) {
          float rot = k1angle[i1] - k2angle[bestIdx2];
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == nBins) bin = 0;
          rotHist[bin].push_back(i1);
        }
      }
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
      for (int i1 : rotHist[i]) if (matches12[i1] >= 0) { matches12[i1] = -1; --nmatches; }
    }
  }
  for (int i1 = 0; i1 < n1; ++i1)
    if (matches12[i1] >= 0) { prev_matched[2 * i1] = k2x[matches12[i1]]; prev_matched[2 * i1 + 1] = k2y[matches12[i1]]; }
  return nmatches;
}
