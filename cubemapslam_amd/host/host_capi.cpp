// host_capi.cpp -- flat C entry points around the C++ mirror classes (cubemap_hot_path.h) so the Python test-suite can
// drive them exactly the way a C++ caller (Tracking / LocalMapping) would.  Test plumbing, not part of the boundary.
#include <cstring>
#include <stdexcept>
#include "cubemap_hot_path.h"
#include "io_formats.h"
using namespace CubemapSLAM;

static thread_local std::string g_err;
extern "C" const char* hm_last_error() { return g_err.c_str(); }
#define HM_TRY(...) try { __VA_ARGS__ } catch (const std::exception& e) { g_err = e.what(); return -1; }

extern "C" int hm_set_camera(const cms_camera* c) {
  HM_TRY(
    const double cde[5] = {c->c, c->d, c->e, c->u0, c->v0};
    std::vector<double> pol(c->pol, c->pol + 5), inv(c->invpol, c->invpol + 12);
    const double h = c->face / 2.0;
    CamModelGeneral::GetCamera()->SetCamParams(cde, pol, inv, c->Iw, c->Ih, h, h, h, h, c->face, c->face, c->fov_deg);
    return 0;)
}
extern "C" int hm_remap(const uint8_t* fish, int fstride, uint8_t* cube, int cstride) {
  HM_TRY(
    CamModelGeneral* cam = CamModelGeneral::GetCamera();
    System sys;
    sys.CreateUndistortRectifyMap();
    cv::Mat f(cam->GetFisheyeHeight(), cam->GetFisheyeWidth(), cv::CV_8U, (void*)fish, (size_t)fstride);
    const int W = 3 * cam->GetCubeFaceWidth();
    cv::Mat c(W, W, cv::CV_8U, cube, (size_t)cstride);
    sys.CvtFisheyeToCubeMap_reverseQuery_withInterpolation(c, f, cv::INTER_LINEAR);
    return 0;)
}
extern "C" int hm_extract(int nfeatures, float scale, int nlevels, int ini_th, int min_th, const uint8_t* img, int stride,
                          const uint8_t* mask, int mstride, cms_keypoint* kps, uint8_t* desc, int cap) {
  HM_TRY(
    const int W = 3 * CamModelGeneral::GetCamera()->GetCubeFaceWidth();
    ORBextractor ex(nfeatures, scale, nlevels, ini_th, min_th);
    cv::Mat image(W, W, cv::CV_8U, (void*)img, (size_t)stride), m(W, W, cv::CV_8U, (void*)mask, (size_t)mstride), d;
    std::vector<cv::KeyPoint> keys;
    ex(image, m, keys, d);
    const int n = (int)keys.size();
    for (int i = 0; i < n && i < cap; ++i) {
      kps[i] = cms_keypoint{keys[i].pt.x, keys[i].pt.y, keys[i].size, keys[i].angle, keys[i].response, keys[i].octave};
      std::memcpy(desc + (size_t)i * 32, d.ptr<uint8_t>(i), 32);
    }
    return n;)
}
// ORBextractor::mvImagePyramid (ORBExtractor.h:89): extract with mbKeepImagePyramid and hand level `level` back (w x h written to wh)
extern "C" int hm_extract_pyramid_level(int nfeatures, float scale, int nlevels, int ini_th, int min_th, const uint8_t* img, int stride, const uint8_t* mask,
                                        int mstride, int level, uint8_t* out, int out_stride, int* wh) {
  HM_TRY(
    const int W = 3 * CamModelGeneral::GetCamera()->GetCubeFaceWidth();
    ORBextractor ex(nfeatures, scale, nlevels, ini_th, min_th);
    ex.mbKeepImagePyramid = true;
    cv::Mat image(W, W, cv::CV_8U, (void*)img, (size_t)stride), m(W, W, cv::CV_8U, (void*)mask, (size_t)mstride), d;
    std::vector<cv::KeyPoint> keys;
    ex(image, m, keys, d);
    if (level < 0 || level >= (int)ex.mvImagePyramid.size()) return -1;
    const cv::Mat& lv = ex.mvImagePyramid[level];
    wh[0] = lv.cols; wh[1] = lv.rows;
    for (int r = 0; r < lv.rows; ++r) std::memcpy(out + (size_t)r * out_stride, lv.ptr<uint8_t>(r), (size_t)lv.cols);
    return (int)keys.size();)
}
// frame-to-frame SearchByProjection: returns matches; cur_mp[j] receives the matched map-point id or -1
extern "C" int hm_search_by_projection(int ncur, const cms_keypoint* cur_k, const uint8_t* cur_d, long* cur_mp, int nlast,
                                       const cms_keypoint* last_k, const uint8_t* last_d, const long* last_mp, const float* proj_xy,
                                       const float* scale_factors, int nlevels, float th, float nnratio, int check_ori) {
  HM_TRY(
    FrameView cur, last;
    auto fill = [](FrameView& f, int n, const cms_keypoint* k, const uint8_t* d) {
      f.mvKeys.resize(n);
      f.mDescriptors.create(n > 0 ? n : 1, 32, cv::CV_8U);
      for (int i = 0; i < n; ++i) {
        f.mvKeys[i].pt = cv::Point2f(k[i].x, k[i].y); f.mvKeys[i].angle = k[i].angle; f.mvKeys[i].octave = k[i].octave;
        std::memcpy(f.mDescriptors.ptr<uint8_t>(i), d + (size_t)i * 32, 32);
      }
    };
    fill(cur, ncur, cur_k, cur_d); fill(last, nlast, last_k, last_d);
    cur.mvpMapPoints.assign(cur_mp, cur_mp + ncur);
    last.mvpMapPoints.assign(last_mp, last_mp + nlast);
    last.mvbOutlier.assign(nlast, 0);
    last.projInCurrent.resize(nlast);
    for (int i = 0; i < nlast; ++i) last.projInCurrent[i] = cv::Point2f(proj_xy[2 * i], proj_xy[2 * i + 1]);
    cur.mvScaleFactors.assign(scale_factors, scale_factors + nlevels);
    ORBMatcher matcher(nnratio, check_ori != 0);
    const int n = matcher.SearchByProjection(cur, last, th, true);
    for (int j = 0; j < ncur; ++j) cur_mp[j] = cur.mvpMapPoints[j];
    return n;)
}
// frame-to-frame SearchByProjection, device path: the last frame's map points come with position + descriptor, the current frame with its pose
extern "C" int hm_search_by_projection_pose(int ncur, const cms_keypoint* cur_k, const uint8_t* cur_d, long* cur_mp, float* Tcw, int nlast,
                                            const cms_keypoint* last_k, const long* last_mp, const uint8_t* last_outlier, const float* mp_pos,
                                            const uint8_t* mp_desc, float th, int check_ori) {
  HM_TRY(
    FrameView cur, last;
    cur.mvKeys.resize(ncur); cur.mDescriptors.create(ncur > 0 ? ncur : 1, 32, cv::CV_8U);
    for (int i = 0; i < ncur; ++i) {
      cur.mvKeys[i].pt = cv::Point2f(cur_k[i].x, cur_k[i].y); cur.mvKeys[i].angle = cur_k[i].angle; cur.mvKeys[i].octave = cur_k[i].octave;
      std::memcpy(cur.mDescriptors.ptr<uint8_t>(i), cur_d + (size_t)i * 32, 32);
    }
    cur.mvpMapPoints.assign(cur_mp, cur_mp + ncur);
    cur.mTcw = cv::Mat(4, 4, cv::CV_32F, Tcw, 16);
    last.mvKeys.resize(nlast); last.mvMapPointPos.resize(nlast); last.mMapPointDescriptors.create(nlast > 0 ? nlast : 1, 32, cv::CV_8U);
    for (int i = 0; i < nlast; ++i) {
      last.mvKeys[i].pt = cv::Point2f(last_k[i].x, last_k[i].y); last.mvKeys[i].angle = last_k[i].angle; last.mvKeys[i].octave = last_k[i].octave;
      for (int c = 0; c < 3; ++c) last.mvMapPointPos[i].v[c] = mp_pos[3 * (size_t)i + c];
      std::memcpy(last.mMapPointDescriptors.ptr<uint8_t>(i), mp_desc + (size_t)i * 32, 32);
    }
    last.mvpMapPoints.assign(last_mp, last_mp + nlast);
    last.mvbOutlier.assign(last_outlier, last_outlier + nlast);
    ORBMatcher matcher(0.9f, check_ori != 0);
    const int n = matcher.SearchByProjection(cur, last, th, true);
    for (int j = 0; j < ncur; ++j) cur_mp[j] = cur.mvpMapPoints[j];
    return n;)
}
// Tracking::SearchLocalPoints: Tcw 16 floats (row major 4x4); map points as flat arrays; outputs per map point + cur_mp[j] updated
extern "C" int hm_search_local_points(int ncur, const cms_keypoint* cur_k, const uint8_t* cur_d, long* cur_mp, const float* scale_factors, int nlevels,
                                      float* Tcw, int nmp, const long* mp_id, const float* pos, const float* normal, const float* min_dist,
                                      const float* max_dist, const uint8_t* mp_desc, float th, float nnratio, uint8_t* in_view, float* proj_xy,
                                      int* level, float* view_cos) {
  HM_TRY(
    FrameView cur;
    cur.mvKeys.resize(ncur);
    cur.mDescriptors.create(ncur > 0 ? ncur : 1, 32, cv::CV_8U);
    for (int i = 0; i < ncur; ++i) {
      cur.mvKeys[i].pt = cv::Point2f(cur_k[i].x, cur_k[i].y); cur.mvKeys[i].angle = cur_k[i].angle; cur.mvKeys[i].octave = cur_k[i].octave;
      std::memcpy(cur.mDescriptors.ptr<uint8_t>(i), cur_d + (size_t)i * 32, 32);
    }
    cur.mvpMapPoints.assign(cur_mp, cur_mp + ncur);
    cur.mvScaleFactors.assign(scale_factors, scale_factors + nlevels);
    cur.mTcw = cv::Mat(4, 4, cv::CV_32F, Tcw, 16);
    std::vector<MapPointView> mps(nmp);
    for (int i = 0; i < nmp; ++i) {
      mps[i].mnId = mp_id[i];
      mps[i].mWorldPos = cv::Mat(3, 1, cv::CV_32F, const_cast<float*>(pos) + 3 * (size_t)i, 4);
      mps[i].mNormalVector = cv::Mat(3, 1, cv::CV_32F, const_cast<float*>(normal) + 3 * (size_t)i, 4);
      mps[i].mfMinDistance = min_dist[i]; mps[i].mfMaxDistance = max_dist[i];
      mps[i].mDescriptor = cv::Mat(1, 32, cv::CV_8U, const_cast<uint8_t*>(mp_desc) + 32 * (size_t)i, 32);
    }
    const int n = Tracking::SearchLocalPoints(cur, mps, th, nnratio);
    for (int j = 0; j < ncur; ++j) cur_mp[j] = cur.mvpMapPoints[j];
    for (int i = 0; i < nmp; ++i) {
      in_view[i] = mps[i].mbTrackInView; proj_xy[2 * i] = mps[i].mTrackProjX; proj_xy[2 * i + 1] = mps[i].mTrackProjY;
      level[i] = mps[i].mnTrackScaleLevel; view_cos[i] = mps[i].mTrackViewCos;
    }
    return n;)
}
// ORBMatcher::SearchByProjection(F, vpMapPoints, th) through the mirror: the caller marks the points in view (what Frame::isInFrustum left in them)
extern "C" int hm_search_by_projection_map(int ncur, const cms_keypoint* cur_k, const uint8_t* cur_d, long* cur_mp, const float* scale_factors, int nlevels,
                                           float* Tcw, int nmp, const long* mp_id, const float* pos, const float* normal, const float* min_dist,
                                           const float* max_dist, const uint8_t* mp_desc, const uint8_t* in_view, float th, float nnratio) {
  HM_TRY(
    FrameView cur;
    cur.mvKeys.resize(ncur);
    cur.mDescriptors.create(ncur > 0 ? ncur : 1, 32, cv::CV_8U);
    for (int i = 0; i < ncur; ++i) {
      cur.mvKeys[i].pt = cv::Point2f(cur_k[i].x, cur_k[i].y); cur.mvKeys[i].angle = cur_k[i].angle; cur.mvKeys[i].octave = cur_k[i].octave;
      std::memcpy(cur.mDescriptors.ptr<uint8_t>(i), cur_d + (size_t)i * 32, 32);
    }
    cur.mvpMapPoints.assign(cur_mp, cur_mp + ncur);
    cur.mvScaleFactors.assign(scale_factors, scale_factors + nlevels);
    cur.mTcw = cv::Mat(4, 4, cv::CV_32F, Tcw, 16);
    std::vector<MapPointView> mps(nmp);
    for (int i = 0; i < nmp; ++i) {
      mps[i].mnId = mp_id[i];
      mps[i].mWorldPos = cv::Mat(3, 1, cv::CV_32F, const_cast<float*>(pos) + 3 * (size_t)i, 4);
      mps[i].mNormalVector = cv::Mat(3, 1, cv::CV_32F, const_cast<float*>(normal) + 3 * (size_t)i, 4);
      mps[i].mfMinDistance = min_dist[i]; mps[i].mfMaxDistance = max_dist[i];
      mps[i].mDescriptor = cv::Mat(1, 32, cv::CV_8U, const_cast<uint8_t*>(mp_desc) + 32 * (size_t)i, 32);
      mps[i].mbTrackInView = in_view[i] != 0;
    }
    ORBMatcher matcher(nnratio, true);
    const int n = matcher.SearchByProjection(cur, mps, th);
    for (int j = 0; j < ncur; ++j) cur_mp[j] = cur.mvpMapPoints[j];
    return n;)
}
// ORBMatcher::SearchForInitialization through the mirror; prev_xy (n1 x 2) is vbPrevMatched, in / out
extern "C" int hm_search_for_initialization(int n1, const cms_keypoint* k1, const uint8_t* d1, int n2, const cms_keypoint* k2, const uint8_t* d2,
                                            float* prev_xy, int window, float nnratio, int check_ori, int* matches12) {
  HM_TRY(
    FrameView f1, f2;
    auto fill = [](FrameView& f, int n, const cms_keypoint* k, const uint8_t* d) {
      f.mvKeys.resize(n);
      f.mDescriptors.create(n > 0 ? n : 1, 32, cv::CV_8U);
      for (int i = 0; i < n; ++i) {
        f.mvKeys[i].pt = cv::Point2f(k[i].x, k[i].y); f.mvKeys[i].angle = k[i].angle; f.mvKeys[i].octave = k[i].octave; f.mvKeys[i].size = k[i].size;
        f.mvKeys[i].response = k[i].response;
        std::memcpy(f.mDescriptors.ptr<uint8_t>(i), d + (size_t)i * 32, 32);
      }
    };
    fill(f1, n1, k1, d1); fill(f2, n2, k2, d2);
    std::vector<cv::Point2f> prev(n1);
    for (int i = 0; i < n1; ++i) prev[i] = cv::Point2f(prev_xy[2 * i], prev_xy[2 * i + 1]);
    std::vector<int> m12;
    ORBMatcher matcher(nnratio, check_ori != 0);
    const int n = matcher.SearchForInitialization(f1, f2, prev, m12, window);
    for (int i = 0; i < n1; ++i) { matches12[i] = m12[i]; prev_xy[2 * i] = prev[i].x; prev_xy[2 * i + 1] = prev[i].y; }
    return n;)
}
// ORBMatcher::SearchForTriangulation through the mirror: two key frames in hm_create_new_map_points' flat layout (nkf = 2), E12 row major
extern "C" int hm_search_for_triangulation(const int* feat_off, const cms_keypoint* kps, const uint8_t* desc, const float* rays, const long* mp,
                                           float* Tcw, const int* node_off2, const int* node_id, const int* node_cnt, const int* node_feat,
                                           float* E12, int check_ori, int cap, int* out_idx1, int* out_idx2) {
  HM_TRY(
    std::vector<KeyFrameView> kfs(2);
    size_t nf_cursor = 0;
    for (int k = 0; k < 2; ++k) {
      KeyFrameView& v = kfs[k];
      const int f0 = feat_off[k], n = feat_off[k + 1] - f0;
      v.mnId = k; v.mvKeys.resize(n); v.mvKeyRays.resize(n); v.mvpMapPoints.assign(mp + f0, mp + f0 + n);
      v.mDescriptors.create(n > 0 ? n : 1, 32, cv::CV_8U);
      for (int i = 0; i < n; ++i) {
        const cms_keypoint& s = kps[f0 + i];
        v.mvKeys[i].pt = cv::Point2f(s.x, s.y); v.mvKeys[i].angle = s.angle; v.mvKeys[i].octave = s.octave;
        std::memcpy(v.mDescriptors.ptr<uint8_t>(i), desc + 32 * (size_t)(f0 + i), 32);
        for (int c = 0; c < 3; ++c) v.mvKeyRays[i].v[c] = rays[3 * (size_t)(f0 + i) + c];
      }
      v.Tcw = cv::Mat(4, 4, cv::CV_32F, Tcw + 16 * (size_t)k, 16);
      for (int e = node_off2[k]; e < node_off2[k + 1]; ++e) {
        std::vector<unsigned> fl(node_feat + nf_cursor, node_feat + nf_cursor + node_cnt[e]);
        nf_cursor += (size_t)node_cnt[e];
        v.mFeatVec.emplace_back((unsigned)node_id[e], std::move(fl));
      }
    }
    std::vector<std::pair<size_t, size_t>> pairs;
    ORBMatcher matcher(0.6f, check_ori != 0);
    const int n = matcher.SearchForTriangulation(kfs[0], kfs[1], cv::Mat(3, 3, cv::CV_32F, E12, 12), pairs);
    for (int i = 0; i < (int)pairs.size() && i < cap; ++i) { out_idx1[i] = (int)pairs[i].first; out_idx2[i] = (int)pairs[i].second; }
    return n;)
}
// LocalMapping::CreateNewMapPoints through the mirror: key frame 0 is the current one, 1..nkf-1 its neighbours.  Flat inputs:
// feat_off[nkf+1]; per feature kps/desc/rays/mp; Tcw nkf x 16; FeatureVector per key frame as node_off2[nkf+1] into (node_id, node_cnt)
// and the features of the nodes concatenated in node_feat; median_depth[nkf].  Outputs up to cap records.
extern "C" int hm_create_new_map_points(int nkf, const int* feat_off, const cms_keypoint* kps, const uint8_t* desc, const float* rays, const long* mp,
                                        float* Tcw, const int* node_off2, const int* node_id, const int* node_cnt, const int* node_feat,
                                        const float* median_depth, int cap, int* out_neigh, int* out_idx1, int* out_idx2, float* out_x3d) {
  HM_TRY(
    std::vector<KeyFrameView> kfs(nkf);
    size_t nf_cursor = 0;
    for (int k = 0; k < nkf; ++k) {
      KeyFrameView& v = kfs[k];
      const int f0 = feat_off[k], n = feat_off[k + 1] - f0;
      v.mnId = k; v.mvKeys.resize(n); v.mvKeyRays.resize(n); v.mvpMapPoints.assign(mp + f0, mp + f0 + n);
      v.mDescriptors.create(n > 0 ? n : 1, 32, cv::CV_8U);
      for (int i = 0; i < n; ++i) {
        const cms_keypoint& s = kps[f0 + i];
        v.mvKeys[i].pt = cv::Point2f(s.x, s.y); v.mvKeys[i].angle = s.angle; v.mvKeys[i].octave = s.octave;
        std::memcpy(v.mDescriptors.ptr<uint8_t>(i), desc + 32 * (size_t)(f0 + i), 32);
        for (int c = 0; c < 3; ++c) v.mvKeyRays[i].v[c] = rays[3 * (size_t)(f0 + i) + c];
      }
      v.Tcw = cv::Mat(4, 4, cv::CV_32F, Tcw + 16 * (size_t)k, 16);
      for (int e = node_off2[k]; e < node_off2[k + 1]; ++e) {
        std::vector<unsigned> fl(node_feat + nf_cursor, node_feat + nf_cursor + node_cnt[e]);
        nf_cursor += (size_t)node_cnt[e];
        v.mFeatVec.emplace_back((unsigned)node_id[e], std::move(fl));
      }
      v.medianDepth = median_depth[k];
    }
    std::vector<const KeyFrameView*> neigh;
    for (int k = 1; k < nkf; ++k) neigh.push_back(&kfs[k]);
    const std::vector<NewMapPoint> pts = LocalMapping::CreateNewMapPoints(kfs[0], neigh);
    const int n = (int)pts.size();
    for (int i = 0; i < n && i < cap; ++i) {
      out_neigh[i] = pts[i].neighbour; out_idx1[i] = pts[i].idx1; out_idx2[i] = pts[i].idx2;
      for (int c = 0; c < 3; ++c) out_x3d[3 * (size_t)i + c] = pts[i].x3D.v[c];
    }
    return n;)
}
// ORBMatcher::Fuse through the mirror (ORBMatcher.cpp:1127-1244): the key frame as flat arrays (n key points, descriptors, the map-point slot per key
// point: in / out), M map points (position, normal, distance range, descriptor, id), skip[M] (pMP->isBad() / IsInKeyFrame, may be NULL).  fused[M] = the
// key point every map point is fused with or -1; mp_slots receives the ADDITIONS (AddMapPoint for a free key point, in list order); for a key point
// that already holds a point the caller decides the Replace by Observations() like the reference does.  Returns nFused.
extern "C" int hm_fuse(int n, const cms_keypoint* kps, const uint8_t* desc, long* mp_slots, float* Tcw, int M, const float* pos, const float* normal,
                       const float* min_dist, const float* max_dist, const uint8_t* mp_desc, const long* mp_id, const uint8_t* skip, float th, int* fused) {
  HM_TRY(
    KeyFrameView v;
    v.mnId = 0; v.mvKeys.resize(n); v.mvpMapPoints.assign(mp_slots, mp_slots + n);
    v.mDescriptors.create(n > 0 ? n : 1, 32, cv::CV_8U);
    for (int i = 0; i < n; ++i) {
      v.mvKeys[i].pt = cv::Point2f(kps[i].x, kps[i].y); v.mvKeys[i].angle = kps[i].angle; v.mvKeys[i].octave = kps[i].octave;
      v.mvKeys[i].size = kps[i].size; v.mvKeys[i].response = kps[i].response;
      std::memcpy(v.mDescriptors.ptr<uint8_t>(i), desc + 32 * (size_t)i, 32);
    }
    v.Tcw = cv::Mat(4, 4, cv::CV_32F, Tcw, 16);
    std::vector<MapPointView> mps(M);
    std::vector<float> store(6 * (size_t)std::max(M, 1));
    std::vector<uint8_t> dstore(32 * (size_t)std::max(M, 1));
    for (int i = 0; i < M; ++i) {
      for (int c = 0; c < 3; ++c) { store[6 * (size_t)i + c] = pos[3 * (size_t)i + c]; store[6 * (size_t)i + 3 + c] = normal[3 * (size_t)i + c]; }
      std::memcpy(&dstore[32 * (size_t)i], mp_desc + 32 * (size_t)i, 32);
      mps[i].mnId = mp_id[i];
      mps[i].mWorldPos = cv::Mat(3, 1, cv::CV_32F, &store[6 * (size_t)i], 4);
      mps[i].mNormalVector = cv::Mat(3, 1, cv::CV_32F, &store[6 * (size_t)i + 3], 4);
      mps[i].mfMinDistance = min_dist[i]; mps[i].mfMaxDistance = max_dist[i];
      mps[i].mDescriptor = cv::Mat(1, 32, cv::CV_8U, &dstore[32 * (size_t)i], 32);
    }
    std::vector<uint8_t> sk;
    if (skip) sk.assign(skip, skip + M);
    std::vector<int> f;
    ORBMatcher matcher;
    const int nf = matcher.Fuse(v, mps, sk, th, f);
    for (int i = 0; i < M; ++i) fused[i] = f[i];
    for (int i = 0; i < n; ++i) mp_slots[i] = v.mvpMapPoints[i];
    return nf;)
}
// local BA through the Optimizer mirror.  Tcw: K x 16 float (row major 4x4), Xw: P x 3 float, observations flat.
extern "C" int hm_local_ba(int K, float* Tcw, const long* kf_id, const uint8_t* kf_fixed, const float* inv_sigma2, int nlevels, int P,
                           float* Xw, int nobs, const int* obs_kf, const int* obs_mp, const cms_keypoint* obs_kp, const float* obs_ray,
                           uint8_t* stop, int* erase_pairs, int erase_cap) {
  HM_TRY(
    LocalBAWindow w;
    w.keyframes.resize(K);
    for (int k = 0; k < K; ++k) {
      w.keyframes[k].mnId = kf_id[k]; w.keyframes[k].fixed = kf_fixed[k] != 0;
      w.keyframes[k].Tcw = cv::Mat(4, 4, cv::CV_32F, Tcw + 16 * (size_t)k, 16);
      w.keyframes[k].mvInvLevelSigma2.assign(inv_sigma2, inv_sigma2 + nlevels);
    }
    w.mappoints.resize(P);
    for (int p = 0; p < P; ++p) { w.mappoints[p].mnId = p; w.mappoints[p].Xw = cv::Mat(3, 1, cv::CV_32F, Xw + 3 * (size_t)p, 4); }
    for (int o = 0; o < nobs; ++o) {
      LocalBAWindow::Obs ob;
      ob.kf = obs_kf[o];
      ob.kp.pt = cv::Point2f(obs_kp[o].x, obs_kp[o].y); ob.kp.octave = obs_kp[o].octave;
      ob.ray.v[0] = obs_ray[3 * o]; ob.ray.v[1] = obs_ray[3 * o + 1]; ob.ray.v[2] = obs_ray[3 * o + 2];
      w.mappoints[obs_mp[o]].observations.push_back(ob);
    }
    Optimizer::LocalBundleAdjustment(&w, reinterpret_cast<bool*>(stop));
    const int n = (int)w.toErase.size();
    for (int i = 0; i < n && i < erase_cap; ++i) { erase_pairs[2 * i] = w.toErase[i].first; erase_pairs[2 * i + 1] = w.toErase[i].second; }
    return n;)
}

// Optimizer::PoseOptimization on a Frame given as flat arrays: Tcw 4x4 float in/out, N key points (x, y, octave), rays N x 3,
// has_mp N, Xw N x 3 (float), inv_sigma2[nlevels]; outlier N out.  Returns the reference's return value.
extern "C" int hm_pose_optimization(float* Tcw, int N, const cms_keypoint* kps, const float* rays, const uint8_t* has_mp, const float* Xw,
                                    const float* inv_sigma2, int nlevels, uint8_t* outlier) {
  HM_TRY(
    PoseFrame fr;
    fr.mTcw = cv::Mat(4, 4, cv::CV_32F, Tcw, 16);
    fr.N = N;
    fr.mvKeys.resize(N); fr.mvKeyRays.resize(N); fr.mvbHasMapPoint.resize(N); fr.mvMapPointPos.resize(N);
    for (int i = 0; i < N; ++i) {
      fr.mvKeys[i].pt = cv::Point2f(kps[i].x, kps[i].y); fr.mvKeys[i].octave = kps[i].octave;
      for (int k = 0; k < 3; ++k) { fr.mvKeyRays[i].v[k] = rays[3 * i + k]; fr.mvMapPointPos[i].v[k] = Xw[3 * i + k]; }
      fr.mvbHasMapPoint[i] = has_mp[i] != 0;
    }
    fr.mvInvLevelSigma2.assign(inv_sigma2, inv_sigma2 + nlevels);
    const int r = Optimizer::PoseOptimization(&fr);
    for (int i = 0; i < N; ++i) outlier[i] = (i < (int)fr.mvbOutlier.size() && fr.mvbOutlier[i]) ? 1 : 0;
    return r;)
}

// System::SaveKeyFrameTrajectoryTUM on n key frames given as time stamps + 4x4 float Tcw
extern "C" int hm_save_trajectory_tum(const char* path, int n, const double* ts, float* Tcw) {
  HM_TRY(
    std::vector<TrajectoryKeyFrame> v(n);
    for (int i = 0; i < n; ++i) { v[i].mTimeStamp = ts[i]; v[i].Tcw = cv::Mat(4, 4, cv::CV_32F, Tcw + 16 * (size_t)i, 16); }
    System::SaveKeyFrameTrajectoryTUM(path, v);
    return n;)
}

// ---- file formats (io_formats.h); host only, usable without a GPU
extern "C" int hm_settings_load(const char* path, cms_camera* cam, cms_orb_params* orb, float* fps, int* with_mask, int* rgb) {
  HM_TRY(
    Settings st;
    if (!st.Load(path)) throw std::runtime_error(std::string("Failed to open settings file at: ") + path);
    if (cam) *cam = st.Camera();
    if (orb) *orb = st.Orb();
    if (fps) *fps = st.Fps();
    if (with_mask) *with_mask = st.WithFisheyeMask();
    if (rgb) *rgb = st.RGB() ? 1 : 0;
    return 0;)
}
// kind 0 = Lafida list, 1 = Fangshan list.  names: cap x name_len chars (NUL terminated).  Returns the number of entries.
extern "C" int hm_load_image_list(const char* path, int kind, int cap, int name_len, char* names, double* timestamps) {
  HM_TRY(
    const ImageList l = kind == 0 ? LoadImageListLafida(path) : LoadImageListFangshan(path);
    const int n = (int)l.names.size();
    for (int i = 0; i < n && i < cap; ++i) {
      std::strncpy(names + (size_t)i * name_len, l.names[i].c_str(), name_len - 1);
      names[(size_t)i * name_len + name_len - 1] = 0;
      timestamps[i] = l.timestamps[i];
    }
    return n;)
}
extern "C" int hm_write_tracking_summary(const char* path, float* times, int n, int frame_counter, char* console, int console_len) {
  HM_TRY(
    std::vector<float> v(times, times + n);
    const std::string text = WriteTrackingSummary(path ? path : "", v, frame_counter);
    std::copy(v.begin(), v.end(), times);
    if (console && console_len > 0) { std::strncpy(console, text.c_str(), console_len - 1); console[console_len - 1] = 0; }
    return 0;)
}
